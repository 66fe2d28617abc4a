// k_biquad_pipe.hip -- the pan/attenuation ramp + time-varying high-shelf of k_biquad_mix.hip for SMALL callbacks
// (a few hundred sources: BASELINE config 2), where there are far fewer recurrences than SIMDs and the launch is
// bound by the length of ONE lane's dependent chain, not by memory.
//
// Same arithmetic, same rounding points, same state as k_biquad_mix (audio_spatializer_3d.cpp:554-609 / :491-552 /
// the engine's AudioEffectFilterInstance): outputs are bitwise identical.  What changes is who does what.  The
// reference's step
//     y = x*b0 + hb1*b1 + hb2*b2 + ha1*a1 + ha2*a2        (x = vol(t) * src, all five coefficients advanced per step)
// is evaluated left to right, so its first three terms u = (x*b0 + hb1*b1) + hb2*b2 depend on the INPUT only.  One
// 64-lane wave per workgroup used to carry ~30 dependent instructions per frame (k_biquad_mix: 35.7 us for 256
// sources, ~65 ns per step at one wave per SIMD); here the workgroup is a software pipeline of eight waves over
// 32-frame tiles, one barrier per tile, and the serial wave keeps only what is truly recurrent:
//
//   wave LOAD     tile p      source rows -> LDS (coalesced 16-byte loads, next tile's loads in flight)
//   wave COEF     tile p      b0, b1, b2 advanced per step (three sequential f32 adds) -> LDS
//   waves FIR x4  tile p-1    vol(t) ramp, x = vol * src, u = (x*b0 + hb1*b1) + hb2*b2 -> LDS (a quarter of the tile each)
//   wave REC      tile p-2    y = (u + ha1*a1) + ha2*a2, a1 / a2 advanced per step, running peak: ~10 VALU per frame
//   wave POST     tile p-3    sum over the 32 sources in fixed order, 256-byte partial-mix store
//
// Measured (256 sources, F = 512, MI355X): 35.4 us -> 20.2 us.  Per-role busy cycles (tools/pipe_busy_probe.py) put REC
// at ~60 cycles per frame, of which the arithmetic is ~30: the rest is LDS latency with eight waves on the CU's one
// LDS pipe.  Tried and measured slower: the coefficient ramps recomputed by every FIR wave instead of a COEF wave
// (32 us: branchy scalar loops), whole-tile read batches in REC (22.6 us), generic-pointer laundering (flat loads).
//
// Lane = (source, ear) in every wave, so per-lane state never moves between lanes; each wave loads and stores the
// part of SpatializerPlaybackData3D (audio_spatializer_3d.h:85-99) it owns.  Bound: the REC wave's chain (~35 cycles
// per frame); HBM traffic is that of k_biquad_mix.  Used when the grid has no more workgroups than the chip has CUs;
// larger callbacks are bandwidth-bound and stay on k_biquad_mix (one wave per 32 sources, many per CU).
#include "gas_biquad.h"
#include <type_traits>

// No FMA contraction: see k_biquad_mix.hip (poles near the unit circle amplify a fused multiply-add's rounding).
#pragma clang fp contract(off)

namespace {

constexpr int SRC_PER_WG = 32;
constexpr int KF = 32; // frames per tile
constexpr int COLS = KF * 2; // floats per source per tile
constexpr int ROW = COLS + 2; // staged tile: row stride in floats, + 2 pad -> conflict-free column reads (k_biquad_mix.hip)
constexpr int PARTS = COLS / 4; // 16-byte pieces per source row per tile
constexpr int LOADS = SRC_PER_WG * PARTS / 64; // staging loads per lane per tile
constexpr int YROW = 66; // y tile: [frame][64 lanes] with stride 66 -> POST's column reads (fixed lane pair, all frames) hit 32 banks
constexpr int N_FIR = 4;
constexpr int FIR_FRAMES = KF / N_FIR; // frames of a tile per FIR wave
enum Role { R_REC = 0, R_COEF = 1, R_FIR0 = 2, R_LOAD = 6, R_POST = 7, N_WAVES = 8 };

#ifdef GAS_STAMPS
__device__ unsigned long long gas_pipe_busy[16]; // diagnostic builds: per role, cycles busy / cycles in the loop
#endif

// LDS pointers with their address space spelled out and their value hidden from constant folding: the LDS image is
// larger than a DS instruction's 16-bit offset field, and left to itself the compiler addresses every access as (one
// base register + a large constant), i.e. one extra VALU add per LDS instruction.  With a tile's base in its own
// register the per-frame displacements fit the immediate field.  (Laundering a GENERIC pointer would turn the
// accesses into flat loads -- measured: slower than the single-wave kernel.)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float *lds_f;
typedef __attribute__((address_space(3))) v2f *lds_f2;
__device__ __forceinline__ lds_f opaque(float *p) {
	lds_f q = (lds_f)p;
	asm volatile("" : "+v"(q));
	return q;
}
__device__ __forceinline__ lds_f2 opaque(float2 *p) {
	lds_f2 q = (lds_f2)reinterpret_cast<v2f *>(p);
	asm volatile("" : "+v"(q));
	return q;
}

struct PipeLds {
	float xs[2][SRC_PER_WG * ROW]; // staged source tile
	float2 cb01[2][KF * 64]; // (b0, b1) per step per lane
	float cb2[2][KF * 64];
	float2 ux[2][KF * 64]; // (u, x) per step per lane: the general form (some lane takes the bypass branch)
	float u1[2][KF * 64]; // u alone when every lane filters (half the store traffic of the hand-over)
	float yo[2][KF * YROW]; // outputs per step per lane
	float bus_w[GAS_MAX_BUSES * 64]; // several buses: [bus][source * 2 + ear] weights for this channel pair
};

template <int MODE, bool F_POW2, bool ALL_FILT>
__global__ __launch_bounds__(N_WAVES * 64) void k_biquad_pipe(gas_group_args g, gas_dev_state st, uint32_t F, uint32_t c0, float mix_rate, float *__restrict__ partials, uint32_t p_offset, uint32_t p_stride, gas_bus_args buses) {
	__shared__ PipeLds L;
	const int lane = threadIdx.x & 63;
	const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int ear = lane & 1;
	const int sl = lane >> 1;
	const uint32_t c = c0 + blockIdx.y;
	const uint32_t e = blockIdx.x * SRC_PER_WG + sl;
	const bool valid = e < g.n;
	const uint32_t ec = valid ? e : g.n - 1;
	const uint32_t n_tiles = F / KF;
	const uint32_t n_valid = g.n - blockIdx.x * SRC_PER_WG < SRC_PER_WG ? g.n - blockIdx.x * SRC_PER_WG : SRC_PER_WG; // sources of this workgroup

	// ---- per-lane state: every wave looks its stream up; each loads only what its role owns -------------------------
	const uint32_t slot = g.slots[ec];
	const uint32_t row = g.rows ? g.rows[ec] : ec;
	const gas_params *P = st.params + slot;
	const size_t stream = ((size_t)slot * 4 + c) * 2 + ear;
	float *bq = st.bq;
	const size_t bs = st.bq_stride;
	bool filt = true;
	bool just_started = false;
	Coeffs target = { 0, 0, 0, 0, 0 };
	float vs = 0.0f, vf = 0.0f;
	if constexpr (MODE == GAS_MODE_MIX_CHANNEL || MODE == GAS_MODE_PROCESS_FRAMES) {
		vs = bq[BQ_PREV * bs + stream]; // get_prev_mix_volume(c), (0,0) when never set (:880-885)
		const float vs_other = bq[BQ_PREV * bs + (stream ^ 1)];
		const float gain = P->linear_attenuation;
		filt = (double)gain >= 0.001; // :503 / :568
		just_started = vs == 0 && vs_other == 0; // is_just_started -> clear_history (:518-521, :583-587)
		if (filt && (role == R_REC || role == R_COEF)) {
			target = highshelf_coeffs(mix_rate, P->attenuation_filter_cutoff_hz, gain);
		}
		if constexpr (MODE == GAS_MODE_MIX_CHANNEL) {
			vf = P->mix_volumes[c][ear]; // :565
		} else {
			// prev_mix_volume(0) = the channel pair holding the largest component (:537-551)
			float max_volume = 0.0f;
			int max_index = 0;
#pragma unroll
			for (int i = 0; i < GAS_MAX_CHANNELS_PER_BUS; i++) {
				if (P->mix_volumes[i][0] > max_volume) {
					max_volume = P->mix_volumes[i][0];
					max_index = i;
				}
				if (P->mix_volumes[i][1] > max_volume) {
					max_volume = P->mix_volumes[i][1];
					max_index = i;
				}
			}
			vf = P->mix_volumes[max_index][ear];
		}
	} else { // GAS_MODE_FX_HIGHSHELF: coefficients snap every call, process_one
		if (role == R_REC || role == R_COEF) {
			target = highshelf_coeffs(mix_rate, P->fx_shelf_cutoff_hz, P->fx_shelf_gain);
		}
	}
	const int Fi = (int)F;
	const float Ff = (float)Fi;
	const float invF = 1.0f / Ff;
	float *my_partial = partials + ((size_t)blockIdx.y * p_stride + p_offset + blockIdx.x) * (size_t)F * 2;
	const size_t bus_plane = (size_t)gridDim.y * p_stride * (size_t)F * 2; // floats between the partial planes of two buses
	const uint32_t n_buses = buses.routes ? buses.n_buses : 0; // 0: the single mix of gas_process_block
	if (n_buses > 0 && role == R_POST) { // only POST reads the weights (its own writes)
		gas_bus_route r = gas_bus_route_default();
		if (valid) {
			r = buses.routes[slot];
		}
		for (uint32_t b = 0; b < n_buses; b++) {
			L.bus_w[b * 64 + lane] = valid ? gas_bus_weight(r, b, (int)c, ear) : 0.0f;
		}
	}

	// ---- role state ------------------------------------------------------------------------------------------------------
	float b0 = 0, b1 = 0, b2 = 0, ib0 = 0, ib1 = 0, ib2 = 0; // COEF
	float a1 = 0, a2 = 0, ia1 = 0, ia2 = 0, ha1 = 0, ha2 = 0; // REC
	float hb1 = 0, hb2 = 0; // FIR: the two inputs in front of this wave's first step of the tile
	float peak = 0.0f; // POST
	float4 pre[LOADS]; // LOAD
	const float *ld_base[LOADS];
	const bool allf = ALL_FILT || __all(filt || !valid); // wave-uniform, and the same in every wave (same lanes)
	if (role == R_COEF) {
		if constexpr (MODE == GAS_MODE_FX_HIGHSHELF) {
			b0 = target.b0;
			b1 = target.b1;
			b2 = target.b2;
		} else {
			b0 = bq[BQ_B0 * bs + stream];
			b1 = bq[BQ_B1 * bs + stream];
			b2 = bq[BQ_B2 * bs + stream];
			if (filt) { // [ENGINE] update_coeffs(F): ramp from the current coefficients to the target
				ib0 = (target.b0 - b0) / Fi;
				ib1 = (target.b1 - b1) / Fi;
				ib2 = (target.b2 - b2) / Fi;
			}
		}
	}
	if (role == R_REC) {
		if constexpr (MODE == GAS_MODE_FX_HIGHSHELF) {
			a1 = target.a1;
			a2 = target.a2;
		} else {
			a1 = bq[BQ_A1 * bs + stream];
			a2 = bq[BQ_A2 * bs + stream];
			if (filt) {
				ia1 = (target.a1 - a1) / Fi;
				ia2 = (target.a2 - a2) / Fi;
			}
		}
		ha1 = bq[BQ_HA1 * bs + stream];
		ha2 = bq[BQ_HA2 * bs + stream];
		if (filt && just_started) {
			ha1 = ha2 = 0;
		}
	}
	// no lane's feedback coefficients move during this callback (wave-uniform; exact: a += 0.0f leaves every a but -0
	// unchanged, and a coefficient that is -0 stays a zero of either sign only in a product with history that is added to u)
	const bool rec_still = MODE == GAS_MODE_FX_HIGHSHELF || __all(ia1 == 0.0f && ia2 == 0.0f);
	if (role == R_FIR0) {
		hb1 = bq[BQ_HB1 * bs + stream];
		hb2 = bq[BQ_HB2 * bs + stream];
		if (filt && just_started) {
			hb1 = hb2 = 0;
		}
	} else if (role == R_LOAD) {
		// load q covers tile element idx = q*64 + lane -> source idx / PARTS, 16-byte part idx % PARTS
#pragma unroll
		for (int q = 0; q < LOADS; q++) {
			const int idx = q * 64 + lane;
			uint32_t le = blockIdx.x * SRC_PER_WG + (idx / PARTS);
			le = le < g.n ? le : g.n - 1;
			const uint32_t lrow = g.rows ? g.rows[le] : le;
			ld_base[q] = reinterpret_cast<const float *>(g.src) + (size_t)lrow * F * 2 + (idx % PARTS) * 4;
			pre[q] = *reinterpret_cast<const float4 *>(ld_base[q]);
		}
	}
	// the FIR waves split a tile's frames; each needs the two inputs in front of its first frame
	const int fir = role - R_FIR0; // 0 .. N_FIR-1 for the FIR waves
	const int k_begin = fir * FIR_FRAMES;

	auto volume = [&](int frame) -> float { // (float)frame_idx / p_frame_count, lerp (:591-592)
		const float fi = (float)frame;
		const float t = F_POW2 ? fi * invF : fi / Ff;
		return vf * t + (1 - t) * vs;
	};

	if (role == R_REC) {
		__builtin_amdgcn_s_setprio(3); // the serial wave sets the pace: it wins the issue arbitration on its SIMD
	}
#ifdef GAS_STAMPS
	unsigned long long busy = 0, t_loop0 = __builtin_amdgcn_s_memtime();
#endif
	for (uint32_t p = 0; p < n_tiles + 3; p++) {
#ifdef GAS_STAMPS
		const unsigned long long t_in = __builtin_amdgcn_s_memtime();
#endif
		if (role == R_LOAD) {
			if (p < n_tiles) {
				float *tb = L.xs[p & 1];
#pragma unroll
				for (int q = 0; q < LOADS; q++) { // rows are 264 B: two 8-byte stores (16-byte ones would misalign)
					const int idx = q * 64 + lane;
					float *d = tb + (idx / PARTS) * ROW + (idx % PARTS) * 4;
					*reinterpret_cast<float2 *>(d) = make_float2(pre[q].x, pre[q].y);
					*reinterpret_cast<float2 *>(d + 2) = make_float2(pre[q].z, pre[q].w);
				}
				if (p + 1 < n_tiles) {
#pragma unroll
					for (int q = 0; q < LOADS; q++) {
						pre[q] = *reinterpret_cast<const float4 *>(ld_base[q] + (size_t)(p + 1) * KF * 2);
					}
				}
			}
		} else if (role == R_COEF) {
			if (p < n_tiles) {
				lds_f2 o01 = opaque(L.cb01[p & 1] + lane);
				lds_f o2 = opaque(L.cb2[p & 1] + lane);
#pragma unroll 8
				for (int k = 0; k < KF; k++) {
					o01[k * 64] = v2f{ b0, b1 };
					o2[k * 64] = b2;
					if constexpr (MODE != GAS_MODE_FX_HIGHSHELF) {
						b0 += ib0;
						b1 += ib1;
						b2 += ib2;
					}
				}
			}
		} else if (role >= R_FIR0 && role < R_FIR0 + N_FIR) {
			const uint32_t t = p - 1;
			if (p >= 1 && t < n_tiles) {
				const lds_f mine = opaque(L.xs[t & 1] + sl * ROW + ear + 2 * k_begin);
				const int i0 = (int)(t * KF);
				if (fir != 0) { // the two inputs in front of this wave's range, recomputed (same tile)
					float xa = mine[-4], xb = mine[-2];
					if constexpr (MODE == GAS_MODE_MIX_CHANNEL) {
						xa = volume(i0 + k_begin - 2) * xa;
						xb = volume(i0 + k_begin - 1) * xb;
					}
					hb2 = xa;
					hb1 = xb;
				}
				// the LDS reads of the range first (the compiler will not move a read above a write it cannot tell apart),
				// then the arithmetic, then the writes
				const lds_f2 c01 = opaque(L.cb01[t & 1] + lane + k_begin * 64);
				const lds_f c2 = opaque(L.cb2[t & 1] + lane + k_begin * 64);
				float xr[FIR_FRAMES], c2r[FIR_FRAMES];
				v2f c01r[FIR_FRAMES];
#pragma unroll
				for (int j = 0; j < FIR_FRAMES; j++) {
					xr[j] = mine[2 * j];
					c01r[j] = c01[j * 64];
					c2r[j] = c2[j * 64];
				}
				v2f ur[FIR_FRAMES];
#pragma unroll
				for (int j = 0; j < FIR_FRAMES; j++) {
					float x = xr[j];
					if constexpr (MODE == GAS_MODE_MIX_CHANNEL) {
						x = volume(i0 + k_begin + j) * x; // :593
					}
					const float u = x * c01r[j].x + hb1 * c01r[j].y + hb2 * c2r[j]; // left to right: (x*b0 + hb1*b1) + hb2*b2
					ur[j] = v2f{ u, x };
					hb2 = hb1;
					hb1 = x;
				}
				if (allf) {
					lds_f out = opaque(L.u1[t & 1] + lane + k_begin * 64);
#pragma unroll
					for (int j = 0; j < FIR_FRAMES; j++) {
						out[j * 64] = ur[j].x;
					}
				} else {
					lds_f2 out = opaque(L.ux[t & 1] + lane + k_begin * 64);
#pragma unroll
					for (int j = 0; j < FIR_FRAMES; j++) {
						out[j * 64] = ur[j];
					}
				}
				if (fir == 0) { // carry the tile's last two inputs into the next tile's first frames
					float xa = mine[2 * (KF - 2 - k_begin)], xb = mine[2 * (KF - 1 - k_begin)];
					if constexpr (MODE == GAS_MODE_MIX_CHANNEL) {
						xa = volume(i0 + KF - 2) * xa;
						xb = volume(i0 + KF - 1) * xb;
					}
					hb2 = xa;
					hb1 = xb;
				}
			}
		} else if (role == R_REC) {
			const uint32_t t = p - 2;
			if (p >= 2 && t < n_tiles) {
				lds_f out = opaque(L.yo[t & 1] + lane);
				// half a tile at a time: 16 reads in flight, the 16-frame recurrence out of registers, 16 writes (measured:
				// per-frame reads cost a full LDS round trip each, 96 cycles per frame; whole-tile batches were slower again).
				// The chain wave is the kernel, so its step is specialised on two wave-uniform facts: every lane filters
				// (no bypass select) and no lane's feedback coefficients move this callback (no parameter change since the
				// last one: a + 0 is a, so the two adds per step can go) -- 9 -> 6 instructions per frame at best.
				auto half_tiles = [&](auto all_filter, auto still) {
					constexpr bool AF = decltype(all_filter)::value, STILL = decltype(still)::value;
#pragma unroll
					for (int k0 = 0; k0 < KF; k0 += 16) {
						v2f ur[16];
						if (AF) {
							const lds_f in = opaque(L.u1[t & 1] + lane);
#pragma unroll
							for (int j = 0; j < 16; j++) {
								ur[j] = v2f{ in[(k0 + j) * 64], 0.0f };
							}
						} else {
							const lds_f2 in = opaque(L.ux[t & 1] + lane);
#pragma unroll
							for (int j = 0; j < 16; j++) {
								ur[j] = in[(k0 + j) * 64];
							}
						}
						float yr[16];
#pragma unroll
						for (int j = 0; j < 16; j++) {
							const float yf = ur[j].x + ha1 * a1 + ha2 * a2; // ... + ha1*a1) + ha2*a2
							ha2 = ha1;
							ha1 = yf;
							yr[j] = (AF || filt) ? yf : ur[j].y; // bypass branch (:530-535, :599-605): the ramped input
							peak = fmaxf(peak, fabsf(yr[j])); // per-source peak over its mixed output (:436-443): one VALU, no LDS
							if constexpr (MODE != GAS_MODE_FX_HIGHSHELF && !STILL) {
								a1 += ia1;
								a2 += ia2;
							}
						}
#pragma unroll
						for (int j = 0; j < 16; j++) {
							out[(k0 + j) * YROW] = yr[j];
						}
					}
				};
				if (allf) {
					if (rec_still) {
						half_tiles(std::true_type(), std::true_type());
					} else {
						half_tiles(std::true_type(), std::false_type());
					}
				} else if (rec_still) {
					half_tiles(std::false_type(), std::true_type());
				} else {
					half_tiles(std::false_type(), std::false_type());
				}
			}
		} else if (role == R_POST) {
			const uint32_t t = p - 3;
			if (p >= 3 && t < n_tiles) {
				// per-source peak over its mixed output (:436-443), stream layout
				// role switch: lane = (frame, ear) sums that column over the workgroup's sources, in order
				const int f = lane >> 1;
				const lds_f col = opaque(L.yo[t & 1] + f * YROW + ear);
				float cv[SRC_PER_WG];
#pragma unroll
				for (int k = 0; k < SRC_PER_WG; k++) {
					cv[k] = col[2 * k];
				}
				if (n_buses == 0) {
					float s = 0.0f;
#pragma unroll
					for (int k = 0; k < SRC_PER_WG; k++) {
						if ((uint32_t)k < n_valid) { // wave-uniform
							s += cv[k];
						}
					}
					my_partial[(size_t)t * COLS + lane] = s;
				} else { // per bus: product with what the source sends there, then the ordered sum (k_biquad_mix.hip)
					for (uint32_t b = 0; b < n_buses; b++) {
						const float *w = L.bus_w + b * 64 + ear;
						float sb = 0.0f;
#pragma unroll
						for (int k = 0; k < SRC_PER_WG; k++) {
							sb += cv[k] * w[2 * k];
						}
						my_partial[b * bus_plane + (size_t)t * COLS + lane] = sb;
					}
				}
			}
		}
#ifdef GAS_STAMPS
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		busy += __builtin_amdgcn_s_memtime() - t_in;
#endif
		__syncthreads();
	}
#ifdef GAS_STAMPS
	if (lane == 0 && blockIdx.x == 0 && blockIdx.y == 0) {
		gas_pipe_busy[role * 2] = busy;
		gas_pipe_busy[role * 2 + 1] = __builtin_amdgcn_s_memtime() - t_loop0;
	}
#endif

	// ---- write the state back: each wave the part it owns ---------------------------------------------------------------
	if (!valid) {
		return;
	}
	if (role == R_REC) {
		// per-source peak over all its channel pairs (:419-444): max is order-independent
		if (gridDim.y == 1) { // one channel pair: this lane is the only writer (and the context skips the zeroing pass)
			g.peaks[(size_t)row * 2 + ear] = peak;
		} else {
			atomicMax(reinterpret_cast<unsigned int *>(g.peaks) + (size_t)row * 2 + ear, __float_as_uint(peak));
		}
	}
	if (role == R_COEF && filt) {
		bq[BQ_B0 * bs + stream] = b0;
		bq[BQ_B1 * bs + stream] = b1;
		bq[BQ_B2 * bs + stream] = b2;
	} else if (role == R_REC && filt) {
		bq[BQ_A1 * bs + stream] = a1;
		bq[BQ_A2 * bs + stream] = a2;
		bq[BQ_HA1 * bs + stream] = ha1;
		bq[BQ_HA2 * bs + stream] = ha2;
	} else if (role == R_FIR0) {
		if (filt) {
			bq[BQ_HB1 * bs + stream] = hb1;
			bq[BQ_HB2 * bs + stream] = hb2;
		}
		if constexpr (MODE == GAS_MODE_MIX_CHANNEL || MODE == GAS_MODE_PROCESS_FRAMES) {
			bq[BQ_PREV * bs + stream] = vf; // set_prev_mix_volume (:551, :608)
		}
	}
}

template <int MODE>
void launch_pipe(hipStream_t stream, dim3 grid, bool f_pow2, bool all_filt, const gas_group_args &g, const gas_dev_state &st, uint32_t frames, uint32_t channel_begin, float mix_rate, float *partials, uint32_t p_offset, uint32_t p_stride, const gas_bus_args &buses) {
	dim3 block(N_WAVES * 64);
	if (f_pow2) {
		if (all_filt) {
			hipLaunchKernelGGL((k_biquad_pipe<MODE, true, true>), grid, block, 0, stream, g, st, frames, channel_begin, mix_rate, partials, p_offset, p_stride, buses);
		} else {
			hipLaunchKernelGGL((k_biquad_pipe<MODE, true, false>), grid, block, 0, stream, g, st, frames, channel_begin, mix_rate, partials, p_offset, p_stride, buses);
		}
	} else {
		if (all_filt) {
			hipLaunchKernelGGL((k_biquad_pipe<MODE, false, true>), grid, block, 0, stream, g, st, frames, channel_begin, mix_rate, partials, p_offset, p_stride, buses);
		} else {
			hipLaunchKernelGGL((k_biquad_pipe<MODE, false, false>), grid, block, 0, stream, g, st, frames, channel_begin, mix_rate, partials, p_offset, p_stride, buses);
		}
	}
}

} // namespace

// all_filt: the caller knows that every source of the launch takes the filter branch (FX_HIGHSHELF always does); the
// general variant selects per lane.
hipError_t gas_launch_biquad_pipe(hipStream_t stream, int mode, const gas_group_args &g, const gas_dev_state &st, uint32_t frames, uint32_t channel_begin, uint32_t channel_count, float mix_rate, float *partials, uint32_t p_offset, uint32_t p_stride, const gas_bus_args &buses) {
	if (g.n == 0) {
		return hipSuccess;
	}
	if (frames % KF != 0) {
		return hipErrorInvalidValue;
	}
	dim3 grid(gas_biquad_partials(g.n), channel_count);
	const bool f_pow2 = (frames & (frames - 1)) == 0;
	switch (mode) {
		case GAS_MODE_MIX_CHANNEL:
			launch_pipe<GAS_MODE_MIX_CHANNEL>(stream, grid, f_pow2, false, g, st, frames, channel_begin, mix_rate, partials, p_offset, p_stride, buses);
			break;
		case GAS_MODE_PROCESS_FRAMES:
			launch_pipe<GAS_MODE_PROCESS_FRAMES>(stream, grid, f_pow2, false, g, st, frames, channel_begin, mix_rate, partials, p_offset, p_stride, buses);
			break;
		case GAS_MODE_FX_HIGHSHELF:
			launch_pipe<GAS_MODE_FX_HIGHSHELF>(stream, grid, f_pow2, true, g, st, frames, channel_begin, mix_rate, partials, p_offset, p_stride, buses);
			break;
		default:
			return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

#ifdef GAS_STAMPS
extern "C" int gas_debug_read_pipe_busy(unsigned long long *out) {
	if (hipDeviceSynchronize() != hipSuccess) {
		return -7;
	}
	return hipMemcpyFromSymbol(out, HIP_SYMBOL(gas_pipe_busy), sizeof(gas_pipe_busy)) == hipSuccess ? 0 : -7;
}
#endif
