// k_biquad_mix.hip -- pan/attenuation volume ramp + time-varying high-shelf biquad for every
// source of a callback, and the per-workgroup partial of the N-source -> stereo sum.
//
// Replaces, for all sources at once:
//   AudioSpatializerInstance3D::mix_channel     audio_spatializer_3d.cpp:554-609   (GAS_MODE_MIX_CHANNEL)
//   AudioSpatializerInstance3D::process_frames  audio_spatializer_3d.cpp:491-552   (GAS_MODE_PROCESS_FRAMES)
//   [ENGINE] AudioEffectFilterInstance::process (high-shelf, 1 stage) as the single effect of an
//   AudioSpatializerEffect chain, audio_spatializer_effect.cpp:52-76             (GAS_MODE_FX_HIGHSHELF)
//   the empty-chain copy, audio_spatializer_effect.cpp:41-46                       (GAS_MODE_COPY)
//   the accumulate + per-source peak of _mix_from_playback_list, audio_spatializer.cpp:432-444,449-461
//
// Geometry (CDNA4, wave64): one 64-lane workgroup = 32 sources x 2 ears; every lane owns one
// serial recurrence (the biquad is a dependent chain over the F frames).
// The 4 KiB-strided source rows are staged through LDS in [32 sources x KF frames] tiles (KF = 32) with
// coalesced 16-byte loads (16 lanes cover one 256-byte run of one row), double-buffered against the
// recurrence.  After each tile the wave switches roles: lane (half, frame, ear) sums the tile's 32
// outputs over sources in fixed order and lanes 0..31 store 128 contiguous bytes of this
// workgroup's partial mix.  No atomics in the sum: k_mix_reduce adds the partials in fixed order.
//
// Bound: HBM once N is large (algorithmic bytes/source = F*8 source + 80 state r/w + 128 params + 8 peak);
// at N = 256 it is latency-bound on the F-step recurrence (8 waves on a 256-CU part) -- DESIGN.md.
#include <cstdlib>

#include "gas_biquad.h"

// No FMA contraction in this file.  The recurrence is f32 with poles that approach the unit circle at low cutoffs /
// small shelf gains (|p| up to 0.999x); there a fused multiply-add in place of the reference's separately rounded
// multiply and add moves the output by up to 1e-3 relative (measured: 9e-4 at 50 Hz), far outside the 1e-5 parity bar,
// while with identical rounding the kernel tracks the CPU restatement to ~1e-7 at every setting.  Costs ~4 more VALU
// instructions per step.
#pragma clang fp contract(off)

namespace {

#ifndef GAS_BQ_NT
#define GAS_BQ_NT 1 // the staging loads of the (once-touched) source rows are non-temporal: 65 536 sources 79.8 -> 73.2 us per launch (round 3, profiles/r03_notes.md); no effect at 256 sources
#endif
typedef float bq_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 bq_row_load(const float *p) {
#if GAS_BQ_NT
	const bq_v4f v = __builtin_nontemporal_load(reinterpret_cast<const bq_v4f *>(p));
	return make_float4(v.x, v.y, v.z, v.w);
#else
	return *reinterpret_cast<const float4 *>(p);
#endif
}

constexpr int SRC_PER_WG = 32;
#ifndef GAS_BIQUAD_KF
#define GAS_BIQUAD_KF 32
#endif
constexpr int KF = GAS_BIQUAD_KF; // frames per staged tile (16 or 32): the per-tile overhead (staging, two barriers, role-switch sum) is paid
                                  // once per KF recurrence steps; measured: 32 is 9 % faster at N = 256 (cfg2), 16 is 4 % faster at N = 65 536
constexpr int COLS = KF * 2; // floats per source per tile
constexpr int ROW = COLS + 2; // LDS row stride in floats: + 2 pad -> (ROW*sl + ear) % 32 = (2 sl + ear) % 32, distinct over a half-wave
constexpr int PARTS = COLS / 4; // 16-byte pieces per source row per tile
constexpr int LOADS = SRC_PER_WG * PARTS / 64; // staging loads per lane per tile
constexpr int GROUPS = 64 / COLS; // lane groups of the role-switch sum (2 at KF = 16, 1 at KF = 32)
static_assert(KF == 16 || KF == 32, "tile size");

// Per-lane recurrence state.
struct LaneState {
	Coeffs co, inc;
	float ha1, ha2, hb1, hb2;
	float vs, vf;
	float peak;
	bool filt, valid;
};

// The tile loop, specialised on the two wave-uniform conditions so neither the IEEE division of the
// non-power-of-two lerp nor the bypass-branch selects are evaluated per step.
template <int MODE, bool F_POW2, bool ALL_FILT>
__device__ __forceinline__ void run_tiles(LaneState &L, float (&tile)[2][SRC_PER_WG * ROW], const float *const (&ld_base)[LOADS], uint32_t F, float *__restrict__ my_partial, int lane, int sl, int ear, const bool rows_mode, float *const (&st_base)[LOADS], uint32_t n_buses = 1, const float *bus_w = nullptr, size_t bus_plane = 0) {
	Coeffs co = L.co, inc = L.inc;
	float ha1 = L.ha1, ha2 = L.ha2, hb1 = L.hb1, hb2 = L.hb2;
	const float vs = L.vs, vf = L.vf;
	const bool filt = L.filt, valid = L.valid;
	float peak = 0.0f;
	constexpr bool f_pow2 = F_POW2;
	constexpr bool all_filt = ALL_FILT;
	const float Ff = (float)(int)F;
	const float invF = 1.0f / Ff;
	const uint32_t n_tiles = F / KF;
	float4 pre[LOADS];
#pragma unroll
	for (int q = 0; q < LOADS; q++) {
		pre[q] = bq_row_load(ld_base[q]);
	}

	for (uint32_t tl = 0; tl < n_tiles; tl++) {
		float *tb = tile[tl & 1];
		// registers -> LDS (two 8-byte stores; rows are 136 B so 16-byte stores would misalign)
#pragma unroll
		for (int q = 0; q < LOADS; q++) {
			int idx = q * 64 + lane;
			float *d = tb + (idx / PARTS) * ROW + (idx % PARTS) * 4;
			*reinterpret_cast<float2 *>(d) = make_float2(pre[q].x, pre[q].y);
			*reinterpret_cast<float2 *>(d + 2) = make_float2(pre[q].z, pre[q].w);
		}
		if (tl + 1 < n_tiles) {
#pragma unroll
			for (int q = 0; q < LOADS; q++) {
				pre[q] = bq_row_load(ld_base[q] + (size_t)(tl + 1) * KF * 2);
			}
		}
		__syncthreads();

		float *mine = tb + sl * ROW + ear;
		const int i0 = (int)(tl * KF);
		// the tile's 16 samples of this stream into registers first: no LDS latency inside the recurrence
		float xr[KF];
#pragma unroll
		for (int k = 0; k < KF; k++) {
			xr[k] = mine[2 * k];
		}
#pragma unroll
		for (int k = 0; k < KF; k++) {
			float x = xr[k];
			float y;
			if constexpr (MODE == GAS_MODE_MIX_CHANNEL) {
				const float fi = (float)(i0 + k);
				const float t = f_pow2 ? fi * invF : fi / Ff; // (float)frame_idx / p_frame_count (:591)
				const float vol = vf * t + (1 - t) * vs; // :592
				x = vol * x; // :593
			}
			if constexpr (MODE == GAS_MODE_COPY) {
				y = x;
			} else if constexpr (MODE == GAS_MODE_FX_AMPLIFY) {
				// [ENGINE] AudioEffectAmplifyInstance::process: dst = src * vol; vol += vol_inc (co.b0 = vol, inc.b0 = vol_inc)
				y = x * co.b0;
				co.b0 += inc.b0;
			} else {
				// [ENGINE] process_one(_interp)
				const float yf = x * co.b0 + hb1 * co.b1 + hb2 * co.b2 + ha1 * co.a1 + ha2 * co.a2;
				if (all_filt) {
					y = yf;
					ha2 = ha1;
					hb2 = hb1;
					hb1 = x;
					ha1 = yf;
				} else {
					// bypass branch (:530-535, :599-605) leaves the processor untouched
					y = filt ? yf : x;
					ha2 = filt ? ha1 : ha2;
					hb2 = filt ? hb1 : hb2;
					hb1 = filt ? x : hb1;
					ha1 = filt ? yf : ha1;
				}
				if constexpr (MODE != GAS_MODE_FX_HIGHSHELF && MODE != GAS_MODE_FX_FILTER) {
					co.b0 += inc.b0;
					co.b1 += inc.b1;
					co.b2 += inc.b2;
					co.a1 += inc.a1;
					co.a2 += inc.a2;
				}
			}
			y = valid ? y : 0.0f;
			const float a = fabsf(y);
			peak = a > peak ? a : peak; // :436-443
			xr[k] = y;
		}
#pragma unroll
		for (int k = 0; k < KF; k++) {
			mine[2 * k] = xr[k];
		}
		__syncthreads();

		if (rows_mode) { // wave-uniform
			// rows-out (a stage of a general effect chain): hand the processed tile back as per-source rows, with the
			// staging loads' own coalesced pattern; no sum here -- k_rows_accumulate mixes the chain's last stage
#pragma unroll
			for (int q = 0; q < LOADS; q++) {
				if (st_base[q]) {
					const int idx = q * 64 + lane;
					const float *t4 = tb + (idx / PARTS) * ROW + (idx % PARTS) * 4;
					*reinterpret_cast<float4 *>(st_base[q] + (size_t)tl * KF * 2) = make_float4(t4[0], t4[1], t4[2], t4[3]);
				}
			}
		} else {
		// role switch: lane (h, j) sums column j = frame*2+ear over its group's sources, in order; the groups
		// (two at KF = 16) are then added in order too
		{
			const int h = lane / COLS, j = lane % COLS;
			constexpr int PER = SRC_PER_WG / GROUPS;
			const float *col = tb + (PER * h) * ROW + j;
			float s = 0.0f;
#pragma unroll
			for (int k = 0; k < PER; k++) {
				s += col[k * ROW];
			}
			if constexpr (GROUPS == 2) {
				const float s_hi = __shfl_down(s, 32);
				s += s_hi;
			}
			if (n_buses == 0) {
				if (lane < COLS) {
					my_partial[(size_t)tl * COLS + j] = s;
				}
			} else {
				// several buses: the same column, weighted per source by what that source sends to the
				// bus (dry 1, send = bus / mix volume; audio_spatializer.cpp:295-313), product then sum like AudioServer's
				// per-bus multiply-accumulate
				for (uint32_t b = 0; b < n_buses; b++) {
					const float *w = bus_w + b * 64 + (PER * h) * 2 + (j & 1);
					float sb = 0.0f;
#pragma unroll
					for (int k = 0; k < PER; k++) {
						sb += col[k * ROW] * w[2 * k];
					}
					if constexpr (GROUPS == 2) {
						sb += __shfl_down(sb, 32);
					}
					if (lane < COLS) {
						my_partial[b * bus_plane + (size_t)tl * COLS + j] = sb;
					}
				}
			}
		}
		}
		// the next iteration writes the other buffer; this one is rewritten two tiles later,
		// after the __syncthreads() that follows that write.
	}

	L.co = co;
	L.ha1 = ha1;
	L.ha2 = ha2;
	L.hb1 = hb1;
	L.hb2 = hb2;
	L.peak = peak;
}

template <int MODE>
__global__ __launch_bounds__(64) void k_biquad_mix(gas_group_args g, gas_dev_state st, uint32_t F, uint32_t c0, float mix_rate, float *__restrict__ partials, uint32_t p_offset, uint32_t p_stride, float *__restrict__ rows_out, gas_bus_args buses, int fx_kind) {
	__shared__ float tile[2][SRC_PER_WG * ROW];
	__shared__ float bus_w[GAS_MAX_BUSES * 64]; // [bus][source * 2 + ear]: what each source sends to each bus for this pair

	const int lane = threadIdx.x;
	const int ear = lane & 1;
	const int sl = lane >> 1;
	const uint32_t c = c0 + blockIdx.y;
	const uint32_t e = blockIdx.x * SRC_PER_WG + sl;
	const bool valid = e < g.n;
	const uint32_t ec = valid ? e : g.n - 1;
	const uint32_t slot = g.slots[ec];
	const uint32_t row = g.rows ? g.rows[ec] : ec;
	const gas_params *P = st.params + slot;

	// Row base of each of this lane's staging loads per tile: load q covers tile element
	// idx = q*64 + lane -> source idx / PARTS, 16-byte part idx % PARTS.
	const float *ld_base[LOADS];
#pragma unroll
	for (int q = 0; q < LOADS; q++) {
		int idx = q * 64 + lane;
		uint32_t le = blockIdx.x * SRC_PER_WG + (idx / PARTS);
		le = le < g.n ? le : g.n - 1;
		uint32_t lrow = g.rows ? g.rows[le] : le;
		ld_base[q] = reinterpret_cast<const float *>(g.src) + (size_t)lrow * F * 2 + (idx % PARTS) * 4;
	}
	// rows-out: dense rows by group entry; entries past the end are not written
	float *st_base[LOADS];
#pragma unroll
	for (int q = 0; q < LOADS; q++) {
		const int idx = q * 64 + lane;
		const uint32_t le = blockIdx.x * SRC_PER_WG + (idx / PARTS);
		st_base[q] = (rows_out && le < g.n) ? rows_out + (size_t)le * F * 2 + (idx % PARTS) * 4 : nullptr;
	}


	// ---- per-lane DSP state (SpatializerPlaybackData3D, audio_spatializer_3d.h:85-99) ----
	const size_t stream = ((size_t)slot * 4 + c) * 2 + ear;
	float *bq = st.bq;
	const size_t bs = st.bq_stride;
	Coeffs co = { 0, 0, 0, 0, 0 }, inc = { 0, 0, 0, 0, 0 };
	float ha1 = 0, ha2 = 0, hb1 = 0, hb2 = 0;
	float vs = 0, vf = 0;
	bool filt = false;

	if constexpr (MODE != GAS_MODE_COPY && MODE != GAS_MODE_FX_AMPLIFY) {
		co.b0 = bq[BQ_B0 * bs + stream];
		co.b1 = bq[BQ_B1 * bs + stream];
		co.b2 = bq[BQ_B2 * bs + stream];
		co.a1 = bq[BQ_A1 * bs + stream];
		co.a2 = bq[BQ_A2 * bs + stream];
		ha1 = bq[BQ_HA1 * bs + stream];
		ha2 = bq[BQ_HA2 * bs + stream];
		hb1 = bq[BQ_HB1 * bs + stream];
		hb2 = bq[BQ_HB2 * bs + stream];
	}
	if constexpr (MODE == GAS_MODE_MIX_CHANNEL || MODE == GAS_MODE_PROCESS_FRAMES) {
		vs = bq[BQ_PREV * bs + stream]; // get_prev_mix_volume(c), (0,0) when never set (:880-885)
		const float vs_other = bq[BQ_PREV * bs + (stream ^ 1)];
		const float gain = P->linear_attenuation;
		filt = (double)gain >= 0.001; // :503 / :568
		if (filt) {
			const Coeffs target = highshelf_coeffs(mix_rate, P->attenuation_filter_cutoff_hz, gain);
			if (vs == 0 && vs_other == 0) { // is_just_started -> clear_history (:518-521, :583-587)
				ha1 = ha2 = hb1 = hb2 = 0;
			}
			// [ENGINE] update_coeffs(F): ramp from the current coefficients to the target.
			const int Fi = (int)F;
			inc.a1 = (target.a1 - co.a1) / Fi;
			inc.a2 = (target.a2 - co.a2) / Fi;
			inc.b0 = (target.b0 - co.b0) / Fi;
			inc.b1 = (target.b1 - co.b1) / Fi;
			inc.b2 = (target.b2 - co.b2) / Fi;
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
	}
	if constexpr (MODE == GAS_MODE_FX_HIGHSHELF) {
		co = highshelf_coeffs(mix_rate, P->fx_shelf_cutoff_hz, P->fx_shelf_gain); // coefficients snap every call
		filt = true;
	}
	if constexpr (MODE == GAS_MODE_FX_FILTER) { // chain position c: its settings, its state stream
		const gas_fx_settings *S = st.fxs + slot;
		co = filter_coeffs(fx_kind, mix_rate, S->filter_cutoff_hz[c], S->filter_resonance[c], S->filter_gain[c]);
		filt = true;
	}
	float amp_db = 0.0f;
	if constexpr (MODE == GAS_MODE_FX_AMPLIFY) {
		// mix_volume_db lives in the PREV field of this position's state stream, B0 says whether a block has run: the
		// instance starts at the resource's volume ([ENGINE] instantiate: mix_volume_db = volume_db), i.e. no ramp
		amp_db = st.fxs[slot].amplify_volume_db[c];
		const bool started = bq[BQ_B0 * bs + stream] != 0.0f;
		const float from_db = started ? bq[BQ_PREV * bs + stream] : amp_db;
		co.b0 = fx_db_to_linear(from_db); // vol
		inc.b0 = (fx_db_to_linear(amp_db) - co.b0) / (float)(int)F; // vol_inc
		filt = true;
	}

	const bool all_filt = __all(filt || !valid);
	const bool f_pow2 = (F & (F - 1)) == 0;
	float *my_partial = partials + ((size_t)blockIdx.y * p_stride + p_offset + blockIdx.x) * (size_t)F * 2;
	const size_t bus_plane = (size_t)gridDim.y * p_stride * (size_t)F * 2; // floats between the partial planes of two buses
	const uint32_t n_buses = buses.routes ? buses.n_buses : 0; // 0: the single mix of gas_process_block
	if (n_buses > 0) {
		gas_bus_route r = gas_bus_route_default();
		if (valid) {
			r = buses.routes[slot];
		}
		for (uint32_t b = 0; b < n_buses; b++) {
			float w = 0.0f;
			if (valid) {
				w = gas_bus_weight(r, b, (int)c, ear);
			}
			bus_w[b * 64 + lane] = w;
		}
		__syncthreads();
	}
	LaneState L{ co, inc, ha1, ha2, hb1, hb2, vs, vf, 0.0f, filt, valid };
	if (f_pow2) {
		if (all_filt) {
			run_tiles<MODE, true, true>(L, tile, ld_base, F, my_partial, lane, sl, ear, rows_out != nullptr, st_base, n_buses, bus_w, bus_plane);
		} else {
			run_tiles<MODE, true, false>(L, tile, ld_base, F, my_partial, lane, sl, ear, rows_out != nullptr, st_base, n_buses, bus_w, bus_plane);
		}
	} else {
		if (all_filt) {
			run_tiles<MODE, false, true>(L, tile, ld_base, F, my_partial, lane, sl, ear, rows_out != nullptr, st_base, n_buses, bus_w, bus_plane);
		} else {
			run_tiles<MODE, false, false>(L, tile, ld_base, F, my_partial, lane, sl, ear, rows_out != nullptr, st_base, n_buses, bus_w, bus_plane);
		}
	}
	co = L.co;
	ha1 = L.ha1;
	ha2 = L.ha2;
	hb1 = L.hb1;
	hb2 = L.hb2;
	const float peak = L.peak;

	if (valid) {
		if constexpr (MODE == GAS_MODE_FX_AMPLIFY) {
			bq[BQ_B0 * bs + stream] = 1.0f;
			bq[BQ_PREV * bs + stream] = amp_db; // mix_volume_db = volume_db
		}
		if constexpr (MODE != GAS_MODE_COPY && MODE != GAS_MODE_FX_AMPLIFY) {
			if (filt) {
				bq[BQ_B0 * bs + stream] = co.b0;
				bq[BQ_B1 * bs + stream] = co.b1;
				bq[BQ_B2 * bs + stream] = co.b2;
				bq[BQ_A1 * bs + stream] = co.a1;
				bq[BQ_A2 * bs + stream] = co.a2;
				bq[BQ_HA1 * bs + stream] = ha1;
				bq[BQ_HA2 * bs + stream] = ha2;
				bq[BQ_HB1 * bs + stream] = hb1;
				bq[BQ_HB2 * bs + stream] = hb2;
			}
		}
		if constexpr (MODE == GAS_MODE_MIX_CHANNEL || MODE == GAS_MODE_PROCESS_FRAMES) {
			bq[BQ_PREV * bs + stream] = vf; // set_prev_mix_volume (:551, :608)
		}
		// per-source peak over all its channel pairs (:419-444): max is order-independent
		if (!rows_out) {
			if (gridDim.y == 1) { // one channel pair: this lane is the only writer (and the context skips the zeroing pass)
				g.peaks[(size_t)row * 2 + ear] = peak;
			} else {
				atomicMax(reinterpret_cast<unsigned int *>(g.peaks) + (size_t)row * 2 + ear, __float_as_uint(peak));
			}
		}
	}
}

} // namespace

uint32_t gas_biquad_partials(uint32_t n) {
	return (n + SRC_PER_WG - 1) / SRC_PER_WG;
}

// Few workgroups (no more than CUs): the launch is bound by one lane's dependent chain, not by memory -> the
// eight-wave software pipeline of k_biquad_pipe.hip (bitwise the same results).  GAS_BIQUAD_PIPE=0 keeps the
// single-wave kernel (development A/B; read per launch so that a test can switch inside one process).
bool gas_biquad_uses_pipe(int mode, uint32_t n, uint32_t channel_count, uint32_t frames, bool rows_out) {
	const char *pipe_env = std::getenv("GAS_BIQUAD_PIPE");
	const bool pipe_on = !(pipe_env && pipe_env[0] == '0');
	return pipe_on && !rows_out && mode != GAS_MODE_COPY && mode != GAS_MODE_FX_FILTER && mode != GAS_MODE_FX_AMPLIFY && gas_biquad_partials(n) * channel_count <= 256 && frames % 32 == 0;
}

hipError_t gas_launch_biquad_mix(hipStream_t stream, int mode, const gas_group_args &g, const gas_dev_state &st, uint32_t frames, uint32_t channel_begin, uint32_t channel_count, float mix_rate, float *partials, uint32_t p_offset, uint32_t p_stride, float *rows_out, const gas_bus_args &buses, int fx_kind) {
	if (g.n == 0) {
		return hipSuccess;
	}
	if (rows_out && channel_count == 1 && g.slots && gas_shelf_scan_applies(mode, g.n, frames)) {
		return gas_launch_shelf_scan(stream, mode, g, st, frames, channel_begin, mix_rate, rows_out, fx_kind);
	}
	dim3 grid(gas_biquad_partials(g.n), channel_count);
	dim3 block(64);
	if (gas_biquad_uses_pipe(mode, g.n, channel_count, frames, rows_out != nullptr)) {
		return gas_launch_biquad_pipe(stream, mode, g, st, frames, channel_begin, channel_count, mix_rate, partials, p_offset, p_stride, buses);
	}
	switch (mode) {
		case GAS_MODE_MIX_CHANNEL:
			hipLaunchKernelGGL(k_biquad_mix<GAS_MODE_MIX_CHANNEL>, grid, block, 0, stream, g, st, frames, channel_begin, mix_rate, partials, p_offset, p_stride, rows_out, buses, fx_kind);
			break;
		case GAS_MODE_PROCESS_FRAMES:
			hipLaunchKernelGGL(k_biquad_mix<GAS_MODE_PROCESS_FRAMES>, grid, block, 0, stream, g, st, frames, channel_begin, mix_rate, partials, p_offset, p_stride, rows_out, buses, fx_kind);
			break;
		case GAS_MODE_FX_HIGHSHELF:
			hipLaunchKernelGGL(k_biquad_mix<GAS_MODE_FX_HIGHSHELF>, grid, block, 0, stream, g, st, frames, channel_begin, mix_rate, partials, p_offset, p_stride, rows_out, buses, fx_kind);
			break;
		case GAS_MODE_FX_FILTER:
			hipLaunchKernelGGL(k_biquad_mix<GAS_MODE_FX_FILTER>, grid, block, 0, stream, g, st, frames, channel_begin, mix_rate, partials, p_offset, p_stride, rows_out, buses, fx_kind);
			break;
		case GAS_MODE_FX_AMPLIFY:
			hipLaunchKernelGGL(k_biquad_mix<GAS_MODE_FX_AMPLIFY>, grid, block, 0, stream, g, st, frames, channel_begin, mix_rate, partials, p_offset, p_stride, rows_out, buses, fx_kind);
			break;
		case GAS_MODE_COPY:
			hipLaunchKernelGGL(k_biquad_mix<GAS_MODE_COPY>, grid, block, 0, stream, g, st, frames, channel_begin, mix_rate, partials, p_offset, p_stride, rows_out, buses, fx_kind);
			break;
		default:
			return hipErrorInvalidValue;
	}
	return hipGetLastError();
}
