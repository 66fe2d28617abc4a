// k_hrtf_uni.hip -- every plain [HRTF] playback of a callback in ONE uniform launch: per-source 256-tap HRIR
// convolution by overlap-save FFT, summed over sources in the frequency domain, with exact per-source peaks for the
// playbacks that need one.
//
// NEW arithmetic (the reference has no HRTF/FFT/convolution); it sits where
// AudioSpatializerInstanceEffect::process_frames runs its effect chain (audio_spatializer_effect.cpp:52-76) and
// feeds the accumulate + per-source peak of _mix_from_playback_list (audio_spatializer.cpp:449-461).  Semantics are
// fixed by oracle/gas_oracle.c (fx_hrtf):
//   x   = ((src.l + src.r) * 0.5) * (g1*t + (1-t)*g0),  t = i/F          (gain ramp as audio_spatializer_3d.cpp:591-592)
//   out = (hrir[dir][L] * x, hrir[dir][R] * x)  over  hist ++ x           (256 taps, direction switches per block)
//
// How it differs from k_hrtf_ols (which keeps the early-reflection, cross-fade, direction-run and rows-out forms):
//   * one kind of workgroup.  k_hrtf_ols splits a callback into frequency-domain workgroups and exact-peak
//     workgroups with their own slot lists; here every source is accumulated in the frequency domain (one forward
//     FFT-512, sum_s Z_s H[d_s], one inverse pair per workgroup) and a source whose peak the host will read (stream
//     ended, audio_spatializer.cpp:464-469; or every source without GAS_FLAG_PEAKS_DRAINING_ONLY) additionally gets
//     its own inverse pair, for the peak only.  Draining playbacks therefore neither re-sort the list nor skew the
//     split of sources over waves (measured with stamps: the 5-vs-4 sources per wave of the split lists set the
//     kernel's end 2 us after its median workgroup).
//   * a short prologue.  With the callback's slots a contiguous range in row order nothing has to be fetched before
//     the first source's frames and history are requested: one round trip to the first transform instead of three
//     (kernel arguments -> slot list -> parameter row -> frames; measured 3.6 us to the first transform of 16 us).
//     Per-source state stores, the write-through of device-published parameter rows and the loads of the carried
//     partial-mix sum (GAS_FLAG_PIPELINED_MIX) are issued behind the source loop, off that path.
//
// Mapping (CDNA4, wave64), algorithm, LDS layout: gas_hrtf_wave.h / DESIGN.md 3.1.  Bound: HBM.  Algorithmic bytes
// per source = F*8 (frames) + 2*hist_len*4 (history r+w) + 24 (gain, direction, previous gain r/w, peak).
#include <cstdlib>

// Source rows are loaded non-temporal (round 3): once-touched frames that do not displace the HRIR rows -- the only
// data of the launch that is re-read -- from the XCD's L2.  Measured, synchronous callback (profiles/r03_notes.md
// section 3): 8192 sources 16.0 -> 14.5 us, 65 536 sources 89.6 -> 80.8 us, 1 M sources 1405 -> 1350 us.  (Round 1
// found no effect on k_hrtf_ols, whose table was twice the size and whose launch was latency-bound elsewhere.)
#define GAS_USE_NT 1
#include "gas_hrtf_wave.h"
#include "gas_biquad.h" // the engine's coefficient preparation (FLT): switches FMA contraction off for what follows ...
#pragma clang fp contract(fast) // ... and this is the compiler's default for device code again (the transforms' bits do not change: checked against the ISA of the build before the include)

#ifndef GAS_UNI12_DEFAULT_MIN
#define GAS_UNI12_DEFAULT_MIN 0 // sources from which the twelve-wave form is the default (0 = only on request)
#endif

#ifndef GAS_UNI12_DIRECT
#define GAS_UNI12_DIRECT 0 // EXPERIMENT: the twelve-wave form takes the HRIR row straight from the L2 into registers at the top of the trip (no LDS table slots, no LDS-DMA): the third wave per SIMD without the extra 8 KiB of LDS traffic per source
#endif

namespace {

#ifdef GAS_STAMPS
#define GAS_UNI_STAMP(i)                                                                                       \
	do {                                                                                                        \
		if (lane == 0 && (blockIdx.x * UW + wave) < 8192) {                                                     \
			gas_stamps[(blockIdx.x * UW + wave) * GAS_STAMP_SLOTS + (i)] = __builtin_amdgcn_s_memrealtime();    \
		}                                                                                                       \
	} while (0)
#else
#define GAS_UNI_STAMP(i) do { } while (0)
#endif

// UW = waves per workgroup (one workgroup per CU): 8 = two waves per SIMD, the general form (stream sampling, two buses);
// 12 = three waves per SIMD (<= 168 VGPRs), float rows on one bus.  What the third wave costs in registers is paid by
// the HRIR row: the twelve-wave form never holds it in VGPRs.  Each wave owns two 4 KiB LDS slots; the stored half
// of the next source's row (256 float4) is copied global -> LDS by four global_load_lds_dwordx4 at the top of a trip
// and read back one trip later, in natural order for bins < 256 and in mirrored order for bins >= 256 (the table's
// Hermitian half, gas_hrtf_wave.h) -- half the L2 -> CU bytes of the register form, no 32-VGPR landing set, and no
// skewed spectrum copy (the products follow the transform they belong to).  Same operations per source, but the
// sources are split over 12 waves instead of 8, so the sums differ from the 8-wave form's in the last bits.
template <int UW>
struct UniCfg {
	static constexpr bool LEAN = UW > 8;
	static constexpr int SLICES = LEAN ? 1 : 2; // exchange slices per wave in the source loop
	static constexpr int TSLOT_F2 = LEAN ? 2 * 512 : 0; // float2 units: two table slots of 256 float4 per wave
};

// LDS (float2 units).  The source loop's exchange slices (+ table slots) and the epilogue's fd[wave][ear][512] + two
// slices + output alias each other.
template <int SQ, int UW>
struct UniLds {
	static constexpr int F = 2 * SQ * 64;
	static constexpr int FD_F2 = UW * 2 * 512; // fd[wave][ear][512]
	static constexpr int EPI_F2 = FD_F2 + 2 * LDS_F2_HALF + F; // + two exchange slices + the [F][2] output
	static constexpr int LOOP_F2 = UW * (UniCfg<UW>::SLICES * LDS_F2_HALF + UniCfg<UW>::TSLOT_F2);
	static constexpr int TOTAL_F2 = EPI_F2 > LOOP_F2 ? EPI_F2 : LOOP_F2;
};

// BUS2 (gas_process_block_buses with [HRTF] sources and one or two buses): every source's output goes to
// bus b with weight w_b = (dry_bus == b) + (send_bus == b ? send : 0) per ear.  Bus 0 keeps the register sums (weighted),
// bus 1's sums live in LDS (64 KB the plain form does not use) and are only touched by sources that reach it; the
// epilogue runs once per bus.  Partial rows of bus b: [b * bus_rows + p_offset + workgroup].
// More than two buses (a playback reaches up to six, audio_spatializer.cpp:283-287): one launch per PAIR of buses --
// bus_base names the pair (buses bus_base, bus_base + 1) and only the last launch commits the per-source state (history
// row, previous gain, peak), so every pass transforms the same windows.
// ER (round 3): the chain [EARLY_REFLECTIONS, HRTF] in this kernel's shape -- the eight-tap gather of oracle
// fx_early_reflections (y[i] = x[i] + sum_k g_k x[i - d_k], per-source ring of er_R frames) in front of the window, the
// rest unchanged.  k_hrtf_ols<ER> (split frequency-domain / exact-peak workgroups, 22.5 us for cfg5) stays for the
// cross-fade / direction-run forms; entries from peak_from on get their exact peak (the context's exact-peak group of
// the chain rides behind its frequency-domain group in one list).  Measured and dropped: pulling the next source's ring
// lines towards the L2 a trip ahead (one dword per 128-byte line) -- 24.6 us instead of 20.4: the tap reads are 64 of
// the launch's 99 MB and already move at the rate a copy gets, the touches only add requests.
// FLT (round 3): the chain [one-biquad filter, HRTF] -- the reference example's high shelf in front of the HRTF
// (examples/godot-gd-spatializer/gd_spatializer.gd:11-20), or any of the engine's other one-stage filters -- in this
// kernel's shape instead of a filter launch that writes rows and an HRTF launch that reads them back (8192 sources:
// 44 -> 31 us with every peak, 38 -> 23 with the draining ones; profiles/r03_notes.md).  The filter is k_shelf_scan's,
// operation for operation (the same bits as the two-launch form): both ears, a lane owns F/64 consecutive frames (a
// transpose through the wave's LDS slice on the way in and out), local response from rest, homogeneous solutions,
// six-step affine scan over the lanes; a source whose poles lie outside r^2 <= 0.9 is walked serially by two lanes, one
// per ear (the engine's form, bitwise k_biquad_mix's).  Filtering the MEAN of the ears instead (the HRTF's input; the
// filter is linear) would halve the work, and was measured and dropped for its numerics: next to the unit circle the
// mean's one extra rounding, or a per-ear walk from a mean history, is amplified ~1 / (1 - r) -- 1e-4 on a source's peak
// for a 90 Hz high-pass.  Coefficients and history: prepared per LANE for the lane's own source in the prologue (one
// f64 sin / cos per 64 sources), parked in LDS, read back wave-uniformly per trip.
template <int SQ, bool SRC_PCM, bool BUS2, int UW, bool ER = false, bool FLT = false>
__global__ __launch_bounds__(UW * 64, UW > 8 ? 3 : GAS_HRTF_WAVES_PER_SIMD) void k_hrtf_uni(gas_group_args g, const uint32_t *__restrict__ peak_bits, uint32_t peak_all, gas_dev_state st, gas_hrtf_table tab, const float2 *__restrict__ tw, float *__restrict__ partials, uint32_t p_offset, gas_cursor *__restrict__ cursors, const float *__restrict__ fade_env, const gas_params *__restrict__ fresh, gas_deferred_reduce job, const gas_bus_route *__restrict__ routes, uint32_t bus_rows, uint32_t bus_base, uint32_t commit, uint32_t nt_hist, uint32_t er_R, uint32_t peak_from, uint32_t peak_bit_base, uint32_t flt_kind, uint32_t flt_pos, float mix_rate) {
	static_assert(!FLT || (!SRC_PCM && !BUS2 && UW == 8 && !ER), "filter in front: float rows, one bus, eight waves");
	static_assert(!ER || (!SRC_PCM && !BUS2 && UW == 8), "early reflections: float rows, one bus, eight waves");
	constexpr bool LEAN = UniCfg<UW>::LEAN;
	constexpr int UNI_SLICES = UniCfg<UW>::SLICES;
	static_assert(!LEAN || (!SRC_PCM && !BUS2), "the twelve-wave form exists for float rows on one bus");
	constexpr int FQ = 2 * SQ; // F / 64
	constexpr int HQ = 8 - SQ; // hist_len / 64
	constexpr int NQ = 8 + SQ; // (hist_len + F) / 64
	constexpr uint32_t F = FQ * 64;
	constexpr uint32_t HL = HQ * 64;
	constexpr int FD_F2 = UniLds<SQ, UW>::FD_F2;
	__shared__ float2 lds_all[UniLds<SQ, UW>::TOTAL_F2];
	__shared__ float2 tw_lds[1024];
	__shared__ float2 bus1_all[BUS2 ? UW * 2 * 512 : 1]; // bus 1: [wave][ear][512], the layout of the epilogue's fd
	constexpr int FLT_W = 13; // b0 b1 b2 a1 a2, then the processor's ha1 ha2 hb1 hb2 of the left ear, of the right ear
	__shared__ float flt_all[FLT ? UW * 64 * FLT_W : 1]; // FLT: [wave][source of the wave][FLT_W]
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	float2 *lds = lds_all + wave * (UNI_SLICES * LDS_F2_HALF);
	float4 *tslot = reinterpret_cast<float4 *>(lds_all + UW * UNI_SLICES * LDS_F2_HALF) + wave * 512; // LEAN: this wave's two table slots
	float *my_partial = partials + ((size_t)p_offset + blockIdx.x) * (size_t)(F * 2);
	GAS_UNI_STAMP(0);
#ifdef GAS_STAMPS
	if (lane == 0 && (blockIdx.x * UW + wave) < 8192) { // shader-clock stamp next to the 100 MHz one: cycles per tick = clock
		gas_stamps[(blockIdx.x * UW + wave) * GAS_STAMP_SLOTS + 6] = __builtin_amdgcn_s_memtime();
	}
#endif

	// ---- prologue: nothing in front of the first source's frames that they do not depend on -------------------
	constexpr int TW_PER = (512 + UW * 64 - 1) / (UW * 64); // 16-byte pieces of the 8 KiB twiddle table per thread
	float4 tw_in[TW_PER];
#pragma unroll
	for (int r = 0; r < TW_PER; r++) {
		tw_in[r] = reinterpret_cast<const float4 *>(tw)[(threadIdx.x + r * UW * 64) & 511];
	}
	uint32_t first, last;
	wave_range(g.n, blockIdx.x * UW + wave, gridDim.x * UW, first, last);
	const bool have = first + lane < last; // <= 64 sources per wave (gas_hrtf_plan): one metadata lane per source
	LaneMeta lm{};
	uint32_t my_entry = first + lane; // list entry this lane's source is (g.order: XCD-affine processing order, k_xcd_order)
	if (have) {
		if (g.order) {
			my_entry = g.order[first + lane];
		}
		const uint32_t e = my_entry;
		lm.slot = g.slots ? g.slots[e] : g.slot_base + e; // a contiguous slot range in row order needs no list at all
		lm.row = g.rows ? g.rows[e] : e;
		lm.prow = g.peak_rows ? g.peak_rows[e] : lm.row;
	}
	gas_audio_frame raw[FQ]; // frames lane + 64 q of the source in flight
	float rawh[HQ]; // its history samples (lane-major rows: one 16-byte access per lane at F = 512)
	if (first < last) { // wave-uniform
		SrcMeta m0{};
		m0.slot = (uint32_t)__builtin_amdgcn_readlane((int)lm.slot, 0);
		m0.row = (uint32_t)__builtin_amdgcn_readlane((int)lm.row, 0);
		load_history<HQ>(st.hrtf_hist + (size_t)m0.slot * HL, lane, rawh, nt_hist != 0);
		if constexpr (!SRC_PCM) {
			load_window<false, FQ>(g, m0, lane, fade_env, raw);
		}
	}
	uint32_t my_flag = 0; // this lane's source needs its exact peak
	float my_w0l = 1.0f, my_w0r = 1.0f, my_w1l = 0.0f, my_w1r = 0.0f; // BUS2: this lane's source on bus 0 / bus 1, per ear
	if constexpr (BUS2) {
		if (have) {
			const gas_bus_route r = routes[lm.slot];
			my_w0l = gas_bus_weight(r, bus_base, 0, 0);
			my_w0r = gas_bus_weight(r, bus_base, 0, 1);
			my_w1l = gas_bus_weight(r, bus_base + 1u, 0, 0);
			my_w1r = gas_bus_weight(r, bus_base + 1u, 0, 1);
		}
#pragma unroll
		for (int j = 0; j < 8; j++) {
			bus1_all[(wave * 2 + 0) * 512 + j * 64 + lane] = make_float2(0.0f, 0.0f);
			bus1_all[(wave * 2 + 1) * 512 + j * 64 + lane] = make_float2(0.0f, 0.0f);
		}
	}
	if (have) {
		const gas_params *P = fresh ? fresh + lm.row : st.params + lm.slot;
		const float2 gd = *reinterpret_cast<const float2 *>(&P->hrtf_gain); // hrtf_gain, hrtf_dir: one 8-byte load
		lm.g0 = st.hrtf_prev_gain[lm.slot];
		lm.g1 = gd.x;
		const uint32_t d = __float_as_uint(gd.y);
		lm.dir = d < tab.dirs ? d : 0;
		lm.pdir = lm.dir;
		if constexpr (SRC_PCM) {
			lm.cur = cursors[lm.slot];
		}
		const uint32_t e = my_entry;
		my_flag = (peak_all || e >= peak_from) ? 1u : (peak_bits ? (peak_bits[(e + peak_bit_base) >> 5] >> ((e + peak_bit_base) & 31)) & 1u : 0u);
		if (BUS2 && !commit) {
			my_flag = 0; // the peak (of y, before any bus factor) is the committing pass's
		}
	}
	if constexpr (SRC_PCM) {
		if (first < last) {
			load_window<true, FQ>(g, bcast_meta<true>(lm, 0, F), lane, fade_env, raw); // needs the cursor
		}
	}
	float *flt_par = flt_all + (FLT ? wave * 64 * FLT_W : 0);
	if constexpr (FLT) {
		if (have) {
			Coeffs co;
			if (flt_kind == GAS_FX_HIGHSHELF) {
				const gas_params *P = st.params + lm.slot;
				co = highshelf_coeffs(mix_rate, P->fx_shelf_cutoff_hz, P->fx_shelf_gain);
			} else {
				const gas_fx_settings *S = st.fxs + lm.slot;
				co = filter_coeffs((int)flt_kind, mix_rate, S->filter_cutoff_hz[flt_pos], S->filter_resonance[flt_pos], S->filter_gain[flt_pos]);
			}
			float *bq = st.bq;
			const size_t bs = st.bq_stride, s0 = ((size_t)lm.slot * 4 + flt_pos) * 2;
			float *my = flt_par + lane * FLT_W;
			my[0] = co.b0;
			my[1] = co.b1;
			my[2] = co.b2;
			my[3] = co.a1;
			my[4] = co.a2;
#pragma unroll
			for (int ear = 0; ear < 2; ear++) {
				my[5 + 4 * ear] = bq[BQ_HA1 * bs + s0 + ear];
				my[6 + 4 * ear] = bq[BQ_HA2 * bs + s0 + ear];
				my[7 + 4 * ear] = bq[BQ_HB1 * bs + s0 + ear];
				my[8 + 4 * ear] = bq[BQ_HB2 * bs + s0 + ear];
			}
			if (commit) { // the snapped coefficients are part of the processor too (k_biquad_mix stores them)
#pragma unroll
				for (int ear = 0; ear < 2; ear++) {
					bq[BQ_B0 * bs + s0 + ear] = co.b0;
					bq[BQ_B1 * bs + s0 + ear] = co.b1;
					bq[BQ_B2 * bs + s0 + ear] = co.b2;
					bq[BQ_A1 * bs + s0 + ear] = co.a1;
					bq[BQ_A2 * bs + s0 + ear] = co.a2;
				}
			}
		}
	}
	// LEAN: the stored half of source i's HRIR row, global -> LDS slot (i & 1), 4 x 1 KiB, no VGPR landing set
	auto table_dma = [&](uint32_t dir, uint32_t i) {
		const float4 *row = tab.spec + (size_t)dir * 256 + lane;
		float4 *dst = tslot + (i & 1u) * 256;
#pragma unroll
		for (int p = 0; p < 4; p++) {
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(row + p * 64), (__attribute__((address_space(3))) void *)(dst + p * 64), 16, 0, 0);
		}
	};
	if constexpr (LEAN && !GAS_UNI12_DIRECT) {
		if (first < last) {
			table_dma((uint32_t)__builtin_amdgcn_readlane((int)lm.dir, 0), 0u);
		}
	}
#pragma unroll
	for (int r = 0; r < TW_PER; r++) {
		if (threadIdx.x + r * UW * 64 < 512) {
			reinterpret_cast<float4 *>(tw_lds)[threadIdx.x + r * UW * 64] = tw_in[r];
		}
	}
	__syncthreads();
	float2 t1[8], t2[8];
#pragma unroll
	for (int k = 0; k < 8; k++) {
		t1[k] = tw_lds[k * 64 + lane];
		t2[k] = tw_lds[(8 + k) * 64 + lane];
	}
	// gain ramp weights of this lane's frames, t = f/F and 1 - t (exact for F = 128 .. 512): the same for every source
	// (LEAN rebuilds them per source, as k_hrtf_multi does: it has no registers to park them in)
	float tq[LEAN ? 1 : FQ], omtq[LEAN ? 1 : FQ];
	if constexpr (!LEAN) {
#pragma unroll
		for (int q = 0; q < FQ; q++) {
			tq[q] = (float)(lane + 64 * q) * (1.0f / (float)F);
			omtq[q] = 1 - tq[q];
		}
	}
	const float lane_f = (float)lane;
	GAS_UNI_STAMP(1);

	// ---- source loop: one forward FFT per source.  8 waves: its spectral products are taken one transform later (the
	// HRIR row can only be requested once the direction is known, and travels to VGPRs while the next source is
	// transformed).  12 waves: the row has been travelling to LDS since the previous trip -----------------------------
	float2 aYL[8], aYR[8]; // sum_s Z_s H_L[d_s], sum_s Z_s H_R[d_s] of this wave's sources
	float2 zp[LEAN ? 1 : 8]; // spectrum of the previous source, its table row (hs) in flight
	float4 hs[LEAN ? 1 : 8];
#pragma unroll
	for (int j = 0; j < 8; j++) {
		aYL[j] = make_float2(0.0f, 0.0f);
		aYR[j] = make_float2(0.0f, 0.0f);
		if constexpr (!LEAN) {
			zp[j] = make_float2(0.0f, 0.0f);
		}
	}
	bool have_prev = false; // wave-uniform
	uint32_t prev_flag = 0, prev_row = 0;
	float pw0l = 1.0f, pw0r = 1.0f, pw1l = 0.0f, pw1r = 0.0f; // BUS2: the previous source's bus weights (wave-uniform)

	// LEAN: the products of a source right after its transform, its row read from the wave's LDS slot (slot_read_spectra)
	float4 hd[LEAN && GAS_UNI12_DIRECT ? 8 : 1]; // GAS_UNI12_DIRECT: this source's row, requested at the top of its trip
	auto products_now = [&](const float2(&z)[8], uint32_t flag, uint32_t row, const float4 *slot) {
		float4 h[8];
		if constexpr (LEAN && GAS_UNI12_DIRECT) {
#pragma unroll
			for (int j = 0; j < 8; j++) {
				h[j] = hd[j];
			}
			finish_spectra(lane, h);
		} else {
			slot_read_spectra(slot, lane, h);
		}
#pragma unroll
		for (int j = 0; j < 8; j++) {
			cmac_fixed(aYL[j], z[j], h[j].x, h[j].y);
			cmac_fixed(aYR[j], z[j], h[j].z, h[j].w);
		}
		if (!flag) { // wave-uniform
			return;
		}
		// exact peak: one ear after the other through the wave's single exchange slice.  The inverse transforms take
		// their twiddles from the LDS copy at the point of use and the register copy is reloaded afterwards: 30 VGPRs
		// that are not live across this (rare) path.  Same twiddle values, same operations: the bits of fft512.
		float pk[2];
#pragma unroll
		for (int ear = 0; ear < 2; ear++) {
			float2 y[8];
#pragma unroll
			for (int j = 0; j < 8; j++) {
				y[j] = cmul_fixed(z[j], ear == 0 ? h[j].x : h[j].z, ear == 0 ? h[j].y : h[j].w);
			}
			if (!GAS_UNI12_DIRECT) {
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the compiler's ds_reads below would wait for the DMA anyway
			}
			fft512_twlds<true>(y, tw_lds, lds, lane);
			float p = 0.0f;
#pragma unroll
			for (int t = 0; t < SQ; t++) {
				p = fmaxf(p, fmaxf(fabsf(y[HQ + t].x), fabsf(y[HQ + t].y)));
			}
			pk[ear] = wave_max(p);
		}
		if (lane == 0) {
			g.peaks[(size_t)row * 2] = pk[0];
			g.peaks[(size_t)row * 2 + 1] = pk[1];
		}
#pragma unroll
		for (int k = 0; k < 8; k++) {
			t1[k] = tw_lds[k * 64 + lane];
			t2[k] = tw_lds[(8 + k) * 64 + lane];
		}
	};

	// 8 waves: products of the previous source (zp x hs), its exact peak if asked for, then the request for `next_dir`'s row
	auto products = [&](bool more, uint32_t next_dir) {
		if constexpr (!LEAN) {
			float2 yl[8], yr[8];
			if (have_prev && BUS2) {
				finish_spectra(lane, hs);
				float2 *b1 = bus1_all + wave * 2 * 512;
				const bool to1 = pw1l != 0.0f || pw1r != 0.0f; // wave-uniform
#pragma unroll
				for (int j = 0; j < 8; j++) {
					yl[j] = cmul_fixed(zp[j], hs[j].x, hs[j].y);
					yr[j] = cmul_fixed(zp[j], hs[j].z, hs[j].w);
					aYL[j] = make_float2(__builtin_fmaf(pw0l, yl[j].x, aYL[j].x), __builtin_fmaf(pw0l, yl[j].y, aYL[j].y));
					aYR[j] = make_float2(__builtin_fmaf(pw0r, yr[j].x, aYR[j].x), __builtin_fmaf(pw0r, yr[j].y, aYR[j].y));
				}
				if (to1) {
#pragma unroll
					for (int j = 0; j < 8; j++) {
						float2 a = b1[j * 64 + lane], b = b1[512 + j * 64 + lane];
						a = make_float2(__builtin_fmaf(pw1l, yl[j].x, a.x), __builtin_fmaf(pw1l, yl[j].y, a.y));
						b = make_float2(__builtin_fmaf(pw1r, yr[j].x, b.x), __builtin_fmaf(pw1r, yr[j].y, b.y));
						b1[j * 64 + lane] = a;
						b1[512 + j * 64 + lane] = b;
					}
				}
			} else if (have_prev) {
				finish_spectra(lane, hs);
#pragma unroll
				for (int j = 0; j < 8; j++) {
					cmac_fixed(aYL[j], zp[j], hs[j].x, hs[j].y);
					cmac_fixed(aYR[j], zp[j], hs[j].z, hs[j].w);
				}
				if (prev_flag) { // wave-uniform: this source's own output spectra (before the row's registers are requested again)
#pragma unroll
					for (int j = 0; j < 8; j++) {
						yl[j] = cmul_fixed(zp[j], hs[j].x, hs[j].y);
						yr[j] = cmul_fixed(zp[j], hs[j].z, hs[j].w);
					}
				}
			}
			if (more) {
				issue_spectra(tab.spec, next_dir, lane, hs);
			}
			if (have_prev && prev_flag) { // wave-uniform: this source's own output, for max |L|, max |R| (:436-443)
				if constexpr (SRC_PCM) { // the stream-sampling form has no registers for the interleaved pair (same operations, same bits)
					fft512<true>(yl, t1, t2, lds, lane);
					fft512<true>(yr, t1, t2, lds + LDS_F2_HALF, lane);
				} else {
					fft512_pair<true>(yl, yr, t1, t2, lds, lds + LDS_F2_HALF, lane);
				}
				float pkl = 0.0f, pkr = 0.0f;
#pragma unroll
				for (int t = 0; t < SQ; t++) { // valid outputs are window positions [512 - S, 512): registers j >= HQ
					pkl = fmaxf(pkl, fmaxf(fabsf(yl[HQ + t].x), fabsf(yl[HQ + t].y)));
					pkr = fmaxf(pkr, fmaxf(fabsf(yr[HQ + t].x), fabsf(yr[HQ + t].y)));
				}
				pkl = wave_max(pkl);
				pkr = wave_max(pkr);
				if (lane == 0) {
					g.peaks[(size_t)prev_row * 2] = pkl;
					g.peaks[(size_t)prev_row * 2 + 1] = pkr;
				}
			}
		}
	};

	for (uint32_t e = first; e < last; e++) {
		const bool has_next = e + 1 < last;
		const SrcMeta m = bcast_meta<SRC_PCM>(lm, e - first, F);
		const SrcMeta mn = bcast_meta<SRC_PCM>(lm, has_next ? e + 1 - first : e - first, F);
		const uint32_t flag = (uint32_t)__builtin_amdgcn_readlane((int)my_flag, (int)(e - first));
		if constexpr (LEAN && !GAS_UNI12_DIRECT) {
			// Everything this trip consumes has landed: the frames and history requested a trip ago, and -- older than
			// those in the wave's vector-memory queue, which retires in order -- this source's table row.  The next
			// source's row starts for the other slot now and has the whole trip to arrive.
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			if (has_next) {
				table_dma(mn.dir, e + 1 - first);
			}
		}
		if constexpr (LEAN && GAS_UNI12_DIRECT) {
			issue_spectra(tab.spec, m.dir, lane, hd); // lands under the window and the transform
		}

		// x_full[lane + 64 q]: q < HQ from the history, the rest from this callback's frames
		float xq[NQ];
#pragma unroll
		for (int q = 0; q < HQ; q++) {
			xq[q] = rawh[q];
		}
		if constexpr (LEAN) {
			float lf = lane_f;
			asm volatile("" : "+v"(lf)); // keeps the ramp weights out of registers across the loop (k_hrtf_multi's form, same bits)
#pragma unroll
			for (int q = 0; q < FQ; q++) {
				const float mono = (raw[q].left + raw[q].right) * 0.5f;
				const float t = (lf + (float)(64 * q)) * (1.0f / (float)F);
				xq[HQ + q] = mono * (m.g1 * t + (1 - t) * m.g0);
			}
		} else if constexpr (ER) {
			// early reflections (oracle fx_early_reflections): this block into the ring, then per frame the taps in tap
			// order, f32; 16 tap loads in flight per pass (more spill), results handed over through the wave's LDS slice
			const gas_params *P = fresh ? fresh + m.row : st.params + m.slot; // device-published rows of this callback are not in the table yet
			const gas_audio_frame *srow = g.src + (size_t)m.row * F;
			const uint32_t er_pos = st.er_pos[m.slot];
			gas_audio_frame *ring = st.er_ring + (size_t)m.slot * er_R;
			float *xs = reinterpret_cast<float *>(lds);
#pragma unroll
			for (int q = 0; q < FQ; q++) { // unrolled (raw[] stays in registers), fenced into passes of two frames
				if (q && q % 2 == 0) {
					__builtin_amdgcn_sched_barrier(0);
				}
				const int f = lane + 64 * q;
				const gas_audio_frame fr = raw[q];
				ring[(er_pos + (uint32_t)f) & (er_R - 1)] = fr;
				float yl = fr.left, yr = fr.right;
#pragma unroll
				for (int k = 0; k < GAS_ER_TAPS; k++) {
					const uint32_t du = P->er_delay[k];
					const int d = (int)(du < er_R - F ? du : er_R - F); // keeps every tap inside row / ring
					const float gk = P->er_gain[k];
					const int i = f - d;
					const gas_audio_frame xp = i >= 0 ? srow[i] : ring[(er_pos + (uint32_t)(i + (int)er_R)) & (er_R - 1)];
					yl = yl + gk * xp.left;
					yr = yr + gk * xp.right;
				}
				xs[f] = (yl + yr) * 0.5f;
			}
			if (lane == 0) {
				st.er_pos[m.slot] = (er_pos + F) & (er_R - 1);
			}
			wave_lds_sync();
#pragma unroll
			for (int q = 0; q < FQ; q++) {
				xq[HQ + q] = xs[lane + 64 * q] * (m.g1 * tq[q] + omtq[q] * m.g0);
			}
			wave_lds_sync();
		} else if constexpr (FLT) {
			constexpr int P = FQ; // consecutive frames per lane in the filter's layout
			const float *hp = flt_par + (e - first) * FLT_W; // wave-uniform address (written before the prologue's barrier)
			Coeffs co;
			co.b0 = hp[0];
			co.b1 = hp[1];
			co.b2 = hp[2];
			co.a1 = hp[3];
			co.a2 = hp[4];
			float2 *xs = lds; // the block's frames, both ears
			float *bq = st.bq;
			const size_t bs = st.bq_stride, s0 = ((size_t)m.slot * 4 + flt_pos) * 2;
#pragma unroll
			for (int q = 0; q < FQ; q++) {
				xs[lane + 64 * q] = make_float2(raw[q].left, raw[q].right);
			}
			wave_lds_sync();
			const bool serial = !(fabsf(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(co.a2)))) <= 0.9f); // wave-uniform; NaN goes serial too
			if (serial) {
				if (lane < 2) { // each ear on its own lane, [ENGINE] process_one's operation order, no contraction
					float a1 = hp[5 + 4 * lane], a2 = hp[6 + 4 * lane], b1 = hp[7 + 4 * lane], b2 = hp[8 + 4 * lane];
					float *sr = reinterpret_cast<float *>(xs);
					for (uint32_t i = 0; i < F; i++) {
						sr[2 * i + lane] = biquad_process_one(co, sr[2 * i + lane], a1, a2, b1, b2);
					}
					bq[BQ_HA1 * bs + s0 + lane] = a1;
					bq[BQ_HA2 * bs + s0 + lane] = a2;
					bq[BQ_HB1 * bs + s0 + lane] = b1;
					bq[BQ_HB2 * bs + s0 + lane] = b2;
				}
			} else {
				float x[2][P], y[2][P], p[P], q2[P];
#pragma unroll
				for (int k = 0; k < P; k++) {
					const float2 v = xs[lane * P + k];
					x[0][k] = v.x;
					x[1][k] = v.y;
				}
				// homogeneous solutions p (state (1, 0)) and q (state (0, 1)): the same in every lane and for both ears
				p[0] = co.a1;
				q2[0] = co.a2;
				p[1] = __builtin_fmaf(co.a1, p[0], co.a2);
				q2[1] = __fmul_rn(co.a1, q2[0]);
#pragma unroll
				for (int k = 2; k < P; k++) {
					p[k] = __builtin_fmaf(co.a1, p[k - 1], __fmul_rn(co.a2, p[k - 2]));
					q2[k] = __builtin_fmaf(co.a1, q2[k - 1], __fmul_rn(co.a2, q2[k - 2]));
				}
				float sx[2], sy[2]; // the state leaving this lane, per ear
#pragma unroll
				for (int ear = 0; ear < 2; ear++) {
					const float ha1 = hp[5 + 4 * ear], ha2 = hp[6 + 4 * ear];
					float xm1 = __shfl_up(x[ear][P - 1], 1, 64), xm2 = __shfl_up(x[ear][P - 2], 1, 64);
					if (lane == 0) {
						xm1 = hp[7 + 4 * ear];
						xm2 = hp[8 + 4 * ear];
					}
					float w1 = 0.0f, w2 = 0.0f;
#pragma unroll
					for (int k = 0; k < P; k++) {
						const float xa = k >= 1 ? x[ear][k - 1] : xm1;
						const float xb = k >= 2 ? x[ear][k - 2] : (k == 1 ? xm1 : xm2);
						const float u = __builtin_fmaf(co.b2, xb, __builtin_fmaf(co.b1, xa, __fmul_rn(co.b0, x[ear][k])));
						const float w = __builtin_fmaf(co.a2, w2, __builtin_fmaf(co.a1, w1, u));
						y[ear][k] = w;
						w2 = w1;
						w1 = w;
					}
					sx[ear] = y[ear][P - 1];
					sy[ear] = y[ear][P - 2];
					if (lane == 0) { // the processor's history enters through lane 0
						sx[ear] = __builtin_fmaf(p[P - 1], ha1, __builtin_fmaf(q2[P - 1], ha2, sx[ear]));
						sy[ear] = __builtin_fmaf(p[P - 2], ha1, __builtin_fmaf(q2[P - 2], ha2, sy[ear]));
					}
				}
				float m00 = p[P - 1], m01 = q2[P - 1], m10 = p[P - 2], m11 = q2[P - 2];
#pragma unroll
				for (int d = 1; d < 64; d *= 2) { // inclusive scan of s_l = M s_(l-1) + c_l: distances 1, 2, 4, ... with M, M^2, M^4, ...
#pragma unroll
					for (int ear = 0; ear < 2; ear++) {
						const float ox = __shfl_up(sx[ear], d, 64), oy = __shfl_up(sy[ear], d, 64);
						if (lane >= d) {
							sx[ear] = __builtin_fmaf(m00, ox, __builtin_fmaf(m01, oy, sx[ear]));
							sy[ear] = __builtin_fmaf(m10, ox, __builtin_fmaf(m11, oy, sy[ear]));
						}
					}
					const float n00 = __builtin_fmaf(m00, m00, __fmul_rn(m01, m10)), n01 = __builtin_fmaf(m00, m01, __fmul_rn(m01, m11));
					const float n10 = __builtin_fmaf(m10, m00, __fmul_rn(m11, m10)), n11 = __builtin_fmaf(m10, m01, __fmul_rn(m11, m11));
					m00 = n00;
					m01 = n01;
					m10 = n10;
					m11 = n11;
				}
#pragma unroll
				for (int ear = 0; ear < 2; ear++) {
					float ix = __shfl_up(sx[ear], 1, 64), iy = __shfl_up(sy[ear], 1, 64); // the state entering this lane
					if (lane == 0) {
						ix = hp[5 + 4 * ear];
						iy = hp[6 + 4 * ear];
					}
#pragma unroll
					for (int k = 0; k < P; k++) {
						y[ear][k] = __builtin_fmaf(p[k], ix, __builtin_fmaf(q2[k], iy, y[ear][k]));
					}
				}
#pragma unroll
				for (int k = 0; k < P; k++) {
					xs[lane * P + k] = make_float2(y[0][k], y[1][k]);
				}
				if (lane == 63) { // the processor after the block: last two outputs, last two inputs
#pragma unroll
					for (int ear = 0; ear < 2; ear++) {
						bq[BQ_HA1 * bs + s0 + ear] = y[ear][P - 1];
						bq[BQ_HA2 * bs + s0 + ear] = y[ear][P - 2];
						bq[BQ_HB1 * bs + s0 + ear] = x[ear][P - 1];
						bq[BQ_HB2 * bs + s0 + ear] = x[ear][P - 2];
					}
				}
			}
			wave_lds_sync();
#pragma unroll
			for (int q = 0; q < FQ; q++) {
				const float2 yy = xs[lane + 64 * q];
				xq[HQ + q] = ((yy.x + yy.y) * 0.5f) * (m.g1 * tq[q] + omtq[q] * m.g0);
			}
			wave_lds_sync();
		} else {
#pragma unroll
			for (int q = 0; q < FQ; q++) {
				const float mono = (raw[q].left + raw[q].right) * 0.5f;
				xq[HQ + q] = mono * (m.g1 * tq[q] + omtq[q] * m.g0);
			}
		}
#ifdef GAS_STAMPS
		if (e == first) {
			GAS_UNI_STAMP(2);
		}
#endif
		if (!BUS2 || commit) { // wave-uniform; a non-committing bus pass leaves the state for the next pass to read
			store_history<HQ>(st.hrtf_hist + (size_t)m.slot * HL, lane, &xq[FQ], nt_hist != 0); // new history = x_full[F .. F + HL)
		}
		if constexpr (SRC_PCM) {
			if (lane == 0 && m.hf) { // advance the playback cursor (audio_spatializer.cpp:378,398)
				cursors[m.slot].pos = m.pos + m.mixed;
				if (m.mixed != F) {
					cursors[m.slot].has_frames = 0;
				}
			}
		}
		if (has_next) {
			load_history<HQ>(st.hrtf_hist + (size_t)mn.slot * HL, lane, rawh, nt_hist != 0);
			load_window<SRC_PCM, FQ>(g, mn, lane, fade_env, raw);
		}
		// z = a + i b : a = x_full[0..512), b = x_full[S..S+512) -- one complex FFT serves both sub-blocks
		float2 zs[8];
#pragma unroll
		for (int j = 0; j < 8; j++) {
			zs[j] = make_float2(xq[j], xq[j + SQ]);
		}
		if constexpr (LEAN) {
			if constexpr (GAS_UNI12_DIRECT) {
				fft512<false>(zs, t1, t2, lds, lane);
			} else {
				fft512_ar<false>(zs, t1, t2, lds, lane);
			}
			products_now(zs, flag, (uint32_t)__builtin_amdgcn_readlane((int)lm.prow, (int)(e - first)), tslot + ((e - first) & 1u) * 256);
		} else {
			fft512<false>(zs, t1, t2, lds, lane);
			// the products of a transform are taken one transform later (measured: taking them in front of the next
			// transform instead, which saves the 16-register copy, costs 0.5 us per launch -- the row needs the time)
			products(true, m.dir);
#pragma unroll
			for (int j = 0; j < 8; j++) {
				zp[j] = zs[j];
			}
			have_prev = true;
			prev_flag = flag;
			prev_row = (uint32_t)__builtin_amdgcn_readlane((int)lm.prow, (int)(e - first)); // where this source's peak goes
			if constexpr (BUS2) {
				const int i = (int)(e - first);
				pw0l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w0l), i));
				pw0r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w0r), i));
				pw1l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w1l), i));
				pw1r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w1r), i));
			}
		}
	}
	GAS_UNI_STAMP(3);

	// ---- behind the loop: what nobody waits for ------------------------------------------------------------------
	if (have && (!BUS2 || commit)) {
		st.hrtf_prev_gain[lm.slot] = lm.g1;
		if (!my_flag) { // "not measured": never passes the gate (audio_spatializer.cpp:464-469)
			*reinterpret_cast<float2 *>(g.peaks + (size_t)lm.prow * 2) = make_float2(__builtin_inff(), __builtin_inff());
		}
		if (fresh) { // device-published parameter rows go through to the slot table (saves the scatter launch)
			const float4 *src4 = reinterpret_cast<const float4 *>(fresh + lm.row);
			float4 *dst4 = reinterpret_cast<float4 *>(st.params + lm.slot);
#pragma unroll
			for (int k = 0; k < 8; k++) {
				dst4[k] = src4[k];
			}
		}
	}
	// GAS_FLAG_PIPELINED_MIX: waves 2..5 of the first workgroups each sum one float4 column of the PREVIOUS callback's
	// partial mixes (k_mix_reduce's job, same code, same bits); the rows are requested here and land under the
	// barrier wait and the last products.  Column -> (workgroup, wave) as in k_hrtf_ols: the 8 columns of one
	// 128-byte line go to two workgroups of the same XCD.
	float4 jr[JOB_ROWS];
	bool job_mine = false;
	uint32_t job_col = 0;
	if constexpr (!SRC_PCM) {
		const uint32_t jw = (uint32_t)(wave - 2), jx = blockIdx.x & 7, ji = blockIdx.x >> 3;
		job_col = (((ji >> 1) * 8 + jx) * 8) + (ji & 1) * 4 + jw;
		job_mine = job.partials != nullptr && wave >= 2 && jw < GAS_HRTF_JOB_WAVES && job_col < job.elems / 4; // wave-uniform
		if (job_mine) {
			job_issue(job, job_col, lane, jr);
		}
	}
	products(false, 0); // 8 waves: the last source's products (and peak)

	// ---- epilogue: spectra of all waves -> fd[wave][ear][j][lane]; wave 0 transforms the left ear's sum and wave 1
	// the right ear's; the workgroup stores one interleaved time-domain partial mix ----------------------------------
	float2 *fd = lds_all;
	float *outp = reinterpret_cast<float *>(lds_all + FD_F2 + 2 * LDS_F2_HALF);
	__syncthreads(); // every wave is done with its exchange slices and table slots (fd aliases them)
#pragma unroll
	for (int j = 0; j < 8; j++) {
		fd[(wave * 2 + 0) * 512 + j * 64 + lane] = aYL[j];
		fd[(wave * 2 + 1) * 512 + j * 64 + lane] = aYR[j];
	}
	__syncthreads(); // fd is complete
	GAS_UNI_STAMP(4);
	if (job_mine) {
		job_finish(job, job_col, lane, jr);
	}
	if (wave < 2) {
		// wave `ear` adds the UW spectra of its ear in wave order: ((w0 + w1) + w2) + ...
		float2 y[8];
#pragma unroll
		for (int j = 0; j < 8; j++) {
			y[j] = fd[wave * 512 + j * 64 + lane];
		}
#pragma unroll
		for (int w = 1; w < UW; w++) {
#pragma unroll
			for (int j = 0; j < 8; j++) {
				y[j] = cadd(y[j], fd[(w * 2 + wave) * 512 + j * 64 + lane]);
			}
		}
		fft512<true>(y, t1, t2, lds_all + FD_F2 + wave * LDS_F2_HALF, lane);
#pragma unroll
		for (int t = 0; t < SQ; t++) {
			const int fa = lane + 64 * t, fb = lane + 64 * (SQ + t);
			outp[fa * 2 + wave] = y[HQ + t].x;
			outp[fb * 2 + wave] = y[HQ + t].y;
		}
	}
	__syncthreads();
	for (int idx = threadIdx.x; idx < (int)(F * 2); idx += UW * 64) {
		my_partial[idx] = outp[idx];
	}
	if constexpr (BUS2) { // the same once more for bus 1, whose spectra have been in LDS all along
		__syncthreads(); // outp has been read
		if (wave < 2) {
			float2 y[8];
#pragma unroll
			for (int j = 0; j < 8; j++) {
				y[j] = bus1_all[wave * 512 + j * 64 + lane];
			}
#pragma unroll
			for (int w = 1; w < UW; w++) {
#pragma unroll
				for (int j = 0; j < 8; j++) {
					y[j] = cadd(y[j], bus1_all[(w * 2 + wave) * 512 + j * 64 + lane]);
				}
			}
			fft512<true>(y, t1, t2, lds_all + FD_F2 + wave * LDS_F2_HALF, lane);
#pragma unroll
			for (int t = 0; t < SQ; t++) {
				const int fa = lane + 64 * t, fb = lane + 64 * (SQ + t);
				outp[fa * 2 + wave] = y[HQ + t].x;
				outp[fb * 2 + wave] = y[HQ + t].y;
			}
		}
		__syncthreads();
		float *bus1_partial = my_partial + (size_t)bus_rows * (size_t)(F * 2);
		for (int idx = threadIdx.x; idx < (int)(F * 2); idx += UW * 64) {
			bus1_partial[idx] = outp[idx];
		}
	}
	GAS_UNI_STAMP(5);
#ifdef GAS_STAMPS
	if (lane == 0 && (blockIdx.x * UW + wave) < 8192) {
		gas_stamps[(blockIdx.x * UW + wave) * GAS_STAMP_SLOTS + 7] = __builtin_amdgcn_s_memtime();
	}
#endif
}

} // namespace

// One workgroup per CU (one residency round), at least one source per wave, at most 64 (one metadata lane each).
// The grid is sized for the eight-wave form; the twelve-wave form runs on the same grid (same partial rows) whenever
// every one of its waves still gets a source.
constexpr uint32_t UNI_W = 8, UNI_W12 = 12;

uint32_t gas_hrtf_uni_waves() {
	return UNI_W;
}

uint32_t gas_hrtf_uni_partials(uint32_t n) {
	const uint32_t want = (n + UNI_W - 1) / UNI_W, need = (n + UNI_W * 64 - 1) / (UNI_W * 64);
	const uint32_t w = want < 256u ? want : 256u; // workgroups resident at once
	return w > need ? w : need;
}

// Which callbacks run three waves per SIMD: those of at least this many sources (0 = none).  Default: the build's
// GAS_UNI12_DEFAULT_MIN, overridden by the environment variable GAS_UNI12_MIN (read once) or gas_tune_uni12_min().
static uint32_t g_uni12_min = [] {
	const char *e = getenv("GAS_UNI12_MIN");
	return e ? (uint32_t)strtoul(e, nullptr, 10) : (uint32_t)GAS_UNI12_DEFAULT_MIN;
}();

static uint32_t uni12_min_sources() {
	return g_uni12_min;
}

extern "C" uint32_t gas_tune_uni12_min(uint32_t min_sources) {
	const uint32_t was = g_uni12_min;
	g_uni12_min = min_sources;
	return was;
}

// From how many sources a callback's history rows (1 KiB each at F = 512, read and written once per callback) are
// accessed non-temporal: when they no longer survive in the 256 MiB Infinity Cache from one callback to the next.
// GAS_NT_HIST_MIN / gas_tune_nt_hist_min() override (measurement).
static uint32_t g_nt_hist_min = [] {
	const char *e = getenv("GAS_NT_HIST_MIN");
	return e ? (uint32_t)strtoul(e, nullptr, 10) : 196608u;
}();

static uint32_t nt_hist_min_sources() {
	return g_nt_hist_min;
}

extern "C" uint32_t gas_tune_nt_hist_min(uint32_t min_sources) {
	const uint32_t was = g_nt_hist_min;
	g_nt_hist_min = min_sources;
	return was;
}

bool gas_hrtf_uni_twelve(uint32_t n, bool streams, bool buses) {
	const uint32_t wgs = gas_hrtf_uni_partials(n), mn = uni12_min_sources();
	return !streams && !buses && mn != 0 && n >= mn && n >= wgs * UNI_W12 && n <= wgs * UNI_W12 * 64;
}

hipError_t gas_launch_hrtf_uni(hipStream_t stream, const gas_group_args &g, const uint32_t *peak_bits, bool peak_all, const gas_dev_state &st, const gas_hrtf_table &tab, const float2 *twiddles, uint32_t frames, uint32_t hist_len, float *partials, uint32_t p_offset, gas_cursor *cursors, const float *fade_env, const gas_params *fresh, const gas_deferred_reduce &job, const gas_bus_route *routes, uint32_t bus_rows, uint32_t bus_base, bool commit, uint32_t er_ring_frames, uint32_t peak_from, uint32_t peak_bit_base, uint32_t flt_kind, uint32_t flt_pos, float mix_rate) {
	if (g.n == 0) {
		return hipSuccess;
	}
	if (frames % 128 != 0 || frames > 512 || hist_len != 512 - frames / 2 || (routes && (cursors || g.order)) || ((er_ring_frames || flt_kind) && (routes || cursors || g.order)) || (er_ring_frames && flt_kind)) {
		return hipErrorInvalidValue;
	}
	const uint32_t wgs = gas_hrtf_uni_partials(g.n);
	const uint32_t all = peak_all ? 1u : 0u;
	const bool twelve = er_ring_frames == 0 && flt_kind == 0 && gas_hrtf_uni_twelve(g.n, cursors != nullptr, routes != nullptr);
	const uint32_t nt_hist = g.n >= nt_hist_min_sources() ? 1u : 0u; // history rows larger than what stays cached between callbacks
	const dim3 grid(wgs), block((twelve ? UNI_W12 : UNI_W) * 64);
#define GAS_UNI_GO(SQv, PCM, BUS, W) hipLaunchKernelGGL((k_hrtf_uni<SQv, PCM, BUS, W>), grid, block, 0, stream, g, peak_bits, all, st, tab, twiddles, partials, p_offset, cursors, fade_env, fresh, job, routes, bus_rows, bus_base, commit ? 1u : 0u, nt_hist, er_ring_frames, peak_from, peak_bit_base, flt_kind, flt_pos, mix_rate)
#define GAS_UNI_CASE(SQv)                      \
	case SQv:                                  \
		if (flt_kind) {                        \
			hipLaunchKernelGGL((k_hrtf_uni<SQv, false, false, 8, false, true>), grid, block, 0, stream, g, peak_bits, all, st, tab, twiddles, partials, p_offset, cursors, fade_env, fresh, job, routes, bus_rows, bus_base, commit ? 1u : 0u, nt_hist, er_ring_frames, peak_from, peak_bit_base, flt_kind, flt_pos, mix_rate); \
		} else if (er_ring_frames) {           \
			hipLaunchKernelGGL((k_hrtf_uni<SQv, false, false, 8, true>), grid, block, 0, stream, g, peak_bits, all, st, tab, twiddles, partials, p_offset, cursors, fade_env, fresh, job, routes, bus_rows, bus_base, commit ? 1u : 0u, nt_hist, er_ring_frames, peak_from, peak_bit_base, flt_kind, flt_pos, mix_rate); \
		} else if (routes) {                   \
			GAS_UNI_GO(SQv, false, true, 8);   \
		} else if (cursors) {                  \
			GAS_UNI_GO(SQv, true, false, 8);   \
		} else if (twelve) {                   \
			GAS_UNI_GO(SQv, false, false, 12); \
		} else {                               \
			GAS_UNI_GO(SQv, false, false, 8);  \
		}                                      \
		break;
	switch (frames / 128) {
		GAS_UNI_CASE(1)
		GAS_UNI_CASE(2)
		GAS_UNI_CASE(3)
		GAS_UNI_CASE(4)
		default:
			return hipErrorInvalidValue;
	}
#undef GAS_UNI_CASE
#undef GAS_UNI_GO
	return hipGetLastError();
}

#ifdef GAS_STAMPS
// Diagnostic builds only: copies the last launch's per-wave stamps ([wave][8] of 100 MHz ticks) to the host.
extern "C" int gas_debug_read_stamps(unsigned long long *out, unsigned long long count) {
	const size_t total = sizeof(gas_stamps) / sizeof(gas_stamps[0]);
	const size_t n = count < total ? count : total;
	if (hipDeviceSynchronize() != hipSuccess) {
		return -7;
	}
	return hipMemcpyFromSymbol(out, HIP_SYMBOL(gas_stamps), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -7;
}
#endif
