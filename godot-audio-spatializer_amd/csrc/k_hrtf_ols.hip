// k_hrtf_ols.hip -- per-source HRTF convolution by overlap-save FFT, fused with the (optional)
// 8-tap early-reflection effect in front of it and with the N-source -> stereo partial sum.
//
// NEW arithmetic (the reference has no HRTF/FFT/convolution); it sits where
// AudioSpatializerInstanceEffect::process_frames runs its effect chain (audio_spatializer_effect.cpp:52-76)
// and feeds the accumulate + per-source peak of _mix_from_playback_list (audio_spatializer.cpp:449-461).
// Semantics are fixed by oracle/gas_oracle.c (fx_early_reflections, fx_hrtf):
//   e   = src + sum_k er_gain[k] * src delayed by er_delay[k]           (only when the chain is [ER, HRTF])
//   x   = ((e.l + e.r) * 0.5) * (g1*t + (1-t)*g0),  t = i/F             (gain ramp as audio_spatializer_3d.cpp:591-592)
//   out = (hrir[dir][L] * x, hrir[dir][R] * x)  over  hist ++ x         (256 taps, direction switches per block)
//
// Algorithm per source and callback (DESIGN.md "k_hrtf_ols"): the callback's F frames are cut into two
// sub-blocks of S = F/2; each needs a 512-sample window (S new + 512-S history >= 255) and the two real
// windows ride one complex FFT as  z = a + i b.  Because h_L and h_R are real,  IFFT(Z * H_L) =
// a*h_L + i b*h_L, so one forward and two inverse 512-point FFTs give both sub-blocks of both ears with
// no spectrum unpacking: 3 FFT-512 per source per callback for any F in {128,256,384,512}.
//
// Mapping (CDNA4, wave64): one wave owns one source at a time; lane l holds points l + 64 j (j = 0..7)
// in registers, so every global access is lane-contiguous.  512 = 8*8*8: three in-register radix-8 passes
// with two 8x8 lane<->register transposes through the wave's private LDS slice (row strides 72 and 66
// float2 = conflict-free for ds_write_b64/ds_read_b64, see MI355X_MICROARCH.md LDS banking).  Twiddles
// and the wave's running stereo sum stay in registers across its sources; HRIR spectra come from a
// lane-major table (16 B/lane, L2/Infinity-Cache resident).  Waves of a workgroup combine through LDS
// into one partial mix; k_mix_reduce adds the partials in fixed order (no float atomics).
//
// Bound: HBM.  Algorithmic bytes/source as group_bytes() (gas_ctx.hip) counts them: F*8 (source) + 2*hist_len*4
// (history read + write) + 24 (hrtf_gain + hrtf_dir read, previous gain read + write, peak write), + the ring
// traffic (8 taps * F * 8 read + F * 8 write) and the 8 gains + 8 delays with early reflections; the HRIR spectra
// table and the output mix are counted once per launch, not per source.  (The full 128-byte gas_params row is NOT
// part of the figure: the kernel reads its two HRTF fields, one 8-byte load.)  This file keeps the forms with
// crossfade / direction runs / early reflections / per-source rows; the plain [HRTF] callback runs k_hrtf_uni.hip.
// source rows non-temporal, like k_hrtf_uni (round 3): the cross-fade form gains 3 %, [ER, HRTF] nothing
#define GAS_USE_NT 1
#include "gas_hrtf_wave.h"

namespace {

// SQ = S/64 = F/128: 4 for F = 512, 2 for F = 256.
//
// PEAKS = true : every source gets its own pair of inverse FFTs, so its output peak (the input of the
//                host's silence gate, audio_spatializer.cpp:436-443,464-469) is exact; the wave sums
//                its sources in the time domain.
// PEAKS = false: the sum over sources commutes with the inverse transform, so a wave accumulates
//                Z_s * H_L[d_s] and Z_s * H_R[d_s] in the frequency domain (one forward FFT per source)
//                and the workgroup runs ONE pair of inverse FFTs at the end.  peaks[] gets +inf
//                ("not measured": never satisfies the gate).  The context routes a source here unless
//                it is marked draining (gas_source_set_draining) or peaks were asked for every source.
//
// Software pipeline over the wave's sources: while source e runs its FFTs, the frames and history of
// source e+1 (issued right after e's were consumed) and its HRIR spectra (issued right after e's
// spectral products) are in flight, so the arithmetic hides the HBM / L2 latency.
template <int SQ, bool PEAKS>
struct HrtfLds {
	static constexpr int F = 2 * SQ * 64;
	static constexpr int FD_F2 = WAVES * 2 * 512;
	static constexpr int TOTAL_F2 = PEAKS ? WAVES * LDS_F2_PER_WAVE : (FD_F2 + 2 * LDS_F2_HALF + F > WAVES * LDS_F2_PER_WAVE ? FD_F2 + 2 * LDS_F2_HALF + F : WAVES * LDS_F2_PER_WAVE);
};

template <int SQ, bool WITH_ER, bool PEAKS, bool SRC_PCM, bool XFADE, bool RUNS = false>
__device__ __forceinline__ void hrtf_body(float2 *lds_all, float2 *tw_lds, const uint32_t wg, const gas_group_args &g, const gas_dev_state &st, const gas_hrtf_table &tab, const float2 *__restrict__ tw, uint32_t n_wgs, uint32_t er_R, float *__restrict__ my_partial, gas_cursor *__restrict__ cursors, const float *__restrict__ fade_env, const gas_params *__restrict__ fresh, gas_audio_frame *__restrict__ rows_out = nullptr, const gas_deferred_reduce job = gas_deferred_reduce(), uint32_t job_col = 0) {
	static_assert(!(WITH_ER && SRC_PCM), "the early-reflection prologue reads float rows");
	constexpr int FQ = 2 * SQ; // F / 64
	constexpr int HQ = 8 - SQ; // hist_len / 64
	constexpr int NQ = 8 + SQ; // (hist_len + F) / 64
	constexpr uint32_t F = FQ * 64;
	constexpr uint32_t HL = HQ * 64;
	// LDS (float2 units): main loop = one exchange slice pair per wave; the PEAKS=false epilogue re-uses
	// the front as fd[wave][ear][512] and needs two more exchange slices + the [F][2] output behind it.
	constexpr int FD_F2 = HrtfLds<SQ, PEAKS>::FD_F2;
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	float2 *lds = lds_all + wave * LDS_F2_PER_WAVE;

	GAS_STAMP(0);
#ifdef GAS_STAMPS
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // diagnostic: the kernel arguments have arrived
	GAS_STAMP(6);
#endif
	// twiddles: the workgroup's 512 threads fetch the 8 KiB table once (16 bytes each) and park it in LDS; the
	// per-wave register copies are filled from there after the first barrier (below, behind the metadata loads)
	float4 tw_in = make_float4(0.f, 0.f, 0.f, 0.f);
	if (GAS_TW_LDS && !(GAS_ABL & 16)) {
		tw_in = reinterpret_cast<const float4 *>(tw)[threadIdx.x];
	}
	float2 t1[8], t2[8];
	if (!GAS_TW_LDS || (GAS_ABL & 16)) {
#pragma unroll
		for (int k = 0; k < 8; k++) {
			t1[k] = (GAS_ABL & 16) ? make_float2(0.001f * (float)(lane + k), 1.0f) : tw[k * 64 + lane];
			t2[k] = (GAS_ABL & 16) ? make_float2(1.0f, 0.002f * (float)(lane - k)) : tw[(8 + k) * 64 + lane];
		}
	}

	// GAS_FLAG_PIPELINED_MIX: one wave also sums the previous callback's partial mixes for one output column
	// (waves 2 .. 5, idle while waves 0 and 1 run the frequency-domain epilogue's inverse FFTs, take one column each)
	constexpr bool JOB_OK = !SRC_PCM && !XFADE; // register budget: not the stream-sampling and cross-fade forms
	constexpr bool JOB_LATE = WITH_ER; // early reflections: no 16 registers to park the rows across the loop -- they are requested where they are summed
	float4 jr[JOB_ROWS];
	bool job_mine = false;
	if constexpr (JOB_OK) {
		// Column -> (workgroup, wave): the 8 float4 columns of one 128-byte line go to waves 2..5 of two workgroups
		// of the same XCD (workgroups are dealt round-robin over the 8 XCDs), so every line of the partials is pulled
		// into one L2 only: job_col holds the workgroup index w; XCD x = w % 8, i = w / 8.
		const uint32_t jw = (uint32_t)(wave - 2), jx = job_col & 7, ji = job_col >> 3;
		job_col = (((ji >> 1) * 8 + jx) * 8) + (ji & 1) * 4 + jw;
		job_mine = job.partials != nullptr && wave >= 2 && jw < GAS_HRTF_JOB_WAVES && job_col < job.elems / 4; // wave-uniform
		if (job_mine && !JOB_LATE) {
			job_issue(job, job_col, lane, jr);
		}
	}

	// running sum of this wave's sources: time domain (PEAKS) or frequency domain
	// XFADE: a source whose direction changed since its previous callback is rendered with both
	// HRIRs and blended with t = f/F across the block.  The weights depend on the frame only, so the frequency-
	// domain path just keeps a second spectrum sum: aY* = sum Z*H[new], bY* = sum Z*H[old] (sources whose direction
	// did not change go into both), and the epilogue outputs t*IFFT(aY) + (1-t)*IFFT(bY).
	float accL[FQ], accR[FQ];
	float2 aYL[8], aYR[8], bYL[8], bYR[8];
#pragma unroll
	for (int t = 0; t < FQ; t++) {
		accL[t] = 0.0f;
		accR[t] = 0.0f;
	}
#pragma unroll
	for (int j = 0; j < 8; j++) {
		aYL[j] = make_float2(0.0f, 0.0f);
		aYR[j] = make_float2(0.0f, 0.0f);
		bYL[j] = make_float2(0.0f, 0.0f);
		bYR[j] = make_float2(0.0f, 0.0f);
	}

	// PEAKS = false, RUNS: time-domain sum of the windows of a run of sources that share one HRIR direction
	// (sum_s Z_s H[d] = FFT(sum_s z_s) H[d]): the forward FFT, the table row and the spectral products are paid per
	// run, not per source.  Runs are long when the caller keeps its list grouped by direction or the context ordered
	// the group (g.order).  Without RUNS every source is transformed on its own: the run bookkeeping costs 4-5 % on
	// lists without runs (measured), so it is a launch-time choice (GAS_FLAG_DIRECTION_RUNS / _ORDER).
	float2 zs[8];
#pragma unroll
	for (int j = 0; j < 8; j++) {
		zs[j] = make_float2(0.0f, 0.0f);
	}
	// SKEW: the spectral products of a transform are taken one transform later.  A source's HRIR row can only be
	// requested once its direction is known (parameter row -> direction -> table row: two dependent round trips,
	// 2.7 us of a 16.7 us launch when the first product waited for them), whereas its frames need one; with the skew
	// the table row of run r travels while run r+1 is loaded and transformed.
	constexpr bool SKEW = !PEAKS && !XFADE;
	float2 zp[8]; // spectrum of the previous run, its table row in flight
	bool have_prev = false; // wave-uniform
#pragma unroll
	for (int j = 0; j < 8; j++) {
		zp[j] = make_float2(0.0f, 0.0f);
	}

	uint32_t first, last;
	wave_range(g.n, wg * WAVES + wave, n_wgs * WAVES, first, last);

	// in-flight buffers of the software pipeline
	float4 hs[8]; // spectra (HL.re, HL.im, HR.re, HR.im) of bins lane + 64 j
#ifndef GAS_HRTF_STAGES
#define GAS_HRTF_STAGES 1 // 2 was measured no faster (17.7 vs 16.6 us at 8192 sources): the loop is not latency-bound per wave
#endif
	constexpr int STAGES = (WITH_ER || SRC_PCM || XFADE) ? 1 : GAS_HRTF_STAGES;
	gas_audio_frame raw[STAGES][FQ]; // frames lane + 64 q of the source row
	float rawh[STAGES][HQ]; // history samples lane + 64 q
	// Prologue, ordered so that no load waits behind one it does not depend on:
	//   level 1  slot / row of each of this wave's sources (one lane per source; no load at all when the
	//            callback's slots are a contiguous range and the rows are in order)
	//   level 2  gain + direction + previous gain (+ stream cursor), and -- needing only level 1 -- the first
	//            source's history and frames
	//   level 3  the first source's spectra (consumed after the forward FFT, so they arrive under it)
	LaneMeta lm{};
	const bool have = first + lane < last; // <= 64 sources per wave (gas_hrtf_plan): one lane per source of this wave
	if (have) {
		const uint32_t e = g.order ? g.order[first + lane] : first + lane; // direction order (k_dir_order) or entry order
		lm.slot = g.slots ? g.slots[e] : g.slot_base + e;
		lm.row = g.rows ? g.rows[e] : e;
	}
#ifdef GAS_STAMPS
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // diagnostic: slot / row lists (and everything issued before) have arrived
	GAS_STAMP(7);
#endif
	if (have) {
		// `fresh`: parameter rows published from device memory for exactly this callback's list (row order) and not
		// yet scattered: consume them here and write them through to the slot table (saves the scatter launch).
		const gas_params *P = fresh ? fresh + lm.row : st.params + lm.slot;
		if (fresh && !GAS_DEFER_WT && !(GAS_ABL & 128)) {
			const float4 *src4 = reinterpret_cast<const float4 *>(P);
			float4 *dst4 = reinterpret_cast<float4 *>(st.params + lm.slot);
#pragma unroll
			for (int k = 0; k < 8; k++) {
				dst4[k] = src4[k];
			}
		}
		const float2 gd = (GAS_ABL & 128) ? make_float2(1.0f, __uint_as_float(lm.slot & 1023u)) : *reinterpret_cast<const float2 *>(&P->hrtf_gain); // hrtf_gain, hrtf_dir: one 8-byte load
		lm.g0 = (GAS_ABL & 128) ? 0.5f : st.hrtf_prev_gain[lm.slot];
		lm.g1 = gd.x;
		const uint32_t d = __float_as_uint(gd.y);
		lm.dir = d < tab.dirs ? d : 0;
		lm.pdir = lm.dir;
		if constexpr (XFADE) {
			const uint32_t pd = st.hrtf_prev_dir[lm.slot]; // previous direction + 1, 0 = none yet
			lm.pdir = pd == 0 ? lm.dir : (pd - 1 < tab.dirs ? pd - 1 : 0);
		}
		if constexpr (SRC_PCM) {
			lm.cur = cursors[lm.slot];
		}
		// per-source state that does not depend on the audio, written once for all of the wave's sources (one lane
		// each) instead of by lane 0 inside the source loop
		if (!(GAS_ABL & 64)) {
			st.hrtf_prev_gain[lm.slot] = lm.g1;
			if constexpr (XFADE) {
				st.hrtf_prev_dir[lm.slot] = lm.dir + 1;
			}
			if constexpr (!PEAKS) { // "not measured": never passes the gate (audio_spatializer.cpp:464-469)
				*reinterpret_cast<float2 *>(g.peaks + (size_t)lm.row * 2) = make_float2(__builtin_inff(), __builtin_inff());
			}
		}
	}
	if (first < last) {
#pragma unroll
		for (int s = 0; s < STAGES; s++) {
			if (first + s < last) { // wave-uniform
				SrcMeta ms{};
				ms.slot = (uint32_t)__builtin_amdgcn_readlane((int)lm.slot, s);
				ms.row = (uint32_t)__builtin_amdgcn_readlane((int)lm.row, s);
				if (!(GAS_ABL & 2)) {
					load_history<HQ>(st.hrtf_hist + (size_t)ms.slot * HL, lane, rawh[s]);
				}
				if constexpr (!WITH_ER && !SRC_PCM) {
					load_window<false, FQ>(g, ms, lane, fade_env, raw[s]); // needs the row only
				}
			}
		}
		if constexpr (!WITH_ER && SRC_PCM) {
#pragma unroll
			for (int s = 0; s < STAGES; s++) {
				if (first + s < last) {
					load_window<true, FQ>(g, bcast_meta<SRC_PCM>(lm, s, F), lane, fade_env, raw[s]); // needs the cursor
				}
			}
		}
		if constexpr (!SKEW) {
			issue_spectra(tab.spec, (uint32_t)__builtin_amdgcn_readlane((int)lm.dir, 0), lane, hs);
		}
	}

	if (GAS_TW_LDS && !(GAS_ABL & 16)) {
		reinterpret_cast<float4 *>(tw_lds)[threadIdx.x] = tw_in;
		__syncthreads();
#pragma unroll
		for (int k = 0; k < 8; k++) {
			t1[k] = tw_lds[k * 64 + lane];
			t2[k] = tw_lds[(8 + k) * 64 + lane];
		}
	}
	GAS_STAMP(1);

	// STAGES sources per trip, each with its own landing registers: while source e transforms, the frames and
	// history of sources e+1 .. e+STAGES are in flight.
	for (uint32_t e0 = first; e0 < last; e0 += STAGES) {
#pragma unroll
		for (int s = 0; s < STAGES; s++) {
		const uint32_t e = e0 + s;
		if (e >= last) { // wave-uniform
			continue;
		}
		const bool has_next = e + 1 < last;
		const bool has_ahead = e + STAGES < last;
		const SrcMeta m = bcast_meta<SRC_PCM>(lm, e - first, F);
		const SrcMeta mn = bcast_meta<SRC_PCM>(lm, has_next ? e + 1 - first : e - first, F);
		const SrcMeta ma = bcast_meta<SRC_PCM>(lm, has_ahead ? e + STAGES - first : e - first, F);

		// x_full[lane + 64 q]: q < HQ from the history, the rest from this callback's frames.
		float xq[NQ];
#pragma unroll
		for (int q = 0; q < HQ; q++) {
			xq[q] = rawh[s][q];
		}
		if constexpr (WITH_ER) {
			// early reflections (oracle fx_early_reflections): taps in order, f32.  The 64 gathers per
			// lane run in a rolled loop and hand their result over through the wave's LDS slice.
			const gas_params *P = fresh ? fresh + m.row : st.params + m.slot; // device-published rows of this callback, not yet in the table
			const gas_audio_frame *srow = g.src + (size_t)m.row * F;
			const uint32_t er_pos = st.er_pos[m.slot];
			gas_audio_frame *ring = st.er_ring + (size_t)m.slot * er_R;
			float *xs = reinterpret_cast<float *>(lds);
#ifndef GAS_ER_PAIRS // the product form: a lane owns frames lane + 64 q, 8-byte accesses, 16 tap loads in flight (two round trips per source at F = 256)
#pragma unroll 2 // 16 tap loads in flight per trip (4 spills, 1 serialises four round trips per source)
			for (int q = 0; q < FQ; q++) {
				const int f = lane + 64 * q;
				gas_audio_frame fr = srow[f];
				ring[(er_pos + (uint32_t)f) & (er_R - 1)] = fr; // this block into the ring
				float yl = fr.left, yr = fr.right;
#pragma unroll
				for (int k = 0; k < GAS_ER_TAPS; k++) {
					const uint32_t du = P->er_delay[k];
					const int d = (int)(du < er_R - F ? du : er_R - F); // keeps every tap inside row/ring
					const float gk = P->er_gain[k];
					const int i = f - d;
					// i >= 0: still inside this callback's source row; else a previous callback's ring frame
					gas_audio_frame xp = i >= 0 ? srow[i] : ring[(er_pos + (uint32_t)(i + (int)er_R)) & (er_R - 1)];
					yl = yl + gk * xp.left;
					yr = yr + gk * xp.right;
				}
				xs[f] = (yl + yr) * 0.5f;
			}
#else
			// EXPERIMENT (round 3, -DGAS_ER_PAIRS, measured SLOWER: cfg5 27.2 vs 22.6 us per launch): a lane owns frame PAIRS
			// (2 lane, 2 lane + 1) + 128 qq, so the row, the ring store and every tap are 16-byte accesses and the 8 tap
			// loads of a pass are all in flight at once.  The gathers are 6.6 of the launch's 22.6 us (with them removed the
			// launch takes 16.0), but halving their instruction count does not pay: odd delays make the 16-byte loads
			// 8-byte-aligned only, the per-lane pointer selects and the straddle fix-ups add VALU work in front of them.
			// The results pass through xs[] by frame index, and each frame adds its taps in tap order: the same bits.  A pair straddles the row / ring boundary (first frame still in the ring, second already in
			// the row: i == -1) or the ring's wrap (ring index er_R - 1) only for odd delays, in one lane per tap: those lanes
			// fetch their second frame separately.
			typedef float er_v4f __attribute__((ext_vector_type(4)));
#pragma unroll
			for (int qq = 0; qq < FQ / 2; qq++) {
				const int f0 = 2 * lane + 128 * qq;
				const er_v4f fr2 = *reinterpret_cast<const er_v4f *>(&srow[f0]);
				*reinterpret_cast<er_v4f *>(&ring[(er_pos + (uint32_t)f0) & (er_R - 1)]) = fr2; // this block into the ring (er_pos, f0 even: no wrap inside a pair)
				er_v4f xp2[GAS_ER_TAPS];
				bool fix[GAS_ER_TAPS];
				const gas_audio_frame *second[GAS_ER_TAPS];
#pragma unroll
				for (int k = 0; k < GAS_ER_TAPS; k++) {
					const uint32_t du = P->er_delay[k];
					const int d = (int)(du < er_R - F ? du : er_R - F); // keeps every tap inside row/ring
					const int i = f0 - d; // first frame of the pair, relative to this callback's row
					const uint32_t j = (er_pos + (uint32_t)(i + (int)er_R)) & (er_R - 1); // its ring index when i < 0
					const gas_audio_frame *p0 = i >= 0 ? &srow[i] : &ring[j];
					// the pair is contiguous unless it straddles row / ring (i == -1) or the ring's end (j == er_R - 1, i < -1)
					fix[k] = i == -1 || (i < -1 && j == er_R - 1);
					second[k] = i + 1 >= 0 ? &srow[i + 1] : &ring[(j + 1) & (er_R - 1)];
					xp2[k] = *reinterpret_cast<const er_v4f *>(fix[k] ? reinterpret_cast<const gas_audio_frame *>(&srow[f0]) : p0); // (a straddling lane loads something harmless here)
				}
				float yl0 = fr2.x, yr0 = fr2.y, yl1 = fr2.z, yr1 = fr2.w;
#pragma unroll
				for (int k = 0; k < GAS_ER_TAPS; k++) {
					const float gk = P->er_gain[k];
					er_v4f x2 = xp2[k];
					if (fix[k]) { // one lane per tap at most (odd delays): both frames by 8-byte loads
						const uint32_t du = P->er_delay[k];
						const int d = (int)(du < er_R - F ? du : er_R - F);
						const int i = f0 - d;
						const uint32_t j = (er_pos + (uint32_t)(i + (int)er_R)) & (er_R - 1);
						const gas_audio_frame a = i >= 0 ? srow[i] : ring[j];
						const gas_audio_frame b2 = *second[k];
						x2 = er_v4f{ a.left, a.right, b2.left, b2.right };
					}
					yl0 = yl0 + gk * x2.x;
					yr0 = yr0 + gk * x2.y;
					yl1 = yl1 + gk * x2.z;
					yr1 = yr1 + gk * x2.w;
				}
				*reinterpret_cast<float2 *>(&xs[f0]) = make_float2((yl0 + yr0) * 0.5f, (yl1 + yr1) * 0.5f);
			}
#endif
			if (lane == 0) {
				st.er_pos[m.slot] = (er_pos + F) & (er_R - 1);
			}
			wave_lds_sync();
#pragma unroll
			for (int q = 0; q < FQ; q++) {
				const int f = lane + 64 * q;
				const float t = (float)f * (1.0f / (float)F);
				xq[HQ + q] = xs[f] * (m.g1 * t + (1 - t) * m.g0);
			}
			wave_lds_sync();
		} else {
#pragma unroll
			for (int q = 0; q < FQ; q++) {
				const int f = lane + 64 * q;
				const float mono = (raw[s][q].left + raw[s][q].right) * 0.5f;
				const float t = (float)f * (1.0f / (float)F); // exact for F = 128/256/512
				xq[HQ + q] = mono * (m.g1 * t + (1 - t) * m.g0);
			}
		}
#ifdef GAS_STAMPS
		if (e == first) {
			GAS_STAMP(2); // the first source's frames, history and parameters have arrived
		}
#endif
		// new history = x_full[F .. F + HL)
		if (!(GAS_ABL & 2)) {
			store_history<HQ>(st.hrtf_hist + (size_t)m.slot * HL, lane, &xq[FQ]);
		}
		if (lane == 0 && !(GAS_ABL & 64)) {
			if constexpr (SRC_PCM) {
				if (m.hf) { // advance the playback cursor (audio_spatializer.cpp:378,398)
					cursors[m.slot].pos = m.pos + m.mixed;
					if (m.mixed != F) {
						cursors[m.slot].has_frames = 0;
					}
				}
			}
		}
		// this stage's landing registers are free again: start the frames and history of source e + STAGES
		if (has_ahead) {
			if (!(GAS_ABL & 2)) {
				load_history<HQ>(st.hrtf_hist + (size_t)ma.slot * HL, lane, rawh[s]);
			}
			if constexpr (!WITH_ER) {
				load_window<SRC_PCM, FQ>(g, ma, lane, fade_env, raw[s]);
			}
		}

		// z = a + i b : a = x_full[0..512), b = x_full[S..S+512)
		const bool changed = XFADE && m.pdir != m.dir; // wave-uniform
		if constexpr (PEAKS) {
			float2 v[8];
#pragma unroll
			for (int j = 0; j < 8; j++) {
				v[j] = make_float2(xq[j], xq[j + SQ]);
			}
			fft512<false>(v, t1, t2, lds, lane);
			finish_spectra(lane, hs);
			float2 yl[8], yr[8];
#pragma unroll
			for (int j = 0; j < 8; j++) {
				yl[j] = cmul(v[j], make_float2(hs[j].x, hs[j].y));
				yr[j] = cmul(v[j], make_float2(hs[j].z, hs[j].w));
			}
			// spectra registers are free again: the old direction's rows (cross-fade) or the next source's
			if (changed) {
				issue_spectra(tab.spec, m.pdir, lane, hs);
			} else if (has_next) {
				issue_spectra(tab.spec, mn.dir, lane, hs);
			}
			float pkl = 0.0f, pkr = 0.0f;
			fft512_pair<true>(yl, yr, t1, t2, lds, lds + LDS_F2_HALF, lane);
			// valid outputs are window positions [512 - S, 512): registers j >= HQ
			float oL[FQ], oR[FQ];
#pragma unroll
			for (int t = 0; t < SQ; t++) {
				oL[t] = yl[HQ + t].x;
				oL[SQ + t] = yl[HQ + t].y;
				oR[t] = yr[HQ + t].x;
				oR[SQ + t] = yr[HQ + t].y;
			}
			if constexpr (XFADE) {
				if (changed) { // render the old direction too and blend, t = frame / F
					finish_spectra(lane, hs);
#pragma unroll
					for (int j = 0; j < 8; j++) {
						yl[j] = cmul(v[j], make_float2(hs[j].x, hs[j].y));
						yr[j] = cmul(v[j], make_float2(hs[j].z, hs[j].w));
					}
					if (has_next) {
						issue_spectra(tab.spec, mn.dir, lane, hs);
					}
					fft512_pair<true>(yl, yr, t1, t2, lds, lds + LDS_F2_HALF, lane);
#pragma unroll
					for (int t = 0; t < SQ; t++) {
						const float ta = (float)(lane + 64 * t) * (1.0f / (float)F);
						const float tb = (float)(lane + 64 * (SQ + t)) * (1.0f / (float)F);
						oL[t] = oL[t] * ta + yl[HQ + t].x * (1 - ta);
						oL[SQ + t] = oL[SQ + t] * tb + yl[HQ + t].y * (1 - tb);
						oR[t] = oR[t] * ta + yr[HQ + t].x * (1 - ta);
						oR[SQ + t] = oR[SQ + t] * tb + yr[HQ + t].y * (1 - tb);
					}
				}
			}
			if (rows_out) { // a stage of a general effect chain: per-source rows out, dense by group entry; no sum, no peak
#pragma unroll
				for (int t = 0; t < FQ; t++) {
					rows_out[(size_t)e * F + lane + 64 * t] = gas_audio_frame{ oL[t], oR[t] };
				}
			} else {
#pragma unroll
				for (int t = 0; t < FQ; t++) {
					accL[t] += oL[t];
					accR[t] += oR[t];
					pkl = fmaxf(pkl, fabsf(oL[t]));
					pkr = fmaxf(pkr, fabsf(oR[t]));
				}
				pkl = wave_max(pkl);
				pkr = wave_max(pkr);
				if (lane == 0) {
					g.peaks[(size_t)m.row * 2] = pkl;
					g.peaks[(size_t)m.row * 2 + 1] = pkr;
				}
			}
		} else {
#pragma unroll
			for (int j = 0; j < 8; j++) {
				zs[j].x += xq[j];
				zs[j].y += xq[j + SQ];
			}
			// the run ends with the wave's sources or when the direction changes; the cross-fade pairs every source
			// with its own previous direction, so it transforms source by source
			const bool flush = !RUNS || XFADE || !has_next || mn.dir != m.dir; // wave-uniform; constant without RUNS
			if (flush) {
				if (!(GAS_ABL & 4)) {
					fft512<false>(zs, t1, t2, lds, lane);
				}
				if constexpr (SKEW) {
					if (have_prev) {
						finish_spectra(lane, hs);
#pragma unroll
						for (int j = 0; j < 8; j++) {
							aYL[j] = cadd(aYL[j], cmul(zp[j], make_float2(hs[j].x, hs[j].y)));
							aYR[j] = cadd(aYR[j], cmul(zp[j], make_float2(hs[j].z, hs[j].w)));
						}
					}
					issue_spectra(tab.spec, m.dir, lane, hs); // this run's row: consumed after the next transform
#pragma unroll
					for (int j = 0; j < 8; j++) {
						zp[j] = zs[j];
						zs[j] = make_float2(0.0f, 0.0f);
					}
					have_prev = true;
				} else {
				finish_spectra(lane, hs);
#pragma unroll
				for (int j = 0; j < 8; j++) {
					const float2 pl = cmul(zs[j], make_float2(hs[j].x, hs[j].y));
					const float2 pr = cmul(zs[j], make_float2(hs[j].z, hs[j].w));
					aYL[j] = cadd(aYL[j], pl);
					aYR[j] = cadd(aYR[j], pr);
					if constexpr (XFADE) {
						if (!changed) { // same HRIR on both sides of the fade
							bYL[j] = cadd(bYL[j], pl);
							bYR[j] = cadd(bYR[j], pr);
						}
					}
				}
				if constexpr (XFADE) {
					if (changed) {
						issue_spectra(tab.spec, m.pdir, lane, hs);
						finish_spectra(lane, hs);
#pragma unroll
						for (int j = 0; j < 8; j++) {
							bYL[j] = cadd(bYL[j], cmul(zs[j], make_float2(hs[j].x, hs[j].y)));
							bYR[j] = cadd(bYR[j], cmul(zs[j], make_float2(hs[j].z, hs[j].w)));
						}
					}
				}
#pragma unroll
				for (int j = 0; j < 8; j++) {
					zs[j] = make_float2(0.0f, 0.0f);
				}
				if (has_next) {
					issue_spectra(tab.spec, mn.dir, lane, hs);
				}
				}
			}
		}
		}
	}

	GAS_STAMP(3);
	if (GAS_DEFER_WT && have && fresh && !(GAS_ABL & 128)) {
		// the device-published rows this launch consumed go through to the slot table here, off the path to the first
		// transform (128 bytes per source, one lane each: 64 different cache lines per instruction)
		const float4 *src4 = reinterpret_cast<const float4 *>(fresh + lm.row);
		float4 *dst4 = reinterpret_cast<float4 *>(st.params + lm.slot);
#pragma unroll
		for (int k = 0; k < 8; k++) {
			dst4[k] = src4[k];
		}
	}
	if constexpr (SKEW) {
		if (have_prev) { // the last run's products
			finish_spectra(lane, hs);
#pragma unroll
			for (int j = 0; j < 8; j++) {
				aYL[j] = cadd(aYL[j], cmul(zp[j], make_float2(hs[j].x, hs[j].y)));
				aYR[j] = cadd(aYR[j], cmul(zp[j], make_float2(hs[j].z, hs[j].w)));
			}
		}
	}
	if (PEAKS && rows_out) {
		return; // rows-out stage: nothing to sum here (wave-uniform for the whole launch)
	}
	if constexpr (PEAKS) {
		// waves -> one partial mix per workgroup; each wave parks its sum in its own LDS slice
		wave_lds_sync();
		float *red = reinterpret_cast<float *>(lds);
#pragma unroll
		for (int t = 0; t < FQ; t++) {
			*reinterpret_cast<float2 *>(red + (lane + 64 * t) * 2) = make_float2(accL[t], accR[t]);
		}
		if constexpr (JOB_OK) {
			if (job_mine) {
				if (JOB_LATE) {
					job_issue(job, job_col, lane, jr);
				}
				job_finish(job, job_col, lane, jr);
			}
		}
		__syncthreads();
		const float *red_all = reinterpret_cast<const float *>(lds_all);
		for (int idx = threadIdx.x; idx < (int)(F * 2); idx += WAVES * 64) {
			float s = 0.0f;
#pragma unroll
			for (int w = 0; w < WAVES; w++) {
				s += red_all[w * LDS_F2_PER_WAVE * 2 + idx];
			}
			my_partial[idx] = s;
		}
	} else {
		// spectra of all waves -> fd[wave][ear][j][lane]; then wave 0 transforms the left ear's sum and
		// wave 1 the right ear's, and the workgroup stores one interleaved time-domain partial.  With XFADE the
		// same round runs twice: new-direction sums weighted by t = f/F, old-direction sums by 1 - t.
		if (GAS_ABL & 32) {
#pragma unroll
			for (int j = 0; j < 8; j++) {
				if (wave == 0 && aYL[j].x + aYR[j].y == 12345.0f) {
					my_partial[lane] = aYL[j].x;
				}
			}
			return;
		}
		float2 *fd = lds_all;
		float *outp = reinterpret_cast<float *>(lds_all + FD_F2 + 2 * LDS_F2_HALF);
		constexpr int ROUNDS = XFADE ? 2 : 1;
#pragma unroll
		for (int round = 0; round < ROUNDS; round++) {
			__syncthreads(); // every wave is done with its exchange slice / the previous round (fd aliases them)
#pragma unroll
			for (int j = 0; j < 8; j++) {
				fd[(wave * 2 + 0) * 512 + j * 64 + lane] = round == 0 ? aYL[j] : bYL[j];
				fd[(wave * 2 + 1) * 512 + j * 64 + lane] = round == 0 ? aYR[j] : bYR[j];
			}
			__syncthreads();
			if (round == 0) {
				GAS_STAMP(4);
			}
			if (!GAS_EPI_DIRECT) {
				// every wave folds one eighth of the bins over the WAVES spectra (fixed order), into slot 0
				constexpr int PER_THREAD = 2 * 512 / (WAVES * 64); // 2 ears x 512 bins over the workgroup
#pragma unroll
				for (int r = 0; r < PER_THREAD; r++) {
					const int idx = threadIdx.x + r * WAVES * 64; // ear * 512 + bin
					float2 sacc = fd[idx];
#pragma unroll
					for (int w = 1; w < WAVES; w++) {
						sacc = cadd(sacc, fd[w * 2 * 512 + idx]);
					}
					fd[idx] = sacc; // only this thread touches column idx: no hazard
				}
				__syncthreads();
			}
			if constexpr (JOB_OK) {
				if (job_mine && round == 0) { // waves 2.. idle while 0 and 1 transform: the previous callback's sum
					if (JOB_LATE) {
						job_issue(job, job_col, lane, jr);
					}
					job_finish(job, job_col, lane, jr);
				}
			}
			if (wave < 2) {
				float2 y[8];
				if (GAS_EPI_DIRECT) {
					// wave `ear` adds the WAVES spectra of its ear itself, in wave order (the same association as the
					// fold above: ((w0 + w1) + w2) + ...), 64 conflict-free 8-byte reads per lane
#pragma unroll
					for (int j = 0; j < 8; j++) {
						y[j] = fd[wave * 512 + j * 64 + lane];
					}
#pragma unroll
					for (int w = 1; w < WAVES; w++) {
#pragma unroll
						for (int j = 0; j < 8; j++) {
							y[j] = cadd(y[j], fd[(w * 2 + wave) * 512 + j * 64 + lane]);
						}
					}
				} else {
#pragma unroll
					for (int j = 0; j < 8; j++) {
						y[j] = fd[wave * 512 + j * 64 + lane];
					}
				}
				fft512<true>(y, t1, t2, lds_all + FD_F2 + wave * LDS_F2_HALF, lane);
#pragma unroll
				for (int t = 0; t < SQ; t++) {
					const int fa = lane + 64 * t, fb = lane + 64 * (SQ + t);
					if constexpr (!XFADE) {
						outp[fa * 2 + wave] = y[HQ + t].x;
						outp[fb * 2 + wave] = y[HQ + t].y;
					} else {
						const float ta = (float)fa * (1.0f / (float)F), tb = (float)fb * (1.0f / (float)F);
						if (round == 0) {
							outp[fa * 2 + wave] = y[HQ + t].x * ta;
							outp[fb * 2 + wave] = y[HQ + t].y * tb;
						} else { // the same lane owns the same elements in both rounds
							outp[fa * 2 + wave] += y[HQ + t].x * (1 - ta);
							outp[fb * 2 + wave] += y[HQ + t].y * (1 - tb);
						}
					}
				}
			}
		}
		__syncthreads();
		for (int idx = threadIdx.x; idx < (int)(F * 2); idx += WAVES * 64) {
			my_partial[idx] = outp[idx];
		}
		GAS_STAMP(5);
	}
	(void)accL;
	(void)accR;
	(void)aYL;
	(void)aYR;
	(void)bYL;
	(void)bYR;
	(void)zs;
	(void)zp;
	(void)have_prev;
	(void)jr;
	(void)job_mine;
}

// One launch per callback for every HRTF source: workgroups [0, wgs_fd) run the frequency-domain body over
// g_fd, the rest run the exact-peak body over g_pk (the draining playbacks).
template <int SQ, bool WITH_ER, bool SRC_PCM, bool XFADE, bool RUNS>
__global__ __launch_bounds__(WAVES * 64, GAS_HRTF_WAVES_PER_SIMD) void k_hrtf_ols(gas_group_args g_fd, gas_group_args g_pk, uint32_t wgs_fd, gas_dev_state st, gas_hrtf_table tab, const float2 *__restrict__ tw, uint32_t er_R, float *__restrict__ partials, uint32_t p_offset, gas_cursor *__restrict__ cursors, const float *__restrict__ fade_env, const gas_params *__restrict__ fresh, gas_deferred_reduce job) {
	constexpr int LDS_F2 = HrtfLds<SQ, false>::TOTAL_F2 > HrtfLds<SQ, true>::TOTAL_F2 ? HrtfLds<SQ, false>::TOTAL_F2 : HrtfLds<SQ, true>::TOTAL_F2;
	__shared__ float2 lds_all[LDS_F2];
	__shared__ float2 tw_lds[GAS_TW_LDS ? 1024 : 1];
	if ((GAS_ABL & 256) && wgs_fd != 0xffffffffu) {
		return;
	}
	float *my_partial = partials + ((size_t)p_offset + blockIdx.x) * (size_t)(2 * SQ * 64 * 2);
	if (blockIdx.x < wgs_fd) {
		hrtf_body<SQ, WITH_ER, false, SRC_PCM, XFADE, RUNS>(lds_all, tw_lds, blockIdx.x, g_fd, st, tab, tw, wgs_fd, er_R, my_partial, cursors, fade_env, fresh, nullptr, job, blockIdx.x);
	} else {
		hrtf_body<SQ, WITH_ER, true, SRC_PCM, XFADE>(lds_all, tw_lds, blockIdx.x - wgs_fd, g_pk, st, tab, tw, gridDim.x - wgs_fd, er_R, my_partial, cursors, fade_env, fresh, nullptr, job, blockIdx.x);
	}
}

// One HRTF stage of a general effect chain: stereo rows in (mono downmix inside), per-source stereo rows out.
template <int SQ, bool XFADE>
__global__ __launch_bounds__(WAVES * 64, GAS_HRTF_WAVES_PER_SIMD) void k_hrtf_rows(gas_group_args g, gas_dev_state st, gas_hrtf_table tab, const float2 *__restrict__ tw, gas_audio_frame *__restrict__ rows_out) {
	__shared__ float2 lds_all[HrtfLds<SQ, true>::TOTAL_F2];
	__shared__ float2 tw_lds[GAS_TW_LDS ? 1024 : 1];
	hrtf_body<SQ, false, true, false, XFADE>(lds_all, tw_lds, blockIdx.x, g, st, tab, tw, gridDim.x, 0, nullptr, nullptr, nullptr, nullptr, rows_out);
}

// The last stage of a general chain left per-source rows: add them into this workgroup's partial mix and take each
// source's peak (audio_spatializer.cpp:449-461).  Wave = source, lane = frame, like k_er_only without the taps.
template <int FQ>
__global__ __launch_bounds__(WAVES * 64) void k_rows_accumulate(gas_group_args g, float *__restrict__ partials, uint32_t p_offset) {
	constexpr uint32_t F = FQ * 64;
	__shared__ float red_all[WAVES * F * 2];
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	float accL[FQ], accR[FQ];
#pragma unroll
	for (int t = 0; t < FQ; t++) {
		accL[t] = 0.0f;
		accR[t] = 0.0f;
	}
	uint32_t first, last;
	wave_range(g.n, blockIdx.x * WAVES + wave, gridDim.x * WAVES, first, last);
	for (uint32_t e = first; e < last; e++) {
		const uint32_t row = g.rows ? g.rows[e] : e; // where this source's peak goes
		float pkl = 0.0f, pkr = 0.0f;
#pragma unroll
		for (int q = 0; q < FQ; q++) {
			const gas_audio_frame fr = g.src[(size_t)e * F + lane + 64 * q]; // dense rows of the previous stage
			accL[q] += fr.left;
			accR[q] += fr.right;
			pkl = fmaxf(pkl, fabsf(fr.left));
			pkr = fmaxf(pkr, fabsf(fr.right));
		}
		pkl = wave_max(pkl);
		pkr = wave_max(pkr);
		if (lane == 0) {
			g.peaks[(size_t)row * 2] = pkl;
			g.peaks[(size_t)row * 2 + 1] = pkr;
		}
	}
	float *red = red_all + wave * F * 2;
#pragma unroll
	for (int t = 0; t < FQ; t++) {
		*reinterpret_cast<float2 *>(red + (lane + 64 * t) * 2) = make_float2(accL[t], accR[t]);
	}
	__syncthreads();
	float *my_partial = partials + ((size_t)p_offset + blockIdx.x) * (F * 2);
	for (int idx = threadIdx.x; idx < (int)(F * 2); idx += WAVES * 64) {
		float sacc = 0.0f;
#pragma unroll
		for (int w = 0; w < WAVES; w++) {
			sacc += red_all[w * F * 2 + idx];
		}
		my_partial[idx] = sacc;
	}
}

// The same for several output buses (effect kinds): every source's rows are added into each bus with
// that source's weight there -- 1 on its dry bus, its send volume on its send bus (audio_spatializer.cpp:274-324 gives
// the volumes, AudioServer the per-bus multiply-accumulate); the peak is that of the rows, before any bus factor.
// Partial rows of bus b: [b * bus_rows + p_offset + workgroup].  Buses beyond n_buses cost nothing (their weights are 0
// and their stores are skipped).
template <int FQ>
__global__ __launch_bounds__(WAVES * 64) void k_rows_accumulate_buses(gas_group_args g, const gas_bus_route *__restrict__ routes, uint32_t n_buses, uint32_t bus_rows, float *__restrict__ partials, uint32_t p_offset) {
	constexpr uint32_t F = FQ * 64;
	constexpr int NB = GAS_MAX_BUSES;
	__shared__ float red_all[WAVES * F * 2];
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	float accL[NB][FQ], accR[NB][FQ];
#pragma unroll
	for (int b = 0; b < NB; b++) {
#pragma unroll
		for (int t = 0; t < FQ; t++) {
			accL[b][t] = 0.0f;
			accR[b][t] = 0.0f;
		}
	}
	uint32_t first, last;
	wave_range(g.n, blockIdx.x * WAVES + wave, gridDim.x * WAVES, first, last);
	for (uint32_t e = first; e < last; e++) {
		const uint32_t row = g.rows ? g.rows[e] : e; // where this source's peak goes
		const gas_bus_route r = routes[g.slots[e]]; // wave-uniform
		gas_audio_frame fr[FQ];
		float pkl = 0.0f, pkr = 0.0f;
#pragma unroll
		for (int q = 0; q < FQ; q++) {
			fr[q] = g.src[(size_t)e * F + lane + 64 * q]; // dense rows of the previous stage
			pkl = fmaxf(pkl, fabsf(fr[q].left));
			pkr = fmaxf(pkr, fabsf(fr[q].right));
		}
#pragma unroll
		for (int b = 0; b < NB; b++) {
			const float wl = gas_bus_weight(r, (uint32_t)b, 0, 0);
			const float wr = gas_bus_weight(r, (uint32_t)b, 0, 1);
			if ((uint32_t)b < n_buses && (wl != 0.0f || wr != 0.0f)) { // wave-uniform
#pragma unroll
				for (int q = 0; q < FQ; q++) {
					accL[b][q] += fr[q].left * wl;
					accR[b][q] += fr[q].right * wr;
				}
			}
		}
		pkl = wave_max(pkl);
		pkr = wave_max(pkr);
		if (lane == 0) {
			g.peaks[(size_t)row * 2] = pkl;
			g.peaks[(size_t)row * 2 + 1] = pkr;
		}
	}
	float *red = red_all + wave * F * 2;
#pragma unroll
	for (int b = 0; b < NB; b++) {
		if ((uint32_t)b < n_buses) { // uniform over the launch
#pragma unroll
			for (int t = 0; t < FQ; t++) {
				*reinterpret_cast<float2 *>(red + (lane + 64 * t) * 2) = make_float2(accL[b][t], accR[b][t]);
			}
			__syncthreads();
			float *my_partial = partials + ((size_t)b * bus_rows + p_offset + blockIdx.x) * (F * 2);
			for (int idx = threadIdx.x; idx < (int)(F * 2); idx += WAVES * 64) {
				float sacc = 0.0f;
#pragma unroll
				for (int w = 0; w < WAVES; w++) {
					sacc += red_all[w * F * 2 + idx];
				}
				my_partial[idx] = sacc;
			}
			__syncthreads();
		}
	}
}

// HRIR [dirs][2][taps] -> lane-major half-spectra table (see issue_spectra), one wave per (direction, ear),
// scaled by 1/512 so the inverse transform needs no normalisation.
__global__ __launch_bounds__(64) void k_hrtf_table(const float *__restrict__ hrir, uint32_t dirs, uint32_t taps, const float2 *__restrict__ tw, float4 *__restrict__ spec) {
	__shared__ float2 lds[LDS_F2_HALF];
	const int lane = threadIdx.x;
	const uint32_t dir = blockIdx.x >> 1, ear = blockIdx.x & 1;
	if (dir >= dirs) {
		return;
	}
	float2 t1[8], t2[8];
#pragma unroll
	for (int k = 0; k < 8; k++) {
		t1[k] = tw[k * 64 + lane];
		t2[k] = tw[(8 + k) * 64 + lane];
	}
	const float *h = hrir + ((size_t)dir * 2 + ear) * taps;
	float2 v[8];
#pragma unroll
	for (int j = 0; j < 8; j++) {
		const uint32_t n = lane + 64 * j;
		v[j] = make_float2(n < taps ? h[n] : 0.0f, 0.0f);
	}
	fft512<false>(v, t1, t2, lds, lane);
	float *out = reinterpret_cast<float *>(spec);
#pragma unroll
	for (int j = 0; j < 4; j++) { // bins 0..255 only (Hermitian half)
		const size_t o = (((size_t)dir * 4 + j) * 64 + lane) * 4 + ear * 2;
		out[o] = v[j].x * (1.0f / 512.0f);
		out[o + 1] = v[j].y * (1.0f / 512.0f);
	}
	if (lane == 0) { // Nyquist bin 256 (lane 0, j = 4) is real: park it in DC's imaginary slot
		out[((size_t)dir * 4 * 64) * 4 + ear * 2 + 1] = v[4].x * (1.0f / 512.0f);
	}
}

// The chain [EARLY_REFLECTIONS] alone: stereo out, lane = frame, wave = source.
template <int FQ>
__global__ __launch_bounds__(WAVES * 64) void k_er_only(gas_group_args g, gas_dev_state st, uint32_t er_R, float *__restrict__ partials, uint32_t p_offset, gas_audio_frame *__restrict__ rows_out) {
	constexpr uint32_t F = FQ * 64;
	__shared__ float red_all[WAVES * F * 2];
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	float accL[FQ], accR[FQ];
#pragma unroll
	for (int t = 0; t < FQ; t++) {
		accL[t] = 0.0f;
		accR[t] = 0.0f;
	}
	uint32_t first, last;
	wave_range(g.n, blockIdx.x * WAVES + wave, gridDim.x * WAVES, first, last);
	for (uint32_t e = first; e < last; e++) {
		const uint32_t slot = g.slots[e];
		const uint32_t row = g.rows ? g.rows[e] : e;
		const gas_params *P = st.params + slot;
		const gas_audio_frame *srow = g.src + (size_t)row * F;
		const uint32_t er_pos = st.er_pos[slot];
		gas_audio_frame *ring = st.er_ring + (size_t)slot * er_R;
		float pkl = 0.0f, pkr = 0.0f;
#pragma unroll
		for (int q = 0; q < FQ; q++) {
			const int f = lane + 64 * q;
			gas_audio_frame fr = srow[f];
			ring[(er_pos + (uint32_t)f) & (er_R - 1)] = fr;
			float yl = fr.left, yr = fr.right;
#pragma unroll
			for (int k = 0; k < GAS_ER_TAPS; k++) {
				const uint32_t du = P->er_delay[k];
				const int d = (int)(du < er_R - F ? du : er_R - F);
				const float gk = P->er_gain[k];
				const int i = f - d;
				gas_audio_frame xp = i >= 0 ? srow[i] : ring[(er_pos + (uint32_t)(i + (int)er_R)) & (er_R - 1)];
				yl = yl + gk * xp.left;
				yr = yr + gk * xp.right;
			}
			if (rows_out) { // a stage of a general effect chain: dense per-source rows, no sum, no peak
				rows_out[(size_t)e * F + f] = gas_audio_frame{ yl, yr };
			} else {
				accL[q] += yl;
				accR[q] += yr;
				pkl = fmaxf(pkl, fabsf(yl));
				pkr = fmaxf(pkr, fabsf(yr));
			}
		}
		pkl = wave_max(pkl);
		pkr = wave_max(pkr);
		if (lane == 0) {
			st.er_pos[slot] = (er_pos + F) & (er_R - 1);
			if (!rows_out) {
				g.peaks[(size_t)row * 2] = pkl;
				g.peaks[(size_t)row * 2 + 1] = pkr;
			}
		}
	}
	if (rows_out) {
		return; // wave-uniform for the whole launch
	}
	float *red = red_all + wave * F * 2;
#pragma unroll
	for (int t = 0; t < FQ; t++) {
		*reinterpret_cast<float2 *>(red + (lane + 64 * t) * 2) = make_float2(accL[t], accR[t]);
	}
	__syncthreads();
	float *my_partial = partials + ((size_t)p_offset + blockIdx.x) * (F * 2);
	for (int idx = threadIdx.x; idx < (int)(F * 2); idx += WAVES * 64) {
		float s = 0.0f;
#pragma unroll
		for (int w = 0; w < WAVES; w++) {
			s += red_all[w * F * 2 + idx];
		}
		my_partial[idx] = s;
	}
}

} // namespace

// Twiddles per lane: [lane][0..7] = W64^(n1 k0), [lane][8..15] = W512^(n0 (k0' + 8 k1)), computed in f64.
void gas_make_twiddles(float2 *host_tw) {
	const double two_pi = 6.2831853071795864769252867666;
	for (int l = 0; l < 64; l++) {
		const int hi = l >> 3, lo = l & 7;
		for (int k = 0; k < 8; k++) {
			double a1 = -two_pi * (double)(hi * k) / 64.0;
			host_tw[k * 64 + l] = make_float2((float)cos(a1), (float)sin(a1)); // [k][lane]: a wave's load of one k is one 512-byte run
			double a2 = -two_pi * (double)(lo * (hi + 8 * k)) / 512.0;
			host_tw[(8 + k) * 64 + l] = make_float2((float)cos(a2), (float)sin(a2));
		}
	}
}

// Launch plan of one k_hrtf_ols launch.  The kernel is resident at GAS_HRTF_WAVES_PER_SIMD waves/SIMD, i.e.
// 256 * 4 * that / WAVES workgroups at a time; the frequency-domain and exact-peak workgroups share that
// budget (an exact-peak source costs about twice a frequency-domain one) so the whole grid drains in one
// even round.  At most 64 sources per wave (one metadata lane per source); beyond that the grid grows.
#ifndef GAS_PK_WEIGHT
#define GAS_PK_WEIGHT 2
#endif
void gas_hrtf_plan(uint32_t n_fd, uint32_t n_pk, gas_hrtf_launch_plan *p) {
	// One residency round: 256 CUs x 1 workgroup of WAVES waves.  A group gets as many workgroups as it has
	// sources / WAVES, capped by its share of the round (exact-peak sources cost two more FFTs each: weight 2) and
	// raised beyond the round only when a wave would otherwise own more than 64 sources.  Sources are split evenly
	// over the group's waves (wave_range).
	const uint32_t budget = 256u * 4u * GAS_HRTF_WAVES_PER_SIMD / WAVES;
	auto wgs_for = [](uint32_t n, uint32_t cap) {
		const uint32_t want = (n + WAVES - 1) / WAVES; // one source per wave at least
		const uint32_t need = (n + WAVES * 64 - 1) / (WAVES * 64); // 64 sources per wave at most
		const uint32_t w = want < cap ? want : cap;
		return w > need ? w : need;
	};
	p->wgs_fd = p->wgs_pk = 0;
	uint32_t budget_fd = budget;
	if (n_pk) {
		uint32_t share = n_fd ? (uint32_t)(((uint64_t)budget * GAS_PK_WEIGHT * n_pk + (n_fd + (uint64_t)GAS_PK_WEIGHT * n_pk) - 1) / (n_fd + (uint64_t)GAS_PK_WEIGHT * n_pk)) : budget;
		share = share < 1 ? 1 : (share > budget - (n_fd ? 1 : 0) ? budget - (n_fd ? 1 : 0) : share);
		p->wgs_pk = wgs_for(n_pk, share);
		budget_fd = budget > p->wgs_pk ? budget - p->wgs_pk : 1;
	}
	if (n_fd) {
		p->wgs_fd = wgs_for(n_fd, budget_fd);
	}
}

uint32_t gas_hrtf_partials(uint32_t n) {
	gas_hrtf_launch_plan p;
	gas_hrtf_plan(n, 0, &p);
	return p.wgs_fd;
}

hipError_t gas_launch_hrtf_ols(hipStream_t stream, bool with_er, bool crossfade, bool runs, const gas_group_args &g_fd, const gas_group_args &g_pk, const gas_dev_state &st, const gas_hrtf_table &tab, const float2 *twiddles, uint32_t frames, uint32_t hist_len, uint32_t er_ring_frames, float *partials, uint32_t p_offset, gas_cursor *cursors, const float *fade_env, const gas_params *fresh, const gas_deferred_reduce &job) {
	if (g_fd.n + g_pk.n == 0) {
		return hipSuccess;
	}
	if (frames % 128 != 0 || frames > 512 || hist_len != 512 - frames / 2 || (with_er && cursors)) {
		return hipErrorInvalidValue;
	}
	gas_hrtf_launch_plan plan;
	gas_hrtf_plan(g_fd.n, g_pk.n, &plan);
	const uint32_t wgs_fd = plan.wgs_fd;
	dim3 grid(plan.wgs_fd + plan.wgs_pk), block(WAVES * 64);
#define GAS_HRTF_LAUNCH2(SQv, ERv, PCMv, XFv, RNv) \
	hipLaunchKernelGGL((k_hrtf_ols<SQv, ERv, PCMv, XFv, RNv>), grid, block, 0, stream, g_fd, g_pk, wgs_fd, st, tab, twiddles, er_ring_frames, partials, p_offset, cursors, fade_env, fresh, job)
#define GAS_HRTF_LAUNCH(SQv, ERv, PCMv, XFv)          \
	if (runs && !XFv) {                               \
		GAS_HRTF_LAUNCH2(SQv, ERv, PCMv, false, true); \
	} else {                                          \
		GAS_HRTF_LAUNCH2(SQv, ERv, PCMv, XFv, false);  \
	}
#define GAS_HRTF_CASE3(SQv, XFv)                      \
	if (with_er) {                                    \
		GAS_HRTF_LAUNCH(SQv, true, false, XFv);       \
	} else if (cursors) {                             \
		GAS_HRTF_LAUNCH(SQv, false, true, XFv);       \
	} else {                                          \
		GAS_HRTF_LAUNCH(SQv, false, false, XFv);      \
	}
#define GAS_HRTF_CASE(SQv)              \
	case SQv:                           \
		if (crossfade) {                \
			GAS_HRTF_CASE3(SQv, true)   \
		} else {                        \
			GAS_HRTF_CASE3(SQv, false)  \
		}                               \
		break;
	switch (frames / 128) {
		GAS_HRTF_CASE(1)
		GAS_HRTF_CASE(2)
		GAS_HRTF_CASE(3)
		GAS_HRTF_CASE(4)
		default:
			return hipErrorInvalidValue;
	}
#undef GAS_HRTF_CASE
#undef GAS_HRTF_CASE3
#undef GAS_HRTF_LAUNCH
#undef GAS_HRTF_LAUNCH2
	return hipGetLastError();
}

hipError_t gas_launch_er_only(hipStream_t stream, const gas_group_args &g, const gas_dev_state &st, uint32_t frames, uint32_t er_ring_frames, float *partials, uint32_t p_offset, uint32_t p_stride, gas_audio_frame *rows_out) {
	(void)p_stride;
	if (g.n == 0) {
		return hipSuccess;
	}
	const uint32_t wgs = gas_hrtf_partials(g.n);
	dim3 grid(wgs), block(WAVES * 64);
	switch (frames / 64) {
		case 2:
			hipLaunchKernelGGL((k_er_only<2>), grid, block, 0, stream, g, st, er_ring_frames, partials, p_offset, rows_out);
			break;
		case 4:
			hipLaunchKernelGGL((k_er_only<4>), grid, block, 0, stream, g, st, er_ring_frames, partials, p_offset, rows_out);
			break;
		case 6:
			hipLaunchKernelGGL((k_er_only<6>), grid, block, 0, stream, g, st, er_ring_frames, partials, p_offset, rows_out);
			break;
		case 8:
			hipLaunchKernelGGL((k_er_only<8>), grid, block, 0, stream, g, st, er_ring_frames, partials, p_offset, rows_out);
			break;
		default:
			return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t gas_launch_hrtf_rows(hipStream_t stream, bool crossfade, const gas_group_args &g, const gas_dev_state &st, const gas_hrtf_table &tab, const float2 *twiddles, uint32_t frames, gas_audio_frame *rows_out) {
	if (g.n == 0) {
		return hipSuccess;
	}
	const uint32_t wgs = gas_hrtf_partials(g.n);
	dim3 grid(wgs), block(WAVES * 64);
#define GAS_ROWS_CASE(SQv)                                                                                              \
	case SQv:                                                                                                           \
		if (crossfade) {                                                                                                \
			hipLaunchKernelGGL((k_hrtf_rows<SQv, true>), grid, block, 0, stream, g, st, tab, twiddles, rows_out);  \
		} else {                                                                                                        \
			hipLaunchKernelGGL((k_hrtf_rows<SQv, false>), grid, block, 0, stream, g, st, tab, twiddles, rows_out); \
		}                                                                                                               \
		break;
	switch (frames / 128) {
		GAS_ROWS_CASE(1)
		GAS_ROWS_CASE(2)
		GAS_ROWS_CASE(3)
		GAS_ROWS_CASE(4)
		default:
			return hipErrorInvalidValue;
	}
#undef GAS_ROWS_CASE
	return hipGetLastError();
}

hipError_t gas_launch_rows_accumulate(hipStream_t stream, const gas_group_args &g, uint32_t frames, float *partials, uint32_t p_offset) {
	if (g.n == 0) {
		return hipSuccess;
	}
	const uint32_t wgs = gas_hrtf_partials(g.n);
	dim3 grid(wgs), block(WAVES * 64);
	switch (frames / 64) {
		case 2:
			hipLaunchKernelGGL((k_rows_accumulate<2>), grid, block, 0, stream, g, partials, p_offset);
			break;
		case 4:
			hipLaunchKernelGGL((k_rows_accumulate<4>), grid, block, 0, stream, g, partials, p_offset);
			break;
		case 6:
			hipLaunchKernelGGL((k_rows_accumulate<6>), grid, block, 0, stream, g, partials, p_offset);
			break;
		case 8:
			hipLaunchKernelGGL((k_rows_accumulate<8>), grid, block, 0, stream, g, partials, p_offset);
			break;
		default:
			return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t gas_launch_rows_accumulate_buses(hipStream_t stream, const gas_group_args &g, uint32_t frames, const gas_bus_route *routes, uint32_t n_buses, uint32_t bus_rows, float *partials, uint32_t p_offset) {
	if (g.n == 0) {
		return hipSuccess;
	}
	if (!g.slots || !routes || n_buses == 0 || n_buses > GAS_MAX_BUSES) {
		return hipErrorInvalidValue;
	}
	const uint32_t wgs = gas_hrtf_partials(g.n);
	dim3 grid(wgs), block(WAVES * 64);
	switch (frames / 64) {
		case 2:
			hipLaunchKernelGGL((k_rows_accumulate_buses<2>), grid, block, 0, stream, g, routes, n_buses, bus_rows, partials, p_offset);
			break;
		case 4:
			hipLaunchKernelGGL((k_rows_accumulate_buses<4>), grid, block, 0, stream, g, routes, n_buses, bus_rows, partials, p_offset);
			break;
		case 6:
			hipLaunchKernelGGL((k_rows_accumulate_buses<6>), grid, block, 0, stream, g, routes, n_buses, bus_rows, partials, p_offset);
			break;
		case 8:
			hipLaunchKernelGGL((k_rows_accumulate_buses<8>), grid, block, 0, stream, g, routes, n_buses, bus_rows, partials, p_offset);
			break;
		default:
			return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t gas_launch_hrtf_table(hipStream_t stream, const float *d_hrir, uint32_t dirs, uint32_t taps, const float2 *twiddles, float4 *spec) {
	hipLaunchKernelGGL(k_hrtf_table, dim3(dirs * 2), dim3(64), 0, stream, d_hrir, dirs, taps, twiddles, spec);
	return hipGetLastError();
}

