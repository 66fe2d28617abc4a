// k_hrtf_multi.hip -- SEVERAL consecutive callbacks of the same plain-[HRTF] list in one launch (GAS_FLAG_PIPELINED_MIX).
//
// Same arithmetic, same per-wave source order, same summation order as k_hrtf_uni run once per callback -- the results
// are bitwise those of the separate launches (tests/test_gpu_pipelined.py) -- but a launch of K blocks pays the
// dispatch gap, the launch ramp and the tail once, and a block's epilogue no longer stops the stream of the next one:
//   * no workgroup barrier between blocks.  A wave that has finished its sources of block b parks its two spectra sums
//     in LDS (fd[wave]) and goes straight on to block b + 1 (whose first source it requested before parking);
//   * the epilogue of block b (sum of the eight waves' spectra in wave order, one inverse FFT per ear, store of the
//     workgroup's partial mix) is the job of ONE PAIR of waves, a different pair every block (2 (b mod 4), + 1), while
//     the other six keep streaming.  Measured on the single-block kernel (profiles/r02_notes.md): 2 us of a 15.6 us
//     launch are the barrier wait for the slowest wave and 1.5 us the epilogue itself, with HBM idle for both.
//   * hand-over through two LDS counters: `arrived` (waves that parked block b: the pair waits for 8 (b + 1)) and
//     `consumed` (epilogue waves that have read fd of block b: a wave parks block b + 1 only when it is 2 (b + 1)).
//     Every wave of the workgroup reaches every increment it owes unconditionally, so the waits terminate.
//   * a source's history row is stored by block b and read back by block b + 1 of the same wave (program order, L2).
// Parameters: block b ramps the gain from block b - 1's target (block 0: the slot's stored gain) to its own; a block
// with device-published rows (fresh[b]) takes gain and direction from them, any other repeats the previous block's.
// Callers: gas_ctx.hip batches consecutive gas_process_block calls of an unchanged list (GAS_FLAG_BATCHED_LAUNCH: calls
// are recorded until the batch is full; gas_ctx_join_outputs / gas_ctx_synchronize / any other entry runs what waits).
// Source rows are loaded non-temporal here: in a batched launch the HRIR rows ARE reused (the next block of an unchanged
// direction meets them again), and once-touched frames that do not displace them from the L2 are worth 4.5 % of the
// launch (10.65 vs 11.15 us per block, profiles/r02_notes.md).  The single-block kernels show no difference.
#define GAS_USE_NT 1
#ifndef GAS_MULTI_L2_PREFETCH
#define GAS_MULTI_L2_PREFETCH 0 // EXPERIMENT (off): touch the rows of the source two trips ahead, one dword per 128-byte line.  Measured SLOWER (8192 sources: 12.0 vs 11.2 us per block; 65536: 87.7 vs 75.1): the launch is bound by what the fabric moves, not by one wave's load latency, and the touches add L2 -> L1 traffic (profiles/r03_notes.md)
#endif
#include "gas_hrtf_wave.h"

namespace {

constexpr int MW = 8; // waves per workgroup, one workgroup per CU
constexpr int MAXB = GAS_HRTF_MULTI_MAX_BLOCKS;

template <int SQ>
struct MultiLds {
	static constexpr int SLICES_F2 = MW * LDS_F2_HALF; // one exchange slice per wave (no transform pair in this kernel)
	static constexpr int FD_F2 = MW * 2 * 512; // fd[wave][ear][512]
	static constexpr int TOTAL_F2 = SLICES_F2 + FD_F2;
	// what is left of the CU's 160 KiB holds the history rows of the wave's sources between blocks (see HIST_LDS)
	static constexpr int HL = (8 - SQ) * 64;
	static constexpr int LEFT_BYTES = 160 * 1024 - TOTAL_F2 * 8 - 1024 * 8 - 64;
	static constexpr int HIST_ROWS = LEFT_BYTES / (MW * HL * 4); // per wave: 6 at F = 512, 4 at F = 256
};

__device__ __forceinline__ void lds_wait_at_least(const uint32_t *flag, uint32_t want) {
	while (__atomic_load_n(flag, __ATOMIC_RELAXED) < want) {
		__builtin_amdgcn_s_sleep(1);
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// HIST_LDS: every wave has at most MultiLds::HIST_ROWS sources, so their history rows stay in LDS from block to block --
// read from HBM by the first block only, written back by the last one only.  Measured before (profiles/r02_notes.md):
// the launch runs at the fabric's rate over its actual traffic, and a row stored by block b did NOT come back out of
// the L2 for block b + 1 (FETCH_SIZE per block barely moved) -- 16.8 of a block's 74 MB were history.
template <int SQ, bool HIST_LDS>
__global__ __launch_bounds__(MW * 64, GAS_HRTF_WAVES_PER_SIMD) void k_hrtf_multi(gas_group_args g, gas_hrtf_blocks mb, const uint32_t *__restrict__ peak_bits, uint32_t peak_all, gas_dev_state st, gas_hrtf_table tab, const float2 *__restrict__ tw, float *__restrict__ partials) {
	constexpr int FQ = 2 * SQ; // F / 64
	constexpr int HQ = 8 - SQ; // hist_len / 64
	constexpr int NQ = 8 + SQ; // (hist_len + F) / 64
	constexpr uint32_t F = FQ * 64;
	constexpr uint32_t HL = HQ * 64;
	__shared__ float2 lds_all[MultiLds<SQ>::TOTAL_F2];
	__shared__ float2 tw_lds[1024];
	__shared__ float hist_lds[HIST_LDS ? MW * MultiLds<SQ>::HIST_ROWS * HL : 4];
	__shared__ uint32_t arrived, consumed;
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	float2 *lds = lds_all + wave * LDS_F2_HALF;
	float *my_hist = hist_lds + (HIST_LDS ? wave * MultiLds<SQ>::HIST_ROWS * HL : 0); // [source of this wave][HL], lane-major rows
	float2 *fd = lds_all + MultiLds<SQ>::SLICES_F2;
	const uint32_t K = mb.k;

	// ---- prologue (k_hrtf_uni's): twiddles, this wave's sources, the first source's data, block 0's parameters ----
	const float4 tw_in = reinterpret_cast<const float4 *>(tw)[threadIdx.x & 511];
	if (threadIdx.x == 0) {
		arrived = 0;
		consumed = 0;
	}
	uint32_t first, last;
	wave_range(g.n, blockIdx.x * MW + wave, gridDim.x * MW, first, last);
	const uint32_t cnt = last - first; // wave-uniform, <= 64: one metadata lane per source
	const bool have = (uint32_t)lane < cnt;
	uint32_t my_slot = 0, my_row = 0, my_dir = 0, my_flag = 0;
	float my_g0 = 0.0f, my_g1 = 0.0f;
	if (have) {
		const uint32_t e = first + lane;
		my_slot = g.slots ? g.slots[e] : g.slot_base + e;
		my_row = g.rows ? g.rows[e] : e;
	}
	gas_audio_frame raw[FQ];
	float rawh[HQ];
	if (cnt > 0) {
		const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)my_slot, 0), r0 = (uint32_t)__builtin_amdgcn_readlane((int)my_row, 0);
		load_history<HQ>(st.hrtf_hist + (size_t)s0 * HL, lane, rawh);
#pragma unroll
		for (int q = 0; q < FQ; q++) {
			raw[q] = (GAS_ABL & 8) ? gas_audio_frame{ (float)lane, (float)r0 } : nt_load_frame(mb.src[0] + (size_t)r0 * F + lane + 64 * q);
		}
	}
	// gain and direction of block b for this lane's source: its device-published row, else what it had
	auto block_params = [&](uint32_t b, float keep_g, uint32_t keep_d, float &g1, uint32_t &d) {
		const gas_params *fr = mb.fresh[b]; // wave-uniform
		g1 = keep_g;
		d = keep_d;
		if (have && (fr || b == 0)) {
			const gas_params *P = fr ? fr + my_row : st.params + my_slot;
			const float2 gd = *reinterpret_cast<const float2 *>(&P->hrtf_gain); // hrtf_gain, hrtf_dir: one 8-byte load
			const uint32_t dd = __float_as_uint(gd.y);
			g1 = gd.x;
			d = dd < tab.dirs ? dd : 0;
		}
	};
	if (have) {
		my_g0 = st.hrtf_prev_gain[my_slot];
		const uint32_t e = first + lane;
		my_flag = peak_all ? 1u : (peak_bits ? (peak_bits[e >> 5] >> (e & 31)) & 1u : 0u);
	}
	block_params(0, 0.0f, 0, my_g1, my_dir);
	if (threadIdx.x < 512) {
		reinterpret_cast<float4 *>(tw_lds)[threadIdx.x] = tw_in;
	}
	__syncthreads(); // the only workgroup barrier of the kernel
	float2 t1[8], t2[8];
#pragma unroll
	for (int k = 0; k < 8; k++) {
		t1[k] = tw_lds[k * 64 + lane];
		t2[k] = tw_lds[(8 + k) * 64 + lane];
	}
	const float lane_f = (float)lane;
#if GAS_MULTI_L2_PREFETCH
	uint32_t pf_word = 0; // the word of the line touch in flight
#endif

	for (uint32_t b = 0; b < K; b++) {
		float *peaks_b = mb.peaks[b];
		// this block's sums, the skewed spectrum and its table row: nothing of them lives across the block boundary
		float2 aYL[8], aYR[8], zp[8];
		float4 hs[8];
#pragma unroll
		for (int j = 0; j < 8; j++) {
			aYL[j] = make_float2(0.0f, 0.0f);
			aYR[j] = make_float2(0.0f, 0.0f);
			zp[j] = make_float2(0.0f, 0.0f);
		}
		// the next block's parameters travel during this block
		float nx_g1 = my_g1;
		uint32_t nx_dir = my_dir;
		if (b + 1 < K) {
			block_params(b + 1, my_g1, my_dir, nx_g1, nx_dir);
		}
		bool have_prev = false; // wave-uniform
		uint32_t prev_flag = 0, prev_row = 0;
		// products of the previous source (zp x hs), its exact peak if asked for, then the request for `next_dir`'s row
		auto products = [&](bool more, uint32_t next_dir) {
			float2 yl[8], yr[8];
			if (have_prev) {
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
			if (have_prev && prev_flag) { // this source's own output, for max |L|, max |R| (audio_spatializer.cpp:436-443)
				fft512<true>(yl, t1, t2, lds, lane); // one after the other (k_hrtf_uni interleaves the pair: same operations, same
				fft512<true>(yr, t1, t2, lds, lane); // bits, but ~20 more live registers than this kernel has)
				float pkl = 0.0f, pkr = 0.0f;
#pragma unroll
				for (int t = 0; t < SQ; t++) {
					pkl = fmaxf(pkl, fmaxf(fabsf(yl[HQ + t].x), fabsf(yl[HQ + t].y)));
					pkr = fmaxf(pkr, fmaxf(fabsf(yr[HQ + t].x), fabsf(yr[HQ + t].y)));
				}
				pkl = wave_max(pkl);
				pkr = wave_max(pkr);
				if (lane == 0) {
					peaks_b[(size_t)prev_row * 2] = pkl;
					peaks_b[(size_t)prev_row * 2 + 1] = pkr;
				}
			}
		};

		for (uint32_t i = 0; i < cnt; i++) {
			const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)my_slot, (int)i);
			const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)my_row, (int)i);
			const uint32_t dir = (uint32_t)__builtin_amdgcn_readlane((int)my_dir, (int)i);
			const float g0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_g0), (int)i));
			const float g1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_g1), (int)i));
			const uint32_t flag = (uint32_t)__builtin_amdgcn_readlane((int)my_flag, (int)i);
			float xq[NQ];
			if (HIST_LDS && b > 0) { // wave-uniform: the row this wave parked in the previous block
				load_history<HQ>(my_hist + (size_t)i * HL, lane, rawh);
			}
#pragma unroll
			for (int q = 0; q < HQ; q++) {
				xq[q] = rawh[q];
			}
			// gain ramp weights t = f / F and 1 - t of this lane's frames (k_hrtf_uni keeps the 2 FQ values in registers across
			// the sources; this kernel has none to spare, so they are rebuilt -- the same exact integers, the same product --
			// from a lane number the compiler cannot see through, or it would hoist them right back)
			float lf = lane_f;
			asm volatile("" : "+v"(lf));
#pragma unroll
			for (int q = 0; q < FQ; q++) {
				const float mono = (raw[q].left + raw[q].right) * 0.5f;
				const float t = (lf + (float)(64 * q)) * (1.0f / (float)F);
				xq[HQ + q] = mono * (g1 * t + (1 - t) * g0);
			}
			// new history = x_full[F .. F + HL): to the slot's row, or (HIST_LDS) parked for the next block and written
			// back by the last one only
			if (!HIST_LDS || b + 1 == K) {
				store_history<HQ>(st.hrtf_hist + (size_t)slot * HL, lane, &xq[FQ]);
			} else {
				store_history<HQ>(my_hist + (size_t)i * HL, lane, &xq[FQ]);
			}
			// the landing registers are free again: the next source of this block, or the first one of the next block
			// (whose history row this wave stored earlier in this block, or just above when it has a single source)
			const bool wrap = i + 1 == cnt;
			if (!wrap || b + 1 < K) {
				const uint32_t ni = wrap ? 0u : i + 1;
				const uint32_t nslot = (uint32_t)__builtin_amdgcn_readlane((int)my_slot, (int)ni);
				const uint32_t nrow = (uint32_t)__builtin_amdgcn_readlane((int)my_row, (int)ni);
				const gas_audio_frame *nsrc = mb.src[wrap ? b + 1 : b] + (size_t)nrow * F;
				if (!HIST_LDS || (b == 0 && !wrap)) { // HIST_LDS: only block 0 reads history rows from memory
					load_history<HQ>(st.hrtf_hist + (size_t)nslot * HL, lane, rawh);
				}
#pragma unroll
				for (int q = 0; q < FQ; q++) {
					raw[q] = (GAS_ABL & 8) ? gas_audio_frame{ (float)lane, (float)nrow } : nt_load_frame(nsrc + lane + 64 * q);
				}
			}
			float2 zs[8];
#pragma unroll
			for (int j = 0; j < 8; j++) {
				zs[j] = make_float2(xq[j], xq[j + SQ]);
			}
			fft512<false>(zs, t1, t2, lds, lane);
			products(true, dir);
#if GAS_MULTI_L2_PREFETCH
			// The row of the source TWO trips ahead is pulled towards the L2 by one dword load per 128-byte line (no
			// landing set: the loaded word is never looked at), issued behind this trip's requests AND behind the
			// table row's (the wave's vector-memory results return in order, and the compiler's wait in front of the spectral
			// products is vmcnt(0): anything requested before them would be waited for there).  A trip's own requests then find their lines a cache hit
			// away instead of an HBM round trip under load (the trip is latency-bound with one source in flight per wave,
			// profiles/r03_notes.md).  The word requested a trip ago is retired first (it has landed long since).
			{
				asm volatile("" ::"v"(pf_word));
				const uint32_t i2 = i + 2, over = i2 / cnt, ni2 = i2 - over * cnt; // cnt >= 1; over <= 2
				if (b + over < K) {
					const uint32_t row2 = (uint32_t)__builtin_amdgcn_readlane((int)my_row, (int)ni2);
					const char *p2 = reinterpret_cast<const char *>(mb.src[b + over] + (size_t)row2 * F);
					pf_word = *reinterpret_cast<const uint32_t *>(p2 + ((uint32_t)lane * 128u) % (F * 8u));
				}
			}
#endif
#pragma unroll
			for (int j = 0; j < 8; j++) {
				zp[j] = zs[j];
			}
			have_prev = true;
			prev_flag = flag;
			prev_row = row;
		}
		// GAS_FLAG_PIPELINED_MIX: four waves of the first workgroups each sum one float4 column of an EARLIER launch's
		// partial mixes (k_mix_reduce's job, same code, same bits); not the two that run this block's epilogue
		const int ep0 = (int)(2 * (b & 3));
		const gas_deferred_reduce job = mb.job[b];
		float4 jr[JOB_ROWS];
		const uint32_t jw = (uint32_t)((wave - ep0 - 2) & 7), jx = blockIdx.x & 7, ji = blockIdx.x >> 3;
		const uint32_t job_col = (((ji >> 1) * 8 + jx) * 8) + (ji & 1) * 4 + jw;
		const bool job_mine = job.partials != nullptr && jw < GAS_HRTF_JOB_WAVES && job_col < job.elems / 4; // wave-uniform
		products(false, 0); // the last source's products (and peak)

		// ---- park this block's sums; one pair of waves turns the eight waves' sums into the partial mix -----------
		lds_wait_at_least(&consumed, 2 * b); // the previous block's epilogue has read fd
#pragma unroll
		for (int j = 0; j < 8; j++) {
			fd[(wave * 2 + 0) * 512 + j * 64 + lane] = aYL[j];
			fd[(wave * 2 + 1) * 512 + j * 64 + lane] = aYR[j];
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		if (lane == 0) {
			__atomic_fetch_add(&arrived, 1u, __ATOMIC_RELAXED);
		}
		if (job_mine) { // (the rows are requested only now: parked across the last products they cost 16 VGPRs too many)
			job_issue(job, job_col, lane, jr);
			job_finish(job, job_col, lane, jr);
		}
		if (wave == ep0 || wave == ep0 + 1) {
			const int ear = wave - ep0;
			lds_wait_at_least(&arrived, MW * (b + 1));
			float2 y[8];
#pragma unroll
			for (int j = 0; j < 8; j++) {
				y[j] = fd[ear * 512 + j * 64 + lane];
			}
#pragma unroll
			for (int w = 1; w < MW; w++) { // ((w0 + w1) + w2) + ... : k_hrtf_uni's order
#pragma unroll
				for (int j = 0; j < 8; j++) {
					y[j] = cadd(y[j], fd[(w * 2 + ear) * 512 + j * 64 + lane]);
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the reads above are done
			if (lane == 0) {
				__atomic_fetch_add(&consumed, 1u, __ATOMIC_RELAXED);
			}
			fft512<true>(y, t1, t2, lds, lane);
			float *my_partial = partials + ((size_t)mb.p_offset[b] + blockIdx.x) * (size_t)(F * 2);
#pragma unroll
			for (int t = 0; t < SQ; t++) {
				const int fa = lane + 64 * t, fb = lane + 64 * (SQ + t);
				my_partial[fa * 2 + ear] = y[HQ + t].x;
				my_partial[fb * 2 + ear] = y[HQ + t].y;
			}
		}
		if (have && !my_flag) { // "not measured": never passes the gate (audio_spatializer.cpp:464-469)
			*reinterpret_cast<float2 *>(peaks_b + (size_t)my_row * 2) = make_float2(__builtin_inff(), __builtin_inff());
		}
		my_g0 = my_g1;
		my_g1 = nx_g1;
		my_dir = nx_dir;
	}

	// ---- behind the blocks: what nobody waits for -------------------------------------------------------------------
	if (have) {
		st.hrtf_prev_gain[my_slot] = my_g0; // the last block's target
		if (mb.last_fresh) { // device-published parameter rows go through to the slot table (saves the scatter launch)
			const float4 *src4 = reinterpret_cast<const float4 *>(mb.last_fresh + my_row);
			float4 *dst4 = reinterpret_cast<float4 *>(st.params + my_slot);
#pragma unroll
			for (int k = 0; k < 8; k++) {
				dst4[k] = src4[k];
			}
		}
	}
}

} // namespace

bool gas_hrtf_multi_hist_in_lds(uint32_t n, uint32_t frames) {
	const uint32_t wgs = gas_hrtf_uni_partials(n);
	const uint32_t per_wave = (n + wgs * MW - 1) / (wgs * MW);
	switch (frames / 128) {
		case 1:
			return per_wave <= (uint32_t)MultiLds<1>::HIST_ROWS;
		case 2:
			return per_wave <= (uint32_t)MultiLds<2>::HIST_ROWS;
		case 3:
			return per_wave <= (uint32_t)MultiLds<3>::HIST_ROWS;
		default:
			return per_wave <= (uint32_t)MultiLds<4>::HIST_ROWS;
	}
}

hipError_t gas_launch_hrtf_multi(hipStream_t stream, const gas_group_args &g, const gas_hrtf_blocks &mb, const uint32_t *peak_bits, bool peak_all, const gas_dev_state &st, const gas_hrtf_table &tab, const float2 *twiddles, uint32_t frames, uint32_t hist_len, float *partials) {
	if (g.n == 0 || mb.k == 0) {
		return hipSuccess;
	}
	const uint32_t wgs = gas_hrtf_uni_partials(g.n);
	// every wave needs a source (the hand-over counts all eight waves), the list in entry order, float rows
	if (frames % 128 != 0 || frames > 512 || hist_len != 512 - frames / 2 || g.order != nullptr || mb.k > (uint32_t)MAXB || g.n < wgs * MW || g.n > wgs * MW * 64 || gas_hrtf_uni_waves() != (uint32_t)MW) {
		return hipErrorInvalidValue;
	}
	dim3 grid(wgs), block(MW * 64);
	const uint32_t all = peak_all ? 1u : 0u;
	const uint32_t per_wave = (g.n + wgs * MW - 1) / (wgs * MW); // the most sources a wave gets (wave_range)
#define GAS_MULTI_CASE(SQv)                                                                                                                     \
	case SQv:                                                                                                                                   \
		if (per_wave <= (uint32_t)MultiLds<SQv>::HIST_ROWS) {                                                                                   \
			hipLaunchKernelGGL((k_hrtf_multi<SQv, true>), grid, block, 0, stream, g, mb, peak_bits, all, st, tab, twiddles, partials);          \
		} else {                                                                                                                                \
			hipLaunchKernelGGL((k_hrtf_multi<SQv, false>), grid, block, 0, stream, g, mb, peak_bits, all, st, tab, twiddles, partials);         \
		}                                                                                                                                       \
		break;
	switch (frames / 128) {
		GAS_MULTI_CASE(1)
		GAS_MULTI_CASE(2)
		GAS_MULTI_CASE(3)
		GAS_MULTI_CASE(4)
		default:
			return hipErrorInvalidValue;
	}
#undef GAS_MULTI_CASE
	return hipGetLastError();
}
