// k_misc.hip -- the small kernels around the two DSP kernels:
//   k_mix_reduce      final, deterministic sum of the per-workgroup partial mixes -> mix_buffer[c][F]
//                     (the serial `+=` of audio_spatializer.cpp:433-434,450-451 turned into a fixed-order tree;
//                     also the zeroing of :335-343 when there is nothing to add)
//   k_scatter_params  publish staged SpatializerParameters PODs into the slot-indexed table
//                     (set_spatializer_parameters, audio_spatializer.cpp:558-564)
//   k_zero_slot       fresh SpatializerPlaybackData for a (re)started playback (audio_spatializer.cpp:69)
#include "gas_device.h"
#include "gas_internal.h"

namespace {

constexpr int RED_COLS = 4; // float4 columns (16 output floats) per workgroup: one wave each

// out[c][i] = sum_p partials[c][p][i].  One wave per float4 column; lane l adds partial rows l, l + 64, ... (four
// independent 16-byte loads in flight per trip), then the 64 lane sums are folded in the fixed order of
// gas_wave_sum.  Bitwise reproducible; no atomics; no LDS.
__global__ __launch_bounds__(RED_COLS * 64) void k_mix_reduce(const float *__restrict__ partials, uint32_t p_count, uint32_t p_stride, uint32_t elems /* F*2, multiple of 4 */, float *__restrict__ out) {
	const int lane = threadIdx.x & 63;
	const uint32_t i4 = blockIdx.x * RED_COLS + (threadIdx.x >> 6); // float4 index within a partial
	const uint32_t c = blockIdx.y;
	const uint32_t e4 = elems / 4;
	if (i4 >= e4) { // wave-uniform
		return;
	}
	float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
	const float4 *p = reinterpret_cast<const float4 *>(partials + (size_t)c * p_stride * elems) + i4;
	uint32_t k = lane;
	for (; k + 3 * 64 < p_count; k += 4 * 64) {
		const float4 a0 = p[(size_t)k * e4];
		const float4 a1 = p[(size_t)(k + 64) * e4];
		const float4 a2 = p[(size_t)(k + 128) * e4];
		const float4 a3 = p[(size_t)(k + 192) * e4];
		gas_mix_column_add(s, a0);
		gas_mix_column_add(s, a1);
		gas_mix_column_add(s, a2);
		gas_mix_column_add(s, a3);
	}
	for (; k < p_count; k += 64) {
		gas_mix_column_add(s, p[(size_t)k * e4]);
	}
	const float4 t = gas_mix_column_fold(s);
	if (lane == 0) {
		reinterpret_cast<float4 *>(out + (size_t)c * elems)[i4] = t;
	}
}

// The same sum for several independent (partials, out) pairs in one launch: blockIdx.y picks the pair.  What
// gas_ctx_join_outputs runs when a batched launch left one pending sum per block.
__global__ __launch_bounds__(RED_COLS * 64) void k_mix_reduce_jobs(gas_reduce_jobs jobs, uint32_t elems) {
	const int lane = threadIdx.x & 63;
	const uint32_t i4 = blockIdx.x * RED_COLS + (threadIdx.x >> 6);
	const uint32_t e4 = elems / 4;
	if (i4 >= e4) { // wave-uniform
		return;
	}
	const uint32_t p_count = jobs.p_count[blockIdx.y];
	float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
	const float4 *p = reinterpret_cast<const float4 *>(jobs.partials[blockIdx.y]) + i4;
	uint32_t k = lane;
	for (; k + 3 * 64 < p_count; k += 4 * 64) {
		const float4 a0 = p[(size_t)k * e4];
		const float4 a1 = p[(size_t)(k + 64) * e4];
		const float4 a2 = p[(size_t)(k + 128) * e4];
		const float4 a3 = p[(size_t)(k + 192) * e4];
		gas_mix_column_add(s, a0);
		gas_mix_column_add(s, a1);
		gas_mix_column_add(s, a2);
		gas_mix_column_add(s, a3);
	}
	for (; k < p_count; k += 64) {
		gas_mix_column_add(s, p[(size_t)k * e4]);
	}
	const float4 t = gas_mix_column_fold(s);
	if (lane == 0) {
		reinterpret_cast<float4 *>(jobs.out[blockIdx.y])[i4] = t;
	}
}

__global__ void k_scatter_params(gas_params *__restrict__ table, const gas_params *__restrict__ upload, const uint32_t *__restrict__ slots, uint32_t n) {
	// 8 lanes move one 128-byte POD as 16-byte pieces
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t e = t >> 3, part = t & 7;
	if (e < n) {
		const float4 *s = reinterpret_cast<const float4 *>(upload + e);
		float4 *d = reinterpret_cast<float4 *>(table + slots[e]);
		d[part] = s[part];
	}
}

__global__ void k_zero_slot(gas_dev_state st, uint32_t slot, uint32_t hist_len, uint32_t er_ring_frames) {
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t < 8) {
		for (int f = 0; f < GAS_BQ_FIELDS; f++) {
			st.bq[(size_t)f * st.bq_stride + (size_t)slot * 8 + t] = 0.0f;
		}
	}
	if (t < hist_len) {
		st.hrtf_hist[(size_t)slot * hist_len + t] = 0.0f;
	}
	if (t == 0) {
		st.was_further[slot] = 0;
		st.hrtf_prev_gain[slot] = 0.0f;
		st.hrtf_prev_dir[slot] = 0;
		if (st.er_pos) {
			st.er_pos[slot] = 0;
		}
	}
	if (st.er_ring) {
		for (uint32_t i = t; i < er_ring_frames; i += gridDim.x * blockDim.x) {
			st.er_ring[(size_t)slot * er_ring_frames + i] = gas_audio_frame{ 0.0f, 0.0f };
		}
	}
}

__global__ void k_noop() {
}

// Same-run copy-bandwidth ceiling: a pure streaming launch over a given byte count -- 16-byte loads
// summed into a register, 16-byte stores of a constant -- timed exactly like the dominant kernel.  No arithmetic to
// speak of and every load independent: what this launch achieves is what the memory system gives a launch of this size.
template <int UNROLL>
__global__ __launch_bounds__(256) void k_stream_probe(const float4 *__restrict__ rd, uint64_t rd_n4, float4 *__restrict__ wr, uint64_t wr_n4, float *__restrict__ sink) {
	const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
	uint64_t i = tid;
	for (; i + (UNROLL - 1) * stride < rd_n4; i += UNROLL * stride) {
		float4 v[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; u++) {
			v[u] = rd[i + u * stride];
		}
#pragma unroll
		for (int u = 0; u < UNROLL; u++) {
			acc.x += v[u].x;
			acc.y += v[u].y;
			acc.z += v[u].z;
			acc.w += v[u].w;
		}
	}
	for (; i < rd_n4; i += stride) {
		const float4 v = rd[i];
		acc.x += v.x;
		acc.y += v.y;
		acc.z += v.z;
		acc.w += v.w;
	}
	const float4 k = make_float4(1.0f, 2.0f, 3.0f, (float)blockIdx.x);
	for (uint64_t j = tid; j < wr_n4; j += stride) {
		wr[j] = k;
	}
	if (acc.x + acc.y + acc.z + acc.w == 12345.678f) { // never true for the zero-filled arena; keeps the loads alive
		sink[tid & 255] = acc.x;
	}
}

// Direction order of a launch group's HRTF sources: order[k] = group entry, such that entries with
// the same HRIR direction are adjacent.  k_hrtf_ols adds the windows of a run of equal-direction sources in the time
// domain and pays one FFT and one table row for the run.  The order is an optimisation only (any permutation is
// correct), so it is built per SEGMENT of DIR_SEG entries: one workgroup, no cross-workgroup step, one launch.
//
// Stable counting sort without atomics (bitwise-reproducible order => bitwise-reproducible sums): wave w owns the
// contiguous entries [w*512, w*512+512) of the segment and walks them 64 at a time; lanes with equal keys find each
// other with one ballot per key bit; hist[w][key] counts the wave's entries per key in order; a per-key prefix over
// the 16 waves and one exclusive scan over the keys give every entry its destination.
#ifndef GAS_DIRSORT_ABL
#define GAS_DIRSORT_ABL 0 // timing experiments (tools/dirsort_probe.sh): 1 synthetic keys (no loads), 2 no rank loop; both keep every index in range
#endif
constexpr int DIR_SEG = GAS_DIR_ORDER_SEGMENT;
constexpr int DIR_WAVES = 16;
constexpr int DIR_ROUNDS = DIR_SEG / (DIR_WAVES * 64);
constexpr int DIR_KEY_BITS = 12; // dirs <= 4096

__global__ __launch_bounds__(DIR_WAVES * 64) void k_dir_order(gas_group_args g, const gas_params *__restrict__ params, const gas_params *__restrict__ fresh, uint32_t dirs, uint32_t *__restrict__ order) {
	extern __shared__ uint32_t dir_lds[];
	uint32_t *base = dir_lds; // [dirs] totals, then exclusive scan
	uint16_t *hist = reinterpret_cast<uint16_t *>(dir_lds + dirs); // [DIR_WAVES][dirs]
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	const uint32_t seg0 = blockIdx.x * DIR_SEG;
	for (uint32_t i = threadIdx.x; i < DIR_WAVES * dirs / 2; i += DIR_WAVES * 64) {
		reinterpret_cast<uint32_t *>(hist)[i] = 0;
	}
	// every load is unconditional (out-of-range lanes read entry 0 and are masked afterwards) and the three
	// dependent levels are issued round by round, so the wave waits for three round trips, not 3 x DIR_ROUNDS
	uint32_t key[DIR_ROUNDS], slot[DIR_ROUNDS], row[DIR_ROUNDS];
	bool valid[DIR_ROUNDS];
#pragma unroll
	for (int r = 0; r < DIR_ROUNDS; r++) {
		const uint32_t e = seg0 + wave * (DIR_ROUNDS * 64) + r * 64 + lane;
		valid[r] = e < g.n;
		const uint32_t ec = valid[r] ? e : 0;
		slot[r] = g.slots ? g.slots[ec] : g.slot_base + ec;
		row[r] = g.rows ? g.rows[ec] : ec;
	}
#pragma unroll
	for (int r = 0; r < DIR_ROUNDS; r++) {
		const gas_params *P = fresh ? fresh + row[r] : params + slot[r];
		const uint32_t d = (GAS_DIRSORT_ABL & 1) ? (slot[r] * 2654435761u) >> 22 : P->hrtf_dir;
		key[r] = valid[r] && d < dirs ? d : 0; // the clamp of k_hrtf_ols
	}
	__syncthreads();
	uint16_t *myh = hist + (size_t)wave * dirs;
	uint32_t rank[DIR_ROUNDS] = {};
	const uint64_t lt = (1ull << lane) - 1;
#pragma unroll
	for (int r = 0; r < ((GAS_DIRSORT_ABL & 2) ? 0 : DIR_ROUNDS); r++) {
		uint64_t same = __ballot(valid[r]);
#pragma unroll
		for (int b = 0; b < DIR_KEY_BITS; b++) {
			const bool bit = (key[r] >> b) & 1;
			const uint64_t bb = __ballot(bit);
			same &= bit ? bb : ~bb;
		}
		const uint32_t before = (uint32_t)__popcll(same & lt);
		const uint32_t count = (uint32_t)__popcll(same);
		uint32_t prev = 0;
		if (valid[r]) {
			prev = myh[key[r]];
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		if (valid[r] && before == count - 1) { // one lane per distinct key of this round
			myh[key[r]] = (uint16_t)(prev + count);
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		rank[r] = prev + before;
	}
	__syncthreads();
	for (uint32_t k = threadIdx.x; k < dirs; k += DIR_WAVES * 64) {
		uint32_t run = 0;
#pragma unroll
		for (int w = 0; w < DIR_WAVES; w++) {
			const uint32_t cnt = hist[(size_t)w * dirs + k];
			hist[(size_t)w * dirs + k] = (uint16_t)run;
			run += cnt;
		}
		base[k] = run;
	}
	__syncthreads();
	if (wave == 0) { // exclusive scan of the per-key totals, 64 keys per trip (DPP row shifts + row broadcasts)
		uint32_t carry = 0;
		for (uint32_t k0 = 0; k0 < dirs; k0 += 64) {
			const uint32_t k = k0 + lane;
			const uint32_t v = k < dirs ? base[k] : 0;
			int inc = (int)v;
#define GAS_DPP_ADD(ctrl, rows) inc += __builtin_amdgcn_update_dpp(0, inc, ctrl, rows, 0xF, false)
			GAS_DPP_ADD(0x111, 0xF); // row_shr:1
			GAS_DPP_ADD(0x112, 0xF); // row_shr:2
			GAS_DPP_ADD(0x114, 0xF); // row_shr:4
			GAS_DPP_ADD(0x118, 0xF); // row_shr:8
			GAS_DPP_ADD(0x142, 0xA); // row_bcast:15 -> rows 1, 3
			GAS_DPP_ADD(0x143, 0xC); // row_bcast:31 -> rows 2, 3
#undef GAS_DPP_ADD
			if (k < dirs) {
				base[k] = carry + (uint32_t)inc - v;
			}
			carry += (uint32_t)__builtin_amdgcn_readlane(inc, 63);
		}
	}
	__syncthreads();
#pragma unroll
	for (int r = 0; r < DIR_ROUNDS; r++) {
		if (valid[r]) {
			const uint32_t e = seg0 + wave * (DIR_ROUNDS * 64) + r * 64 + lane;
			const uint32_t dest = base[key[r]] + hist[(size_t)wave * dirs + key[r]] + rank[r];
			order[seg0 + dest] = e;
		}
	}
}


// k_xcd_order -- XCD-affine processing order for k_hrtf_uni (DESIGN.md 3.1 "table locality").
// Workgroup b of a launch runs on XCD b % 8, and every XCD has its own 4 MiB L2: with random directions each XCD pulls
// its own copy of (nearly) every HRIR spectra row it touches, which at 8192 sources is 20 MB of L2 fills on top of 54 MB
// of algorithmic bytes and at 65536 sources thrashes the L2 outright (measured 1.46x, profiles/r02_*).  This kernel
// permutes the callback's entries so that the sources whose direction lies in eighth k of the table are processed by
// workgroups b = k (mod 8): every XCD then keeps touching the same 1/8 of the table.
//
// One workgroup per SEGMENT of eight consecutive k_hrtf_uni workgroups (one per XCD).  A segment's entries are a
// contiguous range of the list; bucket k's entries fill workgroup 8 s + k's range in list order; what does not fit
// (buckets are never exactly equal) fills the holes the short buckets leave, in list order too.  Stable counting sort,
// no atomics: the order -- and with it the floating-point summation order of the mix -- is a pure function of the
// list and the directions.  The segment is walked in tiles of one entry per thread (every direction of a tile is one
// parallel round trip: a first version with a sequential chunk per thread took 16 us at 65536 sources, all of it load
// latency); pass 1 counts the buckets, pass 2 ranks (ballots within a wave, LDS across waves, a running base across
// tiles) and places.
constexpr int XO_WAVES = 16;
constexpr int XO_THREADS = XO_WAVES * 64;

__device__ __forceinline__ uint32_t xo_first(uint32_t gw, uint32_t base, uint32_t rem) { // wave_range()'s first
	return gw * base + (gw < rem ? gw : rem);
}

__global__ __launch_bounds__(XO_THREADS) void k_xcd_order(gas_group_args g, const gas_params *__restrict__ params, const gas_params *__restrict__ fresh, uint32_t dirs, uint32_t n_waves, uint32_t waves_per_wg, uint32_t *__restrict__ order) {
	// the bucket tables live in LDS and are read by bucket number (eight distinct addresses per wave: broadcasts); as
	// per-thread arrays they cost 190 spilled VGPRs at 1024 threads
	__shared__ uint32_t wcnt[XO_WAVES][8]; // per-wave bucket counts (pass 1: of the segment, pass 2: of the current tile)
	__shared__ uint32_t wpre[XO_WAVES][8]; // pass 2: entries of bucket j in the lower waves of the tile
	__shared__ uint32_t start[9], quota[8], count[8], over_pre[8], hole_pre[9], seen[8];
	const uint32_t t = threadIdx.x;
	const int lane = t & 63, wave = t >> 6;
	const uint32_t base = g.n / n_waves, rem = g.n % n_waves;
	const uint32_t wg0 = blockIdx.x * 8; // first k_hrtf_uni workgroup of this segment
	const uint32_t A = xo_first(wg0 * waves_per_wg, base, rem), S = xo_first((wg0 + 8) * waves_per_wg, base, rem) - A;
	if (t < 9) {
		start[t] = xo_first((wg0 + t) * waves_per_wg, base, rem);
	}
	// bucket of list entry A + i (i < S), 8 for i >= S: every load unconditional, masked afterwards
	auto bucket_of = [&](uint32_t i) -> uint32_t {
		const uint32_t e = A + (i < S ? i : 0);
		const uint32_t slot = g.slots ? g.slots[e] : g.slot_base + e;
		const uint32_t row = g.rows ? g.rows[e] : e;
		const gas_params *P = fresh ? fresh + row : params + slot;
		uint32_t d = P->hrtf_dir;
		d = d < dirs ? d : 0; // the clamp of k_hrtf_uni
		const uint32_t k = d * 8u / dirs;
		return i < S ? (k < 8 ? k : 7) : 8u;
	};
	// this wave's count of bucket `lane` (lanes 0..7) among the 64 values k
	auto wave_counts = [&](uint32_t k) -> uint32_t {
		uint32_t wv = 0;
#pragma unroll
		for (int j = 0; j < 8; j++) {
			const uint32_t c = (uint32_t)__popcll(__ballot(k == (uint32_t)j));
			wv = lane == j ? c : wv;
		}
		return wv;
	};
	// ---- pass 1: bucket totals of the segment (four tiles' loads in flight per trip) -----------------------------
	uint32_t mine = 0; // lanes 0..7: this wave's entries of bucket `lane`
	for (uint32_t i0 = 0; i0 < S; i0 += 4 * XO_THREADS) {
		uint32_t k4[4];
#pragma unroll
		for (int u = 0; u < 4; u++) {
			k4[u] = bucket_of(i0 + u * XO_THREADS + t);
		}
#pragma unroll
		for (int u = 0; u < 4; u++) {
			mine += wave_counts(k4[u]);
		}
	}
	if (lane < 8) {
		wcnt[wave][lane] = mine;
	}
	__syncthreads();
	if (t == 0) { // quotas, the overflow of the long buckets and the holes of the short ones
		uint32_t ov = 0, ho = 0;
		for (int k = 0; k < 8; k++) {
			uint32_t c = 0;
			for (int w = 0; w < XO_WAVES; w++) {
				c += wcnt[w][k];
			}
			const uint32_t q = start[k + 1] - start[k];
			quota[k] = q;
			count[k] = c;
			over_pre[k] = ov;
			hole_pre[k] = ho;
			seen[k] = 0;
			ov += c > q ? c - q : 0u;
			ho += c < q ? q - c : 0u;
		}
		hole_pre[8] = ho;
	}
	// ---- pass 2: rank and place, tile by tile ----------------------------------------------------------------------
	const uint64_t lt = (1ull << lane) - 1;
	for (uint32_t i0 = 0; i0 < S; i0 += 2 * XO_THREADS) {
		uint32_t k2[2];
#pragma unroll
		for (int u = 0; u < 2; u++) {
			k2[u] = bucket_of(i0 + u * XO_THREADS + t);
		}
#pragma unroll
		for (int u = 0; u < 2; u++) {
			const uint32_t i = i0 + u * XO_THREADS + t;
			const uint32_t k = k2[u];
			uint32_t before = 0;
#pragma unroll
			for (int j = 0; j < 8; j++) {
				const uint64_t m = __ballot(k == (uint32_t)j);
				before = k == (uint32_t)j ? (uint32_t)__popcll(m & lt) : before;
			}
			const uint32_t wv = wave_counts(k);
			__syncthreads(); // the previous tile's tables have been read (and thread 0's tables are written)
			if (lane < 8) {
				wcnt[wave][lane] = wv;
			}
			__syncthreads();
			if (t < XO_WAVES * 8) { // thread (w, j): bucket j's entries in the waves below w
				const uint32_t w = t >> 3, j = t & 7;
				uint32_t lower = 0;
				for (uint32_t x = 0; x < w; x++) {
					lower += wcnt[x][j];
				}
				wpre[w][j] = lower;
			}
			__syncthreads();
			if (k < 8) {
				const uint32_t r = seen[k] + wpre[wave][k] + before, q = quota[k];
				uint32_t pos;
				if (r < q) {
					pos = start[k] + r;
				} else {
					const uint32_t o = over_pre[k] + (r - q); // this entry's number among the overflowing ones
					uint32_t j = 0;
					while (j < 7 && o >= hole_pre[j + 1]) {
						j++;
					}
					pos = start[j] + count[j] + (o - hole_pre[j]);
				}
				order[pos] = A + i;
			}
			__syncthreads(); // everyone has read seen[]
			if (t < 8) {
				seen[t] += wpre[XO_WAVES - 1][t] + wcnt[XO_WAVES - 1][t];
			}
		}
	}
}

} // namespace

hipError_t gas_launch_stream_probe(hipStream_t stream, const void *rd, uint64_t rd_bytes, void *wr, uint64_t wr_bytes, uint32_t workgroups, uint32_t unroll, float *sink) {
	const float4 *r = static_cast<const float4 *>(rd);
	float4 *w = static_cast<float4 *>(wr);
	switch (unroll) {
		case 1:
			hipLaunchKernelGGL(k_stream_probe<1>, dim3(workgroups), dim3(256), 0, stream, r, rd_bytes / 16, w, wr_bytes / 16, sink);
			break;
		case 2:
			hipLaunchKernelGGL(k_stream_probe<2>, dim3(workgroups), dim3(256), 0, stream, r, rd_bytes / 16, w, wr_bytes / 16, sink);
			break;
		case 8:
			hipLaunchKernelGGL(k_stream_probe<8>, dim3(workgroups), dim3(256), 0, stream, r, rd_bytes / 16, w, wr_bytes / 16, sink);
			break;
		default:
			hipLaunchKernelGGL(k_stream_probe<4>, dim3(workgroups), dim3(256), 0, stream, r, rd_bytes / 16, w, wr_bytes / 16, sink);
			break;
	}
	return hipGetLastError();
}

hipError_t gas_launch_noop(hipStream_t stream) {
	hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, stream);
	return hipGetLastError();
}

hipError_t gas_launch_xcd_order(hipStream_t stream, const gas_group_args &g, const gas_params *params, const gas_params *fresh, uint32_t dirs, uint32_t hrtf_wgs, uint32_t waves_per_wg, uint32_t *order) {
	if (g.n == 0) {
		return hipSuccess;
	}
	if (hrtf_wgs % 8 != 0 || dirs < 8) {
		return hipErrorInvalidValue;
	}
	hipLaunchKernelGGL(k_xcd_order, dim3(hrtf_wgs / 8), dim3(XO_THREADS), 0, stream, g, params, fresh, dirs, hrtf_wgs * waves_per_wg, waves_per_wg, order);
	return hipGetLastError();
}

bool gas_dir_order_supported(uint32_t dirs) {
	return dirs >= 1 && dirs <= (1u << DIR_KEY_BITS);
}

hipError_t gas_launch_dir_order(hipStream_t stream, const gas_group_args &g, const gas_params *params, const gas_params *fresh, uint32_t dirs, uint32_t *order) {
	if (g.n == 0) {
		return hipSuccess;
	}
	const uint32_t d2 = (dirs + 1) & ~1u; // keeps the u16 table's zeroing loop word-aligned
	const size_t lds = (size_t)d2 * 4 + (size_t)DIR_WAVES * d2 * 2;
	if (lds > 48 * 1024) { // large dynamic LDS needs the opt-in, per device: cheap enough to repeat (the sort runs per publish)
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_dir_order), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		if (e != hipSuccess) {
			return e;
		}
	}
	hipLaunchKernelGGL(k_dir_order, dim3((g.n + DIR_SEG - 1) / DIR_SEG), dim3(DIR_WAVES * 64), lds, stream, g, params, fresh, d2, order);
	return hipGetLastError();
}

hipError_t gas_launch_mix_reduce_jobs(hipStream_t stream, const gas_reduce_jobs &jobs, uint32_t frames) {
	if (jobs.count == 0) {
		return hipSuccess;
	}
	const uint32_t elems = frames * 2;
	dim3 grid((elems / 4 + RED_COLS - 1) / RED_COLS, jobs.count);
	hipLaunchKernelGGL(k_mix_reduce_jobs, grid, dim3(RED_COLS * 64), 0, stream, jobs, elems);
	return hipGetLastError();
}

hipError_t gas_launch_mix_reduce(hipStream_t stream, const float *partials, uint32_t p_count, uint32_t p_stride, uint32_t channels, uint32_t frames, gas_audio_frame *out) {
	const uint32_t elems = frames * 2;
	dim3 grid((elems / 4 + RED_COLS - 1) / RED_COLS, channels);
	hipLaunchKernelGGL(k_mix_reduce, grid, dim3(RED_COLS * 64), 0, stream, partials, p_count, p_stride, elems, reinterpret_cast<float *>(out));
	return hipGetLastError();
}

hipError_t gas_launch_scatter_params(hipStream_t stream, gas_params *table, const gas_params *upload, const uint32_t *slots, uint32_t n) {
	if (n == 0) {
		return hipSuccess;
	}
	const uint32_t threads = n * 8;
	hipLaunchKernelGGL(k_scatter_params, dim3((threads + 255) / 256), dim3(256), 0, stream, table, upload, slots, n);
	return hipGetLastError();
}

namespace {

// Measured HRIR sets come as M irregular (azimuth, elevation) positions; the library indexes directions on an
// azimuth x elevation grid (cell = elevation_index * az_steps + azimuth_index, the arithmetic of
// k_calc_spatialization).  One workgroup per grid cell: every thread scans a strided share of the positions keeping its
// three nearest (smallest chord |cell - position| on the unit sphere, ties to the smaller index: the chord, unlike the
// dot product, still resolves neighbours a fraction of a degree apart in f32), LDS-merges them into the
// workgroup's three nearest, then writes the cell's HRIR pair: the nearest one (interpolation 0) or the blend of the
// three weighted by 1 / angle (1).  NEW (AudioSpatializerHRTF, no reference counterpart): parity unpinned.
struct Near3 {
	float d[3];
	uint32_t i[3];
};

__device__ __forceinline__ void near3_insert(Near3 &n, float d, uint32_t i) {
	// keep (squared chord ascending, index ascending)
	if (d < n.d[2] || (d == n.d[2] && i < n.i[2])) {
		n.d[2] = d;
		n.i[2] = i;
		if (n.d[2] < n.d[1] || (n.d[2] == n.d[1] && n.i[2] < n.i[1])) {
			const float td = n.d[1];
			const uint32_t ti = n.i[1];
			n.d[1] = n.d[2];
			n.i[1] = n.i[2];
			n.d[2] = td;
			n.i[2] = ti;
			if (n.d[1] < n.d[0] || (n.d[1] == n.d[0] && n.i[1] < n.i[0])) {
				const float ud = n.d[0];
				const uint32_t ui = n.i[0];
				n.d[0] = n.d[1];
				n.i[0] = n.i[1];
				n.d[1] = ud;
				n.i[1] = ui;
			}
		}
	}
}

constexpr int REGRID_THREADS = 256;

__global__ __launch_bounds__(REGRID_THREADS) void k_hrtf_regrid(const float *__restrict__ positions /* [m][2] azimuth, elevation (radians) */, const float *__restrict__ hrir /* [m][2][taps] */, uint32_t m, uint32_t taps, uint32_t az_steps, uint32_t el_steps, int interpolation, float *__restrict__ out /* [az_steps * el_steps][2][GAS_HRTF_TAPS] */) {
#pragma clang fp contract(off)
	__shared__ float sd[REGRID_THREADS * 3];
	__shared__ uint32_t si[REGRID_THREADS * 3];
	__shared__ float best_w[3];
	__shared__ uint32_t best_i[3];
	const uint32_t cell = blockIdx.x, ai = cell % az_steps, ei = cell / az_steps;
	const float az = (float)ai * (6.28318530717958647692f / (float)az_steps);
	const float el = el_steps > 1 ? -1.57079632679489661923f + (float)ei * (3.14159265358979323846f / (float)(el_steps - 1)) : 0.0f;
	// unit vector, azimuth from -Z towards +X, elevation from the XZ plane (audio_spatializer_hrtf.cpp fill_pod)
	const float cx = cosf(el) * sinf(az), cy = sinf(el), cz = -cosf(el) * cosf(az);
	Near3 n{ { 8.0f, 8.0f, 8.0f }, { 0xffffffffu, 0xffffffffu, 0xffffffffu } };
	for (uint32_t k = threadIdx.x; k < m; k += REGRID_THREADS) {
		const float paz = positions[2 * k], pel = positions[2 * k + 1];
		const float dx = cosf(pel) * sinf(paz) - cx, dy = sinf(pel) - cy, dz = -cosf(pel) * cosf(paz) - cz;
		near3_insert(n, (dx * dx + dy * dy) + dz * dz, k);
	}
	for (int j = 0; j < 3; j++) {
		sd[threadIdx.x * 3 + j] = n.d[j];
		si[threadIdx.x * 3 + j] = n.i[j];
	}
	__syncthreads();
	if (threadIdx.x == 0) { // 768 candidates: a serial merge in the same (dot, index) order, deterministic
		Near3 g{ { 8.0f, 8.0f, 8.0f }, { 0xffffffffu, 0xffffffffu, 0xffffffffu } };
		for (int t = 0; t < REGRID_THREADS * 3; t++) {
			if (si[t] != 0xffffffffu) {
				near3_insert(g, sd[t], si[t]);
			}
		}
		float w[3] = { 1.0f, 0.0f, 0.0f };
		if (interpolation != 0 && g.i[1] != 0xffffffffu) {
			float sum = 0.0f;
			for (int j = 0; j < 3; j++) {
				const float angle = 2.0f * asinf(fminf(1.0f, 0.5f * sqrtf(g.d[j]))); // chord = 2 sin(angle / 2)
				w[j] = g.i[j] != 0xffffffffu ? 1.0f / (angle + 1e-4f) : 0.0f;
				sum += w[j];
			}
			for (int j = 0; j < 3; j++) {
				w[j] = w[j] / sum;
			}
		}
		for (int j = 0; j < 3; j++) {
			best_w[j] = w[j];
			best_i[j] = g.i[j];
		}
	}
	__syncthreads();
	for (uint32_t t = threadIdx.x; t < 2 * GAS_HRTF_TAPS; t += REGRID_THREADS) {
		const uint32_t ear = t / GAS_HRTF_TAPS, tap = t % GAS_HRTF_TAPS;
		float v = 0.0f;
		if (tap < taps) {
			for (int j = 0; j < 3; j++) {
				if (best_w[j] != 0.0f && best_i[j] != 0xffffffffu) {
					v += best_w[j] * hrir[((size_t)best_i[j] * 2 + ear) * taps + tap];
				}
			}
		}
		out[((size_t)cell * 2 + ear) * GAS_HRTF_TAPS + tap] = v;
	}
}

} // namespace

hipError_t gas_launch_hrtf_regrid(hipStream_t stream, const float *d_positions, const float *d_hrir, uint32_t m, uint32_t taps, uint32_t az_steps, uint32_t el_steps, int interpolation, float *d_out) {
	hipLaunchKernelGGL(k_hrtf_regrid, dim3(az_steps * el_steps), dim3(REGRID_THREADS), 0, stream, d_positions, d_hrir, m, taps, az_steps, el_steps, interpolation, d_out);
	return hipGetLastError();
}

hipError_t gas_launch_zero_slot(hipStream_t stream, const gas_dev_state &st, uint32_t slot, uint32_t hist_len, uint32_t er_ring_frames) {
	uint32_t work = hist_len > 8 ? hist_len : 8;
	if (st.er_ring && er_ring_frames > work) {
		work = er_ring_frames;
	}
	hipLaunchKernelGGL(k_zero_slot, dim3((work + 255) / 256), dim3(256), 0, stream, st, slot, hist_len, er_ring_frames);
	return hipGetLastError();
}
