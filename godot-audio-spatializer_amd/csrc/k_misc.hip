// k_misc.hip -- the small kernels around the two DSP kernels:
//   k_mix_reduce      final, deterministic sum of the per-workgroup partial mixes -> mix_buffer[c][F]
//                     (the serial `+=` of audio_spatializer.cpp:433-434,450-451 turned into a fixed-order tree;
//                     also the zeroing of :335-343 when there is nothing to add)
//   k_scatter_params  publish staged SpatializerParameters PODs into the slot-indexed table
//                     (set_spatializer_parameters, audio_spatializer.cpp:558-564)
//   k_zero_slot       fresh SpatializerPlaybackData for a (re)started playback (audio_spatializer.cpp:69)
#include "gas_internal.h"

namespace {

constexpr int RED_ROWS = 64; // partial rows summed in parallel per output column
constexpr int RED_COLS = 4; // float4 columns (16 output floats) per workgroup; 256 threads = 64 x 4

// out[c][i] = sum_p partials[c][p][i].  Thread (prow, col) adds partial rows prow, prow+64, ... of its float4
// column (four independent 16-byte loads in flight per trip), then the 64 row sums are added in fixed order.
// Bitwise reproducible; no atomics.
__global__ __launch_bounds__(256) void k_mix_reduce(const float *__restrict__ partials, uint32_t p_count, uint32_t p_stride, uint32_t elems /* F*2, multiple of 4 */, float *__restrict__ out) {
	__shared__ float4 red[RED_ROWS][RED_COLS];
	const int col = threadIdx.x & (RED_COLS - 1);
	const int prow = threadIdx.x / RED_COLS;
	const uint32_t i4 = blockIdx.x * RED_COLS + col; // float4 index within a partial
	const uint32_t c = blockIdx.y;
	const uint32_t e4 = elems / 4;
	float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
	if (i4 < e4) {
		const float4 *p = reinterpret_cast<const float4 *>(partials + (size_t)c * p_stride * elems) + i4;
		uint32_t k = prow;
		for (; k + 3 * RED_ROWS < p_count; k += 4 * RED_ROWS) {
			const float4 a0 = p[(size_t)k * e4];
			const float4 a1 = p[(size_t)(k + RED_ROWS) * e4];
			const float4 a2 = p[(size_t)(k + 2 * RED_ROWS) * e4];
			const float4 a3 = p[(size_t)(k + 3 * RED_ROWS) * e4];
			s.x += a0.x; s.y += a0.y; s.z += a0.z; s.w += a0.w;
			s.x += a1.x; s.y += a1.y; s.z += a1.z; s.w += a1.w;
			s.x += a2.x; s.y += a2.y; s.z += a2.z; s.w += a2.w;
			s.x += a3.x; s.y += a3.y; s.z += a3.z; s.w += a3.w;
		}
		for (; k < p_count; k += RED_ROWS) {
			const float4 a = p[(size_t)k * e4];
			s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
		}
	}
	red[prow][col] = s;
	__syncthreads();
	if (prow == 0 && i4 < e4) {
		float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
		for (int q = 0; q < RED_ROWS; q++) {
			const float4 a = red[q][col];
			t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
		}
		reinterpret_cast<float4 *>(out + (size_t)c * elems)[i4] = t;
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

} // namespace

hipError_t gas_launch_noop(hipStream_t stream) {
	hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, stream);
	return hipGetLastError();
}

hipError_t gas_launch_mix_reduce(hipStream_t stream, const float *partials, uint32_t p_count, uint32_t p_stride, uint32_t channels, uint32_t frames, gas_audio_frame *out) {
	const uint32_t elems = frames * 2;
	dim3 grid((elems / 4 + RED_COLS - 1) / RED_COLS, channels);
	hipLaunchKernelGGL(k_mix_reduce, grid, dim3(256), 0, stream, partials, p_count, p_stride, elems, reinterpret_cast<float *>(out));
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

hipError_t gas_launch_zero_slot(hipStream_t stream, const gas_dev_state &st, uint32_t slot, uint32_t hist_len, uint32_t er_ring_frames) {
	uint32_t work = hist_len > 8 ? hist_len : 8;
	if (st.er_ring && er_ring_frames > work) {
		work = er_ring_frames;
	}
	hipLaunchKernelGGL(k_zero_slot, dim3((work + 255) / 256), dim3(256), 0, stream, st, slot, hist_len, er_ring_frames);
	return hipGetLastError();
}
