// k_misc.hip -- the small kernels around the two DSP kernels:
//   k_mix_reduce      final, deterministic sum of the per-workgroup partial mixes -> mix_buffer[c][F]
//                     (the serial `+=` of audio_spatializer.cpp:433-434,450-451 turned into a fixed-order tree;
//                     also the zeroing of :335-343 when there is nothing to add)
//   k_scatter_params  publish staged SpatializerParameters PODs into the slot-indexed table
//                     (set_spatializer_parameters, audio_spatializer.cpp:558-564)
//   k_zero_slot       fresh SpatializerPlaybackData for a (re)started playback (audio_spatializer.cpp:69)
#include "gas_internal.h"

namespace {

constexpr int RED_GROUPS = 16; // partial groups summed in parallel per output element
constexpr int RED_OUT = 16; // output elements per workgroup (256 threads = 16 x 16)

// out[c][i] = sum_p partials[c][p][i]; thread (o, gidx) sums partials gidx, gidx+16, ... in order,
// then the 16 group sums are added in fixed order by gidx 0.  Bitwise reproducible.
__global__ __launch_bounds__(256) void k_mix_reduce(const float *__restrict__ partials, uint32_t p_count, uint32_t p_stride, uint32_t elems /* F*2 */, float *__restrict__ out) {
	__shared__ float red[RED_GROUPS][RED_OUT + 1];
	const int o = threadIdx.x & (RED_OUT - 1);
	const int gidx = threadIdx.x / RED_OUT;
	const uint32_t i = blockIdx.x * RED_OUT + o;
	const uint32_t c = blockIdx.y;
	float s = 0.0f;
	if (i < elems) {
		const float *p = partials + (size_t)c * p_stride * elems + i;
		uint32_t k = gidx;
		// 4 independent loads in flight per trip
		for (; k + 3 * RED_GROUPS < p_count; k += 4 * RED_GROUPS) {
			float a0 = p[(size_t)k * elems];
			float a1 = p[(size_t)(k + RED_GROUPS) * elems];
			float a2 = p[(size_t)(k + 2 * RED_GROUPS) * elems];
			float a3 = p[(size_t)(k + 3 * RED_GROUPS) * elems];
			s += a0;
			s += a1;
			s += a2;
			s += a3;
		}
		for (; k < p_count; k += RED_GROUPS) {
			s += p[(size_t)k * elems];
		}
	}
	red[gidx][o] = s;
	__syncthreads();
	if (gidx == 0 && i < elems) {
		float t = 0.0f;
#pragma unroll
		for (int q = 0; q < RED_GROUPS; q++) {
			t += red[q][o];
		}
		out[(size_t)c * elems + i] = t;
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
		st.hrtf_prev_gain[slot] = 0.0f;
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
	dim3 grid((elems + RED_OUT - 1) / RED_OUT, channels);
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
