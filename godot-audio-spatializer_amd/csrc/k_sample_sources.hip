// k_sample_sources.hip -- SURVEY.md 8f#2: the source window of _mix_from_playback_list
// (audio_spatializer.cpp:367-408) produced on the device from HBM-resident PCM streams.
//
// The reference keeps a 64-frame lookahead per playback so it can fade out when a stream ends abruptly:
//   buf = lookahead[64] ++ fresh[F];  the DSP consumes buf[0..F);  lookahead = buf[F..F+64)      (:369-378,:401-403)
// With the whole stream resident that is simply a 64-frame delay:  row[i] = S[pos - 64 + i]  (zero before the playback's
// start frame: the zeroed initial lookahead :61-63), where pos counts the fresh frames handed out so far.  When the stream runs
// out inside a callback (mixed = len - pos < F) the last 64 valid frames row[mixed .. mixed+64) are scaled by
// the reference's envelope 0.96^(j+1) * (64 - j) / 64 (:380-396; table computed on the host with the same f32
// recurrence), everything after is zero, and has_frames clears (:398); afterwards the row is all zeros (:405-408).
// One wave per playback row, lane-contiguous 8-byte stores; int16 -> float as s / 32768, mono feeds both ears.
#include "gas_internal.h"

namespace {

__device__ __forceinline__ gas_audio_frame load_frame(const void *pcm, uint32_t fmt, uint32_t ch, uint64_t idx) {
	if (fmt == GAS_PCM_S16) {
		const int16_t *p = static_cast<const int16_t *>(pcm);
		if (ch == 1) {
			const float v = (float)p[idx] / 32768.0f;
			return gas_audio_frame{ v, v };
		}
		const short2 s = reinterpret_cast<const short2 *>(p)[idx];
		return gas_audio_frame{ (float)s.x / 32768.0f, (float)s.y / 32768.0f };
	}
	const float *p = static_cast<const float *>(pcm);
	if (ch == 1) {
		const float v = p[idx];
		return gas_audio_frame{ v, v };
	}
	const float2 s = reinterpret_cast<const float2 *>(p)[idx];
	return gas_audio_frame{ s.x, s.y };
}

__global__ __launch_bounds__(256) void k_sample_sources(gas_cursor *__restrict__ cursors, const uint32_t *__restrict__ slots, uint32_t n, uint32_t F, const float *__restrict__ fade_env, gas_audio_frame *__restrict__ rows) {
	const int lane = threadIdx.x & 63;
	const uint32_t e = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (e >= n) {
		return;
	}
	gas_cursor *cp = cursors + slots[e];
	const gas_cursor c = *cp;
	gas_audio_frame *row = rows + (size_t)e * F;
	const uint32_t fmt = c.format_channels >> 8, ch = c.format_channels & 0xff;
	uint32_t mixed = 0;
	if (c.has_frames && c.pcm) {
		const uint64_t left = c.frames > c.pos ? c.frames - c.pos : 0;
		mixed = left < F ? (uint32_t)left : F; // [ENGINE] AudioStreamPlayback::mix return value
	}
	for (uint32_t i = lane; i < F; i += 64) {
		gas_audio_frame v{ 0.0f, 0.0f };
		if (c.has_frames && c.pcm) {
			const int64_t si = (int64_t)c.pos - GAS_LOOKAHEAD_BUFFER_SIZE + (int64_t)i;
			if (mixed == F) {
				if (si >= (int64_t)c.start) {
					v = load_frame(c.pcm, fmt, ch, (uint64_t)si);
				}
			} else if (i < mixed + GAS_LOOKAHEAD_BUFFER_SIZE) { // valid frames end at 64 + mixed
				if (si >= (int64_t)c.start) {
					v = load_frame(c.pcm, fmt, ch, (uint64_t)si);
				}
				if (i >= mixed) { // :389-392
					const float f = fade_env[i - mixed];
					v.left *= f;
					v.right *= f;
				}
			} // else: buf[idx] *= 0.0 (:394) over the zero-filled tail
		}
		row[i] = v;
	}
	if (lane == 0 && c.has_frames) {
		cp->pos = c.pos + mixed;
		if (mixed != F) {
			cp->has_frames = 0; // :398
		}
	}
}

} // namespace

hipError_t gas_launch_sample_sources(hipStream_t stream, gas_cursor *cursors, const uint32_t *slots, uint32_t n, uint32_t frames, const float *fade_env, gas_audio_frame *rows) {
	if (n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_sample_sources, dim3((n + 3) / 4), dim3(256), 0, stream, cursors, slots, n, frames, fade_env, rows);
	return hipGetLastError();
}
