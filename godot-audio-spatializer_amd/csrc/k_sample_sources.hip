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

// [ENGINE] AudioStreamPlaybackResampled::mix, one output frame at 16.16 position `off` (oracle: stream_mix_resampled):
// frames outside [start, len) read as zero.
#pragma clang fp contract(off)
__device__ __forceinline__ gas_audio_frame cubic_frame(const gas_cursor &c, uint32_t fmt, uint32_t ch, uint64_t off) {
	const int64_t q = (int64_t)(off >> 16);
	const float mu = (float)(uint32_t)(off & 0xFFFFu) / 65536.0f;
	gas_audio_frame y[4];
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const int64_t j = q - 3 + k;
		y[k] = (j >= (int64_t)c.start && j < (int64_t)c.frames) ? load_frame(c.pcm, fmt, ch, (uint64_t)j) : gas_audio_frame{ 0.0f, 0.0f };
	}
	const float mu2 = mu * mu;
	const float h11 = mu2 * (mu - 1);
	const float z = mu2 - h11;
	const float h01 = z - h11;
	const float h10 = mu - z;
	gas_audio_frame o;
	o.left = y[1].left + (y[2].left - y[1].left) * h01 + ((y[2].left - y[0].left) * h10 + (y[3].left - y[1].left) * h11) * 0.5f;
	o.right = y[1].right + (y[2].right - y[1].right) * h01 + ((y[2].right - y[0].right) * h10 + (y[3].right - y[1].right) * h11) * 0.5f;
	return o;
}

__global__ __launch_bounds__(256) void k_sample_sources(gas_cursor *__restrict__ cursors, const uint32_t *__restrict__ slots, uint32_t n, uint32_t F, const float *__restrict__ fade_env, gas_audio_frame *__restrict__ rows, const uint32_t *__restrict__ row_inc) {
	const int lane = threadIdx.x & 63;
	const uint32_t e = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (e >= n) {
		return;
	}
	gas_cursor *cp = cursors + slots[e];
	const gas_cursor c = *cp;
	gas_audio_frame *row = rows + (size_t)e * F;
	const uint32_t fmt = c.format_channels >> 8, ch = c.format_channels & 0xff;
	if (c.resampled && c.has_frames && c.pcm) {
		// The window the DSP sees is lookahead[64] ++ fresh[F] cut to F frames (audio_spatializer.cpp:367-378).  The fresh
		// frames are this call's outputs at positions fp_pos + i * inc; the lookahead is the previous call's last 64
		// outputs, regenerated from where and how fast that call ran.  The call reports as mixed the outputs produced
		// before the position's integer part first reaches the end of the stream.
		const uint64_t inc = row_inc ? row_inc[e] : 65536u;
		const uint64_t end_fp = c.frames << 16;
		uint64_t mixed64 = F;
		if (c.fp_pos >= end_fp) {
			mixed64 = 0;
		} else if (inc > 0) {
			const uint64_t need = (end_fp - c.fp_pos + inc - 1) / inc; // first i with fp_pos + i * inc >= end
			mixed64 = need < F ? need : F;
		}
		const uint32_t mixed = (uint32_t)mixed64;
		for (uint32_t i = lane; i < F; i += 64) {
			gas_audio_frame v{ 0.0f, 0.0f };
			if (mixed == F || i < mixed + GAS_LOOKAHEAD_BUFFER_SIZE) { // valid frames end at 64 + mixed
				if (i >= GAS_LOOKAHEAD_BUFFER_SIZE) {
					v = cubic_frame(c, fmt, ch, c.fp_pos + (uint64_t)(i - GAS_LOOKAHEAD_BUFFER_SIZE) * inc);
				} else if (c.resampled == 2) {
					v = cubic_frame(c, fmt, ch, c.fp_prev_pos + (uint64_t)(F - GAS_LOOKAHEAD_BUFFER_SIZE + i) * c.prev_inc);
				}
				if (mixed != F && i >= mixed) { // :389-392
					const float f = fade_env[i - mixed];
					v.left *= f;
					v.right *= f;
				}
			}
			row[i] = v;
		}
		if (lane == 0) {
			cp->fp_prev_pos = c.fp_pos;
			cp->prev_inc = (uint32_t)inc;
			cp->fp_pos = c.fp_pos + (uint64_t)F * inc; // the engine advances over all requested frames
			cp->resampled = 2;
			if (mixed != F) {
				cp->has_frames = 0; // :398
			}
		}
		return;
	}
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

hipError_t gas_launch_sample_sources(hipStream_t stream, gas_cursor *cursors, const uint32_t *slots, uint32_t n, uint32_t frames, const float *fade_env, gas_audio_frame *rows, const uint32_t *row_inc) {
	if (n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_sample_sources, dim3((n + 3) / 4), dim3(256), 0, stream, cursors, slots, n, frames, fade_env, rows, row_inc);
	return hipGetLastError();
}
