// audio_stream_playback_gas_batch.cpp -- see the header.
#include "audio_stream_playback_gas_batch.h"

int AudioStreamPlaybackGasBatch::mix(AudioFrame *p_buffer, float p_rate_scale, int p_frames) {
	if (!active.is_set() || host == nullptr) {
		return 0; // audio_spatializer.cpp:682-685
	}
	static_assert(sizeof(AudioFrame) == sizeof(gas_audio_frame), "AudioFrame is two f32");
	const int rc = gas_host_get_mixed_frames(host, channel, reinterpret_cast<gas_audio_frame *>(p_buffer), p_frames);
	if (rc != GAS_OK) {
		ERR_PRINT_ONCE(vformat("gas_host_get_mixed_frames: %s", gas_strerror(rc)));
		return 0;
	}
	return p_frames; // :686-690
}
