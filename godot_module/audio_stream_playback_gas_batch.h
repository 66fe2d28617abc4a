// audio_stream_playback_gas_batch.h -- the shared proxy: one AudioStreamPlayback per (bus, channel pair) that returns
// the batched mix of EVERY GPU playback routed to that bus.  It plays the part AudioStreamPlaybackSpatial plays for
// one instance (audio_spatializer.h:165-187, audio_spatializer.cpp:624-691).  Godot-side glue, see README.md.
#pragma once

#include "servers/audio/audio_stream.h"

#include "gas_amd_host.h"

class AudioStreamPlaybackGasBatch : public AudioStreamPlayback {
	GDCLASS(AudioStreamPlaybackGasBatch, AudioStreamPlayback);

	gas_host *host = nullptr;
	int channel = 0;
	SafeFlag active;

public:
	void setup(gas_host *p_host, int p_channel) {
		host = p_host;
		channel = p_channel;
	}

	virtual void start(double p_from_pos = 0.0) override { active.set(); }
	virtual void stop() override { active.clear(); }
	virtual bool is_playing() const override { return active.is_set(); }
	virtual int get_loop_count() const override { return 0; }
	virtual double get_playback_position() const override { return 0.0; }
	virtual void tag_used_streams() override {}

	// Audio thread.  The host mixes once per callback (first channel pair asked for again starts a new one, the
	// reference's latch audio_spatializer.cpp:494-508) and hands every pair its row; p_rate_scale is ignored like the
	// reference does (:681-691).
	virtual int mix(AudioFrame *p_buffer, float p_rate_scale, int p_frames) override;
};
