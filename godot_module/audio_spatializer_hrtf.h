// audio_spatializer_hrtf.h -- NEW resource AudioSpatializerHRTF and its instance (BASELINE north star: "plus a new
// AudioSpatializerHRTF resource"): the AudioSpatializer3D parameter maths with a 256-tap HRIR pair per direction
// convolved on the GPU.  Godot-side glue, see README.md; the arithmetic lives behind include/gas_amd.h.
#pragma once

#include "audio_spatializer_3d.h"

#include "gas_amd.h"
#include "gas_amd_host.h"

class AudioStreamPlaybackGasBatch;

// SpatializerPlaybackData (spatializer_parameters.h:69-71) for the per-instance path: the device slot that holds what
// SpatializerPlaybackData3D holds on the CPU (audio_spatializer_3d.h:85-99) plus the overlap-save history.
class SpatializerPlaybackDataGPU : public SpatializerPlaybackData {
	GDCLASS(SpatializerPlaybackDataGPU, SpatializerPlaybackData);

public:
	uint32_t slot = UINT32_MAX;
	~SpatializerPlaybackDataGPU(); // gas_source_free is legal from whichever thread drops the last reference (gas_amd.h)
};

class AudioSpatializerHRTF : public AudioSpatializer3D {
	GDCLASS(AudioSpatializerHRTF, AudioSpatializer3D);

	PackedFloat32Array hrir; // [directions][2 ears][taps], taps <= 256; with `positions` set: [positions][2 ears][taps]
	PackedVector2Array positions; // measured sets (SOFA SourcePosition): (azimuth, elevation) in DEGREES, azimuth counter-clockwise
	bool interpolate = true; // regridding: blend the three nearest measurements instead of taking the nearest
	int azimuth_steps = 32;
	int elevation_steps = 9;
	int taps = 256;
	bool batched = true;
	bool uploaded = false;

protected:
	static void _bind_methods();

public:
	void set_hrir(const PackedFloat32Array &p_hrir);
	PackedFloat32Array get_hrir() const { return hrir; }
	void set_positions(const PackedVector2Array &p_positions); // empty: hrir is already on the grid
	PackedVector2Array get_positions() const { return positions; }
	void set_interpolate(bool p_interpolate);
	bool get_interpolate() const { return interpolate; }
	void set_grid(int p_azimuth_steps, int p_elevation_steps);
	int get_azimuth_steps() const { return azimuth_steps; }
	int get_elevation_steps() const { return elevation_steps; }
	void set_taps(int p_taps);
	int get_taps() const { return taps; }
	void set_batched(bool p_batched) { batched = p_batched; }
	bool is_batched() const { return batched; }

	bool ensure_uploaded(); // gas_hrtf_load once per context
	virtual Ref<AudioSpatializerInstance> instantiate() override;
};

class AudioSpatializerInstanceHRTF : public AudioSpatializerInstance3D {
	GDCLASS(AudioSpatializerInstanceHRTF, AudioSpatializerInstance3D);
	friend class AudioSpatializerHRTF;

	Ref<AudioSpatializerHRTF> hrtf;
	gas_host *host = nullptr; // batched mode: the bus's shared host
	// batched mode, main/physics thread only: engine playback -> host playback id.  The Ref that keeps a playback alive
	// while the audio thread may still call its mix() is held by GasModule until the host reports the id released
	// (gas_host_set_release_fn) -- the role of the list node's Ref in the reference (audio_spatializer.cpp:538-547).
	HashMap<AudioStreamPlayback *, uint32_t> ids;
	bool paused = false; // set_playback_paused's last value: playbacks started while paused start paused

	// Per-instance mode: the POD of the latest parameters, built where update_spatializer_parameters runs (physics
	// thread: it walks the scene tree) and only copied on the audio thread.
	Mutex pod_mutex;
	gas_params latest_pod = {};
	bool have_pod = false;

	void fill_pod(const Ref<SpatializerParameters> &p_parameters, gas_params &r_pod) const;

public:
	virtual bool should_process_frames() const override { return true; }
	virtual bool should_mix_channels() const override { return false; }
	virtual Ref<SpatializerPlaybackData> instantiate_playback_data() override;
	// per-instance path: one launch per playback (README.md mode 1)
	virtual void process_frames(Ref<SpatializerParameters> p_parameters, Ref<SpatializerPlaybackData> p_playback_data, AudioFrame *p_output_buf, const AudioFrame *p_source_buf, int p_frame_count) override;

	// batched path (README.md mode 2; needs these five made virtual in audio_spatializer.h:120-138)
	void start_playback_stream_batched(Ref<AudioStreamPlayback> p_playback, float p_start_time);
	void stop_playback_stream_batched(Ref<AudioStreamPlayback> p_playback);
	bool is_playback_active_batched(Ref<AudioStreamPlayback> p_playback);
	void set_playback_paused_batched(bool p_paused); // audio_spatializer.cpp:115-122: the instance as a whole = each of its playbacks
	bool is_playback_paused_batched(); // :161-170
	float get_playback_position_batched(Ref<AudioStreamPlayback> p_playback); // :144-157
	void update_spatializer_parameters_batched();
	~AudioSpatializerInstanceHRTF();
};
