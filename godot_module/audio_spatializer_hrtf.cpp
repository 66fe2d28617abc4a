// audio_spatializer_hrtf.cpp -- see the header.
#include "audio_spatializer_hrtf.h"

#include "audio_stream_player_spatial.h"
#include "scene/3d/audio_listener_3d.h"
#include "scene/3d/camera_3d.h"
#include "scene/main/viewport.h"
#include "gas_module.h"

SpatializerPlaybackDataGPU::~SpatializerPlaybackDataGPU() {
	if (slot != UINT32_MAX && GasModule::ctx()) {
		gas_source_free(GasModule::ctx(), slot); // takes effect at the next block boundary (deferred delete, audio_spatializer.cpp:538-547)
	}
}

void AudioSpatializerHRTF::set_hrir(const PackedFloat32Array &p_hrir) {
	hrir = p_hrir;
	uploaded = false;
}

void AudioSpatializerHRTF::set_positions(const PackedVector2Array &p_positions) {
	positions = p_positions;
	uploaded = false;
}

void AudioSpatializerHRTF::set_interpolate(bool p_interpolate) {
	interpolate = p_interpolate;
	uploaded = false;
}

void AudioSpatializerHRTF::set_grid(int p_azimuth_steps, int p_elevation_steps) {
	ERR_FAIL_COND(p_azimuth_steps < 1 || p_elevation_steps < 1);
	azimuth_steps = p_azimuth_steps;
	elevation_steps = p_elevation_steps;
	uploaded = false;
}

void AudioSpatializerHRTF::set_taps(int p_taps) {
	ERR_FAIL_COND(p_taps < 1 || p_taps > GAS_HRTF_TAPS);
	taps = p_taps;
	uploaded = false;
}

bool AudioSpatializerHRTF::ensure_uploaded() {
	if (uploaded) {
		return true;
	}
	gas_ctx *c = GasModule::ctx();
	if (!c) {
		return false;
	}
	const int dirs = azimuth_steps * elevation_steps;
	int rc;
	if (positions.size() > 0) {
		// a measured set (what a SOFA file holds): regridded on the device onto azimuth_steps x elevation_steps
		const int m = positions.size();
		ERR_FAIL_COND_V_MSG(hrir.size() != m * 2 * taps, false, "AudioSpatializerHRTF: hrir must hold positions * 2 * taps floats.");
		Vector<float> rad;
		rad.resize(m * 2);
		for (int i = 0; i < m; i++) {
			rad.write[2 * i] = -Math::deg_to_rad(positions[i].x); // SOFA azimuth is counter-clockwise; the library's runs towards +X
			rad.write[2 * i + 1] = Math::deg_to_rad(positions[i].y);
		}
		rc = gas_hrtf_load_positions(c, rad.ptr(), hrir.ptr(), m, taps, azimuth_steps, elevation_steps, interpolate ? 1 : 0, nullptr);
	} else {
		ERR_FAIL_COND_V_MSG(hrir.size() != dirs * 2 * taps, false, "AudioSpatializerHRTF: hrir must hold azimuth_steps * elevation_steps * 2 * taps floats.");
		rc = gas_hrtf_load(c, hrir.ptr(), dirs, taps);
	}
	ERR_FAIL_COND_V_MSG(rc != GAS_OK, false, gas_strerror(rc));
	uploaded = true;
	return true;
}

Ref<AudioSpatializerInstance> AudioSpatializerHRTF::instantiate() {
	if (!ensure_uploaded()) {
		// no GPU / no HRIR set: the stock 3D spatializer (audio_spatializer_3d.cpp:645-652) -- a different sound, so say so
		WARN_PRINT_ONCE("AudioSpatializerHRTF: no usable GPU context or HRIR set; playing through AudioSpatializer3D (pan + distance filter, no HRTF).");
		return AudioSpatializer3D::instantiate();
	}
	Ref<AudioSpatializerInstanceHRTF> ins;
	ins.instantiate();
	ins->base = Ref<AudioSpatializer3D>(this);
	ins->mix_channel_mode = false;
	ins->hrtf = Ref<AudioSpatializerHRTF>(this);
	return ins;
}

void AudioSpatializerHRTF::_bind_methods() {
	ClassDB::bind_method(D_METHOD("set_hrir", "hrir"), &AudioSpatializerHRTF::set_hrir);
	ClassDB::bind_method(D_METHOD("get_hrir"), &AudioSpatializerHRTF::get_hrir);
	ClassDB::bind_method(D_METHOD("set_grid", "azimuth_steps", "elevation_steps"), &AudioSpatializerHRTF::set_grid);
	ClassDB::bind_method(D_METHOD("get_azimuth_steps"), &AudioSpatializerHRTF::get_azimuth_steps);
	ClassDB::bind_method(D_METHOD("get_elevation_steps"), &AudioSpatializerHRTF::get_elevation_steps);
	ClassDB::bind_method(D_METHOD("set_taps", "taps"), &AudioSpatializerHRTF::set_taps);
	ClassDB::bind_method(D_METHOD("get_taps"), &AudioSpatializerHRTF::get_taps);
	ClassDB::bind_method(D_METHOD("set_batched", "batched"), &AudioSpatializerHRTF::set_batched);
	ClassDB::bind_method(D_METHOD("is_batched"), &AudioSpatializerHRTF::is_batched);
	ClassDB::bind_method(D_METHOD("set_positions", "positions"), &AudioSpatializerHRTF::set_positions);
	ClassDB::bind_method(D_METHOD("get_positions"), &AudioSpatializerHRTF::get_positions);
	ClassDB::bind_method(D_METHOD("set_interpolate", "interpolate"), &AudioSpatializerHRTF::set_interpolate);
	ClassDB::bind_method(D_METHOD("get_interpolate"), &AudioSpatializerHRTF::get_interpolate);
	ADD_PROPERTY(PropertyInfo(Variant::PACKED_FLOAT32_ARRAY, "hrir"), "set_hrir", "get_hrir");
	ADD_PROPERTY(PropertyInfo(Variant::PACKED_VECTOR2_ARRAY, "positions"), "set_positions", "get_positions");
	ADD_PROPERTY(PropertyInfo(Variant::BOOL, "interpolate"), "set_interpolate", "get_interpolate");
	ADD_PROPERTY(PropertyInfo(Variant::INT, "taps", PROPERTY_HINT_RANGE, "1,256,1"), "set_taps", "get_taps");
	ADD_PROPERTY(PropertyInfo(Variant::BOOL, "batched"), "set_batched", "is_batched");
}

// SpatializerParameters3D (audio_spatializer_3d.h:61-83) -> the 128-byte POD of include/gas_amd.h.  The HRTF gain is the
// loudest channel pair's volume (what AudioServer would have applied, audio_spatializer.cpp:314-318).  The direction is
// the nearest cell of the HRIR grid towards the listener, the same cell arithmetic gas_calc_spatialization uses when it
// generates parameters on the device (azimuth from -Z towards +X, elevation from the XZ plane).
void AudioSpatializerInstanceHRTF::fill_pod(const Ref<SpatializerParameters> &p_parameters, gas_params &r_pod) const {
	const Ref<SpatializerParameters3D> p3 = p_parameters;
	const Vector<Vector2> v = p_parameters->get_mix_volumes(); // size 4, spatializer_parameters.cpp:45
	float loudest = 0.0f;
	for (int c = 0; c < 4; c++) {
		r_pod.mix_volumes[c][0] = v[c].x;
		r_pod.mix_volumes[c][1] = v[c].y;
		loudest = MAX(loudest, MAX(v[c].x, v[c].y));
	}
	r_pod.pitch_scale = p_parameters->get_pitch_scale();
	r_pod.update_parameters = p_parameters->should_update_parameters();
	if (p3.is_valid()) {
		r_pod.linear_attenuation = p3->get_linear_attenuation();
		r_pod.attenuation_filter_cutoff_hz = p3->get_attenuation_filter_cutoff_hz();
	}
	r_pod.hrtf_gain = loudest;

	// player position in the space of the viewport's listener (the transform calculate_spatialization builds at
	// audio_spatializer_3d.cpp:326-340)
	Vector3 local = Vector3(0, 0, -1);
	const Node3D *player = Object::cast_to<Node3D>(get_audio_player());
	if (player && player->is_inside_tree()) {
		const Viewport *vp = player->get_viewport();
		const Node3D *listener = vp->get_audio_listener_3d() ? (const Node3D *)vp->get_audio_listener_3d() : (const Node3D *)vp->get_camera_3d();
		if (listener) {
			local = listener->get_global_transform().orthonormalized().affine_inverse().xform(player->get_global_transform().origin);
		}
	}
	const int n_az = hrtf->get_azimuth_steps(), n_el = hrtf->get_elevation_steps();
	const double az = Math::atan2((double)local.x, (double)-local.z);
	const double el = Math::atan2((double)local.y, Math::sqrt((double)local.x * local.x + (double)local.z * local.z));
	const int ai = (int)Math::posmod((int64_t)Math::round(az / Math_TAU * n_az), (int64_t)n_az);
	const int ei = n_el > 1 ? CLAMP((int)Math::round((el + Math_PI / 2) / Math_PI * (n_el - 1)), 0, n_el - 1) : 0;
	r_pod.hrtf_dir = (uint32_t)(ei * n_az + ai);
}

Ref<SpatializerPlaybackData> AudioSpatializerInstanceHRTF::instantiate_playback_data() {
	Ref<SpatializerPlaybackDataGPU> d;
	d.instantiate();
	static const int32_t chain[1] = { GAS_FX_HRTF };
	const int rc = gas_source_alloc(GasModule::ctx(), GAS_KIND_EFFECT, chain, 1, &d->slot);
	ERR_FAIL_COND_V_MSG(rc != GAS_OK, Ref<SpatializerPlaybackData>(), gas_strerror(rc));
	return d;
}

void AudioSpatializerInstanceHRTF::process_frames(Ref<SpatializerParameters> p_parameters, Ref<SpatializerPlaybackData> p_playback_data, AudioFrame *p_output_buf, const AudioFrame *p_source_buf, int p_frame_count) {
	SpatializerPlaybackDataGPU *g = Object::cast_to<SpatializerPlaybackDataGPU>(*p_playback_data);
	ERR_FAIL_NULL(g);
	// Audio thread: no scene-tree access here.  The POD was built by update_spatializer_parameters_batched /
	// _per_instance on the physics thread (the reference hands process_frames parameters that calculate_spatialization
	// already computed there, audio_spatializer.cpp:558-574); p_parameters is the same snapshot and is not walked again.
	gas_params pod;
	{
		MutexLock lock(pod_mutex);
		if (!have_pod) {
			for (int i = 0; i < p_frame_count; i++) {
				p_output_buf[i] = AudioFrame(0, 0); // nothing computed yet: audio_spatializer.cpp:330
			}
			return;
		}
		pod = latest_pod;
	}
	gas_params_publish(GasModule::ctx(), g->slot, &pod);
	const int rc = gas_process_frames_1(GasModule::ctx(), g->slot, reinterpret_cast<gas_audio_frame *>(p_output_buf), reinterpret_cast<const gas_audio_frame *>(p_source_buf), p_frame_count);
	ERR_FAIL_COND_MSG(rc != GAS_OK, gas_strerror(rc));
}

// ---- batched mode --------------------------------------------------------------------------------------------------

static int engine_stream_mix(void *p_user, gas_audio_frame *p_buffer, float p_rate_scale, int p_frames) {
	// [ENGINE] the sampler stays on the CPU here; audio thread only (gas_amd_host.h)
	return static_cast<AudioStreamPlayback *>(p_user)->mix(reinterpret_cast<AudioFrame *>(p_buffer), p_rate_scale, p_frames);
}

void AudioSpatializerInstanceHRTF::start_playback_stream_batched(Ref<AudioStreamPlayback> p_playback, float p_start_time) {
	ERR_FAIL_COND(p_playback.is_null());
	static const int32_t chain[1] = { GAS_FX_HRTF };
	if (!host) {
		host = GasModule::host_for(get_audio_player()->get_bus(), GAS_KIND_EFFECT, chain, 1);
		ERR_FAIL_NULL(host);
	}
	p_playback->start(p_start_time); // audio_spatializer.cpp:55-57
	uint32_t id = 0;
	GasModule::retain(p_playback); // alive until the host has finished with it, whoever drops their reference first
	const int rc = gas_host_start_playback(host, engine_stream_mix, p_playback.ptr(), &id);
	if (rc != GAS_OK) {
		GasModule::unretain(p_playback.ptr());
		ERR_FAIL_MSG(gas_strerror(rc));
	}
	ids.insert(p_playback.ptr(), id);
	if (paused) {
		gas_host_set_playback_paused(host, id, 1);
	}
	update_spatializer_parameters_batched(); // parameters exist before the first callback (audio_stream_player_spatial.cpp:76-79)
}

void AudioSpatializerInstanceHRTF::stop_playback_stream_batched(Ref<AudioStreamPlayback> p_playback) {
	if (uint32_t *id = ids.getptr(p_playback.ptr())) {
		gas_host_stop_playback(host, *id); // audio_spatializer.cpp:98-113: the audio thread reaps it; GasModule's Ref stays until then
		ids.erase(p_playback.ptr());
	}
}

bool AudioSpatializerInstanceHRTF::is_playback_active_batched(Ref<AudioStreamPlayback> p_playback) {
	const uint32_t *id = ids.getptr(p_playback.ptr());
	return id && gas_host_is_playback_active(host, *id);
}

void AudioSpatializerInstanceHRTF::set_playback_paused_batched(bool p_paused) {
	// The reference pauses this instance's proxies on AudioServer (audio_spatializer.cpp:115-122: "paused applies to the
	// spatializer as a whole", called by the player at audio_stream_player_spatial.cpp:59,64,103,115).  The batched proxy
	// is shared by every player of the bus, so the pause goes to each playback of THIS instance instead.
	paused = p_paused;
	for (const KeyValue<AudioStreamPlayback *, uint32_t> &kv : ids) {
		gas_host_set_playback_paused(host, kv.value, p_paused);
	}
}

bool AudioSpatializerInstanceHRTF::is_playback_paused_batched() {
	// audio_spatializer.cpp:161-170: false without an active playback, otherwise the (common) pause state
	for (const KeyValue<AudioStreamPlayback *, uint32_t> &kv : ids) {
		if (gas_host_is_playback_active(host, kv.value)) {
			return gas_host_is_playback_paused(host, kv.value) != 0;
		}
	}
	return false;
}

float AudioSpatializerInstanceHRTF::get_playback_position_batched(Ref<AudioStreamPlayback> p_playback) {
	// audio_spatializer.cpp:144-157 forwards to the stream playback; so does this (its position is advanced by the mix()
	// calls the host makes on the audio thread; audio_stream_player_spatial.cpp:347,375,384).  The host's own counter
	// (gas_host_get_playback_position, frames consumed) serves device-stream playbacks, which have no engine playback.
	if (!ids.has(p_playback.ptr())) {
		return 0.0f; // :152-155
	}
	return p_playback->get_playback_position();
}

void AudioSpatializerInstanceHRTF::update_spatializer_parameters_batched() {
	Ref<SpatializerParameters> p = calculate_spatialization(); // AudioSpatializerInstance3D's, unchanged (audio_spatializer_3d.cpp:277-489)
	ERR_FAIL_COND(p.is_null());
	gas_params pod = {};
	fill_pod(p, pod); // physics thread: the scene-tree reads happen here, never on the audio thread
	{
		MutexLock lock(pod_mutex); // per-instance mode copies it on the audio thread
		latest_pod = pod;
		have_pod = true;
	}
	if (!host) {
		return;
	}
	gas_host_collect_released(host); // playbacks the audio thread has finished with: GasModule drops their Refs
	for (const KeyValue<AudioStreamPlayback *, uint32_t> &kv : ids) {
		gas_host_set_spatializer_parameters(host, kv.value, &pod); // audio_spatializer.cpp:558-564, per playback
	}
}

AudioSpatializerInstanceHRTF::~AudioSpatializerInstanceHRTF() {
	for (const KeyValue<AudioStreamPlayback *, uint32_t> &kv : ids) {
		gas_host_stop_playback(host, kv.value); // GasModule keeps the playbacks alive until the audio thread has let go
	}
}
