/*
 * gas_oracle.h -- CPU restatement ("oracle") of the godot-audio-spatializer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product path
 * (godot-audio-spatializer_amd/, include/) may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 *  and cannot be compiled here (every translation unit
 * includes un-vendored Godot engine headers).  The engine
 * primitives (AudioFrame, AudioFilterSW, Math::db_to_linear) are restated from
 * the published godotengine/godot algorithm as recorded in SURVEY.md Appendix B;
 * no engine version is pinned by the reference.  Independent cross-checks that
 * share no code with this file live in tests/test_oracle_*.py (scipy lfilter,
 * RBJ identities, closed forms, float64 numpy convolution).
 *
 * All file:line citations are relative to the reference tree /root/reference/.
 */
#ifndef GAS_ORACLE_H
#define GAS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* audio_spatializer.h:47-52 */
#define GASO_MAX_CHANNELS_PER_BUS 4
#define GASO_LOOKAHEAD 64
#define GASO_MAX_BUSES_PER_PLAYBACK 6

/* NEW (no reference code): HRTF / early-reflection effect sizes. */
#define GASO_HRTF_TAPS 256
#define GASO_ER_TAPS 8
#define GASO_MAX_EFFECTS 4

/* [ENGINE] core/math/audio_frame.h: two interleaved f32. */
typedef struct gaso_frame {
	float l, r;
} gaso_frame;

/* [ENGINE] AudioFilterSW::Coeffs (a1,a2 are stored negated, Appendix B). */
typedef struct gaso_coeffs {
	float a1, a2, b0, b1, b2;
} gaso_coeffs;

/* [ENGINE] AudioFilterSW::Processor: all zero-initialised. */
typedef struct gaso_processor {
	gaso_coeffs coeffs;
	gaso_coeffs incr;
	float ha1, ha2, hb1, hb2;
} gaso_processor;

/* spatializer_parameters.h:39-67 + audio_spatializer_3d.h:61-83 numeric payload,
 * plus the NEW effect parameters (HRTF gain/direction, early reflections,
 * AudioEffectHighShelfFilter gain/cutoff as pushed by _process_effects,
 * gd_spatializer_instance.gd:125-127). */
typedef struct gaso_params {
	float mix_volumes[GASO_MAX_CHANNELS_PER_BUS][2];
	float pitch_scale;
	float linear_attenuation;
	float attenuation_filter_cutoff_hz;
	uint32_t update_parameters;
	float hrtf_gain;
	uint32_t hrtf_dir;
	float fx_shelf_gain;
	float fx_shelf_cutoff_hz;
	float er_gain[GASO_ER_TAPS];
	uint32_t er_delay[GASO_ER_TAPS];
} gaso_params;

/* audio_spatializer_3d.h:85-99 SpatializerPlaybackData3D. */
typedef struct gaso_pdata3d {
	float prev_mix_volumes[GASO_MAX_CHANNELS_PER_BUS][2];
	int32_t prev_count; /* Vector<Vector2>::size(), audio_spatializer_3d.cpp:873-885 */
	gaso_processor filter_processors[8];
} gaso_pdata3d;

/* Effect kinds for the AudioSpatializerEffect chain (audio_spatializer_effect.cpp:52-76). */
enum {
	GASO_FX_HIGHSHELF = 1, /* [ENGINE] AudioEffectHighShelfFilter, FILTER_6DB (gd_spatializer.gd:14-19) */
	GASO_FX_EARLY_REFLECTIONS = 2, /* NEW */
	GASO_FX_HRTF = 3, /* NEW */
	/* audio_spatializer_effect.cpp:79-88 instantiates any AudioEffect: the engine's other one-biquad filters
	 * ([ENGINE] AudioEffectFilter subclasses, FILTER_6DB) and AudioEffectAmplify */
	GASO_FX_LOWPASS = 4,
	GASO_FX_HIGHPASS = 5,
	GASO_FX_BANDPASS = 6,
	GASO_FX_NOTCH = 7,
	GASO_FX_LOWSHELF = 8,
	GASO_FX_AMPLIFY = 9,
};

/* audio_spatializer_effect.h:68-76 SpatializerPlaybackDataEffect: one effect
 * instance (state) per effect per playback. */
typedef struct gaso_fx_state {
	/* HIGHSHELF: [ENGINE] AudioEffectFilterInstance filter_process[2][4], only [ear][0] used at 1 stage */
	gaso_processor shelf[2];
	/* EARLY_REFLECTIONS: ring of past input frames */
	gaso_frame *ring;
	uint32_t ring_frames;
	uint32_t ring_pos;
	/* HRTF: last TAPS-1 gained mono samples + previous gain */
	float hist[GASO_HRTF_TAPS - 1];
	float prev_gain;
	int32_t prev_dir_plus1; /* 0 = no previous callback yet; else previous hrtf_dir + 1 (cross-fade) */
	/* LOWPASS .. LOWSHELF: the settings the instance reads from its resource every block ([ENGINE] AudioEffectFilter:
	 * cutoff 2000, resonance 0.5, gain 1 by default); the processors are shelf[] above */
	float cutoff_hz, resonance, gain;
	/* AMPLIFY: [ENGINE] AudioEffectAmplify::volume_db, the instance's mix_volume_db and whether a block has run (the
	 * instance is created with mix_volume_db = volume_db) */
	float volume_db, amp_mix_volume_db;
	int32_t amp_started;
} gaso_fx_state;

typedef struct gaso_pdata_effect {
	int32_t n_effects;
	int32_t kinds[GASO_MAX_EFFECTS];
	gaso_fx_state fx[GASO_MAX_EFFECTS];
} gaso_pdata_effect;

/* HRIR table: [dirs][2 ears][GASO_HRTF_TAPS] f32. */
typedef struct gaso_hrtf {
	const float *hrir;
	uint32_t dirs;
	int32_t impl; /* 0: direct-form f64 FIR (the checker); 1: overlap-save radix-2 FFT in f32 (the CPU baseline) */
	float *spec; /* impl 1 only: cached HRIR spectra [dirs][2 ears][re|im][spec_len], built by gaso_hrtf_prepare */
	int32_t spec_len;
	int32_t crossfade; /* NEW (8f#4): when a playback's direction changed since its previous callback, blend the old
	                      and the new HRIR's outputs with t = i/n across the block (the volume lerp's analogue,
	                      audio_spatializer_3d.cpp:591-592) instead of switching at the block boundary */
} gaso_hrtf;

/* ---- engine primitives ------------------------------------------------ */
float gaso_db_to_linear(float db);
float gaso_linear_to_db(float lin);
void gaso_highshelf_coeffs(double sampling_rate, double cutoff, double resonance, double gain, int stages, gaso_coeffs *out);
/* the other AudioFilterSW modes (kind = GASO_FX_LOWPASS .. GASO_FX_LOWSHELF), one stage */
void gaso_filter_coeffs(int kind, double sampling_rate, double cutoff, double resonance, double gain, gaso_coeffs *out);
void gaso_processor_update_coeffs(gaso_processor *p, const gaso_coeffs *target, int interp_len);
float gaso_processor_process_one(gaso_processor *p, float x);
float gaso_processor_process_one_interp(gaso_processor *p, float x);

/* ---- AudioSpatializerInstance3D (audio_spatializer_3d.cpp:491-609) ---- */
void gaso_process_frames_3d(const gaso_params *params, gaso_pdata3d *pd, gaso_frame *out, const gaso_frame *src, int n, float mix_rate);
void gaso_mix_channel_3d(const gaso_params *params, gaso_pdata3d *pd, int channel, gaso_frame *out, const gaso_frame *src, int n, float mix_rate);

/* ---- AudioSpatializerInstanceEffect (audio_spatializer_effect.cpp:33-77) ---- */
void gaso_fx_process(int kind, const gaso_params *params, gaso_fx_state *st, const gaso_hrtf *hrtf, const gaso_frame *src, gaso_frame *dst, int n, float mix_rate);
/* Returns a bitmask trace of the ping-pong choices for tests: bit (2j) = dst is temp, bit (2j+1) = src is temp. */
uint32_t gaso_process_frames_effect(const gaso_params *params, gaso_pdata_effect *pd, const gaso_hrtf *hrtf, gaso_frame *out, const gaso_frame *src, int n, gaso_frame *temp, float mix_rate);

/* ---- AudioSpatializerInstance mixer (audio_spatializer.cpp:326-527) ---- */
enum {
	GASO_KIND_3D_MIX = 0, /* AudioSpatializer3D, mix_channel_mode = true  (audio_spatializer_3d.h:145-146) */
	GASO_KIND_3D_PROCESS = 1, /* AudioSpatializer3D, mix_channel_mode = false */
	GASO_KIND_EFFECT = 2, /* AudioSpatializerEffect (audio_spatializer_effect.h:57-58) */
};

/* audio_spatializer.h:55-66 SpatialPlaybackListNode with a synthetic stream
 * standing in for [ENGINE] AudioStreamPlayback::mix.  resampled == 0: a playback class
 * that hands its frames out as they are (rate_scale ignored); resampled != 0:
 * [ENGINE] AudioStreamPlaybackResampled::mix -- 16.16 fixed-point position advanced by
 * rate_scale per output frame, 4-point cubic (Hermite) interpolation over the frames
 * q-3 .. q (recollection of the engine source, parity unpinned). */
typedef struct gaso_playback {
	const gaso_frame *stream; /* stream_frames frames */
	int64_t stream_frames;
	int64_t stream_pos;
	int32_t active;
	int32_t has_frames;
	gaso_frame lookahead[GASO_LOOKAHEAD];
	gaso_pdata3d pd3d;
	gaso_pdata_effect pdfx;
	float last_peak[2];
	int32_t resampled;
	int32_t paused; /* NEW (batched host): a paused playback is neither sampled nor mixed, keeps its state and is not gated.  The reference pauses an instance's own proxies on AudioServer (audio_spatializer.cpp:115-122); with one shared proxy per bus the pause has to live per playback */
	uint64_t mix_offset; /* resampled: position in the stream, 16.16 fixed point */
} gaso_playback;

typedef struct gaso_instance {
	int32_t kind;
	int32_t channel_count; /* audio_spatializer.cpp:172-179 */
	float mix_rate;
	float disable_threshold_db; /* audio_spatializer.h:87, default -80 */
	int32_t channel_mixed[GASO_MAX_CHANNELS_PER_BUS];
	const gaso_hrtf *hrtf;
	/* caller-provided scratch: playback_buffer[n+64], process_buffer[n], temp_buffer[n], fx_temp[n] */
	gaso_frame *playback_buffer, *process_buffer, *temp_buffer, *fx_temp;
	/* mix_buffer[c] each n frames */
	gaso_frame *mix_buffer[GASO_MAX_CHANNELS_PER_BUS];
	int32_t mix_buffer_size;
} gaso_instance;

/* Source window + fade-out (audio_spatializer.cpp:367-408). Fills buf[0..n+64). */
void gaso_fetch_source(gaso_playback *pb, gaso_frame *buf, int n, float pitch_scale, float mix_rate);
/* Per-playback DSP, dispatch as audio_spatializer.cpp:410-462 but WITHOUT the
 * accumulate: writes the playback's contribution for channel c into
 * contrib[c][0..n) and returns its peak. Used by batched-vs-serial tests. */
void gaso_playback_contribution(gaso_instance *inst, const gaso_params *params, gaso_pdata3d *pd3d, gaso_pdata_effect *pdfx, const gaso_frame *buf, int n, gaso_frame *const contrib[GASO_MAX_CHANNELS_PER_BUS], float peak[2]);
/* Whole mixer: playbacks[] is walked in array order (= list order). */
void gaso_mix_from_playback_list(gaso_instance *inst, const gaso_params *const *params, gaso_playback *const *playbacks, int n_playbacks, int n);
int gaso_check_channel_mixed(gaso_instance *inst, int channel);
int gaso_get_mixed_frames(gaso_instance *inst, const gaso_params *const *params, gaso_playback *const *playbacks, int n_playbacks, int channel, gaso_frame *frames, int n);
/* get_bus_map factor for one bus (audio_spatializer.cpp:295-319). */
void gaso_bus_map(int should_mix_channels, int channel, const float bus_volume[4][2], const float mix_volumes[4][2], float out[4][2]);


/* ---- calculate_spatialization arithmetic (audio_spatializer_3d.cpp:103-151, 277-434, 903-938), SURVEY.md 8f#1 ----
 * Physics queries (Area3D override, reverb send: :208-256, :322-331, :398-403) stay on the host and are not
 * restated; everything that is pure arithmetic on (source pose, listener poses, resource properties) is. */
typedef struct gaso_spat3d_config { /* AudioSpatializer3D properties (audio_spatializer_3d.h:171-187) + engine facts */
	int32_t attenuation_model; /* 0 inverse distance, 1 inverse square, 2 logarithmic, 3 disabled (audio_spatializer_3d.h:157-162) */
	float unit_size, max_distance, panning_strength;
	int32_t emission_angle_enabled;
	float emission_angle, emission_angle_filter_attenuation_db;
	float attenuation_filter_cutoff_hz, attenuation_filter_db;
	int32_t doppler_tracking; /* 0 disabled (audio_spatializer_3d.h:164-168) */
	float doppler_speed_of_sound;
	float global_panning_strength; /* audio/general/3d_panning_strength, audio_spatializer_3d.cpp:633 */
	int32_t speaker_mode; /* [ENGINE] AudioServer::SpeakerMode: 0 stereo, 1 3.1, 2 5.1, 3 7.1 */
	uint32_t hrtf_n_az, hrtf_n_el; /* NEW: azimuth x elevation grid of the HRIR set (0 = leave the HRTF fields alone) */
	uint32_t reserved;
} gaso_spat3d_config;

typedef struct gaso_source_pose {
	float position[3]; /* get_global_transform().origin, :280 */
	float volume_db; /* AudioStreamPlayerSpatial::volume_db, :146 */
	float velocity[3]; /* velocity_tracker linear velocity, :296 */
	float max_db; /* :147 */
	float forward[3]; /* get_global_transform().basis.get_column(2), :380 */
	float pitch_scale; /* player pitch scale, :420,:433 */
} gaso_source_pose;

typedef struct gaso_listener { /* listener_node->get_global_transform().orthonormalized(), :343 */
	float basis[3][3]; /* basis[r][c]: global = basis * local + origin */
	float origin[3];
	float velocity[3]; /* get_doppler_tracked_velocity(), :409-413 */
	float pad;
} gaso_listener;

/* One source, all listeners. *was_further mirrors was_further_than_max_distance_last_frame (:474-475).
 * Fills mix_volumes, pitch_scale, linear_attenuation, attenuation_filter_cutoff_hz, update_parameters
 * (and hrtf_gain / hrtf_dir when a grid is configured); other fields of *out are left untouched.
 * Returns has_any_listener_in_range. */
int gaso_calc_spatialization(const gaso_spat3d_config *cfg, const gaso_source_pose *src, const gaso_listener *listeners, int n_listeners, int32_t *was_further, gaso_params *out);

/* The Area3D a source sits in (found by the physics query of audio_spatializer_3d.cpp:208-256; only the numbers the
 * arithmetic reads).  present == 0: no area. */
typedef struct gaso_area_send {
	uint32_t using_reverb_bus; /* Area3D::is_using_reverb_bus(), :349,365,399 */
	float reverb_uniformity; /* :158 */
	float reverb_amount; /* :159 */
	uint32_t present;
} gaso_area_send;

/* gaso_calc_spatialization with the Area3D branches: listener_area_pos[li][3] is the closest point of the area
 * volume to listener li in that listener's space (:350-353, physics query + transform, host side); it widens or
 * vetoes the max-distance test (:364-370) and feeds calc_reverb_vol (:154-197), whose per-listener results are
 * max-combined into out_reverb[4][2] (:399-402) -- the volumes the reference sends to the area's reverb bus (:451-452).
 * area == NULL or !present: identical to gaso_calc_spatialization, out_reverb zeroed. */
int gaso_calc_spatialization_area(const gaso_spat3d_config *cfg, const gaso_source_pose *src, const gaso_listener *listeners, int n_listeners, int32_t *was_further, const gaso_area_send *area, const float *listener_area_pos, gaso_params *out, float (*out_reverb)[2]);

/* ---- batched convenience used by the GPU parity tests and the CPU baseline ----
 * One callback over n_src independent sources of one kind, each row of src is
 * that source's already-windowed F frames (what process_frames/mix_channel see).
 * Accumulates serially in row order in f32 (audio_spatializer.cpp:433-434,450-451);
 * mix64 (optional) receives the same sum accumulated in f64. */
typedef struct gaso_batch_state {
	gaso_pdata3d pd3d;
	gaso_pdata_effect pdfx;
} gaso_batch_state;
void gaso_batch_block(int kind, int channel_count, const gaso_params *params, gaso_batch_state *states, const gaso_hrtf *hrtf, const gaso_frame *src, int n_src, int n, float mix_rate, gaso_frame *mix /* [C][n] */, double *mix64 /* [C][n][2] or NULL */, float *peaks /* [n_src][2] */);

/* HRTF by overlap-save with a plain radix-2 FFT: CPU baseline arithmetic for the
 * HRTF configs, same semantics as GASO_FX_HRTF. */
/* Precompute (malloc) the HRIR spectra for FFT length L so the baseline does not re-transform taps per source. */
void gaso_hrtf_prepare(gaso_hrtf *hrtf, int L);
void gaso_hrtf_release(gaso_hrtf *hrtf);
void gaso_hrtf_ols_radix2(const gaso_params *params, gaso_fx_state *st, const gaso_hrtf *hrtf, const gaso_frame *src, gaso_frame *dst, int n);

#ifdef __cplusplus
}
#endif
#endif
