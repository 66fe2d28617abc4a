/*
 * gas_amd.h -- C ABI of the MI355X-native many-source spatializer.
 *
 * This is the drop-in boundary for ONE path of BuzzLord/godot-audio-spatializer:
 * the per-audio-callback work of AudioSpatializerInstance::_mix_from_playback_list
 * (audio_spatializer.cpp:326-471) and the plugin DSP it dispatches to
 * (audio_spatializer_3d.cpp:491-609, audio_spatializer_effect.cpp:33-77), batched
 * over every active source in one launch group.  Plain pointers and sizes only;
 * no exceptions cross it and nothing aborts: every entry returns 0 or a negative
 * gas_status, mirroring the reference's ERR_FAIL_* "log and return" style
 * (SURVEY.md section 5).  All file:line citations are in the reference tree.
 *
 * Threading contract (same split as audio_spatializer.h:135-138):
 *   - gas_params_publish*, gas_bus_routes_publish : physics thread, may run concurrently with the audio thread.
 *   - gas_source_alloc, gas_source_free : any thread, concurrently with the audio thread (instantiate_playback_data runs
 *     on the physics thread, audio_spatializer.cpp:69; playback data is released wherever its last reference drops).
 *     The slot allocator takes a lock; a free takes effect at the audio thread's next block boundary; a slot is
 *     handed to the audio thread by the caller (the first callback whose list names it), never before alloc returned.
 *   - gas_process_block*, gas_process_frames_1, gas_mix_channel_1, gas_source_set_draining, gas_source_bind_stream,
 *     gas_stream_positions : audio thread only, one caller at a time, never re-entrant per context.
 *   - everything else      : one thread at a time and not concurrently with the audio-thread entries -- the main
 *     thread while audio is stopped, or the audio thread itself between callbacks.
 */
#ifndef GAS_AMD_H
#define GAS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAS_ABI_VERSION 2 /* 2: gas_profile grew (callbacks_per_launch, bytes_per_callback_formula); pause / position / effect kinds */

/* audio_spatializer.h:47-52 */
#define GAS_MAX_CHANNELS_PER_BUS 4
#define GAS_LOOKAHEAD_BUFFER_SIZE 64
#define GAS_MAX_BUSES_PER_PLAYBACK 6

/* NEW (no reference counterpart): sizes of the HRTF / early-reflection effects. */
#define GAS_HRTF_TAPS 256
#define GAS_ER_TAPS 8
#define GAS_MAX_EFFECTS 4

typedef struct gas_ctx gas_ctx;

/* [ENGINE] AudioFrame: two interleaved f32 (left, right), 8 bytes. */
typedef struct gas_audio_frame {
	float left, right;
} gas_audio_frame;

typedef enum gas_status {
	GAS_OK = 0,
	GAS_ERR_INVALID_ARGUMENT = -1,
	GAS_ERR_OUT_OF_SLOTS = -2,
	GAS_ERR_BAD_SLOT = -3,
	GAS_ERR_FRAME_COUNT = -4, /* audio_spatializer.cpp:522 "Unexpected frame count" */
	GAS_ERR_NO_HRTF = -5,
	GAS_ERR_UNSUPPORTED_CHAIN = -6,
	GAS_ERR_DEVICE = -7, /* a HIP call failed; gas_last_device_error() has the text */
	GAS_ERR_NO_DEVICE = -8,
	GAS_ERR_OUT_OF_MEMORY = -9,
	GAS_ERR_KIND_MISMATCH = -10,
	GAS_ERR_BAD_CHANNEL = -11, /* audio_spatializer.cpp:521 "Unexpected channel" */
	GAS_ERR_NO_PARAMS = -12, /* audio_spatializer.cpp:330 parameters.is_null() */
} gas_status;

/* Which AudioSpatializerInstance flavour a source slot belongs to. */
typedef enum gas_source_kind {
	GAS_KIND_3D_MIX = 0, /* AudioSpatializer3D, mix_channel_mode=true : _mix_channel per channel pair (audio_spatializer_3d.cpp:554-609) */
	GAS_KIND_3D_PROCESS = 1, /* AudioSpatializer3D, mix_channel_mode=false: _process_frames (audio_spatializer_3d.cpp:491-552) */
	GAS_KIND_EFFECT = 2, /* AudioSpatializerEffect / AudioSpatializerHRTF: effect chain (audio_spatializer_effect.cpp:33-77) */
} gas_source_kind;

typedef enum gas_effect_kind {
	GAS_FX_HIGHSHELF = 1, /* [ENGINE] AudioEffectHighShelfFilter, FILTER_6DB (gd_spatializer.gd:14-19) */
	GAS_FX_EARLY_REFLECTIONS = 2, /* NEW: 8 stereo delay taps from a per-source ring */
	GAS_FX_HRTF = 3, /* NEW: mono downmix -> gain ramp -> 256-tap HRIR pair, overlap-save FFT */
	/* audio_spatializer_effect.cpp:79-88 instantiates ANY AudioEffect; these are the engine's other one-biquad filters
	 * ([ENGINE] AudioEffectFilter subclasses at FILTER_6DB = one AudioFilterSW stage per ear, coefficients snapped per
	 * block) and AudioEffectAmplify.  Their settings are per playback and chain position: gas_fx_settings. */
	GAS_FX_LOWPASS = 4, /* [ENGINE] AudioEffectLowPassFilter */
	GAS_FX_HIGHPASS = 5, /* [ENGINE] AudioEffectHighPassFilter */
	GAS_FX_BANDPASS = 6, /* [ENGINE] AudioEffectBandPassFilter */
	GAS_FX_NOTCH = 7, /* [ENGINE] AudioEffectNotchFilter */
	GAS_FX_LOWSHELF = 8, /* [ENGINE] AudioEffectLowShelfFilter */
	GAS_FX_AMPLIFY = 9, /* [ENGINE] AudioEffectAmplify: volume ramp previous -> current volume_db across the block */
} gas_effect_kind;

typedef enum gas_mem {
	GAS_MEM_HOST = 0, /* host pointers; the call copies in/out and returns when the result is in *out */
	GAS_MEM_DEVICE = 1, /* device pointers on the context's GPU; the call only enqueues on the context stream */
} gas_mem;

/* Mirrors module initialisation (register_types.cpp:40) + the AudioServer facts the
 * reference reads at run time (mix rate audio_spatializer_3d.cpp:506, channel count
 * audio_spatializer.cpp:176, the fixed 512-frame mix step). */
typedef struct gas_config {
	uint32_t struct_size; /* sizeof(gas_config) */
	int32_t device; /* HIP device ordinal */
	uint32_t max_sources; /* slots of device-resident SpatializerPlaybackData */
	uint32_t frames; /* F: frames per callback, fixed for the context (audio_spatializer.cpp:522); multiple of 128, <= 512 */
	uint32_t channel_count; /* C: AudioServer channel pairs, 1..4 (audio_spatializer.cpp:172-179) */
	float mix_rate; /* AudioServer::get_mix_rate(), e.g. 48000 */
	uint32_t er_ring_frames; /* early-reflection ring length per source (power of two, 0 = effect unavailable) */
	uint32_t flags; /* GAS_FLAG_* */
} gas_config;

/* gas_config.flags */
/* The reference computes every playback's output peak but only reads it once the stream has ended
 * (audio_spatializer.cpp:464-469).  With this flag HRTF sources that are NOT marked draining are summed
 * in the frequency domain (one forward FFT per source, inverse FFTs per workgroup) and report
 * peak = +inf ("not measured", never passes the gate); the mix is unchanged.  Without it every source
 * reports its exact peak, as the reference computes it.  "HRTF sources": chains [HRTF] and [ER, HRTF], and any other
 * chain whose LAST effect is the HRTF when the context runs that stage in the one-launch kernel (no cross-fade /
 * direction-run flags); every other chain reports exact peaks with or without the flag (they cost nothing there). */
#define GAS_FLAG_PEAKS_DRAINING_ONLY 1u
/* NEW: when an HRTF source's direction differs from the one of its previous callback, render the
 * block with both HRIRs and blend old -> new with t = i/F (the analogue of the per-block volume lerp,
 * audio_spatializer_3d.cpp:591-592) instead of switching at the block boundary. */
#define GAS_FLAG_HRTF_CROSSFADE 2u
/* HRTF sources whose hrtf_dir repeats can share one forward FFT (sum_s Z_s H[d] = FFT(sum_s z_s) H[d]).
 * GAS_FLAG_DIRECTION_RUNS: the caller keeps sources with equal hrtf_dir adjacent in the callback's list (as far as
 * it likes: any list is correct); frequency-domain sources then sum each run in the time domain before one FFT
 * (measured 15-19 % less kernel time at 8 sources per direction; 4-5 % MORE on a list without runs, hence a flag).
 * GAS_FLAG_DIRECTION_ORDER: the library builds the grouping itself (a device counting sort per 8192-source segment,
 * re-run after every parameter publish or list change); implies _RUNS.  Pays only when parameters are published
 * much less often than callbacks run.  Results are identical up to f32 summation order; neither
 * flag has an effect together with GAS_FLAG_HRTF_CROSSFADE. */
#define GAS_FLAG_DIRECTION_ORDER 4u
/* Throughput mode for callers that queue many callbacks (offline rendering, benchmarks): the final sum of the
 * per-workgroup partial mixes of gas_process_block(GAS_MEM_DEVICE) is not launched as its own kernel; it is carried
 * out by otherwise idle waves of the NEXT callback's HRTF kernel (one dispatch per callback instead of two).  `out`
 * of such a call is complete only after the next gas_process_block on this context, gas_ctx_join_outputs(),
 * gas_ctx_synchronize(), or any GAS_MEM_HOST / single-instance call -- each in the context's stream order; peaks
 * are not deferred.  Keep `out` valid and unread until then.  Applies to contexts with channel_count == 1 whose
 * callback contains HRTF sources; other callbacks are summed immediately.  Same operations in the same order:
 * results are bitwise identical to the ordered mode.  A synchronous audio callback gains nothing from it. */
#define GAS_FLAG_PIPELINED_MIX 8u
#define GAS_FLAG_DIRECTION_RUNS 16u /* see GAS_FLAG_DIRECTION_ORDER */
/* XCD-affine processing order for plain [HRTF] callbacks (>= 2048 sources): after every parameter publish or list
 * change one small launch (k_xcd_order) permutes the callback's sources so that each of the GPU's eight XCDs (own L2
 * each) keeps working on the same eighth of the HRIR spectra table.  MEASURED (profiles/r02_notes.md): L2 fills drop
 * from 1.46x to 1.05x of the algorithmic bytes at 65536 sources, but the kernel gains only 4 % (the rows are then read
 * in scattered instead of list order) and the ordering launch costs more than that -- so it is OFF unless asked for;
 * kept because it answers what the excess traffic costs.  The mix is the same sum in another (still deterministic)
 * order: results agree to f32 rounding, not bitwise, with the unordered launch. */
#define GAS_FLAG_XCD_ORDER 32u
/* With GAS_FLAG_PIPELINED_MIX, for callers that queue callbacks faster than they consume them (offline rendering,
 * benchmarks): consecutive gas_process_block(GAS_MEM_DEVICE) calls of an unchanged plain-[HRTF] list run as ONE launch
 * per `depth` callbacks (k_hrtf_multi: no dispatch gap between them, a callback's epilogue under the next one's
 * stream; depth 2 unless gas_ctx_set_batch_depth says otherwise).  A call only records its arguments until the batch is
 * full; the call that fills it enqueues all of them.  Same operations in the same order: results are bitwise those of
 * the unbatched mode.  What changes for the caller:
 *  - `src`, `peaks` and device-published parameter rows of a call must stay valid and unmodified until `depth - 1`
 *    further gas_process_block calls on this context (or gas_ctx_join_outputs / gas_ctx_synchronize) have returned;
 *  - `out` is complete only after gas_ctx_join_outputs / gas_ctx_synchronize (or any ordered call), in stream order;
 *  - every call waiting for its batch or its deferred sum needs ITS OWN `out` (and `peaks`) buffer: the sums of the
 *    queued callbacks are written out of call order (a block's sum rides in a later launch, the join sums everything
 *    pending in one parallel launch), so two queued calls that share an `out` race; the unbatched mode's
 *    "last call wins" does not hold here;
 *  - every entry of the context must be called from ONE thread (the recorded calls are run by whoever comes next:
 *    anything that changes what they must see -- a new list, host-published parameters, frees, any other entry that
 *    enqueues work -- first runs them, exactly as the unbatched mode would have);
 *  - callbacks that do not qualify (other source kinds, < 2048 sources, 2 or more channel pairs, streams, cross-fade
 *    or direction flags) run as before. */
#define GAS_FLAG_BATCHED_LAUNCH 64u

/* SpatializerParameters (spatializer_parameters.h:39-67) + SpatializerParameters3D
 * (audio_spatializer_3d.h:61-83) as one 128-byte POD, plus the per-block effect
 * parameters a _process_effects hook would push (audio_spatializer_effect.cpp:90-92,
 * gd_spatializer_instance.gd:125-127).  bus_volumes stay on the host: they only feed
 * AudioServer via get_bus_map (audio_spatializer.cpp:274-324). */
typedef struct gas_params {
	float mix_volumes[GAS_MAX_CHANNELS_PER_BUS][2]; /* must describe 4 channel pairs, spatializer_parameters.cpp:45 */
	float pitch_scale; /* consumed by the host-side sampler only (audio_spatializer.cpp:375) */
	float linear_attenuation; /* high-shelf gain, audio_spatializer_3d.h:67 */
	float attenuation_filter_cutoff_hz; /* audio_spatializer_3d.h:68, default 5000 */
	uint32_t update_parameters; /* spatializer_parameters.h:50, host-side only */
	float hrtf_gain; /* NEW: linear gain ramped across the block before the HRIR */
	uint32_t hrtf_dir; /* NEW: HRIR direction index */
	float fx_shelf_gain; /* GAS_FX_HIGHSHELF gain (linear) */
	float fx_shelf_cutoff_hz; /* GAS_FX_HIGHSHELF cutoff */
	float er_gain[GAS_ER_TAPS]; /* NEW */
	uint32_t er_delay[GAS_ER_TAPS]; /* NEW: 1 .. er_ring_frames - frames */
} gas_params;

/* Settings of the GAS_FX_LOWPASS .. GAS_FX_AMPLIFY effects of one playback, by chain position (what a script sets on
 * the AudioEffect resources from _process_effects, gd_spatializer_instance.gd:125-127).  Position j is read only when
 * effect j of the playback's chain is one of those kinds.  A slot that never got settings has the engine's resource
 * defaults: cutoff 2000 Hz, resonance 0.5, gain 1, volume 0 dB. */
typedef struct gas_fx_settings {
	float filter_cutoff_hz[GAS_MAX_EFFECTS]; /* [ENGINE] AudioEffectFilter::cutoff */
	float filter_resonance[GAS_MAX_EFFECTS]; /* [ENGINE] AudioEffectFilter::resonance */
	float filter_gain[GAS_MAX_EFFECTS]; /* [ENGINE] AudioEffectFilter::gain (linear; shelf kinds) */
	float amplify_volume_db[GAS_MAX_EFFECTS]; /* [ENGINE] AudioEffectAmplify::volume_db */
} gas_fx_settings;

/* Per-kernel device timing collected with HIP events on the context stream. */
typedef struct gas_profile {
	uint64_t launches; /* timed launches of the dominant kernel since the last reset */
	double kernel_ms; /* sum of their durations */
	uint64_t bytes_per_launch; /* bytes the last timed launch had to move: SURVEY.md 8d's per-callback
	                            * formula for a one-callback launch; for a launch of K callbacks (GAS_FLAG_BATCHED_LAUNCH)
	                            * K x (frames + per-block parameters / peaks + partial mix) + the history rows once in and
	                            * once out when they stay in LDS between blocks (K times otherwise) + the HRIR table once */
	char kernel_name[64];
	uint32_t callbacks_per_launch; /* K of the last timed launch (1 unless batched) */
	uint32_t reserved;
	uint64_t bytes_per_callback_formula; /* SURVEY.md 8d's per-callback figure x K: what round 2 reported for batched launches */
} gas_profile;

/* ---- context ---------------------------------------------------------- */
int gas_abi_version(void);
int gas_ctx_create(const gas_config *cfg, gas_ctx **out_ctx);
void gas_ctx_destroy(gas_ctx *ctx);
/* Run on an existing HIP stream (hipStream_t) instead of the context's own. */
int gas_ctx_set_stream(gas_ctx *ctx, void *hip_stream);
int gas_ctx_synchronize(gas_ctx *ctx);
/* GAS_FLAG_BATCHED_LAUNCH: callbacks per launch, 1 (off) .. 16; takes effect with the next batch. */
int gas_ctx_set_batch_depth(gas_ctx *ctx, uint32_t depth);
/* GAS_FLAG_PIPELINED_MIX: enqueue the deferred sum of the last gas_process_block, so that work enqueued on the
 * context's stream after this call sees its `out`.  Non-blocking; no-op when nothing is pending. */
int gas_ctx_join_outputs(gas_ctx *ctx);
int gas_ctx_get_config(gas_ctx *ctx, gas_config *out); /* the configuration the context was created with */
const char *gas_strerror(int status);
const char *gas_last_device_error(gas_ctx *ctx);

/* ---- per-playback state: _instantiate_playback_data (audio_spatializer.cpp:69),
 * deferred delete (audio_spatializer.cpp:538-547) ------------------------ */
/* GAS_KIND_EFFECT chains (audio_spatializer_effect.cpp:33-77): up to GAS_MAX_EFFECTS of GAS_FX_*, in processing
 * order, with at most one EARLY_REFLECTIONS (needs cfg.er_ring_frames) and one HRTF per playback.  [], [HIGHSHELF],
 * [ER], [HRTF] and [ER, HRTF] run as one fused kernel; any other order or kind runs one launch per effect through
 * ping-pong row buffers, as the reference's loop does.  Everything else: GAS_ERR_UNSUPPORTED_CHAIN. */
int gas_source_alloc(gas_ctx *ctx, int kind, const int32_t *effects, uint32_t n_effects, uint32_t *out_slot);
int gas_source_free(gas_ctx *ctx, uint32_t slot); /* takes effect at the next block boundary */
int gas_source_reset(gas_ctx *ctx, uint32_t slot); /* zero the slot's DSP state (a restarted playback) */
/* has_frames cleared (audio_spatializer.cpp:398): from now on the host reads this source's peak.  Changing the
 * flag invalidates the cached slot list: pass `slots` to the next gas_process_block. */
int gas_source_set_draining(gas_ctx *ctx, uint32_t slot, int draining);

/* ---- set_spatializer_parameters (audio_spatializer.cpp:558-564): latest wins,
 * snapshotted once at the start of the next gas_process_block (:328) ------ */
int gas_params_publish(gas_ctx *ctx, uint32_t slot, const gas_params *params);
/* params_mem == GAS_MEM_DEVICE with slots == NULL addresses the slot list of the last gas_process_block in its row
 * order and is DEFERRED: the rows are read when the next gas_process_block (or any other entry that needs the table)
 * is enqueued, so the buffer must stay valid and unmodified on other streams until then. */
int gas_params_publish_batch(gas_ctx *ctx, const uint32_t *slots, const gas_params *params, uint32_t n, int params_mem);

/* Settings of the engine-effect kinds, host arrays; latest wins, snapshotted with the parameters at the start of the
 * next gas_process_block.  Physics thread, like gas_params_publish. */
int gas_fx_settings_publish(gas_ctx *ctx, const uint32_t *slots, const gas_fx_settings *settings, uint32_t n);

/* ---- NEW AudioSpatializerHRTF resource: hrir is [dirs][2 ears][taps] f32, taps <= 256 */
int gas_hrtf_load(gas_ctx *ctx, const float *hrir, uint32_t dirs, uint32_t taps);
/* The same resource from a MEASURED set (what a SOFA file holds: M source positions x 2 receivers x N samples; parsing
 * netCDF / HDF5 is the caller's business, the loader takes the arrays).  positions is [m][2] = (azimuth, elevation) in
 * radians, azimuth from straight ahead (-Z) towards the right (+X), elevation up from the horizontal plane -- SOFA's
 * spherical convention is azimuth counter-clockwise in degrees: azimuth = -radians(sofa_azimuth).  hrir is
 * [m][2 ears][taps], taps <= 256 (shorter sets are zero-padded).  The set is regridded ON THE DEVICE onto the
 * az_steps x el_steps grid the library indexes directions with (hrtf_dir = elevation_index * az_steps + azimuth_index;
 * azimuth_index = round(az / 2pi * az_steps) mod az_steps, elevation_index = round((el + pi/2) / pi * (el_steps - 1)):
 * the cells gas_calc_spatialization writes): interpolation 0 takes the measurement nearest on the sphere, 1 blends the
 * three nearest with weights 1 / (angle + 1e-4) -- a time-domain blend, adequate for dense sets only.  out_hrir, when
 * non-NULL, receives the gridded set [az_steps * el_steps][2][256] (what gas_hrtf_load was then called with).  NEW, no
 * reference counterpart: parity unpinned.  Main thread, like gas_hrtf_load. */
int gas_hrtf_load_positions(gas_ctx *ctx, const float *positions, const float *hrir, uint32_t m, uint32_t taps, uint32_t az_steps, uint32_t el_steps, int interpolation, float *out_hrir);

/* ---- the hot path: body of _mix_from_playback_list (audio_spatializer.cpp:353-470)
 * for n sources at once.  src is [n][frames] AudioFrames, row i already holds the
 * 64-frame-delayed window of source slots[i] (audio_spatializer.cpp:367-378);
 * out is [C][frames] (C = channel_count for GAS_KIND_3D_MIX sources, else row 0 only
 * is non-zero) and is fully overwritten (zeros when n == 0, :335-343); peaks is
 * [n][2] = per-source max |L|, max |R| over its mixed output (:436-443,:453-460),
 * feeding the host's silence gate (:464-469).  slots == NULL reuses the previous
 * call's slot list (n must match).  On error the status is returned and, for host
 * memory, *out is zero-filled. */
int gas_process_block(gas_ctx *ctx, const gas_audio_frame *src, const uint32_t *slots, uint32_t n, uint32_t frames, gas_audio_frame *out, float *peaks, int mem);

/* ---- several output buses ---------------------------------------------------------------------
 * In the reference every playback carries a bus_volumes dictionary (spatializer_parameters.h:39-67: <= 6 buses,
 * audio_spatializer.h:52) and AudioServer multiplies the frames a proxy returns by get_bus_map's per-channel-pair
 * factors for each of them (audio_spatializer.cpp:274-324).  AudioSpatializer3D fills it with the dry bus -- the
 * player's, or the Area3D's override -- at the output volume, and the area's reverb bus at reverb_volume
 * (audio_spatializer_3d.cpp:437-461).  For a mix-channel instance the dry factor is bus/mix = 1 and the send factor
 * is reverb_volume / mix_volume per channel pair and ear (0 where the mix volume is <= 0, :295-313).  Batched, that
 * is: every source names the bus its dry signal goes to and (optionally) a bus that receives it scaled by `send`;
 * one launch produces all buses:
 *     out[b][c][i] = sum over sources with dry_bus == b of y[c][i]  +  sum over sources with send_bus == b of y[c][i] * send[c]
 * (y = the source's mix_channel output for pair c).  GAS_KIND_3D_MIX sources only. */
#define GAS_MAX_BUSES GAS_MAX_BUSES_PER_PLAYBACK
#define GAS_MAX_MORE_SENDS (GAS_MAX_BUSES_PER_PLAYBACK - 2)
#define GAS_BUS_NONE 0xffffffffu
/* A playback reaches up to GAS_MAX_BUSES_PER_PLAYBACK buses (audio_spatializer.cpp:283-287 walks every key of
 * bus_volumes up to that many): the dry bus and up to five sends.  AudioSpatializer3D uses the dry bus and one send
 * (audio_spatializer_3d.cpp:437-461): the first two fields; a script subclass with more buses fills more_*.  A bus
 * named more than once receives the sum of its weights. */
typedef struct gas_bus_route {
	uint32_t dry_bus; /* index < n_buses of the call, or GAS_BUS_NONE */
	uint32_t send_bus; /* first send: index < n_buses, or GAS_BUS_NONE */
	float send[GAS_MAX_CHANNELS_PER_BUS][2]; /* bus_volume / mix_volume per channel pair and ear (gas_host_bus_map arithmetic) */
	uint32_t more_bus[GAS_MAX_MORE_SENDS]; /* further sends; GAS_BUS_NONE = unused */
	float more_send[GAS_MAX_MORE_SENDS][GAS_MAX_CHANNELS_PER_BUS][2];
} gas_bus_route;
/* Host arrays; latest wins, snapshotted at the next gas_process_block_buses.  A slot that never got a route is
 * {dry_bus 0, no send}.  Physics thread, like gas_params_publish. */
int gas_bus_routes_publish(gas_ctx *ctx, const uint32_t *slots, const gas_bus_route *routes, uint32_t n);
/* gas_process_block with out = [n_buses][C][frames] (1 <= n_buses <= GAS_MAX_BUSES).  Either every source is
 * GAS_KIND_3D_MIX (the mix-channel buses, fused into the biquad launch), or every source is a non-empty effect chain
 * on a one-pair context (the chains run staged -- per-source rows -- and the rows are mixed per bus with send[0]);
 * anything else is GAS_ERR_UNSUPPORTED_CHAIN.  Plain [HRTF] sources run fused, one launch per PAIR of buses (the second
 * bus's spectra sums in LDS), the last launch committing the playbacks' state.  slots == NULL reuses the previous call's
 * list (3D mix and fused [HRTF] forms).  Peaks are those of y (before any bus factor), as the reference's gate sees them
 * (audio_spatializer.cpp:436-443). */
int gas_process_block_buses(gas_ctx *ctx, const gas_audio_frame *src, const uint32_t *slots, uint32_t n, uint32_t frames, gas_audio_frame *out, uint32_t n_buses, float *peaks, int mem);

/* ---- compatibility / parity path: the exact _process_frames and _mix_channel
 * signatures for one source (audio_spatializer.h:146,148), host pointers, synchronous. */
int gas_process_frames_1(gas_ctx *ctx, uint32_t slot, gas_audio_frame *out, const gas_audio_frame *src, int frame_count);
int gas_mix_channel_1(gas_ctx *ctx, uint32_t slot, int channel, gas_audio_frame *out, const gas_audio_frame *src, int frame_count);


/* ---- batched parameter generation on the device ---------------------------------
 * The arithmetic of AudioSpatializerInstance3D::calculate_spatialization (audio_spatializer_3d.cpp:277-489)
 * for n sources against n_listeners listeners in one launch: distance attenuation models (:123-151),
 * max-distance cut (:361-374), high-shelf gain (:376-388), emission cone (:378-385), stereo pan law
 * (:103-110), SPCAP surround (:56-98, :903-938), doppler pitch (:405-438), update_parameters latch
 * (:472-479).  Physics queries (Area3D override / reverb send) stay on the host.  Results are written
 * straight into the slots' device-resident parameters (as gas_params_publish would) and optionally
 * copied to out_params for the host's get_bus_map (audio_spatializer.cpp:274-324). */
typedef struct gas_spatializer3d_config { /* AudioSpatializer3D properties, audio_spatializer_3d.h:171-187 */
	int32_t attenuation_model; /* 0 inverse distance, 1 inverse square distance, 2 logarithmic, 3 disabled */
	float unit_size, max_distance, panning_strength;
	int32_t emission_angle_enabled;
	float emission_angle, emission_angle_filter_attenuation_db;
	float attenuation_filter_cutoff_hz, attenuation_filter_db;
	int32_t doppler_tracking; /* 0 = disabled */
	float doppler_speed_of_sound;
	float global_panning_strength; /* project setting audio/general/3d_panning_strength (audio_spatializer_3d.cpp:633) */
	int32_t speaker_mode; /* [ENGINE] AudioServer::SpeakerMode: 0 stereo, 1 3.1, 2 5.1, 3 7.1 */
	uint32_t hrtf_n_az, hrtf_n_el; /* NEW: azimuth x elevation grid of the loaded HRIR set; 0 = do not write hrtf_gain / hrtf_dir */
	uint32_t reserved;
} gas_spatializer3d_config;

typedef struct gas_source_pose {
	float position[3]; /* player global origin */
	float volume_db; /* AudioStreamPlayerSpatial volume_db */
	float velocity[3]; /* tracked linear velocity (used when doppler_tracking != 0) */
	float max_db;
	float forward[3]; /* player global basis column 2 (emission cone axis) */
	float pitch_scale;
} gas_source_pose;

typedef struct gas_listener { /* orthonormalized global transform of the camera / AudioListener3D */
	float basis[3][3]; /* basis[r][c]; global = basis * local + origin */
	float origin[3];
	float velocity[3];
	float pad;
} gas_listener;

#define GAS_MAX_LISTENERS 16
#define GAS_MAX_SPATIALIZER_CONFIGS 64

/* cfgs, cfg_index ([n] or NULL = config 0 for all), listeners and slots are host arrays; poses (and out_params,
 * [n] gas_params, may be NULL) are host or device pointers according to `mem`.  Physics thread. */
int gas_calc_spatialization(gas_ctx *ctx, const gas_spatializer3d_config *cfgs, uint32_t n_cfgs, const uint32_t *cfg_index, const gas_source_pose *poses, const gas_listener *listeners, uint32_t n_listeners, const uint32_t *slots, uint32_t n, gas_params *out_params, int mem);

/* The same with the Area3D branches of calculate_spatialization.  The physics queries stay on the
 * host: which area a source sits in (audio_spatializer_3d.cpp:208-256) and, per listener, the closest point of the
 * area volume in that listener's space (:350-353).  The arithmetic is batched: the widened / vetoed max-distance
 * test (:364-370) and calc_reverb_vol (:154-197), max-combined over the listeners (:399-402) into the volumes the
 * reference sends to the area's reverb bus (:451-452).  Which bus that is (override / reverb bus names) is the
 * caller's bookkeeping. */
typedef struct gas_area_send {
	uint32_t using_reverb_bus; /* Area3D::is_using_reverb_bus() */
	float reverb_uniformity; /* Area3D::get_reverb_uniformity() */
	float reverb_amount; /* Area3D::get_reverb_amount() */
	uint32_t present; /* 0: the source sits in no area (other fields ignored) */
} gas_area_send;

/* areas [n], listener_area_pos [n][n_listeners][3] (may be NULL when no area has uniformity > 0) and out_reverb
 * [n][4] AudioFrames (may be NULL) are host or device pointers according to `mem`, like poses.  areas == NULL is
 * gas_calc_spatialization. */
int gas_calc_spatialization_areas(gas_ctx *ctx, const gas_spatializer3d_config *cfgs, uint32_t n_cfgs, const uint32_t *cfg_index, const gas_source_pose *poses, const gas_listener *listeners, uint32_t n_listeners, const uint32_t *slots, uint32_t n, const gas_area_send *areas, const float *listener_area_pos, gas_params *out_params, gas_audio_frame *out_reverb, int mem);


/* ---- device-resident source sampling ------------------------------------------------
 * The step in front of the path: AudioStreamPlayback::mix into the 64-frame lookahead window with the
 * end-of-stream fade-out (audio_spatializer.cpp:367-408), over PCM that already lives in HBM, so a callback
 * moves no source frames over PCIe.  Streams are 16-bit PCM as in the reference's example asset
 * (speech_orig.wav: mono, 16-bit, 48 kHz); samples convert as s / 32768, mono feeds both ears [ENGINE].
 * Streams are taken to be at the context's mix rate; pitch-scaled playback: gas_stream_set_resampled. */
typedef enum gas_pcm_format {
	GAS_PCM_S16 = 0, /* interleaved little-endian int16 */
	GAS_PCM_F32 = 1, /* interleaved float32 (already decoded) */
} gas_pcm_format;

int gas_stream_create(gas_ctx *ctx, const void *pcm, int format, uint32_t channels /* 1 or 2 */, uint64_t frames, uint32_t *out_stream);
/* Which engine playback class stands behind the stream's playbacks (choose before binding them):
 *   off (default): frames are handed out as they are; pitch_scale must be 1 (or 0 = unset), anything else is
 *                  GAS_ERR_UNSUPPORTED_CHAIN;
 *   on:            [ENGINE] AudioStreamPlaybackResampled::mix (audio_spatializer.cpp:375-378 passes pitch_scale per
 *                  playback and callback; audio_spatializer_3d.cpp:405-434 sets it from doppler): a 16.16 fixed-point
 *                  position advanced by pitch_scale per output frame and 4-point cubic interpolation over the frames
 *                  q-3 .. q, at ANY pitch including 1 (where it is a 2-frame delay).  Recollection of the engine source,
 *                  parity unpinned; the stream is taken to be at the context's mix rate.  pitch_scale is the value last
 *                  published from the host (gas_params_publish*); 0 <= pitch_scale < 32768. */
int gas_stream_set_resampled(gas_ctx *ctx, uint32_t stream, int on);
int gas_stream_destroy(gas_ctx *ctx, uint32_t stream);
/* Length, channel count and sample format of a stream (any of the outputs may be NULL). */
int gas_stream_get_info(gas_ctx *ctx, uint32_t stream, uint64_t *out_frames, uint32_t *out_channels, int *out_format);
/* start_playback_stream (audio_spatializer.cpp:55-63): the slot's playback starts at start_frame of the stream
 * with a zeroed lookahead and has_frames set. */
int gas_source_bind_stream(gas_ctx *ctx, uint32_t slot, uint32_t stream, uint64_t start_frame);
/* Like gas_process_block, but the source windows are produced on the device from the bound streams (cursor
 * advance, lookahead delay, fade-out, zero feed after the end).  has_frames ([n] bytes, host, may be NULL)
 * receives each playback's has_frames flag after this callback (audio_spatializer.cpp:398); slots whose
 * stream ended are marked draining automatically.  out / peaks are host or device pointers per `mem`. */
int gas_process_block_streams(gas_ctx *ctx, const uint32_t *slots, uint32_t n, uint32_t frames, gas_audio_frame *out, float *peaks, uint8_t *has_frames, int mem);
/* Where the playbacks of the LAST gas_process_block_streams list stand in their streams after it, in its row order:
 * out_frames[i] = index of the next stream frame playback i will take ([ENGINE] get_playback_position x mix rate, plus
 * the playback's start frame).  From the host-side mirror of the cursor arithmetic: nothing is read back from the
 * device.  n must be that callback's n.  Audio thread. */
int gas_stream_positions(gas_ctx *ctx, uint32_t n, uint64_t *out_frames);

/* ---- measurement ------------------------------------------------------- */
/* on = 0 off, 1 = bracket the dominant launch of every callback with HIP events, N > 1 = of every Nth callback. */
int gas_profile_enable(gas_ctx *ctx, int on);
int gas_profile_read(gas_ctx *ctx, gas_profile *out, int reset);
/* Same-run copy-bandwidth ceiling: a pure streaming launch (16-byte loads of read_bytes from an arena
 * larger than the Infinity Cache, 16-byte stores of write_bytes) timed with the same HIP-event bracket and marker
 * calibration as the dominant kernel; *out_us = average span of one launch on the GPU timeline over `iters` launches.
 * workgroups x 256 threads, `unroll` (1, 2, 4 or 8) independent loads in flight per thread. */
int gas_bandwidth_probe(gas_ctx *ctx, uint64_t read_bytes, uint64_t write_bytes, uint32_t workgroups, uint32_t unroll, uint32_t iters, double *out_us);
/* Tuning (process-wide, not per context): plain-[HRTF] callbacks of at least `min_sources` float-row sources on one
 * bus run k_hrtf_uni's twelve-wave form (three waves per SIMD, HRIR rows staged through LDS); 0 = never.  The two
 * forms split the sources over waves differently, so their mixes agree to rounding (1e-5 relative RMS), not to
 * the bit.  The build's default: DESIGN.md 3.1; environment variable GAS_UNI12_MIN overrides it at load time.
 * Returns the previous value. */
uint32_t gas_tune_uni12_min(uint32_t min_sources);
/* Tuning (process-wide): plain-[HRTF] callbacks of at least `min_sources` sources read and write their history rows
 * with non-temporal accesses (rows too many to survive in the Infinity Cache from one callback to the next; default
 * 196608, environment variable GAS_NT_HIST_MIN).  A cache hint only: results do not change.  Returns the previous value. */
uint32_t gas_tune_nt_hist_min(uint32_t min_sources);
/* Diagnostic: the processing order the last gas_process_block used for its plain [HRTF] sources (GAS_FLAG_XCD_ORDER
 * only): out[i] = list entry processed i-th; waits for the stream.  GAS_ERR_INVALID_ARGUMENT when that
 * callback ran in list order. */
int gas_ctx_read_hrtf_order(gas_ctx *ctx, uint32_t *out, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif /* GAS_AMD_H */
