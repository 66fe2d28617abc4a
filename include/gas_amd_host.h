/*
 * gas_amd_host.h -- engine-free C++ host layer over gas_amd.h, exported with a C surface.
 *
 * It restates the audio-thread half of AudioSpatializerInstance that stays on the CPU when the DSP moves
 * to the GPU: the playback list (audio_spatializer.h:55-68), start/stop (audio_spatializer.cpp:44-113), the
 * 64-frame source window with end-of-stream fade-out (:367-408), the once-per-callback latch (:494-508),
 * get_mixed_frames (:510-527), the silence gate (:464-469) and the list GC (:473-492) -- but with ONE
 * gas_process_block for all playbacks where the reference loops process_frames / mix_channel per playback.
 * In a Godot build the same class sits behind AudioStreamPlaybackSpatial::mix; here the
 * engine's AudioStreamPlayback::mix is a plain callback.
 *
 * Threading contract (the reference's split, audio_spatializer.h:135-138, with its SafeList / SafeFlag / Mutex roles):
 *   - control entries -- gas_host_start_playback*, gas_host_stop_playback, gas_host_set_spatializer_parameters,
 *     gas_host_set_effect_settings, gas_host_set_playback_disable_threshold_db, gas_host_is_playback_active, gas_host_set_playback_paused,
 *     gas_host_is_playback_paused, gas_host_get_playback_position, gas_host_playback_count, gas_host_set_release_fn,
 *     gas_host_collect_released, gas_host_set_process_effects_fn -- may be
 *     called from any number of threads (main, physics) at any time, concurrently with the audio thread.  They only
 *     queue commands / flip per-playback atomics; a start or a parameter set takes effect at the top of the next
 *     callback, in the order issued (one parameter snapshot per callback, audio_spatializer.cpp:328).
 *   - gas_host_get_mixed_frames is the audio thread: one caller.  It is the only thread that uses the context's slot
 *     API (gas_source_alloc / free / set_draining / bind_stream) and gas_process_block, as gas_amd.h requires; a start
 *     that cannot get a slot (or names an unknown device stream) simply never becomes active.
 *   - nodes are deleted on a control thread (deferred delete, audio_spatializer.cpp:538-547); the stream callback of a
 *     playback is only ever invoked from the audio thread.
 *   - gas_host_create / gas_host_destroy: no other thread inside the host.
 * tests/test_host_tsan.py runs this contract under ThreadSanitizer.
 */
#ifndef GAS_AMD_HOST_H
#define GAS_AMD_HOST_H

#include "gas_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gas_host gas_host;

/* [ENGINE] AudioStreamPlayback::mix(AudioFrame *buffer, float rate_scale, int frames) -> frames mixed. */
typedef int (*gas_host_stream_mix_fn)(void *user, gas_audio_frame *buffer, float rate_scale, int frames);

/* One host = one AudioSpatializerInstance flavour (kind + effect chain) batching all its playbacks. */
int gas_host_create(gas_ctx *ctx, int kind, const int32_t *effects, uint32_t n_effects, gas_host **out_host);
void gas_host_destroy(gas_host *host);

/* start_playback_stream (audio_spatializer.cpp:44-96): allocates the slot, zeroes the lookahead. */
int gas_host_start_playback(gas_host *host, gas_host_stream_mix_fn mix, void *user, uint32_t *out_id);
/* Convenience for tests/tools: a playback over a caller-owned array of frames (zero-filled past its end). */
int gas_host_start_playback_array(gas_host *host, const gas_audio_frame *stream, int64_t stream_frames, uint32_t *out_id);
/* A playback over a device-resident stream (gas_stream_create): nothing crosses PCIe per callback, the source window
 * and fade-out are produced on the GPU (gas_process_block_streams).  A host serves either callback/array playbacks
 * or device-stream playbacks, not a mixture (GAS_ERR_KIND_MISMATCH). */
int gas_host_start_playback_device_stream(gas_host *host, uint32_t stream, uint64_t start_frame, uint32_t *out_id);
/* stop_playback_stream (audio_spatializer.cpp:98-113): active = false; the audio thread reaps it. */
int gas_host_stop_playback(gas_host *host, uint32_t id);
/* set_spatializer_parameters (audio_spatializer.cpp:558-564), per playback. */
int gas_host_set_spatializer_parameters(gas_host *host, uint32_t id, const gas_params *params);
/* Settings of the playback's engine-effect kinds (GAS_FX_LOWPASS .. GAS_FX_AMPLIFY, by chain position): what a script
 * writes to the AudioEffect resources; queued like the parameters, in the order issued.  Control thread. */
int gas_host_set_effect_settings(gas_host *host, uint32_t id, const gas_fx_settings *settings);
void gas_host_set_playback_disable_threshold_db(gas_host *host, float db); /* audio_spatializer.h:87 */
int gas_host_is_playback_active(gas_host *host, uint32_t id);
/* set_playback_paused / is_playback_paused (audio_spatializer.cpp:115-122, :161-170), PER PLAYBACK: the reference
 * pauses the instance's own proxy playbacks on AudioServer, which a host with one shared proxy per bus cannot do
 * without pausing every player.  A paused playback is neither sampled nor mixed from the next callback on; its slot
 * state, its 64-frame lookahead and its stream position are kept; it is not gated and not reaped; un-pausing resumes
 * it where it stopped (also in the middle of its ring-out).  No fade: AudioServer's pause fade works on a proxy's
 * output ([ENGINE]), not inside the module.  A player with polyphony pauses each of its playback ids. */
int gas_host_set_playback_paused(gas_host *host, uint32_t id, int paused);
int gas_host_is_playback_paused(gas_host *host, uint32_t id);
/* get_playback_position (audio_spatializer.cpp:144-157) in frames of the stream consumed so far (what
 * [ENGINE] AudioStreamPlayback::get_playback_position reports, times the mix rate; it runs 64 frames ahead of what is
 * audible, like the reference's): counted by the audio thread for callback and array playbacks, mirrored from the
 * device cursor arithmetic for device-stream playbacks (nothing is read back).  Unknown id: 0 and GAS_ERR_BAD_SLOT. */
int gas_host_get_playback_position(gas_host *host, uint32_t id, uint64_t *out_frames);
int gas_host_playback_count(gas_host *host); /* nodes still on the list */

/* Who owns `user` of gas_host_start_playback: the reference's list node holds a Ref<AudioStreamPlayback> until its
 * deferred delete (audio_spatializer.cpp:538-547), so a caller may drop its own reference right after stop_playback.
 * Here the caller keeps `user` alive until the host says it is finished with it: `fn(fn_user, id, user)` is called once
 * per playback (user is NULL for array / device-stream playbacks), after which the host never calls its stream callback
 * again.  It runs on a CONTROL thread -- inside gas_host_start_playback*, gas_host_collect_released or gas_host_destroy,
 * whichever comes first after the audio thread reaped the playback -- never on the audio thread. */
typedef void (*gas_host_release_fn)(void *fn_user, uint32_t id, void *user);
int gas_host_set_release_fn(gas_host *host, gas_host_release_fn fn, void *fn_user);
/* Release what the audio thread has finished with (call it from the physics tick); returns the number of playbacks. */
int gas_host_collect_released(gas_host *host);

/* _process_effects(params, playback_data) (audio_spatializer_effect.cpp:39,90-92; the example's
 * gd_spatializer_instance.gd:125-127 sets its high-shelf's gain from the parameters there): called on the AUDIO thread,
 * once per audible playback per callback, before the launch, with the parameter row last set for the playback.  The
 * effects' settings live in that row (highshelf_*, er_*, hrtf_* of gas_params), so "edit the effect" = edit the row;
 * a non-zero return publishes the edited row for this callback (and keeps it until the next
 * gas_host_set_spatializer_parameters).  NULL removes the hook. */
typedef int (*gas_host_process_effects_fn)(void *fn_user, uint32_t id, gas_params *params);
int gas_host_set_process_effects_fn(gas_host *host, gas_host_process_effects_fn fn, void *fn_user);

/* get_mixed_frames (audio_spatializer.cpp:510-527): audio thread; returns GAS_OK, GAS_ERR_BAD_CHANNEL
 * ("Unexpected channel") or GAS_ERR_FRAME_COUNT ("Unexpected frame count"). */
int gas_host_get_mixed_frames(gas_host *host, int channel, gas_audio_frame *frames, int frame_count);

/* ---- several GPUs in ONE process (the in-engine variant) ------------------------------------
 * Sources shard by slot across `n_devices` contexts (one per listed HIP device; the same device may be listed more
 * than once, which is how this is tested on a one-GPU box).  Every shard runs its own gas_process_block; each then
 * writes its [C][F] partial mix into a root-resident [G][C][F] buffer over its own link (hipMemcpyPeerAsync) and the
 * root adds the G partials in fixed shard order with the library's deterministic reduce -- the all-to-one pattern
 * SURVEY.md 8e prefers over a ring for a 4 KiB, latency-bound message.  bench.py's one-process-per-GPU RCCL path is
 * the other variant of the same sharding. */
typedef struct gas_multi gas_multi;
int gas_multi_create(const gas_config *cfg /* .device ignored */, const int32_t *devices, uint32_t n_devices, gas_multi **out);
void gas_multi_destroy(gas_multi *m);
uint32_t gas_multi_shards(gas_multi *m);
gas_ctx *gas_multi_shard(gas_multi *m, uint32_t shard); /* the shard's context: sources, parameters, HRIRs are per shard */
uint32_t gas_multi_least_loaded(gas_multi *m); /* shard with the fewest sources registered through gas_multi_note_alloc */
void gas_multi_note_alloc(gas_multi *m, uint32_t shard, int delta); /* +1 after a gas_source_alloc on it, -1 after a free */
/* One callback: src[g] is shard g's [n[g]][frames] host rows, slots[g] its slot list; out is the summed [C][frames] mix
 * (host); peaks[g] (may be NULL) receives shard g's [n[g]][2] peaks.  Host memory; = gas_multi_process_block_mem(.., GAS_MEM_HOST). */
int gas_multi_process_block(gas_multi *m, const gas_audio_frame *const *src, const uint32_t *const *slots, const uint32_t *n, uint32_t frames, gas_audio_frame *out, float *const *peaks);
/* The same with the memory kind spelled out.  GAS_MEM_DEVICE: src[g] / peaks[g] are device pointers on shard g's
 * device, `out` a device pointer on the root device (devices[0]); the call only enqueues -- no allocation, no host
 * wait -- and `out` is complete in the order of gas_multi_root_stream() / after gas_multi_synchronize().  GAS_MEM_HOST
 * stages through per-shard buffers that only ever grow and ends in ONE wait on the root stream, which is
 * event-chained behind every shard. */
int gas_multi_process_block_mem(gas_multi *m, const gas_audio_frame *const *src, const uint32_t *const *slots, const uint32_t *n, uint32_t frames, gas_audio_frame *out, float *const *peaks, int mem);
int gas_multi_synchronize(gas_multi *m);
void *gas_multi_root_stream(gas_multi *m); /* hipStream_t of the root device on which `out` is produced */

/* get_bus_map (audio_spatializer.cpp:274-324) for ONE bus: the per-channel-pair factors AudioServer multiplies the
 * frames returned for channel `channel` by.  Mix-channel instances mixed their volumes in already, so only the
 * requested pair is non-zero and the bus volume is normalised by the mix volume (0 where the mix volume is <= 0,
 * :295-313); other instances pass the raw mix volumes through (:314-318).  bus_volume / mix_volumes / out: [4][2]. */
void gas_host_bus_map(int should_mix_channels, int channel, const float bus_volume[GAS_MAX_CHANNELS_PER_BUS][2], const float mix_volumes[GAS_MAX_CHANNELS_PER_BUS][2], float out[GAS_MAX_CHANNELS_PER_BUS][2]);

#ifdef __cplusplus
}
#endif
#endif
