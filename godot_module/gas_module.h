// gas_module.h -- process-wide owner of the MI355X spatializer context and its batching hosts.
// Godot-side glue (godot_module/README.md): compiled inside the engine tree, not in this repository.
#pragma once

#include "core/string/string_name.h"
#include "core/templates/hash_map.h"
#include "servers/audio/audio_stream.h"

#include "gas_amd.h"
#include "gas_amd_host.h"

class AudioStreamPlaybackGasBatch;

class GasModule {
public:
	// The context every GPU spatializer of the process shares; nullptr when no MI355X is usable (callers then fall back
	// to the stock AudioSpatializer3D path -- the library itself has no CPU path).
	static gas_ctx *ctx();
	// One batching host per (bus, kind, effect chain): all playbacks routed to `bus` through that spatializer flavour.
	static gas_host *host_for(const StringName &p_bus, int p_kind, const int32_t *p_effects, uint32_t p_n_effects);
	// The Ref that keeps an engine playback alive while the audio thread may still call its mix(): taken BEFORE
	// gas_host_start_playback, dropped when the host's release callback names the playback (the list node's Ref in the
	// reference, audio_spatializer.cpp:538-547); unretain() is for a start that failed.
	static void retain(const Ref<AudioStreamPlayback> &p_playback);
	static void unretain(AudioStreamPlayback *p_playback);
	// Module shutdown (uninitialize_audio_spatializer_module): hosts first, then the context.
	static void shutdown();

	static uint32_t max_sources; // project setting audio/gpu_spatializer/max_sources, default 16384
};
