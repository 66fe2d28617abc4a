// gas_module.cpp -- see gas_module.h.
#include "gas_module.h"

#include "core/config/project_settings.h"
#include "core/os/mutex.h"
#include "servers/audio_server.h"

uint32_t GasModule::max_sources = 16384;

namespace {
Mutex gas_mutex;
gas_ctx *the_ctx = nullptr;
bool ctx_failed = false;

struct HostKey {
	StringName bus;
	int kind = 0;
	uint32_t chain = 0; // effect kinds, 4 bits each
	bool operator==(const HostKey &o) const { return bus == o.bus && kind == o.kind && chain == o.chain; }
};
struct HostKeyHasher {
	static uint32_t hash(const HostKey &k) { return hash_murmur3_one_32(k.chain, hash_murmur3_one_32((uint32_t)k.kind, k.bus.hash())); }
};
HashMap<HostKey, gas_host *, HostKeyHasher> hosts;

// The engine playbacks the hosts may still call mix() on, keyed by the pointer handed to gas_host_start_playback as
// `user`.  Control threads only (retain: before the start; released: inside gas_host_start_playback*,
// gas_host_collect_released or gas_host_destroy) -- never the audio thread.
Mutex retained_mutex;
HashMap<AudioStreamPlayback *, Ref<AudioStreamPlayback>> retained;
void released(void *, uint32_t, void *p_user) {
	Ref<AudioStreamPlayback> last; // destroyed after the lock is gone: a playback's destructor may be arbitrary engine code
	MutexLock lock(retained_mutex);
	AudioStreamPlayback *key = static_cast<AudioStreamPlayback *>(p_user);
	if (Ref<AudioStreamPlayback> *r = retained.getptr(key)) {
		last = *r;
		retained.erase(key);
	}
}
} // namespace

void GasModule::retain(const Ref<AudioStreamPlayback> &p_playback) {
	MutexLock lock(retained_mutex);
	retained.insert(p_playback.ptr(), p_playback);
}

void GasModule::unretain(AudioStreamPlayback *p_playback) {
	released(nullptr, 0, p_playback);
}

gas_ctx *GasModule::ctx() {
	MutexLock lock(gas_mutex);
	if (the_ctx || ctx_failed) {
		return the_ctx;
	}
	gas_config cfg = {};
	cfg.struct_size = sizeof(gas_config);
	cfg.device = 0;
	cfg.max_sources = GLOBAL_DEF("audio/gpu_spatializer/max_sources", (int)max_sources);
	cfg.frames = 512; // AudioServer's fixed mix step (audio_spatializer_3d.cpp:590,601 assume it too)
	cfg.channel_count = AudioServer::get_singleton()->get_channel_count(); // audio_spatializer.cpp:176
	cfg.mix_rate = AudioServer::get_singleton()->get_mix_rate(); // audio_spatializer_3d.cpp:506
	cfg.er_ring_frames = 4096;
	cfg.flags = GAS_FLAG_PEAKS_DRAINING_ONLY; // the module only reads a playback's peak once its stream ended (:464-469)
	const int rc = gas_ctx_create(&cfg, &the_ctx);
	if (rc != GAS_OK) {
		ctx_failed = true;
		the_ctx = nullptr;
		WARN_PRINT(vformat("GPU spatializer unavailable (%s); AudioSpatializerHRTF falls back to AudioSpatializer3D.", gas_strerror(rc)));
	}
	return the_ctx;
}

gas_host *GasModule::host_for(const StringName &p_bus, int p_kind, const int32_t *p_effects, uint32_t p_n_effects) {
	gas_ctx *c = ctx();
	ERR_FAIL_NULL_V(c, nullptr);
	HostKey key;
	key.bus = p_bus;
	key.kind = p_kind;
	for (uint32_t i = 0; i < p_n_effects; i++) {
		key.chain |= (uint32_t)(p_effects[i] & 0xf) << (4 * i);
	}
	MutexLock lock(gas_mutex);
	if (gas_host **found = hosts.getptr(key)) {
		return *found;
	}
	gas_host *h = nullptr;
	const int rc = gas_host_create(c, p_kind, p_effects, p_n_effects, &h);
	ERR_FAIL_COND_V_MSG(rc != GAS_OK, nullptr, gas_strerror(rc));
	gas_host_set_release_fn(h, released, nullptr);
	hosts.insert(key, h);
	return h;
}

void GasModule::shutdown() {
	MutexLock lock(gas_mutex);
	for (KeyValue<HostKey, gas_host *> &kv : hosts) {
		gas_host_destroy(kv.value);
	}
	hosts.clear();
	if (the_ctx) {
		gas_ctx_destroy(the_ctx);
		the_ctx = nullptr;
	}
}
