// TEST-ONLY stand-in for the gas_* entries batched_spatializer_host.cpp calls, so that the host layer can be built
// for the CPU under ThreadSanitizer (tests/test_host_tsan.py).  Never part of the product: libgas_amd.so has no CPU
// path.  The stub keeps the real library's threading contract and nothing more: the slot API and gas_process_block
// use plain (non-atomic) state, so a host that called them from two threads would be reported by the sanitizer;
// gas_params_publish takes a mutex like the real one.
#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/gas_amd.h"

struct gas_ctx {
	gas_config cfg{};
	std::vector<int> used; // plain state: audio-thread only
	std::vector<gas_params> params;
	uint32_t last_stream_n = 0;
	std::mutex alloc_mu;
	std::mutex params_mu;
	std::atomic<unsigned long> blocks{ 0 }; // read by the driver thread
	unsigned long allocs = 0, frees = 0;
};

extern "C" {

gas_ctx *gas_stub_ctx_create(uint32_t max_sources, uint32_t frames) {
	gas_ctx *c = new gas_ctx();
	c->cfg.struct_size = sizeof(gas_config);
	c->cfg.max_sources = max_sources;
	c->cfg.frames = frames;
	c->cfg.channel_count = 1;
	c->cfg.mix_rate = 48000.0f;
	c->used.assign(max_sources, 0);
	c->params.resize(max_sources);
	return c;
}

void gas_stub_ctx_destroy(gas_ctx *c) {
	delete c;
}

unsigned long gas_stub_blocks(gas_ctx *c) {
	return c->blocks;
}

int gas_ctx_get_config(gas_ctx *c, gas_config *out) {
	*out = c->cfg;
	return GAS_OK;
}

int gas_source_alloc(gas_ctx *c, int, const int32_t *, uint32_t, uint32_t *out_slot) {
	std::lock_guard<std::mutex> lk(c->alloc_mu); // like the real allocator: open to every thread
	for (uint32_t s = 0; s < c->cfg.max_sources; s++) {
		if (!c->used[s]) {
			c->used[s] = 1;
			c->allocs++;
			*out_slot = s;
			return GAS_OK;
		}
	}
	return GAS_ERR_OUT_OF_SLOTS;
}

int gas_fx_settings_publish(gas_ctx *c, const uint32_t *, const gas_fx_settings *, uint32_t) {
	std::lock_guard<std::mutex> lk(c->params_mu); // like the real one: under the parameter lock
	return GAS_OK;
}

int gas_source_free(gas_ctx *c, uint32_t slot) {
	std::lock_guard<std::mutex> lk(c->alloc_mu);
	if (slot >= c->cfg.max_sources || !c->used[slot]) {
		return GAS_ERR_BAD_SLOT;
	}
	c->used[slot] = 0;
	c->frees++;
	return GAS_OK;
}

int gas_source_set_draining(gas_ctx *c, uint32_t slot, int) {
	return slot < c->cfg.max_sources && c->used[slot] ? GAS_OK : GAS_ERR_BAD_SLOT;
}

int gas_source_bind_stream(gas_ctx *c, uint32_t slot, uint32_t, uint64_t) {
	return slot < c->cfg.max_sources && c->used[slot] ? GAS_OK : GAS_ERR_BAD_SLOT;
}

int gas_params_publish(gas_ctx *c, uint32_t slot, const gas_params *p) {
	if (slot >= c->cfg.max_sources) {
		return GAS_ERR_BAD_SLOT;
	}
	std::lock_guard<std::mutex> lk(c->params_mu);
	c->params[slot] = *p;
	return GAS_OK;
}

// out = sum of the rows scaled by the slot's hrtf_gain; peaks = per-row max |.|: enough arithmetic for the host's gate
int gas_process_block(gas_ctx *c, const gas_audio_frame *src, const uint32_t *slots, uint32_t n, uint32_t frames, gas_audio_frame *out, float *peaks, int) {
	c->blocks++;
	std::memset(out, 0, sizeof(gas_audio_frame) * frames * c->cfg.channel_count);
	std::lock_guard<std::mutex> lk(c->params_mu); // the snapshot of audio_spatializer.cpp:328
	for (uint32_t r = 0; r < n; r++) {
		if (slots[r] >= c->cfg.max_sources || !c->used[slots[r]]) {
			return GAS_ERR_BAD_SLOT;
		}
		const float g = c->params[slots[r]].hrtf_gain;
		float pl = 0, pr = 0;
		for (uint32_t i = 0; i < frames; i++) {
			const gas_audio_frame f = src[(size_t)r * frames + i];
			out[i].left += g * f.left;
			out[i].right += g * f.right;
			const float al = f.left < 0 ? -f.left * g : f.left * g, ar = f.right < 0 ? -f.right * g : f.right * g;
			pl = al > pl ? al : pl;
			pr = ar > pr ? ar : pr;
		}
		peaks[2 * r] = pl;
		peaks[2 * r + 1] = pr;
	}
	return GAS_OK;
}

int gas_process_block_streams(gas_ctx *c, const uint32_t *, uint32_t n, uint32_t frames, gas_audio_frame *out, float *peaks, uint8_t *has_frames, int) {
	c->blocks++;
	std::memset(out, 0, sizeof(gas_audio_frame) * frames * c->cfg.channel_count);
	for (uint32_t r = 0; r < n; r++) {
		peaks[2 * r] = peaks[2 * r + 1] = 0.0f;
		has_frames[r] = 0; // every device stream "ends" at once: exercises the gate + reap path
	}
	c->last_stream_n = n;
	return GAS_OK;
}

int gas_stream_positions(gas_ctx *c, uint32_t n, uint64_t *out_frames) {
	if (n != c->last_stream_n) {
		return GAS_ERR_INVALID_ARGUMENT;
	}
	for (uint32_t r = 0; r < n; r++) {
		out_frames[r] = c->cfg.frames;
	}
	return GAS_OK;
}

} // extern "C"
