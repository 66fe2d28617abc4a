// gas_multi.hip -- several GPUs in one process (include/gas_amd_host.h): per-device
// contexts, all-to-one peer copies of the [C][F] partial mixes into a root-resident [G][C][F] buffer, one ordered sum.
#include <cstring>
#include <new>
#include <vector>

#include "../../include/gas_amd_host.h"
#include "gas_internal.h"

struct gas_multi {
	gas_config cfg{};
	std::vector<gas_ctx *> shard;
	std::vector<int> device;
	std::vector<hipStream_t> stream; // one per shard, created on the shard's device
	std::vector<hipEvent_t> done; // shard g's partial has landed in the root buffer
	std::vector<gas_audio_frame *> d_out; // per shard [C][F] on its own device
	std::vector<float *> d_peaks;
	std::vector<int> load;
	float *d_gather = nullptr; // root device: [C][G][F*2] (the reduce kernel's partial layout)
	gas_audio_frame *d_sum = nullptr; // root device: [C][F]
	hipStream_t root_stream = nullptr;
};

namespace {
#define MHIP(call)                    \
	do {                              \
		if ((call) != hipSuccess) {   \
			return GAS_ERR_DEVICE;    \
		}                             \
	} while (0)
} // namespace

extern "C" {

void gas_multi_destroy(gas_multi *m) {
	if (!m) {
		return;
	}
	for (size_t g = 0; g < m->shard.size(); g++) {
		(void)hipSetDevice(m->device[g]);
		if (m->stream[g]) {
			(void)hipStreamSynchronize(m->stream[g]);
		}
		gas_ctx_destroy(m->shard[g]);
		(void)hipFree(m->d_out[g]);
		(void)hipFree(m->d_peaks[g]);
		if (m->done[g]) {
			(void)hipEventDestroy(m->done[g]);
		}
		if (m->stream[g]) {
			(void)hipStreamDestroy(m->stream[g]);
		}
	}
	if (!m->device.empty()) {
		(void)hipSetDevice(m->device[0]);
		(void)hipFree(m->d_gather);
		(void)hipFree(m->d_sum);
		if (m->root_stream) {
			(void)hipStreamDestroy(m->root_stream);
		}
	}
	delete m;
}

int gas_multi_create(const gas_config *cfg, const int32_t *devices, uint32_t n_devices, gas_multi **out) {
	if (!cfg || !devices || !out || n_devices == 0 || n_devices > 64 || cfg->struct_size != sizeof(gas_config)) {
		return GAS_ERR_INVALID_ARGUMENT;
	}
	*out = nullptr;
	gas_multi *m = new (std::nothrow) gas_multi();
	if (!m) {
		return GAS_ERR_OUT_OF_MEMORY;
	}
	m->cfg = *cfg;
	const size_t mix_frames = (size_t)cfg->channel_count * cfg->frames;
	int rc = GAS_OK;
	for (uint32_t g = 0; g < n_devices && rc == GAS_OK; g++) {
		gas_config c = *cfg;
		c.device = devices[g];
		gas_ctx *ctx = nullptr;
		rc = gas_ctx_create(&c, &ctx);
		if (rc != GAS_OK) {
			break;
		}
		m->shard.push_back(ctx);
		m->device.push_back(devices[g]);
		m->stream.push_back(nullptr);
		m->done.push_back(nullptr);
		m->d_out.push_back(nullptr);
		m->d_peaks.push_back(nullptr);
		m->load.push_back(0);
		if (hipSetDevice(devices[g]) != hipSuccess || hipStreamCreateWithFlags(&m->stream[g], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&m->done[g], hipEventDisableTiming) != hipSuccess || hipMalloc(&m->d_out[g], mix_frames * sizeof(gas_audio_frame)) != hipSuccess || hipMalloc(&m->d_peaks[g], (size_t)cfg->max_sources * 2 * sizeof(float)) != hipSuccess) {
			rc = GAS_ERR_DEVICE;
			break;
		}
		rc = gas_ctx_set_stream(ctx, m->stream[g]);
		if (rc == GAS_OK && g > 0 && devices[g] != devices[0]) {
			int can = 0;
			if (hipDeviceCanAccessPeer(&can, devices[g], devices[0]) == hipSuccess && can) {
				(void)hipDeviceEnablePeerAccess(devices[0], 0); // already-enabled is fine
			}
		}
	}
	if (rc == GAS_OK) {
		if (hipSetDevice(devices[0]) != hipSuccess || hipStreamCreateWithFlags(&m->root_stream, hipStreamNonBlocking) != hipSuccess || hipMalloc(&m->d_gather, mix_frames * n_devices * sizeof(gas_audio_frame)) != hipSuccess || hipMalloc(&m->d_sum, mix_frames * sizeof(gas_audio_frame)) != hipSuccess) {
			rc = GAS_ERR_DEVICE;
		}
	}
	if (rc != GAS_OK) {
		gas_multi_destroy(m);
		return rc;
	}
	*out = m;
	return GAS_OK;
}

uint32_t gas_multi_shards(gas_multi *m) {
	return m ? (uint32_t)m->shard.size() : 0;
}

gas_ctx *gas_multi_shard(gas_multi *m, uint32_t g) {
	return m && g < m->shard.size() ? m->shard[g] : nullptr;
}

uint32_t gas_multi_least_loaded(gas_multi *m) {
	uint32_t best = 0;
	for (uint32_t g = 1; m && g < m->load.size(); g++) {
		if (m->load[g] < m->load[best]) {
			best = g;
		}
	}
	return best;
}

void gas_multi_note_alloc(gas_multi *m, uint32_t g, int delta) {
	if (m && g < m->load.size()) {
		m->load[g] += delta;
	}
}

int gas_multi_process_block(gas_multi *m, const gas_audio_frame *const *src, const uint32_t *const *slots, const uint32_t *n, uint32_t frames, gas_audio_frame *out, float *const *peaks) {
	if (!m || !src || !slots || !n || !out) {
		return GAS_ERR_INVALID_ARGUMENT;
	}
	const uint32_t G = (uint32_t)m->shard.size(), C = m->cfg.channel_count, F = m->cfg.frames;
	const size_t mix_bytes = (size_t)C * F * sizeof(gas_audio_frame);
	auto fail = [&](int code) {
		std::memset(out, 0, mix_bytes);
		return code;
	};
	if (frames != F) {
		return fail(GAS_ERR_FRAME_COUNT);
	}
	// 1. every shard: stage its rows, run its callback on its own stream into its own [C][F] buffer, then write
	//    that partial into the root's gather buffer ([C][G][F*2]: channel-major, the reduce kernel's layout)
	std::vector<gas_audio_frame *> staged(G, nullptr);
	int rc = GAS_OK;
	for (uint32_t g = 0; g < G && rc == GAS_OK; g++) {
		MHIP(hipSetDevice(m->device[g]));
		if (n[g] > 0) {
			if (hipMallocAsync(reinterpret_cast<void **>(&staged[g]), (size_t)n[g] * F * sizeof(gas_audio_frame), m->stream[g]) != hipSuccess || hipMemcpyAsync(staged[g], src[g], (size_t)n[g] * F * sizeof(gas_audio_frame), hipMemcpyHostToDevice, m->stream[g]) != hipSuccess) {
				rc = GAS_ERR_DEVICE;
				break;
			}
		}
		rc = gas_process_block(m->shard[g], staged[g], slots[g], n[g], F, m->d_out[g], m->d_peaks[g], GAS_MEM_DEVICE);
		if (rc == GAS_OK) {
			rc = gas_ctx_join_outputs(m->shard[g]); // the gather below runs on the shard's stream
		}
		if (rc != GAS_OK) {
			break;
		}
		for (uint32_t c = 0; c < C; c++) {
			float *dst = m->d_gather + ((size_t)c * G + g) * F * 2;
			MHIP(hipMemcpyPeerAsync(dst, m->device[0], m->d_out[g] + (size_t)c * F, m->device[g], (size_t)F * sizeof(gas_audio_frame), m->stream[g]));
		}
		if (peaks && peaks[g] && n[g] > 0) {
			MHIP(hipMemcpyAsync(peaks[g], m->d_peaks[g], (size_t)n[g] * 2 * sizeof(float), hipMemcpyDeviceToHost, m->stream[g]));
		}
		if (staged[g]) {
			MHIP(hipFreeAsync(staged[g], m->stream[g]));
		}
		MHIP(hipEventRecord(m->done[g], m->stream[g]));
	}
	if (rc != GAS_OK) {
		for (uint32_t g = 0; g < G; g++) {
			(void)hipSetDevice(m->device[g]);
			(void)hipStreamSynchronize(m->stream[g]);
		}
		return fail(rc);
	}
	// 2. root: wait for the G partials, add them in shard order (deterministic), copy the mix out
	MHIP(hipSetDevice(m->device[0]));
	for (uint32_t g = 0; g < G; g++) {
		MHIP(hipStreamWaitEvent(m->root_stream, m->done[g], 0));
	}
	MHIP(gas_launch_mix_reduce(m->root_stream, m->d_gather, G, G, C, F, m->d_sum));
	MHIP(hipMemcpyAsync(out, m->d_sum, mix_bytes, hipMemcpyDeviceToHost, m->root_stream));
	MHIP(hipStreamSynchronize(m->root_stream));
	for (uint32_t g = 0; g < G; g++) { // peaks copies
		MHIP(hipSetDevice(m->device[g]));
		MHIP(hipStreamSynchronize(m->stream[g]));
	}
	return GAS_OK;
}

} // extern "C"
