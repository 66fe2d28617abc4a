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
	std::vector<gas_audio_frame *> d_stage; // per shard: persistent staging of host rows (grown on demand, never per callback)
	std::vector<size_t> stage_frames;
	float *d_gather = nullptr; // root device: [C][G][F*2] (the reduce kernel's partial layout)
	gas_audio_frame *d_sum = nullptr; // root device: [C][F]
	hipStream_t root_stream = nullptr;
	hipEvent_t root_done = nullptr; // the root has summed the gather buffer of the previous callback: it may be rewritten
	bool root_pending = false;
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
		(void)hipFree(m->d_stage[g]);
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
		if (m->root_done) {
			(void)hipEventDestroy(m->root_done);
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
		m->d_stage.push_back(nullptr);
		m->stage_frames.push_back(0);
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
		if (hipSetDevice(devices[0]) != hipSuccess || hipStreamCreateWithFlags(&m->root_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&m->root_done, hipEventDisableTiming) != hipSuccess || hipMalloc(&m->d_gather, mix_frames * n_devices * sizeof(gas_audio_frame)) != hipSuccess || hipMalloc(&m->d_sum, mix_frames * sizeof(gas_audio_frame)) != hipSuccess) {
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

// One callback over every shard.  mem = GAS_MEM_HOST: src[g] / peaks[g] / out are host arrays and the call returns
// when `out` holds the mix (ONE wait at the end: the root stream is event-chained behind every shard).  mem =
// GAS_MEM_DEVICE: src[g] and peaks[g] live on shard g's device, `out` on the root device (devices[0]); the call only
// enqueues and `out` is complete in the order of gas_multi_synchronize() / the root stream.
int gas_multi_process_block_mem(gas_multi *m, const gas_audio_frame *const *src, const uint32_t *const *slots, const uint32_t *n, uint32_t frames, gas_audio_frame *out, float *const *peaks, int mem) {
	if (!m || !src || !slots || !n || !out || (mem != GAS_MEM_HOST && mem != GAS_MEM_DEVICE)) {
		return GAS_ERR_INVALID_ARGUMENT;
	}
	const uint32_t G = (uint32_t)m->shard.size(), C = m->cfg.channel_count, F = m->cfg.frames;
	const size_t mix_bytes = (size_t)C * F * sizeof(gas_audio_frame);
	auto fail = [&](int code) {
		if (mem == GAS_MEM_HOST) {
			std::memset(out, 0, mix_bytes);
		}
		return code;
	};
	if (frames != F) {
		return fail(GAS_ERR_FRAME_COUNT);
	}
	// 1. every shard: (stage its rows,) run its callback on its own stream into its own [C][F] buffer, then write that
	//    partial into the root's gather buffer over its own link ([C][G][F*2]: channel-major, the reduce kernel's layout)
	int rc = GAS_OK;
	for (uint32_t g = 0; g < G && rc == GAS_OK; g++) {
		MHIP(hipSetDevice(m->device[g]));
		const gas_audio_frame *rows = src[g];
		float *pk = m->d_peaks[g];
		if (mem == GAS_MEM_HOST) {
			const size_t need = (size_t)n[g] * F;
			if (need > m->stage_frames[g]) { // grow-only: a steady callback never allocates
				MHIP(hipStreamSynchronize(m->stream[g]));
				(void)hipFree(m->d_stage[g]);
				m->d_stage[g] = nullptr;
				m->stage_frames[g] = 0;
				if (hipMalloc(&m->d_stage[g], need * sizeof(gas_audio_frame)) != hipSuccess) {
					rc = GAS_ERR_OUT_OF_MEMORY;
					break;
				}
				m->stage_frames[g] = need;
			}
			if (n[g] > 0 && hipMemcpyAsync(m->d_stage[g], src[g], need * sizeof(gas_audio_frame), hipMemcpyHostToDevice, m->stream[g]) != hipSuccess) {
				rc = GAS_ERR_DEVICE;
				break;
			}
			rows = m->d_stage[g];
		} else if (peaks && peaks[g]) {
			pk = peaks[g];
		}
		rc = gas_process_block(m->shard[g], rows, slots[g], n[g], F, m->d_out[g], pk, GAS_MEM_DEVICE);
		if (rc == GAS_OK) {
			rc = gas_ctx_join_outputs(m->shard[g]); // the gather below runs on the shard's stream
		}
		if (rc != GAS_OK) {
			break;
		}
		if (m->root_pending) { // the previous callback's sum must have read the gather buffer before it is rewritten
			MHIP(hipStreamWaitEvent(m->stream[g], m->root_done, 0));
		}
		for (uint32_t c = 0; c < C; c++) {
			float *dst = m->d_gather + ((size_t)c * G + g) * F * 2;
			MHIP(hipMemcpyPeerAsync(dst, m->device[0], m->d_out[g] + (size_t)c * F, m->device[g], (size_t)F * sizeof(gas_audio_frame), m->stream[g]));
		}
		if (mem == GAS_MEM_HOST && peaks && peaks[g] && n[g] > 0) {
			MHIP(hipMemcpyAsync(peaks[g], m->d_peaks[g], (size_t)n[g] * 2 * sizeof(float), hipMemcpyDeviceToHost, m->stream[g]));
		}
		MHIP(hipEventRecord(m->done[g], m->stream[g])); // behind the gather AND the peaks copy of this shard
	}
	if (rc != GAS_OK) {
		for (uint32_t g = 0; g < G; g++) {
			(void)hipSetDevice(m->device[g]);
			(void)hipStreamSynchronize(m->stream[g]);
		}
		return fail(rc);
	}
	// 2. root: behind the G shards (events), add the partials in shard order (deterministic), hand the mix over
	MHIP(hipSetDevice(m->device[0]));
	for (uint32_t g = 0; g < G; g++) {
		MHIP(hipStreamWaitEvent(m->root_stream, m->done[g], 0));
	}
	if (mem == GAS_MEM_DEVICE) {
		MHIP(gas_launch_mix_reduce(m->root_stream, m->d_gather, G, G, C, F, out));
		MHIP(hipEventRecord(m->root_done, m->root_stream));
		m->root_pending = true;
		return GAS_OK; // complete in root-stream order: gas_multi_synchronize()
	}
	MHIP(gas_launch_mix_reduce(m->root_stream, m->d_gather, G, G, C, F, m->d_sum));
	m->root_pending = false; // this call waits for the root below
	MHIP(hipMemcpyAsync(out, m->d_sum, mix_bytes, hipMemcpyDeviceToHost, m->root_stream));
	MHIP(hipStreamSynchronize(m->root_stream)); // the one wait of the call: everything above is chained in front of it
	return GAS_OK;
}

int gas_multi_process_block(gas_multi *m, const gas_audio_frame *const *src, const uint32_t *const *slots, const uint32_t *n, uint32_t frames, gas_audio_frame *out, float *const *peaks) {
	return gas_multi_process_block_mem(m, src, slots, n, frames, out, peaks, GAS_MEM_HOST);
}

int gas_multi_synchronize(gas_multi *m) {
	if (!m) {
		return GAS_ERR_INVALID_ARGUMENT;
	}
	MHIP(hipSetDevice(m->device[0]));
	MHIP(hipStreamSynchronize(m->root_stream));
	return GAS_OK;
}

void *gas_multi_root_stream(gas_multi *m) {
	return m ? m->root_stream : nullptr;
}

} // extern "C"
