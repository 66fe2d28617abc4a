// gas_multi.hip -- several GPUs in one process (include/gas_amd_host.h): per-device
// contexts, all-to-one peer copies of the [C][F] partial mixes into a root-resident [G][C][F] buffer, one ordered sum.
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
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
	// SURVEY.md 8e's preferred form for the 4 KiB message: the shard's own final sum (k_mix_reduce / the fused epilogue)
	// writes its [F] mix straight into its row of the root's gather buffer -- a peer store over the shard's xGMI link, no
	// copy engine in between.  Needs one channel pair (the gather buffer is channel-major) and, across devices, peer
	// access from the shard's device to the root's.  GAS_MULTI_DIRECT=0 keeps the hipMemcpyPeerAsync form everywhere.
	std::vector<uint8_t> direct;
	float *d_gather = nullptr; // root device: 2 x [C][G][F*2] (the reduce kernel's partial layout), alternating per callback
	gas_audio_frame *d_sum = nullptr; // root device: [C][F]
	hipStream_t root_stream = nullptr;
	// Callback t lands in gather half t & 1; root_done[h] = the root has summed half h (two callbacks ago): a shard may
	// rewrite it.  With two halves a shard's next callback never waits for the sum of the one before it.
	hipEvent_t root_done[2] = { nullptr, nullptr };
	bool root_pending[2] = { false, false };
	uint64_t tick = 0;

	// One enqueue thread per shard beyond the first (the caller's thread drives shard 0).  A callback costs a shard
	// about six HIP calls (wait, two launches, event record: 12-20 us of host time, measured with bench.py --in-process);
	// issued from ONE thread that is G x 12-20 us per callback and the GPUs starve.  Each worker touches only its own
	// shard (context, stream, events, staging), so the shards need no lock between them; the hand-over is one generation
	// counter (spin briefly, then sleep on the condition variable: back-to-back callbacks never sleep, a 10.67 ms audio
	// period does not burn a core per GPU).  GAS_MULTI_THREADS=1 turns them on; the default is the caller's thread.
	struct Job {
		const gas_audio_frame *const *src = nullptr;
		const uint32_t *const *slots = nullptr;
		const uint32_t *n = nullptr;
		float *const *peaks = nullptr;
		int mem = GAS_MEM_HOST;
	} job;
	std::vector<std::thread> workers;
	std::vector<int> shard_rc;
	std::mutex go_mu;
	std::condition_variable go_cv;
	std::atomic<uint64_t> go_gen{ 0 };
	std::atomic<uint32_t> left{ 0 };
	std::atomic<bool> quit{ false };
};

namespace {
#define MHIP(call)                    \
	do {                              \
		if ((call) != hipSuccess) {   \
			return GAS_ERR_DEVICE;    \
		}                             \
	} while (0)
} // namespace

// Shard g's part of a callback: (stage its rows,) run its callback on its own stream, land its partial in the root's
// gather buffer, record its event.  Called by exactly one thread per shard.
static int shard_callback(gas_multi *m, uint32_t g) {
	const gas_multi::Job &j = m->job;
	const uint32_t G = (uint32_t)m->shard.size(), C = m->cfg.channel_count, F = m->cfg.frames;
	MHIP(hipSetDevice(m->device[g]));
	const gas_audio_frame *rows = j.src[g];
	float *pk = m->d_peaks[g];
	const uint32_t n_g = j.n[g];
	if (j.mem == GAS_MEM_HOST) {
		const size_t need = (size_t)n_g * F;
		if (need > m->stage_frames[g]) { // grow-only: a steady callback never allocates
			MHIP(hipStreamSynchronize(m->stream[g]));
			(void)hipFree(m->d_stage[g]);
			m->d_stage[g] = nullptr;
			m->stage_frames[g] = 0;
			if (hipMalloc(&m->d_stage[g], need * sizeof(gas_audio_frame)) != hipSuccess) {
				return GAS_ERR_OUT_OF_MEMORY;
			}
			m->stage_frames[g] = need;
		}
		if (n_g > 0 && hipMemcpyAsync(m->d_stage[g], j.src[g], need * sizeof(gas_audio_frame), hipMemcpyHostToDevice, m->stream[g]) != hipSuccess) {
			return GAS_ERR_DEVICE;
		}
		rows = m->d_stage[g];
	} else if (j.peaks && j.peaks[g]) {
		pk = j.peaks[g];
	}
	const bool direct = m->direct[g] != 0;
	const int half = (int)(m->tick & 1);
	float *gather = m->d_gather + (size_t)half * C * G * F * 2;
	if (direct && m->root_pending[half]) { // the shard's own sum writes its gather row: the root must have read what was there
		MHIP(hipStreamWaitEvent(m->stream[g], m->root_done[half], 0));
	}
	gas_audio_frame *shard_out = direct ? reinterpret_cast<gas_audio_frame *>(gather) + (size_t)g * F : m->d_out[g];
	int rc = gas_process_block(m->shard[g], rows, j.slots[g], n_g, F, shard_out, pk, GAS_MEM_DEVICE);
	if (rc == GAS_OK) {
		rc = gas_ctx_join_outputs(m->shard[g]); // the gather below runs on the shard's stream
	}
	if (rc != GAS_OK) {
		return rc;
	}
	if (!direct) {
		if (m->root_pending[half]) { // the sum that read this half must be done before it is rewritten
			MHIP(hipStreamWaitEvent(m->stream[g], m->root_done[half], 0));
		}
		for (uint32_t c = 0; c < C; c++) {
			float *dst = gather + ((size_t)c * G + g) * F * 2;
			MHIP(hipMemcpyPeerAsync(dst, m->device[0], m->d_out[g] + (size_t)c * F, m->device[g], (size_t)F * sizeof(gas_audio_frame), m->stream[g]));
		}
	}
	if (j.mem == GAS_MEM_HOST && j.peaks && j.peaks[g] && n_g > 0) {
		MHIP(hipMemcpyAsync(j.peaks[g], m->d_peaks[g], (size_t)n_g * 2 * sizeof(float), hipMemcpyDeviceToHost, m->stream[g]));
	}
	MHIP(hipEventRecord(m->done[g], m->stream[g])); // behind the gather AND the peaks copy of this shard
	return GAS_OK;
}

static void worker_main(gas_multi *m, uint32_t g) {
	uint64_t seen = 0;
	for (;;) {
		// a new generation: spin a little (queued callers hand the next one over within microseconds), then sleep
		uint64_t gen = m->go_gen.load(std::memory_order_acquire);
		for (int spin = 0; gen == seen && spin < 4000 && !m->quit.load(std::memory_order_relaxed); spin++) {
			__builtin_ia32_pause();
			gen = m->go_gen.load(std::memory_order_acquire);
		}
		if (gen == seen) {
			std::unique_lock<std::mutex> lk(m->go_mu);
			m->go_cv.wait(lk, [&] { return m->go_gen.load(std::memory_order_acquire) != seen || m->quit.load(); });
			gen = m->go_gen.load(std::memory_order_acquire);
		}
		if (m->quit.load()) {
			return;
		}
		seen = gen;
		m->shard_rc[g] = shard_callback(m, g);
		m->left.fetch_sub(1, std::memory_order_acq_rel);
	}
}

extern "C" {

void gas_multi_destroy(gas_multi *m) {
	if (!m) {
		return;
	}
	if (!m->workers.empty()) {
		{
			std::lock_guard<std::mutex> lk(m->go_mu);
			m->quit.store(true);
		}
		m->go_cv.notify_all();
		for (std::thread &t : m->workers) {
			t.join();
		}
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
		for (int h = 0; h < 2; h++) {
			if (m->root_done[h]) {
				(void)hipEventDestroy(m->root_done[h]);
			}
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
		m->direct.push_back(0);
		if (hipSetDevice(devices[g]) != hipSuccess || hipStreamCreateWithFlags(&m->stream[g], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&m->done[g], hipEventDisableTiming) != hipSuccess || hipMalloc(&m->d_out[g], mix_frames * sizeof(gas_audio_frame)) != hipSuccess || hipMalloc(&m->d_peaks[g], (size_t)cfg->max_sources * 2 * sizeof(float)) != hipSuccess) {
			rc = GAS_ERR_DEVICE;
			break;
		}
		rc = gas_ctx_set_stream(ctx, m->stream[g]);
		const char *direct_env = std::getenv("GAS_MULTI_DIRECT");
		const bool direct_ok = cfg->channel_count == 1 && !(direct_env && direct_env[0] == '0');
		if (rc == GAS_OK && devices[g] == devices[0]) {
			m->direct[g] = direct_ok ? 1 : 0;
		} else if (rc == GAS_OK) {
			int can = 0;
			if (hipDeviceCanAccessPeer(&can, devices[g], devices[0]) == hipSuccess && can) {
				const hipError_t pe = hipDeviceEnablePeerAccess(devices[0], 0);
				if (pe == hipErrorPeerAccessAlreadyEnabled) {
					(void)hipGetLastError(); // not an error: another shard on this device enabled it
				}
				m->direct[g] = (direct_ok && (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled)) ? 1 : 0;
			}
		}
	}
	if (rc == GAS_OK) {
		if (hipSetDevice(devices[0]) != hipSuccess || hipStreamCreateWithFlags(&m->root_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&m->root_done[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&m->root_done[1], hipEventDisableTiming) != hipSuccess || hipMalloc(&m->d_gather, 2 * mix_frames * n_devices * sizeof(gas_audio_frame)) != hipSuccess || hipMalloc(&m->d_sum, mix_frames * sizeof(gas_audio_frame)) != hipSuccess) {
			rc = GAS_ERR_DEVICE;
		}
	}
	if (rc != GAS_OK) {
		gas_multi_destroy(m);
		return rc;
	}
	m->shard_rc.assign(n_devices, GAS_OK);
	const char *thr_env = std::getenv("GAS_MULTI_THREADS");
	if (n_devices > 1 && thr_env && thr_env[0] == '1') { // opt-in: on shards that share ONE device the threads only contend for the runtime's locks (measured: no gain), and no box with several GPUs has run them yet
		for (uint32_t g = 1; g < n_devices; g++) {
			m->workers.emplace_back(worker_main, m, g);
		}
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
	// 1. every shard: (stage its rows,) run its callback on its own stream, then land its partial in the root's gather
	//    buffer over its own link ([C][G][F*2]: channel-major, the reduce kernel's layout) -- shard 0 on this thread,
	//    the others on their enqueue threads, all at once
	m->job.src = src;
	m->job.slots = slots;
	m->job.n = n;
	m->job.peaks = peaks;
	m->job.mem = mem;
	int rc = GAS_OK;
	if (!m->workers.empty()) {
		m->left.store(G - 1, std::memory_order_release);
		{
			std::lock_guard<std::mutex> lk(m->go_mu);
			m->go_gen.fetch_add(1, std::memory_order_acq_rel);
		}
		m->go_cv.notify_all();
		m->shard_rc[0] = shard_callback(m, 0);
		while (m->left.load(std::memory_order_acquire) != 0) {
			__builtin_ia32_pause();
		}
		for (uint32_t g = 0; g < G && rc == GAS_OK; g++) {
			rc = m->shard_rc[g];
		}
	} else {
		for (uint32_t g = 0; g < G && rc == GAS_OK; g++) {
			rc = shard_callback(m, g);
		}
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
	const int half = (int)(m->tick & 1);
	const float *gather = m->d_gather + (size_t)half * C * G * F * 2;
	m->tick++;
	if (mem == GAS_MEM_DEVICE) {
		MHIP(gas_launch_mix_reduce(m->root_stream, gather, G, G, C, F, out));
		MHIP(hipEventRecord(m->root_done[half], m->root_stream));
		m->root_pending[half] = true;
		return GAS_OK; // complete in root-stream order: gas_multi_synchronize()
	}
	MHIP(gas_launch_mix_reduce(m->root_stream, gather, G, G, C, F, m->d_sum));
	m->root_pending[half] = false; // this call waits for the root below
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
