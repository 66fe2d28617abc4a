// TEST-ONLY driver (tests/test_host_tsan.py): hammers the host layer's control entries from two threads while a
// third runs the audio callback, under ThreadSanitizer.  Exit code 0 and no sanitizer report = the threading
// contract of include/gas_amd_host.h holds.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../include/gas_amd_host.h"

extern "C" gas_ctx *gas_stub_ctx_create(uint32_t max_sources, uint32_t frames);
extern "C" void gas_stub_ctx_destroy(gas_ctx *);
extern "C" unsigned long gas_stub_blocks(gas_ctx *);

static int counting_feed(void *user, gas_audio_frame *buf, float, int frames) {
	int *left = static_cast<int *>(user); // frames this playback still has; touched by the audio thread only
	const int got = *left < frames ? *left : frames;
	for (int i = 0; i < frames; i++) {
		buf[i].left = buf[i].right = i < got ? 0.25f : 0.0f;
	}
	*left -= got;
	return got;
}

static std::atomic<int> released{ 0 }, hooked{ 0 };

static void on_release(void *, uint32_t, void *user) { // control threads only; `user` may be freed from here on
	if (user) {
		*static_cast<int *>(user) = -1; // would race with counting_feed if the host still called it
	}
	released++;
}

static int effects_hook(void *, uint32_t id, gas_params *p) { // audio thread
	hooked++;
	p->fx_shelf_gain = 0.5f + (float)(id & 3);
	return (id & 1) != 0;
}

int main(int argc, char **argv) {
	const int rounds = argc > 1 ? atoi(argv[1]) : 1500;
	const int device_mode = argc > 2 ? atoi(argv[2]) : 0;
	const uint32_t F = 128;
	gas_ctx *ctx = gas_stub_ctx_create(64, F);
	gas_host *host = nullptr;
	if (gas_host_create(ctx, GAS_KIND_EFFECT, nullptr, 0, &host) != GAS_OK) {
		return 2;
	}
	static std::vector<gas_audio_frame> stream(1000, gas_audio_frame{ 0.5f, -0.5f });
	std::vector<int> budgets((size_t)rounds * 2 + 8, 0);
	std::atomic<bool> stop{ false };
	std::atomic<uint32_t> newest{ 0 };
	std::atomic<int> failures{ 0 };
	gas_host_set_release_fn(host, on_release, nullptr);
	gas_host_set_process_effects_fn(host, effects_hook, nullptr);

	std::thread audio([&] {
		std::vector<gas_audio_frame> out(F);
		while (!stop.load()) {
			const int rc = gas_host_get_mixed_frames(host, 0, out.data(), (int)F);
			if (rc != GAS_OK) {
				failures++;
			}
		}
		for (int i = 0; i < 64; i++) { // let everything ring out and be reaped
			gas_host_get_mixed_frames(host, 0, out.data(), (int)F);
		}
	});
	std::thread physics([&] { // a second control thread: parameters for whatever was started last, queries
		gas_params p{};
		p.pitch_scale = 1.0f;
		while (!stop.load()) {
			const uint32_t id = newest.load();
			if (id) {
				p.hrtf_gain = 0.5f;
				gas_host_set_spatializer_parameters(host, id, &p);
				(void)gas_host_is_playback_active(host, id);
				// pause / resume / position from the second control thread (audio_spatializer.cpp:115-122, :144-170)
				gas_host_set_playback_paused(host, id, (id & 2) != 0);
				(void)gas_host_is_playback_paused(host, id);
				uint64_t pos = 0;
				(void)gas_host_get_playback_position(host, id, &pos);
				gas_host_set_playback_paused(host, id, 0);
				gas_fx_settings fs{};
				fs.filter_cutoff_hz[0] = 1000.0f;
				gas_host_set_effect_settings(host, id, &fs);
			}
			// the context's slot allocator is open to control threads too (gas_amd.h): a throw-away slot per loop
			uint32_t probe = 0;
			if (gas_source_alloc(ctx, GAS_KIND_EFFECT, nullptr, 0, &probe) == GAS_OK) {
				gas_source_free(ctx, probe);
			}
			(void)gas_host_playback_count(host);
			(void)gas_host_collect_released(host);
			gas_host_set_process_effects_fn(host, effects_hook, nullptr); // re-registering while the audio thread runs
			gas_host_set_playback_disable_threshold_db(host, -80.0f);
		}
	});
	// main thread: start / parameters / stop
	gas_params p{};
	p.pitch_scale = 1.0f;
	p.hrtf_gain = 1.0f;
	std::vector<uint32_t> ids;
	for (int r = 0; r < rounds; r++) {
		uint32_t id = 0;
		int rc;
		if (device_mode) {
			rc = gas_host_start_playback_device_stream(host, 0, 0, &id);
		} else if (r & 1) {
			rc = gas_host_start_playback_array(host, stream.data(), (int64_t)stream.size(), &id);
		} else {
			budgets[(size_t)r] = 300 + (r % 7) * 100;
			rc = gas_host_start_playback(host, counting_feed, &budgets[(size_t)r], &id);
		}
		if (rc != GAS_OK) {
			failures++;
			continue;
		}
		newest.store(id);
		gas_host_set_spatializer_parameters(host, id, &p);
		ids.push_back(id);
		if (ids.size() > 24) { // keep the population bounded: stop the oldest (it may have ended on its own already)
			gas_host_stop_playback(host, ids.front());
			ids.erase(ids.begin());
		}
		if ((r & 3) == 0) { // pace the control thread so that thousands of callbacks interleave with the commands
			std::this_thread::sleep_for(std::chrono::microseconds(200));
		}
	}
	for (uint32_t id : ids) {
		gas_host_stop_playback(host, id);
	}
	while (gas_host_playback_count(host) > 0 && gas_stub_blocks(ctx) < 100000000ul) {
		std::this_thread::yield();
	}
	stop.store(true);
	audio.join();
	physics.join();
	const int left = gas_host_playback_count(host);
	gas_host_destroy(host);
	printf("callbacks %lu, playbacks left %d, failures %d, released %d of %d, hook calls %d\n", gas_stub_blocks(ctx), left, failures.load(), released.load(), rounds, hooked.load());
	gas_stub_ctx_destroy(ctx);
	return (left == 0 && failures.load() == 0 && released.load() == rounds && hooked.load() > 0) ? 0 : 1;
}
