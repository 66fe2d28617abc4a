// batched_spatializer_host.cpp -- see include/gas_amd_host.h.  Plain C++17, no engine, no HIP calls of its
// own: everything device-side goes through the C ABI (gas_process_block & friends).
//
// Threading (the split of audio_spatializer.h:135-138, made explicit):
//   control threads (main / physics): start_playback*, stop_playback, set_spatializer_parameters,
//       set_playback_disable_threshold_db, is_playback_active, playback_count -- any number of threads, any time;
//   audio thread: get_mixed_frames, one caller.
// The control side never touches the playback list.  It hands work over through an inbox (one mutex, held for a
// push or a swap -- the role of the reference's parameter mutex, audio_spatializer.cpp:558-574) and flips per-playback
// atomics (the reference's SafeFlag active, audio_spatializer.h:57-66); the audio thread adopts new playbacks, applies
// parameters, reaps ended ones and is the ONLY thread that talks to the gas_ctx slot API (alloc / free / draining /
// bind), which is what include/gas_amd.h asks for.  Reaped nodes are deleted on a control thread (deferred delete,
// audio_spatializer.cpp:538-547): the audio thread only moves them to a graveyard.
#include <atomic>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

#include "../../include/gas_amd_host.h"

namespace {

constexpr int TAIL = GAS_LOOKAHEAD_BUFFER_SIZE; // frames a playback runs ahead of what the DSP sees (audio_spatializer.h:49)

// What control threads may look at once a playback has been handed to the audio thread.
struct Shared {
	std::atomic<bool> active{ true }; // cleared by stop_playback, by the silence gate, or by a start that failed
	// set_playback_paused per playback (the reference pauses an instance's own proxies on AudioServer,
	// audio_spatializer.cpp:115-122; with ONE proxy per bus for all players that would pause every player, so the pause
	// lives here): a paused playback is neither sampled nor mixed, keeps its slot state and lookahead, is not gated and
	// not reaped.  Read by the audio thread at the top of a callback.
	std::atomic<bool> paused{ false };
	std::atomic<uint64_t> position{ 0 }; // frames of the stream consumed so far ([ENGINE] AudioStreamPlayback::get_playback_position x mix rate); written by the audio thread
};

// One playback.  Built on a control thread, then owned by the audio thread until it is buried.
struct Playback {
	uint32_t id = 0;
	uint32_t slot = 0;
	bool has_slot = false;
	std::shared_ptr<Shared> shared;
	// where its frames come from: a callback (the engine's AudioStreamPlayback::mix), a caller-owned array, or a
	// device-resident stream the library samples itself
	gas_host_stream_mix_fn feed = nullptr;
	void *feed_user = nullptr;
	const gas_audio_frame *array = nullptr;
	int64_t array_frames = 0, array_pos = 0;
	bool on_device = false;
	uint32_t device_stream = 0;
	uint64_t device_start = 0; // first frame of this playback in its stream
	uint64_t consumed = 0; // frames taken from the source so far
	// audio-thread state
	bool stream_live = true; // the source still has frames (has_frames, audio_spatializer.h:63)
	bool have_params = false;
	float pitch_scale = 1.0f;
	gas_params params{}; // the row last published for this playback (what a process-effects hook may edit)
	gas_audio_frame tail[TAIL]{}; // the 64 frames sampled ahead last callback

	int pull(gas_audio_frame *dst, int frames) {
		if (feed) {
			const int got = feed(feed_user, dst, pitch_scale, frames);
			consumed += (uint64_t)(got > 0 ? got : 0);
			return got;
		}
		const int64_t left = array_frames - array_pos;
		const int got = (int)(left < frames ? (left < 0 ? 0 : left) : frames);
		std::memcpy(dst, array + array_pos, sizeof(gas_audio_frame) * (size_t)got);
		std::memset(dst + got, 0, sizeof(gas_audio_frame) * (size_t)(frames - got));
		array_pos += got;
		consumed += (uint64_t)got;
		return got;
	}
};

struct Command {
	enum Kind { START, PARAMS, FX_SETTINGS } kind = START;
	uint32_t id = 0;
	std::unique_ptr<Playback> playback; // START
	gas_params params{}; // PARAMS
	gas_fx_settings fx_settings{}; // FX_SETTINGS
};

// [ENGINE] Math::db_to_linear
float db_to_linear(float db) {
	return std::exp(db * 0.11512925464970228420089957273422f);
}

} // namespace

struct gas_host {
	gas_ctx *ctx = nullptr;
	int kind = 0;
	std::vector<int32_t> effects;
	int pairs_visible = 1; // should_mix_channels ? AudioServer channel pairs : 1 (audio_spatializer.cpp:172-179)
	int ctx_pairs = 1; // rows of the library's [C][F] output
	float fade_env[TAIL]; // end-of-stream ramp, audio_spatializer.cpp:382-392 (same f32 products)

	// ---- shared between threads ----
	std::atomic<uint32_t> next_id{ 1 };
	std::atomic<int> node_count{ 0 }; // playbacks started and not yet reaped
	std::atomic<int> mode{ 0 }; // 0 undecided, 1 callback/array playbacks, 2 device-stream playbacks
	std::atomic<float> disable_threshold_db{ -80.0f }; // audio_spatializer.h:87
	std::mutex reg_mu; // id -> shared state, for the control side's lookups
	std::unordered_map<uint32_t, std::shared_ptr<Shared>> registry;
	std::mutex inbox_mu; // commands for the audio thread, and the nodes it has finished with
	std::vector<Command> inbox;
	std::vector<std::unique_ptr<Playback>> graveyard;
	std::mutex hooks_mu; // the two callbacks below are set from a control thread
	gas_host_release_fn release_fn = nullptr; // told about every playback the host has finished with (control thread)
	void *release_user = nullptr;
	gas_host_process_effects_fn effects_fn = nullptr; // _process_effects (audio_spatializer_effect.cpp:39,90-92): called on the audio thread
	void *effects_user = nullptr;

	// ---- audio thread only ----
	// Stored OLDEST first and walked from the back: the same newest-first order as SafeList::insert at head (SURVEY.md
	// Appendix A item 7) gives, with an O(1) start instead of a head insertion into the vector.
	std::vector<std::unique_ptr<Playback>> list;
	std::unordered_map<uint32_t, Playback *> by_id;
	std::vector<Command> batch; // the inbox of this callback
	uint32_t served = 0xFu; // channel pairs handed out since the last mix; all set => the next request mixes (:78-80)
	std::vector<gas_audio_frame> window; // [n + 64]
	std::vector<gas_audio_frame> rows; // [playbacks][n]
	std::vector<uint32_t> slots;
	std::vector<Playback *> row_owner;
	std::vector<float> peaks;
	std::vector<uint8_t> live_out;
	std::vector<uint64_t> positions;
	std::vector<gas_audio_frame> mix; // [ctx_pairs][n]

	// ---------------------------------------------------------------- control side
	std::shared_ptr<Shared> lookup(uint32_t id) {
		std::lock_guard<std::mutex> lk(reg_mu);
		auto it = registry.find(id);
		return it == registry.end() ? nullptr : it->second;
	}

	// Node memory is released here, never on the audio thread.  A registered release function hears about every
	// playback first: from then on the host will not call its stream callback or touch its `user` again (the role of the
	// Ref<AudioStreamPlayback> the reference's list node owns until its deferred delete, audio_spatializer.cpp:538-547).
	int empty_graveyard() {
		std::vector<std::unique_ptr<Playback>> dead;
		{
			std::lock_guard<std::mutex> lk(inbox_mu);
			dead.swap(graveyard);
		}
		gas_host_release_fn fn;
		void *fn_user;
		{
			std::lock_guard<std::mutex> lk(hooks_mu);
			fn = release_fn;
			fn_user = release_user;
		}
		if (fn) {
			for (auto &pb : dead) {
				fn(fn_user, pb->id, pb->feed_user);
			}
		}
		return (int)dead.size();
	}

	int enqueue_start(std::unique_ptr<Playback> pb, uint32_t *out_id) {
		pb->id = next_id.fetch_add(1);
		pb->shared = std::make_shared<Shared>();
		*out_id = pb->id;
		{
			std::lock_guard<std::mutex> lk(reg_mu);
			registry.emplace(pb->id, pb->shared);
		}
		node_count.fetch_add(1);
		Command c;
		c.kind = Command::START;
		c.id = pb->id;
		c.playback = std::move(pb);
		std::lock_guard<std::mutex> lk(inbox_mu);
		inbox.push_back(std::move(c));
		return GAS_OK;
	}

	// ---------------------------------------------------------------- audio side
	void bury(std::unique_ptr<Playback> pb) {
		pb->shared->active.store(false);
		{
			std::lock_guard<std::mutex> lk(reg_mu);
			registry.erase(pb->id);
		}
		node_count.fetch_sub(1);
		std::lock_guard<std::mutex> lk(inbox_mu);
		graveyard.push_back(std::move(pb));
	}

	// New playbacks and parameters queued since the last callback, in the order they were issued.
	void adopt_inbox() {
		batch.clear();
		{
			std::lock_guard<std::mutex> lk(inbox_mu);
			batch.swap(inbox);
		}
		for (Command &c : batch) {
			if (c.kind == Command::START) {
				std::unique_ptr<Playback> pb = std::move(c.playback);
				// _instantiate_playback_data (audio_spatializer.cpp:69): the device-resident slot
				int rc = gas_source_alloc(ctx, kind, effects.data(), (uint32_t)effects.size(), &pb->slot);
				pb->has_slot = rc == GAS_OK;
				if (rc == GAS_OK && pb->on_device) {
					rc = gas_source_bind_stream(ctx, pb->slot, pb->device_stream, pb->device_start);
				}
				if (rc != GAS_OK) { // out of slots / unknown stream: the playback never becomes audible
					if (pb->has_slot) {
						gas_source_free(ctx, pb->slot);
					}
					bury(std::move(pb));
					continue;
				}
				// (the reference marks every channel pair "mixed" when the first playback starts, :78-80, so that the next
				// request mixes; here a playback is only ever adopted inside the request that starts a new mix)
				by_id[pb->id] = pb.get();
				list.push_back(std::move(pb)); // newest = last (walked from the back)
			} else if (c.kind == Command::FX_SETTINGS) {
				auto it = by_id.find(c.id);
				if (it != by_id.end()) {
					gas_fx_settings_publish(ctx, &it->second->slot, &c.fx_settings, 1); // snapshotted with the parameters (:328)
				}
			} else {
				auto it = by_id.find(c.id);
				if (it == by_id.end()) {
					continue; // reaped meanwhile
				}
				Playback *pb = it->second;
				pb->pitch_scale = c.params.pitch_scale;
				pb->have_params = true;
				pb->params = c.params;
				gas_params_publish(ctx, pb->slot, &c.params); // snapshotted by this callback's gas_process_block (:328)
			}
		}
		batch.clear();
	}

	// The window the DSP sees: last callback's 64-frame tail, then fresh frames (audio_spatializer.cpp:367-378).  When
	// the source runs dry inside it, the 64 frames after its last one ramp down and the rest is blanked (:380-398);
	// a multiply by zero keeps a NaN a NaN, as the reference's `*= 0.0` does.
	void build_window(Playback *pb, gas_audio_frame *w, int n) {
		if (!pb->stream_live) {
			std::memset(w, 0, sizeof(gas_audio_frame) * (size_t)(n + TAIL)); // effect tails ring out on silence (:405-408)
			return;
		}
		std::memcpy(w, pb->tail, sizeof(pb->tail));
		const int got = pb->pull(w + TAIL, n);
		if (got == n) {
			std::memcpy(pb->tail, w + n, sizeof(pb->tail));
			return;
		}
		const int ramp_end = got + TAIL < n ? got + TAIL : n;
		for (int i = got; i < ramp_end; i++) {
			w[i].left *= fade_env[i - got];
			w[i].right *= fade_env[i - got];
		}
		for (int i = ramp_end; i < n; i++) {
			w[i].left *= 0.0f;
			w[i].right *= 0.0f;
		}
		pb->stream_live = false;
		gas_source_set_draining(ctx, pb->slot, 1); // from now on the gate reads this playback's peak (:464)
	}

	// The silence gate (audio_spatializer.cpp:464-469) over the peaks the library returned.
	void gate(uint32_t count) {
		const float threshold = db_to_linear(disable_threshold_db.load());
		for (uint32_t r = 0; r < count; r++) {
			Playback *pb = row_owner[r];
			if (pb->stream_live) {
				continue;
			}
			const float l = peaks[2 * r], rr = peaks[2 * r + 1];
			if ((rr > l ? rr : l) <= threshold) {
				pb->shared->active.store(false);
			}
		}
	}

	// audio_spatializer.cpp:326-471 with ONE device launch group for all playbacks
	int mix_all(int n) {
		mix.assign((size_t)ctx_pairs * n, gas_audio_frame{ 0.0f, 0.0f }); // :335-343
		rows.clear();
		slots.clear();
		row_owner.clear();
		const bool device_streams = mode.load() == 2;
		if (!device_streams) {
			window.resize((size_t)n + TAIL);
		}
		gas_host_process_effects_fn fx_hook;
		void *fx_user;
		{
			std::lock_guard<std::mutex> lk(hooks_mu);
			fx_hook = effects_fn;
			fx_user = effects_user;
		}
		for (size_t k = list.size(); k-- > 0;) { // newest first
			Playback *pb = list[k].get();
			if (!pb->shared->active.load() || !pb->have_params || pb->shared->paused.load()) { // :355-357; :330 parameters.is_null(): nothing is mixed; paused: see Shared
				continue;
			}
			if (fx_hook && fx_hook(fx_user, pb->id, &pb->params)) { // process_effects() runs first thing in process_frames (audio_spatializer_effect.cpp:39)
				pb->pitch_scale = pb->params.pitch_scale;
				gas_params_publish(ctx, pb->slot, &pb->params);
			}
			if (!device_streams) {
				build_window(pb, window.data(), n);
				rows.insert(rows.end(), window.begin(), window.begin() + n); // the DSP consumes [0, n)
			}
			slots.push_back(pb->slot);
			row_owner.push_back(pb);
		}
		const uint32_t count = (uint32_t)slots.size();
		peaks.assign((size_t)count * 2 + 2, 0.0f);
		int rc;
		if (device_streams) { // the library samples the windows itself and reports has_frames per row
			live_out.assign((size_t)count + 1, 0);
			rc = gas_process_block_streams(ctx, slots.data(), count, (uint32_t)n, mix.data(), peaks.data(), live_out.data(), GAS_MEM_HOST);
			if (rc == GAS_OK) {
				positions.resize((size_t)count + 1);
				const bool have_pos = gas_stream_positions(ctx, count, positions.data()) == GAS_OK; // the library's mirror of the device cursors
				for (uint32_t r = 0; r < count; r++) {
					Playback *pb = row_owner[r];
					if (have_pos && positions[r] >= pb->device_start) {
						pb->consumed = positions[r] - pb->device_start;
					}
					pb->stream_live = live_out[r] != 0; // audio_spatializer.cpp:398
				}
			}
		} else {
			rc = gas_process_block(ctx, rows.data(), slots.data(), count, (uint32_t)n, mix.data(), peaks.data(), GAS_MEM_HOST);
		}
		if (rc != GAS_OK) {
			return rc; // the library zero-filled the mix
		}
		for (uint32_t r = 0; r < count; r++) {
			row_owner[r]->shared->position.store(row_owner[r]->consumed, std::memory_order_relaxed);
		}
		gate(count);
		return GAS_OK;
	}

	// audio_spatializer.cpp:473-492: inactive playbacks leave the list; their slot is freed at the next block boundary
	void reap() { // one stable compaction pass, whatever the number of playbacks that ended
		size_t keep = 0;
		for (size_t i = 0; i < list.size(); i++) {
			if (list[i]->shared->active.load()) {
				if (keep != i) {
					list[keep] = std::move(list[i]);
				}
				keep++;
				continue;
			}
			std::unique_ptr<Playback> pb = std::move(list[i]);
			by_id.erase(pb->id);
			gas_source_free(ctx, pb->slot);
			bury(std::move(pb));
		}
		list.resize(keep);
	}

	// audio_spatializer.cpp:494-508 as a bit set: a pair that was already handed out since the last mix starts a new
	// callback (and is the only one handed out so far); otherwise it is just marked.
	bool starts_new_callback(int channel) {
		const uint32_t bit = 1u << channel;
		const bool again = (served & bit) != 0;
		served = again ? bit : (served | bit);
		return again;
	}
};

extern "C" {

int gas_host_create(gas_ctx *ctx, int kind, const int32_t *effects, uint32_t n_effects, gas_host **out_host) {
	if (!ctx || !out_host || n_effects > GAS_MAX_EFFECTS || (n_effects && !effects)) {
		return GAS_ERR_INVALID_ARGUMENT;
	}
	// probe the kind/chain (and the context) with a throw-away slot
	uint32_t probe = 0;
	int rc = gas_source_alloc(ctx, kind, effects, n_effects, &probe);
	if (rc != GAS_OK) {
		return rc;
	}
	gas_source_free(ctx, probe);
	gas_host *h = new (std::nothrow) gas_host();
	if (!h) {
		return GAS_ERR_OUT_OF_MEMORY;
	}
	gas_config cfg;
	rc = gas_ctx_get_config(ctx, &cfg);
	if (rc != GAS_OK) {
		delete h;
		return rc;
	}
	h->ctx = ctx;
	h->kind = kind;
	h->ctx_pairs = (int)cfg.channel_count;
	h->pairs_visible = kind == GAS_KIND_3D_MIX ? (int)cfg.channel_count : 1;
	h->effects.assign(effects, effects + n_effects);
	h->list.reserve(1024); // starts do not reallocate on the audio thread until the population passes these
	h->by_id.reserve(2048);
	// ramp k = 0.96^(k+1) * (64 - k) / 64, built with the reference's f32 operations in its order
	float coefficient = 1.0f, step = 0.0f;
	for (int k = 0; k < TAIL; k++) {
		coefficient *= 0.96f;
		h->fade_env[k] = coefficient * ((float)TAIL - step) / (float)TAIL;
		step += 1.0f;
	}
	*out_host = h;
	return GAS_OK;
}

void gas_host_destroy(gas_host *h) { // no thread may be inside the host any more
	if (!h) {
		return;
	}
	for (auto &pb : h->list) {
		gas_source_free(h->ctx, pb->slot);
	}
	{ // everything still on the list or in the inbox is finished with as well
		std::lock_guard<std::mutex> lk(h->inbox_mu);
		for (auto &pb : h->list) {
			h->graveyard.push_back(std::move(pb));
		}
		for (Command &c : h->inbox) {
			if (c.playback) {
				h->graveyard.push_back(std::move(c.playback));
			}
		}
	}
	h->list.clear();
	h->empty_graveyard();
	delete h;
}

static int start_common(gas_host *h, std::unique_ptr<Playback> pb, int want_mode, uint32_t *out_id) {
	int undecided = 0;
	if (!h->mode.compare_exchange_strong(undecided, want_mode) && undecided != want_mode) {
		return GAS_ERR_KIND_MISMATCH; // a host serves callback/array playbacks or device streams, not both
	}
	h->empty_graveyard();
	return h->enqueue_start(std::move(pb), out_id);
}

int gas_host_start_playback(gas_host *h, gas_host_stream_mix_fn mix, void *user, uint32_t *out_id) {
	if (!h || !mix || !out_id) {
		return GAS_ERR_INVALID_ARGUMENT;
	}
	std::unique_ptr<Playback> pb(new (std::nothrow) Playback());
	if (!pb) {
		return GAS_ERR_OUT_OF_MEMORY;
	}
	pb->feed = mix;
	pb->feed_user = user;
	return start_common(h, std::move(pb), 1, out_id);
}

int gas_host_start_playback_array(gas_host *h, const gas_audio_frame *stream, int64_t stream_frames, uint32_t *out_id) {
	if (!h || !out_id || !stream || stream_frames < 0) {
		return GAS_ERR_INVALID_ARGUMENT;
	}
	std::unique_ptr<Playback> pb(new (std::nothrow) Playback());
	if (!pb) {
		return GAS_ERR_OUT_OF_MEMORY;
	}
	pb->array = stream;
	pb->array_frames = stream_frames;
	return start_common(h, std::move(pb), 1, out_id);
}

int gas_host_start_playback_device_stream(gas_host *h, uint32_t stream, uint64_t start_frame, uint32_t *out_id) {
	if (!h || !out_id) {
		return GAS_ERR_INVALID_ARGUMENT;
	}
	std::unique_ptr<Playback> pb(new (std::nothrow) Playback());
	if (!pb) {
		return GAS_ERR_OUT_OF_MEMORY;
	}
	pb->on_device = true;
	pb->device_stream = stream;
	pb->device_start = start_frame;
	return start_common(h, std::move(pb), 2, out_id);
}

int gas_host_stop_playback(gas_host *h, uint32_t id) {
	std::shared_ptr<Shared> s = h ? h->lookup(id) : nullptr;
	if (!s) {
		return GAS_ERR_BAD_SLOT;
	}
	s->active.store(false); // the audio thread reaps it (audio_spatializer.cpp:98-113, :473-482)
	return GAS_OK;
}

int gas_host_set_spatializer_parameters(gas_host *h, uint32_t id, const gas_params *params) {
	if (!h || !h->lookup(id)) {
		return GAS_ERR_BAD_SLOT;
	}
	if (!params) {
		return GAS_ERR_INVALID_ARGUMENT;
	}
	Command c;
	c.kind = Command::PARAMS;
	c.id = id;
	c.params = *params;
	std::lock_guard<std::mutex> lk(h->inbox_mu);
	h->inbox.push_back(std::move(c));
	return GAS_OK;
}

int gas_host_set_effect_settings(gas_host *h, uint32_t id, const gas_fx_settings *settings) {
	if (!h || !h->lookup(id)) {
		return GAS_ERR_BAD_SLOT;
	}
	if (!settings) {
		return GAS_ERR_INVALID_ARGUMENT;
	}
	Command c;
	c.kind = Command::FX_SETTINGS;
	c.id = id;
	c.fx_settings = *settings;
	std::lock_guard<std::mutex> lk(h->inbox_mu);
	h->inbox.push_back(std::move(c));
	return GAS_OK;
}

void gas_host_set_playback_disable_threshold_db(gas_host *h, float db) {
	if (h) {
		h->disable_threshold_db.store(db);
	}
}

int gas_host_is_playback_active(gas_host *h, uint32_t id) {
	std::shared_ptr<Shared> s = h ? h->lookup(id) : nullptr;
	return s && s->active.load() ? 1 : 0;
}

int gas_host_set_playback_paused(gas_host *h, uint32_t id, int paused) {
	std::shared_ptr<Shared> s = h ? h->lookup(id) : nullptr;
	if (!s) {
		return GAS_ERR_BAD_SLOT;
	}
	s->paused.store(paused != 0); // seen by the audio thread at the top of the next callback
	return GAS_OK;
}

int gas_host_is_playback_paused(gas_host *h, uint32_t id) {
	std::shared_ptr<Shared> s = h ? h->lookup(id) : nullptr;
	return s && s->active.load() && s->paused.load() ? 1 : 0;
}

int gas_host_get_playback_position(gas_host *h, uint32_t id, uint64_t *out_frames) {
	if (!out_frames) {
		return GAS_ERR_INVALID_ARGUMENT;
	}
	std::shared_ptr<Shared> s = h ? h->lookup(id) : nullptr;
	if (!s) {
		*out_frames = 0; // audio_spatializer.cpp:152-155: an unknown playback reports 0
		return GAS_ERR_BAD_SLOT;
	}
	*out_frames = s->position.load(std::memory_order_relaxed);
	return GAS_OK;
}

int gas_host_set_release_fn(gas_host *h, gas_host_release_fn fn, void *user) {
	if (!h) {
		return GAS_ERR_INVALID_ARGUMENT;
	}
	std::lock_guard<std::mutex> lk(h->hooks_mu);
	h->release_fn = fn;
	h->release_user = user;
	return GAS_OK;
}

int gas_host_collect_released(gas_host *h) {
	return h ? h->empty_graveyard() : GAS_ERR_INVALID_ARGUMENT;
}

int gas_host_set_process_effects_fn(gas_host *h, gas_host_process_effects_fn fn, void *user) {
	if (!h) {
		return GAS_ERR_INVALID_ARGUMENT;
	}
	std::lock_guard<std::mutex> lk(h->hooks_mu);
	h->effects_fn = fn; // the audio thread picks the pair up at the top of its next callback
	h->effects_user = user;
	return GAS_OK;
}

int gas_host_playback_count(gas_host *h) {
	return h ? h->node_count.load() : 0;
}

void gas_host_bus_map(int should_mix_channels, int channel, const float bus_volume[GAS_MAX_CHANNELS_PER_BUS][2], const float mix_volumes[GAS_MAX_CHANNELS_PER_BUS][2], float out[GAS_MAX_CHANNELS_PER_BUS][2]) {
	for (int c = 0; c < GAS_MAX_CHANNELS_PER_BUS; c++) {
		for (int ear = 0; ear < 2; ear++) {
			float f;
			if (!should_mix_channels) {
				f = mix_volumes[c][ear]; // audio_spatializer.cpp:314-318: AudioServer applies the mix volume
			} else if (c == channel && mix_volumes[c][ear] > 0.0) {
				f = bus_volume[c][ear] / mix_volumes[c][ear]; // :295-313: the mix volume is in the frames already
			} else {
				f = 0.0f;
			}
			out[c][ear] = f;
		}
	}
}

int gas_host_get_mixed_frames(gas_host *h, int channel, gas_audio_frame *frames, int frame_count) {
	if (!h || !frames || frame_count <= 0) {
		return GAS_ERR_INVALID_ARGUMENT;
	}
	if (channel < 0 || channel >= GAS_MAX_CHANNELS_PER_BUS) {
		return GAS_ERR_BAD_CHANNEL;
	}
	if (h->starts_new_callback(channel)) {
		h->adopt_inbox();
		const int rc = h->mix_all(frame_count);
		h->reap();
		if (rc != GAS_OK) {
			std::memset(frames, 0, sizeof(gas_audio_frame) * (size_t)frame_count);
			return rc;
		}
	}
	if (channel >= h->pairs_visible) {
		return GAS_ERR_BAD_CHANNEL; // :521
	}
	if ((size_t)frame_count * h->ctx_pairs != h->mix.size()) {
		return GAS_ERR_FRAME_COUNT; // :522
	}
	std::memcpy(frames, h->mix.data() + (size_t)channel * frame_count, sizeof(gas_audio_frame) * (size_t)frame_count);
	return GAS_OK;
}

} // extern "C"
