// batched_spatializer_host.cpp -- see include/gas_amd_host.h.  Plain C++17, no engine, no HIP calls of its
// own: everything device-side goes through the C ABI (gas_process_block & friends).
#include <cmath>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "../../include/gas_amd_host.h"

namespace {

constexpr int LOOKAHEAD = GAS_LOOKAHEAD_BUFFER_SIZE;

// audio_spatializer.h:55-66 SpatialPlaybackListNode
struct PlaybackNode {
	uint32_t id = 0;
	uint32_t slot = 0;
	gas_host_stream_mix_fn mix = nullptr;
	void *user = nullptr;
	bool active = true;
	bool has_frames = true;
	bool has_params = false;
	gas_params params{};
	gas_audio_frame lookahead[LOOKAHEAD]{};
	bool device_stream = false; // sampled on the GPU (gas_host_start_playback_device_stream)
	// array-backed stream (gas_host_start_playback_array)
	const gas_audio_frame *array = nullptr;
	int64_t array_frames = 0, array_pos = 0;
};

int array_mix(void *user, gas_audio_frame *buffer, float /*rate_scale*/, int frames) {
	PlaybackNode *n = static_cast<PlaybackNode *>(user);
	int64_t left = n->array_frames - n->array_pos;
	int m = (int)(left < frames ? (left < 0 ? 0 : left) : frames);
	std::memcpy(buffer, n->array + n->array_pos, sizeof(gas_audio_frame) * (size_t)m);
	std::memset(buffer + m, 0, sizeof(gas_audio_frame) * (size_t)(frames - m));
	n->array_pos += m;
	return m;
}

// [ENGINE] Math::db_to_linear
float db_to_linear(float db) {
	return std::exp(db * 0.11512925464970228420089957273422f);
}

} // namespace

struct gas_host {
	gas_ctx *ctx = nullptr;
	int kind = 0;
	std::vector<int32_t> effects;
	int channel_count = 1; // should_mix_channels ? AudioServer channels : 1 (audio_spatializer.cpp:172-179)
	int ctx_channels = 1; // rows of the library's [C][F] output
	bool channel_mixed[GAS_MAX_CHANNELS_PER_BUS] = { true, true, true, true }; // :78-80
	float disable_threshold_db = -80.0f; // audio_spatializer.h:87
	uint32_t next_id = 1;
	// newest first, like SafeList::insert at head
	std::vector<std::unique_ptr<PlaybackNode>> list;
	// audio-thread buffers (playback_buffer / mix_buffer, audio_spatializer.h:78-81)
	std::vector<gas_audio_frame> playback_buffer; // [n + 64]
	std::vector<gas_audio_frame> rows; // [active][n]
	std::vector<uint32_t> slots;
	std::vector<PlaybackNode *> row_node;
	std::vector<float> peaks;
	std::vector<gas_audio_frame> mix_buffer; // [ctx_channels][n]
	bool mix_valid = false;
	int mode = 0; // 0 undecided, 1 callback/array playbacks, 2 device-stream playbacks
	std::vector<uint8_t> has_frames_out;

	PlaybackNode *find(uint32_t id) {
		for (auto &n : list) {
			if (n->id == id) {
				return n.get();
			}
		}
		return nullptr;
	}

	// audio_spatializer.cpp:367-408
	void fetch_source(PlaybackNode *pb, gas_audio_frame *buf, int n) {
		if (pb->has_frames) {
			for (int i = 0; i < LOOKAHEAD; i++) {
				buf[i] = pb->lookahead[i];
			}
			const float pitch_scale = pb->params.pitch_scale;
			const int mixed_frames = pb->mix(pb->user, &buf[LOOKAHEAD], pitch_scale, n);
			if (mixed_frames != n) {
				float fadeout_base = 0.96f;
				float fadeout_coefficient = 1;
				float buffer_size_float = (float)LOOKAHEAD;
				float buffer_linear_fade_idx = 0.0f;
				const int fade_limit = mixed_frames + LOOKAHEAD;
				for (int idx = mixed_frames; idx < n; idx++) {
					if (idx < fade_limit) {
						fadeout_coefficient *= fadeout_base;
						const float f = fadeout_coefficient * (buffer_size_float - buffer_linear_fade_idx) / buffer_size_float;
						buf[idx].left *= f;
						buf[idx].right *= f;
						buffer_linear_fade_idx += 1.0f;
					} else {
						buf[idx].left *= 0.0f;
						buf[idx].right *= 0.0f;
					}
				}
				pb->has_frames = false;
				// from now on the gate reads this playback's peak (audio_spatializer.cpp:464)
				gas_source_set_draining(ctx, pb->slot, 1);
			} else {
				for (int i = 0; i < LOOKAHEAD; i++) {
					pb->lookahead[i] = buf[n + i];
				}
			}
		} else {
			std::memset(buf, 0, sizeof(gas_audio_frame) * (size_t)(n + LOOKAHEAD)); // :407
		}
	}

	// audio_spatializer.cpp:326-471, one device launch group for all playbacks
	int mix_from_playback_list(int n) {
		mix_buffer.assign((size_t)ctx_channels * n, gas_audio_frame{ 0.0f, 0.0f }); // :335-343
		playback_buffer.resize((size_t)n + LOOKAHEAD);
		rows.clear();
		slots.clear();
		row_node.clear();
		if (mode == 2) {
			// device-resident streams: the library samples the windows itself and reports has_frames per row
			for (auto &up : list) {
				PlaybackNode *pb = up.get();
				if (pb->active && pb->has_params) {
					slots.push_back(pb->slot);
					row_node.push_back(pb);
				}
			}
			const uint32_t count = (uint32_t)slots.size();
			peaks.assign((size_t)count * 2 + 2, 0.0f);
			has_frames_out.assign((size_t)count + 1, 0);
			const int rc = gas_process_block_streams(ctx, slots.data(), count, (uint32_t)n, mix_buffer.data(), peaks.data(), has_frames_out.data(), GAS_MEM_HOST);
			if (rc != GAS_OK) {
				return rc;
			}
			const float threshold = db_to_linear(disable_threshold_db);
			for (uint32_t r = 0; r < count; r++) {
				PlaybackNode *pb = row_node[r];
				pb->has_frames = has_frames_out[r] != 0; // audio_spatializer.cpp:398
				if (!pb->has_frames) { // :464-469
					const float l = peaks[2 * r], rr = peaks[2 * r + 1];
					if ((rr > l ? rr : l) <= threshold) {
						pb->active = false;
					}
				}
			}
			return GAS_OK;
		}
		for (auto &up : list) {
			PlaybackNode *pb = up.get();
			if (!pb->active) { // :355-357
				continue;
			}
			if (!pb->has_params) { // :330 parameters.is_null(): nothing is mixed for it
				continue;
			}
			fetch_source(pb, playback_buffer.data(), n);
			rows.insert(rows.end(), playback_buffer.begin(), playback_buffer.begin() + n); // DSP consumes [0, n)
			slots.push_back(pb->slot);
			row_node.push_back(pb);
		}
		const uint32_t count = (uint32_t)slots.size();
		peaks.assign((size_t)count * 2 + 2, 0.0f);
		const int rc = gas_process_block(ctx, rows.data(), slots.data(), count, (uint32_t)n, mix_buffer.data(), peaks.data(), GAS_MEM_HOST);
		if (rc != GAS_OK) {
			return rc; // mix_buffer was zero-filled by the library
		}
		const float threshold = db_to_linear(disable_threshold_db);
		for (uint32_t r = 0; r < count; r++) { // :464-469
			PlaybackNode *pb = row_node[r];
			if (!pb->has_frames) {
				const float l = peaks[2 * r], rr = peaks[2 * r + 1];
				if ((rr > l ? rr : l) <= threshold) {
					pb->active = false;
				}
			}
		}
		return GAS_OK;
	}

	// audio_spatializer.cpp:473-492
	void manage_playback_state() {
		for (size_t i = 0; i < list.size();) {
			if (!list[i]->active) {
				gas_source_free(ctx, list[i]->slot); // deferred to the next block boundary by the library
				list.erase(list.begin() + (long)i);
			} else {
				i++;
			}
		}
	}

	// audio_spatializer.cpp:494-508
	bool check_channel_mixed(int channel) {
		if (channel_mixed[channel]) {
			for (bool &m : channel_mixed) {
				m = false;
			}
			channel_mixed[channel] = true;
			return true;
		}
		channel_mixed[channel] = true;
		return false;
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
	h->ctx_channels = (int)cfg.channel_count;
	h->channel_count = kind == GAS_KIND_3D_MIX ? (int)cfg.channel_count : 1;
	h->effects.assign(effects, effects + n_effects);
	*out_host = h;
	return GAS_OK;
}

void gas_host_destroy(gas_host *h) {
	if (!h) {
		return;
	}
	for (auto &n : h->list) {
		gas_source_free(h->ctx, n->slot);
	}
	delete h;
}

int gas_host_start_playback(gas_host *h, gas_host_stream_mix_fn mix, void *user, uint32_t *out_id) {
	if (!h || !mix || !out_id) {
		return GAS_ERR_INVALID_ARGUMENT;
	}
	if (h->mode == 2) {
		return GAS_ERR_KIND_MISMATCH;
	}
	h->mode = 1;
	std::unique_ptr<PlaybackNode> n(new (std::nothrow) PlaybackNode());
	if (!n) {
		return GAS_ERR_OUT_OF_MEMORY;
	}
	int rc = gas_source_alloc(h->ctx, h->kind, h->effects.data(), (uint32_t)h->effects.size(), &n->slot);
	if (rc != GAS_OK) {
		return rc;
	}
	n->id = h->next_id++;
	n->mix = mix;
	n->user = user;
	*out_id = n->id;
	if (h->list.empty()) { // first playback: every channel marked mixed so the next request remixes (:78-80)
		for (bool &m : h->channel_mixed) {
			m = true;
		}
	}
	h->list.insert(h->list.begin(), std::move(n)); // head insertion
	return GAS_OK;
}

int gas_host_start_playback_array(gas_host *h, const gas_audio_frame *stream, int64_t stream_frames, uint32_t *out_id) {
	if (!stream || stream_frames < 0) {
		return GAS_ERR_INVALID_ARGUMENT;
	}
	int rc = gas_host_start_playback(h, array_mix, nullptr, out_id);
	if (rc != GAS_OK) {
		return rc;
	}
	PlaybackNode *n = h->list.front().get();
	n->user = n;
	n->array = stream;
	n->array_frames = stream_frames;
	return GAS_OK;
}

int gas_host_start_playback_device_stream(gas_host *h, uint32_t stream, uint64_t start_frame, uint32_t *out_id) {
	if (!h || !out_id) {
		return GAS_ERR_INVALID_ARGUMENT;
	}
	if (h->mode == 1) {
		return GAS_ERR_KIND_MISMATCH;
	}
	std::unique_ptr<PlaybackNode> n(new (std::nothrow) PlaybackNode());
	if (!n) {
		return GAS_ERR_OUT_OF_MEMORY;
	}
	int rc = gas_source_alloc(h->ctx, h->kind, h->effects.data(), (uint32_t)h->effects.size(), &n->slot);
	if (rc != GAS_OK) {
		return rc;
	}
	rc = gas_source_bind_stream(h->ctx, n->slot, stream, start_frame);
	if (rc != GAS_OK) {
		gas_source_free(h->ctx, n->slot);
		return rc;
	}
	h->mode = 2;
	n->id = h->next_id++;
	n->device_stream = true;
	*out_id = n->id;
	if (h->list.empty()) {
		for (bool &m : h->channel_mixed) {
			m = true;
		}
	}
	h->list.insert(h->list.begin(), std::move(n));
	return GAS_OK;
}

int gas_host_stop_playback(gas_host *h, uint32_t id) {
	PlaybackNode *n = h ? h->find(id) : nullptr;
	if (!n) {
		return GAS_ERR_BAD_SLOT;
	}
	n->active = false;
	return GAS_OK;
}

int gas_host_set_spatializer_parameters(gas_host *h, uint32_t id, const gas_params *params) {
	PlaybackNode *n = h ? h->find(id) : nullptr;
	if (!n || !params) {
		return n ? GAS_ERR_INVALID_ARGUMENT : GAS_ERR_BAD_SLOT;
	}
	n->params = *params;
	n->has_params = true;
	return gas_params_publish(h->ctx, n->slot, params);
}

void gas_host_set_playback_disable_threshold_db(gas_host *h, float db) {
	if (h) {
		h->disable_threshold_db = db;
	}
}

int gas_host_is_playback_active(gas_host *h, uint32_t id) {
	PlaybackNode *n = h ? h->find(id) : nullptr;
	return n && n->active ? 1 : 0;
}

int gas_host_playback_count(gas_host *h) {
	return h ? (int)h->list.size() : 0;
}

void gas_host_bus_map(int should_mix_channels, int channel, const float bus_volume[GAS_MAX_CHANNELS_PER_BUS][2], const float mix_volumes[GAS_MAX_CHANNELS_PER_BUS][2], float out[GAS_MAX_CHANNELS_PER_BUS][2]) {
	for (int c = 0; c < GAS_MAX_CHANNELS_PER_BUS; c++) {
		if (should_mix_channels) { // audio_spatializer.cpp:295-313
			float left = 0.0f, right = 0.0f;
			if (c == channel) {
				if (mix_volumes[c][0] > 0.0) {
					left = bus_volume[c][0] / mix_volumes[c][0];
				}
				if (mix_volumes[c][1] > 0.0) {
					right = bus_volume[c][1] / mix_volumes[c][1];
				}
			}
			out[c][0] = left;
			out[c][1] = right;
		} else { // :314-318
			out[c][0] = mix_volumes[c][0];
			out[c][1] = mix_volumes[c][1];
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
	if (h->check_channel_mixed(channel)) {
		const int rc = h->mix_from_playback_list(frame_count);
		h->manage_playback_state();
		h->mix_valid = rc == GAS_OK;
		if (rc != GAS_OK) {
			std::memset(frames, 0, sizeof(gas_audio_frame) * (size_t)frame_count);
			return rc;
		}
	}
	const int visible = h->kind == GAS_KIND_3D_MIX ? h->channel_count : 1;
	if (channel >= visible) {
		return GAS_ERR_BAD_CHANNEL; // :521
	}
	if ((size_t)frame_count * h->ctx_channels != h->mix_buffer.size()) {
		return GAS_ERR_FRAME_COUNT; // :522
	}
	std::memcpy(frames, h->mix_buffer.data() + (size_t)channel * frame_count, sizeof(gas_audio_frame) * (size_t)frame_count);
	return GAS_OK;
}

} // extern "C"
