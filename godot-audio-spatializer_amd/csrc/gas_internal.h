// gas_internal.h -- context layout and kernel launchers shared by the csrc/ translation units.
// Not part of the ABI (include/gas_amd.h is).
#pragma once

#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/gas_amd.h"

static_assert(sizeof(gas_params) == 128, "gas_params is a 128-byte POD");
static_assert(sizeof(gas_audio_frame) == 8, "AudioFrame is 2 x f32");

// ---------------------------------------------------------------------------
// Device-resident SpatializerPlaybackData (audio_spatializer_3d.h:85-99,
// audio_spatializer_effect.h:68-76), struct-of-arrays by "stream":
//   stream = (slot * 4 + channel_pair) * 2 + ear      (audio_spatializer_3d.cpp:887-894)
// so a wave of consecutive slots reads every field coalesced.
// ---------------------------------------------------------------------------
enum { GAS_BQ_FIELDS = 10 }; // b0 b1 b2 a1 a2 ha1 ha2 hb1 hb2 prev_vol
enum { BQ_B0 = 0, BQ_B1, BQ_B2, BQ_A1, BQ_A2, BQ_HA1, BQ_HA2, BQ_HB1, BQ_HB2, BQ_PREV };

struct gas_dev_state {
	float *bq; // [GAS_BQ_FIELDS][max_sources * 8]
	size_t bq_stride; // max_sources * 8
	float *hrtf_hist; // [max_sources][hist_len]   gained mono history
	float *hrtf_prev_gain; // [max_sources]
	uint32_t *hrtf_prev_dir; // [max_sources] previous callback's direction + 1 (0 = none), GAS_FLAG_HRTF_CROSSFADE
	gas_audio_frame *er_ring; // [max_sources][er_ring_frames]
	uint32_t *er_pos; // [max_sources]
	gas_params *params; // [max_sources]
	gas_fx_settings *fxs; // [max_sources] settings of the engine-effect kinds (GAS_FX_LOWPASS .. GAS_FX_AMPLIFY), by chain position
	uint8_t *was_further; // [max_sources] was_further_than_max_distance_last_frame (audio_spatializer_3d.h:118)
};

// Device-resident playback cursor: what SpatialPlaybackListNode + the engine's sampler hold.
struct gas_cursor {
	const void *pcm; // stream base in HBM
	uint64_t frames; // stream length
	uint64_t pos; // frames consumed so far (fresh-frame cursor; the DSP sees pos - 64)
	uint64_t start; // first frame of this playback: the lookahead in front of it is zero (audio_spatializer.cpp:61-63)
	uint32_t format_channels; // format << 8 | channels
	uint32_t has_frames; // audio_spatializer.h:63
	// resampled playbacks ([ENGINE] AudioStreamPlaybackResampled): the engine's mix_offset, 16.16 fixed point, in frames
	// of the stream; and where / how fast the previous callback ran, from which its last 64 outputs (this callback's
	// lookahead, audio_spatializer.cpp:369-373) are regenerated instead of being stored
	uint64_t fp_pos, fp_prev_pos;
	uint32_t prev_inc;
	uint32_t resampled; // 0 plain, 1 resampled and never mixed yet (zeroed lookahead, :61-63), 2 resampled
};

// What a launch group (one kind/chain) needs.
struct gas_group_args {
	const gas_audio_frame *src; // [n_rows_total][F]
	const uint32_t *rows; // [n] row of src per group entry, or nullptr (= identity)
	const uint32_t *slots; // [n] slot per group entry, or nullptr: slot = slot_base + entry (contiguous range)
	uint32_t slot_base;
	uint32_t n;
	float *peaks; // [n_rows_total][2]
	const uint32_t *peak_rows = nullptr; // k_hrtf_uni: [n] row of `peaks` per group entry when it differs from the row of src (a staged chain's last stage reads dense rows but reports into the callback's rows); nullptr = the src row
	const uint32_t *order = nullptr; // [n] processing order (k_hrtf_ols: entries grouped by HRIR direction; k_hrtf_uni: XCD-affine, k_xcd_order), or nullptr = entry order
};

// GAS_FLAG_PIPELINED_MIX: the final sum of the PREVIOUS callback's partial mixes (exactly k_mix_reduce's job for one
// channel pair), carried out by otherwise idle waves of this callback's k_hrtf_ols launch.  partials == nullptr: none.
#define GAS_HRTF_JOB_WAVES 4 // waves of a k_hrtf_ols workgroup that can each sum one output column of the previous callback
struct gas_deferred_reduce {
	const float *partials = nullptr; // [p_count][elems]
	uint32_t p_count = 0;
	uint32_t elems = 0; // F * 2
	float *out = nullptr;
};

enum gas_biquad_mode {
	GAS_MODE_MIX_CHANNEL = 0, // audio_spatializer_3d.cpp:554-609
	GAS_MODE_PROCESS_FRAMES = 1, // audio_spatializer_3d.cpp:491-552
	GAS_MODE_FX_HIGHSHELF = 2, // [ENGINE] AudioEffectFilterInstance::process, 1 stage
	GAS_MODE_COPY = 3, // empty effect chain, audio_spatializer_effect.cpp:41-46
	GAS_MODE_FX_FILTER = 4, // [ENGINE] AudioEffectFilterInstance::process of the other AudioFilterSW modes, 1 stage; settings from gas_fx_settings
	GAS_MODE_FX_AMPLIFY = 5, // [ENGINE] AudioEffectAmplifyInstance::process
};

struct gas_hrtf_table {
	float4 *spec; // [dirs][4][64] = (HL.re, HL.im, HR.re, HR.im) of bin lane + 64 j < 256, pre-scaled by 1/512; Nyquist in DC.imag
	uint32_t dirs;
};

static_assert(sizeof(gas_bus_route) == 8 + 32 + 4 * GAS_MAX_MORE_SENDS + 32 * GAS_MAX_MORE_SENDS, "gas_bus_route layout");

__host__ __device__ inline gas_bus_route gas_bus_route_default() { // a slot that never got a route: dry bus 0, no send
	gas_bus_route r{};
	r.dry_bus = 0;
	r.send_bus = GAS_BUS_NONE;
	for (int k = 0; k < GAS_MAX_MORE_SENDS; k++) {
		r.more_bus[k] = GAS_BUS_NONE;
	}
	return r;
}

// What a source sends to bus b for channel pair c and ear: dry 1 + the sends that name b, in field order.
__host__ __device__ inline float gas_bus_weight(const gas_bus_route &r, uint32_t b, int c, int ear) {
	float w = (r.dry_bus == b ? 1.0f : 0.0f) + (r.send_bus == b ? r.send[c][ear] : 0.0f);
#pragma unroll
	for (int k = 0; k < GAS_MAX_MORE_SENDS; k++) {
		w += r.more_bus[k] == b ? r.more_send[k][c][ear] : 0.0f;
	}
	return w;
}

// Several output buses: per-slot routes and how many buses the launch writes.  n_buses <= 1 is the
// single mix of gas_process_block (routes unused).  Partial layout with buses: plane (b * channel_count + c).
struct gas_bus_args {
	const gas_bus_route *routes = nullptr; // [max_sources], slot-indexed
	uint32_t n_buses = 1;
};

// Launchers (each only enqueues on `stream`; geometry is validated by the caller).
// Returns the number of partial mixes (per channel) it writes into `partials`
// ([C][P][F*2] floats, row stride P_stride).
uint32_t gas_biquad_partials(uint32_t n); // P for n sources
hipError_t gas_launch_biquad_mix(hipStream_t stream, int mode, const gas_group_args &g, const gas_dev_state &st, uint32_t frames, uint32_t channel_begin, uint32_t channel_count, float mix_rate, float *partials, uint32_t p_offset, uint32_t p_stride, float *rows_out = nullptr /* non-null: per-source rows instead of the partial mix */, const gas_bus_args &buses = gas_bus_args(), int fx_kind = 0 /* GAS_MODE_FX_FILTER: which GAS_FX_* filter */);

// k_shelf_scan.hip: a rows-out filter stage of a staged chain as a parallel scan over the frames (one wave per source),
// for callbacks too small to hide the serial recurrence (chosen inside gas_launch_biquad_mix)
bool gas_shelf_scan_applies(int mode, uint32_t n, uint32_t frames);
hipError_t gas_launch_shelf_scan(hipStream_t stream, int mode, const gas_group_args &g, const gas_dev_state &st, uint32_t frames, uint32_t chain_pos, float mix_rate, float *rows_out, int fx_kind);

// k_biquad_pipe.hip: the same arithmetic as an eight-wave software pipeline per 32 sources, for callbacks with fewer
// workgroups than CUs (chosen inside gas_launch_biquad_mix)
bool gas_biquad_uses_pipe(int mode, uint32_t n, uint32_t channel_count, uint32_t frames, bool rows_out); // the launcher's choice
hipError_t gas_launch_biquad_pipe(hipStream_t stream, int mode, const gas_group_args &g, const gas_dev_state &st, uint32_t frames, uint32_t channel_begin, uint32_t channel_count, float mix_rate, float *partials, uint32_t p_offset, uint32_t p_stride, const gas_bus_args &buses = gas_bus_args());

struct gas_hrtf_launch_plan {
	uint32_t wgs_fd, wgs_pk; // workgroups = partial mixes written
};
void gas_hrtf_plan(uint32_t n_fd, uint32_t n_pk, gas_hrtf_launch_plan *plan);
uint32_t gas_hrtf_partials(uint32_t n); // workgroups of a single-path launch (k_er_only, k_hrtf_rows, k_rows_accumulate)
hipError_t gas_launch_hrtf_ols(hipStream_t stream, bool with_er, bool crossfade, bool runs /* sum runs of equal directions before the FFT */, const gas_group_args &g_fd, const gas_group_args &g_pk, const gas_dev_state &st, const gas_hrtf_table &tab, const float2 *twiddles, uint32_t frames, uint32_t hist_len, uint32_t er_ring_frames, float *partials, uint32_t p_offset, gas_cursor *cursors /* non-null: sample the bound streams in the kernel */, const float *fade_env, const gas_params *fresh /* non-null: unscattered parameter rows in row order */, const gas_deferred_reduce &job = gas_deferred_reduce());
// k_hrtf_uni.hip: all plain [HRTF] sources of a callback in one uniform launch; peak_bits (bit per group entry, or
// nullptr) / peak_all say which sources also get their exact output peak
uint32_t gas_hrtf_uni_partials(uint32_t n); // workgroups (= partial mixes) of a k_hrtf_uni launch

// k_hrtf_multi.hip: K consecutive callbacks of one unchanged plain-[HRTF] list in one launch (GAS_FLAG_PIPELINED_MIX)
#define GAS_HRTF_MULTI_MAX_BLOCKS 16
struct gas_hrtf_blocks {
	uint32_t k = 0; // blocks in this launch
	const gas_audio_frame *src[GAS_HRTF_MULTI_MAX_BLOCKS] = {}; // [n][F] rows of block b
	const gas_params *fresh[GAS_HRTF_MULTI_MAX_BLOCKS] = {}; // device-published parameter rows taking effect at block b, or nullptr
	float *peaks[GAS_HRTF_MULTI_MAX_BLOCKS] = {}; // [n][2] of block b
	uint32_t p_offset[GAS_HRTF_MULTI_MAX_BLOCKS] = {}; // first partial row of block b (workgroup w writes row p_offset[b] + w)
	gas_deferred_reduce job[GAS_HRTF_MULTI_MAX_BLOCKS]; // pending sums of EARLIER launches carried by block b's idle waves
	const gas_params *last_fresh = nullptr; // the last non-null fresh[]: written through to the slot table at the end
};
// several independent deterministic sums (k_mix_reduce's) in one launch
struct gas_reduce_jobs {
	uint32_t count = 0;
	const float *partials[GAS_HRTF_MULTI_MAX_BLOCKS] = {}; // [p_count][F * 2] each
	uint32_t p_count[GAS_HRTF_MULTI_MAX_BLOCKS] = {};
	float *out[GAS_HRTF_MULTI_MAX_BLOCKS] = {};
};
hipError_t gas_launch_mix_reduce_jobs(hipStream_t stream, const gas_reduce_jobs &jobs, uint32_t frames);
bool gas_hrtf_multi_hist_in_lds(uint32_t n, uint32_t frames); // whether a launch over n sources keeps the history rows in LDS between its blocks
hipError_t gas_launch_hrtf_multi(hipStream_t stream, const gas_group_args &g, const gas_hrtf_blocks &mb, const uint32_t *peak_bits, bool peak_all, const gas_dev_state &st, const gas_hrtf_table &tab, const float2 *twiddles, uint32_t frames, uint32_t hist_len, float *partials);
hipError_t gas_launch_hrtf_uni(hipStream_t stream, const gas_group_args &g, const uint32_t *peak_bits, bool peak_all, const gas_dev_state &st, const gas_hrtf_table &tab, const float2 *twiddles, uint32_t frames, uint32_t hist_len, float *partials, uint32_t p_offset, gas_cursor *cursors, const float *fade_env, const gas_params *fresh, const gas_deferred_reduce &job = gas_deferred_reduce(), const gas_bus_route *routes = nullptr /* non-null: the two-bus form */, uint32_t bus_rows = 0, uint32_t bus_base = 0 /* the launch's pair of buses: bus_base, bus_base + 1 */, bool commit = true /* false: leave history / previous gain / peaks to a later pass */, uint32_t er_ring_frames = 0 /* non-zero: the chain [EARLY_REFLECTIONS, HRTF] */, uint32_t peak_from = 0xffffffffu /* entries from here on report their exact peak */, uint32_t peak_bit_base = 0 /* entry e's bit in peak_bits is bit e + peak_bit_base */, uint32_t flt_kind = 0 /* non-zero: the chain [this one-biquad kind, HRTF] */, uint32_t flt_pos = 0 /* its chain position (processor state, effect settings) */, float mix_rate = 0.0f);
hipError_t gas_launch_er_only(hipStream_t stream, const gas_group_args &g, const gas_dev_state &st, uint32_t frames, uint32_t er_ring_frames, float *partials, uint32_t p_offset, uint32_t p_stride, gas_audio_frame *rows_out = nullptr);
// stages of a general effect chain (rows in -> rows out) and its final mix
hipError_t gas_launch_hrtf_rows(hipStream_t stream, bool crossfade, const gas_group_args &g, const gas_dev_state &st, const gas_hrtf_table &tab, const float2 *twiddles, uint32_t frames, gas_audio_frame *rows_out);
hipError_t gas_launch_rows_accumulate(hipStream_t stream, const gas_group_args &g, uint32_t frames, float *partials, uint32_t p_offset);
hipError_t gas_launch_rows_accumulate_buses(hipStream_t stream, const gas_group_args &g, uint32_t frames, const gas_bus_route *routes, uint32_t n_buses, uint32_t bus_rows, float *partials, uint32_t p_offset); // bus b's partial rows: [b * bus_rows + p_offset + workgroup]

hipError_t gas_launch_hrtf_table(hipStream_t stream, const float *d_hrir, uint32_t dirs, uint32_t taps, const float2 *twiddles, float4 *spec);
hipError_t gas_launch_hrtf_regrid(hipStream_t stream, const float *d_positions, const float *d_hrir, uint32_t m, uint32_t taps, uint32_t az_steps, uint32_t el_steps, int interpolation, float *d_out /* [az_steps * el_steps][2][GAS_HRTF_TAPS] */);
void gas_make_twiddles(float2 *host_tw /* [64][16] */);

hipError_t gas_launch_mix_reduce(hipStream_t stream, const float *partials, uint32_t p_count, uint32_t p_stride, uint32_t channels, uint32_t frames, gas_audio_frame *out);
hipError_t gas_launch_scatter_params(hipStream_t stream, gas_params *table, const gas_params *upload, const uint32_t *slots, uint32_t n);
hipError_t gas_launch_calc_spatialization(hipStream_t stream, const gas_spatializer3d_config *cfgs, const uint32_t *cfg_index, const gas_source_pose *poses, const gas_listener *listeners, uint32_t n_listeners, const uint32_t *slots, uint32_t n, gas_params *table, uint8_t *was_further, gas_params *out_params, const gas_area_send *areas = nullptr, const float *listener_area_pos = nullptr, gas_audio_frame *out_reverb = nullptr);
hipError_t gas_launch_sample_sources(hipStream_t stream, gas_cursor *cursors, const uint32_t *slots, uint32_t n, uint32_t frames, const float *fade_env, gas_audio_frame *rows, const uint32_t *row_inc /* 16.16 step per row for resampled playbacks, or nullptr */);
hipError_t gas_launch_noop(hipStream_t stream); // event-timer calibration
hipError_t gas_launch_stream_probe(hipStream_t stream, const void *rd, uint64_t rd_bytes, void *wr, uint64_t wr_bytes, uint32_t workgroups, uint32_t unroll, float *sink); // copy-bandwidth ceiling
#define GAS_DIR_ORDER_SEGMENT 8192
#define GAS_DIR_ORDER_MIN_SOURCES 512
bool gas_dir_order_supported(uint32_t dirs);
// k_xcd_order (k_misc.hip): XCD-affine processing order for a k_hrtf_uni launch of hrtf_wgs workgroups (multiple of 8)
#define GAS_XCD_ORDER_AUTO_MIN 0xFFFFFFFFu // callbacks of at least this many sources are ordered without the flag: never (measured: no net gain at any size); the environment variable of the same name overrides for experiments
hipError_t gas_launch_xcd_order(hipStream_t stream, const gas_group_args &g, const gas_params *params, const gas_params *fresh, uint32_t dirs, uint32_t hrtf_wgs, uint32_t waves_per_wg, uint32_t *order);
uint32_t gas_hrtf_uni_waves();
bool gas_hrtf_uni_twelve(uint32_t n, bool streams, bool buses); // whether a callback of n plain-[HRTF] sources runs k_hrtf_uni's twelve-wave form (its sums differ from the eight-wave form's in the last bits)
hipError_t gas_launch_dir_order(hipStream_t stream, const gas_group_args &g, const gas_params *params, const gas_params *fresh, uint32_t dirs, uint32_t *order);
hipError_t gas_launch_zero_slot(hipStream_t stream, const gas_dev_state &st, uint32_t slot, uint32_t hist_len, uint32_t er_ring_frames);
