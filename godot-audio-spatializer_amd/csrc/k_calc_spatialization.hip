// k_calc_spatialization.hip -- SURVEY.md 8f#1: the arithmetic of
// AudioSpatializerInstance3D::calculate_spatialization (audio_spatializer_3d.cpp:277-489) for every source
// of a physics tick in one launch: one lane per source, listeners in a small uniform array.
// Mirrors oracle/gas_oracle.c::gaso_calc_spatialization statement by statement (float where the reference
// uses real_t = float, double in the pan law :104-109 and where the engine's Math:: overloads promote).
// Bound: trivially HBM (48 B pose in, 56 B of parameters out per source); launch-latency-bound below ~10^5 sources.
#include "gas_internal.h"

// No FMA contraction here: the reference feeds an un-normalised direction into SPCAP (:391), where 1 + dot can cancel
// catastrophically; keeping every product and sum separately rounded keeps the device bit-compatible with the
// op-for-op CPU restatement on such inputs.  This kernel is nowhere near a throughput limit.
#pragma clang fp contract(off)

namespace {

constexpr double CMP_EPSILON_D = 0.00001;

struct V3 {
	float x, y, z;
};
__device__ __forceinline__ float v3_len(V3 a) {
	return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
}
__device__ __forceinline__ V3 v3_normalized(V3 a) {
	const float l2 = a.x * a.x + a.y * a.y + a.z * a.z;
	if (l2 == 0) {
		return V3{ 0, 0, 0 };
	}
	const float l = sqrtf(l2);
	return V3{ a.x / l, a.y / l, a.z / l };
}
__device__ __forceinline__ float v3_dot(V3 a, V3 b) {
	return a.x * b.x + a.y * b.y + a.z * b.z;
}
__device__ __forceinline__ V3 basis_xform_inv(const float (&b)[3][3], V3 v) {
	return V3{ b[0][0] * v.x + b[1][0] * v.y + b[2][0] * v.z, b[0][1] * v.x + b[1][1] * v.y + b[2][1] * v.z, b[0][2] * v.x + b[1][2] * v.y + b[2][2] * v.z };
}
__device__ __forceinline__ float db_to_linear(float db) {
	return expf(db * 0.11512925464970228420089957273422f);
}

// audio_spatializer_3d.cpp:123-151
__device__ float attenuation_db(const gas_spatializer3d_config &cfg, const gas_source_pose &s, float dist) {
	float att = 0;
	switch (cfg.attenuation_model) {
		case 0:
			att = (float)(log(1.0 / ((double)(dist / cfg.unit_size) + CMP_EPSILON_D)) * 8.6858896380650365530225783783321);
			break;
		case 1: {
			float d = dist / cfg.unit_size;
			d *= d;
			att = (float)(log(1.0 / ((double)d + CMP_EPSILON_D)) * 8.6858896380650365530225783783321);
		} break;
		case 2:
			att = (float)(-20 * log((double)(dist / cfg.unit_size) + CMP_EPSILON_D));
			break;
		default:
			break;
	}
	att += s.volume_db;
	if (att > s.max_db) {
		att = s.max_db;
	}
	return att;
}

__device__ void calc_output_vol(const gas_spatializer3d_config &cfg, V3 dir, float (&out)[4][2]) {
	if (cfg.speaker_mode == 0) { // :103-110
		const float pan_strength = cfg.global_panning_strength * cfg.panning_strength;
		const double flatrad = sqrt((double)(dir.x * dir.x + dir.z * dir.z));
		double g = (1.0 - pan_strength) * (1.0 - pan_strength);
		g = g < 0.0 ? 0.0 : (g > 1.0 ? 1.0 : g);
		const double f = (1.0 - g) / (1.0 + g);
		double cosx = dir.x / (flatrad == 0.0 ? 1.0 : flatrad);
		cosx = cosx < -1.0 ? -1.0 : (cosx > 1.0 ? 1.0 : cosx);
		const double fcosx = cosx * f;
		out[0][0] = (float)sqrt((-fcosx + 1.0) / 2.0);
		out[0][1] = (float)sqrt((fcosx + 1.0) / 2.0);
		return;
	}
	// SPCAP :56-98, :903-938
	const float sd[7][3] = { { -1, 0, -1 }, { 1, 0, -1 }, { 0, 0, -1 }, { -1, 0, 1 }, { 1, 0, 1 }, { -1, 0, 0 }, { 1, 0, 0 } };
	float tightness = cfg.global_panning_strength * 2.0f;
	tightness *= cfg.panning_strength;
	const int n = cfg.speaker_mode == 1 ? 3 : (cfg.speaker_mode == 2 ? 5 : 7);
	V3 dirs[7];
	float sq[7], volumes[7];
#pragma unroll
	for (int i = 0; i < 7; i++) {
		dirs[i] = v3_normalized(V3{ sd[i][0], sd[i][1], sd[i][2] });
	}
	float sum_sq = 0.0f;
#pragma unroll
	for (int i = 0; i < 7; i++) {
		if (i < n) {
			float eff = 0.0f;
#pragma unroll
			for (int j = 0; j < 7; j++) {
				if (j < n) {
					eff += (float)(0.5 * (1.0 + v3_dot(dirs[i], dirs[j])));
				}
			}
			const float initial_gain = (float)(0.5 * pow(1.0 + (double)v3_dot(dirs[i], dir), (double)tightness) / eff);
			sq[i] = initial_gain * initial_gain;
			sum_sq += sq[i];
		}
	}
#pragma unroll
	for (int i = 0; i < 7; i++) {
		volumes[i] = i < n ? sqrtf(sq[i] / sum_sq) : 0.0f;
	}
	if (cfg.speaker_mode >= 3) {
		out[3][0] = volumes[5];
		out[3][1] = volumes[6];
	}
	if (cfg.speaker_mode >= 2) {
		out[2][0] = volumes[3];
		out[2][1] = volumes[4];
	}
	out[1][0] = volumes[2];
	out[1][1] = 1.0f;
	out[0][0] = volumes[0];
	out[0][1] = volumes[1];
}

// calc_reverb_vol, audio_spatializer_3d.cpp:154-197 (oracle: calc_reverb_vol)
__device__ void calc_reverb_vol(const gas_spatializer3d_config &cfg, const gas_source_pose &s, const gas_area_send &area, V3 lap, const float (&direct)[4][2], float (&rv)[4][2]) {
#pragma unroll
	for (int i = 0; i < 4; i++) {
		rv[i][0] = 0.0f;
		rv[i][1] = 0.0f;
	}
	const float uniformity = area.reverb_uniformity;
	const float area_send = area.reverb_amount;
	if (uniformity > 0.0f) {
		const float distance = v3_len(lap);
		const float attenuation = db_to_linear(attenuation_db(cfg, s, distance)); // :163
		const int chan_count = cfg.speaker_mode + 1;
		const float center = chan_count == 1 ? 0.5f : (chan_count == 2 ? 0.25f : (chan_count == 3 ? 0.16666f : 0.125f)); // :166
		if (attenuation < 1.0f) { // :170-181
			V3 rev_pos = lap;
			rev_pos.y = 0;
			rev_pos = v3_normalized(rev_pos);
			calc_output_vol(cfg, rev_pos, rv);
			for (int i = 0; i < chan_count; i++) {
				rv[i][0] = rv[i][0] + (center - rv[i][0]) * attenuation;
				rv[i][1] = rv[i][1] + (center - rv[i][1]) * attenuation;
			}
		} else {
			for (int i = 0; i < chan_count; i++) {
				rv[i][0] = center;
				rv[i][1] = center;
			}
		}
		for (int i = 0; i < chan_count; i++) { // :187-190
#pragma unroll
			for (int e = 0; e < 2; e++) {
				const float to = rv[i][e] * attenuation;
				const float v = direct[i][e] + (to - direct[i][e]) * uniformity;
				rv[i][e] = v * area_send;
			}
		}
	} else {
#pragma unroll
		for (int i = 0; i < 4; i++) { // :193-195
			rv[i][0] = direct[i][0] * area_send;
			rv[i][1] = direct[i][1] * area_send;
		}
	}
}

__global__ __launch_bounds__(256) void k_calc_spatialization(const gas_spatializer3d_config *__restrict__ cfgs, const uint32_t *__restrict__ cfg_index, const gas_source_pose *__restrict__ poses, const gas_listener *__restrict__ listeners, uint32_t n_listeners, const uint32_t *__restrict__ slots, uint32_t n, gas_params *__restrict__ table, uint8_t *__restrict__ was_further_tab, gas_params *__restrict__ out_params, const gas_area_send *__restrict__ areas, const float *__restrict__ listener_area_pos, gas_audio_frame *__restrict__ out_reverb) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) {
		return;
	}
	const gas_spatializer3d_config cfg = cfgs[cfg_index ? cfg_index[i] : 0];
	const gas_source_pose s = poses[i];
	const uint32_t slot = slots[i];

	const V3 global_pos{ s.position[0], s.position[1], s.position[2] };
	V3 linear_velocity{ 0, 0, 0 };
	if (cfg.doppler_tracking != 0) {
		linear_velocity = V3{ s.velocity[0], s.velocity[1], s.velocity[2] };
	}
	float log_pitch_scale = 0.0f, log_pitch_weight = 0.0f;
	float output_volume[4][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };
	bool in_range = false;
	float linear_attenuation = 0.0f, cutoff = 5000.0f; // a fresh SpatializerParameters3D (audio_spatializer_3d.h:67-68)
	float best_mult = -1.0f;
	V3 best_local{ 0, 0, -1 };
	// the Area3D branches (:349-353, :364-370, :399-402); areas == nullptr: the source sits in no area
	gas_area_send area{};
	if (areas) {
		area = areas[i];
	}
	const bool area_reverb = area.present != 0 && area.using_reverb_bus != 0;
	const bool area_uniform = area_reverb && area.reverb_uniformity > 0 && listener_area_pos != nullptr;
	float reverb_volume[4][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };

	for (uint32_t li = 0; li < n_listeners; li++) {
		const gas_listener L = listeners[li];
		const V3 rel{ global_pos.x - L.origin[0], global_pos.y - L.origin[1], global_pos.z - L.origin[2] };
		const V3 local_pos = basis_xform_inv(L.basis, rel); // :343
		const float dist = v3_len(local_pos);
		float multiplier = db_to_linear(attenuation_db(cfg, s, dist)); // :359
		V3 lap{ 0, 0, 0 };
		if (area_uniform) {
			const float *q = listener_area_pos + ((size_t)i * n_listeners + li) * 3;
			lap = V3{ q[0], q[1], q[2] };
		}
		if (cfg.max_distance > 0) { // :361-374
			float total_max = cfg.max_distance;
			if (area_uniform) {
				const float l = v3_len(lap);
				total_max = total_max > l ? total_max : l;
			}
			if (dist > total_max || total_max > cfg.max_distance) {
				continue;
			}
			const double m = 1.0 - (double)(dist / cfg.max_distance);
			multiplier = (float)((double)multiplier * (m > 0 ? m : 0));
		}
		in_range = true;
		float db_att = (float)((1.0 - (1.0 < (double)multiplier ? 1.0 : (double)multiplier)) * (double)cfg.attenuation_filter_db); // :376
		if (cfg.emission_angle_enabled) { // :378-385
			const float c = v3_dot(v3_normalized(rel), v3_normalized(V3{ s.forward[0], s.forward[1], s.forward[2] }));
			// [ENGINE] Math::acos clamps its argument (recollection, unpinned): a dot of two normalised f32 vectors can land
			// 1 ulp outside [-1, 1], where the bare libm call returns NaN and the cone test below would silently pass
			const float angle = (c < -1.0f ? 3.14159265358979323846f : (c > 1.0f ? 0.0f : acosf(c))) * (float)(180.0 / 3.14159265358979323846);
			if (angle > cfg.emission_angle) {
				db_att -= -cfg.emission_angle_filter_attenuation_db;
			}
		}
		linear_attenuation = db_to_linear(db_att); // :387 (last listener in range wins)
		cutoff = cfg.attenuation_filter_cutoff_hz; // :388

		float tmp[4][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };
		calc_output_vol(cfg, local_pos, tmp); // :391 (local_pos un-normalised, as the reference passes it)
#pragma unroll
		for (int k = 0; k < 4; k++) { // :393-396, _apply_max_volume :257-265
			tmp[k][0] = multiplier * tmp[k][0];
			tmp[k][1] = multiplier * tmp[k][1];
			output_volume[k][0] = output_volume[k][0] > tmp[k][0] ? output_volume[k][0] : tmp[k][0];
			output_volume[k][1] = output_volume[k][1] > tmp[k][1] ? output_volume[k][1] : tmp[k][1];
		}
		if (area_reverb) { // :399-402
			float tr[4][2];
			calc_reverb_vol(cfg, s, area, lap, tmp, tr);
#pragma unroll
			for (int k = 0; k < 4; k++) {
				reverb_volume[k][0] = reverb_volume[k][0] > tr[k][0] ? reverb_volume[k][0] : tr[k][0];
				reverb_volume[k][1] = reverb_volume[k][1] > tr[k][1] ? reverb_volume[k][1] : tr[k][1];
			}
		}
		if (multiplier > best_mult) {
			best_mult = multiplier;
			best_local = local_pos;
		}
		if (cfg.doppler_tracking != 0) { // :405-431
			const V3 dv{ linear_velocity.x - L.velocity[0], linear_velocity.y - L.velocity[1], linear_velocity.z - L.velocity[2] };
			const V3 lv = basis_xform_inv(L.basis, dv);
			if (lv.x != 0 || lv.y != 0 || lv.z != 0) {
				const float approaching = v3_dot(v3_normalized(local_pos), v3_normalized(lv));
				const float velocity = v3_len(lv);
				float dps = s.pitch_scale * cfg.doppler_speed_of_sound / (cfg.doppler_speed_of_sound + velocity * approaching);
				dps = dps < 0.125f ? 0.125f : (dps > 8.0f ? 8.0f : dps);
				float weight = 0.0f;
#pragma unroll
				for (int k = 0; k < 4; k++) {
					weight = tmp[k][0] > weight ? tmp[k][0] : weight;
					weight = tmp[k][1] > weight ? tmp[k][1] : weight;
				}
				log_pitch_scale += weight * log2f(dps);
				log_pitch_weight += weight;
			}
		}
	}

	gas_params *P = table + slot;
#pragma unroll
	for (int k = 0; k < 4; k++) { // :469
		P->mix_volumes[k][0] = output_volume[k][0];
		P->mix_volumes[k][1] = output_volume[k][1];
	}
	P->pitch_scale = log_pitch_weight > 0 ? powf(2.0f, log_pitch_scale / log_pitch_weight) : s.pitch_scale; // :434-438
	P->linear_attenuation = linear_attenuation;
	P->attenuation_filter_cutoff_hz = cutoff;
	const bool was_further = was_further_tab[slot] != 0; // :472-479
	P->update_parameters = (!in_range && was_further) ? 0u : 1u;
	was_further_tab[slot] = in_range ? 0 : 1;
	if (cfg.hrtf_n_az > 0 && cfg.hrtf_n_el > 0) { // NEW: gain + nearest HRIR grid direction towards the loudest listener
		P->hrtf_gain = in_range ? best_mult : 0.0f;
		const double two_pi = 6.2831853071795864769252867666;
		const double az = atan2((double)best_local.x, (double)-best_local.z);
		const double flat = sqrt((double)best_local.x * best_local.x + (double)best_local.z * best_local.z);
		const double el = atan2((double)best_local.y, flat);
		long long ai = llround(az / two_pi * cfg.hrtf_n_az);
		ai = ((ai % (long long)cfg.hrtf_n_az) + (long long)cfg.hrtf_n_az) % (long long)cfg.hrtf_n_az;
		long long ei = cfg.hrtf_n_el > 1 ? llround((el + two_pi / 4) / (two_pi / 2) * (cfg.hrtf_n_el - 1)) : 0;
		ei = ei < 0 ? 0 : (ei > (long long)cfg.hrtf_n_el - 1 ? (long long)cfg.hrtf_n_el - 1 : ei);
		P->hrtf_dir = (uint32_t)(ei * cfg.hrtf_n_az + ai);
	}
	if (out_reverb) { // what the reference sends to the area's reverb bus (:451-452)
#pragma unroll
		for (int k = 0; k < 4; k++) {
			out_reverb[(size_t)i * 4 + k] = gas_audio_frame{ reverb_volume[k][0], reverb_volume[k][1] };
		}
	}
	if (out_params) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		out_params[i] = *P; // this lane wrote every field it reads back or they were already in the table
	}
}

} // namespace

hipError_t gas_launch_calc_spatialization(hipStream_t stream, const gas_spatializer3d_config *cfgs, const uint32_t *cfg_index, const gas_source_pose *poses, const gas_listener *listeners, uint32_t n_listeners, const uint32_t *slots, uint32_t n, gas_params *table, uint8_t *was_further, gas_params *out_params, const gas_area_send *areas, const float *listener_area_pos, gas_audio_frame *out_reverb) {
	if (n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_calc_spatialization, dim3((n + 255) / 256), dim3(256), 0, stream, cfgs, cfg_index, poses, listeners, n_listeners, slots, n, table, was_further, out_params, areas, listener_area_pos, out_reverb);
	return hipGetLastError();
}
