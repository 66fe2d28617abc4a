/*
 * gas_oracle.c -- CPU restatement of the godot-audio-spatializer hot path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see gas_oracle.h header).
 *
 * Plain C99, IEEE f32 op-for-op where the reference computes in f32; built with
 * -O2 -ffp-contract=off (no fast-math, no FMA contraction) by oracle/Makefile.
 * Citations are file:line in /root/reference/.  [ENGINE] marks arithmetic that
 * lives in the un-vendored Godot engine and is restated from SURVEY.md Appendix B.
 */
#include "gas_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* [ENGINE] Math::db_to_linear / linear_to_db (Appendix B)             */
/* ------------------------------------------------------------------ */
float gaso_db_to_linear(float db) {
	return expf(db * 0.11512925464970228420089957273422f);
}

float gaso_linear_to_db(float lin) {
	return logf(lin) * 8.6858896380650365530225783783321f;
}

/* ------------------------------------------------------------------ */
/* [ENGINE] AudioFilterSW::prepare_coefficients, HIGHSHELF branch       */
/* (call sites audio_spatializer_3d.cpp:504-510, 569-575).              */
/* f64 arithmetic, members stored f32, then normalised by a0 with the   */
/* feedback terms negated (Appendix B).                                 */
/* ------------------------------------------------------------------ */
void gaso_highshelf_coeffs(double sampling_rate, double cutoff, double resonance, double gain, int stages, gaso_coeffs *out) {
	int sr_limit = (int)(sampling_rate / 2) + 512;
	double final_cutoff = (cutoff > sr_limit) ? sr_limit : cutoff;
	if (final_cutoff < 1) {
		final_cutoff = 1;
	}
	double omega = 6.2831853071795864769252867666 * final_cutoff / sampling_rate;
	double sin_v = sin(omega);
	double cos_v = cos(omega);

	double Q = resonance;
	if (Q <= 0.0) {
		Q = 0.0001;
	}
	double tmpgain = gain;
	if (tmpgain < 0.001) {
		tmpgain = 0.001;
	}
	if (stages > 1) {
		Q = (Q > 1.0 ? pow(Q, 1.0 / stages) : Q);
		tmpgain = pow(tmpgain, 1.0 / (stages + 1));
	}

	double tmpq = sqrt(Q);
	if (tmpq <= 0) {
		tmpq = 0.001;
	}
	double beta = sqrt(tmpgain) / tmpq;

	double a0 = (tmpgain + 1.0) - (tmpgain - 1.0) * cos_v + beta * sin_v;
	out->b0 = (float)(tmpgain * ((tmpgain + 1.0) + (tmpgain - 1.0) * cos_v + beta * sin_v));
	out->b1 = (float)(-2.0 * tmpgain * ((tmpgain - 1.0) + (tmpgain + 1.0) * cos_v));
	out->b2 = (float)(tmpgain * ((tmpgain + 1.0) + (tmpgain - 1.0) * cos_v - beta * sin_v));
	out->a1 = (float)(2.0 * ((tmpgain - 1.0) - (tmpgain + 1.0) * cos_v));
	out->a2 = (float)((tmpgain + 1.0) - (tmpgain - 1.0) * cos_v - beta * sin_v);

	out->b0 = (float)(out->b0 / a0);
	out->b1 = (float)(out->b1 / a0);
	out->b2 = (float)(out->b2 / a0);
	out->a1 = (float)(out->a1 / (0.0 - a0));
	out->a2 = (float)(out->a2 / (0.0 - a0));
}

/* [ENGINE] AudioFilterSW::Processor::update_coeffs(len) (Appendix B). */
void gaso_processor_update_coeffs(gaso_processor *p, const gaso_coeffs *target, int interp_len) {
	if (interp_len) {
		gaso_coeffs old = p->coeffs;
		p->incr.a1 = (target->a1 - old.a1) / interp_len;
		p->incr.a2 = (target->a2 - old.a2) / interp_len;
		p->incr.b0 = (target->b0 - old.b0) / interp_len;
		p->incr.b1 = (target->b1 - old.b1) / interp_len;
		p->incr.b2 = (target->b2 - old.b2) / interp_len;
		p->coeffs = old;
	} else {
		p->coeffs = *target;
	}
}

/* [ENGINE] Processor::process_one (Appendix B). */
float gaso_processor_process_one(gaso_processor *p, float x) {
	float pre = x;
	float y = (x * p->coeffs.b0 + p->hb1 * p->coeffs.b1 + p->hb2 * p->coeffs.b2 + p->ha1 * p->coeffs.a1 + p->ha2 * p->coeffs.a2);
	p->ha2 = p->ha1;
	p->hb2 = p->hb1;
	p->hb1 = pre;
	p->ha1 = y;
	return y;
}

/* [ENGINE] Processor::process_one_interp (Appendix B): same, then coeffs += incr. */
float gaso_processor_process_one_interp(gaso_processor *p, float x) {
	float y = gaso_processor_process_one(p, x);
	p->coeffs.b0 += p->incr.b0;
	p->coeffs.b1 += p->incr.b1;
	p->coeffs.b2 += p->incr.b2;
	p->coeffs.a1 += p->incr.a1;
	p->coeffs.a2 += p->incr.a2;
	return y;
}

static void processor_set_filter(gaso_processor *p, int clear_history) {
	if (clear_history) {
		p->ha1 = p->ha2 = p->hb1 = p->hb2 = 0;
	}
}

/* audio_spatializer_3d.cpp:873-885 */
static void pd3d_get_prev(const gaso_pdata3d *pd, int channel, float out[2]) {
	if (pd->prev_count <= channel) {
		out[0] = 0.0f;
		out[1] = 0.0f;
		return;
	}
	out[0] = pd->prev_mix_volumes[channel][0];
	out[1] = pd->prev_mix_volumes[channel][1];
}

static void pd3d_set_prev(gaso_pdata3d *pd, int channel, const float v[2]) {
	if (pd->prev_count <= channel) {
		/* Vector::resize value-initialises the new Vector2 entries to (0,0) [ENGINE]. */
		for (int c = pd->prev_count; c <= channel; c++) {
			pd->prev_mix_volumes[c][0] = 0.0f;
			pd->prev_mix_volumes[c][1] = 0.0f;
		}
		pd->prev_count = channel + 1;
	}
	pd->prev_mix_volumes[channel][0] = v[0];
	pd->prev_mix_volumes[channel][1] = v[1];
}

/* ------------------------------------------------------------------ */
/* AudioSpatializerInstance3D::process_frames                           */
/* audio_spatializer_3d.cpp:491-552                                     */
/* ------------------------------------------------------------------ */
void gaso_process_frames_3d(const gaso_params *params, gaso_pdata3d *pd, gaso_frame *out, const gaso_frame *src, int n, float mix_rate) {
	float prev_vol[2];
	pd3d_get_prev(pd, 0, prev_vol); /* :500 */

	float highshelf_gain = params->linear_attenuation; /* :502 */
	if (highshelf_gain >= 0.001) { /* :503 (float promoted to double against 0.001) */
		gaso_coeffs target;
		gaso_highshelf_coeffs(mix_rate, params->attenuation_filter_cutoff_hz, 1, highshelf_gain, 1, &target); /* :504-510 */
		gaso_processor *pl = &pd->filter_processors[0]; /* :512 */
		gaso_processor *pr = &pd->filter_processors[1]; /* :513 */
		int is_just_started = prev_vol[0] == 0 && prev_vol[1] == 0; /* :518 */
		processor_set_filter(pl, is_just_started);
		gaso_processor_update_coeffs(pl, &target, n);
		processor_set_filter(pr, is_just_started);
		gaso_processor_update_coeffs(pr, &target, n);
		for (int i = 0; i < n; i++) { /* :524-529 */
			gaso_frame mixed = src[i];
			mixed.l = gaso_processor_process_one_interp(pl, mixed.l);
			mixed.r = gaso_processor_process_one_interp(pr, mixed.r);
			out[i] = mixed;
		}
	} else {
		for (int i = 0; i < n; i++) { /* :531-534 */
			out[i] = src[i];
		}
	}

	float max_volume = 0.0f; /* :537-548 */
	int max_index = 0;
	for (int i = 0; i < GASO_MAX_CHANNELS_PER_BUS; i++) {
		if (params->mix_volumes[i][0] > max_volume) {
			max_volume = params->mix_volumes[i][0];
			max_index = i;
		}
		if (params->mix_volumes[i][1] > max_volume) {
			max_volume = params->mix_volumes[i][1];
			max_index = i;
		}
	}
	pd3d_set_prev(pd, 0, params->mix_volumes[max_index]); /* :551 */
}

/* ------------------------------------------------------------------ */
/* AudioSpatializerInstance3D::mix_channel                              */
/* audio_spatializer_3d.cpp:554-609                                     */
/* ------------------------------------------------------------------ */
void gaso_mix_channel_3d(const gaso_params *params, gaso_pdata3d *pd, int channel, gaso_frame *out, const gaso_frame *src, int n, float mix_rate) {
	float vs[2];
	pd3d_get_prev(pd, channel, vs); /* :564 */
	const float vf[2] = { params->mix_volumes[channel][0], params->mix_volumes[channel][1] }; /* :565 */

	float highshelf_gain = params->linear_attenuation;
	if (highshelf_gain >= 0.001) { /* :568 */
		gaso_coeffs target;
		gaso_highshelf_coeffs(mix_rate, params->attenuation_filter_cutoff_hz, 1, highshelf_gain, 1, &target);
		gaso_processor *pl = &pd->filter_processors[channel * 2]; /* :887-894 */
		gaso_processor *pr = &pd->filter_processors[channel * 2 + 1];
		int is_just_started = vs[0] == 0 && vs[1] == 0; /* :583 */
		processor_set_filter(pl, is_just_started);
		gaso_processor_update_coeffs(pl, &target, n);
		processor_set_filter(pr, is_just_started);
		gaso_processor_update_coeffs(pr, &target, n);
		for (int i = 0; i < n; i++) { /* :589-597 */
			float t = (float)i / n;
			/* AudioFrame vol = p_vol_final * t + (1 - t) * p_vol_start; */
			float one_minus = 1 - t;
			float vol_l = vf[0] * t + one_minus * vs[0];
			float vol_r = vf[1] * t + one_minus * vs[1];
			float ml = vol_l * src[i].l;
			float mr = vol_r * src[i].r;
			ml = gaso_processor_process_one_interp(pl, ml);
			mr = gaso_processor_process_one_interp(pr, mr);
			out[i].l = ml;
			out[i].r = mr;
		}
	} else {
		for (int i = 0; i < n; i++) { /* :600-604 */
			float t = (float)i / n;
			float one_minus = 1 - t;
			out[i].l = (vf[0] * t + one_minus * vs[0]) * src[i].l;
			out[i].r = (vf[1] * t + one_minus * vs[1]) * src[i].r;
		}
	}
	pd3d_set_prev(pd, channel, vf); /* :608 */
}

/* ------------------------------------------------------------------ */
/* Effects                                                              */
/* ------------------------------------------------------------------ */

/* [ENGINE] AudioFilterSW::prepare_coefficients, the modes AudioEffectFilter's other subclasses select (recollection of
 * servers/audio/audio_filter_sw.cpp -- unpinned): LOWPASS, HIGHPASS, BANDPASS (Q doubled), NOTCH,
 * LOWSHELF, stages = 1.  RBJ cookbook forms; tests/test_oracle_primitives.py pins them against scipy. */
void gaso_filter_coeffs(int kind, double sampling_rate, double cutoff, double resonance, double gain, gaso_coeffs *out) {
	int sr_limit = (int)(sampling_rate / 2) + 512;
	double final_cutoff = (cutoff > sr_limit) ? sr_limit : cutoff;
	if (final_cutoff < 1) {
		final_cutoff = 1;
	}
	double omega = 6.2831853071795864769252867666 * final_cutoff / sampling_rate;
	double sin_v = sin(omega);
	double cos_v = cos(omega);
	double Q = resonance;
	if (Q <= 0.0) {
		Q = 0.0001;
	}
	if (kind == GASO_FX_BANDPASS) {
		Q *= 2.0;
	}
	double tmpgain = gain;
	if (tmpgain < 0.001) {
		tmpgain = 0.001;
	}
	double alpha = sin_v / (2 * Q);
	double a0 = 1.0 + alpha;
	switch (kind) {
		case GASO_FX_LOWPASS:
			out->b0 = (float)((1.0 - cos_v) / 2.0);
			out->b1 = (float)(1.0 - cos_v);
			out->b2 = (float)((1.0 - cos_v) / 2.0);
			out->a1 = (float)(-2.0 * cos_v);
			out->a2 = (float)(1.0 - alpha);
			break;
		case GASO_FX_HIGHPASS:
			out->b0 = (float)((1.0 + cos_v) / 2.0);
			out->b1 = (float)(-(1.0 + cos_v));
			out->b2 = (float)((1.0 + cos_v) / 2.0);
			out->a1 = (float)(-2.0 * cos_v);
			out->a2 = (float)(1.0 - alpha);
			break;
		case GASO_FX_BANDPASS:
			out->b0 = (float)(alpha * sqrt(Q + 1));
			out->b1 = 0.0f;
			out->b2 = (float)(-alpha * sqrt(Q + 1));
			out->a1 = (float)(-2.0 * cos_v);
			out->a2 = (float)(1.0 - alpha);
			break;
		case GASO_FX_NOTCH:
			out->b0 = 1.0f;
			out->b1 = (float)(-2.0 * cos_v);
			out->b2 = 1.0f;
			out->a1 = (float)(-2.0 * cos_v);
			out->a2 = (float)(1.0 - alpha);
			break;
		default: { /* GASO_FX_LOWSHELF */
			double tmpq = sqrt(Q);
			if (tmpq <= 0) {
				tmpq = 0.001;
			}
			double beta = sqrt(tmpgain) / tmpq;
			a0 = (tmpgain + 1.0) + (tmpgain - 1.0) * cos_v + beta * sin_v;
			out->b0 = (float)(tmpgain * ((tmpgain + 1.0) - (tmpgain - 1.0) * cos_v + beta * sin_v));
			out->b1 = (float)(2.0 * tmpgain * ((tmpgain - 1.0) - (tmpgain + 1.0) * cos_v));
			out->b2 = (float)(tmpgain * ((tmpgain + 1.0) - (tmpgain - 1.0) * cos_v - beta * sin_v));
			out->a1 = (float)(-2.0 * ((tmpgain - 1.0) + (tmpgain + 1.0) * cos_v));
			out->a2 = (float)((tmpgain + 1.0) + (tmpgain - 1.0) * cos_v - beta * sin_v);
		} break;
	}
	out->b0 = (float)(out->b0 / a0);
	out->b1 = (float)(out->b1 / a0);
	out->b2 = (float)(out->b2 / a0);
	out->a1 = (float)(out->a1 / (0.0 - a0));
	out->a2 = (float)(out->a2 / (0.0 - a0));
}

/* [ENGINE] AudioEffectFilterInstance::process for the other filter resources at FILTER_6DB: settings read from the
 * resource every block, coefficients snapped, process_one per ear (same shape as fx_highshelf below). */
static void fx_filter(int kind, gaso_fx_state *st, const gaso_frame *src, gaso_frame *dst, int n, float mix_rate) {
	gaso_coeffs target;
	gaso_filter_coeffs(kind, mix_rate, st->cutoff_hz, st->resonance, st->gain, &target);
	gaso_processor_update_coeffs(&st->shelf[0], &target, 0);
	gaso_processor_update_coeffs(&st->shelf[1], &target, 0);
	for (int i = 0; i < n; i++) {
		dst[i].l = gaso_processor_process_one(&st->shelf[0], src[i].l);
	}
	for (int i = 0; i < n; i++) {
		dst[i].r = gaso_processor_process_one(&st->shelf[1], src[i].r);
	}
}

/* [ENGINE] AudioEffectAmplifyInstance::process (recollection of servers/audio/effects/audio_effect_amplify.cpp):
 *   vol = db_to_linear(mix_volume_db); vol_inc = (db_to_linear(volume_db) - vol) / frame_count;
 *   dst[i] = src[i] * vol; vol += vol_inc;   then mix_volume_db = volume_db. */
static void fx_amplify(gaso_fx_state *st, const gaso_frame *src, gaso_frame *dst, int n) {
	if (!st->amp_started) { /* instantiate(): mix_volume_db = volume_db */
		st->amp_mix_volume_db = st->volume_db;
		st->amp_started = 1;
	}
	float vol = gaso_db_to_linear(st->amp_mix_volume_db);
	float vol_inc = (gaso_db_to_linear(st->volume_db) - vol) / (float)n;
	for (int i = 0; i < n; i++) {
		dst[i].l = src[i].l * vol;
		dst[i].r = src[i].r * vol;
		vol += vol_inc;
	}
	st->amp_mix_volume_db = st->volume_db;
}

/* [ENGINE] AudioEffectFilterInstance::process for AudioEffectHighShelfFilter at
 * FILTER_6DB: coefficients snapped every call (update_coeffs() with no
 * interpolation), process_one over all left samples then all right samples
 * (Appendix B; example gd_spatializer.gd:14-19). */
static void fx_highshelf(const gaso_params *params, gaso_fx_state *st, const gaso_frame *src, gaso_frame *dst, int n, float mix_rate) {
	gaso_coeffs target;
	gaso_highshelf_coeffs(mix_rate, params->fx_shelf_cutoff_hz, 1.0, params->fx_shelf_gain, 1, &target);
	gaso_processor_update_coeffs(&st->shelf[0], &target, 0);
	gaso_processor_update_coeffs(&st->shelf[1], &target, 0);
	for (int i = 0; i < n; i++) {
		dst[i].l = gaso_processor_process_one(&st->shelf[0], src[i].l);
	}
	for (int i = 0; i < n; i++) {
		dst[i].r = gaso_processor_process_one(&st->shelf[1], src[i].r);
	}
}

/* NEW (no reference code): 8-tap early reflections on the stereo frames.
 * y[i] = x[i] + sum_k er_gain[k] * x[i - er_delay[k]], history kept in a
 * per-playback ring of ring_frames frames (zeros before the stream started).
 * Taps accumulate in tap order in f32. 1 <= er_delay[k] <= ring_frames - n. */
static void fx_early_reflections(const gaso_params *params, gaso_fx_state *st, const gaso_frame *src, gaso_frame *dst, int n) {
	const uint32_t R = st->ring_frames;
	uint32_t pos = st->ring_pos;
	/* dst may alias nothing of the ring; src may equal dst is NOT allowed by the chain. */
	for (int i = 0; i < n; i++) {
		st->ring[(pos + (uint32_t)i) % R] = src[i];
	}
	for (int i = 0; i < n; i++) {
		float yl = src[i].l;
		float yr = src[i].r;
		for (int k = 0; k < GASO_ER_TAPS; k++) {
			uint32_t d = params->er_delay[k];
			gaso_frame x = st->ring[(pos + (uint32_t)i + R - d) % R];
			yl = yl + params->er_gain[k] * x.l;
			yr = yr + params->er_gain[k] * x.r;
		}
		dst[i].l = yl;
		dst[i].r = yr;
	}
	st->ring_pos = (pos + (uint32_t)n) % R;
}

/* NEW (no reference code): per-source HRTF.  mono = (l + r) * 0.5; gain ramps
 * like the reference's volume lerp (audio_spatializer_3d.cpp:591-592, t = i/n);
 * x = mono * gain; stereo out = direct-form 256-tap FIR with the HRIR pair of
 * direction hrtf_dir over hist ++ x, accumulated in f64 (the float64 oracle
 * SURVEY.md section 8c asks for), rounded to f32.  The direction switches at the
 * block boundary without cross-fade. */
static void hrtf_make_x(const gaso_params *params, gaso_fx_state *st, const gaso_frame *src, float *xx /* [T-1+n] */, int n) {
	const int H = GASO_HRTF_TAPS - 1;
	memcpy(xx, st->hist, sizeof(float) * H);
	float g0 = st->prev_gain;
	float g1 = params->hrtf_gain;
	for (int i = 0; i < n; i++) {
		float t = (float)i / n;
		float g = g1 * t + (1 - t) * g0;
		float mono = (src[i].l + src[i].r) * 0.5f;
		xx[H + i] = mono * g;
	}
}

static void hrtf_commit(const gaso_params *params, gaso_fx_state *st, const float *xx, int n) {
	const int H = GASO_HRTF_TAPS - 1;
	memmove(st->hist, xx + n, sizeof(float) * H); /* last T-1 samples of hist ++ x */
	st->prev_gain = params->hrtf_gain;
}

static void fx_hrtf(const gaso_params *params, gaso_fx_state *st, const gaso_hrtf *hrtf, const gaso_frame *src, gaso_frame *dst, int n) {
	const int T = GASO_HRTF_TAPS;
	const int H = T - 1;
	float *xx = (float *)malloc(sizeof(float) * (size_t)(H + n));
	hrtf_make_x(params, st, src, xx, n);
	uint32_t dir = params->hrtf_dir < hrtf->dirs ? params->hrtf_dir : 0;
	uint32_t old_dir = dir;
	if (hrtf->crossfade && st->prev_dir_plus1 > 0) {
		old_dir = (uint32_t)(st->prev_dir_plus1 - 1);
	}
	const float *hl = hrtf->hrir + (size_t)dir * 2 * T;
	const float *hr = hl + T;
	const float *ol = hrtf->hrir + (size_t)old_dir * 2 * T;
	const float *orr = ol + T;
	for (int i = 0; i < n; i++) {
		double al = 0.0, ar = 0.0, bl = 0.0, br = 0.0;
		const float *xp = xx + H + i;
		for (int k = 0; k < T; k++) {
			double x = xp[-k];
			al += (double)hl[k] * x;
			ar += (double)hr[k] * x;
		}
		if (old_dir != dir) { /* cross-fade: old HRIR fades out, new fades in, t = i / n */
			for (int k = 0; k < T; k++) {
				double x = xp[-k];
				bl += (double)ol[k] * x;
				br += (double)orr[k] * x;
			}
			double t = (double)((float)i / n);
			al = al * t + bl * (1.0 - t);
			ar = ar * t + br * (1.0 - t);
		}
		dst[i].l = (float)al;
		dst[i].r = (float)ar;
	}
	hrtf_commit(params, st, xx, n);
	st->prev_dir_plus1 = (int32_t)dir + 1;
	free(xx);
}

void gaso_fx_process(int kind, const gaso_params *params, gaso_fx_state *st, const gaso_hrtf *hrtf, const gaso_frame *src, gaso_frame *dst, int n, float mix_rate) {
	switch (kind) {
		case GASO_FX_HIGHSHELF:
			fx_highshelf(params, st, src, dst, n, mix_rate);
			break;
		case GASO_FX_EARLY_REFLECTIONS:
			fx_early_reflections(params, st, src, dst, n);
			break;
		case GASO_FX_LOWPASS:
		case GASO_FX_HIGHPASS:
		case GASO_FX_BANDPASS:
		case GASO_FX_NOTCH:
		case GASO_FX_LOWSHELF:
			fx_filter(kind, st, src, dst, n, mix_rate);
			break;
		case GASO_FX_AMPLIFY:
			fx_amplify(st, src, dst, n);
			break;
		case GASO_FX_HRTF:
			if (hrtf->impl == 1) {
				gaso_hrtf_ols_radix2(params, st, hrtf, src, dst, n);
			} else {
				fx_hrtf(params, st, hrtf, src, dst, n);
			}
			break;
		default:
			for (int i = 0; i < n; i++) {
				dst[i] = src[i];
			}
	}
}

/* ------------------------------------------------------------------ */
/* AudioSpatializerInstanceEffect::process_frames                       */
/* audio_spatializer_effect.cpp:33-77                                   */
/* (_process_effects, :39/:90-92, is the caller's hook: its effect is   */
/* already folded into *params.)                                        */
/* ------------------------------------------------------------------ */
uint32_t gaso_process_frames_effect(const gaso_params *params, gaso_pdata_effect *pd, const gaso_hrtf *hrtf, gaso_frame *out, const gaso_frame *src, int n, gaso_frame *temp, float mix_rate) {
	uint32_t trace = 0;
	int E = pd->n_effects;
	if (E == 0) { /* :41-46 */
		for (int i = 0; i < n; i++) {
			out[i] = src[i];
		}
		return 0;
	}
	for (int j = 0; j < E; j++) { /* :52-76 */
		int is_even = ((j + E) % 2 == 0);
		const gaso_frame *s;
		gaso_frame *d;
		if (j == 0) {
			s = src;
		} else if (is_even) {
			s = out;
		} else {
			s = temp;
			trace |= 1u << (2 * j + 1);
		}
		if (is_even) {
			d = temp;
			trace |= 1u << (2 * j);
		} else {
			d = out;
		}
		gaso_fx_process(pd->kinds[j], params, &pd->fx[j], hrtf, s, d, n, mix_rate);
	}
	return trace;
}

/* ------------------------------------------------------------------ */
/* AudioSpatializerInstance mixer, audio_spatializer.cpp:326-527        */
/* ------------------------------------------------------------------ */

/* Synthetic stand-in for [ENGINE] AudioStreamPlayback::mix: copies what is left
 * of the stream, zero-fills the remainder, returns the number of real frames. */
static int stream_mix(gaso_playback *pb, gaso_frame *dst, int n) {
	int64_t left = pb->stream_frames - pb->stream_pos;
	int m = (int)(left < n ? (left < 0 ? 0 : left) : n);
	for (int i = 0; i < m; i++) {
		dst[i] = pb->stream[pb->stream_pos + i];
	}
	for (int i = m; i < n; i++) {
		dst[i].l = 0.0f;
		dst[i].r = 0.0f;
	}
	pb->stream_pos += m;
	return m;
}

/* [ENGINE] AudioStreamPlaybackResampled::mix (recollection of godotengine/godot servers/audio/audio_stream.cpp,
 * unpinned): mix_increment = uint64((stream_rate * rate_scale / mix_rate) * 65536) with the stream at the mix rate;
 * per output frame  mu = frac / 65536,  y0..y3 = frames q-3 .. q  (zero outside the stream),
 *   mu2 = mu*mu; h11 = mu2*(mu-1); z = mu2-h11; h01 = z-h11; h10 = mu-z;
 *   out = y1 + (y2-y1)*h01 + ((y2-y0)*h10 + (y3-y1)*h11)*0.5          (AudioFrame operators, f32)
 * and the call reports as mixed the frames produced before q first reaches the stream's end. */
static gaso_frame stream_at(const gaso_playback *pb, int64_t j) {
	gaso_frame z = { 0.0f, 0.0f };
	return (j >= 0 && j < pb->stream_frames) ? pb->stream[j] : z;
}

static int stream_mix_resampled(gaso_playback *pb, gaso_frame *dst, int n, float rate_scale, float mix_rate) {
	const uint64_t inc = (uint64_t)(((double)(mix_rate * rate_scale) / (double)mix_rate) * 65536.0);
	int mixed = -1;
	for (int i = 0; i < n; i++) {
		const int64_t q = (int64_t)(pb->mix_offset >> 16);
		const float mu = (float)(pb->mix_offset & 0xFFFFu) / 65536.0f;
		const gaso_frame y0 = stream_at(pb, q - 3), y1 = stream_at(pb, q - 2), y2 = stream_at(pb, q - 1), y3 = stream_at(pb, q);
		if (q >= pb->stream_frames && mixed == -1) {
			mixed = i;
		}
		const float mu2 = mu * mu;
		const float h11 = mu2 * (mu - 1);
		const float z = mu2 - h11;
		const float h01 = z - h11;
		const float h10 = mu - z;
		dst[i].l = y1.l + (y2.l - y1.l) * h01 + ((y2.l - y0.l) * h10 + (y3.l - y1.l) * h11) * 0.5f;
		dst[i].r = y1.r + (y2.r - y1.r) * h01 + ((y2.r - y0.r) * h10 + (y3.r - y1.r) * h11) * 0.5f;
		pb->mix_offset += inc;
	}
	return mixed == -1 ? n : mixed;
}

/* audio_spatializer.cpp:367-408 */
void gaso_fetch_source(gaso_playback *pb, gaso_frame *buf, int n, float pitch_scale, float mix_rate) {
	if (pb->has_frames) {
		for (int i = 0; i < GASO_LOOKAHEAD; i++) { /* :371-373 */
			buf[i] = pb->lookahead[i];
		}
		int mixed_frames = pb->resampled ? stream_mix_resampled(pb, &buf[GASO_LOOKAHEAD], n, pitch_scale, mix_rate) : stream_mix(pb, &buf[GASO_LOOKAHEAD], n); /* :375-378 */
		if (mixed_frames != n) { /* :380-398 */
			float fadeout_base = 0.96;
			float fadeout_coefficient = 1;
			float buffer_size_float = (float)GASO_LOOKAHEAD;
			float buffer_linear_fade_idx = 0.0;
			int fade_limit = mixed_frames + GASO_LOOKAHEAD;
			for (int idx = mixed_frames; idx < n; idx++) {
				if (idx < fade_limit) {
					fadeout_coefficient *= fadeout_base;
					float f = fadeout_coefficient * (buffer_size_float - buffer_linear_fade_idx) / buffer_size_float;
					buf[idx].l *= f;
					buf[idx].r *= f;
					buffer_linear_fade_idx += 1.0;
				} else {
					buf[idx].l *= 0.0f; /* NaN-preserving zeroing, :394 */
					buf[idx].r *= 0.0f;
				}
			}
			pb->has_frames = 0;
		} else {
			for (int i = 0; i < GASO_LOOKAHEAD; i++) { /* :401-403 */
				pb->lookahead[i] = buf[n + i];
			}
		}
	} else {
		for (int i = 0; i < n + GASO_LOOKAHEAD; i++) { /* :407 */
			buf[i].l = 0;
			buf[i].r = 0;
		}
	}
}

static int kind_should_process_frames(int kind) {
	return kind != GASO_KIND_3D_MIX; /* audio_spatializer_3d.h:145; audio_spatializer_effect.h:57 */
}

static int kind_should_mix_channels(int kind) {
	return kind == GASO_KIND_3D_MIX; /* audio_spatializer_3d.h:146; audio_spatializer_effect.h:58 */
}

/* audio_spatializer.cpp:410-462 minus the += into mix_buffer. */
void gaso_playback_contribution(gaso_instance *inst, const gaso_params *params, gaso_pdata3d *pd3d, gaso_pdata_effect *pdfx, const gaso_frame *buf, int n, gaso_frame *const contrib[GASO_MAX_CHANNELS_PER_BUS], float peak[2]) {
	const gaso_frame *processed;
	if (kind_should_process_frames(inst->kind)) { /* :411-414 */
		if (inst->kind == GASO_KIND_3D_PROCESS) {
			gaso_process_frames_3d(params, pd3d, inst->process_buffer, buf, n, inst->mix_rate);
		} else {
			gaso_process_frames_effect(params, pdfx, inst->hrtf, inst->process_buffer, buf, n, inst->fx_temp, inst->mix_rate);
		}
		processed = inst->process_buffer;
	} else {
		processed = buf;
	}
	peak[0] = 0;
	peak[1] = 0;
	if (kind_should_mix_channels(inst->kind)) { /* :421-445 */
		for (int c = 0; c < inst->channel_count; c++) {
			gaso_frame *o = inst->temp_buffer;
			gaso_mix_channel_3d(params, pd3d, c, o, processed, n, inst->mix_rate);
			for (int i = 0; i < n; i++) {
				contrib[c][i] = o[i];
				float l = fabsf(o[i].l);
				if (l > peak[0]) {
					peak[0] = l;
				}
				float r = fabsf(o[i].r);
				if (r > peak[1]) {
					peak[1] = r;
				}
			}
		}
	} else { /* :446-462 */
		for (int i = 0; i < n; i++) {
			contrib[0][i] = processed[i];
			float l = fabsf(processed[i].l);
			if (l > peak[0]) {
				peak[0] = l;
			}
			float r = fabsf(processed[i].r);
			if (r > peak[1]) {
				peak[1] = r;
			}
		}
	}
}

void gaso_mix_from_playback_list(gaso_instance *inst, const gaso_params *const *params, gaso_playback *const *playbacks, int n_playbacks, int n) {
	for (int c = 0; c < inst->channel_count; c++) { /* :335-343 */
		for (int i = 0; i < n; i++) {
			inst->mix_buffer[c][i].l = 0.f;
			inst->mix_buffer[c][i].r = 0.f;
		}
	}
	inst->mix_buffer_size = n;
	gaso_frame *contrib[GASO_MAX_CHANNELS_PER_BUS];
	for (int c = 0; c < GASO_MAX_CHANNELS_PER_BUS; c++) {
		contrib[c] = (gaso_frame *)malloc(sizeof(gaso_frame) * (size_t)n);
	}
	for (int p = 0; p < n_playbacks; p++) { /* :353 */
		gaso_playback *pb = playbacks[p];
		if (!pb->active || pb->paused) { /* :355-357; paused: see gaso_playback */
			continue;
		}
		gaso_frame *buf = inst->playback_buffer;
		gaso_fetch_source(pb, buf, n, params[p]->pitch_scale, inst->mix_rate); /* :375 pitch_scale read per playback */
		float peak[2];
		gaso_playback_contribution(inst, params[p], &pb->pd3d, &pb->pdfx, buf, n, contrib, peak);
		int channels = kind_should_mix_channels(inst->kind) ? inst->channel_count : 1;
		for (int c = 0; c < channels; c++) { /* :433-434, :450-451 */
			for (int i = 0; i < n; i++) {
				inst->mix_buffer[c][i].l += contrib[c][i].l;
				inst->mix_buffer[c][i].r += contrib[c][i].r;
			}
		}
		pb->last_peak[0] = peak[0];
		pb->last_peak[1] = peak[1];
		if (!pb->has_frames) { /* :464-469 */
			float m = peak[1] > peak[0] ? peak[1] : peak[0];
			if (m <= gaso_db_to_linear(inst->disable_threshold_db)) {
				pb->active = 0;
			}
		}
	}
	for (int c = 0; c < GASO_MAX_CHANNELS_PER_BUS; c++) {
		free(contrib[c]);
	}
}

/* audio_spatializer.cpp:494-508 */
int gaso_check_channel_mixed(gaso_instance *inst, int channel) {
	if (inst->channel_mixed[channel]) {
		for (int c = 0; c < GASO_MAX_CHANNELS_PER_BUS; c++) {
			inst->channel_mixed[c] = 0;
		}
		inst->channel_mixed[channel] = 1;
		return 1;
	}
	inst->channel_mixed[channel] = 1;
	return 0;
}

/* audio_spatializer.cpp:510-527; returns 0 ok, -1 on the ERR_FAIL paths (:521-522). */
int gaso_get_mixed_frames(gaso_instance *inst, const gaso_params *const *params, gaso_playback *const *playbacks, int n_playbacks, int channel, gaso_frame *frames, int n) {
	if (gaso_check_channel_mixed(inst, channel)) {
		gaso_mix_from_playback_list(inst, params, playbacks, n_playbacks, n);
	}
	if (channel < 0 || channel >= inst->channel_count) {
		return -1;
	}
	if (n != inst->mix_buffer_size) {
		return -1;
	}
	for (int i = 0; i < n; i++) {
		frames[i] = inst->mix_buffer[channel][i];
	}
	return 0;
}

/* audio_spatializer.cpp:295-319, one bus. */
void gaso_bus_map(int should_mix_channels, int channel, const float bus_volume[4][2], const float mix_volumes[4][2], float out[4][2]) {
	for (int c = 0; c < GASO_MAX_CHANNELS_PER_BUS; c++) {
		if (should_mix_channels) {
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
		} else {
			out[c][0] = mix_volumes[c][0];
			out[c][1] = mix_volumes[c][1];
		}
	}
}

/* ------------------------------------------------------------------ */
/* Batched convenience: what one callback computes for n_src sources    */
/* that are all active with a full window already fetched.              */
/* ------------------------------------------------------------------ */
void gaso_batch_block(int kind, int channel_count, const gaso_params *params, gaso_batch_state *states, const gaso_hrtf *hrtf, const gaso_frame *src, int n_src, int n, float mix_rate, gaso_frame *mix, double *mix64, float *peaks) {
	gaso_instance inst;
	memset(&inst, 0, sizeof(inst));
	inst.kind = kind;
	inst.channel_count = kind_should_mix_channels(kind) ? channel_count : 1;
	inst.mix_rate = mix_rate;
	inst.hrtf = hrtf;
	inst.process_buffer = (gaso_frame *)malloc(sizeof(gaso_frame) * (size_t)n);
	inst.temp_buffer = (gaso_frame *)malloc(sizeof(gaso_frame) * (size_t)n);
	inst.fx_temp = (gaso_frame *)malloc(sizeof(gaso_frame) * (size_t)n);
	gaso_frame *contrib[GASO_MAX_CHANNELS_PER_BUS];
	for (int c = 0; c < GASO_MAX_CHANNELS_PER_BUS; c++) {
		contrib[c] = (gaso_frame *)malloc(sizeof(gaso_frame) * (size_t)n);
	}
	int C = inst.channel_count;
	for (int c = 0; c < C; c++) {
		for (int i = 0; i < n; i++) {
			mix[(size_t)c * n + i].l = 0.f;
			mix[(size_t)c * n + i].r = 0.f;
			if (mix64) {
				mix64[((size_t)c * n + i) * 2] = 0.0;
				mix64[((size_t)c * n + i) * 2 + 1] = 0.0;
			}
		}
	}
	for (int s = 0; s < n_src; s++) {
		float peak[2];
		gaso_playback_contribution(&inst, &params[s], &states[s].pd3d, &states[s].pdfx, src + (size_t)s * n, n, contrib, peak);
		for (int c = 0; c < C; c++) {
			for (int i = 0; i < n; i++) {
				mix[(size_t)c * n + i].l += contrib[c][i].l;
				mix[(size_t)c * n + i].r += contrib[c][i].r;
				if (mix64) {
					mix64[((size_t)c * n + i) * 2] += (double)contrib[c][i].l;
					mix64[((size_t)c * n + i) * 2 + 1] += (double)contrib[c][i].r;
				}
			}
		}
		if (peaks) {
			peaks[2 * s] = peak[0];
			peaks[2 * s + 1] = peak[1];
		}
	}
	for (int c = 0; c < GASO_MAX_CHANNELS_PER_BUS; c++) {
		free(contrib[c]);
	}
	free(inst.process_buffer);
	free(inst.temp_buffer);
	free(inst.fx_temp);
}

/* ------------------------------------------------------------------ */
/* CPU baseline arithmetic for HRTF: overlap-save, plain radix-2 FFT.   */
/* ------------------------------------------------------------------ */
/* Twiddles exp(-2 pi i k / n), k < n/2, computed once per length in f64 (not thread-safe; the
 * oracle is single-threaded test infrastructure). */
#define GASO_FFT_MAX 4096
static float tw_re[GASO_FFT_MAX / 2], tw_im[GASO_FFT_MAX / 2];
static int tw_n = 0;

static void fft_twiddles(int n) {
	if (tw_n == n) {
		return;
	}
	for (int k = 0; k < n / 2; k++) {
		double ang = -2.0 * 3.14159265358979323846 * k / n;
		tw_re[k] = (float)cos(ang);
		tw_im[k] = (float)sin(ang);
	}
	tw_n = n;
}

static void fft_radix2(float *re, float *im, int n, int inverse) {
	fft_twiddles(n);
	for (int i = 1, j = 0; i < n; i++) {
		int bit = n >> 1;
		for (; j & bit; bit >>= 1) {
			j ^= bit;
		}
		j ^= bit;
		if (i < j) {
			float t = re[i];
			re[i] = re[j];
			re[j] = t;
			t = im[i];
			im[i] = im[j];
			im[j] = t;
		}
	}
	for (int len = 2; len <= n; len <<= 1) {
		int half = len >> 1;
		int step = n / len;
		for (int i = 0; i < n; i += len) {
			for (int k = 0; k < half; k++) {
				float wr = tw_re[k * step];
				float wi = inverse ? -tw_im[k * step] : tw_im[k * step];
				int a = i + k, b = a + half;
				float tr = re[b] * wr - im[b] * wi;
				float ti = re[b] * wi + im[b] * wr;
				re[b] = re[a] - tr;
				im[b] = im[a] - ti;
				re[a] += tr;
				im[a] += ti;
			}
		}
	}
}

void gaso_hrtf_ols_radix2(const gaso_params *params, gaso_fx_state *st, const gaso_hrtf *hrtf, const gaso_frame *src, gaso_frame *dst, int n) {
	const int T = GASO_HRTF_TAPS;
	const int H = T - 1;
	int L = 1;
	while (L < n + H) {
		L <<= 1;
	}
	float *xx = (float *)malloc(sizeof(float) * (size_t)(H + n));
	hrtf_make_x(params, st, src, xx, n);
	uint32_t dir = params->hrtf_dir < hrtf->dirs ? params->hrtf_dir : 0;
	const float *h[2] = { hrtf->hrir + (size_t)dir * 2 * T, hrtf->hrir + (size_t)dir * 2 * T + T };
	float *buf = (float *)calloc((size_t)L * 6, sizeof(float));
	float *xr = buf, *xi = buf + L, *hr = buf + 2 * L, *hi = buf + 3 * L, *yr = buf + 4 * L, *yi = buf + 5 * L;
	/* window = last L samples ending at the block end, zero-extended on the left */
	for (int i = 0; i < H + n; i++) {
		xr[L - (H + n) + i] = xx[i];
	}
	fft_radix2(xr, xi, L, 0);
	for (int ear = 0; ear < 2; ear++) {
		if (hrtf->spec && hrtf->spec_len == L) {
			const float *sp = hrtf->spec + ((size_t)dir * 2 + ear) * 2 * L;
			hr = (float *)sp;
			hi = (float *)sp + L;
		} else {
			hr = buf + 2 * L;
			hi = buf + 3 * L;
			memset(hr, 0, sizeof(float) * (size_t)L);
			memset(hi, 0, sizeof(float) * (size_t)L);
			memcpy(hr, h[ear], sizeof(float) * T);
			fft_radix2(hr, hi, L, 0);
		}
		for (int k = 0; k < L; k++) {
			yr[k] = xr[k] * hr[k] - xi[k] * hi[k];
			yi[k] = xr[k] * hi[k] + xi[k] * hr[k];
		}
		fft_radix2(yr, yi, L, 1);
		float scale = 1.0f / L;
		for (int i = 0; i < n; i++) {
			float v = yr[L - n + i] * scale;
			if (ear == 0) {
				dst[i].l = v;
			} else {
				dst[i].r = v;
			}
		}
	}
	hrtf_commit(params, st, xx, n);
	st->prev_dir_plus1 = (int32_t)dir + 1;
	free(buf);
	free(xx);
}

void gaso_hrtf_prepare(gaso_hrtf *hrtf, int L) {
	const int T = GASO_HRTF_TAPS;
	gaso_hrtf_release(hrtf);
	hrtf->spec = (float *)calloc((size_t)hrtf->dirs * 2 * 2 * L, sizeof(float));
	hrtf->spec_len = L;
	for (uint32_t d = 0; d < hrtf->dirs; d++) {
		for (int ear = 0; ear < 2; ear++) {
			float *re = hrtf->spec + ((size_t)d * 2 + ear) * 2 * L;
			float *im = re + L;
			memcpy(re, hrtf->hrir + ((size_t)d * 2 + ear) * T, sizeof(float) * T);
			fft_radix2(re, im, L, 0);
		}
	}
}

void gaso_hrtf_release(gaso_hrtf *hrtf) {
	free(hrtf->spec);
	hrtf->spec = NULL;
	hrtf->spec_len = 0;
}

/* ------------------------------------------------------------------ */
/* calculate_spatialization arithmetic (audio_spatializer_3d.cpp)       */
/* real_t = float; the pan law is double (:104-109).                    */
/* ------------------------------------------------------------------ */
#define GASO_CMP_EPSILON 0.00001 /* [ENGINE] CMP_EPSILON */

typedef struct {
	float x, y, z;
} v3;

static float v3_len(v3 a) {
	return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
}

static v3 v3_normalized(v3 a) { /* [ENGINE] Vector3::normalized: zero stays zero */
	float l2 = a.x * a.x + a.y * a.y + a.z * a.z;
	if (l2 == 0) {
		v3 z = { 0, 0, 0 };
		return z;
	}
	float l = sqrtf(l2);
	v3 r = { a.x / l, a.y / l, a.z / l };
	return r;
}

static float v3_dot(v3 a, v3 b) {
	return a.x * b.x + a.y * b.y + a.z * b.z;
}

/* [ENGINE] Basis::xform_inv for an orthonormal basis: transposed product */
static v3 basis_xform_inv(const float b[3][3], v3 v) {
	v3 r = { b[0][0] * v.x + b[1][0] * v.y + b[2][0] * v.z, b[0][1] * v.x + b[1][1] * v.y + b[2][1] * v.z, b[0][2] * v.x + b[1][2] * v.y + b[2][2] * v.z };
	return r;
}

/* audio_spatializer_3d.cpp:123-151 */
static float get_attenuation_db(const gaso_spat3d_config *cfg, const gaso_source_pose *s, float p_distance) {
	float att = 0;
	switch (cfg->attenuation_model) {
		case 0:
			att = (float)(log(1.0 / ((p_distance / cfg->unit_size) + GASO_CMP_EPSILON)) * 8.6858896380650365530225783783321);
			break;
		case 1: {
			float d = (p_distance / cfg->unit_size);
			d *= d;
			att = (float)(log(1.0 / (d + GASO_CMP_EPSILON)) * 8.6858896380650365530225783783321);
		} break;
		case 2:
			att = (float)(-20 * log(p_distance / cfg->unit_size + GASO_CMP_EPSILON));
			break;
		default:
			break;
	}
	att += s->volume_db;
	if (att > s->max_db) {
		att = s->max_db;
	}
	return att;
}

static const float spcap_dirs[7][3] = { /* :46-54, normalized below */
	{ -1.0f, 0.0f, -1.0f }, { 1.0f, 0.0f, -1.0f }, { 0.0f, 0.0f, -1.0f }, { -1.0f, 0.0f, 1.0f }, { 1.0f, 0.0f, 1.0f }, { -1.0f, 0.0f, 0.0f }, { 1.0f, 0.0f, 0.0f }
};

/* calc_output_vol (:112-121) -> stereo pan law (:103-110) or SPCAP (:56-98, :903-938); out[4][2] pre-zeroed */
static void calc_output_vol(const gaso_spat3d_config *cfg, v3 source_dir, float out[4][2]) {
	if (cfg->speaker_mode == 0) {
		float pan_strength = cfg->global_panning_strength * cfg->panning_strength;
		double flatrad = sqrt(source_dir.x * source_dir.x + source_dir.z * source_dir.z);
		double g = (1.0 - pan_strength) * (1.0 - pan_strength);
		g = g < 0.0 ? 0.0 : (g > 1.0 ? 1.0 : g);
		double f = (1.0 - g) / (1.0 + g);
		double cosx = source_dir.x / (flatrad == 0.0 ? 1.0 : flatrad);
		cosx = cosx < -1.0 ? -1.0 : (cosx > 1.0 ? 1.0 : cosx);
		double fcosx = cosx * f;
		out[0][0] = (float)sqrt((-fcosx + 1.0) / 2.0);
		out[0][1] = (float)sqrt((fcosx + 1.0) / 2.0);
		return;
	}
	float tightness = cfg->global_panning_strength * 2.0f;
	tightness *= cfg->panning_strength;
	static const int counts[4] = { 2, 3, 5, 7 };
	int n = counts[cfg->speaker_mode];
	v3 dirs[7];
	float eff[7], sq[7], volumes[7];
	for (int i = 0; i < n; i++) {
		v3 d = { spcap_dirs[i][0], spcap_dirs[i][1], spcap_dirs[i][2] };
		dirs[i] = v3_normalized(d);
	}
	for (int i = 0; i < n; i++) { /* :910-914 */
		float e = 0.0f;
		for (int j = 0; j < n; j++) {
			e += (float)(0.5 * (1.0 + v3_dot(dirs[i], dirs[j])));
		}
		eff[i] = e;
	}
	float sum_sq = 0.0f;
	for (int i = 0; i < n; i++) { /* :928-932 */
		float initial_gain = (float)(0.5 * pow(1.0 + v3_dot(dirs[i], source_dir), tightness) / eff[i]);
		sq[i] = initial_gain * initial_gain;
		sum_sq += sq[i];
	}
	for (int i = 0; i < n; i++) {
		volumes[i] = sqrtf(sq[i] / sum_sq);
	}
	if (cfg->speaker_mode >= 3) { /* :82-97 */
		out[3][0] = volumes[5];
		out[3][1] = volumes[6];
	}
	if (cfg->speaker_mode >= 2) {
		out[2][0] = volumes[3];
		out[2][1] = volumes[4];
	}
	if (cfg->speaker_mode >= 1) {
		out[1][0] = volumes[2];
		out[1][1] = 1.0f;
	}
	out[0][0] = volumes[0];
	out[0][1] = volumes[1];
}

/* calc_reverb_vol, audio_spatializer_3d.cpp:154-197.  [ENGINE] Vector2::lerp(to, w) = this + (to - this) * w per
 * component in f32; AudioServer::get_channel_count() = speaker_mode + 1. */
static void calc_reverb_vol(const gaso_spat3d_config *cfg, const gaso_source_pose *s, const gaso_area_send *area, v3 listener_area_pos, float direct_path_vol[4][2], float reverb_vol[4][2]) {
	for (int i = 0; i < 4; i++) { /* :155-156 */
		reverb_vol[i][0] = 0.0f;
		reverb_vol[i][1] = 0.0f;
	}
	const float uniformity = area->reverb_uniformity;
	const float area_send = area->reverb_amount;
	if (uniformity > 0.0f) {
		const float distance = v3_len(listener_area_pos);
		const float attenuation = gaso_db_to_linear(get_attenuation_db(cfg, s, distance)); /* :163 */
		static const float center_val[4] = { 0.5f, 0.25f, 0.16666f, 0.125f }; /* :166 */
		const int chan_count = cfg->speaker_mode + 1;
		const float center = center_val[chan_count - 1];
		if (attenuation < 1.0f) { /* :170-181: pan the uniform sound */
			v3 rev_pos = listener_area_pos;
			rev_pos.y = 0;
			rev_pos = v3_normalized(rev_pos);
			calc_output_vol(cfg, rev_pos, reverb_vol);
			for (int i = 0; i < chan_count; i++) {
				reverb_vol[i][0] = reverb_vol[i][0] + (center - reverb_vol[i][0]) * attenuation;
				reverb_vol[i][1] = reverb_vol[i][1] + (center - reverb_vol[i][1]) * attenuation;
			}
		} else {
			for (int i = 0; i < chan_count; i++) {
				reverb_vol[i][0] = center;
				reverb_vol[i][1] = center;
			}
		}
		for (int i = 0; i < chan_count; i++) { /* :187-190 */
			for (int e = 0; e < 2; e++) {
				const float to = reverb_vol[i][e] * attenuation;
				float v = direct_path_vol[i][e] + (to - direct_path_vol[i][e]) * uniformity;
				reverb_vol[i][e] = v * area_send;
			}
		}
	} else {
		for (int i = 0; i < 4; i++) { /* :193-195 */
			reverb_vol[i][0] = direct_path_vol[i][0] * area_send;
			reverb_vol[i][1] = direct_path_vol[i][1] * area_send;
		}
	}
}

int gaso_calc_spatialization(const gaso_spat3d_config *cfg, const gaso_source_pose *s, const gaso_listener *listeners, int n_listeners, int32_t *was_further, gaso_params *out) {
	float reverb[4][2];
	return gaso_calc_spatialization_area(cfg, s, listeners, n_listeners, was_further, NULL, NULL, out, reverb);
}

int gaso_calc_spatialization_area(const gaso_spat3d_config *cfg, const gaso_source_pose *s, const gaso_listener *listeners, int n_listeners, int32_t *was_further, const gaso_area_send *area, const float *listener_area_pos, gaso_params *out, float (*out_reverb)[2]) {
	if (area && !area->present) {
		area = NULL;
	}
	const int area_reverb = area && area->using_reverb_bus;
	const int area_uniform = area_reverb && area->reverb_uniformity > 0 && listener_area_pos; /* :349,365 */
	float reverb_volume[4][2] = { { 0 } };
	v3 global_pos = { s->position[0], s->position[1], s->position[2] };
	v3 linear_velocity = { 0, 0, 0 };
	if (cfg->doppler_tracking != 0) { /* :295-297 */
		linear_velocity.x = s->velocity[0];
		linear_velocity.y = s->velocity[1];
		linear_velocity.z = s->velocity[2];
	}
	float log_pitch_scale = 0.0f, log_pitch_weight = 0.0f;
	float output_volume[4][2] = { { 0 } };
	int has_any_listener_in_range = 0;
	out->linear_attenuation = 0.0f; /* a fresh SpatializerParameters3D (audio_spatializer_3d.h:67-68) */
	out->attenuation_filter_cutoff_hz = 5000.0f;
	float best_mult = -1.0f;
	v3 best_local = { 0, 0, -1 };

	for (int li = 0; li < n_listeners; li++) {
		const gaso_listener *L = &listeners[li];
		v3 rel = { global_pos.x - L->origin[0], global_pos.y - L->origin[1], global_pos.z - L->origin[2] };
		v3 local_pos = basis_xform_inv(L->basis, rel); /* :343 */
		float dist = v3_len(local_pos);
		float multiplier = gaso_db_to_linear(get_attenuation_db(cfg, s, dist)); /* :359 */
		v3 lap = { 0, 0, 0 };
		if (area_uniform) { /* :350-353 */
			lap.x = listener_area_pos[li * 3 + 0];
			lap.y = listener_area_pos[li * 3 + 1];
			lap.z = listener_area_pos[li * 3 + 2];
		}
		if (cfg->max_distance > 0) { /* :361-374 */
			float total_max = cfg->max_distance;
			if (area_uniform) { /* :365-367 */
				const float l = v3_len(lap);
				total_max = total_max > l ? total_max : l;
			}
			if (dist > total_max || total_max > cfg->max_distance) { /* :369 */
				continue;
			}
			double m = 1.0 - (dist / cfg->max_distance);
			multiplier = (float)(multiplier * (m > 0 ? m : 0));
		}
		has_any_listener_in_range = 1;
		float db_att = (float)((1.0 - (1.0 < multiplier ? 1.0 : multiplier)) * cfg->attenuation_filter_db); /* :376 */
		if (cfg->emission_angle_enabled) { /* :378-385 */
			v3 fw = { s->forward[0], s->forward[1], s->forward[2] };
			float c = v3_dot(v3_normalized(rel), v3_normalized(fw));
			/* [ENGINE] Math::acos clamps: x < -1 -> pi, x > 1 -> 0 (recollection of the engine source, unpinned) */
			float angle = (c < -1.0f ? 3.14159265358979323846f : (c > 1.0f ? 0.0f : acosf(c))) * (float)(180.0 / 3.14159265358979323846);
			if (angle > cfg->emission_angle) {
				db_att -= -cfg->emission_angle_filter_attenuation_db;
			}
		}
		out->linear_attenuation = gaso_db_to_linear(db_att); /* :387, last listener in range wins */
		out->attenuation_filter_cutoff_hz = cfg->attenuation_filter_cutoff_hz; /* :388 */

		float tmp_volume[4][2] = { { 0 } };
		calc_output_vol(cfg, local_pos, tmp_volume); /* :391, local_pos is passed un-normalised */
		for (int k = 0; k < 4; k++) { /* :393-396 */
			tmp_volume[k][0] = multiplier * tmp_volume[k][0];
			tmp_volume[k][1] = multiplier * tmp_volume[k][1];
			/* _apply_max_volume (:257-265): MAX(tgt, src) = tgt > src ? tgt : src */
			output_volume[k][0] = output_volume[k][0] > tmp_volume[k][0] ? output_volume[k][0] : tmp_volume[k][0];
			output_volume[k][1] = output_volume[k][1] > tmp_volume[k][1] ? output_volume[k][1] : tmp_volume[k][1];
		}
		if (area_reverb) { /* :399-402 */
			float tmp_reverb[4][2];
			calc_reverb_vol(cfg, s, area, lap, tmp_volume, tmp_reverb);
			for (int k = 0; k < 4; k++) {
				reverb_volume[k][0] = reverb_volume[k][0] > tmp_reverb[k][0] ? reverb_volume[k][0] : tmp_reverb[k][0];
				reverb_volume[k][1] = reverb_volume[k][1] > tmp_reverb[k][1] ? reverb_volume[k][1] : tmp_reverb[k][1];
			}
		}
		if (multiplier > best_mult) {
			best_mult = multiplier;
			best_local = local_pos;
		}
		if (cfg->doppler_tracking != 0) { /* :405-431 */
			v3 dv = { linear_velocity.x - L->velocity[0], linear_velocity.y - L->velocity[1], linear_velocity.z - L->velocity[2] };
			v3 local_velocity = basis_xform_inv(L->basis, dv);
			if (local_velocity.x != 0 || local_velocity.y != 0 || local_velocity.z != 0) {
				float approaching = v3_dot(v3_normalized(local_pos), v3_normalized(local_velocity));
				float velocity = v3_len(local_velocity);
				float dps = s->pitch_scale * cfg->doppler_speed_of_sound / (cfg->doppler_speed_of_sound + velocity * approaching);
				dps = dps < (float)(1 / 8.0) ? (float)(1 / 8.0) : (dps > 8.0f ? 8.0f : dps);
				float weight = 0.0f;
				for (int k = 0; k < 4; k++) {
					weight = tmp_volume[k][0] > weight ? tmp_volume[k][0] : weight;
					weight = tmp_volume[k][1] > weight ? tmp_volume[k][1] : weight;
				}
				log_pitch_scale += weight * log2f(dps);
				log_pitch_weight += weight;
			}
		}
	}
	if (log_pitch_weight > 0) { /* :434-438 */
		out->pitch_scale = powf(2.0f, log_pitch_scale / log_pitch_weight);
	} else {
		out->pitch_scale = s->pitch_scale;
	}
	for (int k = 0; k < 4; k++) { /* :469 */
		out->mix_volumes[k][0] = output_volume[k][0];
		out->mix_volumes[k][1] = output_volume[k][1];
		out_reverb[k][0] = reverb_volume[k][0];
		out_reverb[k][1] = reverb_volume[k][1];
	}
	const int skip_setting_volumes = !has_any_listener_in_range && *was_further; /* :472-479 */
	*was_further = !has_any_listener_in_range;
	out->update_parameters = skip_setting_volumes ? 0u : 1u;

	if (cfg->hrtf_n_az > 0 && cfg->hrtf_n_el > 0) { /* NEW: gain and nearest grid direction towards the loudest listener */
		out->hrtf_gain = has_any_listener_in_range ? best_mult : 0.0f;
		const double two_pi = 6.2831853071795864769252867666;
		double az = atan2((double)best_local.x, (double)-best_local.z);
		double flat = sqrt((double)best_local.x * best_local.x + (double)best_local.z * best_local.z);
		double el = atan2((double)best_local.y, flat);
		long ai = lround(az / two_pi * cfg->hrtf_n_az);
		ai = ((ai % (long)cfg->hrtf_n_az) + (long)cfg->hrtf_n_az) % (long)cfg->hrtf_n_az;
		long ei = cfg->hrtf_n_el > 1 ? lround((el + two_pi / 4) / (two_pi / 2) * (cfg->hrtf_n_el - 1)) : 0;
		ei = ei < 0 ? 0 : (ei > (long)cfg->hrtf_n_el - 1 ? (long)cfg->hrtf_n_el - 1 : ei);
		out->hrtf_dir = (uint32_t)(ei * cfg->hrtf_n_az + ai);
	}
	return has_any_listener_in_range;
}
