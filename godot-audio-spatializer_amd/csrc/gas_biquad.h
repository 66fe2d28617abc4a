// gas_biquad.h -- what the two high-shelf kernels (k_biquad_mix.hip, k_biquad_pipe.hip) share: the engine's
// coefficient preparation.  Anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "gas_internal.h"

// No FMA contraction anywhere in the high-shelf path (the including files say why): the f64 coefficient preparation
// below and the f32 recurrences must round exactly where the engine's C++ does.
#pragma clang fp contract(off)

namespace {

struct Coeffs {
	float b0, b1, b2, a1, a2;
};

// [ENGINE] AudioFilterSW::prepare_coefficients, HIGHSHELF, resonance 1, stages 1:
// f64 arithmetic, members stored f32 before the division by a0, feedback terms negated.
__device__ inline Coeffs highshelf_coeffs(float sampling_rate, float cutoff_hz, float gain_lin) {
	int sr_limit = (int)(sampling_rate / 2) + 512;
	double final_cutoff = ((double)cutoff_hz > sr_limit) ? (double)sr_limit : (double)cutoff_hz;
	if (final_cutoff < 1) {
		final_cutoff = 1;
	}
	double omega = 6.2831853071795864769252867666 * final_cutoff / (double)sampling_rate;
	double sin_v = sin(omega);
	double cos_v = cos(omega);
	double A = gain_lin;
	if (A < 0.001) {
		A = 0.001;
	}
	double beta = sqrt(A); // sqrt(Q) = 1
	double a0 = (A + 1.0) - (A - 1.0) * cos_v + beta * sin_v;
	Coeffs c;
	c.b0 = (float)(A * ((A + 1.0) + (A - 1.0) * cos_v + beta * sin_v));
	c.b1 = (float)(-2.0 * A * ((A - 1.0) + (A + 1.0) * cos_v));
	c.b2 = (float)(A * ((A + 1.0) + (A - 1.0) * cos_v - beta * sin_v));
	c.a1 = (float)(2.0 * ((A - 1.0) - (A + 1.0) * cos_v));
	c.a2 = (float)((A + 1.0) - (A - 1.0) * cos_v - beta * sin_v);
	c.b0 = (float)((double)c.b0 / a0);
	c.b1 = (float)((double)c.b1 / a0);
	c.b2 = (float)((double)c.b2 / a0);
	c.a1 = (float)((double)c.a1 / (0.0 - a0));
	c.a2 = (float)((double)c.a2 / (0.0 - a0));
	return c;
}

// [ENGINE] AudioFilterSW::Processor::process_one: one frame through the processor, the engine's operation order.  For
// translation units that run with FMA contraction on elsewhere (k_hrtf_uni.hip): defined here, it keeps this file's
// contract(off) when inlined.
__device__ __forceinline__ float biquad_process_one(const Coeffs &co, float xi, float &a1, float &a2, float &b1, float &b2) {
	const float yi = xi * co.b0 + b1 * co.b1 + b2 * co.b2 + a1 * co.a1 + a2 * co.a2;
	a2 = a1;
	b2 = b1;
	b1 = xi;
	a1 = yi;
	return yi;
}

// [ENGINE] AudioFilterSW::prepare_coefficients for the other modes AudioEffectFilter's subclasses select (recollection
// of servers/audio/audio_filter_sw.cpp, unpinned like the rest of SURVEY.md Appendix B): LOWPASS, HIGHPASS, BANDPASS
// (Q doubled), NOTCH, LOWSHELF; one stage (FILTER_6DB).  Same shape as above: f64 arithmetic, members stored f32, then
// normalised by a0 with the feedback terms negated.
__device__ inline Coeffs filter_coeffs(int kind, float sampling_rate, float cutoff_hz, float resonance, float gain_lin) {
	int sr_limit = (int)(sampling_rate / 2) + 512;
	double final_cutoff = ((double)cutoff_hz > sr_limit) ? (double)sr_limit : (double)cutoff_hz;
	if (final_cutoff < 1) {
		final_cutoff = 1;
	}
	double omega = 6.2831853071795864769252867666 * final_cutoff / (double)sampling_rate;
	double sin_v = sin(omega);
	double cos_v = cos(omega);
	double Q = resonance;
	if (Q <= 0.0) {
		Q = 0.0001;
	}
	if (kind == GAS_FX_BANDPASS) {
		Q *= 2.0;
	}
	double tmpgain = gain_lin;
	if (tmpgain < 0.001) {
		tmpgain = 0.001;
	}
	double alpha = sin_v / (2 * Q);
	double a0 = 1.0 + alpha;
	Coeffs c;
	switch (kind) {
		case GAS_FX_LOWPASS:
			c.b0 = (float)((1.0 - cos_v) / 2.0);
			c.b1 = (float)(1.0 - cos_v);
			c.b2 = (float)((1.0 - cos_v) / 2.0);
			c.a1 = (float)(-2.0 * cos_v);
			c.a2 = (float)(1.0 - alpha);
			break;
		case GAS_FX_HIGHPASS:
			c.b0 = (float)((1.0 + cos_v) / 2.0);
			c.b1 = (float)(-(1.0 + cos_v));
			c.b2 = (float)((1.0 + cos_v) / 2.0);
			c.a1 = (float)(-2.0 * cos_v);
			c.a2 = (float)(1.0 - alpha);
			break;
		case GAS_FX_BANDPASS:
			c.b0 = (float)(alpha * sqrt(Q + 1));
			c.b1 = 0.0f;
			c.b2 = (float)(-alpha * sqrt(Q + 1));
			c.a1 = (float)(-2.0 * cos_v);
			c.a2 = (float)(1.0 - alpha);
			break;
		case GAS_FX_NOTCH:
			c.b0 = 1.0f;
			c.b1 = (float)(-2.0 * cos_v);
			c.b2 = 1.0f;
			c.a1 = (float)(-2.0 * cos_v);
			c.a2 = (float)(1.0 - alpha);
			break;
		default: { // GAS_FX_LOWSHELF
			double tmpq = sqrt(Q);
			if (tmpq <= 0) {
				tmpq = 0.001;
			}
			double beta = sqrt(tmpgain) / tmpq;
			a0 = (tmpgain + 1.0) + (tmpgain - 1.0) * cos_v + beta * sin_v;
			c.b0 = (float)(tmpgain * ((tmpgain + 1.0) - (tmpgain - 1.0) * cos_v + beta * sin_v));
			c.b1 = (float)(2.0 * tmpgain * ((tmpgain - 1.0) - (tmpgain + 1.0) * cos_v));
			c.b2 = (float)(tmpgain * ((tmpgain + 1.0) - (tmpgain - 1.0) * cos_v - beta * sin_v));
			c.a1 = (float)(-2.0 * ((tmpgain - 1.0) + (tmpgain + 1.0) * cos_v));
			c.a2 = (float)((tmpgain + 1.0) + (tmpgain - 1.0) * cos_v - beta * sin_v);
		} break;
	}
	c.b0 = (float)((double)c.b0 / a0);
	c.b1 = (float)((double)c.b1 / a0);
	c.b2 = (float)((double)c.b2 / a0);
	c.a1 = (float)((double)c.a1 / (0.0 - a0));
	c.a2 = (float)((double)c.a2 / (0.0 - a0));
	return c;
}

// [ENGINE] Math::db_to_linear (f32 expf of db * ln(10)/20)
__device__ inline float fx_db_to_linear(float db) {
	return expf(db * 0.11512925464970228420089957273422f);
}

} // namespace
