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

} // namespace
