// k_shelf_scan.hip -- one biquad STAGE of a staged effect chain (rows in -> rows out) for callbacks with too few
// sources to hide a 512-step recurrence: [ENGINE] AudioEffectFilterInstance::process at FILTER_6DB (one AudioFilterSW
// stage per ear, coefficients snapped per block) -- the high shelf of gd_spatializer.gd:14-19 and the other filter
// kinds -- as effect j of AudioSpatializerInstanceEffect::process_frames' chain (audio_spatializer_effect.cpp:52-76).
//
// Why a second form.  k_biquad_mix gives every (source, ear) one lane that walks the F frames in order: 8192 sources are
// 256 single-wave workgroups, one per CU, and a lone wave issues one VALU instruction per ~7 cycles (profiles/
// r03_notes.md section 2): 512 steps x ~12 instructions = 35 us, whatever the memory system does -- more than the HRTF
// launch the stage feeds.  The recurrence is linear with coefficients that are CONSTANT inside the block, so it splits:
// here ONE WAVE owns one source, lane l its frames [P l, P l + P) (P = F / 64) of both ears, and
//   1. u[k] = b0 x[k] + b1 x[k-1] + b2 x[k-2]                      (the two frames in front of a lane's first come
//      from the lane below; lane 0 takes the processor's hb1, hb2)
//   2. w[k] = u[k] + a1 w[k-1] + a2 w[k-2], w[-1] = w[-2] = 0       (the lane's response from rest)
//   3. what a state (y[-1], y[-2]) entering a lane adds to its k-th output is p[k] y[-1] + q[k] y[-2], with p, q the two
//      homogeneous solutions (the same for every lane); so a lane's outgoing state is s_l = M s_(l-1) + c_l,
//      M = [[p[P-1], q[P-1]], [p[P-2], q[P-2]]], c_l = (w[P-1], w[P-2]) -- an affine recurrence over the 64 lanes, solved
//      by a six-step doubling scan (M, M^2, M^4, ... by squaring; lane 0's c absorbs the processor's ha1, ha2)
//   4. y[k] = w[k] + p[k] s_(l-1).x + q[k] s_(l-1).y
// ~300 VALU instructions per source instead of a 512-step chain: the stage is bound by its 8 KiB of row traffic per
// source.  Same filter, DIFFERENT rounding from the serial form: the scan's sums associate differently and the powers of
// M carry their own error, which a pole radius near 1 amplifies like the recurrence itself does.  A source whose poles
// lie outside r^2 <= 0.9 therefore takes the serial path (two lanes walk the row out of LDS, the engine's operation
// order, no FMA contraction: bitwise k_biquad_mix) -- the wave-uniform choice below; inside it the scan tracks the
// oracle to ~2e-6 relative (tests/test_gpu_chains.py).  NEW arrangement of [ENGINE] arithmetic: parity unpinned like
// the rest of SURVEY.md Appendix B.
#include <cstdlib>

#include "gas_biquad.h"

#pragma clang fp contract(off)

namespace {

constexpr int SCAN_WAVES = 4; // sources per workgroup

__device__ __forceinline__ float lane_up(float v, int d) { // value of lane - d (undefined for lane < d: callers mask)
	return __shfl_up(v, d, 64);
}

template <int P>
__global__ __launch_bounds__(SCAN_WAVES * 64) void k_shelf_scan(gas_group_args g, gas_dev_state st, uint32_t c0, float mix_rate, float *__restrict__ rows_out, int highshelf, int fx_kind) {
	constexpr int F = P * 64;
	__shared__ float stage[SCAN_WAVES][F * 2]; // the serial path's row (x in, y out)
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	const uint32_t e = blockIdx.x * SCAN_WAVES + wave;
	if (e >= g.n) { // wave-uniform; no barrier below
		return;
	}
	const uint32_t slot = g.slots[e];
	const uint32_t row = g.rows ? g.rows[e] : e;
	Coeffs co;
	if (highshelf) {
		const gas_params *Pm = st.params + slot;
		co = highshelf_coeffs(mix_rate, Pm->fx_shelf_cutoff_hz, Pm->fx_shelf_gain);
	} else {
		const gas_fx_settings *S = st.fxs + slot;
		co = filter_coeffs(fx_kind, mix_rate, S->filter_cutoff_hz[c0], S->filter_resonance[c0], S->filter_gain[c0]);
	}
	// processor state of the two ears of chain position c0 (audio_spatializer_3d.cpp:887-894's stream index)
	float *bq = st.bq;
	const size_t bs = st.bq_stride;
	const size_t s0 = ((size_t)slot * 4 + c0) * 2;
	float ha1[2], ha2[2], hb1[2], hb2[2];
#pragma unroll
	for (int ear = 0; ear < 2; ear++) {
		ha1[ear] = bq[BQ_HA1 * bs + s0 + ear];
		ha2[ear] = bq[BQ_HA2 * bs + s0 + ear];
		hb1[ear] = bq[BQ_HB1 * bs + s0 + ear];
		hb2[ear] = bq[BQ_HB2 * bs + s0 + ear];
	}
	// this lane's frames: P consecutive AudioFrames = P / 2 float4
	float x[2][P];
	const float4 *src4 = reinterpret_cast<const float4 *>(g.src + (size_t)row * F + (size_t)lane * P);
#pragma unroll
	for (int k = 0; k < P / 2; k++) {
		const float4 v = src4[k];
		x[0][2 * k] = v.x;
		x[1][2 * k] = v.y;
		x[0][2 * k + 1] = v.z;
		x[1][2 * k + 1] = v.w;
	}
	float y[2][P];
	const bool serial = !(fabsf(co.a2) <= 0.9f); // wave-uniform (coefficients are the source's); NaN coefficients go serial too

	if (serial) {
		// the engine's own loop, two lanes (one per ear) over the row staged in LDS: bitwise k_biquad_mix's FX modes
		float *sr = stage[wave];
#pragma unroll
		for (int k = 0; k < P; k++) {
			sr[((size_t)lane * P + k) * 2] = x[0][k];
			sr[((size_t)lane * P + k) * 2 + 1] = x[1][k];
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		if (lane < 2) {
			float a1 = ha1[lane], a2 = ha2[lane], b1 = hb1[lane], b2 = hb2[lane];
			for (int i = 0; i < F; i++) {
				const float xi = sr[i * 2 + lane];
				const float yi = xi * co.b0 + b1 * co.b1 + b2 * co.b2 + a1 * co.a1 + a2 * co.a2; // [ENGINE] process_one
				a2 = a1;
				b2 = b1;
				b1 = xi;
				a1 = yi;
				sr[i * 2 + lane] = yi;
			}
			bq[BQ_HA1 * bs + s0 + lane] = a1;
			bq[BQ_HA2 * bs + s0 + lane] = a2;
			bq[BQ_HB1 * bs + s0 + lane] = b1;
			bq[BQ_HB2 * bs + s0 + lane] = b2;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
		for (int k = 0; k < P; k++) {
			y[0][k] = sr[((size_t)lane * P + k) * 2];
			y[1][k] = sr[((size_t)lane * P + k) * 2 + 1];
		}
	} else {
		// homogeneous solutions p (state (1, 0)) and q (state (0, 1)): the same in every lane and for both ears
		float p[P], q[P];
		p[0] = co.a1;
		q[0] = co.a2;
		p[1] = __builtin_fmaf(co.a1, p[0], co.a2);
		q[1] = co.a1 * q[0];
#pragma unroll
		for (int k = 2; k < P; k++) {
			p[k] = __builtin_fmaf(co.a1, p[k - 1], co.a2 * p[k - 2]);
			q[k] = __builtin_fmaf(co.a1, q[k - 1], co.a2 * q[k - 2]);
		}
		float sx[2], sy[2]; // the state leaving this lane, per ear
#pragma unroll
		for (int ear = 0; ear < 2; ear++) {
			float xm1 = lane_up(x[ear][P - 1], 1), xm2 = lane_up(x[ear][P - 2], 1);
			if (lane == 0) {
				xm1 = hb1[ear];
				xm2 = hb2[ear];
			}
			float w1 = 0.0f, w2 = 0.0f; // w[k-1], w[k-2]
#pragma unroll
			for (int k = 0; k < P; k++) {
				const float xa = k >= 1 ? x[ear][k - 1] : xm1;
				const float xb = k >= 2 ? x[ear][k - 2] : (k == 1 ? xm1 : xm2);
				const float u = __builtin_fmaf(co.b2, xb, __builtin_fmaf(co.b1, xa, co.b0 * x[ear][k]));
				const float w = __builtin_fmaf(co.a2, w2, __builtin_fmaf(co.a1, w1, u));
				y[ear][k] = w;
				w2 = w1;
				w1 = w;
			}
			sx[ear] = y[ear][P - 1];
			sy[ear] = y[ear][P - 2];
			if (lane == 0) { // the processor's history enters through lane 0
				sx[ear] = __builtin_fmaf(p[P - 1], ha1[ear], __builtin_fmaf(q[P - 1], ha2[ear], sx[ear]));
				sy[ear] = __builtin_fmaf(p[P - 2], ha1[ear], __builtin_fmaf(q[P - 2], ha2[ear], sy[ear]));
			}
		}
		// inclusive scan of s_l = M s_(l-1) + c_l over the lanes, distances 1, 2, 4, ... with M, M^2, M^4, ...
		float m00 = p[P - 1], m01 = q[P - 1], m10 = p[P - 2], m11 = q[P - 2];
#pragma unroll
		for (int d = 1; d < 64; d *= 2) {
#pragma unroll
			for (int ear = 0; ear < 2; ear++) {
				const float ox = lane_up(sx[ear], d), oy = lane_up(sy[ear], d);
				if (lane >= d) {
					sx[ear] = __builtin_fmaf(m00, ox, __builtin_fmaf(m01, oy, sx[ear]));
					sy[ear] = __builtin_fmaf(m10, ox, __builtin_fmaf(m11, oy, sy[ear]));
				}
			}
			const float n00 = __builtin_fmaf(m00, m00, m01 * m10), n01 = __builtin_fmaf(m00, m01, m01 * m11);
			const float n10 = __builtin_fmaf(m10, m00, m11 * m10), n11 = __builtin_fmaf(m10, m01, m11 * m11);
			m00 = n00;
			m01 = n01;
			m10 = n10;
			m11 = n11;
		}
#pragma unroll
		for (int ear = 0; ear < 2; ear++) {
			float ix = lane_up(sx[ear], 1), iy = lane_up(sy[ear], 1); // the state entering this lane
			if (lane == 0) {
				ix = ha1[ear];
				iy = ha2[ear];
			}
#pragma unroll
			for (int k = 0; k < P; k++) {
				y[ear][k] = __builtin_fmaf(p[k], ix, __builtin_fmaf(q[k], iy, y[ear][k]));
			}
		}
		if (lane == 63) { // the processor after the block: last two outputs, last two inputs
#pragma unroll
			for (int ear = 0; ear < 2; ear++) {
				bq[BQ_HA1 * bs + s0 + ear] = y[ear][P - 1];
				bq[BQ_HA2 * bs + s0 + ear] = y[ear][P - 2];
				bq[BQ_HB1 * bs + s0 + ear] = x[ear][P - 1];
				bq[BQ_HB2 * bs + s0 + ear] = x[ear][P - 2];
			}
		}
	}
	if (lane < 2) { // the snapped coefficients are part of the processor too (k_biquad_mix stores them)
		bq[BQ_B0 * bs + s0 + lane] = co.b0;
		bq[BQ_B1 * bs + s0 + lane] = co.b1;
		bq[BQ_B2 * bs + s0 + lane] = co.b2;
		bq[BQ_A1 * bs + s0 + lane] = co.a1;
		bq[BQ_A2 * bs + s0 + lane] = co.a2;
	}
	float4 *dst4 = reinterpret_cast<float4 *>(rows_out + ((size_t)e * F + (size_t)lane * P) * 2);
#pragma unroll
	for (int k = 0; k < P / 2; k++) {
		dst4[k] = make_float4(y[0][2 * k], y[1][2 * k], y[0][2 * k + 1], y[1][2 * k + 1]);
	}
}

} // namespace

// Whether a rows-out filter stage over n sources runs the scan form: callbacks small enough that k_biquad_mix's one
// wave per 32 sources leaves SIMDs empty (below 32768 sources there are fewer than two of its waves per SIMD), frames a
// multiple of 128 (a lane owns an even number of frames).  GAS_SHELF_SCAN=0 keeps the serial stage (A/B).
bool gas_shelf_scan_applies(int mode, uint32_t n, uint32_t frames) {
	const char *e = std::getenv("GAS_SHELF_SCAN");
	if (e && e[0] == '0') {
		return false;
	}
	return (mode == GAS_MODE_FX_HIGHSHELF || mode == GAS_MODE_FX_FILTER) && n >= 512 && n < 32768 && frames % 128 == 0 && frames <= 512;
}

hipError_t gas_launch_shelf_scan(hipStream_t stream, int mode, const gas_group_args &g, const gas_dev_state &st, uint32_t frames, uint32_t chain_pos, float mix_rate, float *rows_out, int fx_kind) {
	if (g.n == 0) {
		return hipSuccess;
	}
	const dim3 grid((g.n + SCAN_WAVES - 1) / SCAN_WAVES), block(SCAN_WAVES * 64);
	const int hs = mode == GAS_MODE_FX_HIGHSHELF ? 1 : 0;
	switch (frames / 64) {
		case 2:
			hipLaunchKernelGGL(k_shelf_scan<2>, grid, block, 0, stream, g, st, chain_pos, mix_rate, rows_out, hs, fx_kind);
			break;
		case 4:
			hipLaunchKernelGGL(k_shelf_scan<4>, grid, block, 0, stream, g, st, chain_pos, mix_rate, rows_out, hs, fx_kind);
			break;
		case 6:
			hipLaunchKernelGGL(k_shelf_scan<6>, grid, block, 0, stream, g, st, chain_pos, mix_rate, rows_out, hs, fx_kind);
			break;
		case 8:
			hipLaunchKernelGGL(k_shelf_scan<8>, grid, block, 0, stream, g, st, chain_pos, mix_rate, rows_out, hs, fx_kind);
			break;
		default:
			return hipErrorInvalidValue;
	}
	return hipGetLastError();
}
