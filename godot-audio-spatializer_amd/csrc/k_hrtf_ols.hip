// k_hrtf_ols.hip -- per-source HRTF convolution by overlap-save FFT, fused with the (optional)
// 8-tap early-reflection effect in front of it and with the N-source -> stereo partial sum.
//
// NEW arithmetic (the reference has no HRTF/FFT/convolution); it sits where
// AudioSpatializerInstanceEffect::process_frames runs its effect chain (audio_spatializer_effect.cpp:52-76)
// and feeds the accumulate + per-source peak of _mix_from_playback_list (audio_spatializer.cpp:449-461).
// Semantics are fixed by oracle/gas_oracle.c (fx_early_reflections, fx_hrtf):
//   e   = src + sum_k er_gain[k] * src delayed by er_delay[k]           (only when the chain is [ER, HRTF])
//   x   = ((e.l + e.r) * 0.5) * (g1*t + (1-t)*g0),  t = i/F             (gain ramp as audio_spatializer_3d.cpp:591-592)
//   out = (hrir[dir][L] * x, hrir[dir][R] * x)  over  hist ++ x         (256 taps, direction switches per block)
//
// Algorithm per source and callback (DESIGN.md "k_hrtf_ols"): the callback's F frames are cut into two
// sub-blocks of S = F/2; each needs a 512-sample window (S new + 512-S history >= 255) and the two real
// windows ride one complex FFT as  z = a + i b.  Because h_L and h_R are real,  IFFT(Z * H_L) =
// a*h_L + i b*h_L, so one forward and two inverse 512-point FFTs give both sub-blocks of both ears with
// no spectrum unpacking: 3 FFT-512 per source per callback for any F in {128,256,384,512}.
//
// Mapping (CDNA4, wave64): one wave owns one source at a time; lane l holds points l + 64 j (j = 0..7)
// in registers, so every global access is lane-contiguous.  512 = 8*8*8: three in-register radix-8 passes
// with two 8x8 lane<->register transposes through the wave's private LDS slice (row strides 72 and 66
// float2 = conflict-free for ds_write_b64/ds_read_b64, see MI355X_MICROARCH.md LDS banking).  Twiddles
// and the wave's running stereo sum stay in registers across its sources; HRIR spectra come from a
// lane-major table (16 B/lane, L2/Infinity-Cache resident).  Waves of a workgroup combine through LDS
// into one partial mix; k_mix_reduce adds the partials in fixed order (no float atomics).
//
// Bound: HBM.  Algorithmic bytes/source = F*8 (source) + 2*hist_len*4 (history r+w) + 128 (params) + 8
// (peak) + 8 (gain state), + the ring traffic (8 taps * F * 8 read + F * 8 write) with early reflections.
#include "gas_internal.h"

namespace {

constexpr int WAVES = 4;
constexpr int LDS_F2_PER_WAVE = 8 * 72; // float2 units; exchange 1 uses 8x72, exchange 2 uses 8x66
constexpr float S2 = 0.70710678118654752440f;

__device__ __forceinline__ float2 cadd(float2 a, float2 b) {
	return make_float2(a.x + b.x, a.y + b.y);
}
__device__ __forceinline__ float2 csub(float2 a, float2 b) {
	return make_float2(a.x - b.x, a.y - b.y);
}
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
	return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { // a * conj(b)
	return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
// multiply by -i (forward) / +i (inverse)
template <bool INV>
__device__ __forceinline__ float2 rot(float2 a) {
	return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

// 8-point DFT in registers (tools/fft512_prototype.py dft8).
template <bool INV>
__device__ __forceinline__ void dft8(float2 (&v)[8]) {
	float2 a0 = cadd(v[0], v[4]), a1 = csub(v[0], v[4]);
	float2 a2 = cadd(v[2], v[6]), a3 = rot<INV>(csub(v[2], v[6]));
	float2 a4 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
	float2 a6 = cadd(v[3], v[7]), a7 = rot<INV>(csub(v[3], v[7]));
	float2 b0 = cadd(a0, a2), b2 = csub(a0, a2);
	float2 b1 = cadd(a1, a3), b3 = csub(a1, a3);
	float2 b4 = cadd(a4, a6), b6 = rot<INV>(csub(a4, a6));
	float2 b5 = cadd(a5, a7), b7 = csub(a5, a7);
	if (INV) { // W8^-1 = (1+i)/sqrt2, W8^-3 = (-1+i)/sqrt2
		b5 = make_float2(S2 * (b5.x - b5.y), S2 * (b5.x + b5.y));
		b7 = make_float2(S2 * (-b7.x - b7.y), S2 * (b7.x - b7.y));
	} else { // W8^1 = (1-i)/sqrt2, W8^3 = (-1-i)/sqrt2
		b5 = make_float2(S2 * (b5.x + b5.y), S2 * (b5.y - b5.x));
		b7 = make_float2(S2 * (b7.y - b7.x), S2 * (-b7.x - b7.y));
	}
	v[0] = cadd(b0, b4);
	v[1] = cadd(b1, b5);
	v[2] = cadd(b2, b6);
	v[3] = cadd(b3, b7);
	v[4] = csub(b0, b4);
	v[5] = csub(b1, b5);
	v[6] = csub(b2, b6);
	v[7] = csub(b3, b7);
}

// Orders this wave's LDS traffic for the compiler; the hardware executes one wave's DS
// instructions in order, so no wait is needed between a wave's own store and load.
__device__ __forceinline__ void wave_lds_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 512-point FFT of one wave: in/out v[j] of lane l = element l + 64 j (natural order both sides).
// t1[k0] = W64^(n1 k0) with n1 = l>>3; t2[k1] = W512^(n0 (k0' + 8 k1)) with n0 = l&7, k0' = l>>3.
template <bool INV>
__device__ __forceinline__ void fft512(float2 (&v)[8], const float2 (&t1)[8], const float2 (&t2)[8], float2 *lds, int lane) {
	const int hi = lane >> 3, lo = lane & 7;
	dft8<INV>(v);
#pragma unroll
	for (int k = 1; k < 8; k++) {
		v[k] = INV ? cmulc(v[k], t1[k]) : cmul(v[k], t1[k]);
	}
#pragma unroll
	for (int k = 0; k < 8; k++) {
		lds[k * 72 + lane] = v[k];
	}
	wave_lds_sync();
#pragma unroll
	for (int k = 0; k < 8; k++) {
		v[k] = lds[hi * 72 + k * 8 + lo];
	}
	wave_lds_sync();
	dft8<INV>(v);
#pragma unroll
	for (int k = 0; k < 8; k++) {
		v[k] = INV ? cmulc(v[k], t2[k]) : cmul(v[k], t2[k]);
	}
#pragma unroll
	for (int k = 0; k < 8; k++) {
		lds[lo * 66 + k * 8 + hi] = v[k];
	}
	wave_lds_sync();
#pragma unroll
	for (int k = 0; k < 8; k++) {
		v[k] = lds[k * 66 + lane];
	}
	wave_lds_sync();
	dft8<INV>(v);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1) {
		float o = __shfl_xor(v, m);
		v = o > v ? o : v;
	}
	return v;
}

// SQ = S/64 = F/128: 4 for F = 512, 2 for F = 256.
template <int SQ, bool WITH_ER>
__global__ __launch_bounds__(WAVES * 64) void k_hrtf_ols(gas_group_args g, gas_dev_state st, gas_hrtf_table tab, const float2 *__restrict__ tw, uint32_t spw, uint32_t er_R, float *__restrict__ partials, uint32_t p_offset, uint32_t p_stride) {
	constexpr int FQ = 2 * SQ; // F / 64
	constexpr int HQ = 8 - SQ; // hist_len / 64
	constexpr int NQ = 8 + SQ; // (hist_len + F) / 64
	constexpr uint32_t F = FQ * 64;
	constexpr uint32_t HL = HQ * 64;

	__shared__ float2 lds_all[WAVES * LDS_F2_PER_WAVE];
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	float2 *lds = lds_all + wave * LDS_F2_PER_WAVE;

	float2 t1[8], t2[8];
#pragma unroll
	for (int k = 0; k < 8; k++) {
		t1[k] = tw[lane * 16 + k];
		t2[k] = tw[lane * 16 + 8 + k];
	}

	float accL[FQ], accR[FQ];
#pragma unroll
	for (int t = 0; t < FQ; t++) {
		accL[t] = 0.0f;
		accR[t] = 0.0f;
	}

	const uint32_t first = (blockIdx.x * WAVES + wave) * spw;
	const uint32_t last = first + spw < g.n ? first + spw : g.n;
	for (uint32_t e = first; e < last; e++) {
		const uint32_t slot = g.slots[e];
		const uint32_t row = g.rows ? g.rows[e] : e;
		const gas_params *P = st.params + slot;
		const float g0 = st.hrtf_prev_gain[slot];
		const float g1 = P->hrtf_gain;
		uint32_t dir = P->hrtf_dir;
		dir = dir < tab.dirs ? dir : 0;

		// HRIR spectra of this direction: issue early, consumed after the forward FFT.
		float4 hs[8];
#pragma unroll
		for (int j = 0; j < 8; j++) {
			hs[j] = tab.spec[((size_t)dir * 8 + j) * 64 + lane];
		}

		// x_full[lane + 64 q]: q < HQ from the history, the rest from this callback's frames.
		float xq[NQ];
		float *hist = st.hrtf_hist + (size_t)slot * HL;
#pragma unroll
		for (int q = 0; q < HQ; q++) {
			xq[q] = hist[lane + 64 * q];
		}
		const gas_audio_frame *srow = g.src + (size_t)row * F;
		uint32_t er_pos = 0;
		gas_audio_frame *ring = nullptr;
		if constexpr (WITH_ER) {
			er_pos = st.er_pos[slot];
			ring = st.er_ring + (size_t)slot * er_R;
		}
#pragma unroll
		for (int q = 0; q < FQ; q++) {
			const int f = lane + 64 * q;
			gas_audio_frame fr = srow[f];
			if constexpr (WITH_ER) {
				// early reflections (oracle fx_early_reflections): taps in order, f32
				ring[(er_pos + (uint32_t)f) & (er_R - 1)] = fr; // this block into the ring
				float yl = fr.left, yr = fr.right;
#pragma unroll
				for (int k = 0; k < GAS_ER_TAPS; k++) {
					const uint32_t du = P->er_delay[k];
					const int d = (int)(du < er_R - F ? du : er_R - F); // keeps every tap inside row/ring
					const float gk = P->er_gain[k];
					const int i = f - d;
					// i >= 0: still inside this callback's source row; else a previous callback's ring frame
					gas_audio_frame xp = i >= 0 ? srow[i] : ring[(er_pos + (uint32_t)(i + (int)er_R)) & (er_R - 1)];
					yl = yl + gk * xp.left;
					yr = yr + gk * xp.right;
				}
				fr.left = yl;
				fr.right = yr;
			}
			const float mono = (fr.left + fr.right) * 0.5f;
			const float t = (float)f * (1.0f / (float)F); // F is a power-of-two multiple: exact for 256/512
			const float gain = g1 * t + (1 - t) * g0;
			xq[HQ + q] = mono * gain;
		}
		// new history = x_full[F .. F + HL)
#pragma unroll
		for (int q = 0; q < HQ; q++) {
			hist[lane + 64 * q] = xq[FQ + q];
		}
		if (lane == 0) {
			st.hrtf_prev_gain[slot] = g1;
			if constexpr (WITH_ER) {
				st.er_pos[slot] = (er_pos + F) & (er_R - 1);
			}
		}

		// z = a + i b : a = x_full[0..512), b = x_full[S..S+512)
		float2 v[8];
#pragma unroll
		for (int j = 0; j < 8; j++) {
			v[j] = make_float2(xq[j], xq[j + SQ]);
		}
		fft512<false>(v, t1, t2, lds, lane);

		float pkl = 0.0f, pkr = 0.0f;
		{
			float2 y[8];
#pragma unroll
			for (int j = 0; j < 8; j++) {
				y[j] = cmul(v[j], make_float2(hs[j].x, hs[j].y));
			}
			fft512<true>(y, t1, t2, lds, lane);
			// valid outputs are window positions [512 - S, 512): registers j >= HQ
#pragma unroll
			for (int t = 0; t < SQ; t++) {
				const float oa = y[HQ + t].x, ob = y[HQ + t].y;
				accL[t] += oa;
				accL[SQ + t] += ob;
				pkl = fmaxf(pkl, fmaxf(fabsf(oa), fabsf(ob)));
			}
		}
		{
			float2 y[8];
#pragma unroll
			for (int j = 0; j < 8; j++) {
				y[j] = cmul(v[j], make_float2(hs[j].z, hs[j].w));
			}
			fft512<true>(y, t1, t2, lds, lane);
#pragma unroll
			for (int t = 0; t < SQ; t++) {
				const float oa = y[HQ + t].x, ob = y[HQ + t].y;
				accR[t] += oa;
				accR[SQ + t] += ob;
				pkr = fmaxf(pkr, fmaxf(fabsf(oa), fabsf(ob)));
			}
		}
		pkl = wave_max(pkl);
		pkr = wave_max(pkr);
		if (lane == 0) {
			g.peaks[(size_t)row * 2] = pkl;
			g.peaks[(size_t)row * 2 + 1] = pkr;
		}
	}

	// waves -> one partial mix per workgroup; each wave parks its sum in its own LDS slice
	wave_lds_sync();
	float *red = reinterpret_cast<float *>(lds);
#pragma unroll
	for (int t = 0; t < FQ; t++) {
		*reinterpret_cast<float2 *>(red + (lane + 64 * t) * 2) = make_float2(accL[t], accR[t]);
	}
	__syncthreads();
	float *my_partial = partials + ((size_t)p_offset + blockIdx.x) * (F * 2);
	const float *red_all = reinterpret_cast<const float *>(lds_all);
#pragma unroll
	for (int r = 0; r < (int)(F * 2) / (WAVES * 64); r++) {
		const int idx = threadIdx.x + r * WAVES * 64;
		float s = 0.0f;
#pragma unroll
		for (int w = 0; w < WAVES; w++) {
			s += red_all[w * LDS_F2_PER_WAVE * 2 + idx];
		}
		my_partial[idx] = s;
	}
	(void)p_stride;
}

// HRIR [dirs][2][taps] -> lane-major spectra table, one wave per (direction, ear), scaled by 1/512
// so the inverse transform needs no normalisation.
__global__ __launch_bounds__(64) void k_hrtf_table(const float *__restrict__ hrir, uint32_t dirs, uint32_t taps, const float2 *__restrict__ tw, float4 *__restrict__ spec) {
	__shared__ float2 lds[LDS_F2_PER_WAVE];
	const int lane = threadIdx.x;
	const uint32_t dir = blockIdx.x >> 1, ear = blockIdx.x & 1;
	if (dir >= dirs) {
		return;
	}
	float2 t1[8], t2[8];
#pragma unroll
	for (int k = 0; k < 8; k++) {
		t1[k] = tw[lane * 16 + k];
		t2[k] = tw[lane * 16 + 8 + k];
	}
	const float *h = hrir + ((size_t)dir * 2 + ear) * taps;
	float2 v[8];
#pragma unroll
	for (int j = 0; j < 8; j++) {
		const uint32_t n = lane + 64 * j;
		v[j] = make_float2(n < taps ? h[n] : 0.0f, 0.0f);
	}
	fft512<false>(v, t1, t2, lds, lane);
	float *out = reinterpret_cast<float *>(spec);
#pragma unroll
	for (int j = 0; j < 8; j++) {
		const size_t o = (((size_t)dir * 8 + j) * 64 + lane) * 4 + ear * 2;
		out[o] = v[j].x * (1.0f / 512.0f);
		out[o + 1] = v[j].y * (1.0f / 512.0f);
	}
}

// The chain [EARLY_REFLECTIONS] alone: stereo out, lane = frame, wave = source.
template <int FQ>
__global__ __launch_bounds__(WAVES * 64) void k_er_only(gas_group_args g, gas_dev_state st, uint32_t spw, uint32_t er_R, float *__restrict__ partials, uint32_t p_offset) {
	constexpr uint32_t F = FQ * 64;
	__shared__ float red_all[WAVES * F * 2];
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	float accL[FQ], accR[FQ];
#pragma unroll
	for (int t = 0; t < FQ; t++) {
		accL[t] = 0.0f;
		accR[t] = 0.0f;
	}
	const uint32_t first = (blockIdx.x * WAVES + wave) * spw;
	const uint32_t last = first + spw < g.n ? first + spw : g.n;
	for (uint32_t e = first; e < last; e++) {
		const uint32_t slot = g.slots[e];
		const uint32_t row = g.rows ? g.rows[e] : e;
		const gas_params *P = st.params + slot;
		const gas_audio_frame *srow = g.src + (size_t)row * F;
		const uint32_t er_pos = st.er_pos[slot];
		gas_audio_frame *ring = st.er_ring + (size_t)slot * er_R;
		float pkl = 0.0f, pkr = 0.0f;
#pragma unroll
		for (int q = 0; q < FQ; q++) {
			const int f = lane + 64 * q;
			gas_audio_frame fr = srow[f];
			ring[(er_pos + (uint32_t)f) & (er_R - 1)] = fr;
			float yl = fr.left, yr = fr.right;
#pragma unroll
			for (int k = 0; k < GAS_ER_TAPS; k++) {
				const uint32_t du = P->er_delay[k];
				const int d = (int)(du < er_R - F ? du : er_R - F);
				const float gk = P->er_gain[k];
				const int i = f - d;
				gas_audio_frame xp = i >= 0 ? srow[i] : ring[(er_pos + (uint32_t)(i + (int)er_R)) & (er_R - 1)];
				yl = yl + gk * xp.left;
				yr = yr + gk * xp.right;
			}
			accL[q] += yl;
			accR[q] += yr;
			pkl = fmaxf(pkl, fabsf(yl));
			pkr = fmaxf(pkr, fabsf(yr));
		}
		pkl = wave_max(pkl);
		pkr = wave_max(pkr);
		if (lane == 0) {
			st.er_pos[slot] = (er_pos + F) & (er_R - 1);
			g.peaks[(size_t)row * 2] = pkl;
			g.peaks[(size_t)row * 2 + 1] = pkr;
		}
	}
	float *red = red_all + wave * F * 2;
#pragma unroll
	for (int t = 0; t < FQ; t++) {
		*reinterpret_cast<float2 *>(red + (lane + 64 * t) * 2) = make_float2(accL[t], accR[t]);
	}
	__syncthreads();
	float *my_partial = partials + ((size_t)p_offset + blockIdx.x) * (F * 2);
#pragma unroll
	for (int r = 0; r < (int)(F * 2) / (WAVES * 64); r++) {
		const int idx = threadIdx.x + r * WAVES * 64;
		float s = 0.0f;
#pragma unroll
		for (int w = 0; w < WAVES; w++) {
			s += red_all[w * F * 2 + idx];
		}
		my_partial[idx] = s;
	}
}

} // namespace

// Twiddles per lane: [lane][0..7] = W64^(n1 k0), [lane][8..15] = W512^(n0 (k0' + 8 k1)), computed in f64.
void gas_make_twiddles(float2 *host_tw) {
	const double two_pi = 6.2831853071795864769252867666;
	for (int l = 0; l < 64; l++) {
		const int hi = l >> 3, lo = l & 7;
		for (int k = 0; k < 8; k++) {
			double a1 = -two_pi * (double)(hi * k) / 64.0;
			host_tw[l * 16 + k] = make_float2((float)cos(a1), (float)sin(a1));
			double a2 = -two_pi * (double)(lo * (hi + 8 * k)) / 512.0;
			host_tw[l * 16 + 8 + k] = make_float2((float)cos(a2), (float)sin(a2));
		}
	}
}

// Sources per wave: enough waves to fill 256 CUs a few times over, few enough partials that
// k_mix_reduce stays cheap.
uint32_t gas_hrtf_partials(uint32_t n, uint32_t *sources_per_wave) {
	uint32_t spw = n / (256u * 8u);
	if (spw < 1) {
		spw = 1;
	}
	if (spw > 16) {
		spw = 16;
	}
	if (sources_per_wave) {
		*sources_per_wave = spw;
	}
	const uint32_t per_wg = spw * WAVES;
	return (n + per_wg - 1) / per_wg;
}

hipError_t gas_launch_hrtf_ols(hipStream_t stream, bool with_er, const gas_group_args &g, const gas_dev_state &st, const gas_hrtf_table &tab, const float2 *twiddles, uint32_t frames, uint32_t hist_len, uint32_t er_ring_frames, float *partials, uint32_t p_offset, uint32_t p_stride) {
	if (g.n == 0) {
		return hipSuccess;
	}
	if (frames % 128 != 0 || frames > 512 || hist_len != 512 - frames / 2) {
		return hipErrorInvalidValue;
	}
	uint32_t spw = 1;
	const uint32_t wgs = gas_hrtf_partials(g.n, &spw);
	dim3 grid(wgs), block(WAVES * 64);
#define GAS_HRTF_CASE(SQv)                                                                                                               \
	case SQv:                                                                                                                            \
		if (with_er) {                                                                                                                   \
			hipLaunchKernelGGL((k_hrtf_ols<SQv, true>), grid, block, 0, stream, g, st, tab, twiddles, spw, er_ring_frames, partials, p_offset, p_stride);  \
		} else {                                                                                                                         \
			hipLaunchKernelGGL((k_hrtf_ols<SQv, false>), grid, block, 0, stream, g, st, tab, twiddles, spw, er_ring_frames, partials, p_offset, p_stride); \
		}                                                                                                                                \
		break;
	switch (frames / 128) {
		GAS_HRTF_CASE(1)
		GAS_HRTF_CASE(2)
		GAS_HRTF_CASE(3)
		GAS_HRTF_CASE(4)
		default:
			return hipErrorInvalidValue;
	}
#undef GAS_HRTF_CASE
	return hipGetLastError();
}

hipError_t gas_launch_er_only(hipStream_t stream, const gas_group_args &g, const gas_dev_state &st, uint32_t frames, uint32_t er_ring_frames, float *partials, uint32_t p_offset, uint32_t p_stride) {
	(void)p_stride;
	if (g.n == 0) {
		return hipSuccess;
	}
	uint32_t spw = 1;
	const uint32_t wgs = gas_hrtf_partials(g.n, &spw);
	dim3 grid(wgs), block(WAVES * 64);
	switch (frames / 64) {
		case 2:
			hipLaunchKernelGGL((k_er_only<2>), grid, block, 0, stream, g, st, spw, er_ring_frames, partials, p_offset);
			break;
		case 4:
			hipLaunchKernelGGL((k_er_only<4>), grid, block, 0, stream, g, st, spw, er_ring_frames, partials, p_offset);
			break;
		case 6:
			hipLaunchKernelGGL((k_er_only<6>), grid, block, 0, stream, g, st, spw, er_ring_frames, partials, p_offset);
			break;
		case 8:
			hipLaunchKernelGGL((k_er_only<8>), grid, block, 0, stream, g, st, spw, er_ring_frames, partials, p_offset);
			break;
		default:
			return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t gas_launch_hrtf_table(hipStream_t stream, const float *d_hrir, uint32_t dirs, uint32_t taps, const float2 *twiddles, float4 *spec) {
	hipLaunchKernelGGL(k_hrtf_table, dim3(dirs * 2), dim3(64), 0, stream, d_hrir, dirs, taps, twiddles, spec);
	return hipGetLastError();
}
