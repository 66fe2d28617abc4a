// gas_fft_pk.h -- the wave-level 512-point FFT of gas_hrtf_wave.h on gfx950's packed-f32 VALU path.
//
// A complex value is one aligned VGPR pair (re, im).  v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 work on both
// halves of such a pair in one issue slot, and their source modifiers (op_sel / op_sel_hi pick the half feeding each
// result half, neg_lo / neg_hi negate it) make "times +-i" and "times (re, im) swapped" free: a complex add or
// subtract is ONE instruction including a rotation of its second operand, a complex multiply is TWO (scalar code:
// 2 and 4).  The compiler finds the plain adds by itself but materialises most swaps and negations as v_mov /
// v_xor (build/asm experiments, profiles/r02_notes.md), so the modifier forms are spelled out as one-instruction
// inline asm; everything else stays ordinary vector code the scheduler can move freely.
// The rounding of every operation equals the scalar formulation's except where a multiply by 1/sqrt2 is fused into
// the following add (one rounding less); parity is against the oracle within its 1e-5 bound either way.
#pragma once
#include <hip/hip_runtime.h>

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f to_v2f(float2 a) {
	return v2f{ a.x, a.y };
}
__device__ __forceinline__ float2 to_f2(v2f a) {
	return make_float2(a.x, a.y);
}

// a - i d = (a.x + d.y, a.y - d.x)
__device__ __forceinline__ v2f pk_a_minus_id(v2f a, v2f d) {
	v2f r;
	asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(d));
	return r;
}
// a + i d = (a.x - d.y, a.y + d.x)
__device__ __forceinline__ v2f pk_a_plus_id(v2f a, v2f d) {
	v2f r;
	asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(d));
	return r;
}
// forward transforms rotate by -i, inverse ones by +i
template <bool INV>
__device__ __forceinline__ v2f pk_add_rot(v2f a, v2f d) {
	return INV ? pk_a_plus_id(a, d) : pk_a_minus_id(a, d);
}
template <bool INV>
__device__ __forceinline__ v2f pk_sub_rot(v2f a, v2f d) {
	return INV ? pk_a_minus_id(a, d) : pk_a_plus_id(a, d);
}

// a * t
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f t) {
	v2f r;
	asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(t)); // (a.x t.x, a.x t.y)
	asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(r) : "v"(a), "v"(t)); // + (-a.y t.y, a.y t.x)
	return r;
}
// a * conj(t)
__device__ __forceinline__ v2f pk_cmulc(v2f a, v2f t) {
	v2f r;
	asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(t)); // (a.x t.x, -a.x t.y)
	asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "+v"(r) : "v"(a), "v"(t)); // + (a.y t.y, a.y t.x)
	return r;
}
// acc + a * t
__device__ __forceinline__ v2f pk_cmac(v2f acc, v2f a, v2f t) {
	asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(a), "v"(t));
	asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(acc) : "v"(a), "v"(t));
	return acc;
}
// b + s * u, s a wave-uniform real
__device__ __forceinline__ v2f pk_axpy(float s, v2f u, v2f b) {
	return __builtin_elementwise_fma(u, v2f{ s, s }, b);
}

constexpr float PK_S2 = 0.70710678118654752440f;

// 8-point DFT in registers: 24 packed adds + 4 packed fmas (the scalar form: 48 adds + 8 multiplies).
template <bool INV>
__device__ __forceinline__ void pk_dft8(v2f (&v)[8]) {
	const v2f a0 = v[0] + v[4], a1 = v[0] - v[4];
	const v2f a2 = v[2] + v[6], d3 = v[2] - v[6]; // a3 = rot(d3)
	const v2f a4 = v[1] + v[5], a5 = v[1] - v[5];
	const v2f a6 = v[3] + v[7], d7 = v[3] - v[7]; // a7 = rot(d7)
	const v2f b0 = a0 + a2, b2 = a0 - a2;
	const v2f b1 = pk_add_rot<INV>(a1, d3), b3 = pk_sub_rot<INV>(a1, d3);
	const v2f b4 = a4 + a6, d6 = a4 - a6; // b6 = rot(d6)
	const v2f b5 = pk_add_rot<INV>(a5, d7), b7 = pk_sub_rot<INV>(a5, d7);
	// W8^1 b5 = S2 (b5 - i b5) forward, S2 (b5 + i b5) inverse;  W8^3 b7 = -S2 (b7 + i b7) forward, -S2 (b7 - i b7) inverse
	const v2f u5 = pk_add_rot<INV>(b5, b5), u7 = pk_sub_rot<INV>(b7, b7);
	v[0] = b0 + b4;
	v[4] = b0 - b4;
	v[2] = pk_add_rot<INV>(b2, d6);
	v[6] = pk_sub_rot<INV>(b2, d6);
	v[1] = pk_axpy(PK_S2, u5, b1);
	v[5] = pk_axpy(-PK_S2, u5, b1);
	v[3] = pk_axpy(-PK_S2, u7, b3);
	v[7] = pk_axpy(PK_S2, u7, b3);
}

__device__ __forceinline__ void pk_wave_lds_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 512-point FFT of one wave, same data layout and twiddle tables as fft512 (gas_hrtf_wave.h).
template <bool INV>
__device__ __forceinline__ void pk_fft512(v2f (&v)[8], const v2f (&t1)[8], const v2f (&t2)[8], v2f *lds, int lane) {
	const int hi = lane >> 3, lo = lane & 7;
	pk_dft8<INV>(v);
#pragma unroll
	for (int k = 1; k < 8; k++) {
		v[k] = INV ? pk_cmulc(v[k], t1[k]) : pk_cmul(v[k], t1[k]);
	}
#pragma unroll
	for (int k = 0; k < 8; k++) {
		lds[k * 72 + lane] = v[k];
	}
	pk_wave_lds_sync();
#pragma unroll
	for (int k = 0; k < 8; k++) {
		v[k] = lds[hi * 72 + k * 8 + lo];
	}
	pk_wave_lds_sync();
	pk_dft8<INV>(v);
#pragma unroll
	for (int k = 0; k < 8; k++) {
		v[k] = INV ? pk_cmulc(v[k], t2[k]) : pk_cmul(v[k], t2[k]);
	}
#pragma unroll
	for (int k = 0; k < 8; k++) {
		lds[lo * 66 + k * 8 + hi] = v[k];
	}
	pk_wave_lds_sync();
#pragma unroll
	for (int k = 0; k < 8; k++) {
		v[k] = lds[k * 66 + lane];
	}
	pk_wave_lds_sync();
	pk_dft8<INV>(v);
}

} // namespace
