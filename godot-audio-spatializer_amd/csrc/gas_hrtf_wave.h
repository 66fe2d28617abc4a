// gas_hrtf_wave.h -- wave-level building blocks shared by the HRTF kernels (k_hrtf_ols.hip, k_hrtf_uni.hip):
// the in-register 512-point FFT of one wave64 (three radix-8 passes, two 8x8 lane<->register transposes through the
// wave's private LDS slice), the Hermitian-half HRIR spectra fetch, the source-window loader (float rows or
// HBM-resident PCM), history rows, the deferred partial-mix sum and the DPP reductions.  NEW arithmetic: the
// reference has no HRTF/FFT; semantics are fixed by oracle/gas_oracle.c (fx_hrtf).
// Everything lives in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "gas_device.h"
#include "gas_internal.h"

namespace {

constexpr int WAVES = 8; // one workgroup = the CU's whole residency at 2 waves/SIMD -> fewest partial mixes
#ifndef GAS_HRTF_WAVES_PER_SIMD
#define GAS_HRTF_WAVES_PER_SIMD 2 // register budget the main kernel is compiled for (VGPR-limited residency)
#endif
constexpr int LDS_F2_HALF = 8 * 72; // float2 units; exchange 1 uses 8x72, exchange 2 uses 8x66
constexpr int LDS_F2_PER_WAVE = 2 * LDS_F2_HALF; // two slices so a pair of transforms can be in flight
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
// acc += z * h and y = z * h with the roundings spelled out (explicit fused multiply-adds, no compiler contraction):
// k_hrtf_uni and k_hrtf_multi must produce the same bits from the same inputs (tests/test_gpu_batched.py), which they
// only do when neither leaves the choice of what to fuse to the optimiser.
__device__ __forceinline__ void cmac_fixed(float2 &acc, float2 z, float hx, float hy) {
#pragma clang fp contract(off)
	acc.x = __builtin_fmaf(z.x, hx, __builtin_fmaf(-z.y, hy, acc.x));
	acc.y = __builtin_fmaf(z.x, hy, __builtin_fmaf(z.y, hx, acc.y));
}
__device__ __forceinline__ float2 cmul_fixed(float2 z, float hx, float hy) {
#pragma clang fp contract(off)
	return make_float2(__builtin_fmaf(z.x, hx, -(z.y * hy)), __builtin_fmaf(z.x, hy, z.y * hx));
}
// Once-touched streams (source rows, history) bypass the caches' retention so they do not evict the HRIR
// spectra table, which is the only data re-read across sources (MI355X_MICROARCH.md nt-weights row).
#ifndef GAS_ABL
#define GAS_ABL 0 // timing experiments only (results are wrong): 1 no table loads, 2 no history traffic, 4 no forward FFT, 8 no row loads, 16 no twiddle loads, 32 no frequency-domain epilogue, 64 no per-source state writes, 128 no parameter loads, 256 empty body (the launch's own floor)
#endif
#ifndef GAS_TW_LDS
#define GAS_TW_LDS 1 // twiddles: one 8 KiB copy per workgroup staged through LDS instead of 8 KiB per wave from L2 (16 MB of L2 traffic at kernel start)
#endif
#ifndef GAS_DEFER_WT
#define GAS_DEFER_WT 1 // write-through of device-published parameter rows after the source loop instead of in the prologue
#endif
#ifndef GAS_EPI_DIRECT
#define GAS_EPI_DIRECT 1 // frequency-domain epilogue: the two transforming waves sum the eight spectra themselves (one barrier and one LDS round trip less)
#endif
// Diagnostic build only (-DGAS_STAMPS): per-wave s_memrealtime stamps (100 MHz) of the last launch, read back with
// gas_debug_read_stamps().  No stamp executes in the product build.
#ifdef GAS_STAMPS
#define GAS_STAMP_SLOTS 8
__device__ unsigned long long gas_stamps[8192 * GAS_STAMP_SLOTS];
#define GAS_STAMP(i)                                                                                                      \
	do {                                                                                                                   \
		if (lane == 0 && (blockIdx.x * WAVES + wave) < 8192) {                                                             \
			gas_stamps[(blockIdx.x * WAVES + wave) * GAS_STAMP_SLOTS + (i)] = __builtin_amdgcn_s_memrealtime();           \
		}                                                                                                                  \
	} while (0)
#else
#define GAS_STAMP(i) do { } while (0)
#endif
#ifdef GAS_USE_NT // set by k_hrtf_multi.hip and (round 3) k_hrtf_uni.hip; k_hrtf_ols showed no effect in round 1 (profiles/r01_notes.md)
#define GAS_NT_LOAD(p) __builtin_nontemporal_load(p)
#define GAS_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define GAS_NT_LOAD(p) (*(p))
#define GAS_NT_STORE(v, p) (*(p) = (v))
#endif
__device__ __forceinline__ gas_audio_frame nt_load_frame(const gas_audio_frame *p) {
	typedef float v2f __attribute__((ext_vector_type(2)));
	const v2f v = GAS_NT_LOAD(reinterpret_cast<const v2f *>(p));
	return gas_audio_frame{ v.x, v.y };
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
#ifndef GAS_FFT_AR
#define GAS_FFT_AR 0 // EXPERIMENT: the eight-wave kernels' exchanges as single ds_write_b64 / ds_read_b64 (fft512_ar) instead of the ds_write2_b64 / ds_read2_b64 pairs the compiler forms (MI355X_MICROARCH.md LDS table: 8 vs 2+2 array cycles per pair of reads, 13 vs 6+6 per pair of writes)
#endif
template <bool INV>
__device__ __forceinline__ void fft512_ar(float2 (&v)[8], const float2 (&t1)[8], const float2 (&t2)[8], float2 *lds, int lane);

template <bool INV>
__device__ __forceinline__ void fft512(float2 (&v)[8], const float2 (&t1)[8], const float2 (&t2)[8], float2 *lds, int lane) {
#if GAS_FFT_AR
	fft512_ar<INV>(v, t1, t2, lds, lane);
	return;
#endif
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

// The same transform with the twiddles fetched from an LDS copy of the table at their point of use instead of
// living in 30 VGPRs across the kernel (tw[k * 64 + lane] = t1[k], tw[(8 + k) * 64 + lane] = t2[k]): 15 more
// conflict-free 8-byte LDS reads per transform, the price of a third wave per SIMD.
template <bool INV>
__device__ __forceinline__ void fft512_twlds(float2 (&v)[8], const float2 *tw, float2 *lds, int lane) {
	const int hi = lane >> 3, lo = lane & 7;
	dft8<INV>(v);
#pragma unroll
	for (int k = 1; k < 8; k++) {
		const float2 t = tw[k * 64 + lane];
		v[k] = INV ? cmulc(v[k], t) : cmul(v[k], t);
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
		const float2 t = tw[(8 + k) * 64 + lane];
		v[k] = INV ? cmulc(v[k], t) : cmul(v[k], t);
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

// ---- LDS traffic the compiler does not see ---------------------------------------------------------------------------
// While a global_load_lds (LDS-DMA) is in flight, hipcc puts `s_waitcnt vmcnt(0)` in front of every ds_read and ds_write
// it emits (SIInsertWaitcnts cannot tell which LDS bytes the DMA writes, and with a store and loads pending on the one
// vmcnt counter of gfx9 it does not count), which drains the wave's whole prefetch -- the next source's frames included
// -- in the middle of the current transform.  The twelve-wave form of k_hrtf_uni therefore issues the LDS traffic of its
// exchanges and the reads of its HRIR slots as inline assembly, each read group followed by its own lgkmcnt(0), and
// orders the one true dependency (DMA -> slot read) itself with a vmcnt wait at the top of the trip.  A wave's DS
// instructions execute in order, so reads placed after the exchange's writes see them without a wait in between.
// The empty asm statements on the loaded scalars keep LLVM from re-vectorising the butterflies that follow into
// v_pk_*_f32 (2.4x the issue cost of a scalar op on gfx950, profiles/r02_notes.md).
typedef float gas_v2f __attribute__((ext_vector_type(2)));
typedef float gas_v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t lds_byte_addr(const void *p) {
	return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

__device__ __forceinline__ float2 opaque_f2(gas_v2f r) {
	float x = r.x, y = r.y;
	asm volatile("" : "+v"(x), "+v"(y));
	return make_float2(x, y);
}

// v[k] = *(float2 *)(addr + k * STRIDE), k = 0..7
template <int STRIDE>
__device__ __forceinline__ void lds_read8_b64(uint32_t addr, float2 (&v)[8]) {
	gas_v2f r0, r1, r2, r3, r4, r5, r6, r7;
	asm volatile("ds_read_b64 %0, %8 offset:%9\n\t"
				 "ds_read_b64 %1, %8 offset:%10\n\t"
				 "ds_read_b64 %2, %8 offset:%11\n\t"
				 "ds_read_b64 %3, %8 offset:%12\n\t"
				 "ds_read_b64 %4, %8 offset:%13\n\t"
				 "ds_read_b64 %5, %8 offset:%14\n\t"
				 "ds_read_b64 %6, %8 offset:%15\n\t"
				 "ds_read_b64 %7, %8 offset:%16\n\t"
				 "s_waitcnt lgkmcnt(0)"
				 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
				 : "v"(addr), "n"(0 * STRIDE), "n"(1 * STRIDE), "n"(2 * STRIDE), "n"(3 * STRIDE), "n"(4 * STRIDE), "n"(5 * STRIDE), "n"(6 * STRIDE), "n"(7 * STRIDE)
				 : "memory");
	v[0] = opaque_f2(r0); v[1] = opaque_f2(r1); v[2] = opaque_f2(r2); v[3] = opaque_f2(r3);
	v[4] = opaque_f2(r4); v[5] = opaque_f2(r5); v[6] = opaque_f2(r6); v[7] = opaque_f2(r7);
}

// *(float2 *)(addr + k * STRIDE) = v[k], k = 0..7
template <int STRIDE>
__device__ __forceinline__ void lds_write8_b64(uint32_t addr, const float2 (&v)[8]) {
	const gas_v2f d0 = { v[0].x, v[0].y }, d1 = { v[1].x, v[1].y }, d2 = { v[2].x, v[2].y }, d3 = { v[3].x, v[3].y };
	const gas_v2f d4 = { v[4].x, v[4].y }, d5 = { v[5].x, v[5].y }, d6 = { v[6].x, v[6].y }, d7 = { v[7].x, v[7].y };
	asm volatile("ds_write_b64 %0, %1 offset:%9\n\t"
				 "ds_write_b64 %0, %2 offset:%10\n\t"
				 "ds_write_b64 %0, %3 offset:%11\n\t"
				 "ds_write_b64 %0, %4 offset:%12\n\t"
				 "ds_write_b64 %0, %5 offset:%13\n\t"
				 "ds_write_b64 %0, %6 offset:%14\n\t"
				 "ds_write_b64 %0, %7 offset:%15\n\t"
				 "ds_write_b64 %0, %8 offset:%16"
				 :
				 : "v"(addr), "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7), "n"(0 * STRIDE), "n"(1 * STRIDE), "n"(2 * STRIDE), "n"(3 * STRIDE), "n"(4 * STRIDE), "n"(5 * STRIDE), "n"(6 * STRIDE), "n"(7 * STRIDE)
				 : "memory");
}

// fft512 with its exchanges issued by lds_write8_b64 / lds_read8_b64 (same butterflies, same twiddles, same bits)
template <bool INV>
__device__ __forceinline__ void fft512_ar(float2 (&v)[8], const float2 (&t1)[8], const float2 (&t2)[8], float2 *lds, int lane) {
	const int hi = lane >> 3, lo = lane & 7;
	const uint32_t base = lds_byte_addr(lds);
	dft8<INV>(v);
#pragma unroll
	for (int k = 1; k < 8; k++) {
		v[k] = INV ? cmulc(v[k], t1[k]) : cmul(v[k], t1[k]);
	}
	lds_write8_b64<72 * 8>(base + (uint32_t)lane * 8u, v); // lds[k * 72 + lane] = v[k]
	lds_read8_b64<64>(base + (uint32_t)(hi * 72 + lo) * 8u, v); // v[k] = lds[hi * 72 + k * 8 + lo]
	dft8<INV>(v);
#pragma unroll
	for (int k = 0; k < 8; k++) {
		v[k] = INV ? cmulc(v[k], t2[k]) : cmul(v[k], t2[k]);
	}
	lds_write8_b64<64>(base + (uint32_t)(lo * 66 + hi) * 8u, v); // lds[lo * 66 + k * 8 + hi] = v[k]
	lds_read8_b64<66 * 8>(base + (uint32_t)lane * 8u, v); // v[k] = lds[k * 66 + lane]
	dft8<INV>(v);
}

// H[lane + 64 j], j = 0..7, out of a 4 KiB LDS slot holding the stored half of one HRIR row (256 float4, bins 0..255,
// Nyquist parked in DC's imaginary parts): bins < 256 as stored, bins >= 256 as conj(H[512 - k]) read backwards; lane 0
// unpacks DC and Nyquist from entry 0 (finish_spectra's result, from LDS instead of registers).
__device__ __forceinline__ void slot_read_spectra(const float4 *slot, int lane, float4 (&h)[8]) {
	const uint32_t a = lds_byte_addr(slot) + (uint32_t)lane * 16u; // entries lane + 64 j, j < 4
	const uint32_t m = lds_byte_addr(slot) + (uint32_t)(64 - lane) * 16u; // entries 64 - lane + 64 t: t = 0..3 <-> j = 7..4
	const uint32_t m4 = lane == 0 ? lds_byte_addr(slot) : m + 3072u; // j = 4: entry 256 - lane; lane 0 (entry 256 = Nyquist) reads entry 0 again
	gas_v4f r0, r1, r2, r3, r4, r5, r6, r7;
	asm volatile("ds_read_b128 %0, %8\n\t"
				 "ds_read_b128 %1, %8 offset:1024\n\t"
				 "ds_read_b128 %2, %8 offset:2048\n\t"
				 "ds_read_b128 %3, %8 offset:3072\n\t"
				 "ds_read_b128 %4, %10\n\t"
				 "ds_read_b128 %5, %9 offset:2048\n\t"
				 "ds_read_b128 %6, %9 offset:1024\n\t"
				 "ds_read_b128 %7, %9\n\t"
				 "s_waitcnt lgkmcnt(0)"
				 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
				 : "v"(a), "v"(m), "v"(m4)
				 : "memory");
	h[0] = make_float4(r0.x, r0.y, r0.z, r0.w); h[1] = make_float4(r1.x, r1.y, r1.z, r1.w); h[2] = make_float4(r2.x, r2.y, r2.z, r2.w); h[3] = make_float4(r3.x, r3.y, r3.z, r3.w);
	h[4] = make_float4(r4.x, -r4.y, r4.z, -r4.w); h[5] = make_float4(r5.x, -r5.y, r5.z, -r5.w); h[6] = make_float4(r6.x, -r6.y, r6.z, -r6.w); h[7] = make_float4(r7.x, -r7.y, r7.z, -r7.w);
	if (lane == 0) {
		h[4] = make_float4(r0.y, 0.0f, r0.w, 0.0f);
		h[0].y = 0.0f;
		h[0].w = 0.0f;
	}
}

// max over the wave of a NON-NEGATIVE value, on the VALU's DPP data path (no LDS round trips):
// row_shr 1,2,4,8 leave each 16-lane row's max in its lane 15, row_bcast15 / row_bcast31 fold the
// rows; zero fill is the identity for non-negative inputs.  Lane 63 holds the result.
// Two independent 512-point FFTs of one wave, interleaved pass by pass so the LDS exchange of one
// overlaps the butterflies of the other (lds0 / lds1 are disjoint slices).
template <bool INV>
__device__ __forceinline__ void fft512_pair(float2 (&a)[8], float2 (&b)[8], const float2 (&t1)[8], const float2 (&t2)[8], float2 *lds0, float2 *lds1, int lane) {
	const int hi = lane >> 3, lo = lane & 7;
	dft8<INV>(a);
#pragma unroll
	for (int k = 1; k < 8; k++) {
		a[k] = INV ? cmulc(a[k], t1[k]) : cmul(a[k], t1[k]);
	}
#pragma unroll
	for (int k = 0; k < 8; k++) {
		lds0[k * 72 + lane] = a[k];
	}
	dft8<INV>(b);
#pragma unroll
	for (int k = 1; k < 8; k++) {
		b[k] = INV ? cmulc(b[k], t1[k]) : cmul(b[k], t1[k]);
	}
#pragma unroll
	for (int k = 0; k < 8; k++) {
		lds1[k * 72 + lane] = b[k];
	}
	wave_lds_sync();
#pragma unroll
	for (int k = 0; k < 8; k++) {
		a[k] = lds0[hi * 72 + k * 8 + lo];
	}
#pragma unroll
	for (int k = 0; k < 8; k++) {
		b[k] = lds1[hi * 72 + k * 8 + lo];
	}
	wave_lds_sync();
	dft8<INV>(a);
#pragma unroll
	for (int k = 0; k < 8; k++) {
		a[k] = INV ? cmulc(a[k], t2[k]) : cmul(a[k], t2[k]);
	}
#pragma unroll
	for (int k = 0; k < 8; k++) {
		lds0[lo * 66 + k * 8 + hi] = a[k];
	}
	dft8<INV>(b);
#pragma unroll
	for (int k = 0; k < 8; k++) {
		b[k] = INV ? cmulc(b[k], t2[k]) : cmul(b[k], t2[k]);
	}
#pragma unroll
	for (int k = 0; k < 8; k++) {
		lds1[lo * 66 + k * 8 + hi] = b[k];
	}
	wave_lds_sync();
#pragma unroll
	for (int k = 0; k < 8; k++) {
		a[k] = lds0[k * 66 + lane];
	}
#pragma unroll
	for (int k = 0; k < 8; k++) {
		b[k] = lds1[k * 66 + lane];
	}
	wave_lds_sync();
	dft8<INV>(a);
	dft8<INV>(b);
}

__device__ __forceinline__ float wave_max(float v) {
	int x = __float_as_int(v);
#define GAS_DPP_MAX(ctrl, row_mask)                                                                    \
	x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(0, x, ctrl, row_mask, 0xF, false))))
	GAS_DPP_MAX(0x111, 0xF); // row_shr:1
	GAS_DPP_MAX(0x112, 0xF); // row_shr:2
	GAS_DPP_MAX(0x114, 0xF); // row_shr:4
	GAS_DPP_MAX(0x118, 0xF); // row_shr:8
	GAS_DPP_MAX(0x142, 0xA); // row_bcast:15 -> rows 1,3
	GAS_DPP_MAX(0x143, 0xC); // row_bcast:31 -> rows 2,3
#undef GAS_DPP_MAX
	return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}

// Wave-uniform description of one source of the callback (scalar registers).
struct SrcMeta {
	uint32_t slot, row, dir;
	uint32_t pdir; // direction of the previous callback (== dir when there was none): cross-fade source (8f#4)
	float g0, g1;
	// device-resident stream cursor (SRC_PCM only)
	const void *pcm;
	uint64_t len, pos, start;
	uint32_t fc, hf, mixed;
};

// Metadata of a wave's sources lives one-source-per-lane in VGPRs: the dependent loads
// (slot list -> parameter table -> direction) are paid once per wave, for all of its sources in
// parallel, instead of once per source on the critical path (dependent scalar loads share lgkmcnt
// with the FFT's LDS exchanges and cost ~16 us per launch when done per source).
struct LaneMeta {
	uint32_t slot, row, dir, pdir;
	uint32_t prow; // row of the peaks array (k_hrtf_uni; == row unless gas_group_args::peak_rows says otherwise)
	float g0, g1;
	gas_cursor cur;
};

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int i) {
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, i);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), i);
	return ((uint64_t)hi << 32) | lo;
}

template <bool SRC_PCM>
__device__ __forceinline__ SrcMeta bcast_meta(const LaneMeta &lm, uint32_t i, uint32_t F) {
	SrcMeta m{};
	m.slot = (uint32_t)__builtin_amdgcn_readlane((int)lm.slot, (int)i);
	m.row = (uint32_t)__builtin_amdgcn_readlane((int)lm.row, (int)i);
	m.dir = (uint32_t)__builtin_amdgcn_readlane((int)lm.dir, (int)i);
	m.pdir = (uint32_t)__builtin_amdgcn_readlane((int)lm.pdir, (int)i);
	m.g0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lm.g0), (int)i));
	m.g1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lm.g1), (int)i));
	if constexpr (SRC_PCM) {
		m.pcm = reinterpret_cast<const void *>(readlane64(reinterpret_cast<uint64_t>(lm.cur.pcm), (int)i));
		m.len = readlane64(lm.cur.frames, (int)i);
		m.pos = readlane64(lm.cur.pos, (int)i);
		m.start = readlane64(lm.cur.start, (int)i);
		m.fc = (uint32_t)__builtin_amdgcn_readlane((int)lm.cur.format_channels, (int)i);
		m.hf = (uint32_t)__builtin_amdgcn_readlane((int)lm.cur.has_frames, (int)i);
		m.hf = (m.hf && m.pcm) ? 1u : 0u;
		const uint64_t left = m.len > m.pos ? m.len - m.pos : 0;
		m.mixed = m.hf ? (left < F ? (uint32_t)left : F) : 0; // [ENGINE] AudioStreamPlayback::mix return value
	}
	return m;
}

// The F frames of the source window the DSP sees (audio_spatializer.cpp:367-408), lane l taking frames l + 64 q.
// Float rows: the caller's [n][F] buffer.  SRC_PCM: sampled here from the HBM-resident stream -- k_sample_sources'
// logic, fused: 64-frame lookahead delay, silence in front of the playback's start, fade-out over the last 64
// valid frames, zero feed afterwards.  The format switch sits outside the frame loop and every load is
// unconditional (out-of-window lanes read index 0 and are masked afterwards), so the loads issue back to back.
template <bool SRC_PCM, int FQ>
__device__ __forceinline__ void load_window(const gas_group_args &g, const SrcMeta &m, int lane, const float *__restrict__ fade_env, gas_audio_frame (&raw)[FQ]) {
	constexpr uint32_t F = FQ * 64;
	if constexpr (!SRC_PCM) {
#pragma unroll
		for (int q = 0; q < FQ; q++) {
			raw[q] = (GAS_ABL & 8) ? gas_audio_frame{ (float)lane, (float)m.row } : nt_load_frame(&g.src[(size_t)m.row * F + lane + 64 * q]);
		}
	} else {
		const void *p = m.hf ? m.pcm : static_cast<const void *>(fade_env); // any readable address when nothing plays
		const int64_t base = (int64_t)m.pos - GAS_LOOKAHEAD_BUFFER_SIZE;
		bool ok[FQ];
		int64_t idx[FQ];
#pragma unroll
		for (int q = 0; q < FQ; q++) {
			const uint32_t f = (uint32_t)(lane + 64 * q);
			const int64_t si = base + f;
			ok[q] = m.hf && (m.mixed == F || f < m.mixed + GAS_LOOKAHEAD_BUFFER_SIZE) && si >= (int64_t)m.start;
			idx[q] = ok[q] ? si : 0;
		}
		const uint32_t fmt = m.fc >> 8, ch = m.fc & 0xff;
		if (fmt == GAS_PCM_S16 && ch == 1) {
			short x[FQ];
#pragma unroll
			for (int q = 0; q < FQ; q++) {
				x[q] = static_cast<const short *>(p)[idx[q]];
			}
#pragma unroll
			for (int q = 0; q < FQ; q++) {
				const float v = (float)x[q] / 32768.0f;
				raw[q] = gas_audio_frame{ v, v };
			}
		} else if (fmt == GAS_PCM_S16) {
			short2 x[FQ];
#pragma unroll
			for (int q = 0; q < FQ; q++) {
				x[q] = static_cast<const short2 *>(p)[idx[q]];
			}
#pragma unroll
			for (int q = 0; q < FQ; q++) {
				raw[q] = gas_audio_frame{ (float)x[q].x / 32768.0f, (float)x[q].y / 32768.0f };
			}
		} else if (ch == 1) {
#pragma unroll
			for (int q = 0; q < FQ; q++) {
				const float v = static_cast<const float *>(p)[idx[q]];
				raw[q] = gas_audio_frame{ v, v };
			}
		} else {
#pragma unroll
			for (int q = 0; q < FQ; q++) {
				const float2 v = static_cast<const float2 *>(p)[idx[q]];
				raw[q] = gas_audio_frame{ v.x, v.y };
			}
		}
		const bool ending = m.hf && m.mixed != F; // :380-396, once per playback
#pragma unroll
		for (int q = 0; q < FQ; q++) {
			const uint32_t f = (uint32_t)(lane + 64 * q);
			float e = ok[q] ? 1.0f : 0.0f;
			if (ending && f >= m.mixed) {
				e *= fade_env[(f - m.mixed) & (GAS_LOOKAHEAD_BUFFER_SIZE - 1)];
			}
			raw[q].left *= e;
			raw[q].right *= e;
		}
	}
}

// HRIR spectra table: the HRIRs are real, so H[512-k] = conj(H[k]); only bins 0..255 are stored per
// direction (4 KiB: float4 = HL.re, HL.im, HR.re, HR.im), with the real Nyquist bin H[256] parked in the
// (zero) imaginary slots of DC.  Lane l needs bins l + 64 j: j < 4 come straight from the table, j >= 4
// are fetched as bin 512 - k -- the same 4 KiB, mirrored addressing -- and conjugated.  This halves the
// table footprint (4 MiB at 1024 directions = one XCD L2) and its L2 -> CU traffic per source.
__device__ __forceinline__ void issue_spectra(const float4 *__restrict__ spec, uint32_t dir, int lane, float4 (&hs)[8]) {
	if (GAS_ABL & 1) {
#pragma unroll
		for (int j = 0; j < 8; j++) {
			hs[j] = make_float4(1.0f, 0.5f, 0.25f, (float)dir);
		}
		return;
	}
	const float4 *base = spec + (size_t)dir * 256;
#pragma unroll
	for (int j = 0; j < 4; j++) {
		hs[j] = base[j * 64 + lane];
	}
#pragma unroll
	for (int j = 4; j < 8; j++) {
		int p = 512 - (lane + 64 * j); // 1..256
		p = p == 256 ? 0 : p; // lane 0, j = 4: the Nyquist bin lives in DC's imaginary slots
		hs[j] = base[p];
	}
}

// Turns the registers filled by issue_spectra into H[lane + 64 j] for every j.
__device__ __forceinline__ void finish_spectra(int lane, float4 (&hs)[8]) {
#pragma unroll
	for (int j = 4; j < 8; j++) {
		hs[j].y = -hs[j].y;
		hs[j].w = -hs[j].w;
	}
	if (lane == 0) {
		hs[4] = make_float4(hs[0].y, 0.0f, hs[0].w, 0.0f);
		hs[0].y = 0.0f;
		hs[0].w = 0.0f;
	}
}

// The previous callback's k_mix_reduce, done by one wave of this workgroup for float4 column `col` (one column per
// job wave; the context only hands a job over when the grid covers every column and p_count <= 256).  The four
// partial rows a lane sums are loaded at kernel start and parked in registers, so the sum itself -- gas_device.h's
// column sum, the very code k_mix_reduce runs, hence the same bits -- finds them landed and hides behind the other
// waves' epilogue.
constexpr int JOB_ROWS = 4; // rows lane, lane + 64, lane + 128, lane + 192
__device__ __forceinline__ void job_issue(const gas_deferred_reduce &j, uint32_t col, int lane, float4 (&jr)[JOB_ROWS]) {
	const uint32_t e4 = j.elems / 4;
	const float4 *p = reinterpret_cast<const float4 *>(j.partials) + col;
#pragma unroll
	for (int r = 0; r < JOB_ROWS; r++) {
		const uint32_t k = (uint32_t)lane + 64u * r;
		const float4 v = p[(size_t)(k < j.p_count ? k : 0) * e4];
		const float keep = k < j.p_count ? 1.0f : 0.0f; // missing rows contribute +0: x + 0 = x changes no bit
		jr[r] = make_float4(keep != 0.0f ? v.x : 0.0f, keep != 0.0f ? v.y : 0.0f, keep != 0.0f ? v.z : 0.0f, keep != 0.0f ? v.w : 0.0f);
	}
}

__device__ __forceinline__ void job_finish(const gas_deferred_reduce &j, uint32_t col, int lane, const float4 (&jr)[JOB_ROWS]) {
	float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
	for (int r = 0; r < JOB_ROWS; r++) {
		gas_mix_column_add(s, jr[r]);
	}
	const float4 t = gas_mix_column_fold(s);
	if (lane == 0) {
		reinterpret_cast<float4 *>(j.out)[col] = t;
	}
}

// Sources [first, last) of wave `gw` out of `n_waves`: an even split (the first n % n_waves waves take one more), so
// every planned workgroup has work.
__device__ __forceinline__ void wave_range(uint32_t n, uint32_t gw, uint32_t n_waves, uint32_t &first, uint32_t &last) {
	const uint32_t base = n / n_waves, rem = n % n_waves;
	first = gw * base + (gw < rem ? gw : rem);
	last = first + base + (gw < rem ? 1u : 0u);
}

// History rows are stored lane-major -- element lane * HQ + q holds x[lane + 64 q] -- so a lane's HQ samples are one
// contiguous 4*HQ-byte piece (one 16-byte access at F = 512 instead of four 4-byte ones: the history cost 3.1 us of
// the 16.7 us kernel as 4-byte accesses).  The layout is private to this file; k_zero_slot only writes zeros.
// `nt` (wave-uniform): non-temporal accesses for rows in HBM that nothing re-reads before the caches have turned over --
// callbacks whose history rows (1 KiB per source, read once and written once per callback) exceed what the Infinity
// Cache keeps from one callback to the next.  Measured on the synchronous kernel (profiles/r03_notes.md section 3): with
// the rows non-temporal 1 M sources take 1264 instead of 1350 us, 65 536 sources 82.6 instead of 80.8 us (there the row
// written by callback t is still cached when callback t + 1 asks for it), so the caller decides by size.
template <int HQ>
__device__ __forceinline__ void load_history(const float *__restrict__ row, int lane, float (&h)[HQ], bool nt = false) {
	if constexpr (HQ % 4 == 0) {
		if (nt) {
#pragma unroll
			for (int q = 0; q < HQ; q += 4) {
				const gas_v4f v = __builtin_nontemporal_load(reinterpret_cast<const gas_v4f *>(row + lane * HQ + q));
				h[q] = v.x; h[q + 1] = v.y; h[q + 2] = v.z; h[q + 3] = v.w;
			}
			return;
		}
#pragma unroll
		for (int q = 0; q < HQ; q += 4) {
			const float4 v = *reinterpret_cast<const float4 *>(row + lane * HQ + q);
			h[q] = v.x; h[q + 1] = v.y; h[q + 2] = v.z; h[q + 3] = v.w;
		}
	} else if constexpr (HQ % 2 == 0) {
#pragma unroll
		for (int q = 0; q < HQ; q += 2) {
			const float2 v = *reinterpret_cast<const float2 *>(row + lane * HQ + q);
			h[q] = v.x; h[q + 1] = v.y;
		}
	} else {
#pragma unroll
		for (int q = 0; q < HQ; q++) {
			h[q] = row[lane * HQ + q];
		}
	}
}

template <int HQ>
#ifndef GAS_HIST_WT
#define GAS_HIST_WT 0 // EXPERIMENT (off): history rows stored write-through (sc1), so that the bytes leave the L2 during the kernel instead of in the flush at its end (MI355X_MICROARCH.md boundary row: + B / 6 TB/s for B dirty bytes).  Measured, synchronous callback: 8192 sources kernel 14.1 -> 15.2 us, callback 19.2 -> 19.8 (the boundary gets ~0.5 us cheaper, the kernel 1.1 us dearer); 65 536 sources callback 86.7 -> 85.9 us (-1 %)
#endif
__device__ __forceinline__ void store_history(float *__restrict__ row, int lane, const float *h, bool nt = false) {
	if constexpr (HQ % 4 == 0) {
#if GAS_HIST_WT
		if (!nt) {
#pragma unroll
			for (int q = 0; q < HQ; q += 4) {
				const gas_v4f v = { h[q], h[q + 1], h[q + 2], h[q + 3] };
				float *p = row + lane * HQ + q;
				asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
			}
			return;
		}
#endif
		if (nt) {
#pragma unroll
			for (int q = 0; q < HQ; q += 4) {
				const gas_v4f v = { h[q], h[q + 1], h[q + 2], h[q + 3] };
				__builtin_nontemporal_store(v, reinterpret_cast<gas_v4f *>(row + lane * HQ + q));
			}
			return;
		}
#pragma unroll
		for (int q = 0; q < HQ; q += 4) {
			*reinterpret_cast<float4 *>(row + lane * HQ + q) = make_float4(h[q], h[q + 1], h[q + 2], h[q + 3]);
		}
	} else if constexpr (HQ % 2 == 0) {
#pragma unroll
		for (int q = 0; q < HQ; q += 2) {
			*reinterpret_cast<float2 *>(row + lane * HQ + q) = make_float2(h[q], h[q + 1]);
		}
	} else {
#pragma unroll
		for (int q = 0; q < HQ; q++) {
			row[lane * HQ + q] = h[q];
		}
	}
}

} // namespace
