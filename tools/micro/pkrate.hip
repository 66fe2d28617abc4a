// pkrate.hip -- dev aid: issue cost of packed-f32 VALU instructions on gfx950 against their scalar forms.
// Each wave runs N iterations of 16 independent instructions of one kind; reports cycles per instruction per wave
// (s_memtime) with 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int N) {
	v2f a[16], b = { 1.0001f, 0.9999f }, c = { 1e-7f, -1e-7f };
	for (int i = 0; i < 16; i++) {
		a[i] = v2f{ (float)threadIdx.x + i, (float)i };
	}
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int n = 0; n < N; n++) {
#define OP_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define OP_PKADDSEL(i) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(c));
#define OP_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_PKFMASEL(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
#define OP_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
#define OP_ADD2(i) asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3" : "+v"(a[i].x), "+v"(a[i].y) : "v"(c.x), "v"(c.y));
		if constexpr (KIND == 0) { REP16(OP_PKADD) }
		if constexpr (KIND == 1) { REP16(OP_PKADDSEL) }
		if constexpr (KIND == 2) { REP16(OP_PKMUL) }
		if constexpr (KIND == 3) { REP16(OP_PKFMA) }
		if constexpr (KIND == 4) { REP16(OP_PKFMASEL) }
		if constexpr (KIND == 5) { REP16(OP_ADD) }
		if constexpr (KIND == 6) { REP16(OP_FMA) }
		if constexpr (KIND == 7) { REP16(OP_ADD2) }
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	float s = 0;
	for (int i = 0; i < 16; i++) {
		s += a[i].x + a[i].y;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		*cyc = t1 - t0;
	}
}

int main() {
	float *out; unsigned long long *cyc, h;
	hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
	const char *names[] = { "v_pk_add_f32", "v_pk_add_f32 op_sel/neg", "v_pk_mul_f32", "v_pk_fma_f32", "v_pk_fma_f32 op_sel/neg", "v_add_f32", "v_fma_f32", "2 x v_add_f32 (one complex add)" };
	const int N = 32768;
	for (int w = 0; w < 40; w++) hipLaunchKernelGGL(k<6>, dim3(256), dim3(512), 0, 0, out, cyc, N); // clock warm-up
	hipDeviceSynchronize();
	for (int threads = 256; threads <= 512; threads += 256) {
		printf("%d wave(s) per SIMD: s_memtime ticks (100 MHz) per instruction-slot per wave; multiply by clock/100MHz for cycles\n", threads / 256);
		for (int kind = 0; kind < 8; kind++) {
			hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
			float best = 1e9;
			for (int it = 0; it < 5; it++) {
				hipEventRecord(e0);
				switch (kind) {
#define C(K) case K: hipLaunchKernelGGL(k<K>, dim3(256), dim3(threads), 0, 0, out, cyc, N); break;
					C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7)
				}
				hipEventRecord(e1); hipEventSynchronize(e1);
				float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
			}
			hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
			printf("  %-34s kernel %.3f ms -> %.2f ns per slot per wave (%.2f cycles at 2.4 GHz)\n", names[kind], best, best * 1e6 / (N * 16.0), best * 1e6 / (N * 16.0) * 2.4);
		}
	}
	return 0;
}
