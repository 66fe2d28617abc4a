// valurate.hip -- dev aid: what one SIMD of gfx950 sustains for the instruction kinds an LDS-free FFT exchange would be
// built from, against plain f32 VALU work, at 1 / 2 / 4 waves per SIMD.  Every wave runs N iterations of 16 independent
// instructions of one kind; the in-kernel clock (s_memtime, shader cycles) of one wave gives cycles per instruction per
// wave, hence per SIMD = that / waves-per-SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, int N) {
	float a[16], b = 1.0001f, c = 1e-7f;
	for (int i = 0; i < 16; i++) {
		a[i] = (float)threadIdx.x + i;
	}
	__syncthreads();
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int n = 0; n < N; n++) {
#define OP_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define OP_MOVDPP(i) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
#define OP_CNDDPP(i) asm volatile("v_cndmask_b32_dpp %0, %1, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 15]) : "vcc");
#define OP_SWAP32(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 8) & 15]));
#define OP_SWAP16(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 8) & 15]));
		if constexpr (KIND == 0) { REP16(OP_FMA) }
		if constexpr (KIND == 1) { REP16(OP_ADD) }
		if constexpr (KIND == 2) { REP16(OP_MOVDPP) }
		if constexpr (KIND == 3) { REP16(OP_CNDDPP) }
		if constexpr (KIND == 4) { OP_SWAP32(0) OP_SWAP32(1) OP_SWAP32(2) OP_SWAP32(3) OP_SWAP32(4) OP_SWAP32(5) OP_SWAP32(6) OP_SWAP32(7) OP_SWAP32(0) OP_SWAP32(1) OP_SWAP32(2) OP_SWAP32(3) OP_SWAP32(4) OP_SWAP32(5) OP_SWAP32(6) OP_SWAP32(7) }
		if constexpr (KIND == 5) { OP_SWAP16(0) OP_SWAP16(1) OP_SWAP16(2) OP_SWAP16(3) OP_SWAP16(4) OP_SWAP16(5) OP_SWAP16(6) OP_SWAP16(7) OP_SWAP16(0) OP_SWAP16(1) OP_SWAP16(2) OP_SWAP16(3) OP_SWAP16(4) OP_SWAP16(5) OP_SWAP16(6) OP_SWAP16(7) }
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	float s = 0;
	for (int i = 0; i < 16; i++) {
		s += a[i];
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		*cyc = t1 - t0;
	}
}

int main() {
	float *out;
	unsigned long long *cyc, h;
	hipMalloc(&out, 256 * 1024 * 4);
	hipMalloc(&cyc, 8);
	const char *names[] = { "v_fma_f32", "v_add_f32", "v_mov_b32_dpp row_ror:8 bank_mask", "v_cndmask_b32_dpp quad_perm", "v_permlane32_swap_b32", "v_permlane16_swap_b32" };
	const int N = 16384;
	for (int w = 0; w < 40; w++) {
		hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, cyc, N); // clock warm-up
	}
	hipDeviceSynchronize();
	for (int threads = 256; threads <= 1024; threads *= 2) {
		printf("%d wave(s) per SIMD\n", threads / 256);
		for (int kind = 0; kind < 6; kind++) {
			hipEvent_t e0, e1;
			hipEventCreate(&e0);
			hipEventCreate(&e1);
			float best = 1e9;
			for (int it = 0; it < 5; it++) {
				hipEventRecord(e0);
				switch (kind) {
#define C(K) case K: hipLaunchKernelGGL(k<K>, dim3(256), dim3(threads), 0, 0, out, cyc, N); break;
					C(0) C(1) C(2) C(3) C(4) C(5)
				}
				hipEventRecord(e1);
				hipEventSynchronize(e1);
				float ms;
				hipEventElapsedTime(&ms, e0, e1);
				best = ms < best ? ms : best;
			}
			hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
			const double per_wave_cyc = (double)h / (N * 16.0);
			printf("  %-38s kernel %.3f ms; in-kernel: %.2f cycles per instruction per wave -> %.2f per SIMD; wall %.2f ns per instruction per wave (clock ~%.2f GHz)\n", names[kind], best, per_wave_cyc, per_wave_cyc / (threads / 256), best * 1e6 / (N * 16.0), per_wave_cyc / (best * 1e6 / (N * 16.0)));
		}
	}
	return 0;
}
