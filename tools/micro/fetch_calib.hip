// fetch_calib.hip -- known-byte streaming reads with 4-, 8- and 16-byte-per-lane loads, to calibrate rocprofv3's
// FETCH_SIZE on gfx950 for each width (MI355X_MICROARCH.md: 16 B/lane reads are reported at exactly 1/2).
// Run under: rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename T>
__global__ __launch_bounds__(256) void k_read(const T *__restrict__ p, size_t n, float *out) {
	float acc = 0.f;
	for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		T v = p[i];
		acc += reinterpret_cast<const float *>(&v)[0];
	}
	if (acc == 123.456f) out[0] = acc;
}
// row-structured 8-byte pattern of k_hrtf_ols: one wave per 4 KiB row, lane l reads float2 at l + 64 q
__global__ __launch_bounds__(256) void k_rows8(const float2 *__restrict__ p, int rows, float *out) {
	const int lane = threadIdx.x & 63;
	const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
	float acc = 0.f;
	if (row < rows) {
#pragma unroll
		for (int q = 0; q < 8; q++) acc += p[(size_t)row * 512 + lane + 64 * q].x;
	}
	if (acc == 123.456f) out[0] = acc;
}
int main() {
	const size_t bytes = 512ull << 20; // 512 MiB > Infinity Cache
	void *buf; float *out;
	CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes)); CK(hipMalloc(&out, 4));
	for (int r = 0; r < 3; r++) {
		hipLaunchKernelGGL(k_read<float>, dim3(4096), dim3(256), 0, 0, (const float *)buf, bytes / 4, out);
		hipLaunchKernelGGL(k_read<float2>, dim3(4096), dim3(256), 0, 0, (const float2 *)buf, bytes / 8, out);
		hipLaunchKernelGGL(k_read<float4>, dim3(4096), dim3(256), 0, 0, (const float4 *)buf, bytes / 16, out);
		hipLaunchKernelGGL(k_rows8, dim3((int)(bytes / 4096 / 4)), dim3(256), 0, 0, (const float2 *)buf, (int)(bytes / 4096), out);
	}
	CK(hipDeviceSynchronize());
	printf("each kernel reads %zu KiB\n", bytes >> 10);
	return 0;
}
