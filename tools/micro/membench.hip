// membench.hip -- standalone probe of the k_hrtf_ols memory skeleton (development aid, not product code).
// Rows of 4 KiB (512 AudioFrames), one wave per row at a time, several rows per wave.
// Build: hipcc --offload-arch=gfx950 -O3 -o membench tools/micro/membench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int F = 512;

// variant bits: 1 = 16-byte row loads (else 8-byte), 2 = history read+write (1 KiB each), 4 = spectra table reads (8 KiB/row),
// 8 = prefetch next row one ahead
template <int V>
__global__ __launch_bounds__(256) void k(const float2 *__restrict__ src, float *__restrict__ hist, const float4 *__restrict__ tab, const unsigned *__restrict__ dirs, float *__restrict__ out, int n, int spw) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int first = (blockIdx.x * 4 + wave) * spw;
	const int last = min(first + spw, n);
	float acc = 0.f;
	float2 r8[8]; float4 r16[4]; float h[4]; float4 t[8];
	auto issue = [&](int e) {
		if constexpr (V & 1) {
			const float4 *p = reinterpret_cast<const float4 *>(src + (size_t)e * F);
#pragma unroll
			for (int q = 0; q < 4; q++) r16[q] = p[lane + 64 * q];
		} else {
#pragma unroll
			for (int q = 0; q < 8; q++) r8[q] = src[(size_t)e * F + lane + 64 * q];
		}
		if constexpr (V & 2) {
#pragma unroll
			for (int q = 0; q < 4; q++) h[q] = hist[(size_t)e * 256 + lane + 64 * q];
		}
		if constexpr ((V & 4) && !(V & 16)) {
			const unsigned d = dirs[e];
#pragma unroll
			for (int j = 0; j < 8; j++) t[j] = tab[((size_t)d * 8 + j) * 64 + lane];
		}
		if constexpr ((V & 4) && (V & 16)) { // Hermitian half table: 4 KiB per direction, mirrored second half
			const unsigned d = dirs[e];
			const float4 *base = tab + (size_t)d * 256;
#pragma unroll
			for (int j = 0; j < 4; j++) t[j] = base[j * 64 + lane];
#pragma unroll
			for (int j = 4; j < 8; j++) { int p = 512 - (lane + 64 * j); p = p == 256 ? 0 : p; t[j] = base[p]; }
		}
	};
	auto consume = [&](int e) {
		float s = 0.f;
		if constexpr (V & 1) {
#pragma unroll
			for (int q = 0; q < 4; q++) s += r16[q].x + r16[q].y + r16[q].z + r16[q].w;
		} else {
#pragma unroll
			for (int q = 0; q < 8; q++) s += r8[q].x + r8[q].y;
		}
		if constexpr (V & 2) {
#pragma unroll
			for (int q = 0; q < 4; q++) { s += h[q]; hist[(size_t)e * 256 + lane + 64 * q] = s; }
		}
		if constexpr (V & 4) {
#pragma unroll
			for (int j = 0; j < 8; j++) s += t[j].x * t[j].y + t[j].z * t[j].w;
		}
		acc += s;
	};
	if constexpr (V & 8) {
		if (first < last) issue(first);
		for (int e = first; e < last; e++) {
			// consume current into temporaries, then issue next (registers reused)
			consume(e);
			if (e + 1 < last) issue(e + 1);
		}
	} else {
		for (int e = first; e < last; e++) { issue(e); consume(e); }
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int V>
float run(const std::vector<float2 *> &srcs, float *hist, float4 *tab, unsigned *dirs, float *out, int n, int spw, int iters) {
	const int wgs = (n + spw * 4 - 1) / (spw * 4);
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<V>, dim3(wgs), dim3(256), 0, 0, srcs[i % srcs.size()], hist, tab, dirs, out, n, spw);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(a));
	for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k<V>, dim3(wgs), dim3(256), 0, 0, srcs[i % srcs.size()], hist, tab, dirs, out, n, spw);
	CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
	float ms; CK(hipEventElapsedTime(&ms, a, b));
	return ms / iters * 1e3f;
}

int main(int argc, char **argv) {
	const int n = argc > 1 ? atoi(argv[1]) : 8192;
	const int nbuf = 12;
	std::vector<float2 *> srcs(nbuf);
	for (auto &p : srcs) { CK(hipMalloc(&p, (size_t)n * F * 8)); CK(hipMemset(p, 1, (size_t)n * F * 8)); }
	float *hist; CK(hipMalloc(&hist, (size_t)n * 256 * 4)); CK(hipMemset(hist, 0, (size_t)n * 256 * 4));
	float4 *tab; CK(hipMalloc(&tab, (size_t)1024 * 8 * 64 * 16)); CK(hipMemset(tab, 0, (size_t)1024 * 8 * 64 * 16));
	std::vector<unsigned> hd(n); for (int i = 0; i < n; i++) hd[i] = (unsigned)((i * 2654435761u) >> 22) & 1023;
	unsigned *dirs; CK(hipMalloc(&dirs, n * 4)); CK(hipMemcpy(dirs, hd.data(), n * 4, hipMemcpyHostToDevice));
	float *out; CK(hipMalloc(&out, (size_t)n * 256 * 4));
	const double row_mb = n * 4096.0 / 1e6;
	printf("n=%d rows (%.1f MB of rows per launch); back-to-back launches, time per launch incl. ~launch gaps\n", n, row_mb);
#define R(V, spw) { float us = run<V>(srcs, hist, tab, dirs, out, n, spw, 200); double mb = row_mb + ((V & 2) ? n * 2048.0 / 1e6 : 0); \
	printf("V=%2d spw=%2d  %7.2f us  rows+hist %.0f MB -> %.2f TB/s%s%s%s%s\n", V, spw, us, mb, mb / us / 1e6 * 1e6 / 1e6, (V & 1) ? " 16B" : " 8B", (V & 2) ? " +hist" : "", (V & 4) ? " +table" : "", (V & 8) ? " +prefetch" : ""); }
	R(0, 4) R(8, 4)
	R(2, 4) R(6, 4) R(14, 4)
	printf("-- half table, random directions\n");
	R(22, 4) R(30, 4)
	// sorted directions: consecutive sources share a direction in runs of 8
	for (int i = 0; i < n; i++) hd[i] = (unsigned)(i / 8) & 1023;
	CK(hipMemcpy(dirs, hd.data(), n * 4, hipMemcpyHostToDevice));
	printf("-- directions sorted (runs of 8)\n");
	R(14, 4) R(30, 4)
	return 0;
}
