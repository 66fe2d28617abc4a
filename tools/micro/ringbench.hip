// ringbench.hip -- development probe (not product code): the loader/consumer LDS-DMA ring that k_hrtf_uni's source
// streaming is built on, as a bare memory skeleton.  Per workgroup: NL loader waves pull 4 KiB source rows + 1 KiB
// history rows into an LDS ring with global_load_lds_dwordx4 (no VGPR landing zone, many sources in flight), NC
// consumer waves take them out (sum + 1 KiB history store).  Checks the data path (every byte summed against a host
// reference) and times the launch like the library does (event pair around one launch, back to back).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/ringbench tools/micro/ringbench.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                         \
	do {                                                              \
		hipError_t e_ = (x);                                          \
		if (e_ != hipSuccess) {                                       \
			printf("%s: %s\n", #x, hipGetErrorString(e_));            \
			exit(1);                                                  \
		}                                                             \
	} while (0)

constexpr int ROW_BYTES = 4096, HIST_BYTES = 1024, SLOT_BYTES = ROW_BYTES + HIST_BYTES;
constexpr int P = 5; // DMA instructions per source (4 row pieces + 1 history piece)

__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
	unsigned keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// Flags live in LDS and are touched with explicit DS instructions: a volatile access through a generic pointer
// becomes flat_load + s_waitcnt vmcnt(0), which would drain every DMA in flight.
__device__ __forceinline__ uint32_t lds_read_u32(uint32_t addr) {
	uint32_t v;
	asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
	return v;
}
__device__ __forceinline__ void lds_write_u32(uint32_t addr, uint32_t v) {
	asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

__device__ __forceinline__ void wait_vm(int n) { // s_waitcnt vmcnt(n), n <= 60
	switch (n) {
#define W(k)                                                 \
	case k:                                                  \
		asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); \
		break;
		W(0) W(5) W(10) W(15) W(20) W(25) W(30) W(35) W(40) W(45) W(50) W(55) W(60)
#undef W
		default:
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	}
}

template <int NC, int NL, int R, int MAXOUT>
__global__ __launch_bounds__((NC + NL) * 64, 2) void k_ring(const char *__restrict__ rows, char *__restrict__ hist, float *__restrict__ out, int n) {
	__shared__ float4 ring4[R * SLOT_BYTES / 16]; // [R][SLOT_BYTES]
	__shared__ uint32_t flags[2 * R];
	char *ring = reinterpret_cast<char *>(ring4);
	const uint32_t full_seq = (uint32_t)(uintptr_t)flags, free_seq = full_seq + 4 * R; // LDS byte addresses
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	if (threadIdx.x < 2 * R) {
		flags[threadIdx.x] = 0;
	}
	__syncthreads();
	const int per = (n + gridDim.x - 1) / gridDim.x;
	const int first = blockIdx.x * per;
	const int cnt = min(per, n - first) > 0 ? min(per, n - first) : 0;
	const uint32_t ring_base = (uint32_t)(uintptr_t)ring;
	if (wave >= NC) {
		// ---- loader ----
		const int L = wave - NC;
		const int mine = cnt > L ? (cnt - L + NL - 1) / NL : 0;
		int issued = 0, signalled = 0, spins = 0;
		while (signalled < mine) {
			while (issued < mine && issued - signalled < MAXOUT) {
				const int i = L + NL * issued, s = i % R, gen = i / R;
				const uint32_t fr = __builtin_amdgcn_readfirstlane(lds_read_u32(free_seq + 4 * s));
				if ((int)fr < gen) {
					break;
				}
				const char *rsrc = rows + (size_t)(first + i) * ROW_BYTES + lane * 16;
				const uint32_t dst = ring_base + s * SLOT_BYTES;
#pragma unroll
				for (int p = 0; p < 4; p++) {
					glds16(rsrc + p * 1024, dst + p * 1024);
				}
				glds16(hist + (size_t)(first + i) * HIST_BYTES + lane * 16, dst + ROW_BYTES);
				issued++;
			}
			if (signalled < issued) {
				wait_vm(P * (issued - signalled - 1));
				const int i = L + NL * signalled;
				lds_write_u32(full_seq + 4 * (i % R), i / R + 1);
				signalled++;
			} else {
				__builtin_amdgcn_s_sleep(2);
				if (++spins > (1 << 18)) { // a protocol bug must not hang the GPU: every wave reaches the end
					if (lane == 0) {
						out[0] = __builtin_nanf("");
					}
					break;
				}
			}
		}
	} else {
		// ---- consumer ----
		float acc = 0.0f;
		for (int i = wave; i < cnt; i += NC) {
			const int s = i % R, gen = i / R;
			int spins = 0;
			while (__builtin_amdgcn_readfirstlane(lds_read_u32(full_seq + 4 * s)) != (uint32_t)(gen + 1) && spins < (1 << 18)) {
				__builtin_amdgcn_s_sleep(1);
				spins++;
			}
			if (spins >= (1 << 18)) {
				acc = __builtin_nanf("");
				break;
			}
			asm volatile("" ::: "memory");
			const float2 *fr = reinterpret_cast<const float2 *>(ring4 + s * (SLOT_BYTES / 16));
			float2 v[8];
#pragma unroll
			for (int q = 0; q < 8; q++) {
				v[q] = fr[lane + 64 * q];
			}
			const float4 hv = ring4[s * (SLOT_BYTES / 16) + ROW_BYTES / 16 + lane];
			float sum = 0.0f;
#pragma unroll
			for (int q = 0; q < 8; q++) {
				sum += v[q].x + v[q].y;
			}
			const float hx = hv.x, hy = hv.y, hz = hv.z, hw = hv.w;
			sum += hx + hy + hz + hw;
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the slot's data is in registers
			lds_write_u32(free_seq + 4 * s, gen + 1);
			// new history out (same bytes every run so the reference stays valid: write back what was read)
			reinterpret_cast<float4 *>(hist + (size_t)(first + i) * HIST_BYTES)[lane] = make_float4(hx, hy, hz, hw);
			acc += sum;
		}
		out[(size_t)blockIdx.x * NC * 64 + wave * 64 + lane] = acc;
	}
}

// Self-service variant: every wave prefetches its own sources D deep into a private ring (no flags, no loaders:
// a wave's own counted vmcnt orders its DMA against its own LDS reads).
template <int D, bool TABLE>
__global__ __launch_bounds__(512, 2) void k_self(const char *__restrict__ rows, char *__restrict__ hist, float *__restrict__ out, int n, const float4 *__restrict__ tab, const unsigned *__restrict__ dirs) {
	__shared__ float4 ring4[8 * D * SLOT_BYTES / 16];
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int n_waves = gridDim.x * 8, gw = blockIdx.x * 8 + wave;
	const int base = n / n_waves, rem = n % n_waves;
	const int first = gw * base + (gw < rem ? gw : rem), cnt = base + (gw < rem ? 1 : 0);
	float4 *my = ring4 + wave * D * (SLOT_BYTES / 16);
	const uint32_t my_base = (uint32_t)(uintptr_t)my;
	auto issue = [&](int k) {
		const uint32_t dst = my_base + (k % D) * SLOT_BYTES;
		const char *rsrc = rows + (size_t)(first + k) * ROW_BYTES + lane * 16;
#pragma unroll
		for (int p = 0; p < 4; p++) {
			glds16(rsrc + p * 1024, dst + p * 1024);
		}
		glds16(hist + (size_t)(first + k) * HIST_BYTES + lane * 16, dst + ROW_BYTES);
	};
	for (int k = 0; k < D && k < cnt; k++) {
		issue(k);
	}
	float acc = 0.0f;
	float4 t[8];
	for (int k = 0; k < cnt; k++) {
		// everything younger than source k's DMA: the DMAs of sources k+1 .. k+D-1, the history stores of k-D+1 .. k-1
		// (one each, issued before the DMA that follows them) -- and, with TABLE, compiler-counted loads (which only
		// ever make the compiler wait longer)
		const int younger = (min(cnt - 1, k + D - 1) - k) * P + max(0, min(k, D - 1));
		if (younger >= 12) {
			asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
		} else if (younger >= 6) {
			asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
		} else {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		const float4 *slot = my + (k % D) * (SLOT_BYTES / 16);
		const float2 *fr = reinterpret_cast<const float2 *>(slot);
		float2 v[8];
#pragma unroll
		for (int q = 0; q < 8; q++) {
			v[q] = fr[lane + 64 * q];
		}
		const float4 hv = slot[ROW_BYTES / 16 + lane];
		float sum = hv.x + hv.y + hv.z + hv.w;
#pragma unroll
		for (int q = 0; q < 8; q++) {
			sum += v[q].x + v[q].y;
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the slot is in registers: it may be refilled
		reinterpret_cast<float4 *>(hist + (size_t)(first + k) * HIST_BYTES)[lane] = hv;
		if (k + D < cnt) {
			issue(k + D);
		}
		if constexpr (TABLE) {
			const unsigned d = dirs[first + k];
			const float4 *tb = tab + (size_t)d * 256;
#pragma unroll
			for (int j = 0; j < 4; j++) {
				t[j] = tb[j * 64 + lane];
			}
#pragma unroll
			for (int j = 4; j < 8; j++) {
				int pp = 512 - (lane + 64 * j);
				pp = pp == 256 ? 0 : pp;
				t[j] = tb[pp];
			}
#pragma unroll
			for (int j = 0; j < 8; j++) {
				sum += t[j].x * t[j].y + t[j].z * t[j].w;
			}
		}
		acc += sum;
	}
	out[(size_t)blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int D, bool TABLE>
void run_self(const char *name, std::vector<char *> &rows, char *hist, float *out, int n, double expect, const float4 *tab, const unsigned *dirs) {
	auto kern = k_self<D, TABLE>;
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	const int wgs = 256;
	for (int i = 0; i < 6; i++) {
		hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), 0, 0, rows[i % rows.size()], hist, out, n, tab, dirs);
	}
	CK(hipDeviceSynchronize());
	std::vector<float> h((size_t)wgs * 512);
	CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
	double got = 0;
	for (float v : h) {
		got += v;
	}
	const int iters = 200;
	CK(hipEventRecord(a));
	for (int i = 0; i < iters; i++) {
		hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), 0, 0, rows[i % rows.size()], hist, out, n, tab, dirs);
	}
	CK(hipEventRecord(b));
	CK(hipEventSynchronize(b));
	float ms;
	CK(hipEventElapsedTime(&ms, a, b));
	const double us = ms / iters * 1e3;
	const double mb = n * (double)(ROW_BYTES + 2 * HIST_BYTES) / 1e6;
	printf("%-28s wgs %4d  %7.2f us/launch (back to back)  %.1f MB -> %.2f TB/s   sum %s (%.6g vs %.6g)\n", name, wgs, us, mb, mb / us, TABLE || fabs(got - expect) <= 1e-3 * fabs(expect) ? "ok" : "MISMATCH", got, expect);
}

template <int NC, int NL, int R, int MAXOUT>
void run(const char *name, std::vector<char *> &rows, char *hist, float *out, int n, int wgs, double expect) {
	const size_t lds = 0;
	auto kern = k_ring<NC, NL, R, MAXOUT>;
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	for (int i = 0; i < 6; i++) {
		hipLaunchKernelGGL(kern, dim3(wgs), dim3((NC + NL) * 64), lds, 0, rows[i % rows.size()], hist, out, n);
	}
	CK(hipDeviceSynchronize());
	// correctness of the last launch
	std::vector<float> h((size_t)wgs * NC * 64);
	CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
	double got = 0;
	for (float v : h) {
		got += v;
	}
	const int iters = 200;
	CK(hipEventRecord(a));
	for (int i = 0; i < iters; i++) {
		hipLaunchKernelGGL(kern, dim3(wgs), dim3((NC + NL) * 64), lds, 0, rows[i % rows.size()], hist, out, n);
	}
	CK(hipEventRecord(b));
	CK(hipEventSynchronize(b));
	float ms;
	CK(hipEventElapsedTime(&ms, a, b));
	const double us = ms / iters * 1e3;
	const double mb = n * (double)(ROW_BYTES + 2 * HIST_BYTES) / 1e6;
	printf("%-28s wgs %4d  %7.2f us/launch (back to back)  %.1f MB -> %.2f TB/s   sum %s (%.6g vs %.6g)\n", name, wgs, us, mb, mb / us, fabs(got - expect) <= 1e-3 * fabs(expect) ? "ok" : "MISMATCH", got, expect);
}

int main(int argc, char **argv) {
	const int n = argc > 1 ? atoi(argv[1]) : 8192;
	const int nbuf = 12;
	std::vector<char *> rows(nbuf);
	std::vector<float> hr((size_t)n * ROW_BYTES / 4);
	double expect = 0;
	for (size_t i = 0; i < hr.size(); i++) {
		hr[i] = (float)((i * 2654435761u >> 20) & 255) / 256.0f;
		expect += hr[i];
	}
	for (auto &p : rows) {
		CK(hipMalloc(&p, (size_t)n * ROW_BYTES));
		CK(hipMemcpy(p, hr.data(), (size_t)n * ROW_BYTES, hipMemcpyHostToDevice));
	}
	std::vector<float> hh((size_t)n * HIST_BYTES / 4);
	for (size_t i = 0; i < hh.size(); i++) {
		hh[i] = (float)((i * 40503u >> 8) & 127) / 128.0f;
		expect += hh[i];
	}
	char *hist;
	CK(hipMalloc(&hist, (size_t)n * HIST_BYTES));
	CK(hipMemcpy(hist, hh.data(), (size_t)n * HIST_BYTES, hipMemcpyHostToDevice));
	float *out;
	CK(hipMalloc(&out, (size_t)1024 * 8 * 64 * 4));
	printf("n = %d sources: %0.1f MB rows + %0.1f MB history read + %0.1f MB history written per launch\n", n, n * 4096.0 / 1e6, n * 1024.0 / 1e6, n * 1024.0 / 1e6);
	float4 *tab;
	CK(hipMalloc(&tab, (size_t)1024 * 256 * 16));
	CK(hipMemset(tab, 0, (size_t)1024 * 256 * 16));
	std::vector<unsigned> hd(n);
	for (int i = 0; i < n; i++) {
		hd[i] = (unsigned)((i * 2654435761u) >> 22) & 1023;
	}
	unsigned *dirs;
	CK(hipMalloc(&dirs, n * 4));
	CK(hipMemcpy(dirs, hd.data(), n * 4, hipMemcpyHostToDevice));
	run_self<1, false>("self D=1", rows, hist, out, n, expect, tab, dirs);
	run_self<2, false>("self D=2", rows, hist, out, n, expect, tab, dirs);
	run_self<3, false>("self D=3", rows, hist, out, n, expect, tab, dirs);
	run_self<1, true>("self D=1 +table(regs)", rows, hist, out, n, expect, tab, dirs);
	run_self<2, true>("self D=2 +table(regs)", rows, hist, out, n, expect, tab, dirs);
	run_self<3, true>("self D=3 +table(regs)", rows, hist, out, n, expect, tab, dirs);
	run<6, 2, 16, 8>("6c+2l ring16 out8", rows, hist, out, n, 256, expect);
	run<6, 2, 20, 10>("6c+2l ring20 out10", rows, hist, out, n, 256, expect);
	run<6, 2, 8, 4>("6c+2l ring8 out4", rows, hist, out, n, 256, expect);
	run<7, 1, 16, 12>("7c+1l ring16 out12", rows, hist, out, n, 256, expect);
	run<6, 2, 16, 8>("6c+2l ring16 out8 512wg", rows, hist, out, n, 512, expect);
	run<4, 4, 16, 4>("4c+4l ring16 out4", rows, hist, out, n, 256, expect);
	run<6, 2, 24, 12>("6c+2l ring24 out12", rows, hist, out, n, 256, expect);
	return 0;
}
