// fftbench.hip -- dev aid: the wave-level FFT-512 of gas_hrtf_wave.h (scalar f32 VALU) against gas_fft_pk.h (packed
// f32), 256 workgroups x 8 waves at 2 waves/SIMD, R transforms per wave back to back on register data.
// Prints time per transform per wave and the largest difference between the two results.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I godot-audio-spatializer_amd/csrc -I include tools/micro/fftbench.hip -o tools/micro/fftbench
#include "gas_hrtf_wave.h"
#include "gas_fft_pk.h" // tools/micro/
#include <cstdio>
#include <cmath>
#include <vector>

template <int MODE, int NW = WAVES> // 0 scalar forward, 1 packed forward, 2 scalar fwd+inv, 3 packed fwd+inv
__global__ __launch_bounds__(NW * 64, 1) void k_fft(const float2 *__restrict__ in, const float2 *__restrict__ tw, float2 *__restrict__ out, int R) {
	__shared__ float2 lds_all[NW * LDS_F2_PER_WAVE];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	float2 *lds = lds_all + wave * LDS_F2_PER_WAVE;
	float2 t1[8], t2[8], v[8];
	for (int k = 0; k < 8; k++) {
		t1[k] = tw[k * 64 + lane];
		t2[k] = tw[(8 + k) * 64 + lane];
		v[k] = in[(size_t)(blockIdx.x * NW + wave) * 512 + k * 64 + lane];
	}
	if constexpr (MODE == 4) { // two independent transforms per step, interleaved (fft512_pair)
		float2 w[8];
		for (int k = 0; k < 8; k++) {
			w[k] = make_float2(v[k].y, v[k].x);
		}
		float2 *lds1 = lds + LDS_F2_HALF;
		for (int r = 0; r < R; r += 2) {
			fft512_pair<false>(v, w, t1, t2, lds, lds1, lane);
			for (int k = 0; k < 8; k++) {
				v[k].x *= (1.0f / 512); v[k].y *= (1.0f / 512); w[k].x *= (1.0f / 512); w[k].y *= (1.0f / 512);
			}
		}
		for (int k = 0; k < 8; k++) {
			v[k].x += w[k].x; v[k].y += w[k].y;
		}
	} else if constexpr (MODE == 0 || MODE == 2) {
		for (int r = 0; r < R; r++) {
			fft512<false>(v, t1, t2, lds, lane);
			if (MODE == 2) {
				fft512<true>(v, t1, t2, lds, lane);
			}
			for (int k = 0; k < 8; k++) {
				v[k].x *= (1.0f / 512); v[k].y *= (1.0f / 512);
			}
		}
	} else {
		v2f p1[8], p2[8], pv[8];
		for (int k = 0; k < 8; k++) {
			p1[k] = to_v2f(t1[k]); p2[k] = to_v2f(t2[k]); pv[k] = to_v2f(v[k]);
		}
		v2f *pl = reinterpret_cast<v2f *>(lds);
		for (int r = 0; r < R; r++) {
			pk_fft512<false>(pv, p1, p2, pl, lane);
			if (MODE == 3) {
				pk_fft512<true>(pv, p1, p2, pl, lane);
			}
			for (int k = 0; k < 8; k++) {
				pv[k] *= (1.0f / 512);
			}
		}
		for (int k = 0; k < 8; k++) {
			v[k] = to_f2(pv[k]);
		}
	}
	for (int k = 0; k < 8; k++) {
		out[(size_t)(blockIdx.x * NW + wave) * 512 + k * 64 + lane] = v[k];
	}
}

int main() {
	const int WG = 256, NW = WG * 16;
	std::vector<float2> h_in((size_t)NW * 512), h_tw(16 * 64), a(h_in.size()), b(h_in.size());
	unsigned s = 12345;
	for (auto &x : h_in) {
		s = s * 1664525u + 1013904223u; x.x = (float)(s >> 8) / 8388608.0f - 1.0f;
		s = s * 1664525u + 1013904223u; x.y = (float)(s >> 8) / 8388608.0f - 1.0f;
	}
	for (int k = 0; k < 8; k++) {
		for (int l = 0; l < 64; l++) {
			const int hi = l >> 3, lo = l & 7;
			const double a1 = -2.0 * M_PI * (hi * k) / 64.0, a2 = -2.0 * M_PI * (lo * (hi + 8 * k)) / 512.0;
			h_tw[k * 64 + l] = make_float2((float)cos(a1), (float)sin(a1));
			h_tw[(8 + k) * 64 + l] = make_float2((float)cos(a2), (float)sin(a2));
		}
	}
	float2 *d_in, *d_tw, *d_out;
	hipMalloc(&d_in, h_in.size() * 8); hipMalloc(&d_tw, h_tw.size() * 8); hipMalloc(&d_out, h_in.size() * 8);
	hipMemcpy(d_in, h_in.data(), h_in.size() * 8, hipMemcpyHostToDevice);
	hipMemcpy(d_tw, h_tw.data(), h_tw.size() * 8, hipMemcpyHostToDevice);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const int R = 64;
	for (int w = 0; w < 200; w++) hipLaunchKernelGGL(k_fft<0>, dim3(WG), dim3(512), 0, 0, d_in, d_tw, d_out, R); // clock warm-up
	for (int mode = 0; mode < 5; mode++) {
		float best = 1e9f;
		for (int it = 0; it < 6; it++) {
			hipEventRecord(e0);
			switch (mode) {
				case 0: hipLaunchKernelGGL(k_fft<0>, dim3(WG), dim3(512), 0, 0, d_in, d_tw, d_out, R); break;
				case 1: hipLaunchKernelGGL(k_fft<1>, dim3(WG), dim3(512), 0, 0, d_in, d_tw, d_out, R); break;
				case 2: hipLaunchKernelGGL(k_fft<2>, dim3(WG), dim3(512), 0, 0, d_in, d_tw, d_out, R); break;
				case 3: hipLaunchKernelGGL(k_fft<3>, dim3(WG), dim3(512), 0, 0, d_in, d_tw, d_out, R); break;
				case 4: hipLaunchKernelGGL(k_fft<4>, dim3(WG), dim3(512), 0, 0, d_in, d_tw, d_out, R); break;
			}
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			best = ms < best ? ms : best;
		}
		hipMemcpy((mode & 1) ? b.data() : a.data(), d_out, h_in.size() * 8, hipMemcpyDeviceToHost);
		const int per = ((mode == 2 || mode == 3) ? 2 : 1) * R;
		printf("mode %d (%s, %s): %.3f ms, %.1f ns per transform per wave (2 waves/SIMD)\n", mode, mode == 4 ? "scalar pair" : (mode & 1) ? "packed" : "scalar", (mode == 2 || mode == 3) ? "fwd+inv" : "fwd", best, best * 1e6 / per);
		if (mode & 1) {
			double md = 0, mx = 0;
			for (size_t i = 0; i < a.size(); i++) {
				md = fmax(md, fabs((double)a[i].x - b[i].x)); md = fmax(md, fabs((double)a[i].y - b[i].y));
				mx = fmax(mx, fabs((double)a[i].x));
			}
			printf("   max |scalar - packed| = %.3e (max |value| %.3e)\n", md, mx);
		}
	}
	// occupancy sweep of the scalar forward transform: transforms per microsecond per CU
	for (int nw = 4; nw <= 16; nw += 4) {
		float best = 1e9f;
		for (int it = 0; it < 6; it++) {
			hipEventRecord(e0);
			switch (nw) {
				case 4: hipLaunchKernelGGL((k_fft<0, 4>), dim3(WG), dim3(256), 0, 0, d_in, d_tw, d_out, 256); break;
				case 8: hipLaunchKernelGGL((k_fft<0, 8>), dim3(WG), dim3(512), 0, 0, d_in, d_tw, d_out, 256); break;
				case 12: hipLaunchKernelGGL((k_fft<0, 12>), dim3(WG), dim3(768), 0, 0, d_in, d_tw, d_out, 256); break;
				case 16: hipLaunchKernelGGL((k_fft<0, 16>), dim3(WG), dim3(1024), 0, 0, d_in, d_tw, d_out, 256); break;
			}
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			best = ms < best ? ms : best;
		}
		printf("%d waves/SIMD: %.3f ms for 256 transforms per wave -> %.2f transforms/us/CU\n", nw / 4, best, nw * 256 / (best * 1e3));
	}
	return 0;
}
