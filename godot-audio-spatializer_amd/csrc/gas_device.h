// gas_device.h -- device-side helpers shared by kernels of different files (wave64, gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

// Sum of v over the 64 lanes in ONE fixed association order (DPP row shifts 1,2,4,8, then row broadcasts 15 and
// 31): bitwise reproducible, identical wherever it is used.  Every lane returns the total.
__device__ __forceinline__ float gas_wave_sum(float v) {
	int x = __float_as_int(v);
#define GAS_DPP_ADD(ctrl, row_mask) x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, ctrl, row_mask, 0xF, false)))
	GAS_DPP_ADD(0x111, 0xF); // row_shr:1
	GAS_DPP_ADD(0x112, 0xF); // row_shr:2
	GAS_DPP_ADD(0x114, 0xF); // row_shr:4
	GAS_DPP_ADD(0x118, 0xF); // row_shr:8
	GAS_DPP_ADD(0x142, 0xA); // row_bcast:15 -> rows 1, 3
	GAS_DPP_ADD(0x143, 0xC); // row_bcast:31 -> rows 2, 3
#undef GAS_DPP_ADD
	return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}

// The final sum of the per-workgroup partial mixes for ONE float4 column (audio_spatializer.cpp:433-434,450-451
// turned into a fixed-order tree): lane l adds partial rows l, l + 64, l + 128, ... in ascending order, then the 64
// lane sums are folded by gas_wave_sum.  k_mix_reduce and the sum carried by k_hrtf_ols (GAS_FLAG_PIPELINED_MIX) both
// go through these two functions, which is what makes the two modes agree to the bit.
__device__ __forceinline__ void gas_mix_column_add(float4 &s, const float4 a) {
	s.x += a.x;
	s.y += a.y;
	s.z += a.z;
	s.w += a.w;
}

__device__ __forceinline__ float4 gas_mix_column_fold(const float4 s) {
	return make_float4(gas_wave_sum(s.x), gas_wave_sum(s.y), gas_wave_sum(s.z), gas_wave_sum(s.w));
}
