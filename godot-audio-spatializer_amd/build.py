"""Builds libgas_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgas_amd.so")
SOURCES = ["gas_ctx.hip", "k_biquad_mix.hip", "k_hrtf_ols.hip", "k_misc.hip", "k_calc_spatialization.hip", "k_sample_sources.hip", "gas_multi.hip", "../host/batched_spatializer_host.cpp"]
HEADERS = [os.path.join(CSRC, "gas_internal.h"), os.path.join(CSRC, "gas_device.h"), os.path.join(HERE, "..", "include", "gas_amd.h"), os.path.join(HERE, "..", "include", "gas_amd_host.h")]
# -fno-slp-vectorize: hipcc's SLP pass packs the FFT butterflies into v_pk_{add,mul,fma}_f32; on gfx950 that
# costs VGPRs (236 vs 188) and 15 % of k_hrtf_ols' time (measured, profiles/r01 notes), so it stays off.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X library cannot be built (there is no CPU fallback)")
    return exe


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    cmd = [hipcc()] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
