"""Builds libgas_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgas_amd.so")
SOURCES = ["gas_ctx.hip", "k_biquad_mix.hip", "k_biquad_pipe.hip", "k_shelf_scan.hip", "k_hrtf_ols.hip", "k_hrtf_uni.hip", "k_hrtf_multi.hip", "k_misc.hip", "k_calc_spatialization.hip", "k_sample_sources.hip", "gas_multi.hip", "../host/batched_spatializer_host.cpp"]
HEADERS = [os.path.join(CSRC, "gas_internal.h"), os.path.join(CSRC, "gas_device.h"), os.path.join(CSRC, "gas_hrtf_wave.h"), os.path.join(CSRC, "gas_biquad.h"), os.path.join(HERE, "..", "include", "gas_amd.h"), os.path.join(HERE, "..", "include", "gas_amd_host.h")]
# -fno-slp-vectorize: hipcc's SLP pass packs the FFT butterflies into v_pk_{add,mul,fma}_f32; on gfx950 that
# costs VGPRs (236 vs 188) and 15 % of k_hrtf_ols' time (measured, profiles/r01 notes), so it stays off.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize"]
OBJ_DIR = os.path.join(HERE, "..", "build", "obj")


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


def _compile_and_link(out, extra_flags, tag, verbose):
    """One hipcc -c per translation unit, in parallel (the HRTF kernels dominate: ~30 s each), then one link."""
    from concurrent.futures import ThreadPoolExecutor

    obj_dir = os.path.abspath(os.path.join(OBJ_DIR, tag))
    os.makedirs(obj_dir, exist_ok=True)
    exe = hipcc()

    def one(src):
        obj = os.path.join(obj_dir, os.path.basename(src).replace(".", "_") + ".o")
        cmd = [exe] + FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(one, SOURCES))
    cmd = [exe, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    return _compile_and_link(LIB, [], "product", verbose)


def build_variant(name, extra_flags, verbose=False):
    """Development aid: an A/B build with extra -D switches into build/variants/libgas_<name>.so (loaded through
    GAS_AMD_LIB by tools/ab_libs.sh); the product library is never replaced by it."""
    out_dir = os.path.join(HERE, "..", "build", "variants")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.abspath(os.path.join(out_dir, f"libgas_{name}.so"))
    return _compile_and_link(out, extra_flags, "variant_" + name, verbose)


if __name__ == "__main__":
    import sys

    if len(sys.argv) >= 3 and sys.argv[1] == "--variant":
        print(build_variant(sys.argv[2], sys.argv[3:], verbose=True))
    else:
        print(build(force=True, verbose=True))
