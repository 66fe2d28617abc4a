"""Round notes helper: times every BASELINE config on one GPU (device-resident inputs) and the PCIe-inclusive
host-memory path of the headline workload.  Prints a small table (copied into profiles/r01_notes.md)."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bench(args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras"] + args, capture_output=True, text=True).stdout
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def host_path(n=8192, F=512, iters=30):
    import godot_audio_spatializer_amd as gas
    from godot_audio_spatializer_amd import synth

    rng = np.random.default_rng(0)
    ctx = gas.SpatializerContext(max_sources=n, frames=F, flags=gas.capi.FLAG_PEAKS_DRAINING_ONLY)
    ctx.hrtf_load(synth.synthetic_hrir(rng, dirs=1024))
    slots = ctx.source_alloc_many(n, 2, (3,))
    ctx.params_publish_batch(slots, synth.draw_params(rng, n))
    src = synth.draw_sources(rng, n, F)
    import torch

    pinned = torch.from_numpy(src).pin_memory().numpy()
    res = {}
    for name, buf in (("pageable", src), ("pinned", pinned)):
        ctx.process_block(buf, slots)
        t0 = time.perf_counter()
        for _ in range(iters):
            ctx.process_block(buf, slots)
        dt = (time.perf_counter() - t0) / iters
        res[name] = (dt * 1e3, n * F / dt)
    ctx.close()
    return res


if __name__ == "__main__":
    for wl, extra in (("hrtf", []), ("hrtf", ["--exact-peaks"]), ("hrtf4096", []), ("biquad", []), ("erhrtf", []), ("biquad", ["--sources-per-gpu", "65536"])):
        r = bench(["--workload", wl] + extra)
        rf = r["roofline"]
        print(f"{wl:9s} {' '.join(extra):24s} n={r['config']['sources_per_gpu']:6d} F={r['config']['frames_per_callback']} ms/step={r['ms_per_step']:.4f} frames/s={r['value']:.3e} kernel={rf['kernel']} {rf['kernel_us']:.1f}us algB={rf['algorithmic_bytes_per_launch']/1e6:.1f}MB frac={rf['frac']:.3f}")
    for k, (ms, fps) in host_path().items():
        print(f"host-memory path ({k}): 8192 HRTF sources, {ms:.3f} ms per callback incl. 32 MiB H2D + D2H, {fps:.3e} AudioFrames/s")
