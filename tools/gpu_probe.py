"""Ad-hoc GPU probe: prints parity errors and raw kernel timings (development aid)."""
import sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import godot_audio_spatializer_amd as gas
from godot_audio_spatializer_amd import synth
from oracle import binding as ob
from test_gpu_parity import run_pair

hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=64)
print("mix256", run_pair(gas, ob, 0, (), 256, 512, 6))
print("proc", run_pair(gas, ob, 1, (), 130, 512, 6))
print("shelf", run_pair(gas, ob, 2, (1,), 77, 512, 5))
print("hrtf512", run_pair(gas, ob, 2, (3,), 300, 512, 5, hrir=hrir))
print("hrtf256", run_pair(gas, ob, 2, (3,), 130, 256, 5, hrir=hrir))
print("erhrtf", run_pair(gas, ob, 2, (2, 3), 90, 256, 20, hrir=hrir, ring=4096, redraw_every=3))

import torch
def timeit(kind, chain, n, frames, ring=0, dirs=1024, iters=50):
    rng = np.random.default_rng(1234)
    ctx = gas.SpatializerContext(max_sources=n, frames=frames, er_ring_frames=ring)
    if 3 in chain:
        ctx.hrtf_load(synth.synthetic_hrir(rng, dirs=dirs))
    slots = ctx.source_alloc_many(n, kind, chain)
    ctx.params_publish_batch(slots, synth.draw_params(rng, n, dirs=dirs, ring_frames=max(ring, 2*frames), frames=frames))
    src = torch.rand(n, frames, 2, device="cuda") - 0.5
    out = torch.zeros(1, frames, 2, device="cuda")
    peaks = torch.zeros(n, 2, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.profile_enable(True)
    rc = ctx.process_block_raw(src.data_ptr(), slots, n, frames, out.data_ptr(), peaks.data_ptr(), 1)
    assert rc == 0, rc
    torch.cuda.synchronize()
    for _ in range(5):
        ctx.process_block_raw(src.data_ptr(), None, n, frames, out.data_ptr(), peaks.data_ptr(), 1)
    torch.cuda.synchronize(); ctx.profile_read(True)
    t0 = time.perf_counter()
    for _ in range(iters):
        ctx.process_block_raw(src.data_ptr(), None, n, frames, out.data_ptr(), peaks.data_ptr(), 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    pr = ctx.profile_read(True)
    kms = pr["kernel_ms"] / max(pr["launches"], 1)
    print(f"{pr['kernel']:28s} n={n:6d} F={frames} wall/blk={dt*1e6:8.1f}us kernel={kms*1e3:8.1f}us bytes={pr['bytes_per_launch']/1e6:.1f}MB -> {pr['bytes_per_launch']/(kms*1e-3)/1e12:.3f} TB/s  frames/s={n*frames/dt:.3e}")
    ctx.close()

timeit(0, (), 256, 512)
timeit(0, (), 4096, 512)
timeit(0, (), 65536, 512)
timeit(2, (3,), 4096, 512)
timeit(2, (3,), 8192, 512)
timeit(2, (3,), 65536, 512)
timeit(2, (2, 3), 4096, 256, ring=4096)
