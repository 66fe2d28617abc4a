"""Times gas_process_block_buses on device memory: 3D-mix kinds (fused) and [HRTF] sources (staged)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402

import godot_audio_spatializer_amd as gas  # noqa: E402
from godot_audio_spatializer_amd import synth  # noqa: E402

K = gas.capi
n, F, steps, n_buses = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 512, 100, 2
rng = np.random.default_rng(0)
for name, kind, chain in (("[HRTF] fused two-bus form", K.KIND_EFFECT, (K.FX_HRTF,)), ("3D mix", K.KIND_3D_MIX, ())):
    ctx = gas.SpatializerContext(max_sources=n, frames=F, flags=K.FLAG_PEAKS_DRAINING_ONLY)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.hrtf_load(synth.synthetic_hrir(rng, dirs=1024))
    slots = ctx.source_alloc_many(n, kind, chain)
    ctx.params_publish_batch(slots, synth.draw_params(rng, n, dirs=1024))
    routes = K.bus_routes(n)
    routes["send_bus"] = 1
    routes["send"][:, 0, :] = 0.3
    ctx.bus_routes_publish(slots, routes)
    src = torch.rand(n, F, 2, device="cuda") - 0.5
    out = torch.zeros(n_buses, 1, F, 2, device="cuda")
    pk = torch.zeros(n, 2, device="cuda")
    s32 = np.ascontiguousarray(slots, np.uint32)

    def step(first=False):
        rc = ctx.lib.gas_process_block_buses(ctx.h, src.data_ptr(), s32.ctypes.data if first else None, n, F, out.data_ptr(), n_buses, pk.data_ptr(), K.MEM_DEVICE)
        assert rc == 0, rc

    step(True)

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {n} sources, {n_buses} buses: {e0.elapsed_time(e1) / steps * 1e3:.1f} us per callback")
    ctx.close()
