"""Times gas_process_block_buses on device memory: 3D-mix kinds (fused) and [HRTF] sources (staged)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402

import godot_audio_spatializer_amd as gas  # noqa: E402
from godot_audio_spatializer_amd import synth  # noqa: E402

K = gas.capi
n, F, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 512, 100
rng = np.random.default_rng(0)
for name, kind, chain, n_buses in (("[HRTF] fused, one launch per pair of buses", K.KIND_EFFECT, (K.FX_HRTF,), 2), ("[HRTF] fused, one launch per pair of buses", K.KIND_EFFECT, (K.FX_HRTF,), 4), ("[HRTF] fused, one launch per pair of buses", K.KIND_EFFECT, (K.FX_HRTF,), 6), ("[HIGHSHELF, HRTF] staged", K.KIND_EFFECT, (K.FX_HIGHSHELF, K.FX_HRTF), 4), ("3D mix", K.KIND_3D_MIX, (), 2), ("3D mix", K.KIND_3D_MIX, (), 6)):
    ctx = gas.SpatializerContext(max_sources=n, frames=F, flags=K.FLAG_PEAKS_DRAINING_ONLY)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.hrtf_load(synth.synthetic_hrir(rng, dirs=1024))
    slots = ctx.source_alloc_many(n, kind, chain)
    ctx.params_publish_batch(slots, synth.draw_params(rng, n, dirs=1024))
    routes = K.bus_routes(n)
    routes["send_bus"] = 1
    routes["send"][:, 0, :] = 0.3
    for k in range(min(n_buses - 2, K.MAX_MORE_SENDS)):  # every source reaches every bus of the call
        routes["more_bus"][:, k] = 2 + k
        routes["more_send"][:, k, 0, :] = 0.2
    ctx.bus_routes_publish(slots, routes)
    src = torch.rand(n, F, 2, device="cuda") - 0.5
    out = torch.zeros(n_buses, 1, F, 2, device="cuda")
    pk = torch.zeros(n, 2, device="cuda")
    s32 = np.ascontiguousarray(slots, np.uint32)

    def step(first=False):
        staged = len(chain) > 1  # the staged form regroups every call: it needs the list
        rc = ctx.lib.gas_process_block_buses(ctx.h, src.data_ptr(), s32.ctypes.data if (first or staged) else None, n, F, out.data_ptr(), n_buses, pk.data_ptr(), K.MEM_DEVICE)
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
