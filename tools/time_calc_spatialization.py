"""Times gas_calc_spatialization for 65 536 sources, host-array and device-array forms."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import torch  # noqa: E402

import godot_audio_spatializer_amd as gas  # noqa: E402
from test_gpu_calc_spatialization import scene  # noqa: E402

K = gas.capi
rng = np.random.default_rng(0)
n = 65536
cfgs, poses, listeners, cfg_index = scene(gas, rng, n, 1, 4)
with gas.SpatializerContext(max_sources=n, frames=512) as ctx:
    slots = ctx.source_alloc_many(n, K.KIND_3D_MIX)
    ctx.calc_spatialization(cfgs, poses, listeners, slots, cfg_index=cfg_index)
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.calc_spatialization(cfgs, poses, listeners, slots, cfg_index=cfg_index, want_params=False)
    dt = (time.perf_counter() - t0) / 20
    print(f"host arrays, no read-back : {dt * 1e6:8.1f} us per tick for {n} sources ({n / dt:.3e} sources/s)")
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.calc_spatialization(cfgs, poses, listeners, slots, cfg_index=cfg_index, want_params=True)
    dt = (time.perf_counter() - t0) / 20
    print(f"host arrays, params read back: {dt * 1e6:8.1f} us per tick")
    d_poses = torch.from_numpy(poses.view(np.uint8).reshape(n, 48).copy()).cuda()
    s = np.ascontiguousarray(slots, np.uint32)
    ci = np.ascontiguousarray(cfg_index, np.uint32)

    def dev():
        rc = ctx.lib.gas_calc_spatialization(ctx.h, cfgs.ctypes.data, len(cfgs), ci.ctypes.data, d_poses.data_ptr(), listeners.ctypes.data, 1, s.ctypes.data, n, None, K.MEM_DEVICE)
        assert rc == 0

    dev()
    t0 = time.perf_counter()
    for _ in range(20):
        dev()
    dt = (time.perf_counter() - t0) / 20
    print(f"device-resident poses        : {dt * 1e6:8.1f} us per tick (includes the slot / config index upload and a stream sync)")
