"""Times a callback with device-resident PCM16 streams: sampler kernel + HRTF path, no PCIe."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402

import godot_audio_spatializer_amd as gas  # noqa: E402
from godot_audio_spatializer_amd import synth  # noqa: E402

K = gas.capi
n, F, steps = 8192, 512, 200
rng = np.random.default_rng(0)
per = (steps + 30) * F
big = (rng.uniform(-0.5, 0.5, n * per) * 32767).astype(np.int16)  # one mono PCM16 stream, a private segment per playback
ctx = gas.SpatializerContext(max_sources=n, frames=F, flags=K.FLAG_PEAKS_DRAINING_ONLY)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.hrtf_load(synth.synthetic_hrir(rng, dirs=1024))
slots = ctx.source_alloc_many(n, K.KIND_EFFECT, (K.FX_HRTF,))
ctx.params_publish_batch(slots, synth.draw_params(rng, n))
sid = ctx.stream_create(big)
for i, s in enumerate(slots):
    ctx.source_bind_stream(s, sid, i * per)
out = torch.zeros(1, F, 2, device="cuda")
pk = torch.zeros(n, 2, device="cuda")
s32 = np.ascontiguousarray(slots, np.uint32)


def step():
    rc = ctx.lib.gas_process_block_streams(ctx.h, s32.ctypes.data, n, F, out.data_ptr(), pk.data_ptr(), None, K.MEM_DEVICE)
    assert rc == 0, rc


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t_enq = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"host enqueue {t_enq * 1e6:.1f} us per callback")
print(f"device-resident PCM16 streams: {n} HRTF sources, {dt * 1e6:.1f} us per callback, {n * F / dt:.3e} AudioFrames/s (stream bytes {n * F * 2 / 1e6:.1f} MB/callback)")
ctx.close()
