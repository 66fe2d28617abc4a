"""Dev aid: host enqueue time vs GPU completion time of the bench's step loop."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402

import godot_audio_spatializer_amd as gas  # noqa: E402
from godot_audio_spatializer_amd import synth  # noqa: E402


def run(n, steps=400):
    rng = np.random.default_rng(0)
    ctx = gas.SpatializerContext(max_sources=n, frames=512, flags=gas.capi.FLAG_PEAKS_DRAINING_ONLY | (gas.capi.FLAG_PIPELINED_MIX if "pipe" in sys.argv else 0))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.hrtf_load(synth.synthetic_hrir(rng, dirs=1024))
    slots = ctx.source_alloc_many(n, 2, (3,))
    p = synth.draw_params(rng, n)
    ctx.params_publish_batch(slots, p)
    pd = torch.from_numpy(p.view(np.uint8).reshape(n, 128).copy()).cuda()
    src = torch.rand(n, 512, 2, device="cuda") - 0.5
    out = torch.zeros(1, 512, 2, device="cuda")
    pk = torch.zeros(n, 2, device="cuda")
    ctx.process_block_raw(src.data_ptr(), slots, n, 512, out.data_ptr(), pk.data_ptr(), 1)
    torch.cuda.synchronize()
    for with_publish in (False, True):
        t0 = time.perf_counter()
        for k in range(steps):
            if with_publish and k % 2 == 0:
                ctx.params_publish_device(pd.data_ptr(), n)
            ctx.process_block_raw(src.data_ptr(), None, n, 512, out.data_ptr(), pk.data_ptr(), 1)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"n={n:6d} publish={with_publish}: enqueue {1e6 * (t1 - t0) / steps:6.1f} us/step, complete {1e6 * (t2 - t0) / steps:6.1f} us/step")
    ctx.close()


if __name__ == "__main__":
    for n in (64, 2048, 8192, 32768):
        run(n)
