#!/usr/bin/env python3
"""Host cost of ONE torch.distributed reduce launch of a 4 KiB tensor on the nccl (= RCCL) backend -- the per-callback
price of the real-time arrangement (bench.py --reduce-bucket 1).  With the one GPU of a development box the process
group has ONE rank, so the collective itself is trivial: what is measured is the host path (Python -> c10d ->
ProcessGroupNCCL -> enqueue), a LOWER bound of what a callback pays per reduce on N ranks.  Usage (GPU box):
python tools/reduce_launch_probe.py"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.zeros(1, 512, 2, device="cuda")
side = torch.cuda.Stream()
for mode in ("current stream, blocking call", "current stream, async_op + wait (the collective runs on the process group's own stream)", "side stream, async_op + wait (round 2's PartialMixReducer)"):
    for _ in range(50):
        dist.reduce(t, dst=0)
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    if mode.startswith("current stream, blocking"):
        for _ in range(n):
            dist.reduce(t, dst=0)
    elif mode.startswith("current stream, async"):
        for _ in range(n):
            h = dist.reduce(t, dst=0, async_op=True)
            h.wait()
    else:
        for _ in range(n):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                h = dist.reduce(t, dst=0, async_op=True)
            h.wait()
    host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / n
    print(f"{mode}: host enqueue {host * 1e6:.1f} us per reduce, {total * 1e6:.1f} us per reduce incl. completion (1 rank, 4 KiB)")
dist.destroy_process_group()
