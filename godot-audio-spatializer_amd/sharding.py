"""Multi-GPU glue: sources shard by slot across the GPUs of one node, one process per GPU
(SURVEY.md section 8e).  Sources are independent units with private state
(audio_spatializer.cpp:353-470 touches only the playback's own data); the only coupling is the
final sum (audio_spatializer.cpp:433-434,450-451), so the single exchange step is a sum-reduce of
each GPU's partial [C][F] mix to the root over RCCL/xGMI (4 KiB per rank at stereo/512: latency-
bound, so it is pipelined behind the next callback's kernels instead of being made bigger).

torch.distributed is plumbing here (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests)."""
import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous, sticky source ranges balanced by count: returns (begin, end) of this rank."""
    base, extra = divmod(int(n_total), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def owner_of(source_index, n_total, world):
    """Rank that owns a global source index (inverse of shard_range)."""
    base, extra = divmod(int(n_total), int(world))
    edge = extra * (base + 1)
    if source_index < edge:
        return source_index // (base + 1)
    return extra + (source_index - edge) // max(base, 1)


def shard_sizes(n_total, world):
    return np.array([shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)])


class PartialMixReducer:
    """Sum-reduce of the per-GPU partial mixes to `root`, optionally pipelined one callback deep.

    reduce(t): starts the reduce of tensor t (in place on root) and returns a handle;
    the caller must wait(handle) before reading t on root or reusing t anywhere.

    Two arrangements (host cost measured with tools/reduce_launch_probe.py, nccl, 4 KiB): with a `comm_stream` the
    collective runs beside the compute stream (async_op + wait: ~50 us of host time per reduce -- for BUCKETS of
    callbacks, where that is spread over the bucket); without one it is a plain stream-ordered call on the current
    stream (~8-12 us of host time, the compute stream waits for the collective -- for the real-time arrangement, one
    reduce per callback, where 50 us of host time per 20 us callback would be the bottleneck)."""

    def __init__(self, dist, root=0, comm_stream=None):
        self.dist = dist
        self.root = root
        self.comm_stream = comm_stream
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1

    def reduce(self, t):
        if self.world == 1:
            return None
        if t.is_cuda and self.dist.get_backend() == "gloo":
            # rehearsal on a box without enough GPUs for RCCL: stage through the host (synchronous)
            h = t.cpu()
            self.dist.reduce(h, dst=self.root, op=self.dist.ReduceOp.SUM)
            if self.dist.get_rank() == self.root:
                t.copy_(h)
            return None
        if self.comm_stream is not None:
            import torch

            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                return self.dist.reduce(t, dst=self.root, op=self.dist.ReduceOp.SUM, async_op=True)
        self.dist.reduce(t, dst=self.root, op=self.dist.ReduceOp.SUM)  # stream-ordered on the current stream (nccl); synchronous on gloo
        return None

    @staticmethod
    def wait(handle):
        if handle is not None:
            handle.wait()
