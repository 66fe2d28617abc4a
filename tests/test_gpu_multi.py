"""Single-process multi-GPU layer (gas_multi_*): shards are placed on the one available GPU, which
exercises everything but the xGMI link itself -- per-shard contexts and streams, the all-to-one gather into the
root buffer, the ordered sum, per-shard peaks."""
import numpy as np
import pytest

from helpers import TOL, rel_rms

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shards,kind_name", [(2, "hrtf"), (3, "mix4"), (1, "hrtf")])
def test_sharded_mix_equals_single_context_and_oracle(gas, ob, shards, kind_name):
    from godot_audio_spatializer_amd import sharding, synth

    K = gas.capi
    rng = np.random.default_rng(31)
    n_total, F = 203, 512
    C = 4 if kind_name == "mix4" else 1
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=16)
    kind, chain, okind, ochain = (K.KIND_EFFECT, (K.FX_HRTF,), ob.KIND_EFFECT, [ob.FX_HRTF]) if kind_name == "hrtf" else (K.KIND_3D_MIX, (), ob.KIND_3D_MIX, [])
    multi = K.MultiContext([0] * shards, max_sources=n_total, frames=F, channel_count=C)
    try:
        ranges = [sharding.shard_range(n_total, g, shards) for g in range(shards)]
        slots = []
        for g, ctx in enumerate(multi.shards):
            ctx.hrtf_load(hrir)
            slots.append(ctx.source_alloc_many(ranges[g][1] - ranges[g][0], kind, chain))
        ora = ob.BatchOracle(okind, n_total, F, channel_count=C, chain=ochain, hrir=hrir)
        for b in range(4):
            if b % 2 == 0:
                p = synth.draw_params(rng, n_total, dirs=16, channel_count=C)
                for g, ctx in enumerate(multi.shards):
                    ctx.params_publish_batch(slots[g], p[ranges[g][0]:ranges[g][1]])
            src = synth.draw_sources(rng, n_total, F)
            mix, peaks = multi.process_block([src[a:e] for a, e in ranges], slots)
            _, rpeaks, r64 = ora.block(p.astype(ob.PARAMS_DTYPE), src, want64=True)
            for c in range(C):
                assert rel_rms(mix[c], r64[c]) <= TOL
            np.testing.assert_allclose(np.concatenate(peaks), rpeaks, rtol=2e-5, atol=1e-7)
    finally:
        multi.close()


def test_multi_argument_checks(gas):
    K = gas.capi
    multi = K.MultiContext([0, 0], max_sources=8, frames=512)
    try:
        assert multi.lib.gas_multi_shards(multi.h) == 2
        assert multi.lib.gas_multi_least_loaded(multi.h) == 0
        multi.lib.gas_multi_note_alloc(multi.h, 0, 3)
        assert multi.lib.gas_multi_least_loaded(multi.h) == 1
        with pytest.raises(gas.GasError):  # slot never allocated on shard 1
            multi.process_block([np.zeros((0, 512, 2), np.float32), np.zeros((1, 512, 2), np.float32)], [np.zeros(0, np.uint32), np.array([5], np.uint32)])
        mix, _ = multi.process_block([np.zeros((0, 512, 2), np.float32)] * 2, [np.zeros(0, np.uint32)] * 2)
        assert not mix.any()
    finally:
        multi.close()


def _n_gpus():
    import torch

    return torch.cuda.device_count()


def _run_multi(gas, ob, devices, device_memory):
    """Sharded callbacks over `devices`; host-memory or device-memory entry; returns (mixes, single-context mixes)."""
    import torch

    from godot_audio_spatializer_amd import sharding, synth

    K = gas.capi
    rng = np.random.default_rng(77)
    n_total, F, shards = 300, 512, len(devices)
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=16)
    multi = K.MultiContext(devices, max_sources=n_total, frames=F)
    single = gas.SpatializerContext(max_sources=n_total, frames=F, device=devices[0])
    got, want = [], []
    try:
        single.hrtf_load(hrir)
        s_slots = single.source_alloc_many(n_total, K.KIND_EFFECT, (K.FX_HRTF,))
        ranges = [sharding.shard_range(n_total, g, shards) for g in range(shards)]
        slots = []
        for g, ctx in enumerate(multi.shards):
            ctx.hrtf_load(hrir)
            slots.append(ctx.source_alloc_many(ranges[g][1] - ranges[g][0], K.KIND_EFFECT, (K.FX_HRTF,)))
        for b in range(5):
            p = synth.draw_params(rng, n_total, dirs=16)
            single.params_publish_batch(s_slots, p)
            for g, ctx in enumerate(multi.shards):
                ctx.params_publish_batch(slots[g], p[ranges[g][0]:ranges[g][1]])
            src = synth.draw_sources(rng, n_total, F)
            want.append(single.process_block(src, s_slots)[0])
            if device_memory:
                d_src = [torch.from_numpy(src[a:e]).to(f"cuda:{devices[g]}") for g, (a, e) in enumerate(ranges)]
                d_out = torch.zeros(1, F, 2, device=f"cuda:{devices[0]}")
                torch.cuda.synchronize()
                multi.process_block_device([t.data_ptr() for t in d_src], slots, d_out.data_ptr())
                multi.synchronize()
                got.append(d_out.cpu().numpy())
            else:
                got.append(multi.process_block([src[a:e] for a, e in ranges], slots)[0])
    finally:
        multi.close()
        single.close()
    return got, want


@pytest.mark.parametrize("threads", ["0", "1"])
@pytest.mark.parametrize("direct", ["1", "0"])
def test_multi_device_memory_entry_on_one_gpu(gas, ob, direct, threads, monkeypatch):
    """gas_multi_process_block_mem(GAS_MEM_DEVICE): no staging, no host wait, gather buffer reuse guarded by the root's
    event -- five queued callbacks over three shards of the one GPU equal the single-context mixes.  direct = 1: every
    shard's own final sum writes its row of the root's gather buffer (SURVEY 8e's all-to-one direct write); 0: the
    hipMemcpyPeerAsync form."""
    monkeypatch.setenv("GAS_MULTI_DIRECT", direct)
    monkeypatch.setenv("GAS_MULTI_THREADS", threads)  # 1: one enqueue thread per shard beyond the first
    got, want = _run_multi(gas, ob, [0, 0, 0], device_memory=True)
    for g, w in zip(got, want):
        assert rel_rms(g[0], w[0]) <= TOL


@pytest.mark.parametrize("direct", ["1", "0"])
@pytest.mark.parametrize("device_memory", [False, True])
def test_multi_over_two_real_gpus(gas, ob, device_memory, direct, monkeypatch):
    """The real peer path (peer stores / hipMemcpyPeerAsync into the root's gather buffer over xGMI): needs two GPUs,
    skipped on the one-GPU boxes this repository is developed on -- so that the first multi-GPU run is not the first
    execution."""
    if _n_gpus() < 2:
        pytest.skip("needs >= 2 GPUs")
    monkeypatch.setenv("GAS_MULTI_DIRECT", direct)
    got, want = _run_multi(gas, ob, [0, 1], device_memory=device_memory)
    for g, w in zip(got, want):
        assert rel_rms(g[0], w[0]) <= TOL


def test_bench_two_rank_nccl_smoke():
    """bench.py --gpus 2 under torch.distributed.run with the nccl (RCCL) backend: the sharded job must run and report
    world 2; skipped without two GPUs."""
    import json
    import os
    import subprocess
    import sys

    if _n_gpus() < 2:
        pytest.skip("needs >= 2 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--sources-per-gpu", "1024"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["sources_total"] == 2048 and "world 2" in line["config"]["parallelism"]
