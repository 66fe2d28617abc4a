"""GAS_FLAG_PIPELINED_MIX: the partial-mix sum of callback t runs on the library's second stream under callback
t+1's DSP kernel.  Same arithmetic in the same order, so every output must be bitwise identical to the ordered
mode once gas_ctx_join_outputs / gas_ctx_synchronize has been called."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render(gas, flags, kind, chain, n, F, T, channel_count=1, host_call_at=None):
    import torch

    from godot_audio_spatializer_amd import synth

    K = gas.capi
    rng = np.random.default_rng(21)
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=48)
    ctx = gas.SpatializerContext(max_sources=n, frames=F, channel_count=channel_count, flags=flags)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.hrtf_load(hrir)
    slots = ctx.source_alloc_many(n, kind, chain)
    for s in slots[::9]:
        ctx.source_set_draining(s, True)
    outs = torch.full((T, channel_count, F, 2), float("nan"), device="cuda")
    peaks = torch.zeros(T, n, 2, device="cuda")
    host_mix = None
    for t in range(T):
        if t % 3 == 0:
            ctx.params_publish_batch(slots, synth.draw_params(rng, n, dirs=48, channel_count=channel_count, frames=F))
        src = synth.draw_sources(rng, n, F)
        if t == host_call_at:  # an ordered (host-memory) call in the middle of the queue
            host_mix, _ = ctx.process_block(src, slots)
            outs[t] = torch.from_numpy(host_mix).cuda()
            continue
        d_src = torch.from_numpy(src).cuda()
        rc = ctx.process_block_raw(d_src.data_ptr(), slots, n, F, outs[t].data_ptr(), peaks[t].data_ptr(), K.MEM_DEVICE)
        assert rc == 0
    ctx.join_outputs()
    res = outs.clone()  # enqueued on the context's stream: must see every mix
    pk = peaks.clone()
    torch.cuda.synchronize()
    ctx.close()
    return res.cpu().numpy(), pk.cpu().numpy()


@pytest.mark.parametrize("case", ["hrtf", "mix_channel_4", "hrtf_with_host_call", "hrtf_f256", "hrtf_f128", "hrtf_f384", "hrtf_small_grid"])
def test_pipelined_mix_is_bitwise_identical(gas, case):
    K = gas.capi
    if case == "mix_channel_4":
        args = dict(kind=K.KIND_3D_MIX, chain=(), n=300, F=512, T=9, channel_count=4)
    elif case.startswith("hrtf_f"):  # other frame counts: other column -> workgroup maps of the carried sum
        args = dict(kind=K.KIND_EFFECT, chain=(K.FX_HRTF,), n=900, F=int(case[6:]), T=7)
    elif case == "hrtf_small_grid":  # too few workgroups to carry a sum: summed immediately
        args = dict(kind=K.KIND_EFFECT, chain=(K.FX_HRTF,), n=40, F=512, T=6)
    else:
        args = dict(kind=K.KIND_EFFECT, chain=(K.FX_HRTF,), n=700, F=512, T=10, host_call_at=4 if case.endswith("host_call") else None)
    base, pk0 = _render(gas, K.FLAG_PEAKS_DRAINING_ONLY, **args)
    pipe, pk1 = _render(gas, K.FLAG_PEAKS_DRAINING_ONLY | K.FLAG_PIPELINED_MIX, **args)
    assert not np.isnan(base).any() and np.abs(base).max() > 0
    assert np.array_equal(base, pipe)
    assert np.array_equal(pk0, pk1)


def test_pipelined_mix_synchronize_covers_the_reduce_stream(gas):
    """gas_ctx_synchronize alone (no join) must leave every queued output complete."""
    import torch

    from godot_audio_spatializer_amd import synth

    K = gas.capi
    rng = np.random.default_rng(2)
    n, F, T = 2048, 512, 6
    outs = {}
    for flags in (0, K.FLAG_PIPELINED_MIX):
        rng = np.random.default_rng(2)
        ctx = gas.SpatializerContext(max_sources=n, frames=F, flags=flags)  # the context's own stream
        ctx.hrtf_load(synth.synthetic_hrir(np.random.default_rng(7), dirs=32))
        slots = ctx.source_alloc_many(n, K.KIND_EFFECT, (K.FX_HRTF,))
        ctx.params_publish_batch(slots, synth.draw_params(rng, n, dirs=32, frames=F))
        src = torch.from_numpy(synth.draw_sources(rng, n, F)).cuda()
        out = torch.zeros(T, 1, F, 2, device="cuda")
        pk = torch.zeros(n, 2, device="cuda")
        torch.cuda.synchronize()
        for t in range(T):
            assert ctx.process_block_raw(src.data_ptr(), slots if t == 0 else None, n, F, out[t].data_ptr(), pk.data_ptr(), K.MEM_DEVICE) == 0
        ctx.synchronize()
        outs[flags] = out.cpu().numpy()
        ctx.close()
    assert np.array_equal(outs[0], outs[K.FLAG_PIPELINED_MIX])
