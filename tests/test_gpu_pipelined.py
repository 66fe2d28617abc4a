"""GAS_FLAG_PIPELINED_MIX: the partial-mix sum of callback t runs on the library's second stream under callback
t+1's DSP kernel.  Same arithmetic in the same order, so every output must be bitwise identical to the ordered
mode once gas_ctx_join_outputs / gas_ctx_synchronize has been called."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render(gas, flags, kind, chain, n, F, T, channel_count=1, host_call_at=None, ring=0):
    import torch

    from godot_audio_spatializer_amd import synth

    K = gas.capi
    rng = np.random.default_rng(21)
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=48)
    ctx = gas.SpatializerContext(max_sources=n, frames=F, channel_count=channel_count, flags=flags, er_ring_frames=ring)
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
            ctx.params_publish_batch(slots, synth.draw_params(rng, n, dirs=48, channel_count=channel_count, frames=F, ring_frames=max(ring, 2 * F)))
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


@pytest.mark.parametrize("case", ["hrtf", "mix_channel_4", "hrtf_with_host_call", "hrtf_f256", "hrtf_f128", "hrtf_f384", "hrtf_small_grid", "er_hrtf", "er_hrtf_f512"])
def test_pipelined_mix_is_bitwise_identical(gas, case):
    K = gas.capi
    if case == "mix_channel_4":
        args = dict(kind=K.KIND_3D_MIX, chain=(), n=300, F=512, T=9, channel_count=4)
    elif case.startswith("hrtf_f"):  # other frame counts: other column -> workgroup maps of the carried sum
        args = dict(kind=K.KIND_EFFECT, chain=(K.FX_HRTF,), n=900, F=int(case[6:]), T=7)
    elif case == "er_hrtf":  # the [ER, HRTF] launch carries the pending sum (requests the rows where it sums them)
        args = dict(kind=K.KIND_EFFECT, chain=(K.FX_EARLY_REFLECTIONS, K.FX_HRTF), n=900, F=256, T=7, ring=2048)
    elif case == "er_hrtf_f512":
        args = dict(kind=K.KIND_EFFECT, chain=(K.FX_EARLY_REFLECTIONS, K.FX_HRTF), n=1300, F=512, T=6, ring=4096)
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


@pytest.mark.parametrize("flags_name", ["ordered", "pipelined"])
def test_device_published_rows_feed_the_er_hrtf_launch(gas, ob, flags_name):
    """Deferred device publish with an [ER, HRTF] list: the launch reads gain, direction and the eight taps from the
    published rows and writes them through -- same mixes as publishing the same rows from the host, and the oracle's."""
    import torch

    from godot_audio_spatializer_amd import synth
    from helpers import TOL, rel_rms

    K = gas.capi
    n, F, ring, dirs, T = 600, 256, 2048, 32, 6
    flags = K.FLAG_PEAKS_DRAINING_ONLY | (K.FLAG_PIPELINED_MIX if flags_name == "pipelined" else 0)
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=dirs)
    chain = (K.FX_EARLY_REFLECTIONS, K.FX_HRTF)
    outs = {}
    for how in ("host", "device"):
        rng = np.random.default_rng(12)
        ctx = gas.SpatializerContext(max_sources=n, frames=F, er_ring_frames=ring, flags=flags)
        ctx.hrtf_load(hrir)
        slots = ctx.source_alloc_many(n, K.KIND_EFFECT, chain)
        ora = ob.BatchOracle(ob.KIND_EFFECT, n, F, chain=list(chain), hrir=hrir, er_ring_frames=ring)
        p = synth.draw_params(rng, n, dirs=dirs, ring_frames=ring, frames=F)
        ctx.params_publish_batch(slots, p)
        d_out = torch.zeros(T, 1, F, 2, device="cuda")
        d_pk = torch.zeros(n, 2, device="cuda")
        keep = []
        for t in range(T):
            src = synth.draw_sources(rng, n, F)
            if t in (2, 3, 5):
                p = synth.draw_params(rng, n, dirs=dirs, ring_frames=ring, frames=F)
                if how == "host":
                    ctx.params_publish_batch(slots, p)
                else:
                    d_p = torch.from_numpy(p.view(np.uint8).reshape(n, -1).copy()).cuda()
                    keep.append(d_p)
                    torch.cuda.synchronize()
                    ctx.params_publish_device(d_p.data_ptr(), n)
            d_src = torch.from_numpy(src).cuda()
            keep.append(d_src)
            torch.cuda.synchronize()
            assert ctx.process_block_raw(d_src.data_ptr(), slots if t == 0 else None, n, F, d_out[t].data_ptr(), d_pk.data_ptr(), K.MEM_DEVICE) == 0
            if how == "device":
                ctx.synchronize()
                _, _, r64 = ora.block(p.astype(ob.PARAMS_DTYPE), src, want64=True)
                assert rel_rms(d_out[t].cpu().numpy()[0], r64[0]) <= TOL
            else:
                ora.block(p.astype(ob.PARAMS_DTYPE), src)
        ctx.synchronize()
        outs[how] = d_out.cpu().numpy()
        ctx.close()
    assert np.array_equal(outs["host"], outs["device"])
