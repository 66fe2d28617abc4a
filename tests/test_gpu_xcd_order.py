"""GAS_FLAG_XCD_ORDER (k_xcd_order): the XCD-affine processing order of plain [HRTF] callbacks.

The order only decides which workgroup sums which source -- the mix is the same sum -- so parity with the oracle
holds at the usual 1e-5 relative RMS; the test also reads the order back and checks that it is a permutation, a pure
function of the list and the directions, and that it does what it is for: workgroup b (XCD b % 8) gets the sources
whose direction lies in eighth b % 8 of the table, up to the few that overflow a bucket's quota."""
import numpy as np
import pytest

from helpers import TOL, rel_rms
from test_gpu_parity import run_pair

pytestmark = pytest.mark.gpu


def _wave_first(n, gw, n_waves):
    base, rem = divmod(n, n_waves)
    return gw * base + min(gw, rem)


@pytest.mark.parametrize("n,frames", [(2048, 512), (3001, 512), (2304, 256)])
def test_xcd_order_parity_with_oracle(gas, ob, n, frames):
    from godot_audio_spatializer_amd import synth

    K = gas.capi
    hrir = synth.synthetic_hrir(np.random.default_rng(5), dirs=64)
    run_pair(gas, ob, K.KIND_EFFECT, (K.FX_HRTF,), n, frames, 5, hrir=hrir, dirs=64, flags=K.FLAG_XCD_ORDER | K.FLAG_PEAKS_DRAINING_ONLY, draining_every=37)


def test_xcd_order_is_a_stable_bucketed_permutation(gas, ob):
    from godot_audio_spatializer_amd import synth

    K = gas.capi
    n, F, dirs = 4100, 512, 64
    rng = np.random.default_rng(11)
    hrir = synth.synthetic_hrir(np.random.default_rng(5), dirs=dirs)
    ctx = gas.SpatializerContext(max_sources=n, frames=F, flags=K.FLAG_XCD_ORDER | K.FLAG_PEAKS_DRAINING_ONLY)
    plain = gas.SpatializerContext(max_sources=n, frames=F, flags=K.FLAG_PEAKS_DRAINING_ONLY)
    try:
        for c in (ctx, plain):
            c.hrtf_load(hrir)
        slots = ctx.source_alloc_many(n, K.KIND_EFFECT, (K.FX_HRTF,))
        pslots = plain.source_alloc_many(n, K.KIND_EFFECT, (K.FX_HRTF,))
        with pytest.raises(gas.GasError):
            ctx.read_hrtf_order(n)  # nothing ran yet
        orders = []
        for b in range(4):
            if b != 1:  # callback 1 reuses callback 0's order (nothing was published)
                p = synth.draw_params(rng, n, dirs=dirs)
                ctx.params_publish_batch(slots, p)
                plain.params_publish_batch(pslots, p)
            src = synth.draw_sources(rng, n, F)
            mix, _ = ctx.process_block(src, slots)
            ref, _ = plain.process_block(src, pslots)
            assert rel_rms(mix[0], ref[0]) <= 2e-6  # the same sum in another order
            order = ctx.read_hrtf_order(n)
            orders.append(order)
            assert np.array_equal(np.sort(order), np.arange(n, dtype=np.uint32))
            # placement: 256 workgroups x 8 waves; workgroup b covers positions [first(8 b), first(8 b + 8))
            bucket = (p["hrtf_dir"].astype(np.int64) * 8 // dirs).clip(0, 7)
            home = 0
            for wg in range(256):
                a, e = _wave_first(n, 8 * wg, 2048), _wave_first(n, 8 * wg + 8, 2048)
                got = bucket[order[a:e]]
                home += int((got == wg % 8).sum())
                # the entries that belong here come first and in list order (stable)
                mine = order[a:e][got == wg % 8]
                assert np.all(np.diff(mine.astype(np.int64)) > 0)
            assert home >= 0.85 * n  # 16 entries per workgroup here: a bucket of a segment (binomial, mean 16, sigma 3.7) overflows its quota by ~1.5
        assert np.array_equal(orders[0], orders[1])
        assert not np.array_equal(orders[1], orders[2])
    finally:
        ctx.close()
        plain.close()


def test_xcd_order_with_device_published_parameters_and_repeatability(gas, ob):
    """The deferred device publish (rows read by the HRTF launch itself) feeds the order kernel the same rows; two
    contexts fed identically produce bitwise identical mixes (the order is deterministic)."""
    import torch

    from godot_audio_spatializer_amd import synth

    K = gas.capi
    n, F, dirs = 2560, 512, 128
    hrir = synth.synthetic_hrir(np.random.default_rng(5), dirs=dirs)
    outs = []
    for rep in range(2):
        rng = np.random.default_rng(3)
        ctx = gas.SpatializerContext(max_sources=n, frames=F, flags=K.FLAG_XCD_ORDER | K.FLAG_PEAKS_DRAINING_ONLY)
        ora = ob.BatchOracle(ob.KIND_EFFECT, n, F, chain=[ob.FX_HRTF], hrir=hrir)
        try:
            ctx.hrtf_load(hrir)
            slots = ctx.source_alloc_many(n, K.KIND_EFFECT, (K.FX_HRTF,))
            p = synth.draw_params(rng, n, dirs=dirs)
            ctx.params_publish_batch(slots, p)
            d_out = torch.zeros(1, F, 2, device="cuda")
            d_peaks = torch.zeros(n, 2, device="cuda")
            got = []
            for b in range(4):
                src = synth.draw_sources(rng, n, F)
                d_src = torch.from_numpy(src).cuda()
                if b in (1, 3):
                    p = synth.draw_params(rng, n, dirs=dirs)
                    d_p = torch.from_numpy(p.view(np.uint8).reshape(n, -1)).cuda()
                    torch.cuda.synchronize()
                    ctx.params_publish_device(d_p.data_ptr(), n)
                torch.cuda.synchronize()
                rc = ctx.process_block_raw(d_src.data_ptr(), slots if b == 0 else None, n, F, d_out.data_ptr(), d_peaks.data_ptr(), K.MEM_DEVICE)
                assert rc == 0
                ctx.synchronize()
                mix = d_out.cpu().numpy()
                _, _, r64 = ora.block(p.astype(ob.PARAMS_DTYPE), src, want64=True)
                assert rel_rms(mix[0], r64[0]) <= TOL
                got.append(mix.copy())
            outs.append(got)
        finally:
            ctx.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
