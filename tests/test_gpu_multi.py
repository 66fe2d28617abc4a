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
