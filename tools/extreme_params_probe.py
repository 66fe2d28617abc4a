import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import godot_audio_spatializer_amd as gas
from godot_audio_spatializer_amd import synth
from oracle import binding as ob
from helpers import rel_rms
K = gas.capi
for cutoff in (5000.0, 500.0, 50.0):
    for gain in (0.5, 0.05, 0.001):
        rng = np.random.default_rng(3)
        n, F = 64, 512
        worst = 0.0
        with gas.SpatializerContext(max_sources=n, frames=F) as ctx:
            slots = ctx.source_alloc_many(n, K.KIND_3D_MIX)
            ora = ob.BatchOracle(ob.KIND_3D_MIX, n, F)
            for b in range(12):
                p = synth.draw_params(rng, n)
                p["linear_attenuation"] = gain * rng.uniform(0.9, 1.1, n)
                p["attenuation_filter_cutoff_hz"] = cutoff
                ctx.params_publish_batch(slots, p)
                src = synth.draw_sources(rng, n, F)
                mix, _ = ctx.process_block(src, slots)
                _, _, r64 = ora.block(p.astype(ob.PARAMS_DTYPE), src, want64=True)
                worst = max(worst, rel_rms(mix[0], r64[0]))
        print(f"cutoff {cutoff:7.1f} Hz gain {gain:6.3f}: worst rel rms over 12 callbacks {worst:.2e}")
