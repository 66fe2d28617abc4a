"""Generates tests/golden/*.npz: seeded inputs and the outputs of the CPU oracle (oracle/gas_oracle.c).

These vectors pin THIS REPOSITORY'S restatement against regressions; they do not come from the reference,
which ships no fixtures and cannot be built here (PARITY UNPINNED).
Run from the repo root:  python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import binding as ob  # noqa: E402

sys.path.insert(0, ROOT)
from godot_audio_spatializer_amd import synth  # noqa: E402

CASES = {
    # name: (kind, chain, n, frames, channel_count, ring, blocks)
    "cfg1_process_frames_n1": (ob.KIND_3D_PROCESS, (), 1, 512, 1, 0, 4),
    "cfg1_mix_channel_n1": (ob.KIND_3D_MIX, (), 1, 512, 1, 0, 4),
    "cfg2_mix_channel_n24": (ob.KIND_3D_MIX, (), 24, 512, 1, 0, 4),
    "mix_channel_4pairs_n6": (ob.KIND_3D_MIX, (), 6, 512, 4, 0, 3),
    "fx_highshelf_n5": (ob.KIND_EFFECT, (ob.FX_HIGHSHELF,), 5, 512, 1, 0, 3),
    "cfg3_hrtf_n12": (ob.KIND_EFFECT, (ob.FX_HRTF,), 12, 512, 1, 0, 4),
    "cfg5_er_hrtf_n8_f256": (ob.KIND_EFFECT, (ob.FX_EARLY_REFLECTIONS, ob.FX_HRTF), 8, 256, 1, 4096, 18),
}


def generate(name):
    kind, chain, n, frames, C, ring, blocks = CASES[name]
    rng = np.random.default_rng(0)  # seed 0 for parity fixtures
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=8) if ob.FX_HRTF in chain else None
    ora = ob.BatchOracle(kind, n, frames, channel_count=C, chain=chain, hrir=hrir, er_ring_frames=max(ring, 1))
    P, S, M, K = [], [], [], []
    for b in range(blocks):
        if b % 2 == 0:
            p = synth.draw_params(rng, n, dirs=8, channel_count=C, ring_frames=max(ring, 2 * frames), frames=frames)
        src = synth.draw_sources(rng, n, frames)
        mix, peaks, _ = ora.block(p.astype(ob.PARAMS_DTYPE), src)
        P.append(p.copy().view(np.uint8).reshape(n, 128))
        S.append(src)
        M.append(mix)
        K.append(peaks)
    out = dict(params=np.stack(P), src=np.stack(S), mix=np.stack(M), peaks=np.stack(K), kind=kind, chain=np.array(chain, np.int32), frames=frames, channel_count=C, ring=ring)
    if hrir is not None:
        out["hrir"] = hrir
    return out


if __name__ == "__main__":
    d = os.path.join(ROOT, "tests", "golden")
    os.makedirs(d, exist_ok=True)
    for name in CASES:
        np.savez_compressed(os.path.join(d, name + ".npz"), **generate(name))
        print("wrote", name)
