"""An independent twin of the reference's time-varying biquad path, written from the prose of SURVEY.md Appendix A
(items 8-10) and Appendix B -- plain Python loops over numpy float32 scalars, no code shared with
oracle/gas_oracle.c -- and compared with the C oracle over several callbacks with changing parameters.
The twin rounds to f32 where the reference does (members are float; coefficients are computed in double, stored as
float, then normalised), so the two must agree to ~1e-6; a float64 twin is NOT a usable check here: the 512 in-place
`coeffs += incr` steps per block drift ~3e-5 in f32 and, with the poles near the unit circle at small shelf gains,
that alone moves the output by ~1e-3.  A semantic slip (ramp start, history clear, lerp form, which pair feeds
prev_mix_volume, stored-negated feedback terms) shows up as a gross mismatch either way."""
import math

import numpy as np
import pytest

from helpers import rel_rms


def highshelf64(sr, cutoff, gain):
    """Appendix B, HIGHSHELF with resonance 1, one stage; a1, a2 returned NEGATED as the engine stores them."""
    cutoff = min(max(cutoff, 1.0), int(sr / 2) + 512)
    w = 2.0 * math.pi * cutoff / sr
    A = max(gain, 0.001)
    beta = math.sqrt(A) / 1.0
    cw, sw = math.cos(w), math.sin(w)
    a0 = (A + 1) - (A - 1) * cw + beta * sw
    f = np.float32
    # members are float: the numerators are stored (rounded) before the division by a0 rounds again
    b0 = f(float(f(A * ((A + 1) + (A - 1) * cw + beta * sw))) / a0)
    b1 = f(float(f(-2 * A * ((A - 1) + (A + 1) * cw))) / a0)
    b2 = f(float(f(A * ((A + 1) + (A - 1) * cw - beta * sw))) / a0)
    a1 = f(float(f(2 * ((A - 1) - (A + 1) * cw))) / -a0)
    a2 = f(float(f((A + 1) - (A - 1) * cw - beta * sw)) / -a0)
    return [b0, b1, b2, a1, a2]


class Proc:
    def __init__(self):
        z = np.float32(0.0)
        self.c = [z] * 5  # b0 b1 b2 a1 a2, zero-initialised
        self.h = [z] * 4  # ha1 ha2 hb1 hb2

    def run(self, xs, target, clear):
        n = np.float32(len(xs))
        if clear:
            self.h = [np.float32(0.0)] * 4
        inc = [(t - c) / n for t, c in zip(target, self.c)]  # update_coeffs(len): ramp from the current coefficients
        out = []
        for x in xs:
            x = np.float32(x)
            b0, b1, b2, a1, a2 = self.c
            ha1, ha2, hb1, hb2 = self.h
            y = x * b0 + hb1 * b1 + hb2 * b2 + ha1 * a1 + ha2 * a2
            self.h = [y, ha1, x, hb1]
            self.c = [c + i for c, i in zip(self.c, inc)]
            out.append(y)
        return out


class Twin3D:
    """SpatializerPlaybackData3D + the two entry points, float64."""

    def __init__(self):
        self.prev = {}
        self.procs = [Proc() for _ in range(8)]

    def get_prev(self, c):
        return self.prev.get(c, (0.0, 0.0))

    def mix_channel(self, mixv, gain, cutoff, c, src):
        F = len(src)
        f = np.float32
        vs, vf = self.get_prev(c), mixv[c]
        out = np.zeros((F, 2), np.float32)
        scaled = np.zeros((F, 2), np.float32)
        for i in range(F):
            t = f(i) / f(F)
            for ear in range(2):
                scaled[i, ear] = (f(vf[ear]) * t + (f(1) - t) * f(vs[ear])) * f(src[i, ear])
        if gain >= 0.001:
            target = highshelf64(48000.0, cutoff, gain)
            clear = vs[0] == 0 and vs[1] == 0
            for ear in range(2):
                out[:, ear] = self.procs[2 * c + ear].run(scaled[:, ear], target, clear)
        else:
            out = scaled
        self.prev[c] = tuple(mixv[c])
        return out

    def process_frames(self, mixv, gain, cutoff, src):
        pv = self.get_prev(0)
        if gain >= 0.001:
            target = highshelf64(48000.0, cutoff, gain)
            clear = pv[0] == 0 and pv[1] == 0
            out = np.stack([self.procs[ear].run(src[:, ear], target, clear) for ear in range(2)], axis=1)
        else:
            out = src.astype(np.float32).copy()
        best, idx = 0.0, 0
        for i in range(4):  # strict '>', first wins, all-zero -> pair 0
            for ear in range(2):
                if mixv[i][ear] > best:
                    best, idx = mixv[i][ear], i
        self.prev[0] = tuple(mixv[idx])
        return out


def schedule(rng, blocks):
    """Parameter sets per block: ramps, a bypass stretch, a silent stretch that re-arms the history clear."""
    ps = []
    for b in range(blocks):
        mixv = rng.uniform(0.05, 1.0, (4, 2))
        gain = float(rng.uniform(0.05, 1.0))
        if b in (3, 4):
            gain = 0.0005  # bypass branch
        if b == 6:
            mixv[:] = 0.0  # previous volume becomes (0, 0): the next filtered block clears the history
        ps.append((mixv, gain, float(rng.choice([2000.0, 5000.0, 9000.0]))))
    return ps


@pytest.mark.parametrize("mode", ["mix_channel", "process_frames"])
def test_float32_twin_agrees_with_c_oracle(ob, mode):
    rng = np.random.default_rng(17)
    F, blocks = 512, 9
    sched = schedule(rng, blocks)
    twin = Twin3D()
    kind = ob.KIND_3D_MIX if mode == "mix_channel" else ob.KIND_3D_PROCESS
    ora = ob.BatchOracle(kind, 1, F, channel_count=2)
    for b, (mixv, gain, cutoff) in enumerate(sched):
        src = rng.uniform(-0.5, 0.5, (F, 2)).astype(np.float32)
        p = np.zeros(1, ob.PARAMS_DTYPE)
        p["mix_volumes"][0] = mixv
        p["linear_attenuation"] = gain
        p["attenuation_filter_cutoff_hz"] = cutoff
        # the oracle sees the f32-rounded parameters; feed the twin the same values
        mv32 = p["mix_volumes"][0].astype(np.float64)
        g32, c32 = float(p["linear_attenuation"][0]), float(p["attenuation_filter_cutoff_hz"][0])
        got, _, _ = ora.block(p, src[None])
        if mode == "mix_channel":
            for c in range(2):
                want = twin.mix_channel(mv32, g32, c32, c, src)
                scale = max(np.sqrt(np.mean(want.astype(np.float64) ** 2)), 1e-9)
                assert np.sqrt(np.mean((got[c] - want.astype(np.float64)) ** 2)) / scale < 2e-6, (b, c)
        else:
            want = twin.process_frames(mv32, g32, c32, src)
            assert rel_rms(got[0], want) < 2e-6, b
