"""Numpy model of the wave-level 512-point FFT used by k_hrtf_ols (64 lanes x 8 points).

Mirrors the kernel's lane/register/LDS index maps one-for-one so the HIP code can be
checked against it.  Run: python tools/fft512_prototype.py
"""
import numpy as np

S2 = np.float32(0.70710678118654752440)


def dft8(v, inv):
    """v: [8][...] complex. Forward uses W8 = exp(-2pi i/8); inverse conjugates."""
    sgn = 1j if inv else -1j  # multiply by -i (fwd) or +i (inv)
    a0 = v[0] + v[4]; a1 = v[0] - v[4]
    a2 = v[2] + v[6]; a3 = (v[2] - v[6]) * sgn
    a4 = v[1] + v[5]; a5 = v[1] - v[5]
    a6 = v[3] + v[7]; a7 = (v[3] - v[7]) * sgn
    b0 = a0 + a2; b2 = a0 - a2
    b1 = a1 + a3; b3 = a1 - a3
    b4 = a4 + a6; b6 = (a4 - a6) * sgn
    b5 = a5 + a7; b7 = a5 - a7
    w1 = (1 + sgn) * S2          # W8^1 = (1 - i)/sqrt2 fwd
    w3 = (-1 + sgn) * S2         # W8^3 = (-1 - i)/sqrt2 fwd
    b5 = b5 * w1; b7 = b7 * w3
    return [b0 + b4, b1 + b5, b2 + b6, b3 + b7, b0 - b4, b1 - b5, b2 - b6, b3 - b7]


def twiddles():
    l = np.arange(64)
    n1 = l >> 3; n0 = l & 7; k0p = l >> 3
    tw1 = np.exp(-2j * np.pi * (n1[None, :] * np.arange(8)[:, None]) / 64)            # [k0][lane]
    tw2 = np.exp(-2j * np.pi * (n0[None, :] * (k0p[None, :] + 8 * np.arange(8)[:, None])) / 512)  # [k1][lane]
    return tw1.astype(np.complex64), tw2.astype(np.complex64)


def fft512_wave(v, inv=False):
    """v: complex64 [8][64] with v[j][l] = x[l + 64 j]; returns same layout of X."""
    tw1, tw2 = twiddles()
    if inv:
        tw1 = tw1.conj(); tw2 = tw2.conj()
    l = np.arange(64)
    u = dft8(v, inv)
    u = [u[k0] * tw1[k0] for k0 in range(8)]
    lds = np.zeros(8 * 72, np.complex64)
    for k0 in range(8):
        lds[k0 * 72 + l] = u[k0]
    k0p = l >> 3; n0 = l & 7
    w = [lds[k0p * 72 + n1 * 8 + n0] for n1 in range(8)]
    t = dft8(w, inv)
    t = [t[k1] * tw2[k1] for k1 in range(8)]
    lds2 = np.zeros(8 * 66, np.complex64)
    for k1 in range(8):
        lds2[n0 * 66 + k1 * 8 + k0p] = t[k1]
    r = [lds2[m * 66 + l] for m in range(8)]
    X = dft8(r, inv)
    return np.stack(X).astype(np.complex64)


def to_lanes(x):
    return x.reshape(8, 64).astype(np.complex64)   # [j][l] = x[l + 64 j]


def from_lanes(v):
    return v.reshape(512)


def hrtf_block(hist, xnew, hl, hr):
    """Overlap-save of one callback: hist[HL], xnew[F] (gained mono), hl/hr[256]. F in {256, 512}."""
    F = len(xnew); S = F // 2; HL = 512 - S
    assert len(hist) == HL
    xf = np.concatenate([hist, xnew]).astype(np.float32)
    z = xf[0:512] + 1j * xf[S:S + 512]
    Z = fft512_wave(to_lanes(z))
    out = np.zeros((F, 2), np.float32)
    for ear, h in enumerate((hl, hr)):
        H = np.fft.fft(np.concatenate([h, np.zeros(256)])).astype(np.complex64) / 512
        y = from_lanes(fft512_wave(Z * to_lanes(H), inv=True))
        out[:S, ear] = y.real[512 - S:]
        out[S:, ear] = y.imag[512 - S:]
    return out, xf[F:F + HL]


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(512) + 1j * rng.standard_normal(512)).astype(np.complex64)
    X = from_lanes(fft512_wave(to_lanes(x)))
    ref = np.fft.fft(x.astype(np.complex128))
    print("fwd rel err", np.abs(X - ref).max() / np.abs(ref).max())
    xi = from_lanes(fft512_wave(to_lanes(X), inv=True)) / 512
    print("inv rel err", np.abs(xi - x).max())
    for F in (512, 256):
        S = F // 2; HL = 512 - S
        hl = (rng.standard_normal(256) * np.exp(-np.arange(256) / 32)).astype(np.float32)
        hr = (rng.standard_normal(256) * np.exp(-np.arange(256) / 32)).astype(np.float32)
        sig = rng.uniform(-0.5, 0.5, 5 * F).astype(np.float32)
        hist = np.zeros(HL, np.float32)
        outs = []
        for b in range(5):
            o, hist = hrtf_block(hist, sig[b * F:(b + 1) * F], hl, hr)
            outs.append(o)
        o = np.concatenate(outs)
        refl = np.convolve(sig.astype(np.float64), hl.astype(np.float64))[:len(sig)]
        refr = np.convolve(sig.astype(np.float64), hr.astype(np.float64))[:len(sig)]
        e = np.sqrt(((o[:, 0] - refl) ** 2 + (o[:, 1] - refr) ** 2).mean()) / np.sqrt((refl ** 2 + refr ** 2).mean())
        print("F", F, "OLS rel rms err", e)
