"""Seeded synthetic workloads of SURVEY.md section 8(d): sources, SpatializerParameters drawn from the
reference's own pan / attenuation formulas, and a synthetic HRIR set.  Harness-side numpy only
(tests and bench.py); nothing here is on the audio path."""
import numpy as np

from .capi import ER_TAPS, HRTF_TAPS, PARAMS_DTYPE


def db_to_linear(db):
    # [ENGINE] Math::db_to_linear
    return np.exp(np.asarray(db, dtype=np.float64) * 0.11512925464970228)


def stereo_pan(azimuth, pan_strength=0.5):
    """calc_output_vol_stereo, audio_spatializer_3d.cpp:103-110 (f64). azimuth 0 = front, +pi/2 = right."""
    cosx = np.clip(np.sin(azimuth), -1.0, 1.0)
    g = np.clip((1.0 - pan_strength) ** 2, 0.0, 1.0)
    f = (1.0 - g) / (1.0 + g)
    fcosx = cosx * f
    return np.stack([np.sqrt((-fcosx + 1.0) / 2.0), np.sqrt((fcosx + 1.0) / 2.0)], axis=-1)


def draw_params(rng, n, dirs=1024, channel_count=1, ring_frames=4096, frames=512, unit_size=10.0, filter_db=-24.0, cutoff_hz=5000.0):
    """One physics tick of parameters for n sources: azimuth ~ U(-pi, pi),
    distance ~ logU(1, 100) m, inverse-distance multiplier min(1, unit_size/d)
    (audio_spatializer_3d.cpp:127), high-shelf gain db_to_linear((1 - mult) * -24 dB) (:376,:387)."""
    p = np.zeros(n, dtype=PARAMS_DTYPE)
    az = rng.uniform(-np.pi, np.pi, n)
    d = np.exp(rng.uniform(np.log(1.0), np.log(100.0), n))
    mult = np.minimum(1.0, unit_size / d)
    pan = stereo_pan(az) * mult[:, None]
    p["mix_volumes"][:, 0, :] = pan
    for c in range(1, channel_count):  # surround pairs: independent plausible gains
        p["mix_volumes"][:, c, :] = stereo_pan(rng.uniform(-np.pi, np.pi, n)) * mult[:, None] * 0.5
    p["pitch_scale"] = 1.0
    p["linear_attenuation"] = db_to_linear((1.0 - np.minimum(1.0, mult)) * filter_db)
    p["attenuation_filter_cutoff_hz"] = cutoff_hz
    p["update_parameters"] = 1
    p["hrtf_gain"] = mult
    p["hrtf_dir"] = rng.integers(0, dirs, n)
    p["fx_shelf_gain"] = p["linear_attenuation"]
    p["fx_shelf_cutoff_hz"] = cutoff_hz
    p["er_gain"] = 0.7 ** np.arange(1, ER_TAPS + 1)
    hi = min(4000, ring_frames - frames)
    p["er_delay"] = rng.integers(48, hi + 1, (n, ER_TAPS))
    return p


def draw_sources(rng, n, frames):
    """i.i.d. uniform(-0.5, 0.5) per ear, float32 [n][frames][2]."""
    return rng.uniform(-0.5, 0.5, (n, frames, 2)).astype(np.float32)


def synthetic_hrir(rng, dirs=1024):
    """Decaying noise exp(-k/32) N(0,1) with a per-direction interaural delay."""
    k = np.arange(HRTF_TAPS)
    h = rng.standard_normal((dirs, 2, HRTF_TAPS)) * np.exp(-k / 32.0)
    itd = rng.integers(0, 32, dirs)
    out = np.zeros_like(h)
    for d in range(dirs):
        s = int(itd[d])
        out[d, 0, :] = h[d, 0, :]
        out[d, 1, s:] = h[d, 1, : HRTF_TAPS - s]
    return (out * 0.25).astype(np.float32)
