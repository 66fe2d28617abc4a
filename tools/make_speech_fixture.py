"""Cuts a small excerpt of the reference's example asset (examples/godot-gd-spatializer/speech_orig.wav: mono,
16-bit, 48 kHz) into tests/golden/speech_excerpt_s16.npy.  Data only (PCM samples); run in the build container
where /root/reference is mounted.  The excerpt is a realistic input signal for the sampler / mixer tests."""
import os
import sys
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/examples/godot-gd-spatializer/speech_orig.wav"
with wave.open(src, "rb") as w:
    assert w.getnchannels() == 1 and w.getsampwidth() == 2 and w.getframerate() == 48000, (w.getnchannels(), w.getsampwidth(), w.getframerate())
    w.setpos(48000)  # skip the first second
    pcm = np.frombuffer(w.readframes(12000), dtype="<i2").copy()  # 0.25 s
np.save(os.path.join(ROOT, "tests", "golden", "speech_excerpt_s16.npy"), pcm)
print(pcm.shape, pcm.dtype, int(np.abs(pcm).max()))
