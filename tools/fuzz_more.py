"""Dev aid: tests/test_gpu_fuzz.py's random configurations for a seed range beyond the suite's (python tools/fuzz_more.py 100 300)."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import godot_audio_spatializer_amd as gas  # noqa: E402
from oracle import binding as ob  # noqa: E402
import test_gpu_fuzz as tf  # noqa: E402

ob.build()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    try:
        tf.test_random_configuration.__wrapped__(gas, ob, seed) if hasattr(tf.test_random_configuration, "__wrapped__") else tf.test_random_configuration(gas, ob, seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED:", str(e)[:300].replace("\n", " "))
print(f"{hi - lo - bad} of {hi - lo} seeds pass")
sys.exit(1 if bad else 0)
