"""ThreadSanitizer run of the host layer (godot-audio-spatializer_amd/host/batched_spatializer_host.cpp) on the CPU:
start / stop / set-parameters / queries from two control threads while a third runs get_mixed_frames, against a
TEST-ONLY stub of the gas_* entries the host calls (tests/host_tsan/gas_stub.cpp -- never part of the product).
This is the threading contract of include/gas_amd_host.h, i.e. the split the reference keeps with SafeList /
SafeFlag / Mutex (audio_spatializer.h:57-68, audio_spatializer.cpp:558-574, deferred delete :538-547)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hammer(tmp_path_factory):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    out = str(tmp_path_factory.mktemp("tsan") / "hammer")
    src = [os.path.join(ROOT, "godot-audio-spatializer_amd", "host", "batched_spatializer_host.cpp"), os.path.join(ROOT, "tests", "host_tsan", "gas_stub.cpp"), os.path.join(ROOT, "tests", "host_tsan", "hammer.cpp")]
    # the multi-GPU entries of gas_amd_host.h live in csrc/gas_multi.hip and are not linked here
    subprocess.check_call([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-o", out] + src)
    return out


@pytest.mark.parametrize("device_mode", [0, 1])
def test_host_layer_is_race_free_under_tsan(hammer, device_mode):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    r = subprocess.run([hammer, "1500", str(device_mode)], capture_output=True, text=True, timeout=240, env=env)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-2000:])
    assert "playbacks left 0" in r.stdout
