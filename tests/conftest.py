import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gas():
    import godot_audio_spatializer_amd as g

    g.build.build()
    return g


@pytest.fixture(scope="session")
def ob():
    from oracle import binding

    binding.build()
    return binding
