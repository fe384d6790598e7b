import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.build()
    binding.set_threads(min(binding.usable_cores(), 16))
    return binding


@pytest.fixture(scope="session")
def blue_noise():
    from sunray_amd import scenes
    return scenes.white_noise_rgba8()
