import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import dm_amd_bootstrap  # noqa: F401,E402  (registers the package in ./diffusion-model_amd as `dm_amd`)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import torch

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = torch.load(os.path.join(GOLDEN, name), weights_only=True)
        return cache[name]

    return load


@pytest.fixture(scope="session")
def dev():
    import torch

    assert torch.cuda.is_available(), "GPU test collected on a machine without a GPU"
    return torch.device("cuda:0")
