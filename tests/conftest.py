import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
PKG = REPO / "handnet-pipeline_amd"
for p in (str(REPO), str(PKG)):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # gpu tests are selected with -m gpu; if someone runs them without a GPU, skip loudly
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def a2j_sd():
    from hn_amd import synth
    return synth.make_a2j_state_dict(seed=0)


@pytest.fixture(scope="session")
def a2j_rgbd_sd():
    from hn_amd import synth
    return synth.make_a2j_state_dict(seed=0, rgbd=True)


@pytest.fixture(scope="session")
def fcos_sd():
    from hn_amd import synth
    return synth.make_fcos_state_dict(seed=0, num_classes=3)
