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
    # The CPU oracle (torch fp32 on the host) is the slow side of every parity test.  A GPU box shows the host's cores to
    # torch but gives the job a share of 16: with the default thread count the oracle ran 3-4x slower than with 16 threads
    # (2.0 vs 0.45 s per 800 x 1088 frame; the parity sweep took 290 of the suite's 550 s).
    try:
        import torch
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 16, torch.get_num_threads() or 16)))
    except Exception:
        pass


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
