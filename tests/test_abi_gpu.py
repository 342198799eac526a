"""The C ABI used from a host that is not Python (examples/abi_smoke.cpp: hipMalloc'd buffers, plain pointers)."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def test_cpp_host_runs_conv_and_nms_through_the_c_abi():
    sys.path.insert(0, str(REPO / "handnet-pipeline_amd"))
    from hn_amd import build
    build.build_library()
    exe = build.build_abi_example()
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)   # a child process, not an exec
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "abi_smoke ok" in r.stdout
