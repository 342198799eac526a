"""The C ABI used from a host that is not Python (examples/abi_smoke.cpp: hipMalloc'd buffers, plain pointers)."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def _write_blob(path, sd, crops, expect):
    """weights + seeded crops + the reference's keypoints for them, in the flat format examples/abi_smoke.cpp reads"""
    import struct
    tensors = [(k, v) for k, v in sd.items() if torch.is_floating_point(v)]
    with open(path, "wb") as f:
        f.write(b"HNB1" + struct.pack("<i", len(tensors)))
        for name, t in tensors:
            t = t.detach().float().contiguous()
            f.write(struct.pack("<i", len(name)) + name.encode() + struct.pack("<i", t.dim()))
            f.write(struct.pack(f"<{t.dim()}q", *t.shape))
            f.write(t.numpy().tobytes())
        k, _, h, w = crops.shape
        f.write(struct.pack("<iii", k, h, w) + crops.float().contiguous().numpy().tobytes())
        f.write(struct.pack("<i", expect.shape[1]) + np.ascontiguousarray(expect, dtype=np.float32).tobytes())


def test_cpp_host_runs_ops_and_a_whole_a2j_forward_through_the_c_abi(tmp_path, a2j_sd, golden_dir):
    """Op-level calls vs host loops, then the MODEL-level ABI: the C++ layer graph loads a reference-layout
    state_dict by name, folds / packs it, runs A2J and matches the keypoints the imported reference produced."""
    sys.path.insert(0, str(REPO / "handnet-pipeline_amd"))
    from hn_amd import build, synth
    build.build_library()
    exe = build.build_abi_example()
    g = np.load(golden_dir / "a2j_forward.npz")
    crops = synth.make_crops(2, 176, seed=int(g["input_seed"]))
    blob = tmp_path / "a2j_model.hnb"
    _write_blob(blob, a2j_sd, crops, g["keypoints"])
    r = subprocess.run([str(exe), str(blob)], capture_output=True, text=True, timeout=300)   # a child process, not an exec
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "abi_smoke ok" in r.stdout and "a2j model forward" in r.stdout
