"""The build's device-code rule (hn_amd/build.py): packed-fp32 instructions with an op_sel bit are refused -- on gfx950 the forms
with op_sel on the second source return wrong values in lanes 48-63 whenever other work shares the card (profiles/NOTEBOOK.md,
round 4; tools/probes/pk_opsel_probe.hip).  hipcc cross-compiles without a GPU, so this runs on the CPU host."""
import shutil
import subprocess

import pytest

from hn_amd import build

SRC = r"""
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const f32x2* a, f32x2* o) {
  f32x2 x = a[threadIdx.x], y = a[threadIdx.x + 64], d;
  asm volatile("%s" : "=v"(d) : "v"(x), "v"(y));
  o[threadIdx.x] = d;
}
"""


def _compile(tmp_path, name, asm):
    src = tmp_path / f"{name}.hip"
    src.write_text(SRC % asm)
    obj = tmp_path / f"{name}.o"
    subprocess.run([build._hipcc(), f"--offload-arch={build.ARCH}", "-O2", "-c", str(src), "-o", str(obj)], check=True,
                   capture_output=True)
    return obj


@pytest.mark.skipif(shutil.which("hipcc") is None and not shutil.os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_build_refuses_packed_fp32_op_sel(tmp_path):
    bad = _compile(tmp_path, "bad", "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]")
    with pytest.raises(RuntimeError, match="op_sel"):
        build._check_packed_opsel(bad, tmp_path)
    good = _compile(tmp_path, "good", "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]")
    build._check_packed_opsel(good, tmp_path)          # op_sel_hi forms are exact (probe) and allowed
    plain = _compile(tmp_path, "plain", "v_pk_fma_f32 %0, %1, %2, %1")
    build._check_packed_opsel(plain, tmp_path)


def test_library_objects_hold_no_such_instruction():
    """Every object of the built library passes the rule (they are checked when they are compiled; this re-checks what is linked)."""
    objdir = build.CSRC / "build"
    objs = sorted(objdir.glob("*.o"))
    if not objs:
        pytest.skip("library not built yet")
    for o in objs:
        build._check_packed_opsel(o, objdir)
