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


def test_every_kernel_with_counted_waits_is_spill_guarded():
    """A register spill inside a kernel that requests operands with LDS-DMA / asm loads and retires them with hand-counted
    `s_waitcnt vmcnt(n)` stores garbage (scratch traffic shares vmcnt).  Every __global__ kernel of a source file that issues
    such loads must be matched by build.NO_SPILL_KERNELS, and the recorded resource usage of every matched kernel of the built
    library must show no VGPR spill and no scratch."""
    import re
    pat = re.compile(r"buffer_load_lds|global_load_lds|s_waitcnt vmcnt")
    unguarded = []
    for src in sorted(build.CSRC.glob("*.hip")) + sorted(build.CSRC.glob("*.h")):
        text = src.read_text()
        if not pat.search(text):
            continue
        for m in re.finditer(r"__global__[^;{]*?\bvoid\s+(\w+)\s*\(", text, flags=re.S):
            name = m.group(1)
            if name.startswith("splitk_reduce"):      # plain compiler-scheduled loads
                continue
            if not any(k in name for k in build.NO_SPILL_KERNELS):
                unguarded.append(f"{src.name}:{name}")
    assert not unguarded, f"kernels with counted waits outside NO_SPILL_KERNELS: {unguarded}"
    res = sorted((build.CSRC / "build").glob("*.resources.txt"))
    if not res:
        pytest.skip("library not built yet")
    seen = 0
    for f in res:
        for line in f.read_text().splitlines():
            if any(k in line for k in build.NO_SPILL_KERNELS):
                seen += 1
                assert " vgpr_spill 0 " in line and " scratch 0 " in line, line
    assert seen >= 10
    assert any("deepk" in l for f in res for l in f.read_text().splitlines() if any(k in l for k in build.NO_SPILL_KERNELS))
