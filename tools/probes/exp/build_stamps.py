"""Build tools/probes/exp/lib_stamps.so: the product library with the f16x3 conv kernel instrumented by s_memtime stamps
(diagnostic build only; selected with HN_LIB_PATH by stamps.py / residency.py, never shipped).  Per workgroup (first
8192 of a launch): [0] entry, [1] index math done, [2] first operand tile landed, [3] fragments loaded, [4] k loop done,
[5] epilogue issued, [6] stores retired, [7] (XCC_ID << 32) | HW_ID."""
import subprocess, sys
from pathlib import Path
R = Path(__file__).resolve().parents[3]
src = (R / "handnet-pipeline_amd/csrc/conv_igemm_f16x3.hip").read_text()
src = src.replace('#include "hn_common.h"', f'#include "{R}/handnet-pipeline_amd/csrc/hn_common.h"')
def rep(a, b):
    global src
    assert a in src, a[:60]
    src = src.replace(a, b, 1)
rep('namespace {\n\ntypedef float f32x4 __attribute__((ext_vector_type(4)));', '''__device__ unsigned long long g_stamps[8 * 65536];
extern "C" int hn_debug_read_stamps(unsigned long long* host, int count) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * count) == hipSuccess ? 0 : 1;
}
#define STAMPOK (threadIdx.x == 0 && blockIdx.x < 8192 && blockIdx.z == 0 && blockIdx.y == 0)
#define STAMP(i) do { if (STAMPOK) g_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));''')
rep('  constexpr int NT = WM * WN * 64;\n  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;', '''  STAMP(0);
  if (STAMPOK) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_stamps[blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | hw;
  }
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;''')
rep('  dma_tile(0);\n  drain_and_barrier();', '  STAMP(1);\n  dma_tile(0);\n  drain_and_barrier();\n  STAMP(2);')
rep('  int cs = 0, ns = 1;\n  for (int t = 0; t < T; ++t) {', '  int cs = 0, ns = 1;\n  STAMP(3);\n  for (int t = 0; t < T; ++t) {')
rep('  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant tail loads must land before LDS is reused / freed\n',
    '  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant tail loads must land before LDS is reused / freed\n  STAMP(4);\n')
rep('    return;\n  }\n  // Scalar path',
    '    STAMP(5);\n    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n    STAMP(6);\n    return;\n  }\n  // Scalar path')
import os
if os.environ.get("STAMPS_UNSWAPPED"):   # timing-only A/B: the round-2 operand order under the v8 epilogue (results are wrong)
    rep('"v_mfma_f32_16x16x32_f16 %0, %2, %1, %0"', '"v_mfma_f32_16x16x32_f16 %0, %1, %2, %0"')
Path("/tmp/hn_stamps.hip").write_text(src)
cc = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", f"-I{R}/include"]
subprocess.run(cc + ["-c", "/tmp/hn_stamps.hip", "-o", "/tmp/hn_stamps.o"], check=True)
objs = [str(o) for o in (R / "handnet-pipeline_amd/csrc/build").glob("*.o") if o.name != "conv_igemm_f16x3.o"]
subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(R / "tools/probes/exp/lib_stamps.so"), "/tmp/hn_stamps.o"] + objs, check=True)
print("built tools/probes/exp/lib_stamps.so")
