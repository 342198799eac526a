"""Build libhandnet_hip.so (gfx950) in-tree with hipcc.

`python -m hn_amd.build` or `hn_amd.build.build_library()`.  Objects are rebuilt only
when their source (or a header) is newer; the shared library lands next to the sources
(`csrc/libhandnet_hip.so`) so that it travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_ROOT = Path(__file__).resolve().parent.parent          # handnet-pipeline_amd/
CSRC = PKG_ROOT / "csrc"
REPO_ROOT = PKG_ROOT.parent
LIB_PATH = CSRC / "libhandnet_hip.so"
ARCH = "gfx950"

# decode / IoU arithmetic must round exactly like the reference's separate torch ops
# (no FMA contraction), so that file is built with contraction off.
#
# The same file is built without the SLP vectoriser: it packed the bilinear blend of the tiled preprocess kernel into
# v_pk_mul_f32 / v_pk_fma_f32 with op_sel:[1,0] (low result from the HIGH register of a source pair), and exactly those
# results -- lanes 48-63, channel 0 of the odd column, nothing else -- came out wrong on ~3 % of the waves whenever a second
# process had work on the same card (tests/test_dist_gpu.py, tools/diag/preprocess_pattern.py, profiles/NOTEBOOK.md round 4);
# alone on the card the same binary is bit-exact.  No other kernel of the library holds such an instruction, and the build
# refuses one (_check_packed_opsel).
# graph_ops.hip: the matrix-vector Linear's eight-wide fp32 FMA chains were packed the same way (round 6; the build refused it).
EXTRA_FLAGS = {"fcos_post.hip": ["-ffp-contract=off", "-fno-slp-vectorize"], "graph_ops.hip": ["-fno-slp-vectorize"]}

# Kernels that request operands with `asm volatile` loads / LDS-DMA and retire them with hand-counted s_waitcnt: the
# compiler cannot see that such a register is still in flight, so a SPILL of it stores garbage (profiles/NOTEBOOK.md, round
# 3).  The build records every kernel's resource usage (csrc/build/<file>.resources.txt) and refuses a spill in these.
# Matched as substrings of the (mangled) kernel name: "conv_igemm_f16x3_" covers every instantiation of the implicit GEMM family
# (plain, deep-k, multi, mixed -- the deep-k kernel's name once slipped through per-kernel entries), the others are the
# remaining files with LDS-DMA / counted waits (tests/test_build_cpu.py keeps this list in step with the sources).
NO_SPILL_KERNELS = ("conv_igemm_f16x3_", "conv3x3_halo_kernel", "conv_stem_pool_direct_kernel", "conv3x3_thin", "conv1x1_stream_kernel")


def _check_resources(src: Path, remarks: str, objdir: Path) -> None:
    name, rows = None, []
    for line in remarks.splitlines():
        if "remark:" not in line:
            continue
        body = line.split("remark:", 1)[1].split("[-Rpass", 1)[0].strip()
        if body.startswith("Function Name:"):
            name = body.split(":", 1)[1].strip()
            rows.append([name, {}])
        elif name and ":" in body:
            k, v = body.rsplit(":", 1)
            rows[-1][1][k.strip()] = v.strip()
    out = []
    for name, res in rows:
        out.append(f"{name}: VGPRs {res.get('VGPRs')} AGPRs {res.get('AGPRs')} SGPRs {res.get('TotalSGPRs')} "
                   f"scratch {res.get('ScratchSize [bytes/lane]')} occupancy {res.get('Occupancy [waves/SIMD]')} "
                   f"vgpr_spill {res.get('VGPRs Spill')} sgpr_spill {res.get('SGPRs Spill')}")
        # (SGPR spills go to VGPR lanes, not to memory: harmless where no asm load is in flight -- the P-form kernel uses
        # plain loads only -- but they are refused in the hand-counted kernels all the same)
        # (... and in the mixed grouped kernel, whose two bodies' scalar preambles do not fit 102 SGPRs together)
        sgpr_ok = res.get("SGPRs Spill", "0") == "0" or "thin_flat" in name or "mixed_kernel" in name
        if any(k in name for k in NO_SPILL_KERNELS) and (res.get("VGPRs Spill", "0") != "0" or not sgpr_ok):
            raise RuntimeError(f"{src.name}: {name} spills registers ({res.get('VGPRs Spill')} VGPR, {res.get('SGPRs Spill')} SGPR): "
                               "its asm loads / counted waits are only correct without spills")
        # (scratch without a spill: a by-value kernel argument whose address escaped -- e.g. handed to a device function by
        # reference -- is copied to private memory and every use becomes a scratch load; seen once, on the thin kernel)
        if any(k in name for k in NO_SPILL_KERNELS) and res.get("ScratchSize [bytes/lane]", "0") != "0":
            raise RuntimeError(f"{src.name}: {name} uses {res.get('ScratchSize [bytes/lane]')} bytes of scratch per lane "
                               "(a kernel argument copied to private memory?): these kernels are written to run from registers")
    (objdir / (src.stem + ".resources.txt")).write_text("\n".join(out) + "\n")


def _llvm_tool(name: str):
    for root in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if root and (Path(root) / "lib" / "llvm" / "bin" / name).exists():
            return str(Path(root) / "lib" / "llvm" / "bin" / name)
    return shutil.which(name)


def _check_packed_opsel(obj: Path, objdir: Path) -> None:
    """Refuse packed-fp32 VALU instructions whose op_sel selects the high source register for the low result (see
    EXTRA_FLAGS): disassemble the gfx950 code object of `obj` and look for them."""
    objcopy, bundler, objdump = (_llvm_tool(n) for n in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump"))
    if not (objcopy and bundler and objdump):
        # a toolchain without the LLVM binutils cannot prove the absence of the instruction: the build FAILS unless the
        # builder explicitly accepts an unchecked library (which then must not share a card with another process)
        what = (f"llvm-objcopy / clang-offload-bundler / llvm-objdump not found: {obj.name} cannot be checked for packed-fp32 "
                "op_sel instructions (wrong results when another process shares the card; DESIGN.md section 5)")
        if os.environ.get("HN_ALLOW_UNCHECKED_BUILD") != "1":
            raise RuntimeError(what + "; set HN_ALLOW_UNCHECKED_BUILD=1 to build without the check")
        print(f"[build] warning: {what}: HN_ALLOW_UNCHECKED_BUILD=1, building unchecked", file=sys.stderr)
        return
    fat, co = objdir / (obj.stem + ".fatbin"), objdir / (obj.stem + ".co")
    try:
        subprocess.run([objcopy, "-O", "binary", "--only-section=.hip_fatbin", str(obj), str(fat)], check=True, capture_output=True)
        if not fat.exists() or fat.stat().st_size == 0:
            return                                   # no device code in this object
        subprocess.run([bundler, "--type=o", f"--targets=hipv4-amdgcn-amd-amdhsa--{ARCH}", f"--input={fat}", f"--output={co}",
                        "--unbundle"], check=True, capture_output=True)
        dis = subprocess.run([objdump, "-d", str(co)], check=True, capture_output=True, text=True).stdout
    finally:
        fat.unlink(missing_ok=True)
        co.unlink(missing_ok=True)
    bad = [l.strip() for l in dis.splitlines() if re.search(r"v_pk_\w+_f32\b.*\bop_sel:\[[0-9,]*1", l)]
    if bad:
        raise RuntimeError(f"{obj.name}: {len(bad)} packed-fp32 instruction(s) with op_sel (unsafe on a shared card, see "
                           f"hn_amd/build.py EXTRA_FLAGS), e.g. `{bad[0]}`")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _newer(src_paths, target: Path) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(p.stat().st_mtime > t for p in src_paths)


def build_library(force: bool = False, verbose: bool = False) -> Path:
    hipcc = _hipcc()
    sources = sorted(CSRC.glob("*.hip"))
    headers = sorted(CSRC.glob("*.h")) + sorted((REPO_ROOT / "include").glob("*.h"))
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    base = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
            "-Wall", "-Wno-unused-function", f"-I{REPO_ROOT / 'include'}"]

    def compile_one(src: Path) -> Path:
        obj = objdir / (src.stem + ".o")
        if force or _newer([src] + headers, obj) or not (objdir / (src.stem + ".resources.txt")).exists():
            cmd = base + EXTRA_FLAGS.get(src.name, []) + ["-Rpass-analysis=kernel-resource-usage", "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src.name}:\n{r.stdout}\n{r.stderr}")
            try:
                _check_resources(src, r.stderr, objdir)
                _check_packed_opsel(obj, objdir)
            except RuntimeError:
                obj.unlink(missing_ok=True)   # do not leave an object behind that a later incremental build would link
                raise
            if verbose and r.stderr.strip():
                print("\n".join(l for l in r.stderr.splitlines() if "kernel-resource-usage" not in l), file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, len(sources) or 1)) as ex:
        objs = list(ex.map(compile_one, sources))
    # relink also when the SET of objects changed (a source file added or removed leaves every mtime older than the library)
    manifest = objdir / "linked_objects.txt"
    wanted = "\n".join(o.name for o in objs)
    if force or _newer(objs, LIB_PATH) or not manifest.exists() or manifest.read_text() != wanted:
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB_PATH)] + [str(o) for o in objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        manifest.write_text(wanted)
    return LIB_PATH


def build_abi_example(force: bool = False) -> Path:
    """examples/abi_smoke.cpp: a C++ host bound to include/handnet_hip.h only (no Python, no torch)."""
    src = REPO_ROOT / "examples" / "abi_smoke.cpp"
    exe = CSRC / "build" / "abi_smoke"
    if force or _newer([src, REPO_ROOT / "include" / "handnet_hip.h", LIB_PATH], exe):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O2", "-std=c++17", f"-I{REPO_ROOT / 'include'}", str(src),
               f"-L{CSRC}", "-lhandnet_hip", "-Wl,-rpath,$ORIGIN/..", "-o", str(exe)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on abi_smoke.cpp:\n{r.stdout}\n{r.stderr}")
    return exe


if __name__ == "__main__":
    p = build_library(force="--force" in sys.argv, verbose=True)
    print(p)
    print(build_abi_example(force="--force" in sys.argv))
