"""CPU tests: the C-ABI library builds for gfx950, loads, and exports exactly the symbols
that include/handnet_hip.h declares (no compute calls without a GPU)."""
import re
import subprocess

from hn_amd import _lib, build


def _declared_symbols():
    text = (build.REPO_ROOT / "include" / "handnet_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hn_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_loads():
    path = build.build_library()
    assert path.exists()
    lib = _lib.load()
    assert lib.hn_abi_version() == _lib.ABI_VERSION


def test_every_declared_symbol_is_exported_and_bound():
    declared = _declared_symbols()
    assert len(declared) >= 20
    out = subprocess.run(["nm", "-D", "--defined-only", str(_lib.lib_path())], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (hn_[a-z0-9_]+)", out))
    missing = [s for s in declared if s not in exported]
    assert not missing, f"declared in handnet_hip.h but not exported: {missing}"
    unbound = [s for s in declared if s not in _lib.SIGNATURES]
    assert not unbound, f"declared but not bound in hn_amd/_lib.py: {unbound}"
    extra = [s for s in _lib.SIGNATURES if s not in declared]
    assert not extra, f"bound but not declared in the header: {extra}"


def test_conv_desc_struct_matches_header():
    text = (build.REPO_ROOT / "include" / "handnet_hip.h").read_text()
    body = re.search(r"typedef struct hn_conv_desc \{(.*?)\} hn_conv_desc;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in re.findall(r"int32_t\s+([^;]+);", body):
        fields += [f.strip() for f in decl.split(",")]
    assert fields == [f for f, _ in _lib.ConvDesc._fields_]


def test_every_struct_of_the_header_has_the_ctypes_layout(tmp_path):
    """The by-value / by-pointer structs of include/handnet_hip.h against their ctypes mirrors in hn_amd/_lib.py: a C probe
    compiled with gcc against the header prints sizeof and every field's offsetof; ctypes must agree field by field (a field
    added on one side only -- hn_convert_opts grew two pointers in ABI v34, hn_thin_member is new in v35 -- shifts everything
    behind it silently otherwise).  Every `typedef struct ... {` of the header must be in the table."""
    import shutil
    import subprocess
    pairs = {"hn_conv_desc": _lib.ConvDesc, "hn_conv_group": _lib.ConvGroup, "hn_conv_multi": _lib.ConvMulti,
             "hn_gn_levels": _lib.GnLevels, "hn_split_levels": _lib.SplitLevels, "hn_thin_levels": _lib.ThinLevels,
             "hn_thin_member": _lib.ThinMember, "hn_thin_affine": _lib.ThinAffine, "hn_fcos_levels": _lib.FcosLevels,
             "hn_convert_opts": _lib.ConvertOpts, "hn_graph_csr": _lib.GraphCsr, "hn_model_config": _lib.ModelConfig}
    header = build.REPO_ROOT / "include" / "handnet_hip.h"
    declared = set(re.findall(r"typedef struct (\w+) \{", header.read_text()))
    assert declared == set(pairs), declared ^ set(pairs)
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    lines = ["#include <stdio.h>", "#include <stddef.h>", f'#include "{header}"', "int main(void) {"]
    for name, cls in pairs.items():
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for field, _ in cls._fields_:
            lines.append(f'  printf("{name}.{field} %zu\\n", offsetof({name}, {field}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "layout"
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-o", str(exe), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr     # (a field of the ctypes class that the header lacks fails HERE, by name)
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for name, cls in pairs.items():
        assert int(got[name]) == C_sizeof(cls), (name, got[name], C_sizeof(cls))
        for field, _ in cls._fields_:
            assert int(got[f"{name}.{field}"]) == getattr(cls, field).offset, (name, field)
        # ... and a field the header has but ctypes lacks shows as a size difference above, unless it hides in tail padding:
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header.read_text(), flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        n_fields = sum(len(d.split(",")) for d in re.findall(r"([^;{}]+);", body))     # declarators per declaration
        assert n_fields == len(cls._fields_), (name, n_fields, len(cls._fields_))


def C_sizeof(cls):
    import ctypes
    return ctypes.sizeof(cls)


def test_argument_errors_do_not_need_a_gpu():
    import ctypes as C
    lib = _lib.load()
    d = _lib.ConvDesc(n=1, h=8, w=8, cin=6, cout=8, r=3, s=3, stride=1, pad=1, dil=1, oh=8, ow=8)
    st = lib.hn_conv2d_nhwc_f32(C.byref(d), 1, 1, None, None, None, None, 1, None)
    assert st == 1 and b"multiple of 4" in lib.hn_last_error()
    assert lib.hn_fcos_nms_scratch_bytes(2, 1000) > 2 * 1024 * 8
    assert lib.hn_groupnorm_scratch_floats(2, 13600, 256, 32) == 2 * 213 * 32 * 2


def test_probes_compile():
    """tools/probes/*.hip are measurement aids quoted in DESIGN.md / profiles/: keep them compiling for gfx950 (compile
    only; they run on the GPU box by hand)."""
    from concurrent.futures import ThreadPoolExecutor
    probes = sorted((build.REPO_ROOT / "tools" / "probes").glob("*.hip"))
    assert len(probes) >= 5

    def compile_one(src):
        r = subprocess.run([build._hipcc(), f"--offload-arch={build.ARCH}", "-O2", "-c", str(src), "-o", "/dev/null"],
                           capture_output=True, text=True)
        return src.name, r.returncode, r.stderr[-400:]
    with ThreadPoolExecutor(max_workers=4) as ex:
        for name, rc, err in ex.map(compile_one, probes):
            assert rc == 0, f"{name}: {err}"


def test_kernel_form_queries_follow_the_launcher():
    """hn_conv2d_f16x3_uses_rs / _uses_halo / _pick_tile are host-only and evaluate the launcher's own planning functions:
    the dominant tower launch (grouped: narrowest member width in d.w, picked tile in d.tile) runs the row-shared-A
    kernel; a small-grid layer that the launcher sends through split-K does not; ResNet-34 layer1 goes to the halo kernel."""
    import ctypes as C
    lib = _lib.load()

    def desc(**kw):
        base = dict(n=32, h=100, w=136, cin=256, cout=256, r=3, s=3, stride=1, pad=1, dil=1, oh=100, ow=136, relu_cols=0)
        base.update(kw)
        return _lib.ConvDesc(**base)
    assert lib.hn_conv2d_f16x3_uses_rs(C.byref(desc())) == 1                                    # plain tower-shaped layer
    # grouped launch as ops.conv2d_nhwc_grouped describes it: first member's geometry, narrowest width, picked tile, no split-K
    assert lib.hn_conv2d_f16x3_uses_rs(C.byref(desc(w=34, tile=1, splitk=-1))) == 1
    assert lib.hn_conv2d_f16x3_uses_rs(C.byref(desc(r=1, s=1, pad=0))) == 0                       # 1x1: per-tap form
    assert lib.hn_conv2d_f16x3_uses_rs(C.byref(desc(stride=2, oh=50, ow=68))) == 0
    small = desc(n=1, h=11, w=11, oh=11, ow=11, splitk=1)                                         # batch-1 A2J layer: split-K
    assert lib.hn_conv2d_f16x3_uses_rs(C.byref(small)) == 0
    small.splitk = -1                                                                             # ... unless the caller forbids it
    assert lib.hn_conv2d_f16x3_uses_rs(C.byref(small)) == (1 if lib.hn_conv2d_f16x3_pick_tile(C.byref(small)) in (1, 2, 4) else 0)
    l1 = desc(h=200, w=272, oh=200, ow=272, cin=64, cout=64, relu_cols=64, out_split=1)
    assert lib.hn_conv2d_f16x3_uses_halo(C.byref(l1), 0) == 1 and lib.hn_conv2d_f16x3_uses_rs(C.byref(l1)) == 0
    l1.n = 1
    assert lib.hn_conv2d_f16x3_uses_halo(C.byref(l1), 0) == 0                                     # too few tiles: implicit GEMM


def test_probe_scripts_parse():
    """tools/ and tools/probes/exp/ hold the A/B drivers behind the numbers in DESIGN.md / profiles/NOTEBOOK.md: keep them
    at least syntactically alive (ast.parse / bash -n; they run on the GPU box by hand)."""
    import ast
    root = build.REPO_ROOT / "tools"
    py = sorted(root.rglob("*.py"))
    sh = sorted(root.rglob("*.sh"))
    assert len(py) >= 10 and len(sh) >= 10
    for f in py:
        ast.parse(f.read_text(), filename=str(f))
    for f in sh:
        r = subprocess.run(["bash", "-n", str(f)], capture_output=True, text=True)
        assert r.returncode == 0, f"{f}: {r.stderr}"


def test_hand_scheduled_kernels_do_not_spill():
    """The kernels that retire `asm` loads / LDS-DMA with hand-counted s_waitcnt are only correct while the register
    allocator spills nothing (a spilled in-flight register is stored before its data has arrived): the build records every
    kernel's resource usage and refuses such a spill; check the record it left."""
    from hn_amd import build
    build.build_library()
    seen = 0
    for f in sorted((build.CSRC / "build").glob("*.resources.txt")):
        for line in f.read_text().splitlines():
            if ": VGPRs" not in line:
                continue
            name, rest = line.split(":", 1)
            if any(k in name for k in build.NO_SPILL_KERNELS):
                seen += 1
                assert " vgpr_spill 0 " in rest + " " and " scratch 0 " in rest, line
                # SGPR spills go to VGPR lanes, not to memory: tolerated in the P-form head kernel (plain loads only) and in the
                # mixed grouped kernel (two bodies' scalar preambles in one kernel); never a VGPR spill or scratch anywhere
                assert rest.strip().endswith("sgpr_spill 0") or "thin_flat" in name or "mixed_kernel" in name, line
    assert seen >= 30    # the instantiations of the implicit GEMM (single, mixed, multi) + halo + stem + the two thin-N kernels


def test_documents_cite_files_that_exist():
    """DESIGN.md / INTEGRATION.md / README.md quote measurements by file name (`profiles/r03e_*.json`, `tools/...`, `csrc/...`):
    every back-quoted path with a known prefix or a profile-file suffix must exist, so a renamed collection cannot leave the
    documents pointing at nothing."""
    import re
    root = build.REPO_ROOT
    prof = {p.name for p in (root / "profiles").iterdir()}
    missing = []
    for doc in ("DESIGN.md", "INTEGRATION.md", "README.md"):
        text = (root / doc).read_text()
        for tok in re.findall(r"`([^`\s]+)`", text):
            tok = tok.rstrip(".,;:)")
            if "*" in tok or "<" in tok or "(" in tok:
                continue
            if tok.startswith(("profiles/", "tools/", "tests/", "oracle/", "include/", "examples/")):
                if not (root / tok.split("::")[0].split(":")[0]).exists():
                    missing.append((doc, tok))
            elif re.fullmatch(r"r0\d[a-z]?_[\w.]+\.(json|txt|csv)", tok):
                if tok not in prof:
                    missing.append((doc, tok))
            elif tok.startswith("csrc/") or tok.startswith("hn_amd/"):
                if not (root / "handnet-pipeline_amd" / tok.split(":")[0]).exists():
                    missing.append((doc, tok))
    assert not missing, missing


def test_documents_do_not_contradict_the_product():
    """Statements that were once true and stayed in a document after the product changed: a non-finite
    RGB pixel RAISES (it does not give "undefined detections"), and the batch-1 drop-in figures are the driver-timed ones."""
    root = build.REPO_ROOT
    integ = (root / "INTEGRATION.md").read_text()
    design = (root / "DESIGN.md").read_text()
    msg = (root / "handnet-pipeline_amd" / "hn_amd" / "pipeline.py").read_text()
    assert "non-finite RGB pixels give undefined detections" not in integ
    assert "non-finite RGB pixels give undefined detections" not in msg
    assert "2.55–2.6 ms per call" not in integ
    assert "2.32–2.36 ms per call at batch 1**" not in design


def test_range_scope_is_host_thread_state():
    """hn_range_scope_begin / _end (host-only): nesting restores the previous switch, another thread never sees this one's."""
    import threading
    from hn_amd import _lib
    lib = _lib.load()
    assert lib.hn_range_check_enabled() == 0
    assert lib.hn_range_scope_end() != 0 and b"no open scope" in lib.hn_last_error()
    assert lib.hn_range_scope_begin(None, 1) == 0 and lib.hn_range_check_enabled() == 1
    assert lib.hn_range_scope_begin(None, 0) == 0 and lib.hn_range_check_enabled() == 0
    other = []
    t = threading.Thread(target=lambda: other.append(lib.hn_range_check_enabled()))
    t.start(); t.join()
    assert other == [0]
    assert lib.hn_range_scope_end() == 0 and lib.hn_range_check_enabled() == 1
    t = threading.Thread(target=lambda: other.append(lib.hn_range_check_enabled()))
    t.start(); t.join()
    assert other == [0, 0]
    assert lib.hn_range_scope_end() == 0 and lib.hn_range_check_enabled() == 0
    for _ in range(8):
        assert lib.hn_range_scope_begin(None, 1) == 0
    assert lib.hn_range_scope_begin(None, 1) != 0            # ninth level: refused, state unchanged
    for _ in range(8):
        assert lib.hn_range_scope_end() == 0
    assert lib.hn_range_check_enabled() == 0
