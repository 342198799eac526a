"""Development switches: older kernel forms and launch structures kept as bit-identity references (tests) and for same-box
A/B timing (tools/, bench.py).

The PRODUCT never reads them from the environment -- the engines and the drop-in modules read only HN_LIB_PATH (another build of
the library), HN_CHECK_RANGE, HN_COMPACT_SPARSE and HN_AUTO_GRAPH.  Two layers:
  * library forms (`ops.set_form(name, on)` -> hn_set_form): which kernel a launch takes;
  * engine forms (this module's ACTIVE dict, read by FCOSEngine / A2JEngine at construction; `forms=` overrides per engine):
    how the layer graph is issued.
`apply_env()` translates the HN_* variables the A/B scripts under tools/ export; bench.py and the tools call it, tests set
what they need explicitly.
"""
from __future__ import annotations

import os

ENGINE_DEFAULTS = dict(
    head_streams=1,         # > 1: the FCOS tower chains on side streams (slower on this platform; kept as a measurement)
    group_convs=True,       # same-shape convolutions of a layer as one grouped launch (towers, A2J heads)
    fuse_stem_pool=True,    # conv1 + bn1 + relu + maxpool as one kernel
    fuse_last_gn=True,      # the towers' last GroupNorm apply pass inside the head-output kernel
    thin_outputs=True,      # thin-N kernels for the <= 16-channel head outputs
    conv_multi=True,        # heterogeneous launches in the A2J engine (<= MULTI_MAX_CROPS crops)
    conv_multi_fcos=True,   # ... for the downsample blocks of the FCOS trunk
)
ACTIVE = dict(ENGINE_DEFAULTS)

# HN_* variable -> (library form, value that turns the form ON)
_LIB_ENV = {
    "HN_CONV_NO_RS": ("conv_no_rs", None), "HN_CONV_NO_RS32": ("conv_no_rs32", None),
    "HN_SPLIT_GENERIC": ("split_generic", None),
    "HN_CONV_NO_HALO": ("conv_no_halo", None), "HN_PREPROCESS_GENERIC": ("preprocess_generic", None),
    "HN_CONV_NO_MULTI": ("conv_no_multi", None), "HN_HALO_STAMPS": ("halo_stamps", None),
    "HN_SPLITK_FILL512": ("splitk_fill512", None), "HN_CONV_NO_STREAM": ("conv_no_stream", None),
    "HN_CONV_NO_MIXED": ("conv_no_mixed", None), "HN_CONV_NO_DEEPK": ("conv_no_deepk", None),
    "HN_CONV_NO_FUSED_REDUCE": ("conv_no_fused_reduce", None), "HN_THIN_NO_GROUP": ("thin_no_group", None),
}


def engine_forms(overrides=None) -> dict:
    f = dict(ACTIVE)
    if overrides:
        unknown = set(overrides) - set(ENGINE_DEFAULTS)
        if unknown:
            raise KeyError(f"unknown engine forms: {sorted(unknown)}")
        f.update(overrides)
    return f


def apply_env(environ=None) -> dict:
    """Development hosts only (bench.py, tools/): translate the HN_* A/B variables into library forms (ops.set_form), ops-level
    switches and the ACTIVE engine forms.  Returns what was set."""
    from . import ops
    env = os.environ if environ is None else environ
    done = {}
    for var, (form, _) in _LIB_ENV.items():
        if var in env:
            ops.set_form(form, True)
            done[var] = form
    if env.get("HN_FUSE_LAST_GN") == "0":
        ops.set_form("no_fuse_last_gn", True)
        ACTIVE["fuse_last_gn"] = False
        done["HN_FUSE_LAST_GN"] = "0"
    if env.get("HN_THIN_OUTPUTS") == "0":
        ops.set_form("no_thin_outputs", True)
        ACTIVE["thin_outputs"] = False
        done["HN_THIN_OUTPUTS"] = "0"
    if env.get("HN_THIN_FORM", "")[:1] in ("t", "f"):
        ops.set_form("thin_form_tap" if env["HN_THIN_FORM"][0] == "t" else "thin_form_flat", True)
        done["HN_THIN_FORM"] = env["HN_THIN_FORM"]
    for var, key in (("HN_GROUP_CONVS", "group_convs"), ("HN_FUSE_STEM_POOL", "fuse_stem_pool"), ("HN_CONV_MULTI", "conv_multi"),
                     ("HN_CONV_MULTI_FCOS", "conv_multi_fcos")):
        if var in env:
            ACTIVE[key] = env[var] != "0"
            done[var] = env[var]
    if "HN_CONV_MULTI" in env and "HN_CONV_MULTI_FCOS" not in env:
        ACTIVE["conv_multi_fcos"] = ACTIVE["conv_multi"]
    if "HN_HEAD_STREAMS" in env:
        ACTIVE["head_streams"] = int(env["HN_HEAD_STREAMS"])
        done["HN_HEAD_STREAMS"] = env["HN_HEAD_STREAMS"]
    if "HN_SPLITK" in env:
        ops.SPLITK = env["HN_SPLITK"] != "0"
        done["HN_SPLITK"] = env["HN_SPLITK"]
    if "HN_SPLITK_EAGER" in env:
        ops.SPLITK_EAGER = env["HN_SPLITK_EAGER"] == "1"
        done["HN_SPLITK_EAGER"] = env["HN_SPLITK_EAGER"]
    for var in env:
        if var.startswith("HN_TUNE_"):           # e.g. HN_TUNE_SPLITK_RED0=1.5 -> hn_set_tuning("splitk_red0", 1.5)
            from . import _lib
            ops.check(_lib.load().hn_set_tuning(var[8:].lower().encode(), float(env[var])), "hn_set_tuning")
            done[var] = env[var]
    if "HN_CANDIDATES_CHUNKED" in env:
        ops.CANDIDATES_CHUNKED = env["HN_CANDIDATES_CHUNKED"] != "0"
        done["HN_CANDIDATES_CHUNKED"] = env["HN_CANDIDATES_CHUNKED"]
    return done
