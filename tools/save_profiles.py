"""Condense rocprofv3 outputs under gpurun_out/ into small tracked files under profiles/.

usage: python tools/save_profiles.py <tag> <stats_dir> [<fetch_dir> <write_dir> [<bench_log>]]
  <stats_dir>  output of `rocprofv3 --kernel-trace --stats --output-format csv -d ...`
  <fetch_dir> / <write_dir>  outputs of the two separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes
  <bench_log>  stdout of the bench.py run of the same collection: its roofline object (dominant kernel,
               launches/step, GFLOP/launch) and the git commit are stamped into the traffic file, and bench.py only
               pairs the traffic figure with a run that has the same three numbers
Writes profiles/<tag>_kernel_stats.csv and profiles/<tag>_traffic.json (+ profiles/traffic_latest.json,
which bench.py reads to fill roofline.traffic for the dominant kernel).
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KB and on gfx950
FETCH_SIZE reports exactly half of a wide (16 B/lane) coalesced read stream (MI355X_MICROARCH.md, HBM).
"""
import collections
import csv
import glob
import json
import shutil
import subprocess
import sys
from pathlib import Path

R = Path(__file__).resolve().parent.parent
tag, stats_dir = sys.argv[1], sys.argv[2]
out = R / "profiles"
out.mkdir(exist_ok=True)
f = glob.glob(f"{stats_dir}/*/*kernel_stats.csv")[0]
shutil.copy(f, out / f"{tag}_kernel_stats.csv")
print("saved", out / f"{tag}_kernel_stats.csv")
if len(sys.argv) >= 5:
    def load(d, name):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(glob.glob(f"{d}/*/*counter_collection.csv")[0])):
            if r["Counter_Name"] == name:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        return acc
    rd, wr = load(sys.argv[3], "FETCH_SIZE"), load(sys.argv[4], "WRITE_SIZE")
    res = {}
    for k, v in rd.items():
        w = wr.get(k, [0.0])
        fetch, write = sum(v) / len(v), sum(w) / len(w)
        res[k] = {"launches": len(v), "FETCH_SIZE_KB_avg": round(fetch, 1), "WRITE_SIZE_KB_avg": round(write, 1),
                  "hbm_bytes_per_launch": int((2 * fetch + write) * 1024)}
    res = dict(sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))
    payload = {"tag": tag, "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes; separate --pmc passes", "kernels": res}
    try:
        payload["commit"] = subprocess.run(["git", "-C", str(R), "rev-parse", "--short", "HEAD"], capture_output=True,
                                           text=True, check=True).stdout.strip()
    except Exception:
        payload["commit"] = None
    import hashlib
    payload["kernel_source_sha16"] = hashlib.sha256(
        (R / "handnet-pipeline_amd" / "csrc" / "conv_igemm_f16x3_kernel.h").read_bytes()).hexdigest()[:16]
    if len(sys.argv) >= 6:
        line = [l for l in open(sys.argv[5]).read().splitlines() if l.startswith("{")][-1]
        roof = json.loads(line)["roofline"]
        payload["bench"] = {k: roof[k] for k in ("kernel", "launches_per_step", "gflop_per_launch", "avg_launch_us")}
    for name in (f"{tag}_traffic.json", "traffic_latest.json"):
        (out / name).write_text(json.dumps(payload, indent=1))
    print("saved", out / f"{tag}_traffic.json")
