"""Condenses the rocprofv3 output of tools/profile_gpu.sh into the small files kept under profiles/:

  profiles/<tag>_kernel_stats_graph.csv   per-kernel stats of the default (graph replay) bench command
  profiles/<tag>_kernel_stats_eager.csv   same for eager launches
  profiles/<tag>_pmc.json                 per-kernel mean counters per launch (SQ sets, FETCH/WRITE)
  profiles/traffic.json                   HBM bytes per launch, read by bench.py for roofline.traffic
  profiles/<tag>_bench.json               the bench line printed under the profiler

usage: python tools/summarize_profiles.py gpurun_out/prof_<tag> <tag>
"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
os.makedirs("profiles", exist_ok=True)


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else None


def first(pattern):
    g = sorted(glob.glob(os.path.join(src, pattern), recursive=True))
    return g[0] if g else None


for mode in ("graph", "eager"):
    f = first(f"{mode}/**/*kernel_stats.csv")
    if f:
        shutil.copy(f, f"profiles/{tag}_kernel_stats_{mode}.csv")

pmc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for d in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write"):
    f = first(f"{d}/**/*counter_collection.csv")
    if not f:
        continue
    seen = set()
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if not k:
            continue
        pmc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        if d == "pmc_sq1" and row["Dispatch_Id"] not in seen:
            seen.add(row["Dispatch_Id"])
            dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)

out = {"_note": "rocprofv3 --pmc in separate passes (tools/profile_gpu.sh); mean per launch over the eager bench "
                "run (S1 workload: 1,048,576 splats @1920x1080, SH degree 3); dur_us_profiled is the launch "
                "duration under the counter pass"}
traffic = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (S1 workload, eager launches, "
                    "mean per launch). Units KB. Per MI355X_MICROARCH.md (HBM section), FETCH_SIZE on gfx950 reports half "
                    "the bytes of wide coalesced streaming reads, so hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)"
                    "*1024; WRITE_SIZE is exact for 16-B/lane stores and float atomics. Gather-heavy kernels "
                    "(project_visible, rasterize*) are outside the calibrated access widths; ratios between builds are "
                    "still valid.", "kernels": {}}
for k in sorted(pmc):
    mean = {c: sum(v) / len(v) for c, v in pmc[k].items()}
    rec = {c: round(v, 1) for c, v in sorted(mean.items())}
    if dur[k]:
        rec["dur_us_profiled"] = round(sum(dur[k]) / len(dur[k]), 2)
    if "SQ_INSTS_VALU" in mean and "SQ_ACTIVE_INST_VALU" in mean and mean["SQ_INSTS_VALU"]:
        rec["valu_quad_cycles_per_inst"] = round(mean["SQ_ACTIVE_INST_VALU"] / mean["SQ_INSTS_VALU"], 3)
    out[k] = rec
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        traffic["kernels"][k] = {"FETCH_SIZE_KB": round(mean["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(mean["WRITE_SIZE"], 1),
                                 "hbm_bytes_per_launch": int((2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024)}
# stage-level sums (the keys bench.py looks up): bytes and VALU instructions per launch of the stage
STAGES = {"project_cull": ["k_project_cull", "k_cull_scan", "k_compact"], "project_visible": ["k_project_visible", "k_walk_count"],
          "map_intersects": ["k_map_intersects"], "tile_bins": ["k_tile_bin_edges"],
          "rasterize": ["k_rasterize_quad", "k_rasterize"], "bwd_zero": ["k_zero_compact_grads"],
          "rasterize_bwd": ["k_rasterize_backward_quad", "k_rasterize_backward"], "project_bwd": ["k_project_backward"],
          "sort": ["k_sort_upsweep", "k_sort_scan", "k_sort_downsweep"]}
traffic["valu_insts"] = {}
traffic["_tag"] = tag


def kernel_sources_sha16():
    """Same fingerprint as bench.py's: bench.py reports these counters as stale once the kernel sources differ."""
    import hashlib

    h = hashlib.sha256()
    files = sorted(glob.glob("brush_amd/csrc/*.hip") + glob.glob("brush_amd/csrc/*.hpp") + ["include/brush_hip.h"])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


traffic["_csrc_sha16"] = kernel_sources_sha16()
try:  # the bench line quotes this: which commit the counters were taken at
    import subprocess
    traffic["_commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    pass
if os.path.exists("profiles/traffic.json"):  # keep the heavy workloads' section (tools/summarize_extra.py)
    try:
        old = json.load(open("profiles/traffic.json"))
        if "extra" in old:
            traffic["extra"] = old["extra"]
    except ValueError:
        pass
for stage, ks in STAGES.items():
    b = [traffic["kernels"][k]["hbm_bytes_per_launch"] for k in ks if k in traffic["kernels"]]
    if b:
        traffic[stage] = int(sum(b))
    v = [out[k]["SQ_INSTS_VALU"] for k in ks if k in out and "SQ_INSTS_VALU" in out[k]]
    if v:
        traffic["valu_insts"][stage] = int(sum(v))
json.dump(out, open(f"profiles/{tag}_pmc.json", "w"), indent=1)
if traffic["kernels"]:
    json.dump(traffic, open("profiles/traffic.json", "w"), indent=1)
b = os.path.join(src, "bench_graph.json")
if os.path.exists(b):
    lines = [l for l in open(b) if l.startswith("{")]
    if lines:
        open(f"profiles/{tag}_bench.json", "w").write(lines[-1])
print("kernels:", ", ".join(sorted(pmc)))
