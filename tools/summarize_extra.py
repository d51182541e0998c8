"""Condenses tools/profile_extra.sh output into profiles/: <tag>_{dense,c3,s3}_kernel_stats.csv (rocprofv3 --stats of the
workload's run), <tag>_extra_stage_times.json (the same runs' hipEvent stage times) and the "extra" section of
profiles/traffic.json (HBM bytes per launch of every stage: (2 * FETCH_SIZE + WRITE_SIZE) KB, MI355X_MICROARCH.md).

usage: python tools/summarize_extra.py gpurun_out/prof_<tag>_extra <tag>"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
STAGES = {"project_cull": ["k_project_cull", "k_cull_scan", "k_compact"], "project_visible": ["k_project_visible", "k_walk_count"],
          "map_intersects": ["k_map_intersects"], "rasterize": ["k_rasterize_quad"], "bwd_zero": ["k_zero_compact_grads"],
          "rasterize_bwd": ["k_rasterize_backward_quad"], "project_bwd": ["k_project_backward"],
          "sort": ["k_sort_upsweep", "k_sort_scan", "k_sort_downsweep", "k_sort_downsweep_big"]}


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else None


tpath = "profiles/traffic.json"
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
traffic.setdefault("extra", {})
times = {}
for w in ("dense", "c3", "S3"):
    d = os.path.join(src, w)
    if not os.path.isdir(d):
        continue
    g = glob.glob(os.path.join(d, "kt", "**", "*kernel_stats.csv"), recursive=True)
    if g:
        shutil.copy(g[0], f"profiles/{tag}_{w.lower()}_kernel_stats.csv")
    ab = os.path.join(d, "ab.json")
    if os.path.exists(ab):
        lines = [l for l in open(ab) if l.startswith("{")]
        if lines:
            times[w] = json.loads(lines[-1])
    per = defaultdict(lambda: defaultdict(list))
    for cn, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k and r["Counter_Name"] == cn:
                    per[k][cn].append(float(r["Counter_Value"]))
    kern = {}
    for k, cs in per.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            f, wr = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]), sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
            kern[k] = {"FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(wr, 1), "hbm_bytes_per_launch": int((2 * f + wr) * 1024)}
    entry = {"kernels": kern}
    for stage, ks in STAGES.items():
        b = [kern[k]["hbm_bytes_per_launch"] for k in ks if k in kern]
        if b:
            entry[stage] = int(sum(b))
    traffic["extra"][w if w != "dense" else "dense_scene"] = entry
try:
    traffic["_commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    pass
json.dump(traffic, open(tpath, "w"), indent=1)
json.dump(times, open(f"profiles/{tag}_extra_stage_times.json", "w"), indent=1)
print("workloads:", ", ".join(times))
