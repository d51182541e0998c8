#!/bin/bash
# HBM write amplification of the radix-sort scatter at the S3 workload (20 971 520 splats @3840x2160, 18 M intersections):
# rocprofv3 --pmc WRITE_SIZE (own pass, kernel trace only) + the bench's own stage timing.  Output: gpurun_out/pmc_sort_s3/summary.json
set -e
OUT=gpurun_out/pmc_sort_s3
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
mkdir -p "$OUT"
B="bench.py --splats 20971520 --width 3840 --height 2160 --max-intersects 24000000 --no-graph --steps 3 --warmup 1 --profile-steps 3 --train-steps 0 --no-cpu-baseline --no-extra"
python3 $B > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/w" -o p -- python3 $B > /dev/null 2> "$OUT/w.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/f" -o p -- python3 $B > /dev/null 2> "$OUT/f.err"
python3 - "$OUT" <<'PY'
import csv, glob, json, re, sys
from collections import defaultdict
out = sys.argv[1]
b = json.loads([l for l in open(f"{out}/bench.json") if l.startswith("{")][-1])
I, V = b["config"]["num_intersections"], b["config"]["num_visible"]
acc = defaultdict(lambda: defaultdict(list))
for d, cn in (("w", "WRITE_SIZE"), ("f", "FETCH_SIZE")):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_sort_[a-z]+)", r["Kernel_Name"])
            if m and r["Counter_Name"] == cn:
                # the tile sort runs on I keys, the depth sort on V keys: tell them apart by the grid size
                acc[(m.group(1), int(r["Grid_Size"]))][cn].append(float(r["Counter_Value"]))
res = {"workload": b["config"]["workload"], "num_visible": V, "num_intersections": I, "stage_ms": b["stage_ms"], "kernels": {}}
for (k, grid), cs in sorted(acc.items()):
    res["kernels"][f"{k} grid={grid}"] = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
# tile sort: 2 passes of 8 bits over I pairs: algorithmic 4 B/key (count) + 16 B/pair (scatter) per pass
alg = I * 20 * 2
t = b["stage_ms"]["tile_sort"] * 1e-3
res["tile_sort"] = {"algorithmic_bytes": alg, "ms": b["stage_ms"]["tile_sort"], "GBs": round(alg / t / 1e9, 1), "frac_of_hbm_peak": round(alg / t / 8e12, 4),
                    "scatter_write_algorithmic_KB_per_launch": round(I * 8 / 1024, 1)}
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
print(json.dumps(res))
PY
