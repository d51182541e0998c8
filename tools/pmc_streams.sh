#!/bin/bash
# L2 <-> memory request counters of the per-splat streaming kernels at a named workload (default S3: 20 971 520 splats
# @3840x2160), one rocprofv3 --pmc pass per counter group with --kernel-trace only (MI355X_MICROARCH.md, PMC slots):
#   tools/pmc_streams.sh <tag> [workload]   -> gpurun_out/pmcs_<tag>/summary.json
set -e
TAG=$1; WL=${2:-S3}
OUT=gpurun_out/pmcs_$TAG
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
mkdir -p "$OUT"
export KERNEL_RE=${KERNEL_RE:-k_project_[a-z_]*|k_rasterize[a-z_]*|k_sort_[a-z_]*|k_map_[a-z_]*|k_walk_[a-z_]*|k_compact}
i=0
for grp in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_WRITE_REQ_sum TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_NORMAL_WRITEBACK_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/g$i" -o p -- python3 tools/ab_stage.py $WL 3 > "$OUT/g$i.json" 2> "$OUT/g$i.err" || echo "group $i failed"
  echo "group $i done"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, re, json, os
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for d in sorted(glob.glob(f"{out}/g*/")):
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            m = re.search("(" + os.environ["KERNEL_RE"] + ")", r["Kernel_Name"])
            if not m: continue
            k = m.group(1)
            if "downsweep" in k or "upsweep" in k: k += f"_grid{r['Grid_Size']}"
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen and d.endswith("g1/"):
                seen.add(r["Dispatch_Id"]); dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {k: dict({c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}, dur_us=round(sum(dur[k]) / max(1, len(dur[k])), 1)) for k, cs in acc.items()}
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
for k, v in sorted(res.items()): print(k, json.dumps(v))
PY
