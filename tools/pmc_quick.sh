#!/bin/bash
# Quick SQ-counter comparison of kernel variants (development aid): tools/pmc_quick.sh <tag> [bench args...]
# KERNEL_RE selects the kernels (default: the compositing kernels), TRAIN_STEPS > 0 adds the training leg.
# The environment (e.g. BRUSH_DETERMINISTIC) is inherited by the profiled process.
set -e
TAG=$1; shift
OUT=gpurun_out/pmcq_$TAG
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
mkdir -p "$OUT"
B="bench.py --no-graph --steps 4 --warmup 2 --profile-steps 0 --train-steps ${TRAIN_STEPS:-0} --no-cpu-baseline $*"
export KERNEL_RE=${KERNEL_RE:-k_rasterize[a-z0-9_]*}
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq1" -o p -- python3 $B > /dev/null 2> "$OUT/sq1.err"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/sq2" -o p -- python3 $B > /dev/null 2> "$OUT/sq2.err"
rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU_TRANS_F32 --output-format csv -d "$OUT/sq3" -o p -- python3 $B > /dev/null 2> "$OUT/sq3.err"
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_LEVEL_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_CYCLES --output-format csv -d "$OUT/sq4" -o p -- python3 $B > /dev/null 2> "$OUT/sq4.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys, re, json, os
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for d in ("sq1", "sq2", "sq3", "sq4"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            m = re.search("(" + os.environ["KERNEL_RE"] + ")", r["Kernel_Name"])
            if not m: continue
            k = m.group(1) + ("_u32" if "ILb1" in r["Kernel_Name"] else "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if d == "sq1" and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"]); dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {k: dict({c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}, dur_us=round(sum(dur[k]) / max(1, len(dur[k])), 1)) for k, cs in acc.items()}
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
for k, v in res.items(): print(k, json.dumps(v))
PY
