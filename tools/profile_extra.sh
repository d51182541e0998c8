#!/bin/bash
# rocprofv3 evidence for the heavy workloads of bench.py's extra_workloads (dense scene, c3, S3), run through gpurun:
#   per workload: --kernel-trace --stats (one run), --pmc FETCH_SIZE and --pmc WRITE_SIZE (one run each, kernel trace only)
# Output: gpurun_out/prof_<tag>_extra/<workload>/..., condensed by tools/summarize_extra.py into profiles/.
set -e
TAG=${1:-r03}
OUT=gpurun_out/prof_${TAG}_extra
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
for W in dense c3 S3; do
  mkdir -p "$OUT/$W"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$W/kt" -o kt -- python3 tools/ab_stage.py $W 6 > "$OUT/$W/ab.json" 2> "$OUT/$W/kt.err"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/$W/fetch" -o p -- python3 tools/ab_stage.py $W 3 > /dev/null 2> "$OUT/$W/fetch.err"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/$W/write" -o p -- python3 tools/ab_stage.py $W 3 > /dev/null 2> "$OUT/$W/write.err"
  echo "$W done"
done
