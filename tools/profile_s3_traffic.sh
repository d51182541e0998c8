#!/bin/bash
# The two HBM-traffic passes of the S3 workload (21 M splats at 4K) on their own: under --pmc the run writes nothing
# for minutes, so a heartbeat file keeps gpurun's silence watchdog quiet.  Output joins gpurun_out/prof_<tag>_extra/S3.
set -e
TAG=${1:-r03}
OUT=gpurun_out/prof_${TAG}_extra/S3
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
mkdir -p "$OUT"
( while true; do sleep 45; date >> "$OUT/heartbeat.log"; done ) &
HB=$!
trap 'kill $HB 2>/dev/null' EXIT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o p -- python3 tools/ab_stage.py S3 2 > /dev/null 2> "$OUT/fetch.err"
echo "S3 fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o p -- python3 tools/ab_stage.py S3 2 > /dev/null 2> "$OUT/write.err"
echo "S3 write done"
