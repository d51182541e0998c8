#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's numbers on the GPU box (run through gpurun):
#   1. kernel trace + stats of the default bench command (graph replay, the timed configuration)
#   2. kernel trace of eager launches (per-kernel durations with the names un-merged)
#   3. PMC passes on eager launches, each in its own run with --kernel-trace only
#      (SQ instruction mix / wait counters, then FETCH_SIZE and WRITE_SIZE separately, as
#      MI355X_MICROARCH.md prescribes)
# Output: gpurun_out/prof_<tag>/..., summarised by tools/summarize_profiles.py into profiles/.
set -e
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
mkdir -p "$OUT"
BENCH_EAGER="bench.py --no-graph --steps 6 --warmup 2 --profile-steps 0 --train-steps 0 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/graph" -o kt -- python3 bench.py --train-steps 20 --no-cpu-baseline --no-extra > "$OUT/bench_graph.json" 2> "$OUT/graph.err"
echo "graph trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/eager" -o kt -- python3 $BENCH_EAGER > "$OUT/bench_eager.json" 2> "$OUT/eager.err"
echo "eager trace done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq1" -o p -- python3 $BENCH_EAGER > /dev/null 2> "$OUT/pmc_sq1.err"
echo "pmc sq1 done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS --output-format csv -d "$OUT/pmc_sq2" -o p -- python3 $BENCH_EAGER > /dev/null 2> "$OUT/pmc_sq2.err"
echo "pmc sq2 done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o p -- python3 $BENCH_EAGER > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o p -- python3 $BENCH_EAGER > /dev/null 2> "$OUT/pmc_write.err"
echo "pmc traffic done"
find "$OUT" -name "*.csv" | head -30
