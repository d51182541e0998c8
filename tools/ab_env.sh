#!/bin/bash
# A/B timing of one environment switch on the GPU box: tools/ab_env.sh VAR v1 v2 ... [-- extra bench args]
# Runs the default bench (graph replay, 200 steps) REPS times per value, alternating the values, and prints the
# median ms/step and the median stage split.
VAR=$1; shift
VALS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done
[ "$1" == "--" ] && shift
REPS=${REPS:-3}
OUT=gpurun_out/ab_$VAR
rm -rf $OUT; mkdir -p $OUT
for r in $(seq 1 $REPS); do
  for v in "${VALS[@]}"; do
    env $VAR=$v python3 bench.py --no-extra --no-cpu-baseline --train-steps 0 --steps 200 --profile-steps 40 "$@" > $OUT/$v.$r.json 2> $OUT/$v.$r.err || { tail -5 $OUT/$v.$r.err; exit 1; }
  done
done
python3 - "$OUT" "$REPS" "${VALS[@]}" <<'PY'
import json, sys, statistics
out, reps, vals = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
for v in vals:
    ds = [json.loads(open(f'{out}/{v}.{r}.json').read().strip().splitlines()[-1]) for r in range(1, reps + 1)]
    ms = sorted(d['ms_per_step'] for d in ds)
    st = {k: round(statistics.median(d['stage_ms'][k] for d in ds) * 1000, 1) for k in ds[0]['stage_ms']}
    print(v, 'graph ms', ms, 'median', statistics.median(ms), st)
PY
