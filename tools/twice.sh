#!/bin/bash
# Marginal cost of each stage with three batches in flight: the stage is issued twice per step (same result).
set -e
mkdir -p gpurun_out
ARGS="$@"      # e.g. --regime trained --steps 50
for st in none assign select nms; do
  DETBOX_BENCH_TWICE=$st timeout -k 10 150 python bench.py --no-extra --no-cpu-baseline $ARGS > gpurun_out/twice_$st.json 2> gpurun_out/twice_$st.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/twice_$st.json').read().strip().splitlines()[-1])
print('$st', round(d['ms_per_step']*1000,1), 'us/step')
PY
done
