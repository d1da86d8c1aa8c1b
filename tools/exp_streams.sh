#!/bin/bash
# Step time of the default workload against the number of batches in flight and the scan's residency.
set -e
mkdir -p gpurun_out
B="python bench.py --no-extra --no-cpu-baseline"
pr() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d['ms_per_step']*1000,1), 'us/step', round(d['value']), d['blocks']['ms_per_step_min'])
PY
}
for s in 3 4 5 6 8 3; do
timeout -k 10 120 $B --streams $s > gpurun_out/st_$s.json 2>gpurun_out/st.err; pr gpurun_out/st_$s.json
done
for c in 1 3; do
DETBOX_SCAN_PER_CU=$c timeout -k 10 120 $B > gpurun_out/st_cu$c.json 2>gpurun_out/st.err; pr gpurun_out/st_cu$c.json
done
timeout -k 10 120 $B --graph > gpurun_out/st_g3.json 2>gpurun_out/st.err; pr gpurun_out/st_g3.json
timeout -k 10 120 $B --graph --streams 4 > gpurun_out/st_g4.json 2>gpurun_out/st.err; pr gpurun_out/st_g4.json
timeout -k 10 120 $B --graph --streams 6 > gpurun_out/st_g6.json 2>gpurun_out/st.err; pr gpurun_out/st_g6.json
