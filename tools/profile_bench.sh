#!/bin/bash
# Profiles the default bench command: per-kernel time (kernel trace + stats) and, in separate passes,
# the L2 memory-side read/write byte counters.  Run on the GPU box from the repo root:
#   bash tools/profile_bench.sh <tag> [bench args...]
set -e -o pipefail
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/stats.log" 2>&1
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 3 --warmup 1 "$@" > "$OUT/fetch.log" 2>&1
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 3 --warmup 1 "$@" > "$OUT/write.log" 2>&1
find "$OUT" -name '*.csv' | head -20
