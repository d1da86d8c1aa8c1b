#!/bin/bash
# Kernel trace of the default bench command (three batches in flight) with per-dispatch timestamps, and the overlap summary.
#   bash tools/trace_overlap.sh <tag> [bench args...]
set -e -o pipefail
TAG=${1:-ovl}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-extra --blocks 3 "$@" > "$OUT/trace.log" 2>&1
F=$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)
python3 "$ROOT/tools/overlap.py" "$F" --json "$OUT/overlap.json" > "$OUT/overlap.txt"
cat "$OUT/overlap.txt"
