#!/bin/bash
# Everything `profiles/` holds under one tag, from the default bench command.  Run on the GPU box from the repo root:
#   bash tools/profile_round.sh <tag>
# kernel trace + stats (default command, and one batch at a time), HBM traffic (FETCH_SIZE and WRITE_SIZE, a pass each),
# SQ instruction counters (one pass), and the kernel trace of the SSD300 workload.
set -e -o pipefail
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
# the kernel sources the counters below are taken on (tools/pmc_*.py stamp profiles/*.json with it)
python3 -c "import sys; sys.path.insert(0, '$ROOT'); from object_detection_amd.build import source_hash; print(source_hash())" > "$OUT/source_sha256.txt"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $B > "$OUT/stats.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats1" -o stats1 -- $B --streams 1 > "$OUT/stats1.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ssd300" -o ssd300 -- $B --workload ssd300 --steps 100 > "$OUT/ssd300.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- $B --steps 3 --warmup 1 --blocks 1 --streams 1 > "$OUT/fetch.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- $B --steps 3 --warmup 1 --blocks 1 --streams 1 > "$OUT/write.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -o sq -- $B --steps 3 --warmup 1 --blocks 1 --streams 1 > "$OUT/sq.log" 2>&1
find "$OUT" -name '*.csv' | head -30
