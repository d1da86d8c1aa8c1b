#!/bin/bash
# SQ counters of the NMS kernel alone (tools/exp_nms.py), two passes.  Run on the GPU box from the repo root:
#   bash tools/pmc_nms.sh <tag> [workload regime]
set -e -o pipefail
TAG=${1:-pmcnms}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES --output-format csv -d "$OUT/a" -o a -- python3 "$ROOT/tools/exp_nms.py" "$@" > "$OUT/a.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d "$OUT/b" -o b -- python3 "$ROOT/tools/exp_nms.py" "$@" > "$OUT/b.log" 2>&1
python3 "$ROOT/tools/pmc_summary.py" "$OUT/a" "$OUT/b"
