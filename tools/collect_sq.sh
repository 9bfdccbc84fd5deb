#!/bin/bash
# Run on the GPU box (through gpurun):  bash tools/collect_sq.sh <tag> <driver args...>
# SQ / TCC counter passes of tools/gpu_atrium_frames.py (one workload, nothing else), condensed by tools/summarize_sq.py.
set -o pipefail
TAG=${1:-run}; shift || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out/sq_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/tools/gpu_atrium_frames.py" "$@" > "$OUT/stats.log" 2>&1
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
P2="SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"
P3="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
P6="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d "$OUT/pmc$i" -o pmc -- python3 "$ROOT/tools/gpu_atrium_frames.py" "$@" > "$OUT/pmc$i.log" 2>&1 || echo "pass $i failed" >> "$OUT/fail.log"
done
cd "$ROOT"
python3 tools/summarize_sq.py "$OUT" "$TAG"
