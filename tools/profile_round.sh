#!/bin/bash
# The profile set of a round, taken on the GPU box from ONE command (the device phase of the headline bench):
#   kernel trace + stats, two PMC passes (FETCH_SIZE / WRITE_SIZE do not share a pass), one SQ pass.
# usage (from the repo root, under gpurun): bash tools/profile_round.sh gpurun_out/r3/prof
# Counters are collected in their own runs, never together with sys/hip/hsa tracing (gpurun refuses that).
set -e
OUT=${1:-gpurun_out/prof}
ROOT=$(pwd)
CMD="python3 $ROOT/bench.py --config se --steps 5 --warmup 2 --cpu-sample 0"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats" -o st -- $CMD > "$ROOT/$OUT/bench_under_rocprof.json" 2> "$ROOT/$OUT/stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$ROOT/$OUT/fetch" -o f -- python3 $ROOT/bench.py --config se --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2> "$ROOT/$OUT/fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$ROOT/$OUT/write" -o w -- python3 $ROOT/bench.py --config se --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2> "$ROOT/$OUT/write.err"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d "$ROOT/$OUT/sq" -o s -- python3 $ROOT/bench.py --config se --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2> "$ROOT/$OUT/sq.err"
cd "$ROOT"
# the per-dispatch tables are large: keep what the summaries need
find "$OUT" -name '*agent_info.csv' -delete
python3 tools/prof_summary.py "$OUT/stats" 20
