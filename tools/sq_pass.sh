#!/bin/bash
# Two rocprofv3 --pmc passes of SQ counters (cycles, then instruction counts; never combined with trace domains other than
# --kernel-trace) over one python command, summarised per kernel.   bash tools/sq_pass.sh <out-tag> <records> tools/kbench.py <args ...>
set -o pipefail
TAG=$1; N=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
largest() { find "$1" -name "$2" -printf '%s %p\n' | sort -rn | head -1 | cut -d' ' -f2-; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS \
  --output-format csv -d "$OUT/cyc" -- python3 "$@" > "$OUT/cyc.log" 2>&1 < /dev/null || { tail -5 "$OUT/cyc.log"; exit 1; }
cp "$(largest "$OUT/cyc" '*counter_collection.csv')" "$OUT/${TAG}_sq_cycles.csv"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVES \
  --output-format csv -d "$OUT/ins" -- python3 "$@" > "$OUT/ins.log" 2>&1 < /dev/null || { tail -5 "$OUT/ins.log"; exit 1; }
cp "$(largest "$OUT/ins" '*counter_collection.csv')" "$OUT/${TAG}_sq_insts.csv"
rm -rf "$OUT/cyc" "$OUT/ins"
python3 tools/sq_summary.py "$OUT/${TAG}_sq_cycles.csv" "$OUT/${TAG}_sq_insts.csv" --records "$N" > "$OUT/${TAG}_sq_summary.jsonl" < /dev/null
cat "$OUT/${TAG}_sq_summary.jsonl"
