#!/bin/bash
# The memory side of a command's kernels: three rocprofv3 --pmc passes of TCC / TCP counters (only with --kernel-trace, the program
# directly after `--`; a failed pass stops the script), one line per kernel.   bash tools/mem_pmc.sh <out-tag> tools/kbench.py <args ...>
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
largest() { find "$1" -name "$2" -printf '%s %p\n' | sort -rn | head -1 | cut -d' ' -f2-; }
i=0
for G in "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
         "TCC_TAG_STALL_sum TCC_BUBBLE_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/g$i" -- python3 "$@" > "$OUT/g$i.log" 2>&1 < /dev/null || { tail -5 "$OUT/g$i.log"; exit 1; }
  cp "$(largest "$OUT/g$i" '*counter_collection.csv')" "$OUT/${TAG}_mem_g$i.csv"
  rm -rf "$OUT/g$i"
done
python3 tools/sq_summary.py "$OUT/${TAG}_mem_g1.csv" "$OUT/${TAG}_mem_g2.csv" "$OUT/${TAG}_mem_g3.csv" > "$OUT/${TAG}_mem_summary.jsonl" < /dev/null
cat "$OUT/${TAG}_mem_summary.jsonl"
