#!/bin/bash
# gpurun -- 'bash tools/placement_pmc.sh': tools/placement_pmc.py plainly, then once per counter group under rocprofv3 --pmc
# (counters only with --kernel-trace; the program directly after `--`).  A GRBM group (GRBM_UTCL2_BUSY GRBM_EA_BUSY
# GRBM_GUI_ACTIVE) was tried once and left a dispatch incomplete (run killed after 7 silent minutes): not collected again.
set -o pipefail
OUT=gpurun_out/placement_pmc
mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/placement_pmc.py --cycles 8 > $OUT/plain.jsonl 2> $OUT/plain.err || exit 1
i=0
for G in "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCC_TAG_STALL_sum TCC_BUBBLE_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum" \
         "TCC_EA0_WRREQ TCC_EA0_RDREQ"; do
  i=$((i+1))
  echo "[placement_pmc] $(date +%T) group $i: $G"
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/g$i -- python3 tools/placement_pmc.py --cycles 8 > $OUT/g$i.jsonl 2> $OUT/g$i.err || { tail -5 $OUT/g$i.err; continue; }
  python3 tools/placement_pmc.py --analyse $OUT/g$i > $OUT/g$i.analysis.json || true
  if [ $i -eq 4 ]; then cp "$(find $OUT/g$i -name '*counter_collection.csv' | head -1)" $OUT/g4_per_channel_counter_collection.csv; fi
  rm -rf $OUT/g$i
done
echo "[placement_pmc] done"
