#!/bin/bash
# gpurun -- 'bash tools/placement_pmc.sh': tools/placement_pmc.py plainly, then once per counter group under rocprofv3 --pmc
# (counters only with --kernel-trace; the program directly after `--`).
#
# NOT collected: the GRBM group (GRBM_UTCL2_BUSY GRBM_EA_BUSY GRBM_GUI_ACTIVE).  It was tried once in round 2 and the run
# was killed after 7 silent minutes with a dispatch incomplete.  What that run left behind was not kept (gpurun_out/ is
# scratch and the call was cut off before its files were merged), so the cause can only be stated from what is known about
# the counters: GRBM_* are block-level (whole-GPU, free-running) counters of the graphics register bus manager, not per-CU /
# per-channel counters that rocprofv3's dispatch-serialised --pmc mode can start and stop around one kernel; the other
# four groups (TCC / TCP) are the per-dispatch kind and have always completed.  The group is dropped for that stated
# reason and this script never retries it: a counter run that hangs costs the box (and a strike).
#
# A profiled run that fails or is killed STOPS the script (exit 1): the GPU may be wedged, and carrying on with the next
# counter group on it — as the first version of this loop did with `continue` — risks a second hang on the same box.
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
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/g$i -- python3 tools/placement_pmc.py --cycles 8 > $OUT/g$i.jsonl 2> $OUT/g$i.err || { tail -5 $OUT/g$i.err; echo "[placement_pmc] group $i failed: stopping (no further GPU step on this box)"; exit 1; }
  python3 tools/placement_pmc.py --analyse $OUT/g$i > $OUT/g$i.analysis.json || true
  if [ $i -eq 4 ]; then cp "$(find $OUT/g$i -name '*counter_collection.csv' -printf '%s %p\n' | sort -rn | head -1 | cut -d' ' -f2-)" $OUT/g4_per_channel_counter_collection.csv; fi
  rm -rf $OUT/g$i
done
echo "[placement_pmc] done"
