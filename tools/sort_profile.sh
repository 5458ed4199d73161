#!/bin/bash
# Sort evidence on the GPU box: sortbench unprofiled, under rocprofv3 --kernel-trace --stats, and under the two PMC passes
# (separate runs, never combined with other trace domains).  bash tools/sort_profile.sh <tag> [records] [extra sortbench args]
set -o pipefail
TAG=${1:-r03_sort}
N=${2:-1e9}
shift 2
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
largest() { find "$1" -name "$2" -printf '%s %p\n' | sort -rn | head -1 | cut -d' ' -f2-; }
step() { echo "[sort_profile] $(date +%T) $*"; }
step "sortbench unprofiled"
python3 tools/sortbench.py --records "$N" --rounds 3 --random-index --skip-agg "$@" > "$OUT/${TAG}_sortbench_random_index.jsonl" 2> "$OUT/sort.err" || exit 1
python3 tools/sortbench.py --records "$N" --rounds 3 --skip-agg "$@" > "$OUT/${TAG}_sortbench_read_order.jsonl" 2>> "$OUT/sort.err" || exit 1
step "kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_sort" -- python3 tools/sortbench.py --records "$N" --rounds 3 --random-index --skip-agg "$@" \
  > "$OUT/${TAG}_sortbench_under_rocprof.jsonl" 2> "$OUT/prof_sort.err" || exit 1
cp "$(largest "$OUT/prof_sort" '*kernel_stats.csv')" "$OUT/${TAG}_sort_kernel_stats.csv"
rm -rf "$OUT/prof_sort"
if [ -z "$NO_PMC" ]; then
for C in FETCH_SIZE WRITE_SIZE; do
  step "pmc $C"
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_sort/$C" -- python3 tools/sortbench.py --records "$N" --rounds 1 --random-index --skip-agg "$@" \
    > "$OUT/pmc_sort_$C.log" 2>&1 || exit 1
  cp "$(largest "$OUT/pmc_sort/$C" '*counter_collection.csv')" "$OUT/${TAG}_pmc_${C}_sort.csv"
done
fi
step done
