#!/bin/bash
# Evidence run of a round, on the GPU box (gpurun -- 'bash tools/profile_round.sh r02_b'):
#   1. python bench.py                                        -> <tag>_bench_1e9.json
#   2. rocprofv3 --kernel-trace --stats over the same command -> <tag>_bench_1e9_kernel_stats.csv (+ the line it printed)
#   3. rocprofv3 --kernel-trace --stats over tools/sortbench.py at 1e9 (random index; whitelist barcodes in read order)
#      -> <tag>_sort_1e9_kernel_stats.csv, <tag>_sort_whitelist_1e9_kernel_stats.csv
#   4. two --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with each other or with other trace domains) over
#      tools/kbench.py and tools/sortbench.py -> pmc CSVs + pmc_traffic.json (round-tagged)
# Everything lands under gpurun_out/<tag>/; copy what is to be judged into profiles/.
set -o pipefail
TAG=${1:-r04}
N=${2:-1e9}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
step() { echo "[profile_round] $(date +%T) $*"; }
# rocprofv3 writes one file per process it saw: the one to keep is the LARGEST (the python process that ran the kernels)
largest() { find "$1" -name "$2" -printf '%s %p\n' | sort -rn | head -1 | cut -d' ' -f2-; }

step "bench (unprofiled)"
python3 bench.py > "$OUT/${TAG}_bench_1e9.json" 2> "$OUT/bench.err" || exit 1

step "bench under rocprofv3 --kernel-trace --stats"
# (The default placement probing stays ON: since round 3 it is the library's and times the generator / reduce kernels, not decode,
# so the decode launches of the profiled process are the first-placement probe (2), the kept arrays' probe (2), warm-up and the timed
# steps — their rocprofv3 average agrees with the HIP-event time of the timed steps to 0.1 %.  With --placement-tries 1 the profiled
# process would time whatever first placement it drew: 10.17 ms twice in round 3 against 9.36 for the probed headline.)
# (--no-e2e-leg: that leg launches the same decode / encode instantiations on ring slots and on a 1e8-record copy; in the stats file their
# short launches would be averaged with the headline's — 89 decode launches of 1.9 ms on average instead of 17 of 9.45, r04_fin's first run)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bench" -- python3 bench.py --no-cpu-baseline --no-e2e-leg \
  > "$OUT/${TAG}_bench_1e9_under_rocprof.json" 2> "$OUT/prof_bench.err" || exit 1
cp "$(largest "$OUT/prof_bench" '*kernel_stats.csv')" "$OUT/${TAG}_bench_1e9_kernel_stats.csv"

step "sortbench (unprofiled)"
python3 tools/sortbench.py --records "$N" --rounds 3 --random-index --skip-agg --presorted > "$OUT/${TAG}_sortbench_random_index.jsonl" 2> "$OUT/sort.err" || exit 1
python3 tools/sortbench.py --records "$N" --rounds 3 --skip-agg > "$OUT/${TAG}_sortbench_read_order.jsonl" 2>> "$OUT/sort.err" || exit 1

step "sortbench, barcodes from a whitelist of 1e5 in read order (all passes)"
python3 tools/sortbench.py --records "$N" --rounds 3 --skip-agg --whitelist 100000 > "$OUT/${TAG}_sortbench_whitelist_read_order.jsonl" 2>> "$OUT/sort.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_sort_wl" -- python3 tools/sortbench.py --records "$N" --rounds 3 --skip-agg --whitelist 100000 \
  > "$OUT/${TAG}_sortbench_whitelist_under_rocprof.jsonl" 2> "$OUT/prof_sort_wl.err" || exit 1
cp "$(largest "$OUT/prof_sort_wl" '*kernel_stats.csv')" "$OUT/${TAG}_sort_whitelist_1e9_kernel_stats.csv"
rm -rf "$OUT/prof_sort_wl"

step "sortbench under rocprofv3 --kernel-trace --stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_sort" -- python3 tools/sortbench.py --records "$N" --rounds 3 --random-index --skip-agg \
  > "$OUT/${TAG}_sortbench_under_rocprof.jsonl" 2> "$OUT/prof_sort.err" || exit 1
cp "$(largest "$OUT/prof_sort" '*kernel_stats.csv')" "$OUT/${TAG}_sort_1e9_kernel_stats.csv"

for C in FETCH_SIZE WRITE_SIZE; do
  step "pmc $C over kbench"
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc/$C" -- python3 tools/kbench.py --records "$N" --rounds 2 \
    > "$OUT/pmc_kbench_$C.log" 2>&1 || exit 1
  cp "$(largest "$OUT/pmc/$C" '*counter_collection.csv')" "$OUT/${TAG}_pmc_${C}_kbench_1e9.csv"
  step "pmc $C over sortbench"
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_sort/$C" -- python3 tools/sortbench.py --records "$N" --rounds 1 --random-index --skip-agg \
    > "$OUT/pmc_sort_$C.log" 2>&1 || exit 1
  cp "$(largest "$OUT/pmc_sort/$C" '*counter_collection.csv')" "$OUT/${TAG}_pmc_${C}_sort_1e9.csv"
done
python3 tools/pmc_traffic.py "$OUT/pmc" "$N" 16,12 "$TAG" > "$OUT/pmc_traffic_16_12.json" || exit 1
# the runtime-length kernels (round 4: through the code stream): the same two passes at (31,31), 5e8 records
for C in FETCH_SIZE WRITE_SIZE; do
  step "pmc $C over kbench, lens 31,31"
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_gen/$C" -- python3 tools/kbench.py --records 5e8 --lens 31,31 --rounds 2 \
    --kernels decode,encode,unpack,pack > "$OUT/pmc_kbench_gen_$C.log" 2>&1 < /dev/null || exit 1
done
python3 tools/pmc_traffic.py "$OUT/pmc_gen" 5e8 31,31 "$TAG" > "$OUT/pmc_traffic_31_31.json" || exit 1
python3 - "$OUT/pmc_traffic_16_12.json" "$OUT/pmc_traffic_31_31.json" > "$OUT/pmc_traffic.json" <<'PY' || exit 1
import json, sys
a, g = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
a["runtime_length_31_31"] = {k: v for k, v in g.items() if not k.startswith("_")}
a["runtime_length_31_31"]["_records"] = g["_records"]
print(json.dumps(a, indent=1))
PY
step "the multi-GPU sort rehearsed on this GPU (8 and 2 contexts)"
python3 tools/sortbench.py --records "$N" --shards 8 --rounds 3 > "$OUT/${TAG}_sort_contexts_rehearsal.jsonl" 2>> "$OUT/sort.err" < /dev/null || exit 1
python3 tools/sortbench.py --records "$N" --shards 2 --rounds 3 >> "$OUT/${TAG}_sort_contexts_rehearsal.jsonl" 2>> "$OUT/sort.err" < /dev/null || exit 1
rm -rf "$OUT/prof_bench" "$OUT/prof_sort" "$OUT/pmc" "$OUT/pmc_sort" "$OUT/pmc_gen"   # the raw traces are large; the summaries above are what is kept
step done
