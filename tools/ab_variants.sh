#!/bin/bash
# A/B of variant libraries on ONE box by kernel time: for every variants/libibu_<name>.so given, tools/sortbench.py (1e9 records,
# random index) runs under rocprofv3 --kernel-trace --stats and the line of kernels matching <pattern> is kept.
#   [SB_ARGS="--lens 32,12"] bash tools/ab_variants.sh <out-tag> <kernel-pattern> <name> [<name> ...]
set -o pipefail
TAG=$1; PAT=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for v in "$@"; do
  so=$PWD/variants/libibu_$v.so
  [ -f "$so" ] || { echo "$v: no such library" >> "$OUT/summary.txt"; continue; }
  IBU_HIP_SO=$so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/p" -- python3 tools/sortbench.py --records 1e9 --rounds 3 --random-index --skip-agg ${SB_ARGS:-} \
    > "$OUT/$v.json" 2> "$OUT/$v.err" || { echo "$v: run failed" >> "$OUT/summary.txt"; rm -rf "$OUT/p"; continue; }
  f=$(find "$OUT/p" -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" "$PAT" "$v" "$OUT/$v.json" >> "$OUT/summary.txt" <<'PY'
import csv, json, re, sys
f, pat, v, js = sys.argv[1:5]
rows = [r for r in csv.DictReader(open(f)) if re.search(pat, r["Name"])]
sec = json.loads(open(js).read().strip().splitlines()[-1])["seconds"]
print(v, " ".join(f"{r['Name'].split('(')[0][-40:]}: n={r['Calls']} avg_ms={float(r['AverageNs']) / 1e6:.3f}" for r in rows), f"sort_s={sec}")
PY
  fi
  rm -rf "$OUT/p"
done
cat "$OUT/summary.txt"
