#!/bin/bash
# Average time of the kernels matching a pattern in one tools/sortbench.py run under rocprofv3 --kernel-trace --stats (guarded:
# a missing stats file ends the script, nothing ever reads stdin).  bash tools/kernel_time.sh <pattern> <sortbench args ...>
set -o pipefail
PAT=$1; shift
OUT=gpurun_out/ktime
mkdir -p "$OUT"
export TMPDIR=/tmp
rm -rf "$OUT/p"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/p" -- python3 tools/sortbench.py "$@" > "$OUT/run.jsonl" 2> "$OUT/run.err" < /dev/null || { echo "run failed"; exit 1; }
f=$(find "$OUT/p" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] || { echo "no stats file"; exit 1; }
python3 - "$f" "$PAT" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print(f"{r['Name'].split('(')[0][-48:]:50s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1e3:9.1f}")
PY
rm -rf "$OUT/p"
