#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 --pmc SQ_* passes (counter_collection.csv files; all files given are merged).

  python tools/sq_summary.py <csv> [<csv> ...] [--pattern REGEX] [--records N]

Cycle counters (SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_*) are printed as fractions of the waves'
cycles; SQ_INSTS_* as thread-instructions per record when --records is given (wave instructions x 64 / N), else per launch.
Any other counter (TCC_*, TCP_*: tools/mem_pmc.sh) is printed under its own name, per record with --records, else per launch.
Counters are averaged over the launches of a kernel."""
import argparse
import collections
import csv
import json
import re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv", nargs="+")
    ap.add_argument("--pattern", default="ibu_k_")
    ap.add_argument("--records", type=float, default=0)
    a = ap.parse_args()
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in a.csv:
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if not re.search(a.pattern, r["Kernel_Name"]):
                continue
            key = (f, r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])   # one row per XCD / SE instance: sum
            names[(f, r["Dispatch_Id"])] = r["Kernel_Name"].split("(")[0]
        for (ff, d, c), v in per_dispatch.items():
            acc[names[(ff, d)]][c].append(v)
    for k in sorted(acc):
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        out = {"kernel": k[-60:], "launches": max(len(v) for v in acc[k].values())}
        wc = c.get("SQ_WAVE_CYCLES")
        for n, v in sorted(c.items()):
            if n.startswith("SQ_INSTS_"):
                out[n[9:].lower() + ("_per_record" if a.records else "")] = round(v * 64 / a.records, 2) if a.records else v
            elif n.startswith("SQ_") and wc and n != "SQ_WAVE_CYCLES":
                out[n[3:].lower() + "_frac"] = round(v / wc, 3)
            elif n != "SQ_WAVE_CYCLES":        # TCC_* / TCP_* (tools/mem_pmc.sh) and SQ cycle counters without SQ_WAVE_CYCLES: own name
                out[n + ("_per_record" if a.records else "")] = round(v / a.records, 4) if a.records else v
        print(json.dumps(out))


if __name__ == "__main__":
    main()
