#!/usr/bin/env python3
"""Throughput of ibu_sort_records on device-resident records.
  python tools/sortbench.py [--records 1e8,1e9] [--lens 16,12] [--rounds 3]"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", default="1e8")
    ap.add_argument("--lens", default="16,12")
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    import ibu_amd as ia

    bc_len, umi_len = (int(x) for x in a.lens.split(","))
    ctx = ia.Context(0)
    for n in (int(float(x)) for x in a.records.split(",")):
        d, t = ctx.alloc(24 * n), ctx.alloc(24 * n)
        ts = []
        for _ in range(a.rounds + 1):
            ctx.generate(0x1B00005, 0, n, bc_len, umi_len, d)  # random barcode/UMI, increasing index: unsorted
            ctx.synchronize()
            t0 = time.perf_counter()
            ctx.sort_records(d, t, n)
            ctx.synchronize()
            ts.append(time.perf_counter() - t0)
        assert ctx.is_sorted(d, n)
        sec = statistics.median(ts[1:])
        idx_bytes = max(1, ((n - 1).bit_length() + 7) // 8)
        passes = (2 * bc_len + 7) // 8 + (2 * umi_len + 7) // 8 + idx_bytes
        print(json.dumps({"n": n, "lens": [bc_len, umi_len], "seconds": round(sec, 4), "M_records_per_s": round(n / sec / 1e6, 1),
                          "passes": passes, "GBps_at_72B_per_record_pass": round(n * 72 * passes / sec / 1e9)}), flush=True)
        d.free()
        t.free()


if __name__ == "__main__":
    main()
