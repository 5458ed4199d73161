#!/usr/bin/env python3
"""Per-barcode aggregation (BarcodeAnalyzer, parallel.rs:72-98) on device-resident SORTED records: ibu_barcode_counts
timed by phase — the size query (count pass + scan + 16-byte read-back) and the emit call (count + scan + emit + finish)
into preallocated device arrays, no download.
  python tools/aggbench.py [--records 1e9] [--lens 10,12] [--rounds 5]
bc_len 10 gives 2^20 distinct barcodes (a single-cell whitelist's order of magnitude); 16 gives ~n runs of length one."""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", default="1e9")
    ap.add_argument("--lens", default="10,12;16,12")
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    import ibu_amd as ia
    from ibu_amd import _dptr, _check, lib

    ctx = ia.Context(0)
    for n in (int(float(x)) for x in a.records.split(",")):
        d, t = ctx.alloc(24 * n), ctx.alloc(24 * n)
        for lens in a.lens.split(";"):
            bc_len, umi_len = (int(x) for x in lens.split(","))
            ctx.generate(0x1B00005, 0, n, bc_len, umi_len, d)
            ctx.sort_records(d, t, n)
            ctx.synchronize()
            nb, npairs = C.c_size_t(), C.c_size_t()
            q = []
            for _ in range(a.rounds + 1):
                t0 = time.perf_counter()
                _check(lib.ibu_barcode_counts(ctx._c, _dptr(d), n, None, None, None, 0, C.byref(nb), C.byref(npairs), None))
                q.append(time.perf_counter() - t0)
            u = nb.value
            d_b, d_c, d_u = ctx.alloc(8 * u), ctx.alloc(8 * u), ctx.alloc(8 * u)
            e = []
            for _ in range(a.rounds + 1):
                t0 = time.perf_counter()
                _check(lib.ibu_barcode_counts(ctx._c, _dptr(d), n, _dptr(d_b), _dptr(d_c), _dptr(d_u), u, C.byref(nb),
                                              C.byref(npairs), None))
                ctx.synchronize()
                e.append(time.perf_counter() - t0)
            import numpy as np
            counts = d_c.download(np.uint64)
            assert int(counts.sum()) == n and int(d_u.download(np.uint64).sum()) == npairs.value
            qs, es = statistics.median(q[1:]), statistics.median(e[1:])
            # algorithmic bytes: the barcode and UMI words of every record once per pass (16 B; the hardware fetches the
            # whole 24-byte record) + 24 B per distinct barcode written; the emit call runs the count pass again
            print(json.dumps({"n": n, "lens": [bc_len, umi_len], "distinct_barcodes": u, "barcode_umi_pairs": npairs.value,
                              "size_query_ms": round(qs * 1e3, 3), "emit_call_ms": round(es * 1e3, 3),
                              "size_query_GBps_of_24B": round(24 * n / qs / 1e9),
                              "emit_call_GBps_of_24B_x2": round((48 * n + 24 * u) / es / 1e9)}), flush=True)
            for x in (d_b, d_c, d_u):
                x.free()
        d.free()
        t.free()


if __name__ == "__main__":
    main()
