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


def _rand_bits(torch, g, nbits, count):
    """count random nbits-bit values as int64 bit patterns (nbits up to 64)."""
    if nbits <= 62:
        return torch.randint(0, 1 << nbits, (count,), generator=g, device="cuda", dtype=torch.int64)
    hi = torch.randint(0, 1 << (nbits - 32), (count,), generator=g, device="cuda", dtype=torch.int64)
    return (hi << 32) | torch.randint(0, 1 << 32, (count,), generator=g, device="cuda", dtype=torch.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", default="1e9")
    ap.add_argument("--lens", default="10,12;16,12")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--whitelist", type=int, default=0, help="K > 0: barcodes drawn from K distinct ones, skewed (rank ~ K u^3), as tools/sortbench.py --whitelist")
    a = ap.parse_args()
    if a.whitelist:
        import torch                                         # before the library, as bench.py does
        torch.cuda.init()
    import ibu_amd as ia
    from ibu_amd import _dptr, _check, lib

    ctx = ia.Context(0)
    for n in (int(float(x)) for x in a.records.split(",")):
        d, t = ctx.alloc(24 * n), ctx.alloc(24 * n)
        for lens in a.lens.split(";"):
            bc_len, umi_len = (int(x) for x in lens.split(","))
            ctx.generate(0x1B00005, 0, n, bc_len, umi_len, d)
            if a.whitelist:                                    # replace the barcode column
                cols = [ctx.alloc(8 * n) for _ in range(3)]
                ctx.deserialize(d, n, cols[0], cols[1], cols[2])
                g = torch.Generator(device="cuda").manual_seed(0x1B00007)
                wl = _rand_bits(torch, g, 2 * bc_len, a.whitelist)
                bc = torch.as_tensor(cols[0], device="cuda").view(torch.int64)
                for lo in range(0, n, 1 << 26):
                    hi = min(n, lo + (1 << 26))
                    u = torch.rand(hi - lo, generator=g, device="cuda", dtype=torch.float64)
                    bc[lo:hi] = wl[(u * u * u * a.whitelist).to(torch.int64).clamp_(max=a.whitelist - 1)]
                del u
                torch.cuda.synchronize()
                ctx.serialize(cols[0], cols[1], cols[2], n, d)
                for c in cols:
                    c.free()
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
            print(json.dumps({"n": n, "lens": [bc_len, umi_len], "whitelist": a.whitelist or None, "distinct_barcodes": u, "barcode_umi_pairs": npairs.value,
                              "size_query_ms": round(qs * 1e3, 3), "emit_call_ms": round(es * 1e3, 3),
                              "size_query_GBps_of_24B": round(24 * n / qs / 1e9),
                              "emit_call_GBps_of_24B_x2": round((48 * n + 24 * u) / es / 1e9)}), flush=True)
            for x in (d_b, d_c, d_u):
                x.free()
        d.free()
        t.free()


if __name__ == "__main__":
    main()
