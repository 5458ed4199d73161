#!/usr/bin/env python3
"""ibu_load_bgzf_to_device end to end: a file of synthetic records written from device memory, compressed to BGZF on the host (the
test-data writer of tools/gzutil.py: what `bgzip -l 1` writes), loaded back with the compressed bytes crossing the link and every
block inflated on the device; checked with K4 against the resident records.  Beside it the same file through the Reader (blocks
inflated on the host cores) and the plain file through ibu_load_to_device.
  python tools/bgzf_load.py [--records 1e8] [--level 1] [--dir /tmp]     (IBU_TRACE_SORT=1: the call's phases)"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", default="1e8")
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--skip-reader", action="store_true")
    a = ap.parse_args()
    import ibu_amd as ia
    from gzutil import bgzf_parallel
    n = int(float(a.records))
    ctx = ia.Context(0)
    d = ctx.alloc(24 * n)
    ctx.generate(0x1B00004, 0, n, 16, 12, d)
    want = ctx.reduce(d, n)
    ring = {"slots": 4, "slot_records": 4 << 20, "feeder_threads": 8}
    out = {"records": n, "level": a.level}
    with tempfile.TemporaryDirectory(dir=a.dir) as td:
        p, bg = os.path.join(td, "r.ibu"), os.path.join(td, "r.bgzf")
        w = ia.Writer.from_path(p, ia.Header(16, 12))
        w.write_batch_device(ctx, d, n, ring=ring)
        w.finish()
        w.close()
        t0 = time.perf_counter()
        out["bgzf_bytes"] = bgzf_parallel(p, bg, level=a.level, workers=min(16, os.cpu_count() or 1))
        out["host_compress_seconds"] = round(time.perf_counter() - t0, 2)
        ctx.generate(1, 0, n, 16, 12, d)                          # (the destination holds something else before every load)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            _, q, got, st = ctx.load_bgzf_to_device(bg, ring=ring, d_records=d, cap_records=n)
            ts.append(time.perf_counter() - t0)
            assert got == n and ctx.reduce(d, n) == want
        out["load_bgzf_to_device"] = {"seconds": round(min(ts), 4), "calls": [round(t, 4) for t in ts], "G_records_per_s": round(n / min(ts) / 1e9, 3),
                                      "GBps_of_records": round(24 * n / min(ts) / 1e9, 1), "GBps_over_the_link": round(out["bgzf_bytes"] / min(ts) / 1e9, 1)}
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            _, q, got, st = ctx.load_to_device(p, ring=ring, d_records=d, cap_records=n)
            ts.append(time.perf_counter() - t0)
        assert got == n and ctx.reduce(d, n) == want
        out["load_to_device_plain_file"] = {"seconds": round(min(ts), 4), "G_records_per_s": round(n / min(ts) / 1e9, 3)}
        if not a.skip_reader:
            r = ia.Reader.from_path(bg)
            t0 = time.perf_counter()
            tot, st = r.process_device(ctx, ia.PROC_REDUCE, ring=ring)
            dt = time.perf_counter() - t0
            r.close()
            assert tot == want
            out["reader_host_inflate_reduce"] = {"seconds": round(dt, 4), "G_records_per_s": round(n / dt / 1e9, 3)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
