#!/usr/bin/env python3
"""Write / read speed of the device memory, chunk by chunk.

The rate of the streaming kernels follows the physical pages of the arrays they WRITE (bench.py: placement probing).
This tool asks how that property is laid out: it allocates the card in chunks (hipMalloc through the C ABI), and times
per chunk  generate (write only, 24 B/record),  reduce (read only)  and  copy from one fixed source chunk (read + write).
One JSON line per chunk, then a summary (quantiles, run lengths of fast / slow chunks in allocation order).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunk-mib", type=int, default=1024)
    ap.add_argument("--max-chunks", type=int, default=262)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--repeat-map", type=int, default=2, help="measure the whole map this many times (is the property stable?)")
    a = ap.parse_args()
    import ibu_amd as ia

    ctx = ia.Context(0)
    nbytes = a.chunk_mib << 20
    n = nbytes // 24
    chunks = []
    for _ in range(a.max_chunks):
        try:
            chunks.append(ctx.alloc(nbytes))
        except Exception:
            break
    src = chunks.pop()   # the fixed source of the copies
    ctx.generate(7, 0, n, 16, 12, src)
    ctx.synchronize()

    def timed(fn):
        best = 1e9
        for _ in range(a.reps + 1):
            ctx.synchronize()
            t0 = time.perf_counter()
            fn()
            ctx.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        return best

    maps = []
    for rep in range(a.repeat_map):
        rows = []
        for i, c in enumerate(chunks):
            w = timed(lambda: ctx.generate(1, 0, n, 16, 12, c))
            r = timed(lambda: ctx.reduce(c, n, fetch=False))
            cp = timed(lambda: ctx.copy(c, src, 24 * n))
            rows.append({"map": rep, "chunk": i, "ptr": hex(c.ptr), "write_GBps": round(24 * n / w / 1e6, 1), "read_GBps": round(24 * n / r / 1e6, 1),
                         "copy_GBps": round(48 * n / cp / 1e6, 1)})
            print(json.dumps(rows[-1]), flush=True)
        maps.append(rows)
    def q(v, f):
        s = sorted(v)
        return s[min(len(s) - 1, int(f * len(s)))]
    summ = {"chunks": len(chunks), "chunk_mib": a.chunk_mib}
    for key in ("write_GBps", "read_GBps", "copy_GBps"):
        v = [r[key] for r in maps[0]]
        summ[key] = {"min": min(v), "q10": q(v, 0.1), "median": q(v, 0.5), "q90": q(v, 0.9), "max": max(v)}
        if len(maps) > 1:
            v1 = [r[key] for r in maps[1]]
            m0, m1 = sum(v) / len(v), sum(v1) / len(v1)
            sxy = sum((x - m0) * (y - m1) for x, y in zip(v, v1))
            sxx = sum((x - m0) ** 2 for x in v)
            syy = sum((y - m1) ** 2 for y in v1)
            summ[key]["corr_map0_map1"] = round(sxy / (sxx * syy) ** 0.5, 3) if sxx and syy else None
    print(json.dumps({"summary": summ}))
    for c in chunks:
        c.free()
    src.free()
    ctx.close()


if __name__ == "__main__":
    main()
