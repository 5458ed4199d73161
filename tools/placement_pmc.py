#!/usr/bin/env python3
"""Which memory-system counter follows the per-placement spread of the read+write kernels?

One process, CYCLES allocation cycles.  Every cycle allocates the five bench arrays afresh (hipMalloc through the C ABI;
allocation order and filler allocations vary per cycle so the physical pages differ), then launches
    generate, decode x R, encode x R, copy x R, reduce x R
and frees everything.  Run plainly it prints one JSON line per cycle with HIP-event times.  Run under
    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d DIR -- python3 tools/placement_pmc.py ...
the dispatch order is deterministic, so `--analyse DIR` joins kernel_trace.csv (durations) with counter_collection.csv
(counters) per dispatch and reports, per kernel, duration and counter per cycle plus their correlation over the cycles.
"""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KERNELS = ("decode", "encode", "copy", "reduce", "generate")


def short(name):
    for k in KERNELS:
        if f"ibu_k_{k}" in name and "_tail" not in name:
            return k
    return None


def analyse(d):
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not kt or not cc:
        raise SystemExit(f"no kernel_trace / counter_collection csv under {d}")
    dur = {}
    with open(kt[0]) as f:
        for r in csv.DictReader(f):
            dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    ctr = defaultdict(dict)
    with open(cc[0]) as f:
        for r in csv.DictReader(f):
            ctr[r["Dispatch_Id"]][r["Counter_Name"]] = ctr[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    per = defaultdict(list)  # kernel -> [(dispatch, ns, {counter: v})]
    for did in sorted(dur, key=int):
        k = short(dur[did][0])
        if k:
            per[k].append((int(did), dur[did][1], ctr.get(did, {})))
    out = {}
    for k, rows in per.items():
        names = sorted({c for _, _, cs in rows for c in cs})
        ns = [r[1] for r in rows]
        rec = {"launches": len(rows), "ms": [round(x / 1e6, 3) for x in ns], "counters": {}}
        for c in names:
            v = [r[2].get(c, 0.0) for r in rows]
            mn, mv = sum(ns) / len(ns), sum(v) / len(v)
            sxx = sum((a - mn) ** 2 for a in ns)
            syy = sum((b - mv) ** 2 for b in v)
            sxy = sum((a - mn) * (b - mv) for a, b in zip(ns, v))
            rec["counters"][c] = {"values": [round(x, 1) for x in v],
                                  "corr_with_ms": round(sxy / (sxx * syy) ** 0.5, 3) if sxx > 0 and syy > 0 else None,
                                  "per_ms": [round(b / (a / 1e6), 1) for a, b in zip(ns, v)]}
        out[k] = rec
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=float, default=1e9)
    ap.add_argument("--cycles", type=int, default=8)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--lens", default="16,12")
    ap.add_argument("--analyse", default="")
    ap.add_argument("--blocks", default="", help="comma list of blocks_per_cu caps: time decode / encode at each one in every cycle")
    ap.add_argument("--so", action="append", default=[], help="tag=path of another build: its decode / encode timed on the same arrays in every cycle")
    a = ap.parse_args()
    if a.analyse:
        return analyse(a.analyse)
    import ibu_amd as ia

    n = int(a.records)
    bc_len, umi_len = (int(x) for x in a.lens.split(","))
    ctx = ia.Context(0)
    sizes = {"recs": 24 * n, "bc": bc_len * n, "umi": umi_len * n, "idx": 8 * n, "back": 24 * n}
    orders = [["recs", "back", "bc", "umi", "idx"], ["idx", "umi", "bc", "back", "recs"], ["bc", "recs", "idx", "back", "umi"]]
    fillers = [0, 0, 3 << 30, 300_000_000, 17 << 30, 1 << 21, 40 << 30, 5_000_000_000]
    import time

    def timed(fn):
        ts = []
        for _ in range(a.reps):
            ctx.synchronize()
            t0 = time.perf_counter()
            fn()
            ctx.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return round(min(ts), 3)

    variants = []
    if a.so:
        import ctypes as C
        from ibu_amd import _lib
        for spec in a.so:
            tag, path = spec.split("=", 1)
            vl = _lib.load(path)
            vc = C.c_void_p()
            assert vl.ibu_ctx_create(0, C.byref(vc)) == 0
            variants.append((tag, vl, vc))
    held = []  # filler blocks that stay allocated across cycles: they fragment what later cycles get
    for cyc in range(a.cycles):
        order = orders[cyc % len(orders)]
        fill = fillers[cyc % len(fillers)]
        bufs, tmp = {}, []
        for k in order:
            if fill:
                tmp.append(ctx.alloc(fill // 8))
            bufs[k] = ctx.alloc(sizes[k])
        for t in tmp[1::2]:   # free every other small block: holes between the big arrays
            t.free()
        held.extend(tmp[0::2])
        ctx.generate(1, 0, n, bc_len, umi_len, bufs["recs"])
        line = {"cycle": cyc, "order": order, "filler": fill,
                "decode_ms": timed(lambda: ctx.decode_ascii(bufs["recs"], n, bc_len, umi_len, bufs["bc"], bufs["umi"], bufs["idx"])),
                "encode_ms": timed(lambda: ctx.encode_ascii(bufs["bc"], bufs["umi"], bufs["idx"], n, bc_len, umi_len, bufs["back"])),
                "copy_ms": timed(lambda: ctx.copy(bufs["back"], bufs["recs"], 24 * n)),
                "reduce_ms": timed(lambda: ctx.reduce(bufs["recs"], n)),
                "ptrs": {k: hex(v.ptr) for k, v in bufs.items()}}
        if a.blocks:   # does a slow placement prefer fewer resident waves?
            sweep = {}
            for bpc in (int(x) for x in a.blocks.split(",")):
                ctx.set_option("blocks_per_cu", bpc)
                sweep[bpc] = [timed(lambda: ctx.decode_ascii(bufs["recs"], n, bc_len, umi_len, bufs["bc"], bufs["umi"], bufs["idx"])),
                              timed(lambda: ctx.encode_ascii(bufs["bc"], bufs["umi"], bufs["idx"], n, bc_len, umi_len, bufs["back"]))]
            ctx.set_option("blocks_per_cu", 7)   # the library default
            line["blocks_sweep_decode_encode_ms"] = sweep
        for tag, vl, vc in variants:
            P = lambda k: C.c_void_p(bufs[k].ptr)
            def vdec():
                assert vl.ibu_decode_ascii(vc, P("recs"), n, bc_len, umi_len, P("bc"), P("umi"), P("idx"), None) == 0
                assert vl.ibu_ctx_synchronize(vc, None) == 0
            def venc():
                assert vl.ibu_encode_ascii(vc, P("bc"), P("umi"), P("idx"), 0, n, bc_len, umi_len, P("back"), None) == 0
                assert vl.ibu_ctx_synchronize(vc, None) == 0
            line[f"{tag}_decode_encode_ms"] = [timed(vdec), timed(venc)]
        print(json.dumps(line), flush=True)
        for b in bufs.values():
            b.free()
        if len(held) > 6:
            for h in held[:3]:
                h.free()
            del held[:3]
    ctx.close()


if __name__ == "__main__":
    main()
