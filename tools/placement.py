#!/usr/bin/env python3
"""Does decode/encode throughput depend on where the five arrays sit in HBM?  Same kernel, same n,
different allocation orders / pads / one slab with staggered offsets — all in one process.
  python tools/placement.py [--records 1e9] [--rounds 5]"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=float, default=1e9)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--lens", default="16,12")
    a = ap.parse_args()
    import torch

    from ibu_amd import _lib

    lib = _lib.load(_lib.SO_PATH)
    n = int(a.records)
    bc_len, umi_len = (int(x) for x in a.lens.split(","))
    dev = torch.device("cuda", 0)
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    st = C.c_void_p(ts.cuda_stream)
    ctx = C.c_void_p()
    assert lib.ibu_ctx_create(0, C.byref(ctx)) == 0
    sizes = {"recs": 24 * n, "bc": bc_len * n, "umi": umi_len * n, "idx": 8 * n, "back": 24 * n}

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        ts_ = []
        for _ in range(a.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert fn() == 0
            e1.record()
            e1.synchronize()
            ts_.append(e0.elapsed_time(e1))
        return statistics.median(ts_)

    def run(tag, ptrs):
        p = {k: C.c_void_p(v) for k, v in ptrs.items()}
        assert lib.ibu_generate(ctx, 1, 0, n, bc_len, umi_len, p["recs"], st) == 0
        dec = timeit(lambda: lib.ibu_decode_ascii(ctx, p["recs"], n, bc_len, umi_len, p["bc"], p["umi"], p["idx"], st))
        enc = timeit(lambda: lib.ibu_encode_ascii(ctx, p["bc"], p["umi"], p["idx"], 0, n, bc_len, umi_len, p["back"], st))
        red = timeit(lambda: lib.ibu_reduce(ctx, p["recs"], n, st))
        b = n * (24 + bc_len + umi_len + 8)
        print(json.dumps({"tag": tag, "n": n, "decode_ms": round(dec, 3), "decode_GBps": round(b / dec / 1e6),
                          "encode_ms": round(enc, 3), "encode_GBps": round(b / enc / 1e6),
                          "reduce_GBps": round(24 * n / red / 1e6),
                          "ptrs": {k: hex(v) for k, v in ptrs.items()}}), flush=True)

    def separate(order, pad=0):
        keep, ptrs = [], {}
        for k in order:
            t = torch.empty(sizes[k], dtype=torch.uint8, device=dev)
            keep.append(t)
            ptrs[k] = t.data_ptr()
            if pad:
                keep.append(torch.empty(pad, dtype=torch.uint8, device=dev))
        return keep, ptrs

    def slab(order, stagger):
        total = sum(sizes.values()) + (len(order) + 1) * (1 << 30)
        t = torch.empty(total, dtype=torch.uint8, device=dev)
        base = (t.data_ptr() + (1 << 21) - 1) & ~((1 << 21) - 1)
        ptrs, off = {}, 0
        for i, k in enumerate(order):
            ptrs[k] = base + off + i * stagger
            off += (sizes[k] + (1 << 21) - 1) & ~((1 << 21) - 1)
            off += 1 << 21
        return [t], ptrs

    cases = [
        ("bench-order", lambda: separate(["recs", "back", "bc", "umi", "idx"])),
        ("reverse-order", lambda: separate(["idx", "umi", "bc", "back", "recs"])),
        ("pad-1GiB", lambda: separate(["recs", "back", "bc", "umi", "idx"], pad=1 << 30)),
        ("pad-300MB", lambda: separate(["recs", "back", "bc", "umi", "idx"], pad=300_000_000)),
        ("slab-2MiB-aligned", lambda: slab(["recs", "back", "bc", "umi", "idx"], 0)),
        ("slab-stagger-4KiB", lambda: slab(["recs", "back", "bc", "umi", "idx"], 4096)),
        ("slab-stagger-272KiB", lambda: slab(["recs", "back", "bc", "umi", "idx"], 272 * 1024)),
        ("bench-order-again", lambda: separate(["recs", "back", "bc", "umi", "idx"])),
    ]
    for tag, mk in cases:
        keep, ptrs = mk()
        run(tag, ptrs)
        del keep
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
