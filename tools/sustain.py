#!/usr/bin/env python3
"""Per-launch kernel time when launches are queued back to back (no host sync in between) versus
isolated: separates a sustained-load effect (clocks / power) from a placement effect.
  python tools/sustain.py [--records 1e9] [--reps 12]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=float, default=1e9)
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--lens", default="16,12")
    ap.add_argument("--blocks", type=int, default=0)
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
    if a.blocks:
        assert lib.ibu_ctx_set_option(ctx, b"blocks_per_cu", a.blocks) == 0
    buf = lambda b: torch.empty(b, dtype=torch.uint8, device=dev)
    recs, back, bc, umi, idx = buf(24 * n), buf(24 * n), buf(bc_len * n), buf(umi_len * n), buf(8 * n)
    p = lambda t: C.c_void_p(t.data_ptr())
    dec = lambda: lib.ibu_decode_ascii(ctx, p(recs), n, bc_len, umi_len, p(bc), p(umi), p(idx), st)
    enc = lambda: lib.ibu_encode_ascii(ctx, p(bc), p(umi), p(idx), 0, n, bc_len, umi_len, p(back), st)
    red = lambda: lib.ibu_reduce(ctx, p(recs), n, st)
    assert lib.ibu_generate(ctx, 1, 0, n, bc_len, umi_len, p(recs), st) == 0
    dec(); enc(); red()
    torch.cuda.synchronize()

    def seq(tag, fns, sync_between=False, sleep=0.0):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(fns) + 1)]
        ev[0].record()
        for i, f in enumerate(fns):
            assert f() == 0
            ev[i + 1].record()
            if sync_between:
                ev[i + 1].synchronize()
                if sleep:
                    time.sleep(sleep)
                    ev[i + 1] = torch.cuda.Event(enable_timing=True)  # restart the clock after the pause
                    ev[i + 1].record()
        torch.cuda.synchronize()
        ms = []
        for i in range(len(fns)):
            try:
                ms.append(round(ev[i].elapsed_time(ev[i + 1]), 3))
            except RuntimeError:
                ms.append(None)
        print(json.dumps({"tag": tag, "ms": ms}), flush=True)

    r = a.reps
    seq("decode back-to-back", [dec] * r)
    seq("decode synced", [dec] * r, sync_between=True)
    seq("encode back-to-back", [enc] * r)
    seq("encode synced", [enc] * r, sync_between=True)
    seq("dec,enc alternating back-to-back", [dec, enc] * r)
    seq("dec,enc alternating synced", [dec, enc] * (r // 2), sync_between=True)
    seq("reduce back-to-back", [red] * r)
    seq("dec,red alternating back-to-back", [dec, red] * (r // 2))
    seq("decode after 50 ms idle each", [dec] * 6, sync_between=True, sleep=0.05)


if __name__ == "__main__":
    main()
