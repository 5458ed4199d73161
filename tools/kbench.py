#!/usr/bin/env python3
"""Kernel micro-benchmark: every hot-path kernel at one size, HIP-event timed on one stream,
interleaved rounds (A/B deltas come from one process, cdna_hip_programming.md rule 24).
  python tools/kbench.py [--records 2e8] [--lens 16,12] [--rounds 5] [--kernels decode,encode,...]
Prints one JSON line per kernel: median / min ms and algorithmic GB/s."""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=float, default=2e8)
    ap.add_argument("--lens", default="16,12")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--kernels", default="decode,encode,deserialize,serialize,reduce,unpack,pack,generate")
    ap.add_argument("--tag", default=os.environ.get("IBU_HIP_SO", "default"))
    a = ap.parse_args()
    import torch

    import ibu_amd

    n = int(a.records)
    bc_len, umi_len = (int(x) for x in a.lens.split(","))
    dev = torch.device("cuda", 0)
    ctx = ibu_amd.Context(0)
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    st = ts.cuda_stream
    buf = lambda b: torch.empty(b, dtype=torch.uint8, device=dev)
    recs, back = buf(n * 24), buf(n * 24)
    bc, umi, idx = buf(n * bc_len), buf(n * umi_len), buf(n * 8)
    c0, c1 = buf(n * 8), buf(n * 8)
    ctx.generate(1, 0, n, bc_len, umi_len, recs, stream=st)
    ctx.decode_ascii(recs, n, bc_len, umi_len, bc, umi, idx, stream=st)
    ctx.deserialize(recs, n, c0, c1, idx, stream=st)
    torch.cuda.synchronize()
    ops = {
        "decode": (lambda: ctx.decode_ascii(recs, n, bc_len, umi_len, bc, umi, idx, stream=st), 24 + bc_len + umi_len + 8),
        "encode": (lambda: ctx.encode_ascii(bc, umi, idx, n, bc_len, umi_len, back, stream=st), 24 + bc_len + umi_len + 8),
        "deserialize": (lambda: ctx.deserialize(recs, n, c0, c1, idx, stream=st), 48),
        "serialize": (lambda: ctx.serialize(c0, c1, idx, n, back, stream=st), 48),
        "reduce": (lambda: ctx.reduce(recs, n, stream=st, reset=False, fetch=False), 24),
        "unpack": (lambda: ctx.unpack_2bit(c0, n, bc_len, bc, stream=st), 8 + bc_len),
        "pack": (lambda: ctx.pack_2bit(bc, n, bc_len, c1, stream=st), 8 + bc_len),
        "generate": (lambda: ctx.generate(1, 0, n, bc_len, umi_len, back, stream=st), 24),
    }
    names = [k for k in a.kernels.split(",") if k in ops]
    times = {k: [] for k in names}
    for k in names:  # warm-up
        ops[k][0]()
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for k in names:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops[k][0]()
            e1.record()
            e1.synchronize()
            times[k].append(e0.elapsed_time(e1))
    ctx.codec_status(stream=st)
    for k in names:
        med, mn = statistics.median(times[k]), min(times[k])
        print(json.dumps({"tag": a.tag, "kernel": k, "n": n, "lens": [bc_len, umi_len], "ms_med": round(med, 4),
                          "ms_min": round(mn, 4), "GBps_med": round(n * ops[k][1] / med / 1e6, 1),
                          "GBps_best": round(n * ops[k][1] / mn / 1e6, 1),
                          "blocks_per_cu": os.environ.get("IBU_BLOCKS_PER_CU", "8")}), flush=True)


if __name__ == "__main__":
    main()
