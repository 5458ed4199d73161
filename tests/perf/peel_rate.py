#!/usr/bin/env python3
"""Wall-clock form of the peel check (moved out of `pytest -m gpu` in round 3: a 1.2x ratio between two timed runs is a
flake on a pool whose allocations differ by 20 %; tests/test_gpu_parity.py::test_odd_record_shard_takes_the_tiled_kernels
asserts on the path taken instead).  Run by hand on the GPU box:

  python tests/perf/peel_rate.py [records]

Prints one JSON line: decode + encode + reduce of a shard that starts at record 0 / 1 / 3 of a larger buffer."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ibu_amd  # noqa: E402


def main():
    n, bc_len, umi_len = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000, 16, 12
    ctx = ibu_amd.Context(0)
    recs, back = ctx.alloc((n + 4) * 24), ctx.alloc((n + 4) * 24)
    bc, umi, idx = ctx.alloc((n + 4) * bc_len), ctx.alloc((n + 4) * umi_len), ctx.alloc((n + 4) * 8)
    ctx.generate(0x1B00002, 0, n + 4, bc_len, umi_len, recs)

    def run(k):
        def once():
            ctx.decode_ascii(recs.ptr + 24 * k, n, bc_len, umi_len, bc.ptr + k * bc_len, umi.ptr + k * umi_len, idx.ptr + 8 * k)
            ctx.encode_ascii(bc.ptr + k * bc_len, umi.ptr + k * umi_len, idx.ptr + 8 * k, n, bc_len, umi_len, back.ptr + 24 * k)
            ctx.reduce(recs.ptr + 24 * k, n)
        once()
        ctx.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            once()
            ctx.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best

    t = {k: run(k) for k in (0, 1, 3)}
    print(json.dumps({"records": n, "seconds_aligned": t[0], "seconds_at_record_1": t[1], "seconds_at_record_3": t[3],
                      "ratio_1": t[1] / t[0], "ratio_3": t[3] / t[0]}))
    ctx.close()


if __name__ == "__main__":
    main()
