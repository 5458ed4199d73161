#!/usr/bin/env python3
"""Throughput of the device DEFLATE decoder (ibu_inflate_blocks_device) on the BGZF form of a records file.
  python tools/inflatebench.py [--records 1e8] [--lens 16,12] [--level 1] [--rounds 3]
A file of synthetic records is written from device memory, compressed to BGZF blocks on the host (the test-data writer of
tools/gzutil.py: what `bgzip -l LEVEL` writes), scanned (ibu_bgzf_scan), its COMPRESSED bytes are uploaded and every block is
inflated on the device; the result is checked with K4 against the resident records."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", default="1e8")
    ap.add_argument("--lens", default="16,12")
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--dir", default=None)
    a = ap.parse_args()
    import ibu_amd as ia
    from gzutil import bgzf_parallel
    n = int(float(a.records))
    bc_len, umi_len = (int(x) for x in a.lens.split(","))
    ctx = ia.Context(0)
    d = ctx.alloc(24 * n)
    ctx.generate(0x1B00004, 0, n, bc_len, umi_len, d)
    want = ctx.reduce(d, n)
    with tempfile.TemporaryDirectory(dir=a.dir) as td:
        plain, bg = os.path.join(td, "r.bin"), os.path.join(td, "r.bgzf")
        with open(plain, "wb") as f:                                   # the records alone (no header): the blocks then hold whole bytes of them
            step = 1 << 24
            for r0 in range(0, n, step):
                k = min(step, n - r0)
                f.write(ia.DeviceBuffer.wrap(ctx, d.ptr + 24 * r0, 24 * k).download().tobytes())
        t0 = time.perf_counter()
        gz_bytes = bgzf_parallel(plain, bg, level=a.level, workers=min(16, os.cpu_count() or 1))
        t_comp = time.perf_counter() - t0
        comp = np.fromfile(bg, dtype=np.uint8)
    t0 = time.perf_counter()
    blocks, consumed, out_bytes, rc = ia.bgzf_scan(comp)
    t_scan = time.perf_counter() - t0
    assert rc == 0 and consumed == len(comp) and out_bytes == 24 * n, (rc, consumed, out_bytes)
    d_comp = ctx.alloc(len(comp) + 2048)
    d_comp.upload(comp)
    d_out = ctx.alloc(24 * n)
    ts = []
    for _ in range(a.rounds + 1):
        ctx.synchronize()
        t0 = time.perf_counter()
        st, first = ctx.inflate_blocks(d_comp, blocks, d_out)          # (uploads the 32-byte descriptors, downloads the status words)
        ts.append(time.perf_counter() - t0)
        assert first is None and not st.any(), first
    got = ctx.reduce(d_out, n)
    sec = min(ts[1:])
    print(json.dumps({"records": n, "lens": [bc_len, umi_len], "level": a.level, "bgzf_bytes": int(gz_bytes), "ratio": round(gz_bytes / (24 * n), 4),
                      "blocks": len(blocks), "scan_seconds": round(t_scan, 4), "inflate_seconds": round(sec, 5), "all_rounds": [round(t, 5) for t in ts],
                      "out_GBps": round(24 * n / sec / 1e9, 1), "in_GBps": round(gz_bytes / sec / 1e9, 1), "records_per_s": round(n / sec),
                      "equal_to_resident_records": got == want, "host_compress_seconds": round(t_comp, 2)}), flush=True)
    assert got == want


if __name__ == "__main__":
    main()
