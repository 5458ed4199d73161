#!/usr/bin/env python3
"""Kernel micro-benchmark / A-B harness.

Times the hot-path kernels with HIP events on one stream.  Variants (different builds of
libibu_hip.so and/or different residency caps) are run in interleaved rounds inside ONE
process (cdna_hip_programming.md rule 24), all on the same buffers.

  python tools/kbench.py [--records 2e8] [--lens "16,12 31,31 ..."] [--rounds 7] [--kernels decode,encode,...]
                         [--so tag=path ...] [--blocks 8,7,6]
Prints one JSON line per (variant, blocks, kernel): median / best ms and algorithmic GB/s."""
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
    ap.add_argument("--records", type=float, default=2e8)
    ap.add_argument("--lens", default="16,12")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--kernels", default="decode,encode,deserialize,serialize,reduce,unpack,pack,generate,copy")
    ap.add_argument("--so", action="append", default=[], help="tag=path of an alternative build (repeatable)")
    ap.add_argument("--blocks", default="", help="comma list of blocks_per_cu caps to sweep (default: library default)")
    ap.add_argument("--alloc-probe-tries", type=int, default=0,
                    help="N > 1: the arrays come from ibu_device_alloc under the context option alloc_probe_tries = N (placement probing as a "
                         "library property) instead of torch.empty")
    a = ap.parse_args()
    import torch

    from ibu_amd import _lib

    n = int(a.records)
    pairs = [tuple(int(x) for x in pr.split(",")) for pr in a.lens.replace(";", " ").split()]
    bc_max, umi_max = max(p_[0] for p_ in pairs), max(p_[1] for p_ in pairs)
    dev = torch.device("cuda", 0)
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    st = C.c_void_p(ts.cuda_stream)
    buf = lambda b: torch.empty(b, dtype=torch.uint8, device=dev)
    if a.alloc_probe_tries > 1:
        alib = _lib.load(_lib.SO_PATH)
        actx = C.c_void_p()
        assert alib.ibu_ctx_create(0, C.byref(actx)) == 0
        assert alib.ibu_ctx_set_option(actx, b"alloc_probe_tries", a.alloc_probe_tries) == 0

        class LibBuf:                                           # the two members of a tensor this script uses
            def __init__(self, nbytes):
                self.ptr, self.nbytes = C.c_void_p(), nbytes
                assert alib.ibu_device_alloc(actx, nbytes, C.byref(self.ptr)) == 0

            def data_ptr(self):
                return self.ptr.value

            def numel(self):
                return self.nbytes

        buf = LibBuf
    recs, back = buf(n * 24), buf(n * 24)
    bc, umi, idx = buf(n * bc_max), buf(n * umi_max), buf(n * 8)   # ONE set of arrays for every pair of lengths: same placement
    c0, c1 = buf(n * 8), buf(n * 8)
    p = lambda t: C.c_void_p(t.data_ptr())

    variants = [("default", _lib.SO_PATH)] + [tuple(s.split("=", 1)) for s in a.so]
    blocks = [int(b) for b in a.blocks.split(",") if b] or [0]
    cfgs = []
    for tag, path in variants:
        lib = _lib.load(path)
        ctx = C.c_void_p()
        assert lib.ibu_ctx_create(0, C.byref(ctx)) == 0, tag
        cfgs.append((tag, lib, ctx))

    def ops_for(lib, ctx, bc_len, umi_len):
        return {
            "decode": (lambda: lib.ibu_decode_ascii(ctx, p(recs), n, bc_len, umi_len, p(bc), p(umi), p(idx), st), 24 + bc_len + umi_len + 8),
            "encode": (lambda: lib.ibu_encode_ascii(ctx, p(bc), p(umi), p(idx), 0, n, bc_len, umi_len, p(back), st), 24 + bc_len + umi_len + 8),
            "deserialize": (lambda: lib.ibu_deserialize(ctx, p(recs), n, p(c0), p(c1), p(idx), st), 48),
            "serialize": (lambda: lib.ibu_serialize(ctx, p(c0), p(c1), p(idx), n, p(back), st), 48),
            "reduce": (lambda: lib.ibu_reduce(ctx, p(recs), n, st), 24),
            "unpack": (lambda: lib.ibu_unpack_2bit(ctx, p(c0), n, bc_len, p(bc), st), 8 + bc_len),
            "pack": (lambda: lib.ibu_pack_2bit(ctx, p(bc), n, bc_len, p(c1), st), 8 + bc_len),
            "generate": (lambda: lib.ibu_generate(ctx, 1, 0, n, bc_len, umi_len, p(back), st), 24),
            "copy": (lambda: lib.ibu_device_copy(ctx, p(back), p(recs), 24 * n, st), 48),
            "mismatch": (lambda: lib.ibu_records_first_mismatch(ctx, p(recs), p(back), n, C.byref(C.c_uint64()), st), 48),
        }

    tag0, lib0, ctx0 = cfgs[0]
    print(json.dumps({"gpu": torch.cuda.get_device_name(dev), "uuid": str(getattr(torch.cuda.get_device_properties(dev), "uuid", ""))}), flush=True)
    for bc_len, umi_len in pairs:
        assert lib0.ibu_generate(ctx0, 1, 0, n, bc_len, umi_len, p(recs), st) == 0
        assert lib0.ibu_decode_ascii(ctx0, p(recs), n, bc_len, umi_len, p(bc), p(umi), p(idx), st) == 0
        assert lib0.ibu_deserialize(ctx0, p(recs), n, p(c0), p(c1), p(idx), st) == 0
        torch.cuda.synchronize()

        # measurement-only read:write mix kernels (tools/native/hbm_mix.hip), timed in the same process as a yardstick
        mix_so = os.path.join(ROOT, "tools", "native", "libhbm_mix.so")
        mix_ops = {}
        if os.path.exists(mix_so):
            mix = C.CDLL(mix_so)
            mix.hbm_mix.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
            span = min(recs.numel(), back.numel())
            for tag_, r_, w_ in (("mix10", 1, 0), ("mix30", 3, 0), ("mix11", 1, 1), ("mix23", 2, 3), ("mix32", 3, 2),
                                   ("pipe11", 11, 1), ("pipe23", 12, 3), ("pipe32", 13, 2)):  # r_ >= 10: software-pipelined form
                steps = span // 16 // max(r_ % 10, w_) // 4 * 4
                mix_ops[tag_] = (lambda r_=r_, w_=w_, steps=steps: mix.hbm_mix(r_, w_, recs.data_ptr(), back.data_ptr(), steps, 256 * 7, st),
                                 (r_ % 10 + w_) * 16 * steps / n)
            # the yardstick of the single-column pack ON PACK'S OWN ARRAYS (bc -> c1, same placement): bc_len bytes read, 8 written
            # per row as R : W plain 16-byte streams.  31 bases has no small ratio: 4 : 1 (32 bytes read) is the nearest, its
            # rate is computed on the bytes it really moves.
            ratio = {8: (1, 1), 12: (3, 2), 16: (2, 1), 32: (4, 1), 31: (4, 1), 5: (5, 8), 7: (7, 8), 20: (5, 2)}.get(bc_len)
            if ratio:
                r_, w_ = ratio
                steps = min(n * bc_len // 16 // r_, n * 8 // 16 // w_) // 4 * 4   # both streams inside their arrays (31 bases: 4 : 1 over 31/32 of the rows)
                assert r_ * 16 * steps <= n * bc_len and w_ * 16 * steps <= n * 8, "yardstick would leave its arrays"
                for nb in (7, 8, 4):
                    mix_ops[f"ypack{nb}"] = (lambda r_=r_, w_=w_, steps=steps, nb=nb: mix.hbm_mix(r_, w_, bc.data_ptr(), c1.data_ptr(), steps, 256 * nb, st),
                                             (r_ + w_) * 16 * steps / n)
                    if (w_, r_) in ((1, 1), (2, 3), (1, 2), (1, 4)) and umi_max >= bc_len:   # unpack's mix: codes -> the UMI column's array (bc keeps valid ASCII for pack)
                        mix_ops[f"yunpack{nb}"] = (lambda r_=r_, w_=w_, steps=steps, nb=nb: mix.hbm_mix(w_, r_, c0.data_ptr(), umi.data_ptr(), steps, 256 * nb, st),
                                                   (r_ + w_) * 16 * steps / n)

        names = [k for k in a.kernels.split(",") if k in ops_for(lib0, ctx0, bc_len, umi_len)]
        runs = []  # (tag, blocks, kernel, fn, bytes_per_record, lib, ctx)
        for tag, lib, ctx in cfgs:
            ops = ops_for(lib, ctx, bc_len, umi_len)
            for b in blocks:
                for k in names:
                    runs.append((tag, b, k, ops[k][0], ops[k][1], lib, ctx))
        for k in a.kernels.split(","):
            if k in mix_ops:
                runs.append(("mix", 0, k, mix_ops[k][0], mix_ops[k][1], lib0, ctx0))
        times = {(r[0], r[1], r[2]): [] for r in runs}

        def run(r):
            tag, b, k, fn, _, lib, ctx = r
            if b:
                assert lib.ibu_ctx_set_option(ctx, b"blocks_per_cu", b) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn()
            e1.record()
            e1.synchronize()
            assert rc == 0, (tag, k, rc)
            return e0.elapsed_time(e1)

        for r in runs:  # warm-up (module load, occupancy query)
            run(r)
        for _ in range(a.rounds):
            for r in runs:
                times[(r[0], r[1], r[2])].append(run(r))
        for tag, lib, ctx in cfgs:
            if lib.ibu_codec_status(ctx, st, None, None) != 0:  # expected only with IBU_PROBE builds (their output is wrong)
                print(json.dumps({"note": f"codec status of {tag!r} reports invalid rows (probe build in the mix?)"}), flush=True)
        for r in runs:
            t = times[(r[0], r[1], r[2])]
            med, mn = statistics.median(t), min(t)
            print(json.dumps({"tag": r[0], "blocks_per_cu": r[1] or "default", "kernel": r[2], "n": n, "lens": [bc_len, umi_len],
                              "ms_med": round(med, 4), "ms_min": round(mn, 4), "GBps_med": round(n * r[4] / med / 1e6, 1),
                              "GBps_best": round(n * r[4] / mn / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
