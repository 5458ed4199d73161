#!/usr/bin/env python3
"""Throughput of ibu_sort_records on device-resident records.
  python tools/sortbench.py [--records 1e8,1e9] [--lens 16,12] [--rounds 3]"""
import argparse
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


def shards_main(a, ia, bc_len, umi_len):
    k = a.shards
    ctxs = [ia.Context(0) for _ in range(k)]
    if a.sort_first:                                           # the round-3 form of the call (every shard sorted, exchanged, sorted again)
        ctxs[0].set_option("sort_compact", 0)
    for n in (int(float(x)) for x in a.records.split(",")):
        per, cap = n // k, int(n // k * 1.25) + k + 1
        bufs = [(c.alloc(24 * cap), c.alloc(24 * cap)) for c in ctxs]
        ts, outs = [], None
        for _ in range(a.rounds + 1):
            for i, (c, (d, t)) in enumerate(zip(ctxs, bufs)):     # shard i = rows [i per, (i + 1) per) of the generator's stream
                c.generate(0x1B00005, i * per, per, bc_len, umi_len, d)
                c.synchronize()
            t0 = time.perf_counter()
            outs = ia.Context.sort_records_contexts(ctxs, [(d, t, per, cap) for d, t in bufs])
            ts.append(time.perf_counter() - t0)
        assert sum(outs) == per * k and all(c.is_sorted(d, m) for c, (d, _), m in zip(ctxs, bufs, outs))
        sec = statistics.median(ts[1:])
        print(json.dumps({"n": per * k, "lens": [bc_len, umi_len], "contexts_on_device_0": k, "form": "sort first (forced)" if a.sort_first else "default",
                          "seconds": round(sec, 4),
                          "M_records_per_s": round(per * k / sec / 1e6, 1), "records_per_shard_after": outs,
                          "note": "one GPU: the K local sorts share it and the exchange is a device-local copy"}), flush=True)
        for d, t in bufs:
            d.free()
            t.free()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", default="1e8")
    ap.add_argument("--lens", default="16,12")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--random-index", action="store_true",
                    help="index column in random order (default: increasing = records in read order, whose index passes are skipped)")
    ap.add_argument("--variants", default="0", help="comma list of sort_variant values to time (A/B in one process)")
    ap.add_argument("--compact", default="", help="comma list of sort_compact values to time (0 = 24-byte passes only, k = compact tile shape k; "
                                                  "default: the library's default)")
    ap.add_argument("--skip-agg", action="store_true")
    ap.add_argument("--guess", default="", help="comma list of sort_guess values to time (1 = the library default: speculation from 2^23 records; k = from k records)")
    ap.add_argument("--hybrid", default="", help="comma list of sort_hybrid values (0 = all passes, 1 = default)")
    ap.add_argument("--whitelist", type=int, default=0,
                    help="K > 0: barcodes drawn from K distinct ones with a skewed distribution (rank ~ K u^3: a single-cell run, where a few "
                         "thousand cells hold most reads), random UMIs, records in read order; built on the device with torch")
    ap.add_argument("--sort-first", action="store_true", help="with --shards: force the round-3 form of ibu_sort_records_contexts (sort_compact = 0 on the first context)")
    ap.add_argument("--shards", type=int, default=0,
                    help="K > 0: time ibu_sort_records_contexts over K contexts instead — all on device 0 (the one-GPU rehearsal of the "
                         "multi-GPU sort: K local sorts, the device-to-device exchange, K sorts of what arrived), n / K records each")
    ap.add_argument("--presorted", action="store_true", help="also time ibu_sort_records on the SORTED result (the already-sorted fast exit: one read-only census)")
    a = ap.parse_args()
    if a.whitelist:
        import torch                                         # before the library, as bench.py does: its HIP runtime is the process's
        torch.cuda.init()
    import ibu_amd as ia

    bc_len, umi_len = (int(x) for x in a.lens.split(","))
    if a.shards:
        return shards_main(a, ia, bc_len, umi_len)
    ctx = ia.Context(0)
    for n in (int(float(x)) for x in a.records.split(",")):
        d, t = ctx.alloc(24 * n), ctx.alloc(24 * n)
        cols = [ctx.alloc(8 * n) for _ in range(4)] if (a.random_index or a.whitelist) else None
        if a.whitelist:
            g = torch.Generator(device="cuda").manual_seed(0x1B00007)
            wl = _rand_bits(torch, g, 2 * bc_len, a.whitelist)
            bc, um, ix = (torch.as_tensor(c, device="cuda").view(torch.int64) for c in cols[:3])
            step = 1 << 26
            for lo in range(0, n, step):                     # in pieces: the temporaries stay small
                hi = min(n, lo + step)
                u = torch.rand(hi - lo, generator=g, device="cuda", dtype=torch.float64)
                bc[lo:hi] = wl[(u * u * u * a.whitelist).to(torch.int64).clamp_(max=a.whitelist - 1)]
                um[lo:hi] = _rand_bits(torch, g, 2 * umi_len, hi - lo)
                ix[lo:hi] = (torch.randint(0, 1 << 30, (hi - lo,), generator=g, device="cuda", dtype=torch.int64) if a.random_index
                             else torch.arange(lo, hi, device="cuda", dtype=torch.int64))
            del u
            torch.cuda.synchronize()
        for variant, compact, guess, hybrid in ((int(v), c, g, h) for v in a.variants.split(",") for c in (a.compact.split(",") if a.compact else [None])
                                                for g in (a.guess.split(",") if a.guess else [None]) for h in (a.hybrid.split(",") if a.hybrid else [None])):
            ctx.set_option("sort_variant", variant)
            if guess is not None:
                ctx.set_option("sort_guess", int(guess))
            if hybrid is not None:
                ctx.set_option("sort_hybrid", int(hybrid))
            if compact is not None:
                ctx.set_option("sort_compact", int(compact))
            ts = []
            for _ in range(a.rounds + 1):
                if a.whitelist:
                    ctx.serialize(cols[0], cols[1], cols[2], n, d)
                else:
                    ctx.generate(0x1B00005, 0, n, bc_len, umi_len, d)  # random barcode/UMI, increasing index: unsorted
                if a.random_index and not a.whitelist:  # replace the index column by random 30-bit values (another stream's barcode column)
                    ctx.deserialize(d, n, cols[0], cols[1], cols[2])
                    ctx.generate(0x1B00006, 0, n, 15, 1, t)
                    ctx.deserialize(t, n, cols[3], cols[2], cols[2])
                    ctx.serialize(cols[0], cols[1], cols[3], n, d)
                ctx.synchronize()
                t0 = time.perf_counter()
                ctx.sort_records(d, t, n)
                ctx.synchronize()
                ts.append(time.perf_counter() - t0)
            assert os.environ.get("IBU_HIP_SO") or ctx.is_sorted(d, n)  # probe builds (IBU_HIP_SO=...): wrong output by design; variants >= 4 exist in probe builds only: wrong output by design
            agg, nb = None, None
            if not a.skip_agg:
                t0 = time.perf_counter()
                bcs, counts, uniq = ctx.barcode_counts(d, n)  # BarcodeAnalyzer on the sorted records (size query + emit + download)
                agg = round(time.perf_counter() - t0, 4)
                assert int(counts.sum()) == n
                nb = int(len(bcs))
            presorted = None
            if a.presorted:
                tp = []
                for _ in range(a.rounds + 1):
                    ctx.synchronize()
                    t0 = time.perf_counter()
                    ctx.sort_records(d, t, n)
                    ctx.synchronize()
                    tp.append(time.perf_counter() - t0)
                presorted = round(statistics.median(tp[1:]) * 1e3, 3)
            sec = statistics.median(ts[1:])
            idx_bytes = 4 if a.random_index else 0  # 30 random bits -> 4 digit passes; index-ordered input -> skipped
            passes = (2 * bc_len + 7) // 8 + (2 * umi_len + 7) // 8 + idx_bytes
            # algorithmic traffic: census 24 + histogram 24 once, 48 per pass, 48 for the copy back after an odd number of passes
            alg = n * (48 + 48 * passes + (48 if passes & 1 else 0))
            print(json.dumps({"n": n, "lens": [bc_len, umi_len], "variant": variant, "compact": compact if compact is None else int(compact), "guess": guess, "hybrid": hybrid, "index": "random" if a.random_index else "increasing (read order)", "whitelist": a.whitelist or None,
                              "seconds": round(sec, 4), "best": round(min(ts[1:]), 4), "M_records_per_s": round(n / sec / 1e6, 1),
                              "passes": passes, "presorted_input_ms": presorted, "barcode_counts_seconds": agg, "distinct_barcodes": nb,
                              "algorithmic_GB": round(alg / 1e9, 1), "GBps_algorithmic": round(alg / sec / 1e9)}), flush=True)
        d.free()
        t.free()
        for c in cols or []:
            c.free()


if __name__ == "__main__":
    main()
