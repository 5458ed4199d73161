#!/usr/bin/env python3
"""Distributed sort of a record stream across ranks (one process per GPU): every rank generates its shard on device,
ibu_amd.sharding.distributed_sort runs the sample sort (local radix sort, splitters, ONE all_to_all_single of the
records — RCCL over xGMI with --backend nccl — local radix sort), then the result is checked: every rank sorted,
rank boundaries ordered, global count / sums / XORs preserved.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/sharded_sort.py
      [--records 1e8] [--backend nccl|gloo] [--share-gpu]
Rank 0 prints one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=float, default=1e8, help="records per rank")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="one rank, but every collective runs (RCCL rehearsal on a one-GPU box)")
    ap.add_argument("--lens", default="16,12")
    ap.add_argument("--no-compact", action="store_true", help="ship 24-byte records even when 12-byte elements would do")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist

    import ibu_amd as ia
    from ibu_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if a.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or a.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29532")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    coll_dev = dev if a.backend == "nccl" else torch.device("cpu")
    ctx = ia.Context(local)
    n = int(a.records)
    bc_len, umi_len = (int(x) for x in a.lens.split(","))
    first, end = sharding.rank_shard(n * world, world, rank)
    buf = torch.empty(n * 24, dtype=torch.uint8, device=dev)
    ctx.generate(0x1B00007, first, n, bc_len, umi_len, buf)
    before = sharding.global_totals(ctx.reduce(buf, n), device=coll_dev, force=a.force_dist)
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    stats = {}
    out, n_out = sharding.distributed_sort(sharding.DeviceSortOps(ctx), buf, n, stats=stats, force=a.force_dist, compact=not a.no_compact)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    ok = ctx.is_sorted(out, n_out) if n_out else True
    ops = sharding.DeviceSortOps(ctx)
    edge = (ops.fetch(out, 0), ops.fetch(out, n_out - 1)) if n_out else None
    edges = [None] * world
    if use_dist:
        dist.all_gather_object(edges, (ok, n_out, edge, stats.get("sent_bytes", 0), stats.get("exchange_seconds", 0.0)))  # result check only
    else:
        edges = [(ok, n_out, edge, 0, 0.0)]
    after = sharding.global_totals(ctx.reduce(out, n_out), device=coll_dev, force=a.force_dist)
    if rank == 0:
        keys = [(sharding._rec_key(e[2][0]), sharding._rec_key(e[2][1])) for e in edges if e[1]]
        ordered = all(keys[i][1] <= keys[i + 1][0] for i in range(len(keys) - 1))
        print(json.dumps({"ranks": world, "records_per_rank_in": n, "records_per_rank_out": [e[1] for e in edges],
                          "seconds": round(dt, 4), "M_records_per_s": round(n * world / dt / 1e6, 1),
                          "backend": a.backend, "transport": ("RCCL" if a.backend == "nccl" else "gloo, staged through the host (rehearsal)"),
                          "bytes_per_record_on_the_wire": stats.get("bytes_per_record_on_the_wire"), "varying_key_bytes": stats.get("varying_key_bytes"),
                          "exchange_sent_bytes_per_rank": [e[3] for e in edges], "exchange_seconds_max": round(max(e[4] for e in edges), 4),
                          "exchange_GBps_per_rank": round(max(e[3] for e in edges) / max(max(e[4] for e in edges), 1e-9) / 1e9, 2),
                          "every_rank_sorted": all(e[0] for e in edges), "rank_ranges_ordered": ordered,
                          "multiset_preserved": before == after, "count": after["count"]}), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
