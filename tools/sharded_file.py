#!/usr/bin/env python3
"""BASELINE configs[3] end to end: ONE file, MmapReader range-sharded across the ranks (one process per GPU,
mmap.rs:297-307 split), each rank streaming its shard through its pinned ring to its GPU; the only cross-rank
traffic is the {count, sums} all-reduce + XOR all-gather of ibu_amd.sharding.  PCIe-inclusive — not bench.py's value.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/sharded_file.py FILE
      [--proc reduce|decode] [--backend nccl|gloo] [--share-gpu] [--make RECORDS]
Rank 0 prints one JSON line.  --make writes FILE first (rank 0, counter-based records) if it does not exist."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--proc", default="reduce", choices=["reduce", "decode"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0 (with --backend gloo)")
    ap.add_argument("--make", type=float, default=0, help="records to generate into PATH when it is missing")
    ap.add_argument("--lens", default="16,12")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist

    import ibu_amd as ia
    from ibu_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if a.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(local)
    coll_dev = torch.device("cuda", local) if a.backend == "nccl" else torch.device("cpu")
    ctx = ia.Context(local)
    bc_len, umi_len = (int(x) for x in a.lens.split(","))

    if rank == 0 and a.make and not os.path.exists(a.path):
        n = int(a.make)
        d = ctx.alloc(24 * n)
        ctx.generate(0x1B00003, 0, n, bc_len, umi_len, d)
        w = ia.Writer.from_path(a.path, ia.Header(bc_len, umi_len))
        w.write_batch_device(ctx, d, n)
        w.finish()
        w.close()
        d.free()
    if world > 1:
        dist.barrier()

    m = ia.MmapReader.new(a.path)
    n_global, h = m.len(), m.header()
    first, end = sharding.rank_shard(n_global, world, rank)
    k = end - first
    sink = None
    if a.proc == "decode":
        keep = [ctx.alloc(max(k, 1) * h.bc_len), ctx.alloc(max(k, 1) * h.umi_len), ctx.alloc(max(k, 2) * 8)]
        sink = tuple(keep)
    m.process_device(ctx, ia.PROC_REDUCE, shard=rank, n_shards=world)  # warm-up: ring allocation, page cache
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    if a.proc == "reduce":
        red, st = m.process_device(ctx, ia.PROC_REDUCE, shard=rank, n_shards=world)
    else:
        _, st = m.process_device(ctx, ia.PROC_DECODE, shard=rank, n_shards=world, sink=sink)
        # totals of what was decoded: the index column is on the device; count and index sum come from it
        idx = sink[2].download(np.uint64, count=k)
        red = {"count": k, "sum": [0, 0, int(idx.sum(dtype=np.uint64)) if k else 0], "xor": [0, 0, 0]}
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tot = sharding.global_totals(red, device=coll_dev)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert tot["count"] == n_global, (tot["count"], n_global)
    if rank == 0:
        print(json.dumps({"file_records": n_global, "ranks": world, "proc": a.proc, "seconds": round(dt, 4),
                          "M_records_per_s": round(n_global / dt / 1e6, 1), "file_GBps": round(24 * n_global / dt / 1e9, 2),
                          "count": tot["count"], "sums": tot["sum"], "xors": tot["xor"],
                          "rank0_shard": [first, end], "rank0_batches": st.batches}), flush=True)
    m.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
