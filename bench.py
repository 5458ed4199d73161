#!/usr/bin/env python3
"""bench.py — records/s of the IBU hot path (fused 2-bit decode + encode of 24-byte records)
on N MI355X GPUs, with the roofline of the dominant kernel and a CPU baseline beside it.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1
the driver launches one rank per GPU with torch.distributed.run.  A *step* is one pass of the
hot path over the rank's shard: K2 decode (AoS records -> barcode ASCII + UMI ASCII + index
column) followed by K3 encode (those columns -> AoS records).  Inputs are resident in HBM
before the timed region (synthetic, generated on device by the counter-based generator of
SURVEY §8d, so every rank materialises its own contiguous record range of the global stream —
the static split of src/io/mmap.rs:297-307).  No collective on the data path; ONE all-reduce
at the end carries the global record count and field sums (BASELINE north_star).

Output: one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--records", type=float, default=1e9, help="records per GPU (weak scaling)")
    p.add_argument("--bc-len", type=int, default=16)
    p.add_argument("--umi-len", type=int, default=12)
    p.add_argument("--seed", type=lambda s: int(s, 0), default=0x1B00003)
    p.add_argument("--cpu-sample", type=float, default=0, help="records for the CPU baseline (0 = auto)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-verify", action="store_true")
    # rehearsal of the N>1 code path on a box with fewer GPUs than ranks (never used by the driver):
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: collectives on CPU tensors")
    p.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (with --backend gloo)")
    return p.parse_args()


def usable_cores():
    """CPUs this job may really use: the affinity mask limited by the cgroup CPU quota (a 1-GPU box of this pool
    shows 256 CPUs in the mask but is throttled to its share; 256 runnable threads under that quota thrash)."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(args):
    """The oracle (a C restatement of the reference's CPU path: static range split over OS threads,
    1Mi-record batches, scalar 2-bit codec) timed on this box's host cores on a bounded sample."""
    from oracle import oracle as orc  # the checker, used here only as the reported baseline

    threads = usable_cores()
    n = int(args.cpu_sample) or 100_000_000
    t1, chk = orc.bench_decode_encode(min(n, 4_000_000), args.bc_len, args.umi_len, args.seed, threads)  # warm-up + rate probe
    rate = min(n, 4_000_000) / max(t1, 1e-6)
    # about 1.5 s of wall time on every usable core (16 cores -> ~25 CPU-seconds): repeat the pass over the sample
    reps = max(1, int(rate * 1.5 / n))
    t, chk = orc.bench_decode_encode(n, args.bc_len, args.umi_len, args.seed, threads, reps)
    assert chk != 2**64 - 1, "oracle round trip failed"
    n_total = n * reps
    return {
        "value": n_total / t, "unit": "records/s", "cores": threads, "kind": "port",
        "sample": f"{reps} pass(es) over {n} records bc_len={args.bc_len} umi_len={args.umi_len}, decode+encode, static split over "
                  f"{threads} OS threads (C restatement of the reference's std::thread path; the reference is Rust "
                  f"and cannot be built here)",
        "seconds": t, "cpu_seconds": t * threads,
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import ibu_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":  # "nccl" IS RCCL on ROCm
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctx = ibu_amd.Context(local_rank)
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")  # where the few collective words live

    n = int(args.records)
    bc_len, umi_len = args.bc_len, args.umi_len
    n_global = n * world
    from ibu_amd import sharding

    first, end = sharding.rank_shard(n_global, world, rank)  # contiguous record-range split (mmap.rs:297-307)
    assert end - first == n

    def buf(nbytes):
        return torch.empty(nbytes, dtype=torch.uint8, device=dev)

    recs, back = buf(n * 24), buf(n * 24)
    bc, umi, idx = buf(n * bc_len), buf(n * umi_len), buf(n * 8)
    # a real (non-null) stream: the ABI reads a NULL stream as "the context's own stream", and
    # torch.cuda.Event only sees work on torch's current stream
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    st = tstream.cuda_stream
    assert st != 0
    ctx.generate(args.seed, first, n, bc_len, umi_len, recs, stream=st)
    torch.cuda.synchronize()

    def step(ev=None):
        if ev:
            ev[0].record()
        ctx.decode_ascii(recs, n, bc_len, umi_len, bc, umi, idx, stream=st)
        if ev:
            ev[1].record()
        ctx.encode_ascii(bc, umi, idx, n, bc_len, umi_len, back, stream=st)
        if ev:
            ev[2].record()

    for _ in range(args.warmup):
        step()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]

    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    dec_ms = sum(e[0].elapsed_time(e[1]) for e in events) / args.steps
    enc_ms = sum(e[1].elapsed_time(e[2]) for e in events) / args.steps

    # ---- outside the timed region: this box's own copy ceiling (plain dwordx4 copy kernel, recs -> scratch) ----
    # the second denominator SURVEY 8d asks for next to the 8 TB/s spec peak; boxes of this pool differ by ~20 %
    copy_ms = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.copy(bc, recs, min(bc.numel(), recs.numel()), stream=st)
        e1.record()
        e1.synchronize()
        copy_ms.append(e0.elapsed_time(e1))
    copy_GBps = 2 * min(bc.numel(), recs.numel()) / (sorted(copy_ms)[len(copy_ms) // 2] * 1e-3) / 1e9
    # the copy overwrote the decoded barcode column: decode once more so the verification below sees real data
    ctx.decode_ascii(recs, n, bc_len, umi_len, bc, umi, idx, stream=st)

    # ---- outside the timed region: correctness of what was timed ------------------------------
    ctx.codec_status(stream=st)  # raises if any record failed to encode
    red = ctx.reduce(recs, n, stream=st)
    verified = None
    if not args.no_verify:
        red_back = ctx.reduce(back, n, stream=st)
        chunk = 1 << 30  # compare in 1 GiB pieces: torch.equal materialises a mask as large as its inputs
        same = all(bool(torch.equal(recs[o:o + chunk], back[o:o + chunk])) for o in range(0, recs.numel(), chunk))
        verified = same and red == red_back and red["count"] == n
        if not verified:
            raise SystemExit("round trip encode(decode(x)) != x")
    # the one cross-GPU exchange: global count + wrapping field sums (4 x i64 over RCCL)
    g = sharding.global_totals(red, device=coll_dev)
    tot = [g["count"]] + g["sum"]
    assert tot[0] == n_global
    assert tot[3] == sharding.expected_index_sum(n_global)  # index column is 0..n_global-1

    if rank == 0:
        dec_bytes = n * (24 + bc_len + umi_len + 8)
        enc_bytes = dec_bytes
        achieved = dec_bytes / (dec_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp):  # PMC-derived HBM bytes per decode launch, recorded by profiles/collect_pmc.py
            try:
                with open(tp) as f:
                    rec = json.load(f).get(f"decode_{bc_len}_{umi_len}")
                if rec:
                    traffic = rec["hbm_bytes_per_record"] * n
            except (OSError, ValueError, KeyError):
                traffic = None
        out = {
            "metric": "records/s, fused 2-bit decode+encode of 24-byte IBU records (HBM-resident)",
            "value": n_global * args.steps / elapsed,
            "unit": "records/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": f"{n:.3g} records/GPU bc_len={bc_len} umi_len={umi_len}, K2 decode + K3 encode per step "
                            f"(BASELINE configs[3] shape; contiguous record-range shard per rank, no data-path collective)",
                "records_per_gpu": n, "records_total": n_global, "bc_len": bc_len, "umi_len": umi_len,
                "parallelism": f"range-shard x{world}",
            },
            "kernel_ms": {"decode": dec_ms, "encode": enc_ms},
            "kernel_GBps": {"decode": achieved, "encode": enc_bytes / (enc_ms * 1e-3) / 1e9},
            "roofline": {
                "bound": "hbm", "kernel": "ibu_k_decode", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "bytes_per_record": 24 + bc_len + umi_len + 8,
                "copy_ceiling": copy_GBps, "frac_of_copy_ceiling": achieved / copy_GBps,
            },
            "gpu": {"name": torch.cuda.get_device_name(dev), "uuid": str(getattr(torch.cuda.get_device_properties(dev), "uuid", ""))},
            "verified_roundtrip": verified,
            "global_count": tot[0],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
