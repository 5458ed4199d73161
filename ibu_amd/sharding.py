"""Multi-GPU sharding of a record stream: one process per GPU, contiguous record ranges, no
data-path collective.

The partition is the reference's static split (src/io/mmap.rs:297-307: `per = len / n`, the
remainder goes to the LAST shard), applied to ranks instead of OS threads.  Outputs concatenate in
rank order (the `Writer::ingest` pattern, writer.rs:477-482).  The only value that crosses GPUs is
what the reference's processors accumulate in `on_batch_complete` (examples/parallel.rs:28-35):
the record count and the three wrapping field sums — ONE all-reduce of 4 x i64 over RCCL (backend
"nccl" on ROCm; "gloo" in the CPU tests).  XOR has no RCCL reduction op, so the three XOR words
ride an all-gather of the same size and are combined on the host.
"""
from . import shard_range

_MASK = (1 << 64) - 1


def rank_shard(n_global, world, rank):
    """[start, end) of `rank`'s records in a global stream of `n_global` (mmap.rs:297-307)."""
    return shard_range(n_global, world, rank)


def _to_i64(v):
    v &= _MASK
    return v - (1 << 64) if v >= (1 << 63) else v


def global_totals(local, device=None, group=None):
    """Combine per-rank `{"count", "sum"[3], "xor"[3]}` (Context.reduce) into the global one.

    Wrapping u64 adds are two's-complement i64 adds, so SUM over int64 is bit-exact.  Without an
    initialised process group (single GPU) the local value is returned unchanged."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return {"count": local["count"], "sum": list(local["sum"]), "xor": list(local["xor"])}
    world = dist.get_world_size(group)
    t = torch.tensor([_to_i64(local["count"])] + [_to_i64(v) for v in local["sum"]], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    x = torch.tensor([_to_i64(v) for v in local["xor"]], dtype=torch.int64, device=device)
    xs = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(xs, x, group=group)
    tot = [int(v) & _MASK for v in t.tolist()]
    xor = [0, 0, 0]
    for part in xs:
        for k, v in enumerate(part.tolist()):
            xor[k] ^= int(v) & _MASK
    return {"count": tot[0], "sum": tot[1:], "xor": xor}


def expected_index_sum(n_global):
    """Sum of the index column 0..n_global-1 modulo 2^64 (closed form used by bench.py's self-check)."""
    return (n_global * (n_global - 1) // 2) & _MASK
