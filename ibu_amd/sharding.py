"""Multi-GPU sharding of a record stream: one process per GPU, contiguous record ranges, no
data-path collective for the codec / reduce path; the distributed SORT at the bottom of this module is the one
operation with a real exchange step (one all-to-all of the records over RCCL / xGMI).

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


# ---------------------------------------------------------------------------------------------------------------
# Distributed sort by (barcode, umi, index) across ranks: sample sort.  The ONE place on this path with a real
# exchange step, so the one place RCCL moves bulk data: an all-to-all of the records over xGMI (every GPU ships
# about (W-1)/W of its shard; point-to-point links, so all 7 are busy at once).
#
#   1. every rank sorts its shard locally (ibu_sort_records)
#   2. every rank contributes evenly spaced samples; all ranks sort the gathered samples and keep W-1 splitters
#   3. each rank finds the splitters' lower bounds in its sorted shard (binary search: ~log2(n) 24-byte probes each)
#   4. send counts are exchanged, records travel in ONE all_to_all_single (uneven splits)
#   5. each rank sorts what it received (W sorted runs) — rank r now holds the r-th key range, globally ordered
#
# The device work goes through `ops` so the control flow (splitters, bounds, counts, exchange) is also exercised on
# CPU under gloo with a numpy stand-in (tests/test_sharding_gloo.py); DeviceSortOps is the product implementation.
# ---------------------------------------------------------------------------------------------------------------
_REC = 24


def _rec_key(b):
    """24 little-endian bytes -> (barcode, umi, index) tuple (Record's derived Ord, record.rs:58)."""
    return (int.from_bytes(b[0:8], "little"), int.from_bytes(b[8:16], "little"), int.from_bytes(b[16:24], "little"))


class DeviceSortOps:
    """Product implementation: records live in a torch uint8 CUDA tensor, sorted by the HIP radix sort."""

    def __init__(self, ctx):
        self.ctx = ctx

    def empty(self, nbytes, like):
        import torch
        return torch.empty(max(int(nbytes), _REC), dtype=torch.uint8, device=like.device)

    def local_sort(self, buf, n):
        if n > 1:
            tmp = self.empty(n * _REC, buf)
            self.ctx.sort_records(buf, tmp, n)
        self.ctx.synchronize()

    def fetch(self, buf, i):  # one record as 24 bytes
        return bytes(buf[i * _REC:(i + 1) * _REC].cpu().numpy())

    def sample(self, buf, n, idx):  # records at the given indices, concatenated
        import torch
        if not idx:
            return b""
        rows = buf[: n * _REC].view(n, _REC)[torch.tensor(idx, dtype=torch.long, device=buf.device)]
        return bytes(rows.cpu().numpy().tobytes())


def _lower_bound(ops, buf, n, key):
    lo, hi = 0, n
    while lo < hi:
        mid = (lo + hi) // 2
        if _rec_key(ops.fetch(buf, mid)) < key:
            lo = mid + 1
        else:
            hi = mid
    return lo


def distributed_sort(ops, buf, n, samples_per_rank=None, group=None):
    """Sort the records of all ranks globally.  `buf`: this rank's n records (24 n bytes, uint8 tensor).  Returns
    (out_buf, n_out): rank r holds the r-th contiguous range of the global order; sum of n_out == sum of n."""
    import torch
    import torch.distributed as dist

    ops.local_sort(buf, n)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return buf, n
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    s = samples_per_rank or 64 * world
    idx = [(k * n) // s for k in range(s)] if n >= s else list(range(n))
    mine = ops.sample(buf, n, idx)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine, group=group)
    keys = sorted(_rec_key(blob[o:o + _REC]) for blob in gathered for o in range(0, len(blob), _REC))
    splitters = [keys[(k * len(keys)) // world] for k in range(1, world)] if keys else []
    bounds = [0] + [_lower_bound(ops, buf, n, sp) for sp in splitters] + [n]
    if not splitters:
        bounds = [0] * world + [n]  # nothing to split on: everything goes to the last rank
    send = [bounds[j + 1] - bounds[j] for j in range(world)]
    matrix = [None] * world
    dist.all_gather_object(matrix, send, group=group)
    recv = [matrix[src][rank] for src in range(world)]
    n_out = sum(recv)
    out = ops.empty(n_out * _REC, buf)
    in_splits, out_splits = [c * _REC for c in send], [c * _REC for c in recv]
    if dist.get_backend(group) == "gloo" and buf.device.type != "cpu":  # rehearsal transport: stage through the host
        src, dst = buf[: n * _REC].cpu(), torch.empty(n_out * _REC, dtype=torch.uint8)
        dist.all_to_all_single(dst, src, out_splits, in_splits, group=group)
        out[: n_out * _REC].copy_(dst)
    else:  # RCCL over xGMI (or gloo on CPU tensors in the logic tests)
        dist.all_to_all_single(out[: n_out * _REC], buf[: n * _REC], out_splits, in_splits, group=group)
    ops.local_sort(out, n_out)
    return out, n_out
