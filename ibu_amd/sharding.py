"""Multi-GPU sharding of a record stream: one process per GPU, contiguous record ranges, no
data-path collective for the codec / reduce path; the distributed SORT at the bottom of this module is the one
operation with a real exchange step (one bulk exchange of the records — as 12-byte compacted keys where the keys allow — over
RCCL / xGMI).

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


def global_totals(local, device=None, group=None, force=False):
    """Combine per-rank `{"count", "sum"[3], "xor"[3]}` (Context.reduce) into the global one.

    Wrapping u64 adds are two's-complement i64 adds, so SUM over int64 is bit-exact.  Without an
    initialised process group (single GPU) the local value is returned unchanged; so it is with a group of one rank
    unless `force` (a one-GPU box rehearsing the collectives of the N > 1 runs)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
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
# exchange step, so the one place RCCL moves bulk data: an all-to-all exchange of the records over xGMI (every GPU ships
# about (W-1)/W of its shard; point-to-point links, so all 7 are busy at once).
#
#   1. every rank sorts its shard locally (ibu_sort_records, on torch's current stream)
#   2. every rank contributes s evenly spaced sample records; ONE all_gather of [s, 24]-byte tensors; every rank sorts
#      the W*s samples (device radix sort) and keeps W-1 splitters — all on the device
#   3. ONE kernel finds the lower bound of every splitter in the sorted shard (ibu_lower_bound_records)
#   4. send counts -> receive counts with ONE small all_to_all_single of W int64; the only host synchronisation of
#      the whole sort reads this rank's row and column of the count matrix (all_to_all_single wants Python lists)
#   5. the records travel in ONE grouped batch of point-to-point messages (_exchange_p2p) — as 12-BYTE ELEMENTS when at most 12 bytes of the
#      24-byte key vary over ALL ranks (16/12 records with indices below 2^32: 11): every rank's OR / AND words ride the
#      sample gather, every rank derives the same plan (ibu_key_plan_init), compacts its sorted shard (ibu_records_compact),
#      ships half the bytes over the per-link-bound xGMI mesh and expands what it received (ibu_records_expand)
#   6. each rank sorts what it received (W sorted runs) — rank r now holds the r-th key range, globally ordered
#
# No pickled objects, no per-record host probes (round 1 did ~(W-1) log2 n `.cpu()` round trips here).  Everything a
# rank launches — sort, gather, search, collectives — goes to torch's CURRENT stream, so the exchange is ordered before
# the second sort without a host wait (round 1 sorted on the context's own stream: a race under nccl).
#
# The device work goes through `ops` so the control flow (samples, splitters, bounds, counts, exchange) is also exercised
# on CPU under gloo with a numpy stand-in (tests/test_sharding_gloo.py); DeviceSortOps is the product implementation.
# ---------------------------------------------------------------------------------------------------------------
_REC = 24
_ELEM = 12   # a record's varying key bytes (ibu_records_compact)


def _rec_key(b):
    """24 little-endian bytes -> (barcode, umi, index) tuple (Record's derived Ord, record.rs:58)."""
    return (int.from_bytes(b[0:8], "little"), int.from_bytes(b[8:16], "little"), int.from_bytes(b[16:24], "little"))


class DeviceSortOps:
    """Product implementation: records live in a torch uint8 CUDA tensor; every launch goes to torch's current stream."""

    def __init__(self, ctx):
        self.ctx = ctx

    def _stream(self, like):
        import torch
        h = torch.cuda.current_stream(like.device).cuda_stream
        # the C ABI reads a NULL stream as "the context's own stream" — a non-blocking stream that torch's (and RCCL's)
        # work on the legacy default stream is NOT ordered with.  distributed_sort runs inside scope(), which never
        # leaves the default stream current.
        assert h != 0, "DeviceSortOps needs a non-default torch stream to be current (use ops.scope(buf))"
        return h

    def scope(self, like):
        """Context manager: makes a real (non-default) torch stream current for the duration, ordered behind the caller's
        current stream on entry and in front of it on exit, so that every kernel of this library, every torch op and
        every collective of the sort is issued to ONE stream.  (With the legacy default stream current the handle is 0,
        which the C ABI takes for the context's own stream: kernels and collectives would race.)"""
        import contextlib

        import torch

        @contextlib.contextmanager
        def _scope():
            cur = torch.cuda.current_stream(like.device)
            if cur.cuda_stream != 0:
                yield
                return
            side = self._side = getattr(self, "_side", None) or torch.cuda.Stream(device=like.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                yield
            cur.wait_stream(side)
        return _scope()

    def hand_over(self, t, like):
        """A tensor produced inside scope() and used by the caller afterwards: tell the caching allocator."""
        import torch
        t.record_stream(torch.cuda.current_stream(like.device))
        return t

    def empty(self, nbytes, like):
        import torch
        return torch.empty(max(int(nbytes), _REC), dtype=torch.uint8, device=like.device)

    def local_sort(self, buf, n):
        """In place, on the current stream (ibu_sort_records waits for that stream once itself: the census read-back)."""
        if n > 1:
            tmp = self.empty(n * _REC, buf)
            self.ctx.sort_records(buf, tmp, n, stream=self._stream(buf))

    def rows(self, buf, n, idx):
        """Records at the positions in the int64 tensor `idx` -> [len(idx), 24] uint8 tensor on the same device."""
        return buf[: n * _REC].view(n, _REC)[idx]

    def lower_bounds(self, buf, n, keys):
        """keys: [k, 24] uint8 (device) -> int64 tensor [k] of first positions with record >= key."""
        import torch
        k = keys.shape[0]
        pos = torch.empty(max(k, 1), dtype=torch.int64, device=buf.device)
        if k:
            self.ctx.lower_bound(buf, n, keys.contiguous(), k, pos, stream=self._stream(buf))
        return pos[:k]

    def fetch(self, buf, i):  # one record as 24 bytes (tools / tests only)
        return bytes(buf[i * _REC:(i + 1) * _REC].cpu().numpy())

    # compacted keys: the exchange format
    def census_words(self, buf, n):
        """(OR[3], AND[3]) of this rank's records as Python ints (one small read-back)."""
        c = self.ctx.census(buf, n, stream=self._stream(buf))
        return c["or"], c["and"]

    def key_plan(self, or_words, and_words):
        from . import key_plan
        return key_plan(or_words, and_words)

    def compact(self, plan, buf, n):
        elems = self.empty(n * _ELEM, buf)
        if n:
            self.ctx.compact(plan, buf, n, elems, stream=self._stream(buf))
        return elems

    def expand(self, plan, elems, n):
        out = self.empty(n * _REC, elems)
        if n:
            self.ctx.expand(plan, elems, n, out, stream=self._stream(elems))
        return out


def _collective(t, group):
    """The tensor a collective of this process group can take: itself under nccl, a CPU copy under gloo (rehearsal)."""
    import torch.distributed as dist
    return t.cpu() if dist.get_backend(group) == "gloo" and t.device.type != "cpu" else t


_P2P_CHUNK = 1 << 28   # bytes per message of the exchange


def _exchange_p2p(landed, payload, in_splits, out_splits, rank, world, group, chunk=None):
    """The bulk exchange of the sort over RCCL: grouped point-to-point messages of at most `chunk` bytes, the rank's own
    part as a device copy.  Not ONE all_to_all_single, because that call returned wrong bytes for large messages on this
    stack (ROCm 7.0 RCCL 2.26.6 under torch 2.10: a one-rank self exchange of 1.2e9 bytes or more came back different
    from its input, 6e8 bytes came back right, on any stream and for any dtype — profiles/README.md, r02_al); xGMI is
    point-to-point anyway, and a grouped batch keeps all seven links busy at once."""
    import torch.distributed as dist

    chunk = chunk or _P2P_CHUNK
    soff, roff = [0], [0]
    for c in in_splits:
        soff.append(soff[-1] + c)
    for c in out_splits:
        roff.append(roff[-1] + c)
    if in_splits[rank]:
        landed[roff[rank]: roff[rank] + out_splits[rank]].copy_(payload[soff[rank]: soff[rank] + in_splits[rank]])
    p2p = []
    for j in range(world):
        if j == rank:
            continue
        peer = dist.get_global_rank(group, j) if group is not None else j
        for o in range(0, in_splits[j], chunk):
            p2p.append(dist.P2POp(dist.isend, payload[soff[j] + o: soff[j] + min(o + chunk, in_splits[j])], peer, group))
        for o in range(0, out_splits[j], chunk):
            p2p.append(dist.P2POp(dist.irecv, landed[roff[j] + o: roff[j] + min(o + chunk, out_splits[j])], peer, group))
    if p2p:
        for req in dist.batch_isend_irecv(p2p):
            req.wait()


def _distributed_sort(ops, buf, n, samples_per_rank=None, group=None, stats=None, force=False, compact=True):
    """Sort the records of all ranks globally.  `buf`: this rank's n records (24 n bytes, uint8 tensor).  Returns
    (out_buf, n_out): rank r holds the r-th contiguous range of the global order; sum of n_out == sum of n.
    `stats` (dict, optional) receives the exchange's byte counts.  `force`: run every collective even in a group of
    one rank (a one-GPU box rehearsing the RCCL calls of the N > 1 runs).  `compact`: ship 12-byte elements instead of
    24-byte records when the keys of all ranks allow it (False: always records)."""
    import torch
    import torch.distributed as dist

    ops.local_sort(buf, n)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return buf, n
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = buf.device
    s = int(samples_per_rank or 64 * world)
    # 2. samples: s evenly spaced records (fewer than s records: all of them), padded with all-ones records, which sort
    #    last and are never picked because the splitter positions count valid samples only
    valid = min(n, s)
    samp = torch.full((s, _REC), 0xFF, dtype=torch.uint8, device=dev)
    if valid:
        idx = (torch.arange(valid, device=dev, dtype=torch.int64) * n) // valid
        samp[:valid] = ops.rows(buf, n, idx)
    use_compact = bool(compact) and hasattr(ops, "census_words")
    words = [0, 0, 0, -1, -1, -1]                                  # OR / AND identities (as i64)
    if use_compact:
        o, a = ops.census_words(buf, n)
        words = [_to_i64(v) for v in list(o) + list(a)]
    meta = torch.tensor([valid] + words, dtype=torch.int64, device=dev)
    all_samp = torch.empty((world * s, _REC), dtype=torch.uint8, device=_collective(samp, group).device)
    all_meta = torch.empty(world * 7, dtype=torch.int64, device=all_samp.device)
    dist.all_gather_into_tensor(all_samp, _collective(samp, group), group=group)
    dist.all_gather_into_tensor(all_meta, _collective(meta, group), group=group)
    all_samp, all_meta = all_samp.to(dev), all_meta.to(dev).view(world, 7)
    plan = None
    if use_compact:                                                # every rank combines the same words: the same plan
        rows = all_meta[:, 1:].tolist()
        o = [0, 0, 0]
        a = [_MASK, _MASK, _MASK]
        for r in rows:
            for f in range(3):
                o[f] |= int(r[f]) & _MASK
                a[f] &= int(r[3 + f]) & _MASK
        plan = ops.key_plan(o, a)
        if plan.k > 12:
            plan = None
    flat = all_samp.reshape(-1).contiguous()
    ops.local_sort(flat, world * s)
    total_valid = all_meta[:, 0].sum()                             # stays on the device
    pick = (torch.arange(1, world, device=dev, dtype=torch.int64) * total_valid) // world
    splitters = flat.view(world * s, _REC)[pick]                   # [W-1, 24]; all-ones rows if nobody has a record
    # 3. + 4. bounds and counts
    bounds = ops.lower_bounds(buf, n, splitters)
    edges = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), bounds.to(torch.int64),
                       torch.full((1,), n, dtype=torch.int64, device=dev)])
    send_t = edges[1:] - edges[:-1]
    recv_t = torch.empty_like(_collective(send_t, group))
    dist.all_to_all_single(recv_t, _collective(send_t, group), group=group)
    send, recv = [int(v) for v in send_t.tolist()], [int(v) for v in recv_t.tolist()]   # the one host synchronisation
    n_out = sum(recv)
    width = _ELEM if plan is not None else _REC                    # bytes per record on the wire
    payload = ops.compact(plan, buf, n) if plan is not None else buf
    landed = ops.empty(n_out * width, buf)
    in_splits, out_splits = [c * width for c in send], [c * width for c in recv]
    # 5. the exchange (timed only when the caller asked for stats: the timing needs two device synchronisations)
    import time
    if stats is not None and dev.type != "cpu":
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    if dist.get_backend(group) == "gloo" and dev.type != "cpu":  # rehearsal transport: stage through the host
        src, dst = payload[: n * width].cpu(), torch.empty(n_out * width, dtype=torch.uint8)
        dist.all_to_all_single(dst, src, out_splits, in_splits, group=group)
        landed[: n_out * width].copy_(dst)
    elif dist.get_backend(group) == "gloo":  # CPU tensors in the logic tests
        dist.all_to_all_single(landed[: n_out * width], payload[: n * width], out_splits, in_splits, group=group)
    else:  # RCCL over xGMI, ordered on the current stream
        _exchange_p2p(landed, payload, in_splits, out_splits, rank, world, group)
    if stats is not None:
        if dev.type != "cpu":
            torch.cuda.synchronize(dev)
        stats["exchange_seconds"] = time.perf_counter() - t0
        stats.update(sent_bytes=(n - send[rank]) * width, received_bytes=(n_out - recv[rank]) * width, kept_bytes=send[rank] * width,
                     samples_per_rank=s, bytes_per_record_on_the_wire=width, varying_key_bytes=(plan.k if plan is not None else None))
    out = ops.expand(plan, landed, n_out) if plan is not None else landed
    # 6.
    ops.local_sort(out, n_out)
    return out, n_out


def distributed_sort(ops, buf, n, samples_per_rank=None, group=None, stats=None, force=False, compact=True):
    """Sort the records of all ranks globally (see _distributed_sort).  Device ops run inside ops.scope(): one stream
    for the library's kernels, torch's ops and the collectives, whatever stream the caller had current."""
    if not hasattr(ops, "scope"):
        return _distributed_sort(ops, buf, n, samples_per_rank, group, stats, force, compact)
    with ops.scope(buf):
        out, n_out = _distributed_sort(ops, buf, n, samples_per_rank, group, stats, force, compact)
    if out is not buf:
        ops.hand_over(out, buf)
    return out, n_out
