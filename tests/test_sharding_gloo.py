"""The N>1 path on CPU: world_size-2/3 `gloo` process groups run the rank sharding + the one
cross-rank exchange (ibu_amd.sharding) that bench.py uses over RCCL.  Each rank plays a GPU: it
takes its contiguous record range (mmap.rs:297-307), reduces it with the CPU oracle standing in
for the device kernel, and the combined totals must equal the oracle's reduce of the whole stream
— including u64 wrap-around in the sums."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_global, seed, lens, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from ibu_amd import sharding
    from oracle import oracle as orc

    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        a, b = sharding.rank_shard(n_global, world, rank)
        # every rank materialises only its own range of the global counter-based stream
        recs = orc.generate(seed, a, b - a, *lens)
        local = orc.reduce_records(recs)
        tot = sharding.global_totals(local)
        np.save(os.path.join(out_dir, f"r{rank}.npy"),
                np.array([a, b, tot["count"]] + tot["sum"] + tot["xor"], dtype=np.uint64))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_global,lens", [(2, 100_003, (16, 12)), (2, 1, (16, 12)), (3, 50_000, (32, 32))])
def test_rank_shards_and_global_totals(tmp_path, oracle, world, n_global, lens):
    import torch.multiprocessing as mp

    seed = 0x1B00003
    mp.spawn(_worker, args=(world, _free_port(), n_global, seed, lens, str(tmp_path)), nprocs=world, join=True)
    whole = oracle.reduce_records(oracle.generate(seed, 0, n_global, *lens))
    want = [whole["count"]] + whole["sum"] + whole["xor"]
    prev_end = 0
    for r in range(world):
        got = [int(v) for v in np.load(tmp_path / f"r{r}.npy")]
        a, b = got[0], got[1]
        assert (a, b) == tuple(oracle.shard_range(n_global, world, r))
        assert a == prev_end  # contiguous, in rank order
        prev_end = b
        assert got[2:] == want, f"rank {r}"
    assert prev_end == n_global
    if lens == (32, 32):  # full-range u64 fields: the sums really wrapped
        assert sum(int(v) for v in oracle.generate(seed, 0, n_global, *lens)["barcode"]) >= 2**64


def test_global_totals_without_process_group_is_identity():
    from ibu_amd import sharding

    loc = {"count": 5, "sum": [1, 2**64 - 1, 3], "xor": [7, 8, 9]}
    assert sharding.global_totals(loc) == loc
    assert sharding.expected_index_sum(10) == 45
    assert sharding.expected_index_sum(2**33) == (2**33 * (2**33 - 1) // 2) % 2**64


# ---- distributed sample sort: control flow on CPU (numpy stand-in for the device ops) ------------------------------
class _NumpySortOps:
    """Test stand-in for DeviceSortOps: same interface, records in a CPU uint8 tensor, sorted / searched by the oracle."""

    def __init__(self, orc):
        self.orc = orc

    def empty(self, nbytes, like):
        import torch
        return torch.empty(max(int(nbytes), 24), dtype=torch.uint8)

    def _recs(self, buf, n):
        return np.frombuffer(buf[: n * 24].numpy().tobytes(), dtype=self.orc.REC_DTYPE)

    def local_sort(self, buf, n):
        import torch
        if n > 1:
            buf[: n * 24] = torch.from_numpy(np.frombuffer(self.orc.sort_records(self._recs(buf, n)).tobytes(), dtype=np.uint8).copy())

    def rows(self, buf, n, idx):
        return buf[: n * 24].view(n, 24)[idx]

    def lower_bounds(self, buf, n, keys):
        import torch
        recs = self._recs(buf, n)
        ks = np.frombuffer(keys.contiguous().numpy().tobytes(), dtype=self.orc.REC_DTYPE)
        return torch.tensor([self.orc.lower_bound(recs, k) for k in ks], dtype=torch.int64)

    def fetch(self, buf, i):
        return bytes(buf[i * 24:(i + 1) * 24].numpy())

    # compacted keys (the exchange format): the numpy statement of tests/keyplan_np.py
    def census_words(self, buf, n):
        from tests import keyplan_np
        return keyplan_np.census_words(self._recs(buf, n))

    def key_plan(self, or_words, and_words):
        from tests import keyplan_np
        self.plans = getattr(self, "plans", []) + [keyplan_np.Plan(or_words, and_words)]
        return self.plans[-1]

    def compact(self, plan, buf, n):
        import torch
        from tests import keyplan_np
        e = keyplan_np.compact(plan, self._recs(buf, n)).reshape(-1)
        return torch.from_numpy(e.copy()) if n else torch.empty(24, dtype=torch.uint8)

    def expand(self, plan, elems, n):
        import torch
        from tests import keyplan_np
        if n == 0:
            return torch.empty(24, dtype=torch.uint8)
        r = keyplan_np.expand(plan, elems[: n * 12].numpy().reshape(n, 12))
        return torch.from_numpy(np.frombuffer(r.tobytes(), dtype=np.uint8).copy())


def _sort_worker(rank, world, port, counts, seed, lens, skew, out_dir, force=False, compact=True):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from ibu_amd import sharding
    from oracle import oracle as orc

    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        first = sum(counts[:rank])
        recs = orc.generate(seed, first, counts[rank], *lens)
        if skew:  # heavy duplicates: few distinct barcodes, so splitters collide
            recs["barcode"] %= 3
            recs["umi"] %= 2
        buf = torch.from_numpy(np.frombuffer(recs.tobytes(), dtype=np.uint8).copy()) if counts[rank] else torch.empty(24, dtype=torch.uint8)
        stats = {}
        out, n_out = sharding.distributed_sort(_NumpySortOps(orc), buf, counts[rank], samples_per_rank=16, force=force, stats=stats,
                                               compact=compact)
        if rank == 0:
            with open(os.path.join(out_dir, "wire.txt"), "w") as f:
                f.write(str(stats["bytes_per_record_on_the_wire"]))
        tot = sharding.global_totals(orc.reduce_records(recs), force=force)
        assert tot["count"] >= counts[rank]
        np.save(os.path.join(out_dir, f"s{rank}.npy"), out[: n_out * 24].numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("counts,lens,skew", [([5000, 7003], (16, 12), False), ([4000, 0, 6001], (16, 12), False),
                                              ([3000, 3000], (4, 4), True), ([1, 2, 3], (32, 32), False), ([0, 0], (16, 12), False)])
@pytest.mark.parametrize("compact", [True, False])
def test_distributed_sort_control_flow(tmp_path, oracle, counts, lens, skew, compact):
    import torch.multiprocessing as mp

    world, seed = len(counts), 0x1B00006
    mp.spawn(_sort_worker, args=(world, _free_port(), counts, seed, lens, skew, str(tmp_path), False, compact), nprocs=world, join=True)
    # 12-byte elements on the wire whenever the keys of ALL ranks vary in at most 12 bytes ((32,32): 8 + 8 + index bytes > 12)
    # (no record anywhere: the OR / AND identities make every byte look varying, and there is nothing to ship)
    assert int(open(tmp_path / "wire.txt").read()) == (12 if compact and lens != (32, 32) and sum(counts) else 24)
    allrecs = oracle.generate(seed, 0, sum(counts), *lens)
    if skew:
        allrecs["barcode"] %= 3
        allrecs["umi"] %= 2
    want = oracle.sort_records(allrecs).tobytes()
    parts = [np.load(tmp_path / f"s{r}.npy").tobytes() for r in range(world)]
    assert b"".join(parts) == want  # rank order IS the global order; nothing lost, nothing duplicated
    if not skew and sum(counts) > 1000:  # samples balance the ranges roughly (no rank ends up with everything)
        assert max(len(p) for p in parts) < 0.8 * len(want)


def test_group_of_one_rank_runs_every_collective_when_forced(tmp_path, oracle):
    """`force=True` (bench.py / tools/sharded_sort.py --force-dist): a single rank still goes through the sample
    gather, the count exchange and the record all-to-all (zero splitters) — what a one-GPU box runs to put the RCCL
    calls of the N > 1 path on real hardware before an 8-GPU node ever does."""
    import torch.multiprocessing as mp

    counts, lens, seed = [6007], (16, 12), 0x1B00006
    mp.spawn(_sort_worker, args=(1, _free_port(), counts, seed, lens, False, str(tmp_path), True), nprocs=1, join=True)
    want = oracle.sort_records(oracle.generate(seed, 0, counts[0], *lens)).tobytes()
    assert np.load(tmp_path / "s0.npy").tobytes() == want


# ---- the bulk exchange as grouped point-to-point messages (what the RCCL path uses instead of one all_to_all_single) --
def _p2p_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from ibu_amd import sharding

    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        # rank r sends (r + 1) * 700 * (j + 1) bytes to rank j (0 bytes from rank 1 to rank 2), byte k of that message = (r, j, k)
        def size(r, j):
            return 0 if (r, j) == (1, 2) else (r + 1) * 700 * (j + 1)

        def msg(r, j):
            k = np.arange(size(r, j))
            return ((k * 7 + r * 31 + j * 101) % 251).astype(np.uint8)

        in_splits = [size(rank, j) for j in range(world)]
        out_splits = [size(j, rank) for j in range(world)]
        payload = torch.from_numpy(np.concatenate([msg(rank, j) for j in range(world)]))
        landed = torch.zeros(sum(out_splits), dtype=torch.uint8)
        sharding._exchange_p2p(landed, payload, in_splits, out_splits, rank, world, None, chunk=1000)   # several chunks per pair
        want = np.concatenate([msg(j, rank) for j in range(world)])
        assert landed.numpy().tobytes() == want.tobytes()
        ref = torch.zeros_like(landed)
        dist.all_to_all_single(ref, payload, out_splits, in_splits)
        assert torch.equal(ref, landed)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("1")
    finally:
        dist.destroy_process_group()


def test_exchange_as_point_to_point_chunks_equals_all_to_all(tmp_path):
    import torch.multiprocessing as mp

    world = 3
    mp.spawn(_p2p_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
