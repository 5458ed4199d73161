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
