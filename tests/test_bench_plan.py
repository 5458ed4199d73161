"""bench.py's shard plan on CPU: strong scaling = BASELINE configs[3] (ONE stream of --records records split by the
reference's static split, src/io/mmap.rs:297-307), weak scaling = --records on every rank."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.parametrize("world", [1, 2, 4, 8, 7])
@pytest.mark.parametrize("records", [1_000_000_000, 1_000_000_007, 5, 0])
def test_strong_plan_is_the_reference_split(world, records, oracle):
    import bench

    prev_end, total = 0, 0
    for rank in range(world):
        n_global, first, n = bench.plan_shard(float(records), "strong", world, rank)
        assert n_global == records
        assert (first, first + n) == tuple(oracle.shard_range(records, world, rank))
        assert first == prev_end
        prev_end, total = first + n, total + n
        if rank < world - 1:
            assert n == records // world
    assert total == records  # the remainder went to the last rank


@pytest.mark.parametrize("world", [1, 2, 8])
def test_weak_plan_keeps_per_rank_size(world):
    import bench

    for rank in range(world):
        n_global, first, n = bench.plan_shard(1e6, "weak", world, rank)
        assert (n_global, first, n) == (1_000_000 * world, 1_000_000 * rank, 1_000_000)


def test_default_mode_is_strong(monkeypatch):
    import bench

    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    a = bench.parse()
    assert a.scaling == "strong" and a.records == 1e9 and (a.bc_len, a.umi_len) == (16, 12)
