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


def test_gpus_flag_means_n(monkeypatch):
    """`--gpus N` is honoured or refused, never ignored: under a launcher it must equal WORLD_SIZE;
    run bare with N > 1 the ranks are started as a child torch.distributed.run on 127.0.0.1."""
    import bench

    what, world = bench.resolve_world(bench.parse(["--gpus", "4"]), ["--gpus", "4"], {"WORLD_SIZE": "4"})
    assert (what, world) == ("run", 4)
    assert bench.resolve_world(bench.parse([]), [], {"WORLD_SIZE": "8"}) == ("run", 8)      # flag absent: the launcher decides
    assert bench.resolve_world(bench.parse([]), [], {}) == ("run", 1)
    assert bench.resolve_world(bench.parse(["--gpus", "1"]), ["--gpus", "1"], {}) == ("run", 1)
    with pytest.raises(SystemExit) as ei:
        bench.resolve_world(bench.parse(["--gpus", "8"]), ["--gpus", "8"], {"WORLD_SIZE": "1"})
    assert "contradicts WORLD_SIZE=1" in str(ei.value)
    argv = ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    what, cmd = bench.resolve_world(bench.parse(argv), argv, {})
    assert what == "spawn" and cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-len(argv) - 1].endswith("bench.py") and cmd[-len(argv):] == argv


def test_bare_run_with_contradicting_world_size_fails_before_any_gpu_call():
    """The whole program, as the driver would start it wrongly: non-zero exit and a message, no torch import needed."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "contradicts WORLD_SIZE=1" in r.stderr
