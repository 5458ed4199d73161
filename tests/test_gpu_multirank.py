"""Two ranks (one process each, gloo collectives) sharing the one GPU of the test box: the N>1 code paths of
bench.py and of the sharded-file tool end to end — rank shards, per-rank contexts, the single cross-rank exchange."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(nproc, script, *args):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), script, *args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_two_ranks_on_one_gpu_strong():
    """Default for N > 1 = BASELINE configs[3]: --records is the WHOLE stream, range-sharded over the ranks."""
    out = _torchrun(2, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--records", "100000001", "--backend", "gloo",
                    "--share-gpu", "--no-cpu-baseline")
    assert out["n_gpus"] == 2 and out["global_count"] == 100_000_001 and out["verified_roundtrip"] is True
    assert out["scaling"] == "strong" and out["config"]["records_total"] == 100_000_001
    assert out["config"]["records_per_rank"] == [50_000_000, 50_000_001]  # remainder to the last shard (mmap.rs:297-307)
    assert out["kernel_ms_ranks"]["decode"]["min"] <= out["kernel_ms_ranks"]["decode"]["max"]
    assert out["allreduce_ms"] > 0 and "config2_32_32" not in out
    # small shards: auto tries; the probing is the library's (ibu_device_alloc_probed per array), every candidate's time in the line
    assert out["placement"]["tries"] >= 4 and set(out["placement"]["per_array"]) == {"records", "bc", "umi", "idx", "output"}
    assert all(r["tries"] >= 2 and len(r["ms"]) == r["tries"] and 0 <= r["chosen"] < r["tries"] for r in out["placement"]["per_array"].values())
    assert out["value_first_placement"] > 0 and out["placement"]["first_placement_decode_frac"] > 0
    assert abs(out["value"] - 100_000_001 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]


def test_bench_two_ranks_on_one_gpu_weak():
    out = _torchrun(2, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--records", "5e7", "--scaling", "weak",
                    "--backend", "gloo", "--share-gpu", "--no-cpu-baseline")
    assert out["n_gpus"] == 2 and out["global_count"] == 100_000_000 and out["verified_roundtrip"] is True
    assert out["scaling"] == "weak" and out["config"]["records_total"] == 100_000_000
    assert out["config"]["records_per_rank"] == [50_000_000, 50_000_000]


@pytest.mark.parametrize("proc", ["reduce", "decode"])
def test_sharded_file_two_ranks(tmp_path, proc, oracle):
    n = 3_000_001
    path = str(tmp_path / "shared.ibu")
    out = _torchrun(2, "tools/sharded_file.py", path, "--make", str(n), "--proc", proc, "--backend", "gloo", "--share-gpu")
    assert out["file_records"] == n and out["count"] == n and out["ranks"] == 2
    assert out["rank0_shard"] == [0, n // 2]
    want = oracle.reduce_records(oracle.generate(0x1B00003, 0, n, 16, 12))
    assert out["sums"][2] == want["sum"][2]
    if proc == "reduce":
        assert out["sums"] == want["sum"] and out["xors"] == want["xor"]


@pytest.mark.parametrize("compact", [True, False])
def test_distributed_sort_two_ranks(compact):
    """Sample sort across two ranks (device radix sorts, one all-to-all — staged through the host under gloo here, RCCL in
    production): every rank sorted, rank ranges ordered, multiset preserved.  16/12 records vary in 10 bytes over both
    ranks, so the exchange ships 12-byte elements (compact) — half the bytes of the 24-byte records (--no-compact)."""
    extra = [] if compact else ["--no-compact"]
    out = _torchrun(2, "tools/sharded_sort.py", "--records", "3000001", "--backend", "gloo", "--share-gpu", *extra)
    assert out["every_rank_sorted"] and out["rank_ranges_ordered"] and out["multiset_preserved"]
    assert out["count"] == 6_000_002 and sum(out["records_per_rank_out"]) == 6_000_002
    assert max(out["records_per_rank_out"]) < 0.6 * 6_000_002  # the splitter balanced the two ranges
    width = 12 if compact else 24
    assert out["bytes_per_record_on_the_wire"] == width and (out["varying_key_bytes"] == 10) == compact   # 4 barcode + 3 UMI + 3 index bytes (6e6 records)
    # about half of every shard travels; the transport says what carried it (gloo + host staging here, RCCL in production)
    assert all(0.3 * 3_000_001 * width < b < 0.7 * 3_000_001 * width for b in out["exchange_sent_bytes_per_rank"])
    assert out["backend"] == "gloo" and out["exchange_seconds_max"] > 0
