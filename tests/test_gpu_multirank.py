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


def test_bench_bare_with_gpus_2_starts_its_own_ranks_or_fails_loudly():
    """`python bench.py --gpus 2` run bare: bench.py starts the two ranks itself as a child
    torch.distributed.run — on this one-GPU box the ranks share cuda:0 over gloo — and the line says n_gpus: 2.  Never a
    silent one-GPU run."""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--records", "20000001", "--backend", "gloo",
                        "--share-gpu", "--no-cpu-baseline", "--placement-tries", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["global_count"] == 20_000_001 and out["config"]["records_per_rank"] == [10_000_000, 10_000_001]


def test_bench_one_gpu_line_carries_the_e2e_leg(tmp_path):
    """The driver's N = 1 line: headline + an `e2e` object with the three file -> result rates, each checked against K4 of the
    resident copy inside bench.py (small sizes here: the rates are not asserted)."""
    r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--records", "3000001", "--no-cpu-baseline", "--no-wide-leg",
                        "--no-sort-leg", "--placement-tries", "1", "--e2e-records", "700003", "--e2e-dir", str(tmp_path)],
                       capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    e = out["e2e"]
    assert "error" not in e, e
    for leg in ("load_to_device", "mmap_process_devices_decode", "gzip_reader_process_device_decode"):
        assert e[leg]["records_per_s"] > 0 and e[leg]["GBps_of_file"] > 0 and e[leg]["totals_equal_resident_copy"] is True, (leg, e[leg])
    assert e["mmap_process_devices_decode"]["devices"] >= 1 and 0 < e["gzip_reader_process_device_decode"]["gz_ratio"] < 1
    assert not list(tmp_path.iterdir())                       # the leg removes its files


def test_bench_sort_leg_carries_the_multi_context_rehearsal():
    """The sort leg of the driver's N = 1 line: the three single-GPU sorts, the per-barcode aggregation and the multi-GPU form of
    the sort rehearsed with 8 contexts on this GPU, each checked inside bench.py."""
    r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--records", "3000001", "--no-cpu-baseline", "--no-wide-leg",
                        "--no-e2e-leg", "--placement-tries", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    sl = out["sort_leg"]
    assert "error" not in sl and sl["read_order"]["sorted_and_multiset_preserved"] and sl["barcode_counts"]["counts_add_up"], sl
    cr = sl["contexts_rehearsal"]
    assert "error" not in cr and cr["contexts_on_this_gpu"] == 8 and cr["globally_sorted_and_multiset_preserved"] is True, cr
    assert sum(cr["records_per_shard_after"]) == cr["records"] == 3_000_000


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
