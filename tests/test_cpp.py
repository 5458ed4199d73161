"""The C++ mirror of the reference's public API (include/ibu.hpp, above the C ABI) and the reference's two
example programs restated in C++ (examples/*.cpp).  tests/cpp/test_host.cpp restates the reference's own unit
tests one by one; here they are built (g++) and run.  Device legs run under -m gpu."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "bin")


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    return BIN


def _run(args, **kw):
    return subprocess.run(args, capture_output=True, text=True, timeout=600, **kw)


def _compressed_fixture(d, oracle):
    """plain + gzip + multi-member gzip + BGZF + bzip2 + xz + zstd + a truncated BGZF of the same 50 000 records."""
    import bz2
    import ctypes as C
    import gzip
    import lzma

    import ibu_amd as ia
    from tests.bgzf import bgzf_compress

    recs = oracle.generate(11, 0, 50_000, 16, 12)
    w = ia.Writer.from_path(d / "plain.ibu", ia.Header(16, 12))
    w.write_batch(recs)
    w.finish()
    w.close()
    raw = (d / "plain.ibu").read_bytes()
    (d / "a.gz").write_bytes(gzip.compress(raw, 1))
    (d / "multi.gz").write_bytes(gzip.compress(raw[:100_003], 1) + gzip.compress(raw[100_003:], 1))
    # payloads separated by more than (inflate threads x chunk) bytes of EMPTY members: batches that deliver nothing must not
    # recycle the chunk buffers the caller is still reading (pgzip.cpp next_batch) — ASan / TSan see it if they do
    hole = gzip.compress(b"", 6) * 2500
    (d / "holes.gz").write_bytes(gzip.compress(raw[:300_011], 1) + hole + gzip.compress(raw[300_011:700_001], 6) + hole + gzip.compress(raw[700_001:], 9) + hole)
    (d / "a.bgz").write_bytes(bgzf_compress(raw))
    (d / "cut.bgz").write_bytes(bgzf_compress(raw)[:200_000])
    (d / "a.bz2").write_bytes(bz2.compress(raw, 1))
    (d / "a.xz").write_bytes(lzma.compress(raw, preset=0))
    z = C.CDLL("libzstd.so.1")
    z.ZSTD_compressBound.restype = z.ZSTD_compress.restype = C.c_size_t
    z.ZSTD_compressBound.argtypes = [C.c_size_t]
    z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
    buf = C.create_string_buffer(z.ZSTD_compressBound(len(raw)))
    k = z.ZSTD_compress(buf, len(buf), raw, len(raw), 1)
    (d / "a.zst").write_bytes(buf.raw[:k])


def test_cpp_mirror_passes_the_reference_unit_tests(built, tmp_path, oracle):
    fx = tmp_path / "compressed"
    fx.mkdir()
    _compressed_fixture(fx, oracle)
    r = _run([os.path.join(built, "test_host")], env={**os.environ, "TMPDIR": str(tmp_path), "IBU_TEST_COMPRESSED_DIR": str(fx)})
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failed" in r.stdout and r.stdout.count("\nok ") + r.stdout.startswith("ok ") >= 30


@pytest.mark.skipif(not os.environ.get("IBU_RUN_ASAN"), reason="set IBU_RUN_ASAN=1: builds a sanitized library (~1 min)")
def test_cpp_mirror_under_asan_ubsan(tmp_path, oracle):
    """Host AddressSanitizer + UBSan (leaks included) over the whole C++ mirror suite and every input decoder."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "asan"])
    fx = tmp_path / "compressed"
    fx.mkdir()
    _compressed_fixture(fx, oracle)
    env = {**os.environ, "TMPDIR": str(tmp_path), "IBU_TEST_COMPRESSED_DIR": str(fx), "IBU_PGZ_THREADS": "4", "IBU_PGZ_CHUNK": "4096",
           "ASAN_OPTIONS": "detect_leaks=1:halt_on_error=1", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"}
    r = _run([os.path.join(BIN, "asan", "test_host")], env=env)
    assert r.returncode == 0 and "0 failed" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_plain_c_client_of_the_abi(built):
    """gcc -std=c11 -pedantic: the header and the library are usable from C with nothing but pointers and sizes."""
    r = _run([os.path.join(built, "abi_smoke")])
    assert r.returncode == 0 and "abi_smoke ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.skipif(not os.environ.get("IBU_RUN_ASAN"), reason="set IBU_RUN_ASAN=1: builds a sanitized library (~1 min)")
def test_cpp_mirror_under_tsan(tmp_path, oracle):
    """ThreadSanitizer over process_parallel's workers, the parallel loaders, the BGZF inflate threads and the parallel
    gzip inflate (many small chunks per batch: IBU_PGZ_CHUNK)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "tsan"])
    fx = tmp_path / "compressed"
    fx.mkdir()
    _compressed_fixture(fx, oracle)
    env = {**os.environ, "TMPDIR": str(tmp_path), "IBU_TEST_COMPRESSED_DIR": str(fx), "IBU_PGZ_THREADS": "4", "IBU_PGZ_CHUNK": "4096",
           "TSAN_OPTIONS": "halt_on_error=1"}
    r = _run([os.path.join(BIN, "tsan", "test_host")], env=env)
    assert r.returncode == 0 and "0 failed" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr, r.stdout[-2000:] + r.stderr[-4000:]


def test_roundtrip_example_1e6(built, tmp_path, kat):
    """BASELINE configs[0]: examples/roundtrip.rs shape at 1e6 records — CPU plumbing only."""
    r = _run([os.path.join(built, "roundtrip"), "1000000", "--json", "--dir", str(tmp_path)])
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out["records"] == 1_000_000 and out["checksum"] == "0x0000000000000000"  # degenerate XOR
    assert out["sums"] == [499_999_500_000] * 3
    assert not os.listdir(tmp_path)  # the example removes its file


def test_parallel_example(built, tmp_path):
    """examples/parallel.rs shape: process_parallel(proc, 0) sums of the three fields."""
    n = 3_000_000
    r = _run([os.path.join(built, "parallel"), str(n), "--json", "--dir", str(tmp_path)])
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    want_idx = n * (n - 1) // 2
    want_bc = sum(range(1_000_000)) * 3
    assert out["sums"][2] == want_idx and out["sums"][0] == want_bc
    assert out["sums"][1] == sum((i * 31) % 1_000_000 for i in range(n))


def test_random_example(built, tmp_path):
    """examples/random.rs shape: value ranges and CLI; the UMI is NOT masked to umi_len."""
    import numpy as np

    sys_path = os.path.join(ROOT)
    import sys
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    import ibu_amd as ia

    p = tmp_path / "rand.ibu"
    r = _run([os.path.join(built, "random"), str(p), "--records", "0.05", "--barcodes", "37", "--max-index", "500", "--seed", "42"])
    assert r.returncode == 0, r.stderr
    assert "Finished generating 50000 records" in r.stderr
    h, recs = ia.load_to_vec(p)
    assert (h.bc_len, h.umi_len, h.sorted()) == (16, 12, False) and len(recs) == 50_000
    assert recs["barcode"].max() < 37 and recs["index"].max() < 500 and len(np.unique(recs["barcode"])) == 37
    assert recs["umi"].max() > 2**24  # full-range u64 under umi_len = 12
    again = tmp_path / "again.ibu"
    _run([os.path.join(built, "random"), str(again), "--records", "0.05", "--barcodes", "37", "--max-index", "500", "--seed", "42"])
    assert again.read_bytes() == p.read_bytes()  # a seed reproduces the file
    bad = _run([os.path.join(built, "random"), str(tmp_path / "bad.ibu"), "--bc-len", "33"])
    assert bad.returncode == 1 and "InvalidBarcodeLength" in bad.stderr  # header.validate() before anything is written


@pytest.mark.gpu
def test_cpp_device_tests(built, tmp_path, oracle):
    fx = tmp_path / "compressed"
    fx.mkdir()
    _compressed_fixture(fx, oracle)
    r = _run([os.path.join(built, "test_device")], env={**os.environ, "TMPDIR": str(tmp_path), "IBU_TEST_COMPRESSED_DIR": str(fx)})
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failed" in r.stdout


@pytest.mark.gpu
def test_examples_device_legs(built, tmp_path):
    r = _run([os.path.join(built, "roundtrip"), "2000000", "--device", "--json", "--dir", str(tmp_path)])
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout)["device"] is True
    r = _run([os.path.join(built, "parallel"), "5000000", "--device", "--json", "--dir", str(tmp_path)])
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out["device"] is True and out["sums"][2] == 5_000_000 * 4_999_999 // 2


@pytest.mark.gpu
def test_sort_file_example(built, tmp_path, oracle):
    """random -> sort_file: the output is the oracle's sort of the input under a header with the sorted flag."""
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import ibu_amd as ia

    src, dst = tmp_path / "in.ibu", tmp_path / "out.ibu"
    r = _run([os.path.join(built, "random"), str(src), "--records", "0.3", "--barcodes", "200", "--max-index", "100000", "--seed", "7"])
    assert r.returncode == 0, r.stderr
    r = _run([os.path.join(built, "sort_file"), str(src), str(dst), "--top", "3"])
    assert r.returncode == 0, r.stderr
    assert "300000 records, 200 barcodes" in r.stdout and r.stdout.count("distinct UMIs") == 3
    h_in, recs = ia.load_to_vec(src)
    h_out, got = ia.load_to_vec(dst)
    assert not h_in.sorted() and h_out.sorted() and (h_out.bc_len, h_out.umi_len) == (16, 12)
    assert got.tobytes() == oracle.sort_records(recs).tobytes()
    again = tmp_path / "again.ibu"  # a sorted file goes through unchanged (flag trusted, then verified)
    r = _run([os.path.join(built, "sort_file"), str(dst), str(again)])
    assert r.returncode == 0 and again.read_bytes() == dst.read_bytes()
    multi = tmp_path / "multi.ibu"  # the same file through three contexts and ONE call of the multi-GPU sort: the same bytes
    r = _run([os.path.join(built, "sort_file"), str(src), str(multi), "--contexts", "3"])
    assert r.returncode == 0, r.stderr
    assert "300000 records over 3 contexts" in r.stdout and multi.read_bytes() == dst.read_bytes()


@pytest.mark.gpu
def test_stream_pull_example(built, tmp_path, oracle):
    """random -> gzip -> stream_pull: the per-barcode record counts gathered batch by batch from the pull stream are the
    oracle's counts over the whole file (BarcodeAnalyzer, parallel.rs:72-98)."""
    import gzip
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import ibu_amd as ia

    src = tmp_path / "in.ibu"
    r = _run([os.path.join(built, "random"), str(src), "--records", "0.4", "--barcodes", "300", "--max-index", "100000", "--seed", "11"])
    assert r.returncode == 0, r.stderr
    gz = tmp_path / "in.ibu.gz"
    gz.write_bytes(gzip.compress(src.read_bytes(), 1))
    _, recs = ia.load_to_vec(src)
    bcs, counts, _ = oracle.barcode_counts(oracle.sort_records(recs))
    want = sorted(zip((int(c) for c in counts), (int(b) for b in bcs)), key=lambda t: (-t[0], t[1]))[:4]
    for path in (src, gz):
        r = _run([os.path.join(built, "stream_pull"), str(path), "--top", "4", "--slot-records", "98304", "--json"])
        assert r.returncode == 0, r.stderr
        out = json.loads(r.stdout)
        assert out["records"] == 400_000 and out["barcodes"] == 300 and out["batches"] == 5   # 4 x 98 304 + 6 784
        assert [(c, b) for b, c in out["top"]] == want
