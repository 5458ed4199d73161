"""GPU parity tests for the device-backed stream entry points (csrc/stream.cpp), through the C ABI:
load_to_device (reader.rs:510-535), Writer::write_batch from device memory (writer.rs:315-351),
one shard of MmapReader::process_parallel per GPU (mmap.rs:286-332) and the streaming Reader
incl. the gzip path (reader.rs:279-306, :345-352).  Everything is compared byte for byte with the
CPU oracle on the same seeded file; small ring slots force many ring wrap-arounds."""
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x1B00004
SMALL_RING = {"slots": 3, "slot_records": 4096, "feeder_threads": 2}  # wraps the ring many times
ODD_RING = {"slots": 2, "slot_records": 1000, "feeder_threads": 1}    # slot size is rounded up to 1024


@pytest.fixture(scope="module")
def ia():
    import ibu_amd
    return ibu_amd


@pytest.fixture(scope="module")
def ctx(ia):
    c = ia.Context(0)
    yield c
    c.close()


def _write_file(oracle, path, n, bc_len=16, umi_len=12, sorted_flag=False, first=0):
    recs = oracle.generate(SEED, first, n, bc_len, umi_len)
    h = oracle.header_new(bc_len, umi_len)
    if sorted_flag:
        h.flags |= 1
    w = oracle.Writer(header=h, path=str(path))
    if n:
        w.write_batch(recs)
    w.finish()
    w.drop()
    assert os.path.getsize(path) == 32 + 24 * n
    return recs


@pytest.mark.parametrize("n", [0, 1, 127, 4096, 4097, 100_000, 1_000_003])
@pytest.mark.parametrize("ring", [None, SMALL_RING, ODD_RING])
def test_load_to_device_equals_load_to_vec(ia, ctx, oracle, tmp_path, n, ring):
    if ring is ODD_RING and n > 200_000:
        pytest.skip("1 Ki-record slots over 1e6 records adds nothing but time")
    p = tmp_path / "load.ibu"
    recs = _write_file(oracle, p, n, sorted_flag=True)
    h, dptr, got_n, st = ctx.load_to_device(p, ring=ring)
    try:
        oh, orecs = oracle.load_to_vec(str(p))
        assert got_n == n == len(orecs)
        assert (h.bc_len, h.umi_len, h.sorted()) == (16, 12, True)
        assert st.records == n and st.bytes_h2d == 24 * n
        if n:
            got = ia.DeviceBuffer.wrap(ctx, dptr, 24 * n).download().tobytes()
            assert got == recs.tobytes() == np.asarray(orecs).tobytes()
            assert ctx.reduce(dptr, n) == oracle.reduce_records(recs)
    finally:
        ctx.free(dptr)


def test_load_to_device_into_caller_buffer_and_errors(ia, ctx, oracle, tmp_path):
    p = tmp_path / "own.ibu"
    n = 10_000
    recs = _write_file(oracle, p, n)
    d = ctx.alloc(24 * n)
    h, dptr, got_n, _ = ctx.load_to_device(p, d_records=d, cap_records=n)
    assert dptr == d.ptr and got_n == n
    assert d.download().tobytes() == recs.tobytes()
    with pytest.raises(ia.IbuError) as e:  # too small a buffer is an argument error, not a partial load
        ctx.load_to_device(p, d_records=d, cap_records=n - 1)
    assert e.value.kind == "InvalidArg"
    # InvalidMapSize on a truncated file (reader.rs:520-525), Io on a missing one
    with open(p, "r+b") as f:
        f.truncate(32 + 24 * n - 5)
    with pytest.raises(ia.IbuError) as e:
        ctx.load_to_device(p)
    assert e.value.kind == "InvalidMapSize"
    with pytest.raises(ia.IbuError) as e:
        ctx.load_to_device(tmp_path / "missing.ibu")
    assert e.value.kind == "Io"


@pytest.mark.parametrize("n", [0, 1, 49_152, 49_153, 100_000, 300_001])
@pytest.mark.parametrize("ring", [None, SMALL_RING])
def test_write_batch_device_file_equals_oracle_file(ia, ctx, oracle, tmp_path, n, ring):
    bc_len, umi_len = 16, 12
    d = ctx.alloc(max(n, 1) * 24)
    ctx.generate(SEED, 5, n, bc_len, umi_len, d)
    p = tmp_path / "dev.ibu"
    w = ia.Writer.from_path(p, ia.Header(bc_len, umi_len))
    st = w.write_batch_device(ctx, d, n, ring=ring)
    assert w.records_written() == n and st.records == n and st.bytes_d2h == 24 * n
    w.finish()
    w.close()
    q = tmp_path / "cpu.ibu"
    _write_file(oracle, q, n, bc_len, umi_len, first=5)
    assert p.read_bytes() == q.read_bytes()


def test_write_batch_device_on_a_producer_stream(ia, ctx, oracle, tmp_path):
    """ibu_writer_write_batch_device_on: the records are produced on the CALLER's stream (here: generated and sorted on a
    second stream, 30e6 records so that the sort is still running when the writer is called) and the copies wait for it —
    the file holds the sorted records, not what the buffer held before."""
    n, bc_len, umi_len = 30_000_000, 16, 12
    other = ia.Context(0)                                 # its stream serves as "the caller's stream": not this context's
    s = other.stream
    d, t = ctx.alloc(n * 24), ctx.alloc(n * 24)
    ctx.generate(SEED, 0, n, bc_len, umi_len, d)          # on the context's stream ...
    ctx.synchronize()
    want = ctx.reduce(d, n)
    h = ia.Header(bc_len, umi_len)
    h.set_sorted()
    p = tmp_path / "sorted.ibu"
    w = ia.Writer.from_path(p, h)
    ctx.sort_records(d, t, n, stream=s)                   # ... sorted on the caller's
    w.write_batch_device(ctx, d, n, stream=s)
    w.finish()
    w.close()
    other.close()
    hdr, recs = ia.load_to_vec(p)
    assert hdr.sorted() and len(recs) == n
    keys = np.stack([recs["barcode"], recs["umi"], recs["index"]], axis=1)
    a, b = keys[:-1], keys[1:]
    lt = (a[:, 0] < b[:, 0]) | ((a[:, 0] == b[:, 0]) & ((a[:, 1] < b[:, 1]) | ((a[:, 1] == b[:, 1]) & (a[:, 2] <= b[:, 2]))))
    assert bool(lt.all())
    assert oracle.reduce_records(recs) == want


def test_write_batch_device_mixes_with_host_writes(ia, ctx, oracle):
    """Buffered rule of writer.rs:321-351 holds when device batches interleave with write_record."""
    n = 1000
    recs = oracle.generate(SEED, 0, n, 8, 8)
    d = ctx.upload(recs)
    w = ia.Writer.new_headless()
    w.write_record(ia.Record(1, 2, 3))
    w.write_batch_device(ctx, d, n, ring=ODD_RING)
    w.write_record(ia.Record(4, 5, 6))
    w.finish()
    want = ia.Record(1, 2, 3).as_bytes() + recs.tobytes() + ia.Record(4, 5, 6).as_bytes()
    assert w.records_written() == n + 2
    assert w.inner_bytes() == want


@pytest.mark.parametrize("n", [0, 1, 3, 10_000, 1_000_003])
@pytest.mark.parametrize("n_shards", [1, 2, 4, 7])
def test_mmap_process_device_reduce_shards_sum_to_whole(ia, ctx, oracle, tmp_path, n, n_shards):
    p = tmp_path / "m.ibu"
    recs = _write_file(oracle, p, n)
    m = ia.MmapReader.new(p)
    want = oracle.reduce_records(recs)
    count, sums, xors = 0, [0, 0, 0], [0, 0, 0]
    for s in range(n_shards):
        res, st = m.process_device(ctx, ia.PROC_REDUCE, shard=s, n_shards=n_shards, ring=SMALL_RING)
        a, b = oracle.shard_range(n, n_shards, s)
        assert res == oracle.reduce_records(recs[a:b]), (s, a, b)  # Q5: len < shards leaves all but the last empty
        assert st.records == b - a
        count += res["count"]
        sums = [(x + y) % 2**64 for x, y in zip(sums, res["sum"])]
        xors = [x ^ y for x, y in zip(xors, res["xor"])]
    assert {"count": count, "sum": sums, "xor": xors} == want
    m.close()


@pytest.mark.parametrize("lens", [(16, 12), (32, 32), (15, 11)])
def test_mmap_process_device_decode_shards_concatenate(ia, ctx, oracle, tmp_path, lens):
    bc_len, umi_len = lens
    n, n_shards = 200_003, 3
    p = tmp_path / "d.ibu"
    recs = _write_file(oracle, p, n, bc_len, umi_len)
    bc, umi, idx = oracle.decode_records(recs, bc_len, umi_len)
    m = ia.MmapReader.new(p)
    got_bc, got_umi, got_idx = [], [], []
    for s in range(n_shards):
        a, b = ia.shard_range(n, n_shards, s)
        k = b - a
        d_bc, d_umi, d_idx = ctx.alloc(k * bc_len), ctx.alloc(k * umi_len), ctx.alloc(k * 8)
        _, st = m.process_device(ctx, ia.PROC_DECODE, shard=s, n_shards=n_shards, sink=(d_bc, d_umi, d_idx),
                                 ring=SMALL_RING)
        assert st.records == k and st.batches == (k + 4095) // 4096
        got_bc.append(d_bc.download().tobytes())
        got_umi.append(d_umi.download().tobytes())
        got_idx.append(d_idx.download().tobytes())
    # outputs concatenate in shard order (the Writer::ingest pattern, writer.rs:477-482)
    assert b"".join(got_bc) == bc.tobytes()
    assert b"".join(got_umi) == umi.tobytes()
    assert b"".join(got_idx) == idx.tobytes()
    m.close()


@pytest.mark.parametrize("compressed", [False, True, "bgzf", "gzip-many-chunks"])
@pytest.mark.parametrize("n", [0, 1, 49_152, 100_000, 500_003])
def test_reader_process_device_plain_and_gzip(ia, ctx, oracle, tmp_path, compressed, n, monkeypatch):
    if compressed == "gzip-many-chunks":   # the parallel inflate with dozens of chunks per batch feeding the pinned slots
        monkeypatch.setenv("IBU_PGZ_THREADS", "6")
        monkeypatch.setenv("IBU_PGZ_CHUNK", "16384")
    p = tmp_path / "r.ibu"
    recs = _write_file(oracle, p, n)
    path = p
    if compressed == "bgzf":  # bgzip'd input: block-parallel inflate on the host, same bytes
        from tests.bgzf import bgzf_compress
        path = tmp_path / "r.ibu.bgz"
        path.write_bytes(bgzf_compress(p.read_bytes()))
    elif compressed:
        path = tmp_path / "r.ibu.gz"
        raw = p.read_bytes()
        half = len(raw) // 2
        with open(path, "wb") as f:  # two gzip members: multi-member streams must be followed to the end
            f.write(gzip.compress(raw[:half], 1))
            f.write(gzip.compress(raw[half:], 1))
    r = ia.Reader.from_path(path)
    res, st = r.process_device(ctx, ia.PROC_REDUCE, ring=SMALL_RING)
    assert res == oracle.reduce_records(recs)
    assert st.records == n
    r.close()
    # decode sink through the same path
    r = ia.Reader.from_path(path)
    d_bc, d_umi, d_idx = ctx.alloc(max(n, 1) * 16), ctx.alloc(max(n, 1) * 12), ctx.alloc(max(n, 2) * 8)
    r.process_device(ctx, ia.PROC_DECODE, sink=(d_bc, d_umi, d_idx), ring=SMALL_RING)
    bc, umi, idx = oracle.decode_records(recs, 16, 12)
    assert d_bc.download(count=n * 16).tobytes() == bc.tobytes()
    assert d_umi.download(count=n * 12).tobytes() == umi.tobytes()
    assert d_idx.download(count=n * 8).tobytes() == idx.tobytes()
    r.close()


@pytest.mark.parametrize("form", ["plain", "gzip"])
def test_reader_process_device_from_a_pipe(ia, ctx, oracle, tmp_path, form):
    """Reader::from_stdin's case (reader.rs:389-396): a descriptor that cannot seek and reads short — the pinned slots are filled
    from whatever read(2) hands over, records cut across reads included."""
    import os
    import threading
    n = 300_007
    p = tmp_path / "pipe.ibu"
    recs = _write_file(oracle, p, n)
    data = p.read_bytes() if form == "plain" else gzip.compress(p.read_bytes(), 1)
    rfd, wfd = os.pipe()

    def feed():
        rng = np.random.default_rng(5)
        pos = 0
        with os.fdopen(wfd, "wb", buffering=0) as w:
            while pos < len(data):
                k = int(rng.integers(1, 60_001))
                w.write(data[pos:pos + k])
                pos += k

    t = threading.Thread(target=feed)
    t.start()
    try:
        r = ia.Reader(rfd)
        res, st = r.process_device(ctx, ia.PROC_REDUCE, ring=SMALL_RING)
        r.close()
    finally:
        t.join()
        os.close(rfd)
    assert res == oracle.reduce_records(recs) and st.records == n


def test_reader_process_device_after_partial_host_iteration(ia, ctx, oracle, tmp_path):
    """The device stream takes over where the host iterator stopped (records already yielded are not re-read)."""
    n = 60_000
    p = tmp_path / "part.ibu"
    recs = _write_file(oracle, p, n)
    r = ia.Reader.from_path(p)
    head = [next(r) for _ in range(7)]
    assert [tuple(x) for x in head] == [tuple(int(v) for v in recs[i]) for i in range(7)]
    res, _ = r.process_device(ctx, ia.PROC_REDUCE, ring=ODD_RING)
    assert res == oracle.reduce_records(recs[7:])
    r.close()


def test_reader_process_device_truncated_stream(ia, ctx, oracle, tmp_path):
    """Q8 / reader.rs:232-237: a trailing partial record is TruncatedRecord; nothing after it is processed."""
    n = 1000
    p = tmp_path / "t.ibu"
    _write_file(oracle, p, n)
    with open(p, "r+b") as f:
        f.truncate(32 + 24 * n - 5)
    r = ia.Reader.from_path(p)
    with pytest.raises(ia.IbuError) as e:
        r.process_device(ctx, ia.PROC_REDUCE, ring=SMALL_RING)
    assert e.value.kind == "TruncatedRecord"
    r.close()
    # the context is still usable afterwards (ring drained, no stuck stream)
    d = ctx.alloc(24 * 256)
    ctx.generate(1, 0, 256, 16, 12, d)
    assert ctx.reduce(d, 256)["count"] == 256


def test_stream_stats_report_kernel_and_total_time(ia, ctx, oracle, tmp_path):
    n = 2_000_000
    p = tmp_path / "s.ibu"
    _write_file(oracle, p, n)
    m = ia.MmapReader.new(p)
    _, st = m.process_device(ctx, ia.PROC_REDUCE, ring={"slots": 4, "slot_records": 262_144, "feeder_threads": 4})
    assert st.records == n and st.batches == (n + 262_143) // 262_144 and st.bytes_h2d == 24 * n
    assert 0 < st.seconds_kernel <= st.seconds_total
    m.close()


# ---- host <-> host codec pipelines (csrc/codec_stream.cpp) ----------------------------------------------------------
TINY_RING = {"slots": 2, "slot_records": 1000, "feeder_threads": 1}


@pytest.mark.parametrize("n", [0, 1, 127, 1024, 1025, 100_000, 700_001])
@pytest.mark.parametrize("lens", [(16, 12), (32, 32), (15, 11)])
@pytest.mark.parametrize("ring", [None, SMALL_RING, TINY_RING])
def test_mmap_decode_to_host(ia, ctx, oracle, tmp_path, n, lens, ring):
    if ring is TINY_RING and n > 200_000:
        pytest.skip("1 Ki-record slots over 7e5 records adds nothing but time")
    bc_len, umi_len = lens
    p = tmp_path / "dh.ibu"
    recs = _write_file(oracle, p, n, bc_len, umi_len)
    wbc, wumi, widx = oracle.decode_records(recs, bc_len, umi_len)
    m = ia.MmapReader.new(p)
    bc, umi, idx, st = m.decode_to_host(ctx, ring=ring)
    assert bc.tobytes() == wbc.tobytes() and umi.tobytes() == wumi.tobytes() and idx.tobytes() == widx.tobytes()
    assert st.records == n and st.bytes_h2d == 24 * n and st.bytes_d2h == n * (bc_len + umi_len + 8)
    # shards concatenate; skipped columns stay untouched
    parts = [m.decode_to_host(ctx, shard=s, n_shards=3, ring=ring, want=("umi",)) for s in range(3)]
    assert all(pt[0] is None and pt[2] is None for pt in parts)
    assert b"".join(pt[1].tobytes() for pt in parts) == wumi.tobytes()
    m.close()


@pytest.mark.parametrize("n", [0, 1, 127, 1024, 1025, 100_000, 700_001])
@pytest.mark.parametrize("lens", [(16, 12), (32, 32), (15, 11)])
@pytest.mark.parametrize("with_index", [True, False])
def test_writer_write_ascii_batch(ia, ctx, oracle, tmp_path, n, lens, with_index):
    bc_len, umi_len = lens
    recs = oracle.generate(SEED, 9, n, bc_len, umi_len)  # index column 9, 10, ...
    bc, umi, idx = oracle.decode_records(recs, bc_len, umi_len)
    if n > 10:  # lower case is accepted (record.rs:22-25 table is case-blind here, emitted upper on decode)
        bc = bc.copy()
        bc.reshape(-1)[: bc_len * 5] = np.frombuffer(bc.reshape(-1)[: bc_len * 5].tobytes().lower(), dtype=np.uint8)
    p = tmp_path / "enc.ibu"
    w = ia.Writer.from_path(p, ia.Header(bc_len, umi_len))
    ring = SMALL_RING if n < 300_000 else None
    st = w.write_ascii_batch(ctx, bc, umi, bc_len, umi_len, index=idx if with_index else None, first_index=9, ring=ring)
    assert st.records == n and w.records_written() == n
    w.finish()
    w.close()
    q = tmp_path / "want.ibu"
    _write_file(oracle, q, n, bc_len, umi_len, first=9)
    assert p.read_bytes() == q.read_bytes()


def test_writer_write_ascii_batch_invalid_base(ia, ctx, oracle):
    bc_len, umi_len, n = 16, 12, 10_000
    recs = oracle.generate(SEED, 0, n, bc_len, umi_len)
    bc, umi, idx = oracle.decode_records(recs, bc_len, umi_len)
    bc = bc.copy().reshape(n, bc_len)
    umi = umi.copy().reshape(n, umi_len)
    bc[5000, 3] = ord("N")
    umi[7777, 0] = 0
    w = ia.Writer.new_headless()
    with pytest.raises(ia.IbuError) as e:
        w.write_ascii_batch(ctx, bc, umi, bc_len, umi_len, ring=TINY_RING)  # 1024-row batches
    assert e.value.kind == "InvalidBase" and e.value.first_bad == 5000 and e.value.n_bad == 2
    w.finish()
    # batches before the one holding row 5000 (rows 0..4095) were written, nothing after
    assert w.records_written() == 4096
    assert w.inner_bytes() == recs[:4096].tobytes()
    # the context and the writer stay usable
    w.write_ascii_batch(ctx, bc[:100], umi[:100], bc_len, umi_len, first_index=4096, ring=TINY_RING)
    assert w.records_written() == 4196


def test_host_codec_roundtrip_large(ia, ctx, oracle, tmp_path):
    """file -> decode_to_host -> write_ascii_batch -> identical file (3e6 records, default ring)."""
    n = 3_000_000
    p = tmp_path / "big.ibu"
    _write_file(oracle, p, n)
    m = ia.MmapReader.new(p)
    bc, umi, idx, st = m.decode_to_host(ctx)
    q = tmp_path / "back.ibu"
    w = ia.Writer.from_path(q, m.header())
    w.write_ascii_batch(ctx, bc, umi, 16, 12, index=idx)
    w.finish()
    w.close()
    m.close()
    assert p.read_bytes() == q.read_bytes()


@pytest.mark.parametrize("kind", ["plain", "gzip"])
def test_decode_sink_too_small_is_an_error_not_an_overrun(ia, ctx, oracle, tmp_path, kind):
    """ibu_decode_sink_t.cap_records (ABI revision 3): a compressed stream does not say how many records it holds, so a
    sink sized for fewer must end in InvalidArg — with the rows that did fit decoded and nothing written past them."""
    import gzip
    n, cap = 30_000, 10_000
    path = tmp_path / ("s.ibu" + (".gz" if kind == "gzip" else ""))
    recs = _write_file(oracle, tmp_path / "plain.ibu", n, 16, 12)
    raw = open(tmp_path / "plain.ibu", "rb").read()
    with open(path, "wb") as f:
        f.write(gzip.compress(raw, 1) if kind == "gzip" else raw)
    guard = 4096  # bytes behind the capacity that must stay untouched
    d_bc, d_umi, d_idx = ctx.alloc(cap * 16 + guard), ctx.alloc(cap * 12 + guard), ctx.alloc(cap * 8 + guard)
    for d in (d_bc, d_umi, d_idx):
        d.upload(np.full(d.nbytes, 0xEE, dtype=np.uint8))
    r = ia.Reader.from_path(path)
    with pytest.raises(ia.IbuError) as e:
        r.process_device(ctx, ia.PROC_DECODE, sink=(d_bc, d_umi, d_idx, cap), ring=SMALL_RING)
    assert e.value.kind == "InvalidArg" and e.value.b == cap and cap < e.value.a <= n
    r.close()
    for d, w in ((d_bc, 16), (d_umi, 12), (d_idx, 8)):
        assert (d.download(offset=cap * w) == 0xEE).all()          # nothing past the capacity
    bc, _, _ = oracle.decode_records(recs, 16, 12)
    done = (cap // 4096) * 4096                                     # whole batches that fitted were decoded
    assert d_bc.download(count=done * 16).tobytes() == bc[: done * 16].tobytes()
    # the mmap form knows the shard size up front: refused before any work
    m = ia.MmapReader.new(tmp_path / "plain.ibu")
    with pytest.raises(ia.IbuError) as e:
        m.process_device(ctx, ia.PROC_DECODE, sink=(d_bc, d_umi, d_idx, cap), ring=SMALL_RING)
    assert e.value.kind == "InvalidArg" and (e.value.a, e.value.b) == (n, cap)
    _, st = m.process_device(ctx, ia.PROC_DECODE, shard=0, n_shards=3, sink=(d_bc, d_umi, d_idx, cap), ring=SMALL_RING)
    assert st.records == n // 3
    with pytest.raises(TypeError):
        m.process_device(ctx, ia.PROC_DECODE, sink=(d_bc.ptr, d_umi.ptr, d_idx.ptr), ring=SMALL_RING)  # raw pointers need a capacity
    m.close()


# ---- process_parallel with a GPU per worker, one call (ibu_mmap_process_devices / _contexts; mmap.rs:286-332) ---------------
@pytest.mark.parametrize("n", [0, 1, 3, 10_000, 1_000_003])
@pytest.mark.parametrize("devices", [(0,), (0, 0), (0, 0, 0, 0, 0)])
def test_process_devices_reduce_equals_the_oracles_process_parallel(ia, oracle, tmp_path, n, devices):
    """One call drives one host thread + context per listed device over the reference's static split; the per-device
    partials are the oracle's per-shard sums and their host-side total is what the reference's process_parallel (the
    oracle's restatement, same number of workers) reports.  N DISTINCT devices is unmeasured here: the test box has one
    GPU, so the list repeats ordinal 0 (legal, and the same code path: a thread and a context per entry)."""
    p = tmp_path / "m.ibu"
    recs = _write_file(oracle, p, n)
    m = ia.MmapReader.new(p)
    total, parts, stats = m.process_devices(devices, ia.PROC_REDUCE, ring=SMALL_RING)
    want = oracle.Mmap(str(p)).process_parallel(len(devices), cores=64)
    assert total == {"count": want.count, "sum": list(want.sum), "xor": list(want.xor_)} == oracle.reduce_records(recs)
    for i in range(len(devices)):
        a, b = oracle.shard_range(n, len(devices), i)
        assert parts[i] == oracle.reduce_records(recs[a:b]) and stats[i].records == b - a
    # ... and equals the single-device call
    c = ia.Context(0)
    one, _ = m.process_device(c, ia.PROC_REDUCE, ring=SMALL_RING)
    assert one == total
    # the context form reuses caller-owned contexts (twice: the rings survive)
    cs = [ia.Context(0) for _ in devices]
    for _ in range(2):
        t2, p2, _ = m.process_devices(proc=ia.PROC_REDUCE, ring=SMALL_RING, contexts=cs)
        assert t2 == total and p2 == parts
    with pytest.raises(ia.IbuError) as e:
        m.process_devices(proc=ia.PROC_REDUCE, contexts=[cs[0], cs[0]])
    assert e.value.kind == "InvalidArg"
    for x in cs + [c]:
        x.close()
    m.close()


def test_process_devices_all_visible_devices(ia, oracle, tmp_path):
    """devices = () -> every visible device, as num_threads == 0 means every core (mmap.rs:292-296)."""
    n = 70_001
    p = tmp_path / "all.ibu"
    recs = _write_file(oracle, p, n)
    m = ia.MmapReader.new(p)
    total, parts, _ = m.process_devices((), ia.PROC_REDUCE)
    assert len(parts) == ia.device_count() >= 1 and total == oracle.reduce_records(recs)
    m.close()


@pytest.mark.parametrize("lens", [(16, 12), (15, 11)])
def test_process_devices_decode_concatenates_in_shard_order(ia, ctx, oracle, tmp_path, lens):
    bc_len, umi_len = lens
    n, devices = 200_003, (0, 0, 0)
    p = tmp_path / "d.ibu"
    recs = _write_file(oracle, p, n, bc_len, umi_len)
    bc, umi, idx = oracle.decode_records(recs, bc_len, umi_len)
    m = ia.MmapReader.new(p)
    sinks = []
    for s in range(len(devices)):
        a, b = ia.shard_range(n, len(devices), s)
        sinks.append((ctx.alloc((b - a) * bc_len), ctx.alloc((b - a) * umi_len), ctx.alloc((b - a) * 8), b - a))
    count, _, stats = m.process_devices(devices, ia.PROC_DECODE, sinks=sinks, ring=SMALL_RING)
    assert count == n and sum(s.records for s in stats) == n
    assert b"".join(s[0].download().tobytes() for s in sinks) == bc.tobytes()
    assert b"".join(s[1].download().tobytes() for s in sinks) == umi.tobytes()
    assert b"".join(s[2].download().tobytes() for s in sinks) == idx.tobytes()
    m.close()


def test_process_devices_first_error_in_worker_order_wins(ia, ctx, oracle, tmp_path):
    """mmap.rs:326-328 (Q12): handles are joined in spawn order and the first Err is the call's.  Worker 1 and worker 2
    both fail (sinks too small); the error reported is worker 1's, with ITS detail — not worker 2's, whichever finished
    first.  A bad device ordinal fails before any worker starts."""
    n = 90_000
    p = tmp_path / "e.ibu"
    _write_file(oracle, p, n)
    m = ia.MmapReader.new(p)
    k = n // 3
    big = lambda: (ctx.alloc(k * 16), ctx.alloc(k * 12), ctx.alloc(k * 8), k)
    sinks = [big(), big()[:3] + (k - 7,), big()[:3] + (k - 9,)]
    with pytest.raises(ia.IbuError) as e:
        m.process_devices((0, 0, 0), ia.PROC_DECODE, sinks=sinks, ring=SMALL_RING)
    assert e.value.kind == "InvalidArg" and (e.value.a, e.value.b) == (k, k - 7)
    with pytest.raises(ia.IbuError) as e:
        m.process_devices((0, 99), ia.PROC_REDUCE)
    assert e.value.kind == "NoDevice"
    with pytest.raises(ValueError):
        m.process_devices((0, 0), ia.PROC_DECODE, sinks=sinks[:1])
    m.close()


def test_alloc_probed_keeps_one_usable_candidate(ia, ctx, oracle):
    """ibu_device_alloc_probed: `tries` candidates are timed (write + read over the whole range), one is kept, the rest
    freed; what comes back is ordinary device memory.  No rate is asserted (that is bench.py's to report)."""
    n = 2_000_003
    free0 = None
    buf, rep = ctx.alloc_probed(24 * n, 4)
    assert rep["tries"] == 4 and 0 <= rep["chosen"] < 4 and len(rep["ms"]) == 4 and all(v > 0 for v in rep["ms"])
    assert rep["ms"][rep["chosen"]] == min(rep["ms"])
    ctx.generate(SEED, 0, n, 16, 12, buf)
    want = oracle.generate(SEED, 0, n, 16, 12)
    assert buf.download().tobytes() == want.tobytes()
    assert ctx.reduce(buf, n) == oracle.reduce_records(want)       # the probe left the context's accumulator reset
    buf.free()
    small, rep = ctx.alloc_probed(1000, 8)                          # too small to measure: a plain allocation
    assert rep["tries"] == 1 and rep["chosen"] == 0
    small.free()
    one, rep = ctx.alloc_probed(24 * n, 1)
    assert rep["tries"] == 1
    one.free()
    del free0


def test_alloc_probe_tries_option_makes_placement_the_librarys(ia, oracle, tmp_path):
    """Option "alloc_probe_tries": the arrays the library allocates for a caller — ibu_device_alloc, the destination of
    ibu_load_to_device — are placement-probed from 256 MiB on with an explicit count (from 1 GiB on in the default auto mode);
    below that they are plain allocations.  What comes back is ordinary memory with the right contents; the option is validated."""
    c = ia.Context(0)
    ring = {"slots": 3, "slot_records": 1 << 20, "feeder_threads": 4}
    try:
        with pytest.raises(ia.IbuError):
            c.set_option("alloc_probe_tries", -1)
        with pytest.raises(ia.IbuError):
            c.set_option("alloc_probe_tries", 17)
        c.set_option("alloc_probe_tries", 0)                      # auto (the default): probes from 1 GiB on when three candidates fit
        c.set_option("alloc_probe_tries", 3)
        n = (256 << 20) // 24 + 1001                              # just past the threshold
        d = c.alloc(24 * n)
        c.generate(SEED, 5, n, 16, 12, d)
        want = c.reduce(d, n)                                     # the probe left the accumulator reset
        assert want["count"] == n
        p = tmp_path / "big.ibu"
        w = ia.Writer.from_path(str(p), ia.Header(16, 12))
        w.write_batch_device(c, d, n, ring=ring)
        w.finish()
        w.close()
        d.free()
        _, dptr, got_n, _ = c.load_to_device(str(p), ring=ring)   # the library allocates the destination: probed
        assert got_n == n and c.reduce(dptr, n) == want
        c.free(dptr)
        small = c.alloc(4096)                                     # below the threshold: plain
        small.free()
    finally:
        c.close()


def test_probed_allocations_leave_the_reduce_accumulator_alone(ia, oracle, capfd):
    """reset; reduce(A); <a probed allocation>; reduce(B); fetch adds up A and B (the probe used to run its read half
    into the context's accumulator and leave it reset).  Also the default auto mode: 1.2 GiB draws candidates and says so."""
    c = ia.Context(0)
    try:
        n = 1_000_003
        a, b = c.alloc(24 * n), c.alloc(24 * n)
        c.generate(SEED, 0, n, 16, 12, a)
        c.generate(SEED, n, n, 16, 12, b)
        want = oracle.reduce_records(oracle.generate(SEED, 0, 2 * n, 16, 12))
        c.reduce(a, n, reset=True, fetch=False)
        big, rep = c.alloc_probed(300 << 20, 3)                   # explicit probing in the middle of an accumulation
        assert rep["tries"] == 3
        c.reduce(b, n, reset=False, fetch=False)
        assert c.reduce_fetch() == want
        big.free()
        capfd.readouterr()
        c.reduce(a, n, reset=True, fetch=False)
        auto = c.alloc(1288490189)                                # 1.2 GiB under the default option: auto-probed (IBU_TRACE_SORT says so)
        c.reduce(b, n, reset=False, fetch=False)
        assert c.reduce_fetch() == want
        err = capfd.readouterr().err
        if os.environ.get("IBU_TRACE_SORT", "") not in ("", "0"):     # probed — or, when the driver was slow to hand out memory, knowingly not
            assert "ibu alloc: 1288490189 bytes probed" in err or "ibu alloc: 1288490189 bytes not probed: the allocation took" in err, err
        auto.free()
        c.set_option("alloc_probe_tries", 1)                      # never probe
        plain = c.alloc(1288490189)
        assert "probed" not in capfd.readouterr().err
        plain.free()
    finally:
        c.close()


def test_an_allocation_waits_for_candidates_still_being_freed_instead_of_failing(ia, capfd):
    """The candidates a placement probe did not keep are freed on a helper thread (the driver clears VRAM at free time: seconds for
    tens of GB).  Until it is done their memory is still taken: an allocation of the library's that does not fit in the meantime
    waits for the helper and tries again — the caller sees the result the synchronous free would have given, not an out-of-memory
    error.  Three candidates of 0.3 of the free memory each, then the same size again at once."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    free_b, total_b = C.c_size_t(0), C.c_size_t(0)
    c = ia.Context(0)
    try:
        assert hip.hipMemGetInfo(C.byref(free_b), C.byref(total_b)) == 0
        size = (int(free_b.value * 0.30) // 4096) * 4096           # three fit, a fourth does not; two kept ones do
        assert size >= (1 << 30)
        c.set_option("alloc_probe_tries", 3)
        capfd.readouterr()
        a = c.alloc(size)                                          # three candidates, two of them on their way back
        b = c.alloc(size)                                          # 0.1 of the memory is free right now unless the helper has finished
        err = capfd.readouterr().err
        n = 1_000_003
        c.generate(SEED, 0, n, 16, 12, b)                          # ordinary memory
        assert c.reduce(b, n)["count"] == n
        if os.environ.get("IBU_TRACE_SORT", "") not in ("", "0"):
            assert "bytes probed: 3 candidates" in err, err
            # (whether the second call had to wait depends on how fast the driver cleared 0.6 of the memory; when it had to, it says so)
            assert "still does not fit" not in err, err
        a.free()
        b.free()
        c.set_option("alloc_probe_tries", 0)                       # (setting the option waits for whatever is still being freed)
        assert hip.hipMemGetInfo(C.byref(free_b), C.byref(total_b)) == 0
        assert free_b.value >= 3 * size                            # everything is back
    finally:
        c.close()
