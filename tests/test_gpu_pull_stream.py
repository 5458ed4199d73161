"""GPU tests of the pull-style device record stream (ibu_stream_*, csrc/stream.cpp) through the C ABI: the device form of
Reader::read_batch + Iterator (reader.rs:218-242, :279-306) and of the per-batch loop of process_parallel (mmap.rs:312-320).
Every source (plain file, gzip, BGZF, a pipe, a truncated file, mmap shards) is pulled batch by batch and compared with what
the CPU oracle's Reader yields: the concatenation of the batches byte for byte, the batch boundaries on whole refills of
49 152 records, the records delivered in front of a TruncatedRecord and its position."""
import gzip
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x1B00005
REFILL = 49_152                                                     # records per reference refill (reader.rs:14)
SMALL = {"slots": 3, "slot_records": 4096, "feeder_threads": 2}      # slots smaller than one refill: the staging path
MID = {"slots": 3, "slot_records": 2 * REFILL + 1000, "feeder_threads": 2}   # two whole refills per batch (rounded up to 128)
ONE = {"slots": 2, "slot_records": REFILL, "feeder_threads": 1}      # exactly one refill per slot


@pytest.fixture(scope="module")
def ia():
    import ibu_amd
    return ibu_amd


@pytest.fixture(scope="module")
def ctx(ia):
    c = ia.Context(0)
    yield c
    c.close()


def _write_file(oracle, path, n, bc_len=16, umi_len=12, first=0):
    recs = oracle.generate(SEED, first, n, bc_len, umi_len)
    w = oracle.Writer(header=oracle.header_new(bc_len, umi_len), path=str(path))
    if n:
        w.write_batch(recs)
    w.finish()
    w.drop()
    return recs


def _pull_all(stream):
    """[(first_index, n, host records)] of every batch, released in order."""
    out = []
    for b in stream:
        with b:
            out.append((b.first_index, b.n, b.download().copy()))
    return out


def _oracle_iterate(oracle, path):
    """What the reference's iterator yields: (records before the end or the first Err, the Err or None)."""
    r = oracle.Reader(path=str(path))
    got, err = [], None
    while True:
        try:
            x = r.next()
        except oracle.OracleError as e:
            err = e
            break
        if x is None:
            break
        got.append(x)
    return got, err


def _as_tuples(recs):
    return [(int(a), int(b), int(c)) for a, b, c in zip(recs["barcode"], recs["umi"], recs["index"])]


@pytest.mark.parametrize("form", ["plain", "gzip", "bgzf"])
@pytest.mark.parametrize("n", [0, 1, REFILL - 1, REFILL, 100_000, 500_003])
@pytest.mark.parametrize("ring", [SMALL, ONE, MID, None])
def test_reader_stream_batches_concatenate_to_the_oracles_record_sequence(ia, ctx, oracle, tmp_path, form, n, ring):
    if ring is None and n not in (0, 500_003):
        pytest.skip("the default ring (96 MiB slots) takes one batch for all of these")
    p = tmp_path / "s.ibu"
    recs = _write_file(oracle, p, n)
    path = p
    if form == "gzip":
        path = tmp_path / "s.ibu.gz"
        raw = p.read_bytes()
        with open(path, "wb") as f:                                 # two members
            f.write(gzip.compress(raw[:len(raw) // 3], 1))
            f.write(gzip.compress(raw[len(raw) // 3:], 1))
    elif form == "bgzf":
        from tests.bgzf import bgzf_compress
        path = tmp_path / "s.ibu.bgz"
        path.write_bytes(bgzf_compress(p.read_bytes()))
    r = ia.Reader.from_path(path)
    with r.device_stream(ctx, ring=ring) as s:
        h = s.header()
        assert (h.bc_len, h.umi_len) == (16, 12)
        batches = _pull_all(s)
        assert s.next_batch() is None and s.next_batch() is None     # the end is the end on every later call
        st = s.stats()
    r.close()
    assert st.records == n and st.bytes_h2d == 24 * n and st.batches == len(batches)
    cat = np.concatenate([b[2] for b in batches]) if batches else np.empty(0, ia.REC_DTYPE)
    assert cat.tobytes() == recs.tobytes()
    # first_index counts the records delivered before the batch; a Reader source delivers whole refills per batch
    pos = 0
    slot = (ring or {}).get("slot_records", 4 * ia.BATCH_SIZE)
    slot = (slot + 127) // 128 * 128
    for k, (first, bn, _) in enumerate(batches):
        assert first == pos and 0 < bn <= slot
        if k + 1 < len(batches):
            assert bn % REFILL == 0 or slot < REFILL, (k, bn)
        pos += bn
    # the oracle's read_batch sequence covers the same records (its batches are single refills)
    o = oracle.Reader(path=str(p))
    seen = 0
    while o.read_batch():
        seen = (o.bytes_read - 32) // 24                             # bytes_read counts the header too
    assert seen == n


@pytest.mark.parametrize("ring", [SMALL, ONE, MID])
@pytest.mark.parametrize("n,cut", [(1, 5), (1000, 5), (REFILL, 1), (REFILL + 1, 23), (2 * REFILL + 77, 10), (5 * REFILL + 4096, 7),
                                   (3 * REFILL, 24 * 5 + 3)])
def test_truncated_stream_delivers_exactly_what_the_reference_iterator_yields(ia, ctx, oracle, tmp_path, ring, n, cut):
    """Quirk Q8 (reader.rs:232-237): the final, partial refill is dropped whole; TruncatedRecord.pos is the reference's."""
    p = tmp_path / "t.ibu"
    _write_file(oracle, p, n)
    with open(p, "r+b") as f:
        f.truncate(32 + 24 * n - cut)
    want, err = _oracle_iterate(oracle, p)
    assert err is not None and err.name == "TruncatedRecord"
    r = ia.Reader.from_path(p)
    s = r.device_stream(ctx, ring=ring)
    got = []
    with pytest.raises(ia.IbuError) as e:
        for b in s:
            with b:
                got.extend(_as_tuples(b.download()))
    assert e.value.kind == "TruncatedRecord" and e.value.pos == err.a
    assert len(got) == len(want) == (n * 24 - cut) // (REFILL * 24) * REFILL
    assert got == want
    with pytest.raises(ia.IbuError) as e2:                           # the error is the stream's state from now on
        s.next_batch()
    assert e2.value.kind == "TruncatedRecord" and e2.value.pos == err.a
    s.close()
    r.close()
    d = ctx.alloc(24 * 256)                                          # the context is usable afterwards
    ctx.generate(1, 0, 256, 16, 12, d)
    assert ctx.reduce(d, 256)["count"] == 256


@pytest.mark.parametrize("form", ["plain", "gzip"])
def test_stream_from_a_pipe(ia, ctx, oracle, tmp_path, form):
    """Reader::from_stdin's case (reader.rs:389-396): a descriptor that cannot seek and reads short."""
    n = 300_007
    p = tmp_path / "pipe.ibu"
    recs = _write_file(oracle, p, n)
    data = p.read_bytes() if form == "plain" else gzip.compress(p.read_bytes(), 1)
    rfd, wfd = os.pipe()

    def feed():
        rng = np.random.default_rng(7)
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
        with r.device_stream(ctx, ring=MID) as s:
            batches = _pull_all(s)
        r.close()
    finally:
        t.join()
        os.close(rfd)
    assert np.concatenate([b[2] for b in batches]).tobytes() == recs.tobytes()


def test_stream_takes_over_after_partial_host_iteration(ia, ctx, oracle, tmp_path):
    """Records the host iterator already yielded are not delivered again; the rest of its buffer goes first."""
    n = 3 * REFILL + 500
    p = tmp_path / "part.ibu"
    recs = _write_file(oracle, p, n)
    r = ia.Reader.from_path(p)
    head = [next(r) for _ in range(7)]
    assert [tuple(x) for x in head] == _as_tuples(recs[:7])
    with r.device_stream(ctx, ring=ONE) as s:
        batches = _pull_all(s)
    r.close()
    assert batches[0][0] == 0 and batches[0][1] == REFILL - 7         # what was left of the reader's own refill
    assert np.concatenate([b[2] for b in batches]).tobytes() == recs[7:].tobytes()


@pytest.mark.parametrize("n_shards", [1, 3, 8])
@pytest.mark.parametrize("n", [0, 5, 4096, 100_003])
def test_mmap_stream_shards_concatenate_and_number_records_by_their_position(ia, ctx, oracle, tmp_path, n, n_shards):
    p = tmp_path / "m.ibu"
    recs = _write_file(oracle, p, n)
    m = ia.MmapReader.new(p)
    parts = []
    for sh in range(n_shards):
        a, b = ia.shard_range(n, n_shards, sh)                       # mmap.rs:297-307
        with m.device_stream(ctx, shard=sh, n_shards=n_shards, ring=SMALL) as s:
            batches = _pull_all(s)
        pos = a
        for first, bn, host in batches:
            assert first == pos and np.array_equal(host["index"], np.arange(pos, pos + bn, dtype=np.uint64))
            pos += bn
        assert pos == b
        parts.extend(x[2] for x in batches)
    cat = np.concatenate(parts) if parts else np.empty(0, ia.REC_DTYPE)
    assert cat.tobytes() == recs.tobytes()
    m.close()


def test_batches_may_be_held_and_released_in_any_order_but_not_all_of_them(ia, ctx, oracle, tmp_path):
    n = 10 * 4096
    p = tmp_path / "h.ibu"
    recs = _write_file(oracle, p, n)
    m = ia.MmapReader.new(p)
    s = m.device_stream(ctx, ring=SMALL)                             # 3 slots
    a, b = s.next_batch(), s.next_batch()
    c = s.next_batch()                                               # every slot is now held
    with pytest.raises(ia.IbuError) as e:
        s.next_batch()
    assert e.value.kind == "InvalidArg"
    host_b = b.download().copy()
    b.release()                                                      # out of order: the middle one first
    with pytest.raises(ia.IbuError) as e:                            # only what the stream handed out and still holds can be released
        ia._check(ia.lib.ibu_stream_release(s._s, a.ptr + 24, None))
    assert e.value.kind == "InvalidArg"
    host_a, host_c = a.download().copy(), c.download().copy()
    a.release()
    c.release()
    rest = _pull_all(s)
    s.close()
    got = np.concatenate([host_a, host_b, host_c] + [x[2] for x in rest])
    assert got.tobytes() == recs.tobytes()
    m.close()


def test_one_free_slot_keeps_the_stream_moving_while_the_others_are_held(ia, ctx, oracle, tmp_path):
    """slots - 1 batches held for the whole stream: everything else flows through the one slot left (the producer takes any
    slot the caller does not hold, not the next one in ring order)."""
    n = 12 * 4096 + 5
    p = tmp_path / "one.ibu"
    recs = _write_file(oracle, p, n)
    m = ia.MmapReader.new(p)
    with m.device_stream(ctx, ring=SMALL) as s:                      # 3 slots
        a, b = s.next_batch(), s.next_batch()
        host = [a.download().copy(), b.download().copy()]
        ptrs = {a.ptr, b.ptr}
        while True:
            c = s.next_batch()
            if c is None:
                break
            assert c.ptr not in ptrs                                 # always the third slot
            host.append(c.download().copy())
            c.release()
        a.release()
        b.release()
    assert np.concatenate(host).tobytes() == recs.tobytes()
    m.close()


def test_the_ring_is_lent_while_a_stream_is_open(ia, ctx, oracle, tmp_path):
    p = tmp_path / "l.ibu"
    _write_file(oracle, p, 10_000)
    m = ia.MmapReader.new(p)
    s = m.device_stream(ctx, ring=SMALL)
    for call in (lambda: ctx.load_to_device(p, ring=SMALL), lambda: m.process_device(ctx, ia.PROC_REDUCE, ring=SMALL),
                 lambda: m.device_stream(ctx, ring=SMALL)):
        with pytest.raises(ia.IbuError) as e:
            call()
        assert e.value.kind == "InvalidArg"
    d = ctx.alloc(24 * 1000)                                         # kernel entry points are not ring users
    ctx.generate(3, 0, 1000, 16, 12, d)
    assert ctx.reduce(d, 1000)["count"] == 1000
    s.close()                                                        # closing with nothing pulled is fine
    res, _ = m.process_device(ctx, ia.PROC_REDUCE, ring=SMALL)
    assert res["count"] == 10_000
    m.close()


def test_caller_side_processor_decode_then_sort_each_batch(ia, ctx, oracle, tmp_path):
    """The user half of ParallelProcessor (parallel.rs:100-190) on device batches: the caller's own per-batch work — here the
    library's decode and sort — on a SECOND stream (another context's), with next()/release() ordering that stream."""
    n, bc_len, umi_len = 5 * REFILL + 321, 16, 12
    p = tmp_path / "proc.ibu.gz"
    raw = tmp_path / "proc.ibu"
    recs = _write_file(oracle, raw, n, bc_len, umi_len)
    p.write_bytes(gzip.compress(raw.read_bytes(), 1))
    other = ia.Context(0)
    st = other.stream
    slot = 2 * REFILL
    d_bc, d_umi, d_idx = ctx.alloc(n * bc_len), ctx.alloc(n * umi_len), ctx.alloc(n * 8)
    d_sorted, d_tmp = ctx.alloc(slot * 24), ctx.alloc(slot * 24)
    r = ia.Reader.from_path(p)
    sorted_batches = []
    with r.device_stream(ctx, ring={"slots": 3, "slot_records": slot, "feeder_threads": 2}) as s:
        while True:
            b = s.next_batch(stream=st)
            if b is None:
                break
            row = b.first_index
            other.decode_ascii(b.ptr, b.n, bc_len, umi_len, d_bc.ptr + row * bc_len, d_umi.ptr + row * umi_len, d_idx.ptr + row * 8, stream=st)
            other.copy(d_sorted, b.ptr, b.n * 24, stream=st)
            b.release(stream=st)                                     # the slot may be refilled once decode and copy have run
            other.sort_records(d_sorted, d_tmp, b.n, stream=st)
            other.synchronize(st)
            sorted_batches.append((row, b.n, d_sorted.download(count=b.n * 24).view(ia.REC_DTYPE).copy()))
    r.close()
    other.synchronize(st)
    bc, umi, idx = oracle.decode_records(recs, bc_len, umi_len)
    assert d_bc.download().tobytes() == bc.tobytes() and d_umi.download().tobytes() == umi.tobytes()
    assert d_idx.download(np.uint64).tobytes() == idx.tobytes()
    for row, bn, got in sorted_batches:
        assert got.tobytes() == oracle.sort_records(recs[row:row + bn]).tobytes()
    other.close()


def test_the_copies_fill_exactly_the_batch_and_nothing_behind_it(ia, ctx, oracle, tmp_path):
    """Guard check on the ring's device slots: a short last batch leaves the rest of its slot untouched."""
    ring = {"slots": 2, "slot_records": 4096, "feeder_threads": 1}
    p = tmp_path / "g.ibu"
    _write_file(oracle, p, 3 * 4096)
    m = ia.MmapReader.new(p)
    with m.device_stream(ctx, ring=ring) as s:
        ptrs = sorted(set(iter(lambda: _take(s), None)))
    assert len(ptrs) == 2
    pattern = np.full(4096 * 24, 0xA5, np.uint8)
    for q in ptrs:                                                   # the ring outlives the stream: same slots next time
        ia.DeviceBuffer.wrap(ctx, q, pattern.size).upload(pattern)
    n = 4096 + 37
    q2 = tmp_path / "g2.ibu"
    recs = _write_file(oracle, q2, n, first=9)
    m2 = ia.MmapReader.new(q2)
    with m2.device_stream(ctx, ring=ring) as s:
        b0 = s.next_batch()
        b1 = s.next_batch()
        assert {b0.ptr, b1.ptr} == set(ptrs) and b1.n == 37
        whole = ia.DeviceBuffer.wrap(ctx, b1.ptr, pattern.size).download()
        assert whole[:37 * 24].tobytes() == recs[4096:].tobytes()
        assert np.all(whole[37 * 24:] == 0xA5)
        b0.release()
        b1.release()
    m.close()
    m2.close()


def _take(s):
    """Pull and release one batch; its device pointer, None at the end."""
    b = s.next_batch()
    if b is None:
        return None
    p = b.ptr
    b.release()
    return p


def test_process_device_and_the_pull_stream_are_one_pipeline(ia, ctx, oracle, tmp_path):
    """ibu_mmap_process_device / ibu_reader_process_device are the stream with a built-in loop body: same results, same stats."""
    n = 700_001
    p = tmp_path / "one.ibu"
    recs = _write_file(oracle, p, n)
    m = ia.MmapReader.new(p)
    res, st = m.process_device(ctx, ia.PROC_REDUCE, ring=MID)
    assert res == oracle.reduce_records(recs)
    acc = None
    with m.device_stream(ctx, ring=MID) as s:
        nb = 0
        for b in s:
            with b:
                ctx.reduce(b.ptr, b.n, reset=(nb == 0), fetch=False)
                nb += 1
        acc = ctx.reduce_fetch()
        st2 = s.stats()
    assert acc == res and st2.records == st.records == n and st2.batches == st.batches == nb
    m.close()


def test_destroying_the_context_under_an_open_stream_is_safe(ia, oracle, tmp_path):
    """ibu_ctx_destroy with a stream still open: the producer is stopped and the ring returned before the context goes; the handle
    stays closable, every other call on it reports the destroyed context."""
    p = tmp_path / "o.ibu"
    _write_file(oracle, p, 50_000)
    c = ia.Context(0)
    m = ia.MmapReader.new(p)
    s = m.device_stream(c, ring=SMALL)
    b = s.next_batch()
    assert b.n == 4096
    c.close()                                                        # destroys the context; the stream becomes an orphan
    with pytest.raises(ia.IbuError) as e:
        s.next_batch()
    assert e.value.kind == "InvalidArg"
    with pytest.raises(ia.IbuError):
        b.release()
    s.close()
    m.close()


@pytest.mark.parametrize("ring", [SMALL, ONE, MID])
def test_a_source_error_arrives_after_the_whole_refills_in_front_of_it(ia, ctx, oracle, tmp_path, ring):
    """reader.rs:225-230: `inner.read` failing inside a refill returns the error at once — the refills in front of it were yielded,
    the one under way is lost.  A file-like source that raises EIO after 3.5 refills: three refills of records, then Io."""
    n = 6 * REFILL
    p = tmp_path / "io.ibu"
    recs = _write_file(oracle, p, n)
    data = p.read_bytes()
    fail_at = 32 + 24 * (3 * REFILL + REFILL // 2) + 7

    class Flaky:
        def __init__(self):
            self.pos = 0

        def read(self, k):
            if self.pos >= fail_at:
                raise OSError(5, "injected")
            k = min(k, fail_at - self.pos, 100_003)
            out = data[self.pos:self.pos + k]
            self.pos += k
            return out

    r = ia.Reader(Flaky())
    s = r.device_stream(ctx, ring=ring)
    got = []
    with pytest.raises(ia.IbuError) as e:
        for b in s:
            with b:
                got.append(b.download().copy())
    assert e.value.kind == "Io" and e.value.os_errno == 5
    assert np.concatenate(got).tobytes() == recs[:3 * REFILL].tobytes()
    s.close()
    r.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("IBU_FUZZ_SEEDS_STREAM", "24"))))
def test_pull_stream_fuzz_against_the_oracles_iterator(ia, ctx, oracle, tmp_path, seed):
    """Seeded fuzz: random sizes, ring shapes, source forms (plain / gzip / BGZF Reader, mmap shards), records already taken by
    the host iterator, and cuts inside the last record — what the stream delivers (and the error it ends with) is what the
    oracle's iterator yields on the same bytes."""
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([0, 1, 5, 4095, 4096, REFILL - 1, REFILL, REFILL + 1, 2 * REFILL + 3, int(rng.integers(1, 400_000))]))
    lens = [(16, 12), (32, 32), (5, 7), (31, 1)][seed % 4]
    ring = {"slots": int(rng.integers(2, 6)), "slot_records": int(rng.choice([128, 1000, 4096, REFILL - 128, REFILL, REFILL + 128, 3 * REFILL, 200_000])),
            "feeder_threads": int(rng.integers(1, 5))}
    p = tmp_path / "f.ibu"
    recs = _write_file(oracle, p, n, *lens)
    cut = int(rng.choice([0, 0, int(rng.integers(1, 24)), 24 * int(rng.integers(0, 3)) + int(rng.integers(1, 24))])) if n else 0
    cut = min(cut, 24 * n - 1) if n else 0
    if cut:
        with open(p, "r+b") as f:
            f.truncate(32 + 24 * n - cut)
    form = ["plain", "gzip", "bgzf", "mmap"][int(rng.integers(0, 4))]
    if form == "mmap" and cut % 24:
        form = "plain"                                               # a map refuses such a file at open (InvalidMapSize)
    want, err = _oracle_iterate(oracle, p)
    taken = int(rng.integers(0, 20)) if form != "mmap" and len(want) > 40 else 0
    got = []
    if form == "mmap":
        m = ia.MmapReader.new(p)
        nsh = int(rng.integers(1, 5))
        for sh in range(nsh):
            with m.device_stream(ctx, shard=sh, n_shards=nsh, ring=ring) as s:
                for b in s:
                    with b:
                        got.extend(_as_tuples(b.download()))
        m.close()
        assert got == want and err is None
        return
    path = p
    if form == "gzip":
        path = tmp_path / "f.ibu.gz"
        path.write_bytes(gzip.compress(p.read_bytes(), 1))
    elif form == "bgzf":
        from tests.bgzf import bgzf_compress
        path = tmp_path / "f.ibu.bgz"
        path.write_bytes(bgzf_compress(p.read_bytes()))
    r = ia.Reader.from_path(path)
    head = [tuple(next(r)) for _ in range(taken)]
    s = r.device_stream(ctx, ring=ring)
    seen_err = None
    try:
        for b in s:
            with b:
                got.extend(_as_tuples(b.download()))
    except ia.IbuError as e:
        seen_err = e
    s.close()
    r.close()
    assert head + got == want, (seed, form, n, cut, ring, taken, len(got), len(want))
    if err is None:
        assert seen_err is None
    else:
        assert seen_err is not None and seen_err.kind == err.name and seen_err.pos == err.a
