"""Pins the CPU oracle against the reference's own known-answer material (tests/golden/kat.json,
written by tests/golden/make_golden.py from the literals the reference's unit tests assert).
CPU-only.  The oracle is the checker for every GPU parity test, so it is pinned first."""
import os
import struct

import numpy as np
import pytest


def _recs(orc, formula, n):
    i = np.arange(n, dtype=np.uint64)
    cols = {
        "(i, 2i, 3i)": (i, 2 * i, 3 * i),
        "(i, 0, 0)": (i, 0 * i, 0 * i),
        "(i % 1000, i % 500, i)": (i % 1000, i % 500, i),
        "(i % 1e6, 31 i % 1e6, i)": (i % 1_000_000, (i * 31) % 1_000_000, i),
    }[formula]
    return orc.records_array(np.stack(cols, axis=1))


def _file(orc, tmp_path, name, recs, bc=16, umi=12, sorted_=False):
    h = orc.header_new(bc, umi)
    if sorted_:
        orc.lib.orc_header_set_sorted(h)
    p = str(tmp_path / name)
    w = orc.Writer(h, path=p)
    if len(recs):
        w.write_batch(recs)
    w.finish()
    w.drop()
    return p


# ---- B1-B5 header ----------------------------------------------------------------------------
def test_sizes_and_magic(oracle, kat):
    import ctypes
    assert ctypes.sizeof(oracle.Header) == kat["sizes"]["header"] == 32
    assert ctypes.sizeof(oracle.Record) == kat["sizes"]["record"] == 24
    assert struct.pack("<I", 0x21554249).hex() == kat["magic_bytes"]["hex"]


def test_header_new_bytes(oracle, kat):
    h = oracle.header_new(16, 12)
    assert bytes(h).hex() == kat["header_new_16_12"]["hex"]
    assert not oracle.lib.orc_header_sorted(h)
    oracle.lib.orc_header_set_sorted(h)
    oracle.lib.orc_header_set_sorted(h)  # idempotent, header.rs:266-269
    assert oracle.lib.orc_header_sorted(h)
    assert h.flags == 1
    assert bytes(h).hex() == kat["header_sorted_16_12"]["hex"]
    assert bytes(oracle.header_new(20, 10)).hex() == kat["header_roundtrip_20_10"]["hex"]


def test_validate(oracle, kat):
    for bc, umi in kat["validate_ok"]["cases"]:
        oracle.header_validate(oracle.header_new(bc, umi))
    for case in kat["validate_err"]["cases"]:
        h = oracle.header_from_bytes(bytes.fromhex(case["hex"]))
        with pytest.raises(oracle.OracleError) as ei:
            oracle.header_validate(h)
        assert (ei.value.name, ei.value.a, ei.value.b) == (case["kind"], case["a"], case["b"]), case["src"]


def test_validate_order(oracle):
    # header.rs:167-187: magic is checked before version before bc_len before umi_len
    h = oracle.header_from_bytes(struct.pack("<IIIIQ8s", 1, 9, 0, 0, 0, b"\0" * 8))
    with pytest.raises(oracle.OracleError) as ei:
        oracle.header_validate(h)
    assert ei.value.name == "InvalidMagicNumber"
    h.magic = 0x21554249
    with pytest.raises(oracle.OracleError) as ei:
        oracle.header_validate(h)
    assert ei.value.name == "InvalidVersion"
    h.version = 2
    with pytest.raises(oracle.OracleError) as ei:
        oracle.header_validate(h)
    assert ei.value.name == "InvalidBarcodeLength"


# ---- B6-B8, B17 record -----------------------------------------------------------------------
def test_record_bytes(oracle, kat):
    for case in kat["record_bytes"]["cases"]:
        a = oracle.records_array([case["rec"]])
        assert a.tobytes().hex() == case["hex"], case["src"]
        back = np.frombuffer(bytes.fromhex(case["hex"]), dtype=oracle.REC_DTYPE)
        assert [int(back[0][k]) for k in ("barcode", "umi", "index")] == case["rec"]


def test_ordering(oracle, kat):
    o = kat["ordering"]
    got = oracle.sort_records(oracle.records_array(o["unsorted"]))
    assert got.tobytes() == oracle.records_array(o["sorted"]).tobytes()
    for a, b in o["greater"]:
        assert oracle.record_cmp(a, b) > 0 and oracle.record_cmp(b, a) < 0
    assert oracle.record_cmp([1, 2, 3], [1, 2, 3]) == 0
    assert oracle.is_sorted(got) and not oracle.is_sorted(oracle.records_array(o["unsorted"]))


def test_readme_file(oracle, kat):
    k = kat["readme_file"]
    h = oracle.header_new(16, 12)
    oracle.lib.orc_header_set_sorted(h)
    w = oracle.Writer(h)
    for r in k["records"]:
        w.write_record(r)
    w.finish()
    data = w.into_inner()
    assert len(data) == k["len"] == 80
    assert data.hex() == k["hex"]


# ---- B9-B13 writer -----------------------------------------------------------------------------
def test_writer_lengths(oracle, kat):
    k = kat["writer_lengths"]
    h = oracle.header_new(16, 12)
    assert len(oracle.Writer(h).into_inner()) == k["new"]
    assert len(oracle.Writer(None).into_inner()) == k["headless"]
    w = oracle.Writer(h)
    w.write_record(k["one_record"]["rec"])
    assert w.records_written == 1
    w.finish()
    assert w.into_inner().hex() == k["one_record"]["hex"]
    w = oracle.Writer(h)
    w.write_batch(oracle.records_array(k["batch3"]["recs"]))
    assert w.records_written == 3
    w.finish()
    out = w.into_inner()
    assert len(out) == k["batch3"]["len"] and out.hex() == k["batch3"]["hex"]
    w = oracle.Writer(None)
    w.write_record(k["headless_one"]["rec"])
    w.finish()
    assert len(w.into_inner()) == k["headless_one"]["len"]


def test_into_inner_does_not_flush(oracle):
    # writer.rs:507-511: ManuallyDrop skips Drop, so buffered records are lost
    w = oracle.Writer(oracle.header_new(16, 12))
    w.write_record([1, 2, 3])
    assert len(w.into_inner()) == 32


def test_writer_flush_boundary(oracle, kat):
    k = kat["writer_buffer"]
    w = oracle.Writer(oracle.header_new(16, 12))
    for i in range(k["records_per_buffer"]):
        w.write_record([i, 0, 0])
    assert len(w.inner()) == 32  # writer.rs:779 "Buffer shouldn't be flushed yet"
    w.write_record([999, 0, 0])
    assert len(w.inner()) == 32 + k["buffer_bytes"]
    assert w.records_written == k["records_per_buffer"] + 1


def test_writer_direct_path(oracle, kat):
    n = kat["writer_buffer"]["direct_batch_records"]
    recs = _recs(oracle, "(i, 2i, 3i)", n)
    w = oracle.Writer(oracle.header_new(16, 12))
    w.write_record([7, 7, 7])  # pending data must be flushed first, writer.rs:327
    w.write_batch(recs)
    assert w.records_written == n + 1
    assert w.sink_writes == 2  # one flush of the pending record, one direct write_all
    assert len(w.inner()) == 32 + 24 * (n + 1)
    # a batch of exactly buffer size is NOT direct (strict >, writer.rs:325)
    w2 = oracle.Writer(None)
    w2.write_batch(recs[:49152])
    assert w2.sink_writes == 1 and len(w2.inner()) == 49152 * 24  # filled -> flushed at :344
    w3 = oracle.Writer(None)
    w3.write_batch(recs[:49151])
    assert w3.sink_writes == 0 and len(w3.inner()) == 0


def test_writer_mixed_and_counter(oracle, kat):
    k = kat["writer_mixed"]
    w = oracle.Writer(oracle.header_new(16, 12))
    w.write_record(k["write_record"])
    w.write_batch(oracle.records_array(k["write_batch"]))
    for i in range(10, 13):  # write_iter = write_record per item, writer.rs:383-391
        w.write_record([i, i * 2, i * 3])
    assert w.records_written == 6
    w.finish()
    data = w.into_inner()
    assert data.hex() == k["hex"]
    got = oracle.Reader(data).collect()
    assert [list(r) for r in got] == k["expect"]
    w = oracle.Writer(oracle.header_new(16, 12))
    w.write_batch(oracle.records_array(np.zeros((0, 3))))
    assert w.records_written == 0  # writer.rs:823-832


def test_writer_ingest(oracle, kat):
    k = kat["writer_ingest"]
    main = oracle.Writer(oracle.header_new(16, 12))
    aux = oracle.Writer(None)
    for r in k["aux_records"]:
        aux.write_record(r)
    main.ingest(aux)
    assert main.records_written == k["main_records_written"]
    assert len(aux.inner()) == k["aux_inner_len_after"]
    main.finish()
    assert [list(r) for r in oracle.Reader(main.into_inner()).collect()] == k["aux_records"]


def test_ingest_non_headless_copies_header(oracle):
    # quirk Q15, writer.rs:466-468: a headered aux writer leaks its 32 header bytes into the stream
    main = oracle.Writer(None)
    aux = oracle.Writer(oracle.header_new(16, 12))
    aux.write_record([1, 2, 3])
    main.ingest(aux)
    main.finish()
    assert len(main.inner()) == 32 + 24
    assert main.records_written == (32 + 24) // 24


# ---- B14-B15 reader ----------------------------------------------------------------------------
def test_reader_stream_100k(oracle, kat):
    k = kat["reader_stream"]
    recs = _recs(oracle, k["formula"], k["n"])
    w = oracle.Writer(oracle.header_new(16, 12))
    w.write_batch(recs)
    w.finish()
    data = w.into_inner()
    for max_read in (0, 4096, 1000):  # short reads must not change the result, reader.rs:224-231
        r = oracle.Reader(data, max_read=max_read)
        caps = []
        while r.read_batch():
            caps.append((r.bytes_read - 32 - sum(caps) * 24) // 24)
        assert caps == k["refills"]
        got = oracle.Reader(data, max_read=max_read).collect()
        assert np.array(got, dtype=np.uint64).tobytes() == np.stack(
            [recs["barcode"], recs["umi"], recs["index"]], axis=1).tobytes()


def test_reader_small(oracle, kat):
    w = oracle.Writer(oracle.header_new(16, 12))
    w.write_record([1, 2, 3])
    w.finish()
    data = w.into_inner()
    r = oracle.Reader(data)
    assert r.bytes_read == 32
    assert r.read_batch() is True and r.read_batch() is False  # reader.rs:639-653
    h = oracle.Reader(data).header()
    assert (h.bc_len, h.umi_len, h.magic, h.version) == (16, 12, 0x21554249, 2)
    empty = oracle.Writer(oracle.header_new(16, 12))
    empty.finish()
    assert oracle.Reader(empty.into_inner()).collect() == []  # reader.rs:595-604
    w = oracle.Writer(oracle.header_new(16, 12))
    w.write_batch(_recs(oracle, "(i, 2i, 3i)", 10))
    w.finish()
    r = oracle.Reader(w.into_inner())
    r.collect()
    assert r.bytes_read == kat["reader_stream"]["bytes_read_small"]["expect"]


def test_reader_truncated(oracle, kat, tmp_path):
    k = kat["truncated"]
    data = bytes.fromhex(k["hex"])
    r = oracle.Reader(data)
    with pytest.raises(oracle.OracleError) as ei:
        r.next()
    assert ei.value.name == k["stream"]["kind"] and ei.value.a == k["stream"]["pos"]
    p = tmp_path / "trunc.ibu"
    p.write_bytes(data)
    with pytest.raises(oracle.OracleError) as ei:
        oracle.load_to_vec(str(p))
    assert ei.value.name == k["load_to_vec"]["kind"]
    with pytest.raises(oracle.OracleError) as ei:
        oracle.Mmap(str(p))
    assert ei.value.name == "InvalidMapSize"  # mmap.rs:155-157


def test_truncation_poisons_whole_last_buffer(oracle):
    # quirk Q8, reader.rs:232-237: complete records of the last refill are not yielded either
    w = oracle.Writer(oracle.header_new(16, 12))
    w.write_batch(_recs(oracle, "(i, 2i, 3i)", 49152 + 10))
    w.finish()
    data = w.into_inner()[:-5]
    r = oracle.Reader(data)
    n = 0
    with pytest.raises(oracle.OracleError) as ei:
        while r.next() is not None:
            n += 1
    assert n == 49152
    assert ei.value.a == 32 + 49152 * 24 + 9 * 24  # bytes_read_before + floor24(read)


def test_reader_bad_header(oracle):
    with pytest.raises(oracle.OracleError) as ei:
        oracle.Reader(b"\0" * 32)
    assert ei.value.name == "InvalidMagicNumber"
    with pytest.raises(oracle.OracleError) as ei:
        oracle.Reader(b"IBU!")
    assert ei.value.name == "Io"  # read_exact -> UnexpectedEof -> IbuError::Io


# ---- load_to_vec / mmap --------------------------------------------------------------------------
def test_load_to_vec(oracle, tmp_path):
    recs = oracle.records_array([[1, 2, 3], [4, 5, 6], [7, 8, 9]])
    h, got = oracle.load_to_vec(_file(oracle, tmp_path, "a.ibu", recs))
    assert (h.bc_len, h.umi_len) == (16, 12) and got.tobytes() == recs.tobytes()
    h, got = oracle.load_to_vec(_file(oracle, tmp_path, "e.ibu", recs[:0]))
    assert len(got) == 0
    with pytest.raises(oracle.OracleError) as ei:
        oracle.load_to_vec(str(tmp_path / "missing.ibu"))
    assert ei.value.name == "Io"


def test_mmap_slices(oracle, kat, tmp_path):
    k = kat["mmap"]
    m = oracle.Mmap(_file(oracle, tmp_path, "s.ibu", _recs(oracle, "(i, 2i, 3i)", k["slice_100"]["n"])))
    assert len(m) == 100
    for c in k["slice_100"]["checks"]:
        s = m.slice(c["s"], c["e"])
        assert len(s) == c["e"] - c["s"]
        assert [int(x) for x in s[0]] == c["first"] and [int(x) for x in s[-1]] == c["last"]
    m1 = oracle.Mmap(_file(oracle, tmp_path, "one.ibu", oracle.records_array([[1, 2, 3]])))
    for c in k["slice_errors_len1"]:
        with pytest.raises(oracle.OracleError) as ei:
            m1.slice(c["s"], c["e"])
        assert (ei.value.name, ei.value.a, ei.value.b) == ("InvalidIndex", c["idx"], c["max"])
    big = k["large"]
    mb = oracle.Mmap(_file(oracle, tmp_path, "l.ibu", _recs(oracle, big["formula"], big["n"])))
    s = mb.slice(big["s"], big["e"])
    assert len(s) == 10 and int(s[0]["index"]) == big["first_index"]


def test_mmap_parallel(oracle, kat, tmp_path):
    k = kat["mmap"]
    m = oracle.Mmap(_file(oracle, tmp_path, "p.ibu", _recs(oracle, "(i, 2i, 3i)", 10000)))
    r = m.process_parallel(k["parallel_10000"]["threads"])
    assert r.count == k["parallel_10000"]["count"]
    assert (r.sum[0] + r.sum[1] + r.sum[2]) % 2**64 == k["parallel_10000"]["sum"]
    assert r.batches == 4
    m = oracle.Mmap(_file(oracle, tmp_path, "a.ibu", _recs(oracle, "(i, 0, 0)", 1000)))
    assert m.process_parallel(0).count == k["parallel_auto_1000"]["count"]
    m = oracle.Mmap(_file(oracle, tmp_path, "e.ibu", oracle.records_array(np.zeros((0, 3)))))
    assert len(m) == 0
    r = m.process_parallel(2)
    assert r.count == 0 and r.batches == 0  # quirk Q6


def test_parallel_split_quirks(oracle, tmp_path):
    # mmap.rs:297-307: per = len / n, the remainder goes to the LAST shard only
    assert [oracle.shard_range(10, 4, i) for i in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 10)]
    # quirk Q5: len < n  ->  per = 0, all but the last shard empty
    assert [oracle.shard_range(3, 8, i) for i in range(8)] == [(0, 0)] * 7 + [(0, 3)]
    assert oracle.shard_range(0, 2, 1) == (0, 0)
    m = oracle.Mmap(_file(oracle, tmp_path, "q.ibu", _recs(oracle, "(i, 2i, 3i)", 3)))
    r = m.process_parallel(8)
    assert r.count == 3 and r.batches == 1
    # threads are clamped to the core count, mmap.rs:295
    assert m.process_parallel(100, cores=2).batches == 2  # n clamped to 2: shards [0,1) and [1,3)
    # batches of 1Mi records: 2.5 Mi records on one thread -> 3 on_batch_complete calls
    n = 2 * 1024 * 1024 + 1234
    mb = oracle.Mmap(_file(oracle, tmp_path, "b.ibu", _recs(oracle, "(i, 2i, 3i)", n)))
    r = mb.process_parallel(1)
    assert r.count == n and r.batches == 3
    with pytest.raises(oracle.OracleError) as ei:
        mb.process_parallel_fail(4, fail_index=3 * (n - 1))  # index column is 3i
    assert ei.value.name == "Process"


def test_roundtrip_example_1e6(oracle, kat, tmp_path):
    """BASELINE config 1 (examples/roundtrip.rs) at N = 1e6, CPU plumbing only."""
    k = kat["roundtrip_1e6"]
    recs = _recs(oracle, k["formula"], k["n"])
    p = str(tmp_path / "rt.ibu")
    h = oracle.header_from_bytes(bytes.fromhex(k["header"]))
    w = oracle.Writer(h, path=p)
    w.write_batch(recs[:1000])
    for r in recs[1000:1010]:
        w.write_record([int(r["barcode"]), int(r["umi"]), int(r["index"])])
    w.write_batch(recs[1010:])
    w.finish()
    w.drop()
    assert os.path.getsize(p) == k["file_len"]
    hh, got = oracle.load_to_vec(p)
    assert oracle.lib.orc_header_sorted(hh) and got.tobytes() == recs.tobytes()
    red = oracle.reduce_records(got)
    assert red["count"] == k["n"] and red["sum"] == k["sums"] and red["xor"] == k["xors"]
    r = oracle.Reader(path=p)
    x = 0
    cnt = 0
    while r.read_batch():
        cnt += 1
    assert cnt == 21  # ceil(1e6 / 49152) refills


# ---- codec (parity UNPINNED: only the table and the cap are stated by the reference) ---------------
def test_codec_examples(oracle, kat):
    c = kat["codec"]
    for ex in c["examples"]:
        code = oracle.pack_2bit(ex["seq"].encode())
        assert code == ex["code"], ex
        assert oracle.unpack_2bit(code, len(ex["seq"])) == ex["seq"].upper().encode()
    for base, v in c["table"].items():  # record.rs:22-25
        assert oracle.pack_2bit(base.encode()) == v
    for bad in c["invalid"]:
        with pytest.raises(oracle.OracleError) as ei:
            oracle.pack_2bit(bad.encode())
        assert ei.value.name == "InvalidBase"
    for ln in (0, 33):
        with pytest.raises(oracle.OracleError):
            oracle.unpack_2bit(0, ln)
    with pytest.raises(oracle.OracleError):
        oracle.pack_2bit(b"A" * 33)
    # bits >= 2*len are ignored on unpack (records are never range-checked: examples/random.rs:44-47)
    assert oracle.unpack_2bit(2**64 - 1, 3) == b"TTT"
    # every vector says where it comes from; only the order-independent ones follow from the reference alone
    assert all(ex.get("prov") in ("table", "recalled", "derived-lsb") for ex in c["examples"])
    assert c["parity"].startswith("UNPINNED")


def test_codec_examples_msb_first(oracle, kat):
    """The hedge (IBU_BASE_ORDER_MSB_FIRST / ORC_ORDER_MSB_FIRST): first base most significant."""
    c = kat["codec"]
    for ex in c["examples_msb_first"]:
        code = oracle.pack_2bit(ex["seq"].encode(), oracle.MSB_FIRST)
        assert code == ex["code"], ex
        assert oracle.unpack_2bit(code, len(ex["seq"]), oracle.MSB_FIRST) == ex["seq"].upper().encode()
    # vectors that follow from the code table alone are the same under both orders
    for a, b in zip(c["examples"], c["examples_msb_first"]):
        if a["prov"] == "table":
            assert a == b
    d = c["deciding_vector"]
    assert oracle.pack_2bit(b"ACGT") == d["lsb_first"] and oracle.pack_2bit(b"ACGT", oracle.MSB_FIRST) == d["msb_first"]
    assert oracle.unpack_2bit((2**64 - 1) << 6, 3, oracle.MSB_FIRST) == b"AAA"  # bits >= 2*len ignored here too
    for bad in c["invalid"]:
        with pytest.raises(oracle.OracleError):
            oracle.pack_2bit(bad.encode(), oracle.MSB_FIRST)


@pytest.mark.parametrize("order", [0, 1])
def test_codec_against_numpy(oracle, order):
    """Independent numpy statement of both bit orders, all lengths 1..32."""
    rng = np.random.default_rng(7)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    for ln in range(1, 33):
        codes = rng.integers(0, 2**64, size=257, dtype=np.uint64)
        pos = np.arange(ln, dtype=np.uint64) if order == 0 else np.arange(ln - 1, -1, -1).astype(np.uint64)
        shifts = (2 * pos)[None, :]  # bit position of base i
        want = lut[((codes[:, None] >> shifts) & np.uint64(3)).astype(np.int64)].reshape(-1)
        got = oracle.unpack_column(codes, ln, order)
        assert got.tobytes() == want.tobytes()
        back, fb, nb = oracle.pack_column(got, codes.size, ln, order)
        mask = np.uint64(2**64 - 1) if ln == 32 else np.uint64((1 << (2 * ln)) - 1)
        assert nb == 0 and fb is None and (back == (codes & mask)).all()
        low, _, nb = oracle.pack_column(np.frombuffer(got.tobytes().lower(), dtype=np.uint8), codes.size, ln, order)
        assert nb == 0 and (low == back).all()
        # the two orders of one sequence are each other's pair-reversal
        other, _, _ = oracle.pack_column(got, codes.size, ln, 1 - order)
        rev = np.zeros_like(back)
        for i in range(ln):
            rev |= ((back >> np.uint64(2 * i)) & np.uint64(3)) << np.uint64(2 * (ln - 1 - i))
        assert (other == rev).all()


def test_codec_invalid_rows(oracle):
    a = np.frombuffer(b"ACGT" * 8, dtype=np.uint8).copy()  # 8 rows of len 4
    a[4 * 3 + 1] = ord("N")
    a[4 * 6 + 0] = 0
    codes, fb, nb = oracle.pack_column(a, 8, 4)
    assert (fb, nb) == (3, 2) and codes[3] == 0 and codes[6] == 0 and codes[0] == 0b11100100
    recs, fb, nb = oracle.encode_records(a, a, None, 8, 4, 4, first_index=100)
    assert (fb, nb) == (3, 2) and int(recs[7]["index"]) == 107


def test_generator_and_columns(oracle):
    recs = oracle.generate(0x1B00001, 5, 1000, 16, 12)
    assert int(recs["index"][0]) == 5 and int(recs["barcode"].max()) < 2**32 and int(recs["umi"].max()) < 2**24
    # shard independence: generating [5,1005) in two pieces gives the same records
    two = np.concatenate([oracle.generate(0x1B00001, 5, 400, 16, 12), oracle.generate(0x1B00001, 405, 600, 16, 12)])
    assert two.tobytes() == recs.tobytes()
    # splitmix64 reference value (Vigna's test vector: seed 0 -> first output)
    assert oracle.lib.orc_splitmix64(0) == 0xE220A8397B1DCDAF
    bc, umi, idx = oracle.deserialize(recs)
    assert (bc == recs["barcode"]).all() and (idx == recs["index"]).all()
    assert oracle.serialize(bc, umi, idx).tobytes() == recs.tobytes()
    flat = np.frombuffer(recs.tobytes(), dtype="<u8").reshape(-1, 3)  # numpy view = bytemuck::cast_slice
    assert (flat[:, 0] == bc).all() and (flat[:, 1] == umi).all()
    b, u, i = oracle.decode_records(recs, 16, 12)
    back, fb, nb = oracle.encode_records(b, u, i, 1000, 16, 12)
    assert nb == 0 and back.tobytes() == recs.tobytes()
    full = oracle.generate(3, 0, 100, 32, 32)
    b, u, i = oracle.decode_records(full, 32, 32)
    back, _, nb = oracle.encode_records(b, u, i, 100, 32, 32)
    assert nb == 0 and back.tobytes() == full.tobytes()


def test_barcode_counts_match_hashmap_semantics(oracle):
    """parallel.rs:72-98 keeps HashMap<barcode, count>; np.unique is an independent statement of that map."""
    rng = np.random.default_rng(3)
    n = 20_000
    recs = oracle.generate(0x1B00005, 0, n, 16, 12)
    recs["barcode"] = rng.integers(0, 300, n, dtype=np.uint64)
    recs["umi"] = rng.integers(0, 40, n, dtype=np.uint64)
    srt = oracle.sort_records(recs)
    b, c, u = oracle.barcode_counts(srt)
    keys, counts = np.unique(recs["barcode"], return_counts=True)
    assert b.tolist() == keys.tolist() and c.tolist() == counts.tolist()
    pairs = np.unique(np.stack([recs["barcode"], recs["umi"]], axis=1), axis=0)
    pk, pu = np.unique(pairs[:, 0], return_counts=True)
    assert pk.tolist() == keys.tolist() and u.tolist() == pu.tolist()
    assert int(c.sum()) == n
    e = oracle.barcode_counts(recs[:0])
    assert all(len(x) == 0 for x in e)
