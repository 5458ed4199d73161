"""Property tests (hypothesis): the host Writer / Reader of the product library against the CPU oracle over random
operation sequences — the buffered/direct write rule (writer.rs:321-351), the flush points (:220-226, :260-273), ingest
(:477-482), refill with short reads (reader.rs:224-231), truncation (:232-237, quirk Q8) and the error state after
it (Q9).  The reference has no property tests; these widen its fixed cases around the same contracts."""
import io

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import ibu_amd as ia

BUF_RECORDS = 49_152  # DEFAULT_BUFFER_SIZE / 24


def _recs(n, salt):
    a = np.empty(n, dtype=ia.REC_DTYPE)
    i = np.arange(n, dtype=np.uint64)
    a["barcode"], a["umi"], a["index"] = i * np.uint64(3) + np.uint64(salt), i ^ np.uint64(salt * 7919), i
    return a


# sizes cluster around the interesting boundaries: 0, 1, the 49 152-record buffer and batches larger than it
sizes = st.one_of(st.integers(0, 3), st.integers(BUF_RECORDS - 2, BUF_RECORDS + 2), st.integers(0, 2 * BUF_RECORDS + 5),
                  st.integers(1, 300))
ops = st.lists(st.one_of(st.tuples(st.just("record"), st.integers(1, 40)), st.tuples(st.just("batch"), sizes),
                         st.tuples(st.just("ingest"), st.integers(0, 300)), st.tuples(st.just("finish"), st.just(0))),
               min_size=1, max_size=12)


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(ops=ops, headless=st.booleans())
def test_writer_matches_oracle_after_every_operation(oracle, ops, headless):
    hp = None if headless else ia.Header(16, 12)
    ho = None if headless else oracle.header_new(16, 12)
    w = ia.Writer.new_headless() if headless else ia.Writer.new(None, hp)
    o = oracle.Writer(header=ho)
    salt = 1
    for kind, n in ops:
        salt += 1
        if kind == "record":
            for r in _recs(n, salt):
                w.write_record(ia.Record(int(r["barcode"]), int(r["umi"]), int(r["index"])))
                o.write_record((int(r["barcode"]), int(r["umi"]), int(r["index"])))
        elif kind == "batch":
            recs = _recs(n, salt)
            w.write_batch(recs)
            o.write_batch(recs)
        elif kind == "ingest":
            recs = _recs(n, salt)
            wa, oa = ia.Writer.new_headless(), oracle.Writer(header=None)
            wa.write_batch(recs)
            oa.write_batch(recs)
            w.ingest(wa)
            o.ingest(oa)
            assert wa.inner_bytes() == oa.inner() == b""  # the auxiliary sink is drained
        else:
            w.finish()
            o.finish()
        # what has reached the sink so far, and the counter, agree after EVERY step (not only at the end)
        assert w.records_written() == o.records_written
        assert w.inner_bytes() == o.inner(), (kind, n)
    w.finish()
    o.finish()
    assert w.inner_bytes() == o.inner()


class _Dribble(io.RawIOBase):
    """A source whose reads return at most `chunk(pos)` bytes: exercises the refill loop's short-read handling."""

    def __init__(self, data, chunks):
        self.b, self.p, self.chunks, self.k = data, 0, chunks, 0

    def read(self, n=-1):
        c = self.chunks[self.k % len(self.chunks)]
        self.k += 1
        out = self.b[self.p:self.p + min(n, c)]
        self.p += len(out)
        return out


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(n=st.one_of(st.integers(0, 5), st.integers(BUF_RECORDS - 1, BUF_RECORDS + 1), st.integers(0, 120_000)),
       cut=st.integers(0, 23), chunks=st.lists(st.integers(1, 700_000), min_size=1, max_size=5))
def test_reader_matches_oracle_with_short_reads_and_truncation(oracle, n, cut, chunks):
    recs = _recs(n, 5)
    w = ia.Writer.new(None, ia.Header(16, 12))
    w.write_batch(recs)
    w.finish()
    data = w.inner_bytes()
    if cut and n:
        data = data[:-cut]  # a trailing partial record
    r = ia.Reader.new(_Dribble(data, chunks))
    o = oracle.Reader(data=data, max_read=0)  # the oracle reads whole refills; results must not depend on read sizes
    got, want = [], []
    err_p = err_o = None
    try:
        for rec in r:
            got.append(tuple(rec))
    except ia.IbuError as e:
        err_p = (e.kind, e.pos)
    try:
        want = o.collect()
    except oracle.OracleError as e:
        err_o = e
    if cut and n:
        # Q8: the whole final refill is poisoned — every record before that refill was yielded, none of it after
        assert err_p is not None and err_p[0] == "TruncatedRecord" and err_o is not None
        full_refills = ((n * 24 - cut) // (BUF_RECORDS * 24)) * BUF_RECORDS
        assert len(got) == full_refills
        assert err_p[1] == 32 + 24 * ((n * 24 - cut) // 24)  # pos = bytes read before + floor24(read) (reader.rs:232-237)
    else:
        assert err_p is None and err_o is None
        assert got == want and len(got) == n
        assert r.bytes_read == len(data)
    assert got == [tuple(int(v) for v in x) for x in recs[:len(got)]]


# ---- damaged compressed input: every decoder fails cleanly (an IbuError) or yields a prefix — never hangs or crashes ----
def _zstd(data):
    import ctypes as C
    z = C.CDLL("libzstd.so.1")
    z.ZSTD_compressBound.restype = z.ZSTD_compress.restype = C.c_size_t
    z.ZSTD_compressBound.argtypes = [C.c_size_t]
    z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
    buf = C.create_string_buffer(z.ZSTD_compressBound(len(data)))
    k = z.ZSTD_compress(buf, len(buf), data, len(data), 1)
    return buf.raw[:k]


def _compressed_blobs():
    import bz2
    import gzip
    import lzma

    from tests.bgzf import bgzf_compress

    recs = _recs(20_000, 3)
    w = ia.Writer.new(None, ia.Header(16, 12))
    w.write_batch(recs)
    w.finish()
    raw = w.inner_bytes()
    return raw, {"gz": gzip.compress(raw, 1), "bgzf": bgzf_compress(raw, block=8000), "bz2": bz2.compress(raw, 1),
                 "xz": lzma.compress(raw, preset=0), "zst": _zstd(raw)}


_RAW, _BLOBS = None, None


@settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(fmt=st.sampled_from(["gz", "bgzf", "bz2", "xz", "zst"]), damage=st.sampled_from(["flip", "cut", "zero_run", "dup_tail"]),
       where=st.floats(0.0, 1.0), byte=st.integers(1, 255))
def test_damaged_compressed_input_fails_cleanly(tmp_path, fmt, damage, where, byte):
    global _RAW, _BLOBS
    if _BLOBS is None:
        _RAW, _BLOBS = _compressed_blobs()
    blob = bytearray(_BLOBS[fmt])
    pos = min(len(blob) - 1, max(6, int(where * len(blob))))  # keep the magic so the format is still sniffed
    if damage == "flip":
        blob[pos] ^= byte
    elif damage == "cut":
        del blob[pos:]
    elif damage == "zero_run":
        blob[pos:pos + 64] = bytes(min(64, len(blob) - pos))
    else:
        blob += blob[-pos // 4:]  # garbage after the end of the stream
    p = tmp_path / f"damaged.{fmt}"
    p.write_bytes(bytes(blob))
    want = np.frombuffer(_RAW[32:], dtype=ia.REC_DTYPE)
    try:
        r = ia.Reader.from_path(p)
        got = list(r)
    except ia.IbuError as e:
        assert e.kind in ("Niffler", "TruncatedRecord", "Io", "InvalidMagicNumber", "InvalidVersion",
                          "InvalidBarcodeLength", "InvalidUmiLength"), e.kind
        return
    # no error: then what came out is the original stream, possibly followed by records decoded from appended bytes
    assert len(got) >= 0
    k = min(len(got), len(want))
    if damage in ("cut",):
        assert [tuple(x) for x in got[:k]] == [tuple(int(v) for v in x) for x in want[:k]]
