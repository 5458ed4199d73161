"""ctypes + numpy face of the CPU oracle (oracle/ibu_oracle.c).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; nothing under
ibu_amd/ may import this module.  Parity status: see oracle/ibu_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libibu_oracle.so")

REC_DTYPE = np.dtype([("barcode", "<u8"), ("umi", "<u8"), ("index", "<u8")])
assert REC_DTYPE.itemsize == 24

KIND_NAMES = {
    0: "Ok", 1: "Io", 2: "Niffler", 3: "InvalidMagicNumber", 4: "TruncatedRecord", 5: "InvalidVersion",
    6: "InvalidBarcodeLength", 7: "InvalidUmiLength", 8: "InvalidMapSize", 9: "InvalidIndex",
    10: "Process", 11: "InvalidBase", 12: "SeqLen",
}


class Header(C.Structure):
    _fields_ = [("magic", C.c_uint32), ("version", C.c_uint32), ("bc_len", C.c_uint32),
                ("umi_len", C.c_uint32), ("flags", C.c_uint64), ("reserved", C.c_uint8 * 8)]


class Record(C.Structure):
    _fields_ = [("barcode", C.c_uint64), ("umi", C.c_uint64), ("index", C.c_uint64)]


class Err(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_uint64), ("b", C.c_uint64)]


class Reduce(C.Structure):
    _fields_ = [("count", C.c_uint64), ("sum", C.c_uint64 * 3), ("xor_", C.c_uint64 * 3),
                ("batches", C.c_uint64)]


class OracleError(Exception):
    def __init__(self, kind, a=0, b=0):
        self.kind, self.a, self.b = kind, a, b
        self.name = KIND_NAMES.get(kind, str(kind))
        super().__init__(f"{self.name}(a={a}, b={b})")


def build():
    """(Re)build libibu_oracle.so with gcc — building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "libibu_oracle.so"])


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    vp, sz, u64, u32, i32 = C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_int
    P = C.POINTER
    sig = {
        "orc_header_new": (None, [P(Header), u32, u32]),
        "orc_header_set_sorted": (None, [P(Header)]),
        "orc_header_sorted": (i32, [P(Header)]),
        "orc_header_validate": (i32, [P(Header), P(Err)]),
        "orc_record_cmp": (i32, [P(Record), P(Record)]),
        "orc_writer_new_mem": (vp, [P(Header)]),
        "orc_writer_new_file": (vp, [C.c_char_p, P(Header)]),
        "orc_writer_write_record": (i32, [vp, P(Record)]),
        "orc_writer_write_batch": (i32, [vp, vp, sz]),
        "orc_writer_ingest": (i32, [vp, vp]),
        "orc_writer_finish": (i32, [vp]),
        "orc_writer_records_written": (u64, [vp]),
        "orc_writer_inner": (vp, [vp, P(sz)]),
        "orc_writer_sink_writes": (u64, [vp]),
        "orc_writer_drop": (None, [vp]),
        "orc_writer_forget": (None, [vp]),
        "orc_reader_new_mem": (i32, [vp, sz, sz, P(vp), P(Err)]),
        "orc_reader_new_file": (i32, [C.c_char_p, P(vp), P(Err)]),
        "orc_reader_header": (None, [vp, P(Header)]),
        "orc_reader_read_batch": (i32, [vp, P(i32), P(Err)]),
        "orc_reader_next": (i32, [vp, P(Record), P(i32), P(Err)]),
        "orc_reader_bytes_read": (u64, [vp]),
        "orc_reader_free": (None, [vp]),
        "orc_load_to_vec": (i32, [C.c_char_p, P(Header), P(vp), P(sz), P(Err)]),
        "orc_free": (None, [vp]),
        "orc_mmap_new": (i32, [C.c_char_p, P(vp), P(Err)]),
        "orc_mmap_len": (sz, [vp]),
        "orc_mmap_header": (None, [vp, P(Header)]),
        "orc_mmap_slice": (i32, [vp, sz, sz, P(vp), P(sz), P(Err)]),
        "orc_mmap_free": (None, [vp]),
        "orc_shard_range": (None, [sz, sz, sz, P(sz), P(sz)]),
        "orc_mmap_process_parallel": (i32, [vp, sz, sz, P(Reduce), P(Err)]),
        "orc_mmap_process_parallel_fail": (i32, [vp, sz, sz, u64, P(Err)]),
        "orc_reduce_records": (None, [vp, sz, P(Reduce)]),
        "orc_deserialize": (None, [vp, sz, vp, vp, vp]),
        "orc_serialize": (None, [vp, vp, vp, sz, vp]),
        "orc_pack_2bit": (i32, [vp, u32, P(u64)]),
        "orc_unpack_2bit": (i32, [u64, u32, vp]),
        "orc_unpack_column": (i32, [vp, sz, u32, vp]),
        "orc_pack_column": (i32, [vp, sz, u32, vp, P(u64), P(u64)]),
        "orc_decode_records": (i32, [vp, sz, u32, u32, vp, vp, vp]),
        "orc_encode_records": (i32, [vp, vp, vp, u64, sz, u32, u32, vp, P(u64), P(u64)]),
        "orc_pack_2bit_order": (i32, [vp, u32, i32, P(u64)]),
        "orc_unpack_2bit_order": (i32, [u64, u32, i32, vp]),
        "orc_unpack_column_order": (i32, [vp, sz, u32, i32, vp]),
        "orc_pack_column_order": (i32, [vp, sz, u32, i32, vp, P(u64), P(u64)]),
        "orc_decode_records_order": (i32, [vp, sz, u32, u32, i32, vp, vp, vp]),
        "orc_encode_records_order": (i32, [vp, vp, vp, u64, sz, u32, u32, i32, vp, P(u64), P(u64)]),
        "orc_splitmix64": (u64, [u64]),
        "orc_generate": (None, [u64, u64, sz, u32, u32, vp]),
        "orc_sort_records": (None, [vp, sz]),
        "orc_is_sorted": (i32, [vp, sz]),
        "orc_lower_bound": (sz, [vp, sz, vp]),
        "orc_bench_decode_encode": (C.c_double, [sz, u32, u32, u64, i32, i32, P(u64)]),
        "orc_bench_reduce": (C.c_double, [sz, u64, i32, P(Reduce)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _check(rc, err=None):
    if rc:
        raise OracleError(rc, err.a if err is not None else 0, err.b if err is not None else 0)


# ---- header -----------------------------------------------------------------------------
def header_new(bc_len, umi_len):
    h = Header()
    lib.orc_header_new(C.byref(h), bc_len, umi_len)
    return h


def header_from_bytes(b):
    assert len(b) == 32
    return Header.from_buffer_copy(bytes(b))


def header_validate(h):
    e = Err()
    _check(lib.orc_header_validate(C.byref(h), C.byref(e)), e)


def record_cmp(a, b):
    ra, rb = Record(*[int(x) for x in a]), Record(*[int(x) for x in b])
    return lib.orc_record_cmp(C.byref(ra), C.byref(rb))


def records_array(rows):
    """list of (bc, umi, idx) or (n,3) uint64 array -> structured AoS array."""
    a = np.asarray(rows, dtype=np.uint64).reshape(-1, 3)
    out = np.empty(a.shape[0], dtype=REC_DTYPE)
    out["barcode"], out["umi"], out["index"] = a[:, 0], a[:, 1], a[:, 2]
    return out


# ---- writer -----------------------------------------------------------------------------
class Writer:
    def __init__(self, header=None, path=None):
        hp = C.byref(header) if header is not None else None
        self._w = lib.orc_writer_new_file(path.encode(), hp) if path else lib.orc_writer_new_mem(hp)
        if not self._w:
            raise OracleError(1)

    def write_record(self, rec):
        r = Record(*[int(x) for x in rec])
        _check(lib.orc_writer_write_record(self._w, C.byref(r)))

    def write_batch(self, recs):
        recs = np.ascontiguousarray(recs, dtype=REC_DTYPE)
        _check(lib.orc_writer_write_batch(self._w, _ptr(recs), recs.shape[0]))

    def ingest(self, other):
        _check(lib.orc_writer_ingest(self._w, other._w))

    def finish(self):
        _check(lib.orc_writer_finish(self._w))

    @property
    def records_written(self):
        return lib.orc_writer_records_written(self._w)

    @property
    def sink_writes(self):
        return lib.orc_writer_sink_writes(self._w)

    def inner(self):
        n = C.c_size_t()
        p = lib.orc_writer_inner(self._w, C.byref(n))
        return C.string_at(p, n.value) if n.value else b""

    def drop(self):
        if self._w:
            lib.orc_writer_drop(self._w)
            self._w = None

    def into_inner(self):
        b = self.inner()
        lib.orc_writer_forget(self._w)
        self._w = None
        return b

    def __del__(self):
        try:
            self.drop()
        except Exception:
            pass


# ---- reader -----------------------------------------------------------------------------
class Reader:
    def __init__(self, data=None, path=None, max_read=0):
        e, out = Err(), C.c_void_p()
        if path is not None:
            rc = lib.orc_reader_new_file(path.encode(), C.byref(out), C.byref(e))
        else:
            self._keep = np.frombuffer(bytes(data), dtype=np.uint8)
            rc = lib.orc_reader_new_mem(_ptr(self._keep), self._keep.size, max_read, C.byref(out), C.byref(e))
        _check(rc, e)
        self._r = out

    def header(self):
        h = Header()
        lib.orc_reader_header(self._r, C.byref(h))
        return h

    def read_batch(self):
        e, has = Err(), C.c_int()
        _check(lib.orc_reader_read_batch(self._r, C.byref(has), C.byref(e)), e)
        return bool(has.value)

    def next(self):
        e, got, r = Err(), C.c_int(), Record()
        _check(lib.orc_reader_next(self._r, C.byref(r), C.byref(got), C.byref(e)), e)
        return (r.barcode, r.umi, r.index) if got.value else None

    def collect(self):
        out = []
        while True:
            r = self.next()
            if r is None:
                return out
            out.append(r)

    @property
    def bytes_read(self):
        return lib.orc_reader_bytes_read(self._r)

    def __del__(self):
        if getattr(self, "_r", None):
            lib.orc_reader_free(self._r)
            self._r = None


def load_to_vec(path):
    e, h, p, n = Err(), Header(), C.c_void_p(), C.c_size_t()
    _check(lib.orc_load_to_vec(path.encode(), C.byref(h), C.byref(p), C.byref(n), C.byref(e)), e)
    # (c_char * nbytes).from_address: string_at takes a C int and fails beyond 2 GiB
    recs = np.frombuffer((C.c_char * (n.value * 24)).from_address(p.value), dtype=REC_DTYPE).copy() if n.value \
        else np.empty(0, dtype=REC_DTYPE)
    lib.orc_free(p)
    return h, recs


class Mmap:
    def __init__(self, path):
        e, out = Err(), C.c_void_p()
        _check(lib.orc_mmap_new(path.encode(), C.byref(out), C.byref(e)), e)
        self._m = out

    def __len__(self):
        return lib.orc_mmap_len(self._m)

    def header(self):
        h = Header()
        lib.orc_mmap_header(self._m, C.byref(h))
        return h

    def slice(self, start, end):
        e, p, n = Err(), C.c_void_p(), C.c_size_t()
        _check(lib.orc_mmap_slice(self._m, start, end, C.byref(p), C.byref(n), C.byref(e)), e)
        return np.frombuffer(C.string_at(p, n.value * 24), dtype=REC_DTYPE).copy()

    def process_parallel(self, num_threads, cores=8):
        e, r = Err(), Reduce()
        _check(lib.orc_mmap_process_parallel(self._m, num_threads, cores, C.byref(r), C.byref(e)), e)
        return r

    def process_parallel_fail(self, num_threads, fail_index, cores=8):
        e = Err()
        _check(lib.orc_mmap_process_parallel_fail(self._m, num_threads, cores, fail_index, C.byref(e)), e)

    def __del__(self):
        if getattr(self, "_m", None):
            lib.orc_mmap_free(self._m)
            self._m = None


def shard_range(length, n, i):
    s, e = C.c_size_t(), C.c_size_t()
    lib.orc_shard_range(length, n, i, C.byref(s), C.byref(e))
    return s.value, e.value


# ---- flat checkers ------------------------------------------------------------------------
def reduce_records(recs):
    recs = np.ascontiguousarray(recs, dtype=REC_DTYPE)
    r = Reduce()
    lib.orc_reduce_records(_ptr(recs), recs.shape[0], C.byref(r))
    return {"count": r.count, "sum": list(r.sum), "xor": list(r.xor_)}


def deserialize(recs):
    recs = np.ascontiguousarray(recs, dtype=REC_DTYPE)
    n = recs.shape[0]
    bc, umi, idx = (np.empty(n, dtype=np.uint64) for _ in range(3))
    lib.orc_deserialize(_ptr(recs), n, _ptr(bc), _ptr(umi), _ptr(idx))
    return bc, umi, idx


def serialize(bc, umi, idx):
    bc, umi, idx = (np.ascontiguousarray(x, dtype=np.uint64) for x in (bc, umi, idx))
    out = np.empty(bc.shape[0], dtype=REC_DTYPE)
    lib.orc_serialize(_ptr(bc), _ptr(umi), _ptr(idx), bc.shape[0], _ptr(out))
    return out


LSB_FIRST, MSB_FIRST = 0, 1  # bit order of the 2-bit codec (ibu_oracle.h ORC_ORDER_*); LSB_FIRST is the default everywhere


def pack_2bit(seq, order=LSB_FIRST):
    b = np.frombuffer(bytes(seq), dtype=np.uint8)
    out = C.c_uint64()
    _check(lib.orc_pack_2bit_order(_ptr(b) if b.size else None, b.size, order, C.byref(out)))
    return out.value


def unpack_2bit(code, length, order=LSB_FIRST):
    out = np.empty(max(length, 1), dtype=np.uint8)
    _check(lib.orc_unpack_2bit_order(code, length, order, _ptr(out)))
    return out[:length].tobytes()


def unpack_column(codes, length, order=LSB_FIRST):
    codes = np.ascontiguousarray(codes, dtype=np.uint64)
    out = np.empty(codes.shape[0] * length, dtype=np.uint8)
    _check(lib.orc_unpack_column_order(_ptr(codes), codes.shape[0], length, order, _ptr(out)))
    return out


def pack_column(ascii_, n, length, order=LSB_FIRST):
    """-> (codes, first_bad, n_bad); offending rows come back as 0 (no exception)."""
    a = np.ascontiguousarray(ascii_, dtype=np.uint8)
    out = np.empty(n, dtype=np.uint64)
    fb, nb = C.c_uint64(), C.c_uint64()
    rc = lib.orc_pack_column_order(_ptr(a), n, length, order, _ptr(out), C.byref(fb), C.byref(nb))
    if rc not in (0, 11):
        _check(rc)
    return out, (None if nb.value == 0 else fb.value), nb.value


def decode_records(recs, bc_len, umi_len, order=LSB_FIRST):
    recs = np.ascontiguousarray(recs, dtype=REC_DTYPE)
    n = recs.shape[0]
    bc = np.empty(n * bc_len, dtype=np.uint8)
    umi = np.empty(n * umi_len, dtype=np.uint8)
    idx = np.empty(n, dtype=np.uint64)
    _check(lib.orc_decode_records_order(_ptr(recs), n, bc_len, umi_len, order, _ptr(bc), _ptr(umi), _ptr(idx)))
    return bc, umi, idx


def encode_records(bc, umi, idx, n, bc_len, umi_len, first_index=0, order=LSB_FIRST):
    """-> (records, first_bad, n_bad)."""
    bc = np.ascontiguousarray(bc, dtype=np.uint8)
    umi = np.ascontiguousarray(umi, dtype=np.uint8)
    ip = None
    if idx is not None:
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        ip = _ptr(idx)
    out = np.empty(n, dtype=REC_DTYPE)
    fb, nb = C.c_uint64(), C.c_uint64()
    rc = lib.orc_encode_records_order(_ptr(bc), _ptr(umi), ip, first_index, n, bc_len, umi_len, order, _ptr(out),
                                      C.byref(fb), C.byref(nb))
    if rc not in (0, 11):
        _check(rc)
    return out, (None if nb.value == 0 else fb.value), nb.value


def generate(seed, first, n, bc_len, umi_len):
    out = np.empty(n, dtype=REC_DTYPE)
    lib.orc_generate(seed, first, n, bc_len, umi_len, _ptr(out))
    return out


def sort_records(recs):
    out = np.ascontiguousarray(recs, dtype=REC_DTYPE).copy()
    lib.orc_sort_records(_ptr(out), out.shape[0])
    return out


def lower_bound(sorted_recs, key):
    """First index whose record is >= key (a 1-element REC_DTYPE array or a (barcode, umi, index) tuple)."""
    r = np.ascontiguousarray(sorted_recs, dtype=REC_DTYPE)
    k = np.zeros(1, dtype=REC_DTYPE)
    k[0] = tuple(int(v) for v in key) if isinstance(key, tuple) else key
    return int(lib.orc_lower_bound(_ptr(r), r.shape[0], _ptr(k)))


def barcode_counts(sorted_recs):
    """BarcodeAnalyzer (parallel.rs:72-98) over sorted records -> (barcodes, counts, unique_umis), ascending barcode."""
    r = np.ascontiguousarray(sorted_recs, dtype=REC_DTYPE)
    n = r.shape[0]
    b, c, u = (np.empty(max(n, 1), dtype=np.uint64) for _ in range(3))
    lib.orc_barcode_counts.restype = C.c_size_t
    lib.orc_barcode_counts.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    k = lib.orc_barcode_counts(_ptr(r), n, _ptr(b), _ptr(c), _ptr(u))
    return b[:k].copy(), c[:k].copy(), u[:k].copy()


def is_sorted(recs):
    recs = np.ascontiguousarray(recs, dtype=REC_DTYPE)
    return bool(lib.orc_is_sorted(_ptr(recs), recs.shape[0]))


def bench_decode_encode(n, bc_len, umi_len, seed, threads, reps=1):
    """Seconds for `reps` decode+encode passes over n records on `threads` threads, and the round-trip checksum."""
    c = C.c_uint64()
    t = lib.orc_bench_decode_encode(n, bc_len, umi_len, seed, threads, reps, C.byref(c))
    return t, c.value


def bench_reduce(n, seed, threads):
    r = Reduce()
    t = lib.orc_bench_reduce(n, seed, threads, C.byref(r))
    return t, {"count": r.count, "sum": list(r.sum), "xor": list(r.xor_)}


def bench_phases(path, n, threads):
    """The phases of the reference's example programs on a file of n records (ibu_oracle.h: orc_bench_phases) ->
    ({"write_record": s, "reader_xor": s, "load_to_vec": s, "process_parallel_1": s, "process_parallel_T": s}, xor, sums)."""
    lib.orc_bench_phases.restype = C.c_int
    lib.orc_bench_phases.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                                     C.POINTER(C.c_uint64), C.POINTER(Err)]
    sec, x, sums, e = (C.c_double * 5)(), C.c_uint64(), (C.c_uint64 * 3)(), Err()
    _check(lib.orc_bench_phases(str(path).encode(), n, threads, sec, C.byref(x), sums, C.byref(e)), e)
    names = ["write_record", "reader_xor", "load_to_vec", "process_parallel_1", "process_parallel_T"]
    return dict(zip(names, (float(v) for v in sec))), x.value, [int(v) for v in sums]
