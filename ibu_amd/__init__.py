"""ibu_amd — MI355X-native batch record-stream + 2-bit codec path for the IBU format.

A thin Python mirror of the reference crate's public API (src/lib.rs:178-181:
Header, Record, HEADER_SIZE, MAGIC, RECORD_SIZE, VERSION, IbuError, load_to_vec, MmapReader,
Reader, Writer, ParallelProcessor, ParallelReader) over the C ABI of include/ibu_hip.h, plus
the device context that exposes the HIP kernels.  All work happens in libibu_hip.so; this
package binds it and nothing else (no numpy/torch arithmetic stands in for a kernel).
"""
import copy
import ctypes as C
import weakref
from collections import namedtuple

import numpy as np

from . import _lib
from ._lib import CAllocProbe, CDecodeSink, CErrorDetail, CKeyPlan, CHeader, CNumaInfo, CProcessorVTable, CRecord, CReduceResult, CRingConfig, CStreamStats

lib = _lib.load()

MAGIC = 0x21554249  # src/constructs/header.rs:5
VERSION = 2  # header.rs:6
HEADER_SIZE = 32  # header.rs:7
RECORD_SIZE = 24  # record.rs:3
DEFAULT_BUFFER_SIZE = 48 * 1024 * RECORD_SIZE  # reader.rs:14, writer.rs:10
BATCH_SIZE = 1024 * 1024  # mmap.rs:284

PROC_REDUCE, PROC_DECODE = 1, 2

#: numpy view of a `&[Record]` (bytemuck::cast_slice)
REC_DTYPE = np.dtype([("barcode", "<u8"), ("umi", "<u8"), ("index", "<u8")])


# ---- errors (src/error.rs:56-128) ---------------------------------------------------------------
class IbuError(Exception):
    """`kind` is the reference's variant name; payload fields follow the variant."""

    def __init__(self, code, detail):
        self.code = code
        self.kind = lib.ibu_status_name(code).decode()
        self.a, self.b, self.os_errno = detail.a, detail.b, detail.os_errno
        self.expected, self.actual = detail.a, detail.b  # InvalidMagicNumber / InvalidVersion
        self.pos = detail.a  # TruncatedRecord
        self.idx, self.max = detail.a, detail.b  # InvalidIndex
        self.length = detail.a  # InvalidBarcodeLength / InvalidUmiLength
        self.first_bad, self.n_bad = detail.a, detail.b  # InvalidBase
        super().__init__(f"{self.kind}: {detail.message.decode(errors='replace')}")


def _check(rc):
    if rc:
        d = CErrorDetail()
        lib.ibu_last_error(C.byref(d))
        raise IbuError(rc, d)


# ---- Header / Record ---------------------------------------------------------------------------------
class Header:
    """src/constructs/header.rs:44-61 — 32-byte POD."""

    def __init__(self, bc_len, umi_len):  # Header::new :84-93
        self._h = CHeader()
        lib.ibu_header_init(C.byref(self._h), bc_len, umi_len)

    magic = property(lambda s: s._h.magic, lambda s, v: setattr(s._h, "magic", v))
    version = property(lambda s: s._h.version, lambda s, v: setattr(s._h, "version", v))
    bc_len = property(lambda s: s._h.bc_len, lambda s, v: setattr(s._h, "bc_len", v))
    umi_len = property(lambda s: s._h.umi_len, lambda s, v: setattr(s._h, "umi_len", v))
    flags = property(lambda s: s._h.flags, lambda s, v: setattr(s._h, "flags", v))
    reserved = property(lambda s: bytes(s._h.reserved))

    def set_sorted(self):  # :111-113
        lib.ibu_header_set_sorted(C.byref(self._h))

    def sorted(self):  # :130-132
        return bool(lib.ibu_header_sorted(C.byref(self._h)))

    def validate(self):  # :167-187
        _check(lib.ibu_header_validate(C.byref(self._h)))

    def as_bytes(self):  # :203-205
        buf = C.create_string_buffer(HEADER_SIZE)
        _check(lib.ibu_header_as_bytes(C.byref(self._h), buf, HEADER_SIZE))
        return buf.raw

    @classmethod
    def from_bytes(cls, b):  # :226-228 (the reference panics on a wrong length; here InvalidArg)
        h = cls.__new__(cls)
        h._h = CHeader()
        _check(lib.ibu_header_from_bytes(bytes(b), len(b), C.byref(h._h)))
        return h

    @classmethod
    def _wrap(cls, ch):
        h = cls.__new__(cls)
        h._h = ch
        return h

    def __eq__(self, o):
        return isinstance(o, Header) and self.as_bytes() == o.as_bytes()

    def __hash__(self):
        return hash(self.as_bytes())

    def __repr__(self):
        return (f"Header {{ magic: {self.magic:#x}, version: {self.version}, bc_len: {self.bc_len}, "
                f"umi_len: {self.umi_len}, flags: {self.flags} }}")


class Record(namedtuple("Record", ["barcode", "umi", "index"])):
    """src/constructs/record.rs:58-66 — tuple order is the derived lexicographic Ord."""
    __slots__ = ()

    def __new__(cls, barcode=0, umi=0, index=0):
        return super().__new__(cls, int(barcode), int(umi), int(index))

    def _c(self):
        return CRecord(self.barcode, self.umi, self.index)

    def as_bytes(self):  # :108-110
        buf = C.create_string_buffer(RECORD_SIZE)
        r = self._c()
        _check(lib.ibu_record_as_bytes(C.byref(r), buf, RECORD_SIZE))
        return buf.raw

    @classmethod
    def from_bytes(cls, b):  # :130-132
        r = CRecord()
        _check(lib.ibu_record_from_bytes(bytes(b), len(b), C.byref(r)))
        return cls(r.barcode, r.umi, r.index)

    def cmp(self, other):
        a, b = self._c(), Record(*other)._c()
        return lib.ibu_record_cmp(C.byref(a), C.byref(b))


def records_array(rows):
    """iterable of Record / 3-tuples, or an (n,3) integer array -> contiguous AoS array."""
    if isinstance(rows, np.ndarray) and rows.dtype == REC_DTYPE:
        return np.ascontiguousarray(rows)
    a = np.asarray(list(rows) if not isinstance(rows, np.ndarray) else rows, dtype=np.uint64).reshape(-1, 3)
    out = np.empty(a.shape[0], dtype=REC_DTYPE)
    out["barcode"], out["umi"], out["index"] = a[:, 0], a[:, 1], a[:, 2]
    return out


def _hptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- Writer (src/io/writer.rs) ---------------------------------------------------------------------------
class Writer:
    """`Writer<W>`; W is a path (from_path), a file descriptor, a Python file-like (callback
    sink) or memory (`Writer::new(Vec::new(), header)`)."""

    def __init__(self, inner=None, header=None, _handle=None):
        self._w = _handle
        self._cb = None
        if _handle is not None:
            return
        hp = C.byref(header._h) if header is not None else None
        out = C.c_void_p()
        if inner is None or isinstance(inner, (bytearray, list)):  # Vec<u8>
            _check(lib.ibu_writer_open_mem(hp, C.byref(out)))
        elif isinstance(inner, int):
            _check(lib.ibu_writer_open_fd(inner, hp, C.byref(out)))
        else:  # any object with .write(bytes) [and .flush()]
            def _wr(_u, data, n, _f=inner):
                try:
                    _f.write(C.string_at(data, n))
                    return 0
                except OSError as e:
                    return e.errno or 5
            def _fl(_u, _f=inner):
                try:
                    if hasattr(_f, "flush"):
                        _f.flush()
                    return 0
                except OSError as e:
                    return e.errno or 5
            self._cb = (_lib.WRITE_FN(_wr), _lib.FLUSH_FN(_fl))
            _check(lib.ibu_writer_open_callback(self._cb[0], self._cb[1], None, hp, C.byref(out)))
        self._w = out

    @classmethod
    def new(cls, inner, header):  # Writer::new :129-143
        return cls(inner, header)

    @classmethod
    def new_headless(cls, inner=None):  # :169-179
        return cls(inner, None)

    @classmethod
    def from_path(cls, path, header):  # :556-559
        out = C.c_void_p()
        _check(lib.ibu_writer_open_path(str(path).encode(), C.byref(header._h) if header else None, C.byref(out)))
        return cls(_handle=out)

    @classmethod
    def from_stdout(cls, header):  # :587-589
        return cls(1, header)

    @classmethod
    def from_optional_path(cls, path, header):  # :617-626
        return cls.from_path(path, header) if path is not None else cls.from_stdout(header)

    def write_record(self, record):  # :260-273
        r = Record(*record)._c()
        _check(lib.ibu_writer_write_record(self._w, C.byref(r)))

    def write_batch(self, records):  # :315-318
        a = records_array(records)
        _check(lib.ibu_writer_write_batch(self._w, _hptr(a), a.shape[0]))

    def write_iter(self, records):  # :383-391
        for r in records:
            self.write_record(r)

    def write_batch_device(self, ctx, d_records, n, ring=None, stream=None):
        """Device-resident AoS records -> pinned ring -> this writer (write_batch rules).  `stream`: the stream the
        records were produced on (None: the context's own)."""
        st = CStreamStats()
        _check(lib.ibu_writer_write_batch_device_on(self._w, ctx._c, _ring(ring), _dptr(d_records), n, stream, C.byref(st)))
        return st

    def write_ascii_batch(self, ctx, bc_ascii, umi_ascii, bc_len, umi_len, index=None, first_index=0, ring=None):
        """Host ASCII rows (uint8 arrays of n*bc_len / n*umi_len bytes) -> 2-bit records -> this writer, encoded on
        the GPU (README.md:38-47's encode-then-write loop as one batch call)."""
        bc = np.ascontiguousarray(bc_ascii, dtype=np.uint8).reshape(-1)
        umi = np.ascontiguousarray(umi_ascii, dtype=np.uint8).reshape(-1)
        n = bc.size // bc_len
        assert bc.size == n * bc_len and umi.size == n * umi_len
        idx = None if index is None else np.ascontiguousarray(index, dtype=np.uint64)
        st = CStreamStats()
        _check(lib.ibu_writer_write_ascii_batch(self._w, ctx._c, _ring(ring), _hptr(bc), _hptr(umi),
                                                _hptr(idx) if idx is not None else None, first_index, n, bc_len, umi_len,
                                                C.byref(st)))
        return st

    def ingest(self, other):  # :477-482
        _check(lib.ibu_writer_ingest(self._w, other._w))

    def finish(self):  # :429-433
        _check(lib.ibu_writer_finish(self._w))

    def records_written(self):  # :207-209
        return lib.ibu_writer_records_written(self._w)

    def inner_bytes(self):
        """What a `Vec<u8>` sink holds right now (tests peek at `writer.inner`)."""
        p, n = C.c_void_p(), C.c_size_t()
        _check(lib.ibu_writer_mem_view(self._w, C.byref(p), C.byref(n)))
        return C.string_at(p, n.value) if n.value else b""

    def into_inner(self):  # :507-511 — no flush
        p, n = C.c_void_p(), C.c_size_t()
        w, self._w = self._w, None
        _check(lib.ibu_writer_into_inner(w, C.byref(p), C.byref(n)))
        if not p:
            return None
        data = C.string_at(p, n.value)
        lib.ibu_free(p)
        return data

    def close(self):  # Drop :519-523
        if getattr(self, "_w", None):
            lib.ibu_writer_close(self._w)
            self._w = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


# ---- Reader (src/io/reader.rs) -------------------------------------------------------------------------------
class Reader:
    """`Reader<R>`; R is bytes (Cursor), a file descriptor or a file-like with .readinto/.read."""

    def __init__(self, inner=None, _handle=None):
        self._r, self._keep, self._cb = _handle, None, None
        if _handle is not None:
            return
        out = C.c_void_p()
        if isinstance(inner, (bytes, bytearray, memoryview, np.ndarray)):
            self._keep = np.frombuffer(bytes(inner), dtype=np.uint8)
            _check(lib.ibu_reader_open_mem(_hptr(self._keep), self._keep.size, C.byref(out)))
        elif isinstance(inner, int):
            _check(lib.ibu_reader_open_fd(inner, C.byref(out)))
        else:
            def _rd(_u, dst, cap, got, _f=inner):
                try:
                    b = _f.read(cap)
                    C.memmove(dst, b, len(b))
                    got[0] = len(b)
                    return 0
                except OSError as e:
                    return e.errno or 5
            self._cb = _lib.READ_FN(_rd)
            _check(lib.ibu_reader_open_callback(self._cb, None, C.byref(out)))
        self._r = out

    @classmethod
    def new(cls, inner):  # :152-176
        return cls(inner)

    @classmethod
    def from_path(cls, path):  # :345-352 (format sniffed: gzip / BGZF / bzip2 / xz / zstd, like niffler)
        out = C.c_void_p()
        _check(lib.ibu_reader_open_path(str(path).encode(), C.byref(out)))
        return cls(_handle=out)

    @classmethod
    def from_stdin(cls):  # :389-396
        out = C.c_void_p()
        _check(lib.ibu_reader_open_fd(0, C.byref(out)))
        return cls(_handle=out)

    @classmethod
    def from_optional_path(cls, path):  # :425-434
        return cls.from_path(path) if path is not None else cls.from_stdin()

    def header(self):  # :244-246
        h = CHeader()
        _check(lib.ibu_reader_header(self._r, C.byref(h)))
        return Header._wrap(h)

    def read_batch(self):  # :218-242
        has = C.c_int32()
        _check(lib.ibu_reader_read_batch(self._r, C.byref(has)))
        return bool(has.value)

    @property
    def bytes_read(self):
        return lib.ibu_reader_bytes_read(self._r)

    def buffered(self):
        """Zero-copy numpy view of the records left in the current refill."""
        p, n = C.c_void_p(), C.c_size_t()
        _check(lib.ibu_reader_buffered(self._r, C.byref(p), C.byref(n)))
        if not n.value:
            return np.empty(0, dtype=REC_DTYPE)
        raw = (C.c_uint8 * (n.value * RECORD_SIZE)).from_address(p.value)
        raw._owner = self  # valid until the next read_batch(); keeps the reader itself alive
        return np.frombuffer(raw, dtype=REC_DTYPE)

    def consume(self, n):
        _check(lib.ibu_reader_consume(self._r, n))

    def __iter__(self):
        return self

    def __next__(self):  # Iterator::next :279-306; an Err item is raised
        r, got = CRecord(), C.c_int32()
        _check(lib.ibu_reader_next(self._r, C.byref(r), C.byref(got)))
        if not got.value:
            raise StopIteration
        return Record(r.barcode, r.umi, r.index)

    def process_device(self, ctx, proc=PROC_REDUCE, sink=None, ring=None):
        """Stream the rest of this reader (plain or gzip) through the pinned ring to a device
        processor.  Returns (result, stats).  PROC_DECODE: sink = (d_bc, d_umi, d_index[, cap_records]); without
        cap_records the capacity is what the DeviceBuffers hold.  A stream longer than that raises InvalidArg."""
        h = self.header()
        return ctx._run_proc(lambda c, rg, s, st: lib.ibu_reader_process_device(self._r, c, rg, proc, s, st),
                             proc, sink, ring, (h.bc_len, h.umi_len))

    def device_stream(self, ctx, ring=None):
        """Pull-style device stream over the rest of this reader (ibu_stream_open_reader): iterate DeviceBatches."""
        return DeviceStream._open(lambda out: lib.ibu_stream_open_reader(self._r, ctx._c, _ring(ring), out), ctx, self)

    def close(self):
        for st in list(getattr(self, "_streams", ())):   # a producer thread must not outlive the reader it reads
            st.close()
        if getattr(self, "_r", None):
            lib.ibu_reader_close(self._r)
            self._r = None

    __del__ = close


def load_to_vec(path):  # src/io/reader.rs:510-535
    h, p, n = CHeader(), C.c_void_p(), C.c_size_t()
    _check(lib.ibu_load_to_vec(str(path).encode(), C.byref(h), C.byref(p), C.byref(n)))
    # (c_char * nbytes).from_address: string_at takes a C int and fails beyond 2 GiB
    recs = (np.frombuffer((C.c_char * (n.value * RECORD_SIZE)).from_address(p.value), dtype=REC_DTYPE).copy()
            if n.value else np.empty(0, REC_DTYPE))
    lib.ibu_free(p)
    return Header._wrap(h), recs


# ---- parallel (src/parallel.rs, src/io/mmap.rs) ------------------------------------------------------------------
class ParallelProcessor:
    """src/parallel.rs:100-190.  Subclass and override process_record (and optionally
    on_batch_complete / set_tid / get_tid).  `clone()` is `P: Clone`: shallow copy, so shared
    accumulators (the Arc<Mutex<..>> pattern) stay shared."""

    def process_record(self, record):
        raise NotImplementedError

    def on_batch_complete(self):  # default Ok(()) :137-139
        return None

    def set_tid(self, tid):  # default no-op :166-168
        pass

    def get_tid(self):  # default None :187-189
        return None

    def clone(self):
        return copy.copy(self)


class ProcessError(Exception):
    """Raise from process_record / on_batch_complete to return IbuError::Process."""


def shard_range(length, n_shards, shard):  # mmap.rs:297-307
    s, e = C.c_size_t(), C.c_size_t()
    _check(lib.ibu_shard_range(length, n_shards, shard, C.byref(s), C.byref(e)))
    return s.value, e.value


class MmapReader:
    """src/io/mmap.rs:99-332 (MmapReader + its ParallelReader impl)."""

    def __init__(self, path=None, _handle=None):
        if _handle is None:
            _handle = C.c_void_p()
            _check(lib.ibu_mmap_open(str(path).encode(), C.byref(_handle)))
        self._m = _handle

    @classmethod
    def new(cls, path):  # :143-161
        return cls(path)

    def clone(self):  # Arc clone :99
        out = C.c_void_p()
        _check(lib.ibu_mmap_clone(self._m, C.byref(out)))
        return MmapReader(_handle=out)

    def len(self):  # :178-180
        return lib.ibu_mmap_len(self._m)

    __len__ = len

    def header(self):  # :201-203
        h = CHeader()
        _check(lib.ibu_mmap_header(self._m, C.byref(h)))
        return Header._wrap(h)

    def slice(self, start, end):  # :253-270 — zero-copy view into the map
        p, n = C.c_void_p(), C.c_size_t()
        _check(lib.ibu_mmap_slice(self._m, start, end, C.byref(p), C.byref(n)))
        raw = (C.c_uint8 * (n.value * RECORD_SIZE)).from_address(p.value)
        raw._owner = self  # the borrow `&'a [Record]`: the view keeps this handle (and so the map) alive
        return np.frombuffer(raw, dtype=REC_DTYPE)

    def map_ptr(self):
        return lib.ibu_mmap_base(self._m)

    def process_parallel(self, processor, num_threads):  # :286-332, host processor (user code)
        clones, errors = {}, []

        def _clone(_u):
            c = processor.clone()
            key = len(clones) + 1
            clones[key] = c
            return key

        def _drop(_k):
            pass

        def _proc(k, rec):
            try:
                r = rec[0]
                clones[k].process_record(Record(r.barcode, r.umi, r.index))
                return 0
            except Exception as e:  # -> IbuError::Process
                errors.append(e)
                return 1

        def _batch(k):
            try:
                clones[k].on_batch_complete()
                return 0
            except Exception as e:
                errors.append(e)
                return 1

        vt = CProcessorVTable(_lib.CLONE_FN(_clone), _lib.DROP_FN(_drop), _lib.PROCESS_FN(_proc),
                              _lib.BATCH_FN(_batch), _lib.TID_FN())
        _check(lib.ibu_mmap_process_parallel(self._m, C.byref(vt), None, num_threads))

    def process_device(self, ctx, proc=PROC_REDUCE, shard=0, n_shards=1, sink=None, ring=None):
        """One shard of the static split on one GPU, through the pinned ring."""
        h = self.header()
        return ctx._run_proc(
            lambda c, rg, s, st: lib.ibu_mmap_process_device(self._m, c, rg, proc, shard, n_shards, s, st),
            proc, sink, ring, (h.bc_len, h.umi_len))

    def process_devices(self, devices=(), proc=PROC_REDUCE, sinks=None, ring=None, contexts=None):
        """process_parallel(processor, n) with a GPU per worker, in ONE call (mmap.rs:286-332; ibu_mmap_process_devices):
        worker i = one host thread + one context on devices[i], shard i of the static split, first error in worker order
        wins.  devices = () means every visible device.  contexts: a list of Contexts to reuse instead of device ordinals
        (ibu_mmap_process_contexts).  PROC_REDUCE -> (total, [partial per device], [stats]); PROC_DECODE: `sinks` = one
        (d_bc, d_umi, d_index, cap_records) per device, each on its device -> (count, None, [stats])."""
        n = len(contexts) if contexts is not None else (len(devices) or device_count())
        stats = (CStreamStats * max(n, 1))()
        total = CReduceResult()
        if proc == PROC_REDUCE:
            parts = (CReduceResult * max(n, 1))()
            sink_p = C.cast(parts, C.c_void_p)
        else:
            if sinks is None or len(sinks) != n:
                raise ValueError("PROC_DECODE needs one sink per device")
            arr = (CDecodeSink * n)(*[CDecodeSink(_dptr(a), _dptr(b), _dptr(c), int(cap)) for a, b, c, cap in sinks])
            sink_p = C.cast(arr, C.c_void_p)
        if contexts is not None:
            cs = (C.c_void_p * n)(*[c._c for c in contexts])
            _check(lib.ibu_mmap_process_contexts(self._m, cs, n, _ring(ring), proc, sink_p, C.byref(total), stats))
        else:
            devs = (C.c_int32 * len(devices))(*devices) if len(devices) else None
            _check(lib.ibu_mmap_process_devices(self._m, devs, len(devices), _ring(ring), proc, sink_p, C.byref(total), stats))
        as_dict = lambda r: {"count": r.count, "sum": list(r.sum), "xor": list(r.xor_)}
        if proc == PROC_REDUCE:
            return as_dict(total), [as_dict(parts[i]) for i in range(n)], list(stats)[:n]
        return total.count, None, list(stats)[:n]

    def device_stream(self, ctx, shard=0, n_shards=1, ring=None):
        """Pull-style device stream over one shard of the static split (ibu_stream_open_mmap): iterate DeviceBatches."""
        return DeviceStream._open(lambda out: lib.ibu_stream_open_mmap(self._m, ctx._c, _ring(ring), shard, n_shards, out), ctx, self)

    def decode_to_host(self, ctx, shard=0, n_shards=1, ring=None, want=("bc", "umi", "index"), out=None):
        """One shard -> (barcode ASCII [n, bc_len], UMI ASCII [n, umi_len], index [n]) as numpy arrays in host
        memory, unpacked on the GPU.  Columns not in `want` come back as None.  out = (bc, umi, index) arrays of those
        shapes to fill instead of fresh ones (memory that has been written before takes no page faults: 1.7x the rate)."""
        a, b = shard_range(self.len(), n_shards, shard)
        n, h = b - a, self.header()
        if out is not None:
            bc, umi, idx = out
            for arr, shape, dt in ((bc, (n, h.bc_len), np.uint8), (umi, (n, h.umi_len), np.uint8), (idx, (n,), np.uint64)):
                if arr is not None and (arr.shape != shape or arr.dtype != dt or not arr.flags.c_contiguous):
                    raise ValueError("out: C-contiguous arrays of shapes (n, bc_len) u8, (n, umi_len) u8, (n,) u64")
        else:
            bc = np.empty((n, h.bc_len), dtype=np.uint8) if "bc" in want else None
            umi = np.empty((n, h.umi_len), dtype=np.uint8) if "umi" in want else None
            idx = np.empty(n, dtype=np.uint64) if "index" in want else None
        st = CStreamStats()
        _check(lib.ibu_mmap_decode_to_host(self._m, ctx._c, _ring(ring), shard, n_shards,
                                           _hptr(bc) if bc is not None else None, _hptr(umi) if umi is not None else None,
                                           _hptr(idx) if idx is not None else None, C.byref(st)))
        return bc, umi, idx, st

    def close(self):
        for st in list(getattr(self, "_streams", ())):   # a producer thread must not outlive the map it copies from
            st.close()
        if getattr(self, "_m", None):
            lib.ibu_mmap_close(self._m)
            self._m = None

    __del__ = close


# ---- device side ----------------------------------------------------------------------------------------------------
def _dptr(x):
    """int | DeviceBuffer | torch.Tensor | None -> c_void_p."""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "ptr"):
        return C.c_void_p(x.ptr)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    raise TypeError(f"not a device pointer: {type(x)}")


def _ring(r):
    if r is None:
        return None
    if isinstance(r, CRingConfig):
        return C.byref(r)
    return C.byref(CRingConfig(r.get("slots", 0), r.get("slot_records", 0), r.get("feeder_threads", 0), 0))


def key_plan(or_words, and_words):
    """ibu_key_plan_t for records whose OR / AND words are given (Context.census, combined over the shards with | and &).
    plan.k = number of varying key bytes; compact / expand need plan.k <= 12."""
    plan = CKeyPlan()
    _check(lib.ibu_key_plan_init((C.c_uint64 * 3)(*[int(v) & (2**64 - 1) for v in or_words]),
                                 (C.c_uint64 * 3)(*[int(v) & (2**64 - 1) for v in and_words]), C.byref(plan)))
    return plan


def device_count():
    n = C.c_int32()
    rc = lib.ibu_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class DeviceBuffer:
    """hipMalloc'ed bytes owned through the context (for callers without torch)."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes, self.owned = ctx, int(nbytes), True
        p = C.c_void_p()
        _check(lib.ibu_device_alloc(ctx._c, self.nbytes, C.byref(p)))
        self.ptr = p.value
        ctx._buffers.add(self)   # a context that closes first frees what it still owns (Context.close)

    @classmethod
    def wrap(cls, ctx, ptr, nbytes):
        """View of device memory owned by someone else (e.g. what load_to_device returned): never freed here."""
        b = cls.__new__(cls)
        b.ctx, b.ptr, b.nbytes, b.owned = ctx, int(ptr), int(nbytes), False
        return b

    @property
    def __cuda_array_interface__(self):
        """Flat uint8 view for consumers of the CUDA array interface (torch.as_tensor(buf, device=...) on ROCm too):
        bench.py uses it to compare buffers with torch as a checker that is not this library."""
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2, "strides": None}

    def upload(self, host):
        a = np.ascontiguousarray(host)
        assert a.nbytes <= self.nbytes
        _check(lib.ibu_memcpy_h2d(self.ctx._c, self.ptr, _hptr(a), a.nbytes, None))
        self.ctx.synchronize()
        return self

    def download(self, dtype=np.uint8, count=None, offset=0):
        dt = np.dtype(dtype)
        nbytes = (self.nbytes - offset) if count is None else count * dt.itemsize
        out = np.empty(nbytes // dt.itemsize, dtype=dt)
        _check(lib.ibu_memcpy_d2h(self.ctx._c, _hptr(out), self.ptr + offset, out.nbytes, None))
        self.ctx.synchronize()
        return out

    def free(self):
        if getattr(self, "ptr", 0) and getattr(self, "owned", False) and getattr(self.ctx, "_c", None):
            lib.ibu_device_free(self.ctx._c, self.ptr)
        self.ptr = 0

    __del__ = free


class DeviceBatch:
    """One device-resident batch of a DeviceStream: `n` AoS records at device pointer `ptr` (a ring slot), the first of them
    record number `first_index` of the stream.  release() (or leaving the `with` block) gives the slot back: it is refilled once
    the work queued so far on `stream` (default: the context's) has run."""

    def __init__(self, stream, ptr, n, first_index):
        self._s, self.ptr, self.n, self.first_index, self.nbytes = stream, ptr, n, first_index, n * RECORD_SIZE

    def download(self):
        """The batch as a host REC_DTYPE array (synchronises the context's stream)."""
        return DeviceBuffer.wrap(self._s.ctx, self.ptr, self.nbytes).download().view(REC_DTYPE)

    def release(self, stream=None):
        if self.ptr:
            p, self.ptr = self.ptr, 0
            _check(lib.ibu_stream_release(self._s._s, p, stream))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.release()


class DeviceStream:
    """ibu_stream_t — the device form of `Reader::read_batch` + `Iterator` (reader.rs:218-242, :279-306) and of the per-batch loop
    of process_parallel (mmap.rs:312-320): iterate to pull one DeviceBatch at a time, run any kernels on it, release it.
    A source error (TruncatedRecord, Io, Niffler ...) is raised by the iteration after the batches in front of it."""

    @classmethod
    def _open(cls, call, ctx, source):
        out = C.c_void_p()
        _check(call(C.byref(out)))
        s = cls.__new__(cls)
        s._s, s.ctx, s._source = out, ctx, source   # the source (Reader / MmapReader) must outlive the stream:
        if not hasattr(source, "_streams"):         # closing the source closes the streams still open on it first
            source._streams = weakref.WeakSet()
        source._streams.add(s)
        return s

    def header(self):
        h = CHeader()
        _check(lib.ibu_stream_header(self._s, C.byref(h)))
        return Header._wrap(h)

    def next_batch(self, stream=None):
        """-> DeviceBatch, or None at the end of the stream.  `stream`: the stream the batch will be read on (None: the context's)."""
        p, n, first = C.c_void_p(), C.c_size_t(), C.c_uint64()
        _check(lib.ibu_stream_next(self._s, stream, C.byref(p), C.byref(n), C.byref(first)))
        if n.value == 0:
            return None
        return DeviceBatch(self, p.value, n.value, first.value)

    def __iter__(self):
        return self

    def __next__(self):
        b = self.next_batch()
        if b is None:
            raise StopIteration
        return b

    def stats(self):
        st = CStreamStats()
        _check(lib.ibu_stream_stats(self._s, C.byref(st)))
        return st

    def close(self):
        if getattr(self, "_s", None):
            lib.ibu_stream_close(self._s)
            self._s = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


INFLATE_BLOCK_DTYPE = np.dtype([("comp_offset", "<u8"), ("out_offset", "<i8"), ("comp_len", "<u4"), ("out_len", "<u4"), ("crc32", "<u4"),
                                ("reserved", "<u4")])   # ibu_inflate_block_t


def bgzf_scan(buf, final=True, cap=None):
    """ibu_bgzf_scan over `buf` (bytes-like): (ctypes array of ibu_inflate_block_t, consumed bytes, uncompressed bytes, status).
    status is 0 or 2 = IBU_ERR_NIFFLER (the blocks in front of the bad spot are described).  cap: at most that many blocks."""
    from ._lib import CInflateBlock
    a = np.frombuffer(buf, dtype=np.uint8)
    step = 1 << 16
    tmp = (CInflateBlock * step)()
    parts, pos, total, rc, left = [], 0, 0, 0, (None if cap is None else int(cap))
    while pos < len(a) and (left is None or left > 0):
        n, consumed, out_bytes = C.c_size_t(), C.c_size_t(), C.c_uint64()
        want = step if left is None else min(step, left)
        rc = lib.ibu_bgzf_scan(_hptr(a[pos:]), len(a) - pos, 1 if final else 0, tmp, want, C.byref(n), C.byref(consumed), C.byref(out_bytes))
        if n.value:
            part = np.frombuffer(tmp, dtype=INFLATE_BLOCK_DTYPE, count=n.value).copy()
            part["comp_offset"] += pos
            part["out_offset"] += total
            parts.append(part)
            if left is not None:
                left -= n.value
        pos += consumed.value
        total += out_bytes.value
        if rc or consumed.value == 0 or n.value < want:
            break
    allb = np.concatenate(parts) if parts else np.empty(0, INFLATE_BLOCK_DTYPE)
    blocks = (CInflateBlock * len(allb)).from_buffer_copy(allb.tobytes()) if len(allb) else (CInflateBlock * 0)()
    return blocks, pos, total, rc


def numa_of_pci(pci_bus_id, sysfs_root=None):
    """(node, cpulist, usable_cpus) of a PCI function from a sysfs tree (ibu_numa_of_pci); node -1 = the platform does not say."""
    node, usable, buf = C.c_int32(), C.c_int32(), C.create_string_buffer(256)
    _check(lib.ibu_numa_of_pci(None if sysfs_root is None else str(sysfs_root).encode(), str(pci_bus_id).encode(), C.byref(node),
                               buf, 256, C.byref(usable)))
    return node.value, buf.value.decode(), usable.value


class Context:
    """ibu_ctx_t: one per host thread and GPU.  Every method launches asynchronously on
    `stream` (default: the context's own stream) unless it says it synchronises."""

    def __init__(self, device=0):
        c = C.c_void_p()
        _check(lib.ibu_ctx_create(device, C.byref(c)))
        self._c = c
        self._buffers = weakref.WeakSet()   # live DeviceBuffers allocated through this context

    @property
    def stream(self):
        return lib.ibu_ctx_stream(self._c)

    def synchronize(self, stream=None):
        _check(lib.ibu_ctx_synchronize(self._c, stream))

    def set_option(self, key, value):
        _check(lib.ibu_ctx_set_option(self._c, key.encode(), int(value)))

    def numa(self):
        """Where the device hangs off the host and where the pinned ring landed (ibu_ctx_numa)."""
        i = CNumaInfo()
        _check(lib.ibu_ctx_numa(self._c, C.byref(i)))
        return {"mode": i.mode, "node": i.node, "usable_cpus": i.usable_cpus, "ring_node": i.ring_node,
                "ring_placed": bool(i.ring_placed), "pci_bus_id": i.pci_bus_id.decode(), "cpulist": i.cpulist.decode()}

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def alloc_probed(self, nbytes, tries):
        """ibu_device_alloc_probed: up to `tries` candidate allocations (all held at once), the one a write + read streams
        over fastest is kept, the others freed.  -> (DeviceBuffer, {"tries", "chosen", "ms": [...]})."""
        p, rep = C.c_void_p(), CAllocProbe()
        _check(lib.ibu_device_alloc_probed(self._c, int(nbytes), int(tries), C.byref(p), C.byref(rep)))
        b = DeviceBuffer.wrap(self, p.value, nbytes)
        b.owned = True
        self._buffers.add(b)
        return b, {"tries": rep.tries, "chosen": rep.chosen, "ms": [round(float(rep.ms[k]), 4) for k in range(rep.tries)]}

    def upload(self, host):
        a = np.ascontiguousarray(host)
        return DeviceBuffer(self, max(a.nbytes, 16)).upload(a)

    # K1 / K1'
    def deserialize(self, d_records, n, d_barcode, d_umi, d_index, stream=None):
        _check(lib.ibu_deserialize(self._c, _dptr(d_records), n, _dptr(d_barcode), _dptr(d_umi), _dptr(d_index), stream))

    def serialize(self, d_barcode, d_umi, d_index, n, d_records, stream=None):
        _check(lib.ibu_serialize(self._c, _dptr(d_barcode), _dptr(d_umi), _dptr(d_index), n, _dptr(d_records), stream))

    # column codec
    def unpack_2bit(self, d_codes, n, length, d_ascii, stream=None):
        _check(lib.ibu_unpack_2bit(self._c, _dptr(d_codes), n, length, _dptr(d_ascii), stream))

    def pack_2bit(self, d_ascii, n, length, d_codes, stream=None):
        _check(lib.ibu_pack_2bit(self._c, _dptr(d_ascii), n, length, _dptr(d_codes), stream))

    # K2 / K3
    def decode_ascii(self, d_records, n, bc_len, umi_len, d_bc, d_umi, d_index, stream=None):
        _check(lib.ibu_decode_ascii(self._c, _dptr(d_records), n, bc_len, umi_len, _dptr(d_bc), _dptr(d_umi),
                                    _dptr(d_index), stream))

    def encode_ascii(self, d_bc, d_umi, d_index, n, bc_len, umi_len, d_records, first_index=0, stream=None):
        _check(lib.ibu_encode_ascii(self._c, _dptr(d_bc), _dptr(d_umi), _dptr(d_index), first_index, n, bc_len,
                                    umi_len, _dptr(d_records), stream))

    def codec_status(self, stream=None):
        """Synchronises; raises IbuError(InvalidBase) if any pack/encode since the last call saw a bad byte."""
        fb, nb = C.c_uint64(), C.c_uint64()
        _check(lib.ibu_codec_status(self._c, stream, C.byref(fb), C.byref(nb)))

    # K4
    def reduce(self, d_records, n, stream=None, reset=True, fetch=True):
        if reset:
            _check(lib.ibu_reduce_reset(self._c, stream))
        _check(lib.ibu_reduce(self._c, _dptr(d_records), n, stream))
        return self.reduce_fetch(stream) if fetch else None

    def reduce_fetch(self, stream=None):
        r = CReduceResult()
        _check(lib.ibu_reduce_fetch(self._c, stream, C.byref(r)))
        return {"count": r.count, "sum": list(r.sum), "xor": list(r.xor_)}

    def generate(self, seed, first, n, bc_len, umi_len, d_records, stream=None):
        _check(lib.ibu_generate(self._c, seed, first, n, bc_len, umi_len, _dptr(d_records), stream))

    def copy(self, d_dst, d_src, nbytes, stream=None):
        _check(lib.ibu_device_copy(self._c, _dptr(d_dst), _dptr(d_src), nbytes, stream))

    def sort_records(self, d_records, d_tmp, n, stream=None):
        _check(lib.ibu_sort_records(self._c, _dptr(d_records), _dptr(d_tmp), n, stream))

    @staticmethod
    def sort_records_contexts(ctxs, shards):
        """The sort over several shards, one per context (= per GPU), in one call (ibu_sort_records_contexts): shards =
        [(d_records, d_tmp, n, capacity_in_records), ...]; returns the new record counts — shard i holds the i-th
        contiguous range of the global order."""
        if len(shards) != len(ctxs):   # the C side reads n_ctxs entries of both arrays
            raise ValueError(f"{len(ctxs)} contexts but {len(shards)} shards")
        arr = (C.c_void_p * len(ctxs))(*[c._c for c in ctxs])
        sh = (_lib.CSortShard * len(shards))()
        for k, (r, t, n, cap) in enumerate(shards):
            sh[k].d_records, sh[k].d_tmp, sh[k].n, sh[k].capacity = _dptr(r).value, _dptr(t).value, n, cap
        _check(lib.ibu_sort_records_contexts(arr, len(ctxs), sh))
        return [int(x.n) for x in sh]

    def barcode_counts(self, d_sorted_records, n, unique_umis=True, stream=None):
        """BarcodeAnalyzer (parallel.rs:72-98) on sorted device records ->
        (barcodes, counts, unique_umis | None) as numpy u64 arrays in ascending barcode order."""
        nb, npairs = C.c_size_t(), C.c_size_t()
        _check(lib.ibu_barcode_counts(self._c, _dptr(d_sorted_records), n, None, None, None, 0, C.byref(nb),
                                      C.byref(npairs), stream))
        u = nb.value
        if u == 0:
            e = np.empty(0, np.uint64)
            return e, e.copy(), (e.copy() if unique_umis else None)
        d_b, d_c = self.alloc(8 * u), self.alloc(8 * u)
        d_u = self.alloc(8 * u) if unique_umis else None
        _check(lib.ibu_barcode_counts(self._c, _dptr(d_sorted_records), n, _dptr(d_b), _dptr(d_c), _dptr(d_u), u,
                                      C.byref(nb), C.byref(npairs), stream))
        self.synchronize(stream)
        return (d_b.download(np.uint64), d_c.download(np.uint64), d_u.download(np.uint64) if d_u else None)

    def inflate_blocks(self, d_comp, blocks, d_out, stream=None):
        """ibu_inflate_blocks_device: the deflate blocks `blocks` (a ctypes array of ibu_inflate_block_t, or what bgzf_scan
        returned) of the compressed bytes at d_comp -> d_out.  Returns (status per block as numpy u32, first bad block or None);
        synchronises."""
        n = len(blocks)
        if n == 0:
            return np.empty(0, np.uint32), None
        raw = np.frombuffer(bytes(blocks), dtype=np.uint8) if not isinstance(blocks, np.ndarray) else blocks.view(np.uint8)
        d_blocks = self.upload(raw)
        d_status = self.alloc(4 * n + 4)
        first = np.full(1, 0xFFFFFFFF, np.uint32)
        _check(lib.ibu_memcpy_h2d(self._c, d_status.ptr + 4 * n, _hptr(first), 4, stream))
        _check(lib.ibu_inflate_blocks_device(self._c, _dptr(d_comp), d_blocks.ptr, n, _dptr(d_out), d_status.ptr, d_status.ptr + 4 * n, stream))
        self.synchronize(stream)
        st = d_status.download(np.uint32)
        d_blocks.free()
        d_status.free()
        return st[:n].copy(), (None if st[n] == 0xFFFFFFFF else int(st[n]))

    def lower_bound(self, d_sorted_records, n, d_keys, k, d_pos, stream=None):
        """d_pos[j] (u64, device) = first position whose record is >= key j (24-byte records in d_keys); asynchronous."""
        _check(lib.ibu_lower_bound_records(self._c, _dptr(d_sorted_records), n, _dptr(d_keys), k, _dptr(d_pos), stream))

    # compacted keys (the exchange format of the multi-GPU sort)
    def census(self, d_records, n, stream=None):
        """{"or": [3], "and": [3], "index_drops": bool, "order_drops": bool} of n device records; synchronises."""
        out = (C.c_uint64 * 8)()
        _check(lib.ibu_records_census(self._c, _dptr(d_records), n, out, stream))
        return {"or": [int(v) for v in out[0:3]], "and": [int(v) for v in out[3:6]], "index_drops": bool(out[6]),
                "order_drops": bool(out[7])}

    def compact(self, plan, d_records, n, d_elems, stream=None):
        """n records -> n 12-byte elements (plan: key_plan(); plan.k <= 12); asynchronous."""
        _check(lib.ibu_records_compact(self._c, C.byref(plan), _dptr(d_records), n, _dptr(d_elems), stream))

    def expand(self, plan, d_elems, n, d_records, stream=None):
        """n 12-byte elements -> the n records they stand for; asynchronous."""
        _check(lib.ibu_records_expand(self._c, C.byref(plan), _dptr(d_elems), n, _dptr(d_records), stream))

    def first_mismatch(self, d_a, d_b, n, stream=None):
        """Index of the first of n records in which the two device slices differ; n if they are equal
        (`a == b` on record slices: Record derives PartialEq / Eq, record.rs:58)."""
        f = C.c_uint64()
        _check(lib.ibu_records_first_mismatch(self._c, _dptr(d_a), _dptr(d_b), n, C.byref(f), stream))
        return int(f.value)

    def records_equal(self, d_a, d_b, n, stream=None):
        return self.first_mismatch(d_a, d_b, n, stream) == n

    def is_sorted(self, d_records, n, stream=None):
        s = C.c_int32()
        _check(lib.ibu_is_sorted(self._c, _dptr(d_records), n, stream, C.byref(s)))
        return bool(s.value)

    # streams
    def load_to_device(self, path, ring=None, d_records=None, cap_records=0):
        """Device analogue of load_to_vec -> (Header, device pointer (int) or the given buffer, n, stats)."""
        h, n, st = CHeader(), C.c_size_t(), CStreamStats()
        p = C.c_void_p(_dptr(d_records).value if d_records is not None else None)
        _check(lib.ibu_load_to_device(self._c, str(path).encode(), _ring(ring), C.byref(h), C.byref(p), cap_records,
                                      C.byref(n), C.byref(st)))
        return Header._wrap(h), p.value, n.value, st

    def load_bgzf_to_device(self, path, ring=None, d_records=None, cap_records=0):
        """ibu_load_bgzf_to_device: a BGZF file of the records -> device records, inflated on the device (the compressed bytes cross
        the link) -> (Header, device pointer (int) or the given buffer, n, stats)."""
        h, n, st = CHeader(), C.c_size_t(), CStreamStats()
        p = C.c_void_p(_dptr(d_records).value if d_records is not None else None)
        _check(lib.ibu_load_bgzf_to_device(self._c, str(path).encode(), _ring(ring), C.byref(h), C.byref(p), cap_records,
                                           C.byref(n), C.byref(st)))
        return Header._wrap(h), p.value, n.value, st

    def load_bgzf_shard_to_device(self, path, shard, n_shards, ring=None, d_records=None, cap_records=0):
        """ibu_load_bgzf_shard_to_device: shard `shard` of `n_shards` of the file's records (the split of process_parallel)
        -> (Header, device pointer, n, first record's number, stats)."""
        h, n, first, st = CHeader(), C.c_size_t(), C.c_uint64(), CStreamStats()
        p = C.c_void_p(_dptr(d_records).value if d_records is not None else None)
        _check(lib.ibu_load_bgzf_shard_to_device(self._c, str(path).encode(), _ring(ring), shard, n_shards, C.byref(h), C.byref(p), cap_records,
                                                 C.byref(n), C.byref(first), C.byref(st)))
        return Header._wrap(h), p.value, n.value, first.value, st

    def free(self, ptr):
        _check(lib.ibu_device_free(self._c, ptr))

    def _run_proc(self, call, proc, sink, ring, lens=(1, 1)):
        st = CStreamStats()
        if proc == PROC_REDUCE:
            r = CReduceResult()
            _check(call(self._c, _ring(ring), C.cast(C.byref(r), C.c_void_p), C.byref(st)))
            return {"count": r.count, "sum": list(r.sum), "xor": list(r.xor_)}, st
        cols = list(sink[:3])
        if len(sink) > 3:
            cap = int(sink[3])
        else:  # capacity = what the buffers hold (DeviceBuffer / torch tensors know their size)
            sizes = [(c, w) for c, w in zip(cols, (lens[0], lens[1], 8)) if c is not None]
            if not all(hasattr(c, "nbytes") or hasattr(c, "numel") for c, _ in sizes):
                raise TypeError("a decode sink of raw pointers needs its capacity: sink=(d_bc, d_umi, d_index, cap_records)")
            cap = min((c.nbytes if hasattr(c, "nbytes") else c.numel() * c.element_size()) // w for c, w in sizes) if sizes else 0
        s = CDecodeSink(*(_dptr(x) for x in cols), cap)
        _check(call(self._c, _ring(ring), C.cast(C.byref(s), C.c_void_p), C.byref(st)))
        return None, st

    def close(self):
        if getattr(self, "_c", None):
            for b in list(getattr(self, "_buffers", ())):   # buffers that outlive the context would leak their HBM
                b.free()
            lib.ibu_ctx_destroy(self._c)
            self._c = None

    __del__ = close


__all__ = ["Header", "Record", "HEADER_SIZE", "MAGIC", "RECORD_SIZE", "VERSION", "IbuError", "load_to_vec",
           "MmapReader", "Reader", "Writer", "ParallelProcessor", "ProcessError", "shard_range", "Context",
           "DeviceBuffer", "DeviceStream", "DeviceBatch", "numa_of_pci", "records_array", "key_plan", "REC_DTYPE", "device_count", "PROC_REDUCE", "PROC_DECODE",
           "DEFAULT_BUFFER_SIZE", "BATCH_SIZE"]
