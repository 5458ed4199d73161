"""ctypes loader for libibu_hip.so (the C ABI of include/ibu_hip.h).

The shared library is the product; this module only binds it.  If it is missing the import
FAILS — there is no Python or CPU fallback for any kernel entry point.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("IBU_HIP_SO") or os.path.join(_HERE, "libibu_hip.so")  # IBU_HIP_SO: A/B builds of the same ABI

u8p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
vp, sz, u64, u32, i32 = C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_int32


class CHeader(C.Structure):  # ibu_header_t
    _fields_ = [("magic", u32), ("version", u32), ("bc_len", u32), ("umi_len", u32), ("flags", u64),
                ("reserved", C.c_uint8 * 8)]


class CRecord(C.Structure):  # ibu_record_t
    _fields_ = [("barcode", u64), ("umi", u64), ("index", u64)]


class CErrorDetail(C.Structure):  # ibu_error_detail_t
    _fields_ = [("code", i32), ("os_errno", i32), ("a", u64), ("b", u64), ("message", C.c_char * 232)]


class CReduceResult(C.Structure):  # ibu_reduce_result_t
    _fields_ = [("count", u64), ("sum", u64 * 3), ("xor_", u64 * 3)]


class CRingConfig(C.Structure):  # ibu_ring_config_t
    _fields_ = [("slots", u32), ("slot_records", u32), ("feeder_threads", u32), ("reserved", u32)]


class CStreamStats(C.Structure):  # ibu_stream_stats_t
    _fields_ = [("records", u64), ("bytes_h2d", u64), ("bytes_d2h", u64), ("batches", u64),
                ("seconds_total", C.c_double), ("seconds_kernel", C.c_double), ("numa_node", i32), ("ring_node", i32)]


class CNumaInfo(C.Structure):  # ibu_numa_info_t
    _fields_ = [("mode", i32), ("node", i32), ("usable_cpus", i32), ("ring_node", i32), ("ring_placed", i32), ("reserved", i32),
                ("pci_bus_id", C.c_char * 32), ("cpulist", C.c_char * 256)]


class CKeyPlan(C.Structure):  # ibu_key_plan_t
    _fields_ = [("csel", (u32 * 3) * 4), ("xsel", (u32 * 2) * 6), ("k", u32), ("index_bytes", u32), ("base", u64 * 3)]


class CInflateBlock(C.Structure):  # ibu_inflate_block_t
    _fields_ = [("comp_offset", C.c_uint64), ("out_offset", C.c_int64), ("comp_len", C.c_uint32), ("out_len", C.c_uint32),
                ("crc32", C.c_uint32), ("reserved", C.c_uint32)]


class CSortShard(C.Structure):  # ibu_sort_shard_t
    _fields_ = [("d_records", C.c_void_p), ("d_tmp", C.c_void_p), ("n", C.c_size_t), ("capacity", C.c_size_t)]


class CAllocProbe(C.Structure):  # ibu_alloc_probe_t
    _fields_ = [("tries", u32), ("chosen", u32), ("ms", C.c_float * 16)]


class CDecodeSink(C.Structure):  # ibu_decode_sink_t
    _fields_ = [("d_bc_ascii", vp), ("d_umi_ascii", vp), ("d_index", vp), ("cap_records", sz)]


WRITE_FN = C.CFUNCTYPE(i32, vp, u8p, sz)
FLUSH_FN = C.CFUNCTYPE(i32, vp)
READ_FN = C.CFUNCTYPE(i32, vp, u8p, sz, C.POINTER(sz))
CLONE_FN = C.CFUNCTYPE(vp, vp)
DROP_FN = C.CFUNCTYPE(None, vp)
PROCESS_FN = C.CFUNCTYPE(i32, vp, C.POINTER(CRecord))
BATCH_FN = C.CFUNCTYPE(i32, vp)
TID_FN = C.CFUNCTYPE(None, vp, sz)


class CProcessorVTable(C.Structure):  # ibu_processor_vtable_t
    _fields_ = [("clone", CLONE_FN), ("drop", DROP_FN), ("process_record", PROCESS_FN),
                ("on_batch_complete", BATCH_FN), ("set_tid", TID_FN)]


P = C.POINTER
# name -> (restype, argtypes); must list every function include/ibu_hip.h declares
# (tests/test_abi_symbols.py compares this table and the .so against the header).
SIGNATURES = {
    "ibu_last_error": (None, [P(CErrorDetail)]),
    "ibu_status_name": (C.c_char_p, [i32]),
    "ibu_version": (C.c_char_p, []),
    "ibu_abi_revision": (u32, []),
    "ibu_free": (None, [vp]),
    "ibu_header_init": (None, [P(CHeader), u32, u32]),
    "ibu_header_set_sorted": (None, [P(CHeader)]),
    "ibu_header_sorted": (i32, [P(CHeader)]),
    "ibu_header_validate": (i32, [P(CHeader)]),
    "ibu_header_from_bytes": (i32, [C.c_char_p, sz, P(CHeader)]),
    "ibu_header_as_bytes": (i32, [P(CHeader), C.c_char_p, sz]),
    "ibu_record_from_bytes": (i32, [C.c_char_p, sz, P(CRecord)]),
    "ibu_record_as_bytes": (i32, [P(CRecord), C.c_char_p, sz]),
    "ibu_record_cmp": (i32, [P(CRecord), P(CRecord)]),
    "ibu_writer_open_callback": (i32, [WRITE_FN, FLUSH_FN, vp, P(CHeader), P(vp)]),
    "ibu_writer_open_path": (i32, [C.c_char_p, P(CHeader), P(vp)]),
    "ibu_writer_open_fd": (i32, [C.c_int, P(CHeader), P(vp)]),
    "ibu_writer_open_mem": (i32, [P(CHeader), P(vp)]),
    "ibu_writer_write_record": (i32, [vp, P(CRecord)]),
    "ibu_writer_write_batch": (i32, [vp, vp, sz]),
    "ibu_writer_ingest": (i32, [vp, vp]),
    "ibu_writer_finish": (i32, [vp]),
    "ibu_writer_records_written": (u64, [vp]),
    "ibu_writer_mem_view": (i32, [vp, P(vp), P(sz)]),
    "ibu_writer_into_inner": (i32, [vp, P(vp), P(sz)]),
    "ibu_writer_close": (None, [vp]),
    "ibu_reader_open_callback": (i32, [READ_FN, vp, P(vp)]),
    "ibu_reader_open_mem": (i32, [vp, sz, P(vp)]),
    "ibu_reader_open_path": (i32, [C.c_char_p, P(vp)]),
    "ibu_reader_open_fd": (i32, [C.c_int, P(vp)]),
    "ibu_reader_header": (i32, [vp, P(CHeader)]),
    "ibu_reader_read_batch": (i32, [vp, P(i32)]),
    "ibu_reader_next": (i32, [vp, P(CRecord), P(i32)]),
    "ibu_reader_buffered": (i32, [vp, P(vp), P(sz)]),
    "ibu_reader_consume": (i32, [vp, sz]),
    "ibu_reader_bytes_read": (u64, [vp]),
    "ibu_reader_close": (None, [vp]),
    "ibu_load_to_vec": (i32, [C.c_char_p, P(CHeader), P(vp), P(sz)]),
    "ibu_mmap_open": (i32, [C.c_char_p, P(vp)]),
    "ibu_mmap_clone": (i32, [vp, P(vp)]),
    "ibu_mmap_len": (sz, [vp]),
    "ibu_mmap_header": (i32, [vp, P(CHeader)]),
    "ibu_mmap_slice": (i32, [vp, sz, sz, P(vp), P(sz)]),
    "ibu_mmap_base": (vp, [vp]),
    "ibu_mmap_close": (None, [vp]),
    "ibu_shard_range": (i32, [sz, sz, sz, P(sz), P(sz)]),
    "ibu_mmap_process_parallel": (i32, [vp, P(CProcessorVTable), vp, sz]),
    "ibu_ctx_create": (i32, [i32, P(vp)]),
    "ibu_ctx_destroy": (None, [vp]),
    "ibu_ctx_device": (i32, [vp]),
    "ibu_ctx_stream": (vp, [vp]),
    "ibu_ctx_synchronize": (i32, [vp, vp]),
    "ibu_device_count": (i32, [P(i32)]),
    "ibu_ctx_set_option": (i32, [vp, C.c_char_p, C.c_int64]),
    "ibu_device_copy": (i32, [vp, vp, vp, sz, vp]),
    "ibu_mmap_decode_to_host": (i32, [vp, vp, vp, sz, sz, vp, vp, vp, vp]),
    "ibu_writer_write_ascii_batch": (i32, [vp, vp, vp, vp, vp, vp, u64, sz, u32, u32, vp]),
    "ibu_barcode_counts": (i32, [vp, vp, sz, vp, vp, vp, sz, P(sz), P(sz), vp]),
    "ibu_bgzf_scan": (i32, [vp, sz, i32, P(CInflateBlock), sz, P(sz), P(sz), P(C.c_uint64)]),
    "ibu_inflate_blocks_device": (i32, [vp, vp, vp, sz, vp, vp, vp, vp]),
    "ibu_device_alloc": (i32, [vp, sz, P(vp)]),
    "ibu_device_alloc_probed": (i32, [vp, sz, u32, P(vp), P(CAllocProbe)]),
    "ibu_device_free": (i32, [vp, vp]),
    "ibu_memcpy_h2d": (i32, [vp, vp, vp, sz, vp]),
    "ibu_memcpy_d2h": (i32, [vp, vp, vp, sz, vp]),
    "ibu_deserialize": (i32, [vp, vp, sz, vp, vp, vp, vp]),
    "ibu_serialize": (i32, [vp, vp, vp, vp, sz, vp, vp]),
    "ibu_unpack_2bit": (i32, [vp, vp, sz, u32, vp, vp]),
    "ibu_pack_2bit": (i32, [vp, vp, sz, u32, vp, vp]),
    "ibu_decode_ascii": (i32, [vp, vp, sz, u32, u32, vp, vp, vp, vp]),
    "ibu_encode_ascii": (i32, [vp, vp, vp, vp, u64, sz, u32, u32, vp, vp]),
    "ibu_codec_status": (i32, [vp, vp, P(u64), P(u64)]),
    "ibu_reduce_reset": (i32, [vp, vp]),
    "ibu_reduce": (i32, [vp, vp, sz, vp]),
    "ibu_reduce_fetch": (i32, [vp, vp, P(CReduceResult)]),
    "ibu_generate": (i32, [vp, u64, u64, sz, u32, u32, vp, vp]),
    "ibu_sort_records": (i32, [vp, vp, vp, sz, vp]),
    "ibu_lower_bound_records": (i32, [vp, vp, sz, vp, sz, vp, vp]),
    "ibu_is_sorted": (i32, [vp, vp, sz, vp, P(i32)]),
    "ibu_records_first_mismatch": (i32, [vp, vp, vp, sz, P(u64), vp]),
    "ibu_records_census": (i32, [vp, vp, sz, P(u64), vp]),
    "ibu_key_plan_init": (i32, [P(u64), P(u64), P(CKeyPlan)]),
    "ibu_records_compact": (i32, [vp, P(CKeyPlan), vp, sz, vp, vp]),
    "ibu_records_expand": (i32, [vp, P(CKeyPlan), vp, sz, vp, vp]),
    "ibu_load_to_device": (i32, [vp, C.c_char_p, P(CRingConfig), P(CHeader), P(vp), sz, P(sz), P(CStreamStats)]),
    "ibu_load_bgzf_to_device": (i32, [vp, C.c_char_p, P(CRingConfig), P(CHeader), P(vp), sz, P(sz), P(CStreamStats)]),
    "ibu_load_bgzf_shard_to_device": (i32, [vp, C.c_char_p, P(CRingConfig), sz, sz, P(CHeader), P(vp), sz, P(sz), P(C.c_uint64), P(CStreamStats)]),
    "ibu_writer_write_batch_device": (i32, [vp, vp, P(CRingConfig), vp, sz, P(CStreamStats)]),
    "ibu_writer_write_batch_device_on": (i32, [vp, vp, P(CRingConfig), vp, sz, vp, P(CStreamStats)]),
    "ibu_mmap_process_device": (i32, [vp, vp, P(CRingConfig), i32, sz, sz, vp, P(CStreamStats)]),
    "ibu_mmap_process_devices": (i32, [vp, P(i32), sz, P(CRingConfig), i32, vp, P(CReduceResult), P(CStreamStats)]),
    "ibu_sort_records_contexts": (i32, [P(vp), sz, P(CSortShard)]),
    "ibu_mmap_process_contexts": (i32, [vp, P(vp), sz, P(CRingConfig), i32, vp, P(CReduceResult), P(CStreamStats)]),
    "ibu_reader_process_device": (i32, [vp, vp, P(CRingConfig), i32, vp, P(CStreamStats)]),
    "ibu_ctx_numa": (i32, [vp, P(CNumaInfo)]),
    "ibu_numa_of_pci": (i32, [C.c_char_p, C.c_char_p, P(i32), C.c_char_p, sz, P(i32)]),
    "ibu_stream_open_reader": (i32, [vp, vp, P(CRingConfig), P(vp)]),
    "ibu_stream_open_mmap": (i32, [vp, vp, P(CRingConfig), sz, sz, P(vp)]),
    "ibu_stream_header": (i32, [vp, P(CHeader)]),
    "ibu_stream_next": (i32, [vp, vp, P(vp), P(sz), P(u64)]),
    "ibu_stream_release": (i32, [vp, vp, vp]),
    "ibu_stream_stats": (i32, [vp, P(CStreamStats)]),
    "ibu_stream_close": (None, [vp]),
}


def load(path=SO_PATH):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C ibu_amd/csrc`).  ibu_amd has no fallback for its HIP extension.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    return lib
