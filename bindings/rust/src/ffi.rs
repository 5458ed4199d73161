//! Raw `extern "C"` declarations — one-to-one with include/ibu_hip.h (ABI revision 4).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
#[derive(Copy, Clone, Debug, PartialEq, Eq, Hash, bytemuck::Pod, bytemuck::Zeroable)]
pub struct ibu_header_t {
    pub magic: u32,
    pub version: u32,
    pub bc_len: u32,
    pub umi_len: u32,
    pub flags: u64,
    pub reserved: [u8; 8],
}
#[repr(C)]
#[derive(Copy, Clone, Debug, Default, PartialEq, Eq, PartialOrd, Ord, Hash, bytemuck::Pod, bytemuck::Zeroable)]
pub struct ibu_record_t {
    pub barcode: u64,
    pub umi: u64,
    pub index: u64,
}
#[repr(C)]
pub struct ibu_error_detail_t {
    pub code: i32,
    pub os_errno: i32,
    pub a: u64,
    pub b: u64,
    pub message: [c_char; 232],
}
#[repr(C)]
#[derive(Default, Debug, Clone, Copy)]
pub struct ibu_reduce_result_t {
    pub count: u64,
    pub sum: [u64; 3],
    pub xor: [u64; 3],
}
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct ibu_ring_config_t {
    pub slots: u32,
    pub slot_records: u32,
    pub feeder_threads: u32,
    pub reserved: u32,
}
#[repr(C)]
#[derive(Default, Debug, Clone, Copy)]
pub struct ibu_stream_stats_t {
    pub records: u64,
    pub bytes_h2d: u64,
    pub bytes_d2h: u64,
    pub batches: u64,
    pub seconds_total: f64,
    pub seconds_kernel: f64,
    pub numa_node: i32,
    pub ring_node: i32,
}
/// `ibu_ctx_numa`: where the context's device hangs off the host and where its pinned ring landed.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct ibu_numa_info_t {
    pub mode: i32,
    pub node: i32,
    pub usable_cpus: i32,
    pub ring_node: i32,
    pub ring_placed: i32,
    pub reserved: i32,
    pub pci_bus_id: [c_char; 32],
    pub cpulist: [c_char; 256],
}
#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct ibu_alloc_probe_t {
    pub tries: u32,
    pub chosen: u32,
    pub ms: [f32; 16],
}

/// One BGZF block as `ibu_bgzf_scan` describes it and `ibu_inflate_blocks_device` inflates it.
#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct ibu_inflate_block_t {
    pub comp_offset: u64,
    pub out_offset: i64,
    pub comp_len: u32,
    pub out_len: u32,
    pub crc32: u32,
    pub reserved: u32,
}

/// One shard of `ibu_sort_records_contexts`: `n` records in `d_records` (room for `capacity`), `d_tmp` = capacity * 24 bytes.
#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct ibu_sort_shard_t {
    pub d_records: *mut c_void,
    pub d_tmp: *mut c_void,
    pub n: usize,
    pub capacity: usize,
}

#[repr(C)]
pub struct ibu_decode_sink_t {
    pub d_bc_ascii: *mut u8,
    pub d_umi_ascii: *mut u8,
    pub d_index: *mut u64,
    /// rows every non-NULL column can hold (ABI revision 4): a longer stream is IBU_ERR_INVALID_ARG, never an overrun
    pub cap_records: usize,
}
/// Plan of the compacted keys (the varying bytes of a set of records as 12-byte elements): ibu_key_plan_init.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct ibu_key_plan_t {
    pub csel: [[u32; 3]; 4],
    pub xsel: [[u32; 2]; 6],
    pub k: u32,
    pub index_bytes: u32,
    pub base: [u64; 3],
}
#[repr(C)]
pub struct ibu_processor_vtable_t {
    pub clone: Option<unsafe extern "C" fn(*mut c_void) -> *mut c_void>,
    pub drop: Option<unsafe extern "C" fn(*mut c_void)>,
    pub process_record: Option<unsafe extern "C" fn(*mut c_void, *const ibu_record_t) -> i32>,
    pub on_batch_complete: Option<unsafe extern "C" fn(*mut c_void) -> i32>,
    pub set_tid: Option<unsafe extern "C" fn(*mut c_void, usize)>,
}
pub enum ibu_writer_t {}
pub enum ibu_reader_t {}
pub enum ibu_mmap_t {}
pub enum ibu_ctx_t {}
pub enum ibu_stream_t {}
pub type ibu_write_fn = unsafe extern "C" fn(*mut c_void, *const u8, usize) -> i32;
pub type ibu_flush_fn = unsafe extern "C" fn(*mut c_void) -> i32;
pub type ibu_read_fn = unsafe extern "C" fn(*mut c_void, *mut u8, usize, *mut usize) -> i32;

extern "C" {
    pub fn ibu_last_error(out: *mut ibu_error_detail_t);
    pub fn ibu_status_name(status: i32) -> *const c_char;
    pub fn ibu_version() -> *const c_char;
    pub fn ibu_abi_revision() -> u32;
    pub fn ibu_free(p: *mut c_void);
    pub fn ibu_header_init(h: *mut ibu_header_t, bc_len: u32, umi_len: u32);
    pub fn ibu_header_set_sorted(h: *mut ibu_header_t);
    pub fn ibu_header_sorted(h: *const ibu_header_t) -> i32;
    pub fn ibu_header_validate(h: *const ibu_header_t) -> i32;
    pub fn ibu_header_from_bytes(bytes: *const u8, len: usize, out: *mut ibu_header_t) -> i32;
    pub fn ibu_header_as_bytes(h: *const ibu_header_t, out: *mut u8, cap: usize) -> i32;
    pub fn ibu_record_from_bytes(bytes: *const u8, len: usize, out: *mut ibu_record_t) -> i32;
    pub fn ibu_record_as_bytes(r: *const ibu_record_t, out: *mut u8, cap: usize) -> i32;
    pub fn ibu_record_cmp(a: *const ibu_record_t, b: *const ibu_record_t) -> i32;
    pub fn ibu_writer_open_callback(wr: ibu_write_fn, fl: Option<ibu_flush_fn>, user: *mut c_void,
                                    header: *const ibu_header_t, out: *mut *mut ibu_writer_t) -> i32;
    pub fn ibu_writer_open_path(path: *const c_char, header: *const ibu_header_t, out: *mut *mut ibu_writer_t) -> i32;
    pub fn ibu_writer_open_fd(fd: c_int, header: *const ibu_header_t, out: *mut *mut ibu_writer_t) -> i32;
    pub fn ibu_writer_open_mem(header: *const ibu_header_t, out: *mut *mut ibu_writer_t) -> i32;
    pub fn ibu_writer_write_record(w: *mut ibu_writer_t, r: *const ibu_record_t) -> i32;
    pub fn ibu_writer_write_batch(w: *mut ibu_writer_t, recs: *const ibu_record_t, n: usize) -> i32;
    pub fn ibu_writer_ingest(w: *mut ibu_writer_t, other_mem: *mut ibu_writer_t) -> i32;
    pub fn ibu_writer_finish(w: *mut ibu_writer_t) -> i32;
    pub fn ibu_writer_records_written(w: *const ibu_writer_t) -> u64;
    pub fn ibu_writer_mem_view(w: *const ibu_writer_t, data: *mut *const u8, len: *mut usize) -> i32;
    pub fn ibu_writer_into_inner(w: *mut ibu_writer_t, data: *mut *mut u8, len: *mut usize) -> i32;
    pub fn ibu_writer_close(w: *mut ibu_writer_t);
    pub fn ibu_reader_open_callback(rd: ibu_read_fn, user: *mut c_void, out: *mut *mut ibu_reader_t) -> i32;
    pub fn ibu_reader_open_mem(data: *const u8, len: usize, out: *mut *mut ibu_reader_t) -> i32;
    pub fn ibu_reader_open_path(path: *const c_char, out: *mut *mut ibu_reader_t) -> i32;
    pub fn ibu_reader_open_fd(fd: c_int, out: *mut *mut ibu_reader_t) -> i32;
    pub fn ibu_reader_header(r: *const ibu_reader_t, out: *mut ibu_header_t) -> i32;
    pub fn ibu_reader_read_batch(r: *mut ibu_reader_t, has_data: *mut i32) -> i32;
    pub fn ibu_reader_next(r: *mut ibu_reader_t, out: *mut ibu_record_t, got: *mut i32) -> i32;
    pub fn ibu_reader_buffered(r: *mut ibu_reader_t, recs: *mut *const ibu_record_t, n: *mut usize) -> i32;
    pub fn ibu_reader_consume(r: *mut ibu_reader_t, n: usize) -> i32;
    pub fn ibu_reader_bytes_read(r: *const ibu_reader_t) -> u64;
    pub fn ibu_reader_close(r: *mut ibu_reader_t);
    pub fn ibu_load_to_vec(path: *const c_char, header: *mut ibu_header_t, records: *mut *mut ibu_record_t,
                           n: *mut usize) -> i32;
    pub fn ibu_mmap_open(path: *const c_char, out: *mut *mut ibu_mmap_t) -> i32;
    pub fn ibu_mmap_clone(m: *mut ibu_mmap_t, out: *mut *mut ibu_mmap_t) -> i32;
    pub fn ibu_mmap_len(m: *const ibu_mmap_t) -> usize;
    pub fn ibu_mmap_header(m: *const ibu_mmap_t, out: *mut ibu_header_t) -> i32;
    pub fn ibu_mmap_slice(m: *const ibu_mmap_t, start: usize, end: usize, recs: *mut *const ibu_record_t,
                          n: *mut usize) -> i32;
    pub fn ibu_mmap_base(m: *const ibu_mmap_t) -> *const c_void;
    pub fn ibu_mmap_close(m: *mut ibu_mmap_t);
    pub fn ibu_shard_range(len: usize, n_shards: usize, shard: usize, start: *mut usize, end: *mut usize) -> i32;
    pub fn ibu_mmap_process_parallel(m: *const ibu_mmap_t, vt: *const ibu_processor_vtable_t, user: *mut c_void,
                                     num_threads: usize) -> i32;
    pub fn ibu_ctx_create(device: i32, out: *mut *mut ibu_ctx_t) -> i32;
    pub fn ibu_ctx_destroy(ctx: *mut ibu_ctx_t);
    pub fn ibu_ctx_device(ctx: *const ibu_ctx_t) -> i32;
    pub fn ibu_ctx_stream(ctx: *const ibu_ctx_t) -> *mut c_void;
    pub fn ibu_ctx_synchronize(ctx: *mut ibu_ctx_t, stream: *mut c_void) -> i32;
    pub fn ibu_device_count(n: *mut i32) -> i32;
    pub fn ibu_ctx_set_option(ctx: *mut ibu_ctx_t, key: *const c_char, value: i64) -> i32;
    pub fn ibu_device_copy(ctx: *mut ibu_ctx_t, d_dst: *mut c_void, d_src: *const c_void, bytes: usize,
                           stream: *mut c_void) -> i32;
    pub fn ibu_barcode_counts(ctx: *mut ibu_ctx_t, d_sorted_records: *const c_void, n: usize, d_barcodes: *mut u64,
                              d_counts: *mut u64, d_unique_umis: *mut u64, cap: usize, n_barcodes: *mut usize,
                              n_barcode_umi_pairs: *mut usize, stream: *mut c_void) -> i32;
    pub fn ibu_bgzf_scan(buf: *const u8, len: usize, is_final: i32, blocks: *mut ibu_inflate_block_t, cap: usize, n_blocks: *mut usize,
                         consumed: *mut usize, out_bytes: *mut u64) -> i32;
    pub fn ibu_inflate_blocks_device(ctx: *mut ibu_ctx_t, d_comp: *const c_void, d_blocks: *const ibu_inflate_block_t, n: usize,
                                     d_out: *mut c_void, d_status: *mut u32, d_first_bad: *mut u32, stream: *mut c_void) -> i32;
    pub fn ibu_mmap_decode_to_host(m: *const ibu_mmap_t, ctx: *mut ibu_ctx_t, cfg: *const ibu_ring_config_t, shard: usize,
                                   n_shards: usize, h_bc_ascii: *mut u8, h_umi_ascii: *mut u8, h_index: *mut u64,
                                   stats: *mut ibu_stream_stats_t) -> i32;
    pub fn ibu_writer_write_ascii_batch(w: *mut ibu_writer_t, ctx: *mut ibu_ctx_t, cfg: *const ibu_ring_config_t,
                                        h_bc_ascii: *const u8, h_umi_ascii: *const u8, h_index: *const u64, first_index: u64,
                                        n: usize, bc_len: u32, umi_len: u32, stats: *mut ibu_stream_stats_t) -> i32;
    pub fn ibu_device_alloc(ctx: *mut ibu_ctx_t, bytes: usize, d_ptr: *mut *mut c_void) -> i32;
    pub fn ibu_device_alloc_probed(ctx: *mut ibu_ctx_t, bytes: usize, tries: u32, d_ptr: *mut *mut c_void,
                                   report: *mut ibu_alloc_probe_t) -> i32;
    pub fn ibu_device_free(ctx: *mut ibu_ctx_t, d_ptr: *mut c_void) -> i32;
    pub fn ibu_memcpy_h2d(ctx: *mut ibu_ctx_t, d_dst: *mut c_void, h_src: *const c_void, bytes: usize,
                          stream: *mut c_void) -> i32;
    pub fn ibu_memcpy_d2h(ctx: *mut ibu_ctx_t, h_dst: *mut c_void, d_src: *const c_void, bytes: usize,
                          stream: *mut c_void) -> i32;
    pub fn ibu_deserialize(ctx: *mut ibu_ctx_t, d_records: *const c_void, n: usize, d_barcode: *mut u64,
                           d_umi: *mut u64, d_index: *mut u64, stream: *mut c_void) -> i32;
    pub fn ibu_serialize(ctx: *mut ibu_ctx_t, d_barcode: *const u64, d_umi: *const u64, d_index: *const u64, n: usize,
                         d_records: *mut c_void, stream: *mut c_void) -> i32;
    pub fn ibu_unpack_2bit(ctx: *mut ibu_ctx_t, d_codes: *const u64, n: usize, len: u32, d_ascii: *mut u8,
                           stream: *mut c_void) -> i32;
    pub fn ibu_pack_2bit(ctx: *mut ibu_ctx_t, d_ascii: *const u8, n: usize, len: u32, d_codes: *mut u64,
                         stream: *mut c_void) -> i32;
    pub fn ibu_decode_ascii(ctx: *mut ibu_ctx_t, d_records: *const c_void, n: usize, bc_len: u32, umi_len: u32,
                            d_bc_ascii: *mut u8, d_umi_ascii: *mut u8, d_index: *mut u64, stream: *mut c_void) -> i32;
    pub fn ibu_encode_ascii(ctx: *mut ibu_ctx_t, d_bc_ascii: *const u8, d_umi_ascii: *const u8, d_index: *const u64,
                            first_index: u64, n: usize, bc_len: u32, umi_len: u32, d_records: *mut c_void,
                            stream: *mut c_void) -> i32;
    pub fn ibu_codec_status(ctx: *mut ibu_ctx_t, stream: *mut c_void, first_bad_record: *mut u64,
                            n_bad_records: *mut u64) -> i32;
    pub fn ibu_reduce_reset(ctx: *mut ibu_ctx_t, stream: *mut c_void) -> i32;
    pub fn ibu_reduce(ctx: *mut ibu_ctx_t, d_records: *const c_void, n: usize, stream: *mut c_void) -> i32;
    pub fn ibu_reduce_fetch(ctx: *mut ibu_ctx_t, stream: *mut c_void, out: *mut ibu_reduce_result_t) -> i32;
    pub fn ibu_generate(ctx: *mut ibu_ctx_t, seed: u64, first: u64, n: usize, bc_len: u32, umi_len: u32,
                        d_records: *mut c_void, stream: *mut c_void) -> i32;
    pub fn ibu_sort_records(ctx: *mut ibu_ctx_t, d_records: *mut c_void, d_tmp: *mut c_void, n: usize,
                            stream: *mut c_void) -> i32;
    pub fn ibu_sort_records_contexts(ctxs: *const *mut ibu_ctx_t, n_ctxs: usize, shards: *mut ibu_sort_shard_t) -> i32;
    pub fn ibu_lower_bound_records(ctx: *mut ibu_ctx_t, d_sorted_records: *const c_void, n: usize, d_keys: *const c_void, k: usize,
                                   d_pos: *mut u64, stream: *mut c_void) -> i32;
    pub fn ibu_records_first_mismatch(ctx: *mut ibu_ctx_t, d_a: *const c_void, d_b: *const c_void, n: usize, first: *mut u64,
                                      stream: *mut c_void) -> i32;
    pub fn ibu_records_census(ctx: *mut ibu_ctx_t, d_records: *const c_void, n: usize, out: *mut u64, stream: *mut c_void) -> i32;
    pub fn ibu_key_plan_init(or_words: *const u64, and_words: *const u64, plan: *mut ibu_key_plan_t) -> i32;
    pub fn ibu_records_compact(ctx: *mut ibu_ctx_t, plan: *const ibu_key_plan_t, d_records: *const c_void, n: usize,
                               d_elems: *mut c_void, stream: *mut c_void) -> i32;
    pub fn ibu_records_expand(ctx: *mut ibu_ctx_t, plan: *const ibu_key_plan_t, d_elems: *const c_void, n: usize,
                              d_records: *mut c_void, stream: *mut c_void) -> i32;
    pub fn ibu_is_sorted(ctx: *mut ibu_ctx_t, d_records: *const c_void, n: usize, stream: *mut c_void,
                         sorted: *mut i32) -> i32;
    pub fn ibu_load_to_device(ctx: *mut ibu_ctx_t, path: *const c_char, cfg: *const ibu_ring_config_t,
                              header: *mut ibu_header_t, d_records: *mut *mut c_void, cap_records: usize,
                              n: *mut usize, stats: *mut ibu_stream_stats_t) -> i32;
    pub fn ibu_load_bgzf_to_device(ctx: *mut ibu_ctx_t, path: *const c_char, cfg: *const ibu_ring_config_t,
                              header: *mut ibu_header_t, d_records: *mut *mut c_void, cap_records: usize,
                              n: *mut usize, stats: *mut ibu_stream_stats_t) -> i32;
    pub fn ibu_load_bgzf_shard_to_device(ctx: *mut ibu_ctx_t, path: *const c_char, cfg: *const ibu_ring_config_t, shard: usize, n_shards: usize,
                                         header: *mut ibu_header_t, d_records: *mut *mut c_void, cap_records: usize, n: *mut usize,
                                         first_record: *mut u64, stats: *mut ibu_stream_stats_t) -> i32;
    pub fn ibu_writer_write_batch_device(w: *mut ibu_writer_t, ctx: *mut ibu_ctx_t, cfg: *const ibu_ring_config_t,
                                         d_records: *const c_void, n: usize, stats: *mut ibu_stream_stats_t) -> i32;
    pub fn ibu_writer_write_batch_device_on(w: *mut ibu_writer_t, ctx: *mut ibu_ctx_t, cfg: *const ibu_ring_config_t,
                                            d_records: *const c_void, n: usize, producer_stream: *mut c_void,
                                            stats: *mut ibu_stream_stats_t) -> i32;
    pub fn ibu_mmap_process_device(m: *const ibu_mmap_t, ctx: *mut ibu_ctx_t, cfg: *const ibu_ring_config_t,
                                   proc_: i32, shard: usize, n_shards: usize, sink: *mut c_void,
                                   stats: *mut ibu_stream_stats_t) -> i32;
    pub fn ibu_mmap_process_devices(m: *const ibu_mmap_t, devices: *const i32, n_devices: usize, cfg: *const ibu_ring_config_t,
                                    proc_: i32, sinks: *mut c_void, total: *mut ibu_reduce_result_t,
                                    stats: *mut ibu_stream_stats_t) -> i32;
    pub fn ibu_mmap_process_contexts(m: *const ibu_mmap_t, ctxs: *const *mut ibu_ctx_t, n_ctxs: usize,
                                     cfg: *const ibu_ring_config_t, proc_: i32, sinks: *mut c_void,
                                     total: *mut ibu_reduce_result_t, stats: *mut ibu_stream_stats_t) -> i32;
    pub fn ibu_reader_process_device(r: *mut ibu_reader_t, ctx: *mut ibu_ctx_t, cfg: *const ibu_ring_config_t,
                                     proc_: i32, sink: *mut c_void, stats: *mut ibu_stream_stats_t) -> i32;
    pub fn ibu_ctx_numa(ctx: *const ibu_ctx_t, out: *mut ibu_numa_info_t) -> i32;
    pub fn ibu_numa_of_pci(sysfs_root: *const c_char, pci_bus_id: *const c_char, node: *mut i32, cpulist: *mut c_char,
                           cap: usize, usable_cpus: *mut i32) -> i32;
    pub fn ibu_stream_open_reader(r: *mut ibu_reader_t, ctx: *mut ibu_ctx_t, cfg: *const ibu_ring_config_t,
                                  out: *mut *mut ibu_stream_t) -> i32;
    pub fn ibu_stream_open_mmap(m: *const ibu_mmap_t, ctx: *mut ibu_ctx_t, cfg: *const ibu_ring_config_t, shard: usize,
                                n_shards: usize, out: *mut *mut ibu_stream_t) -> i32;
    pub fn ibu_stream_header(s: *const ibu_stream_t, out: *mut ibu_header_t) -> i32;
    pub fn ibu_stream_next(s: *mut ibu_stream_t, stream: *mut c_void, d_records: *mut *const c_void, n: *mut usize,
                           first_index: *mut u64) -> i32;
    pub fn ibu_stream_release(s: *mut ibu_stream_t, d_records: *const c_void, stream: *mut c_void) -> i32;
    pub fn ibu_stream_stats(s: *const ibu_stream_t, out: *mut ibu_stream_stats_t) -> i32;
    pub fn ibu_stream_close(s: *mut ibu_stream_t);
}
