/*
 * ibu_hip.h — C ABI of libibu_hip.so, the MI355X (gfx950) batch record-stream + 2-bit codec
 * path for the IBU binary format.
 *
 * This is the drop-in boundary.  The reference (noamteyssier/ibu, Rust) has no FFI of its
 * own; its boundary is the crate's public API (src/lib.rs:178-181).  Every entry point below
 * names the reference item it stands behind, so that a Rust `extern "C"` shim (source in
 * bindings/rust/, walkthrough in INTEGRATION.md) can re-create `Header / Record / Reader /
 * Writer / MmapReader / ParallelProcessor` on top of it.
 *
 * Conventions
 *   - plain C types only: pointers, sizes, fixed-width ints.  No C++/torch types.
 *   - every fallible function returns int32_t: 0 = IBU_OK, 1..10 = the ten IbuError variants
 *     in declaration order (src/error.rs:56-128), 11.. = codec / argument errors that the
 *     reference expresses as panics or leaves to `bitnuc`, >=100 = HIP runtime.
 *   - the payload of the failing call (expected/actual, pos, idx/max, errno, message) is
 *     kept per calling thread and read with ibu_last_error().
 *   - nothing throws or aborts across this boundary.
 *   - host pointers are `h_` / unprefixed; device pointers are prefixed `d_`.  `stream` is a
 *     hipStream_t passed as void* (NULL = the context's own stream).  Launch functions are
 *     asynchronous, allocate nothing and never synchronise: they can be captured into a hipGraph and replayed
 *     (tests/test_gpu_parity.py::test_launch_functions_can_be_captured_into_a_hip_graph).
 *   - the caller owns every buffer it passes; the library owns only what *_open/_create
 *     returned, until the matching *_close/_destroy.
 *
 * 2-bit codec convention (the reference only documents the table, src/constructs/record.rs:19-27;
 * its README defers to the `bitnuc` crate which is not a dependency — parity for the codec is
 * therefore UNPINNED):
 *   A/a=00 C/c=01 G/g=10 T/t=11; base i of a sequence occupies bits [2i, 2i+1] (first base in
 *   the least-significant bits: "ACGT" -> 0b11100100); len in 1..=32; bits >= 2*len are
 *   ignored on unpack and written as zero on pack; unpack emits upper-case ASCII; any other
 *   input byte on pack is IBU_ERR_INVALID_BASE.
 */
#ifndef IBU_HIP_H
#define IBU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants (src/constructs/header.rs:5-7, record.rs:3, io/reader.rs:14, io/mmap.rs:284) */
#define IBU_MAGIC 0x21554249u /* "IBU!" little-endian */
#define IBU_VERSION 2u
#define IBU_HEADER_SIZE 32
#define IBU_RECORD_SIZE 24
#define IBU_DEFAULT_BUFFER_SIZE (48 * 1024 * IBU_RECORD_SIZE) /* 1 179 648 B */
#define IBU_BATCH_SIZE (1024 * 1024)                          /* records per parallel batch */
#define IBU_MAX_SEQ_LEN 32
#define IBU_FLAG_SORTED 1ull

/* ---- POD types, layout-identical to the #[repr(C)] structs ---------------------------- */
typedef struct ibu_header { /* src/constructs/header.rs:48-61 */
  uint32_t magic;
  uint32_t version;
  uint32_t bc_len;
  uint32_t umi_len;
  uint64_t flags;
  uint8_t reserved[8];
} ibu_header_t;

typedef struct ibu_record { /* src/constructs/record.rs:58-66 */
  uint64_t barcode;
  uint64_t umi;
  uint64_t index;
} ibu_record_t;

/* ---- status codes ---------------------------------------------------------------------- */
enum {
  IBU_OK = 0,
  IBU_ERR_IO = 1,               /* IbuError::Io                 error.rs:58  detail.os_errno        */
  IBU_ERR_NIFFLER = 2,          /* IbuError::Niffler            error.rs:62  (decompression)        */
  IBU_ERR_INVALID_MAGIC = 3,    /* InvalidMagicNumber{expected,actual}       detail.a=exp, .b=act   */
  IBU_ERR_TRUNCATED_RECORD = 4, /* TruncatedRecord{pos}                      detail.a=pos           */
  IBU_ERR_INVALID_VERSION = 5,  /* InvalidVersion{expected,actual}           detail.a=exp, .b=act   */
  IBU_ERR_INVALID_BC_LEN = 6,   /* InvalidBarcodeLength(u32)                 detail.a=len           */
  IBU_ERR_INVALID_UMI_LEN = 7,  /* InvalidUmiLength(u32)                     detail.a=len           */
  IBU_ERR_INVALID_MAP_SIZE = 8, /* InvalidMapSize                                                   */
  IBU_ERR_INVALID_INDEX = 9,    /* InvalidIndex{idx,max}                     detail.a=idx, .b=max   */
  IBU_ERR_PROCESS = 10,         /* Process(Box<dyn Error>)   user processor returned non-zero:
                                   detail.a = that value                                            */
  IBU_ERR_INVALID_BASE = 11,    /* codec: byte outside ACGTacgt; detail.a = first bad record,
                                   detail.b = number of offending records                           */
  IBU_ERR_SEQ_LEN = 12,         /* codec: len outside 1..=32; detail.a=len                          */
  IBU_ERR_INVALID_ARG = 13,     /* NULL/size misuse — where the reference would panic
                                   (Header::from_bytes on a wrong length, header.rs:226)            */
  IBU_ERR_HIP = 100,            /* any hipError_t != hipSuccess; detail.a = hipError_t              */
  IBU_ERR_NO_DEVICE = 101       /* no usable gfx950 device: the device path FAILS, never falls back */
};

typedef struct ibu_error_detail {
  int32_t code;
  int32_t os_errno;
  uint64_t a;
  uint64_t b;
  char message[232]; /* Display text in the reference's wording (error.rs:57-127) */
} ibu_error_detail_t;

/* Detail of the last failing call made by the calling thread. */
void ibu_last_error(ibu_error_detail_t* out);
/* Static name of a status code ("InvalidMagicNumber", ...). */
const char* ibu_status_name(int32_t status);
/* Library version string and ABI revision (bumped on any signature change). */
const char* ibu_version(void);
uint32_t ibu_abi_revision(void);
/* Free a buffer returned by ibu_load_to_vec / ibu_writer_into_inner. */
void ibu_free(void* p);

/* ======================================================================================= */
/* A2  Header                                                     src/constructs/header.rs */
/* ======================================================================================= */
void ibu_header_init(ibu_header_t* h, uint32_t bc_len, uint32_t umi_len); /* Header::new :84-93    */
void ibu_header_set_sorted(ibu_header_t* h);                              /* set_sorted  :111-113  */
int32_t ibu_header_sorted(const ibu_header_t* h);                         /* sorted      :130-132  */
int32_t ibu_header_validate(const ibu_header_t* h);                       /* validate    :167-187  */
int32_t ibu_header_from_bytes(const uint8_t* bytes, size_t len, ibu_header_t* out); /* :226-228    */
int32_t ibu_header_as_bytes(const ibu_header_t* h, uint8_t* out, size_t cap);       /* :203-205    */

/* ======================================================================================= */
/* A1  Record                                                     src/constructs/record.rs */
/* ======================================================================================= */
int32_t ibu_record_from_bytes(const uint8_t* bytes, size_t len, ibu_record_t* out); /* :130-132    */
int32_t ibu_record_as_bytes(const ibu_record_t* r, uint8_t* out, size_t cap);       /* :108-110    */
/* derive(Ord): lexicographic (barcode, umi, index) -> -1/0/+1                         :58         */
int32_t ibu_record_cmp(const ibu_record_t* a, const ibu_record_t* b);

/* ======================================================================================= */
/* A3/A4  Writer<W>                                                    src/io/writer.rs     */
/* ======================================================================================= */
typedef struct ibu_writer ibu_writer_t;
/* W: Write as two callbacks.  write must consume all `len` bytes or return non-zero (errno). */
typedef int32_t (*ibu_write_fn)(void* user, const uint8_t* data, size_t len);
typedef int32_t (*ibu_flush_fn)(void* user);

/* Writer::new(inner, header) :129-143 — header written immediately, NOT validated (quirk Q2).
 * header == NULL gives Writer::new_headless :169-179. */
int32_t ibu_writer_open_callback(ibu_write_fn wr, ibu_flush_fn fl, void* user,
                                 const ibu_header_t* header, ibu_writer_t** out);
int32_t ibu_writer_open_path(const char* path, const ibu_header_t* header, ibu_writer_t** out); /* from_path :556-559 */
int32_t ibu_writer_open_fd(int fd, const ibu_header_t* header, ibu_writer_t** out); /* from_stdout :587-589 (fd 1) */
int32_t ibu_writer_open_mem(const ibu_header_t* header, ibu_writer_t** out);        /* Writer<Vec<u8>>            */

int32_t ibu_writer_write_record(ibu_writer_t* w, const ibu_record_t* r);                /* :260-273 */
int32_t ibu_writer_write_batch(ibu_writer_t* w, const ibu_record_t* recs, size_t n);    /* :315-351 */
int32_t ibu_writer_ingest(ibu_writer_t* w, ibu_writer_t* other_mem);                    /* :477-482 */
int32_t ibu_writer_finish(ibu_writer_t* w);                                             /* :429-433 */
uint64_t ibu_writer_records_written(const ibu_writer_t* w);                             /* :207-209 */
/* Bytes the inner sink has received so far (what `into_inner()` of a Vec<u8> writer would hold). */
int32_t ibu_writer_mem_view(const ibu_writer_t* w, const uint8_t** data, size_t* len);
/* into_inner :507-511 — NO flush (ManuallyDrop skips Drop); frees the writer; mem writers hand
 * their bytes to the caller (release with ibu_free), other sinks return data=NULL. */
int32_t ibu_writer_into_inner(ibu_writer_t* w, uint8_t** data, size_t* len);
/* Drop :519-523 — finish().ok(), errors swallowed, then free. */
void ibu_writer_close(ibu_writer_t* w);

/* ======================================================================================= */
/* A5  Reader<R>                                                       src/io/reader.rs     */
/* ======================================================================================= */
typedef struct ibu_reader ibu_reader_t;
/* R: Read as one callback: fill up to cap bytes, *got = bytes read (0 = EOF); non-zero = errno. */
typedef int32_t (*ibu_read_fn)(void* user, uint8_t* dst, size_t cap, size_t* got);

/* Reader::new :152-176 — read_exact 32 B, validate. */
int32_t ibu_reader_open_callback(ibu_read_fn rd, void* user, ibu_reader_t** out);
/* Reader::new(Cursor<&[u8]>) — bytes are borrowed, not copied. */
int32_t ibu_reader_open_mem(const uint8_t* data, size_t len, ibu_reader_t** out);
/* Reader::from_path :345-352 — sniffs the format by magic bytes (niffler's role): gzip (BGZF input is inflated
 * block-parallel), bzip2, xz and zstd are decoded (the last three through the host's libbz2 / liblzma / libzstd,
 * bound at run time; IBU_ERR_NIFFLER if the library is absent); plain files pass through. */
int32_t ibu_reader_open_path(const char* path, ibu_reader_t** out);
int32_t ibu_reader_open_fd(int fd, ibu_reader_t** out); /* from_stdin :389-396 (fd 0), sniffs too */

int32_t ibu_reader_header(const ibu_reader_t* r, ibu_header_t* out);   /* header()    :244-246 */
int32_t ibu_reader_read_batch(ibu_reader_t* r, int32_t* has_data);     /* read_batch  :218-242 */
/* Iterator::next :279-306 — *got = 1 and *out filled, or *got = 0 (None); error = Some(Err). */
int32_t ibu_reader_next(ibu_reader_t* r, ibu_record_t* out, int32_t* got);
/* Zero-copy view of the records left in the current buffer and how many to mark consumed
 * (feeds the device ring; no reference equivalent — Reader keeps `buffer` private). */
int32_t ibu_reader_buffered(ibu_reader_t* r, const ibu_record_t** recs, size_t* n);
int32_t ibu_reader_consume(ibu_reader_t* r, size_t n);
uint64_t ibu_reader_bytes_read(const ibu_reader_t* r);                 /* field :107-108        */
void ibu_reader_close(ibu_reader_t* r);

/* A6  load_to_vec :510-535 — uncompressed files only (quirk Q10); *records from ibu_free. */
int32_t ibu_load_to_vec(const char* path, ibu_header_t* header, ibu_record_t** records, size_t* n);

/* ======================================================================================= */
/* A7/A8  MmapReader + ParallelReader                                  src/io/mmap.rs       */
/* ======================================================================================= */
typedef struct ibu_mmap ibu_mmap_t;
int32_t ibu_mmap_open(const char* path, ibu_mmap_t** out);              /* MmapReader::new :143-161 */
int32_t ibu_mmap_clone(ibu_mmap_t* m, ibu_mmap_t** out);                /* Clone (Arc<Mmap>) :99    */
size_t ibu_mmap_len(const ibu_mmap_t* m);                               /* len    :178-180          */
int32_t ibu_mmap_header(const ibu_mmap_t* m, ibu_header_t* out);        /* header :201-203          */
/* slice :253-270 — InvalidIndex{idx:end,max:len} if start>=len || end>len || end<=start (Q7). */
int32_t ibu_mmap_slice(const ibu_mmap_t* m, size_t start, size_t end, const ibu_record_t** recs,
                       size_t* n);
const void* ibu_mmap_base(const ibu_mmap_t* m); /* Arc::ptr_eq analogue, mmap.rs:541 */
void ibu_mmap_close(ibu_mmap_t* m);

/* Static contiguous range split of mmap.rs:297-307: per = len/n, remainder to the LAST shard.
 * Used for OS threads by the reference and for GPUs / ranks by this library. */
int32_t ibu_shard_range(size_t len, size_t n_shards, size_t shard, size_t* start, size_t* end);

/* ParallelProcessor :100-190 as a C vtable.  `clone` is P: Clone (one copy per worker);
 * process_record / on_batch_complete return 0 or a non-zero user code -> IBU_ERR_PROCESS.
 * set_tid is part of the trait but is never called by process_parallel (quirk Q4). */
typedef struct ibu_processor_vtable {
  void* (*clone)(void* user);
  void (*drop)(void* user_clone);
  int32_t (*process_record)(void* user_clone, const ibu_record_t* rec);
  int32_t (*on_batch_complete)(void* user_clone); /* may be NULL: default Ok(()) */
  void (*set_tid)(void* user_clone, size_t tid);  /* may be NULL */
} ibu_processor_vtable_t;

/* process_parallel :286-332 with a HOST processor (user code, exactly the reference's
 * contract: n = 0 -> all cores, else min(n, cores); 1Mi-record batches; join in spawn order,
 * first Err wins — quirk Q12). */
int32_t ibu_mmap_process_parallel(const ibu_mmap_t* m, const ibu_processor_vtable_t* vt, void* user,
                                  size_t num_threads);

/* ======================================================================================= */
/* Device context                                                                           */
/* ======================================================================================= */
typedef struct ibu_ctx ibu_ctx_t;
/* One context per host thread and device; owns a stream, a status slot and reduce scratch.
 * Fails with IBU_ERR_NO_DEVICE when the device is absent or not gfx950. */
int32_t ibu_ctx_create(int32_t device, ibu_ctx_t** out);
void ibu_ctx_destroy(ibu_ctx_t* ctx);
int32_t ibu_ctx_device(const ibu_ctx_t* ctx);
/* Where the context's device hangs off the host and where its pinned ring landed (option "numa").  The reference's workers are
 * OS threads over one shared map (mmap.rs:308-322) and leave placement to the kernel; with a GPU per worker the copy
 * page cache -> pinned ring -> PCIe is local only if ring and feeder threads sit on the GPU's socket. */
typedef struct ibu_numa_info {
  int32_t mode;          /* option "numa": 1 auto, 0 off */
  int32_t node;          /* NUMA node of the device's PCI function; -1: the platform does not say */
  int32_t usable_cpus;   /* CPUs of that node the process may run on (what the feeder threads are pinned to when mode = 1) */
  int32_t ring_node;     /* node the pinned ring's pages are on, as the kernel reports it (move_pages); -1: no ring yet / it will not say */
  int32_t ring_placed;   /* 1: the ring was allocated under a preferred-node policy the kernel accepted */
  int32_t reserved;
  char pci_bus_id[32];   /* "0000:c1:00.0" */
  char cpulist[256];     /* the node's CPUs as sysfs spells them, "" when unknown */
} ibu_numa_info_t;
int32_t ibu_ctx_numa(const ibu_ctx_t* ctx, ibu_numa_info_t* out);
/* The lookup on its own, against any sysfs tree (sysfs_root NULL = "/sys"): *node = NUMA node of PCI function `pci_bus_id`
 * (-1 unknown), cpulist = the node's CPUs ("" unknown), *usable_cpus (nullable) = how many of them the calling thread may use. */
int32_t ibu_numa_of_pci(const char* sysfs_root, const char* pci_bus_id, int32_t* node, char* cpulist, size_t cap,
                        int32_t* usable_cpus);
void* ibu_ctx_stream(const ibu_ctx_t* ctx);        /* hipStream_t */
int32_t ibu_ctx_synchronize(ibu_ctx_t* ctx, void* stream);
int32_t ibu_device_count(int32_t* n);
/* Tuning knobs (all optional; defaults are the measured best for MI355X):
 *   "blocks_per_cu"  1..8   cap on resident 256-thread workgroups per CU for the persistent grids
 *   "sort_variant"   0..6   tile shape of the 24-byte radix passes (0 = default: 2560-record tiles; the rest are A/B shapes of the
 *                           same algorithm kept for measurement: ibu_amd/csrc/sort.hip, kSweep)
 *   "sort_compact"   0..8   compact-key passes of the sort: when at most 12 bytes of the 24-byte key vary (16/12 records
 *                           with indices below 2^32: 11) the passes move 12-byte elements instead of records, when 13 .. 16
 *                           vary, 16-byte elements.  0 = never
 *                           (24-byte passes only), 1 = default tile shape, 2..8 = A/B tile shapes (sort.hip, kCompact).
 *                           Same result either way, byte for byte.
 *   "sort_guess"     0 | 1 | k  large compact-key sorts read the records once instead of twice: a census of three sample ranges
 *                           guesses the varying bytes, the compress pass runs on the guess and takes the exact census on the
 *                           way; a guess that missed a byte is detected and the sort continues from the exact census.
 *                           0 = never, 1 = inputs of 2^17 records and more (default), k = of k records and more.
 *   "sort_hybrid"    0 | 1 | 2  PREFIX + FINISH: instead of one pass per varying key byte, passes over the most significant P
 *                           varying bytes only, then one finishing pass that completes every run of equal prefix inside LDS.
 *                           P comes from a pair count over sample ranges (compact keys) or from n (24-byte records); runs too
 *                           long for the finishing kernel make it report an overflow, and all passes run.  0 = never,
 *                           1 = when passes are saved (default), 2 = whenever one is (tests).  Same bytes either way.
 *   "sort_idx64"     0 | 1  test knob: 1 = element / record positions are 64-bit at any size (what inputs of 2^32 records and more
 *                           take), so that those kernels can be checked against the oracle at sizes it can sort.
 *   "base_order"     0 | 1  bit order of the 2-bit codec for every pack / unpack / decode / encode issued through
 *                           this context (device kernels and the stream entry points alike):
 *                             0 = IBU_BASE_ORDER_LSB_FIRST (default): base i at bits [2i, 2i+1], "ACGT" -> 0b11100100
 *                                 (bitnuc's convention as recalled; README.md:45 names the crate, record.rs:19-27
 *                                 gives only the code table);
 *                             1 = IBU_BASE_ORDER_MSB_FIRST: base i at bits [2(len-1-i), ...], "ACGT" -> 0b00011011.
 *                           The reference holds no codec code and no vector, so the order is UNPINNED; the second
 *                           value is the hedge: if an external bitnuc vector shows the other order, callers flip this
 *                           option and no kernel changes.
 *   "alloc_probe_tries" 0..16  placement probing as a property of the library, for the arrays the library allocates and that stay
 *                           resident — ibu_device_alloc, the destination of ibu_load_to_device, the context's sort scratch.
 *                           0 = auto (default): an allocation of at least 1 GiB of which at least three candidates fit in the
 *                           device's free memory draws up to four and keeps the fastest, anything else is a plain
 *                           allocation; the drawing ends early at a candidate whose hipMalloc took longer than 10 ms + 2 ms
 *                           per GB (the driver hands out memory slowly right after large frees: more candidates would
 *                           multiply that wait, not the choice); 1 = never probe; k > 1 = allocations of at least 256 MiB
 *                           draw k candidates, however long they take.  What
 *                           probing is and why: ibu_device_alloc_probed.  What it costs: a write + read of every candidate
 *                           (~0.3 ms per GB each, twice), the candidates' memory (k x bytes at the peak; the ones not kept
 *                           are freed by a helper thread right after the choice — the driver clears VRAM when it is freed,
 *                           15-25 GB/s, which a caller need not wait for; the next probing allocation and ibu_ctx_destroy
 *                           join that thread, and so does any allocation of the LIBRARY's that would otherwise fail for want
 *                           of memory — it then tries once more; a caller about to allocate with an allocator of its own
 *                           can wait for that memory by setting this option, to any value), and one synchronisation of the
 *                           context's stream per probed allocation.  It never touches the reduce
 *                           accumulator (reset / reduce ... / fetch may span allocations).  IBU_TRACE_SORT=1 prints what was
 *                           drawn and chosen.
 *   "inflate_one_launch" 0..49152  a test knob: ibu_load_bgzf_*_to_device launches its decoder AHEAD of the copies (the waves wait for
 *                           their blocks to arrive) for files of more blocks than this; 0 (default) = one round of the decoder's short
 *                           form, 49 152 blocks: smaller files get one launch behind the last copy.
 *   "bgzf_device"    0 | 1  1 (default): ibu_reader_process_device on an untouched BGZF file inflates on the device (see there).
 *   "bgzf_range_bytes"  >= 0  a test knob: the compressed bytes per range of that path (0 = default: 3.2 GB).
 *   "load_piece_delay_ms" 0..10000  a test knob: the BGZF loads sleep that long before every piece they copy (a slow source: the waves
 *                           of a launch that runs ahead give up after ~4 s, and what they left is inflated once everything has arrived).
 *   "release_staging"    1  one-shot: frees the device staging ibu_load_bgzf_to_device / _shard_ keep between calls (the size of the
 *                           compressed bytes of the largest load so far); the next load allocates it again.
 *   "numa"           0 | 1  1 = auto (default): the context looks up the NUMA node its device hangs off (PCI bus id ->
 *                           /sys/bus/pci/devices/<bdf>/numa_node -> that node's cpulist) and keeps its host side there: the pinned
 *                           ring is allocated under a preferred-node policy, the threads that fill it (the stream producer and its
 *                           feeders, the preads of ibu_load_to_device, the inflate workers a stream starts) run on the node's
 *                           CPUs (those of them the process may use).  A platform that does not say (node -1, no sysfs, a
 *                           kernel that refuses set_mempolicy) changes nothing: threads and pages go where they went before.
 *                           0 = off.  ibu_ctx_numa says what was found and where the ring landed.
 *   "peer_access"    0 | 1  the multi-GPU sort's pulls (ibu_sort_records_contexts): 1 (default) = the pulling context enables direct
 *                           peer access to the shard's device where the topology has it (copies go over xGMI without staging; it
 *                           stays enabled for the process), 0 = never: the runtime stages the copies through the host.
 *   "sort_pull_streams" 0 | 1  test knob of the multi-GPU sort: its pulls travel on one stream per PEER DEVICE (xGMI is point to point:
 *                           copies queued on one stream would use one link at a time); 1 = peers on the puller's own device get
 *                           such streams too, so that a rehearsal on one GPU runs the fork / join of those streams.
 *   "trace_rows"     0 | 1  tests: one stderr line per kernel launch of the streaming entry points saying how many rows took
 *                           the tiled and how many the one-thread-per-row kernel.  A context starts with the value the
 *                           environment variable IBU_TRACE_ROWS had when the library first created a context (read once).
 * Unknown keys / out-of-range values return IBU_ERR_INVALID_ARG. */
#define IBU_BASE_ORDER_LSB_FIRST 0
#define IBU_BASE_ORDER_MSB_FIRST 1
int32_t ibu_ctx_set_option(ibu_ctx_t* ctx, const char* key, int64_t value);
/* Device memory helpers for callers without their own allocator (tests in C, Rust shim).  Large allocations are placement-probed
 * as option "alloc_probe_tries" says (default auto: >= 1 GiB with room for three candidates; ibu_device_alloc_probed below). */
int32_t ibu_device_alloc(ibu_ctx_t* ctx, size_t bytes, void** d_ptr);
int32_t ibu_device_free(ibu_ctx_t* ctx, void* d_ptr);
/* ibu_device_alloc with PLACEMENT PROBING, for arrays that stay resident (no reference counterpart: the crate holds its records in
 * a Vec, reader.rs:528; this is the device-side "where does the Vec live" decision).  On MI355X the rate of a streaming kernel
 * depends on which physical pages the driver handed out — the same kernel on the same GPU runs 9.4 ... 11.3 ms from one
 * allocation to the next, and an allocation keeps its rate for as long as it lives.  Allocates up to `tries` candidates of
 * `bytes` bytes (all held at once, so that they are different pages; stops quietly at the first that does not fit), streams
 * one write and one read over each, keeps the fastest in *d_ptr and frees the others.  tries <= 1, or fewer than 4096
 * records' worth of bytes: a plain allocation.  `report` (nullable) says what was measured; candidate 0 is what
 * ibu_device_alloc would have returned.  Contents unspecified.  Runs on the context's stream and synchronises it; the reduce
 * accumulator is not touched (the probe reduces into scratch of its own), so reset / reduce ... / fetch may span allocations.
 * Holds tries x bytes of device memory while it measures.  Release with ibu_device_free. */
#define IBU_ALLOC_PROBE_MAX 16
typedef struct ibu_alloc_probe {
  uint32_t tries;                  /* candidates that were allocated and timed (<= the request) */
  uint32_t chosen;                 /* the one kept */
  float ms[IBU_ALLOC_PROBE_MAX];   /* write + read time of each candidate over its whole range (0 when nothing was timed) */
} ibu_alloc_probe_t;
int32_t ibu_device_alloc_probed(ibu_ctx_t* ctx, size_t bytes, uint32_t tries, void** d_ptr, ibu_alloc_probe_t* report);
int32_t ibu_memcpy_h2d(ibu_ctx_t* ctx, void* d_dst, const void* h_src, size_t bytes, void* stream);
int32_t ibu_memcpy_d2h(ibu_ctx_t* ctx, void* h_dst, const void* d_src, size_t bytes, void* stream);

/* ======================================================================================= */
/* Hot-path kernels (all asynchronous on `stream`)                                          */
/* ======================================================================================= */
/* K1  deserialise: AoS 24-byte records -> three u64 columns.  The reference's cast_slice
 * (reader.rs:301,531; mmap.rs:268) followed by the field access every consumer performs.
 * 24 B read + 24 B written per record. */
int32_t ibu_deserialize(ibu_ctx_t* ctx, const void* d_records, size_t n, uint64_t* d_barcode,
                        uint64_t* d_umi, uint64_t* d_index, void* stream);
/* K1' serialise: three u64 columns -> AoS records (Record::new + write_batch's cast_slice,
 * record.rs:87-93, writer.rs:315-318).  48 B per record. */
int32_t ibu_serialize(ibu_ctx_t* ctx, const uint64_t* d_barcode, const uint64_t* d_umi,
                      const uint64_t* d_index, size_t n, void* d_records, void* stream);
/* 2-bit unpack of one u64 column to n*len ASCII bytes (row i at d_ascii + i*len).          */
int32_t ibu_unpack_2bit(ibu_ctx_t* ctx, const uint64_t* d_codes, size_t n, uint32_t len,
                        uint8_t* d_ascii, void* stream);
/* 2-bit pack of n rows of len ASCII bytes into one u64 column; invalid bytes are reported
 * through ibu_codec_status(). */
int32_t ibu_pack_2bit(ibu_ctx_t* ctx, const uint8_t* d_ascii, size_t n, uint32_t len,
                      uint64_t* d_codes, void* stream);
/* K2  fused decode: AoS records -> barcode ASCII (n*bc_len), UMI ASCII (n*umi_len), index
 * column.  24 B read + (bc_len+umi_len+8) B written per record.  Any output may be NULL to
 * skip that column. */
int32_t ibu_decode_ascii(ibu_ctx_t* ctx, const void* d_records, size_t n, uint32_t bc_len,
                         uint32_t umi_len, uint8_t* d_bc_ascii, uint8_t* d_umi_ascii,
                         uint64_t* d_index, void* stream);
/* K3  fused encode: barcode/UMI ASCII + index column -> AoS records.  d_index == NULL writes
 * index = first_index + i (the usual "record number" use, README.md:38-47). */
int32_t ibu_encode_ascii(ibu_ctx_t* ctx, const uint8_t* d_bc_ascii, const uint8_t* d_umi_ascii,
                         const uint64_t* d_index, uint64_t first_index, size_t n, uint32_t bc_len,
                         uint32_t umi_len, void* d_records, void* stream);
/* Invalid-base report for every pack/encode launched on this context since the last call.
 * Synchronises `stream`, returns IBU_OK or IBU_ERR_INVALID_BASE (detail.a = first offending
 * record index, detail.b = offending record count) and re-arms the slot. */
int32_t ibu_codec_status(ibu_ctx_t* ctx, void* stream, uint64_t* first_bad_record,
                         uint64_t* n_bad_records);

/* K4  reduce: the device ParallelProcessor.  Fixed processors restating the reference's
 * in-repo ones: count (lib.rs:117-129), wrapping sum of the three fields
 * (examples/parallel.rs:21-36, mmap.rs:358-373), XOR of the three fields
 * (examples/roundtrip.rs:84-87).  24 B read per record. */
typedef struct ibu_reduce_result {
  uint64_t count;
  uint64_t sum[3]; /* barcode, umi, index — wrapping (mod 2^64) */
  uint64_t xor_[3];
} ibu_reduce_result_t;
/* Accumulates INTO the context's device accumulator (call ibu_reduce_reset first). */
int32_t ibu_reduce_reset(ibu_ctx_t* ctx, void* stream);
int32_t ibu_reduce(ibu_ctx_t* ctx, const void* d_records, size_t n, void* stream);
/* Synchronises `stream` and copies the accumulator out. */
int32_t ibu_reduce_fetch(ibu_ctx_t* ctx, void* stream, ibu_reduce_result_t* out);

/* Counter-based synthetic records: r(i,k) = splitmix64(seed + 3*i + k);
 * barcode = r(i,0) & mask(2*bc_len), umi = r(i,1) & mask(2*umi_len), index = i, for
 * i in [first, first+n).  Shard-independent, so each GPU materialises its own range. */
int32_t ibu_generate(ibu_ctx_t* ctx, uint64_t seed, uint64_t first, size_t n, uint32_t bc_len,
                     uint32_t umi_len, void* d_records, void* stream);

/* Streaming device-to-device copy of `bytes` bytes (ranges must not overlap): the device form of the
 * reference's memcpy hot loops (Writer::write_slice copy_from_slice writer.rs:335-347, Writer::ingest
 * append writer.rs:477-482 when both writers' batches live in HBM).  Also the on-device copy ceiling
 * bench.py prices the other kernels against.  16-B aligned ranges take the dwordx4 kernel, anything
 * else a byte kernel. */
int32_t ibu_device_copy(ibu_ctx_t* ctx, void* d_dst, const void* d_src, size_t bytes, void* stream);

/* Device-side sort by (barcode, umi, index) — the order `derive(Ord)` defines (record.rs:58)
 * and the header's sorted flag promises (header.rs:111-113).  d_tmp: n*24 B scratch.  The context
 * additionally keeps (and grows on demand) about 1.75 B per record of its own scratch.  Any n the
 * device can hold (n < 2^40).  NOT purely asynchronous: the call synchronises `stream` a few times (64- to 128-byte
 * read-backs pick the path: the census of the varying bytes, from 2^17 records on a sample census, from 8192 records on a
 * pair count that estimates the runs of equal prefix, and the finishing kernel's overflow flag); the last kernels may still be queued
 * when it returns.
 * Stable radix sort over the key bytes that vary; when at most 16 of them do (and d_records is 16-byte
 * aligned) it runs on 12- or 16-byte compacted keys held in d_tmp (and, for 16-byte keys, in the head of
 * d_records) (option "sort_compact").  Large inputs whose keys are well spread take passes over the most significant
 * varying bytes only and one finishing pass (option "sort_hybrid"); the result is the same bytes on every path. */
int32_t ibu_sort_records(ibu_ctx_t* ctx, void* d_records, void* d_tmp, size_t n, void* stream);

/* The same order over SEVERAL shards, one per context (= per GPU), in one call — the multi-GPU form of the sort (SURVEY 8f-2:
 * sample sort with one device-to-device exchange), as ibu_mmap_process_contexts is the multi-GPU form of process_parallel.
 * EXPERIMENTAL: rehearsed with up to 16 contexts on one GPU, never run on two distinct GPUs (no such box in any round).
 * shards[i] lives on ctxs[i]'s device: n records in d_records, which has room for `capacity` records; d_tmp: capacity * 24
 * bytes of scratch on the same device.  On return shard i holds the i-th contiguous range of the global order and
 * shards[i].n says how many records that is (their sum is unchanged).  Evenly spaced samples of every shard, in proportion to
 * its size (max(16 384, 512 x n_ctxs) records in all, at most 2^19), pick the splitters; every record travels once to the owner of
 * its range (hipMemcpyPeerAsync: over xGMI between GPUs).  Three forms (multi_sort.cpp):
 *   partition first on elements — at most 11 key bytes vary over ALL shards (16/12 records with indices below 2^32), at most 32
 *     shards, 16-byte aligned buffers: a shard is compacted to 12-byte elements (one plan for all shards from the combined census
 *     words — of sample ranges when the shards are large, checked against the exact census the partition pass takes on its way;
 *     a miss re-runs that pass only), the elements are put in the order of 256 sampled key ranges by one pass of the sort's own
 *     kernels, the owners are cut on the exact counts at range boundaries, pull their pieces (12 bytes per record on the links)
 *     and sort them straight into records;
 *   partition first on records — any other key or alignment, at most 32 shards: the same on 24-byte records (their key range in
 *     the digit side stream, one 24-byte pass into the shard's scratch, the owners pull over their own records and sort once —
 *     on the census words that pass took over all shards and on one sampled prefix estimate made for all owners, not one each);
 *   sort first — more than 32 shards, option "sort_compact" = 0 on ctxs[0], or a range cut of the first two forms that does not
 *     fit a shard's capacity: every shard sorted where it lives, cut at n_ctxs - 1 splitters by binary search, 24-byte records
 *     exchanged (12-byte elements when at most 12 bytes vary), owners sort again.
 * Nothing is sorted twice in the first two.  How even the shares come out: the partition-first forms cut at the boundaries of 256
 * sampled ranges, so an owner's load is quantised to about total / 256 — 3 % of a share with 8 shards, 12 % with 32 —; the
 * sort-first form cuts at sampled quantiles of sorted shards, a few percent at any shard count.  In the partition-first forms the
 * host joins its worker threads where it needs every shard's answer (samples, range counts — copied out before the partition pass's
 * scatter kernel has run, so the plan is made while it runs) and once at the end; the exchange and the owners' sorts are ordered on
 * the devices (a pull waits for the event of the shard it reads, a sort for the pulls that read its scratch).  An owner's pulls go out on one stream per
 * peer device, so that the point-to-point links carry their pieces at the same time.
 * One host thread per context; the first error in context order is the call's.  A shard that would receive more than its
 * capacity: IBU_ERR_INVALID_ARG (detail.a = records it would receive, detail.b = its capacity) before anything has moved between
 * shards: every shard still holds its own records (untouched, or sorted locally on the sort-first path) — leave headroom for
 * uneven splits (see above for how even the shares come out; many equal records all go to one owner).  After ANY
 * OTHER error (a failed copy or kernel once the exchange has begun) the contents of the shards are unspecified.  Capacity: at
 * least n_ctxs + 1 records and 24 x capacity >= 32 x (n_ctxs - 1) bytes (the sort-first form stages the splitters and their
 * positions in d_tmp).  n_ctxs == 1 is ibu_sort_records.  Two contexts may share a device (a rehearsal on one GPU); the same
 * context twice is refused.  Peer access between the devices involved is enabled where the topology has it and stays enabled.
 * Synchronous. */
typedef struct ibu_sort_shard {
  void* d_records; /* device, 8-byte aligned: `capacity` records of room, `n` of them valid */
  void* d_tmp;     /* device, 8-byte aligned: capacity * 24 bytes of scratch                 */
  size_t n;        /* in: records of this shard; out: records of its range of the global order */
  size_t capacity;
} ibu_sort_shard_t;
int32_t ibu_sort_records_contexts(ibu_ctx_t* const* ctxs, size_t n_ctxs, ibu_sort_shard_t* shards);
/* Per-barcode aggregation of SORTED device records: the device form of the reference's BarcodeAnalyzer
 * processor (src/parallel.rs:72-98 — HashMap<barcode, count> merged in on_batch_complete).  Writes, in
 * ascending barcode order, d_barcodes[k], d_counts[k] (records with that barcode) and, when
 * d_unique_umis != NULL, the number of distinct (barcode, umi) pairs of barcode k.  *n_barcodes
 * receives the number of distinct barcodes, *n_barcode_umi_pairs (nullable) the number of distinct
 * pairs.  Size query: d_barcodes = d_counts = NULL and cap = 0.  cap too small: IBU_ERR_INVALID_ARG with
 * *n_barcodes set.  The input must be sorted (ibu_sort_records / a file whose header says sorted);
 * unsorted input yields the run-length encoding of the barcode column instead.  Synchronises `stream`
 * once (the counts come back to size the output).  n < 2^40.  Reads the records once where barcodes
 * repeat (at most 32 run heads per 8192 records: the count pass keeps them), twice where they do not;
 * the size query alone is one read — a caller that knows a bound (its whitelist) skips it and passes
 * that bound as cap. */
int32_t ibu_barcode_counts(ibu_ctx_t* ctx, const void* d_sorted_records, size_t n, uint64_t* d_barcodes,
                           uint64_t* d_counts, uint64_t* d_unique_umis, size_t cap, size_t* n_barcodes,
                           size_t* n_barcode_umi_pairs, void* stream);
/* BGZF / DEFLATE on the device (k_inflate.hip).  A bgzip file — to niffler (src/io/reader.rs:345-352) a gzip stream of many
 * members — is a chain of independent deflate blocks of at most 64 KiB whose compressed and uncompressed sizes stand in their
 * headers and trailers: the blocks can be found without inflating them, their COMPRESSED bytes can cross the PCIe link (half the
 * bytes of a records file) and every block can go to a wave of its own.
 * ibu_bgzf_scan (host): walks the block headers in buf[0, len) and describes up to `cap` whole blocks: comp_offset / comp_len = the
 * raw deflate bytes inside the member, out_len and crc32 from its trailer, out_offset = the sum of the out_len before it.
 * *consumed = bytes of buf the described blocks cover (the next call starts there), *out_bytes = their uncompressed size.  A
 * member that is not a BGZF block, or one that is cut off with final != 0: IBU_ERR_NIFFLER (the blocks in front of it are
 * described, *n_blocks says how many); a block that is not whole yet with final == 0 ends the walk quietly.
 * ibu_inflate_blocks_device: inflates the n blocks d_blocks describes (device memory, comp_offset relative to d_comp, out_offset
 * — signed — relative to d_out) — d_comp must be readable IBU_INFLATE_PAD bytes past the last block's end — and verifies length,
 * the end of the deflate stream on the block's last byte and the CRC-32, exactly as the host decoder does.  d_status[i] = 0 good /
 * 1 not a valid deflate stream of these sizes / 2 CRC-32 mismatch; *d_first_bad (device; the caller sets it to 0xFFFFFFFF) = the
 * lowest bad block.  A block writes only its own out_len bytes.  Asynchronous on `stream`; the lanes' tables live in the context's
 * sort scratch (calls of more than 49 152 blocks; one such call per context at a time, as for the sort).  One LANE per block: 64
 * blocks per wave; BGZF level 1 of 16/12 records: 53 GB/s of output for 1e8 records, 100 GB/s for 3e8 (the 16 host inflate threads
 * of the same box: 9.6).
 * A wave takes ~46 ms for its 64 blocks whatever the call's size, so a call wants tens of thousands of blocks:
 * ibu_load_bgzf_to_device (above) is the library's own use of it; the streams keep the host inflate (a ring slot holds too few). */
#define IBU_INFLATE_PAD 2048
typedef struct ibu_inflate_block {
  uint64_t comp_offset;
  int64_t out_offset;
  uint32_t comp_len, out_len, crc32, reserved;
} ibu_inflate_block_t;
int32_t ibu_bgzf_scan(const uint8_t* buf, size_t len, int32_t final, ibu_inflate_block_t* blocks, size_t cap, size_t* n_blocks,
                      size_t* consumed, uint64_t* out_bytes);
int32_t ibu_inflate_blocks_device(ibu_ctx_t* ctx, const void* d_comp, const ibu_inflate_block_t* d_blocks, size_t n, void* d_out,
                                  uint32_t* d_status, uint32_t* d_first_bad, void* stream);
/* For each of the k key records d_keys[j] (24 B each): the first position p in the SORTED device records with
 * records[p] >= key under ibu_record_cmp (record.rs:58), i.e. slice::partition_point(|r| r < key); n if there is none.
 * Positions land in d_pos[0..k) on the device; asynchronous on `stream`.  This is the splitter search of the
 * multi-GPU sample sort (ibu_amd/sharding.py): one launch for all splitters instead of a host binary search. */
int32_t ibu_lower_bound_records(ibu_ctx_t* ctx, const void* d_sorted_records, size_t n, const void* d_keys, size_t k,
                                uint64_t* d_pos, void* stream);
/* ---- compacted keys: the bytes of a set of records that VARY, as 12-byte elements --------------------------------------
 * A record is 24 bytes, but of 16-base barcodes, 12-base UMIs and indices below 2^32 only 4 + 3 + 4 bytes differ between
 * records.  ibu_sort_records sorts such inputs as 12-byte elements internally; these entry points expose the same
 * transformation so that the multi-GPU sort (ibu_amd/sharding.py) ships elements instead of records through its one
 * all-to-all — half the bytes over the point-to-point xGMI links.  There is no reference counterpart (the crate has no sort
 * and no exchange); the records that come out are the records that went in (record.rs:58-66), byte for byte.
 *
 * ibu_records_census: out[0..2] = OR of barcode / umi / index over the n records, out[3..5] = AND (n = 0: 0 and ~0, the
 *   identities, so words of several shards combine with | and &), out[6] != 0: some index is smaller than its
 *   predecessor's, out[7] != 0: some record is smaller than its predecessor (not sorted).  Synchronises `stream`.
 * ibu_key_plan_init: the plan for records whose OR / AND words are given (one shard's, or all shards' combined — every
 *   rank must use the same plan).  plan->k = number of varying bytes; compact / expand need k <= 12
 *   (IBU_ERR_INVALID_ARG otherwise).  Element byte j = the j-th least significant varying byte of the key (index bytes
 *   first, barcode bytes last): elements compared as 96-bit little-endian integers order like the records.
 * ibu_records_compact: n records -> n elements (12 n bytes at d_elems, 4-byte aligned).  ibu_records_expand: the inverse.
 *   Asynchronous on `stream`; record arrays that are 8- but not 16-byte aligned (a shard at an odd record) peel one record. */
typedef struct ibu_key_plan {
  uint32_t csel[4][3]; /* byte-gather selectors, records -> elements (v_perm_b32); row 3: the sort's 16-byte elements */
  uint32_t xsel[6][2]; /* elements -> records */
  uint32_t k;          /* varying bytes */
  uint32_t index_bytes; /* how many of them belong to the index (the least significant element bytes) */
  uint64_t base[3];    /* the constant bytes of barcode / umi / index */
} ibu_key_plan_t;
int32_t ibu_records_census(ibu_ctx_t* ctx, const void* d_records, size_t n, uint64_t out[8], void* stream);
int32_t ibu_key_plan_init(const uint64_t or_words[3], const uint64_t and_words[3], ibu_key_plan_t* plan);
int32_t ibu_records_compact(ibu_ctx_t* ctx, const ibu_key_plan_t* plan, const void* d_records, size_t n, void* d_elems, void* stream);
int32_t ibu_records_expand(ibu_ctx_t* ctx, const ibu_key_plan_t* plan, const void* d_elems, size_t n, void* d_records, void* stream);
/* `a == b` on two device-resident slices of n records each (Record derives PartialEq / Eq, record.rs:58): *first = the
 * index of the first record that differs, n if none does.  8-byte aligned inputs (16-byte: the fast path).  Synchronises. */
int32_t ibu_records_first_mismatch(ibu_ctx_t* ctx, const void* d_a, const void* d_b, size_t n, uint64_t* first, void* stream);
/* 1 if the n records are non-decreasing under ibu_record_cmp. Synchronises. */
int32_t ibu_is_sorted(ibu_ctx_t* ctx, const void* d_records, size_t n, void* stream,
                      int32_t* sorted);

/* ======================================================================================= */
/* Record streams: file / mmap / gzip  <->  device, through a pinned ring                   */
/* ======================================================================================= */
typedef struct ibu_ring_config {
  uint32_t slots;        /* pinned hipHostMalloc staging slots (>= 2; default 4)             */
  uint32_t slot_records; /* records per slot (default IBU_BATCH_SIZE * 4 = 96 MiB)           */
  uint32_t feeder_threads; /* host threads copying mmap pages into a slot (default 4)        */
  uint32_t reserved;
} ibu_ring_config_t;

typedef struct ibu_stream_stats {
  uint64_t records;
  uint64_t bytes_h2d;
  uint64_t bytes_d2h;
  uint64_t batches;
  double seconds_total;
  double seconds_kernel; /* sum of hipEvent kernel spans */
  int32_t numa_node;     /* ABI revision 4: node of the device the stream fed (-1: unknown or option "numa" = 0) */
  int32_t ring_node;     /* node the pinned ring's pages are on (-1: the kernel would not say) */
} ibu_stream_stats_t;

/* Device analogue of load_to_vec (reader.rs:510-535): whole uncompressed file -> device AoS
 * buffer of *n records.  If *d_records is NULL the library hipMallocs it (release with
 * ibu_device_free); otherwise cap_records bounds it. */
int32_t ibu_load_to_device(ibu_ctx_t* ctx, const char* path, const ibu_ring_config_t* cfg,
                           ibu_header_t* header, void** d_records, size_t cap_records, size_t* n,
                           ibu_stream_stats_t* stats);
/* The same for a BGZF (bgzip) file of the records, INFLATED ON THE DEVICE: the compressed bytes cross the link (half of them for a
 * 16/12 records file) and every block inflates straight to its place among the records (ibu_inflate_blocks_device below; the block
 * headers are walked on a thread of their own while the copies run; 1e8 records: 0.074 s against 0.25 s through the Reader; 1e9:
 * 0.31 s — the PLAIN file of the same records: 0.44 s; large files get one launch of the decoder ahead of the copies whose waves wait
 * for their blocks to arrive).  The result is what ibu_load_to_device gives for the gunzipped file — the
 * reference's load_to_vec does not decompress (reader.rs:510-535 reads the file as it is; its Reader does, through niffler,
 * :345-352): this is the bulk form of that Reader path.  Header too short: IBU_ERR_IO; invalid header: as ibu_header_validate;
 * (length - 32) % 24 != 0: IBU_ERR_INVALID_MAP_SIZE; a member that is not a BGZF block, a file that ends inside one, a block that
 * does not inflate to its announced length and CRC-32: IBU_ERR_NIFFLER (an ordinary gzip file: use the Reader).  The context keeps
 * the file size + 36 bytes per block of device memory as staging until it is destroyed (it grows only; option "release_staging" frees
 * it).  A file larger than the device's memory is loaded range by range with the _shard_ form below (every call walks the block
 * headers again: 10 ms per 6 GB). */
int32_t ibu_load_bgzf_to_device(ibu_ctx_t* ctx, const char* path, const ibu_ring_config_t* cfg, ibu_header_t* header, void** d_records,
                                size_t cap_records, size_t* n, ibu_stream_stats_t* stats);
/* ... and shard `shard` of `n_shards` of its records — the contiguous range ibu_shard_range gives (the split of process_parallel,
 * mmap.rs:297-307), *first_record (nullable) = its first record's number: every device of a node loads its own range of the same file,
 * only that range's blocks cross its link (the at most two blocks that straddle the range's ends are inflated on the host, like the
 * blocks that hold the header).  ibu_load_bgzf_to_device is shard 0 of 1. */
int32_t ibu_load_bgzf_shard_to_device(ibu_ctx_t* ctx, const char* path, const ibu_ring_config_t* cfg, size_t shard, size_t n_shards,
                                      ibu_header_t* header, void** d_records, size_t cap_records, size_t* n, uint64_t* first_record,
                                      ibu_stream_stats_t* stats);

/* Device analogue of Writer::write_batch (writer.rs:315-351): n device-resident AoS records
 * are copied back through the ring and appended to the writer (same buffered/direct rules).
 * Stream ordering: the copies are ordered behind ibu_ctx_stream(ctx).  Records produced by a kernel entry point that was
 * given another stream: use ibu_writer_write_batch_device_on below (or complete them first).  ibu_codec_status reads ONE
 * status word per context: encodes issued on several streams share it. */
int32_t ibu_writer_write_batch_device(ibu_writer_t* w, ibu_ctx_t* ctx, const ibu_ring_config_t* cfg,
                                      const void* d_records, size_t n, ibu_stream_stats_t* stats);
/* The same, with the stream the records were produced on (a kernel entry point that was given `stream`, e.g. a sort on the
 * caller's stream): the copies wait for the work queued on `producer_stream` so far; NULL = ibu_ctx_stream(ctx), i.e. the
 * call above. */
int32_t ibu_writer_write_batch_device_on(ibu_writer_t* w, ibu_ctx_t* ctx, const ibu_ring_config_t* cfg,
                                         const void* d_records, size_t n, void* producer_stream, ibu_stream_stats_t* stats);

/* ---- pull-style device record stream ------------------------------------------------------------------------------------
 * The device form of Reader::read_batch + Iterator (reader.rs:218-242, :279-306) and of the per-batch loop of process_parallel
 * (mmap.rs:312-320): the CALLER pulls one device-resident batch at a time and runs whatever it likes on it — its own kernels,
 * or this library's (decode, sort, reduce, aggregation) — the user half of ParallelProcessor (parallel.rs:100-190) with the
 * records already in HBM.  ibu_mmap_process_device / ibu_reader_process_device (below) are this loop with one of the two
 * built-in processors as the body; there is one pipeline.
 *
 *   source --producer thread (+ feeders)--> pinned slot --copy stream H2D--> device slot --ibu_stream_next--> caller
 *
 * open: the stream borrows the context's ring (cfg as for the other stream calls) and starts a producer thread that fills
 *   pinned slots from the source and queues their H2D copies; the first batch is on its way before the first next().  Until
 *   close, the context's other ring users (load_to_device, write_batch_device, process_device, another stream) return
 *   IBU_ERR_INVALID_ARG; kernel entry points, the sort and the host<->host codec pipelines are unaffected.  A reader source is
 *   borrowed, read from where it stands (records already buffered by read_batch / next come first) and must not be touched
 *   until close; an mmap source is one shard of the static split (ibu_shard_range).
 * next: *d_records = the batch (a device ring slot: *n records of 24 bytes, 16-byte aligned), *first_index = the number of the
 *   batch's first record (mmap: its position in the map; reader: records this stream delivered before it).  `stream`
 *   (NULL = ibu_ctx_stream(ctx)) is the stream the caller will read the batch on: it is made to wait for the batch's copy, so
 *   work queued on it afterwards sees the records.  *n == 0 with IBU_OK: the end of the stream (again on every later call).
 *   Batches are whole slots except the last; their concatenation is the source's record sequence.  Blocks while no batch is ready.
 * release: the caller is done queueing work on the batch; the slot is refilled once everything queued on `stream` so far has
 *   run.  Batches may be held and released in any order, at most slots - 1 at a time: next() with every slot held returns
 *   IBU_ERR_INVALID_ARG instead of waiting for ever.
 * Errors of the source surface in order: next() first hands out every batch in front of the error, then returns it, and
 *   keeps returning it.  A stream that ends inside a record is TruncatedRecord{pos} with the reference's position, after
 *   exactly the records the reference's iterator yields before its Err: whole refills of IBU_DEFAULT_BUFFER_SIZE counted from
 *   where the stream took over; the complete records of the final partial refill are dropped with it (quirk Q8,
 *   reader.rs:232-237).
 * close: stops the producer (a producer blocked inside the source — a pipe nobody writes to — is waited for: close the writing end
 *   first), waits for the copies in flight and for the context's stream, returns the ring.  Batches still held are invalid
 *   afterwards.  The reader / map stays open and is the caller's to close.  ibu_ctx_destroy under an open stream shuts the
 *   stream down first (producer joined, ring returned); the handle stays valid for ibu_stream_close only.
 * A source error (Io, Niffler) inside a refill behaves like the truncation: the whole refills in front of it are delivered,
 *   the refill under way is lost with the error (reader.rs:225-230). */
typedef struct ibu_stream ibu_stream_t;
int32_t ibu_stream_open_reader(ibu_reader_t* r, ibu_ctx_t* ctx, const ibu_ring_config_t* cfg, ibu_stream_t** out);
int32_t ibu_stream_open_mmap(const ibu_mmap_t* m, ibu_ctx_t* ctx, const ibu_ring_config_t* cfg, size_t shard, size_t n_shards,
                             ibu_stream_t** out);
int32_t ibu_stream_header(const ibu_stream_t* s, ibu_header_t* out);
int32_t ibu_stream_next(ibu_stream_t* s, void* stream, const void** d_records, size_t* n, uint64_t* first_index);
int32_t ibu_stream_release(ibu_stream_t* s, const void* d_records, void* stream);
/* records / batches / bytes_h2d delivered so far, seconds since open, the NUMA fields; seconds_kernel is 0 (the kernels are the caller's) */
int32_t ibu_stream_stats(const ibu_stream_t* s, ibu_stream_stats_t* out);
void ibu_stream_close(ibu_stream_t* s);

/* Device processors for process_parallel. */
enum {
  IBU_PROC_REDUCE = 1, /* count + sums + xors -> ibu_reduce_result_t                         */
  IBU_PROC_DECODE = 2  /* fused decode of every batch into caller-provided device columns    */
};
typedef struct ibu_decode_sink {
  uint8_t* d_bc_ascii; /* cap_records * bc_len bytes (NULL: skip the column) */
  uint8_t* d_umi_ascii;
  uint64_t* d_index;
  size_t cap_records;  /* rows every non-NULL column can hold (ABI revision 3).  A compressed stream does not say how many
                        * records it holds before it ends: the moment a batch would not fit, the call stops, drains the ring
                        * and returns IBU_ERR_INVALID_ARG (detail.a = rows needed so far, detail.b = cap_records) — never a
                        * write past the columns.  An mmap shard is checked before any work starts. */
} ibu_decode_sink_t;

/* Device analogue of MmapReader::process_parallel (mmap.rs:286-332) for ONE shard of the
 * static split: shard `shard` of `n_shards` (one per GPU / rank) streams its record range
 * through the pinned ring in IBU_BATCH_SIZE-multiples, H2D || kernel overlapped on two
 * streams.  `sink` is ibu_reduce_result_t* or ibu_decode_sink_t* according to `proc`. */
int32_t ibu_mmap_process_device(const ibu_mmap_t* m, ibu_ctx_t* ctx, const ibu_ring_config_t* cfg,
                                int32_t proc, size_t shard, size_t n_shards, void* sink,
                                ibu_stream_stats_t* stats);

/* MmapReader::process_parallel(processor, n) (mmap.rs:286-332) with a GPU per worker, in ONE call: worker i = one host thread
 * + one context on devices[i], shard i of the static split (ibu_shard_range(len, n_devices, i): per = len / n, remainder to
 * the last), workers joined in spawn order, the first error in that order is the call's (quirk Q12; the other workers still
 * run to completion, as the reference's detached threads do).  n_devices == 0: every visible device, as num_threads == 0
 * means every core (mmap.rs:292-296); an ordinal may appear more than once (two workers sharing a GPU).
 *   proc == IBU_PROC_REDUCE: `sinks` = NULL or ibu_reduce_result_t[n_devices] (the per-device partials); *total (nullable) =
 *     their sum — count and the three sums wrapping mod 2^64, the three XORs — added on the host: seven words per device,
 *     no collective (the path shards with no exchange step).
 *   proc == IBU_PROC_DECODE: `sinks` = ibu_decode_sink_t[n_devices]; sink i's columns live on devices[i] and hold shard i's
 *     rows (row r of the shard at column + r * len); total->count = records decoded.
 * `stats`: NULL or ibu_stream_stats_t[n_devices].  The context form takes contexts the caller keeps (their rings, streams and
 * scratch are reused from call to call); the device form creates and destroys one context per entry. */
int32_t ibu_mmap_process_devices(const ibu_mmap_t* m, const int32_t* devices, size_t n_devices, const ibu_ring_config_t* cfg,
                                 int32_t proc, void* sinks, ibu_reduce_result_t* total, ibu_stream_stats_t* stats);
int32_t ibu_mmap_process_contexts(const ibu_mmap_t* m, ibu_ctx_t* const* ctxs, size_t n_ctxs, const ibu_ring_config_t* cfg,
                                  int32_t proc, void* sinks, ibu_reduce_result_t* total, ibu_stream_stats_t* stats);

/* Streaming Reader (plain or gzip; reader.rs:345-352 path) -> device processor: host inflate
 * thread -> pinned ring -> H2D || kernel.  Consumes the reader to EOF.
 * A reader opened by ibu_reader_open_path on a BGZF file from which nothing has been read yet: the library reads the file itself —
 * ranges of about 6 GB of records through ibu_load_bgzf_shard_to_device, the compressed bytes over the link, the blocks inflated on
 * the device — and runs the processor over every range (context option "bgzf_device" = 1, the default; 0: through the Reader's host
 * inflate): 1e8 records REDUCE / DECODE 1.3 G records/s instead of 0.4.  Same results; the reader stands at its end afterwards; stats:
 * bytes_h2d = the compressed bytes.  A file that load does not take (a foreign member, a cut, a length that is no whole number of
 * records, a block that does not inflate) goes through the Reader's own path and fails as it fails there. */
int32_t ibu_reader_process_device(ibu_reader_t* r, ibu_ctx_t* ctx, const ibu_ring_config_t* cfg,
                                  int32_t proc, void* sink, ibu_stream_stats_t* stats);

/* ---- host <-> host codec pipelines (data starts and ends in host memory / a file) ------------------------ */
/* One shard of an MmapReader -> barcode / UMI ASCII and the index column IN HOST MEMORY: the reference's
 * consumer loop (MmapReader::slice / process_parallel -> Record -> sequences, mmap.rs:253-332 + the 2-bit
 * table record.rs:19-27) with the unpacking done on the GPU: map -> pinned ring -> H2D -> K2 decode -> D2H ->
 * caller buffers, three streams overlapped.  Buffers hold shard_records rows (row i of the shard at
 * h_bc_ascii + i*bc_len ...); any of them may be NULL to skip that column.
 * WHEN NOT TO CALL THIS: whenever the columns are wanted in HOST memory and nothing else runs on the device.  Every record
 * crosses PCIe twice (24 B in, bc_len + umi_len + 8 B out) between two host copies (map -> pinned, pinned -> caller, the
 * second one faulting in the caller's fresh pages), and the kernel is 2 % of the wall time: 0.3-0.7 G records/s at 16/12 on
 * a one-GPU box of this pool into FRESH output arrays at every size from 1e6 to 1e9 records (tools/e2e.py rows "mmap
 * decode_to_host": the rate does not grow with the size, so there is no crossover) — 58 of 145 ms at 1e8 records are the
 * kernel's page faults on 3.6 GB of new pages, which neither more threads nor MADV_POPULATE_WRITE ahead of the copies make
 * cheaper (profiles/r04_k_*) — and 1.15-1.2 G records/s into arrays that have been written before (a reused buffer); a
 * plain loop over MmapReader::slice with a scalar 2-bit unpack on the same box's 16 CPUs decodes AND re-encodes 0.55-0.84 G
 * records/s (bench.py: cpu_baseline) and pays the same page faults.  The call exists so that the
 * API is complete for consumers that hold their sequences in host memory; the device pays off when the columns STAY
 * resident — ibu_mmap_process_device(s) with IBU_PROC_DECODE (2.2 G records/s, PCIe-bound one way) and everything behind
 * it (sort, per-barcode aggregation, re-encoding) at HBM rates. */
int32_t ibu_mmap_decode_to_host(const ibu_mmap_t* m, ibu_ctx_t* ctx, const ibu_ring_config_t* cfg, size_t shard,
                                size_t n_shards, uint8_t* h_bc_ascii, uint8_t* h_umi_ascii, uint64_t* h_index,
                                ibu_stream_stats_t* stats);
/* n rows of barcode / UMI ASCII in host memory (+ optional index column; NULL -> first_index + i) -> 2-bit
 * records -> the writer: README.md:38-47's "sequences -> Record::new -> writer.write_record" loop as one
 * batch call (H2D -> K3 encode -> D2H -> Writer::write_batch's buffered/direct rule, writer.rs:321-351).
 * A byte outside ACGTacgt: IBU_ERR_INVALID_BASE (detail.a = first offending row, detail.b = offending
 * rows); batches before the first offending one are already written, nothing from it on is. */
int32_t ibu_writer_write_ascii_batch(ibu_writer_t* w, ibu_ctx_t* ctx, const ibu_ring_config_t* cfg,
                                     const uint8_t* h_bc_ascii, const uint8_t* h_umi_ascii, const uint64_t* h_index,
                                     uint64_t first_index, size_t n, uint32_t bc_len, uint32_t umi_len,
                                     ibu_stream_stats_t* stats);

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif /* IBU_HIP_H */
