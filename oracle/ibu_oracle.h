/*
 * ibu_oracle.h — CPU ORACLE for the IBU hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the algorithm of noamteyssier/ibu v0.2.1 for the path this
 * repository accelerates.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this; the product (libibu_hip.so, ibu_amd/) never does.
 *
 * Parity status
 *   - record/header layout, Writer, Reader, load_to_vec, MmapReader::slice, process_parallel
 *     split: PINNED by the reference's own unit-test assertions, committed as
 *     tests/golden/ (JSON) (see tests/golden/make_golden.py for provenance, file:line).
 *     The reference is Rust and cannot be built in this image (no cargo/rustc, deps not
 *     vendored), so there is no oracle/_ref build.
 *   - 2-bit ASCII<->u64 codec: PARITY UNPINNED.  The reference contains no codec code and no
 *     test at that boundary; only the table A=00 C=01 G=10 T=11 and the 32-base cap are
 *     stated (src/constructs/record.rs:19-27).  README.md:45 points at the `bitnuc` crate,
 *     which is not in Cargo.toml (no pinned version).  Bit order follows bitnuc's published
 *     convention (first base in the least-significant bits; as_2bit(b"ACGT") == 0b11100100).
 *
 * Error convention: functions return 0 or an ORC_E_* ordinal that equals the 1-based
 * declaration order of IbuError (src/error.rs:56-128); payloads go to orc_err.
 */
#ifndef IBU_ORACLE_H
#define IBU_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#define ORC_MAGIC 0x21554249u           /* src/constructs/header.rs:5 */
#define ORC_VERSION 2u                  /* header.rs:6 */
#define ORC_HEADER_SIZE 32              /* header.rs:7, asserted :248-251 */
#define ORC_RECORD_SIZE 24              /* record.rs:3, asserted :149-152 */
#define ORC_BUFFER_SIZE (48 * 1024 * 24) /* reader.rs:14, writer.rs:10 */
#define ORC_BATCH_SIZE (1024 * 1024)    /* mmap.rs:284 */

enum {
  ORC_OK = 0,
  ORC_E_IO = 1,
  ORC_E_NIFFLER = 2,
  ORC_E_MAGIC = 3,
  ORC_E_TRUNCATED = 4,
  ORC_E_VERSION = 5,
  ORC_E_BC_LEN = 6,
  ORC_E_UMI_LEN = 7,
  ORC_E_MAP_SIZE = 8,
  ORC_E_INDEX = 9,
  ORC_E_PROCESS = 10,
  ORC_E_BASE = 11, /* codec: invalid base (bitnuc: NucleotideError::InvalidBase) */
  ORC_E_SEQ_LEN = 12
};

typedef struct orc_header {
  uint32_t magic, version, bc_len, umi_len;
  uint64_t flags;
  uint8_t reserved[8];
} orc_header;

typedef struct orc_record {
  uint64_t barcode, umi, index;
} orc_record;

typedef struct orc_err {
  int32_t kind;
  uint64_t a, b;
} orc_err;

/* ---- header / record ---- */
void orc_header_new(orc_header* h, uint32_t bc_len, uint32_t umi_len);
void orc_header_set_sorted(orc_header* h);
int orc_header_sorted(const orc_header* h);
int orc_header_validate(const orc_header* h, orc_err* e);
int orc_record_cmp(const orc_record* a, const orc_record* b);

/* ---- Writer over a Vec<u8> sink or a FILE* ---- */
typedef struct orc_writer orc_writer;
orc_writer* orc_writer_new_mem(const orc_header* header_or_null);
orc_writer* orc_writer_new_file(const char* path, const orc_header* header_or_null);
int orc_writer_write_record(orc_writer* w, const orc_record* r);
int orc_writer_write_batch(orc_writer* w, const orc_record* r, size_t n);
int orc_writer_ingest(orc_writer* w, orc_writer* other_mem);
int orc_writer_finish(orc_writer* w);
uint64_t orc_writer_records_written(const orc_writer* w);
const uint8_t* orc_writer_inner(const orc_writer* w, size_t* len); /* mem sink bytes so far */
uint64_t orc_writer_sink_writes(const orc_writer* w); /* number of write_all calls on inner */
void orc_writer_drop(orc_writer* w);   /* Drop: finish().ok() */
void orc_writer_forget(orc_writer* w); /* into_inner: no flush */

/* ---- Reader over memory (with an optional max bytes per read() to model short reads) ---- */
typedef struct orc_reader orc_reader;
int orc_reader_new_mem(const uint8_t* data, size_t len, size_t max_read, orc_reader** out, orc_err* e);
int orc_reader_new_file(const char* path, orc_reader** out, orc_err* e);
void orc_reader_header(const orc_reader* r, orc_header* h);
int orc_reader_read_batch(orc_reader* r, int* has_data, orc_err* e);
int orc_reader_next(orc_reader* r, orc_record* out, int* got, orc_err* e);
uint64_t orc_reader_bytes_read(const orc_reader* r);
void orc_reader_free(orc_reader* r);

/* ---- load_to_vec ---- */
int orc_load_to_vec(const char* path, orc_header* h, orc_record** recs, size_t* n, orc_err* e);
void orc_free(void* p);

/* ---- MmapReader ---- */
typedef struct orc_mmap orc_mmap;
int orc_mmap_new(const char* path, orc_mmap** out, orc_err* e);
size_t orc_mmap_len(const orc_mmap* m);
void orc_mmap_header(const orc_mmap* m, orc_header* h);
int orc_mmap_slice(const orc_mmap* m, size_t start, size_t end, const orc_record** recs, size_t* n,
                   orc_err* e);
void orc_mmap_free(orc_mmap* m);

void orc_shard_range(size_t len, size_t n, size_t i, size_t* start, size_t* end);

typedef struct orc_reduce {
  uint64_t count;
  uint64_t sum[3];
  uint64_t xor_[3];
  uint64_t batches; /* on_batch_complete calls */
} orc_reduce;
/* process_parallel with the reference's in-repo processors folded into one; cores = what
 * num_cpus::get() would report (passed in so tests are deterministic). */
int orc_mmap_process_parallel(const orc_mmap* m, size_t num_threads, size_t cores, orc_reduce* out,
                              orc_err* e);
/* Same walk, failing (Process error) when a record with index == fail_index is seen. */
int orc_mmap_process_parallel_fail(const orc_mmap* m, size_t num_threads, size_t cores,
                                   uint64_t fail_index, orc_err* e);

/* ---- flat-array forms used as the checker for the device kernels ---- */
void orc_reduce_records(const orc_record* r, size_t n, orc_reduce* out);
void orc_deserialize(const orc_record* r, size_t n, uint64_t* bc, uint64_t* umi, uint64_t* idx);
void orc_serialize(const uint64_t* bc, const uint64_t* umi, const uint64_t* idx, size_t n, orc_record* r);

/* ---- 2-bit codec (convention in the header comment) ---- */
int orc_pack_2bit(const uint8_t* seq, uint32_t len, uint64_t* out);       /* 0, ORC_E_BASE, ORC_E_SEQ_LEN */
int orc_unpack_2bit(uint64_t code, uint32_t len, uint8_t* out);           /* 0, ORC_E_SEQ_LEN */
int orc_unpack_column(const uint64_t* codes, size_t n, uint32_t len, uint8_t* ascii);
/* returns 0 or ORC_E_BASE; *first_bad = first offending row, *n_bad = offending rows */
int orc_pack_column(const uint8_t* ascii, size_t n, uint32_t len, uint64_t* codes, uint64_t* first_bad,
                    uint64_t* n_bad);
int orc_decode_records(const orc_record* r, size_t n, uint32_t bc_len, uint32_t umi_len, uint8_t* bc,
                       uint8_t* umi, uint64_t* idx);
int orc_encode_records(const uint8_t* bc, const uint8_t* umi, const uint64_t* idx_or_null,
                       uint64_t first_index, size_t n, uint32_t bc_len, uint32_t umi_len,
                       orc_record* r, uint64_t* first_bad, uint64_t* n_bad);

/* The same six with an explicit bit order (ORC_ORDER_*; the forms above are ORC_ORDER_LSB_FIRST). */
#define ORC_ORDER_LSB_FIRST 0 /* base i at bits [2i, 2i+1]: "ACGT" -> 0b11100100 (default; bitnuc's convention as recalled) */
#define ORC_ORDER_MSB_FIRST 1 /* base i at bits [2(len-1-i), ...]: "ACGT" -> 0b00011011 (the hedge) */
int orc_pack_2bit_order(const uint8_t* seq, uint32_t len, int order, uint64_t* out);
int orc_unpack_2bit_order(uint64_t code, uint32_t len, int order, uint8_t* out);
int orc_unpack_column_order(const uint64_t* codes, size_t n, uint32_t len, int order, uint8_t* ascii);
int orc_pack_column_order(const uint8_t* ascii, size_t n, uint32_t len, int order, uint64_t* codes, uint64_t* first_bad,
                          uint64_t* n_bad);
int orc_decode_records_order(const orc_record* r, size_t n, uint32_t bc_len, uint32_t umi_len, int order, uint8_t* bc,
                             uint8_t* umi, uint64_t* idx);
int orc_encode_records_order(const uint8_t* bc, const uint8_t* umi, const uint64_t* idx_or_null, uint64_t first_index,
                             size_t n, uint32_t bc_len, uint32_t umi_len, int order, orc_record* r, uint64_t* first_bad,
                             uint64_t* n_bad);

/* ---- synthetic inputs ---- */
uint64_t orc_splitmix64(uint64_t x);
void orc_generate(uint64_t seed, uint64_t first, size_t n, uint32_t bc_len, uint32_t umi_len,
                  orc_record* r);
void orc_sort_records(orc_record* r, size_t n);
size_t orc_barcode_counts(const orc_record* sorted, size_t n, uint64_t* barcodes, uint64_t* counts, uint64_t* uniq);
int orc_is_sorted(const orc_record* r, size_t n);
/* slice::partition_point(|r| r < key) on sorted records: first index whose record is >= key (n if none) */
size_t orc_lower_bound(const orc_record* sorted, size_t n, const orc_record* key);

/* ---- cpu_baseline legs for bench.py: static range split over `threads` OS threads, like
 * process_parallel (mmap.rs:297-322).  Returns seconds. ---- */
double orc_bench_decode_encode(size_t n, uint32_t bc_len, uint32_t umi_len, uint64_t seed, int threads, int reps,
                               uint64_t* checksum); /* seconds for `reps` passes over n records */
double orc_bench_reduce(size_t n, uint64_t seed, int threads, orc_reduce* out);
/* The phases of the reference's two example programs on a file of n records (16,12; sorted flag set), each timed on its own:
 *   seconds[0]  the write_record loop: Record(i % 1e6, 31 i % 1e6, i) for i in 0..n, then finish   examples/roundtrip.rs:33-49
 *   seconds[1]  the streaming Reader, XOR of the three fields of every record                      examples/roundtrip.rs:80-100
 *   seconds[2]  load_to_vec                                                                        examples/roundtrip.rs:122-131
 *   seconds[3]  MmapReader::process_parallel, sum of the three fields, 1 thread                    examples/parallel.rs:93-105
 *   seconds[4]  the same on `threads` threads
 * *xor_checksum = the streaming read's checksum, sums[3] = process_parallel's field sums (threads run).  0 or an ORC_E_* code.
 * The file is left in place (the caller removes it). */
int orc_bench_phases(const char* path, size_t n, int threads, double seconds[5], uint64_t* xor_checksum, uint64_t sums[3], orc_err* e);

#endif
