/*
 * ibu_oracle.c — CPU ORACLE (test infrastructure, see ibu_oracle.h for the parity status).
 * Every function cites the lines of /root/reference it restates.  Plain C11 + pthreads.
 */
#define _GNU_SOURCE
#include "ibu_oracle.h"

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

_Static_assert(sizeof(orc_header) == ORC_HEADER_SIZE, "header.rs:248-251");
_Static_assert(sizeof(orc_record) == ORC_RECORD_SIZE, "record.rs:149-152");

static int fail(orc_err* e, int kind, uint64_t a, uint64_t b) {
  if (e) {
    e->kind = kind;
    e->a = a;
    e->b = b;
  }
  return kind;
}

/* ------------------------------------------------------------------ header.rs:84-93 */
void orc_header_new(orc_header* h, uint32_t bc_len, uint32_t umi_len) {
  h->magic = ORC_MAGIC;
  h->version = ORC_VERSION;
  h->bc_len = bc_len;
  h->umi_len = umi_len;
  h->flags = 0;
  memset(h->reserved, 0, 8);
}
/* header.rs:111-113 */
void orc_header_set_sorted(orc_header* h) { h->flags |= 1; }
/* header.rs:130-132 */
int orc_header_sorted(const orc_header* h) { return (h->flags & 1) != 0; }
/* header.rs:167-187 — order: magic, version, bc_len, umi_len */
int orc_header_validate(const orc_header* h, orc_err* e) {
  if (h->magic != ORC_MAGIC) return fail(e, ORC_E_MAGIC, ORC_MAGIC, h->magic);
  if (h->version != ORC_VERSION) return fail(e, ORC_E_VERSION, ORC_VERSION, h->version);
  if (h->bc_len == 0 || h->bc_len > 32) return fail(e, ORC_E_BC_LEN, h->bc_len, 0);
  if (h->umi_len == 0 || h->umi_len > 32) return fail(e, ORC_E_UMI_LEN, h->umi_len, 0);
  return ORC_OK;
}
/* record.rs:58 — derive(PartialOrd, Ord) on {barcode, umi, index} is lexicographic */
int orc_record_cmp(const orc_record* a, const orc_record* b) {
  if (a->barcode != b->barcode) return a->barcode < b->barcode ? -1 : 1;
  if (a->umi != b->umi) return a->umi < b->umi ? -1 : 1;
  if (a->index != b->index) return a->index < b->index ? -1 : 1;
  return 0;
}

/* =============================================================== Writer, writer.rs */
struct orc_writer {
  int is_file;
  FILE* fp;
  uint8_t* inner; /* Vec<u8> sink */
  size_t inner_len, inner_cap;
  uint8_t* buffer; /* vec![0u8; DEFAULT_BUFFER_SIZE]  writer.rs:134 */
  size_t pos;
  uint64_t records_written;
  uint64_t sink_writes;
};

static int sink_write_all(orc_writer* w, const uint8_t* p, size_t n) {
  w->sink_writes++;
  if (w->is_file) return fwrite(p, 1, n, w->fp) == n ? 0 : ORC_E_IO;
  if (w->inner_len + n > w->inner_cap) {
    size_t cap = w->inner_cap ? w->inner_cap : 4096;
    while (cap < w->inner_len + n) cap *= 2;
    uint8_t* q = (uint8_t*)realloc(w->inner, cap);
    if (!q) return ORC_E_IO;
    w->inner = q;
    w->inner_cap = cap;
  }
  memcpy(w->inner + w->inner_len, p, n);
  w->inner_len += n;
  return 0;
}

/* writer.rs:129-143 (new: header bytes written at once, no validate) / :169-179 (headless) */
static orc_writer* writer_new(FILE* fp, int is_file, const orc_header* h) {
  orc_writer* w = (orc_writer*)calloc(1, sizeof *w);
  w->is_file = is_file;
  w->fp = fp;
  w->buffer = (uint8_t*)calloc(1, ORC_BUFFER_SIZE);
  if (h) {
    if (sink_write_all(w, (const uint8_t*)h, ORC_HEADER_SIZE)) {
      free(w->buffer);
      free(w);
      return NULL;
    }
    w->sink_writes = 0; /* count only record-path writes */
  }
  return w;
}
orc_writer* orc_writer_new_mem(const orc_header* h) { return writer_new(NULL, 0, h); }
orc_writer* orc_writer_new_file(const char* path, const orc_header* h) {
  FILE* fp = fopen(path, "wb");
  if (!fp) return NULL;
  return writer_new(fp, 1, h);
}
/* writer.rs:220-226 */
static int flush_buffer(orc_writer* w) {
  if (w->pos > 0) {
    if (sink_write_all(w, w->buffer, w->pos)) return ORC_E_IO;
    w->pos = 0;
  }
  return 0;
}
/* writer.rs:260-273 */
int orc_writer_write_record(orc_writer* w, const orc_record* r) {
  if (w->pos + ORC_RECORD_SIZE > ORC_BUFFER_SIZE) {
    int rc = flush_buffer(w);
    if (rc) return rc;
  }
  memcpy(w->buffer + w->pos, r, ORC_RECORD_SIZE);
  w->pos += ORC_RECORD_SIZE;
  w->records_written += 1;
  return 0;
}
/* writer.rs:321-351 */
static int write_slice(orc_writer* w, const uint8_t* bytes, size_t len) {
  size_t num_records = len / ORC_RECORD_SIZE;
  if (len > ORC_BUFFER_SIZE) { /* :325-331 direct path */
    int rc = flush_buffer(w);
    if (rc) return rc;
    if (sink_write_all(w, bytes, len)) return ORC_E_IO;
    w->records_written += num_records;
    return 0;
  }
  const uint8_t* remaining = bytes; /* :335-347 */
  size_t rem = len;
  while (rem) {
    size_t available = ORC_BUFFER_SIZE - w->pos;
    size_t to_write = rem < available ? rem : available;
    memcpy(w->buffer + w->pos, remaining, to_write);
    w->pos += to_write;
    remaining += to_write;
    rem -= to_write;
    if (w->pos >= ORC_BUFFER_SIZE) {
      int rc = flush_buffer(w);
      if (rc) return rc;
    }
  }
  w->records_written += num_records;
  return 0;
}
/* writer.rs:315-318 */
int orc_writer_write_batch(orc_writer* w, const orc_record* r, size_t n) {
  return write_slice(w, (const uint8_t*)r, n * ORC_RECORD_SIZE);
}
/* writer.rs:477-482 */
int orc_writer_ingest(orc_writer* w, orc_writer* other) {
  int rc = flush_buffer(other);
  if (rc) return rc;
  rc = write_slice(w, other->inner, other->inner_len);
  if (rc) return rc;
  other->inner_len = 0; /* other.inner.clear() */
  return 0;
}
/* writer.rs:429-433 */
int orc_writer_finish(orc_writer* w) {
  int rc = flush_buffer(w);
  if (rc) return rc;
  if (w->is_file && fflush(w->fp)) return ORC_E_IO;
  return 0;
}
uint64_t orc_writer_records_written(const orc_writer* w) { return w->records_written; }
const uint8_t* orc_writer_inner(const orc_writer* w, size_t* len) {
  *len = w->inner_len;
  return w->inner;
}
uint64_t orc_writer_sink_writes(const orc_writer* w) { return w->sink_writes; }
static void writer_free(orc_writer* w) {
  if (w->is_file && w->fp) fclose(w->fp);
  free(w->inner);
  free(w->buffer);
  free(w);
}
/* writer.rs:519-523 */
void orc_writer_drop(orc_writer* w) {
  (void)orc_writer_finish(w);
  writer_free(w);
}
/* writer.rs:507-511 */
void orc_writer_forget(orc_writer* w) { writer_free(w); }

/* =============================================================== Reader, reader.rs */
struct orc_reader {
  const uint8_t* src; /* memory source */
  size_t src_len, src_pos, max_read;
  FILE* fp;
  uint8_t* buffer; /* capacity DEFAULT_BUFFER_SIZE  reader.rs:164 */
  orc_header header;
  size_t pos, cap;
  uint64_t bytes_read;
  int eof;
};

/* one call of R::read */
static int src_read(orc_reader* r, uint8_t* dst, size_t want, size_t* got) {
  if (r->fp) {
    *got = fread(dst, 1, want, r->fp);
    if (*got == 0 && ferror(r->fp)) return ORC_E_IO;
    return 0;
  }
  size_t left = r->src_len - r->src_pos;
  size_t n = want < left ? want : left;
  if (r->max_read && n > r->max_read) n = r->max_read;
  memcpy(dst, r->src + r->src_pos, n);
  r->src_pos += n;
  *got = n;
  return 0;
}
/* reader.rs:152-176 */
static int reader_init(orc_reader* r, orc_err* e) {
  uint8_t hb[ORC_HEADER_SIZE];
  size_t have = 0;
  while (have < ORC_HEADER_SIZE) { /* read_exact */
    size_t got = 0;
    if (src_read(r, hb + have, ORC_HEADER_SIZE - have, &got)) return fail(e, ORC_E_IO, 0, 0);
    if (got == 0) return fail(e, ORC_E_IO, 0, 0); /* UnexpectedEof -> IbuError::Io */
    have += got;
  }
  memcpy(&r->header, hb, ORC_HEADER_SIZE);
  int rc = orc_header_validate(&r->header, e);
  if (rc) return rc;
  r->buffer = (uint8_t*)malloc(ORC_BUFFER_SIZE);
  r->pos = r->cap = 0;
  r->bytes_read = ORC_HEADER_SIZE;
  r->eof = 0;
  return 0;
}
int orc_reader_new_mem(const uint8_t* data, size_t len, size_t max_read, orc_reader** out, orc_err* e) {
  orc_reader* r = (orc_reader*)calloc(1, sizeof *r);
  r->src = data;
  r->src_len = len;
  r->max_read = max_read;
  int rc = reader_init(r, e);
  if (rc) {
    free(r);
    return rc;
  }
  *out = r;
  return 0;
}
int orc_reader_new_file(const char* path, orc_reader** out, orc_err* e) {
  FILE* fp = fopen(path, "rb");
  if (!fp) return fail(e, ORC_E_IO, (uint64_t)errno, 0);
  orc_reader* r = (orc_reader*)calloc(1, sizeof *r);
  r->fp = fp;
  int rc = reader_init(r, e);
  if (rc) {
    fclose(fp);
    free(r);
    return rc;
  }
  *out = r;
  return 0;
}
void orc_reader_header(const orc_reader* r, orc_header* h) { *h = r->header; }
/* reader.rs:218-242 */
int orc_reader_read_batch(orc_reader* r, int* has_data, orc_err* e) {
  size_t read = 0;
  while (read < ORC_BUFFER_SIZE) {
    size_t got = 0;
    if (src_read(r, r->buffer + read, ORC_BUFFER_SIZE - read, &got)) return fail(e, ORC_E_IO, 0, 0);
    if (got == 0) break;
    read += got;
  }
  if (read % ORC_RECORD_SIZE != 0) {
    size_t non_rem = read - read % ORC_RECORD_SIZE;
    return fail(e, ORC_E_TRUNCATED, r->bytes_read + non_rem, 0);
  }
  r->pos = 0;
  r->cap = read / ORC_RECORD_SIZE;
  r->bytes_read += read;
  *has_data = read > 0;
  return 0;
}
/* reader.rs:279-306 */
int orc_reader_next(orc_reader* r, orc_record* out, int* got, orc_err* e) {
  *got = 0;
  if (r->eof) return 0;
  if (r->pos >= r->cap) {
    int has = 0;
    int rc = orc_reader_read_batch(r, &has, e);
    if (rc) return rc; /* Some(Err(e)); eof stays false (quirk Q9) */
    if (!has) r->eof = 1;
  }
  if (r->eof) return 0;
  memcpy(out, r->buffer + ORC_RECORD_SIZE * r->pos, ORC_RECORD_SIZE);
  r->pos += 1;
  *got = 1;
  return 0;
}
uint64_t orc_reader_bytes_read(const orc_reader* r) { return r->bytes_read; }
void orc_reader_free(orc_reader* r) {
  if (r->fp) fclose(r->fp);
  free(r->buffer);
  free(r);
}

/* =============================================================== load_to_vec, reader.rs:510-535 */
int orc_load_to_vec(const char* path, orc_header* h, orc_record** recs, size_t* n, orc_err* e) {
  FILE* fp = fopen(path, "rb");
  if (!fp) return fail(e, ORC_E_IO, (uint64_t)errno, 0);
  uint8_t hb[ORC_HEADER_SIZE];
  if (fread(hb, 1, ORC_HEADER_SIZE, fp) != ORC_HEADER_SIZE) {
    fclose(fp);
    return fail(e, ORC_E_IO, 0, 0);
  }
  memcpy(h, hb, ORC_HEADER_SIZE);
  int rc = orc_header_validate(h, e);
  if (rc) {
    fclose(fp);
    return rc;
  }
  struct stat st;
  if (fstat(fileno(fp), &st)) {
    fclose(fp);
    return fail(e, ORC_E_IO, (uint64_t)errno, 0);
  }
  size_t data_size = (size_t)st.st_size - ORC_HEADER_SIZE;
  if (data_size % ORC_RECORD_SIZE != 0) {
    fclose(fp);
    return fail(e, ORC_E_MAP_SIZE, 0, 0);
  }
  size_t num = data_size / ORC_RECORD_SIZE;
  orc_record* v = (orc_record*)calloc(num ? num : 1, sizeof *v); /* vec![Record::default(); n] */
  if (fread(v, ORC_RECORD_SIZE, num, fp) != num) {
    free(v);
    fclose(fp);
    return fail(e, ORC_E_IO, 0, 0);
  }
  fclose(fp);
  *recs = v;
  *n = num;
  return 0;
}
void orc_free(void* p) { free(p); }

/* =============================================================== MmapReader, mmap.rs */
struct orc_mmap {
  uint8_t* map;
  size_t map_len;
  orc_header header;
  size_t len;
};
/* mmap.rs:143-161 */
int orc_mmap_new(const char* path, orc_mmap** out, orc_err* e) {
  int fd = open(path, O_RDONLY);
  if (fd < 0) return fail(e, ORC_E_IO, (uint64_t)errno, 0);
  struct stat st;
  if (fstat(fd, &st)) {
    close(fd);
    return fail(e, ORC_E_IO, (uint64_t)errno, 0);
  }
  size_t flen = (size_t)st.st_size;
  if (flen < ORC_HEADER_SIZE) { /* reference panics on &map[0..32]; the oracle reports Io */
    close(fd);
    return fail(e, ORC_E_IO, 0, 0);
  }
  uint8_t* p = (uint8_t*)mmap(NULL, flen, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return fail(e, ORC_E_IO, (uint64_t)errno, 0);
  orc_mmap* m = (orc_mmap*)calloc(1, sizeof *m);
  m->map = p;
  m->map_len = flen;
  memcpy(&m->header, p, ORC_HEADER_SIZE);
  int rc = orc_header_validate(&m->header, e);
  if (!rc && (flen - ORC_HEADER_SIZE) % ORC_RECORD_SIZE != 0) rc = fail(e, ORC_E_MAP_SIZE, 0, 0);
  if (rc) {
    munmap(p, flen);
    free(m);
    return rc;
  }
  m->len = (flen - ORC_HEADER_SIZE) / ORC_RECORD_SIZE;
  *out = m;
  return 0;
}
size_t orc_mmap_len(const orc_mmap* m) { return m->len; }
void orc_mmap_header(const orc_mmap* m, orc_header* h) { *h = m->header; }
/* mmap.rs:253-270 */
int orc_mmap_slice(const orc_mmap* m, size_t start, size_t end, const orc_record** recs, size_t* n,
                   orc_err* e) {
  if (start >= m->len || end > m->len) return fail(e, ORC_E_INDEX, end, m->len);
  if (end <= start) return fail(e, ORC_E_INDEX, end, m->len);
  *recs = (const orc_record*)(m->map + ORC_HEADER_SIZE + start * ORC_RECORD_SIZE);
  *n = end - start;
  return 0;
}
void orc_mmap_free(orc_mmap* m) {
  munmap(m->map, m->map_len);
  free(m);
}

/* mmap.rs:297-307 */
void orc_shard_range(size_t len, size_t n, size_t i, size_t* start, size_t* end) {
  size_t per = len / n, rem = len % n;
  *start = i * per;
  *end = (i == n - 1) ? *start + per + rem : *start + per;
}

typedef struct pp_worker {
  const orc_mmap* m;
  size_t start, end;
  int fail_mode;
  uint64_t fail_index;
  /* "global" accumulators shared through a mutex, as examples/parallel.rs:28-35 */
  pthread_mutex_t* mu;
  orc_reduce* global;
  int rc;
  orc_err err;
} pp_worker;

/* mmap.rs:310-322 */
static void* pp_run(void* arg) {
  pp_worker* w = (pp_worker*)arg;
  size_t batch_start = w->start;
  while (batch_start < w->end) {
    size_t batch_end = batch_start + ORC_BATCH_SIZE;
    if (batch_end > w->end) batch_end = w->end;
    const orc_record* s;
    size_t n;
    w->rc = orc_mmap_slice(w->m, batch_start, batch_end, &s, &n, &w->err);
    if (w->rc) return NULL;
    orc_reduce local;
    memset(&local, 0, sizeof local);
    for (size_t i = 0; i < n; i++) { /* process_record */
      if (w->fail_mode && s[i].index == w->fail_index) {
        w->rc = fail(&w->err, ORC_E_PROCESS, s[i].index, 0);
        return NULL;
      }
      local.count += 1;
      local.sum[0] += s[i].barcode;
      local.sum[1] += s[i].umi;
      local.sum[2] += s[i].index;
      local.xor_[0] ^= s[i].barcode;
      local.xor_[1] ^= s[i].umi;
      local.xor_[2] ^= s[i].index;
    }
    pthread_mutex_lock(w->mu); /* on_batch_complete */
    w->global->count += local.count;
    for (int k = 0; k < 3; k++) {
      w->global->sum[k] += local.sum[k];
      w->global->xor_[k] ^= local.xor_[k];
    }
    w->global->batches += 1;
    pthread_mutex_unlock(w->mu);
    batch_start += ORC_BATCH_SIZE;
  }
  return NULL;
}

static int pp_drive(const orc_mmap* m, size_t num_threads, size_t cores, int fail_mode,
                    uint64_t fail_index, orc_reduce* out, orc_err* e) {
  /* mmap.rs:292-296 */
  size_t nt = num_threads == 0 ? cores : (num_threads < cores ? num_threads : cores);
  pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
  orc_reduce global;
  memset(&global, 0, sizeof global);
  pp_worker* ws = (pp_worker*)calloc(nt, sizeof *ws);
  pthread_t* th = (pthread_t*)calloc(nt, sizeof *th);
  for (size_t i = 0; i < nt; i++) {
    orc_shard_range(m->len, nt, i, &ws[i].start, &ws[i].end);
    ws[i].m = m;
    ws[i].mu = &mu;
    ws[i].global = &global;
    ws[i].fail_mode = fail_mode;
    ws[i].fail_index = fail_index;
    pthread_create(&th[i], NULL, pp_run, &ws[i]);
  }
  int rc = 0;
  /* mmap.rs:326-328: join in spawn order, first Err returned.  The reference drops the
   * remaining handles (threads detach); the oracle joins them so memory can be freed. */
  for (size_t i = 0; i < nt; i++) {
    pthread_join(th[i], NULL);
    if (!rc && ws[i].rc) {
      rc = ws[i].rc;
      if (e) *e = ws[i].err;
    }
  }
  if (out) *out = global;
  free(ws);
  free(th);
  return rc;
}
int orc_mmap_process_parallel(const orc_mmap* m, size_t num_threads, size_t cores, orc_reduce* out,
                              orc_err* e) {
  return pp_drive(m, num_threads, cores, 0, 0, out, e);
}
int orc_mmap_process_parallel_fail(const orc_mmap* m, size_t num_threads, size_t cores,
                                   uint64_t fail_index, orc_err* e) {
  return pp_drive(m, num_threads, cores, 1, fail_index, NULL, e);
}

/* =============================================================== flat checkers */
/* count: lib.rs:117-129; sums: examples/parallel.rs:21-36 (release-mode wrapping add);
 * xor: examples/roundtrip.rs:84-87 (per field, so the example's single checksum is the
 * XOR of the three). */
void orc_reduce_records(const orc_record* r, size_t n, orc_reduce* out) {
  memset(out, 0, sizeof *out);
  for (size_t i = 0; i < n; i++) {
    out->count += 1;
    out->sum[0] += r[i].barcode;
    out->sum[1] += r[i].umi;
    out->sum[2] += r[i].index;
    out->xor_[0] ^= r[i].barcode;
    out->xor_[1] ^= r[i].umi;
    out->xor_[2] ^= r[i].index;
  }
}
/* field access on a cast_slice'd &[Record]  (reader.rs:301, mmap.rs:268) */
void orc_deserialize(const orc_record* r, size_t n, uint64_t* bc, uint64_t* umi, uint64_t* idx) {
  for (size_t i = 0; i < n; i++) {
    bc[i] = r[i].barcode;
    umi[i] = r[i].umi;
    idx[i] = r[i].index;
  }
}
/* Record::new record.rs:87-93 */
void orc_serialize(const uint64_t* bc, const uint64_t* umi, const uint64_t* idx, size_t n, orc_record* r) {
  for (size_t i = 0; i < n; i++) {
    r[i].barcode = bc[i];
    r[i].umi = umi[i];
    r[i].index = idx[i];
  }
}

/* =============================================================== 2-bit codec
 * table: record.rs:22-25; cap: record.rs:27, header.rs:180-185; bit order: bitnuc
 * (unpinned, see header). */
/* The code table of record.rs:22-25 as a byte lookup (0xFF = not a base): A/a=0 C/c=1 G/g=2 T/t=3.  A table
 * rather than a switch so that the cpu_baseline leg is not dominated by mispredicted branches on random bases. */
static const uint8_t kPackLut[256] = {[0 ... 255] = 0xFF, ['A'] = 0, ['a'] = 0, ['C'] = 1, ['c'] = 1,
                                      ['G'] = 2, ['g'] = 2, ['T'] = 3, ['t'] = 3};
/* Bit order.  ORC_ORDER_LSB_FIRST (0, the default everywhere): base i of the sequence sits at bits [2i, 2i+1], so
 * "ACGT" -> 0b11100100 (bitnuc's convention as recalled: README.md:45 names the crate, nothing in the reference pins
 * it).  ORC_ORDER_MSB_FIRST (1): base i sits at bits [2(len-1-i), 2(len-1-i)+1] — the sequence read as a base-4
 * number, first base most significant, "ACGT" -> 0b00011011.  Both use the table of record.rs:22-25 and ignore the bits
 * at and above 2*len on unpack.  The second order is a hedge: one external vector decides between them. */
int orc_pack_2bit_order(const uint8_t* seq, uint32_t len, int order, uint64_t* out) {
  if (len == 0 || len > 32) return ORC_E_SEQ_LEN;
  const uint8_t* lut = kPackLut;
  uint64_t v = 0;
  uint8_t bad = 0;
  for (uint32_t i = 0; i < len; i++) {
    const uint8_t c = lut[seq[i]];
    bad |= c;
    v |= (uint64_t)(c & 3) << (order == ORC_ORDER_MSB_FIRST ? 2 * (len - 1 - i) : 2 * i);
  }
  if (bad & 0x80) return ORC_E_BASE;
  *out = v;
  return 0;
}
int orc_pack_2bit(const uint8_t* seq, uint32_t len, uint64_t* out) { return orc_pack_2bit_order(seq, len, ORC_ORDER_LSB_FIRST, out); }
int orc_unpack_2bit_order(uint64_t code, uint32_t len, int order, uint8_t* out) {
  static const uint8_t lut[4] = {'A', 'C', 'G', 'T'};
  if (len == 0 || len > 32) return ORC_E_SEQ_LEN;
  for (uint32_t i = 0; i < len; i++) out[i] = lut[(code >> (order == ORC_ORDER_MSB_FIRST ? 2 * (len - 1 - i) : 2 * i)) & 3];
  return 0;
}
int orc_unpack_2bit(uint64_t code, uint32_t len, uint8_t* out) { return orc_unpack_2bit_order(code, len, ORC_ORDER_LSB_FIRST, out); }
int orc_unpack_column_order(const uint64_t* codes, size_t n, uint32_t len, int order, uint8_t* ascii) {
  if (len == 0 || len > 32) return ORC_E_SEQ_LEN;
  for (size_t i = 0; i < n; i++) orc_unpack_2bit_order(codes[i], len, order, ascii + i * len);
  return 0;
}
int orc_unpack_column(const uint64_t* codes, size_t n, uint32_t len, uint8_t* ascii) {
  return orc_unpack_column_order(codes, n, len, ORC_ORDER_LSB_FIRST, ascii);
}
int orc_pack_column_order(const uint8_t* ascii, size_t n, uint32_t len, int order, uint64_t* codes, uint64_t* first_bad,
                          uint64_t* n_bad) {
  if (len == 0 || len > 32) return ORC_E_SEQ_LEN;
  uint64_t fb = UINT64_MAX, nb = 0;
  for (size_t i = 0; i < n; i++) {
    uint64_t v = 0;
    if (orc_pack_2bit_order(ascii + i * len, len, order, &v)) {
      if (fb == UINT64_MAX) fb = i;
      nb++;
      v = 0; /* offending rows are written as zero; the call as a whole is an error */
    }
    codes[i] = v;
  }
  if (first_bad) *first_bad = fb;
  if (n_bad) *n_bad = nb;
  return nb ? ORC_E_BASE : 0;
}
int orc_pack_column(const uint8_t* ascii, size_t n, uint32_t len, uint64_t* codes, uint64_t* first_bad,
                    uint64_t* n_bad) {
  return orc_pack_column_order(ascii, n, len, ORC_ORDER_LSB_FIRST, codes, first_bad, n_bad);
}
int orc_decode_records_order(const orc_record* r, size_t n, uint32_t bc_len, uint32_t umi_len, int order, uint8_t* bc,
                             uint8_t* umi, uint64_t* idx) {
  if (bc_len == 0 || bc_len > 32 || umi_len == 0 || umi_len > 32) return ORC_E_SEQ_LEN;
  for (size_t i = 0; i < n; i++) {
    if (bc) orc_unpack_2bit_order(r[i].barcode, bc_len, order, bc + i * bc_len);
    if (umi) orc_unpack_2bit_order(r[i].umi, umi_len, order, umi + i * umi_len);
    if (idx) idx[i] = r[i].index;
  }
  return 0;
}
int orc_decode_records(const orc_record* r, size_t n, uint32_t bc_len, uint32_t umi_len, uint8_t* bc,
                       uint8_t* umi, uint64_t* idx) {
  return orc_decode_records_order(r, n, bc_len, umi_len, ORC_ORDER_LSB_FIRST, bc, umi, idx);
}
int orc_encode_records_order(const uint8_t* bc, const uint8_t* umi, const uint64_t* idx, uint64_t first_index,
                             size_t n, uint32_t bc_len, uint32_t umi_len, int order, orc_record* r, uint64_t* first_bad,
                             uint64_t* n_bad) {
  if (bc_len == 0 || bc_len > 32 || umi_len == 0 || umi_len > 32) return ORC_E_SEQ_LEN;
  uint64_t fb = UINT64_MAX, nb = 0;
  for (size_t i = 0; i < n; i++) {
    uint64_t b = 0, u = 0;
    int bad = orc_pack_2bit_order(bc + i * bc_len, bc_len, order, &b);
    if (bad) b = 0;
    int bad2 = orc_pack_2bit_order(umi + i * umi_len, umi_len, order, &u);
    if (bad2) u = 0;
    if (bad || bad2) {
      if (fb == UINT64_MAX) fb = i;
      nb++;
    }
    r[i].barcode = b;
    r[i].umi = u;
    r[i].index = idx ? idx[i] : first_index + i;
  }
  if (first_bad) *first_bad = fb;
  if (n_bad) *n_bad = nb;
  return nb ? ORC_E_BASE : 0;
}
int orc_encode_records(const uint8_t* bc, const uint8_t* umi, const uint64_t* idx, uint64_t first_index,
                       size_t n, uint32_t bc_len, uint32_t umi_len, orc_record* r, uint64_t* first_bad,
                       uint64_t* n_bad) {
  return orc_encode_records_order(bc, umi, idx, first_index, n, bc_len, umi_len, ORC_ORDER_LSB_FIRST, r, first_bad, n_bad);
}

/* =============================================================== synthetic inputs, SURVEY §8d */
uint64_t orc_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
static uint64_t mask2(uint32_t len) { return len >= 32 ? ~0ull : ((1ull << (2 * len)) - 1); }
void orc_generate(uint64_t seed, uint64_t first, size_t n, uint32_t bc_len, uint32_t umi_len,
                  orc_record* r) {
  uint64_t mb = mask2(bc_len), mu = mask2(umi_len);
  for (size_t k = 0; k < n; k++) {
    uint64_t i = first + k;
    r[k].barcode = orc_splitmix64(seed + 3 * i + 0) & mb;
    r[k].umi = orc_splitmix64(seed + 3 * i + 1) & mu;
    r[k].index = i;
  }
}
static int cmp_q(const void* a, const void* b) { return orc_record_cmp((const orc_record*)a, (const orc_record*)b); }
/* records.sort() record.rs:199 */
void orc_sort_records(orc_record* r, size_t n) { qsort(r, n, sizeof *r, cmp_q); }
size_t orc_lower_bound(const orc_record* r, size_t n, const orc_record* key) {
  size_t lo = 0, hi = n;
  while (lo < hi) {
    size_t mid = lo + (hi - lo) / 2;
    if (orc_record_cmp(&r[mid], key) < 0) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}
int orc_is_sorted(const orc_record* r, size_t n) {
  for (size_t i = 1; i < n; i++)
    if (orc_record_cmp(&r[i - 1], &r[i]) > 0) return 0;
  return 1;
}

/* BarcodeAnalyzer, src/parallel.rs:72-98: process_record does *stats.entry(record.barcode) += 1 and
 * on_batch_complete merges the per-thread maps, so the result is the multiset {barcode -> count} whatever
 * the thread split.  Restated over records sorted by (barcode, umi, index): one entry per run, listed in
 * ascending barcode order (a HashMap has no order; sorted keys make the result comparable).  uniq[k] =
 * distinct (barcode, umi) pairs of barcode k — the UMI-dedup count, this build's addition (any out may be NULL).
 * Returns the number of distinct barcodes. */
size_t orc_barcode_counts(const orc_record* r, size_t n, uint64_t* barcodes, uint64_t* counts, uint64_t* uniq) {
  size_t k = 0;
  for (size_t i = 0; i < n; i++) {
    int new_bc = i == 0 || r[i].barcode != r[i - 1].barcode;
    int new_pair = new_bc || r[i].umi != r[i - 1].umi;
    if (new_bc) {
      if (barcodes) barcodes[k] = r[i].barcode;
      if (counts) counts[k] = 0;
      if (uniq) uniq[k] = 0;
      k++;
    }
    if (counts) counts[k - 1]++;
    if (uniq && new_pair) uniq[k - 1]++;
  }
  return k;
}

/* =============================================================== cpu_baseline legs */
static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
typedef struct bw {
  size_t start, end;
  uint32_t bc_len, umi_len;
  const orc_record* in;
  orc_record* out;
  uint8_t *bc, *umi;
  uint64_t* idx;
  uint64_t checksum;
  orc_reduce red;
  int reps; /* passes over the shard inside the timed region */
} bw;
static void* bw_codec(void* arg) {
  bw* w = (bw*)arg;
  /* batches of BATCH_SIZE like the reference's worker loop (mmap.rs:312-320) */
  for (int rep = 0; rep < (w->reps > 0 ? w->reps : 1); rep++)
  for (size_t s = w->start; s < w->end; s += ORC_BATCH_SIZE) {
    size_t e = s + ORC_BATCH_SIZE < w->end ? s + ORC_BATCH_SIZE : w->end;
    orc_decode_records(w->in + s, e - s, w->bc_len, w->umi_len, w->bc + s * w->bc_len,
                       w->umi + s * w->umi_len, w->idx + s);
    uint64_t fb, nb;
    orc_encode_records(w->bc + s * w->bc_len, w->umi + s * w->umi_len, w->idx + s, 0, e - s, w->bc_len,
                       w->umi_len, w->out + s, &fb, &nb);
  }
  uint64_t c = 0;
  for (size_t i = w->start; i < w->end; i++) c ^= w->out[i].barcode ^ w->out[i].umi ^ w->out[i].index;
  w->checksum = c;
  return NULL;
}
double orc_bench_decode_encode(size_t n, uint32_t bc_len, uint32_t umi_len, uint64_t seed, int threads, int reps,
                               uint64_t* checksum) {
  orc_record* in = (orc_record*)malloc(n * sizeof *in);
  orc_record* out = (orc_record*)malloc(n * sizeof *out);
  uint8_t* bc = (uint8_t*)malloc(n * bc_len);
  uint8_t* umi = (uint8_t*)malloc(n * umi_len);
  uint64_t* idx = (uint64_t*)malloc(n * 8);
  orc_generate(seed, 0, n, bc_len, umi_len, in);
  memset(out, 0, n * sizeof *out); /* touch pages outside the timed region */
  memset(bc, 0, n * bc_len);
  memset(umi, 0, n * umi_len);
  memset(idx, 0, n * 8);
  bw* ws = (bw*)calloc((size_t)threads, sizeof *ws);
  pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof *th);
  double t0 = now_s();
  for (int i = 0; i < threads; i++) {
    orc_shard_range(n, (size_t)threads, (size_t)i, &ws[i].start, &ws[i].end);
    ws[i].bc_len = bc_len;
    ws[i].umi_len = umi_len;
    ws[i].in = in;
    ws[i].out = out;
    ws[i].bc = bc;
    ws[i].umi = umi;
    ws[i].idx = idx;
    ws[i].reps = reps;
    pthread_create(&th[i], NULL, bw_codec, &ws[i]);
  }
  uint64_t c = 0;
  for (int i = 0; i < threads; i++) {
    pthread_join(th[i], NULL);
    c ^= ws[i].checksum;
  }
  double t1 = now_s();
  if (memcmp(in, out, n * sizeof *in) != 0) c = ~0ull; /* round trip must be exact */
  if (checksum) *checksum = c;
  free(ws);
  free(th);
  free(in);
  free(out);
  free(bc);
  free(umi);
  free(idx);
  return t1 - t0;
}
static void* bw_reduce(void* arg) {
  bw* w = (bw*)arg;
  orc_reduce_records(w->in + w->start, w->end - w->start, &w->red);
  return NULL;
}
double orc_bench_reduce(size_t n, uint64_t seed, int threads, orc_reduce* out) {
  orc_record* in = (orc_record*)malloc(n * sizeof *in);
  orc_generate(seed, 0, n, 16, 12, in);
  bw* ws = (bw*)calloc((size_t)threads, sizeof *ws);
  pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof *th);
  double t0 = now_s();
  for (int i = 0; i < threads; i++) {
    orc_shard_range(n, (size_t)threads, (size_t)i, &ws[i].start, &ws[i].end);
    ws[i].in = in;
    pthread_create(&th[i], NULL, bw_reduce, &ws[i]);
  }
  orc_reduce tot;
  memset(&tot, 0, sizeof tot);
  for (int i = 0; i < threads; i++) {
    pthread_join(th[i], NULL);
    tot.count += ws[i].red.count;
    for (int k = 0; k < 3; k++) {
      tot.sum[k] += ws[i].red.sum[k];
      tot.xor_[k] ^= ws[i].red.xor_[k];
    }
  }
  double t1 = now_s();
  if (out) *out = tot;
  free(ws);
  free(th);
  free(in);
  return t1 - t0;
}

/* ---- the example programs' phases (bench.py: cpu_baseline.phases) ------------------------------------------------ */
int orc_bench_phases(const char* path, size_t n, int threads, double seconds[5], uint64_t* xor_checksum, uint64_t sums[3], orc_err* e) {
  orc_header h;
  orc_header_new(&h, 16, 12);
  orc_header_set_sorted(&h);
  /* examples/roundtrip.rs:33-49 — one write_record per record */
  double t0 = now_s();
  orc_writer* w = orc_writer_new_file(path, &h);
  if (!w) return ORC_E_IO;
  for (size_t i = 0; i < n; i++) {
    orc_record r = {(uint64_t)i % 1000000u, ((uint64_t)i * 31u) % 1000000u, (uint64_t)i};
    int rc = orc_writer_write_record(w, &r);
    if (rc) { orc_writer_drop(w); return rc; }
  }
  int rc = orc_writer_finish(w);
  orc_writer_drop(w);
  if (rc) return rc;
  seconds[0] = now_s() - t0;
  /* examples/roundtrip.rs:80-100 — the iterator, XOR of every field */
  t0 = now_s();
  orc_reader* rd = NULL;
  rc = orc_reader_new_file(path, &rd, e);
  if (rc) return rc;
  uint64_t x = 0, seen = 0;
  for (;;) {
    orc_record r;
    int got = 0;
    rc = orc_reader_next(rd, &r, &got, e);
    if (rc) { orc_reader_free(rd); return rc; }
    if (!got) break;
    x ^= r.barcode ^ r.umi ^ r.index;
    seen++;
  }
  orc_reader_free(rd);
  seconds[1] = now_s() - t0;
  if (seen != n) return ORC_E_IO;
  if (xor_checksum) *xor_checksum = x;
  /* examples/roundtrip.rs:122-131 */
  t0 = now_s();
  orc_record* recs = NULL;
  size_t got_n = 0;
  orc_header hh;
  rc = orc_load_to_vec(path, &hh, &recs, &got_n, e);
  if (rc) return rc;
  seconds[2] = now_s() - t0;
  orc_free(recs);
  if (got_n != n) return ORC_E_IO;
  /* examples/parallel.rs:93-105 — process_parallel with the summing processor, T = 1 and T = threads */
  orc_mmap* m = NULL;
  rc = orc_mmap_new(path, &m, e);
  if (rc) return rc;
  orc_reduce red;
  for (int pass = 0; pass < 2; pass++) {
    const size_t T = pass == 0 ? 1 : (size_t)(threads > 0 ? threads : 1);
    t0 = now_s();
    rc = orc_mmap_process_parallel(m, T, T, &red, e);
    seconds[3 + pass] = now_s() - t0;
    if (rc) { orc_mmap_free(m); return rc; }
  }
  orc_mmap_free(m);
  if (red.count != n) return ORC_E_IO;
  if (sums) { sums[0] = red.sum[0]; sums[1] = red.sum[1]; sums[2] = red.sum[2]; }
  return 0;
}
