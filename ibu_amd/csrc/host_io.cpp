// host_io.cpp — host half of libibu_hip.so: the reference's Header / Record / Writer / Reader /
// load_to_vec / MmapReader / ParallelReader behaviour behind the C ABI of include/ibu_hip.h.
//
// This is I/O plumbing (syscalls, buffering, error mapping): it performs no codec or record
// arithmetic — that lives only in the k_*.hip kernel files.  Reference lines are cited per function; the
// quirk numbers (Q1..Q15) refer to SURVEY.md Appendix C.
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cstddef>
#include <cstdlib>

#include <atomic>
#include <future>
#include <memory>
#include <new>
#include <thread>
#include <utility>
#include <string>
#include <vector>

#include "common.hpp"
#include "host_io.hpp"
#include "pgzip.hpp"

static_assert(sizeof(ibu_header_t) == IBU_HEADER_SIZE, "header.rs:248-251");
static_assert(sizeof(ibu_record_t) == IBU_RECORD_SIZE, "record.rs:149-152");
static_assert(offsetof(ibu_header_t, flags) == 16 && offsetof(ibu_header_t, reserved) == 24, "repr(C)");
static_assert(offsetof(ibu_record_t, umi) == 8 && offsetof(ibu_record_t, index) == 16, "repr(C)");

using namespace ibu;

// ------------------------------------------------------------------------------------------
// misc
// ------------------------------------------------------------------------------------------
extern "C" void ibu_last_error(ibu_error_detail_t* out) {
  if (out) *out = tls_error();
}
extern "C" const char* ibu_status_name(int32_t s) {
  switch (s) {
    case IBU_OK: return "Ok";
    case IBU_ERR_IO: return "Io";
    case IBU_ERR_NIFFLER: return "Niffler";
    case IBU_ERR_INVALID_MAGIC: return "InvalidMagicNumber";
    case IBU_ERR_TRUNCATED_RECORD: return "TruncatedRecord";
    case IBU_ERR_INVALID_VERSION: return "InvalidVersion";
    case IBU_ERR_INVALID_BC_LEN: return "InvalidBarcodeLength";
    case IBU_ERR_INVALID_UMI_LEN: return "InvalidUmiLength";
    case IBU_ERR_INVALID_MAP_SIZE: return "InvalidMapSize";
    case IBU_ERR_INVALID_INDEX: return "InvalidIndex";
    case IBU_ERR_PROCESS: return "Process";
    case IBU_ERR_INVALID_BASE: return "InvalidBase";
    case IBU_ERR_SEQ_LEN: return "SeqLen";
    case IBU_ERR_INVALID_ARG: return "InvalidArg";
    case IBU_ERR_HIP: return "Hip";
    case IBU_ERR_NO_DEVICE: return "NoDevice";
    default: return "Unknown";
  }
}
extern "C" const char* ibu_version(void) { return "ibu_hip 0.5.0 (format v2, reference ibu 0.2.1, gfx950)"; }
// 2: + device_copy, barcode_counts, decode_to_host, write_ascii_batch, ctx_set_option
// 3: ibu_decode_sink_t.cap_records (layout change), + lower_bound_records, "base_order" / "sort_variant" options
// 4: ibu_stream_stats_t + numa_node / ring_node (layout change), + ibu_stream_* (pull stream), ibu_ctx_numa, ibu_numa_of_pci, options "numa",
//    "peer_access", "alloc_probe_tries" = 0 (auto, the new default)
extern "C" uint32_t ibu_abi_revision(void) { return 4; }
extern "C" void ibu_free(void* p) { free(p); }

// ------------------------------------------------------------------------------------------
// Header (src/constructs/header.rs)
// ------------------------------------------------------------------------------------------
extern "C" void ibu_header_init(ibu_header_t* h, uint32_t bc_len, uint32_t umi_len) {  // :84-93
  h->magic = IBU_MAGIC;
  h->version = IBU_VERSION;
  h->bc_len = bc_len;
  h->umi_len = umi_len;
  h->flags = 0;
  memset(h->reserved, 0, sizeof h->reserved);
}
extern "C" void ibu_header_set_sorted(ibu_header_t* h) { h->flags |= IBU_FLAG_SORTED; }          // :111-113
extern "C" int32_t ibu_header_sorted(const ibu_header_t* h) { return (h->flags & IBU_FLAG_SORTED) != 0; }  // :130-132
extern "C" int32_t ibu_header_validate(const ibu_header_t* h) {                                   // :167-187
  if (!h) return err_arg("header is NULL");
  if (h->magic != IBU_MAGIC) return err_magic(h->magic);
  if (h->version != IBU_VERSION) return err_version(h->version);
  if (h->bc_len == 0 || h->bc_len > 32) return err_bc_len(h->bc_len);
  if (h->umi_len == 0 || h->umi_len > 32) return err_umi_len(h->umi_len);
  return IBU_OK;
}
extern "C" int32_t ibu_header_from_bytes(const uint8_t* bytes, size_t len, ibu_header_t* out) {  // :226-228
  if (!bytes || !out || len != IBU_HEADER_SIZE) return err_arg("Header::from_bytes needs exactly 32 bytes");
  memcpy(out, bytes, IBU_HEADER_SIZE);
  return IBU_OK;
}
extern "C" int32_t ibu_header_as_bytes(const ibu_header_t* h, uint8_t* out, size_t cap) {        // :203-205
  if (!h || !out || cap < IBU_HEADER_SIZE) return err_arg("Header::as_bytes needs 32 bytes of room");
  memcpy(out, h, IBU_HEADER_SIZE);
  return IBU_OK;
}

// ------------------------------------------------------------------------------------------
// Record (src/constructs/record.rs)
// ------------------------------------------------------------------------------------------
extern "C" int32_t ibu_record_from_bytes(const uint8_t* bytes, size_t len, ibu_record_t* out) {  // :130-132
  if (!bytes || !out || len != IBU_RECORD_SIZE) return err_arg("Record::from_bytes needs exactly 24 bytes");
  memcpy(out, bytes, IBU_RECORD_SIZE);
  return IBU_OK;
}
extern "C" int32_t ibu_record_as_bytes(const ibu_record_t* r, uint8_t* out, size_t cap) {        // :108-110
  if (!r || !out || cap < IBU_RECORD_SIZE) return err_arg("Record::as_bytes needs 24 bytes of room");
  memcpy(out, r, IBU_RECORD_SIZE);
  return IBU_OK;
}
extern "C" int32_t ibu_record_cmp(const ibu_record_t* a, const ibu_record_t* b) {  // derive(Ord) :58
  const uint64_t ka[3] = {a->barcode, a->umi, a->index}, kb[3] = {b->barcode, b->umi, b->index};
  for (int k = 0; k < 3; ++k)
    if (ka[k] != kb[k]) return ka[k] < kb[k] ? -1 : 1;
  return 0;
}

// ------------------------------------------------------------------------------------------
// sinks / sources
// ------------------------------------------------------------------------------------------
namespace {

struct Sink {
  virtual ~Sink() {}
  virtual int write_all(const uint8_t* p, size_t n) = 0;  // 0 or errno
  virtual int flush() { return 0; }
};
struct MemSink : Sink {
  std::vector<uint8_t> v;
  int write_all(const uint8_t* p, size_t n) override {
    try {
      v.insert(v.end(), p, p + n);
    } catch (const std::bad_alloc&) {
      return ENOMEM;
    }
    return 0;
  }
};
struct FdSink : Sink {
  int fd;
  bool owned;
  FdSink(int f, bool o) : fd(f), owned(o) {}
  ~FdSink() override {
    if (owned && fd >= 0) close(fd);
  }
  int write_all(const uint8_t* p, size_t n) override {
    while (n) {
      ssize_t k = ::write(fd, p, n);
      if (k < 0) {
        if (errno == EINTR) continue;
        return errno;
      }
      p += k;
      n -= (size_t)k;
    }
    return 0;
  }
};
struct CallbackSink : Sink {
  ibu_write_fn wr;
  ibu_flush_fn fl;
  void* user;
  CallbackSink(ibu_write_fn w, ibu_flush_fn f, void* u) : wr(w), fl(f), user(u) {}
  int write_all(const uint8_t* p, size_t n) override { return wr(user, p, n); }
  int flush() override { return fl ? fl(user) : 0; }
};

struct Source {
  virtual ~Source() {}
  virtual int read(uint8_t* dst, size_t cap, size_t* got) = 0;  // 0 or errno; *got == 0 is EOF
};
struct MemSource : Source {
  const uint8_t* p;
  size_t len, pos = 0;
  MemSource(const uint8_t* d, size_t n) : p(d), len(n) {}
  int read(uint8_t* dst, size_t cap, size_t* got) override {
    size_t k = len - pos < cap ? len - pos : cap;
    memcpy(dst, p + pos, k);
    pos += k;
    *got = k;
    return 0;
  }
};
struct FdSource : Source {
  int fd;
  bool owned;
  // A regular file is read with eight preads at once when the caller asks for a lot (a ring slot, a batch of compressed
  // input): one read(2) copies ~10 GB/s out of the page cache, the pinned ring and PCIe take 50.
  bool regular = false;
  off_t pos = 0;
  FdSource(int f, bool o) : fd(f), owned(o) {
    struct stat st;
    if (fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && !getenv("IBU_NO_PARALLEL_READ")) {
      const off_t cur = lseek(fd, 0, SEEK_CUR);
      if (cur >= 0) { regular = true; pos = cur; }
    }
  }
  ~FdSource() override {
    if (owned && fd >= 0) close(fd);
  }
  int read(uint8_t* dst, size_t cap, size_t* got) override {
    constexpr size_t kParMin = (size_t)8 << 20;
    if (regular && cap >= kParMin) {
      constexpr unsigned kPieces = 8;
      const size_t per = ((cap / kPieces) + 4095) & ~(size_t)4095;
      size_t done[kPieces] = {0};
      int err[kPieces] = {0};
      const off_t base = pos;
      const int f = fd;
      run_pieces(kPieces, [&, base, f](unsigned i) {    // never throws
        const size_t off = (size_t)i * per;
        if (off >= cap) return;
        const size_t len = off + per < cap ? per : cap - off;
        while (done[i] < len) {
          const ssize_t k = pread(f, dst + off + done[i], len - done[i], base + (off_t)(off + done[i]));
          if (k < 0) { if (errno == EINTR) continue; err[i] = errno; return; }
          if (k == 0) return;                           // end of file inside this piece
          done[i] += (size_t)k;
        }
      });
      size_t total = 0;
      for (unsigned i = 0; i < kPieces; ++i) {           // the contiguous prefix that was read
        const size_t off = (size_t)i * per;
        if (off >= cap) break;
        const size_t len = off + per < cap ? per : cap - off;
        if (err[i] && done[i] == 0 && total == 0) return err[i];
        total += done[i];
        if (done[i] < len) break;
      }
      pos = base + (off_t)total;
      (void)lseek(fd, pos, SEEK_SET);                  // keep the descriptor's own offset where a plain read would leave it
      *got = total;
      return 0;
    }
    for (;;) {
      ssize_t k = ::read(fd, dst, cap);
      if (k < 0) {
        if (errno == EINTR) continue;
        return errno;
      }
      if (regular) pos += k;
      *got = (size_t)k;
      return 0;
    }
  }
};
struct CallbackSource : Source {
  ibu_read_fn rd;
  void* user;
  CallbackSource(ibu_read_fn r, void* u) : rd(r), user(u) {}
  int read(uint8_t* dst, size_t cap, size_t* got) override { return rd(user, dst, cap, got); }
};
// A source with a few already-consumed bytes pushed back in front (format sniffing).
struct PrefixSource : Source {
  std::vector<uint8_t> pre;
  size_t pos = 0;
  std::unique_ptr<Source> inner;
  int read(uint8_t* dst, size_t cap, size_t* got) override {
    if (pos < pre.size()) {
      size_t k = pre.size() - pos < cap ? pre.size() - pos : cap;
      memcpy(dst, pre.data() + pos, k);
      pos += k;
      *got = k;
      return 0;
    }
    return inner->read(dst, cap, got);
  }
};
// gzip (possibly multi-member) inflate over another source — the part of niffler that
// Reader::from_path relies on (reader.rs:348-352).  Errors surface as EPROTO -> the reader maps
// them to IBU_ERR_NIFFLER.
struct GzSource : Source {
  std::unique_ptr<Source> inner;
  z_stream zs;
  std::vector<uint8_t> in;
  bool inited = false, inner_eof = false, member_done = false;
  explicit GzSource(std::unique_ptr<Source> s) : inner(std::move(s)), in(1 << 18) {
    memset(&zs, 0, sizeof zs);
    inited = inflateInit2(&zs, 15 + 16) == Z_OK;
  }
  ~GzSource() override {
    if (inited) inflateEnd(&zs);
  }
  int read(uint8_t* dst, size_t cap, size_t* got) override {
    *got = 0;
    if (!inited) return EPROTO;
    const size_t want = cap > (1u << 30) ? (size_t)(1u << 30) : cap;
    zs.next_out = dst;
    zs.avail_out = (uInt)want;
    while (zs.avail_out == (uInt)want) {
      if (zs.avail_in == 0 && !inner_eof) {
        size_t k = 0;
        int rc = inner->read(in.data(), in.size(), &k);
        if (rc) return rc;
        if (k == 0) inner_eof = true;
        zs.next_in = in.data();
        zs.avail_in = (uInt)k;
      }
      if (member_done) {
        if (zs.avail_in == 0 && inner_eof) return 0;  // clean EOF after the last member
        if (inflateReset(&zs) != Z_OK) return EPROTO;
        member_done = false;
      }
      if (zs.avail_in == 0 && inner_eof) return EPROTO;  // stream ends inside a member
      int rc = inflate(&zs, Z_NO_FLUSH);
      if (rc == Z_STREAM_END) member_done = true;
      else if (rc != Z_OK && rc != Z_BUF_ERROR) return EPROTO;
    }
    *got = want - zs.avail_out;
    return 0;
  }
};

// bzip2 / xz / zstd input — the rest of niffler's format set (reader.rs:348-352 sniffs all four).  This image
// ships the runtime libraries (libbz2.so.1, liblzma.so.5, libzstd.so.1) but not their headers, so the three
// decoders are bound with dlopen and the handful of prototypes / stream structs below (all of them frozen ABI
// of those libraries).  A missing library gives IBU_ERR_NIFFLER when such a file is opened; nothing else depends
// on them.  Concatenated streams / frames are followed to the end, like the gzip path.
struct DlLib {
  void* h = nullptr;
  explicit DlLib(const char* const* names) {
    for (; *names && !h; ++names) h = dlopen(*names, RTLD_NOW | RTLD_LOCAL);
  }
  template <class F> F sym(const char* n) const { return h ? reinterpret_cast<F>(dlsym(h, n)) : nullptr; }
};

struct InflatingSource : Source {  // shared refill loop: inner bytes -> `in`, subclass turns them into output
  std::unique_ptr<Source> inner;
  std::vector<uint8_t> in;
  size_t in_pos = 0, in_len = 0;
  bool inner_eof = false;
  explicit InflatingSource(std::unique_ptr<Source> s) : inner(std::move(s)), in(1 << 18) {}
  int fill() {  // 0 / errno; sets inner_eof
    if (in_pos < in_len || inner_eof) return 0;
    size_t k = 0;
    int rc = inner->read(in.data(), in.size(), &k);
    if (rc) return rc;
    in_pos = 0;
    in_len = k;
    if (k == 0) inner_eof = true;
    return 0;
  }
};

struct BzSource : InflatingSource {
  struct bz_stream {
    char* next_in; unsigned avail_in, total_in_lo32, total_in_hi32;
    char* next_out; unsigned avail_out, total_out_lo32, total_out_hi32;
    void* state; void* (*bzalloc)(void*, int, int); void (*bzfree)(void*, void*); void* opaque;
  };
  typedef int (*init_fn)(bz_stream*, int, int);
  typedef int (*run_fn)(bz_stream*);
  init_fn init = nullptr; run_fn run = nullptr, end = nullptr;
  bz_stream zs;
  bool live = false, done_stream = true;
  explicit BzSource(std::unique_ptr<Source> s) : InflatingSource(std::move(s)) {
    static const char* const names[] = {"libbz2.so.1.0", "libbz2.so.1", "libbz2.so", nullptr};
    static DlLib lib(names);
    init = lib.sym<init_fn>("BZ2_bzDecompressInit");
    run = lib.sym<run_fn>("BZ2_bzDecompress");
    end = lib.sym<run_fn>("BZ2_bzDecompressEnd");
    memset(&zs, 0, sizeof zs);
  }
  ~BzSource() override { if (live) end(&zs); }
  bool usable() const { return init && run && end; }
  int read(uint8_t* dst, size_t cap, size_t* got) override {
    *got = 0;
    if (!usable()) return EPROTO;
    const unsigned want = cap > (1u << 30) ? (1u << 30) : (unsigned)cap;
    size_t produced = 0;
    while (produced == 0) {
      int rc = fill();
      if (rc) return rc;
      if (done_stream) {
        if (in_pos == in_len && inner_eof) return 0;  // clean end after a whole stream
        if (live) { end(&zs); live = false; }
        memset(&zs, 0, sizeof zs);
        if (init(&zs, 0, 0) != 0) return EPROTO;
        live = true;
        done_stream = false;
      }
      if (in_pos == in_len && inner_eof) return EPROTO;  // ends inside a stream
      zs.next_in = reinterpret_cast<char*>(in.data() + in_pos);
      zs.avail_in = (unsigned)(in_len - in_pos);
      zs.next_out = reinterpret_cast<char*>(dst);
      zs.avail_out = want;
      const int r = run(&zs);
      in_pos = in_len - zs.avail_in;
      produced = want - zs.avail_out;
      if (r == 4 /*BZ_STREAM_END*/) done_stream = true;
      else if (r != 0 /*BZ_OK*/) return EPROTO;
    }
    *got = produced;
    return 0;
  }
};

struct XzSource : InflatingSource {
  struct lzma_stream {
    const uint8_t* next_in; size_t avail_in; uint64_t total_in;
    uint8_t* next_out; size_t avail_out; uint64_t total_out;
    const void* allocator; void* internal;
    void *rp1, *rp2, *rp3, *rp4; uint64_t ri1, ri2; size_t ri3, ri4; int re1, re2;
  };
  typedef int (*dec_fn)(lzma_stream*, uint64_t, uint32_t);
  typedef int (*code_fn)(lzma_stream*, int);
  typedef void (*end_fn)(lzma_stream*);
  dec_fn dec = nullptr; code_fn code = nullptr; end_fn end = nullptr;
  lzma_stream zs;
  bool live = false, finished = false;
  explicit XzSource(std::unique_ptr<Source> s) : InflatingSource(std::move(s)) {
    static const char* const names[] = {"liblzma.so.5", "liblzma.so", nullptr};
    static DlLib lib(names);
    dec = lib.sym<dec_fn>("lzma_stream_decoder");
    code = lib.sym<code_fn>("lzma_code");
    end = lib.sym<end_fn>("lzma_end");
    memset(&zs, 0, sizeof zs);
    if (usable() && dec(&zs, UINT64_MAX, 0x08 /*LZMA_CONCATENATED*/) == 0) live = true;
  }
  ~XzSource() override { if (live) end(&zs); }
  bool usable() const { return dec && code && end; }
  int read(uint8_t* dst, size_t cap, size_t* got) override {
    *got = 0;
    if (!live) return EPROTO;
    if (finished) return 0;
    zs.next_out = dst;
    zs.avail_out = cap;
    while (zs.avail_out == cap) {
      int rc = fill();
      if (rc) return rc;
      zs.next_in = in.data() + in_pos;
      zs.avail_in = in_len - in_pos;
      const int r = code(&zs, inner_eof && zs.avail_in == 0 ? 3 /*LZMA_FINISH*/ : 0 /*LZMA_RUN*/);
      in_pos = in_len - zs.avail_in;
      if (r == 1 /*LZMA_STREAM_END*/) { finished = true; break; }
      if (r != 0 /*LZMA_OK*/) return EPROTO;  // incl. LZMA_BUF_ERROR: truncated input
    }
    *got = cap - zs.avail_out;
    return 0;
  }
};

struct ZstdSource : InflatingSource {
  struct Buf { const void* p; size_t size, pos; };
  struct OBuf { void* p; size_t size, pos; };
  typedef void* (*create_fn)();
  typedef size_t (*free_fn)(void*);
  typedef size_t (*init_fn)(void*);
  typedef size_t (*run_fn)(void*, OBuf*, Buf*);
  typedef unsigned (*iserr_fn)(size_t);
  create_fn create = nullptr; free_fn freef = nullptr; init_fn init = nullptr; run_fn run = nullptr; iserr_fn iserr = nullptr;
  void* ds = nullptr;
  size_t hint = 0;  // last return value: 0 = a frame just ended
  bool started = false;
  explicit ZstdSource(std::unique_ptr<Source> s) : InflatingSource(std::move(s)) {
    static const char* const names[] = {"libzstd.so.1", "libzstd.so", nullptr};
    static DlLib lib(names);
    create = lib.sym<create_fn>("ZSTD_createDStream");
    freef = lib.sym<free_fn>("ZSTD_freeDStream");
    init = lib.sym<init_fn>("ZSTD_initDStream");
    run = lib.sym<run_fn>("ZSTD_decompressStream");
    iserr = lib.sym<iserr_fn>("ZSTD_isError");
    if (usable()) {
      ds = create();
      if (ds && iserr(init(ds))) { freef(ds); ds = nullptr; }
    }
  }
  ~ZstdSource() override { if (ds) freef(ds); }
  bool usable() const { return create && freef && init && run && iserr; }
  int read(uint8_t* dst, size_t cap, size_t* got) override {
    *got = 0;
    if (!ds) return EPROTO;
    OBuf o{dst, cap, 0};
    while (o.pos == 0) {
      int rc = fill();
      if (rc) return rc;
      if (in_pos == in_len && inner_eof) {
        if (started && hint != 0) return EPROTO;  // ends inside a frame
        return 0;
      }
      Buf i{in.data(), in_len, in_pos};
      hint = run(ds, &o, &i);
      if (iserr(hint)) return EPROTO;
      started = true;
      in_pos = i.pos;
    }
    *got = o.pos;
    return 0;
  }
};

// BGZF (bgzip) input: a gzip file whose members are <= 64 KiB blocks that carry their own compressed size in a
// "BC" extra subfield.  To niffler / flate2's MultiGzDecoder it is just a multi-member gzip stream, inflated by
// one thread; because the block boundaries are known WITHOUT inflating, this source reads a batch of blocks and
// inflates them on several threads.  Same output bytes, same error class (EPROTO -> IBU_ERR_NIFFLER); CRC32 and
// ISIZE of every block are verified.  A member that is not a BGZF block hands the rest of the stream to the
// sequential GzSource (mixed files stay correct).
// Bytes without value-initialisation: std::vector<uint8_t>::resize memsets what the inflate threads are about to
// overwrite (24 GB of memset on the thread every batch waits for, at 1e9 records).
struct RawBytes {
  uint8_t* p = nullptr;
  size_t n = 0, cap = 0;
  RawBytes() {}
  RawBytes(const RawBytes&) = delete;
  RawBytes& operator=(const RawBytes&) = delete;
  ~RawBytes() { free(p); }
  bool resize_uninit(size_t want) {
    if (want > cap) {
      uint8_t* q = static_cast<uint8_t*>(realloc(p, want ? want : 1));
      if (!q) return false;
      p = q;
      cap = want;
    }
    n = want;
    return true;
  }
  uint8_t* data() { return p; }
  size_t size() const { return n; }
  void swap(RawBytes& o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); }
};
struct BgzfSource : Source {
  std::unique_ptr<Source> inner;
  std::unique_ptr<Source> fallback;   // sequential inflate once a non-BGZF member shows up
  std::vector<uint8_t> comp;
  RawBytes out;                       // the batch being handed out by read()
  RawBytes next_out;                  // the batch being inflated in the background while `out` is consumed
  std::future<int> next;              // pending background refill (at most one; it alone touches inner/comp/eof/fallback)
  size_t out_pos = 0;
  bool eof = false;
  int pending_err = 0;                // a bad spot was met: reported once what lies in front of it has been handed out
  unsigned threads;
  struct Block { size_t coff, clen, ooff, isize; uint32_t crc; };
  std::unique_ptr<pgz::WorkerPool> pool;                       // threads - 1 workers, started with the first batch
  std::vector<std::unique_ptr<pgz::RawInflater>> raws;         // one decoder (tables, 64 KiB buffer) per thread, kept
  explicit BgzfSource(std::unique_ptr<Source> s) : inner(std::move(s)) {
    const char* e = getenv("IBU_BGZF_THREADS");
    size_t c = e ? (size_t)atol(e) : ibu::inflate_threads();
    threads = (unsigned)(c < 1 ? 1 : (c > 64 ? 64 : c));
    const char* b = getenv("IBU_BGZF_BATCH");          // compressed bytes per batch; 16 ... 128 MiB measured flat (382 ... 409 M records/s)
    batch_comp = b ? (size_t)atol(b) : (size_t)32 << 20;
    if (batch_comp < ((size_t)128 << 10)) batch_comp = (size_t)128 << 10;
  }
  size_t batch_comp;
  int read_exact(uint8_t* dst, size_t n, size_t* got_total) {
    size_t have = 0;
    while (have < n) {
      size_t k = 0;
      int rc = inner->read(dst + have, n - have, &k);
      if (rc) return rc;
      if (k == 0) break;
      have += k;
    }
    *got_total = have;
    return 0;
  }
  static int inflate_block(z_stream* zs, const uint8_t* c, size_t clen, uint8_t* o, size_t isize, uint32_t crc) {
    if (inflateReset(zs) != Z_OK) return EPROTO;
    zs->next_in = const_cast<uint8_t*>(c);
    zs->avail_in = (uInt)clen;
    zs->next_out = o;
    zs->avail_out = (uInt)isize;
    const int rc = inflate(zs, Z_FINISH);
    if (rc != Z_STREAM_END || zs->avail_out != 0 || zs->avail_in != 0) return EPROTO;
    if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), o, (uInt)isize) != crc) return EPROTO;
    return 0;
  }
  ~BgzfSource() override { if (next.valid()) (void)next.get(); }  // never leave the worker running over freed members
  // Input is read in large pieces and the block headers are parsed from memory (one read(2) per 16 MiB instead of three
  // per block: at 1e9 records that was 1.4 M system calls on the thread every batch waits for); blocks are inflated
  // straight out of `comp`, a block that is not whole yet stays for the next batch.
  size_t comp_pos = 0, comp_len = 0;                  // unparsed bytes: comp[comp_pos, comp_len)
  bool inner_eof = false;
  int refill(RawBytes& dst) {
    dst.n = 0;
    std::vector<Block> blocks;
    size_t total_out = 0;
    const size_t kBatchComp = batch_comp;
    const size_t kPadBytes = 512;
    if (comp_pos) {                                   // leftover of the previous batch to the front
      memmove(comp.data(), comp.data() + comp_pos, comp_len - comp_pos);
      comp_len -= comp_pos;
      comp_pos = 0;
    }
    if (comp.size() < kBatchComp + (128u << 10) + kPadBytes) comp.resize(kBatchComp + (128u << 10) + kPadBytes);
    while (!inner_eof && comp_len < kBatchComp + (128u << 10)) {
      size_t got = 0;
      int rc = inner->read(comp.data() + comp_len, kBatchComp + (128u << 10) - comp_len, &got);
      if (rc) return rc;
      if (got == 0) inner_eof = true;
      comp_len += got;
    }
    memset(comp.data() + comp_len, 0, kPadBytes);     // the symbol loop may read (not use) a few bytes behind a block
    // A bad spot (a block that is cut off, a size that cannot be) does not take the blocks in front of it with it: they
    // are inflated and handed out like any batch — a sequential inflate delivers them, too — and the error (pending_err)
    // comes with the read() behind them.
    while ((comp_pos < comp_len || inner_eof) && !pending_err) {
      const size_t avail = comp_len - comp_pos;
      if (avail == 0) { eof = true; break; }
      const uint8_t* hd = comp.data() + comp_pos;
      if (avail < 12) { if (inner_eof) pending_err = EPROTO; break; }
      const bool bgzf_like = hd[0] == 0x1f && hd[1] == 0x8b && hd[2] == 8 && (hd[3] & 4);
      const size_t xlen = bgzf_like ? (size_t)(hd[10] | (hd[11] << 8)) : 0;
      size_t bsize = 0;
      if (bgzf_like) {
        if (avail < 12 + xlen) { if (inner_eof) pending_err = EPROTO; break; }
        const uint8_t* extra = hd + 12;
        for (size_t p = 0; p + 4 <= xlen;) {
          const size_t slen = (size_t)(extra[p + 2] | (extra[p + 3] << 8));
          if (extra[p] == 'B' && extra[p + 1] == 'C' && slen == 2 && p + 6 <= xlen) bsize = (size_t)(extra[p + 4] | (extra[p + 5] << 8)) + 1;
          p += 4 + slen;
        }
      }
      if (!bgzf_like || bsize < 12 + 2 + xlen + 8) {
        // not a BGZF block: everything not yet parsed goes back in front of the inner source and the sequential
        // decoder takes over from here
        std::unique_ptr<PrefixSource> ps(new PrefixSource);
        ps->pre.assign(hd, hd + avail);
        ps->inner = std::move(inner);
        fallback.reset(new GzSource(std::move(ps)));
        comp_pos = comp_len;
        break;
      }
      if (avail < bsize) { if (inner_eof) pending_err = EPROTO; break; }  // the block is not whole yet / the stream ends inside it
      const uint8_t* tr = hd + bsize - 8;
      Block b;
      b.coff = comp_pos + 12 + xlen;
      b.clen = bsize - 12 - xlen - 8;
      b.crc = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
      b.isize = (size_t)tr[4] | ((size_t)tr[5] << 8) | ((size_t)tr[6] << 16) | ((size_t)tr[7] << 24);
      if (b.isize > 65536) { pending_err = EPROTO; break; }
      b.ooff = total_out;
      total_out += b.isize;
      blocks.push_back(b);
      comp_pos += bsize;
    }
    if (pending_err) eof = true;                       // nothing is read behind a bad spot
    if (!dst.resize_uninit(total_out)) return ENOMEM;
    if (blocks.empty()) return 0;
    const unsigned nt = blocks.size() < threads ? (unsigned)blocks.size() : threads;
    std::vector<int> rcs(nt, 0);
    std::vector<size_t> bad_at(nt, ~(size_t)0);        // per task: the first block that did not inflate
    auto work = [&](unsigned t) {
      const bool use_zlib = getenv("IBU_BGZF_ZLIB") != nullptr;   // A/B and second witness: zlib's inflate + crc32
      z_stream zs;
      memset(&zs, 0, sizeof zs);
      if (use_zlib && inflateInit2(&zs, -15) != Z_OK) { rcs[t] = EPROTO; return; }
      try {                                            // a worker thread must not throw
      if (!raws[t]) raws[t].reset(new pgz::RawInflater);
      pgz::RawInflater& raw = *raws[t];
      for (size_t i = t; i < blocks.size(); i += nt) {
        const Block& b = blocks[i];
        if (b.isize == 0 && b.clen <= 2) continue;  // empty block (the EOF marker)
        int rc;
        if (use_zlib) rc = inflate_block(&zs, comp.data() + b.coff, b.clen, dst.data() + b.ooff, b.isize, b.crc);
        else {
          uint32_t crc = 0;
          rc = raw.inflate(comp.data() + b.coff, b.clen, dst.data() + b.ooff, b.isize, &crc);
          if (rc == 0 && crc != b.crc) rc = EPROTO;
        }
        if (rc) { rcs[t] = rc; bad_at[t] = i; break; }
      }
      } catch (...) {
        rcs[t] = ENOMEM;
      }
      if (use_zlib) inflateEnd(&zs);
    };
    if (raws.size() < threads) raws.resize(threads);
    if (!pool) pool.reset(new pgz::WorkerPool(threads - 1));
    pool->run(nt, work);   // never throws: the share of a worker that could not be started is inflated by the others
    size_t first_bad = ~(size_t)0;
    for (unsigned t = 0; t < nt; ++t) {
      if (rcs[t] && bad_at[t] == ~(size_t)0) return rcs[t];          // not a block's fault (no memory, no zlib state)
      if (bad_at[t] < first_bad) first_bad = bad_at[t];
    }
    if (first_bad != ~(size_t)0) {                       // blocks in front of the first bad one are good and go out
      dst.n = blocks[first_bad].ooff;
      pending_err = EPROTO;
      eof = true;                                      // nothing is read behind the bad spot
    }
    return 0;
  }
  // refill() grows vectors: bad_alloc must not leave as an exception (this is called under extern "C" entry points)
  int refill_noexcept() {
    try { return refill(next_out); }
    catch (const std::bad_alloc&) { return ENOMEM; }
    catch (...) { return EIO; }
  }
  bool start_refill() {
    try {
      next = std::async(std::launch::async, [this] { return refill_noexcept(); });
      return true;
    } catch (...) {  // std::system_error (EAGAIN) / bad_alloc from the thread start
      return false;
    }
  }
  int read(uint8_t* dst, size_t cap, size_t* got) override {
    *got = 0;
    for (;;) {
      if (out_pos < out.size()) {
        const size_t k = out.size() - out_pos < cap ? out.size() - out_pos : cap;
        memcpy(dst, out.data() + out_pos, k);
        out_pos += k;
        *got = k;
        return 0;
      }
      // current batch drained: take the one inflated in the background (or start the first), then immediately start
      // the next so that reading + inflating batch k+1 overlaps the caller's consumption of batch k
      if (!next.valid()) {
        if (pending_err) return pending_err;   // everything in front of the bad spot has been handed out
        if (fallback) return fallback->read(dst, cap, got);
        if (eof) return 0;
        if (!start_refill()) {   // no thread to be had: inflate the batch on this one
          const int rc = refill_noexcept();
          if (rc) return rc;
          out.swap(next_out);
          out_pos = 0;
          continue;
        }
      }
      int rc;
      try { rc = next.get(); }  // synchronises with everything the worker wrote (eof, fallback, next_out)
      catch (const std::bad_alloc&) { rc = ENOMEM; }
      catch (...) { rc = EIO; }
      if (rc) return rc;
      out.swap(next_out);
      out_pos = 0;
      if (!eof && !fallback) (void)start_refill();  // failure: the next round inflates inline
    }
  }
};


// Ordinary (non-BGZF) gzip input, inflated on several threads: pgzip.hpp.  Same output bytes and error class as
// GzSource (the sequential zlib path, still used with IBU_NO_PARALLEL_GZIP=1 or on a one-core host); the next batch is
// decoded in the background while the current one is handed out.  IBU_PGZ_THREADS / IBU_PGZ_CHUNK (bytes of compressed
// input per thread and batch, default 4 MiB) are test knobs.
struct ParGzSource : Source {
  std::unique_ptr<Source> inner;
  std::unique_ptr<pgz::ParallelGunzip> dec;
  std::vector<pgz::Span> out, next_out;              // pieces of the decoder's chunk buffers (valid while the NEXT batch decodes)
  std::future<int> next;
  size_t span_i = 0, span_off = 0;
  bool eof = false, next_eof = false;
  static unsigned env_threads() {
    const char* e = getenv("IBU_PGZ_THREADS");
    size_t c = e ? (size_t)atol(e) : ibu::inflate_threads();
    return (unsigned)(c < 1 ? 1 : (c > 64 ? 64 : c));
  }
  static size_t env_chunk() {
    const char* e = getenv("IBU_PGZ_CHUNK");
    return e ? (size_t)atol(e) : (size_t)4 << 20;   // measured on the 32-thread box: 0.5 / 1 / 2 / 4 / 8 / 16 MiB -> 253 / 299 / 330 / 356 / 341 / 327 M records/s
  }
  explicit ParGzSource(std::unique_ptr<Source> s) : inner(std::move(s)) {
    Source* in = inner.get();
    dec.reset(new pgz::ParallelGunzip([in](uint8_t* d, size_t cap, size_t* got) { return in->read(d, cap, got); }, env_threads(), env_chunk()));
  }
  ~ParGzSource() override {
    if (next.valid()) (void)next.get();
    if (getenv("IBU_PGZ_TRACE")) {                     // where the time of the parallel inflate went (wall seconds per phase)
      const pgz::Stats& t = dec->stats();
      fprintf(stderr, "[pgzip] in %llu B out %llu B batches %llu chunks accepted %llu discarded %llu no-candidate %llu markers %llu | "
              "read %.3f (join %.3f, helper busy %.3f) find %.3f decode %.3f windows %.3f patch+crc %.3f carry %.3f s\n",
              (unsigned long long)t.bytes_in, (unsigned long long)t.bytes_out, (unsigned long long)t.batches,
              (unsigned long long)t.chunks_accepted, (unsigned long long)t.chunks_discarded, (unsigned long long)t.candidates_missing,
              (unsigned long long)t.marker_symbols, t.s_read, t.s_join_wait, t.s_helper_read, t.s_find, t.s_decode, t.s_windows, t.s_patch_crc, t.s_carry);
    }
  }
  int refill_noexcept() {
    try { return dec->next_batch(next_out, &next_eof); }
    catch (const std::bad_alloc&) { return ENOMEM; }
    catch (...) { return EIO; }
  }
  bool start_refill() {
    try {
      next = std::async(std::launch::async, [this] { return refill_noexcept(); });
      return true;
    } catch (...) {
      return false;
    }
  }
  int read(uint8_t* dst, size_t cap, size_t* got) override {
    *got = 0;
    for (;;) {
      if (span_i < out.size()) {                       // hand out pieces until `cap` is full or the batch is used up
        size_t done = 0;
        while (span_i < out.size() && done < cap) {
          const pgz::Span& sp = out[span_i];
          const size_t k = sp.n - span_off < cap - done ? sp.n - span_off : cap - done;
          const uint8_t* src = sp.p + span_off;
          uint8_t* d = dst + done;
          if (k >= ((size_t)4 << 20)) {                // one thread copies ~8 GB/s, the inflate delivers 5-6
            const size_t per = ((k / 4) + 4095) & ~(size_t)4095;
            run_pieces(4, [=](unsigned i) {
              const size_t off = (size_t)i * per;
              if (off < k) memcpy(d + off, src + off, off + per < k ? per : k - off);
            });
          } else if (k) {
            memcpy(d, src, k);
          }
          done += k;
          span_off += k;
          if (span_off == sp.n) { ++span_i; span_off = 0; }
        }
        *got = done;
        return 0;
      }
      if (eof) return 0;
      int rc;
      if (next.valid()) {
        try { rc = next.get(); }
        catch (const std::bad_alloc&) { rc = ENOMEM; }
        catch (...) { rc = EIO; }
      } else {
        rc = refill_noexcept();                        // the first batch (or no thread to be had): decode it here
      }
      if (rc) return rc;
      out.swap(next_out);                              // the batch just consumed is dead: its buffers are the ones the refill
      span_i = span_off = 0;                           // started below decodes into
      eof = next_eof;
      if (!eof) (void)start_refill();                  // failure: the next round decodes inline
    }
  }
};

}  // namespace

// ------------------------------------------------------------------------------------------
// Writer (src/io/writer.rs)
// ------------------------------------------------------------------------------------------
struct ibu_writer {
  std::unique_ptr<Sink> inner;
  MemSink* mem = nullptr;  // set when inner is a Vec<u8>
  std::vector<uint8_t> buffer;
  size_t pos = 0;
  uint64_t records_written = 0;
};

namespace {

int32_t writer_flush_buffer(ibu_writer* w) {  // writer.rs:220-226
  if (w->pos > 0) {
    int e = w->inner->write_all(w->buffer.data(), w->pos);
    if (e) return err_io(e, "write");
    w->pos = 0;
  }
  return IBU_OK;
}

int32_t writer_write_slice(ibu_writer* w, const uint8_t* bytes, size_t len) {  // writer.rs:321-351
  const size_t num_records = len / IBU_RECORD_SIZE;
  if (len > w->buffer.size()) {  // larger than the buffer: flush what is pending, then write through
    int32_t rc = writer_flush_buffer(w);
    if (rc) return rc;
    int e = w->inner->write_all(bytes, len);
    if (e) return err_io(e, "write");
    w->records_written += num_records;
    return IBU_OK;
  }
  while (len) {
    const size_t room = w->buffer.size() - w->pos;
    const size_t k = len < room ? len : room;
    memcpy(w->buffer.data() + w->pos, bytes, k);
    w->pos += k;
    bytes += k;
    len -= k;
    if (w->pos >= w->buffer.size()) {
      int32_t rc = writer_flush_buffer(w);
      if (rc) return rc;
    }
  }
  w->records_written += num_records;
  return IBU_OK;
}

int32_t writer_make(std::unique_ptr<Sink> sink, MemSink* mem, const ibu_header_t* header, ibu_writer_t** out) {
  if (!out) return err_arg("out is NULL");
  std::unique_ptr<ibu_writer> w(new (std::nothrow) ibu_writer);
  if (!w) return err_io(ENOMEM, "alloc");
  w->inner = std::move(sink);
  w->mem = mem;
  if (header) {  // Writer::new: header goes out at once, unvalidated (Q2)  writer.rs:129-143
    int e = w->inner->write_all(reinterpret_cast<const uint8_t*>(header), IBU_HEADER_SIZE);
    if (e) return err_io(e, "write header");
  }
  w->buffer.assign(IBU_DEFAULT_BUFFER_SIZE, 0);
  *out = w.release();
  return IBU_OK;
}

}  // namespace

// The walk over BGZF block headers (the rule of BgzfSource::refill above, without the inflating): see include/ibu_hip.h.
extern "C" int32_t ibu_bgzf_scan(const uint8_t* buf, size_t len, int32_t final, ibu_inflate_block_t* blocks, size_t cap, size_t* n_blocks,
                                 size_t* consumed, uint64_t* out_bytes) {
  if (!n_blocks || !consumed) return err_arg("n_blocks / consumed is NULL");
  *n_blocks = 0;
  *consumed = 0;
  if (out_bytes) *out_bytes = 0;
  if ((!buf && len) || (!blocks && cap)) return err_arg("NULL argument");
  size_t pos = 0, nb = 0;
  uint64_t total = 0;
  int32_t rc = IBU_OK;
  while (pos < len && nb < cap) {
    const size_t avail = len - pos;
    const uint8_t* hd = buf + pos;
    if (avail < 12) { if (final) rc = err_niffler("the stream ends inside a BGZF block header"); break; }
    const bool bgzf_like = hd[0] == 0x1f && hd[1] == 0x8b && hd[2] == 8 && (hd[3] & 4);
    if (!bgzf_like) { rc = err_niffler("not a BGZF block (a gzip member without the BC extra field)"); break; }
    const size_t xlen = (size_t)(hd[10] | (hd[11] << 8));
    if (avail < 12 + xlen) { if (final) rc = err_niffler("the stream ends inside a BGZF block header"); break; }
    size_t bsize = 0;
    for (size_t p = 0; p + 4 <= xlen;) {
      const uint8_t* extra = hd + 12;
      const size_t slen = (size_t)(extra[p + 2] | (extra[p + 3] << 8));
      if (extra[p] == 'B' && extra[p + 1] == 'C' && slen == 2 && p + 6 <= xlen) bsize = (size_t)(extra[p + 4] | (extra[p + 5] << 8)) + 1;
      p += 4 + slen;
    }
    if (bsize < 12 + 2 + xlen + 8) { rc = err_niffler("not a BGZF block (no usable BC extra field)"); break; }
    if (avail < bsize) { if (final) rc = err_niffler("the stream ends inside a BGZF block"); break; }
    const uint8_t* tr = hd + bsize - 8;
    ibu_inflate_block_t b;
    b.comp_offset = pos + 12 + xlen;
    b.comp_len = (uint32_t)(bsize - 12 - xlen - 8);
    b.crc32 = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
    b.out_len = (uint32_t)tr[4] | ((uint32_t)tr[5] << 8) | ((uint32_t)tr[6] << 16) | ((uint32_t)tr[7] << 24);
    b.reserved = 0;
    if (b.out_len > 65536) { rc = err_niffler("a BGZF block announces more than 64 KiB"); break; }
    b.out_offset = (int64_t)total;
    total += b.out_len;
    blocks[nb++] = b;
    pos += bsize;
  }
  *n_blocks = nb;
  *consumed = pos;
  if (out_bytes) *out_bytes = total;
  return rc;
}

extern "C" int32_t ibu_writer_open_callback(ibu_write_fn wr, ibu_flush_fn fl, void* user, const ibu_header_t* header,
                                            ibu_writer_t** out) {
  if (!wr) return err_arg("write callback is NULL");
  return writer_make(std::unique_ptr<Sink>(new CallbackSink(wr, fl, user)), nullptr, header, out);
}
extern "C" int32_t ibu_writer_open_path(const char* path, const ibu_header_t* header, ibu_writer_t** out) {
  if (!path) return err_arg("path is NULL");
  int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);  // File::create  writer.rs:557
  if (fd < 0) return err_io(errno, path);
  return writer_make(std::unique_ptr<Sink>(new FdSink(fd, true)), nullptr, header, out);
}
extern "C" int32_t ibu_writer_open_fd(int fd, const ibu_header_t* header, ibu_writer_t** out) {
  if (fd < 0) return err_arg("fd < 0");
  return writer_make(std::unique_ptr<Sink>(new FdSink(fd, false)), nullptr, header, out);
}
extern "C" int32_t ibu_writer_open_mem(const ibu_header_t* header, ibu_writer_t** out) {
  MemSink* m = new MemSink;
  return writer_make(std::unique_ptr<Sink>(m), m, header, out);
}

extern "C" int32_t ibu_writer_write_record(ibu_writer_t* w, const ibu_record_t* r) {  // writer.rs:260-273
  if (!w || !r) return err_arg("writer or record is NULL");
  if (w->pos + IBU_RECORD_SIZE > w->buffer.size()) {
    int32_t rc = writer_flush_buffer(w);
    if (rc) return rc;
  }
  memcpy(w->buffer.data() + w->pos, r, IBU_RECORD_SIZE);
  w->pos += IBU_RECORD_SIZE;
  w->records_written += 1;
  return IBU_OK;
}
extern "C" int32_t ibu_writer_write_batch(ibu_writer_t* w, const ibu_record_t* recs, size_t n) {  // :315-318
  if (!w || (!recs && n)) return err_arg("writer or records is NULL");
  return writer_write_slice(w, reinterpret_cast<const uint8_t*>(recs), n * IBU_RECORD_SIZE);
}
int32_t ibu::writer_write_bytes(ibu_writer_t* w, const uint8_t* bytes, size_t len) {
  return writer_write_slice(w, bytes, len);
}
extern "C" int32_t ibu_writer_ingest(ibu_writer_t* w, ibu_writer_t* other) {  // writer.rs:477-482
  if (!w || !other || !other->mem) return err_arg("ingest needs a memory-backed (Vec<u8>) source writer");
  int32_t rc = writer_flush_buffer(other);
  if (rc) return rc;
  rc = writer_write_slice(w, other->mem->v.data(), other->mem->v.size());  // Q15: header bytes included if any
  if (rc) return rc;
  other->mem->v.clear();
  return IBU_OK;
}
extern "C" int32_t ibu_writer_finish(ibu_writer_t* w) {  // writer.rs:429-433
  if (!w) return err_arg("writer is NULL");
  int32_t rc = writer_flush_buffer(w);
  if (rc) return rc;
  int e = w->inner->flush();
  if (e) return err_io(e, "flush");
  return IBU_OK;
}
extern "C" uint64_t ibu_writer_records_written(const ibu_writer_t* w) { return w ? w->records_written : 0; }
extern "C" int32_t ibu_writer_mem_view(const ibu_writer_t* w, const uint8_t** data, size_t* len) {
  if (!w || !w->mem || !data || !len) return err_arg("not a memory-backed writer");
  *data = w->mem->v.data();
  *len = w->mem->v.size();
  return IBU_OK;
}
extern "C" int32_t ibu_writer_into_inner(ibu_writer_t* w, uint8_t** data, size_t* len) {  // writer.rs:507-511
  if (!w) return err_arg("writer is NULL");
  if (data) *data = nullptr;
  if (len) *len = 0;
  int32_t rc = IBU_OK;
  if (w->mem && data && len) {
    size_t n = w->mem->v.size();
    uint8_t* p = static_cast<uint8_t*>(malloc(n ? n : 1));
    if (!p) rc = err_io(ENOMEM, "alloc");
    else {
      memcpy(p, w->mem->v.data(), n);
      *data = p;
      *len = n;
    }
  }
  delete w;  // no flush: buffered records are dropped exactly as ManuallyDrop does
  return rc;
}
extern "C" void ibu_writer_close(ibu_writer_t* w) {  // Drop  writer.rs:519-523
  if (!w) return;
  (void)ibu_writer_finish(w);
  delete w;
}

// ------------------------------------------------------------------------------------------
// Reader (src/io/reader.rs)
// ------------------------------------------------------------------------------------------
struct ibu_reader {
  std::unique_ptr<Source> inner;
  std::string bgzf_path;           // ibu_reader_open_path of a BGZF file: its path (the device processors may read the file themselves)
  bool compressed = false;
  std::vector<uint8_t> buffer;  // capacity DEFAULT_BUFFER_SIZE
  ibu_header_t header;
  size_t pos = 0, cap = 0;
  uint64_t bytes_read = 0;
  bool eof = false;
};

namespace {

int32_t src_error(const ibu_reader* r, int e, const char* what) {
  if (r && r->compressed && e == EPROTO) return err_niffler("corrupt or truncated compressed stream");
  return err_io(e, what);
}

int32_t reader_make(std::unique_ptr<Source> src, bool compressed, ibu_reader_t** out) {  // reader.rs:152-176
  if (!out) return err_arg("out is NULL");
  std::unique_ptr<ibu_reader> r(new (std::nothrow) ibu_reader);
  if (!r) return err_io(ENOMEM, "alloc");
  r->inner = std::move(src);
  r->compressed = compressed;
  uint8_t hb[IBU_HEADER_SIZE];
  size_t have = 0;
  while (have < IBU_HEADER_SIZE) {  // read_exact
    size_t got = 0;
    int e = r->inner->read(hb + have, IBU_HEADER_SIZE - have, &got);
    if (e) return src_error(r.get(), e, "read header");
    if (got == 0) return err_io(0, "read header");  // UnexpectedEof is an io::Error
    have += got;
  }
  memcpy(&r->header, hb, IBU_HEADER_SIZE);  // pod_read_unaligned
  int32_t rc = ibu_header_validate(&r->header);
  if (rc) return rc;
  r->buffer.resize(IBU_DEFAULT_BUFFER_SIZE);
  r->bytes_read = IBU_HEADER_SIZE;
  *out = r.release();
  return IBU_OK;
}

// niffler::send::get_reader: sniff the first bytes, wrap in a decoder when compressed.
int32_t reader_make_sniffed(std::unique_ptr<Source> src, ibu_reader_t** out) {
  std::unique_ptr<PrefixSource> ps(new PrefixSource);
  ps->pre.resize(18);  // 5 bytes decide the format (niffler); 18 tell a BGZF block from a plain gzip member
  size_t have = 0;
  while (have < 18) {
    size_t got = 0;
    int e = src->read(ps->pre.data() + have, 18 - have, &got);
    if (e) return err_io(e, "read");
    if (got == 0) break;
    have += got;
  }
  ps->pre.resize(have);
  if (have < 5) return err_niffler("file too short to sniff its format");
  const uint8_t* m = ps->pre.data();
  ps->inner = std::move(src);
  if (m[0] == 0x1f && m[1] == 0x8b) {
    const bool bgzf = have >= 18 && m[2] == 8 && (m[3] & 4) && m[12] == 'B' && m[13] == 'C' && m[14] == 2 && m[15] == 0;
    if (bgzf && !getenv("IBU_NO_PARALLEL_BGZF"))
      return reader_make(std::unique_ptr<Source>(new BgzfSource(std::move(ps))), true, out);
    if (!getenv("IBU_NO_PARALLEL_GZIP") && ParGzSource::env_threads() > 1)
      return reader_make(std::unique_ptr<Source>(new ParGzSource(std::move(ps))), true, out);
    return reader_make(std::unique_ptr<Source>(new GzSource(std::move(ps))), true, out);
  }
  if (m[0] == 0x42 && m[1] == 0x5a && m[2] == 0x68) {  // "BZh"
    std::unique_ptr<BzSource> z(new BzSource(std::move(ps)));
    if (!z->usable()) return err_niffler("bzip2 input: libbz2.so.1 is not available on this host");
    return reader_make(std::move(z), true, out);
  }
  if (m[0] == 0xfd && m[1] == 0x37 && m[2] == 0x7a && m[3] == 0x58 && m[4] == 0x5a) {  // FD "7zXZ"
    std::unique_ptr<XzSource> z(new XzSource(std::move(ps)));
    if (!z->usable()) return err_niffler("xz input: liblzma.so.5 is not available on this host");
    return reader_make(std::move(z), true, out);
  }
  if (m[0] == 0x28 && m[1] == 0xb5 && m[2] == 0x2f && m[3] == 0xfd) {  // zstd frame magic
    std::unique_ptr<ZstdSource> z(new ZstdSource(std::move(ps)));
    if (!z->usable()) return err_niffler("zstd input: libzstd.so.1 is not available on this host");
    return reader_make(std::move(z), true, out);
  }
  return reader_make(std::move(ps), false, out);
}

}  // namespace

extern "C" int32_t ibu_reader_open_callback(ibu_read_fn rd, void* user, ibu_reader_t** out) {
  if (!rd) return err_arg("read callback is NULL");
  return reader_make(std::unique_ptr<Source>(new CallbackSource(rd, user)), false, out);
}
extern "C" int32_t ibu_reader_open_mem(const uint8_t* data, size_t len, ibu_reader_t** out) {
  if (!data && len) return err_arg("data is NULL");
  return reader_make(std::unique_ptr<Source>(new MemSource(data, len)), false, out);
}
extern "C" int32_t ibu_reader_open_path(const char* path, ibu_reader_t** out) {  // reader.rs:345-352
  if (!path) return err_arg("path is NULL");
  int fd = ::open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return err_io(errno, path);
  const int32_t rc = reader_make_sniffed(std::unique_ptr<Source>(new FdSource(fd, true)), out);
  if (rc == IBU_OK && dynamic_cast<BgzfSource*>((*out)->inner.get())) {
    try { (*out)->bgzf_path = path; } catch (...) {}     // (no memory for the name: the reader works without it)
  }
  return rc;
}
// The path of a BGZF file behind a reader nothing has been read from yet beyond the header (else NULL): ibu_reader_process_device may then
// load the file itself, the compressed bytes over the link and the blocks inflated on the device (stream.cpp).
const char* ibu::reader_bgzf_path_if_untouched(const ibu_reader_t* r) {
  if (!r || r->bgzf_path.empty() || r->eof || r->pos != r->cap || r->bytes_read != IBU_HEADER_SIZE) return nullptr;
  return r->bgzf_path.c_str();
}
// ... and once it has: the reader stands at its end, as if every record had been read through it.
void ibu::reader_set_drained(ibu_reader_t* r, uint64_t records) {
  r->eof = true;
  r->pos = r->cap = 0;
  r->bytes_read = IBU_HEADER_SIZE + IBU_RECORD_SIZE * records;
  r->inner.reset(new MemSource(nullptr, 0));             // (the inflate threads of the host path are not needed any more)
}
extern "C" int32_t ibu_reader_open_fd(int fd, ibu_reader_t** out) {  // reader.rs:389-396
  if (fd < 0) return err_arg("fd < 0");
  return reader_make_sniffed(std::unique_ptr<Source>(new FdSource(fd, false)), out);
}
extern "C" int32_t ibu_reader_header(const ibu_reader_t* r, ibu_header_t* out) {
  if (!r || !out) return err_arg("reader or out is NULL");
  *out = r->header;
  return IBU_OK;
}
extern "C" int32_t ibu_reader_read_batch(ibu_reader_t* r, int32_t* has_data) {  // reader.rs:218-242
  if (!r) return err_arg("reader is NULL");
  size_t read = 0;
  while (read < r->buffer.size()) {
    size_t got = 0;
    int e = r->inner->read(r->buffer.data() + read, r->buffer.size() - read, &got);
    if (e) return src_error(r, e, "read");
    if (got == 0) break;
    read += got;
  }
  if (read % IBU_RECORD_SIZE != 0)  // Q8: the whole refill is rejected
    return err_truncated(r->bytes_read + (read - read % IBU_RECORD_SIZE));
  r->pos = 0;
  r->cap = read / IBU_RECORD_SIZE;
  r->bytes_read += read;
  if (has_data) *has_data = read > 0;
  return IBU_OK;
}
extern "C" int32_t ibu_reader_next(ibu_reader_t* r, ibu_record_t* out, int32_t* got) {  // reader.rs:279-306
  if (!r || !out || !got) return err_arg("reader, out or got is NULL");
  *got = 0;
  if (r->eof) return IBU_OK;
  if (r->pos >= r->cap) {
    int32_t has = 0;
    int32_t rc = ibu_reader_read_batch(r, &has);
    if (rc) return rc;  // Q9: eof stays false
    if (!has) r->eof = true;
  }
  if (r->eof) return IBU_OK;
  memcpy(out, r->buffer.data() + IBU_RECORD_SIZE * r->pos, IBU_RECORD_SIZE);
  r->pos += 1;
  *got = 1;
  return IBU_OK;
}
extern "C" int32_t ibu_reader_buffered(ibu_reader_t* r, const ibu_record_t** recs, size_t* n) {
  if (!r || !recs || !n) return err_arg("reader, recs or n is NULL");
  *recs = reinterpret_cast<const ibu_record_t*>(r->buffer.data() + IBU_RECORD_SIZE * r->pos);
  *n = r->cap - r->pos;
  return IBU_OK;
}
extern "C" int32_t ibu_reader_consume(ibu_reader_t* r, size_t n) {
  if (!r || n > r->cap - r->pos) return err_arg("consume beyond the buffered records");
  r->pos += n;
  return IBU_OK;
}
int32_t ibu::reader_read_direct(ibu_reader_t* r, uint8_t* dst, size_t cap_bytes, size_t* got_bytes, bool* eof) {
  *got_bytes = 0;
  *eof = false;
  if (!r || r->pos < r->cap) return err_arg("reader_read_direct: the reader's own buffer still holds records");
  cap_bytes -= cap_bytes % IBU_RECORD_SIZE;
  size_t read = 0;
  while (read < cap_bytes) {
    size_t got = 0;
    int e = r->inner->read(dst + read, cap_bytes - read, &got);
    if (e) {                                           // (reader.rs:225-230: the refill under way is lost with the error, the ones in front of it were yielded)
      *got_bytes = read - read % IBU_RECORD_SIZE;
      return src_error(r, e, "read");
    }
    if (got == 0) { *eof = true; break; }
    read += got;
  }
  if (read % IBU_RECORD_SIZE != 0) {                   // only at the end of the stream: it ends inside a record
    *got_bytes = read - read % IBU_RECORD_SIZE;        // the complete records in front of the cut (the pull stream hands out their whole refills)
    return err_truncated(r->bytes_read + *got_bytes);
  }
  r->bytes_read += read;
  *got_bytes = read;
  return IBU_OK;
}
extern "C" uint64_t ibu_reader_bytes_read(const ibu_reader_t* r) { return r ? r->bytes_read : 0; }
extern "C" void ibu_reader_close(ibu_reader_t* r) { delete r; }

// ------------------------------------------------------------------------------------------
// load_to_vec (src/io/reader.rs:510-535)
// ------------------------------------------------------------------------------------------
int32_t ibu::open_plain_file(const char* path, int* fd_out, ibu_header_t* header, size_t* n_records) {
  int fd = ::open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return err_io(errno, path);
  uint8_t hb[IBU_HEADER_SIZE];
  size_t have = 0;
  while (have < IBU_HEADER_SIZE) {
    ssize_t k = ::read(fd, hb + have, IBU_HEADER_SIZE - have);
    if (k < 0 && errno == EINTR) continue;
    if (k <= 0) {
      int e = k < 0 ? errno : 0;
      close(fd);
      return err_io(e, "read header");
    }
    have += (size_t)k;
  }
  memcpy(header, hb, IBU_HEADER_SIZE);
  int32_t rc = ibu_header_validate(header);
  if (rc) {
    close(fd);
    return rc;
  }
  struct stat st;
  if (fstat(fd, &st)) {
    int e = errno;
    close(fd);
    return err_io(e, "metadata");
  }
  const size_t data_size = (size_t)st.st_size - IBU_HEADER_SIZE;
  if (data_size % IBU_RECORD_SIZE != 0) {
    close(fd);
    return err_map_size();
  }
  *n_records = data_size / IBU_RECORD_SIZE;
  *fd_out = fd;
  return IBU_OK;
}

extern "C" int32_t ibu_load_to_vec(const char* path, ibu_header_t* header, ibu_record_t** records, size_t* n) {
  if (!path || !header || !records || !n) return err_arg("NULL argument");
  int fd = -1;
  size_t num = 0;
  int32_t rc = open_plain_file(path, &fd, header, &num);
  if (rc) return rc;
  // vec![Record::default(); n] (reader.rs:528).  Large vectors are 2 MiB aligned and advised huge: the cost of this
  // function is first-touch page faults, and a huge page takes one fault for 512 small ones where THP allows it.
  ibu_record_t* v = nullptr;
  const size_t bytes_needed = (num ? num : 1) * sizeof(ibu_record_t);
  if (bytes_needed >= ((size_t)8 << 20)) {
    const size_t huge = (size_t)2 << 20;
    void* pv = nullptr;
    if (posix_memalign(&pv, huge, (bytes_needed + huge - 1) & ~(huge - 1)) == 0) {
      (void)madvise(pv, (bytes_needed + huge - 1) & ~(huge - 1), MADV_HUGEPAGE);
      v = static_cast<ibu_record_t*>(pv);  // every byte is overwritten by the read below (no zero-fill pass needed)
    }
  } else {
    v = static_cast<ibu_record_t*>(calloc(num ? num : 1, sizeof(ibu_record_t)));
  }
  if (!v) {
    close(fd);
    return err_io(ENOMEM, "alloc");
  }
  // read_exact.  Same bytes as the reference's single read_exact (reader.rs:531-532); large files are split over
  // a few threads with pread so the page faults of the fresh allocation and the kernel copies overlap.
  uint8_t* p = reinterpret_cast<uint8_t*>(v);
  const size_t total = num * IBU_RECORD_SIZE;
  auto read_range = [&](size_t off, size_t len) -> int {
    while (len) {
      ssize_t k = ::pread(fd, p + off, len, (off_t)(IBU_HEADER_SIZE + off));
      if (k < 0 && errno == EINTR) continue;
      if (k < 0) return errno;
      if (k == 0) return EIO;  // the file shrank underneath us
      off += (size_t)k;
      len -= (size_t)k;
    }
    return 0;
  };
  int err = 0;
  const size_t kPar = (size_t)64 << 20;
  if (total <= kPar) {
    err = read_range(0, total);
  } else {
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = hw ? (hw < 8 ? hw : 8) : 4;
    if (nt > total / kPar + 1) nt = total / kPar + 1;
    const size_t per = ((total / nt) + 4095) & ~(size_t)4095;
    int rcs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    run_pieces((unsigned)nt, [&](unsigned i) {   // never throws; nt <= 8
      const size_t off = (size_t)i * per;
      if (off >= total) return;
      rcs[i] = read_range(off, off + per < total ? per : total - off);
    });
    for (int r : rcs)
      if (r && !err) err = r;
  }
  if (err) {
    free(v);
    close(fd);
    return err_io(err == EIO ? 0 : err, "read records");
  }
  close(fd);
  *records = v;
  *n = num;
  return IBU_OK;
}

// ------------------------------------------------------------------------------------------
// MmapReader (src/io/mmap.rs)
// ------------------------------------------------------------------------------------------
namespace {
struct Mapping {  // Arc<Mmap>
  uint8_t* base = nullptr;
  size_t len = 0;
  std::atomic<int> refs{1};
};
}  // namespace
struct ibu_mmap {
  Mapping* map;
  ibu_header_t header;
  size_t len;
};

extern "C" int32_t ibu_mmap_open(const char* path, ibu_mmap_t** out) {  // mmap.rs:143-161
  if (!path || !out) return err_arg("NULL argument");
  int fd = ::open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return err_io(errno, path);
  struct stat st;
  if (fstat(fd, &st)) {
    int e = errno;
    close(fd);
    return err_io(e, "metadata");
  }
  const size_t flen = (size_t)st.st_size;
  if (flen < IBU_HEADER_SIZE) {  // the reference panics slicing map[0..32]; an error here
    close(fd);
    return err_arg("file shorter than the 32-byte header (the reference panics here)");
  }
  void* p = mmap(nullptr, flen, PROT_READ, MAP_PRIVATE, fd, 0);
  int e = errno;
  close(fd);
  if (p == MAP_FAILED) return err_io(e, "mmap");
  ibu_header_t h;
  memcpy(&h, p, IBU_HEADER_SIZE);
  int32_t rc = ibu_header_validate(&h);
  if (!rc && (flen - IBU_HEADER_SIZE) % IBU_RECORD_SIZE != 0) rc = err_map_size();
  if (rc) {
    munmap(p, flen);
    return rc;
  }
  Mapping* mp = new Mapping;
  mp->base = static_cast<uint8_t*>(p);
  mp->len = flen;
  ibu_mmap* m = new ibu_mmap;
  m->map = mp;
  m->header = h;
  m->len = (flen - IBU_HEADER_SIZE) / IBU_RECORD_SIZE;
  *out = m;
  return IBU_OK;
}
extern "C" int32_t ibu_mmap_clone(ibu_mmap_t* m, ibu_mmap_t** out) {  // derive(Clone) mmap.rs:99
  if (!m || !out) return err_arg("NULL argument");
  m->map->refs.fetch_add(1, std::memory_order_relaxed);
  *out = new ibu_mmap(*m);
  return IBU_OK;
}
extern "C" size_t ibu_mmap_len(const ibu_mmap_t* m) { return m ? m->len : 0; }
extern "C" int32_t ibu_mmap_header(const ibu_mmap_t* m, ibu_header_t* out) {
  if (!m || !out) return err_arg("NULL argument");
  *out = m->header;
  return IBU_OK;
}
extern "C" int32_t ibu_mmap_slice(const ibu_mmap_t* m, size_t start, size_t end, const ibu_record_t** recs,
                                  size_t* n) {  // mmap.rs:253-270
  if (!m || !recs || !n) return err_arg("NULL argument");
  if (start >= m->len || end > m->len) return err_index(end, m->len);  // Q7: idx is always `end`
  if (end <= start) return err_index(end, m->len);
  *recs = reinterpret_cast<const ibu_record_t*>(m->map->base + IBU_HEADER_SIZE + start * IBU_RECORD_SIZE);
  *n = end - start;
  return IBU_OK;
}
extern "C" const void* ibu_mmap_base(const ibu_mmap_t* m) { return m ? m->map->base : nullptr; }
extern "C" void ibu_mmap_close(ibu_mmap_t* m) {
  if (!m) return;
  if (m->map->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) {
    munmap(m->map->base, m->map->len);
    delete m->map;
  }
  delete m;
}

extern "C" int32_t ibu_shard_range(size_t len, size_t n_shards, size_t shard, size_t* start, size_t* end) {
  if (!start || !end || n_shards == 0 || shard >= n_shards) return err_arg("shard out of range");
  const size_t per = len / n_shards, rem = len % n_shards;  // mmap.rs:297-298
  *start = shard * per;
  *end = shard == n_shards - 1 ? *start + per + rem : *start + per;  // :301-307
  return IBU_OK;
}

// num_cpus::get() (mmap.rs:292): CPUs this process may run on — the affinity mask, further limited by the
// cgroup CPU quota when one is set (num_cpus does the same: ceil(quota / period), cgroup v2 then v1).
static size_t cgroup_cpu_quota() {
  long long quota = -1, period = -1;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // v2: "<quota|max> <period>"
    char q[32] = {0};
    if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
    fclose(f);
  } else {
    if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
    if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = -1; fclose(g); }
  }
  if (quota > 0 && period > 0) return (size_t)((quota + period - 1) / period);
  return 0;
}
size_t ibu::host_cores() {
  size_t n = 0;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) {
    int c = CPU_COUNT(&set);
    if (c > 0) n = (size_t)c;
  }
  if (!n) {
    unsigned h = std::thread::hardware_concurrency();
    n = h ? h : 1;
  }
  const size_t q = cgroup_cpu_quota();
  if (q && q < n) n = q;
  return n;
}
size_t ibu::inflate_threads() {
  size_t n = 0;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) {
    int c = CPU_COUNT(&set);
    if (c > 0) n = (size_t)c;
  }
  if (!n) {
    unsigned h = std::thread::hardware_concurrency();
    n = h ? h : 1;
  }
  const size_t q = cgroup_cpu_quota();
  if (q && 2 * q < n) n = 2 * q;
  return n < 1 ? 1 : (n > 64 ? 64 : n);
}

extern "C" int32_t ibu_mmap_process_parallel(const ibu_mmap_t* m, const ibu_processor_vtable_t* vt, void* user,
                                             size_t num_threads) {  // mmap.rs:286-332
  if (!m || !vt || !vt->process_record) return err_arg("NULL reader / vtable / process_record");
  const size_t cores = host_cores();
  const size_t nt = num_threads == 0 ? cores : (num_threads < cores ? num_threads : cores);
  struct Worker {
    std::thread th;
    int32_t rc = IBU_OK;
    ibu_error_detail_t detail;
  };
  std::vector<Worker> ws;
  try { ws.resize(nt); } catch (...) { return caught_io("process_parallel: worker table"); }
  size_t spawned = 0;
  int32_t spawn_rc = IBU_OK;
  for (size_t i = 0; i < nt; ++i) {
    size_t start = 0, end = 0;
    ibu_shard_range(m->len, nt, i, &start, &end);
    void* clone = vt->clone ? vt->clone(user) : user;  // processor.clone()  :309
    Worker* w = &ws[i];
    try {
    w->th = std::thread([m, vt, clone, start, end, w]() {
      size_t batch_start = start;
      while (batch_start < end && w->rc == IBU_OK) {
        const size_t batch_end = batch_start + IBU_BATCH_SIZE < end ? batch_start + IBU_BATCH_SIZE : end;
        const ibu_record_t* recs;
        size_t n;
        w->rc = ibu_mmap_slice(m, batch_start, batch_end, &recs, &n);
        for (size_t k = 0; k < n && w->rc == IBU_OK; ++k) {
          int32_t u = vt->process_record(clone, &recs[k]);
          if (u) w->rc = err_process((uint64_t)(uint32_t)u);
        }
        if (w->rc == IBU_OK && vt->on_batch_complete) {
          int32_t u = vt->on_batch_complete(clone);
          if (u) w->rc = err_process((uint64_t)(uint32_t)u);
        }
        batch_start += IBU_BATCH_SIZE;
      }
      if (w->rc) w->detail = tls_error();  // the payload lives in the worker's thread-local slot
      if (vt->clone && vt->drop) vt->drop(clone);
    });
    } catch (...) {  // thread::spawn panics in the reference (mmap.rs:308); here: join what runs, report Io(EAGAIN)
      if (vt->clone && vt->drop) vt->drop(clone);
      spawn_rc = caught_io("process_parallel: cannot start a worker thread");
      break;
    }
    ++spawned;
  }
  // Join in spawn order, first Err wins (Q12).  The reference drops the remaining handles and
  // lets those threads run on detached; here they are joined so `user` may be freed on return.
  int32_t rc = IBU_OK;
  for (size_t i = 0; i < spawned; ++i) {
    ws[i].th.join();
    if (rc == IBU_OK && ws[i].rc) {
      rc = ws[i].rc;
      tls_error() = ws[i].detail;
    }
  }
  if (spawn_rc && rc == IBU_OK) return err_io(EAGAIN, "process_parallel: cannot start a worker thread");
  return rc;
}
