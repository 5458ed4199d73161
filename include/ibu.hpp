// ibu.hpp — the reference crate's public API (src/lib.rs:178-181), restated in C++17 ABOVE the C ABI of
// ibu_hip.h.  Header-only; links against libibu_hip.so.  Same names, argument meaning and error behaviour
// as the Rust items, so tests written against it read like the reference's own (tests/cpp/).
//
//   Rust                                   here
//   ------------------------------------   ---------------------------------------------------------------
//   Header / Record (repr(C) PODs)         ibu::Header / ibu::Record — the ABI structs themselves
//   Result<T, IbuError>                    T, or throws ibu::IbuError {kind(), expected/actual/pos/idx/max}
//   Writer<W: Write>                       ibu::Writer — sinks: std::ostream&, Vec<u8> (to_vec), path, stdout
//   Reader<R: Read> + Iterator             ibu::Reader — sources: std::istream&, byte slice, path, stdin;
//                                          next() -> std::optional<Record>; range-for
//   load_to_vec                            ibu::load_to_vec
//   MmapReader (Clone = Arc)               ibu::MmapReader (copy = clone of the shared map)
//   ParallelProcessor / ParallelReader     any copyable type with process_record(const Record&) and
//                                          optionally on_batch_complete(), set_tid(size_t), get_tid();
//                                          MmapReader::process_parallel(proc, num_threads)
//   — (new: device path)                   ibu::device::Context, DeviceBuffer; Writer::write_batch_device,
//                                          MmapReader::process_device_*, Reader::process_device_*
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <istream>
#include <optional>
#include <ostream>
#include <stdexcept>
#include <string>
#include <array>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "ibu_hip.h"

namespace ibu {

constexpr uint32_t MAGIC = IBU_MAGIC;          // header.rs:5
constexpr uint32_t VERSION = IBU_VERSION;      // header.rs:6
constexpr size_t HEADER_SIZE = IBU_HEADER_SIZE;  // header.rs:7
constexpr size_t RECORD_SIZE = IBU_RECORD_SIZE;  // record.rs:3
constexpr size_t DEFAULT_BUFFER_SIZE = IBU_DEFAULT_BUFFER_SIZE;  // reader.rs:14, writer.rs:10 (48 Ki records)

// ---- error.rs:56-128 ------------------------------------------------------------------------------------------
class IbuError : public std::runtime_error {
 public:
  enum Kind {
    Io = IBU_ERR_IO, Niffler = IBU_ERR_NIFFLER, InvalidMagicNumber = IBU_ERR_INVALID_MAGIC,
    TruncatedRecord = IBU_ERR_TRUNCATED_RECORD, InvalidVersion = IBU_ERR_INVALID_VERSION,
    InvalidBarcodeLength = IBU_ERR_INVALID_BC_LEN, InvalidUmiLength = IBU_ERR_INVALID_UMI_LEN,
    InvalidMapSize = IBU_ERR_INVALID_MAP_SIZE, InvalidIndex = IBU_ERR_INVALID_INDEX, Process = IBU_ERR_PROCESS,
    InvalidBase = IBU_ERR_INVALID_BASE, SeqLen = IBU_ERR_SEQ_LEN, InvalidArg = IBU_ERR_INVALID_ARG,
    Hip = IBU_ERR_HIP, NoDevice = IBU_ERR_NO_DEVICE
  };
  IbuError(int32_t code, const ibu_error_detail_t& d)
      : std::runtime_error(std::string(ibu_status_name(code)) + ": " + d.message), code_(code), d_(d) {}
  Kind kind() const { return static_cast<Kind>(code_); }
  const char* name() const { return ibu_status_name(code_); }
  uint64_t expected() const { return d_.a; }  // InvalidMagicNumber / InvalidVersion
  uint64_t actual() const { return d_.b; }
  uint64_t pos() const { return d_.a; }       // TruncatedRecord
  uint64_t idx() const { return d_.a; }       // InvalidIndex
  uint64_t max() const { return d_.b; }
  uint64_t length() const { return d_.a; }    // InvalidBarcodeLength / InvalidUmiLength
  uint64_t first_bad() const { return d_.a; } // InvalidBase
  uint64_t n_bad() const { return d_.b; }
  int os_errno() const { return d_.os_errno; }

 private:
  int32_t code_;
  ibu_error_detail_t d_;
};
inline void check(int32_t rc) {
  if (rc == IBU_OK) return;
  ibu_error_detail_t d;
  ibu_last_error(&d);
  throw IbuError(rc, d);
}

// ---- constructs/record.rs ------------------------------------------------------------------------------------
struct Record : ibu_record_t {
  Record() : ibu_record_t{0, 0, 0} {}                                        // Default
  Record(uint64_t barcode_, uint64_t umi_, uint64_t index_) : ibu_record_t{barcode_, umi_, index_} {}  // new :87-93
  const uint8_t* as_bytes() const { return reinterpret_cast<const uint8_t*>(this); }  // :108-110
  static Record from_bytes(const uint8_t* bytes, size_t len) {               // :130-132 (panics there, throws here)
    Record r;
    check(ibu_record_from_bytes(bytes, len, &r));
    return r;
  }
  int cmp(const Record& o) const { return ibu_record_cmp(this, &o); }        // derive(Ord) :58
  friend bool operator==(const Record& a, const Record& b) { return a.cmp(b) == 0; }
  friend bool operator!=(const Record& a, const Record& b) { return a.cmp(b) != 0; }
  friend bool operator<(const Record& a, const Record& b) { return a.cmp(b) < 0; }
  friend bool operator>(const Record& a, const Record& b) { return a.cmp(b) > 0; }
  friend bool operator<=(const Record& a, const Record& b) { return a.cmp(b) <= 0; }
  friend bool operator>=(const Record& a, const Record& b) { return a.cmp(b) >= 0; }
};
static_assert(sizeof(Record) == RECORD_SIZE, "Record is the 24-byte POD (record.rs:149-152)");

// ---- constructs/header.rs ------------------------------------------------------------------------------------
struct Header : ibu_header_t {
  Header() { std::memset(this, 0, sizeof *this); }
  Header(uint32_t bc_len_, uint32_t umi_len_) { ibu_header_init(this, bc_len_, umi_len_); }  // new :84-93
  void set_sorted() { ibu_header_set_sorted(this); }                         // :111-113
  bool sorted() const { return ibu_header_sorted(this) != 0; }               // :130-132
  void validate() const { check(ibu_header_validate(this)); }                // :167-187
  const uint8_t* as_bytes() const { return reinterpret_cast<const uint8_t*>(this); }  // :203-205
  static Header from_bytes(const uint8_t* bytes, size_t len) {               // :226-228
    Header h;
    check(ibu_header_from_bytes(bytes, len, &h));
    return h;
  }
  friend bool operator==(const Header& a, const Header& b) { return std::memcmp(&a, &b, sizeof a) == 0; }
  friend bool operator!=(const Header& a, const Header& b) { return !(a == b); }
};
static_assert(sizeof(Header) == HEADER_SIZE, "Header is the 32-byte POD (header.rs:248-251)");

struct RecordSlice {  // &[Record]
  const Record* ptr = nullptr;
  size_t len = 0;
  const Record* begin() const { return ptr; }
  const Record* end() const { return ptr + len; }
  size_t size() const { return len; }
  const Record& operator[](size_t i) const { return ptr[i]; }
};

namespace device {
class Context;
}
using StreamStats = ibu_stream_stats_t;
using RingConfig = ibu_ring_config_t;
using ReduceResult = ibu_reduce_result_t;
using DecodeSink = ibu_decode_sink_t;   // {d_bc_ascii, d_umi_ascii, d_index, cap_records}: device columns and the rows they hold
using AllocProbe = ibu_alloc_probe_t;
using NumaInfo = ibu_numa_info_t;
class DeviceStream;

// ---- io/writer.rs ----------------------------------------------------------------------------------------------
class Writer {
 public:
  Writer(std::ostream& inner, const Header& header) { open_stream(&inner, &header); }  // Writer::new :129-143
  static Writer new_headless(std::ostream& inner) { Writer w; w.open_stream(&inner, nullptr); return w; }  // :169-179
  static Writer to_vec(const Header& header) { Writer w; check(ibu_writer_open_mem(&header, &w.w_)); return w; }
  static Writer to_vec_headless() { Writer w; check(ibu_writer_open_mem(nullptr, &w.w_)); return w; }
  static Writer from_path(const std::string& path, const Header& header) {   // :556-559
    Writer w; check(ibu_writer_open_path(path.c_str(), &header, &w.w_)); return w;
  }
  static Writer from_stdout(const Header& header) { Writer w; check(ibu_writer_open_fd(1, &header, &w.w_)); return w; }  // :587-589
  static Writer from_optional_path(const std::optional<std::string>& path, const Header& header) {  // :617-626
    return path ? from_path(*path, header) : from_stdout(header);
  }
  Writer(Writer&& o) noexcept : w_(o.w_), os_(o.os_) { o.w_ = nullptr; rebind(); }
  Writer& operator=(Writer&& o) noexcept { close(); w_ = o.w_; os_ = o.os_; o.w_ = nullptr; rebind(); return *this; }
  Writer(const Writer&) = delete;
  Writer& operator=(const Writer&) = delete;
  ~Writer() { close(); }                                                     // Drop :519-523 (flush, errors swallowed)

  void write_record(const Record& r) { check(ibu_writer_write_record(w_, &r)); }              // :260-273
  void write_batch(const Record* recs, size_t n) { check(ibu_writer_write_batch(w_, recs, n)); }  // :315-318
  void write_batch(const std::vector<Record>& recs) { write_batch(recs.data(), recs.size()); }
  template <class It>
  void write_iter(It first, It last) { for (; first != last; ++first) write_record(*first); }  // :383-391
  void ingest(Writer& other) { check(ibu_writer_ingest(w_, other.w_)); }                       // :477-482
  void finish() { check(ibu_writer_finish(w_)); }                                             // :429-433
  uint64_t records_written() const { return ibu_writer_records_written(w_); }                 // :207-209
  // what the Vec<u8> sink holds right now (tests peek at `writer.inner`)
  std::vector<uint8_t> inner() const {
    const uint8_t* p = nullptr; size_t n = 0;
    check(ibu_writer_mem_view(w_, &p, &n));
    return std::vector<uint8_t>(p, p + n);
  }
  std::vector<uint8_t> into_inner() {                                        // :507-511 — NO flush
    uint8_t* p = nullptr; size_t n = 0;
    ibu_writer_t* w = w_; w_ = nullptr;
    check(ibu_writer_into_inner(w, &p, &n));
    std::vector<uint8_t> v(p, p + n);
    ibu_free(p);
    return v;
  }
  // device-resident AoS records -> pinned ring -> this writer, same buffered/direct rule (writer.rs:321-351)
  // producer_stream: the stream the records were produced on (nullptr: the context's own)
  inline StreamStats write_batch_device(device::Context& ctx, const void* d_records, size_t n, const RingConfig* ring = nullptr,
                                        void* producer_stream = nullptr);
  // n rows of host ASCII (+ optional index column; nullptr -> first_index + i) -> GPU 2-bit encode -> this writer
  // (README.md:38-47's encode-then-write loop as one batch call).  Throws InvalidBase{first_bad, n_bad}.
  inline StreamStats write_ascii_batch(device::Context& ctx, const uint8_t* bc_ascii, const uint8_t* umi_ascii, const uint64_t* index,
                                       size_t n, uint32_t bc_len, uint32_t umi_len, uint64_t first_index = 0,
                                       const RingConfig* ring = nullptr);
  ibu_writer_t* raw() const { return w_; }

 private:
  Writer() = default;
  void open_stream(std::ostream* os, const Header* h) {
    os_ = os;
    check(ibu_writer_open_callback(&Writer::wr, &Writer::fl, os_, h, &w_));
  }
  void rebind() {}  // the callback user pointer is the std::ostream itself, which does not move
  void close() { if (w_) { ibu_writer_close(w_); w_ = nullptr; } }
  static int32_t wr(void* u, const uint8_t* d, size_t n) {
    auto* os = static_cast<std::ostream*>(u);
    os->write(reinterpret_cast<const char*>(d), (std::streamsize)n);
    return os->good() ? 0 : 5 /*EIO*/;
  }
  static int32_t fl(void* u) {
    auto* os = static_cast<std::ostream*>(u);
    os->flush();
    return os->good() ? 0 : 5;
  }
  ibu_writer_t* w_ = nullptr;
  std::ostream* os_ = nullptr;
};

// ---- io/reader.rs ----------------------------------------------------------------------------------------------
class Reader {
 public:
  explicit Reader(std::istream& inner) { check(ibu_reader_open_callback(&Reader::rd, &inner, &r_)); }  // new :152-176
  Reader(const uint8_t* bytes, size_t len) { check(ibu_reader_open_mem(bytes, len, &r_)); }            // new(Cursor<&[u8]>)
  explicit Reader(const std::vector<uint8_t>& bytes) : Reader(bytes.data(), bytes.size()) {}
  static Reader from_path(const std::string& path) { Reader r; check(ibu_reader_open_path(path.c_str(), &r.r_)); return r; }  // :345-352
  static Reader from_stdin() { Reader r; check(ibu_reader_open_fd(0, &r.r_)); return r; }             // :389-396
  static Reader from_optional_path(const std::optional<std::string>& path) { return path ? from_path(*path) : from_stdin(); }  // :425-434
  Reader(Reader&& o) noexcept : r_(o.r_) { o.r_ = nullptr; }
  Reader& operator=(Reader&& o) noexcept { close(); r_ = o.r_; o.r_ = nullptr; return *this; }
  Reader(const Reader&) = delete;
  Reader& operator=(const Reader&) = delete;
  ~Reader() { close(); }

  Header header() const { Header h; check(ibu_reader_header(r_, &h)); return h; }                     // :244-246
  bool read_batch() { int32_t has = 0; check(ibu_reader_read_batch(r_, &has)); return has != 0; }     // :218-242
  std::optional<Record> next() {                                             // Iterator::next :279-306; Some(Err) throws
    Record r; int32_t got = 0;
    check(ibu_reader_next(r_, &r, &got));
    if (!got) return std::nullopt;
    return r;
  }
  uint64_t bytes_read() const { return ibu_reader_bytes_read(r_); }
  std::vector<Record> collect() { std::vector<Record> v; while (auto r = next()) v.push_back(*r); return v; }
  struct iterator {
    Reader* rd; std::optional<Record> cur;
    const Record& operator*() const { return *cur; }
    iterator& operator++() { cur = rd->next(); return *this; }
    bool operator!=(const iterator&) const { return cur.has_value(); }
  };
  iterator begin() { return iterator{this, next()}; }
  iterator end() { return iterator{this, std::nullopt}; }
  // device: stream the rest of this reader (plain or gzip) through the pinned ring
  inline std::pair<ReduceResult, StreamStats> process_device_reduce(device::Context& ctx, const RingConfig* ring = nullptr);
  // sink.cap_records: rows the columns hold; a longer stream throws IbuError (InvalidArg) instead of writing past them.
  // (The sink is ONE argument on purpose: revision 3 had squeezed `size_t cap_records` in front of other defaulted size_t
  // parameters, and a revision-2 positional call still compiled with shifted meanings.  Old calls no longer compile.)
  inline StreamStats process_device_decode(device::Context& ctx, const DecodeSink& sink, const RingConfig* ring = nullptr);
  // pull-style device stream over the rest of this reader: the caller takes one device-resident batch at a time (DeviceStream below)
  inline DeviceStream device_stream(device::Context& ctx, const RingConfig* ring = nullptr);
  ibu_reader_t* raw() const { return r_; }

 private:
  Reader() = default;
  void close() { if (r_) { ibu_reader_close(r_); r_ = nullptr; } }
  static int32_t rd(void* u, uint8_t* dst, size_t cap, size_t* got) {
    auto* is = static_cast<std::istream*>(u);
    is->read(reinterpret_cast<char*>(dst), (std::streamsize)cap);
    *got = (size_t)is->gcount();
    if (is->bad()) return 5;
    if (is->eof()) is->clear(is->rdstate() & ~std::ios::failbit);  // short read at EOF is not an error
    return 0;
  }
  ibu_reader_t* r_ = nullptr;
};

inline std::pair<Header, std::vector<Record>> load_to_vec(const std::string& path) {  // reader.rs:510-535
  Header h; ibu_record_t* p = nullptr; size_t n = 0;
  check(ibu_load_to_vec(path.c_str(), &h, &p, &n));
  std::vector<Record> v(n);
  if (n) std::memcpy(static_cast<void*>(v.data()), p, n * RECORD_SIZE);
  ibu_free(p);
  return {h, std::move(v)};
}

inline std::pair<size_t, size_t> shard_range(size_t len, size_t n_shards, size_t shard) {  // mmap.rs:297-307
  size_t s = 0, e = 0;
  check(ibu_shard_range(len, n_shards, shard, &s, &e));
  return {s, e};
}

// ---- parallel.rs + io/mmap.rs --------------------------------------------------------------------------------
// A processor error: throw ProcessError (or anything) from process_record / on_batch_complete; it reaches the
// caller of process_parallel as IbuError{Process} (first error in handle order wins — mmap.rs:326-328).
struct ProcessError : std::runtime_error {
  int32_t code;
  explicit ProcessError(const std::string& what, int32_t code_ = 1) : std::runtime_error(what), code(code_) {}
};
namespace detail {
template <class P, class = void> struct has_batch : std::false_type {};
template <class P> struct has_batch<P, std::void_t<decltype(std::declval<P&>().on_batch_complete())>> : std::true_type {};
template <class P, class = void> struct has_set_tid : std::false_type {};
template <class P> struct has_set_tid<P, std::void_t<decltype(std::declval<P&>().set_tid(size_t{}))>> : std::true_type {};
template <class F> int32_t guarded(F&& f) {
  try { f(); return 0; }
  catch (const ProcessError& e) { return e.code ? e.code : 1; }
  catch (...) { return 1; }
}
}  // namespace detail

class MmapReader {
 public:
  static constexpr size_t BATCH_SIZE = IBU_BATCH_SIZE;                       // mmap.rs:284
  explicit MmapReader(const std::string& path) { check(ibu_mmap_open(path.c_str(), &m_)); }  // new :143-161
  MmapReader(const MmapReader& o) { check(ibu_mmap_clone(o.m_, &m_)); }      // Clone: shares the map (Arc)
  MmapReader& operator=(const MmapReader& o) { if (this != &o) { close(); check(ibu_mmap_clone(o.m_, &m_)); } return *this; }
  MmapReader(MmapReader&& o) noexcept : m_(o.m_) { o.m_ = nullptr; }
  ~MmapReader() { close(); }
  size_t len() const { return ibu_mmap_len(m_); }                            // :178-180
  Header header() const { Header h; check(ibu_mmap_header(m_, &h)); return h; }  // :201-203
  RecordSlice slice(size_t start, size_t end) const {                        // :253-270
    const ibu_record_t* p = nullptr; size_t n = 0;
    check(ibu_mmap_slice(m_, start, end, &p, &n));
    return RecordSlice{static_cast<const Record*>(p), n};
  }
  const void* map_ptr() const { return ibu_mmap_base(m_); }                  // Arc::ptr_eq analogue (mmap.rs:541)

  // ParallelReader::process_parallel :286-332.  P: copyable (Clone), process_record(const Record&) [,
  // on_batch_complete(), set_tid(size_t)].  num_threads 0 = all cores.
  template <class P>
  void process_parallel(const P& processor, size_t num_threads) const {
    ibu_processor_vtable_t vt{};
    vt.clone = [](void* u) -> void* { return new P(*static_cast<const P*>(u)); };
    vt.drop = [](void* c) { delete static_cast<P*>(c); };
    vt.process_record = [](void* c, const ibu_record_t* r) -> int32_t {
      return detail::guarded([&] { static_cast<P*>(c)->process_record(*static_cast<const Record*>(r)); });
    };
    if constexpr (detail::has_batch<P>::value)
      vt.on_batch_complete = [](void* c) -> int32_t { return detail::guarded([&] { static_cast<P*>(c)->on_batch_complete(); }); };
    if constexpr (detail::has_set_tid<P>::value)
      vt.set_tid = [](void* c, size_t tid) { static_cast<P*>(c)->set_tid(tid); };
    check(ibu_mmap_process_parallel(m_, &vt, const_cast<P*>(&processor), num_threads));
  }
  // device: ONE shard of the same static split per GPU / rank
  inline std::pair<ReduceResult, StreamStats> process_device_reduce(device::Context& ctx, size_t shard = 0, size_t n_shards = 1,
                                                                     const RingConfig* ring = nullptr) const;
  inline StreamStats process_device_decode(device::Context& ctx, const DecodeSink& sink, size_t shard = 0, size_t n_shards = 1,
                                           const RingConfig* ring = nullptr) const;
  // process_parallel(processor, n) with a GPU per worker, ONE call (mmap.rs:286-332): worker i = a host thread + a context on
  // devices[i], shard i of the static split; an empty list = every visible device (num_threads == 0 = every core).  The
  // reduce form returns the total (wrapping sums, XORs) and the per-device partials; the decode form fills sinks[i] on devices[i].
  inline std::pair<ReduceResult, std::vector<ReduceResult>> process_devices_reduce(const std::vector<int32_t>& devices = {},
                                                                                   const RingConfig* ring = nullptr) const;
  inline void process_devices_decode(const std::vector<int32_t>& devices, std::vector<DecodeSink>& sinks, const RingConfig* ring = nullptr) const;
  // one shard -> ASCII barcodes / UMIs + index column in HOST memory, unpacked on the GPU
  struct Decoded { std::vector<uint8_t> bc, umi; std::vector<uint64_t> index; StreamStats stats; };
  inline Decoded decode_to_host(device::Context& ctx, size_t shard = 0, size_t n_shards = 1, const RingConfig* ring = nullptr) const;
  // pull-style device stream over one shard of the static split (DeviceStream below)
  inline DeviceStream device_stream(device::Context& ctx, size_t shard = 0, size_t n_shards = 1, const RingConfig* ring = nullptr) const;
  ibu_mmap_t* raw() const { return m_; }

 private:
  void close() { if (m_) { ibu_mmap_close(m_); m_ = nullptr; } }
  ibu_mmap_t* m_ = nullptr;
};

// ---- device path (no reference equivalent: the MI355X side of the boundary) --------------------------------
namespace device {
inline int device_count() { int32_t n = 0; return ibu_device_count(&n) == IBU_OK ? n : 0; }

class Context {
 public:
  explicit Context(int device = 0) { check(ibu_ctx_create(device, &c_)); }   // throws NoDevice without a gfx950 GPU
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  ~Context() { if (c_) ibu_ctx_destroy(c_); }
  ibu_ctx_t* raw() const { return c_; }
  void synchronize(void* stream = nullptr) { check(ibu_ctx_synchronize(c_, stream)); }
  void set_option(const char* key, int64_t v) { check(ibu_ctx_set_option(c_, key, v)); }

  void decode_ascii(const void* d_recs, size_t n, const Header& h, uint8_t* d_bc, uint8_t* d_umi, uint64_t* d_idx, void* st = nullptr) {
    check(ibu_decode_ascii(c_, d_recs, n, h.bc_len, h.umi_len, d_bc, d_umi, d_idx, st));
  }
  void encode_ascii(const uint8_t* d_bc, const uint8_t* d_umi, const uint64_t* d_idx, size_t n, const Header& h, void* d_recs,
                    uint64_t first_index = 0, void* st = nullptr) {
    check(ibu_encode_ascii(c_, d_bc, d_umi, d_idx, first_index, n, h.bc_len, h.umi_len, d_recs, st));
  }
  void codec_status(void* st = nullptr) { check(ibu_codec_status(c_, st, nullptr, nullptr)); }  // throws InvalidBase
  void deserialize(const void* d_recs, size_t n, uint64_t* bc, uint64_t* umi, uint64_t* idx, void* st = nullptr) {
    check(ibu_deserialize(c_, d_recs, n, bc, umi, idx, st));
  }
  void serialize(const uint64_t* bc, const uint64_t* umi, const uint64_t* idx, size_t n, void* d_recs, void* st = nullptr) {
    check(ibu_serialize(c_, bc, umi, idx, n, d_recs, st));
  }
  void unpack_2bit(const uint64_t* codes, size_t n, uint32_t len, uint8_t* ascii, void* st = nullptr) { check(ibu_unpack_2bit(c_, codes, n, len, ascii, st)); }
  void pack_2bit(const uint8_t* ascii, size_t n, uint32_t len, uint64_t* codes, void* st = nullptr) { check(ibu_pack_2bit(c_, ascii, n, len, codes, st)); }
  ReduceResult reduce(const void* d_recs, size_t n, void* st = nullptr) {
    ReduceResult r;
    check(ibu_reduce_reset(c_, st));
    check(ibu_reduce(c_, d_recs, n, st));
    check(ibu_reduce_fetch(c_, st, &r));
    return r;
  }
  void generate(uint64_t seed, uint64_t first, size_t n, const Header& h, void* d_recs, void* st = nullptr) {
    check(ibu_generate(c_, seed, first, n, h.bc_len, h.umi_len, d_recs, st));
  }
  void copy(void* d_dst, const void* d_src, size_t bytes, void* st = nullptr) { check(ibu_device_copy(c_, d_dst, d_src, bytes, st)); }
  void sort_records(void* d_recs, void* d_tmp, size_t n, void* st = nullptr) { check(ibu_sort_records(c_, d_recs, d_tmp, n, st)); }
  // the sort over several shards, one per context (= per GPU), in one call: shard i ends up with the i-th range of the global
  // order and shards[i].n says how long it is (ibu_sort_records_contexts)
  static void sort_records_contexts(const std::vector<Context*>& ctxs, std::vector<ibu_sort_shard_t>& shards) {
    if (shards.size() != ctxs.size()) {                        // the C side reads n_ctxs entries of both arrays
      ibu_error_detail_t d{};
      d.code = IBU_ERR_INVALID_ARG;
      snprintf(d.message, sizeof d.message, "Invalid argument: one shard per context");
      throw IbuError(IBU_ERR_INVALID_ARG, d);
    }
    std::vector<ibu_ctx_t*> raw;
    for (Context* c : ctxs) raw.push_back(c->c_);
    check(ibu_sort_records_contexts(raw.data(), raw.size(), shards.data()));
  }
  void lower_bound(const void* d_sorted, size_t n, const void* d_keys, size_t k, uint64_t* d_pos, void* st = nullptr) { check(ibu_lower_bound_records(c_, d_sorted, n, d_keys, k, d_pos, st)); }
  // index of the first record that differs, n if the slices are equal (Record: PartialEq, record.rs:58)
  size_t first_mismatch(const void* d_a, const void* d_b, size_t n, void* st = nullptr) { uint64_t f = 0; check(ibu_records_first_mismatch(c_, d_a, d_b, n, &f, st)); return (size_t)f; }
  bool is_sorted(const void* d_recs, size_t n, void* st = nullptr) { int32_t s = 0; check(ibu_is_sorted(c_, d_recs, n, st, &s)); return s != 0; }
  // compacted keys (the exchange format of the multi-GPU sort): OR / AND census, plan, records <-> 12-byte elements
  std::array<uint64_t, 8> census(const void* d_recs, size_t n, void* st = nullptr) { std::array<uint64_t, 8> c{}; check(ibu_records_census(c_, d_recs, n, c.data(), st)); return c; }
  static ibu_key_plan_t key_plan(const uint64_t or_words[3], const uint64_t and_words[3]) { ibu_key_plan_t p; check(ibu_key_plan_init(or_words, and_words, &p)); return p; }
  void compact(const ibu_key_plan_t& plan, const void* d_recs, size_t n, void* d_elems, void* st = nullptr) { check(ibu_records_compact(c_, &plan, d_recs, n, d_elems, st)); }
  void expand(const ibu_key_plan_t& plan, const void* d_elems, size_t n, void* d_recs, void* st = nullptr) { check(ibu_records_expand(c_, &plan, d_elems, n, d_recs, st)); }
  // BarcodeAnalyzer (parallel.rs:72-98) on sorted device records: (barcode, records, distinct UMIs), ascending barcode
  inline std::vector<std::tuple<uint64_t, uint64_t, uint64_t>> barcode_counts(const void* d_sorted, size_t n);
  // load_to_vec, device form -> (header, device pointer owned by the caller: release with free(), n)
  std::tuple<Header, void*, size_t> load_to_device(const std::string& path, const RingConfig* ring = nullptr, StreamStats* stats = nullptr) {
    Header h; void* p = nullptr; size_t n = 0;
    check(ibu_load_to_device(c_, path.c_str(), ring, &h, &p, 0, &n, stats));
    return {h, p, n};
  }
  // The same for a BGZF (bgzip) file of the records, inflated on the device: the compressed bytes cross the link.
  std::tuple<Header, void*, size_t> load_bgzf_to_device(const std::string& path, const RingConfig* ring = nullptr, StreamStats* stats = nullptr) {
    Header h; void* p = nullptr; size_t n = 0;
    check(ibu_load_bgzf_to_device(c_, path.c_str(), ring, &h, &p, 0, &n, stats));
    return {h, p, n};
  }
  // ... shard `shard` of `n_shards` of its records (the split of process_parallel) -> (header, device pointer, n, first record's number)
  std::tuple<Header, void*, size_t, uint64_t> load_bgzf_shard_to_device(const std::string& path, size_t shard, size_t n_shards,
                                                                        const RingConfig* ring = nullptr, StreamStats* stats = nullptr) {
    Header h; void* p = nullptr; size_t n = 0; uint64_t first = 0;
    check(ibu_load_bgzf_shard_to_device(c_, path.c_str(), ring, shard, n_shards, &h, &p, 0, &n, &first, stats));
    return {h, p, n, first};
  }
  void* alloc(size_t bytes) { void* p = nullptr; check(ibu_device_alloc(c_, bytes, &p)); return p; }
  // for arrays that stay resident: up to `tries` candidates, the one that streams fastest is kept (ibu_device_alloc_probed)
  void* alloc_probed(size_t bytes, uint32_t tries, AllocProbe* report = nullptr) {
    void* p = nullptr; check(ibu_device_alloc_probed(c_, bytes, tries, &p, report)); return p;
  }
  void free(void* p) { check(ibu_device_free(c_, p)); }
  // where the device hangs off the host and where the pinned ring landed (option "numa")
  NumaInfo numa() const { NumaInfo i; check(ibu_ctx_numa(c_, &i)); return i; }
  void upload(void* d_dst, const void* h_src, size_t bytes) { check(ibu_memcpy_h2d(c_, d_dst, h_src, bytes, nullptr)); synchronize(); }
  void download(void* h_dst, const void* d_src, size_t bytes) { check(ibu_memcpy_d2h(c_, h_dst, d_src, bytes, nullptr)); synchronize(); }

 private:
  ibu_ctx_t* c_ = nullptr;
};

class DeviceBuffer {  // RAII hipMalloc through the context
 public:
  DeviceBuffer(Context& ctx, size_t bytes) : ctx_(&ctx), bytes_(bytes), p_(ctx.alloc(bytes ? bytes : 16)) {}
  DeviceBuffer(DeviceBuffer&& o) noexcept : ctx_(o.ctx_), bytes_(o.bytes_), p_(o.p_) { o.p_ = nullptr; }
  DeviceBuffer(const DeviceBuffer&) = delete;
  ~DeviceBuffer() { if (p_) ibu_device_free(ctx_->raw(), p_); }
  void* ptr() const { return p_; }
  template <class T> T* as() const { return static_cast<T*>(p_); }
  size_t bytes() const { return bytes_; }
  template <class T> void upload(const std::vector<T>& v) { ctx_->upload(p_, v.data(), v.size() * sizeof(T)); }
  template <class T> std::vector<T> download(size_t count) const { std::vector<T> v(count); if (count) ctx_->download(static_cast<void*>(v.data()), p_, count * sizeof(T)); return v; }

 private:
  Context* ctx_;
  size_t bytes_;
  void* p_;
};

inline std::vector<std::tuple<uint64_t, uint64_t, uint64_t>> Context::barcode_counts(const void* d_sorted, size_t n) {
  size_t nb = 0, np = 0;
  check(ibu_barcode_counts(c_, d_sorted, n, nullptr, nullptr, nullptr, 0, &nb, &np, nullptr));
  std::vector<std::tuple<uint64_t, uint64_t, uint64_t>> out;
  if (!nb) return out;
  DeviceBuffer b(*this, 8 * nb), c(*this, 8 * nb), u(*this, 8 * nb);
  check(ibu_barcode_counts(c_, d_sorted, n, b.as<uint64_t>(), c.as<uint64_t>(), u.as<uint64_t>(), nb, &nb, &np, nullptr));
  auto hb = b.download<uint64_t>(nb), hc = c.download<uint64_t>(nb), hu = u.download<uint64_t>(nb);
  out.reserve(nb);
  for (size_t k = 0; k < nb; ++k) out.emplace_back(hb[k], hc[k], hu[k]);
  return out;
}
}  // namespace device

inline StreamStats Writer::write_batch_device(device::Context& ctx, const void* d_records, size_t n, const RingConfig* ring, void* producer_stream) {
  StreamStats st{};
  check(ibu_writer_write_batch_device_on(w_, ctx.raw(), ring, d_records, n, producer_stream, &st));
  return st;
}
inline StreamStats Writer::write_ascii_batch(device::Context& ctx, const uint8_t* bc_ascii, const uint8_t* umi_ascii, const uint64_t* index,
                                             size_t n, uint32_t bc_len, uint32_t umi_len, uint64_t first_index, const RingConfig* ring) {
  StreamStats st{};
  check(ibu_writer_write_ascii_batch(w_, ctx.raw(), ring, bc_ascii, umi_ascii, index, first_index, n, bc_len, umi_len, &st));
  return st;
}
inline MmapReader::Decoded MmapReader::decode_to_host(device::Context& ctx, size_t shard, size_t n_shards, const RingConfig* ring) const {
  const auto range = shard_range(len(), n_shards, shard);
  const size_t n = range.second - range.first;
  const Header h = header();
  Decoded d;
  d.bc.resize(n * h.bc_len); d.umi.resize(n * h.umi_len); d.index.resize(n);
  d.stats = StreamStats{};
  check(ibu_mmap_decode_to_host(m_, ctx.raw(), ring, shard, n_shards, d.bc.data(), d.umi.data(), d.index.data(), &d.stats));
  return d;
}
// ---- pull-style device record stream (ibu_stream_*) -------------------------------------------------------------------
// The device form of Reader::read_batch + Iterator (reader.rs:218-242, :279-306) and of process_parallel's per-batch loop
// (mmap.rs:312-320): `while (auto b = stream.next()) { ...kernels on b->d_records...; }` — a batch gives its ring slot back
// when it goes out of scope (work queued on its stream so far is waited for before the slot is refilled).
class DeviceBatch {
 public:
  const void* d_records = nullptr;   // n AoS records in a device ring slot
  size_t n = 0;
  uint64_t first_index = 0;          // number of the batch's first record (mmap: position in the map; reader: records delivered before)
  DeviceBatch(ibu_stream_t* s, const void* d, size_t n_, uint64_t first, void* stream) : d_records(d), n(n_), first_index(first), s_(s), stream_(stream) {}
  DeviceBatch(DeviceBatch&& o) noexcept : d_records(o.d_records), n(o.n), first_index(o.first_index), s_(o.s_), stream_(o.stream_) { o.s_ = nullptr; }
  DeviceBatch(const DeviceBatch&) = delete;
  DeviceBatch& operator=(const DeviceBatch&) = delete;
  ~DeviceBatch() { if (s_) (void)ibu_stream_release(s_, d_records, stream_); }
  void release() { if (s_) { ibu_stream_t* s = s_; s_ = nullptr; check(ibu_stream_release(s, d_records, stream_)); } }
 private:
  ibu_stream_t* s_;
  void* stream_;
};
class DeviceStream {
 public:
  explicit DeviceStream(ibu_stream_t* s) : s_(s) {}
  DeviceStream(DeviceStream&& o) noexcept : s_(o.s_) { o.s_ = nullptr; }
  DeviceStream(const DeviceStream&) = delete;
  DeviceStream& operator=(const DeviceStream&) = delete;
  ~DeviceStream() { if (s_) ibu_stream_close(s_); }
  Header header() const { Header h; check(ibu_stream_header(s_, &h)); return h; }
  // `stream`: the hipStream_t the batch will be read on (nullptr: the context's).  nullopt at the end; a source error
  // (TruncatedRecord, Io, Niffler ...) throws after the batches in front of it have been handed out.
  std::optional<DeviceBatch> next(void* stream = nullptr) {
    const void* d = nullptr; size_t n = 0; uint64_t first = 0;
    check(ibu_stream_next(s_, stream, &d, &n, &first));
    if (n == 0) return std::nullopt;
    return std::optional<DeviceBatch>(std::in_place, s_, d, n, first, stream);
  }
  StreamStats stats() const { StreamStats st{}; check(ibu_stream_stats(s_, &st)); return st; }
  ibu_stream_t* raw() const { return s_; }
 private:
  ibu_stream_t* s_;
};
inline DeviceStream Reader::device_stream(device::Context& ctx, const RingConfig* ring) {
  ibu_stream_t* s = nullptr;
  check(ibu_stream_open_reader(r_, ctx.raw(), ring, &s));
  return DeviceStream(s);
}
inline DeviceStream MmapReader::device_stream(device::Context& ctx, size_t shard, size_t n_shards, const RingConfig* ring) const {
  ibu_stream_t* s = nullptr;
  check(ibu_stream_open_mmap(m_, ctx.raw(), ring, shard, n_shards, &s));
  return DeviceStream(s);
}
inline std::pair<ReduceResult, StreamStats> Reader::process_device_reduce(device::Context& ctx, const RingConfig* ring) {
  ReduceResult r{}; StreamStats st{};
  check(ibu_reader_process_device(r_, ctx.raw(), ring, IBU_PROC_REDUCE, &r, &st));
  return {r, st};
}
inline StreamStats Reader::process_device_decode(device::Context& ctx, const DecodeSink& sink_in, const RingConfig* ring) {
  ibu_decode_sink_t sink = sink_in; StreamStats st{};
  check(ibu_reader_process_device(r_, ctx.raw(), ring, IBU_PROC_DECODE, &sink, &st));
  return st;
}
inline std::pair<ReduceResult, StreamStats> MmapReader::process_device_reduce(device::Context& ctx, size_t shard, size_t n_shards,
                                                                               const RingConfig* ring) const {
  ReduceResult r{}; StreamStats st{};
  check(ibu_mmap_process_device(m_, ctx.raw(), ring, IBU_PROC_REDUCE, shard, n_shards, &r, &st));
  return {r, st};
}
inline StreamStats MmapReader::process_device_decode(device::Context& ctx, const DecodeSink& sink_in, size_t shard, size_t n_shards,
                                                     const RingConfig* ring) const {
  ibu_decode_sink_t sink = sink_in; StreamStats st{};
  check(ibu_mmap_process_device(m_, ctx.raw(), ring, IBU_PROC_DECODE, shard, n_shards, &sink, &st));
  return st;
}
inline std::pair<ReduceResult, std::vector<ReduceResult>> MmapReader::process_devices_reduce(const std::vector<int32_t>& devices,
                                                                                             const RingConfig* ring) const {
  size_t n = devices.size();
  if (n == 0) n = (size_t)device::device_count();
  ReduceResult total{};
  std::vector<ReduceResult> parts(n ? n : 1);
  check(ibu_mmap_process_devices(m_, devices.empty() ? nullptr : devices.data(), devices.size(), ring, IBU_PROC_REDUCE, parts.data(), &total, nullptr));
  parts.resize(n);
  return {total, parts};
}
inline void MmapReader::process_devices_decode(const std::vector<int32_t>& devices, std::vector<DecodeSink>& sinks, const RingConfig* ring) const {
  if (devices.empty() || sinks.size() != devices.size()) {
    ibu_error_detail_t d{};
    d.code = IBU_ERR_INVALID_ARG;
    snprintf(d.message, sizeof d.message, "Invalid argument: one DecodeSink per listed device");
    throw IbuError(IBU_ERR_INVALID_ARG, d);
  }
  check(ibu_mmap_process_devices(m_, devices.data(), devices.size(), ring, IBU_PROC_DECODE, sinks.data(), nullptr, nullptr));
}

}  // namespace ibu
