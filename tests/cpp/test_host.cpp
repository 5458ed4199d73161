// The reference's unit tests, restated against the C++ mirror (include/ibu.hpp) of its public API.
// Each TEST names the reference test it restates (file:line under /root/reference/src).  CPU only.
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>

#include "check.hpp"
#include "ibu.hpp"

using namespace ibu;

static std::string tmp_path(const char* stem) {
  const char* d = std::getenv("TMPDIR");
  return std::string(d ? d : "/tmp") + "/ibu_cpp_" + stem + "_" + std::to_string(getpid()) + ".ibu";
}
struct TempFile {  // the reference writes fixed names into the CWD and deletes them (reader.rs:680, mmap.rs:377)
  std::string path;
  explicit TempFile(const char* stem) : path(tmp_path(stem)) {}
  ~TempFile() { unlink(path.c_str()); }
};
static std::vector<uint8_t> file_of(const Header& h, const std::vector<Record>& recs) {  // reader.rs:543-550 helper
  Writer w = Writer::to_vec(h);
  w.write_batch(recs);
  w.finish();
  return w.inner();
}
static void write_file(const std::string& path, const Header& h, const std::vector<Record>& recs) {
  Writer w = Writer::from_path(path, h);
  for (auto& r : recs) w.write_record(r);
  w.finish();
}
static std::vector<Record> ramp(size_t n, uint64_t a = 1, uint64_t b = 2, uint64_t c = 3) {
  std::vector<Record> v;
  for (size_t i = 0; i < n; ++i) v.emplace_back(i * a, i * b, i * c);
  return v;
}

// ---------------------------------------------------------------- constructs/header.rs
TEST(header_creation) {  // header.rs:236
  Header h(16, 12);
  CHECK_EQ(h.magic, MAGIC); CHECK_EQ(h.version, VERSION); CHECK_EQ(h.bc_len, 16u); CHECK_EQ(h.umi_len, 12u);
  CHECK_EQ(h.flags, 0ull); CHECK(!h.sorted());
  for (uint8_t b : h.reserved) CHECK_EQ(b, 0);
}
TEST(header_size) { CHECK_EQ(sizeof(Header), 32u); CHECK_EQ(HEADER_SIZE, 32u); }  // :248
TEST(header_sorted_flag) {  // :254
  Header h(16, 12);
  CHECK(!h.sorted());
  h.set_sorted(); CHECK(h.sorted()); CHECK_EQ(h.flags & 1, 1ull);
  h.set_sorted(); CHECK(h.sorted()); CHECK_EQ(h.flags, 1ull);  // idempotent
}
TEST(header_validation) {  // :273-348
  Header(16, 12).validate(); Header(1, 1).validate(); Header(32, 32).validate();
  Header m(16, 12); m.magic = 0x12345678;
  CHECK_THROWS(InvalidMagicNumber, m.validate(), { CHECK_EQ(e.expected(), (uint64_t)MAGIC); CHECK_EQ(e.actual(), 0x12345678ull); });
  Header v(16, 12); v.version = 1;
  CHECK_THROWS(InvalidVersion, v.validate(), { CHECK_EQ(e.expected(), 2ull); CHECK_EQ(e.actual(), 1ull); });
  CHECK_THROWS(InvalidBarcodeLength, Header(0, 12).validate(), CHECK_EQ(e.length(), 0ull));
  CHECK_THROWS(InvalidBarcodeLength, Header(33, 12).validate(), CHECK_EQ(e.length(), 33ull));
  CHECK_THROWS(InvalidUmiLength, Header(16, 0).validate(), CHECK_EQ(e.length(), 0ull));
  CHECK_THROWS(InvalidUmiLength, Header(16, 33).validate(), CHECK_EQ(e.length(), 33ull));
  Header both(0, 0); both.magic = 1;  // order: magic first (header.rs:168-186)
  CHECK_THROWS(InvalidMagicNumber, both.validate(), {});
}
TEST(header_bytes_roundtrip) {  // :351, :362
  Header h(16, 12);
  CHECK(Header::from_bytes(h.as_bytes(), HEADER_SIZE) == h);
  h.set_sorted();
  Header back = Header::from_bytes(h.as_bytes(), HEADER_SIZE);
  CHECK(back == h); CHECK(back.sorted());
  CHECK_THROWS(InvalidArg, Header::from_bytes(h.as_bytes(), 31), {});  // the reference panics on a wrong length
}
TEST(header_constants) {  // :374, :381
  const uint8_t want[4] = {0x49, 0x42, 0x55, 0x21};  // "IBU!"
  uint32_t m = MAGIC;
  CHECK(std::memcmp(&m, want, 4) == 0);
  CHECK_EQ(VERSION, 2u);
  Header a(16, 12), b = a;  // derives Clone, Copy, PartialEq (:386)
  CHECK(a == b); b.set_sorted(); CHECK(a != b);
}

// ---------------------------------------------------------------- constructs/record.rs
TEST(record_basics) {  // record.rs:140-161
  Record r(1, 2, 3);
  CHECK_EQ(r.barcode, 1ull); CHECK_EQ(r.umi, 2ull); CHECK_EQ(r.index, 3ull);
  CHECK_EQ(sizeof(Record), 24u); CHECK_EQ(RECORD_SIZE, 24u);
  Record d; CHECK_EQ(d.barcode + d.umi + d.index, 0ull);
}
TEST(record_ordering) {  // :164-232
  CHECK(Record(0, 0, 0) < Record(1, 0, 0)); CHECK(Record(0, 0, 0) < Record(0, 1, 0)); CHECK(Record(0, 0, 0) < Record(0, 0, 1));
  CHECK(Record(1, 0, 0) > Record(0, 9, 9));    // barcode dominates
  CHECK(Record(1, 1, 0) > Record(1, 0, 9));    // then umi
  CHECK(Record(1, 1, 0) > Record(0, 1, 1));    // :230
  std::vector<Record> v;
  for (int b : {1, 0}) for (int u : {1, 0}) for (int i : {1, 0}) v.emplace_back(b, u, i);
  std::sort(v.begin(), v.end());
  size_t k = 0;
  for (int b : {0, 1}) for (int u : {0, 1}) for (int i : {0, 1}) CHECK(v[k++] == Record(b, u, i));
}
TEST(record_bytes) {  // :235-264
  Record r(0x123456789ABCDEF0ull, 0xFEDCBA9876543210ull, ~0ull);
  const uint8_t want[24] = {0xf0, 0xde, 0xbc, 0x9a, 0x78, 0x56, 0x34, 0x12, 0x10, 0x32, 0x54, 0x76, 0x98, 0xba, 0xdc, 0xfe,
                            0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff};
  CHECK(std::memcmp(r.as_bytes(), want, 24) == 0);
  CHECK(Record::from_bytes(r.as_bytes(), 24) == r);
  Record z; CHECK(Record::from_bytes(z.as_bytes(), 24) == z);
  Record m(~0ull, ~0ull, ~0ull); CHECK(Record::from_bytes(m.as_bytes(), 24) == m);
  CHECK_THROWS(InvalidArg, Record::from_bytes(r.as_bytes(), 23), {});
}
TEST(record_equality_and_large_values) {  // :302-321
  CHECK(Record(1, 2, 3) == Record(1, 2, 3)); CHECK(Record(1, 2, 3) != Record(1, 2, 4));
  Record big(~0ull, ~0ull - 1, ~0ull - 2);
  CHECK_EQ(big.barcode, ~0ull); CHECK_EQ(big.umi, ~0ull - 1); CHECK_EQ(big.index, ~0ull - 2);
}

// ---------------------------------------------------------------- io/writer.rs
TEST(writer_creation_and_headless) {  // writer.rs:636-658
  Writer w = Writer::to_vec(Header(16, 12));
  CHECK_EQ(w.records_written(), 0ull); CHECK_EQ(w.inner().size(), 32u);  // header written immediately
  Writer hl = Writer::to_vec_headless();
  CHECK_EQ(hl.inner().size(), 0u);
}
TEST(writer_single_and_batch) {  // :661-694
  Writer w = Writer::to_vec(Header(16, 12));
  w.write_record(Record(1, 2, 3)); w.finish();
  CHECK_EQ(w.records_written(), 1ull); CHECK_EQ(w.inner().size(), 32u + 24u);
  w.write_batch({Record(4, 5, 6), Record(7, 8, 9), Record(10, 11, 12)}); w.finish();
  CHECK_EQ(w.records_written(), 4ull); CHECK_EQ(w.inner().size(), 32u + 4 * 24u);
  Writer hl = Writer::to_vec_headless();
  hl.write_record(Record(1, 2, 3)); hl.finish();
  CHECK_EQ(hl.inner().size(), 24u);
}
TEST(writer_readme_file) {  // README.md:66-85 / lib.rs:38-71: 80 bytes
  Header h(16, 12); h.set_sorted();
  Writer w = Writer::to_vec(h);
  w.write_record(Record(0x1100, 0x100011, 0)); w.write_record(Record(0x1101, 0x100010, 1)); w.finish();
  auto b = w.inner();
  CHECK_EQ(b.size(), 80u); CHECK_EQ(b[16], 1);
  const uint8_t rec0[8] = {0x00, 0x11, 0, 0, 0, 0, 0, 0};
  CHECK(std::memcmp(b.data() + 32, rec0, 8) == 0);
}
TEST(writer_iter_and_large_batch) {  // :697-719
  Writer w = Writer::to_vec(Header(16, 12));
  auto recs = ramp(10);
  w.write_iter(recs.begin(), recs.end()); w.finish();
  CHECK_EQ(w.records_written(), 10ull);
  Writer big = Writer::to_vec(Header(16, 12));
  big.write_batch(ramp(100000));  // > 1 179 648 B: direct path (writer.rs:325-331)
  CHECK_EQ(big.records_written(), 100000ull);
  CHECK_EQ(big.inner().size(), 32u + 100000u * 24u);  // written through without finish()
}
TEST(writer_ingest) {  // :722-741
  Writer main_w = Writer::to_vec(Header(16, 12));
  Writer aux = Writer::to_vec_headless();
  aux.write_record(Record(1, 2, 3)); aux.write_record(Record(4, 5, 6));
  main_w.ingest(aux);
  CHECK_EQ(main_w.records_written(), 2ull);
  CHECK_EQ(aux.inner().size(), 0u);  // aux sink cleared
  main_w.finish();
  CHECK_EQ(main_w.inner().size(), 32u + 48u);
}
TEST(writer_roundtrip_and_flush_boundary) {  // :744-787
  auto recs = ramp(1000, 1, 2, 3);
  auto bytes = file_of(Header(16, 12), recs);
  Reader r(bytes);
  CHECK(r.collect() == recs);
  Writer w = Writer::to_vec(Header(16, 12));
  for (size_t i = 0; i < 49152; ++i) w.write_record(Record(i, i, i));
  CHECK_EQ(w.inner().size(), 32u);                       // exactly one buffer: nothing flushed yet
  w.write_record(Record(0, 0, 0));
  CHECK_EQ(w.inner().size(), 32u + 1179648u);            // the 49 153rd flushes the full buffer first
}
TEST(writer_counter_drop_empty_mixed) {  // :790-865
  Writer w = Writer::to_vec(Header(16, 12));
  w.write_record(Record(1, 1, 1)); CHECK_EQ(w.records_written(), 1ull);
  w.write_batch(ramp(5)); CHECK_EQ(w.records_written(), 6ull);
  w.write_batch(nullptr, 0); CHECK_EQ(w.records_written(), 6ull);  // empty batch is fine (:823)
  TempFile f("drop");
  { Writer d = Writer::from_path(f.path, Header(16, 12)); d.write_record(Record(9, 9, 9)); }  // Drop flushes (:810)
  struct stat st; CHECK(stat(f.path.c_str(), &st) == 0); CHECK_EQ((size_t)st.st_size, 56u);
  Writer m = Writer::to_vec(Header(16, 12));
  m.write_record(Record(1, 2, 3));
  m.write_batch({Record(4, 5, 6), Record(7, 8, 9)});
  std::vector<Record> it = {Record(10, 11, 12), Record(11, 22, 33), Record(12, 24, 36)};
  m.write_iter(it.begin(), it.end());
  m.finish();
  auto bytes = m.inner();
  Reader r(bytes);
  auto got = r.collect();
  CHECK_EQ(got.size(), 6u); CHECK(got[5] == Record(12, 24, 36));
}
TEST(writer_into_inner_does_not_flush) {  // :507-511 ManuallyDrop
  Writer w = Writer::to_vec(Header(16, 12));
  w.write_record(Record(1, 2, 3));
  CHECK_EQ(w.into_inner().size(), 32u);
}
TEST(writer_does_not_validate_header) {  // quirk Q2 (writer.rs:129-143)
  Writer w = Writer::to_vec(Header(0, 99));
  auto b = w.inner();
  CHECK_EQ(b.size(), 32u);
  CHECK_THROWS(InvalidBarcodeLength, Reader r(b), {});
}
TEST(writer_generic_ostream_sink) {  // Writer<W: Write>
  std::ostringstream os;
  {
    Writer w(os, Header(8, 8));
    w.write_batch(ramp(3));
    w.finish();
    CHECK_EQ(os.str().size(), 32u + 72u);
  }
  std::istringstream is(os.str());
  Reader r(is);
  CHECK(r.collect() == ramp(3));
}

// ---------------------------------------------------------------- io/reader.rs
TEST(reader_creation_and_invalid_header) {  // reader.rs:553-575
  auto bytes = file_of(Header(16, 12), {Record(1, 2, 3)});
  Reader r(bytes);
  CHECK(r.header() == Header(16, 12));
  std::vector<uint8_t> zeros(32, 0);
  CHECK_THROWS(InvalidMagicNumber, Reader bad(zeros), {});
  std::vector<uint8_t> shorty(10, 0);
  CHECK_THROWS(Io, Reader bad(shorty), {});  // UnexpectedEof is an io::Error
}
TEST(reader_iterator_empty_large) {  // :578-616
  std::vector<Record> three = {Record(1, 2, 3), Record(4, 5, 6), Record(7, 8, 9)};
  auto b3 = file_of(Header(16, 12), three);
  Reader r(b3);
  size_t k = 0;
  for (const Record& rec : r) CHECK(rec == three[k++]);
  CHECK_EQ(k, 3u);
  auto be = file_of(Header(16, 12), {});
  Reader e(be); CHECK(!e.next().has_value());
  auto big = ramp(100000, 1, 2, 3);
  auto bb = file_of(Header(16, 12), big);
  Reader rb(bb);
  CHECK(rb.collect() == big);  // refills of 49152 / 49152 / 1696
}
TEST(reader_truncated_and_manual_batches) {  // :619-653
  auto b = file_of(Header(16, 12), {Record(1, 2, 3)});
  b.resize(b.size() - 5);
  Reader r(b);
  CHECK_THROWS(TruncatedRecord, r.next(), CHECK_EQ(e.pos(), 32ull));
  auto one = file_of(Header(16, 12), {Record(1, 2, 3)});
  Reader m(one);
  CHECK(m.read_batch()); CHECK(!m.read_batch());
}
TEST(reader_bytes_read_tracking) {  // :744-766
  auto b = file_of(Header(16, 12), ramp(10));
  Reader r(b);
  CHECK_EQ(r.bytes_read(), 32ull);
  r.collect();
  CHECK_EQ(r.bytes_read(), 32ull + 240ull);
}
TEST(load_to_vec_cases) {  // :669-741
  TempFile f("ltv");
  auto recs = ramp(1000, 1, 2, 3);
  write_file(f.path, Header(16, 12), recs);
  auto [h, got] = load_to_vec(f.path);
  CHECK(h == Header(16, 12)); CHECK(got == recs);
  write_file(f.path, Header(16, 12), {});
  CHECK_EQ(load_to_vec(f.path).second.size(), 0u);
  write_file(f.path, Header(16, 12), {Record(1, 2, 3)});
  CHECK(truncate(f.path.c_str(), 32 + 24 - 5) == 0);
  CHECK_THROWS(InvalidMapSize, load_to_vec(f.path), {});
  CHECK_THROWS(Io, load_to_vec(f.path + ".missing"), CHECK(e.os_errno() != 0));
}

// ---------------------------------------------------------------- io/mmap.rs + parallel.rs
struct SumProcessor {  // the reference's TestProcessor (parallel.rs:311-352, mmap.rs:340-373): Arc'd globals, local state
  std::shared_ptr<std::atomic<uint64_t>> global_count = std::make_shared<std::atomic<uint64_t>>(0);
  std::shared_ptr<std::atomic<uint64_t>> global_sum = std::make_shared<std::atomic<uint64_t>>(0);
  std::shared_ptr<std::atomic<uint64_t>> batches = std::make_shared<std::atomic<uint64_t>>(0);
  uint64_t local_count = 0, local_sum = 0;
  std::optional<size_t> tid;
  void process_record(const Record& r) { local_count += 1; local_sum += r.barcode + r.umi + r.index; }
  void on_batch_complete() {
    global_count->fetch_add(local_count); global_sum->fetch_add(local_sum); batches->fetch_add(1);
    local_count = local_sum = 0;
  }
  void set_tid(size_t t) { tid = t; }
  std::optional<size_t> get_tid() const { return tid; }
};
TEST(processor_basic_functionality) {  // parallel.rs:355-381, :461-483
  SumProcessor p;
  SumProcessor c = p;  // clone: independent local state, shared globals
  CHECK(!c.get_tid().has_value()); c.set_tid(42); CHECK_EQ(*c.get_tid(), 42u);
  c.process_record(Record(1, 2, 3)); c.process_record(Record(4, 5, 6));
  CHECK_EQ(c.local_count, 2ull); CHECK_EQ(c.local_sum, 21ull);
  c.on_batch_complete();
  CHECK_EQ(c.local_count, 0ull); CHECK_EQ(p.global_count->load(), 2ull); CHECK_EQ(p.global_sum->load(), 21ull);
  SumProcessor c2 = p; c2.set_tid(2);
  CHECK_EQ(*c.get_tid(), 42u); CHECK_EQ(*c2.get_tid(), 2u);
  CHECK(c.global_count == c2.global_count);  // Arc::ptr_eq
}
TEST(mmap_creation_and_slices) {  // mmap.rs:376-452
  TempFile f("mmap");
  write_file(f.path, Header(16, 12), ramp(100));
  MmapReader m(f.path);
  CHECK_EQ(m.len(), 100u); CHECK(m.header() == Header(16, 12));
  auto s = m.slice(10, 20);
  CHECK_EQ(s.size(), 10u); CHECK(s[0] == Record(10, 20, 30)); CHECK(s[9] == Record(19, 38, 57));
  CHECK_EQ(m.slice(50, 51).size(), 1u);
  write_file(f.path, Header(16, 12), {Record(1, 2, 3)});
  MmapReader one(f.path);
  CHECK_THROWS(InvalidIndex, one.slice(0, 2), { CHECK_EQ(e.idx(), 2ull); CHECK_EQ(e.max(), 1ull); });
  CHECK_THROWS(InvalidIndex, one.slice(1, 1), { CHECK_EQ(e.idx(), 1ull); CHECK_EQ(e.max(), 1ull); });
  CHECK_THROWS(InvalidIndex, one.slice(1, 0), { CHECK_EQ(e.idx(), 0ull); CHECK_EQ(e.max(), 1ull); });  // Q7: reports `end`
}
TEST(mmap_parallel_processing) {  // :455-519
  TempFile f("par");
  write_file(f.path, Header(16, 12), ramp(10000));
  MmapReader m(f.path);
  SumProcessor p;
  m.process_parallel(p, 4);
  CHECK_EQ(p.global_count->load(), 10000ull);
  CHECK_EQ(p.global_sum->load(), 299970000ull);  // sum(i + 2i + 3i)
  write_file(f.path, Header(16, 12), ramp(1000));
  MmapReader a(f.path);
  SumProcessor pa;
  a.process_parallel(pa, 0);  // 0 = all cores
  CHECK_EQ(pa.global_count->load(), 1000ull);
  write_file(f.path, Header(16, 12), {});
  MmapReader e(f.path);
  CHECK_EQ(e.len(), 0u);
  SumProcessor pe;
  e.process_parallel(pe, 2);
  CHECK_EQ(pe.global_count->load(), 0ull); CHECK_EQ(pe.batches->load(), 0ull);  // Q6: no batch for empty ranges
}
TEST(mmap_clone_large_and_constants) {  // :522-573
  TempFile f("clone");
  std::vector<Record> recs;
  for (uint64_t i = 0; i < 100000; ++i) recs.emplace_back(i % 1000, i % 500, i);
  write_file(f.path, Header(16, 12), recs);
  MmapReader m(f.path);
  MmapReader c = m;
  CHECK_EQ(c.len(), m.len()); CHECK(c.map_ptr() == m.map_ptr());  // Arc::ptr_eq
  auto s = m.slice(50000, 50010);
  CHECK_EQ(s.size(), 10u); CHECK_EQ(s[0].index, 50000ull);
  CHECK_EQ(MmapReader::BATCH_SIZE, 1024u * 1024u);
  SumProcessor p;
  m.process_parallel(p, 3);
  CHECK_EQ(p.global_count->load(), 100000ull);
}
struct FailAt {  // parallel.rs:414-436 ErrorProcessor / error.rs:344-369
  uint64_t fail_on;
  void process_record(const Record& r) { if (r.index == fail_on) throw ProcessError("Test error", 7); }
};
struct Minimal { void process_record(const Record&) {} };  // parallel.rs:439-458: defaults suffice
TEST(processor_errors_and_defaults) {
  TempFile f("err");
  write_file(f.path, Header(16, 12), ramp(100, 1, 1, 1));
  MmapReader m(f.path);
  CHECK_THROWS(Process, m.process_parallel(FailAt{5}, 2), {});
  m.process_parallel(FailAt{1000}, 2);  // never fails
  m.process_parallel(Minimal{}, 0);
}
TEST(shard_range_is_the_static_split) {  // mmap.rs:297-307 incl. quirk Q5
  CHECK(shard_range(10, 3, 0) == std::make_pair((size_t)0, (size_t)3));
  CHECK(shard_range(10, 3, 2) == std::make_pair((size_t)6, (size_t)10));  // remainder to the LAST
  CHECK(shard_range(2, 4, 0) == std::make_pair((size_t)0, (size_t)0));    // len < n: all but the last are empty
  CHECK(shard_range(2, 4, 3) == std::make_pair((size_t)0, (size_t)2));
  CHECK_THROWS(InvalidArg, shard_range(10, 0, 0), {});
}

// ---------------------------------------------------------------- error.rs
TEST(error_display_messages) {  // error.rs:196-248: the wording the reference asserts
  Header m(16, 12); m.magic = 0x12345678;
  try { m.validate(); CHECK(false); } catch (const IbuError& e) {
    std::string s = e.what();
    CHECK(s.find("0x21554249") != std::string::npos); CHECK(s.find("0x12345678") != std::string::npos);
  }
  Header v(16, 12); v.version = 1;
  try { v.validate(); CHECK(false); } catch (const IbuError& e) {
    std::string s = e.what();
    CHECK(s.find("expected (2)") != std::string::npos); CHECK(s.find("found (1)") != std::string::npos);
  }
  try { Header(33, 12).validate(); CHECK(false); } catch (const IbuError& e) {
    std::string s = e.what();
    CHECK(s.find("33") != std::string::npos); CHECK(s.find("1-32") != std::string::npos);
  }
  TempFile f("msg");
  write_file(f.path, Header(16, 12), {Record(1, 2, 3)});
  CHECK(truncate(f.path.c_str(), 32 + 19) == 0);
  try { load_to_vec(f.path); CHECK(false); } catch (const IbuError& e) { CHECK(std::string(e.what()).find("not a multiple") != std::string::npos); }
  MmapReader* none = nullptr; (void)none;
  write_file(f.path, Header(16, 12), ramp(50));
  MmapReader mm(f.path);
  try { mm.slice(0, 100); CHECK(false); } catch (const IbuError& e) {
    std::string s = e.what();
    CHECK(s.find("100") != std::string::npos); CHECK(s.find("50") != std::string::npos);
  }
  try { mm.process_parallel(FailAt{3}, 1); CHECK(false); } catch (const IbuError& e) { CHECK(std::string(e.what()).find("Processing error") != std::string::npos); }
}

// ---------------------------------------------------------------- niffler's formats (files prepared by tests/test_cpp.py)
TEST(compressed_inputs_decode_to_the_plain_records) {  // reader.rs:345-352
  const char* dir = std::getenv("IBU_TEST_COMPRESSED_DIR");
  if (!dir) return;  // only run with the fixture directory
  auto want = load_to_vec(std::string(dir) + "/plain.ibu").second;
  size_t seen = 0;
  for (const char* name : {"plain.ibu", "a.gz", "multi.gz", "holes.gz", "a.bgz", "a.bz2", "a.xz", "a.zst"}) {
    Reader r = Reader::from_path(std::string(dir) + "/" + name);
    CHECK(r.header() == Header(16, 12));
    CHECK(r.collect() == want);
    ++seen;
  }
  CHECK_EQ(seen, 8u);
  CHECK_THROWS(Niffler, Reader::from_path(std::string(dir) + "/cut.bgz").collect(), {});
}

// ---------------------------------------------------------------- device path without a device
TEST(device_context_fails_loudly_without_gpu) {
  if (device::device_count() > 0) return;  // on a GPU box the device tests cover it
  CHECK_THROWS(NoDevice, device::Context ctx(0), {});
}

int main(int argc, char** argv) { return run_all(argc, argv); }
