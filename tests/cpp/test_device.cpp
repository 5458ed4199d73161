// Device path through the C++ mirror (include/ibu.hpp), checked against the CPU oracle (oracle/ibu_oracle.h —
// test infrastructure, linked here as the checker only).  Needs an MI355X: run by tests/test_cpp.py under -m gpu.
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <memory>
#include <vector>

#include "check.hpp"
#include "ibu.hpp"
extern "C" {
#include "ibu_oracle.h"
}

using namespace ibu;
using device::Context;
using device::DeviceBuffer;

static Context& ctx() { static Context c(0); return c; }
static std::vector<Record> oracle_records(uint64_t seed, uint64_t first, size_t n, uint32_t bc, uint32_t umi) {
  std::vector<Record> v(n);
  orc_generate(seed, first, n, bc, umi, reinterpret_cast<orc_record*>(v.data()));
  return v;
}
static std::string tmp_path(const char* stem) {
  const char* d = std::getenv("TMPDIR");
  return std::string(d ? d : "/tmp") + "/ibu_cppdev_" + stem + "_" + std::to_string(getpid()) + ".ibu";
}

TEST(generate_decode_encode_reduce_match_the_oracle) {
  for (auto lens : {std::pair<uint32_t, uint32_t>{16, 12}, {32, 32}, {15, 11}, {1, 1}}) {
    for (size_t n : {size_t(0), size_t(1), size_t(127), size_t(129), size_t(100003)}) {
      const Header h(lens.first, lens.second);
      auto want = oracle_records(0x1B00001, 5, n, h.bc_len, h.umi_len);
      DeviceBuffer recs(ctx(), n * 24), back(ctx(), n * 24), bc(ctx(), n * h.bc_len), umi(ctx(), n * h.umi_len), idx(ctx(), n * 8);
      ctx().generate(0x1B00001, 5, n, h, recs.ptr());
      CHECK(recs.download<Record>(n) == want);
      ctx().decode_ascii(recs.ptr(), n, h, bc.as<uint8_t>(), umi.as<uint8_t>(), idx.as<uint64_t>());
      std::vector<uint8_t> wbc(n * h.bc_len), wumi(n * h.umi_len);
      std::vector<uint64_t> widx(n);
      orc_decode_records(reinterpret_cast<const orc_record*>(want.data()), n, h.bc_len, h.umi_len, wbc.data(), wumi.data(), widx.data());
      CHECK(bc.download<uint8_t>(wbc.size()) == wbc);
      CHECK(umi.download<uint8_t>(wumi.size()) == wumi);
      CHECK(idx.download<uint64_t>(n) == widx);
      ctx().encode_ascii(bc.as<uint8_t>(), umi.as<uint8_t>(), idx.as<uint64_t>(), n, h, back.ptr());
      ctx().codec_status();
      CHECK(back.download<Record>(n) == want);
      CHECK_EQ(ctx().first_mismatch(recs.ptr(), back.ptr(), n), n);   // `recs == back` on the device (Record: PartialEq)
      if (n > 2) {                                                     // ... and a difference is found where it is
        auto other = want;
        other[n / 2].umi ^= 1;
        other[n - 1].index += 1;
        back.upload(other);
        CHECK_EQ(ctx().first_mismatch(recs.ptr(), back.ptr(), n), n / 2);
      }
      orc_reduce r;
      orc_reduce_records(reinterpret_cast<const orc_record*>(want.data()), n, &r);
      auto got = ctx().reduce(recs.ptr(), n);
      CHECK_EQ(got.count, r.count);
      for (int k = 0; k < 3; ++k) { CHECK_EQ(got.sum[k], r.sum[k]); CHECK_EQ(got.xor_[k], r.xor_[k]); }
    }
  }
}
TEST(invalid_base_is_reported_with_first_record_and_count) {
  const Header h(16, 12);
  const size_t n = 1000;
  std::vector<uint8_t> bc(n * 16, 'A'), umi(n * 12, 'c');
  bc[16 * 7 + 3] = 'N'; umi[12 * 500] = 0; bc[16 * 999] = 'x';
  DeviceBuffer dbc(ctx(), bc.size()), dumi(ctx(), umi.size()), out(ctx(), n * 24);
  dbc.upload(bc); dumi.upload(umi);
  ctx().encode_ascii(dbc.as<uint8_t>(), dumi.as<uint8_t>(), nullptr, n, h, out.ptr(), 100);
  CHECK_THROWS(InvalidBase, ctx().codec_status(), { CHECK_EQ(e.first_bad(), 7ull); CHECK_EQ(e.n_bad(), 3ull); });
  ctx().codec_status();  // re-armed
  auto recs = out.download<Record>(n);
  CHECK(recs[0] == Record(0, 0x555555, 100));  // 'A' = 00, 'c' = 01 in every 2-bit field; index = first_index + i
  CHECK_EQ(recs[7].barcode, 0ull);             // the offending field is written as 0
}
TEST(sort_matches_oracle_and_aggregation_follows) {
  for (size_t n : {size_t(2), size_t(1025), size_t(200003)}) {
    auto recs = oracle_records(0x1B00005, 0, n, 6, 5);
    std::reverse(recs.begin(), recs.end());
    DeviceBuffer d(ctx(), n * 24), t(ctx(), n * 24);
    d.upload(recs);
    CHECK(n <= 2 || !ctx().is_sorted(d.ptr(), n));
    ctx().sort_records(d.ptr(), t.ptr(), n);
    orc_sort_records(reinterpret_cast<orc_record*>(recs.data()), n);
    CHECK(d.download<Record>(n) == recs);
    CHECK(ctx().is_sorted(d.ptr(), n));
    std::vector<uint64_t> b(n), c(n), u(n);
    size_t k = orc_barcode_counts(reinterpret_cast<const orc_record*>(recs.data()), n, b.data(), c.data(), u.data());
    auto got = ctx().barcode_counts(d.ptr(), n);
    CHECK_EQ(got.size(), k);
    for (size_t i = 0; i < k; ++i) CHECK(got[i] == std::make_tuple(b[i], c[i], u[i]));
  }
}
TEST(sort_over_several_contexts_is_one_order) {
  // ibu_sort_records_contexts: three shards on three contexts (all on this box's GPU) come back as the ranges of one order
  const size_t counts[3] = {70001, 0, 130007}, total = 200008, cap = total;
  auto recs = oracle_records(0x1B00009, 0, total, 16, 12);
  std::reverse(recs.begin(), recs.end());
  device::Context c1(0), c2(0);
  std::vector<device::Context*> cs = {&ctx(), &c1, &c2};
  std::vector<std::unique_ptr<DeviceBuffer>> bufs;
  std::vector<ibu_sort_shard_t> shards;
  size_t at = 0;
  for (int i = 0; i < 3; ++i) {
    bufs.emplace_back(new DeviceBuffer(*cs[i], cap * 24));
    bufs.emplace_back(new DeviceBuffer(*cs[i], cap * 24));
    if (counts[i]) bufs[2 * i]->upload(std::vector<Record>(recs.begin() + at, recs.begin() + at + counts[i]));
    shards.push_back({bufs[2 * i]->ptr(), bufs[2 * i + 1]->ptr(), counts[i], cap});
    at += counts[i];
  }
  device::Context::sort_records_contexts(cs, shards);
  orc_sort_records(reinterpret_cast<orc_record*>(recs.data()), total);
  std::vector<Record> got;
  for (int i = 0; i < 3; ++i) {
    auto part = bufs[2 * i]->download<Record>(shards[i].n);
    got.insert(got.end(), part.begin(), part.end());
  }
  CHECK_EQ(got.size(), total);
  CHECK(got == recs);
  // records that cannot be split (all the same) in shards none of which holds them all: refused with the numbers.  (A shard that is
  // merely small is no longer a reason: since round 4 the cut between two owners respects their capacities where it can.)
  std::vector<Record> same(1000, recs[0]);
  for (int i = 0; i < 3; ++i) {
    bufs[2 * i]->upload(same);
    shards[i].n = 1000;
    shards[i].capacity = 1500;
  }
  CHECK_THROWS(InvalidArg, device::Context::sort_records_contexts(cs, shards), {});
}
TEST(compacted_keys_round_trip) {
  // census -> plan -> 12-byte elements -> records: the exchange format of the multi-GPU sort (ibu_records_compact / _expand)
  const size_t n = 100003;
  auto recs = oracle_records(0x1B00005, 0, n, 16, 12);
  DeviceBuffer d(ctx(), n * 24), e(ctx(), n * 12), back(ctx(), n * 24);
  d.upload(recs);
  auto c = ctx().census(d.ptr(), n);
  uint64_t o[3] = {0, 0, 0}, a[3] = {~0ull, ~0ull, ~0ull};
  for (const Record& r : recs) { o[0] |= r.barcode; o[1] |= r.umi; o[2] |= r.index; a[0] &= r.barcode; a[1] &= r.umi; a[2] &= r.index; }
  for (int f = 0; f < 3; ++f) { CHECK_EQ(c[f], o[f]); CHECK_EQ(c[3 + f], a[f]); }
  CHECK_EQ(c[6], 0ull);   // the generator's index column increases
  CHECK(c[7] != 0);       // but the records are not sorted
  const ibu_key_plan_t plan = device::Context::key_plan(c.data(), c.data() + 3);
  CHECK_EQ(plan.k, 4u + 3u + 3u);   // 32-bit barcodes, 24-bit UMIs, indices below 2^24
  ctx().compact(plan, d.ptr(), n, e.ptr());
  ctx().expand(plan, e.ptr(), n, back.ptr());
  CHECK_EQ(ctx().first_mismatch(d.ptr(), back.ptr(), n), n);
}
TEST(file_streams_to_and_from_the_device) {
  const size_t n = 300001;
  const Header h(16, 12);
  auto recs = oracle_records(0x1B00004, 0, n, 16, 12);
  const std::string path = tmp_path("stream");
  const RingConfig ring{3, 8192, 2, 0};
  {  // device -> file == host-written file
    DeviceBuffer d(ctx(), n * 24);
    d.upload(recs);
    Writer w = Writer::from_path(path, h);
    auto st = w.write_batch_device(ctx(), d.ptr(), n, &ring);
    w.finish();
    CHECK_EQ(st.records, (uint64_t)n); CHECK_EQ(w.records_written(), (uint64_t)n);
  }
  auto [hh, back] = load_to_vec(path);
  CHECK(hh == h); CHECK(back == recs);
  {  // load_to_device == load_to_vec
    auto [hd, dptr, dn] = ctx().load_to_device(path, &ring);
    CHECK(hd == h); CHECK_EQ(dn, n);
    std::vector<Record> got(n);
    ctx().download(static_cast<void*>(got.data()), dptr, n * 24);
    ctx().free(dptr);
    CHECK(got == recs);
  }
  MmapReader m(path);
  orc_reduce want;
  orc_reduce_records(reinterpret_cast<const orc_record*>(recs.data()), n, &want);
  uint64_t count = 0, sum2 = 0;
  for (size_t s = 0; s < 3; ++s) {  // one shard per "GPU": sums add up (mmap.rs:297-307 split)
    auto [r, st] = m.process_device_reduce(ctx(), s, 3, &ring);
    auto range = shard_range(n, 3, s);
    CHECK_EQ(r.count, (uint64_t)(range.second - range.first));
    count += r.count; sum2 += r.sum[2];
  }
  CHECK_EQ(count, want.count); CHECK_EQ(sum2, want.sum[2]);
  {  // the same three workers in ONE call (process_parallel with a GPU per worker; the box has one GPU, so it is listed thrice)
    auto [total, parts] = m.process_devices_reduce({0, 0, 0}, &ring);
    CHECK_EQ(total.count, want.count);
    for (int f = 0; f < 3; ++f) { CHECK_EQ(total.sum[f], want.sum[f]); CHECK_EQ(total.xor_[f], want.xor_[f]); }
    CHECK_EQ(parts.size(), (size_t)3);
    for (size_t s = 0; s < 3; ++s) { auto range = shard_range(n, 3, s); CHECK_EQ(parts[s].count, (uint64_t)(range.second - range.first)); }
    auto [all, per_device] = m.process_devices_reduce();   // every visible device
    CHECK_EQ(all.count, want.count); CHECK_EQ(per_device.size(), (size_t)device::device_count());
    AllocProbe rep{};
    void* p = ctx().alloc_probed(24 * n, 3, &rep);          // resident arrays may choose their placement
    CHECK_EQ(rep.tries, 3u); CHECK(rep.chosen < 3u);
    ctx().free(p);
  }
  DeviceBuffer bc(ctx(), n * 16), umi(ctx(), n * 12), idx(ctx(), n * 8);
  Reader r = Reader::from_path(path);
  r.process_device_decode(ctx(), DecodeSink{bc.as<uint8_t>(), umi.as<uint8_t>(), idx.as<uint64_t>(), n}, &ring);
  std::vector<uint8_t> wbc(n * 16), wumi(n * 12);
  std::vector<uint64_t> widx(n);
  orc_decode_records(reinterpret_cast<const orc_record*>(recs.data()), n, 16, 12, wbc.data(), wumi.data(), widx.data());
  CHECK(bc.download<uint8_t>(wbc.size()) == wbc); CHECK(umi.download<uint8_t>(wumi.size()) == wumi); CHECK(idx.download<uint64_t>(n) == widx);
  unlink(path.c_str());
}
TEST(pull_stream_hands_out_device_batches_in_order) {
  // Reader::read_batch + Iterator on the device (reader.rs:218-242, :279-306): batches concatenate to the file's records; a
  // truncated file throws TruncatedRecord after exactly the whole refills in front of the cut (quirk Q8)
  const Header h(16, 12);
  const size_t n = 3 * 49152 + 777;
  auto recs = oracle_records(0x1B00007, 0, n, 16, 12);
  const std::string path = tmp_path("pull");
  {
    Writer w = Writer::from_path(path, h);
    w.write_batch(recs.data(), recs.size());
    w.finish();
  }
  RingConfig ring{3, 49152, 2, 0};
  {
    Reader r = Reader::from_path(path);
    DeviceStream s = r.device_stream(ctx(), &ring);
    CHECK(s.header() == h);
    std::vector<Record> got;
    uint64_t expect_first = 0;
    while (auto b = s.next()) {
      CHECK_EQ(b->first_index, expect_first);
      std::vector<Record> part(b->n);
      ctx().download(static_cast<void*>(part.data()), b->d_records, b->n * 24);
      got.insert(got.end(), part.begin(), part.end());
      expect_first += b->n;
    }                                                               // (each batch released by its destructor)
    CHECK(got == recs);
    CHECK_EQ(s.stats().records, (uint64_t)n);
  }
  {
    MmapReader m(path);
    uint64_t count = 0;
    ctx().reduce(nullptr, 0);                                       // reset
    for (size_t sh = 0; sh < 2; ++sh) {
      DeviceStream s = m.device_stream(ctx(), sh, 2, &ring);
      while (auto b = s.next()) { check(ibu_reduce(ctx().raw(), b->d_records, b->n, nullptr)); count += b->n; }
    }
    ReduceResult r{};
    check(ibu_reduce_fetch(ctx().raw(), nullptr, &r));
    orc_reduce want;
    orc_reduce_records(reinterpret_cast<const orc_record*>(recs.data()), n, &want);
    CHECK_EQ(count, (uint64_t)n); CHECK_EQ(r.count, want.count);
    for (int f = 0; f < 3; ++f) { CHECK_EQ(r.sum[f], want.sum[f]); CHECK_EQ(r.xor_[f], want.xor_[f]); }
  }
  CHECK_EQ(truncate(path.c_str(), 32 + 24 * n - 5), 0);
  {
    Reader r = Reader::from_path(path);
    DeviceStream s = r.device_stream(ctx(), &ring);
    size_t seen = 0;
    bool threw = false;
    try {
      while (auto b = s.next()) seen += b->n;
    } catch (const IbuError& e) {
      threw = e.kind() == IbuError::TruncatedRecord;
    }
    CHECK(threw);
    CHECK_EQ(seen, (size_t)(3 * 49152));
  }
  unlink(path.c_str());
}
TEST(argument_errors_surface_as_ibu_errors) {
  const Header h(16, 12);
  DeviceBuffer d(ctx(), 24 * 16);
  CHECK_THROWS(InvalidBarcodeLength, ctx().decode_ascii(d.ptr(), 16, Header(0, 12), nullptr, nullptr, nullptr), {});
  CHECK_THROWS(InvalidUmiLength, ctx().decode_ascii(d.ptr(), 16, Header(16, 33), nullptr, nullptr, nullptr), {});
  CHECK_THROWS(InvalidArg, ctx().sort_records(d.ptr(), nullptr, 16), {});
  CHECK_THROWS(InvalidArg, ctx().copy(d.ptr(), d.ptr(), 64), {});
  CHECK_THROWS(Io, ctx().load_to_device("/nonexistent/file.ibu"), {});
}

TEST(bgzf_file_inflates_on_the_device) {
  // ibu_load_bgzf_to_device: the compressed bytes cross the link, every block inflates on the device; the result is load_to_vec of the
  // gunzipped file.  Fixtures from tests/test_cpp.py (the same 50 000 records plain, as BGZF blocks, and that BGZF file cut short).
  const char* dir = std::getenv("IBU_TEST_COMPRESSED_DIR");
  if (!dir) return;
  const std::string d(dir);
  auto recs = oracle_records(11, 0, 50000, 16, 12);
  const RingConfig ring{3, 8192, 2, 0};
  StreamStats st{};
  auto [h, dptr, n] = ctx().load_bgzf_to_device(d + "/a.bgz", &ring, &st);
  CHECK(h == Header(16, 12)); CHECK_EQ(n, recs.size()); CHECK_EQ(st.records, (uint64_t)recs.size());
  std::vector<Record> got(n);
  ctx().download(static_cast<void*>(got.data()), dptr, n * 24);
  ctx().free(dptr);
  CHECK(got == recs);
  {  // two shards of the same file: the ranges of process_parallel, each loaded on its own
    size_t at = 0;
    for (size_t sh = 0; sh < 2; ++sh) {
      auto [hs, ps, ns, first] = ctx().load_bgzf_shard_to_device(d + "/a.bgz", sh, 2, &ring);
      CHECK(hs == Header(16, 12)); CHECK_EQ(first, (uint64_t)at); CHECK_EQ(ns, sh ? recs.size() - recs.size() / 2 : recs.size() / 2);
      std::vector<Record> part(ns);
      ctx().download(static_cast<void*>(part.data()), ps, ns * 24);
      ctx().free(ps);
      CHECK(std::equal(part.begin(), part.end(), recs.begin() + (ptrdiff_t)at));
      at += ns;
    }
    CHECK_EQ(at, recs.size());
  }
  CHECK_THROWS(Niffler, ctx().load_bgzf_to_device(d + "/cut.bgz"), {});     // ends inside a block
  CHECK_THROWS(Niffler, ctx().load_bgzf_to_device(d + "/a.gz"), {});        // an ordinary gzip member: the Reader's business
  CHECK_THROWS(Niffler, ctx().load_bgzf_to_device(d + "/plain.ibu"), {});   // not compressed
}

int main(int argc, char** argv) {
  if (device::device_count() == 0) { std::fprintf(stderr, "no GPU: the device tests cannot run\n"); return 2; }
  return run_all(argc, argv);
}
