"""GPU parity for the device sort by (barcode, umi, index) — the order `#[derive(Ord)]` gives Record
(record.rs:58-66, test record.rs:184-232) and the header's sorted flag promises (header.rs:111-113).
Identical records are indistinguishable, so the sorted byte string is unique: the bar is equality with
the oracle's qsort under record_cmp, byte for byte."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x1B00005
# around the 128-record census tile, the scatter tiles (2560 records since round 3, 2 Ki before; 5 Ki elements on compact keys; other shapes), the
# grid's multiple of eight tiles and the scan block of 256 tiles (524 288 records / 1 310 720 elements; 1024 tiles before)
SIZES = [0, 1, 2, 3, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 2047, 2048, 2049, 2559, 2560, 2561, 4096, 5119, 5120, 5121, 16_383, 16_384, 16_385,
         32_768, 100_000, 524_287, 524_288, 524_289, 655_359, 655_360, 655_361, 1_000_003, 1_310_719, 1_310_720, 1_310_721, 2_097_151, 2_097_152, 2_097_153,
         5_000_001]


@pytest.fixture(scope="module")
def ia():
    import ibu_amd
    return ibu_amd


@pytest.fixture(scope="module")
def ctx(ia):
    c = ia.Context(0)
    yield c
    c.close()


def _sort_on_device(ctx, recs):
    n = len(recs)
    d = ctx.upload(recs) if n else ctx.alloc(16)
    t = ctx.alloc(max(n, 1) * 24)
    ctx.sort_records(d, t, n)
    ctx.synchronize()
    return d.download(count=n * 24).tobytes(), d


def _shuffled(oracle, n, bc_len, umi_len, seed=SEED):
    recs = oracle.generate(seed, 0, n, bc_len, umi_len)
    np.random.default_rng(n).shuffle(recs)  # the generator's index column is already increasing
    return recs


@pytest.fixture(scope="module")
def ctx24(ia):
    """A context that never takes the compact-key path (sort_compact = 0) nor the prefix + finish path (sort_hybrid = 0): the
    plain 24-byte LSD passes stay covered."""
    c = ia.Context(0)
    c.set_option("sort_compact", 0)
    c.set_option("sort_hybrid", 0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_pf(ia):
    """24-byte records, prefix + finish whenever it saves a pass (sort_hybrid = 2): LSD over the top P varying bytes, then
    ibu_k_sort_finish completes every run of equal prefix in LDS — or overflows on long runs and falls back to all passes."""
    c = ia.Context(0)
    c.set_option("sort_compact", 0)
    c.set_option("sort_hybrid", 2)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx64(ia):
    """Every sort through the 64-bit index kernels (sort_idx64 = 1: what inputs of 2^32 records and more take — compact-key passes,
    24-byte passes, position tables — forced at sizes the oracle can check byte for byte), speculation from 131 072 records."""
    c = ia.Context(0)
    c.set_option("sort_idx64", 1)
    yield c
    c.close()


@pytest.mark.parametrize("n", [0, 1, 129, 5121, 70_001, 131_072, 524_289, 1_310_721, 3_000_001])
@pytest.mark.parametrize("lens", [(16, 12), (32, 12), (32, 32)])
def test_sort_with_64_bit_indices(ctx64, oracle, n, lens, capfd):
    recs = _shuffled(oracle, n, *lens)
    if lens == (32, 32) and n > 1:
        recs["index"] = np.random.default_rng(n).integers(0, 2**64, n, dtype=np.uint64)
    elif n:
        recs["index"] = np.random.default_rng(n).integers(0, 2**30, n, dtype=np.uint64)
    capfd.readouterr()
    got, d = _sort_on_device(ctx64, recs)
    trace = capfd.readouterr().err
    assert got == oracle.sort_records(recs).tobytes()
    if n >= 131_072 and lens != (32, 32):
        assert "path=compact-prefix+finish" in trace, trace
    asc = oracle.sort_records(recs)
    assert _sort_on_device(ctx64, asc)[0] == asc.tobytes()         # already sorted
    idx_order = recs.copy()
    idx_order["index"] = np.arange(n, dtype=np.uint64)
    assert _sort_on_device(ctx64, idx_order)[0] == oracle.sort_records(idx_order).tobytes()


@pytest.mark.parametrize("n", SIZES)
def test_sort_random_16_12(ctx, ctx24, ctx_pf, oracle, n):
    recs = _shuffled(oracle, n, 16, 12)
    want = oracle.sort_records(recs).tobytes()
    got, d = _sort_on_device(ctx, recs)          # 11 varying bytes: 12-byte elements (compact-key passes)
    assert got == want
    assert ctx.is_sorted(d, n)
    assert _sort_on_device(ctx24, recs)[0] == want
    assert _sort_on_device(ctx_pf, recs)[0] == want
    idx_order = recs.copy()
    idx_order["index"] = np.arange(n, dtype=np.uint64)   # input in index order: the index bytes ride along unsorted
    want = oracle.sort_records(idx_order).tobytes()
    assert _sort_on_device(ctx, idx_order)[0] == want
    assert _sort_on_device(ctx24, idx_order)[0] == want
    assert _sort_on_device(ctx_pf, idx_order)[0] == want


def _full_range(n, seed):
    """(32,32) records with every one of the 24 bytes varying: BASELINE configs[2]'s widths with full-range values."""
    import ibu_amd as ia
    rng = np.random.default_rng(seed)
    recs = np.empty(n, dtype=ia.REC_DTYPE)
    for f in ("barcode", "umi", "index"):
        recs[f] = rng.integers(0, 2**64, n, dtype=np.uint64)
    return recs


@pytest.mark.parametrize("n", [300, 2047, 2048, 2049, 2560, 2561, 4096, 70_001, 1_000_003, 3_000_001])
def test_wide_keys_take_prefix_and_finish(ctx, ctx24, oracle, n, capfd):
    """More than 16 varying key bytes (full-range (32,32) records: 24): instead of one LSD pass per byte, P passes over the
    most significant varying bytes and one finishing pass (ibu_k_sort_finish).  Same bytes as the oracle's qsort and as
    the plain passes; the trace says which path ran."""
    recs = _full_range(n, n)
    want = oracle.sort_records(recs).tobytes()
    capfd.readouterr()
    got, d = _sort_on_device(ctx, recs)
    trace = capfd.readouterr().err
    assert got == want and ctx.is_sorted(d, n)
    P = 1 if n < 9 * 256 else 2 if n < 9 * 65_536 else 3   # the fewest prefix bytes that leave at most 8 records per segment
    assert f"path=prefix+finish prefix_passes={P} of 24 varying bytes" in trace, trace
    assert _sort_on_device(ctx24, recs)[0] == want


FIN_T, FIN_M = 1024, 256     # ibu_k_sort_finish: records per tile, longest run of equal prefix it accepts (= its look-ahead)


def test_wide_keys_prefix_and_finish_segment_edges(ctx_pf, oracle, ia, capfd):
    """Runs of equal prefix (segments) of every awkward shape: exactly the FIN_M records the finishing kernel accepts,
    one more (overflow -> all passes), runs crossing the tile boundaries, a run reaching the end of the array, exact
    duplicates inside a run, the array ending inside the look-ahead window — with one prefix byte (fewer than 2304
    records) and with two."""
    rng = np.random.default_rng(99)
    M, T = FIN_M, FIN_T

    def build(run_lengths, dup=False):
        recs = _full_range(sum(run_lengths), 7)
        pos = 0
        for k, m in enumerate(run_lengths):                 # the two top bytes of the barcode = the run's id (k, k): one prefix per run
            recs["barcode"][pos:pos + m] = (np.uint64(k * 257) << np.uint64(48)) | (recs["barcode"][pos:pos + m] & np.uint64((1 << 48) - 1))
            pos += m
        if dup:
            recs[1::2] = recs[0:len(recs[1::2]) * 2:2]                # every record twice, in place: equal neighbours inside runs
        rng.shuffle(recs)
        return recs

    cases = [([M] * 8, False), ([M + 1] + [100] * 17, True), ([150, 200, 250, M, 168, 10] + [7] * 30, False),   # one prefix byte (n < 2304)
             ([200] * 11 + [90], False), ([M] * 4 + [M] * 4 + [1], False), ([100] * 3 + [2000], True),
             ([M] * 40 + [3] * 200, False), ([40] * 120 + [M + 1] + [40] * 100, True), ([2, M - 2, M, 1, M - 1, M] * 9, False),   # two
             ([1] * 150 + [254] * 100, False), ([100] * 200 + [4 * T], True)]
    for lengths, overflow in cases:
        assert len(lengths) <= 255 + 1 and (sum(lengths) < 2304 or 2304 <= sum(lengths) < 9 * 65_536)
        for dup in (False, True):
            recs = build(lengths, dup)
            capfd.readouterr()
            got, _ = _sort_on_device(ctx_pf, recs)
            trace = capfd.readouterr().err
            assert got == oracle.sort_records(recs).tobytes(), (lengths, dup)
            assert ("overflowed" in trace) == overflow and (("path=prefix+finish" in trace) != overflow), (lengths, trace)
            assert f"prefix_passes={1 if len(recs) < 2304 else 2} " in trace or overflow, trace


def test_wide_keys_heavy_prefixes_fall_back_to_all_passes(ctx, oracle, capfd):
    """Keys that are not well spread — here two values of the most significant varying bytes — make runs far longer than the
    finishing kernel accepts: it raises its overflow flag and the sort runs every pass on the prefix-sorted records."""
    n = 300_007
    recs = _full_range(n, 3)
    recs["barcode"] = (recs["barcode"] & np.uint64((1 << 40) - 1)) | (np.uint64(0xABCDEF) << np.uint64(40)) * (recs["index"] & np.uint64(1))
    capfd.readouterr()
    got, d = _sort_on_device(ctx, recs)
    trace = capfd.readouterr().err
    assert got == oracle.sort_records(recs).tobytes()
    assert "prefix+finish overflowed" in trace and "path=24-byte" in trace, trace


@pytest.mark.parametrize("kind", ["two_prefixes", "whitelist"])
def test_wide_keys_sample_estimate_avoids_the_overflow(ctx, oracle, kind, capfd):
    """From 1.05 M records on the 24-byte path asks the sample ranges before it commits to a prefix: keys with two distinct
    top bytes, or barcodes from a whitelist of 4000, take a longer prefix (or all passes) at once instead of overflowing."""
    n = 2_000_003
    recs = _full_range(n, 17)
    rng = np.random.default_rng(5)
    if kind == "two_prefixes":
        recs["barcode"] = (recs["barcode"] & np.uint64((1 << 40) - 1)) | (np.uint64(0xABCDEF) << np.uint64(40)) * (recs["index"] & np.uint64(1))
    else:
        wl = recs["barcode"][:4000].copy()
        recs["barcode"] = wl[rng.integers(0, 4000, n)]
    capfd.readouterr()
    got, d = _sort_on_device(ctx, recs)
    trace = capfd.readouterr().err
    assert got == oracle.sort_records(recs).tobytes()
    assert "sample estimate" in trace and "overflowed" not in trace, trace


@pytest.mark.parametrize("n", [9_000, 300_007, 2_000_003])
def test_wide_keys_long_runs_in_order_pass_through(ctx, oracle, n, capfd):
    """Runs of equal prefix far longer than the finishing kernel ranks (here 5000 records with one barcode and one UMI, spanning
    several of its tiles) are no overflow when they are in order already — the stable prefix passes keep the input's index order —
    and an inversion inside such a run still is one."""
    recs = _full_range(n, 23)
    recs["barcode"] &= np.uint64((1 << 63) - 1)                     # the long run's prefix is its own: top bit set only there
    a = n // 3
    recs["barcode"][a:a + 5000] = np.uint64(0xF123456789ABCDEF)
    recs["umi"][a:a + 5000] = np.uint64(77)
    recs["index"][a:a + 5000] = np.arange(5000, dtype=np.uint64) * np.uint64(3)
    want = oracle.sort_records(recs).tobytes()
    capfd.readouterr()
    got, _ = _sort_on_device(ctx, recs)
    trace = capfd.readouterr().err
    assert got == want
    assert "path=prefix+finish" in trace and "overflowed" not in trace, trace
    recs["index"][a + 4000], recs["index"][a + 4001] = recs["index"][a + 4001], recs["index"][a + 4000]   # one inversion deep inside the run
    capfd.readouterr()
    got, _ = _sort_on_device(ctx, recs)
    trace = capfd.readouterr().err
    assert got == oracle.sort_records(recs).tobytes()
    assert "overflowed" in trace, trace


def test_wide_keys_of_a_shard_at_an_odd_record(ctx, oracle):
    n = 200_001
    recs = _full_range(n, 11)
    d, t = ctx.alloc((n + 1) * 24), ctx.alloc((n + 1) * 24)
    ctx.copy(d.ptr + 24, ctx.upload(recs), n * 24)
    ctx.sort_records(d.ptr + 24, t.ptr + 24, n)
    ctx.synchronize()
    assert d.download(count=n * 24, offset=24).tobytes() == oracle.sort_records(recs).tobytes()


@pytest.mark.parametrize("compact", [2, 3, 4, 5, 6, 7, 8])
def test_sort_compact_tile_shapes(ia, oracle, compact):
    """The tile shapes of the compact-key passes (ibu_ctx_set_option "sort_compact") are the same algorithm: same bytes."""
    c = ia.Context(0)
    try:
        c.set_option("sort_compact", compact)
        for n in (1, 4097, 8193, 70_001, 3_000_001):
            recs = _shuffled(oracle, n, 16, 12)
            recs["index"] = np.random.default_rng(n).integers(0, 2**30, n, dtype=np.uint64)
            assert _sort_on_device(c, recs)[0] == oracle.sort_records(recs).tobytes(), (compact, n)
        wide = _shuffled(oracle, 70_001, 32, 12)           # 8 + 3 + 3 varying bytes: the 16-byte element shapes (1 .. 4)
        assert _sort_on_device(c, wide)[0] == oracle.sort_records(wide).tobytes(), compact
        with pytest.raises(ia.IbuError):
            c.set_option("sort_compact", 99)
    finally:
        c.close()


@pytest.fixture(scope="module")
def ctx_guess(ia):
    """A context whose sorts speculate from 131 072 records on (which is also the library's default since round 3): compress on a sampled guess
    of the varying bytes, exact census in the same pass."""
    c = ia.Context(0)
    c.set_option("sort_guess", 131_072)
    c.set_option("sort_hybrid", 0)                     # the plain element passes behind the guess (prefix + finish: ctx_guess_pf)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_guess_pf(ia):
    """Speculation from 131 072 records on AND prefix + finish on the elements (the default for large inputs): a pair count
    over the sample ranges estimates how long the runs of equal prefix will be; short runs -> P element passes over the
    top P bytes + ibu_k_sort_finish_elems, long runs -> the plain passes."""
    c = ia.Context(0)
    c.set_option("sort_guess", 131_072)
    yield c
    c.close()


@pytest.mark.parametrize("n", [131_072, 200_003, 1_000_003])
@pytest.mark.parametrize("lens", [(16, 12), (32, 12)])   # 10-11 varying bytes: 12-byte elements; 14-15: 16-byte elements
@pytest.mark.parametrize("case", ["random_index", "index_order", "guess_misses_a_umi_byte", "guess_misses_an_index_byte",
                                  "index_order_broken_outside_the_samples", "already_sorted", "one_umi_byte_constant_in_the_samples",
                                  "sorted_in_the_samples_only"])
def test_sort_on_a_sampled_guess(ctx_guess, ctx24, oracle, n, lens, case, capfd):
    """The speculative path: the samples are the first, the middle and the last 32 768 records.  Whatever the samples
    suggest, the result is the oracle's — a guess that does not cover the truth is detected by the exact census."""
    from tests import keyplan_np as kp
    recs = _shuffled(oracle, n, *lens)
    rng = np.random.default_rng(n)
    quarter = n // 4 + 7                                   # a row no sample range contains
    if n < 140_000:
        quarter = 32_768 + (n // 2 - 32_768) // 2          # between the first and the middle sample
    assert 32_768 <= quarter < n // 2 - 1
    if case == "random_index":
        recs["index"] = rng.integers(0, 2**30, n, dtype=np.uint64)
    elif case == "index_order":
        recs["index"] = np.arange(n, dtype=np.uint64)
    elif case == "guess_misses_a_umi_byte":
        recs["index"] = rng.integers(0, 2**30, n, dtype=np.uint64)
        recs["umi"][quarter] |= np.uint64(1) << np.uint64(44)           # byte 5 of the UMI varies in ONE record
    elif case == "guess_misses_an_index_byte":
        recs["index"] = rng.integers(0, 2**16, n, dtype=np.uint64)
        recs["index"][quarter] = 2**40 + 5
    elif case == "index_order_broken_outside_the_samples":
        recs["index"] = np.arange(n, dtype=np.uint64)
        recs["barcode"] %= 50                                  # ties, so that the index order decides
        recs["umi"] %= 3
        recs["index"][quarter], recs["index"][quarter + 1] = recs["index"][quarter + 1], recs["index"][quarter]
        recs["barcode"][quarter + 1], recs["umi"][quarter + 1] = recs["barcode"][quarter], recs["umi"][quarter]
    elif case == "already_sorted":
        recs = oracle.sort_records(recs)
    elif case == "sorted_in_the_samples_only":                 # one swap where no sample looks: the census finds it, no speculation
        recs = oracle.sort_records(recs)
        recs[[quarter, quarter + 1]] = recs[[quarter + 1, quarter]]
        assert recs[quarter].tobytes() != recs[quarter + 1].tobytes()
    elif case == "one_umi_byte_constant_in_the_samples":
        recs["index"] = np.arange(n, dtype=np.uint64)
        recs["umi"] &= np.uint64(0xFFFF)
        recs["umi"][quarter] |= np.uint64(0x7F0000)
    want = oracle.sort_records(recs).tobytes()
    capfd.readouterr()
    assert _sort_on_device(ctx_guess, recs)[0] == want
    trace = capfd.readouterr().err                         # IBU_TRACE_SORT=1 (conftest): which path the library took
    k = kp.Plan(*kp.census_words(recs)).k                  # varying key bytes of the whole input
    width = 12 if k <= 12 else 16
    expect = {"random_index": f"path=compact-speculated element_bytes={width} passes={k} first_digit_guess=hit",
              "index_order": f"path=compact-speculated element_bytes={width}",
              "guess_misses_a_umi_byte": "guess did not cover",
              "guess_misses_an_index_byte": "guess did not cover",
              "index_order_broken_outside_the_samples": "first_digit_guess=miss",
              "already_sorted": "samples in order: read-only census first",   # round 3: sorted input costs ONE read-only pass again
              "sorted_in_the_samples_only": "samples in order: read-only census first",
              "one_umi_byte_constant_in_the_samples": "guess did not cover"}[case]
    assert expect in trace, trace
    if case == "already_sorted":
        assert "already sorted" in trace and "speculative compress pass was spent" not in trace and "path=" not in trace, trace
    if case == "sorted_in_the_samples_only":
        assert f"path=compact element_bytes={width}" in trace and "speculated" not in trace, trace
    if "did not cover" in expect:                          # ... and the sort went on from the exact census
        assert (f"path=compact element_bytes={12 if k <= 12 else 16}" if k <= 16 else "path=24-byte") in trace, trace
    assert _sort_on_device(ctx24, recs)[0] == want


@pytest.mark.parametrize("n", [8191, 8192, 8193, 20_001, 70_001, 131_071])
@pytest.mark.parametrize("lens", [(16, 12), (32, 12)])
@pytest.mark.parametrize("case", ["random_index", "index_order", "heavy_run", "few_barcodes"])
def test_sort_prefix_and_finish_below_the_speculation_threshold(ctx, oracle, n, lens, case, capfd):
    """Inputs too small for the sample census (< 2^17 records) plan from the exact census; from 8192 records the same pair-count
    estimate decides between the top-P passes + finishing kernel and all passes.  Always the oracle's bytes."""
    recs = _shuffled(oracle, n, *lens)
    rng = np.random.default_rng(n + len(case))
    recs["index"] = np.arange(n, dtype=np.uint64) if case == "index_order" else rng.integers(0, 2**30, n, dtype=np.uint64)
    if case == "heavy_run":                                    # 3000 equal (barcode, umi) pairs, spread: every sample range sees them
        at = rng.choice(n, 3000, replace=False)
        recs["barcode"][at] = recs["barcode"][0]
        recs["umi"][at] = recs["umi"][0]
    elif case == "few_barcodes":                               # 16 distinct barcodes: a barcode-only prefix would leave runs of n / 16
        recs["barcode"] = recs["barcode"][:16][rng.integers(0, 16, n)]
    want = oracle.sort_records(recs).tobytes()
    capfd.readouterr()
    got, _ = _sort_on_device(ctx, recs)
    trace = capfd.readouterr().err
    assert got == want
    if not trace:
        pytest.skip("no trace: IBU_TRACE_SORT is not set")
    if n < 8192:
        assert "prefix+finish" not in trace and "path=compact element_bytes" in trace, trace
    elif case in ("random_index", "index_order"):
        assert "path=compact-prefix+finish" in trace and "(exact plan)" in trace and "overflowed" not in trace, trace
    elif case == "heavy_run":                                  # seen by the estimate: a prefix that reaches into the index bytes, or none
        assert "overflowed" not in trace, trace
    else:
        assert "overflowed" not in trace, trace
        if "path=compact-prefix+finish" in trace:
            assert int(trace.split("prefix_passes=")[1].split()[0]) > (4 if lens[0] == 16 else 8), trace


@pytest.mark.parametrize("n", [131_072, 200_003, 1_000_003, 5_000_001])
@pytest.mark.parametrize("lens", [(16, 12), (32, 12)])   # 12-byte and 16-byte elements
@pytest.mark.parametrize("case", ["random_index", "index_order", "whitelist_barcodes", "heavy_run_outside_the_samples", "duplicates",
                                  "guess_misses_a_umi_byte", "one_heavy_barcode", "heavy_barcode_outside_the_samples",
                                  "heavy_run_in_read_order"])
def test_sort_prefix_and_finish_on_elements(ctx_guess_pf, ctx24, oracle, n, lens, case, capfd):
    """The compact-key sort of large inputs: well-spread keys (short estimated runs) take P prefix passes + the finishing
    kernel; keys with few distinct prefixes (barcodes from a whitelist) take a longer prefix or the plain passes; a heavy run
    the samples did not see overflows the finishing kernel, and all passes run.  Always the oracle's bytes."""
    recs = _shuffled(oracle, n, *lens)
    rng = np.random.default_rng(n + len(case))
    recs["index"] = rng.integers(0, 2**30, n, dtype=np.uint64)
    # a row that NO sample sees: neither the census' three ranges (first / middle / last 32 768 records) nor the pair estimate's
    # evenly spaced ranges of 2048 records (sort.hip: 48 of them, fewer while the hash tables must fit in 24 n bytes)
    slots = 1 << 18
    while slots > 1024 and slots * 96 + 128 > 24 * n:
        slots >>= 1
    nranges = 48
    while nranges > 3 and nranges * 2048 > slots * 3 // 8:
        nranges //= 2
    while nranges > 3 and nranges * 2048 > n // 32:
        nranges //= 2
    stride = (n - 2048) // (nranges - 1)
    quarter = next(q for q in range(32_768 + 64, n // 2 - 3064, 997)
                   if all(q + 3000 + 64 <= r * stride or q >= r * stride + 2048 + 64 for r in range(nranges)))
    if case == "index_order":
        recs["index"] = np.arange(n, dtype=np.uint64)
    elif case == "whitelist_barcodes":                         # ~n / 500 distinct barcodes: runs of ~500 under a barcode-only prefix
        wl = np.unique(recs["barcode"][: max(n // 500, 2)])
        recs["barcode"] = wl[rng.integers(0, len(wl), n)]
    elif case == "heavy_run_outside_the_samples":              # 3000 records with one (barcode, umi) where no sample looks
        span = 3000
        recs["barcode"][quarter:quarter + span] = recs["barcode"][quarter]
        recs["umi"][quarter:quarter + span] = recs["umi"][quarter]
    elif case == "heavy_run_in_read_order":                     # records in read order (index increasing), 3000 of them with one
        recs["index"] = np.arange(n, dtype=np.uint64)          # (barcode, umi): the run is far longer than the finishing kernel ranks,
        span = 3000                                            # but the stable passes leave it in
        top = np.uint64(1) << np.uint64(2 * lens[0] - 1)       # index order: passed through as it is.  (Its prefix is its own: the
        recs["barcode"] &= top - np.uint64(1)                  # top barcode bit is set only there — records that merely share the
        recs["barcode"][quarter:quarter + span] = top | np.uint64(0x1234567)   # prefix would sit between them in input order.)
        recs["umi"][quarter:quarter + span] = recs["umi"][quarter]
    elif case == "heavy_barcode_outside_the_samples":           # 3000 records of one barcode (UMIs stay random) where no sample looks
        span = 3000
        recs["barcode"][quarter:quarter + span] = recs["barcode"][quarter]
    elif case == "duplicates":
        recs[1::2] = recs[0:len(recs[1::2]) * 2:2]
        rng.shuffle(recs)
    elif case == "one_heavy_barcode":                          # 2 % of the records carry one barcode, spread over the whole input
        recs["barcode"][rng.random(n) < 0.02] = recs["barcode"][0]
    elif case == "guess_misses_a_umi_byte":
        recs["umi"][quarter] |= np.uint64(1) << np.uint64(44)
    want = oracle.sort_records(recs).tobytes()
    capfd.readouterr()
    got, d = _sort_on_device(ctx_guess_pf, recs)
    trace = capfd.readouterr().err
    assert got == want and ctx_guess_pf.is_sorted(d, n)
    if case in ("random_index", "index_order", "duplicates"):
        assert "path=compact-prefix+finish" in trace and "overflowed" not in trace, trace
    elif case == "heavy_run_outside_the_samples":               # 3000 records with ONE (barcode, umi): no prefix short of the index splits
        assert "path=compact-prefix+finish" in trace and "overflowed" in trace, trace   # them, so the retry overflows too (or is not worth it)
        assert "all " in trace.split("overflowed")[-1], trace
    elif case == "heavy_run_in_read_order":                     # a long run that is already in order is not an overflow
        assert "path=compact-prefix+finish" in trace and "overflowed" not in trace, trace
    elif case == "heavy_barcode_outside_the_samples":           # ... with random UMIs the retry's longer prefix (barcode + UMI bytes) succeeds
        assert "path=compact-prefix+finish" in trace and trace.count("overflowed") == 1 and "retrying with prefix_passes" in trace, trace
    elif case == "guess_misses_a_umi_byte":
        assert "guess did not cover" in trace and "prefix+finish" not in trace, trace
    elif case == "one_heavy_barcode":                          # the sample's most frequent prefix says so BEFORE anything overflows:
        assert "overflowed" not in trace, trace                #   a prefix that also splits the heavy barcode, or the plain passes
        if "path=compact-prefix+finish" in trace:
            p = int(trace.split("prefix_passes=")[1].split()[0])
            assert p > (4 if lens[0] == 16 else 8), trace
    elif case == "whitelist_barcodes":                         # never a 4-byte (barcode-only) prefix: longer, or the plain passes
        assert "overflowed" not in trace, trace
        if "path=compact-prefix+finish" in trace:
            p = int(trace.split("prefix_passes=")[1].split()[0])
            assert p > (4 if lens[0] == 16 else 8), trace


@pytest.mark.parametrize("n", [200_003, 1_000_003, 5_000_001])
@pytest.mark.parametrize("lens", [(16, 12), (32, 12)])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_finishing_kernel_ranks_runs_of_every_length_up_to_its_limit(ctx_guess_pf, oracle, n, lens, seed, capfd):
    """Runs of equal prefix of 3 .. 256 elements — the worklist of the finishing kernel, and beyond 64 the word-by-word walk
    through its head bitmap — placed where no sample looks, so that the estimate still takes the short prefix: each run is one
    barcode (its own) with random UMIs, unsorted inside.  All ranked, no overflow, the oracle's bytes."""
    recs = _shuffled(oracle, n, *lens)
    rng = np.random.default_rng(n + seed)
    recs["index"] = rng.integers(0, 2**30, n, dtype=np.uint64)
    slots = 1 << 18
    while slots > 1024 and slots * 96 + 128 > 24 * n:
        slots >>= 1
    nranges = 48
    while nranges > 3 and nranges * 2048 > slots * 3 // 8:
        nranges //= 2
    while nranges > 3 and nranges * 2048 > n // 32:
        nranges //= 2
    stride = (n - 2048) // (nranges - 1)
    at = next(q for q in range(32_768 + 64, n // 2 - 3064, 997)
              if all(q + 3000 + 64 <= r * stride or q >= r * stride + 2048 + 64 for r in range(nranges)))
    lengths = [3, 4, 7, 17, 63, 64, 65, 66, 127, 128, 129, 200, 255, 256, 256, 31, 2, 1, 193]
    rng.shuffle(lengths)
    pos = at
    top = np.uint64(1) << np.uint64(2 * lens[0] - 1)
    shift = np.uint64(2 * lens[0] - 16)                        # the shortest prefix the estimate may take: two bytes
    for k, ln in enumerate(lengths):                           # run k: barcode drawn at random (its place in the sorted order: anywhere)
        b = rng.integers(0, 1 << (2 * lens[0] - 1), dtype=np.uint64) | np.uint64(k)
        others = (recs["barcode"] >> shift) == (b >> shift)   # ... and alone under every prefix: a run of 256 must stay one of 256
        recs["barcode"][others] |= top
        recs["barcode"][pos:pos + ln] = b
        pos += ln
    assert pos - at <= 3000
    want = oracle.sort_records(recs).tobytes()
    capfd.readouterr()
    got, _ = _sort_on_device(ctx_guess_pf, recs)
    trace = capfd.readouterr().err
    assert got == want
    if trace:
        assert "path=compact-prefix+finish" in trace and "overflowed" not in trace, trace


@pytest.mark.parametrize("lens,n", [((32, 32), 1_200_007), ((32, 12), 1_000_003), ((32, 12), 200_003)])
@pytest.mark.parametrize("index", ["random", "read_order"])
def test_sort_prefixes_longer_than_eight_bytes(ctx, oracle, lens, n, index, capfd):
    """Wide barcodes from a whitelist: all eight barcode bytes vary and say little, so the shortest prefix with short runs reaches
    into the UMI — 10 bytes.  The estimate looks at prefixes of 9 .. 16 bytes when 1 .. 8 found nothing (24-byte records at
    (32,32): 20 varying bytes; 16-byte elements at (32,12): 15), and takes them where that still saves passes."""
    recs = _shuffled(oracle, n, *lens)
    rng = np.random.default_rng(n)
    recs["barcode"] = recs["barcode"][:16][rng.integers(0, 16, n)]
    recs["index"] = rng.integers(0, 2**30, n, dtype=np.uint64) if index == "random" else np.arange(n, dtype=np.uint64)
    want = oracle.sort_records(recs).tobytes()
    capfd.readouterr()
    got, _ = _sort_on_device(ctx, recs)
    trace = capfd.readouterr().err
    assert got == want
    if not trace:
        pytest.skip("no trace: IBU_TRACE_SORT is not set")
    assert "overflowed" not in trace, trace
    if lens == (32, 32):                                       # 8 + 8 key bytes (+ 4 index bytes when those are random)
        assert "path=prefix+finish prefix_passes=10 of " + ("20" if index == "random" else "16") in trace, trace
    elif index == "random":                                    # 8 + 3 key bytes + 4 index bytes in a 16-byte element
        assert "path=compact-prefix+finish element_bytes=16 prefix_passes=10 of 15" in trace, trace
    else:                                                      # 11 passes in read order: a 10-byte prefix saves nothing
        assert "prefix+finish" not in trace, trace


@pytest.mark.parametrize("seed", range(int(os.environ.get("IBU_FUZZ_SEEDS", "32"))))   # a soak run sets more (profiles/README.md r03_soak)
def test_sort_fuzz_key_structures(ctx, ctx_pf, ctx64, oracle, ia, seed):
    """Seeded fuzz over what decides the sort's path: which key bytes vary (1 .. 24 of them, anywhere in the record), how the values
    are distributed (uniform / a few heavy values / Zipf-like / blocks of equal keys), whether stretches of the input are already in
    order, and sizes on both sides of the speculation threshold.  Default context and forced prefix + finish: the oracle's bytes."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([70_001, 131_072, 200_003, 524_289, 1_200_007] if seed < 24 else [8192, 9001, 30_011, 131_071, 262_144, 2_000_003]))
    nbytes = int(rng.integers(1, 25))
    which = np.sort(rng.permutation(24)[:nbytes])
    raw = np.tile(rng.integers(0, 256, 24, dtype=np.uint8), (n, 1))
    dist = seed % 4
    for b in which:
        if dist == 0:
            col = rng.integers(0, 256, n, dtype=np.uint8)
        elif dist == 1:                                        # a few heavy values
            col = rng.choice(np.array([3, 200, 77, 255, 0], dtype=np.uint8), n, p=[0.6, 0.2, 0.1, 0.05, 0.05])
        elif dist == 2:                                        # Zipf-like
            col = np.minimum(rng.zipf(1.3, n), 255).astype(np.uint8)
        else:                                                  # blocks of equal values
            col = np.repeat(rng.integers(0, 256, n // 997 + 1, dtype=np.uint8), 997)[:n]
        raw[:, b] = col
    raw[0, which], raw[1, which] = 0, 255                      # every chosen byte really varies
    recs = raw.reshape(-1).view(ia.REC_DTYPE).copy()
    if seed % 3 == 0:                                          # a third of the input already in order
        k = n // 3
        recs[k:2 * k] = oracle.sort_records(recs[k:2 * k])
    want = oracle.sort_records(recs).tobytes()
    assert _sort_on_device(ctx, recs)[0] == want, (seed, n, nbytes, dist)
    assert _sort_on_device(ctx_pf, recs)[0] == want, (seed, n, nbytes, dist)
    assert _sort_on_device(ctx64, recs)[0] == want, (seed, n, nbytes, dist)


@pytest.mark.parametrize("n", [2, 129, 5000, 300_007])
@pytest.mark.parametrize("nbytes", [1, 5, 11, 12, 13, 14, 16, 17, 24])
def test_sort_with_scattered_varying_bytes(ctx, oracle, ia, n, nbytes):
    """The compact-key path gathers whichever bytes vary — not only the low bytes of each field — and carries the constant
    ones through the census' AND words; 12 varying bytes still fit a 12-byte element, 13 .. 16 a 16-byte one (whose second
    buffer is the record array itself: odd and even pass counts end differently), 17 take the 24-byte passes."""
    rng = np.random.default_rng(nbytes * 1000 + n)
    which = np.sort(rng.permutation(24)[:nbytes])          # byte positions of the record that vary
    raw = np.tile(rng.integers(0, 256, 24, dtype=np.uint8), (n, 1))   # every other byte: the same value in all records
    raw[:, which] = rng.integers(0, 256, (n, nbytes), dtype=np.uint8)
    raw[0, which], raw[1, which] = 0, 255                   # every chosen byte really varies
    recs = raw.reshape(-1).view(ia.REC_DTYPE)
    assert _sort_on_device(ctx, recs)[0] == oracle.sort_records(recs).tobytes()


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6])
def test_sort_other_tile_shapes(ia, oracle, variant):
    """The A/B tile shapes (ibu_ctx_set_option "sort_variant") are the same algorithm: same bytes."""
    c = ia.Context(0)
    try:
        c.set_option("sort_variant", variant)
        c.set_option("sort_compact", 0)                  # the 24-byte passes are what these shapes belong to
        for n in (1, 4097, 70_001, 3_000_001):
            recs = _shuffled(oracle, n, 16, 12)
            recs["index"] = np.random.default_rng(n).integers(0, 2**30, n, dtype=np.uint64)
            assert _sort_on_device(c, recs)[0] == oracle.sort_records(recs).tobytes(), (variant, n)
        with pytest.raises(ia.IbuError):
            c.set_option("sort_variant", 99)
    finally:
        c.close()


def test_lower_bound_records_matches_partition_point(ctx, oracle, ia):
    """ibu_lower_bound_records (the splitter search of the multi-GPU sample sort) = slice::partition_point(|r| r < key)
    on the sorted records: keys below / above everything, present keys (first of a run of duplicates), absent keys."""
    n = 100_003
    recs = _shuffled(oracle, n, 6, 3)                 # short fields: many exact (barcode, umi) ties
    recs["index"] %= 7
    srt = oracle.sort_records(recs)
    keys = np.concatenate([srt[[0, 1, n // 3, n // 2, n - 1]], ia.records_array([(0, 0, 0), (2**64 - 1, 2**64 - 1, 2**64 - 1), (5, 0, 0),
                                                                                 (int(srt["barcode"][n // 2]), int(srt["umi"][n // 2]) + 1, 0)])])
    d_s, d_k, d_p = ctx.upload(srt), ctx.upload(keys), ctx.alloc(8 * len(keys))
    ctx.lower_bound(d_s, n, d_k, len(keys), d_p)
    got = d_p.download(np.uint64).tolist()
    want = [oracle.lower_bound(srt, k) for k in keys]
    assert got == want
    assert got[0] == 0 and got[6] == n                # all-ones key: past the end
    ctx.lower_bound(d_s, 0, d_k, len(keys), d_p)      # empty shard: every bound is 0
    assert d_p.download(np.uint64).tolist() == [0] * len(keys)
    ctx.lower_bound(d_s, n, d_k, 0, d_p)              # no keys: nothing to do


def test_sort_of_a_shard_at_an_odd_record(ctx, oracle):
    """Records and scratch 8-byte but not 16-byte aligned (a shard starting at an odd record): census peels, the first
    count and every scatter take their 8-byte paths; same bytes."""
    n = 300_001
    recs = _shuffled(oracle, n, 16, 12)
    d, t = ctx.alloc((n + 1) * 24), ctx.alloc((n + 1) * 24)
    ctx.copy(d.ptr + 24, ctx.upload(recs), n * 24)
    ctx.sort_records(d.ptr + 24, t.ptr + 24, n)
    ctx.synchronize()
    assert d.download(count=n * 24, offset=24).tobytes() == oracle.sort_records(recs).tobytes()
    assert ctx.is_sorted(d.ptr + 24, n)


@pytest.mark.parametrize("n", [1, 1025, 70_001])
@pytest.mark.parametrize("lens", [(32, 32), (1, 1), (4, 28), (31, 3)])
def test_sort_other_widths(ctx, oracle, n, lens):
    """(32,32): all 24 digit positions vary (no pass is skipped, an even/odd pass count both occur)."""
    recs = _shuffled(oracle, n, *lens)
    if lens == (32, 32) and n > 1:
        recs["index"] = np.random.default_rng(5).integers(0, 2**64, n, dtype=np.uint64)
    got, _ = _sort_on_device(ctx, recs)
    assert got == oracle.sort_records(recs).tobytes()


def test_sort_is_lexicographic_on_the_reference_example(ctx, oracle, kat, ia):
    """record.rs:184-232: the 8 permutations of {0,1}^3 sort lexicographically; (1,1,0) > (0,1,1)."""
    rows = [(b, u, i) for b in (1, 0) for u in (1, 0) for i in (1, 0)]
    recs = ia.records_array(rows)
    got, _ = _sort_on_device(ctx, recs)
    want = ia.records_array(sorted(rows))
    assert got == want.tobytes() == oracle.sort_records(recs).tobytes()


def test_sort_heavy_duplicates_and_ties(ctx, oracle, ia):
    """Few distinct barcodes / UMIs: long runs of equal high digits, order decided by the low fields."""
    n = 200_003
    rng = np.random.default_rng(11)
    recs = np.empty(n, dtype=ia.REC_DTYPE)
    recs["barcode"] = rng.integers(0, 5, n, dtype=np.uint64) * 0x0101010101010101
    recs["umi"] = rng.integers(0, 3, n, dtype=np.uint64) << 40
    recs["index"] = rng.integers(0, 1000, n, dtype=np.uint64)  # many exact duplicates as well
    got, d = _sort_on_device(ctx, recs)
    assert got == oracle.sort_records(recs).tobytes()
    assert ctx.is_sorted(d, n)


def test_sort_degenerate_inputs(ctx, oracle, ia):
    n = 50_000
    same = ia.records_array([(7, 7, 7)] * n)                       # no digit varies: zero passes
    assert _sort_on_device(ctx, same)[0] == same.tobytes()
    asc = oracle.sort_records(_shuffled(oracle, n, 16, 12))        # already sorted
    assert _sort_on_device(ctx, asc)[0] == asc.tobytes()
    desc = asc[::-1].copy()                                        # reverse sorted
    assert _sort_on_device(ctx, desc)[0] == asc.tobytes()
    one_bit = ia.records_array([(0, 0, i & 1) for i in range(n)])  # a single varying digit: one pass + copy back
    assert _sort_on_device(ctx, one_bit)[0] == oracle.sort_records(one_bit).tobytes()
    hi = ia.records_array([(((i * 2654435761) & 0xFF) << 56, 0, 0) for i in range(n)])  # only the top barcode byte
    assert _sort_on_device(ctx, hi)[0] == oracle.sort_records(hi).tobytes()


def test_sort_skips_index_passes_only_when_input_is_in_index_order(ctx, oracle, ia):
    """Stable LSD + index as the least significant field: input in non-decreasing index order needs no index passes.
    One inversion anywhere (here: the very last pair, across a 32 Ki chunk edge, and in the middle) must switch them on."""
    n = 70_000
    recs = oracle.generate(SEED, 0, n, 16, 12)           # index 0..n-1 increasing, barcodes random
    recs["barcode"] %= 50                                 # many ties on barcode (and some on umi) -> index order matters
    recs["umi"] %= 7
    assert _sort_on_device(ctx, recs)[0] == oracle.sort_records(recs).tobytes()
    dup = recs.copy()
    dup["index"] //= 3                                     # non-decreasing with repeats: still index order
    assert _sort_on_device(ctx, dup)[0] == oracle.sort_records(dup).tobytes()
    for pos in (n - 1, 32_768, n // 2, 1):
        bad = recs.copy()
        bad["index"][pos] = bad["index"][pos - 1] - 1 if bad["index"][pos - 1] else 0
        bad["index"][pos - 1] += 5                         # recs[pos].index < recs[pos-1].index
        bad["barcode"][pos], bad["umi"][pos] = bad["barcode"][pos - 1], bad["umi"][pos - 1]  # a tie decided by the index
        assert _sort_on_device(ctx, bad)[0] == oracle.sort_records(bad).tobytes(), pos


def test_sort_then_write_sorted_file_roundtrip(ctx, oracle, ia, tmp_path):
    """The use the flag exists for: sort on device, write with set_sorted(), read back, still sorted."""
    n = 120_000
    recs = _shuffled(oracle, n, 16, 12)
    _, d = _sort_on_device(ctx, recs)
    h = ia.Header(16, 12)
    h.set_sorted()
    p = tmp_path / "sorted.ibu"
    w = ia.Writer.from_path(p, h)
    w.write_batch_device(ctx, d, n)
    w.finish()
    w.close()
    hh, back = ia.load_to_vec(p)
    assert hh.sorted() and back.tobytes() == oracle.sort_records(recs).tobytes()
    assert oracle.is_sorted(back)


def test_sort_large_permutation_properties(ctx, ia):
    """5e7 records (beyond what the oracle sorts in seconds): sortedness + multiset preserved (count, sums, XORs)."""
    n = 50_000_000
    d, t = ctx.alloc(n * 24), ctx.alloc(n * 24)
    # decreasing index within increasing seed blocks: definitely not sorted, every field varies
    ctx.generate(SEED, 0, n, 16, 12, d)
    before = ctx.reduce(d, n)
    assert not ctx.is_sorted(d, n)
    ctx.sort_records(d, t, n)
    assert ctx.is_sorted(d, n)
    assert ctx.reduce(d, n) == before


def test_sort_argument_errors(ctx, ia):
    d = ctx.alloc(24 * 16)
    with pytest.raises(ia.IbuError) as e:
        ctx.sort_records(d, None, 16)
    assert e.value.kind == "InvalidArg"
    ctx.sort_records(d, None, 1)  # n < 2: nothing to do, scratch not needed


# ---- per-barcode aggregation on sorted records: BarcodeAnalyzer, src/parallel.rs:72-98 ---------------------------
@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 8191, 8192, 8193, 100_000, 1_000_003])
@pytest.mark.parametrize("n_barcodes,n_umis", [(1, 1), (7, 3), (300, 40), (1 << 40, 1 << 20)])
def test_barcode_counts(ctx, oracle, ia, n, n_barcodes, n_umis):
    rng = np.random.default_rng(n + n_barcodes)
    recs = np.empty(n, dtype=ia.REC_DTYPE)
    recs["barcode"] = rng.integers(0, n_barcodes, n, dtype=np.uint64)
    recs["umi"] = rng.integers(0, n_umis, n, dtype=np.uint64)
    recs["index"] = np.arange(n, dtype=np.uint64)
    srt = oracle.sort_records(recs)
    d = ctx.upload(srt) if n else ctx.alloc(16)
    b, c, u = ctx.barcode_counts(d, n)
    wb, wc, wu = oracle.barcode_counts(srt)
    assert b.tobytes() == wb.tobytes()
    assert c.tobytes() == wc.tobytes()
    assert u.tobytes() == wu.tobytes()
    if n:  # the HashMap the reference's processor builds, stated independently
        keys, counts = np.unique(recs["barcode"], return_counts=True)
        assert b.tolist() == keys.tolist() and c.tolist() == counts.tolist()
    b2, c2, u2 = ctx.barcode_counts(d, n, unique_umis=False)
    assert u2 is None and b2.tobytes() == wb.tobytes() and c2.tobytes() == wc.tobytes()


@pytest.mark.parametrize("peel", [0, 1])
def test_barcode_counts_serves_sparse_segments_from_the_stash_and_walks_the_dense_ones(ctx, oracle, ia, peel):
    """The count pass keeps the first 32 run heads of every 8192-record segment; the emit pass writes a segment with at most 32 heads
    from that stash and reads the records again only for the others.  Segments with 0 (one run goes on), 1, 31, 32, 33, 34, 64 and
    8192 heads side by side, a run that crosses a segment boundary or a new one starting exactly on it, with and without the peeled
    first record (which shifts every segment by one row)."""
    import ctypes as C
    rng = np.random.default_rng(3232 + peel)
    seg = 8192
    heads_per_seg = [1, 0, 31, 32, 33, 0, 34, 64, 8192, 2, 32, 0, 33, 1, 5]
    starts_new = [True, False, True, False, True, False, False, True, True, False, True, False, True, True, False]   # a head on the segment's first row
    bc = np.empty(0, np.uint64)
    cur = 10
    for h, first in zip(heads_per_seg, starts_new):
        col = np.full(seg, cur, np.uint64)
        if h:
            rows = np.arange(seg) if h == seg else np.sort(rng.choice(np.arange(1, seg), h - (1 if first else 0), replace=False))
            if first and h != seg:
                rows = np.concatenate(([0], rows))
            step = np.zeros(seg, np.uint64)
            step[rows] = rng.integers(1, 1000, len(rows)).astype(np.uint64)
            col = cur + np.cumsum(step)
            cur = int(col[-1])
        bc = np.concatenate((bc, col.astype(np.uint64)))
    tail = np.full(77, cur + 5, np.uint64)                         # the n % 128 rest: one more run
    bc = np.concatenate((np.full(peel, 3, np.uint64), bc, tail))
    n = len(bc)
    recs = np.empty(n, dtype=ia.REC_DTYPE)
    recs["barcode"] = bc
    recs["umi"] = rng.integers(0, 6, n, dtype=np.uint64)
    recs["index"] = np.arange(n, dtype=np.uint64)
    srt = oracle.sort_records(recs)
    assert srt["barcode"].tobytes() == bc.tobytes()                # (the construction is sorted by barcode already)
    buf = ctx.alloc(24 * (n + 2))
    base = buf.ptr + (24 if peel else 0)                          # peel = 1: start at an odd record (8- not 16-byte aligned)
    ia.lib.ibu_memcpy_h2d(ctx._c, C.c_void_p(base), srt.ctypes.data_as(C.c_void_p), 24 * n, None)
    ctx.synchronize()
    b, c, u = ctx.barcode_counts(base, n)
    wb, wc, wu = oracle.barcode_counts(srt)
    assert (b.tobytes(), c.tobytes(), u.tobytes()) == (wb.tobytes(), wc.tobytes(), wu.tobytes())
    b2, c2, _ = ctx.barcode_counts(base, n, unique_umis=False)
    assert (b2.tobytes(), c2.tobytes()) == (wb.tobytes(), wc.tobytes())
    buf.free()


@pytest.mark.parametrize("n", [1, 2, 127, 128, 129, 255, 8192 + 1, 8192 + 128 + 5, 3 * 8192 + 129, 100_001])
@pytest.mark.parametrize("n_barcodes,n_umis", [(1, 1), (5, 2), (1 << 40, 1 << 20)])
def test_barcode_counts_of_a_shard_at_an_odd_record(ctx, oracle, ia, n, n_barcodes, n_umis):
    """Sorted records that start at an odd record of a larger buffer (8- but not 16-byte aligned): one row is peeled in
    front of the tiled segments, the n % 128 rest follows them; the runs that straddle the three parts are counted once."""
    rng = np.random.default_rng(3 * n + n_barcodes)
    recs = np.empty(n + 1, dtype=ia.REC_DTYPE)
    recs["barcode"] = rng.integers(0, n_barcodes, n + 1, dtype=np.uint64)
    recs["umi"] = rng.integers(0, n_umis, n + 1, dtype=np.uint64)
    recs["index"] = np.arange(n + 1, dtype=np.uint64)
    srt = oracle.sort_records(recs)
    d = ctx.upload(srt)
    b, c, u = ctx.barcode_counts(d.ptr + 24, n)
    wb, wc, wu = oracle.barcode_counts(srt[1:])
    assert (b.tobytes(), c.tobytes(), u.tobytes()) == (wb.tobytes(), wc.tobytes(), wu.tobytes())


def test_barcode_counts_after_device_sort_and_capacity_error(ctx, oracle, ia):
    import ctypes as C
    n = 300_000
    recs = oracle.generate(SEED, 0, n, 4, 6)  # 4-base barcodes: 256 distinct values, long runs
    np.random.default_rng(1).shuffle(recs)
    d, t = ctx.upload(recs), ctx.alloc(24 * n)
    ctx.sort_records(d, t, n)
    b, c, u = ctx.barcode_counts(d, n)
    wb, wc, wu = oracle.barcode_counts(oracle.sort_records(recs))
    assert (b.tobytes(), c.tobytes(), u.tobytes()) == (wb.tobytes(), wc.tobytes(), wu.tobytes())
    assert int(c.sum()) == n and len(b) <= 256
    # capacity too small: InvalidArg, and the needed size is reported
    nb = C.c_size_t()
    small_b, small_c = ctx.alloc(8), ctx.alloc(8)
    rc = ia.lib.ibu_barcode_counts(ctx._c, C.c_void_p(d.ptr), n, C.c_void_p(small_b.ptr), C.c_void_p(small_c.ptr), None,
                                     1, C.byref(nb), None, None)
    assert rc != 0 and nb.value == len(wb)


def test_sort_with_8_byte_aligned_buffers(ctx, oracle, ia):
    """Buffers that are 8- but not 16-byte aligned take the plain staging path (and the byte copy-back): same result."""
    n = 70_001
    recs = _shuffled(oracle, n, 16, 12)
    d, t = ctx.alloc(n * 24 + 16), ctx.alloc(n * 24 + 16)
    view = ia.DeviceBuffer.wrap(ctx, d.ptr + 8, n * 24)
    view.upload(recs)
    ctx.sort_records(d.ptr + 8, t.ptr + 8, n)
    ctx.synchronize()
    assert view.download().tobytes() == oracle.sort_records(recs).tobytes()


@pytest.mark.parametrize("n", [4_294_000_000, 4_400_000_000])
def test_sort_and_barcode_counts_beyond_2_pow_32_records(ia, n):
    """More than 2^32 records on one GPU (4.4e9 x 24 B = 106 GB + as much scratch: the part has 288 GB): the sort's
    tile positions switch to 64 bits, the run tables of the aggregation too.  Size-independent properties: sorted,
    multiset preserved (count, wrapping sums, XORs), per-barcode counts add up.  
    And just below 2^32: the largest input of the compact-key passes, whose element positions are 32-bit."""
    bc_len, umi_len = 8, 12
    assert abs(n - 2**32) < 2**27
    ctx = ia.Context(0)
    recs = tmp = None
    try:
        recs, tmp = ctx.alloc(n * 24), ctx.alloc(n * 24)
        ctx.generate(0x1B00007, 0, n, bc_len, umi_len, recs)
        before = ctx.reduce(recs, n)
        assert before["count"] == n and before["sum"][2] == (n * (n - 1) // 2) % 2**64
        assert not ctx.is_sorted(recs, n)
        ctx.sort_records(recs, tmp, n)
        assert ctx.is_sorted(recs, n)
        assert ctx.reduce(recs, n) == before
        tmp.free()
        bcs, counts, uniq = ctx.barcode_counts(recs, n)
        assert len(bcs) == 4**bc_len and int(counts.sum()) == n   # 65 536 barcodes, every one drawn ~67 000 times
        assert (np.diff(bcs.astype(np.int64)) > 0).all()
        assert int(uniq.sum()) <= n and int(uniq.min()) > 60_000    # ~67 000 draws from 16.7 M UMIs: almost all distinct
        # first and last records against the closed form of the generator: smallest barcode is 0, largest 4^8 - 1
        head = ia.DeviceBuffer.wrap(ctx, recs.ptr, 24).download(np.uint64)
        tail = ia.DeviceBuffer.wrap(ctx, recs.ptr + (n - 1) * 24, 24).download(np.uint64)
        assert int(head[0]) == 0 and int(tail[0]) == 4**bc_len - 1
    finally:
        for b in (recs, tmp):          # before the context goes: a buffer cannot be freed through a closed context
            if b is not None:
                b.free()
        ctx.close()


def test_context_close_frees_the_buffers_it_still_owns(ia):
    """A DeviceBuffer that outlives its Context must not keep its HBM: close() frees what is still alive (two 120 GB
    buffers in a row through two contexts fit only if the first one went)."""
    big = 120 * 10**9
    c1 = ia.Context(0)
    held = c1.alloc(big)
    c1.close()
    assert held.ptr == 0
    c2 = ia.Context(0)
    try:
        a, b = c2.alloc(big), c2.alloc(big)
        a.free()
        b.free()
    finally:
        c2.close()


# ---- compacted keys: census / plan / records <-> 12-byte elements (the exchange format of the multi-GPU sort) -------
@pytest.mark.parametrize("n", [0, 1, 2, 127, 128, 129, 255, 256, 257, 5000, 300_007])
@pytest.mark.parametrize("lens", [(16, 12), (8, 8), (1, 1), (20, 3)])
@pytest.mark.parametrize("offset", [0, 24])
def test_compact_and_expand_match_the_numpy_statement(ctx, oracle, ia, n, lens, offset):
    """ibu_records_census / ibu_key_plan_init / ibu_records_compact / ibu_records_expand against tests/keyplan_np.py;
    offset 24: the records start at an odd record of a larger buffer (8-byte aligned: one thread per record)."""
    from tests import keyplan_np as kp
    recs = _shuffled(oracle, n + 1, *lens)[: n + 1]
    recs["index"] = np.random.default_rng(n).integers(0, 2**31, n + 1, dtype=np.uint64)
    d = ctx.upload(recs)
    part = recs[offset // 24: offset // 24 + n]
    c = ctx.census(d.ptr + offset, n)
    assert (c["or"], c["and"]) == kp.census_words(part)
    if n > 1:
        assert c["order_drops"] == (oracle.sort_records(part).tobytes() != part.tobytes())
        assert c["index_drops"] == bool((np.diff(part["index"].astype(np.int64)) < 0).any())
    plan, want = ia.key_plan(c["or"], c["and"]), kp.Plan(c["or"], c["and"])
    assert plan.k == want.k and plan.index_bytes == want.index_bytes and [int(b) for b in plan.base] == want.base
    if n == 0 or plan.k > 12:
        return
    e = ctx.alloc(12 * n + 16)
    ctx.compact(plan, d.ptr + offset, n, e)
    ctx.synchronize()
    elems = e.download(count=12 * n).reshape(n, 12)
    assert elems.tobytes() == kp.compact(want, part).tobytes()
    # elements compared as 96-bit little-endian integers order like the records
    as_int = [int.from_bytes(bytes(r), "little") for r in elems[: min(n, 2000)]]
    keys = [(int(r["barcode"]), int(r["umi"]), int(r["index"])) for r in part[: min(n, 2000)]]
    assert sorted(range(len(keys)), key=keys.__getitem__) == sorted(range(len(as_int)), key=lambda i: (as_int[i], i)) or len(set(keys)) < len(keys)
    back = ctx.alloc(24 * n + 48)
    ctx.expand(plan, e, n, back.ptr + offset)
    ctx.synchronize()
    assert ia.DeviceBuffer.wrap(ctx, back.ptr + offset, 24 * n).download().tobytes() == part.tobytes()
    # elements at a 4-byte boundary (the second element buffer inside tmp starts at 12 n)
    e4 = ctx.alloc(12 * n + 16)
    ctx.compact(plan, d.ptr + offset, n, e4.ptr + 4)
    ctx.expand(plan, e4.ptr + 4, n, back.ptr + offset)
    ctx.synchronize()
    assert ia.DeviceBuffer.wrap(ctx, back.ptr + offset, 24 * n).download().tobytes() == part.tobytes()


def test_compact_rejects_more_than_12_varying_bytes(ctx, oracle, ia):
    recs = _shuffled(oracle, 1000, 32, 32)
    d = ctx.upload(recs)
    c = ctx.census(d, 1000)
    plan = ia.key_plan(c["or"], c["and"])
    assert plan.k > 12
    e = ctx.alloc(12 * 1000)
    with pytest.raises(ia.IbuError) as err:
        ctx.compact(plan, d, 1000, e)
    assert err.value.kind == "InvalidArg"
    with pytest.raises(ia.IbuError):
        ctx.expand(plan, e, 1000, d)


# ---- the sort over several shards, one per context, in one call (ibu_sort_records_contexts) ------------------------------
@pytest.mark.parametrize("counts", [[5000], [3000, 7001], [0, 4000, 1], [100_003, 0, 250_000, 77], [1, 1, 1], [0, 0],
                                    [1_000_003, 999_999, 1_300_001]])
@pytest.mark.parametrize("lens,compact", [((16, 12), True), ((16, 12), False), ((32, 32), True)])
def test_sort_records_contexts_is_the_global_order(ia, oracle, counts, lens, compact, capfd):
    """Shards on several contexts (here: all on the box's one GPU, the rehearsal the API allows) come back as the contiguous ranges
    of ONE sorted sequence: their concatenation is the oracle's sort of all records, byte for byte, and the counts add up.
    16/12 keys travel as 12-byte elements (11 varying bytes over all shards), full-range (32,32) keys and contexts told not to
    compact (sort_compact = 0 on the first) as 24-byte records."""
    total = sum(counts)
    recs = oracle.generate(SEED + len(counts), 0, total, *lens)
    rng = np.random.default_rng(total + len(counts))
    rng.shuffle(recs)
    recs["index"] = rng.integers(0, 2**30, total, dtype=np.uint64)
    want = oracle.sort_records(recs).tobytes()
    cap = max(total, len(counts) + 1)
    ctxs = [ia.Context(0) for _ in counts]
    try:
        shards, at = [], 0
        for c, n in zip(ctxs, counts):
            d, t = c.alloc(24 * cap), c.alloc(24 * cap)
            if n:
                d.upload(recs[at:at + n])
            shards.append((d, t, n, cap))
            at += n
        if not compact:
            ctxs[0].set_option("sort_compact", 0)
        capfd.readouterr()
        out = ia.Context.sort_records_contexts(ctxs, shards)
        trace = capfd.readouterr().err
        assert sum(out) == total
        if trace and len(counts) > 1:
            assert f"exchange={12 if compact and lens == (16, 12) and total else 24} bytes per record" in trace, trace   # (no records: every byte "varies")
        got = b"".join(shards[k][0].download(count=24 * out[k]).tobytes() for k in range(len(counts)))
        assert got == want
        if total >= 100_000 and len(counts) > 1:                # the samples balance well-spread keys
            assert max(out) <= 1.3 * total / len(counts), out
    finally:
        for c in ctxs:
            c.close()


def test_sort_records_contexts_refuses_what_does_not_fit(ia, oracle):
    """A shard that would receive more than its capacity: InvalidArg with the numbers, every shard still holding its own records;
    the same context twice, a shard above its capacity, NULL buffers: refused before anything runs."""
    n = 50_000
    a = oracle.generate(SEED, 0, n, 16, 12)
    a["barcode"] |= np.uint64(1) << np.uint64(31)             # shard 0 holds the large keys, shard 1 the small ones:
    b = oracle.generate(SEED + 1, 0, n, 16, 12)
    b["barcode"] &= (np.uint64(1) << np.uint64(31)) - np.uint64(1)   # sorted globally, every record changes sides
    c0, c1 = ia.Context(0), ia.Context(0)
    try:
        cap = n + n // 8                                       # headroom: the cut between two owners falls between two of 256 sampled ranges
        d0, t0, d1, t1 = c0.alloc(24 * cap), c0.alloc(24 * cap), c1.alloc(24 * cap), c1.alloc(24 * cap)
        d0.upload(a)
        d1.upload(b)
        out = ia.Context.sort_records_contexts([c0, c1], [(d0, t0, n, cap), (d1, t1, n, cap)])
        assert sum(out) == 2 * n and abs(out[0] - n) <= n // 50, out
        got = d0.download(count=24 * out[0]).tobytes() + d1.download(count=24 * out[1]).tobytes()
        assert got == oracle.sort_records(b).tobytes() + oracle.sort_records(a).tobytes()
        # uneven: everything belongs to one owner's half, which has no room for it
        small = oracle.generate(SEED + 2, 0, 1000, 16, 12)
        ds, ts = c1.upload(small), c1.alloc(24 * 1000)
        d0.upload(a)
        with pytest.raises(ia.IbuError) as e:
            ia.Context.sort_records_contexts([c0, c1], [(d0, t0, n, n), (ds, ts, 1000, 1000)])
        assert e.value.kind == "InvalidArg" and e.value.b == 1000 and e.value.a > 1000, (e.value.kind, e.value.a, e.value.b)
        # nothing moved between the shards: each still holds ITS records (untouched when the shards are partitioned first,
        # sorted locally on the sort-first path: the order is not part of the contract)
        assert oracle.sort_records(d0.download(ia.REC_DTYPE, count=n)).tobytes() == oracle.sort_records(a).tobytes()
        assert oracle.sort_records(ds.download(ia.REC_DTYPE, count=1000)).tobytes() == oracle.sort_records(small).tobytes()
        for bad in ([(d0, t0, n, n), (d0, t0, n, n)],):
            with pytest.raises(ia.IbuError) as e:
                ia.Context.sort_records_contexts([c0, c0], bad)
            assert e.value.kind == "InvalidArg"
        with pytest.raises(ia.IbuError) as e:
            ia.Context.sort_records_contexts([c0, c1], [(d0, t0, n + 1, n), (d1, t1, n, n)])
        assert e.value.kind == "InvalidArg"
        with pytest.raises(ia.IbuError) as e:
            ia.Context.sort_records_contexts([c0, c1], [(0, 0, n, n), (d1, t1, n, n)])
        assert e.value.kind == "InvalidArg"
    finally:
        c0.close()
        c1.close()


def test_sort_records_contexts_with_sixteen_shards_and_with_equal_records(ia, oracle):
    """Sixteen contexts on one GPU (a host thread each); and records that are all the same: every splitter equals every record,
    so one owner gets them all — fine when it has the room (capacity = everything)."""
    n_each, k = 20_011, 16
    recs = oracle.generate(SEED + 77, 0, n_each * k, 16, 12)
    np.random.default_rng(3).shuffle(recs)
    ctxs = [ia.Context(0) for _ in range(k)]
    try:
        cap = n_each * 2
        shards = []
        for i, c in enumerate(ctxs):
            d, t = c.alloc(24 * cap), c.alloc(24 * cap)
            d.upload(recs[i * n_each:(i + 1) * n_each])
            shards.append((d, t, n_each, cap))
        out = ia.Context.sort_records_contexts(ctxs, shards)
        assert sum(out) == n_each * k and max(out) <= cap
        got = b"".join(shards[i][0].download(count=24 * out[i]).tobytes() for i in range(k))
        assert got == oracle.sort_records(recs).tobytes()
        same = np.repeat(recs[:1], 3000)
        cap2 = 3 * 3000
        sh2 = []
        for c in ctxs[:3]:
            d, t = c.alloc(24 * cap2), c.alloc(24 * cap2)
            d.upload(same)
            sh2.append((d, t, 3000, cap2))
        out = ia.Context.sort_records_contexts(ctxs[:3], sh2)
        assert sorted(out) == [0, 0, 9000]
        j = out.index(9000)
        assert sh2[j][0].download(count=24 * 9000).tobytes() == np.repeat(recs[:1], 9000).tobytes()
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("miss", [False, True])
def test_sort_records_contexts_on_a_sampled_plan(ia, oracle, capfd, miss):
    """Shards of 2^17 records and more: the plan (which key bytes vary) is guessed from three sample ranges of every shard and the
    partition pass takes the exact census on its way.  miss: a UMI byte that is constant in every sample range and varies in between —
    the call notices, says so, and runs again on the exact plan; either way the result is the oracle's order."""
    counts = [200_003, 150_001, 262_144]
    total = sum(counts)
    recs = oracle.generate(SEED + 31, 0, total, 16, 10)         # 10-base UMIs: their fourth key byte is constant (zero)
    rng = np.random.default_rng(77)
    rng.shuffle(recs)
    recs["index"] = rng.integers(0, 2**24, total, dtype=np.uint64)
    if miss:                                                   # rows 40 000 .. 60 000 of shard 0 lie in none of its sample ranges
        recs["umi"][40_000:60_000] |= rng.integers(1, 256, 20_000, dtype=np.uint64) << np.uint64(24)
    want = oracle.sort_records(recs).tobytes()
    cap = total
    ctxs = [ia.Context(0) for _ in counts]
    try:
        shards, at = [], 0
        for c, n in zip(ctxs, counts):
            d, t = c.alloc(24 * cap), c.alloc(24 * cap)
            d.upload(recs[at:at + n])
            shards.append((d, t, n, cap))
            at += n
        capfd.readouterr()
        out = ia.Context.sort_records_contexts(ctxs, shards)
        trace = capfd.readouterr().err
        assert sum(out) == total
        assert b"".join(shards[k][0].download(count=24 * out[k]).tobytes() for k in range(3)) == want
        if trace:
            assert ("the sampled plan missed a varying byte" in trace) == miss, trace
            assert "partition first" in trace
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("IBU_FUZZ_SEEDS_MC", "12"))))
def test_sort_records_contexts_fuzz(ia, oracle, seed):
    """Seeded fuzz of the multi-context sort: 2 .. 6 shards of uneven sizes (empty ones included), keys that compact or do not,
    duplicates, skew — the concatenation of the returned ranges is the oracle's sort of everything."""
    rng = np.random.default_rng(5000 + seed)
    w = int(rng.integers(2, 7))
    counts = [int(rng.choice([0, 1, 7, 1000, 20_000, 150_000])) for _ in range(w)]
    total = sum(counts)
    lens = [(16, 12), (32, 32), (8, 8), (24, 10)][seed % 4]
    recs = oracle.generate(SEED + seed, 0, max(total, 1), *lens)[:total]
    rng.shuffle(recs)
    if seed % 3 == 0 and total:
        recs["index"] = rng.integers(0, 2**30, total, dtype=np.uint64)
    if seed % 5 == 1 and total > 10:                          # many equal records
        recs[: total // 2] = recs[0]
    if seed % 5 == 2 and total > 10:                          # few barcodes
        recs["barcode"] = recs["barcode"][:5][rng.integers(0, 5, total)]
    want = oracle.sort_records(recs).tobytes()
    cap = total + w + 1
    ctxs = [ia.Context(0) for _ in range(w)]
    try:
        shards, at = [], 0
        for c, n in zip(ctxs, counts):
            c.set_option("sort_pull_streams", seed % 2)            # odd seeds: the pulls on per-peer streams, as between distinct GPUs
            if seed % 7 == 3:
                c.set_option("sort_compact", 0)                    # (on the first context this selects the sort-first form)
            d, t = c.alloc(24 * cap), c.alloc(24 * cap)
            if n:
                d.upload(recs[at:at + n])
            shards.append((d, t, n, cap))
            at += n
        out = ia.Context.sort_records_contexts(ctxs, shards)
        assert sum(out) == total
        assert b"".join(shards[k][0].download(count=24 * out[k]).tobytes() for k in range(w)) == want, (seed, counts, lens)
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("k", [33, 64, 128])
def test_sort_records_contexts_with_many_shards_at_a_quarter_of_headroom(ia, oracle, k, capfd):
    """More than 32 shards take the sort-first form (W - 1 splitters from sorted shards: shares within a few percent), so that
    1.25 x the even share is room enough — with the 256 fixed ranges of the partition-first forms an owner of 128 would get one or
    two ranges, 50-100 % off its share."""
    n_each = 3001
    total = n_each * k
    recs = oracle.generate(SEED + k, 0, total, 16, 12)
    np.random.default_rng(k).shuffle(recs)
    cap = n_each * 5 // 4
    ctxs = [ia.Context(0) for _ in range(k)]
    try:
        shards = []
        for i, c in enumerate(ctxs):
            d, t = c.alloc(24 * cap), c.alloc(24 * cap)
            d.upload(recs[i * n_each:(i + 1) * n_each])
            shards.append((d, t, n_each, cap))
        capfd.readouterr()
        out = ia.Context.sort_records_contexts(ctxs, shards)
        trace = capfd.readouterr().err
        assert sum(out) == total and max(out) <= cap
        assert b"".join(shards[i][0].download(count=24 * out[i]).tobytes() for i in range(k)) == oracle.sort_records(recs).tobytes()
        if trace:
            assert "(sort first)" in trace and "partition first" not in trace
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("lens", [(16, 12), (32, 32)])
def test_sort_records_contexts_falls_back_when_the_range_cut_does_not_fit(ia, oracle, lens, capfd):
    """Capacity = the even share + a little: the partition-first cut (whole ranges of ~1/256 of the records, the last owner
    taking what the others left) may not fit where the sort-first cut (sampled quantiles of sorted shards: finer) does — the call
    then falls back instead of failing, and says so.  Which headroom separates the two depends on where the sampled
    boundaries fall for this data, so a few are tried: every call that succeeds returns the oracle's order, a call that fails
    leaves every shard with its own records, and for at least one headroom the fallback is what made the call succeed."""
    k, n_each = 4, 100_000
    total = k * n_each
    recs = oracle.generate(SEED + 404, 0, total, *lens)
    np.random.default_rng(404).shuffle(recs)
    want = oracle.sort_records(recs).tobytes()
    fell_back_and_succeeded = 0
    for extra in (300, 200, 120, 60):
        cap = n_each + extra
        ctxs = [ia.Context(0) for _ in range(k)]
        try:
            shards = []
            for i, c in enumerate(ctxs):
                d, t = c.alloc(24 * cap), c.alloc(24 * cap)
                d.upload(recs[i * n_each:(i + 1) * n_each])
                shards.append((d, t, n_each, cap))
            capfd.readouterr()
            try:
                out = ia.Context.sort_records_contexts(ctxs, shards)
            except ia.IbuError as e:                              # not even the finer cut fits: nothing moved between the shards
                assert e.kind == "InvalidArg" and e.a > cap == e.b
                for i in range(k):
                    mine = shards[i][0].download(ia.REC_DTYPE, count=n_each)
                    assert oracle.sort_records(mine).tobytes() == oracle.sort_records(recs[i * n_each:(i + 1) * n_each]).tobytes()
                continue
            trace = capfd.readouterr().err
            assert sum(out) == total and max(out) <= cap
            assert b"".join(shards[i][0].download(count=24 * out[i]).tobytes() for i in range(k)) == want
            if "falling back to the sort-first form" in trace:
                assert "(sort first)" in trace
                fell_back_and_succeeded += 1
        finally:
            for c in ctxs:
                c.close()
    if os.environ.get("IBU_TRACE_SORT", "") not in ("", "0"):
        assert fell_back_and_succeeded >= 1


def test_sort_records_contexts_orders_exchange_and_sorts_on_the_devices(ia, oracle, capfd):
    """The partition-first forms join their host threads where the host needs every shard's answer and once at the end; the
    exchange and the owners' sorts are chained by stream order and events (no join in between).  Also with direct peer access
    switched off on every context (option "peer_access" = 0: the branch a topology without it takes), and with the pulls on one
    stream per peer (option "sort_pull_streams" = 1: what distinct devices get, so that their links work at the same time)."""
    k, n_each = 8, 150_000
    total = k * n_each
    for lens in ((16, 12), (32, 32)):
        recs = oracle.generate(SEED + 808, 0, total, *lens)
        np.random.default_rng(808).shuffle(recs)
        want = oracle.sort_records(recs).tobytes()
        for peer, own_streams in ((1, 0), (0, 0), (1, 1)):
            cap = n_each * 5 // 4
            ctxs = [ia.Context(0) for _ in range(k)]
            try:
                shards = []
                for i, c in enumerate(ctxs):
                    c.set_option("peer_access", peer)
                    c.set_option("sort_pull_streams", own_streams)   # 1: the per-peer pull streams a multi-GPU run uses, forced on one GPU
                    d, t = c.alloc(24 * cap), c.alloc(24 * cap)
                    d.upload(recs[i * n_each:(i + 1) * n_each])
                    shards.append((d, t, n_each, cap))
                capfd.readouterr()
                out = ia.Context.sort_records_contexts(ctxs, shards)
                trace = capfd.readouterr().err
                assert b"".join(shards[i][0].download(count=24 * out[i]).tobytes() for i in range(k)) == want
                if trace:
                    assert "partition first" in trace and "host joins: samples, range counts (handed over before the partition's scatter), end" in trace, trace
            finally:
                for c in ctxs:
                    c.close()


@pytest.mark.parametrize("kind", ["random", "whitelist"])
def test_sort_records_contexts_wide_keys_share_one_prefix_estimate(ia, oracle, capfd, kind):
    """Wide keys (the 24-byte partition-first form): the owners' sorts take ONE sampled prefix estimate, made on the biggest shard's
    partitioned records for the whole, instead of one each — big enough that the estimate samples (1.05 M records
    hold its tables), with well-spread keys and with barcodes from a short whitelist (long runs of equal barcode: the estimate must
    reach into the UMI or give up, and every owner follows it)."""
    import re
    k, n_each = 4, 1_200_000
    total = k * n_each
    recs = oracle.generate(SEED + 909, 0, total, 32, 32)
    rng = np.random.default_rng(909)
    if kind == "whitelist":
        recs["barcode"] = recs["barcode"][rng.integers(0, 3000, total)]
    rng.shuffle(recs)
    want = oracle.sort_records(recs).tobytes()
    cap = n_each * 5 // 4
    ctxs = [ia.Context(0) for _ in range(k)]
    try:
        shards = []
        for i, c in enumerate(ctxs):
            d, t = c.alloc(24 * cap), c.alloc(24 * cap)
            d.upload(recs[i * n_each:(i + 1) * n_each])
            shards.append((d, t, n_each, cap))
        capfd.readouterr()
        out = ia.Context.sort_records_contexts(ctxs, shards)
        trace = capfd.readouterr().err
        assert sum(out) == total
        assert b"".join(shards[i][0].download(count=24 * out[i]).tobytes() for i in range(k)) == want
        if trace:
            m = re.search(r"one prefix estimate for all owners: (-?\d+)", trace)
            assert m and int(m.group(1)) >= 0, trace
            shared = int(m.group(1))
            owners = [int(x) for x in re.findall(r"path=prefix\+finish prefix_passes=(\d+)", trace)]
            if shared > 0:
                assert owners and all(p == shared for p in owners), trace      # every owner took the shared length
            assert trace.count("sample estimate") <= 1, trace                  # ... and none sampled for itself
    finally:
        for c in ctxs:
            c.close()
