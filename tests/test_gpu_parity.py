"""GPU parity tests: every HIP kernel, called through the C ABI (ibu_amd -> libibu_hip.so),
bit-exact against the CPU oracle on the same seeded inputs.  Integer / byte work: the bar is
equality of every byte, no tolerance.  Run on the MI355X box with `pytest -m gpu`."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x1B00002
# sizes straddling the 128-record wave tile, the 49 152-record reader buffer and the grid
SIZES = [0, 1, 2, 63, 127, 128, 129, 255, 257, 4096, 49_152, 100_000, 1_000_003]


@pytest.fixture(scope="module")
def ia():
    import ibu_amd
    return ibu_amd


@pytest.fixture(scope="module")
def ctx(ia):
    c = ia.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_msb(ia):
    """A context with the hedge bit order (first base most significant): ibu_ctx_set_option(ctx, "base_order", 1)."""
    c = ia.Context(0)
    c.set_option("base_order", 1)
    yield c
    c.close()


@pytest.fixture(params=[0, 1], ids=["lsb_first", "msb_first"])
def ctx_order(request, ctx, ctx_msb):
    """(context, order): every codec parity test below runs under both bit orders against the oracle's same switch."""
    return (ctx, 0) if request.param == 0 else (ctx_msb, 1)


def _up(ctx, a):
    return ctx.upload(np.ascontiguousarray(a))


@pytest.mark.parametrize("n", SIZES)
def test_generate_matches_oracle(ctx, oracle, n):
    d = ctx.alloc(max(n, 1) * 24)
    ctx.generate(SEED, 7, n, 16, 12, d)
    ctx.synchronize()
    assert d.download(count=n * 24).tobytes() == oracle.generate(SEED, 7, n, 16, 12).tobytes()


@pytest.mark.parametrize("n", SIZES)
def test_deserialize_serialize(ctx, oracle, n):
    recs = oracle.generate(SEED, 0, n, 32, 32)
    d_recs = _up(ctx, recs) if n else ctx.alloc(16)
    cols = [ctx.alloc(max(n, 2) * 8) for _ in range(3)]
    ctx.deserialize(d_recs, n, *cols)
    want = oracle.deserialize(recs)
    for c, w in zip(cols, want):
        assert c.download(np.uint64, count=n).tobytes() == w.tobytes()
    d_back = ctx.alloc(max(n, 1) * 24)
    ctx.serialize(*cols, n, d_back)
    assert d_back.download(count=n * 24).tobytes() == recs.tobytes()
    assert oracle.serialize(*want).tobytes() == recs.tobytes()


@pytest.mark.parametrize("n", [0, 1, 127, 128, 129, 100_000, 1_000_003])
@pytest.mark.parametrize("lens", [(16, 12), (32, 32), (1, 1), (15, 11), (20, 8), (4, 28), (31, 3), (12, 16)])
def test_decode_encode(ctx_order, oracle, n, lens):
    ctx, order = ctx_order
    bc_len, umi_len = lens
    recs = oracle.generate(SEED, 0, n, bc_len, umi_len)
    d_recs = _up(ctx, recs) if n else ctx.alloc(16)
    d_bc, d_umi, d_idx = ctx.alloc(max(n, 1) * bc_len), ctx.alloc(max(n, 1) * umi_len), ctx.alloc(max(n, 2) * 8)
    ctx.decode_ascii(d_recs, n, bc_len, umi_len, d_bc, d_umi, d_idx)
    bc, umi, idx = oracle.decode_records(recs, bc_len, umi_len, order)
    assert d_bc.download(count=n * bc_len).tobytes() == bc.tobytes()
    assert d_umi.download(count=n * umi_len).tobytes() == umi.tobytes()
    assert d_idx.download(np.uint64, count=n).tobytes() == idx.tobytes()
    d_back = ctx.alloc(max(n, 1) * 24)
    ctx.encode_ascii(d_bc, d_umi, d_idx, n, bc_len, umi_len, d_back)
    ctx.codec_status()
    want, fb, nb = oracle.encode_records(bc, umi, idx, n, bc_len, umi_len, order=order)
    assert nb == 0
    got = d_back.download(count=n * 24)
    assert got.tobytes() == want.tobytes() == recs.tobytes()


def test_all_length_pairs_small(ctx_order, oracle):
    """Every (bc_len, umi_len) in {1..32}^2 on one-and-a-bit tiles, under both bit orders."""
    ctx, order = ctx_order
    n = 128 + 37
    d_recs = ctx.alloc(n * 24)
    d_bc, d_umi, d_idx, d_back = ctx.alloc(n * 32), ctx.alloc(n * 32), ctx.alloc(n * 8), ctx.alloc(n * 24)
    full = oracle.generate(SEED, 0, n, 32, 32)  # full-range values: bits above 2*len must be ignored (F7/Q13)
    d_recs.upload(full)
    for bc_len in range(1, 33):
        for umi_len in range(1, 33):
            ctx.decode_ascii(d_recs, n, bc_len, umi_len, d_bc, d_umi, d_idx)
            bc, umi, idx = oracle.decode_records(full, bc_len, umi_len, order)
            assert d_bc.download(count=n * bc_len).tobytes() == bc.tobytes(), (bc_len, umi_len)
            assert d_umi.download(count=n * umi_len).tobytes() == umi.tobytes(), (bc_len, umi_len)
            ctx.encode_ascii(d_bc, d_umi, d_idx, n, bc_len, umi_len, d_back)
            ctx.codec_status()
            want, _, nb = oracle.encode_records(bc, umi, idx, n, bc_len, umi_len, order=order)
            assert nb == 0 and d_back.download().tobytes() == want.tobytes(), (bc_len, umi_len)


def test_base_order_option_and_deciding_vector(ia, ctx, ctx_msb, kat):
    """The option is per context, validated, and the ONE vector that would settle the unpinned convention
    (bitnuc::as_2bit(b"ACGT"): 228 under the default order, 27 under the hedge) comes out of the device kernels."""
    d = kat["codec"]["deciding_vector"]
    seq = np.frombuffer(b"ACGT" * 130, dtype=np.uint8)       # one full tile + 2 rows: tiled and tail kernels
    for c, want in ((ctx, d["lsb_first"]), (ctx_msb, d["msb_first"])):
        d_codes = c.alloc(130 * 8)
        c.pack_2bit(_up(c, seq), 130, 4, d_codes)
        c.codec_status()
        assert set(d_codes.download(np.uint64).tolist()) == {want}
        d_ascii = c.alloc(130 * 4)
        c.unpack_2bit(d_codes, 130, 4, d_ascii)
        assert d_ascii.download().tobytes() == seq.tobytes()
    for bad in (2, -1, 7):
        with pytest.raises(ia.IbuError) as e:
            ctx.set_option("base_order", bad)
        assert e.value.kind == "InvalidArg"
    ctx.set_option("base_order", 0)  # explicit default is accepted


def test_msb_first_on_peeled_and_unaligned_bases(ctx_msb, oracle):
    """The hedge order through the peel (shard at record 3) and through the whole-input tail fallback (odd byte bases)."""
    n, bc_len, umi_len = 20_003, 16, 12
    recs = oracle.generate(SEED, 0, n + 3, bc_len, umi_len)
    d_all = _up(ctx_msb, recs)
    bc, umi, idx = oracle.decode_records(recs[3:], bc_len, umi_len, 1)
    for off_bc, off_umi in ((3 * bc_len, 3 * umi_len), (1, 5)):
        d_bc, d_umi, d_idx = ctx_msb.alloc((n + 4) * bc_len), ctx_msb.alloc((n + 4) * umi_len), ctx_msb.alloc((n + 4) * 8)
        ctx_msb.decode_ascii(d_all.ptr + 72, n, bc_len, umi_len, d_bc.ptr + off_bc, d_umi.ptr + off_umi, d_idx.ptr + 24)
        assert d_bc.download(count=n * bc_len, offset=off_bc).tobytes() == bc.tobytes()
        assert d_umi.download(count=n * umi_len, offset=off_umi).tobytes() == umi.tobytes()
        d_back = ctx_msb.alloc((n + 3) * 24)
        ctx_msb.encode_ascii(d_bc.ptr + off_bc, d_umi.ptr + off_umi, d_idx.ptr + 24, n, bc_len, umi_len, d_back.ptr + 72)
        ctx_msb.codec_status()
        assert d_back.download(count=n * 24, offset=72).tobytes() == recs[3:].tobytes()


def test_decode_skips_null_columns(ctx, oracle):
    n = 1000
    recs = oracle.generate(SEED, 0, n, 16, 12)
    d_recs, d_bc = _up(ctx, recs), ctx.alloc(n * 16)
    ctx.decode_ascii(d_recs, n, 16, 12, d_bc, None, None)
    assert d_bc.download().tobytes() == oracle.decode_records(recs, 16, 12)[0].tobytes()


def test_encode_without_index_column(ctx, oracle):
    n = 777
    recs = oracle.generate(SEED, 0, n, 16, 12)
    bc, umi, _ = oracle.decode_records(recs, 16, 12)
    d_back = ctx.alloc(n * 24)
    ctx.encode_ascii(_up(ctx, bc), _up(ctx, umi), None, n, 16, 12, d_back, first_index=10**12)
    ctx.codec_status()
    want, _, _ = oracle.encode_records(bc, umi, None, n, 16, 12, first_index=10**12)
    assert d_back.download().tobytes() == want.tobytes()


@pytest.mark.parametrize("off", [1, 3, 8])
def test_unaligned_bases_take_the_tail_path(ctx, oracle, off):
    """Column / record bases that are not 16-byte aligned (a record slice starting at an odd
    record, ASCII at an arbitrary row) must still be exact."""
    n, bc_len, umi_len = 5000, 15, 12
    recs = oracle.generate(SEED, 0, n + 1, bc_len, umi_len)
    d_all = _up(ctx, recs)
    d_bc, d_umi, d_idx = ctx.alloc(n * bc_len + 64), ctx.alloc(n * umi_len + 64), ctx.alloc(n * 8 + 64)
    ctx.decode_ascii(d_all.ptr + 24, n, bc_len, umi_len, d_bc.ptr + off, d_umi.ptr + off, d_idx.ptr + 8)
    bc, umi, idx = oracle.decode_records(recs[1:], bc_len, umi_len)
    assert d_bc.download(count=n * bc_len, offset=off).tobytes() == bc.tobytes()
    assert d_umi.download(count=n * umi_len, offset=off).tobytes() == umi.tobytes()
    assert d_idx.download(np.uint64, count=n, offset=8).tobytes() == idx.tobytes()
    d_back = ctx.alloc((n + 1) * 24)
    ctx.encode_ascii(d_bc.ptr + off, d_umi.ptr + off, d_idx.ptr + 8, n, bc_len, umi_len, d_back.ptr + 24)
    ctx.codec_status()
    assert d_back.download(count=n * 24, offset=24).tobytes() == recs[1:].tobytes()


@pytest.mark.parametrize("k", [1, 2, 3, 5, 15])
@pytest.mark.parametrize("lens", [(16, 12), (15, 11), (32, 32), (10, 8)])
def test_shard_starting_at_record_k_is_peeled(ia, ctx, oracle, k, lens):
    """A resident shard that starts at record k of a larger buffer (the reference's split gives `per = len / n`,
    which may be odd: mmap.rs:297-307): records, columns and ASCII rows of the shard are 8-byte or less aligned.  The
    launchers peel rows until every array is 16-byte aligned and run the tiled kernels on the rest; every kernel must
    give exactly the bytes of the aligned call, and bad rows keep the CALLER's numbering."""
    bc_len, umi_len = lens
    n = 70_001
    recs = oracle.generate(SEED, 0, n + k, bc_len, umi_len)
    shard = recs[k:]
    d_all = _up(ctx, recs)
    want_bc, want_umi, want_idx = oracle.decode_records(shard, bc_len, umi_len)
    # the output columns belong to the larger buffer too: row k of each column
    d_bc, d_umi, d_idx = ctx.alloc((n + k) * bc_len), ctx.alloc((n + k) * umi_len), ctx.alloc((n + k) * 8)
    pb, pu, pi = d_bc.ptr + k * bc_len, d_umi.ptr + k * umi_len, d_idx.ptr + 8 * k
    ctx.decode_ascii(d_all.ptr + 24 * k, n, bc_len, umi_len, pb, pu, pi)
    assert d_bc.download(count=n * bc_len, offset=k * bc_len).tobytes() == want_bc.tobytes()
    assert d_umi.download(count=n * umi_len, offset=k * umi_len).tobytes() == want_umi.tobytes()
    assert d_idx.download(np.uint64, count=n, offset=8 * k).tobytes() == want_idx.tobytes()
    d_back = ctx.alloc((n + k) * 24)
    ctx.encode_ascii(pb, pu, pi, n, bc_len, umi_len, d_back.ptr + 24 * k)
    ctx.codec_status()
    assert d_back.download(count=n * 24, offset=24 * k).tobytes() == shard.tobytes()
    # index column synthesised from first_index: the peeled rows and the tiled rows number consistently
    ctx.encode_ascii(pb, pu, None, n, bc_len, umi_len, d_back.ptr + 24 * k, first_index=1000)
    ctx.codec_status()
    got = d_back.download(np.uint64, count=3 * n, offset=24 * k).reshape(n, 3)
    assert (got[:, 2] == np.arange(1000, 1000 + n, dtype=np.uint64)).all()
    # reduce, deserialize / serialize, generate, sortedness on the same shard view
    assert ctx.reduce(d_all.ptr + 24 * k, n) == oracle.reduce_records(shard)
    c0, c1, c2 = ctx.alloc((n + k) * 8), ctx.alloc((n + k) * 8), ctx.alloc((n + k) * 8)
    ctx.deserialize(d_all.ptr + 24 * k, n, c0.ptr + 8 * k, c1.ptr + 8 * k, c2.ptr + 8 * k)
    cols = np.frombuffer(shard.tobytes(), dtype=np.uint64).reshape(n, 3)
    for f, c in enumerate((c0, c1, c2)):
        assert c.download(np.uint64, count=n, offset=8 * k).tobytes() == np.ascontiguousarray(cols[:, f]).tobytes()
    ctx.serialize(c0.ptr + 8 * k, c1.ptr + 8 * k, c2.ptr + 8 * k, n, d_back.ptr + 24 * k)
    ctx.synchronize()
    assert d_back.download(count=n * 24, offset=24 * k).tobytes() == shard.tobytes()
    ctx.generate(SEED, k, n, bc_len, umi_len, d_back.ptr + 24 * k)
    ctx.synchronize()
    assert d_back.download(count=n * 24, offset=24 * k).tobytes() == shard.tobytes()
    # single columns
    ctx.unpack_2bit(c0.ptr + 8 * k, n, bc_len, pb)
    ctx.synchronize()
    assert d_bc.download(count=n * bc_len, offset=k * bc_len).tobytes() == oracle.unpack_column(np.ascontiguousarray(cols[:, 0]), bc_len).tobytes()
    ctx.pack_2bit(pb, n, bc_len, c1.ptr + 8 * k)
    ctx.codec_status()
    mask = np.uint64((1 << (2 * bc_len)) - 1) if bc_len < 32 else np.uint64(2**64 - 1)
    assert (c1.download(np.uint64, count=n, offset=8 * k) == (cols[:, 0] & mask)).all()


@pytest.mark.parametrize("seed", range(int(os.environ.get("IBU_FUZZ_SEEDS_CODEC", "40"))))
def test_codec_fuzz(ia, ctx, ctx_msb, oracle, seed):
    """Seeded fuzz of the fused codec and the single-column codec: any (bc_len, umi_len) in {1..32}^2, any n up to ~70 000, a shard
    that starts at record k of a larger buffer (peeled rows), either bit order, full-range field values (bits above 2 len
    ignored), a NULL output column now and then, and in a third of the cases a few offending bytes anywhere — every output
    byte, the first bad row and the count equal the oracle's."""
    rng = np.random.default_rng(9000 + seed)
    order = int(rng.integers(0, 2))
    c = ctx_msb if order else ctx
    bc_len, umi_len = int(rng.integers(1, 33)), int(rng.integers(1, 33))
    n = int(rng.choice([1, 63, 128, 129, 300, 5000, 33_333, 70_001]))
    k = int(rng.choice([0, 0, 1, 2, 3, 7]))
    recs = oracle.generate(SEED + seed, 0, n + k, 32, 32)      # full-range u64 fields
    shard = recs[k:]
    d_all = _up(c, recs)
    want_bc, want_umi, want_idx = oracle.decode_records(shard, bc_len, umi_len, order)
    d_bc, d_umi, d_idx = c.alloc((n + k) * bc_len), c.alloc((n + k) * umi_len), c.alloc((n + k) * 8)
    pb, pu, pi = d_bc.ptr + k * bc_len, d_umi.ptr + k * umi_len, d_idx.ptr + 8 * k
    skip = int(rng.integers(0, 6))                             # 1: no barcode column, 2: no UMI column
    c.decode_ascii(d_all.ptr + 24 * k, n, bc_len, umi_len, None if skip == 1 else pb, None if skip == 2 else pu, pi)
    if skip != 1:
        assert d_bc.download(count=n * bc_len, offset=k * bc_len).tobytes() == want_bc.tobytes(), (seed, bc_len, umi_len, n, k, order)
    if skip != 2:
        assert d_umi.download(count=n * umi_len, offset=k * umi_len).tobytes() == want_umi.tobytes(), (seed, bc_len, umi_len, n, k, order)
    assert d_idx.download(np.uint64, count=n, offset=8 * k).tobytes() == want_idx.tobytes()
    bc, umi = want_bc.copy().reshape(-1), want_umi.copy().reshape(-1)
    if seed % 3 == 0:                                          # offending bytes, anywhere
        for _ in range(int(rng.integers(1, 6))):
            col = bc if rng.integers(0, 2) else umi
            col[int(rng.integers(0, col.size))] = int(rng.choice([0, ord("N"), ord("n"), 0xC1, ord("U"), 255, ord("@")]))
    if rng.integers(0, 2):
        low = rng.integers(0, 2, bc.size).astype(bool)         # lower case packs like upper case
        bc = np.where(low & (bc >= 65) & (bc <= 90), bc | 0x20, bc).astype(np.uint8)
    d_bc.upload(np.concatenate([np.zeros(k * bc_len, np.uint8), bc]))
    d_umi.upload(np.concatenate([np.zeros(k * umi_len, np.uint8), umi]))
    d_back = c.alloc((n + k) * 24)
    c.encode_ascii(pb, pu, pi, n, bc_len, umi_len, d_back.ptr + 24 * k)
    want, fb, nb = oracle.encode_records(bc, umi, want_idx, n, bc_len, umi_len, order=order)
    try:
        c.codec_status()
        got_bad = (None, 0)
    except ia.IbuError as e:
        assert e.kind == "InvalidBase"
        got_bad = (e.first_bad, e.n_bad)
    assert got_bad == ((fb, nb) if nb else (None, 0)), (seed, bc_len, umi_len, n, k, order)
    assert d_back.download(count=n * 24, offset=24 * k).tobytes() == want.tobytes(), (seed, bc_len, umi_len, n, k, order)
    # the single-column kernels on the barcode column
    d_codes = c.alloc((n + k) * 8)
    c.pack_2bit(pb, n, bc_len, d_codes.ptr + 8 * k)
    wc, fb, nb = oracle.pack_column(bc, n, bc_len, order)
    try:
        c.codec_status()
        got_bad = (None, 0)
    except ia.IbuError as e:
        got_bad = (e.first_bad, e.n_bad)
    assert got_bad == ((fb, nb) if nb else (None, 0))
    assert d_codes.download(np.uint64, count=n, offset=8 * k).tobytes() == wc.tobytes()
    c.unpack_2bit(d_codes.ptr + 8 * k, n, bc_len, pb)
    c.synchronize()
    assert d_bc.download(count=n * bc_len, offset=k * bc_len).tobytes() == oracle.unpack_column(wc, bc_len, order).tobytes()
    for b_ in (d_all, d_bc, d_umi, d_idx, d_back, d_codes):
        b_.free()


@pytest.mark.parametrize("k", [1, 3])
def test_bad_rows_keep_the_callers_numbering_when_peeled(ia, ctx, oracle, k):
    bc_len, umi_len, n = 16, 12, 40_000
    recs = oracle.generate(SEED, 0, n, bc_len, umi_len)
    bc, umi, idx = oracle.decode_records(recs, bc_len, umi_len)
    bc, umi = bc.copy(), umi.copy()
    bad_rows = [0, 2, 1000, n - 1]        # a peeled row, a row of the first tile, a tiled row, a row of the rest
    for r in bad_rows:
        bc[r * bc_len + 3] = ord("N")
    want, fb, nb = oracle.encode_records(bc, umi, idx, n, bc_len, umi_len)
    pad_bc = np.concatenate([np.zeros(k * bc_len, np.uint8), bc])
    pad_umi = np.concatenate([np.zeros(k * umi_len, np.uint8), umi])
    pad_idx = np.concatenate([np.zeros(k, np.uint64), idx])
    d_bc, d_umi, d_idx, d_out = _up(ctx, pad_bc), _up(ctx, pad_umi), _up(ctx, pad_idx), ctx.alloc((n + k) * 24)
    for first_bad_expected, rows in ((0, bad_rows), ):
        ctx.encode_ascii(d_bc.ptr + k * bc_len, d_umi.ptr + k * umi_len, d_idx.ptr + 8 * k, n, bc_len, umi_len, d_out.ptr + 24 * k)
        with pytest.raises(ia.IbuError) as ei:
            ctx.codec_status()
        assert (ei.value.first_bad, ei.value.n_bad) == (fb, nb) == (first_bad_expected, len(rows))
        assert d_out.download(count=n * 24, offset=24 * k).tobytes() == want.tobytes()
    # only a tiled row bad: its number must include the peeled rows in front of it
    bc[0 * bc_len + 3] = ord("A"); bc[2 * bc_len + 3] = ord("A")
    want, fb, nb = oracle.encode_records(bc, umi, idx, n, bc_len, umi_len)
    d_bc = _up(ctx, np.concatenate([np.zeros(k * bc_len, np.uint8), bc]))
    ctx.encode_ascii(d_bc.ptr + k * bc_len, d_umi.ptr + k * umi_len, d_idx.ptr + 8 * k, n, bc_len, umi_len, d_out.ptr + 24 * k)
    with pytest.raises(ia.IbuError) as ei:
        ctx.codec_status()
    assert (ei.value.first_bad, ei.value.n_bad) == (fb, nb) == (1000, 2)
    d_codes = ctx.alloc((n + k) * 8)
    ctx.pack_2bit(d_bc.ptr + k * bc_len, n, bc_len, d_codes.ptr + 8 * k)
    with pytest.raises(ia.IbuError) as ei:
        ctx.codec_status()
    assert (ei.value.first_bad, ei.value.n_bad) == (1000, 2)


def test_odd_record_shard_takes_the_tiled_kernels(ia, ctx, oracle, capfd, monkeypatch):
    """The peel: decode / encode / reduce of a shard that starts at record 1 or 3 of a larger buffer
    (8-byte aligned base) must run through the TILED kernels for all but a handful of rows — it used to fall back to the
    one-thread-per-record kernel for every record (~10x slower).  Asserted on the path taken (option trace_rows: rows per
    kernel of every launch), not on a clock: the wall-clock form of this check lives in tests/perf/peel_rate.py."""
    n, bc_len, umi_len = 1_000_003, 16, 12
    recs, back = ctx.alloc((n + 4) * 24), ctx.alloc((n + 4) * 24)
    bc, umi, idx = ctx.alloc((n + 4) * bc_len), ctx.alloc((n + 4) * umi_len), ctx.alloc((n + 4) * 8)
    ctx.generate(SEED, 0, n + 4, bc_len, umi_len, recs)
    host = recs.download(ia.REC_DTYPE)
    ctx.set_option("trace_rows", 1)       # a context option since round 4 (it was an environment lookup per launch)
    for k in (0, 1, 3):
        capfd.readouterr()
        ctx.decode_ascii(recs.ptr + 24 * k, n, bc_len, umi_len, bc.ptr + k * bc_len, umi.ptr + k * umi_len, idx.ptr + 8 * k)
        ctx.encode_ascii(bc.ptr + k * bc_len, umi.ptr + k * umi_len, idx.ptr + 8 * k, n, bc_len, umi_len, back.ptr + 24 * k)
        got = ctx.reduce(recs.ptr + 24 * k, n)
        ctx.synchronize()
        lines = [ln for ln in capfd.readouterr().err.splitlines() if ln.startswith("ibu rows:")]
        assert len(lines) >= 3, lines
        for ln in lines:
            f = dict(kv.split("=") for kv in ln.split()[2:])
            assert int(f["n"]) == n and int(f["head"]) < 16 and int(f["rest"]) < int(f["tile"]), (k, ln)
            assert int(f["tiled"]) >= n - 16 - int(f["tile"]), (k, ln)
        ctx.codec_status()
        want = host[k:k + n]
        assert got == oracle.reduce_records(want)
        assert back.download(count=n * 24, offset=24 * k).tobytes() == want.tobytes()
    ctx.set_option("trace_rows", 0)


@pytest.mark.parametrize("length", [1, 3, 4, 8, 12, 13, 16, 24, 31, 32])
@pytest.mark.parametrize("n", [0, 1, 129, 65_537])
def test_pack_unpack_columns(ctx_order, oracle, n, length):
    ctx, order = ctx_order
    rng = np.random.default_rng(length * 1000 + n)
    codes = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    d_codes = _up(ctx, codes) if n else ctx.alloc(16)
    d_ascii = ctx.alloc(max(n, 1) * length)
    ctx.unpack_2bit(d_codes, n, length, d_ascii)
    want = oracle.unpack_column(codes, length, order)
    got = d_ascii.download(count=n * length)
    assert got.tobytes() == want.tobytes()
    d_back = ctx.alloc(max(n, 2) * 8)
    ctx.pack_2bit(d_ascii, n, length, d_back)
    ctx.codec_status()
    wcodes, _, nb = oracle.pack_column(want, n, length, order)
    assert nb == 0 and d_back.download(np.uint64, count=n).tobytes() == wcodes.tobytes()
    # lower-case input packs to the same codes
    d_low = _up(ctx, np.frombuffer(want.tobytes().lower(), dtype=np.uint8)) if n else ctx.alloc(16)
    ctx.pack_2bit(d_low, n, length, d_back)
    ctx.codec_status()
    assert d_back.download(np.uint64, count=n).tobytes() == wcodes.tobytes()


@pytest.mark.parametrize("n", [200, 70_001])
@pytest.mark.parametrize("lens", [(16, 12), (15, 7)])
def test_invalid_bases_reported(ia, ctx, oracle, n, lens):
    bc_len, umi_len = lens
    recs = oracle.generate(SEED, 0, n, bc_len, umi_len)
    bc, umi, idx = oracle.decode_records(recs, bc_len, umi_len)
    bc, umi = bc.copy(), umi.copy()
    bad_rows = sorted({5, 130, n // 2, n - 1})
    bc[bad_rows[0] * bc_len] = ord("N")
    umi[bad_rows[1] * umi_len + umi_len - 1] = 0
    bc[bad_rows[2] * bc_len + bc_len - 1] = ord("U")
    umi[bad_rows[2] * umi_len] = ord(" ")  # same row bad in both fields: counted once
    umi[bad_rows[3] * umi_len] = 0xC1      # high-bit byte that upper-cases to 'A'-like pattern
    d_back = ctx.alloc(n * 24)
    ctx.encode_ascii(_up(ctx, bc), _up(ctx, umi), _up(ctx, idx), n, bc_len, umi_len, d_back)
    want, fb, nb = oracle.encode_records(bc, umi, idx, n, bc_len, umi_len)
    with pytest.raises(ia.IbuError) as ei:
        ctx.codec_status()
    assert ei.value.kind == "InvalidBase"
    assert (ei.value.first_bad, ei.value.n_bad) == (fb, nb) == (bad_rows[0], len(bad_rows))
    assert d_back.download().tobytes() == want.tobytes()  # offending fields are written as zero
    ctx.codec_status()  # slot re-armed
    d_codes = ctx.alloc(n * 8)
    ctx.pack_2bit(_up(ctx, bc), n, bc_len, d_codes)
    wc, fb, nb = oracle.pack_column(bc, n, bc_len)
    with pytest.raises(ia.IbuError) as ei:
        ctx.codec_status()
    assert (ei.value.first_bad, ei.value.n_bad) == (fb, nb)
    assert d_codes.download(np.uint64).tobytes() == wc.tobytes()


@pytest.mark.parametrize("lens", [(5, 5), (13, 16), (16, 13), (31, 3), (17, 19), (9, 32)])
@pytest.mark.parametrize("n", [131, 40_003])
def test_invalid_bases_in_runtime_length_fields(ia, ctx_order, oracle, n, lens):
    """The runtime-length kernels pack a field chunk by chunk (its code stream) and attribute an offending byte to its ROW on a
    slow path taken only by the tiles that hold one: bad rows in the first, a middle and the last tile, at the first and last
    base of a row, in a row that straddles two 16-byte chunks, in both fields of one row, in neighbouring rows — first bad row,
    count and the zeroed fields equal the oracle's, under both bit orders, for lengths on either side of a specialised one."""
    ctx, order = ctx_order
    bc_len, umi_len = lens
    recs = oracle.generate(SEED + 9, 0, n, bc_len, umi_len)
    bc, umi, idx = oracle.decode_records(recs, bc_len, umi_len, order)
    bc, umi = bc.copy(), umi.copy()
    rows = sorted({1, 2, 127, 128, n // 2, n - 2, n - 1})
    bc[rows[0] * bc_len] = ord("N")                            # first base
    bc[rows[1] * bc_len + bc_len - 1] = ord("n")               # last base, the row after
    umi[rows[2] * umi_len + umi_len // 2] = 0x00               # last row of the first tile
    umi[rows[3] * umi_len] = 0xFF                              # first row of the second tile
    bc[rows[4] * bc_len + bc_len // 2] = ord("U")
    umi[rows[4] * umi_len + umi_len - 1] = ord("-")            # the same row bad in both fields: counted once
    bc[rows[5] * bc_len] = ord("g") + 1                        # 'h'
    umi[rows[6] * umi_len + umi_len - 1] = ord("T") ^ 0x80     # the very last byte of the column
    d_back = ctx.alloc(n * 24)
    ctx.encode_ascii(_up(ctx, bc), _up(ctx, umi), _up(ctx, idx), n, bc_len, umi_len, d_back)
    want, fb, nb = oracle.encode_records(bc, umi, idx, n, bc_len, umi_len, order=order)
    with pytest.raises(ia.IbuError) as ei:
        ctx.codec_status()
    assert ei.value.kind == "InvalidBase"
    assert (ei.value.first_bad, ei.value.n_bad) == (fb, nb) == (rows[0], len(rows))
    assert d_back.download().tobytes() == want.tobytes()
    for col, length in ((bc, bc_len), (umi, umi_len)):          # the single-column kernel on the same columns
        d_codes = ctx.alloc(n * 8)
        ctx.pack_2bit(_up(ctx, col), n, length, d_codes)
        wc, fb, nb = oracle.pack_column(col, n, length, order)
        with pytest.raises(ia.IbuError) as ei:
            ctx.codec_status()
        assert (ei.value.first_bad, ei.value.n_bad) == (fb, nb)
        assert d_codes.download(np.uint64).tobytes() == wc.tobytes()


def test_every_byte_value_in_a_runtime_length_row(ctx, oracle):
    """All 256 byte values in every position of a 7-base row (a length without a specialisation: rows straddle chunks)."""
    base = b"ACGTACG"
    rows = []
    for pos in range(7):
        for b in range(256):
            r = bytearray(base)
            r[pos] = b
            rows.append(bytes(r))
    a = np.frombuffer(b"".join(rows), dtype=np.uint8)
    n = len(rows)
    d_codes = ctx.alloc(n * 8)
    ctx.pack_2bit(_up(ctx, a), n, 7, d_codes)
    want, fb, nb = oracle.pack_column(a, n, 7)
    fbd, nbd = None, 0
    try:
        ctx.codec_status()
    except Exception as e:
        fbd, nbd = e.first_bad, e.n_bad
    assert (fbd, nbd) == (fb, nb) and nb == 7 * (256 - 8)
    assert d_codes.download(np.uint64).tobytes() == want.tobytes()


def test_every_byte_value_classified(ctx, oracle):
    """All 256 byte values in every position of a 4-base row."""
    rows = []
    for pos in range(4):
        for b in range(256):
            r = bytearray(b"ACGT")
            r[pos] = b
            rows.append(bytes(r))
    a = np.frombuffer(b"".join(rows), dtype=np.uint8)
    n = len(rows)
    d_codes = ctx.alloc(n * 8)
    ctx.pack_2bit(_up(ctx, a), n, 4, d_codes)
    want, fb, nb = oracle.pack_column(a, n, 4)
    fbd, nbd = None, 0
    try:
        ctx.codec_status()
    except Exception as e:
        fbd, nbd = e.first_bad, e.n_bad
    assert (fbd, nbd) == (fb, nb) and nb == 4 * (256 - 8)
    assert d_codes.download(np.uint64).tobytes() == want.tobytes()


@pytest.mark.parametrize("n", SIZES + [3_000_001])
def test_reduce(ctx, oracle, n):
    recs = oracle.generate(SEED, 0, n, 32, 32)  # full-range u64: sums must wrap mod 2^64
    d = _up(ctx, recs) if n else ctx.alloc(16)
    got = ctx.reduce(d, n)
    assert got == oracle.reduce_records(recs)


def test_reduce_accumulates_across_calls_and_shards(ctx, ia, oracle):
    """Shard partials combine to the whole (the multi-GPU reduction is a sum / xor of these)."""
    n = 1_000_000
    recs = oracle.generate(SEED, 0, n, 16, 12)
    d = _up(ctx, recs)
    whole = ctx.reduce(d, n)
    ctx.reduce(d, 0, reset=True, fetch=False)
    for i in range(8):
        s, e = ia.shard_range(n, 8, i)
        ctx.reduce(d.ptr + 24 * s, e - s, reset=False, fetch=False)
    assert ctx.reduce_fetch() == whole == oracle.reduce_records(recs)


def test_reference_example_values(ctx, oracle, kat):
    """examples/roundtrip.rs records at N = 1e6 (BASELINE config 1) through the device reduce."""
    k = kat["roundtrip_1e6"]
    i = np.arange(k["n"], dtype=np.uint64)
    recs = oracle.records_array(np.stack([i % 1_000_000, (i * 31) % 1_000_000, i], axis=1))
    got = ctx.reduce(_up(ctx, recs), k["n"])
    assert got["count"] == k["n"] and got["sum"] == k["sums"] and got["xor"] == k["xors"]


@pytest.mark.parametrize("nbytes", [0, 1, 15, 16, 17, 4096, 1_000_003, 48_000_016])
@pytest.mark.parametrize("off", [(0, 0), (8, 0), (0, 3), (16, 32)])
def test_device_copy(ia, ctx, nbytes, off):
    """ibu_device_copy: the streaming memcpy (writer.rs:335-347) on device; unaligned ranges take the byte kernel."""
    rng = np.random.default_rng(nbytes + 7 * off[0] + off[1])
    host = rng.integers(0, 256, nbytes + 64, dtype=np.uint8)
    src, dst = ctx.upload(host), ctx.upload(np.zeros(nbytes + 64, dtype=np.uint8))
    ctx.copy(dst.ptr + off[0], src.ptr + off[1], nbytes)
    ctx.synchronize()
    got = dst.download()
    assert got[off[0]:off[0] + nbytes].tobytes() == host[off[1]:off[1] + nbytes].tobytes()
    assert not got[:off[0]].any() and not got[off[0] + nbytes:].any()  # nothing outside the range is touched
    if nbytes > 16:
        with pytest.raises(ia.IbuError) as e:
            ctx.copy(src.ptr + 8, src.ptr, nbytes)
        assert e.value.kind == "InvalidArg"


def test_is_sorted(ctx, oracle):
    n = 100_000
    recs = oracle.generate(SEED, 0, n, 8, 8)
    assert ctx.is_sorted(_up(ctx, recs), n) == oracle.is_sorted(recs)
    s = oracle.sort_records(recs)
    assert ctx.is_sorted(_up(ctx, s), n) is True
    s2 = s.copy()
    s2[[n - 2, n - 1]] = s2[[n - 1, n - 2]]
    assert ctx.is_sorted(_up(ctx, s2), n) == oracle.is_sorted(s2)


def test_argument_errors(ia, ctx):
    d = ctx.alloc(4096)
    for bc, umi, kind in [(0, 12, "InvalidBarcodeLength"), (33, 12, "InvalidBarcodeLength"),
                          (16, 0, "InvalidUmiLength"), (16, 33, "InvalidUmiLength")]:
        with pytest.raises(ia.IbuError) as ei:
            ctx.decode_ascii(d, 10, bc, umi, d, d, d)
        assert ei.value.kind == kind
    with pytest.raises(ia.IbuError) as ei:
        ctx.unpack_2bit(d, 10, 33, d)
    assert ei.value.kind == "SeqLen"
    with pytest.raises(ia.IbuError) as ei:
        ctx.deserialize(d.ptr + 4, 10, d, d, d)
    assert ei.value.kind == "InvalidArg"
    with pytest.raises(ia.IbuError) as ei:
        ia.Context(99)
    assert ei.value.kind == "NoDevice"


def test_large_roundtrip_properties(ctx):
    """Size-independent properties at a size the oracle would be slow on (5e7 records, 16/12):
    encode(decode(x)) == x byte for byte (checked on device via reduce of both + is-equal of
    checksums over 8 shards), and count/sums of generated index column are closed-form."""
    n, bc_len, umi_len = 50_000_000, 16, 12
    d_recs, d_back = ctx.alloc(n * 24), ctx.alloc(n * 24)
    d_bc, d_umi, d_idx = ctx.alloc(n * bc_len), ctx.alloc(n * umi_len), ctx.alloc(n * 8)
    ctx.generate(SEED, 0, n, bc_len, umi_len, d_recs)
    ctx.decode_ascii(d_recs, n, bc_len, umi_len, d_bc, d_umi, d_idx)
    ctx.encode_ascii(d_bc, d_umi, d_idx, n, bc_len, umi_len, d_back)
    ctx.codec_status()
    a, b = ctx.reduce(d_recs, n), ctx.reduce(d_back, n)
    assert a == b and a["count"] == n
    assert a["sum"][2] == (n * (n - 1) // 2) % 2**64
    # spot-check raw bytes of the first and last MiB
    for off in (0, n * 24 - (1 << 20)):
        assert d_recs.download(count=1 << 20, offset=off).tobytes() == d_back.download(count=1 << 20, offset=off).tobytes()


def test_all_rows_invalid_is_counted_exactly(ia, ctx):
    """Every row bad (the per-wave tally is flushed once per wave: no atomics in the loop, same cost as valid input)."""
    n, bc_len, umi_len = 1_000_003, 16, 12
    bc = ctx.upload(np.full(n * bc_len, ord("N"), dtype=np.uint8))
    umi = ctx.upload(np.full(n * umi_len, ord("A"), dtype=np.uint8))
    out = ctx.alloc(n * 24)
    ctx.encode_ascii(bc, umi, None, n, bc_len, umi_len, out)
    with pytest.raises(ia.IbuError) as e:
        ctx.codec_status()
    assert e.value.kind == "InvalidBase" and e.value.first_bad == 0 and e.value.n_bad == n
    recs = out.download(np.uint64, count=3 * n).reshape(n, 3)
    assert not recs[:, 0].any() and not recs[:, 1].any() and (recs[:, 2] == np.arange(n, dtype=np.uint64)).all()
    ctx.codec_status()  # re-armed


def test_runtime_length_decode_is_a_prefix_of_the_32_base_decode_1e9(ia, ctx):
    """A size-independent property that ties the runtime-length kernels to a specialised one at BASELINE's full size: with
    the first base in the least significant bits, the L-base decode of a code word is the first L bases of its 32-base
    decode, whatever the bits above 2 L hold (F7).  1e9 full-range records through ibu_k_decode<32,32> and through the
    runtime-length kernel at (31,17), (5,29): pieces at the start, in the middle and at the end compared on the host."""
    n = 1_000_000_000
    recs = ctx.alloc(n * 24)
    ctx.generate(0x1B00002, 0, n, 32, 32, recs)                # full-range u64 fields
    bc32, umi32 = ctx.alloc(n * 32), ctx.alloc(n * 32)
    ctx.decode_ascii(recs, n, 32, 32, bc32, umi32, None)
    piece = 200_000
    spots = [0, n // 2 - 77, n - piece]

    def rows(buf, length, lo):
        return ia.DeviceBuffer.wrap(ctx, buf.ptr + lo * length, piece * length).download().reshape(piece, length)

    ref = [(rows(bc32, 32, lo), rows(umi32, 32, lo)) for lo in spots]
    for bc_len, umi_len in ((31, 17), (5, 29)):
        bc, umi = ctx.alloc(n * bc_len), ctx.alloc(n * umi_len)
        ctx.decode_ascii(recs, n, bc_len, umi_len, bc, umi, None)
        for lo, (rb, ru) in zip(spots, ref):
            assert np.array_equal(rows(bc, bc_len, lo), rb[:, :bc_len]), (bc_len, lo)
            assert np.array_equal(rows(umi, umi_len, lo), ru[:, :umi_len]), (umi_len, lo)
        bc.free()
        umi.free()
    for b_ in (recs, bc32, umi32):
        b_.free()


@pytest.mark.parametrize("lens", [(16, 12), (32, 32), (31, 31), (17, 19)])
def test_full_size_roundtrip_properties_1e9(ia, ctx, lens):
    """BASELINE.json's full size (1e9 records; configs[2] and configs[3] shapes) through size-independent properties:
    encode(decode(x)) == x, count and closed-form sums, decoded alphabet, index column, and sort -> sorted with the
    multiset preserved.  (Bit-exactness against the oracle is established at the sizes the oracle finishes in seconds.)"""
    n = 1_000_000_000
    bc_len, umi_len = lens
    seed, first = 0x1B00003, 12345
    recs, back = ctx.alloc(n * 24), ctx.alloc(n * 24)
    bc, umi, idx = ctx.alloc(n * bc_len), ctx.alloc(n * umi_len), ctx.alloc(n * 8)
    ctx.generate(seed, first, n, bc_len, umi_len, recs)
    red = ctx.reduce(recs, n)
    assert red["count"] == n
    assert red["sum"][2] == (n * first + n * (n - 1) // 2) % 2**64  # index = first + i
    ctx.decode_ascii(recs, n, bc_len, umi_len, bc, umi, idx)
    ctx.encode_ascii(bc, umi, idx, n, bc_len, umi_len, back)
    ctx.codec_status()  # every decoded byte was a valid base
    assert ctx.reduce(back, n) == red
    # byte-for-byte equality of the round trip over the WHOLE buffers (24 GB each): on the device, and — the device
    # comparison is itself under test — on the host for every 10th piece of 20 M records
    assert ctx.first_mismatch(recs, back, n) == n
    step = 20_000_000
    for lo in range(0, n, 10 * step):
        k = min(step, n - lo)
        a = ia.DeviceBuffer.wrap(ctx, recs.ptr + lo * 24, k * 24).download()
        b = ia.DeviceBuffer.wrap(ctx, back.ptr + lo * 24, k * 24).download()
        assert np.array_equal(a, b), f"round trip differs in records [{lo}, {lo + k})"
    head = bc.download(count=bc_len * 1000)
    assert set(head.tolist()) <= set(b"ACGT")
    tail_idx = ia.DeviceBuffer.wrap(ctx, idx.ptr + (n - 5) * 8, 40).download(np.uint64)
    assert tail_idx.tolist() == [first + n - 5 + k for k in range(5)]
    for b_ in (bc, umi, idx):
        b_.free()
    # sort at full size: sortedness + multiset (count, wrapping sums, XORs) preserved
    assert not ctx.is_sorted(recs, n)
    ctx.sort_records(recs, back, n)
    assert ctx.is_sorted(recs, n)
    assert ctx.reduce(recs, n) == red


def test_full_size_sort_and_decode_against_the_oracle_on_key_ranges_1e9(ia, ctx, oracle):
    """ONE oracle comparison at BASELINE.json's full size (the 1e9 results had only been compared with
    themselves).  The device generates and sorts 1e9 16/12 records; the oracle generates the same stream on the host in chunks of
    1e7 and keeps the records whose barcode falls into 16 narrow ranges (about 1e5 records each), sorts each kept set with its
    qsort — and the device's sorted output between lower_bound((lo, 0, 0)) and lower_bound((hi, 0, 0)) must be those bytes
    (order: record.rs:58).  The decode kernel then runs over all 1e9 sorted records and its columns over the same 16 slices
    must be the oracle's decode of the kept records (record.rs:19-27)."""
    n, bc_len, umi_len, seed = 1_000_000_000, 16, 12, 0x1B00003
    k_ranges, width = 16, 429_497                                # 2^32 barcodes x 1e-4: ~1e5 of the 1e9 records per range
    rng = np.random.default_rng(20251005)
    los = np.sort(rng.integers(0, 2**32 - width, k_ranges, dtype=np.uint64))
    assert np.all(np.diff(los) > width)
    edges = np.stack([los, los + np.uint64(width)], axis=1).reshape(-1)   # lo0, hi0, lo1, hi1, ...
    recs, tmp = ctx.alloc(n * 24), ctx.alloc(n * 24)
    ctx.generate(seed, 0, n, bc_len, umi_len, recs)
    ctx.sort_records(recs, tmp, n)                               # (queued; the host filters meanwhile)
    kept = [[] for _ in range(k_ranges)]
    chunk = 10_000_000
    for lo in range(0, n, chunk):
        part = oracle.generate(seed, lo, min(chunk, n - lo), bc_len, umi_len)
        where = np.searchsorted(edges, part["barcode"], side="right")   # odd: inside a range
        inside = np.flatnonzero(where & 1)
        for r in np.unique(where[inside] >> 1):
            kept[int(r)].append(part[inside[(where[inside] >> 1) == r]])
        del part
    ctx.synchronize()
    keys = np.zeros(2 * k_ranges, dtype=ia.REC_DTYPE)
    keys["barcode"] = edges
    d_keys, d_pos = ctx.upload(keys), ctx.alloc(8 * 2 * k_ranges)
    ctx.lower_bound(recs, n, d_keys, 2 * k_ranges, d_pos)
    ctx.synchronize()
    pos = d_pos.download(np.uint64).reshape(-1, 2)
    bc, umi, idx = ctx.alloc(n * bc_len), ctx.alloc(n * umi_len), ctx.alloc(n * 8)
    ctx.decode_ascii(recs, n, bc_len, umi_len, bc, umi, idx)     # K2 over all 1e9 sorted records
    ctx.synchronize()
    total = 0
    for r in range(k_ranges):
        want = oracle.sort_records(np.concatenate(kept[r]))
        p0, p1 = int(pos[r][0]), int(pos[r][1])
        assert p1 - p0 == len(want) and len(want) > 50_000, (r, p0, p1, len(want))
        got = ia.DeviceBuffer.wrap(ctx, recs.ptr + 24 * p0, 24 * (p1 - p0)).download()
        assert got.tobytes() == want.tobytes(), f"sorted output differs from the oracle in key range {r}"
        wbc, wumi, widx = oracle.decode_records(want, bc_len, umi_len)
        assert ia.DeviceBuffer.wrap(ctx, bc.ptr + bc_len * p0, bc_len * (p1 - p0)).download().tobytes() == wbc.tobytes(), r
        assert ia.DeviceBuffer.wrap(ctx, umi.ptr + umi_len * p0, umi_len * (p1 - p0)).download().tobytes() == wumi.tobytes(), r
        assert ia.DeviceBuffer.wrap(ctx, idx.ptr + 8 * p0, 8 * (p1 - p0)).download().tobytes() == widx.tobytes(), r
        total += len(want)
    assert total > 1_000_000
    for b_ in (recs, tmp, bc, umi, idx, d_keys, d_pos):
        b_.free()


_GRAPH_SCRIPT = r"""
import sys
import numpy as np
import torch                       # before the library: both must bind the HIP runtime torch ships (as bench.py does)
sys.path.insert(0, sys.argv[1])
import ibu_amd as ia
from oracle import oracle
SEED = 0x1B00002
n, bc_len, umi_len = 100_003, 16, 12
c = ia.Context(0)
side = torch.cuda.Stream()
st = side.cuda_stream
big = c.alloc(24 * (n + 1))
recs = ia.DeviceBuffer.wrap(c, big.ptr + 24, 24 * n)         # 8- but not 16-byte aligned: peeled rows
bc, umi, idx, back = c.alloc(n * bc_len), c.alloc(n * umi_len), c.alloc(n * 8), c.alloc(n * 24)
c0, c1, c2, codes, asc, copy = c.alloc(n * 8), c.alloc(n * 8), c.alloc(n * 8), c.alloc(n * 8), c.alloc(n * 12), c.alloc(n * 24)

def launches():
    c.decode_ascii(recs, n, bc_len, umi_len, bc, umi, idx, stream=st)
    c.encode_ascii(bc, umi, idx, n, bc_len, umi_len, back, stream=st)
    c.deserialize(recs, n, c0, c1, c2, stream=st)
    c.serialize(c0, c1, c2, n, copy, stream=st)
    c.unpack_2bit(c1, n, 12, asc, stream=st)
    c.pack_2bit(asc, n, 12, codes, stream=st)
    c.reduce(recs, n, stream=st, reset=True, fetch=False)

c.generate(SEED, 0, n, bc_len, umi_len, recs, stream=st)
launches()                                                   # warm-up outside the capture: module load, occupancy queries
c.synchronize(st)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    launches()
for seed in (SEED + 1, SEED + 2):
    c.generate(seed, 7, n, bc_len, umi_len, recs, stream=st)  # new input, same buffers; not part of the graph
    c.synchronize(st)
    g.replay()
    torch.cuda.synchronize()
    want = oracle.generate(seed, 7, n, bc_len, umi_len)
    wbc, wumi, widx = oracle.decode_records(want, bc_len, umi_len)
    assert bc.download().tobytes() == wbc.tobytes() and umi.download().tobytes() == wumi.tobytes()
    assert idx.download(np.uint64).tobytes() == widx.tobytes()
    assert back.download().tobytes() == want.tobytes() == copy.download().tobytes()
    assert codes.download(np.uint64).tobytes() == want["umi"].tobytes()
    assert c.reduce_fetch(st) == oracle.reduce_records(want)
    c.codec_status(st)
c.close()
print("GRAPH_OK")
"""


def test_launch_functions_can_be_captured_into_a_hip_graph():
    """include/ibu_hip.h: "Launch functions are asynchronous, allocate nothing and never synchronise (graph-capturable)".  The
    hot path's launches (K2 decode, K3 encode, K1 / K1', column unpack / pack, K4 reduce) are captured ONCE into a hipGraph
    (through torch's graph capture on a side stream) and replayed over new input: every replay's outputs are the oracle's for the
    records then in the buffer — head-peel and tail kernels included (n is not a multiple of the tile and the record buffer starts
    at an odd record).  In a process of its own: torch must bind the HIP runtime before the library does."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _GRAPH_SCRIPT, root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "GRAPH_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_generate_into_an_8_byte_aligned_buffer(ia, ctx, oracle):
    """Not 16-B aligned: every record takes the one-thread-per-record kernel; same bytes."""
    n = 10_007
    d = ctx.alloc(n * 24 + 16)
    ctx.generate(SEED, 3, n, 20, 7, d.ptr + 8)
    ctx.synchronize()
    got = ia.DeviceBuffer.wrap(ctx, d.ptr + 8, n * 24).download()
    assert got.tobytes() == oracle.generate(SEED, 3, n, 20, 7).tobytes()


def test_two_contexts_on_two_host_threads(ia, oracle):
    """'One context per host thread; the library is re-entrant across contexts' (include/ibu_hip.h): two threads, each
    with its own context and stream, run the whole kernel set concurrently; both must match the oracle, and an error
    raised in one thread must not leak into the other's error slot."""
    import threading

    results, errors = {}, []

    def worker(tag, seed, lens, n):
        try:
            c = ia.Context(0)
            want = oracle.generate(seed, 0, n, *lens)
            for _ in range(5):
                d = c.alloc(n * 24)
                c.generate(seed, 0, n, *lens, d)
                bc, umi, idx, back = c.alloc(n * lens[0]), c.alloc(n * lens[1]), c.alloc(n * 8), c.alloc(n * 24)
                c.decode_ascii(d, n, lens[0], lens[1], bc, umi, idx)
                c.encode_ascii(bc, umi, idx, n, lens[0], lens[1], back)
                c.codec_status()
                assert back.download().tobytes() == want.tobytes()
                assert c.reduce(d, n) == oracle.reduce_records(want)
                t = c.alloc(n * 24)
                c.sort_records(back, t, n)
                assert c.is_sorted(back, n)
                if tag == "a":  # provoke an error in this thread only
                    with pytest.raises(ia.IbuError) as e:
                        c.decode_ascii(d, n, 0, 12, bc, umi, idx)
                    assert e.value.kind == "InvalidBarcodeLength"
            results[tag] = True
            c.close()
        except Exception as ex:  # noqa: BLE001 - reported to the main thread
            errors.append((tag, repr(ex)))

    ts = [threading.Thread(target=worker, args=("a", 11, (16, 12), 300_007)),
          threading.Thread(target=worker, args=("b", 12, (32, 32), 200_003))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert results == {"a": True, "b": True}


def test_hip_runtime_errors_surface_as_status(ia, ctx):
    """A failing HIP call (here: an impossible allocation) comes back as IBU_ERR_HIP with the hipError_t in the detail —
    no exception crosses the ABI, and the context keeps working."""
    with pytest.raises(ia.IbuError) as e:
        ctx.alloc(1 << 50)
    assert e.value.kind == "Hip" and e.value.a != 0
    d = ctx.alloc(24 * 256)
    ctx.generate(1, 0, 256, 16, 12, d)
    assert ctx.reduce(d, 256)["count"] == 256


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 1000, 4097, 1_000_003])
def test_first_mismatch_is_slice_equality_with_a_position(ia, ctx, oracle, n):
    """ibu_records_first_mismatch = `a == b` on record slices (Record: PartialEq / Eq, record.rs:58) with the position:
    checked against numpy on equal slices, a difference in every field of the first / middle / last record, several
    differences (the FIRST one counts), and inputs that are only 8-byte aligned (the word kernel)."""
    recs = oracle.generate(SEED, 0, max(n, 1), 16, 12)[:n]
    raw = np.frombuffer(recs.tobytes(), dtype=np.uint64).copy()
    a, b = ctx.alloc(max(24 * n, 16) + 32), ctx.alloc(max(24 * n, 16) + 32)
    for shift in (0, 8):                                   # 16-B aligned / 8-B aligned device pointers
        da = ia.DeviceBuffer.wrap(ctx, a.ptr + shift, max(24 * n, 16))
        db = ia.DeviceBuffer.wrap(ctx, b.ptr + shift, max(24 * n, 16))
        if n:
            da.upload(raw.view(np.uint8))
            db.upload(raw.view(np.uint8))
        assert ctx.first_mismatch(da, db, n) == n and ctx.records_equal(da, db, n)
        if n == 0:
            continue
        for rec in sorted({0, n // 2, n - 1}):
            for field in range(3):
                other = raw.copy()
                other[3 * rec + field] ^= np.uint64(1) << np.uint64(17 * field + 3)
                db.upload(other.view(np.uint8))
                assert ctx.first_mismatch(da, db, n) == rec, (shift, rec, field)
                assert ctx.first_mismatch(da, db, rec) == rec        # a prefix that ends in front of the difference is equal
        if n > 10:
            other = raw.copy()
            for rec in (n - 1, n // 3, n // 2):
                other[3 * rec + 1] += np.uint64(1)
            db.upload(other.view(np.uint8))
            assert ctx.first_mismatch(da, db, n) == n // 3
    with pytest.raises(ia.IbuError) as e:
        ctx.first_mismatch(a.ptr + 4, b, 1)
    assert e.value.kind == "InvalidArg"
