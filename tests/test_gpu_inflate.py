"""GPU parity for the device DEFLATE decoder (k_inflate.hip) behind ibu_bgzf_scan / ibu_inflate_blocks_device: the blocks of BGZF
streams written by zlib at every level and strategy inflate on the device to the bytes zlib's own inflate gives, a block is
accepted exactly when the host decoder accepts it (length, end of the deflate stream on the block's last byte, CRC-32), and
nothing is written outside a block's own output range.  The reference reads such files as multi-member gzip through niffler
(src/io/reader.rs:345-352); the oracle for the bytes is zlib (the record-stream parity of the whole path: test_gpu_streams.py,
test_gpu_pull_stream.py)."""
import struct
import zlib

import numpy as np
import pytest


pytestmark = pytest.mark.gpu

SEED = 0x1B00007
PAD = 2048
GUARD = 4096
NIFFLER = 2          # IBU_ERR_NIFFLER (include/ibu_hip.h)


@pytest.fixture(scope="module")
def ia():
    import ibu_amd
    return ibu_amd


@pytest.fixture(scope="module")
def ctx(ia):
    c = ia.Context(0)
    yield c
    c.close()


def _bgzf(data, block=0xFF00, level=1, strategy=zlib.Z_DEFAULT_STRATEGY, eof=True):
    out = bytearray()
    for off in range(0, len(data), block):
        chunk = data[off:off + block]
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        cd = c.compress(chunk) + c.flush()
        if len(cd) + 25 > 0xFFFF:                                       # (what bgzip does with a block that grew: stored)
            c = zlib.compressobj(0, zlib.DEFLATED, -15)
            cd = c.compress(chunk) + c.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(cd) + 25)
        out += cd + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    if eof:
        out += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    return bytes(out)


def _inflate_on_device(ia, ctx, comp, blocks, out_bytes):
    """-> (output bytes, status array, first bad, guard zones intact)"""
    d_comp = ctx.alloc(len(comp) + PAD)
    d_comp.upload(np.frombuffer(comp, np.uint8))
    d_out = ctx.alloc(out_bytes + 2 * GUARD)
    d_out.upload(np.full(out_bytes + 2 * GUARD, 0xA5, np.uint8))
    st, first = ctx.inflate_blocks(d_comp, blocks, d_out.ptr + GUARD)
    got = d_out.download(np.uint8)
    d_comp.free()
    d_out.free()
    guards_ok = bool((got[:GUARD] == 0xA5).all() and (got[GUARD + out_bytes:] == 0xA5).all())
    return got[GUARD:GUARD + out_bytes].tobytes(), st, first, guards_ok


def _kinds(oracle, rng):
    recs = oracle.generate(SEED, 0, 40_000, 16, 12)
    period = rng.integers(0, 256, 20_000, dtype=np.uint8).tobytes()
    words = [bytes(rng.integers(97, 123, rng.integers(2, 9), dtype=np.uint8)) for _ in range(300)]
    text = b" ".join(words[i] for i in rng.integers(0, 300, 60_000))
    return {
        "records": recs.tobytes(),
        "random": rng.integers(0, 256, 200_000, dtype=np.uint8).tobytes(),          # incompressible: stored blocks
        "zeros": bytes(150_000),                                                      # distance 1, length 258
        "far": period * 9,                                                            # matches 20 000 bytes back: the global read-back
        "text": text,                                                                 # long codes, short and long distances
        "skewed": bytes(rng.choice(np.array([0, 0, 0, 0, 0, 0, 0, 1, 2, 255], np.uint8), 180_000)),
        "one": b"x",
        "empty": b"",
    }


@pytest.mark.parametrize("level,strategy", [(0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY),
                                            (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE),
                                            (9, zlib.Z_FILTERED)])
def test_blocks_inflate_to_what_zlib_gives(ia, ctx, oracle, level, strategy):
    rng = np.random.default_rng(SEED + level + 16 * strategy)
    for name, data in _kinds(oracle, rng).items():
        for block in (0xFF00, 4093):
            comp = _bgzf(data, block=block, level=level, strategy=strategy)
            blocks, consumed, out_bytes, rc = ia.bgzf_scan(comp)
            assert rc == 0 and consumed == len(comp) and out_bytes == len(data), (name, level, strategy)
            assert len(blocks) == (len(data) + block - 1) // block + 1                  # + the empty EOF block
            got, st, first, guards = _inflate_on_device(ia, ctx, comp, blocks, len(data))
            assert first is None and not st.any(), (name, level, strategy, block, st.nonzero()[0][:5], first)
            assert got == data, (name, level, strategy, block)
            assert guards
            if len(data) > 200_000:
                break


def test_a_bad_block_is_named_and_its_neighbours_are_not_touched(ia, ctx, oracle):
    recs = oracle.generate(SEED, 0, 30_000, 16, 12).tobytes()
    block = 0xFF00
    good = _bgzf(recs, block=block, level=6)
    blocks, _, out_bytes, _ = ia.bgzf_scan(good)
    nb = len(blocks)
    rng = np.random.default_rng(SEED)
    for trial in range(12):
        comp = bytearray(good)
        victim = int(rng.integers(0, nb - 1))
        b = blocks[victim]
        kind = trial % 4
        if kind == 0:                                                 # a flipped bit somewhere in the deflate data
            at = b.comp_offset + int(rng.integers(0, b.comp_len))
            comp[at] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:                                               # the trailer's CRC
            struct.pack_into("<I", comp, b.comp_offset + b.comp_len, b.crc32 ^ 0x10)
        elif kind == 2:                                               # the trailer's length: one too many
            struct.pack_into("<I", comp, b.comp_offset + b.comp_len + 4, b.out_len + 1)
        else:                                                         # a block cut short by a byte (its last byte belongs to the next one now)
            pass
        bl, _, ob, rc = ia.bgzf_scan(bytes(comp))
        assert rc == 0 and len(bl) == nb
        if kind == 3:
            bl[victim].comp_len -= 1
        got, st, first, guards = _inflate_on_device(ia, ctx, bytes(comp), bl, ob)
        assert guards
        # zlib on the same block: does it take it?
        bb = bl[victim]
        try:
            d = zlib.decompressobj(-15)
            o = d.decompress(bytes(comp[bb.comp_offset:bb.comp_offset + bb.comp_len]))
            zl_ok = d.eof and not d.unused_data and len(o) == bb.out_len and zlib.crc32(o) == bb.crc32
        except zlib.error:
            zl_ok = False
        assert (st[victim] == 0) == zl_ok, (trial, kind, st[victim])
        if kind == 1:
            assert st[victim] == 2
        others = np.delete(st, victim)
        assert not others.any()
        assert first == (None if zl_ok else victim)
        for i, x in enumerate(bl):                                    # every good block's bytes are right
            if i != victim or zl_ok:
                assert got[x.out_offset:x.out_offset + x.out_len] == recs[blocks[i].out_offset:blocks[i].out_offset + blocks[i].out_len], (trial, i)


def test_noise_ends_every_wave(ia, ctx):
    """Random bytes as deflate data: every block is refused (or, once in a long while, is a valid tiny stream), no wave spins, nothing
    is written outside the blocks' output ranges."""
    rng = np.random.default_rng(SEED + 9)
    nb, clen, olen = 512, 3000, 8192
    comp = rng.integers(0, 256, nb * clen, dtype=np.uint8).tobytes()
    from ibu_amd._lib import CInflateBlock
    blocks = (CInflateBlock * nb)()
    for i in range(nb):
        blocks[i].comp_offset, blocks[i].comp_len, blocks[i].out_offset, blocks[i].out_len, blocks[i].crc32 = i * clen, clen, i * olen, olen, 0
    got, st, first, guards = _inflate_on_device(ia, ctx, comp, blocks, nb * olen)
    assert guards and st.all() and first == 0


def test_a_window_of_blocks_lands_around_a_slot(ia, ctx, oracle):
    """out_offset is signed: the first block of a batch may begin in front of the bytes the batch keeps (it lands in the headroom),
    the last may end behind them — what the stream does with a slot of whole refills."""
    data = oracle.generate(SEED, 0, 20_000, 16, 12).tobytes()
    comp = _bgzf(data, block=50_000, level=1)
    blocks, _, out_bytes, _ = ia.bgzf_scan(comp)
    shift = 12_345                                                    # the slot begins 12 345 bytes into the first block
    for b in blocks:
        b.out_offset -= shift
    d_comp = ctx.alloc(len(comp) + PAD)
    d_comp.upload(np.frombuffer(comp, np.uint8))
    d_out = ctx.alloc(out_bytes + 65536)
    st, first = ctx.inflate_blocks(d_comp, blocks, d_out.ptr + shift)
    assert first is None and not st.any()
    assert d_out.download(np.uint8)[:out_bytes].tobytes() == data
    d_comp.free()
    d_out.free()


# ---- ibu_load_bgzf_to_device: load_to_vec of the gunzipped file, inflated on the device -----------------------------------------------
@pytest.mark.parametrize("n", [0, 1, 1000, 100_003, 1_000_003])
@pytest.mark.parametrize("block,level", [(0xFF00, 1), (4093, 6), (20, 1), (0xFF00, 0)])
def test_load_bgzf_to_device_gives_the_records_of_the_gunzipped_file(ia, ctx, oracle, tmp_path, n, block, level):
    if block == 20 and n > 1000:
        pytest.skip("tiny blocks only for small files")
    recs = oracle.generate(SEED + n, 0, n, 16, 12)
    plain = struct.pack("<IIIIQ8s", 0x21554249, 2, 16, 12, 0, b"\0" * 8) + recs.tobytes()
    p = tmp_path / "r.ibu.gz"
    p.write_bytes(_bgzf(plain, block=block, level=level))
    ring = {"slots": 3, "slot_records": 40_000, "feeder_threads": 3}     # several ring slots per file, block edges off the slot edges
    h, dptr, got_n, st = ctx.load_bgzf_to_device(str(p), ring=ring)
    assert (h.bc_len, h.umi_len, got_n) == (16, 12, n)
    assert st.records == n and st.bytes_h2d == p.stat().st_size
    if n:
        got = ia.DeviceBuffer.wrap(ctx, dptr, 24 * n).download().tobytes()
        assert got == recs.tobytes()
    ctx.free(dptr)
    # into the caller's buffer; one record short: refused
    buf = ctx.alloc(24 * max(n, 1) + 64)
    buf.upload(np.full(24 * max(n, 1) + 64, 0xA5, np.uint8))
    _, q, got_n, _ = ctx.load_bgzf_to_device(str(p), ring=ring, d_records=buf, cap_records=n)
    assert q == buf.ptr and got_n == n
    whole = buf.download(np.uint8)
    assert whole[:24 * n].tobytes() == recs.tobytes() and (whole[24 * n:] == 0xA5).all()   # nothing behind the records
    if n:
        with pytest.raises(ia.IbuError) as e:
            ctx.load_bgzf_to_device(str(p), ring=ring, d_records=buf, cap_records=n - 1)
        assert e.value.kind == "InvalidArg"
    buf.free()


def test_load_bgzf_to_device_refuses_what_the_reader_refuses(ia, ctx, oracle, tmp_path):
    import gzip
    n = 50_000
    recs = oracle.generate(SEED, 0, n, 16, 12)
    hdr = struct.pack("<IIIIQ8s", 0x21554249, 2, 16, 12, 0, b"\0" * 8)
    plain = hdr + recs.tobytes()
    good = _bgzf(plain, level=1)
    blocks, _, _, _ = ia.bgzf_scan(good)

    def load(data):
        p = tmp_path / "x.gz"
        p.write_bytes(data)
        return ctx.load_bgzf_to_device(str(p))

    def kind_of(data):
        with pytest.raises(ia.IbuError) as e:
            load(data)
        return e.value.kind

    assert kind_of(gzip.compress(plain)) == "Niffler"                       # an ordinary gzip member: the Reader's business
    assert kind_of(plain) == "Niffler"                                      # not compressed at all
    assert kind_of(good[:-60]) == "Niffler"                                 # cut inside the last data block's trailer / the EOF block
    assert kind_of(good[:len(good) // 2]) == "Niffler"                      # cut inside a block
    assert kind_of(good[:7]) == "Niffler"                                   # cut inside the first header
    assert kind_of(b"") == "Io"                                             # nothing there: no header to read
    b = blocks[len(blocks) // 2]
    bad = bytearray(good)
    bad[b.comp_offset + b.comp_len // 2] ^= 0x40
    assert kind_of(bytes(bad)) == "Niffler"                                 # a block that does not inflate (or not to its CRC)
    bad = bytearray(good)
    struct.pack_into("<I", bad, b.comp_offset + b.comp_len, b.crc32 ^ 1)
    assert kind_of(bytes(bad)) == "Niffler"                                 # the trailer's CRC
    bad = bytearray(good)                                                   # the first block (inflated on the host for the header)
    struct.pack_into("<I", bad, blocks[0].comp_offset + blocks[0].comp_len, blocks[0].crc32 ^ 1)
    assert kind_of(bytes(bad)) == "Niffler"
    assert kind_of(_bgzf(plain + b"\x01\x02\x03")) == "InvalidMapSize"      # (length - 32) % 24 != 0
    assert kind_of(_bgzf(hdr[:20])) == "Io"                                 # shorter than a header
    assert kind_of(_bgzf(b"\0" * 32 + recs.tobytes())) == "InvalidMagicNumber"
    assert kind_of(_bgzf(struct.pack("<IIIIQ8s", 0x21554249, 2, 0, 12, 0, b"\0" * 8))) == "InvalidBarcodeLength"
    h, dptr, got_n, _ = load(good)                                          # and the file itself loads
    assert got_n == n
    ctx.free(dptr)
    ctx.set_option("release_staging", 1)                                    # the staging goes back; the next load allocates it again
    with pytest.raises(ia.IbuError):
        ctx.set_option("release_staging", 0)
    h, dptr, got_n, _ = load(good)
    assert got_n == n and ia.DeviceBuffer.wrap(ctx, dptr, 24 * n).download().tobytes() == recs.tobytes()
    ctx.free(dptr)


@pytest.mark.parametrize("kind", ["records", "decoys"])
def test_load_bgzf_to_device_walks_a_large_file_in_pieces(ia, oracle, tmp_path, capfd, kind):
    """From 32 MiB of BGZF on, the block headers are walked in eight pieces side by side, every piece from a GUESSED block start (the
    bytes a bgzip header begins with); the guesses are checked against the chain in front of them and anything off sends the call back
    to the plain walk.  "decoys": stored blocks (level 0) of records whose bytes spell that header thousands of times — every piece's
    first guess is wrong, the result must not be."""
    import os
    n = 2_000_003
    recs = oracle.generate(SEED + 77, 0, n, 32, 32)
    level = 1
    if kind == "decoys":
        level = 0                                                   # stored: the records' bytes stand in the file as they are
        raw = recs.view(np.uint8).reshape(n, 24)
        decoy = np.frombuffer(bytes([0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0x00, 0x42, 0x43, 0x02, 0x00, 0x11, 0x22, 0, 0, 0, 0, 0, 0]), np.uint8)
        raw[::97] = decoy
    plain = struct.pack("<IIIIQ8s", 0x21554249, 2, 32, 32, 0, b"\0" * 8) + recs.tobytes()
    p = tmp_path / "big.ibu.gz"
    p.write_bytes(_bgzf(plain, level=level))
    assert p.stat().st_size >= 32 << 20
    c = ia.Context(0)
    try:
        capfd.readouterr()
        h, dptr, got_n, st = c.load_bgzf_to_device(str(p), ring={"slots": 3, "slot_records": 1 << 19, "feeder_threads": 4})
        err = capfd.readouterr().err
        assert got_n == n and (h.bc_len, h.umi_len) == (32, 32)
        assert ia.DeviceBuffer.wrap(c, dptr, 24 * n).download().tobytes() == recs.tobytes()
        c.free(dptr)
        if os.environ.get("IBU_TRACE_SORT", "") not in ("", "0"):
            assert ("walked in 8 pieces side by side" if kind == "records" else "walked in one go") in err, err
    finally:
        c.close()


@pytest.mark.parametrize("n,block,level", [(1_000_003, 0xFF00, 1), (100_003, 4093, 6), (777, 20, 1), (5, 0xFF00, 1), (0, 0xFF00, 1)])
@pytest.mark.parametrize("n_shards", [2, 3, 8])
def test_load_bgzf_shards_are_the_ranges_of_process_parallel(ia, ctx, oracle, tmp_path, n, block, level, n_shards):
    """Every device of a node loads its own range of the same BGZF file: shard i of k is records [i * (n / k), (i + 1) * (n / k)) with
    the remainder in the last (mmap.rs:297-307); the blocks wholly inside a range inflate on the device, the two that straddle its
    ends (and the header's) on the host; nothing is written outside the shard's buffer."""
    recs = oracle.generate(SEED + 5 * n, 0, n, 16, 12)
    plain = struct.pack("<IIIIQ8s", 0x21554249, 2, 16, 12, 0, b"\0" * 8) + recs.tobytes()
    p = tmp_path / "s.ibu.gz"
    p.write_bytes(_bgzf(plain, block=block, level=level))
    ring = {"slots": 3, "slot_records": 30_000, "feeder_threads": 3}
    per = n // n_shards
    got_all = b""
    for i in range(n_shards):
        want_first, want_n = i * per, (per if i + 1 < n_shards else n - per * (n_shards - 1))
        buf = ctx.alloc(24 * want_n + 128)
        buf.upload(np.full(24 * want_n + 128, 0x5A, np.uint8))
        h, q, got_n, first, st = ctx.load_bgzf_shard_to_device(str(p), i, n_shards, ring=ring, d_records=buf.ptr + 64, cap_records=want_n)
        assert (got_n, first) == (want_n, want_first) and (h.bc_len, h.umi_len) == (16, 12), (i, got_n, first)
        whole = buf.download(np.uint8)
        assert (whole[:64] == 0x5A).all() and (whole[64 + 24 * want_n:] == 0x5A).all()
        assert st.bytes_h2d <= p.stat().st_size
        got_all += whole[64:64 + 24 * want_n].tobytes()
        buf.free()
    assert got_all == recs.tobytes()
    with pytest.raises(ia.IbuError) as e:
        ctx.load_bgzf_shard_to_device(str(p), n_shards, n_shards)
    assert e.value.kind == "InvalidArg"
    # the library allocates a shard's buffer as well
    h, q, got_n, first, _ = ctx.load_bgzf_shard_to_device(str(p), n_shards - 1, n_shards, ring=ring)
    assert first == per * (n_shards - 1) and got_n == n - first
    if got_n:
        assert ia.DeviceBuffer.wrap(ctx, q, 24 * got_n).download().tobytes() == recs[first:].tobytes()
    if q:
        ctx.free(q)


@pytest.mark.parametrize("n,block", [(200_003, 0xFF00), (50_001, 4093), (3_000, 20)])
def test_load_bgzf_with_the_decoder_launched_ahead_of_the_copies(ia, oracle, tmp_path, n, block):
    """Large files get ONE launch of the decoder ahead of the copies: its waves wait (asleep, reading past the caches, for a bounded
    time) until the copy stream has said that their blocks have arrived.  Forced here for small files (option "inflate_one_launch"),
    with ring slots much smaller than the file so that the launch really is ahead; whole file and shards; a corrupt block still
    fails the call."""
    recs = oracle.generate(SEED + 3 * n, 0, n, 16, 12)
    plain = struct.pack("<IIIIQ8s", 0x21554249, 2, 16, 12, 0, b"\0" * 8) + recs.tobytes()
    good = _bgzf(plain, block=block, level=1)
    p = tmp_path / "ahead.ibu.gz"
    p.write_bytes(good)
    c = ia.Context(0)
    try:
        c.set_option("inflate_one_launch", 8)
        ring = {"slots": 3, "slot_records": 8_000, "feeder_threads": 2}
        for rep in range(2):                                           # (the second call: staging and marks reused)
            h, dptr, got_n, st = c.load_bgzf_to_device(str(p), ring=ring)
            assert got_n == n and ia.DeviceBuffer.wrap(c, dptr, 24 * n).download().tobytes() == recs.tobytes()
            c.free(dptr)
        got = b""
        for i in range(3):
            h, q, k, first, _ = c.load_bgzf_shard_to_device(str(p), i, 3, ring=ring)
            got += ia.DeviceBuffer.wrap(c, q, 24 * k).download().tobytes() if k else b""
            if q:
                c.free(q)
        assert got == recs.tobytes()
        blocks, _, _, _ = ia.bgzf_scan(good)
        b = blocks[len(blocks) * 2 // 3]
        bad = bytearray(good)
        struct.pack_into("<I", bad, b.comp_offset + b.comp_len, b.crc32 ^ 4)
        p.write_bytes(bytes(bad))
        with pytest.raises(ia.IbuError) as e:
            c.load_bgzf_to_device(str(p), ring=ring)
        assert e.value.kind == "Niffler"
        with pytest.raises(ia.IbuError):
            c.set_option("inflate_one_launch", 1 << 20)
    finally:
        c.close()


def test_load_bgzf_from_a_source_slower_than_the_waves_wait(ia, oracle, tmp_path, capfd):
    """The waves of the launch that runs ahead of the copies give up after ~4 s without their blocks — a disk that slow is no reason to
    fail: what they left is inflated once everything has arrived.  1.6 s before each of five pieces."""
    import os
    n = 120_001
    recs = oracle.generate(SEED + 31, 0, n, 16, 12)
    plain = struct.pack("<IIIIQ8s", 0x21554249, 2, 16, 12, 0, b"\0" * 8) + recs.tobytes()
    p = tmp_path / "slow.ibu.gz"
    p.write_bytes(_bgzf(plain, level=1))
    c = ia.Context(0)
    try:
        c.set_option("inflate_one_launch", 4)
        c.set_option("load_piece_delay_ms", 1600)
        ring = {"slots": 2, "slot_records": 14_000, "feeder_threads": 2}   # 1.44 MB of BGZF in pieces of 336 KB
        capfd.readouterr()
        h, dptr, got_n, st = c.load_bgzf_to_device(str(p), ring=ring)
        err = capfd.readouterr().err
        assert got_n == n and ia.DeviceBuffer.wrap(c, dptr, 24 * n).download().tobytes() == recs.tobytes()
        c.free(dptr)
        if os.environ.get("IBU_TRACE_SORT", "") not in ("", "0"):
            assert "came later than the waves waited" in err, err
    finally:
        c.close()


def _fuzz_data(rng):
    """A byte string stitched from pieces of different character: random, runs, short periods, long-distance repeats, records."""
    parts = []
    for _ in range(int(rng.integers(1, 9))):
        kind = int(rng.integers(0, 6))
        n = int(rng.integers(1, 120_000))
        if kind == 0:
            parts.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
        elif kind == 1:
            parts.append(bytes([int(rng.integers(0, 256))]) * n)
        elif kind == 2:
            p = rng.integers(0, 256, int(rng.integers(2, 40)), dtype=np.uint8).tobytes()
            parts.append((p * (n // len(p) + 1))[:n])
        elif kind == 3:
            p = rng.integers(0, 256, int(rng.integers(300, 30_000)), dtype=np.uint8).tobytes()
            parts.append((p * (n // len(p) + 2))[:n])
        elif kind == 4:
            parts.append(bytes(rng.choice(np.array([65, 67, 71, 84, 10], np.uint8), n)))
        else:
            r = np.zeros(n // 24 + 1, dtype=[("b", "<u8"), ("u", "<u8"), ("i", "<u8")])
            r["b"] = rng.integers(0, 1 << 32, len(r))
            r["u"] = rng.integers(0, 1 << 24, len(r))
            r["i"] = np.arange(len(r))
            parts.append(r.tobytes()[:n])
    return b"".join(parts)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("IBU_FUZZ_SEEDS_INFLATE", "6"))))
def test_inflate_fuzz(ia, ctx, seed):
    """Seeded fuzz (IBU_FUZZ_SEEDS_INFLATE widens it): stitched data, any level / strategy / block size, then bits flipped in some blocks:
    the device's bytes = zlib's, and the device refuses exactly the blocks zlib refuses (length, end of stream, CRC-32)."""
    rng = np.random.default_rng(SEED * 1000 + seed)
    data = _fuzz_data(rng)
    level = int(rng.integers(0, 10))
    strategy = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.integers(0, 5))]
    block = int(rng.choice([0xFF00, 0xF000, 30_000, 4093, 513, 97]))   # (bgzip stops at 0xFF00: an incompressible block still fits the 16-bit BSIZE)
    if block < 1000:
        data = data[:60_000]
    comp = bytearray(_bgzf(data, block=block, level=level, strategy=strategy))
    blocks, consumed, out_bytes, rc = ia.bgzf_scan(bytes(comp))
    assert rc == 0 and out_bytes == len(data)
    nb = len(blocks)
    hurt = set()
    for _ in range(int(rng.integers(0, 4))):                        # flipped bits in the deflate data of a few blocks
        v = int(rng.integers(0, nb))
        if blocks[v].comp_len:
            comp[blocks[v].comp_offset + int(rng.integers(0, blocks[v].comp_len))] ^= 1 << int(rng.integers(0, 8))
            hurt.add(v)
    got, st, first, guards = _inflate_on_device(ia, ctx, bytes(comp), blocks, len(data))
    assert guards
    want_bad = []
    for v in sorted(hurt):
        b = blocks[v]
        try:
            d = zlib.decompressobj(-15)
            o = d.decompress(bytes(comp[b.comp_offset:b.comp_offset + b.comp_len]))
            ok = d.eof and not d.unused_data and len(o) == b.out_len and zlib.crc32(o) == b.crc32
        except zlib.error:
            ok = False
        if not ok:
            want_bad.append(v)
    assert sorted(st.nonzero()[0].tolist()) == want_bad, (seed, level, strategy, block, st.nonzero()[0][:8], want_bad)
    assert first == (want_bad[0] if want_bad else None)
    for i, b in enumerate(blocks):
        if i not in want_bad:
            assert got[b.out_offset:b.out_offset + b.out_len] == data[b.out_offset:b.out_offset + b.out_len], (seed, i)


def test_reader_process_device_of_a_bgzf_file_inflates_on_the_device(ia, oracle, tmp_path):
    """`Reader::from_path(bgzf).process(...)` with the records on the device: the library reads the file itself, the compressed bytes cross
    the link, the blocks inflate on the device (option "bgzf_device" = 1, the default) — same results as through the Reader's host inflate
    (= 0); a file that load does not take (cut inside a record, a bad block) goes through the Reader's own path and fails as it fails there;
    a reader something has been read from keeps to the host path."""
    n = 300_007
    recs = oracle.generate(SEED + 404, 0, n, 16, 12)
    plain = struct.pack("<IIIIQ8s", 0x21554249, 2, 16, 12, 0, b"\0" * 8) + recs.tobytes()
    good = _bgzf(plain, level=1)
    p = tmp_path / "f.ibu.gz"
    p.write_bytes(good)
    want = oracle.reduce_records(recs)
    bc, umi, idx = oracle.decode_records(recs, 16, 12)
    ring = {"slots": 3, "slot_records": 60_000, "feeder_threads": 2}
    c = ia.Context(0)
    try:
        for dev in (1, 0):
            c.set_option("bgzf_device", dev)
            r = ia.Reader.from_path(p)
            res, st = r.process_device(c, ia.PROC_REDUCE, ring=ring)
            assert res == want and st.records == n
            assert (st.bytes_h2d == len(good)) == bool(dev), (dev, st.bytes_h2d, len(good))   # the compressed file / the records crossed the link
            assert not r.read_batch()                                 # the reader stands at its end
            r.close()
            r = ia.Reader.from_path(p)
            d_bc, d_umi, d_idx = c.alloc(n * 16), c.alloc(n * 12), c.alloc(n * 8)
            r.process_device(c, ia.PROC_DECODE, sink=(d_bc, d_umi, d_idx), ring=ring)
            assert (d_bc.download().tobytes(), d_umi.download().tobytes(), d_idx.download().tobytes()) == (bc.tobytes(), umi.tobytes(), idx.tobytes())
            r.close()
            r = ia.Reader.from_path(p)                                # a sink that is too small: InvalidArg either way
            with pytest.raises(ia.IbuError) as e:
                r.process_device(c, ia.PROC_DECODE, sink=(d_bc, d_umi, d_idx, n - 1), ring=ring)
            assert e.value.kind == "InvalidArg"
            r.close()
            for b in (d_bc, d_umi, d_idx):
                b.free()
        c.set_option("bgzf_device", 1)
        for rng_bytes in (len(good) // 2 + 1, len(good) // 5, 40_000):   # the file in 2, 6 and ~90 ranges: rows land where they belong
            c.set_option("bgzf_range_bytes", rng_bytes)
            r = ia.Reader.from_path(p)
            res, st = r.process_device(c, ia.PROC_REDUCE, ring=ring)
            assert res == want and st.records == n and st.bytes_h2d <= len(good)
            r.close()
            r = ia.Reader.from_path(p)
            d_bc, d_umi, d_idx = c.alloc(n * 16), c.alloc(n * 12), c.alloc(n * 8)
            r.process_device(c, ia.PROC_DECODE, sink=(d_bc, d_umi, d_idx), ring=ring)
            assert (d_bc.download().tobytes(), d_umi.download().tobytes(), d_idx.download().tobytes()) == (bc.tobytes(), umi.tobytes(), idx.tobytes())
            r.close()
            for b in (d_bc, d_umi, d_idx):
                b.free()
            c.set_option("release_staging", 1)                        # (the next round starts without the buffers of this one)
        c.set_option("bgzf_range_bytes", 0)
        r = ia.Reader.from_path(p)                                    # three records taken on the host first: the host path goes on from there
        head = [next(r) for _ in range(3)]
        res, st = r.process_device(c, ia.PROC_REDUCE, ring=ring)
        assert res == oracle.reduce_records(recs[3:]) and st.bytes_h2d != len(good)
        r.close()
        kinds = {}
        for name, data in (("cut", _bgzf(plain[:-5], level=1)), ("bad", None)):
            if data is None:
                blocks, _, _, _ = ia.bgzf_scan(good)
                b = blocks[len(blocks) // 2]
                data = bytearray(good)
                struct.pack_into("<I", data, b.comp_offset + b.comp_len, b.crc32 ^ 2)
                data = bytes(data)
            q = tmp_path / (name + ".gz")
            q.write_bytes(data)
            for dev in (1, 0):
                c.set_option("bgzf_device", dev)
                r = ia.Reader.from_path(q)
                with pytest.raises(ia.IbuError) as e:
                    r.process_device(c, ia.PROC_REDUCE, ring=ring)
                kinds[(name, dev)] = (e.value.kind, getattr(e.value, "pos", None))
                r.close()
        assert kinds[("cut", 1)] == kinds[("cut", 0)] and kinds[("cut", 0)][0] == "TruncatedRecord", kinds
        assert kinds[("bad", 1)] == kinds[("bad", 0)] and kinds[("bad", 0)][0] == "Niffler", kinds
    finally:
        c.close()
