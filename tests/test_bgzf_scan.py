"""ibu_bgzf_scan (host only): the walk over the block headers of a BGZF stream — what ibu_load_bgzf_to_device and ibu_inflate_blocks_device
build on.  The reference reads such files as multi-member gzip through niffler (src/io/reader.rs:345-352); the framing is the SAM/BAM
specification's (section 4.1).  Runs without a GPU."""
import struct

import numpy as np
import pytest

from tests.bgzf import bgzf_compress

NIFFLER = 2          # IBU_ERR_NIFFLER (include/ibu_hip.h)


@pytest.fixture(scope="module")
def ia():
    import ibu_amd
    return ibu_amd


def test_scan_describes_whole_blocks_and_names_what_is_not_bgzf(ia):
    data = bytes(range(256)) * 1000
    comp = bgzf_compress(data, level=1)
    blocks, consumed, out_bytes, rc = ia.bgzf_scan(comp)
    assert rc == 0 and consumed == len(comp) and out_bytes == len(data)
    pos = 0
    for b in blocks:                                                  # the descriptors are the members' own fields
        assert b.comp_offset == pos + 18 and comp[pos:pos + 4] == b"\x1f\x8b\x08\x04"
        bsize = struct.unpack_from("<H", comp, pos + 16)[0] + 1
        assert b.comp_len == bsize - 26
        assert (b.crc32, b.out_len) == struct.unpack_from("<II", comp, pos + bsize - 8)
        pos += bsize
    assert [b.out_offset for b in blocks] == list(np.cumsum([0] + [b.out_len for b in blocks])[:-1])
    # a buffer that ends inside a block: more may come (final = 0) / the stream is cut off (final = 1)
    cut = len(comp) - 40
    b0, c0, o0, rc0 = ia.bgzf_scan(comp[:cut], final=False)
    assert rc0 == 0 and len(b0) == len(blocks) - 2 and c0 == blocks[len(b0)].comp_offset - 18
    b1, c1, o1, rc1 = ia.bgzf_scan(comp[:cut], final=True)
    assert rc1 == NIFFLER and len(b1) == len(b0) and c1 == c0 and o1 == o0
    # cap: the walk stops after `cap` blocks and says where
    b2, c2, _, rc2 = ia.bgzf_scan(comp, cap=3)
    assert rc2 == 0 and len(b2) == 3 and c2 == blocks[3].comp_offset - 18
    # an ordinary gzip member is not a BGZF block
    import gzip
    b3, c3, _, rc3 = ia.bgzf_scan(gzip.compress(data[:1000]))
    assert rc3 == NIFFLER and len(b3) == 0 and c3 == 0
    assert ia.bgzf_scan(b"")[3] == 0


def test_scan_walks_many_blocks_in_chunks(ia):
    """More blocks than one call of the C function describes for the Python wrapper (65 536): the wrapper's offsets carry over."""
    data = bytes(1_400_000)
    comp = bgzf_compress(data, block=20)
    blocks, consumed, out_bytes, rc = ia.bgzf_scan(comp)
    assert rc == 0 and consumed == len(comp) and out_bytes == len(data) and len(blocks) == 70_001
    offs = np.array([b.out_offset for b in blocks[::997]])
    assert (offs == 20 * np.arange(0, 70_001, 997)).all()
    assert blocks[-1].out_len == 0 and blocks[-2].comp_offset + blocks[-2].comp_len + 8 + 18 == blocks[-1].comp_offset
