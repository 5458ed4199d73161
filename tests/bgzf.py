"""Test helper: write BGZF (bgzip) streams."""
def bgzf_compress(data, block=0xFF00, level=1, eof=True):
    """A BGZF (bgzip) stream of `data`: gzip members of <= 64 KiB each carrying their compressed size in a "BC"
    extra subfield, closed by the 28-byte empty EOF block — what `bgzip` writes (SAM/BAM spec, section 4.1)."""
    import struct
    import zlib

    out = bytearray()
    for off in range(0, len(data), block):
        chunk = data[off:off + block]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        cd = c.compress(chunk) + c.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(cd) + 25)
        out += cd + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    if eof:
        out += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    return bytes(out)
