#!/usr/bin/env python3
"""Test-data writers for the compressed-input paths (tools/e2e.py, bench.py's e2e leg): a records file as ONE gzip member
compressed in parallel the way pigz does it, as many members, and as BGZF blocks.  Not part of the product."""
import os
import zlib
from concurrent.futures import ThreadPoolExecutor as ProcessPoolExecutor   # threads: zlib releases the GIL while it compresses, and a
# fork of a process that holds an initialised HIP runtime, torch and feeder threads can deadlock — the callers are such processes


def _gz_member(args):
    path, off, length, level = args
    with open(path, "rb") as f:
        f.seek(off)
        data = f.read(length)
    c = zlib.compressobj(level, zlib.DEFLATED, 31)  # 31 = gzip container
    return c.compress(data) + c.flush()


def _deflate_piece(args):
    path, off, length, level, last = args
    with open(path, "rb") as f:
        f.seek(off)
        data = f.read(length)
    c = zlib.compressobj(level, zlib.DEFLATED, -15)  # raw deflate; every piece starts with an empty window
    return c.compress(data) + c.flush(zlib.Z_FINISH if last else zlib.Z_SYNC_FLUSH), zlib.crc32(data), len(data)


def gzip_single_member(src, dst, level=1, piece_bytes=64 << 20, workers=16):
    """ONE gzip member (what `gzip` / `pigz` write: one header, one deflate stream, one trailer), compressed in parallel
    the way pigz does it: pieces end in a sync flush (byte-aligned, not final), the last one finishes the stream."""
    import struct
    size = os.path.getsize(src)
    offs = list(range(0, size, piece_bytes))
    jobs = [(src, off, min(piece_bytes, size - off), level, off == offs[-1]) for off in offs]
    crc, total = 0, 0
    with ProcessPoolExecutor(max_workers=workers) as ex, open(dst, "wb") as out:
        out.write(b"\x1f\x8b\x08\x00\0\0\0\0\x00\xff")
        for blob, c, k in ex.map(_deflate_piece, jobs):
            out.write(blob)
            crc = _crc32_combine(crc, c, k)
            total += k
        out.write(struct.pack("<II", crc, total & 0xFFFFFFFF))
    return os.path.getsize(dst)


def _crc32_combine(crc1, crc2, len2):
    """zlib's crc32_combine (GF(2) matrix method), which the Python module does not export."""
    if len2 == 0:
        return crc1

    def times(mat, vec):
        s, i = 0, 0
        while vec:
            if vec & 1:
                s ^= mat[i]
            vec >>= 1
            i += 1
        return s

    def square(mat):
        return [times(mat, mat[n]) for n in range(32)]

    odd = [0xEDB88320] + [1 << n for n in range(31)]
    even = square(odd)
    odd = square(even)
    while True:
        even = square(odd)
        if len2 & 1:
            crc1 = times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = square(even)
        if len2 & 1:
            crc1 = times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return crc1 ^ crc2


def _bgzf_range(args):
    import struct
    path, off, length, level = args
    with open(path, "rb") as f:
        f.seek(off)
        data = f.read(length)
    out = bytearray()
    for o in range(0, len(data), 0xFF00):
        chunk = data[o:o + 0xFF00]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        cd = c.compress(chunk) + c.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(cd) + 25)
        out += cd + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    return bytes(out)


def bgzf_parallel(src, dst, level=1, range_bytes=0xFF00 * 256, workers=16):
    """What `bgzip` writes: <= 64 KiB gzip blocks with a BC size subfield + the empty EOF block."""
    size = os.path.getsize(src)
    jobs = [(src, off, min(range_bytes, size - off), level) for off in range(0, size, range_bytes)]
    with ProcessPoolExecutor(max_workers=workers) as ex, open(dst, "wb") as out:
        for blob in ex.map(_bgzf_range, jobs):
            out.write(blob)
        out.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return os.path.getsize(dst)


def gzip_parallel(src, dst, level=1, member_bytes=64 << 20, workers=16):
    """A multi-member .gz of `src` (what `pigz`/`bgzip` write); members compressed in parallel."""
    size = os.path.getsize(src)
    jobs = [(src, off, min(member_bytes, size - off), level) for off in range(0, size, member_bytes)]
    with ProcessPoolExecutor(max_workers=workers) as ex, open(dst, "wb") as out:
        for blob in ex.map(_gz_member, jobs):
            out.write(blob)
    return os.path.getsize(dst)
