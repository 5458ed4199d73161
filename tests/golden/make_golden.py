#!/usr/bin/env python3
"""Writes tests/golden/kat.json — the known-answer vectors that pin the oracle.

Provenance.  The reference (noamteyssier/ibu v0.2.1) is Rust and cannot be compiled or
imported in this image, so no reference *outputs* can be captured.  What its own unit tests
assert, however, are literal values (sizes, field values, byte counts, error payloads); this
script writes exactly those literals down, each with the reference file:line that asserts
it, and expands the `#[repr(C)]` little-endian layouts (header.rs:48-61, record.rs:62-66)
with nothing but `struct.pack`.  It deliberately imports neither the oracle nor the product,
so the vectors are independent of both.

Run:  python tests/golden/make_golden.py     (rewrites kat.json next to this file)
"""
import json
import os
import struct

MAGIC = 0x21554249  # header.rs:5
VERSION = 2  # header.rs:6


def header_bytes(bc_len, umi_len, flags=0, magic=MAGIC, version=VERSION, reserved=b"\0" * 8):
    # #[repr(C)] {u32 magic, u32 version, u32 bc_len, u32 umi_len, u64 flags, [u8;8]}
    return struct.pack("<IIIIQ8s", magic, version, bc_len, umi_len, flags, reserved)


def record_bytes(barcode, umi, index):
    # #[repr(C)] {u64 barcode, u64 umi, u64 index}
    return struct.pack("<QQQ", barcode, umi, index)


def main():
    kat = {}

    # B1-B3 -------------------------------------------------------------------------------
    kat["sizes"] = {"header": 32, "record": 24, "src": "header.rs:248-251, record.rs:149-152"}
    kat["magic_bytes"] = {"hex": b"IBU!".hex(), "src": "header.rs:374-378"}
    kat["header_new_16_12"] = {
        "bc_len": 16,
        "umi_len": 12,
        "hex": header_bytes(16, 12).hex(),
        "fields": {"magic": MAGIC, "version": VERSION, "flags": 0, "reserved": "00" * 8},
        "src": "header.rs:236-245",
    }
    kat["header_sorted_16_12"] = {
        "hex": header_bytes(16, 12, flags=1).hex(),
        "flags": 1,
        "src": "header.rs:254-270 (set twice -> still 1), :362-371",
    }
    kat["header_roundtrip_20_10"] = {"hex": header_bytes(20, 10).hex(), "src": "header.rs:351-359"}
    kat["validate_ok"] = {"cases": [[16, 12], [1, 1], [32, 32]], "src": "header.rs:273-282"}
    # B4 ----------------------------------------------------------------------------------
    kat["validate_err"] = {
        "cases": [
            {"hex": header_bytes(16, 12, magic=0x12345678).hex(), "kind": "InvalidMagicNumber",
             "a": MAGIC, "b": 0x12345678, "src": "header.rs:285-297"},
            {"hex": header_bytes(16, 12, version=1).hex(), "kind": "InvalidVersion",
             "a": VERSION, "b": 1, "src": "header.rs:300-312"},
            {"hex": header_bytes(0, 12).hex(), "kind": "InvalidBarcodeLength", "a": 0, "b": 0,
             "src": "header.rs:318-322"},
            {"hex": header_bytes(33, 12).hex(), "kind": "InvalidBarcodeLength", "a": 33, "b": 0,
             "src": "header.rs:325-329"},
            {"hex": header_bytes(16, 0).hex(), "kind": "InvalidUmiLength", "a": 0, "b": 0,
             "src": "header.rs:336-340"},
            {"hex": header_bytes(16, 33).hex(), "kind": "InvalidUmiLength", "a": 33, "b": 0,
             "src": "header.rs:343-347"},
            # B5: 32 zero bytes -> InvalidMagicNumber from Reader::new
            {"hex": ("00" * 32), "kind": "InvalidMagicNumber", "a": MAGIC, "b": 0,
             "src": "reader.rs:568-575, lib.rs:160-169"},
        ]
    }
    # B6-B7 -------------------------------------------------------------------------------
    kat["record_bytes"] = {
        "cases": [
            {"rec": [0x123456789ABCDEF0, 0xFEDCBA9876543210, 2**64 - 1],
             "hex": record_bytes(0x123456789ABCDEF0, 0xFEDCBA9876543210, 2**64 - 1).hex(),
             "src": "record.rs:235-243"},
            {"rec": [0, 0, 0], "hex": record_bytes(0, 0, 0).hex(), "src": "record.rs:246-253"},
            {"rec": [2**64 - 1] * 3, "hex": record_bytes(2**64 - 1, 2**64 - 1, 2**64 - 1).hex(),
             "src": "record.rs:256-264"},
            {"rec": [0x1234, 0x5678, 42], "hex": record_bytes(0x1234, 0x5678, 42).hex(),
             "src": "record.rs:140-146"},
        ]
    }
    # B8: README example file ---------------------------------------------------------------
    readme = header_bytes(16, 12, flags=1) + record_bytes(0x1100, 0x100011, 0) + record_bytes(0x1101, 0x100010, 1)
    kat["readme_file"] = {
        "hex": readme.hex(),
        "len": 80,
        "records": [[0x1100, 0x100011, 0], [0x1101, 0x100010, 1]],
        "src": "README.md:66-85, lib.rs:38-71",
    }
    # B9: writer lengths ------------------------------------------------------------------
    kat["writer_lengths"] = {
        "new": 32, "headless": 0,
        "one_record": {"rec": [0x1234, 0x5678, 42], "len": 56,
                       "hex": (header_bytes(16, 12) + record_bytes(0x1234, 0x5678, 42)).hex()},
        "batch3": {"recs": [[1, 2, 3], [4, 5, 6], [7, 8, 9]], "len": 104,
                   "hex": (header_bytes(16, 12) + b"".join(record_bytes(*r) for r in [[1, 2, 3], [4, 5, 6], [7, 8, 9]])).hex()},
        "headless_one": {"rec": [1, 2, 3], "len": 24},
        "src": "writer.rs:636-694, :159-168",
    }
    # B10/B11 -----------------------------------------------------------------------------
    kat["writer_buffer"] = {
        "buffer_bytes": 48 * 1024 * 24,
        "records_per_buffer": 49152,
        "note": "49152 write_record calls leave inner at 32 B; the 49153rd flushes 1179648 B",
        "direct_batch_records": 100000,
        "src": "writer.rs:10, :767-787, :709-719, code :260-273,:325-331",
    }
    # B12: mixed write order ----------------------------------------------------------------
    mixed = [[1, 2, 3], [4, 5, 6], [7, 8, 9], [10, 20, 30], [11, 22, 33], [12, 24, 36]]
    kat["writer_mixed"] = {
        "write_record": [1, 2, 3],
        "write_batch": [[4, 5, 6], [7, 8, 9]],
        "write_iter": "Record::new(i, i*2, i*3) for i in 10..13",
        "expect": mixed,
        "hex": (header_bytes(16, 12) + b"".join(record_bytes(*r) for r in mixed)).hex(),
        "src": "writer.rs:835-865",
    }
    # B13: ingest ---------------------------------------------------------------------------
    kat["writer_ingest"] = {"aux_records": [[1, 2, 3], [4, 5, 6]], "main_records_written": 2,
                            "aux_inner_len_after": 0, "src": "writer.rs:722-741"}
    # B14: streaming read -------------------------------------------------------------------
    kat["reader_stream"] = {
        "n": 100000, "formula": "(i, 2i, 3i)", "refills": [49152, 49152, 1696],
        "bytes_read_small": {"records": 10, "expect": 32 + 10 * 24},
        "src": "reader.rs:607-616, :639-653, :744-766",
    }
    # B15: truncation -------------------------------------------------------------------------
    one = header_bytes(16, 12) + record_bytes(1, 2, 3)
    kat["truncated"] = {
        "hex": one[:-5].hex(),
        "stream": {"kind": "TruncatedRecord", "pos": 32},
        "load_to_vec": {"kind": "InvalidMapSize"},
        "src": "reader.rs:619-636, :723-741, code :232-237",
    }
    # B16: mmap -------------------------------------------------------------------------------
    kat["mmap"] = {
        "slice_100": {"formula": "(i, 2i, 3i)", "n": 100,
                      "checks": [{"s": 0, "e": 100, "first": [0, 0, 0], "last": [99, 198, 297]},
                                 {"s": 10, "e": 20, "first": [10, 20, 30], "last": [19, 38, 57]},
                                 {"s": 50, "e": 51, "first": [50, 100, 150], "last": [50, 100, 150]}],
                      "src": "mmap.rs:396-423"},
        "slice_errors_len1": [{"s": 0, "e": 2, "idx": 2, "max": 1}, {"s": 1, "e": 1, "idx": 1, "max": 1},
                              {"s": 1, "e": 0, "idx": 0, "max": 1}],
        "slice_errors_src": "mmap.rs:426-452",
        "parallel_10000": {"formula": "(i, 2i, 3i)", "threads": 4, "count": 10000, "sum": 299970000,
                           "src": "mmap.rs:455-481"},
        "parallel_auto_1000": {"formula": "(i, 0, 0)", "threads": 0, "count": 1000, "src": "mmap.rs:484-500"},
        "empty": {"threads": 2, "count": 0, "len": 0, "src": "mmap.rs:503-519"},
        "large": {"formula": "(i % 1000, i % 500, i)", "n": 100000, "s": 50000, "e": 50010,
                  "first_index": 50000, "src": "mmap.rs:546-565"},
    }
    # B17: ordering -----------------------------------------------------------------------------
    kat["ordering"] = {
        "unsorted": [[1, 1, 1], [0, 1, 1], [1, 0, 1], [0, 0, 1], [1, 1, 0], [0, 1, 0], [1, 0, 0], [0, 0, 0]],
        "sorted": [[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1], [1, 0, 0], [1, 0, 1], [1, 1, 0], [1, 1, 1]],
        "greater": [[[1, 1, 0], [0, 1, 1]]],
        "src": "record.rs:184-232",
    }
    # cfg 1: examples/roundtrip.rs:21-22,34-38 at N = 1e6 (derived from the formula)
    n = 10**6
    s = [0, 0, 0]
    x = [0, 0, 0]
    for i in range(n):
        f = (i % 1_000_000, (i * 31) % 1_000_000, i)
        for k in range(3):
            s[k] = (s[k] + f[k]) & (2**64 - 1)
            x[k] ^= f[k]
    kat["roundtrip_1e6"] = {
        "n": n, "formula": "(i % 1e6, 31 i % 1e6, i)", "header": header_bytes(16, 12, flags=1).hex(),
        "file_len": 32 + 24 * n, "sums": s, "xors": x, "checksum": x[0] ^ x[1] ^ x[2],
        "src": "examples/roundtrip.rs:21-22,34-38,84-87",
    }
    # 2-bit codec: the only stated facts (record.rs:22-27) + bitnuc's published "ACGT" example.
    kat["codec"] = {
        "parity": "UNPINNED - the reference holds no codec code or test; see oracle/ibu_oracle.h",
        "table": {"A": 0, "C": 1, "G": 2, "T": 3},
        "max_len": 32,
        # provenance per vector.  "table": follows from the code table of record.rs:22-25 alone, whatever the bit order
        # (single bases, homopolymers).  "recalled": bitnuc's README example as remembered — NOT verifiable here (the
        # crate is neither vendored nor in Cargo.toml).  "derived-lsb" / "derived-msb": the table + the stated order.
        "examples": [
            {"seq": "ACGT", "code": 0b11100100, "prov": "recalled", "src": "bitnuc README (as_2bit(b\"ACGT\") == 0b11100100), from memory"},
            {"seq": "A", "code": 0, "prov": "table"}, {"seq": "T", "code": 3, "prov": "table"},
            {"seq": "TA", "code": 3, "prov": "derived-lsb"}, {"seq": "AT", "code": 12, "prov": "derived-lsb"},
            {"seq": "T" * 32, "code": 2**64 - 1, "prov": "table"}, {"seq": "A" * 32, "code": 0, "prov": "table"},
            {"seq": "acgt", "code": 0b11100100, "prov": "derived-lsb"},
        ],
        # the hedge: the same sequences under IBU_BASE_ORDER_MSB_FIRST (base i at bits [2(len-1-i), ...]: the sequence
        # read as a base-4 number).  ONE external vector decides: bitnuc::as_2bit(b"ACGT") is 228 (0b11100100) under the
        # default order and 27 (0b00011011) under this one.
        "examples_msb_first": [
            {"seq": "ACGT", "code": 0b00011011, "prov": "derived-msb"},
            {"seq": "A", "code": 0, "prov": "table"}, {"seq": "T", "code": 3, "prov": "table"},
            {"seq": "TA", "code": 12, "prov": "derived-msb"}, {"seq": "AT", "code": 3, "prov": "derived-msb"},
            {"seq": "T" * 32, "code": 2**64 - 1, "prov": "table"}, {"seq": "A" * 32, "code": 0, "prov": "table"},
            {"seq": "acgt", "code": 0b00011011, "prov": "derived-msb"},
            {"seq": "C" + "A" * 31, "code": 1 << 62, "prov": "derived-msb"},
        ],
        "deciding_vector": {"call": "bitnuc::as_2bit(b\"ACGT\")", "lsb_first": 228, "msb_first": 27},
        "invalid": ["ACGN", "ACG ", "ACG\x00", "XCGT", "AC-T", "ACGU"],
        "src": "record.rs:19-27, header.rs:180-185, README.md:42-45",
    }

    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat.json")
    with open(out, "w") as f:
        json.dump(kat, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
