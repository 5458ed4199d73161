#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rates of the stream entry points on one GPU — never bench.py's `value`.

  python tools/e2e.py [--records 1e8] [--dir /tmp] [--gzip] [--verify-cpu]

Stages (BASELINE configs[1] and configs[4] shapes):
  1. generate on device -> Writer::write_batch from device memory -> file        (D2H + write(2))
  2. load_to_device                                                            (pread -> pinned ring -> H2D)
  3. MmapReader::process_device REDUCE / DECODE                                 (page cache -> ring -> H2D || kernel)
  4. --gzip: Reader::from_path(.gz)::process_device DECODE                      (host inflate -> ring -> H2D || kernel)
  5. --verify-cpu: device decode == numpy unpack of the host load_to_vec of the same file (bit-exact, configs[1])
Prints one JSON line per stage."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


from gzutil import bgzf_parallel, gzip_parallel, gzip_single_member  # noqa: E402  (tools/gzutil.py)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=float, default=1e8)
    ap.add_argument("--dir", default="/tmp")
    ap.add_argument("--lens", default="16,12")
    ap.add_argument("--gzip", action="store_true")
    ap.add_argument("--verify-cpu", action="store_true")
    ap.add_argument("--gzip-fast", action="store_true", help="with --gzip: leave out the sequential-zlib legs and the thread sweep (full-size runs)")
    ap.add_argument("--slots", type=int, default=4)
    ap.add_argument("--slot-records", type=int, default=4 << 20)
    ap.add_argument("--feeders", type=int, default=8)
    ap.add_argument("--skip-host-codec", action="store_true", help="leave out the host<->host codec stages (they hold 36 B/record in host arrays)")
    a = ap.parse_args()
    import ibu_amd as ia

    n = int(a.records)
    bc_len, umi_len = (int(x) for x in a.lens.split(","))
    ring = {"slots": a.slots, "slot_records": a.slot_records, "feeder_threads": a.feeders}
    ctx = ia.Context(0)
    path = os.path.join(a.dir, f"ibu_e2e_{os.getpid()}.ibu")
    seed = 0x1B00002
    file_bytes = 32 + 24 * n

    def emit(stage, seconds, st=None, records=None, **kw):
        k = records or n
        rec = {"stage": stage, "records": k, "seconds": round(seconds, 4), "M_records_per_s": round(k / seconds / 1e6, 1),
               "file_GBps": round((32 + 24 * k) / seconds / 1e9, 2)}
        if st is not None:
            rec.update(batches=st.batches, kernel_seconds=round(st.seconds_kernel, 4), h2d=st.bytes_h2d, d2h=st.bytes_d2h)
        rec.update(kw)
        print(json.dumps(rec), flush=True)

    try:
        # 1. device -> file
        d = ctx.alloc(24 * n)
        ctx.generate(seed, 0, n, bc_len, umi_len, d)
        want = ctx.reduce(d, n)
        h = ia.Header(bc_len, umi_len)
        t0 = time.perf_counter()
        w = ia.Writer.from_path(path, h)
        st = w.write_batch_device(ctx, d, n, ring=ring)
        w.finish()
        w.close()
        emit("write_batch_device -> file", time.perf_counter() - t0, st)
        assert os.path.getsize(path) == file_bytes
        d.free()

        # 2. file -> device
        for rep in ("cold-ish", "page-cache"):
            t0 = time.perf_counter()
            hh, dptr, got_n, st = ctx.load_to_device(path, ring=ring)
            dt = time.perf_counter() - t0
            assert got_n == n and ctx.reduce(dptr, n) == want
            ctx.free(dptr)
            emit(f"load_to_device ({rep})", dt, st)

        # 3. mmap -> ring -> device processors
        m = ia.MmapReader.new(path)
        t0 = time.perf_counter()
        res, st = m.process_device(ctx, ia.PROC_REDUCE, ring=ring)
        emit("mmap process_device REDUCE", time.perf_counter() - t0, st)
        assert res == want
        d_bc, d_umi, d_idx = ctx.alloc(n * bc_len), ctx.alloc(n * umi_len), ctx.alloc(n * 8)
        t0 = time.perf_counter()
        _, st = m.process_device(ctx, ia.PROC_DECODE, sink=(d_bc, d_umi, d_idx), ring=ring)
        emit("mmap process_device DECODE", time.perf_counter() - t0, st)
        plain = [d_bc.download().tobytes(), d_umi.download().tobytes(), d_idx.download().tobytes()]
        # 3p. the same through the PULL stream (ibu_stream_*): the caller takes device batches and launches the decode itself
        for rep in ("first", "second"):
            t0 = time.perf_counter()
            with m.device_stream(ctx, ring=ring) as s_:
                for b in s_:
                    ctx.decode_ascii(b.ptr, b.n, bc_len, umi_len, d_bc.ptr + b.first_index * bc_len, d_umi.ptr + b.first_index * umi_len,
                                     d_idx.ptr + b.first_index * 8)
                    b.release()
                ctx.synchronize()
                st = s_.stats()
            emit(f"mmap device_stream pull -> caller-launched DECODE ({rep})", time.perf_counter() - t0, st, numa=ctx.numa())
        assert [d_bc.download().tobytes(), d_umi.download().tobytes(), d_idx.download().tobytes()] == plain
        # 3a. the one-call multi-device form (ibu_mmap_process_devices / _contexts): a host thread + context per listed device.
        # The box has ONE GPU, so the lists repeat ordinal 0 — this measures the call's own overhead (contexts, rings) and
        # whether two workers overlap better on one PCIe link, not multi-GPU scaling (unmeasured).
        for devices in ((0,), (0, 0), (0, 0, 0, 0)):
            t0 = time.perf_counter()
            total, parts, sts = m.process_devices(devices, ia.PROC_REDUCE)   # no ring argument: the one-shot default (3 x 24 MiB)
            dt = time.perf_counter() - t0
            assert total == want and len(parts) == len(devices)
            emit(f"mmap process_devices REDUCE, devices={list(devices)} (contexts created inside the call)", dt,
                 kernel_seconds=round(sum(s_.seconds_kernel for s_ in sts), 4))
        cs = [ia.Context(0) for _ in range(2)]
        for rep in ("first call", "second call"):
            t0 = time.perf_counter()
            total, parts, sts = m.process_devices(proc=ia.PROC_REDUCE, ring=ring, contexts=cs)
            dt = time.perf_counter() - t0
            assert total == want
            emit(f"mmap process_contexts REDUCE, 2 caller-owned contexts on device 0 ({rep})", dt)
        for c_ in cs:
            c_.close()

        # 3b. host <-> host codec pipelines: file -> ASCII in host memory, and back into a file
        if not a.skip_host_codec:
            cring = {"slots": a.slots, "slot_records": 1 << 20, "feeder_threads": a.feeders}
            for rep in ("first call (allocates the ring)", "steady"):
                t0 = time.perf_counter()
                h_bc, h_umi, h_idx, st = m.decode_to_host(ctx, ring=cring)
                emit(f"mmap decode_to_host, {rep}", time.perf_counter() - t0, st)
            assert [h_bc.tobytes(), h_umi.tobytes(), h_idx.tobytes()] == plain
            t0 = time.perf_counter()                         # into the arrays of the last call: pages that exist (a reused buffer)
            _, _, _, st = m.decode_to_host(ctx, ring=cring, out=(h_bc, h_umi, h_idx))
            emit("mmap decode_to_host, into arrays written before (no page faults)", time.perf_counter() - t0, st)
            assert [h_bc.tobytes(), h_umi.tobytes(), h_idx.tobytes()] == plain
            # the same call on a hundredth and a tenth of the file (one shard of the static split): the rate at the sizes where a
            # caller might hope the GPU pays for host -> host decoding.  It does not at any size: see include/ibu_hip.h
            for parts in (100, 10):
                if n // parts >= 1000:
                    t0 = time.perf_counter()
                    _, _, _, st = m.decode_to_host(ctx, shard=0, n_shards=parts, ring=cring)
                    emit(f"mmap decode_to_host, shard 0 of {parts}", time.perf_counter() - t0, st, records=st.records)
            back = path + ".back"
            t0 = time.perf_counter()
            w = ia.Writer.from_path(back, m.header())
            st = w.write_ascii_batch(ctx, h_bc, h_umi, bc_len, umi_len, index=h_idx, ring=cring)
            w.finish()
            w.close()
            emit("host ASCII -> write_ascii_batch -> file", time.perf_counter() - t0, st)
            with open(path, "rb") as f1, open(back, "rb") as f2:
                same = all(f1.read(1 << 26) == f2.read(1 << 26) for _ in range(file_bytes // (1 << 26) + 1))
            os.unlink(back)
            assert same, "re-encoded file differs"
            del h_bc, h_umi, h_idx
        m.close()

        # 5. configs[1]: device decode bit-exact vs the HOST load_to_vec of the same file, unpacked here with numpy
        #    (an independent statement of record.rs:19-27: base i = bits [2i, 2i+1], A C G T = 0 1 2 3)
        if a.verify_cpu:
            t0 = time.perf_counter()
            _, recs = ia.load_to_vec(path)
            t1 = time.perf_counter()
            lut = np.frombuffer(b"ACGT", dtype=np.uint8)
            ok = recs["index"].tobytes() == plain[2]
            step = 5_000_000
            for lo in range(0, n, step):
                hi = min(n, lo + step)
                for col, ln, got in ((recs["barcode"], bc_len, plain[0]), (recs["umi"], umi_len, plain[1])):
                    sh = (2 * np.arange(ln, dtype=np.uint64))[None, :]
                    asc = lut[((col[lo:hi, None] >> sh) & np.uint64(3)).astype(np.intp)]
                    ok = ok and asc.tobytes() == got[lo * ln:hi * ln]
            t2 = time.perf_counter()
            emit("host load_to_vec + numpy unpack (checker)", t2 - t0, None, load_seconds=round(t1 - t0, 3),
                 unpack_seconds=round(t2 - t1, 3), device_output_bit_exact=bool(ok))
            assert ok, "device decode differs from the host load_to_vec + numpy unpack"
            del recs

        # 3b. the same plain file through the STREAMING reader (Reader::from_path -> read(2) / pread into the ring)
        r = ia.Reader.from_path(path)
        t0 = time.perf_counter()
        _, st = r.process_device(ctx, ia.PROC_DECODE, sink=(d_bc, d_umi, d_idx), ring=ring)
        dt = time.perf_counter() - t0
        r.close()
        same = [d_bc.download().tobytes(), d_umi.download().tobytes(), d_idx.download().tobytes()] == plain
        emit("plain file, streaming Reader process_device DECODE", dt, st, equals_plain_path=same)
        assert same

        # 4. gzip stream
        if a.gzip:
            def gz_leg(label, gzpath, gz_bytes, tc, **env):
                saved = {k: os.environ.get(k) for k in env}
                for k, v in env.items():
                    os.environ[k] = str(v)          # read by the library when the reader is opened
                try:
                    r = ia.Reader.from_path(gzpath)
                    t0 = time.perf_counter()
                    _, st = r.process_device(ctx, ia.PROC_DECODE, sink=(d_bc, d_umi, d_idx), ring=ring)
                    dt = time.perf_counter() - t0
                    r.close()
                finally:
                    for k, v in saved.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v
                same = [d_bc.download().tobytes(), d_umi.download().tobytes(), d_idx.download().tobytes()] == plain
                emit(label, dt, st, gz_bytes=gz_bytes, gz_ratio=round(gz_bytes / file_bytes, 3), compress_seconds=round(tc, 2),
                     equals_plain_path=same, **{k.lower(): v for k, v in env.items()})
                assert same

            gz = path + ".gz"
            t0 = time.perf_counter()
            gz_bytes = gzip_parallel(path, gz)
            tc = time.perf_counter() - t0
            if not a.gzip_fast:
                gz_leg("gzip (64 MiB members) Reader process_device DECODE, sequential zlib", gz, gz_bytes, tc, IBU_NO_PARALLEL_GZIP=1)
            gz_leg("gzip (64 MiB members) Reader process_device DECODE, parallel inflate", gz, gz_bytes, tc)
            os.unlink(gz)
            # configs[4] literally: ONE gzip member (one deflate stream) — nothing in the container says where to start
            t0 = time.perf_counter()
            gz_bytes = gzip_single_member(path, gz)
            tc = time.perf_counter() - t0
            if not a.gzip_fast:
                gz_leg("gzip (one member) Reader process_device DECODE, sequential zlib", gz, gz_bytes, tc, IBU_NO_PARALLEL_GZIP=1)
            for th in (() if a.gzip_fast else (2, 4, 8, 16, 32)):
                gz_leg("gzip (one member) Reader process_device DECODE, parallel inflate", gz, gz_bytes, tc, IBU_PGZ_THREADS=th)
            gz_leg("gzip (one member) Reader process_device DECODE, parallel inflate", gz, gz_bytes, tc)
            os.unlink(gz)
            # the same records bgzip'd: block boundaries are known without inflating -> parallel inflate on the host
            bgz = path + ".bgz"
            t0 = time.perf_counter()
            bgz_bytes = bgzf_parallel(path, bgz)
            tc = time.perf_counter() - t0
            r = ia.Reader.from_path(bgz)
            t0 = time.perf_counter()
            _, st = r.process_device(ctx, ia.PROC_DECODE, sink=(d_bc, d_umi, d_idx), ring=ring)
            dt = time.perf_counter() - t0
            r.close()
            same = [d_bc.download().tobytes(), d_umi.download().tobytes(), d_idx.download().tobytes()] == plain
            emit("BGZF Reader process_device DECODE (parallel inflate)", dt, st, gz_bytes=bgz_bytes,
                 gz_ratio=round(bgz_bytes / file_bytes, 3), compress_seconds=round(tc, 2), equals_plain_path=same)
            assert same
            os.unlink(bgz)
    finally:
        if os.path.exists(path):
            os.unlink(path)
        ctx.close()


if __name__ == "__main__":
    main()
