#!/usr/bin/env python3
"""bench.py — records/s of the IBU hot path (fused 2-bit decode + encode of 24-byte records)
on N MI355X GPUs, with the roofline of the dominant kernel and a CPU baseline beside it.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1
the driver launches one rank per GPU with torch.distributed.run.  A *step* is one pass of the
hot path over the rank's shard: K2 decode (AoS records -> barcode ASCII + UMI ASCII + index
column) followed by K3 encode (those columns -> AoS records).  Inputs are resident in HBM
before the timed region (synthetic, generated on device by the counter-based generator of
SURVEY §8d, so every rank materialises its own contiguous record range of the global stream —
the static split of src/io/mmap.rs:297-307).  No collective on the data path; ONE all-reduce
at the end carries the global record count and field sums (BASELINE north_star).

Scaling modes (`--scaling`): BASELINE configs[3] is ONE stream of 1e9 records range-sharded over the GPUs of a
node, so for N > 1 the default is **strong** scaling: `--records` is the GLOBAL record count and rank r owns
`rank_shard(records, N, r)` (per = records / N, remainder to the last rank).  `--scaling weak` keeps `--records`
records on every GPU instead (global = records x N).  At N = 1 the two coincide.  The JSON line names the mode, the
per-rank shard sizes, the slowest and fastest rank's kernel times and the time of the one cross-GPU exchange.

On one GPU a short second leg runs BASELINE configs[2] (bc_len = umi_len = 32, the maximum width) and is reported
under `config2_32_32`; it is outside the timed region of the headline value.

Output: one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=None,
                   help="ranks = GPUs of this node.  Under torch.distributed.run it must equal WORLD_SIZE (anything else is an error); "
                        "run bare with N > 1, bench.py starts the N ranks itself as a child torch.distributed.run and returns its "
                        "exit code.  Default: WORLD_SIZE, or 1")
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--records", type=float, default=1e9,
                   help="strong scaling: records in the whole stream; weak scaling: records per GPU")
    p.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                   help="strong (default): --records is the global count, range-sharded over the ranks (configs[3]); "
                        "weak: --records per GPU")
    p.add_argument("--no-wide-leg", action="store_true", help="skip the (32,32) leg (configs[2]) on one GPU")
    p.add_argument("--wide-records", type=float, default=0, help="records of the (32,32) leg (0 = same as --records)")
    p.add_argument("--no-sort-leg", action="store_true", help="skip the sort + per-barcode aggregation leg on one GPU")
    p.add_argument("--no-e2e-leg", action="store_true", help="skip the file -> result (PCIe-inclusive) leg on one GPU")
    p.add_argument("--e2e-records", type=float, default=1e8, help="records of the e2e leg's file (24 B each)")
    p.add_argument("--bgzf-large-records", type=float, default=5e8,
                   help="records of the e2e leg's LARGE BGZF file (ibu_load_bgzf_to_device with its decoder launched ahead of the copies, beside "
                        "the plain file of the same records; 12 + 6 GB in the e2e directory, ~20 s); 0 = skip")
    p.add_argument("--e2e-dir", default=None, help="where the e2e leg writes its files (default: the system temp dir)")
    p.add_argument("--bc-len", type=int, default=16)
    p.add_argument("--umi-len", type=int, default=12)
    p.add_argument("--seed", type=lambda s: int(s, 0), default=0x1B00003)
    p.add_argument("--cpu-sample", type=float, default=0, help="records for the CPU baseline (0 = auto)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-verify", action="store_true")
    p.add_argument("--placement-tries", type=int, default=0,
                   help="candidates per resident array for the library's placement probing (ibu_device_alloc_probed: all "
                        "candidates of an array held at once, a write and a read streamed over each, the fastest kept, the "
                        "others freed) BEFORE the timed region; every probe is printed.  0 (default) = as many as fit, at most "
                        "16: 3-5 for the arrays of 1e9 records on one GPU, 16 for the shards of an 8-GPU run.  1 = no bench-side "
                        "probing: on one GPU the arrays then come from plain ibu_device_alloc calls under the LIBRARY's default "
                        "option (alloc_probe_tries = auto since round 5), on several GPUs from plain hipMalloc.  Every run also "
                        "reports value_first_placement (plain hipMalloc) and, on one GPU, value_library_default_placement.  "
                        "See place_leg()")
    # rehearsal of the N>1 code path on a box with fewer GPUs than ranks (never used by the driver):
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: collectives on CPU tensors")
    p.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (with --backend gloo)")
    p.add_argument("--force-dist", action="store_true",
                   help="initialise the process group and run every collective even with one rank (a one-GPU box can then "
                        "execute the RCCL code path of the N > 1 runs: init, barrier, all_reduce, all_gather)")
    return p.parse_args(argv)


def resolve_world(args, argv, env):
    """What `--gpus N` means.  Returns ("run", world) when this process is one of `world` ranks (or the only one), or
    ("spawn", cmd) when it was started bare with N > 1 and has to start the ranks itself — as a CHILD process, before
    anything touches the GPU (never an exec).  A `--gpus` that contradicts WORLD_SIZE is an error, not a silent
    one-GPU run that prints n_gpus: 1."""
    ws = env.get("WORLD_SIZE")
    if ws is not None:
        world = int(ws)
        if args.gpus is not None and args.gpus != world:
            raise SystemExit(f"bench.py: --gpus {args.gpus} contradicts WORLD_SIZE={world} set by the launcher; "
                             f"start it as `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...` "
                             f"or run `python bench.py --gpus {args.gpus}` bare")
        return "run", world
    n = args.gpus or 1
    if n < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if n == 1:
        return "run", 1
    import socket
    with socket.socket() as sk:                      # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return "spawn", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                     "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def usable_cores():
    """CPUs this job may really use: the affinity mask limited by the cgroup CPU quota (a 1-GPU box of this pool
    shows 256 CPUs in the mask but is throttled to its share; 256 runnable threads under that quota thrash)."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return n


class LegCheckFailed(Exception):
    """A correctness check of a side leg (sort, e2e, rehearsal) failed.  Recorded in the line as that leg's `error` — the
    headline stands on its own and is still printed — and turned into a non-zero exit status after the line is out."""


def cpu_baseline(args):
    """The oracle (a C restatement of the reference's CPU path: static range split over OS threads,
    1Mi-record batches, scalar 2-bit codec) timed on this box's host cores on a bounded sample."""
    from oracle import oracle as orc  # the checker, used here only as the reported baseline

    def cpu_phases(args, quota):
        """BASELINE.md §2's CPU phases — the reference's two example programs — from the oracle's restatement, each timed on its own on a
        bounded sample (about 1-3 s in all): the write_record loop (examples/roundtrip.rs:33-49), the streaming Reader with its XOR
        checksum (:80-100), load_to_vec (:122-131), MmapReader::process_parallel summing the fields (examples/parallel.rs:93-105) on 1
        thread and on the quota's worth, and the scalar 2-bit decode+encode on 1 thread (the headline baseline is the same on `threads`)."""
        import tempfile

        from oracle import oracle as orc  # the checker, used here only as the reported baseline

        n = 20_000_000
        path = os.path.join(args.e2e_dir or tempfile.gettempdir(), f"ibu_bench_cpu_{os.getpid()}.ibu")
        try:
            sec, xor, sums = orc.bench_phases(path, n, quota)
        finally:
            if os.path.exists(path):
                os.unlink(path)
        want = [499_999_500_000 * (n // 1_000_000), None, n * (n - 1) // 2]   # barcode = i % 1e6, index = i
        ok = sums[0] == want[0] and sums[2] == want[2] % 2**64
        n1 = 4_000_000
        t1, chk = orc.bench_decode_encode(n1, args.bc_len, args.umi_len, args.seed, 1)
        rate = lambda s_: {"seconds": s_, "records_per_s": n / s_, "GBps_of_file": 24 * n / s_ / 1e9}
        return {
            "sample": f"{n} records (16,12), the file in {os.path.dirname(path)} (page cache); codec leg {n1} records",
            "write_record_loop": dict(rate(sec["write_record"]), threads=1, ref="examples/roundtrip.rs:33-49"),
            "streaming_reader_xor": dict(rate(sec["reader_xor"]), threads=1, xor_checksum=xor, ref="examples/roundtrip.rs:80-100"),
            "load_to_vec": dict(rate(sec["load_to_vec"]), threads=1, ref="examples/roundtrip.rs:122-131"),
            "process_parallel_sum_T1": dict(rate(sec["process_parallel_1"]), threads=1, ref="examples/parallel.rs:93-105"),
            "process_parallel_sum_Tquota": dict(rate(sec["process_parallel_T"]), threads=quota, field_sums_as_expected=bool(ok)),
            "decode_encode_T1": {"seconds": t1, "records_per_s": n1 / t1, "threads": 1, "roundtrip_exact": chk != 2**64 - 1,
                                 "what": "scalar 2-bit unpack + pack of every record (the headline baseline on one thread)"},
        }

    quota = usable_cores()
    n = int(args.cpu_sample) or 100_000_000
    # the strongest baseline this box gives: the CPU quota's worth of threads, or twice that (under a cgroup quota the
    # second thread per CPU still pays: the box's 16-CPU quota gives 650 M records/s at 16 threads, 810 at 32, less at 64)
    best = None
    for threads in sorted({quota, min(2 * quota, len(os.sched_getaffinity(0)))}):
        t1, chk = orc.bench_decode_encode(min(n, 8_000_000), args.bc_len, args.umi_len, args.seed, threads)  # warm-up + rate probe
        rate = min(n, 8_000_000) / max(t1, 1e-6)
        if best is None or rate > best[1]:
            best = (threads, rate)
    threads, rate = best
    # about 1.5 s of wall time (16 CPUs -> ~25 CPU-seconds): repeat the pass over the sample
    reps = max(1, int(rate * 1.5 / n))
    t, chk = orc.bench_decode_encode(n, args.bc_len, args.umi_len, args.seed, threads, reps)
    assert chk != 2**64 - 1, "oracle round trip failed"
    n_total = n * reps
    try:
        phases = cpu_phases(args, quota)
    except Exception as e:  # the baseline number stands on its own
        phases = {"error": f"{type(e).__name__}: {e}"}
    return {
        "value": n_total / t, "unit": "records/s",
        # `cores` is the contract's name for the THREADS the baseline ran on (it may exceed the CPUs the cgroup grants: under a
        # quota the second thread per CPU still pays); what the box really gives is beside it
        "cores": threads, "threads": threads, "cpu_quota": quota, "cores_visible": len(os.sched_getaffinity(0)),
        "kind": "port",
        "sample": f"{reps} pass(es) over {n} records bc_len={args.bc_len} umi_len={args.umi_len}, decode+encode, static split over "
                  f"{threads} OS threads on a {quota}-CPU quota (C restatement of the reference's std::thread path; the reference is Rust "
                  f"and cannot be built here)",
        "seconds": t, "cpu_seconds": t * min(threads, quota),
        "phases": phases,
    }


def plan_shard(records, scaling, world, rank):
    """(n_global, first, n) of `rank`: strong = ONE stream of `records` split by the reference's static split
    (src/io/mmap.rs:297-307: per = len / n, the remainder to the LAST shard); weak = `records` on every rank."""
    from ibu_amd import sharding

    n_global = int(records) * (world if scaling == "weak" else 1)
    first, end = sharding.rank_shard(n_global, world, rank)
    return n_global, first, end - first


def traffic_from_profile(bc_len, umi_len, n):
    """PMC-derived HBM bytes of one decode launch.  The counters cannot be read from inside this process (rocprofv3
    collects them around it, in separate --pmc passes as the guide prescribes), so the figure comes from the file
    `tools/pmc_traffic.py` wrote from those passes; the line names the file, its round tag and its hash."""
    import hashlib

    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(tp, "rb") as f:
            raw = f.read()
        doc = json.loads(raw)
        rec = doc.get(f"decode_{bc_len}_{umi_len}")
        if not rec:
            return None, None
        src = {"file": "profiles/pmc_traffic.json", "round": doc.get("_round"), "sha16": hashlib.sha256(raw).hexdigest()[:16],
               "hbm_bytes_per_record": rec["hbm_bytes_per_record"], "measured_at_records": doc.get("_records"),
               "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/kbench.py, FETCH_SIZE doubled (gfx950)"}
        return rec["hbm_bytes_per_record"] * n, src
    except (OSError, ValueError, KeyError):
        return None, None


def place_leg(make, tries, set_bytes, torch, dev, sharers=1, ctx=None, library_default=False):
    """Placement probing — since round 3 a LIBRARY call, not bench-side logic.  On this part the rate of a read+write
    streaming kernel depends on WHERE the driver put the arrays' physical pages: the same kernel on the same GPU runs
    9.4 ... 11.3 ms from one allocation to the next (profiles/README.md: pool survey, r02_placement_pmc, r02_ag — physical
    regions in buddy-block runs, L2 tag-pipeline stalls, not translation), and an allocation keeps its speed for as long
    as it lives.  A job that keeps its shard resident can therefore choose, and any caller of the C ABI can do so with
    `ibu_device_alloc_probed(ctx, bytes, tries, &ptr, &report)`: up to `tries` candidates of ONE array held at once, a write
    and a read streamed over each, the fastest kept, the rest freed.

    What this function does around it: (1) a first set of arrays from plain `ibu_device_alloc` is timed with one
    decode + encode — what a caller that does not probe gets, reported as `first_placement_*` and at top level as
    `value_first_placement`; (1b, one GPU) the same from plain `ibu_device_alloc` calls under the library's DEFAULT option (auto
    probing: `value_library_default_placement`); (2) unless --placement-tries 1, every array is allocated again through the probing
    call with as many candidates as fit, and the faster of that set and the library-default set (same decode + encode probe) is
    what the timed region runs on.  All of it happens before the timed region; every candidate's time is in the line (`placement`)."""
    free_b, _ = torch.cuda.mem_get_info(dev)
    if ctx is not None:
        ctx.set_option("alloc_probe_tries", 1)           # plain hipMalloc: what the library did by default until round 4
    first = make(1)
    if ctx is not None:
        ctx.set_option("alloc_probe_tries", 0)           # back to the library's default (auto)
    first_probe = first.probe()
    bytes_per_launch = first.n * (24 + first.bc_len + first.umi_len + 8)
    info = {"tries": 1, "first_placement_probe_ms_decode_encode": [round(v, 3) for v in first_probe],
            "first_placement_decode_frac": round(bytes_per_launch / (first_probe[0] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "first_placement_records_per_s": first.n / ((first_probe[0] + first_probe[1]) * 1e-3)}
    if library_default:
        # (1b) what a caller of ibu_device_alloc gets under the library's DEFAULT option (round 5: "alloc_probe_tries" = 0 = auto —
        # arrays of 1 GiB and more draw up to four candidates when three fit): the same five arrays from plain ctx.alloc calls
        first.free()
        t0 = time.perf_counter()
        try:
            first = make(1)
        except Exception as e:                           # the headline must not die on a side measurement: plain arrays again
            info["library_default_error"] = f"{type(e).__name__}: {e}"
            ctx.set_option("alloc_probe_tries", 1)
            first = make(1)
            ctx.set_option("alloc_probe_tries", 0)
        alloc_s = time.perf_counter() - t0
        dprobe = first.probe()
        info.update(library_default_probe_ms_decode_encode=[round(v, 3) for v in dprobe],
                    library_default_decode_frac=round(bytes_per_launch / (dprobe[0] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    library_default_records_per_s=first.n / ((dprobe[0] + dprobe[1]) * 1e-3),
                    library_default_alloc_seconds=round(alloc_s, 3),
                    library_default_option='alloc_probe_tries = 0 (auto): ibu_device_alloc of >= 1 GiB draws up to 4 candidates when 3 fit')
    # 0 = auto: as many candidates as fit beside the arrays already chosen, at most 16.  The largest array (24 B/record)
    # sets the bound: the other arrays of the set stay allocated while it is probed.
    biggest = max(24 * first.n, 1)
    # On one GPU the library-default set STAYS while the probed set is built (two sets resident: 168 GB of the 288 at 1e9 records),
    # and the faster of the two — by the same decode + encode probe, before the timed region — is the one the headline runs on: the
    # per-array probe (a write and a read over each array alone) removes bad draws but does not rank good ones (profiles/README.md
    # r05_h: in r05_fin2 it kept a set that ran 0.786 where the default set ran 0.801).  Everything is in the line.
    hold = first if library_default else None
    resident = set_bytes * (2 if hold is not None else 1)
    fit = int((free_b * 0.94 / max(sharers, 1) - resident) // biggest) + 1
    tries = max(1, min(tries or 16, fit, 16))
    if tries == 1:
        return first, info
    if hold is None:
        first.free()
    first = None
    leg = make(tries)
    probe = leg.probe()
    info.update(tries=tries, library_call="ibu_device_alloc_probed(ctx, bytes, tries, &ptr, &report) per array",
                per_array=leg.reports, probed_set_probe_ms_decode_encode=[round(v, 3) for v in probe])
    if hold is not None:
        dprobe = info["library_default_probe_ms_decode_encode"]
        if sum(dprobe) <= sum(probe):                    # the library-default set is at least as fast: keep it, free the probed one
            leg.free()
            leg, probe = hold, hold.probe()
            info["kept"] = "the library-default set (it probed at least as fast as the bench-probed set)"
        else:
            hold.free()
            info["kept"] = "the bench-probed set"
    info["kept_probe_ms_decode_encode"] = [round(v, 3) for v in probe]
    return leg, info


def sort_and_aggregate(ctx, n, bc_len, umi_len, seed, rounds=3, torch=None, dev=None):
    """Device sort by (barcode, umi, index) (record.rs:58 ordering, header.rs:111-113 sorted flag) of the headline's
    synthetic records, in read order (index increasing: the index passes are skipped) and with the index column replaced
    by random 30-bit values, then the per-barcode aggregation (parallel.rs:72-98) of the sorted records.  Third input: what
    a single-cell run looks like — barcodes drawn from a whitelist of 100 000 with a skewed distribution (rank ~ K u^3), in
    read order: such keys do not spread, every varying key byte gets its pass.  Checked, not trusted: sorted, and count /
    wrapping sums / XORs unchanged."""
    import statistics
    import ctypes as C

    from ibu_amd import _check, _dptr, lib

    d, t = ctx.alloc(24 * n), ctx.alloc(24 * n)
    cols = [ctx.alloc(8 * n) for _ in range(4)]
    out = {"workload": f"{n:.3g} records bc_len={bc_len} umi_len={umi_len}: ibu_sort_records on resident records, median of {rounds}"}
    try:
        names = ["read_order", "random_index"]
        if torch is not None and 2 * bc_len <= 62:
            names.append("whitelist_read_order")

        def whitelist_column(col):                             # barcodes for every record, drawn from kwl distinct ones
            kwl = 100_000
            g = torch.Generator(device=dev).manual_seed(seed + 2)
            wl = torch.randint(0, 1 << (2 * bc_len), (kwl,), generator=g, device=dev, dtype=torch.int64)
            bcv = torch.as_tensor(col, device=dev).view(torch.int64)
            for lo in range(0, n, 1 << 26):                    # in pieces: the temporaries stay small
                hi = min(n, lo + (1 << 26))
                u = torch.rand(hi - lo, generator=g, device=dev, dtype=torch.float64)
                bcv[lo:hi] = wl[(u * u * u * kwl).to(torch.int64).clamp_(max=kwl - 1)]
            torch.cuda.synchronize()

        for name in names:
            ts = []
            if name == "whitelist_read_order":
                whitelist_column(cols[0])
            for _ in range(rounds + 1):
                ctx.generate(seed, 0, n, bc_len, umi_len, d)
                if name == "whitelist_read_order":             # the whitelist barcodes replace the generator's
                    ctx.deserialize(d, n, cols[1], cols[2], cols[3])
                    ctx.serialize(cols[0], cols[2], cols[3], n, d)
                if name == "random_index":     # another stream's 15-base barcode column as the index column
                    ctx.deserialize(d, n, cols[0], cols[1], cols[2])
                    ctx.generate(seed + 1, 0, n, 15, 1, t)
                    ctx.deserialize(t, n, cols[3], cols[2], cols[2])
                    ctx.serialize(cols[0], cols[1], cols[3], n, d)
                before = ctx.reduce(d, n)
                ctx.synchronize()
                t0 = time.perf_counter()
                ctx.sort_records(d, t, n)
                ctx.synchronize()
                ts.append(time.perf_counter() - t0)
            sec = statistics.median(ts[1:])
            ok = bool(ctx.is_sorted(d, n)) and ctx.reduce(d, n) == before
            out[name] = {"seconds": sec, "records_per_s": n / sec, "sorted_and_multiset_preserved": ok}
            if name == "whitelist_read_order":
                out[name]["input"] = "barcodes from 100000 distinct, rank ~ K u^3; random UMIs; index increasing"
            if not ok:
                raise LegCheckFailed("device sort: result not sorted or records changed")
        # aggregation of the sorted records: size query (count pass + scan), then the whole call into device arrays
        nb, npairs = C.c_size_t(), C.c_size_t()
        _check(lib.ibu_barcode_counts(ctx._c, _dptr(d), n, None, None, None, 0, C.byref(nb), C.byref(npairs), None))
        u = max(nb.value, 1)
        outs = [ctx.alloc(8 * u) for _ in range(3)]
        ts = []
        for _ in range(rounds + 1):
            t0 = time.perf_counter()
            _check(lib.ibu_barcode_counts(ctx._c, _dptr(d), n, _dptr(outs[0]), _dptr(outs[1]), _dptr(outs[2]), u, C.byref(nb),
                                          C.byref(npairs), None))
            ctx.synchronize()
            ts.append(time.perf_counter() - t0)
        # the check runs on the device at every size (round 2 summed the column on the host and skipped it above 2^27
        # barcodes, i.e. at the driver's size): (barcode, count, unique) rows are serialised into the free scratch array and
        # reduced with K4 — sum of the counts column == n, sum of the distinct-UMI column == the pair count, and
        # barcodes <= pairs <= n
        ctx.serialize(outs[0], outs[1], outs[2], nb.value, t)
        agg = ctx.reduce(t, nb.value)
        adds_up = (agg["count"] == nb.value and agg["sum"][1] == n and agg["sum"][2] == npairs.value and nb.value <= npairs.value <= n)
        out["barcode_counts"] = {"seconds": statistics.median(ts[1:]), "distinct_barcodes": nb.value, "barcode_umi_pairs": npairs.value,
                                 "counts_add_up": bool(adds_up), "checked": "device-side K4 reduce of the (barcode, count, unique) columns"}
        if not adds_up:
            raise LegCheckFailed("ibu_barcode_counts: the counts column does not add up to the record count")
        for b in outs:
            b.free()
    finally:
        for b in [d, t] + cols:
            b.free()
    return out


def e2e_leg(ctx, ia, n, bc_len, umi_len, seed, tmpdir, torch=None):
    """File -> result rates on this box, PCIe included — reported beside the headline, NEVER part of `value` (BASELINE.md §3
    asks for the kernel-resident and the file -> result rates separately; configs[4] is the gzip form).  A file of `n`
    synthetic records is written from device memory (`Writer::write_batch` of a device slice), then read back three ways:
      load_to_device        reader.rs:510-535   pread -> pinned ring -> H2D
      mmap -> DECODE        mmap.rs:286-332     ibu_mmap_process_devices over EVERY visible device (one call, a host thread +
                                                context per device, shard i of the static split on device i): map -> ring -> H2D || K2
      gzip -> DECODE        reader.rs:345-352   the same file as ONE gzip member (level 1) -> parallel host inflate -> ring -> H2D || K2
    Every result is checked against K4 (count, wrapping sums, XORs) of the resident copy the file was written from: the
    loaded records directly, the decoded columns after re-encoding them on the device."""
    import tempfile

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gzutil import gzip_single_member   # test-data writer (pigz-style parallel deflate into one member)

    t_leg = time.perf_counter()
    n = int(n)
    path = os.path.join(tmpdir or tempfile.gettempdir(), f"ibu_bench_e2e_{os.getpid()}.ibu")
    file_bytes = 32 + 24 * n
    ring = {"slots": 4, "slot_records": 4 << 20, "feeder_threads": 8}
    out = {"workload": f"{n:.3g} records bc_len={bc_len} umi_len={umi_len}, a {file_bytes / 1e9:.2f} GB file in {os.path.dirname(path)}; "
                       f"wall-clock rates including PCIe and host work; never part of `value`"}

    def rate(seconds, st=None, **kw):
        r = {"seconds": seconds, "GBps_of_file": file_bytes / seconds / 1e9, "records_per_s": n / seconds}
        if st is not None:
            ks = st if isinstance(st, float) else st.seconds_kernel
            r.update(kernel_seconds=ks, kernel_share_of_wall=ks / seconds)
        r.update(kw)
        return r

    def wadd(a, b):
        return {"count": a["count"] + b["count"], "sum": [(x + y) & (2**64 - 1) for x, y in zip(a["sum"], b["sum"])],
                "xor": [x ^ y for x, y in zip(a["xor"], b["xor"])]}

    gz = path + ".gz"
    ctxs = []
    try:
        d = ctx.alloc(24 * n)
        ctx.generate(seed, 0, n, bc_len, umi_len, d)
        want = ctx.reduce(d, n)
        t0 = time.perf_counter()
        w = ia.Writer.from_path(path, ia.Header(bc_len, umi_len))
        st = w.write_batch_device(ctx, d, n, ring=ring)
        w.finish()
        w.close()
        out["write_batch_device_to_file"] = rate(time.perf_counter() - t0)
        d.free()
        assert os.path.getsize(path) == file_bytes

        dst = ctx.alloc(24 * n)                          # the caller's buffer: the transfer alone (allocation is timed on its own below)
        calls = []
        for rep in range(3):                             # (the pinned ring exists already: write_batch_device above allocated it)
            t0 = time.perf_counter()
            _, dptr, got_n, st = ctx.load_to_device(path, ring=ring, d_records=dst, cap_records=n)
            calls.append(time.perf_counter() - t0)
            if not (got_n == n and ctx.reduce(dptr, n) == want):
                raise LegCheckFailed("e2e: load_to_device returned other records than were written")
        dst.free()
        # the fastest of three calls, all three in the line: right after the legs above freed ~200 GB a call has been seen to take
        # twice as long as its neighbours (0.046 / 0.098 s in profiles r05_r) — presumably the driver clearing the freed VRAM with the
        # DMA engines the H2D copies use; the mmap leg further down, a second later, runs at the link's rate again
        out["load_to_device"] = rate(min(calls), st, calls_seconds=[round(c_, 4) for c_ in calls], totals_equal_resident_copy=True,
                                     destination="a caller-owned device buffer (no allocation inside the call)")
        # the same call with the destination allocated by the LIBRARY (placement-probed under the default option from 1 GiB on: up to
        # four candidates are allocated, a write + read streamed over each, the fastest kept), then one decode of the records where the
        # library put them into columns from ibu_device_alloc: what a file-based caller that does nothing about placement sees
        lib_calls = []
        for rep in range(2):
            t0 = time.perf_counter()
            _, dptr, got_n, st = ctx.load_to_device(path, ring=ring)
            lib_calls.append(time.perf_counter() - t0)
            if not (got_n == n and ctx.reduce(dptr, n) == want):
                raise LegCheckFailed("e2e: load_to_device (library-allocated destination) returned other records than were written")
            if rep == 0:
                ctx.free(dptr)
        lib = rate(min(lib_calls), st, calls_seconds=[round(c_, 4) for c_ in lib_calls], totals_equal_resident_copy=True,
                   destination="allocated inside the call, option alloc_probe_tries = 0 (auto); allocating right after large frees "
                               "waits for the driver to hand out cleared VRAM: profiles/r05_h_alloc_probe_cost.jsonl")
        if torch is not None:
            cols = [ctx.alloc(n * bc_len), ctx.alloc(n * umi_len), ctx.alloc(n * 8)]
            ms = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.ExternalStream(ctx.stream))
                ctx.decode_ascii(dptr, n, bc_len, umi_len, *cols)
                e1.record(torch.cuda.ExternalStream(ctx.stream))
                e1.synchronize()
                ms.append(e0.elapsed_time(e1))
            for b in cols:
                b.free()
            gbps = n * (24 + bc_len + umi_len + 8) / (min(ms[1:]) * 1e-3) / 1e9
            lib["then_decode"] = {"decode_ms_on_the_loaded_records": min(ms[1:]), "decode_GBps": gbps, "decode_frac_of_peak": gbps / HBM_PEAK_GBPS}
        ctx.free(dptr)
        out["load_to_device_library_allocated"] = lib

        m = ia.MmapReader.new(path)
        ndev = ia.device_count()
        ctxs = [ia.Context(i) for i in range(ndev)]      # caller-owned contexts: their rings survive from call to call
        shards = [ia.shard_range(n, ndev, i) for i in range(ndev)]
        sinks = []
        for c, (a0, a1) in zip(ctxs, shards):
            k = max(a1 - a0, 1)
            sinks.append((c.alloc(k * bc_len), c.alloc(k * umi_len), c.alloc(k * 8), k))
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            count, _, sts = m.process_devices(proc=ia.PROC_DECODE, sinks=sinks, ring=ring, contexts=ctxs)
            dt = time.perf_counter() - t0
            best = rate(dt, max(s_.seconds_kernel for s_ in sts), first_call_seconds=best["seconds"] if best else None,
                        devices=ndev, call="ibu_mmap_process_contexts(m, ctxs, n_devices, ring, IBU_PROC_DECODE, sinks, &total, stats)")
        got = {"count": 0, "sum": [0, 0, 0], "xor": [0, 0, 0]}
        for c, (a0, a1), (s_bc, s_umi, s_idx, _) in zip(ctxs, shards, sinks):
            if a1 > a0:                                  # the decoded columns, re-encoded where they live, reduced with K4
                back = c.alloc(24 * (a1 - a0))
                c.encode_ascii(s_bc, s_umi, s_idx, a1 - a0, bc_len, umi_len, back)
                c.codec_status()
                got = wadd(got, c.reduce(back, a1 - a0))
                back.free()
        if count != n or got != want:
            raise LegCheckFailed("e2e: mmap -> DECODE over the devices does not reproduce the records written")
        best["totals_equal_resident_copy"] = True
        out["mmap_process_devices_decode"] = best
        m.close()

        t0 = time.perf_counter()
        gz_bytes = gzip_single_member(path, gz, level=1, workers=max(2, min(16, usable_cores())))
        tc = time.perf_counter() - t0
        s_bc, s_umi, s_idx, cap = sinks[0]
        if cap < n:                                      # several devices: device 0's sink held one shard only
            for b in (s_bc, s_umi, s_idx):
                b.free()
            s_bc, s_umi, s_idx = ctxs[0].alloc(n * bc_len), ctxs[0].alloc(n * umi_len), ctxs[0].alloc(n * 8)
            sinks[0] = (s_bc, s_umi, s_idx, n)
        r = ia.Reader.from_path(gz)
        t0 = time.perf_counter()
        _, st = r.process_device(ctxs[0], ia.PROC_DECODE, sink=(s_bc, s_umi, s_idx, n), ring=ring)
        dt = time.perf_counter() - t0
        r.close()
        back = ctxs[0].alloc(24 * n)
        ctxs[0].encode_ascii(s_bc, s_umi, s_idx, n, bc_len, umi_len, back)
        ctxs[0].codec_status()
        ok = ctxs[0].reduce(back, n) == want
        back.free()
        if not ok:
            raise LegCheckFailed("e2e: gzip -> DECODE does not reproduce the records written")
        out["gzip_reader_process_device_decode"] = rate(dt, st, gz_bytes=gz_bytes, gz_ratio=gz_bytes / file_bytes, compress_seconds=tc,
                                                        input="ONE gzip member, level 1 (what `gzip -1` writes), inflated on the host cores",
                                                        totals_equal_resident_copy=True)
        # the BGZF form of the same file (what `bgzip -l 1` writes) two ways: the Reader (blocks inflated on the host cores, the pull
        # stream's path) and ibu_load_bgzf_to_device (the COMPRESSED bytes cross the link, every block is inflated on the device)
        from gzutil import bgzf_parallel
        bg = path + ".bgzf"
        t0 = time.perf_counter()
        bg_bytes = bgzf_parallel(path, bg, level=1, workers=max(2, min(16, usable_cores())))
        tc = time.perf_counter() - t0
        dst = ctxs[0].alloc(24 * n)
        calls = []
        for rep in range(3):
            t0 = time.perf_counter()
            _, dptr, got_n, st = ctxs[0].load_bgzf_to_device(bg, ring=ring, d_records=dst, cap_records=n)
            calls.append(time.perf_counter() - t0)
            if not (got_n == n and ctxs[0].reduce(dptr, n) == want):
                raise LegCheckFailed("e2e: load_bgzf_to_device returned other records than were written")
        # ... and all the way to the ASCII columns (what the two Reader legs above deliver): the load, then K2 over the loaded records
        t0 = time.perf_counter()
        _, dptr, got_n, _ = ctxs[0].load_bgzf_to_device(bg, ring=ring, d_records=dst, cap_records=n)
        ctxs[0].decode_ascii(dptr, n, bc_len, umi_len, s_bc, s_umi, s_idx)
        ctxs[0].synchronize()
        dt_cols = time.perf_counter() - t0
        back = ctxs[0].alloc(24 * n)
        ctxs[0].encode_ascii(s_bc, s_umi, s_idx, n, bc_len, umi_len, back)
        ctxs[0].codec_status()
        ok = ctxs[0].reduce(back, n) == want
        back.free()
        dst.free()
        if not ok:
            raise LegCheckFailed("e2e: BGZF load -> DECODE does not reproduce the records written")
        # the Reader API on the same file (`Reader::from_path(bgzf)` + a device processor): by default the library reads an untouched
        # BGZF file itself and inflates on the device (option "bgzf_device" = 1); = 0: the Reader's host inflate (block-parallel)
        for opt, key, how in ((1, "bgzf_reader_process_device_decode", "BGZF blocks, level 1, read by the library and inflated ON THE DEVICE (the default for an untouched BGZF file)"),
                              (0, "bgzf_reader_process_device_decode_host_inflate", "BGZF blocks, level 1, inflated on the host cores (block-parallel)")):
            ctxs[0].set_option("bgzf_device", opt)
            best = None
            for rep in range(2 if opt else 1):
                r = ia.Reader.from_path(bg)
                t0 = time.perf_counter()
                _, st = r.process_device(ctxs[0], ia.PROC_DECODE, sink=(s_bc, s_umi, s_idx, n), ring=ring)
                dt = time.perf_counter() - t0
                r.close()
                best = dt if best is None or dt < best else best
            back = ctxs[0].alloc(24 * n)
            ctxs[0].encode_ascii(s_bc, s_umi, s_idx, n, bc_len, umi_len, back)
            ctxs[0].codec_status()
            ok = ctxs[0].reduce(back, n) == want
            back.free()
            if not ok:
                raise LegCheckFailed("e2e: BGZF Reader -> DECODE does not reproduce the records written")
            out[key] = rate(best, st, bgzf_bytes=bg_bytes, input=how, totals_equal_resident_copy=True)
        ctxs[0].set_option("bgzf_device", 1)
        out["bgzf_load_to_device_then_decode"] = rate(dt_cols, None, totals_equal_resident_copy=True,
                                                      what="ibu_load_bgzf_to_device + ibu_decode_ascii: BGZF file -> ASCII columns on the device")
        out["bgzf_load_to_device_device_inflate"] = rate(min(calls), None, calls_seconds=[round(c_, 4) for c_ in calls], bgzf_bytes=bg_bytes, compress_seconds=tc,
                                                         bytes_over_the_link=bg_bytes, totals_equal_resident_copy=True,
                                                         call="ibu_load_bgzf_to_device(ctx, path, ring, &header, &d_records, cap, &n, &stats): "
                                                              "one lane per BGZF block (k_inflate.hip)")
    finally:
        for c in ctxs:
            c.close()
        for f in (path, gz, path + ".bgzf"):
            if os.path.exists(f):
                os.unlink(f)
    out["numa"] = ctx.numa()   # where the device hangs off the host, and where the pinned ring of the runs above landed (option "numa")
    out["leg_seconds"] = time.perf_counter() - t_leg
    return out


def bgzf_large_leg(ctx, ia, n, bc_len, umi_len, seed, tmpdir):
    """ibu_load_bgzf_to_device on a file large enough for its launch AHEAD of the copies (more than 49 152 blocks): a file of `n` synthetic
    records written from device memory, compressed to BGZF blocks on the host cores (tools/gzutil.py: what `bgzip -l 1` writes), loaded
    with the compressed bytes crossing the link and every block inflated on the device — beside the PLAIN file of the same records through
    ibu_load_to_device.  Both results checked with K4 against the resident records.  Outside the timed region, never part of `value`."""
    import tempfile

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gzutil import bgzf_parallel
    n = int(n)
    ring = {"slots": 4, "slot_records": 4 << 20, "feeder_threads": 8}
    d = ctx.alloc(24 * n)
    path = os.path.join(tmpdir or tempfile.gettempdir(), f"ibu_bench_big_{os.getpid()}.ibu")
    bg = path + ".bgzf"
    out = {"workload": f"{n:.3g} records bc_len={bc_len} umi_len={umi_len}: a {24 * n / 1e9:.1f} GB file and its BGZF form (level 1) in {tmpdir or tempfile.gettempdir()}"}
    try:
        ctx.generate(seed, 0, n, bc_len, umi_len, d)
        want = ctx.reduce(d, n)
        w = ia.Writer.from_path(path, ia.Header(bc_len, umi_len))
        w.write_batch_device(ctx, d, n, ring=ring)
        w.finish()
        w.close()
        t0 = time.perf_counter()
        bg_bytes = bgzf_parallel(path, bg, level=1, workers=max(2, min(16, usable_cores())))
        out["host_compress_seconds"] = round(time.perf_counter() - t0, 2)
        out["bgzf_bytes"] = bg_bytes
        for name, call in (("load_bgzf_to_device", lambda: ctx.load_bgzf_to_device(bg, ring=ring, d_records=d, cap_records=n)),
                           ("load_to_device_plain_file", lambda: ctx.load_to_device(path, ring=ring, d_records=d, cap_records=n))):
            ts = []
            for _ in range(3):
                ctx.generate(1, 0, min(n, 1 << 20), bc_len, umi_len, d)   # (the destination's head holds something else before every load)
                t0 = time.perf_counter()
                _, dptr, got_n, _ = call()
                ts.append(time.perf_counter() - t0)
                if not (got_n == n and ctx.reduce(dptr, n) == want):
                    raise LegCheckFailed(f"bgzf_large: {name} returned other records than were written")
            out[name] = {"seconds": min(ts), "calls_seconds": [round(t, 4) for t in ts], "records_per_s": n / min(ts), "GBps_of_records": 24 * n / min(ts) / 1e9,
                         "totals_equal_resident_copy": True}
        out["load_bgzf_to_device"]["GBps_over_the_link"] = bg_bytes / out["load_bgzf_to_device"]["seconds"] / 1e9
        out["bgzf_over_plain"] = out["load_to_device_plain_file"]["seconds"] / out["load_bgzf_to_device"]["seconds"]
    finally:
        d.free()
        for f in (path, bg):
            if os.path.exists(f):
                os.unlink(f)
    return out


def sort_contexts_rehearsal(ia, n, bc_len, umi_len, seed, k=8, rounds=3):
    """ibu_sort_records_contexts — the sort over several GPUs behind ONE call of the C ABI — REHEARSED with k contexts that all
    sit on this one GPU (the call allows it): the k shards' kernels share the device and the exchange is a device-local copy, so
    the time says what the call's control flow and its 3-passes-before-the-exchange cost, NOT how it scales; distinct GPUs have
    never run (no such box in any round).  Checked: every shard sorted, the shards' boundaries in order, count / wrapping sums /
    XORs of all shards together unchanged."""
    import statistics

    import numpy as np

    ctxs = [ia.Context(0) for _ in range(k)]
    try:
        per, cap = n // k, int(n // k * 1.25) + k + 1
        bufs = [(c.alloc(24 * cap), c.alloc(24 * cap)) for c in ctxs]
        ts, outs, before = [], None, None
        for _ in range(rounds + 1):
            tot = {"count": 0, "sum": [0, 0, 0], "xor": [0, 0, 0]}
            for i, (c, (d, _)) in enumerate(zip(ctxs, bufs)):  # shard i = rows [i per, (i + 1) per) of the generator's stream
                c.generate(seed, i * per, per, bc_len, umi_len, d)
                r = c.reduce(d, per)
                tot = {"count": tot["count"] + r["count"], "sum": [(a + b) & (2**64 - 1) for a, b in zip(tot["sum"], r["sum"])],
                       "xor": [a ^ b for a, b in zip(tot["xor"], r["xor"])]}
            before = tot
            t0 = time.perf_counter()
            outs = ia.Context.sort_records_contexts(ctxs, [(d, t, per, cap) for d, t in bufs])
            ts.append(time.perf_counter() - t0)
        after = {"count": 0, "sum": [0, 0, 0], "xor": [0, 0, 0]}
        edges = []
        ok = sum(outs) == per * k
        for c, (d, _), m in zip(ctxs, bufs, outs):
            ok = ok and (m == 0 or bool(c.is_sorted(d, m)))
            r = c.reduce(d, m)
            after = {"count": after["count"] + r["count"], "sum": [(a + b) & (2**64 - 1) for a, b in zip(after["sum"], r["sum"])],
                     "xor": [a ^ b for a, b in zip(after["xor"], r["xor"])]}
            if m:
                edges.append((tuple(int(x) for x in d.download(np.uint64, count=3)),
                              tuple(int(x) for x in d.download(np.uint64, count=3, offset=24 * (m - 1)))))
        ok = ok and after == before and all(edges[i][1] <= edges[i + 1][0] for i in range(len(edges) - 1))
        sec = statistics.median(ts[1:])
        return {"contexts_on_this_gpu": k, "records": per * k, "seconds": sec, "records_per_s": per * k / sec, "records_per_shard_after": outs,
                "globally_sorted_and_multiset_preserved": bool(ok),
                "note": "a rehearsal on ONE GPU (shared by the contexts; the exchange is a device-local copy): not a scaling measurement"}
    finally:
        for c in ctxs:
            c.close()


class Leg:
    """One resident workload: this rank's shard [first, first + n) of the synthetic stream, its output columns and the
    re-encoded records.  step() = K2 decode followed by K3 encode."""

    def __init__(self, ctx, torch, dev, st, seed, first, n, bc_len, umi_len, tries=1):
        self.ctx, self.torch, self.dev, self.st, self.n, self.bc_len, self.umi_len = ctx, torch, dev, st, n, bc_len, umi_len
        self.reports = {}

        def buf(name, nbytes):   # the library's allocator: plain, or with placement probing (ibu_device_alloc_probed)
            nbytes = max(nbytes, 16)
            if tries <= 1:
                return ctx.alloc(nbytes)
            b, rep = ctx.alloc_probed(nbytes, tries)
            self.reports[name] = rep
            return b

        # largest arrays first: they are probed while the least memory is taken
        self.recs, self.back = buf("records", n * 24), buf("output", n * 24)
        self.bc, self.umi, self.idx = buf("bc", n * bc_len), buf("umi", n * umi_len), buf("idx", n * 8)
        ctx.generate(seed, first, n, bc_len, umi_len, self.recs, stream=st)
        torch.cuda.synchronize()

    def step(self, ev=None):
        c, n, st = self.ctx, self.n, self.st
        if ev:
            ev[0].record()
        c.decode_ascii(self.recs, n, self.bc_len, self.umi_len, self.bc, self.umi, self.idx, stream=st)
        if ev:
            ev[1].record()
        c.encode_ascii(self.bc, self.umi, self.idx, n, self.bc_len, self.umi_len, self.back, stream=st)
        if ev:
            ev[2].record()

    def _timed_once(self, fn):
        torch = self.torch
        fn()                                            # untimed first touch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1)

    def probe_decode_arrays(self, a):
        """ms of one decode over the arrays in `a` (recs -> bc, umi, idx), whichever sets they come from (placement probing)."""
        return self._timed_once(lambda: self.ctx.decode_ascii(a["recs"], self.n, self.bc_len, self.umi_len, a["bc"], a["umi"], a["idx"],
                                                              stream=self.st))

    def probe_encode_arrays(self, a):
        """ms of one encode over the arrays in `a` (bc, umi, idx -> back; the columns hold decoded data in every set)."""
        return self._timed_once(lambda: self.ctx.encode_ascii(a["bc"], a["umi"], a["idx"], self.n, self.bc_len, self.umi_len, a["back"],
                                                              stream=self.st))

    def probe(self):
        """(decode ms, encode ms) of one step after one untimed step (placement probing, outside every timed region)."""
        torch = self.torch
        self.step()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        self.step(ev)
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])

    def timed(self, steps, warmup, barrier):
        torch = self.torch
        for _ in range(warmup):
            self.step()
        events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            self.step(events[k])
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        dec_ms = sum(e[0].elapsed_time(e[1]) for e in events) / steps
        enc_ms = sum(e[1].elapsed_time(e[2]) for e in events) / steps
        return elapsed, dec_ms, enc_ms

    def copy_ceiling(self):
        """This box's own copy ceiling (plain dwordx4 copy kernel, recs -> the barcode column), outside the timed region:
        the second denominator SURVEY 8d asks for next to the 8 TB/s spec peak; boxes of this pool differ by ~20 %."""
        torch, nbytes = self.torch, min(self.bc.nbytes, self.recs.nbytes)
        ms = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.ctx.copy(self.bc, self.recs, nbytes, stream=self.st)
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        # the copy overwrote the decoded barcode column: decode once more so the verification sees real data
        self.ctx.decode_ascii(self.recs, self.n, self.bc_len, self.umi_len, self.bc, self.umi, self.idx, stream=self.st)
        return 2 * nbytes / (sorted(ms)[len(ms) // 2] * 1e-3) / 1e9

    def verify(self, full):
        """Correctness of what was timed: no record failed to encode, encode(decode(x)) == x byte for byte."""
        torch, n = self.torch, self.n
        self.ctx.codec_status(stream=self.st)  # raises if any record failed to encode
        red = self.ctx.reduce(self.recs, n, stream=self.st)
        ok = None
        if full:
            red_back = self.ctx.reduce(self.back, n, stream=self.st)
            # whole-buffer equality twice: the library's own slice comparison (ibu_records_first_mismatch) and torch.equal
            # in 1 GiB pieces (torch.equal materialises a mask as large as its inputs)
            first_diff = self.ctx.first_mismatch(self.recs, self.back, n, stream=self.st)
            chunk = 1 << 30
            nb = n * 24
            ta, tb = torch.as_tensor(self.recs, device=self.dev), torch.as_tensor(self.back, device=self.dev)   # __cuda_array_interface__ views
            assert ta.data_ptr() == self.recs.ptr and ta.numel() >= nb and ta.dtype == torch.uint8
            same = all(bool(torch.equal(ta[o:min(o + chunk, nb)], tb[o:min(o + chunk, nb)])) for o in range(0, nb, chunk))
            del ta, tb
            ok = same and first_diff == n and red == red_back and red["count"] == n
            if not ok:
                raise SystemExit("round trip encode(decode(x)) != x")
        return red, ok

    def free(self):
        for b in (self.recs, self.back, self.bc, self.umi, self.idx):
            if b is not None:
                b.free()
        self.recs = self.back = self.bc = self.umi = self.idx = None
        self.torch.cuda.empty_cache()


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    what, world = resolve_world(args, argv, os.environ)
    if what == "spawn":                              # `python bench.py --gpus N` run bare: N ranks as a child job
        import subprocess
        raise SystemExit(subprocess.call(world))
    import torch
    import torch.distributed as dist

    import ibu_amd
    from ibu_amd import sharding

    exit_code = 0
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu:
        local_rank = 0
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "nccl":  # "nccl" IS RCCL on ROCm
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctx = ibu_amd.Context(local_rank)
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")  # where the few collective words live

    bc_len, umi_len = args.bc_len, args.umi_len
    # strong: --records is the whole stream (configs[3]: 1e9 records range-sharded over the GPUs); weak: per GPU
    n_global, first, n = plan_shard(args.records, args.scaling, world, rank)  # contiguous record-range split
    if args.scaling == "weak":
        assert n == int(args.records)

    # a real (non-null) stream: the ABI reads a NULL stream as "the context's own stream", and
    # torch.cuda.Event only sees work on torch's current stream
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    st = tstream.cuda_stream
    assert st != 0

    def barrier():
        if use_dist:
            dist.barrier()

    leg, placement = place_leg(lambda tries: Leg(ctx, torch, dev, st, args.seed, first, n, bc_len, umi_len, tries), args.placement_tries,
                               n * (48 + bc_len + umi_len + 8), torch, dev, sharers=world if args.share_gpu else 1, ctx=ctx,
                               library_default=(world == 1))
    elapsed, dec_ms, enc_ms = leg.timed(args.steps, args.warmup, barrier)
    per_rank = [[float(n), dec_ms, enc_ms, elapsed]]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        mine = torch.tensor(per_rank[0], dtype=torch.float64, device=coll_dev)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        per_rank = [p.tolist() for p in parts]
        elapsed = float(t.item())

    copy_GBps = leg.copy_ceiling()
    red, verified = leg.verify(not args.no_verify)
    # the one cross-GPU exchange: global count + wrapping field sums (4 x i64 over RCCL); timed on its second call
    # (the first pays the communicator's lazy set-up)
    g = sharding.global_totals(red, device=coll_dev, force=args.force_dist)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    g = sharding.global_totals(red, device=coll_dev, force=args.force_dist)
    torch.cuda.synchronize()
    allreduce_ms = (time.perf_counter() - t0) * 1e3
    tot = [g["count"]] + g["sum"]
    assert tot[0] == n_global
    assert tot[3] == sharding.expected_index_sum(n_global)  # index column is 0..n_global-1

    wide = None
    if world == 1 and not args.no_wide_leg and (bc_len, umi_len) != (32, 32):
        # BASELINE configs[2]: maximum width, encode+decode on one GPU — a short leg outside the headline's timed region
        leg.free()
        nw = int(args.wide_records) or n
        wl, w_placement = place_leg(lambda tries: Leg(ctx, torch, dev, st, 0x1B00002, 0, nw, 32, 32, tries), min(args.placement_tries or 2, 2),
                                    nw * (48 + 72), torch, dev, ctx=ctx)
        w_steps = max(1, min(args.steps, 5))
        w_el, w_dec, w_enc = wl.timed(w_steps, 1, barrier)
        _, w_ok = wl.verify(not args.no_verify)
        wb = nw * 96
        wide = {
            "workload": f"{nw:.3g} records bc_len=32 umi_len=32 (BASELINE configs[2]), K2 decode + K3 encode per step",
            "steps": w_steps, "warmup": 1, "value": nw * w_steps / w_el, "unit": "records/s", "ms_per_step": w_el / w_steps * 1e3,
            "kernel_ms": {"decode": w_dec, "encode": w_enc},
            "kernel_GBps": {"decode": wb / (w_dec * 1e-3) / 1e9, "encode": wb / (w_enc * 1e-3) / 1e9},
            "roofline": {"bound": "hbm", "kernel": "ibu_k_decode<32,32>", "achieved": wb / (w_dec * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": wb / (w_dec * 1e-3) / 1e9 / HBM_PEAK_GBPS, "bytes_per_record": 96},
            "verified_roundtrip": w_ok, "placement": w_placement,
        }
        wl.free()

    sort_leg = None
    if world == 1 and not args.no_sort_leg:
        # SURVEY 8f-2 / 8f-3 on the same box: device sort by (barcode, umi, index) of the headline's records and the per-barcode
        # aggregation of the result — outside the headline's timed region, never part of `value`
        leg.free()
        torch.cuda.empty_cache()
        try:
            sort_leg = sort_and_aggregate(ctx, n, bc_len, umi_len, args.seed, torch=torch, dev=dev)
        except Exception as e:  # the headline stands on its own: a failure here is reported, not fatal
            sort_leg = {"error": f"{type(e).__name__}: {e}"}
        try:                 # the multi-GPU form of the sort (one call of the C ABI), rehearsed with 8 contexts on this GPU
            torch.cuda.empty_cache()
            sort_leg["contexts_rehearsal"] = sort_contexts_rehearsal(ibu_amd, n, bc_len, umi_len, args.seed)
            if not sort_leg["contexts_rehearsal"]["globally_sorted_and_multiset_preserved"]:
                raise LegCheckFailed("ibu_sort_records_contexts: the shards do not form one sorted sequence of the input's records")
        except Exception as e:
            sort_leg["contexts_rehearsal"] = {"error": f"{type(e).__name__}: {e}"}

    e2e = None
    if world == 1 and not args.no_e2e_leg:
        # BASELINE.md §3 / configs[4]: file -> result with PCIe and the host in the path; outside the timed region, never `value`
        leg.free()
        torch.cuda.empty_cache()
        try:
            e2e = e2e_leg(ctx, ibu_amd, args.e2e_records, bc_len, umi_len, args.seed, args.e2e_dir, torch=torch)
        except Exception as e:  # the headline stands on its own: a failure here is reported, not fatal
            e2e = {"error": f"{type(e).__name__}: {e}"}
        if args.bgzf_large_records > 0:
            try:
                torch.cuda.empty_cache()
                e2e["bgzf_large"] = bgzf_large_leg(ctx, ibu_amd, args.bgzf_large_records, bc_len, umi_len, args.seed, args.e2e_dir)
            except Exception as e:
                e2e["bgzf_large"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        bpr = 24 + bc_len + umi_len + 8
        # rank 0's own launch: its records x algorithmic bytes / its measured decode time
        dec_bytes = n * bpr
        achieved = dec_bytes / (dec_ms * 1e-3) / 1e9
        traffic, traffic_src = traffic_from_profile(bc_len, umi_len, n)
        shard_note = ("ONE stream range-sharded over the ranks (strong scaling, BASELINE configs[3])" if args.scaling == "strong"
                      else "every rank holds --records records of a stream of records x ranks (weak scaling)")
        out = {
            "metric": "records/s, fused 2-bit decode+encode of 24-byte IBU records (HBM-resident)",
            "value": n_global * args.steps / elapsed,
            # what a caller that takes plain allocations gets (one decode + one encode on the first set of arrays, this rank, before
            # any probing: placement.first_placement_*), scaled to the job like `value`; `value` itself is measured on arrays from
            # ibu_device_alloc_probed unless --placement-tries 1.  Compare THIS number with round 1 / BENCH_r01.
            "value_first_placement": placement["first_placement_records_per_s"] * (n_global / max(n, 1)),
            # the same on arrays from plain ibu_device_alloc calls under the library's default option (auto probing): what a caller
            # that does nothing about placement gets since round 5
            "value_library_default_placement": (placement["library_default_records_per_s"] * (n_global / max(n, 1))
                                                if "library_default_records_per_s" in placement else None),
            "unit": "records/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": f"{n_global:.3g} records bc_len={bc_len} umi_len={umi_len}, K2 decode + K3 encode per step; {shard_note}; "
                            f"contiguous record-range shard per rank (mmap.rs:297-307), no data-path collective",
                "records_total": n_global, "records_per_gpu": n, "records_per_rank": [int(p[0]) for p in per_rank],
                "bc_len": bc_len, "umi_len": umi_len, "parallelism": f"range-shard x{world}",
            },
            "kernel_ms": {"decode": dec_ms, "encode": enc_ms},
            "kernel_ms_ranks": {"decode": {"min": min(p[1] for p in per_rank), "max": max(p[1] for p in per_rank)},
                                "encode": {"min": min(p[2] for p in per_rank), "max": max(p[2] for p in per_rank)},
                                "wall_ms_per_step": {"min": min(p[3] for p in per_rank) / args.steps * 1e3,
                                                     "max": max(p[3] for p in per_rank) / args.steps * 1e3}},
            "launch_gap_ms_per_step": elapsed / args.steps * 1e3 - (dec_ms + enc_ms),
            "allreduce_ms": allreduce_ms,
            "kernel_GBps": {"decode": achieved, "encode": dec_bytes / (enc_ms * 1e-3) / 1e9},
            "roofline": {
                "bound": "hbm", "kernel": f"ibu_k_decode<{bc_len},{umi_len}>", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                "bytes_per_record": bpr, "records_per_launch": n,
                "copy_ceiling": copy_GBps, "frac_of_copy_ceiling": achieved / copy_GBps,
            },
            "gpu": {"name": torch.cuda.get_device_name(dev), "uuid": str(getattr(torch.cuda.get_device_properties(dev), "uuid", ""))},
            "verified_roundtrip": verified,
            "placement": placement,
            "global_count": tot[0],
        }
        if wide:
            out["config2_32_32"] = wide
        if sort_leg:
            out["sort_leg"] = sort_leg
        if e2e:
            out["e2e"] = e2e
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
        failed = [k for k, leg_ in (("sort_leg", sort_leg), ("e2e", e2e)) if isinstance(leg_, dict) and
                  (str(leg_.get("error", "")).startswith("LegCheckFailed") or str(leg_.get("contexts_rehearsal", {}).get("error", "")).startswith("LegCheckFailed")
                   or str(leg_.get("bgzf_large", {}).get("error", "")).startswith("LegCheckFailed"))]
        if failed:                                   # the line is out; a WRONG result in a side leg still fails the run
            exit_code = 3
            print(f"bench.py: correctness check failed in {failed}", file=sys.stderr)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        raise SystemExit(exit_code)


if __name__ == "__main__":
    main()
