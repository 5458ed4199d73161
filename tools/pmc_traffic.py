#!/usr/bin/env python3
"""HBM traffic per record from two rocprofv3 PMC passes over tools/kbench.py.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc/FETCH_SIZE -- python3 tools/kbench.py --records 2e8 --rounds 2
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc/WRITE_SIZE -- python3 tools/kbench.py --records 2e8 --rounds 2
  python tools/pmc_traffic.py gpurun_out/pmc 2e8 16,12 [round-tag] > profiles/pmc_traffic.json
(tools/profile_round.sh runs all of it; with a sibling directory <pmc>_sort holding the same two passes over
tools/sortbench.py the sort kernels get a "sort" section: bytes per record per launch of each kernel.)

FETCH_SIZE / WRITE_SIZE are in KiB (separate passes: FETCH_SIZE costs 3 of the 4 TCC slots,
WRITE_SIZE 2).  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly
half of the bytes of a wide coalesced streaming read, so reads = 2 * FETCH_SIZE * 1024;
WRITE_SIZE is exact for 16-B-per-lane stores.  The calibration rows prove both on this box:
ibu_k_reduce reads exactly 24 B/record and ibu_k_generate writes exactly 24 B/record.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(pmc_dir, counter, info=None):
    """kernel name -> mean counter value per dispatch (main kernels only: tails are negligible).

    rocprofv3 writes one counter_collection.csv per PROCESS it saw (the python launcher, helper processes): every file
    is read and merged, never just the first of the glob (round 2 read files[0], which for the sort passes was another
    process's file, and the "sort" section came out empty).  info (optional dict) receives, per kernel, the register /
    scratch figures the profiler records with every dispatch."""
    files = sorted(glob.glob(os.path.join(pmc_dir, counter, "**", "*counter_collection.csv"), recursive=True))
    if not files:
        raise SystemExit(f"no counter_collection.csv under {pmc_dir}/{counter}")
    acc = defaultdict(list)
    for path in files:
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
                if info is not None:
                    info[row["Kernel_Name"]] = {"Scratch_Size": int(row.get("Scratch_Size") or 0), "VGPR_Count": int(row.get("VGPR_Count") or 0),
                                                "SGPR_Count": int(row.get("SGPR_Count") or 0), "LDS_Block_Size": int(row.get("LDS_Block_Size") or 0),
                                                "dispatches": len(acc[row["Kernel_Name"]])}
    if not acc:
        raise SystemExit(f"no {counter} rows in {files}")
    return {k: sum(v) / len(v) for k, v in acc.items()}


def short(name):
    for k in ("decode", "encode", "deserialize", "serialize", "reduce", "unpack", "pack", "generate", "copy"):
        if f"ibu_k_{k}" in name and "_tail" not in name and "_bytes" not in name:
            return k
    return None


def main():
    pmc_dir, n, lens = sys.argv[1], float(sys.argv[2]), sys.argv[3]
    bc_len, umi_len = (int(x) for x in lens.split(","))
    fetch, write = per_kernel(pmc_dir, "FETCH_SIZE"), per_kernel(pmc_dir, "WRITE_SIZE")
    alg = {"decode": 24 + bc_len + umi_len + 8, "encode": 24 + bc_len + umi_len + 8, "deserialize": 48, "serialize": 48,
           "reduce": 24, "unpack": 8 + bc_len, "pack": 8 + bc_len, "generate": 24, "copy": 48}
    out = {"_method": __doc__.split("FETCH_SIZE / WRITE_SIZE are", 1)[1].strip().replace("\n", " "),
           "_round": sys.argv[4] if len(sys.argv) > 4 else None, "_records": n, "_lens": [bc_len, umi_len]}
    for name, f_kib in sorted(fetch.items()):
        k = short(name)
        if not k:
            continue
        w_kib = write.get(name, 0.0)
        rd, wr = 2 * f_kib * 1024 / n, w_kib * 1024 / n
        key = f"{k}_{bc_len}_{umi_len}" if k in ("decode", "encode") else k
        out[key] = {"kernel": name.split("(")[0], "FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib,
                    "read_bytes_per_record": round(rd, 3), "write_bytes_per_record": round(wr, 3),
                    "hbm_bytes_per_record": round(rd + wr, 3), "algorithmic_bytes_per_record": alg[k]}
    sort_dir = pmc_dir.rstrip("/") + "_sort"
    if os.path.isdir(sort_dir):  # every ibu_k_sort_* kernel: mean bytes per record per launch (FETCH doubled as above)
        info = {}
        sf, sw = per_kernel(sort_dir, "FETCH_SIZE", info), per_kernel(sort_dir, "WRITE_SIZE", info)
        out["sort"] = {name.split("(")[0].replace("void ", ""): dict(
            read_bytes_per_record=round(2 * f_kib * 1024 / n, 3), write_bytes_per_record=round(sw.get(name, 0.0) * 1024 / n, 3), **info.get(name, {}))
            for name, f_kib in sorted(sf.items()) if "ibu_k_sort" in name or "ibu_k_copy" in name}
        if not out["sort"]:
            raise SystemExit(f"no ibu_k_sort_* kernel in the counter files under {sort_dir}")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
