#!/usr/bin/env python3
"""What placement probing costs at allocation time (option "alloc_probe_tries": 1 = plain, 0 = auto, k = k candidates), by size.
  python tools/alloc_cost.py [--gb 1.2,2.4,8,24] [--fill-gb 0]
--fill-gb G: allocate G GB first (a device that is partly in use: fewer candidates fit)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", default="1.2,2.4,8,24")
    ap.add_argument("--fill-gb", type=float, default=0)
    a = ap.parse_args()
    import ibu_amd as ia
    ctx = ia.Context(0)
    fill = ctx.alloc(int(a.fill_gb * 1e9)) if a.fill_gb else None
    for gb in (float(x) for x in a.gb.split(",")):
        nbytes = int(gb * 1e9)
        for mode in (1, 0, 4, 8):
            ctx.set_option("alloc_probe_tries", mode)
            ts = []
            for _ in range(2):
                t0 = time.perf_counter()
                b = ctx.alloc(nbytes)
                ctx.synchronize()
                t1 = time.perf_counter()
                b.free()
                ctx.synchronize()
                ts.append((t1 - t0, time.perf_counter() - t1))
            print(json.dumps({"GB": gb, "alloc_probe_tries": mode, "alloc_s": [round(t[0], 4) for t in ts], "free_s": [round(t[1], 4) for t in ts]}), flush=True)
    if fill:
        fill.free()
    ctx.close()


if __name__ == "__main__":
    main()
