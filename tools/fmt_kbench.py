import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if "kernel" in r: print(r["tag"], r["kernel"], r["ms_med"], r["GBps_med"])
    else: print(r)
