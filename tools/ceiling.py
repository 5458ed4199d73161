#!/usr/bin/env python3
"""On-device copy ceilings for this box (denominators beside the 8 TB/s spec peak): a plain
device-to-device copy (1 read + 1 write stream) and a fill (write only), HIP-event timed."""
import json, statistics, sys
import torch
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 24_000_000_000
dev = torch.device("cuda", 0)
a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
a.fill_(1); torch.cuda.synchronize()
def t(fn, rounds=7):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)
c = t(lambda: b.copy_(a)); f = t(lambda: b.fill_(3)); r = t(lambda: a.view(torch.int64).sum())
print(json.dumps({"bytes": n, "copy_ms": c, "copy_GBps_rw": 2 * n / c / 1e6, "fill_ms": f, "fill_GBps": n / f / 1e6,
                  "read_sum_ms": r, "read_GBps": n / r / 1e6}))
