#!/usr/bin/env python3
"""Thread scaling of the CPU baseline (oracle decode+encode) on this host, and what the cgroup allows."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from oracle import oracle as orc

for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
print("affinity", len(os.sched_getaffinity(0)), "usable", bench.usable_cores(), flush=True)
for th in (1, 8, 16, 32, 64, 128, 256):
    n = 20_000_000 * min(th, 16)
    t, c = orc.bench_decode_encode(n, 16, 12, 1, th, 1)
    print(th, "threads", n, "records", round(t, 3), "s", round(n / t / 1e6, 1), "M rec/s", flush=True)
