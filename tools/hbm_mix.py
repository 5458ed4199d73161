#!/usr/bin/env python3
"""HBM ceiling of this box per read:write mix (measurement-only kernel tools/native/hbm_mix.hip).
  python tools/hbm_mix.py [--gb 48]"""
import argparse, ctypes as C, json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--gb", type=float, default=48); a = ap.parse_args()
    so = os.path.join(ROOT, "tools", "native", "libhbm_mix.so")
    if not os.path.exists(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so,
                               os.path.join(ROOT, "tools", "native", "hbm_mix.hip")])
    import torch  # before the library: both must bind the HIP runtime torch ships
    lib = C.CDLL(so)
    lib.hbm_mix.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
    dev = torch.device("cuda", 0); ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts)
    total = int(a.gb * 1e9)
    src = torch.zeros(total, dtype=torch.uint8, device=dev); dst = torch.empty(total, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for r, w in [(1, 0), (3, 0), (1, 1), (2, 3), (3, 2), (1, 2), (2, 1), (1, 3), (3, 1), (4, 1)]:
        steps = total // 16 // max(r, w, 1) // 4 * 4
        for blocks in (256 * 8, 256 * 7, 256 * 4):
            ms = []
            for i in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); rc = lib.hbm_mix(r, w, src.data_ptr(), dst.data_ptr(), steps, blocks, ts.cuda_stream); e1.record(); e1.synchronize()
                assert rc == 0, rc
                if i: ms.append(e0.elapsed_time(e1))
            m = statistics.median(ms)
            print(json.dumps({"read_chunks": r, "write_chunks": w, "blocks": blocks, "ms": round(m, 3),
                              "GBps": round((r + w) * 16 * steps / m / 1e6)}), flush=True)
if __name__ == "__main__": main()
