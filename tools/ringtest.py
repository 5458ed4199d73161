import sys, time, json, os
sys.path.insert(0, '/root/repo')
import ibu_amd as ia
n = 300_000_000
ctx0 = ia.Context(0)
path = f"/tmp/ring_{os.getpid()}.ibu"
d = ctx0.alloc(24*n); ctx0.generate(5, 0, n, 16, 12, d)
w = ia.Writer.from_path(path, ia.Header(16,12)); w.write_batch_device(ctx0, d, n); w.finish(); w.close(); d.free()
m = ia.MmapReader.new(path)
for slot_records in (1<<18, 1<<19, 1<<20, 1<<21, 1<<22):
    for slots in (3, 4):
        ring = {"slots": slots, "slot_records": slot_records, "feeder_threads": 4}
        c = ia.Context(0)
        out = []
        for rep in range(3):
            t0 = time.perf_counter(); res, st = m.process_device(c, ia.PROC_REDUCE, ring=ring); out.append(round(time.perf_counter()-t0, 4))
        print(json.dumps({"slot_MiB": slot_records*24/2**20, "slots": slots, "seconds_first_second_third": out, "GBps_warm": round(24*n/min(out[1:])/1e9, 1)}), flush=True)
        c.close()
os.unlink(path)
