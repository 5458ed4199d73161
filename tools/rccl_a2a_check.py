#!/usr/bin/env python3
"""Does dist.all_to_all_single over RCCL return its input?  One rank exchanging with itself, message sizes around 2^30
bytes, default and side stream, with and without split lists.  (Found while rehearsing ibu_amd.sharding.distributed_sort on
one GPU: 1.2e9 bytes and more came back different.  The sort therefore ships grouped point-to-point messages of at most
256 MiB, sharding._exchange_p2p.)   python tools/rccl_a2a_check.py"""
import json
import os

import torch
import torch.distributed as dist


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)

    def run(nbytes, side, splits):
        def body():
            inp = torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device=dev)
            out = torch.zeros_like(inp)
            if splits:
                dist.all_to_all_single(out, inp, [nbytes], [nbytes])
            else:
                dist.all_to_all_single(out, inp)
            return inp, out
        if side:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                inp, out = body()
            torch.cuda.current_stream().wait_stream(s)
        else:
            inp, out = body()
        torch.cuda.synchronize()
        bad = sum(int((out[o:o + (1 << 28)] != inp[o:o + (1 << 28)]).sum()) for o in range(0, nbytes, 1 << 28))
        print(json.dumps({"bytes": nbytes, "side_stream": side, "split_lists": splits, "bytes_that_differ": bad}), flush=True)

    for nb in (1 << 29, 600_000_000, (1 << 30) - 4096, 1 << 30, (1 << 30) + 4096, 1_200_000_000, 4_800_000_000):
        for side in (False, True):
            run(nb, side, True)
        run(nb, False, False)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
