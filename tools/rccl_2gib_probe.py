"""Reproducer for the round-1 finding "a single transfer of >= 2 GiB per peer through torch's all_to_all_single / RCCL comes back wrong" — the reason every
message of gkc_exchange stays below 1 GiB (csrc/gkc_dist.hip MSG_CHUNK). One rank (the only configuration this project could run):
    python tools/rccl_2gib_probe.py            -> prints, per size, whether out == in after dist.all_to_all_single and after the library's own gkc path"""
import os
import socket
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    torch.cuda.set_device(0)
    for gib in (1.0, 1.9, 2.0, 2.5, 3.0):
        n = int(gib * (1 << 30))
        inp = torch.empty(n, dtype=torch.uint8, device="cuda")
        inp.copy_((torch.arange(n, device="cuda", dtype=torch.int64) * 2654435761 >> 7).to(torch.uint8))
        out = torch.zeros_like(inp)
        try:
            dist.all_to_all_single(out, inp)
            torch.cuda.synchronize()
            bad = int((out != inp).sum().item())
            first = int(torch.nonzero(out != inp)[0].item()) if bad else -1
            print("all_to_all_single %.1f GiB: %s" % (gib, "ok" if bad == 0 else "CORRUPT: %d bytes differ, first at offset %d" % (bad, first)))
        except Exception as e:      # noqa
            print("all_to_all_single %.1f GiB: raised %s" % (gib, str(e)[:200]))
        del inp, out
        torch.cuda.empty_cache()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
