"""Does RCCL accept two ranks on ONE device? (It does not: 'Duplicate GPU detected' — which is why tests/test_gpu_dist.py runs its two
ranks over the host-staged gloo transport.) Run on the GPU box: python tools/rccl_dup_probe.py"""
import os
import socket
import sys

import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(rank, port, q):
    import __graft_entry__ as ge
    gkc = ge.load().gkc
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    c = gkc.Counter(0)
    box = [gkc.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    try:
        comm = gkc.Comm.rccl(c, box[0], 2, rank)
        q.put((rank, "accepted"))
        comm.close()
    except Exception as e:       # noqa
        q.put((rank, "refused: %s" % e))
    dist.destroy_process_group()


if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=main, args=(r, port, q)) for r in range(2)]
    [p.start() for p in ps]
    for _ in range(2):
        try:
            print(q.get(timeout=120))
        except Exception:
            print("no answer (hang)")
    for p in ps:
        p.join(timeout=10)
        if p.is_alive():
            p.terminate()
