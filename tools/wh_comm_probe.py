"""One rank of a throw-away RCCL clique through the library's own binding: wh_comm_init with the unique id given on the command
line, one 64-word integer all-reduce and one Field128 all-gather + reduce, results checked, exit status 0.  bench.py starts one of
these per rank before it binds its own contexts (try_wh_comm): a child that hangs can be killed, a rank that hangs inside its
own communicator cannot.  Usage: python tools/wh_comm_probe.py <id hex | self> <rank> <world> <device>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, device = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    import numpy as np
    import torch

    import whisper_amd as W
    from whisper_amd.dist import allgather_reduce_dev, comm_init, comm_unique_id, hh_allreduce_u32_dev

    # the id's bootstrap address lives in the process that made it: bench.py's rank 0 for a real clique, this process for "self"
    idb = comm_unique_id() if sys.argv[1] == "self" else bytes.fromhex(sys.argv[1])

    dev = torch.device(f"cuda:{device}")
    torch.cuda.set_device(dev)
    ctx = W.Prio3(W.Prio3Params(W.SUMVEC, 2, 5, 3), device=device)
    comm_init(ctx, idb, rank, world)
    t = torch.full((64,), rank + 1, dtype=torch.int32, device=dev)
    part = torch.zeros(8, 2, dtype=torch.int64, device=dev)
    part[:, 0] = rank + 1                                  # Field128 elements rank + 1 (low word)
    out = torch.zeros_like(part)
    torch.cuda.synchronize()
    hh_allreduce_u32_dev(ctx, t.data_ptr(), 64)
    allgather_reduce_dev(ctx, part.data_ptr(), out.data_ptr(), 8)
    ctx.synchronize()
    want = world * (world + 1) // 2
    ok = bool((t.cpu().numpy() == want).all()) and bool((out.cpu().numpy()[:, 0] == want).all()) and bool((out.cpu().numpy()[:, 1] == 0).all())
    ctx.close()
    return 0 if ok else 4


if __name__ == "__main__":
    sys.exit(main())
