"""The N > 1 path on CPU: world_size 2, gloo.  Each rank verifies its shard (the oracle stands in
for the GPU compute here), partial aggregates are all-gathered and reduced mod p; the result must
equal the single-process aggregate."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    import whisper_amd as W
    from whisper_amd.dist import shard_bounds, merge_partials
    from common import VK, blob_of, make_reports, server_views
    p = W.Prio3Params(W.SUMVEC, 16, 8, 4)
    op = O.params(O.SUMVEC, 16, 8, 4)
    lr, hr, _ = make_reports(p, n, seed=21, nthreads=2)
    alice, _ = server_views(lr, hr)
    lo, hi = shard_bounds(n, world, rank)
    blob, offsets = blob_of(alice[lo:hi])
    _, _, status, outs = O.prio3_prepare_batch(op, VK, blob, offsets, nthreads=1)
    assert not status.any()
    keep = np.ones(hi - lo, np.uint8)
    keep[0] = 0
    partial = torch.from_numpy(O.prio3_aggregate(outs, p.output_len, keep).copy())
    total = merge_partials(partial, lambda rows: torch.from_numpy(O.prio3_aggregate(rows.numpy(), p.output_len).copy()))
    if rank == 0:
        q.put((lo, hi, total.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds():
    from whisper_amd.dist import shard_bounds
    for n in (0, 1, 7, 8, 125000):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_allgather_reduce_matches_single_process():
    import oracle_lib as O
    import whisper_amd as W
    from whisper_amd.dist import shard_bounds
    from common import VK, blob_of, make_reports, server_views
    n, world = 11, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    lo, hi, total = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    p = W.Prio3Params(W.SUMVEC, 16, 8, 4)
    op = O.params(O.SUMVEC, 16, 8, 4)
    lr, hr, _ = make_reports(p, n, seed=21, nthreads=2)
    alice, _ = server_views(lr, hr)
    blob, offsets = blob_of(alice)
    _, _, _, outs = O.prio3_prepare_batch(op, VK, blob, offsets, nthreads=1)
    keep = np.ones(n, np.uint8)
    for r in range(world):
        keep[shard_bounds(n, world, r)[0]] = 0
    assert total == O.prio3_aggregate(outs, p.output_len, keep).tobytes()


def _hh_worker(rank, world, port, n, depth, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    from whisper_amd.dist import shard_bounds, all_reduce_sketch_u32
    keys = _hh_keys(n, depth)
    lo, hi = shard_bounds(n, world, rank)
    dat, _, _, status = O.hh_eval_and_check_batch(depth, bytes(16), keys[lo:hi], nthreads=1)
    assert not status.any()
    keep = np.ones(hi - lo, np.uint8)
    keep[0] = 0
    acc = torch.from_numpy(O.hh_bucket_sum_u32(dat, keep).astype(np.uint32).view(np.int32).copy())
    all_reduce_sketch_u32(acc)
    if rank == 0:
        q.put(acc.numpy().view(np.uint32).tobytes())
    dist.barrier()
    dist.destroy_process_group()


def _hh_keys(n, depth):
    import oracle_lib as O
    rng = np.random.default_rng(31)
    keys = np.zeros((n, O.hh_key_chain_len(depth)), np.uint8)
    for i in range(n):
        bk = np.where(rng.integers(0, 2, 258) == 1, 1, 0xffffffff).astype(np.uint32)
        bk[257] = 1
        a, _ = O.hh_gen_key_chain(depth, i, int(rng.integers(0, 1 << depth)), bk, rng.integers(0, 256, 32, dtype=np.uint8).tobytes())
        keys[i] = np.frombuffer(a, np.uint8)
    return keys


def test_two_rank_sketch_allreduce_matches_single_process():
    """Heavy hitters over two ranks: key chains sharded by index, per-rank sketch sums, integer all-reduce."""
    import oracle_lib as O
    from whisper_amd.dist import shard_bounds
    n, world, depth = 9, 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hh_worker, args=(r, world, port, n, depth, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    total = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    dat, _, _, _ = O.hh_eval_and_check_batch(depth, bytes(16), _hh_keys(n, depth), nthreads=1)
    keep = np.ones(n, np.uint8)
    for r in range(world):
        keep[shard_bounds(n, world, r)[0]] = 0
    assert total == O.hh_bucket_sum_u32(dat, keep).astype(np.uint32).tobytes()
