"""Multi-GPU plumbing: one process per GPU, reports sharded by index, partial aggregates combined
by an all-gather (RCCL over xGMI when the backend is "nccl") followed by a mod-p reduction.

The reference has no intra-server distribution (one process, rayon + 32 tokio tasks,
server-batch-prio3/src/main.rs:282-335); what is reproduced here is its final merge of partial
aggregates (main.rs:327-335, AggregateShare::merge vdaf.rs:658-676).  RCCL has no mod-p reduce op,
so raw 16-byte limbs are all-gathered and summed on the GPU (wh_f128_sum_rows_dev).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from ._lib import lib, WhError

COMM_ID_BYTES = 128


# ---- the C-ABI combine (comm.hip): RCCL bound by the library itself, one context per GPU ----
def comm_unique_id():
    """ncclGetUniqueId through the library: rank 0 makes it, every rank passes it to comm_init."""
    buf = np.zeros(COMM_ID_BYTES, np.uint8)
    rc = lib.wh_comm_unique_id(buf.ctypes.data)
    if rc:
        raise WhError(rc, "wh_comm_unique_id (is RCCL present?)")
    return buf.tobytes()


def comm_init(obj, unique_id, rank, world):
    """Bind a context (Prio3 or HeavyHitters instance) to rank `rank` of `world`; unique_id None = a lone GPU without RCCL."""
    idb = None if unique_id is None else np.frombuffer(bytes(unique_id), np.uint8).copy()
    obj._chk(lib.wh_comm_init(obj._ctx, None if idb is None else idb.ctypes.data, rank, world), "wh_comm_init")


def comm_init_from_torch(obj, group=None):
    """The same, with the unique id carried by an initialised torch.distributed group (any backend)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        return comm_init(obj, None, 0, 1)
    box = [comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    comm_init(obj, box[0], rank, world)


def comm_rank(obj):
    r, w = C.c_int32(), C.c_int32()
    obj._chk(lib.wh_comm_rank(obj._ctx, C.byref(r), C.byref(w)), "wh_comm_rank")
    return r.value, w.value


def allgather_reduce_dev(obj, d_partial, d_agg, length=None):
    """d_partial[len] Field128 (device pointer) -> d_agg[len] = sum over ranks mod p, asynchronous on the context stream."""
    if length is None:
        obj._chk(lib.wh_prio3_allgather_reduce_dev(obj._ctx, d_partial, d_agg), "wh_prio3_allgather_reduce_dev")
    else:
        obj._chk(lib.wh_f128_allgather_reduce_dev(obj._ctx, d_partial, length, d_agg), "wh_f128_allgather_reduce_dev")


def hh_allreduce_u32_dev(obj, d_sketch, words):
    obj._chk(lib.wh_hh_allreduce_u32_dev(obj._ctx, d_sketch, words), "wh_hh_allreduce_u32_dev")


def shard_bounds(n, world_size, rank):
    """Contiguous slice [lo, hi) of n reports owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_partials(partial, group=None):
    """partial: uint8 tensor [output_len*16] (this rank's aggregate share of its shard).
    Returns a uint8 tensor [world, output_len*16] holding every rank's partial."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return partial.reshape(1, -1).clone()
    if partial.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal backend (CPU tests, several ranks on one GPU): gloo gathers host tensors only
        host = torch.empty(world * partial.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(host, partial.contiguous().cpu(), group=group)
        return host.to(partial.device).reshape(world, -1)
    out = torch.empty(world * partial.numel(), dtype=torch.uint8, device=partial.device)
    dist.all_gather_into_tensor(out, partial.contiguous(), group=group)
    return out.reshape(world, -1)


def merge_partials(partial, sum_rows, group=None):
    """all-gather + reduction; `sum_rows(rows[world, L]) -> [L]` is the mod-p row sum
    (Prio3.sum_rows_dev on the GPU)."""
    return sum_rows(all_gather_partials(partial, group))


def all_reduce_sketch_u32(acc, group=None):
    """Heavy-hitter sketches: counters live in Z/2^32 (AggRing = u32, hhcore/src/lib.rs:13), so the per-GPU partial
    sums combine with a plain integer all-reduce (ncclSum on int32 wraps exactly like u32::wrapping_add,
    common/src/group.rs:88-91) -- the counterpart of CountSketch::insert over partial sketches (countsketch.rs:22-28).
    acc: int32/uint32 tensor; reduced in place and returned."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return acc
    view = acc.view(torch.int32)
    if view.is_cuda and dist.get_backend(group) == "gloo":
        host = view.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        view.copy_(host.to(view.device))
        return acc
    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group)
    return acc
