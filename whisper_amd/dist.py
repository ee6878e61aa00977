"""Multi-GPU plumbing: one process per GPU, reports sharded by index, partial aggregates combined
by an all-gather (RCCL over xGMI when the backend is "nccl") followed by a mod-p reduction.

The reference has no intra-server distribution (one process, rayon + 32 tokio tasks,
server-batch-prio3/src/main.rs:282-335); what is reproduced here is its final merge of partial
aggregates (main.rs:327-335, AggregateShare::merge vdaf.rs:658-676).  RCCL has no mod-p reduce op,
so raw 16-byte limbs are all-gathered and summed on the GPU (wh_f128_sum_rows_dev).
"""
import torch
import torch.distributed as dist


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
