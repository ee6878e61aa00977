#!/usr/bin/env python3
"""Benchmark of the hot path: batched Prio3 verification + aggregation on MI355X.

One step = one pass over one batch of encoded client reports already resident in HBM:
  wh_prio3_prepare_batch_dev (parse, share expansion + measurement sponges, Fiat-Shamir hashing, query + RLC tag)
  + wh_prio3_aggregate_dev (masked sum of out shares)
  + for N > 1: RCCL all-gather of the per-GPU partial aggregates and an on-GPU mod-p reduction.
Reports shard across ranks with no data-path collective (weak scaling: fixed reports per GPU).
Every report comes from a distinct synthetic client, sharded on the same GPU before the timed region
(wh_client_shard_many_gpu, byte-identical to the CPU client); --tiled makes a few on the CPU and repeats them instead.

  python bench.py --gpus 1 --steps 5 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import whisper_amd as W  # noqa: E402
from whisper_amd.dist import all_gather_partials, allgather_reduce_dev, comm_init_from_torch, hh_allreduce_u32_dev  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import VK, make_reports  # noqa: E402  (deterministic synthetic clients; no oracle involved)

WORKLOADS = {
    # BASELINE.json configs[2]: Prio3 SumVec large vec_size/chunk_size, 1M clients over 8 GPUs
    "sumvec_large": dict(circuit=W.SUMVEC, bits=16, length=1024, chunk=128, reports_per_gpu=125_000,
                         desc="Prio3 SumVec bits=16 vec_size=1024 chunk_size=128, 125k reports/GPU (1M clients over 8 GPUs)"),
    # BASELINE.json configs[0]: the reference's own configs/prio3-server-{alice,bob}.json sizes
    "sumvec_cfg1": dict(circuit=W.SUMVEC, bits=16, length=128, chunk=64, reports_per_gpu=500_000,
                        desc="Prio3 SumVec bits=16 vec_size=128 chunk_size=64 (configs/prio3-server-alice.json sizes)"),
    # BASELINE.json configs[1]
    "histogram": dict(circuit=W.HISTOGRAM, bits=0, length=128, chunk=64, reports_per_gpu=100_000,
                      desc="Prio3 Histogram length=128 chunk_size=64, 100k clients"),
    # the servers' third aggregation function (agg_fn = average, AVG_BITS = 64, bin-utils/src/lib.rs:34): PolyEval circuit
    "average": dict(circuit=W.AVERAGE, bits=64, length=0, chunk=0, reports_per_gpu=500_000,
                    desc="Prio3 Average bits=64 (AVG_BITS), 500k clients"),
}
# heavy hitters (server-hh): (repetitions, buckets) from the recovery threshold, bin-utils/src/lib.rs:66-88; one step =
# expansion (eval_and_check) of clients x repetitions key chains + masked sketch sums
HH_WORKLOADS = {
    "hh_t0.01": dict(reps=14, buckets=512, clients_per_gpu=1000,
                     desc="heavy hitters, recovery threshold 0.01: 14 repetitions x 512 buckets, 1000 clients/GPU"),
    "hh_t0.001": dict(reps=17, buckets=2048, clients_per_gpu=250,
                      desc="heavy hitters, recovery threshold 0.001: 17 repetitions x 2048 buckets, 250 clients/GPU"),
}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# Keccak-f[1600] on the vector ALUs.  Measured issue model of a gfx950 SIMD (tools/inst_rate_gen.py, profiles/r2_inst_rate*.txt):
# two VALU instructions start per four cycles, from two different wavefronts; the first slot takes anything, the second only
# "full-rate" instructions (VOP1/VOP2 class, v_bitop3_b32), never v_alignbit_b32.  A round is 122 v_bitop3/v_xor and 58
# v_alignbit: with the rotations in the first slot (keccak.hpp raises the wavefront's priority for them) the floor is
# 180 instructions x 2 cycles = 360 cycles per wavefront-round, 8 640 per wavefront-permutation; 1 024 SIMDs x 64 lanes at the
# 2.4 GHz maximum clock give 18.2 G permutations/s: the issue-rate peak.  The measured ceiling of the same round code with
# nothing else in the kernel (tools/keccak_bench.hip after two seconds of load, profiles/r2b_keccak_bench.txt) is 13.69 G/s at
# 6 wavefronts per SIMD and 12.96 G/s at 3, where the chip is power-limited (1.29 kW of its 1.4 kW cap, profiles/
# r2_power_probe.txt) and holds 1.95-2.07 GHz, not 2.4.
KECCAK_ISSUE_CYCLES = 24 * 180 * 2
MAX_CLOCK_HZ = 2.4e9
KECCAK_PEAK = 1024 * 64 * MAX_CLOCK_HZ / KECCAK_ISSUE_CYCLES
KECCAK_PEAK_MEASURED = 13.69e9
KECCAK_PEAK_MEASURED_3_WAVES = 12.96e9
K_MEAS_CLOCK_GHZ = 1.86      # in-kernel clock of k_meas (s_memtime / s_memrealtime, tools/meas_bench.hip, profiles/r2b_meas_bench.txt)


def alg_bytes_per_report(s, role):
    """Algorithmic bytes each kernel must move for one report ('Kernels')."""
    IL, PL, VL, OL, C = s.input_len, s.proof_len, s.verifier_len, s.output_len, s.calls
    meas = 16 * IL + 64                                                      # Leader: share absorbed from the report; Helper: share written
    hash_rd = 32 * PL + 32 * VL + 64 + 64 + 16 + 16 + 16 + 16               # absorbed bytes: 2 proofs, pp, parts, blinds, nonce, id
    hash_wr = 16 * (VL + 5) + 16 * (C + 1) + 48 * C + 848 + 112 + 16         # rlc, weights, weight limbs, power tables, scalars, testing id
    query_rd = 16 * IL + 16 * PL + 32 * VL + 16 * (VL + 5) + 48 * C + 848 + 112 + (16 * PL if role == 0 else 0)
    query_wr = 16 * OL + 16
    expand = (64 + 32 * PL) if role == 1 else 0                              # the two proof shares of a Helper
    # chain (short shares: every sponge of the report in one launch) = what expand + meas + hash move; a Helper also reads back
    # the shares it has just written
    chain = expand + meas + hash_rd + hash_wr + ((16 * IL + 32 * PL) if role == 1 else 0)
    return {"expand": expand, "meas": meas, "hash": hash_rd + hash_wr, "query": query_rd + query_wr, "parse": 64 + 56,
            "aggregate": 16 * OL + 1, "chain": chain, "sort": 16 * 6 + 32 + 32}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="sumvec_large", choices=sorted(WORKLOADS) + sorted(HH_WORKLOADS))
    ap.add_argument("--reports-per-gpu", type=int, default=0)
    ap.add_argument("--unique", type=int, default=128, help="with --tiled: distinct clients generated on the CPU")
    ap.add_argument("--tiled", dest="distinct", action="store_false",
                    help="--unique distinct clients sharded on the CPU and tiled, instead of every report from a distinct client sharded on the GPU")
    ap.add_argument("--in-flight", type=int, default=1, help="contexts the steps alternate between (batches in flight)")
    ap.add_argument("--no-sort", dest="with_sort", action="store_false",
                    help="leave the sort by testing id, the gather of ids + tags and the keep mask from the group test's bad ids out of the step "
                         "(the step is then prepare + masked aggregate with a ready-made mask)")
    ap.add_argument("--with-sort", dest="with_sort", action="store_true", help="(default) the whole GPU side of the server step")
    ap.add_argument("--no-extra", action="store_true",
                    help="default run only: skip the short lines for the other configurations (config 1, Histogram, Average, the two heavy-hitter workloads) under `extra`")
    ap.add_argument("--host-buffers", choices=("pageable", "pinned"), default=None,
                    help="time the host-buffer entry points instead (wh_prio3_prepare_batch + wh_prio3_aggregate): reports start in host "
                         "memory, tags / ids / status / aggregate end there; PCIe inclusive")
    ap.add_argument("--data-rank", type=int, default=None,
                    help="make the synthetic clients of this rank instead of the process's own (tests: a one-rank run over another rank's shard)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.set_defaults(with_sort=True)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the measured path)")
    # WH_BENCH_BACKEND=gloo rehearses the N > 1 path with several ranks on one GPU (RCCL wants one GPU per rank)
    backend = os.environ.get("WH_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if args.workload in HH_WORKLOADS:
        return bench_hh(args, rank, local_rank, world, dev, backend)
    result = bench_prio3(args, args.workload, args.reports_per_gpu, args.steps, args.warmup, rank, local_rank, world, dev, backend,
                         cpu=not args.no_cpu_baseline and world == 1)      # the CPU baseline: rank 0 at N = 1 only
    if (rank == 0 and world == 1 and not args.no_extra and not args.no_cpu_baseline and not args.host_buffers and args.workload == "sumvec_large"
            and not args.reports_per_gpu):
        # the other Prio3 configurations of BASELINE.json (config 1 = the reference's shipped JSON, config 2 = Histogram) and the
        # servers' third aggregation function, each at its own per-GPU size: short lines, same step, same clock
        result["extra"] = {}
        for name in ("sumvec_cfg1", "histogram", "average"):
            r = bench_prio3(args, name, 0, max(args.steps, 10), max(args.warmup, 2), rank, local_rank, world, dev, backend, cpu=False)
            result["extra"][name] = {"metric": r["metric"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
                                     "steps": r["steps"], "warmup": r["warmup"], "workload": r["config"]["workload"],
                                     "reports_per_gpu": r["config"]["reports_per_gpu"], "status_nonzero": r["config"]["status_nonzero"],
                                     "sort_and_keep_mask_in_step": r["config"]["sort_and_keep_mask_in_step"],
                                     "valu_frac": r["valu"]["frac"], "roofline": {k: r["roofline"].get(k) for k in ("bound", "kernel", "frac", "achieved", "peak", "unit")},
                                     "kernels": {k: round(v["ms_per_step"], 4) for k, v in r["kernels"].items()}}
        # the heavy-hitter servers' step (BASELINE configs 4 and 5: eval_and_check of every key chain + the sketch sums), every key
        # chain of the step distinct
        import copy
        for name in ("hh_t0.01", "hh_t0.001"):
            a2 = copy.copy(args)
            a2.workload, a2.steps, a2.warmup, a2.reports_per_gpu = name, max(args.steps, 5), max(args.warmup, 2), 0
            r = bench_hh(a2, rank, local_rank, world, dev, backend, emit=False)
            result["extra"][name] = {"metric": r["metric"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
                                     "steps": r["steps"], "warmup": r["warmup"], "workload": r["config"]["workload"],
                                     "key_chains_per_step_per_gpu": r["config"]["key_chains_per_step_per_gpu"],
                                     "distinct_key_chains": r["config"]["distinct_key_chains"], "status_nonzero": r["config"]["status_nonzero"],
                                     "roofline": {k: r["roofline"].get(k) for k in ("bound", "kernel", "frac", "achieved", "peak", "unit")},
                                     "kernels": {k: round(v["ms_per_step"], 4) for k, v in r["kernels"].items()}}
    if rank == 0 and result is not None:
        print(json.dumps(result))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return result


def bench_prio3(args, workload, reports_per_gpu, steps, warmup, rank, local_rank, world, dev, backend, cpu=True):
    """One Prio3 workload: make the clients, time `steps` steps, return rank 0's result line (None on other ranks)."""
    if world > 1:
        import torch.distributed as dist
    wl = WORKLOADS[workload]
    n = reports_per_gpu or wl["reports_per_gpu"]
    params = W.Prio3Params(wl["circuit"], wl["bits"], wl["length"], wl["chunk"])
    s = params.sizes
    K = max(1, min(args.unique, n // 2 if n >= 2 else 1))
    nthreads = max(1, min(16, os.cpu_count() or 1))

    vdaf = W.Prio3(params, device=local_rank)
    t_gen = time.time()
    if args.distinct:
        # ---- synthetic clients: n DISTINCT clients sharded on this GPU (wh_client_shard_many_gpu, byte-identical to the CPU
        # client); this server sees the Leader share of even clients and the Helper share of odd ones
        drank = rank if args.data_rank is None else args.data_rank
        rng = np.random.default_rng(1000 + drank)
        if wl["circuit"] == W.SUMVEC:
            meas = rng.integers(0, 1 << wl["bits"], (n, wl["length"]), dtype=np.uint64)
        elif wl["circuit"] == W.HISTOGRAM:
            meas = rng.integers(0, wl["length"], (n, 1), dtype=np.uint64)
        else:
            meas = rng.integers(0, 1 << 16, (n, 1), dtype=np.uint64)       # rng.gen::<u16>() as u128, client-batch-prio3 main.rs:101
        nonces = rng.integers(0, 256, (n, 16), dtype=np.uint8)
        rands = rng.integers(0, 256, (n, 144), dtype=np.uint8)
        L, H = s.leader_report_len, s.helper_report_len
        d_l = torch.empty(n * L, dtype=torch.uint8, device=dev)
        d_h = torch.empty(n * H, dtype=torch.uint8, device=dev)
        t_shard = time.time()
        vdaf.shard_batched_many_dev(meas, nonces, rands, drank << 40, d_l.data_ptr(), d_h.data_ptr())
        t_shard = time.time() - t_shard
        del meas
        pairs = n // 2
        d_blob = torch.empty(pairs * (L + H) + (L if n % 2 else 0) + 1, dtype=torch.uint8, device=dev)
        v = d_blob[: pairs * (L + H)].view(pairs, L + H)
        v[:, :L] = d_l.view(n, L)[0:2 * pairs:2]
        v[:, L:] = d_h.view(n, H)[1:2 * pairs:2]
        if n % 2:
            d_blob[pairs * (L + H): pairs * (L + H) + L] = d_l.view(n, L)[n - 1]
        idx = torch.arange(n + 1, device=dev, dtype=torch.int64)
        d_off = ((idx // 2) * (L + H) + (idx % 2) * L).contiguous()
        K = min(128, max(1, n // 2))
        off_h = d_off[: 2 * K + 1].cpu().numpy()
        head = d_blob[: int(off_h[-1])].cpu().numpy()
        tile_reports = [head[int(off_h[i]):int(off_h[i + 1])] for i in range(2 * K)]
        del d_l, d_h, v
        torch.cuda.empty_cache()
        n_distinct = n
    else:
        # ---- synthetic clients: K distinct clients sharded on the CPU and tiled; this server sees the Leader
        # share of even clients and the Helper share of odd ones (client-batch-prio3 main.rs:114-115)
        lr, hr, _ = make_reports(params, 2 * K, seed=1000 + rank, first_client_id=rank << 40, nthreads=nthreads)
        tile_reports = [(lr[i] if i % 2 == 0 else hr[i]) for i in range(2 * K)]
        tile = np.concatenate(tile_reports)
        tile_off = np.zeros(2 * K + 1, np.int64)
        tile_off[1:] = np.cumsum([len(r) for r in tile_reports])
        reps = (n + 2 * K - 1) // (2 * K)
        d_tile = torch.from_numpy(tile).to(dev)
        d_blob = d_tile.repeat(reps)
        offs = (torch.from_numpy(tile_off[:-1]).to(dev)[None, :] + (torch.arange(reps, device=dev, dtype=torch.int64) * int(tile_off[-1]))[:, None]).reshape(-1)
        d_off = torch.cat([offs, torch.tensor([reps * int(tile_off[-1])], device=dev, dtype=torch.int64)])[: n + 1].contiguous()
        n_distinct = min(n, 2 * K)
        t_shard = None
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen
    keep = np.ones(n, np.uint8)
    keep[:: 1000] = 0                      # ~0.1 % of the clients flagged by the group test
    d_keep = torch.from_numpy(keep).to(dev)

    if args.host_buffers:
        return bench_host_buffers(args, vdaf, params, n, d_blob, d_off, keep, world, rank)

    # one lane = one context (stream, workspace) with its own outputs; --in-flight 2 alternates the steps between two
    # lanes, as a server does that starts on the next batch while the last one is still in its query / aggregation phase
    class Lane:
        def __init__(self, v):
            self.vdaf = v
            self.ext = torch.cuda.ExternalStream(v.stream, device=dev)
            self.tags = torch.zeros(16 * n, dtype=torch.uint8, device=dev)
            self.ids = torch.zeros(16 * n, dtype=torch.uint8, device=dev)
            self.status = torch.zeros(n, dtype=torch.int32, device=dev)
            self.agg = torch.zeros(16 * s.output_len, dtype=torch.uint8, device=dev)
            self.total = torch.zeros(16 * s.output_len, dtype=torch.uint8, device=dev)
            if args.with_sort:
                self.order = torch.zeros(n, dtype=torch.int32, device=dev)
                self.sids, self.stags = torch.zeros_like(self.ids), torch.zeros_like(self.tags)
                self.keep = torch.zeros(n, dtype=torch.uint8, device=dev)
    lanes = [Lane(vdaf)] + [Lane(W.Prio3(params, device=local_rank)) for _ in range(max(1, args.in_flight) - 1)]
    # N > 1: the combine goes through the library's own RCCL binding (wh_comm_init / wh_prio3_allgather_reduce_dev); the gloo
    # rehearsal (several ranks on one GPU, which RCCL refuses) keeps torch's all-gather + wh_f128_sum_rows_dev
    wh_comm, wh_comm_why = world > 1 and backend == "nccl" and os.environ.get("WH_BENCH_COMBINE", "wh_comm") != "torch", ""
    if wh_comm:
        wh_comm, wh_comm_why = try_wh_comm([ln.vdaf for ln in lanes], dev,
                                           lambda v: allgather_reduce_dev(v, lanes[0].agg.data_ptr(), lanes[0].total.data_ptr()))
        if not wh_comm and rank == 0:
            print(f"bench.py: falling back to torch.distributed for the combine ({wh_comm_why})", file=sys.stderr)
    d_status = lanes[0].status
    torch.cuda.synchronize()
    bad_ids = []
    if args.with_sort:
        # the verdict a group test would return: the testing ids of the clients keep[] drops (from one untimed pass)
        ln = lanes[0]
        ln.vdaf.prepare_batch_dev(VK, d_blob.data_ptr(), d_off.data_ptr(), n, ln.tags.data_ptr(), ln.ids.data_ptr(), ln.status.data_ptr())
        ln.vdaf.synchronize()
        ids_h = ln.ids.view(n, 16)[torch.from_numpy(np.flatnonzero(keep == 0)).to(dev)].cpu().numpy()
        bad_ids = [int.from_bytes(r.tobytes(), "little") for r in ids_h]

    def step(i):
        ln = lanes[i % len(lanes)]
        if args.with_sort:
            # the whole GPU side of run_vdaf_prepare (main.rs:75-118): verify, order for the group test (the sort runs beside the
            # sponge kernels), mask from the group test's verdict, masked aggregate
            ln.vdaf.prepare_sorted_batch_dev(VK, d_blob.data_ptr(), d_off.data_ptr(), n, ln.tags.data_ptr(), ln.ids.data_ptr(), ln.status.data_ptr(),
                                             ln.order.data_ptr(), ln.sids.data_ptr(), ln.stags.data_ptr())
            ln.vdaf.keep_mask_dev(ln.ids.data_ptr(), n, bad_ids, ln.keep.data_ptr())
            ln.vdaf.aggregate_dev(ln.keep.data_ptr(), ln.agg.data_ptr())
        else:
            ln.vdaf.prepare_batch_dev(VK, d_blob.data_ptr(), d_off.data_ptr(), n, ln.tags.data_ptr(), ln.ids.data_ptr(), ln.status.data_ptr())
            ln.vdaf.aggregate_dev(d_keep.data_ptr(), ln.agg.data_ptr())
        if wh_comm:
            # RCCL all-gather of the raw limbs on the context stream, then the on-GPU mod-p row sum, inside the library
            allgather_reduce_dev(ln.vdaf, ln.agg.data_ptr(), ln.total.data_ptr())
        elif world > 1:
            with torch.cuda.stream(ln.ext):
                rows = all_gather_partials(ln.agg)
            ln.vdaf.sum_rows_dev(rows.data_ptr(), world, s.output_len, None, ln.total.data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        for ln in lanes:
            ln.vdaf.synchronize()

    for i in range(max(warmup, len(lanes) if warmup else 0)):
        step(i)
    barrier()
    for ln in lanes:
        ln.vdaf.set_profiling(True)
        ln.vdaf.kernel_times(reset=True)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    ktimes = {}
    for ln in lanes:
        for k, (ms, cnt) in ln.vdaf.kernel_times(reset=True).items():
            ktimes[k] = (ktimes.get(k, (0.0, 0))[0] + ms, ktimes.get(k, (0.0, 0))[1] + cnt)
        ln.vdaf.set_profiling(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_bad_status = int((d_status != 0).sum().item())
    # the result of the last step: this GPU's masked aggregate, or the combined one on every rank
    import hashlib
    agg_bytes = (lanes[(steps - 1) % len(lanes)].total if world > 1 else lanes[(steps - 1) % len(lanes)].agg).cpu().numpy().tobytes()
    # what leaves the GPU per step for the host's group test: the sorted ids and tags, 32 B per report -- timed on its own (it
    # is PCIe, not part of `value`), pinned destination, on the context stream
    d2h = None
    if args.with_sort and rank == 0:
        ln = lanes[0]
        h_ids = torch.empty(16 * n, dtype=torch.uint8).pin_memory()
        h_tags = torch.empty(16 * n, dtype=torch.uint8).pin_memory()
        # on torch's own stream (the context is idle here): the pinned-memory allocator remembers the streams a block was
        # used on, and the context's stream is gone by the time these tensors are freed
        h_ids.copy_(ln.sids, non_blocking=True); h_tags.copy_(ln.stags, non_blocking=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            h_ids.copy_(ln.sids, non_blocking=True); h_tags.copy_(ln.stags, non_blocking=True)
            torch.cuda.synchronize()
        d2h = {"bytes_per_step": 32 * n, "ms_per_step": 1e3 * (time.perf_counter() - t1) / 5,
               "note": "sorted testing ids + tags to pinned host memory for general_binary_split_test; outside the timed region and outside `value`"}

    # ---- per-kernel roofline numbers (rank 0), HIP events on the launch stream
    n_lead = (n + 1) // 2
    n_help = n // 2
    per = {k: n_lead * alg_bytes_per_report(s, 0)[k] + n_help * alg_bytes_per_report(s, 1)[k]
           for k in ("parse", "expand", "meas", "hash", "query", "aggregate", "chain", "sort")}
    kernels = {}
    for k, (ms, cnt) in ktimes.items():
        if cnt:
            kernels[k] = {"ms_per_step": ms / steps, "launches_per_step": cnt / steps}
            if k == "sort" and args.with_sort:
                kernels[k]["overlapped"] = "own stream, beside the sponge kernels (wh_prio3_prepare_sorted_batch_dev); not part of the step's sum"
            if k == "expand":
                kernels[k]["overlapped"] = "side stream, beside k_meas: the event time includes the wait for free wave slots; not part of the step's sum"
            if k == "hash" and not os.environ.get("WH_HASH_B_MAIN") and not os.environ.get("WH_NO_SIDE_STREAM"):
                kernels[k]["note"] = ("job A (joint / query randomness, barycentric weights) only; job B (hash of the public parts, "
                                      "RLC coefficients, testing ids) runs on the side stream beside k_meas, untimed")
            if k in per and ms > 0:      # algorithmic bytes of the step over the kernel's time: its own HBM roofline figure
                kernels[k]["alg_gb_per_s"] = per[k] / (ms / steps * 1e-3) / 1e9
                kernels[k]["hbm_frac"] = kernels[k]["alg_gb_per_s"] / HBM_PEAK_GBS
    # expand and hash have a launch on the side stream whose event time includes waiting for free slots
    dom = max((k for k in kernels if k in per and k not in ("expand", "sort")), key=lambda k: kernels[k]["ms_per_step"])
    launches = kernels[dom]["launches_per_step"]
    avg_ms = kernels[dom]["ms_per_step"] / launches
    bytes_per_launch = per[dom] / launches
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    # HBM traffic of the same kernel from the PMC passes committed under profiles/ (collected with
    # rocprofv3 --pmc in separate runs; only valid for the workload and size it was taken on)
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        ent = pmc.get(workload)
        if ent and ent["reports_per_gpu"] == n and ("k_" + dom) in ent["kernels"]:
            kk = ent["kernels"]["k_" + dom]
            traffic = (kk["fetch_bytes"] + kk["write_bytes"]) * (1.0 / launches if launches > 1 else 1.0)
            traffic_src = pmc["source"]
    except (OSError, ValueError, KeyError):
        pass
    # SHAKE128 work of the step, for the VALU side of the story
    def perms(role):
        IL, PL, VL = s.input_len, s.proof_len, s.verifier_len
        blk = lambda nbytes: -(-nbytes // 168)
        jr = blk(42 + 16 * IL + 1)
        qr = blk(25 + 32 * s.joint_rand_len + 32 * PL + 1)
        hs = blk(25 + 64 + 32 * VL + 1)
        rlc = 1 + blk(16 * (VL + 5)) - 1 + 1
        small = 2 + 2 + 1 + 2 + 1 + 1
        ex = (blk(16 * IL) + 2 * blk(16 * PL) + 3) if role == 1 else 0
        return jr + qr + hs + rlc + small + ex
    total_perms = n_lead * perms(0) + n_help * perms(1)
    blk_ = lambda nbytes: -(-nbytes // 168)
    meas_perms = (n_lead + n_help) * blk_(42 + 16 * s.input_len + 1) + n_help * (blk_(16 * s.input_len) + 1)   # k_meas alone
    meas_ms = kernels["meas"]["ms_per_step"] if "meas" in kernels else None
    if dom == "chain":                         # the one-launch form holds every sponge of the step
        meas_perms, meas_ms = total_perms, kernels["chain"]["ms_per_step"]

    result = None
    if rank == 0:
        value = world * n * steps / elapsed
        result = {
            "metric": "verified+aggregated client reports/sec (node), Prio3 " + {W.SUMVEC: "SumVec", W.HISTOGRAM: "Histogram", W.AVERAGE: "Average"}[wl["circuit"]],
            "value": value, "unit": "reports/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u128 (Field128 mod p) + u64 (Keccak lanes)", "data": "synthetic",
            "config": {"workload": wl["desc"], "reports_per_gpu": n, "distinct_clients": n_distinct, "clients_made_on": "gpu (wh_client_shard_many_gpu)" if args.distinct else "cpu, tiled",
                       "leader_report_bytes": s.leader_report_len, "helper_report_bytes": s.helper_report_len,
                       "roles": "alternating Leader/Helper", "inputs": "resident in HBM", "batches_in_flight": len(lanes), "sort_and_keep_mask_in_step": bool(args.with_sort),
                       "parallelism": f"reports sharded over {world} GPU(s); all-gather + mod-p reduce of partial aggregates",
                       "combine": ("none (one GPU)" if world == 1 else "wh_comm: the library's RCCL binding (wh_prio3_allgather_reduce_dev)" if wh_comm
                                   else f"torch.distributed all_gather ({backend}) + wh_f128_sum_rows_dev" + (f"; wh_comm unavailable: {wh_comm_why}" if wh_comm_why else "")),
                       "status_nonzero": n_bad_status, "aggregate_sha256": hashlib.sha256(agg_bytes).hexdigest(),
                       "aggregate_hex": agg_bytes.hex() if len(agg_bytes) <= 4096 else None, "client_gen_s": round(t_gen, 2),
                       "gpu_shard_clients_per_s": round(n / t_shard) if t_shard else None},
            "roofline": roofline_obj(dom, kernels, meas_perms, meas_ms, achieved, traffic, traffic_src, avg_ms, bytes_per_launch, launches),
            "valu": {"keccak_f1600_per_step": total_perms,
                     "keccak_f1600_per_s": total_perms * steps / elapsed,
                     "keccak_f1600_peak_per_s": KECCAK_PEAK,
                     "keccak_f1600_peak_measured_per_s": KECCAK_PEAK_MEASURED,
                     "frac": total_perms * steps / elapsed / KECCAK_PEAK,
                     "frac_of_measured": total_perms * steps / elapsed / KECCAK_PEAK_MEASURED,
                     "note": "whole step (query and aggregation included) against the permutation-only peaks of the roofline object"},
            "kernels": kernels,
        }
        if d2h:
            result["d2h_tags_ids"] = d2h
        if cpu:
            result["cpu_baseline"] = cpu_baseline(wl, tile_reports, nthreads, args.cpu_seconds)
    for ln in lanes:
        ln.vdaf.close()
    del d_blob, d_off, lanes
    torch.cuda.empty_cache()
    return result




def try_wh_comm(objs, dev, trial):
    """Bind the contexts to the library's own RCCL communicator (wh_comm_init) and run one trial collective.  Every rank
    reports whether that worked; unless all did, every rank falls back to torch.distributed's collective (also RCCL) followed by
    the on-GPU reduction -- the combine is the same arithmetic either way, and a rank must never wait on a collective that
    another rank has given up on.  Returns (use_wh_comm, why_not)."""
    import torch.distributed as dist

    def all_ok(flag):
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())
    why = ""
    try:
        from whisper_amd.dist import comm_unique_id
        comm_unique_id()                      # does RCCL load in this process at all?
        ok = True
    except Exception as e:                    # noqa: BLE001
        ok, why = False, f"RCCL binding: {e}"
    if not all_ok(ok):
        return False, why or "RCCL binding failed on another rank"
    # The library's communicator over more than one GPU is code that no test box could run.  Rehearse it in child processes
    # first: one per rank, own unique id, two small collectives, a time limit.  A child that hangs is killed and every rank
    # takes the torch.distributed path; a rank that hung inside its own communicator could only be killed from outside.
    import subprocess
    box = [comm_unique_id().hex() if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "wh_comm_probe.py")
    limit = float(os.environ.get("WH_BENCH_PROBE_SECONDS", "120"))
    try:
        child = subprocess.Popen([sys.executable, probe, box[0], str(dist.get_rank()), str(dist.get_world_size()), str(dev.index or 0)],
                                 stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        try:
            _, err = child.communicate(timeout=limit)
            ok = child.returncode == 0
            if not ok:
                why = f"rehearsal child exited {child.returncode}: {err.decode(errors='replace').strip().splitlines()[-1:]}"
        except subprocess.TimeoutExpired:
            child.kill()
            child.communicate()
            ok, why = False, f"rehearsal child did not finish within {limit:.0f} s"
    except OSError as e:
        ok, why = False, f"rehearsal child: {e}"
    if not all_ok(ok):
        return False, why or "the rehearsal of the library's communicator failed on another rank"
    try:
        for o in objs:
            comm_init_from_torch(o)
        ok = True
    except Exception as e:                    # noqa: BLE001
        ok, why = False, f"wh_comm_init: {e}"
    if not all_ok(ok):                        # before anyone enters a collective that a failed rank will never join
        return False, why or "wh_comm_init failed on another rank"
    try:
        for o in objs:
            trial(o)
            o.synchronize()
        ok = True
    except Exception as e:                    # noqa: BLE001
        ok, why = False, f"trial collective: {e}"
    if not all_ok(ok):
        return False, why or "trial collective failed on another rank"
    return True, ""


def roofline_obj(dom, kernels, meas_perms, meas_ms, hbm_achieved, traffic, traffic_src, avg_ms, bytes_per_launch, launches):
    """The dominant kernel against the resource that binds it.  k_meas (the measurement-share sponges) is bound by vector-ALU
    issue: achieved = its Keccak-f[1600] per launch over its HIP-event duration, peak = the issue-rate figure at the maximum
    clock; the measured ceiling of the bare instruction stream and the clock the kernel actually holds stand beside it.  Its
    HBM figure (algorithmic bytes, PMC traffic) is kept in `hbm`."""
    hbm = {"achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBS, "traffic": traffic,
           "traffic_source": traffic_src, "alg_bytes_per_launch": bytes_per_launch}
    if dom not in ("meas", "chain") or not meas_ms:
        return dict(hbm, bound="hbm", kernel="k_" + dom, avg_launch_ms=avg_ms)
    ach = meas_perms / (meas_ms * 1e-3) / 1e9
    return {"bound": "valu", "kernel": "k_" + dom, "achieved": ach, "peak": KECCAK_PEAK / 1e9, "unit": "G Keccak-f[1600]/s",
            "frac": ach / (KECCAK_PEAK / 1e9), "traffic": traffic,
            "peak_derivation": f"24 rounds x 180 instructions x 2 cycles (two VALU issue slots per four cycles and SIMD, measured: "
                               f"tools/inst_rate_gen.py) = {KECCAK_ISSUE_CYCLES} cycles per wavefront-permutation; 1024 SIMDs x 64 lanes x {MAX_CLOCK_HZ / 1e9} GHz",
            "peak_measured": KECCAK_PEAK_MEASURED / 1e9, "frac_of_measured": ach / (KECCAK_PEAK_MEASURED / 1e9),
            "peak_measured_source": "tools/keccak_bench.hip, profiles/r2b_keccak_bench.txt: same round code, nothing else, 6 waves/SIMD, power-limited at 1.95 GHz; "
                                    f"{KECCAK_PEAK_MEASURED_3_WAVES / 1e9} at the 3 waves/SIMD k_meas has",
            "frac_of_measured_3_waves": ach / (KECCAK_PEAK_MEASURED_3_WAVES / 1e9),
            "clock_ghz_in_kernel": K_MEAS_CLOCK_GHZ,
            "frac_of_peak_at_kernel_clock": ach / (KECCAK_PEAK / 1e9 * K_MEAS_CLOCK_GHZ * 1e9 / MAX_CLOCK_HZ),
            "clock_source": "s_memtime / s_memrealtime stamps of a diagnostic build, tools/meas_bench.hip, profiles/r2b_meas_bench.txt; board power "
                            "1.29 kW of 1.4 kW while the step runs (profiles/r2_power_probe.txt)",
            "keccak_f1600_per_launch": meas_perms, "avg_launch_ms": meas_ms, "hbm": hbm}


def bench_host_buffers(args, vdaf, params, n, d_blob, d_off, keep, world, rank):
    """The call INTEGRATION.md binds for a Rust server holding Vec<Vec<u8>>: wh_prio3_prepare_batch (host blob in, tags / ids /
    status out; chunked, double-buffered H2D under the verification) + wh_prio3_aggregate (host keep mask in, aggregate out)."""
    import ctypes as C
    s = params.sizes
    off = d_off.cpu().numpy().astype(np.uint64)
    nbytes = int(off[-1])
    if args.host_buffers == "pinned":
        p = C.c_void_p()
        if W.lib.wh_host_alloc(nbytes + 1, C.byref(p)):
            raise SystemExit("wh_host_alloc failed")
        blob = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes + 1,))
    else:
        blob = np.empty(nbytes + 1, np.uint8)
    step_b = 1 << 30
    for lo in range(0, nbytes, step_b):                     # device -> host in pieces (no second 19 GB staging tensor)
        hi = min(nbytes, lo + step_b)
        blob[lo:hi] = d_blob[lo:hi].cpu().numpy()
    del d_blob
    torch.cuda.empty_cache()

    def step():
        tags, ids, status = vdaf.prepare_batch_host(VK, blob, off)
        return tags, status, vdaf.aggregate(keep)

    for _ in range(max(1, args.warmup)):
        step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tags, status, agg = step()
    elapsed = time.perf_counter() - t0
    result = {
        "metric": "verified+aggregated client reports/sec (node), Prio3 SumVec, host buffers in and out (PCIe inclusive)",
        "value": n * args.steps / elapsed, "unit": "reports/s", "n_gpus": 1, "steps": args.steps, "warmup": max(1, args.warmup),
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u128 (Field128 mod p) + u64 (Keccak lanes)", "data": "synthetic",
        "config": {"workload": WORKLOADS[args.workload]["desc"], "reports_per_gpu": n, "inputs": f"host memory ({args.host_buffers})",
                   "entry_points": "wh_prio3_prepare_batch + wh_prio3_aggregate", "bytes_in_per_step": nbytes,
                   "chunk_mb": int(os.environ.get("WH_HOST_CHUNK_MB", "2048")), "status_nonzero": int((status != 0).sum())},
        "pcie": {"h2d_gb_per_s": nbytes * args.steps / elapsed / 1e9, "note": "report bytes over the whole step time (verification of chunk k runs under the copy of chunk k + 1)"},
    }
    vdaf.close()
    if args.host_buffers == "pinned":
        W.lib.wh_host_free(p)
    return result if rank == 0 else None


def bench_hh(args, rank, local_rank, world, dev, backend, emit=True):
    """server-hh's expansion + aggregation (server-hh/src/main.rs:236-283): per repetition, eval_and_check of every client's
    key chain, then the sum of the kept clients' bucket vectors; across GPUs an integer all-reduce of the sketches."""
    import torch.distributed as dist
    from whisper_amd import hh
    from whisper_amd.dist import all_reduce_sketch_u32
    wl = HH_WORKLOADS[args.workload]
    reps, N = wl["reps"], wl["buckets"]
    clients = args.reports_per_gpu or wl["clients_per_gpu"]
    depth = N.bit_length() - 1
    n = clients * reps                                   # key chains per step, repetition-major
    ln = hh.key_chain_len(depth)
    rng = np.random.default_rng(0x5748 + rank)
    # every key chain of the step from a distinct client generator run (wh_hh_gen_key_chain on the host cores, a few seconds);
    # --tiled makes --unique of them and repeats those instead (the round-1/2 lines: 128 key chains = 3.8 MB, an XCD's L2)
    K = n if args.distinct else max(1, min(args.unique, n))
    t_gen = time.time()
    uniq = np.zeros((K, ln), np.uint8)
    buckets = np.where(rng.integers(0, 2, (K, 258)) == 1, 1, 0xffffffff).astype(np.uint32)
    buckets[:, 257] = 1
    alphas = rng.integers(0, N, K)
    seeds = rng.integers(0, 256, (K, 32), dtype=np.uint8)

    def gen(lo, hi):
        for i in range(lo, hi):
            uniq[i], _ = hh.gen_key_chain(depth, i, int(alphas[i]), buckets[i], seeds[i].tobytes())
    from concurrent.futures import ThreadPoolExecutor
    nthreads = max(1, min(16, os.cpu_count() or 1))
    step_k = (K + 4 * nthreads - 1) // (4 * nthreads)
    with ThreadPoolExecutor(nthreads) as ex:                     # the generator is C code behind ctypes: the threads run side by side
        list(ex.map(lambda lo: gen(lo, min(K, lo + step_k)), range(0, K, step_k)))
    t_gen = time.time() - t_gen
    keys = uniq if K == n else uniq[np.arange(n) % K].copy()
    del uniq
    keys[:, :8] = ((rank << 40) + np.arange(n, dtype=np.uint64) % clients).view(np.uint8).reshape(n, 8)   # client ids
    h = hh.HeavyHitters(N, device=local_rank)
    ext = torch.cuda.ExternalStream(h.stream, device=dev)
    # the library's own RCCL binding; the gloo rehearsal keeps torch's; WH_BENCH_COMBINE=torch forces torch.distributed's collective
    wh_comm, wh_comm_why = world > 1 and backend == "nccl" and os.environ.get("WH_BENCH_COMBINE", "wh_comm") != "torch", ""
    d_trial = torch.zeros(64, dtype=torch.int32, device=dev)
    if wh_comm:
        wh_comm, wh_comm_why = try_wh_comm([h], dev, lambda o: hh_allreduce_u32_dev(o, d_trial.data_ptr(), 64))
        if not wh_comm and rank == 0:
            print(f"bench.py: falling back to torch.distributed for the combine ({wh_comm_why})", file=sys.stderr)
    d_keys = torch.from_numpy(keys).to(dev)
    d_dat = torch.empty((n, N, 258), dtype=torch.int32, device=dev) if os.environ.get("WH_BENCH_HH_MATERIALIZE") else None
    d_tags = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
    d_ids = torch.zeros((n, 16), dtype=torch.uint8, device=dev)
    d_status = torch.zeros(n, dtype=torch.int32, device=dev)
    keep = np.ones(n, np.uint8)
    keep[:: 100] = 0                                     # 1 % of the clients flagged by the group test
    d_keep = torch.from_numpy(keep).to(dev)
    d_acc = torch.zeros((reps, N, 258), dtype=torch.int32, device=dev)
    words = N * 258
    torch.cuda.synchronize()

    bad_index = np.flatnonzero(keep == 0).astype(np.uint64)
    fused = not os.environ.get("WH_BENCH_HH_MATERIALIZE")

    def step():
        if fused:
            # the sum of every key chain's buckets inside the expansion (one sketch per repetition), then the key chains the group
            # test rejected are expanded once more and taken out again: nothing is materialised
            h.eval_and_sketch_dev(VK, d_keys.data_ptr(), ln, n, clients, d_acc.data_ptr(), d_tags.data_ptr(), d_ids.data_ptr(), d_status.data_ptr())
            h.sketch_subtract_dev(VK, d_keys.data_ptr(), ln, bad_index, clients, d_acc.data_ptr())
        else:
            h.eval_and_check_dev(VK, d_keys.data_ptr(), ln, n, d_dat.data_ptr(), d_tags.data_ptr(), d_ids.data_ptr(), d_status.data_ptr())
            for r in range(reps):                            # one sketch per repetition (aggregate_hhs)
                h.bucket_sum_u32_dev(d_dat.data_ptr() + 4 * r * clients * words, d_keep.data_ptr() + r * clients, clients, words,
                                     d_acc.data_ptr() + 4 * r * words)
        if wh_comm:
            hh_allreduce_u32_dev(h, d_acc.data_ptr(), reps * words)
        elif world > 1:
            with torch.cuda.stream(ext):
                all_reduce_sketch_u32(d_acc)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        h.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    h.set_profiling(True)
    h.kernel_times(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    ktimes = h.kernel_times(reset=True)
    h.set_profiling(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_bad_status = int((d_status != 0).sum().item())
    import hashlib
    sketch_sha = hashlib.sha256(d_acc.cpu().numpy().tobytes()).hexdigest()      # the last step's (combined) sketches
    result = None
    if rank == 0:
        kernels = {k: {"ms_per_step": ms / args.steps, "launches_per_step": cnt / args.steps} for k, (ms, cnt) in ktimes.items() if cnt}
        if "hh_sum" in kernels and fused:
            kernels["hh_sum"]["overlapped"] = "second stream, under k_hh_support / k_hh_chain / k_hh_final; not part of the step's sum"
        main_ms = kernels["hh_main"]["ms_per_step"]          # all launches of the step: the batch, and the rejected key chains again
        # k_hh_main per key chain: the key chain read, the corrected leaf hashes and the bucket vector written -- into the caller's
        # array (materialised path) or into the context's sub-batch scratch, which k_hh_sum_u32 reads back on the second stream
        # (sketch path); the rejected key chains are read once more and go into the sketch by atomics
        main_bytes = n * (ln + 4 * words + 64 * N + 4 * 258 + 20) + (len(bad_index) * ln if fused else 0)
        # AES blocks the algorithm needs per key chain: tree, leaf hashes, 65 counters per leaf, proof chain,
        # 14 per support key, testing id; in k_hh_main: all but the chain and the support keys
        aes_all = 2 * (N - 1) + 2 * N + 65 * N + 4 * N + 257 * 14 + 1
        aes_main = 2 * (N - 1) + 2 * N + 65 * N + 1
        aes_sub = 2 * (N - 1) + 65 * N + 1                    # a subtract launch skips the leaf hashes
        main_blocks = n * aes_main + (len(bad_index) * aes_sub if fused else 0)
        # two bounds of the table AES, both from tools/aes_microbench.hip: instruction issue (per block and wavefront
        # 160 x (v_perm 4.2 + ds_read 2.6) + 16 x 2.6 + 80 x 2.3 cycles on one of 4 SIMDs) and the LDS itself
        # (176 ds_read_b32 per block, a CU serves 32 lanes per clock); the lower one binds
        issue_cycles = 160 * (4.2 + 2.6) + 16 * 2.6 + 80 * 2.3
        aes_peak = min(256 * 2.4e9 * 4 * 64 / issue_cycles, 256 * 2.4e9 * 32 / 176)
        traffic, traffic_src = None, None        # PMC passes committed under profiles/ (tools/hh_pmc.sh), same workload and size only
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f)
            ent = pmc.get(args.workload)
            if ent and ent["key_chains_per_gpu"] == n:
                traffic = ent["kernels"]["k_hh_main"]["fetch_bytes"] + ent["kernels"]["k_hh_main"]["write_bytes"]
                traffic_src = pmc["source"]
        except (OSError, ValueError, KeyError):
            pass
        value = world * n * args.steps / elapsed
        result = {
            "metric": "expanded+aggregated heavy-hitter key chains/sec (node), eval_and_check + sketch sums",
            "value": value, "unit": "key chains/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 (AES-128 state bytes) + u32 (counters mod 2^32)", "data": "synthetic",
            "config": {"workload": wl["desc"], "clients_per_gpu": clients, "repetitions": reps, "buckets": N,
                       "key_chains_per_step_per_gpu": n, "key_chain_bytes": ln, "distinct_key_chains": K, "key_chains_tiled": K != n,
                       "inputs": "resident in HBM", "status_nonzero": n_bad_status, "client_gen_s": round(t_gen, 2),
                       "sketch_sha256": sketch_sha,
                       "parallelism": f"key chains sharded over {world} GPU(s); integer all-reduce of the sketches",
                       "combine": ("none (one GPU)" if world == 1 else "wh_comm: the library's RCCL binding (wh_hh_allreduce_u32_dev)" if wh_comm
                                   else f"torch.distributed all_reduce ({backend})" + (f"; wh_comm unavailable: {wh_comm_why}" if wh_comm_why else ""))},
            "roofline": {"bound": "lds", "kernel": "k_hh_main", "achieved": main_blocks / (main_ms * 1e-3) / 1e9, "peak": aes_peak / 1e9,
                         "unit": "G AES-128 blocks/s", "frac": main_blocks / (main_ms * 1e-3) / aes_peak, "traffic": traffic,
                         "peak_derivation": "table AES: 176 ds_read_b32 per block at 32 lanes/clk/CU = 111.7 G blocks/s at 2.4 GHz; instruction "
                                            "issue (VALU and LDS instructions of a SIMD do not overlap) 119.7 G blocks/s; the lower binds",
                         "ms_per_step_all_launches": main_ms, "aes_blocks_per_step": main_blocks,
                         "sketch_sum": ("wh_hh_eval_and_sketch_batch_dev: a sub-batch's bucket vectors are written to a context scratch and summed by "
                                        "k_hh_sum_u32 on a second stream under the proof kernels (NOT fused into the expansion: atomics lost, profiles/"
                                        "r2_hh_fused_atomics.txt); the few rejected key chains are re-expanded and subtracted with atomic adds") if fused
                                       else "materialised by the caller, then k_hh_sum_u32 per repetition",
                         "hbm": {"achieved": main_bytes / (main_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": main_bytes / (main_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                                 "alg_bytes_per_step": main_bytes}},
            "valu": {"aes_blocks_per_step": n * aes_all, "aes_blocks_per_s": n * aes_all * args.steps / elapsed * world,
                     "k_hh_main_aes_blocks_per_s": main_blocks / (main_ms * 1e-3), "aes_bound_blocks_per_s": aes_peak,
                     "frac": main_blocks / (main_ms * 1e-3) / aes_peak,
                     "note": "bound = the lower of the LDS rate (176 table reads per block at 32 lanes/clk/CU: 111.7 G blocks/s) and "
                             "instruction issue (160 x (v_perm_b32 4.2 + ds_read_b32 2.6) + 16 x 2.6 + 80 x 2.3 cycles per block and "
                             "wavefront; VALU and LDS instructions of a SIMD do not overlap: 119.7 G blocks/s) "
                             "(tools/aes_microbench.hip, profiles/r1_aes_microbench.txt); frac is k_hh_main's"},
            "kernels": kernels,
        }
        if not args.no_cpu_baseline and emit and world == 1:
            result["cpu_baseline"] = cpu_baseline_hh(depth, keys, reps, clients, max(1, min(16, os.cpu_count() or 1)), args.cpu_seconds)
        if emit:
            print(json.dumps(result))
    if world > 1 and emit:
        dist.barrier()
        dist.destroy_process_group()
    h.close()
    return result


def cpu_baseline_hh(depth, keys, reps, clients, nthreads, budget_s):
    """The oracle's eval_and_check (AES-NI when the host has it) + bucket sums on the host cores over a bounded sample
    of the same key chains.  The only place the heavy-hitter bench touches oracle/."""
    import oracle_lib as O
    m = min(len(keys), nthreads * 64)
    done, spent = 0, 0.0
    while True:
        t0 = time.perf_counter()
        dat, _, _, status = O.hh_eval_and_check_batch(depth, VK, keys[:m], nthreads=nthreads)
        O.hh_bucket_sum_u32(dat)
        dt = time.perf_counter() - t0
        assert not status.any()
        done, spent = m, dt
        want = min(len(keys), int(m * min(8.0, budget_s / max(dt, 1e-3))))
        if dt >= budget_s / 2 or want <= m:
            break
        m = want
    return {"value": done / spent, "unit": "key chains/s", "cores": nthreads, "kind": "port",
            "sample": f"{done} key chains of the same workload in {spent:.2f} s, {nthreads} threads, "
                      f"{'AES-NI' if O.aes_backend() else 'portable AES'}; oracle/vdpf.c = C restatement of eval_and_check"}


def cpu_baseline(wl, reports, nthreads, budget_s):
    """The oracle (CPU restatement of the reference's algorithm as written: per-wire inverse FFT +
    Horner, both runs, SHAKE128 per report) timed on this host's cores over a bounded sample of the
    same reports.  The only place bench.py touches oracle/."""
    import oracle_lib as O
    op = O.params(wl["circuit"], wl["bits"], wl["length"], wl["chunk"])
    sample = list(reports[: max(2, 2 * nthreads)])
    done, spent = 0, 0.0
    while True:
        offsets = np.zeros(len(sample) + 1, np.uint64)
        offsets[1:] = np.cumsum([len(r) for r in sample])
        blob = np.concatenate(sample)
        t0 = time.perf_counter()
        _, _, status, outs = O.prio3_prepare_batch(op, VK, blob, offsets, nthreads=nthreads)
        O.prio3_aggregate(outs, O.sizes(op).output_len)
        dt = time.perf_counter() - t0
        assert not status.any()
        done, spent = len(sample), dt
        if dt >= budget_s / 2:
            break
        # grow the sample (cycling through the distinct reports) towards the time budget
        want = int(len(sample) * min(8.0, budget_s / max(dt, 1e-3)))
        want -= want % 2
        if want <= len(sample):
            break
        sample = [reports[i % len(reports)] for i in range(want)]
    return {"value": done / spent, "unit": "reports/s", "cores": nthreads, "kind": "port",
            "sample": f"{done} reports (alternating Leader/Helper) of the same workload in {spent:.2f} s, "
                      f"{nthreads} threads; oracle/ = C restatement of the reference algorithm as written"}


if __name__ == "__main__":
    main()
