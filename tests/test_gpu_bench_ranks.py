"""bench.py's world > 1 branch on the GPU: two ranks share the one card (gloo rendezvous, because RCCL refuses two ranks on one
device), each verifies its own shard of the reports through the C ABI, the partial aggregates are combined and the line carries
the whole job.  The RCCL form of the combine (wh_comm_init / wh_prio3_allgather_reduce_dev) is the same branch with world = 1
in tests/test_gpu_api_state.py; RCCL over two GPUs has not run anywhere yet."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, extra):
    env = dict(os.environ, WH_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 prints ONE line
    return json.loads(lines[0])


def test_bench_two_ranks_one_gpu_prio3():
    one = _run(1, ["--workload", "histogram", "--reports-per-gpu", "3000"])
    two = _run(2, ["--workload", "histogram", "--reports-per-gpu", "3000"])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert two["config"]["reports_per_gpu"] == 3000 and two["config"]["status_nonzero"] == 0
    # whole-job value: both shards counted (the two ranks share one card, so no speed-up is asserted)
    assert two["value"] > 0 and abs(two["value"] * two["ms_per_step"] * 1e-3 - 2 * 3000) < 1
    assert abs(one["value"] * one["ms_per_step"] * 1e-3 - 3000) < 1
    assert two["roofline"]["bound"] in ("valu", "hbm") and "cpu_baseline" not in two
    assert "all_gather (gloo)" in two["config"]["combine"] and one["config"]["combine"] == "none (one GPU)"
    assert two["config"]["sort_and_keep_mask_in_step"] is True


def test_two_rank_combined_aggregate_is_the_sum_of_the_shards_aggregates():
    """The bytes, not the rate: the aggregate the two-rank run ends with (all-gather of both partials + on-GPU mod-p sum) equals the
    element-wise sum mod p of what two one-rank runs compute over rank 0's and rank 1's shard (Aggregatable::merge,
    vdaf.rs:658-676; the reference's property run_vdaf_batched checks, vdaf.rs:889-897, is on the same sums)."""
    P128 = 2**128 - 7 * 2**66 + 1
    args = ["--workload", "histogram", "--reports-per-gpu", "3000"]
    parts = [_run(1, args + ["--data-rank", str(r)])["config"]["aggregate_hex"] for r in (0, 1)]
    two = _run(2, args)["config"]["aggregate_hex"]
    assert parts[0] != parts[1] and len(two) == 2 * 16 * 128
    dec = lambda h: [int.from_bytes(bytes.fromhex(h[32 * e:32 * e + 32]), "little") for e in range(len(h) // 32)]
    a, b, t = dec(parts[0]), dec(parts[1]), dec(two)
    assert all(x < P128 for x in t)
    assert t == [(x + y) % P128 for x, y in zip(a, b)]
    # 3 000 Histogram clients per shard, 3 of each dropped by the keep mask: the shares of the two ROLES cancel only across the
    # two servers, so nothing more can be said about one server's sums -- except that a different shard gives different bytes


def test_comm_probe_child_sequence_world_one():
    """tools/wh_comm_probe.py is what bench.py --gpus N starts per rank before it trusts the library's RCCL binding: the same
    child, world = 1 (its own unique id), must bind RCCL, run both collectives and exit 0 on a one-GPU box."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wh_comm_probe.py"), "self", "0", "1", "0"], capture_output=True,
                       text=True, timeout=200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]


def test_bench_two_ranks_one_gpu_heavy_hitters():
    """The heavy-hitter workload's world > 1 branch: key chains sharded by rank, sketches combined by an integer all-reduce."""
    two = _run(2, ["--workload", "hh_t0.01", "--reports-per-gpu", "8", "--unique", "16"])
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["unit"] == "key chains/s"
    n = two["config"]["key_chains_per_step_per_gpu"]
    assert n == 8 * two["config"]["repetitions"] and two["config"]["status_nonzero"] == 0
    assert abs(two["value"] * two["ms_per_step"] * 1e-3 - 2 * n) < 1
    assert two["roofline"]["bound"] == "lds" and "all_reduce" in two["config"]["combine"]
