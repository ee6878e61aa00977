"""CPU-side tests: the C-ABI library loads and exports every declared symbol, the CPU client
agrees with the oracle's sharder byte for byte, and the host group-testing mirror behaves as the
reference's own test expects (common/src/grouptest.rs:373-447)."""
import os
import random
import re

import numpy as np
import pytest

import oracle_lib as O
import whisper_amd as W
from common import VK, make_reports, det

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "whisper_gfx950.h")).read()
    names = set(re.findall(r"\b(wh_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 17
    for n in names:
        assert hasattr(W.lib, n), n
    # nothing under whisper_amd/ refers to the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "whisper_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in src and "oracle_lib" not in src and "wh_oracle" not in src, f


def test_sizes_match_oracle():
    for circ, bits, ln, ch in [(0, 16, 128, 64), (0, 16, 1024, 128), (0, 2, 5, 3), (1, 0, 128, 64), (1, 0, 11, 3), (2, 64, 0, 0)]:
        p = W.Prio3Params(circ, bits, ln, ch)
        s = O.sizes(O.params(circ, bits, ln, ch))
        for f in ("input_len", "proof_len", "verifier_len", "output_len", "joint_rand_len", "calls", "m", "arity",
                  "leader_report_len", "helper_report_len"):
            assert getattr(p.sizes, f) == getattr(s, f), f
    assert W.Prio3Params(0, 16, 128, 64).sizes.leader_report_len == 45225
    assert W.Prio3Params(0, 16, 128, 64).sizes.helper_report_len == 4345
    with pytest.raises(W.WhError):
        W.Prio3Params(0, 16, 0, 4)
    with pytest.raises(W.WhError):
        W.Prio3Params(0, 16, 4, 4, num_aggregators=1)
    # check_num_aggregators (vdaf/prio3.rs:2945-2957): at most 254, in the library and in the oracle
    assert W.Prio3Params(0, 16, 4, 4, num_aggregators=254).sizes.leader_report_len == O.sizes(O.params(0, 16, 4, 4, num_aggregators=254)).leader_report_len
    with pytest.raises(W.WhError):
        W.Prio3Params(0, 16, 4, 4, num_aggregators=255)
    with pytest.raises(ValueError):
        O.sizes(O.params(0, 16, 4, 4, num_aggregators=255))
    # a public share carries one 16-byte part per aggregator, twice per report (vdaf.rs:148-162)
    for na in (3, 5):
        p3, s3 = W.Prio3Params(0, 16, 4, 4, num_aggregators=na), O.sizes(O.params(0, 16, 4, 4, num_aggregators=na))
        p2 = W.Prio3Params(0, 16, 4, 4)
        assert p3.sizes.leader_report_len == s3.leader_report_len == p2.sizes.leader_report_len + 32 * (na - 2)
        assert p3.sizes.helper_report_len == s3.helper_report_len == p2.sizes.helper_report_len + 32 * (na - 2)


@pytest.mark.parametrize("na", (3, 4))
@pytest.mark.parametrize("circ,bits,ln,ch", [(0, 16, 3, 7), (1, 0, 11, 3), (2, 8, 0, 0)])
def test_client_matches_oracle_sharder_many_aggregators(circ, bits, ln, ch, na):
    """shard_batched is generic in the number of aggregators (prio3.rs:537-991; its own test runs three, :3077-3085): the CPU
    client and the oracle's sharder agree byte for byte, and the oracle's prepare_batched on every aggregator's report gives
    proof tags that cancel (vdaf.rs:889-897; prepare_encoded_key negates a Helper's, main.rs:53-59)."""
    from common import measurements
    p = W.Prio3Params(circ, bits, ln, ch, num_aggregators=na)
    op = O.params(circ, bits, ln, ch, num_aggregators=na)
    n = 3
    meas = measurements(p, n, 9)
    nonces = det(b"nonce9", 16 * n).reshape(n, 16)
    rands = det(b"rand9", 16 * (5 * na - 1) * n).reshape(n, -1)
    lr, hr = W.shard_batched_many(p, meas, nonces, rands, 11, 2)
    assert hr.shape == (n, na - 1, p.sizes.helper_report_len)
    vk = bytes(range(16))
    for i in range(n):
        l2, h2 = O.prio3_shard_batched(op, [int(x) for x in meas[i]], nonces[i].tobytes(), rands[i].tobytes(), 11 + i)
        assert lr[i].tobytes() == l2 and hr[i].tobytes() == h2
        tags = []
        for a in range(na):
            rc, tag, tid, out = O.prio3_prepare_encoded_key(op, vk, l2 if a == 0 else hr[i, a - 1].tobytes())
            assert rc == 0
            tags.append(int.from_bytes(tag, "little"))
        assert tags[0] == sum(tags[1:]) % O.P128


@pytest.mark.parametrize("circ,bits,ln,ch", [(0, 2, 5, 3), (0, 16, 3, 7), (0, 16, 16, 8), (1, 0, 4, 2), (1, 0, 11, 3), (2, 64, 0, 0), (2, 8, 0, 0)])
def test_client_matches_oracle_sharder(circ, bits, ln, ch):
    p = W.Prio3Params(circ, bits, ln, ch)
    op = O.params(circ, bits, ln, ch)
    lr, hr, meas = make_reports(p, 3, seed=5, first_client_id=2**40 + 3, nthreads=2)
    nonces = det(b"nonce5", 48).reshape(3, 16)
    rands = det(b"rand5", 432).reshape(3, 144)
    for i in range(3):
        l2, h2 = O.prio3_shard_batched(op, [int(x) for x in meas[i]], nonces[i].tobytes(), rands[i].tobytes(), 2**40 + 3 + i)
        assert lr[i].tobytes() == l2
        assert hr[i].tobytes() == h2


def test_sampler_skip_branch_client_and_oracle_agree():
    """The sampler's skip branch (prng.rs:73-99) with the test bound lowered to ~p/3: the CPU client
    and the oracle's sharder still agree byte for byte, and the oracle's two aggregators still get
    tags that cancel (vdaf.rs:889-897)."""
    bound = O.P128 // 3 + 999
    p = W.Prio3Params(0, 16, 16, 8)
    op = O.params(0, 16, 16, 8)
    try:
        O.set_test_sampler_bound(bound)
        W.client_set_test_sampler_bound(bound)
        lr, hr, meas = make_reports(p, 3, seed=9, first_client_id=77, nthreads=2)
        nonces = det(b"nonce9", 48).reshape(3, 16)
        rands = det(b"rand9", 432).reshape(3, 144)
        for i in range(3):
            l2, h2 = O.prio3_shard_batched(op, [int(x) for x in meas[i]], nonces[i].tobytes(), rands[i].tobytes(), 77 + i)
            assert lr[i].tobytes() == l2 and hr[i].tobytes() == h2
            ta, _, sa, _ = O.prio3_prepare_batch(op, VK, np.frombuffer(l2, np.uint8), np.array([0, len(l2)], np.uint64))
            tb, _, sb, _ = O.prio3_prepare_batch(op, VK, np.frombuffer(h2, np.uint8), np.array([0, len(h2)], np.uint64))
            assert not sa.any() and not sb.any() and ta.tobytes() == tb.tobytes()
        # and the unlowered reports differ (the hook really changes the stream)
        O.set_test_sampler_bound(None)
        l3, _ = O.prio3_shard_batched(op, [int(x) for x in meas[0]], nonces[0].tobytes(), rands[0].tobytes(), 77)
        assert l3 != lr[0].tobytes()
    finally:
        O.set_test_sampler_bound(None)
        W.client_set_test_sampler_bound(None)


def test_group_testing_mirror():
    # same scenario as grouptest.rs:373-447: 300 tags, 3 corrupted and 2 deleted on Bob's side
    rnd = random.Random(7)
    ids = sorted(rnd.getrandbits(128) for _ in range(300))
    tags0 = [str(i).encode() for i in range(300)]
    tags1 = list(tags0)
    for i in (20, 68, 100):
        tags1[i] = b"INCONSISTENCY"
    p0 = [W.ClientProofTag(i, t) for i, t in zip(ids, tags0)]
    p1 = [W.ClientProofTag(i, t) for i, t in zip(ids, tags1)]
    for i in (200, 4):
        del p1[i]
    for size in (32, 16):
        bad0, bad1, rounds = W.general_binary_split_test(p0, p1, bytes([1] * 16), 5, size)
        assert len(bad0) == 5 and len(bad1) == 3
        assert bad1 == {ids[20], ids[68], ids[100]}
        assert bad0 == bad1 | {ids[200], ids[4]}
    assert W.general_binary_split_test(p0, p0, bytes(16), 5, 16) == (set(), set(), 1)


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(W.WhError):
        W.new_sum_vec_256(2, 16, 8, 4)


def test_bucket_recover_like_the_reference():
    """hhcore/src/bucket.rs:153-207 (test_bucket_recover) with seeded noise: the string with the most votes in a bucket
    is read back, sign handling included; and a sketch reads back only strings that hash to their own bucket."""
    from whisper_amd import hh
    s1, s2 = b"hello my name is curious george!", b"asdfiej;lj2938psh,mcs932shdfpqw8"
    rng = np.random.default_rng(0)
    noise = lambda: hh.bytes_to_bucket(bytes(rng.integers(48, 122, 32, dtype=np.uint8)), bool(rng.integers(0, 2)))
    b = (100 * hh.bytes_to_bucket(s1, False) + 60 * hh.bytes_to_bucket(s2, True)).astype(np.uint32)
    for _ in range(30):
        b = (b + noise()).astype(np.uint32)
    assert hh.recover_bucket(b) == s1
    b = (b + 160 * hh.bytes_to_bucket(s2, True)).astype(np.uint32)
    for _ in range(80):
        b = (b + noise()).astype(np.uint32)
    assert hh.recover_bucket(b) == s2
    # CountSketch::recover: one heavy string in its own bucket of a 64-bucket sketch, repetition 3
    sign, idx = hh.get_sign_and_bkt(s1, 64, 3)
    sk = np.zeros((64, 258), np.uint32)
    sk[idx] = 50 * hh.bytes_to_bucket(s1, sign)
    sk[(idx + 1) % 64] = 50 * hh.bytes_to_bucket(s1, sign)        # the same votes in the wrong bucket are not reported
    assert hh.recover(sk, 49, 3) == {s1}
    assert hh.recover(sk, 50, 3) == set()


def test_entry_points_reject_a_null_context_without_touching_a_gpu():
    """Argument checks come before any HIP call: every batch entry point returns WH_ERR_PARAM (-1) for a null context."""
    from whisper_amd._lib import lib
    n = 4
    buf = np.zeros(4096, np.uint8)
    p = buf.ctypes.data
    assert lib.wh_prio3_prepare_batch(None, p, p, p, n, p, p, p) == -1
    assert lib.wh_prio3_prepare_batch_dev(None, p, p, p, n, p, p, p) == -1
    assert lib.wh_prio3_aggregate(None, p, p) == -1
    assert lib.wh_flp_query_batch(None, p, p, p, p, 1, n, p, p) == -1
    assert lib.wh_flp_prove_batch(None, p, p, p, n, p, p) == -1
    assert lib.wh_client_shard_many_gpu(None, p, 1, p, p, 0, n, p, p) == -1
    assert lib.wh_sort_by_testing_id_dev(None, p, n, p, None, 0, None, None) == -1
    assert lib.wh_sort_by_testing_id(None, p, n, p) == -1
    assert lib.wh_keep_mask_dev(None, p, n, p, 1, p) == -1
    assert lib.wh_hh_eval_and_check_batch(None, p, p, 64, n, 9, p, p, p) == -1
    assert lib.wh_hh_aggregate(None, p, p) == -1


def test_missing_rccl_is_an_error_code_not_a_crash():
    """A host without RCCL (here: WH_RCCL_LIBRARY names a file that does not exist, so only that file is tried): the
    communicator entry points return WH_ERR_UNSUPPORTED and wh_comm_why carries the loader's message -- bench.py's probe
    ('does RCCL load in this process at all?') and its torch.distributed fallback depend on that return.  The loader's
    message is read with ONE dlerror() call (a second call returns NULL; building a std::string from it crashed)."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "import whisper_amd._lib as L\n"
        "buf = (C.c_uint8 * 128)()\n"
        "rc = L.lib.wh_comm_unique_id(buf)\n"
        "why = L.lib.wh_comm_why().decode()\n"
        "rc2 = L.lib.wh_comm_unique_id(buf)\n"
        "print(rc, rc2, why)\n"
        "sys.exit(0 if rc == -103 and rc2 == -103 and 'RCCL not found' in why and 'no_such_dir' in why else 1)\n")
    env = dict(os.environ, WH_RCCL_LIBRARY="/no_such_dir/librccl.so.1", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stdout + r.stderr


def test_context_knobs_are_settable_without_a_gpu_error_path():
    """wh_ctx_set_option / wh_ctx_get_option reject a null context and an unknown option before anything else."""
    import ctypes as C
    v = C.c_uint64()
    assert W.lib.wh_ctx_set_option(None, 0, 1) == -1 and W.lib.wh_ctx_get_option(None, 0, C.byref(v)) == -1
