"""Prio3::shard_batched on the GPU (wh_client_shard_many_gpu; vdaf/prio3.rs:537-991 + vdaf.rs:148-162) against the oracle's
shard_batched (pinned through its shard by the draft-07 vectors) and the CPU client, byte for byte, and then through the
verifier: the GPU-made reports are accepted and aggregate to the sum of the measurements."""
import numpy as np
import pytest

import oracle_lib as O
import whisper_amd as W
from common import det

pytestmark = pytest.mark.gpu

CASES = [
    ("sumvec_16_16_8", (W.SUMVEC, 16, 16, 8), 70),
    ("sumvec_cfg1", (W.SUMVEC, 16, 128, 64), 33),
    ("sumvec_large", (W.SUMVEC, 16, 1024, 128), 9),
    ("sumvec_ragged", (W.SUMVEC, 3, 11, 7), 5),
    ("sumvec_bits1", (W.SUMVEC, 1, 200, 16), 3),
    ("histogram_128_64", (W.HISTOGRAM, 0, 128, 64), 17),
    ("histogram_11_3", (W.HISTOGRAM, 0, 11, 3), 4),
    ("average_64", (W.AVERAGE, 64, 0, 0), 66),
    ("average_9", (W.AVERAGE, 9, 0, 0), 1),
]


def measurements(circ, bits, ln, n, tag):
    if circ == W.SUMVEC:
        m = np.frombuffer(det(tag, 8 * ln * n), np.uint64).reshape(n, ln)
        return m & np.uint64((1 << bits) - 1)
    raw = np.frombuffer(det(tag, 8 * n), np.uint64).reshape(n, 1)
    if circ == W.HISTOGRAM:
        return raw % np.uint64(ln)
    return raw if bits == 64 else raw & np.uint64((1 << bits) - 1)


@pytest.mark.parametrize("name,cfg,n", CASES, ids=[c[0] for c in CASES])
def test_gpu_shard_matches_cpu_client_and_oracle(name, cfg, n):
    circ, bits, ln, ch = cfg
    p = W.Prio3Params(circ, bits, ln, ch)
    op = O.params(circ, bits, ln, ch)
    meas = measurements(circ, bits, ln, n, b"sm" + name.encode())
    nonces = np.frombuffer(det(b"sn" + name.encode(), 16 * n), np.uint8).reshape(n, 16)
    rands = np.frombuffer(det(b"sr" + name.encode(), 144 * n), np.uint8).reshape(n, 144)
    first = (1 << 40) + 12345
    vdaf = W.Prio3(p)
    try:
        lr, hr = vdaf.shard_batched_many(meas, nonces, rands, first)
        lr_cpu, hr_cpu = W.shard_batched_many(p, meas, nonces, rands, first, nthreads=4)
        assert np.array_equal(lr, lr_cpu)
        assert np.array_equal(hr, hr_cpu)
        for i in (0, n - 1):
            want = O.prio3_shard_batched(op, [int(x) for x in meas[i]], nonces[i].tobytes(), rands[i].tobytes(), first + i)
            assert lr[i].tobytes() == want[0] and hr[i].tobytes() == want[1], i
        # sub-batches of one client give the same bytes
        lr1, hr1 = vdaf.shard_batched_many(meas[:1], nonces[:1], rands[:1], first)
        assert np.array_equal(lr1[0], lr[0]) and np.array_equal(hr1[0], hr[0])
        # and both servers accept them
        vk = bytes(det(b"svk", 16))
        res = [vdaf.prepare_encoded_keys(vk, [r.tobytes() for r in rep]) for rep in (lr, hr)]
        assert not res[0][2].any() and not res[1][2].any()
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    finally:
        vdaf.close()


def test_gpu_shard_rejects_out_of_range_measurement():
    p = W.Prio3Params(W.SUMVEC, 4, 8, 4)
    vdaf = W.Prio3(p)
    try:
        meas = np.zeros((3, 8), np.uint64)
        meas[1, 5] = 16
        z = np.zeros((3, 16), np.uint8); r = np.zeros((3, 144), np.uint8)
        with pytest.raises(W.WhError):
            vdaf.shard_batched_many(meas, z, r)
        meas[1, 5] = 15
        lr, hr = vdaf.shard_batched_many(meas, z, r)
        lr_cpu, hr_cpu = W.shard_batched_many(p, meas, z, r)
        assert np.array_equal(lr, lr_cpu) and np.array_equal(hr, hr_cpu)
        lr0, hr0 = vdaf.shard_batched_many(meas[:0], z[:0], r[:0])
        assert lr0.shape[0] == 0 and hr0.shape[0] == 0
    finally:
        vdaf.close()


def test_gpu_shard_random_shapes():
    """Seeded random circuit shapes (ragged last chunks, chunk > input, one-call circuits): the GPU sharder, the CPU client
    and the verifier agree on every one."""
    rng = np.random.default_rng(20240807)
    shapes = []
    for _ in range(14):
        bits, ln = int(rng.integers(1, 17)), int(rng.integers(1, 41))
        shapes.append((W.SUMVEC, bits, ln, int(rng.integers(1, bits * ln + 3))))
    for _ in range(8):
        ln = int(rng.integers(2, 61))
        shapes.append((W.HISTOGRAM, 0, ln, int(rng.integers(1, ln + 3))))
    for bits in (1, 2, 31, 33, 63):
        shapes.append((W.AVERAGE, bits, 0, 0))
    vk = bytes(det(b"rvk", 16))
    for k, (circ, bits, ln, ch) in enumerate(shapes):
        p = W.Prio3Params(circ, bits, ln, ch)
        n = 3
        meas = measurements(circ, bits, ln, n, b"rs%d" % k)
        nonces = np.frombuffer(det(b"rn%d" % k, 16 * n), np.uint8).reshape(n, 16)
        rands = np.frombuffer(det(b"rr%d" % k, 144 * n), np.uint8).reshape(n, 144)
        vdaf = W.Prio3(p)
        try:
            lr, hr = vdaf.shard_batched_many(meas, nonces, rands, 1000 * k)
            lr_cpu, hr_cpu = W.shard_batched_many(p, meas, nonces, rands, 1000 * k, nthreads=2)
            assert np.array_equal(lr, lr_cpu) and np.array_equal(hr, hr_cpu), (circ, bits, ln, ch)
            res = [vdaf.prepare_encoded_keys(vk, [r.tobytes() for r in rep]) for rep in (lr, hr)]
            assert not res[0][2].any() and not res[1][2].any(), (circ, bits, ln, ch)
            assert np.array_equal(res[0][0], res[1][0]), (circ, bits, ln, ch)
        finally:
            vdaf.close()
