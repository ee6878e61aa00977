"""Shared test helpers: deterministic synthetic clients (client-batch-prio3/src/main.rs:67-153)."""
import hashlib

import numpy as np

import whisper_amd as W

MASTER_SEED = 0x5748495350455231  # "WHISPER1" 
VK = hashlib.shake_128(b"verify-key" + MASTER_SEED.to_bytes(8, "little")).digest(16)


def det(tag, n):
    return np.frombuffer(hashlib.shake_128(tag).digest(n), dtype=np.uint8)


def measurements(params, n, seed):
    c = params.c
    raw = np.frombuffer(hashlib.shake_128(b"meas%d" % seed).digest(8 * n * max(1, c.len if c.circuit == W.SUMVEC else 1)), dtype=np.uint64)
    if c.circuit == W.SUMVEC:
        mask = np.uint64((1 << c.bits) - 1) if c.bits < 64 else np.uint64(2**64 - 1)
        return (raw.reshape(n, c.len) & mask).copy()
    if c.circuit == W.HISTOGRAM:
        return (raw.reshape(n, 1) % np.uint64(c.len)).copy()
    return (raw.reshape(n, 1) & np.uint64((1 << min(c.bits, 16)) - 1)).copy()   # rng.gen::<u16>() as u128, main.rs:101


def make_reports(params, n, seed=1, first_client_id=0, nthreads=8):
    """n clients -> (leader_reports [n,L], helper_reports [n,H], measurements)."""
    meas = measurements(params, n, seed)
    nonces = det(b"nonce%d" % seed, 16 * n).reshape(n, 16)
    rands = det(b"rand%d" % seed, 144 * n).reshape(n, 144)
    lr, hr = W.shard_batched_many(params, meas, nonces, rands, first_client_id, nthreads)
    return lr, hr, meas


def server_views(lr, hr, bad=()):
    """What Alice and Bob receive: the meta-client alternates roles (main.rs:114-115; here per
    client instead of per socket) and tampers Bob's copy of the bad clients (main.rs:146-149)."""
    n = lr.shape[0]
    alice, bob = [], []
    for i in range(n):
        a, b = (lr[i], hr[i]) if i % 2 == 0 else (hr[i], lr[i])
        a, b = a.tobytes(), bytearray(b.tobytes())
        if i in bad:
            b[-16] = (b[-16] + 1) & 0xFF
        alice.append(a)
        bob.append(bytes(b))
    return alice, bob


def blob_of(reports):
    offsets = np.zeros(len(reports) + 1, np.uint64)
    offsets[1:] = np.cumsum([len(r) for r in reports])
    return np.frombuffer(b"".join(reports), dtype=np.uint8), offsets
