"""Host-side mirror of the reference's heavy-hitter server surface.

Names follow the Rust crates: `BCSketch::gen_key_chain` / `eval_and_check` (hhcore/src/protocol.rs:17-28),
`aggregate_hhs` (server-hh/src/main.rs:53-110), `Bucket` = 256 data + sign_data + ctr counters of `AggRing = u32`
(hhcore/src/bucket.rs:5-12, hhcore/src/lib.rs:13).  Key chains cross this API as the bincode bytes the servers
receive (serialize/src/lib.rs:72-80); `eval_and_check` runs over a whole batch, as the GPU wants it.
"""
import ctypes as C
import hashlib

import numpy as np

from ._lib import lib, WhError, OPTIONS
from .prio3 import KERNELS

BUCKET_WORDS = 258  # STRING_SIZE + 2
MAX_DEPTH = 11      # num_buckets <= 2048 (server-hh/src/main.rs:428-436)


def key_chain_len(depth):
    n = int(lib.wh_hh_key_chain_len(depth))
    if not n:
        raise WhError(-1, f"depth {depth} outside 1..{MAX_DEPTH}")
    return n


def gen_key_chain(depth, client_id, alpha, bucket, rng_seed):
    """CPU client: `bucket.gen_key_chain(id, IntModN::<2^depth>::from_u16(alpha))` -> (alice bytes, bob bytes)."""
    bk = np.ascontiguousarray(bucket, dtype=np.uint32)
    if bk.shape != (BUCKET_WORDS,):
        raise ValueError("bucket must hold 258 u32")
    n = 16 + 1 + 16 + 8 + 18 * depth + 64 + 8 + 4 * BUCKET_WORDS + 8 + 257 * 111
    k0 = np.zeros(n, np.uint8); k1 = np.zeros(n, np.uint8)
    cid = np.frombuffer(int(client_id).to_bytes(16, "little"), np.uint8).copy()
    seed = np.frombuffer(bytes(rng_seed), np.uint8).copy()
    if seed.size != 32:
        raise ValueError("rng_seed must be 32 bytes")
    rc = lib.wh_hh_gen_key_chain(depth, cid.ctypes.data, alpha, bk.ctypes.data, seed.ctypes.data, k0.ctypes.data, k1.ctypes.data)
    if rc:
        raise WhError(rc, "gen_key_chain")
    return k0, k1


# ---- host side of the sketch: what the clients put in and what the servers read out (no GPU work) ----
def get_sign_and_bkt(v, num_buckets, rep_count):
    """hhcore/src/lib.rs:15-24: SHA-256(v || rep_count as u16 LE) -> (sign, bucket index)."""
    h = hashlib.sha256(bytes(v) + int(rep_count).to_bytes(2, "little")).digest()
    return bool(h[0] & 1), int.from_bytes(h[3:11], "little") % num_buckets


def bytes_to_bucket(s, sign):
    """hhcore/src/utils.rs:43-66: one counter per bit of the string (bits of a byte LSB first, common/src/lib.rs:172-226),
    +1 / -1 by the bit and flipped when `sign` is false; sign_data = +1 / -1; ctr = 1.  uint32[258]."""
    bits = np.unpackbits(np.frombuffer(bytes(s), np.uint8), bitorder="little")
    if bits.size != BUCKET_WORDS - 2:
        raise ValueError("client strings are 32 bytes")
    one, minus = np.uint32(1), np.uint32(0xffffffff)
    out = np.empty(BUCKET_WORDS, np.uint32)
    out[:256] = np.where(bits == 1, one if sign else minus, minus if sign else one)
    out[256] = one if sign else minus
    out[257] = 1
    return out


def _positive(v):
    return v < (1 << 16)                     # Ordered for u32, common/src/group.rs:139-142


def recover_bucket(bucket):
    """Bucket::recover + bits_to_bytes (hhcore/src/bucket.rs:112-120, common/src/lib.rs:208-217) -> 32 bytes."""
    b = np.asarray(bucket, np.uint32)
    bits = _positive(b[:256]) if _positive(b[256]) else ~_positive(b[:256])
    return np.packbits(bits.astype(np.uint8), bitorder="little").tobytes()


def recover(sketch, threshold, rep_count):
    """CountSketch::recover (hhcore/src/countsketch.rs:30-46): the strings of the buckets whose counter exceeds the
    threshold and that hash back to their own bucket.  sketch: uint32 [num_buckets, 258] (both servers' sketches added)."""
    sk = np.asarray(sketch, np.uint32)
    out = set()
    for i in np.nonzero(sk[:, 257] > np.uint32(threshold))[0]:
        cand = recover_bucket(sk[i])
        if get_sign_and_bkt(cand, sk.shape[0], rep_count)[1] == i:
            out.add(cand)
    return out


class HeavyHitters:
    """One server's heavy-hitter expansion + aggregation bound to one GPU."""

    def __init__(self, num_buckets, device=0):
        depth = int(num_buckets).bit_length() - 1
        if num_buckets != 1 << depth or not 1 <= depth <= MAX_DEPTH:
            raise ValueError("num_buckets must be a power of two in 2..2048")
        self.num_buckets, self.depth = num_buckets, depth
        self.key_len = key_chain_len(depth)
        self._ctx = C.c_void_p()
        rc = lib.wh_hh_ctx_create(device, C.byref(self._ctx))
        if rc:
            raise WhError(rc, "wh_hh_ctx_create (is a GPU present? there is no CPU fallback)")
        self.n_last = 0

    def close(self):
        if self._ctx:
            lib.wh_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc:
            raise WhError(rc, f"{what}: {lib.wh_last_error(self._ctx).decode()}")

    @property
    def stream(self):
        return lib.wh_ctx_stream(self._ctx)

    def synchronize(self):
        self._chk(lib.wh_ctx_synchronize(self._ctx), "synchronize")

    def set_option(self, name, value):
        """wh_ctx_set_option: a tuning / test knob of this context (names: whisper_amd._lib.OPTIONS)."""
        self._chk(lib.wh_ctx_set_option(self._ctx, OPTIONS[name], int(value)), f"set_option({name})")

    def eval_and_check(self, verify_key, key_chains):
        """key_chains: uint8 [n, stride >= key_len].  Returns (tags [n] of 64-char str, testing_ids [n] int, status int32[n]);
        the bucket vectors stay in HBM for aggregate_hhs / buckets."""
        keys = np.ascontiguousarray(key_chains, dtype=np.uint8)
        n, stride = keys.shape
        vk = np.frombuffer(bytes(verify_key), np.uint8).copy()
        tags = np.zeros((n, 64), np.uint8); ids = np.zeros((n, 16), np.uint8); status = np.zeros(n, np.int32)
        self._chk(lib.wh_hh_eval_and_check_batch(self._ctx, vk.ctypes.data, keys.ctypes.data, stride, n, self.depth,
                                                 tags.ctypes.data, ids.ctypes.data, status.ctypes.data), "hh_eval_and_check")
        self.n_last = n
        return ([bytes(t).decode("ascii") if s == 0 else "" for t, s in zip(tags, status)],
                [int.from_bytes(bytes(i), "little") for i in ids], status)

    def eval_and_check_dev(self, verify_key, d_keys, stride, n, d_dat, d_tags, d_ids, d_status):
        """Device pointers (ints); d_dat may be None to keep the bucket vectors in the context.  Asynchronous."""
        vk = np.frombuffer(bytes(verify_key), np.uint8).copy()
        self._chk(lib.wh_hh_eval_and_check_batch_dev(self._ctx, vk.ctypes.data, d_keys, stride, n, self.depth, d_dat, d_tags,
                                                     d_ids, d_status), "hh_eval_and_check_dev")
        self.n_last = n

    def eval_and_sketch_dev(self, verify_key, d_keys, stride, n, group_size, d_sketch, d_tags, d_ids, d_status, accumulate=False):
        """eval_and_check fused with aggregate_hhs' sum: key chain i is added into sketch i // group_size (device pointers, ints);
        no bucket vectors are materialised.  Asynchronous."""
        vk = np.frombuffer(bytes(verify_key), np.uint8).copy()
        self._chk(lib.wh_hh_eval_and_sketch_batch_dev(self._ctx, vk.ctypes.data, d_keys, stride, n, self.depth, group_size, d_sketch,
                                                      1 if accumulate else 0, d_tags, d_ids, d_status), "hh_eval_and_sketch_dev")
        self.n_last = 0

    def sketch_subtract_dev(self, verify_key, d_keys, stride, bad_index, group_size, d_sketch):
        """Take the key chains d_keys[bad_index] (rejected by the group test) out of the sketches they were summed into."""
        vk = np.frombuffer(bytes(verify_key), np.uint8).copy()
        idx = np.ascontiguousarray(bad_index, dtype=np.uint64)
        self._chk(lib.wh_hh_sketch_subtract_dev(self._ctx, vk.ctypes.data, d_keys, stride, self.depth, idx.ctypes.data if idx.size else None,
                                                idx.size, group_size, d_sketch), "hh_sketch_subtract_dev")

    def buckets(self):
        """Vec<Bucket<u32>> of every key chain of the last batch: uint32 [n, num_buckets, 258]."""
        out = np.zeros((self.n_last, self.num_buckets, BUCKET_WORDS), np.uint32)
        if self.n_last:
            self._chk(lib.wh_hh_buckets(self._ctx, out.ctypes.data), "hh_buckets")
        return out

    def aggregate_hhs(self, keep=None):
        """Sum of the kept clients' bucket vectors: uint32 [num_buckets, 258]."""
        acc = np.zeros((self.num_buckets, BUCKET_WORDS), np.uint32)
        k = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
        self._chk(lib.wh_hh_aggregate(self._ctx, None if k is None else k.ctypes.data, acc.ctypes.data), "hh_aggregate")
        return acc

    def aggregate_hhs_dev(self, d_keep, d_acc, accumulate=False):
        self._chk(lib.wh_hh_aggregate_dev(self._ctx, d_keep, d_acc, 1 if accumulate else 0), "hh_aggregate_dev")

    def bucket_sum_u32_dev(self, d_buckets, d_keep, n, words, d_acc, accumulate=False):
        """aggregate_hhs over caller-owned device buffers: buckets[n][words] u32 -> acc[words]."""
        fn = lib.wh_hh_bucket_accumulate_u32_dev if accumulate else lib.wh_hh_bucket_sum_u32_dev
        self._chk(fn(self._ctx, d_buckets, d_keep, n, words, d_acc), "hh_bucket_sum")

    def bucket_sum_f128_dev(self, d_buckets, d_keep, n, words, d_acc, accumulate=False):
        """The Field128 extension of aggregate_hhs: buckets[n][words] of 16-byte residues -> acc[words] (device pointers)."""
        fn = lib.wh_hh_bucket_accumulate_f128_dev if accumulate else lib.wh_hh_bucket_sum_f128_dev
        self._chk(fn(self._ctx, d_buckets, d_keep, n, words, d_acc), "hh_bucket_sum_f128")

    def sort_by_testing_id_dev(self, d_ids, n, d_order, d_tags=None, tag_bytes=64, d_sorted_ids=None, d_sorted_tags=None):
        """The server's stable sort by testing id (server-hh/src/main.rs:175, 266) on device buffers (raw pointers)."""
        self._chk(lib.wh_sort_by_testing_id_dev(self._ctx, d_ids, n, d_order, d_tags, tag_bytes, d_sorted_ids, d_sorted_tags),
                  "sort_by_testing_id_dev")

    def keep_mask_dev(self, d_ids, n, bad_ids, d_keep):
        """keep[i] = testing id i not in bad_ids (iterable of ints, the group test's verdict); device buffers as raw pointers."""
        bad = np.frombuffer(b"".join(int(x).to_bytes(16, "little") for x in bad_ids), np.uint8) if len(bad_ids) else np.zeros(0, np.uint8)
        self._chk(lib.wh_keep_mask_dev(self._ctx, d_ids, n, bad.ctypes.data if len(bad) else None, len(bad) // 16, d_keep), "keep_mask_dev")

    def set_profiling(self, on=True):
        self._chk(lib.wh_ctx_set_profiling(self._ctx, int(on)), "set_profiling")

    def kernel_times(self, reset=True):
        """{kernel: (milliseconds, launches)} accumulated by HIP events on the context stream since the last reset."""
        ms = np.zeros(len(KERNELS), np.float64)
        cnt = np.zeros(len(KERNELS), np.uint64)
        self._chk(lib.wh_ctx_kernel_times(self._ctx, ms.ctypes.data, cnt.ctypes.data, int(reset)), "kernel_times")
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(KERNELS)}
