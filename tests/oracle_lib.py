"""ctypes view of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load this module; the
product package (whisper_amd/) never does.
"""
import ctypes as C
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = C.CDLL(os.path.join(_ROOT, "oracle", "liboracle.so"))

SUMVEC, HISTOGRAM, AVERAGE = 0, 1, 2
P128 = 2**128 - 7 * 2**66 + 1

OK, ERR_PARAM, ERR_SHORT, ERR_MODULUS, ERR_AGG_ID, ERR_NUM_QUERIES, ERR_QUERY_RAND = 0, -1, -2, -3, -4, -5, -6


class Params(C.Structure):
    _fields_ = [("circuit", C.c_uint32), ("bits", C.c_uint32), ("len", C.c_uint32), ("chunk", C.c_uint32),
                ("num_aggregators", C.c_uint32), ("id", C.c_uint32)]


class Sizes(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "input_len", "proof_len", "verifier_len", "output_len", "joint_rand_len", "prove_rand_len",
        "query_rand_len", "calls", "m", "arity", "leader_report_len", "helper_report_len")]


def params(circuit, bits=0, length=0, chunk=0, num_aggregators=2, alg_id=0):
    return Params(circuit, bits, length, chunk, num_aggregators, alg_id)


def sizes(p):
    s = Sizes()
    rc = _lib.wo_sizes_of(C.byref(p), C.byref(s))
    if rc:
        raise ValueError(f"bad params rc={rc}")
    return s


def _buf(n):
    return (C.c_uint8 * n)()


def _b(x):
    return (C.c_uint8 * len(x)).from_buffer_copy(bytes(x))


def f_enc(v):
    return int(v % P128).to_bytes(16, "little")


def f_dec(b):
    return int.from_bytes(bytes(b), "little")


def f_mul(a, b):
    o = _buf(16); _lib.wo_f128_mul(_b(f_enc(a)), _b(f_enc(b)), o); return f_dec(o)


def f_add(a, b):
    o = _buf(16); _lib.wo_f128_add(_b(f_enc(a)), _b(f_enc(b)), o); return f_dec(o)


def f_sub(a, b):
    o = _buf(16); _lib.wo_f128_sub(_b(f_enc(a)), _b(f_enc(b)), o); return f_dec(o)


def f_inv(a):
    o = _buf(16); _lib.wo_f128_inv(_b(f_enc(a)), o); return f_dec(o)


def f_constants():
    p, r2, g = _buf(16), _buf(16), _buf(16)
    mu = C.c_uint64()
    _lib.wo_f128_constants(p, r2, g, C.byref(mu))
    return f_dec(p), f_dec(r2), f_dec(g), mu.value


def f_root(l):
    o = _buf(16); _lib.wo_f128_root(C.c_uint(l), o); return f_dec(o)


def shake128(data, n):
    o = _buf(n); _lib.wo_shake128(_b(data) if data else None, C.c_size_t(len(data)), o, C.c_size_t(n)); return bytes(o)


def set_test_sampler_bound(bound=None):
    """Test hook: chunks >= bound are skipped by the sampler (None restores p)."""
    _lib.wo_set_test_sampler_bound(None if bound is None else _b(int(bound).to_bytes(16, "little")))


def xof_expand(seed, dst, binder, n):
    o = _buf(16 * n)
    _lib.wo_xof_expand_f128(_b(seed), _b(dst), C.c_size_t(len(dst)), _b(binder) if binder else None,
                            C.c_size_t(len(binder)), C.c_size_t(n), o)
    return bytes(o)


def xof_derive_seed(seed, dst, binder):
    o = _buf(16)
    _lib.wo_xof_derive_seed(_b(seed), _b(dst), C.c_size_t(len(dst)), _b(binder) if binder else None,
                            C.c_size_t(len(binder)), o)
    return bytes(o)


def fft(vals, size, inverse=False):
    inp = b"".join(f_enc(v) for v in vals)
    o = _buf(16 * size)
    fn = _lib.wo_ifft_f128 if inverse else _lib.wo_fft_f128
    rc = fn(_b(inp), C.c_size_t(len(vals)), C.c_size_t(size), o)
    assert rc == 0
    return [f_dec(bytes(o)[16 * i:16 * i + 16]) for i in range(size)]


def _meas(meas):
    if isinstance(meas, int):
        meas = [meas]
    return (C.c_uint64 * len(meas))(*meas)


def flp_encode(p, meas):
    s = sizes(p); o = _buf(16 * s.input_len)
    rc = _lib.wo_flp_encode(C.byref(p), _meas(meas), o); assert rc == 0, rc
    return bytes(o)


def flp_prove(p, inp, prove_rand, joint_rand):
    s = sizes(p); o = _buf(16 * s.proof_len)
    rc = _lib.wo_flp_prove(C.byref(p), _b(inp), _b(prove_rand), _b(joint_rand), o); assert rc == 0, rc
    return bytes(o)


def flp_query(p, inp, proof, query_rand, joint_rand, num_shares):
    s = sizes(p); o = _buf(16 * s.verifier_len)
    rc = _lib.wo_flp_query(C.byref(p), _b(inp), _b(proof), _b(query_rand), _b(joint_rand), C.c_uint32(num_shares), o)
    return rc, bytes(o)


def flp_decide(p, verifier):
    return bool(_lib.wo_flp_decide(C.byref(p), _b(verifier)))


def flp_truncate(p, inp):
    s = sizes(p); o = _buf(16 * s.output_len)
    rc = _lib.wo_flp_truncate(C.byref(p), _b(inp), o); assert rc == 0
    return bytes(o)


def prio3_rand_size(p):
    """Prio3::random_size with joint randomness (prio3.rs:495-513): per Helper two share seeds and a blind,
    the Leader's blind, the prove-rand seed."""
    return 16 * (3 * (p.num_aggregators - 1) + 2)


def prio3_shard(p, meas, nonce, rand):
    s = sizes(p); na = p.num_aggregators
    ps = _buf(16 * na); ls = _buf(16 * (s.input_len + s.proof_len) + 16); hs = _buf(48 * (na - 1))
    rc = _lib.wo_prio3_shard(C.byref(p), _meas(meas), _b(nonce), _b(rand), ps, ls, hs); assert rc == 0, rc
    hs = bytes(hs)
    return bytes(ps), [bytes(ls)] + [hs[48 * i:48 * i + 48] for i in range(na - 1)]


def prio3_prepare_init(p, verify_key, agg_id, nonce, public_share, input_share):
    s = sizes(p)
    prep = _buf(16 * s.verifier_len + 16); out = _buf(16 * s.output_len)
    rc = _lib.wo_prio3_prepare_init(C.byref(p), _b(verify_key), C.c_uint32(agg_id), _b(nonce), _b(public_share),
                                    _b(input_share), C.c_size_t(len(input_share)), prep, out)
    return rc, bytes(prep), bytes(out)


def prio3_shard_batched(p, meas, nonce, rand, client_id):
    """-> (leader report, the Helpers' reports back to back: num_aggregators - 1 of helper_report_len bytes)"""
    s = sizes(p)
    assert len(rand) == 16 * (5 * p.num_aggregators - 1)
    lr = _buf(s.leader_report_len); hr = _buf(s.helper_report_len * (p.num_aggregators - 1))
    rc = _lib.wo_prio3_shard_batched(C.byref(p), _meas(meas), _b(nonce), _b(rand), _b(int(client_id).to_bytes(16, "big")), lr, hr)
    assert rc == 0, rc
    return bytes(lr), bytes(hr)


def prio3_prepare_encoded_key(p, verify_key, report):
    s = sizes(p)
    tag = _buf(16); tid = _buf(16); out = _buf(16 * s.output_len)
    rc = _lib.wo_prio3_prepare_encoded_key(C.byref(p), _b(verify_key), _b(report), C.c_size_t(len(report)), tag, tid, out)
    return rc, bytes(tag), bytes(tid), bytes(out)


def prio3_prepare_batch(p, verify_key, blob, offsets, nthreads=1):
    """blob: bytes/np.uint8 array; offsets: n+1 uint64.  Returns (tags[n,16], ids[n,16], status[n], out_shares[n,output_len*16])."""
    s = sizes(p)
    blob = np.ascontiguousarray(np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    tags = np.zeros((n, 16), np.uint8); ids = np.zeros((n, 16), np.uint8)
    status = np.zeros(n, np.int32); outs = np.zeros((n, 16 * s.output_len), np.uint8)
    _lib.wo_prio3_prepare_batch.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_uint] + [C.c_void_p] * 4
    rc = _lib.wo_prio3_prepare_batch(C.cast(C.byref(p), C.c_void_p), C.cast(_b(verify_key), C.c_void_p), blob.ctypes.data,
                                     offsets.ctypes.data, n, nthreads, tags.ctypes.data, ids.ctypes.data,
                                     status.ctypes.data, outs.ctypes.data)
    assert rc == 0, rc
    return tags, ids, status, outs


def prio3_aggregate(out_shares, output_len, keep=None):
    out_shares = np.ascontiguousarray(out_shares, dtype=np.uint8)
    n = out_shares.shape[0]
    agg = np.zeros(16 * output_len, np.uint8)
    keep_p = None
    if keep is not None:
        keep = np.ascontiguousarray(keep, dtype=np.uint8); keep_p = keep.ctypes.data
    _lib.wo_prio3_aggregate.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
    _lib.wo_prio3_aggregate(out_shares.ctypes.data, n, output_len, keep_p, agg.ctypes.data)
    return agg


def aes128_encrypt_block(key, block):
    o = _buf(16); _lib.wo_aes128_encrypt_block(_b(key), _b(block), o); return bytes(o)


def hh_bucket_sum_u32(buckets, keep=None):
    buckets = np.ascontiguousarray(buckets, dtype=np.uint32)
    n, words = buckets.shape[0], int(np.prod(buckets.shape[1:]))
    acc = np.zeros(words, np.uint32)
    keep_p = None
    if keep is not None:
        keep = np.ascontiguousarray(keep, dtype=np.uint8); keep_p = keep.ctypes.data
    _lib.wo_hh_bucket_sum_u32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    _lib.wo_hh_bucket_sum_u32(buckets.ctypes.data, keep_p, n, words, acc.ctypes.data)
    return acc.reshape(buckets.shape[1:])


# ---- heavy-hitter key chains (oracle/vdpf.c) ----
ERR_KEY_FORMAT = -9
BUCKET_WORDS = 258


def sha256(data):
    o = _buf(32); _lib.wo_sha256(_b(data) if data else None, C.c_size_t(len(data)), o); return bytes(o)


def mmo_fixed(key_byte, block):
    o = _buf(16); _lib.wo_mmo_fixed(C.c_int(key_byte), _b(block), o); return bytes(o)


def aes_backend():
    return int(_lib.wo_aes_backend())


_lib.wo_hh_key_chain_len.restype = C.c_uint64


def hh_key_chain_len(depth):
    return int(_lib.wo_hh_key_chain_len(C.c_uint32(depth)))


def hh_gen_key_chain(depth, client_id, alpha, bucket, rng_seed):
    """bucket: 258 u32 (256 data, sign_data, ctr).  Returns the two servers' bincode key chains."""
    n = hh_key_chain_len(depth)
    k0, k1 = _buf(n), _buf(n)
    bk = np.ascontiguousarray(bucket, dtype=np.uint32)
    assert bk.shape == (BUCKET_WORDS,)
    _lib.wo_hh_gen_key_chain.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = _lib.wo_hh_gen_key_chain(depth, C.cast(_b(int(client_id).to_bytes(16, "little")), C.c_void_p), alpha,
                                  bk.ctypes.data, C.cast(_b(rng_seed), C.c_void_p), C.cast(k0, C.c_void_p), C.cast(k1, C.c_void_p))
    if rc:
        raise ValueError(f"gen_key_chain rc={rc}")
    return bytes(k0), bytes(k1)


def hh_eval_and_check(depth, key_chain, want_dat=True):
    dat = np.zeros(((1 << depth), BUCKET_WORDS), np.uint32) if want_dat else None
    tag = _buf(64)
    _lib.wo_hh_eval_and_check.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    rc = _lib.wo_hh_eval_and_check(depth, C.cast(_b(key_chain), C.c_void_p), len(key_chain),
                                   dat.ctypes.data if want_dat else None, C.cast(tag, C.c_void_p))
    return rc, dat, bytes(tag)


def hh_eval_and_check_batch(depth, verify_key, keys, nthreads=1, want_dat=True):
    """keys: np.uint8 [n, stride].  Returns (dat[n, 2^depth, 258] or None, tags[n, 64], ids[n, 16], status[n])."""
    keys = np.ascontiguousarray(keys, dtype=np.uint8)
    n, stride = keys.shape
    dat = np.zeros((n, 1 << depth, BUCKET_WORDS), np.uint32) if want_dat else None
    tags = np.zeros((n, 64), np.uint8); ids = np.zeros((n, 16), np.uint8); status = np.zeros(n, np.int32)
    _lib.wo_hh_eval_and_check_batch.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = _lib.wo_hh_eval_and_check_batch(depth, C.cast(_b(verify_key), C.c_void_p), keys.ctypes.data, stride, n, nthreads,
                                         dat.ctypes.data if want_dat else None, tags.ctypes.data, ids.ctypes.data,
                                         status.ctypes.data)
    assert rc == 0, rc
    return dat, tags, ids, status


# ---- Prio3Count over Field64 (oracle/count64.c) ----
P64 = 2**64 - 2**32 + 1


def count_shard(num_aggregators, meas, nonce, rand):
    ls = _buf(48); hs = _buf(32 * (num_aggregators - 1))
    rc = _lib.wo_count_shard(C.c_uint32(num_aggregators), C.c_uint64(meas), _b(nonce), _b(rand), ls, hs); assert rc == 0, rc
    hs = bytes(hs)
    return [bytes(ls)] + [hs[32 * i:32 * i + 32] for i in range(num_aggregators - 1)]


def count_prepare_init(num_aggregators, verify_key, agg_id, nonce, input_share):
    prep = _buf(32); out = _buf(8)
    rc = _lib.wo_count_prepare_init(C.c_uint32(num_aggregators), _b(verify_key), C.c_uint32(agg_id), _b(nonce), _b(input_share),
                                    C.c_size_t(len(input_share)), prep, out)
    return rc, bytes(prep), bytes(out)


def count_prepare_init_encoded(num_aggregators, verify_key, key):
    prep = _buf(32); out = _buf(8)
    rc = _lib.wo_count_prepare_init_encoded(C.c_uint32(num_aggregators), _b(verify_key), _b(key) if len(key) else _buf(1),
                                            C.c_size_t(len(key)), prep, out)
    return rc, bytes(prep), bytes(out)


def count_prepare_finish(num_aggregators, prep_shares):
    """0 accept, 1 reject, < 0 decode error; prep_shares: one 32-byte share per aggregator"""
    return _lib.wo_count_prepare_finish(C.c_uint32(num_aggregators), _b(b"".join(prep_shares)))


def count_aggregate(out_shares, keep=None):
    n = len(out_shares)
    agg = _buf(8)
    k = None if keep is None else _b(bytes(bytearray(int(x) for x in keep)))
    _lib.wo_count_aggregate.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    _lib.wo_count_aggregate(_b(b"".join(out_shares)) if n else None, n, k, agg)
    return bytes(agg)


def f64_expand(seed, usage, binder, n):
    out = (C.c_uint64 * n)()
    _lib.wo_f64_expand.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    _lib.wo_f64_expand(_b(seed), usage, _b(binder) if binder else None, len(binder), out, n)
    return list(out)
