"""A second, independent restatement of the batched verifier, in pure Python (hashlib's SHAKE128 and Python integers).

TEST INFRASTRUCTURE ONLY.  oracle/ (C) and the GPU product were first written by one reader of the reference; this file was
written later, from the Rust sources alone, without looking at oracle/*.c, so that a misreading shared by the two shows up as a
difference here (tests/test_py_restatement.py).  It keeps the reference's shape: per-wire inverse DFT + Horner, both runs, one
report at a time.  Like the oracle it cannot be pinned against the Rust binary for the batched-only glue ("parity unpinned",
DESIGN.md section 7); what it adds is independence.  All paths are relative to /root/reference.

  server-batch-prio3/src/main.rs:33-60            prepare_encoded_key
  batched-flpcp-rs/src/vdaf.rs:148-192, 339-346   VdafBatchedKey encoding, domain_separation_tag
  batched-flpcp-rs/src/vdaf/prio3.rs:69-78, 482-493, 1549-1875, 2466-2879
  batched-flpcp-rs/src/vdaf/xof.rs:98-139, 197-218; prng.rs:56-99; field.rs:256-269, 405-423
  batched-flpcp-rs/src/flp.rs:359-512, 658-724; flp/types.rs:242-313, 383-470, 577-676, 686-698, 742-777
  batched-flpcp-rs/src/fft.rs:31-124; polynomial.rs:171-183, 223-231; fp.rs:391-427
  common/src/prg.rs:639-655                       Prf (AES-128, one block; FIPS-197 restated below)
"""
import hashlib

P = 340282366920938462946865773367900766209           # fp.rs:392
_R = 1 << 128
_G = 107630958476043550189608038630704257141 * pow(_R, -1, P) % P     # fp.rs:395, stored in Montgomery form
SUMVEC, HISTOGRAM, AVERAGE = 0, 1, 2
VERSION = 7                                           # vdaf.rs:28
(DST_MEASUREMENT_SHARE, DST_PROOF_SHARE, DST_JOINT_RANDOMNESS, DST_PROVE_RANDOMNESS, DST_QUERY_RANDOMNESS, DST_JOINT_RAND_SEED,
 DST_JOINT_RAND_PART, DST_QUERY_RAND_PART, DST_RLC_RANDOMNESS, DST_HASH_PART) = range(1, 11)          # prio3.rs:69-78


class Malformed(Exception):
    """What is an unwrap() abort in the reference (main.rs:46-49): a report that does not decode or cannot be queried."""


def root(l):
    """field.rs:711-717 reads roots[l] (fp.rs:398-420) = g^(2^(num_roots - l))."""
    return pow(_G, 1 << (66 - l), P)


def next_pow2(x):
    r = 1
    while r < x:
        r <<= 1
    return r


def log2(x):                                           # fp.rs:424-427
    y = x.bit_length() - 1
    return y + (1 if x > (1 << y) else 0)


def bitrev(d, x):                                      # fft.rs:118-124
    y = 0
    for i in range(d):
        y += ((x >> i) & 1) << (d - i)
    return y >> 1


def dft(inp, size):                                    # fft.rs:31-85
    d = log2(size)
    assert size == 1 << d
    out = [0] * size
    for i in range(size):
        j = bitrev(d, i)
        out[i] = inp[j] if j < len(inp) else 0
    for l in range(1, d + 1):
        w = 1
        r = root(l)
        y = 1 << (l - 1)
        chunk = (size // y) >> 1
        for i in range(y):
            if i:
                w = w * r % P
            for j in range(chunk):
                x = (j << l) + i
                u, v = out[x], w * out[x + y] % P
                out[x], out[x + y] = (u + v) % P, (u - v) % P
    return out


def dft_inv_finish(out, size, size_inv):               # fft.rs:102-115
    out[0] = out[0] * size_inv % P
    out[size >> 1] = out[size >> 1] * size_inv % P
    for i in range(1, size >> 1):
        tmp = out[i] * size_inv % P
        out[i] = out[size - i] * size_inv % P
        out[size - i] = tmp


def poly_eval(poly, at):                               # polynomial.rs:171-183
    if not poly:
        return 0
    r = poly[-1]
    for c in reversed(poly[:-1]):
        r = (r * at + c) % P
    return r


# ------------------------------------------------------------------------------------------ XOF ----
class Xof:
    """XofShake128 (xof.rs:197-218): SHAKE128(len(dst) . dst . seed . updates)."""

    def __init__(self, seed, dst):
        assert len(seed) == 16 and len(dst) < 256
        self.h = hashlib.shake_128(bytes([len(dst)]) + bytes(dst) + bytes(seed))

    def update(self, data):
        self.h.update(bytes(data))

    def into_seed(self):                               # xof.rs:121-126
        return self.h.digest(16)

    def into_field_vec(self, n):
        """Prng over the seed stream (prng.rs:56-99): consecutive 16-byte chunks, little endian, those >= p skipped
        (field.rs:405-423 with the all-ones mask)."""
        want = 16 * n
        while True:
            buf = self.h.digest(want)
            vals = [v for v in (int.from_bytes(buf[k:k + 16], "little") for k in range(0, len(buf), 16)) if v < P]
            if len(vals) >= n:
                return vals[:n]
            want += 16 * (n - len(vals) + 1)


def seed_stream(seed, dst, binder):                    # xof.rs:131-138
    x = Xof(seed, dst)
    x.update(binder)
    return x


def enc(v):                                            # field.rs:598-607: 16 bytes, little endian
    return int(v).to_bytes(16, "little")


# -------------------------------------------------------------------------------------- circuits ----
class Circuit:
    def __init__(self, kind, bits=0, length=0, chunk=0):
        self.kind, self.bits, self.length, self.chunk = kind, bits, length, chunk
        if kind == SUMVEC:                             # types.rs:577, 650-676
            self.id = 2
            self.input_len = bits * length
            self.calls = -(-self.input_len // chunk)
            self.arity, self.output_len, self.joint_rand_len = 2 * chunk, length, 1
        elif kind == HISTOGRAM:                        # types.rs:383, 444-470
            self.id = 3
            self.input_len = length
            self.calls = -(-length // chunk)
            self.arity, self.output_len, self.joint_rand_len = 2 * chunk, length, 2
        else:                                          # Average, types.rs:242, 286-312
            self.id = 0xFFFF0000
            self.input_len = self.calls = bits
            self.arity, self.output_len, self.joint_rand_len = 1, 1, 1
        self.degree = 2                                # Mul and x^2 - x
        self.m = next_pow2(1 + self.calls)
        self.proof_len = self.arity + self.degree * (self.m - 1) + 1
        self.verifier_len = 2 + self.arity
        self.query_rand_len = 1

    def dst(self, usage):                              # vdaf.rs:339-346
        return bytes([VERSION, 0]) + self.id.to_bytes(4, "big") + usage.to_bytes(2, "big")

    def gadget_call(self, inp):
        if self.kind == AVERAGE:                       # PolyEval of poly_range_check(0, 2) = x^2 - x (polynomial.rs:223-231)
            return poly_eval([0, P - 1, 1], inp[0])
        return sum(inp[2 * k] * inp[2 * k + 1] for k in range(len(inp) // 2)) % P     # ParallelSum<Mul>, gadgets.rs:231-293

    def valid(self, call, inp, joint_rand, num_shares):
        """Type::valid with `call` standing for the (shim) gadget."""
        if self.kind == AVERAGE:                       # types.rs:268-277, 686-698
            acc, r = 0, joint_rand[0]
            for x in inp:
                acc = (acc + r * call([x])) % P
                r = r * joint_rand[0] % P
            return acc
        rc = self._range_checks(call, inp, joint_rand[0], num_shares)
        if self.kind == SUMVEC:                        # types.rs:623-639
            return rc
        sum_check = (-pow(num_shares, -1, P)) % P      # types.rs:410-437
        for x in inp:
            sum_check = (sum_check + x) % P
        return (joint_rand[1] * rc + joint_rand[1] * joint_rand[1] % P * sum_check) % P

    def _range_checks(self, call, inp, jr, num_shares):     # types.rs:742-777
        inv = pow(num_shares, -1, P)
        out, r_power = 0, jr
        for c0 in range(0, len(inp), self.chunk):
            piece = inp[c0:c0 + self.chunk]
            padded = []
            for x in piece:
                padded += [r_power * x % P, (x - inv) % P]
                r_power = r_power * jr % P
            for _ in range(self.chunk - len(piece)):
                padded += [0, (-inv) % P]
            out = (out + call(padded)) % P
        return out

    def truncate(self, inp):
        if self.kind == HISTOGRAM:                     # types.rs:439-442
            return list(inp)
        width = self.bits                              # types.rs:641-648, 279-283; field.rs:256-269
        return [sum((1 << l) * b for l, b in enumerate(inp[k:k + width])) % P for k in range(0, len(inp), width)]

    def query(self, inp, proof, query_rand, joint_rand, num_shares):      # flp.rs:359-479
        assert len(inp) == self.input_len and len(proof) == self.proof_len
        assert len(query_rand) == 1 and len(joint_rand) == self.joint_rand_len
        m, r = self.m, query_rand[0]
        if pow(r, m, P) == 1:
            raise Malformed("query randomness is an m-th root of unity")
        # QueryShimGadget::new (flp.rs:678-712)
        f_vals = [[proof[w]] + [0] * self.calls for w in range(self.arity)]
        gpoly = proof[self.arity:]
        p = m * self.degree
        size = next_pow2(p)
        p_vals = dft(gpoly, size)
        step = 1 << (log2(p) - log2(m))
        p_at_r = poly_eval(gpoly, r)
        ct = [1]

        def shim_call(args):                           # flp.rs:716-724
            for w, a in enumerate(args):
                f_vals[w][ct[0]] = a
            outp = p_vals[ct[0] * step]
            ct[0] += 1
            return outp

        verifier = [self.valid(shim_call, inp, joint_rand, num_shares)]
        m_inv = pow(m, -1, P)
        for w in range(self.arity):
            f = dft(f_vals[w], m)
            dft_inv_finish(f, m, m_inv)
            verifier.append(poly_eval(f, r))
        verifier.append(p_at_r)
        assert len(verifier) == self.verifier_len
        return verifier

    def decide(self, verifier):                        # flp.rs:483-512
        assert len(verifier) == self.verifier_len
        if verifier[0] != 0:
            return False
        return self.gadget_call(verifier[1:1 + self.arity]) == verifier[1 + self.arity]


# ---------------------------------------------------------------------------------------- decode ----
class Cursor:
    def __init__(self, data):
        self.d, self.pos = bytes(data), 0

    def take(self, n):
        if self.pos + n > len(self.d):
            raise Malformed("input ends early")
        out = self.d[self.pos:self.pos + n]
        self.pos += n
        return out

    def elem(self):
        v = int.from_bytes(self.take(16), "little")
        if v >= P:
            raise Malformed("element is not below the modulus")
        return v


def decode_key(c, num_aggregators, data):
    """VdafBatchedKey::decode_with_param (vdaf.rs:164-192) with the Prio3 decoders (prio3.rs:1563-1875)."""
    cur = Cursor(data)
    k = {"client_id": int.from_bytes(cur.take(16), "big")}                      # codec.rs:259-262
    k["public_share"] = [cur.take(16) for _ in range(num_aggregators)] if c.joint_rand_len else None
    k["agg_id"] = cur.take(1)[0]
    if k["agg_id"] >= num_aggregators:                                          # prio3.rs:1496-1501
        raise Malformed("unexpected aggregator id")
    leader = k["agg_id"] == 0
    if leader:
        k["meas"] = [cur.elem() for _ in range(c.input_len)]
        k["proof1"] = [cur.elem() for _ in range(c.proof_len)]
    else:
        k["meas_seed"], k["proof1_seed"] = cur.take(16), cur.take(16)
    k["jr_blind"] = cur.take(16) if c.joint_rand_len else None
    if leader:
        k["proof2"] = [cur.elem() for _ in range(c.proof_len)]
    else:
        k["proof2_seed"] = cur.take(16)
    k["public_share_second"] = [cur.take(16) for _ in range(num_aggregators)] if c.joint_rand_len else None
    nq = int.from_bytes(cur.take(8), "big")                                     # codec.rs:243-246
    if 16 * nq > len(cur.d):
        raise Malformed("input ends early")
    k["pp1"] = [cur.elem() for _ in range(nq)]
    k["pp2"] = [cur.elem() for _ in range(nq)]
    k["qr_blind"] = cur.take(16)                                                # query_rand_len > 0 for every circuit
    k["nonce"] = cur.take(16)
    return k


def derive_seed(c, parts):                             # prio3.rs:482-493
    x = Xof(bytes(16), c.dst(DST_JOINT_RAND_SEED))
    for part in parts:
        x.update(part)
    return x.into_seed()


def prepare_batched(c, num_aggregators, verify_key, k):
    """Prio3::prepare_batched (prio3.rs:2466-2879): (proof tag, out share)."""
    agg_id = k["agg_id"]
    leader = agg_id == 0
    ab = bytes([agg_id])
    if leader:
        meas, proof1, proof2 = k["meas"], k["proof1"], k["proof2"]
    else:                                              # :2480-2536
        meas = seed_stream(k["meas_seed"], c.dst(DST_MEASUREMENT_SHARE), ab).into_field_vec(c.input_len)
        proof1 = seed_stream(k["proof1_seed"], c.dst(DST_PROOF_SHARE), ab).into_field_vec(c.proof_len)
        proof2 = seed_stream(k["proof2_seed"], c.dst(DST_PROOF_SHARE), ab).into_field_vec(c.proof_len)
    out_share = c.truncate(meas)                       # :2496
    # joint randomness (:2541-2602)
    x = Xof(k["jr_blind"], c.dst(DST_JOINT_RAND_PART))
    x.update(ab)
    x.update(k["nonce"])
    for e in meas:
        x.update(enc(e))
    own_jr_part = x.into_seed()
    parts = list(k["public_share"])
    parts[agg_id] = own_jr_part
    jr_seed = derive_seed(c, parts)
    jr_fused = seed_stream(jr_seed, c.dst(DST_JOINT_RANDOMNESS), b"").into_field_vec(2 * c.joint_rand_len)
    jr_split = [jr_fused[:c.joint_rand_len], jr_fused[c.joint_rand_len:]]
    # query randomness (:2604-2683)
    x = Xof(k["qr_blind"], c.dst(DST_QUERY_RAND_PART))
    for e in jr_fused:
        x.update(enc(e))
    for e in proof1:
        x.update(enc(e))
    for e in proof2:
        x.update(enc(e))
    own_qr_part = x.into_seed()
    parts2 = list(k["public_share_second"])
    parts2[agg_id] = own_qr_part
    qr_seed = derive_seed(c, parts2)                   # the same DST_JOINT_RAND_SEED
    qr_fused = seed_stream(qr_seed, c.dst(DST_QUERY_RANDOMNESS), b"").into_field_vec(2 * c.query_rand_len)
    qr_split = [qr_fused[:1], qr_fused[1:]]
    # both runs (:2685-2701)
    ver1 = c.query(meas, proof1, qr_split[0], jr_split[0], num_aggregators)
    ver2 = c.query(meas, proof2, qr_split[1], jr_split[1], num_aggregators)
    check_count = len(ver1)                            # run 1's differences are computed and dropped (:2711-2724)
    if len(k["pp1"]) != len(ver1) or len(k["pp2"]) != len(ver2):
        raise Malformed("public proof length")
    check_count += len(ver2)
    diff = [(y - xq) % P for xq, y in zip(k["pp2"], ver2)] if leader else list(ver2)     # :2728-2739
    check_count += 2
    decide_zero = [0 if c.decide(k["pp1"]) else 1, 0 if c.decide(k["pp2"]) else 1]       # :2762-2786
    check_count += 2
    rand_diffs = [1 if k["public_share"][agg_id] != own_jr_part else 0,                  # :2790-2805
                  1 if k["public_share_second"][agg_id] != own_qr_part else 0]
    # hash of the public parts (:2808-2840)
    h = Xof(bytes(16), c.dst(DST_HASH_PART))
    for part in k["public_share"]:
        h.update(part)
    for part in k["public_share_second"]:
        h.update(part)
    for e in k["pp1"]:
        h.update(enc(e))
    for e in k["pp2"]:
        h.update(enc(e))
    hash0 = seed_stream(h.into_seed(), c.dst(DST_HASH_PART), b"").into_field_vec(1)[0]
    check_count += num_aggregators - 1
    if leader:                                         # :2844-2855
        hash_zero = [(-hash0) % P] * (num_aggregators - 1)
    else:
        hash_zero = [0] * (num_aggregators - 1)
        hash_zero[agg_id - 1] = hash0
    r = Xof(verify_key, c.dst(DST_RLC_RANDOMNESS))      # :2859-2862
    r.update(k["nonce"])
    rlc = r.into_field_vec(check_count)
    tag = 0
    for xv, y in zip(diff + decide_zero + rand_diffs + hash_zero, rlc):                  # zip stops at the shorter (:2865-2871)
        tag = (tag + xv * y) % P
    return tag, out_share


def prepare_encoded_key(c, num_aggregators, verify_key, data):
    """server-batch-prio3/src/main.rs:33-60: (testing id, tag as this server reports it, out share)."""
    k = decode_key(c, num_aggregators, data)
    tag, out_share = prepare_batched(c, num_aggregators, verify_key, k)
    testing_id = int.from_bytes(aes128_encrypt_block(verify_key, k["client_id"].to_bytes(16, "little")), "little")
    if k["agg_id"] != 0:
        tag = (-tag) % P
    return testing_id, tag, out_share


# ---------------------------------------------------------------------------- AES-128 (FIPS-197) ----
def _xt(b):
    b <<= 1
    return (b ^ 0x11b) & 0xff if b & 0x100 else b


def _sbox():
    # multiplicative inverse in GF(2^8) followed by the affine map
    inv = [0] * 256
    for a in range(1, 256):
        for b in range(1, 256):
            x, y, prod = a, b, 0
            while y:
                if y & 1:
                    prod ^= x
                x = _xt(x)
                y >>= 1
            if prod == 1:
                inv[a] = b
                break
    box = []
    for a in range(256):
        v = inv[a]
        r = v
        for s in range(1, 5):
            r ^= ((v << s) | (v >> (8 - s))) & 0xff
        box.append(r ^ 0x63)
    return box


_SBOX = None


def aes128_encrypt_block(key, block):
    global _SBOX
    if _SBOX is None:
        _SBOX = _sbox()
    S = _SBOX
    w = [list(key[4 * i:4 * i + 4]) for i in range(4)]
    rcon = 1
    for i in range(4, 44):
        t = list(w[i - 1])
        if i % 4 == 0:
            t = [S[t[1]] ^ rcon, S[t[2]], S[t[3]], S[t[0]]]
            rcon = _xt(rcon)
        w.append([a ^ b for a, b in zip(w[i - 4], t)])
    rk = [sum(w[4 * r:4 * r + 4], []) for r in range(11)]
    st = [b ^ k for b, k in zip(block, rk[0])]          # column-major: byte 4 c + r
    for rnd in range(1, 11):
        st = [S[b] for b in st]
        st = [st[4 * ((c + r) & 3) + r] for c in range(4) for r in range(4)]          # ShiftRows
        if rnd < 10:
            out = []
            for c in range(4):
                a = st[4 * c:4 * c + 4]
                out += [_xt(a[0]) ^ _xt(a[1]) ^ a[1] ^ a[2] ^ a[3], a[0] ^ _xt(a[1]) ^ _xt(a[2]) ^ a[2] ^ a[3],
                        a[0] ^ a[1] ^ _xt(a[2]) ^ _xt(a[3]) ^ a[3], _xt(a[0]) ^ a[0] ^ a[1] ^ a[2] ^ _xt(a[3])]
            st = out
        st = [b ^ k for b, k in zip(st, rk[rnd])]
    return bytes(st)
