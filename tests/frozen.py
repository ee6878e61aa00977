"""Loader of the frozen known-answer fixtures (tests/golden/batched, tests/golden/hh; made by tests/golden/make_frozen_fixtures.py
from oracle/).  They do not pin parity with the reference binary -- nothing in this image can -- they freeze today's agreed
answers so that oracle and product cannot drift together unnoticed."""
import ctypes as C
import glob
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def cases(kind):
    return sorted(glob.glob(os.path.join(HERE, "golden", kind, "*.json")))


def load(path):
    with open(path) as f:
        return json.load(f)


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def product_reports(fx):
    """Both encoded reports of every client from the PRODUCT's CPU client (wh_client_shard_batched)."""
    import whisper_amd as W
    c = fx["circuit"]
    p = W.Prio3Params(c["circuit"], c["bits"], c["len"], c["chunk"])
    out = []
    for cl in fx["clients"]:
        m = np.array(cl["measurement"], np.uint64)
        nonce, rand = np.frombuffer(bytes.fromhex(cl["nonce"]), np.uint8), np.frombuffer(bytes.fromhex(cl["rand"]), np.uint8)
        cid = np.frombuffer(int(cl["client_id"]).to_bytes(16, "big"), np.uint8)
        lr, hr = np.zeros(p.leader_report_len, np.uint8), np.zeros(p.helper_report_len, np.uint8)
        rc = W.lib.wh_client_shard_batched(C.byref(p.c), m.ctypes.data, nonce.ctypes.data, rand.ctypes.data, cid.ctypes.data, lr.ctypes.data, hr.ctypes.data)
        assert rc == 0
        out.append((lr.tobytes(), hr.tobytes()))
    return p, out


def server_views(fx, reports):
    alice, bob = [], []
    for i, (lr, hr) in enumerate(reports):
        a, b = (lr, hr) if i % 2 == 0 else (hr, lr)
        if i == fx["tampered_on_bob"]:
            b = bytearray(b); b[-16] = (b[-16] + 1) & 0xff; b = bytes(b)
        alice.append(a); bob.append(b)
    return {"alice": alice, "bob": bob}


def product_key_chains(fx):
    from whisper_amd import hh
    out = []
    for kc in fx["key_chains"]:
        a, b = hh.gen_key_chain(fx["depth"], int(kc["client_id"]), kc["alpha"], np.array(kc["bucket"], np.uint32), bytes.fromhex(kc["rng_seed"]))
        out.append((np.asarray(a, np.uint8), np.asarray(b, np.uint8)))
    return out
