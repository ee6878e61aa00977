"""Prio3Count (Prio3<Count<Field64>, XofShake128, 16>, batched-flpcp-rs/src/vdaf/prio3.rs:81, 390-401) through the C ABI:
the thin host-side mirror the tests drive.  Elements are 8-byte little-endian Field64 residues."""
import ctypes as C

import numpy as np

from ._lib import lib, WhError

P64 = 2**64 - 2**32 + 1


class Prio3Count:
    """One aggregator of a server built with Prio3::new_count(num_aggregators)."""

    def __init__(self, num_aggregators=2, device=0):
        self.num_aggregators = num_aggregators
        self._ctx = C.c_void_p()
        rc = lib.wh_hh_ctx_create(device, C.byref(self._ctx))
        if rc:
            raise WhError(rc, "wh_hh_ctx_create")
        self.n_last = 0

    def close(self):
        if getattr(self, "_ctx", None):
            lib.wh_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc:
            raise WhError(rc, f"{what}: {lib.wh_last_error(self._ctx).decode()}")

    @staticmethod
    def encode_vdaf_key(agg_id, input_share, nonce):
        """VdafKey::get_encoded (vdaf.rs:209-217) with Count's empty public share"""
        return bytes([agg_id]) + bytes(input_share) + bytes(nonce)

    def prepare_init_encoded_keys(self, verify_key, keys):
        """first map of run_vdaf_prepare (server-base-prio3/src/main.rs:41-61): -> (prep_shares [n,32] u8, status [n] i32)"""
        n = len(keys)
        offsets = np.zeros(n + 1, np.uint64)
        if n:
            offsets[1:] = np.cumsum([len(k) for k in keys])
        blob = np.frombuffer(b"".join(keys) + b"\0", np.uint8)
        prep = np.zeros((n, 32), np.uint8)
        status = np.zeros(n, np.int32)
        vk = np.frombuffer(bytes(verify_key), np.uint8)
        self._chk(lib.wh_prio3_count_prepare_init_batch(self._ctx, self.num_aggregators, vk.ctypes.data, blob.ctypes.data, offsets.ctypes.data, n,
                                                        prep.ctypes.data, status.ctypes.data), "count_prepare_init")
        self.n_last = n
        return prep, status

    def prepare_finish(self, prep_shares_by_aggregator):
        """second map (main.rs:68-100): sums the shares, decides; -> status [n] i32"""
        arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in prep_shares_by_aggregator]
        n = self.n_last
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        status = np.zeros(n, np.int32)
        self._chk(lib.wh_prio3_count_prepare_finish_batch(self._ctx, len(arrs), ptrs, n, status.ctypes.data), "count_prepare_finish")
        return status

    def out_shares(self):
        out = np.zeros((self.n_last, 8), np.uint8)
        self._chk(lib.wh_prio3_count_out_shares(self._ctx, out.ctypes.data), "count_out_shares")
        return out

    def aggregate(self, keep=None):
        agg = np.zeros(8, np.uint8)
        k = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
        self._chk(lib.wh_prio3_count_aggregate(self._ctx, None if k is None else k.ctypes.data, agg.ctypes.data), "count_aggregate")
        return agg

    def set_test_sampler_bound(self, bound=None):
        self._chk(lib.wh_prio3_count_set_test_sampler_bound(self._ctx, int(bound or 0)), "count_set_test_sampler_bound")

    def test_f64_xof_expand(self, seeds, usage, binder, n_elems, bound=P64):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint8)
        n = seeds.shape[0]
        b = np.frombuffer(bytes(binder) + b"\0", np.uint8)
        out = np.zeros((n, 5), np.uint64)
        self._chk(lib.wh_test_f64_xof_expand(self._ctx, seeds.ctypes.data, n, usage, b.ctypes.data, len(binder), n_elems, int(bound), out.ctypes.data),
                  "test_f64_xof_expand")
        return out[:, :n_elems]


def new_count(num_aggregators=2, device=0):
    return Prio3Count(num_aggregators, device)
