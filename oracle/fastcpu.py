"""Loader of oracle/librs_oracle_fast.so -- the TIMED leg of bench.py's cpu_baseline (TEST INFRASTRUCTURE ONLY).

The same CPU restatement as oracle/oracle.py compiled with the arithmetic Microsoft SEAL publishes for this path
(Harvey lazy NTT with Shoup quotients and Barrett dyadic products: rs_fastcpu.c; Barrett instead of a 128-bit `%` in
every other modular product: rs_oracle.c -DRSO_FAST_MULMOD).  tests/test_oracle.py asserts that its results equal the
`%`-based checker's bit for bit.  One use as a checker since round 6: the COMPLETE inner-product check of a headline proof
(all 96 slabs, tests/proof_check.py groth16_check(all_slabs=True)) runs inner_product_limb here -- the `%` build would take
eight minutes on 128 threads -- with slabs of the same proof recomputed by the `%` checker beside it."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import oracle as O

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librs_oracle_fast.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(_HERE, f) for f in ("rs_oracle.c", "rs_fastcpu.c", "rs_oracle.h")]
        if not os.path.exists(_SO) or any(os.path.getmtime(_SO) < os.path.getmtime(s) for s in srcs):
            subprocess.check_call(["make", "-C", _HERE, "librs_oracle_fast.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(_SO)
        L.rso_ctx_create.restype = C.c_void_p
        L.rso_ctx_create.argtypes = [C.c_int, C.c_int, O.u64p, C.c_int, C.c_int, O.u64p]
        L.rso_ctx_destroy.argtypes = [C.c_void_p]
        L.rsf_ctx_create.restype = C.c_void_p
        L.rsf_ctx_create.argtypes = [C.c_void_p]
        L.rsf_ctx_destroy.argtypes = [C.c_void_p]
        L.rsf_ntt_fwd.argtypes = [C.c_void_p, C.c_int, C.c_int, O.u64p]
        L.rsf_ntt_inv.argtypes = [C.c_void_p, C.c_int, C.c_int, O.u64p]
        L.rsf_inner_product_mt.restype = C.c_size_t
        L.rsf_inner_product_mt.argtypes = [C.c_void_p, O.u64p, C.c_size_t, O.u64p, O.u8p, C.c_size_t, O.u64p, C.c_int]
        L.rsf_inner_product_limb.restype = None
        L.rsf_inner_product_limb.argtypes = [C.c_void_p, C.c_int, O.u64p, C.c_size_t, C.c_size_t, C.c_size_t, O.u64p, C.c_size_t, O.u64p, C.c_int]
        L.rso_witness_map.argtypes = [C.c_uint64, C.c_size_t, C.POINTER(O.R1CS), C.c_int] + [O.u64p] * 12
        L.rso_witness_map_mt.argtypes = [C.c_uint64, C.c_size_t, C.POINTER(O.R1CS), C.c_int] + [O.u64p] * 12 + [C.c_int]
        L.rso_max_threads.restype = C.c_int
        _lib = L
    return _lib


class FastCtx:
    """SEAL-arithmetic twin of oracle.Ctx for the inner product and the standalone transforms."""

    def __init__(self, N, q, N_enc, Q):
        self.N, self.L, self.N_enc, self.K = N, len(q), N_enc, len(Q)
        qa, Qa = (C.c_uint64 * self.L)(*q), (C.c_uint64 * self.K)(*Q)
        self.base = lib().rso_ctx_create(N, self.L, qa, N_enc, self.K, Qa)
        assert self.base, "bad parameters"
        self.h = lib().rsf_ctx_create(self.base)

    def enc_shape(self, *lead):
        return tuple(lead) + (self.L, 2, self.K, self.N_enc)

    def ntt(self, modset, index, a, inverse=False):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        (lib().rsf_ntt_inv if inverse else lib().rsf_ntt_fwd)(self.h, modset, index, O.p64(a))
        return a

    def inner_product(self, encs, rings, kinds=None, threads=1, window=0):
        """EncodingElem::inner_product; same contract as oracle.Ctx.inner_product."""
        encs, rings = np.ascontiguousarray(encs), np.ascontiguousarray(rings)
        T = rings.shape[0]
        assert encs.shape[0] == (window or T)
        out = np.zeros(self.enc_shape(), dtype=np.uint64)
        kp = None
        if kinds is not None:
            kinds = np.ascontiguousarray(kinds, dtype=np.uint8)
            kp = kinds.ctypes.data_as(O.u8p)
        used = lib().rsf_inner_product_mt(self.h, O.p64(encs), window, O.p64(rings), kp, T, O.p64(out), threads)
        return out, int(used)

    def inner_product_limb(self, limb, key, rows, acc, t0=0, window=None, threads=0):
        """acc [2][K][N_enc] += ring limb `limb` of <key, rows>: key [W][2][K][N_enc] (the limb's slice of a key vector, W stored
        elements; term t0 + t reads element (t0 + t) mod window), rows [T][N].  All 2 K slabs of the limb in one pass, the
        plaintext of a term transformed once (rs_fastcpu.c rsf_inner_product_limb)."""
        assert key.dtype == np.uint64 and key.flags.c_contiguous and key.shape[1:] == (2, self.K, self.N_enc)
        assert acc.dtype == np.uint64 and acc.flags.c_contiguous and acc.shape == (2, self.K, self.N_enc)
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        assert rows.ndim == 2 and rows.shape[1] == self.N
        w = key.shape[0] if window is None else window
        assert w <= key.shape[0] or window is None
        lib().rsf_inner_product_limb(self.h, limb, O.p64(key), 2 * self.K * self.N_enc, w, t0, O.p64(rows), rows.shape[0], O.p64(acc), threads)

    def __del__(self):
        if getattr(self, "h", None):
            lib().rsf_ctx_destroy(self.h)
            lib().rso_ctx_destroy(self.base)
            self.h = None


def max_threads():
    return int(lib().rso_max_threads())


def witness_map(q, cs, limb, assignment, d1=None, d2=None, d3=None, threads=1):
    """oracle.witness_map (the reference's O(m^2) algorithm) with Barrett products; cs: oracle.R1CSHandle."""
    assignment = np.ascontiguousarray(assignment, dtype=np.uint64)
    S, m = assignment.shape[1], cs.m
    o = {k: np.empty((m, S), dtype=np.uint64) for k in ("A_io", "B_io", "C_io", "A_mid", "B_mid", "C_mid")}
    o["Z"] = np.empty(m + 1, dtype=np.uint64)
    o["H"] = np.empty((m + 1, S), dtype=np.uint64)
    ds = [None if d is None else np.ascontiguousarray(d, dtype=np.uint64) for d in (d1, d2, d3)]
    args = [q, S, cs.ref(), limb, O.p64(assignment), O.p64(ds[0]), O.p64(ds[1]), O.p64(ds[2]),
            O.p64(o["A_io"]), O.p64(o["B_io"]), O.p64(o["C_io"]), O.p64(o["A_mid"]), O.p64(o["B_mid"]), O.p64(o["C_mid"]),
            O.p64(o["Z"]), O.p64(o["H"])]
    if threads == 1:
        lib().rso_witness_map(*args)
    else:
        lib().rso_witness_map_mt(*args, threads)
    return o
