"""ctypes loader for the CPU oracle (oracle/rs_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under ringsnark_amd/ does.  See oracle/rs_oracle.h for the parity status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librs_oracle.so")

MAXL, MAXK = 8, 12
KIND_POLY, KIND_ONE = 0, 2

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("rs_oracle.c", "rs_identities.c", "rs_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(_SO) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "librs_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class R1CS(C.Structure):
    _fields_ = [
        ("m", C.c_size_t),
        ("n_vars", C.c_size_t),
        ("n_inputs", C.c_size_t),
        ("row_ptr", u32p * 3),
        ("col", u32p * 3),
        ("coeff", u64p * 3),
        ("nnz", C.c_size_t * 3),
        ("pidx", C.POINTER(C.c_int32) * 3),
        ("ptab", u64p),
        ("ptab_L", C.c_size_t),
        ("ptab_N", C.c_size_t),
        ("ptab_slot0", C.c_size_t),
    ]


class WMVectors(C.Structure):
    _fields_ = [(k, u64p) for k in ("A_io", "B_io", "C_io", "A_mid", "B_mid", "C_mid", "H")] + [("stride", C.c_size_t * 7), ("Z", u64p), ("blocked", C.c_int)]


class Groth16PK(C.Structure):
    _fields_ = [("s_pows", u64p), ("delta_ts", u64p), ("delta_mid", u64p), ("alpha", u64p), ("beta", u64p)]


class RinocchioPK(C.Structure):
    _fields_ = [
        ("s_pows", u64p),
        ("alpha_s_pows", u64p),
        ("beta_prods", u64p),
        ("beta_rv_ts", u64p),
        ("beta_rw_ts", u64p),
        ("beta_ry_ts", u64p),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.rso_mulmod.restype = C.c_uint64
        L.rso_mulmod.argtypes = [C.c_uint64] * 3
        L.rso_powmod.restype = C.c_uint64
        L.rso_powmod.argtypes = [C.c_uint64] * 3
        L.rso_invmod.restype = C.c_uint64
        L.rso_invmod.argtypes = [C.c_uint64] * 2
        L.rso_is_prime.argtypes = [C.c_uint64]
        L.rso_get_primes.argtypes = [C.c_uint64, C.c_int, C.c_int, u64p]
        L.rso_coeff_modulus_create.argtypes = [C.c_uint64, C.POINTER(C.c_int), C.c_int, u64p]
        L.rso_minimal_primitive_root.argtypes = [C.c_uint64, C.c_uint64, u64p]
        L.rso_ntt_create.restype = C.c_void_p
        L.rso_ntt_create.argtypes = [C.c_int, C.c_uint64]
        L.rso_ntt_destroy.argtypes = [C.c_void_p]
        L.rso_ntt_fwd.argtypes = [C.c_void_p, u64p]
        L.rso_ntt_inv.argtypes = [C.c_void_p, u64p]
        L.rso_ctx_create.restype = C.c_void_p
        L.rso_ctx_create.argtypes = [C.c_int, C.c_int, u64p, C.c_int, C.c_int, u64p]
        L.rso_ctx_destroy.argtypes = [C.c_void_p]
        for name in ("rso_ring_add", "rso_ring_sub", "rso_ring_mul"):
            getattr(L, name).argtypes = [C.c_void_p, u64p, u64p, u64p]
        L.rso_ring_neg.argtypes = [C.c_void_p, u64p, u64p]
        L.rso_ring_mul_scalar.argtypes = [C.c_void_p, u64p, u64p, C.c_uint64]
        L.rso_ring_inv.argtypes = [C.c_void_p, u64p, u64p]
        L.rso_ring_is_zero.argtypes = [C.c_void_p, u64p]
        L.rso_batch_encode.argtypes = [C.c_void_p, C.c_int, u64p, u64p]
        L.rso_batch_decode.argtypes = [C.c_void_p, C.c_int, u64p, u64p]
        L.rso_multiply_plain.argtypes = [C.c_void_p, C.c_int, u64p, u64p]
        L.rso_ct_add.argtypes = [C.c_void_p, u64p, u64p]
        L.rso_enc_mul_ring.argtypes = [C.c_void_p, u64p, u64p]
        L.rso_enc_add.argtypes = [C.c_void_p, u64p, u64p]
        L.rso_inner_product.restype = C.c_size_t
        L.rso_inner_product.argtypes = [C.c_void_p, u64p, u64p, u8p, C.c_size_t, u64p]
        L.rso_keygen.argtypes = [C.c_void_p, C.c_uint64, u64p]
        L.rso_encrypt_symmetric.argtypes = [C.c_void_p, C.c_int, u64p, u64p, C.c_uint64, u64p]
        L.rso_decrypt.argtypes = [C.c_void_p, C.c_int, u64p, u64p, u64p]
        L.rso_enc_encode.argtypes = [C.c_void_p, u64p, u64p, C.c_uint64, u64p]
        L.rso_enc_decode.argtypes = [C.c_void_p, u64p, u64p, u64p]
        L.rso_noise_budget.argtypes = [C.c_void_p, C.c_int, u64p, u64p]
        L.rso_enc_decode_checked.argtypes = [C.c_void_p, u64p, u64p, u64p]
        L.rso_interpolate.argtypes = [C.c_uint64, C.c_size_t, C.c_size_t, u64p, u64p]
        L.rso_interpolate_nodes.argtypes = [C.c_uint64, C.c_size_t, C.c_size_t, u64p, u64p, u64p]
        L.rso_eval.argtypes = [C.c_uint64, C.c_size_t, C.c_size_t, u64p, C.c_uint64, u64p]
        L.rso_poly_mul.argtypes = [C.c_uint64, C.c_size_t, C.c_size_t, u64p, C.c_size_t, u64p, u64p]
        L.rso_poly_div.restype = C.c_size_t
        L.rso_poly_div.argtypes = [C.c_uint64, C.c_size_t, C.c_size_t, u64p, C.c_size_t, u64p, u64p]
        L.rso_poly_div_general.restype = C.c_size_t
        L.rso_poly_div_general.argtypes = L.rso_poly_div.argtypes
        L.rso_vanishing.argtypes = [C.c_uint64, C.c_size_t, u64p]
        L.rso_r1cs_evaluate.argtypes = [C.c_uint64, C.c_size_t, C.POINTER(R1CS), C.c_int, C.c_int, u64p, u64p]
        L.rso_witness_map.argtypes = [C.c_uint64, C.c_size_t, C.POINTER(R1CS), C.c_int] + [u64p] * 12
        L.rso_witness_map_mt.argtypes = [C.c_uint64, C.c_size_t, C.POINTER(R1CS), C.c_int] + [u64p] * 12 + [C.c_int]
        L.rso_max_threads.restype = C.c_int
        L.rso_inner_product_mt.restype = C.c_size_t
        L.rso_inner_product_mt.argtypes = [C.c_void_p, u64p, C.c_size_t, u64p, C.POINTER(C.c_uint8), C.c_size_t, u64p, C.c_int]
        L.rso_inner_product_slab.argtypes = [C.c_void_p, C.c_int, C.c_int, u64p, C.c_size_t, C.c_size_t, C.c_size_t, u64p,
                                             C.c_size_t, u64p, C.c_int]
        L.rso_groth16_prove.argtypes = [C.c_void_p, C.POINTER(R1CS), C.POINTER(Groth16PK), u64p, u64p, C.POINTER(C.c_int)]
        L.rso_rinocchio_prove.argtypes = [C.c_void_p, C.POINTER(R1CS), C.POINTER(RinocchioPK)] + [u64p] * 5 + [C.POINTER(C.c_int)]
        L.rso_groth16_prove_kinds.argtypes = [C.c_void_p, C.POINTER(R1CS), C.POINTER(Groth16PK), u64p, u8p, u64p, C.POINTER(C.c_int)]
        L.rso_groth16_prove_kinds.restype = None
        L.rso_rinocchio_prove_kinds.argtypes = [C.c_void_p, C.POINTER(R1CS), C.POINTER(RinocchioPK), u64p, u8p] + [u64p] * 4 + [C.POINTER(C.c_int)]
        L.rso_rinocchio_prove_kinds.restype = None
        L.rso_fill_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, u64p]
        L.rso_witness_identities.restype = C.c_size_t
        L.rso_witness_identities.argtypes = [C.c_uint64, C.c_size_t, C.POINTER(R1CS), C.c_int, u64p, C.c_size_t, u64p, u64p, u64p,
                                             C.POINTER(WMVectors), u64p, C.c_int, u8p, C.c_int]
        _lib = L
    return _lib


def p64(a):
    """uint64 pointer into a C-contiguous numpy uint64 array (None -> NULL)."""
    if a is None:
        return None
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(u64p)


def coeff_modulus_create(factor, bit_sizes):
    out = (C.c_uint64 * len(bit_sizes))()
    bs = (C.c_int * len(bit_sizes))(*bit_sizes)
    rc = lib().rso_coeff_modulus_create(factor, bs, len(bit_sizes), out)
    assert rc == 0
    return [int(x) for x in out]


def get_primes(factor, bits, count):
    out = (C.c_uint64 * count)()
    assert lib().rso_get_primes(factor, bits, count, out) == 0
    return [int(x) for x in out]


def minimal_primitive_root(degree, q):
    r = C.c_uint64()
    assert lib().rso_minimal_primitive_root(degree, q, C.byref(r)) == 0
    return int(r.value)


class NTT:
    def __init__(self, logn, q):
        self.h = lib().rso_ntt_create(logn, q)
        assert self.h, "no NTT tables for (logn=%d, q=%d)" % (logn, q)
        self.n, self.q = 1 << logn, q

    def fwd(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        lib().rso_ntt_fwd(self.h, p64(a))
        return a

    def inv(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        lib().rso_ntt_inv(self.h, p64(a))
        return a

    def __del__(self):
        if getattr(self, "h", None):
            lib().rso_ntt_destroy(self.h)
            self.h = None


class Ctx:
    """Oracle context: ring (N, q[L]) + encoding contexts (N_enc, Q[K])."""

    def __init__(self, N, q, N_enc, Q):
        self.N, self.L, self.N_enc, self.K = N, len(q), N_enc, len(Q)
        self.q, self.Q = [int(x) for x in q], [int(x) for x in Q]
        qa = (C.c_uint64 * self.L)(*self.q)
        Qa = (C.c_uint64 * self.K)(*self.Q)
        self.h = lib().rso_ctx_create(N, self.L, qa, N_enc, self.K, Qa)
        assert self.h, "rso_ctx_create failed"
        self.ring_words = self.L * N
        self.ct_words = 2 * self.K * N_enc
        self.enc_words = self.L * self.ct_words

    # ---- shapes
    def ring_shape(self, *lead):
        return tuple(lead) + (self.L, self.N)

    def enc_shape(self, *lead):
        return tuple(lead) + (self.L, 2, self.K, self.N_enc)

    def random_ring(self, seed, count=None):
        shape = self.ring_shape(*([count] if count is not None else []))
        out = np.empty(shape, dtype=np.uint64)
        flat = out.reshape(-1, self.L, self.N)
        for t in range(flat.shape[0]):
            for i in range(self.L):
                lib().rso_fill_uniform(seed * 1000003 + t * 17 + i, self.q[i], self.N, p64(flat[t, i]))
        return out

    def random_enc(self, seed, count=None):
        """Uniform residues < Q_j in encoding layout ("synthetic CRS")."""
        shape = self.enc_shape(*([count] if count is not None else []))
        out = np.empty(shape, dtype=np.uint64)
        flat = out.reshape(-1, self.L, 2, self.K, self.N_enc)
        for t in range(flat.shape[0]):
            for i in range(self.L):
                for c in range(2):
                    for j in range(self.K):
                        lib().rso_fill_uniform(
                            seed * 7919 + ((t * self.L + i) * 2 + c) * self.K + j, self.Q[j], self.N_enc, p64(flat[t, i, c, j])
                        )
        return out

    # ---- ring ops
    def _bin(self, fn, a, b):
        d = np.empty_like(a)
        fa, fb, fd = a.reshape(-1, self.ring_words), b.reshape(-1, self.ring_words), d.reshape(-1, self.ring_words)
        for t in range(fa.shape[0]):
            fn(self.h, p64(fd[t]), p64(fa[t]), p64(fb[t]))
        return d

    def ring_add(self, a, b):
        return self._bin(lib().rso_ring_add, a, b)

    def ring_sub(self, a, b):
        return self._bin(lib().rso_ring_sub, a, b)

    def ring_mul(self, a, b):
        return self._bin(lib().rso_ring_mul, a, b)

    def ring_neg(self, a):
        d = np.empty_like(a)
        fa, fd = a.reshape(-1, self.ring_words), d.reshape(-1, self.ring_words)
        for t in range(fa.shape[0]):
            lib().rso_ring_neg(self.h, p64(fd[t]), p64(fa[t]))
        return d

    def ring_mul_scalar(self, a, s):
        d = np.empty_like(a)
        fa, fd = a.reshape(-1, self.ring_words), d.reshape(-1, self.ring_words)
        for t in range(fa.shape[0]):
            lib().rso_ring_mul_scalar(self.h, p64(fd[t]), p64(fa[t]), s)
        return d

    def ring_inv(self, a):
        d = np.empty_like(a)
        fa, fd = a.reshape(-1, self.ring_words), d.reshape(-1, self.ring_words)
        ok = True
        for t in range(fa.shape[0]):
            ok = bool(lib().rso_ring_inv(self.h, p64(fd[t]), p64(fa[t]))) and ok
        return d, ok

    def ring_scalar(self, s):
        """RingElem(uint64) promoted to a polynomial (seal_ring.tcc:265-277)."""
        out = np.empty(self.ring_shape(), dtype=np.uint64)
        for i in range(self.L):
            out[i, :] = s % self.q[i]
        return out

    # ---- encoding ops
    def batch_encode(self, limb, values):
        out = np.empty(self.N_enc, dtype=np.uint64)
        lib().rso_batch_encode(self.h, limb, p64(np.ascontiguousarray(values)), p64(out))
        return out

    def batch_decode(self, limb, plain):
        out = np.empty(self.N, dtype=np.uint64)
        lib().rso_batch_decode(self.h, limb, p64(np.ascontiguousarray(plain)), p64(out))
        return out

    def multiply_plain(self, limb, ct, plain):
        ct = np.ascontiguousarray(ct).copy()
        lib().rso_multiply_plain(self.h, limb, p64(ct), p64(np.ascontiguousarray(plain)))
        return ct

    def enc_mul_ring(self, enc, ring):
        enc = np.ascontiguousarray(enc).copy()
        lib().rso_enc_mul_ring(self.h, p64(enc), p64(np.ascontiguousarray(ring)))
        return enc

    def enc_add(self, a, b):
        a = np.ascontiguousarray(a).copy()
        lib().rso_enc_add(self.h, p64(a), p64(np.ascontiguousarray(b)))
        return a

    def inner_product(self, encs, rings, kinds=None, threads=1, window=0):
        """EncodingElem::inner_product; threads != 1: terms spread over OpenMP threads (0 = all cores);
        window != 0 (threads != 1 only): encs holds `window` elements, term t uses encs[t % window]."""
        encs, rings = np.ascontiguousarray(encs), np.ascontiguousarray(rings)
        T = rings.shape[0]
        assert encs.shape[0] == (window or T) and (threads != 1 or not window)
        out = np.zeros(self.enc_shape(), dtype=np.uint64)
        kp = None
        if kinds is not None:
            kinds = np.ascontiguousarray(kinds, dtype=np.uint8)
            kp = kinds.ctypes.data_as(u8p)
        if threads == 1:
            used = lib().rso_inner_product(self.h, p64(encs), p64(rings), kp, T, p64(out))
        else:
            used = lib().rso_inner_product_mt(self.h, p64(encs), window, p64(rings), kp, T, p64(out), threads)
        return out, int(used)

    def inner_product_slab(self, limb, j, ct_slabs, rows, acc, t0=0, window=None, threads=0):
        """acc[N_enc] += sum_t ct_slabs[(t0+t) % window] * NTT_Qj(lift(encode(rows[t]))): one (limb, component,
        prime) slab of an inner product.  ct_slabs [W][N_enc] (that slab of every stored key element), rows [T][N]."""
        ct_slabs, rows = np.ascontiguousarray(ct_slabs, dtype=np.uint64), np.ascontiguousarray(rows, dtype=np.uint64)
        assert ct_slabs.shape[1] == self.N_enc and rows.shape[1] == self.N and acc.shape == (self.N_enc,)
        W = ct_slabs.shape[0] if window is None else window
        assert W <= ct_slabs.shape[0]
        lib().rso_inner_product_slab(self.h, limb, j, p64(ct_slabs), self.N_enc, W, t0, p64(rows), rows.shape[0], p64(acc), threads)
        return acc

    def keygen(self, seed):
        sk = np.empty((self.K, self.N_enc), dtype=np.uint64)
        lib().rso_keygen(self.h, seed, p64(sk))
        return sk

    def enc_encode(self, sk, rings, seed):
        rings = np.ascontiguousarray(rings).reshape(-1, self.L, self.N)
        out = np.empty(self.enc_shape(rings.shape[0]), dtype=np.uint64)
        for t in range(rings.shape[0]):
            lib().rso_enc_encode(self.h, p64(sk), p64(rings[t]), seed * 65537 + t, p64(out[t]))
        return out

    def enc_decode(self, sk, enc):
        out = np.empty(self.ring_shape(), dtype=np.uint64)
        lib().rso_enc_decode(self.h, p64(sk), p64(np.ascontiguousarray(enc)), p64(out))
        return out

    def noise_budget(self, sk, enc):
        """Decryptor::invariant_noise_budget of the L ciphertexts of one encoding element (bits; 0 = spent)."""
        enc = np.ascontiguousarray(enc).reshape(self.L, self.ct_words)
        return [int(lib().rso_noise_budget(self.h, i, p64(sk), p64(enc[i]))) for i in range(self.L)]

    def enc_decode_checked(self, sk, enc):
        """decode with the reference's guard (seal_ring.tcc:443-454): (ring, -1) or (None, i) for the first ciphertext
        #i whose noise budget is spent (the reference throws decoding_error there)."""
        out = np.empty(self.ring_shape(), dtype=np.uint64)
        i = int(lib().rso_enc_decode_checked(self.h, p64(sk), p64(np.ascontiguousarray(enc)), p64(out)))
        return (out, -1) if i < 0 else (None, i)

    def __del__(self):
        if getattr(self, "h", None):
            lib().rso_ctx_destroy(self.h)
            self.h = None


# ---- slot-array algebra (one prime, arrays [n][S]) -------------------------------------------
def interpolate(q, y):
    y = np.ascontiguousarray(y, dtype=np.uint64)
    n, S = y.shape
    out = np.empty_like(y)
    lib().rso_interpolate(q, S, n, p64(y), p64(out))
    return out


def interpolate_nodes(q, x, y):
    y = np.ascontiguousarray(y, dtype=np.uint64)
    x = np.ascontiguousarray(x, dtype=np.uint64)
    n, S = y.shape
    out = np.empty_like(y)
    lib().rso_interpolate_nodes(q, S, n, p64(x), p64(y), p64(out))
    return out


def poly_eval(q, coeffs, x):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
    n, S = coeffs.shape
    out = np.empty(S, dtype=np.uint64)
    lib().rso_eval(q, S, n, p64(coeffs), x, p64(out))
    return out


def poly_mul(q, a, b):
    a, b = np.ascontiguousarray(a, dtype=np.uint64), np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty((a.shape[0] + b.shape[0] - 1, a.shape[1]), dtype=np.uint64)
    lib().rso_poly_mul(q, a.shape[1], a.shape[0], p64(a), b.shape[0], p64(b), p64(out))
    return out


def poly_div(q, num, den_scalars):
    num = np.ascontiguousarray(num, dtype=np.uint64)
    den = np.ascontiguousarray(den_scalars, dtype=np.uint64)
    out = np.zeros((num.shape[0] - den.shape[0] + 1, num.shape[1]), dtype=np.uint64)
    n = lib().rso_poly_div(q, num.shape[1], num.shape[0], p64(num), den.shape[0], p64(den), p64(out))
    return out, int(n)


def poly_div_general(q, num, den):
    num = np.ascontiguousarray(num, dtype=np.uint64)
    den = np.ascontiguousarray(den, dtype=np.uint64)
    out = np.zeros((num.shape[0] - den.shape[0] + 1, num.shape[1]), dtype=np.uint64)
    n = lib().rso_poly_div_general(q, num.shape[1], num.shape[0], p64(num), den.shape[0], p64(den), p64(out))
    return out, int(n)


def vanishing(q, m):
    out = np.empty(m + 1, dtype=np.uint64)
    lib().rso_vanishing(q, m, p64(out))
    return out


class R1CSHandle:
    """Keeps the numpy buffers behind an rso_r1cs alive.

    mats: dict a/b/c -> (row_ptr uint32[m+1], col uint32[nnz], coeff uint64[L][nnz]).
    poly_idx / poly_table: coefficients that are general ring elements (rs_oracle.h): poly_idx[name] int32[nnz]
    (-1 = the scalar of mats), poly_table uint64[n_poly][L][N]."""

    def __init__(self, m, n_vars, n_inputs, mats, poly_idx=None, poly_table=None, slot0=0):
        self.m, self.n_vars, self.n_inputs = m, n_vars, n_inputs
        self.s = R1CS()
        self.s.m, self.s.n_vars, self.s.n_inputs = m, n_vars, n_inputs
        self._keep = []
        self._args = (m, n_vars, n_inputs, mats, poly_idx, poly_table)
        for k, name in enumerate("abc"):
            rp, col, cf = mats[name]
            rp = np.ascontiguousarray(rp, dtype=np.uint32)
            col = np.ascontiguousarray(col, dtype=np.uint32)
            cf = np.ascontiguousarray(cf, dtype=np.uint64)
            self._keep += [rp, col, cf]
            self.s.row_ptr[k] = rp.ctypes.data_as(u32p)
            self.s.col[k] = col.ctypes.data_as(u32p)
            self.s.coeff[k] = cf.ctypes.data_as(u64p)
            self.s.nnz[k] = col.shape[0]
            if poly_table is not None and poly_idx is not None and poly_idx.get(name) is not None:
                pi = np.ascontiguousarray(poly_idx[name], dtype=np.int32)
                assert pi.shape == col.shape
                self._keep.append(pi)
                self.s.pidx[k] = pi.ctypes.data_as(C.POINTER(C.c_int32))
        if poly_table is not None:
            pt = np.ascontiguousarray(poly_table, dtype=np.uint64)
            assert pt.ndim == 3
            self._keep.append(pt)
            self.s.ptab = pt.ctypes.data_as(u64p)
            self.s.ptab_L, self.s.ptab_N, self.s.ptab_slot0 = pt.shape[1], pt.shape[2], slot0

    def at_slots(self, slot0):
        """The same system for calls whose arrays start at ring slot `slot0` of every limb."""
        return self if slot0 == self.s.ptab_slot0 else R1CSHandle(*self._args, slot0=slot0)

    def ref(self):
        return C.byref(self.s)


def r1cs_evaluate(q, cs, which, limb, assignment):
    assignment = np.ascontiguousarray(assignment, dtype=np.uint64)
    S = assignment.shape[1]
    out = np.empty((cs.m, S), dtype=np.uint64)
    lib().rso_r1cs_evaluate(q, S, cs.ref(), which, limb, p64(assignment), p64(out))
    return out


def max_threads():
    return int(lib().rso_max_threads())


def witness_map(q, cs, limb, assignment, d1=None, d2=None, d3=None, threads=1):
    """One limb; assignment [n_vars][S].  Returns dict of A_io..C_mid [m][S], Z [m+1], H [m+1][S].
    threads != 1: the S slots are spread over OpenMP threads (0 = all cores)."""
    assignment = np.ascontiguousarray(assignment, dtype=np.uint64)
    S, m = assignment.shape[1], cs.m
    o = {k: np.empty((m, S), dtype=np.uint64) for k in ("A_io", "B_io", "C_io", "A_mid", "B_mid", "C_mid")}
    o["Z"] = np.empty(m + 1, dtype=np.uint64)
    o["H"] = np.empty((m + 1, S), dtype=np.uint64)
    ds = [None if d is None else np.ascontiguousarray(d, dtype=np.uint64) for d in (d1, d2, d3)]
    args = [q, S, cs.ref(), limb, p64(assignment), p64(ds[0]), p64(ds[1]), p64(ds[2]),
            p64(o["A_io"]), p64(o["B_io"]), p64(o["C_io"]), p64(o["A_mid"]), p64(o["B_mid"]), p64(o["C_mid"]),
            p64(o["Z"]), p64(o["H"])]
    if threads == 1:
        lib().rso_witness_map(*args)
    else:
        lib().rso_witness_map_mt(*args, threads)
    return o


IDENTITY_NAMES = ("A_io", "B_io", "C_io", "A_mid", "B_mid", "C_mid", "H")


def witness_identities(q, cs, limb, assignment, vectors, points, d1=None, d2=None, d3=None, Z=None, threads=0, blocked=False):
    """COMPLETE check of a witness map computed elsewhere (rs_identities.c): one limb, EVERY slot.
    assignment [n_vars][S]; vectors: dict name -> [m][S] ([m+1][S] for H) -- arrays may be strided views of
    [rows][L][N] host arrays sliced to one limb (row stride taken from the array, slots contiguous); points: <= 4 integers in [m, q); d1..d3 [S]; Z [m+1].
    Returns (number of failing slots, bad [S] uint8 bit mask by IDENTITY_NAMES order)."""
    def rows(a):
        assert a.dtype == np.uint64 and a.ndim == 2 and a.strides[1] == 8 and a.strides[0] % 8 == 0, (a.dtype, a.shape, a.strides)
        return a.ctypes.data_as(u64p), a.strides[0] // 8

    v = WMVectors()
    if blocked:  # blocked=True: every array is C-contiguous [S/32][rows][32] (32 slots of a block together, row after row)
        S = assignment.shape[0] * 32
        assert assignment.dtype == np.uint64 and assignment.flags["C_CONTIGUOUS"] and assignment.shape[1:] == (cs.n_vars, 32), assignment.shape
        ap, astride = assignment.ctypes.data_as(u64p), 0
        v.blocked = 1
    else:
        S = assignment.shape[1]
        ap, astride = rows(assignment)
    for k, name in enumerate(IDENTITY_NAMES):
        a = vectors.get(name)
        if a is None:
            continue
        if blocked:
            assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"] and a.shape == (S // 32, cs.m + (name == "H"), 32), (name, a.shape)
            ptr = a.ctypes.data_as(u64p)
        else:
            assert a.shape == (cs.m + (name == "H"), S), (name, a.shape)
            ptr, v.stride[k] = rows(a)
        setattr(v, name, ptr)
    if Z is not None:
        Z = np.ascontiguousarray(Z, dtype=np.uint64)
        assert Z.shape == (cs.m + 1,)
        v.Z = p64(Z)
    ds = [None if d is None else np.ascontiguousarray(d, dtype=np.uint64) for d in (d1, d2, d3)]
    pts = np.ascontiguousarray([int(p) for p in points], dtype=np.uint64)
    bad = np.zeros(S, dtype=np.uint8)
    n = lib().rso_witness_identities(q, S, cs.ref(), limb, ap, astride, p64(ds[0]), p64(ds[1]), p64(ds[2]), C.byref(v),
                                     p64(pts), len(pts), bad.ctypes.data_as(u8p), threads)
    n = int(n)
    if n == 2**64 - 1:
        raise ValueError("witness_identities: bad arguments (a point below m, or more than 4 points)")
    if n == 2**64 - 2:
        raise AssertionError("witness_identities: Z is not prod (x - i)")
    return n, bad


def _kinds_arg(kinds, n):
    if kinds is None:
        return None, None
    k = np.ascontiguousarray(kinds, dtype=np.uint8)
    assert k.shape == (n,)
    return k, k.ctypes.data_as(u8p)


def groth16_prove(ctx, cs, pk, assignment, kinds=None):
    """pk: dict s_pows, delta_ts, delta_mid, alpha, beta (numpy, encoding layout).  kinds: per assignment wire,
    KIND_ONE for a RingElem holding Scalar 1 (seal_ring.tcc:525-527), None = all polynomials."""
    keep = {k: np.ascontiguousarray(v, dtype=np.uint64) for k, v in pk.items()}
    s = Groth16PK(*[p64(keep[k]) for k in ("s_pows", "delta_ts", "delta_mid", "alpha", "beta")])
    proof = np.zeros(ctx.enc_shape(3), dtype=np.uint64)
    empty = (C.c_int * 3)()
    assignment = np.ascontiguousarray(assignment, dtype=np.uint64)
    keep_k, kp = _kinds_arg(kinds, assignment.shape[0])
    lib().rso_groth16_prove_kinds(ctx.h, cs.ref(), C.byref(s), p64(assignment), kp, p64(proof), empty)
    return proof, [int(e) for e in empty]


def rinocchio_prove(ctx, cs, pk, assignment, d1=None, d2=None, d3=None, kinds=None):
    keep = {k: np.ascontiguousarray(v, dtype=np.uint64) for k, v in pk.items()}
    s = RinocchioPK(*[p64(keep[k]) for k in ("s_pows", "alpha_s_pows", "beta_prods", "beta_rv_ts", "beta_rw_ts", "beta_ry_ts")])
    proof = np.zeros(ctx.enc_shape(9), dtype=np.uint64)
    empty = (C.c_int * 9)()
    assignment = np.ascontiguousarray(assignment, dtype=np.uint64)
    ds = [None if d is None else np.ascontiguousarray(d, dtype=np.uint64) for d in (d1, d2, d3)]
    keep_k, kp = _kinds_arg(kinds, assignment.shape[0])
    lib().rso_rinocchio_prove_kinds(ctx.h, cs.ref(), C.byref(s), p64(assignment), kp, p64(ds[0]), p64(ds[1]), p64(ds[2]), p64(proof), empty)
    return proof, [int(e) for e in empty]
