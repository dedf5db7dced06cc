"""Device context: the Python host side above the C ABI (include/ringsnark_amd.h).

PyTorch is used for what it is good at here -- device memory, streams, torch.distributed -- and
nothing else: every arithmetic operation goes through librs_hip.so.  Residues live in int64 CUDA
tensors (bit-identical to the uint64 boundary layout; all values are < 2^50).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .params import RingParams
from .r1cs import R1CS


def to_device(a: np.ndarray, device) -> torch.Tensor:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return torch.from_numpy(a.view(np.int64)).to(device)


def to_host(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().contiguous().numpy().view(np.uint64)


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous() and t.dtype == torch.int64, (t.device, t.dtype, t.is_contiguous())
    return C.c_void_p(t.data_ptr())


class HostWords:
    """Page-locked host buffer of uint64 words (rs_host_alloc); `array` is a numpy view of it."""

    def __init__(self, dev, words):
        self.dev, self.words = dev, int(words)
        p = C.c_void_p()
        _lib.check(dev.lib.rs_host_alloc(dev.h, self.words * 8, C.byref(p)))
        self.ptr = p.value
        self.array = np.ctypeslib.as_array((C.c_uint64 * self.words).from_address(self.ptr))

    def fill_from(self, tensor, offset_words=0):
        """device tensor (int64) -> this buffer at offset_words (rs_download)."""
        t = tensor.contiguous()
        assert offset_words + t.numel() <= self.words
        _lib.check(self.dev.lib.rs_download(self.dev.h, C.c_void_p(self.ptr + 8 * offset_words), C.c_void_p(t.data_ptr()), t.numel() * 8, self.dev.stream()))
        self.dev.sync()

    def __del__(self):
        if getattr(self, "ptr", None) and self.dev.lib is not None and getattr(self.dev, "h", None):
            self.array = None
            self.dev.lib.rs_host_free(self.dev.h, C.c_void_p(self.ptr))
            self.ptr = None


class DeviceR1CS:
    def __init__(self, dev, cs: R1CS):
        self.dev, self.cs = dev, cs
        self.m, self.n_vars, self.n_inputs = cs.m, cs.n_vars, cs.n_inputs
        rp = (_lib.u32p * 3)()
        col = (_lib.u32p * 3)()
        cf = (_lib.u64p * 3)()
        nnz = (C.c_size_t * 3)()
        keep = []
        for k, name in enumerate("abc"):
            r, c, f = cs.mats[name]
            r = np.ascontiguousarray(r, dtype=np.uint32)
            c = np.ascontiguousarray(c, dtype=np.uint32)
            f = np.ascontiguousarray(f, dtype=np.uint64)
            assert f.shape == (dev.L, c.shape[0])
            keep += [r, c, f]
            rp[k] = r.ctypes.data_as(_lib.u32p)
            col[k] = c.ctypes.data_as(_lib.u32p)
            cf[k] = f.ctypes.data_as(_lib.u64p)
            nnz[k] = c.shape[0]
        h = C.c_void_p()
        if cs.poly_table is None:
            _lib.check(dev.lib.rs_r1cs_create(dev.h, cs.m, cs.n_vars, cs.n_inputs, rp, col, cf, nnz, C.byref(h)))
        else:  # coefficients that are general ring elements (relations/variable.tcc:246-254)
            i32p = C.POINTER(C.c_int32)
            pidx = (i32p * 3)()
            for k, name in enumerate("abc"):
                pi = np.ascontiguousarray(cs.poly_idx[name], dtype=np.int32)
                assert pi.shape[0] == nnz[k]
                keep.append(pi)
                pidx[k] = pi.ctypes.data_as(i32p)
            tab = np.ascontiguousarray(cs.poly_table, dtype=np.uint64)
            assert tab.shape[1:] == (dev.L, dev.N)
            _lib.check(dev.lib.rs_r1cs_create_poly(dev.h, cs.m, cs.n_vars, cs.n_inputs, rp, col, cf, nnz, pidx,
                                                   tab.ctypes.data_as(_lib.u64p), tab.shape[0], C.byref(h)))
        self.h = h

    def __del__(self):
        if getattr(self, "h", None) and self.dev.lib is not None:
            self.dev.lib.rs_r1cs_destroy(self.h)
            self.h = None


class Device:
    """rs_ctx wrapper.  Mirrors RingElem::set_context + EncodingElem::set_context
    (seal/seal_ring.hpp:52-58, 266-320): one object per (process, GPU)."""

    def __init__(self, prm: RingParams, device_index: int = 0):
        self.lib = _lib.load()  # raises if librs_hip.so is missing -- no fallback
        if not torch.cuda.is_available():
            raise RuntimeError("ringsnark_amd needs a HIP device (torch.cuda.is_available() is False)")
        self.prm = prm
        self.N, self.L, self.N_enc, self.K = prm.N, prm.L, prm.N_enc, prm.K
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        q = (C.c_uint64 * self.L)(*prm.q)
        Q = (C.c_uint64 * self.K)(*prm.Q)
        h = C.c_void_p()
        _lib.check(self.lib.rs_ctx_create(device_index, self.N, self.L, q, self.N_enc, self.K, Q, C.byref(h)))
        self.h = h
        self.ring_words, self.enc_words = prm.ring_words, prm.enc_words

    def __del__(self):
        if getattr(self, "h", None) and self.lib is not None:
            self.lib.rs_ctx_destroy(self.h)
            self.h = None

    # ---- helpers
    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def ring_empty(self, *lead):
        return torch.empty(tuple(lead) + (self.L, self.N), dtype=torch.int64, device=self.device)

    def enc_empty(self, *lead):
        return torch.empty(tuple(lead) + (self.L, 2, self.K, self.N_enc), dtype=torch.int64, device=self.device)

    def put(self, a):
        return to_device(a, self.device)

    def sync(self):
        _lib.check(self.lib.rs_sync(self.h, self.stream()))

    @staticmethod
    def _count(t, words):
        assert t.numel() % words == 0
        return t.numel() // words

    # ---- a4
    def ntt(self, data, modset, index, inverse=False):
        """In-place batched negacyclic NTT over [batch][N_enc]."""
        batch = self._count(data, self.N_enc)
        fn = self.lib.rs_ntt_inverse if inverse else self.lib.rs_ntt_forward
        _lib.check(fn(self.h, modset, index, _ptr(data), batch, self.stream()))
        return data

    # ---- a1-a3
    def _bin(self, fn, a, b):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        _lib.check(fn(self.h, _ptr(out), _ptr(a), _ptr(b), self._count(a, self.ring_words), self.stream()))
        return out

    def ring_add(self, a, b):
        return self._bin(self.lib.rs_ring_add, a, b)

    def ring_sub(self, a, b):
        return self._bin(self.lib.rs_ring_sub, a, b)

    def ring_mul(self, a, b):
        return self._bin(self.lib.rs_ring_mul, a, b)

    def ring_neg(self, a):
        out = torch.empty_like(a)
        _lib.check(self.lib.rs_ring_neg(self.h, _ptr(out), _ptr(a), self._count(a, self.ring_words), self.stream()))
        return out

    def ring_add_scalar(self, a, s):
        a = a.contiguous()
        out = torch.empty_like(a)
        _lib.check(self.lib.rs_ring_add_scalar(self.h, _ptr(out), _ptr(a), s, self._count(a, self.ring_words), self.stream()))
        return out

    def ring_mul_scalar(self, a, s):
        a = a.contiguous()
        out = torch.empty_like(a)
        _lib.check(self.lib.rs_ring_mul_scalar(self.h, _ptr(out), _ptr(a), s, self._count(a, self.ring_words), self.stream()))
        return out

    def ring_inv(self, a):
        """Raises RsError(RS_ERR_NOT_INVERTIBLE, "element is not invertible in ring")."""
        out = torch.empty_like(a)
        _lib.check(self.lib.rs_ring_inv(self.h, _ptr(out), _ptr(a), self._count(a, self.ring_words), self.stream()))
        return out

    def ring_is_zero(self, a):
        count = self._count(a, self.ring_words)
        flags = (C.c_uint8 * count)()
        _lib.check(self.lib.rs_ring_is_zero(self.h, _ptr(a), count, flags, self.stream()))
        return [bool(f) for f in flags]

    # ---- a5-a8
    def batch_encode(self, rings):
        count = self._count(rings, self.ring_words)
        out = torch.empty((count, self.L, self.N_enc), dtype=torch.int64, device=self.device)
        _lib.check(self.lib.rs_batch_encode(self.h, _ptr(rings), _ptr(out), count, self.stream()))
        return out

    def enc_mul_ring(self, enc, ring):
        enc = enc.clone()
        _lib.check(self.lib.rs_enc_mul_ring(self.h, _ptr(enc), _ptr(ring), self._count(enc, self.enc_words), self.stream()))
        return enc

    def enc_add(self, a, b):
        out = torch.empty_like(a)
        _lib.check(self.lib.rs_enc_add(self.h, _ptr(out), _ptr(a), _ptr(b), self._count(a, self.enc_words), self.stream()))
        return out

    def enc_reduce(self, enc):
        """In-place x mod Q_j on integer sums of encoding elements (multi-GPU all-reduce epilogue)."""
        _lib.check(self.lib.rs_enc_reduce(self.h, _ptr(enc), self._count(enc, self.enc_words), self.stream()))
        return enc

    # ---- 8(f) f2: instance map with evaluation
    def instance_map_eval(self, dcs, s):
        """r1cs_to_qrp_instance_map_with_evaluation (r1cs_to_qrp.tcc:76-116): returns At, Bt, Ct
        [n_vars+1][L][N], Ht [m+1][L][N], Zt [L][N] for the point s [L][N]."""
        n1 = dcs.cs.n_vars + 1
        At, Bt, Ct = self.ring_empty(n1), self.ring_empty(n1), self.ring_empty(n1)
        Ht, Zt = self.ring_empty(dcs.cs.m + 1), self.ring_empty()
        _lib.check(self.lib.rs_instance_map_eval(self.h, dcs.h, _ptr(s), _ptr(At), _ptr(Bt), _ptr(Ct), _ptr(Ht), _ptr(Zt),
                                                 self.stream()))
        return At, Bt, Ct, Ht, Zt

    # ---- 8(f) f4
    def enc_serialize(self, enc, empty=None):
        """Encoding elements (a proof, a key vector) -> bytes in the wire format of ringsnark_amd.h."""
        import numpy as np
        count = self._count(enc, self.enc_words)
        size = self.lib.rs_enc_wire_size(self.h, count)
        buf = np.empty(size, dtype=np.uint8)
        em = None
        if empty is not None:
            em = np.ascontiguousarray(empty, dtype=np.uint8)
            assert em.size == count
        _lib.check(self.lib.rs_enc_serialize(self.h, _ptr(enc), None if em is None else em.ctypes.data_as(_lib.u8p), count,
                                             buf.ctypes.data_as(C.c_void_p), size, self.stream()))
        return buf.tobytes()

    def enc_deserialize(self, data):
        """bytes -> (encodings [count][L][2][K][N_enc] on the device, empty flags).  Validates the stream."""
        import numpy as np
        buf = np.frombuffer(data, dtype=np.uint8)
        cnt = C.c_size_t(0)
        _lib.check(self.lib.rs_enc_deserialize(self.h, buf.ctypes.data_as(C.c_void_p), buf.size, None, None, 0, C.byref(cnt),
                                               self.stream()))
        out = self.enc_empty(cnt.value)
        em = np.zeros(cnt.value, dtype=np.uint8)
        _lib.check(self.lib.rs_enc_deserialize(self.h, buf.ctypes.data_as(C.c_void_p), buf.size, _ptr(out),
                                               em.ctypes.data_as(_lib.u8p), cnt.value, C.byref(cnt), self.stream()))
        return out, em

    # ---- 8(f) f2 / f3
    def enc_decode(self, sk, enc):
        """EncodingElem::decode (seal_ring.tcc:435-477): sk [K][N_enc] NTT form -> ring elements.  Raises
        RsError(RS_ERR_NOISE, "ciphertext #i has remaining noise budget 0 <= 0") like the reference's decoding_error
        (seal_ring.tcc:446-454) when a ciphertext's invariant noise budget is spent."""
        count = self._count(enc, self.enc_words)
        out = self.ring_empty(count) if enc.dim() > 4 else self.ring_empty()
        _lib.check(self.lib.rs_enc_decode(self.h, _ptr(sk), _ptr(enc), count, _ptr(out), self.stream()))
        return out

    def enc_noise_budget(self, sk, enc):
        """Decryptor::invariant_noise_budget of every ciphertext (bits; 0 = spent): int array [count][L]."""
        count = self._count(enc, self.enc_words)
        out = (C.c_int * (count * self.L))()
        _lib.check(self.lib.rs_enc_noise_budget(self.h, _ptr(sk), _ptr(enc), count, out, self.stream()))
        return np.array(out, dtype=np.int64).reshape(count, self.L)

    def enc_encode(self, sk, rings, seed):
        """EncodingElem::encode (seal_ring.tcc:324-359); element k uses the oracle's stream seed*65537 + k."""
        count = self._count(rings, self.ring_words)
        out = self.enc_empty(count) if rings.dim() > 2 else self.enc_empty()
        _lib.check(self.lib.rs_enc_encode(self.h, _ptr(sk), _ptr(rings), count, C.c_uint64((seed * 65537) % 2**64), _ptr(out),
                                          self.stream()))
        return out

    # ---- a9
    def inner_product(self, encs, rings, kinds=None, want_used=True):
        """EncodingElem::inner_product.  Returns (out, used); used == 0 <=> EMPTY element."""
        T = self._count(rings, self.ring_words)
        assert self._count(encs, self.enc_words) == T
        out = self.enc_empty()
        used = C.c_size_t(0)
        kp = None
        if kinds is not None:
            kinds = np.ascontiguousarray(kinds, dtype=np.uint8)
            kp = kinds.ctypes.data_as(_lib.u8p)
        _lib.check(self.lib.rs_inner_product(self.h, _ptr(encs), _ptr(rings), kp, T, _ptr(out),
                                             C.byref(used) if want_used else None, self.stream()))
        return out, int(used.value)

    def msm(self, crs_list, vecs, n_groups, want_used=False, crs_len=None, window=0):
        """vecs: list of (coeff tensor [T][L][N], kinds or None, group) or, for a SLOT-CONSTANT vector (one value per
        (term, limb) in every slot: coefficients_for_Z), (tensor [T][L], kinds, group, True).  window != 0: the CRS tensors hold
        `window` elements and logical element t is read from t % window (crs_len = logical length).
        CRS vectors given as HostWords (host_alloc) are streamed from host memory (rs_msm_hostkey)."""
        n_crs = len(crs_list)
        on_host = isinstance(crs_list[0], HostWords)
        assert all(isinstance(c, HostWords) == on_host for c in crs_list)
        if crs_len is None:
            crs_len = crs_list[0].words // self.enc_words if on_host else self._count(crs_list[0], self.enc_words)
        crs = (C.c_void_p * n_crs)(*[(c.ptr if on_host else c.data_ptr()) for c in crs_list])
        mv = (_lib.MsmVec * len(vecs))()
        keep = []
        for k, vec in enumerate(vecs):
            coeff, kinds, group = vec[:3]
            slot_const = len(vec) > 3 and bool(vec[3])
            assert coeff.is_contiguous()
            mv[k].d_coeff = coeff.data_ptr()
            mv[k].T = self._count(coeff, self.L if slot_const else self.ring_words)
            mv[k].group = group
            mv[k].slot_const = 1 if slot_const else 0
            if kinds is not None:
                kk = np.ascontiguousarray(kinds, dtype=np.uint8)
                keep.append(kk)
                mv[k].h_kinds = kk.ctypes.data_as(_lib.u8p)
        out = self.enc_empty(n_crs, n_groups)
        used = (C.c_size_t * len(vecs))()
        fn = self.lib.rs_msm_hostkey if on_host else self.lib.rs_msm
        _lib.check(fn(self.h, crs, n_crs, crs_len, window, mv, len(vecs), n_groups, _ptr(out), used if want_used else None, self.stream()))
        return out, [int(u) for u in used]

    # ---- a10-a14
    def r1cs(self, cs: R1CS):
        return DeviceR1CS(self, cs)

    def r1cs_evaluate(self, dcs, which, mode, assignment):
        out = self.ring_empty(dcs.m)
        _lib.check(self.lib.rs_r1cs_evaluate(self.h, dcs.h, which, mode, _ptr(assignment), _ptr(out), self.stream()))
        return out

    def interpolate(self, y):
        n = self._count(y, self.ring_words)
        out = torch.empty_like(y)
        _lib.check(self.lib.rs_interpolate(self.h, _ptr(y), _ptr(out), n, self.stream()))
        return out

    def witness_map(self, dcs, assignment, d1=None, d2=None, d3=None, want=("A_io", "B_io", "C_io", "A_mid", "B_mid", "C_mid", "H"),
                    rows=None):
        """rows: {name: (lo, hi)} -- keep only rows [lo, hi) of those outputs (rs_witness_map_rows: a rank of a limb
        group keeps the rows of its term range); outputs not named keep every row."""
        m = dcs.m
        names = ("A_io", "B_io", "C_io", "A_mid", "B_mid", "C_mid", "H")
        if rows is not None:
            full = {k: (0, m + 1 if k == "H" else m) for k in names}
            rr = {k: tuple(int(x) for x in rows.get(k, full[k])) for k in names}
            for a, b in (("A_io", "A_mid"), ("B_io", "B_mid"), ("C_io", "C_mid")):  # one pass writes both: same range
                if (a in rows) != (b in rows):  # a range given for one of the pair holds for both
                    rr[a] = rr[b] = rr[a if a in rows else b]
                assert rr[a] == rr[b] or a not in want or b not in want, (a, b, rr[a], rr[b])
                if a not in want:
                    rr[a] = rr[b]
                if b not in want:
                    rr[b] = rr[a]
            o = {k: (torch.empty((rr[k][1] - rr[k][0], self.L, self.N), dtype=torch.int64, device=self.device) if k in want else None) for k in names}
            flat = (C.c_size_t * 14)(*[x for k in names for x in rr[k]])
            Z = np.zeros((self.L, m + 1), dtype=np.uint64)
            _lib.check(self.lib.rs_witness_map_rows(
                self.h, dcs.h, _ptr(assignment), _ptr(d1), _ptr(d2), _ptr(d3), flat, _ptr(o["A_io"]), _ptr(o["B_io"]), _ptr(o["C_io"]),
                _ptr(o["A_mid"]), _ptr(o["B_mid"]), _ptr(o["C_mid"]), _ptr(o["H"]), Z.ctypes.data_as(_lib.u64p), self.stream()))
            o["Z"] = Z
            return o
        o = {}
        for k in ("A_io", "B_io", "C_io", "A_mid", "B_mid", "C_mid"):
            o[k] = self.ring_empty(m) if k in want else None
        o["H"] = self.ring_empty(m + 1) if "H" in want else None
        Z = np.zeros((self.L, m + 1), dtype=np.uint64)
        _lib.check(self.lib.rs_witness_map(
            self.h, dcs.h, _ptr(assignment), _ptr(d1), _ptr(d2), _ptr(d3), _ptr(o["A_io"]), _ptr(o["B_io"]), _ptr(o["C_io"]),
            _ptr(o["A_mid"]), _ptr(o["B_mid"]), _ptr(o["C_mid"]), _ptr(o["H"]), Z.ctypes.data_as(_lib.u64p), self.stream()))
        o["Z"] = Z
        return o

    def witness_map_slots(self, dcs, assignment, slot0, nslots, d1=None, d2=None, d3=None,
                          want=("A_io", "B_io", "C_io", "A_mid", "B_mid", "C_mid", "H")):
        """The witness map on slots [slot0, slot0+nslots) of every limb; outputs compact [t][L][nslots]."""
        m = dcs.m
        o = {}
        mk = lambda rows: torch.empty((rows, self.L, nslots), dtype=torch.int64, device=self.device)
        for k in ("A_io", "B_io", "C_io", "A_mid", "B_mid", "C_mid"):
            o[k] = mk(m) if k in want else None
        o["H"] = mk(m + 1) if "H" in want else None
        Z = np.zeros((self.L, m + 1), dtype=np.uint64)
        _lib.check(self.lib.rs_witness_map_slots(
            self.h, dcs.h, _ptr(assignment), _ptr(d1), _ptr(d2), _ptr(d3), slot0, nslots, _ptr(o["A_io"]), _ptr(o["B_io"]),
            _ptr(o["C_io"]), _ptr(o["A_mid"]), _ptr(o["B_mid"]), _ptr(o["C_mid"]), _ptr(o["H"]), Z.ctypes.data_as(_lib.u64p),
            self.stream()))
        o["Z"] = Z
        return o

    # ---- a11: util/polynomials.tcc:62-81
    def _poly(self, fn, a, b, rows, out=None):
        """out: optional caller buffer of the nominal row count (the library zeroes the rows beyond the result)."""
        na, nb = self._count(a, self.ring_words), self._count(b, self.ring_words)
        if out is None:
            out = torch.empty((max(rows(na, nb), 1), self.L, self.N), dtype=torch.int64, device=self.device)
        assert out.shape[0] >= max(rows(na, nb), 0)
        n = C.c_size_t(0)
        _lib.check(fn(self.h, _ptr(a) if na else None, na, _ptr(b) if nb else None, nb, _ptr(out), C.byref(n), self.stream()))
        return out[:n.value]

    def poly_multiply(self, a, b, out=None):
        """multiply(x, y): coefficient vectors [n][L][N]; the result is normalised like Boost's polynomial."""
        return self._poly(self.lib.rs_poly_multiply, a, b, lambda na, nb: na + nb - 1 if na and nb else 0, out)

    def poly_add(self, a, b, out=None):
        return self._poly(self.lib.rs_poly_add, a, b, lambda na, nb: max(na, nb), out)

    def poly_divide(self, num, den, out=None):
        """divide(numerator, denominator): the quotient; RsError(RS_ERR_NOT_INVERTIBLE) unless the divisor's leading
        coefficient is a unit."""
        return self._poly(self.lib.rs_poly_divide, num, den, lambda nn, nd: nn - nd + 1, out)

    # ---- a15 / a16
    def host_alloc(self, words):
        """Page-locked host memory of `words` uint64 (rs_host_alloc): where a proving key larger than HBM lives."""
        return HostWords(self, words)

    def _kinds(self, kinds, n):
        """per-wire representation (RS_KIND_*) as a host uint8 array, or (None, None)"""
        if kinds is None:
            return None, None
        k = np.ascontiguousarray(kinds, dtype=np.uint8)
        assert k.shape == (n,), (k.shape, n)
        return k, k.ctypes.data_as(_lib.u8p)

    def groth16_prove(self, dcs, pk, assignment, want_empty=True, window=0, kinds=None):
        """pk: dict s_pows, delta_ts, delta_mid, alpha, beta (CUDA tensors).  window != 0: the key vectors hold
        `window` elements each, element t read from t % window (tiled synthetic key, ringsnark_amd.h).
        Key VECTORS given as HostWords: a host-resident key, streamed tile by tile (rs_groth16_pk.host_key).
        kinds [n_vars]: RS_KIND_ONE for assignment wires held as RingElem Scalar 1 (rs_groth16_prove_kinds)."""
        host_key = isinstance(pk["s_pows"], HostWords)
        addr = lambda v: None if v is None else (v.ptr if isinstance(v, HostWords) else v.data_ptr())
        assert all(isinstance(pk[k], HostWords) == host_key for k in ("s_pows", "delta_ts") + (("delta_mid",) if pk.get("delta_mid") is not None else ()))
        s = _lib.Groth16PK(addr(pk["s_pows"]), addr(pk["delta_ts"]), addr(pk.get("delta_mid")),
                           pk["alpha"].data_ptr(), pk["beta"].data_ptr(), window, 1 if host_key else 0)
        proof = self.enc_empty(3)
        empty = (C.c_int * 3)()
        keep, kp = self._kinds(kinds, dcs.n_vars)
        _lib.check(self.lib.rs_groth16_prove_kinds(self.h, dcs.h, C.byref(s), _ptr(assignment), kp, _ptr(proof),
                                                   empty if want_empty else None, self.stream()))
        return proof, [int(e) for e in empty]

    def rinocchio_prove(self, dcs, pk, assignment, d1=None, d2=None, d3=None, window=0, kinds=None):
        host_key = isinstance(pk.get("s_pows"), HostWords)
        g = lambda k: None if pk.get(k) is None else (pk[k].ptr if isinstance(pk[k], HostWords) else pk[k].data_ptr())
        s = _lib.RinocchioPK(g("s_pows"), g("alpha_s_pows"), g("beta_prods"), g("beta_rv_ts"), g("beta_rw_ts"), g("beta_ry_ts"),
                             window, 1 if host_key else 0)
        proof = self.enc_empty(9)
        empty = (C.c_int * 9)()
        keep, kp = self._kinds(kinds, dcs.n_vars)
        _lib.check(self.lib.rs_rinocchio_prove_kinds(self.h, dcs.h, C.byref(s), _ptr(assignment), kp, _ptr(d1), _ptr(d2), _ptr(d3),
                                                     _ptr(proof), empty, self.stream()))
        return proof, [int(e) for e in empty]

    # ---- measurement / synthetic workloads
    def set_profiling(self, on):
        _lib.check(self.lib.rs_set_profiling(self.h, 1 if on else 0))

    def last_timings(self):
        t = _lib.Timings()
        _lib.check(self.lib.rs_last_timings(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in _lib.Timings._fields_}

    def measure_peaks(self):
        """{hbm_copy_gbs, hbm_read_gbs, hbm_inplace_gbs, fp64_fma_T, fp64_mulmod_G, int_montmul_G} measured on this device now (rs_measure_peaks)"""
        p = _lib.Peaks()
        _lib.check(self.lib.rs_measure_peaks(self.h, C.byref(p), self.stream()))
        return {k: getattr(p, k) for k, _ in _lib.Peaks._fields_}

    def profile_read(self):
        """[{name, launches, total_ms, alg_bytes, fp64_ops}] per kernel since profiling was switched on, by time."""
        cap = 64
        arr = (_lib.KernelStat * cap)()
        n = C.c_int(0)
        _lib.check(self.lib.rs_profile_read(self.h, arr, cap, C.byref(n)))
        return [{"name": arr[k].name.decode(), "launches": arr[k].launches, "total_ms": arr[k].total_ms,
                 "alg_bytes": arr[k].alg_bytes, "fp64_ops": arr[k].fp64_ops} for k in range(min(cap, n.value))]

    def fill_uniform(self, t, layout, seed):
        words = self.ring_words if layout == 0 else self.enc_words
        _lib.check(self.lib.rs_fill_uniform(self.h, _ptr(t), self._count(t, words), layout, seed, self.stream()))
        return t

    def chain_assignment(self, assignment, m):
        _lib.check(self.lib.rs_chain_assignment(self.h, _ptr(assignment), m, self.stream()))
        return assignment
