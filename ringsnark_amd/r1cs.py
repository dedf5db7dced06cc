"""Host-side R1CS container in CSR form + the synthetic circuits of SURVEY.md section 8(d).

Mirrors what the hot path reads from the reference's r1cs_constraint_system<RingT>
(relations/constraint_satisfaction_problems/r1cs/r1cs.hpp:118-123): for every constraint i the
three linear combinations a, b, c as lists of (index, coeff) with index 0 = the constant one
(relations/variable.tcc:246-254).  A coefficient is a RingT: either a slot-constant ring scalar, stored
reduced per RNS limb as uint64[L][nnz] (a signed literal c < 0 is -c's negation mod q_i, i.e.
-RingT::one()*|c|), or a general ring element (one residue per NTT slot: the DFT
constraint of benchmarks/bench_ntt_SEAL.cpp:46-53 multiplies variables by powers of a polynomial), kept
once in `poly_table` [n_poly][L][N] and referenced per non-zero by `poly_idx` (-1 = the scalar).
"""
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np


@dataclass
class R1CS:
    m: int  # constraints
    n_vars: int  # variables, excluding the constant one
    n_inputs: int  # primary inputs (the first n_inputs variables)
    mats: Dict[str, Tuple[np.ndarray, np.ndarray, np.ndarray]]  # a/b/c -> (row_ptr, col, coeff[L][nnz])
    poly_idx: Optional[Dict[str, np.ndarray]] = None  # a/b/c -> int32[nnz], -1 = scalar coefficient of mats
    poly_table: Optional[np.ndarray] = None  # uint64 [n_poly][L][N]: coefficients that are general ring elements

    @property
    def n_aux(self):
        return self.n_vars - self.n_inputs

    def nnz(self, name):
        return int(self.mats[name][1].shape[0])


def from_rows(m, n_vars, n_inputs, rows: Dict[str, List[List[Tuple[int, object]]]], q: List[int]) -> R1CS:
    """rows[name][i] = [(index, coeff), ...]; coeff: a signed int (slot-constant scalar) or a uint64 array [L][N]
    (a general ring element)."""
    mats, pidx, table = {}, {}, []
    for name in "abc":
        rp = np.zeros(m + 1, dtype=np.uint32)
        col, cf, pi = [], [], []
        for i, terms in enumerate(rows[name]):
            for idx, c in terms:
                assert 0 <= idx <= n_vars
                col.append(idx)
                if isinstance(c, np.ndarray):
                    assert c.ndim == 2 and c.shape[0] == len(q)
                    pi.append(len(table))
                    table.append(np.ascontiguousarray(c, dtype=np.uint64))
                    cf.append(0)
                else:
                    pi.append(-1)
                    cf.append(c)
            rp[i + 1] = len(col)
        coeff = np.array([[c % p for c in cf] for p in q], dtype=np.uint64).reshape(len(q), len(cf))
        mats[name] = (rp, np.array(col, dtype=np.uint32), coeff)
        pidx[name] = np.array(pi, dtype=np.int32)
    if not table:
        return R1CS(m, n_vars, n_inputs, mats)
    return R1CS(m, n_vars, n_inputs, mats, pidx, np.stack(table))


def chain_r1cs(m: int, q: List[int]) -> R1CS:
    """x_i * x_{i+1} = x_{i+2}, i < m; variables x_0..x_{m+1}; x_0, x_1 public (n_aux = m)."""
    rows = {"a": [[(i + 1, 1)] for i in range(m)], "b": [[(i + 2, 1)] for i in range(m)], "c": [[(i + 3, 1)] for i in range(m)]}
    return from_rows(m, m + 2, 2, rows, q)


def wide_r1cs(m: int, q: List[int], seed: int = 11, width: int = 8, n_inputs: int = 2) -> R1CS:
    """(sum of `width` earlier variables with small signed coefficients, plus a constant)
    * (x_{i+1}) = x_{i+2}.  Exercises multi-term linear combinations, the constant-one column
    and negative coefficients."""
    rng = np.random.RandomState(seed)
    rows = {"a": [], "b": [], "c": []}
    for i in range(m):
        terms = [(0, int(rng.randint(1, 5)))]
        for _ in range(width):
            terms.append((int(rng.randint(1, i + 3)), int(rng.randint(-3, 4)) or 1))
        rows["a"].append(terms)
        rows["b"].append([(i + 2, 1)])
        rows["c"].append([(i + 3, 1)])
    return from_rows(m, m + 2, n_inputs, rows, q)


def wide_poly_r1cs(m: int, q: List[int], N: int, seed: int = 17, width: int = 4, n_inputs: int = 2, aux_only: bool = False,
                   constants: bool = True) -> R1CS:
    """wide_r1cs with general ring elements among the coefficients: in `a` one polynomial coefficient per row on an
    arbitrary earlier variable (primary inputs included) and, on every third row, on the constant one; in `b` a
    polynomial coefficient on x_{i+1} on every other row:  <a_i, x> * (u_i x_{i+1}) = x_{i+2}.  Exercises every place
    a coefficient is read (evaluate in its three modes, the constant part of the mid vectors, the instance map).
    aux_only: polynomial coefficients multiply auxiliary variables only (the constant one and the primary inputs keep
    slot-constant scalars, so the device keeps its linear-form io vectors).  constants = False: no term on the constant
    one (the reference's witness map counts such terms in BOTH its io and its mid pass, r1cs_to_qrp.tcc:175-201, so a
    circuit with constants does not pass the reference's own verifier; end-to-end tests use circuits without them)."""
    rng = np.random.RandomState(seed)
    rand_ring = lambda: np.stack([rng.randint(0, p, N, dtype=np.int64).astype(np.uint64) for p in q])
    rows = {"a": [], "b": [], "c": []}
    for i in range(m):
        terms = [(0, rand_ring() if i % 3 == 0 and not aux_only else int(rng.randint(1, 5)))] if constants else []
        for k in range(width):
            idx = int(rng.randint(1, i + 3))
            poly = k == 0 and not (aux_only and idx <= n_inputs)
            terms.append((idx, rand_ring() if poly else (int(rng.randint(-3, 4)) or 1)))
        rows["a"].append(terms)
        rows["b"].append([(i + 2, rand_ring() if i % 2 == 0 and not (aux_only and i + 2 <= n_inputs) else 1)])
        rows["c"].append([(i + 3, 1)])
    return from_rows(m, m + 2, n_inputs, rows, q)


def dft_r1cs(q: List[int], N: int, root_pows: np.ndarray) -> R1CS:
    """The ONE-constraint circuit of the reference's benchmarks/bench_ntt_SEAL.cpp:28-55 (BASELINE.json configs[0]):
    N + 1 variables, all of them primary inputs (:28-29,37);  (x_1 + sum_{i=1}^{N-1} row^i * x_{i+1}) * 1 = x_{N+1}
    with row^i the i-th slot-wise power of the ring element `rs` whose residues are root_pows (:40-53).  root_pows:
    uint64 [L][N] (the reference fills it with the powers of the first prime's minimal 2N-th root of unity)."""
    rs = np.ascontiguousarray(root_pows, dtype=np.uint64)
    assert rs.shape == (len(q), N)
    qa = [int(p) for p in q]
    terms = [(1, 1)]
    row = rs.copy()
    for i in range(1, N):
        terms.append((i + 1, row.copy()))
        row = np.stack([(row[l].astype(object) * rs[l].astype(object) % qa[l]).astype(np.uint64) for l in range(len(q))])
    rows = {"a": [terms], "b": [[(0, 1)]], "c": [[(N + 1, 1)]]}
    return from_rows(1, N + 1, N + 1, rows, q)


def solve_forward(cs: R1CS, x0, x1, ring_mul, ring_lincomb):
    """Fill the assignment of chain/wide circuits: x_{i+2} = <a_i, x> * <b_i, x>  (b_i = x_{i+1}, possibly scaled).

    ring_lincomb(terms, assignment_list) and ring_mul(a, b) are supplied by the caller (CPU
    oracle in tests, device ring ops in the bench)."""
    asg = [x0, x1]
    rp, col, _ = cs.mats["a"]
    for i in range(cs.m):
        asg.append(ring_mul(ring_lincomb("a", i, asg), ring_lincomb("b", i, asg)))
    return asg


def logreg_r1cs(q: List[int], num_features: int = 256) -> R1CS:
    """The circuit of the reference's benchmarks/bench_logistic_regression_inference.cpp:72-125 (BASELINE.json
    configs[4]), variable for variable: per feature i two input ciphertexts in1[i], in2[i] of two ring elements each
    (allocated interleaved, :78-82), five outputs, the four products per feature, s02 and s11.  With 256 features:
    1031 constraints, 2055 variables, set_input_sizes(2*256+5 = 517) (:72,85).

      per feature (:94-107)   in1[i][0]*in2[i][0] = p00[i];  in1[i][0]*in2[i][1] = p01[i];
                              in1[i][1]*in2[i][0] = p10[i];  in1[i][1]*in2[i][1] = p11[i]
      sums (linear)           s0 = sum p00,  s1 = sum (p01 + p10),  s2 = sum p11     (the ciphertext product's components)
      degree-2 sigmoid (:116-125)  s0*s0 = out0;  (2 s0)*s1 = out1;  s0*s2 = s02;  s1*s1 = s11;
                              1 * (2 s02 + s11) = out2;  s1*s2 = out3;  s2*s2 = out4
    Variable k (1-based, 0 = the constant one): in1[i][j] = 4i+1+j, in2[i][j] = 4i+3+j, out[k] = 4F+1+k,
    p00[i] = 4F+6+i, p01[i] = 5F+6+i, p10[i] = 6F+6+i, p11[i] = 7F+6+i, s02 = 8F+6, s11 = 8F+7."""
    F = num_features
    in1 = lambda i, j: 4 * i + 1 + j
    in2 = lambda i, j: 4 * i + 3 + j
    out = lambda k: 4 * F + 1 + k
    p = lambda which, i: (4 + which) * F + 6 + i
    s02, s11 = 8 * F + 6, 8 * F + 7
    rows = {"a": [], "b": [], "c": []}

    def add(a, b, c):
        rows["a"].append(a)
        rows["b"].append(b)
        rows["c"].append(c)

    for i in range(F):
        add([(in1(i, 0), 1)], [(in2(i, 0), 1)], [(p(0, i), 1)])
        add([(in1(i, 0), 1)], [(in2(i, 1), 1)], [(p(1, i), 1)])
        add([(in1(i, 1), 1)], [(in2(i, 0), 1)], [(p(2, i), 1)])
        add([(in1(i, 1), 1)], [(in2(i, 1), 1)], [(p(3, i), 1)])
    s0 = [(p(0, i), 1) for i in range(F)]
    s1 = sorted([(p(1, i), 1) for i in range(F)] + [(p(2, i), 1) for i in range(F)])
    s2 = [(p(3, i), 1) for i in range(F)]
    add(s0, s0, [(out(0), 1)])
    add([(k, 2) for k, _ in s0], s1, [(out(1), 1)])
    add(s0, s2, [(s02, 1)])
    add(s1, s1, [(s11, 1)])
    add([(0, 1)], [(s02, 2), (s11, 1)], [(out(2), 1)])
    add(s1, s2, [(out(3), 1)])
    add(s2, s2, [(out(4), 1)])
    n_vars = 8 * F + 7
    return from_rows(4 * F + 7, n_vars, 2 * F + 5, rows, q)


def logreg_assignment(num_features, inputs, ring_mul, ring_add, ring_mul_scalar):
    """Full assignment [n_vars] of logreg_r1cs from the 4F input ring elements (in1[i][0], in1[i][1], in2[i][0],
    in2[i][1] per feature; bench_logistic_regression_inference.cpp:147-205).  The ring operations are supplied by
    the caller (CPU oracle in tests, batched device ops in the bench); they must accept leading batch dimensions."""
    F = num_features
    x = inputs.reshape((F, 4) + tuple(inputs.shape[1:]))
    a0, a1, b0, b1 = x[:, 0], x[:, 1], x[:, 2], x[:, 3]
    p00, p01, p10, p11 = ring_mul(a0, b0), ring_mul(a0, b1), ring_mul(a1, b0), ring_mul(a1, b1)

    def total(v):  # pairwise tree sum of F ring elements
        while v.shape[0] > 1:
            h = v.shape[0] // 2
            head = ring_add(v[:h], v[h:2 * h])
            v = head if v.shape[0] == 2 * h else _cat([head, v[2 * h:]])
        return v[0]

    s0, s1, s2 = total(p00), total(ring_add(p01, p10)), total(p11)
    s_02, s_11 = ring_mul(s0, s2), ring_mul(s1, s1)
    outs = [ring_mul(s0, s0), ring_mul(ring_mul_scalar(s0, 2), s1), ring_add(ring_mul_scalar(s_02, 2), s_11), ring_mul(s1, s2),
            ring_mul(s2, s2)]
    parts = [inputs] + [o[None] for o in outs] + [p00, p01, p10, p11, s_02[None], s_11[None]]
    return _cat(parts)


def _cat(parts):
    if isinstance(parts[0], np.ndarray):
        return np.ascontiguousarray(np.concatenate(parts))
    import torch
    return torch.cat(parts).contiguous()
