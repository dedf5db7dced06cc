"""Host-side R1CS container in CSR form + the synthetic circuits of SURVEY.md section 8(d).

Mirrors what the hot path reads from the reference's r1cs_constraint_system<RingT>
(relations/constraint_satisfaction_problems/r1cs/r1cs.hpp:118-123): for every constraint i the
three linear combinations a, b, c as lists of (index, coeff) with index 0 = the constant one
(relations/variable.tcc:246-254).  Coefficients are slot-constant ring scalars, stored reduced
per RNS limb as uint64[L][nnz] (a signed literal c < 0 is -c's negation mod q_i, i.e.
-RingT::one()*|c|).
"""
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np


@dataclass
class R1CS:
    m: int  # constraints
    n_vars: int  # variables, excluding the constant one
    n_inputs: int  # primary inputs (the first n_inputs variables)
    mats: Dict[str, Tuple[np.ndarray, np.ndarray, np.ndarray]]  # a/b/c -> (row_ptr, col, coeff[L][nnz])

    @property
    def n_aux(self):
        return self.n_vars - self.n_inputs

    def nnz(self, name):
        return int(self.mats[name][1].shape[0])


def from_rows(m, n_vars, n_inputs, rows: Dict[str, List[List[Tuple[int, int]]]], q: List[int]) -> R1CS:
    """rows[name][i] = [(index, signed_int_coeff), ...]."""
    mats = {}
    for name in "abc":
        rp = np.zeros(m + 1, dtype=np.uint32)
        col, cf = [], []
        for i, terms in enumerate(rows[name]):
            for idx, c in terms:
                assert 0 <= idx <= n_vars
                col.append(idx)
                cf.append(c)
            rp[i + 1] = len(col)
        coeff = np.array([[c % p for c in cf] for p in q], dtype=np.uint64).reshape(len(q), len(cf))
        mats[name] = (rp, np.array(col, dtype=np.uint32), coeff)
    return R1CS(m, n_vars, n_inputs, mats)


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


def solve_forward(cs: R1CS, x0, x1, ring_mul, ring_lincomb):
    """Fill the assignment of chain/wide circuits: x_{i+2} = <a_i, x> * x_{i+1}.

    ring_lincomb(terms, assignment_list) and ring_mul(a, b) are supplied by the caller (CPU
    oracle in tests, device ring ops in the bench)."""
    asg = [x0, x1]
    rp, col, _ = cs.mats["a"]
    for i in range(cs.m):
        a_val = ring_lincomb("a", i, asg)
        asg.append(ring_mul(a_val, asg[i + 1]))
    return asg
