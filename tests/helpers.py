"""Shared builders for the parity tests (CPU oracle side)."""
import numpy as np

from oracle import oracle as O
from ringsnark_amd import params as P
from ringsnark_amd import r1cs as R


def oracle_ctx(prm):
    return O.Ctx(prm.N, prm.q, prm.N_enc, prm.Q)


def oracle_cs(cs):
    return O.R1CSHandle(cs.m, cs.n_vars, cs.n_inputs, cs.mats, cs.poly_idx, cs.poly_table)


def lincomb_oracle(ctx, cs):
    def f(name, i, asg):
        rp, col, cf = cs.mats[name]
        acc = np.zeros(ctx.ring_shape(), dtype=np.uint64)
        pidx = cs.poly_idx[name] if cs.poly_idx is not None else None
        for e in range(rp[i], rp[i + 1]):
            if pidx is not None and pidx[e] >= 0:
                coeff = cs.poly_table[pidx[e]]
            else:
                coeff = np.stack([np.full(ctx.N, cf[l, e], dtype=np.uint64) for l in range(ctx.L)])
            v = ctx.ring_scalar(1) if col[e] == 0 else asg[col[e] - 1]
            acc = ctx.ring_add(acc, ctx.ring_mul(v, coeff))
        return acc
    return f


def make_assignment(ctx, cs, seed=7):
    x0, x1 = ctx.random_ring(seed), ctx.random_ring(seed + 1)
    asg = R.solve_forward(cs, x0, x1, ctx.ring_mul, lincomb_oracle(ctx, cs))
    return np.ascontiguousarray(np.stack(asg))


def limb_slices(ctx, vec):
    """[t][L][N] -> list over limbs of contiguous [t][N]."""
    return [np.ascontiguousarray(vec[:, i, :]) for i in range(ctx.L)]


def dft_circuit(prm, ctx, seed=5):
    """The statement of the reference's benchmarks/bench_ntt_SEAL.cpp:28-83 on the ring of `prm`: the one-constraint
    DFT circuit (ringsnark_amd.r1cs.dft_r1cs) with root_pows = powers of the minimal primitive 2N-th root of unity of
    the FIRST ring prime (:40-47; for the other limbs the same integers are reduced mod q_i -- the reference hands a
    vector of N words to a constructor that adopts L*N, so what it holds there is not defined), and a satisfying
    assignment: x_1..x_N Scalars (the reference's plaintext coefficients, :70-75), x_{N+1} = the evaluated sum (:77-78).
    Returns (cs, assignment [N+1][L][N])."""
    N, q0 = prm.N, int(prm.q[0])
    root = O.minimal_primitive_root(2 * N, q0)
    pw = [1]
    for _ in range(N - 1):
        pw.append(pw[-1] * root % q0)
    root_pows = np.array([[v % int(p) for v in pw] for p in prm.q], dtype=np.uint64)
    cs = R.dft_r1cs(prm.q, N, root_pows)
    rng = np.random.RandomState(seed)
    xs = [ctx.ring_scalar(int(v)) for v in rng.randint(0, 2**16, N)]
    xs.append(lincomb_oracle(ctx, cs)("a", 0, xs))
    return cs, np.ascontiguousarray(np.stack(xs))
