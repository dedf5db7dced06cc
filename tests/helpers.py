"""Shared builders for the parity tests (CPU oracle side)."""
import numpy as np

from oracle import oracle as O
from ringsnark_amd import params as P
from ringsnark_amd import r1cs as R


def oracle_ctx(prm):
    return O.Ctx(prm.N, prm.q, prm.N_enc, prm.Q)


def oracle_cs(cs):
    return O.R1CSHandle(cs.m, cs.n_vars, cs.n_inputs, cs.mats)


def lincomb_oracle(ctx, cs):
    def f(name, i, asg):
        rp, col, cf = cs.mats[name]
        acc = np.zeros(ctx.ring_shape(), dtype=np.uint64)
        for e in range(rp[i], rp[i + 1]):
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
