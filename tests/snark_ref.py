"""ringGroth16 generator and verifier restated for the tests (TEST INFRASTRUCTURE, CPU, toy sizes).

Follows ringsnark/zk_proof_systems/groth16/groth16.tcc:5-66 (generator) and :117-170 (verifier),
ringsnark/reductions/r1cs_to_qrp/r1cs_to_qrp.tcc:76-116 (instance map with evaluation) and
ringsnark/util/evaluation_domain.tcc:21-50 (Lagrange polynomials / vanishing polynomial at a point),
on top of the CPU oracle's ring arithmetic.  The prover between them is the device's.  The point
of the exercise: a proof produced by the HIP path must satisfy the reference's verification
equation  A*B == alpha*beta + gamma*f_io + delta*C  under a real (encrypted) proving key.
"""
import numpy as np

from oracle import oracle as O
from tests import helpers as H


class Ring:
    """Ring elements as [L][N] uint64 arrays with the oracle's slot-wise operations."""

    def __init__(self, ctx):
        self.ctx = ctx

    def scalar(self, v):
        return self.ctx.ring_scalar(int(v))

    def add(self, a, b):
        return self.ctx.ring_add(a, b)

    def sub(self, a, b):
        return self.ctx.ring_sub(a, b)

    def mul(self, a, b):
        return self.ctx.ring_mul(a, b)

    def inv(self, a):
        d, ok = self.ctx.ring_inv(a)
        assert ok, "element is not invertible in ring"
        return d

    def random_invertible(self, rng):
        out = np.empty(self.ctx.ring_shape(), dtype=np.uint64)
        for i, q in enumerate(self.ctx.q):
            out[i] = rng.randint(1, min(q, 2**62), size=self.ctx.N, dtype=np.int64).astype(np.uint64) % np.uint64(q)
            out[i][out[i] == 0] = 1
        return out

    def random_exceptional(self, rng, m):
        """RingElem::random_exceptional_element(domain): s - i invertible for every node i < m."""
        out = np.empty(self.ctx.ring_shape(), dtype=np.uint64)
        for i, q in enumerate(self.ctx.q):
            out[i] = (rng.randint(0, 2**62, size=self.ctx.N, dtype=np.int64).astype(np.uint64) % np.uint64(q - m)) + np.uint64(m)
        return out


def lagrange_at(R, s, m):
    """evaluation_domain.tcc:21-41: u_j = prod_{i != j} (s - i) / (j - i)."""
    u = []
    for j in range(m):
        num, den = R.scalar(1), R.scalar(1)
        for i in range(m):
            if i != j:
                num = R.mul(num, R.sub(s, R.scalar(i)))
                den = R.mul(den, R.sub(R.scalar(j), R.scalar(i)))
        u.append(R.mul(num, R.inv(den)))
    return u


def instance_map_with_evaluation(R, cs, s):
    """r1cs_to_qrp.tcc:76-116: At/Bt/Ct[k] = A_k(s) for variables k = 0..n_vars, Ht = s^i, Zt = Z(s)."""
    m = cs.m
    u = lagrange_at(R, s, m)
    out = {}
    for name in "abc":
        rp, col, cf = cs.mats[name]
        pidx = cs.poly_idx[name] if cs.poly_idx is not None else None
        vals = [R.scalar(0) for _ in range(cs.n_vars + 1)]
        for i in range(m):
            for e in range(rp[i], rp[i + 1]):
                if pidx is not None and pidx[e] >= 0:  # a coefficient that is a general ring element
                    coeff = cs.poly_table[pidx[e]]
                else:
                    coeff = np.stack([np.full(R.ctx.N, cf[l, e], dtype=np.uint64) for l in range(R.ctx.L)])
                vals[col[e]] = R.add(vals[col[e]], R.mul(u[i], coeff))
        out[name] = vals
    Ht = [R.scalar(1)]
    for _ in range(m):
        Ht.append(R.mul(Ht[-1], s))
    Zt = R.sub(s, R.scalar(0))
    for i in range(1, m):
        Zt = R.mul(Zt, R.sub(s, R.scalar(i)))
    return out["a"], out["b"], out["c"], Ht, Zt


def groth16_generator(ctx, cs, seed, encode, imap=None):
    """groth16.tcc:5-66.  `encode(sk, rings[count][L][N], seed) -> encodings`; `imap(s)` (optional)
    replaces the CPU instance map with evaluation; returns (pk, vk)."""
    R = Ring(ctx)
    rng = np.random.RandomState(seed)
    s = R.random_exceptional(rng, cs.m)
    At, Bt, Ct, Ht, Zt = imap(s) if imap else instance_map_with_evaluation(R, cs, s)
    sk = ctx.keygen(seed + 1)
    alpha, beta, gamma, delta = (R.random_invertible(rng) for _ in range(4))
    delta_inv = R.inv(delta)
    s_pows = Ht[: cs.m + 1]
    delta_ts = [R.mul(R.mul(x, Zt), delta_inv) for x in s_pows]
    delta_mid = []
    for i in range(cs.n_aux):
        idx = i + cs.n_inputs + 1
        t = R.add(R.add(R.mul(beta, At[idx]), R.mul(alpha, Bt[idx])), Ct[idx])
        delta_mid.append(R.mul(t, delta_inv))
    pk = {
        "alpha": encode(sk, np.stack([alpha]), seed + 10)[0],
        "beta": encode(sk, np.stack([beta]), seed + 11)[0],
        "s_pows": encode(sk, np.stack(s_pows), seed + 12),
        "delta_mid": encode(sk, np.stack(delta_mid), seed + 13),
        "delta_ts": encode(sk, np.stack(delta_ts), seed + 14),
    }
    vk = {"s": s, "alpha": alpha, "beta": beta, "gamma": gamma, "delta": delta, "sk": sk}
    return pk, vk


def poly_eval_ring(R, coeffs, s):
    """util/polynomials.tcc:46-53 with a ring-element argument."""
    acc = R.scalar(0)
    for c in reversed(list(coeffs)):
        acc = R.add(R.mul(acc, s), c)
    return acc


def groth16_verifier(ctx, cs, vk, primary, A, B, C):
    """groth16.tcc:117-170 on DECODED proof elements A, B, C ([L][N] each)."""
    R = Ring(ctx)
    padded = np.zeros((cs.n_vars,) + ctx.ring_shape(), dtype=np.uint64)
    padded[: cs.n_inputs] = primary
    ocs = H.oracle_cs(cs)
    io_s = []
    for which in range(3):
        coeffs = np.empty((cs.m,) + ctx.ring_shape(), dtype=np.uint64)
        for limb, q in enumerate(ctx.q):
            ev = O.r1cs_evaluate(q, ocs, which, limb, np.ascontiguousarray(padded[:, limb, :]))
            coeffs[:, limb, :] = O.interpolate(q, ev)
        io_s.append(poly_eval_ring(R, coeffs, vk["s"]))
    f_io = R.add(R.add(R.mul(vk["beta"], io_s[0]), R.mul(vk["alpha"], io_s[1])), io_s[2])
    f_io = R.mul(f_io, R.inv(vk["gamma"]))
    rhs = R.add(R.add(R.mul(vk["alpha"], vk["beta"]), R.mul(vk["gamma"], f_io)), R.mul(vk["delta"], C))
    return bool((R.mul(A, B) == rhs).all())


def rinocchio_generator(ctx, cs, seed, encode):
    """rinocchio.tcc:5-72 (the key elements the prover reads).  Returns (pk, vk)."""
    R = Ring(ctx)
    rng = np.random.RandomState(seed)
    s = R.random_exceptional(rng, cs.m)
    At, Bt, Ct, Ht, Zt = instance_map_with_evaluation(R, cs, s)
    sk = ctx.keygen(seed + 1)
    alpha, r_v, r_w = (R.random_invertible(rng) for _ in range(3))
    r_y = R.mul(r_v, r_w)
    beta = R.random_invertible(rng)  # random_nonzero_element; invertible is a special case
    s_pows = Ht[: cs.m + 1]
    alpha_s_pows = [R.mul(x, alpha) for x in s_pows]
    linchecks = []
    for i in range(cs.n_aux):
        idx = i + cs.n_inputs + 1
        t = R.add(R.add(R.mul(r_v, At[idx]), R.mul(r_w, Bt[idx])), R.mul(r_y, Ct[idx]))
        linchecks.append(R.mul(t, beta))
    beta_Zt = R.mul(beta, Zt)
    pk = {
        "s_pows": encode(sk, np.stack(s_pows), seed + 10),
        "alpha_s_pows": encode(sk, np.stack(alpha_s_pows), seed + 11),
        "beta_prods": encode(sk, np.stack(linchecks), seed + 12),
        "beta_rv_ts": encode(sk, np.stack([R.mul(beta_Zt, r_v)]), seed + 13)[0],
        "beta_rw_ts": encode(sk, np.stack([R.mul(beta_Zt, r_w)]), seed + 14)[0],
        "beta_ry_ts": encode(sk, np.stack([R.mul(beta_Zt, r_y)]), seed + 15)[0],
    }
    vk = {"s": s, "alpha": alpha, "beta": beta, "r_v": r_v, "r_w": r_w, "r_y": r_y, "sk": sk, "Zt": Zt}
    return pk, vk


def rinocchio_verifier(ctx, cs, vk, primary, decs):
    """rinocchio.tcc:192-300 on the nine DECODED proof elements
    (V_mid, V_mid', W_mid, W_mid', Y_mid, Y_mid', H, H', L_beta)."""
    R = Ring(ctx)
    V, Vp, W, Wp, Y, Yp, Hh, Hp, Lb = decs
    L = R.mul(R.add(R.add(R.mul(V, vk["r_v"]), R.mul(W, vk["r_w"])), R.mul(Y, vk["r_y"])), vk["beta"])
    padded = np.zeros((cs.n_vars,) + ctx.ring_shape(), dtype=np.uint64)
    padded[: cs.n_inputs] = primary
    ocs = H.oracle_cs(cs)
    io_s = []
    for which in range(3):
        coeffs = np.empty((cs.m,) + ctx.ring_shape(), dtype=np.uint64)
        for limb, q in enumerate(ctx.q):
            ev = O.r1cs_evaluate(q, ocs, which, limb, np.ascontiguousarray(padded[:, limb, :]))
            coeffs[:, limb, :] = O.interpolate(q, ev)
        io_s.append(poly_eval_ring(R, coeffs, vk["s"]))
    Pv = R.sub(R.mul(R.add(V, io_s[0]), R.add(W, io_s[1])), R.add(Y, io_s[2]))
    checks = {
        "V' = alpha V": (Vp == R.mul(V, vk["alpha"])).all(),
        "W' = alpha W": (Wp == R.mul(W, vk["alpha"])).all(),
        "Y' = alpha Y": (Yp == R.mul(Y, vk["alpha"])).all(),
        "H' = alpha H": (Hp == R.mul(Hh, vk["alpha"])).all(),
        "L_beta = L": (Lb == L).all(),
        "P = H Z(s)": (Pv == R.mul(Hh, vk["Zt"])).all(),
    }
    return all(bool(v) for v in checks.values()), checks
