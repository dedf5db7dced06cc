"""Checks of a device proof at sizes where the oracle's O(m^2) witness map cannot be run in full
(test infrastructure; used by the -m gpu tests at configuration scale and by bench.py's untimed post-run check).

Two parts, both against the CPU oracle / exact integer arithmetic:
  1. the device witness map on sampled (limb, slot) columns, through polynomial identities at random points
     (O(m) integer operations per column and point);
  2. full (limb, component, prime) slabs of proof elements, recomputed by the oracle (OpenMP over terms) from the
     device's coefficient vectors and the proving key.
"""
import time

import numpy as np


def _column_identities(q, m, cs, limb, x, g, ds, rng, points=2, rinocchio=False):
    """One column.  x: assignment [n_vars]; g: dict of device coefficient vectors of this column; ds = (d1, d2, d3)
    slot values (0 without ZK).  Identities, at random r with L_j(r) = prod_{i != j}(r - i) / w_j,
    w_j = (-1)^(m-1-j) j! (m-1-j)!:
         X_io(r) + X_mid(r) = sum_j (x_j + const_j) L_j(r)   the reference counts index-0 terms in both passes
                                                              (r1cs_to_qrp.tcc:175-201)
         X_io(r)            = sum_j x^io_j L_j(r)
         H(r) Z(r)          = A(r) B(r) - C(r) + Z(r) (d2 A(r) + d1 B(r) - d3 + d1 d2 Z(r))   (r1cs_to_qrp.tcc:230-253)
    with A, B, C the interpolants of the full evaluations (the assignment satisfies the system)."""
    fact = [1] * m
    for j in range(1, m):
        fact[j] = fact[j - 1] * j % q
    inv_w = [pow(fact[j] * fact[m - 1 - j] % q, q - 2, q) for j in range(m)]

    def evals(name, mode):
        rp, col, cf = cs.mats[name]
        out = [0] * m
        for i in range(m):
            acc = 0
            for e in range(int(rp[i]), int(rp[i + 1])):
                c = int(col[e])
                if c == 0:
                    if mode != "vars":
                        acc += int(cf[limb, e])
                elif mode == "const":
                    continue
                elif mode in ("full", "vars") or (c - 1) < cs.n_inputs:
                    acc += int(cf[limb, e]) * x[c - 1]
            out[i] = acc % q
        return out

    y = {n: evals(n, "full") for n in "abc"}
    y_io = {n: evals(n, "io") for n in "abc"}
    y_const = {n: evals(n, "const") for n in "abc"}
    d1, d2, d3 = ds
    for _ in range(points):
        r = (int(rng.randint(m, 2**31)) * 65537 + 12345) % q
        if r < m:
            r += m
        pre = [1] * (m + 1)
        for j in range(m):
            pre[j + 1] = pre[j] * (r - j) % q
        suf = [1] * (m + 1)
        for j in range(m - 1, -1, -1):
            suf[j] = suf[j + 1] * (r - j) % q
        Zr = pre[m]

        def lagr(yv):
            acc = 0
            for j in range(m):
                t = yv[j] * pre[j] % q * suf[j + 1] % q * inv_w[j]
                acc += -t if (m - 1 - j) & 1 else t
            return acc % q

        def horner(coeffs):
            acc = 0
            for c in reversed(coeffs):
                acc = (acc * r + int(c)) % q
            return acc

        full = {n: lagr(y[n]) for n in "abc"}
        for n, N_ in (("a", "A"), ("b", "B"), ("c", "C")):
            io_r = lagr(y_io[n])
            if N_ + "_io" in g and horner(g[N_ + "_io"]) != io_r:
                return "%s_io interpolant mismatch" % N_
            if N_ + "_mid" in g and horner(g[N_ + "_mid"]) != (full[n] + lagr(y_const[n]) - io_r) % q:
                return "%s_mid interpolant mismatch" % N_
        if "H" in g:
            A, B, C = full["a"], full["b"], full["c"]
            rhs = (A * B - C + Zr * (d2 * A + d1 * B - d3 + d1 * d2 * Zr)) % q
            if horner(g["H"]) * Zr % q != rhs:
                return "H(r) Z(r) identity fails"
    return None


def check_columns(prm, cs, asg, w, cols, rng, ds=(None, None, None)):
    """asg: device assignment [n_vars][L][N]; w: dict of device coefficient vectors [rows][L][N]."""
    from ringsnark_amd.device import to_host
    for (limb, slot) in cols:
        x = [int(v) for v in to_host(asg[:, limb, slot].contiguous())]
        g = {k: to_host(v[:, limb, slot].contiguous()) for k, v in w.items() if k != "Z" and v is not None}
        dv = tuple(0 if d is None else int(to_host(d[limb, slot].contiguous().reshape(1))[0]) for d in ds)
        err = _column_identities(int(prm.q[limb]), cs.m, cs, limb, x, g, dv, rng)
        if err:
            return "%s at limb %d slot %d" % (err, limb, slot)
    return None


def check_all_columns(prm, cs, asg, w, ds=(None, None, None), limbs=None, slot0=0, nslots=None, seed=77, points=2, Z=None):
    """EVERY (limb, slot) column of a device witness map through the identities that define its outputs
    (oracle/rs_identities.c: C + OpenMP, O(m + nnz) per column and point; points shared by the slots of a limb).
    asg: device assignment [n_vars][L][N]; w: dict of device vectors [rows][L][S] (S = N, or a slot range
    [slot0, slot0 + nslots) of every limb: the compact outputs of rs_witness_map_slots); ds: device ring elements [L][N].
    limbs: the limb indices of `prm` to check (default all).  Returns (error string or None, info dict)."""
    from oracle import oracle as O
    from ringsnark_amd.device import to_host
    from tests import helpers as H

    t0 = time.perf_counter()
    ocs = H.oracle_cs(cs)
    nslots = prm.N - slot0 if nslots is None else nslots
    rng = np.random.RandomState(seed)
    names = [k for k in O.IDENTITY_NAMES if w.get(k) is not None]
    checked = 0
    t_copy = 0.0
    for limb in (range(prm.L) if limbs is None else limbs):
        q = int(prm.q[limb])
        pts = [cs.m + (int(rng.randint(1, 2**31)) * 65537 + 12345) % (q - cs.m) for _ in range(points)]
        tc = time.perf_counter()
        blocked = nslots % 32 == 0
        if blocked:  # [rows][slots] -> [slots/32][rows][32] on the device: a block of 32 columns then streams through host memory
            blk = lambda t: to_host(t.reshape(t.shape[0], nslots // 32, 32).permute(1, 0, 2).contiguous())
        else:
            blk = lambda t: to_host(t.contiguous())
        a = blk(asg[:, limb, slot0:slot0 + nslots])
        for k in names:
            assert w[k].shape[2] == nslots, (k, w[k].shape, nslots)
        vec = {k: blk(w[k][:, limb, :]) for k in names}
        dv = [None if d is None else to_host(d[limb, slot0:slot0 + nslots].contiguous()) for d in ds]
        t_copy += time.perf_counter() - tc
        n_bad, bad = O.witness_identities(q, ocs.at_slots(slot0), limb, a, vec, pts, *dv, Z=None if Z is None else Z[limb], threads=0,
                                          blocked=blocked)
        if n_bad:
            s = int(np.flatnonzero(bad)[0])
            failed = [O.IDENTITY_NAMES[b] for b in range(7) if bad[s] >> b & 1]
            return ("%d of %d columns of limb %d fail; first: slot %d, identities %s" % (n_bad, nslots, limb, slot0 + s, failed),
                    {"columns_failed": n_bad})
        checked += nslots
        del a, vec
    return None, {"columns": checked, "points_per_column": points, "vectors": names,
                  "seconds": round(time.perf_counter() - t0, 1), "copy_seconds": round(t_copy, 1)}


def slab_inner_product(octx, acc, key_slab, vec, limb, j, T, kinds=None, step=4096):
    """acc += the (limb, ., j) slab of <key, vec[:T]> by the CPU oracle; key_slab [W][N_enc] (W < T: tiled key)."""
    from ringsnark_amd.device import to_host
    assert kinds is None
    for t0 in range(0, T, step):
        rows = to_host(vec[t0:min(T, t0 + step), limb, :].contiguous())
        octx.inner_product_slab(limb, j, key_slab, rows, acc, t0=t0, window=key_slab.shape[0], threads=0)


def key_slab(t, l, c, j, n_enc):
    from ringsnark_amd.device import to_host
    return to_host(t[..., l, c, j, :].contiguous()).reshape(-1, n_enc)


def limb_inner_product(fctx, acc, key_limb, vec, limb, T, step=4096):
    """acc [2][K][N_enc] += ring limb `limb` of <key, vec[:T]>, all its (component, prime) slabs at once, by the CPU oracle in
    its SEAL-arithmetic build (oracle/fastcpu.py inner_product_limb); key_limb [W][2][K][N_enc] on the host (W < T: tiled key)."""
    from ringsnark_amd.device import to_host
    for t0 in range(0, T, step):
        rows = to_host(vec[t0:min(T, t0 + step), limb, :].contiguous())
        fctx.inner_product_limb(limb, key_limb, rows, acc, t0=t0, window=key_limb.shape[0], threads=0)


def groth16_all_slabs(prm, cs, asg, pk_host, proof, w, m):
    """EVERY (element, limb, component, prime) slab of a ringGroth16 proof (3 L 2 K of them: 96 at the headline) recomputed from
    the coefficient vectors w (device) and the key (pk_host: name -> [W][L][2][K][N_enc] host array, or a callable limb -> the
    limb's [W][2][K][N_enc] slice) -- groth16.tcc:89-112 term for term, one ring limb at a time so that the plaintext of a term
    is transformed once for the limb's 2 K slabs.  Returns (error or None, seconds)."""
    from oracle import fastcpu as F
    from ringsnark_amd.device import to_host
    t0 = time.perf_counter()
    fctx = F.FastCtx(prm.N, prm.q, prm.N_enc, prm.Q)
    Qv = np.array(prm.Q, dtype=np.uint64).reshape(1, prm.K, 1)

    def limb_of(name, l):
        v = pk_host[name]
        return v(l) if callable(v) else np.ascontiguousarray(v[:, l] if v.ndim == 5 else v[l])
    for l in range(prm.L):
        ks = limb_of("s_pows", l)
        for e, elem in enumerate(("A", "B")):  # groth16.tcc:89-95, 97-103
            acc = np.zeros((2, prm.K, prm.N_enc), dtype=np.uint64)
            limb_inner_product(fctx, acc, ks, w[elem + "_io"], l, m)
            limb_inner_product(fctx, acc, ks, w[elem + "_mid"], l, m)
            acc = (acc + limb_of("alpha" if elem == "A" else "beta", l)) % Qv
            if not (acc == to_host(proof[e, l].contiguous())).all():
                bad = np.argwhere((acc != to_host(proof[e, l].contiguous())).any(axis=2))[0]
                return "proof element %s slab (limb %d, component %d, prime %d) differs from the CPU oracle" % (elem, l, bad[0], bad[1]), time.perf_counter() - t0
        del ks
        acc = np.zeros((2, prm.K, prm.N_enc), dtype=np.uint64)  # groth16.tcc:105-112
        limb_inner_product(fctx, acc, limb_of("delta_ts", l), w["H"], l, m + 1)
        limb_inner_product(fctx, acc, limb_of("delta_mid", l), asg[cs.n_inputs:], l, cs.n_aux)
        if not (acc == to_host(proof[2, l].contiguous())).all():
            bad = np.argwhere((acc != to_host(proof[2, l].contiguous())).any(axis=2))[0]
            return "proof element C slab (limb %d, component %d, prime %d) differs from the CPU oracle" % (l, bad[0], bad[1]), time.perf_counter() - t0
    return None, time.perf_counter() - t0


def groth16_check(dev, prm, cs, dcs, asg, pk, proof, m, W=None, seed=5, n_cols=None, n_slabs=3, all_slabs=False):
    """ringGroth16 proof (groth16.tcc:70-115) computed by the device for (cs, asg, pk): (1) witness-map identities
    on EVERY column (n_cols = None; round 4 sampled n_cols random columns + the two corners, kept for callers that ask
    for it); (2) n_slabs full (limb, component, prime) slabs -- A, C, B, then A, C, B again on other coordinates, ... --
    recomputed by the CPU oracle from the (now fully checked) coefficient vectors and the key.  pk: dict of device
    tensors; its entries are RELEASED (the caller must hold no other reference when memory is tight).
    Returns (ok, info)."""
    import torch

    from ringsnark_amd.device import to_host
    from tests import helpers as H

    t_start = time.perf_counter()
    octx = H.oracle_ctx(prm)
    rng = np.random.RandomState(seed)
    slabs = [("A", int(rng.randint(prm.L)), int(rng.randint(2)), int(rng.randint(prm.K))),
             ("C", int(rng.randint(prm.L)), int(rng.randint(2)), int(rng.randint(prm.K)))]
    slabs.append(("B", int(rng.randint(prm.L)), int(rng.randint(2)), int(rng.randint(prm.K))))  # drawn last: A and C keep their round-3 slabs
    while len(slabs) < n_slabs:  # further slabs walk the limbs, components and primes
        k = len(slabs)
        cand = ("ACB"[k % 3], (slabs[k - 3][1] + 1) % prm.L, (slabs[k - 3][2] + 1) % 2, (slabs[k - 3][3] + 1 + k // 6) % prm.K)
        slabs.append(cand)
    slabs = slabs[:n_slabs]
    key = {}
    for elem, l, c, j in slabs:
        for nme in {"A": ("s_pows", "alpha"), "B": ("s_pows", "beta"), "C": ("delta_ts", "delta_mid")}[elem]:
            key[(nme, l, c, j)] = key_slab(pk[nme], l, c, j, prm.N_enc)
    # all_slabs: the whole key on the host, one limb slice at a time when asked for ([W][2][K][N_enc]: W 2 K N_enc 8 bytes each)
    pk_host = {k: to_host(v.contiguous()) for k, v in pk.items()} if all_slabs else None
    proof_h = {(e, l, c, j): to_host(proof[{"A": 0, "B": 1, "C": 2}[e], l, c, j].contiguous()) for e, l, c, j in slabs}
    for k in list(pk.keys()):
        del pk[k]
    torch.cuda.empty_cache()
    # the prover's own witness map, re-run through the same chunking (deterministic: identical vectors)
    w = dev.witness_map(dcs, asg, want=("A_io", "A_mid", "B_io", "B_mid", "H"))
    torch.cuda.synchronize()
    wv = {k: w[k] for k in ("A_io", "A_mid", "B_io", "B_mid", "H")}
    if n_cols is None:
        err, col_info = check_all_columns(prm, cs, asg, wv, seed=seed + 1)
        n_checked = prm.L * prm.N
    else:
        cols = [(int(rng.randint(prm.L)), int(rng.randint(prm.N))) for _ in range(n_cols)] + [(0, 0), (prm.L - 1, prm.N - 1)]
        err, col_info, n_checked = check_columns(prm, cs, asg, wv, cols, rng), {}, len(cols)
    if err:
        return False, {"error": err}
    t_cols = time.perf_counter() - t_start
    for elem, l, c, j in slabs:
        acc = np.zeros(prm.N_enc, dtype=np.uint64)
        if elem in ("A", "B"):  # groth16.tcc:89-95, 97-103
            slab_inner_product(octx, acc, key[("s_pows", l, c, j)], w[elem + "_io"], l, j, m)
            slab_inner_product(octx, acc, key[("s_pows", l, c, j)], w[elem + "_mid"], l, j, m)
            acc = (acc + key[("alpha" if elem == "A" else "beta", l, c, j)][0]) % np.uint64(prm.Q[j])
        else:  # groth16.tcc:105-112
            slab_inner_product(octx, acc, key[("delta_ts", l, c, j)], w["H"], l, j, m + 1)
            slab_inner_product(octx, acc, key[("delta_mid", l, c, j)], asg[cs.n_inputs:], l, j, cs.n_aux)
        if not (acc == proof_h[(elem, l, c, j)]).all():
            return False, {"error": "proof element %s slab (limb %d, component %d, prime %d) differs from the CPU oracle" % (elem, l, c, j)}
    all_cols = n_checked == prm.L * prm.N
    n_total = 3 * prm.L * 2 * prm.K
    slabs_all = None
    if all_slabs:  # the complete inner-product check: every slab, SEAL-arithmetic build of the oracle (bit-identical to the `%` checker: tests/test_oracle.py)
        err, secs = groth16_all_slabs(prm, cs, asg, pk_host, proof, w, m)
        if err:
            return False, {"error": err}
        slabs_all = {"slabs": "%d of %d" % (n_total, n_total), "seconds": round(secs, 1),
                     "how": "oracle/rs_fastcpu.c rsf_inner_product_limb (Harvey / Barrett arithmetic), one ring limb at a time; the %d slabs above "
                            "also by the %%-based checker" % len(slabs)}
    return True, {"kind": ("every witness-map column (identities at 2 random points per limb, C + OpenMP); " if all_cols else
                           "SAMPLED columns; ") + "%d of %d proof slabs recomputed by the CPU oracle from the checked vectors" % (
                               n_total if all_slabs else len(slabs), n_total),
                  "sample": "%s of %d witness-map columns x 2 random points; %d of %d (element, limb, component, prime) slabs of the proof" % (
                      "all" if all_cols else n_checked, prm.L * prm.N, n_total if all_slabs else len(slabs), n_total),
                  "columns": "all" if all_cols else n_checked, "columns_checked": n_checked, "points_per_column": 2,
                  "columns_detail": col_info,
                  "slabs": ["%s[limb %d][comp %d][prime %d]" % s for s in slabs],
                  "slabs_checked": n_total if all_slabs else len(slabs), "slabs_total": n_total, "all_slabs": slabs_all,
                  "seconds": round(time.perf_counter() - t_start, 1), "columns_seconds": round(t_cols, 1)}


def rinocchio_check(dev, prm, cs, dcs, asg, pk, proof, m, ds=(None, None, None), seed=6, n_cols=None):
    """Rinocchio proof {A,A',B,B',C,C',D,D',F} (rinocchio.tcc:75-190): witness-map identities on EVERY column
    (n_cols = None; an integer samples that many, as round 4 did), with the ZK patch when ds are given, Z included; and,
    without ZK shifts on them, the slabs D' = <alpha_s_pows, H> and F = <beta_prods, aux> (non-ZK) against the CPU oracle."""
    import torch

    from ringsnark_amd.device import to_host
    from tests import helpers as H

    octx = H.oracle_ctx(prm)
    rng = np.random.RandomState(seed)
    zk = ds[0] is not None
    w = dev.witness_map(dcs, asg, *ds, want=("A_mid", "B_mid", "C_mid", "H"))
    torch.cuda.synchronize()
    wv = {k: w[k] for k in ("A_mid", "B_mid", "C_mid", "H")}
    if n_cols is None:
        err, col_info = check_all_columns(prm, cs, asg, wv, ds, seed=seed + 1, Z=w["Z"])
        n_checked = prm.L * prm.N
    else:
        cols = [(int(rng.randint(prm.L)), int(rng.randint(prm.N))) for _ in range(n_cols)] + [(prm.L - 1, prm.N - 1)]
        err, col_info, n_checked = check_columns(prm, cs, asg, wv, cols, rng, ds), {}, len(cols)
    if err:
        return False, {"error": err}
    # D' (index 7) carries no ZK shift (rinocchio.tcc:167-174 shifts A..C' only); F (index 8) only without ZK
    checks = [(7, "alpha_s_pows", "H", m + 1)]
    if not zk and cs.n_aux:
        checks.append((8, "beta_prods", "aux", cs.n_aux))
    done = []
    for idx, kname, vname, T in checks:
        l, c, j = int(rng.randint(prm.L)), int(rng.randint(2)), int(rng.randint(prm.K))
        acc = np.zeros(prm.N_enc, dtype=np.uint64)
        vec = asg[cs.n_inputs:] if vname == "aux" else w[vname]
        slab_inner_product(octx, acc, key_slab(pk[kname], l, c, j, prm.N_enc), vec, l, j, T)
        if not (acc == to_host(proof[idx, l, c, j].contiguous())).all():
            return False, {"error": "proof element %d slab (limb %d, component %d, prime %d) differs from the CPU oracle" % (idx, l, c, j)}
        done.append("elem%d[limb %d][comp %d][prime %d]" % (idx, l, c, j))
    all_cols = n_checked == prm.L * prm.N
    return True, {"kind": ("every witness-map column; " if all_cols else "SAMPLED columns; ") + "sampled proof slabs",
                  "sample": "%s of %d witness-map columns; %d of %d (element, limb, component, prime) slabs of the proof" % (
                      "all" if all_cols else n_checked, prm.L * prm.N, len(done), 9 * prm.L * 2 * prm.K),
                  "columns": "all" if all_cols else n_checked, "columns_detail": col_info, "slabs": done}
