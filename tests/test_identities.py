"""The complete witness-map checker (oracle/rs_identities.c) against the oracle's own O(m^2) map (CPU).

The checker is what bench.py and the configuration-scale GPU tests use where the reference's map cannot be run; here
it is pinned at sizes where it can: everything the oracle's map produces passes in every slot, and every single-word
corruption of any output vector -- or of the assignment -- is reported in exactly the slot and identity it belongs to."""
import numpy as np
import pytest

from oracle import oracle as O
from ringsnark_amd import params as P
from ringsnark_amd import r1cs as R
from tests import helpers as H


def _points(q, m, seed, n=2):
    rng = np.random.RandomState(seed)
    return [m + int(rng.randint(0, 2**31)) * 65537 % (q - m) for _ in range(n)]


def _case(preset, m, zk, poly=False, seed=3):
    prm = P.preset(preset)
    ctx = H.oracle_ctx(prm)
    if poly:
        cs = R.wide_poly_r1cs(m, prm.q, prm.N)
    else:
        cs = R.wide_r1cs(m, prm.q) if m > 1 else R.chain_r1cs(m, prm.q)
    asg = H.make_assignment(ctx, cs, seed=seed)
    ds = [ctx.random_ring(40 + k) for k in range(3)] if zk else [None] * 3
    return prm, ctx, cs, asg, ds


@pytest.mark.parametrize("preset,m,zk,poly", [("toy", 1, False, False), ("toy", 5, False, False), ("toy", 12, True, False),
                                              ("toy", 16, True, False), ("toy49", 33, True, False), ("toy60", 20, True, False),
                                              ("toy", 9, True, True)])
def test_the_oracle_map_satisfies_every_identity_in_every_slot(preset, m, zk, poly):
    prm, ctx, cs, asg, ds = _case(preset, m, zk, poly)
    ocs = H.oracle_cs(cs)
    for limb in range(prm.L):
        q = int(prm.q[limb])
        d = [None if x is None else np.ascontiguousarray(x[limb]) for x in ds]
        w = O.witness_map(q, ocs, limb, np.ascontiguousarray(asg[:, limb, :]), *d)
        # strided views of the [rows][L][N] arrays, as the configuration-scale callers pass them
        n_bad, bad = O.witness_identities(q, ocs, limb, asg[:, limb, :], {k: w[k] for k in O.IDENTITY_NAMES},
                                          _points(q, m, 11 + limb, 3), *d, Z=w["Z"], threads=2)
        assert n_bad == 0 and not bad.any()


def test_a_single_wrong_word_is_found_where_it_is():
    prm, ctx, cs, asg, ds = _case("toy49", 24, True)
    ocs = H.oracle_cs(cs)
    limb = prm.L - 1
    q = int(prm.q[limb])
    d = [np.ascontiguousarray(x[limb]) for x in ds]
    a = np.ascontiguousarray(asg[:, limb, :])
    w = O.witness_map(q, ocs, limb, a, *d)
    pts = _points(q, cs.m, 5)
    rng = np.random.RandomState(1)
    for bit, name in enumerate(O.IDENTITY_NAMES):
        row, slot = int(rng.randint(w[name].shape[0])), int(rng.randint(prm.N))
        v = {k: w[k] for k in O.IDENTITY_NAMES}
        v[name] = w[name].copy()
        v[name][row, slot] = (int(v[name][row, slot]) + 1) % q
        n_bad, bad = O.witness_identities(q, ocs, limb, a, v, pts, *d)
        assert n_bad == 1 and bad[slot] == 1 << bit and bad.sum() == 1 << bit, (name, n_bad, bad[slot])
    # the last row of H (the reference's H has m + 1 coefficients, r1cs_to_qrp.tcc:225-253)
    v = {k: w[k] for k in O.IDENTITY_NAMES}
    v["H"] = w["H"].copy()
    v["H"][cs.m, 3] = (int(v["H"][cs.m, 3]) + 5) % q
    n_bad, bad = O.witness_identities(q, ocs, limb, a, v, pts, *d)
    assert n_bad == 1 and bad[3] == 1 << 6
    # a wrong auxiliary wire: the mid vectors and H of that slot no longer match (the io vectors still do)
    a2 = a.copy()
    a2[cs.n_inputs + 2, 7] ^= 1
    n_bad, bad = O.witness_identities(q, ocs, limb, a2, {k: w[k] for k in O.IDENTITY_NAMES}, pts, *d)
    assert n_bad == 1 and bad[7] & 0b0111000 and not bad[7] & 0b0000111
    # a wrong ZK shift
    d2 = [x.copy() for x in d]
    d2[2][9] = (int(d2[2][9]) + 1) % q
    n_bad, bad = O.witness_identities(q, ocs, limb, a, {k: w[k] for k in O.IDENTITY_NAMES}, pts, *d2)
    assert n_bad == 1 and bad[9] == 1 << 6
    # Z
    Zbad = w["Z"].copy()
    Zbad[1] = (int(Zbad[1]) + 1) % q
    with pytest.raises(AssertionError):
        O.witness_identities(q, ocs, limb, a, {"H": w["H"]}, pts, *d, Z=Zbad)
    # points must lie outside the domain
    with pytest.raises(ValueError):
        O.witness_identities(q, ocs, limb, a, {"H": w["H"]}, [cs.m - 1], *d)


def test_subsets_of_vectors_and_slot_ranges():
    """Callers check what they hold: ringGroth16 has no C vectors, a rank of the slot split holds a slot range."""
    prm, ctx, cs, asg, ds = _case("toy", 10, False)
    ocs = H.oracle_cs(cs)
    q = int(prm.q[0])
    a = np.ascontiguousarray(asg[:, 0, :])
    w = O.witness_map(q, ocs, 0, a)
    pts = _points(q, cs.m, 8)
    lo, hi = 5, 5 + 21
    sub = {k: w[k][:, lo:hi] for k in ("A_io", "A_mid", "B_io", "B_mid", "H")}
    n_bad, bad = O.witness_identities(q, ocs.at_slots(lo), 0, a[:, lo:hi], sub, pts)
    assert n_bad == 0 and bad.shape == (21,)
    sub["B_mid"] = sub["B_mid"].copy()
    sub["B_mid"][4, 20] ^= 2
    n_bad, bad = O.witness_identities(q, ocs.at_slots(lo), 0, a[:, lo:hi], sub, pts)
    assert n_bad == 1 and bad[20] == 1 << 4


def test_blocked_layout_gives_the_same_verdicts():
    """[S/32][rows][32]: the layout the configuration-scale callers hand over (a block of 32 columns streams through
    memory); same answers as the row-major form, corruption found in the same slot."""
    prm, ctx, cs, asg, ds = _case("toy", 11, True)
    ocs = H.oracle_cs(cs)
    q = int(prm.q[1])
    assert prm.N % 32 == 0
    d = [np.ascontiguousarray(x[1]) for x in ds]
    a = np.ascontiguousarray(asg[:, 1, :])
    w = O.witness_map(q, ocs, 1, a, *d)
    blk = lambda t: np.ascontiguousarray(t.reshape(t.shape[0], -1, 32).transpose(1, 0, 2))
    pts = _points(q, cs.m, 21)
    v = {k: blk(w[k]) for k in O.IDENTITY_NAMES}
    n_bad, bad = O.witness_identities(q, ocs, 1, blk(a), v, pts, *d, Z=w["Z"], blocked=True)
    assert n_bad == 0
    wrong = w["C_mid"].copy()
    wrong[5, prm.N - 1] ^= 4
    v["C_mid"] = blk(wrong)
    n_bad, bad = O.witness_identities(q, ocs, 1, blk(a), v, pts, *d, blocked=True)
    assert n_bad == 1 and bad[prm.N - 1] == 1 << 5
