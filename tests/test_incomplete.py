"""Incomplete transforms (ringsnark_amd/csrc/witness_inc.hpp): the witness map's long products on ring primes WITHOUT a root of
unity of the product's order -- what the reference's own recipe produces (default_double_batching_modulus,
seal/seal_util.hpp:20-32: q_i = 1 mod 2 N_inner only; examples/example_SEAL.cpp:15-22) while its O(m^2) map
(reductions/r1cs_to_qrp/r1cs_to_qrp.tcc:149-259) works for any prime.  The multi-pass path runs as for well-endowed primes,
every transform stopped at the prime's 2-adicity; the pointwise step multiplies residues modulo x^G - eta.

  * against the CPU oracle's literal O(m^2) map at sizes it can follow, small LDS tiles forcing the multi-pass path: one to
    four stages short, both arithmetics, primes whose 2-adicity differs per limb (generic kernel sub_ntt_kernel);
  * at the headline's sizes and ON THE HEADLINE'S PRIMES (preset toyC3: 2-adicity 15, 15, 14, 14 on a 32-slot ring) through the
    tuned kernels (sub_ntt_wide_kernel<., INC>, sub_ntt_w12_kernel<., INC>): every column through the identities that define
    the map's outputs (oracle/rs_identities.c), and bit-equal to the two-dimensional block convolutions it replaces and --
    forced on well-endowed primes -- to complete transforms."""
import numpy as np
import pytest

from oracle import oracle as O
from ringsnark_amd import params as P
from ringsnark_amd import r1cs as R
from tests import helpers as H
from tests import proof_check

pytestmark = pytest.mark.gpu

KEYS = ("A_io", "B_io", "C_io", "A_mid", "B_mid", "C_mid", "H")


def _set_tuning(key, value):
    from ringsnark_amd import _lib
    _lib.check(_lib.load().rs_set_tuning(key, value))


class _knobs:
    """Set tuning knobs, restore the library defaults on exit (plans read them when they are built: use a fresh Device inside)."""
    DEFAULTS = {b"witness_lds_logM": 13, b"witness_force_bc": 0, b"witness_inc": 1, b"witness_bc2": 1, b"force_int_arith": 0,
                b"witness_sub_log": 12, b"witness_sub_ct": 2}

    def __init__(self, **kv):
        self.kv = {k.encode(): v for k, v in kv.items()}

    def __enter__(self):
        for k, v in self.kv.items():
            _set_tuning(k, v)

    def __exit__(self, *exc):
        for k in self.kv:
            _set_tuning(k, self.DEFAULTS[k])


def _host(t):
    from ringsnark_amd.device import to_host
    return to_host(t)


@pytest.mark.parametrize("name,lds,force,m,kind,zk", [
    ("toy", 6, 7, 300, "wide", True),        # 2M = 2^10 on "2-adicity 7": Newton and H three stages short, tree levels 8, 9 one and two
    ("toy", 6, 7, 600, "chain", False),      # 2M = 2^11: four stages short (leaves of 16 words)
    ("toy", 6, 6, 100, "wide", True),        # 2-adicity = the tile: every multi-pass transform is incomplete
    ("toy", 6, 7, 70, "many_inputs", False),  # generic evaluate + interpolate io path (70 primary inputs); 2M = 2^8: one stage short
    ("toy", 6, 7, 512, "wide", True),        # m = M: Z has M + 1 coefficients
    ("toy60", 6, 7, 300, "wide", True),      # Montgomery integers
    ("toy60", 6, 7, 600, "chain", False),
    ("toyR", 6, 0, 600, "wide", True),       # recipe primes as they come (2-adicity 7, 7)
    ("toy49", 6, 7, 300, "wide", True),      # 49-bit primes: the reductions inside the polynomial products matter
    ("toy", 6, 7, 300, "int", True),         # the FP64 primes on the integer arithmetic
])
def test_incomplete_transforms_match_oracle(name, lds, force, m, kind, zk):
    from ringsnark_amd.device import Device
    prm = P.preset(name)
    int_arith = kind == "int"
    with _knobs(witness_lds_logM=lds, witness_force_bc=force, force_int_arith=int(int_arith)):
        dev = Device(prm)
        ctx = H.oracle_ctx(prm)
        cs = {"wide": lambda: R.wide_r1cs(m, prm.q), "int": lambda: R.wide_r1cs(m, prm.q), "chain": lambda: R.chain_r1cs(m, prm.q),
              "many_inputs": lambda: R.wide_r1cs(m, prm.q, n_inputs=70)}[kind]()
        asg = H.make_assignment(ctx, cs)
        ds = [ctx.random_ring(60 + k) for k in range(3)] if zk else [None] * 3
        dev.set_profiling(True)
        w = dev.witness_map(dev.r1cs(cs), dev.put(asg), *[dev.put(d) if d is not None else None for d in ds])
        got = {k: _host(v) if k != "Z" else v for k, v in w.items()}
        names = {k["name"] for k in dev.profile_read()}
        dev.set_profiling(False)
    assert not any(n.startswith("bc") for n in names), names  # not the block convolutions
    assert any(n.startswith("sub_ntt_kernel<2") for n in names) and any(n.startswith("cross_kernel") for n in names), names
    ocs = H.oracle_cs(cs)
    for limb in range(prm.L):
        dl = [np.ascontiguousarray(d[limb]) if d is not None else None for d in ds]
        exp = O.witness_map(prm.q[limb], ocs, limb, np.ascontiguousarray(asg[:, limb, :]), *dl)
        for k in KEYS:
            assert (got[k][:, limb, :] == exp[k]).all(), (k, limb)
        assert (got["Z"][limb] == exp["Z"]).all()


def test_rinocchio_and_groth16_provers_on_incomplete_transforms_match_oracle():
    """Both provers end to end (Rinocchio interpolates C: the coset form of H does not apply, the long-division form runs)."""
    from ringsnark_amd.device import Device
    prm = P.preset("toyR")
    m = 150  # 2M = 2^9 on 2-adicity 7
    with _knobs(witness_lds_logM=6):
        dev, ctx = Device(prm), H.oracle_ctx(prm)
        cs = R.wide_r1cs(m, prm.q)
        asg = H.make_assignment(ctx, cs)
        ds = [ctx.random_ring(60 + k) for k in range(3)]
        pk = dict(s_pows=ctx.random_enc(71, m + 1), delta_ts=ctx.random_enc(72, m + 1), delta_mid=ctx.random_enc(73, cs.n_aux),
                  alpha=ctx.random_enc(74), beta=ctx.random_enc(75))
        rk = dict(s_pows=ctx.random_enc(81, m + 1), alpha_s_pows=ctx.random_enc(82, m + 1), beta_prods=ctx.random_enc(83, cs.n_aux),
                  beta_rv_ts=ctx.random_enc(84), beta_rw_ts=ctx.random_enc(85), beta_ry_ts=ctx.random_enc(86))
        dcs, dasg = dev.r1cs(cs), dev.put(asg)
        dev.set_profiling(True)
        gp, gempty = dev.groth16_prove(dcs, {k: dev.put(v) for k, v in pk.items()}, dasg)
        rp, rempty = dev.rinocchio_prove(dcs, {k: dev.put(v) for k, v in rk.items()}, dasg, *[dev.put(d) for d in ds])
        gp, rp = _host(gp), _host(rp)
        names = {k["name"] for k in dev.profile_read()}
        dev.set_profiling(False)
    assert not any(n.startswith("bc") for n in names) and any(n.startswith("sub_ntt_kernel<3") for n in names), names
    ocs = H.oracle_cs(cs)
    exp, exp_empty = O.groth16_prove(ctx, ocs, pk, asg)
    assert gempty == exp_empty and (gp == exp).all()
    exp, exp_empty = O.rinocchio_prove(ctx, ocs, rk, asg, *ds)
    assert rempty == exp_empty and (rp == exp).all()


def _run_large(prm, m, zk, want=KEYS, **knobs):
    import torch

    from ringsnark_amd.device import Device
    octx = H.oracle_ctx(prm)
    cs = R.chain_r1cs(m, prm.q)
    with _knobs(**knobs):
        dev = Device(prm)  # fresh context: plans are cached per context and read the knobs when they are built
        asg = dev.ring_empty(m + 2)
        dev.fill_uniform(asg[:2], 0, 9)
        dev.chain_assignment(asg, m)
        ds = [dev.put(octx.random_ring(60 + k)) for k in range(3)] if zk else [None] * 3
        dev.set_profiling(True)
        w = dev.witness_map(dev.r1cs(cs), asg, *ds, want=want)
        torch.cuda.synchronize()
        names = {k["name"] for k in dev.profile_read()}
        dev.set_profiling(False)
    return dev, cs, asg, ds, w, names


@pytest.mark.parametrize("m,zk", [(10000, True), (16384, False), (20000, True), (40000, False), (65536, True), (100000, False), (131072, True)])
def test_headline_primes_through_the_tuned_kernels(m, zk):
    """The ring primes of the headline (2-adicity 15, 15, 14, 14: per-limb launches of the tuned sub-transform kernels), M = 2^14
    (the one-tile size: a prime without a 2^15-th root takes the multi-pass path there) .. 2^17 (four stages short at most: 2^18 constraints on these primes take the block convolutions): every column through the identities, and every vector bit-equal to the two-dimensional block convolutions."""
    prm = P.preset("toyC3")
    dev, cs, asg, ds, w, names = _run_large(prm, m, zk)
    logM = (m - 1).bit_length()
    assert not any(n.startswith("bc") for n in names), names
    # transforms of length 2M: logM + 1 - 14 stages short on the 2-adicity-14 primes, one fewer on the others
    top = logM + 1 - 14
    tuned = {n for n in names if n.startswith("sub_ntt_wide_kernel") or n.startswith("sub_ntt_w12_kernel")}
    assert any(n.endswith(", %d>" % top) for n in tuned) and any(n.endswith(", %d>" % (top - 1)) for n in tuned), names
    err, info = proof_check.check_all_columns(prm, cs, asg, {k: w[k] for k in KEYS}, tuple(ds), seed=m % 1000, Z=w["Z"])
    assert err is None and info["columns"] == prm.L * prm.N, err
    got = {k: _host(w[k]) for k in KEYS}
    if not zk:
        assert not got["H"][m - 1:].any()
    del dev, asg, w
    dev, cs, asg, ds, w, names = _run_large(prm, m, zk, witness_inc=0)
    assert any(n.startswith("bc2_" if logM >= 15 else "bc_") for n in names), names  # M = 2^14: the pairwise block convolutions
    for k in KEYS:
        assert (_host(w[k]) == got[k]).all(), k


@pytest.mark.parametrize("m,force,sub_log", [(40000, 14, 12), (40000, 15, 13), (65536, 14, 13), (65536, 16, 12), (131072, 14, 12), (131072, 15, 12)])
def test_incomplete_transforms_equal_complete_ones(m, force, sub_log):
    """Forced on primes that have the roots (toy44: = 1 mod 2^20): bit-equal to the complete transforms, on blocks of 2^12 and 2^13."""
    prm = P.preset("toy44")
    dev, cs, asg, ds, w, names = _run_large(prm, m, True, witness_force_bc=force, witness_sub_log=sub_log)
    assert not any(n.startswith("bc") for n in names), names
    inc = (m - 1).bit_length() + 1 - force
    assert any(n.endswith(", %d>" % inc) for n in names if n.startswith("sub_ntt_w")), names
    got = {k: _host(w[k]) for k in KEYS}
    del dev, asg, w
    dev, cs, asg, ds, w, names = _run_large(prm, m, True, witness_sub_log=sub_log)
    assert not any(n.endswith(", %d>" % inc) for n in names if n.startswith("sub_ntt_w")), names
    for k in KEYS:
        assert (_host(w[k]) == got[k]).all(), k


def test_generic_and_tuned_sub_transform_kernels_agree_on_incomplete_transforms():
    prm = P.preset("toyC3")
    runs = {}
    for ct in (0, 2):
        dev, cs, asg, ds, w, names = _run_large(prm, 40000, True, witness_sub_ct=ct, witness_sub_log=13 if ct == 0 else 12)
        assert any(n.startswith("sub_ntt_kernel<2") for n in names) == (ct == 0), names
        runs[ct] = {k: _host(w[k]) for k in KEYS}
        del dev, asg, w
    for k in KEYS:
        assert (runs[0][k] == runs[2][k]).all(), k


@pytest.mark.parametrize("m", [1048576, 2097152])
def test_beyond_the_two_adicity_at_2_20_and_2_21(m):
    """toy44's primes are = 1 mod 2^20: 2^20 constraints (transforms of 2^21 words, one stage short) took the two-level block
    convolutions until round 6 and 2^21 constraints were refused; both now run the multi-pass path (every column through the identities)."""
    prm = P.preset("toy44")
    want = ("A_mid", "B_mid", "H")
    dev, cs, asg, ds, w, names = _run_large(prm, m, False, want=want)
    assert not any(n.startswith("bc") for n in names), names
    inc = (m - 1).bit_length() + 1 - 20
    assert any(n.endswith(", %d>" % inc) for n in names if n.startswith("sub_ntt_w")), names
    err, info = proof_check.check_all_columns(prm, cs, asg, {k: w[k] for k in want}, tuple(ds), seed=5, Z=w["Z"])
    assert err is None and info["columns"] == prm.L * prm.N, err
