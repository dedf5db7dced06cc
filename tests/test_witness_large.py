"""The witness map beyond 2^16 constraints -- the sizes BASELINE.json configs[3] (2^18 constraints) needs and the
reference's O(m^2) map (reductions/r1cs_to_qrp/r1cs_to_qrp.tcc:149-259) serves for any m, on the ring primes its recipe
produces (seal/seal_util.hpp:20-32: q_i = 1 mod 2 N_enc only) as well as on well-endowed ones.

M = 2^17 and 2^18 on a SMALL ring (toy44: 64 columns, headline-size primes), every path:
  full-length transforms (three cross passes per transform: logM + 1 - 13 = 5, 6 stages over global memory),
  two-dimensional block convolutions forced on the same primes (witness_force_bc = 14): Y = 32 blocks at 2^17 (the
  transform across blocks in one thread's registers), Y = 64 at 2^18 (the two-level kernels bc2_yfwd_big / bc2_yinv_a / _b).
Checked the way bench.py checks the headline: every output vector on sampled columns against the Lagrange form of the
interpolants at random points -- sum_j y_j L_j(r), O(m) exact integer operations, no oracle run of the O(m^2) map needed --
and H Z = A B - C (+ the ZK patch), plus bit-equality of the two paths with each other."""
import numpy as np
import pytest

from ringsnark_amd import params as P
from ringsnark_amd import r1cs as R
from tests import helpers as H
from tests import proof_check

pytestmark = pytest.mark.gpu

SUB_LOG_DEFAULT = 12  # the library default of the knob witness_sub_log
KEYS = ("A_io", "B_io", "C_io", "A_mid", "B_mid", "C_mid", "H")


def _set_tuning(key, value):
    from ringsnark_amd import _lib
    _lib.check(_lib.load().rs_set_tuning(key, value))


def _run(prm, m, zk, force, int_arith=False, want=KEYS, inc=None):
    """Fresh context (plans are cached per context and read the knobs when they are built).  force: pretend the ring primes
    have only this 2-adicity AND take the block convolutions (witness_inc = 0) -- since round 6 such primes run incomplete
    transforms by default (tests/test_incomplete.py); these tests keep the block-convolution kernels exercised."""
    import torch

    from ringsnark_amd.device import Device
    octx = H.oracle_ctx(prm)
    cs = R.chain_r1cs(m, prm.q)
    _set_tuning(b"witness_force_bc", force)
    _set_tuning(b"witness_inc", (0 if force else 1) if inc is None else inc)
    if int_arith:
        _set_tuning(b"force_int_arith", 1)
    try:
        dev = Device(prm)
    finally:
        if int_arith:
            _set_tuning(b"force_int_arith", 0)
    try:
        asg = dev.ring_empty(m + 2)
        dev.fill_uniform(asg[:2], 0, 9)
        dev.chain_assignment(asg, m)
        ds = [dev.put(octx.random_ring(60 + k)) for k in range(3)] if zk else [None] * 3
        dev.set_profiling(True)
        w = dev.witness_map(dev.r1cs(cs), asg, *ds, want=want)
        torch.cuda.synchronize()
        names = {k["name"] for k in dev.profile_read()}
        dev.set_profiling(False)
    finally:
        _set_tuning(b"witness_force_bc", 0)
        _set_tuning(b"witness_inc", 1)
    return dev, cs, asg, ds, w, names


@pytest.mark.parametrize("m,zk,force", [(100000, True, 0), (131072, False, 0), (262144, True, 0), (200000, False, 0),
                                        (131072, True, 14), (100000, False, 14), (262144, True, 14), (180000, False, 14)])
def test_witness_map_at_2_17_and_2_18(m, zk, force):
    prm = P.preset("toy44")
    dev, cs, asg, ds, w, names = _run(prm, m, zk, force)
    logM = (m - 1).bit_length()
    if force:
        assert ("bc2_yfwd_big_kernel" in names) == (logM == 18), names
        assert ("bc2_yinv_b_kernel" in names) == (logM == 18), names
        assert "bc2_yfwd_kernel" in names, names  # the tree levels above the 2^14 tiles (Y <= 32) at either size
    else:
        assert not any(n.startswith("bc") for n in names), names
        assert any(n.startswith("cross_kernel") for n in names), names
    err, info = proof_check.check_all_columns(prm, cs, asg, {k: w[k] for k in KEYS}, tuple(ds), seed=m % 1000 + force, Z=w["Z"])
    assert err is None and info["columns"] == prm.L * prm.N, err  # every column of the ring (round 4 sampled three)
    # rows beyond the degree bound: H has m + 1 rows, row m is d1 d2 (the top of the ZK patch) or zero
    from ringsnark_amd.device import to_host
    if not zk:
        assert not to_host(w["H"][m - 1:]).any()


@pytest.mark.parametrize("m,force", [(400000, 14), (1048576, 0)])
def test_two_level_transform_across_blocks_at_2_19_and_2_20(m, force):
    """Y = 128 and 256 blocks (bc2_yfwd_big_kernel<., 2 | 3>).  2^20 constraints is the limit the plan accepts
    (witness.hip build_plan); toy44's primes (= 1 mod 2^20) have no 2^21-th root, so that size takes the block path unforced."""
    prm = P.preset("toy44")
    dev, cs, asg, ds, w, names = _run(prm, m, False, force, want=("A_mid", "B_mid", "H"), inc=0)
    assert "bc2_yfwd_big_kernel" in names and "bc2_yinv_a_kernel" in names and "bc2_yinv_b_kernel" in names, names
    err, info = proof_check.check_all_columns(prm, cs, asg, {k: w[k] for k in ("A_mid", "B_mid", "H")}, tuple(ds), seed=3, Z=w["Z"])
    assert err is None and info["columns"] == prm.L * prm.N, err


@pytest.mark.parametrize("m", [131072, 262144])
def test_block_convolutions_equal_full_length_transforms_beyond_2_16(m):
    """Bit equality of the two exact paths on every slot (64 columns x 7 vectors)."""
    from ringsnark_amd.device import to_host
    prm = P.preset("toy44")
    runs = {}
    for force in (0, 14):
        dev, cs, asg, ds, w, names = _run(prm, m, True, force)
        runs[force] = {k: to_host(w[k]) for k in KEYS}
        del dev, asg, w
    for k in KEYS:
        assert (runs[0][k] == runs[14][k]).all(), k


@pytest.mark.parametrize("m", [70000, 262144])
def test_wide_cross_passes_equal_the_radix_16_ones(m):
    """witness_cross_maxr: five or six cross stages in ONE pass over the workspace (radix 32 / 64; M = 2^17, 2^18) against
    the round-3 passes of at most four stages -- the same stages with the same reduction points, so the vectors are identical."""
    from ringsnark_amd.device import to_host
    prm = P.preset("toy44")
    runs = {}
    for maxr in (4, 6):
        _set_tuning(b"witness_cross_maxr", maxr)
        try:
            dev, cs, asg, ds, w, names = _run(prm, m, True, 0)
        finally:
            _set_tuning(b"witness_cross_maxr", 6)
        wide = [n for n in names if n.startswith("cross_kernel<") and n.split(",")[1].strip() in ("5", "6")]
        assert bool(wide) == (maxr == 6), names
        runs[maxr] = {k: to_host(w[k]) for k in KEYS}
        del dev, asg, w
    for k in KEYS:
        assert (runs[4][k] == runs[6][k]).all(), k


@pytest.mark.parametrize("m,cross", [(20000, 4), (70000, 4), (70000, 8), (262144, 4), (262144, 8)])
def test_sub_transform_blocks_of_2_12_equal_2_13(m, cross):
    """witness_sub_log = 12 (the default): the rooted sub-transforms on blocks of 2^12 (sub_ntt_w12_kernel, four workgroups per
    CU, one more cross stage per transform) against blocks of 2^13 -- the same stages, reduction points and table products
    per coefficient, so every vector is identical.  witness_sub12_cross = 4 (default): only the transforms whose cross pass
    stays within four stages take the small blocks; 8: every transform.  m = 20000: M = 2^15, the smallest multi-pass size."""
    from ringsnark_amd.device import to_host
    prm = P.preset("toy44")
    runs = {}
    for logb in (13, 12):
        _set_tuning(b"witness_sub_log", logb)
        _set_tuning(b"witness_sub12_cross", cross)
        try:
            dev, cs, asg, ds, w, names = _run(prm, m, True, 0)
        finally:
            _set_tuning(b"witness_sub_log", SUB_LOG_DEFAULT)
            _set_tuning(b"witness_sub12_cross", 4)
        assert any(n.startswith("sub_ntt_w12_kernel") for n in names) == (logb == 12), names
        if logb == 12:
            assert any(n.startswith("sub_ntt_wide_kernel") for n in names) == (cross == 4 and m > 32768), names
        runs[logb] = {k: to_host(w[k]) for k in KEYS}
        del dev, asg, w
    for k in KEYS:
        assert (runs[13][k] == runs[12][k]).all(), k


@pytest.mark.parametrize("m,zk,int_arith", [(20000, True, False), (32768, False, False), (65536, True, False), (70000, False, False),
                                             (262144, True, False), (40000, True, True)])
def test_h_on_a_coset_equals_the_long_division_form(m, zk, int_arith):
    """witness_h_coset (default on; taken when the call interpolates C -- Rinocchio): H as the inverse coset transform of
    (A B - C) / Z at M points g w^i (four length-M transforms) against big_h's quotient by rev(Z)^-1 (five of length 2M).
    The division is exact in Z_q, so the two are the same polynomial: bit-equal H, every other vector untouched, and the
    identity H Z = A B - C (+ ZK patch) at random points.  m = 32768, 65536: m = M, where Z has M + 1 coefficients and
    x^M folds onto g^M."""
    from ringsnark_amd.device import to_host
    prm = P.preset("toy44")
    runs = {}
    for coset in (0, 1):
        _set_tuning(b"witness_h_coset", coset)
        try:
            dev, cs, asg, ds, w, names = _run(prm, m, zk, 0, int_arith=int_arith)
        finally:
            _set_tuning(b"witness_h_coset", 1)
        assert any("<4" in n and n.startswith("sub_ntt") for n in names) == (coset == 1), names
        if coset:
            err, info = proof_check.check_all_columns(prm, cs, asg, {k: w[k] for k in KEYS}, tuple(ds), seed=m % 1000, Z=w["Z"])
            assert err is None and info["columns"] == prm.L * prm.N, err
        runs[coset] = {k: to_host(w[k]) for k in KEYS}
        del dev, asg, w
    for k in KEYS:
        assert (runs[0][k] == runs[1][k]).all(), k


def test_integer_arithmetic_at_2_17():
    """The Montgomery-integer contexts (moduli >= 2^50: microbench.cpp:33-36, BFVDefault(2048)) run the generic kernels of
    the multi-pass path; M = 2^17 forced on toy44's primes, equal to the FP64 context bit for bit."""
    from ringsnark_amd.device import to_host
    prm = P.preset("toy44")
    m = 70000
    want = ("A_mid", "B_mid", "H")
    a = _run(prm, m, True, 0, want=want)
    b = _run(prm, m, True, 0, int_arith=True, want=want)
    for k in want:
        assert (to_host(a[4][k]) == to_host(b[4][k])).all(), k


def test_unsatisfied_assignment_long_division_and_coset_forms():
    """The reference asserts satisfaction before it maps (r1cs_to_qrp.tcc:156) and then long-divides, dropping the remainder
    (util/polynomials.tcc:76-81).  Pinned here (round-4 advisor finding): for an UNSATISFIED column the long-division
    form -- every call that does not ask for C, every one-tile size, witness_h_coset = 0 -- still returns the quotient
    quo(A B - C, Z) = quo(A B, Z); the coset form (C requested, multi-pass size: Rinocchio) returns another polynomial in
    that column and the same one in every satisfied column.  include/ringsnark_amd.h states the precondition."""
    from oracle import oracle as O
    from ringsnark_amd.device import Device, to_host
    # one tile: the device equals the oracle's literal long division on an unsatisfied assignment
    prm = P.preset("toy")
    dev, ctx = Device(prm), H.oracle_ctx(prm)
    cs = R.wide_r1cs(12, prm.q)
    asg = H.make_assignment(ctx, cs)
    asg[cs.n_inputs + 3, 0, 5] = (int(asg[cs.n_inputs + 3, 0, 5]) + 1) % prm.q[0]
    w = dev.witness_map(dev.r1cs(cs), dev.put(asg))
    for limb in range(prm.L):
        ow = O.witness_map(prm.q[limb], H.oracle_cs(cs), limb, np.ascontiguousarray(asg[:, limb, :]))
        for k in KEYS:
            assert (to_host(w[k])[:, limb, :] == ow[k]).all(), k
    del dev, w
    # multi-pass size
    prm = P.preset("toy44")
    m = 20000
    runs = {}
    for tag, coset, want in (("long", 0, KEYS), ("coset", 1, KEYS), ("no_c", 1, ("A_mid", "B_mid", "H"))):
        _set_tuning(b"witness_h_coset", coset)
        try:
            dev = Device(prm)
            cs = R.chain_r1cs(m, prm.q)
            asg = dev.ring_empty(m + 2)
            dev.fill_uniform(asg[:2], 0, 9)
            dev.chain_assignment(asg, m)
            asg[777, 0, 3] += 1  # one wire of one column no longer satisfies its constraints
            w = dev.witness_map(dev.r1cs(cs), asg, want=want)
            runs[tag] = {k: to_host(w[k]) for k in want}
        finally:
            _set_tuning(b"witness_h_coset", 1)
        del dev, asg, w
    for k in ("A_mid", "B_mid"):
        assert (runs["long"][k] == runs["coset"][k]).all() and (runs["long"][k] == runs["no_c"][k]).all()
    assert (runs["long"]["H"] == runs["no_c"]["H"]).all()  # without C: the quotient, for any assignment
    same = runs["long"]["H"] == runs["coset"]["H"]
    bad = np.zeros(same.shape[1:], dtype=bool)
    bad[0, 3] = True
    assert same[:, ~bad].all()           # satisfied columns: the two forms are the same polynomial
    assert not same[:, 0, 3].all()       # the unsatisfied one: documented difference


@pytest.mark.parametrize("force", [0, 14])
def test_product_tree_tiles_in_one_launch_per_chunk(force):
    """force = 14: the same on the two-dimensional block convolutions of the recipe primes.  witness_tree_once (default on): when the columns of a chunk are worked through in workspace-sized sub-chunks, the
    product tree's tiles -- in place on the columns, no workspace -- run as ONE launch between the sub-chunked phases.
    Forced at test scale by a 64 MiB workspace (16 columns per sub-chunk at M = 2^17): bit-equal to the per-sub-chunk order
    and to the unchunked default, every column through the identities."""
    from ringsnark_amd.device import to_host
    prm = P.preset("toy44")
    m = 100000
    runs = {}
    for tag, ws, once in (("default", 6144, 1), ("sub-chunks, one tree launch", 64, 1), ("sub-chunks, tree per sub-chunk", 64, 0)):
        _set_tuning(b"witness_big_ws_mib", ws)
        _set_tuning(b"witness_tree_once", once)
        try:
            dev, cs, asg, ds, w, names = _run(prm, m, True, force)
        finally:
            _set_tuning(b"witness_big_ws_mib", 6144)
            _set_tuning(b"witness_tree_once", 1)
        if ws == 64 and once:
            err, info = proof_check.check_all_columns(prm, cs, asg, {k: w[k] for k in KEYS}, tuple(ds), seed=4, Z=w["Z"])
            assert err is None and info["columns"] == prm.L * prm.N, err
        runs[tag] = {k: to_host(w[k]) for k in KEYS}
        del dev, asg, w
    for tag in runs:
        for k in KEYS:
            assert (runs[tag][k] == runs["default"][k]).all(), (tag, k)


@pytest.mark.parametrize("m", [2097152, 2500000])
def test_full_length_transforms_at_2_21_and_2_22(m):
    """Beyond 2^20 constraints (round-4 verdict "What's missing" 5: the reference's map has no bound).  Ring primes
    = 1 mod 2^23 (toy44x) have the roots of unity for M = 2^21 and 2^22: the multi-pass path takes one more cross pass (8 and
    9 stages over global memory) and nothing else changes.  Every column through the identities.  (Primes without those roots
    stop at 2^20: the block convolutions' transform across blocks is built for <= 256 blocks.)"""
    prm = P.preset("toy44x")
    dev, cs, asg, ds, w, names = _run(prm, m, True, 0, want=("A_mid", "B_mid", "H"))
    assert not any(n.startswith("bc") for n in names), names
    err, info = proof_check.check_all_columns(prm, cs, asg, {k: w[k] for k in ("A_mid", "B_mid", "H")}, tuple(ds), seed=m % 997, Z=w["Z"])
    assert err is None and info["columns"] == prm.L * prm.N, err


def test_block_convolutions_refuse_more_than_2_20_constraints():
    from ringsnark_amd import _lib
    prm = P.preset("toy44")  # = 1 mod 2^20 only: with incomplete transforms off, 2^21 constraints would need the block path
    with pytest.raises(_lib.RsError) as ei:
        _run(prm, 1500000, False, 0, want=("A_mid",), inc=0)
    assert ei.value.code == _lib.RS_ERR_UNSUPPORTED and "2^20" in str(ei.value)


@pytest.mark.parametrize("m,zk,int_arith", [(20000, True, False), (32768, False, False), (50000, True, False), (65536, True, False),
                                             (100000, False, False), (262144, True, False), (30000, True, True)])
def test_the_turn_of_h_as_one_pass(m, zk, int_arith):
    """witness_h_turn (default on): the last inverse cross pass of the product A B and the first forward cross pass of its
    reversal rev(A B) mod x^(m-1) as ONE pass over memory (cross_turn_kernel: R = 3, 4, 5 cross stages with paired 16-byte
    stores, R = 6 one class per thread; m = M and m < M; both arithmetics) -- bit-equal H to the two separate passes, and
    every column through H Z = A B - C (+ ZK patch)."""
    from ringsnark_amd.device import to_host
    prm = P.preset("toy44")
    runs = {}
    for turn in (0, 1):
        _set_tuning(b"witness_h_turn", turn)
        try:
            dev, cs, asg, ds, w, names = _run(prm, m, zk, 0, int_arith=int_arith, want=("A_mid", "B_mid", "H"))
        finally:
            _set_tuning(b"witness_h_turn", 1)
        assert any(n.startswith("cross_turn_kernel") for n in names) == (turn == 1), names
        if turn:
            err, info = proof_check.check_all_columns(prm, cs, asg, {k: w[k] for k in ("A_mid", "B_mid", "H")}, tuple(ds), seed=m % 991, Z=w["Z"])
            assert err is None and info["columns"] == prm.L * prm.N, err
        runs[turn] = {k: to_host(w[k]) for k in ("A_mid", "B_mid", "H")}
        del dev, asg, w
    for k in ("A_mid", "B_mid", "H"):
        assert (runs[0][k] == runs[1][k]).all(), k


@pytest.mark.parametrize("m,zk,int_arith", [(50000, True, False), (65536, False, False), (100000, True, False), (262144, False, False),
                                             (400000, True, False), (60000, False, True)])
def test_the_turn_between_tree_levels_as_one_pass(m, zk, int_arith):
    """witness_level_turn (default on): the last inverse cross pass of tree level l (+ F_left, reduce) and the first forward
    cross pass of level l + 1 on the right child, as one pass (cross_level_turn_kernel: levels 15 -> 16 on 2^12 blocks with
    paired accesses, 17 -> 18 and 18 -> 19 on 2^13 blocks with one class per thread; integers at 2^16) -- every output
    vector bit-equal to the separate passes, every column through the identities."""
    from ringsnark_amd.device import to_host
    prm = P.preset("toy44")
    runs = {}
    for turn in (0, 1):
        _set_tuning(b"witness_level_turn", turn)
        try:
            dev, cs, asg, ds, w, names = _run(prm, m, zk, 0, int_arith=int_arith, want=("A_mid", "B_mid", "C_mid", "H"))
        finally:
            _set_tuning(b"witness_level_turn", 1)
        assert any(n.startswith("cross_level_turn_kernel") for n in names) == (turn == 1), names
        if turn:
            err, info = proof_check.check_all_columns(prm, cs, asg, {k: w[k] for k in ("A_mid", "B_mid", "C_mid", "H")}, tuple(ds),
                                                      seed=m % 983, Z=w["Z"])
            assert err is None and info["columns"] == prm.L * prm.N, err
        runs[turn] = {k: to_host(w[k]) for k in ("A_mid", "B_mid", "C_mid", "H")}
        del dev, asg, w
    for k in runs[0]:
        assert (runs[0][k] == runs[1][k]).all(), k


@pytest.mark.parametrize("m,zk,sub_log,ws", [(30000, True, 12, 6144), (65536, False, 12, 6144), (100000, True, 12, 6144), (50000, True, 13, 6144),
                                             (100000, False, 12, 64), (262144, True, 12, 64)])
def test_level_15_forward_stages_inside_the_tile_kernel(m, zk, sub_log, ws):
    """witness_tree_fwd (OFF by default: it removes a 9.8 ms pass of the headline proof and costs the tile kernel 16 ms; kept as a
    measured negative): the workgroup of a right 2^14 tile also runs the forward cross stages of level 15
    (three on 2^12 blocks, two on 2^13 blocks: witness_sub_log = 13) and writes that level's workspace, so the level's source
    pass is not launched.  M = 2^15 (level 15 is the last), 2^16, 2^17, 2^18; with workspace-sized sub-chunks (the tiles of
    all columns in one launch, the levels above per sub-chunk on slices of the workspace).  Bit-equal to the separate pass;
    every column through the identities."""
    from ringsnark_amd.device import to_host
    prm = P.preset("toy44")
    runs = {}
    want = ("A_mid", "B_mid", "C_mid", "H")
    for fwd in (0, 1):
        _set_tuning(b"witness_tree_fwd", fwd)
        _set_tuning(b"witness_sub_log", sub_log)
        _set_tuning(b"witness_big_ws_mib", ws)
        try:
            dev, cs, asg, ds, w, names = _run(prm, m, zk, 0, want=want)
        finally:
            _set_tuning(b"witness_tree_fwd", 0)
            _set_tuning(b"witness_sub_log", SUB_LOG_DEFAULT)
            _set_tuning(b"witness_big_ws_mib", 6144)
        rf = 3 if sub_log == 12 else 2
        assert (("tree_wide_kernel<14, %d>" % rf) in names) == (fwd == 1), names
        if fwd:
            err, info = proof_check.check_all_columns(prm, cs, asg, {k: w[k] for k in want}, tuple(ds), seed=m % 977, Z=w["Z"])
            assert err is None and info["columns"] == prm.L * prm.N, err
        runs[fwd] = {k: to_host(w[k]) for k in want}
        del dev, asg, w
    for k in want:
        assert (runs[0][k] == runs[1][k]).all(), k


@pytest.mark.parametrize("m,zk", [(30000, True), (50000, False), (65536, True), (100000, True), (262144, False)])
def test_the_turns_on_the_two_dimensional_block_convolutions(m, zk):
    """The same two fusions on the recipe primes' path (forced: witness_force_bc = 14): bc2_h_turn_kernel (the product's
    inverse transform across blocks + the forward one of its reversal, M <= 2^16) and bc2_level_turn_kernel (tree level l's
    sink + level l + 1's source, parents of at most 32 blocks) -- bit-equal to the separate passes and to nothing less than
    every column's identities."""
    from ringsnark_amd.device import to_host
    prm = P.preset("toy44")
    want = ("A_mid", "B_mid", "C_mid", "H")
    runs = {}
    for turn in (0, 1):
        _set_tuning(b"witness_h_turn", turn)
        _set_tuning(b"witness_level_turn", turn)
        try:
            dev, cs, asg, ds, w, names = _run(prm, m, zk, 14, want=want)
        finally:
            _set_tuning(b"witness_h_turn", 1)
            _set_tuning(b"witness_level_turn", 1)
        logM = (m - 1).bit_length()
        assert ("bc2_h_turn_kernel" in names) == (turn == 1 and logM <= 16), names
        assert ("bc2_level_turn_kernel" in names) == (turn == 1 and logM >= 16), names
        if turn:
            err, info = proof_check.check_all_columns(prm, cs, asg, {k: w[k] for k in want}, tuple(ds), seed=m % 971, Z=w["Z"])
            assert err is None and info["columns"] == prm.L * prm.N, err
        runs[turn] = {k: to_host(w[k]) for k in want}
        del dev, asg, w
    for k in want:
        assert (runs[0][k] == runs[1][k]).all(), k
