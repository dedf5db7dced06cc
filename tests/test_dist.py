"""Multi-rank prover (ringsnark_amd/dist.py) over gloo, world_size 2, on CPU.

The arithmetic backend is an oracle-backed stand-in (tests may use the oracle); what is under test
is the production sharding / collective code: limb split, term split + all-reduce + mod-Q
epilogue, all-gather assembly.  The result must equal the single-process oracle proof bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O
from ringsnark_amd import dist as RD
from ringsnark_amd import params as P
from ringsnark_amd import r1cs as R
from tests import helpers as H


class OracleBackend:
    def __init__(self, ctx):
        self.ctx = ctx
        self.N = ctx.N

    @staticmethod
    def _np(t):
        return t.contiguous().numpy().view(np.uint64)

    @staticmethod
    def _t(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int64))

    def witness_slots(self, cs, assignment, slot0, nslots, want, ds=(None, None, None)):
        """The oracle's O(m^2) map on the slot range only (every slot is an independent problem)."""
        asg = self._np(assignment)
        out = {k: np.zeros((cs.m + (1 if k == "H" else 0), self.ctx.L, nslots), dtype=np.uint64) for k in want}
        Z = np.zeros((self.ctx.L, cs.m + 1), dtype=np.uint64)
        ocs = H.oracle_cs(cs)
        sl = slice(slot0, slot0 + nslots)
        for limb in range(self.ctx.L):
            dl = [None if d is None else np.ascontiguousarray(self._np(d)[limb, sl]) for d in ds]
            w = O.witness_map(self.ctx.q[limb], ocs, limb, np.ascontiguousarray(asg[:, limb, sl]), *dl)
            for k in want:
                out[k][:, limb, :] = w[k]
            Z[limb] = w["Z"]
        res = {k: self._t(v) for k, v in out.items()}
        res["Z"] = Z
        return res

    def witness(self, cs, assignment, want, ds=(None, None, None), rows=None):
        w = self.witness_slots(cs, assignment, 0, self.ctx.N, want, ds)
        if rows is not None:
            for k, (lo, hi) in rows.items():
                w[k] = w[k][lo:hi]
        return w

    def msm(self, crs_list, vecs, n_groups, addends=None, want_used=False):
        outs, used = [], [0] * len(vecs)
        for ci, ks in enumerate(crs_list):
            crs = self._np(ks.tensor)
            if ks.window:  # tiled key: logical term t is stored element t % window
                crs = np.concatenate([crs] * (-(-ks.length // ks.window)))[:ks.length]
            row = []
            for g in range(n_groups):
                acc = np.zeros(self.ctx.enc_shape(), dtype=np.uint64)
                for vi, vec in enumerate(vecs):
                    v, kinds, gg = vec[:3]
                    if gg != g or v.shape[0] == 0:
                        continue
                    if len(vec) > 3 and vec[3]:  # slot-constant vector [rows][L]: the ring elements with the value in every slot
                        v = self._t(np.repeat(self._np(v)[:, :, None], self.ctx.N, axis=2))
                    ip, u = self.ctx.inner_product(crs[:v.shape[0]], self._np(v), kinds)
                    if ci == 0:
                        used[vi] = u
                    if u:
                        acc = self.ctx.enc_add(acc, ip)
                if addends is not None:
                    acc = self.ctx.enc_add(acc, self._np(addends[g]))
                row.append(acc)
            outs.append(np.stack(row))
        return self._t(np.stack(outs)), used

    def enc_add(self, a, b):
        sh = a.shape
        r = np.stack([self.ctx.enc_add(x, y) for x, y in zip(self._np(a).reshape((-1,) + self.ctx.enc_shape()),
                                                             self._np(b).reshape((-1,) + self.ctx.enc_shape()))])
        return self._t(r.reshape(sh))

    def enc_mul_ring(self, enc, ring):
        return self._t(self.ctx.enc_mul_ring(self._np(enc).reshape(self.ctx.enc_shape()).copy(), self._np(ring)))

    def enc_reduce(self, piece):
        a = self._np(piece).copy()
        for j, Q in enumerate(self.ctx.Q):
            a[..., j, :] %= np.uint64(Q)
        return self._t(a)

    def scalar_rows(self, z):
        return self._t(np.ascontiguousarray(z.T))  # [rows][L]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rinocchio_key(ctx, m, n_aux, zk):
    return dict(s_pows=ctx.random_enc(81, m + 1), alpha_s_pows=ctx.random_enc(82, m + 1), beta_prods=ctx.random_enc(83, n_aux),
                beta_rv_ts=ctx.random_enc(84), beta_rw_ts=ctx.random_enc(85), beta_ry_ts=ctx.random_enc(86))


def _worker(rank, world, port, preset, m, q_override, tmp, prover="groth16", zk=False, split="slots", chunk_bytes=None, relay=True,
            scalar_wires=False):
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    RD.WITNESS_SPLIT = split  # how the ranks of one limb group share the witness map (dist.py)
    RD.RELAY = relay
    if chunk_bytes:
        RD.SLOT_CHUNK_BYTES = chunk_bytes  # tiny sub-ranges of slots: several pipelined exchange steps at toy scale
    RD.STATS.reset(True)  # transport statistics (what bench.py --gpus N prints): written beside the verdict
    try:
        if preset == "toy4":  # four ring limbs (the headline's limb count) at toy scale
            prm = P.make_params(32, [30, 30, 30, 30], 64, [40, 40, 41], ring_factor=1 << 12, name="toy4")
        else:
            prm = P.preset(preset)
        if q_override:
            prm = P.RingParams(prm.N, prm.q[:q_override], prm.N_enc, prm.Q)
        ctx_full = H.oracle_ctx(prm)
        cs_full = R.wide_r1cs(m, prm.q)
        asg = H.make_assignment(ctx_full, cs_full)
        if prover == "groth16":
            pk = dict(s_pows=ctx_full.random_enc(71, m + 1), delta_ts=ctx_full.random_enc(72, m + 1),
                      delta_mid=ctx_full.random_enc(73, cs_full.n_aux), alpha=ctx_full.random_enc(74), beta=ctx_full.random_enc(75))
        else:
            pk = _rinocchio_key(ctx_full, m, cs_full.n_aux, zk)
        ds = [ctx_full.random_ring(60 + k) for k in range(3)] if zk else [None] * 3
        kinds = aux_kinds = None
        if scalar_wires:  # auxiliary wires held as RingElem Scalars: 1 (passes the key element through), 0 (skipped), 7 (flattened)
            asg = asg.copy()
            a0 = cs_full.n_inputs
            kinds = np.zeros(cs_full.n_vars, dtype=np.uint8)
            for k, (val, kind) in enumerate(((1, O.KIND_ONE), (0, O.KIND_POLY), (7, O.KIND_POLY), (1, O.KIND_ONE))):
                asg[a0 + 2 * k] = val
                kinds[a0 + 2 * k] = kind
            aux_kinds = kinds[a0:]
        plan = RD.make_plan(world, rank, prm.L)
        tg = RD.groups_for(plan)
        prm_local = P.RingParams(prm.N, [prm.q[i] for i in plan.limbs], prm.N_enc, prm.Q)
        backend = OracleBackend(H.oracle_ctx(prm_local))
        cs_local = R.wide_r1cs(m, prm_local.q)
        t = OracleBackend._t
        pk_local = {k: t(v[:, plan.limbs] if v.ndim == 5 else v[plan.limbs]) for k, v in pk.items()}
        asg_local = t(asg[:, plan.limbs])
        if prover == "groth16":
            got = RD.groth16_prove_sharded(backend, plan, tg, cs_local, pk_local, asg_local, m, cs_full.n_inputs, cs_full.n_aux,
                                           aux_kinds=aux_kinds)
            exp, exp_empty, got_empty = O.groth16_prove(ctx_full, H.oracle_cs(cs_full), pk, asg, kinds)[0], None, None
        else:
            dl = [None if d is None else t(d[plan.limbs]) for d in ds]
            got, got_empty = RD.rinocchio_prove_sharded(backend, plan, tg, cs_local, pk_local, asg_local, m, cs_full.n_inputs,
                                                        cs_full.n_aux, *dl, aux_kinds=aux_kinds)
            exp, exp_empty = O.rinocchio_prove(ctx_full, H.oracle_cs(cs_full), pk, asg, *ds, kinds=kinds)
        if rank == 0:
            ok = bool((OracleBackend._np(got) == exp).all()) and got_empty == exp_empty
            open(tmp, "w").write("ok" if ok else "mismatch")
            import json
            pooled = len(RD._POOL.bufs)
            RD.release_buffers()
            open(tmp + ".stats", "w").write(json.dumps({"phases": RD.STATS.read(), "pooled_before_release": pooled, "pooled_after": len(RD._POOL.bufs)}))
    finally:
        dist.destroy_process_group()


def test_transport_statistics_and_buffer_release(tmp_path):
    """What `bench.py --gpus N` prints per collective (dist.STATS) and the lifetime of the pooled re-shard buffers (round-5
    advice: release_buffers): one limb on two ranks -> one slot -> term exchange, one all-reduce, one all-gather."""
    import json
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), "toy", 7, 1, out, "groth16", False, "slots"), nprocs=2, join=True)
    assert open(out).read() == "ok"
    st = json.load(open(out + ".stats"))
    ph = st["phases"]
    assert set(ph) == {"slot_to_term_exchange", "all_reduce_partial_sums", "all_gather_proof"}, ph
    assert all(v["calls"] >= 1 and v["bytes"] > 0 and v["ms"] >= 0 for v in ph.values()), ph
    # a rank of a 2-rank group sends and receives half of its four compact vectors' rows; the proof is 3 encoding elements of one limb
    prm = P.preset("toy")
    assert ph["all_gather_proof"]["bytes"] == 3 * 2 * prm.K * prm.N_enc * 8 and ph["all_reduce_partial_sums"]["bytes"] == ph["all_gather_proof"]["bytes"]
    assert st["pooled_before_release"] > 0 and st["pooled_after"] == 0


@pytest.mark.parametrize("q_override,split,desc", [
    (None, "slots", "limb split: 2 limbs over 2 ranks"),
    (1, "slots", "one limb on 2 ranks: slot-sharded witness map, row exchange, term-sharded MSM, all-reduce"),
    (1, "replicate", "one limb on 2 ranks: witness map on both, term-sharded MSM, all-reduce (the default plan)")])
def test_sharded_groth16_equals_single_process(tmp_path, q_override, split, desc):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), "toy", 7, q_override, out, "groth16", False, split), nprocs=2, join=True)
    assert open(out).read() == "ok", desc


@pytest.mark.parametrize("q_override,zk", [(None, False), (1, True), (1, False)])
def test_sharded_rinocchio_equals_single_process(tmp_path, q_override, zk):
    """rinocchio.tcc:75-190 sharded (BASELINE.json configs[3]): limb split, and slots / terms split inside a limb."""
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), "toy", 6, q_override, out, "rinocchio", zk), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_sharded_groth16_three_ranks_uneven_ranges(tmp_path):
    """3 ranks on one limb: uneven slot blocks (N = 32 -> 12, 12, 8) and term ranges."""
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(3, _free_port(), "toy", 8, 1, out), nprocs=3, join=True)
    assert open(out).read() == "ok"


@pytest.mark.parametrize("preset,prover,zk,chunk,relay,desc", [
    ("toy4", "groth16", False, None, True, "the headline's plan: 4 limb groups x 2 term shards; every rank relays for the other three pairs"),
    ("toy4", "groth16", False, 2048, True, "the same with the slot range cut into sub-ranges: pipelined exchange steps"),
    ("toy", "rinocchio", True, 4096, True, "configs[3]'s plan: 2 limb groups x 4 term shards, sub-ranges, relays through the other group"),
    ("toy", "groth16", False, None, False, "2 x 4 without relays (direct links only)"),
    ("toyC3", "groth16", False, None, False, "the headline's plan on the headline's ring primes (the reference recipe's: preset C3 on a 32-slot ring), direct links"),
])
def test_eight_ranks_relayed_slot_reshard(tmp_path, preset, prover, zk, chunk, relay, desc):
    """N = 8 over gloo: the slot -> term re-shard of ringsnark_amd/dist.py with its two-hop relays through the ranks of the
    OTHER limb groups (xGMI is point to point: a pair that shares a limb owns one of its seven links), in sub-ranges of
    slots.  The sharded proof equals the one-process proof bit for bit."""
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(8, _free_port(), preset, 7, None, out, prover, zk, "slots", chunk, relay), nprocs=8, join=True)
    assert open(out).read() == "ok", desc


@pytest.mark.parametrize("prover,q_override,zk", [("groth16", 1, False), ("groth16", None, False), ("rinocchio", 1, True)])
def test_sharded_provers_pass_scalar_one_wires_through(tmp_path, prover, q_override, zk):
    """Scalar-1 auxiliary wires (seal_ring.tcc:525-527) across the term split (the kinds are sliced with the terms) and the
    limb split (a property of the wire: the same on every limb): the sharded proof equals the oracle's proof WITH kinds."""
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), "toy", 9, q_override, out, prover, zk, "slots", None, True, True), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_relay_parts_cover_every_message():
    """the split of a message into direct halves and relayed parts: sizes add up, the balance is the one DESIGN.md derives
    (relays are opt-in, RINGSNARK_RELAY=1: switched on here for the arithmetic)"""
    saved = RD.RELAY  # whatever RINGSNARK_RELAY made it (round-5 advice: do not assert the environment)
    try:
        RD.RELAY = False
        assert RD._split_parts(800, 2, 6) == (800, 0, 0)  # relays off (the default): direct links of the limb group only
        RD.RELAY = True
        _relay_parts()
    finally:
        RD.RELAY = saved


def _relay_parts():
    for numel in (1, 7, 1000, 12345678):
        for g, n_rel in ((2, 6), (4, 4), (2, 0), (3, 5)):
            d1, d2, r = RD._split_parts(numel, g, n_rel)
            assert d1 + d2 + n_rel * r == numel and min(d1, d2, r) >= 0
            if n_rel and numel > 1000:
                assert abs((d1 + d2) - 2 * (g - 1) * r) <= 2 * (g - 1) + n_rel  # d = 2 (g - 1) r up to rounding
    assert RD._split_parts(800, 2, 6) == (100, 100, 100)  # the headline at N = 8: a quarter direct, an eighth per relay
    assert RD._sub_ranges(8, 8, 3) == [(8, 4), (12, 4), (16, 0)]


def test_shard_plans():
    for world, L, exp in [(1, 4, (1, 1)), (2, 4, (2, 1)), (4, 4, (4, 1)), (8, 4, (4, 2)), (8, 6, (2, 4)), (3, 4, (1, 3))]:
        plans = [RD.make_plan(world, r, L) for r in range(world)]
        assert (plans[0].limb_groups, plans[0].term_shards) == exp
        # the slot blocks of a limb group tile [0, N), even aligned
        for N in (32, 8192, 6):
            blocks = [plans[0].slot_range(N, s) for s in range(plans[0].term_shards)]
            assert blocks[0][0] == 0 and sum(b[1] for b in blocks) == N
            assert all(b[0] % 2 == 0 and b[1] % 2 == 0 for b in blocks)
            assert all(x[0] + x[1] == y[0] for x, y in zip(blocks, blocks[1:]) if y[1])
        # every (limb, term) cell is covered exactly once
        T = 37
        cover = np.zeros((L, T), dtype=int)
        for p in plans:
            lo, hi = p.term_range(T)
            for i in p.limbs:
                cover[i, lo:hi] += 1
        assert (cover == 1).all()


def _gpu_worker(rank, world, port, m, q_override, tmp, prover="groth16", zk=False, split="slots", transport="gloo", relay=False,
                preset="toy"):
    """Every rank drives the REAL device backend.  transport "gloo": all ranks on cuda:0, gloo moves the collectives
    (one-GPU boxes); "nccl": rank r on cuda:r, RCCL over xGMI -- the production transport (needs >= world devices)."""
    if transport == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world,
                                device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    dev_index = rank if transport == "nccl" else 0
    RD.WITNESS_SPLIT = split
    RD.RELAY = relay
    try:
        from tests.dist_check import sharded_proof_matches_oracle
        ok = sharded_proof_matches_oracle(rank, world, dev_index, preset, m, q_override, prover, zk)
        if rank == 0:
            open(tmp, "w").write("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("q_override,m,split", [(None, 9, "replicate"), (1, 9, "slots"), (1, 8, "slots"), (1, 9, "replicate")])
def test_sharded_groth16_on_device_backend(tmp_path, q_override, m, split):
    """m even and odd: the windows of s_pows (m of m+1 entries); both ways of sharing the witness map inside a limb group."""
    out = str(tmp_path / "result.txt")
    mp.spawn(_gpu_worker, args=(2, _free_port(), m, q_override, out, "groth16", False, split), nprocs=2, join=True)
    assert open(out).read() == "ok"


@pytest.mark.gpu
@pytest.mark.parametrize("q_override,m,zk,split", [(None, 7, True, "slots"), (1, 8, True, "slots"), (1, 9, False, "slots"), (1, 8, True, "replicate")])
def test_sharded_rinocchio_on_device_backend(tmp_path, q_override, m, zk, split):
    """The sharded Rinocchio prover (limb split; slot-sharded witness map + row exchange + term-sharded MSM) with
    both ranks on the real device backend, against the single-process oracle."""
    out = str(tmp_path / "result.txt")
    mp.spawn(_gpu_worker, args=(2, _free_port(), m, q_override, out, "rinocchio", zk, split), nprocs=2, join=True)
    assert open(out).read() == "ok"


def _spawn_with_deadline(fn, args, nprocs, seconds):
    """mp.spawn that cannot hang the suite: the ranks are killed and the test fails when they have not finished in time
    (a collective that never completes -- a transport this code has not met yet -- must cost one test, not the run)."""
    import time
    ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    deadline = time.time() + seconds
    while not ctx.join(timeout=5):
        if time.time() > deadline:
            for p_ in ctx.processes:
                if p_.is_alive():
                    p_.terminate()
            for p_ in ctx.processes:
                p_.join(10)
                if p_.is_alive():
                    p_.kill()
            pytest.fail("the %d ranks did not finish within %d s (killed)" % (nprocs, seconds))


def test_limb_only_plans_take_the_fused_device_prover():
    """A rank that owns whole limbs (N <= L) must run rs_groth16_prove on the key ranges bench.py / groth16_key_ranges
    give it: s_pows is stored on m of its m + 1 entries, which the fused prover never reads beyond (the path
    was dead because it demanded hi == T)."""
    class Dcs:
        m, n_vars, n_inputs = 9, 11, 2

    class FakeDev:
        calls = []

        def groth16_prove(self, dcs, pk, asg, want_empty, window, kinds=None):
            self.calls.append((sorted(pk), {k: len(v) for k, v in pk.items() if k in ("s_pows", "delta_ts", "delta_mid")}, window))
            return ["proof"]

    class Backend(RD.DeviceBackend):
        def __init__(self, dev):
            self.dev = dev

    m, n_aux = Dcs.m, Dcs.n_vars - Dcs.n_inputs
    for world, L in ((1, 2), (2, 2), (2, 4), (4, 4)):
        for W in (None, 4):  # whole ranges, and windows of 4 elements
            plan = RD.make_plan(world, 0, L)
            assert plan.term_shards == 1
            rg = RD.groth16_key_ranges(plan, m, n_aux)
            T = {"s_pows": m + 1, "delta_ts": m + 1, "delta_mid": n_aux}
            pk = {k: RD.TiledKey(list(range(min(hi - lo, W or hi - lo))), lo, hi, T[k]) for k, (lo, hi) in rg.items()}
            pk.update(alpha="a", beta="b")
            dev = FakeDev()
            dev.calls = []
            assert Backend(dev).groth16_prove_local(Dcs, pk, "asg") == "proof"
            names, lens, window = dev.calls[0]
            assert names == ["alpha", "beta", "delta_mid", "delta_ts", "s_pows"] and window == (W or 0)
            assert lens == ({"s_pows": m, "delta_ts": m + 1, "delta_mid": n_aux} if W is None else {"s_pows": 4, "delta_ts": 4, "delta_mid": 4})
    # a term-sharded rank (N > L) does not hold whole vectors: piecewise plan
    plan = RD.make_plan(4, 3, 2)
    rg = RD.groth16_key_ranges(plan, m, n_aux)
    pk = {k: RD.TiledKey(list(range(max(1, hi - lo))), lo, hi, m + 1) for k, (lo, hi) in rg.items()}
    assert RD.fused_groth16_key(dict(pk, alpha="a", beta="b"), m, n_aux) is None
    # mixed windows: refused
    pk = {"s_pows": RD.TiledKey(list(range(4)), 0, m, m + 1), "delta_ts": RD.TiledKey(list(range(m + 1)), 0, m + 1, m + 1),
          "delta_mid": RD.TiledKey(list(range(n_aux)), 0, n_aux, n_aux), "alpha": "a", "beta": "b"}
    assert RD.fused_groth16_key(pk, m, n_aux) is None


def test_late_shards_get_empty_not_negative_ranges():
    """rng_mid of rinocchio_prove_sharded and slot_range for shards beyond the data."""
    plan = RD.make_plan(8, 7, 1)  # 8 ranks on one limb
    for m in (1, 3, 6, 7, 8, 9):
        for s in range(8):
            lo, hi = plan.term_range(m + 1, s)
            mlo, mhi = min(lo, m), max(min(lo, m), min(hi, m))
            assert 0 <= mlo <= mhi <= m
    assert plan.slot_range(6, 7) == (6, 0)


def test_key_windows_cover_every_slice_the_sharded_prover_takes():
    """bench.py allocates only a window of each key vector per rank; the windows must be exactly the
    ranges groth16_prove_sharded reads (s_pows on m of its m+1 entries), for every plan shape."""
    from ringsnark_amd import dist as RD
    for world, L in ((1, 4), (2, 4), (4, 4), (8, 4), (4, 2), (8, 2), (3, 4), (6, 4)):
        for m in (1, 2, 7, 1024, 8192 * world, 8192 * world + 1):
            n_aux = m
            covered = {k: [] for k in ("s_pows", "delta_ts", "delta_mid")}
            for rank in range(world):
                plan = RD.make_plan(world, rank, L)
                rg = RD.groth16_key_ranges(plan, m, n_aux)
                T = {"s_pows": m + 1, "delta_ts": m + 1, "delta_mid": n_aux}
                for k, (lo, hi) in rg.items():
                    w = RD.TiledKey(list(range(lo, max(hi, lo + 1))), lo, hi, T[k])
                    assert list(w[lo:hi].tensor) == list(range(lo, hi))  # the slice the prover takes is inside the window
                    if plan.limb_group == 0:
                        covered[k].append((lo, hi))
            # the term shards of one limb group tile [0, used) without gaps or overlaps
            for k, used in (("s_pows", m), ("delta_ts", m + 1), ("delta_mid", n_aux)):
                pieces = sorted(p for p in covered[k] if p[0] < p[1])
                assert pieces[0][0] == 0 and pieces[-1][1] == used
                assert all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))


def _sleeper(rank, seconds):
    import time
    time.sleep(seconds)


def test_spawn_with_deadline_kills_stuck_ranks():
    """the guard of the RCCL tests: ranks that do not finish are killed and the test fails -- it does not hang the suite"""
    import time
    t0 = time.time()
    with pytest.raises(pytest.fail.Exception):
        _spawn_with_deadline(_sleeper, (120,), 2, 3)
    assert time.time() - t0 < 60
    _spawn_with_deadline(_sleeper, (0,), 2, 60)  # ranks that do finish: returns
