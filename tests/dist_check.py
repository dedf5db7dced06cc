"""One sharded proof on the REAL device backend against the one-process CPU oracle (test infrastructure).  Every rank of an
initialised torch.distributed world calls it; rank 0 learns the verdict.  Used by the device-backend tests of
tests/test_dist.py / tests/test_zz_rccl.py and by `bench.py --gpus N` (its `transport_check`: until a multi-GPU run has been
measured, every N > 1 bench run proves a small statement over the same process group and plan shape and compares it bit for
bit -- round-5 advisor finding)."""
import numpy as np

from oracle import oracle as O
from ringsnark_amd import dist as RD
from ringsnark_amd import params as P
from ringsnark_amd import r1cs as R
from tests import helpers as H


def rinocchio_key(ctx, m, n_aux):
    return dict(s_pows=ctx.random_enc(81, m + 1), alpha_s_pows=ctx.random_enc(82, m + 1), beta_prods=ctx.random_enc(83, n_aux),
                beta_rv_ts=ctx.random_enc(84), beta_rw_ts=ctx.random_enc(85), beta_ry_ts=ctx.random_enc(86))


def sharded_proof_matches_oracle(rank, world, dev_index, preset="toy", m=9, q_override=None, prover="groth16", zk=False):
    """groth16_prove_sharded / rinocchio_prove_sharded (ringsnark_amd/dist.py) on `world` ranks, rank r on device dev_index,
    for a wide synthetic R1CS of m constraints on `preset` (q_override: keep that many ring limbs): True on rank 0 iff the
    assembled proof equals the oracle's one-process proof bit for bit (and the EMPTY flags agree)."""
    from ringsnark_amd.device import Device, to_host
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
        pk = rinocchio_key(ctx_full, m, cs_full.n_aux)
    ds = [ctx_full.random_ring(60 + k) for k in range(3)] if zk else [None] * 3
    plan = RD.make_plan(world, rank, prm.L)
    tg = RD.groups_for(plan)
    prm_local = P.RingParams(prm.N, [prm.q[i] for i in plan.limbs], prm.N_enc, prm.Q)
    dev = Device(prm_local, dev_index)
    dcs = dev.r1cs(R.wide_r1cs(m, prm_local.q))
    pk_local = {}
    ranges = (RD.groth16_key_ranges if prover == "groth16" else RD.rinocchio_key_ranges)(plan, m, cs_full.n_aux)
    for k, v in pk.items():
        if v.ndim == 5:  # key vector: keep only this rank's limbs AND the term window it reads
            lo, hi = ranges[k]
            pk_local[k] = RD.TiledKey(dev.put(np.ascontiguousarray(v[lo:max(hi, lo + 1)][:, plan.limbs])), lo, hi, v.shape[0])
        else:
            pk_local[k] = dev.put(np.ascontiguousarray(v[plan.limbs]))
    dasg = dev.put(np.ascontiguousarray(asg[:, plan.limbs]))
    if prover == "groth16":
        got = RD.groth16_prove_sharded(RD.DeviceBackend(dev), plan, tg, dcs, pk_local, dasg, m, cs_full.n_inputs, cs_full.n_aux)
        got_empty = exp_empty = None
        exp = O.groth16_prove(ctx_full, H.oracle_cs(cs_full), pk, asg)[0] if rank == 0 else None
    else:
        dl = [None if d is None else dev.put(np.ascontiguousarray(d[plan.limbs])) for d in ds]
        got, got_empty = RD.rinocchio_prove_sharded(RD.DeviceBackend(dev), plan, tg, dcs, pk_local, dasg, m, cs_full.n_inputs,
                                                    cs_full.n_aux, *dl)
        exp, exp_empty = O.rinocchio_prove(ctx_full, H.oracle_cs(cs_full), pk, asg, *ds) if rank == 0 else (None, None)
    RD.release_buffers()
    if rank != 0:
        return None
    return bool((to_host(got) == exp).all()) and got_empty == exp_empty
