"""Multi-GPU ringGroth16 prover: one process per GPU, torch.distributed (RCCL over xGMI).

Sharding:
  * RNS limbs first.  A ring limb i is an independent problem end to end -- its own N witness-map
    columns over F_{q_i} and its own BGV encoding context (ciphertexts [2][K][N_enc] with plain
    modulus q_i) -- so a rank simply builds its context over a SUBSET of the ring primes and holds
    the matching slice [:, limbs] of the CRS and of the assignment.  No exchange is needed until
    the proof is assembled.
  * Constraints (MSM terms) second, when there are more ranks than limbs: the ranks sharing a limb
    split the term range; their partial encoding sums are combined by ONE all-reduce(SUM) of
    3 * L_local encoding elements (residues < 2^50, so the integer sum of <= 2^13 partials cannot
    overflow int64) followed by a reduction mod Q_j.  The witness map of a shared limb is
    replicated inside the pair (it is column-parallel, not term-parallel).
  * The proof {A, B, C} is assembled by an all-gather over the limb axis.

`backend` abstracts the arithmetic (Device in production; the CPU tests drive the same code over
gloo with an oracle-backed stand-in to check the sharding and the collectives).
"""
import math
from dataclasses import dataclass
from typing import List

import torch
import torch.distributed as dist


@dataclass
class ShardPlan:
    world: int
    rank: int
    L: int
    limb_groups: int  # G_l
    term_shards: int  # G_t
    limbs: List[int]  # ring limbs owned by this rank
    term_shard: int
    limb_group: int

    def term_range(self, T):
        per = (T + self.term_shards - 1) // self.term_shards
        lo = min(T, self.term_shard * per)
        return lo, min(T, lo + per)


class TermWindow:
    """A key vector of logical length T of which this rank stores only terms [lo, hi)."""

    def __init__(self, store, lo, hi, T):
        self.store, self.lo, self.hi, self.T = store, lo, hi, T

    def __getitem__(self, sl):
        assert isinstance(sl, slice) and sl.step is None
        a, b = sl.start or 0, self.T if sl.stop is None else sl.stop
        assert self.lo <= a <= b <= self.hi, "term range outside this rank's window"
        return self.store[a - self.lo:b - self.lo]


def make_plan(world, rank, L) -> ShardPlan:
    g_l = math.gcd(world, L)
    g_t = world // g_l
    limb_group, term_shard = rank % g_l, rank // g_l
    limbs = [i for i in range(L) if i % g_l == limb_group]
    return ShardPlan(world, rank, L, g_l, g_t, limbs, term_shard, limb_group)


def groups_for(plan: ShardPlan):
    """Process groups: ranks sharing a limb group (term all-reduce).  Every rank must call this
    (dist.new_group is collective)."""
    term_groups = []
    for lg in range(plan.limb_groups):
        ranks = [lg + plan.limb_groups * s for s in range(plan.term_shards)]
        term_groups.append(dist.new_group(ranks) if plan.term_shards > 1 else None)
    return term_groups[plan.limb_group]


def groth16_key_ranges(plan: ShardPlan, m, n_aux):
    """Term range [lo, hi) of every key vector that THIS rank reads in groth16_prove_sharded:
    s_pows is used on its first m entries (A and B have m coefficients, groth16.tcc:89-101),
    delta_ts on all m + 1 (H), delta_mid on n_aux.  A rank that stores only windows of the key
    (bench.py) must allocate exactly these."""
    return {"s_pows": plan.term_range(m), "delta_ts": plan.term_range(m + 1), "delta_mid": plan.term_range(n_aux)}


def groth16_prove_sharded(backend, plan: ShardPlan, term_group, cs_local, pk_local, assignment_local, m, n_inputs, n_aux):
    """groth16::prover (zk_proof_systems/groth16/groth16.tcc:70-115) on this rank's shard.

    *_local hold only this rank's limbs.  Returns the full proof [3][L][2][K][N_enc] (int64) on
    every rank."""
    w = backend.witness(cs_local, assignment_local, want=("A_io", "A_mid", "B_io", "B_mid", "H"))
    lead = plan.term_shard == 0  # exactly one shard per limb group adds alpha / beta
    ranges = groth16_key_ranges(plan, m, n_aux)
    lo, hi = ranges["s_pows"]
    ab = backend.msm([pk_local["s_pows"][lo:hi]],
                     [(w["A_io"][lo:hi], 0), (w["A_mid"][lo:hi], 0), (w["B_io"][lo:hi], 1), (w["B_mid"][lo:hi], 1)], 2,
                     addends=[pk_local["alpha"], pk_local["beta"]] if lead else None)
    lo, hi = ranges["delta_ts"]
    c = backend.msm([pk_local["delta_ts"][lo:hi]], [(w["H"][lo:hi], 0)], 1)
    if n_aux:
        lo, hi = ranges["delta_mid"]
        aux = assignment_local[n_inputs:]
        c2 = backend.msm([pk_local["delta_mid"][lo:hi]], [(aux[lo:hi], 0)], 1)
        c = backend.enc_add(c, c2)
    piece = torch.cat([ab.reshape((2,) + tuple(ab.shape[-4:])), c.reshape((1,) + tuple(c.shape[-4:]))], dim=0).contiguous()
    if plan.term_shards > 1:
        dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=term_group)
        piece = backend.enc_reduce(piece)
    if plan.world == 1:
        return piece
    # all-gather over every rank, then keep one copy per limb (shard 0 of each limb group)
    pieces = [torch.empty_like(piece) for _ in range(plan.world)]
    dist.all_gather(pieces, piece)
    full = torch.empty((3, plan.L) + tuple(piece.shape[2:]), dtype=piece.dtype, device=piece.device)
    for lg in range(plan.limb_groups):
        limbs = [i for i in range(plan.L) if i % plan.limb_groups == lg]
        full[:, limbs] = pieces[lg]  # rank lg is term shard 0 of limb group lg
    return full


class DeviceBackend:
    """Production backend: ringsnark_amd.device.Device over this rank's limb subset."""

    def __init__(self, dev):
        self.dev = dev

    def witness(self, dcs, assignment, want):
        return self.dev.witness_map(dcs, assignment, want=want)

    def msm(self, crs_list, vecs, n_groups, addends=None):
        out, _ = self.dev.msm(crs_list, [(v.contiguous(), None, g) for v, g in vecs], n_groups)
        out = out[0]
        if addends is not None:
            out = torch.stack([self.dev.enc_add(out[g], addends[g]) for g in range(n_groups)])
        return out

    def enc_add(self, a, b):
        return self.dev.enc_add(a.contiguous(), b.contiguous())

    def enc_reduce(self, piece):
        return self.dev.enc_reduce(piece)
