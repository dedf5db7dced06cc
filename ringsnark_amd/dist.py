"""Multi-GPU provers: one process per GPU, torch.distributed (RCCL over xGMI).

Sharding:
  * RNS limbs first.  A ring limb i is an independent problem end to end -- its own N witness-map
    columns over F_{q_i} and its own BGV encoding context (ciphertexts [2][K][N_enc] with plain
    modulus q_i) -- so a rank builds its context over a SUBSET of the ring primes and holds the
    matching slice [:, limbs] of the key and of the assignment.  No exchange until the proof is
    assembled.
  * When there are more ranks than limbs, the G_t ranks sharing a limb set ("limb group") split
      - the INNER PRODUCTS by terms (constraints): each rank multiplies its term range of the key,
        and the partial encoding sums are combined by ONE all-reduce(SUM) (residues < 2^50, so integer
        sums of <= 2^13 partials cannot overflow int64) followed by a reduction mod Q_j;
      - the WITNESS MAP not at all by default (every rank of the group runs it whole and keeps the rows of its
        term range: WITNESS_SPLIT = "replicate"), or by NTT slots (it is column parallel: rank s takes slots
        [s N/G_t, (s+1) N/G_t) of every limb it owns, rs_witness_map_slots) followed by ONE pairwise exchange
        (point-to-point sends over xGMI, issued as one batch) that turns the slot-sharded coefficient vectors
        into term-sharded ones -- see WITNESS_SPLIT below for the trade.
  * The proof is assembled by an all-gather over the limb axis.

`backend` abstracts the arithmetic (DeviceBackend in production; the CPU tests drive the same code over
gloo with an oracle-backed stand-in to check the sharding and the collectives).
"""
import math
import os
from collections import namedtuple
from dataclasses import dataclass
from typing import List

import numpy as np
import torch
import torch.distributed as dist

RS_KIND_POLY, RS_KIND_ONE = 0, 2


@dataclass
class ShardPlan:
    world: int
    rank: int
    L: int
    limb_groups: int  # G_l
    term_shards: int  # G_t: ranks per limb group (they split slots for the witness map, terms for the MSM)
    limbs: List[int]  # ring limbs owned by this rank
    term_shard: int
    limb_group: int

    def term_range(self, T, shard=None):
        s = self.term_shard if shard is None else shard
        per = (T + self.term_shards - 1) // self.term_shards
        lo = min(T, s * per)
        return lo, min(T, lo + per)

    def slot_range(self, N, shard=None):
        """(slot0, nslots) of the witness-map share of a shard: contiguous, even-aligned blocks."""
        s = self.term_shard if shard is None else shard
        per = -(-N // self.term_shards)
        per += per & 1
        lo = min(N, s * per)
        return lo, min(N, lo + per) - lo

    def group_ranks(self):
        """Global ranks of this rank's limb group, by shard index."""
        return [self.limb_group + self.limb_groups * s for s in range(self.term_shards)]


# A slice of a key vector handed to backend.msm: `tensor` holds the stored elements, `length` logical terms are
# read, logical term t from tensor[t % window] when window != 0 (tiled key, include/ringsnark_amd.h crs_window).
KeySlice = namedtuple("KeySlice", "tensor length window")


class TiledKey:
    """A key vector of logical length T of which this rank reads terms [lo, hi) only.  `store` holds either
    exactly those hi - lo elements, or fewer ("window"): then logical term t lives at store[(t - lo) % len(store)]
    (tiled synthetic key for statements whose key exceeds HBM)."""

    def __init__(self, store, lo, hi, T):
        self.store, self.lo, self.hi, self.T = store, lo, hi, T
        n = len(store)
        assert n >= 1 and (n >= hi - lo or n & (n - 1) == 0), "a window must be a power of two"
        self.window = n if n < hi - lo else 0

    def __getitem__(self, sl):
        assert isinstance(sl, slice) and sl.step is None
        a, b = sl.start or 0, self.T if sl.stop is None else sl.stop
        assert self.lo <= a <= b <= self.hi, "term range outside this rank's window"
        if self.window:
            assert a == self.lo, "a tiled key is read from the start of the rank's range"
            return KeySlice(self.store, b - a, self.window)
        return KeySlice(self.store[a - self.lo:b - self.lo], b - a, 0)


TermWindow = TiledKey  # older name


def key_slice(v, a, b):
    """v: TiledKey or a plain tensor holding the whole vector."""
    if isinstance(v, TiledKey):
        return v[a:b]
    return KeySlice(v[a:b], b - a, 0)


def make_plan(world, rank, L) -> ShardPlan:
    g_l = math.gcd(world, L)
    g_t = world // g_l
    limb_group, term_shard = rank % g_l, rank // g_l
    limbs = [i for i in range(L) if i % g_l == limb_group]
    return ShardPlan(world, rank, L, g_l, g_t, limbs, term_shard, limb_group)


def groups_for(plan: ShardPlan):
    """Process group of this rank's limb group (the ranks that exchange witness rows and all-reduce partial
    sums).  Every rank must call this (dist.new_group is collective)."""
    mine = None
    for lg in range(plan.limb_groups):
        ranks = [lg + plan.limb_groups * s for s in range(plan.term_shards)]
        g = dist.new_group(ranks) if plan.term_shards > 1 else None
        if lg == plan.limb_group:
            mine = g
    return mine


def groth16_key_ranges(plan: ShardPlan, m, n_aux):
    """Term range [lo, hi) of every key vector that THIS rank reads in groth16_prove_sharded:
    s_pows is used on its first m entries (A and B have m coefficients, groth16.tcc:89-101),
    delta_ts on all m + 1 (H), delta_mid on n_aux.  A rank that stores only windows of the key
    (bench.py) must allocate exactly these."""
    return {"s_pows": plan.term_range(m), "delta_ts": plan.term_range(m + 1), "delta_mid": plan.term_range(n_aux)}


def rinocchio_key_ranges(plan: ShardPlan, m, n_aux):
    """rinocchio.tcc:106-163: s_pows / alpha_s_pows on m + 1 entries (h and z; the *_mid vectors use the first
    m of them), beta_prods on n_aux."""
    return {"s_pows": plan.term_range(m + 1), "alpha_s_pows": plan.term_range(m + 1), "beta_prods": plan.term_range(n_aux)}


# ---------------------------------------------------------------------------------------------------
# slot-sharded witness map -> term-sharded coefficient vectors
# ---------------------------------------------------------------------------------------------------
def _p2p_start(ops_send, ops_recv, group):
    """Launch one batch of point-to-point transfers inside `group`: (tensor, global peer rank) lists; returns a
    function that waits for it.  RCCL runs the batch as one grouped exchange over the direct xGMI links, on its own
    stream: whatever the caller launches before waiting overlaps the transfer.  gloo (CPU tests, single-GPU
    rehearsals) moves host tensors, so device tensors are staged through the host there."""
    if not ops_send and not ops_recv:
        return lambda: None
    stage = dist.get_backend(group) == "gloo" and any(t.is_cuda for t, _ in ops_send + ops_recv)
    if stage:
        send_h = [(t.cpu(), p) for t, p in ops_send]
        recv_h = [(torch.empty(t.shape, dtype=t.dtype), p) for t, p in ops_recv]
    else:
        send_h, recv_h = ops_send, ops_recv
    ops = [dist.P2POp(dist.isend, t.contiguous(), p, group) for t, p in send_h] + [dist.P2POp(dist.irecv, t, p, group) for t, p in recv_h]
    works = dist.batch_isend_irecv(ops)

    def wait():
        for r in works:
            r.wait()
        if stage:
            for (dst, _), (src, _) in zip(ops_recv, recv_h):
                dst.copy_(src)
    return wait


def _p2p(ops_send, ops_recv, group):
    _p2p_start(ops_send, ops_recv, group)()


# How the ranks of one limb group (N > L: two ranks per limb at N = 8) share the witness map:
#   "replicate" (default)  each runs it whole and keeps its term range: no exchange.  Per rank at the headline and N = 8:
#                          134 ms of witness map + 24 ms of inner products.
#   "slots"                each maps half the NTT slots (67 ms), then one batch of point-to-point transfers re-shards the
#                          five coefficient vectors from slots to terms: 5.4 GiB per rank each way over ONE xGMI link
#                          (>= 75 ms at the link's 76.8 GB/s per direction) -- it pays only if the exchange sustains
#                          more than ~85 GB/s, which no measurement supports yet.
WITNESS_SPLIT = os.environ.get("RINGSNARK_WITNESS_SPLIT", "replicate")
if WITNESS_SPLIT not in ("replicate", "slots"):
    raise ValueError("RINGSNARK_WITNESS_SPLIT must be 'replicate' or 'slots', not %r" % WITNESS_SPLIT)


def sharded_witness(backend, plan: ShardPlan, group, cs_local, assignment_local, want, ranges, ds=(None, None, None), defer=False):
    """The witness map of this rank's limbs, returned TERM-sharded: {k: rows [lo_k, hi_k) of vector k, all N
    slots}, plus "Z" (host array [L_local][m+1]).  ranges[k] = function shard -> (lo, hi) of vector k.
    With one rank per limb group this is the plain witness map; otherwise every rank maps its slot range
    (rs_witness_map_slots) and one batch of point-to-point transfers re-shards slots -> terms.
    defer=True: returns (out, finish); the vectors are complete only after finish() -- work that does not read them
    (the inner product over the auxiliary inputs) goes in between and overlaps the exchange."""
    if plan.term_shards == 1 or WITNESS_SPLIT == "replicate":
        # every rank of the limb group runs the whole witness map of its limbs and keeps the rows of its term range
        # (only those rows are written: five full-length vectors of a three-limb configs[3] rank would be 480 GiB)
        me = plan.term_shard
        out = backend.witness(cs_local, assignment_local, want, ds, rows={k: ranges[k](me) for k in want} if plan.term_shards > 1 else None)
        return (out, lambda: None) if defer else out
    N = backend.N
    s0, ns = plan.slot_range(N)
    if ns < 2:  # rs_witness_map_slots takes even, non-empty slot ranges: every shard of the group needs a share
        raise ValueError("slot split of N = %d slots over %d ranks leaves rank %d without a share; use "
                         "RINGSNARK_WITNESS_SPLIT=replicate or fewer ranks per limb" % (N, plan.term_shards, plan.rank))
    wc = backend.witness_slots(cs_local, assignment_local, s0, ns, want, ds)  # compact [rows][L][ns]
    peers = plan.group_ranks()
    me = plan.term_shard
    out, sends, recvs, pending = {"Z": wc["Z"]}, [], [], []
    for k in want:
        lo, hi = ranges[k](me)
        full = torch.empty((hi - lo,) + tuple(wc[k].shape[1:-1]) + (N,), dtype=wc[k].dtype, device=wc[k].device)
        full[..., s0:s0 + ns] = wc[k][lo:hi]
        out[k] = full
        for s, peer in enumerate(peers):
            if s == me:
                continue
            plo, phi = ranges[k](s)
            if phi > plo and ns > 0:
                sends.append((wc[k][plo:phi], peer))
            ps0, pns = plan.slot_range(N, s)
            if hi > lo and pns > 0:
                buf = torch.empty((hi - lo,) + tuple(wc[k].shape[1:-1]) + (pns,), dtype=wc[k].dtype, device=wc[k].device)
                recvs.append((buf, peer))
                pending.append((full, ps0, pns, buf))
    wait = _p2p_start(sends, recvs, group)

    def finish():
        wait()
        for full, ps0, pns, buf in pending:
            full[..., ps0:ps0 + pns] = buf
    if defer:
        return out, finish
    finish()
    return out


def _gather_limbs(plan: ShardPlan, piece, n_elems):
    """all-gather over every rank, keep one copy per limb (shard 0 of each limb group holds the reduced sums)."""
    if plan.world == 1:
        return piece
    pieces = [torch.empty_like(piece) for _ in range(plan.world)]
    dist.all_gather(pieces, piece)
    full = torch.empty((n_elems, plan.L) + tuple(piece.shape[2:]), dtype=piece.dtype, device=piece.device)
    for lg in range(plan.limb_groups):
        limbs = [i for i in range(plan.L) if i % plan.limb_groups == lg]
        full[:, limbs] = pieces[lg]  # rank lg is shard 0 of limb group lg
    return full


def groth16_prove_sharded(backend, plan: ShardPlan, term_group, cs_local, pk_local, assignment_local, m, n_inputs, n_aux,
                          aux_kinds=None):
    """groth16::prover (zk_proof_systems/groth16/groth16.tcc:70-115) on this rank's shard.

    *_local hold only this rank's limbs.  aux_kinds [n_aux] (uint8, RS_KIND_*; None = polynomials): auxiliary wires held
    as RingElem Scalar 1 pass their key element through (seal_ring.tcc:525-527) -- a property of the wire, the same on
    every limb.  Returns the full proof [3][L][2][K][N_enc] (int64) on every rank."""
    if aux_kinds is not None:
        aux_kinds = np.ascontiguousarray(aux_kinds, dtype=np.uint8)
        assert aux_kinds.shape == (n_aux,)
    if plan.term_shards == 1 and hasattr(backend, "groth16_prove_local"):
        # limbs only: this rank's limbs of the proof are an ordinary proof over its own context (the fused device prover,
        # rs_groth16_prove), and nothing is exchanged before the proof is assembled
        kinds_all = None if aux_kinds is None else np.concatenate([np.zeros(n_inputs, dtype=np.uint8), aux_kinds])
        piece = backend.groth16_prove_local(cs_local, pk_local, assignment_local, kinds_all)
        if piece is not None:
            return _gather_limbs(plan, piece.contiguous(), 3)
    rng_ab = lambda s: plan.term_range(m, s)
    rng_h = lambda s: plan.term_range(m + 1, s)
    w, finish = sharded_witness(backend, plan, term_group, cs_local, assignment_local, ("A_io", "A_mid", "B_io", "B_mid", "H"),
                                {"A_io": rng_ab, "A_mid": rng_ab, "B_io": rng_ab, "B_mid": rng_ab, "H": rng_h}, defer=True)
    lead = plan.term_shard == 0  # exactly one shard per limb group adds alpha / beta
    ranges = groth16_key_ranges(plan, m, n_aux)
    c2 = None
    if n_aux:  # <delta_mid, aux> reads the assignment only: it runs while the coefficient rows are exchanged
        lo, hi = ranges["delta_mid"]
        aux = assignment_local[n_inputs:]
        c2, _ = backend.msm([key_slice(pk_local["delta_mid"], lo, hi)], [(aux[lo:hi], None if aux_kinds is None else aux_kinds[lo:hi], 0)], 1)
    finish()
    lo, hi = ranges["s_pows"]
    ab, _ = backend.msm([key_slice(pk_local["s_pows"], lo, hi)], [(w["A_io"], None, 0), (w["A_mid"], None, 0), (w["B_io"], None, 1), (w["B_mid"], None, 1)], 2,
                        addends=[pk_local["alpha"], pk_local["beta"]] if lead else None)
    lo, hi = ranges["delta_ts"]
    c, _ = backend.msm([key_slice(pk_local["delta_ts"], lo, hi)], [(w["H"], None, 0)], 1)
    ab, c = ab[0], c[0]
    if c2 is not None:
        c = backend.enc_add(c, c2[0])
    piece = torch.cat([ab.reshape((2,) + tuple(ab.shape[-4:])), c.reshape((1,) + tuple(c.shape[-4:]))], dim=0).contiguous()
    if plan.term_shards > 1:
        if hasattr(backend, "check_allreduce_headroom"):
            backend.check_allreduce_headroom(plan.term_shards)
        dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=term_group)
        piece = backend.enc_reduce(piece)
    return _gather_limbs(plan, piece, 3)


def rinocchio_prove_sharded(backend, plan: ShardPlan, term_group, cs_local, pk_local, assignment_local, m, n_inputs, n_aux,
                            d1=None, d2=None, d3=None, aux_kinds=None):
    """rinocchio::prover (zk_proof_systems/rinocchio/rinocchio.tcc:75-190) on this rank's shard: the ten inner
    products of :106-163 over this rank's term range in one grouped pass over both key vectors, F over its range
    of beta_prods (:176-185), one all-reduce of the eleven partial sums, then the ZK shifts (:167-174, 181-183) on
    the reduced sums.  d1,d2,d3: ring elements [L_local][N] or all None.  Returns (proof [9][L][2][K][N_enc],
    empty[9]) on every rank."""
    zk = d1 is not None
    assert (d1 is None) == (d2 is None) == (d3 is None)
    def rng_mid(s):  # the *_mid vectors have m rows: a late shard's range may be empty, never negative
        lo, hi = plan.term_range(m + 1, s)
        return min(lo, m), max(min(lo, m), min(hi, m))
    rng_h = lambda s: plan.term_range(m + 1, s)
    w = sharded_witness(backend, plan, term_group, cs_local, assignment_local, ("A_mid", "B_mid", "C_mid", "H"),
                        {"A_mid": rng_mid, "B_mid": rng_mid, "C_mid": rng_mid, "H": rng_h}, (d1, d2, d3))
    ranges = rinocchio_key_ranges(plan, m, n_aux)
    lo, hi = ranges["s_pows"]
    # coefficients_for_Z as ring elements (slot constant); its monic leading coefficient is the RingElem
    # Scalar 1 (evaluation_domain.tcc:55-58), which passes the ciphertext through (seal_ring.tcc:525-527)
    zrows = backend.broadcast_scalars(w["Z"][:, lo:hi])
    zkinds = np.full(hi - lo, RS_KIND_POLY, dtype=np.uint8)
    if hi == m + 1 and hi > lo:
        zkinds[-1] = RS_KIND_ONE
    mo, used = backend.msm([key_slice(pk_local["s_pows"], lo, hi), key_slice(pk_local["alpha_s_pows"], lo, hi)],
                           [(w["A_mid"], None, 0), (w["B_mid"], None, 1), (w["C_mid"], None, 2), (w["H"], None, 3), (zrows, zkinds, 4)], 5,
                           want_used=True)
    enc_shape = tuple(mo.shape[-4:])
    used_f = 0
    if n_aux:
        flo, fhi = ranges["beta_prods"]
        fk = None if aux_kinds is None else np.ascontiguousarray(aux_kinds, dtype=np.uint8)[flo:fhi]  # Scalar-1 wires (seal_ring.tcc:525-527)
        f, uf = backend.msm([key_slice(pk_local["beta_prods"], flo, fhi)], [(assignment_local[n_inputs:][flo:fhi], fk, 0)], 1, want_used=True)
        f, used_f = f.reshape((1,) + enc_shape), uf[0]
    else:
        f = torch.zeros((1,) + enc_shape, dtype=mo.dtype, device=mo.device)
    piece = torch.cat([mo.reshape((10,) + enc_shape), f], dim=0).contiguous()
    counts = torch.tensor(list(used) + [used_f], dtype=torch.int64)
    if plan.term_shards > 1:
        if hasattr(backend, "check_allreduce_headroom"):
            backend.check_allreduce_headroom(plan.term_shards)
        dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=term_group)
        piece = backend.enc_reduce(piece)
        cnt = counts.to(piece.device) if dist.get_backend(term_group) != "gloo" else counts
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=term_group)
        counts = cnt.cpu()
    counts = [int(x) for x in counts]
    slot = lambda c, g: piece[c * 5 + g]
    elems, empty = [], []
    for k in range(4):  # A, A', B, B', C, C', D, D'
        for c in range(2):
            elems.append(slot(c, k))
            empty.append(counts[k] == 0)
    elems.append(piece[10])
    empty.append(counts[5] == 0)

    def add_scaled(idx, enc, d):  # elems[idx] += d * enc   (RingT * EncT then +=, rinocchio.tcc:168-173, 181-183)
        t = backend.enc_mul_ring(enc, d)
        elems[idx] = t if empty[idx] else backend.enc_add(elems[idx], t)
        empty[idx] = False

    if zk:
        for k, d in enumerate((d1, d2, d3)):
            add_scaled(2 * k, slot(0, 4), d)
            add_scaled(2 * k + 1, slot(1, 4), d)
        if n_aux:
            for d, name in ((d1, "beta_rv_ts"), (d2, "beta_rw_ts"), (d3, "beta_ry_ts")):
                add_scaled(8, pk_local[name], d)
    out = torch.stack([e.reshape(enc_shape) for e in elems]).contiguous()
    return _gather_limbs(plan, out, 9), [int(e) for e in empty]


def fused_groth16_key(pk_local, m, n_aux):
    """Key of the fused device prover (rs_groth16_prove) from a rank's local key, or None when the rank does not hold
    what that prover reads: s_pows on terms [0, m) (A and B have m coefficients, groth16.tcc:89-101; groth16_key_ranges
    stores exactly those m of its m + 1 entries), delta_ts on [0, m], delta_mid on [0, n_aux); all three stored in full
    or all three with the same window.  Returns {s_pows, delta_ts, delta_mid, alpha, beta, window}."""
    need = {"s_pows": m, "delta_ts": m + 1, "delta_mid": n_aux}
    windows, stores = set(), {}
    for k, n in need.items():
        v = pk_local.get(k)
        if isinstance(v, TiledKey):
            if v.lo != 0 or v.hi < n:
                return None
            windows.add(v.window)
            stores[k] = v.store
        else:
            if v is None or len(v) < n:
                return None
            windows.add(0)
            stores[k] = v
    if len(windows) != 1:
        return None
    return dict(stores, alpha=pk_local["alpha"], beta=pk_local["beta"], window=windows.pop())


class DeviceBackend:
    """Production backend: ringsnark_amd.device.Device over this rank's limb subset."""

    def __init__(self, dev):
        self.dev = dev
        self.N = dev.N

    def witness(self, dcs, assignment, want, ds=(None, None, None), rows=None):
        """rows: {name: (lo, hi)} -> those outputs hold rows [lo, hi) only (rs_witness_map_rows)"""
        return self.dev.witness_map(dcs, assignment, *ds, want=want, rows=rows)

    def witness_slots(self, dcs, assignment, slot0, nslots, want, ds=(None, None, None)):
        return self.dev.witness_map_slots(dcs, assignment, slot0, nslots, *ds, want=want)

    def msm(self, crs_list, vecs, n_groups, addends=None, want_used=False):
        """crs_list: KeySlice per key vector (same length / window).  Returns ([n_crs][n_groups] encodings, used)."""
        ks = crs_list[0]
        assert all(k.length == ks.length and k.window == ks.window for k in crs_list)
        out, used = self.dev.msm([k.tensor for k in crs_list], [(v.contiguous(), kinds, g) for v, kinds, g in vecs], n_groups,
                                 want_used=want_used, crs_len=ks.length, window=ks.window)
        if addends is not None:
            out = torch.stack([torch.stack([self.dev.enc_add(out[c][g], addends[g]) for g in range(n_groups)]) for c in range(len(crs_list))])
        return out, used

    def groth16_prove_local(self, dcs, pk_local, assignment, kinds=None):
        """The whole prover on this rank's context (rs_groth16_prove); None when a key vector does not start at term 0
        or stops short of what the fused prover reads, or the vectors are not equally windowed (then the caller runs
        the piecewise plan)."""
        pk1 = fused_groth16_key(pk_local, dcs.m, dcs.n_vars - dcs.n_inputs)
        if pk1 is None:
            return None
        window = pk1.pop("window")
        return self.dev.groth16_prove(dcs, pk1, assignment, want_empty=False, window=window, kinds=kinds)[0]

    def enc_add(self, a, b):
        return self.dev.enc_add(a.contiguous(), b.contiguous())

    def enc_mul_ring(self, enc, ring):
        return self.dev.enc_mul_ring(enc.contiguous(), ring.contiguous())

    def enc_reduce(self, piece):
        return self.dev.enc_reduce(piece)

    def check_allreduce_headroom(self, shards):
        """The all-reduce adds `shards` canonical residues as int64: needs shards * max(Q_j) < 2^63."""
        assert shards * max(int(x) for x in self.dev.prm.Q) < 2**63, "too many term shards for these moduli"

    def broadcast_scalars(self, z):
        """[L][rows] host residues -> ring elements [rows][L][N] with the value in every slot."""
        t = self.dev.put(np.ascontiguousarray(z.T))  # [rows][L]
        return t.unsqueeze(-1).expand(t.shape[0], t.shape[1], self.N).contiguous()
