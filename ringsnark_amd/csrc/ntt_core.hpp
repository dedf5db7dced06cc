// ntt_core.hpp -- workgroup-cooperative radix-2^R number-theoretic transform over an LDS tile.
//
// One workgroup owns one polynomial of n = 2^logn FP64-held residues (f64mod.hpp).  Stages are
// processed in "rounds" of R <= MAXR stages: a thread pulls the 2^R elements of one radix-2^R
// butterfly into registers, runs R stages there and writes them back, so a length-8192 transform
// with MAXR = 4 costs 4 exchanges instead of 13.  The first round can read its operands straight
// from global memory and the last can write straight back (IO functors), which removes two more
// LDS passes.  LDS indices are padded by one slot per 16 (pidx) so that every stride pattern of
// every round is bank-conflict free for ds_read/write_b64 (MI355X: 64 banks x 4 B, two 32-lane
// groups per b64 access).
//
// Twiddles: table entry tw[M*root + i] serves group i of the stage that has M groups, inside the
// sub-transform rooted at decimation-tree node `root` (root = 1 for a whole transform).  The
// same code runs the negacyclic transforms of the encoding contexts (SEAL order) and the cyclic
// transforms of the witness map; only the tables differ.
#pragma once
#include <hip/hip_runtime.h>

#include "intmod.hpp"

namespace rs {

// ---- LDS address map ---------------------------------------------------------------------------
// MI355X LDS (MI355X_MICROARCH.md): ds_read_b64 is served in two 32-lane groups, bank pair =
// (8-byte word index) mod 32; ds_write_b64 in four 16-lane groups, word index mod 16.  The rounds
// of a transform touch the tile in three lane patterns: contiguous (gap >= 64), blocks of 8 lanes
// 64 words apart (gap 8), and a stride of 8 or 16 words (last round).  RS_LDS_SWIZZLE = 1 maps
// element i to   i ^ f(i),  f = GF(2)-linear in bits 4..7 of i, changing bits 0..4 only:
//     a0 = b0^b4, a1 = b1^b5, a2 = b2^b6, a3 = b3^b6, a4 = b4^b7
// which is a bijection on every aligned block of 256 and conflict-free for all three patterns
// (reads 2 array cycles, writes 4; tools/lds_conflicts.py).  RS_LDS_SWIZZLE = 0 is the older
// one-pad-per-16 layout (every read pattern 2-way conflicted: 4 cycles; gap-8 writes 8).
// Because f is linear and a round's element offsets occupy bits disjoint from its base index,
// the address of element e splits into pidx(base) ^ pidx(e*step), the second part wave-uniform.
#ifndef RS_LDS_SWIZZLE
#define RS_LDS_SWIZZLE 0
#endif
#ifndef RS_UNIFORM_TWIDDLES
#define RS_UNIFORM_TWIDDLES 1
#endif
constexpr int PAD_SHIFT = 4;
#if RS_LDS_SWIZZLE
__host__ __device__ __forceinline__ int pidx(int i) {
  const int t = i >> 4;
  return i ^ (t & 7) ^ ((t & 12) << 1);
}
__host__ __device__ inline size_t padded_len(size_t n) { return n; }
__host__ __device__ __forceinline__ int pcomb(int pb, int poff) { return pb ^ poff; }
__host__ __device__ __forceinline__ int pnext(int pi) { return pi ^ 1; }  // address of element i+1, i even
#else
__host__ __device__ __forceinline__ int pidx(int i) { return i + (i >> PAD_SHIFT); }
__host__ __device__ inline size_t padded_len(size_t n) { return n + (n >> PAD_SHIFT); }
__host__ __device__ __forceinline__ int pcomb(int pb, int poff) { return pb + poff; }
__host__ __device__ __forceinline__ int pnext(int pi) { return pi + 1; }
#endif
// LDS address of a lane's j-th own position lane + 64*j, given p0 = pidx(lane): 64*j is a multiple
// of the pad period, so the address is p0 plus a compile-time constant (it folds into the ds
// instruction's immediate offset instead of costing three integer ops per access).
__host__ __device__ __forceinline__ int own_pidx(int p0, int ln, int j) {
#if RS_LDS_SWIZZLE
  return pidx(ln + 64 * j);
#else
  return p0 + 64 * j + ((64 * j) >> PAD_SHIFT);
#endif
}
// smallest block whose addresses are closed under pidx and translate with the block offset
// (LdsBlockIO): wave-private transforms need n / W >= this
constexpr int LDS_BLOCK_MIN = RS_LDS_SWIZZLE ? 256 : 128;

// element accessors for a round: LDS tile or caller-supplied functors.  A round addresses its 2^R
// elements as base + e*step; for the LDS tile the mapped address splits into
// pcomb(pidx(base), poff(e)) with poff uniform across the wave (round_poff below), so an access
// costs one vector op; functors over global memory just use base + e*step.
// T: the value type of the arithmetic (double for f64mod.hpp, uint64_t for intmod.hpp)
template <class T>
struct LdsIOT {
  T *s;
  __device__ __forceinline__ int pbase(int base) const { return pidx(base); }
  __device__ __forceinline__ T load(int, int pb, int, int poff) const { return s[pcomb(pb, poff)]; }
  __device__ __forceinline__ void store(int, int pb, int, int poff, T v) const { s[pcomb(pb, poff)] = v; }
};
using LdsIO = LdsIOT<double>;
// mapped offset of element offset eoff = e*step inside a radix group of E elements
__device__ __forceinline__ int round_poff(int eoff, int step, int E) {
#if RS_LDS_SWIZZLE
  return pidx(eoff);
#else
  // exact because a group never straddles a 16-slot pad boundary in a way that depends on the lane
  if (step >= 16) return eoff + (eoff >> PAD_SHIFT);
  if (E * step >= 16) return eoff + (eoff >> PAD_SHIFT);
  return eoff;
#endif
}

// number of stages of the next round when `rem` stages remain: spread evenly over the minimum
// number of rounds
__host__ __device__ __forceinline__ int pick_radix(int rem, int maxr) {
  const int rounds = (rem + maxr - 1) / maxr;
  return (rem + rounds - 1) / rounds;
}

// ---- forward (Cooley-Tukey, natural in -> bit-reversed out) -----------------------------------
// stage s (0-based) has 2^s groups and gap n >> (s+1); butterfly (x, y) -> (x + w*y, x - w*y).
// `logsub`: the tile holds 2^(logtot-logsub) independent length-2^logsub transforms on consecutive
// blocks (logsub == logtot for a single transform).
// Who executes a round: the whole workgroup (default) or one wave on its own sub-block.
struct Lanes {
  int tid, nthr;
  bool opaque;  // recompute addresses inside the round (for rounds that sit in a caller's loop)
};
__device__ __forceinline__ Lanes block_lanes() { return Lanes{(int)threadIdx.x, (int)blockDim.x, false}; }
__device__ __forceinline__ Lanes wave_lanes(bool opaque = false) { return Lanes{(int)(threadIdx.x & 63), 64, opaque}; }
// wave-scope ordering of LDS traffic: DS operations of one wave execute in issue order, so a
// compiler-level fence is all that is needed between a wave-private exchange's writes and reads.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A source functor may declare `static constexpr bool zero_upper = true`: it feeds the FIRST round of a transform
// whose upper half is zero padding, so elements e >= E/2 are known zeros and the first stage is a copy
// (x + w*0, x - w*0) -- no loads, no multiplies, bit-identical.
template <class F, class = void>
struct zero_upper_of {
  static constexpr bool value = false;
};
template <class F>
struct zero_upper_of<F, decltype((void)F::zero_upper)> {
  static constexpr bool value = F::zero_upper;
};

// T, M: value and modulus type of the arithmetic, deduced from the twiddle table and the modulus
template <int R, class In, class Out, class T, class M>
__device__ __forceinline__ void fwd_round(const In in, const Out out, int logtot, int logsub, int s0,
                                          const T *__restrict__ tw, int root, const M mod,
                                          uint32_t red_mask, const Lanes ln = block_lanes()) {
  constexpr int E = 1 << R;
  const int lstep = logsub - s0 - R;  // log2 of the smallest gap in this round
  const int sstep_ = 1 << lstep;
  const int ngroups = (1 << logtot) >> R;
  for (int grp_ = ln.tid; grp_ < ngroups; grp_ += ln.nthr) {
    // Opaque copies: inside a caller's loop every address below is loop invariant, and hoisting
    // all of them (dozens per round, several rounds) costs far more registers than recomputing.
    int grp = grp_, sstep = sstep_;
    if (ln.opaque) asm volatile("" : "+v"(grp), "+s"(sstep));
    const int lo = grp & (sstep - 1);
    int hi_all = grp >> lstep;
#if RS_UNIFORM_TWIDDLES
    // The 64 lanes of a wave hold 64 consecutive groups (tid = 64*wave + lane, strides are multiples of 64), so
    // with a smallest gap of >= 64 they sit in ONE aligned block of 2^lstep groups: hi_all, and with it every
    // twiddle index of the round, is wave-uniform.  Saying so turns the twiddle fetches into scalar loads
    // (no vector address arithmetic, no VMEM issue slots, operands straight from SGPRs).
    if (lstep >= 6) hi_all = __builtin_amdgcn_readfirstlane(hi_all);
#endif
    const int hi = hi_all & ((1 << s0) - 1);
    const int base = (hi_all << (logsub - s0)) + lo;
    T v[E];
    constexpr bool ZU = zero_upper_of<In>::value;
    const int pbi = in.pbase(base), pbo = out.pbase(base);
#pragma unroll
    for (int e = 0; e < (ZU ? E / 2 : E); e++) v[e] = in.load(base, pbi, e * sstep, round_poff(e * sstep, sstep, E));
#pragma unroll
    for (int k = 0; k < R; k++) {
      if ((red_mask >> (s0 + k)) & 1u) {
#pragma unroll
        for (int e = 0; e < ((ZU && k == 0) ? E / 2 : E); e++) v[e] = reduce(v[e], mod);
      }
      if (ZU && k == 0) {
#pragma unroll
        for (int e = 0; e < E / 2; e++) v[e + E / 2] = v[e];
        continue;
      }
      const int half = E >> (k + 1);
      const int twbase = ((1 << (s0 + k)) * root) + (hi << k);
#pragma unroll
      for (int blk = 0; blk < (1 << k); blk++) {
        const T w = tw[twbase + blk];
#pragma unroll
        for (int e0 = 0; e0 < half; e0++) {
          const int ia = blk * 2 * half + e0, ib = ia + half;
          const T t = mulmod(v[ib], w, mod);
          const T a = v[ia];
          v[ia] = addm(a, t, mod);
          v[ib] = subm(a, t, mod);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < E; e++) out.store(base, pbo, e * sstep, round_poff(e * sstep, sstep, E), v[e]);
  }
}

template <int MAXR, class In, class Out, class T, class M>
__device__ __forceinline__ void fwd_round_dispatch(int R, const In in, const Out out, int logtot, int logsub, int s0,
                                                   const T *__restrict__ tw, int root, const M mod,
                                                   uint32_t red_mask, const Lanes ln = block_lanes()) {
  if (MAXR >= 5 && R == 5)
    fwd_round<(MAXR >= 5 ? 5 : 1)>(in, out, logtot, logsub, s0, tw, root, mod, red_mask, ln);
  else if (MAXR >= 4 && R == 4)
    fwd_round<(MAXR >= 4 ? 4 : 1)>(in, out, logtot, logsub, s0, tw, root, mod, red_mask, ln);
  else if (R == 3)
    fwd_round<3>(in, out, logtot, logsub, s0, tw, root, mod, red_mask, ln);
  else if (R == 2)
    fwd_round<2>(in, out, logtot, logsub, s0, tw, root, mod, red_mask, ln);
  else
    fwd_round<1>(in, out, logtot, logsub, s0, tw, root, mod, red_mask, ln);
}

// Whole forward transform(s) of the LDS tile; ends with a barrier.  Caller must have synchronised
// after filling the tile.  first_in / last_out replace the LDS tile for the first round's loads
// and the last round's stores.
template <int MAXR, class In, class Out, class T, class M>
__device__ __forceinline__ void lds_ntt_fwd_io(T *s, const In first_in, const Out last_out, int logtot, int logsub,
                                               const T *__restrict__ tw, int root, const M mod,
                                               uint32_t red_mask) {
  const LdsIOT<T> lds{s};
  int st = 0;
  while (st < logsub) {
    const int R = pick_radix(logsub - st, MAXR);
    const bool first = st == 0, last = st + R >= logsub;
    if (first && last)
      fwd_round_dispatch<MAXR>(R, first_in, last_out, logtot, logsub, st, tw, root, mod, red_mask);
    else if (first)
      fwd_round_dispatch<MAXR>(R, first_in, lds, logtot, logsub, st, tw, root, mod, red_mask);
    else if (last)
      fwd_round_dispatch<MAXR>(R, lds, last_out, logtot, logsub, st, tw, root, mod, red_mask);
    else
      fwd_round_dispatch<MAXR>(R, lds, lds, logtot, logsub, st, tw, root, mod, red_mask);
    __syncthreads();
    st += R;
  }
}
template <int MAXR = 3, class T, class M>
__device__ __forceinline__ void lds_ntt_fwd(T *s, int logn, const T *__restrict__ tw, int root,
                                            const M mod, uint32_t red_mask) {
  const LdsIOT<T> lds{s};
  lds_ntt_fwd_io<MAXR>(s, lds, lds, logn, logn, tw, root, mod, red_mask);
}
template <int MAXR = 3, class T, class M>
__device__ __forceinline__ void lds_bntt_fwd(T *s, int logtot, int logsub, const T *__restrict__ tw,
                                             const M mod, uint32_t red_mask) {
  const LdsIOT<T> lds{s};
  lds_ntt_fwd_io<MAXR>(s, lds, lds, logtot, logsub, tw, 1, mod, red_mask);
}

// ---- inverse (Gentleman-Sande, bit-reversed in -> natural out, NOT scaled by n^-1) -----------
// inverse stage u (0-based) has gap 2^u and n >> (u+1) groups; (a, b) -> (a + b, (a - b)*w).
template <int R, class In, class Out, class T, class M>
__device__ __forceinline__ void inv_round(const In in, const Out out, int logtot, int logsub, int u0,
                                          const T *__restrict__ itw, int root, const M mod,
                                          uint32_t red_mask, const Lanes ln = block_lanes()) {
  constexpr int E = 1 << R;
  const int g0_ = 1 << u0;
  const int ngroups = (1 << logtot) >> R;
  const int gpb_log = logsub - u0 - R;  // log2 of radix groups per sub-transform (per lo)
  for (int grp_ = ln.tid; grp_ < ngroups; grp_ += ln.nthr) {
    int grp = grp_, g0 = g0_;
    if (ln.opaque) asm volatile("" : "+v"(grp), "+s"(g0));
    const int lo = grp & (g0 - 1);
    int hi_all = grp >> u0;
#if RS_UNIFORM_TWIDDLES
    if (u0 >= 6) hi_all = __builtin_amdgcn_readfirstlane(hi_all);  // wave-uniform, as in fwd_round
#endif
    const int hi = hi_all & ((1 << gpb_log) - 1);
    const int base = (hi_all << (u0 + R)) + lo;
    T v[E];
    const int pbi = in.pbase(base), pbo = out.pbase(base);
#pragma unroll
    for (int e = 0; e < E; e++) v[e] = in.load(base, pbi, e * g0, round_poff(e * g0, g0, E));
#pragma unroll
    for (int k = 0; k < R; k++) {
      if ((red_mask >> (u0 + k)) & 1u) {
#pragma unroll
        for (int e = 0; e < E; e++) v[e] = reduce(v[e], mod);
      }
      const int Mg = (1 << logsub) >> (u0 + k + 1);
      const int twbase = Mg * root + (hi << (R - 1 - k));
#pragma unroll
      for (int e = 0; e < E; e++) {
        if (e & (1 << k)) continue;
        const T w = itw[twbase + (e >> (k + 1))];
        const T a = v[e], b = v[e + (1 << k)];
        v[e] = addm(a, b, mod);
        v[e + (1 << k)] = mulmod(subm(a, b, mod), w, mod);
      }
    }
#pragma unroll
    for (int e = 0; e < E; e++) out.store(base, pbo, e * g0, round_poff(e * g0, g0, E), v[e]);
  }
}

template <int MAXR, class In, class Out, class T, class M>
__device__ __forceinline__ void inv_round_dispatch(int R, const In in, const Out out, int logtot, int logsub, int u0,
                                                   const T *__restrict__ itw, int root, const M mod,
                                                   uint32_t red_mask, const Lanes ln = block_lanes()) {
  if (MAXR >= 5 && R == 5)
    inv_round<(MAXR >= 5 ? 5 : 1)>(in, out, logtot, logsub, u0, itw, root, mod, red_mask, ln);
  else if (MAXR >= 4 && R == 4)
    inv_round<(MAXR >= 4 ? 4 : 1)>(in, out, logtot, logsub, u0, itw, root, mod, red_mask, ln);
  else if (R == 3)
    inv_round<3>(in, out, logtot, logsub, u0, itw, root, mod, red_mask, ln);
  else if (R == 2)
    inv_round<2>(in, out, logtot, logsub, u0, itw, root, mod, red_mask, ln);
  else
    inv_round<1>(in, out, logtot, logsub, u0, itw, root, mod, red_mask, ln);
}

template <int MAXR, class In, class Out, class T, class M>
__device__ __forceinline__ void lds_ntt_inv_io(T *s, const In first_in, const Out last_out, int logtot, int logsub,
                                               const T *__restrict__ itw, int root, const M mod,
                                               uint32_t red_mask) {
  const LdsIOT<T> lds{s};
  int st = 0;
  while (st < logsub) {
    const int R = pick_radix(logsub - st, MAXR);
    const bool first = st == 0, last = st + R >= logsub;
    if (first && last)
      inv_round_dispatch<MAXR>(R, first_in, last_out, logtot, logsub, st, itw, root, mod, red_mask);
    else if (first)
      inv_round_dispatch<MAXR>(R, first_in, lds, logtot, logsub, st, itw, root, mod, red_mask);
    else if (last)
      inv_round_dispatch<MAXR>(R, lds, last_out, logtot, logsub, st, itw, root, mod, red_mask);
    else
      inv_round_dispatch<MAXR>(R, lds, lds, logtot, logsub, st, itw, root, mod, red_mask);
    __syncthreads();
    st += R;
  }
}
template <int MAXR = 3, class T, class M>
__device__ __forceinline__ void lds_ntt_inv(T *s, int logn, const T *__restrict__ itw, int root,
                                            const M mod, uint32_t red_mask) {
  const LdsIOT<T> lds{s};
  lds_ntt_inv_io<MAXR>(s, lds, lds, logn, logn, itw, root, mod, red_mask);
}
template <int MAXR = 3, class T, class M>
__device__ __forceinline__ void lds_bntt_inv(T *s, int logtot, int logsub, const T *__restrict__ itw,
                                             const M mod, uint32_t red_mask) {
  const LdsIOT<T> lds{s};
  lds_ntt_inv_io<MAXR>(s, lds, lds, logtot, logsub, itw, 1, mod, red_mask);
}


// ---- wave-private tail ("wp") ----------------------------------------------------------------
// With W = 2^logw waves in the workgroup, the first logw stages of a forward transform are the
// only ones that mix data of different waves.  After them the tile is W independent
// sub-transforms on contiguous blocks of n/W elements, rooted at decimation-tree nodes W + wave:
// each wave finishes its own block with wave-private LDS exchanges and NO workgroup barrier, so
// waves drift apart and their global-memory, LDS and FP64 phases overlap.  (The inverse runs the
// private rounds first and the cross-wave round last.)  Requires n / W >= 128.
//
// An offset view of the tile: valid because block offsets are multiples of LDS_BLOCK_MIN (pidx
// maps such a block onto itself and commutes with its offset).
struct LdsBlockIO {
  double *s;  // s + pidx(block_offset)
  __device__ __forceinline__ int pbase(int base) const { return pidx(base); }
  __device__ __forceinline__ double load(int, int pb, int, int poff) const { return s[pcomb(pb, poff)]; }
  __device__ __forceinline__ void store(int, int pb, int, int poff, double v) const { s[pcomb(pb, poff)] = v; }
};

// Forward.  first_in feeds the cross-wave round (global or LDS); the result of the private rounds
// goes to last_out, addressed with block-local indices + the wave's block offset added by the
// functor's owner (see WaveOut below).  Ends WITHOUT a workgroup barrier: each wave's block is
// complete (for that wave) on return.
template <int MAXR, class In, class OutFactory, int CROSSR = 4, bool OPAQUE = false>
__device__ __forceinline__ void lds_ntt_fwd_wp(double *s, const In first_in, const OutFactory make_out, int logn, int logw,
                                               const double *__restrict__ tw, const Mod mod, uint32_t red_mask,
                                               int root0 = 1) {
  const LdsIO lds{s};
  // cross-wave rounds: logw stages in rounds of at most CROSSR (each ends with a barrier)
  for (int st = 0; st < logw;) {
    const int R = pick_radix(logw - st, CROSSR);
    const Lanes bl{(int)threadIdx.x, (int)blockDim.x, OPAQUE};
    if (st == 0)
      fwd_round_dispatch<CROSSR>(R, first_in, lds, logn, logn, st, tw, root0, mod, red_mask, bl);
    else
      fwd_round_dispatch<CROSSR>(R, lds, lds, logn, logn, st, tw, root0, mod, red_mask, bl);
    __syncthreads();
    st += R;
  }
  const int wave = threadIdx.x >> 6;
  const int logb = logn - logw;
  const int off = wave << logb;
  const LdsBlockIO blk{s + pidx(off)};
  const int root = (root0 << logw) + wave;
  const uint32_t mask = red_mask >> logw;
  const Lanes ln = wave_lanes(OPAQUE);
  int st = 0;
  while (st < logb) {
    const int R = pick_radix(logb - st, MAXR);
    if (st + R >= logb)
      fwd_round_dispatch<MAXR>(R, blk, make_out(off), logb, logb, st, tw, root, mod, mask, ln);
    else
      fwd_round_dispatch<MAXR>(R, blk, blk, logb, logb, st, tw, root, mod, mask, ln);
    wave_sync();
    st += R;
  }
}

// Inverse.  make_in(off) feeds the first private round with block-local indices; the cross-wave
// round writes to last_out.  Ends with a workgroup barrier after the cross-wave round.
template <int MAXR, class InFactory, class Out, int CROSSR = 4>
__device__ __forceinline__ void lds_ntt_inv_wp(double *s, const InFactory make_in, const Out last_out, int logn, int logw,
                                               const double *__restrict__ itw, const Mod mod, uint32_t red_mask,
                                               int root0 = 1) {
  const LdsIO lds{s};
  const int wave = threadIdx.x >> 6;
  const int logb = logn - logw;
  const int off = wave << logb;
  const LdsBlockIO blk{s + pidx(off)};
  // inverse stage u of the block == inverse stage u of the whole transform; group index inside the
  // block's subtree: node = M_block * (W + wave) + i  with M_block groups per block at that stage
  const int root = (root0 << logw) + wave;
  const Lanes ln = wave_lanes();
  int st = 0;
  while (st < logb) {
    const int R = pick_radix(logb - st, MAXR);
    if (st == 0)
      inv_round_dispatch<MAXR>(R, make_in(off), blk, logb, logb, st, itw, root, mod, red_mask, ln);
    else
      inv_round_dispatch<MAXR>(R, blk, blk, logb, logb, st, itw, root, mod, red_mask, ln);
    wave_sync();
    st += R;
  }
  __syncthreads();
  for (int st = 0; st < logw;) {
    const int R = pick_radix(logw - st, CROSSR);
    if (st + R >= logw)
      inv_round_dispatch<CROSSR>(R, lds, last_out, logn, logn, logb + st, itw, root0, mod, red_mask);
    else
      inv_round_dispatch<CROSSR>(R, lds, lds, logn, logn, logb + st, itw, root0, mod, red_mask);
    __syncthreads();
    st += R;
  }
}

// Batched wave-private transforms: the tile holds 2^(logn - logsub) transforms of length 2^logsub,
// each spanning 2^cw wave blocks (cw = logsub - (logn - logw) >= 1).  Only the cw top stages of
// every transform cross waves; below them wave w finishes its block as the sub-transform rooted at
// node 2^cw + (w mod 2^cw).  (The product tree's levels whose nodes are larger than a block.)
template <int MAXR, class In, class OutFactory, int CROSSR = 3>
__device__ __forceinline__ void lds_bntt_fwd_wp(double *s, const In first_in, const OutFactory make_out, int logn, int logw,
                                                int logsub, const double *__restrict__ tw, const Mod mod, uint32_t red_mask) {
  const LdsIO lds{s};
  const int logb = logn - logw, cw = logsub - logb;
  for (int st = 0; st < cw;) {
    const int R = pick_radix(cw - st, CROSSR);
    if (st == 0)
      fwd_round_dispatch<CROSSR>(R, first_in, lds, logn, logsub, st, tw, 1, mod, red_mask);
    else
      fwd_round_dispatch<CROSSR>(R, lds, lds, logn, logsub, st, tw, 1, mod, red_mask);
    __syncthreads();
    st += R;
  }
  const int wave = threadIdx.x >> 6;
  const int off = wave << logb;
  const LdsBlockIO blk{s + pidx(off)};
  const int root = (1 << cw) + (wave & ((1 << cw) - 1));
  const uint32_t mask = red_mask >> cw;
  const Lanes ln = wave_lanes();
  int st = 0;
  while (st < logb) {
    const int R = pick_radix(logb - st, MAXR);
    if (st + R >= logb)
      fwd_round_dispatch<MAXR>(R, blk, make_out(off), logb, logb, st, tw, root, mod, mask, ln);
    else
      fwd_round_dispatch<MAXR>(R, blk, blk, logb, logb, st, tw, root, mod, mask, ln);
    wave_sync();
    st += R;
  }
}
template <int MAXR, class InFactory, class Out, int CROSSR = 3>
__device__ __forceinline__ void lds_bntt_inv_wp(double *s, const InFactory make_in, const Out last_out, int logn, int logw,
                                                int logsub, const double *__restrict__ itw, const Mod mod, uint32_t red_mask) {
  const LdsIO lds{s};
  const int logb = logn - logw, cw = logsub - logb;
  const int wave = threadIdx.x >> 6;
  const int off = wave << logb;
  const LdsBlockIO blk{s + pidx(off)};
  const int root = (1 << cw) + (wave & ((1 << cw) - 1));
  const Lanes ln = wave_lanes();
  int st = 0;
  while (st < logb) {
    const int R = pick_radix(logb - st, MAXR);
    if (st == 0)
      inv_round_dispatch<MAXR>(R, make_in(off), blk, logb, logb, st, itw, root, mod, red_mask, ln);
    else
      inv_round_dispatch<MAXR>(R, blk, blk, logb, logb, st, itw, root, mod, red_mask, ln);
    wave_sync();
    st += R;
  }
  __syncthreads();
  for (int st = 0; st < cw;) {
    const int R = pick_radix(cw - st, CROSSR);
    if (st + R >= cw)
      inv_round_dispatch<CROSSR>(R, lds, last_out, logn, logsub, logb + st, itw, 1, mod, red_mask);
    else
      inv_round_dispatch<CROSSR>(R, lds, lds, logn, logsub, logb + st, itw, 1, mod, red_mask);
    __syncthreads();
    st += R;
  }
}

}  // namespace rs
