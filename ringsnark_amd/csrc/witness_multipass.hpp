// witness_multipass.hpp -- multi-pass column transforms: cross passes over global memory and rooted LDS sub-transforms (witness.hip)
#pragma once
#include "witness_cols.hpp"
#include "witness_inc.hpp"

namespace rs {

// =============================================================================================
// Multi-pass column transforms for M > 2^g_witness_lds_logM (a column no longer fits one LDS tile).
// A cyclic transform of length n = n1 * Bn over a column held in global memory is
//     forward:  log2(n1) "cross" stages (gap >= Bn; twiddles depend on the block index only),
//               then n1 independent length-Bn sub-transforms rooted at tree nodes n1 + b, in LDS;
//     inverse:  the sub-transforms first, then the cross stages.
// Both reuse the round functions of ntt_core.hpp (global-memory functors / `root`).
// =============================================================================================
struct TabPtrs {
  const void *t[RS_MAX_L];  // tables of the context's arithmetic (8-byte words)
  const void *w1, *w3;      // MODE 4: two other workspaces of the same shape (spectra, lazily reduced)
};
#ifndef RS_WORKSPACE_NT
#define RS_WORKSPACE_NT 1
#endif
// two adjacent words moved by one 16-byte access (the paired cross passes below)
template <class T>
struct Pair {
  T x, y;
};
template <class T>
__device__ __forceinline__ Pair<T> ld_pair(const T *p) {
  typedef T V2 __attribute__((ext_vector_type(2)));
  const V2 v = *reinterpret_cast<const V2 *>(p);
  return Pair<T>{v.x, v.y};
}
template <class T>
__device__ __forceinline__ void st_pair(T *p, T x, T y) {
  typedef T V2 __attribute__((ext_vector_type(2)));
  V2 v;
  v.x = x;
  v.y = y;
  *reinterpret_cast<V2 *>(p) = v;
}
template <class T>
struct GlobalIOT {
  T *p;
  __device__ __forceinline__ int pbase(int) const { return 0; }
  __device__ __forceinline__ Pair<T> load2(int k) const {
    typedef T V2 __attribute__((ext_vector_type(2)));
#if RS_WORKSPACE_NT
    const V2 v = __builtin_nontemporal_load(reinterpret_cast<const V2 *>(p + k));
#else
    const V2 v = *reinterpret_cast<const V2 *>(p + k);
#endif
    return Pair<T>{v.x, v.y};
  }
  __device__ __forceinline__ void store2(int k, T x, T y) const {
    typedef T V2 __attribute__((ext_vector_type(2)));
    V2 v;
    v.x = x;
    v.y = y;
#if RS_WORKSPACE_NT
    __builtin_nontemporal_store(v, reinterpret_cast<V2 *>(p + k));
#else
    *reinterpret_cast<V2 *>(p + k) = v;
#endif
  }
#if RS_WORKSPACE_NT  // the multi-pass workspaces are streamed once per pass and are far larger than L2 and the Infinity Cache
  __device__ __forceinline__ T load(int base, int, int eoff, int) const { return __builtin_nontemporal_load(p + base + eoff); }
  __device__ __forceinline__ void store(int base, int, int eoff, int, T v) const { __builtin_nontemporal_store(v, p + base + eoff); }
#else
  __device__ __forceinline__ T load(int base, int, int eoff, int) const { return p[base + eoff]; }
  __device__ __forceinline__ void store(int base, int, int eoff, int, T v) const { p[base + eoff] = v; }
#endif
};
using GlobalF64IO = GlobalIOT<double>;

// ---- cross passes with fused sources and sinks ----------------------------------------------------
// The first forward pass of a transform reads its input through a source functor (padding,
// scaling, centring, node splitting, reversal happen on the fly, from the caller's buffer); the
// last inverse pass hands its output to a sink functor (truncation, node recombination, H
// extraction).  Every other pass works in place on the workspace.  This removes the separate
// element-wise launches (and their HBM round trips) around every multi-pass transform.
enum CrossSrc { CS_PLAIN = 0, CS_SCALE_PAD, CS_FILL_RIGHT, CS_PAD_CENTER, CS_REV_TRUNC, CS_COSET };
enum CrossDst { CD_PLAIN = 0, CD_TAKE_LOW, CD_COMBINE, CD_COMBINE_CANON, CD_H_FINISH, CD_H_FINISH_CANON, CD_H_COSET, CD_H_COSET_CANON };
struct CrossArgs {
  void *W;          // workspace columns [ncols][2^logtot]   (8-byte words of the context's arithmetic)
  const void *src;  // source columns (CS_*): [ncols][M] (CS_REV_TRUNC: [ncols][2M])
  void *dst;        // sink columns (CD_*): [ncols][M]
  int logtot, logsub, s0, logM, l, m;
  size_t col0;
  unsigned S, slots_per_limb;
};
template <int SRC, class Mt>
struct CrossIn {
  using T = typename ArithOf<Mt>::T;
  // Every source feeds the FIRST round of a transform whose upper half is zero padding (length-2M transforms of M
  // inputs; per tree node (F_right, 0)): the round skips those loads and its first stage is a copy (ntt_core.hpp)
  // (CS_COSET: a full-length transform of M coefficients times g^k -- `invfact` points at that table; no padding)
  static constexpr bool zero_upper = SRC != CS_COSET;
  const T *p;  // this column of the source
  const T *invfact;
  Mt mod;
  int M, m, n, h;
  __device__ __forceinline__ int pbase(int) const { return 0; }
  __device__ __forceinline__ T load(int base, int, int eoff, int) const {
    const int k = base + eoff;
    if (SRC == CS_SCALE_PAD) return k < M ? mulmod(p[k], invfact[k], mod) : T(0);  // values * 1/k!, zero padded
    if (SRC == CS_FILL_RIGHT) return (k & (n - 1)) < h ? p[k + h] : T(0);         // per node: (F_right, 0)
    if (SRC == CS_PAD_CENTER) return k < M ? center(p[k], mod) : T(0);
    if (SRC == CS_REV_TRUNC) return k < m - 1 ? reduce(p[2 * m - 2 - k], mod) : T(0);  // T_k = P_{2m-2-k}, k < m-1
    if (SRC == CS_COSET) return mulmod(center(p[k], mod), invfact[k], mod);            // a_k g^k
    return p[k];
  }
  // elements k, k + 1 (k even; M, n/2 even: both fall on the same side of every boundary but CS_REV_TRUNC's)
  __device__ __forceinline__ Pair<T> load2(int k) const {
    if (SRC == CS_REV_TRUNC) return Pair<T>{load(k, 0, 0, 0), load(k + 1, 0, 0, 0)};  // reversed, odd-aligned: two words
    Pair<T> r{T(0), T(0)};
    if (SRC == CS_SCALE_PAD) {
      if (k < M) {
        const Pair<T> a = ld_pair(p + k), f = ld_pair(invfact + k);
        r = Pair<T>{mulmod(a.x, f.x, mod), mulmod(a.y, f.y, mod)};
      }
    } else if (SRC == CS_FILL_RIGHT) {
      if ((k & (n - 1)) < h) r = ld_pair(p + k + h);
    } else if (SRC == CS_PAD_CENTER) {
      if (k < M) {
        const Pair<T> a = ld_pair(p + k);
        r = Pair<T>{center(a.x, mod), center(a.y, mod)};
      }
    } else if (SRC == CS_COSET) {
      const Pair<T> a = ld_pair(p + k), f = ld_pair(invfact + k);
      r = Pair<T>{mulmod(center(a.x, mod), f.x, mod), mulmod(center(a.y, mod), f.y, mod)};
    } else {
      r = ld_pair(p + k);
    }
    return r;
  }
};
template <int DST, class Mt>
struct CrossOut {
  using T = typename ArithOf<Mt>::T;
  T *p;  // this column of the sink
  const T *invfact;
  Mt mod;
  int M, m, n, h;
  __device__ __forceinline__ int pbase(int) const { return 0; }
  __device__ __forceinline__ void store(int base, int, int eoff, int, T v) const {
    const int k = base + eoff;
    if (DST == CD_TAKE_LOW) {  // Newton coefficients k < m of the length-2M convolution
      if (k < M) p[k] = (invfact[k] != T(0)) ? reduce(v, mod) : T(0);
    } else if (DST == CD_COMBINE || DST == CD_COMBINE_CANON) {  // F_node = (F_left, 0) + D_left * F_right
      const T f = reduce(addm(v, ((k & (n - 1)) < h ? p[k] : T(0)), mod), mod);
      p[k] = DST == CD_COMBINE_CANON ? canon(f, mod) : f;
    } else if (DST == CD_H_FINISH) {  // H_j = U_{m-2-j}; positions j > m-2 are cleared by h_patch_kernel
      if (k <= m - 2) p[m - 2 - k] = reduce(v, mod);
    } else if (DST == CD_H_FINISH_CANON) {  // no ZK patch to add: the finished column, canonical, zero above m-2
      if (k <= m - 2)
        p[m - 2 - k] = canon(v, mod);
      else if (k < M)
        p[k] = T(0);
    } else if (DST == CD_H_COSET) {  // H_k = v_k g^-k / M (`invfact` points at that table); positions above m-2: h_patch_kernel
      if (k <= m - 2) p[k] = reduce(mulmod(reduce(v, mod), invfact[k], mod), mod);
    } else if (DST == CD_H_COSET_CANON) {
      p[k] = k <= m - 2 ? canon(mulmod(reduce(v, mod), invfact[k], mod), mod) : T(0);
    } else {
      p[k] = v;
    }
  }
  __device__ __forceinline__ void store2(int k, T x, T y) const {
    if (DST == CD_TAKE_LOW) {
      if (k < M) {
        const Pair<T> f = ld_pair(invfact + k);
        st_pair(p + k, (f.x != T(0)) ? reduce(x, mod) : T(0), (f.y != T(0)) ? reduce(y, mod) : T(0));
      }
    } else if (DST == CD_COMBINE || DST == CD_COMBINE_CANON) {
      Pair<T> a{T(0), T(0)};
      if ((k & (n - 1)) < h) a = ld_pair(p + k);
      const T f0 = reduce(addm(x, a.x, mod), mod), f1 = reduce(addm(y, a.y, mod), mod);
      st_pair(p + k, DST == CD_COMBINE_CANON ? canon(f0, mod) : f0, DST == CD_COMBINE_CANON ? canon(f1, mod) : f1);
    } else if (DST == CD_H_FINISH || DST == CD_H_FINISH_CANON || DST == CD_H_COSET || DST == CD_H_COSET_CANON) {
      store(k, 0, 0, 0, x);  // reversed and odd-aligned (H_FINISH), or a limit (m - 2) of either parity: two words
      store(k + 1, 0, 0, 0, y);
    } else {
      st_pair(p + k, x, y);
    }
  }
};

// The rounds of ntt_core.hpp (fwd_round / inv_round) for the cross passes, TWO adjacent groups per thread: the two share
// every twiddle, and every access to the workspace, the source and the sink is a 16-byte one (a wave moves 1 KiB per
// instruction instead of 512 B).  Same stages, same reduction points per coefficient: the stored values are identical.
template <int R, class In, class Out, class T, class M>
__device__ __forceinline__ void fwd_round2(const In in, const Out out, int logtot, int logsub, int s0, const T *__restrict__ tw,
                                           const M mod, uint32_t red_mask, const Lanes ln) {
  constexpr int E = 1 << R;
  const int lstep = logsub - s0 - R, sstep = 1 << lstep;  // lstep >= 6: the wave's groups share hi_all
  const int npairs = (1 << logtot) >> (R + 1);
  for (int gp = ln.tid; gp < npairs; gp += ln.nthr) {
    const int grp = 2 * gp;
    const int lo = grp & (sstep - 1);
    const int hi_all = __builtin_amdgcn_readfirstlane(grp >> lstep);
    const int hi = hi_all & ((1 << s0) - 1);
    const int base = (hi_all << (logsub - s0)) + lo;
    T v[2][E];
    constexpr bool ZU = zero_upper_of<In>::value;  // elements e >= E/2 are known zeros: no loads, stage 0 is a copy
#pragma unroll
    for (int e = 0; e < (ZU ? E / 2 : E); e++) {
      const Pair<T> x = in.load2(base + e * sstep);
      v[0][e] = x.x;
      v[1][e] = x.y;
    }
#pragma unroll
    for (int k = 0; k < R; k++) {
      if ((red_mask >> (s0 + k)) & 1u) {
#pragma unroll
        for (int e = 0; e < ((ZU && k == 0) ? E / 2 : E); e++) {
          v[0][e] = reduce(v[0][e], mod);
          v[1][e] = reduce(v[1][e], mod);
        }
      }
      if (ZU && k == 0) {
#pragma unroll
        for (int e = 0; e < E / 2; e++) {
          v[0][e + E / 2] = v[0][e];
          v[1][e + E / 2] = v[1][e];
        }
        continue;
      }
      const int half = E >> (k + 1);
      const int twbase = (1 << (s0 + k)) + (hi << k);
#pragma unroll
      for (int blk = 0; blk < (1 << k); blk++) {
        const T w = ld_const(tw + twbase + blk);  // wave-uniform (hi_all): scalar cache, not a vector load behind the data loads
#pragma unroll
        for (int e0 = 0; e0 < half; e0++) {
          const int ia = blk * 2 * half + e0, ib = ia + half;
#pragma unroll
          for (int c = 0; c < 2; c++) {
            const T t = mulmod(v[c][ib], w, mod);
            const T a = v[c][ia];
            v[c][ia] = addm(a, t, mod);
            v[c][ib] = subm(a, t, mod);
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < E; e++) out.store2(base + e * sstep, v[0][e], v[1][e]);
  }
}
template <int R, class In, class Out, class T, class M>
__device__ __forceinline__ void inv_round2(const In in, const Out out, int logtot, int logsub, int u0, const T *__restrict__ itw,
                                           const M mod, uint32_t red_mask, const Lanes ln) {
  constexpr int E = 1 << R;
  const int g0 = 1 << u0;  // u0 >= 6
  const int npairs = (1 << logtot) >> (R + 1);
  const int gpb_log = logsub - u0 - R;
  for (int gp = ln.tid; gp < npairs; gp += ln.nthr) {
    const int grp = 2 * gp;
    const int lo = grp & (g0 - 1);
    const int hi_all = __builtin_amdgcn_readfirstlane(grp >> u0);
    const int hi = hi_all & ((1 << gpb_log) - 1);
    const int base = (hi_all << (u0 + R)) + lo;
    T v[2][E];
#pragma unroll
    for (int e = 0; e < E; e++) {
      const Pair<T> x = in.load2(base + e * g0);
      v[0][e] = x.x;
      v[1][e] = x.y;
    }
#pragma unroll
    for (int k = 0; k < R; k++) {
      if ((red_mask >> (u0 + k)) & 1u) {
#pragma unroll
        for (int e = 0; e < E; e++) {
          v[0][e] = reduce(v[0][e], mod);
          v[1][e] = reduce(v[1][e], mod);
        }
      }
      const int Mg = (1 << logsub) >> (u0 + k + 1);
      const int twbase = Mg + (hi << (R - 1 - k));
#pragma unroll
      for (int e = 0; e < E; e++) {
        if (e & (1 << k)) continue;
        const T w = ld_const(itw + twbase + (e >> (k + 1)));
#pragma unroll
        for (int c = 0; c < 2; c++) {
          const T a = v[c][e], b = v[c][e + (1 << k)];
          v[c][e] = addm(a, b, mod);
          v[c][e + (1 << k)] = mulmod(subm(a, b, mod), w, mod);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < E; e++) out.store2(base + e * g0, v[0][e], v[1][e]);
  }
}

// cross stages [s0, s0+R) of batched length-2^logsub transforms inside columns of length 2^logtot.
// grid (x, columns).  MODE: CrossSrc for forward passes, CrossDst for inverse passes.
// V = 2: the paired rounds (16-byte accesses; launch_cross checks the alignment of the buffers).
template <bool INV, int R, int MODE, class CPS, int V = 1>
__global__ void __launch_bounds__(256) cross_kernel(CrossArgs a, CPS plans) {
  using T = typename CPS::T;
  using Mt = typename CPS::M;
  const size_t col = blockIdx.y;
  const ColPlanT<Mt> &P = plans.l[((a.col0 + col) % a.S) / a.slots_per_limb];
  const GlobalIOT<T> io{static_cast<T *>(a.W) + (col << a.logtot)};
  const Lanes ln{(int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(gridDim.x * blockDim.x)};
  const int M = 1 << a.logM, n = 1 << a.l;
  if (INV) {
    if (MODE == CD_PLAIN) {
      if (V == 2)
        inv_round2<R>(io, io, a.logtot, a.logsub, a.s0, P.itw, P.mod, P.imask[a.logsub], ln);
      else
        inv_round<R>(io, io, a.logtot, a.logsub, a.s0, P.itw, 1, P.mod, P.imask[a.logsub], ln);
    } else {
      const CrossOut<MODE, Mt> out{static_cast<T *>(a.dst) + (col << a.logM), (MODE == CD_H_COSET || MODE == CD_H_COSET_CANON) ? P.cos_h : P.invfact,
                                   P.mod, M, a.m, n, n >> 1};
      if (V == 2)
        inv_round2<R>(io, out, a.logtot, a.logsub, a.s0, P.itw, P.mod, P.imask[a.logsub], ln);
      else
        inv_round<R>(io, out, a.logtot, a.logsub, a.s0, P.itw, 1, P.mod, P.imask[a.logsub], ln);
    }
  } else {
    if (MODE == CS_PLAIN) {
      if (V == 2)
        fwd_round2<R>(io, io, a.logtot, a.logsub, a.s0, P.tw, P.mod, P.fmask[a.logsub], ln);
      else
        fwd_round<R>(io, io, a.logtot, a.logsub, a.s0, P.tw, 1, P.mod, P.fmask[a.logsub], ln);
    } else {
      const size_t stride = MODE == CS_REV_TRUNC ? (size_t)2 << a.logM : (size_t)1 << a.logM;
      const CrossIn<MODE, Mt> in{static_cast<const T *>(a.src) + col * stride, MODE == CS_COSET ? P.cos_g : P.invfact, P.mod, M, a.m, n, n >> 1};
      if (V == 2)
        fwd_round2<R>(in, io, a.logtot, a.logsub, a.s0, P.tw, P.mod, P.fmask[a.logsub], ln);
      else
        fwd_round<R>(in, io, a.logtot, a.logsub, a.s0, P.tw, 1, P.mod, P.fmask[a.logsub], ln);
    }
  }
}

// ---- the turn of H (round 5): the LAST inverse cross pass of P = A B and the FIRST forward cross pass of T = rev(P) mod
// x^(m-1), as ONE pass over memory.  big_h (witness.hip) runs "inverse cross stages of the product, in place" and then
// "forward cross stages of the reversed, truncated product, from W2 into W1": two HBM-bound passes, 7 n words of traffic per
// column (n = 2M), of which the second re-reads what the first has just written.  Both passes hold a RESIDUE CLASS mod
// B = n / 2^R in a thread -- the inverse one positions rho + e B, the forward one positions j + e' B -- and
//     T_k = P_{2m-2-k}  (k < m - 1, zero otherwise),      k = j + e' B   <->   i = 2m - 2 - k = rho + (E0 - e') B,
//     rho = (2m - 2 - j) mod B,  E0 = (2m - 2 - j) div B,
// so the thread that finishes the inverse stages of class rho holds every input of the forward group of class j: the
// register tile is reversed (compile time) and shifted by E - 1 - E0 (a per-lane amount: a barrel of R select rounds), the
// truncation is a select, and the forward stages follow -- 4 n words per column instead of 7 n.  Same stages, reduction
// masks, twiddles and reduce-on-load as cross_kernel<true, R, CD_PLAIN> followed by cross_kernel<false, R, CS_REV_TRUNC>: the
// stored words are identical (knob witness_h_turn = 0 restores the two passes; tests compare).
// V = 2: forward pair (j, j + 1), j even, stored with 16-byte accesses; its inverse classes rho, rho - 1 are odd aligned
// (2m - 2 is even) and are loaded as 8-byte words.  V = 1 (R = 6: 64 words per group): one class per thread.
// Needs ONE pass each way: logtot - logB = R <= 6 (FP64) / 4 (integers).  grid (x, columns).
template <int R, class CPS, int V>
__global__ void __launch_bounds__(256) cross_turn_kernel(CrossArgs a, CPS plans) {
  using T = typename CPS::T;
  using Mt = typename CPS::M;
  constexpr int E = 1 << R;
  const size_t col = blockIdx.y;
  const ColPlanT<Mt> &P = plans.l[((a.col0 + col) % a.S) / a.slots_per_limb];
  const Mt mod = P.mod;
  const T *__restrict__ win = static_cast<const T *>(a.W) + (col << a.logtot);
  T *__restrict__ wout = static_cast<T *>(a.dst) + (col << a.logtot);
  const T *__restrict__ tw = P.tw;
  const T *__restrict__ itw = P.itw;
  const int logB = a.logtot - R, B = 1 << logB;
  const int q = 2 * a.m - 2;
  const uint32_t imask = P.imask[a.logtot] >> logB, fmask = P.fmask[a.logtot];
  const int ngroups = B / V;
  for (int g = (int)(blockIdx.x * blockDim.x + threadIdx.x); g < ngroups; g += (int)(gridDim.x * blockDim.x)) {
    const int j = V * g;
    T v[V][E];
    int E0[V];
#pragma unroll
    for (int c = 0; c < V; c++) {
      const int i0 = q - j - c;  // >= 0: q >= M >= B > j + c
      const int rho = i0 & (B - 1);
      E0[c] = i0 >> logB;
#pragma unroll
      for (int e = 0; e < E; e++) {
#if RS_WORKSPACE_NT
        v[c][e] = __builtin_nontemporal_load(win + rho + e * B);
#else
        v[c][e] = win[rho + e * B];
#endif
      }
    }
    // inverse stages logB .. logB + R - 1 of the length-2^logtot transform (inv_round2 with u0 = logB, hi = 0)
#pragma unroll
    for (int k = 0; k < R; k++) {
      if ((imask >> k) & 1u) {
#pragma unroll
        for (int c = 0; c < V; c++)
#pragma unroll
          for (int e = 0; e < E; e++) v[c][e] = reduce(v[c][e], mod);
      }
      const int twbase = 1 << (R - 1 - k);
#pragma unroll
      for (int e = 0; e < E; e++) {
        if (e & (1 << k)) continue;
        const T w = itw[twbase + (e >> (k + 1))];
#pragma unroll
        for (int c = 0; c < V; c++) {
          const T x = v[c][e], y = v[c][e + (1 << k)];
          v[c][e] = addm(x, y, mod);
          v[c][e + (1 << k)] = mulmod(subm(x, y, mod), w, mod);
        }
      }
    }
    // T_{j + c + e' B} = P_{rho_c + (E0_c - e') B}: reverse, shift by E - 1 - E0_c, truncate at m - 1, reduce (CS_REV_TRUNC)
    T x[V][E];
#pragma unroll
    for (int c = 0; c < V; c++) {
      T u[E];
#pragma unroll
      for (int e = 0; e < E; e++) u[e] = v[c][E - 1 - e];
      const int sh = E - 1 - E0[c];  // 0 <= sh < E; entries shifted in from beyond the tile belong to k > 2m - 2: truncated below
#pragma unroll
      for (int b = 0; b < R; b++) {
        const bool on = (sh >> b) & 1;
#pragma unroll
        for (int e = 0; e < E; e++) {
          const T far = (e + (1 << b) < E) ? u[e + (1 << b)] : T(0);
          u[e] = on ? far : u[e];
        }
      }
#pragma unroll
      for (int e = 0; e < E / 2; e++) x[c][e] = (j + c + e * B < a.m - 1) ? reduce(u[e], mod) : T(0);
    }
    // forward stages 0 .. R - 1 on a zero-padded input (fwd_round2 with s0 = 0, hi = 0; stage 0 is a copy)
#pragma unroll
    for (int k = 0; k < R; k++) {
      if ((fmask >> k) & 1u) {
#pragma unroll
        for (int c = 0; c < V; c++)
#pragma unroll
          for (int e = 0; e < (k == 0 ? E / 2 : E); e++) x[c][e] = reduce(x[c][e], mod);
      }
      if (k == 0) {
#pragma unroll
        for (int c = 0; c < V; c++)
#pragma unroll
          for (int e = 0; e < E / 2; e++) x[c][e + E / 2] = x[c][e];
        continue;
      }
      const int half = E >> (k + 1);
#pragma unroll
      for (int blk = 0; blk < (1 << k); blk++) {
        const T w = tw[(1 << k) + blk];
#pragma unroll
        for (int e0 = 0; e0 < half; e0++) {
          const int ia = blk * 2 * half + e0, ib = ia + half;
#pragma unroll
          for (int c = 0; c < V; c++) {
            const T t = mulmod(x[c][ib], w, mod);
            const T z = x[c][ia];
            x[c][ia] = addm(z, t, mod);
            x[c][ib] = subm(z, t, mod);
          }
        }
      }
    }
    const GlobalIOT<T> out{wout};
#pragma unroll
    for (int e = 0; e < E; e++) {
      if (V == 2)
        out.store2(j + e * B, x[0][e], x[V - 1][e]);
      else
        out.store(j + e * B, 0, 0, 0, x[0][e]);
    }
  }
}

// ---- the turn between two tree levels (round 5): the last inverse cross pass of level l (sink CD_COMBINE: F_node = (F_left, 0)
// + D_left F_right) and the first forward cross pass of level l + 1 (source CS_FILL_RIGHT: the parent's input is its RIGHT
// child, zero padded), as one pass.  A thread owns a residue class j mod B of a PARENT node: the classes of both children
// (2^RL words each) are the class of the parent (2^(RL+1) words).  It finishes both children, writes the left child's
// coefficients (level l + 1 adds them back as F_left), keeps the right child's in registers -- nothing reads them from the
// columns again: level l + 1 overwrites the whole parent -- and runs the parent's forward stages on them, writing the
// parent's class of the workspace it has just read.  3 words of traffic per coefficient instead of 4.  Same stages, masks,
// twiddles and reductions as cross_kernel<true, RL, CD_COMBINE> + cross_kernel<false, RL + 1, CS_FILL_RIGHT> (knob
// witness_level_turn = 0 restores them).  Needs the same block size at both levels and one cross pass each.
template <int RL, class CPS, int V>
__global__ void __launch_bounds__(256) cross_level_turn_kernel(CrossArgs a, CPS plans) {
  using T = typename CPS::T;
  using Mt = typename CPS::M;
  constexpr int EL = 1 << RL, E = 2 * EL, R = RL + 1;
  const size_t col = blockIdx.y;
  const ColPlanT<Mt> &P = plans.l[((a.col0 + col) % a.S) / a.slots_per_limb];
  const Mt mod = P.mod;
  const GlobalIOT<T> w{static_cast<T *>(a.W) + (col << a.logtot)};
  T *__restrict__ xcol = static_cast<T *>(a.dst) + (col << a.logM);
  const T *__restrict__ tw = P.tw;
  const T *__restrict__ itw = P.itw;
  const int l = a.l;  // the CHILD level: nodes of 2^l coefficients; parents of 2^(l+1)
  const int logB = l - RL, B = 1 << logB;
  const uint32_t imask = P.imask[l] >> logB, fmask = P.fmask[l + 1];
  const int per_parent = B / V, ngroups = ((1 << a.logtot) >> (l + 1)) * per_parent;
  for (int g = (int)(blockIdx.x * blockDim.x + threadIdx.x); g < ngroups; g += (int)(gridDim.x * blockDim.x)) {
    const int pn = __builtin_amdgcn_readfirstlane(g / per_parent);  // B / V >= 128: a wave stays inside one parent
    const int j = V * (g - pn * per_parent);
    const int base = (pn << (l + 1)) + j;
    T v[2][V][EL];  // [child][word of the pair][e]
#pragma unroll
    for (int ch = 0; ch < 2; ch++)
#pragma unroll
      for (int e = 0; e < EL; e++) {
        const int k = base + (ch << l) + e * B;
        if (V == 2) {
          const Pair<T> x = w.load2(k);
          v[ch][0][e] = x.x;
          v[ch][V - 1][e] = x.y;
        } else {
          v[ch][0][e] = w.load(k, 0, 0, 0);
        }
      }
    // inverse stages logB .. l - 1 of both children's length-2^l transforms
#pragma unroll
    for (int k = 0; k < RL; k++) {
      if ((imask >> k) & 1u) {
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
#pragma unroll
          for (int c = 0; c < V; c++)
#pragma unroll
            for (int e = 0; e < EL; e++) v[ch][c][e] = reduce(v[ch][c][e], mod);
      }
      const int twbase = 1 << (RL - 1 - k);
#pragma unroll
      for (int e = 0; e < EL; e++) {
        if (e & (1 << k)) continue;
        const T wt = itw[twbase + (e >> (k + 1))];
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
#pragma unroll
          for (int c = 0; c < V; c++) {
            const T x = v[ch][c][e], y = v[ch][c][e + (1 << k)];
            v[ch][c][e] = addm(x, y, mod);
            v[ch][c][e + (1 << k)] = mulmod(subm(x, y, mod), wt, mod);
          }
      }
    }
    // CD_COMBINE: + F_left on the lower half of each child (e < EL / 2), reduce; the left child goes back to the columns
    T x[V][E];
#pragma unroll
    for (int ch = 0; ch < 2; ch++)
#pragma unroll
      for (int e = 0; e < EL; e++) {
        const int k = base + (ch << l) + e * B;
        T a0 = T(0), a1 = T(0);  // (F_left, 0): the lower half of the child
        if (e < EL / 2) {
          if (V == 2) {
            const Pair<T> fl = ld_pair(xcol + k);
            a0 = fl.x;
            a1 = fl.y;
          } else {
            a0 = xcol[k];
          }
        }
        T f0 = reduce(addm(v[ch][0][e], a0, mod), mod), f1 = T(0);
        if (V == 2) f1 = reduce(addm(v[ch][V - 1][e], a1, mod), mod);
        if (ch == 0) {
          if (V == 2)
            st_pair(xcol + k, f0, f1);
          else
            xcol[k] = f0;
        } else {
          x[0][e] = f0;
          if (V == 2) x[V - 1][e] = f1;
        }
      }
    // forward stages 0 .. RL of the parent's length-2^(l+1) transform on (F_right, 0): stage 0 is a copy
#pragma unroll
    for (int k = 0; k < R; k++) {
      if ((fmask >> k) & 1u) {
#pragma unroll
        for (int c = 0; c < V; c++)
#pragma unroll
          for (int e = 0; e < (k == 0 ? E / 2 : E); e++) x[c][e] = reduce(x[c][e], mod);
      }
      if (k == 0) {
#pragma unroll
        for (int c = 0; c < V; c++)
#pragma unroll
          for (int e = 0; e < E / 2; e++) x[c][e + E / 2] = x[c][e];
        continue;
      }
      const int half = E >> (k + 1);
#pragma unroll
      for (int blk = 0; blk < (1 << k); blk++) {
        const T wt = tw[(1 << k) + blk];
#pragma unroll
        for (int e0 = 0; e0 < half; e0++) {
          const int ia = blk * 2 * half + e0, ib = ia + half;
#pragma unroll
          for (int c = 0; c < V; c++) {
            const T t = mulmod(x[c][ib], wt, mod);
            const T z = x[c][ia];
            x[c][ia] = addm(z, t, mod);
            x[c][ib] = subm(z, t, mod);
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < E; e++) {
      if (V == 2)
        w.store2(base + e * B, x[0][e], x[V - 1][e]);
      else
        w.store(base + e * B, 0, 0, 0, x[0][e]);
    }
  }
}

// Sub-transforms on blocks of Bn = 2^logB doubles.  MODE 0: forward, 1: inverse, 2: forward,
// multiply by tab[(blk % tab_period) * Bn + j], inverse (fused); 3: like 2 with a per-column table
// (another workspace of the same shape, lazily reduced): tab[blk * Bn + j].  Block blk belongs to column
// blk / blocks_per_col; inside its transform (n1 = 2^log_n1 blocks) it is block blk % n1.
template <int MODE, class CPS>
__global__ void __launch_bounds__(1024)
sub_ntt_kernel(typename CPS::T *__restrict__ X, int logB, int log_n1, TabPtrs tabs, unsigned tab_period,
               unsigned blocks_per_col, size_t col0, unsigned S, unsigned slots_per_limb, CPS plans) {
  using T = typename CPS::T;
  using Mt = typename CPS::M;
  constexpr bool FP = std::is_same<Mt, Mod>::value;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *s = reinterpret_cast<T *>(smem);
  const int Bn = 1 << logB;
  const size_t blk = blockIdx.x;
  const size_t col = blk / blocks_per_col;
  const int limb = (int)(((col0 + col) % S) / slots_per_limb);
  const ColPlanT<Mt> &P = plans.l[limb];
  const Mt mod = P.mod;
  const int root = (1 << log_n1) + (int)(blk & ((1u << log_n1) - 1));
  const int logn = logB + log_n1;
  T *x = X + blk * (size_t)Bn;
  for (int i = threadIdx.x; i < Bn; i += blockDim.x) s[pidx(i)] = x[i];
  __syncthreads();
  // incomplete transform (witness_inc.hpp): this prime has no root of unity of order 2^logn -- the forward transform stops
  // `inc` stages early, the pointwise step is a product modulo x^(2^inc) - eta per leaf, the inverse starts at stage inc
  const int inc = P.inc(logn);
  if (inc > 0) {
    if (MODE == 0 || MODE >= 2) lds_ntt_fwd_part<3>(s, logB, logB - inc, P.tw, root, mod, P.fmask[logn] >> log_n1);
    if (MODE == 2 || MODE == 3) {
      const T *tab = MODE == 2 ? static_cast<const T *>(tabs.t[limb]) + (size_t)(blk % tab_period) * Bn
                               : static_cast<const T *>(tabs.t[0]) + blk * (size_t)Bn;
      inc_pointwise_tile<MODE == 3>(s, logB, inc, tab, P.tw, root, mod);
    }
    if (MODE >= 1) lds_ntt_inv_part<3>(s, logB, inc, P.itw, root, mod, P.imask[logn]);
    for (int i = threadIdx.x; i < Bn; i += blockDim.x) x[i] = s[pidx(i)];
    return;
  }
  int logw = 0;
  while ((64 << logw) < (int)blockDim.x) logw++;
  const bool wp = FP && logw >= 1 && logw <= 4 && logB - logw >= 8;
  if (MODE == 0 || MODE >= 2) {
    bool done = false;
    if constexpr (FP) {
      if (wp) {
        lds_ntt_fwd_wp<4, LdsIO, ColBlockFactory, 3>(s, LdsIO{s}, ColBlockFactory{s}, logB, logw, P.tw, mod, P.fmask[logn] >> log_n1, root);
        __syncthreads();
        done = true;
      }
    }
    if (!done) lds_ntt_fwd<3>(s, logB, P.tw, root, mod, P.fmask[logn] >> log_n1);
  }
  if (MODE == 2) {  // table of slot-constant spectra: a table constant
    const T *tab = static_cast<const T *>(tabs.t[limb]) + (size_t)(blk % tab_period) * Bn;
    for (int i = threadIdx.x; i < Bn; i += blockDim.x) s[pidx(i)] = mulmod(reduce(s[pidx(i)], mod), tab[i], mod);
    __syncthreads();
  }
  if (MODE == 3) {  // the other workspace: a spectrum computed on the device (data x data)
    const T *tab = static_cast<const T *>(tabs.t[0]) + blk * (size_t)Bn;
    for (int i = threadIdx.x; i < Bn; i += blockDim.x)
      s[pidx(i)] = mulmod_dd(reduce(s[pidx(i)], mod), reduce(tab[i], mod), mod);
    __syncthreads();
  }
  if (MODE == 4) {  // coset form of H: (spectrum * spectrum of A - spectrum of C) / Z on the coset (big_h_coset)
    const T *w1 = static_cast<const T *>(tabs.w1) + blk * (size_t)Bn, *w3 = static_cast<const T *>(tabs.w3) + blk * (size_t)Bn;
    const T *zi = static_cast<const T *>(tabs.t[limb]) + (size_t)(blk % tab_period) * Bn;
    for (int i = threadIdx.x; i < Bn; i += blockDim.x) {
      const T ab = mulmod_dd(reduce(s[pidx(i)], mod), reduce(w1[i], mod), mod);
      s[pidx(i)] = mulmod(reduce(subm(ab, reduce(w3[i], mod), mod), mod), zi[i], mod);
    }
    __syncthreads();
  }
  if (MODE >= 1) {
    bool done = false;
    if constexpr (FP) {
      if (wp) {
        lds_ntt_inv_wp<4, ColBlockFactory, LdsIO, 3>(s, ColBlockFactory{s}, LdsIO{s}, logB, logw, P.itw, mod, P.imask[logn], root);
        done = true;
      }
    }
    if (!done) lds_ntt_inv<3>(s, logB, P.itw, root, mod, P.imask[logn]);
  }
  for (int i = threadIdx.x; i < Bn; i += blockDim.x) x[i] = s[pidx(i)];
}

// Last forward round of a fused sub-transform: spectrum times a table that is itself a lazily reduced
// spectrum (MODE 3: the other workspace).
struct SubMulLazyOut {
  double *sb;
  const double *dh;
  Mod mod;
  __device__ __forceinline__ int pbase(int base) const { return pidx(base); }
  __device__ __forceinline__ void store(int base, int pb, int eoff, int poff, double v) const {
    sb[pcomb(pb, poff)] = mulmod(reduce(v, mod), reduce(dh[base + eoff], mod), mod);
  }
};
struct SubMulLazyFactory {
  double *s;
  const double *dh_tile;
  Mod mod;
  __device__ __forceinline__ SubMulLazyOut operator()(int off) const { return SubMulLazyOut{s + pidx(off), dh_tile + off, mod}; }
};

#ifndef RS_SUB_MAXR
#define RS_SUB_MAXR 4  // radix of the wave-private rounds of sub_ntt_ct_kernel
#endif
#ifdef RS_EXPERIMENTS  // superseded A/B variant (witness_sub_ct = 1): experiments build only
// sub_ntt_kernel for the production tile (Bn = 2^LOGB, compile time; 512 threads, two workgroups per CU):
//   * the cross-wave round of the forward transform reads the block straight from global memory and the
//     cross-wave round of the inverse writes it straight back (no staging pass, no extra barriers);
//   * the table product rides the last forward round's store (no separate pointwise pass);
//   * forward-only blocks (MODE 0) are stored by the wave that finished them.
// Same arithmetic and operation order per coefficient as sub_ntt_kernel: results are identical.
template <int MODE, int LOGB>
__global__ void __launch_bounds__(512, 4)
sub_ntt_ct_kernel(double *__restrict__ X, int log_n1, TabPtrs tabs, unsigned tab_period, unsigned blocks_per_col, size_t col0,
                  unsigned S, unsigned slots_per_limb, ColPlans plans) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  constexpr int LOGW = 3, Bn = 1 << LOGB;
  const size_t blk = blockIdx.x;
  const size_t col = blk / blocks_per_col;
  const int limb = (int)(((col0 + col) % S) / slots_per_limb);
  const ColPlan &P = plans.l[limb];
  const Mod mod = P.mod;
  const int root = (1 << log_n1) + (int)(blk & ((1u << log_n1) - 1));
  const int logn = LOGB + log_n1;
  double *x = X + blk * (size_t)Bn;
  const GlobalF64IO gio{x};
  const ColBlockFactory bf{s};
  const uint32_t fmask = P.fmask[logn] >> log_n1, imask = P.imask[logn];
  if (MODE == 0) {
    lds_ntt_fwd_wp<RS_SUB_MAXR, GlobalF64IO, ColBlockFactory, 3>(s, gio, bf, LOGB, LOGW, P.tw, mod, fmask, root);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int BS = Bn >> LOGW;
    const int off = wave * BS;
    const double *sb = s + pidx(off);
    const int p0 = pidx(lane);
#pragma unroll
    for (int j = 0; j < BS / 64; j++) x[off + lane + 64 * j] = sb[own_pidx(p0, lane, j)];
    return;
  }
  if (MODE == 2) {
    const double *tab = static_cast<const double *>(tabs.t[limb]) + (size_t)(blk % tab_period) * Bn;
    lds_ntt_fwd_wp<RS_SUB_MAXR, GlobalF64IO, TreeMulFactory, 3>(s, gio, TreeMulFactory{s, tab, mod}, LOGB, LOGW, P.tw, mod, fmask, root);
  } else {
    const double *tab = static_cast<const double *>(tabs.t[0]) + blk * (size_t)Bn;
    lds_ntt_fwd_wp<RS_SUB_MAXR, GlobalF64IO, SubMulLazyFactory, 3>(s, gio, SubMulLazyFactory{s, tab, mod}, LOGB, LOGW, P.tw, mod, fmask, root);
  }
  lds_ntt_inv_wp<RS_SUB_MAXR, ColBlockFactory, GlobalF64IO, 3>(s, bf, gio, LOGB, LOGW, P.itw, mod, imask, root);
}

#endif  // RS_EXPERIMENTS

// sub_ntt_ct_kernel in the wide form of ntt_wide.hpp (g_witness_sub_ct == 2): 256 threads x 32 coefficients per block of
// 2^13, persistent, two workgroups per CU.  Forward rounds (4, 5, 4 stages); round 3 leaves every thread with 16
// CONSECUTIVE spectrum points per group, which is exactly the operand set of the inverse's first round, so the table
// product and inverse stages 0..3 follow in registers: the fused forward-multiply-inverse exchanges the tile four
// times (eight LDS passes) instead of seven.  The twiddles of a block depend on its position in the long transform
// (root), so they are fetched per block from the L2-resident table.  Same stage arithmetic and reduction points as
// sub_ntt_ct_kernel: the stored (lazily reduced) values are identical.
struct SubTw {  // twiddle fetch: 2^k consecutive table entries, 16-byte loads where the run allows
  template <int CNT>
  __device__ static __forceinline__ void run(const double *__restrict__ p, double *dst) {
#ifdef RS_SUBW_ABLATE_TW  // experiment: no twiddle / table traffic (wrong results)
#pragma unroll
    for (int i = 0; i < CNT; i++) dst[i] = 3.0 + i + (double)threadIdx.x;
    return;
#endif
    if (CNT == 1) {
      dst[0] = p[0];
    } else {
#pragma unroll
      for (int i = 0; i < CNT / 2; i++) {
        const double2 v = reinterpret_cast<const double2 *>(p)[i];
        dst[2 * i] = v.x;
        dst[2 * i + 1] = v.y;
      }
    }
  }
};
// INC > 0: incomplete transforms (witness_inc.hpp) -- every column of the launch belongs to a prime whose transform of this
// length stops INC stages early: forward round 3 runs 4 - INC stages, the table product is a product of polynomials of
// 2^INC coefficients per leaf (the leaf's eta is +- the last twiddle its lane has just used), inverse round 1 starts at stage INC.
template <int MODE, int INC = 0>
__global__ void __launch_bounds__(256, 2)
sub_ntt_wide_kernel(double *__restrict__ X, int log_n1, TabPtrs tabs, unsigned tab_period, unsigned blocks_per_col, size_t col0,
                    unsigned S_, unsigned slots_per_limb, ColPlans plans, unsigned long long nblocks,
                    const double *__restrict__ Xsrc /* null: in place.  Else block b reads block b >> 1 of Xsrc: the two
                    sub-transforms (roots 2 and 3) of ONE zero-padded block of 2^13 coefficients (two-dimensional block convolutions) */) {
  using S = WideShape<13>;
  static_assert(MODE != 4 || INC == 0, "the coset form of H needs full-length transforms");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  const int t = threadIdx.x;
  u64x2 pre[16];
  auto issue_loads = [&](unsigned long long b) {
    const u64x2 *src = reinterpret_cast<const u64x2 *>(Xsrc ? Xsrc + (b >> 1) * (size_t)S::N : X + b * (size_t)S::N) + t;
#pragma unroll
    for (int e = 0; e < 16; e++) pre[e] = src[(S::S / 2) * e];
  };
  unsigned long long blk = blockIdx.x;
  if (blk < nblocks) issue_loads(blk);
  for (; blk < nblocks; blk += gridDim.x) {
    const size_t col = blk / blocks_per_col;
    const int limb = (int)(((col0 + col) % S_) / slots_per_limb);
    const ColPlan &P = plans.l[limb];
    const Mod mod = P.mod;
    const int root = (1 << log_n1) + (int)(blk & ((1u << log_n1) - 1));
    const int logn = 13 + log_n1;
    const uint32_t fmask = P.fmask[logn] >> log_n1, imask = P.imask[logn];
    const double *__restrict__ tw = P.tw;
    const double *__restrict__ itw = P.itw;
    double v[2][16];
#pragma unroll
    for (int e = 0; e < 16; e++) {
      v[0][e] = u64_bits_as_double(pre[e].x);
      v[1][e] = u64_bits_as_double(pre[e].y);
      pin(v[0][e]);
      pin(v[1][e]);
    }
    mem_fence();
    // the lane's round-2 twiddles are requested BEFORE the next block's prefetch: vector loads complete in order, so waiting
    // for them later (vmcnt(16)) leaves the sixteen prefetch loads in flight; round 1's wave-uniform twiddles come through
    // the scalar cache (ld_const) and wait for no vector load at all -- the prefetch has rounds 1 and 2 to land (round 6:
    // it used to be drained by the first twiddle of round 1, a vector load issued right after it)
    double w2[31];
    {
      const int hi = t >> 4;
      SubTw::run<1>(tw + (root << 4) + hi, w2);
      SubTw::run<2>(tw + (root << 5) + (hi << 1), w2 + 1);
      SubTw::run<4>(tw + (root << 6) + (hi << 2), w2 + 3);
      SubTw::run<8>(tw + (root << 7) + (hi << 3), w2 + 7);
      SubTw::run<16>(tw + (root << 8) + (hi << 4), w2 + 15);
    }
    mem_fence();
    const unsigned long long bn = blk + gridDim.x;
    if (bn < nblocks) issue_loads(bn);
    mem_fence();
    // ---- forward round 1: stages 0..3 on elements 2t+c + 512e, twiddles tw[2^k root + blk] (uniform)
#pragma unroll
    for (int c = 0; c < 2; c++)
      reg_fwd_stages<4, true>(v[c], mod, fmask, [&](int k, int b) { return ld_const(tw + (root << k) + b); });
    __syncthreads();  // the previous block's last-round reads of the tile are done
    {
      const int pb = S::px(2 * t);
#pragma unroll
      for (int e = 0; e < 16; e++) {
        s[pb + S::SP * e] = v[0][e];
        s[pb + S::SP * e + 1] = v[1][e];
      }
    }
    __syncthreads();
    // ---- forward round 2: stages 4..8 on hi*512 + lo + 16e
    {
      const int lo = t & 15, hi = t >> 4;
      const int pb = hi * S::SP + lo;
      double x[32];
#pragma unroll
      for (int e = 0; e < 32; e++) x[e] = s[pb + 17 * e];
      reg_fwd_stages<5, true>(x, mod, fmask >> 4, [&](int k, int b) { return w2[(1 << k) - 1 + b]; });
#pragma unroll
      for (int e = 0; e < 32; e++) s[pb + 17 * e] = x[e];
    }
    __syncthreads();
    // ---- forward round 3 (stages 9..12) on 16 consecutive points, table product, inverse round 1 (stages 0..3)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int g = t + 256 * j;
      const int pb = S::px(16 * g);
      double w[15];
      if (INC < 4) SubTw::run<1>(tw + (root << 9) + g, w);
      if (INC < 3) SubTw::run<2>(tw + (root << 10) + (g << 1), w + 1);
      if (INC < 2) SubTw::run<4>(tw + (root << 11) + (g << 2), w + 3);
      if (INC < 1) SubTw::run<8>(tw + (root << 12) + (g << 3), w + 7);
      if (INC == 4) w[0] = tw[((root << 9) + g) >> 1];  // the parent of leaf g: its twiddle is the square root of the leaf's eta
      double x[16];
#pragma unroll
      for (int e = 0; e < 16; e++) x[e] = s[pb + e];
      reg_fwd_stages<4, true, 4 - INC>(x, mod, fmask >> 9, [&](int k, int b) { return w[(1 << k) - 1 + b]; });
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 16; e++) s[pb + e] = x[e];
        continue;
      }
      {
        // The table entries of the wave's 64 groups are 1024 consecutive words: fetched with coalesced 16-byte loads
        // and handed to their owners through the wave's range of the tile, which is free once x has been read (a
        // thread fetching its own 128-byte run touches 64 different lines per instruction).
        const int wave = t >> 6, lane = t & 63;
        const int r0 = (j * 256 + wave * 64) * 16;
        const int p0 = S::px(r0 + 2 * lane);
        auto stage = [&](const double *tab) {  // the wave's 1024 entries of `tab` -> its range of the tile
          const double2 *t2 = reinterpret_cast<const double2 *>(tab + r0) + lane;
#pragma unroll
          for (int i = 0; i < 8; i++) {
#ifdef RS_SUBW_ABLATE_TW
            const double2 v2 = make_double2(3.0 + i, 5.0 + lane);
#else
            const double2 v2 = t2[64 * i];
#endif
            s[p0 + S::px128(i)] = v2.x;
            s[p0 + S::px128(i) + 1] = v2.y;
          }
          wave_sync();
        };
        if (MODE == 4) {  // (x * spectrum of A - spectrum of C) / Z on the coset: three staged operands
          stage(static_cast<const double *>(tabs.w1) + blk * (size_t)S::N);
#pragma unroll
          for (int e = 0; e < 16; e++) x[e] = mulmod(reduce(x[e], mod), reduce(s[pb + e], mod), mod);
          wave_sync();
          stage(static_cast<const double *>(tabs.w3) + blk * (size_t)S::N);
#pragma unroll
          for (int e = 0; e < 16; e++) x[e] = reduce(x[e] - reduce(s[pb + e], mod), mod);
          wave_sync();
          stage(static_cast<const double *>(tabs.t[limb]) + (size_t)(blk % tab_period) * S::N);
#pragma unroll
          for (int e = 0; e < 16; e++) x[e] = mulmod(x[e], s[pb + e], mod);
        } else {
          stage((MODE == 2) ? static_cast<const double *>(tabs.t[limb]) + (size_t)(blk % tab_period) * S::N
                            : static_cast<const double *>(tabs.t[0]) + blk * (size_t)S::N);
          if (INC > 0) {  // 16 >> INC leaves of 2^INC words: products modulo x^(2^INC) - eta (witness_inc.hpp)
            constexpr int G = 1 << INC;
#pragma unroll
            for (int q = 0; q < 16 / G; q++) {
              double xq[G], tq[G];
#pragma unroll
              for (int e = 0; e < G; e++) {
                xq[e] = reduce(x[q * G + e], mod);
                tq[e] = MODE == 2 ? s[pb + q * G + e] : reduce(s[pb + q * G + e], mod);
              }
              const double wq = INC == 4 ? w[0] : w[(1 << (3 - (INC & 3))) - 1 + (q >> 1)];
              const bool neg = INC == 4 ? (g & 1) : (q & 1);
              inc_polymul<INC, MODE != 2>(xq, tq, neg ? -wq : wq, mod);
#pragma unroll
              for (int e = 0; e < G; e++) x[q * G + e] = xq[e];
            }
          } else if (MODE == 2) {
            if ((P.pwmask >> logn) & 1u) {  // primes above ~2^46 only (a guarded pass, not a select)
#pragma unroll
              for (int e = 0; e < 16; e++) x[e] = reduce(x[e], mod);
            }
#pragma unroll
            for (int e = 0; e < 16; e++) x[e] = mulmod(x[e], s[pb + e], mod);
          } else {
#pragma unroll
            for (int e = 0; e < 16; e++) x[e] = mulmod(reduce(x[e], mod), reduce(s[pb + e], mod), mod);
          }
        }
      }
      // inverse stage k of the block: twiddle itw[(n >> (k+1)) root + (position >> (k+1))]
      if (INC < 1) SubTw::run<8>(itw + ((size_t)root << 12) + (g << 3), w);
      if (INC < 2) SubTw::run<4>(itw + ((size_t)root << 11) + (g << 2), w + 8);
      if (INC < 3) SubTw::run<2>(itw + ((size_t)root << 10) + (g << 1), w + 12);
      if (INC < 4) SubTw::run<1>(itw + ((size_t)root << 9) + g, w + 14);
      reg_inv_stages<4, true, 4, INC>(x, mod, imask, [&](int k, int i) { return w[16 - (16 >> k) + i]; });
#pragma unroll
      for (int e = 0; e < 16; e++) s[pb + e] = x[e];
    }
    if (MODE == 0) {  // forward only: every wave streams out the ranges its own groups cover
      wave_sync();
      const int wave = t >> 6, lane = t & 63;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int r0 = (j * 256 + wave * 64) * 16;
        const int p0 = S::px(r0 + 2 * lane);
        double2 *d2 = reinterpret_cast<double2 *>(X + blk * (size_t)S::N + r0) + lane;
#pragma unroll
        for (int i = 0; i < 8; i++) d2[64 * i] = make_double2(s[p0 + S::px128(i)], s[p0 + S::px128(i) + 1]);
      }
      continue;
    }
    __syncthreads();
    // ---- inverse round 2: stages 4..8; block of stage 4+k: (hi << (4-k)) + (e >> (k+1))
    {
      const int lo = t & 15, hi = t >> 4;
      const int pb = hi * S::SP + lo;
      double w[31];
      SubTw::run<16>(itw + ((size_t)root << 8) + (hi << 4), w);
      SubTw::run<8>(itw + ((size_t)root << 7) + (hi << 3), w + 16);
      SubTw::run<4>(itw + ((size_t)root << 6) + (hi << 2), w + 24);
      SubTw::run<2>(itw + ((size_t)root << 5) + (hi << 1), w + 28);
      SubTw::run<1>(itw + ((size_t)root << 4) + hi, w + 30);
      double x[32];
#pragma unroll
      for (int e = 0; e < 32; e++) x[e] = s[pb + 17 * e];
      reg_inv_stages<5, true>(x, mod, imask >> 4, [&](int k, int i) { return w[32 - (32 >> k) + i]; });
#pragma unroll
      for (int e = 0; e < 32; e++) s[pb + 17 * e] = x[e];
    }
    __syncthreads();
    // ---- inverse round 3: stages 9..12 on elements 2t+c + 512e; block of stage 9+k: e >> (k+1) of 8 >> k
    {
      const int pb = S::px(2 * t);
#pragma unroll
      for (int e = 0; e < 16; e++) {
        v[0][e] = s[pb + S::SP * e];
        v[1][e] = s[pb + S::SP * e + 1];
      }
    }
#pragma unroll
    for (int c = 0; c < 2; c++)
      reg_inv_stages<4, true>(v[c], mod, imask >> 9, [&](int k, int i) { return ld_const(itw + ((8 >> k) * root) + i); });
    {
      double2 *dst = reinterpret_cast<double2 *>(X + blk * (size_t)S::N) + t;
#pragma unroll
      for (int e = 0; e < 16; e++) dst[(S::S / 2) * e] = make_double2(v[0][e], v[1][e]);
    }
  }
}

// ---- rooted sub-transforms on blocks of 2^12: 256 threads x 16 coefficients -------------------------------------------
// One block of 2^12 in the wide form with 16 coefficients per thread: rounds of 4 | 4 | 4 stages each way; the last
// forward round leaves 16 consecutive spectrum points per thread, the operand set of the inverse's first round, so the
// fused forward-multiply-inverse exchanges the 34 KiB tile four times, like the 2^13 kernel's 4 | 5 | 4.  In: v[e] =
// element t + 256 e.  Out (MODE >= 2): v[e] = element t + 256 e after the 12 inverse stages; MODE 0: the spectrum is
// written to fwd_out (the wave that finished a range streams it out) and v is dead.  `root`: tree node of the block in
// the long transform; fmask / imask: reduction bits of its 12 stages; tab: the 4096 table entries of this block.
template <int MODE, int INC = 0>
__device__ __forceinline__ void w12_block(double (&v)[16], double *s, const double *__restrict__ tw, const double *__restrict__ itw,
                                          const Mod mod, int root, uint32_t fmask, uint32_t imask, bool pw_reduce,
                                          const double *__restrict__ tab, double *__restrict__ fwd_out,
                                          const double *__restrict__ w1 = nullptr, const double *__restrict__ w3 = nullptr) {
  constexpr int SP = 272;  // px(i) = i + (i >> 4); elements 256 apart are 272 slots apart
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int pt = t + (t >> 4);
  // ---- forward round 1: stages 0..3 on elements t + 256 e (uniform twiddles)
  reg_fwd_stages<4, true>(v, mod, fmask, [&](int k, int b) { return ld_const(tw + (root << k) + b); });  // wave-uniform: scalar cache (ntt_wide.hpp)
  __syncthreads();  // the previous block's last-round reads of the tile are done
#pragma unroll
  for (int e = 0; e < 16; e++) s[pt + SP * e] = v[e];
  __syncthreads();
  // ---- forward round 2: stages 4..7 on hi*256 + lo + 16 e
  {
    const int lo = t & 15, hi = t >> 4;
    const int pb = hi * SP + lo;
    double w[15];
    SubTw::run<1>(tw + (root << 4) + hi, w);
    SubTw::run<2>(tw + (root << 5) + (hi << 1), w + 1);
    SubTw::run<4>(tw + (root << 6) + (hi << 2), w + 3);
    SubTw::run<8>(tw + (root << 7) + (hi << 3), w + 7);
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = s[pb + 17 * e];
    reg_fwd_stages<4, true>(v, mod, fmask >> 4, [&](int k, int b) { return w[(1 << k) - 1 + b]; });
#pragma unroll
    for (int e = 0; e < 16; e++) s[pb + 17 * e] = v[e];
  }
  __syncthreads();
  // ---- forward round 3 (stages 8..11) on the 16 consecutive points 16 t .., table product, inverse round 1 (stages 0..3)
  const int pb3 = 17 * t;
  {
    static_assert(MODE != 4 || INC == 0, "the coset form of H needs full-length transforms");
    double w[15];
    if (INC < 4) SubTw::run<1>(tw + (root << 8) + t, w);
    if (INC < 3) SubTw::run<2>(tw + (root << 9) + (t << 1), w + 1);
    if (INC < 2) SubTw::run<4>(tw + (root << 10) + (t << 2), w + 3);
    if (INC < 1) SubTw::run<8>(tw + (root << 11) + (t << 3), w + 7);
    if (INC == 4) w[0] = tw[((root << 8) + t) >> 1];  // the parent of leaf t (witness_inc.hpp)
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = s[pb3 + e];
    reg_fwd_stages<4, true, 4 - INC>(v, mod, fmask >> 8, [&](int k, int b) { return w[(1 << k) - 1 + b]; });
    const int r0 = wave * 1024;                              // the wave's 64 groups: 1024 consecutive points
    const int p0 = r0 + (r0 >> 4) + 2 * lane + (lane >> 3);  // px(r0 + 2 lane)
    if (MODE == 0) {  // forward only: the wave streams its own range out
#pragma unroll
      for (int e = 0; e < 16; e++) s[pb3 + e] = v[e];
      wave_sync();
      double2 *d2 = reinterpret_cast<double2 *>(fwd_out + r0) + lane;
#pragma unroll
      for (int i = 0; i < 8; i++) d2[64 * i] = make_double2(s[p0 + 136 * i], s[p0 + 136 * i + 1]);
      return;
    }
    {  // table entries of the wave's range: coalesced 16-byte loads, handed to their owners through the wave's part of the tile
      auto stage = [&](const double *__restrict__ tb) {
        const double2 *t2 = reinterpret_cast<const double2 *>(tb + r0) + lane;
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const double2 v2 = t2[64 * i];
          s[p0 + 136 * i] = v2.x;
          s[p0 + 136 * i + 1] = v2.y;
        }
        wave_sync();
      };
      if (MODE == 4) {  // (x * spectrum of A - spectrum of C) / Z on the coset: three staged operands
        stage(w1);
#pragma unroll
        for (int e = 0; e < 16; e++) v[e] = mulmod(reduce(v[e], mod), reduce(s[pb3 + e], mod), mod);
        wave_sync();
        stage(w3);
#pragma unroll
        for (int e = 0; e < 16; e++) v[e] = reduce(v[e] - reduce(s[pb3 + e], mod), mod);
        wave_sync();
        stage(tab);
#pragma unroll
        for (int e = 0; e < 16; e++) v[e] = mulmod(v[e], s[pb3 + e], mod);
      } else {
        stage(tab);
        if (INC > 0) {  // 16 >> INC leaves of 2^INC words: products modulo x^(2^INC) - eta (witness_inc.hpp)
          constexpr int G = 1 << INC;
#pragma unroll
          for (int q = 0; q < 16 / G; q++) {
            double xq[G], tq[G];
#pragma unroll
            for (int e = 0; e < G; e++) {
              xq[e] = reduce(v[q * G + e], mod);
              tq[e] = MODE == 2 ? s[pb3 + q * G + e] : reduce(s[pb3 + q * G + e], mod);
            }
            const double wq = INC == 4 ? w[0] : w[(1 << (3 - (INC & 3))) - 1 + (q >> 1)];
            const bool neg = INC == 4 ? (t & 1) : (q & 1);
            inc_polymul<INC, MODE != 2>(xq, tq, neg ? -wq : wq, mod);
#pragma unroll
            for (int e = 0; e < G; e++) v[q * G + e] = xq[e];
          }
        } else if (MODE == 2) {
          if (pw_reduce) {  // primes above ~2^46 only (a guarded pass, not a select)
#pragma unroll
            for (int e = 0; e < 16; e++) v[e] = reduce(v[e], mod);
          }
#pragma unroll
          for (int e = 0; e < 16; e++) v[e] = mulmod(v[e], s[pb3 + e], mod);
        } else {
#pragma unroll
          for (int e = 0; e < 16; e++) v[e] = mulmod(reduce(v[e], mod), reduce(s[pb3 + e], mod), mod);
        }
      }
    }
    if (INC < 1) SubTw::run<8>(itw + ((size_t)root << 11) + (t << 3), w);
    if (INC < 2) SubTw::run<4>(itw + ((size_t)root << 10) + (t << 2), w + 8);
    if (INC < 3) SubTw::run<2>(itw + ((size_t)root << 9) + (t << 1), w + 12);
    if (INC < 4) SubTw::run<1>(itw + ((size_t)root << 8) + t, w + 14);
    reg_inv_stages<4, true, 4, INC>(v, mod, imask, [&](int k, int i) { return w[16 - (16 >> k) + i]; });
#pragma unroll
    for (int e = 0; e < 16; e++) s[pb3 + e] = v[e];
  }
  __syncthreads();
  // ---- inverse round 2: stages 4..7; block of stage 4+k: (hi << (3-k)) + (e >> (k+1))
  {
    const int lo = t & 15, hi = t >> 4;
    const int pb = hi * SP + lo;
    double w[15];
    SubTw::run<8>(itw + ((size_t)root << 7) + (hi << 3), w);
    SubTw::run<4>(itw + ((size_t)root << 6) + (hi << 2), w + 8);
    SubTw::run<2>(itw + ((size_t)root << 5) + (hi << 1), w + 12);
    SubTw::run<1>(itw + ((size_t)root << 4) + hi, w + 14);
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = s[pb + 17 * e];
    reg_inv_stages<4, true>(v, mod, imask >> 4, [&](int k, int i) { return w[16 - (16 >> k) + i]; });
#pragma unroll
    for (int e = 0; e < 16; e++) s[pb + 17 * e] = v[e];
  }
  __syncthreads();
  // ---- inverse round 3: stages 8..11 on elements t + 256 e; block of stage 8+k: e >> (k+1) of 8 >> k
#pragma unroll
  for (int e = 0; e < 16; e++) v[e] = s[pt + SP * e];
  reg_inv_stages<4, true>(v, mod, imask >> 8, [&](int k, int i) { return ld_const(itw + ((8 >> k) * root) + i); });
}

// sub_ntt_wide_kernel on blocks of 2^12 (knob witness_sub_log = 12): 116-128 registers, a 34 KiB tile -- FOUR workgroups
// (16 waves, four per SIMD) per CU instead of two of 2^13 at two waves per SIMD.  The transform one level up gets one more
// cross stage (still one pass over the workspace).  Stage arithmetic, reduction points and table products per coefficient
// are those of the 2^13 kernel: the stored values are identical.
template <int MODE, int INC = 0>
__global__ void __launch_bounds__(256, 4)
sub_ntt_w12_kernel(double *__restrict__ X, int log_n1, TabPtrs tabs, unsigned tab_period, unsigned blocks_per_col, size_t col0,
                   unsigned S_, unsigned slots_per_limb, ColPlans plans, unsigned long long nblocks) {
  constexpr int N = 4096;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  const int t = threadIdx.x;
  for (unsigned long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const size_t col = blk / blocks_per_col;
    const int limb = (int)(((col0 + col) % S_) / slots_per_limb);
    const ColPlan &P = plans.l[limb];
    const int root = (1 << log_n1) + (int)(blk & ((1u << log_n1) - 1));
    const int logn = 12 + log_n1;
    double *xb = X + blk * (size_t)N;
    double v[16];
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = __builtin_nontemporal_load(xb + t + 256 * e);
    const double *tab = (MODE == 2 || MODE == 4) ? static_cast<const double *>(tabs.t[limb]) + (size_t)(blk % tab_period) * N
                        : MODE == 3              ? static_cast<const double *>(tabs.t[0]) + blk * (size_t)N
                                                 : nullptr;
    w12_block<MODE, INC>(v, s, P.tw, P.itw, P.mod, root, P.fmask[logn] >> log_n1, P.imask[logn], (P.pwmask >> logn) & 1u, tab, xb,
                    MODE == 4 ? static_cast<const double *>(tabs.w1) + blk * (size_t)N : nullptr,
                    MODE == 4 ? static_cast<const double *>(tabs.w3) + blk * (size_t)N : nullptr);
    if (MODE == 0) continue;
#pragma unroll
    for (int e = 0; e < 16; e++) __builtin_nontemporal_store(v[e], xb + t + 256 * e);
  }
}

#ifdef RS_EXPERIMENTS  // superseded A/B variant (witness_sub_ct = 3, measured 11 % slower): experiments build only
// sub_ntt_wide_kernel at FOUR waves per SIMD (g_witness_sub_ct == 3): 512 threads x 16 coefficients per block of 2^13,
// <= 128 registers, two workgroups (16 waves) per CU.  Forward rounds of 4, 3 and 2 stages, then the same fused middle
// as the 32-coefficient form on 16 consecutive points (forward stages 9..12, table product, inverse stages 0..3), then
// the mirror image: six tile exchanges instead of four, twice the waves to hide them behind.  Same stages, reduction
// points and products: identical stored values.
template <int MODE>
__global__ void __launch_bounds__(512, 4)
sub_ntt_wide16_kernel(double *__restrict__ X, int log_n1, TabPtrs tabs, unsigned tab_period, unsigned blocks_per_col, size_t col0,
                      unsigned S_, unsigned slots_per_limb, ColPlans plans, unsigned long long nblocks) {
  using S = WideShape<13>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int pt = t + (t >> 4);  // px(t), t < 512
  double pre[16];
  auto issue_loads = [&](unsigned long long b) {
    const double *src = X + b * (size_t)S::N + t;
#pragma unroll
    for (int e = 0; e < 16; e++) pre[e] = src[512 * e];
  };
  unsigned long long blk = blockIdx.x;
  if (blk < nblocks) issue_loads(blk);
  for (; blk < nblocks; blk += gridDim.x) {
    const size_t col = blk / blocks_per_col;
    const int limb = (int)(((col0 + col) % S_) / slots_per_limb);
    const ColPlan &P = plans.l[limb];
    const Mod mod = P.mod;
    const int root = (1 << log_n1) + (int)(blk & ((1u << log_n1) - 1));
    const int logn = 13 + log_n1;
    const uint32_t fmask = P.fmask[logn] >> log_n1, imask = P.imask[logn];
    const double *__restrict__ tw = P.tw;
    const double *__restrict__ itw = P.itw;
    double v[16];
#pragma unroll
    for (int e = 0; e < 16; e++) {
      v[e] = pre[e];
      pin(v[e]);
    }
    mem_fence();
    const unsigned long long bn = blk + gridDim.x;
    if (bn < nblocks) issue_loads(bn);
    mem_fence();
    // ---- forward round 1: stages 0..3 on elements t + 512 e (uniform twiddles)
    reg_fwd_stages<4, true>(v, mod, fmask, [&](int k, int b) { return tw[(root << k) + b]; });
    __syncthreads();  // the previous block's last-round reads of the tile are done
#pragma unroll
    for (int e = 0; e < 16; e++) s[pt + S::SP * e] = v[e];
    __syncthreads();
    // ---- forward round 2: stages 4..6 on hi*512 + lo + 64 e; hi = wave + 8 j is wave-uniform
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int hi = __builtin_amdgcn_readfirstlane(wave + 8 * j);
      const int pb = hi * S::SP + lane + (lane >> 4);
      double x[8];
#pragma unroll
      for (int e = 0; e < 8; e++) x[e] = s[pb + 68 * e];
      reg_fwd_stages<3, true>(x, mod, fmask >> 4, [&](int k, int b) { return tw[(root << (4 + k)) + (hi << k) + b]; });
#pragma unroll
      for (int e = 0; e < 8; e++) s[pb + 68 * e] = x[e];
    }
    __syncthreads();
    // ---- forward round 3: stages 7..8 on hi*64 + lo + 16 e
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int g = t + 512 * j, lo = g & 15, hi = g >> 4;
      const int pb = hi * 68 + (hi >> 3) * 16 + lo;
      double x[4];
#pragma unroll
      for (int e = 0; e < 4; e++) x[e] = s[pb + 17 * e];
      const double w0 = tw[(root << 7) + hi];
      const double2 w12 = reinterpret_cast<const double2 *>(tw + (root << 8) + (hi << 1))[0];
      reg_fwd_stages<2, true>(x, mod, fmask >> 7, [&](int k, int b) { return k == 0 ? w0 : (b == 0 ? w12.x : w12.y); });
#pragma unroll
      for (int e = 0; e < 4; e++) s[pb + 17 * e] = x[e];
    }
    __syncthreads();
    // ---- the middle on the 16 consecutive points of group t: forward stages 9..12, table product, inverse stages 0..3
    {
      const int pb = 17 * t + (t >> 5) * 16;  // px(16 t)
      double w[15];
      SubTw::run<1>(tw + (root << 9) + t, w);
      SubTw::run<2>(tw + (root << 10) + (t << 1), w + 1);
      SubTw::run<4>(tw + (root << 11) + (t << 2), w + 3);
      SubTw::run<8>(tw + (root << 12) + (t << 3), w + 7);
      double x[16];
#pragma unroll
      for (int e = 0; e < 16; e++) x[e] = s[pb + e];
      reg_fwd_stages<4, true>(x, mod, fmask >> 9, [&](int k, int b) { return w[(1 << k) - 1 + b]; });
      if (MODE != 0) {
        // the wave's 64 groups are 1024 consecutive table words: coalesced loads, handed over through its (free) range
        const double *tab = (MODE == 2) ? static_cast<const double *>(tabs.t[limb]) + (size_t)(blk % tab_period) * S::N
                                        : static_cast<const double *>(tabs.t[0]) + blk * (size_t)S::N;
        const int r0 = wave * 1024;
        const int p0 = S::px(r0 + 2 * lane);
        const double2 *t2 = reinterpret_cast<const double2 *>(tab + r0) + lane;
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const double2 v2 = t2[64 * i];
          s[p0 + S::px128(i)] = v2.x;
          s[p0 + S::px128(i) + 1] = v2.y;
        }
        wave_sync();
        if (MODE == 2) {
#pragma unroll
          for (int e = 0; e < 16; e++) x[e] = mulmod(reduce(x[e], mod), s[pb + e], mod);
        } else {
#pragma unroll
          for (int e = 0; e < 16; e++) x[e] = mulmod(reduce(x[e], mod), reduce(s[pb + e], mod), mod);
        }
        SubTw::run<8>(itw + ((size_t)root << 12) + (t << 3), w);
        SubTw::run<4>(itw + ((size_t)root << 11) + (t << 2), w + 8);
        SubTw::run<2>(itw + ((size_t)root << 10) + (t << 1), w + 12);
        SubTw::run<1>(itw + ((size_t)root << 9) + t, w + 14);
        reg_inv_stages<4, true>(x, mod, imask, [&](int k, int i) { return w[16 - (16 >> k) + i]; });
      }
#pragma unroll
      for (int e = 0; e < 16; e++) s[pb + e] = x[e];
    }
    if (MODE == 0) {  // forward only: every wave streams out the 1024 points its own groups cover
      wave_sync();
      const int r0 = wave * 1024;
      const int p0 = S::px(r0 + 2 * lane);
      double2 *d2 = reinterpret_cast<double2 *>(X + blk * (size_t)S::N + r0) + lane;
#pragma unroll
      for (int i = 0; i < 8; i++) d2[64 * i] = make_double2(s[p0 + S::px128(i)], s[p0 + S::px128(i) + 1]);
      continue;
    }
    __syncthreads();
    // ---- inverse round 3: stages 4..5 on hi*64 + lo + 16 e; block of stage 4+k: (hi << (1-k)) + (e >> (k+1)) of 256 >> k
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int g = t + 512 * j, lo = g & 15, hi = g >> 4;
      const int pb = hi * 68 + (hi >> 3) * 16 + lo;
      double x[4];
#pragma unroll
      for (int e = 0; e < 4; e++) x[e] = s[pb + 17 * e];
      const double2 w01 = reinterpret_cast<const double2 *>(itw + ((size_t)root << 8) + (hi << 1))[0];
      const double w2 = itw[((size_t)root << 7) + hi];
      reg_inv_stages<2, true>(x, mod, imask >> 4, [&](int k, int i) { return k == 0 ? (i == 0 ? w01.x : w01.y) : w2; });
#pragma unroll
      for (int e = 0; e < 4; e++) s[pb + 17 * e] = x[e];
    }
    __syncthreads();
    // ---- inverse round 2: stages 6..8 on hi*512 + lo + 64 e; block of stage 6+k: (hi << (2-k)) + (e >> (k+1)) of 64 >> k
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int hi = __builtin_amdgcn_readfirstlane(wave + 8 * j);
      const int pb = hi * S::SP + lane + (lane >> 4);
      double x[8];
#pragma unroll
      for (int e = 0; e < 8; e++) x[e] = s[pb + 68 * e];
      reg_inv_stages<3, true>(x, mod, imask >> 6, [&](int k, int i) { return itw[((size_t)root << (6 - k)) + (hi << (2 - k)) + i]; });
#pragma unroll
      for (int e = 0; e < 8; e++) s[pb + 68 * e] = x[e];
    }
    __syncthreads();
    // ---- inverse round 1: stages 9..12 on elements t + 512 e; block of stage 9+k: e >> (k+1) of 8 >> k
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = s[pt + S::SP * e];
    reg_inv_stages<4, true>(v, mod, imask >> 9, [&](int k, int i) { return itw[((8 >> k) * root) + i]; });
    {
      double *dst = X + blk * (size_t)S::N + t;
#pragma unroll
      for (int e = 0; e < 16; e++) dst[512 * e] = v[e];
    }
  }
}
#endif  // RS_EXPERIMENTS

// ZK patch of the multi-pass H: H += d2*A + d1*B + d1*d2*Z, H[0] -= d3; then canonical form.
template <class CPS>
__global__ void __launch_bounds__(256)
h_patch_kernel(typename CPS::T *__restrict__ H, const typename CPS::T *__restrict__ A, const typename CPS::T *__restrict__ B, int logM,
               int m, size_t cols, size_t col0, unsigned S, unsigned slots_per_limb, CPS plans, const uint64_t *__restrict__ d1,
               const uint64_t *__restrict__ d2, const uint64_t *__restrict__ d3, ColMap cm) {
  using T = typename CPS::T;
  const size_t M = (size_t)1 << logM, total = cols * M, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t col = i / M, k = i % M, gcol = (col0 + col) % S;
    const ColPlanT<typename CPS::M> &P = plans.l[gcol / slots_per_limb];
    const typename CPS::M mod = P.mod;
    T h = ((long long)k <= (long long)m - 2) ? H[i] : T(0);
    if (d1) {
      int dlimb, dslot;
      cm.locate(gcol, dlimb, dslot);
      const size_t di = cm.in_index(dlimb, dslot);
      const T e1 = center(from_res<T>(d1[di]), mod), e2 = center(from_res<T>(d2[di]), mod);
      h = addm(h, addm(addm(mulmod_dd(e2, center(A[i], mod), mod), mulmod_dd(e1, center(B[i], mod), mod), mod),
                       mulmod(mulmod_dd(e1, e2, mod), P.ztab[k], mod), mod), mod);
      if (k == 0) h = subm(h, center(from_res<T>(d3[di]), mod), mod);
    }
    H[i] = canon(h, mod);
  }
}

}  // namespace rs
