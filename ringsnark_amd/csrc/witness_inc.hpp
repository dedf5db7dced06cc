// witness_inc.hpp -- incomplete transforms: full-length cyclic convolutions over primes WITHOUT a root of unity of that order
#pragma once
#include "ntt_core.hpp"

namespace rs {

// =============================================================================================
// The reference's recipe for the ring primes (default_double_batching_modulus, seal/seal_util.hpp:20-32) only makes
// q_i = 1 mod 2 N_inner: 2-adicity a = 14 or 15 at the headline shape, while the witness map of 2^16 constraints multiplies
// polynomials of 2^17 coefficients.  A cyclic transform of length 2^n splits x^(2^n) - 1 down the decimation tree
//     x^(2s) - eta  =  (x^s - w)(x^s + w),   w^2 = eta,
// and stage d needs a primitive 2^(d+1)-th root of unity only -- whatever n is.  With 2-adicity a < n the first a stages
// exist (same twiddle table tw[node], same butterflies, same cross passes as a full transform) and leave the residues of
// the input modulo the 2^a polynomials  x^G - eta_g,  G = 2^inc, inc = n - a: G CONSECUTIVE words per leaf g, with
//     eta_g = +tw[node >> 1] (g even), -tw[node >> 1] (g odd),   node = 2^a + g   (the parent's twiddle is its square root).
// The "pointwise" product of two such spectra is the product of two polynomials of G coefficients modulo x^G - eta_g
// (inc_polymul below: G^2 modular products instead of G), and the inverse transform starts at stage inc (gap G).  By the
// Chinese remainder theorem the result is the cyclic convolution of length 2^n -- exact, hence bit-identical to what
// full-length transforms or the block convolutions return.  Against the two-dimensional block convolutions this path
// replaces (witness_bc.hpp: twice the sub-transform work, 1.4 x the traffic, transforms across blocks) it costs
// inc stages LESS per transform and (G - 1) more products per coefficient in the pointwise step; nothing else changes:
// no padding, no extra pass over memory, every fused source / sink / turn of the full-length path is kept.
// =============================================================================================
constexpr int RS_INC_MAX = 4;  // G <= 16: a thread of the tuned sub-transform kernels holds 16 consecutive words

// x <- x * t  in  Z_p[X] / (X^G - eta),  G = 2^INC.
// FP64: x reduced (|x| <= p/2 + slack), t balanced table constants (DD: lazily reduced data, reduced here); result reduced.
// Sums of at most four products (|.| <= 0.75 p each) are taken before a reduction: |acc| <= 0.51 p + 3 p < 2^52 for p < 2^50.
template <int INC, bool DD, class T, class M>
__device__ __forceinline__ void inc_polymul(T (&x)[1 << INC], const T (&t)[1 << INC], const T eta, const M mod) {
  constexpr int G = 1 << INC;
  T o[G];
#pragma unroll
  for (int k = 0; k < G; k++) {
    T lo = T(0), hi = T(0);
    int cl = 0, ch = 0;
#pragma unroll
    for (int i = 0; i < G; i++) {
      const int j = (k - i) & (G - 1);
      const T pr = DD ? mulmod_dd(x[i], t[j], mod) : mulmod(x[i], t[j], mod);
      if (i <= k) {
        lo = cl ? addm(lo, pr, mod) : pr;
        if (++cl == 4 && i < k) {
          lo = reduce(lo, mod);
          cl = 1;
        }
      } else {
        hi = ch ? addm(hi, pr, mod) : pr;
        if (++ch == 4 && i < G - 1) {
          hi = reduce(hi, mod);
          ch = 1;
        }
      }
    }
    o[k] = (k < G - 1) ? reduce(addm(lo, mulmod(reduce(hi, mod), eta, mod), mod), mod) : reduce(lo, mod);
  }
#pragma unroll
  for (int k = 0; k < G; k++) x[k] = o[k];
}

// eta of leaf g (0-based among the 2^nst leaves of the sub-transform rooted at `root` after nst stages): see above
template <class T, class M>
__device__ __forceinline__ T inc_eta(const T *__restrict__ tw, int root, int nst, int g, const M mod) {
  const T w = tw[((root << nst) + g) >> 1];
  return (g & 1) ? negm(w, mod) : w;
}

// The first `nst` stages of a forward transform of the LDS tile (lds_ntt_fwd of ntt_core.hpp stopped early) and the inverse
// stages u0 .. logn-1 (lds_ntt_inv started late): all-barrier rounds, any arithmetic.
template <int MAXR, class T, class M>
__device__ __forceinline__ void lds_ntt_fwd_part(T *s, int logn, int nst, const T *__restrict__ tw, int root, const M mod, uint32_t red_mask) {
  const LdsIOT<T> lds{s};
  int st = 0;
  while (st < nst) {
    const int R = pick_radix(nst - st, MAXR);
    fwd_round_dispatch<MAXR>(R, lds, lds, logn, logn, st, tw, root, mod, red_mask);
    __syncthreads();
    st += R;
  }
}
template <int MAXR, class T, class M>
__device__ __forceinline__ void lds_ntt_inv_part(T *s, int logn, int u0, const T *__restrict__ itw, int root, const M mod, uint32_t red_mask) {
  const LdsIOT<T> lds{s};
  int st = u0;
  while (st < logn) {
    const int R = pick_radix(logn - st, MAXR);
    inv_round_dispatch<MAXR>(R, lds, lds, logn, logn, st, itw, root, mod, red_mask);
    __syncthreads();
    st += R;
  }
}

// The pointwise step of the generic sub-transform kernel (any arithmetic, any inc; not the tuned path): a thread computes
// the output words tid, tid + nthr, ... (at most 16 of them) of the tile from the leaves they belong to, with run-time loops,
// keeps them in registers until every thread has read its operands, then writes them back.
// tab: the Bn table words of this block (table constants; DD: another workspace's spectrum, lazily reduced).  Ends with a barrier.
template <bool DD, class T, class M>
__device__ __forceinline__ void inc_pointwise_tile(T *s, int logB, int inc, const T *__restrict__ tab, const T *__restrict__ tw, int root,
                                                   const M mod) {
  const int G = 1 << inc, Bn = 1 << logB, nst = logB - inc;
  T o[16];
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int idx = (int)threadIdx.x + r * (int)blockDim.x;
    if (idx >= Bn) break;
    const int g = idx >> inc, k = idx & (G - 1);
    T lo = T(0), hi = T(0);
    int cl = 0, ch = 0;
    for (int i = 0; i < G; i++) {
      const int j = (k - i) & (G - 1);
      const T xi = reduce(s[pidx(g * G + i)], mod);
      const T pr = DD ? mulmod_dd(xi, reduce(tab[g * G + j], mod), mod) : mulmod(xi, tab[g * G + j], mod);
      if (i <= k) {
        lo = addm(lo, pr, mod);
        if (++cl == 4) {
          lo = reduce(lo, mod);
          cl = 1;
        }
      } else {
        hi = addm(hi, pr, mod);
        if (++ch == 4) {
          hi = reduce(hi, mod);
          ch = 1;
        }
      }
    }
    o[r] = reduce(addm(lo, mulmod(reduce(hi, mod), inc_eta(tw, root, nst, g, mod), mod), mod), mod);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int idx = (int)threadIdx.x + r * (int)blockDim.x;
    if (idx >= Bn) break;
    s[pidx(idx)] = o[r];
  }
  __syncthreads();
}

}  // namespace rs
