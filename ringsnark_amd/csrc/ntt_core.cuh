// ntt_core.cuh -- workgroup-cooperative radix-2^R number-theoretic transform over an LDS tile.
//
// One workgroup owns one polynomial of n = 2^logn FP64-held residues in LDS (f64mod.hpp).  Stages
// are processed in "rounds" of R <= 3 stages: a thread pulls the 2^R elements of one radix-2^R
// butterfly into registers, runs R stages there and writes them back, so a length-8192 transform
// costs 5 LDS round trips instead of 13.  LDS indices are padded by one slot per 16 (pidx) so
// that every stride pattern of every round is bank-conflict free for ds_read/write_b64
// (MI355X: 64 banks x 4 B, two 32-lane groups per b64 access).
//
// Twiddles: table entry tw[M*root + i] serves group i of the stage that has M groups, inside the
// sub-transform rooted at decimation-tree node `root` (root = 1 for a whole transform).  The
// same code runs the negacyclic transforms of the encoding contexts (SEAL order) and the cyclic
// transforms of the witness map; only the tables differ.
#pragma once
#include <hip/hip_runtime.h>

#include "f64mod.hpp"

namespace rs {

constexpr int PAD_SHIFT = 4;
__host__ __device__ __forceinline__ int pidx(int i) { return i + (i >> PAD_SHIFT); }
__host__ __device__ inline size_t padded_len(size_t n) { return n + (n >> PAD_SHIFT); }

// ---- forward (Cooley-Tukey, natural in -> bit-reversed out) -----------------------------------
// stage s (0-based) has 2^s groups and gap n >> (s+1); butterfly (x, y) -> (x + w*y, x - w*y).
template <int R>
__device__ __forceinline__ void fwd_round(double *__restrict__ s, int logn, int s0,
                                          const double *__restrict__ tw, int root, const Mod mod,
                                          uint32_t red_mask) {
  constexpr int E = 1 << R;
  const int n = 1 << logn;
  const int lstep = logn - s0 - R;  // log2 of the smallest gap in this round
  const int sstep = 1 << lstep;
  const int ngroups = n >> R;
  for (int grp = threadIdx.x; grp < ngroups; grp += blockDim.x) {
    const int lo = grp & (sstep - 1), hi = grp >> lstep;
    const int base = (hi << (logn - s0)) + lo;
    double v[E];
#pragma unroll
    for (int e = 0; e < E; e++) v[e] = s[pidx(base + e * sstep)];
#pragma unroll
    for (int k = 0; k < R; k++) {
      if ((red_mask >> (s0 + k)) & 1u) {
#pragma unroll
        for (int e = 0; e < E; e++) v[e] = reduce(v[e], mod);
      }
      const int half = E >> (k + 1);
      const int twbase = ((1 << (s0 + k)) * root) + (hi << k);
#pragma unroll
      for (int blk = 0; blk < (1 << k); blk++) {
        const double w = tw[twbase + blk];
#pragma unroll
        for (int e0 = 0; e0 < half; e0++) {
          const int ia = blk * 2 * half + e0, ib = ia + half;
          const double t = mulmod(v[ib], w, mod);
          const double a = v[ia];
          v[ia] = a + t;
          v[ib] = a - t;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < E; e++) s[pidx(base + e * sstep)] = v[e];
  }
}

__device__ __forceinline__ int pick_radix(int rem) { return (rem >= 3 && rem != 4) ? 3 : (rem >= 2 ? 2 : 1); }

// Whole forward transform of the LDS tile; ends with a barrier.  Caller must have synchronised
// after filling the tile.
__device__ __forceinline__ void lds_ntt_fwd(double *s, int logn, const double *__restrict__ tw, int root,
                                            const Mod mod, uint32_t red_mask) {
  int st = 0;
  while (st < logn) {
    const int R = pick_radix(logn - st);
    if (R == 3)
      fwd_round<3>(s, logn, st, tw, root, mod, red_mask);
    else if (R == 2)
      fwd_round<2>(s, logn, st, tw, root, mod, red_mask);
    else
      fwd_round<1>(s, logn, st, tw, root, mod, red_mask);
    __syncthreads();
    st += R;
  }
}

// ---- inverse (Gentleman-Sande, bit-reversed in -> natural out, NOT scaled by n^-1) -----------
// inverse stage u (0-based) has gap 2^u and n >> (u+1) groups; (a, b) -> (a + b, (a - b)*w).
template <int R>
__device__ __forceinline__ void inv_round(double *__restrict__ s, int logn, int u0,
                                          const double *__restrict__ itw, int root, const Mod mod,
                                          uint32_t red_mask) {
  constexpr int E = 1 << R;
  const int n = 1 << logn;
  const int g0 = 1 << u0;
  const int ngroups = n >> R;
  for (int grp = threadIdx.x; grp < ngroups; grp += blockDim.x) {
    const int lo = grp & (g0 - 1), hi = grp >> u0;
    const int base = (hi << (u0 + R)) + lo;
    double v[E];
#pragma unroll
    for (int e = 0; e < E; e++) v[e] = s[pidx(base + e * g0)];
#pragma unroll
    for (int k = 0; k < R; k++) {
      if ((red_mask >> (u0 + k)) & 1u) {
#pragma unroll
        for (int e = 0; e < E; e++) v[e] = reduce(v[e], mod);
      }
      const int M = n >> (u0 + k + 1);
      const int twbase = M * root + (hi << (R - 1 - k));
#pragma unroll
      for (int e = 0; e < E; e++) {
        if (e & (1 << k)) continue;
        const double w = itw[twbase + (e >> (k + 1))];
        const double a = v[e], b = v[e + (1 << k)];
        v[e] = a + b;
        v[e + (1 << k)] = mulmod(a - b, w, mod);
      }
    }
#pragma unroll
    for (int e = 0; e < E; e++) s[pidx(base + e * g0)] = v[e];
  }
}

__device__ __forceinline__ void lds_ntt_inv(double *s, int logn, const double *__restrict__ itw, int root,
                                            const Mod mod, uint32_t red_mask) {
  int st = 0;
  while (st < logn) {
    const int R = pick_radix(logn - st);
    if (R == 3)
      inv_round<3>(s, logn, st, itw, root, mod, red_mask);
    else if (R == 2)
      inv_round<2>(s, logn, st, itw, root, mod, red_mask);
    else
      inv_round<1>(s, logn, st, itw, root, mod, red_mask);
    __syncthreads();
    st += R;
  }
}

}  // namespace rs
