// witness_bc.hpp -- block convolutions (pairwise and two-dimensional) for ring primes without a 2M-th root of unity: device kernels (witness.hip)
#pragma once
#include "witness_cols.hpp"
#include "witness_multipass.hpp"

namespace rs {

// =============================================================================================
// Block convolutions: the witness map for ring primes WITHOUT a 2M-th root of unity.
// The reference's recipe (seal/seal_util.hpp:20-32) only makes q_i = 1 mod 2*N_inner, and its O(m^2) algorithm
// works for any prime; the transforms above need q_i = 1 mod 2M (2^17 at the headline).  When a ring prime falls
// short, every product longer than the largest supported transform (2^bcLog = 2B) is computed blockwise:
//     X = sum_i X_i x^(iB),  Y = sum_j Y_j x^(jB)   (blocks of B coefficients)
//     X*Y = sum_k x^(kB) * ( sum_{i+j=k} X_i*Y_j ),   each X_i*Y_j (< 2B coefficients) by one cyclic transform of length 2B
// i.e. forward transforms of the blocks (bc_fwd_kernel), per output block pair k the sum of pointwise products and ONE
// inverse transform (bc_mac_kernel), and an overlap-add with the step's sink (bc_out_kernel).  Exact, hence
// bit-identical; (n/B)^2 pointwise products instead of n log n butterflies for the part above 2B.
// =============================================================================================
enum BcSrc { BS_SCALE = 0, BS_CENTER, BS_REVTRUNC, BS_RIGHT };
enum BcY { BY_E = 0, BY_S, BY_D, BY_DATA };
enum BcDst { BD_NEWTON = 0, BD_PLAIN_SCALED, BD_HFIN, BD_COMBINE, BD_COMBINE_CANON };
struct BcArgs {
  const void *src;   // source columns
  void *Xhat;        // [ncols * units][nxb][2B] spectra of the source blocks
  const void *Yhat;  // BY_DATA: [ncols][nyb][2B] spectra of the other operand
  void *Wc;          // [ncols * units][nk][2B] block-pair products
  void *dst;
  int bcLog, logM, m, l;  // l: tree level (node size 2^l) for BS_RIGHT / BY_D / BD_COMBINE
  int nxb, nyb, nk, units;
  size_t col0;
  unsigned S, slots_per_limb;
};

template <int SRC, class CPS>
__global__ void __launch_bounds__(1024) bc_fwd_kernel(BcArgs a, CPS plans) {
  using T = typename CPS::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *s = reinterpret_cast<T *>(smem);
  const int B = 1 << (a.bcLog - 1);
  const size_t bid = blockIdx.x;
  const int blk = (int)(bid % a.nxb), unit = (int)((bid / a.nxb) % a.units);
  const size_t col = bid / ((size_t)a.nxb * a.units), M = (size_t)1 << a.logM;
  const ColPlanT<typename CPS::M> &P = plans.l[((a.col0 + col) % a.S) / a.slots_per_limb];
  const typename CPS::M mod = P.mod;
  const T *src = static_cast<const T *>(a.src);
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    const size_t k = (size_t)blk * B + i;
    T v = T(0);
    if (SRC == BS_SCALE) {
      if (k < M) v = mulmod(src[col * M + k], P.invfact[k], mod);
    } else if (SRC == BS_CENTER) {
      if (k < M) v = center(src[col * M + k], mod);
    } else if (SRC == BS_REVTRUNC) {  // T_k = P_{2m-2-k}, k < m-1, from a [ncols][2M] buffer
      if ((long long)k < (long long)a.m - 1) v = reduce(src[col * 2 * M + (size_t)(2 * a.m - 2) - k], mod);
    } else {  // BS_RIGHT: F_right of node `unit` at level l
      const size_t n = (size_t)1 << a.l, h = n >> 1;
      if (k < h) v = src[col * M + (size_t)unit * n + h + k];
    }
    s[pidx(i)] = v;
    s[pidx(B + i)] = T(0);
  }
  __syncthreads();
  lds_ntt_fwd<3>(s, a.bcLog, P.tw, 1, mod, P.fmask[a.bcLog]);
  T *out = static_cast<T *>(a.Xhat) + bid * (size_t)(2 * B);
  for (int i = threadIdx.x; i < 2 * B; i += blockDim.x) out[i] = reduce(s[pidx(i)], mod);
}

template <int YK, class CPS>
__global__ void __launch_bounds__(1024) bc_mac_kernel(BcArgs a, CPS plans) {
  using T = typename CPS::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *s = reinterpret_cast<T *>(smem);
  const int B2 = 1 << a.bcLog;
  const size_t bid = blockIdx.x;
  const int k = (int)(bid % a.nk), unit = (int)((bid / a.nk) % a.units);
  const size_t col = bid / ((size_t)a.nk * a.units);
  const ColPlanT<typename CPS::M> &P = plans.l[((a.col0 + col) % a.S) / a.slots_per_limb];
  const typename CPS::M mod = P.mod;
  const T *X = static_cast<const T *>(a.Xhat) + (col * a.units + unit) * (size_t)a.nxb * B2;
  const T *Y;
  if (YK == BY_E)
    Y = P.bc_e;
  else if (YK == BY_S)
    Y = P.bc_s;
  else if (YK == BY_D)
    Y = P.bc_d + (size_t)(a.l - a.bcLog - 1) * ((size_t)1 << a.logM) + (size_t)unit * ((size_t)1 << a.l);
  else
    Y = static_cast<const T *>(a.Yhat) + col * (size_t)a.nyb * B2;
  const int i0 = k >= a.nyb ? k - a.nyb + 1 : 0, i1 = k < a.nxb ? k : a.nxb - 1;
  for (int e = threadIdx.x; e < B2; e += blockDim.x) {
    T acc = T(0);
    int since = 0;
    for (int ib = i0; ib <= i1; ib++) {
      const T x = X[(size_t)ib * B2 + e], y = Y[(size_t)(k - ib) * B2 + e];
      acc = addm(acc, YK == BY_DATA ? mulmod_dd(x, y, mod) : mulmod(x, y, mod), mod);
      if (++since == 4) {
        since = 0;
        acc = reduce(acc, mod);
      }
    }
    s[pidx(e)] = reduce(acc, mod);
  }
  __syncthreads();
  lds_ntt_inv<3>(s, a.bcLog, P.itw, 1, mod, P.imask[a.bcLog]);
  T *out = static_cast<T *>(a.Wc) + bid * (size_t)B2;
  for (int e = threadIdx.x; e < B2; e += blockDim.x) out[e] = reduce(s[pidx(e)], mod);
}

// overlap-add of the block-pair products + the step's sink; one thread per output coefficient
template <int DST, class CPS>
__global__ void __launch_bounds__(256) bc_out_kernel(BcArgs a, CPS plans, size_t ncols, size_t per_unit) {
  using T = typename CPS::T;
  const int B = 1 << (a.bcLog - 1);
  const size_t M = (size_t)1 << a.logM, total = ncols * a.units * per_unit, stride = (size_t)gridDim.x * blockDim.x;
  const T *W = static_cast<const T *>(a.Wc);
  T *dst = static_cast<T *>(a.dst);
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const size_t t = idx % per_unit, cu = idx / per_unit, unit = cu % a.units, col = cu / a.units;
    const ColPlanT<typename CPS::M> &P = plans.l[((a.col0 + col) % a.S) / a.slots_per_limb];
    const typename CPS::M mod = P.mod;
    const size_t kb = t / B, r = t % B;
    T v = T(0);
    if (kb < (size_t)a.nk) v = W[(cu * a.nk + kb) * (size_t)(2 * B) + r];
    if (kb >= 1 && kb - 1 < (size_t)a.nk) v = addm(v, W[(cu * a.nk + kb - 1) * (size_t)(2 * B) + B + r], mod);
    if (DST == BD_NEWTON) {  // Newton coefficients: the low M terms, zero beyond m
      dst[col * M + t] = (P.invfact[t] != T(0)) ? reduce(v, mod) : T(0);
    } else if (DST == BD_PLAIN_SCALED) {  // data x data product: the 1/(2B) of the inverse transform is applied here
      dst[col * 2 * M + t] = mulmod(reduce(v, mod), P.bc_inv2b, mod);
    } else if (DST == BD_HFIN) {  // H_j = U_{m-2-j}
      if ((long long)t <= (long long)a.m - 2) dst[col * M + (size_t)(a.m - 2) - t] = reduce(v, mod);
    } else {  // F_node = (F_left, 0) + x^h F_right + d * F_right: both extra terms sit at this very position
      const size_t pos = col * M + unit * ((size_t)1 << a.l) + t;
      const T f = reduce(addm(v, dst[pos], mod), mod);
      dst[pos] = DST == BD_COMBINE_CANON ? canon(f, mod) : f;
    }
  }
}

// =============================================================================================
// Two-dimensional block convolutions (WitnessPlan::bc2; FP64, ring primes with 2-adicity >= 14, M >= 2^15).
// A polynomial of n B-coefficient blocks, B = 2^13, is the bivariate  F(x, y) = sum_i f_i(x) y^i  at y = x^B.  The product
// of two such polynomials has degree < 2B in x and < 2n in y, so it IS the two-dimensional cyclic convolution of size
// 2B x Y, Y = 2n, of the zero-padded operands -- and a two-dimensional transform only needs a 2B-th and a Y-th root of
// unity (there are no twiddles between the dimensions, unlike the one-dimensional transform of length 2B*Y that the
// primes of the reference's recipe do not support).  Per convolution:
//     bc2_yfwd_kernel   the step's source functor, then the Y-point transform ACROSS the blocks (half of them zero), per
//                       coefficient position: [Y][B] words out
//     sub_ntt_wide_kernel  per block: the 2B-point transform of the zero-padded block = the two B-point sub-transforms
//                       rooted at nodes 2 and 3 of the SAME input (Xsrc), the product with the two-dimensional spectrum
//                       of the other operand, the inverse sub-transforms -- the tuned kernel of the multi-pass path
//     bc2_yinv_kernel   the last inverse stage of the 2B-point transforms (u +- v), the inverse Y-point transform across
//                       blocks, the overlap-add (coefficient k B + r = low half of block k + high half of block k-1: both in
//                       this thread's registers) and the step's sink functor
// against the pairwise form above: (blocks)^2 block products re-read from memory become Y log Y butterflies in registers.
// Exact, hence bit-identical.
// =============================================================================================
struct Bc2Args {
  const double *src;  // source columns
  double *Wy;         // [ncols * units][Y][B]
  double *Ws;         // [ncols * units][Y][2][B]
  double *dst;
  int logM, m, l, units;
  size_t col0;
  unsigned S, slots_per_limb;
};
constexpr int BC2_LOGB = 13, BC2_B = 1 << BC2_LOGB;

template <int SRC, int LOGY>
__global__ void __launch_bounds__(256) bc2_yfwd_kernel(Bc2Args a, ColPlans plans) {
  constexpr int Y = 1 << LOGY, NX = Y / 2, B = BC2_B;
  const int r = 2 * (int)(blockIdx.x * 256 + threadIdx.x);
  const size_t cu = blockIdx.y, unit = cu % (size_t)a.units, col = cu / (size_t)a.units, M = (size_t)1 << a.logM;
  const ColPlan &P = plans.l[((a.col0 + col) % a.S) / a.slots_per_limb];
  const Mod mod = P.mod;
  double v0[Y], v1[Y];
#pragma unroll
  for (int i = 0; i < NX; i++) {
    const size_t k = (size_t)i * B + r;  // position inside the operand (pairs k, k + 1 never straddle a limit: all are even)
    double x0 = 0.0, x1 = 0.0;
    if (SRC == BS_SCALE) {
      if (k < M) {
        const double2 d = *reinterpret_cast<const double2 *>(a.src + col * M + k), f = *reinterpret_cast<const double2 *>(P.invfact + k);
        x0 = mulmod(d.x, f.x, mod);
        x1 = mulmod(d.y, f.y, mod);
      }
    } else if (SRC == BS_CENTER) {
      if (k < M) {
        const double2 d = *reinterpret_cast<const double2 *>(a.src + col * M + k);
        x0 = center(d.x, mod);
        x1 = center(d.y, mod);
      }
    } else if (SRC == BS_REVTRUNC) {  // T_k = P_{2m-2-k}, k < m-1, from a [ncols][2M] buffer
      const long long lim = (long long)a.m - 1;
      if ((long long)k < lim) x0 = reduce(a.src[col * 2 * M + (size_t)(2 * a.m - 2) - k], mod);
      if ((long long)k + 1 < lim) x1 = reduce(a.src[col * 2 * M + (size_t)(2 * a.m - 2) - k - 1], mod);
    } else {  // BS_RIGHT: F_right of node `unit` at level l
      const size_t n = (size_t)1 << a.l, h = n >> 1;
      const double2 d = *reinterpret_cast<const double2 *>(a.src + col * M + unit * n + h + k);
      x0 = d.x;
      x1 = d.y;
    }
    v0[i] = x0;
    v1[i] = x1;
  }
  const double *__restrict__ tw = P.tw;
  reg_fwd_stages_zu<LOGY>(v0, mod, P.fmask[LOGY], [&](int st, int blk) { return tw[(1 << st) + blk]; });
  reg_fwd_stages_zu<LOGY>(v1, mod, P.fmask[LOGY], [&](int st, int blk) { return tw[(1 << st) + blk]; });
  double *out = a.Wy + cu * (size_t)Y * B + r;
#pragma unroll
  for (int y = 0; y < Y; y++) *reinterpret_cast<double2 *>(out + (size_t)y * B) = make_double2(reduce(v0[y], mod), reduce(v1[y], mod));
}

template <int DST, int LOGY>
__global__ void __launch_bounds__(256) bc2_yinv_kernel(Bc2Args a, ColPlans plans) {
  constexpr int Y = 1 << LOGY, B = BC2_B;
  const int r = 2 * (int)(blockIdx.x * 256 + threadIdx.x);
  const size_t cu = blockIdx.y, unit = cu % (size_t)a.units, col = cu / (size_t)a.units, M = (size_t)1 << a.logM;
  const ColPlan &P = plans.l[((a.col0 + col) % a.S) / a.slots_per_limb];
  const Mod mod = P.mod;
  const double *__restrict__ itw = P.itw;
  // lo / hi: coefficients r (+1) and B + r (+1) of the 2B-point blocks; [c]: the two adjacent positions of this thread
  double lo[2][Y], hi[2][Y];
  const double *in = a.Ws + cu * (size_t)Y * 2 * B + r;
#pragma unroll
  for (int y = 0; y < Y; y++) {
    const double2 u = *reinterpret_cast<const double2 *>(in + (size_t)(2 * y) * B), w = *reinterpret_cast<const double2 *>(in + (size_t)(2 * y + 1) * B);
    lo[0][y] = reduce(u.x + w.x, mod);  // last inverse stage of the 2B-point transform: its twiddle is 1
    hi[0][y] = reduce(u.x - w.x, mod);
    lo[1][y] = reduce(u.y + w.y, mod);
    hi[1][y] = reduce(u.y - w.y, mod);
  }
#pragma unroll
  for (int c = 0; c < 2; c++) {
    reg_inv_stages<LOGY, true>(lo[c], mod, P.imask[LOGY], [&](int k, int i) { return itw[(Y >> (k + 1)) + i]; });
    reg_inv_stages<LOGY, true>(hi[c], mod, P.imask[LOGY], [&](int k, int i) { return itw[(Y >> (k + 1)) + i]; });
  }
#pragma unroll
  for (int k = 0; k < Y; k++) {
    const size_t t = (size_t)k * B + r;  // output coefficient (and t + 1)
    double o0 = lo[0][k], o1 = lo[1][k];
    if (k >= 1) {
      o0 += hi[0][k - 1];
      o1 += hi[1][k - 1];
    }
    if (DST == BD_NEWTON) {  // Newton coefficients: the low M terms, zero beyond m
      if (t < M) {
        const double2 f = *reinterpret_cast<const double2 *>(P.invfact + t);
        *reinterpret_cast<double2 *>(a.dst + col * M + t) = make_double2(f.x != 0.0 ? reduce(o0, mod) : 0.0, f.y != 0.0 ? reduce(o1, mod) : 0.0);
      }
    } else if (DST == BD_PLAIN_SCALED) {  // data x data product: the scale of both inverse transforms is applied here
      *reinterpret_cast<double2 *>(a.dst + col * 2 * M + t) = make_double2(mulmod(reduce(o0, mod), P.b2_inv, mod), mulmod(reduce(o1, mod), P.b2_inv, mod));
    } else if (DST == BD_HFIN) {  // H_j = U_{m-2-j}
      const long long top = (long long)a.m - 2;
      if ((long long)t <= top) a.dst[col * M + (size_t)(top - (long long)t)] = reduce(o0, mod);
      if ((long long)t + 1 <= top) a.dst[col * M + (size_t)(top - (long long)t - 1)] = reduce(o1, mod);
    } else {  // F_node = (F_left, 0) + x^h F_right + d * F_right: both extra terms sit at this very position
      double2 *p = reinterpret_cast<double2 *>(a.dst + col * M + unit * ((size_t)1 << a.l) + t);
      const double2 d = *p;
      const double f0 = reduce(o0 + d.x, mod), f1 = reduce(o1 + d.y, mod);
      *p = DST == BD_COMBINE_CANON ? make_double2(canon(f0, mod), canon(f1, mod)) : make_double2(f0, f1);
    }
  }
}

// ---- the two turns of the multi-pass path (witness_multipass.hpp: cross_turn_kernel, cross_level_turn_kernel) on the
// two-dimensional convolutions: a thread of the across-block passes owns a POSITION r inside the blocks (all Y blocks of
// it), and where the sink of one convolution is the source of the next at the same positions, the two HBM-bound passes
// are one.  Same arithmetic as bc2_yinv_kernel followed by bc2_yfwd_kernel; one-level transforms only (Y <= 32).
//
// Tree levels l -> l + 1: both children of a parent node at position r -- the last stage of the 2B-point transforms, the
// inverse transform across blocks, the overlap-add, the recombination (BD_COMBINE) -- both back to the columns, and the
// right one (the parent's whole input, BS_RIGHT) through the parent's forward transform across blocks into Wy without
// being read again.
template <int LOGYC>
__global__ void __launch_bounds__(256) bc2_level_turn_kernel(Bc2Args a, ColPlans plans) {
  constexpr int YC = 1 << LOGYC, YP = 2 * YC, B = BC2_B;
  const int r = 2 * (int)(blockIdx.x * 256 + threadIdx.x);
  const size_t parents = (size_t)a.units / 2, cpn = blockIdx.y, parent = cpn % parents, col = cpn / parents, M = (size_t)1 << a.logM;
  const ColPlan &P = plans.l[((a.col0 + col) % a.S) / a.slots_per_limb];
  const Mod mod = P.mod;
  const double *__restrict__ itw = P.itw;
  const double *__restrict__ tw = P.tw;
  double v0[YP], v1[YP];
#pragma unroll
  for (int ch = 0; ch < 2; ch++) {
    const size_t unit = 2 * parent + ch, cu = col * (size_t)a.units + unit;
    double lo[2][YC], hi[2][YC];
    const double *in = a.Ws + cu * (size_t)YC * 2 * B + r;
#pragma unroll
    for (int y = 0; y < YC; y++) {
      const double2 u = *reinterpret_cast<const double2 *>(in + (size_t)(2 * y) * B), w = *reinterpret_cast<const double2 *>(in + (size_t)(2 * y + 1) * B);
      lo[0][y] = reduce(u.x + w.x, mod);
      hi[0][y] = reduce(u.x - w.x, mod);
      lo[1][y] = reduce(u.y + w.y, mod);
      hi[1][y] = reduce(u.y - w.y, mod);
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
      reg_inv_stages<LOGYC, true>(lo[c], mod, P.imask[LOGYC], [&](int k, int i) { return itw[(YC >> (k + 1)) + i]; });
      reg_inv_stages<LOGYC, true>(hi[c], mod, P.imask[LOGYC], [&](int k, int i) { return itw[(YC >> (k + 1)) + i]; });
    }
#pragma unroll
    for (int k = 0; k < YC; k++) {
      const size_t t = (size_t)k * B + r;
      double o0 = lo[0][k], o1 = lo[1][k];
      if (k >= 1) {
        o0 += hi[0][k - 1];
        o1 += hi[1][k - 1];
      }
      double2 *p = reinterpret_cast<double2 *>(a.dst + col * M + unit * ((size_t)1 << a.l) + t);
      const double2 d = *p;
      const double f0 = reduce(o0 + d.x, mod), f1 = reduce(o1 + d.y, mod);
      // both children go back to the columns: the parent's recombination reads F_left AND x^h F_right at their positions
      // (D_left = x^h + d: only d F_right is a convolution here, bc2_yinv_kernel BD_COMBINE)
      *p = make_double2(f0, f1);
      if (ch == 1) {  // the right child is also the parent's transform input, (F_right, 0): not read back
        v0[k] = f0;
        v1[k] = f1;
      }
    }
  }
  reg_fwd_stages_zu<LOGYC + 1>(v0, mod, P.fmask[LOGYC + 1], [&](int st, int blk) { return tw[(1 << st) + blk]; });
  reg_fwd_stages_zu<LOGYC + 1>(v1, mod, P.fmask[LOGYC + 1], [&](int st, int blk) { return tw[(1 << st) + blk]; });
  double *out = a.Wy + cpn * (size_t)YP * B + r;
#pragma unroll
  for (int y = 0; y < YP; y++) *reinterpret_cast<double2 *>(out + (size_t)y * B) = make_double2(reduce(v0[y], mod), reduce(v1[y], mod));
}

// The turn of H: P = A B (sink BD_PLAIN_SCALED) straight into T = rev(P) mod x^(m-1) (source BS_REVTRUNC).  T_k = P_{2m-2-k}:
// the forward positions r, r + 1 (r even) take their inputs from the product's positions rho = (2m - 2 - r) mod B and
// rho - 1 -- odd aligned: 8-byte loads -- block K0 - i for block i, K0 = (2m - 2 - r - c) div B: the register tile is
// reversed and shifted by Y - 1 - K0 (a barrel of LOGY select rounds) and truncated at m - 1.  The 2M-word product buffer
// is neither written nor read.
template <int LOGY>
__global__ void __launch_bounds__(256) bc2_h_turn_kernel(Bc2Args a, ColPlans plans) {
  constexpr int Y = 1 << LOGY, B = BC2_B;
  const int r = 2 * (int)(blockIdx.x * 256 + threadIdx.x);
  const size_t cu = blockIdx.y, col = cu;  // units = 1
  const ColPlan &P = plans.l[((a.col0 + col) % a.S) / a.slots_per_limb];
  const Mod mod = P.mod;
  const double *__restrict__ itw = P.itw;
  const double *__restrict__ tw = P.tw;
  const int q = 2 * a.m - 2;
  double x[2][Y];
#pragma unroll
  for (int c = 0; c < 2; c++) {
    const int i0 = q - r - c, rho = i0 & (B - 1), K0 = i0 >> BC2_LOGB;
    double lo[Y], hi[Y];
    const double *in = a.Ws + cu * (size_t)Y * 2 * B + rho;
#pragma unroll
    for (int y = 0; y < Y; y++) {
      const double u = in[(size_t)(2 * y) * B], w = in[(size_t)(2 * y + 1) * B];
      lo[y] = reduce(u + w, mod);
      hi[y] = reduce(u - w, mod);
    }
    reg_inv_stages<LOGY, true>(lo, mod, P.imask[LOGY], [&](int k, int i) { return itw[(Y >> (k + 1)) + i]; });
    reg_inv_stages<LOGY, true>(hi, mod, P.imask[LOGY], [&](int k, int i) { return itw[(Y >> (k + 1)) + i]; });
    double u_[Y];  // u_[e] = P at block Y - 1 - e of class rho
#pragma unroll
    for (int k = 0; k < Y; k++) {
      double o = lo[k];
      if (k >= 1) o += hi[k - 1];
      u_[Y - 1 - k] = mulmod(reduce(o, mod), P.b2_inv, mod);
    }
    const int sh = Y - 1 - K0;
#pragma unroll
    for (int b = 0; b < LOGY; b++) {
      const bool on = (sh >> b) & 1;
#pragma unroll
      for (int e = 0; e < Y; e++) {
        const double far = (e + (1 << b) < Y) ? u_[e + (1 << b)] : 0.0;
        u_[e] = on ? far : u_[e];
      }
    }
#pragma unroll
    for (int i = 0; i < Y / 2; i++) x[c][i] = ((long long)i * B + r + c < (long long)a.m - 1) ? reduce(u_[i], mod) : 0.0;
  }
  reg_fwd_stages_zu<LOGY>(x[0], mod, P.fmask[LOGY], [&](int st, int blk) { return tw[(1 << st) + blk]; });
  reg_fwd_stages_zu<LOGY>(x[1], mod, P.fmask[LOGY], [&](int st, int blk) { return tw[(1 << st) + blk]; });
  double *out = a.Wy + cu * (size_t)Y * B + r;
#pragma unroll
  for (int y = 0; y < Y; y++) *reinterpret_cast<double2 *>(out + (size_t)y * B) = make_double2(reduce(x[0][y], mod), reduce(x[1][y], mod));
}

// ---- Y > 32 blocks (M >= 2^18 at B = 2^13): the transform across blocks in two levels -----------------------------
// A thread cannot hold Y = 64 .. 256 values of two positions.  Y = R x 32, R = 2^S1:
//   forward   the first S1 stages mix elements 32 apart, the last five are R independent 32-point sub-transforms rooted
//             at nodes R + part.  A workgroup owns ONE part: element j of it depends on the inputs j + 32 e only (the
//             upper half of them padding), which are read through the step's source functor once per part (R/2 loads
//             per element; the parts of a position are neighbouring workgroups, so the re-reads hit the caches)
//   inverse   bc2_yinv_a_kernel: per group of 32 consecutive blocks the last stage of the 2B-point transforms and the
//             first five stages across blocks, in place on Ws; bc2_yinv_b_kernel: per element j the last S1 stages on
//             j + 32 g, the overlap-add (which needs the HIGH halves of element j - 1: carried by the same thread) and
//             the step's sink functor.
// Same arithmetic as the one-level kernels up to the points of lazy reduction; outputs are canonical, hence identical.
template <int SRC, int S1>
__global__ void __launch_bounds__(256) bc2_yfwd_big_kernel(Bc2Args a, ColPlans plans) {
  constexpr int R = 1 << S1, Y = 32 * R, B = BC2_B;
  const int part = (int)(blockIdx.x % R);
  const int r = 2 * (int)((blockIdx.x / R) * 256 + threadIdx.x);
  const size_t cu = blockIdx.y, unit = cu % (size_t)a.units, col = cu / (size_t)a.units, M = (size_t)1 << a.logM;
  const ColPlan &P = plans.l[((a.col0 + col) % a.S) / a.slots_per_limb];
  const Mod mod = P.mod;
  const double *__restrict__ tw = P.tw;
  const uint32_t mask = P.fmask[S1 + 5];
  double v0[32], v1[32];
#pragma unroll
  for (int j = 0; j < 32; j++) {
    double x0[R], x1[R];
#pragma unroll
    for (int e = 0; e < R / 2; e++) {
      const size_t k = (size_t)(j + 32 * e) * B + r;  // position inside the operand (pairs k, k + 1 never straddle a limit)
      double y0 = 0.0, y1 = 0.0;
      if (SRC == BS_SCALE) {
        if (k < M) {
          const double2 d = *reinterpret_cast<const double2 *>(a.src + col * M + k), f = *reinterpret_cast<const double2 *>(P.invfact + k);
          y0 = mulmod(d.x, f.x, mod);
          y1 = mulmod(d.y, f.y, mod);
        }
      } else if (SRC == BS_CENTER) {
        if (k < M) {
          const double2 d = *reinterpret_cast<const double2 *>(a.src + col * M + k);
          y0 = center(d.x, mod);
          y1 = center(d.y, mod);
        }
      } else if (SRC == BS_REVTRUNC) {
        const long long lim = (long long)a.m - 1;
        if ((long long)k < lim) y0 = reduce(a.src[col * 2 * M + (size_t)(2 * a.m - 2) - k], mod);
        if ((long long)k + 1 < lim) y1 = reduce(a.src[col * 2 * M + (size_t)(2 * a.m - 2) - k - 1], mod);
      } else {  // BS_RIGHT
        const size_t n = (size_t)1 << a.l, h = n >> 1;
        const double2 d = *reinterpret_cast<const double2 *>(a.src + col * M + unit * n + h + k);
        y0 = d.x;
        y1 = d.y;
      }
      x0[e] = y0;
      x1[e] = y1;
    }
    // stages 0 .. S1-1 on the elements j + 32 e (stage 0 is a copy: the upper half is padding); keep element `part`
    reg_fwd_stages_zu<S1>(x0, mod, mask, [&](int st, int blk) { return tw[(1 << st) + blk]; });
    reg_fwd_stages_zu<S1>(x1, mod, mask, [&](int st, int blk) { return tw[(1 << st) + blk]; });
    double s0 = x0[0], s1 = x1[0];
#pragma unroll
    for (int e = 1; e < R; e++) {
      s0 = part == e ? x0[e] : s0;
      s1 = part == e ? x1[e] : s1;
    }
    v0[j] = s0;
    v1[j] = s1;
    // bound the loads in flight (fully hoisted, the R/2 x 32 source loads of a part would not fit the register file)
    if (R >= 4 && (j % (16 / R)) == 16 / R - 1) mem_fence();
  }
  // stages S1 .. S1+4: the 32-point sub-transform rooted at node R + part
  reg_fwd_stages<5, true>(v0, mod, mask >> S1, [&](int k, int blk) { return tw[(R << k) + (part << k) + blk]; });
  reg_fwd_stages<5, true>(v1, mod, mask >> S1, [&](int k, int blk) { return tw[(R << k) + (part << k) + blk]; });
  double *out = a.Wy + cu * (size_t)Y * B + (size_t)(32 * part) * B + r;
#pragma unroll
  for (int y = 0; y < 32; y++) *reinterpret_cast<double2 *>(out + (size_t)y * B) = make_double2(reduce(v0[y], mod), reduce(v1[y], mod));
}

template <int S1>
__global__ void __launch_bounds__(256) bc2_yinv_a_kernel(Bc2Args a, ColPlans plans) {
  constexpr int R = 1 << S1, Y = 32 * R, B = BC2_B;
  const int g = (int)(blockIdx.x % R);
  const int r = 2 * (int)((blockIdx.x / R) * 256 + threadIdx.x);
  const size_t cu = blockIdx.y, col = cu / (size_t)a.units;
  const ColPlan &P = plans.l[((a.col0 + col) % a.S) / a.slots_per_limb];
  const Mod mod = P.mod;
  const double *__restrict__ itw = P.itw;
  double lo[2][32], hi[2][32];
  double *ws = a.Ws + cu * (size_t)Y * 2 * B + (size_t)(64 * g) * B + r;
#pragma unroll
  for (int y = 0; y < 32; y++) {
    const double2 u = *reinterpret_cast<const double2 *>(ws + (size_t)(2 * y) * B), w = *reinterpret_cast<const double2 *>(ws + (size_t)(2 * y + 1) * B);
    lo[0][y] = reduce(u.x + w.x, mod);  // last inverse stage of the 2B-point transform: its twiddle is 1
    hi[0][y] = reduce(u.x - w.x, mod);
    lo[1][y] = reduce(u.y + w.y, mod);
    hi[1][y] = reduce(u.y - w.y, mod);
  }
  // inverse stages 0..4 across blocks; block of stage k: (32 g + e) >> (k + 1)
#pragma unroll
  for (int c = 0; c < 2; c++) {
    reg_inv_stages<5, true>(lo[c], mod, P.imask[S1 + 5], [&](int k, int i) { return itw[(Y >> (k + 1)) + (g << (4 - k)) + i]; });
    reg_inv_stages<5, true>(hi[c], mod, P.imask[S1 + 5], [&](int k, int i) { return itw[(Y >> (k + 1)) + (g << (4 - k)) + i]; });
  }
#pragma unroll
  for (int y = 0; y < 32; y++) {
    *reinterpret_cast<double2 *>(ws + (size_t)(2 * y) * B) = make_double2(lo[0][y], lo[1][y]);
    *reinterpret_cast<double2 *>(ws + (size_t)(2 * y + 1) * B) = make_double2(hi[0][y], hi[1][y]);
  }
}

template <int DST, int S1>
__global__ void __launch_bounds__(256) bc2_yinv_b_kernel(Bc2Args a, ColPlans plans) {
  constexpr int R = 1 << S1, Y = 32 * R, B = BC2_B;
  const int j = (int)(blockIdx.x % 32), jp = (j + 31) % 32;
  const int r = 2 * (int)((blockIdx.x / 32) * 256 + threadIdx.x);
  const size_t cu = blockIdx.y, unit = cu % (size_t)a.units, col = cu / (size_t)a.units, M = (size_t)1 << a.logM;
  const ColPlan &P = plans.l[((a.col0 + col) % a.S) / a.slots_per_limb];
  const Mod mod = P.mod;
  const double *__restrict__ itw = P.itw;
  // lo: low halves of blocks j + 32 g; hi: high halves of blocks jp + 32 g (block k - 1 of the overlap-add)
  double lo[2][R], hi[2][R];
  const double *in = a.Ws + cu * (size_t)Y * 2 * B + r;
#pragma unroll
  for (int g = 0; g < R; g++) {
    const double2 u = *reinterpret_cast<const double2 *>(in + (size_t)(2 * (32 * g + j)) * B), w = *reinterpret_cast<const double2 *>(in + (size_t)(2 * (32 * g + jp) + 1) * B);
    lo[0][g] = u.x;
    lo[1][g] = u.y;
    hi[0][g] = w.x;
    hi[1][g] = w.y;
  }
  // inverse stages 5 .. 5+S1-1: element g <-> block j + 32 g, block of stage 5 + k: g >> (k + 1)
#pragma unroll
  for (int c = 0; c < 2; c++) {
    reg_inv_stages<S1, true>(lo[c], mod, P.imask[S1 + 5] >> 5, [&](int k, int i) { return itw[(R >> (k + 1)) + i]; });
    reg_inv_stages<S1, true>(hi[c], mod, P.imask[S1 + 5] >> 5, [&](int k, int i) { return itw[(R >> (k + 1)) + i]; });
  }
#pragma unroll
  for (int e = 0; e < R; e++) {
    const size_t t = (size_t)(j + 32 * e) * B + r;  // output coefficient (and t + 1): block k = j + 32 e
    double o0 = lo[0][e], o1 = lo[1][e];
    if (j >= 1) {  // block k - 1 = jp + 32 e
      o0 += hi[0][e];
      o1 += hi[1][e];
    } else if (e >= 1) {  // block k - 1 = 31 + 32 (e - 1)
      o0 += hi[0][e - 1];
      o1 += hi[1][e - 1];
    }
    if (DST == BD_NEWTON) {
      if (t < M) {
        const double2 f = *reinterpret_cast<const double2 *>(P.invfact + t);
        *reinterpret_cast<double2 *>(a.dst + col * M + t) = make_double2(f.x != 0.0 ? reduce(o0, mod) : 0.0, f.y != 0.0 ? reduce(o1, mod) : 0.0);
      }
    } else if (DST == BD_PLAIN_SCALED) {
      *reinterpret_cast<double2 *>(a.dst + col * 2 * M + t) = make_double2(mulmod(reduce(o0, mod), P.b2_inv, mod), mulmod(reduce(o1, mod), P.b2_inv, mod));
    } else if (DST == BD_HFIN) {
      const long long top = (long long)a.m - 2;
      if ((long long)t <= top) a.dst[col * M + (size_t)(top - (long long)t)] = reduce(o0, mod);
      if ((long long)t + 1 <= top) a.dst[col * M + (size_t)(top - (long long)t - 1)] = reduce(o1, mod);
    } else {
      double2 *p = reinterpret_cast<double2 *>(a.dst + col * M + unit * ((size_t)1 << a.l) + t);
      const double2 d = *p;
      const double f0 = reduce(o0 + d.x, mod), f1 = reduce(o1 + d.y, mod);
      *p = DST == BD_COMBINE_CANON ? make_double2(canon(f0, mod), canon(f1, mod)) : make_double2(f0, f1);
    }
  }
}

}  // namespace rs
