// witness_tree_wide.hpp -- the product tree on 2^13 / 2^14 tiles in the wide form: 32 coefficients per thread (witness.hip)
#pragma once
#include "witness_cols.hpp"

namespace rs {

// =============================================================================================
// Product-tree levels 1..13 on a 2^13 tile in the wide form (g_witness_tree_ct == 2): 256 threads x 32 coefficients.
//
// tree_columns_kernel gives a lane 16 coefficients spread over its wave's block, runs every level's transforms in
// radix-8 LDS rounds (138 tile passes per tile) and spends 40 % of its VALU instructions on addresses and selects.
// Here the tile lives in LDS between levels and a level l (nodes of n = 2^l coefficients = W = n/32 threads) is
//     read  "cross" layout   a thread holds, for 32/W values of e, ALL W elements 32*tn + e of its node: the right half
//                            is the transform's input (F_right, 0), the left half waits in registers (F_left)
//     l-5 cross stages       in registers; their twiddles depend on the register index only: scalar operands
//     exchange               to the consecutive layout (a thread holds 32 consecutive coefficients)
//     last 5 forward stages, the product with the spectrum of D_left, first 5 inverse stages: in registers, the lane's
//                            own twiddles fetched from the L1/L2-resident tables
//     exchange               back to the cross layout
//     l-5 inverse cross stages, + F_left, reduce -> written back in place
// i.e. six tile passes per level (ten for l >= 11, whose 6..8 cross stages take two rounds) instead of 10..22, levels
// 1..5 entirely in registers, and compile-time addresses throughout.  Same stages, reduction points and products as
// tree_levels_lds: the stored values are identical.
// LDS address of tile position p: p + p/32 (a thread's 32 consecutive coefficients start 33 words apart).
// =============================================================================================
__device__ __forceinline__ int tw_addr(int p) { return p + (p >> 5); }
template <bool WG>
__device__ __forceinline__ void tw_sync() {
  if (WG)
    __syncthreads();
  else
    wave_sync();
}
// forward stages of a register tile whose upper half is zero padding: stage 0 is a copy (x + w*0, x - w*0)
template <int R, class TwFn>
__device__ __forceinline__ void reg_fwd_stages_zu(double (&v)[1 << R], const Mod mod, uint32_t red_mask, TwFn tw) {
  constexpr int E = 1 << R;
  if (red_mask & 1u) {
#pragma unroll
    for (int e = 0; e < E / 2; e++) v[e] = reduce(v[e], mod);
  }
#pragma unroll
  for (int e = 0; e < E / 2; e++) v[e + E / 2] = v[e];
#pragma unroll
  for (int k = 1; k < R; k++) {
    if ((red_mask >> k) & 1u) {
#pragma unroll
      for (int e = 0; e < E; e++) v[e] = reduce(v[e], mod);
    }
    const int half = E >> (k + 1);
#pragma unroll
    for (int blk = 0; blk < (1 << k); blk++) {
      const double w = tw(k, blk);
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
}
// 2^k consecutive table entries -> registers, 16-byte loads where the run allows
template <int CNT>
__device__ __forceinline__ void tw_run(const double *__restrict__ p, double *dst) {
#ifdef RS_TREEW_ABLATE_TW  // experiment: no per-lane table traffic (wrong results)
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
// The lane's own twiddles of the middle of level LV (thread u of the node): forward stages LV-5..LV-1 / inverse stages 0..4.
// (Requesting them earlier -- before the cross round, before the product -- was tried: no gain in time, and the extra
// live registers push F_left of levels 11..13 into scratch, 100 GiB of HBM traffic per proof.)
template <int LV>
__device__ __forceinline__ void tree_wide_mid_tw_fwd(const ColPlan &P, int u, double (&w)[31]) {
  constexpr int c = LV - 5;
  const double *__restrict__ tw = P.tw;
  tw_run<1>(tw + (1 << c) + u, w);
  tw_run<2>(tw + (2 << c) + (u << 1), w + 1);
  tw_run<4>(tw + (4 << c) + (u << 2), w + 3);
  tw_run<8>(tw + (8 << c) + (u << 3), w + 7);
  tw_run<16>(tw + (16 << c) + (u << 4), w + 15);
}
template <int LV>
__device__ __forceinline__ void tree_wide_mid_tw_inv(const ColPlan &P, int u, double (&w)[31]) {
  // inverse stage k: block (32 u + e) >> (k+1) of the n >> (k+1) blocks
  const double *__restrict__ itw = P.itw;
  constexpr int n = 1 << LV;
  tw_run<16>(itw + (n >> 1) + (u << 4), w);
  tw_run<8>(itw + (n >> 2) + (u << 3), w + 16);
  tw_run<4>(itw + (n >> 3) + (u << 2), w + 24);
  tw_run<2>(itw + (n >> 4) + (u << 1), w + 28);
  tw_run<1>(itw + (n >> 5) + u, w + 30);
}
// The middle of a level on the consecutive layout: forward stages l-5..l-1, product with the spectrum of D_left,
// inverse stages 0..4.  u = thread index inside the node (0 for l = 5), b = the thread's 32 coefficients.
// dh_wave: the level's table at the first coefficient of the WAVE (2048 consecutive entries for its 64 threads); they are
// fetched with coalesced 16-byte loads and handed to their owners through the wave's own (at this point free) region of
// the tile -- a thread fetching its own 256-byte run touches 64 different lines per instruction.
template <int LV, bool ZU>
__device__ __forceinline__ void tree_wide_middle(double (&b)[32], double *s, const ColPlan &P, const Mod mod,
                                                 const double *__restrict__ dh_wave, int u) {
  constexpr int c = LV - 5;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  {
    const double2 *src = reinterpret_cast<const double2 *>(dh_wave) + lane;
    // tw_addr(2048 wave + 128 i + 2 lane) = pa0 + 132 i.  The base is made opaque HERE, per level: left to itself the compiler
    // hoists the sixteen addresses out of the levels as kernel-lifetime registers, spills them (10 dwords at 256 VGPRs), and
    // every reload -- a vector-memory operation -- puts an s_waitcnt vmcnt(0) behind the table load issued just before it:
    // the sixteen loads of a level's table were serialised (round 6; found in the ISA).
    int pa0 = tw_addr(2048 * wave + 2 * lane);
    asm volatile("" : "+v"(pa0));
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const double2 v = src[64 * i];
      s[pa0 + 132 * i] = v.x;
      s[pa0 + 132 * i + 1] = v.y;
    }
  }
  const uint32_t fmask = P.fmask[LV] >> c, imask = P.imask[LV];
  double w[31];
  tree_wide_mid_tw_fwd<LV>(P, u, w);
  if (ZU)
    reg_fwd_stages_zu<5>(b, mod, fmask, [&](int k, int blk) { return w[(1 << k) - 1 + blk]; });
  else
    reg_fwd_stages<5, true>(b, mod, fmask, [&](int k, int blk) { return w[(1 << k) - 1 + blk]; });
  wave_sync();
  if ((P.pwmask >> LV) & 1u) {  // primes above ~2^46: the spectrum may exceed the multiplier range (a guarded pass, not a select)
#pragma unroll
    for (int e = 0; e < 32; e++) b[e] = reduce(b[e], mod);
  }
#pragma unroll
  for (int e = 0; e < 32; e++) b[e] = mulmod(b[e], s[33 * t + e], mod);
  tree_wide_mid_tw_inv<LV>(P, u, w);
  reg_inv_stages<5, true>(b, mod, imask, [&](int k, int i) { return w[32 - (32 >> k) + i]; });
}

// One level 6 <= LV <= 10: the node's W = 2^(LV-5) <= 32 threads, one cross round of LV-5 stages each way.
template <int LV>
__device__ __forceinline__ void tree_wide_level(double *s, const ColPlan &P, const Mod mod, const double *__restrict__ dh_tile, int t) {
  constexpr int c = LV - 5, W = 1 << c, Q = 32 / W;
  constexpr bool WG = false;  // a node is at most 32 threads: wave-private exchanges
  const int u = t & (W - 1), tb = t - u;
  const double *__restrict__ tw = P.tw;
  const double *__restrict__ itw = P.itw;
  double X[Q][W], Lf[Q][W / 2];
#pragma unroll
  for (int k = 0; k < Q; k++)
#pragma unroll
    for (int tn = 0; tn < W / 2; tn++) {
      Lf[k][tn] = s[tw_addr(32 * (tb + tn) + u + W * k)];
      X[k][tn] = s[tw_addr(32 * (tb + tn + W / 2) + u + W * k)];
    }
#pragma unroll
  for (int k = 0; k < Q; k++)
    reg_fwd_stages_zu<c>(X[k], mod, P.fmask[LV], [&](int st, int blk) { return ld_const(tw + (1 << st) + blk); });  // compile-time index: scalar cache (ntt_wide.hpp ld_const)
#pragma unroll
  for (int k = 0; k < Q; k++)
#pragma unroll
    for (int tn = 0; tn < W; tn++) s[tw_addr(32 * (tb + tn) + u + W * k)] = X[k][tn];
  tw_sync<WG>();
  {
    double b[32];
#pragma unroll
    for (int e = 0; e < 32; e++) b[e] = s[33 * t + e];
    tree_wide_middle<LV, false>(b, s, P, mod, dh_tile + 2048 * (t >> 6), u);
#pragma unroll
    for (int e = 0; e < 32; e++) s[33 * t + e] = b[e];
  }
  tw_sync<WG>();
#pragma unroll
  for (int k = 0; k < Q; k++)
#pragma unroll
    for (int tn = 0; tn < W; tn++) X[k][tn] = s[tw_addr(32 * (tb + tn) + u + W * k)];
#pragma unroll
  for (int k = 0; k < Q; k++) {
    reg_inv_stages<c, true>(X[k], mod, P.imask[LV] >> 5, [&](int st, int i) { return ld_const(itw + (W >> (st + 1)) + i); });
#pragma unroll
    for (int tn = 0; tn < W; tn++) {
      const double v = reduce(X[k][tn] + (tn < W / 2 ? Lf[k][tn] : 0.0), mod);
      s[tw_addr(32 * (tb + tn) + u + W * k)] = v;
    }
  }
  wave_sync();  // the next level's nodes are at most 64 threads = one wave
}

// One level 11 <= LV <= 13: W = 64..256 threads, LV-5 = 6..8 cross stages in two rounds (the top LV-10 over tn_hi, then
// five over tn_lo; a thread index inside the node is tn = 32 tn_hi + tn_lo).
template <int LV>
__device__ __forceinline__ void tree_wide_level_big(double *s, const ColPlan &P, const Mod mod, const double *__restrict__ dh_tile, int t) {
  constexpr int c = LV - 5, c1 = c - 5, R1 = 1 << c1, W = 1 << c, Q1 = 32 / R1, n = 1 << LV;
  constexpr bool WG = LV >= 12;  // level 11: the node is one wave
  const int a = t & (W - 1), tb = t - a;
  const double *__restrict__ tw = P.tw;
  const double *__restrict__ itw = P.itw;
  const uint32_t fmask = P.fmask[LV], imask = P.imask[LV];
  const int th2 = a >> 5, e2 = a & 31;  // round X2: thread (tn_hi, e) holds all 32 tn_lo
  // round X1: register (k, tn_hi) = element (tn_hi, m = a + W k), m = 32 tn_lo + e
  double X[Q1][R1], Lf[Q1][R1 / 2];
  auto x1_addr = [&](int k, int tn_hi) {
    const int m = a + W * k;
    return tw_addr(32 * (tb + 32 * tn_hi + (m >> 5)) + (m & 31));
  };
#pragma unroll
  for (int k = 0; k < Q1; k++)
#pragma unroll
    for (int th = 0; th < R1 / 2; th++) {
      Lf[k][th] = s[x1_addr(k, th)];
      X[k][th] = s[x1_addr(k, th + R1 / 2)];
    }
#pragma unroll
  for (int k = 0; k < Q1; k++) {
    reg_fwd_stages_zu<c1>(X[k], mod, fmask, [&](int st, int blk) { return ld_const(tw + (1 << st) + blk); });  // compile-time index: scalar cache (ntt_wide.hpp ld_const)
#pragma unroll
    for (int th = 0; th < R1; th++) s[x1_addr(k, th)] = X[k][th];
  }
  tw_sync<WG>();
  {
    double y[32], w2[31];
    // stage c1 + k: block (tn >> (5 - k)) = (tn_hi << k) + (tn_lo >> (5 - k))
    tw_run<1>(tw + (1 << c1) + th2, w2);
    tw_run<2>(tw + (2 << c1) + (th2 << 1), w2 + 1);
    tw_run<4>(tw + (4 << c1) + (th2 << 2), w2 + 3);
    tw_run<8>(tw + (8 << c1) + (th2 << 3), w2 + 7);
    tw_run<16>(tw + (16 << c1) + (th2 << 4), w2 + 15);
#pragma unroll
    for (int tl = 0; tl < 32; tl++) y[tl] = s[tw_addr(32 * (tb + 32 * th2 + tl) + e2)];
    reg_fwd_stages<5, true>(y, mod, fmask >> c1, [&](int k, int blk) { return w2[(1 << k) - 1 + blk]; });
#pragma unroll
    for (int tl = 0; tl < 32; tl++) s[tw_addr(32 * (tb + 32 * th2 + tl) + e2)] = y[tl];
  }
  tw_sync<WG>();
  {
    double b[32];
#pragma unroll
    for (int e = 0; e < 32; e++) b[e] = s[33 * t + e];
    tree_wide_middle<LV, false>(b, s, P, mod, dh_tile + 2048 * (t >> 6), a);
#pragma unroll
    for (int e = 0; e < 32; e++) s[33 * t + e] = b[e];
  }
  tw_sync<WG>();
  {
    double y[32], w2[31];
    // inverse stage 5 + k: block tn >> (k+1) = (tn_hi << (4-k)) + (tn_lo >> (k+1)) of the n >> (6+k)
    tw_run<16>(itw + (n >> 6) + (th2 << 4), w2);
    tw_run<8>(itw + (n >> 7) + (th2 << 3), w2 + 16);
    tw_run<4>(itw + (n >> 8) + (th2 << 2), w2 + 24);
    tw_run<2>(itw + (n >> 9) + (th2 << 1), w2 + 28);
    tw_run<1>(itw + (n >> 10) + th2, w2 + 30);
#pragma unroll
    for (int tl = 0; tl < 32; tl++) y[tl] = s[tw_addr(32 * (tb + 32 * th2 + tl) + e2)];
    reg_inv_stages<5, true>(y, mod, imask >> 5, [&](int k, int i) { return w2[32 - (32 >> k) + i]; });
#pragma unroll
    for (int tl = 0; tl < 32; tl++) s[tw_addr(32 * (tb + 32 * th2 + tl) + e2)] = y[tl];
  }
  tw_sync<WG>();
#pragma unroll
  for (int k = 0; k < Q1; k++) {
#pragma unroll
    for (int th = 0; th < R1; th++) X[k][th] = s[x1_addr(k, th)];
    reg_inv_stages<c1, true>(X[k], mod, imask >> 10, [&](int st, int i) { return ld_const(itw + (R1 >> (st + 1)) + i); });
#pragma unroll
    for (int th = 0; th < R1; th++) s[x1_addr(k, th)] = reduce(X[k][th] + (th < R1 / 2 ? Lf[k][th] : 0.0), mod);
  }
  tw_sync<(LV >= 11)>();  // levels 12, 13: nodes of 2 and 4 waves
}

// levels 1..4 of one 16-coefficient node at column position gpos, in registers (the arithmetic of school_levels_lds)
__device__ __forceinline__ void tree_school16(double (&v)[16], int gpos, int logM, const ColPlan &P) {
  const Mod mod = P.mod;
  const int dstride = (1 << logM) / 2 + 1;
#pragma unroll
  for (int l = 1; l <= SCHOOL_LEVELS; l++) {
    const int n = 1 << l, h = n >> 1;
#pragma unroll
    for (int sub = 0; sub < (16 >> l); sub++) {
      const int gnode = (gpos >> l) + sub;
      const double *dl = P.dlow + (size_t)l * dstride + (size_t)gnode * h;
      double out[16];
#pragma unroll
      for (int k = 0; k < n; k++) out[k] = 0.0;
#pragma unroll
      for (int b = 0; b < h; b++) {
        const double fr = v[sub * n + h + b];
        out[h + b] = addm(out[h + b], fr, mod);
#pragma unroll
        for (int a = 0; a < h; a++) out[a + b] = addm(out[a + b], mulmod(fr, dl[a], mod), mod);
      }
#pragma unroll
      for (int k = 0; k < n; k++) {
        const double left = (k < h) ? v[sub * n + k] : 0.0;
        v[sub * n + k] = reduce(addm(out[k], left, mod), mod);
      }
    }
  }
}

// LOGT = 13: 256 threads, two workgroups per CU; LOGT = 14: 512 threads, one workgroup per CU (the same 8 waves per CU)
// and one more level inside the tile -- one level less through the multi-pass transforms (two cross passes and a
// sub-transform pass over the whole column workspace).
// RF > 0 (round 5): the workgroup of a RIGHT tile -- the right child of a level-(LOGT+1) node, i.e. that level's whole
// transform input (F_right, 0) -- also runs the RF forward cross stages of that level (blocks of 2^(LOGT+1-RF) words; stage 0
// is a copy) on the finished tile and writes the node's 2^(LOGT+1) workspace words to Wout [ncols][2^logM]: the separate
// source pass cross_kernel<false, RF, CS_FILL_RIGHT> (a read of half the columns and a write of the workspace, 10 ms per
// headline proof) disappears into a kernel that leaves HBM idle.  Same values: the pass would read these canonical
// coefficients back and run the same stages.
template <int LOGT, int RF = 0>
__global__ void __launch_bounds__(1 << (LOGT - 5), 2)
tree_wide_kernel(double *__restrict__ cols, int logM, size_t col0, unsigned S, unsigned slots_per_limb, ColPlans plans,
                 double *__restrict__ Wout = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  const int t = threadIdx.x;
  const unsigned nb = 1u << (logM - LOGT);
  const size_t col = blockIdx.x / nb;
  const int pos0 = (int)(blockIdx.x % nb) << LOGT;
  const ColPlan &P = plans.l[((col0 + col) % S) / slots_per_limb];
  const Mod mod = P.mod;
  const size_t M = (size_t)1 << logM;
  double *c = cols + col * M + pos0;
  const int wave = t >> 6, lane = t & 63;
  {
    // The tile enters and leaves through the LDS tile in wave-sized transposes: a wave's 64 threads own 2048
    // consecutive coefficients, which it moves with fully coalesced 16-byte accesses (a thread reading or writing its
    // own 256-byte run directly touches each 128-byte line with eight separate 16-byte accesses: measured 4.8x the
    // written bytes at the memory interface).
    {
      const double2 *src = reinterpret_cast<const double2 *>(c + 2048 * wave) + lane;
      int pa0 = tw_addr(2048 * wave + 2 * lane);  // tw_addr(2048 wave + 128 i + 2 lane) = pa0 + 132 i
      asm volatile("" : "+v"(pa0));
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const double2 v = src[64 * i];
        s[pa0 + 132 * i] = v.x;
        s[pa0 + 132 * i + 1] = v.y;
      }
    }
    wave_sync();
    double r[32];
#pragma unroll
    for (int e = 0; e < 32; e++) r[e] = s[33 * t + e];
    wave_sync();
    // levels 1..4 (schoolbook) on the two 16-coefficient halves
    {
      double v[16];
#pragma unroll
      for (int j = 0; j < 2; j++) {
#pragma unroll
        for (int e = 0; e < 16; e++) v[e] = r[16 * j + e];
        tree_school16(v, pos0 + 32 * t + 16 * j, logM, P);
#pragma unroll
        for (int e = 0; e < 16; e++) r[16 * j + e] = v[e];
      }
    }
    // level 5: the node is the thread's own 32 coefficients
    {
      double b[32];
#pragma unroll
      for (int e = 0; e < 16; e++) b[e] = r[16 + e];
      tree_wide_middle<5, true>(b, s, P, mod, P.dhat + (size_t)5 * M + pos0 + 2048 * wave, 0);
#pragma unroll
      for (int e = 0; e < 32; e++) s[33 * t + e] = reduce(b[e] + (e < 16 ? r[e] : 0.0), mod);
    }
  }
  wave_sync();
  tree_wide_level<6>(s, P, mod, P.dhat + (size_t)6 * M + pos0, t);
  tree_wide_level<7>(s, P, mod, P.dhat + (size_t)7 * M + pos0, t);
  tree_wide_level<8>(s, P, mod, P.dhat + (size_t)8 * M + pos0, t);
  tree_wide_level<9>(s, P, mod, P.dhat + (size_t)9 * M + pos0, t);
  tree_wide_level<10>(s, P, mod, P.dhat + (size_t)10 * M + pos0, t);
  tree_wide_level_big<11>(s, P, mod, P.dhat + (size_t)11 * M + pos0, t);
  tree_wide_level_big<12>(s, P, mod, P.dhat + (size_t)12 * M + pos0, t);
  tree_wide_level_big<13>(s, P, mod, P.dhat + (size_t)13 * M + pos0, t);
  if (LOGT >= 14) tree_wide_level_big<(LOGT >= 14 ? 14 : 13)>(s, P, mod, P.dhat + (size_t)14 * M + pos0, t);
  {  // the last level ended with a workgroup barrier: every coefficient of the tile is final
    double2 *dst = reinterpret_cast<double2 *>(c + 2048 * wave) + lane;
    int pa0 = tw_addr(2048 * wave + 2 * lane);
    asm volatile("" : "+v"(pa0));
#pragma unroll
    for (int i = 0; i < 16; i++) dst[64 * i] = make_double2(canon(s[pa0 + 132 * i], mod), canon(s[pa0 + 132 * i + 1], mod));
  }
  if (RF > 0) {
    const unsigned ti = blockIdx.x % nb;
    if (ti & 1u) {  // a right child: the forward cross stages of the parent's transform (fwd_round2, s0 = 0, zero-padded input)
      constexpr int RR = RF > 0 ? RF : 1, E = 1 << RR, LOGB = LOGT + 1 - RR, B = 1 << LOGB;
      double *__restrict__ wnode = Wout + col * M + ((size_t)(ti >> 1) << (LOGT + 1));
      const double *__restrict__ tw = P.tw;
      const uint32_t fmask = P.fmask[LOGT + 1];
      for (int j = 2 * t; j < B; j += 2 << (LOGT - 5)) {
        double x[2][E];
#pragma unroll
        for (int e = 0; e < E / 2; e++) {
          const int pa = tw_addr(j + B * e);
          x[0][e] = canon(s[pa], mod);
          x[1][e] = canon(s[pa + 1], mod);  // j is even and below the tile's 32-word run boundary: the neighbour word
        }
#pragma unroll
        for (int k = 0; k < RR; k++) {
          if ((fmask >> k) & 1u) {
#pragma unroll
            for (int cc = 0; cc < 2; cc++)
#pragma unroll
              for (int e = 0; e < (k == 0 ? E / 2 : E); e++) x[cc][e] = reduce(x[cc][e], mod);
          }
          if (k == 0) {
#pragma unroll
            for (int cc = 0; cc < 2; cc++)
#pragma unroll
              for (int e = 0; e < E / 2; e++) x[cc][e + E / 2] = x[cc][e];
            continue;
          }
          const int half = E >> (k + 1);
#pragma unroll
          for (int blk = 0; blk < (1 << k); blk++) {
            const double wt = tw[(1 << k) + blk];
#pragma unroll
            for (int e0 = 0; e0 < half; e0++) {
              const int ia = blk * 2 * half + e0, ib = ia + half;
#pragma unroll
              for (int cc = 0; cc < 2; cc++) {
                const double tt = mulmod(x[cc][ib], wt, mod);
                const double z = x[cc][ia];
                x[cc][ia] = z + tt;
                x[cc][ib] = z - tt;
              }
            }
          }
        }
#pragma unroll
        for (int e = 0; e < E; e++) {
          typedef double V2 __attribute__((ext_vector_type(2)));
          V2 o;
          o.x = x[0][e];
          o.y = x[1][e];
          __builtin_nontemporal_store(o, reinterpret_cast<V2 *>(wnode + j + B * e));
        }
      }
    }
  }
}

}  // namespace rs
