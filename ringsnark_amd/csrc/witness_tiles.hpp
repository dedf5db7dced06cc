// witness_tiles.hpp -- single-tile column kernels of the witness map: a column (or a tile of it) lives in one LDS tile (witness.hip)
#pragma once
#include "witness_cols.hpp"

namespace rs {

// Product-tree levels 1..SCHOOL_LEVELS by schoolbook products in registers, on 2^logB consecutive
// Newton coefficients at column position pos0 held in the (offset) tile s; one thread per node of
// size 2^SCHOOL_LEVELS, executed by the lanes `ln` (a workgroup or one wave).
template <class CP>
__device__ __forceinline__ void school_levels_lds(typename CP::T *s, int logB, int logM, int pos0, const CP &P, const Lanes ln) {
  using T = typename CP::T;
  const typename CP::M mod = P.mod;
  const int Bn = 1 << logB, M = 1 << logM;
  const int lv = logB < SCHOOL_LEVELS ? logB : SCHOOL_LEVELS;
  const int nn = 1 << lv;
  const int dstride = M / 2 + 1;
  for (int node = ln.tid; node < (Bn >> lv); node += ln.nthr) {
    T v[1 << SCHOOL_LEVELS];
#pragma unroll
    for (int k = 0; k < (1 << SCHOOL_LEVELS); k++) v[k] = (k < nn) ? s[pidx(node * nn + k)] : T(0);
#pragma unroll
    for (int l = 1; l <= SCHOOL_LEVELS; l++) {
      if (l > lv) break;
      const int n = 1 << l, h = n >> 1;
#pragma unroll
      for (int sub = 0; sub < ((1 << SCHOOL_LEVELS) >> l); sub++) {
        if (sub * n >= nn) break;
        const int gnode = ((pos0 + node * nn) >> l) + sub;  // node index at level l within the column
        const T *dl = P.dlow + (size_t)l * dstride + (size_t)gnode * h;
        T out[1 << SCHOOL_LEVELS];
#pragma unroll
        for (int k = 0; k < n; k++) out[k] = T(0);
        // D_left * F_right, D_left = x^h + sum dl[a] x^a
#pragma unroll
        for (int b = 0; b < h; b++) {
          const T fr = v[sub * n + h + b];
          out[h + b] = addm(out[h + b], fr, mod);
#pragma unroll
          for (int a = 0; a < h; a++) out[a + b] = addm(out[a + b], mulmod(fr, dl[a], mod), mod);
        }
#pragma unroll
        for (int k = 0; k < n; k++) {
          const T left = (k < h) ? v[sub * n + k] : T(0);
          v[sub * n + k] = reduce(addm(out[k], left, mod), mod);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < (1 << SCHOOL_LEVELS); k++)
      if (k < nn) s[pidx(node * nn + k)] = v[k];
  }
}

// Newton -> monomial product tree on an LDS tile holding Bn = 2^logB consecutive Newton
// coefficients of a column, starting at column position pos0 (a multiple of Bn); the tile's
// second half [Bn, 2Bn) is scratch.  Runs levels 1..logB (node sizes 2..Bn).  Tables are
// indexed by the position inside the whole column (length M = 2^logM).
__host__ __device__ __forceinline__ int tree_scratch_offset(int Bn) { return Bn >= LDS_BLOCK_MIN ? Bn : LDS_BLOCK_MIN; }
template <class CP>
__device__ __forceinline__ void tree_levels_lds(typename CP::T *s, int logB, int logM, int pos0, const CP &P) {
  using T = typename CP::T;
  const typename CP::M mod = P.mod;
  const int Bn = 1 << logB, M = 1 << logM;
  school_levels_lds(s, logB, logM, pos0, P, block_lanes());
  __syncthreads();
  // transform levels: B[node] = (F_right, 0) -> batched length-n transforms -> * spectrum of D_left
  // -> inverse -> + F_left.  B is an offset tile starting at a multiple of LDS_BLOCK_MIN.
  T *Bt = s + pidx(tree_scratch_offset(Bn));
  for (int l = SCHOOL_LEVELS + 1; l <= logB; l++) {
    const int n = 1 << l, h = n >> 1;
    for (int i = threadIdx.x; i < Bn; i += blockDim.x) {
      const int k = i & (n - 1);
      Bt[pidx(i)] = (k < h) ? s[pidx(i + h)] : T(0);
    }
    __syncthreads();
    lds_bntt_fwd(Bt, logB, l, P.tw, mod, P.fmask[l]);
    const T *dh = P.dhat + (size_t)l * M + pos0;
    for (int i = threadIdx.x; i < Bn; i += blockDim.x) Bt[pidx(i)] = mulmod(reduce(Bt[pidx(i)], mod), dh[i], mod);
    __syncthreads();
    lds_bntt_inv(Bt, logB, l, P.itw, mod, P.imask[l]);
    for (int i = threadIdx.x; i < Bn; i += blockDim.x) {
      const int k = i & (n - 1);
      const T left = (k < h) ? s[pidx(i)] : T(0);
      s[pidx(i)] = reduce(addm(Bt[pidx(i)], left, mod), mod);
    }
    __syncthreads();
  }
}

// One workgroup per column: values at 0..m-1 (cols[col][0..M)) -> monomial coefficients in place.
// LDS: 2M padded doubles (A = [0,M) current polynomials, B = [M,2M) scratch).
// Column c belongs to limb (c % S) / slots_per_limb (several vectors of S columns are batched).
template <class CPS>
__global__ void __launch_bounds__(1024)
interp_columns_kernel(typename CPS::T *__restrict__ cols, int logM, unsigned S, unsigned slots_per_limb, CPS plans) {
  using T = typename CPS::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *s = reinterpret_cast<T *>(smem);
  const int M = 1 << logM;
  const size_t col = blockIdx.x;
  const ColPlanT<typename CPS::M> &P = plans.l[(col % S) / slots_per_limb];
  const typename CPS::M mod = P.mod;
  T *c = cols + col * (size_t)M;
  // 1. g_j = y_j / j!  (zero for j >= m), zero-padded to 2M
  for (int j = threadIdx.x; j < M; j += blockDim.x) {
    s[pidx(j)] = mulmod(c[j], P.invfact[j], mod);
    s[pidx(M + j)] = T(0);
  }
  __syncthreads();
  lds_ntt_fwd<4>(s, logM + 1, P.tw, 1, mod, P.fwd_mask2);
  for (int j = threadIdx.x; j < 2 * M; j += blockDim.x) s[pidx(j)] = mulmod(reduce(s[pidx(j)], mod), P.ehat[j], mod);
  __syncthreads();
  lds_ntt_inv<4>(s, logM + 1, P.itw, 1, mod, P.inv_mask2);
  // Newton coefficients f_k = s[k], k < m; everything at k >= m is discarded (invfact is zero
  // there only for the INPUT; the convolution tail must be cleared explicitly).
  for (int j = threadIdx.x; j < M; j += blockDim.x) {
    const T inv_nonzero = P.invfact[j];
    s[pidx(j)] = (inv_nonzero != T(0)) ? reduce(s[pidx(j)], mod) : T(0);
  }
  __syncthreads();
  tree_levels_lds(s, logM, logM, 0, P);
  for (int j = threadIdx.x; j < M; j += blockDim.x) c[j] = canon(s[pidx(j)], mod);
}


// radix of the LDS rounds of the product tree's level transforms (stages per LDS round trip)
#ifndef RS_TREE_MAXR
#define RS_TREE_MAXR 3
#endif
// Source / sink functors of the product tree's wave-private levels (block-local indices).
// First forward round of a level-l transform: element offset eoff inside the node is a left
// position iff eoff < h; the transform's input there is F_right (the node's right half), zero above.
struct TreeRightIn {
  static constexpr bool zero_upper = true;
  const double *sb;
  int h;
  __device__ __forceinline__ int pbase(int base) const { return base; }
  __device__ __forceinline__ double load(int base, int, int eoff, int) const {
    return eoff < h ? sb[pidx(base + eoff + h)] : 0.0;
  }
};
// Last forward round: spectrum of F_right times the precomputed spectrum of D_left.
struct TreeMulOut {
  double *sb;
  const double *dh;  // level table at this block
  Mod mod;
  __device__ __forceinline__ int pbase(int base) const { return pidx(base); }
  __device__ __forceinline__ void store(int base, int pb, int eoff, int poff, double v) const {
    sb[pcomb(pb, poff)] = mulmod(reduce(v, mod), dh[base + eoff], mod);
  }
};

struct TreeMulFactory {  // per-block TreeMulOut (dh_tile: the level table at the tile's first coefficient)
  double *s;
  const double *dh_tile;
  Mod mod;
  __device__ __forceinline__ TreeMulOut operator()(int off) const { return TreeMulOut{s + pidx(off), dh_tile + off, mod}; }
};

// Newton -> monomial, one column per workgroup, tile = M doubles only (two workgroups per CU):
// a level's F_left values wait in registers while the node regions are overwritten in place with
// (F_right, 0), transformed, multiplied by the spectrum of D_left and transformed back.  Wave w
// owns block w of M/W coefficients; every level whose nodes fit a block (n <= M/W) runs without a
// single workgroup barrier.
// LOGT_CT != 0: the tile size is a compile-time constant and the level loop is unrolled, so every
// round of every level is specialised (constant gaps, radices and masks of addresses).
// NEWTON (single-tile columns only, logT == logM): the tile starts as VALUES at the nodes and the
// kernel first converts them to Newton coefficients, f = low half of g * e with g_j = y_j / j!,
// e_i = (-1)^i / i!.  The length-2M cyclic convolution is never formed: the 2M-point transform of a
// zero-padded input is the pair of M-point sub-transforms rooted at decimation-tree nodes 2 (the
// cyclic one: bins [0, M) of the table `ehat`) and 3 (the negacyclic one: bins [M, 2M)), and the low
// half of the inverse is the SUM of the two M-point inverses -- two passes over an M-sized tile, so
// the whole interpolation of a column runs in ONE launch at two workgroups per CU.
template <int THREADS, int LOGT_CT = 0, bool NEWTON = false>
__global__ void __launch_bounds__(THREADS, THREADS == 1024 ? 4 : THREADS / 128)  // two workgroups per CU (1024 threads: one, a 2^14 tile)
tree_columns_kernel(double *__restrict__ cols, int logM, int logT_arg, size_t col0, unsigned S, unsigned slots_per_limb,
                    ColPlans plans) {
  const int logT = LOGT_CT ? LOGT_CT : logT_arg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  constexpr int LOGW = THREADS == 1024 ? 4 : (THREADS == 512 ? 3 : (THREADS == 256 ? 2 : (THREADS == 128 ? 1 : 0)));
  constexpr int EPT = 16;  // coefficients per lane: T / THREADS <= 16
  const int M = 1 << logM;
  // workgroup = one tile of T = 2^logT coefficients: levels 1..logT of the tree below position pos0
  const unsigned nb = 1u << (logM - logT);
  const size_t col = blockIdx.x / nb;
  const int pos0 = (int)(blockIdx.x % nb) << logT;
  const ColPlan &P = plans.l[((col0 + col) % S) / slots_per_limb];
  const Mod mod = P.mod;
  double *c = cols + col * (size_t)M + pos0;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int logb = logT - LOGW, bsz = 1 << logb, off = wave << logb;
  const int per = bsz >> 6;  // own positions: off + lane + 64*j, j < per
  double *sb = s + pidx(off);
  const LdsBlockIO blk{sb};
  const Lanes wl = wave_lanes();
  if (NEWTON) {
    const ColBlockFactory bf{s};
    const LdsIO lds{s};
    double u[EPT];
#pragma unroll
    for (int half = 0; half < 2; half++) {
      // tile <- g (recomputed from the column for the second pass: the first one overwrote it).
      // `ln`: fresh copies of the lane index keep the 16 tile addresses of each phase from being
      // hoisted over the transforms, spilled and reloaded one by one.
      int ln = lane;
      asm volatile("" : "+v"(ln));
      int p0 = pidx(ln);
#pragma unroll
      for (int j = 0; j < EPT; j++)
        if (j < per) {
          const int k = off + ln + 64 * j;
          sb[own_pidx(p0, ln, j)] = mulmod(c[k], P.invfact[k], mod);
        }
      __syncthreads();
      lds_ntt_fwd_wp<3, LdsIO, ColBlockFactory, 3>(s, lds, bf, logT, LOGW, P.tw, mod, P.fwd_mask2 >> 1, 2 + half);
      const double *eh = P.ehat + (size_t)half * M + off;
      ln = lane;
      asm volatile("" : "+v"(ln));
      p0 = pidx(ln);
#pragma unroll
      for (int j = 0; j < EPT; j++)
        if (j < per) {
          const int pi = own_pidx(p0, ln, j);
          sb[pi] = mulmod(reduce(sb[pi], mod), eh[ln + 64 * j], mod);
        }
      wave_sync();
      lds_ntt_inv_wp<3, ColBlockFactory, LdsIO, 3>(s, bf, lds, logT, LOGW, P.itw, mod, P.inv_mask2, 2 + half);
      if (half == 0) {
        ln = lane;
        asm volatile("" : "+v"(ln));
        p0 = pidx(ln);
#pragma unroll
        for (int j = 0; j < EPT; j++)
          if (j < per) u[j] = sb[own_pidx(p0, ln, j)];
        __syncthreads();  // every wave has saved its block before the tile is refilled
      }
    }
    // Newton coefficients k < m; the convolution tail is discarded
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int p0 = pidx(ln);
#pragma unroll
    for (int j = 0; j < EPT; j++)
      if (j < per) {
        const int pi = own_pidx(p0, ln, j);
        sb[pi] = (P.invfact[off + ln + 64 * j] != 0.0) ? reduce(u[j] + sb[pi], mod) : 0.0;
      }
  } else {
#pragma unroll
    for (int j = 0; j < EPT; j++)
      if (j < per) sb[pidx(lane + 64 * j)] = c[off + lane + 64 * j];
  }
  wave_sync();
  school_levels_lds(sb, logb, logM, pos0 + off, P, wl);
  wave_sync();
  // this lane's 16 coefficients travel from level to level in registers: a level's F_left values are
  // exactly what its predecessor's recombination just wrote at the same positions
  double r[EPT];
#pragma unroll
  for (int j = 0; j < EPT; j++)
    if (j < per) r[j] = sb[pidx(lane + 64 * j)];
#pragma unroll LOGT_CT ? 32 : 1
  for (int l = SCHOOL_LEVELS + 1; l <= (LOGT_CT ? LOGT_CT : 20); l++) {
    if (l > logT) break;
    const int n = 1 << l, h = n >> 1;
    const bool priv = l <= logb;
    // fresh copy per level: otherwise the 16 tile addresses are hoisted out of the level loop,
    // spilled, and every use becomes a serialised scratch reload (s_waitcnt vmcnt(0))
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int p0 = pidx(ln);
    const double *dh = P.dhat + (size_t)l * M + pos0 + off;
    if (priv) {
      // Nodes inside the wave's block, no workgroup barrier.  The first forward round reads
      // (F_right, 0) straight out of the right halves, the last one multiplies by the spectrum of
      // D_left on its way back to the tile: no separate split and pointwise passes.
      wave_sync();
      const TreeRightIn rin{sb, h};
      const TreeMulOut mout{sb, dh, mod};
      for (int st = 0; st < l;) {
        const int R = pick_radix(l - st, RS_TREE_MAXR);
        if (st == 0)
          fwd_round_dispatch<RS_TREE_MAXR>(R, rin, blk, logb, l, st, P.tw, 1, mod, P.fmask[l], wl);
        else if (st + R >= l)
          fwd_round_dispatch<RS_TREE_MAXR>(R, blk, mout, logb, l, st, P.tw, 1, mod, P.fmask[l], wl);
        else
          fwd_round_dispatch<RS_TREE_MAXR>(R, blk, blk, logb, l, st, P.tw, 1, mod, P.fmask[l], wl);
        wave_sync();
        st += R;
      }
      for (int st = 0; st < l;) {
        const int R = pick_radix(l - st, RS_TREE_MAXR);
        inv_round_dispatch<RS_TREE_MAXR>(R, blk, blk, logb, l, st, P.itw, 1, mod, P.imask[l], wl);
        wave_sync();
        st += R;
      }
    } else {
      // nodes span 2^(l - logb) waves: only that many top stages cross waves (workgroup barriers); the
      // rest of the forward transform, and the bottom of the inverse, stay inside the wave's block
      __syncthreads();
      lds_bntt_fwd_wp<RS_TREE_MAXR, TreeRightIn, TreeMulFactory, 3>(s, TreeRightIn{s, h}, TreeMulFactory{s, dh - off, mod}, logT, LOGW, l,
                                                         P.tw, mod, P.fmask[l]);
      lds_bntt_inv_wp<RS_TREE_MAXR, ColBlockFactory, LdsIO, 3>(s, ColBlockFactory{s}, LdsIO{s}, logT, LOGW, l, P.itw, mod, P.imask[l]);
    }
#pragma unroll
    for (int j = 0; j < EPT; j++)
      if (j < per) {
        const int i = off + ln + 64 * j, pi = own_pidx(p0, ln, j);
        r[j] = reduce(sb[pi] + (((i & (n - 1)) < h) ? r[j] : 0.0), mod);
        sb[pi] = r[j];
      }
    if (priv) wave_sync(); else __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < EPT; j++)
    if (j < per) c[off + lane + 64 * j] = canon(r[j], mod);
}

// H = quo(A*B, Z) per column + the ZK patch of r1cs_to_qrp.tcc:230-235.  The reference divides
// A*B - C by Z and drops the remainder (Boost long division, util/polynomials.tcc:76-81); since
// deg C < deg Z, quo(A*B - C, Z) = quo(A*B, Z): C is not needed.  With P = A*B (degree 2m-2):
//     rev(H) = rev(P) * rev(Z)^-1  mod x^(m-1)
// i.e. five length-2M cyclic transforms per column against the precomputed spectrum `shat`.
// A, B: [cols][M] canonical doubles; H: [cols][M].  d1,d2,d3: ring elements [L][N] (u64) or NULL.
template <class CPS>
__global__ void __launch_bounds__(1024)
h_columns_kernel(const typename CPS::T *__restrict__ A, const typename CPS::T *__restrict__ Bc, typename CPS::T *__restrict__ H,
                 int logM, int m, unsigned slots_per_limb, CPS plans, const uint64_t *__restrict__ d1,
                 const uint64_t *__restrict__ d2, const uint64_t *__restrict__ d3, ColMap cm) {
  using T = typename CPS::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *s = reinterpret_cast<T *>(smem);
  const int M = 1 << logM, M2 = 2 * M;
  const size_t col = blockIdx.x;
  const ColPlanT<typename CPS::M> &P = plans.l[col / slots_per_limb];
  const typename CPS::M mod = P.mod;
  const T *srcA = A + col * (size_t)M, *srcB = Bc + col * (size_t)M;
  T r[16];  // this thread's slice of a spectrum: positions tid + k*blockDim
  for (int pass = 0; pass < 2; pass++) {
    const T *src = pass ? srcB : srcA;
    for (int k = threadIdx.x; k < M; k += blockDim.x) {
      s[pidx(k)] = center(src[k], mod);
      s[pidx(M + k)] = T(0);
    }
    __syncthreads();
    lds_ntt_fwd<4>(s, logM + 1, P.tw, 1, mod, P.fwd_mask2);
    int tid = threadIdx.x;  // fresh copy per pass: keeps the 16 tile addresses from being hoisted and spilled
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int p = tid + k * blockDim.x;
      if (p < M2) {
        const T v = reduce(s[pidx(p)], mod);
        r[k] = pass ? mulmod_dd(r[k], v, mod) : v;  // spectrum of A times spectrum of B: data x data
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int p = threadIdx.x + k * blockDim.x;
    if (p < M2) s[pidx(p)] = r[k];
  }
  __syncthreads();
  lds_ntt_inv<4>(s, logM + 1, P.itw, 1, mod, P.inv_mask2);  // 2M * (A*B), coefficients 0 .. 2m-2
  // T_i = P_{2m-2-i}, i < m-1, zero-padded
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int i = threadIdx.x + k * blockDim.x;
    if (i < M2) r[k] = (i < m - 1) ? reduce(s[pidx(2 * m - 2 - i)], mod) : T(0);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int i = threadIdx.x + k * blockDim.x;
    if (i < M2) s[pidx(i)] = r[k];
  }
  __syncthreads();
  lds_ntt_fwd<4>(s, logM + 1, P.tw, 1, mod, P.fwd_mask2);
  for (int i = threadIdx.x; i < M2; i += blockDim.x) s[pidx(i)] = mulmod(reduce(s[pidx(i)], mod), P.shat[i], mod);
  __syncthreads();
  lds_ntt_inv<4>(s, logM + 1, P.itw, 1, mod, P.inv_mask2);  // U_i = rev(H)_i, i < m-1
  T e1 = T(0), e2 = T(0), e3 = T(0), e12 = T(0);
  const bool zk = d1 != nullptr;
  if (zk) {
    int dlimb, dslot;
    cm.locate(col, dlimb, dslot);
    const size_t di = cm.in_index(dlimb, dslot);
    e1 = center(from_res<T>(d1[di]), mod);
    e2 = center(from_res<T>(d2[di]), mod);
    e3 = center(from_res<T>(d3[di]), mod);
    e12 = mulmod_dd(e1, e2, mod);
  }
  T *dst = H + col * (size_t)M;
  for (int k = threadIdx.x; k < M; k += blockDim.x) {
    T h = (k <= m - 2) ? reduce(s[pidx(m - 2 - k)], mod) : T(0);
    if (zk) {
      h = addm(h, addm(addm(mulmod_dd(e2, center(srcA[k], mod), mod), mulmod_dd(e1, center(srcB[k], mod), mod), mod),
                       mulmod(e12, P.ztab[k], mod), mod), mod);
      if (k == 0) h = subm(h, e3, mod);
    }
    dst[k] = canon(h, mod);
  }
}

// H on an M-sized tile (M >= 1024), two workgroups per CU, wave-private transforms.  Every
// length-2M transform of h_columns_kernel has a zero-padded input, so it is the pair of M-point
// sub-transforms rooted at decimation-tree nodes 2 and 3 (bins [0, M) and [M, 2M) of the spectra
// `shat`), and the inverse's low / high halves are the sum / difference of the two M-point inverses:
//     P = A*B:  u = inv2(fwd2 A . fwd2 B), v = inv3(fwd3 A . fwd3 B),  P_low = u + v, P_high = u - v
//     U = T*S mod x^(m-1):  U = inv2(fwd2 T . shat[0,M)) + inv3(fwd3 T . shat[M,2M))
// Ten M-point transforms instead of five 2M-point ones, none of them with all-workgroup barriers
// between rounds.  Lane l of wave w owns positions off + l + 64 j; `u` is parked in the output
// column (L2) while the second half runs.
template <int THREADS, int LOGM_CT = 0>
__global__ void __launch_bounds__(THREADS, THREADS == 1024 ? 4 : THREADS / 128)
h_tile_kernel(const double *__restrict__ A, const double *__restrict__ Bc, double *__restrict__ H, int logM_arg, int m,
              unsigned slots_per_limb, ColPlans plans, const uint64_t *__restrict__ d1, const uint64_t *__restrict__ d2,
              const uint64_t *__restrict__ d3, ColMap cm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  constexpr int LOGW = THREADS == 1024 ? 4 : (THREADS == 512 ? 3 : (THREADS == 256 ? 2 : (THREADS == 128 ? 1 : 0)));
  constexpr int EPT = 16;
  const int logM = LOGM_CT ? LOGM_CT : logM_arg;
  const int M = 1 << logM;
  const size_t col = blockIdx.x;
  const ColPlan &P = plans.l[col / slots_per_limb];
  const Mod mod = P.mod;
  const double *srcA = A + col * (size_t)M, *srcB = Bc + col * (size_t)M;
  double *dst = H + col * (size_t)M;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int logb = logM - LOGW, off = wave << logb;
  double *sb = s + pidx(off);
  const ColBlockFactory bf{s};
  const LdsIO lds{s};
  const uint32_t fmask = P.fwd_mask2 >> 1, imask = P.inv_mask2;
  // `ln`: fresh copies of the lane index keep each phase's 16 tile addresses from being hoisted over
  // the transforms, spilled and reloaded one by one
#define RS_FRESH_LANE()        \
  int ln = lane;               \
  asm volatile("" : "+v"(ln)); \
  const int p0 __attribute__((unused)) = pidx(ln)
  double r[EPT];
#pragma unroll
  for (int half = 0; half < 2; half++) {
    {
      RS_FRESH_LANE();
#pragma unroll
      for (int j = 0; j < EPT; j++) sb[pidx(ln + 64 * j)] = center(srcA[off + ln + 64 * j], mod);
    }
    __syncthreads();
    lds_ntt_fwd_wp<3, LdsIO, ColBlockFactory, 3>(s, lds, bf, logM, LOGW, P.tw, mod, fmask, 2 + half);
    {
      RS_FRESH_LANE();
#pragma unroll
      for (int j = 0; j < EPT; j++) r[j] = reduce(sb[pidx(ln + 64 * j)], mod);
    }
    __syncthreads();  // every wave has its slice of the spectrum of A before the tile is refilled
    {
      RS_FRESH_LANE();
#pragma unroll
      for (int j = 0; j < EPT; j++) sb[pidx(ln + 64 * j)] = center(srcB[off + ln + 64 * j], mod);
    }
    __syncthreads();
    lds_ntt_fwd_wp<3, LdsIO, ColBlockFactory, 3>(s, lds, bf, logM, LOGW, P.tw, mod, fmask, 2 + half);
    {
      RS_FRESH_LANE();
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        const int pi = pidx(ln + 64 * j);
        sb[pi] = mulmod(r[j], reduce(sb[pi], mod), mod);
      }
    }
    wave_sync();
    lds_ntt_inv_wp<3, ColBlockFactory, LdsIO, 3>(s, bf, lds, logM, LOGW, P.itw, mod, imask, 2 + half);
    if (half == 0) {
      RS_FRESH_LANE();
#pragma unroll
      for (int j = 0; j < EPT; j++) dst[off + ln + 64 * j] = reduce(sb[pidx(ln + 64 * j)], mod);  // park u
      __syncthreads();
    }
  }
  // T_k = P_{2m-2-k}, k < m-1, zero-padded, scattered into the tile from the own slices of
  // P_low = u + v (index i) and P_high = u - v (index i + M)
  {
    RS_FRESH_LANE();
#pragma unroll
    for (int j = 0; j < EPT; j++) r[j] = reduce(sb[pidx(ln + 64 * j)], mod);  // v
  }
  __syncthreads();
  {
    RS_FRESH_LANE();
#pragma unroll
    for (int j = 0; j < EPT; j++) sb[pidx(ln + 64 * j)] = 0.0;
  }
  __syncthreads();
  {
    RS_FRESH_LANE();
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      const int i = off + ln + 64 * j;
      const double u = dst[i];
      const int klo = 2 * m - 2 - i, khi = klo - M;
      if (klo >= 0 && klo < m - 1) s[pidx(klo)] = reduce(u + r[j], mod);
      if (khi >= 0 && khi < m - 1) s[pidx(khi)] = reduce(u - r[j], mod);
    }
  }
  __syncthreads();
  {
    RS_FRESH_LANE();
#pragma unroll
    for (int j = 0; j < EPT; j++) r[j] = sb[pidx(ln + 64 * j)];  // own slice of T, for the second half
  }
  __syncthreads();  // the cross-wave round below writes every block: all slices must be saved first
  double uu[EPT];
#pragma unroll
  for (int half = 0; half < 2; half++) {
    if (half == 1) {
      RS_FRESH_LANE();
#pragma unroll
      for (int j = 0; j < EPT; j++) sb[pidx(ln + 64 * j)] = r[j];
      __syncthreads();
    }
    lds_ntt_fwd_wp<3, LdsIO, ColBlockFactory, 3>(s, lds, bf, logM, LOGW, P.tw, mod, fmask, 2 + half);
    {
      RS_FRESH_LANE();
      const double *sh = P.shat + (size_t)half * M + off;
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        const int pi = pidx(ln + 64 * j);
        sb[pi] = mulmod(reduce(sb[pi], mod), sh[ln + 64 * j], mod);
      }
    }
    wave_sync();
    lds_ntt_inv_wp<3, ColBlockFactory, LdsIO, 3>(s, bf, lds, logM, LOGW, P.itw, mod, imask, 2 + half);
    if (half == 0) {
      RS_FRESH_LANE();
#pragma unroll
      for (int j = 0; j < EPT; j++) uu[j] = sb[pidx(ln + 64 * j)];
      __syncthreads();
    }
  }
  // U = uu + tile (own slice) back into the tile, then H_j = U_{m-2-j} + ZK patch
  {
    RS_FRESH_LANE();
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      const int pi = pidx(ln + 64 * j);
      sb[pi] = reduce(uu[j] + sb[pi], mod);
    }
  }
  __syncthreads();
  double e1 = 0.0, e2 = 0.0, e3 = 0.0, e12 = 0.0;
  const bool zk = d1 != nullptr;
  if (zk) {
    int dlimb, dslot;
    cm.locate(col, dlimb, dslot);
    const size_t di = cm.in_index(dlimb, dslot);
    e1 = center(from_u64(d1[di]), mod);
    e2 = center(from_u64(d2[di]), mod);
    e3 = center(from_u64(d3[di]), mod);
    e12 = mulmod(e1, e2, mod);
  }
  {
    RS_FRESH_LANE();
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      const int k = off + ln + 64 * j;
      double h = (k <= m - 2) ? s[pidx(m - 2 - k)] : 0.0;
      if (zk) {
        h += mulmod(e2, center(srcA[k], mod), mod) + mulmod(e1, center(srcB[k], mod), mod) + mulmod(e12, P.ztab[k], mod);
        if (k == 0) h -= e3;
      }
      dst[k] = canon(h, mod);
    }
  }
#undef RS_FRESH_LANE
}

}  // namespace rs
