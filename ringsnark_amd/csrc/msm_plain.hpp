// msm_plain.hpp -- plaintext side of the inner product: batch encoding, inverse transform, centred lift (rows a5, a6 of SURVEY.md section 8; msm.hip)
#pragma once
#include "ntt_core.hpp"
#include "ntt_wide.hpp"
#include "rs_internal.hpp"

namespace rs {

constexpr int MAX_GROUP_VECS = 4;
constexpr int MAX_GROUPS = 6;

struct TileBlockFactory {
  double *s;
  __device__ __forceinline__ LdsBlockIO operator()(int off) const { return LdsBlockIO{s + pidx(off)}; }
};

struct PlainGroup {
  const uint64_t *coeff[MAX_GROUP_VECS];
  const uint8_t *kinds[MAX_GROUP_VECS];
  unsigned *nz[MAX_GROUP_VECS];
  unsigned long long T[MAX_GROUP_VECS];
  unsigned char slot_const[MAX_GROUP_VECS];  // coeff[v] is [T][L]: one value per (term, limb), in every slot (rs_msm_vec::slot_const)
  int n;
  MsmLin lin;  // optional extra vector in linear form (count == 0: none); wide plaintext kernel only
};
struct PlainArgs {
  PlainGroup g[MAX_GROUPS];
  const uint64_t *ones_plain;  // [L][N_enc] canonical plaintext of the ring element (1, ..., 1): slot-constant vectors are value x this
};

// radix of the wave-private rounds: 3 keeps the accumulators + a round inside 128 VGPRs (no scratch)
#ifndef RS_PLAIN_MAXR
#define RS_PLAIN_MAXR 3
#endif
// grid (terms in tile, L, groups); EPT = max elements per thread (16 only for N_enc = 16384)
// M: the context's arithmetic.  C rows are the centred plaintext INTEGERS (signed doubles / int64_t), so the
// sum over a group's vectors is taken after the lift and one row serves all K data primes.
template <int EPT, int LOGN_CT = 0, class M = Mod>  // LOGN_CT != 0: transform length fixed at compile time (rounds specialised)
__global__ void __launch_bounds__(1024)
plain_center_kernel(PlainArgs args, typename ArithOf<M>::Lift *__restrict__ C, unsigned long long t0, unsigned long long tile_terms,
                    int N, int L, int logn_arg, const uint32_t *__restrict__ index_map,
                    const NttTableT<typename ArithOf<M>::T, M> *__restrict__ plain_tabs, int out_f64) {
  using T = typename ArithOf<M>::T;
  using Lift = typename ArithOf<M>::Lift;
  constexpr bool FP = std::is_same<M, Mod>::value;
  const int logn = LOGN_CT ? LOGN_CT : logn_arg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *s = reinterpret_cast<T *>(smem);
  const int n = 1 << logn;
  const unsigned long long tt = blockIdx.x, t = t0 + tt;
  const int limb = blockIdx.y, g = blockIdx.z;
  const PlainGroup &G = args.g[g];
  const NttTableT<T, M> tab = plain_tabs[limb];
  const M mod = tab.mod;
  // wave-private inverse transform when every wave gets a block of >= 256 coefficients (FP64 arithmetic)
  int logw = 0;
  while ((64 << logw) < (int)blockDim.x) logw++;
  const bool wp = FP && logn - logw >= 8 && logw >= 1 && logw <= 4;
  Lift acc[EPT];
#pragma unroll
  for (int k = 0; k < EPT; k++) acc[k] = Lift(0);
  for (int v = 0; v < G.n; v++) {
    if (t >= G.T[v]) continue;
    const int kind = G.kinds[v] ? (int)G.kinds[v][t] : RS_KIND_POLY;
    if (kind == RS_KIND_ONE) {  // Scalar 1: plaintext is the constant polynomial 1
      if (threadIdx.x == 0) {
        acc[0] += Lift(1);
        if (G.nz[v]) atomicOr(&G.nz[v][t], 1u);
      }
      continue;
    }
    if (G.slot_const[v]) {
      // a slot-constant ring element c: its batch encoding is c x encode(1, ..., 1) (the inverse transform is linear), so the
      // plaintext needs no transform -- one modular product per coefficient against the context's table; c = 0 is_zero()
      const uint64_t cres = G.coeff[v][(size_t)t * L + limb];
      if (cres == 0) continue;  // uniform over the workgroup
      if (threadIdx.x == 0 && G.nz[v]) atomicOr(&G.nz[v][t], 1u);
      const T c = center(from_res<T>(cres), mod);
      const uint64_t *ones = args.ones_plain + (size_t)limb * n;
#pragma unroll
      for (int k = 0; k < EPT; k++) {
        const int p = threadIdx.x + k * blockDim.x;
        if (p < n) acc[k] += lift_centered(canon(mulmod_dd(c, from_res<T>(ones[p]), mod), mod), mod);
      }
      continue;
    }
    if (N < n) {  // slots beyond N stay zero (seal_ring.tcc:350-351); with N == n the scatter covers the tile
      for (int p = threadIdx.x; p < n; p += blockDim.x) s[pidx(p)] = T(0);
      __syncthreads();
    }
    const uint64_t *src = G.coeff[v] + ((size_t)t * L + limb) * (size_t)N;
    bool nz = false;
    // loads in unrolled batches of 8 (a rolled loop waits for each coefficient and each map entry
    // in turn; 16 at once do not fit beside the accumulators)
    int tid = threadIdx.x;  // fresh copy per vector: keeps the 32 load addresses out of loop-invariant hoisting
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int k0 = 0; k0 < EPT; k0 += 8) {
      uint64_t val[8];
      uint32_t pos[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int x = tid + (k0 + k) * blockDim.x;
        if (x < N) {
          val[k] = src[x];
          pos[k] = index_map[x];
        }
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int x = tid + (k0 + k) * blockDim.x;
        if (x < N) {
          nz |= (val[k] != 0);
          s[pidx((int)pos[k])] = from_res<T>(val[k]);
        }
      }
    }
    if (!__syncthreads_or(nz)) continue;  // is_zero term (this limb): contributes nothing
    if (threadIdx.x == 0 && G.nz[v]) atomicOr(&G.nz[v][t], 1u);
    if constexpr (FP) {
      if (wp)
        lds_ntt_inv_wp<RS_PLAIN_MAXR, TileBlockFactory, LdsIO, 3>(s, TileBlockFactory{s}, LdsIO{s}, logn, logw, tab.d_itw, mod, tab.inv_red_mask);
      else
        lds_ntt_inv(s, logn, tab.d_itw, 1, mod, tab.inv_red_mask);
    } else {
      lds_ntt_inv(s, logn, tab.d_itw, 1, mod, tab.inv_red_mask);
    }
    tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int k = 0; k < EPT; k++) {
      const int p = tid + k * blockDim.x;
      if (p < n) {
        const T c = canon(mulmod(reduce(s[pidx(p)], mod), tab.ninv, mod), mod);
        acc[k] += lift_centered(c, mod);
      }
    }
    __syncthreads();
  }
  Lift *dst = C + (((size_t)g * tile_terms + tt) * L + limb) * (size_t)n;
#pragma unroll
  for (int k = 0; k < EPT; k++) {
    const int p = threadIdx.x + k * blockDim.x;
    if (p < n) {
      if constexpr (!FP) {
        // hybrid contexts (integer ring side, FP64 encoding side): the row goes to the FP64 multiply-accumulate as a
        // double -- exact, the host checked that the group's sum of lifts stays below 2^53
        if (out_f64) {
          const double d = (double)acc[k];
          dst[p] = (Lift)__double_as_longlong(d);
          continue;
        }
      }
      dst[p] = acc[k];
    }
  }
}

// plain_center_kernel in the wide form of ntt_wide.hpp (N_enc = 8192, FP64): 256 threads x 32 coefficients, persistent
// (a workgroup keeps one ring limb: its per-lane twiddles and the mapped scatter addresses of the batching index map
// stay in registers), two workgroups per CU.  The scatter fills the tile, the inverse transform runs in three rounds
// (4, 5, 4 stages, n^-1 folded into the last stage) and leaves thread t with coefficients 2t+c + 512 e, which are
// lifted, summed over the group's vectors and stored with 16-byte accesses.
// PAIRED rows (the layout mac_kernel_v3 reads): word 2 n' + {0, 1} = coefficient n' + {0, 4096}, n' < 4096, so the
// two operands of the forward transform's first stage arrive in one 16-byte load.
// MULTI = false (every group has one vector): no accumulators, and the next item's coefficients are requested while
// the current one is transformed.
struct PlainTwPtrs {
  const double *itw[RS_MAX_L];  // inverse twiddle tables of the ring primes (kernel-argument pointers: global loads)
};
// LIN: groups may carry a vector in linear form (MsmLin): its plaintext is accumulated from the encoded ring elements.
template <bool MULTI, bool PAIRED, int NE, bool LIN = false>  // NE = N / 512: 16-byte coefficient pairs per thread (16 at N = 8192)
__global__ void __launch_bounds__(256, 2)
plain_center_wide_kernel(PlainArgs args, double *__restrict__ C, unsigned long long t0, unsigned long long tile_terms,
                         unsigned long long tt_count, int n_groups, int N, int L, const uint32_t *__restrict__ index_map,
                         const NttTable *__restrict__ plain_tabs, PlainTwPtrs twp) {
  using S = WideShape<13>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  double *twl = s + S::TILE;
  const int t = threadIdx.x;
  const int limb = (int)(blockIdx.x % (unsigned)L), slot = (int)(blockIdx.x / (unsigned)L), nslots = (int)(gridDim.x / (unsigned)L);
  const Mod mod = plain_tabs[limb].mod;
  const double *__restrict__ itw = twp.itw[limb];
  const double ninv = uniform_f64(plain_tabs[limb].ninv);
  const uint32_t red_mask = plain_tabs[limb].inv_red_mask;
  for (int i = t; i < S::TWL; i += 256) twl[i] = itw[i];
  // round 1 (inverse stages 0..3 on 16 consecutive points): 15 twiddles per group that only this lane uses.  They are
  // re-read from the (L2-resident) table for every transform instead of living in 60 registers: the accumulators of
  // a multi-vector group and a 32-point register tile do not fit beside them.
  auto load_tw1 = [&](double (&tw1)[2][15]) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      int g = t + 256 * j;
      asm volatile("" : "+v"(g));  // opaque: the loads must not be hoisted out of the item loop (that is 60 live registers)
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const double *p = itw + (S::N >> (k + 1)) + (g << (3 - k));
        if (k == 3) {
          tw1[j][14] = p[0];
        } else {
#pragma unroll
          for (int i = 0; i < (4 >> k); i++) {
            const double2 v2 = reinterpret_cast<const double2 *>(p)[i];
            tw1[j][16 - (16 >> k) + 2 * i] = v2.x;
            tw1[j][16 - (16 >> k) + 2 * i + 1] = v2.y;
          }
        }
      }
    }
  };
  double tw3[14];  // last round, stages 9..11: block e >> (k+1) of the 8 >> k blocks of stage 9+k (uniform)
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int i = 0; i < (8 >> k); i++) tw3[16 - (16 >> k) + i] = uniform_f64(itw[(8 >> k) + i]);
  const double w_last = uniform_f64(mulmod(itw[1], ninv, mod));
  // mapped tile addresses of the slots this thread scatters to: ring slot x = 2t+c + 512 e -> px(index_map[x]), two per word
  uint32_t spos[NE];
#pragma unroll
  for (int e = 0; e < NE; e++) {
    const uint2 m2 = reinterpret_cast<const uint2 *>(index_map)[t + 256 * e];
    spos[e] = (uint32_t)S::px((int)m2.x) | ((uint32_t)S::px((int)m2.y) << 16);
  }
  __syncthreads();
  const unsigned long long items = tt_count * (unsigned long long)n_groups;
  u64x2 pre[NE];
  auto src_of = [&](unsigned long long item, int v) -> const uint64_t * {
    const int g = (int)(item % (unsigned)n_groups);
    const unsigned long long term = t0 + item / (unsigned)n_groups;
    return args.g[g].coeff[v] + ((size_t)term * L + limb) * (size_t)N;
  };
  auto issue_loads = [&](const uint64_t *src) {
    const u64x2 *s2 = reinterpret_cast<const u64x2 *>(src) + t;
#pragma unroll
    for (int e = 0; e < NE; e++) pre[e] = stream_load(s2 + 256 * e);
  };
  unsigned long long item = (unsigned long long)slot;
  __shared__ unsigned nzflag[2][4];
  int nzpar = 0;
  // MULTI == false: the single vector of every group; a term beyond its length or a constant-1 term has nothing to load
  auto loadable = [&](unsigned long long it) {
    const PlainGroup &G = args.g[it % (unsigned)n_groups];
    const unsigned long long term = t0 + it / (unsigned)n_groups;
    return term < G.T[0] && !(G.kinds[0] && G.kinds[0][term] == RS_KIND_ONE);
  };
  if (!MULTI && item < items && loadable(item)) issue_loads(src_of(item, 0));
  for (; item < items; item += (unsigned long long)nslots) {
    const int g = (int)(item % (unsigned)n_groups);
    const unsigned long long tt = item / (unsigned)n_groups, term = t0 + tt;
    const PlainGroup &G = args.g[g];
    double acc[2][16];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[c][e] = 0.0;
    const int nv = MULTI ? G.n : 1;
    // MULTI == false: the next item's coefficients are requested during this item's transform -- or right here when this
    // item has nothing to transform (vectors of different lengths, constant-1 terms): every path issues it exactly once
    auto prefetch_next = [&]() {
      if (!MULTI) {
        const unsigned long long nxt = item + (unsigned long long)nslots;
        if (nxt < items && loadable(nxt)) issue_loads(src_of(nxt, 0));
      }
    };
    for (int v = 0; v < nv; v++) {
      if (term >= G.T[v]) {
        prefetch_next();
        continue;
      }
      const int kind = G.kinds[v] ? (int)G.kinds[v][term] : RS_KIND_POLY;
      if (kind == RS_KIND_ONE) {  // Scalar 1: the plaintext is the constant polynomial 1
        if (t == 0) {
          acc[0][0] += 1.0;
          if (G.nz[v]) atomicOr(&G.nz[v][term], 1u);
        }
        prefetch_next();
        continue;
      }
      double tw1[2][15];
      if (!MULTI) load_tw1(tw1);  // before the prefetch below: waiting for them leaves the younger loads in flight
      if (MULTI) issue_loads(src_of(item, v));
      if (NE < 16) {  // slots beyond N stay zero (seal_ring.tcc:350-351)
        for (int i = t; i < S::TILE; i += 256) s[i] = 0.0;
        __syncthreads();
      }
      bool nz = false;
#pragma unroll
      for (int e = 0; e < NE; e++) {
        nz |= (pre[e].x | pre[e].y) != 0;
        s[spos[e] & 0xffffu] = from_u64(pre[e].x);
        s[spos[e] >> 16] = from_u64(pre[e].y);
      }
      mem_fence();
      if (MULTI) load_tw1(tw1);  // after the scatter: the coefficient registers are free
      prefetch_next();  // next item's coefficients: in flight during this transform
      mem_fence();
      // "is any coefficient of this limb non-zero", over the workgroup -- the barrier that also publishes the scattered tile.
      // Not __syncthreads_or: its lowering keeps the thread ids it needs in registers across the item loop, which in the LIN
      // instantiation spills two of them, and the scratch reload right here is followed by s_waitcnt vmcnt(0) -- draining the
      // next item's prefetch and the twiddle loads issued just above (round 6; found in the ISA).  One ballot per wave, one
      // flag word per wave in LDS, two alternating flag sets (an item without a non-zero coefficient skips the later barriers).
      {
        const unsigned long long bal = __ballot(nz);
        if ((t & 63) == 0) nzflag[nzpar][t >> 6] = bal != 0ull;
      }
      __syncthreads();
      const bool any = (nzflag[nzpar][0] | nzflag[nzpar][1] | nzflag[nzpar][2] | nzflag[nzpar][3]) != 0u;
      nzpar ^= 1;
      if (!any) continue;  // is_zero term (this limb): contributes nothing; nobody reads the tile
      if (t == 0 && G.nz[v]) atomicOr(&G.nz[v][term], 1u);
      // round 1: inverse stages 0..3 on 16 consecutive points
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int pb = S::px(16 * (t + 256 * j));
        double x[16];
#pragma unroll
        for (int e = 0; e < 16; e++) x[e] = s[pb + e];
        reg_inv_stages<4, true>(x, mod, red_mask, [&](int k, int i) { return tw1[j][16 - (16 >> k) + i]; });
#pragma unroll
        for (int e = 0; e < 16; e++) s[pb + e] = x[e];
      }
      __syncthreads();
      {  // round 2: inverse stages 4..8 on hi*512 + lo + 16 e
        const int lo = t & 15, hi = t >> 4;
        const int pb = hi * S::SP + lo;
        double x[32];
#pragma unroll
        for (int e = 0; e < 32; e++) x[e] = s[pb + 17 * e];
        reg_inv_stages<5, true>(x, mod, red_mask >> 4, [&](int k, int i) { return twl[(S::N >> (5 + k)) + (hi << (4 - k)) + i]; });
#pragma unroll
        for (int e = 0; e < 32; e++) s[pb + 17 * e] = x[e];
      }
      __syncthreads();
      double w[2][16];
      {  // round 3: inverse stages 9..12 on 2t+c + 512 e, the scaling folded into the last stage
        const int pb = S::px(2 * t);
#pragma unroll
        for (int e = 0; e < 16; e++) {
          w[0][e] = s[pb + S::SP * e];
          w[1][e] = s[pb + S::SP * e + 1];
        }
      }
      __syncthreads();  // the tile may be refilled
#pragma unroll
      for (int c = 0; c < 2; c++) {
        reg_inv_stages<4, true, 3>(w[c], mod, red_mask >> 9, [&](int k, int i) { return tw3[16 - (16 >> k) + i]; });
        if ((red_mask >> 12) & 1u) {
#pragma unroll
          for (int e = 0; e < 16; e++) w[c][e] = reduce(w[c][e], mod);
        }
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const double a = w[c][e], b = w[c][e + 8];
          acc[c][e] += center_balanced(mulmod(a + b, ninv, mod), mod);
          acc[c][e + 8] += center_balanced(mulmod(a - b, w_last, mod), mod);
        }
      }
    }
    if (LIN && G.lin.count && term < G.lin.T) {
      // plaintext of the linear-form vector: sum_e lv_e[term] * P_{k_e}, coefficient by coefficient (positions 2t+c + 512 e)
#pragma unroll
      for (int half = 0; half < 2; half++) {
        double a[2][8];
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
          for (int e = 0; e < 8; e++) a[c][e] = 0.0;
        for (int x = 0; x < G.lin.count; x++) {
          const double lv = center(G.lin.Lcols[((size_t)G.lin.col[x] * L + limb) * G.lin.Mlen + term], mod);
          const u64x2 *pp = reinterpret_cast<const u64x2 *>(G.lin.P + ((size_t)G.lin.k[x] * L + limb) * (size_t)S::N) + t;
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const u64x2 pv = pp[256 * (8 * half + e)];
            a[0][e] += mulmod(from_u64(pv.x), lv, mod);
            a[1][e] += mulmod(from_u64(pv.y), lv, mod);
          }
          if ((x & 3) == 3) {
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
              for (int e = 0; e < 8; e++) a[c][e] = reduce(a[c][e], mod);
          }
        }
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
          for (int e = 0; e < 8; e++) acc[c][8 * half + e] += center_balanced(reduce(a[c][e], mod), mod);
      }
    }
    double *dst = C + (((size_t)g * tile_terms + tt) * L + limb) * (size_t)S::N;
    double2 *d2 = reinterpret_cast<double2 *>(dst);
    if (PAIRED) {
#pragma unroll
      for (int e = 0; e < 8; e++)
#pragma unroll
        for (int c = 0; c < 2; c++) d2[2 * t + c + 512 * e] = make_double2(acc[c][e], acc[c][e + 8]);
    } else {
#pragma unroll
      for (int e = 0; e < 16; e++) d2[t + 256 * e] = make_double2(acc[0][e], acc[1][e]);
    }
  }
}

// a5: BatchEncoder::encode -> canonical coefficient-form plaintext.  grid (count, L)
template <class M>
__global__ void __launch_bounds__(1024)
batch_encode_kernel(const uint64_t *__restrict__ rings, uint64_t *__restrict__ plain, int N, int L, int logn,
                    const uint32_t *__restrict__ index_map, const NttTableT<typename ArithOf<M>::T, M> *__restrict__ plain_tabs) {
  using T = typename ArithOf<M>::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *s = reinterpret_cast<T *>(smem);
  const int n = 1 << logn;
  const size_t k = blockIdx.x;
  const int limb = blockIdx.y;
  const NttTableT<T, M> tab = plain_tabs[limb];
  const M mod = tab.mod;
  for (int p = threadIdx.x; p < n; p += blockDim.x) s[pidx(p)] = T(0);
  __syncthreads();
  const uint64_t *src = rings + (k * L + limb) * (size_t)N;
  for (int x = threadIdx.x; x < N; x += blockDim.x) s[pidx((int)index_map[x])] = from_res<T>(src[x]);
  __syncthreads();
  lds_ntt_inv(s, logn, tab.d_itw, 1, mod, tab.inv_red_mask);
  uint64_t *dst = plain + (k * L + limb) * (size_t)n;
  for (int p = threadIdx.x; p < n; p += blockDim.x)
    dst[p] = to_res(canon(mulmod(reduce(s[pidx(p)], mod), tab.ninv, mod), mod));
}

}  // namespace rs
