// msm.hip -- EncodingElem pieces (rows a5-a8) and the encoding inner product / ring-MSM (row a9)
// of SURVEY.md section 8, restructured for the GPU:
//
//   reference (seal_ring.tcc:415-431, per term and ring limb i):
//       plain = iNTT_{q_i}(scatter(b_t[i]))                 BatchEncoder::encode      (:534)
//       for j < K:  P_j = NTT_{Q_j}(centred_lift(plain))    multiply_plain_inplace    (:536)
//                   tmp[c][j] = ct_t[i][c][j] * P_j
//       res += tmp                                           add_inplace              (:494)
//
//   here, three kernels over a tile of terms:
//     plain_center_kernel  one workgroup per (group, term, limb): scatter + inverse NTT mod q_i in
//                          LDS, centred lift, SUM over the coefficient vectors of the group ->
//                          C[g][t][i][N_enc] (signed doubles).  Summing after the lift is exact:
//                          ct*(P1) + ct*(P2) = ct*NTT(lift(p1)+lift(p2)) in Z_{Q_j}.
//     mac_kernel           one workgroup per (limb, prime j, term chunk): C mod Q_j -> forward NTT
//                          in LDS -> multiply with both ciphertext polynomials streamed from HBM,
//                          accumulate in registers across the chunk's terms (lazy reduction).
//                          Every ciphertext word is read exactly once per pass.
//     reduce_kernel        sums the per-chunk partial accumulators (+ optional addends).
//
// Skipped (is_zero) terms contribute a zero plaintext, which is the identity for the sum, so the
// value-level result equals the reference's; the EMPTY result (all terms skipped) is reported
// through the used-term counts.
#include <algorithm>
#include <cstring>
#include <type_traits>

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
  int n;
  MsmLin lin;  // optional extra vector in linear form (count == 0: none); wide plaintext kernel only
};
struct PlainArgs {
  PlainGroup g[MAX_GROUPS];
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
      const bool any = __syncthreads_or(nz);
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
          acc[c][e] += lift_centered(canon(mulmod(a + b, ninv, mod), mod), mod);
          acc[c][e + 8] += lift_centered(canon(mulmod(a - b, w_last, mod), mod), mod);
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
          for (int e = 0; e < 8; e++) acc[c][8 * half + e] += lift_centered(canon(a[c][e], mod), mod);
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

template <class L_>
struct alignas(16) LiftPair {
  L_ x, y;
};
struct MacArgs {
  const void *C[2];       // per group: [tile_terms][L][n] centred plaintext integers (double or int64_t)
  const uint64_t *crs[2]; // per CRS vector: element 0 of the tile, [terms][L][2][K][n]
  uint64_t *partial;      // [n_chunks][n_sets_total][L][2][K][n]
  int set_index[4];       // which set slot (c * n_groups + g) each accumulator set writes
  int n_sets_total;
  unsigned long long terms[2];  // per group: number of valid terms in this tile
  unsigned long long tile_terms;
  int terms_per_chunk, n_chunks;
  int accumulate;       // 1: add onto the existing partial slot
  int acc_period;       // terms between lazy reductions of the accumulators
  int reduce_u;         // 1: bring NTT outputs back to |u| <= p/2 before the MAC (large primes)
};

// The dominant kernel.  Accumulator set (c, g): sum_t crs[c][t] * NTT(C[g][t]).
template <int NG, int NC, int PAIRS, class M = Mod>
__global__ void __launch_bounds__(1024)
mac_kernel(MacArgs a, int L, int K, int logn, const NttTableT<typename ArithOf<M>::T, M> *__restrict__ coeff_tabs) {
  using T = typename ArithOf<M>::T;
  using Lift = typename ArithOf<M>::Lift;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *s = reinterpret_cast<T *>(smem);
  const int n = 1 << logn;
  // XCD-aware mapping: blocks b, b+8, b+16, ... (same XCD, dispatched back to back) take the K
  // primes of one (limb, chunk), so the K readers of a C row share it through that XCD's L2.
  const unsigned b = blockIdx.x;
  const int j = (int)((b >> 3) % (unsigned)K);
  const unsigned r = (b & 7u) + 8u * (b / (8u * (unsigned)K));
  if (r >= (unsigned)(a.n_chunks * L)) return;
  const int limb = (int)(r % (unsigned)L), chunk = (int)(r / (unsigned)L);
  const NttTableT<T, M> tab = coeff_tabs[j];
  const M mod = tab.mod;
  constexpr int NS = NG * NC;
  // PAIRS = n / (2 * blockDim): 4, or 8 for N_enc = 16384
  T acc[NS][2][2 * PAIRS];
  const size_t enc_words = (size_t)L * 2 * K * n;
#pragma unroll
  for (int st = 0; st < NS; st++)
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int k = 0; k < PAIRS; k++) {
        acc[st][c][2 * k] = acc[st][c][2 * k + 1] = T(0);
        const int pp = threadIdx.x + k * blockDim.x;
        if (a.accumulate && pp < (n >> 1)) {
          const uint64_t *pv = a.partial + ((size_t)chunk * a.n_sets_total + a.set_index[st]) * enc_words +
                               (((size_t)limb * 2 + c) * K + j) * (size_t)n;
          const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(pv)[pp];
          acc[st][c][2 * k] = from_res<T>(v.x);
          acc[st][c][2 * k + 1] = from_res<T>(v.y);
        }
      }
  const unsigned long long tbeg = (unsigned long long)chunk * a.terms_per_chunk;
  const unsigned long long tend = min(tbeg + (unsigned long long)a.terms_per_chunk, a.tile_terms);
  int since = 0;
  for (unsigned long long t = tbeg; t < tend; t++) {
#pragma unroll
    for (int g = 0; g < NG; g++) {
      if (t >= a.terms[g]) continue;
      // centred plaintext (|c| < members * q_i / 2) -> residues mod Q_j.  Integer arithmetic: the residue goes
      // in as c*R, the transform is linear, so the spectrum comes out in Montgomery form and ct * spectrum below
      // is ONE reduction per product.
      const LiftPair<Lift> *src = reinterpret_cast<const LiftPair<Lift> *>(static_cast<const Lift *>(a.C[g]) + ((size_t)t * L + limb) * (size_t)n);
      for (int pp = threadIdx.x; pp < (n >> 1); pp += blockDim.x) {
        const LiftPair<Lift> v = src[pp];
        const int pi = pidx(2 * pp);
        s[pi] = to_mont(lift_residue(v.x, mod), mod);
        s[pnext(pi)] = to_mont(lift_residue(v.y, mod), mod);
      }
      __syncthreads();
      lds_ntt_fwd(s, logn, tab.d_tw, 1, mod, tab.fwd_red_mask);
#pragma unroll
      for (int k = 0; k < PAIRS; k++) {
        const int pp = threadIdx.x + k * blockDim.x;
        if (pp < (n >> 1)) {
          const int pi = pidx(2 * pp);
          T u0 = s[pi], u1 = s[pnext(pi)];
          if (a.reduce_u) {
            u0 = reduce(u0, mod);
            u1 = reduce(u1, mod);
          }
#pragma unroll
          for (int cc = 0; cc < NC; cc++) {
#pragma unroll
            for (int c = 0; c < 2; c++) {
              const uint64_t *ct = a.crs[cc] + (size_t)t * enc_words + (((size_t)limb * 2 + c) * K + j) * (size_t)n;
              const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(ct)[pp];
              const int st = cc * NG + g;
              acc[st][c][2 * k] = addm(acc[st][c][2 * k], mulmod(from_res<T>(v.x), u0, mod), mod);
              acc[st][c][2 * k + 1] = addm(acc[st][c][2 * k + 1], mulmod(from_res<T>(v.y), u1, mod), mod);
            }
          }
        }
      }
      __syncthreads();
    }
    if (++since >= a.acc_period) {
      since = 0;
#pragma unroll
      for (int st = 0; st < NS; st++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
          for (int k = 0; k < 2 * PAIRS; k++) acc[st][c][k] = reduce(acc[st][c][k], mod);
    }
  }
#pragma unroll
  for (int st = 0; st < NS; st++)
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int k = 0; k < PAIRS; k++) {
        const int pp = threadIdx.x + k * blockDim.x;
        if (pp < (n >> 1)) {
          uint64_t *pv = a.partial + ((size_t)chunk * a.n_sets_total + a.set_index[st]) * enc_words +
                         (((size_t)limb * 2 + c) * K + j) * (size_t)n;
          ulonglong2 o;
          o.x = to_res(canon(acc[st][c][2 * k], mod));
          o.y = to_res(canon(acc[st][c][2 * k + 1], mod));
          reinterpret_cast<ulonglong2 *>(pv)[pp] = o;
        }
      }
}

// mac_kernel_v2: the streaming form of the dominant kernel (one accumulator set per launch,
// 2048 <= N_enc <= 8192).  Differences from mac_kernel:
//   * the twiddle table of Q_j lives in LDS next to the tile (68 + 64 KiB), so inside the term loop
//     the ONLY vector-memory operations are the streamed operands;
//   * the ciphertext words of term t are loaded into registers before the transform of term t
//     starts (512-thread shape: also the plaintext row of term t+1), so HBM latency and transfer
//     overlap the FP64 work -- provided nothing in the loop forces an early s_waitcnt vmcnt(0);
//   * the transform is the wave-private form (ntt_core.hpp): the cross-wave stages, then each wave
//     finishes its own block and multiplies exactly that block into its accumulators -- two or
//     three workgroup barriers per term instead of seven.
// radix of the wave-private rounds inside mac_kernel_v2: 3 keeps the kernel free of VGPR spills (a
// scratch reload inside the term loop costs an s_waitcnt vmcnt(0), which drains the prefetched
// ciphertext loads and serialises stream and transform)
#ifndef RS_MAC_MAXR
#define RS_MAC_MAXR 3
#endif
struct MacArgs2 {
  const double *C;      // [tile_terms][L][n] plaintext rows of the group
  const uint64_t *crs;  // first ciphertext of the tile
  uint64_t *partial;    // accumulator slot 0 of this set: [n_chunks] stride part_stride
  size_t part_stride;   // words between consecutive chunks
  unsigned long long terms;  // valid terms in this tile
  int terms_per_chunk, n_chunks;
  int accumulate, acc_period, reduce_u;
};
// THREADS = 1024 (default at N_enc = 8192): 16 waves, 118 VGPRs, four waves per SIMD, the plaintext
// row loaded where it is used.  THREADS = 512: 8 waves, ~240 VGPRs, the plaintext row of term t+1
// prefetched as well.  Both use radix-8 private rounds (RS_MAC_MAXR) to stay free of scratch.
// LOGN_CT != 0: transform length fixed at compile time (rounds specialised).  ABLATE (experiments,
// tools/mac_ablate.py): 1 = skip the transform, 2 = skip the ciphertext loads, 4 = skip the C loads;
// a compile-time parameter because a run-time branch around each load makes the compiler wait for
// every load right where it is issued.
template <int THREADS, int LOGN_CT = 0, int ABLATE = 0>
__global__ void __launch_bounds__(THREADS)
mac_kernel_v2(MacArgs2 a, int L, int K, int logn_arg, const NttTable *__restrict__ coeff_tabs) {
  const int logn = LOGN_CT ? LOGN_CT : logn_arg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  constexpr int LOGW = THREADS == 512 ? 3 : 4;
  constexpr int PP = 8192 / (2 * THREADS);  // coefficient pairs per lane at n = 8192 (fewer for smaller n)
  const int n = 1 << logn;
  double *twl = s + padded_len((size_t)n);
  const unsigned b = blockIdx.x;
  const int j = (int)((b >> 3) % (unsigned)K);
  const unsigned r = (b & 7u) + 8u * (b / (8u * (unsigned)K));
  if (r >= (unsigned)(a.n_chunks * L)) return;
  const int limb = (int)(r % (unsigned)L), chunk = (int)(r / (unsigned)L);
  const Mod mod = coeff_tabs[j].mod;
  const uint32_t red_mask = coeff_tabs[j].fwd_red_mask;
  {
    const double *gtw = coeff_tabs[j].d_tw;
    for (int i = threadIdx.x; i < n; i += THREADS) twl[i] = gtw[i];
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int bpairs = n >> (LOGW + 1);      // pairs per wave block
  const int pbase = wave * bpairs + lane;  // + 64*k, k < PP
  const size_t enc_words = (size_t)L * 2 * K * n;
  const size_t slab = (((size_t)limb * 2) * K + j) * (size_t)n;  // component 0; component 1 is + K*n
  const size_t comp = (size_t)K * n;
  uint64_t *part = a.partial + (size_t)chunk * a.part_stride + slab;
  double acc[2][2 * PP];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int k = 0; k < PP; k++) {
      acc[c][2 * k] = acc[c][2 * k + 1] = 0.0;
      if (a.accumulate && 64 * k + lane < bpairs) {
        const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(part + c * comp)[pbase + 64 * k];
        acc[c][2 * k] = from_u64(v.x);
        acc[c][2 * k + 1] = from_u64(v.y);
      }
    }
  const unsigned long long tbeg = (unsigned long long)chunk * a.terms_per_chunk;
  const unsigned long long tend = min(tbeg + (unsigned long long)a.terms_per_chunk, a.terms);
  const TileBlockFactory bf{s};
  const LdsIO lds{s};
  int since = 0;
  const double2 *crow = reinterpret_cast<const double2 *>(a.C + ((size_t)tbeg * L + limb) * (size_t)n);
  const uint64_t *ctp = a.crs + (size_t)tbeg * enc_words + slab;
  // 512 threads: the next plaintext row is prefetched across the transform; 1024 threads (half the
  // registers per lane, twice the waves to hide the L2 latency): loaded where it is used
  constexpr bool PREFETCH_C = THREADS == 512;
  double2 cn[PP];
  if (PREFETCH_C && tbeg < tend) {
#pragma unroll
    for (int k = 0; k < PP; k++)
      if (64 * k + lane < bpairs) cn[k] = crow[pbase + 64 * k];
  }
  for (unsigned long long t = tbeg; t < tend; t++) {
    const int pbl = pbase;
    if (!PREFETCH_C) {
#pragma unroll
      for (int k = 0; k < PP; k++)
        if (64 * k + lane < bpairs) cn[k] = crow[pbase + 64 * k];
    }
    // plaintext row -> tile, reduced mod Q_j
#pragma unroll
    for (int k = 0; k < PP; k++)
      if (64 * k + lane < bpairs) {
        const int pi = pidx(2 * (pbl + 64 * k));
        s[pi] = reduce(cn[k].x, mod);
        s[pnext(pi)] = reduce(cn[k].y, mod);
      }
    __syncthreads();
    // stream: this term's ciphertext words and the next plaintext row, in flight during the transform
    ulonglong2 ct[2][PP];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int k = 0; k < PP; k++)
        if (64 * k + lane < bpairs) {
          if (ABLATE & 2)
            ct[c][k] = make_ulonglong2(12345ull + k, 6789ull + c);
          else
            ct[c][k] = reinterpret_cast<const ulonglong2 *>(ctp + c * comp)[pbase + 64 * k];
        }
    crow += (size_t)L * (n >> 1);
    ctp += enc_words;
    if (PREFETCH_C && t + 1 < tend && !(ABLATE & 4)) {
#pragma unroll
      for (int k = 0; k < PP; k++)
        if (64 * k + lane < bpairs) cn[k] = crow[pbase + 64 * k];
    }
    // the loads above must be ISSUED before the transform (the scheduler would otherwise sink them
    // next to their uses, after the transform, to save registers)
    __builtin_amdgcn_sched_barrier(0);
    if (!(ABLATE & 1)) lds_ntt_fwd_wp<RS_MAC_MAXR, LdsIO, TileBlockFactory, 3, true>(s, lds, bf, logn, LOGW, twl, mod, red_mask);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < PP; k++)
      if (64 * k + lane < bpairs) {
        const int pi = pidx(2 * (pbl + 64 * k));
        double u0 = s[pi], u1 = s[pnext(pi)];
        if (a.reduce_u) {
          u0 = reduce(u0, mod);
          u1 = reduce(u1, mod);
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
          acc[c][2 * k] += mulmod(from_u64(ct[c][k].x), u0, mod);
          acc[c][2 * k + 1] += mulmod(from_u64(ct[c][k].y), u1, mod);
        }
      }
    if (++since >= a.acc_period) {
      since = 0;
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int k = 0; k < 2 * PP; k++) acc[c][k] = reduce(acc[c][k], mod);
    }
    wave_sync();  // this wave's block may now be overwritten by the next row (wave-private region)
  }
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int k = 0; k < PP; k++)
      if (64 * k + lane < bpairs) {
        ulonglong2 o;
        o.x = to_u64(canon(acc[c][2 * k], mod));
        o.y = to_u64(canon(acc[c][2 * k + 1], mod));
        reinterpret_cast<ulonglong2 *>(part + c * comp)[pbase + 64 * k] = o;
      }
}

// mac_kernel_v3 (N_enc = 8192, FP64): the wide form of ntt_wide.hpp applied to the inner product.
// A polynomial's accumulators (2 components x 8192) do not fit beside a 32-coefficient-per-thread transform, so a
// workgroup of 256 threads owns HALF of the spectrum of one (limb, prime): the first stage (gap 4096) is applied
// while the plaintext row is loaded -- half h keeps x[n] + w*x[n+4096] (h = 0) or x[n] - w*x[n+4096] (h = 1) -- and
// the remaining 12 stages are the 4096-point sub-transform rooted at node 2 + h, in three radix-16 rounds with 16
// coefficients per thread: two tile exchanges per term, the lane's own (round-3) twiddles in registers for the whole chunk.
// The spectrum half is exactly the contiguous half [4096 h, 4096 h + 4096) of both ciphertext components, which the
// workgroup streams with 16-byte loads issued before the transform; a wave-private pass turns round 3's 16
// consecutive points per thread into the lane-contiguous layout of those loads.  The plaintext row is read by both
// halves (from L2: it is shared by the 2 K workgroups of a (term, limb)) and stage 0's multiply is done twice:
// +8 % FP64 work against five fewer LDS passes per term and two independent workgroups per CU.
// Up to two groups that multiply the SAME key vector (A and B of groth16.tcc:89-103 against s_pows) run in one
// launch as neighbouring workgroups of one XCD, so the second read of a ciphertext word is served on-die.
// 1: the next plaintext row is requested before the multiply-accumulate (64 more live registers: spills at 256)
#ifndef RS_MAC3_ABLATE
#define RS_MAC3_ABLATE 0
#endif
struct MacArgs3 {
  const double *C[2];        // [tile_terms][L][n] plaintext rows per group
  uint64_t *partial[2];      // accumulator set per group: [n_chunks] stride part_stride
  unsigned long long terms[2];
  const uint64_t *crs;       // first ciphertext of the tile
  size_t part_stride;
  int n_groups, terms_per_chunk, n_chunks, accumulate, acc_period, reduce_u;
  int paired;                // rows in plain_center_wide_kernel's paired layout
  uint32_t red_mask[RS_MAX_K];  // bit s: reduce before stage s of the forward transform mod Q_j (start bound = max |C|)
};
// LOGN = 14 (N_enc = 16384: the shapes of BASELINE configs[3] / [4] and of the reference's microbench.cpp:13-14): a
// workgroup owns a QUARTER of the spectrum -- the first TWO stages (gaps 8192 and 4096) are applied while the row is
// loaded, the other 12 are the 4096-point sub-transform rooted at node 4 + quarter; everything after the load is the
// LOGN = 13 kernel.  Rows are in natural order (PAIRED = false).
template <bool PAIRED, int LOGN = 13>
__global__ void __launch_bounds__(256, 2)
mac_kernel_v3(MacArgs3 a, int L, int K, const NttTable *__restrict__ coeff_tabs) {
  constexpr int n = 1 << LOGN, H = 4096, LOGP = LOGN - 12, PARTS = 1 << LOGP;
  static_assert(LOGN == 13 || (LOGN == 14 && !PAIRED), "half spectrum at 8192 points, quarter spectrum at 16384");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  // block -> XCD slot x (blocks go to XCDs round-robin) and a position q in that XCD's sequence.  The 2K workgroups
  // that read one plaintext row (both halves of every prime) and the groups that read the same ciphertext words are
  // consecutive in ONE XCD's sequence: one of them fetches from memory, the others hit that XCD's L2.
  const unsigned b = blockIdx.x, x = b & 7u;
  unsigned q = b >> 3;
  const int g = (int)(q % (unsigned)a.n_groups);
  q /= (unsigned)a.n_groups;
  const unsigned hj = q % ((unsigned)PARTS * (unsigned)K);
  const unsigned rr = (q / ((unsigned)PARTS * (unsigned)K)) * 8u + x;  // (chunk, limb)
  const int h = (int)(hj & (unsigned)(PARTS - 1)), j = (int)(hj >> LOGP);
  const int limb = (int)(rr % (unsigned)L), chunk = (int)(rr / (unsigned)L);
  if (chunk >= a.n_chunks) return;
  const Mod mod = coeff_tabs[j].mod;
  const double *__restrict__ tw = coeff_tabs[j].d_tw;
  const uint32_t red_mask = a.red_mask[j];
  const int root = PARTS + h;
  // per-lane twiddles of rounds 2 and 3, fixed for the whole chunk
  const int lo = t & 15, hi = t >> 4;
  // round-2 twiddles tw[(root << (4+k)) + (hi << k) + b] (16 lanes share each) come from an LDS copy of the table's
  // first 1024 entries; the round-3 twiddles are the lane's own and stay in registers
  double *twl = s + 2 * (H + H / 16);
  for (int i = t; i < 1024; i += 256) twl[i] = tw[i];
  __syncthreads();
  double tw3[15];
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int bk = 0; bk < (1 << k); bk++) tw3[(1 << k) - 1 + bk] = tw[(root << (8 + k)) + (t << k) + bk];
#pragma unroll
  for (int i = 0; i < 15; i++) pin(tw3[i]);
  // wave-uniform twiddles of stage 0 and round 1, as scalar registers: fetched through the table pointer inside the
  // term loop they would be vector loads, and waiting for the youngest vector load drains the ciphertext stream
  const double w0 = uniform_f64(tw[1]);
  const double w1 = uniform_f64(tw[2 + (h >> 1)]);  // LOGN = 14: stage 1 of this quarter's half
  double tw1[15];
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int bk = 0; bk < (1 << k); bk++) tw1[(1 << k) - 1 + bk] = uniform_f64(tw[(root << k) + bk]);
  const size_t enc_words = (size_t)L * 2 * K * n;
  const size_t slab = (((size_t)limb * 2) * K + j) * (size_t)n + (size_t)h * H;  // component 0; component 1 is + K*n
  const size_t comp = (size_t)K * n;
  uint64_t *part = a.partial[g] + (size_t)chunk * a.part_stride + slab;
  const int r0 = wave * 1024;  // the wave's range of the half spectrum: 64 round-3 groups
  double acc[2][16];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int i = 0; i < 8; i++) {
      acc[c][2 * i] = acc[c][2 * i + 1] = 0.0;
      if (a.accumulate) {
        const u64x2 v = reinterpret_cast<const u64x2 *>(part + c * comp + r0)[lane + 64 * i];
        acc[c][2 * i] = from_u64(v.x);
        acc[c][2 * i + 1] = from_u64(v.y);
      }
    }
  const unsigned long long tbeg = (unsigned long long)chunk * a.terms_per_chunk;
  const unsigned long long tend = min(tbeg + (unsigned long long)a.terms_per_chunk, a.terms[g]);
  const double *crow = a.C[g] + ((size_t)tbeg * L + limb) * (size_t)n + (PAIRED ? 2 * t : t);
  const uint64_t *ctp = a.crs + (size_t)tbeg * enc_words + slab + r0;
  double cl[16], ch[16];
  // The row in two batches of 32 registers, the first requested before the previous term's multiply-accumulate and
  // the second after it (all 64 at once do not fit beside it).  Paired rows (plain_center_wide_kernel): one 16-byte
  // load brings x[n'] and x[n' + 4096]; plain rows: the multiplied operands x[n' + 4096] first.
  auto load_row_a = [&]() {
#pragma unroll
    for (int e = 0; e < 16; e++) {
#if RS_MAC3_ABLATE & 1  // experiment: no plaintext-row traffic (wrong results)
      if (PAIRED ? e < 8 : true) ch[e] = 5.0 + t;
      if (PAIRED && e < 8) cl[e] = 3.0 + e;
#else
      if (PAIRED) {
        if (e < 8) {
          const double2 x2 = reinterpret_cast<const double2 *>(crow)[256 * e];
          cl[e] = x2.x;
          ch[e] = x2.y;
        }
      } else {
        ch[e] = crow[256 * e + H];
      }
#endif
    }
  };
  auto load_row_b = [&]() {
#pragma unroll
    for (int e = 0; e < 16; e++) {
#if RS_MAC3_ABLATE & 1
      if (PAIRED ? e >= 8 : true) cl[e] = 3.0 + e;
      if (PAIRED && e >= 8) ch[e] = 5.0 + t;
#else
      if (PAIRED) {
        if (e >= 8) {
          const double2 x2 = reinterpret_cast<const double2 *>(crow)[256 * e];
          cl[e] = x2.x;
          ch[e] = x2.y;
        }
      } else {
        cl[e] = crow[256 * e];
      }
#endif
    }
  };
  // Software pipeline over the terms of the chunk, one term deep: the spectrum of term t is parked in tile t % 2 and
  // multiplied into the accumulators during iteration t + 1, AFTER that iteration's plaintext-row loads have been
  // issued and BEFORE its transform, so the ciphertext loads of term t (issued before the transform of term t) have a
  // whole term to land and the row loads of term t + 1 fly during the multiply-accumulate.
  u64x2 ct[2][8];
  auto issue_ct = [&]() {
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
#if RS_MAC3_ABLATE & 2  // experiment: no ciphertext traffic (wrong results)
        ct[c][i] = u64x2{12345ull + i, 6789ull + c};
#else
        ct[c][i] = stream_load(reinterpret_cast<const u64x2 *>(ctp + c * comp) + lane + 64 * i);
#endif
      }
    ctp += enc_words;
  };
  int since = 0;
  auto mac = [&](const double *tile) {
    const int p0 = r0 + (r0 >> 4) + 2 * lane + (lane >> 3);  // px(r0 + 2 lane)
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const double u0 = tile[p0 + 136 * i], u1 = tile[p0 + 136 * i + 1];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        acc[c][2 * i] += mulmod(from_u64(ct[c][i].x), u0, mod);
        acc[c][2 * i + 1] += mulmod(from_u64(ct[c][i].y), u1, mod);
      }
    }
    if (++since >= a.acc_period) {
      since = 0;
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[c][i] = reduce(acc[c][i], mod);
    }
  };
  constexpr int TILE = H + H / 16;
  for (unsigned long long tt = tbeg; tt < tend; tt++) {
    double *tile = s + (int)((tt - tbeg) & 1) * TILE;
    double v[16];
    if (LOGN == 13) {
      load_row_a();
      mem_fence();
      if (tt > tbeg) mac(s + (int)((tt - tbeg + 1) & 1) * TILE);  // term tt - 1
      mem_fence();
      load_row_b();
      crow += (size_t)L * n;
      mem_fence();
      // stage 0 (gap 4096): this half's operand of the 4096-point sub-transform
#pragma unroll
      for (int e = 0; e < 16; e++) {
        double bq = ch[e];
        if (red_mask & 1u) bq = reduce(bq, mod);
        ch[e] = mulmod(bq, w0, mod);
        pin(ch[e]);
      }
#pragma unroll
      for (int e = 0; e < 16; e++) {
        double aq = cl[e];
        if (red_mask & 1u) aq = reduce(aq, mod);
        v[e] = h ? aq - ch[e] : aq + ch[e];
        pin(v[e]);
      }
    } else {
      // stages 0 (gap 8192) and 1 (gap 4096) on x[n'], x[n' + 4096], x[n' + 8192], x[n' + 12288], n' = t + 256 e: the two
      // operands multiplied by stage 0's twiddle are requested before the previous term's multiply-accumulate, the
      // other two after it, one at a time (all 64 words at once do not fit beside the accumulators)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        cl[e] = crow[256 * e + 2 * H];
        ch[e] = crow[256 * e + 3 * H];
      }
      mem_fence();
      if (tt > tbeg) mac(s + (int)((tt - tbeg + 1) & 1) * TILE);  // term tt - 1
      mem_fence();
#pragma unroll
      for (int e = 0; e < 16; e++) v[e] = crow[256 * e];
      mem_fence();
#pragma unroll
      for (int e = 0; e < 16; e++) {
        double x2 = cl[e], x3 = ch[e];
        if (red_mask & 1u) {
          x2 = reduce(x2, mod);
          x3 = reduce(x3, mod);
        }
        cl[e] = mulmod(x2, w0, mod);
        ch[e] = mulmod(x3, w0, mod);
        pin(cl[e]);
        pin(ch[e]);
      }
#pragma unroll
      for (int e = 0; e < 16; e++) {  // u0 = x0 +- w0 x2 (this quarter's half of stage 0)
        double x0 = v[e];
        if (red_mask & 1u) x0 = reduce(x0, mod);
        cl[e] = (h & 2) ? x0 - cl[e] : x0 + cl[e];
        pin(cl[e]);
      }
      mem_fence();
#pragma unroll
      for (int e = 0; e < 16; e++) v[e] = crow[256 * e + H];
      crow += (size_t)L * n;
      mem_fence();
#pragma unroll
      for (int e = 0; e < 16; e++) {  // u1 = x1 +- w0 x3, then stage 1: v = u0 +- w1 u1
        double x1 = v[e];
        if (red_mask & 1u) x1 = reduce(x1, mod);
        double u1 = (h & 2) ? x1 - ch[e] : x1 + ch[e];
        if (red_mask & 2u) {
          u1 = reduce(u1, mod);
          cl[e] = reduce(cl[e], mod);
        }
        u1 = mulmod(u1, w1, mod);
        v[e] = (h & 1) ? cl[e] - u1 : cl[e] + u1;
        pin(v[e]);
      }
    }
    mem_fence();
    issue_ct();  // after the row registers are dead: the two never overlap
    mem_fence();
#if RS_MAC3_ABLATE & 4  // experiment: no transform (wrong results)
    tile[17 * t] = v[0] + v[5] + v[9] + v[15];
    wave_sync();
    continue;
#endif
    // round 1: stages LOGP..LOGP+3 on elements t + 256 e (uniform twiddles)
    reg_fwd_stages<4, true>(v, mod, red_mask >> LOGP, [&](int k, int bk) { return tw1[(1 << k) - 1 + bk]; });
    {  // tile tt % 2 was last read by the multiply-accumulate of term tt - 2, two barriers ago
      const int pb = t + (t >> 4);
#pragma unroll
      for (int e = 0; e < 16; e++) tile[pb + 272 * e] = v[e];
    }
    __syncthreads();
    {  // round 2: stages 5..8 on hi*256 + lo + 16 e
      const int pb = hi * 272 + lo;
#pragma unroll
      for (int e = 0; e < 16; e++) v[e] = tile[pb + 17 * e];
      reg_fwd_stages<4, true>(v, mod, red_mask >> (LOGP + 4), [&](int k, int bk) { return twl[(root << (4 + k)) + (hi << k) + bk]; });
#pragma unroll
      for (int e = 0; e < 16; e++) tile[pb + 17 * e] = v[e];
    }
    wave_sync();  // a round-2 group (256 elements) is 16 consecutive threads, who also own it in round 3: no workgroup barrier
    {  // round 3: stages 9..12 on 16 consecutive points, parked for the wave-private transposition
      const int pb = 17 * t;
#pragma unroll
      for (int e = 0; e < 16; e++) v[e] = tile[pb + e];
      reg_fwd_stages<4, true>(v, mod, red_mask >> (LOGP + 8), [&](int k, int bk) { return tw3[(1 << k) - 1 + bk]; });
#pragma unroll
      for (int e = 0; e < 16; e++) tile[pb + e] = a.reduce_u ? reduce(v[e], mod) : v[e];
    }
    wave_sync();
  }
  if (tend > tbeg) mac(s + (int)((tend - tbeg + 1) & 1) * TILE);  // the last term
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int i = 0; i < 8; i++) {
      u64x2 o;
      o.x = to_u64(canon(acc[c][2 * i], mod));
      o.y = to_u64(canon(acc[c][2 * i + 1], mod));
      reinterpret_cast<u64x2 *>(part + c * comp + r0)[lane + 64 * i] = o;
    }
}

// mac_kernel_v4 (N_enc = 16384 -- described here -- and 8192; FP64, TWO key vectors): Rinocchio multiplies every coefficient vector into BOTH s_pows and
// alpha_s_pows (rinocchio.tcc:106-160: a, b, c, h, z against each), so the plaintext spectrum -- 82 % of mac_kernel_v3's
// arithmetic -- is wanted twice.  Four accumulator sets (2 keys x 2 components) of a spectrum quarter do not fit the
// registers of 256 threads next to a 16-coefficient-per-thread transform; here 512 threads (one workgroup per CU, the same
// 8 waves) own the quarter with EIGHT points each: the accumulators of both keys take the registers one key took, the
// ciphertext streams of both keys the registers one stream took, and the transform runs once per term.
//   stages 0, 1       while the row is loaded (as mac_kernel_v3<false, 14>), on elements t + 512 e
//   round 1           sub-stages 0..2 on t + 512 e           wave-uniform twiddles        -> tile, WORKGROUP barrier
//   round 2           sub-stages 3..5 on 512 w + lane + 64 e  wave w's own 512 elements:
//   round 3           sub-stages 6..8 on 64 (t/8) + t%8 + 8 e   every later exchange is wave-private
//   round 4           sub-stages 9..11 on 8 t + e             the lane's own twiddles, in registers for the whole chunk
// One workgroup barrier per term; the spectrum of term t is multiplied into the accumulators during iteration t + 1 (tiles
// alternate), under the row loads of that term.  Tile position of element i: i + i/8.
// Up to RS_MAC4_GROUPS coefficient vectors (groups) run in one launch as neighbouring workgroups of one XCD, so that one of
// them fetches a ciphertext word from memory and the others find it in that XCD's L2.
constexpr int RS_MAC4_GROUPS = 6;
struct MacArgs4 {
  const double *C[RS_MAC4_GROUPS];        // [tile_terms][L][n] plaintext rows per group
  uint64_t *partial[2][RS_MAC4_GROUPS];   // accumulator set per (key, group): [n_chunks] stride part_stride
  unsigned long long terms[RS_MAC4_GROUPS];
  const uint64_t *crs[2];                 // first ciphertext of the tile, per key vector
  size_t part_stride;
  int n_groups, terms_per_chunk, n_chunks, accumulate, acc_period, reduce_u;
  uint32_t red_mask[RS_MAX_K];
};
// LOGN = 13 (N_enc = 8192): the workgroup owns HALF of the spectrum and folds one stage while the row is loaded (PAIRED:
// rows in plain_center_wide_kernel's paired layout, one 16-byte load per operand pair).
template <int LOGN, bool PAIRED>
__global__ void __launch_bounds__(512, 2)
mac_kernel_v4(MacArgs4 a, int L, int K, const NttTable *__restrict__ coeff_tabs) {
  constexpr int n = 1 << LOGN, H = 4096, LOGP = LOGN - 12, PARTS = 1 << LOGP, TILE = H + H / 8;
  static_assert(LOGN == 14 ? !PAIRED : LOGN == 13, "quarter spectrum at 16384 points, half spectrum at 8192");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const unsigned b = blockIdx.x, x = b & 7u;  // XCD slot and position in its sequence, as mac_kernel_v3
  unsigned q = b >> 3;
  const int g = (int)(q % (unsigned)a.n_groups);
  q /= (unsigned)a.n_groups;
  const unsigned hj = q % ((unsigned)PARTS * (unsigned)K);
  const unsigned rr = (q / ((unsigned)PARTS * (unsigned)K)) * 8u + x;  // (chunk, limb)
  const int h = (int)(hj & (unsigned)(PARTS - 1)), j = (int)(hj >> LOGP);
  const int limb = (int)(rr % (unsigned)L), chunk = (int)(rr / (unsigned)L);
  if (chunk >= a.n_chunks) return;
  const Mod mod = coeff_tabs[j].mod;
  const double *__restrict__ tw = coeff_tabs[j].d_tw;
  const uint32_t red_mask = a.red_mask[j];
  const int root = PARTS + h;
  const double w0 = uniform_f64(tw[1]);
  const double w1 = uniform_f64(tw[2 + (h >> 1)]);  // LOGN = 14: stage 1 of this quarter's half
  double tw1[7], tw2[7], tw3[7], tw4[7];
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int bk = 0; bk < (1 << k); bk++) {
      tw1[(1 << k) - 1 + bk] = uniform_f64(tw[(root << k) + bk]);
      tw2[(1 << k) - 1 + bk] = uniform_f64(tw[(root << (3 + k)) + (wave << k) + bk]);
      tw3[(1 << k) - 1 + bk] = tw[(root << (6 + k)) + ((t >> 3) << k) + bk];
      tw4[(1 << k) - 1 + bk] = tw[(root << (9 + k)) + (t << k) + bk];
    }
#pragma unroll
  for (int i = 0; i < 7; i++) {
    pin(tw3[i]);
    pin(tw4[i]);
  }
  const size_t enc_words = (size_t)L * 2 * K * n;
  const size_t slab = (((size_t)limb * 2) * K + j) * (size_t)n + (size_t)h * H;  // component 0; component 1 is + K*n
  const size_t comp = (size_t)K * n;
  const int r0 = wave * 512;  // the wave's range of the quarter: its 64 round-4 groups
  uint64_t *part[2];
  double acc[2][2][8];
#pragma unroll
  for (int kx = 0; kx < 2; kx++) {
    part[kx] = a.partial[kx][g] + (size_t)chunk * a.part_stride + slab + r0;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        acc[kx][c][2 * i] = acc[kx][c][2 * i + 1] = 0.0;
        if (a.accumulate) {
          const u64x2 v = reinterpret_cast<const u64x2 *>(part[kx] + c * comp)[lane + 64 * i];
          acc[kx][c][2 * i] = from_u64(v.x);
          acc[kx][c][2 * i + 1] = from_u64(v.y);
        }
      }
  }
  const unsigned long long tbeg = (unsigned long long)chunk * a.terms_per_chunk;
  const unsigned long long tend = min(tbeg + (unsigned long long)a.terms_per_chunk, a.terms[g]);
  const double *crow = a.C[g] + ((size_t)tbeg * L + limb) * (size_t)n + (PAIRED ? 2 * t : t);
  const uint64_t *ctp0 = a.crs[0] + (size_t)tbeg * enc_words + slab + r0;
  const uint64_t *ctp1 = a.crs[1] + (size_t)tbeg * enc_words + slab + r0;
  u64x2 ct[2][2][4];
  auto issue_ct = [&]() {
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        ct[0][c][i] = stream_load(reinterpret_cast<const u64x2 *>(ctp0 + c * comp) + lane + 64 * i);
        ct[1][c][i] = stream_load(reinterpret_cast<const u64x2 *>(ctp1 + c * comp) + lane + 64 * i);
      }
    ctp0 += enc_words;
    ctp1 += enc_words;
  };
  int since = 0;
  auto mac = [&](const double *tile) {
    const int p0 = r0 + (r0 >> 3) + 2 * lane + (lane >> 2);  // position of element r0 + 2 lane
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const double u0 = tile[p0 + 144 * i], u1 = tile[p0 + 144 * i + 1];
#pragma unroll
      for (int kx = 0; kx < 2; kx++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
          acc[kx][c][2 * i] += mulmod(from_u64(ct[kx][c][i].x), u0, mod);
          acc[kx][c][2 * i + 1] += mulmod(from_u64(ct[kx][c][i].y), u1, mod);
        }
    }
    if (++since >= a.acc_period) {
      since = 0;
#pragma unroll
      for (int kx = 0; kx < 2; kx++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
          for (int i = 0; i < 8; i++) acc[kx][c][i] = reduce(acc[kx][c][i], mod);
    }
  };
  for (unsigned long long tt = tbeg; tt < tend; tt++) {
    double *tile = s + (int)((tt - tbeg) & 1) * TILE;
    double v[8], c2[8], c3[8];
    if (LOGN == 13) {
      // stage 0 (gap 4096) on x[n'], x[n' + 4096], n' = t + 512 e: this half's operand of the 4096-point sub-transform
#pragma unroll
      for (int e = 0; e < 8; e++) {
        if (PAIRED) {
          const double2 x2 = reinterpret_cast<const double2 *>(crow)[512 * e];
          v[e] = x2.x;
          c2[e] = x2.y;
        } else {
          v[e] = crow[512 * e];
          c2[e] = crow[512 * e + H];
        }
      }
      crow += (size_t)L * n;
      mem_fence();
      if (tt > tbeg) mac(s + (int)((tt - tbeg + 1) & 1) * TILE);  // term tt - 1, under the row loads
      mem_fence();
#pragma unroll
      for (int e = 0; e < 8; e++) {
        double x0 = v[e], x1 = c2[e];
        if (red_mask & 1u) {
          x0 = reduce(x0, mod);
          x1 = reduce(x1, mod);
        }
        x1 = mulmod(x1, w0, mod);
        v[e] = h ? x0 - x1 : x0 + x1;
        pin(v[e]);
      }
    } else {
      // stages 0 (gap 8192) and 1 (gap 4096) on x[n'], x[n' + 4096], x[n' + 8192], x[n' + 12288], n' = t + 512 e
#pragma unroll
      for (int e = 0; e < 8; e++) {
        c2[e] = crow[512 * e + 2 * H];
        c3[e] = crow[512 * e + 3 * H];
      }
      mem_fence();
      if (tt > tbeg) mac(s + (int)((tt - tbeg + 1) & 1) * TILE);  // term tt - 1, under the row loads
      mem_fence();
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = crow[512 * e];
      mem_fence();
#pragma unroll
      for (int e = 0; e < 8; e++) {
        double x2 = c2[e], x3 = c3[e];
        if (red_mask & 1u) {
          x2 = reduce(x2, mod);
          x3 = reduce(x3, mod);
        }
        c2[e] = mulmod(x2, w0, mod);
        c3[e] = mulmod(x3, w0, mod);
        pin(c2[e]);
        pin(c3[e]);
      }
#pragma unroll
      for (int e = 0; e < 8; e++) {  // u0 = x0 +- w0 x2 (this quarter's half of stage 0)
        double x0 = v[e];
        if (red_mask & 1u) x0 = reduce(x0, mod);
        c2[e] = (h & 2) ? x0 - c2[e] : x0 + c2[e];
        pin(c2[e]);
      }
      mem_fence();
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = crow[512 * e + H];
      crow += (size_t)L * n;
      mem_fence();
#pragma unroll
      for (int e = 0; e < 8; e++) {  // u1 = x1 +- w0 x3, then stage 1: v = u0 +- w1 u1
        double x1 = v[e];
        if (red_mask & 1u) x1 = reduce(x1, mod);
        double u1 = (h & 2) ? x1 - c3[e] : x1 + c3[e];
        if (red_mask & 2u) {
          u1 = reduce(u1, mod);
          c2[e] = reduce(c2[e], mod);
        }
        u1 = mulmod(u1, w1, mod);
        v[e] = (h & 1) ? c2[e] - u1 : c2[e] + u1;
        pin(v[e]);
      }
    }
    mem_fence();
    issue_ct();  // both keys' words of this term: a whole transform to land
    mem_fence();
    reg_fwd_stages<3, true>(v, mod, red_mask >> LOGP, [&](int k, int bk) { return tw1[(1 << k) - 1 + bk]; });
    {  // tile tt % 2 was last read by the multiply-accumulate of term tt - 2, before the previous barrier
      const int pb = t + (t >> 3);
#pragma unroll
      for (int e = 0; e < 8; e++) tile[pb + 576 * e] = v[e];
    }
    __syncthreads();
    {  // round 2: the wave's own 512 elements
      const int pb = wave * 576 + lane + (lane >> 3);
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = tile[pb + 72 * e];
      reg_fwd_stages<3, true>(v, mod, red_mask >> (LOGP + 3), [&](int k, int bk) { return tw2[(1 << k) - 1 + bk]; });
#pragma unroll
      for (int e = 0; e < 8; e++) tile[pb + 72 * e] = v[e];
    }
    wave_sync();
    {  // round 3: 64-element groups of 8 consecutive threads
      const int pb = (t >> 3) * 72 + (t & 7);
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = tile[pb + 9 * e];
      reg_fwd_stages<3, true>(v, mod, red_mask >> (LOGP + 6), [&](int k, int bk) { return tw3[(1 << k) - 1 + bk]; });
#pragma unroll
      for (int e = 0; e < 8; e++) tile[pb + 9 * e] = v[e];
    }
    wave_sync();
    {  // round 4: 8 consecutive points
      const int pb = 9 * t;
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = tile[pb + e];
      reg_fwd_stages<3, true>(v, mod, red_mask >> (LOGP + 9), [&](int k, int bk) { return tw4[(1 << k) - 1 + bk]; });
#pragma unroll
      for (int e = 0; e < 8; e++) tile[pb + e] = a.reduce_u ? reduce(v[e], mod) : v[e];
    }
    wave_sync();
  }
  if (tend > tbeg) mac(s + (int)((tend - tbeg + 1) & 1) * TILE);  // the last term
#pragma unroll
  for (int kx = 0; kx < 2; kx++)
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        u64x2 o;
        o.x = to_u64(canon(acc[kx][c][2 * i], mod));
        o.y = to_u64(canon(acc[kx][c][2 * i + 1], mod));
        reinterpret_cast<u64x2 *>(part[kx] + c * comp)[lane + 64 * i] = o;
      }
}

// out[set] = sum_chunk partial[chunk][set] (+ addend[set]) mod Q_j
struct ReduceArgs {
  const uint64_t *addend[12];
};
__global__ void __launch_bounds__(256)
reduce_kernel(const uint64_t *__restrict__ partial, uint64_t *__restrict__ out, ReduceArgs add, int n_chunks,
              int n_sets, size_t enc_words, int n, int K, const uint64_t *__restrict__ Qint) {
  const size_t total = (size_t)n_sets * enc_words;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t set = i / enc_words, w = i % enc_words;
    const uint64_t Q = Qint[(w / (size_t)n) % (size_t)K];
    uint64_t sum = 0;  // modular running sum: residues may be 61 bits wide, a plain sum of many of them would wrap
    for (int c = 0; c < n_chunks; c++) {
      sum += partial[((size_t)c * n_sets + set) * enc_words + w];
      sum = sum >= Q ? sum - Q : sum;
    }
    if (add.addend[set]) {
      sum += add.addend[set][w];
      sum = sum >= Q ? sum - Q : sum;
    }
    out[i] = sum;
  }
}

// a8: EncodingElem::operator+= (dyadic add mod Q_j)
__global__ void __launch_bounds__(256)
enc_add_kernel(uint64_t *__restrict__ dst, const uint64_t *__restrict__ x, const uint64_t *__restrict__ y,
               size_t words, int n, int K, const uint64_t *__restrict__ Qint) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) {
    const uint64_t Q = Qint[(i / (size_t)n) % (size_t)K];
    uint64_t sm = x[i] + y[i];
    dst[i] = sm >= Q ? sm - Q : sm;
  }
}

// canonicalise integer sums of residues (after an all-reduce of partial encoding sums)
__global__ void __launch_bounds__(256)
enc_reduce_kernel(uint64_t *__restrict__ x, size_t words, int n, int K, const uint64_t *__restrict__ Qint) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride)
    x[i] = x[i] % Qint[(i / (size_t)n) % (size_t)K];
}

static int tile_threads(int logn) { return std::max(64, std::min(1024, (1 << logn) >> 3)); }

struct MsmScratch {
  void *d_plain_tabs = nullptr, *d_coeff_tabs = nullptr;  // device copies of the context's tables (NttTable or NttTableI)
  uint64_t *d_Qint = nullptr;
  void *d_coeff_tabs_f64 = nullptr;  // hybrid contexts: FP64 tables of the data primes beside the integer ones
  // host-resident keys: copy stream and the events of the two staging buffers (copied: data landed; freed: its readers ran)
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_freed[2] = {nullptr, nullptr};
  template <class M>
  const NttTableT<typename ArithOf<M>::T, M> *plain() const {
    return static_cast<const NttTableT<typename ArithOf<M>::T, M> *>(d_plain_tabs);
  }
  template <class M>
  const NttTableT<typename ArithOf<M>::T, M> *coeff() const {
    return static_cast<const NttTableT<typename ArithOf<M>::T, M> *>(d_coeff_tabs);
  }
};

static std::map<rs_ctx *, MsmScratch> g_scratch;
static std::mutex g_scratch_mu;

template <class Table>
static void *copy_tables(const Table *h, int n) {
  void *d = nullptr;
  RS_HIP(hipMalloc(&d, sizeof(Table) * n));
  RS_HIP(hipMemcpy(d, h, sizeof(Table) * n, hipMemcpyHostToDevice));
  return d;
}
static MsmScratch &scratch_for(rs_ctx *ctx) {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  auto it = g_scratch.find(ctx);
  if (it != g_scratch.end()) return it->second;
  MsmScratch sc;
  if (ctx->use_int) {
    sc.d_plain_tabs = copy_tables(ctx->plain_i, ctx->L);
    sc.d_coeff_tabs = copy_tables(ctx->coeff_i, ctx->K);
    if (ctx->hybrid) sc.d_coeff_tabs_f64 = copy_tables(ctx->coeff, ctx->K);
  } else {
    sc.d_plain_tabs = copy_tables(ctx->plain, ctx->L);
    sc.d_coeff_tabs = copy_tables(ctx->coeff, ctx->K);
  }
  RS_HIP(hipMalloc(&sc.d_Qint, sizeof(uint64_t) * ctx->K));
  RS_HIP(hipMemcpy(sc.d_Qint, ctx->Q, sizeof(uint64_t) * ctx->K, hipMemcpyHostToDevice));
  return g_scratch[ctx] = sc;
}
void msm_scratch_release(rs_ctx *ctx) {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  auto it = g_scratch.find(ctx);
  if (it == g_scratch.end()) return;
  (void)hipFree(it->second.d_plain_tabs);
  (void)hipFree(it->second.d_coeff_tabs);
  (void)hipFree(it->second.d_Qint);
  if (it->second.d_coeff_tabs_f64) (void)hipFree(it->second.d_coeff_tabs_f64);
  if (it->second.copy_stream) {
    (void)hipStreamDestroy(it->second.copy_stream);
    for (int b = 0; b < 2; b++) {
      (void)hipEventDestroy(it->second.ev_copied[b]);
      (void)hipEventDestroy(it->second.ev_freed[b]);
    }
  }
  g_scratch.erase(it);
}

template <int NG, int NC, int PAIRS, class M>
static void launch_mac(rs_ctx *ctx, const MacArgs &a, const MsmScratch &sc, hipStream_t st) {
  const size_t lds = padded_len((size_t)ctx->N_enc) * sizeof(double);
  const int rows = a.n_chunks * ctx->L;
  const unsigned blocks = (unsigned)(((rows + 7) / 8) * 8 * ctx->K);
  RS_HIP(hipFuncSetAttribute((const void *)mac_kernel<NG, NC, PAIRS, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((mac_kernel<NG, NC, PAIRS, M>), dim3(blocks), dim3(tile_threads(ctx->logN_enc)), lds, st, a, ctx->L,
                     ctx->K, ctx->logN_enc, sc.template coeff<M>());
  RS_HIP(hipGetLastError());
}

extern int g_mac_variant, g_mac_ablate, g_plain_variant, g_mac_chunk_units, g_mac_share_keys;
static void launch_mac_v2(rs_ctx *ctx, const MacArgs2 &a, const MsmScratch &sc, hipStream_t st) {
  const size_t lds = (padded_len((size_t)ctx->N_enc) + (size_t)ctx->N_enc) * sizeof(double);
  const int rows = a.n_chunks * ctx->L;
  const unsigned blocks = (unsigned)(((rows + 7) / 8) * 8 * ctx->K);
  if (g_mac_variant == 3 && ctx->logN_enc == 13) {  // 16 waves of 128 VGPRs (4 waves per SIMD), plaintext row not prefetched
    RS_HIP(hipFuncSetAttribute((const void *)mac_kernel_v2<1024, 13>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((mac_kernel_v2<1024, 13>), dim3(blocks), dim3(1024), lds, st, a, ctx->L, ctx->K, ctx->logN_enc, sc.coeff<Mod>());
    RS_HIP(hipGetLastError());
    return;
  }
  if (ctx->logN_enc == 13 && g_mac_variant != 4) {
#define RS_MAC_LAUNCH(AB)                                                                                             \
  do {                                                                                                                \
    RS_HIP(hipFuncSetAttribute((const void *)mac_kernel_v2<512, 13, AB>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                               (int)lds));                                                                           \
    hipLaunchKernelGGL((mac_kernel_v2<512, 13, AB>), dim3(blocks), dim3(512), lds, st, a, ctx->L, ctx->K,             \
                       ctx->logN_enc, sc.coeff<Mod>());                                                               \
  } while (0)
#ifdef RS_EXPERIMENTS  // timing-only ablations (wrong results): never part of the release library
    switch (g_mac_ablate) {
      case 1: RS_MAC_LAUNCH(1); break;
      case 2: RS_MAC_LAUNCH(2); break;
      case 6: RS_MAC_LAUNCH(6); break;
      default: RS_MAC_LAUNCH(0); break;
    }
#else
    RS_MAC_LAUNCH(0);
#endif
#undef RS_MAC_LAUNCH
    RS_HIP(hipGetLastError());
    return;
  }
  RS_HIP(hipFuncSetAttribute((const void *)mac_kernel_v2<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(mac_kernel_v2<512>, dim3(blocks), dim3(512), lds, st, a, ctx->L, ctx->K, ctx->logN_enc, sc.coeff<Mod>());
  RS_HIP(hipGetLastError());
}

// Reduction points of the forward transform mod p when the inputs are bounded by |x| <= b0 (not by p): bit s = every value
// is brought back to |v| <= p/2 before stage s, so that multiplier operands stay below 2^50 (f64mod.hpp).
static uint32_t fwd_reduce_mask_from(uint64_t p, int logn, double b0, double *end_bound) {
  const double lim = 1125899906842624.0 / (double)p;
  double B = b0 / (double)p;
  uint32_t mask = 0;
  for (int s = 0; s < logn; s++) {
    if (B > lim) {
      mask |= 1u << s;
      B = 0.51;
    }
    B += 0.75;
  }
  *end_bound = B * (double)p;  // bound of the spectrum values
  return mask;
}
static void launch_mac_v3(rs_ctx *ctx, const MacArgs3 &a, const MsmScratch &sc, hipStream_t st) {
  const size_t lds = (size_t)(2 * (4096 + 256) + 1024) * sizeof(double);  // two tiles + round-2 twiddles
  const unsigned rows = (unsigned)ctx->L * (unsigned)a.n_chunks;  // (chunk, limb), spread over the XCDs
  // FP64 tables of the data primes: the context's own, or -- hybrid context -- the copies kept beside the integer tables
  const NttTable *tabs = ctx->use_int ? static_cast<const NttTable *>(sc.d_coeff_tabs_f64) : sc.coeff<Mod>();
  const unsigned parts = (unsigned)ctx->N_enc / 4096u;  // workgroups per (limb, prime): halves at 8192 points, quarters at 16384
  const unsigned blocks = ((rows + 7) / 8) * 8 * parts * (unsigned)ctx->K * (unsigned)a.n_groups;
  if (ctx->N_enc == 16384) {
    RS_HIP(hipFuncSetAttribute((const void *)mac_kernel_v3<false, 14>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((mac_kernel_v3<false, 14>), dim3(blocks), dim3(256), lds, st, a, ctx->L, ctx->K, tabs);
  } else if (a.paired) {
    RS_HIP(hipFuncSetAttribute((const void *)mac_kernel_v3<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(mac_kernel_v3<true>, dim3(blocks), dim3(256), lds, st, a, ctx->L, ctx->K, tabs);
  } else {
    RS_HIP(hipFuncSetAttribute((const void *)mac_kernel_v3<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(mac_kernel_v3<false>, dim3(blocks), dim3(256), lds, st, a, ctx->L, ctx->K, tabs);
  }
  RS_HIP(hipGetLastError());
}

static void launch_mac_v4(rs_ctx *ctx, const MacArgs4 &a, bool paired, const MsmScratch &sc, hipStream_t st) {
  const size_t lds = (size_t)2 * (4096 + 512) * sizeof(double);  // two tiles
  const unsigned rows = (unsigned)ctx->L * (unsigned)a.n_chunks;
  const NttTable *tabs = ctx->use_int ? static_cast<const NttTable *>(sc.d_coeff_tabs_f64) : sc.coeff<Mod>();
  const unsigned parts = (unsigned)ctx->N_enc / 4096u;
  const unsigned blocks = ((rows + 7) / 8) * 8 * parts * (unsigned)ctx->K * (unsigned)a.n_groups;
#define RS_MAC4_LAUNCH(LOGN_, PAIRED_)                                                                                          \
  do {                                                                                                                         \
    RS_HIP(hipFuncSetAttribute((const void *)mac_kernel_v4<LOGN_, PAIRED_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((mac_kernel_v4<LOGN_, PAIRED_>), dim3(blocks), dim3(512), lds, st, a, ctx->L, ctx->K, tabs);             \
  } while (0)
  if (ctx->N_enc == 16384) RS_MAC4_LAUNCH(14, false);
  else if (paired) RS_MAC4_LAUNCH(13, true);
  else RS_MAC4_LAUNCH(13, false);
#undef RS_MAC4_LAUNCH
  RS_HIP(hipGetLastError());
}

int g_mac_ablate = 0;
int g_mac_share_keys = 1;  // tuning knob "mac_share_keys": two key vectors share the plaintext spectrum (mac_kernel_v4; 8192 and 16384 points)
int g_msm_host_tile = 1024;  // tuning knob "msm_host_tile": terms per staging buffer of a host-resident key
int g_mac_chunk_units = 768;  // tuning knob "mac_chunk_units": (limb, prime, chunk) units per MAC launch (term chunks = units / (L K))
int g_plain_variant = 1;  // 1: plain_center_wide_kernel at N_enc = 8192; 0: plain_center_kernel
int g_mac_variant = 5;  // 5: half-spectrum wide kernel at N_enc = 8192 (else as 3); 3: streaming kernel, 1024-thread shape at N_enc = 8192 (else as 2); 2: 512-thread streaming kernel; 1: generic kernel

// Core grouped MSM.  addends: optional per-output (n_crs * n_groups) device pointers to encoding
// elements added to the result (pk.alpha / pk.beta of groth16.tcc:95,103).
// crs_window != 0: every CRS vector is stored as `crs_window` consecutive elements and logical
// element t lives at index t % crs_window (tiled keys, see ringsnark_amd.h); tiles never straddle the wrap.
template <class M>
static void msm_run_arith(rs_ctx *ctx, const uint64_t *const *d_crs, int n_crs, size_t crs_len, const rs_msm_vec *vecs, int n_vecs,
                          int n_groups, uint64_t *d_out, const uint64_t *const *addends, size_t *h_used, hipStream_t st,
                          size_t crs_window, const MsmLin *lin, bool crs_on_host) {
  using Lift = typename ArithOf<M>::Lift;
  constexpr bool FP = std::is_same<M, Mod>::value;
  RS_REQUIRE(n_crs >= 1 && n_crs <= 2, "n_crs must be 1 or 2");
  RS_REQUIRE(n_groups >= 1 && n_groups <= MAX_GROUPS, "too many groups");
  RS_REQUIRE(n_crs * n_groups <= 12, "too many outputs");
  const int L = ctx->L, K = ctx->K, n = ctx->N_enc;
  const size_t enc_words = ctx->enc_words();
  MsmScratch &sc = scratch_for(ctx);

  // group bookkeeping
  PlainArgs pa;
  memset(&pa, 0, sizeof(pa));
  size_t Tmax = 0;
  std::vector<size_t> group_T(n_groups, 0);
  std::vector<unsigned *> nz_ptr(n_vecs, nullptr);
  size_t nz_total = 0, kinds_total = 0;
  for (int v = 0; v < n_vecs; v++) {
    RS_REQUIRE(vecs[v].group >= 0 && vecs[v].group < n_groups, "group index out of range");
    RS_REQUIRE(vecs[v].T <= crs_len, "coefficient vector longer than the CRS vector");
    RS_REQUIRE(vecs[v].d_coeff || vecs[v].T == 0, "null coefficient vector");
    Tmax = std::max(Tmax, vecs[v].T);
    group_T[vecs[v].group] = std::max(group_T[vecs[v].group], vecs[v].T);
    if (h_used) nz_total += vecs[v].T;
    if (vecs[v].h_kinds) kinds_total += vecs[v].T;
  }
  unsigned *d_nz = nullptr;
  uint8_t *d_kinds = nullptr;
  if (nz_total) {
    d_nz = (unsigned *)ws_get(ctx, 2, nz_total * sizeof(unsigned));
    RS_HIP(hipMemsetAsync(d_nz, 0, nz_total * sizeof(unsigned), st));
  }
  if (kinds_total) d_kinds = (uint8_t *)ws_get(ctx, 3, kinds_total);
  {
    size_t nzo = 0, ko = 0;
    for (int v = 0; v < n_vecs; v++) {
      PlainGroup &G = pa.g[vecs[v].group];
      RS_REQUIRE(G.n < MAX_GROUP_VECS, "too many vectors in one group");
      G.coeff[G.n] = vecs[v].d_coeff;
      G.T[G.n] = vecs[v].T;
      if (h_used) {
        nz_ptr[v] = d_nz + nzo;
        G.nz[G.n] = nz_ptr[v];
        nzo += vecs[v].T;
      }
      if (vecs[v].h_kinds) {
        RS_HIP(hipMemcpyAsync(d_kinds + ko, vecs[v].h_kinds, vecs[v].T, hipMemcpyHostToDevice, st));
        G.kinds[G.n] = d_kinds + ko;
        ko += vecs[v].T;
      }
      G.n++;
    }
  }

  bool has_lin = false;
  if (lin)
    for (int g = 0; g < n_groups; g++)
      if (lin[g].count > 0) {
        RS_REQUIRE(lin[g].T <= crs_len && !h_used, "linear-form vector: too long, or used-term counts requested");
        pa.g[g].lin = lin[g];
        has_lin = true;
        Tmax = std::max<size_t>(Tmax, (size_t)lin[g].T);
        group_T[g] = std::max<size_t>(group_T[g], (size_t)lin[g].T);
      }
  const int n_sets = n_crs * n_groups;
  // tiling: C workspace <= ~2 GiB
  const size_t c_bytes_per_term = (size_t)n_groups * L * n * sizeof(double);
  size_t tile_terms = std::max<size_t>(1, std::min<size_t>(Tmax, ((size_t)2 << 30) / c_bytes_per_term));
  if (crs_window) {
    size_t p2 = 1;
    while (p2 * 2 <= tile_terms) p2 *= 2;
    tile_terms = std::min(p2, crs_window);
    RS_REQUIRE(crs_window % tile_terms == 0, "crs_window must be a multiple of the term tile (use a power of two)");
  }
  // Host-resident key (crs_on_host: d_crs are HOST pointers -- a proving key larger than HBM, e.g. the 384 GiB key of the
  // 2^16-constraint headline on one GPU): the term tiles are streamed through two device staging buffers; the copy of
  // tile k+1 runs on its own stream under the kernels of tile k (pinned host memory, rs_host_alloc, for real overlap).
  uint64_t *stage = nullptr;
  size_t stage_words = 0;
  if (crs_on_host) {
    tile_terms = std::min<size_t>(tile_terms, (size_t)std::max(1, g_msm_host_tile));
    if (crs_window) {
      size_t p2 = 1;
      while (p2 * 2 <= tile_terms) p2 *= 2;
      tile_terms = std::min(p2, crs_window);
    }
    stage_words = tile_terms * enc_words;
    stage = (uint64_t *)ws_get(ctx, 7, (size_t)2 * n_crs * stage_words * sizeof(uint64_t));
    if (!sc.copy_stream) {
      RS_HIP(hipStreamCreateWithFlags(&sc.copy_stream, hipStreamNonBlocking));
      for (int b = 0; b < 2; b++) {
        RS_HIP(hipEventCreateWithFlags(&sc.ev_copied[b], hipEventDisableTiming));
        RS_HIP(hipEventCreateWithFlags(&sc.ev_freed[b], hipEventDisableTiming));
      }
    }
    // the staging buffers may still be read by an earlier call on another stream: order the copy stream after `st`
    RS_HIP(hipEventRecord(sc.ev_freed[0], st));
    RS_HIP(hipEventRecord(sc.ev_freed[1], st));
  }
  auto stage_at = [&](int buf, int c) { return stage + ((size_t)buf * n_crs + c) * stage_words; };
  auto issue_copy = [&](int tile, size_t t0) {  // tile -> staging buffer tile % 2, on the copy stream
    const int buf = tile & 1;
    const size_t tt = std::min(tile_terms, Tmax - t0);
    RS_HIP(hipStreamWaitEvent(sc.copy_stream, sc.ev_freed[buf], 0));
    for (int c = 0; c < n_crs; c++)
      RS_HIP(hipMemcpyAsync(stage_at(buf, c), d_crs[c] + (crs_window ? t0 % crs_window : t0) * enc_words, tt * enc_words * sizeof(uint64_t),
                            hipMemcpyHostToDevice, sc.copy_stream));
    RS_HIP(hipEventRecord(sc.ev_copied[buf], sc.copy_stream));
  };
  int cur_tile = 0;
  auto crs_at = [&](int c, size_t t0) -> const uint64_t * {
    if (crs_on_host) return stage_at(cur_tile & 1, c);
    return d_crs[c] + (crs_window ? t0 % crs_window : t0) * enc_words;
  };
  int n_chunks = (int)std::min<size_t>(tile_terms, (size_t)std::max(1, (g_mac_chunk_units + L * K - 1) / (L * K)));
  if (Tmax == 0) n_chunks = 1;
  Lift *d_C = (Lift *)ws_get(ctx, 0, std::max<size_t>(256, tile_terms * c_bytes_per_term));
  uint64_t *d_partial = (uint64_t *)ws_get(ctx, 1, (size_t)n_chunks * n_sets * enc_words * sizeof(uint64_t));
  const size_t lds = padded_len((size_t)n) * sizeof(double);
  const int thr = tile_threads(ctx->logN_enc);
  const bool big = n > 8192;  // one accumulator set per MAC launch
  const bool plain16 = n >= 2048;  // 16 coefficients per thread, wave-private inverse transform
  const int plain_thr = plain16 ? n / 16 : thr;
  const bool plain13 = false;  // measured: no gain from a compile-time length here
  (void)plain13;
  bool v3 = false, plain_wide = false, hybrid = false;
  if constexpr (FP) {
    v3 = g_mac_variant == 5 && (n == 8192 || n == 16384);
    plain_wide = g_plain_variant == 1 && n == 8192 && (ctx->N == 8192 || ctx->N == 4096);
  } else {
    // Hybrid context: ring primes beyond 2^50 (SEAL's 54-bit BFVDefault(2048) prime of the reference's logistic-regression
    // benchmark) on the integer arithmetic, data primes below 2^50: the plaintext row is produced by the integer kernel and
    // handed, as exact doubles, to the FP64 multiply-accumulate of the data primes.  Needs |sum of a group's lifts| < 2^53.
    uint64_t maxq = 0;
    for (int i = 0; i < L; i++) maxq = std::max(maxq, ctx->q[i]);
    int max_vecs = 1;
    for (int g = 0; g < n_groups; g++) max_vecs = std::max(max_vecs, pa.g[g].n);
    hybrid = ctx->hybrid && g_mac_variant == 5 && (n == 8192 || n == 16384) &&
             (double)max_vecs * (0.5 * (double)maxq + 1.0) < 9007199254740992.0;
    v3 = hybrid;
  }
  RS_REQUIRE(!has_lin || plain_wide, "linear-form vectors need the wide plaintext kernel");
  const bool paired = v3 && plain_wide;  // row layout of this call: written by the plaintext kernel, read by the MAC
  bool multi = false;
  for (int g = 0; g < n_groups; g++) multi = multi || pa.g[g].n > 1;
  if (plain16)
    RS_HIP(hipFuncSetAttribute((const void *)plain_center_kernel<16, 0, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  else
    RS_HIP(hipFuncSetAttribute((const void *)plain_center_kernel<8, 0, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

  // rooflines: per (term, limb) a plaintext row costs one inverse transform of
  // length n, the batching scatter and the centred lift; per (term, limb, prime) the MAC reads two
  // ciphertext polynomials and one plaintext row, runs one forward transform and 2n multiply-adds
  const double nd = (double)n, logn_d = (double)ctx->logN_enc;
  if (Tmax == 0) RS_HIP(hipMemsetAsync(d_partial, 0, (size_t)n_chunks * n_sets * enc_words * sizeof(uint64_t), st));

  int tile_idx = 0;
  if (crs_on_host && Tmax) issue_copy(0, 0);
  for (size_t t0 = 0; t0 < Tmax; t0 += tile_terms, tile_idx++) {
    const size_t tt = std::min(tile_terms, Tmax - t0);
    cur_tile = tile_idx;
    if (crs_on_host) {
      if (t0 + tile_terms < Tmax) issue_copy(tile_idx + 1, t0 + tile_terms);  // under this tile's kernels
      RS_HIP(hipStreamWaitEvent(st, sc.ev_copied[tile_idx & 1], 0));
    }
    double rows_in = 0;  // coefficient rows (term, limb) read by this tile
    for (int v = 0; v < n_vecs; v++) rows_in += (double)(vecs[v].T > t0 ? std::min(tt, vecs[v].T - t0) : 0) * L;
    {
    ProfScope prof_plain(ctx, st, plain_wide ? "plain_center_wide_kernel" : "plain_center_kernel", rows_in * (double)ctx->N * 8.0 + (double)tt * L * n_groups * nd * 8.0,
                         rows_in * (ntt_fp64(nd, logn_d) + 14.0 * nd));
    if (plain_wide) {
      if constexpr (FP) {
        const int wl = (int)WideShape<13>::LDS_BYTES;
        const unsigned long long items = (unsigned long long)tt * n_groups;
        const unsigned slots = (unsigned)std::max<unsigned long long>(1, std::min<unsigned long long>(items, (512 + L - 1) / L));
        const dim3 grid(slots * (unsigned)L);
        PlainTwPtrs twp;
        memset(&twp, 0, sizeof(twp));
        for (int i = 0; i < L; i++) twp.itw[i] = ctx->plain[i].d_itw;
#define RS_PLAIN_WIDE_NE(MULTI_, PAIRED_, NE_, LIN_)                                                                          \
  do {                                                                                                                        \
    RS_HIP(hipFuncSetAttribute((const void *)plain_center_wide_kernel<MULTI_, PAIRED_, NE_, LIN_>,                            \
                               hipFuncAttributeMaxDynamicSharedMemorySize, wl));                                              \
    hipLaunchKernelGGL((plain_center_wide_kernel<MULTI_, PAIRED_, NE_, LIN_>), grid, dim3(256), wl, st, pa,                   \
                       reinterpret_cast<double *>(d_C), (unsigned long long)t0, (unsigned long long)tile_terms,               \
                       (unsigned long long)tt, n_groups, ctx->N, L, ctx->d_index_map, sc.plain<Mod>(), twp);                  \
  } while (0)
#define RS_PLAIN_WIDE(MULTI_, PAIRED_)                                  \
  do {                                                                  \
    if (ctx->N == 8192) {                                               \
      if (has_lin) RS_PLAIN_WIDE_NE(MULTI_, PAIRED_, 16, true);         \
      else RS_PLAIN_WIDE_NE(MULTI_, PAIRED_, 16, false);                \
    } else {                                                            \
      if (has_lin) RS_PLAIN_WIDE_NE(MULTI_, PAIRED_, 8, true);          \
      else RS_PLAIN_WIDE_NE(MULTI_, PAIRED_, 8, false);                 \
    }                                                                   \
  } while (0)
        if (multi && paired) RS_PLAIN_WIDE(true, true);
        else if (multi) RS_PLAIN_WIDE(true, false);
        else if (paired) RS_PLAIN_WIDE(false, true);
        else RS_PLAIN_WIDE(false, false);
#undef RS_PLAIN_WIDE_NE
#undef RS_PLAIN_WIDE
      }
    } else if (plain16)
      hipLaunchKernelGGL((plain_center_kernel<16, 0, M>), dim3((unsigned)tt, L, n_groups), dim3(plain_thr), lds, st, pa, d_C,
                         (unsigned long long)t0, (unsigned long long)tile_terms, ctx->N, L, ctx->logN_enc,
                         ctx->d_index_map, sc.template plain<M>(), hybrid ? 1 : 0);
    else
      hipLaunchKernelGGL((plain_center_kernel<8, 0, M>), dim3((unsigned)tt, L, n_groups), dim3(thr), lds, st, pa, d_C,
                         (unsigned long long)t0, (unsigned long long)tile_terms, ctx->N, L, ctx->logN_enc,
                         ctx->d_index_map, sc.template plain<M>(), hybrid ? 1 : 0);
    }
    RS_HIP(hipGetLastError());
    MacArgs base;
    memset(&base, 0, sizeof(base));
    base.partial = d_partial;
    base.n_sets_total = n_sets;
    base.tile_terms = tt;
    base.n_chunks = n_chunks;
    base.terms_per_chunk = (int)((tt + n_chunks - 1) / n_chunks);
    base.accumulate = tile_idx > 0;
    uint64_t maxQ = 0;
    for (int j = 0; j < K; j++) maxQ = std::max(maxQ, ctx->Q[j]);
    base.acc_period = (int)std::max(1.0, std::floor(4503599627370496.0 / (0.8 * (double)maxQ)) - 1.0);
    base.acc_period = std::min(base.acc_period, 1 << 20);
    base.reduce_u = maxQ >= (1ull << 45);
    auto group_terms = [&](int g) -> unsigned long long {
      return group_T[g] > t0 ? (unsigned long long)std::min(tt, group_T[g] - t0) : 0ull;
    };
    auto Cptr = [&](int g) { return d_C + (size_t)g * tile_terms * L * n; };
    auto run = [&](int NG, int NC, const int *gs, const int *cs) {
      MacArgs a = base;
      for (int g = 0; g < NG; g++) {
        a.C[g] = Cptr(gs[g]);
        a.terms[g] = group_terms(gs[g]);
      }
      for (int c = 0; c < NC; c++) a.crs[c] = crs_at(cs[c], t0);
      for (int c = 0; c < NC; c++)
        for (int g = 0; g < NG; g++) a.set_index[c * NG + g] = cs[c] * n_groups + gs[g];
      double units = 0;  // (term, limb, prime) transforms
      for (int g = 0; g < NG; g++) units += (double)a.terms[g] * L * K;
      ProfScope prof(ctx, st, "mac_kernel",
                     (double)tt * NC * (double)enc_words * 8.0 + units * nd * 8.0 / K + (double)NG * NC * (double)enc_words * 8.0,
                     units * (ntt_fp64(nd, logn_d) + NC * 15.0 * nd));
      if (big)
        launch_mac<1, 1, 8, M>(ctx, a, sc, st);
      else if (NG == 2 && NC == 1)
        launch_mac<2, 1, 4, M>(ctx, a, sc, st);
      else if (NG == 1 && NC == 2)
        launch_mac<1, 2, 4, M>(ctx, a, sc, st);
      else
        launch_mac<1, 1, 4, M>(ctx, a, sc, st);
    };
    // streaming kernel, one accumulator set (CRS vector c, group g) per launch.  With two CRS
    // vectors (Rinocchio's s_pows / alpha_s_pows) the plaintext transform is repeated per vector:
    // measured faster than the generic kernel that shares it (g_mac_variant == 2: generic for n_crs == 2).
    const bool v2 = FP && g_mac_variant >= 2 && (n_crs == 1 || g_mac_variant != 2) && n >= 2048 && n <= 8192;
    if (v3) {
      {
        // bound of a plaintext row: the sum of the centred lifts of the group's vectors
        uint64_t maxq = 0;
        for (int i = 0; i < L; i++) maxq = std::max(maxq, ctx->q[i]);
        const double b0 = 0.5 * (double)maxq * MAX_GROUP_VECS + (double)MAX_GROUP_VECS;
        if (n_crs == 2 && g_mac_share_keys) {
          // Rinocchio's ten inner products (rinocchio.tcc:106-160): every vector against both key vectors, one transform
          for (int g0 = 0; g0 < n_groups; g0 += RS_MAC4_GROUPS) {
            const int ng = std::min(RS_MAC4_GROUPS, n_groups - g0);
            MacArgs4 a4;
            memset(&a4, 0, sizeof(a4));
            unsigned long long tmax = 0;
            double terms = 0;
            for (int gi = 0; gi < ng; gi++) {
              a4.C[gi] = reinterpret_cast<const double *>(Cptr(g0 + gi));
              a4.terms[gi] = group_terms(g0 + gi);
              for (int c = 0; c < 2; c++) a4.partial[c][gi] = d_partial + (size_t)(c * n_groups + g0 + gi) * enc_words;
              tmax = std::max(tmax, a4.terms[gi]);
              terms += (double)a4.terms[gi];
            }
            a4.crs[0] = crs_at(0, t0);
            a4.crs[1] = crs_at(1, t0);
            a4.part_stride = (size_t)n_sets * enc_words;
            a4.n_groups = ng;
            a4.n_chunks = base.n_chunks;
            a4.terms_per_chunk = base.terms_per_chunk;
            a4.accumulate = base.accumulate;
            a4.acc_period = base.acc_period;
            for (int jj = 0; jj < K; jj++) {
              double end = 0;
              a4.red_mask[jj] = fwd_reduce_mask_from(ctx->Q[jj], ctx->logN_enc, b0, &end);
              if (end > 562949953421312.0) a4.reduce_u = 1;
            }
            ProfScope prof(ctx, st, n == 16384 ? "mac_kernel_v4<14, false>" : (paired ? "mac_kernel_v4<13, true>" : "mac_kernel_v4<13, false>"), (double)tmax * 2.0 * (double)enc_words * 8.0 + terms * (double)L * nd * 8.0 + 2.0 * ng * (double)enc_words * 8.0,
                           terms * L * K * (ntt_fp64(nd, logn_d) + 8.0 * nd / 2.0 + 2.0 * 15.0 * nd));
            launch_mac_v4(ctx, a4, paired, sc, st);
          }
        } else
        // chunks: two workgroups per CU in one wave of workgroups (512), shared by the groups of a launch
        for (int c = 0; c < n_crs; c++)
          for (int g0 = 0; g0 < n_groups; g0 += 2) {
            const int ng = std::min(2, n_groups - g0);
            MacArgs3 a3;
            memset(&a3, 0, sizeof(a3));
            unsigned long long tmax = 0;
            for (int gi = 0; gi < ng; gi++) {
              a3.C[gi] = reinterpret_cast<const double *>(Cptr(g0 + gi));
              a3.terms[gi] = group_terms(g0 + gi);
              a3.partial[gi] = d_partial + (size_t)(c * n_groups + g0 + gi) * enc_words;
              tmax = std::max(tmax, a3.terms[gi]);
            }
            a3.crs = crs_at(c, t0);
            a3.part_stride = (size_t)n_sets * enc_words;
            a3.n_groups = ng;
            a3.paired = paired;
            a3.n_chunks = base.n_chunks;
            a3.terms_per_chunk = base.terms_per_chunk;
            a3.accumulate = base.accumulate;
            a3.acc_period = base.acc_period;
            a3.reduce_u = 0;
            for (int jj = 0; jj < K; jj++) {
              double end = 0;
              a3.red_mask[jj] = fwd_reduce_mask_from(ctx->Q[jj], ctx->logN_enc, b0, &end);
              if (end > 562949953421312.0) a3.reduce_u = 1;  // a spectrum value times a canonical ciphertext word: |u| <= 2^49
            }
            double terms = 0;
            for (int gi = 0; gi < ng; gi++) terms += (double)a3.terms[gi];
            // ciphertext words once per launch (the second group's read is served on-die), every plaintext row once,
            // the accumulator sets written once
            ProfScope prof(ctx, st, n == 16384 ? "mac_kernel_v3<false, 14>" : "mac_kernel_v3", (double)tmax * (double)enc_words * 8.0 + terms * (double)L * nd * 8.0 + ng * (double)enc_words * 8.0,
                           terms * L * K * (ntt_fp64(nd, logn_d) + 8.0 * nd / 2.0 + 15.0 * nd));
            launch_mac_v3(ctx, a3, sc, st);
          }
      }
    } else if (v2) {
      for (int c = 0; c < n_crs; c++)
      for (int g = 0; g < n_groups; g++) {
        MacArgs2 a2;
        a2.C = reinterpret_cast<const double *>(Cptr(g));
        a2.terms = group_terms(g);
        a2.crs = crs_at(c, t0);
        a2.partial = d_partial + (size_t)(c * n_groups + g) * enc_words;
        a2.part_stride = (size_t)n_sets * enc_words;
        a2.terms_per_chunk = base.terms_per_chunk;
        a2.n_chunks = base.n_chunks;
        a2.accumulate = base.accumulate;
        a2.acc_period = base.acc_period;
        a2.reduce_u = base.reduce_u;
        // every ciphertext word of the group's terms once, every plaintext row once (shared by the K
        // prime workgroups through L2), the accumulator set written once
        const double terms = (double)a2.terms;
        ProfScope prof(ctx, st, (g_mac_variant == 3 && ctx->logN_enc == 13) ? "mac_kernel_v2<1024, 13, 0>" : "mac_kernel_v2", terms * ((double)enc_words * 8.0 + (double)L * nd * 8.0) + (double)enc_words * 8.0,
                       terms * L * K * (ntt_fp64(nd, logn_d) + 15.0 * nd));
        launch_mac_v2(ctx, a2, sc, st);
      }
    } else if (big) {  // one (crs, group) pair per launch
      for (int c = 0; c < n_crs; c++)
        for (int g = 0; g < n_groups; g++) {
          const int gs[1] = {g}, cs1[1] = {c};
          run(1, 1, gs, cs1);
        }
    } else if (n_crs == 1) {  // share each ciphertext read between two groups
      int g = 0;
      const int c0[1] = {0};
      for (; g + 1 < n_groups; g += 2) {
        const int gs[2] = {g, g + 1};
        run(2, 1, gs, c0);
      }
      if (g < n_groups) {
        const int gs[1] = {g};
        run(1, 1, gs, c0);
      }
    } else {  // two CRS vectors: share each plaintext NTT between them
      const int cs[2] = {0, 1};
      for (int g = 0; g < n_groups; g++) {
        const int gs[1] = {g};
        run(1, 2, gs, cs);
      }
    }
    if (crs_on_host) RS_HIP(hipEventRecord(sc.ev_freed[tile_idx & 1], st));  // its readers are enqueued: the buffer may be refilled after them
  }
  ReduceArgs ra;
  memset(&ra, 0, sizeof(ra));
  if (addends)
    for (int s_ = 0; s_ < n_sets; s_++) ra.addend[s_] = addends[s_];
  {
    const size_t total = (size_t)n_sets * enc_words;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(reduce_kernel, dim3(blocks), dim3(256), 0, st, d_partial, d_out, ra, n_chunks, n_sets, enc_words, n,
                       K, sc.d_Qint);
    RS_HIP(hipGetLastError());
  }
  if (h_used) {
    std::vector<unsigned> h(nz_total);
    if (nz_total) RS_HIP(hipMemcpyAsync(h.data(), d_nz, nz_total * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    RS_HIP(hipStreamSynchronize(st));
    size_t off = 0;
    for (int v = 0; v < n_vecs; v++) {
      size_t cnt = 0;
      for (size_t t = 0; t < vecs[v].T; t++) cnt += h[off + t] != 0;
      h_used[v] = cnt;
      off += vecs[v].T;
    }
  }
}

void msm_run(rs_ctx *ctx, const uint64_t *const *d_crs, int n_crs, size_t crs_len, const rs_msm_vec *vecs, int n_vecs,
             int n_groups, uint64_t *d_out, const uint64_t *const *addends, size_t *h_used, hipStream_t st,
             size_t crs_window, const MsmLin *lin = nullptr, bool crs_on_host = false) {
  RS_DISPATCH_ARITH(ctx, (msm_run_arith<Mod>(ctx, d_crs, n_crs, crs_len, vecs, n_vecs, n_groups, d_out, addends, h_used, st, crs_window, lin, crs_on_host)),
                    (msm_run_arith<ModI>(ctx, d_crs, n_crs, crs_len, vecs, n_vecs, n_groups, d_out, addends, h_used, st, crs_window, lin, crs_on_host)));
}
// can a call with linear-form vectors be served? (FP64 context on the wide plaintext kernel)
bool msm_supports_lin(const rs_ctx *ctx) {
  return !ctx->use_int && g_plain_variant == 1 && ctx->N_enc == 8192 && (ctx->N == 8192 || ctx->N == 4096);
}
void batch_encode_run(rs_ctx *ctx, const uint64_t *d_rings, uint64_t *d_plain, size_t count, hipStream_t st) {
  if (!count) return;
  MsmScratch &sc = scratch_for(ctx);
  const size_t lds = padded_len((size_t)ctx->N_enc) * sizeof(double);
  if (ctx->use_int) {
    RS_HIP(hipFuncSetAttribute((const void *)batch_encode_kernel<ModI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(batch_encode_kernel<ModI>, dim3((unsigned)count, ctx->L), dim3(tile_threads(ctx->logN_enc)), lds, st, d_rings, d_plain,
                       ctx->N, ctx->L, ctx->logN_enc, ctx->d_index_map, sc.plain<ModI>());
  } else {
    RS_HIP(hipFuncSetAttribute((const void *)batch_encode_kernel<Mod>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(batch_encode_kernel<Mod>, dim3((unsigned)count, ctx->L), dim3(tile_threads(ctx->logN_enc)), lds, st, d_rings, d_plain,
                       ctx->N, ctx->L, ctx->logN_enc, ctx->d_index_map, sc.plain<Mod>());
  }
  RS_HIP(hipGetLastError());
}

void enc_add_run(rs_ctx *ctx, uint64_t *dst, const uint64_t *x, const uint64_t *y, size_t count, hipStream_t st) {
  const size_t words = count * ctx->enc_words();
  if (!words) return;
  MsmScratch &sc = scratch_for(ctx);
  const unsigned blocks = (unsigned)std::min<size_t>((words + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(enc_add_kernel, dim3(blocks), dim3(256), 0, st, dst, x, y, words, ctx->N_enc, ctx->K, sc.d_Qint);
  RS_HIP(hipGetLastError());
}

}  // namespace rs

using namespace rs;

extern "C" {

int rs_batch_encode(rs_ctx *ctx, const uint64_t *d_rings, uint64_t *d_plain, size_t count, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_rings && d_plain, "null argument");
  batch_encode_run(ctx, d_rings, d_plain, count, S(stream));
  RS_API_END
}

int rs_msm(rs_ctx *ctx, const uint64_t *const *d_crs, int n_crs, size_t crs_len, size_t crs_window, const rs_msm_vec *vecs,
           int n_vecs, int n_groups, uint64_t *d_out, size_t *h_used, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_crs && vecs && d_out && n_vecs >= 1, "null argument");
  WsScope ws_scope(ctx, S(stream));
  msm_run(ctx, d_crs, n_crs, crs_len, vecs, n_vecs, n_groups, d_out, nullptr, h_used, S(stream), crs_window);
  RS_API_END
}

int rs_msm_hostkey(rs_ctx *ctx, const uint64_t *const *h_crs, int n_crs, size_t crs_len, size_t crs_window, const rs_msm_vec *vecs,
                   int n_vecs, int n_groups, uint64_t *d_out, size_t *h_used, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && h_crs && vecs && d_out && n_vecs >= 1, "null argument");
  WsScope ws_scope(ctx, S(stream));
  msm_run(ctx, h_crs, n_crs, crs_len, vecs, n_vecs, n_groups, d_out, nullptr, h_used, S(stream), crs_window, nullptr, true);
  RS_HIP(hipStreamSynchronize(S(stream)));  // the caller may release or rewrite the host key on return
  RS_API_END
}

int rs_host_alloc(rs_ctx *ctx, size_t bytes, void **h_ptr) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(h_ptr && bytes, "null argument");
  RS_HIP(hipHostMalloc(h_ptr, bytes, hipHostMallocDefault));
  RS_API_END
}
int rs_host_free(rs_ctx *ctx, void *h_ptr) {
  RS_API_BEGIN_CTX(ctx)
  if (h_ptr) RS_HIP(hipHostFree(h_ptr));
  RS_API_END
}

int rs_inner_product(rs_ctx *ctx, const uint64_t *d_encs, const uint64_t *d_rings, const uint8_t *h_kinds, size_t T,
                     uint64_t *d_out, size_t *h_used, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_out && (T == 0 || (d_encs && d_rings)), "null argument");
  WsScope ws_scope(ctx, S(stream));
  rs_msm_vec v{d_rings, h_kinds, T, 0};
  const uint64_t *crs[1] = {d_encs};
  msm_run(ctx, crs, 1, T, &v, 1, 1, d_out, nullptr, h_used, S(stream), 0);
  RS_API_END
}

int rs_enc_mul_ring(rs_ctx *ctx, uint64_t *d_enc, const uint64_t *d_ring, size_t count, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_enc && d_ring, "null argument");
  WsScope ws_scope(ctx, S(stream));
  uint64_t *tmp = (uint64_t *)ws_get(ctx, 4, ctx->enc_words() * sizeof(uint64_t));
  for (size_t k = 0; k < count; k++) {
    rs_msm_vec v{d_ring + k * ctx->ring_words(), nullptr, 1, 0};
    const uint64_t *crs[1] = {d_enc + k * ctx->enc_words()};
    msm_run(ctx, crs, 1, 1, &v, 1, 1, tmp, nullptr, nullptr, S(stream), 0);
    RS_HIP(hipMemcpyAsync(d_enc + k * ctx->enc_words(), tmp, ctx->enc_words() * sizeof(uint64_t),
                          hipMemcpyDeviceToDevice, S(stream)));
  }
  RS_API_END
}

int rs_enc_reduce(rs_ctx *ctx, uint64_t *d_enc, size_t count, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_enc, "null argument");
  const size_t words = count * ctx->enc_words();
  if (words) {
    MsmScratch &sc = scratch_for(ctx);
    const unsigned blocks = (unsigned)std::min<size_t>((words + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(enc_reduce_kernel, dim3(blocks), dim3(256), 0, S(stream), d_enc, words, ctx->N_enc, ctx->K, sc.d_Qint);
    RS_HIP(hipGetLastError());
  }
  RS_API_END
}

int rs_enc_add(rs_ctx *ctx, uint64_t *d_dst, const uint64_t *d_a, const uint64_t *d_b, size_t count, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_dst && d_a && d_b, "null argument");
  enc_add_run(ctx, d_dst, d_a, d_b, count, S(stream));
  RS_API_END
}

}  // extern "C"
