// witness.hip -- r1cs_to_qrp_witness_map, quasi-linear.
//
// The reference interpolates on the domain {0..m-1} with an O(m^2) Lagrange routine
// (util/polynomials.tcc:10-43), multiplies A*B by schoolbook and long-divides by Z
// (util/polynomials.tcc:62-81, util/evaluation_domain.tcc:54-84): ~18 m^2 ring operations.
// Every ring operation is slot-wise, so one ring limb is N independent problems over the prime
// field F_{q_i} ("columns"), and every result is a canonical residue, so ANY exact algorithm is
// bit-identical.  Per column this file computes, with M = next_pow2(m):
//
//   interpolation (values y_j at j = 0..m-1  ->  monomial coefficients):
//     1. Newton (falling-factorial) coefficients by one convolution:
//            f = (y_j / j!) * ((-1)^k / k!)                    [cyclic NTT of length 2M]
//     2. Newton -> monomial by a product tree: node [a, a+n) holds
//            F_node = F_left + D_left * F_right,  D_left = prod_{j in left half}(x - j)
//        levels n <= 8 by schoolbook in registers, larger levels by batched length-n cyclic NTTs
//        against precomputed spectra of D_left.
//   H = (A*B - C) / Z:  evaluate A, B, C on a coset g*<w_M> that avoids the domain, divide
//        pointwise by Z there, transform back (deg H <= m-2 < M).  ZK patch terms
//        (r1cs_to_qrp.tcc:230-235) are added coefficient-wise.
//
// All transforms run inside one workgroup's LDS tile (ntt_core.cuh); data are transposed once
// from the boundary layout [term][limb][slot] to column-major [limb][slot][M] and back.
// Requires q_i = 1 mod 4M (cyclic NTT of length 2M) and M <= 8192 in this round.
#include <algorithm>
#include <cstring>

#include "ntt_core.cuh"
#include "rs_internal.hpp"

namespace rs {

constexpr int SCHOOL_LEVELS = 3;  // tree levels with node size <= 8 use schoolbook products

struct LimbPlan {
  uint64_t p = 0;
  Mod mod{0, 0};
  double *d_tw = nullptr, *d_itw = nullptr;  // cyclic tables, 2M entries
  double *d_invfact = nullptr;               // [M]  1/j! (0 for j >= m)
  double *d_ehat = nullptr;                  // [2M] spectrum of (-1)^k/k!, scaled by 1/(2M)
  double *d_dhat = nullptr;                  // [logM+1][M] spectra of D_left per level, scaled by 1/n
  double *d_dlow = nullptr;                  // [SCHOOL_LEVELS+1][M/2] low coefficients of D_left
  double *d_gpow = nullptr;                  // [M] g^k
  double *d_ginv = nullptr;                  // [M] g^-k / M
  double *d_zinv = nullptr;                  // [M] 1 / Z(g w^i) in transform order
  double *d_ztab = nullptr;                  // [M] Z_k (0 beyond m)
  uint32_t fwd_mask2 = 0, inv_mask2 = 0;     // reduce masks for length 2M
  std::vector<uint64_t> Z;                   // m+1 coefficients of the vanishing polynomial
};

struct WitnessPlan {
  size_t m = 0, M = 0;
  int logM = 0;
  std::vector<LimbPlan> limb;
};

// ---- host-side helpers (integer arithmetic; builds the tables above) -------------------------
namespace hostw {
using namespace host;

struct CycTab {
  uint64_t p;
  std::vector<uint64_t> tw, itw;  // tw[Mg + i] = w_{2Mg}^{bitrev(i)}
};
static CycTab make_cyc(uint64_t p, int logn_max) {
  CycTab t;
  t.p = p;
  const size_t n = (size_t)1 << logn_max;
  t.tw.assign(n, 1);
  t.itw.assign(n, 1);
  const uint64_t wtop = some_primitive_root((uint64_t)n, p);  // primitive n-th root
  for (int lg = 0; (1u << lg) < n; lg++) {
    const size_t Mg = (size_t)1 << lg;  // groups
    // w_{2Mg} = wtop^(n / 2Mg)
    const uint64_t w2 = powmod(wtop, (uint64_t)(n / (2 * Mg)), p);
    std::vector<uint64_t> pw(Mg);
    uint64_t c = 1;
    for (size_t e = 0; e < Mg; e++) {
      pw[e] = c;
      c = mulmod(c, w2, p);
    }
    for (size_t i = 0; i < Mg; i++) {
      const uint64_t v = pw[bitrev((uint32_t)i, lg)];
      t.tw[Mg + i] = v;
      t.itw[Mg + i] = invmod(v, p);
    }
  }
  return t;
}
static void ntt_fwd(std::vector<uint64_t> &a, int logn, const CycTab &t) {
  const size_t n = (size_t)1 << logn;
  const uint64_t p = t.p;
  for (size_t m = 1, gap = n >> 1; m < n; m <<= 1, gap >>= 1)
    for (size_t i = 0; i < m; i++) {
      const uint64_t W = t.tw[m + i];
      for (size_t j = 2 * i * gap; j < 2 * i * gap + gap; j++) {
        const uint64_t u = a[j], v = mulmod(a[j + gap], W, p);
        a[j] = addmod(u, v, p);
        a[j + gap] = submod(u, v, p);
      }
    }
}
static void ntt_inv(std::vector<uint64_t> &a, int logn, const CycTab &t) {
  const size_t n = (size_t)1 << logn;
  const uint64_t p = t.p;
  for (size_t m = n >> 1, gap = 1; m >= 1; m >>= 1, gap <<= 1)
    for (size_t i = 0; i < m; i++) {
      const uint64_t W = t.itw[m + i];
      for (size_t j = 2 * i * gap; j < 2 * i * gap + gap; j++) {
        const uint64_t u = a[j], v = a[j + gap];
        a[j] = addmod(u, v, p);
        a[j + gap] = mulmod(submod(u, v, p), W, p);
      }
    }
  const uint64_t ninv = invmod((uint64_t)n % p, p);
  for (auto &x : a) x = mulmod(x, ninv, p);
}
static int clog2(size_t x) {
  int l = 0;
  while (((size_t)1 << l) < x) l++;
  return l;
}
static std::vector<uint64_t> polymul(const std::vector<uint64_t> &a, const std::vector<uint64_t> &b, const CycTab &t) {
  const size_t need = a.size() + b.size() - 1;
  if (std::min(a.size(), b.size()) <= 16) {
    std::vector<uint64_t> o(need, 0);
    for (size_t i = 0; i < a.size(); i++)
      for (size_t j = 0; j < b.size(); j++) o[i + j] = addmod(o[i + j], mulmod(a[i], b[j], t.p), t.p);
    return o;
  }
  const int lg = clog2(need);
  std::vector<uint64_t> fa(a), fb(b);
  fa.resize((size_t)1 << lg, 0);
  fb.resize((size_t)1 << lg, 0);
  ntt_fwd(fa, lg, t);
  ntt_fwd(fb, lg, t);
  for (size_t i = 0; i < fa.size(); i++) fa[i] = mulmod(fa[i], fb[i], t.p);
  ntt_inv(fa, lg, t);
  fa.resize(need);
  return fa;
}
}  // namespace hostw

static double *up(const std::vector<double> &h) {
  double *d = nullptr;
  RS_HIP(hipMalloc(&d, std::max<size_t>(1, h.size()) * sizeof(double)));
  if (!h.empty()) RS_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
  return d;
}

static WitnessPlan *build_plan(rs_ctx *ctx, size_t m) {
  using namespace hostw;
  RS_REQUIRE(m >= 1, "need at least one constraint");
  WitnessPlan *P = new WitnessPlan();
  P->m = m;
  P->logM = std::max(1, clog2(m));
  P->M = (size_t)1 << P->logM;
  const size_t M = P->M;
  const int logM = P->logM;
  if (M > 8192)
    throw Error(RS_ERR_UNSUPPORTED, "witness map for more than 8192 constraints needs the multi-pass column transform, not built in this round");
  P->limb.resize(ctx->L);
  for (int li = 0; li < ctx->L; li++) {
    LimbPlan &lp = P->limb[li];
    const uint64_t p = ctx->q[li];
    if (host::two_adicity(p) < logM + 1)
      throw Error(RS_ERR_UNSUPPORTED, "ring prime lacks the 2-adicity for the quasi-linear witness map (need q = 1 mod 2*next_pow2(m)*2)");
    RS_REQUIRE(p > 2 * M, "ring prime too small for the evaluation domain");
    lp.p = p;
    lp.mod = Mod{(double)p, 1.0 / (double)p};
    const CycTab T = make_cyc(p, logM + 1);
    auto bal = [&](uint64_t v) { return host::balanced(v, p); };
    {
      std::vector<double> tw(2 * M), itw(2 * M);
      for (size_t k = 0; k < 2 * M; k++) tw[k] = bal(T.tw[k]), itw[k] = bal(T.itw[k]);
      lp.d_tw = up(tw);
      lp.d_itw = up(itw);
    }
    lp.fwd_mask2 = fwd_reduce_mask(p, logM + 1);
    lp.inv_mask2 = inv_reduce_mask(p, logM + 1);
    // factorials
    std::vector<uint64_t> fact(M), ifact(M);
    fact[0] = 1;
    for (size_t j = 1; j < M; j++) fact[j] = mulmod(fact[j - 1], (uint64_t)j % p, p);
    ifact[M - 1] = invmod(fact[M - 1], p);
    for (size_t j = M - 1; j > 0; j--) ifact[j - 1] = mulmod(ifact[j], (uint64_t)j % p, p);
    {
      std::vector<double> v(M, 0.0);
      for (size_t j = 0; j < m; j++) v[j] = bal(ifact[j]);
      lp.d_invfact = up(v);
      std::vector<uint64_t> e(2 * M, 0);
      for (size_t k = 0; k < m; k++) e[k] = (k & 1) ? (p - ifact[k]) % p : ifact[k];
      ntt_fwd(e, logM + 1, T);
      const uint64_t s2 = invmod((uint64_t)(2 * M) % p, p);
      std::vector<double> eh(2 * M);
      for (size_t k = 0; k < 2 * M; k++) eh[k] = bal(mulmod(e[k], s2, p));
      lp.d_ehat = up(eh);
    }
    // subproduct tree: prod[l][i] = prod_{j in [i 2^l, (i+1) 2^l)} (x - j), low 2^l coefficients
    std::vector<std::vector<std::vector<uint64_t>>> prod(logM + 1);
    prod[0].resize(M);
    for (size_t i = 0; i < M; i++) prod[0][i] = {(p - (uint64_t)i % p) % p};
    for (int l = 1; l <= logM; l++) {
      const size_t h = (size_t)1 << (l - 1);
      prod[l].resize(M >> l);
      for (size_t i = 0; i < (M >> l); i++) {
        const auto &a = prod[l - 1][2 * i], &b = prod[l - 1][2 * i + 1];
        std::vector<uint64_t> ab = polymul(a, b, T);  // degree <= 2h-2
        std::vector<uint64_t> r(2 * h, 0);
        for (size_t k = 0; k < ab.size(); k++) r[k] = ab[k];
        for (size_t k = 0; k < h; k++) r[h + k] = addmod(r[h + k], addmod(a[k], b[k], p), p);
        prod[l][i] = r;
      }
    }
    // D_left spectra (levels > SCHOOL_LEVELS) and low coefficients (levels <= SCHOOL_LEVELS)
    {
      std::vector<double> dhat((size_t)(logM + 1) * M, 0.0), dlow((size_t)(SCHOOL_LEVELS + 1) * (M / 2 + 1), 0.0);
      for (int l = 1; l <= logM; l++) {
        const size_t n = (size_t)1 << l, h = n >> 1;
        for (size_t i = 0; i < (M >> l); i++) {
          const auto &dl = prod[l - 1][2 * i];  // h low coefficients, monic of degree h
          if (l <= SCHOOL_LEVELS) {
            for (size_t k = 0; k < h; k++) dlow[(size_t)l * (M / 2 + 1) + i * h + k] = bal(dl[k]);
          } else {
            std::vector<uint64_t> f(n, 0);
            for (size_t k = 0; k < h; k++) f[k] = dl[k];
            f[h] = 1;
            ntt_fwd(f, l, T);
            const uint64_t sc = invmod((uint64_t)n % p, p);
            for (size_t k = 0; k < n; k++) dhat[(size_t)l * M + i * n + k] = bal(mulmod(f[k], sc, p));
          }
        }
      }
      lp.d_dhat = up(dhat);
      lp.d_dlow = up(dlow);
    }
    // Z = prod_{j<m} (x - j): product of the maximal aligned blocks of [0, m)
    {
      std::vector<uint64_t> Z = {1};
      size_t start = 0;
      for (int l = logM; l >= 0; l--) {
        const size_t len = (size_t)1 << l;
        if (start + len <= m) {
          std::vector<uint64_t> blk = prod[l][start >> l];
          blk.push_back(1);
          Z = polymul(Z, blk, T);
          start += len;
        }
      }
      RS_REQUIRE(Z.size() == m + 1 && start == m, "internal: vanishing polynomial size");
      lp.Z = Z;
      std::vector<double> zt(M, 0.0);
      for (size_t k = 0; k < M && k <= m; k++) zt[k] = bal(Z[k]);
      lp.d_ztab = up(zt);
      // coset generator g with Z(g w^i) != 0 for all i
      bool ok = false;
      // g must avoid {j * w^-i}: every g <= m-1 is itself a domain point (i = 0), so start above it
      for (uint64_t g = (uint64_t)m + 1; g < (uint64_t)m + 1000 && !ok; g++) {
        std::vector<uint64_t> zc(M, 0), gp(M);
        uint64_t c = 1;
        for (size_t k = 0; k < M; k++) {
          gp[k] = c;
          if (k <= m) zc[k] = mulmod(Z[k], c, p);
          c = mulmod(c, g, p);
        }
        const uint64_t gM = c;  // g^M
        ntt_fwd(zc, logM, T);
        if (m == M)
          for (auto &x : zc) x = addmod(x, gM, p);
        ok = true;
        for (auto x : zc) ok = ok && x != 0;
        if (!ok) continue;
        std::vector<double> zi(M), gpow(M), ginv(M);
        const uint64_t gi = invmod(g, p), Minv = invmod((uint64_t)M % p, p);
        uint64_t ci = Minv;
        for (size_t k = 0; k < M; k++) {
          zi[k] = bal(invmod(zc[k], p));
          gpow[k] = bal(gp[k]);
          ginv[k] = bal(ci);
          ci = mulmod(ci, gi, p);
        }
        lp.d_zinv = up(zi);
        lp.d_gpow = up(gpow);
        lp.d_ginv = up(ginv);
      }
      RS_REQUIRE(ok, "internal: no coset generator found");
    }
  }
  return P;
}

static void free_plan(WitnessPlan *P) {
  for (auto &lp : P->limb) {
    double *ptrs[] = {lp.d_tw, lp.d_itw, lp.d_invfact, lp.d_ehat, lp.d_dhat, lp.d_dlow, lp.d_gpow, lp.d_ginv, lp.d_zinv, lp.d_ztab};
    for (double *q : ptrs)
      if (q) (void)hipFree(q);
  }
  delete P;
}

WitnessPlan *get_plan(rs_ctx *ctx, size_t m) {
  auto it = ctx->plans.find(m);
  if (it != ctx->plans.end()) return it->second;
  WitnessPlan *P = build_plan(ctx, m);
  ctx->plans[m] = P;
  return P;
}

// ---- device kernels ----------------------------------------------------------------------------

struct ColPlan {  // per-limb device pointers handed to the column kernels
  Mod mod;
  const double *tw, *itw, *invfact, *ehat, *dhat, *dlow, *gpow, *ginv, *zinv, *ztab;
  uint32_t fwd_mask2, inv_mask2;
  uint32_t fmask[16], imask[16];  // reduce masks for transforms of length 2^l
};
struct ColPlans {
  ColPlan l[RS_MAX_L];
};

// [rows][S] u64 (term-major, S = L*N) -> [S][M] f64 (column-major), rows >= m zero-filled.
__global__ void __launch_bounds__(256) transpose_in_kernel(const uint64_t *__restrict__ src, double *__restrict__ dst,
                                                           size_t m, size_t S, size_t M) {
  __shared__ double tile[32][33];
  const size_t s0 = (size_t)blockIdx.x * 32, r0 = (size_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const size_t r = r0 + k, sl = s0 + tx;
    tile[k][tx] = (r < m && sl < S) ? from_u64(src[r * S + sl]) : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const size_t sl = s0 + k, r = r0 + tx;
    if (sl < S && r < M) dst[sl * M + r] = tile[tx][k];
  }
}
// [S][M] f64 canonical -> [rows][S] u64 for rows < m_out
__global__ void __launch_bounds__(256) transpose_out_kernel(const double *__restrict__ src, uint64_t *__restrict__ dst,
                                                            size_t m_out, size_t S, size_t M) {
  __shared__ double tile[32][33];
  const size_t s0 = (size_t)blockIdx.x * 32, r0 = (size_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const size_t sl = s0 + k, r = r0 + tx;
    tile[k][tx] = (sl < S && r < M) ? src[sl * M + r] : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const size_t r = r0 + k, sl = s0 + tx;
    if (r < m_out && sl < S) dst[r * S + sl] = to_u64(tile[tx][k]);
  }
}

// batched forward / inverse sub-transforms: the tile of 2^logtot entries is 2^(logtot-logn)
// independent length-2^logn cyclic transforms on consecutive blocks (all with root 1).
template <int R>
__device__ __forceinline__ void bfwd_round(double *__restrict__ s, int logtot, int logn, int s0,
                                           const double *__restrict__ tw, const Mod mod, uint32_t red_mask) {
  constexpr int E = 1 << R;
  const int lstep = logn - s0 - R, sstep = 1 << lstep;
  const int ngroups = (1 << logtot) >> R;
  for (int grp = threadIdx.x; grp < ngroups; grp += blockDim.x) {
    const int lo = grp & (sstep - 1), hi_all = grp >> lstep;
    const int hi = hi_all & ((1 << s0) - 1);
    const int base = (hi_all << (logn - s0)) + lo;
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
      const int twbase = (1 << (s0 + k)) + (hi << k);
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
template <int R>
__device__ __forceinline__ void binv_round(double *__restrict__ s, int logtot, int logn, int u0,
                                           const double *__restrict__ itw, const Mod mod, uint32_t red_mask) {
  constexpr int E = 1 << R;
  const int g0 = 1 << u0;
  const int ngroups = (1 << logtot) >> R;
  const int groups_per_blk_log = logn - u0 - R;  // log2 of radix groups per sub-transform (per lo)
  for (int grp = threadIdx.x; grp < ngroups; grp += blockDim.x) {
    const int lo = grp & (g0 - 1), hi_all = grp >> u0;
    const int hi = hi_all & ((1 << groups_per_blk_log) - 1);
    const int base = (hi_all << (u0 + R)) + lo;
    double v[E];
#pragma unroll
    for (int e = 0; e < E; e++) v[e] = s[pidx(base + e * g0)];
#pragma unroll
    for (int k = 0; k < R; k++) {
      if ((red_mask >> (u0 + k)) & 1u) {
#pragma unroll
        for (int e = 0; e < E; e++) v[e] = reduce(v[e], mod);
      }
      const int Mg = (1 << logn) >> (u0 + k + 1);
      const int twbase = Mg + (hi << (R - 1 - k));
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
__device__ __forceinline__ void lds_bntt_fwd(double *s, int logtot, int logn, const double *tw, const Mod mod, uint32_t mask) {
  int st = 0;
  while (st < logn) {
    const int R = pick_radix(logn - st);
    if (R == 3)
      bfwd_round<3>(s, logtot, logn, st, tw, mod, mask);
    else if (R == 2)
      bfwd_round<2>(s, logtot, logn, st, tw, mod, mask);
    else
      bfwd_round<1>(s, logtot, logn, st, tw, mod, mask);
    __syncthreads();
    st += R;
  }
}
__device__ __forceinline__ void lds_bntt_inv(double *s, int logtot, int logn, const double *itw, const Mod mod, uint32_t mask) {
  int st = 0;
  while (st < logn) {
    const int R = pick_radix(logn - st);
    if (R == 3)
      binv_round<3>(s, logtot, logn, st, itw, mod, mask);
    else if (R == 2)
      binv_round<2>(s, logtot, logn, st, itw, mod, mask);
    else
      binv_round<1>(s, logtot, logn, st, itw, mod, mask);
    __syncthreads();
    st += R;
  }
}

// One workgroup per column: values at 0..m-1 (cols[col][0..M)) -> monomial coefficients in place.
// LDS: 2M padded doubles (A = [0,M) current polynomials, B = [M,2M) scratch).
// Column c belongs to limb (c % S) / slots_per_limb (several vectors of S columns are batched).
__global__ void __launch_bounds__(1024)
interp_columns_kernel(double *__restrict__ cols, int logM, unsigned S, unsigned slots_per_limb, ColPlans plans) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  const int M = 1 << logM;
  const size_t col = blockIdx.x;
  const ColPlan &P = plans.l[(col % S) / slots_per_limb];
  const Mod mod = P.mod;
  double *c = cols + col * (size_t)M;
  // 1. g_j = y_j / j!  (zero for j >= m), zero-padded to 2M
  for (int j = threadIdx.x; j < M; j += blockDim.x) {
    s[pidx(j)] = mulmod(c[j], P.invfact[j], mod);
    s[pidx(M + j)] = 0.0;
  }
  __syncthreads();
  lds_ntt_fwd(s, logM + 1, P.tw, 1, mod, P.fwd_mask2);
  for (int j = threadIdx.x; j < 2 * M; j += blockDim.x) s[pidx(j)] = mulmod(reduce(s[pidx(j)], mod), P.ehat[j], mod);
  __syncthreads();
  lds_ntt_inv(s, logM + 1, P.itw, 1, mod, P.inv_mask2);
  // Newton coefficients f_k = s[k], k < m; everything at k >= m is discarded (invfact is zero
  // there only for the INPUT; the convolution tail must be cleared explicitly).
  for (int j = threadIdx.x; j < M; j += blockDim.x) {
    const double inv_nonzero = P.invfact[j];
    s[pidx(j)] = (inv_nonzero != 0.0) ? reduce(s[pidx(j)], mod) : 0.0;
  }
  __syncthreads();
  // 2a. schoolbook levels: one thread per node of size 2^SCHOOL_LEVELS
  {
    const int lv = logM < SCHOOL_LEVELS ? logM : SCHOOL_LEVELS;
    const int nn = 1 << lv;
    const int dstride = M / 2 + 1;
    for (int node = threadIdx.x; node < (M >> lv); node += blockDim.x) {
      double v[1 << SCHOOL_LEVELS];
#pragma unroll
      for (int k = 0; k < (1 << SCHOOL_LEVELS); k++) v[k] = (k < nn) ? s[pidx(node * nn + k)] : 0.0;
#pragma unroll
      for (int l = 1; l <= SCHOOL_LEVELS; l++) {
        if (l > lv) break;
        const int n = 1 << l, h = n >> 1;
        // sub-nodes of this thread's node at level l
#pragma unroll
        for (int sub = 0; sub < ((1 << SCHOOL_LEVELS) >> l); sub++) {
          if (sub * n >= nn) break;
          const int gnode = (node * nn) / n + sub;  // global node index at level l
          const double *dl = P.dlow + (size_t)l * dstride + (size_t)gnode * h;
          double out[1 << SCHOOL_LEVELS];
#pragma unroll
          for (int k = 0; k < n; k++) out[k] = 0.0;
          // D_left * F_right, D_left = x^h + sum dl[a] x^a
#pragma unroll
          for (int b = 0; b < h; b++) {
            const double fr = v[sub * n + h + b];
            out[h + b] += fr;
#pragma unroll
            for (int a = 0; a < h; a++) out[a + b] += mulmod(dl[a], fr, mod);
          }
#pragma unroll
          for (int k = 0; k < n; k++) {
            const double left = (k < h) ? v[sub * n + k] : 0.0;
            v[sub * n + k] = reduce(out[k] + left, mod);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < (1 << SCHOOL_LEVELS); k++)
        if (k < nn) s[pidx(node * nn + k)] = v[k];
    }
    __syncthreads();
  }
  // 2b. transform levels
  double *B = s;  // B region addressed as index M + i through pidx
  for (int l = SCHOOL_LEVELS + 1; l <= logM; l++) {
    const int n = 1 << l, h = n >> 1;
    // B[node] = (F_right, 0)
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
      const int k = i & (n - 1);
      B[pidx(M + i)] = (k < h) ? s[pidx(i + h)] : 0.0;
    }
    __syncthreads();
    // batched length-n transforms over the B half: shift the tile base so indices run 0..M-1
    // (pidx is not shift-invariant, so transforms address B through an offset tile)
    double *Bt = s + pidx(M);  // valid because M is a multiple of 16: pidx(M + i) = pidx(M) + pidx(i)
    lds_bntt_fwd(Bt, logM, l, P.tw, mod, P.fmask[l]);
    const double *dh = P.dhat + (size_t)l * M;
    for (int i = threadIdx.x; i < M; i += blockDim.x) Bt[pidx(i)] = mulmod(reduce(Bt[pidx(i)], mod), dh[i], mod);
    __syncthreads();
    lds_bntt_inv(Bt, logM, l, P.itw, mod, P.imask[l]);
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
      const int k = i & (n - 1);
      const double left = (k < h) ? s[pidx(i)] : 0.0;
      s[pidx(i)] = reduce(Bt[pidx(i)] + left, mod);
    }
    __syncthreads();
  }
  for (int j = threadIdx.x; j < M; j += blockDim.x) c[j] = canon(s[pidx(j)], mod);
}


// H = (A*B - C)/Z per column + the ZK patch of r1cs_to_qrp.tcc:230-235.  A, B, C, H: [cols][M]
// canonical doubles.  d1,d2,d3: ring elements [L][N] (u64) or NULL.
__global__ void __launch_bounds__(1024)
h_columns_kernel(const double *__restrict__ A, const double *__restrict__ Bc, const double *__restrict__ Cc,
                 double *__restrict__ H, int logM, unsigned slots_per_limb, ColPlans plans,
                 const uint64_t *__restrict__ d1, const uint64_t *__restrict__ d2, const uint64_t *__restrict__ d3) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  const int M = 1 << logM;
  const size_t col = blockIdx.x;
  const ColPlan &P = plans.l[col / slots_per_limb];
  const Mod mod = P.mod;
  const double *srcs[3] = {A + col * (size_t)M, Bc + col * (size_t)M, Cc + col * (size_t)M};
  double r[8];
#pragma unroll
  for (int pass = 0; pass < 3; pass++) {
    for (int k = threadIdx.x; k < M; k += blockDim.x) s[pidx(k)] = mulmod(srcs[pass][k], P.gpow[k], mod);
    __syncthreads();
    lds_ntt_fwd(s, logM, P.tw, 1, mod, P.fmask[logM]);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int p = threadIdx.x + k * blockDim.x;
      if (p < M) {
        const double v = reduce(s[pidx(p)], mod);
        if (pass == 0)
          r[k] = v;
        else if (pass == 1)
          r[k] = mulmod(r[k], v, mod);
        else
          r[k] = mulmod(reduce(r[k] - v, mod), P.zinv[p], mod);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int p = threadIdx.x + k * blockDim.x;
    if (p < M) s[pidx(p)] = r[k];
  }
  __syncthreads();
  lds_ntt_inv(s, logM, P.itw, 1, mod, P.imask[logM]);
  double e1 = 0.0, e2 = 0.0, e3 = 0.0, e12 = 0.0;
  const bool zk = d1 != nullptr;
  if (zk) {
    e1 = center(from_u64(d1[col]), mod);
    e2 = center(from_u64(d2[col]), mod);
    e3 = center(from_u64(d3[col]), mod);
    e12 = mulmod(e1, e2, mod);
  }
  double *dst = H + col * (size_t)M;
  for (int k = threadIdx.x; k < M; k += blockDim.x) {
    double h = mulmod(reduce(s[pidx(k)], mod), P.ginv[k], mod);
    if (zk) {
      h += mulmod(e2, center(srcs[0][k], mod), mod) + mulmod(e1, center(srcs[1][k], mod), mod) +
           mulmod(e12, P.ztab[k], mod);
      if (k == 0) h -= e3;
    }
    dst[k] = canon(h, mod);
  }
}

// coefficients_for_X_mid = interp(full) - interp(io) + interp(constant part), in place over `full`.
// (The reference evaluates index-0 terms in BOTH the io and the mid pass, r1cs_to_qrp.tcc:175-201.)
__global__ void __launch_bounds__(256)
mid_kernel(double *__restrict__ full, const double *__restrict__ io, const double *__restrict__ cst /* [L][M] or null */,
           size_t M, size_t S, unsigned slots_per_limb, ColPlans plans) {
  const size_t total = S * M, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t col = i / M, k = i % M;
    const int limb = (int)(col / slots_per_limb);
    const Mod mod = plans.l[limb].mod;
    double v = full[i] - io[i];
    if (cst) v += cst[(size_t)limb * M + k];
    full[i] = canon(v, mod);
  }
}

// a14: linear_combination::evaluate for every constraint (relations/variable.tcc:246-254).
// grid (m, ceil(L*N/512)); each thread handles two adjacent slots.
__global__ void __launch_bounds__(256)
r1cs_eval_kernel(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col, const double *__restrict__ coeff,
                 size_t nnz, const uint64_t *__restrict__ asg, uint64_t *__restrict__ out, int N, int L, int mode,
                 unsigned n_inputs, const Mod *__restrict__ qmod) {
  const size_t row = blockIdx.x;
  const size_t S = (size_t)L * N;
  const size_t pair = (size_t)blockIdx.y * blockDim.x + threadIdx.x;
  if (2 * pair >= S) return;
  const int limb = (int)((2 * pair) / (size_t)N);
  const Mod mod = qmod[limb];
  double a0 = 0.0, a1 = 0.0;
  int since = 0;
  for (uint32_t e = row_ptr[row]; e < row_ptr[row + 1]; e++) {
    const uint32_t c = col[e];
    const double cf = coeff[(size_t)limb * nnz + e];
    if (c == 0) {
      a0 += cf;
      a1 += cf;
    } else {
      const bool is_input = (c - 1) < n_inputs;
      if ((mode == RS_EVAL_IO && !is_input) || (mode == RS_EVAL_MID && is_input)) continue;
      const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(asg + (size_t)(c - 1) * S)[pair];
      a0 += mulmod(from_u64(v.x), cf, mod);
      a1 += mulmod(from_u64(v.y), cf, mod);
    }
    if (++since == 4) {
      since = 0;
      a0 = reduce(a0, mod);
      a1 = reduce(a1, mod);
    }
  }
  ulonglong2 o;
  o.x = to_u64(canon(a0, mod));
  o.y = to_u64(canon(a1, mod));
  reinterpret_cast<ulonglong2 *>(out + row * S)[pair] = o;
}

static ColPlans make_colplans(rs_ctx *ctx, const WitnessPlan *P) {
  ColPlans cp;
  memset(&cp, 0, sizeof(cp));
  for (int i = 0; i < ctx->L; i++) {
    const LimbPlan &lp = P->limb[i];
    ColPlan &c = cp.l[i];
    c.mod = lp.mod;
    c.tw = lp.d_tw;
    c.itw = lp.d_itw;
    c.invfact = lp.d_invfact;
    c.ehat = lp.d_ehat;
    c.dhat = lp.d_dhat;
    c.dlow = lp.d_dlow;
    c.gpow = lp.d_gpow;
    c.ginv = lp.d_ginv;
    c.zinv = lp.d_zinv;
    c.ztab = lp.d_ztab;
    c.fwd_mask2 = lp.fwd_mask2;
    c.inv_mask2 = lp.inv_mask2;
    for (int l = 0; l < 16; l++) {
      c.fmask[l] = fwd_reduce_mask(lp.p, l);
      c.imask[l] = inv_reduce_mask(lp.p, l);
    }
  }
  return cp;
}

static int col_threads(size_t M) { return (int)std::max<size_t>(64, std::min<size_t>(1024, M / 8)); }

static void launch_interp(rs_ctx *ctx, const WitnessPlan *P, const ColPlans &cp, double *cols, size_t ncols, size_t S,
                          size_t slots_per_limb, hipStream_t st) {
  (void)ctx;
  const size_t lds = padded_len(2 * P->M) * sizeof(double);
  RS_HIP(hipFuncSetAttribute((const void *)interp_columns_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(interp_columns_kernel, dim3((unsigned)ncols), dim3(col_threads(2 * P->M)), lds, st, cols, P->logM,
                     (unsigned)S, (unsigned)slots_per_limb, cp);
  RS_HIP(hipGetLastError());
}

void r1cs_evaluate_run(rs_ctx *ctx, const rs_r1cs *cs, int which, int mode, const uint64_t *d_asg, uint64_t *d_out,
                       hipStream_t st) {
  const size_t S = ctx->ring_words();
  const unsigned by = (unsigned)((S / 2 + 255) / 256);
  hipLaunchKernelGGL(r1cs_eval_kernel, dim3((unsigned)cs->m, by), dim3(256), 0, st, cs->d_row_ptr[which], cs->d_col[which],
                     cs->d_coeff[which], cs->nnz[which], d_asg, d_out, ctx->N, ctx->L, mode, (unsigned)cs->n_inputs,
                     ctx->d_qmod);
  RS_HIP(hipGetLastError());
}

// Witness map driver.  outs[k] (k = A_io,B_io,C_io,A_mid,B_mid,C_mid,H) may be null.
void witness_run(rs_ctx *ctx, const rs_r1cs *cs, const uint64_t *d_asg, const uint64_t *d1, const uint64_t *d2,
                 const uint64_t *d3, uint64_t *const outs[7], uint64_t *h_Z, hipStream_t st) {
  RS_REQUIRE((d1 && d2 && d3) || (!d1 && !d2 && !d3), "d1,d2,d3 must be all set or all null");
  const size_t m = cs->m;
  WitnessPlan *P = get_plan(ctx, m);
  const ColPlans cp = make_colplans(ctx, P);
  const size_t M = P->M, S = ctx->ring_words(), N = (size_t)ctx->N;
  if (h_Z)
    for (int i = 0; i < ctx->L; i++) memcpy(h_Z + (size_t)i * (m + 1), P->limb[i].Z.data(), sizeof(uint64_t) * (m + 1));
  const bool needH = outs[6] != nullptr;
  bool need_io[3], need_full[3];
  for (int w = 0; w < 3; w++) {
    need_io[w] = outs[w] != nullptr || outs[3 + w] != nullptr;
    need_full[w] = outs[3 + w] != nullptr || needH;
  }
  // column-major workspace: slots 0..2 = io, 3..5 = full, 6 = H
  const size_t vec = S * M;
  double *colbuf = (double *)ws_get(ctx, 5, 7 * vec * sizeof(double));
  uint64_t *evalbuf = (uint64_t *)ws_get(ctx, 6, std::max<size_t>(m, 1) * S * sizeof(uint64_t));
  auto colv = [&](int k) { return colbuf + (size_t)k * vec; };
  const dim3 tgrid((unsigned)((S + 31) / 32), (unsigned)((M + 31) / 32));
  for (int w = 0; w < 3; w++) {
    if (need_io[w]) {
      r1cs_evaluate_run(ctx, cs, w, RS_EVAL_IO, d_asg, evalbuf, st);
      hipLaunchKernelGGL(transpose_in_kernel, tgrid, dim3(256), 0, st, evalbuf, colv(w), m, S, M);
    }
    if (need_full[w]) {
      r1cs_evaluate_run(ctx, cs, w, RS_EVAL_FULL, d_asg, evalbuf, st);
      hipLaunchKernelGGL(transpose_in_kernel, tgrid, dim3(256), 0, st, evalbuf, colv(3 + w), m, S, M);
    }
  }
  RS_HIP(hipGetLastError());
  // one batched interpolation over every needed vector (skipped vectors cost nothing but are
  // laid out contiguously, so launch per contiguous run)
  for (int k = 0; k < 6; k++) {
    const bool need = k < 3 ? need_io[k] : need_full[k - 3];
    if (!need) continue;
    int e = k;
    while (e + 1 < 6 && (e + 1 < 3 ? need_io[e + 1] : need_full[e + 1 - 3])) e++;
    launch_interp(ctx, P, cp, colv(k), (size_t)(e - k + 1) * S, S, N, st);
    k = e;
  }
  if (needH) {
    const size_t lds = padded_len(M) * sizeof(double);
    RS_HIP(hipFuncSetAttribute((const void *)h_columns_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(h_columns_kernel, dim3((unsigned)S), dim3(col_threads(M)), lds, st, colv(3), colv(4), colv(5),
                       colv(6), P->logM, (unsigned)N, cp, d1, d2, d3);
    RS_HIP(hipGetLastError());
  }
  // constant-term correction for the mid vectors
  double *d_const = nullptr;
  bool any_const = false;
  for (int w = 0; w < 3; w++) any_const = any_const || cs->has_const[w];
  if (any_const) {
    // [3][L][M] values -> interpolate as 3*L single columns
    std::vector<double> hc((size_t)3 * ctx->L * M, 0.0);
    for (int w = 0; w < 3; w++)
      for (int i = 0; i < ctx->L; i++)
        for (size_t r = 0; r < m; r++) hc[((size_t)w * ctx->L + i) * M + r] = (double)cs->h_const[w][(size_t)i * m + r];
    d_const = (double *)ws_get(ctx, 4, hc.size() * sizeof(double));
    RS_HIP(hipMemcpyAsync(d_const, hc.data(), hc.size() * sizeof(double), hipMemcpyHostToDevice, st));
    RS_HIP(hipStreamSynchronize(st));  // hc goes out of scope
    launch_interp(ctx, P, cp, d_const, (size_t)3 * ctx->L, (size_t)ctx->L, 1, st);
  }
  const unsigned eb = (unsigned)std::min<size_t>((vec + 255) / 256, 256 * 16);
  for (int w = 0; w < 3; w++) {
    if (!outs[3 + w]) continue;
    const double *cst = (d_const && cs->has_const[w]) ? d_const + (size_t)w * ctx->L * M : nullptr;
    hipLaunchKernelGGL(mid_kernel, dim3(eb), dim3(256), 0, st, colv(3 + w), colv(w), cst, M, S, (unsigned)N, cp);
  }
  RS_HIP(hipGetLastError());
  for (int k = 0; k < 7; k++) {
    if (!outs[k]) continue;
    const size_t rows = std::min(m, M);  // H row m (only when m == M) is written below
    hipLaunchKernelGGL(transpose_out_kernel, tgrid, dim3(256), 0, st, colv(k), outs[k], k == 6 ? std::min(m + 1, M) : rows, S, M);
  }
  RS_HIP(hipGetLastError());
  if (needH && m == M) {  // H[m] = d1*d2*Z[m] = d1*d2 (Z monic), zero without ZK
    uint64_t *top = outs[6] + m * S;
    if (d1)
      RS_REQUIRE(rs_ring_mul(ctx, top, d1, d2, 1, (rs_stream)st) == RS_OK, rs_last_error());
    else
      RS_HIP(hipMemsetAsync(top, 0, S * sizeof(uint64_t), st));
  }
}

}  // namespace rs

using namespace rs;

extern "C" {

void rs_witness_plans_destroy(rs_ctx *ctx) {
  for (auto &kv : ctx->plans) free_plan(kv.second);
  ctx->plans.clear();
}

int rs_r1cs_create(rs_ctx *ctx, size_t m, size_t n_vars, size_t n_inputs, const uint32_t *const h_row_ptr[3],
                   const uint32_t *const h_col[3], const uint64_t *const h_coeff[3], const size_t nnz[3], rs_r1cs **out) {
  RS_API_BEGIN
  RS_REQUIRE(ctx && out && h_row_ptr && h_col && h_coeff && nnz, "null argument");
  RS_REQUIRE(m >= 1 && n_inputs <= n_vars, "bad R1CS shape");
  rs_r1cs *cs = new rs_r1cs();
  cs->m = m;
  cs->n_vars = n_vars;
  cs->n_inputs = n_inputs;
  cs->L = ctx->L;
  for (int w = 0; w < 3; w++) {
    const size_t z = nnz[w];
    cs->nnz[w] = z;
    RS_REQUIRE(h_row_ptr[w][0] == 0 && h_row_ptr[w][m] == z, "row_ptr does not match nnz");
    cs->h_const[w].assign((size_t)ctx->L * m, 0);
    cs->has_const[w] = false;
    std::vector<double> cf((size_t)ctx->L * std::max<size_t>(z, 1), 0.0);
    for (size_t r = 0; r < m; r++)
      for (uint32_t e = h_row_ptr[w][r]; e < h_row_ptr[w][r + 1]; e++) {
        RS_REQUIRE(h_col[w][e] <= n_vars, "column index out of range");
        for (int i = 0; i < ctx->L; i++) {
          const uint64_t c = h_coeff[w][(size_t)i * z + e] % ctx->q[i];
          cf[(size_t)i * z + e] = host::balanced(c, ctx->q[i]);
          if (h_col[w][e] == 0) {
            cs->h_const[w][(size_t)i * m + r] = host::addmod(cs->h_const[w][(size_t)i * m + r], c, ctx->q[i]);
            if (c) cs->has_const[w] = true;
          }
        }
      }
    RS_HIP(hipMalloc(&cs->d_row_ptr[w], sizeof(uint32_t) * (m + 1)));
    RS_HIP(hipMemcpy(cs->d_row_ptr[w], h_row_ptr[w], sizeof(uint32_t) * (m + 1), hipMemcpyHostToDevice));
    RS_HIP(hipMalloc(&cs->d_col[w], sizeof(uint32_t) * std::max<size_t>(z, 1)));
    if (z) RS_HIP(hipMemcpy(cs->d_col[w], h_col[w], sizeof(uint32_t) * z, hipMemcpyHostToDevice));
    RS_HIP(hipMalloc(&cs->d_coeff[w], sizeof(double) * cf.size()));
    RS_HIP(hipMemcpy(cs->d_coeff[w], cf.data(), sizeof(double) * cf.size(), hipMemcpyHostToDevice));
  }
  *out = cs;
  RS_API_END
}

void rs_r1cs_destroy(rs_r1cs *cs) {
  if (!cs) return;
  for (int w = 0; w < 3; w++) {
    if (cs->d_row_ptr[w]) (void)hipFree(cs->d_row_ptr[w]);
    if (cs->d_col[w]) (void)hipFree(cs->d_col[w]);
    if (cs->d_coeff[w]) (void)hipFree(cs->d_coeff[w]);
  }
  delete cs;
}

int rs_r1cs_evaluate(rs_ctx *ctx, const rs_r1cs *cs, int which, int mode, const uint64_t *d_assignment, uint64_t *d_out,
                     rs_stream stream) {
  RS_API_BEGIN
  RS_REQUIRE(ctx && cs && d_assignment && d_out, "null argument");
  RS_REQUIRE(which >= 0 && which < 3 && mode >= 0 && mode <= 2, "bad selector");
  r1cs_evaluate_run(ctx, cs, which, mode, d_assignment, d_out, S(stream));
  RS_API_END
}

int rs_witness_map(rs_ctx *ctx, const rs_r1cs *cs, const uint64_t *d_assignment, const uint64_t *d_d1,
                   const uint64_t *d_d2, const uint64_t *d_d3, uint64_t *d_A_io, uint64_t *d_B_io, uint64_t *d_C_io,
                   uint64_t *d_A_mid, uint64_t *d_B_mid, uint64_t *d_C_mid, uint64_t *d_H, uint64_t *h_Z,
                   rs_stream stream) {
  RS_API_BEGIN
  RS_REQUIRE(ctx && cs && d_assignment, "null argument");
  std::lock_guard<std::mutex> lk(ctx->mu);
  uint64_t *outs[7] = {d_A_io, d_B_io, d_C_io, d_A_mid, d_B_mid, d_C_mid, d_H};
  witness_run(ctx, cs, d_assignment, d_d1, d_d2, d_d3, outs, h_Z, S(stream));
  RS_API_END
}

int rs_interpolate(rs_ctx *ctx, const uint64_t *d_y, uint64_t *d_out, size_t n, rs_stream stream) {
  RS_API_BEGIN
  RS_REQUIRE(ctx && d_y && d_out && n >= 1, "null argument");
  std::lock_guard<std::mutex> lk(ctx->mu);
  WitnessPlan *P = get_plan(ctx, n);
  const ColPlans cp = make_colplans(ctx, P);
  const size_t M = P->M, S_ = ctx->ring_words();
  double *colbuf = (double *)ws_get(ctx, 5, S_ * M * sizeof(double));
  const dim3 tgrid((unsigned)((S_ + 31) / 32), (unsigned)((M + 31) / 32));
  hipLaunchKernelGGL(transpose_in_kernel, tgrid, dim3(256), 0, S(stream), d_y, colbuf, n, S_, M);
  launch_interp(ctx, P, cp, colbuf, S_, S_, (size_t)ctx->N, S(stream));
  hipLaunchKernelGGL(transpose_out_kernel, tgrid, dim3(256), 0, S(stream), colbuf, d_out, n, S_, M);
  RS_HIP(hipGetLastError());
  RS_API_END
}

}  // extern "C"
