// witness.hip -- r1cs_to_qrp_witness_map, quasi-linear: plans, launch orchestration, C ABI.
//
// The reference interpolates on the domain {0..m-1} with an O(m^2) Lagrange routine
// (util/polynomials.tcc:10-43), multiplies A*B by schoolbook and long-divides by Z
// (util/polynomials.tcc:62-81, util/evaluation_domain.tcc:54-84): ~18 m^2 ring operations.
// Every ring operation is slot-wise, so one ring limb is N independent problems over the prime
// field F_{q_i} ("columns"), and every result is a canonical residue, so ANY exact algorithm is
// bit-identical.  Per column, with M = next_pow2(m):
//
//   interpolation (values y_j at j = 0..m-1  ->  monomial coefficients):
//     1. Newton (falling-factorial) coefficients by one convolution:
//            f = (y_j / j!) * ((-1)^k / k!)                    [cyclic, length 2M]
//     2. Newton -> monomial by a product tree: node [a, a+n) holds
//            F_node = F_left + D_left * F_right,  D_left = prod_{j in left half}(x - j)
//        levels n <= 16 by schoolbook in registers, larger levels by batched length-n cyclic transforms
//        against precomputed spectra of D_left.
//   H = quo(A*B, Z) (= quo(A*B - C, Z): deg C < deg Z) through rev(H) = rev(A*B) * rev(Z)^-1 mod x^(m-1), the power
//        series rev(Z)^-1 precomputed per (prime, m).  ZK patch terms (r1cs_to_qrp.tcc:230-235) added coefficient-wise.
//
// Where the kernels live (DESIGN.md section 3 describes each):
//   witness_cols.hpp        column plans (per-limb table pointers), ColMap, layout transposes
//   witness_tiles.hpp       M <= 2^14: one column = one LDS tile (tree_columns_kernel, h_tile_kernel, generic fallbacks)
//   witness_tree_wide.hpp   tree_wide_kernel<13 | 14>: the product tree's tiles of the multi-pass path
//   witness_multipass.hpp   M > 2^14: cross passes over global memory + rooted 2^13 sub-transforms (sub_ntt_wide_kernel)
//   witness_bc.hpp          ring primes without a 2M-th root of unity: pairwise and two-dimensional block convolutions
//   witness_eval.hpp        a14 (linear_combination::evaluate) into columns, io vectors, io / mid output
// This file: the per-(context, m) plan and its host-side table construction, the launch sequences of the three paths
// (single tile, multi-pass, block convolutions), the chunking of columns, and the extern "C" entry points.
#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <type_traits>

#include "ntt_core.hpp"
#include "ntt_wide.hpp"
#include "rs_internal.hpp"
#include "witness_cols.hpp"
#include "witness_tiles.hpp"
#include "witness_tree_wide.hpp"
#include "witness_eval.hpp"
#include "witness_multipass.hpp"
#include "witness_bc.hpp"

namespace rs {


// Device tables of one limb: arrays of 8-byte TABLE CONSTANTS of the context's arithmetic (balanced doubles for
// the FP64 arithmetic, Montgomery-form integers for the integer one; the zero constant is the zero word in both).
struct LimbPlan {
  uint64_t p = 0;
  void *d_tw = nullptr, *d_itw = nullptr;  // cyclic tables, 2M entries
  void *d_invfact = nullptr;               // [M]  1/j! (0 for j >= m)
  void *d_ehat = nullptr;                  // [2M] spectrum of (-1)^k/k!, scaled by 1/(2M)
  void *d_dhat = nullptr;                  // [logM+1][M] spectra of D_left per level, scaled by 1/n
  void *d_dlow = nullptr;                  // [SCHOOL_LEVELS+1][M/2] low coefficients of D_left
  void *d_shat = nullptr;                  // [2M] spectrum of rev(Z)^-1 mod x^(m-1), scaled 1/(2M)^2
  void *d_ztab = nullptr;                  // [M] Z_k (0 beyond m)
  // coset form of H (big_h_coset; full-length plans): g^k; g^-k / M; 1 / Z(g w^i) in the forward transform's output order
  void *d_cos_g = nullptr, *d_cos_h = nullptr, *d_cos_z = nullptr;  // [M] each
  // block-convolution path (WitnessPlan::bcLog != 0): spectra of the B-coefficient blocks of the same polynomials,
  // transform length 2B = 2^bcLog, scaled by 1/(2B)
  void *d_bc_e = nullptr;                  // [M/B][2B] blocks of (-1)^k/k!
  void *d_bc_s = nullptr;                  // [M/B][2B] blocks of rev(Z)^-1 mod x^(m-1)
  void *d_bc_d = nullptr;                  // [logM - bcLog][M] per level l > bcLog: [node][block][2B] blocks of D_left's low part
  // two-dimensional form of the same tables (WitnessPlan::bc2): per spectrum point, the Y-point transform ACROSS the
  // zero-padded sequence of blocks (Y = 2 x blocks of the operand), scaled by 1/(2B Y):
  void *d_b2_e = nullptr, *d_b2_s = nullptr;  // [Y][2B], Y = 2M/B
  void *d_b2_d = nullptr;                     // [logM - bcLog][2M]: per level l, [node][Y_l][2B], Y_l = 2^l / B
  uint32_t fwd_mask2 = 0, inv_mask2 = 0;     // reduce masks for length 2M
  int adic = 64;                             // incomplete transforms (WitnessPlan::incomplete): d_tw / d_itw hold 2^adic entries and
                                             // every spectrum table of a longer transform is in the incomplete form (witness_inc.hpp)
  std::vector<uint64_t> Z;                   // m+1 coefficients of the vanishing polynomial
};

struct WitnessPlan {
  size_t m = 0, M = 0;
  int logM = 0;
  // 0: every ring prime has a 2M-th root of unity (q = 1 mod 2M): full-length transforms.  Otherwise the largest
  // transform length every prime supports is 2^bcLog < 2M (capped at 2^13, one LDS tile) and every product longer
  // than that is a BLOCK convolution over blocks of B = 2^(bcLog-1) coefficients (see "block convolutions" below):
  // what makes the witness map work for the primes the reference's own recipe produces, which only guarantee
  // q = 1 mod 2*N_inner (seal/seal_util.hpp:20-32).
  int bcLog = 0;
  // Block convolutions as TWO-DIMENSIONAL transforms (FP64 arithmetic, primes with 2-adicity >= 14, M >= 2^15; see
  // "two-dimensional block convolutions" below): blocks of B = 2^13 coefficients, bcLog = 14.
  bool bc2 = false;
  // Some ring prime lacks a 2M-th root of unity and the columns take the multi-pass path with INCOMPLETE transforms
  // (witness_inc.hpp; LimbPlan::adic per prime) instead of block convolutions: bcLog = 0, the full-length launch sequences run.
  bool incomplete = false;
  uint64_t knob_sig = 0;  // plan_knob_sig() when the plan was built: get_plan rebuilds when a knob it depends on has changed
  std::vector<LimbPlan> limb;
  // coefficients_for_Z of every limb as the compact [m + 1][L] device array the inner products take a slot-constant
  // vector in (rs_msm_vec::slot_const): a per-(context, m) constant, uploaded once (witness_Z_rows)
  uint64_t *d_Zt = nullptr;
};

// ---- host-side helpers (integer arithmetic; builds the tables above) -------------------------
namespace hostw {
using namespace host;

struct CycTab {
  uint64_t p;
  int logmax;                     // transforms up to length 2^logmax
  std::vector<uint64_t> tw, itw;  // tw[Mg + i] = w_{2Mg}^{bitrev(i)}
};
static CycTab make_cyc(uint64_t p, int logn_max) {
  CycTab t;
  t.p = p;
  t.logmax = logn_max;
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
// nst >= 0: the first nst stages only (incomplete transforms, witness_inc.hpp: leaves of 2^(logn - nst) consecutive words)
static void ntt_fwd(std::vector<uint64_t> &a, int logn, const CycTab &t, int nst = -1) {
  const size_t n = (size_t)1 << logn;
  const uint64_t p = t.p;
  const size_t mend = nst < 0 ? n : (size_t)1 << nst;
  for (size_t m = 1, gap = n >> 1; m < mend; m <<= 1, gap >>= 1)
    for (size_t i = 0; i < m; i++) {
      const uint64_t W = t.tw[m + i];
      for (size_t j = 2 * i * gap; j < 2 * i * gap + gap; j++) {
        const uint64_t u = a[j], v = mulmod(a[j + gap], W, p);
        a[j] = addmod(u, v, p);
        a[j + gap] = submod(u, v, p);
      }
    }
}
// u0 > 0: the inverse of an incomplete transform -- stages u0 .. logn-1, scaled by 2^-(logn - u0)
static void ntt_inv(std::vector<uint64_t> &a, int logn, const CycTab &t, int u0 = 0) {
  const size_t n = (size_t)1 << logn;
  const uint64_t p = t.p;
  for (size_t m = n >> (u0 + 1), gap = (size_t)1 << u0; m >= 1; m >>= 1, gap <<= 1)
    for (size_t i = 0; i < m; i++) {
      const uint64_t W = t.itw[m + i];
      for (size_t j = 2 * i * gap; j < 2 * i * gap + gap; j++) {
        const uint64_t u = a[j], v = a[j + gap];
        a[j] = addmod(u, v, p);
        a[j + gap] = mulmod(submod(u, v, p), W, p);
      }
    }
  const uint64_t ninv = invmod((uint64_t)(n >> u0) % p, p);
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
  if (lg > t.logmax && lg - t.logmax <= 4) {
    // the prime has no root of unity of that order: incomplete transforms (witness_inc.hpp) -- the first logmax stages, then
    // the product of the residues modulo x^G - eta per leaf, G = 2^(lg - logmax)
    const int inc = lg - t.logmax, nst = t.logmax;
    const size_t G = (size_t)1 << inc, n = (size_t)1 << lg;
    std::vector<uint64_t> fa(a), fb(b), out(n);
    fa.resize(n, 0);
    fb.resize(n, 0);
    ntt_fwd(fa, lg, t, nst);
    ntt_fwd(fb, lg, t, nst);
    for (size_t g = 0; g < (n >> inc); g++) {
      const uint64_t w = t.tw[(((size_t)1 << nst) + g) >> 1], eta = (g & 1) ? (t.p - w) % t.p : w;
      const uint64_t *x = &fa[g * G], *y = &fb[g * G];
      for (size_t k = 0; k < G; k++) {
        uint64_t lo = 0, hi = 0;
        for (size_t i = 0; i < G; i++) {
          const uint64_t pr = mulmod(x[i], y[(k - i) & (G - 1)], t.p);
          if (i <= k) lo = addmod(lo, pr, t.p);
          else hi = addmod(hi, pr, t.p);
        }
        out[g * G + k] = addmod(lo, mulmod(hi, eta, t.p), t.p);
      }
    }
    ntt_inv(out, lg, t, inc);
    out.resize(need);
    return out;
  }
  if (lg > t.logmax) {
    // ... more than four stages short: block convolution over blocks of Bh = 2^(logmax-1)
    // coefficients (each block product fits one transform of length 2 Bh), overlap-added
    const size_t Bh = (size_t)1 << (t.logmax - 1);
    const size_t nab = (a.size() + Bh - 1) / Bh, nbb = (b.size() + Bh - 1) / Bh;
    auto spectra = [&](const std::vector<uint64_t> &x, size_t nb) {
      std::vector<std::vector<uint64_t>> sp(nb);
      for (size_t i = 0; i < nb; i++) {
        sp[i].assign(2 * Bh, 0);
        for (size_t k = 0; k < Bh && i * Bh + k < x.size(); k++) sp[i][k] = x[i * Bh + k];
        ntt_fwd(sp[i], t.logmax, t);
      }
      return sp;
    };
    const auto sa = spectra(a, nab), sb = spectra(b, nbb);
    std::vector<uint64_t> o(need + 2 * Bh, 0);
    for (size_t k = 0; k + 1 < nab + nbb; k++) {
      std::vector<uint64_t> acc(2 * Bh, 0);
      for (size_t i = (k >= nbb ? k - nbb + 1 : 0); i <= k && i < nab; i++)
        for (size_t x = 0; x < 2 * Bh; x++) acc[x] = addmod(acc[x], mulmod(sa[i][x], sb[k - i][x], t.p), t.p);
      ntt_inv(acc, t.logmax, t);
      for (size_t x = 0; x < 2 * Bh; x++) o[k * Bh + x] = addmod(o[k * Bh + x], acc[x], t.p);
    }
    o.resize(need);
    return o;
  }
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

static void *up(const std::vector<uint64_t> &h) {
  void *d = nullptr;
  RS_HIP(hipMalloc(&d, std::max<size_t>(1, h.size()) * sizeof(uint64_t)));
  if (!h.empty()) RS_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
  return d;
}
// the 8-byte word of a table constant / of a data value in the context's arithmetic
static uint64_t word_of(double d) {
  uint64_t u;
  memcpy(&u, &d, 8);
  return u;
}
static uint64_t word_of(uint64_t u) { return u; }
static uint64_t konst_word(const rs_ctx *ctx, uint64_t v, uint64_t p) {
  return ctx->use_int ? word_of(HostArith<ModI>::konst(v, p)) : word_of(HostArith<Mod>::konst(v, p));
}
static uint64_t plain_word(const rs_ctx *ctx, uint64_t v, uint64_t p) {
  return ctx->use_int ? word_of(HostArith<ModI>::plain(v, p)) : word_of(HostArith<Mod>::plain(v, p));
}

int g_witness_h_coset = 1;      // tuning knob "witness_h_coset": H on a coset (four length-M transforms) when the call interpolates C; 0: always big_h
int g_witness_sub_log = 12;     // tuning knob "witness_sub_log": 12 = rooted sub-transforms on blocks of 2^12 (sub_ntt_w12_kernel) where sub_block_log says so; 13: never
int g_witness_sub12_cross = 4;  // tuning knob "witness_sub12_cross": most cross stages of a transform that takes 2^12 blocks
int g_witness_cross_pair = 1;  // tuning knob "witness_cross_pair": two groups per thread and 16-byte accesses in the cross passes (0: the round-3 form)
int g_witness_cross_maxr = 6;  // tuning knob "witness_cross_maxr": most stages of one cross pass (FP64 arithmetic; 4 = the round-3 passes)
int g_witness_force_bc = 0;  // tuning knob "witness_force_bc": pretend the ring primes have only this 2-adicity (tests)
int g_witness_bc2 = 1;       // tuning knob "witness_bc2": two-dimensional block convolutions where they apply (0: the pairwise form)

// tuning knob "witness_inc": ring primes without a 2M-th root of unity run the multi-pass path on INCOMPLETE transforms
// (witness_inc.hpp) where the conditions of build_plan hold; 0: the block convolutions (round 3's path for those primes)
int g_witness_inc = 1;
extern int g_witness_lds_logM, g_witness_tree_log, g_witness_tree_ct;
static bool single_tile_ok(int logM);
// largest tile of the product tree in the multi-pass path: full transforms of that length run inside the tile kernels
static int tree_tile_log(bool fp, int logM) {
  const int logT = std::min(g_witness_lds_logM, logM);
  return (fp && logT == 13 && logM >= 15 && g_witness_tree_ct == 2 && g_witness_tree_log >= 14) ? 14 : logT;
}

static void free_plan_tables(WitnessPlan *P);
static WitnessPlan *build_plan(rs_ctx *ctx, size_t m) {
  using namespace hostw;
  RS_REQUIRE(m >= 1, "need at least one constraint");
  WitnessPlan *P = new WitnessPlan();
  P->m = m;
  P->logM = std::max(1, clog2(m));
  P->M = (size_t)1 << P->logM;
  const size_t M = P->M;
  const int logM = P->logM;
  if (logM > 22)
    throw Error(RS_ERR_UNSUPPORTED, "witness map beyond 2^22 constraints is not supported");
  P->limb.resize(ctx->L);
  int vmin = 64;
  for (int li = 0; li < ctx->L; li++) vmin = std::min(vmin, host::two_adicity(ctx->q[li]));
  if (g_witness_force_bc > 0) vmin = std::min(vmin, g_witness_force_bc);  // tests: the block path on well-endowed primes
  const bool blocked = vmin < logM + 1;
  // Incomplete transforms (witness_inc.hpp): the multi-pass path as it is, every transform longer than 2^(a prime's
  // 2-adicity) stopped that many stages early.  Needs: columns that take the multi-pass path; full transforms inside the
  // product tree's tiles; at most RS_INC_MAX stages missing, all of them inside the LAST round of a sub-transform block.
  {
    const int logT = std::min(g_witness_lds_logM, logM);
    // (M = 2^14 on the FP64 arithmetic normally runs in ONE 2^14 tile -- single_tile_ok -- whose Newton conversion needs a
    // complete 2^15-point transform: a prime without it takes the multi-pass path on 2^13 tiles instead, one stage short)
    const bool multi = logM > g_witness_lds_logM;
    P->incomplete = blocked && g_witness_inc && multi && vmin >= tree_tile_log(!ctx->use_int, logM) && logM + 1 - vmin <= RS_INC_MAX &&
                    std::min(logT, 12) > RS_INC_MAX;
  }
  const bool bcpath = blocked && !P->incomplete;
  // full-length transforms serve 2^21 and 2^22 constraints as they serve 2^20 (one more cross pass), complete or not; the
  // block convolutions stop at 2^20 (the two-level transform across blocks is built for Y <= 256 blocks of 2^13)
  if (bcpath && logM > 20)
    throw Error(RS_ERR_UNSUPPORTED, "witness map beyond 2^20 constraints needs ring primes = 1 mod 2^(log2 M - 3) (full-length transforms, "
                                    "at most four stages short); the block convolutions of other primes stop at 2^20");
  P->bc2 = bcpath && g_witness_bc2 && !ctx->use_int && vmin >= 14 && logM >= 15;
  P->bcLog = bcpath ? (P->bc2 ? 14 : std::min(vmin, 13)) : 0;
  // every context prime is 1 mod 2*N_enc with N_enc >= 16, so the 2-adicity is at least 5
  RS_REQUIRE(!bcpath || P->bcLog > SCHOOL_LEVELS, "ring prime with too little 2-adicity for the witness map");
  const size_t Bc = bcpath ? (size_t)1 << (P->bcLog - 1) : 0, nblk = bcpath ? std::max<size_t>(1, M / Bc) : 0;
  // one host thread per ring limb: the tables of different primes are independent (product tree, Newton iteration for
  // rev(Z)^-1 -- 0.6 s per limb at the headline, the bulk of a process's first proof)
  auto build_limb = [&](int li) {
    LimbPlan &lp = P->limb[li];
    const uint64_t p = ctx->q[li];
    RS_REQUIRE(p > 2 * M, "ring prime too small for the evaluation domain");
    lp.p = p;
    // longest transform the device tables serve: the block length (block convolutions), this prime's 2-adicity
    // (incomplete transforms: longer ones stop there), else 2M
    lp.adic = 64;
    if (P->incomplete) {
      int a = host::two_adicity(p);
      if (g_witness_force_bc > 0) a = std::min(a, g_witness_force_bc);
      if (a < logM + 1) lp.adic = a;
    }
    const int tabLog = bcpath ? P->bcLog : std::min(logM + 1, lp.adic);
    auto inc_of = [&](int logn) { return logn > lp.adic ? logn - lp.adic : 0; };
    const CycTab T = make_cyc(p, tabLog);
    auto bal = [&](uint64_t v) { return konst_word(ctx, v, p); };
    {
      const size_t tn = (size_t)1 << tabLog;
      std::vector<uint64_t> tw(tn), itw(tn);
      for (size_t k = 0; k < tn; k++) tw[k] = bal(T.tw[k]), itw[k] = bal(T.itw[k]);
      lp.d_tw = up(tw);
      lp.d_itw = up(itw);
    }
    // spectra (scaled by 1/(2 Bc)) of the Bc-coefficient blocks of a polynomial: [blocks][2 Bc]
    auto block_spectra = [&](const std::vector<uint64_t> &poly, size_t blocks, bool raw = false) {
      std::vector<uint64_t> out(blocks * 2 * Bc, 0);
      const uint64_t sc = invmod((uint64_t)(2 * Bc) % p, p);
      for (size_t b = 0; b < blocks; b++) {
        std::vector<uint64_t> f(2 * Bc, 0);
        for (size_t k = 0; k < Bc && b * Bc + k < poly.size(); k++) f[k] = poly[b * Bc + k];
        ntt_fwd(f, P->bcLog, T);
        for (size_t k = 0; k < 2 * Bc; k++) out[b * 2 * Bc + k] = raw ? mulmod(f[k], sc, p) : bal(mulmod(f[k], sc, p));
      }
      return out;
    };
    // bc2: the Y-point transform across the (zero-padded) blocks of such spectra, point by point, scaled by 1/Y; output
    // [Y][2 Bc] in the order the device's forward transform across blocks leaves its results (host ntt_fwd order)
    auto across_blocks = [&](const std::vector<uint64_t> &poly, size_t blocks, uint64_t *dst) {
      const std::vector<uint64_t> sp = block_spectra(poly, blocks, true);
      const size_t Y = 2 * blocks;
      const int logY = clog2(Y);
      const uint64_t sc = invmod((uint64_t)Y % p, p);
      std::vector<uint64_t> v(Y);
      for (size_t k = 0; k < 2 * Bc; k++) {
        for (size_t y = 0; y < Y; y++) v[y] = y < blocks ? sp[y * 2 * Bc + k] : 0;
        ntt_fwd(v, logY, T);
        for (size_t y = 0; y < Y; y++) dst[y * 2 * Bc + k] = bal(mulmod(v[y], sc, p));
      }
    };
    lp.fwd_mask2 = fwd_reduce_mask(p, logM + 1);
    lp.inv_mask2 = inv_reduce_mask(p, logM + 1);
    // factorials
    std::vector<uint64_t> fact(M), ifact(M);
    fact[0] = 1;
    for (size_t j = 1; j < M; j++) fact[j] = mulmod(fact[j - 1], (uint64_t)j % p, p);
    ifact[M - 1] = invmod(fact[M - 1], p);
    for (size_t j = M - 1; j > 0; j--) ifact[j - 1] = mulmod(ifact[j], (uint64_t)j % p, p);
    {
      std::vector<uint64_t> v(M, 0);
      for (size_t j = 0; j < m; j++) v[j] = bal(ifact[j]);
      lp.d_invfact = up(v);
      std::vector<uint64_t> e(2 * M, 0);
      for (size_t k = 0; k < m; k++) e[k] = (k & 1) ? (p - ifact[k]) % p : ifact[k];
      if (bcpath) {
        e.resize(M);
        lp.d_bc_e = up(block_spectra(e, nblk));
        if (P->bc2) {
          std::vector<uint64_t> t2(2 * nblk * 2 * Bc);
          across_blocks(e, nblk, t2.data());
          lp.d_b2_e = up(t2);
        }
      } else {
        const int nst = logM + 1 - inc_of(logM + 1);  // the inverse undoes nst stages: scale 2^-nst
        ntt_fwd(e, logM + 1, T, nst);
        const uint64_t s2 = invmod(((uint64_t)1 << nst) % p, p);
        std::vector<uint64_t> eh(2 * M);
        for (size_t k = 0; k < 2 * M; k++) eh[k] = bal(mulmod(e[k], s2, p));
        lp.d_ehat = up(eh);
      }
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
      std::vector<uint64_t> dhat((size_t)(logM + 1) * M, 0), dlow((size_t)(SCHOOL_LEVELS + 1) * (M / 2 + 1), 0);
      std::vector<uint64_t> bcd(bcpath && logM > P->bcLog ? (size_t)(logM - P->bcLog) * M : 0, 0);
      std::vector<uint64_t> b2d(P->bc2 && logM > P->bcLog ? (size_t)(logM - P->bcLog) * 2 * M : 0, 0);
      for (int l = 1; l <= logM; l++) {
        const size_t n = (size_t)1 << l, h = n >> 1;
        for (size_t i = 0; i < (M >> l); i++) {
          const auto &dl = prod[l - 1][2 * i];  // h low coefficients, monic of degree h
          if (l <= SCHOOL_LEVELS) {
            for (size_t k = 0; k < h; k++) dlow[(size_t)l * (M / 2 + 1) + i * h + k] = bal(dl[k]);
          } else if (bcpath && l > P->bcLog) {
            // node i of level l: the h / Bc blocks of D_left's low part (the monic x^h term is added by the sink)
            const std::vector<uint64_t> sp = block_spectra(dl, h / Bc);
            std::copy(sp.begin(), sp.end(), bcd.begin() + (size_t)(l - P->bcLog - 1) * M + i * n);
            if (P->bc2) across_blocks(dl, h / Bc, b2d.data() + (size_t)(l - P->bcLog - 1) * 2 * M + i * 2 * n);
          } else {
            std::vector<uint64_t> f(n, 0);
            for (size_t k = 0; k < h; k++) f[k] = dl[k];
            f[h] = 1;
            const int nst = l - inc_of(l);
            ntt_fwd(f, l, T, nst);
            const uint64_t sc = invmod(((uint64_t)1 << nst) % p, p);
            for (size_t k = 0; k < n; k++) dhat[(size_t)l * M + i * n + k] = bal(mulmod(f[k], sc, p));
          }
        }
      }
      // PRECONDITION of the kernels that skip the reduction before the table product (ColPlan::pwmask, fwd_end_needs_reduce in
      // rs_core.hip; tree_wide_kernel, sub_ntt_wide_kernel MODE 2): the spectrum may be as large as 2^50, so mulmod's
      // |a b| <= p 2^49 holds only for BALANCED table entries, |s| <= p/2.  Every entry goes through bal(); checked here so
      // that a future table built any other way fails at plan time, not as a wrong residue.
      if (!ctx->use_int) {
        auto balanced_table = [&](const std::vector<uint64_t> &t) {
          for (uint64_t wd : t) {
            double d;
            memcpy(&d, &wd, 8);
            if (!(d <= 0.5 * (double)p && d >= -0.5 * (double)p)) return false;
          }
          return true;
        };
        RS_REQUIRE(balanced_table(dhat) && balanced_table(bcd) && balanced_table(b2d), "internal: a spectrum table is not balanced (|s| <= p/2)");
      }
      lp.d_dhat = up(dhat);
      lp.d_dlow = up(dlow);
      if (!bcd.empty()) lp.d_bc_d = up(bcd);
      if (!b2d.empty()) lp.d_b2_d = up(b2d);
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
      std::vector<uint64_t> zt(M, 0);
      for (size_t k = 0; k < M && k <= m; k++) zt[k] = bal(Z[k]);
      lp.d_ztab = up(zt);
      if (!blocked && M >= 2) {
        // H on a coset (Rinocchio, where C is interpolated anyway): H(g w^i) = (A B - C)(g w^i) / Z(g w^i) at the M points
        // g w^i, none of which may be a root of Z (an integer 0 .. m-1; the point g w^0 = g itself is one for g < m): try
        // g = m + 1, m + 2, ... until Z has no zero there (a given g fails with probability ~ m M / q)
        const uint64_t mi = invmod((uint64_t)M % p, p);
        for (uint64_t g = (uint64_t)m + 1;; g++) {
          RS_REQUIRE(g < (uint64_t)m + 1000, "internal: no coset for the vanishing polynomial");
          std::vector<uint64_t> gp(M), zc(M, 0);
          gp[0] = 1;
          for (size_t k = 1; k < M; k++) gp[k] = mulmod(gp[k - 1], g % p, p);
          for (size_t k = 0; k < M && k <= m; k++) zc[k] = mulmod(Z[k], gp[k], p);
          if (m == M) zc[0] = addmod(zc[0], mulmod(gp[M - 1], g % p, p), p);  // x^M = g^M on the coset
          ntt_fwd(zc, logM, T);
          bool ok = true;
          for (size_t k = 0; k < M && ok; k++) ok = zc[k] != 0;
          if (!ok) continue;
          // batch inversion of the M values
          std::vector<uint64_t> pre(M);
          uint64_t acc = 1;
          for (size_t k = 0; k < M; k++) {
            pre[k] = acc;
            acc = mulmod(acc, zc[k], p);
          }
          uint64_t inv = invmod(acc, p);
          std::vector<uint64_t> zi(M), gh(M), gg(M);
          for (size_t k = M; k-- > 0;) {
            zi[k] = bal(mulmod(inv, pre[k], p));
            inv = mulmod(inv, zc[k], p);
          }
          const uint64_t ginv = invmod(g % p, p);
          uint64_t gi = mi;  // g^-k / M
          for (size_t k = 0; k < M; k++) {
            gg[k] = bal(gp[k]);
            gh[k] = bal(gi);
            gi = mulmod(gi, ginv, p);
          }
          lp.d_cos_g = up(gg);
          lp.d_cos_h = up(gh);
          lp.d_cos_z = up(zi);
          break;
        }
      }
      // S = rev(Z)^-1 mod x^(m-1) (Newton iteration): quo(P, Z) = rev(rev(P) * S mod x^(m-1)) for
      // deg P = 2m-2.  Spectrum at length 2M, scaled by 1/(2M)^2 (two unscaled inverse transforms).
      std::vector<uint64_t> shat(2 * M, 0);
      if (m >= 2) {
        std::vector<uint64_t> f(m - 1);
        for (size_t i2 = 0; i2 + 1 < m; i2++) f[i2] = Z[m - i2];  // rev(Z), constant term Z[m] = 1
        std::vector<uint64_t> g = {1};
        while (g.size() < m - 1) {
          const size_t k2 = std::min(2 * g.size(), m - 1);
          std::vector<uint64_t> fk(f.begin(), f.begin() + k2);
          std::vector<uint64_t> fg = polymul(fk, g, T);
          fg.resize(k2);
          for (auto &x : fg) x = (p - x) % p;  // -f*g
          fg[0] = addmod(fg[0], 2, p);         // 2 - f*g
          std::vector<uint64_t> ng = polymul(g, fg, T);
          ng.resize(k2);
          g = ng;
        }
        for (size_t i2 = 0; i2 < g.size(); i2++) shat[i2] = g[i2];
        if (!bcpath) {
          const int nst = logM + 1 - inc_of(logM + 1);
          ntt_fwd(shat, logM + 1, T, nst);
          const uint64_t s2 = invmod(((uint64_t)1 << nst) % p, p), s4 = mulmod(s2, s2, p);
          for (auto &x : shat) x = mulmod(x, s4, p);
        }
      }
      if (bcpath) {
        shat.resize(M);  // S itself, m - 1 <= M coefficients
        lp.d_bc_s = up(block_spectra(shat, nblk));
        if (P->bc2) {
          std::vector<uint64_t> t2(2 * nblk * 2 * Bc);
          across_blocks(shat, nblk, t2.data());
          lp.d_b2_s = up(t2);
        }
      } else {
        std::vector<uint64_t> sh(2 * M);
        for (size_t k = 0; k < 2 * M; k++) sh[k] = bal(shat[k]);
        lp.d_shat = up(sh);
      }
    }
  };
  {
    std::vector<std::thread> workers;
    std::vector<std::string> errs(ctx->L);
    std::vector<int> codes(ctx->L, RS_OK);
    for (int li = 0; li < ctx->L; li++)
      workers.emplace_back([&, li] {
        try {
          RS_HIP(hipSetDevice(ctx->device));  // a new thread starts on device 0
          build_limb(li);
        } catch (const Error &e) {
          codes[li] = e.code;
          errs[li] = e.what();
        } catch (const std::exception &e) {
          codes[li] = RS_ERR_INVALID;
          errs[li] = e.what();
        }
      });
    for (auto &w : workers) w.join();
    for (int li = 0; li < ctx->L; li++)
      if (codes[li] != RS_OK) {
        free_plan_tables(P);
        delete P;
        throw Error(codes[li], errs[li]);
      }
  }
  return P;
}

static void free_plan_tables(WitnessPlan *P) {
  for (auto &lp : P->limb) {
    void *ptrs[] = {lp.d_tw, lp.d_itw, lp.d_invfact, lp.d_ehat, lp.d_dhat, lp.d_dlow, lp.d_shat, lp.d_ztab, lp.d_bc_e, lp.d_bc_s, lp.d_bc_d,
                    lp.d_b2_e, lp.d_b2_s, lp.d_b2_d, lp.d_cos_g, lp.d_cos_h, lp.d_cos_z};
    for (void *q : ptrs)
      if (q) (void)hipFree(q);
  }
}
static void free_plan(WitnessPlan *P) {
  free_plan_tables(P);
  if (P->d_Zt) (void)hipFree(P->d_Zt);
  delete P;
}

// the knobs build_plan's choice of path (full length / incomplete / block convolutions) and table forms depend on
static uint64_t plan_knob_sig() {
  uint64_t h = 1469598103934665603ull;
  for (int v : {g_witness_lds_logM, g_witness_tree_log, g_witness_tree_ct, g_witness_inc, g_witness_bc2, g_witness_force_bc})
    h = (h ^ (uint64_t)(uint32_t)v) * 1099511628211ull;
  return h;
}
WitnessPlan *get_plan(rs_ctx *ctx, size_t m) {
  const uint64_t sig = plan_knob_sig();
  auto it = ctx->plans.find(m);
  if (it != ctx->plans.end()) {
    if (it->second->knob_sig == sig) return it->second;
    // a tuning knob changed since the plan was built (tests, A/B tools): its tables may be in another form -- rebuild
    RS_HIP(hipDeviceSynchronize());
    free_plan(it->second);
    ctx->plans.erase(it);
  }
  WitnessPlan *P = build_plan(ctx, m);
  P->knob_sig = sig;
  ctx->plans[m] = P;
  return P;
}



// Z as the provers hand it to the inner products: [m + 1][L] values on the device, built at first use (one blocking
// upload per plan; every later proof reads the cached array -- no host transpose, no synchronisation inside a proof)
const uint64_t *witness_Z_rows(rs_ctx *ctx, size_t m) {
  WitnessPlan *P = get_plan(ctx, m);
  if (!P->d_Zt) {
    const int L = ctx->L;
    std::vector<uint64_t> zt((size_t)L * (m + 1));
    for (int i = 0; i < L; i++)
      for (size_t t = 0; t <= m; t++) zt[t * L + i] = P->limb[i].Z[t];
    RS_HIP(hipMalloc(&P->d_Zt, zt.size() * sizeof(uint64_t)));
    RS_HIP(hipMemcpy(P->d_Zt, zt.data(), zt.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
  }
  return P->d_Zt;
}

// Column plans of limbs limb0, limb0+1, ...: entry k serves the k-th limb of a chunk
template <class M>
static ColPlansT<M> make_colplans(rs_ctx *ctx, const WitnessPlan *P, int limb0 = 0) {
  using T = typename ArithOf<M>::T;
  ColPlansT<M> cp;
  memset(&cp, 0, sizeof(cp));
  for (int i = limb0; i < ctx->L; i++) {
    const LimbPlan &lp = P->limb[i];
    ColPlanT<M> &c = cp.l[i - limb0];
    c.mod = HostArith<M>::make(lp.p);
    c.tw = static_cast<const T *>(lp.d_tw);
    c.itw = static_cast<const T *>(lp.d_itw);
    c.invfact = static_cast<const T *>(lp.d_invfact);
    c.ehat = static_cast<const T *>(lp.d_ehat);
    c.dhat = static_cast<const T *>(lp.d_dhat);
    c.dlow = static_cast<const T *>(lp.d_dlow);
    c.shat = static_cast<const T *>(lp.d_shat);
    c.ztab = static_cast<const T *>(lp.d_ztab);
    c.bc_e = static_cast<const T *>(lp.d_bc_e);
    c.bc_s = static_cast<const T *>(lp.d_bc_s);
    c.bc_d = static_cast<const T *>(lp.d_bc_d);
    c.b2_e = static_cast<const T *>(lp.d_b2_e);
    c.b2_s = static_cast<const T *>(lp.d_b2_s);
    c.b2_d = static_cast<const T *>(lp.d_b2_d);
    c.cos_g = static_cast<const T *>(lp.d_cos_g);
    c.cos_h = static_cast<const T *>(lp.d_cos_h);
    c.cos_z = static_cast<const T *>(lp.d_cos_z);
    c.bc_inv2b = P->bcLog ? HostArith<M>::konst(host::invmod(((uint64_t)1 << P->bcLog) % lp.p, lp.p), lp.p) : T(0);
    c.b2_inv = P->bc2 ? HostArith<M>::konst(host::invmod((uint64_t)(4 * P->M) % lp.p, lp.p), lp.p) : T(0);
    c.fwd_mask2 = lp.fwd_mask2;
    c.inv_mask2 = lp.inv_mask2;
    c.pwmask = 0;
    c.adic = lp.adic;
    for (int l = 0; l < 24; l++) {
      c.fmask[l] = fwd_reduce_mask(lp.p, l);
      // an incomplete transform's inverse starts at stage inc(l) on reduced values (inc_polymul)
      c.imask[l] = inv_reduce_mask(lp.p, l, c.inc(l));
      if (fwd_end_needs_reduce(lp.p, l)) c.pwmask |= 1u << l;
    }
  }
  return cp;
}

static int col_threads(size_t M) { return (int)std::max<size_t>(64, std::min<size_t>(1024, M / 8)); }

int g_witness_lds_logM = 13;  // columns up to 2^13 run entirely inside one LDS tile
int g_witness_tree_log = 14;  // largest tile of the wide product-tree kernel: 13 or 14 (tuning knob "witness_tree_log")
int g_witness_tree_ct = 2;    // 2: wide product-tree kernel (tree_wide_kernel) for 2^13 tiles; 1: level-unrolled tree_columns_kernel; 0: level loop
int g_witness_sub_ct = 2;     // 1: compile-time-length sub-transform kernel for 2^13 blocks of the multi-pass path

// Newton -> monomial levels 1..logT on tiles of 2^logT coefficients of [ncols][M] columns; with
// `newton` (logT == logM) the tiles hold values and the Newton conversion runs first, in the same launch
// FP64 instructions (per lane) of the product tree on one tile of T = 2^logT coefficients: levels
// 1..4 by schoolbook (120 modular multiplies + as many additions per 16 coefficients), every level
// above by a forward and an inverse batched transform plus the spectrum product and the recombination
// pw_reduce: the spectrum is reduced before it meets the table entry (3 more instructions per coefficient and level;
// the wide kernel does it only where ColPlan::pwmask asks for it)
static double tree_fp64(double T, int logT, bool pw_reduce = true) {
  double f = T / 16.0 * (120.0 * 7.0 + 4.0 * 16.0 * 3.0);
  for (int l = SCHOOL_LEVELS + 1; l <= logT; l++) f += 2.0 * ntt_fp64(T, l) + (pw_reduce ? 10.0 : 7.0) * T;
  return f;
}
// Wout / rf (2^14 tiles of the wide kernel only): the right tiles also run the rf forward cross stages of level 15 and write
// that level's workspace [ncols][2^logM] (tree_wide_kernel<14, RF>); the caller then skips the level's source pass.
static bool wide_ok_for_fwd(int logT, int rf) { return g_witness_tree_ct == 2 && logT == 14 && (rf == 2 || rf == 3); }
static void launch_tree_tiles(rs_ctx *ctx, double *cols, size_t ncols, size_t col0, int logM, int logT, size_t S,
                              size_t slots_per_limb, const ColPlans &cp, hipStream_t st, bool newton = false, double *Wout = nullptr, int rf = 0) {
  const size_t T = (size_t)1 << logT;
  const double tiles = (double)(ncols << (logM - logT));
  // Newton conversion (single-tile columns): two passes of forward + inverse M-point transforms and two pointwise products
  const double newton_fp64 = newton ? 4.0 * ntt_fp64((double)T, logT) + 31.0 * (double)T : 0.0;
  // names as rocprofv3 prints them (a prefix of "rs::<name>") so that profiles/ and the live record can be joined
  const bool ct13 = logT == 13 && g_witness_tree_ct;
  const bool wide = !newton && g_witness_tree_ct == 2 && (logT == 13 || logT == 14);
  RS_REQUIRE(!Wout || (wide_ok_for_fwd(logT, rf) && !newton), "tree tiles: forward stages of the next level need the wide 2^14 tile");
  const char *pname = wide ? (logT == 14 ? (Wout ? (rf == 3 ? "tree_wide_kernel<14, 3>" : "tree_wide_kernel<14, 2>") : "tree_wide_kernel<14, 0>") : "tree_wide_kernel<13, 0>")
                      : ct13 ? (newton ? "tree_columns_kernel<512, 13, true>" : "tree_columns_kernel<512, 13, false>")
                           : (newton ? "tree_columns_kernel<NEWTON>" : "tree_columns_kernel");
  bool pw = !wide;  // the model count follows what the wide kernel executes: the pre-product reduction per level only where asked for
  if (wide)
    for (int i = 0; i < RS_MAX_L; i++) pw = pw || ((cp.l[i].pwmask >> logT) & 1u);
  // with Wout: half of the tiles also write a 2^(logT+1)-word node of the next level's workspace, rf - 1 stages each
  ProfScope prof(ctx, st, pname, tiles * (double)T * (Wout ? 24.0 : 16.0),
                 tiles * (tree_fp64((double)T, logT, pw) + newton_fp64) + (Wout ? tiles / 2.0 * ntt_fp64(2.0 * (double)T, rf - 1) : 0.0));
  const size_t lds1 = padded_len(T) * sizeof(double);
  const unsigned grid = (unsigned)(ncols << (logM - logT));
  const int thr = (int)std::max<size_t>(64, std::min<size_t>(1024, T / 16));  // 1024 only for a 2^14 tile (one workgroup per CU)
  RS_REQUIRE(wide || (T / thr <= 16 && logT >= 6), "tree tile out of range");
  RS_REQUIRE(!newton || logT == logM, "fused Newton conversion needs single-tile columns");
#define RS_TREE_LAUNCH_K(KERN)                                                                                   \
  do {                                                                                                           \
    set_max_dyn_lds((const void *)KERN, (int)lds1);     \
    hipLaunchKernelGGL(KERN, dim3(grid), dim3(thr), lds1, st, cols, logM, logT, col0, (unsigned)S,               \
                       (unsigned)slots_per_limb, cp);                                                            \
  } while (0)
#define RS_TREE_LAUNCH(THR)                                          \
  do {                                                               \
    if (newton)                                                      \
      RS_TREE_LAUNCH_K((tree_columns_kernel<THR, 0, true>));         \
    else                                                             \
      RS_TREE_LAUNCH_K((tree_columns_kernel<THR, 0, false>));        \
  } while (0)
  if (wide) {
    const int wl = (int)((T + T / 32) * sizeof(double));
    if (logT == 14 && Wout && rf == 3) {
      set_max_dyn_lds((const void *)tree_wide_kernel<14, 3>, wl);
      hipLaunchKernelGGL((tree_wide_kernel<14, 3>), dim3(grid), dim3(512), wl, st, cols, logM, col0, (unsigned)S, (unsigned)slots_per_limb, cp, Wout);
    } else if (logT == 14 && Wout) {
      set_max_dyn_lds((const void *)tree_wide_kernel<14, 2>, wl);
      hipLaunchKernelGGL((tree_wide_kernel<14, 2>), dim3(grid), dim3(512), wl, st, cols, logM, col0, (unsigned)S, (unsigned)slots_per_limb, cp, Wout);
    } else if (logT == 14) {
      set_max_dyn_lds((const void *)tree_wide_kernel<14, 0>, wl);
      hipLaunchKernelGGL((tree_wide_kernel<14, 0>), dim3(grid), dim3(512), wl, st, cols, logM, col0, (unsigned)S, (unsigned)slots_per_limb, cp,
                         (double *)nullptr);
    } else {
      set_max_dyn_lds((const void *)tree_wide_kernel<13, 0>, wl);
      hipLaunchKernelGGL((tree_wide_kernel<13, 0>), dim3(grid), dim3(256), wl, st, cols, logM, col0, (unsigned)S, (unsigned)slots_per_limb, cp,
                         (double *)nullptr);
    }
  } else if (thr == 512 && logT == 13 && g_witness_tree_ct) {
    if (newton)
      RS_TREE_LAUNCH_K((tree_columns_kernel<512, 13, true>));
    else
      RS_TREE_LAUNCH_K((tree_columns_kernel<512, 13, false>));
  } else if (thr == 1024) RS_TREE_LAUNCH(1024);
  else if (thr == 512) RS_TREE_LAUNCH(512);
  else if (thr == 256) RS_TREE_LAUNCH(256);
  else if (thr == 128) RS_TREE_LAUNCH(128);
  else RS_TREE_LAUNCH(64);
#undef RS_TREE_LAUNCH
#undef RS_TREE_LAUNCH_K
  RS_HIP(hipGetLastError());
}

// The same tile work for any arithmetic (the integer contexts): levels 1..logT of the product tree on tiles of
// 2^logT Newton coefficients, tile + scratch in LDS, all-barrier rounds.
template <class CPS>
__global__ void __launch_bounds__(1024)
tree_tiles_generic_kernel(typename CPS::T *__restrict__ cols, int logM, int logT, size_t col0, unsigned S, unsigned slots_per_limb,
                          CPS plans) {
  using T = typename CPS::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *s = reinterpret_cast<T *>(smem);
  const unsigned nb = 1u << (logM - logT);
  const size_t col = blockIdx.x / nb;
  const int pos0 = (int)(blockIdx.x % nb) << logT, Tn = 1 << logT;
  const ColPlanT<typename CPS::M> &P = plans.l[((col0 + col) % S) / slots_per_limb];
  T *c = cols + col * ((size_t)1 << logM) + pos0;
  for (int i = threadIdx.x; i < Tn; i += blockDim.x) s[pidx(i)] = c[i];
  __syncthreads();
  tree_levels_lds(s, logT, logM, pos0, P);
  for (int i = threadIdx.x; i < Tn; i += blockDim.x) c[i] = reduce(s[pidx(i)], P.mod);
}
template <class M>
static void launch_tree_tiles_generic(rs_ctx *ctx, typename ArithOf<M>::T *cols, size_t ncols, size_t col0, int logM, int logT, size_t S,
                                      size_t slots_per_limb, const ColPlansT<M> &cp, hipStream_t st) {
  const size_t T = (size_t)1 << logT;
  const size_t lds = padded_len(tree_scratch_offset((int)T) + T) * sizeof(uint64_t);
  ProfScope prof(ctx, st, "tree_tiles_generic_kernel", (double)(ncols << (logM - logT)) * (double)T * 16.0,
                 (double)(ncols << (logM - logT)) * tree_fp64((double)T, logT));
  set_max_dyn_lds((const void *)tree_tiles_generic_kernel<ColPlansT<M>>, (int)lds);
  hipLaunchKernelGGL(tree_tiles_generic_kernel<ColPlansT<M>>, dim3((unsigned)(ncols << (logM - logT))), dim3(col_threads(2 * T)), lds, st,
                     cols, logM, logT, col0, (unsigned)S, (unsigned)slots_per_limb, cp);
  RS_HIP(hipGetLastError());
}

template <bool INV, int MODE, class M, int V>
static void launch_cross_pass(int R, const dim3 &grid, const CrossArgs &a, const ColPlansT<M> &cp, hipStream_t st) {
  using CPS = ColPlansT<M>;
  if constexpr (std::is_same<M, Mod>::value) {  // radix 32 / 64 passes: six cross stages (M = 2^18) in ONE pass over the workspace instead of two
    if (R == 6) {
      hipLaunchKernelGGL((cross_kernel<INV, 6, MODE, CPS, 1>), grid, dim3(256), 0, st, a, cp);  // 64 elements per thread already
      return;
    }
    if (R == 5) {
      hipLaunchKernelGGL((cross_kernel<INV, 5, MODE, CPS, V>), grid, dim3(256), 0, st, a, cp);
      return;
    }
  }
  switch (R) {
    case 4: hipLaunchKernelGGL((cross_kernel<INV, 4, MODE, CPS, V>), grid, dim3(256), 0, st, a, cp); break;
    case 3: hipLaunchKernelGGL((cross_kernel<INV, 3, MODE, CPS, V>), grid, dim3(256), 0, st, a, cp); break;
    case 2: hipLaunchKernelGGL((cross_kernel<INV, 2, MODE, CPS, V>), grid, dim3(256), 0, st, a, cp); break;
    default: hipLaunchKernelGGL((cross_kernel<INV, 1, MODE, CPS, V>), grid, dim3(256), 0, st, a, cp); break;
  }
}

// Cross stages of the length-2^logsub transforms in W[ncols][2^logtot]: forward stages
// [0, logsub-logB) (the first pass reads through source MODE from a.src), or inverse stages
// [logB, logsub) (the last pass writes through sink MODE to a.dst).
// algorithmic 8-byte words per column of a cross pass that reads through source / writes through sink MODE
template <bool INV, int MODE>
static double cross_words(const CrossArgs &a, bool special) {
  const double n = (double)((size_t)1 << a.logtot), M = (double)((size_t)1 << a.logM);
  if (!special || MODE == 0) return 2.0 * n;
  if (!INV) return n + (MODE == CS_FILL_RIGHT ? M / 2.0 : M);                  // source words + workspace written
  return n + (MODE == CD_COMBINE || MODE == CD_COMBINE_CANON ? 1.5 * M : M);  // workspace read + sink traffic
}
// profile name of one instantiation, as rocprofv3 prints it ("rs::cross_kernel<false, 4, 1, ..."): static storage
static const char *cross_name(bool inv, int R, int mode) {
  static std::mutex mu;
  static std::map<int, std::string> names;
  std::lock_guard<std::mutex> lk(mu);
  const int key = (inv ? 1 : 0) | (R << 1) | (mode << 8);
  auto it = names.find(key);
  if (it == names.end())
    it = names.emplace(key, std::string("cross_kernel<") + (inv ? "true" : "false") + ", " + std::to_string(R) + ", " + std::to_string(mode) + ",").first;
  return it->second.c_str();
}
template <bool INV, int MODE, class M>
static void launch_cross(rs_ctx *ctx, CrossArgs a, size_t ncols, int logB, const ColPlansT<M> &cp, hipStream_t st) {
  const int ncross = a.logsub - logB;
  const size_t groups = ((size_t)1 << a.logtot);
  int done = 0;
  while (done < ncross) {
    // FP64: up to six stages per pass (64 strided elements per thread: the pass is HBM bound, the registers are idle) --
    // a transform with five or six cross stages (M = 2^17, 2^18) crosses the workspace once instead of twice
    const int R = pick_radix(ncross - done, std::is_same<M, Mod>::value ? std::max(1, std::min(6, g_witness_cross_maxr)) : 4);
    a.s0 = INV ? logB + done : done;
    const bool special = INV ? (done + R >= ncross) : (done == 0);
    // two adjacent groups per thread, 16-byte accesses (cross_kernel<..., 2>): needs wave-uniform twiddles for 128
    // consecutive groups (smallest gap >= 2^7) and 16-byte aligned columns
    const bool paired = g_witness_cross_pair && R <= 5 && logB >= 8 && a.logM >= 2 &&
                        (((uintptr_t)a.W | (uintptr_t)a.src | (uintptr_t)a.dst) & 15) == 0;
    const unsigned gx = (unsigned)std::max<size_t>(1, std::min<size_t>((groups >> R) / (paired ? 512 : 256), 1024));
    const dim3 grid(gx, (unsigned)ncols);
    ProfScope prof(ctx, st, cross_name(INV, R, special ? MODE : 0), (double)ncols * 8.0 * cross_words<INV, MODE>(a, special),
                   (double)ncols * ntt_fp64((double)groups, (special && !INV) ? R - 1 : R));  // a source pass: stage 0 meets zero padding, a copy
    if (special && paired)
      launch_cross_pass<INV, MODE, M, 2>(R, grid, a, cp, st);
    else if (special)
      launch_cross_pass<INV, MODE, M, 1>(R, grid, a, cp, st);
    else if (paired)
      launch_cross_pass<INV, 0, M, 2>(R, grid, a, cp, st);
    else
      launch_cross_pass<INV, 0, M, 1>(R, grid, a, cp, st);
    done += R;
  }
  RS_HIP(hipGetLastError());
}

// Block of the rooted sub-transforms of a multi-pass transform of length 2^logsub: the LDS tile (2^13), or 2^12 for the
// FP64 contexts (knob witness_sub_log = 12; sub_ntt_w12_kernel, four workgroups per CU: 14 % faster per coefficient)
// where the extra cross stage keeps the cross pass at four stages or fewer (knob witness_sub12_cross) -- a five-stage
// FORWARD pass from a source costs more than the smaller block saves (measured on the headline).
template <class M>
static int sub_block_log(int logT, int logsub) {
  return (std::is_same<M, Mod>::value && g_witness_sub_log == 12 && logT == 13 && logsub - 12 <= g_witness_sub12_cross) ? 12 : logT;
}

// FP64 instructions per coefficient of the pointwise step of a fused sub-transform: one modular product, or (incomplete
// transforms, witness_inc.hpp) 2^inc of them, their sums and reductions, and the product with eta
static double pointwise_fp64(int inc) { return inc ? 7.0 * (double)(1 << inc) + 13.0 : 7.0; }

// `inc`: stages every transform of this launch stops short (the same for every column: launch_sub splits otherwise)
template <int MODE, class M>
static void launch_sub_inc(rs_ctx *ctx, typename ArithOf<M>::T *X, size_t ncols, size_t col0, int logtot, int logsub, int logB,
                           const TabPtrs *tabs, size_t tab_period, size_t S, size_t spl, const ColPlansT<M> &cp, hipStream_t st, int inc) {
  constexpr bool FP = std::is_same<M, Mod>::value;
  const size_t lds = padded_len((size_t)1 << logB) * sizeof(double);
  const size_t bpc = (size_t)1 << (logtot - logB);
  static const char *const names[5] = {"sub_ntt_kernel<0", "sub_ntt_kernel<1", "sub_ntt_kernel<2", "sub_ntt_kernel<3", "sub_ntt_kernel<4"};
  static const char *const names_ct[5] = {"sub_ntt_ct_kernel<0, 13>", "sub_ntt_ct_kernel<1, 13>", "sub_ntt_ct_kernel<2, 13>", "sub_ntt_ct_kernel<3, 13>", "sub_ntt_ct_kernel<4, 13>"};
  // names as rocprofv3 prints them: "sub_ntt_wide_kernel<MODE, INC>"
  static const char *const names_wide[5][5] = {
      {"sub_ntt_wide_kernel<0, 0>", "sub_ntt_wide_kernel<0, 1>", "sub_ntt_wide_kernel<0, 2>", "sub_ntt_wide_kernel<0, 3>", "sub_ntt_wide_kernel<0, 4>"},
      {"sub_ntt_wide_kernel<1, 0>", "", "", "", ""},
      {"sub_ntt_wide_kernel<2, 0>", "sub_ntt_wide_kernel<2, 1>", "sub_ntt_wide_kernel<2, 2>", "sub_ntt_wide_kernel<2, 3>", "sub_ntt_wide_kernel<2, 4>"},
      {"sub_ntt_wide_kernel<3, 0>", "sub_ntt_wide_kernel<3, 1>", "sub_ntt_wide_kernel<3, 2>", "sub_ntt_wide_kernel<3, 3>", "sub_ntt_wide_kernel<3, 4>"},
      {"sub_ntt_wide_kernel<4, 0>", "", "", "", ""}};
  static const char *const names_w12[5][5] = {
      {"sub_ntt_w12_kernel<0, 0>", "sub_ntt_w12_kernel<0, 1>", "sub_ntt_w12_kernel<0, 2>", "sub_ntt_w12_kernel<0, 3>", "sub_ntt_w12_kernel<0, 4>"},
      {"sub_ntt_w12_kernel<1, 0>", "", "", "", ""},
      {"sub_ntt_w12_kernel<2, 0>", "sub_ntt_w12_kernel<2, 1>", "sub_ntt_w12_kernel<2, 2>", "sub_ntt_w12_kernel<2, 3>", "sub_ntt_w12_kernel<2, 4>"},
      {"sub_ntt_w12_kernel<3, 0>", "sub_ntt_w12_kernel<3, 1>", "sub_ntt_w12_kernel<3, 2>", "sub_ntt_w12_kernel<3, 3>", "sub_ntt_w12_kernel<3, 4>"},
      {"sub_ntt_w12_kernel<4, 0>", "", "", "", ""}};
  RS_REQUIRE(inc >= 0 && inc <= RS_INC_MAX && (inc == 0 || (MODE != 1 && MODE != 4)) && logB > inc, "internal: sub-transform launch out of range");
#ifdef RS_EXPERIMENTS
  const bool ct = FP && logB == 13 && MODE != 1 && g_witness_sub_ct && (MODE != 4 || g_witness_sub_ct == 2) && (inc == 0 || g_witness_sub_ct == 2);  // MODE 4: generic, wide and 2^12 kernels only
#else
  const bool ct = FP && logB == 13 && MODE != 1 && g_witness_sub_ct == 2;  // 0: the generic kernel; 1 and 3 exist in the experiments build only
#endif
  const double Bn = (double)((size_t)1 << logB), blocks = (double)(ncols * bpc);
  static const char *const names_w16[5] = {"sub_ntt_wide16_kernel<0>", "sub_ntt_wide16_kernel<1>", "sub_ntt_wide16_kernel<2>", "sub_ntt_wide16_kernel<3>", "sub_ntt_wide16_kernel<4>"};
  const bool w12 = FP && logB == 12 && MODE != 1 && g_witness_sub_log == 12;
  ProfScope prof(ctx, st, w12 ? names_w12[MODE][inc] : ct ? (g_witness_sub_ct == 3 ? names_w16[MODE] : g_witness_sub_ct == 2 ? names_wide[MODE][inc] : names_ct[MODE]) : names[MODE], blocks * Bn * (MODE == 4 ? 32.0 : MODE == 3 ? 24.0 : 16.0),
                 blocks * ((MODE >= 2 ? 2.0 : 1.0) * ntt_fp64(Bn, logB - inc) + (MODE == 4 ? 24.0 * Bn : MODE >= 2 ? pointwise_fp64(inc) * Bn : 0.0)));
  static TabPtrs none{};
  const TabPtrs &tp = tabs ? *tabs : none;
  if constexpr (FP) {
    const unsigned long long nb = (unsigned long long)(ncols * bpc);
    if (logB == 12 && MODE != 1 && g_witness_sub_log == 12) {
      const int wl = 4352 * (int)sizeof(double);
#define RS_W12_LAUNCH(INC)                                                                                                              \
  hipLaunchKernelGGL((sub_ntt_w12_kernel<MODE, INC>), dim3((unsigned)std::min<unsigned long long>(nb, 1024)), dim3(256), wl, st, X,     \
                     logsub - logB, tp, (unsigned)std::max<size_t>(1, tab_period), (unsigned)bpc, col0, (unsigned)S, (unsigned)spl, cp, nb)
      if constexpr (MODE == 4) {
        RS_W12_LAUNCH(0);
      } else {
        switch (inc) {
          case 0: RS_W12_LAUNCH(0); break;
          case 1: RS_W12_LAUNCH(1); break;
          case 2: RS_W12_LAUNCH(2); break;
          case 3: RS_W12_LAUNCH(3); break;
          default: RS_W12_LAUNCH(4); break;
        }
      }
#undef RS_W12_LAUNCH
      RS_HIP(hipGetLastError());
      return;
    }
#ifdef RS_EXPERIMENTS
    if (logB == 13 && MODE != 1 && MODE != 4 && g_witness_sub_ct == 3 && inc == 0) {
      const int wl = (int)(WideShape<13>::TILE * sizeof(double));
      set_max_dyn_lds((const void *)sub_ntt_wide16_kernel<MODE>, wl);
      hipLaunchKernelGGL((sub_ntt_wide16_kernel<MODE>), dim3((unsigned)std::min<unsigned long long>(nb, 512)), dim3(512), wl, st, X,
                         logsub - logB, tp, (unsigned)std::max<size_t>(1, tab_period), (unsigned)bpc, col0, (unsigned)S, (unsigned)spl, cp, nb);
      RS_HIP(hipGetLastError());
      return;
    }
#endif
    if (logB == 13 && MODE != 1 && g_witness_sub_ct == 2) {
      const int wl = (int)(WideShape<13>::TILE * sizeof(double));
#define RS_WIDE_LAUNCH(INC)                                                                                                             \
  do {                                                                                                                                  \
    set_max_dyn_lds((const void *)sub_ntt_wide_kernel<MODE, INC>, wl);                                                                  \
    hipLaunchKernelGGL((sub_ntt_wide_kernel<MODE, INC>), dim3((unsigned)std::min<unsigned long long>(nb, 512)), dim3(256), wl, st, X,   \
                       logsub - logB, tp, (unsigned)std::max<size_t>(1, tab_period), (unsigned)bpc, col0, (unsigned)S, (unsigned)spl, cp, \
                       nb, (const double *)nullptr);                                                                                    \
  } while (0)
      if constexpr (MODE == 4) {
        RS_WIDE_LAUNCH(0);
      } else {
        switch (inc) {
          case 0: RS_WIDE_LAUNCH(0); break;
          case 1: RS_WIDE_LAUNCH(1); break;
          case 2: RS_WIDE_LAUNCH(2); break;
          case 3: RS_WIDE_LAUNCH(3); break;
          default: RS_WIDE_LAUNCH(4); break;
        }
      }
#undef RS_WIDE_LAUNCH
      RS_HIP(hipGetLastError());
      return;
    }
#ifdef RS_EXPERIMENTS
    if (logB == 13 && MODE != 1 && MODE != 4 && g_witness_sub_ct && inc == 0) {
      set_max_dyn_lds((const void *)sub_ntt_ct_kernel<MODE, 13>, (int)lds);
      hipLaunchKernelGGL((sub_ntt_ct_kernel<MODE, 13>), dim3((unsigned)(ncols * bpc)), dim3(512), lds, st, X, logsub - logB, tp,
                         (unsigned)std::max<size_t>(1, tab_period), (unsigned)bpc, col0, (unsigned)S, (unsigned)spl, cp);
      RS_HIP(hipGetLastError());
      return;
    }
#endif
  }
  // the generic kernel reads every column's inc from its plan
  set_max_dyn_lds((const void *)sub_ntt_kernel<MODE, ColPlansT<M>>, (int)lds);
  const int thr = (int)std::max<size_t>(64, std::min<size_t>(1024, ((size_t)1 << logB) / 16));
  hipLaunchKernelGGL((sub_ntt_kernel<MODE, ColPlansT<M>>), dim3((unsigned)(ncols * bpc)), dim3(thr), lds, st, X, logB, logsub - logB, tp,
                     (unsigned)std::max<size_t>(1, tab_period), (unsigned)bpc, col0, (unsigned)S, (unsigned)spl, cp);
  RS_HIP(hipGetLastError());
}

// Sub-transforms of the length-2^logsub transforms in X[ncols][2^logtot].  The tuned kernels take the number of stages an
// incomplete transform stops short as a template parameter, so columns of primes with different 2-adicity (column c belongs
// to limb ((col0 + c) % S) / spl) go in separate launches -- one launch whenever they agree (always at the headline, where a
// chunk of columns is one limb).
template <int MODE, class M>
static void launch_sub(rs_ctx *ctx, typename ArithOf<M>::T *X, size_t ncols, size_t col0, int logtot, int logsub, int logB,
                       const TabPtrs *tabs, size_t tab_period, size_t S, size_t spl, const ColPlansT<M> &cp, hipStream_t st) {
  auto inc_at = [&](size_t c) { return cp.l[((col0 + c) % S) / spl].inc(logsub); };
  bool same = true;
  for (size_t c = 0; c < ncols && same; c += spl - (col0 + c) % spl) same = inc_at(c) == inc_at(0);
  if (same) {
    launch_sub_inc<MODE, M>(ctx, X, ncols, col0, logtot, logsub, logB, tabs, tab_period, S, spl, cp, st, ncols ? inc_at(0) : 0);
    return;
  }
  for (size_t c = 0; c < ncols;) {
    size_t e = std::min(ncols, c + spl - (col0 + c) % spl);
    while (e < ncols && inc_at(e) == inc_at(c)) e = std::min(ncols, e + spl);  // neighbouring limbs that agree: one launch
    TabPtrs tp = tabs ? *tabs : TabPtrs{};
    if (MODE == 3 && tabs) tp.t[0] = static_cast<const typename ArithOf<M>::T *>(tabs->t[0]) + (c << logtot);  // the other workspace: same shape as X
    launch_sub_inc<MODE, M>(ctx, X + (c << logtot), e - c, col0 + c, logtot, logsub, logB, tabs ? &tp : nullptr, tab_period, S, spl, cp, st, inc_at(c));
    c = e;
  }
}


// one two-dimensional block convolution of `ncols * units` operands of Y/2 blocks each.  MODE 2: against the table
// `tab` ([units][Y][2B] per limb, limbs from limb0 on); MODE 3: against the data spectra `other` ([ncols*units][Y][2][B], same
// layout as Ws); MODE 0: forward only (Ws receives the spectra; no sink).
// skip_fwd / skip_inv: the transform across blocks at that end is run by a turn kernel (bc2_level_turn_kernel, bc2_h_turn_kernel)
template <int SRC, int DST, int MODE>
static void bc2_conv(rs_ctx *ctx, Bc2Args a, int logY, size_t ncols, const TabPtrs *tab, const double *other, const ColPlans &cp,
                     hipStream_t st, bool skip_fwd = false, bool skip_inv = false) {
  const size_t Y = (size_t)1 << logY, cu = ncols * (size_t)a.units;
  const dim3 grid((unsigned)(BC2_B / 2 / 256), (unsigned)cu);
  // Y <= 32: the transform across blocks in one thread's registers; Y = 64 .. 256 (M >= 2^18): in two levels
  RS_REQUIRE(cu <= 65535 && logY >= 2 && logY <= 8, "two-dimensional block convolution out of range");
  const dim3 grid_parts(grid.x << std::max(0, logY - 5), grid.y);  // one workgroup per (position range, part of 32 blocks)
  if (!skip_fwd) {
    // words: the Y/2 source blocks (read once per part in the two-level form) + Y blocks written
    const double reads = logY > 5 ? (double)(Y / 2) * (double)(1 << (logY - 5)) : (double)(Y / 2);
    ProfScope prof(ctx, st, logY > 5 ? "bc2_yfwd_big_kernel" : "bc2_yfwd_kernel", (double)cu * ((double)Y + reads) * BC2_B * 8.0,
                   (double)cu * BC2_B * ntt_fp64((double)Y, logY));
    switch (logY) {
      case 2: hipLaunchKernelGGL((bc2_yfwd_kernel<SRC, 2>), grid, dim3(256), 0, st, a, cp); break;
      case 3: hipLaunchKernelGGL((bc2_yfwd_kernel<SRC, 3>), grid, dim3(256), 0, st, a, cp); break;
      case 4: hipLaunchKernelGGL((bc2_yfwd_kernel<SRC, 4>), grid, dim3(256), 0, st, a, cp); break;
      case 5: hipLaunchKernelGGL((bc2_yfwd_kernel<SRC, 5>), grid, dim3(256), 0, st, a, cp); break;
      case 6: hipLaunchKernelGGL((bc2_yfwd_big_kernel<SRC, 1>), grid_parts, dim3(256), 0, st, a, cp); break;
      case 7: hipLaunchKernelGGL((bc2_yfwd_big_kernel<SRC, 2>), grid_parts, dim3(256), 0, st, a, cp); break;
      default: hipLaunchKernelGGL((bc2_yfwd_big_kernel<SRC, 3>), grid_parts, dim3(256), 0, st, a, cp); break;
    }
  }
  {
    const unsigned long long nb = (unsigned long long)(cu * Y * 2);
    const double Bn = (double)BC2_B;
    static const char *const names[4] = {"sub_ntt_wide_kernel<0, 0>", "sub_ntt_wide_kernel<1, 0>", "sub_ntt_wide_kernel<2, 0>", "sub_ntt_wide_kernel<3, 0>"};
    ProfScope prof(ctx, st, names[MODE], (double)nb * Bn * (MODE == 3 ? 24.0 : 16.0),
                   (double)nb * ((MODE >= 2 ? 2.0 : 1.0) * ntt_fp64(Bn, BC2_LOGB) + (MODE >= 2 ? 7.0 * Bn : 0.0)));
    TabPtrs tp{};
    if (MODE == 2) tp = *tab;
    if (MODE == 3) tp.t[0] = other;
    {
      const int wl = (int)(WideShape<13>::TILE * sizeof(double));
      set_max_dyn_lds((const void *)sub_ntt_wide_kernel<MODE>, wl);
      hipLaunchKernelGGL((sub_ntt_wide_kernel<MODE>), dim3((unsigned)std::min<unsigned long long>(nb, 512)), dim3(256), wl, st, a.Ws, 1, tp,
                         (unsigned)((size_t)a.units * Y * 2), (unsigned)((size_t)a.units * Y * 2), a.col0, a.S, a.slots_per_limb, cp, nb,
                         (const double *)a.Wy);
    }
  }
  if (skip_inv) {
    RS_HIP(hipGetLastError());
    return;
  }
  if (MODE != 0 && logY <= 5) {
    ProfScope prof(ctx, st, "bc2_yinv_kernel", (double)cu * (double)Y * BC2_B * 24.0, (double)cu * 2.0 * BC2_B * ntt_fp64((double)Y, logY));
    switch (logY) {
      case 2: hipLaunchKernelGGL((bc2_yinv_kernel<DST, 2>), grid, dim3(256), 0, st, a, cp); break;
      case 3: hipLaunchKernelGGL((bc2_yinv_kernel<DST, 3>), grid, dim3(256), 0, st, a, cp); break;
      case 4: hipLaunchKernelGGL((bc2_yinv_kernel<DST, 4>), grid, dim3(256), 0, st, a, cp); break;
      default: hipLaunchKernelGGL((bc2_yinv_kernel<DST, 5>), grid, dim3(256), 0, st, a, cp); break;
    }
  } else if (MODE != 0) {
    {  // in place on Ws: 2 x Y x 2B words
      ProfScope prof(ctx, st, "bc2_yinv_a_kernel", (double)cu * (double)Y * BC2_B * 32.0, (double)cu * 2.0 * BC2_B * ntt_fp64((double)Y, 5));
      switch (logY) {
        case 6: hipLaunchKernelGGL((bc2_yinv_a_kernel<1>), grid_parts, dim3(256), 0, st, a, cp); break;
        case 7: hipLaunchKernelGGL((bc2_yinv_a_kernel<2>), grid_parts, dim3(256), 0, st, a, cp); break;
        default: hipLaunchKernelGGL((bc2_yinv_a_kernel<3>), grid_parts, dim3(256), 0, st, a, cp); break;
      }
    }
    const dim3 grid32(grid.x * 32, grid.y);
    ProfScope prof(ctx, st, "bc2_yinv_b_kernel", (double)cu * (double)Y * BC2_B * 24.0, (double)cu * 2.0 * BC2_B * ntt_fp64((double)Y, logY - 5));
    switch (logY) {
      case 6: hipLaunchKernelGGL((bc2_yinv_b_kernel<DST, 1>), grid32, dim3(256), 0, st, a, cp); break;
      case 7: hipLaunchKernelGGL((bc2_yinv_b_kernel<DST, 2>), grid32, dim3(256), 0, st, a, cp); break;
      default: hipLaunchKernelGGL((bc2_yinv_b_kernel<DST, 3>), grid32, dim3(256), 0, st, a, cp); break;
    }
  }
  RS_HIP(hipGetLastError());
}

int g_witness_h_turn = 1;  // tuning knob "witness_h_turn": fuse the last inverse cross pass of A B with the first forward pass of rev(A B)
// The turn of H as one pass (cross_turn_kernel): reads a.W (the product's workspace, sub-transformed), writes a.dst (the
// workspace of T = rev(P) mod x^(m-1), cross stages done).  Returns false when the two transforms need more than one cross
// pass each (the caller then runs the two passes).
template <class M>
static bool launch_cross_turn(rs_ctx *ctx, CrossArgs a, size_t ncols, int logB, const ColPlansT<M> &cp, hipStream_t st) {
  using CPS = ColPlansT<M>;
  constexpr bool FP = std::is_same<M, Mod>::value;
  const int R = a.logtot - logB;
  const int maxr = FP ? std::max(1, std::min(6, g_witness_cross_maxr)) : 4;
  if (!g_witness_h_turn || R < 1 || R > maxr || a.logsub != a.logtot || logB < 8) return false;
  if ((((uintptr_t)a.W | (uintptr_t)a.dst) & 15) != 0) return false;
  // cross_turn_kernel indexes the product at i0 = 2m - 2 - j - c >= 0 and shifts by E - 1 - (i0 >> logB) >= 0: holds for
  // M = next_pow2(m) (2m - 2 >= M >= B, 2m - 2 < 2M = 2^logtot) -- enforced, not assumed (round-5 advice): else the two passes
  if (2 * (long long)a.m - 2 < ((long long)1 << logB) || 2 * (long long)a.m - 2 >= ((long long)1 << a.logtot)) return false;
  const size_t B = (size_t)1 << logB;
  const bool pair = R <= 5;
  const unsigned gx = (unsigned)std::max<size_t>(1, std::min<size_t>((B / (pair ? 2 : 1)) / 256, 1024));
  const dim3 grid(gx, (unsigned)ncols);
  const double n = (double)((size_t)1 << a.logtot);
  static const char *const names[7] = {"", "cross_turn_kernel<1", "cross_turn_kernel<2", "cross_turn_kernel<3", "cross_turn_kernel<4", "cross_turn_kernel<5",
                                       "cross_turn_kernel<6"};
  ProfScope prof(ctx, st, names[R], (double)ncols * 8.0 * 2.0 * n, (double)ncols * (ntt_fp64(n, R) + ntt_fp64(n, R - 1)));
  switch (R) {
    case 1: hipLaunchKernelGGL((cross_turn_kernel<1, CPS, 2>), grid, dim3(256), 0, st, a, cp); break;
    case 2: hipLaunchKernelGGL((cross_turn_kernel<2, CPS, 2>), grid, dim3(256), 0, st, a, cp); break;
    case 3: hipLaunchKernelGGL((cross_turn_kernel<3, CPS, 2>), grid, dim3(256), 0, st, a, cp); break;
    case 4: hipLaunchKernelGGL((cross_turn_kernel<4, CPS, 2>), grid, dim3(256), 0, st, a, cp); break;
    case 5:
      if constexpr (FP) hipLaunchKernelGGL((cross_turn_kernel<5, CPS, 2>), grid, dim3(256), 0, st, a, cp);
      break;
    default:
      if constexpr (FP) hipLaunchKernelGGL((cross_turn_kernel<6, CPS, 1>), grid, dim3(256), 0, st, a, cp);
      break;
  }
  RS_HIP(hipGetLastError());
  return true;
}

int g_witness_level_turn = 1;  // tuning knob "witness_level_turn": fuse the last inverse cross pass of tree level l with the first forward pass of level l + 1
// The turn between tree levels l = a.l and l + 1 as one pass (cross_level_turn_kernel) on the workspace a.W and the columns
// a.dst; false when the two levels differ in block size or need more than one cross pass each.
template <class M>
static bool launch_level_turn(rs_ctx *ctx, CrossArgs a, size_t ncols, int logB, int logB_next, const ColPlansT<M> &cp, hipStream_t st) {
  using CPS = ColPlansT<M>;
  constexpr bool FP = std::is_same<M, Mod>::value;
  const int RL = a.l - logB;
  const int maxr = FP ? std::max(1, std::min(6, g_witness_cross_maxr)) : 4;
  if (!g_witness_level_turn || logB != logB_next || RL < 1 || RL + 1 > maxr || logB < 8 || a.l + 1 > a.logtot) return false;
  if ((((uintptr_t)a.W | (uintptr_t)a.dst) & 15) != 0) return false;
  const bool pair = RL <= 3;
  const size_t groups = (((size_t)1 << a.logtot) >> (a.l + 1)) * (((size_t)1 << logB) / (pair ? 2 : 1));
  const unsigned gx = (unsigned)std::max<size_t>(1, std::min<size_t>(groups / 256, 1024));
  const dim3 grid(gx, (unsigned)ncols);
  const double n = (double)((size_t)1 << a.logtot);
  static const char *const names[6] = {"", "cross_level_turn_kernel<1", "cross_level_turn_kernel<2", "cross_level_turn_kernel<3", "cross_level_turn_kernel<4",
                                       "cross_level_turn_kernel<5"};
  // words per coefficient position of the column: workspace read + written, the children's lower halves read, the left child written
  ProfScope prof(ctx, st, names[RL], (double)ncols * 8.0 * 3.0 * n, (double)ncols * (ntt_fp64(n, RL) + ntt_fp64(n / 2.0, RL)));
  switch (RL) {
    case 1: hipLaunchKernelGGL((cross_level_turn_kernel<1, CPS, 2>), grid, dim3(256), 0, st, a, cp); break;
    case 2: hipLaunchKernelGGL((cross_level_turn_kernel<2, CPS, 2>), grid, dim3(256), 0, st, a, cp); break;
    case 3: hipLaunchKernelGGL((cross_level_turn_kernel<3, CPS, 2>), grid, dim3(256), 0, st, a, cp); break;
    case 4:
      if constexpr (FP) hipLaunchKernelGGL((cross_level_turn_kernel<4, CPS, 1>), grid, dim3(256), 0, st, a, cp);
      break;
    default:
      if constexpr (FP) hipLaunchKernelGGL((cross_level_turn_kernel<5, CPS, 1>), grid, dim3(256), 0, st, a, cp);
      break;
  }
  RS_HIP(hipGetLastError());
  return true;
}

// tuning knob "witness_tree_fwd": the tile kernel runs the forward cross stages of the first level above the tiles.  OFF by
// default -- measured (profiles/r05_knob_ab_tree_once.txt): it removes a 9.8 ms pass and costs the tile kernel 16 ms (176 ->
// 192 ms per headline proof): one workgroup per CU has nothing to hide its epilogue's LDS reads and stores behind.
int g_witness_tree_fwd = 0;
// Can the wide 2^14 tile kernel run the forward cross stages of level 15 (2 or 3 of them: blocks of 2^13 / 2^12)?
template <class M>
static bool tree_fwd_stages(const WitnessPlan *P) {
  if constexpr (!std::is_same<M, Mod>::value) return false;
  const int logM = P->logM, logT = std::min(g_witness_lds_logM, logM);
  if (!g_witness_tree_fwd || !(logT == 13 && logM >= 15 && g_witness_tree_ct == 2 && g_witness_tree_log >= 14)) return false;
  const int rf = 15 - sub_block_log<M>(logT, 15);
  return rf == 2 || rf == 3;
}

// multi-pass interpolation of `ncols` columns X[ncols][M] in place; W: workspace [ncols][2M].
// phases: 1 = values -> Newton coefficients, 2 = the product tree's tiles (in place on X: no workspace, so the caller may run
// it ONCE over all the columns of a chunk instead of per workspace-sized sub-chunk), 4 = the levels above the tiles.
template <class M>
// tree_fwd (phases 2 and 4 must agree): the tile kernel of the right children also runs the forward cross stages of the first
// level above the tiles, into W as [ncols][M] (tree_fwd_stages() says whether it can) -- that level's source pass is skipped.
static void big_interp(rs_ctx *ctx, const WitnessPlan *P, const ColPlansT<M> &cp, typename ArithOf<M>::T *X, typename ArithOf<M>::T *W,
                       size_t ncols, size_t col0, size_t S, size_t spl, int limb0, hipStream_t st, int phases = 7, bool tree_fwd = false) {
  using T = typename ArithOf<M>::T;
  constexpr bool FP = std::is_same<M, Mod>::value;
  const int logM = P->logM, logT = std::min(g_witness_lds_logM, logM);
  int logB = sub_block_log<M>(logT, logM + 1);  // block of the rooted sub-transforms
  const size_t Mlen = P->M;
  CrossArgs a{};
  a.W = W;
  a.src = X;
  a.dst = X;
  a.logM = logM;
  a.l = 1;
  a.m = (int)P->m;
  a.S = (unsigned)S;
  a.slots_per_limb = (unsigned)spl;
  a.col0 = col0;
  TabPtrs tp{};
  if (phases & 1) {
    // values -> Newton coefficients: one cyclic convolution of length 2M
    a.logtot = a.logsub = logM + 1;
    launch_cross<false, CS_SCALE_PAD, M>(ctx, a, ncols, logB, cp, st);
    for (int i = limb0; i < ctx->L; i++) tp.t[i - limb0] = P->limb[i].d_ehat;
    launch_sub<2, M>(ctx, W, ncols, col0, logM + 1, logM + 1, logB, &tp, (2 * Mlen) >> logB, S, spl, cp, st);
    launch_cross<true, CD_TAKE_LOW, M>(ctx, a, ncols, logB, cp, st);
  }
  // product tree: levels <= logTree inside LDS tiles (the wide kernel takes 2^14 tiles: one multi-pass level less)
  int logTree = logT;
  if constexpr (FP) {
    if (logT == 13 && logM >= 15 && g_witness_tree_ct == 2 && g_witness_tree_log >= 14) logTree = 14;
    const int rf = tree_fwd ? logTree + 1 - sub_block_log<M>(logT, logTree + 1) : 0;
    if (phases & 2) launch_tree_tiles(ctx, X, ncols, col0, logM, logTree, S, spl, cp, st, false, tree_fwd ? W : nullptr, rf);
  } else {
    if (phases & 2) launch_tree_tiles_generic<M>(ctx, X, ncols, col0, logM, logT, S, spl, cp, st);
  }
  if (!(phases & 4)) return;
  // levels above: F_node = F_left + D_left * F_right with multi-pass transforms of length 2^l
  a.logtot = logM;
  bool fwd_done = tree_fwd;  // the forward cross pass of this level was run by the previous level's turn (or by the tile kernel)
  for (int l = logTree + 1; l <= logM; l++) {
    a.l = l;
    a.logsub = l;
    logB = sub_block_log<M>(logT, l);
    if (!fwd_done) launch_cross<false, CS_FILL_RIGHT, M>(ctx, a, ncols, logB, cp, st);
    fwd_done = false;
    for (int i = limb0; i < ctx->L; i++) tp.t[i - limb0] = static_cast<const T *>(P->limb[i].d_dhat) + (size_t)l * Mlen;
    launch_sub<2, M>(ctx, W, ncols, col0, logM, l, logB, &tp, Mlen >> logB, S, spl, cp, st);
    if (l == logM) {
      launch_cross<true, CD_COMBINE_CANON, M>(ctx, a, ncols, logB, cp, st);
    } else {
      // this level's last inverse cross pass and the next level's first forward pass as one pass over memory, when the
      // two levels share their block size (cross_level_turn_kernel); else the inverse pass alone
      fwd_done = launch_level_turn<M>(ctx, a, ncols, logB, sub_block_log<M>(logT, l + 1), cp, st);
      if (!fwd_done) launch_cross<true, CD_COMBINE, M>(ctx, a, ncols, logB, cp, st);
    }
  }
}

// multi-pass H = quo(A*B, Z) (+ ZK patch) for `ncols` columns; W1, W2: workspaces [ncols][2M]
template <class M>
static void big_h(rs_ctx *ctx, const WitnessPlan *P, const ColPlansT<M> &cp, const typename ArithOf<M>::T *A,
                  const typename ArithOf<M>::T *B, typename ArithOf<M>::T *H, typename ArithOf<M>::T *W1, typename ArithOf<M>::T *W2,
                  size_t ncols, size_t col0, size_t S, size_t spl, const uint64_t *d1, const uint64_t *d2, const uint64_t *d3,
                  const ColMap &cm, int limb0, hipStream_t st) {
  const int logM = P->logM, logB = sub_block_log<M>(std::min(g_witness_lds_logM, logM), logM + 1);
  const size_t Mlen = P->M;
  CrossArgs a{};
  a.logM = logM;
  a.l = 1;
  a.m = (int)P->m;
  a.S = (unsigned)S;
  a.slots_per_limb = (unsigned)spl;
  a.col0 = col0;
  a.logtot = a.logsub = logM + 1;
  TabPtrs tp{};
  // W1 = spectrum of A; W2 = A * B (spectrum product inside the sub-transform kernel of B)
  a.W = W1;
  a.src = A;
  launch_cross<false, CS_PAD_CENTER, M>(ctx, a, ncols, logB, cp, st);
  launch_sub<0, M>(ctx, W1, ncols, col0, logM + 1, logM + 1, logB, nullptr, 1, S, spl, cp, st);
  a.W = W2;
  a.src = B;
  launch_cross<false, CS_PAD_CENTER, M>(ctx, a, ncols, logB, cp, st);
  tp.t[0] = W1;
  launch_sub<3, M>(ctx, W2, ncols, col0, logM + 1, logM + 1, logB, &tp, 1, S, spl, cp, st);
  // U = rev(P) * rev(Z)^-1 mod x^(m-1): the product's last inverse cross pass and the first forward pass of its reversal are
  // one pass over memory when each transform has a single cross pass (cross_turn_kernel); otherwise the two passes
  a.dst = W1;
  const bool turned = launch_cross_turn<M>(ctx, a, ncols, logB, cp, st);  // a.W = W2 -> W1
  if (!turned) launch_cross<true, CD_PLAIN, M>(ctx, a, ncols, logB, cp, st);
  a.W = W1;
  a.src = W2;
  if (!turned) launch_cross<false, CS_REV_TRUNC, M>(ctx, a, ncols, logB, cp, st);
  for (int i = limb0; i < ctx->L; i++) tp.t[i - limb0] = P->limb[i].d_shat;
  launch_sub<2, M>(ctx, W1, ncols, col0, logM + 1, logM + 1, logB, &tp, (2 * Mlen) >> logB, S, spl, cp, st);
  a.dst = H;
  if (!d1) {  // d1 = d2 = d3 = 0 (groth16.tcc:82-84): nothing to patch, the last pass writes the finished column
    launch_cross<true, CD_H_FINISH_CANON, M>(ctx, a, ncols, logB, cp, st);
    return;
  }
  launch_cross<true, CD_H_FINISH, M>(ctx, a, ncols, logB, cp, st);
  const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((ncols * Mlen + 255) / 256, 256 * 16));
  ProfScope prof(ctx, st, "h_patch_kernel", (double)ncols * (double)Mlen * (d1 ? 32.0 : 16.0), d1 ? 24.0 * (double)ncols * (double)Mlen : 0.0);
  hipLaunchKernelGGL(h_patch_kernel<ColPlansT<M>>, dim3(blocks), dim3(256), 0, st, H, A, B, logM, (int)P->m, ncols, col0, (unsigned)S,
                     (unsigned)spl, cp, d1, d2, d3, cm);
  RS_HIP(hipGetLastError());
}

// H on a coset, when C's coefficients are at hand (Rinocchio keeps C_mid, rinocchio.tcc:75-190; ringGroth16 never
// interpolates C and takes big_h): with the M points g w^i, none a root of Z,
//     H(g w^i) = (A(g w^i) B(g w^i) - C(g w^i)) / Z(g w^i),   deg H <= m - 2 < M,
// so H is the inverse coset transform of that quotient: FOUR transforms of length M (three forward, one inverse, the
// pointwise step inside the sub-transform kernel of B) instead of big_h's five of length 2M.  The division is exact in
// Z_q, so H is the polynomial the reference's long division (util/polynomials.tcc:62-81) returns; the ZK patch follows as
// in big_h.  W1, W2: workspaces [ncols][2M] (W1 holds the spectra of A and of C, W2 that of B and the result).
template <class M>
static void big_h_coset(rs_ctx *ctx, const WitnessPlan *P, const ColPlansT<M> &cp, const typename ArithOf<M>::T *A,
                        const typename ArithOf<M>::T *B, const typename ArithOf<M>::T *Cc, typename ArithOf<M>::T *H,
                        typename ArithOf<M>::T *W1, typename ArithOf<M>::T *W2, size_t ncols, size_t col0, size_t S, size_t spl,
                        const uint64_t *d1, const uint64_t *d2, const uint64_t *d3, const ColMap &cm, int limb0, hipStream_t st) {
  using T = typename ArithOf<M>::T;
  const int logM = P->logM, logB = sub_block_log<M>(std::min(g_witness_lds_logM, logM), logM);
  const size_t Mlen = P->M;
  T *W3 = W1 + ncols * Mlen;  // the second half of the [ncols][2M] workspace
  CrossArgs a{};
  a.logM = logM;
  a.l = 1;
  a.m = (int)P->m;
  a.S = (unsigned)S;
  a.slots_per_limb = (unsigned)spl;
  a.col0 = col0;
  a.logtot = a.logsub = logM;
  TabPtrs tp{};
  const T *srcs[3] = {A, Cc, B};
  T *dsts[3] = {W1, W3, W2};
  for (int k = 0; k < 3; k++) {
    a.W = dsts[k];
    a.src = srcs[k];
    launch_cross<false, CS_COSET, M>(ctx, a, ncols, logB, cp, st);
    if (k < 2) launch_sub<0, M>(ctx, dsts[k], ncols, col0, logM, logM, logB, nullptr, 1, S, spl, cp, st);
  }
  for (int i = limb0; i < ctx->L; i++) tp.t[i - limb0] = P->limb[i].d_cos_z;
  tp.w1 = W1;
  tp.w3 = W3;
  launch_sub<4, M>(ctx, W2, ncols, col0, logM, logM, logB, &tp, Mlen >> logB, S, spl, cp, st);
  a.W = W2;
  a.dst = H;
  if (!d1) {
    launch_cross<true, CD_H_COSET_CANON, M>(ctx, a, ncols, logB, cp, st);
    return;
  }
  launch_cross<true, CD_H_COSET, M>(ctx, a, ncols, logB, cp, st);
  const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((ncols * Mlen + 255) / 256, 256 * 16));
  ProfScope prof(ctx, st, "h_patch_kernel", (double)ncols * (double)Mlen * 32.0, 24.0 * (double)ncols * (double)Mlen);
  hipLaunchKernelGGL(h_patch_kernel<ColPlansT<M>>, dim3(blocks), dim3(256), 0, st, H, A, B, logM, (int)P->m, ncols, col0, (unsigned)S,
                     (unsigned)spl, cp, d1, d2, d3, cm);
  RS_HIP(hipGetLastError());
}

int g_witness_tree_once = 1;  // tuning knob "witness_tree_once": the product tree's tiles in one launch per chunk of columns
int g_witness_big_ws_mib = 6 * 1024;  // tuning knob "witness_big_ws_mib": the two [cols][2M] workspaces of a multi-pass sub-chunk
static size_t big_chunk_cols(const WitnessPlan *P) {
  // two [cols][2M] workspaces within ~6 GiB
  const size_t per_col = 4 * P->M * sizeof(double);
  return std::max<size_t>(1, ((size_t)g_witness_big_ws_mib << 20) / per_col);
}

// Columns handled by the M-tile kernels (fused Newton + tree, h_tile): 2^10 .. 2^13 at two workgroups
// per CU, and 2^14 (a 136 KiB tile, one 1024-thread workgroup per CU) when the tile knob is at its
// natural setting -- one launch instead of the multi-pass path.
static bool single_tile_ok(int logM) {
  if (logM < 10) return false;
  return logM <= g_witness_lds_logM || (logM == 14 && g_witness_lds_logM == 13);
}

// ---- block-convolution path: host side ---------------------------------------------------------------------
template <int SRC, int YK, int DST, class M>
static void bc_conv(rs_ctx *ctx, BcArgs a, size_t ncols, size_t per_unit, const ColPlansT<M> &cp, hipStream_t st) {
  using CPS = ColPlansT<M>;
  const size_t B2 = (size_t)1 << a.bcLog, lds = padded_len(B2) * sizeof(uint64_t);
  const int thr = col_threads(B2);
  const double nfwd = (double)ncols * a.units * a.nxb, nmac = (double)ncols * a.units * a.nk;
  double pairs = 0;  // pointwise block products
  for (int k = 0; k < a.nk; k++) pairs += std::min(k, a.nxb - 1) - std::max(0, k - a.nyb + 1) + 1;
  {
    ProfScope prof(ctx, st, "bc_fwd_kernel", nfwd * (double)B2 * 12.0, nfwd * ntt_fp64((double)B2, a.bcLog));
    set_max_dyn_lds((const void *)bc_fwd_kernel<SRC, CPS>, (int)lds);
    hipLaunchKernelGGL((bc_fwd_kernel<SRC, CPS>), dim3((unsigned)(ncols * a.units * a.nxb)), dim3(thr), lds, st, a, cp);
  }
  {
    ProfScope prof(ctx, st, "bc_mac_kernel", nmac * (double)B2 * 8.0 + (double)ncols * a.units * pairs * (double)B2 * 8.0,
                   nmac * ntt_fp64((double)B2, a.bcLog) + (double)ncols * a.units * pairs * (double)B2 * 7.0);
    set_max_dyn_lds((const void *)bc_mac_kernel<YK, CPS>, (int)lds);
    hipLaunchKernelGGL((bc_mac_kernel<YK, CPS>), dim3((unsigned)(ncols * a.units * a.nk)), dim3(thr), lds, st, a, cp);
  }
  {
    const size_t total = ncols * a.units * per_unit;
    ProfScope prof(ctx, st, "bc_out_kernel", (double)total * 24.0, (double)total * 3.0);
    hipLaunchKernelGGL((bc_out_kernel<DST, CPS>), dim3((unsigned)std::max<size_t>(1, std::min<size_t>((total + 255) / 256, 256 * 32))), dim3(256), 0,
                       st, a, cp, ncols, per_unit);
  }
  RS_HIP(hipGetLastError());
}

// columns per chunk such that the block workspaces (spectra + pair products, up to ~8M words per column) stay within ~6 GiB
static size_t bc_chunk_cols(const WitnessPlan *P) {
  return std::max<size_t>(1, ((size_t)g_witness_big_ws_mib << 20) / ((P->bc2 ? 12 : 10) * P->M * sizeof(double)));
}

template <class M>
static void bc_interp(rs_ctx *ctx, const WitnessPlan *P, const ColPlansT<M> &cp, typename ArithOf<M>::T *X, size_t ncols, size_t col0,
                      size_t S, size_t spl, hipStream_t st) {
  using T = typename ArithOf<M>::T;
  const int logM = P->logM, bc = P->bcLog;
  const size_t Mlen = P->M, B = (size_t)1 << (bc - 1);
  T *Xhat = (T *)ws_get(ctx, 12, ncols * 2 * Mlen * sizeof(T));
  T *Wc = (T *)ws_get(ctx, 13, ncols * 2 * Mlen * sizeof(T));
  BcArgs a{};
  a.src = X;
  a.dst = X;
  a.Xhat = Xhat;
  a.Wc = Wc;
  a.bcLog = bc;
  a.logM = logM;
  a.m = (int)P->m;
  a.col0 = col0;
  a.S = (unsigned)S;
  a.slots_per_limb = (unsigned)spl;
  // values -> Newton coefficients: low M terms of (y_k / k!) * ((-1)^k / k!)
  a.units = 1;
  a.nxb = a.nyb = a.nk = (int)(Mlen / B);
  bc_conv<BS_SCALE, BY_E, BD_NEWTON, M>(ctx, a, ncols, Mlen, cp, st);
  // product tree: levels <= bc inside LDS tiles (transforms of length <= 2^bc) ...
  if constexpr (std::is_same<M, Mod>::value) {
    // ... through the wide tile kernel where it exists (2^13 / 2^14 tiles: the recipe primes of the headline shape have
    // 2-adicity 14, so the whole 2^14 tile of tree_wide_kernel<14> is available to them)
    if ((bc == 13 || bc == 14) && g_witness_tree_ct == 2 && (bc == 13 || g_witness_tree_log >= 14))
      launch_tree_tiles(ctx, X, ncols, col0, logM, bc, S, spl, cp, st);
    else
      launch_tree_tiles_generic<M>(ctx, X, ncols, col0, logM, bc, S, spl, cp, st);
  } else {
    launch_tree_tiles_generic<M>(ctx, X, ncols, col0, logM, bc, S, spl, cp, st);
  }
  // ... and above: F_node = F_left + (x^h + d) * F_right with d * F_right as a block convolution
  for (int l = bc + 1; l <= logM; l++) {
    a.l = l;
    a.units = (int)(Mlen >> l);
    a.nxb = a.nyb = (int)(((size_t)1 << (l - 1)) / B);
    a.nk = 2 * a.nxb - 1;
    if (l == logM)
      bc_conv<BS_RIGHT, BY_D, BD_COMBINE_CANON, M>(ctx, a, ncols, (size_t)1 << l, cp, st);
    else
      bc_conv<BS_RIGHT, BY_D, BD_COMBINE, M>(ctx, a, ncols, (size_t)1 << l, cp, st);
  }
}

template <class M>
static void bc_h(rs_ctx *ctx, const WitnessPlan *P, const ColPlansT<M> &cp, const typename ArithOf<M>::T *A, const typename ArithOf<M>::T *Bc,
                 typename ArithOf<M>::T *H, size_t ncols, size_t col0, size_t S, size_t spl, const uint64_t *d1, const uint64_t *d2,
                 const uint64_t *d3, const ColMap &cm, hipStream_t st) {
  using T = typename ArithOf<M>::T;
  using CPS = ColPlansT<M>;
  const int logM = P->logM, bc = P->bcLog;
  const size_t Mlen = P->M, B = (size_t)1 << (bc - 1), B2 = 2 * B, nb = Mlen / B;
  T *Xhat = (T *)ws_get(ctx, 12, ncols * 2 * Mlen * sizeof(T));
  T *Wc = (T *)ws_get(ctx, 13, ncols * 4 * Mlen * sizeof(T));
  T *Yhat = (T *)ws_get(ctx, 6, ncols * 2 * Mlen * sizeof(T));
  T *Pbuf = (T *)ws_get(ctx, 15, ncols * 2 * Mlen * sizeof(T));
  BcArgs a{};
  a.bcLog = bc;
  a.logM = logM;
  a.m = (int)P->m;
  a.col0 = col0;
  a.S = (unsigned)S;
  a.slots_per_limb = (unsigned)spl;
  a.units = 1;
  a.nxb = a.nyb = (int)nb;
  // spectra of B's blocks (the "other operand" of the data x data product): a forward pass on its own
  {
    BcArgs b = a;
    b.src = Bc;
    b.Xhat = Yhat;
    const size_t lds = padded_len(B2) * sizeof(uint64_t);
    ProfScope prof(ctx, st, "bc_fwd_kernel", (double)ncols * nb * (double)B2 * 12.0, (double)ncols * nb * ntt_fp64((double)B2, bc));
    set_max_dyn_lds((const void *)bc_fwd_kernel<BS_CENTER, CPS>, (int)lds);
    hipLaunchKernelGGL((bc_fwd_kernel<BS_CENTER, CPS>), dim3((unsigned)(ncols * nb)), dim3(col_threads(B2)), lds, st, b, cp);
  }
  // P = A * B, 2M coefficients
  a.src = A;
  a.Xhat = Xhat;
  a.Yhat = Yhat;
  a.Wc = Wc;
  a.dst = Pbuf;
  a.nk = 2 * (int)nb - 1;
  bc_conv<BS_CENTER, BY_DATA, BD_PLAIN_SCALED, M>(ctx, a, ncols, 2 * Mlen, cp, st);
  // U = rev(P) * rev(Z)^-1 mod x^(m-1);  H_j = U_{m-2-j}
  a.src = Pbuf;
  a.dst = H;
  a.nk = (int)nb;
  bc_conv<BS_REVTRUNC, BY_S, BD_HFIN, M>(ctx, a, ncols, Mlen, cp, st);
  const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((ncols * Mlen + 255) / 256, 256 * 16));
  ProfScope prof(ctx, st, "h_patch_kernel", (double)ncols * (double)Mlen * (d1 ? 32.0 : 16.0), d1 ? 24.0 * (double)ncols * (double)Mlen : 0.0);
  hipLaunchKernelGGL(h_patch_kernel<CPS>, dim3(blocks), dim3(256), 0, st, H, A, Bc, logM, (int)P->m, ncols, col0, (unsigned)S, (unsigned)spl, cp,
                     d1, d2, d3, cm);
  RS_HIP(hipGetLastError());
}

// ---- two-dimensional block convolutions: host side ----------------------------------------------------------
// workspaces: Wy [ncols][2M] and Ws [ncols][4M] words per convolution in flight (+ the same again and a [ncols][2M]
// product buffer for H); bc_chunk_cols keeps a chunk of columns within ~6 GiB
static void bc2_interp(rs_ctx *ctx, const WitnessPlan *P, const ColPlans &cp, double *X, size_t ncols, size_t col0, size_t S, size_t spl,
                       int limb0, hipStream_t st, int phases = 7) {  // phases: as big_interp
  const int logM = P->logM;
  const size_t Mlen = P->M;
  Bc2Args a{};
  a.src = X;
  a.dst = X;
  if (phases & 5) {  // the tree tiles work in place: no workspace (and `ncols` may then be a whole chunk)
    a.Wy = (double *)ws_get(ctx, 12, ncols * 2 * Mlen * sizeof(double));
    a.Ws = (double *)ws_get(ctx, 13, ncols * 4 * Mlen * sizeof(double));
  }
  a.logM = logM;
  a.m = (int)P->m;
  a.col0 = col0;
  a.S = (unsigned)S;
  a.slots_per_limb = (unsigned)spl;
  TabPtrs tp{};
  if (phases & 1) {
    // values -> Newton coefficients: low M terms of (y_k / k!) * ((-1)^k / k!)
    a.units = 1;
    for (int i = limb0; i < ctx->L; i++) tp.t[i - limb0] = P->limb[i].d_b2_e;
    bc2_conv<BS_SCALE, BD_NEWTON, 2>(ctx, a, logM + 1 - BC2_LOGB, ncols, &tp, nullptr, cp, st);
  }
  // product tree: levels <= 14 inside LDS tiles, the levels above as block convolutions F_node = F_left + (x^h + d) * F_right
  if (phases & 2)
    launch_tree_tiles(ctx, X, ncols, col0, logM, (g_witness_tree_ct == 2 && g_witness_tree_log >= 14) ? 14 : 13, S, spl, cp, st);
  if (!(phases & 4)) return;
  const int first = (g_witness_tree_ct == 2 && g_witness_tree_log >= 14) ? 15 : 14;
  bool fwd_done = false;  // this level's transform across blocks was run by the previous level's turn
  for (int l = first; l <= logM; l++) {
    a.l = l;
    a.units = (int)(Mlen >> l);
    for (int i = limb0; i < ctx->L; i++)
      tp.t[i - limb0] = static_cast<const double *>(P->limb[i].d_b2_d) + (size_t)(l - P->bcLog - 1) * 2 * Mlen;
    if (l == logM) {
      bc2_conv<BS_RIGHT, BD_COMBINE_CANON, 2>(ctx, a, l - BC2_LOGB, ncols, &tp, nullptr, cp, st, fwd_done);
      fwd_done = false;
      continue;
    }
    // level l's inverse transform across blocks + sink and level l + 1's source + forward transform as one pass
    // (bc2_level_turn_kernel; one-level transforms: the parent has at most 32 blocks)
    const int logYc = l - BC2_LOGB;
    const bool turn = g_witness_level_turn && logYc >= 2 && logYc <= 4 && (size_t)ncols * (size_t)(a.units / 2) <= 65535;
    bc2_conv<BS_RIGHT, BD_COMBINE, 2>(ctx, a, logYc, ncols, &tp, nullptr, cp, st, fwd_done, turn);
    fwd_done = turn;
    if (turn) {
      const size_t cpn = ncols * (size_t)(a.units / 2), Yp = (size_t)2 << logYc;
      const dim3 grid((unsigned)(BC2_B / 2 / 256), (unsigned)cpn);
      // words per parent and position: both children's spectra read (2 Yc x 2), their lower halves... the children read, the left written, Wy written
      ProfScope prof(ctx, st, "bc2_level_turn_kernel", (double)cpn * BC2_B * 8.0 * (2.0 * Yp + 1.5 * Yp + Yp),
                     (double)cpn * BC2_B * (2.0 * ntt_fp64((double)(Yp / 2), logYc) * 2.0 + ntt_fp64((double)Yp, logYc + 1)));
      switch (logYc) {
        case 2: hipLaunchKernelGGL((bc2_level_turn_kernel<2>), grid, dim3(256), 0, st, a, cp); break;
        case 3: hipLaunchKernelGGL((bc2_level_turn_kernel<3>), grid, dim3(256), 0, st, a, cp); break;
        default: hipLaunchKernelGGL((bc2_level_turn_kernel<4>), grid, dim3(256), 0, st, a, cp); break;
      }
      RS_HIP(hipGetLastError());
    }
  }
}

static void bc2_h(rs_ctx *ctx, const WitnessPlan *P, const ColPlans &cp, const double *A, const double *Bc, double *H, size_t ncols,
                  size_t col0, size_t S, size_t spl, const uint64_t *d1, const uint64_t *d2, const uint64_t *d3, const ColMap &cm, int limb0,
                  hipStream_t st) {
  const int logM = P->logM, logY = logM + 1 - BC2_LOGB;
  const size_t Mlen = P->M;
  double *Wy = (double *)ws_get(ctx, 12, ncols * 2 * Mlen * sizeof(double));
  double *Ws = (double *)ws_get(ctx, 13, ncols * 4 * Mlen * sizeof(double));
  double *WsA = (double *)ws_get(ctx, 6, ncols * 4 * Mlen * sizeof(double));
  double *Pbuf = (double *)ws_get(ctx, 15, ncols * 2 * Mlen * sizeof(double));
  Bc2Args a{};
  a.logM = logM;
  a.m = (int)P->m;
  a.col0 = col0;
  a.S = (unsigned)S;
  a.slots_per_limb = (unsigned)spl;
  a.units = 1;
  a.Wy = Wy;
  // the two-dimensional spectrum of A ...
  a.src = A;
  a.Ws = WsA;
  bc2_conv<BS_CENTER, BD_PLAIN_SCALED, 0>(ctx, a, logY, ncols, nullptr, nullptr, cp, st);
  // ... P = A * B, 2M coefficients
  a.src = Bc;
  a.Ws = Ws;
  a.dst = Pbuf;
  // the turn of H (bc2_h_turn_kernel): the product's inverse transform across blocks and the forward one of its reversal as
  // one pass, the 2M-word product buffer neither written nor read (one-level transforms of at most 16 blocks: M <= 2^16)
  // (the turn kernel reverses around 2m - 2: needs B <= 2m - 2 < 2M, true for M = next_pow2(m) -- checked, as in launch_cross_turn)
  const bool turn = g_witness_h_turn && logY >= 2 && logY <= 4 && 2 * (long long)P->m - 2 >= (long long)BC2_B &&
                    2 * (long long)P->m - 2 < ((long long)2 << logM);
  bc2_conv<BS_CENTER, BD_PLAIN_SCALED, 3>(ctx, a, logY, ncols, nullptr, WsA, cp, st, false, turn);
  if (turn) {
    const size_t Y = (size_t)1 << logY;
    const dim3 grid((unsigned)(BC2_B / 2 / 256), (unsigned)ncols);
    ProfScope prof(ctx, st, "bc2_h_turn_kernel", (double)ncols * BC2_B * 8.0 * (2.0 * Y + Y), (double)ncols * BC2_B * 3.0 * ntt_fp64((double)Y, logY));
    switch (logY) {
      case 2: hipLaunchKernelGGL((bc2_h_turn_kernel<2>), grid, dim3(256), 0, st, a, cp); break;
      case 3: hipLaunchKernelGGL((bc2_h_turn_kernel<3>), grid, dim3(256), 0, st, a, cp); break;
      default: hipLaunchKernelGGL((bc2_h_turn_kernel<4>), grid, dim3(256), 0, st, a, cp); break;
    }
    RS_HIP(hipGetLastError());
  }
  // U = rev(P) * rev(Z)^-1 mod x^(m-1);  H_j = U_{m-2-j}
  TabPtrs tp{};
  for (int i = limb0; i < ctx->L; i++) tp.t[i - limb0] = P->limb[i].d_b2_s;
  a.src = Pbuf;
  a.dst = H;
  bc2_conv<BS_REVTRUNC, BD_HFIN, 2>(ctx, a, logY, ncols, &tp, nullptr, cp, st, turn);
  const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((ncols * Mlen + 255) / 256, 256 * 16));
  ProfScope prof(ctx, st, "h_patch_kernel", (double)ncols * (double)Mlen * (d1 ? 32.0 : 16.0), d1 ? 24.0 * (double)ncols * (double)Mlen : 0.0);
  hipLaunchKernelGGL(h_patch_kernel<ColPlans>, dim3(blocks), dim3(256), 0, st, H, A, Bc, logM, (int)P->m, ncols, col0, (unsigned)S, (unsigned)spl, cp,
                     d1, d2, d3, cm);
  RS_HIP(hipGetLastError());
}

// Interpolate `ncols` columns in place.  Column c belongs to chunk-local limb (c % S) / slots_per_limb
// (several vectors of S columns are batched); cp is shifted so that entry 0 is limb0.
template <class M>
static void launch_interp(rs_ctx *ctx, const WitnessPlan *P, const ColPlansT<M> &cp, typename ArithOf<M>::T *cols, size_t ncols, size_t S,
                          size_t slots_per_limb, int limb0, hipStream_t st) {
  using T = typename ArithOf<M>::T;
  constexpr bool FP = std::is_same<M, Mod>::value;
  if (P->bcLog) {  // a ring prime without a 2M-th root of unity: block convolutions
    const size_t chunk = std::min(ncols, bc_chunk_cols(P));
    if constexpr (FP) {
      if (P->bc2 && chunk < ncols && g_witness_tree_once) {  // the product tree's tiles in one launch (see the full-length path below)
        for (size_t c0 = 0; c0 < ncols; c0 += chunk)
          bc2_interp(ctx, P, cp, cols + c0 * P->M, std::min(chunk, ncols - c0), c0, S, slots_per_limb, limb0, st, 1);
        bc2_interp(ctx, P, cp, cols, ncols, 0, S, slots_per_limb, limb0, st, 2);
        for (size_t c0 = 0; c0 < ncols; c0 += chunk)
          bc2_interp(ctx, P, cp, cols + c0 * P->M, std::min(chunk, ncols - c0), c0, S, slots_per_limb, limb0, st, 4);
        return;
      }
    }
    for (size_t c0 = 0; c0 < ncols; c0 += chunk) {
      if constexpr (FP) {
        if (P->bc2) {
          bc2_interp(ctx, P, cp, cols + c0 * P->M, std::min(chunk, ncols - c0), c0, S, slots_per_limb, limb0, st);
          continue;
        }
      }
      bc_interp<M>(ctx, P, cp, cols + c0 * P->M, std::min(chunk, ncols - c0), c0, S, slots_per_limb, st);
    }
    return;
  }
  if constexpr (FP) {
    if (single_tile_ok(P->logM) && !P->incomplete) {
      // one launch, tile = M, two workgroups per CU: Newton conversion by the two rooted M-point
      // sub-transforms, then the product tree in place
      launch_tree_tiles(ctx, cols, ncols, 0, P->logM, P->logM, S, slots_per_limb, cp, st, true);
      return;
    }
  }
  if (P->logM <= g_witness_lds_logM) {
    // the 2M convolution tile, or the product tree's tile + scratch when M is below the LDS block size
    const size_t lds = std::max(padded_len(2 * P->M), padded_len(tree_scratch_offset((int)P->M) + P->M)) * sizeof(double);
    set_max_dyn_lds((const void *)interp_columns_kernel<ColPlansT<M>>, (int)lds);
    ProfScope prof(ctx, st, "interp_columns_kernel", (double)ncols * (double)P->M * 16.0,
                   (double)ncols * (2.0 * ntt_fp64(2.0 * (double)P->M, P->logM + 1) + 21.0 * (double)P->M + tree_fp64((double)P->M, P->logM)));
    hipLaunchKernelGGL(interp_columns_kernel<ColPlansT<M>>, dim3((unsigned)ncols), dim3(col_threads(2 * P->M)), lds, st, cols, P->logM,
                       (unsigned)S, (unsigned)slots_per_limb, cp);
    RS_HIP(hipGetLastError());
    return;
  }
  const size_t chunk = std::min(ncols, big_chunk_cols(P));
  // the tile kernel also writes level 15's workspace for the right children ([ncols][M], for ALL the columns it covers)
  // when that fits beside everything else (24 GiB; a configs[3] rank is tight)
  const bool tfwd = tree_fwd_stages<M>(P) && ncols * P->M * sizeof(double) <= ((size_t)24 << 30);
  T *W = (T *)ws_get(ctx, 12, std::max(chunk * 2 * P->M, (tfwd && chunk < ncols && g_witness_tree_once) ? ncols * P->M : 0) * sizeof(double));
  if (chunk < ncols && g_witness_tree_once) {
    // the tiles of the product tree work in place on the columns: ONE launch over all of them between the sub-chunked
    // phases (tile kernels like long launches: 183.5 -> 176 ms per headline proof when every launch covers a whole chunk,
    // profiles/r05_knob_ab_big_ws.txt; the workspace-bound phases keep their 6 GiB sub-chunks, which they prefer)
    for (size_t c0 = 0; c0 < ncols; c0 += chunk)
      big_interp<M>(ctx, P, cp, cols + c0 * P->M, W, std::min(chunk, ncols - c0), c0, S, slots_per_limb, limb0, st, 1);
    big_interp<M>(ctx, P, cp, cols, W, ncols, 0, S, slots_per_limb, limb0, st, 2, tfwd);
    for (size_t c0 = 0; c0 < ncols; c0 += chunk)  // a sub-chunk's levels above the tiles work on its slice of the [ncols][M] workspace
      big_interp<M>(ctx, P, cp, cols + c0 * P->M, W + (tfwd ? c0 * P->M : 0), std::min(chunk, ncols - c0), c0, S, slots_per_limb, limb0, st, 4, tfwd);
    return;
  }
  for (size_t c0 = 0; c0 < ncols; c0 += chunk) {
    const size_t nc = std::min(chunk, ncols - c0);
    big_interp<M>(ctx, P, cp, cols + c0 * P->M, W, nc, c0, S, slots_per_limb, limb0, st, 7, tfwd);
  }
}

// H for the S columns of a chunk (A, B, H: [S][M]); `spl` columns per limb, cm locates d1..d3
template <class M>
static void launch_h(rs_ctx *ctx, const WitnessPlan *P, const ColPlansT<M> &cp, const typename ArithOf<M>::T *A,
                     const typename ArithOf<M>::T *B, typename ArithOf<M>::T *H, size_t S, size_t spl, const uint64_t *d1,
                     const uint64_t *d2, const uint64_t *d3, const ColMap &cm, hipStream_t st,
                     const typename ArithOf<M>::T *Cc = nullptr /* coefficients of C when the call interpolates them: the coset form */) {
  using T = typename ArithOf<M>::T;
  constexpr bool FP = std::is_same<M, Mod>::value;
  const size_t Mlen = P->M;
  if (P->bcLog) {
    const size_t chunk = std::min(S, bc_chunk_cols(P));
    for (size_t c0 = 0; c0 < S; c0 += chunk) {
      if constexpr (FP) {
        if (P->bc2) {
          bc2_h(ctx, P, cp, A + c0 * Mlen, B + c0 * Mlen, H + c0 * Mlen, std::min(chunk, S - c0), c0, S, spl, d1, d2, d3, cm, cm.limb0, st);
          continue;
        }
      }
      bc_h<M>(ctx, P, cp, A + c0 * Mlen, B + c0 * Mlen, H + c0 * Mlen, std::min(chunk, S - c0), c0, S, spl, d1, d2, d3, cm, st);
    }
    return;
  }
  if constexpr (FP) {
    if (single_tile_ok(P->logM) && !P->incomplete) {
      const size_t lds1 = padded_len(Mlen) * sizeof(double);
      const int thr = (int)(Mlen / 16);
      // ten M-point transforms, four pointwise products, the ZK patch
      ProfScope prof(ctx, st, "h_tile_kernel", (double)S * (double)Mlen * 24.0,
                     (double)S * (10.0 * ntt_fp64((double)Mlen, P->logM) + (d1 ? 52.0 : 28.0) * (double)Mlen));
#define RS_H_LAUNCH(KERN)                                                                                            \
  do {                                                                                                               \
    set_max_dyn_lds((const void *)KERN, (int)lds1);         \
    hipLaunchKernelGGL(KERN, dim3((unsigned)S), dim3(thr), lds1, st, A, B, H, P->logM, (int)P->m, (unsigned)spl, cp, \
                       d1, d2, d3, cm);                                                                              \
  } while (0)
      if (thr == 1024) RS_H_LAUNCH((h_tile_kernel<1024, 0>));
      else if (thr == 512 && g_witness_tree_ct) RS_H_LAUNCH((h_tile_kernel<512, 13>));
      else if (thr == 512) RS_H_LAUNCH((h_tile_kernel<512, 0>));
      else if (thr == 256) RS_H_LAUNCH((h_tile_kernel<256, 0>));
      else if (thr == 128) RS_H_LAUNCH((h_tile_kernel<128, 0>));
      else RS_H_LAUNCH((h_tile_kernel<64, 0>));
#undef RS_H_LAUNCH
      RS_HIP(hipGetLastError());
      return;
    }
  }
  if (P->logM <= g_witness_lds_logM) {
    const size_t lds = padded_len(2 * Mlen) * sizeof(double);
    set_max_dyn_lds((const void *)h_columns_kernel<ColPlansT<M>>, (int)lds);
    ProfScope prof(ctx, st, "h_columns_kernel", (double)S * (double)Mlen * 24.0,
                   (double)S * (5.0 * ntt_fp64(2.0 * (double)Mlen, P->logM + 1) + (d1 ? 52.0 : 28.0) * (double)Mlen));
    hipLaunchKernelGGL(h_columns_kernel<ColPlansT<M>>, dim3((unsigned)S), dim3(col_threads(2 * Mlen)), lds, st, A, B, H, P->logM, (int)P->m,
                       (unsigned)spl, cp, d1, d2, d3, cm);
    RS_HIP(hipGetLastError());
    return;
  }
  const size_t chunk = std::min(S, big_chunk_cols(P));
  T *W1 = (T *)ws_get(ctx, 12, chunk * 2 * Mlen * sizeof(double));
  T *W2 = (T *)ws_get(ctx, 13, chunk * 2 * Mlen * sizeof(double));
  for (size_t c0 = 0; c0 < S; c0 += chunk) {
    const size_t nc = std::min(chunk, S - c0);
    if (Cc && g_witness_h_coset && P->limb[cm.limb0].d_cos_z)
      big_h_coset<M>(ctx, P, cp, A + c0 * Mlen, B + c0 * Mlen, Cc + c0 * Mlen, H + c0 * Mlen, W1, W2, nc, c0, S, spl, d1, d2, d3, cm, cm.limb0, st);
    else
      big_h<M>(ctx, P, cp, A + c0 * Mlen, B + c0 * Mlen, H + c0 * Mlen, W1, W2, nc, c0, S, spl, d1, d2, d3, cm, cm.limb0, st);
  }
}

void r1cs_evaluate_run(rs_ctx *ctx, const rs_r1cs *cs, int which, int mode, const uint64_t *d_asg, uint64_t *d_out,
                       hipStream_t st) {
  const size_t S = ctx->ring_words();
  const unsigned by = (unsigned)((S / 2 + 255) / 256);
  if (ctx->use_int)
    hipLaunchKernelGGL(r1cs_eval_kernel<ModI>, dim3((unsigned)cs->m, by), dim3(256), 0, st, cs->d_row_ptr[which], cs->d_col[which],
                       reinterpret_cast<const uint64_t *>(cs->d_coeff[which]), cs->nnz[which], d_asg, d_out, ctx->N, ctx->L, mode,
                       (unsigned)cs->n_inputs, ctx->d_qmod_i, cs->d_pidx[which], reinterpret_cast<const uint64_t *>(cs->d_ptab));
  else
    hipLaunchKernelGGL(r1cs_eval_kernel<Mod>, dim3((unsigned)cs->m, by), dim3(256), 0, st, cs->d_row_ptr[which], cs->d_col[which],
                       cs->d_coeff[which], cs->nnz[which], d_asg, d_out, ctx->N, ctx->L, mode, (unsigned)cs->n_inputs, ctx->d_qmod,
                       cs->d_pidx[which], cs->d_ptab);
  RS_HIP(hipGetLastError());
}

constexpr size_t IO_SHORTCUT_MAX_INPUTS = 64;
// does the witness map of this system compute the io vectors as linear forms of the primary inputs (cs->d_io_* hold them
// after the first witness_run)?
// The shortcut needs slot-constant coefficients on the constant one and on the primary inputs (the L_k are slot constant).
bool witness_io_shortcut(const rs_r1cs *cs) { return cs->n_inputs <= IO_SHORTCUT_MAX_INPUTS && !cs->io_poly; }

// Per-circuit cache for the io shortcut: L_k = interp(column k of X), k = 0 (constant) .. n_inputs.
template <class M_>
static void build_io_cache(rs_ctx *ctx, const rs_r1cs *cs, const WitnessPlan *P, const ColPlansT<M_> &cp, hipStream_t st) {
  using T = typename ArithOf<M_>::T;
  if (cs->io_built) return;
  const size_t m = cs->m, M = P->M, L = (size_t)ctx->L;
  std::vector<T> cols;  // [ncols][L][M] data values
  std::vector<int> hk[3], hc[3];
  int ncols = 0;
  for (int w = 0; w < 3; w++) {
    const size_t z = cs->nnz[w];
    for (size_t k = 0; k <= cs->n_inputs; k++) {
      std::vector<uint64_t> y(L * m, 0);
      bool any = false;
      for (size_t r = 0; r < m; r++)
        for (uint32_t e = cs->h_row_ptr[w][r]; e < cs->h_row_ptr[w][r + 1]; e++)
          if (cs->h_col[w][e] == k)
            for (size_t i = 0; i < L; i++) {
              const uint64_t c = cs->h_coeff[w][i * z + e] % ctx->q[i];
              y[i * m + r] = host::addmod(y[i * m + r], c, ctx->q[i]);
              any = any || c != 0;
            }
      if (!any) continue;
      cols.resize((size_t)(ncols + 1) * L * M, T(0));
      for (size_t i = 0; i < L; i++)
        for (size_t r = 0; r < m; r++) cols[((size_t)ncols * L + i) * M + r] = HostArith<M_>::plain(y[i * m + r], ctx->q[i]);
      hk[w].push_back((int)k);
      hc[w].push_back(ncols);
      ncols++;
    }
  }
  rs_r1cs *mc = const_cast<rs_r1cs *>(cs);
  RS_HIP(hipMalloc(&mc->d_io_cols, std::max<size_t>(1, cols.size()) * sizeof(double)));
  if (ncols) {
    RS_HIP(hipMemcpy(mc->d_io_cols, cols.data(), cols.size() * sizeof(double), hipMemcpyHostToDevice));
    launch_interp<M_>(ctx, P, cp, reinterpret_cast<T *>(mc->d_io_cols), (size_t)ncols * L, L, 1, 0, st);
    RS_HIP(hipStreamSynchronize(st));
  }
  for (int w = 0; w < 3; w++) {
    mc->io_count[w] = (int)hk[w].size();
    mc->io_const_col[w] = -1;
    for (size_t c = 0; c < hk[w].size(); c++)
      if (hk[w][c] == 0) mc->io_const_col[w] = hc[w][c];
    const size_t n = std::max<size_t>(1, hk[w].size());
    RS_HIP(hipMalloc(&mc->d_io_k[w], n * sizeof(int)));
    RS_HIP(hipMalloc(&mc->d_io_c[w], n * sizeof(int)));
    if (!hk[w].empty()) {
      RS_HIP(hipMemcpy(mc->d_io_k[w], hk[w].data(), hk[w].size() * sizeof(int), hipMemcpyHostToDevice));
      RS_HIP(hipMemcpy(mc->d_io_c[w], hc[w].data(), hc[w].size() * sizeof(int), hipMemcpyHostToDevice));
    }
  }
  mc->io_M = M;
  mc->io_built = true;
}

int g_witness_col_budget_mib = 16 * 1024;  // column workspace of one chunk (tuning knob "witness_col_budget_mib")

// One chunk of the witness map: limbs [limb0, limb0 + nl), slots [cm.slot0, cm.slot0 + cm.ns) of each.
template <class M_>
static void witness_chunk(rs_ctx *ctx, const rs_r1cs *cs, WitnessPlan *P, const uint64_t *d_asg, const uint64_t *d1,
                          const uint64_t *d2, const uint64_t *d3, uint64_t *const outs[7], const ColMap &cm, int nl,
                          const void *d_const_, hipStream_t st, const size_t (*rows)[2]) {
  using T = typename ArithOf<M_>::T;
  using CPS = ColPlansT<M_>;
  const T *d_const = static_cast<const T *>(d_const_);
  const T *io_cols = reinterpret_cast<const T *>(cs->d_io_cols);
  const T *coeff[3] = {reinterpret_cast<const T *>(cs->d_coeff[0]), reinterpret_cast<const T *>(cs->d_coeff[1]),
                       reinterpret_cast<const T *>(cs->d_coeff[2])};
  const M_ *qmod = CtxArith<M_>::qmod(ctx);
  const size_t m = cs->m, M = P->M;
  const size_t C = (size_t)nl * cm.ns;  // columns in this chunk
  const CPS cp = make_colplans<M_>(ctx, P, cm.limb0);
  const bool needH = outs[6] != nullptr;
  const bool shortcut = witness_io_shortcut(cs);
  bool need_io[3], need_full[3], need_cst[3];
  for (int w = 0; w < 3; w++) {
    need_io[w] = outs[w] != nullptr || outs[3 + w] != nullptr;
    need_full[w] = outs[3 + w] != nullptr || (needH && w < 2);  // H needs A and B only
    // a constant part that differs per slot (polynomial coefficients on the constant one) is evaluated and
    // interpolated column by column like the other vectors; the shortcut never sees such a system
    need_cst[w] = cs->const_poly[w] && outs[3 + w] != nullptr;
  }
  // column-major workspace, only the vectors this call needs: io (fallback path only), full, per-slot constant parts, H
  auto needed = [&](int k) {
    return k < 3 ? (need_io[k] && !shortcut) : (k < 6 ? need_full[k - 3] : (k == 6 ? needH : need_cst[k - 7]));
  };
  static const int order[10] = {0, 1, 2, 3, 4, 5, 7, 8, 9, 6};  // the interpolated vectors adjacent, H last
  int slot_of[10], nvec = 0;
  for (int k = 0; k < 10; k++) slot_of[k] = -1;
  for (int o = 0; o < 10; o++)
    if (needed(order[o])) slot_of[order[o]] = nvec++;
  const size_t vec = C * M;
  T *colbuf = (T *)ws_get(ctx, 5, std::max<size_t>(1, (size_t)nvec * vec) * sizeof(T));
  auto colv = [&](int k) { return colbuf + (size_t)slot_of[k] * vec; };
  RS_REQUIRE((C % 2) == 0 && (M % 2) == 0, "column tiles move slot pairs and row pairs");
  const dim3 tgrid((unsigned)((C + 63) / 64), (unsigned)((M + 63) / 64));    // transposing kernels: 64 x 64 tiles
  const dim3 tgrid_ev((unsigned)((C + 63) / 64), (unsigned)((M + 31) / 32));  // the evaluation: 64 columns x 32 rows
  const T *ptab = reinterpret_cast<const T *>(cs->d_ptab);
  // the map of output vector k: its row range (rows == null: every row)
  auto cm_for = [&](int k) {
    ColMap c = cm;
    if (rows) {
      c.row0 = rows[k][0];
      c.row1 = rows[k][1];
    }
    return c;
  };
  for (int w = 0; w < 3; w++)
    for (int kind = 0; kind < 3; kind++) {  // io, full, constant part
      const int k = kind == 2 ? 7 + w : 3 * kind + w;
      if (!needed(k)) continue;
      // per (row, slot): 8 bytes of assignment per non-zero + 8 bytes of column written
      ProfScope prof(ctx, st, "r1cs_eval_cols_kernel", (double)C * 8.0 * ((double)cs->nnz[w] + (double)M), 7.0 * (double)C * (double)cs->nnz[w]);
      hipLaunchKernelGGL(r1cs_eval_cols_kernel<M_>, tgrid_ev, dim3(256), 0, st, cs->d_row_ptr[w], cs->d_col[w], coeff[w],
                         cs->nnz[w], d_asg, colv(k), m, C, M, kind == 2 ? (int)RS_EVAL_CONST : (kind ? (int)RS_EVAL_FULL : (int)RS_EVAL_IO),
                         (unsigned)cs->n_inputs, qmod, cm, cs->d_pidx[w], ptab);
    }
  RS_HIP(hipGetLastError());
  // one batched interpolation: the needed io / full / constant vectors are adjacent in the workspace
  {
    int n9 = 0;
    for (int k = 0; k < 10; k++) n9 += (k != 6) && needed(k);
    if (n9) launch_interp<M_>(ctx, P, cp, colbuf, (size_t)n9 * C, C, (size_t)cm.ns, cm.limb0, st);
  }
  if (needH) launch_h<M_>(ctx, P, cp, colv(3), colv(4), colv(6), C, (size_t)cm.ns, d1, d2, d3, cm, st, need_full[2] ? colv(5) : (const T *)nullptr);
  const unsigned eb = (unsigned)std::min<size_t>((vec + 255) / 256, 256 * 16);
  if (!shortcut) {
    // fallback: X_mid = interp(full) - interp(io) + interp(constant part), combined in column-major form
    for (int w = 0; w < 3; w++) {
      if (!outs[3 + w]) continue;
      // the constant part: slot constant (interpolated once per call), or per slot when polynomial coefficients
      // multiply the constant one (then the column holds EVERY index-0 term, scalar ones included)
      const T *cst = (!need_cst[w] && d_const && cs->has_const[w]) ? d_const + (size_t)w * ctx->L * M : nullptr;
      ProfScope prof(ctx, st, "mid_kernel", (double)vec * 24.0, 3.0 * (double)vec);
      hipLaunchKernelGGL(mid_kernel<CPS>, dim3(eb), dim3(256), 0, st, colv(3 + w), colv(w), cst, M, C, (unsigned)cm.ns, cp, cm.limb0,
                         need_cst[w] ? colv(7 + w) : (const T *)nullptr);
    }
    RS_HIP(hipGetLastError());
    for (int k = 0; k < 6; k++)
      if (outs[k]) {
        ProfScope prof(ctx, st, "transpose_out_kernel", (double)C * (double)m * 16.0, 0.0);
        hipLaunchKernelGGL(transpose_out_kernel<T>, tgrid, dim3(256), 0, st, colv(k), outs[k], m, C, M, cm_for(k));
      }
  } else {
    for (int w = 0; w < 3; w++) {
      if (!need_io[w]) continue;
      IoDesc io{cs->d_io_k[w], cs->d_io_c[w], cs->io_count[w]};
      if (outs[3 + w]) {  // io (if wanted) and mid in one pass over the interpolated columns
        const T *cst = cs->io_const_col[w] >= 0 ? io_cols + (size_t)cs->io_const_col[w] * ctx->L * M : nullptr;
        // 8 bytes of column read, 8 or 16 written, the primary inputs re-read per row (L2 resident)
        ProfScope prof(ctx, st, "io_mid_out_kernel", (double)C * (double)m * (outs[w] ? 24.0 : 16.0),
                       (double)C * (double)m * (7.0 * io.count + 4.0));
        hipLaunchKernelGGL(io_mid_out_kernel<M_>, tgrid, dim3(256), 0, st, colv(3 + w), io, io_cols, d_asg, cst, outs[w],
                           outs[3 + w], m, C, M, qmod, cm_for(3 + w));  // io and mid of one matrix share their row range (checked by the caller)
      } else {  // io alone: no column work at all
        const unsigned by = (unsigned)((C / 2 + 255) / 256);
        const ColMap cw = cm_for(w);
        const size_t r1 = std::min(m, cw.row1);
        if (r1 <= cw.row0) continue;
        ProfScope prof(ctx, st, "io_coeff_kernel", (double)C * (double)(r1 - cw.row0) * 8.0, (double)C * (double)(r1 - cw.row0) * 7.0 * io.count);
        hipLaunchKernelGGL(io_coeff_kernel<M_>, dim3((unsigned)(r1 - cw.row0), by), dim3(256), 0, st, io, io_cols, d_asg, outs[w], C, M, qmod, cw);
      }
    }
  }
  RS_HIP(hipGetLastError());
  if (needH) {
    {
      ProfScope prof(ctx, st, "transpose_out_kernel", (double)C * (double)m * 16.0, 0.0);
      hipLaunchKernelGGL(transpose_out_kernel<T>, tgrid, dim3(256), 0, st, colv(6), outs[6], std::min(m + 1, M), C, M, cm_for(6));
    }
    const ColMap ch = cm_for(6);
    if (m == M && m >= ch.row0 && m < ch.row1)  // row m does not exist in the M-row column tile
      hipLaunchKernelGGL(h_top_kernel<M_>, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, outs[6] + (m - ch.row0) * cm.out_stride(), d1, d2, C,
                         qmod, cm);
    RS_HIP(hipGetLastError());
  }
}

// Witness map driver.  outs[k] (k = A_io,B_io,C_io,A_mid,B_mid,C_mid,H) may be null.  Slots
// [slot0, slot0 + nslots) of every limb are processed; `compact` selects the output layout
// [t][L][nslots] (a slot-sharded rank) instead of the full [t][L][N].  The columns are
// worked through in chunks whose column-major workspace stays within g_witness_col_budget_mib
// (at m = 2^16 and the headline ring that is one limb at a time: 3 x 4 GiB instead of 7 x 16 GiB).
template <class M_>
static void witness_run_arith(rs_ctx *ctx, const rs_r1cs *cs, const uint64_t *d_asg, const uint64_t *d1, const uint64_t *d2,
                              const uint64_t *d3, uint64_t *const outs[7], uint64_t *h_Z, hipStream_t st, int slot0, int nslots,
                              bool compact, const size_t (*rows)[2]) {
  using T = typename ArithOf<M_>::T;
  RS_REQUIRE((d1 && d2 && d3) || (!d1 && !d2 && !d3), "d1,d2,d3 must be all set or all null");
  if (nslots < 0) nslots = ctx->N - slot0;
  RS_REQUIRE(slot0 >= 0 && nslots >= 2 && slot0 + nslots <= ctx->N && !(slot0 & 1) && !(nslots & 1),
             "slot range must be even-aligned and inside the ring");
  const size_t m = cs->m;
  if (rows) {
    for (int k = 0; k < 7; k++)
      RS_REQUIRE(!outs[k] || (rows[k][0] <= rows[k][1] && rows[k][1] <= m + (k == 6 ? 1 : 0)), "row range outside the vector");
    for (int w = 0; w < 3; w++)
      RS_REQUIRE(!outs[w] || !outs[3 + w] || (rows[w][0] == rows[3 + w][0] && rows[w][1] == rows[3 + w][1]),
                 "the io and mid vectors of one matrix take the same row range");
  }
  WitnessPlan *P = get_plan(ctx, m);
  const size_t M = P->M;
  const int L = ctx->L;
  if (h_Z)
    for (int i = 0; i < L; i++) memcpy(h_Z + (size_t)i * (m + 1), P->limb[i].Z.data(), sizeof(uint64_t) * (m + 1));
  const bool shortcut = witness_io_shortcut(cs);
  if (shortcut) build_io_cache<M_>(ctx, cs, P, make_colplans<M_>(ctx, P), st);
  // fallback path: interpolated constant parts [3][L][M], once per call
  T *d_const = nullptr;
  if (!shortcut && (cs->has_const[0] || cs->has_const[1] || cs->has_const[2]) && (outs[3] || outs[4] || outs[5])) {
    std::vector<T> hc((size_t)3 * L * M, T(0));
    for (int w = 0; w < 3; w++)
      for (int i = 0; i < L; i++)
        for (size_t r = 0; r < m; r++) hc[((size_t)w * L + i) * M + r] = HostArith<M_>::plain(cs->h_const[w][(size_t)i * m + r], ctx->q[i]);
    d_const = (T *)ws_get(ctx, 4, hc.size() * sizeof(T));
    RS_HIP(hipMemcpyAsync(d_const, hc.data(), hc.size() * sizeof(T), hipMemcpyHostToDevice, st));
    RS_HIP(hipStreamSynchronize(st));  // hc goes out of scope
    launch_interp<M_>(ctx, P, make_colplans<M_>(ctx, P), d_const, (size_t)3 * L, (size_t)L, 1, 0, st);
  }
  // chunking: as many whole limbs as fit the budget, else pieces of one limb (multiples of 64 slots)
  int nvec = 0;
  {
    const bool needH = outs[6] != nullptr;
    for (int w = 0; w < 3; w++) {
      const bool need_io = outs[w] || outs[3 + w], need_full = outs[3 + w] || (needH && w < 2);
      nvec += (need_io && !shortcut) + need_full + (cs->const_poly[w] && outs[3 + w]);
    }
    nvec += needH;
  }
  const size_t budget_cols =
      std::max<size_t>(64, ((size_t)g_witness_col_budget_mib << 20) / (std::max(1, nvec) * M * sizeof(double)));
  ColMap cm{0, nslots, slot0, ctx->N, L, compact ? nslots : ctx->N, compact ? slot0 : 0};
  if ((size_t)nslots <= budget_cols) {
    const int per = (int)std::max<size_t>(1, std::min<size_t>((size_t)L, budget_cols / (size_t)nslots));
    for (int l0 = 0; l0 < L; l0 += per) {
      cm.limb0 = l0;
      witness_chunk<M_>(ctx, cs, P, d_asg, d1, d2, d3, outs, cm, std::min(per, L - l0), d_const, st, rows);
    }
  } else {
    const int piece = (int)std::max<size_t>(64, (budget_cols / 64) * 64);
    for (int l0 = 0; l0 < L; l0++)
      for (int s0 = 0; s0 < nslots; s0 += piece) {
        cm.limb0 = l0;
        cm.slot0 = slot0 + s0;
        cm.ns = std::min(piece, nslots - s0);
        witness_chunk<M_>(ctx, cs, P, d_asg, d1, d2, d3, outs, cm, 1, d_const, st, rows);
      }
  }
}
void witness_run(rs_ctx *ctx, const rs_r1cs *cs, const uint64_t *d_asg, const uint64_t *d1, const uint64_t *d2,
                 const uint64_t *d3, uint64_t *const outs[7], uint64_t *h_Z, hipStream_t st, int slot0 = 0, int nslots = -1,
                 bool compact = false, const size_t (*rows)[2] = nullptr) {
  RS_DISPATCH_ARITH(ctx, (witness_run_arith<Mod>(ctx, cs, d_asg, d1, d2, d3, outs, h_Z, st, slot0, nslots, compact, rows)),
                    (witness_run_arith<ModI>(ctx, cs, d_asg, d1, d2, d3, outs, h_Z, st, slot0, nslots, compact, rows)));
}

template <class M_>
static void interpolate_arith(rs_ctx *ctx, const uint64_t *d_y, uint64_t *d_out, size_t n, hipStream_t st) {
  using T = typename ArithOf<M_>::T;
  WitnessPlan *P = get_plan(ctx, n);
  const ColPlansT<M_> cp = make_colplans<M_>(ctx, P);
  const size_t M = P->M, S_ = ctx->ring_words();
  T *colbuf = (T *)ws_get(ctx, 5, S_ * M * sizeof(T));
  const dim3 tgrid((unsigned)((S_ + 31) / 32), (unsigned)((M + 31) / 32));
  const ColMap cm{0, ctx->N, 0, ctx->N, ctx->L, ctx->N, 0};
  hipLaunchKernelGGL(transpose_in_kernel<T>, tgrid, dim3(256), 0, st, d_y, colbuf, n, S_, M);
  launch_interp<M_>(ctx, P, cp, colbuf, S_, S_, (size_t)ctx->N, 0, st);
  const dim3 ogrid((unsigned)((S_ + 63) / 64), (unsigned)((M + 63) / 64));
  hipLaunchKernelGGL(transpose_out_kernel<T>, ogrid, dim3(256), 0, st, colbuf, d_out, n, S_, M, cm);
  RS_HIP(hipGetLastError());
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
  return rs_r1cs_create_poly(ctx, m, n_vars, n_inputs, h_row_ptr, h_col, h_coeff, nnz, nullptr, nullptr, 0, out);
}

int rs_r1cs_create_poly(rs_ctx *ctx, size_t m, size_t n_vars, size_t n_inputs, const uint32_t *const h_row_ptr[3],
                        const uint32_t *const h_col[3], const uint64_t *const h_coeff[3], const size_t nnz[3],
                        const int32_t *const h_poly_idx[3], const uint64_t *h_poly_table, size_t n_poly, rs_r1cs **out) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && out && h_row_ptr && h_col && h_coeff && nnz, "null argument");
  RS_REQUIRE(m >= 1 && n_inputs <= n_vars, "bad R1CS shape");
  RS_REQUIRE(n_poly == 0 || (h_poly_idx && h_poly_table), "polynomial coefficient table without indices");
  struct Holder {  // frees a partly built object when a check below throws
    rs_r1cs *p;
    ~Holder() { rs_r1cs_destroy(p); }
  } holder{new rs_r1cs()};
  rs_r1cs *cs = holder.p;
  cs->m = m;
  cs->n_vars = n_vars;
  cs->n_inputs = n_inputs;
  cs->L = ctx->L;
  cs->n_poly = n_poly;
  const size_t rw = ctx->ring_words();
  if (n_poly) {  // the table as constants of the context's arithmetic, in the ring layout [n_poly][L][N]
    cs->h_ptab.assign(h_poly_table, h_poly_table + n_poly * rw);
    std::vector<uint64_t> pt(n_poly * rw);
    for (size_t k = 0; k < n_poly; k++)
      for (int i = 0; i < ctx->L; i++)
        for (int x = 0; x < ctx->N; x++) {
          const size_t at = (k * ctx->L + i) * (size_t)ctx->N + x;
          cs->h_ptab[at] %= ctx->q[i];
          pt[at] = konst_word(ctx, cs->h_ptab[at], ctx->q[i]);
        }
    RS_HIP(hipMalloc(&cs->d_ptab, sizeof(double) * pt.size()));
    RS_HIP(hipMemcpy(cs->d_ptab, pt.data(), sizeof(double) * pt.size(), hipMemcpyHostToDevice));
  }
  for (int w = 0; w < 3; w++) {
    const size_t z = nnz[w];
    cs->nnz[w] = z;
    RS_REQUIRE(h_row_ptr[w] && (h_col[w] || z == 0) && (h_coeff[w] || z == 0), "null matrix array");
    RS_REQUIRE(h_row_ptr[w][0] == 0 && h_row_ptr[w][m] == z, "row_ptr does not match nnz");
    for (size_t r = 0; r < m; r++) RS_REQUIRE(h_row_ptr[w][r] <= h_row_ptr[w][r + 1], "row_ptr is not monotone");
    cs->h_row_ptr[w].assign(h_row_ptr[w], h_row_ptr[w] + m + 1);
    cs->h_col[w].assign(h_col[w], h_col[w] + z);
    cs->h_coeff[w].assign(h_coeff[w], h_coeff[w] + (size_t)ctx->L * z);
    cs->h_const[w].assign((size_t)ctx->L * m, 0);
    cs->has_const[w] = false;
    const int32_t *pidx = (n_poly && h_poly_idx[w]) ? h_poly_idx[w] : nullptr;
    bool any_poly = false;
    for (size_t e = 0; pidx && e < z; e++) {
      RS_REQUIRE(pidx[e] < 0 || (size_t)pidx[e] < n_poly, "polynomial coefficient index out of range");
      if (pidx[e] < 0) continue;
      any_poly = true;
      if (h_col[w][e] <= n_inputs) cs->io_poly = true;
      if (h_col[w][e] == 0) cs->const_poly[w] = true;
    }
    std::vector<uint64_t> cf((size_t)ctx->L * std::max<size_t>(z, 1), 0);  // table constants of the context's arithmetic
    for (size_t r = 0; r < m; r++)
      for (uint32_t e = h_row_ptr[w][r]; e < h_row_ptr[w][r + 1]; e++) {
        RS_REQUIRE(h_col[w][e] <= n_vars, "column index out of range");
        const bool is_poly = pidx && pidx[e] >= 0;
        for (int i = 0; i < ctx->L; i++) {
          const uint64_t c = is_poly ? 0 : h_coeff[w][(size_t)i * z + e] % ctx->q[i];  // the scalar slot of a polynomial term is unused
          cs->h_coeff[w][(size_t)i * z + e] = c;
          cf[(size_t)i * z + e] = konst_word(ctx, c, ctx->q[i]);
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
    if (any_poly) {
      cs->h_pidx[w].assign(pidx, pidx + z);
      RS_HIP(hipMalloc(&cs->d_pidx[w], sizeof(int32_t) * z));
      RS_HIP(hipMemcpy(cs->d_pidx[w], pidx, sizeof(int32_t) * z, hipMemcpyHostToDevice));
    }
  }
  holder.p = nullptr;
  *out = cs;
  RS_API_END
}

void rs_r1cs_destroy(rs_r1cs *cs) {
  if (!cs) return;
  for (int w = 0; w < 3; w++) {
    if (cs->d_row_ptr[w]) (void)hipFree(cs->d_row_ptr[w]);
    if (cs->d_col[w]) (void)hipFree(cs->d_col[w]);
    if (cs->d_coeff[w]) (void)hipFree(cs->d_coeff[w]);
    if (cs->d_io_k[w]) (void)hipFree(cs->d_io_k[w]);
    if (cs->d_io_c[w]) (void)hipFree(cs->d_io_c[w]);
    if (cs->d_pidx[w]) (void)hipFree(cs->d_pidx[w]);
  }
  if (cs->d_io_cols) (void)hipFree(cs->d_io_cols);
  if (cs->d_ptab) (void)hipFree(cs->d_ptab);
  delete cs;
}

int rs_r1cs_evaluate(rs_ctx *ctx, const rs_r1cs *cs, int which, int mode, const uint64_t *d_assignment, uint64_t *d_out,
                     rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && cs && d_assignment && d_out, "null argument");
  RS_REQUIRE(which >= 0 && which < 3 && mode >= 0 && mode <= 2, "bad selector");
  r1cs_evaluate_run(ctx, cs, which, mode, d_assignment, d_out, S(stream));
  RS_API_END
}

int rs_witness_map(rs_ctx *ctx, const rs_r1cs *cs, const uint64_t *d_assignment, const uint64_t *d_d1,
                   const uint64_t *d_d2, const uint64_t *d_d3, uint64_t *d_A_io, uint64_t *d_B_io, uint64_t *d_C_io,
                   uint64_t *d_A_mid, uint64_t *d_B_mid, uint64_t *d_C_mid, uint64_t *d_H, uint64_t *h_Z,
                   rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && cs && d_assignment, "null argument");
  WsScope ws_scope(ctx, S(stream));
  uint64_t *outs[7] = {d_A_io, d_B_io, d_C_io, d_A_mid, d_B_mid, d_C_mid, d_H};
  witness_run(ctx, cs, d_assignment, d_d1, d_d2, d_d3, outs, h_Z, S(stream));
  RS_API_END
}

int rs_witness_map_slots(rs_ctx *ctx, const rs_r1cs *cs, const uint64_t *d_assignment, const uint64_t *d_d1,
                         const uint64_t *d_d2, const uint64_t *d_d3, int slot0, int nslots, uint64_t *d_A_io,
                         uint64_t *d_B_io, uint64_t *d_C_io, uint64_t *d_A_mid, uint64_t *d_B_mid, uint64_t *d_C_mid,
                         uint64_t *d_H, uint64_t *h_Z, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && cs && d_assignment, "null argument");
  WsScope ws_scope(ctx, S(stream));
  uint64_t *outs[7] = {d_A_io, d_B_io, d_C_io, d_A_mid, d_B_mid, d_C_mid, d_H};
  witness_run(ctx, cs, d_assignment, d_d1, d_d2, d_d3, outs, h_Z, S(stream), slot0, nslots, true);
  RS_API_END
}

int rs_witness_map_rows(rs_ctx *ctx, const rs_r1cs *cs, const uint64_t *d_assignment, const uint64_t *d_d1,
                        const uint64_t *d_d2, const uint64_t *d_d3, const size_t *h_rows, uint64_t *d_A_io, uint64_t *d_B_io,
                        uint64_t *d_C_io, uint64_t *d_A_mid, uint64_t *d_B_mid, uint64_t *d_C_mid, uint64_t *d_H, uint64_t *h_Z,
                        rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && cs && d_assignment && h_rows, "null argument");
  WsScope ws_scope(ctx, S(stream));
  uint64_t *outs[7] = {d_A_io, d_B_io, d_C_io, d_A_mid, d_B_mid, d_C_mid, d_H};
  size_t rows[7][2];
  for (int k = 0; k < 7; k++) rows[k][0] = h_rows[2 * k], rows[k][1] = h_rows[2 * k + 1];
  witness_run(ctx, cs, d_assignment, d_d1, d_d2, d_d3, outs, h_Z, S(stream), 0, -1, false, rows);
  RS_API_END
}

int rs_interpolate(rs_ctx *ctx, const uint64_t *d_y, uint64_t *d_out, size_t n, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_y && d_out && n >= 1, "null argument");
  WsScope ws_scope(ctx, S(stream));
  RS_DISPATCH_ARITH(ctx, (interpolate_arith<Mod>(ctx, d_y, d_out, n, S(stream))), (interpolate_arith<ModI>(ctx, d_y, d_out, n, S(stream))));
  RS_API_END
}

}  // extern "C"
