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
// All transforms run inside one workgroup's LDS tile (ntt_core.hpp); data are transposed once
// from the boundary layout [term][limb][slot] to column-major [limb][slot][M] and back.
// Requires q_i = 1 mod 4M (cyclic NTT of length 2M) and M <= 8192 in this round.
#include <algorithm>
#include <cstring>
#include <string>
#include <type_traits>

#include "ntt_core.hpp"
#include "ntt_wide.hpp"
#include "rs_internal.hpp"

namespace rs {

constexpr int SCHOOL_LEVELS = 4;  // tree levels with node size <= 8 use schoolbook products

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
  std::vector<LimbPlan> limb;
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
  if (lg > t.logmax) {
    // the prime has no root of unity of that order: block convolution over blocks of Bh = 2^(logmax-1)
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

int g_witness_force_bc = 0;  // tuning knob "witness_force_bc": pretend the ring primes have only this 2-adicity (tests)
int g_witness_bc2 = 1;       // tuning knob "witness_bc2": two-dimensional block convolutions where they apply (0: the pairwise form)

static WitnessPlan *build_plan(rs_ctx *ctx, size_t m) {
  using namespace hostw;
  RS_REQUIRE(m >= 1, "need at least one constraint");
  WitnessPlan *P = new WitnessPlan();
  P->m = m;
  P->logM = std::max(1, clog2(m));
  P->M = (size_t)1 << P->logM;
  const size_t M = P->M;
  const int logM = P->logM;
  if (logM > 20)
    throw Error(RS_ERR_UNSUPPORTED, "witness map beyond 2^20 constraints is not supported");
  P->limb.resize(ctx->L);
  int vmin = 64;
  for (int li = 0; li < ctx->L; li++) vmin = std::min(vmin, host::two_adicity(ctx->q[li]));
  if (g_witness_force_bc > 0) vmin = std::min(vmin, g_witness_force_bc);  // tests: the block path on well-endowed primes
  const bool blocked = vmin < logM + 1;
  P->bc2 = blocked && g_witness_bc2 && !ctx->use_int && vmin >= 14 && logM >= 15;
  P->bcLog = blocked ? (P->bc2 ? 14 : std::min(vmin, 13)) : 0;
  // every context prime is 1 mod 2*N_enc with N_enc >= 16, so the 2-adicity is at least 5
  RS_REQUIRE(!blocked || P->bcLog > SCHOOL_LEVELS, "ring prime with too little 2-adicity for the witness map");
  const int tabLog = blocked ? P->bcLog : logM + 1;  // longest transform the device tables serve
  const size_t Bc = blocked ? (size_t)1 << (P->bcLog - 1) : 0, nblk = blocked ? std::max<size_t>(1, M / Bc) : 0;
  for (int li = 0; li < ctx->L; li++) {
    LimbPlan &lp = P->limb[li];
    const uint64_t p = ctx->q[li];
    RS_REQUIRE(p > 2 * M, "ring prime too small for the evaluation domain");
    lp.p = p;
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
      if (blocked) {
        e.resize(M);
        lp.d_bc_e = up(block_spectra(e, nblk));
        if (P->bc2) {
          std::vector<uint64_t> t2(2 * nblk * 2 * Bc);
          across_blocks(e, nblk, t2.data());
          lp.d_b2_e = up(t2);
        }
      } else {
        ntt_fwd(e, logM + 1, T);
        const uint64_t s2 = invmod((uint64_t)(2 * M) % p, p);
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
      std::vector<uint64_t> bcd(blocked && logM > P->bcLog ? (size_t)(logM - P->bcLog) * M : 0, 0);
      std::vector<uint64_t> b2d(P->bc2 && logM > P->bcLog ? (size_t)(logM - P->bcLog) * 2 * M : 0, 0);
      for (int l = 1; l <= logM; l++) {
        const size_t n = (size_t)1 << l, h = n >> 1;
        for (size_t i = 0; i < (M >> l); i++) {
          const auto &dl = prod[l - 1][2 * i];  // h low coefficients, monic of degree h
          if (l <= SCHOOL_LEVELS) {
            for (size_t k = 0; k < h; k++) dlow[(size_t)l * (M / 2 + 1) + i * h + k] = bal(dl[k]);
          } else if (blocked && l > P->bcLog) {
            // node i of level l: the h / Bc blocks of D_left's low part (the monic x^h term is added by the sink)
            const std::vector<uint64_t> sp = block_spectra(dl, h / Bc);
            std::copy(sp.begin(), sp.end(), bcd.begin() + (size_t)(l - P->bcLog - 1) * M + i * n);
            if (P->bc2) across_blocks(dl, h / Bc, b2d.data() + (size_t)(l - P->bcLog - 1) * 2 * M + i * 2 * n);
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
        if (!blocked) {
          ntt_fwd(shat, logM + 1, T);
          const uint64_t s2 = invmod((uint64_t)(2 * M) % p, p), s4 = mulmod(s2, s2, p);
          for (auto &x : shat) x = mulmod(x, s4, p);
        }
      }
      if (blocked) {
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
  }
  return P;
}

static void free_plan(WitnessPlan *P) {
  for (auto &lp : P->limb) {
    void *ptrs[] = {lp.d_tw, lp.d_itw, lp.d_invfact, lp.d_ehat, lp.d_dhat, lp.d_dlow, lp.d_shat, lp.d_ztab, lp.d_bc_e, lp.d_bc_s, lp.d_bc_d,
                    lp.d_b2_e, lp.d_b2_s, lp.d_b2_d};
    for (void *q : ptrs)
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

// per-limb device pointers handed to the column kernels; M: the context's arithmetic (tables hold table
// constants of that arithmetic: balanced doubles, or Montgomery-form integers)
template <class M_>
struct ColPlanT {
  using M = M_;
  using T = typename ArithOf<M_>::T;
  M mod;
  const T *tw, *itw, *invfact, *ehat, *dhat, *dlow, *shat, *ztab;
  const T *bc_e, *bc_s, *bc_d;  // block-convolution path (LimbPlan)
  const T *b2_e, *b2_s, *b2_d;  // ... in its two-dimensional form
  T bc_inv2b;                   // 1 / (2B) as a table constant
  T b2_inv;                     // 1 / (2B * 2M/B) = 1 / (4M): both unscaled inverse transforms of a two-dimensional data x data product
  uint32_t fwd_mask2, inv_mask2;
  uint32_t fmask[24], imask[24];  // reduce masks for transforms of length 2^l (FP64 arithmetic)
};
template <class M_>
struct ColPlansT {
  using M = M_;
  using T = typename ArithOf<M_>::T;
  ColPlanT<M_> l[RS_MAX_L];
};
using ColPlan = ColPlanT<Mod>;
using ColPlans = ColPlansT<Mod>;
struct ColBlockFactory {
  double *s;
  __device__ __forceinline__ LdsBlockIO operator()(int off) const { return LdsBlockIO{s + pidx(off)}; }
};

// Which columns a launch works on, and where they live in the boundary layouts.  The witness map is
// column-parallel (one column = one NTT slot of one ring limb), so a call may process any sub-range
// of slots of any sub-range of limbs: column c of the chunk is slot `slot0 + c % ns` of limb
// `limb0 + c / ns`.  Inputs (assignment, d1..d3) are always in the full layout [..][L][N]; outputs are
// [t][L][out_N] with slot s stored at s - out_slot0 (out_N = N, out_slot0 = 0: the full layout;
// out_N = ns, out_slot0 = slot0: the compact layout of a slot-sharded rank).
// slot0, ns, out_slot0 are even (lanes move slot PAIRS with 16-byte accesses).
struct ColMap {
  int limb0, ns, slot0, N, L, out_N, out_slot0;
  __device__ __forceinline__ void locate(size_t c, int &limb, int &slot) const {
    limb = limb0 + (int)(c / (size_t)ns);
    slot = slot0 + (int)(c % (size_t)ns);
  }
  __device__ __forceinline__ size_t in_index(int limb, int slot) const { return (size_t)limb * N + slot; }
  __device__ __forceinline__ size_t out_index(int limb, int slot) const { return (size_t)limb * out_N + (slot - out_slot0); }
  __host__ __device__ __forceinline__ size_t in_stride() const { return (size_t)L * N; }
  __host__ __device__ __forceinline__ size_t out_stride() const { return (size_t)L * out_N; }
};

// [rows][S] u64 (term-major, S = L*N) -> [S][M] f64 (column-major), rows >= m zero-filled.
template <class T>
__global__ void __launch_bounds__(256) transpose_in_kernel(const uint64_t *__restrict__ src, T *__restrict__ dst,
                                                           size_t m, size_t S, size_t M) {
  __shared__ T tile[32][33];
  const size_t s0 = (size_t)blockIdx.x * 32, r0 = (size_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const size_t r = r0 + k, sl = s0 + tx;
    tile[k][tx] = (r < m && sl < S) ? from_res<T>(src[r * S + sl]) : T(0);
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const size_t sl = s0 + k, r = r0 + tx;
    if (sl < S && r < M) dst[sl * M + r] = tile[tx][k];
  }
}
// [C][M] f64 canonical columns -> [rows][L][out_N] u64 for rows < m_out
template <class T>
__global__ void __launch_bounds__(256) transpose_out_kernel(const T *__restrict__ src, uint64_t *__restrict__ dst,
                                                            size_t m_out, size_t C, size_t M, ColMap cm) {
  __shared__ T tile[32][33];
  const size_t s0 = (size_t)blockIdx.x * 32, r0 = (size_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const size_t c = s0 + k, r = r0 + tx;
    tile[k][tx] = (c < C && r < M) ? src[c * M + r] : T(0);
  }
  __syncthreads();
  const size_t c = s0 + tx;
  if (c >= C) return;
  int limb, slot;
  cm.locate(c, limb, slot);
  const size_t o = cm.out_index(limb, slot), So = cm.out_stride();
  for (int k = ty; k < 32; k += 8) {
    const size_t r = r0 + k;
    if (r < m_out) dst[r * So + o] = to_res(tile[tx][k]);
  }
}

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
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const double2 v = src[64 * i];
      const int pa = tw_addr(2048 * wave + 128 * i + 2 * lane);
      s[pa] = v.x;
      s[pa + 1] = v.y;
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
#pragma unroll
  for (int e = 0; e < 32; e++) b[e] = mulmod(reduce(b[e], mod), s[33 * t + e], mod);
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
    reg_fwd_stages_zu<c>(X[k], mod, P.fmask[LV], [&](int st, int blk) { return tw[(1 << st) + blk]; });
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
    reg_inv_stages<c, true>(X[k], mod, P.imask[LV] >> 5, [&](int st, int i) { return itw[(W >> (st + 1)) + i]; });
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
    reg_fwd_stages_zu<c1>(X[k], mod, fmask, [&](int st, int blk) { return tw[(1 << st) + blk]; });
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
    reg_inv_stages<c1, true>(X[k], mod, imask >> 10, [&](int st, int i) { return itw[(R1 >> (st + 1)) + i]; });
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
template <int LOGT>
__global__ void __launch_bounds__(1 << (LOGT - 5), 2)
tree_wide_kernel(double *__restrict__ cols, int logM, size_t col0, unsigned S, unsigned slots_per_limb, ColPlans plans) {
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
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const double2 v = src[64 * i];
        const int pa = tw_addr(2048 * wave + 128 * i + 2 * lane);
        s[pa] = v.x;
        s[pa + 1] = v.y;
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
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int pa = tw_addr(2048 * wave + 128 * i + 2 * lane);
      dst[64 * i] = make_double2(canon(s[pa], mod), canon(s[pa + 1], mod));
    }
  }
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

// Input/primary coefficient vectors without interpolation (io shortcut): interpolation is linear
// and the io evaluations depend on the n_inputs primary variables only, so
//     X_io[t] = Lconst[t] + sum_{k <= n_inputs} x_k (*) L_k[t],   L_k = interp(column k of X)
// with slot-constant L_k computed once per circuit.  grid (m, slot pairs / 256).
struct IoDesc {
  const int *k;       // variable index (0 = constant one)
  const int *column;  // column index into Lcols
  int count;
};
template <class M>
__global__ void __launch_bounds__(256)
io_coeff_kernel(IoDesc io, const typename ArithOf<M>::T *__restrict__ Lcols /* [ncols][Ltot][M] */, const uint64_t *__restrict__ asg,
                uint64_t *__restrict__ out, size_t C, size_t Mlen, const M *__restrict__ qmod, ColMap cm) {
  using T = typename ArithOf<M>::T;
  const size_t t = blockIdx.x;
  const size_t c = 2 * ((size_t)blockIdx.y * blockDim.x + threadIdx.x);
  if (c >= C) return;
  int limb, slot;
  cm.locate(c, limb, slot);
  const size_t pair = cm.in_index(limb, slot) >> 1, Si = cm.in_stride();
  const M mod = qmod[limb];
  T a0 = T(0), a1 = T(0);
  for (int k = 0; k < io.count; k++) {
    const T lv = center(Lcols[((size_t)io.column[k] * cm.L + limb) * Mlen + t], mod);
    const int v_ = io.k[k];
    if (v_ == 0) {
      a0 = addm(a0, lv, mod);
      a1 = addm(a1, lv, mod);
    } else {
      const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(asg + (size_t)(v_ - 1) * Si)[pair];
      a0 = addm(a0, mulmod_dd(from_res<T>(v.x), lv, mod), mod);
      a1 = addm(a1, mulmod_dd(from_res<T>(v.y), lv, mod), mod);
    }
    if ((k & 3) == 3) {
      a0 = reduce(a0, mod);
      a1 = reduce(a1, mod);
    }
  }
  ulonglong2 o;
  o.x = to_res(canon(a0, mod));
  o.y = to_res(canon(a1, mod));
  reinterpret_cast<ulonglong2 *>(out + t * cm.out_stride())[cm.out_index(limb, slot) >> 1] = o;
}

// Column-major interpolated `full` vector -> term-major io AND mid vectors in one pass:
//   io[t]  = Lconst[t] + sum_k x_k (*) L_k[t]          (io shortcut, as io_coeff_kernel)
//   mid[t] = full[t] - io[t] + const[limb][t]
// i.e. transpose + io + mid fused: the column tile is transposed through LDS, the io value is
// computed where it is needed, and both results are written once (16 bytes per lane).
// grid (C/64, M/32).
template <class M>
__global__ void __launch_bounds__(256)
io_mid_out_kernel(const typename ArithOf<M>::T *__restrict__ cols, IoDesc io,
                  const typename ArithOf<M>::T *__restrict__ Lcols /* [ncols][Ltot][M] */, const uint64_t *__restrict__ asg,
                  const typename ArithOf<M>::T *__restrict__ cst /* [Ltot][M] or null */, uint64_t *__restrict__ io_out /* or null */,
                  uint64_t *__restrict__ mid_out, size_t m, size_t C, size_t Mlen, const M *__restrict__ qmod, ColMap cm) {
  using T = typename ArithOf<M>::T;
  __shared__ T tile[64][33];  // [column][row]
  const size_t s0 = (size_t)blockIdx.x * 64, r0 = (size_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 64; k += 8) {
    const size_t c = s0 + k, r = r0 + tx;
    tile[k][tx] = (c < C && r < Mlen) ? cols[c * Mlen + r] : T(0);
  }
  __syncthreads();
  const size_t c = s0 + 2 * tx;  // this lane's column pair (ns is even: both slots in one limb)
  if (c >= C) return;
  int limb, slot;
  cm.locate(c, limb, slot);
  const size_t pair = cm.in_index(limb, slot) >> 1, Si = cm.in_stride();
  const size_t opair = cm.out_index(limb, slot) >> 1, So = cm.out_stride();
  const M mod = qmod[limb];
  for (int k = ty; k < 32; k += 8) {
    const size_t r = r0 + k;
    if (r >= m) continue;
    T a0 = T(0), a1 = T(0);
    for (int e = 0; e < io.count; e++) {
      const T lv = center(Lcols[((size_t)io.column[e] * cm.L + limb) * Mlen + r], mod);
      const int kk = io.k[e];
      if (kk == 0) {
        a0 = addm(a0, lv, mod);
        a1 = addm(a1, lv, mod);
      } else {
        const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(asg + (size_t)(kk - 1) * Si)[pair];
        a0 = addm(a0, mulmod_dd(from_res<T>(v.x), lv, mod), mod);
        a1 = addm(a1, mulmod_dd(from_res<T>(v.y), lv, mod), mod);
      }
      if ((e & 3) == 3) {
        a0 = reduce(a0, mod);
        a1 = reduce(a1, mod);
      }
    }
    a0 = canon(a0, mod);
    a1 = canon(a1, mod);
    if (io_out) {
      ulonglong2 o;
      o.x = to_res(a0);
      o.y = to_res(a1);
      reinterpret_cast<ulonglong2 *>(io_out + r * So)[opair] = o;
    }
    const T cc = cst ? cst[(size_t)limb * Mlen + r] : T(0);
    ulonglong2 o;
    o.x = to_res(canon(addm(subm(tile[2 * tx][k], a0, mod), cc, mod), mod));
    o.y = to_res(canon(addm(subm(tile[2 * tx + 1][k], a1, mod), cc, mod), mod));
    reinterpret_cast<ulonglong2 *>(mid_out + r * So)[opair] = o;
  }
}

// coefficients_for_X_mid = interp(full) - interp(io) + interp(constant part), in place over `full`.
// (The reference evaluates index-0 terms in BOTH the io and the mid pass, r1cs_to_qrp.tcc:175-201.)
template <class CPS>
__global__ void __launch_bounds__(256)
mid_kernel(typename CPS::T *__restrict__ full, const typename CPS::T *__restrict__ io,
           const typename CPS::T *__restrict__ cst /* [Ltot][M] or null */, size_t M, size_t S, unsigned slots_per_limb, CPS plans,
           int limb0, const typename CPS::T *__restrict__ cst_cols /* [S][M] or null: a constant part that differs per slot */) {
  using T = typename CPS::T;
  const size_t total = S * M, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t col = i / M, k = i % M;
    const int limb = (int)(col / slots_per_limb);  // chunk-local: plans are shifted by limb0
    const typename CPS::M mod = plans.l[limb].mod;
    T v = subm(full[i], io[i], mod);
    if (cst) v = addm(v, cst[(size_t)(limb0 + limb) * M + k], mod);
    if (cst_cols) v = addm(v, cst_cols[i], mod);
    full[i] = canon(v, mod);
  }
}

// one row of linear_combination::evaluate for a slot pair: sum_e coeff_e * x_{col_e} (index 0 = the constant one).
// coeff_e is a slot-constant scalar, or -- pidx[e] >= 0 -- a general ring element: row pidx[e] of the table, whose two
// residues for this slot pair sit at ptab_pair + pidx[e] * Si (the table has the assignment's [L][N] layout).
#define RS_EVAL_CONST 3 /* internal mode: the index-0 terms only (the constant part of a mid vector) */
template <class M>
__device__ __forceinline__ void eval_row_pair(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col,
                                              const typename ArithOf<M>::T *__restrict__ coeff_limb, size_t row,
                                              const uint64_t *__restrict__ asg, size_t Si, size_t pair, int mode, unsigned n_inputs,
                                              const M mod, typename ArithOf<M>::T &o0, typename ArithOf<M>::T &o1,
                                              const int32_t *__restrict__ pidx, const typename ArithOf<M>::T *__restrict__ ptab) {
  using T = typename ArithOf<M>::T;
  T a0 = T(0), a1 = T(0);
  int since = 0;
  for (uint32_t e = row_ptr[row]; e < row_ptr[row + 1]; e++) {
    const uint32_t cv = col[e];
    T cf0 = coeff_limb[e], cf1 = cf0;  // table constants
    if (pidx) {
      const int32_t pk = pidx[e];
      if (pk >= 0) {
        const T *pc = ptab + (size_t)pk * Si + 2 * pair;
        cf0 = pc[0];
        cf1 = pc[1];
      }
    }
    if (cv == 0) {
      a0 = addm(a0, konst_value(cf0, mod), mod);
      a1 = addm(a1, konst_value(cf1, mod), mod);
    } else {
      const bool is_input = (cv - 1) < n_inputs;
      if ((mode == RS_EVAL_IO && !is_input) || (mode == RS_EVAL_MID && is_input) || mode == RS_EVAL_CONST) continue;
      const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(asg + (size_t)(cv - 1) * Si)[pair];
      a0 = addm(a0, mulmod(from_res<T>(v.x), cf0, mod), mod);
      a1 = addm(a1, mulmod(from_res<T>(v.y), cf1, mod), mod);
    }
    if (++since == 4) {
      since = 0;
      a0 = reduce(a0, mod);
      a1 = reduce(a1, mod);
    }
  }
  o0 = canon(a0, mod);
  o1 = canon(a1, mod);
}

// a14: linear_combination::evaluate for every constraint (relations/variable.tcc:246-254).
// grid (m, ceil(L*N/512)); each thread handles two adjacent slots.
template <class M>
__global__ void __launch_bounds__(256)
r1cs_eval_kernel(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col,
                 const typename ArithOf<M>::T *__restrict__ coeff, size_t nnz, const uint64_t *__restrict__ asg,
                 uint64_t *__restrict__ out, int N, int L, int mode, unsigned n_inputs, const M *__restrict__ qmod,
                 const int32_t *__restrict__ pidx, const typename ArithOf<M>::T *__restrict__ ptab) {
  using T = typename ArithOf<M>::T;
  const size_t row = blockIdx.x;
  const size_t S = (size_t)L * N;
  const size_t pair = (size_t)blockIdx.y * blockDim.x + threadIdx.x;
  if (2 * pair >= S) return;
  const int limb = (int)((2 * pair) / (size_t)N);
  T a0, a1;
  eval_row_pair<M>(row_ptr, col, coeff + (size_t)limb * nnz, row, asg, S, pair, mode, n_inputs, qmod[limb], a0, a1, pidx, ptab);
  ulonglong2 o;
  o.x = to_res(a0);
  o.y = to_res(a1);
  reinterpret_cast<ulonglong2 *>(out + row * S)[pair] = o;
}

// linear_combination::evaluate straight into the column-major layout of the witness map
// (r1cs_eval_kernel + transpose fused; rows >= m are the zero padding of the columns).
// grid (C/64, M/32): 64 columns x 32 rows per workgroup.
template <class M>
__global__ void __launch_bounds__(256)
r1cs_eval_cols_kernel(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col,
                      const typename ArithOf<M>::T *__restrict__ coeff, size_t nnz, const uint64_t *__restrict__ asg,
                      typename ArithOf<M>::T *__restrict__ cols, size_t m, size_t C, size_t Mlen, int mode, unsigned n_inputs,
                      const M *__restrict__ qmod, ColMap cm, const int32_t *__restrict__ pidx,
                      const typename ArithOf<M>::T *__restrict__ ptab) {
  using T = typename ArithOf<M>::T;
  __shared__ T tile[64][33];  // [column][row]
  const size_t s0 = (size_t)blockIdx.x * 64, r0 = (size_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const size_t c = s0 + 2 * tx;  // column pair of this lane, 16-byte loads of the assignment
  if (c < C) {
    int limb, slot;
    cm.locate(c, limb, slot);
    const size_t pair = cm.in_index(limb, slot) >> 1, Si = cm.in_stride();
    const M mod = qmod[limb];
    for (int k = ty; k < 32; k += 8) {
      const size_t row = r0 + k;
      T a0 = T(0), a1 = T(0);
      if (row < m) eval_row_pair<M>(row_ptr, col, coeff + (size_t)limb * nnz, row, asg, Si, pair, mode, n_inputs, mod, a0, a1, pidx, ptab);
      tile[2 * tx][k] = a0;
      tile[2 * tx + 1][k] = a1;
    }
  }
  __syncthreads();
  for (int k = ty; k < 64; k += 8) {
    const size_t cc = s0 + k, r = r0 + tx;
    if (cc < C && r < Mlen) cols[cc * Mlen + r] = tile[k][tx];
  }
}

// H[m] when m == M (the column tile holds M rows only): d1*d2*Z[m] = d1*d2 (Z monic), zero without ZK
template <class M>
__global__ void __launch_bounds__(256)
h_top_kernel(uint64_t *__restrict__ top, const uint64_t *__restrict__ d1, const uint64_t *__restrict__ d2, size_t C,
             const M *__restrict__ qmod, ColMap cm) {
  using T = typename ArithOf<M>::T;
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  int limb, slot;
  cm.locate(c, limb, slot);
  const M mod = qmod[limb];
  uint64_t v = 0;
  if (d1) {
    const size_t di = cm.in_index(limb, slot);
    v = to_res(canon(mulmod_dd(center(from_res<T>(d1[di]), mod), center(from_res<T>(d2[di]), mod), mod), mod));
  }
  top[cm.out_index(limb, slot)] = v;
}

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
};
#ifndef RS_WORKSPACE_NT
#define RS_WORKSPACE_NT 1
#endif
template <class T>
struct GlobalIOT {
  T *p;
  __device__ __forceinline__ int pbase(int) const { return 0; }
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
enum CrossSrc { CS_PLAIN = 0, CS_SCALE_PAD, CS_FILL_RIGHT, CS_PAD_CENTER, CS_REV_TRUNC };
enum CrossDst { CD_PLAIN = 0, CD_TAKE_LOW, CD_COMBINE, CD_COMBINE_CANON, CD_H_FINISH, CD_H_FINISH_CANON };
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
    return p[k];
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
    } else {
      p[k] = v;
    }
  }
};

// cross stages [s0, s0+R) of batched length-2^logsub transforms inside columns of length 2^logtot.
// grid (x, columns).  MODE: CrossSrc for forward passes, CrossDst for inverse passes.
template <bool INV, int R, int MODE, class CPS>
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
      inv_round<R>(io, io, a.logtot, a.logsub, a.s0, P.itw, 1, P.mod, P.imask[a.logsub], ln);
    } else {
      const CrossOut<MODE, Mt> out{static_cast<T *>(a.dst) + (col << a.logM), P.invfact, P.mod, M, a.m, n, n >> 1};
      inv_round<R>(io, out, a.logtot, a.logsub, a.s0, P.itw, 1, P.mod, P.imask[a.logsub], ln);
    }
  } else {
    if (MODE == CS_PLAIN) {
      fwd_round<R>(io, io, a.logtot, a.logsub, a.s0, P.tw, 1, P.mod, P.fmask[a.logsub], ln);
    } else {
      const size_t stride = MODE == CS_REV_TRUNC ? (size_t)2 << a.logM : (size_t)1 << a.logM;
      const CrossIn<MODE, Mt> in{static_cast<const T *>(a.src) + col * stride, P.invfact, P.mod, M, a.m, n, n >> 1};
      fwd_round<R>(in, io, a.logtot, a.logsub, a.s0, P.tw, 1, P.mod, P.fmask[a.logsub], ln);
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
template <int MODE>
__global__ void __launch_bounds__(256, 2)
sub_ntt_wide_kernel(double *__restrict__ X, int log_n1, TabPtrs tabs, unsigned tab_period, unsigned blocks_per_col, size_t col0,
                    unsigned S_, unsigned slots_per_limb, ColPlans plans, unsigned long long nblocks,
                    const double *__restrict__ Xsrc /* null: in place.  Else block b reads block b >> 1 of Xsrc: the two
                    sub-transforms (roots 2 and 3) of ONE zero-padded block of 2^13 coefficients (two-dimensional block convolutions) */) {
  using S = WideShape<13>;
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
    const unsigned long long bn = blk + gridDim.x;
    if (bn < nblocks) issue_loads(bn);
    mem_fence();
    // ---- forward round 1: stages 0..3 on elements 2t+c + 512e, twiddles tw[2^k root + blk] (uniform)
#pragma unroll
    for (int c = 0; c < 2; c++)
      reg_fwd_stages<4, true>(v[c], mod, fmask, [&](int k, int b) { return tw[(root << k) + b]; });
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
      double w[31];
      SubTw::run<1>(tw + (root << 4) + hi, w);
      SubTw::run<2>(tw + (root << 5) + (hi << 1), w + 1);
      SubTw::run<4>(tw + (root << 6) + (hi << 2), w + 3);
      SubTw::run<8>(tw + (root << 7) + (hi << 3), w + 7);
      SubTw::run<16>(tw + (root << 8) + (hi << 4), w + 15);
      double x[32];
#pragma unroll
      for (int e = 0; e < 32; e++) x[e] = s[pb + 17 * e];
      reg_fwd_stages<5, true>(x, mod, fmask >> 4, [&](int k, int b) { return w[(1 << k) - 1 + b]; });
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
      SubTw::run<1>(tw + (root << 9) + g, w);
      SubTw::run<2>(tw + (root << 10) + (g << 1), w + 1);
      SubTw::run<4>(tw + (root << 11) + (g << 2), w + 3);
      SubTw::run<8>(tw + (root << 12) + (g << 3), w + 7);
      double x[16];
#pragma unroll
      for (int e = 0; e < 16; e++) x[e] = s[pb + e];
      reg_fwd_stages<4, true>(x, mod, fmask >> 9, [&](int k, int b) { return w[(1 << k) - 1 + b]; });
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 16; e++) s[pb + e] = x[e];
        continue;
      }
      {
        // The table entries of the wave's 64 groups are 1024 consecutive words: fetched with coalesced 16-byte loads
        // and handed to their owners through the wave's range of the tile, which is free once x has been read (a
        // thread fetching its own 128-byte run touches 64 different lines per instruction).
        const double *tab = (MODE == 2) ? static_cast<const double *>(tabs.t[limb]) + (size_t)(blk % tab_period) * S::N
                                        : static_cast<const double *>(tabs.t[0]) + blk * (size_t)S::N;
        const int wave = t >> 6, lane = t & 63;
        const int r0 = (j * 256 + wave * 64) * 16;
        const int p0 = S::px(r0 + 2 * lane);
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
        if (MODE == 2) {
#pragma unroll
          for (int e = 0; e < 16; e++) x[e] = mulmod(reduce(x[e], mod), s[pb + e], mod);
        } else {
#pragma unroll
          for (int e = 0; e < 16; e++) x[e] = mulmod(reduce(x[e], mod), reduce(s[pb + e], mod), mod);
        }
      }
      // inverse stage k of the block: twiddle itw[(n >> (k+1)) root + (position >> (k+1))]
      SubTw::run<8>(itw + ((size_t)root << 12) + (g << 3), w);
      SubTw::run<4>(itw + ((size_t)root << 11) + (g << 2), w + 8);
      SubTw::run<2>(itw + ((size_t)root << 10) + (g << 1), w + 12);
      SubTw::run<1>(itw + ((size_t)root << 9) + g, w + 14);
      reg_inv_stages<4, true>(x, mod, imask, [&](int k, int i) { return w[16 - (16 >> k) + i]; });
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
      reg_inv_stages<4, true>(v[c], mod, imask >> 9, [&](int k, int i) { return itw[((8 >> k) * root) + i]; });
    {
      double2 *dst = reinterpret_cast<double2 *>(X + blk * (size_t)S::N) + t;
#pragma unroll
      for (int e = 0; e < 16; e++) dst[(S::S / 2) * e] = make_double2(v[0][e], v[1][e]);
    }
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
    c.bc_inv2b = P->bcLog ? HostArith<M>::konst(host::invmod(((uint64_t)1 << P->bcLog) % lp.p, lp.p), lp.p) : T(0);
    c.b2_inv = P->bc2 ? HostArith<M>::konst(host::invmod((uint64_t)(4 * P->M) % lp.p, lp.p), lp.p) : T(0);
    c.fwd_mask2 = lp.fwd_mask2;
    c.inv_mask2 = lp.inv_mask2;
    for (int l = 0; l < 24; l++) {
      c.fmask[l] = fwd_reduce_mask(lp.p, l);
      c.imask[l] = inv_reduce_mask(lp.p, l);
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
static double tree_fp64(double T, int logT) {
  double f = T / 16.0 * (120.0 * 7.0 + 4.0 * 16.0 * 3.0);
  for (int l = SCHOOL_LEVELS + 1; l <= logT; l++) f += 2.0 * ntt_fp64(T, l) + 10.0 * T;
  return f;
}
static void launch_tree_tiles(rs_ctx *ctx, double *cols, size_t ncols, size_t col0, int logM, int logT, size_t S,
                              size_t slots_per_limb, const ColPlans &cp, hipStream_t st, bool newton = false) {
  const size_t T = (size_t)1 << logT;
  const double tiles = (double)(ncols << (logM - logT));
  // Newton conversion (single-tile columns): two passes of forward + inverse M-point transforms and two pointwise products
  const double newton_fp64 = newton ? 4.0 * ntt_fp64((double)T, logT) + 31.0 * (double)T : 0.0;
  // names as rocprofv3 prints them (a prefix of "rs::<name>") so that profiles/ and the live record can be joined
  const bool ct13 = logT == 13 && g_witness_tree_ct;
  const bool wide = !newton && g_witness_tree_ct == 2 && (logT == 13 || logT == 14);
  const char *pname = wide ? (logT == 14 ? "tree_wide_kernel<14>" : "tree_wide_kernel<13>")
                      : ct13 ? (newton ? "tree_columns_kernel<512, 13, true>" : "tree_columns_kernel<512, 13, false>")
                           : (newton ? "tree_columns_kernel<NEWTON>" : "tree_columns_kernel");
  ProfScope prof(ctx, st, pname, tiles * (double)T * 16.0, tiles * (tree_fp64((double)T, logT) + newton_fp64));
  const size_t lds1 = padded_len(T) * sizeof(double);
  const unsigned grid = (unsigned)(ncols << (logM - logT));
  const int thr = (int)std::max<size_t>(64, std::min<size_t>(1024, T / 16));  // 1024 only for a 2^14 tile (one workgroup per CU)
  RS_REQUIRE(wide || (T / thr <= 16 && logT >= 6), "tree tile out of range");
  RS_REQUIRE(!newton || logT == logM, "fused Newton conversion needs single-tile columns");
#define RS_TREE_LAUNCH_K(KERN)                                                                                   \
  do {                                                                                                           \
    RS_HIP(hipFuncSetAttribute((const void *)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));     \
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
    if (logT == 14) {
      RS_HIP(hipFuncSetAttribute((const void *)tree_wide_kernel<14>, hipFuncAttributeMaxDynamicSharedMemorySize, wl));
      hipLaunchKernelGGL(tree_wide_kernel<14>, dim3(grid), dim3(512), wl, st, cols, logM, col0, (unsigned)S, (unsigned)slots_per_limb, cp);
    } else {
      RS_HIP(hipFuncSetAttribute((const void *)tree_wide_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, wl));
      hipLaunchKernelGGL(tree_wide_kernel<13>, dim3(grid), dim3(256), wl, st, cols, logM, col0, (unsigned)S, (unsigned)slots_per_limb, cp);
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
  RS_HIP(hipFuncSetAttribute((const void *)tree_tiles_generic_kernel<ColPlansT<M>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(tree_tiles_generic_kernel<ColPlansT<M>>, dim3((unsigned)(ncols << (logM - logT))), dim3(col_threads(2 * T)), lds, st,
                     cols, logM, logT, col0, (unsigned)S, (unsigned)slots_per_limb, cp);
  RS_HIP(hipGetLastError());
}

template <bool INV, int MODE, class M>
static void launch_cross_pass(int R, const dim3 &grid, const CrossArgs &a, const ColPlansT<M> &cp, hipStream_t st) {
  using CPS = ColPlansT<M>;
  switch (R) {
    case 4: hipLaunchKernelGGL((cross_kernel<INV, 4, MODE, CPS>), grid, dim3(256), 0, st, a, cp); break;
    case 3: hipLaunchKernelGGL((cross_kernel<INV, 3, MODE, CPS>), grid, dim3(256), 0, st, a, cp); break;
    case 2: hipLaunchKernelGGL((cross_kernel<INV, 2, MODE, CPS>), grid, dim3(256), 0, st, a, cp); break;
    default: hipLaunchKernelGGL((cross_kernel<INV, 1, MODE, CPS>), grid, dim3(256), 0, st, a, cp); break;
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
    const int R = pick_radix(ncross - done, 4);
    a.s0 = INV ? logB + done : done;
    const unsigned gx = (unsigned)std::max<size_t>(1, std::min<size_t>((groups >> R) / 256, 1024));
    const dim3 grid(gx, (unsigned)ncols);
    const bool special = INV ? (done + R >= ncross) : (done == 0);
    ProfScope prof(ctx, st, cross_name(INV, R, special ? MODE : 0), (double)ncols * 8.0 * cross_words<INV, MODE>(a, special),
                   (double)ncols * ntt_fp64((double)groups, R));
    if (special)
      launch_cross_pass<INV, MODE, M>(R, grid, a, cp, st);
    else
      launch_cross_pass<INV, 0, M>(R, grid, a, cp, st);
    done += R;
  }
  RS_HIP(hipGetLastError());
}

template <int MODE, class M>
static void launch_sub(rs_ctx *ctx, typename ArithOf<M>::T *X, size_t ncols, size_t col0, int logtot, int logsub, int logB,
                       const TabPtrs *tabs, size_t tab_period, size_t S, size_t spl, const ColPlansT<M> &cp, hipStream_t st) {
  constexpr bool FP = std::is_same<M, Mod>::value;
  const size_t lds = padded_len((size_t)1 << logB) * sizeof(double);
  const size_t bpc = (size_t)1 << (logtot - logB);
  static const char *const names[4] = {"sub_ntt_kernel<0", "sub_ntt_kernel<1", "sub_ntt_kernel<2", "sub_ntt_kernel<3"};
  static const char *const names_ct[4] = {"sub_ntt_ct_kernel<0, 13>", "sub_ntt_ct_kernel<1, 13>", "sub_ntt_ct_kernel<2, 13>", "sub_ntt_ct_kernel<3, 13>"};
  static const char *const names_wide[4] = {"sub_ntt_wide_kernel<0>", "sub_ntt_wide_kernel<1>", "sub_ntt_wide_kernel<2>", "sub_ntt_wide_kernel<3>"};
#ifdef RS_EXPERIMENTS
  const bool ct = FP && logB == 13 && MODE != 1 && g_witness_sub_ct;
#else
  const bool ct = FP && logB == 13 && MODE != 1 && g_witness_sub_ct == 2;  // 0: the generic kernel; 1 and 3 exist in the experiments build only
#endif
  const double Bn = (double)((size_t)1 << logB), blocks = (double)(ncols * bpc);
  static const char *const names_w16[4] = {"sub_ntt_wide16_kernel<0>", "sub_ntt_wide16_kernel<1>", "sub_ntt_wide16_kernel<2>", "sub_ntt_wide16_kernel<3>"};
  ProfScope prof(ctx, st, ct ? (g_witness_sub_ct == 3 ? names_w16[MODE] : g_witness_sub_ct == 2 ? names_wide[MODE] : names_ct[MODE]) : names[MODE], blocks * Bn * (MODE == 3 ? 24.0 : 16.0),
                 blocks * ((MODE >= 2 ? 2.0 : 1.0) * ntt_fp64(Bn, logB) + (MODE >= 2 ? 7.0 * Bn : 0.0)));
  static TabPtrs none{};
  const TabPtrs &tp = tabs ? *tabs : none;
  if constexpr (FP) {
#ifdef RS_EXPERIMENTS
    if (logB == 13 && MODE != 1 && g_witness_sub_ct == 3) {
      const int wl = (int)(WideShape<13>::TILE * sizeof(double));
      const unsigned long long nb = (unsigned long long)(ncols * bpc);
      RS_HIP(hipFuncSetAttribute((const void *)sub_ntt_wide16_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, wl));
      hipLaunchKernelGGL((sub_ntt_wide16_kernel<MODE>), dim3((unsigned)std::min<unsigned long long>(nb, 512)), dim3(512), wl, st, X,
                         logsub - logB, tp, (unsigned)std::max<size_t>(1, tab_period), (unsigned)bpc, col0, (unsigned)S, (unsigned)spl, cp, nb);
      RS_HIP(hipGetLastError());
      return;
    }
#endif
    if (logB == 13 && MODE != 1 && g_witness_sub_ct == 2) {
      const int wl = (int)(WideShape<13>::TILE * sizeof(double));
      const unsigned long long nb = (unsigned long long)(ncols * bpc);
      RS_HIP(hipFuncSetAttribute((const void *)sub_ntt_wide_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, wl));
      hipLaunchKernelGGL((sub_ntt_wide_kernel<MODE>), dim3((unsigned)std::min<unsigned long long>(nb, 512)), dim3(256), wl, st, X,
                         logsub - logB, tp, (unsigned)std::max<size_t>(1, tab_period), (unsigned)bpc, col0, (unsigned)S, (unsigned)spl, cp, nb,
                         (const double *)nullptr);
      RS_HIP(hipGetLastError());
      return;
    }
#ifdef RS_EXPERIMENTS
    if (logB == 13 && MODE != 1 && g_witness_sub_ct) {
      RS_HIP(hipFuncSetAttribute((const void *)sub_ntt_ct_kernel<MODE, 13>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((sub_ntt_ct_kernel<MODE, 13>), dim3((unsigned)(ncols * bpc)), dim3(512), lds, st, X, logsub - logB, tp,
                         (unsigned)std::max<size_t>(1, tab_period), (unsigned)bpc, col0, (unsigned)S, (unsigned)spl, cp);
      RS_HIP(hipGetLastError());
      return;
    }
#endif
  }
  RS_HIP(hipFuncSetAttribute((const void *)sub_ntt_kernel<MODE, ColPlansT<M>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int thr = (int)std::max<size_t>(64, std::min<size_t>(1024, ((size_t)1 << logB) / 16));
  hipLaunchKernelGGL((sub_ntt_kernel<MODE, ColPlansT<M>>), dim3((unsigned)(ncols * bpc)), dim3(thr), lds, st, X, logB, logsub - logB, tp,
                     (unsigned)std::max<size_t>(1, tab_period), (unsigned)bpc, col0, (unsigned)S, (unsigned)spl, cp);
  RS_HIP(hipGetLastError());
}

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

// one two-dimensional block convolution of `ncols * units` operands of Y/2 blocks each.  MODE 2: against the table
// `tab` ([units][Y][2B] per limb, limbs from limb0 on); MODE 3: against the data spectra `other` ([ncols*units][Y][2][B], same
// layout as Ws); MODE 0: forward only (Ws receives the spectra; no sink).
template <int SRC, int DST, int MODE>
static void bc2_conv(rs_ctx *ctx, Bc2Args a, int logY, size_t ncols, const TabPtrs *tab, const double *other, const ColPlans &cp,
                     hipStream_t st) {
  const size_t Y = (size_t)1 << logY, cu = ncols * (size_t)a.units;
  const dim3 grid((unsigned)(BC2_B / 2 / 256), (unsigned)cu);
  RS_REQUIRE(cu <= 65535 && logY >= 2 && logY <= 5, "two-dimensional block convolution out of range");
  {
    ProfScope prof(ctx, st, "bc2_yfwd_kernel", (double)cu * (double)Y * BC2_B * 12.0, (double)cu * BC2_B * ntt_fp64((double)Y, logY));
    switch (logY) {
      case 2: hipLaunchKernelGGL((bc2_yfwd_kernel<SRC, 2>), grid, dim3(256), 0, st, a, cp); break;
      case 3: hipLaunchKernelGGL((bc2_yfwd_kernel<SRC, 3>), grid, dim3(256), 0, st, a, cp); break;
      case 4: hipLaunchKernelGGL((bc2_yfwd_kernel<SRC, 4>), grid, dim3(256), 0, st, a, cp); break;
      default: hipLaunchKernelGGL((bc2_yfwd_kernel<SRC, 5>), grid, dim3(256), 0, st, a, cp); break;
    }
  }
  {
    const unsigned long long nb = (unsigned long long)(cu * Y * 2);
    const double Bn = (double)BC2_B;
    static const char *const names[4] = {"sub_ntt_wide_kernel<0>", "sub_ntt_wide_kernel<1>", "sub_ntt_wide_kernel<2>", "sub_ntt_wide_kernel<3>"};
    ProfScope prof(ctx, st, names[MODE], (double)nb * Bn * (MODE == 3 ? 24.0 : 16.0),
                   (double)nb * ((MODE >= 2 ? 2.0 : 1.0) * ntt_fp64(Bn, BC2_LOGB) + (MODE >= 2 ? 7.0 * Bn : 0.0)));
    TabPtrs tp{};
    if (MODE == 2) tp = *tab;
    if (MODE == 3) tp.t[0] = other;
    const int wl = (int)(WideShape<13>::TILE * sizeof(double));
    RS_HIP(hipFuncSetAttribute((const void *)sub_ntt_wide_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, wl));
    hipLaunchKernelGGL((sub_ntt_wide_kernel<MODE>), dim3((unsigned)std::min<unsigned long long>(nb, 512)), dim3(256), wl, st, a.Ws, 1, tp,
                       (unsigned)((size_t)a.units * Y * 2), (unsigned)((size_t)a.units * Y * 2), a.col0, a.S, a.slots_per_limb, cp, nb,
                       (const double *)a.Wy);
  }
  if (MODE != 0) {
    ProfScope prof(ctx, st, "bc2_yinv_kernel", (double)cu * (double)Y * BC2_B * 24.0, (double)cu * 2.0 * BC2_B * ntt_fp64((double)Y, logY));
    switch (logY) {
      case 2: hipLaunchKernelGGL((bc2_yinv_kernel<DST, 2>), grid, dim3(256), 0, st, a, cp); break;
      case 3: hipLaunchKernelGGL((bc2_yinv_kernel<DST, 3>), grid, dim3(256), 0, st, a, cp); break;
      case 4: hipLaunchKernelGGL((bc2_yinv_kernel<DST, 4>), grid, dim3(256), 0, st, a, cp); break;
      default: hipLaunchKernelGGL((bc2_yinv_kernel<DST, 5>), grid, dim3(256), 0, st, a, cp); break;
    }
  }
  RS_HIP(hipGetLastError());
}

// multi-pass interpolation of `ncols` columns X[ncols][M] in place; W: workspace [ncols][2M]
template <class M>
static void big_interp(rs_ctx *ctx, const WitnessPlan *P, const ColPlansT<M> &cp, typename ArithOf<M>::T *X, typename ArithOf<M>::T *W,
                       size_t ncols, size_t col0, size_t S, size_t spl, int limb0, hipStream_t st) {
  using T = typename ArithOf<M>::T;
  constexpr bool FP = std::is_same<M, Mod>::value;
  const int logM = P->logM, logB = std::min(g_witness_lds_logM, logM);
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
  // values -> Newton coefficients: one cyclic convolution of length 2M
  a.logtot = a.logsub = logM + 1;
  launch_cross<false, CS_SCALE_PAD, M>(ctx, a, ncols, logB, cp, st);
  for (int i = limb0; i < ctx->L; i++) tp.t[i - limb0] = P->limb[i].d_ehat;
  launch_sub<2, M>(ctx, W, ncols, col0, logM + 1, logM + 1, logB, &tp, (2 * Mlen) >> logB, S, spl, cp, st);
  launch_cross<true, CD_TAKE_LOW, M>(ctx, a, ncols, logB, cp, st);
  // product tree: levels <= logTree inside LDS tiles (the wide kernel takes 2^14 tiles: one multi-pass level less)
  int logTree = logB;
  if constexpr (FP) {
    if (logB == 13 && logM >= 15 && g_witness_tree_ct == 2 && g_witness_tree_log >= 14) logTree = 14;
    launch_tree_tiles(ctx, X, ncols, col0, logM, logTree, S, spl, cp, st);
  } else {
    launch_tree_tiles_generic<M>(ctx, X, ncols, col0, logM, logB, S, spl, cp, st);
  }
  // levels above: F_node = F_left + D_left * F_right with multi-pass transforms of length 2^l
  a.logtot = logM;
  for (int l = logTree + 1; l <= logM; l++) {
    a.l = l;
    a.logsub = l;
    launch_cross<false, CS_FILL_RIGHT, M>(ctx, a, ncols, logB, cp, st);
    for (int i = limb0; i < ctx->L; i++) tp.t[i - limb0] = static_cast<const T *>(P->limb[i].d_dhat) + (size_t)l * Mlen;
    launch_sub<2, M>(ctx, W, ncols, col0, logM, l, logB, &tp, Mlen >> logB, S, spl, cp, st);
    if (l == logM)
      launch_cross<true, CD_COMBINE_CANON, M>(ctx, a, ncols, logB, cp, st);
    else
      launch_cross<true, CD_COMBINE, M>(ctx, a, ncols, logB, cp, st);
  }
}

// multi-pass H = quo(A*B, Z) (+ ZK patch) for `ncols` columns; W1, W2: workspaces [ncols][2M]
template <class M>
static void big_h(rs_ctx *ctx, const WitnessPlan *P, const ColPlansT<M> &cp, const typename ArithOf<M>::T *A,
                  const typename ArithOf<M>::T *B, typename ArithOf<M>::T *H, typename ArithOf<M>::T *W1, typename ArithOf<M>::T *W2,
                  size_t ncols, size_t col0, size_t S, size_t spl, const uint64_t *d1, const uint64_t *d2, const uint64_t *d3,
                  const ColMap &cm, int limb0, hipStream_t st) {
  const int logM = P->logM, logB = std::min(g_witness_lds_logM, logM);
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
  launch_cross<true, CD_PLAIN, M>(ctx, a, ncols, logB, cp, st);
  // U = rev(P) * rev(Z)^-1 mod x^(m-1)
  a.W = W1;
  a.src = W2;
  launch_cross<false, CS_REV_TRUNC, M>(ctx, a, ncols, logB, cp, st);
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

static size_t big_chunk_cols(const WitnessPlan *P) {
  // two [cols][2M] workspaces within ~6 GiB
  const size_t per_col = 4 * P->M * sizeof(double);
  return std::max<size_t>(1, ((size_t)6 << 30) / per_col);
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
    RS_HIP(hipFuncSetAttribute((const void *)bc_fwd_kernel<SRC, CPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((bc_fwd_kernel<SRC, CPS>), dim3((unsigned)(ncols * a.units * a.nxb)), dim3(thr), lds, st, a, cp);
  }
  {
    ProfScope prof(ctx, st, "bc_mac_kernel", nmac * (double)B2 * 8.0 + (double)ncols * a.units * pairs * (double)B2 * 8.0,
                   nmac * ntt_fp64((double)B2, a.bcLog) + (double)ncols * a.units * pairs * (double)B2 * 7.0);
    RS_HIP(hipFuncSetAttribute((const void *)bc_mac_kernel<YK, CPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
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
  return std::max<size_t>(1, ((size_t)6 << 30) / ((P->bc2 ? 12 : 10) * P->M * sizeof(double)));
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
    RS_HIP(hipFuncSetAttribute((const void *)bc_fwd_kernel<BS_CENTER, CPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
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
                       int limb0, hipStream_t st) {
  const int logM = P->logM;
  const size_t Mlen = P->M;
  Bc2Args a{};
  a.src = X;
  a.dst = X;
  a.Wy = (double *)ws_get(ctx, 12, ncols * 2 * Mlen * sizeof(double));
  a.Ws = (double *)ws_get(ctx, 13, ncols * 4 * Mlen * sizeof(double));
  a.logM = logM;
  a.m = (int)P->m;
  a.col0 = col0;
  a.S = (unsigned)S;
  a.slots_per_limb = (unsigned)spl;
  TabPtrs tp{};
  // values -> Newton coefficients: low M terms of (y_k / k!) * ((-1)^k / k!)
  a.units = 1;
  for (int i = limb0; i < ctx->L; i++) tp.t[i - limb0] = P->limb[i].d_b2_e;
  bc2_conv<BS_SCALE, BD_NEWTON, 2>(ctx, a, logM + 1 - BC2_LOGB, ncols, &tp, nullptr, cp, st);
  // product tree: levels <= 14 inside LDS tiles, the levels above as block convolutions F_node = F_left + (x^h + d) * F_right
  launch_tree_tiles(ctx, X, ncols, col0, logM, (g_witness_tree_ct == 2 && g_witness_tree_log >= 14) ? 14 : 13, S, spl, cp, st);
  const int first = (g_witness_tree_ct == 2 && g_witness_tree_log >= 14) ? 15 : 14;
  for (int l = first; l <= logM; l++) {
    a.l = l;
    a.units = (int)(Mlen >> l);
    for (int i = limb0; i < ctx->L; i++)
      tp.t[i - limb0] = static_cast<const double *>(P->limb[i].d_b2_d) + (size_t)(l - P->bcLog - 1) * 2 * Mlen;
    if (l == logM)
      bc2_conv<BS_RIGHT, BD_COMBINE_CANON, 2>(ctx, a, l - BC2_LOGB, ncols, &tp, nullptr, cp, st);
    else
      bc2_conv<BS_RIGHT, BD_COMBINE, 2>(ctx, a, l - BC2_LOGB, ncols, &tp, nullptr, cp, st);
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
  bc2_conv<BS_CENTER, BD_PLAIN_SCALED, 3>(ctx, a, logY, ncols, nullptr, WsA, cp, st);
  // U = rev(P) * rev(Z)^-1 mod x^(m-1);  H_j = U_{m-2-j}
  TabPtrs tp{};
  for (int i = limb0; i < ctx->L; i++) tp.t[i - limb0] = P->limb[i].d_b2_s;
  a.src = Pbuf;
  a.dst = H;
  bc2_conv<BS_REVTRUNC, BD_HFIN, 2>(ctx, a, logY, ncols, &tp, nullptr, cp, st);
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
    if (single_tile_ok(P->logM)) {
      // one launch, tile = M, two workgroups per CU: Newton conversion by the two rooted M-point
      // sub-transforms, then the product tree in place
      launch_tree_tiles(ctx, cols, ncols, 0, P->logM, P->logM, S, slots_per_limb, cp, st, true);
      return;
    }
  }
  if (P->logM <= g_witness_lds_logM) {
    // the 2M convolution tile, or the product tree's tile + scratch when M is below the LDS block size
    const size_t lds = std::max(padded_len(2 * P->M), padded_len(tree_scratch_offset((int)P->M) + P->M)) * sizeof(double);
    RS_HIP(hipFuncSetAttribute((const void *)interp_columns_kernel<ColPlansT<M>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope prof(ctx, st, "interp_columns_kernel", (double)ncols * (double)P->M * 16.0,
                   (double)ncols * (2.0 * ntt_fp64(2.0 * (double)P->M, P->logM + 1) + 21.0 * (double)P->M + tree_fp64((double)P->M, P->logM)));
    hipLaunchKernelGGL(interp_columns_kernel<ColPlansT<M>>, dim3((unsigned)ncols), dim3(col_threads(2 * P->M)), lds, st, cols, P->logM,
                       (unsigned)S, (unsigned)slots_per_limb, cp);
    RS_HIP(hipGetLastError());
    return;
  }
  const size_t chunk = std::min(ncols, big_chunk_cols(P));
  T *W = (T *)ws_get(ctx, 12, chunk * 2 * P->M * sizeof(double));
  for (size_t c0 = 0; c0 < ncols; c0 += chunk) {
    const size_t nc = std::min(chunk, ncols - c0);
    big_interp<M>(ctx, P, cp, cols + c0 * P->M, W, nc, c0, S, slots_per_limb, limb0, st);
  }
}

// H for the S columns of a chunk (A, B, H: [S][M]); `spl` columns per limb, cm locates d1..d3
template <class M>
static void launch_h(rs_ctx *ctx, const WitnessPlan *P, const ColPlansT<M> &cp, const typename ArithOf<M>::T *A,
                     const typename ArithOf<M>::T *B, typename ArithOf<M>::T *H, size_t S, size_t spl, const uint64_t *d1,
                     const uint64_t *d2, const uint64_t *d3, const ColMap &cm, hipStream_t st) {
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
    if (single_tile_ok(P->logM)) {
      const size_t lds1 = padded_len(Mlen) * sizeof(double);
      const int thr = (int)(Mlen / 16);
      // ten M-point transforms, four pointwise products, the ZK patch
      ProfScope prof(ctx, st, "h_tile_kernel", (double)S * (double)Mlen * 24.0,
                     (double)S * (10.0 * ntt_fp64((double)Mlen, P->logM) + (d1 ? 52.0 : 28.0) * (double)Mlen));
#define RS_H_LAUNCH(KERN)                                                                                            \
  do {                                                                                                               \
    RS_HIP(hipFuncSetAttribute((const void *)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));         \
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
    RS_HIP(hipFuncSetAttribute((const void *)h_columns_kernel<ColPlansT<M>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
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
                          const void *d_const_, hipStream_t st) {
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
  const dim3 tgrid((unsigned)((C + 31) / 32), (unsigned)((M + 31) / 32));
  const dim3 tgrid64((unsigned)((C + 63) / 64), (unsigned)((M + 31) / 32));
  const T *ptab = reinterpret_cast<const T *>(cs->d_ptab);
  for (int w = 0; w < 3; w++)
    for (int kind = 0; kind < 3; kind++) {  // io, full, constant part
      const int k = kind == 2 ? 7 + w : 3 * kind + w;
      if (!needed(k)) continue;
      // per (row, slot): 8 bytes of assignment per non-zero + 8 bytes of column written
      ProfScope prof(ctx, st, "r1cs_eval_cols_kernel", (double)C * 8.0 * ((double)cs->nnz[w] + (double)M), 7.0 * (double)C * (double)cs->nnz[w]);
      hipLaunchKernelGGL(r1cs_eval_cols_kernel<M_>, tgrid64, dim3(256), 0, st, cs->d_row_ptr[w], cs->d_col[w], coeff[w],
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
  if (needH) launch_h<M_>(ctx, P, cp, colv(3), colv(4), colv(6), C, (size_t)cm.ns, d1, d2, d3, cm, st);
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
        hipLaunchKernelGGL(transpose_out_kernel<T>, tgrid, dim3(256), 0, st, colv(k), outs[k], m, C, M, cm);
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
        hipLaunchKernelGGL(io_mid_out_kernel<M_>, tgrid64, dim3(256), 0, st, colv(3 + w), io, io_cols, d_asg, cst, outs[w],
                           outs[3 + w], m, C, M, qmod, cm);
      } else {  // io alone: no column work at all
        const unsigned by = (unsigned)((C / 2 + 255) / 256);
        ProfScope prof(ctx, st, "io_coeff_kernel", (double)C * (double)m * 8.0, (double)C * (double)m * 7.0 * io.count);
        hipLaunchKernelGGL(io_coeff_kernel<M_>, dim3((unsigned)m, by), dim3(256), 0, st, io, io_cols, d_asg, outs[w], C, M, qmod, cm);
      }
    }
  }
  RS_HIP(hipGetLastError());
  if (needH) {
    {
      ProfScope prof(ctx, st, "transpose_out_kernel", (double)C * (double)m * 16.0, 0.0);
      hipLaunchKernelGGL(transpose_out_kernel<T>, tgrid, dim3(256), 0, st, colv(6), outs[6], std::min(m + 1, M), C, M, cm);
    }
    if (m == M)  // row m does not exist in the M-row column tile
      hipLaunchKernelGGL(h_top_kernel<M_>, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, outs[6] + m * cm.out_stride(), d1, d2, C,
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
                              bool compact) {
  using T = typename ArithOf<M_>::T;
  RS_REQUIRE((d1 && d2 && d3) || (!d1 && !d2 && !d3), "d1,d2,d3 must be all set or all null");
  if (nslots < 0) nslots = ctx->N - slot0;
  RS_REQUIRE(slot0 >= 0 && nslots >= 2 && slot0 + nslots <= ctx->N && !(slot0 & 1) && !(nslots & 1),
             "slot range must be even-aligned and inside the ring");
  const size_t m = cs->m;
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
      witness_chunk<M_>(ctx, cs, P, d_asg, d1, d2, d3, outs, cm, std::min(per, L - l0), d_const, st);
    }
  } else {
    const int piece = (int)std::max<size_t>(64, (budget_cols / 64) * 64);
    for (int l0 = 0; l0 < L; l0++)
      for (int s0 = 0; s0 < nslots; s0 += piece) {
        cm.limb0 = l0;
        cm.slot0 = slot0 + s0;
        cm.ns = std::min(piece, nslots - s0);
        witness_chunk<M_>(ctx, cs, P, d_asg, d1, d2, d3, outs, cm, 1, d_const, st);
      }
  }
}
void witness_run(rs_ctx *ctx, const rs_r1cs *cs, const uint64_t *d_asg, const uint64_t *d1, const uint64_t *d2,
                 const uint64_t *d3, uint64_t *const outs[7], uint64_t *h_Z, hipStream_t st, int slot0 = 0, int nslots = -1,
                 bool compact = false) {
  RS_DISPATCH_ARITH(ctx, (witness_run_arith<Mod>(ctx, cs, d_asg, d1, d2, d3, outs, h_Z, st, slot0, nslots, compact)),
                    (witness_run_arith<ModI>(ctx, cs, d_asg, d1, d2, d3, outs, h_Z, st, slot0, nslots, compact)));
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
  hipLaunchKernelGGL(transpose_out_kernel<T>, tgrid, dim3(256), 0, st, colbuf, d_out, n, S_, M, cm);
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

int rs_interpolate(rs_ctx *ctx, const uint64_t *d_y, uint64_t *d_out, size_t n, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_y && d_out && n >= 1, "null argument");
  WsScope ws_scope(ctx, S(stream));
  RS_DISPATCH_ARITH(ctx, (interpolate_arith<Mod>(ctx, d_y, d_out, n, S(stream))), (interpolate_arith<ModI>(ctx, d_y, d_out, n, S(stream))));
  RS_API_END
}

}  // extern "C"
