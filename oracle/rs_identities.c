/*
 * rs_identities.c -- COMPLETE check of a witness map at sizes where the reference's O(m^2) map cannot be run
 * (TEST INFRASTRUCTURE ONLY; part of librs_oracle.so, see rs_oracle.h).
 *
 * reductions/r1cs_to_qrp/r1cs_to_qrp.tcc:149-259 defines every output as an interpolant over the nodes 0..m-1
 * (util/evaluation_domain.tcc:8-13) of evaluations of the constraint system, or by an exact division:
 *     X_mid = interpolate(evaluate_X(0 || aux))                         :166-187
 *     X_io  = interpolate(evaluate_X(primary || 0))                     :189-208
 *     H     = d2 A + d1 B - d3 + d1 d2 Z + (A B - C) / Z                :225-253,  A, B, C over the full assignment
 * A polynomial of degree < m is determined by its values on the nodes, so "vector V is the interpolant of the
 * evaluations y" is the polynomial identity  V(x) = sum_j y_j L_j(x),  L_j the Lagrange basis of the nodes
 * (util/polynomials.tcc:10-43 builds exactly that sum), and the H equation, multiplied by Z, is one as well.  Both are
 * tested at points r drawn by the caller (Schwartz-Zippel: a wrong vector passes one point with probability
 * <= (m+1)/q < 2^-25 for the configuration primes; the callers use two or more points).  Per slot and point this is
 * O(m + nnz) products instead of the map's O(m^2); every slot of a limb uses the same r, so L_j(r) and r^k are
 * computed once and the work per slot is a handful of dot products.
 *
 * Nothing here shares code or tables with the device library: inputs are the constraint system, the assignment and
 * the device's output vectors, all as plain host arrays.
 */
#include "rs_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

static inline uint64_t addm(uint64_t a, uint64_t b, uint64_t q) {
  uint64_t s = a + b;
  return s >= q ? s - q : s;
}
static inline uint64_t subm(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }
static inline uint64_t mulm(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }
static uint64_t powm(uint64_t a, uint64_t e, uint64_t q) {
  uint64_t r = 1 % q;
  while (e) {
    if (e & 1) r = mulm(r, a, q);
    a = mulm(a, a, q);
    e >>= 1;
  }
  return r;
}

/* x mod q; most values met here are small (absent constant terms, sums of a few products of 44-bit residues) */
static inline uint64_t red128(u128 x, uint64_t q) {
  const uint64_t hi = (uint64_t)(x >> 64), lo = (uint64_t)x;
  if (!hi) return lo < q ? lo : lo % q;
  return (uint64_t)(x % q);
}

/* how many products of two residues < q a 128-bit accumulator (holding one reduced value) takes before a reduction */
static size_t burst_of(uint64_t q) {
  int bits = 64 - __builtin_clzll(q);
  int room = 127 - 2 * bits;
  if (room > 30) room = 30;
  return room < 1 ? 1 : (size_t)1 << room;
}

#define RSI_BLOCK 32 /* slots per work item: 256 contiguous bytes of every row */
#define RSI_MAXP 4

/* L_j(r) = prod_{i != j} (r - i) / prod_{i != j} (j - i) for the nodes 0..m-1; r must not be a node.
 * prod_{i != j}(j - i) = (-1)^(m-1-j) j! (m-1-j)!. */
static void lagrange_weights(uint64_t q, size_t m, uint64_t r, uint64_t *Lw, uint64_t *Zr) {
  uint64_t *pre = (uint64_t *)malloc(sizeof(uint64_t) * (m + 1)), *suf = (uint64_t *)malloc(sizeof(uint64_t) * (m + 1));
  uint64_t *fact = (uint64_t *)malloc(sizeof(uint64_t) * (m + 1));
  pre[0] = 1 % q;
  for (size_t j = 0; j < m; j++) pre[j + 1] = mulm(pre[j], subm(r, (uint64_t)j % q, q), q);
  suf[m] = 1 % q;
  for (size_t j = m; j-- > 0;) suf[j] = mulm(suf[j + 1], subm(r, (uint64_t)j % q, q), q);
  fact[0] = 1 % q;
  for (size_t j = 1; j <= m; j++) fact[j] = mulm(fact[j - 1], (uint64_t)j % q, q);
  /* 1 / (j! (m-1-j)!) for all j from ONE inversion: inv_fact[m-1] by Fermat, then downwards */
  uint64_t *ifact = (uint64_t *)malloc(sizeof(uint64_t) * (m + 1));
  ifact[m - 1] = powm(fact[m - 1], q - 2, q);
  for (size_t j = m - 1; j > 0; j--) ifact[j - 1] = mulm(ifact[j], (uint64_t)j % q, q);
  for (size_t j = 0; j < m; j++) {
    uint64_t w = mulm(mulm(pre[j], suf[j + 1], q), mulm(ifact[j], ifact[m - 1 - j], q), q);
    Lw[j] = ((m - 1 - j) & 1) ? (w ? q - w : 0) : w;
  }
  *Zr = pre[m];
  free(pre);
  free(suf);
  free(fact);
  free(ifact);
}

/* See rs_oracle.h.  Returns the number of slots with at least one failing identity. */
size_t rso_witness_identities(uint64_t q, size_t S, const rso_r1cs *cs, int limb, const uint64_t *assignment,
                              size_t asg_stride, const uint64_t *d1, const uint64_t *d2, const uint64_t *d3,
                              const rso_wm_vectors *v, const uint64_t *points, int n_points, uint8_t *bad, int threads) {
  const size_t m = cs->m, ni = cs->n_inputs;
  if (n_points < 1 || n_points > RSI_MAXP || m < 1) return (size_t)-1;
  if (v->blocked && S % RSI_BLOCK) return (size_t)-1;
  if (m >= q) return (size_t)-1; /* the nodes 0..m-1 must be distinct residues */
  const size_t burst = burst_of(q);
  uint64_t *Lw[RSI_MAXP], *rp[RSI_MAXP], Zr[RSI_MAXP];
  for (int p = 0; p < n_points; p++) {
    const uint64_t r = points[p] % q;
    if (r < m) return (size_t)-1;
    Lw[p] = (uint64_t *)malloc(sizeof(uint64_t) * m);
    rp[p] = (uint64_t *)malloc(sizeof(uint64_t) * (m + 1));
    lagrange_weights(q, m, r, Lw[p], &Zr[p]);
    rp[p][0] = 1 % q;
    for (size_t k = 1; k <= m; k++) rp[p][k] = mulm(rp[p][k - 1], r, q);
  }
  const uint64_t *vec[7] = {v->A_io, v->B_io, v->C_io, v->A_mid, v->B_mid, v->C_mid, v->H};
  size_t n_bad = 0;
  const long long n_blocks = (long long)((S + RSI_BLOCK - 1) / RSI_BLOCK);
#ifdef _OPENMP
  const int nt = threads > 0 ? threads : omp_get_max_threads();
#else
  (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt) reduction(+ : n_bad)
  for (long long blk = 0; blk < n_blocks; blk++) {
    const size_t lo = (size_t)blk * RSI_BLOCK, w = (S - lo < RSI_BLOCK) ? S - lo : RSI_BLOCK;
    /* blocked layout: the 32 slots of a block are contiguous row after row -- [S/32][rows][32] -- so a block streams
     * through memory instead of touching one page per row */
    const uint64_t *a_base = v->blocked ? assignment + (size_t)blk * cs->n_vars * RSI_BLOCK : assignment + lo;
    const size_t a_stride = v->blocked ? RSI_BLOCK : asg_stride;
    /* Lagrange sums of the three kinds of evaluation of a, b, c -- io: constant + primary terms, aux: auxiliary
     * terms, cst: constant terms (the reference's io AND mid passes both see index-0 terms, :175-201) */
    u128 lag[RSI_MAXP][3][3][RSI_BLOCK];
    u128 hor[RSI_MAXP][7][RSI_BLOCK];
    memset(lag, 0, sizeof lag);
    memset(hor, 0, sizeof hor);
    size_t since = 0;
    for (size_t i = 0; i <= m; i++) {
      if (i < m) {
        for (int which = 0; which < 3; which++) {
          const uint32_t *rowp = cs->row_ptr[which], *col = cs->col[which];
          const uint64_t *cf = cs->coeff[which] + (size_t)limb * cs->nnz[which];
          const int32_t *pidx = cs->pidx[which];
          u128 ev[3][RSI_BLOCK];
          memset(ev, 0, sizeof ev);
          size_t terms = 0;
          for (uint32_t e = rowp[i]; e < rowp[i + 1]; e++) {
            const uint32_t c = col[e];
            const int kind = c == 0 ? 2 : ((size_t)(c - 1) < ni ? 0 : 1);
            const uint64_t *a = c ? a_base + (size_t)(c - 1) * a_stride : NULL;
            if (pidx && pidx[e] >= 0) { /* coefficient = a general ring element (rs_oracle.h, rso_r1cs) */
              const uint64_t *pc = cs->ptab + ((size_t)pidx[e] * cs->ptab_L + (size_t)limb) * cs->ptab_N + cs->ptab_slot0 + lo;
              for (size_t s = 0; s < w; s++) ev[kind][s] += a ? (u128)a[s] * (pc[s] % q) : (u128)(pc[s] % q);
            } else {
              const uint64_t cc = cf[e] % q;
              if (a)
                for (size_t s = 0; s < w; s++) ev[kind][s] += (u128)a[s] * cc;
              else
                for (size_t s = 0; s < w; s++) ev[kind][s] += cc;
            }
            if (++terms >= burst) {
              for (int k = 0; k < 3; k++)
                for (size_t s = 0; s < w; s++) ev[k][s] %= q;
              terms = 0;
            }
          }
          for (size_t s = 0; s < w; s++) {
            const uint64_t e_cst = red128(ev[2][s], q), e_io = addm(red128(ev[0][s], q), e_cst, q), e_aux = red128(ev[1][s], q);
            for (int p = 0; p < n_points; p++) {
              lag[p][which][0][s] += (u128)e_io * Lw[p][i];
              lag[p][which][1][s] += (u128)e_aux * Lw[p][i];
              lag[p][which][2][s] += (u128)e_cst * Lw[p][i];
            }
          }
        }
      }
      for (int k = 0; k < 7; k++) {
        if (!vec[k] || (i == m && k != 6)) continue; /* H has m + 1 rows (:225-253), the others m */
        const uint64_t *row = v->blocked ? vec[k] + ((size_t)blk * (m + (k == 6)) + i) * RSI_BLOCK : vec[k] + i * v->stride[k] + lo;
        for (int p = 0; p < n_points; p++) {
          const uint64_t rk = rp[p][i];
          for (size_t s = 0; s < w; s++) hor[p][k][s] += (u128)row[s] * rk;
        }
      }
      if (++since >= burst || i == m) {
        for (int p = 0; p < n_points; p++) {
          for (int a = 0; a < 9; a++)
            for (size_t s = 0; s < w; s++) (&lag[p][0][0][0])[a * RSI_BLOCK + s] %= q;
          for (int k = 0; k < 7; k++)
            for (size_t s = 0; s < w; s++) hor[p][k][s] %= q;
        }
        since = 0;
      }
    }
    for (size_t s = 0; s < w; s++) {
      uint8_t fail = 0;
      for (int p = 0; p < n_points; p++) {
        uint64_t full[3];
        for (int which = 0; which < 3; which++) {
          const uint64_t io = (uint64_t)lag[p][which][0][s], aux = (uint64_t)lag[p][which][1][s],
                         cst = (uint64_t)lag[p][which][2][s];
          full[which] = addm(io, aux, q);
          if (vec[which] && (uint64_t)hor[p][which][s] != io) fail |= (uint8_t)(1u << which);
          if (vec[3 + which] && (uint64_t)hor[p][3 + which][s] != addm(aux, cst, q)) fail |= (uint8_t)(1u << (3 + which));
        }
        if (vec[6]) {
          const uint64_t x1 = d1 ? d1[lo + s] % q : 0, x2 = d2 ? d2[lo + s] % q : 0, x3 = d3 ? d3[lo + s] % q : 0;
          /* H Z = A B - C + Z (d2 A + d1 B - d3 + d1 d2 Z) */
          uint64_t patch = addm(mulm(x2, full[0], q), mulm(x1, full[1], q), q);
          patch = subm(patch, x3, q);
          patch = addm(patch, mulm(mulm(x1, x2, q), Zr[p], q), q);
          const uint64_t rhs = addm(subm(mulm(full[0], full[1], q), full[2], q), mulm(Zr[p], patch, q), q);
          if (mulm((uint64_t)hor[p][6][s], Zr[p], q) != rhs) fail |= 1u << 6;
        }
      }
      if (bad) bad[lo + s] = fail;
      n_bad += fail != 0;
    }
  }
  /* Z itself (util/evaluation_domain.tcc:54-60): Z(r) = prod (r - i) */
  if (v->Z) {
    for (int p = 0; p < n_points && n_bad != (size_t)-2; p++) {
      u128 acc = 0;
      for (size_t k = 0; k <= m; k++) acc = (acc + (u128)(v->Z[k] % q) * rp[p][k]) % q;
      if ((uint64_t)acc != Zr[p]) n_bad = (size_t)-2;
    }
  }
  for (int p = 0; p < n_points; p++) {
    free(Lw[p]);
    free(rp[p]);
  }
  return n_bad;
}
