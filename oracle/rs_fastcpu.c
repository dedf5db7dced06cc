/*
 * rs_fastcpu.c -- the TIMED CPU leg of bench.py's cpu_baseline (TEST INFRASTRUCTURE ONLY).
 *
 * rs_oracle.c is the checker: it reduces every product with a 128-bit `%` (a hardware divide per butterfly), which
 * is the clearest statement of the arithmetic but several times slower than what Microsoft SEAL -- the reference's
 * backend, un-vendored -- executes.  A CPU baseline timed on it would understate the reference.
 * This file restates EncodingElem::inner_product (ringsnark/seal/seal_ring.tcc:361-433) with the arithmetic SEAL 4.x
 * publishes for it:
 *   - negacyclic NTT: Harvey's lazy butterflies with Shoup-precomputed twiddle quotients, values kept in [0, 4q)
 *     forward / [0, 2q) inverse and corrected once at the end (SEAL util/ntt.cpp ntt_negacyclic_harvey,
 *     util/dwthandler.h Arithmetic<...>::mul_root = multiply_uint_mod_lazy with MultiplyUIntModOperand);
 *   - dyadic products: Barrett reduction of the 128-bit product with the precomputed ratio floor(2^128 / q)
 *     (SEAL util/uintarithsmallmod.h multiply_uint_mod / barrett_reduce_128, as dyadic_product_coeffmod uses it);
 * and the reference's own algorithmic structure per term and ring limb: one BatchEncoder::encode (scatter + inverse NTT
 * mod q_i), K forward NTTs mod Q_j of the centred lift, 2K dyadic products, 2K dyadic additions, and the two full
 * EncodingElem copies of seal_ring.tcc:417,485-488.  Results are bit-identical to rs_oracle.c (tests/test_oracle.py).
 */
#include <stdlib.h>
#include <string.h>

#include "rs_oracle.h"

typedef unsigned __int128 u128;

typedef struct rsf_ntt {
  uint64_t q, two_q;
  size_t n;
  uint64_t *w, *wq;   /* forward root powers (bit-reversed order) and their Shoup quotients floor(w 2^64 / q) */
  uint64_t *iw, *iwq; /* inverse */
  uint64_t ninv, ninvq;
  uint64_t ratio_hi, ratio_lo; /* floor(2^128 / q) */
} rsf_ntt;

struct rsf_ctx {
  const rso_ctx *base;
  rsf_ntt plain[RSO_MAXL], coeff[RSO_MAXK];
};

static inline uint64_t shoup_quot(uint64_t w, uint64_t q) { return (uint64_t)(((u128)w << 64) / q); }
/* x * w mod q in [0, 2q): multiply_uint_mod_lazy */
static inline uint64_t mul_lazy(uint64_t x, uint64_t w, uint64_t wq, uint64_t q) {
  const uint64_t hi = (uint64_t)(((u128)x * wq) >> 64);
  return x * w - hi * q;
}
/* Barrett reduction of a 128-bit value below q 2^64 (barrett_reduce_128) */
static inline uint64_t barrett128(u128 z, const rsf_ntt *t) {
  const uint64_t z0 = (uint64_t)z, z1 = (uint64_t)(z >> 64);
  /* floor(z * ratio / 2^128), keeping the carries SEAL keeps */
  const uint64_t carry = (uint64_t)(((u128)z0 * t->ratio_lo) >> 64);
  const u128 mid1 = (u128)z0 * t->ratio_hi + carry;
  const u128 mid2 = (u128)z1 * t->ratio_lo + (uint64_t)mid1;
  const uint64_t quo = z1 * t->ratio_hi + (uint64_t)(mid1 >> 64) + (uint64_t)(mid2 >> 64);
  uint64_t r = z0 - quo * t->q;
  return r >= t->q ? r - t->q : r;
}

static void table_init(rsf_ntt *f, const rso_ntt *t) {
  f->q = t->q;
  f->two_q = 2 * t->q;
  f->n = t->n;
  f->w = (uint64_t *)malloc(sizeof(uint64_t) * 4 * t->n);
  f->wq = f->w + t->n;
  f->iw = f->wq + t->n;
  f->iwq = f->iw + t->n;
  for (size_t k = 0; k < t->n; k++) {
    f->w[k] = t->rp[k];
    f->wq[k] = shoup_quot(t->rp[k], t->q);
    f->iw[k] = t->irp[k];
    f->iwq[k] = shoup_quot(t->irp[k], t->q);
  }
  f->ninv = t->ninv;
  f->ninvq = shoup_quot(t->ninv, t->q);
  /* floor(2^128 / q): long division of 2^128 by a 64-bit q */
  const u128 top = (((u128)1) << 127) / t->q; /* floor(2^127 / q) */
  u128 ratio = top << 1;
  /* correct the doubled floor: 2^128 - ratio q in [0, 2q) */
  const u128 rem = (u128)0 - ratio * t->q; /* arithmetic mod 2^128 */
  if (rem >= t->q) ratio += 1;
  f->ratio_hi = (uint64_t)(ratio >> 64);
  f->ratio_lo = (uint64_t)ratio;
}

rsf_ctx *rsf_ctx_create(const rso_ctx *base) {
  rsf_ctx *c = (rsf_ctx *)calloc(1, sizeof(rsf_ctx));
  c->base = base;
  for (int i = 0; i < base->L; i++) table_init(&c->plain[i], base->plain[i]);
  for (int j = 0; j < base->K; j++) table_init(&c->coeff[j], base->coeff[j]);
  return c;
}
void rsf_ctx_destroy(rsf_ctx *c) {
  if (!c) return;
  for (int i = 0; i < c->base->L; i++) free(c->plain[i].w);
  for (int j = 0; j < c->base->K; j++) free(c->coeff[j].w);
  free(c);
}

/* forward: natural in -> bit-reversed out, inputs in [0, q), outputs canonical */
static void ntt_fwd_lazy(const rsf_ntt *t, uint64_t *a) {
  const uint64_t q = t->q, two_q = t->two_q;
  const size_t n = t->n;
  for (size_t m = 1, gap = n >> 1; m < n; m <<= 1, gap >>= 1)
    for (size_t i = 0; i < m; i++) {
      const uint64_t W = t->w[m + i], Wq = t->wq[m + i];
      uint64_t *x = a + 2 * i * gap, *y = x + gap;
      for (size_t j = 0; j < gap; j++) {
        uint64_t u = x[j];
        u -= (u >= two_q) ? two_q : 0; /* guard: u in [0, 2q) */
        const uint64_t v = mul_lazy(y[j], W, Wq, q);
        x[j] = u + v;         /* [0, 4q) */
        y[j] = u + two_q - v; /* [0, 4q) */
      }
    }
  for (size_t j = 0; j < n; j++) {
    uint64_t v = a[j];
    v -= (v >= two_q) ? two_q : 0;
    a[j] = v >= q ? v - q : v;
  }
}
/* inverse: bit-reversed in -> natural out, scaled by n^-1, outputs canonical */
static void ntt_inv_lazy(const rsf_ntt *t, uint64_t *a) {
  const uint64_t q = t->q, two_q = t->two_q;
  const size_t n = t->n;
  for (size_t m = n >> 1, gap = 1; m >= 1; m >>= 1, gap <<= 1)
    for (size_t i = 0; i < m; i++) {
      const uint64_t W = t->iw[m + i], Wq = t->iwq[m + i];
      uint64_t *x = a + 2 * i * gap, *y = x + gap;
      for (size_t j = 0; j < gap; j++) {
        const uint64_t u = x[j], v = y[j]; /* both in [0, 2q) */
        uint64_t s = u + v;
        x[j] = s >= two_q ? s - two_q : s;
        y[j] = mul_lazy(u + two_q - v, W, Wq, q);
      }
    }
  for (size_t j = 0; j < n; j++) {
    const uint64_t v = mul_lazy(a[j], t->ninv, t->ninvq, q);
    a[j] = v >= q ? v - q : v;
  }
}

void rsf_ntt_fwd(const rsf_ctx *c, int modset, int index, uint64_t *a) { ntt_fwd_lazy(modset ? &c->coeff[index] : &c->plain[index], a); }
void rsf_ntt_inv(const rsf_ctx *c, int modset, int index, uint64_t *a) { ntt_inv_lazy(modset ? &c->coeff[index] : &c->plain[index], a); }

static inline uint64_t lift(uint64_t cf, uint64_t t, uint64_t Q) { /* as rs_oracle.c lift_centered */
  if (cf >= ((t + 1) >> 1)) {
    const uint64_t neg = (t - cf) % Q;
    return neg ? Q - neg : 0;
  }
  return cf % Q;
}

/* tmp = enc * ring  (EncodingElem::operator*=, seal_ring.tcc:530-544) */
static void enc_mul_ring_fast(const rsf_ctx *c, uint64_t *enc, const uint64_t *ring, uint64_t *plain, uint64_t *P) {
  const rso_ctx *b = c->base;
  const size_t n = (size_t)b->N_enc;
  for (int i = 0; i < b->L; i++) {
    memset(plain, 0, sizeof(uint64_t) * n);
    for (int x = 0; x < b->N; x++) plain[b->index_map[x]] = ring[(size_t)i * b->N + x];
    ntt_inv_lazy(&c->plain[i], plain);
    uint64_t *ct = enc + (size_t)i * 2 * b->K * n;
    for (int j = 0; j < b->K; j++) {
      const rsf_ntt *t = &c->coeff[j];
      for (size_t x = 0; x < n; x++) P[x] = lift(plain[x], b->q[i], b->Q[j]);
      ntt_fwd_lazy(t, P);
      for (int comp = 0; comp < 2; comp++) {
        uint64_t *p = ct + ((size_t)comp * b->K + j) * n;
        for (size_t x = 0; x < n; x++) p[x] = barrett128((u128)p[x] * P[x], t);
      }
    }
  }
}
static void enc_add_fast(const rso_ctx *b, uint64_t *enc, const uint64_t *o) {
  const size_t n = (size_t)b->N_enc;
  for (int i = 0; i < b->L; i++)
    for (int comp = 0; comp < 2; comp++)
      for (int j = 0; j < b->K; j++) {
        const size_t off = (((size_t)i * 2 + comp) * b->K + j) * n;
        const uint64_t Q = b->Q[j];
        for (size_t x = 0; x < n; x++) {
          const uint64_t s = enc[off + x] + o[off + x];
          enc[off + x] = s >= Q ? s - Q : s;
        }
      }
}

#ifdef _OPENMP
#include <omp.h>
#endif
/* EncodingElem::inner_product with the terms spread over `threads` threads (threads <= 0: all cores); window as in
 * rso_inner_product_mt.  Same results as rso_inner_product / rso_inner_product_mt. */
size_t rsf_inner_product_mt(const rsf_ctx *c, const uint64_t *encs, size_t window, const uint64_t *rings, const uint8_t *kinds,
                            size_t T, uint64_t *out, int threads) {
  const rso_ctx *b = c->base;
  const size_t ew = rso_enc_words(b), rw = rso_ring_words(b), n = (size_t)b->N_enc;
#ifdef _OPENMP
  const int nt = threads > 0 ? threads : omp_get_max_threads();
#else
  const int nt = 1;
  (void)threads;
#endif
  uint64_t *part = (uint64_t *)calloc((size_t)nt * ew, sizeof(uint64_t));
  size_t *used = (size_t *)calloc((size_t)nt, sizeof(size_t));
#pragma omp parallel num_threads(nt)
  {
#ifdef _OPENMP
    const int id = omp_get_thread_num();
#else
    const int id = 0;
#endif
    uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * (ew + 2 * n)), *plain = tmp + ew, *P = plain + n;
    uint64_t *acc = part + (size_t)id * ew;
#pragma omp for schedule(static)
    for (long long t = 0; t < (long long)T; t++) {
      const int kind = kinds ? kinds[t] : RSO_KIND_POLY;
      const uint64_t *r = rings + (size_t)t * rw;
      if (kind == RSO_KIND_POLY && rso_ring_is_zero(b, r)) continue;
      memcpy(tmp, encs + (window ? (size_t)t % window : (size_t)t) * ew, sizeof(uint64_t) * ew); /* seal_ring.tcc:417 */
      if (kind != RSO_KIND_ONE) enc_mul_ring_fast(c, tmp, r, plain, P);
      enc_add_fast(b, acc, tmp);
      used[id]++;
    }
    free(tmp);
  }
  size_t total = 0;
  memset(out, 0, sizeof(uint64_t) * ew);
  for (int k = 0; k < nt; k++) {
    enc_add_fast(b, out, part + (size_t)k * ew);
    total += used[k];
  }
  free(part);
  free(used);
  return total;
}

/* ONE RING LIMB of EncodingElem::inner_product (seal_ring.tcc:361-433), every (component, prime) slab of it at once:
 *     acc[comp][j][.] += sum_{t < T, rows[t] != 0}  ct[(t0 + t) % window][comp][j][.] * NTT_Qj(lift(iNTT_qi(scatter(rows[t]))))
 * ct: the limb's slice of the key vector, `window` stored elements of [2][K][N_enc] words, ct_stride words apart (tiled
 * key: pass window >= t0 + T for an ordinary one); rows [T][N]: the limb's N values of each coefficient; acc [2][K][N_enc]
 * canonical residues.  The plaintext of a term is transformed ONCE for the limb's 2 K slabs (the per-slab form,
 * rso_inner_product_slab, repeats it per slab): what makes a check of ALL 96 slabs of a headline proof affordable
 * (tests/proof_check.py).  Terms over `threads` threads (<= 0: all cores).  Same arithmetic as rsf_inner_product_mt. */
void rsf_inner_product_limb(const rsf_ctx *c, int limb, const uint64_t *ct, size_t ct_stride, size_t window, size_t t0,
                            const uint64_t *rows, size_t T, uint64_t *acc, int threads) {
  const rso_ctx *b = c->base;
  const size_t n = (size_t)b->N_enc, sw = (size_t)2 * b->K * n;
#ifdef _OPENMP
  const int nt = threads > 0 ? threads : omp_get_max_threads();
#else
  const int nt = 1;
  (void)threads;
#endif
  uint64_t *part = (uint64_t *)calloc((size_t)nt * sw, sizeof(uint64_t));
#pragma omp parallel num_threads(nt)
  {
#ifdef _OPENMP
    const int id = omp_get_thread_num();
#else
    const int id = 0;
#endif
    uint64_t *plain = (uint64_t *)malloc(sizeof(uint64_t) * 2 * n), *P = plain + n, *a = part + (size_t)id * sw;
#pragma omp for schedule(static)
    for (long long t = 0; t < (long long)T; t++) {
      const uint64_t *row = rows + (size_t)t * b->N;
      int nz = 0;
      for (int x = 0; x < b->N && !nz; x++) nz = row[x] != 0;
      if (!nz) continue; /* is_zero terms are skipped (seal_ring.tcc:391-396): they contribute nothing */
      memset(plain, 0, sizeof(uint64_t) * n);
      for (int x = 0; x < b->N; x++) plain[b->index_map[x]] = row[x];
      ntt_inv_lazy(&c->plain[limb], plain);
      const uint64_t *cw = ct + (((size_t)t0 + (size_t)t) % window) * ct_stride;
      for (int j = 0; j < b->K; j++) {
        const rsf_ntt *tab = &c->coeff[j];
        const uint64_t Q = b->Q[j];
        for (size_t x = 0; x < n; x++) P[x] = lift(plain[x], b->q[limb], Q);
        ntt_fwd_lazy(tab, P);
        for (int comp = 0; comp < 2; comp++) {
          const uint64_t *p = cw + ((size_t)comp * b->K + j) * n;
          uint64_t *o = a + ((size_t)comp * b->K + j) * n;
          for (size_t x = 0; x < n; x++) {
            const uint64_t s = o[x] + barrett128((u128)p[x] * P[x], tab);
            o[x] = s >= Q ? s - Q : s;
          }
        }
      }
    }
    free(plain);
  }
  for (int k = 0; k < nt; k++)
    for (int cj = 0; cj < 2 * b->K; cj++) {
      const uint64_t Q = b->Q[cj % b->K];
      uint64_t *o = acc + (size_t)cj * n;
      const uint64_t *p = part + (size_t)k * sw + (size_t)cj * n;
      for (size_t x = 0; x < n; x++) {
        const uint64_t s = o[x] + p[x];
        o[x] = s >= Q ? s - Q : s;
      }
    }
  free(part);
}
