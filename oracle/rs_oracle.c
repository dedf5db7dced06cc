/*
 * rs_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see rs_oracle.h for the parity status).
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates, or
 * -- for arithmetic that lives in the un-vendored Microsoft SEAL 4.x / SEAL-Polytools
 * submodules -- the published SEAL algorithm it follows.
 */
#include "rs_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------------
 * modular arithmetic (SEAL util/uintarithsmallmod.h: multiply_uint_mod, exponentiate_uint_mod,
 * try_invert_uint_mod).  Results are canonical residues, so the reduction method is free.
 * ---------------------------------------------------------------------------------------- */
#ifndef RSO_FAST_MULMOD
uint64_t rso_mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }
#else
/* librs_oracle_fast.so (the TIMED cpu_baseline leg of bench.py, never the checker): the same function with SEAL's
 * Barrett reduction (util/uintarithsmallmod.h barrett_reduce_128) instead of a hardware divide per product; the
 * ratio floor(2^128 / q) of the last modulus is cached per thread.  Same canonical results (tests/test_oracle.py). */
/* initial-exec: a plain %fs-relative load (the default dynamic TLS model of a shared object calls __tls_get_addr per use) */
static __thread uint64_t tl_q __attribute__((tls_model("initial-exec"))), tl_hi __attribute__((tls_model("initial-exec"))),
    tl_lo __attribute__((tls_model("initial-exec")));
uint64_t rso_mulmod(uint64_t a, uint64_t b, uint64_t q) {
  if (q != tl_q) {
    u128 ratio = ((((u128)1) << 127) / q) << 1;
    if ((u128)0 - ratio * q >= q) ratio += 1;
    tl_q = q;
    tl_hi = (uint64_t)(ratio >> 64);
    tl_lo = (uint64_t)ratio;
  }
  const u128 z = (u128)a * b;
  const uint64_t z0 = (uint64_t)z, z1 = (uint64_t)(z >> 64);
  const uint64_t carry = (uint64_t)(((u128)z0 * tl_lo) >> 64);
  const u128 mid1 = (u128)z0 * tl_hi + carry;
  const u128 mid2 = (u128)z1 * tl_lo + (uint64_t)mid1;
  const uint64_t quo = z1 * tl_hi + (uint64_t)(mid1 >> 64) + (uint64_t)(mid2 >> 64);
  const uint64_t r = z0 - quo * q;
  return r >= q ? r - q : r;
}
#endif
static inline uint64_t addmod(uint64_t a, uint64_t b, uint64_t q) {
  uint64_t s = a + b;
  return s >= q ? s - q : s;
}
static inline uint64_t submod(uint64_t a, uint64_t b, uint64_t q) {
  return a >= b ? a - b : a + q - b;
}
static inline uint64_t negmod(uint64_t a, uint64_t q) { return a ? q - a : 0; }
uint64_t rso_powmod(uint64_t a, uint64_t e, uint64_t q) {
  uint64_t r = 1 % q;
  a %= q;
  while (e) {
    if (e & 1) r = rso_mulmod(r, a, q);
    a = rso_mulmod(a, a, q);
    e >>= 1;
  }
  return r;
}
uint64_t rso_invmod(uint64_t a, uint64_t q) { return rso_powmod(a, q - 2, q); }

/* deterministic Miller-Rabin for 64-bit (SEAL util::is_prime is probabilistic; same set). */
int rso_is_prime(uint64_t n) {
  if (n < 2) return 0;
  static const uint64_t small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  for (size_t i = 0; i < 12; i++) {
    if (n == small[i]) return 1;
    if (n % small[i] == 0) return 0;
  }
  uint64_t d = n - 1;
  int r = 0;
  while (!(d & 1)) {
    d >>= 1;
    r++;
  }
  for (size_t i = 0; i < 12; i++) {
    uint64_t x = rso_powmod(small[i], d, n);
    if (x == 1 || x == n - 1) continue;
    int comp = 1;
    for (int k = 1; k < r; k++) {
      x = rso_mulmod(x, x, n);
      if (x == n - 1) {
        comp = 0;
        break;
      }
    }
    if (comp) return 0;
  }
  return 1;
}

/* SEAL util::get_primes(factor, bit_size, count) (native/src/seal/util/numth.cpp):
 * value = ((2^bit_size - 1) / factor) * factor + 1, step down by factor while > 2^(bit_size-1). */
int rso_get_primes(uint64_t factor, int bit_size, int count, uint64_t *out) {
  uint64_t value = (((uint64_t)1 << bit_size) - 1) / factor * factor + 1;
  uint64_t lower = (uint64_t)1 << (bit_size - 1);
  int found = 0;
  while (found < count && value > lower) {
    if (rso_is_prime(value)) out[found++] = value;
    value -= factor;
  }
  return found == count ? 0 : -1;
}

/* SEAL CoeffModulus::Create(poly_modulus_degree, bit_sizes) (modulus.cpp): count primes needed
 * per bit size, generate them with get_primes(2N, ...), then hand them out from the BACK of
 * each list, i.e. the later-found (smaller) prime of a size is assigned to the earlier slot. */
int rso_coeff_modulus_create(uint64_t factor, const int *bit_sizes, int count, uint64_t *out) {
  int need[64] = {0};
  uint64_t *table[64] = {0};
  int have[64] = {0};
  for (int i = 0; i < count; i++) need[bit_sizes[i]]++;
  for (int b = 0; b < 64; b++) {
    if (!need[b]) continue;
    table[b] = (uint64_t *)malloc(sizeof(uint64_t) * need[b]);
    if (rso_get_primes(factor, b, need[b], table[b])) return -1;
    have[b] = need[b];
  }
  for (int i = 0; i < count; i++) out[i] = table[bit_sizes[i]][--have[bit_sizes[i]]];
  for (int b = 0; b < 64; b++) free(table[b]);
  return 0;
}

/* SEAL util::try_primitive_root + try_minimal_primitive_root (numth.cpp): find any primitive
 * degree-th root, then take the minimum over root * (root^2)^i, i < degree/2.  The candidate
 * search in SEAL is randomised; the MINIMAL root is independent of which one is found first. */
int rso_minimal_primitive_root(uint64_t degree, uint64_t q, uint64_t *root) {
  if ((q - 1) % degree) return -1;
  uint64_t size_quotient = (q - 1) / degree;
  uint64_t r = 0;
  for (uint64_t g = 2; g < q; g++) {
    uint64_t cand = rso_powmod(g, size_quotient, q);
    if (rso_powmod(cand, degree >> 1, q) == q - 1) {
      r = cand;
      break;
    }
  }
  if (!r) return -1;
  uint64_t gen_sq = rso_mulmod(r, r, q), cur = r, best = r;
  for (uint64_t i = 0; i < (degree >> 1); i++) {
    if (cur < best) best = cur;
    cur = rso_mulmod(cur, gen_sq, q);
  }
  *root = best;
  return 0;
}

static uint32_t bitrev32(uint32_t x, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
}

/* ------------------------------------------------------------------------------------------
 * SEAL util::NTTTables::initialize (ntt.cpp): root = minimal primitive 2n-th root;
 * root_powers[bitrev(i)] = root^i.  The inverse table here holds the element-wise inverses
 * (any correct inverse transform yields the same canonical output).
 * ---------------------------------------------------------------------------------------- */
rso_ntt *rso_ntt_create(int logn, uint64_t q) {
  rso_ntt *t = (rso_ntt *)calloc(1, sizeof(rso_ntt));
  t->q = q;
  t->logn = logn;
  t->n = (size_t)1 << logn;
  if (rso_minimal_primitive_root((uint64_t)2 << logn, q, &t->psi)) {
    free(t);
    return NULL;
  }
  t->rp = (uint64_t *)malloc(sizeof(uint64_t) * t->n);
  t->irp = (uint64_t *)malloc(sizeof(uint64_t) * t->n);
  uint64_t p = 1;
  for (size_t i = 0; i < t->n; i++) {
    uint32_t k = bitrev32((uint32_t)i, logn);
    t->rp[k] = p;
    t->irp[k] = rso_invmod(p, q);
    p = rso_mulmod(p, t->psi, q);
  }
  t->ninv = rso_invmod((uint64_t)t->n % q, q);
  return t;
}
void rso_ntt_destroy(rso_ntt *t) {
  if (!t) return;
  free(t->rp);
  free(t->irp);
  free(t);
}

/* SEAL ntt_negacyclic_harvey (Cooley-Tukey, natural in -> bit-reversed out); the lazy [0,4q)
 * bookkeeping of the original is an implementation detail, final values are canonical. */
void rso_ntt_fwd(const rso_ntt *t, uint64_t *a) {
  const uint64_t q = t->q;
  size_t n = t->n;
  for (size_t m = 1, gap = n >> 1; m < n; m <<= 1, gap >>= 1) {
    for (size_t i = 0; i < m; i++) {
      uint64_t W = t->rp[m + i];
      size_t j1 = 2 * i * gap, j2 = j1 + gap;
      for (size_t j = j1; j < j2; j++) {
        uint64_t u = a[j], v = rso_mulmod(a[j + gap], W, q);
        a[j] = addmod(u, v, q);
        a[j + gap] = submod(u, v, q);
      }
    }
  }
}
/* SEAL inverse_ntt_negacyclic_harvey (Gentleman-Sande, bit-reversed in -> natural out, * n^-1). */
void rso_ntt_inv(const rso_ntt *t, uint64_t *a) {
  const uint64_t q = t->q;
  size_t n = t->n;
  for (size_t m = n >> 1, gap = 1; m >= 1; m >>= 1, gap <<= 1) {
    for (size_t i = 0; i < m; i++) {
      uint64_t W = t->irp[m + i];
      size_t j1 = 2 * i * gap, j2 = j1 + gap;
      for (size_t j = j1; j < j2; j++) {
        uint64_t u = a[j], v = a[j + gap];
        a[j] = addmod(u, v, q);
        a[j + gap] = rso_mulmod(submod(u, v, q), W, q);
      }
    }
  }
  for (size_t j = 0; j < n; j++) a[j] = rso_mulmod(a[j], t->ninv, q);
}

/* ------------------------------------------------------------------------------------------
 * context (seal/seal_ring.hpp:52-58, 266-320): ring Z_q[X]/(X^N+1), q = prod q_i; one BGV
 * encoding context per ring limb with plain modulus q_i, degree N_enc, data primes Q_j.
 * ---------------------------------------------------------------------------------------- */
static int ilog2(size_t x) {
  int l = 0;
  while (((size_t)1 << l) < x) l++;
  return l;
}
rso_ctx *rso_ctx_create(int N, int L, const uint64_t *q, int N_enc, int K, const uint64_t *Q) {
  if (L > RSO_MAXL || K > RSO_MAXK || N > N_enc) return NULL;
  rso_ctx *c = (rso_ctx *)calloc(1, sizeof(rso_ctx));
  c->N = N;
  c->L = L;
  c->N_enc = N_enc;
  c->K = K;
  c->logN_enc = ilog2((size_t)N_enc);
  for (int i = 0; i < L; i++) {
    c->q[i] = q[i];
    c->plain[i] = rso_ntt_create(c->logN_enc, q[i]);
    if (!c->plain[i]) return NULL;
  }
  for (int j = 0; j < K; j++) {
    c->Q[j] = Q[j];
    c->coeff[j] = rso_ntt_create(c->logN_enc, Q[j]);
    if (!c->coeff[j]) return NULL;
  }
  /* SEAL BatchEncoder::populate_matrix_reps_index_map (batchencoder.cpp): gen = 3, m = 2n. */
  c->index_map = (uint32_t *)malloc(sizeof(uint32_t) * N_enc);
  uint64_t gen = 3, pos = 1, mm = (uint64_t)N_enc << 1;
  size_t row = (size_t)N_enc >> 1;
  for (size_t i = 0; i < row; i++) {
    uint64_t i1 = (pos - 1) >> 1, i2 = (mm - pos - 1) >> 1;
    c->index_map[i] = bitrev32((uint32_t)i1, c->logN_enc);
    c->index_map[row | i] = bitrev32((uint32_t)i2, c->logN_enc);
    pos = (pos * gen) & (mm - 1);
  }
  return c;
}
void rso_ctx_destroy(rso_ctx *c) {
  if (!c) return;
  for (int i = 0; i < c->L; i++) rso_ntt_destroy(c->plain[i]);
  for (int j = 0; j < c->K; j++) rso_ntt_destroy(c->coeff[j]);
  free(c->index_map);
  free(c);
}
size_t rso_ring_words(const rso_ctx *c) { return (size_t)c->L * c->N; }
size_t rso_ct_words(const rso_ctx *c) { return (size_t)2 * c->K * c->N_enc; }
size_t rso_enc_words(const rso_ctx *c) { return (size_t)c->L * 2 * c->K * c->N_enc; }

/* ------------------------------------------------------------------------------------------
 * RingElem arithmetic: polynomials are always held in NTT form (seal_ring.tcc:270), so the
 * SealPoly *_inplace calls at seal_ring.tcc:66,108,113,160,165,191,200 are dyadic per limb.
 * ---------------------------------------------------------------------------------------- */
void rso_ring_add(const rso_ctx *c, uint64_t *d, const uint64_t *a, const uint64_t *b) {
  for (int i = 0; i < c->L; i++)
    for (int x = 0; x < c->N; x++) d[i * c->N + x] = addmod(a[i * c->N + x], b[i * c->N + x], c->q[i]);
}
void rso_ring_sub(const rso_ctx *c, uint64_t *d, const uint64_t *a, const uint64_t *b) {
  for (int i = 0; i < c->L; i++)
    for (int x = 0; x < c->N; x++) d[i * c->N + x] = submod(a[i * c->N + x], b[i * c->N + x], c->q[i]);
}
void rso_ring_mul(const rso_ctx *c, uint64_t *d, const uint64_t *a, const uint64_t *b) {
  for (int i = 0; i < c->L; i++)
    for (int x = 0; x < c->N; x++)
      d[i * c->N + x] = rso_mulmod(a[i * c->N + x], b[i * c->N + x], c->q[i]);
}
void rso_ring_neg(const rso_ctx *c, uint64_t *d, const uint64_t *a) {
  for (int i = 0; i < c->L; i++)
    for (int x = 0; x < c->N; x++) d[i * c->N + x] = negmod(a[i * c->N + x], c->q[i]);
}
void rso_ring_mul_scalar(const rso_ctx *c, uint64_t *d, const uint64_t *a, uint64_t s) {
  for (int i = 0; i < c->L; i++) {
    uint64_t si = s % c->q[i];
    for (int x = 0; x < c->N; x++) d[i * c->N + x] = rso_mulmod(a[i * c->N + x], si, c->q[i]);
  }
}
/* SealPoly::invert_inplace (called at seal_ring.tcc:76-96): slot-wise inverse, false if any
 * slot is zero. */
int rso_ring_inv(const rso_ctx *c, uint64_t *d, const uint64_t *a) {
  for (size_t k = 0; k < rso_ring_words(c); k++)
    if (a[k] == 0) return 0;
  for (int i = 0; i < c->L; i++)
    for (int x = 0; x < c->N; x++) d[i * c->N + x] = rso_invmod(a[i * c->N + x], c->q[i]);
  return 1;
}
int rso_ring_is_zero(const rso_ctx *c, const uint64_t *a) {
  for (size_t k = 0; k < rso_ring_words(c); k++)
    if (a[k]) return 0;
  return 1;
}

/* ------------------------------------------------------------------------------------------
 * SEAL BatchEncoder::encode(values, plain) as used at seal_ring.tcc:352,534: the N limb values
 * go to slots index_map[0..N), remaining slots are zero (the TODO at seal_ring.tcc:350-351),
 * then an in-place inverse negacyclic NTT mod t = q_limb.
 * ---------------------------------------------------------------------------------------- */
void rso_batch_encode(const rso_ctx *c, int limb, const uint64_t *values, uint64_t *plain) {
  memset(plain, 0, sizeof(uint64_t) * c->N_enc);
  for (int i = 0; i < c->N; i++) plain[c->index_map[i]] = values[i];
  rso_ntt_inv(c->plain[limb], plain);
}
/* SEAL BatchEncoder::decode as used at seal_ring.tcc:456 (first N slots kept, :470). */
void rso_batch_decode(const rso_ctx *c, int limb, const uint64_t *plain, uint64_t *values) {
  uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * c->N_enc);
  memcpy(tmp, plain, sizeof(uint64_t) * c->N_enc);
  rso_ntt_fwd(c->plain[limb], tmp);
  for (int i = 0; i < c->N; i++) values[i] = tmp[c->index_map[i]];
  free(tmp);
}

/* centered lift of a mod-t coefficient into Z_Q (SEAL Evaluator::transform_to_ntt_inplace on a
 * Plaintext: coefficients >= plain_upper_half_threshold = (t+1)/2 get plain_upper_half_increment
 * = Q - t added, i.e. the value becomes (c - t) mod Q_j). */
static inline uint64_t lift_centered(uint64_t cf, uint64_t t, uint64_t Q) {
  if (cf >= ((t + 1) >> 1)) {
    uint64_t neg = (t - cf) % Q; /* |c - t| */
    return neg ? Q - neg : 0;
  }
  return cf % Q;
}

/* SEAL Evaluator::multiply_plain_inplace(ct, plain) as used at seal_ring.tcc:536 with an
 * NTT-form BGV ciphertext and a coefficient-form plaintext: lift, K forward NTTs, then a dyadic
 * product into both ciphertext polynomials. */
void rso_multiply_plain(const rso_ctx *c, int limb, uint64_t *ct, const uint64_t *plain) {
  size_t n = c->N_enc;
  uint64_t *P = (uint64_t *)malloc(sizeof(uint64_t) * n);
  for (int j = 0; j < c->K; j++) {
    for (size_t x = 0; x < n; x++) P[x] = lift_centered(plain[x], c->q[limb], c->Q[j]);
    rso_ntt_fwd(c->coeff[j], P);
    for (int comp = 0; comp < 2; comp++) {
      uint64_t *p = ct + ((size_t)comp * c->K + j) * n;
      for (size_t x = 0; x < n; x++) p[x] = rso_mulmod(p[x], P[x], c->Q[j]);
    }
  }
  free(P);
}
/* SEAL Evaluator::add_inplace as used at seal_ring.tcc:494. */
void rso_ct_add(const rso_ctx *c, uint64_t *ct, const uint64_t *o) {
  size_t n = c->N_enc;
  for (int comp = 0; comp < 2; comp++)
    for (int j = 0; j < c->K; j++) {
      size_t off = ((size_t)comp * c->K + j) * n;
      for (size_t x = 0; x < n; x++) ct[off + x] = addmod(ct[off + x], o[off + x], c->Q[j]);
    }
}
/* EncodingElem::operator*=(RingElem) polynomial branch, seal_ring.tcc:530-544. */
void rso_enc_mul_ring(const rso_ctx *c, uint64_t *enc, const uint64_t *ring) {
  uint64_t *plain = (uint64_t *)malloc(sizeof(uint64_t) * c->N_enc);
  for (int i = 0; i < c->L; i++) {
    rso_batch_encode(c, i, ring + (size_t)i * c->N, plain);
    rso_multiply_plain(c, i, enc + (size_t)i * rso_ct_words(c), plain);
  }
  free(plain);
}
/* EncodingElem::operator+= non-empty branch, seal_ring.tcc:489-506. */
void rso_enc_add(const rso_ctx *c, uint64_t *enc, const uint64_t *o) {
  for (int i = 0; i < c->L; i++)
    rso_ct_add(c, enc + (size_t)i * rso_ct_words(c), o + (size_t)i * rso_ct_words(c));
}

/* EncodingElem::inner_product, seal_ring.tcc:361-433.  The mod-switch checkpoints (:385-410,
 * :418-429) are dead code in every shipped binary and are not restated.
 * Terms with b.is_zero() are skipped (:391-396,:416); Scalar 1 passes the ciphertext through
 * unchanged (:525-527); res starts EMPTY (:412) and the first term is assigned (:485-488). */
size_t rso_inner_product(const rso_ctx *c, const uint64_t *encs, const uint64_t *rings,
                         const uint8_t *kinds, size_t T, uint64_t *out) {
  size_t ew = rso_enc_words(c), rw = rso_ring_words(c), used = 0;
  uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * ew);
  memset(out, 0, sizeof(uint64_t) * ew);
  for (size_t t = 0; t < T; t++) {
    int kind = kinds ? kinds[t] : RSO_KIND_POLY;
    const uint64_t *b = rings + t * rw;
    if (kind == RSO_KIND_POLY && rso_ring_is_zero(c, b)) continue;
    memcpy(tmp, encs + t * ew, sizeof(uint64_t) * ew); /* tmp = a[t] * b[t]  (:417) */
    if (kind != RSO_KIND_ONE) rso_enc_mul_ring(c, tmp, b);
    if (used == 0)
      memcpy(out, tmp, sizeof(uint64_t) * ew); /* res += tmp on empty res (:485-488) */
    else
      rso_enc_add(c, out, tmp);
    used++;
  }
  free(tmp);
  return used;
}

/* ---- multi-core forms used by bench.py (all-core cpu_baseline, post-run check) and by the
 * configuration-scale golden vectors.  Same arithmetic as above, OpenMP over terms -- the
 * parallelisation SURVEY.md 8(d) prescribes for the CPU baseline ("all host cores with OpenMP over
 * terms"); the reference itself runs inner_product serially (seal_ring.tcc:415-431). ---- */
#ifdef _OPENMP
#include <omp.h>
#endif
static int pick_threads(int threads) {
#ifdef _OPENMP
  return threads > 0 ? threads : omp_get_max_threads();
#else
  (void)threads;
  return 1;
#endif
}
int rso_max_threads(void) { return pick_threads(0); }

/* inner_product with the terms spread over `threads` threads (per-thread partial sums, added at the
 * end: modular addition is associative, the result is the same canonical element).  window != 0:
 * only `window` encodings are stored and term t uses encs[t % window] (timing samples). */
size_t rso_inner_product_mt(const rso_ctx *c, const uint64_t *encs, size_t window, const uint64_t *rings,
                            const uint8_t *kinds, size_t T, uint64_t *out, int threads) {
  const size_t ew = rso_enc_words(c), rw = rso_ring_words(c);
  const int nt = pick_threads(threads);
  uint64_t *part = (uint64_t *)calloc((size_t)nt * ew, sizeof(uint64_t));
  size_t *used = (size_t *)calloc((size_t)nt, sizeof(size_t));
#pragma omp parallel num_threads(nt)
  {
#ifdef _OPENMP
    const int id = omp_get_thread_num();
#else
    const int id = 0;
#endif
    uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * ew), *acc = part + (size_t)id * ew;
#pragma omp for schedule(static)
    for (long long t = 0; t < (long long)T; t++) {
      const int kind = kinds ? kinds[t] : RSO_KIND_POLY;
      const uint64_t *b = rings + (size_t)t * rw;
      if (kind == RSO_KIND_POLY && rso_ring_is_zero(c, b)) continue;
      memcpy(tmp, encs + (window ? (size_t)t % window : (size_t)t) * ew, sizeof(uint64_t) * ew);
      if (kind != RSO_KIND_ONE) rso_enc_mul_ring(c, tmp, b);
      rso_enc_add(c, acc, tmp);
      used[id]++;
    }
    free(tmp);
  }
  size_t total = 0;
  memset(out, 0, sizeof(uint64_t) * ew);
  for (int k = 0; k < nt; k++) {
    rso_enc_add(c, out, part + (size_t)k * ew);
    total += used[k];
  }
  free(part);
  free(used);
  return total;
}

/* One (limb, component, prime j) slab of an inner product, accumulated into acc[N_enc]:
 *     acc += sum_{t < T} ct[(t0 + t) % window] * NTT_{Q_j}(lift(iNTT_{q_limb}(scatter(rows[t]))))
 * ct: the slab (limb, component, j) of each STORED key element, ct_stride words apart; window =
 * number of stored elements (tiled key; pass window >= t0 + T for an ordinary key); rows: the limb's
 * N values of each coefficient [T][N].  What the device proof's slab must equal, term range by term
 * range (bench.py post-run check). */
void rso_inner_product_slab(const rso_ctx *c, int limb, int j, const uint64_t *ct, size_t ct_stride,
                            size_t window, size_t t0, const uint64_t *rows, size_t T, uint64_t *acc,
                            int threads) {
  const size_t n = (size_t)c->N_enc;
  const int nt = pick_threads(threads);
  const uint64_t Q = c->Q[j], t_mod = c->q[limb];
  uint64_t *part = (uint64_t *)calloc((size_t)nt * n, sizeof(uint64_t));
#pragma omp parallel num_threads(nt)
  {
#ifdef _OPENMP
    const int id = omp_get_thread_num();
#else
    const int id = 0;
#endif
    uint64_t *P = (uint64_t *)malloc(sizeof(uint64_t) * n), *a = part + (size_t)id * n;
#pragma omp for schedule(static)
    for (long long t = 0; t < (long long)T; t++) {
      const uint64_t *row = rows + (size_t)t * c->N;
      int nz = 0;
      for (int x = 0; x < c->N && !nz; x++) nz = row[x] != 0;
      if (!nz) continue; /* zero plaintext: contributes nothing */
      rso_batch_encode(c, limb, row, P);
      for (size_t x = 0; x < n; x++) P[x] = lift_centered(P[x], t_mod, Q);
      rso_ntt_fwd(c->coeff[j], P);
      const uint64_t *cw = ct + (((size_t)t0 + (size_t)t) % window) * ct_stride;
      for (size_t x = 0; x < n; x++) a[x] = addmod(a[x], rso_mulmod(cw[x], P[x], Q), Q);
    }
    free(P);
  }
  for (int k = 0; k < nt; k++)
    for (size_t x = 0; x < n; x++) acc[x] = addmod(acc[x], part[(size_t)k * n + x], Q);
  free(part);
}

/* rso_witness_map with the S independent slots spread over threads (each thread runs the
 * reference's O(m^2) map on its own block of slots).  Layouts as rso_witness_map. */
void rso_witness_map_mt(uint64_t q, size_t S, const rso_r1cs *cs, int limb, const uint64_t *assignment,
                        const uint64_t *d1, const uint64_t *d2, const uint64_t *d3, uint64_t *A_io,
                        uint64_t *B_io, uint64_t *C_io, uint64_t *A_mid, uint64_t *B_mid,
                        uint64_t *C_mid, uint64_t *Z, uint64_t *H, int threads) {
  const int nt = pick_threads(threads);
  const size_t m = cs->m, nv = cs->n_vars;
#pragma omp parallel for schedule(static) num_threads(nt)
  for (int k = 0; k < nt; k++) {
    const size_t lo = S * (size_t)k / (size_t)nt, hi = S * (size_t)(k + 1) / (size_t)nt, w = hi - lo;
    if (!w) continue;
    uint64_t *asg = (uint64_t *)malloc(sizeof(uint64_t) * nv * w);
    uint64_t *dd[3] = {NULL, NULL, NULL};
    const uint64_t *ds[3] = {d1, d2, d3};
    for (size_t v = 0; v < nv; v++) memcpy(asg + v * w, assignment + v * S + lo, sizeof(uint64_t) * w);
    for (int e = 0; e < 3; e++)
      if (ds[e]) {
        dd[e] = (uint64_t *)malloc(sizeof(uint64_t) * w);
        memcpy(dd[e], ds[e] + lo, sizeof(uint64_t) * w);
      }
    uint64_t *o[7];
    for (int e = 0; e < 7; e++) o[e] = (uint64_t *)malloc(sizeof(uint64_t) * (m + 1) * w);
    uint64_t *Zl = (uint64_t *)malloc(sizeof(uint64_t) * (m + 1));
    rso_r1cs cs_at = *cs; /* this thread's slots start at ring slot ptab_slot0 + lo */
    cs_at.ptab_slot0 += lo;
    rso_witness_map(q, w, &cs_at, limb, asg, dd[0], dd[1], dd[2], o[0], o[1], o[2], o[3], o[4], o[5], Zl, o[6]);
    uint64_t *dst[7] = {A_io, B_io, C_io, A_mid, B_mid, C_mid, H};
    for (int e = 0; e < 7; e++) {
      const size_t rows = e == 6 ? m + 1 : m;
      if (dst[e])
        for (size_t r = 0; r < rows; r++) memcpy(dst[e] + r * S + lo, o[e] + r * w, sizeof(uint64_t) * w);
      free(o[e]);
    }
    if (k == 0 && Z) memcpy(Z, Zl, sizeof(uint64_t) * (m + 1));
    free(Zl);
    free(asg);
    for (int e = 0; e < 3; e++) free(dd[e]);
  }
}

/* ------------------------------------------------------------------------------------------
 * PRNG + BGV symmetric encryption.  NOT SEAL's sampler (Blake2xb / centred binomial): the
 * prover never inspects ciphertext randomness, so any valid BGV ciphertext exercises the path
 * (SURVEY.md section 8(d) "Synthetic CRS").  Structure follows SEAL util::encrypt_zero_symmetric
 * for scheme_type::bgv: c1 = a, c0 = -(a*s + t*e), then + plain (lifted, NTT form).
 * ---------------------------------------------------------------------------------------- */
uint64_t rso_splitmix64(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
void rso_fill_uniform(uint64_t seed, uint64_t q, size_t n, uint64_t *out) {
  uint64_t s = seed;
  for (size_t i = 0; i < n; i++) out[i] = rso_splitmix64(&s) % q;
}
static void sample_ternary_ntt(const rso_ctx *c, uint64_t *st, uint64_t *dst /* [K][N_enc] */) {
  size_t n = c->N_enc;
  int8_t *tr = (int8_t *)malloc(n);
  for (size_t x = 0; x < n; x++) tr[x] = (int8_t)(rso_splitmix64(st) % 3) - 1;
  for (int j = 0; j < c->K; j++) {
    uint64_t *p = dst + (size_t)j * n;
    for (size_t x = 0; x < n; x++) p[x] = tr[x] < 0 ? c->Q[j] - 1 : (uint64_t)tr[x];
    rso_ntt_fwd(c->coeff[j], p);
  }
  free(tr);
}
void rso_keygen(const rso_ctx *c, uint64_t seed, uint64_t *sk) {
  uint64_t st = seed ^ 0x5EC2E7ull;
  sample_ternary_ntt(c, &st, sk);
}
void rso_encrypt_symmetric(const rso_ctx *c, int limb, const uint64_t *sk, const uint64_t *plain,
                           uint64_t seed, uint64_t *ct) {
  size_t n = c->N_enc;
  uint64_t st = seed;
  uint64_t *e = (uint64_t *)malloc(sizeof(uint64_t) * c->K * n);
  uint64_t *P = (uint64_t *)malloc(sizeof(uint64_t) * n);
  sample_ternary_ntt(c, &st, e);
  for (int j = 0; j < c->K; j++) {
    uint64_t Q = c->Q[j], tq = c->q[limb] % Q;
    uint64_t *c0 = ct + (size_t)j * n, *c1 = ct + ((size_t)c->K + j) * n;
    for (size_t x = 0; x < n; x++) P[x] = lift_centered(plain[x], c->q[limb], Q);
    rso_ntt_fwd(c->coeff[j], P);
    for (size_t x = 0; x < n; x++) {
      uint64_t a = rso_splitmix64(&st) % Q;
      uint64_t as = rso_mulmod(a, sk[(size_t)j * n + x], Q);
      uint64_t te = rso_mulmod(tq, e[(size_t)j * n + x], Q);
      c1[x] = a;
      c0[x] = addmod(negmod(addmod(as, te, Q), Q), P[x], Q);
    }
  }
  free(e);
  free(P);
}
/* SEAL Decryptor::bgv_decrypt: (c0 + c1*s) -> coefficient form -> centred mod Q -> mod t.
 * The CRT composition uses Garner mixed-radix digits to avoid big integers. */
void rso_decrypt(const rso_ctx *c, int limb, const uint64_t *sk, const uint64_t *ct,
                 uint64_t *plain) {
  size_t n = c->N_enc;
  int K = c->K;
  uint64_t t = c->q[limb];
  uint64_t *v = (uint64_t *)malloc(sizeof(uint64_t) * K * n);
  for (int j = 0; j < K; j++) {
    uint64_t Q = c->Q[j];
    for (size_t x = 0; x < n; x++)
      v[(size_t)j * n + x] = addmod(ct[(size_t)j * n + x],
                                    rso_mulmod(ct[((size_t)K + j) * n + x], sk[(size_t)j * n + x], Q), Q);
    rso_ntt_inv(c->coeff[j], v + (size_t)j * n);
  }
  /* mixed-radix digits of floor(Q/2): digits of Q-1 are (Q_k - 1); halve from the top. */
  uint64_t half[RSO_MAXK];
  {
    uint64_t carry = 0;
    for (int k = K - 1; k >= 0; k--) {
      u128 cur = (u128)(c->Q[k] - 1) + (u128)carry * c->Q[k];
      half[k] = (uint64_t)(cur >> 1);
      carry = (uint64_t)(cur & 1);
    }
  }
  uint64_t Qmodt = 1 % t;
  for (int k = 0; k < K; k++) Qmodt = rso_mulmod(Qmodt, c->Q[k] % t, t);
  uint64_t prod_inv[RSO_MAXK]; /* (prod_{i<k} Q_i)^{-1} mod Q_k */
  for (int k = 1; k < K; k++) {
    uint64_t prod = 1;
    for (int i = 0; i < k; i++) prod = rso_mulmod(prod, c->Q[i] % c->Q[k], c->Q[k]);
    prod_inv[k] = rso_invmod(prod, c->Q[k]);
  }
  for (size_t x = 0; x < n; x++) {
    uint64_t d[RSO_MAXK];
    d[0] = v[x];
    for (int k = 1; k < K; k++) { /* Garner: value = d0 + Q0*(d1 + Q1*(d2 + ...)) */
      uint64_t Qk = c->Q[k], acc = 0;
      for (int i = k - 1; i >= 0; i--) acc = addmod(rso_mulmod(acc, c->Q[i] % Qk, Qk), d[i] % Qk, Qk);
      d[k] = rso_mulmod(submod(v[(size_t)k * n + x], acc, Qk), prod_inv[k], Qk);
    }
    int upper = 0; /* value > floor(Q/2) ? */
    for (int k = K - 1; k >= 0; k--) {
      if (d[k] != half[k]) {
        upper = d[k] > half[k];
        break;
      }
    }
    uint64_t r = 0;
    for (int k = K - 1; k >= 0; k--) r = addmod(rso_mulmod(r, c->Q[k] % t, t), d[k] % t, t);
    plain[x] = upper ? submod(r, Qmodt, t) : r;
  }
  free(v);
}
/* SEAL 4.x Decryptor::invariant_noise_budget for scheme_type::bgv (un-vendored dependency; its published algorithm --
 * PARITY UNPINNED like the rest of rows a4-a9): noise polynomial = c0 + c1 s mod Q in coefficient form (for BGV that is
 * m + t e; the multiplication by the plain modulus is the BFV branch), CRT-composed, infinity norm over the centred
 * representatives, budget = max(0, bit_count(Q) - significant_bits(norm) - 1).  The composition here is an explicit
 * multi-word integer (K <= 12 words), not the digit comparison the device library uses. */
#define RSO_BIGW (RSO_MAXK + 2)
static void big_muladd(uint64_t *w, uint64_t m, uint64_t a) {
  u128 carry = a;
  for (int i = 0; i < RSO_BIGW; i++) {
    u128 cur = (u128)w[i] * m + carry;
    w[i] = (uint64_t)cur;
    carry = cur >> 64;
  }
}
static int big_cmp(const uint64_t *a, const uint64_t *b) {
  for (int i = RSO_BIGW - 1; i >= 0; i--)
    if (a[i] != b[i]) return a[i] > b[i] ? 1 : -1;
  return 0;
}
static void big_sub(uint64_t *d, const uint64_t *a, const uint64_t *b) { /* d = a - b, a >= b */
  uint64_t borrow = 0;
  for (int i = 0; i < RSO_BIGW; i++) {
    u128 cur = (u128)a[i] - b[i] - borrow;
    d[i] = (uint64_t)cur;
    borrow = (uint64_t)(cur >> 64) & 1;
  }
}
static int big_bits(const uint64_t *w) {
  for (int i = RSO_BIGW - 1; i >= 0; i--)
    if (w[i]) return 64 * i + 64 - __builtin_clzll(w[i]);
  return 0;
}
int rso_noise_budget(const rso_ctx *c, int limb, const uint64_t *sk, const uint64_t *ct) {
  (void)limb; /* the budget does not depend on the plain modulus for BGV */
  size_t n = c->N_enc;
  int K = c->K;
  uint64_t *v = (uint64_t *)malloc(sizeof(uint64_t) * K * n);
  for (int j = 0; j < K; j++) {
    uint64_t Q = c->Q[j];
    for (size_t x = 0; x < n; x++)
      v[(size_t)j * n + x] = addmod(ct[(size_t)j * n + x], rso_mulmod(ct[((size_t)K + j) * n + x], sk[(size_t)j * n + x], Q), Q);
    rso_ntt_inv(c->coeff[j], v + (size_t)j * n);
  }
  uint64_t Qbig[RSO_BIGW] = {1}, half[RSO_BIGW], norm[RSO_BIGW] = {0};
  for (int k = 0; k < K; k++) big_muladd(Qbig, c->Q[k], 0);
  { /* half = floor(Q / 2) */
    uint64_t carry = 0;
    for (int i = RSO_BIGW - 1; i >= 0; i--) {
      half[i] = (Qbig[i] >> 1) | (carry << 63);
      carry = Qbig[i] & 1;
    }
  }
  uint64_t prod_inv[RSO_MAXK];
  for (int k = 1; k < K; k++) {
    uint64_t prod = 1;
    for (int i = 0; i < k; i++) prod = rso_mulmod(prod, c->Q[i] % c->Q[k], c->Q[k]);
    prod_inv[k] = rso_invmod(prod, c->Q[k]);
  }
  for (size_t x = 0; x < n; x++) {
    uint64_t d[RSO_MAXK];
    d[0] = v[x];
    for (int k = 1; k < K; k++) { /* Garner digits, as rso_decrypt */
      uint64_t Qk = c->Q[k], acc = 0;
      for (int i = k - 1; i >= 0; i--) acc = addmod(rso_mulmod(acc, c->Q[i] % Qk, Qk), d[i] % Qk, Qk);
      d[k] = rso_mulmod(submod(v[(size_t)k * n + x], acc, Qk), prod_inv[k], Qk);
    }
    uint64_t val[RSO_BIGW] = {0}, mag[RSO_BIGW];
    for (int k = K - 1; k >= 0; k--) big_muladd(val, c->Q[k], d[k]); /* val = d0 + Q0 (d1 + Q1 (...)) */
    if (big_cmp(val, half) > 0)
      big_sub(mag, Qbig, val);
    else
      memcpy(mag, val, sizeof(mag));
    if (big_cmp(mag, norm) > 0) memcpy(norm, mag, sizeof(norm));
  }
  free(v);
  int diff = big_bits(Qbig) - big_bits(norm) - 1;
  return diff > 0 ? diff : 0;
}
/* EncodingElem::decode WITH the guard of seal_ring.tcc:443-454: returns -1 when every ciphertext has budget left (ring
 * = the decoding), else the index i of the first "ciphertext #i has remaining noise budget 0 <= 0" (decoding_error). */
int rso_enc_decode_checked(const rso_ctx *c, const uint64_t *sk, const uint64_t *enc, uint64_t *ring) {
  for (int i = 0; i < c->L; i++)
    if (rso_noise_budget(c, i, sk, enc + (size_t)i * rso_ct_words(c)) <= 0) return i;
  rso_enc_decode(c, sk, enc, ring);
  return -1;
}
/* EncodingElem::encode for one element, seal_ring.tcc:349-356. */
void rso_enc_encode(const rso_ctx *c, const uint64_t *sk, const uint64_t *ring, uint64_t seed,
                    uint64_t *enc) {
  uint64_t *plain = (uint64_t *)malloc(sizeof(uint64_t) * c->N_enc);
  for (int i = 0; i < c->L; i++) {
    rso_batch_encode(c, i, ring + (size_t)i * c->N, plain);
    rso_encrypt_symmetric(c, i, sk, plain, seed * 1315423911ull + (uint64_t)i + 1,
                          enc + (size_t)i * rso_ct_words(c));
  }
  free(plain);
}
/* EncodingElem::decode, seal_ring.tcc:435-477, without the guard (rso_enc_decode_checked has it). */
void rso_enc_decode(const rso_ctx *c, const uint64_t *sk, const uint64_t *enc, uint64_t *ring) {
  uint64_t *plain = (uint64_t *)malloc(sizeof(uint64_t) * c->N_enc);
  for (int i = 0; i < c->L; i++) {
    rso_decrypt(c, i, sk, enc + (size_t)i * rso_ct_words(c), plain);
    rso_batch_decode(c, i, plain, ring + (size_t)i * c->N);
  }
  free(plain);
}

/* ------------------------------------------------------------------------------------------
 * Generic ring algebra.  Every RingElem operation is slot-wise, so one limb with S slots is S
 * independent problems over F_q; arrays are [n][S].  Domain nodes, s[], phi and b in
 * interpolate() are slot-constant scalars (RingT(i), util/evaluation_domain.tcc:8-13).
 * ---------------------------------------------------------------------------------------- */
/* util/polynomials.tcc:10-43, literally. */
void rso_interpolate_nodes(uint64_t q, size_t S, size_t n, const uint64_t *x, const uint64_t *y,
                           uint64_t *coeffs) {
  uint64_t *s = (uint64_t *)calloc(n, sizeof(uint64_t));
  uint64_t *ff = (uint64_t *)malloc(sizeof(uint64_t) * S);
  memset(coeffs, 0, sizeof(uint64_t) * n * S);
  s[n - 1] = negmod(x[0] % q, q); /* :18 */
  for (size_t i = 1; i < n; i++) { /* :20-25 */
    for (size_t j = n - i - 1; j < n - 1; j++) s[j] = submod(s[j], rso_mulmod(x[i] % q, s[j + 1], q), q);
    s[n - 1] = submod(s[n - 1], x[i] % q, q);
  }
  for (size_t j = 0; j < n; j++) { /* :26-41 */
    uint64_t xj = x[j] % q;
    uint64_t phi = (uint64_t)n % q; /* :27 */
    for (size_t k = n - 1; k > 0; k--)
      phi = addmod(rso_mulmod(phi, xj, q), rso_mulmod(s[k], (uint64_t)k % q, q), q); /* :30-31 */
    uint64_t phi_inv = rso_invmod(phi, q);
    for (size_t v = 0; v < S; v++) ff[v] = rso_mulmod(y[j * S + v], phi_inv, q); /* :33 */
    uint64_t b = 1 % q; /* :34 */
    for (size_t k = n; k-- > 0;) { /* :35-40 */
      uint64_t *ck = coeffs + k * S;
      for (size_t v = 0; v < S; v++) ck[v] = addmod(ck[v], rso_mulmod(b, ff[v], q), q);
      b = addmod(rso_mulmod(b, xj, q), s[k], q);
    }
  }
  free(s);
  free(ff);
}
void rso_interpolate(uint64_t q, size_t S, size_t n, const uint64_t *y, uint64_t *coeffs) {
  uint64_t *x = (uint64_t *)malloc(sizeof(uint64_t) * n);
  for (size_t i = 0; i < n; i++) x[i] = (uint64_t)i;
  rso_interpolate_nodes(q, S, n, x, y, coeffs);
  free(x);
}
/* util/polynomials.tcc:46-53 (Horner). */
void rso_eval(uint64_t q, size_t S, size_t n, const uint64_t *coeffs, uint64_t x, uint64_t *out) {
  x %= q;
  memcpy(out, coeffs + (n - 1) * S, sizeof(uint64_t) * S);
  for (size_t i = n - 1; i-- > 0;)
    for (size_t v = 0; v < S; v++) out[v] = addmod(rso_mulmod(out[v], x, q), coeffs[i * S + v], q);
}
/* util/polynomials.tcc:62-66: Boost polynomial operator*= is the schoolbook product. */
void rso_poly_mul(uint64_t q, size_t S, size_t na, const uint64_t *a, size_t nb, const uint64_t *b,
                  uint64_t *out) {
  memset(out, 0, sizeof(uint64_t) * (na + nb - 1) * S);
  for (size_t i = 0; i < na; i++)
    for (size_t j = 0; j < nb; j++) {
      uint64_t *o = out + (i + j) * S;
      const uint64_t *ai = a + i * S, *bj = b + j * S;
      for (size_t v = 0; v < S; v++) o[v] = addmod(o[v], rso_mulmod(ai[v], bj[v], q), q);
    }
}
static size_t normalised_len(size_t S, size_t n, const uint64_t *p) {
  while (n > 0) {
    int zero = 1;
    for (size_t v = 0; v < S && zero; v++) zero = p[(n - 1) * S + v] == 0;
    if (!zero) break;
    n--;
  }
  return n;
}
/* util/polynomials.tcc:76-81: Boost polynomial operator/= (long division); divisor
 * coefficients are slot-constant scalars here (Z(x), util/evaluation_domain.tcc:81-84). */
size_t rso_poly_div(uint64_t q, size_t S, size_t nn, const uint64_t *num, size_t nd,
                    const uint64_t *den, uint64_t *out) {
  if (nn < nd) return 0;
  size_t nq = nn - nd + 1;
  uint64_t *r = (uint64_t *)malloc(sizeof(uint64_t) * nn * S);
  memcpy(r, num, sizeof(uint64_t) * nn * S);
  uint64_t lead_inv = rso_invmod(den[nd - 1] % q, q);
  for (size_t k = nq; k-- > 0;) {
    uint64_t *ok = out + k * S;
    for (size_t v = 0; v < S; v++) ok[v] = rso_mulmod(r[(k + nd - 1) * S + v], lead_inv, q);
    for (size_t j = 0; j < nd; j++) {
      uint64_t dj = den[j] % q;
      uint64_t *rj = r + (k + j) * S;
      for (size_t v = 0; v < S; v++) rj[v] = submod(rj[v], rso_mulmod(ok[v], dj, q), q);
    }
  }
  free(r);
  return normalised_len(S, nq, out);
}
/* same with a per-slot divisor den[nd][S] (util/division_test.cpp:28-49). */
size_t rso_poly_div_general(uint64_t q, size_t S, size_t nn, const uint64_t *num, size_t nd,
                            const uint64_t *den, uint64_t *out) {
  if (nn < nd) return 0;
  size_t nq = nn - nd + 1;
  uint64_t *r = (uint64_t *)malloc(sizeof(uint64_t) * nn * S);
  memcpy(r, num, sizeof(uint64_t) * nn * S);
  for (size_t k = nq; k-- > 0;) {
    uint64_t *ok = out + k * S;
    for (size_t v = 0; v < S; v++)
      ok[v] = rso_mulmod(r[(k + nd - 1) * S + v], rso_invmod(den[(nd - 1) * S + v], q), q);
    for (size_t j = 0; j < nd; j++) {
      uint64_t *rj = r + (k + j) * S;
      for (size_t v = 0; v < S; v++) rj[v] = submod(rj[v], rso_mulmod(ok[v], den[j * S + v], q), q);
    }
  }
  free(r);
  return normalised_len(S, nq, out);
}
/* util/evaluation_domain.tcc:54-60: Z = prod_{i<m} (x - i) by successive products. */
void rso_vanishing(uint64_t q, size_t m, uint64_t *Z) {
  memset(Z, 0, sizeof(uint64_t) * (m + 1));
  Z[0] = 0;
  Z[1] = 1 % q; /* (x - 0) */
  for (size_t i = 1; i < m; i++) {
    uint64_t ni = negmod((uint64_t)i % q, q);
    for (size_t k = i + 1; k >= 1; k--) Z[k] = addmod(Z[k - 1], rso_mulmod(Z[k], ni, q), q);
    Z[0] = rso_mulmod(Z[0], ni, q);
  }
}

/* relations/variable.tcc:246-254: acc += (index==0 ? one : assignment[index-1]) * coeff.  coeff is a RingT: a
 * slot-constant scalar (cf) or a general ring element (row of ptab: one residue per slot). */
void rso_r1cs_evaluate(uint64_t q, size_t S, const rso_r1cs *cs, int which, int limb,
                       const uint64_t *assignment, uint64_t *out) {
  const uint32_t *rp = cs->row_ptr[which], *col = cs->col[which];
  const uint64_t *cf = cs->coeff[which] + (size_t)limb * cs->nnz[which];
  const int32_t *pidx = cs->pidx[which];
  memset(out, 0, sizeof(uint64_t) * cs->m * S);
  for (size_t i = 0; i < cs->m; i++) {
    uint64_t *o = out + i * S;
    for (uint32_t e = rp[i]; e < rp[i + 1]; e++) {
      const uint64_t *a = col[e] ? assignment + (size_t)(col[e] - 1) * S : NULL;
      if (pidx && pidx[e] >= 0) {
        const uint64_t *pc = cs->ptab + ((size_t)pidx[e] * cs->ptab_L + (size_t)limb) * cs->ptab_N + cs->ptab_slot0;
        for (size_t v = 0; v < S; v++) o[v] = addmod(o[v], a ? rso_mulmod(a[v], pc[v] % q, q) : pc[v] % q, q);
        continue;
      }
      uint64_t cc = cf[e] % q;
      if (!a) {
        for (size_t v = 0; v < S; v++) o[v] = addmod(o[v], cc, q);
      } else {
        for (size_t v = 0; v < S; v++) o[v] = addmod(o[v], rso_mulmod(a[v], cc, q), q);
      }
    }
  }
}

/* reductions/r1cs_to_qrp/r1cs_to_qrp.tcc:149-259 for one limb (S slots). */
void rso_witness_map(uint64_t q, size_t S, const rso_r1cs *cs, int limb, const uint64_t *assignment,
                     const uint64_t *d1, const uint64_t *d2, const uint64_t *d3, uint64_t *A_io,
                     uint64_t *B_io, uint64_t *C_io, uint64_t *A_mid, uint64_t *B_mid,
                     uint64_t *C_mid, uint64_t *Z, uint64_t *H) {
  size_t m = cs->m, nv = cs->n_vars, ni = cs->n_inputs;
  size_t vec = m * S;
  uint64_t *asg = (uint64_t *)calloc(nv * S, sizeof(uint64_t));
  uint64_t *ev = (uint64_t *)malloc(sizeof(uint64_t) * vec);
  uint64_t *aA = (uint64_t *)malloc(sizeof(uint64_t) * vec);
  uint64_t *aB = (uint64_t *)malloc(sizeof(uint64_t) * vec);
  uint64_t *aC = (uint64_t *)malloc(sizeof(uint64_t) * vec);
  uint64_t *outs_mid[3] = {A_mid, B_mid, C_mid}, *outs_io[3] = {A_io, B_io, C_io};
  uint64_t *outs_full[3] = {aA, aB, aC};
  /* :166-187  auxiliary_assignment = zeros(primary) || aux ; interpolate a/b/c_mid */
  memcpy(asg + ni * S, assignment + ni * S, sizeof(uint64_t) * (nv - ni) * S);
  for (int w = 0; w < 3; w++) {
    rso_r1cs_evaluate(q, S, cs, w, limb, asg, ev);
    rso_interpolate(q, S, m, ev, outs_mid[w]);
  }
  /* :189-208  primary_assignment = primary || zeros ; interpolate a/b/c_io */
  memset(asg, 0, sizeof(uint64_t) * nv * S);
  memcpy(asg, assignment, sizeof(uint64_t) * ni * S);
  for (int w = 0; w < 3; w++) {
    rso_r1cs_evaluate(q, S, cs, w, limb, asg, ev);
    rso_interpolate(q, S, m, ev, outs_io[w]);
  }
  /* :211 */
  rso_vanishing(q, m, Z);
  /* :214-223 full assignment */
  for (int w = 0; w < 3; w++) {
    rso_r1cs_evaluate(q, S, cs, w, limb, assignment, ev);
    rso_interpolate(q, S, m, ev, outs_full[w]);
  }
  /* :225-235  H = d2*A + d1*B ; H[0] -= d3 ; H += d1*d2*Z */
  memset(H, 0, sizeof(uint64_t) * (m + 1) * S);
  for (size_t i = 0; i < m; i++)
    for (size_t v = 0; v < S; v++) {
      uint64_t t1 = d2 ? rso_mulmod(d2[v], aA[i * S + v], q) : 0;
      uint64_t t2 = d1 ? rso_mulmod(d1[v], aB[i * S + v], q) : 0;
      H[i * S + v] = addmod(t1, t2, q);
    }
  if (d3)
    for (size_t v = 0; v < S; v++) H[v] = submod(H[v], d3[v], q);
  if (d1 && d2)
    for (size_t i = 0; i <= m; i++)
      for (size_t v = 0; v < S; v++)
        H[i * S + v] = addmod(H[i * S + v], rso_mulmod(rso_mulmod(d1[v], d2[v], q), Z[i], q), q);
  /* :237-245  (A*B - C) / Z */
  uint64_t *prod = (uint64_t *)malloc(sizeof(uint64_t) * (2 * m - 1) * S);
  rso_poly_mul(q, S, m, aA, m, aB, prod);
  for (size_t k = 0; k < vec; k++) prod[k] = submod(prod[k], aC[k], q);
  size_t nn = normalised_len(S, 2 * m - 1, prod);
  if (nn >= m + 1) {
    uint64_t *quo = (uint64_t *)malloc(sizeof(uint64_t) * (nn - m) * S);
    size_t nq = rso_poly_div(q, S, nn, prod, m + 1, Z, quo);
    /* :250-253 */
    for (size_t i = 0; i < nq && i <= m; i++)
      for (size_t v = 0; v < S; v++) H[i * S + v] = addmod(H[i * S + v], quo[i * S + v], q);
    free(quo);
  }
  free(prod);
  free(asg);
  free(ev);
  free(aA);
  free(aB);
  free(aC);
}

/* ------------------------------------------------------------------------------------------
 * Provers.  The witness map runs limb by limb (gather [t][limb][N] -> [t][N]).
 * ---------------------------------------------------------------------------------------- */
typedef struct wit {
  uint64_t *A_io, *B_io, *C_io, *A_mid, *B_mid, *C_mid, *H; /* ring layout [.][L][N] */
  uint64_t *Z;                                              /* [L][m+1] scalars     */
} wit;
static void wit_free(wit *w) {
  free(w->A_io); free(w->B_io); free(w->C_io); free(w->A_mid); free(w->B_mid); free(w->C_mid);
  free(w->H); free(w->Z);
}
static void witness_all_limbs(const rso_ctx *c, const rso_r1cs *cs, const uint64_t *assignment,
                              const uint64_t *d1, const uint64_t *d2, const uint64_t *d3, wit *w) {
  size_t m = cs->m, N = (size_t)c->N, L = (size_t)c->L, rw = L * N;
  uint64_t **dst[7] = {&w->A_io, &w->B_io, &w->C_io, &w->A_mid, &w->B_mid, &w->C_mid, &w->H};
  for (int k = 0; k < 7; k++) *dst[k] = (uint64_t *)calloc((m + 1) * rw, sizeof(uint64_t));
  w->Z = (uint64_t *)malloc(sizeof(uint64_t) * L * (m + 1));
  uint64_t *asg = (uint64_t *)malloc(sizeof(uint64_t) * cs->n_vars * N);
  uint64_t *tmp[7];
  for (int k = 0; k < 7; k++) tmp[k] = (uint64_t *)malloc(sizeof(uint64_t) * (m + 1) * N);
  for (size_t i = 0; i < L; i++) {
    for (size_t v = 0; v < cs->n_vars; v++)
      memcpy(asg + v * N, assignment + v * rw + i * N, sizeof(uint64_t) * N);
    rso_witness_map(c->q[i], N, cs, (int)i, asg, d1 ? d1 + i * N : NULL, d2 ? d2 + i * N : NULL,
                    d3 ? d3 + i * N : NULL, tmp[0], tmp[1], tmp[2], tmp[3], tmp[4], tmp[5],
                    w->Z + i * (m + 1), tmp[6]);
    for (int k = 0; k < 7; k++) {
      size_t rows = (k == 6) ? m + 1 : m;
      for (size_t t = 0; t < rows; t++)
        memcpy(*dst[k] + t * rw + i * N, tmp[k] + t * N, sizeof(uint64_t) * N);
    }
  }
  for (int k = 0; k < 7; k++) free(tmp[k]);
  free(asg);
}
/* res (possibly empty) += other (possibly empty); seal_ring.tcc:479-488. */
static void enc_acc(const rso_ctx *c, uint64_t *res, int *res_empty, const uint64_t *o, int o_empty) {
  if (o_empty) return;
  if (*res_empty) {
    memcpy(res, o, sizeof(uint64_t) * rso_enc_words(c));
    *res_empty = 0;
  } else
    rso_enc_add(c, res, o);
}

/* zk_proof_systems/groth16/groth16.tcc:70-115.  asg_kinds (may be NULL: every wire a polynomial): the representation
 * of each assignment wire, RSO_KIND_ONE for a RingElem holding Scalar 1 -- auxiliary_input goes to inner_product as it
 * is (:108-111), where operator*= passes the ciphertext through unchanged for such a wire (seal_ring.tcc:525-527)
 * instead of multiplying by the batch encoding of all-ones (a different plaintext whenever N_enc > N).  Every other
 * Scalar is flattened by to_poly() there (:529), so its row of `assignment` (all slots = the scalar) is exact. */
void rso_groth16_prove_kinds(const rso_ctx *c, const rso_r1cs *cs, const rso_groth16_pk *pk,
                             const uint64_t *assignment, const uint8_t *asg_kinds, uint64_t *proof, int *empty) {
  size_t m = cs->m, ew = rso_enc_words(c), rw = rso_ring_words(c);
  size_t n_aux = cs->n_vars - cs->n_inputs;
  wit w;
  witness_all_limbs(c, cs, assignment, NULL, NULL, NULL, &w); /* :82-84, d1=d2=d3=0 */
  uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * ew);
  uint64_t *a = proof, *b = proof + ew, *cc = proof + 2 * ew;
  /* :89-95 */
  empty[0] = rso_inner_product(c, pk->s_pows, w.A_io, NULL, m, a) == 0;
  int e = rso_inner_product(c, pk->s_pows, w.A_mid, NULL, m, tmp) == 0;
  enc_acc(c, a, &empty[0], tmp, e);
  enc_acc(c, a, &empty[0], pk->alpha, 0);
  /* :97-103 */
  empty[1] = rso_inner_product(c, pk->s_pows, w.B_io, NULL, m, b) == 0;
  e = rso_inner_product(c, pk->s_pows, w.B_mid, NULL, m, tmp) == 0;
  enc_acc(c, b, &empty[1], tmp, e);
  enc_acc(c, b, &empty[1], pk->beta, 0);
  /* :105-112 */
  empty[2] = rso_inner_product(c, pk->delta_ts, w.H, NULL, m + 1, cc) == 0;
  if (n_aux) {
    e = rso_inner_product(c, pk->delta_mid, assignment + cs->n_inputs * rw, asg_kinds ? asg_kinds + cs->n_inputs : NULL, n_aux, tmp) == 0;
    enc_acc(c, cc, &empty[2], tmp, e);
  }
  free(tmp);
  wit_free(&w);
}
void rso_groth16_prove(const rso_ctx *c, const rso_r1cs *cs, const rso_groth16_pk *pk,
                       const uint64_t *assignment, uint64_t *proof, int *empty) {
  rso_groth16_prove_kinds(c, cs, pk, assignment, NULL, proof, empty);
}

/* zk_proof_systems/rinocchio/rinocchio.tcc:75-190.  asg_kinds: as for rso_groth16_prove_kinds; auxiliary_input is the
 * operand of <beta_prods, aux> (:176-180). */
void rso_rinocchio_prove_kinds(const rso_ctx *c, const rso_r1cs *cs, const rso_rinocchio_pk *pk,
                               const uint64_t *assignment, const uint8_t *asg_kinds, const uint64_t *d1, const uint64_t *d2,
                               const uint64_t *d3, uint64_t *proof, int *empty) {
  size_t m = cs->m, ew = rso_enc_words(c), rw = rso_ring_words(c), N = (size_t)c->N;
  size_t n_aux = cs->n_vars - cs->n_inputs;
  int use_zk = d1 && d2 && d3; /* :81-90 */
  wit w;
  witness_all_limbs(c, cs, assignment, d1, d2, d3, &w);
  /* coefficients_for_Z as ring elements; the leading coefficient is the RingElem Scalar 1 that
   * Boost's product of {-x_i, one} leaves untouched (evaluation_domain.tcc:55-58), so it takes
   * the Scalar-1 fast path of operator*= (seal_ring.tcc:525-527). */
  uint64_t *z = (uint64_t *)malloc(sizeof(uint64_t) * (m + 1) * rw);
  uint8_t *zk = (uint8_t *)calloc(m + 1, 1);
  for (size_t t = 0; t <= m; t++)
    for (size_t i = 0; i < (size_t)c->L; i++)
      for (size_t x = 0; x < N; x++) z[t * rw + i * N + x] = w.Z[i * (m + 1) + t];
  zk[m] = RSO_KIND_ONE;
  uint64_t *z_enc = (uint64_t *)malloc(sizeof(uint64_t) * ew);
  uint64_t *az_enc = (uint64_t *)malloc(sizeof(uint64_t) * ew);
  uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * ew);
  /* :106-163 (the ten OpenMP sections) */
  const uint64_t *mids[3] = {w.A_mid, w.B_mid, w.C_mid};
  for (int k = 0; k < 3; k++) {
    empty[2 * k] = rso_inner_product(c, pk->s_pows, mids[k], NULL, m, proof + (2 * k) * ew) == 0;
    empty[2 * k + 1] =
        rso_inner_product(c, pk->alpha_s_pows, mids[k], NULL, m, proof + (2 * k + 1) * ew) == 0;
  }
  empty[6] = rso_inner_product(c, pk->s_pows, w.H, NULL, m + 1, proof + 6 * ew) == 0;
  empty[7] = rso_inner_product(c, pk->alpha_s_pows, w.H, NULL, m + 1, proof + 7 * ew) == 0;
  rso_inner_product(c, pk->s_pows, z, zk, m + 1, z_enc);
  rso_inner_product(c, pk->alpha_s_pows, z, zk, m + 1, az_enc);
  /* :167-174 */
  if (use_zk) {
    const uint64_t *ds[3] = {d1, d2, d3};
    for (int k = 0; k < 3; k++) {
      memcpy(tmp, z_enc, sizeof(uint64_t) * ew);
      rso_enc_mul_ring(c, tmp, ds[k]);
      enc_acc(c, proof + (2 * k) * ew, &empty[2 * k], tmp, 0);
      memcpy(tmp, az_enc, sizeof(uint64_t) * ew);
      rso_enc_mul_ring(c, tmp, ds[k]);
      enc_acc(c, proof + (2 * k + 1) * ew, &empty[2 * k + 1], tmp, 0);
    }
  }
  /* :176-185 */
  empty[8] = 1;
  memset(proof + 8 * ew, 0, sizeof(uint64_t) * ew);
  if (n_aux) {
    empty[8] =
        rso_inner_product(c, pk->beta_prods, assignment + cs->n_inputs * rw, asg_kinds ? asg_kinds + cs->n_inputs : NULL, n_aux, proof + 8 * ew) == 0;
    if (use_zk) {
      const uint64_t *ds[3] = {d1, d2, d3};
      const uint64_t *bs[3] = {pk->beta_rv_ts, pk->beta_rw_ts, pk->beta_ry_ts};
      for (int k = 0; k < 3; k++) {
        memcpy(tmp, bs[k], sizeof(uint64_t) * ew);
        rso_enc_mul_ring(c, tmp, ds[k]);
        enc_acc(c, proof + 8 * ew, &empty[8], tmp, 0);
      }
    }
  }
  free(tmp);
  free(z);
  free(zk);
  free(z_enc);
  free(az_enc);
  wit_free(&w);
}
void rso_rinocchio_prove(const rso_ctx *c, const rso_r1cs *cs, const rso_rinocchio_pk *pk,
                         const uint64_t *assignment, const uint64_t *d1, const uint64_t *d2,
                         const uint64_t *d3, uint64_t *proof, int *empty) {
  rso_rinocchio_prove_kinds(c, cs, pk, assignment, NULL, d1, d2, d3, proof, empty);
}
