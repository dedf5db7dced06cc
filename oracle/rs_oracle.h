/*
 * rs_oracle.h -- CPU oracle for the ringSNARK prover hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's algorithm for the path named in
 * BASELINE.json.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product (ringsnark_amd/) never does.
 *
 * PARITY STATUS (see DESIGN.md "Oracle"):
 *   - rows a10-a14 (interpolate / multiply / divide / vanishing / witness map / evaluate):
 *     pinned by the reference's own known-answer tests (util/interpolation_test.cpp:29-55,
 *     util/division_test.cpp:28-49) and by oracle/_ref (the reference's relations/ +
 *     gadgetlib/ headers compiled as they lie) for linear_combination::evaluate.
 *   - rows a4-a9 (NTT order, root choice, batching index map, centered lift, ciphertext bytes):
 *     PARITY UNPINNED.  Microsoft SEAL 4.x and SEAL-Polytools are un-vendored, unpinned
 *     submodules (.gitmodules:4-9; README.md:48 "tested with versions 4.0.0 to 4.1.1"); the
 *     functions below restate SEAL's published algorithms (cited inline) and are anchored on
 *     the reference's call sites (seal/seal_ring.tcc:324-548) and on homomorphic
 *     self-consistency (encode -> inner_product -> decode == ring inner product).
 *   - SURVEY 8(f) f2 / f3 (EncodingElem::encode / ::decode, seal_ring.tcc:324-359, 435-477): the
 *     BGV steps follow SEAL's Encryptor / Decryptor structure; the RANDOMNESS does not (splitmix64
 *     stream, ternary error instead of Blake2xb / centred binomial) -- PARITY UNPINNED for the
 *     ciphertext bytes, pinned for every decryption.
 *   - rows a15 / a16 end to end: tests/snark_ref.py restates the generators and verifiers
 *     (groth16.tcc:5-66,117-170; rinocchio.tcc:5-72,192-300); proofs of the oracle prover and of
 *     the HIP prover satisfy the reference's verification equations under a real encrypted key.
 *
 * Layouts (all uint64_t, little endian, canonical residues):
 *   ring element      [L][N]            NTT-slot order         (seal/seal_ring.tcc:270)
 *   ciphertext        [2][K][N_enc]     SEAL NTT order         (seal/seal_ring.hpp:225)
 *   encoding element  [L][2][K][N_enc]  one ciphertext per ring limb
 */
#ifndef RS_ORACLE_H
#define RS_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSO_MAXL 8
#define RSO_MAXK 12

/* ---- modular arithmetic / primes (SEAL util/numth) ---- */
uint64_t rso_mulmod(uint64_t a, uint64_t b, uint64_t q);
uint64_t rso_powmod(uint64_t a, uint64_t e, uint64_t q);
uint64_t rso_invmod(uint64_t a, uint64_t q); /* q prime */
int rso_is_prime(uint64_t n);
/* SEAL util::get_primes(factor, bit_size, count): primes = 1 mod factor, scanning downward. */
int rso_get_primes(uint64_t factor, int bit_size, int count, uint64_t *out);
/* SEAL CoeffModulus::Create(N, bit_sizes): later-found (smaller) prime of a size goes first. */
int rso_coeff_modulus_create(uint64_t factor, const int *bit_sizes, int count, uint64_t *out);
/* SEAL util::try_minimal_primitive_root(degree, q). */
int rso_minimal_primitive_root(uint64_t degree, uint64_t q, uint64_t *root);

/* ---- negacyclic NTT (SEAL util::NTTTables + ntt_negacyclic_harvey) ---- */
typedef struct rso_ntt {
  uint64_t q, psi, ninv;
  int logn;
  size_t n;
  uint64_t *rp;  /* rp[k]  = psi^{bitrev(k,logn)}  */
  uint64_t *irp; /* irp[k] = rp[k]^{-1}           */
} rso_ntt;
rso_ntt *rso_ntt_create(int logn, uint64_t q);
void rso_ntt_destroy(rso_ntt *t);
void rso_ntt_fwd(const rso_ntt *t, uint64_t *a); /* natural in -> bit-reversed out */
void rso_ntt_inv(const rso_ntt *t, uint64_t *a); /* bit-reversed in -> natural out */

/* ---- context: ring (N, q[L]) + encoding contexts (N_enc, Q[K], plain modulus q_i) ---- */
typedef struct rso_ctx {
  int N, L, N_enc, K, logN_enc;
  uint64_t q[RSO_MAXL], Q[RSO_MAXK];
  rso_ntt *plain[RSO_MAXL]; /* size N_enc mod q_i : BatchEncoder of encoding context i */
  rso_ntt *coeff[RSO_MAXK]; /* size N_enc mod Q_j */
  uint32_t *index_map;      /* BatchEncoder matrix_reps_index_map, N_enc entries */
} rso_ctx;
rso_ctx *rso_ctx_create(int N, int L, const uint64_t *q, int N_enc, int K, const uint64_t *Q);
void rso_ctx_destroy(rso_ctx *c);
size_t rso_ring_words(const rso_ctx *c); /* L*N */
size_t rso_ct_words(const rso_ctx *c);   /* 2*K*N_enc */
size_t rso_enc_words(const rso_ctx *c);  /* L*2*K*N_enc */

/* ---- ring element ops: dyadic on [L][N] (SealPoly *_inplace; seal_ring.tcc:62-247) ---- */
void rso_ring_add(const rso_ctx *c, uint64_t *dst, const uint64_t *a, const uint64_t *b);
void rso_ring_sub(const rso_ctx *c, uint64_t *dst, const uint64_t *a, const uint64_t *b);
void rso_ring_mul(const rso_ctx *c, uint64_t *dst, const uint64_t *a, const uint64_t *b);
void rso_ring_neg(const rso_ctx *c, uint64_t *dst, const uint64_t *a);
void rso_ring_mul_scalar(const rso_ctx *c, uint64_t *dst, const uint64_t *a, uint64_t s);
int rso_ring_inv(const rso_ctx *c, uint64_t *dst, const uint64_t *a); /* 0 if not invertible */
int rso_ring_is_zero(const rso_ctx *c, const uint64_t *a);

/* ---- encoding ops (seal_ring.tcc:324-548) ---- */
void rso_batch_encode(const rso_ctx *c, int limb, const uint64_t *values, uint64_t *plain);
void rso_batch_decode(const rso_ctx *c, int limb, const uint64_t *plain, uint64_t *values);
void rso_multiply_plain(const rso_ctx *c, int limb, uint64_t *ct, const uint64_t *plain);
void rso_ct_add(const rso_ctx *c, uint64_t *ct, const uint64_t *other);
/* EncodingElem::operator*=(RingElem) for a polynomial operand (seal_ring.tcc:509-548). */
void rso_enc_mul_ring(const rso_ctx *c, uint64_t *enc, const uint64_t *ring);
void rso_enc_add(const rso_ctx *c, uint64_t *enc, const uint64_t *other);

#define RSO_KIND_POLY 0 /* polynomial operand; all-zero values are skipped (is_zero)        */
#define RSO_KIND_ONE 2  /* RingElem holding Scalar 1: ciphertext passes through unchanged  */
/* EncodingElem::inner_product (seal_ring.tcc:361-433).  kinds may be NULL (all POLY).
 * Returns the number of non-skipped terms; 0 means the reference returns an EMPTY element
 * (out is then all zero). */
size_t rso_inner_product(const rso_ctx *c, const uint64_t *encs, const uint64_t *rings,
                         const uint8_t *kinds, size_t T, uint64_t *out);

/* ---- multi-core forms (bench.py cpu_baseline / post-run check; golden vectors at configuration
 * scale).  threads <= 0: all cores OpenMP reports. ---- */
int rso_max_threads(void);
size_t rso_inner_product_mt(const rso_ctx *c, const uint64_t *encs, size_t window, const uint64_t *rings,
                            const uint8_t *kinds, size_t T, uint64_t *out, int threads);
void rso_inner_product_slab(const rso_ctx *c, int limb, int j, const uint64_t *ct, size_t ct_stride,
                            size_t window, size_t t0, const uint64_t *rows, size_t T, uint64_t *acc,
                            int threads);

/* ---- BGV symmetric encryption restated (for CRS fixtures / homomorphism checks only) ---- */
void rso_keygen(const rso_ctx *c, uint64_t seed, uint64_t *sk /* [K][N_enc] NTT form */);
void rso_encrypt_symmetric(const rso_ctx *c, int limb, const uint64_t *sk, const uint64_t *plain,
                           uint64_t seed, uint64_t *ct);
void rso_decrypt(const rso_ctx *c, int limb, const uint64_t *sk, const uint64_t *ct,
                 uint64_t *plain);
/* EncodingElem::encode / decode for one ring element (seal_ring.tcc:324-359, 435-477). */
void rso_enc_encode(const rso_ctx *c, const uint64_t *sk, const uint64_t *ring, uint64_t seed,
                    uint64_t *enc);
void rso_enc_decode(const rso_ctx *c, const uint64_t *sk, const uint64_t *enc, uint64_t *ring);
/* Decryptor::invariant_noise_budget (SEAL 4.x, bgv) of one ciphertext, and decode with the reference's guard
 * (seal_ring.tcc:443-454): -1 = decoded, i >= 0 = "ciphertext #i has remaining noise budget 0 <= 0". */
int rso_noise_budget(const rso_ctx *c, int limb, const uint64_t *sk, const uint64_t *ct);
int rso_enc_decode_checked(const rso_ctx *c, const uint64_t *sk, const uint64_t *enc, uint64_t *ring);

/* ---- generic ring algebra over one prime q, S independent slots, arrays [n][S] ---- */
/* util/polynomials.tcc:10-43 with x_j = j (util/evaluation_domain.tcc:8-13). */
void rso_interpolate(uint64_t q, size_t S, size_t n, const uint64_t *y, uint64_t *coeffs);
/* general nodes x[n] (scalars) for the reference's known-answer tests */
void rso_interpolate_nodes(uint64_t q, size_t S, size_t n, const uint64_t *x, const uint64_t *y,
                           uint64_t *coeffs);
/* util/polynomials.tcc:46-53 */
void rso_eval(uint64_t q, size_t S, size_t n, const uint64_t *coeffs, uint64_t x, uint64_t *out);
/* util/polynomials.tcc:62-66 (schoolbook), out [na+nb-1][S] */
void rso_poly_mul(uint64_t q, size_t S, size_t na, const uint64_t *a, size_t nb, const uint64_t *b,
                  uint64_t *out);
/* util/polynomials.tcc:76-81: quotient by a slot-constant monic-or-not divisor den[nd] (scalars).
 * out [nn-nd+1][S]; returns normalised length (trailing zero coefficients stripped). */
size_t rso_poly_div(uint64_t q, size_t S, size_t nn, const uint64_t *num, size_t nd,
                    const uint64_t *den, uint64_t *out);
/* general polynomial / polynomial division per slot (division_test.cpp:28-49) */
size_t rso_poly_div_general(uint64_t q, size_t S, size_t nn, const uint64_t *num, size_t nd,
                            const uint64_t *den, uint64_t *out);
/* util/evaluation_domain.tcc:54-60, Z[m+1] scalars */
void rso_vanishing(uint64_t q, size_t m, uint64_t *Z);

/* ---- R1CS in CSR form (relations/variable.tcc:246-254) ---- */
typedef struct rso_r1cs {
  size_t m, n_vars, n_inputs; /* constraints, variables (excl. constant one), primary inputs */
  /* for M in {a,b,c}: row_ptr[m+1], col[nnz] (0 = constant 1, k>=1 = variable k-1),
   * coeff[L][nnz] slot-constant residues */
  const uint32_t *row_ptr[3];
  const uint32_t *col[3];
  const uint64_t *coeff[3];
  size_t nnz[3];
  /* Coefficients that are general ring elements (linear_term<RingT>::coeff is a RingT, relations/variable.hpp; the
   * DFT constraint of benchmarks/bench_ntt_SEAL.cpp:46-53 multiplies variables by powers of a POLYNOMIAL):
   * pidx[M][e] >= 0 selects row pidx of ptab [n_poly][ptab_L][ptab_N] instead of coeff[M][.][e]; pidx[M] == NULL or
   * pidx[M][e] < 0: the slot-constant scalar.  ptab_slot0: the ring slot that slot 0 of the `assignment` / `out`
   * arrays of a call corresponds to (calls on a slot sub-range of a limb). */
  const int32_t *pidx[3];
  const uint64_t *ptab;
  size_t ptab_L, ptab_N, ptab_slot0;
} rso_r1cs;
/* linear_combination::evaluate for every constraint: out[m][S], assignment [n_vars][S]. */
void rso_r1cs_evaluate(uint64_t q, size_t S, const rso_r1cs *cs, int which, int limb,
                       const uint64_t *assignment, uint64_t *out);

/* reductions/r1cs_to_qrp/r1cs_to_qrp.tcc:149-259, one limb, S slots.
 * assignment [n_vars][S]; d1,d2,d3 [S] (NULL = zero).  Outputs: *_io,*_mid [m][S];
 * Z [m+1] scalars; H [m+1][S]. */
void rso_witness_map(uint64_t q, size_t S, const rso_r1cs *cs, int limb, const uint64_t *assignment,
                     const uint64_t *d1, const uint64_t *d2, const uint64_t *d3, uint64_t *A_io,
                     uint64_t *B_io, uint64_t *C_io, uint64_t *A_mid, uint64_t *B_mid,
                     uint64_t *C_mid, uint64_t *Z, uint64_t *H);

void rso_witness_map_mt(uint64_t q, size_t S, const rso_r1cs *cs, int limb, const uint64_t *assignment,
                        const uint64_t *d1, const uint64_t *d2, const uint64_t *d3, uint64_t *A_io,
                        uint64_t *B_io, uint64_t *C_io, uint64_t *A_mid, uint64_t *B_mid,
                        uint64_t *C_mid, uint64_t *Z, uint64_t *H, int threads);

/* ---- rs_identities.c: COMPLETE check of a witness map computed elsewhere (the device), one limb, S slots, by the
 * polynomial identities that define its outputs (r1cs_to_qrp.tcc:149-259) at n_points <= 4 points r >= m shared by all
 * slots: X_io(r) = sum_j evaluate_X(primary || 0)_j L_j(r), X_mid(r) = sum_j evaluate_X(0 || aux)_j L_j(r),
 * H(r) Z(r) = A(r) B(r) - C(r) + Z(r) (d2 A(r) + d1 B(r) - d3 + d1 d2 Z(r)) -- O(m + nnz) per slot and point.
 * assignment: row v at assignment + v * asg_stride, S slots each; vectors: row k at ptr + k * stride[.] (NULL = not
 * checked); H has m + 1 rows; Z [m + 1] scalars (NULL = not checked); d1, d2, d3 [S] (NULL = 0).
 * bad[S] (may be NULL): bit k set = identity k failed (0..2 A/B/C_io, 3..5 A/B/C_mid, 6 H).
 * Returns the number of failing slots; (size_t)-1 on bad arguments (a point below m), (size_t)-2 if Z is wrong. */
typedef struct rso_wm_vectors {
  const uint64_t *A_io, *B_io, *C_io, *A_mid, *B_mid, *C_mid, *H;
  size_t stride[7]; /* words between rows, per vector in the order above */
  const uint64_t *Z;
  int blocked; /* 1: assignment and vectors are laid out [S/32][rows][32] (S a multiple of 32; strides unused): a block of
                  32 slots streams through memory row after row */
} rso_wm_vectors;
size_t rso_witness_identities(uint64_t q, size_t S, const rso_r1cs *cs, int limb, const uint64_t *assignment,
                              size_t asg_stride, const uint64_t *d1, const uint64_t *d2, const uint64_t *d3,
                              const rso_wm_vectors *v, const uint64_t *points, int n_points, uint8_t *bad, int threads);

/* ---- provers (zk_proof_systems/groth16/groth16.tcc:70-115, rinocchio/rinocchio.tcc:75-190).
 * All vectors in ring layout [count][L][N] / encoding layout [count][L][2][K][N_enc].
 * empty[k]=1 marks a proof element the reference leaves EMPTY. */
typedef struct rso_groth16_pk {
  const uint64_t *s_pows;    /* m+1 */
  const uint64_t *delta_ts;  /* m+1 */
  const uint64_t *delta_mid; /* n_aux */
  const uint64_t *alpha, *beta;
} rso_groth16_pk;
void rso_groth16_prove(const rso_ctx *c, const rso_r1cs *cs, const rso_groth16_pk *pk,
                       const uint64_t *assignment /* [n_vars][L][N] */, uint64_t *proof /* [3] */,
                       int *empty /* [3] */);

/* the same with the representation of every assignment wire (asg_kinds [n_vars], NULL = all polynomials): a wire
 * holding Scalar 1 (RSO_KIND_ONE) passes the ciphertext through in <delta_mid, aux> (seal_ring.tcc:525-527) */
void rso_groth16_prove_kinds(const rso_ctx *c, const rso_r1cs *cs, const rso_groth16_pk *pk,
                             const uint64_t *assignment, const uint8_t *asg_kinds, uint64_t *proof, int *empty);

typedef struct rso_rinocchio_pk {
  const uint64_t *s_pows, *alpha_s_pows; /* m+1 each */
  const uint64_t *beta_prods;            /* n_aux */
  const uint64_t *beta_rv_ts, *beta_rw_ts, *beta_ry_ts;
} rso_rinocchio_pk;
/* d1,d2,d3: ring elements [L][N], or all NULL for the non-ZK branch (rinocchio.tcc:81-90). */
void rso_rinocchio_prove(const rso_ctx *c, const rso_r1cs *cs, const rso_rinocchio_pk *pk,
                         const uint64_t *assignment, const uint64_t *d1, const uint64_t *d2,
                         const uint64_t *d3, uint64_t *proof /* [9] */, int *empty /* [9] */);

void rso_rinocchio_prove_kinds(const rso_ctx *c, const rso_r1cs *cs, const rso_rinocchio_pk *pk,
                               const uint64_t *assignment, const uint8_t *asg_kinds, const uint64_t *d1, const uint64_t *d2,
                               const uint64_t *d3, uint64_t *proof /* [9] */, int *empty /* [9] */);

/* ---- rs_fastcpu.c (librs_oracle_fast.so only): the timed CPU-baseline leg -- EncodingElem::inner_product with SEAL's
 * published arithmetic (Harvey lazy NTT with Shoup quotients, Barrett dyadic products).  Same results as above. ---- */
typedef struct rsf_ctx rsf_ctx;
rsf_ctx *rsf_ctx_create(const rso_ctx *base);
void rsf_ctx_destroy(rsf_ctx *c);
void rsf_ntt_fwd(const rsf_ctx *c, int modset /*0 plain q_i, 1 coeff Q_j*/, int index, uint64_t *a);
void rsf_ntt_inv(const rsf_ctx *c, int modset, int index, uint64_t *a);
size_t rsf_inner_product_mt(const rsf_ctx *c, const uint64_t *encs, size_t window, const uint64_t *rings,
                            const uint8_t *kinds, size_t T, uint64_t *out, int threads);

/* deterministic PRNG shared with the test-suite (splitmix64) */
uint64_t rso_splitmix64(uint64_t *state);
void rso_fill_uniform(uint64_t seed, uint64_t q, size_t n, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif
