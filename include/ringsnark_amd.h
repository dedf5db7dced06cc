/*
 * ringsnark_amd.h -- C ABI of librs_hip.so: the MI355X (gfx950) implementation of the ringSNARK
 * prover hot path.  Plain pointers and sizes only; every bulk argument
 * named d_* is a DEVICE pointer (hipMalloc / torch tensor storage), every h_* is a HOST pointer.
 *
 * The reference has no FFI: its boundary is the C++ concept <RingT, EncT> the provers are
 * templated on (ringsnark/zk_proof_systems/r1cs_ppzksnark.hpp:173-188).
 * Each entry point below names the reference member/function it replaces; the header-only
 * adapters in include/ringsnark_amd/ring.hpp give these the reference's names and signatures,
 * INTEGRATION.md shows the binding.
 *
 * Layouts (uint64_t canonical residues, little endian; identical to the reference's):
 *   ring element      [L][N]            NTT-slot order, limb-major     (seal/seal_ring.tcc:270)
 *   ciphertext        [2][K][N_enc]     SEAL NTT order
 *   encoding element  [L][2][K][N_enc]  one BGV ciphertext per ring limb (seal/seal_ring.hpp:225)
 *   vectors           [count][...]      element-major ("term-major")
 *
 * All functions return RS_OK (0) or an error code; rs_last_error() gives the message
 * (thread-local).  Entry points are re-entrant with respect to a shared context (the reference
 * calls inner_product from 10 OpenMP sections, rinocchio.tcc:106-163): enqueueing is serialised by a
 * mutex, and the context's workspace buffers carry an event of their last use, which a call on
 * ANOTHER stream waits for on the device -- callers need neither distinct streams nor host
 * synchronisation between calls.  Calls that use workspace therefore do not overlap on the device.
 * A call switches the calling thread's HIP device to the context's and restores it on return.
 * stream is a hipStream_t passed as void* (NULL = default).
 */
#ifndef RINGSNARK_AMD_H
#define RINGSNARK_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RS_OK 0
#define RS_ERR_INVALID 1        /* bad argument / unsupported parameter combination */
#define RS_ERR_HIP 2            /* a HIP runtime call failed */
#define RS_ERR_UNSUPPORTED 3    /* valid in the reference, not built here (message says what) */
#define RS_ERR_NOT_INVERTIBLE 4 /* "element is not invertible in ring" (seal_ring.tcc:93,98) */
#define RS_ERR_NOISE 5          /* decoding_error: "ciphertext #i has remaining noise budget 0 <= 0" (seal_ring.tcc:446-454) */

#define RS_MAX_L 8
#define RS_MAX_K 12

typedef struct rs_ctx rs_ctx;
typedef void *rs_stream;

const char *rs_last_error(void);
int rs_version(void);

/* RingElem::set_context + EncodingElem::set_context(s) (seal/seal_ring.hpp:52-58, 266-320):
 * ring Z_q[X]/(X^N+1), q = prod q[i]; L encoding contexts of degree N_enc with data primes Q[j]
 * and plain modulus q[i].  Requires q[i], Q[j] prime, = 1 mod 2*N_enc, < 2^62, pairwise
 * distinct.  Moduli below 2^50 run on the exact-FP64 arithmetic (the tuned kernels); a context with a modulus in
 * [2^50, 2^62) runs on Montgomery integers (the generic kernels; a ring-side-only big prime keeps the data side on FP64). */
int rs_ctx_create(int device, int N, int L, const uint64_t *q, int N_enc, int K, const uint64_t *Q,
                  rs_ctx **out);
void rs_ctx_destroy(rs_ctx *ctx);

/* device memory helpers for hosts that do not bring their own allocator */
int rs_malloc(rs_ctx *ctx, size_t bytes, void **d_ptr);
int rs_free(rs_ctx *ctx, void *d_ptr);
int rs_upload(rs_ctx *ctx, void *d_dst, const void *h_src, size_t bytes, rs_stream stream);
int rs_download(rs_ctx *ctx, void *h_dst, const void *d_src, size_t bytes, rs_stream stream);
int rs_sync(rs_ctx *ctx, rs_stream stream);

/* ---- a4: negacyclic NTT / inverse NTT over Z_p[X]/(X^N_enc + 1), in place, batched ----------
 * Replaces SealPoly::ntt_inplace / intt_inplace (microbench.cpp:150,155; examples/example_SEAL
 * .cpp:71) and the transforms inside BatchEncoder::encode / Evaluator::multiply_plain_inplace
 * (seal_ring.tcc:534,536).  Forward: natural order in, SEAL (bit-reversed) order out. */
#define RS_MOD_PLAIN 0 /* p = q[index]: plain modulus of encoding context `index` */
#define RS_MOD_COEFF 1 /* p = Q[index] */
int rs_ntt_forward(rs_ctx *ctx, int modset, int index, uint64_t *d_data, size_t batch, rs_stream stream);
int rs_ntt_inverse(rs_ctx *ctx, int modset, int index, uint64_t *d_data, size_t batch, rs_stream stream);

/* ---- a1-a3: RingElem arithmetic, dyadic on [count][L][N] (seal_ring.tcc:62-247) ---------- */
int rs_ring_add(rs_ctx *ctx, uint64_t *d_dst, const uint64_t *d_a, const uint64_t *d_b, size_t count, rs_stream stream);
int rs_ring_sub(rs_ctx *ctx, uint64_t *d_dst, const uint64_t *d_a, const uint64_t *d_b, size_t count, rs_stream stream);
int rs_ring_mul(rs_ctx *ctx, uint64_t *d_dst, const uint64_t *d_a, const uint64_t *d_b, size_t count, rs_stream stream);
int rs_ring_neg(rs_ctx *ctx, uint64_t *d_dst, const uint64_t *d_a, size_t count, rs_stream stream);
/* SealPoly::{add,subtract,multiply}_scalar_inplace: scalar reduced per limb */
int rs_ring_add_scalar(rs_ctx *ctx, uint64_t *d_dst, const uint64_t *d_a, uint64_t scalar, size_t count, rs_stream stream);
int rs_ring_mul_scalar(rs_ctx *ctx, uint64_t *d_dst, const uint64_t *d_a, uint64_t scalar, size_t count, rs_stream stream);
/* SealPoly::invert_inplace: slot-wise inverse; returns RS_ERR_NOT_INVERTIBLE (dst undefined) if
 * any slot of any element is zero.  Synchronises the stream. */
int rs_ring_inv(rs_ctx *ctx, uint64_t *d_dst, const uint64_t *d_a, size_t count, rs_stream stream);
/* RingElem::is_zero per element -> h_flags[count] (1 = zero).  Synchronises the stream. */
int rs_ring_is_zero(rs_ctx *ctx, const uint64_t *d_a, size_t count, uint8_t *h_flags, rs_stream stream);

/* ---- a5-a8: EncodingElem pieces --------------------------------------------------------- */
/* BatchEncoder::encode as used at seal_ring.tcc:352,534: slot scatter + inverse NTT mod q_i.
 * d_plain [count][L][N_enc] coefficient form, canonical. */
int rs_batch_encode(rs_ctx *ctx, const uint64_t *d_rings, uint64_t *d_plain, size_t count, rs_stream stream);
/* EncodingElem::operator*=(RingElem), polynomial operand (seal_ring.tcc:530-544): d_enc[k] *= d_ring[k]. */
int rs_enc_mul_ring(rs_ctx *ctx, uint64_t *d_enc, const uint64_t *d_ring, size_t count, rs_stream stream);
/* EncodingElem::operator+= on non-empty operands (seal_ring.tcc:489-506). */
int rs_enc_add(rs_ctx *ctx, uint64_t *d_dst, const uint64_t *d_a, const uint64_t *d_b, size_t count, rs_stream stream);

/* ---- the two ends of the encoding scheme, either side of the prover ---- */
/* EncodingElem::decode (ringsnark/seal/seal_ring.tcc:435-477; called by the verifiers at
 * groth16.tcc:118-121, rinocchio.tcc:203-214): BGV decryption (c0 + c1*s, coefficient form, centred
 * composition mod Q, reduction mod t = q_i) + BatchEncoder::decode.  d_sk: secret key [K][N_enc] in NTT
 * form; d_enc [count][L][2][K][N_enc]; d_rings [count][L][N].
 * The reference's guard (seal_ring.tcc:446-454) is reproduced: a ciphertext whose invariant noise budget is <= 0 -- the
 * encoding parameters were too small, or the prover spent more budget than they allow -- makes the call return
 * RS_ERR_NOISE with the reference's message ("ciphertext #i has remaining noise budget 0 <= 0"; i = the ring limb, as
 * there; the adapters throw decoding_error).  Every decoding is still written (they are garbage for the ciphertexts
 * that failed), so a caller that wants them regardless can ignore that one status.  Synchronises. */
int rs_enc_decode(rs_ctx *ctx, const uint64_t *d_sk, const uint64_t *d_enc, size_t count, uint64_t *d_rings, rs_stream stream);
/* Decryptor::invariant_noise_budget (SEAL 4.x, scheme bgv; the quantity the guard above tests, seal_ring.tcc:446) of
 * every ciphertext: h_budget[count][L] = max(0, bit_count(Q) - significant_bits(|| c0 + c1 s mod Q ||_inf, centred) - 1).
 * Computed inside the decryption (the centred CRT composition already holds the value).  Synchronises. */
int rs_enc_noise_budget(rs_ctx *ctx, const uint64_t *d_sk, const uint64_t *d_enc, size_t count, int *h_budget, rs_stream stream);
/* EncodingElem::encode (ringsnark/seal/seal_ring.tcc:324-359; called by the generators at
 * groth16.tcc:57-62, rinocchio.tcc:48-61): BatchEncoder::encode + symmetric BGV encryption
 * c1 = a, c0 = -(a*s + t*e) + m.  Element k draws its randomness from the stream `seed + k`
 * (splitmix64, ternary error: the CPU oracle's recipe -- SEAL's Blake2xb/CBD sampler is not
 * restated, so ciphertext BYTES are not SEAL's; the scheme and every decryption are).  Synchronises. */
int rs_enc_encode(rs_ctx *ctx, const uint64_t *d_sk, const uint64_t *d_rings, size_t count, uint64_t seed, uint64_t *d_enc,
                  rs_stream stream);

/* ---- wire format of encoding elements (proofs, key vectors) ------------------
 * The reference declares (de)serialisation of keys and proofs but does not implement it
 * (zk_proof_systems/r1cs_ppzksnark.hpp:43-47,142-146; relations/variable.tcc:391-414 throws).
 * Little endian: "RSNKENC1", u32 N, L, N_enc, K, u64 q[L], Q[K], u64 count, u8 empty[count], pad to
 * 8 bytes, then count elements [L][2][K][N_enc] of canonical u64 residues (the ABI layout, so a
 * SEAL-based verifier can adopt `Ciphertext::data()` limb by limb).  h_empty (may be NULL) marks
 * EMPTY elements (seal_ring.tcc:412,432).  Deserialisation validates magic, dimensions, moduli,
 * sizes and residue ranges; d_enc == NULL only returns the element count. */
size_t rs_enc_wire_size(const rs_ctx *ctx, size_t count);
int rs_enc_serialize(rs_ctx *ctx, const uint64_t *d_enc, const uint8_t *h_empty, size_t count, void *h_buf, size_t buf_bytes,
                     rs_stream stream);
int rs_enc_deserialize(rs_ctx *ctx, const void *h_buf, size_t buf_bytes, uint64_t *d_enc, uint8_t *h_empty, size_t capacity,
                       size_t *h_count, rs_stream stream);

/* Canonicalise integer sums of encoding elements in place (x mod Q_j): the epilogue of the
 * multi-GPU all-reduce of partial inner products; inputs < 2^63. */
int rs_enc_reduce(rs_ctx *ctx, uint64_t *d_enc, size_t count, rs_stream stream);

/* ---- a9: EncodingElem::inner_product (seal_ring.tcc:361-433), the "ring-MSM" ------------- */
#define RS_KIND_POLY 0 /* polynomial operand; an all-zero value contributes nothing (is_zero skip) */
#define RS_KIND_ONE 2  /* RingElem holding Scalar 1: ciphertext passes through (seal_ring.tcc:525-527) */
/* out = sum_t d_encs[t] * d_rings[t].  *h_used = number of non-skipped terms; 0 means the
 * reference returns an EMPTY EncodingElem (d_out is then all zero).  Synchronises only if
 * h_used != NULL. */
int rs_inner_product(rs_ctx *ctx, const uint64_t *d_encs, const uint64_t *d_rings, const uint8_t *h_kinds,
                     size_t T, uint64_t *d_out, size_t *h_used, rs_stream stream);

/* Grouped form used by the provers: several coefficient vectors against several CRS vectors in
 * one pass.  Vectors of one group are summed AFTER the centred lift, which is bit-identical to
 * adding their separate inner products (the ciphertext ring is distributive; DESIGN.md "MSM").
 *   d_out[c][g] = sum_{v in group g} sum_{t < vec[v].T} d_crs[c][t] * vec[v].d_coeff[t]      */
/* ZERO-INITIALISE the struct (`rs_msm_vec v = {0}`): slot_const was added (rs_version() >= 101) where callers of version
 * 100 had tail padding; values other than 0 and 1 are rejected with RS_ERR_INVALID. */
typedef struct rs_msm_vec {
  const uint64_t *d_coeff; /* [T][L][N]; slot_const: [T][L] */
  const uint8_t *h_kinds;  /* host [T] or NULL */
  size_t T;                /* terms (<= crs_len) */
  int group;               /* output group index */
  int slot_const;          /* 1: every ring element of the vector holds ONE value per limb in all its slots and d_coeff is the
                            * compact [T][L] array of those values -- coefficients_for_Z (util/evaluation_domain.tcc:54-60: slot
                            * constant by construction), which Rinocchio multiplies into both key vectors (rinocchio.tcc:150-160).
                            * Same result as the expanded [T][L][N] vector (the plaintext of such an element is value x
                            * encode(1,...,1)); (m+1) ring elements -- 24 GiB on a configs[3] rank -- are never materialised. */
} rs_msm_vec;
/* crs_window: 0, or the number of elements actually stored per CRS vector -- logical element t
 * is then read from index t % crs_window ("tiled" key: how a proving key larger than HBM, e.g. the
 * 384 GiB key of the 2^16-constraint headline, is stood in for on one GPU; every term still streams
 * a distinct-address 2 MiB element from HBM, windows are far larger than any cache).  Must be a
 * multiple of the internal term tile: use a power of two. */
int rs_msm(rs_ctx *ctx, const uint64_t *const *d_crs, int n_crs, size_t crs_len, size_t crs_window,
           const rs_msm_vec *vecs, int n_vecs, int n_groups, uint64_t *d_out /* [n_crs][n_groups] enc elems */,
           size_t *h_used /* [n_vecs] or NULL */, rs_stream stream);

/* rs_msm with the CRS vectors in HOST memory (h_crs[c]: host pointers, pinned memory recommended -- rs_host_alloc): how a
 * proving key larger than HBM is used on one GPU (the 384 GiB key of the 2^16-constraint headline,
 * zk_proof_systems/groth16/groth16.hpp:34-37; SURVEY.md section 7 "host-pinned streaming").  The term tiles are
 * streamed through two device staging buffers, the copy of tile k+1 on its own stream under the kernels of tile k;
 * throughput is bounded by the host link (PCIe), not by HBM.  Same results as rs_msm.  Synchronises. */
int rs_msm_hostkey(rs_ctx *ctx, const uint64_t *const *h_crs, int n_crs, size_t crs_len, size_t crs_window,
                   const rs_msm_vec *vecs, int n_vecs, int n_groups, uint64_t *d_out, size_t *h_used, rs_stream stream);
/* page-locked host memory for such keys (hipHostMalloc / hipHostFree) */
int rs_host_alloc(rs_ctx *ctx, size_t bytes, void **h_ptr);
int rs_host_free(rs_ctx *ctx, void *h_ptr);

/* ---- a14: R1CS in CSR form + linear_combination::evaluate (relations/variable.tcc:246-254) -- */
typedef struct rs_r1cs rs_r1cs;
/* For M in {a,b,c}: h_row_ptr[M][m+1], h_col[M][nnz] (0 = constant one, k>=1 = variable k-1),
 * h_coeff[M][L][nnz] slot-constant residues (what gadgetlib produces). */
int rs_r1cs_create(rs_ctx *ctx, size_t m, size_t n_vars, size_t n_inputs, const uint32_t *const h_row_ptr[3],
                   const uint32_t *const h_col[3], const uint64_t *const h_coeff[3], const size_t nnz[3],
                   rs_r1cs **out);
/* The same with coefficients that are GENERAL ring elements: linear_term<RingT>::coeff is a RingT
 * (relations/variable.hpp), and linear_combination::evaluate multiplies by it whatever it holds
 * (relations/variable.tcc:246-254) -- the DFT constraint of benchmarks/bench_ntt_SEAL.cpp:46-53 multiplies the
 * variables by powers of a polynomial (`row * vars[i]`).  h_poly_idx[M][nnz] (an entry of the outer array may be
 * NULL): -1 = the slot-constant scalar h_coeff[M][.][e] as above, k >= 0 = row k of h_poly_table [n_poly][L][N]
 * (ring layout; h_coeff[M][.][e] is then ignored).  Slot-constant coefficients stay the fast case: the witness
 * map's linear-form io vectors are used whenever no polynomial coefficient multiplies the
 * constant one or a primary input. */
int rs_r1cs_create_poly(rs_ctx *ctx, size_t m, size_t n_vars, size_t n_inputs, const uint32_t *const h_row_ptr[3],
                        const uint32_t *const h_col[3], const uint64_t *const h_coeff[3], const size_t nnz[3],
                        const int32_t *const h_poly_idx[3], const uint64_t *h_poly_table, size_t n_poly, rs_r1cs **out);
void rs_r1cs_destroy(rs_r1cs *cs);
#define RS_EVAL_FULL 0 /* full assignment                      (r1cs_to_qrp.tcc:216-220) */
#define RS_EVAL_IO 1   /* primary || zeros                      (r1cs_to_qrp.tcc:189-201) */
#define RS_EVAL_MID 2  /* zeros || auxiliary                    (r1cs_to_qrp.tcc:166-179) */
int rs_r1cs_evaluate(rs_ctx *ctx, const rs_r1cs *cs, int which /*0=a,1=b,2=c*/, int mode,
                     const uint64_t *d_assignment /* [n_vars][L][N] */, uint64_t *d_out /* [m][L][N] */,
                     rs_stream stream);

/* r1cs_to_qrp_instance_map_with_evaluation (reductions/r1cs_to_qrp/r1cs_to_qrp.tcc:76-116, with
 * evaluate_all_lagrange_polynomials / compute_vanishing_polynomial, util/evaluation_domain.tcc:21-50):
 * what generator and verifier compute from the constraint system and the secret point s
 * (groth16.tcc:7-9,127-128; rinocchio.tcc:7-9,219-220).  d_s [L][N]; outputs d_At, d_Bt, d_Ct
 * [n_vars+1][L][N] (A_k(s) per variable, k = 0 the constant one), d_Ht [m+1][L][N] (powers of s),
 * d_Zt [L][N] (Z(s)).  Division free, so s may coincide with a node in some slots (as in the
 * reference's product loop); returns RS_ERR_NOT_INVERTIBLE with the reference's message ("t cannot be
 * one of the values in the domain") only when s IS a domain element, i.e. equals RingT(j) in every
 * slot of every limb (evaluation_domain.tcc:24-26).  Synchronises. */
int rs_instance_map_eval(rs_ctx *ctx, const rs_r1cs *cs, const uint64_t *d_s, uint64_t *d_At, uint64_t *d_Bt, uint64_t *d_Ct,
                         uint64_t *d_Ht, uint64_t *d_Zt, rs_stream stream);


/* ---- a10-a13: r1cs_to_qrp_witness_map (reductions/r1cs_to_qrp/r1cs_to_qrp.tcc:149-259) ----
 * Outputs in ring layout: A_io..C_mid [m][L][N], H [m+1][L][N]; h_Z [L][m+1] slot-constant
 * scalars (coefficients_for_Z).  d1,d2,d3: ring elements [L][N] or all NULL (zero).  Any output
 * pointer may be NULL to skip it.  Exact quasi-linear algorithm: cyclic transforms of length 2*next_pow2(m) -- complete when
 * q_i has the 2-adicity for them (q_i = 1 mod 4*next_pow2(m)), INCOMPLETE otherwise at multi-pass sizes (m > 2^14: the
 * transform stops at the prime's 2-adicity, up to four stages short, and the pointwise step multiplies residues modulo
 * x^G - eta: csrc/witness_inc.hpp -- the case of the ring primes the reference's own recipe yields, seal_util.hpp:20-32);
 * block convolutions over the largest transform the primes support elsewhere.  Same results on every path.
 * PRECONDITION when C_mid / C_io is requested at multi-pass sizes (M > 2^14, full-length transforms): the assignment
 * SATISFIES the constraint system.  H is then recovered from values on a coset, H = (A B - C) / Z point by point, which
 * equals the reference's quotient (util/polynomials.tcc:76-81 drops the remainder) only when Z divides A B - C; for an
 * unsatisfied assignment that form returns a different polynomial than the long division (which the other shapes, and
 * every call that does not ask for C, still compute: quo(A B - C, Z) = quo(A B, Z) for any assignment).  The reference
 * asserts satisfaction before proving (r1cs_to_qrp.tcc:156); rs_rinocchio_prove inherits the precondition. */
int rs_witness_map(rs_ctx *ctx, const rs_r1cs *cs, const uint64_t *d_assignment, const uint64_t *d_d1,
                   const uint64_t *d_d2, const uint64_t *d_d3, uint64_t *d_A_io, uint64_t *d_B_io,
                   uint64_t *d_C_io, uint64_t *d_A_mid, uint64_t *d_B_mid, uint64_t *d_C_mid, uint64_t *d_H,
                   uint64_t *h_Z, rs_stream stream);
/* The same map restricted to NTT slots [slot0, slot0 + nslots) of every limb (the witness map is
 * slot-parallel: SURVEY.md 8(e), ranks sharing a limb split its slots).  Inputs in the full layout;
 * outputs COMPACT: A_io..C_mid [m][L][nslots], H [m+1][L][nslots].  slot0 and nslots even. */
int rs_witness_map_slots(rs_ctx *ctx, const rs_r1cs *cs, const uint64_t *d_assignment, const uint64_t *d_d1,
                         const uint64_t *d_d2, const uint64_t *d_d3, int slot0, int nslots, uint64_t *d_A_io,
                         uint64_t *d_B_io, uint64_t *d_C_io, uint64_t *d_A_mid, uint64_t *d_B_mid, uint64_t *d_C_mid,
                         uint64_t *d_H, uint64_t *h_Z, rs_stream stream);
/* The same map keeping only a ROW range of every output: h_rows[7][2] = {lo, hi} per vector in the order A_io, B_io,
 * C_io, A_mid, B_mid, C_mid, H; output k holds rows [lo, hi) -- row r at index r - lo, [hi - lo][L][N].  The ranks that
 * share a ring limb each run the whole map of that limb and keep the rows of their TERM range of the inner products
 * (ringsnark_amd/dist.py); full-length outputs (five vectors of 96 GiB for three limbs of the
 * configs[3] shape at 2^18 constraints) would not fit.  The io and mid vectors of one matrix take the same range. */
int rs_witness_map_rows(rs_ctx *ctx, const rs_r1cs *cs, const uint64_t *d_assignment, const uint64_t *d_d1,
                        const uint64_t *d_d2, const uint64_t *d_d3, const size_t *h_rows, uint64_t *d_A_io, uint64_t *d_B_io,
                        uint64_t *d_C_io, uint64_t *d_A_mid, uint64_t *d_B_mid, uint64_t *d_C_mid, uint64_t *d_H, uint64_t *h_Z,
                        rs_stream stream);
/* util/polynomials.tcc:10-43 on the domain {0..n-1}: d_y, d_out [n][L][N] (may alias). */
int rs_interpolate(rs_ctx *ctx, const uint64_t *d_y, uint64_t *d_out, size_t n, rs_stream stream);

/* ---- a11: multiply / add / divide of polynomials with ring-element coefficients
 * (util/polynomials.tcc:62-81, Boost.Math polynomial in the reference).  Vectors [n][L][N], coefficient k in row k;
 * slot-wise polynomial arithmetic over the fields F_{q_i}, schoolbook as in the reference.  Operands and results are
 * normalised the way Boost's polynomial is: *h_len (may be NULL) = length after stripping trailing coefficients equal
 * to RingT(0); rows at or beyond *h_len of the output are zero.  Synchronise.
 *   multiply: d_out [na+nb-1]     add: d_out [max(na,nb)]
 *   divide:   d_quot [nn-nd+1] (nothing to write if nn < nd: the quotient is the zero polynomial, *h_len = 0); the
 *             divisor's leading coefficient must be a unit, else RS_ERR_NOT_INVERTIBLE ("element is not invertible
 *             in ring", what RingElem::operator/ throws inside Boost's division).  If the denominator has zero
 *             leading coefficients that the numerator does not match, the quotient is longer than nn-nd+1 rows:
 *             RS_ERR_INVALID -- pass the denominator's normalised length (ring.hpp's divide does). */
int rs_poly_multiply(rs_ctx *ctx, const uint64_t *d_a, size_t na, const uint64_t *d_b, size_t nb, uint64_t *d_out, size_t *h_len,
                     rs_stream stream);
int rs_poly_add(rs_ctx *ctx, const uint64_t *d_a, size_t na, const uint64_t *d_b, size_t nb, uint64_t *d_out, size_t *h_len,
                rs_stream stream);
int rs_poly_divide(rs_ctx *ctx, const uint64_t *d_num, size_t nn, const uint64_t *d_den, size_t nd, uint64_t *d_quot, size_t *h_len,
                   rs_stream stream);

/* ---- a15 / a16: provers -------------------------------------------------------------------
 * groth16::prover (zk_proof_systems/groth16/groth16.tcc:70-115).  pk vectors are device
 * resident; proof = {A,B,C} [3] encoding elements; h_empty[k] = 1 if the reference would leave
 * element k EMPTY. */
typedef struct rs_groth16_pk {
  const uint64_t *d_s_pows;    /* [m+1] */
  const uint64_t *d_delta_ts;  /* [m+1] */
  const uint64_t *d_delta_mid; /* [n_aux] */
  const uint64_t *d_alpha, *d_beta;
  size_t window; /* 0 = vectors stored in full; else see crs_window of rs_msm */
  int host_key;  /* 1: d_s_pows, d_delta_ts, d_delta_mid are HOST pointers (see rs_msm_hostkey); alpha / beta stay on the device */
} rs_groth16_pk;
int rs_groth16_prove(rs_ctx *ctx, const rs_r1cs *cs, const rs_groth16_pk *pk, const uint64_t *d_assignment,
                     uint64_t *d_proof, int *h_empty /* [3] or NULL */, rs_stream stream);
/* The same with the REPRESENTATION of every assignment wire: h_assignment_kinds [n_vars] (RS_KIND_*), NULL = all
 * polynomials.  The reference hands auxiliary_input to inner_product as it is (groth16.tcc:108-111), and
 * EncodingElem::operator*= passes the ciphertext through unchanged for a RingElem holding Scalar 1
 * (seal/seal_ring.tcc:525-527) -- which is NOT the product with the batch encoding of all-ones when N_enc > N, so the
 * proof bytes depend on it (decryptions do not).  Every other Scalar is flattened by to_poly() there (:529): its row of
 * d_assignment (all slots = the scalar) is exact under RS_KIND_POLY, a Scalar 0 is skipped like a zero polynomial. */
int rs_groth16_prove_kinds(rs_ctx *ctx, const rs_r1cs *cs, const rs_groth16_pk *pk, const uint64_t *d_assignment,
                           const uint8_t *h_assignment_kinds, uint64_t *d_proof, int *h_empty, rs_stream stream);

/* rinocchio::prover (zk_proof_systems/rinocchio/rinocchio.tcc:75-190); d1,d2,d3 are the ZK
 * blinding elements the reference samples at :88-90 (all NULL = non-ZK branch).  proof =
 * {A,A',B,B',C,C',D,D',F} [9]. */
typedef struct rs_rinocchio_pk {
  const uint64_t *d_s_pows, *d_alpha_s_pows; /* [m+1] */
  const uint64_t *d_beta_prods;              /* [n_aux] */
  const uint64_t *d_beta_rv_ts, *d_beta_rw_ts, *d_beta_ry_ts;
  size_t window; /* 0 = vectors stored in full; else see crs_window of rs_msm */
  int host_key;  /* 1: d_s_pows, d_alpha_s_pows, d_beta_prods are HOST pointers (see rs_msm_hostkey) */
} rs_rinocchio_pk;
int rs_rinocchio_prove(rs_ctx *ctx, const rs_r1cs *cs, const rs_rinocchio_pk *pk, const uint64_t *d_assignment,
                       const uint64_t *d_d1, const uint64_t *d_d2, const uint64_t *d_d3, uint64_t *d_proof,
                       int *h_empty /* [9] or NULL */, rs_stream stream);
/* with wire representations (see rs_groth16_prove_kinds): auxiliary_input is the operand of <beta_prods, aux>
 * (rinocchio.tcc:176-180). */
int rs_rinocchio_prove_kinds(rs_ctx *ctx, const rs_r1cs *cs, const rs_rinocchio_pk *pk, const uint64_t *d_assignment,
                             const uint8_t *h_assignment_kinds, const uint64_t *d_d1, const uint64_t *d_d2, const uint64_t *d_d3,
                             uint64_t *d_proof, int *h_empty, rs_stream stream);

/* ---- measurement hooks (bench.py): per-phase device time of the last prover call, in ms ---- */
typedef struct rs_timings {
  float evaluate_ms, witness_ms, msm_ms, total_ms;
} rs_timings;
int rs_last_timings(rs_ctx *ctx, rs_timings *out);
int rs_set_profiling(rs_ctx *ctx, int enabled);
/* Measured denominators for the rooflines, on THIS device, now (the reference's micro-benchmark of the same
 * primitives: microbench.cpp:147-205): device-to-device copy bandwidth (GB/s, read + written bytes of a 2 GiB copy with
 * 16-byte accesses; also as a read-only stream and as an in-place update), the v_fma_f64 issue rate (T lane-operations/s), the exact-FP64 modular multiply of f64mod.hpp and the
 * Montgomery product of intmod.hpp on a 60-bit prime (G modular multiplies/s).  About 50 ms.  Synchronises. */
typedef struct rs_peaks {
  double hbm_copy_gbs, fp64_fma_T, fp64_mulmod_G, int_montmul_G;
  double hbm_read_gbs;    /* read-only stream of 2 GiB (the shape of the inner products' key traffic) */
  double hbm_inplace_gbs; /* every word of 2 GiB read and written back in place (read + written bytes; the shape of the workspace passes) */
} rs_peaks;
int rs_measure_peaks(rs_ctx *ctx, rs_peaks *out, rs_stream stream);
/* Per-kernel device time of everything launched on the context since profiling was enabled (or
 * since the last read): HIP events on the launch stream around every launch, summed per kernel,
 * sorted by time.  alg_bytes / fp64_ops: the ALGORITHMIC HBM bytes and FP64 instructions (per lane)
 * of those launches as DESIGN.md section 3 defines them -- the numerators of the rooflines.
 * Returns the number of distinct kernels in *n_out (may exceed capacity); clears the record. */
typedef struct rs_kernel_stat {
  char name[48];
  int launches;
  float total_ms;
  double alg_bytes, fp64_ops;
} rs_kernel_stat;
int rs_profile_read(rs_ctx *ctx, rs_kernel_stat *out, int capacity, int *n_out);
/* process-wide kernel-shape knobs; results are identical for every accepted value:
 *   "ntt_variant" (14: wide kernels of ntt_wide.hpp, default), "ntt_wide_grid", "mac_variant" (5: mac_kernel_v3),
 *   "plain_variant" (1: plain_center_wide_kernel), "witness_lds_logM", "witness_sub_ct" (2: sub_ntt_wide_kernel),
 *   "witness_tree_ct" (2: tree_wide_kernel), "witness_tree_log" (14), "mac_chunk_units", "mac_share_keys" (1: mac_kernel_v4, one plaintext spectrum for two key vectors), "prover_lin_io" (1), "witness_col_budget_mib", "witness_force_bc", "witness_inc" (1: ring primes without a 2M-th root of unity run INCOMPLETE transforms on the multi-pass path, csrc/witness_inc.hpp; 0: block convolutions), "witness_bc2" (1), "msm_host_tile" (1024 terms per staging buffer), "force_int_arith",
 *   "witness_cross_maxr" (6: most stages of one cross pass), "witness_cross_pair" (1: two groups per thread, 16-byte accesses), "witness_sub_log" (12: rooted sub-transforms on 2^12 blocks, sub_ntt_w12_kernel, where the cross pass stays within "witness_sub12_cross" = 4 stages; 13: never), "int_ntt_variant" (1: ntt_io_kernel), "witness_h_coset" (1: H as an inverse coset transform when the call interpolates C).
 * (Variants that alter results -- timing ablations -- exist only as compile-time macros / the separate experiments
 * build, `make -C ringsnark_amd/csrc experiments`; never in the release library.) */
int rs_set_tuning(const char *key, int value);

/* synthetic-workload helpers for the benchmark harness (device-side generators) */
int rs_fill_uniform(rs_ctx *ctx, uint64_t *d_dst, size_t count, int layout /*0 ring,1 enc*/, uint64_t seed, rs_stream stream);
/* chain circuit x_{i+2} = x_i * x_{i+1}: fills d_assignment[2..m+2) from [0..2) */
int rs_chain_assignment(rs_ctx *ctx, uint64_t *d_assignment, size_t m, rs_stream stream);

#ifdef __cplusplus
}
#endif
#endif
