// encoding.hip -- SURVEY.md section 8(f) rows f2 / f3: the two ends of the encoding scheme that sit
// either side of the prover.
//
//   rs_enc_decode   EncodingElem::decode  (ringsnark/seal/seal_ring.tcc:435-477): BGV decryption
//                   (c0 + c1*s -> coefficient form -> centred mod Q -> mod t) followed by
//                   BatchEncoder::decode (forward NTT mod t + slot gather).  What the verifier runs on
//                   the proof elements (groth16.tcc:118-121, rinocchio.tcc:203-214).
//   rs_enc_encode   EncodingElem::encode  (ringsnark/seal/seal_ring.tcc:324-359): BatchEncoder::encode
//                   + symmetric BGV encryption.  What the generator runs on every CRS element.
//
// Randomness: SEAL's sampler (Blake2xb / SHAKE, centred binomial) is not restated;
// the ciphertexts follow the CPU oracle's recipe (oracle/rs_oracle.c: splitmix64 stream, ternary
// error), which is all the parity tests can pin.  Any valid BGV ciphertext exercises the prover.
#include <algorithm>
#include <cstring>

#include "ntt_core.hpp"
#include "rs_internal.hpp"

namespace rs {

// ---- decode ----------------------------------------------------------------------------------
// v_j = iNTT_{Q_j}(c0 + c1 * s), canonical doubles.  grid (count * L, K)
__global__ void __launch_bounds__(1024)
decrypt_dot_kernel(const uint64_t *__restrict__ enc, const uint64_t *__restrict__ sk, double *__restrict__ V, int K, int logn,
                   const NttTable *__restrict__ coeff_tabs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  const int n = 1 << logn;
  const size_t el = blockIdx.x;  // (element, limb)
  const int j = blockIdx.y;
  const NttTable tab = coeff_tabs[j];
  const Mod mod = tab.mod;
  const uint64_t *c0 = enc + (el * 2 * K + j) * (size_t)n, *c1 = c0 + (size_t)K * n;
  const uint64_t *sj = sk + (size_t)j * n;
  for (int x = threadIdx.x; x < n; x += blockDim.x)
    s[pidx(x)] = from_u64(c0[x]) + mulmod(from_u64(c1[x]), from_u64(sj[x]), mod);
  __syncthreads();
  lds_ntt_inv(s, logn, tab.d_itw, 1, mod, tab.inv_red_mask);
  double *dst = V + (el * K + j) * (size_t)n;
  for (int x = threadIdx.x; x < n; x += blockDim.x) dst[x] = canon(mulmod(reduce(s[pidx(x)], mod), tab.ninv, mod), mod);
}

// Constants of the CRT composition (SEAL Decryptor::bgv_decrypt composes to the centred
// representative mod Q before reducing mod t): Garner mixed-radix digits, no big integers.
struct CrtConsts {
  int K;
  Mod Qmod[RS_MAX_K];
  double half[RS_MAX_K];                  // mixed-radix digits of floor(Q/2)
  double prod_inv[RS_MAX_K];              // (prod_{i<k} Q_i)^-1 mod Q_k
  double Qi_mod_Qk[RS_MAX_K][RS_MAX_K];   // [i][k] = Q_i mod Q_k
};
struct CrtLimb {
  Mod tmod;
  double Qk_mod_t[RS_MAX_K];
  double Q_mod_t;
};

// ---- the noise guard of EncodingElem::decode (seal_ring.tcc:443-454) ----------------------------------------------
// SEAL 4.x Decryptor::invariant_noise_budget, scheme bgv (un-vendored dependency; its published algorithm): the noise
// polynomial is c0 + c1 s mod Q in coefficient form -- for BGV that IS m + t e, no scaling by the plain modulus (the
// multiplication by t is the BFV branch) -- CRT-composed, its infinity norm taken on the centred representatives, and
//     budget = max(0, bit_count(Q) - significant_bits(norm) - 1).
// The composition already runs here in mixed radix (digits d_k, value = d_0 + Q_0 (d_1 + Q_1 (...))), so the bit length
// of a centred magnitude is found WITHOUT big integers: thr[b][.] holds the digits of 2^b - 1 for b < bit_count(Q)
// (host, exact), |v| >= 2^b is a lexicographic digit comparison, and significant_bits(|v|) = #{b : |v| >= 2^b} by
// binary search (the predicate is monotone).  For v > floor(Q/2) the magnitude is Q - v = (Q - 1 - v) + 1 and
// Q - 1 - v has digits Q_k - 1 - d_k.
template <class T>
__device__ __forceinline__ int magnitude_bits(const T *d, bool upper, const T *Qk, const T *__restrict__ thr, int tb, int K) {
  T w[RS_MAX_K];
  for (int k = 0; k < K; k++) w[k] = upper ? Qk[k] - (T)1 - d[k] : d[k];
  int lo = 0, hi = tb;  // lo = number of b with |v| >= 2^b
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const T *t = thr + (size_t)mid * RS_MAX_K;
    int cmp = 0;
    for (int k = K - 1; k >= 0 && cmp == 0; k--) cmp = w[k] > t[k] ? 1 : (w[k] < t[k] ? -1 : 0);
    // lower half: |v| = w >= 2^b  <=>  w > 2^b - 1;   upper half: |v| = w + 1 >= 2^b  <=>  w >= 2^b - 1
    if (upper ? cmp >= 0 : cmp > 0) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// centred CRT composition mod t, forward NTT mod t, slot gather.  grid (count * L)
__global__ void __launch_bounds__(1024)
crt_decode_kernel(const double *__restrict__ V, uint64_t *__restrict__ rings, int N, int L, int logn, CrtConsts cc,
                  const CrtLimb *__restrict__ limbs, const uint32_t *__restrict__ index_map,
                  const NttTable *__restrict__ plain_tabs, const double *__restrict__ thr, int tb, int *__restrict__ noise_bits) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_bits;
  double *s = reinterpret_cast<double *>(smem);
  const int n = 1 << logn, K = cc.K;
  if (threadIdx.x == 0) s_bits = 0;
  __syncthreads();
  int my_bits = 0;
  double Qk[RS_MAX_K];
  for (int k = 0; k < K; k++) Qk[k] = cc.Qmod[k].p;
  const size_t el = blockIdx.x;
  const int limb = (int)(el % (size_t)L);
  const CrtLimb cl = limbs[limb];
  const NttTable tab = plain_tabs[limb];
  const Mod tmod = cl.tmod;
  const double *v = V + el * (size_t)K * n;
  for (int x = threadIdx.x; x < n; x += blockDim.x) {
    double d[RS_MAX_K];
    d[0] = v[x];
    for (int k = 1; k < K; k++) {  // value = d0 + Q0*(d1 + Q1*(d2 + ...))
      const Mod mk = cc.Qmod[k];
      double acc = 0.0;
      for (int i = k - 1; i >= 0; i--) acc = reduce(mulmod(acc, cc.Qi_mod_Qk[i][k], mk) + reduce(d[i], mk), mk);
      d[k] = canon(mulmod(reduce(v[(size_t)k * n + x] - acc, mk), cc.prod_inv[k], mk), mk);
    }
    bool upper = false;  // value > floor(Q/2)?
    for (int k = K - 1; k >= 0; k--)
      if (d[k] != cc.half[k]) {
        upper = d[k] > cc.half[k];
        break;
      }
    my_bits = max(my_bits, magnitude_bits<double>(d, upper, Qk, thr, tb, K));
    double r = 0.0;
    for (int k = K - 1; k >= 0; k--) r = reduce(mulmod(r, cl.Qk_mod_t[k], tmod) + reduce(d[k], tmod), tmod);
    if (upper) r -= cl.Q_mod_t;
    s[pidx(x)] = reduce(r, tmod);
  }
  atomicMax(&s_bits, my_bits);
  __syncthreads();
  if (threadIdx.x == 0) noise_bits[el] = s_bits;  // significant bits of the infinity norm of this ciphertext's noise polynomial
  if (!rings) return;                              // rs_enc_noise_budget: the budget only
  lds_ntt_fwd(s, logn, tab.d_tw, 1, tmod, tab.fwd_red_mask);  // BatchEncoder::decode
  uint64_t *dst = rings + el * (size_t)N;
  for (int i = threadIdx.x; i < N; i += blockDim.x) dst[i] = to_u64(canon(s[pidx((int)index_map[i])], tmod));
}

// ---- encode ----------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix_at(uint64_t seed, uint64_t k) {  // k-th output (1-based) of the stream
  uint64_t z = seed + k * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// One workgroup per (element, limb, prime j): c1 = a, c0 = -(a*s + t*e) + NTT(lift(BatchEncode(ring limb))).
// Stream layout of oracle/rs_oracle.c rso_encrypt_symmetric: draws 1..n are the ternary error,
// draw n + j*n + x + 1 is a_j[x].  grid (count * L, K); PER = n / blockDim <= 16
__global__ void __launch_bounds__(1024)
encode_kernel(const uint64_t *__restrict__ rings, const uint64_t *__restrict__ sk, uint64_t *__restrict__ enc, uint64_t seed0,
              int N, int L, int K, int logn, const uint32_t *__restrict__ index_map, const NttTable *__restrict__ plain_tabs,
              const NttTable *__restrict__ coeff_tabs, const uint64_t *__restrict__ qint, const uint64_t *__restrict__ Qint) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  const int n = 1 << logn;
  const size_t el = blockIdx.x;
  const size_t elem = el / (size_t)L;
  const int limb = (int)(el % (size_t)L), j = blockIdx.y;
  const NttTable pt = plain_tabs[limb], ct = coeff_tabs[j];
  const Mod tmod = pt.mod, mod = ct.mod;
  const uint64_t t = qint[limb], Q = Qint[j];
  const uint64_t seed = (seed0 + elem) * 1315423911ull + (uint64_t)limb + 1;  // rso_enc_encode's per-limb stream
  // BatchEncoder::encode: slot scatter + inverse NTT mod t
  for (int p = threadIdx.x; p < n; p += blockDim.x) s[pidx(p)] = 0.0;
  __syncthreads();
  const uint64_t *src = rings + el * (size_t)N;
  for (int x = threadIdx.x; x < N; x += blockDim.x) s[pidx((int)index_map[x])] = from_u64(src[x]);
  __syncthreads();
  lds_ntt_inv(s, logn, pt.d_itw, 1, tmod, pt.inv_red_mask);
  // centred lift to Z_{Q_j} (in place; every thread touches only its own positions)
  for (int p = threadIdx.x; p < n; p += blockDim.x) {
    const double c = canon(mulmod(reduce(s[pidx(p)], tmod), pt.ninv, tmod), tmod);
    s[pidx(p)] = reduce(center(c, tmod), mod);
  }
  __syncthreads();
  lds_ntt_fwd(s, logn, ct.d_tw, 1, mod, ct.fwd_red_mask);
  double P[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int p = threadIdx.x + k * blockDim.x;
    if (p < n) P[k] = reduce(s[pidx(p)], mod);
  }
  __syncthreads();
  // ternary error, NTT form
  for (int p = threadIdx.x; p < n; p += blockDim.x) s[pidx(p)] = (double)((int)(splitmix_at(seed, (uint64_t)p + 1) % 3) - 1);
  __syncthreads();
  lds_ntt_fwd(s, logn, ct.d_tw, 1, mod, ct.fwd_red_mask);
  const double tq = reduce(from_u64(t % Q), mod);
  uint64_t *c0 = enc + (el * 2 * K + j) * (size_t)n, *c1 = c0 + (size_t)K * n;
  const uint64_t *sj = sk + (size_t)j * n;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int p = threadIdx.x + k * blockDim.x;
    if (p < n) {
      const uint64_t a = splitmix_at(seed, (uint64_t)n + (uint64_t)j * n + (uint64_t)p + 1) % Q;
      const double as = mulmod(from_u64(a), from_u64(sj[p]), mod);
      const double te = mulmod(tq, reduce(s[pidx(p)], mod), mod);
      c1[p] = a;
      c0[p] = to_u64(canon(P[k] - as - te, mod));
    }
  }
}

// ---- the same three kernels on the integer (Montgomery) arithmetic of intmod.hpp: contexts with a modulus >= 2^50.
// Values are canonical residues; table constants are in Montgomery form (mulmod = data x constant, mulmod_dd = data x data).
__global__ void __launch_bounds__(1024)
decrypt_dot_kernel_int(const uint64_t *__restrict__ enc, const uint64_t *__restrict__ sk, uint64_t *__restrict__ V, int K, int logn,
                       const NttTableI *__restrict__ coeff_tabs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t *s = reinterpret_cast<uint64_t *>(smem);
  const int n = 1 << logn;
  const size_t el = blockIdx.x;
  const int j = blockIdx.y;
  const NttTableI tab = coeff_tabs[j];
  const ModI mod = tab.mod;
  const uint64_t *c0 = enc + (el * 2 * K + j) * (size_t)n, *c1 = c0 + (size_t)K * n;
  const uint64_t *sj = sk + (size_t)j * n;
  for (int x = threadIdx.x; x < n; x += blockDim.x) s[pidx(x)] = addm(c0[x], mulmod_dd(c1[x], sj[x], mod), mod);
  __syncthreads();
  lds_ntt_inv(s, logn, tab.d_itw, 1, mod, 0u);
  uint64_t *dst = V + (el * K + j) * (size_t)n;
  for (int x = threadIdx.x; x < n; x += blockDim.x) dst[x] = mulmod(s[pidx(x)], tab.ninv, mod);
}

struct CrtConstsI {
  int K;
  ModI Qmod[RS_MAX_K];
  uint64_t half[RS_MAX_K];                 // mixed-radix digits of floor(Q/2)
  uint64_t prod_inv[RS_MAX_K];             // (prod_{i<k} Q_i)^-1 mod Q_k, Montgomery form
  uint64_t Qi_mod_Qk[RS_MAX_K][RS_MAX_K];  // [i][k] = Q_i mod Q_k, Montgomery form
};
struct CrtLimbI {
  ModI tmod;
  uint64_t Qk_mod_t[RS_MAX_K];  // Montgomery form
  uint64_t Q_mod_t;             // value
};

__global__ void __launch_bounds__(1024)
crt_decode_kernel_int(const uint64_t *__restrict__ V, uint64_t *__restrict__ rings, int N, int L, int logn, CrtConstsI cc,
                      const CrtLimbI *__restrict__ limbs, const uint32_t *__restrict__ index_map,
                      const NttTableI *__restrict__ plain_tabs, const uint64_t *__restrict__ thr, int tb, int *__restrict__ noise_bits) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_bits;
  uint64_t *s = reinterpret_cast<uint64_t *>(smem);
  const int n = 1 << logn, K = cc.K;
  if (threadIdx.x == 0) s_bits = 0;
  __syncthreads();
  int my_bits = 0;
  uint64_t Qk[RS_MAX_K];
  for (int k = 0; k < K; k++) Qk[k] = cc.Qmod[k].p;
  const size_t el = blockIdx.x;
  const int limb = (int)(el % (size_t)L);
  const CrtLimbI cl = limbs[limb];
  const NttTableI tab = plain_tabs[limb];
  const ModI tmod = cl.tmod;
  const uint64_t *v = V + el * (size_t)K * n;
  for (int x = threadIdx.x; x < n; x += blockDim.x) {
    uint64_t d[RS_MAX_K];
    d[0] = v[x];
    for (int k = 1; k < K; k++) {  // value = d0 + Q0*(d1 + Q1*(d2 + ...))
      const ModI mk = cc.Qmod[k];
      uint64_t acc = 0;
      for (int i = k - 1; i >= 0; i--) acc = addm(mulmod(acc, cc.Qi_mod_Qk[i][k], mk), d[i] % mk.p, mk);
      d[k] = mulmod(subm(v[(size_t)k * n + x], acc, mk), cc.prod_inv[k], mk);
    }
    bool upper = false;  // value > floor(Q/2)?
    for (int k = K - 1; k >= 0; k--)
      if (d[k] != cc.half[k]) {
        upper = d[k] > cc.half[k];
        break;
      }
    my_bits = max(my_bits, magnitude_bits<uint64_t>(d, upper, Qk, thr, tb, K));
    uint64_t r = 0;
    for (int k = K - 1; k >= 0; k--) r = addm(mulmod(r, cl.Qk_mod_t[k], tmod), d[k] % tmod.p, tmod);
    if (upper) r = subm(r, cl.Q_mod_t, tmod);
    s[pidx(x)] = r;
  }
  atomicMax(&s_bits, my_bits);
  __syncthreads();
  if (threadIdx.x == 0) noise_bits[el] = s_bits;
  if (!rings) return;
  lds_ntt_fwd(s, logn, tab.d_tw, 1, tmod, 0u);  // BatchEncoder::decode
  uint64_t *dst = rings + el * (size_t)N;
  for (int i = threadIdx.x; i < N; i += blockDim.x) dst[i] = s[pidx((int)index_map[i])];
}

__global__ void __launch_bounds__(1024)
encode_kernel_int(const uint64_t *__restrict__ rings, const uint64_t *__restrict__ sk, uint64_t *__restrict__ enc, uint64_t seed0,
                  int N, int L, int K, int logn, const uint32_t *__restrict__ index_map, const NttTableI *__restrict__ plain_tabs,
                  const NttTableI *__restrict__ coeff_tabs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t *s = reinterpret_cast<uint64_t *>(smem);
  const int n = 1 << logn;
  const size_t el = blockIdx.x;
  const size_t elem = el / (size_t)L;
  const int limb = (int)(el % (size_t)L), j = blockIdx.y;
  const NttTableI pt = plain_tabs[limb], ct = coeff_tabs[j];
  const ModI tmod = pt.mod, mod = ct.mod;
  const uint64_t t = tmod.p, Q = mod.p;
  const uint64_t seed = (seed0 + elem) * 1315423911ull + (uint64_t)limb + 1;
  for (int p = threadIdx.x; p < n; p += blockDim.x) s[pidx(p)] = 0;
  __syncthreads();
  const uint64_t *src = rings + el * (size_t)N;
  for (int x = threadIdx.x; x < N; x += blockDim.x) s[pidx((int)index_map[x])] = src[x];
  __syncthreads();
  lds_ntt_inv(s, logn, pt.d_itw, 1, tmod, 0u);
  for (int p = threadIdx.x; p < n; p += blockDim.x) {
    const uint64_t c = mulmod(s[pidx(p)], pt.ninv, tmod);
    s[pidx(p)] = lift_residue(lift_centered(c, tmod), mod);
  }
  __syncthreads();
  lds_ntt_fwd(s, logn, ct.d_tw, 1, mod, 0u);
  uint64_t P[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int p = threadIdx.x + k * blockDim.x;
    if (p < n) P[k] = s[pidx(p)];
  }
  __syncthreads();
  for (int p = threadIdx.x; p < n; p += blockDim.x) {
    const int e = (int)(splitmix_at(seed, (uint64_t)p + 1) % 3) - 1;
    s[pidx(p)] = e < 0 ? Q - 1 : (uint64_t)e;
  }
  __syncthreads();
  lds_ntt_fwd(s, logn, ct.d_tw, 1, mod, 0u);
  const uint64_t tq = t % Q;
  uint64_t *c0 = enc + (el * 2 * K + j) * (size_t)n, *c1 = c0 + (size_t)K * n;
  const uint64_t *sj = sk + (size_t)j * n;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int p = threadIdx.x + k * blockDim.x;
    if (p < n) {
      const uint64_t a = splitmix_at(seed, (uint64_t)n + (uint64_t)j * n + (uint64_t)p + 1) % Q;
      const uint64_t as = mulmod_dd(a, sj[p], mod);
      const uint64_t te = mulmod_dd(tq, s[pidx(p)], mod);
      c1[p] = a;
      c0[p] = subm(subm(P[k], as, mod), te, mod);
    }
  }
}

struct TabCopiesI {
  NttTableI *d_plain = nullptr, *d_coeff = nullptr;
  explicit TabCopiesI(rs_ctx *ctx) {
    RS_HIP(hipMalloc(&d_plain, sizeof(NttTableI) * ctx->L));
    RS_HIP(hipMemcpy(d_plain, ctx->plain_i, sizeof(NttTableI) * ctx->L, hipMemcpyHostToDevice));
    RS_HIP(hipMalloc(&d_coeff, sizeof(NttTableI) * ctx->K));
    RS_HIP(hipMemcpy(d_coeff, ctx->coeff_i, sizeof(NttTableI) * ctx->K, hipMemcpyHostToDevice));
  }
  ~TabCopiesI() {
    (void)hipFree(d_plain);
    (void)hipFree(d_coeff);
  }
};

static int enc_threads(int logn) { return (int)std::max(64, std::min(1024, (1 << logn) / 8)); }

// small device copies of the tables (allocated per call: generator / verifier paths are not hot)
struct TabCopies {
  NttTable *d_plain = nullptr, *d_coeff = nullptr;
  uint64_t *d_q = nullptr, *d_Q = nullptr;
  explicit TabCopies(rs_ctx *ctx) {
    RS_HIP(hipMalloc(&d_plain, sizeof(NttTable) * ctx->L));
    RS_HIP(hipMemcpy(d_plain, ctx->plain, sizeof(NttTable) * ctx->L, hipMemcpyHostToDevice));
    RS_HIP(hipMalloc(&d_coeff, sizeof(NttTable) * ctx->K));
    RS_HIP(hipMemcpy(d_coeff, ctx->coeff, sizeof(NttTable) * ctx->K, hipMemcpyHostToDevice));
    RS_HIP(hipMalloc(&d_q, sizeof(uint64_t) * ctx->L));
    RS_HIP(hipMemcpy(d_q, ctx->q, sizeof(uint64_t) * ctx->L, hipMemcpyHostToDevice));
    RS_HIP(hipMalloc(&d_Q, sizeof(uint64_t) * ctx->K));
    RS_HIP(hipMemcpy(d_Q, ctx->Q, sizeof(uint64_t) * ctx->K, hipMemcpyHostToDevice));
  }
  ~TabCopies() {
    (void)hipFree(d_plain);
    (void)hipFree(d_coeff);
    (void)hipFree(d_q);
    (void)hipFree(d_Q);
  }
};

// Host side of the noise guard: bit_count(Q) and the mixed-radix digits (radices Q_0, Q_1, ...) of 2^b - 1 for every
// b < bit_count(Q), by exact multi-word arithmetic (K <= 12 words of 62 bits).
struct BigU {
  uint64_t w[RS_MAX_K + 2];
  BigU() { memset(w, 0, sizeof(w)); }
  void mul_add(uint64_t m, uint64_t a) {  // this = this * m + a
    unsigned __int128 carry = a;
    for (int i = 0; i < RS_MAX_K + 2; i++) {
      const unsigned __int128 cur = (unsigned __int128)w[i] * m + carry;
      w[i] = (uint64_t)cur;
      carry = cur >> 64;
    }
  }
  uint64_t divmod(uint64_t dv) {  // this /= dv, returns the remainder
    unsigned __int128 rem = 0;
    for (int i = RS_MAX_K + 1; i >= 0; i--) {
      const unsigned __int128 cur = (rem << 64) | w[i];
      w[i] = (uint64_t)(cur / dv);
      rem = cur % dv;
    }
    return (uint64_t)rem;
  }
  int bits() const {
    for (int i = RS_MAX_K + 1; i >= 0; i--)
      if (w[i]) return 64 * i + 64 - __builtin_clzll(w[i]);
    return 0;
  }
};

// thr[b * RS_MAX_K + k] = digit k of 2^b - 1; returns bit_count(Q)
static int noise_thresholds(const rs_ctx *ctx, std::vector<uint64_t> &thr) {
  BigU Q;
  Q.w[0] = 1;
  for (int k = 0; k < ctx->K; k++) Q.mul_add(ctx->Q[k], 0);
  const int tb = Q.bits();
  thr.assign((size_t)tb * RS_MAX_K, 0);
  for (int b = 0; b < tb; b++) {
    BigU v;  // 2^b - 1
    for (int i = 0; i < b / 64; i++) v.w[i] = ~0ull;
    if (b % 64) v.w[b / 64] = (1ull << (b % 64)) - 1;
    for (int k = 0; k < ctx->K; k++) thr[(size_t)b * RS_MAX_K + k] = v.divmod(ctx->Q[k]);
  }
  return tb;
}

}  // namespace rs

using namespace rs;

// EncodingElem::decode and Decryptor::invariant_noise_budget share everything up to the centred composition.
// d_rings == nullptr: budgets only.  h_budget [count * L] (may be null).  Returns RS_ERR_NOISE -- after writing every
// decoding -- when d_rings is given and a ciphertext has no budget left (seal_ring.tcc:446-454).
static int decode_impl(rs_ctx *ctx, const uint64_t *d_sk, const uint64_t *d_enc, size_t count, uint64_t *d_rings, int *h_budget,
                       rs_stream stream);

extern "C" {

int rs_enc_decode(rs_ctx *ctx, const uint64_t *d_sk, const uint64_t *d_enc, size_t count, uint64_t *d_rings, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_sk && d_enc && d_rings, "null argument");
  return decode_impl(ctx, d_sk, d_enc, count, d_rings, nullptr, stream);
  RS_API_END
}

int rs_enc_noise_budget(rs_ctx *ctx, const uint64_t *d_sk, const uint64_t *d_enc, size_t count, int *h_budget, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_sk && d_enc && h_budget, "null argument");
  return decode_impl(ctx, d_sk, d_enc, count, nullptr, h_budget, stream);
  RS_API_END
}

}  // extern "C"

// the guard's verdict on the significant-bit counts the kernels found
static int noise_verdict(rs_ctx *ctx, const int *d_bits, size_t count, int tb, bool guard, int *h_budget) {
  std::vector<int> bits(count * (size_t)ctx->L);
  RS_HIP(hipMemcpy(bits.data(), d_bits, bits.size() * sizeof(int), hipMemcpyDeviceToHost));
  long long bad = -1;
  for (size_t e = 0; e < bits.size(); e++) {
    const int budget = std::max(0, tb - bits[e] - 1);  // "The -1 accounts for scaling the invariant noise by 2" (SEAL)
    if (h_budget) h_budget[e] = budget;
    if (budget <= 0 && bad < 0) bad = (long long)e;
  }
  if (guard && bad >= 0) {
    // the reference's message (seal_ring.tcc:450-453); the budget it prints is max(0, .), i.e. 0
    // exactly the reference's text for one element (its decode takes one EncodingElem); a batch call says which element it was
    throw rs::Error(RS_ERR_NOISE, "ciphertext #" + std::to_string((int)(bad % ctx->L)) + " has remaining noise budget 0 <= 0" +
                                      (count > 1 ? " (element " + std::to_string((size_t)(bad / ctx->L)) + " of the batch)" : std::string()));
  }
  return RS_OK;
}

static int decode_impl(rs_ctx *ctx, const uint64_t *d_sk, const uint64_t *d_enc, size_t count, uint64_t *d_rings, int *h_budget,
                       rs_stream stream) {
  if (count == 0) return RS_OK;
  WsScope ws_scope(ctx, S(stream));  // holds ctx->mu: the lazily built constants below are built once
  // constants of the context, built at the first call and kept on the device (round-5 advice: no hipMalloc / hipFree, no
  // pageable upload -- an implicit device-wide synchronisation each -- per verifier decode): the digits of 2^b - 1
  if (!ctx->d_noise_thr) {
    std::vector<uint64_t> thr_h;
    ctx->noise_tb = noise_thresholds(ctx, thr_h);
    if (!ctx->use_int)
      for (auto &x : thr_h) {  // digits < Q_k < 2^50: exact doubles
        const double d = (double)x;
        memcpy(&x, &d, 8);
      }
    void *p = nullptr;
    RS_HIP(hipMalloc(&p, thr_h.size() * 8));
    if (hipMemcpy(p, thr_h.data(), thr_h.size() * 8, hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(p);
      throw rs::Error(RS_ERR_HIP, "upload of the noise-threshold table failed");
    }
    ctx->d_noise_thr = p;
  }
  const int tb = ctx->noise_tb;
  void *d_thr = ctx->d_noise_thr;
  int *d_bits = (int *)ws_get(ctx, 9, sizeof(int) * count * (size_t)ctx->L);
  const int L = ctx->L, K = ctx->K, n = ctx->N_enc;
  hipStream_t st = S(stream);
  if (ctx->use_int) {  // the same composition on the integer arithmetic
    using namespace host;
    CrtConstsI cc;
    memset(&cc, 0, sizeof(cc));
    cc.K = K;
    {
      uint64_t carry = 0;
      for (int k = K - 1; k >= 0; k--) {
        const unsigned __int128 cur = (unsigned __int128)(ctx->Q[k] - 1) + (unsigned __int128)carry * ctx->Q[k];
        cc.half[k] = (uint64_t)(cur >> 1);
        carry = (uint64_t)(cur & 1);
      }
    }
    for (int k = 0; k < K; k++) {
      const uint64_t Qk = ctx->Q[k];
      cc.Qmod[k] = HostArith<ModI>::make(Qk);
      uint64_t prod = 1 % Qk;
      for (int i = 0; i < k; i++) prod = mulmod(prod, ctx->Q[i] % Qk, Qk);
      cc.prod_inv[k] = HostArith<ModI>::konst(k ? invmod(prod, Qk) : 1, Qk);
      for (int i = 0; i < K; i++) cc.Qi_mod_Qk[i][k] = HostArith<ModI>::konst(ctx->Q[i] % Qk, Qk);
    }
    std::vector<CrtLimbI> hl(L);
    for (int i = 0; i < L; i++) {
      const uint64_t t = ctx->q[i];
      memset(&hl[i], 0, sizeof(CrtLimbI));
      hl[i].tmod = HostArith<ModI>::make(t);
      uint64_t Qm = 1 % t;
      for (int k = 0; k < K; k++) {
        hl[i].Qk_mod_t[k] = HostArith<ModI>::konst(ctx->Q[k] % t, t);
        Qm = mulmod(Qm, ctx->Q[k] % t, t);
      }
      hl[i].Q_mod_t = Qm;
    }
    if (!ctx->d_crt_limbs) {
      void *p = nullptr;
      RS_HIP(hipMalloc(&p, sizeof(CrtLimbI) * L));
      if (hipMemcpy(p, hl.data(), sizeof(CrtLimbI) * L, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(p);
        throw rs::Error(RS_ERR_HIP, "upload of the CRT constants failed");
      }
      ctx->d_crt_limbs = p;
    }
    const CrtLimbI *d_limbs = static_cast<const CrtLimbI *>(ctx->d_crt_limbs);
    TabCopiesI tabs(ctx);
    uint64_t *V = (uint64_t *)ws_get(ctx, 14, count * (size_t)L * K * n * sizeof(uint64_t));
    const size_t lds = padded_len((size_t)n) * sizeof(uint64_t);
    const int thr = enc_threads(ctx->logN_enc);
    set_max_dyn_lds((const void *)decrypt_dot_kernel_int, (int)lds);
    set_max_dyn_lds((const void *)crt_decode_kernel_int, (int)lds);
    hipLaunchKernelGGL(decrypt_dot_kernel_int, dim3((unsigned)(count * L), K), dim3(thr), lds, st, d_enc, d_sk, V, K, ctx->logN_enc,
                       tabs.d_coeff);
    hipLaunchKernelGGL(crt_decode_kernel_int, dim3((unsigned)(count * L)), dim3(thr), lds, st, V, d_rings, ctx->N, L, ctx->logN_enc,
                       cc, d_limbs, ctx->d_index_map, tabs.d_plain, (const uint64_t *)d_thr, tb, d_bits);
    RS_HIP(hipGetLastError());
    RS_HIP(hipStreamSynchronize(st));
    return noise_verdict(ctx, d_bits, count, tb, d_rings != nullptr, h_budget);
  }
  // host constants of the CRT composition
  CrtConsts cc;
  memset(&cc, 0, sizeof(cc));
  cc.K = K;
  using namespace host;
  {
    uint64_t carry = 0;
    for (int k = K - 1; k >= 0; k--) {  // digits of Q-1 are (Q_k - 1); halve from the top
      const unsigned __int128 cur = (unsigned __int128)(ctx->Q[k] - 1) + (unsigned __int128)carry * ctx->Q[k];
      cc.half[k] = (double)(uint64_t)(cur >> 1);
      carry = (uint64_t)(cur & 1);
    }
  }
  for (int k = 0; k < K; k++) {
    const uint64_t Qk = ctx->Q[k];
    cc.Qmod[k] = Mod{(double)Qk, 1.0 / (double)Qk};
    uint64_t prod = 1 % Qk;
    for (int i = 0; i < k; i++) prod = mulmod(prod, ctx->Q[i] % Qk, Qk);
    cc.prod_inv[k] = k ? balanced(invmod(prod, Qk), Qk) : 1.0;
    for (int i = 0; i < K; i++) cc.Qi_mod_Qk[i][k] = balanced(ctx->Q[i] % Qk, Qk);
  }
  std::vector<CrtLimb> hl(L);
  for (int i = 0; i < L; i++) {
    const uint64_t t = ctx->q[i];
    hl[i].tmod = Mod{(double)t, 1.0 / (double)t};
    uint64_t Qm = 1 % t;
    for (int k = 0; k < K; k++) {
      hl[i].Qk_mod_t[k] = balanced(ctx->Q[k] % t, t);
      Qm = mulmod(Qm, ctx->Q[k] % t, t);
    }
    hl[i].Q_mod_t = balanced(Qm, t);
  }
  if (!ctx->d_crt_limbs) {
    void *p = nullptr;
    RS_HIP(hipMalloc(&p, sizeof(CrtLimb) * L));
    if (hipMemcpy(p, hl.data(), sizeof(CrtLimb) * L, hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(p);
      throw rs::Error(RS_ERR_HIP, "upload of the CRT constants failed");
    }
    ctx->d_crt_limbs = p;
  }
  const CrtLimb *d_limbs = static_cast<const CrtLimb *>(ctx->d_crt_limbs);
  TabCopies tabs(ctx);
  double *V = (double *)ws_get(ctx, 14, count * (size_t)L * K * n * sizeof(double));
  const size_t lds = padded_len((size_t)n) * sizeof(double);
  const int thr = enc_threads(ctx->logN_enc);
  set_max_dyn_lds((const void *)decrypt_dot_kernel, (int)lds);
  set_max_dyn_lds((const void *)crt_decode_kernel, (int)lds);
  hipLaunchKernelGGL(decrypt_dot_kernel, dim3((unsigned)(count * L), K), dim3(thr), lds, st, d_enc, d_sk, V, K, ctx->logN_enc,
                     tabs.d_coeff);
  hipLaunchKernelGGL(crt_decode_kernel, dim3((unsigned)(count * L)), dim3(thr), lds, st, V, d_rings, ctx->N, L, ctx->logN_enc,
                     cc, d_limbs, ctx->d_index_map, tabs.d_plain, (const double *)d_thr, tb, d_bits);
  RS_HIP(hipGetLastError());
  RS_HIP(hipStreamSynchronize(st));  // the table copies die with this call
  return noise_verdict(ctx, d_bits, count, tb, d_rings != nullptr, h_budget);
}

extern "C" {

int rs_enc_encode(rs_ctx *ctx, const uint64_t *d_sk, const uint64_t *d_rings, size_t count, uint64_t seed, uint64_t *d_enc,
                  rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_sk && d_rings && d_enc, "null argument");
  if (count == 0) return RS_OK;
  RS_REQUIRE(ctx->N_enc <= 16 * 1024, "encoding degree out of range");
  WsScope ws_scope(ctx, S(stream));
  hipStream_t st = S(stream);
  if (ctx->use_int) {
    TabCopiesI tabs(ctx);
    const size_t lds = padded_len((size_t)ctx->N_enc) * sizeof(uint64_t);
    const int thr = std::max(enc_threads(ctx->logN_enc), ctx->N_enc / 16);
    set_max_dyn_lds((const void *)encode_kernel_int, (int)lds);
    hipLaunchKernelGGL(encode_kernel_int, dim3((unsigned)(count * ctx->L), ctx->K), dim3(thr), lds, st, d_rings, d_sk, d_enc, seed,
                       ctx->N, ctx->L, ctx->K, ctx->logN_enc, ctx->d_index_map, tabs.d_plain, tabs.d_coeff);
    RS_HIP(hipGetLastError());
    RS_HIP(hipStreamSynchronize(st));
    return RS_OK;
  }
  TabCopies tabs(ctx);
  const size_t lds = padded_len((size_t)ctx->N_enc) * sizeof(double);
  const int thr = std::max(enc_threads(ctx->logN_enc), ctx->N_enc / 16);
  set_max_dyn_lds((const void *)encode_kernel, (int)lds);
  hipLaunchKernelGGL(encode_kernel, dim3((unsigned)(count * ctx->L), ctx->K), dim3(thr), lds, st, d_rings, d_sk, d_enc, seed,
                     ctx->N, ctx->L, ctx->K, ctx->logN_enc, ctx->d_index_map, tabs.d_plain, tabs.d_coeff, tabs.d_q, tabs.d_Q);
  RS_HIP(hipGetLastError());
  RS_HIP(hipStreamSynchronize(st));
  RS_API_END
}

}  // extern "C"

// ---- wire format -----------------------------------------------------------------
// The reference declares serialisation of keys and proofs but never implements it
// (r1cs_ppzksnark.hpp:43-47,142-146; variable.tcc:391-414 throws).  Format (little endian):
//   u8[8]  magic "RSNKENC1"
//   u32    N, L, N_enc, K
//   u64    q[L], Q[K]
//   u64    count                       number of encoding elements
//   u8     empty[count]                1 = EMPTY element (seal_ring.tcc:412,432), payload all zero
//   pad to a multiple of 8 bytes
//   u64    payload[count][L][2][K][N_enc]   canonical residues, the layout of the ABI
namespace rs {
static const char kMagic[8] = {'R', 'S', 'N', 'K', 'E', 'N', 'C', '1'};
static size_t wire_header_bytes(const rs_ctx *ctx, size_t count) {
  const size_t raw = 8 + 16 + 8 * (size_t)(ctx->L + ctx->K) + 8 + count;
  return (raw + 7) & ~(size_t)7;
}
}  // namespace rs

extern "C" {

size_t rs_enc_wire_size(const rs_ctx *ctx, size_t count) {
  if (!ctx) return 0;
  return wire_header_bytes(ctx, count) + count * ctx->enc_words() * sizeof(uint64_t);
}

int rs_enc_serialize(rs_ctx *ctx, const uint64_t *d_enc, const uint8_t *h_empty, size_t count, void *h_buf, size_t buf_bytes,
                     rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && h_buf && (d_enc || count == 0), "null argument");
  RS_REQUIRE(buf_bytes >= rs_enc_wire_size(ctx, count), "buffer too small (rs_enc_wire_size)");
  uint8_t *p = (uint8_t *)h_buf;
  const size_t hb = wire_header_bytes(ctx, count);
  memset(p, 0, hb);
  memcpy(p, kMagic, 8);
  const uint32_t dims[4] = {(uint32_t)ctx->N, (uint32_t)ctx->L, (uint32_t)ctx->N_enc, (uint32_t)ctx->K};
  memcpy(p + 8, dims, 16);
  memcpy(p + 24, ctx->q, 8 * (size_t)ctx->L);
  memcpy(p + 24 + 8 * (size_t)ctx->L, ctx->Q, 8 * (size_t)ctx->K);
  const uint64_t c64 = count;
  uint8_t *q = p + 24 + 8 * (size_t)(ctx->L + ctx->K);
  memcpy(q, &c64, 8);
  for (size_t i = 0; i < count; i++) q[8 + i] = h_empty ? (h_empty[i] ? 1 : 0) : 0;
  if (count) {
    RS_HIP(hipMemcpyAsync(p + hb, d_enc, count * ctx->enc_words() * sizeof(uint64_t), hipMemcpyDeviceToHost, S(stream)));
    RS_HIP(hipStreamSynchronize(S(stream)));
    if (h_empty)
      for (size_t i = 0; i < count; i++)
        if (h_empty[i]) memset(p + hb + i * ctx->enc_words() * sizeof(uint64_t), 0, ctx->enc_words() * sizeof(uint64_t));
  }
  RS_API_END
}

int rs_enc_deserialize(rs_ctx *ctx, const void *h_buf, size_t buf_bytes, uint64_t *d_enc, uint8_t *h_empty, size_t capacity,
                       size_t *h_count, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && h_buf && h_count, "null argument");
  const uint8_t *p = (const uint8_t *)h_buf;
  const size_t fixed = 24 + 8 * (size_t)(ctx->L + ctx->K) + 8;
  RS_REQUIRE(buf_bytes >= fixed, "truncated header");
  RS_REQUIRE(memcmp(p, kMagic, 8) == 0, "bad magic: not a ringsnark_amd encoding stream");
  uint32_t dims[4];
  memcpy(dims, p + 8, 16);
  RS_REQUIRE((int)dims[0] == ctx->N && (int)dims[1] == ctx->L && (int)dims[2] == ctx->N_enc && (int)dims[3] == ctx->K,
             "stream was written for different ring / encoding dimensions");
  RS_REQUIRE(memcmp(p + 24, ctx->q, 8 * (size_t)ctx->L) == 0 && memcmp(p + 24 + 8 * (size_t)ctx->L, ctx->Q, 8 * (size_t)ctx->K) == 0,
             "stream was written for different moduli");
  uint64_t c64;
  memcpy(&c64, p + 24 + 8 * (size_t)(ctx->L + ctx->K), 8);
  // The count is untrusted: bound it by what the buffer can hold BEFORE any size arithmetic (every
  // element costs one flag byte + enc_bytes of payload), so that nothing below can wrap.
  const size_t enc_bytes = ctx->enc_words() * sizeof(uint64_t);
  RS_REQUIRE(c64 <= (uint64_t)((buf_bytes - fixed) / (1 + enc_bytes)), "truncated payload");
  const size_t count = (size_t)c64;
  RS_REQUIRE(buf_bytes >= rs_enc_wire_size(ctx, count), "truncated payload");
  const uint8_t *em = p + fixed;
  const size_t hb = wire_header_bytes(ctx, count);
  for (size_t i = 0; i < count; i++) RS_REQUIRE(em[i] <= 1, "corrupt empty flag");
  for (size_t i = fixed + count; i < hb; i++) RS_REQUIRE(p[i] == 0, "non-zero header padding");
  if (!d_enc) {  // size query (header validated)
    *h_count = count;
    return RS_OK;
  }
  RS_REQUIRE(capacity >= count, "destination holds fewer elements than the stream");
  const uint64_t *payload = (const uint64_t *)(p + hb);
  // canonical-residue check on the host: a proof from the wire is untrusted input
  const size_t n = (size_t)ctx->N_enc, per = ctx->enc_words();
  for (size_t i = 0; i < count; i++)
    for (int l = 0; l < ctx->L; l++)
      for (int c = 0; c < 2; c++)
        for (int j = 0; j < ctx->K; j++) {
          const uint64_t *row = payload + i * per + (((size_t)l * 2 + c) * ctx->K + j) * n;
          const uint64_t Qj = ctx->Q[j];
          if (em[i]) {  // EMPTY element (seal_ring.tcc:412,432): the payload must be all zero
            for (size_t x = 0; x < n; x++) RS_REQUIRE(row[x] == 0, "non-zero payload of an EMPTY element");
          } else {
            for (size_t x = 0; x < n; x++) RS_REQUIRE(row[x] < Qj, "residue out of range");
          }
        }
  if (count) {
    RS_HIP(hipMemcpyAsync(d_enc, payload, count * per * sizeof(uint64_t), hipMemcpyHostToDevice, S(stream)));
    RS_HIP(hipStreamSynchronize(S(stream)));
  }
  // outputs are written only after every check has passed
  if (h_empty) memcpy(h_empty, em, count);
  *h_count = count;
  RS_API_END
}

}  // extern "C"

// ---- r1cs_to_qrp_instance_map_with_evaluation --------------------------------------
// ringsnark/reductions/r1cs_to_qrp/r1cs_to_qrp.tcc:76-116 with util/evaluation_domain.tcc:21-50.
// What generator and verifier run before anything else (groth16.tcc:7-9, 127-128; rinocchio.tcc:7-9,
// 219-220).  The reference spends O(m^2) ring multiplications on the Lagrange polynomials; here
//     u_j(s) = Z(s) * (s - j)^-1 * c_j,   c_j = 1 / prod_{i != j} (j - i) = (-1)^(m-1-j) / (j! (m-1-j)!)
// (slot-constant c_j), one batched ring inversion, and At / Bt / Ct = transpose(A/B/C) * u through
// the same sparse kernel as row a14 on the transposed CSR.  Every value is a canonical residue of an
// exact ring expression, hence bit-identical to the reference's order of operations.
namespace rs {
void r1cs_evaluate_run(rs_ctx *ctx, const rs_r1cs *cs, int which, int mode, const uint64_t *d_asg, uint64_t *d_out,
                       hipStream_t st);

// Per slot, with d_j = s - j:  Ht[j] = s^j (j <= m),  Zt = prod_j d_j,  and the Lagrange values
//     u_j = c_j * prod_{i != j} d_i = c_j * (prod_{i < j} d_i) * (prod_{i > j} d_i)
// by a backward pass (suffix products parked in U) and a forward pass (running prefix): no
// division, so a slot in which s happens to equal a domain element is handled exactly like the
// reference's product loop (evaluation_domain.tcc:28-39).  hit[0] / hit[1] = min / max over slots of
// the domain index the slot equals (0xFFFFFFFF: none): s IS a domain element -- the only case the
// reference rejects (evaluation_domain.tcc:24-26) -- iff hit[0] == hit[1] != 0xFFFFFFFF.
__global__ void __launch_bounds__(256)
lagrange_kernel(const uint64_t *__restrict__ s, uint64_t *__restrict__ Ht, uint64_t *__restrict__ U, uint64_t *__restrict__ Zt,
                const double *__restrict__ c /* [L][m] */, unsigned *__restrict__ hit, size_t m, int N, int L,
                const Mod *__restrict__ qmod) {
  const size_t S = (size_t)L * N, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S) return;
  const int limb = (int)(i / (size_t)N);
  const Mod mod = qmod[limb];
  const double sv = center(from_u64(s[i]), mod);
  unsigned my_hit = 0xFFFFFFFFu;
  double suf = 1.0;
  for (size_t j = m; j-- > 0;) {
    U[j * S + i] = to_u64(canon(suf, mod));
    const double d = reduce(sv - (double)j, mod);
    if (canon(d, mod) == 0.0) my_hit = (unsigned)j;
    suf = mulmod(suf, d, mod);
  }
  Zt[i] = to_u64(canon(suf, mod));
  double pre = 1.0, pw = 1.0;
  for (size_t j = 0; j <= m; j++) {
    Ht[j * S + i] = to_u64(canon(pw, mod));
    pw = mulmod(pw, sv, mod);
    if (j < m) {
      const double v = mulmod(pre, center(from_u64(U[j * S + i]), mod), mod);
      U[j * S + i] = to_u64(canon(mulmod(v, c[(size_t)limb * m + j], mod), mod));
      pre = mulmod(pre, reduce(sv - (double)j, mod), mod);
    }
  }
  atomicMin(&hit[0], my_hit);
  atomicMax(&hit[1], my_hit);
}
__global__ void __launch_bounds__(256)
lagrange_kernel_int(const uint64_t *__restrict__ s, uint64_t *__restrict__ Ht, uint64_t *__restrict__ U, uint64_t *__restrict__ Zt,
                    const uint64_t *__restrict__ c /* [L][m], Montgomery form */, unsigned *__restrict__ hit, size_t m, int N, int L,
                    const ModI *__restrict__ qmod) {
  const size_t S = (size_t)L * N, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S) return;
  const int limb = (int)(i / (size_t)N);
  const ModI mod = qmod[limb];
  const uint64_t sv = s[i];
  unsigned my_hit = 0xFFFFFFFFu;
  uint64_t suf = 1;
  for (size_t j = m; j-- > 0;) {
    U[j * S + i] = suf;
    const uint64_t d = subm(sv, small_val((uint64_t)j, mod), mod);
    if (d == 0) my_hit = (unsigned)j;
    suf = mulmod_dd(suf, d, mod);
  }
  Zt[i] = suf;
  uint64_t pre = 1, pw = 1;
  for (size_t j = 0; j <= m; j++) {
    Ht[j * S + i] = pw;
    pw = mulmod_dd(pw, sv, mod);
    if (j < m) {
      const uint64_t v = mulmod_dd(pre, U[j * S + i], mod);
      U[j * S + i] = mulmod(v, c[(size_t)limb * m + j], mod);
      pre = mulmod_dd(pre, subm(sv, small_val((uint64_t)j, mod), mod), mod);
    }
  }
  atomicMin(&hit[0], my_hit);
  atomicMax(&hit[1], my_hit);
}
}  // namespace rs

extern "C" {

int rs_instance_map_eval(rs_ctx *ctx, const rs_r1cs *cs, const uint64_t *d_s, uint64_t *d_At, uint64_t *d_Bt, uint64_t *d_Ct,
                         uint64_t *d_Ht, uint64_t *d_Zt, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && cs && d_s && d_At && d_Bt && d_Ct && d_Ht && d_Zt, "null argument");
  const size_t m = cs->m, SW = ctx->ring_words();
  const int L = ctx->L;
  hipStream_t st = S(stream);
  uint64_t *D = nullptr;
  RS_HIP(hipMalloc(&D, m * SW * sizeof(uint64_t)));
  struct Guard {
    std::vector<void *> p;
    rs_r1cs *tr = nullptr;
    ~Guard() {
      for (void *x : p) (void)hipFree(x);
      if (tr) rs_r1cs_destroy(tr);
    }
  } guard;
  guard.p.push_back(D);
  // slot-constant factors c_j = 1 / prod_{i != j} (j - i) = (-1)^(m-1-j) / (j! (m-1-j)!)
  std::vector<double> hc((size_t)L * m);
  std::vector<uint64_t> hci(ctx->use_int ? (size_t)L * m : 0);  // the same constants for the integer arithmetic
  for (int l = 0; l < L; l++) {
    const uint64_t q = ctx->q[l];
    RS_REQUIRE(q > m, "ring prime too small for the evaluation domain");
    std::vector<uint64_t> fact(m);
    fact[0] = 1;
    for (size_t j = 1; j < m; j++) fact[j] = host::mulmod(fact[j - 1], (uint64_t)j % q, q);
    for (size_t j = 0; j < m; j++) {
      uint64_t v = host::invmod(host::mulmod(fact[j], fact[m - 1 - j], q), q);
      if ((m - 1 - j) & 1) v = v ? q - v : 0;
      hc[(size_t)l * m + j] = host::balanced(v, q);
      if (ctx->use_int) hci[(size_t)l * m + j] = HostArith<ModI>::konst(v, q);
    }
  }
  double *d_c = nullptr;
  RS_HIP(hipMalloc(&d_c, hc.size() * sizeof(double)));
  guard.p.push_back(d_c);
  unsigned *d_hit = nullptr;
  RS_HIP(hipMalloc(&d_hit, 2 * sizeof(unsigned)));
  guard.p.push_back(d_hit);
  const unsigned hit0[2] = {0xFFFFFFFFu, 0u};
  unsigned hit[2];
  static_assert(sizeof(double) == sizeof(uint64_t), "constant buffers are shared between the arithmetics");
  RS_HIP(hipMemcpyAsync(d_c, ctx->use_int ? (const void *)hci.data() : (const void *)hc.data(), hc.size() * sizeof(double),
                        hipMemcpyHostToDevice, st));
  RS_HIP(hipMemcpyAsync(d_hit, hit0, sizeof(hit0), hipMemcpyHostToDevice, st));
  if (ctx->use_int)
    hipLaunchKernelGGL(lagrange_kernel_int, dim3((unsigned)((SW + 255) / 256)), dim3(256), 0, st, d_s, d_Ht, D, d_Zt,
                       reinterpret_cast<const uint64_t *>(d_c), d_hit, m, ctx->N, L, ctx->d_qmod_i);
  else
    hipLaunchKernelGGL(lagrange_kernel, dim3((unsigned)((SW + 255) / 256)), dim3(256), 0, st, d_s, d_Ht, D, d_Zt, d_c, d_hit, m,
                       ctx->N, L, ctx->d_qmod);
  RS_HIP(hipGetLastError());
  RS_HIP(hipMemcpyAsync(hit, d_hit, sizeof(hit), hipMemcpyDeviceToHost, st));
  RS_HIP(hipStreamSynchronize(st));  // hc, hit
  // evaluation_domain.tcc:24-26: rejected only when s equals a domain element AS A RING ELEMENT
  // (every slot of every limb equal to the same j)
  if (hit[0] == hit[1] && hit[0] != 0xFFFFFFFFu)
    throw Error(RS_ERR_NOT_INVERTIBLE, "t cannot be one of the values in the domain");
  // transposed system: row k (variable k, 0 = the constant one) holds (constraint i + 1, coeff)
  const size_t rows = cs->n_vars + 1;
  std::vector<uint32_t> rp[3], col[3];
  std::vector<uint64_t> cf[3];
  std::vector<int32_t> pi[3];  // polynomial coefficients travel with their terms (same table)
  const uint32_t *rpp[3], *colp[3];
  const uint64_t *cfp[3];
  const int32_t *pip[3];
  size_t nnz[3];
  for (int w = 0; w < 3; w++) {
    const size_t z = cs->nnz[w];
    nnz[w] = z;
    rp[w].assign(rows + 1, 0);
    for (size_t e = 0; e < z; e++) rp[w][cs->h_col[w][e] + 1]++;
    for (size_t k = 0; k < rows; k++) rp[w][k + 1] += rp[w][k];
    col[w].resize(std::max<size_t>(z, 1));
    cf[w].resize(std::max<size_t>((size_t)L * z, 1));
    pi[w].assign(std::max<size_t>(z, 1), -1);
    std::vector<uint32_t> fill(rp[w].begin(), rp[w].end() - 1);
    for (size_t i = 0; i < m; i++)
      for (uint32_t e = cs->h_row_ptr[w][i]; e < cs->h_row_ptr[w][i + 1]; e++) {
        const uint32_t dst = fill[cs->h_col[w][e]]++;
        col[w][dst] = (uint32_t)(i + 1);
        for (int l = 0; l < L; l++) cf[w][(size_t)l * z + dst] = cs->h_coeff[w][(size_t)l * z + e];
        if (!cs->h_pidx[w].empty()) pi[w][dst] = cs->h_pidx[w][e];
      }
    rpp[w] = rp[w].data();
    colp[w] = col[w].data();
    cfp[w] = cf[w].data();
    pip[w] = cs->h_pidx[w].empty() ? nullptr : pi[w].data();
  }
  RS_REQUIRE(rs_r1cs_create_poly(ctx, rows, m, 0, rpp, colp, cfp, nnz, pip, cs->n_poly ? cs->h_ptab.data() : nullptr, cs->n_poly,
                                 &guard.tr) == RS_OK,
             rs_last_error());
  uint64_t *outs[3] = {d_At, d_Bt, d_Ct};
  for (int w = 0; w < 3; w++) r1cs_evaluate_run(ctx, guard.tr, w, RS_EVAL_FULL, D, outs[w], st);
  RS_HIP(hipStreamSynchronize(st));
  RS_API_END
}

}  // extern "C"
