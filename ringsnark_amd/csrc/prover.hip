// prover.hip -- groth16::prover (row a15) and rinocchio::prover (row a16) on the device, plus the
// synthetic-workload generators used by the benchmark harness.
#include <algorithm>
#include <cstring>

#include "rs_internal.hpp"

namespace rs {
void msm_run(rs_ctx *ctx, const uint64_t *const *d_crs, int n_crs, size_t crs_len, const rs_msm_vec *vecs, int n_vecs,
             int n_groups, uint64_t *d_out, const uint64_t *const *addends, size_t *h_used, hipStream_t st,
             size_t crs_window, const MsmLin *lin = nullptr, bool crs_on_host = false);
bool msm_supports_lin(const rs_ctx *ctx);
void batch_encode_run(rs_ctx *ctx, const uint64_t *d_rings, uint64_t *d_plain, size_t count, hipStream_t st);
bool witness_io_shortcut(const rs_r1cs *cs);
void enc_add_run(rs_ctx *ctx, uint64_t *dst, const uint64_t *x, const uint64_t *y, size_t count, hipStream_t st);
const uint64_t *witness_Z_rows(rs_ctx *ctx, size_t m);
void witness_run(rs_ctx *ctx, const rs_r1cs *cs, const uint64_t *d_asg, const uint64_t *d1, const uint64_t *d2,
                 const uint64_t *d3, uint64_t *const outs[7], uint64_t *h_Z, hipStream_t st, int slot0 = 0, int nslots = -1,
                 bool compact = false, const size_t (*rows)[2] = nullptr);
void msm_scratch_release(rs_ctx *ctx);

int g_prover_lin_io = 1;  // tuning knob "prover_lin_io": io vectors of groth16::prover as linear forms (MsmLin)

__global__ void __launch_bounds__(256) fill_ones_kernel(uint64_t *p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 1;
}
static void fill_ones(rs_ctx *, uint64_t *p, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(fill_ones_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n);
}

struct PhaseTimer {
  rs_ctx *ctx;
  hipStream_t st;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  explicit PhaseTimer(rs_ctx *c, hipStream_t s) : ctx(c), st(s) {
    if (!ctx->profiling) return;
    for (auto &e : ev) RS_HIP(hipEventCreate(&e));
  }
  void mark(int k) {
    if (ctx->profiling) RS_HIP(hipEventRecord(ev[k], st));
  }
  void finish() {
    if (!ctx->profiling) return;
    RS_HIP(hipEventSynchronize(ev[2]));
    float w = 0, msm = 0;
    RS_HIP(hipEventElapsedTime(&w, ev[0], ev[1]));
    RS_HIP(hipEventElapsedTime(&msm, ev[1], ev[2]));
    ctx->timings.evaluate_ms = 0;
    ctx->timings.witness_ms = w;
    ctx->timings.msm_ms = msm;
    ctx->timings.total_ms = w + msm;
    for (auto &e : ev) (void)hipEventDestroy(e);
  }
};

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void __launch_bounds__(256)
fill_uniform_kernel(uint64_t *__restrict__ dst, size_t words, size_t inner, int nmod, const uint64_t *__restrict__ mods,
                    uint64_t seed) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) {
    const uint64_t p = mods[(i / inner) % (size_t)nmod];
    dst[i] = mix64(mix64(seed) ^ (uint64_t)i) % p;
  }
}

// x_{i+2} = x_i * x_{i+1}; one thread per slot, values carried in registers.
template <class M>
__global__ void __launch_bounds__(256)
chain_kernel(uint64_t *__restrict__ asg, size_t m, int N, int L, const M *__restrict__ qmod) {
  using T = typename ArithOf<M>::T;
  const size_t S = (size_t)L * N;
  const size_t sl = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (sl >= S) return;
  const M mod = qmod[sl / (size_t)N];
  T a = center(from_res<T>(asg[sl]), mod), b = center(from_res<T>(asg[S + sl]), mod);
  for (size_t i = 0; i < m; i++) {
    const T c = reduce(mulmod_dd(a, b, mod), mod);
    asg[(i + 2) * S + sl] = to_res(canon(c, mod));
    a = b;
    b = c;
  }
}

}  // namespace rs

using namespace rs;

extern "C" {

int rs_groth16_prove(rs_ctx *ctx, const rs_r1cs *cs, const rs_groth16_pk *pk, const uint64_t *d_assignment,
                     uint64_t *d_proof, int *h_empty, rs_stream stream) {
  return rs_groth16_prove_kinds(ctx, cs, pk, d_assignment, nullptr, d_proof, h_empty, stream);
}

int rs_groth16_prove_kinds(rs_ctx *ctx, const rs_r1cs *cs, const rs_groth16_pk *pk, const uint64_t *d_assignment,
                           const uint8_t *h_assignment_kinds, uint64_t *d_proof, int *h_empty, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && cs && pk && d_assignment && d_proof, "null argument");
  RS_REQUIRE(pk->d_s_pows && pk->d_delta_ts && pk->d_alpha && pk->d_beta, "incomplete proving key");
  WsScope ws_scope(ctx, S(stream));
  hipStream_t st = S(stream);
  const size_t m = cs->m, rw = ctx->ring_words(), ew = ctx->enc_words();
  const size_t n_aux = cs->n_vars - cs->n_inputs;
  RS_REQUIRE(n_aux == 0 || pk->d_delta_mid, "delta_mid missing");
  const bool host_key = pk->host_key != 0;
  memset(&ctx->timings, 0, sizeof(ctx->timings));
  PhaseTimer pt(ctx, st);
  pt.mark(0);
  // witness map (groth16.tcc:82-84: d1 = d2 = d3 = 0); C_io / C_mid are not consumed by the prover
  uint64_t *wbuf = (uint64_t *)ws_get(ctx, 8, (5 * m + 1) * rw * sizeof(uint64_t));
  uint64_t *A_io = wbuf, *A_mid = wbuf + m * rw, *B_io = wbuf + 2 * m * rw, *B_mid = wbuf + 3 * m * rw, *H = wbuf + 4 * m * rw;
  // The io vectors are linear forms of the primary inputs (witness.hip): when the inner product can take them in that
  // form (MsmLin) they are neither written by the witness map nor read and transformed by the inner product.
  const size_t nk = cs->n_inputs + 1;  // [1, x_1 .. x_n_inputs]; their encodings are staged in the (then unused) io rows
  const bool lin = g_prover_lin_io && msm_supports_lin(ctx) && witness_io_shortcut(cs) &&
                   nk * std::max<size_t>(rw, (size_t)ctx->L * ctx->N_enc) <= m * rw;
  uint64_t *outs[7] = {lin ? nullptr : A_io, lin ? nullptr : B_io, nullptr, A_mid, B_mid, nullptr, H};
  witness_run(ctx, cs, d_assignment, nullptr, nullptr, nullptr, outs, nullptr, st);
  pt.mark(1);
  // A = <s_pows, A_io> + <s_pows, A_mid> + alpha ; B likewise with beta   (groth16.tcc:89-103)
  {
    const uint64_t *crs[1] = {pk->d_s_pows};
    const uint64_t *add[2] = {pk->d_alpha, pk->d_beta};
    if (lin) {
      // plaintexts of [1, x_1 .. x_n_inputs]: (n_inputs + 1) batch encodings, staged in the (unused) A_io rows
      uint64_t *rings = A_io, *P = B_io;  // [nk][L][N] and [nk][L][N_enc] words
      fill_ones(ctx, rings, rw, st);
      if (cs->n_inputs) RS_HIP(hipMemcpyAsync(rings + rw, d_assignment, cs->n_inputs * rw * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
      batch_encode_run(ctx, rings, P, nk, st);
      MsmLin ln[2];
      for (int w = 0; w < 2; w++) {
        ln[w].k = cs->d_io_k[w];
        ln[w].col = cs->d_io_c[w];
        ln[w].count = cs->io_count[w];
        ln[w].Lcols = cs->d_io_cols;
        ln[w].Mlen = cs->io_M;
        ln[w].P = P;
        ln[w].T = m;
      }
      rs_msm_vec v[2] = {{A_mid, nullptr, m, 0}, {B_mid, nullptr, m, 1}};
      msm_run(ctx, crs, 1, m + 1, v, 2, 2, d_proof, add, nullptr, st, pk->window, ln, host_key);
    } else {
      rs_msm_vec v[4] = {{A_io, nullptr, m, 0}, {A_mid, nullptr, m, 0}, {B_io, nullptr, m, 1}, {B_mid, nullptr, m, 1}};
      msm_run(ctx, crs, 1, m + 1, v, 4, 2, d_proof, add, nullptr, st, pk->window, nullptr, host_key);
    }
  }
  // C = <delta_ts, H> (+ <delta_mid, aux>)                                 (groth16.tcc:105-112)
  size_t used_h = 1, used_aux = 0;
  uint64_t *C = d_proof + 2 * ew;
  if (n_aux) {
    const uint64_t *crs[1] = {pk->d_delta_mid};
    // the wires as the caller holds them: a Scalar-1 wire passes its key element through (seal_ring.tcc:525-527)
    rs_msm_vec v{d_assignment + cs->n_inputs * rw, h_assignment_kinds ? h_assignment_kinds + cs->n_inputs : nullptr, n_aux, 0};
    msm_run(ctx, crs, 1, n_aux, &v, 1, 1, C, nullptr, h_empty ? &used_aux : nullptr, st, pk->window, nullptr, host_key);
  }
  {
    const uint64_t *crs[1] = {pk->d_delta_ts};
    rs_msm_vec v{H, nullptr, m + 1, 0};
    const uint64_t *add[1] = {n_aux ? C : nullptr};
    msm_run(ctx, crs, 1, m + 1, &v, 1, 1, C, add, h_empty ? &used_h : nullptr, st, pk->window, nullptr, host_key);
  }
  pt.mark(2);
  pt.finish();
  if (host_key || h_assignment_kinds) RS_HIP(hipStreamSynchronize(st));  // the caller may release or rewrite the host key / the kinds on return
  if (h_empty) {
    h_empty[0] = h_empty[1] = 0;  // alpha / beta are always added
    h_empty[2] = (used_h == 0 && used_aux == 0) ? 1 : 0;
  }
  RS_API_END
}

int rs_rinocchio_prove(rs_ctx *ctx, const rs_r1cs *cs, const rs_rinocchio_pk *pk, const uint64_t *d_assignment,
                       const uint64_t *d_d1, const uint64_t *d_d2, const uint64_t *d_d3, uint64_t *d_proof, int *h_empty,
                       rs_stream stream) {
  return rs_rinocchio_prove_kinds(ctx, cs, pk, d_assignment, nullptr, d_d1, d_d2, d_d3, d_proof, h_empty, stream);
}

int rs_rinocchio_prove_kinds(rs_ctx *ctx, const rs_r1cs *cs, const rs_rinocchio_pk *pk, const uint64_t *d_assignment,
                             const uint8_t *h_assignment_kinds, const uint64_t *d_d1, const uint64_t *d_d2, const uint64_t *d_d3,
                             uint64_t *d_proof, int *h_empty, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && cs && pk && d_assignment && d_proof, "null argument");
  RS_REQUIRE(pk->d_s_pows && pk->d_alpha_s_pows, "incomplete proving key");
  RS_REQUIRE((d_d1 && d_d2 && d_d3) || (!d_d1 && !d_d2 && !d_d3), "d1,d2,d3 must be all set or all null");
  WsScope ws_scope(ctx, S(stream));
  hipStream_t st = S(stream);
  const size_t m = cs->m, rw = ctx->ring_words(), ew = ctx->enc_words();
  const size_t n_aux = cs->n_vars - cs->n_inputs;
  const bool zk = d_d1 != nullptr;  // rinocchio.tcc:81-90
  RS_REQUIRE(n_aux == 0 || pk->d_beta_prods, "beta_prods missing");
  RS_REQUIRE(!zk || n_aux == 0 || (pk->d_beta_rv_ts && pk->d_beta_rw_ts && pk->d_beta_ry_ts), "beta_r*_ts missing");
  memset(&ctx->timings, 0, sizeof(ctx->timings));
  PhaseTimer pt(ctx, st);
  pt.mark(0);
  uint64_t *wbuf = (uint64_t *)ws_get(ctx, 8, (4 * m + 1) * rw * sizeof(uint64_t));
  uint64_t *A_mid = wbuf, *B_mid = wbuf + m * rw, *C_mid = wbuf + 2 * m * rw, *H = wbuf + 3 * m * rw;
  uint64_t *outs[7] = {nullptr, nullptr, nullptr, A_mid, B_mid, C_mid, H};
  witness_run(ctx, cs, d_assignment, d_d1, d_d2, d_d3, outs, nullptr, st);
  // coefficients_for_Z are slot constant: they go to the inner products as the compact [m+1][L] array of their values
  // (rs_msm_vec::slot_const) -- round 3 materialised m + 1 ring elements for them (a fifth of the prover's vectors).
  // A constant of the (context, m) plan: cached on the device, no host transpose and no synchronisation inside the proof.
  const uint64_t *dZ = witness_Z_rows(ctx, m);
  pt.mark(1);
  // the ten inner products of rinocchio.tcc:106-163 in one grouped pass over both CRS vectors
  uint64_t *mo = (uint64_t *)ws_get(ctx, 10, 11 * ew * sizeof(uint64_t));  // [2][5] + tmp
  uint64_t *tmp = mo + 10 * ew;
  std::vector<uint8_t> zkinds(m + 1, RS_KIND_POLY);
  zkinds[m] = RS_KIND_ONE;  // leading coefficient of Z is the RingElem Scalar 1 (evaluation_domain.tcc:55-58)
  size_t used[5] = {0, 0, 0, 0, 0};
  {
    const uint64_t *crs[2] = {pk->d_s_pows, pk->d_alpha_s_pows};
    rs_msm_vec v[5] = {{A_mid, nullptr, m, 0}, {B_mid, nullptr, m, 1}, {C_mid, nullptr, m, 2}, {H, nullptr, m + 1, 3},
                       {dZ, zkinds.data(), m + 1, 4, 1}};
    msm_run(ctx, crs, 2, m + 1, v, 5, 5, mo, nullptr, used, st, pk->window, nullptr, pk->host_key != 0);
  }
  auto slot = [&](int c, int g) { return mo + ((size_t)c * 5 + g) * ew; };
  int empty[9];
  for (int k = 0; k < 4; k++) {
    empty[2 * k] = empty[2 * k + 1] = used[k] == 0;
    RS_HIP(hipMemcpyAsync(d_proof + (size_t)(2 * k) * ew, slot(0, k), ew * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    RS_HIP(hipMemcpyAsync(d_proof + (size_t)(2 * k + 1) * ew, slot(1, k), ew * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
  }
  auto add_scaled = [&](uint64_t *dst, int *dst_empty, const uint64_t *enc, const uint64_t *d) {
    // dst += d * enc   (RingT * EncT then +=, rinocchio.tcc:168-173, 181-183)
    rs_msm_vec v{d, nullptr, 1, 0};
    const uint64_t *crs[1] = {enc};
    const uint64_t *add[1] = {*dst_empty ? nullptr : dst};
    msm_run(ctx, crs, 1, 1, &v, 1, 1, *dst_empty ? dst : tmp, add, nullptr, st, 0);
    if (!*dst_empty)
      RS_HIP(hipMemcpyAsync(dst, tmp, ew * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    *dst_empty = 0;
  };
  if (zk) {  // rinocchio.tcc:167-174
    const uint64_t *ds[3] = {d_d1, d_d2, d_d3};
    for (int k = 0; k < 3; k++) {
      add_scaled(d_proof + (size_t)(2 * k) * ew, &empty[2 * k], slot(0, 4), ds[k]);
      add_scaled(d_proof + (size_t)(2 * k + 1) * ew, &empty[2 * k + 1], slot(1, 4), ds[k]);
    }
  }
  // F (rinocchio.tcc:176-185)
  empty[8] = 1;
  uint64_t *F = d_proof + 8 * ew;
  RS_HIP(hipMemsetAsync(F, 0, ew * sizeof(uint64_t), st));
  if (n_aux) {
    size_t used_f = 0;
    const uint64_t *crs[1] = {pk->d_beta_prods};
    rs_msm_vec v{d_assignment + cs->n_inputs * rw, h_assignment_kinds ? h_assignment_kinds + cs->n_inputs : nullptr, n_aux, 0};
    msm_run(ctx, crs, 1, n_aux, &v, 1, 1, F, nullptr, &used_f, st, pk->window, nullptr, pk->host_key != 0);
    empty[8] = used_f == 0;
    if (zk) {
      add_scaled(F, &empty[8], pk->d_beta_rv_ts, d_d1);
      add_scaled(F, &empty[8], pk->d_beta_rw_ts, d_d2);
      add_scaled(F, &empty[8], pk->d_beta_ry_ts, d_d3);
    }
  }
  pt.mark(2);
  pt.finish();
  RS_HIP(hipStreamSynchronize(st));  // zkinds is a host temporary referenced by async copies
  if (h_empty) memcpy(h_empty, empty, sizeof(empty));
  RS_API_END
}

int rs_fill_uniform(rs_ctx *ctx, uint64_t *d_dst, size_t count, int layout, uint64_t seed, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_dst && (layout == 0 || layout == 1), "bad argument");
  WsScope ws_scope(ctx, S(stream));
  uint64_t *mods = (uint64_t *)ws_get(ctx, 11, sizeof(uint64_t) * (RS_MAX_L + RS_MAX_K));
  RS_HIP(hipMemcpyAsync(mods, ctx->q, sizeof(uint64_t) * ctx->L, hipMemcpyHostToDevice, S(stream)));
  RS_HIP(hipMemcpyAsync(mods + RS_MAX_L, ctx->Q, sizeof(uint64_t) * ctx->K, hipMemcpyHostToDevice, S(stream)));
  const size_t words = count * (layout == 0 ? ctx->ring_words() : ctx->enc_words());
  if (words) {
    const unsigned blocks = (unsigned)std::min<size_t>((words + 255) / 256, 256 * 32);
    if (layout == 0)
      hipLaunchKernelGGL(fill_uniform_kernel, dim3(blocks), dim3(256), 0, S(stream), d_dst, words, (size_t)ctx->N, ctx->L,
                         mods, seed);
    else
      hipLaunchKernelGGL(fill_uniform_kernel, dim3(blocks), dim3(256), 0, S(stream), d_dst, words, (size_t)ctx->N_enc,
                         ctx->K, mods + RS_MAX_L, seed);
    RS_HIP(hipGetLastError());
  }
  RS_API_END
}

int rs_chain_assignment(rs_ctx *ctx, uint64_t *d_assignment, size_t m, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_assignment, "null argument");
  const size_t S_ = ctx->ring_words();
  if (ctx->use_int)
    hipLaunchKernelGGL(chain_kernel<ModI>, dim3((unsigned)((S_ + 255) / 256)), dim3(256), 0, S(stream), d_assignment, m, ctx->N,
                       ctx->L, ctx->d_qmod_i);
  else
    hipLaunchKernelGGL(chain_kernel<Mod>, dim3((unsigned)((S_ + 255) / 256)), dim3(256), 0, S(stream), d_assignment, m, ctx->N,
                       ctx->L, ctx->d_qmod);
  RS_HIP(hipGetLastError());
  RS_API_END
}

}  // extern "C"
