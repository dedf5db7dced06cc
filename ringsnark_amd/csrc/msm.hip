// msm.hip -- EncodingElem pieces (rows a5-a8) and the encoding inner product / ring-MSM (row a9)
// of SURVEY.md section 8, restructured for the GPU:
//
//   reference (seal_ring.tcc:415-431, per term and ring limb i):
//       plain = iNTT_{q_i}(scatter(b_t[i]))                 BatchEncoder::encode      (:534)
//       for j < K:  P_j = NTT_{Q_j}(centred_lift(plain))    multiply_plain_inplace    (:536)
//                   tmp[c][j] = ct_t[i][c][j] * P_j
//       res += tmp                                           add_inplace              (:494)
//
//   here, three kernels over a tile of terms:
//     plain_center_kernel  one workgroup per (group, term, limb): scatter + inverse NTT mod q_i in
//                          LDS, centred lift, SUM over the coefficient vectors of the group ->
//                          C[g][t][i][N_enc] (signed doubles).  Summing after the lift is exact:
//                          ct*(P1) + ct*(P2) = ct*NTT(lift(p1)+lift(p2)) in Z_{Q_j}.
//     mac_kernel           one workgroup per (limb, prime j, term chunk): C mod Q_j -> forward NTT
//                          in LDS -> multiply with both ciphertext polynomials streamed from HBM,
//                          accumulate in registers across the chunk's terms (lazy reduction).
//                          Every ciphertext word is read exactly once per pass.
//     reduce_kernel        sums the per-chunk partial accumulators (+ optional addends).
//
// Skipped (is_zero) terms contribute a zero plaintext, which is the identity for the sum, so the
// value-level result equals the reference's; the EMPTY result (all terms skipped) is reported
// through the used-term counts.
//
// Kernels: msm_plain.hpp (plaintext rows), msm_mac.hpp (multiply-accumulate, reduction).  This file: scratch tables, launch
// geometry, the tiled / host-streamed key loop (msm_run) and the extern "C" entry points.
#include <algorithm>
#include <cstring>
#include <type_traits>

#include "ntt_core.hpp"
#include "ntt_wide.hpp"
#include "rs_internal.hpp"
#include "msm_plain.hpp"
#include "msm_mac.hpp"

namespace rs {

static int tile_threads(int logn) { return std::max(64, std::min(1024, (1 << logn) >> 3)); }

struct MsmScratch {
  void *d_plain_tabs = nullptr, *d_coeff_tabs = nullptr;  // device copies of the context's tables (NttTable or NttTableI)
  uint64_t *d_Qint = nullptr;
  uint64_t *d_ones_plain = nullptr;  // [L][N_enc]: batch encoding of the ring element (1, ..., 1), built at first use (slot-constant vectors)
  void *d_coeff_tabs_f64 = nullptr;  // hybrid contexts: FP64 tables of the data primes beside the integer ones
  // host-resident keys: copy stream and the events of the two staging buffers (copied: data landed; freed: its readers ran)
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_freed[2] = {nullptr, nullptr};
  template <class M>
  const NttTableT<typename ArithOf<M>::T, M> *plain() const {
    return static_cast<const NttTableT<typename ArithOf<M>::T, M> *>(d_plain_tabs);
  }
  template <class M>
  const NttTableT<typename ArithOf<M>::T, M> *coeff() const {
    return static_cast<const NttTableT<typename ArithOf<M>::T, M> *>(d_coeff_tabs);
  }
};

static std::map<rs_ctx *, MsmScratch> g_scratch;
static std::mutex g_scratch_mu;

template <class Table>
static void *copy_tables(const Table *h, int n) {
  void *d = nullptr;
  RS_HIP(hipMalloc(&d, sizeof(Table) * n));
  RS_HIP(hipMemcpy(d, h, sizeof(Table) * n, hipMemcpyHostToDevice));
  return d;
}
static MsmScratch &scratch_for(rs_ctx *ctx) {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  auto it = g_scratch.find(ctx);
  if (it != g_scratch.end()) return it->second;
  MsmScratch sc;
  if (ctx->use_int) {
    sc.d_plain_tabs = copy_tables(ctx->plain_i, ctx->L);
    sc.d_coeff_tabs = copy_tables(ctx->coeff_i, ctx->K);
    if (ctx->hybrid) sc.d_coeff_tabs_f64 = copy_tables(ctx->coeff, ctx->K);
  } else {
    sc.d_plain_tabs = copy_tables(ctx->plain, ctx->L);
    sc.d_coeff_tabs = copy_tables(ctx->coeff, ctx->K);
  }
  RS_HIP(hipMalloc(&sc.d_Qint, sizeof(uint64_t) * ctx->K));
  RS_HIP(hipMemcpy(sc.d_Qint, ctx->Q, sizeof(uint64_t) * ctx->K, hipMemcpyHostToDevice));
  return g_scratch[ctx] = sc;
}
void msm_scratch_release(rs_ctx *ctx) {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  auto it = g_scratch.find(ctx);
  if (it == g_scratch.end()) return;
  (void)hipFree(it->second.d_plain_tabs);
  (void)hipFree(it->second.d_coeff_tabs);
  (void)hipFree(it->second.d_Qint);
  if (it->second.d_ones_plain) (void)hipFree(it->second.d_ones_plain);
  if (it->second.d_coeff_tabs_f64) (void)hipFree(it->second.d_coeff_tabs_f64);
  if (it->second.copy_stream) {
    (void)hipStreamDestroy(it->second.copy_stream);
    for (int b = 0; b < 2; b++) {
      (void)hipEventDestroy(it->second.ev_copied[b]);
      (void)hipEventDestroy(it->second.ev_freed[b]);
    }
  }
  g_scratch.erase(it);
}

template <int NG, int NC, int PAIRS, class M>
static void launch_mac(rs_ctx *ctx, const MacArgs &a, const MsmScratch &sc, hipStream_t st) {
  const size_t lds = padded_len((size_t)ctx->N_enc) * sizeof(double);
  const int rows = a.n_chunks * ctx->L;
  const unsigned blocks = (unsigned)(((rows + 7) / 8) * 8 * ctx->K);
  set_max_dyn_lds((const void *)mac_kernel<NG, NC, PAIRS, M>, (int)lds);
  hipLaunchKernelGGL((mac_kernel<NG, NC, PAIRS, M>), dim3(blocks), dim3(tile_threads(ctx->logN_enc)), lds, st, a, ctx->L,
                     ctx->K, ctx->logN_enc, sc.template coeff<M>());
  RS_HIP(hipGetLastError());
}

extern int g_mac_variant, g_mac_ablate, g_plain_variant, g_mac_chunk_units, g_mac_share_keys;
static void launch_mac_v2(rs_ctx *ctx, const MacArgs2 &a, const MsmScratch &sc, hipStream_t st) {
  const size_t lds = (padded_len((size_t)ctx->N_enc) + (size_t)ctx->N_enc) * sizeof(double);
  const int rows = a.n_chunks * ctx->L;
  const unsigned blocks = (unsigned)(((rows + 7) / 8) * 8 * ctx->K);
  if (g_mac_variant == 3 && ctx->logN_enc == 13) {  // 16 waves of 128 VGPRs (4 waves per SIMD), plaintext row not prefetched
    set_max_dyn_lds((const void *)mac_kernel_v2<1024, 13>, (int)lds);
    hipLaunchKernelGGL((mac_kernel_v2<1024, 13>), dim3(blocks), dim3(1024), lds, st, a, ctx->L, ctx->K, ctx->logN_enc, sc.coeff<Mod>());
    RS_HIP(hipGetLastError());
    return;
  }
  if (ctx->logN_enc == 13 && g_mac_variant != 4) {
#define RS_MAC_LAUNCH(AB)                                                                                             \
  do {                                                                                                                \
    set_max_dyn_lds((const void *)mac_kernel_v2<512, 13, AB>, \
                               (int)lds);                                                                           \
    hipLaunchKernelGGL((mac_kernel_v2<512, 13, AB>), dim3(blocks), dim3(512), lds, st, a, ctx->L, ctx->K,             \
                       ctx->logN_enc, sc.coeff<Mod>());                                                               \
  } while (0)
#ifdef RS_EXPERIMENTS  // timing-only ablations (wrong results): never part of the release library
    switch (g_mac_ablate) {
      case 1: RS_MAC_LAUNCH(1); break;
      case 2: RS_MAC_LAUNCH(2); break;
      case 6: RS_MAC_LAUNCH(6); break;
      default: RS_MAC_LAUNCH(0); break;
    }
#else
    RS_MAC_LAUNCH(0);
#endif
#undef RS_MAC_LAUNCH
    RS_HIP(hipGetLastError());
    return;
  }
  set_max_dyn_lds((const void *)mac_kernel_v2<512>, (int)lds);
  hipLaunchKernelGGL(mac_kernel_v2<512>, dim3(blocks), dim3(512), lds, st, a, ctx->L, ctx->K, ctx->logN_enc, sc.coeff<Mod>());
  RS_HIP(hipGetLastError());
}

// Reduction points of the forward transform mod p when the inputs are bounded by |x| <= b0 (not by p): bit s = every value
// is brought back to |v| <= p/2 before stage s, so that multiplier operands stay below 2^50 (f64mod.hpp).
static uint32_t fwd_reduce_mask_from(uint64_t p, int logn, double b0, double *end_bound) {
  const double lim = 1125899906842624.0 / (double)p;
  double B = b0 / (double)p;
  uint32_t mask = 0;
  for (int s = 0; s < logn; s++) {
    if (B > lim) {
      mask |= 1u << s;
      B = 0.51;
    }
    B += 0.75;
  }
  *end_bound = B * (double)p;  // bound of the spectrum values
  return mask;
}
static void launch_mac_v3(rs_ctx *ctx, const MacArgs3 &a, const MsmScratch &sc, hipStream_t st) {
  const size_t lds = (size_t)(2 * (4096 + 256) + 1024) * sizeof(double);  // two tiles + round-2 twiddles
  const unsigned rows = (unsigned)ctx->L * (unsigned)a.n_chunks;  // (chunk, limb), spread over the XCDs
  // FP64 tables of the data primes: the context's own, or -- hybrid context -- the copies kept beside the integer tables
  const NttTable *tabs = ctx->use_int ? static_cast<const NttTable *>(sc.d_coeff_tabs_f64) : sc.coeff<Mod>();
  const unsigned parts = (unsigned)ctx->N_enc / 4096u;  // workgroups per (limb, prime): halves at 8192 points, quarters at 16384
  const unsigned blocks = ((rows + 7) / 8) * 8 * parts * (unsigned)ctx->K * (unsigned)a.n_groups;
  if (ctx->N_enc == 16384) {
    set_max_dyn_lds((const void *)mac_kernel_v3<false, 14>, (int)lds);
    hipLaunchKernelGGL((mac_kernel_v3<false, 14>), dim3(blocks), dim3(256), lds, st, a, ctx->L, ctx->K, tabs);
  } else if (a.paired) {
    set_max_dyn_lds((const void *)mac_kernel_v3<true>, (int)lds);
    hipLaunchKernelGGL(mac_kernel_v3<true>, dim3(blocks), dim3(256), lds, st, a, ctx->L, ctx->K, tabs);
  } else {
    set_max_dyn_lds((const void *)mac_kernel_v3<false>, (int)lds);
    hipLaunchKernelGGL(mac_kernel_v3<false>, dim3(blocks), dim3(256), lds, st, a, ctx->L, ctx->K, tabs);
  }
  RS_HIP(hipGetLastError());
}

// mac_kernel_v4 with ONE key vector (mac_variant 6, N_enc = 8192): two 512-thread workgroups per CU at <= 128 VGPRs
static void launch_mac_v4_one(rs_ctx *ctx, const MacArgs4 &a, bool paired, const MsmScratch &sc, hipStream_t st) {
  const size_t lds = (size_t)2 * (4096 + 512) * sizeof(double);  // two tiles
  const unsigned rows = (unsigned)ctx->L * (unsigned)a.n_chunks;
  const NttTable *tabs = ctx->use_int ? static_cast<const NttTable *>(sc.d_coeff_tabs_f64) : sc.coeff<Mod>();
  const unsigned blocks = ((rows + 7) / 8) * 8 * 2u * (unsigned)ctx->K * (unsigned)a.n_groups;
  if (paired) {
    set_max_dyn_lds((const void *)mac_kernel_v4<13, true, 1>, (int)lds);
    hipLaunchKernelGGL((mac_kernel_v4<13, true, 1>), dim3(blocks), dim3(512), lds, st, a, ctx->L, ctx->K, tabs);
  } else {
    set_max_dyn_lds((const void *)mac_kernel_v4<13, false, 1>, (int)lds);
    hipLaunchKernelGGL((mac_kernel_v4<13, false, 1>), dim3(blocks), dim3(512), lds, st, a, ctx->L, ctx->K, tabs);
  }
  RS_HIP(hipGetLastError());
}

static void launch_mac_v4(rs_ctx *ctx, const MacArgs4 &a, bool paired, const MsmScratch &sc, hipStream_t st) {
  const size_t lds = (size_t)2 * (4096 + 512) * sizeof(double);  // two tiles
  const unsigned rows = (unsigned)ctx->L * (unsigned)a.n_chunks;
  const NttTable *tabs = ctx->use_int ? static_cast<const NttTable *>(sc.d_coeff_tabs_f64) : sc.coeff<Mod>();
  const unsigned parts = (unsigned)ctx->N_enc / 4096u;
  const unsigned blocks = ((rows + 7) / 8) * 8 * parts * (unsigned)ctx->K * (unsigned)a.n_groups;
#define RS_MAC4_LAUNCH(LOGN_, PAIRED_)                                                                                          \
  do {                                                                                                                         \
    set_max_dyn_lds((const void *)mac_kernel_v4<LOGN_, PAIRED_>, (int)lds); \
    hipLaunchKernelGGL((mac_kernel_v4<LOGN_, PAIRED_>), dim3(blocks), dim3(512), lds, st, a, ctx->L, ctx->K, tabs);             \
  } while (0)
  if (ctx->N_enc == 16384) RS_MAC4_LAUNCH(14, false);
  else if (paired) RS_MAC4_LAUNCH(13, true);
  else RS_MAC4_LAUNCH(13, false);
#undef RS_MAC4_LAUNCH
  RS_HIP(hipGetLastError());
}

__global__ void __launch_bounds__(256) fill_value_kernel(uint64_t *p, size_t n, uint64_t v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
// [T][L] values -> [T][L][N] ring elements with the value in every slot
__global__ void __launch_bounds__(256) broadcast_rows_kernel(const uint64_t *__restrict__ vals, uint64_t *__restrict__ out, size_t total, int N) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) out[i] = vals[i / (size_t)N];
}
void batch_encode_run(rs_ctx *ctx, const uint64_t *d_rings, uint64_t *d_plain, size_t count, hipStream_t st);

int g_mac_ablate = 0;
int g_mac_ct_temporal = 0;  // tuning knob "mac_ct_temporal": mac_kernel_v3 reads the ciphertext words with temporal loads (A/B: does the second group's read of a shared key vector find them in L2 more often?)
int g_mac_share_keys = 1;  // tuning knob "mac_share_keys": two key vectors share the plaintext spectrum (mac_kernel_v4; 8192 and 16384 points)
int g_msm_host_tile = 1024;  // tuning knob "msm_host_tile": terms per staging buffer of a host-resident key
int g_msm_c_mib = 2048;       // tuning knob "msm_c_mib": workspace of the centred plaintext rows of one term tile
int g_mac_chunk_units = 768;  // tuning knob "mac_chunk_units": (limb, prime, chunk) units per MAC launch (term chunks = units / (L K))
int g_plain_variant = 1;  // 1: plain_center_wide_kernel at N_enc = 8192; 0: plain_center_kernel
int g_mac_variant = 5;  // 6: as 5, one-key launches at N_enc = 8192 in the 512-thread shape (mac_kernel_v4<13, ., 1>: four waves per SIMD); 5: half-spectrum wide kernel at N_enc = 8192 (else as 3); 3: streaming kernel, 1024-thread shape at N_enc = 8192 (else as 2); 2: 512-thread streaming kernel; 1: generic kernel

// Core grouped MSM.  addends: optional per-output (n_crs * n_groups) device pointers to encoding
// elements added to the result (pk.alpha / pk.beta of groth16.tcc:95,103).
// crs_window != 0: every CRS vector is stored as `crs_window` consecutive elements and logical
// element t lives at index t % crs_window (tiled keys, see ringsnark_amd.h); tiles never straddle the wrap.
template <class M>
static void msm_run_arith(rs_ctx *ctx, const uint64_t *const *d_crs, int n_crs, size_t crs_len, const rs_msm_vec *vecs, int n_vecs,
                          int n_groups, uint64_t *d_out, const uint64_t *const *addends, size_t *h_used, hipStream_t st,
                          size_t crs_window, const MsmLin *lin, bool crs_on_host) {
  using Lift = typename ArithOf<M>::Lift;
  constexpr bool FP = std::is_same<M, Mod>::value;
  RS_REQUIRE(n_crs >= 1 && n_crs <= 2, "n_crs must be 1 or 2");
  RS_REQUIRE(n_groups >= 1 && n_groups <= MAX_GROUPS, "too many groups");
  RS_REQUIRE(n_crs * n_groups <= 12, "too many outputs");
  const int L = ctx->L, K = ctx->K, n = ctx->N_enc;
  const size_t enc_words = ctx->enc_words();
  MsmScratch &sc = scratch_for(ctx);

  // group bookkeeping
  PlainArgs pa;
  memset(&pa, 0, sizeof(pa));
  size_t Tmax = 0;
  std::vector<size_t> group_T(n_groups, 0);
  std::vector<unsigned *> nz_ptr(n_vecs, nullptr);
  size_t nz_total = 0, kinds_total = 0;
  for (int v = 0; v < n_vecs; v++) {
    RS_REQUIRE(vecs[v].group >= 0 && vecs[v].group < n_groups, "group index out of range");
    RS_REQUIRE(vecs[v].T <= crs_len, "coefficient vector longer than the CRS vector");
    RS_REQUIRE(vecs[v].d_coeff || vecs[v].T == 0, "null coefficient vector");
    RS_REQUIRE(vecs[v].slot_const == 0 || vecs[v].slot_const == 1, "rs_msm_vec::slot_const must be 0 or 1 (zero-initialise the struct)");
    Tmax = std::max(Tmax, vecs[v].T);
    group_T[vecs[v].group] = std::max(group_T[vecs[v].group], vecs[v].T);
    if (h_used) nz_total += vecs[v].T;
    if (vecs[v].h_kinds) kinds_total += vecs[v].T;
  }
  unsigned *d_nz = nullptr;
  uint8_t *d_kinds = nullptr;
  if (nz_total) {
    d_nz = (unsigned *)ws_get(ctx, 2, nz_total * sizeof(unsigned));
    RS_HIP(hipMemsetAsync(d_nz, 0, nz_total * sizeof(unsigned), st));
  }
  if (kinds_total) d_kinds = (uint8_t *)ws_get(ctx, 3, kinds_total);
  // Slot-constant vectors (rs_msm_vec::slot_const: [T][L] values).  The generic plaintext kernel multiplies the value into
  // the plaintext of (1, ..., 1); the wide 8192-point kernel has no such path: there the rows are expanded into a workspace.
  bool any_sc = false;
  for (int v = 0; v < n_vecs; v++) any_sc = any_sc || vecs[v].slot_const;
  const bool sc_native = !(std::is_same<M, Mod>::value && g_plain_variant == 1 && n == 8192 && (ctx->N == 8192 || ctx->N == 4096));
  std::vector<const uint64_t *> coeff_ptr(n_vecs);
  for (int v = 0; v < n_vecs; v++) coeff_ptr[v] = vecs[v].d_coeff;
  if (any_sc && sc_native && !sc.d_ones_plain) {
    const size_t rw = ctx->ring_words();
    uint64_t *ones = (uint64_t *)ws_get(ctx, 14, rw * sizeof(uint64_t));
    hipLaunchKernelGGL(fill_value_kernel, dim3((unsigned)((rw + 255) / 256)), dim3(256), 0, st, ones, rw, 1ull);
    RS_HIP(hipMalloc(&sc.d_ones_plain, (size_t)L * n * sizeof(uint64_t)));
    batch_encode_run(ctx, ones, sc.d_ones_plain, 1, st);  // sc is the context's entry itself (scratch_for returns a reference)
    // the table is cached for the life of the context and read by later calls on ANY stream: it must be complete
    // before its pointer is visible to them (once per context)
    RS_HIP(hipStreamSynchronize(st));
  }
  if (any_sc && !sc_native) {
    size_t rows = 0;
    for (int v = 0; v < n_vecs; v++) rows += vecs[v].slot_const ? vecs[v].T : 0;
    uint64_t *buf = (uint64_t *)ws_get(ctx, 14, std::max<size_t>(1, rows) * ctx->ring_words() * sizeof(uint64_t));
    size_t at = 0;
    for (int v = 0; v < n_vecs; v++)
      if (vecs[v].slot_const && vecs[v].T) {
        const size_t total = vecs[v].T * ctx->ring_words();
        hipLaunchKernelGGL(broadcast_rows_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 8192)), dim3(256), 0, st, vecs[v].d_coeff,
                           buf + at, total, ctx->N);
        coeff_ptr[v] = buf + at;
        at += total;
      }
    RS_HIP(hipGetLastError());
  }
  pa.ones_plain = sc.d_ones_plain;
  {
    size_t nzo = 0, ko = 0;
    for (int v = 0; v < n_vecs; v++) {
      PlainGroup &G = pa.g[vecs[v].group];
      RS_REQUIRE(G.n < MAX_GROUP_VECS, "too many vectors in one group");
      G.coeff[G.n] = coeff_ptr[v];
      G.slot_const[G.n] = (vecs[v].slot_const && sc_native) ? 1 : 0;
      G.T[G.n] = vecs[v].T;
      if (h_used) {
        nz_ptr[v] = d_nz + nzo;
        G.nz[G.n] = nz_ptr[v];
        nzo += vecs[v].T;
      }
      if (vecs[v].h_kinds) {
        RS_HIP(hipMemcpyAsync(d_kinds + ko, vecs[v].h_kinds, vecs[v].T, hipMemcpyHostToDevice, st));
        G.kinds[G.n] = d_kinds + ko;
        ko += vecs[v].T;
      }
      G.n++;
    }
  }

  bool has_lin = false;
  if (lin)
    for (int g = 0; g < n_groups; g++)
      if (lin[g].count > 0) {
        RS_REQUIRE(lin[g].T <= crs_len && !h_used, "linear-form vector: too long, or used-term counts requested");
        pa.g[g].lin = lin[g];
        has_lin = true;
        Tmax = std::max<size_t>(Tmax, (size_t)lin[g].T);
        group_T[g] = std::max<size_t>(group_T[g], (size_t)lin[g].T);
      }
  const int n_sets = n_crs * n_groups;
  // tiling: C workspace <= ~2 GiB (knob msm_c_mib)
  const size_t c_bytes_per_term = (size_t)n_groups * L * n * sizeof(double);
  size_t tile_terms = std::max<size_t>(1, std::min<size_t>(Tmax, ((size_t)g_msm_c_mib << 20) / c_bytes_per_term));
  if (crs_window) {
    size_t p2 = 1;
    while (p2 * 2 <= tile_terms) p2 *= 2;
    tile_terms = std::min(p2, crs_window);
    RS_REQUIRE(crs_window % tile_terms == 0, "crs_window must be a multiple of the term tile (use a power of two)");
  }
  // Host-resident key (crs_on_host: d_crs are HOST pointers -- a proving key larger than HBM, e.g. the 384 GiB key of the
  // 2^16-constraint headline on one GPU): the term tiles are streamed through two device staging buffers; the copy of
  // tile k+1 runs on its own stream under the kernels of tile k (pinned host memory, rs_host_alloc, for real overlap).
  uint64_t *stage = nullptr;
  size_t stage_words = 0;
  if (crs_on_host) {
    tile_terms = std::min<size_t>(tile_terms, (size_t)std::max(1, g_msm_host_tile));
    if (crs_window) {
      size_t p2 = 1;
      while (p2 * 2 <= tile_terms) p2 *= 2;
      tile_terms = std::min(p2, crs_window);  // may no longer divide the window: issue_copy splits a tile at the wrap
    }
    stage_words = tile_terms * enc_words;
    stage = (uint64_t *)ws_get(ctx, 7, (size_t)2 * n_crs * stage_words * sizeof(uint64_t));
    if (!sc.copy_stream) {
      RS_HIP(hipStreamCreateWithFlags(&sc.copy_stream, hipStreamNonBlocking));
      for (int b = 0; b < 2; b++) {
        RS_HIP(hipEventCreateWithFlags(&sc.ev_copied[b], hipEventDisableTiming));
        RS_HIP(hipEventCreateWithFlags(&sc.ev_freed[b], hipEventDisableTiming));
      }
    }
    // the staging buffers may still be read by an earlier call on another stream: order the copy stream after `st`
    RS_HIP(hipEventRecord(sc.ev_freed[0], st));
    RS_HIP(hipEventRecord(sc.ev_freed[1], st));
  }
  auto stage_at = [&](int buf, int c) { return stage + ((size_t)buf * n_crs + c) * stage_words; };
  auto issue_copy = [&](int tile, size_t t0) {  // tile -> staging buffer tile % 2, on the copy stream
    const int buf = tile & 1;
    const size_t tt = std::min(tile_terms, Tmax - t0);
    RS_HIP(hipStreamWaitEvent(sc.copy_stream, sc.ev_freed[buf], 0));
    // the staging buffer linearises a tile that straddles the wrap of a windowed key (a window of 1536 elements with
    // 1024-element staging tiles: the tile at 1024 is elements 1024..1535, then 0..511)
    const size_t o0 = crs_window ? t0 % crs_window : t0;
    const size_t first = crs_window ? std::min(tt, crs_window - o0) : tt;
    for (int c = 0; c < n_crs; c++) {
      RS_HIP(hipMemcpyAsync(stage_at(buf, c), d_crs[c] + o0 * enc_words, first * enc_words * sizeof(uint64_t), hipMemcpyHostToDevice,
                            sc.copy_stream));
      for (size_t done = first; done < tt;) {  // wrapped remainder (several rounds if the tile exceeds the window)
        const size_t part = std::min(tt - done, crs_window);
        RS_HIP(hipMemcpyAsync(stage_at(buf, c) + done * enc_words, d_crs[c], part * enc_words * sizeof(uint64_t), hipMemcpyHostToDevice,
                              sc.copy_stream));
        done += part;
      }
    }
    RS_HIP(hipEventRecord(sc.ev_copied[buf], sc.copy_stream));
  };
  int cur_tile = 0;
  auto crs_at = [&](int c, size_t t0) -> const uint64_t * {
    if (crs_on_host) return stage_at(cur_tile & 1, c);
    return d_crs[c] + (crs_window ? t0 % crs_window : t0) * enc_words;
  };
  // mac_kernel_v4 (two key vectors) runs ONE workgroup per CU: half the workgroup slots, half the chunks (measured at the
  // configs[3] shape: 111 -> 106 ms)
  const int chunk_units = (n_crs == 2 && g_mac_share_keys && g_mac_variant >= 5 && (n == 8192 || n == 16384)) ? (g_mac_chunk_units + 1) / 2
                                                                                                              : g_mac_chunk_units;
  int n_chunks = (int)std::min<size_t>(tile_terms, (size_t)std::max(1, (chunk_units + L * K - 1) / (L * K)));
  {  // the (chunk, limb) rows of a launch are dealt to the 8 XCDs: a row count that is not a multiple of 8 leaves slots idle
    int step = 8;
    while (step > 1 && (L * (step / 2)) % 8 == 0) step /= 2;  // smallest chunk-count granule with L * granule = 0 mod 8
    if (n_chunks >= step && (size_t)((n_chunks + step - 1) / step * step) <= tile_terms) n_chunks = (n_chunks + step - 1) / step * step;
  }
  if (Tmax == 0) n_chunks = 1;
  Lift *d_C = (Lift *)ws_get(ctx, 0, std::max<size_t>(256, tile_terms * c_bytes_per_term));
  uint64_t *d_partial = (uint64_t *)ws_get(ctx, 1, (size_t)n_chunks * n_sets * enc_words * sizeof(uint64_t));
  const size_t lds = padded_len((size_t)n) * sizeof(double);
  const int thr = tile_threads(ctx->logN_enc);
  const bool big = n > 8192;  // one accumulator set per MAC launch
  const bool plain16 = n >= 2048;  // 16 coefficients per thread, wave-private inverse transform
  const int plain_thr = plain16 ? n / 16 : thr;
  const bool plain13 = false;  // measured: no gain from a compile-time length here
  (void)plain13;
  bool v3 = false, plain_wide = false, hybrid = false;
  if constexpr (FP) {
    v3 = g_mac_variant >= 5 && (n == 8192 || n == 16384);  // 6: one-key launches in the 512-thread shape (mac_kernel_v4<13, ., 1>)
    plain_wide = g_plain_variant == 1 && n == 8192 && (ctx->N == 8192 || ctx->N == 4096);
  } else {
    // Hybrid context: ring primes beyond 2^50 (SEAL's 54-bit BFVDefault(2048) prime of the reference's logistic-regression
    // benchmark) on the integer arithmetic, data primes below 2^50: the plaintext row is produced by the integer kernel and
    // handed, as exact doubles, to the FP64 multiply-accumulate of the data primes.  Needs |sum of a group's lifts| < 2^53.
    uint64_t maxq = 0;
    for (int i = 0; i < L; i++) maxq = std::max(maxq, ctx->q[i]);
    int max_vecs = 1;
    for (int g = 0; g < n_groups; g++) max_vecs = std::max(max_vecs, pa.g[g].n);
    hybrid = ctx->hybrid && g_mac_variant >= 5 && (n == 8192 || n == 16384) &&
             (double)max_vecs * (0.5 * (double)maxq + 1.0) < 9007199254740992.0;
    v3 = hybrid;
  }
  RS_REQUIRE(!has_lin || plain_wide, "linear-form vectors need the wide plaintext kernel");
  const bool paired = v3 && plain_wide;  // row layout of this call: written by the plaintext kernel, read by the MAC
  bool multi = false;
  for (int g = 0; g < n_groups; g++) multi = multi || pa.g[g].n > 1;
  if (plain16)
    set_max_dyn_lds((const void *)plain_center_kernel<16, 0, M>, (int)lds);
  else
    set_max_dyn_lds((const void *)plain_center_kernel<8, 0, M>, (int)lds);

  // rooflines: per (term, limb) a plaintext row costs one inverse transform of
  // length n, the batching scatter and the centred lift; per (term, limb, prime) the MAC reads two
  // ciphertext polynomials and one plaintext row, runs one forward transform and 2n multiply-adds
  const double nd = (double)n, logn_d = (double)ctx->logN_enc;
  if (Tmax == 0) RS_HIP(hipMemsetAsync(d_partial, 0, (size_t)n_chunks * n_sets * enc_words * sizeof(uint64_t), st));

  int tile_idx = 0;
  if (crs_on_host && Tmax) issue_copy(0, 0);
  for (size_t t0 = 0; t0 < Tmax; t0 += tile_terms, tile_idx++) {
    const size_t tt = std::min(tile_terms, Tmax - t0);
    cur_tile = tile_idx;
    if (crs_on_host) {
      if (t0 + tile_terms < Tmax) issue_copy(tile_idx + 1, t0 + tile_terms);  // under this tile's kernels
      RS_HIP(hipStreamWaitEvent(st, sc.ev_copied[tile_idx & 1], 0));
    }
    double rows_in = 0;  // coefficient rows (term, limb) read by this tile
    for (int v = 0; v < n_vecs; v++) rows_in += (double)(vecs[v].T > t0 ? std::min(tt, vecs[v].T - t0) : 0) * L;
    {
    ProfScope prof_plain(ctx, st, plain_wide ? "plain_center_wide_kernel" : "plain_center_kernel", rows_in * (double)ctx->N * 8.0 + (double)tt * L * n_groups * nd * 8.0,
                         rows_in * (ntt_fp64(nd, logn_d) + 14.0 * nd));
    if (plain_wide) {
      if constexpr (FP) {
        const int wl = (int)WideShape<13>::LDS_BYTES;
        const unsigned long long items = (unsigned long long)tt * n_groups;
        const unsigned slots = (unsigned)std::max<unsigned long long>(1, std::min<unsigned long long>(items, (512 + L - 1) / L));
        const dim3 grid(slots * (unsigned)L);
        PlainTwPtrs twp;
        memset(&twp, 0, sizeof(twp));
        for (int i = 0; i < L; i++) twp.itw[i] = ctx->plain[i].d_itw;
#define RS_PLAIN_WIDE_NE(MULTI_, PAIRED_, NE_, LIN_)                                                                          \
  do {                                                                                                                        \
    set_max_dyn_lds((const void *)plain_center_wide_kernel<MULTI_, PAIRED_, NE_, LIN_>, \
                               wl);                                              \
    hipLaunchKernelGGL((plain_center_wide_kernel<MULTI_, PAIRED_, NE_, LIN_>), grid, dim3(256), wl, st, pa,                   \
                       reinterpret_cast<double *>(d_C), (unsigned long long)t0, (unsigned long long)tile_terms,               \
                       (unsigned long long)tt, n_groups, ctx->N, L, ctx->d_index_map, sc.plain<Mod>(), twp);                  \
  } while (0)
#define RS_PLAIN_WIDE(MULTI_, PAIRED_)                                  \
  do {                                                                  \
    if (ctx->N == 8192) {                                               \
      if (has_lin) RS_PLAIN_WIDE_NE(MULTI_, PAIRED_, 16, true);         \
      else RS_PLAIN_WIDE_NE(MULTI_, PAIRED_, 16, false);                \
    } else {                                                            \
      if (has_lin) RS_PLAIN_WIDE_NE(MULTI_, PAIRED_, 8, true);          \
      else RS_PLAIN_WIDE_NE(MULTI_, PAIRED_, 8, false);                 \
    }                                                                   \
  } while (0)
        if (multi && paired) RS_PLAIN_WIDE(true, true);
        else if (multi) RS_PLAIN_WIDE(true, false);
        else if (paired) RS_PLAIN_WIDE(false, true);
        else RS_PLAIN_WIDE(false, false);
#undef RS_PLAIN_WIDE_NE
#undef RS_PLAIN_WIDE
      }
    } else if (plain16)
      hipLaunchKernelGGL((plain_center_kernel<16, 0, M>), dim3((unsigned)tt, L, n_groups), dim3(plain_thr), lds, st, pa, d_C,
                         (unsigned long long)t0, (unsigned long long)tile_terms, ctx->N, L, ctx->logN_enc,
                         ctx->d_index_map, sc.template plain<M>(), hybrid ? 1 : 0);
    else
      hipLaunchKernelGGL((plain_center_kernel<8, 0, M>), dim3((unsigned)tt, L, n_groups), dim3(thr), lds, st, pa, d_C,
                         (unsigned long long)t0, (unsigned long long)tile_terms, ctx->N, L, ctx->logN_enc,
                         ctx->d_index_map, sc.template plain<M>(), hybrid ? 1 : 0);
    }
    RS_HIP(hipGetLastError());
    MacArgs base;
    memset(&base, 0, sizeof(base));
    base.partial = d_partial;
    base.n_sets_total = n_sets;
    base.tile_terms = tt;
    base.n_chunks = n_chunks;
    base.terms_per_chunk = (int)((tt + n_chunks - 1) / n_chunks);
    base.accumulate = tile_idx > 0;
    uint64_t maxQ = 0;
    for (int j = 0; j < K; j++) maxQ = std::max(maxQ, ctx->Q[j]);
    base.acc_period = (int)std::max(1.0, std::floor(4503599627370496.0 / (0.8 * (double)maxQ)) - 1.0);
    base.acc_period = std::min(base.acc_period, 1 << 20);
    base.reduce_u = maxQ >= (1ull << 45);
    auto group_terms = [&](int g) -> unsigned long long {
      return group_T[g] > t0 ? (unsigned long long)std::min(tt, group_T[g] - t0) : 0ull;
    };
    auto Cptr = [&](int g) { return d_C + (size_t)g * tile_terms * L * n; };
    auto run = [&](int NG, int NC, const int *gs, const int *cs) {
      MacArgs a = base;
      for (int g = 0; g < NG; g++) {
        a.C[g] = Cptr(gs[g]);
        a.terms[g] = group_terms(gs[g]);
      }
      for (int c = 0; c < NC; c++) a.crs[c] = crs_at(cs[c], t0);
      for (int c = 0; c < NC; c++)
        for (int g = 0; g < NG; g++) a.set_index[c * NG + g] = cs[c] * n_groups + gs[g];
      double units = 0;  // (term, limb, prime) transforms
      for (int g = 0; g < NG; g++) units += (double)a.terms[g] * L * K;
      ProfScope prof(ctx, st, "mac_kernel",
                     (double)tt * NC * (double)enc_words * 8.0 + units * nd * 8.0 / K + (double)NG * NC * (double)enc_words * 8.0,
                     units * (ntt_fp64(nd, logn_d) + NC * 15.0 * nd));
      if (big)
        launch_mac<1, 1, 8, M>(ctx, a, sc, st);
      else if (NG == 2 && NC == 1)
        launch_mac<2, 1, 4, M>(ctx, a, sc, st);
      else if (NG == 1 && NC == 2)
        launch_mac<1, 2, 4, M>(ctx, a, sc, st);
      else
        launch_mac<1, 1, 4, M>(ctx, a, sc, st);
    };
    // streaming kernel, one accumulator set (CRS vector c, group g) per launch.  With two CRS
    // vectors (Rinocchio's s_pows / alpha_s_pows) the plaintext transform is repeated per vector:
    // measured faster than the generic kernel that shares it (g_mac_variant == 2: generic for n_crs == 2).
    const bool v2 = FP && g_mac_variant >= 2 && (n_crs == 1 || g_mac_variant != 2) && n >= 2048 && n <= 8192;
    if (v3) {
      {
        // bound of a plaintext row: the sum of the centred lifts of the group's vectors
        uint64_t maxq = 0;
        for (int i = 0; i < L; i++) maxq = std::max(maxq, ctx->q[i]);
        const double b0 = 0.5 * (double)maxq * MAX_GROUP_VECS + (double)MAX_GROUP_VECS;
        if (n_crs == 2 && g_mac_share_keys) {
          // Rinocchio's ten inner products (rinocchio.tcc:106-160): every vector against both key vectors, one transform
          for (int g0 = 0; g0 < n_groups; g0 += RS_MAC4_GROUPS) {
            const int ng = std::min(RS_MAC4_GROUPS, n_groups - g0);
            MacArgs4 a4;
            memset(&a4, 0, sizeof(a4));
            unsigned long long tmax = 0;
            double terms = 0;
            for (int gi = 0; gi < ng; gi++) {
              a4.C[gi] = reinterpret_cast<const double *>(Cptr(g0 + gi));
              a4.terms[gi] = group_terms(g0 + gi);
              for (int c = 0; c < 2; c++) a4.partial[c][gi] = d_partial + (size_t)(c * n_groups + g0 + gi) * enc_words;
              tmax = std::max(tmax, a4.terms[gi]);
              terms += (double)a4.terms[gi];
            }
            a4.crs[0] = crs_at(0, t0);
            a4.crs[1] = crs_at(1, t0);
            a4.part_stride = (size_t)n_sets * enc_words;
            a4.n_groups = ng;
            a4.n_chunks = base.n_chunks;
            a4.terms_per_chunk = base.terms_per_chunk;
            a4.accumulate = base.accumulate;
            a4.acc_period = base.acc_period;
            for (int jj = 0; jj < K; jj++) {
              double end = 0;
              a4.red_mask[jj] = fwd_reduce_mask_from(ctx->Q[jj], ctx->logN_enc, b0, &end);
              if (end > 562949953421312.0) a4.reduce_u = 1;
            }
            ProfScope prof(ctx, st, n == 16384 ? "mac_kernel_v4<14, false>" : (paired ? "mac_kernel_v4<13, true>" : "mac_kernel_v4<13, false>"), (double)tmax * 2.0 * (double)enc_words * 8.0 + terms * (double)L * nd * 8.0 + 2.0 * ng * (double)enc_words * 8.0,
                           terms * L * K * (ntt_fp64(nd, logn_d) + 8.0 * nd / 2.0 + 2.0 * 15.0 * nd));
            launch_mac_v4(ctx, a4, paired, sc, st);
          }
        } else if (g_mac_variant == 6 && n == 8192) {
          // one key vector per launch in the 512-thread shape of mac_kernel_v4 (four waves per SIMD), two groups that read it
          // as neighbouring workgroups of an XCD (A and B against s_pows, groth16.tcc:89-103)
          for (int c = 0; c < n_crs; c++)
            for (int g0 = 0; g0 < n_groups; g0 += 2) {
              const int ng = std::min(2, n_groups - g0);
              MacArgs4 a4;
              memset(&a4, 0, sizeof(a4));
              unsigned long long tmax = 0;
              double terms = 0;
              for (int gi = 0; gi < ng; gi++) {
                a4.C[gi] = reinterpret_cast<const double *>(Cptr(g0 + gi));
                a4.terms[gi] = group_terms(g0 + gi);
                a4.partial[0][gi] = d_partial + (size_t)(c * n_groups + g0 + gi) * enc_words;
                tmax = std::max(tmax, a4.terms[gi]);
                terms += (double)a4.terms[gi];
              }
              a4.crs[0] = crs_at(c, t0);
              a4.part_stride = (size_t)n_sets * enc_words;
              a4.n_groups = ng;
              a4.n_chunks = base.n_chunks;
              a4.terms_per_chunk = base.terms_per_chunk;
              a4.accumulate = base.accumulate;
              a4.acc_period = base.acc_period;
              for (int jj = 0; jj < K; jj++) {
                double end = 0;
                a4.red_mask[jj] = fwd_reduce_mask_from(ctx->Q[jj], ctx->logN_enc, b0, &end);
                if (end > 562949953421312.0) a4.reduce_u = 1;
              }
              ProfScope prof(ctx, st, paired ? "mac_kernel_v4<13, true, 1>" : "mac_kernel_v4<13, false, 1>",
                             (double)tmax * (double)enc_words * 8.0 + terms * (double)L * nd * 8.0 + ng * (double)enc_words * 8.0,
                             terms * L * K * (ntt_fp64(nd, logn_d) + 8.0 * nd / 2.0 + 15.0 * nd));
              launch_mac_v4_one(ctx, a4, paired, sc, st);
            }
        } else
        // chunks: two workgroups per CU in one wave of workgroups (512), shared by the groups of a launch
        for (int c = 0; c < n_crs; c++)
          for (int g0 = 0; g0 < n_groups; g0 += 2) {
            const int ng = std::min(2, n_groups - g0);
            MacArgs3 a3;
            memset(&a3, 0, sizeof(a3));
            unsigned long long tmax = 0;
            for (int gi = 0; gi < ng; gi++) {
              a3.C[gi] = reinterpret_cast<const double *>(Cptr(g0 + gi));
              a3.terms[gi] = group_terms(g0 + gi);
              a3.partial[gi] = d_partial + (size_t)(c * n_groups + g0 + gi) * enc_words;
              tmax = std::max(tmax, a3.terms[gi]);
            }
            a3.crs = crs_at(c, t0);
            a3.part_stride = (size_t)n_sets * enc_words;
            a3.n_groups = ng;
            a3.paired = paired;
            a3.ct_temporal = g_mac_ct_temporal;
            a3.n_chunks = base.n_chunks;
            a3.terms_per_chunk = base.terms_per_chunk;
            a3.accumulate = base.accumulate;
            a3.acc_period = base.acc_period;
            a3.reduce_u = 0;
            for (int jj = 0; jj < K; jj++) {
              double end = 0;
              a3.red_mask[jj] = fwd_reduce_mask_from(ctx->Q[jj], ctx->logN_enc, b0, &end);
              if (end > 562949953421312.0) a3.reduce_u = 1;  // a spectrum value times a canonical ciphertext word: |u| <= 2^49
            }
            double terms = 0;
            for (int gi = 0; gi < ng; gi++) terms += (double)a3.terms[gi];
            // ciphertext words once per launch (the second group's read is served on-die), every plaintext row once,
            // the accumulator sets written once
            ProfScope prof(ctx, st, n == 16384 ? "mac_kernel_v3<false, 14>" : "mac_kernel_v3", (double)tmax * (double)enc_words * 8.0 + terms * (double)L * nd * 8.0 + ng * (double)enc_words * 8.0,
                           terms * L * K * (ntt_fp64(nd, logn_d) + 8.0 * nd / 2.0 + 15.0 * nd));
            launch_mac_v3(ctx, a3, sc, st);
          }
      }
    } else if (v2) {
      for (int c = 0; c < n_crs; c++)
      for (int g = 0; g < n_groups; g++) {
        MacArgs2 a2;
        a2.C = reinterpret_cast<const double *>(Cptr(g));
        a2.terms = group_terms(g);
        a2.crs = crs_at(c, t0);
        a2.partial = d_partial + (size_t)(c * n_groups + g) * enc_words;
        a2.part_stride = (size_t)n_sets * enc_words;
        a2.terms_per_chunk = base.terms_per_chunk;
        a2.n_chunks = base.n_chunks;
        a2.accumulate = base.accumulate;
        a2.acc_period = base.acc_period;
        a2.reduce_u = base.reduce_u;
        // every ciphertext word of the group's terms once, every plaintext row once (shared by the K
        // prime workgroups through L2), the accumulator set written once
        const double terms = (double)a2.terms;
        ProfScope prof(ctx, st, (g_mac_variant == 3 && ctx->logN_enc == 13) ? "mac_kernel_v2<1024, 13, 0>" : "mac_kernel_v2", terms * ((double)enc_words * 8.0 + (double)L * nd * 8.0) + (double)enc_words * 8.0,
                       terms * L * K * (ntt_fp64(nd, logn_d) + 15.0 * nd));
        launch_mac_v2(ctx, a2, sc, st);
      }
    } else if (big) {  // one (crs, group) pair per launch
      for (int c = 0; c < n_crs; c++)
        for (int g = 0; g < n_groups; g++) {
          const int gs[1] = {g}, cs1[1] = {c};
          run(1, 1, gs, cs1);
        }
    } else if (n_crs == 1) {  // share each ciphertext read between two groups
      int g = 0;
      const int c0[1] = {0};
      for (; g + 1 < n_groups; g += 2) {
        const int gs[2] = {g, g + 1};
        run(2, 1, gs, c0);
      }
      if (g < n_groups) {
        const int gs[1] = {g};
        run(1, 1, gs, c0);
      }
    } else {  // two CRS vectors: share each plaintext NTT between them
      const int cs[2] = {0, 1};
      for (int g = 0; g < n_groups; g++) {
        const int gs[1] = {g};
        run(1, 2, gs, cs);
      }
    }
    if (crs_on_host) RS_HIP(hipEventRecord(sc.ev_freed[tile_idx & 1], st));  // its readers are enqueued: the buffer may be refilled after them
  }
  ReduceArgs ra;
  memset(&ra, 0, sizeof(ra));
  if (addends)
    for (int s_ = 0; s_ < n_sets; s_++) ra.addend[s_] = addends[s_];
  {
    const size_t total = (size_t)n_sets * enc_words;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(reduce_kernel, dim3(blocks), dim3(256), 0, st, d_partial, d_out, ra, n_chunks, n_sets, enc_words, n,
                       K, sc.d_Qint);
    RS_HIP(hipGetLastError());
  }
  if (h_used) {
    std::vector<unsigned> h(nz_total);
    if (nz_total) RS_HIP(hipMemcpyAsync(h.data(), d_nz, nz_total * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    RS_HIP(hipStreamSynchronize(st));
    size_t off = 0;
    for (int v = 0; v < n_vecs; v++) {
      size_t cnt = 0;
      for (size_t t = 0; t < vecs[v].T; t++) cnt += h[off + t] != 0;
      h_used[v] = cnt;
      off += vecs[v].T;
    }
  }
}

void msm_run(rs_ctx *ctx, const uint64_t *const *d_crs, int n_crs, size_t crs_len, const rs_msm_vec *vecs, int n_vecs,
             int n_groups, uint64_t *d_out, const uint64_t *const *addends, size_t *h_used, hipStream_t st,
             size_t crs_window, const MsmLin *lin = nullptr, bool crs_on_host = false) {
  RS_DISPATCH_ARITH(ctx, (msm_run_arith<Mod>(ctx, d_crs, n_crs, crs_len, vecs, n_vecs, n_groups, d_out, addends, h_used, st, crs_window, lin, crs_on_host)),
                    (msm_run_arith<ModI>(ctx, d_crs, n_crs, crs_len, vecs, n_vecs, n_groups, d_out, addends, h_used, st, crs_window, lin, crs_on_host)));
}
// can a call with linear-form vectors be served? (FP64 context on the wide plaintext kernel)
bool msm_supports_lin(const rs_ctx *ctx) {
  return !ctx->use_int && g_plain_variant == 1 && ctx->N_enc == 8192 && (ctx->N == 8192 || ctx->N == 4096);
}
void batch_encode_run(rs_ctx *ctx, const uint64_t *d_rings, uint64_t *d_plain, size_t count, hipStream_t st) {
  if (!count) return;
  MsmScratch &sc = scratch_for(ctx);
  const size_t lds = padded_len((size_t)ctx->N_enc) * sizeof(double);
  if (ctx->use_int) {
    set_max_dyn_lds((const void *)batch_encode_kernel<ModI>, (int)lds);
    hipLaunchKernelGGL(batch_encode_kernel<ModI>, dim3((unsigned)count, ctx->L), dim3(tile_threads(ctx->logN_enc)), lds, st, d_rings, d_plain,
                       ctx->N, ctx->L, ctx->logN_enc, ctx->d_index_map, sc.plain<ModI>());
  } else {
    set_max_dyn_lds((const void *)batch_encode_kernel<Mod>, (int)lds);
    hipLaunchKernelGGL(batch_encode_kernel<Mod>, dim3((unsigned)count, ctx->L), dim3(tile_threads(ctx->logN_enc)), lds, st, d_rings, d_plain,
                       ctx->N, ctx->L, ctx->logN_enc, ctx->d_index_map, sc.plain<Mod>());
  }
  RS_HIP(hipGetLastError());
}

void enc_add_run(rs_ctx *ctx, uint64_t *dst, const uint64_t *x, const uint64_t *y, size_t count, hipStream_t st) {
  const size_t words = count * ctx->enc_words();
  if (!words) return;
  MsmScratch &sc = scratch_for(ctx);
  const size_t pairs = words / 2;  // N_enc is even: whole 16-byte words
  const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((pairs + 2047) / 2048, 256 * 8));
  if (pairs * 16 >= ((size_t)64 << 20))  // non-temporal accesses for operands beyond the caches
    hipLaunchKernelGGL(enc_add_kernel<true>, dim3(blocks), dim3(256), 0, st, dst, x, y, pairs, ctx->logN_enc, ctx->K, sc.d_Qint);
  else
    hipLaunchKernelGGL(enc_add_kernel<false>, dim3(blocks), dim3(256), 0, st, dst, x, y, pairs, ctx->logN_enc, ctx->K, sc.d_Qint);
  RS_HIP(hipGetLastError());
}

}  // namespace rs

using namespace rs;

extern "C" {

int rs_batch_encode(rs_ctx *ctx, const uint64_t *d_rings, uint64_t *d_plain, size_t count, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_rings && d_plain, "null argument");
  batch_encode_run(ctx, d_rings, d_plain, count, S(stream));
  RS_API_END
}

int rs_msm(rs_ctx *ctx, const uint64_t *const *d_crs, int n_crs, size_t crs_len, size_t crs_window, const rs_msm_vec *vecs,
           int n_vecs, int n_groups, uint64_t *d_out, size_t *h_used, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_crs && vecs && d_out && n_vecs >= 1, "null argument");
  WsScope ws_scope(ctx, S(stream));
  msm_run(ctx, d_crs, n_crs, crs_len, vecs, n_vecs, n_groups, d_out, nullptr, h_used, S(stream), crs_window);
  RS_API_END
}

int rs_msm_hostkey(rs_ctx *ctx, const uint64_t *const *h_crs, int n_crs, size_t crs_len, size_t crs_window, const rs_msm_vec *vecs,
                   int n_vecs, int n_groups, uint64_t *d_out, size_t *h_used, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && h_crs && vecs && d_out && n_vecs >= 1, "null argument");
  WsScope ws_scope(ctx, S(stream));
  msm_run(ctx, h_crs, n_crs, crs_len, vecs, n_vecs, n_groups, d_out, nullptr, h_used, S(stream), crs_window, nullptr, true);
  RS_HIP(hipStreamSynchronize(S(stream)));  // the caller may release or rewrite the host key on return
  RS_API_END
}

int rs_host_alloc(rs_ctx *ctx, size_t bytes, void **h_ptr) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(h_ptr && bytes, "null argument");
  RS_HIP(hipHostMalloc(h_ptr, bytes, hipHostMallocDefault));
  RS_API_END
}
int rs_host_free(rs_ctx *ctx, void *h_ptr) {
  RS_API_BEGIN_CTX(ctx)
  if (h_ptr) RS_HIP(hipHostFree(h_ptr));
  RS_API_END
}

int rs_inner_product(rs_ctx *ctx, const uint64_t *d_encs, const uint64_t *d_rings, const uint8_t *h_kinds, size_t T,
                     uint64_t *d_out, size_t *h_used, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_out && (T == 0 || (d_encs && d_rings)), "null argument");
  WsScope ws_scope(ctx, S(stream));
  rs_msm_vec v{d_rings, h_kinds, T, 0};
  const uint64_t *crs[1] = {d_encs};
  msm_run(ctx, crs, 1, T, &v, 1, 1, d_out, nullptr, h_used, S(stream), 0);
  RS_API_END
}

int rs_enc_mul_ring(rs_ctx *ctx, uint64_t *d_enc, const uint64_t *d_ring, size_t count, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_enc && d_ring, "null argument");
  WsScope ws_scope(ctx, S(stream));
  uint64_t *tmp = (uint64_t *)ws_get(ctx, 4, ctx->enc_words() * sizeof(uint64_t));
  for (size_t k = 0; k < count; k++) {
    rs_msm_vec v{d_ring + k * ctx->ring_words(), nullptr, 1, 0};
    const uint64_t *crs[1] = {d_enc + k * ctx->enc_words()};
    msm_run(ctx, crs, 1, 1, &v, 1, 1, tmp, nullptr, nullptr, S(stream), 0);
    RS_HIP(hipMemcpyAsync(d_enc + k * ctx->enc_words(), tmp, ctx->enc_words() * sizeof(uint64_t),
                          hipMemcpyDeviceToDevice, S(stream)));
  }
  RS_API_END
}

int rs_enc_reduce(rs_ctx *ctx, uint64_t *d_enc, size_t count, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_enc, "null argument");
  const size_t words = count * ctx->enc_words();
  if (words) {
    MsmScratch &sc = scratch_for(ctx);
    const unsigned blocks = (unsigned)std::min<size_t>((words + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(enc_reduce_kernel, dim3(blocks), dim3(256), 0, S(stream), d_enc, words, ctx->N_enc, ctx->K, sc.d_Qint);
    RS_HIP(hipGetLastError());
  }
  RS_API_END
}

int rs_enc_add(rs_ctx *ctx, uint64_t *d_dst, const uint64_t *d_a, const uint64_t *d_b, size_t count, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_dst && d_a && d_b, "null argument");
  enc_add_run(ctx, d_dst, d_a, d_b, count, S(stream));
  RS_API_END
}

}  // extern "C"
