// rs_core.hip -- context, twiddle tables, batched negacyclic NTT (row a4) and the dyadic RingElem
// kernels (rows a1-a3) of SURVEY.md section 8.
#include <algorithm>
#include <cstring>

#include "ntt_core.hpp"
#include "ntt_wide.hpp"
#include "rs_internal.hpp"

namespace rs {

static thread_local std::string g_last_error;
void set_last_error(const std::string &m) { g_last_error = m; }

void *ws_get(rs_ctx *ctx, int slot, size_t bytes) {
  DeviceBuf &b = ctx->ws[slot];
  if (b.used && b.last_stream != ctx->cur_stream) {
    // the previous user ran on another stream: order this call's work after it (device side)
    RS_HIP(hipStreamWaitEvent(ctx->cur_stream, b.last_use, 0));
    b.last_stream = ctx->cur_stream;  // the wait is enqueued; later ws_get calls of this scope need not repeat it
  }
  if (b.bytes < bytes) {
    if (b.p) {
      if (b.used) RS_HIP(hipEventSynchronize(b.last_use));  // nothing may still read the old buffer
      RS_HIP(hipStreamSynchronize(ctx->cur_stream));
      RS_HIP(hipFree(b.p));
    }
    b.p = nullptr;
    b.bytes = 0;
    RS_HIP(hipMalloc(&b.p, bytes));
    b.bytes = bytes;
  }
  ctx->ws_touched |= 1u << slot;
  return b.p;
}
static hipEvent_t prof_event(rs_ctx *ctx) {
  if (!ctx->prof_pool.empty()) {
    hipEvent_t e = ctx->prof_pool.back();
    ctx->prof_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  RS_HIP(hipEventCreate(&e));
  return e;
}
ProfScope::ProfScope(rs_ctx *c, hipStream_t s, const char *name, double alg_bytes, double fp64_ops) : ctx(c), st(s) {
  if (!ctx->profiling) return;
  ProfRec r{name, prof_event(ctx), prof_event(ctx), alg_bytes, fp64_ops};
  RS_HIP(hipEventRecord(r.e0, st));
  ctx->prof.push_back(r);
  idx = (int)ctx->prof.size() - 1;
}
ProfScope::~ProfScope() {
  if (idx >= 0) (void)hipEventRecord(ctx->prof[idx].e1, st);
}
WsScope::~WsScope() {
  for (int k = 0; k < 16; k++) {
    if (!((ctx->ws_touched >> k) & 1u)) continue;
    DeviceBuf &b = ctx->ws[k];
    if (!b.last_use && hipEventCreateWithFlags(&b.last_use, hipEventDisableTiming) != hipSuccess) {
      // cannot stamp: fall back to draining the stream so the buffer is certainly free
      (void)hipStreamSynchronize(ctx->cur_stream);
      b.used = false;
      continue;
    }
    if (hipEventRecord(b.last_use, ctx->cur_stream) != hipSuccess) {
      (void)hipStreamSynchronize(ctx->cur_stream);
      b.used = false;
      continue;
    }
    b.last_stream = ctx->cur_stream;
    b.used = true;
  }
  ctx->ws_touched = 0;
}

// Stages before which every value must be brought back to |v| <= p/2 so that mulmod operands
// stay below 2^50 (f64mod.hpp).  B tracks the worst-case magnitude in units of p.
uint32_t fwd_reduce_mask(uint64_t p, int logn) {
  const double lim = 1125899906842624.0 / (double)p;  // 2^50 / p
  double B = 1.0;
  uint32_t mask = 0;
  for (int s = 0; s < logn; s++) {
    if (B > lim) {
      mask |= 1u << s;
      B = 0.51;
    }
    B += 0.75;
  }
  return mask;
}
// Whether the outputs of a forward transform of length 2^logn (masks as above, inputs |v| <= p) may exceed 2^50: then
// they are reduced before the pointwise product with a (balanced) table entry; below that the product is exact as is.
// PRECONDITION on the table: balanced entries, |s| <= p/2 -- then |v s| <= 2^50 p/2 = p 2^49, mulmod's bound (f64mod.hpp;
// the corner 2^50 x p/2 is case 2 / 3 of tests/f64mod_check.cpp for every preset prime and 2^50 - 27).  A canonical [0, p)
// table would be a factor of two outside it.  witness.hip build_plan checks the tables it uploads; sub_ntt_wide_kernel
// MODE 2 callers hand in tables of the same plan.
bool fwd_end_needs_reduce(uint64_t p, int logn) {
  const double lim = 1125899906842624.0 / (double)p;
  double B = 1.0;
  for (int s = 0; s < logn; s++) {
    if (B > lim) B = 0.51;
    B += 0.75;
  }
  return B > lim;
}
// u0 > 0: the inverse of an incomplete transform (witness_inc.hpp) starts at stage u0, on values |v| <= p
uint32_t inv_reduce_mask(uint64_t p, int logn, int u0) {
  const double lim = 1125899906842624.0 / (double)p;
  double B = 1.0;
  uint32_t mask = 0;
  for (int u = u0; u < logn; u++) {
    if (2.0 * B > lim) {
      mask |= 1u << u;
      B = 0.51;
    }
    B = std::max(2.0 * B, 0.75);
  }
  return mask;
}

template <class T>
static T *upload_words(const std::vector<T> &h) {
  T *d = nullptr;
  RS_HIP(hipMalloc(&d, h.size() * sizeof(T)));
  RS_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

// SEAL NTTTables layout: root = minimal primitive 2n-th root psi, tw[bitrev(i)] = psi^i.
template <class M>
NttTableT<typename HostArith<M>::T, M> make_negacyclic_table(uint64_t p, int logn) {
  using namespace host;
  using T = typename HostArith<M>::T;
  NttTableT<T, M> t;
  t.p = p;
  t.mod = HostArith<M>::make(p);
  t.logn = logn;
  const size_t n = (size_t)1 << logn;
  const uint64_t psi = minimal_primitive_root((uint64_t)2 << logn, p);
  std::vector<T> tw(n), itw(n);
  uint64_t pw = 1;
  for (size_t i = 0; i < n; i++) {
    const uint32_t k = bitrev((uint32_t)i, logn);
    tw[k] = HostArith<M>::konst(pw, p);
    itw[k] = HostArith<M>::konst(invmod(pw, p), p);
    pw = mulmod(pw, psi, p);
  }
  t.d_tw = upload_words(tw);
  t.d_itw = upload_words(itw);
  t.ninv = HostArith<M>::konst(invmod((uint64_t)n % p, p), p);
  t.fwd_red_mask = fwd_reduce_mask(p, logn);
  t.inv_red_mask = inv_reduce_mask(p, logn);
  return t;
}
template NttTable make_negacyclic_table<Mod>(uint64_t, int);
template NttTableI make_negacyclic_table<ModI>(uint64_t, int);

// ---------------------------------------------------------------------------------------------
// a4: batched negacyclic NTT.  One workgroup per polynomial; global <-> LDS traffic is one
// coalesced 16-byte-per-lane read and one write of the polynomial (16*n algorithmic bytes).
// ---------------------------------------------------------------------------------------------
struct GlobalU64In {
  const uint64_t *p;
  __device__ __forceinline__ int pbase(int) const { return 0; }
  __device__ __forceinline__ double load(int base, int, int eoff, int) const { return from_u64(p[base + eoff]); }
  __device__ __forceinline__ void store(int, int, int, int, double) const {}
};
template <bool SCALE>
struct GlobalCanonOut {
  uint64_t *p;
  Mod mod;
  double ninv;
  __device__ __forceinline__ int pbase(int) const { return 0; }
  __device__ __forceinline__ double load(int, int, int, int) const { return 0.0; }
  __device__ __forceinline__ void store(int base, int, int eoff, int, double v) const {
    if (SCALE) v = mulmod(reduce(v, mod), ninv, mod);
    p[base + eoff] = to_u64(canon(v, mod));
  }
};

// DIN: first round reads the polynomial straight from global memory; DOUT: last round writes the
// canonical result straight back.  THREADS is the launch bound (n / 2^MAXR threads do all the work).
template <bool INV, int MAXR, bool DIN, bool DOUT, int THREADS>
__global__ void __launch_bounds__(THREADS) ntt_kernel(uint64_t *__restrict__ data, int logn,
                                                      const double *__restrict__ tw, Mod mod, double ninv,
                                                      uint32_t red_mask) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  const int n = 1 << logn;
  uint64_t *poly = data + (size_t)blockIdx.x * n;
  if (!DIN) {
    const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(poly);
    for (int i = threadIdx.x; i < (n >> 1); i += blockDim.x) {
      const ulonglong2 v = src[i];
      const int pi = pidx(2 * i);
      s[pi] = from_u64(v.x);
      s[pnext(pi)] = from_u64(v.y);
    }
    __syncthreads();
  }
  const LdsIO lds{s};
  const GlobalU64In gin{poly};
  const GlobalCanonOut<INV> gout{poly, mod, ninv};
  if (INV) {
    if (DIN && DOUT)
      lds_ntt_inv_io<MAXR>(s, gin, gout, logn, logn, tw, 1, mod, red_mask);
    else if (DIN)
      lds_ntt_inv_io<MAXR>(s, gin, lds, logn, logn, tw, 1, mod, red_mask);
    else if (DOUT)
      lds_ntt_inv_io<MAXR>(s, lds, gout, logn, logn, tw, 1, mod, red_mask);
    else
      lds_ntt_inv_io<MAXR>(s, lds, lds, logn, logn, tw, 1, mod, red_mask);
  } else {
    if (DIN && DOUT)
      lds_ntt_fwd_io<MAXR>(s, gin, gout, logn, logn, tw, 1, mod, red_mask);
    else if (DIN)
      lds_ntt_fwd_io<MAXR>(s, gin, lds, logn, logn, tw, 1, mod, red_mask);
    else if (DOUT)
      lds_ntt_fwd_io<MAXR>(s, lds, gout, logn, logn, tw, 1, mod, red_mask);
    else
      lds_ntt_fwd_io<MAXR>(s, lds, lds, logn, logn, tw, 1, mod, red_mask);
  }
  if (!DOUT) {
    ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(poly);
    for (int i = threadIdx.x; i < (n >> 1); i += blockDim.x) {
      const int pi = pidx(2 * i);
      double a = s[pi], b = s[pnext(pi)];
      if (INV) {
        a = mulmod(reduce(a, mod), ninv, mod);
        b = mulmod(reduce(b, mod), ninv, mod);
      }
      ulonglong2 o;
      o.x = to_u64(canon(a, mod));
      o.y = to_u64(canon(b, mod));
      dst[i] = o;
    }
  }
}

// Wave-private variant (ntt_core.hpp "wp"): one cross-wave round, then every wave finishes its own
// contiguous block without workgroup barriers, and moves it between LDS and global memory itself
// with fully coalesced 16-byte accesses.
int g_ntt_repeat = 1;
// Break chip-wide lockstep: every workgroup of a launch runs the same load -> compute -> store
// sequence, so without help all of them hit HBM at the same time and then all leave it idle.
// The first generation of workgroups starts after a pseudo-random delay of up to `units` x ~1.7 us;
// later generations inherit the spread.
__device__ __forceinline__ void stagger_start(int units) {
  if (units <= 0 || blockIdx.x >= 1024) return;
  const unsigned h = (blockIdx.x * 2654435761u) >> 28;  // 0..15
  const int n = (int)(h * (unsigned)units) >> 2;
  for (int k = 0; k < n; k++) __builtin_amdgcn_s_sleep(16);  // 16 * 64 cycles ~ 0.43 us
}
struct BlockFactory {
  double *s;
  __device__ __forceinline__ LdsBlockIO operator()(int off) const { return LdsBlockIO{s + pidx(off)}; }
};
template <bool INV, int MAXR, int THREADS, int MINW = 1>
__global__ void __launch_bounds__(THREADS, MINW) ntt_kernel_wp(uint64_t *__restrict__ data, int logn, int logw,
                                                         const double *__restrict__ tw, Mod mod, double ninv,
                                                         uint32_t red_mask, int repeat) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  const int n = 1 << logn;
  uint64_t *poly = data + (size_t)blockIdx.x * n;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int logb = logn - logw, bsz = 1 << logb, off = wave << logb;
  const BlockFactory bf{s};
  stagger_start(repeat >> 8);
  repeat &= 255;
  if (!INV) {
    lds_ntt_fwd_wp<MAXR>(s, GlobalU64In{poly}, bf, logn, logw, tw, mod, red_mask);
    for (int r = 1; r < repeat; r++) {  // experiment only: re-run the transform on the LDS tile
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += blockDim.x) s[pidx(i)] = reduce(s[pidx(i)], mod);
      __syncthreads();
      lds_ntt_fwd_wp<MAXR>(s, LdsIO{s}, bf, logn, logw, tw, mod, red_mask);
    }
    ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(poly + off);
    for (int i = lane; i < (bsz >> 1); i += 64) {
      const int pi = pidx(off + 2 * i);
      ulonglong2 o;
      o.x = to_u64(canon(s[pi], mod));
      o.y = to_u64(canon(s[pnext(pi)], mod));
      dst[i] = o;
    }
  } else {
    const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(poly + off);
    for (int i = lane; i < (bsz >> 1); i += 64) {
      const ulonglong2 v = src[i];
      const int pi = pidx(off + 2 * i);
      s[pi] = from_u64(v.x);
      s[pnext(pi)] = from_u64(v.y);
    }
    wave_sync();
    lds_ntt_inv_wp<MAXR>(s, bf, GlobalCanonOut<true>{poly, mod, ninv}, logn, logw, tw, mod, red_mask);
  }
}

template <bool INV, int MAXR, int THREADS, int MINW = 1>
static void launch_ntt_wp(const NttTable &t, uint64_t *d_data, size_t batch, hipStream_t st) {
  const size_t lds = padded_len((size_t)1 << t.logn) * sizeof(double);
  int logw = 0;
  while ((64 << logw) < THREADS) logw++;
  auto kern = ntt_kernel_wp<INV, MAXR, THREADS, MINW>;
  set_max_dyn_lds((const void *)kern, (int)lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(THREADS), lds, st, d_data, t.logn, logw, INV ? t.d_itw : t.d_tw,
                     t.mod, t.ninv, INV ? t.inv_red_mask : t.fwd_red_mask, g_ntt_repeat);
  RS_HIP(hipGetLastError());
}


// Persistent, software-pipelined forward transform (the streaming form of ntt_kernel_wp): one
// workgroup per CU loops over polynomials of one prime.  The twiddle table lives in LDS next to
// the tile, so the loop's only vector-memory traffic is the data itself, and the NEXT polynomial is
// loaded into registers (in the first round's radix-4 access pattern) while the current one is
// transformed.  1024 threads: two cross-wave radix-4 rounds, then nine wave-private stages per
// 512-point block, then each wave streams its block out.
template <int G>  // radix-4 groups per thread in the first round: n / 4096
__global__ void __launch_bounds__(1024)
ntt_fwd_stream_kernel(uint64_t *__restrict__ data, unsigned long long batch, int logn, const double *__restrict__ tw, Mod mod,
                      uint32_t red_mask) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  const int n = 1 << logn;
  double *twl = s + padded_len((size_t)n);
  for (int i = threadIdx.x; i < n; i += 1024) twl[i] = tw[i];
  const int q = n >> 2;  // gap of the first radix-4 round
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int logb = logn - 4, off = wave << logb;
  const LdsIO lds{s};
  const LdsBlockIO blk{s + pidx(off)};
  const Lanes wl = wave_lanes(true);
  uint64_t pre[G][4];
  // The finished polynomial leaves through registers ONE ITERATION LATE: its stores are issued after the next
  // polynomial's first round has consumed the prefetched loads.  gfx9-family counters track loads and stores in
  // one vmcnt and they complete out of order with respect to each other, so waiting for a load that was issued
  // before stores means vmcnt(0), i.e. draining those stores; issued in this order the only stores in flight at
  // the wait are a whole transform old.
  constexpr int NOUT = 2 * G;  // 16-byte pairs per lane of a wave block: (n / 16) / 2 / 64
  ulonglong2 outr[NOUT];
  unsigned long long p_out = ~0ull;
  unsigned long long p = blockIdx.x;
  if (p < batch) {
    const uint64_t *src = data + p * (size_t)n;
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
      for (int e = 0; e < 4; e++) pre[g][e] = src[threadIdx.x + g * 1024 + e * q];
  }
  __syncthreads();
  const double w1 = twl[1], w2 = twl[2], w3 = twl[3];
  for (; p < batch; p += gridDim.x) {
    // stages 0,1 from registers (root 1: twiddles tw[1]; tw[2], tw[3])
#pragma unroll
    for (int g = 0; g < G; g++) {
      const int base = threadIdx.x + g * 1024;
      double v0 = from_u64(pre[g][0]), v1 = from_u64(pre[g][1]), v2 = from_u64(pre[g][2]), v3 = from_u64(pre[g][3]);
      double t = mulmod(v2, w1, mod), u = mulmod(v3, w1, mod);
      double a0 = v0 + t, a2 = v0 - t, a1 = v1 + u, a3 = v1 - u;
      t = mulmod(a1, w2, mod);
      u = mulmod(a3, w3, mod);
      const int pb = pidx(base);  // base < q: the quarter offsets occupy disjoint bits
      s[pb] = a0 + t;
      s[pcomb(pb, pidx(q))] = a0 - t;
      s[pcomb(pb, pidx(2 * q))] = a2 + u;
      s[pcomb(pb, pidx(3 * q))] = a2 - u;
    }
    __syncthreads();
    if (p_out != ~0ull) {  // the previous polynomial, held in registers since the end of the last iteration
      ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(data + p_out * (size_t)n + off);
#pragma unroll
      for (int k = 0; k < NOUT; k++) dst[lane + 64 * k] = outr[k];
    }
    const unsigned long long pn = p + gridDim.x;
    if (pn < batch) {
      const uint64_t *src = data + pn * (size_t)n;
#pragma unroll
      for (int g = 0; g < G; g++)
#pragma unroll
        for (int e = 0; e < 4; e++) pre[g][e] = src[threadIdx.x + g * 1024 + e * q];
    }
    fwd_round<2>(lds, lds, logn, logn, 2, twl, 1, mod, red_mask, Lanes{(int)threadIdx.x, 1024, true});
    __syncthreads();
    for (int st = 0; st < logb;) {
      const int R = pick_radix(logb - st, 3);
      fwd_round_dispatch<3>(R, blk, blk, logb, logb, st, twl, 16 + wave, mod, red_mask >> 4, wl);
      wave_sync();
      st += R;
    }
#pragma unroll
    for (int k = 0; k < NOUT; k++) {
      const int pi = pidx(off + 2 * (lane + 64 * k));
      outr[k].x = to_u64(canon(s[pi], mod));
      outr[k].y = to_u64(canon(s[pnext(pi)], mod));
    }
    p_out = p;
    __syncthreads();
  }
  if (p_out != ~0ull) {
    ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(data + p_out * (size_t)n + off);
#pragma unroll
    for (int k = 0; k < NOUT; k++) dst[lane + 64 * k] = outr[k];
  }
}

// The transform for ANY arithmetic (used by the integer contexts; the FP64 contexts run the tuned variants above):
// one workgroup per polynomial, tile in LDS, radix-8 rounds with workgroup barriers.
template <bool INV, class T, class M>
__global__ void __launch_bounds__(1024) ntt_generic_kernel(uint64_t *__restrict__ data, int logn, const T *__restrict__ tw, M mod,
                                                           T ninv) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *s = reinterpret_cast<T *>(smem);
  const int n = 1 << logn;
  uint64_t *poly = data + (size_t)blockIdx.x * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s[pidx(i)] = from_res<T>(poly[i]);
  __syncthreads();
  if (INV)
    lds_ntt_inv<3>(s, logn, tw, 1, mod, 0u);
  else
    lds_ntt_fwd<3>(s, logn, tw, 1, mod, 0u);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    T v = s[pidx(i)];
    if (INV) v = mulmod(reduce(v, mod), ninv, mod);
    poly[i] = to_res(canon(v, mod));
  }
}
// The same transform with the first round reading the polynomial straight from global memory and the last round writing the
// canonical result straight back (no staging passes), radix-16 rounds: three tile exchanges at 16384 points instead of the
// generic kernel's five + two (round 4; the integer contexts -- microbench.cpp:33-36's 59/60-bit primes).
template <class T>
struct GlobalResIn {
  const uint64_t *p;
  __device__ __forceinline__ int pbase(int) const { return 0; }
  __device__ __forceinline__ T load(int base, int, int eoff, int) const { return from_res<T>(p[base + eoff]); }
  __device__ __forceinline__ void store(int, int, int, int, T) const {}
};
template <bool SCALE, class T, class M>
struct GlobalResOut {
  uint64_t *p;
  M mod;
  T ninv;
  __device__ __forceinline__ int pbase(int) const { return 0; }
  __device__ __forceinline__ T load(int, int, int, int) const { return T(0); }
  __device__ __forceinline__ void store(int base, int, int eoff, int, T v) const {
    if (SCALE) v = mulmod(reduce(v, mod), ninv, mod);
    p[base + eoff] = to_res(canon(v, mod));
  }
};
template <bool INV, class T, class M>
__global__ void __launch_bounds__(1024) ntt_io_kernel(uint64_t *__restrict__ data, int logn, const T *__restrict__ tw, M mod, T ninv) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *s = reinterpret_cast<T *>(smem);
  uint64_t *poly = data + ((size_t)blockIdx.x << logn);
  const GlobalResIn<T> gin{poly};
  const GlobalResOut<INV, T, M> gout{poly, mod, ninv};
  if (INV)
    lds_ntt_inv_io<4>(s, gin, gout, logn, logn, tw, 1, mod, 0u);
  else
    lds_ntt_fwd_io<4>(s, gin, gout, logn, logn, tw, 1, mod, 0u);
}
int g_int_ntt_variant = 1;  // tuning knob "int_ntt_variant": 1 = ntt_io_kernel (lengths >= 2^10), 0 = ntt_generic_kernel
void launch_ntt_int(rs_ctx *ctx, const NttTableI &t, uint64_t *d_data, size_t batch, bool inverse, hipStream_t st) {
  (void)ctx;
  if (batch == 0) return;
  if (g_int_ntt_variant == 1 && t.logn >= 10) {  // at least two rounds, so that the in-place global reads all precede the writes
    const size_t lds = padded_len((size_t)1 << t.logn) * sizeof(uint64_t);
    const int thr = std::max(64, std::min(1024, (1 << t.logn) >> 4));
    if (inverse) {
      set_max_dyn_lds((const void *)ntt_io_kernel<true, uint64_t, ModI>, (int)lds);
      hipLaunchKernelGGL((ntt_io_kernel<true, uint64_t, ModI>), dim3((unsigned)batch), dim3(thr), lds, st, d_data, t.logn, t.d_itw, t.mod, t.ninv);
    } else {
      set_max_dyn_lds((const void *)ntt_io_kernel<false, uint64_t, ModI>, (int)lds);
      hipLaunchKernelGGL((ntt_io_kernel<false, uint64_t, ModI>), dim3((unsigned)batch), dim3(thr), lds, st, d_data, t.logn, t.d_tw, t.mod, t.ninv);
    }
    RS_HIP(hipGetLastError());
    return;
  }
  const size_t lds = padded_len((size_t)1 << t.logn) * sizeof(uint64_t);
  const int thr = std::max(64, std::min(1024, (1 << t.logn) >> 3));
  if (inverse) {
    set_max_dyn_lds((const void *)ntt_generic_kernel<true, uint64_t, ModI>, (int)lds);
    hipLaunchKernelGGL((ntt_generic_kernel<true, uint64_t, ModI>), dim3((unsigned)batch), dim3(thr), lds, st, d_data, t.logn, t.d_itw, t.mod, t.ninv);
  } else {
    set_max_dyn_lds((const void *)ntt_generic_kernel<false, uint64_t, ModI>, (int)lds);
    hipLaunchKernelGGL((ntt_generic_kernel<false, uint64_t, ModI>), dim3((unsigned)batch), dim3(thr), lds, st, d_data, t.logn, t.d_tw, t.mod, t.ninv);
  }
  RS_HIP(hipGetLastError());
}

bool g_force_int = false;  // tuning knob "force_int_arith" (tests: both arithmetics on the same primes)
int g_ntt_variant = 14;  // tuning knob (rs_set_tuning("ntt_variant", v)): see launch_ntt

template <bool INV, int MAXR, bool DIN, bool DOUT, int THREADS>
static void launch_ntt_variant(const NttTable &t, uint64_t *d_data, size_t batch, hipStream_t st) {
  const size_t lds = padded_len((size_t)1 << t.logn) * sizeof(double);
  const int n = 1 << t.logn;
  const int thr = std::max(64, std::min(THREADS, n >> MAXR));
  auto kern = ntt_kernel<INV, MAXR, DIN, DOUT, THREADS>;
  set_max_dyn_lds((const void *)kern, (int)lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(thr), lds, st, d_data, t.logn, INV ? t.d_itw : t.d_tw, t.mod,
                     t.ninv, INV ? t.inv_red_mask : t.fwd_red_mask);
  RS_HIP(hipGetLastError());
}

// ntt_wide.hpp kernels: persistent, two workgroups of 256 threads per CU (16384 points: one of 512)
int g_ntt_wide_grid = 256;  // tuning knob "ntt_wide_grid": CUs to fill (workgroups = this x what fits one CU)
template <int LOGN, bool RED>
static void launch_ntt_wide_shape(const NttTable &t, uint64_t *d_data, size_t batch, bool inverse, hipStream_t st) {
  using S = WideShape<LOGN>;
  const int lds = (int)S::LDS_BYTES;
  const size_t per_cu = std::min<size_t>(8 / (S::T / 64), (size_t)(160 * 1024) / S::LDS_BYTES);  // 2 waves per SIMD
  const unsigned grid = (unsigned)std::min<size_t>(batch, (size_t)g_ntt_wide_grid * per_cu);
  if (inverse) {
    auto kern = ntt_inv_wide_kernel<LOGN, RED>;
    set_max_dyn_lds((const void *)kern, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(S::T), lds, st, d_data, (unsigned long long)batch, t.d_itw, t.mod, t.ninv,
                       t.inv_red_mask);
  } else {
    auto kern = ntt_fwd_wide_kernel<LOGN, RED>;
    set_max_dyn_lds((const void *)kern, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(S::T), lds, st, d_data, (unsigned long long)batch, t.d_tw, t.mod, t.fwd_red_mask);
  }
  RS_HIP(hipGetLastError());
}
static bool launch_ntt_wide(const NttTable &t, uint64_t *d_data, size_t batch, bool inverse, hipStream_t st) {
  const bool red = (inverse ? t.inv_red_mask : t.fwd_red_mask) != 0;
  if (t.logn == 14)
    red ? launch_ntt_wide_shape<14, true>(t, d_data, batch, inverse, st) : launch_ntt_wide_shape<14, false>(t, d_data, batch, inverse, st);
  else if (t.logn == 13)
    red ? launch_ntt_wide_shape<13, true>(t, d_data, batch, inverse, st) : launch_ntt_wide_shape<13, false>(t, d_data, batch, inverse, st);
  else if (t.logn == 12)
    red ? launch_ntt_wide_shape<12, true>(t, d_data, batch, inverse, st) : launch_ntt_wide_shape<12, false>(t, d_data, batch, inverse, st);
  else
    return false;
  return true;
}

void launch_ntt(rs_ctx *ctx, const NttTable &t, uint64_t *d_data, size_t batch, bool inverse, hipStream_t st) {
  (void)ctx;
  if (batch == 0) return;
#define RS_NTT_CASE(V, MAXR, DIN_F, DOUT_F, DIN_I, DOUT_I, THR)                         \
  case V:                                                                              \
    if (inverse)                                                                       \
      launch_ntt_variant<true, MAXR, DIN_I, DOUT_I, THR>(t, d_data, batch, st);        \
    else                                                                               \
      launch_ntt_variant<false, MAXR, DIN_F, DOUT_F, THR>(t, d_data, batch, st);       \
    break;
  if (g_ntt_variant == 14 && launch_ntt_wide(t, d_data, batch, inverse, st)) return;
  if ((g_ntt_variant == 12 || g_ntt_variant == 14) && !inverse && t.logn >= 12 && t.logn <= 13 && t.fwd_red_mask < 4) {
    const size_t lds = (padded_len((size_t)1 << t.logn) + ((size_t)1 << t.logn)) * sizeof(double);
    const unsigned grid = (unsigned)std::min<size_t>(batch, 256);
    if (t.logn == 13) {
      set_max_dyn_lds((const void *)ntt_fwd_stream_kernel<2>, (int)lds);
      hipLaunchKernelGGL(ntt_fwd_stream_kernel<2>, dim3(grid), dim3(1024), lds, st, d_data, (unsigned long long)batch, t.logn,
                         t.d_tw, t.mod, t.fwd_red_mask);
    } else {
      set_max_dyn_lds((const void *)ntt_fwd_stream_kernel<1>, (int)lds);
      hipLaunchKernelGGL(ntt_fwd_stream_kernel<1>, dim3(grid), dim3(1024), lds, st, d_data, (unsigned long long)batch, t.logn,
                         t.d_tw, t.mod, t.fwd_red_mask);
    }
    RS_HIP(hipGetLastError());
    return;
  }
  const int wp_waves = (g_ntt_variant == 9 || g_ntt_variant == 13) ? 16 : (g_ntt_variant == 10 ? 4 : 8);
  const bool wp_ok = (1 << t.logn) >= wp_waves * LDS_BLOCK_MIN;  // wave-private blocks need n / W >= LDS_BLOCK_MIN
  if (wp_ok && g_ntt_variant >= 8 && g_ntt_variant <= 14) {
    switch (g_ntt_variant) {
      case 8:
      case 12:  // streaming forward kernel not applicable (inverse, or shape): wave-private kernel
      case 14:  // wide kernels not applicable (shape)
        inverse ? launch_ntt_wp<true, 4, 512>(t, d_data, batch, st) : launch_ntt_wp<false, 4, 512>(t, d_data, batch, st);
        break;
      case 9:
        inverse ? launch_ntt_wp<true, 3, 1024>(t, d_data, batch, st) : launch_ntt_wp<false, 3, 1024>(t, d_data, batch, st);
        break;
      case 13:  // radix-8 rounds at 8 waves per SIMD (<= 64 VGPRs), two workgroups per CU
        inverse ? launch_ntt_wp<true, 3, 1024, 8>(t, d_data, batch, st) : launch_ntt_wp<false, 3, 1024, 8>(t, d_data, batch, st);
        break;
      case 10:
        inverse ? launch_ntt_wp<true, 5, 256>(t, d_data, batch, st) : launch_ntt_wp<false, 5, 256>(t, d_data, batch, st);
        break;
      default:
        inverse ? launch_ntt_wp<true, 3, 512>(t, d_data, batch, st) : launch_ntt_wp<false, 3, 512>(t, d_data, batch, st);
        break;
    }
    return;
  }
  switch (g_ntt_variant) {
    RS_NTT_CASE(1, 4, false, false, false, false, 512)
    RS_NTT_CASE(2, 4, true, false, false, true, 512)
    RS_NTT_CASE(3, 4, true, true, true, true, 512)
    RS_NTT_CASE(4, 5, false, false, false, false, 256)
    RS_NTT_CASE(5, 5, true, false, false, true, 256)
    RS_NTT_CASE(6, 5, true, true, true, true, 256)
    RS_NTT_CASE(7, 3, true, false, false, true, 1024)
    default:  // 0, and any wave-private request on a transform too short for it
      RS_NTT_CASE(0, 3, false, false, false, false, 1024)
  }
#undef RS_NTT_CASE
}

// ---------------------------------------------------------------------------------------------
// a1-a3: dyadic RingElem kernels on [count][L][N]; 16 bytes per lane, grid-stride.
// ---------------------------------------------------------------------------------------------
enum RingOp { OP_ADD, OP_SUB, OP_MUL, OP_NEG, OP_ADD_SCALAR, OP_MUL_SCALAR };

template <class T>
struct ScalarPerLimbT {
  T v[RS_MAX_L];
};

template <int OP, class M>
__device__ __forceinline__ uint64_t ring_apply(uint64_t a, uint64_t b, typename ArithOf<M>::T sc, const M m) {
  using T = typename ArithOf<M>::T;
  const T x = from_res<T>(a);
  T r;
  if (OP == OP_ADD)
    r = addm(x, from_res<T>(b), m);
  else if (OP == OP_SUB)
    r = subm(x, from_res<T>(b), m);
  else if (OP == OP_MUL)
    r = mulmod_dd(x, center(from_res<T>(b), m), m);
  else if (OP == OP_NEG)
    r = negm(x, m);
  else if (OP == OP_ADD_SCALAR)
    r = addm(x, sc, m);  // sc: the scalar's residue as a data value
  else
    r = mulmod(x, sc, m);  // sc: the scalar's residue as a table constant
  return to_res(canon(r, m));
}

// Rows a1 / a2 (SealPoly dyadic ops, seal_ring.tcc:62-247): HBM-bound streams -- 16 or 24 algorithmic bytes per residue.
// The shape that reaches the device's copy rate (rs_measure_peaks' peak_copy_kernel): a workgroup moves CONTIGUOUS 32 KiB
// pieces of every operand, eight 16-byte loads per operand in flight per lane before the first use, pieces dealt round-robin
// to the workgroups; NT: non-temporal accesses (large batches, touched once).  `pairs` 16-byte words; limb of word i =
// (i >> logN) % L -- uniform over a piece when a limb row holds at least a piece (logN >= 12), else per access.
template <int OP, class M, bool NT>
__global__ void __launch_bounds__(256) ring_pointwise_kernel(uint64_t *__restrict__ dst, const uint64_t *__restrict__ a,
                                                             const uint64_t *__restrict__ b, size_t pairs, int logN, int L,
                                                             const M *__restrict__ qmod, ScalarPerLimbT<typename ArithOf<M>::T> sc) {
  constexpr bool BIN = OP == OP_ADD || OP == OP_SUB || OP == OP_MUL;
  const u64x2 *a2 = reinterpret_cast<const u64x2 *>(a), *b2 = reinterpret_cast<const u64x2 *>(b);
  u64x2 *d2 = reinterpret_cast<u64x2 *>(dst);
  const size_t full = pairs & ~(size_t)2047;
  const bool uniform = logN >= 12;
  for (size_t base = (size_t)blockIdx.x * 2048; base < full; base += (size_t)gridDim.x * 2048) {
    u64x2 va[8], vb[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const size_t i = base + threadIdx.x + 256 * k;
      va[k] = NT ? __builtin_nontemporal_load(a2 + i) : a2[i];
      vb[k] = va[k];
      if (BIN) vb[k] = NT ? __builtin_nontemporal_load(b2 + i) : b2[i];
    }
    const int limb_u = (int)(((2 * base) >> logN) % (size_t)L);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const size_t i = base + threadIdx.x + 256 * k;
      const int limb = uniform ? limb_u : (int)((unsigned)((2 * i) >> logN) % (unsigned)L);
      const M m = qmod[limb];
      u64x2 o;
      o.x = ring_apply<OP, M>(va[k].x, vb[k].x, sc.v[limb], m);
      o.y = ring_apply<OP, M>(va[k].y, vb[k].y, sc.v[limb], m);
      if (NT) __builtin_nontemporal_store(o, d2 + i);
      else d2[i] = o;
    }
  }
  for (size_t i = full + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) {  // the tail (< 32 KiB)
    const int limb = (int)((unsigned)((2 * i) >> logN) % (unsigned)L);
    const M m = qmod[limb];
    const u64x2 x = a2[i], y = BIN ? b2[i] : x;
    u64x2 o;
    o.x = ring_apply<OP, M>(x.x, y.x, sc.v[limb], m);
    o.y = ring_apply<OP, M>(x.y, y.y, sc.v[limb], m);
    d2[i] = o;
  }
}

template <int OP, class M>
static void launch_pointwise_arith(rs_ctx *ctx, uint64_t *dst, const uint64_t *a, const uint64_t *b, size_t count,
                                   uint64_t scalar, hipStream_t st) {
  const size_t pairs = count * ctx->ring_words() / 2;
  if (!pairs) return;
  ScalarPerLimbT<typename ArithOf<M>::T> sc{};
  for (int i = 0; i < ctx->L; i++)
    sc.v[i] = OP == OP_MUL_SCALAR ? HostArith<M>::konst(scalar, ctx->q[i]) : HostArith<M>::plain(scalar, ctx->q[i]);
  int logN = 0;
  while ((1 << logN) < ctx->N) logN++;
  RS_REQUIRE((1 << logN) == ctx->N, "ring degree must be a power of two");
  const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((pairs + 2047) / 2048, 256 * 8));
  // non-temporal accesses once the operands are larger than the caches they would otherwise sweep (256 MiB Infinity Cache)
  if (pairs * 16 >= ((size_t)64 << 20))
    hipLaunchKernelGGL((ring_pointwise_kernel<OP, M, true>), dim3(blocks), dim3(256), 0, st, dst, a, b, pairs, logN, ctx->L,
                       CtxArith<M>::qmod(ctx), sc);
  else
    hipLaunchKernelGGL((ring_pointwise_kernel<OP, M, false>), dim3(blocks), dim3(256), 0, st, dst, a, b, pairs, logN, ctx->L,
                       CtxArith<M>::qmod(ctx), sc);
  RS_HIP(hipGetLastError());
}
template <int OP>
static void launch_pointwise(rs_ctx *ctx, uint64_t *dst, const uint64_t *a, const uint64_t *b, size_t count,
                             uint64_t scalar, hipStream_t st) {
  RS_DISPATCH_ARITH(ctx, (launch_pointwise_arith<OP, Mod>(ctx, dst, a, b, count, scalar, st)),
                    (launch_pointwise_arith<OP, ModI>(ctx, dst, a, b, count, scalar, st)));
}

// slot-wise inverse by Fermat (a^(p-2)); flags[0] |= 1 if any slot is zero.
__global__ void __launch_bounds__(256) ring_inv_kernel(uint64_t *__restrict__ dst, const uint64_t *__restrict__ a,
                                                       size_t words, int N, int L, const Mod *__restrict__ qmod,
                                                       const uint64_t *__restrict__ qint, unsigned *flags) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  bool any_zero = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) {
    const int limb = (int)((i / (size_t)N) % (size_t)L);
    const Mod m = qmod[limb];
    const uint64_t e = qint[limb] - 2;
    const uint64_t av = a[i];
    any_zero |= (av == 0);
    double base = center(from_u64(av), m), acc = 1.0;
    for (int bit = 0; bit < 52; bit++) {
      if ((e >> bit) & 1ull) acc = mulmod(acc, base, m);
      base = mulmod(base, base, m);
    }
    dst[i] = to_u64(canon(acc, m));
  }
  if (any_zero) atomicOr(flags, 1u);
}
// the same in the Montgomery domain: base and accumulator are kept as x*R, every product is one reduction
__global__ void __launch_bounds__(256) ring_inv_kernel_int(uint64_t *__restrict__ dst, const uint64_t *__restrict__ a,
                                                           size_t words, int N, int L, const ModI *__restrict__ qmod, unsigned *flags) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  bool any_zero = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) {
    const ModI m = qmod[(i / (size_t)N) % (size_t)L];
    const uint64_t e = m.p - 2, av = a[i];
    any_zero |= (av == 0);
    uint64_t base = to_mont(av, m), acc = to_mont(1, m);
    for (int bit = 0; bit < 62; bit++) {
      if ((e >> bit) & 1ull) acc = montmul(acc, base, m);
      base = montmul(base, base, m);
    }
    dst[i] = montmul(acc, 1, m);
  }
  if (any_zero) atomicOr(flags, 1u);
}

// per-element all-zero test: flags[k] = 1 if element k has a nonzero word
__global__ void __launch_bounds__(256) ring_nonzero_kernel(const uint64_t *__restrict__ a, size_t words_per_elem,
                                                           unsigned *__restrict__ flags) {
  const uint64_t *e = a + (size_t)blockIdx.x * words_per_elem;
  bool nz = false;
  for (size_t i = threadIdx.x; i < words_per_elem; i += blockDim.x) nz |= (e[i] != 0);
  if (__syncthreads_or(nz) && threadIdx.x == 0) flags[blockIdx.x] = 1u;
}

}  // namespace rs

using namespace rs;

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char *rs_last_error(void) { return g_last_error.c_str(); }
int rs_version(void) { return 101; }  // 101: rs_msm_vec::slot_const (struct must be zero-initialised), rs_enc_noise_budget, RS_ERR_NOISE

int rs_ctx_create(int device, int N, int L, const uint64_t *q, int N_enc, int K, const uint64_t *Q, rs_ctx **out) {
  RS_API_BEGIN
  RS_REQUIRE(out && q && Q, "null argument");
  RS_REQUIRE(L >= 1 && L <= RS_MAX_L && K >= 1 && K <= RS_MAX_K, "L or K out of range");
  RS_REQUIRE(N >= 2 && (N & (N - 1)) == 0 && N_enc >= N && (N_enc & (N_enc - 1)) == 0, "N, N_enc must be powers of two, N <= N_enc");
  RS_REQUIRE(N_enc >= 16, "N_enc must be >= 16");
  if (N_enc > 16384)
    throw Error(RS_ERR_UNSUPPORTED, "N_enc > 16384 does not fit one workgroup's LDS tile (160 KiB); not built in this round");
  int ndev = 0;
  RS_HIP(hipGetDeviceCount(&ndev));
  if (ndev <= 0) throw Error(RS_ERR_HIP, "no HIP device: librs_hip has no CPU fallback");
  RS_REQUIRE(device >= 0 && device < ndev, "device index out of range");
  DeviceGuard device_guard(device);  // the caller's current device is restored on return
  bool any_big = false;
  for (int i = 0; i < L + K; i++) {
    const uint64_t p = i < L ? q[i] : Q[i - L];
    RS_REQUIRE(host::is_prime(p), "modulus is not prime");
    RS_REQUIRE((p - 1) % (2 * (uint64_t)N_enc) == 0, "modulus must be 1 mod 2*N_enc (batching, seal_ring.hpp:297)");
    RS_REQUIRE(p < (1ull << 62), "modulus must be below 2^62 (SEAL's own limit is 61 bits)");
    if (p >= (1ull << 50)) any_big = true;  // beyond the exact-FP64 range: the whole context runs on Montgomery integers
    for (int k = 0; k < i; k++) RS_REQUIRE(p != (k < L ? q[k] : Q[k - L]), "moduli must be pairwise distinct");
  }
  rs_ctx *c = new rs_ctx();
  c->device = device;
  c->N = N;
  c->L = L;
  c->N_enc = N_enc;
  c->K = K;
  c->logN_enc = 0;
  while ((1 << c->logN_enc) < N_enc) c->logN_enc++;
  c->use_int = any_big || g_force_int;
  for (int i = 0; i < L; i++) c->q[i] = q[i];
  for (int j = 0; j < K; j++) c->Q[j] = Q[j];
  if (c->use_int) {
    std::vector<ModI> qm(L), Qm(K);
    for (int i = 0; i < L; i++) {
      c->plain_i[i] = make_negacyclic_table<ModI>(q[i], c->logN_enc);
      qm[i] = c->plain_i[i].mod;
    }
    for (int j = 0; j < K; j++) {
      c->coeff_i[j] = make_negacyclic_table<ModI>(Q[j], c->logN_enc);
      Qm[j] = c->coeff_i[j].mod;
    }
    RS_HIP(hipMalloc(&c->d_qmod_i, sizeof(ModI) * L));
    RS_HIP(hipMemcpy(c->d_qmod_i, qm.data(), sizeof(ModI) * L, hipMemcpyHostToDevice));
    RS_HIP(hipMalloc(&c->d_Qmod_i, sizeof(ModI) * K));
    RS_HIP(hipMemcpy(c->d_Qmod_i, Qm.data(), sizeof(ModI) * K, hipMemcpyHostToDevice));
    // Hybrid: only ring primes are beyond 2^50 (the 54-bit BFVDefault(2048) prime of bench_logistic_regression_inference.cpp
    // :20-27 under 48/49-bit data primes): the inner products -- everything mod Q_j -- keep the FP64 kernels (msm.hip)
    bool small_Q = !g_force_int;
    for (int j = 0; j < K; j++) small_Q = small_Q && Q[j] < (1ull << 50);
    for (int i = 0; i < L; i++) small_Q = small_Q && q[i] < (1ull << 54);
    c->hybrid = small_Q;
    if (c->hybrid)
      for (int j = 0; j < K; j++) c->coeff[j] = make_negacyclic_table<Mod>(Q[j], c->logN_enc);
  } else {
    std::vector<Mod> qm(L), Qm(K);
    for (int i = 0; i < L; i++) {
      c->plain[i] = make_negacyclic_table<Mod>(q[i], c->logN_enc);
      qm[i] = c->plain[i].mod;
    }
    for (int j = 0; j < K; j++) {
      c->coeff[j] = make_negacyclic_table<Mod>(Q[j], c->logN_enc);
      Qm[j] = c->coeff[j].mod;
    }
    RS_HIP(hipMalloc(&c->d_qmod, sizeof(Mod) * L));
    RS_HIP(hipMemcpy(c->d_qmod, qm.data(), sizeof(Mod) * L, hipMemcpyHostToDevice));
    RS_HIP(hipMalloc(&c->d_Qmod, sizeof(Mod) * K));
    RS_HIP(hipMemcpy(c->d_Qmod, Qm.data(), sizeof(Mod) * K, hipMemcpyHostToDevice));
  }
  // BatchEncoder slot map (SEAL batchencoder.cpp populate_matrix_reps_index_map): generator 3
  // of Z_{2n}^*, row 0 = powers 3^i, row 1 = their negatives, bit-reversed positions.
  {
    std::vector<uint32_t> map(N_enc);
    const uint64_t mm = (uint64_t)N_enc << 1;
    const size_t row = (size_t)N_enc >> 1;
    uint64_t pos = 1;
    for (size_t i = 0; i < row; i++) {
      map[i] = host::bitrev((uint32_t)((pos - 1) >> 1), c->logN_enc);
      map[row | i] = host::bitrev((uint32_t)((mm - pos - 1) >> 1), c->logN_enc);
      pos = (pos * 3) & (mm - 1);
    }
    RS_HIP(hipMalloc(&c->d_index_map, sizeof(uint32_t) * N_enc));
    RS_HIP(hipMemcpy(c->d_index_map, map.data(), sizeof(uint32_t) * N_enc, hipMemcpyHostToDevice));
  }
  *out = c;
  RS_API_END
}

void rs_witness_plans_destroy(rs_ctx *ctx);  // witness.hip

void rs_ctx_destroy(rs_ctx *c) {
  if (!c) return;
  int prev_device = -1;
  (void)hipGetDevice(&prev_device);
  (void)hipSetDevice(c->device);
  rs_witness_plans_destroy(c);
  rs::msm_scratch_release(c);
  for (int i = 0; i < c->L; i++) free_table(c->plain[i]), free_table(c->plain_i[i]);
  for (int j = 0; j < c->K; j++) free_table(c->coeff[j]), free_table(c->coeff_i[j]);
  if (c->d_qmod) (void)hipFree(c->d_qmod);
  if (c->d_Qmod) (void)hipFree(c->d_Qmod);
  if (c->d_qmod_i) (void)hipFree(c->d_qmod_i);
  if (c->d_Qmod_i) (void)hipFree(c->d_Qmod_i);
  if (c->d_index_map) (void)hipFree(c->d_index_map);
  if (c->d_noise_thr) (void)hipFree(c->d_noise_thr);
  if (c->d_crt_limbs) (void)hipFree(c->d_crt_limbs);
  for (auto &r : c->prof) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  for (auto e : c->prof_pool) (void)hipEventDestroy(e);
  for (auto &b : c->ws) {
    if (b.last_use) (void)hipEventDestroy(b.last_use);
    if (b.p) (void)hipFree(b.p);
  }
  if (prev_device >= 0) (void)hipSetDevice(prev_device);
  delete c;
}

int rs_malloc(rs_ctx *ctx, size_t bytes, void **d_ptr) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_ptr, "null argument");
  RS_HIP(hipMalloc(d_ptr, bytes));
  RS_API_END
}
int rs_free(rs_ctx *ctx, void *d_ptr) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx, "null argument");
  if (d_ptr) RS_HIP(hipFree(d_ptr));
  RS_API_END
}
int rs_upload(rs_ctx *ctx, void *d_dst, const void *h_src, size_t bytes, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx, "null argument");
  RS_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, S(stream)));
  RS_HIP(hipStreamSynchronize(S(stream)));
  RS_API_END
}
int rs_download(rs_ctx *ctx, void *h_dst, const void *d_src, size_t bytes, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx, "null argument");
  RS_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, S(stream)));
  RS_HIP(hipStreamSynchronize(S(stream)));
  RS_API_END
}
int rs_sync(rs_ctx *ctx, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx, "null argument");
  RS_HIP(hipStreamSynchronize(S(stream)));
  RS_API_END
}

static void ntt_entry(rs_ctx *ctx, int modset, int index, uint64_t *d_data, size_t batch, bool inverse, hipStream_t st) {
  RS_REQUIRE(ctx, "null context");
  RS_REQUIRE(modset == RS_MOD_PLAIN || modset == RS_MOD_COEFF, "bad modulus set");
  RS_REQUIRE(index >= 0 && index < (modset == RS_MOD_PLAIN ? ctx->L : ctx->K), "modulus index out of range");
  if (ctx->use_int)
    launch_ntt_int(ctx, modset == RS_MOD_PLAIN ? ctx->plain_i[index] : ctx->coeff_i[index], d_data, batch, inverse, st);
  else
    launch_ntt(ctx, modset == RS_MOD_PLAIN ? ctx->plain[index] : ctx->coeff[index], d_data, batch, inverse, st);
}

int rs_ntt_forward(rs_ctx *ctx, int modset, int index, uint64_t *d_data, size_t batch, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  ntt_entry(ctx, modset, index, d_data, batch, false, S(stream));
  RS_API_END
}
int rs_ntt_inverse(rs_ctx *ctx, int modset, int index, uint64_t *d_data, size_t batch, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  ntt_entry(ctx, modset, index, d_data, batch, true, S(stream));
  RS_API_END
}

#define RS_POINTWISE(NAME, OP, HAS_B)                                                                        \
  int NAME(rs_ctx *ctx, uint64_t *d_dst, const uint64_t *d_a, const uint64_t *d_b, size_t count, rs_stream s) { \
    RS_API_BEGIN_CTX(ctx)                                                                                    \
    RS_REQUIRE(ctx && d_dst && d_a && (d_b || !HAS_B), "null argument");                                     \
    launch_pointwise<OP>(ctx, d_dst, d_a, d_b, count, 0, S(s));                                              \
    RS_API_END                                                                                               \
  }
RS_POINTWISE(rs_ring_add, OP_ADD, true)
RS_POINTWISE(rs_ring_sub, OP_SUB, true)
RS_POINTWISE(rs_ring_mul, OP_MUL, true)

int rs_ring_neg(rs_ctx *ctx, uint64_t *d_dst, const uint64_t *d_a, size_t count, rs_stream s) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_dst && d_a, "null argument");
  launch_pointwise<OP_NEG>(ctx, d_dst, d_a, nullptr, count, 0, S(s));
  RS_API_END
}
int rs_ring_add_scalar(rs_ctx *ctx, uint64_t *d_dst, const uint64_t *d_a, uint64_t scalar, size_t count, rs_stream s) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_dst && d_a, "null argument");
  launch_pointwise<OP_ADD_SCALAR>(ctx, d_dst, d_a, nullptr, count, scalar, S(s));
  RS_API_END
}
int rs_ring_mul_scalar(rs_ctx *ctx, uint64_t *d_dst, const uint64_t *d_a, uint64_t scalar, size_t count, rs_stream s) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_dst && d_a, "null argument");
  launch_pointwise<OP_MUL_SCALAR>(ctx, d_dst, d_a, nullptr, count, scalar, S(s));
  RS_API_END
}

int rs_ring_inv(rs_ctx *ctx, uint64_t *d_dst, const uint64_t *d_a, size_t count, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_dst && d_a, "null argument");
  const size_t words = count * ctx->ring_words();
  if (words) {
    WsScope ws_scope(ctx, S(stream));
    char *ws = (char *)ws_get(ctx, 7, 256);
    unsigned *flags = (unsigned *)ws;
    uint64_t *qint = (uint64_t *)(ws + 64);
    RS_HIP(hipMemsetAsync(flags, 0, 4, S(stream)));
    RS_HIP(hipMemcpyAsync(qint, ctx->q, sizeof(uint64_t) * ctx->L, hipMemcpyHostToDevice, S(stream)));
    const unsigned blocks = (unsigned)std::min<size_t>((words + 255) / 256, 256 * 16);
    if (ctx->use_int)
      hipLaunchKernelGGL(ring_inv_kernel_int, dim3(blocks), dim3(256), 0, S(stream), d_dst, d_a, words, ctx->N, ctx->L,
                         ctx->d_qmod_i, flags);
    else
      hipLaunchKernelGGL(ring_inv_kernel, dim3(blocks), dim3(256), 0, S(stream), d_dst, d_a, words, ctx->N, ctx->L,
                         ctx->d_qmod, qint, flags);
    unsigned h = 0;
    RS_HIP(hipMemcpyAsync(&h, flags, 4, hipMemcpyDeviceToHost, S(stream)));
    RS_HIP(hipStreamSynchronize(S(stream)));
    if (h) throw Error(RS_ERR_NOT_INVERTIBLE, "element is not invertible in ring");
  }
  RS_API_END
}

int rs_ring_is_zero(rs_ctx *ctx, const uint64_t *d_a, size_t count, uint8_t *h_flags, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_a && h_flags, "null argument");
  if (count) {
    WsScope ws_scope(ctx, S(stream));
    unsigned *flags = (unsigned *)ws_get(ctx, 7, std::max<size_t>(256, count * 4));
    RS_HIP(hipMemsetAsync(flags, 0, count * 4, S(stream)));
    hipLaunchKernelGGL(ring_nonzero_kernel, dim3((unsigned)count), dim3(256), 0, S(stream), d_a, ctx->ring_words(), flags);
    std::vector<unsigned> h(count);
    RS_HIP(hipMemcpyAsync(h.data(), flags, count * 4, hipMemcpyDeviceToHost, S(stream)));
    RS_HIP(hipStreamSynchronize(S(stream)));
    for (size_t k = 0; k < count; k++) h_flags[k] = h[k] ? 0 : 1;
  }
  RS_API_END
}

int rs_set_tuning(const char *key, int value) {
  RS_API_BEGIN
  RS_REQUIRE(key, "null argument");
  if (std::string(key) == "ntt_variant")
    g_ntt_variant = value;
  else if (std::string(key) == "force_int_arith")  // contexts created from now on use the Montgomery path regardless of prime size
    g_force_int = value != 0;
#ifdef RS_EXPERIMENTS  // these two CHANGE THE RESULTS (timing experiments, tools/): `make experiments` only
  else if (std::string(key) == "ntt_repeat")
    g_ntt_repeat = value;
  else if (std::string(key) == "mac_ablate")
    g_mac_ablate = value;
#endif
  else if (std::string(key) == "mac_variant")
    g_mac_variant = value;
  else if (std::string(key) == "mac_ct_temporal")
    g_mac_ct_temporal = value ? 1 : 0;
  else if (std::string(key) == "plain_variant")
    g_plain_variant = value;
  else if (std::string(key) == "prover_lin_io")
    g_prover_lin_io = value;
  else if (std::string(key) == "msm_host_tile")
    g_msm_host_tile = std::max(1, value);
  else if (std::string(key) == "msm_c_mib") {
    RS_REQUIRE(value >= 1, "msm_c_mib must be positive");
    g_msm_c_mib = value;
  } else if (std::string(key) == "mac_chunk_units")
    g_mac_chunk_units = std::max(1, value);
  else if (std::string(key) == "mac_share_keys")
    g_mac_share_keys = value != 0;
  else if (std::string(key) == "ntt_wide_grid")
    g_ntt_wide_grid = std::max(1, value);
  else if (std::string(key) == "int_ntt_variant")
    g_int_ntt_variant = value;
  else if (std::string(key) == "witness_sub_log") {
    RS_REQUIRE(value == 12 || value == 13, "witness_sub_log must be 12 or 13");
    g_witness_sub_log = value;
  } else if (std::string(key) == "witness_h_coset") {
    g_witness_h_coset = value ? 1 : 0;
  } else if (std::string(key) == "witness_sub12_cross") {
    RS_REQUIRE(value >= 1 && value <= 8, "witness_sub12_cross must be in [1, 8]");
    g_witness_sub12_cross = value;
  } else if (std::string(key) == "witness_cross_pair")
    g_witness_cross_pair = value ? 1 : 0;
  else if (std::string(key) == "witness_cross_maxr") {
    RS_REQUIRE(value >= 1 && value <= 6, "witness_cross_maxr must be in [1, 6]");
    g_witness_cross_maxr = value;
  } else if (std::string(key) == "witness_force_bc") {
    RS_REQUIRE(value == 0 || (value >= 5 && value <= 20), "witness_force_bc must be 0 or in [5, 20]");
    g_witness_force_bc = value;  // takes effect for plans built afterwards (plans are cached per context and size)
  } else if (std::string(key) == "witness_bc2") {
    g_witness_bc2 = value ? 1 : 0;  // takes effect for plans built afterwards, like witness_force_bc
  } else if (std::string(key) == "witness_inc") {
    g_witness_inc = value ? 1 : 0;  // incomplete transforms instead of block convolutions; plans built afterwards
  } else if (std::string(key) == "witness_tree_log") {
    RS_REQUIRE(value == 13 || value == 14, "witness_tree_log must be 13 or 14");
    g_witness_tree_log = value;
  } else if (std::string(key) == "witness_sub_ct") {
#ifndef RS_EXPERIMENTS
    // 0: generic kernel, 2: sub_ntt_wide_kernel (default).  The superseded A/B variants 1 (sub_ntt_ct_kernel) and 3
    // (sub_ntt_wide16_kernel) are compiled into the experiments build only (make -C ringsnark_amd/csrc experiments)
    if (value != 0 && value != 2) throw Error(RS_ERR_UNSUPPORTED, "witness_sub_ct 1 and 3 exist in the experiments build only");
#endif
    g_witness_sub_ct = value;
  }
  else if (std::string(key) == "witness_tree_ct")
    g_witness_tree_ct = value;
  else if (std::string(key) == "witness_tree_fwd")
    g_witness_tree_fwd = value ? 1 : 0;
  else if (std::string(key) == "witness_level_turn")
    g_witness_level_turn = value ? 1 : 0;
  else if (std::string(key) == "witness_h_turn")
    g_witness_h_turn = value ? 1 : 0;
  else if (std::string(key) == "witness_tree_once")
    g_witness_tree_once = value ? 1 : 0;
  else if (std::string(key) == "witness_big_ws_mib") {
    RS_REQUIRE(value >= 64, "witness_big_ws_mib must be at least 64");
    g_witness_big_ws_mib = value;
  } else if (std::string(key) == "witness_col_budget_mib") {
    RS_REQUIRE(value >= 1, "witness_col_budget_mib must be positive");
    g_witness_col_budget_mib = value;
  } else if (std::string(key) == "witness_lds_logM") {
    RS_REQUIRE(value >= 6 && value <= 13, "witness_lds_logM must be in [6, 13]");
    g_witness_lds_logM = value;
  }
  else
    throw Error(RS_ERR_INVALID, std::string("unknown tuning key ") + key);
  RS_API_END
}

int rs_set_profiling(rs_ctx *ctx, int enabled) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx, "null argument");
  ctx->profiling = enabled != 0;
  RS_API_END
}
int rs_profile_read(rs_ctx *ctx, rs_kernel_stat *out, int capacity, int *n_out) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && n_out && (out || capacity == 0), "null argument");
  std::unique_lock<std::mutex> lk(ctx->mu);
  std::vector<rs_kernel_stat> agg;
  for (auto &r : ctx->prof) {
    RS_HIP(hipEventSynchronize(r.e1));
    float ms = 0;
    RS_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
    ctx->prof_pool.push_back(r.e0);
    ctx->prof_pool.push_back(r.e1);
    size_t k = 0;
    for (; k < agg.size(); k++)
      if (strncmp(agg[k].name, r.name, sizeof(agg[k].name)) == 0) break;
    if (k == agg.size()) {
      rs_kernel_stat z;
      memset(&z, 0, sizeof(z));
      strncpy(z.name, r.name, sizeof(z.name) - 1);
      agg.push_back(z);
    }
    agg[k].launches++;
    agg[k].total_ms += ms;
    agg[k].alg_bytes += r.bytes;
    agg[k].fp64_ops += r.fp64;
  }
  ctx->prof.clear();
  std::sort(agg.begin(), agg.end(), [](const rs_kernel_stat &a, const rs_kernel_stat &b) { return a.total_ms > b.total_ms; });
  *n_out = (int)agg.size();
  for (int k = 0; k < capacity && k < (int)agg.size(); k++) out[k] = agg[k];
  RS_API_END
}
// ---- measured denominators of the rooflines ("use the measured copy bandwidth as the denominator too",
// "report achieved modmul/s against a measured modmul micro-benchmark peak"; the reference's own pattern: microbench.cpp:147-205)
}  // extern "C"
namespace rs {
// OP 0: v_fma_f64 chains; 1: the six-instruction exact FP64 modular multiply (f64mod.hpp); 2: the Montgomery product on
// 64-bit integers (intmod.hpp).  Eight independent chains per lane, 2048 workgroups of 256 threads (8 waves per SIMD).
constexpr int PEAK_ITERS = 2048;
template <int OP>
__global__ void __launch_bounds__(256) peak_rate_kernel(uint64_t *out, double a0, double b0, Mod mf, ModI mi) {
  double x[8];
  uint64_t y[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    x[i] = a0 + threadIdx.x + i;
    y[i] = (uint64_t)(threadIdx.x * 8 + i + 3) % mi.p;
  }
  const uint64_t c = (uint64_t)(b0 * 1e6) % mi.p;
  for (int it = 0; it < PEAK_ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) x[i] = __builtin_fma(x[i], b0, 0.3333333333333333);
      if (OP == 1) x[i] = mulmod(x[i], b0, mf);
      if (OP == 2) y[i] = montmul(y[i], c, mi);
    }
  }
  double sx = 0;
  uint64_t sy = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) sx += x[i], sy += y[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sy + (uint64_t)(long long)sx;
}
// streaming copy: a workgroup moves CONTIGUOUS 32 KiB pieces (eight 16-byte loads in flight per lane, then the eight
// stores), pieces dealt round-robin to the workgroups; NT: non-temporal accesses (n16 a multiple of 2048)
template <bool NT>
__global__ void __launch_bounds__(256) peak_copy_kernel(const u64x2 *__restrict__ src, u64x2 *__restrict__ dst, size_t n16) {
  for (size_t base = (size_t)blockIdx.x * 2048; base + 2048 <= n16; base += (size_t)gridDim.x * 2048) {
    u64x2 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = NT ? __builtin_nontemporal_load(src + base + threadIdx.x + 256 * k) : src[base + threadIdx.x + 256 * k];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (NT) __builtin_nontemporal_store(v[k], dst + base + threadIdx.x + 256 * k);
      else dst[base + threadIdx.x + 256 * k] = v[k];
    }
  }
}
// read-only stream (MODE 0: the sum of every word lands in one word per thread) and in-place update (MODE 1: every word
// + 1, written back where it was read) over the same contiguous 32 KiB pieces
template <bool NT, int MODE>
__global__ void __launch_bounds__(256) peak_stream_kernel(u64x2 *__restrict__ buf, uint64_t *__restrict__ sums, size_t n16) {
  uint64_t acc = 0;
  for (size_t base = (size_t)blockIdx.x * 2048; base + 2048 <= n16; base += (size_t)gridDim.x * 2048) {
    u64x2 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = NT ? __builtin_nontemporal_load(buf + base + threadIdx.x + 256 * k) : buf[base + threadIdx.x + 256 * k];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (MODE == 0) {
        acc += v[k].x ^ v[k].y;
      } else {
        v[k].x += 1;
        v[k].y += 1;
        if (NT) __builtin_nontemporal_store(v[k], buf + base + threadIdx.x + 256 * k);
        else buf[base + threadIdx.x + 256 * k] = v[k];
      }
    }
  }
  if (MODE == 0) sums[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}
}  // namespace rs

extern "C" {
int rs_measure_peaks(rs_ctx *ctx, rs_peaks *out, rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && out, "null argument");
  hipStream_t st = S(stream);
  memset(out, 0, sizeof(*out));
  struct Events {  // released on every path (a failed allocation below throws)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Events() {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
    }
  } ev;
  struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
      if (p) (void)hipFree(p);
    }
  };
  RS_HIP(hipEventCreate(&ev.e0));
  RS_HIP(hipEventCreate(&ev.e1));
  hipEvent_t &e0 = ev.e0, &e1 = ev.e1;
  auto timed = [&](auto &&launch, int reps) {
    launch();  // warm-up
    RS_HIP(hipEventRecord(e0, st));
    for (int r = 0; r < reps; r++) launch();
    RS_HIP(hipEventRecord(e1, st));
    RS_HIP(hipEventSynchronize(e1));
    float ms = 0;
    RS_HIP(hipEventElapsedTime(&ms, e0, e1));
    return (double)ms * 1e-3 / reps;
  };
  // device-to-device copy of 1 GiB (16-byte accesses): read + written bytes per second
  {
    const size_t bytes = (size_t)2 << 30;
    DevBuf buf_a, buf_b;
    RS_HIP(hipMalloc(&buf_a.p, bytes));
    RS_HIP(hipMalloc(&buf_b.p, bytes));
    void *a = buf_a.p, *b = buf_b.p;
    RS_HIP(hipMemsetAsync(a, 1, bytes, st));
    double best = 0;
    for (unsigned blocks : {256u * 2, 256u * 4, 256u * 8, 256u * 16}) {  // the best of four grid sizes x two access kinds (2 GiB read + 2 GiB written each)
      double sec = timed([&] { hipLaunchKernelGGL(peak_copy_kernel<true>, dim3(blocks), dim3(256), 0, st, (const u64x2 *)a, (u64x2 *)b, bytes / 16); }, 10);
      best = std::max(best, 2.0 * (double)bytes / sec / 1e9);
      sec = timed([&] { hipLaunchKernelGGL(peak_copy_kernel<false>, dim3(blocks), dim3(256), 0, st, (const u64x2 *)a, (u64x2 *)b, bytes / 16); }, 10);
      best = std::max(best, 2.0 * (double)bytes / sec / 1e9);
    }
    {  // ... and the runtime's own device-to-device copy
      const double sec = timed([&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, st); }, 10);
      best = std::max(best, 2.0 * (double)bytes / sec / 1e9);
    }
    out->hbm_copy_gbs = best;
    // the two other shapes the streaming kernels have: read only (the inner products read their key once) and
    // in place (the passes over the witness map's workspaces)
    double best_r = 0, best_i = 0;
    for (unsigned blocks : {256u * 2, 256u * 4, 256u * 8, 256u * 16}) {
      double sec = timed([&] { hipLaunchKernelGGL((peak_stream_kernel<true, 0>), dim3(blocks), dim3(256), 0, st, (u64x2 *)a, (uint64_t *)b, bytes / 16); }, 10);
      best_r = std::max(best_r, (double)bytes / sec / 1e9);
      sec = timed([&] { hipLaunchKernelGGL((peak_stream_kernel<false, 0>), dim3(blocks), dim3(256), 0, st, (u64x2 *)a, (uint64_t *)b, bytes / 16); }, 10);
      best_r = std::max(best_r, (double)bytes / sec / 1e9);
      sec = timed([&] { hipLaunchKernelGGL((peak_stream_kernel<true, 1>), dim3(blocks), dim3(256), 0, st, (u64x2 *)a, (uint64_t *)b, bytes / 16); }, 10);
      best_i = std::max(best_i, 2.0 * (double)bytes / sec / 1e9);
      sec = timed([&] { hipLaunchKernelGGL((peak_stream_kernel<false, 1>), dim3(blocks), dim3(256), 0, st, (u64x2 *)a, (uint64_t *)b, bytes / 16); }, 10);
      best_i = std::max(best_i, 2.0 * (double)bytes / sec / 1e9);
    }
    out->hbm_read_gbs = best_r;
    out->hbm_inplace_gbs = best_i;
  }
  {
    const unsigned blocks = 256 * 8;
    DevBuf buf_d;
    RS_HIP(hipMalloc(&buf_d.p, (size_t)blocks * 256 * sizeof(uint64_t)));
    uint64_t *d = (uint64_t *)buf_d.p;
    const Mod mf = HostArith<Mod>::make(ctx->Q[0] < (1ull << 50) ? ctx->Q[0] : 1125899906826241ull);
    const ModI mi = HostArith<ModI>::make(1152921504606830593ull);  // a 60-bit prime (microbench.cpp:35-36 sizes)
    const double lanes = (double)blocks * 256.0 * PEAK_ITERS * 8.0;  // lane-operations of the inner statement per launch
    out->fp64_fma_T = lanes / timed([&] { hipLaunchKernelGGL(peak_rate_kernel<0>, dim3(blocks), dim3(256), 0, st, d, 1.5, 1.0000001, mf, mi); }, 3) / 1e12;
    out->fp64_mulmod_G = lanes / timed([&] { hipLaunchKernelGGL(peak_rate_kernel<1>, dim3(blocks), dim3(256), 0, st, d, 1.5, 12345.0, mf, mi); }, 3) / 1e9;
    out->int_montmul_G = lanes / timed([&] { hipLaunchKernelGGL(peak_rate_kernel<2>, dim3(blocks), dim3(256), 0, st, d, 1.5, 12345.0, mf, mi); }, 3) / 1e9;
  }
  RS_HIP(hipGetLastError());
  RS_API_END
}

int rs_last_timings(rs_ctx *ctx, rs_timings *out) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && out, "null argument");
  *out = ctx->timings;
  RS_API_END
}

}  // extern "C"
