// ntt_wide.hpp -- the negacyclic transform with wide per-thread radix groups (FP64 arithmetic).
//
// ntt_core.hpp's rounds give a thread 2^R <= 16 elements and a length-8192 transform five LDS round
// trips; measured, those kernels spend time(VALU) + time(LDS) because the waves of a workgroup move through
// the exchange and butterfly phases together.  Here a polynomial of n = 2^LOGN residues belongs to
// T = n/32 threads holding 32 elements each, and the stages split into THREE rounds (S = n/16):
//
//   forward   round 1  stages 0..3             two radix-16 groups per thread on elements 2t+c + S*e (c = 0,1):
//                                              the thread's 16-byte global loads ARE its operands; twiddles are
//                                              wave-uniform (scalar operands)
//             round 2  stages 4..LOGN-5        radix-2^(LOGN-8) groups base + 16*e; twiddles from an LDS table
//             round 3  stages LOGN-4..LOGN-1   radix-16 groups of 16 CONSECUTIVE elements; each thread always
//                                              owns the same two groups, so its 2 x 15 twiddles stay in registers
//                                              for the whole (persistent) kernel
//   inverse   the mirror image (Gentleman-Sande): contiguous groups first, the S-strided round last, n^-1 folded
//             into the last stage's two multipliers.
//
// so the tile is exchanged twice instead of five times, plus one wave-private pass that turns round 3's
// 128-byte-per-lane runs into coalesced 16-byte-per-lane global accesses.  Address arithmetic is amortised over
// 16..32 elements and every stage offers 8..16 independent butterflies per thread to hide FP64 latency at 2 waves
// per SIMD.
//
// The kernels are persistent (grid = workgroups that fit the chip) and software-pipelined through registers:
// the next polynomial's loads are issued a whole transform ahead; stores of a finished polynomial and the loads
// that follow are issued back to back, stores first, because loads and stores share vmcnt on this ISA and a
// wait for a load issued after stores drains those stores.
#pragma once
#include "ntt_core.hpp"

namespace rs {

template <int LOGN>
struct WideShape {
  static constexpr int N = 1 << LOGN;
  static constexpr int T = N / 32;      // threads per polynomial, 32 elements each
  static constexpr int S = N / 16;      // element stride of round 1's radix-16 groups
  // Round shapes: 4096 / 8192 points run 4 | LOGN-8 | 4 stages, the last round on two groups of 16 consecutive elements
  // per thread; 16384 points (512 threads, one workgroup per CU) run 4 | 5 | 5, the last round on ONE group of 32
  // consecutive elements per thread.
  static constexpr int R3 = LOGN == 14 ? 5 : 4;   // stages of the round on consecutive elements
  static constexpr int R2 = LOGN - 4 - R3;        // stages of the middle round
  static constexpr int E2 = 1 << R2;    // its radix
  static constexpr int G2 = 32 / E2;    // middle-round groups per thread
  static constexpr int E3 = 1 << R3;    // consecutive elements per group of the last round
  static constexpr int G3 = 32 / E3;    // its groups per thread (thread t: groups t + T*j)
  static constexpr int ST2 = E3;        // element stride inside a middle-round group
  static constexpr int TWL = 1 << (4 + R2);  // LDS twiddle table: entries [0, n/16)
  static_assert(LOGN == 12 || LOGN == 13 || LOGN == 14, "shapes: 4096, 8192 and 16384 points");
  // LDS address map: one pad slot per 16 elements (round 3: lane stride 17), plus 16 slots per S elements when
  // S + S/16 would otherwise be a multiple of the 32 bank pairs (round 2 reads 16-lane runs that are S apart)
  static constexpr int PAD2 = ((S + S / 16) % 32 == 0) ? 16 : 0;
  static constexpr int SP = S + S / 16 + PAD2;  // mapped distance of elements S apart
  __host__ __device__ static constexpr int px(int i) { return i + (i >> 4) + (i / S) * PAD2; }
  // mapped offset of element base + ST2*e from element base, base = (multiple of S) + (less than ST2)
  __host__ __device__ static constexpr int px2(int e) { return ST2 * e + ((ST2 * e) >> 4); }
  // mapped offset of element r + 128*i from element r, r = (multiple of 1024) + (less than 128)
  __host__ __device__ static constexpr int px128(int i) { return 136 * i + ((128 * i) / S) * PAD2; }
  static constexpr int TILE = N + N / 16 + 16 * PAD2;
  static constexpr size_t LDS_BYTES = (size_t)(TILE + TWL) * sizeof(double);
};

// Group j of thread t in the round on consecutive elements.  One middle-round group per thread (8192, 16384 points): the
// thread's groups are NEIGHBOURS (32 consecutive elements), inside the S-element block its middle-round group lies in, so
// the exchange between the two rounds stays within the block's 16 / 32 consecutive threads.  4096 points: groups t, t + T.
template <int LOGN>
__device__ __forceinline__ int wide_group(int t, int j) {
  using S = WideShape<LOGN>;
  return S::G2 == 1 ? S::G3 * t + j : t + S::T * j;
}

// R forward stages on a register tile of 2^R values; tw(k, blk) = twiddle of block blk of local stage k.
// NST < R: only the first NST of them (incomplete transforms, witness_inc.hpp).
template <int R, bool RED, int NST = R, class TwFn>
__device__ __forceinline__ void reg_fwd_stages(double (&v)[1 << R], const Mod mod, uint32_t red_mask, TwFn tw) {
  constexpr int E = 1 << R;
#pragma unroll
  for (int k = 0; k < NST; k++) {
    if (RED && ((red_mask >> k) & 1u)) {
#pragma unroll
      for (int e = 0; e < E; e++) v[e] = reduce(v[e], mod);
    }
    const int half = E >> (k + 1);
#pragma unroll
    for (int blk = 0; blk < (1 << k); blk++) {
      const double w = tw(k, blk);
#pragma unroll
      for (int e0 = 0; e0 < half; e0++) {
        const int ia = blk * 2 * half + e0, ib = ia + half;
        const double t = mulmod(v[ib], w, mod);
        const double a = v[ia];
        v[ia] = a + t;
        v[ib] = a - t;
      }
    }
  }
}
// Inverse stages K0..NST-1 of a register tile of 2^R values (local gap 2^k at local stage k);
// tw(k, i) = twiddle of butterfly block i = e >> (k+1).  K0 > 0: incomplete transforms (witness_inc.hpp).
template <int R, bool RED, int NST = R, int K0 = 0, class TwFn>
__device__ __forceinline__ void reg_inv_stages(double (&v)[1 << R], const Mod mod, uint32_t red_mask, TwFn tw) {
  constexpr int E = 1 << R;
#pragma unroll
  for (int k = K0; k < NST; k++) {
    if (RED && ((red_mask >> k) & 1u)) {
#pragma unroll
      for (int e = 0; e < E; e++) v[e] = reduce(v[e], mod);
    }
#pragma unroll
    for (int e = 0; e < E; e++) {
      if (e & (1 << k)) continue;
      const double w = tw(k, e >> (k + 1));
      const double a = v[e], b = v[e + (1 << k)];
      v[e] = a + b;
      v[e + (1 << k)] = mulmod(a - b, w, mod);
    }
  }
}

// 16-byte streaming accesses.  RS_WIDE_NT = 1 marks them non-temporal (the data is touched once per launch).
#ifndef RS_WIDE_NT
#define RS_WIDE_NT 1
#endif
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u64x2 stream_load(const u64x2 *p) {
#if RS_WIDE_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ void stream_store(u64x2 *p, u64x2 v) {
#if RS_WIDE_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// A table constant at a WAVE-UNIFORM address, through the scalar data cache (s_load_*: lgkmcnt, not vmcnt).  Inside a
// persistent kernel that also stores, the compiler reads such a constant with a VECTOR load (the table is not provably
// invariant across the loop's stores), and the s_waitcnt vmcnt(0) before its first use drains every vector load issued before
// it -- the prefetch of the next block included (found in the ISA of sub_ntt_wide_kernel, round 6).  A load through the
// CONSTANT address space is invariant by definition: uniform address -> SMEM.  The tables are written once, at plan / context
// creation, long before any kernel reads them.
template <class T>
__device__ __forceinline__ T ld_const(const T *p) {
  typedef const T __attribute__((address_space(4))) *cptr4;
  return *(cptr4)(unsigned long long)p;
}
// a wave-uniform value as scalar registers
__device__ __forceinline__ double uniform_f64(double x) {
  union { double d; int i[2]; } u;
  u.d = x;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
  u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.d;
}
__device__ __forceinline__ double u64_bits_as_double(uint64_t u) {
  union { uint64_t u; double d; } x;
  x.u = u;
  return x.d;
}
__device__ __forceinline__ uint64_t double_bits_as_u64(double d) {
  union { uint64_t u; double d; } x;
  x.d = d;
  return x.u;
}

// Scheduling fences.  pin(x): x is materialised HERE (a load feeding it has been waited for) and volatile asms keep
// their order, so a sequence of pins fixes the order in which prefetched registers are consumed.  mem_fence():
// no memory operation moves across.  Together they keep "wait for the prefetched loads -> issue the stores ->
// issue the next loads" in exactly that order (see the header comment), which the scheduler otherwise undoes.
__device__ __forceinline__ void pin(double &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void mem_fence() { asm volatile("" ::: "memory"); }

// First element of the j-th (j = 0, 1) 1024-element range covered by the consecutive-element groups of wave `wave`:
// 32 consecutive elements per thread (8192 points: two neighbouring groups of 16; 16384 points: one group of 32), or --
// 4096 points -- the groups t and t + T.
template <int LOGN>
__device__ __forceinline__ int wide_wave_range(int wave, int j) {
  using S = WideShape<LOGN>;
  return (S::G3 == 2 && S::G2 != 1) ? (j * S::T + wave * 64) * 16 : (2 * wave + j) * 1024;
}

// ---- forward ------------------------------------------------------------------------------------------------
// Output of a polynomial (canonical u64 bit patterns parked in the tile by round 3): every wave streams out the
// two 1024-element ranges its own round-3 groups cover, 16 bytes per lane.
template <int LOGN>
__device__ __forceinline__ void wide_fwd_flush(const double *s, uint64_t *__restrict__ dst) {
  using S = WideShape<LOGN>;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int r0 = wide_wave_range<LOGN>(wave, j);
    const int p0 = S::px(r0 + 2 * lane);
    u64x2 *d2 = reinterpret_cast<u64x2 *>(dst + r0) + lane;
#pragma unroll
    for (int i = 0; i < 8; i++) {  // elements r0 + 2*lane + 128*i (+1)
      u64x2 o;
      o.x = double_bits_as_u64(s[p0 + S::px128(i)]);
      o.y = double_bits_as_u64(s[p0 + S::px128(i) + 1]);
      stream_store(d2 + 64 * i, o);
    }
  }
}

template <int LOGN, bool RED>
__device__ __forceinline__ void wide_fwd_body(double *s, const double *twl, const double (&tw3)[WideShape<LOGN>::G3][WideShape<LOGN>::E3 - 1],
                                              double (&v)[2][16], const double *__restrict__ tw, const Mod mod, uint32_t red_mask) {
  using S = WideShape<LOGN>;
  const int t = threadIdx.x;
  // round 1: stages 0..3 on elements 2t+c + S*e, twiddles tw[2^k + blk] (the same for every thread)
#pragma unroll
  for (int c = 0; c < 2; c++)
    reg_fwd_stages<4, RED>(v[c], mod, red_mask, [&](int k, int blk) { return tw[(1 << k) + blk]; });
  __syncthreads();  // every wave has streamed the previous polynomial out of the tile
  {
    const int pb = S::px(2 * t);
#pragma unroll
    for (int e = 0; e < 16; e++) {
      s[pb + S::SP * e] = v[0][e];
      s[pb + S::SP * e + 1] = v[1][e];
    }
  }
  __syncthreads();
  // round 2: stages 4..4+R2-1 on groups base + ST2*e, base = hi*S + lo
#pragma unroll
  for (int j = 0; j < S::G2; j++) {
    const int g = t + S::T * j, lo = g & (S::ST2 - 1), hi = g / S::ST2;
    const int pb = hi * S::SP + S::px(lo);
    double x[S::E2];
#pragma unroll
    for (int e = 0; e < S::E2; e++) x[e] = s[pb + S::px2(e)];
    reg_fwd_stages<S::R2, RED>(x, mod, red_mask >> 4, [&](int k, int blk) { return twl[(16 << k) + (hi << k) + blk]; });
#pragma unroll
    for (int e = 0; e < S::E2; e++) s[pb + S::px2(e)] = x[e];
  }
  // 8192 / 16384 points: a thread's round-2 group and its round-3 groups lie in the same S-element block, which 16 / 32
  // consecutive threads own in both rounds -- no workgroup barrier (4096 points: two round-2 groups per thread, two blocks)
  if (S::G2 == 1) wave_sync(); else __syncthreads();
  // round 3: stages LOGN-R3..LOGN-1 on E3 consecutive elements; results parked canonical for the flush
#pragma unroll
  for (int j = 0; j < S::G3; j++) {
    const int g = wide_group<LOGN>(t, j);
    const int pb = S::px(S::E3 * g);
    double x[S::E3];
#pragma unroll
    for (int e = 0; e < S::E3; e++) x[e] = s[pb + e + (e >> 4)];
    reg_fwd_stages<S::R3, RED>(x, mod, red_mask >> (LOGN - S::R3), [&](int k, int blk) { return tw3[j][(1 << k) - 1 + blk]; });
#pragma unroll
    for (int e = 0; e < S::E3; e++) s[pb + e + (e >> 4)] = u64_bits_as_double(to_u64(canon(x[e], mod)));
  }
  wave_sync();
}


template <int LOGN, bool RED>
__global__ void __launch_bounds__(WideShape<LOGN>::T, 2)
ntt_fwd_wide_kernel(uint64_t *__restrict__ data, unsigned long long batch, const double *__restrict__ tw, Mod mod,
                    uint32_t red_mask) {
  using S = WideShape<LOGN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  double *twl = s + S::TILE;
  const int t = threadIdx.x;
  for (int i = t; i < S::TWL; i += S::T) twl[i] = tw[i];
  double tw3[S::G3][S::E3 - 1];
#pragma unroll
  for (int j = 0; j < S::G3; j++) {
    const int g = wide_group<LOGN>(t, j);
#pragma unroll
    for (int k = 0; k < S::R3; k++)
#pragma unroll
      for (int blk = 0; blk < (1 << k); blk++) tw3[j][(1 << k) - 1 + blk] = tw[(1 << (LOGN - S::R3 + k)) + (g << k) + blk];
  }
  // the loop must not inherit these loads as "possibly still in flight" (it would wait for ALL memory traffic,
  // prefetch included, at their first use in every iteration)
#pragma unroll
  for (int j = 0; j < S::G3; j++)
#pragma unroll
    for (int i = 0; i < S::E3 - 1; i++) pin(tw3[j][i]);
  u64x2 pre[16];
  auto issue_loads = [&](unsigned long long q) {
    const u64x2 *src = reinterpret_cast<const u64x2 *>(data + q * (size_t)S::N) + t;
#pragma unroll
    for (int e = 0; e < 16; e++) pre[e] = stream_load(src + (S::S / 2) * e);
  };
  unsigned long long p = blockIdx.x;
  if (p < batch) issue_loads(p);
  unsigned long long p_out = ~0ull;
  for (; p < batch; p += gridDim.x) {
    double v[2][16];
#pragma unroll
    for (int e = 0; e < 16; e++) {
      v[0][e] = from_u64(pre[e].x);
      v[1][e] = from_u64(pre[e].y);
      pin(v[0][e]);
      pin(v[1][e]);
    }
    mem_fence();
#ifndef RS_WIDE_EXP_NOMEM
    if (p_out != ~0ull) wide_fwd_flush<LOGN>(s, data + p_out * (size_t)S::N);
    mem_fence();
    const unsigned long long pn = p + gridDim.x;
    if (pn < batch) issue_loads(pn);
    mem_fence();
#endif
    wide_fwd_body<LOGN, RED>(s, twl, tw3, v, tw, mod, red_mask);
    p_out = p;
  }
  if (p_out != ~0ull) wide_fwd_flush<LOGN>(s, data + p_out * (size_t)S::N);
}

// ---- inverse ------------------------------------------------------------------------------------------------
template <int LOGN, bool RED>
__global__ void __launch_bounds__(WideShape<LOGN>::T, 2)
ntt_inv_wide_kernel(uint64_t *__restrict__ data, unsigned long long batch, const double *__restrict__ itw, Mod mod,
                    double ninv, uint32_t red_mask) {
  using S = WideShape<LOGN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  double *twl = s + S::TILE;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  for (int i = t; i < S::TWL; i += S::T) twl[i] = itw[i];
  // round 1 (inverse stages 0..3 on 16 consecutive elements): block i of stage k of group g is
  // (16 g + e) >> (k+1) = (g << (3-k)) + (e >> (k+1)) among the n >> (k+1) blocks of that stage
  double tw1[S::G3][S::E3 - 1];
#pragma unroll
  for (int j = 0; j < S::G3; j++) {
    const int g = wide_group<LOGN>(t, j);
#pragma unroll
    for (int k = 0; k < S::R3; k++)
#pragma unroll
      for (int i = 0; i < (S::E3 >> (k + 1)); i++) tw1[j][S::E3 - (S::E3 >> k) + i] = itw[(S::N >> (k + 1)) + (g << (S::R3 - 1 - k)) + i];
  }
#pragma unroll
  for (int j = 0; j < S::G3; j++)
#pragma unroll
    for (int i = 0; i < S::E3 - 1; i++) pin(tw1[j][i]);
  // n^-1 folded into the last stage: (a + b) * ninv and (a - b) * (w * ninv)
  const double w_last = mulmod(itw[1], ninv, mod);
  u64x2 pre[16];
  auto issue_loads = [&](unsigned long long q) {
    const uint64_t *src = data + q * (size_t)S::N;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int r0 = wide_wave_range<LOGN>(wave, j);
      const u64x2 *s2 = reinterpret_cast<const u64x2 *>(src + r0) + lane;
#pragma unroll
      for (int i = 0; i < 8; i++) pre[j * 8 + i] = stream_load(s2 + 64 * i);
    }
  };
  double v[2][16];
  auto store_out = [&](unsigned long long q) {
    u64x2 *dst = reinterpret_cast<u64x2 *>(data + q * (size_t)S::N) + t;
#pragma unroll
    for (int e = 0; e < 16; e++) {
      u64x2 o;
      o.x = to_u64(canon(v[0][e], mod));
      o.y = to_u64(canon(v[1][e], mod));
      stream_store(dst + (S::S / 2) * e, o);
    }
  };
  unsigned long long p = blockIdx.x;
  if (p < batch) issue_loads(p);
  unsigned long long p_out = ~0ull;
  for (; p < batch; p += gridDim.x) {
    // the tile is free: every wave passed the barrier that follows its round-3 reads of the previous polynomial
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int r0 = wide_wave_range<LOGN>(wave, j);
      const int p0 = S::px(r0 + 2 * lane);
#pragma unroll
      for (int i = 0; i < 8; i++) {
        s[p0 + S::px128(i)] = from_u64(pre[j * 8 + i].x);
        s[p0 + S::px128(i) + 1] = from_u64(pre[j * 8 + i].y);
      }
    }
    mem_fence();
    if (p_out != ~0ull) store_out(p_out);
    mem_fence();
    const unsigned long long pn = p + gridDim.x;
    if (pn < batch) issue_loads(pn);
    mem_fence();
    wave_sync();
    // round 1: inverse stages 0..R3-1 on E3 consecutive elements
#pragma unroll
    for (int j = 0; j < S::G3; j++) {
      const int g = wide_group<LOGN>(t, j);
      const int pb = S::px(S::E3 * g);
      double x[S::E3];
#pragma unroll
      for (int e = 0; e < S::E3; e++) x[e] = s[pb + e + (e >> 4)];
      reg_inv_stages<S::R3, RED>(x, mod, red_mask, [&](int k, int i) { return tw1[j][S::E3 - (S::E3 >> k) + i]; });
#pragma unroll
      for (int e = 0; e < S::E3; e++) s[pb + e + (e >> 4)] = x[e];
    }
    if (S::G2 == 1) wave_sync(); else __syncthreads();  // 8192 / 16384 points: rounds 1 and 2 share their 16- / 32-thread blocks
    // round 2: inverse stages R3..R3+R2-1 on groups hi*S + lo + ST2*e; block of stage R3+k: (hi << (R2-1-k)) + (e >> (k+1))
#pragma unroll
    for (int j = 0; j < S::G2; j++) {
      const int g = t + S::T * j, lo = g & (S::ST2 - 1), hi = g / S::ST2;
      const int pb = hi * S::SP + S::px(lo);
      double x[S::E2];
#pragma unroll
      for (int e = 0; e < S::E2; e++) x[e] = s[pb + S::px2(e)];
      reg_inv_stages<S::R2, RED>(x, mod, red_mask >> S::R3,
                                 [&](int k, int i) { return twl[(S::N >> (S::R3 + 1 + k)) + (hi << (S::R2 - 1 - k)) + i]; });
#pragma unroll
      for (int e = 0; e < S::E2; e++) s[pb + S::px2(e)] = x[e];
    }
    __syncthreads();
    // round 3: inverse stages LOGN-4..LOGN-1 on elements 2t+c + S*e; block of stage LOGN-4+k is e >> (k+1) of 8 >> k
    {
      const int pb = S::px(2 * t);
#pragma unroll
      for (int e = 0; e < 16; e++) {
        v[0][e] = s[pb + S::SP * e];
        v[1][e] = s[pb + S::SP * e + 1];
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; c++) {
      reg_inv_stages<4, RED, 3>(v[c], mod, red_mask >> (LOGN - 4), [&](int k, int i) { return itw[(8 >> k) + i]; });
      // last stage (gap n/2) with the scaling folded in; a + b is a legal multiplier operand by the same bound the
      // reduction mask keeps for a - b (inv_reduce_mask)
      if (RED && ((red_mask >> (LOGN - 1)) & 1u)) {
#pragma unroll
        for (int e = 0; e < 16; e++) v[c][e] = reduce(v[c][e], mod);
      }
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const double a = v[c][e], b = v[c][e + 8];
        v[c][e] = mulmod(a + b, ninv, mod);
        v[c][e + 8] = mulmod(a - b, w_last, mod);
      }
    }
    p_out = p;
  }
  if (p_out != ~0ull) store_out(p_out);
}

}  // namespace rs
