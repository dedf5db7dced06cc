// msm_mac.hpp -- multiply-accumulate kernels of the inner product (row a9): generic, streaming, half / quarter spectrum, two key vectors; partial-sum reduction (msm.hip)
#pragma once
#include "ntt_core.hpp"
#include "ntt_wide.hpp"
#include "rs_internal.hpp"
#include "msm_plain.hpp"

namespace rs {

template <class L_>
struct alignas(16) LiftPair {
  L_ x, y;
};
struct MacArgs {
  const void *C[2];       // per group: [tile_terms][L][n] centred plaintext integers (double or int64_t)
  const uint64_t *crs[2]; // per CRS vector: element 0 of the tile, [terms][L][2][K][n]
  uint64_t *partial;      // [n_chunks][n_sets_total][L][2][K][n]
  int set_index[4];       // which set slot (c * n_groups + g) each accumulator set writes
  int n_sets_total;
  unsigned long long terms[2];  // per group: number of valid terms in this tile
  unsigned long long tile_terms;
  int terms_per_chunk, n_chunks;
  int accumulate;       // 1: add onto the existing partial slot
  int acc_period;       // terms between lazy reductions of the accumulators
  int reduce_u;         // 1: bring NTT outputs back to |u| <= p/2 before the MAC (large primes)
};

// The dominant kernel.  Accumulator set (c, g): sum_t crs[c][t] * NTT(C[g][t]).
template <int NG, int NC, int PAIRS, class M = Mod>
__global__ void __launch_bounds__(1024)
mac_kernel(MacArgs a, int L, int K, int logn, const NttTableT<typename ArithOf<M>::T, M> *__restrict__ coeff_tabs) {
  using T = typename ArithOf<M>::T;
  using Lift = typename ArithOf<M>::Lift;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *s = reinterpret_cast<T *>(smem);
  const int n = 1 << logn;
  // XCD-aware mapping: blocks b, b+8, b+16, ... (same XCD, dispatched back to back) take the K
  // primes of one (limb, chunk), so the K readers of a C row share it through that XCD's L2.
  const unsigned b = blockIdx.x;
  const int j = (int)((b >> 3) % (unsigned)K);
  const unsigned r = (b & 7u) + 8u * (b / (8u * (unsigned)K));
  if (r >= (unsigned)(a.n_chunks * L)) return;
  const int limb = (int)(r % (unsigned)L), chunk = (int)(r / (unsigned)L);
  const NttTableT<T, M> tab = coeff_tabs[j];
  const M mod = tab.mod;
  constexpr int NS = NG * NC;
  // PAIRS = n / (2 * blockDim): 4, or 8 for N_enc = 16384
  T acc[NS][2][2 * PAIRS];
  const size_t enc_words = (size_t)L * 2 * K * n;
#pragma unroll
  for (int st = 0; st < NS; st++)
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int k = 0; k < PAIRS; k++) {
        acc[st][c][2 * k] = acc[st][c][2 * k + 1] = T(0);
        const int pp = threadIdx.x + k * blockDim.x;
        if (a.accumulate && pp < (n >> 1)) {
          const uint64_t *pv = a.partial + ((size_t)chunk * a.n_sets_total + a.set_index[st]) * enc_words +
                               (((size_t)limb * 2 + c) * K + j) * (size_t)n;
          const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(pv)[pp];
          acc[st][c][2 * k] = from_res<T>(v.x);
          acc[st][c][2 * k + 1] = from_res<T>(v.y);
        }
      }
  const unsigned long long tbeg = (unsigned long long)chunk * a.terms_per_chunk;
  const unsigned long long tend = min(tbeg + (unsigned long long)a.terms_per_chunk, a.tile_terms);
  int since = 0;
  for (unsigned long long t = tbeg; t < tend; t++) {
#pragma unroll
    for (int g = 0; g < NG; g++) {
      if (t >= a.terms[g]) continue;
      // centred plaintext (|c| < members * q_i / 2) -> residues mod Q_j.  Integer arithmetic: the residue goes
      // in as c*R, the transform is linear, so the spectrum comes out in Montgomery form and ct * spectrum below
      // is ONE reduction per product.
      const LiftPair<Lift> *src = reinterpret_cast<const LiftPair<Lift> *>(static_cast<const Lift *>(a.C[g]) + ((size_t)t * L + limb) * (size_t)n);
      for (int pp = threadIdx.x; pp < (n >> 1); pp += blockDim.x) {
        const LiftPair<Lift> v = src[pp];
        const int pi = pidx(2 * pp);
        s[pi] = to_mont(lift_residue(v.x, mod), mod);
        s[pnext(pi)] = to_mont(lift_residue(v.y, mod), mod);
      }
      __syncthreads();
      lds_ntt_fwd(s, logn, tab.d_tw, 1, mod, tab.fwd_red_mask);
#pragma unroll
      for (int k = 0; k < PAIRS; k++) {
        const int pp = threadIdx.x + k * blockDim.x;
        if (pp < (n >> 1)) {
          const int pi = pidx(2 * pp);
          T u0 = s[pi], u1 = s[pnext(pi)];
          if (a.reduce_u) {
            u0 = reduce(u0, mod);
            u1 = reduce(u1, mod);
          }
#pragma unroll
          for (int cc = 0; cc < NC; cc++) {
#pragma unroll
            for (int c = 0; c < 2; c++) {
              const uint64_t *ct = a.crs[cc] + (size_t)t * enc_words + (((size_t)limb * 2 + c) * K + j) * (size_t)n;
              const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(ct)[pp];
              const int st = cc * NG + g;
              acc[st][c][2 * k] = addm(acc[st][c][2 * k], mulmod(from_res<T>(v.x), u0, mod), mod);
              acc[st][c][2 * k + 1] = addm(acc[st][c][2 * k + 1], mulmod(from_res<T>(v.y), u1, mod), mod);
            }
          }
        }
      }
      __syncthreads();
    }
    if (++since >= a.acc_period) {
      since = 0;
#pragma unroll
      for (int st = 0; st < NS; st++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
          for (int k = 0; k < 2 * PAIRS; k++) acc[st][c][k] = reduce(acc[st][c][k], mod);
    }
  }
#pragma unroll
  for (int st = 0; st < NS; st++)
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int k = 0; k < PAIRS; k++) {
        const int pp = threadIdx.x + k * blockDim.x;
        if (pp < (n >> 1)) {
          uint64_t *pv = a.partial + ((size_t)chunk * a.n_sets_total + a.set_index[st]) * enc_words +
                         (((size_t)limb * 2 + c) * K + j) * (size_t)n;
          ulonglong2 o;
          o.x = to_res(canon(acc[st][c][2 * k], mod));
          o.y = to_res(canon(acc[st][c][2 * k + 1], mod));
          reinterpret_cast<ulonglong2 *>(pv)[pp] = o;
        }
      }
}

// mac_kernel_v2: the streaming form of the dominant kernel (one accumulator set per launch,
// 2048 <= N_enc <= 8192).  Differences from mac_kernel:
//   * the twiddle table of Q_j lives in LDS next to the tile (68 + 64 KiB), so inside the term loop
//     the ONLY vector-memory operations are the streamed operands;
//   * the ciphertext words of term t are loaded into registers before the transform of term t
//     starts (512-thread shape: also the plaintext row of term t+1), so HBM latency and transfer
//     overlap the FP64 work -- provided nothing in the loop forces an early s_waitcnt vmcnt(0);
//   * the transform is the wave-private form (ntt_core.hpp): the cross-wave stages, then each wave
//     finishes its own block and multiplies exactly that block into its accumulators -- two or
//     three workgroup barriers per term instead of seven.
// radix of the wave-private rounds inside mac_kernel_v2: 3 keeps the kernel free of VGPR spills (a
// scratch reload inside the term loop costs an s_waitcnt vmcnt(0), which drains the prefetched
// ciphertext loads and serialises stream and transform)
#ifndef RS_MAC_MAXR
#define RS_MAC_MAXR 3
#endif
struct MacArgs2 {
  const double *C;      // [tile_terms][L][n] plaintext rows of the group
  const uint64_t *crs;  // first ciphertext of the tile
  uint64_t *partial;    // accumulator slot 0 of this set: [n_chunks] stride part_stride
  size_t part_stride;   // words between consecutive chunks
  unsigned long long terms;  // valid terms in this tile
  int terms_per_chunk, n_chunks;
  int accumulate, acc_period, reduce_u;
};
// THREADS = 1024 (default at N_enc = 8192): 16 waves, 118 VGPRs, four waves per SIMD, the plaintext
// row loaded where it is used.  THREADS = 512: 8 waves, ~240 VGPRs, the plaintext row of term t+1
// prefetched as well.  Both use radix-8 private rounds (RS_MAC_MAXR) to stay free of scratch.
// LOGN_CT != 0: transform length fixed at compile time (rounds specialised).  ABLATE (experiments,
// tools/mac_ablate.py): 1 = skip the transform, 2 = skip the ciphertext loads, 4 = skip the C loads;
// a compile-time parameter because a run-time branch around each load makes the compiler wait for
// every load right where it is issued.
template <int THREADS, int LOGN_CT = 0, int ABLATE = 0>
__global__ void __launch_bounds__(THREADS)
mac_kernel_v2(MacArgs2 a, int L, int K, int logn_arg, const NttTable *__restrict__ coeff_tabs) {
  const int logn = LOGN_CT ? LOGN_CT : logn_arg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  constexpr int LOGW = THREADS == 512 ? 3 : 4;
  constexpr int PP = 8192 / (2 * THREADS);  // coefficient pairs per lane at n = 8192 (fewer for smaller n)
  const int n = 1 << logn;
  double *twl = s + padded_len((size_t)n);
  const unsigned b = blockIdx.x;
  const int j = (int)((b >> 3) % (unsigned)K);
  const unsigned r = (b & 7u) + 8u * (b / (8u * (unsigned)K));
  if (r >= (unsigned)(a.n_chunks * L)) return;
  const int limb = (int)(r % (unsigned)L), chunk = (int)(r / (unsigned)L);
  const Mod mod = coeff_tabs[j].mod;
  const uint32_t red_mask = coeff_tabs[j].fwd_red_mask;
  {
    const double *gtw = coeff_tabs[j].d_tw;
    for (int i = threadIdx.x; i < n; i += THREADS) twl[i] = gtw[i];
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int bpairs = n >> (LOGW + 1);      // pairs per wave block
  const int pbase = wave * bpairs + lane;  // + 64*k, k < PP
  const size_t enc_words = (size_t)L * 2 * K * n;
  const size_t slab = (((size_t)limb * 2) * K + j) * (size_t)n;  // component 0; component 1 is + K*n
  const size_t comp = (size_t)K * n;
  uint64_t *part = a.partial + (size_t)chunk * a.part_stride + slab;
  double acc[2][2 * PP];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int k = 0; k < PP; k++) {
      acc[c][2 * k] = acc[c][2 * k + 1] = 0.0;
      if (a.accumulate && 64 * k + lane < bpairs) {
        const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(part + c * comp)[pbase + 64 * k];
        acc[c][2 * k] = from_u64(v.x);
        acc[c][2 * k + 1] = from_u64(v.y);
      }
    }
  const unsigned long long tbeg = (unsigned long long)chunk * a.terms_per_chunk;
  const unsigned long long tend = min(tbeg + (unsigned long long)a.terms_per_chunk, a.terms);
  const TileBlockFactory bf{s};
  const LdsIO lds{s};
  int since = 0;
  const double2 *crow = reinterpret_cast<const double2 *>(a.C + ((size_t)tbeg * L + limb) * (size_t)n);
  const uint64_t *ctp = a.crs + (size_t)tbeg * enc_words + slab;
  // 512 threads: the next plaintext row is prefetched across the transform; 1024 threads (half the
  // registers per lane, twice the waves to hide the L2 latency): loaded where it is used
  constexpr bool PREFETCH_C = THREADS == 512;
  double2 cn[PP];
  if (PREFETCH_C && tbeg < tend) {
#pragma unroll
    for (int k = 0; k < PP; k++)
      if (64 * k + lane < bpairs) cn[k] = crow[pbase + 64 * k];
  }
  for (unsigned long long t = tbeg; t < tend; t++) {
    const int pbl = pbase;
    if (!PREFETCH_C) {
#pragma unroll
      for (int k = 0; k < PP; k++)
        if (64 * k + lane < bpairs) cn[k] = crow[pbase + 64 * k];
    }
    // plaintext row -> tile, reduced mod Q_j
#pragma unroll
    for (int k = 0; k < PP; k++)
      if (64 * k + lane < bpairs) {
        const int pi = pidx(2 * (pbl + 64 * k));
        s[pi] = reduce(cn[k].x, mod);
        s[pnext(pi)] = reduce(cn[k].y, mod);
      }
    __syncthreads();
    // stream: this term's ciphertext words and the next plaintext row, in flight during the transform
    ulonglong2 ct[2][PP];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int k = 0; k < PP; k++)
        if (64 * k + lane < bpairs) {
          if (ABLATE & 2)
            ct[c][k] = make_ulonglong2(12345ull + k, 6789ull + c);
          else
            ct[c][k] = reinterpret_cast<const ulonglong2 *>(ctp + c * comp)[pbase + 64 * k];
        }
    crow += (size_t)L * (n >> 1);
    ctp += enc_words;
    if (PREFETCH_C && t + 1 < tend && !(ABLATE & 4)) {
#pragma unroll
      for (int k = 0; k < PP; k++)
        if (64 * k + lane < bpairs) cn[k] = crow[pbase + 64 * k];
    }
    // the loads above must be ISSUED before the transform (the scheduler would otherwise sink them
    // next to their uses, after the transform, to save registers)
    __builtin_amdgcn_sched_barrier(0);
    if (!(ABLATE & 1)) lds_ntt_fwd_wp<RS_MAC_MAXR, LdsIO, TileBlockFactory, 3, true>(s, lds, bf, logn, LOGW, twl, mod, red_mask);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < PP; k++)
      if (64 * k + lane < bpairs) {
        const int pi = pidx(2 * (pbl + 64 * k));
        double u0 = s[pi], u1 = s[pnext(pi)];
        if (a.reduce_u) {
          u0 = reduce(u0, mod);
          u1 = reduce(u1, mod);
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
          acc[c][2 * k] += mulmod(from_u64(ct[c][k].x), u0, mod);
          acc[c][2 * k + 1] += mulmod(from_u64(ct[c][k].y), u1, mod);
        }
      }
    if (++since >= a.acc_period) {
      since = 0;
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int k = 0; k < 2 * PP; k++) acc[c][k] = reduce(acc[c][k], mod);
    }
    wave_sync();  // this wave's block may now be overwritten by the next row (wave-private region)
  }
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int k = 0; k < PP; k++)
      if (64 * k + lane < bpairs) {
        ulonglong2 o;
        o.x = to_u64(canon(acc[c][2 * k], mod));
        o.y = to_u64(canon(acc[c][2 * k + 1], mod));
        reinterpret_cast<ulonglong2 *>(part + c * comp)[pbase + 64 * k] = o;
      }
}

// mac_kernel_v3 (N_enc = 8192, FP64): the wide form of ntt_wide.hpp applied to the inner product.
// A polynomial's accumulators (2 components x 8192) do not fit beside a 32-coefficient-per-thread transform, so a
// workgroup of 256 threads owns HALF of the spectrum of one (limb, prime): the first stage (gap 4096) is applied
// while the plaintext row is loaded -- half h keeps x[n] + w*x[n+4096] (h = 0) or x[n] - w*x[n+4096] (h = 1) -- and
// the remaining 12 stages are the 4096-point sub-transform rooted at node 2 + h, in three radix-16 rounds with 16
// coefficients per thread: two tile exchanges per term, the lane's own (round-3) twiddles in registers for the whole chunk.
// The spectrum half is exactly the contiguous half [4096 h, 4096 h + 4096) of both ciphertext components, which the
// workgroup streams with 16-byte loads issued before the transform; a wave-private pass turns round 3's 16
// consecutive points per thread into the lane-contiguous layout of those loads.  The plaintext row is read by both
// halves (from L2: it is shared by the 2 K workgroups of a (term, limb)) and stage 0's multiply is done twice:
// +8 % FP64 work against five fewer LDS passes per term and two independent workgroups per CU.
// Up to two groups that multiply the SAME key vector (A and B of groth16.tcc:89-103 against s_pows) run in one
// launch as neighbouring workgroups of one XCD, so the second read of a ciphertext word is served on-die.
// 1: the next plaintext row is requested before the multiply-accumulate (64 more live registers: spills at 256)
#ifndef RS_MAC3_ABLATE
#define RS_MAC3_ABLATE 0
#endif
struct MacArgs3 {
  const double *C[2];        // [tile_terms][L][n] plaintext rows per group
  uint64_t *partial[2];      // accumulator set per group: [n_chunks] stride part_stride
  unsigned long long terms[2];
  const uint64_t *crs;       // first ciphertext of the tile
  size_t part_stride;
  int n_groups, terms_per_chunk, n_chunks, accumulate, acc_period, reduce_u;
  int paired;                // rows in plain_center_wide_kernel's paired layout
  int ct_temporal;           // tuning knob "mac_ct_temporal": ciphertext words with ordinary (temporal) loads instead of non-temporal ones
  uint32_t red_mask[RS_MAX_K];  // bit s: reduce before stage s of the forward transform mod Q_j (start bound = max |C|)
};
// LOGN = 14 (N_enc = 16384: the shapes of BASELINE configs[3] / [4] and of the reference's microbench.cpp:13-14): a
// workgroup owns a QUARTER of the spectrum -- the first TWO stages (gaps 8192 and 4096) are applied while the row is
// loaded, the other 12 are the 4096-point sub-transform rooted at node 4 + quarter; everything after the load is the
// LOGN = 13 kernel.  Rows are in natural order (PAIRED = false).
template <bool PAIRED, int LOGN = 13>
__global__ void __launch_bounds__(256, 2)
mac_kernel_v3(MacArgs3 a, int L, int K, const NttTable *__restrict__ coeff_tabs) {
  constexpr int n = 1 << LOGN, H = 4096, LOGP = LOGN - 12, PARTS = 1 << LOGP;
  static_assert(LOGN == 13 || (LOGN == 14 && !PAIRED), "half spectrum at 8192 points, quarter spectrum at 16384");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  // block -> XCD slot x (blocks go to XCDs round-robin) and a position q in that XCD's sequence.  The 2K workgroups
  // that read one plaintext row (both halves of every prime) and the groups that read the same ciphertext words are
  // consecutive in ONE XCD's sequence: one of them fetches from memory, the others hit that XCD's L2.
  const unsigned b = blockIdx.x, x = b & 7u;
  unsigned q = b >> 3;
  const int g = (int)(q % (unsigned)a.n_groups);
  q /= (unsigned)a.n_groups;
  const unsigned hj = q % ((unsigned)PARTS * (unsigned)K);
  const unsigned rr = (q / ((unsigned)PARTS * (unsigned)K)) * 8u + x;  // (chunk, limb)
  const int h = (int)(hj & (unsigned)(PARTS - 1)), j = (int)(hj >> LOGP);
  const int limb = (int)(rr % (unsigned)L), chunk = (int)(rr / (unsigned)L);
  if (chunk >= a.n_chunks) return;
  const Mod mod = coeff_tabs[j].mod;
  const double *__restrict__ tw = coeff_tabs[j].d_tw;
  const uint32_t red_mask = a.red_mask[j];
  const int root = PARTS + h;
  // per-lane twiddles of rounds 2 and 3, fixed for the whole chunk
  const int lo = t & 15, hi = t >> 4;
  // round-2 twiddles tw[(root << (4+k)) + (hi << k) + b] (16 lanes share each) come from an LDS copy of the table's
  // first 1024 entries; the round-3 twiddles are the lane's own and stay in registers
  double *twl = s + 2 * (H + H / 16);
  for (int i = t; i < 1024; i += 256) twl[i] = tw[i];
  __syncthreads();
  double tw3[15];
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int bk = 0; bk < (1 << k); bk++) tw3[(1 << k) - 1 + bk] = tw[(root << (8 + k)) + (t << k) + bk];
#pragma unroll
  for (int i = 0; i < 15; i++) pin(tw3[i]);
  // wave-uniform twiddles of stage 0 and round 1, as scalar registers: fetched through the table pointer inside the
  // term loop they would be vector loads, and waiting for the youngest vector load drains the ciphertext stream
  // twiddles of the folded stages, with the sign of this workgroup's half: x + w y for the low half, x - w y for the high
  const double w0 = uniform_f64(tw[1]);
  const double w1 = uniform_f64(tw[2 + (h >> 1)]);  // LOGN = 14: stage 1 of this quarter's half
  const double w0s = (LOGN == 13 ? (h & 1) : (h & 2)) ? -w0 : w0, w1s = (h & 1) ? -w1 : w1;
  double tw1[15];
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int bk = 0; bk < (1 << k); bk++) tw1[(1 << k) - 1 + bk] = uniform_f64(tw[(root << k) + bk]);
  const size_t enc_words = (size_t)L * 2 * K * n;
  const size_t slab = (((size_t)limb * 2) * K + j) * (size_t)n + (size_t)h * H;  // component 0; component 1 is + K*n
  const size_t comp = (size_t)K * n;
  uint64_t *part = a.partial[g] + (size_t)chunk * a.part_stride + slab;
  const int r0 = wave * 1024;  // the wave's range of the half spectrum: 64 round-3 groups
  double acc[2][16];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int i = 0; i < 8; i++) {
      acc[c][2 * i] = acc[c][2 * i + 1] = 0.0;
      if (a.accumulate) {
        const u64x2 v = reinterpret_cast<const u64x2 *>(part + c * comp + r0)[lane + 64 * i];
        acc[c][2 * i] = from_u64(v.x);
        acc[c][2 * i + 1] = from_u64(v.y);
      }
    }
  const unsigned long long tbeg = (unsigned long long)chunk * a.terms_per_chunk;
  const unsigned long long tend = min(tbeg + (unsigned long long)a.terms_per_chunk, a.terms[g]);
  const double *crow = a.C[g] + ((size_t)tbeg * L + limb) * (size_t)n + (PAIRED ? 2 * t : t);
  const uint64_t *ctp = a.crs + (size_t)tbeg * enc_words + slab + r0;
  double cl[16], ch[16];
  // The row in two batches of 32 registers, the first requested before the previous term's multiply-accumulate and
  // the second after it (all 64 at once do not fit beside it).  Paired rows (plain_center_wide_kernel): one 16-byte
  // load brings x[n'] and x[n' + 4096]; plain rows: the multiplied operands x[n' + 4096] first.
  auto load_row_a = [&]() {
#pragma unroll
    for (int e = 0; e < 16; e++) {
#if RS_MAC3_ABLATE & 1  // experiment: no plaintext-row traffic (wrong results)
      if (PAIRED ? e < 8 : true) ch[e] = 5.0 + t;
      if (PAIRED && e < 8) cl[e] = 3.0 + e;
#else
      if (PAIRED) {
        if (e < 8) {
          const double2 x2 = reinterpret_cast<const double2 *>(crow)[256 * e];
          cl[e] = x2.x;
          ch[e] = x2.y;
        }
      } else {
        ch[e] = crow[256 * e + H];
      }
#endif
    }
  };
  auto load_row_b = [&]() {
#pragma unroll
    for (int e = 0; e < 16; e++) {
#if RS_MAC3_ABLATE & 1
      if (PAIRED ? e >= 8 : true) cl[e] = 3.0 + e;
      if (PAIRED && e >= 8) ch[e] = 5.0 + t;
#else
      if (PAIRED) {
        if (e >= 8) {
          const double2 x2 = reinterpret_cast<const double2 *>(crow)[256 * e];
          cl[e] = x2.x;
          ch[e] = x2.y;
        }
      } else {
        cl[e] = crow[256 * e];
      }
#endif
    }
  };
  // Software pipeline over the terms of the chunk, one term deep: the spectrum of term t is parked in tile t % 2 and
  // multiplied into the accumulators during iteration t + 1, AFTER that iteration's plaintext-row loads have been
  // issued and BEFORE its transform, so the ciphertext loads of term t (issued before the transform of term t) have a
  // whole term to land and the row loads of term t + 1 fly during the multiply-accumulate.
  u64x2 ct[2][8];
  const bool temporal = a.ct_temporal != 0;
  auto issue_ct = [&]() {
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
#if RS_MAC3_ABLATE & 2  // experiment: no ciphertext traffic (wrong results)
        ct[c][i] = u64x2{12345ull + i, 6789ull + c};
#else
        const u64x2 *cp = reinterpret_cast<const u64x2 *>(ctp + c * comp) + lane + 64 * i;
        ct[c][i] = temporal ? *cp : stream_load(cp);
#endif
      }
    ctp += enc_words;
  };
  int since = 0;
  auto mac = [&](const double *tile) {
    const int p0 = r0 + (r0 >> 4) + 2 * lane + (lane >> 3);  // px(r0 + 2 lane)
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const double u0 = tile[p0 + 136 * i], u1 = tile[p0 + 136 * i + 1];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        acc[c][2 * i] += mulmod(from_u64(ct[c][i].x), u0, mod);
        acc[c][2 * i + 1] += mulmod(from_u64(ct[c][i].y), u1, mod);
      }
    }
    if (++since >= a.acc_period) {
      since = 0;
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[c][i] = reduce(acc[c][i], mod);
    }
  };
  constexpr int TILE = H + H / 16;
  for (unsigned long long tt = tbeg; tt < tend; tt++) {
    double *tile = s + (int)((tt - tbeg) & 1) * TILE;
    double v[16];
    if (LOGN == 13) {
      load_row_a();
      mem_fence();
      if (tt > tbeg) mac(s + (int)((tt - tbeg + 1) & 1) * TILE);  // term tt - 1
      mem_fence();
      load_row_b();
      crow += (size_t)L * n;
      mem_fence();
      // stage 0 (gap 4096): this half's operand of the 4096-point sub-transform, x[n'] + (+-w0) x[n' + 4096] (the sign of
      // the half rides on the twiddle: mulmod(a, -w) = -mulmod(a, w) exactly).  The reductions the mask asks for are whole
      // guarded passes over the registers: inside the element loops the compiler turns them into compute-and-select.
      if (red_mask & 1u) {
#pragma unroll
        for (int e = 0; e < 16; e++) ch[e] = reduce(ch[e], mod);
      }
#pragma unroll
      for (int e = 0; e < 16; e++) {
        ch[e] = mulmod(ch[e], w0s, mod);
        pin(ch[e]);
      }
      if (red_mask & 1u) {
#pragma unroll
        for (int e = 0; e < 16; e++) cl[e] = reduce(cl[e], mod);
      }
#pragma unroll
      for (int e = 0; e < 16; e++) {
        v[e] = cl[e] + ch[e];
        pin(v[e]);
      }
    } else {
      // stages 0 (gap 8192) and 1 (gap 4096) on x[n'], x[n' + 4096], x[n' + 8192], x[n' + 12288], n' = t + 256 e: the two
      // operands multiplied by stage 0's twiddle are requested before the previous term's multiply-accumulate, the
      // other two after it, one at a time (all 64 words at once do not fit beside the accumulators)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        cl[e] = crow[256 * e + 2 * H];
        ch[e] = crow[256 * e + 3 * H];
      }
      mem_fence();
      if (tt > tbeg) mac(s + (int)((tt - tbeg + 1) & 1) * TILE);  // term tt - 1
      mem_fence();
#pragma unroll
      for (int e = 0; e < 16; e++) v[e] = crow[256 * e];
      mem_fence();
      if (red_mask & 1u) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
          cl[e] = reduce(cl[e], mod);
          ch[e] = reduce(ch[e], mod);
        }
      }
#pragma unroll
      for (int e = 0; e < 16; e++) {  // (+-w0) x2, (+-w0) x3: the sign of this quarter's half of stage 0 rides on the twiddle
        cl[e] = mulmod(cl[e], w0s, mod);
        ch[e] = mulmod(ch[e], w0s, mod);
        pin(cl[e]);
        pin(ch[e]);
      }
      if (red_mask & 1u) {
#pragma unroll
        for (int e = 0; e < 16; e++) v[e] = reduce(v[e], mod);
      }
#pragma unroll
      for (int e = 0; e < 16; e++) {  // u0 = x0 +- w0 x2
        cl[e] = v[e] + cl[e];
        pin(cl[e]);
      }
      mem_fence();
#pragma unroll
      for (int e = 0; e < 16; e++) v[e] = crow[256 * e + H];
      crow += (size_t)L * n;
      mem_fence();
      if (red_mask & 1u) {
#pragma unroll
        for (int e = 0; e < 16; e++) v[e] = reduce(v[e], mod);
      }
#pragma unroll
      for (int e = 0; e < 16; e++) {  // u1 = x1 +- w0 x3
        ch[e] = v[e] + ch[e];
        pin(ch[e]);
      }
      if (red_mask & 2u) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
          ch[e] = reduce(ch[e], mod);
          cl[e] = reduce(cl[e], mod);
        }
      }
#pragma unroll
      for (int e = 0; e < 16; e++) {  // stage 1: v = u0 + (+-w1) u1
        v[e] = cl[e] + mulmod(ch[e], w1s, mod);
        pin(v[e]);
      }
    }
    mem_fence();
    issue_ct();  // after the row registers are dead: the two never overlap
    mem_fence();
#if RS_MAC3_ABLATE & 4  // experiment: no transform (wrong results)
    tile[17 * t] = v[0] + v[5] + v[9] + v[15];
    wave_sync();
    continue;
#endif
    // round 1: stages LOGP..LOGP+3 on elements t + 256 e (uniform twiddles)
    reg_fwd_stages<4, true>(v, mod, red_mask >> LOGP, [&](int k, int bk) { return tw1[(1 << k) - 1 + bk]; });
    {  // tile tt % 2 was last read by the multiply-accumulate of term tt - 2, two barriers ago
      const int pb = t + (t >> 4);
#pragma unroll
      for (int e = 0; e < 16; e++) tile[pb + 272 * e] = v[e];
    }
    __syncthreads();
    {  // round 2: stages 5..8 on hi*256 + lo + 16 e
      const int pb = hi * 272 + lo;
#pragma unroll
      for (int e = 0; e < 16; e++) v[e] = tile[pb + 17 * e];
      reg_fwd_stages<4, true>(v, mod, red_mask >> (LOGP + 4), [&](int k, int bk) { return twl[(root << (4 + k)) + (hi << k) + bk]; });
#pragma unroll
      for (int e = 0; e < 16; e++) tile[pb + 17 * e] = v[e];
    }
    wave_sync();  // a round-2 group (256 elements) is 16 consecutive threads, who also own it in round 3: no workgroup barrier
    {  // round 3: stages 9..12 on 16 consecutive points, parked for the wave-private transposition
      const int pb = 17 * t;
#pragma unroll
      for (int e = 0; e < 16; e++) v[e] = tile[pb + e];
      reg_fwd_stages<4, true>(v, mod, red_mask >> (LOGP + 8), [&](int k, int bk) { return tw3[(1 << k) - 1 + bk]; });
      if (a.reduce_u) {  // a guarded pass: as a per-element choice it becomes compute-and-select
#pragma unroll
        for (int e = 0; e < 16; e++) v[e] = reduce(v[e], mod);
      }
#pragma unroll
      for (int e = 0; e < 16; e++) tile[pb + e] = v[e];
    }
    wave_sync();
  }
  if (tend > tbeg) mac(s + (int)((tend - tbeg + 1) & 1) * TILE);  // the last term
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int i = 0; i < 8; i++) {
      u64x2 o;
      o.x = to_u64(canon(acc[c][2 * i], mod));
      o.y = to_u64(canon(acc[c][2 * i + 1], mod));
      reinterpret_cast<u64x2 *>(part + c * comp + r0)[lane + 64 * i] = o;
    }
}

// mac_kernel_v4 (N_enc = 16384 -- described here -- and 8192; FP64, TWO key vectors): Rinocchio multiplies every coefficient vector into BOTH s_pows and
// alpha_s_pows (rinocchio.tcc:106-160: a, b, c, h, z against each), so the plaintext spectrum -- 82 % of mac_kernel_v3's
// arithmetic -- is wanted twice.  Four accumulator sets (2 keys x 2 components) of a spectrum quarter do not fit the
// registers of 256 threads next to a 16-coefficient-per-thread transform; here 512 threads (one workgroup per CU, the same
// 8 waves) own the quarter with EIGHT points each: the accumulators of both keys take the registers one key took, the
// ciphertext streams of both keys the registers one stream took, and the transform runs once per term.
//   stages 0, 1       while the row is loaded (as mac_kernel_v3<false, 14>), on elements t + 512 e
//   round 1           sub-stages 0..2 on t + 512 e           wave-uniform twiddles        -> tile, WORKGROUP barrier
//   round 2           sub-stages 3..5 on 512 w + lane + 64 e  wave w's own 512 elements:
//   round 3           sub-stages 6..8 on 64 (t/8) + t%8 + 8 e   every later exchange is wave-private
//   round 4           sub-stages 9..11 on 8 t + e             the lane's own twiddles, in registers for the whole chunk
// One workgroup barrier per term; the spectrum of term t is multiplied into the accumulators during iteration t + 1 (tiles
// alternate), under the row loads of that term.  Tile position of element i: i + i/8.
// Up to RS_MAC4_GROUPS coefficient vectors (groups) run in one launch as neighbouring workgroups of one XCD, so that one of
// them fetches a ciphertext word from memory and the others find it in that XCD's L2.
constexpr int RS_MAC4_GROUPS = 6;
struct MacArgs4 {
  const double *C[RS_MAC4_GROUPS];        // [tile_terms][L][n] plaintext rows per group
  uint64_t *partial[2][RS_MAC4_GROUPS];   // accumulator set per (key, group): [n_chunks] stride part_stride
  unsigned long long terms[RS_MAC4_GROUPS];
  const uint64_t *crs[2];                 // first ciphertext of the tile, per key vector
  size_t part_stride;
  int n_groups, terms_per_chunk, n_chunks, accumulate, acc_period, reduce_u;
  uint32_t red_mask[RS_MAX_K];
};
// LOGN = 13 (N_enc = 8192): the workgroup owns HALF of the spectrum and folds one stage while the row is loaded (PAIRED:
// rows in plain_center_wide_kernel's paired layout, one 16-byte load per operand pair).
// NKEY = 1 (round 6; N_enc = 8192): ONE key vector in the same shape -- half the accumulators and half the ciphertext registers,
// so the kernel fits 128 VGPRs and TWO 512-thread workgroups share a CU: four waves per SIMD instead of mac_kernel_v3's two
// (256 threads x 16 points, 240 VGPRs), the latency of the tile exchanges and of the one workgroup barrier per term hidden by
// twice the waves.  Same arithmetic per coefficient as mac_kernel_v3 (stage 0 folded into the row load, 12 stages, the
// multiply-accumulate one term behind): bit-identical results (mac_variant 6 / 5).
template <int LOGN, bool PAIRED, int NKEY = 2>
__global__ void __launch_bounds__(512, (NKEY == 1 ? 4 : 2))
mac_kernel_v4(MacArgs4 a, int L, int K, const NttTable *__restrict__ coeff_tabs) {
  constexpr int n = 1 << LOGN, H = 4096, LOGP = LOGN - 12, PARTS = 1 << LOGP, TILE = H + H / 8;
  static_assert(LOGN == 14 ? !PAIRED : LOGN == 13, "quarter spectrum at 16384 points, half spectrum at 8192");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *s = reinterpret_cast<double *>(smem);
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const unsigned b = blockIdx.x, x = b & 7u;  // XCD slot and position in its sequence, as mac_kernel_v3
  unsigned q = b >> 3;
  const int g = (int)(q % (unsigned)a.n_groups);
  q /= (unsigned)a.n_groups;
  const unsigned hj = q % ((unsigned)PARTS * (unsigned)K);
  const unsigned rr = (q / ((unsigned)PARTS * (unsigned)K)) * 8u + x;  // (chunk, limb)
  const int h = (int)(hj & (unsigned)(PARTS - 1)), j = (int)(hj >> LOGP);
  const int limb = (int)(rr % (unsigned)L), chunk = (int)(rr / (unsigned)L);
  if (chunk >= a.n_chunks) return;
  const Mod mod = coeff_tabs[j].mod;
  const double *__restrict__ tw = coeff_tabs[j].d_tw;
  const uint32_t red_mask = a.red_mask[j];
  const int root = PARTS + h;
  const double w0 = uniform_f64(tw[1]);
  const double w1 = uniform_f64(tw[2 + (h >> 1)]);  // LOGN = 14: stage 1 of this quarter's half
  const double w0s = (LOGN == 13 ? (h & 1) : (h & 2)) ? -w0 : w0, w1s = (h & 1) ? -w1 : w1;  // with the sign of this workgroup's half
  double tw1[7], tw2[7], tw3[7], tw4[7];
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int bk = 0; bk < (1 << k); bk++) {
      tw1[(1 << k) - 1 + bk] = uniform_f64(tw[(root << k) + bk]);
      tw2[(1 << k) - 1 + bk] = uniform_f64(tw[(root << (3 + k)) + (wave << k) + bk]);
      tw3[(1 << k) - 1 + bk] = tw[(root << (6 + k)) + ((t >> 3) << k) + bk];
      tw4[(1 << k) - 1 + bk] = tw[(root << (9 + k)) + (t << k) + bk];
    }
#pragma unroll
  for (int i = 0; i < 7; i++) {
    if (NKEY > 1) pin(tw3[i]);
    pin(tw4[i]);
  }
  const size_t enc_words = (size_t)L * 2 * K * n;
  const size_t slab = (((size_t)limb * 2) * K + j) * (size_t)n + (size_t)h * H;  // component 0; component 1 is + K*n
  const size_t comp = (size_t)K * n;
  const int r0 = wave * 512;  // the wave's range of the quarter: its 64 round-4 groups
  uint64_t *part[NKEY];
  double acc[NKEY][2][8];
#pragma unroll
  for (int kx = 0; kx < NKEY; kx++) {
    part[kx] = a.partial[kx][g] + (size_t)chunk * a.part_stride + slab + r0;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        acc[kx][c][2 * i] = acc[kx][c][2 * i + 1] = 0.0;
        if (a.accumulate) {
          const u64x2 v = reinterpret_cast<const u64x2 *>(part[kx] + c * comp)[lane + 64 * i];
          acc[kx][c][2 * i] = from_u64(v.x);
          acc[kx][c][2 * i + 1] = from_u64(v.y);
        }
      }
  }
  const unsigned long long tbeg = (unsigned long long)chunk * a.terms_per_chunk;
  const unsigned long long tend = min(tbeg + (unsigned long long)a.terms_per_chunk, a.terms[g]);
  const double *crow = a.C[g] + ((size_t)tbeg * L + limb) * (size_t)n + (PAIRED ? 2 * t : t);
  const uint64_t *ctp0 = a.crs[0] + (size_t)tbeg * enc_words + slab + r0;
  const uint64_t *ctp1 = NKEY > 1 ? a.crs[1] + (size_t)tbeg * enc_words + slab + r0 : nullptr;
  u64x2 ct[NKEY][2][4];
  auto issue_ct = [&]() {
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        ct[0][c][i] = stream_load(reinterpret_cast<const u64x2 *>(ctp0 + c * comp) + lane + 64 * i);
        if (NKEY > 1) ct[NKEY - 1][c][i] = stream_load(reinterpret_cast<const u64x2 *>(ctp1 + c * comp) + lane + 64 * i);
      }
    ctp0 += enc_words;
    if (NKEY > 1) ctp1 += enc_words;
  };
  int since = 0;
  auto mac = [&](const double *tile) {
    const int p0 = r0 + (r0 >> 3) + 2 * lane + (lane >> 2);  // position of element r0 + 2 lane
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const double u0 = tile[p0 + 144 * i], u1 = tile[p0 + 144 * i + 1];
#pragma unroll
      for (int kx = 0; kx < NKEY; kx++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
          acc[kx][c][2 * i] += mulmod(from_u64(ct[kx][c][i].x), u0, mod);
          acc[kx][c][2 * i + 1] += mulmod(from_u64(ct[kx][c][i].y), u1, mod);
        }
    }
    if (++since >= a.acc_period) {
      since = 0;
#pragma unroll
      for (int kx = 0; kx < NKEY; kx++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
          for (int i = 0; i < 8; i++) acc[kx][c][i] = reduce(acc[kx][c][i], mod);
    }
  };
  for (unsigned long long tt = tbeg; tt < tend; tt++) {
    double *tile = s + (int)((tt - tbeg) & 1) * TILE;
    double v[8], c2[8], c3[8];
    if (LOGN == 13) {
      // NKEY = 1 (128 registers, four waves per SIMD): the multiply-accumulate of term tt - 1 FIRST, then the row loads -- the
      // row registers would not fit beside the ciphertext words it consumes; the other three waves of the SIMD cover the latency
      if (NKEY == 1 && tt > tbeg) {
        mac(s + (int)((tt - tbeg + 1) & 1) * TILE);
        mem_fence();
      }
      // stage 0 (gap 4096) on x[n'], x[n' + 4096], n' = t + 512 e: this half's operand of the 4096-point sub-transform
#pragma unroll
      for (int e = 0; e < 8; e++) {
        if (PAIRED) {
          const double2 x2 = reinterpret_cast<const double2 *>(crow)[512 * e];
          v[e] = x2.x;
          c2[e] = x2.y;
        } else {
          v[e] = crow[512 * e];
          c2[e] = crow[512 * e + H];
        }
      }
      crow += (size_t)L * n;
      mem_fence();
      if (NKEY > 1 && tt > tbeg) mac(s + (int)((tt - tbeg + 1) & 1) * TILE);  // term tt - 1, under the row loads
      mem_fence();
      if (red_mask & 1u) {  // a guarded pass: inside the element loop the reduction becomes compute-and-select
#pragma unroll
        for (int e = 0; e < 8; e++) {
          v[e] = reduce(v[e], mod);
          c2[e] = reduce(c2[e], mod);
        }
      }
#pragma unroll
      for (int e = 0; e < 8; e++) {  // the sign of the half rides on the twiddle: mulmod(a, -w) = -mulmod(a, w) exactly
        v[e] = v[e] + mulmod(c2[e], w0s, mod);
        pin(v[e]);
      }
    } else {
      // stages 0 (gap 8192) and 1 (gap 4096) on x[n'], x[n' + 4096], x[n' + 8192], x[n' + 12288], n' = t + 512 e
#pragma unroll
      for (int e = 0; e < 8; e++) {
        c2[e] = crow[512 * e + 2 * H];
        c3[e] = crow[512 * e + 3 * H];
      }
      mem_fence();
      if (tt > tbeg) mac(s + (int)((tt - tbeg + 1) & 1) * TILE);  // term tt - 1, under the row loads
      mem_fence();
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = crow[512 * e];
      mem_fence();
      if (red_mask & 1u) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
          c2[e] = reduce(c2[e], mod);
          c3[e] = reduce(c3[e], mod);
        }
      }
#pragma unroll
      for (int e = 0; e < 8; e++) {  // (+-w0) x2, (+-w0) x3: the sign of this quarter's half of stage 0 rides on the twiddle
        c2[e] = mulmod(c2[e], w0s, mod);
        c3[e] = mulmod(c3[e], w0s, mod);
        pin(c2[e]);
        pin(c3[e]);
      }
      if (red_mask & 1u) {
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = reduce(v[e], mod);
      }
#pragma unroll
      for (int e = 0; e < 8; e++) {  // u0 = x0 +- w0 x2
        c2[e] = v[e] + c2[e];
        pin(c2[e]);
      }
      mem_fence();
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = crow[512 * e + H];
      crow += (size_t)L * n;
      mem_fence();
      if (red_mask & 1u) {
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = reduce(v[e], mod);
      }
#pragma unroll
      for (int e = 0; e < 8; e++) {  // u1 = x1 +- w0 x3
        c3[e] = v[e] + c3[e];
        pin(c3[e]);
      }
      if (red_mask & 2u) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
          c3[e] = reduce(c3[e], mod);
          c2[e] = reduce(c2[e], mod);
        }
      }
#pragma unroll
      for (int e = 0; e < 8; e++) {  // stage 1: v = u0 + (+-w1) u1
        v[e] = c2[e] + mulmod(c3[e], w1s, mod);
        pin(v[e]);
      }
    }
    mem_fence();
    issue_ct();  // both keys' words of this term: a whole transform to land
    mem_fence();
    reg_fwd_stages<3, true>(v, mod, red_mask >> LOGP, [&](int k, int bk) { return tw1[(1 << k) - 1 + bk]; });
    {  // tile tt % 2 was last read by the multiply-accumulate of term tt - 2, before the previous barrier
      const int pb = t + (t >> 3);
#pragma unroll
      for (int e = 0; e < 8; e++) tile[pb + 576 * e] = v[e];
    }
    __syncthreads();
    {  // round 2: the wave's own 512 elements
      const int pb = wave * 576 + lane + (lane >> 3);
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = tile[pb + 72 * e];
      reg_fwd_stages<3, true>(v, mod, red_mask >> (LOGP + 3), [&](int k, int bk) { return tw2[(1 << k) - 1 + bk]; });
#pragma unroll
      for (int e = 0; e < 8; e++) tile[pb + 72 * e] = v[e];
    }
    wave_sync();
    {  // round 3: 64-element groups of 8 consecutive threads
      const int pb = (t >> 3) * 72 + (t & 7);
      if (NKEY == 1) {  // 128 registers: this round's seven twiddles are re-read per term (eight lanes share each; L1 / L2 resident)
        int g3 = t >> 3;
        asm volatile("" : "+v"(g3));  // opaque: the loads must not be hoisted out of the term loop
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
          for (int bk = 0; bk < (1 << k); bk++) tw3[(1 << k) - 1 + bk] = tw[(root << (6 + k)) + (g3 << k) + bk];
      }
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = tile[pb + 9 * e];
      reg_fwd_stages<3, true>(v, mod, red_mask >> (LOGP + 6), [&](int k, int bk) { return tw3[(1 << k) - 1 + bk]; });
#pragma unroll
      for (int e = 0; e < 8; e++) tile[pb + 9 * e] = v[e];
    }
    wave_sync();
    {  // round 4: 8 consecutive points
      const int pb = 9 * t;
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = tile[pb + e];
      reg_fwd_stages<3, true>(v, mod, red_mask >> (LOGP + 9), [&](int k, int bk) { return tw4[(1 << k) - 1 + bk]; });
      if (a.reduce_u) {
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = reduce(v[e], mod);
      }
#pragma unroll
      for (int e = 0; e < 8; e++) tile[pb + e] = v[e];
    }
    wave_sync();
  }
  if (tend > tbeg) mac(s + (int)((tend - tbeg + 1) & 1) * TILE);  // the last term
#pragma unroll
  for (int kx = 0; kx < NKEY; kx++)
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        u64x2 o;
        o.x = to_u64(canon(acc[kx][c][2 * i], mod));
        o.y = to_u64(canon(acc[kx][c][2 * i + 1], mod));
        reinterpret_cast<u64x2 *>(part[kx] + c * comp)[lane + 64 * i] = o;
      }
}

// out[set] = sum_chunk partial[chunk][set] (+ addend[set]) mod Q_j
struct ReduceArgs {
  const uint64_t *addend[12];
};
__global__ void __launch_bounds__(256)
reduce_kernel(const uint64_t *__restrict__ partial, uint64_t *__restrict__ out, ReduceArgs add, int n_chunks,
              int n_sets, size_t enc_words, int n, int K, const uint64_t *__restrict__ Qint) {
  const size_t total = (size_t)n_sets * enc_words;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t set = i / enc_words, w = i % enc_words;
    const uint64_t Q = Qint[(w / (size_t)n) % (size_t)K];
    uint64_t sum = 0;  // modular running sum: residues may be 61 bits wide, a plain sum of many of them would wrap
    for (int c = 0; c < n_chunks; c++) {
      sum += partial[((size_t)c * n_sets + set) * enc_words + w];
      sum = sum >= Q ? sum - Q : sum;
    }
    if (add.addend[set]) {
      sum += add.addend[set][w];
      sum = sum >= Q ? sum - Q : sum;
    }
    out[i] = sum;
  }
}

// a8: EncodingElem::operator+= (dyadic add mod Q_j, seal_ring.tcc:479-506): an HBM-bound stream of 24 bytes per residue, in the
// shape of ring_pointwise_kernel (rs_core.hip): contiguous 32 KiB pieces per workgroup, eight 16-byte loads per operand in
// flight per lane.  `pairs` 16-byte words; the prime of word i is (i >> logn) % K, uniform over a piece for logn >= 12.
template <bool NT>
__global__ void __launch_bounds__(256)
enc_add_kernel(uint64_t *__restrict__ dst, const uint64_t *__restrict__ x, const uint64_t *__restrict__ y,
               size_t pairs, int logn, int K, const uint64_t *__restrict__ Qint) {
  const u64x2 *x2 = reinterpret_cast<const u64x2 *>(x), *y2 = reinterpret_cast<const u64x2 *>(y);
  u64x2 *d2 = reinterpret_cast<u64x2 *>(dst);
  const size_t full = pairs & ~(size_t)2047;
  const bool uniform = logn >= 12;
  for (size_t base = (size_t)blockIdx.x * 2048; base < full; base += (size_t)gridDim.x * 2048) {
    u64x2 va[8], vb[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const size_t i = base + threadIdx.x + 256 * k;
      va[k] = NT ? __builtin_nontemporal_load(x2 + i) : x2[i];
      vb[k] = NT ? __builtin_nontemporal_load(y2 + i) : y2[i];
    }
    const uint64_t Qu = Qint[((2 * base) >> logn) % (size_t)K];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const size_t i = base + threadIdx.x + 256 * k;
      const uint64_t Q = uniform ? Qu : Qint[(unsigned)((2 * i) >> logn) % (unsigned)K];
      u64x2 o;
      o.x = va[k].x + vb[k].x;
      o.y = va[k].y + vb[k].y;
      o.x = o.x >= Q ? o.x - Q : o.x;
      o.y = o.y >= Q ? o.y - Q : o.y;
      if (NT) __builtin_nontemporal_store(o, d2 + i);
      else d2[i] = o;
    }
  }
  for (size_t i = full + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) {  // the tail
    const uint64_t Q = Qint[(unsigned)((2 * i) >> logn) % (unsigned)K];
    const u64x2 p = x2[i], q = y2[i];
    u64x2 o;
    o.x = p.x + q.x;
    o.y = p.y + q.y;
    o.x = o.x >= Q ? o.x - Q : o.x;
    o.y = o.y >= Q ? o.y - Q : o.y;
    d2[i] = o;
  }
}

// canonicalise integer sums of residues (after an all-reduce of partial encoding sums)
__global__ void __launch_bounds__(256)
enc_reduce_kernel(uint64_t *__restrict__ x, size_t words, int n, int K, const uint64_t *__restrict__ Qint) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride)
    x[i] = x[i] % Qint[(i / (size_t)n) % (size_t)K];
}

}  // namespace rs
