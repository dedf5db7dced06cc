// witness_cols.hpp -- column plans, the column <-> boundary layout map and the transposes of the witness map (witness.hip)
#pragma once
#include "ntt_core.hpp"
#include "ntt_wide.hpp"
#include "rs_internal.hpp"

namespace rs {

constexpr int SCHOOL_LEVELS = 4;  // tree levels with node size <= 8 use schoolbook products

// per-limb device pointers handed to the column kernels; M: the context's arithmetic (tables hold table
// constants of that arithmetic: balanced doubles, or Montgomery-form integers)
template <class M_>
struct ColPlanT {
  using M = M_;
  using T = typename ArithOf<M_>::T;
  M mod;
  const T *tw, *itw, *invfact, *ehat, *dhat, *dlow, *shat, *ztab;
  const T *bc_e, *bc_s, *bc_d;  // block-convolution path (LimbPlan)
  const T *b2_e, *b2_s, *b2_d;  // ... in its two-dimensional form
  const T *cos_g, *cos_h, *cos_z;  // coset form of H (big_h_coset): g^k, g^-k / M, 1 / Z(g w^i) in transform order
  T bc_inv2b;                   // 1 / (2B) as a table constant
  T b2_inv;                     // 1 / (2B * 2M/B) = 1 / (4M): both unscaled inverse transforms of a two-dimensional data x data product
  uint32_t fwd_mask2, inv_mask2;
  uint32_t fmask[24], imask[24];  // reduce masks for transforms of length 2^l (FP64 arithmetic)
  uint32_t pwmask;                // bit l: the spectrum of a length-2^l forward transform must be reduced before it meets a table entry
  // incomplete transforms (witness_inc.hpp): the tables tw / itw hold 2^adic entries and a transform of length 2^logn > 2^adic
  // runs its first adic stages only (inc(logn) = logn - adic stages short; spectra tables of that length are in the same form)
  int adic;
  __host__ __device__ __forceinline__ int inc(int logn) const { return logn > adic ? logn - adic : 0; }
};
template <class M_>
struct ColPlansT {
  using M = M_;
  using T = typename ArithOf<M_>::T;
  ColPlanT<M_> l[RS_MAX_L];
};
using ColPlan = ColPlanT<Mod>;
using ColPlans = ColPlansT<Mod>;
struct ColBlockFactory {
  double *s;
  __device__ __forceinline__ LdsBlockIO operator()(int off) const { return LdsBlockIO{s + pidx(off)}; }
};

// Which columns a launch works on, and where they live in the boundary layouts.  The witness map is
// column-parallel (one column = one NTT slot of one ring limb), so a call may process any sub-range
// of slots of any sub-range of limbs: column c of the chunk is slot `slot0 + c % ns` of limb
// `limb0 + c / ns`.  Inputs (assignment, d1..d3) are always in the full layout [..][L][N]; outputs are
// [t][L][out_N] with slot s stored at s - out_slot0 (out_N = N, out_slot0 = 0: the full layout;
// out_N = ns, out_slot0 = slot0: the compact layout of a slot-sharded rank).
// slot0, ns, out_slot0 are even (lanes move slot PAIRS with 16-byte accesses).
// row0, row1: only output rows [row0, row1) are written, row r at index r - row0 of the output (a rank of a limb group
// keeps the rows of its TERM range only, ringsnark_amd/dist.py; witness.hip sets them per output vector).
struct ColMap {
  int limb0, ns, slot0, N, L, out_N, out_slot0;
  size_t row0 = 0, row1 = ~(size_t)0;
  __device__ __forceinline__ void locate(size_t c, int &limb, int &slot) const {
    limb = limb0 + (int)(c / (size_t)ns);
    slot = slot0 + (int)(c % (size_t)ns);
  }
  __device__ __forceinline__ size_t in_index(int limb, int slot) const { return (size_t)limb * N + slot; }
  __device__ __forceinline__ size_t out_index(int limb, int slot) const { return (size_t)limb * out_N + (slot - out_slot0); }
  __host__ __device__ __forceinline__ size_t in_stride() const { return (size_t)L * N; }
  __host__ __device__ __forceinline__ size_t out_stride() const { return (size_t)L * out_N; }
};

// The kernels that move data between the term-major boundary layout and the column-major working layout do it in tiles of
// 64 columns x 64 rows through LDS: 512-byte segments on both sides, 16 bytes per lane, eight loads per lane in flight before
// the first is consumed (round 6; the 32 x 32 tiles before moved 8 bytes per lane in 256-byte segments).  grid (C/64, M/64),
// columns fastest: numbering the tiles in super-blocks instead (so that the resident tiles cover longer contiguous runs of
// every column) measured 1-3 % slower -- profiles/r06_tiles_ab.txt.
template <class T>
struct alignas(16) Pair2 {
  T x, y;
};

// [rows][S] u64 (term-major, S = L*N) -> [S][M] f64 (column-major), rows >= m zero-filled.
template <class T>
__global__ void __launch_bounds__(256) transpose_in_kernel(const uint64_t *__restrict__ src, T *__restrict__ dst,
                                                           size_t m, size_t S, size_t M) {
  __shared__ T tile[32][33];
  const size_t s0 = (size_t)blockIdx.x * 32, r0 = (size_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const size_t r = r0 + k, sl = s0 + tx;
    tile[k][tx] = (r < m && sl < S) ? from_res<T>(src[r * S + sl]) : T(0);
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const size_t sl = s0 + k, r = r0 + tx;
    if (sl < S && r < M) dst[sl * M + r] = tile[tx][k];
  }
}
// [C][M] f64 canonical columns -> [rows][L][out_N] u64 for rows < m_out (M, C even: the column tile holds a power of two
// of rows, and chunks hold slot pairs)
template <class T>
__global__ void __launch_bounds__(256) transpose_out_kernel(const T *__restrict__ src, uint64_t *__restrict__ dst,
                                                            size_t m_out, size_t C, size_t M, ColMap cm) {
  __shared__ T tile[64][65];  // [column][row]
  const size_t s0 = (size_t)blockIdx.x * 64, r0 = (size_t)blockIdx.y * 64;
  if (r0 + 64 <= cm.row0 || r0 >= cm.row1 || r0 >= m_out) return;  // no row of this tile is wanted
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  Pair2<T> v[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const size_t c = s0 + ty + 8 * j, r = r0 + 2 * tx;
    v[j] = (c < C && r < M) ? *reinterpret_cast<const Pair2<T> *>(src + c * M + r) : Pair2<T>{T(0), T(0)};
  }
#pragma unroll
  for (int j = 0; j < 8; j++) {
    tile[ty + 8 * j][2 * tx] = v[j].x;
    tile[ty + 8 * j][2 * tx + 1] = v[j].y;
  }
  __syncthreads();
  const size_t c = s0 + 2 * tx;  // this lane's column pair (ns is even: both slots in one limb)
  if (c >= C) return;
  int limb, slot;
  cm.locate(c, limb, slot);
  const size_t o = cm.out_index(limb, slot) >> 1, So = cm.out_stride();
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int k = ty + 8 * j;
    const size_t r = r0 + k;
    if (r < m_out && r >= cm.row0 && r < cm.row1) {
      ulonglong2 w;
      w.x = to_res(tile[2 * tx][k]);
      w.y = to_res(tile[2 * tx + 1][k]);
      reinterpret_cast<ulonglong2 *>(dst + (r - cm.row0) * So)[o] = w;
    }
  }
}

}  // namespace rs
