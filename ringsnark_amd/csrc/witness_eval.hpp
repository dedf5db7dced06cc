// witness_eval.hpp -- constraint evaluation into columns, io vectors without interpolation, io / mid output (witness.hip)
#pragma once
#include "witness_cols.hpp"

namespace rs {

// Input/primary coefficient vectors without interpolation (io shortcut): interpolation is linear
// and the io evaluations depend on the n_inputs primary variables only, so
//     X_io[t] = Lconst[t] + sum_{k <= n_inputs} x_k (*) L_k[t],   L_k = interp(column k of X)
// with slot-constant L_k computed once per circuit.  grid (m, slot pairs / 256).
struct IoDesc {
  const int *k;       // variable index (0 = constant one)
  const int *column;  // column index into Lcols
  int count;
};
template <class M>
__global__ void __launch_bounds__(256)
io_coeff_kernel(IoDesc io, const typename ArithOf<M>::T *__restrict__ Lcols /* [ncols][Ltot][M] */, const uint64_t *__restrict__ asg,
                uint64_t *__restrict__ out, size_t C, size_t Mlen, const M *__restrict__ qmod, ColMap cm) {
  using T = typename ArithOf<M>::T;
  const size_t t = cm.row0 + blockIdx.x;  // grid.x = the rows wanted
  const size_t c = 2 * ((size_t)blockIdx.y * blockDim.x + threadIdx.x);
  if (c >= C) return;
  int limb, slot;
  cm.locate(c, limb, slot);
  const size_t pair = cm.in_index(limb, slot) >> 1, Si = cm.in_stride();
  const M mod = qmod[limb];
  T a0 = T(0), a1 = T(0);
  for (int k = 0; k < io.count; k++) {
    const T lv = center(Lcols[((size_t)io.column[k] * cm.L + limb) * Mlen + t], mod);
    const int v_ = io.k[k];
    if (v_ == 0) {
      a0 = addm(a0, lv, mod);
      a1 = addm(a1, lv, mod);
    } else {
      const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(asg + (size_t)(v_ - 1) * Si)[pair];
      a0 = addm(a0, mulmod_dd(from_res<T>(v.x), lv, mod), mod);
      a1 = addm(a1, mulmod_dd(from_res<T>(v.y), lv, mod), mod);
    }
    if ((k & 3) == 3) {
      a0 = reduce(a0, mod);
      a1 = reduce(a1, mod);
    }
  }
  ulonglong2 o;
  o.x = to_res(canon(a0, mod));
  o.y = to_res(canon(a1, mod));
  reinterpret_cast<ulonglong2 *>(out + (t - cm.row0) * cm.out_stride())[cm.out_index(limb, slot) >> 1] = o;
}

// Column-major interpolated `full` vector -> term-major io AND mid vectors in one pass:
//   io[t]  = Lconst[t] + sum_k x_k (*) L_k[t]          (io shortcut, as io_coeff_kernel)
//   mid[t] = full[t] - io[t] + const[limb][t]
// i.e. transpose + io + mid fused: the column tile is transposed through LDS, the io value is
// computed where it is needed, and both results are written once (16 bytes per lane).
// grid (C/64, M/64): tiles of 64 columns x 64 rows (witness_cols.hpp); tiles past row m are skipped unread.
template <class M>
__global__ void __launch_bounds__(256)
io_mid_out_kernel(const typename ArithOf<M>::T *__restrict__ cols, IoDesc io,
                  const typename ArithOf<M>::T *__restrict__ Lcols /* [ncols][Ltot][M] */, const uint64_t *__restrict__ asg,
                  const typename ArithOf<M>::T *__restrict__ cst /* [Ltot][M] or null */, uint64_t *__restrict__ io_out /* or null */,
                  uint64_t *__restrict__ mid_out, size_t m, size_t C, size_t Mlen, const M *__restrict__ qmod, ColMap cm) {
  using T = typename ArithOf<M>::T;
  __shared__ T tile[64][65];  // [column][row]
  const size_t s0 = (size_t)blockIdx.x * 64, r0 = (size_t)blockIdx.y * 64;
  if (r0 + 64 <= cm.row0 || r0 >= cm.row1 || r0 >= m) return;  // no row of this tile is wanted (uniform: before the barrier)
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  {
    Pair2<T> v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const size_t c = s0 + ty + 8 * j, r = r0 + 2 * tx;
      v[j] = (c < C && r < Mlen) ? *reinterpret_cast<const Pair2<T> *>(cols + c * Mlen + r) : Pair2<T>{T(0), T(0)};
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      tile[ty + 8 * j][2 * tx] = v[j].x;
      tile[ty + 8 * j][2 * tx + 1] = v[j].y;
    }
  }
  __syncthreads();
  const size_t c = s0 + 2 * tx;  // this lane's column pair (ns is even: both slots in one limb)
  if (c >= C) return;
  int limb, slot;
  cm.locate(c, limb, slot);
  const size_t pair = cm.in_index(limb, slot) >> 1, Si = cm.in_stride();
  const size_t opair = cm.out_index(limb, slot) >> 1, So = cm.out_stride();
  const M mod = qmod[limb];
  // the primary inputs of this lane's two slots do not depend on the row: the first XC terms keep them in registers (round 5:
  // they were re-read from L2 for every row -- two 16-byte loads per lane and row beside 8 bytes of column data)
  constexpr int XC = 4;
  T xa[XC], xb[XC];
#pragma unroll
  for (int e = 0; e < XC; e++) {
    xa[e] = xb[e] = T(0);
    if (e < io.count && io.k[e] != 0) {
      const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(asg + (size_t)(io.k[e] - 1) * Si)[pair];
      xa[e] = from_res<T>(v.x);
      xb[e] = from_res<T>(v.y);
    }
  }
  for (int k = ty; k < 64; k += 8) {
    const size_t r = r0 + k;
    if (r >= m || r < cm.row0 || r >= cm.row1) continue;
    const size_t ro = r - cm.row0;  // output row
    T a0 = T(0), a1 = T(0);
#pragma unroll
    for (int e = 0; e < XC; e++) {
      if (e >= io.count) break;
      const T lv = center(Lcols[((size_t)io.column[e] * cm.L + limb) * Mlen + r], mod);
      if (io.k[e] == 0) {
        a0 = addm(a0, lv, mod);
        a1 = addm(a1, lv, mod);
      } else {
        a0 = addm(a0, mulmod_dd(xa[e], lv, mod), mod);
        a1 = addm(a1, mulmod_dd(xb[e], lv, mod), mod);
      }
      if ((e & 3) == 3) {
        a0 = reduce(a0, mod);
        a1 = reduce(a1, mod);
      }
    }
    for (int e = XC; e < io.count; e++) {
      const T lv = center(Lcols[((size_t)io.column[e] * cm.L + limb) * Mlen + r], mod);
      const int kk = io.k[e];
      if (kk == 0) {
        a0 = addm(a0, lv, mod);
        a1 = addm(a1, lv, mod);
      } else {
        const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(asg + (size_t)(kk - 1) * Si)[pair];
        a0 = addm(a0, mulmod_dd(from_res<T>(v.x), lv, mod), mod);
        a1 = addm(a1, mulmod_dd(from_res<T>(v.y), lv, mod), mod);
      }
      if ((e & 3) == 3) {
        a0 = reduce(a0, mod);
        a1 = reduce(a1, mod);
      }
    }
    a0 = canon(a0, mod);
    a1 = canon(a1, mod);
    if (io_out) {
      ulonglong2 o;
      o.x = to_res(a0);
      o.y = to_res(a1);
      reinterpret_cast<ulonglong2 *>(io_out + ro * So)[opair] = o;
    }
    const T cc = cst ? cst[(size_t)limb * Mlen + r] : T(0);
    ulonglong2 o;
    o.x = to_res(canon(addm(subm(tile[2 * tx][k], a0, mod), cc, mod), mod));
    o.y = to_res(canon(addm(subm(tile[2 * tx + 1][k], a1, mod), cc, mod), mod));
    reinterpret_cast<ulonglong2 *>(mid_out + ro * So)[opair] = o;
  }
}

// coefficients_for_X_mid = interp(full) - interp(io) + interp(constant part), in place over `full`.
// (The reference evaluates index-0 terms in BOTH the io and the mid pass, r1cs_to_qrp.tcc:175-201.)
template <class CPS>
__global__ void __launch_bounds__(256)
mid_kernel(typename CPS::T *__restrict__ full, const typename CPS::T *__restrict__ io,
           const typename CPS::T *__restrict__ cst /* [Ltot][M] or null */, size_t M, size_t S, unsigned slots_per_limb, CPS plans,
           int limb0, const typename CPS::T *__restrict__ cst_cols /* [S][M] or null: a constant part that differs per slot */) {
  using T = typename CPS::T;
  const size_t total = S * M, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t col = i / M, k = i % M;
    const int limb = (int)(col / slots_per_limb);  // chunk-local: plans are shifted by limb0
    const typename CPS::M mod = plans.l[limb].mod;
    T v = subm(full[i], io[i], mod);
    if (cst) v = addm(v, cst[(size_t)(limb0 + limb) * M + k], mod);
    if (cst_cols) v = addm(v, cst_cols[i], mod);
    full[i] = canon(v, mod);
  }
}

// one row of linear_combination::evaluate for a slot pair: sum_e coeff_e * x_{col_e} (index 0 = the constant one).
// coeff_e is a slot-constant scalar, or -- pidx[e] >= 0 -- a general ring element: row pidx[e] of the table, whose two
// residues for this slot pair sit at ptab_pair + pidx[e] * Si (the table has the assignment's [L][N] layout).
#define RS_EVAL_CONST 3 /* internal mode: the index-0 terms only (the constant part of a mid vector) */
template <class M>
__device__ __forceinline__ void eval_row_pair(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col,
                                              const typename ArithOf<M>::T *__restrict__ coeff_limb, size_t row,
                                              const uint64_t *__restrict__ asg, size_t Si, size_t pair, int mode, unsigned n_inputs,
                                              const M mod, typename ArithOf<M>::T &o0, typename ArithOf<M>::T &o1,
                                              const int32_t *__restrict__ pidx, const typename ArithOf<M>::T *__restrict__ ptab) {
  using T = typename ArithOf<M>::T;
  T a0 = T(0), a1 = T(0);
  int since = 0;
  for (uint32_t e = row_ptr[row]; e < row_ptr[row + 1]; e++) {
    const uint32_t cv = col[e];
    T cf0 = coeff_limb[e], cf1 = cf0;  // table constants
    if (pidx) {
      const int32_t pk = pidx[e];
      if (pk >= 0) {
        const T *pc = ptab + (size_t)pk * Si + 2 * pair;
        cf0 = pc[0];
        cf1 = pc[1];
      }
    }
    if (cv == 0) {
      a0 = addm(a0, konst_value(cf0, mod), mod);
      a1 = addm(a1, konst_value(cf1, mod), mod);
    } else {
      const bool is_input = (cv - 1) < n_inputs;
      if ((mode == RS_EVAL_IO && !is_input) || (mode == RS_EVAL_MID && is_input) || mode == RS_EVAL_CONST) continue;
      const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(asg + (size_t)(cv - 1) * Si)[pair];
      a0 = addm(a0, mulmod(from_res<T>(v.x), cf0, mod), mod);
      a1 = addm(a1, mulmod(from_res<T>(v.y), cf1, mod), mod);
    }
    if (++since == 4) {
      since = 0;
      a0 = reduce(a0, mod);
      a1 = reduce(a1, mod);
    }
  }
  o0 = canon(a0, mod);
  o1 = canon(a1, mod);
}

// eval_row_pair for R rows (row0, row0 + step, ...) of one slot pair AT ONCE: the R chains of dependent loads advance together,
// every load unconditional on a clamped index (loads under a lane predicate would be waited for inside their branch, one row
// after the other), the terms a row does not have or does not want are dropped at the accumulation.  Same sums in the same
// order as eval_row_pair (rows >= m evaluate to zero).
template <class M, int R>
__device__ __forceinline__ void eval_rows_pair(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col,
                                               const typename ArithOf<M>::T *__restrict__ coeff_limb, size_t row0, size_t step, size_t m,
                                               const uint64_t *__restrict__ asg, size_t Si, size_t pair, int mode, unsigned n_inputs,
                                               const M mod, typename ArithOf<M>::T (&o0)[R], typename ArithOf<M>::T (&o1)[R],
                                               const int32_t *__restrict__ pidx, const typename ArithOf<M>::T *__restrict__ ptab) {
  using T = typename ArithOf<M>::T;
  uint32_t e[R], end[R];
  T a0[R], a1[R];
  int since[R];
#pragma unroll
  for (int j = 0; j < R; j++) {
    const size_t row = row0 + (size_t)j * step;
    const bool ok = row < m;
    e[j] = ok ? row_ptr[row] : 0u;
    end[j] = ok ? row_ptr[row + 1] : 0u;
    a0[j] = a1[j] = T(0);
    since[j] = 0;
  }
  for (;;) {
    bool more = false;
#pragma unroll
    for (int j = 0; j < R; j++) more |= e[j] < end[j];
    if (!more) break;  // some row has a term: the matrix has non-zeros, index 0 is a valid clamp
    uint32_t cv[R];
    T cf0[R], cf1[R];
#pragma unroll
    for (int j = 0; j < R; j++) {
      const uint32_t i = e[j] < end[j] ? e[j] : 0u;
      cv[j] = col[i];
      cf0[j] = cf1[j] = coeff_limb[i];  // table constants
    }
    if (pidx) {
#pragma unroll
      for (int j = 0; j < R; j++) {
        const int32_t pk = pidx[e[j] < end[j] ? e[j] : 0u];
        if (pk >= 0) {
          const T *pc = ptab + (size_t)pk * Si + 2 * pair;
          cf0[j] = pc[0];
          cf1[j] = pc[1];
        }
      }
    }
    ulonglong2 v[R];
    if (mode != RS_EVAL_CONST) {
#pragma unroll
      for (int j = 0; j < R; j++) v[j] = reinterpret_cast<const ulonglong2 *>(asg + (size_t)(cv[j] ? cv[j] - 1 : 0u) * Si)[pair];
    } else {
#pragma unroll
      for (int j = 0; j < R; j++) v[j] = ulonglong2{0ull, 0ull};
    }
#pragma unroll
    for (int j = 0; j < R; j++) {
      if (e[j] >= end[j]) continue;
      e[j]++;
      if (cv[j] == 0) {
        a0[j] = addm(a0[j], konst_value(cf0[j], mod), mod);
        a1[j] = addm(a1[j], konst_value(cf1[j], mod), mod);
      } else {
        const bool is_input = (cv[j] - 1) < n_inputs;
        if ((mode == RS_EVAL_IO && !is_input) || (mode == RS_EVAL_MID && is_input) || mode == RS_EVAL_CONST) continue;
        a0[j] = addm(a0[j], mulmod(from_res<T>(v[j].x), cf0[j], mod), mod);
        a1[j] = addm(a1[j], mulmod(from_res<T>(v[j].y), cf1[j], mod), mod);
      }
      if (++since[j] == 4) {
        since[j] = 0;
        a0[j] = reduce(a0[j], mod);
        a1[j] = reduce(a1[j], mod);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < R; j++) {
    o0[j] = canon(a0[j], mod);
    o1[j] = canon(a1[j], mod);
  }
}

// a14: linear_combination::evaluate for every constraint (relations/variable.tcc:246-254).
// grid (m, ceil(L*N/512)); each thread handles two adjacent slots.
template <class M>
__global__ void __launch_bounds__(256)
r1cs_eval_kernel(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col,
                 const typename ArithOf<M>::T *__restrict__ coeff, size_t nnz, const uint64_t *__restrict__ asg,
                 uint64_t *__restrict__ out, int N, int L, int mode, unsigned n_inputs, const M *__restrict__ qmod,
                 const int32_t *__restrict__ pidx, const typename ArithOf<M>::T *__restrict__ ptab) {
  using T = typename ArithOf<M>::T;
  const size_t row = blockIdx.x;
  const size_t S = (size_t)L * N;
  const size_t pair = (size_t)blockIdx.y * blockDim.x + threadIdx.x;
  if (2 * pair >= S) return;
  const int limb = (int)((2 * pair) / (size_t)N);
  T a0, a1;
  eval_row_pair<M>(row_ptr, col, coeff + (size_t)limb * nnz, row, asg, S, pair, mode, n_inputs, qmod[limb], a0, a1, pidx, ptab);
  ulonglong2 o;
  o.x = to_res(a0);
  o.y = to_res(a1);
  reinterpret_cast<ulonglong2 *>(out + row * S)[pair] = o;
}

// linear_combination::evaluate straight into the column-major layout of the witness map
// (r1cs_eval_kernel + transpose fused; rows >= m are the zero padding of the columns).
// grid (C/64, M/32): 64 columns x 32 rows per workgroup (17 KiB of LDS: eight workgroups per CU -- the evaluation is a chain of
// dependent loads, row_ptr -> col / coeff -> assignment, and lives on resident waves; 64-row tiles measured 45 % slower).
// Each lane evaluates its FOUR rows together (eval_rows_pair): four chains in flight instead of one after the other.
template <class M>
__global__ void __launch_bounds__(256)
r1cs_eval_cols_kernel(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col,
                      const typename ArithOf<M>::T *__restrict__ coeff, size_t nnz, const uint64_t *__restrict__ asg,
                      typename ArithOf<M>::T *__restrict__ cols, size_t m, size_t C, size_t Mlen, int mode, unsigned n_inputs,
                      const M *__restrict__ qmod, ColMap cm, const int32_t *__restrict__ pidx,
                      const typename ArithOf<M>::T *__restrict__ ptab) {
  using T = typename ArithOf<M>::T;
  __shared__ T tile[64][33];  // [column][row]
  const size_t s0 = (size_t)blockIdx.x * 64, r0 = (size_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const size_t c = s0 + 2 * tx;  // column pair of this lane, 16-byte loads of the assignment
  if (c < C) {
    int limb, slot;
    cm.locate(c, limb, slot);
    const size_t pair = cm.in_index(limb, slot) >> 1, Si = cm.in_stride();
    const M mod = qmod[limb];
    T a0[4], a1[4];
    eval_rows_pair<M, 4>(row_ptr, col, coeff + (size_t)limb * nnz, r0 + ty, 8, m, asg, Si, pair, mode, n_inputs, mod, a0, a1, pidx, ptab);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      tile[2 * tx][ty + 8 * j] = a0[j];
      tile[2 * tx + 1][ty + 8 * j] = a1[j];
    }
  }
  __syncthreads();
  for (int k = ty; k < 64; k += 8) {
    const size_t cc = s0 + k, r = r0 + tx;
    if (cc < C && r < Mlen) cols[cc * Mlen + r] = tile[k][tx];
  }
}

// H[m] when m == M (the column tile holds M rows only): d1*d2*Z[m] = d1*d2 (Z monic), zero without ZK
template <class M>
__global__ void __launch_bounds__(256)
h_top_kernel(uint64_t *__restrict__ top, const uint64_t *__restrict__ d1, const uint64_t *__restrict__ d2, size_t C,
             const M *__restrict__ qmod, ColMap cm) {
  using T = typename ArithOf<M>::T;
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  int limb, slot;
  cm.locate(c, limb, slot);
  const M mod = qmod[limb];
  uint64_t v = 0;
  if (d1) {
    const size_t di = cm.in_index(limb, slot);
    v = to_res(canon(mulmod_dd(center(from_res<T>(d1[di]), mod), center(from_res<T>(d2[di]), mod), mod), mod));
  }
  top[cm.out_index(limb, slot)] = v;
}

}  // namespace rs
