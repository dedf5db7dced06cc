// poly.hip -- the generic polynomial helpers of the reference's util/polynomials.tcc:62-81 (row a11):
// multiply, add and divide of polynomials whose coefficients are ring elements.  Every ring operation is
// slot-wise, so these are L*N independent polynomial operations over the prime fields F_{q_i}.  The prover
// hot path does not go through them (the witness map has its own quasi-linear form, witness.hip); they exist
// so that code written against the reference's helpers -- and the reference's own util/division_test.cpp --
// runs on the device.  Schoolbook, as in the reference (Boost.Math polynomial): O(na*nb) ring operations.
#include <algorithm>
#include <vector>

#include "rs_internal.hpp"

namespace rs {

// out[k] = sum_i a[i] * b[k - i], one thread per (k, slot pair)
template <class M>
__global__ void __launch_bounds__(256)
poly_mul_kernel(const uint64_t *__restrict__ a, size_t na, const uint64_t *__restrict__ b, size_t nb, uint64_t *__restrict__ out,
                int N, int L, const M *__restrict__ qmod) {
  using T = typename ArithOf<M>::T;
  const size_t S = (size_t)L * N, k = blockIdx.x;
  const size_t pair = (size_t)blockIdx.y * blockDim.x + threadIdx.x;
  if (2 * pair >= S) return;
  const M mod = qmod[(2 * pair) / (size_t)N];
  const size_t i0 = k >= nb ? k - nb + 1 : 0, i1 = std::min(k, na - 1);
  T a0 = T(0), a1 = T(0);
  int since = 0;
  for (size_t i = i0; i <= i1; i++) {
    const ulonglong2 x = reinterpret_cast<const ulonglong2 *>(a + i * S)[pair];
    const ulonglong2 y = reinterpret_cast<const ulonglong2 *>(b + (k - i) * S)[pair];
    a0 = addm(a0, mulmod_dd(from_res<T>(x.x), center(from_res<T>(y.x), mod), mod), mod);
    a1 = addm(a1, mulmod_dd(from_res<T>(x.y), center(from_res<T>(y.y), mod), mod), mod);
    if (++since == 4) {
      since = 0;
      a0 = reduce(a0, mod);
      a1 = reduce(a1, mod);
    }
  }
  ulonglong2 o;
  o.x = to_res(canon(a0, mod));
  o.y = to_res(canon(a1, mod));
  reinterpret_cast<ulonglong2 *>(out + k * S)[pair] = o;
}

// Long division per slot: rem (a copy of the numerator, nn rows) is reduced in place, quot gets nn - nd + 1
// rows.  lead_inv: inverse of the divisor's leading coefficient [L][N].  One thread per slot.
template <class M>
__global__ void __launch_bounds__(256)
poly_div_kernel(uint64_t *__restrict__ rem, size_t nn, const uint64_t *__restrict__ den, size_t nd,
                const uint64_t *__restrict__ lead_inv, uint64_t *__restrict__ quot, int N, int L, const M *__restrict__ qmod) {
  using T = typename ArithOf<M>::T;
  const size_t S = (size_t)L * N, sl = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (sl >= S) return;
  const M mod = qmod[sl / (size_t)N];
  const T li = to_mont(center(from_res<T>(lead_inv[sl]), mod), mod);  // used as a multiplier below
  for (size_t k = nn - nd + 1; k-- > 0;) {
    const T qk = reduce(mulmod(center(from_res<T>(rem[(k + nd - 1) * S + sl]), mod), li, mod), mod);
    quot[k * S + sl] = to_res(canon(qk, mod));
    const T qk_m = to_mont(qk, mod);
    for (size_t j = 0; j < nd; j++) {
      const T r = subm(from_res<T>(rem[(k + j) * S + sl]), mulmod(center(from_res<T>(den[j * S + sl]), mod), qk_m, mod), mod);
      rem[(k + j) * S + sl] = to_res(canon(r, mod));
    }
  }
}

template <class M>
__global__ void __launch_bounds__(256)
poly_add_kernel(const uint64_t *__restrict__ a, size_t na, const uint64_t *__restrict__ b, size_t nb, uint64_t *__restrict__ out,
                size_t S, int N, int L, const M *__restrict__ qmod) {
  using T = typename ArithOf<M>::T;
  const size_t rows = std::max(na, nb), total = rows * S, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t r = i / S, sl = i % S;
    const M mod = qmod[sl / (size_t)N];
    const T x = r < na ? from_res<T>(a[i]) : T(0), y = r < nb ? from_res<T>(b[i]) : T(0);
    out[i] = to_res(canon(addm(x, y, mod), mod));
  }
}

// per-row "has a non-zero word" flags
__global__ void __launch_bounds__(256) rows_nonzero_kernel(const uint64_t *__restrict__ a, size_t S, unsigned *__restrict__ flags) {
  const uint64_t *e = a + (size_t)blockIdx.x * S;
  bool nz = false;
  for (size_t i = threadIdx.x; i < S; i += blockDim.x) nz |= (e[i] != 0);
  if (__syncthreads_or(nz) && threadIdx.x == 0) flags[blockIdx.x] = 1u;
}

// Boost's polynomial normalisation: length after stripping trailing coefficients equal to RingT(0)
static size_t normalised_len(rs_ctx *ctx, const uint64_t *d, size_t n, hipStream_t st) {
  if (n == 0) return 0;
  unsigned *flags = (unsigned *)ws_get(ctx, 7, std::max<size_t>(256, n * 4));
  RS_HIP(hipMemsetAsync(flags, 0, n * 4, st));
  hipLaunchKernelGGL(rows_nonzero_kernel, dim3((unsigned)n), dim3(256), 0, st, d, ctx->ring_words(), flags);
  std::vector<unsigned> h(n);
  RS_HIP(hipMemcpyAsync(h.data(), flags, n * 4, hipMemcpyDeviceToHost, st));
  RS_HIP(hipStreamSynchronize(st));
  while (n > 0 && !h[n - 1]) n--;
  return n;
}

}  // namespace rs

using namespace rs;

extern "C" {

int rs_poly_multiply(rs_ctx *ctx, const uint64_t *d_a, size_t na, const uint64_t *d_b, size_t nb, uint64_t *d_out, size_t *h_len,
                     rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_out && (d_a || na == 0) && (d_b || nb == 0), "null argument");
  WsScope ws_scope(ctx, S(stream));
  hipStream_t st = S(stream);
  const size_t S_ = ctx->ring_words();
  const size_t nominal = (na && nb) ? na + nb - 1 : 0;  // rows the caller allocated
  // operands are normalised first, as Boost's polynomial(vector) constructor does
  na = normalised_len(ctx, d_a, na, st);
  nb = normalised_len(ctx, d_b, nb, st);
  size_t len = 0, written = 0;
  if (na && nb) {
    const size_t rows = written = na + nb - 1;
    const unsigned by = (unsigned)((S_ / 2 + 255) / 256);
    if (ctx->use_int)
      hipLaunchKernelGGL(poly_mul_kernel<ModI>, dim3((unsigned)rows, by), dim3(256), 0, st, d_a, na, d_b, nb, d_out, ctx->N, ctx->L, ctx->d_qmod_i);
    else
      hipLaunchKernelGGL(poly_mul_kernel<Mod>, dim3((unsigned)rows, by), dim3(256), 0, st, d_a, na, d_b, nb, d_out, ctx->N, ctx->L, ctx->d_qmod);
    RS_HIP(hipGetLastError());
    len = normalised_len(ctx, d_out, rows, st);
  }
  // the header's contract: every row of the nominal output at or beyond the result's length is zero
  if (nominal > written) RS_HIP(hipMemsetAsync(d_out + written * S_, 0, (nominal - written) * S_ * sizeof(uint64_t), st));
  if (h_len) *h_len = len;
  RS_API_END
}

int rs_poly_add(rs_ctx *ctx, const uint64_t *d_a, size_t na, const uint64_t *d_b, size_t nb, uint64_t *d_out, size_t *h_len,
                rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_out && (d_a || na == 0) && (d_b || nb == 0), "null argument");
  WsScope ws_scope(ctx, S(stream));
  hipStream_t st = S(stream);
  const size_t rows = std::max(na, nb), S_ = ctx->ring_words();
  if (rows) {
    const unsigned blocks = (unsigned)std::min<size_t>((rows * S_ + 255) / 256, 4096);
    if (ctx->use_int)
      hipLaunchKernelGGL(poly_add_kernel<ModI>, dim3(blocks), dim3(256), 0, st, d_a, na, d_b, nb, d_out, S_, ctx->N, ctx->L, ctx->d_qmod_i);
    else
      hipLaunchKernelGGL(poly_add_kernel<Mod>, dim3(blocks), dim3(256), 0, st, d_a, na, d_b, nb, d_out, S_, ctx->N, ctx->L, ctx->d_qmod);
    RS_HIP(hipGetLastError());
  }
  if (h_len) *h_len = normalised_len(ctx, d_out, rows, st);
  RS_API_END
}

int rs_poly_divide(rs_ctx *ctx, const uint64_t *d_num, size_t nn, const uint64_t *d_den, size_t nd, uint64_t *d_quot, size_t *h_len,
                   rs_stream stream) {
  RS_API_BEGIN_CTX(ctx)
  RS_REQUIRE(ctx && d_quot && d_den && (d_num || nn == 0), "null argument");
  hipStream_t st = S(stream);
  const size_t S_ = ctx->ring_words();
  size_t len = 0;
  const size_t nominal = nn >= nd ? nn - nd + 1 : 0;  // rows the caller allocated
  {
    WsScope ws_scope(ctx, st);
    nn = normalised_len(ctx, d_num, nn, st);
    nd = normalised_len(ctx, d_den, nd, st);
  }
  RS_REQUIRE(nd >= 1, "division by the zero polynomial");
  // a denominator that loses more trailing zero coefficients than the numerator has a quotient LONGER than the
  // nominal nn - nd + 1 rows: the caller must pass normalised lengths then (the C++ adapter does)
  RS_REQUIRE(nn < nd || nn - nd + 1 <= nominal, "quotient exceeds nn - nd + 1 rows: strip the denominator's zero leading coefficients");
  const size_t written = nn >= nd ? nn - nd + 1 : 0;
  if (nominal > written) RS_HIP(hipMemsetAsync(d_quot + written * S_, 0, (nominal - written) * S_ * sizeof(uint64_t), st));
  if (nn >= nd) {
    // the leading coefficient must be a unit of the ring (Boost divides by it: RingElem::operator/ ->
    // invert_inplace -> "element is not invertible in ring", seal_ring.tcc:87-103)
    uint64_t *lead_inv = nullptr, *rem = nullptr;
    RS_HIP(hipMalloc(&lead_inv, S_ * sizeof(uint64_t)));
    struct Guard {
      void *a, *b;
      ~Guard() {
        (void)hipFree(a);
        (void)hipFree(b);
      }
    } guard{lead_inv, nullptr};
    const int rc = rs_ring_inv(ctx, lead_inv, d_den + (nd - 1) * S_, 1, stream);
    if (rc == RS_ERR_NOT_INVERTIBLE) throw Error(RS_ERR_NOT_INVERTIBLE, "element is not invertible in ring");
    RS_REQUIRE(rc == RS_OK, rs_last_error());
    RS_HIP(hipMalloc(&rem, nn * S_ * sizeof(uint64_t)));
    guard.b = rem;
    RS_HIP(hipMemcpyAsync(rem, d_num, nn * S_ * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    if (ctx->use_int)
      hipLaunchKernelGGL(poly_div_kernel<ModI>, dim3((unsigned)((S_ + 255) / 256)), dim3(256), 0, st, rem, nn, d_den, nd, lead_inv, d_quot,
                         ctx->N, ctx->L, ctx->d_qmod_i);
    else
      hipLaunchKernelGGL(poly_div_kernel<Mod>, dim3((unsigned)((S_ + 255) / 256)), dim3(256), 0, st, rem, nn, d_den, nd, lead_inv, d_quot,
                         ctx->N, ctx->L, ctx->d_qmod);
    RS_HIP(hipGetLastError());
    WsScope ws_scope(ctx, st);
    len = normalised_len(ctx, d_quot, nn - nd + 1, st);  // synchronises: rem / lead_inv may be freed
  }
  if (h_len) *h_len = len;
  RS_API_END
}

}  // extern "C"
