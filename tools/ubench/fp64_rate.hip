// FP64 instruction-rate microbenchmark (gfx950): cycles per wave-instruction for the ops of mulmod.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITERS 4096
template <int OP>
__global__ void k(double *out, double a0, double b0) {
  double x[8];
  for (int i = 0; i < 8; i++) x[i] = a0 + threadIdx.x + i;
  const double b = b0, c = 1.0 / 3.0;
  for (int it = 0; it < N_ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) x[i] = x[i] * b;
      if (OP == 1) x[i] = __builtin_fma(x[i], b, c);
      if (OP == 2) x[i] = x[i] + b;
      if (OP == 3) x[i] = __builtin_rint(x[i]) + 0.0;  // rndne (+0 folds away)
      if (OP == 4) x[i] = (x[i] + 6755399441055744.0) - 6755399441055744.0;  // magic rint: 2 adds
      if (OP == 5) { double h = x[i] * b; double l = __builtin_fma(x[i], b, -h); double kq = __builtin_rint(h * c); x[i] = __builtin_fma(-kq, 3.0, h) + l; }
      if (OP == 6) { double h = x[i] * b; double l = __builtin_fma(x[i], b, -h); double kq = (__builtin_fma(h, c, 6755399441055744.0)) - 6755399441055744.0; x[i] = __builtin_fma(-kq, 3.0, h) + l; }
    }
  }
  double s = 0; for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name, int ops_per) {
  double *d; hipMalloc(&d, 256 * 8 * 1024 * sizeof(double));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8, thr = 256;  // 8 waves/SIMD
  k<OP><<<blocks, thr>>>(d, 1.5, 1.0000001);
  hipEventRecord(e0); k<OP><<<blocks, thr>>>(d, 1.5, 1.0000001); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double winstr = (double)blocks * (thr / 64) * N_ITERS * 8 * ops_per;  // wave-instructions
  double per_simd = winstr / (256 * 4);
  printf("%-28s %.3f ms  %.2f cycles/wave-instr/SIMD (at 2.4 GHz)  %.2f T lane-ops/s\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd, winstr * 64 / ms / 1e9);
  hipFree(d);
}
int main() {
  run<0>("v_mul_f64", 1); run<1>("v_fma_f64", 1); run<2>("v_add_f64", 1); run<3>("v_rndne_f64", 1);
  run<4>("magic rint (2 adds)", 2); run<5>("mulmod (rndne) 6 ops", 6); run<6>("mulmod (magic fma) 6 ops", 6);
  return 0;
}
