// Can an HBM-bound streaming kernel run UNDER a register-hungry, LDS-hungry FP64 kernel on the same CUs?
// (DESIGN.md section 7 "next" #2: the prover alternates VALU-bound tile kernels and HBM-bound passes; each idles the other's
// resource.)  A: a stand-in for the tile kernels -- 256 threads, NV live doubles per lane (2 NV VGPRs), 74 KiB of LDS, two
// workgroups per CU, FP64 FMAs only.  B: a 2 GiB streaming copy.  Times: A alone, B alone, A and B on two streams.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/coresidency.hip -o /tmp/coresidency && /tmp/coresidency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NV>
__global__ void __launch_bounds__(256, 2) tile_like(double *out, int iters, double b) {
  extern __shared__ double s[];
  double x[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) x[i] = 1.0 + threadIdx.x + i;
  s[threadIdx.x] = b;
  __syncthreads();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NV; i++) x[i] = __builtin_fma(x[i], b, x[(i + 1) % NV] * 1e-9);
    if ((it & 63) == 63) {  // a tile exchange now and then
      s[(threadIdx.x * 33 + it) & 8191] = x[it % NV == 0 ? 0 : 1];
      __syncthreads();
      x[0] += s[(threadIdx.x + it) & 8191];
    }
  }
  double acc = 0;
#pragma unroll
  for (int i = 0; i < NV; i++) acc += x[i];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) copy_k(const u64x2 *__restrict__ src, u64x2 *__restrict__ dst, size_t n16) {
  for (size_t base = (size_t)blockIdx.x * 2048; base + 2048 <= n16; base += (size_t)gridDim.x * 2048) {
    u64x2 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(src + base + threadIdx.x + 256 * k);
#pragma unroll
    for (int k = 0; k < 8; k++) __builtin_nontemporal_store(v[k], dst + base + threadIdx.x + 256 * k);
  }
}
template <int NV>
static int run(const char *label, int lds_bytes, int a_launches = 1, bool b_memcpy = false) {
  hipStream_t sa, sb;
  CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  double *out;
  CHECK(hipMalloc(&out, 512 * 256 * 8));
  const size_t bytes = (size_t)2 << 30;
  void *a, *b;
  CHECK(hipMalloc(&a, bytes));
  CHECK(hipMalloc(&b, bytes));
  CHECK(hipMemset(a, 1, bytes));
  CHECK(hipFuncSetAttribute((const void *)tile_like<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 20000, copies = 12;
  auto launch_a = [&]() { for (int l = 0; l < a_launches; l++) hipLaunchKernelGGL(tile_like<NV>, dim3(512), dim3(256), lds_bytes, sa, out, iters / a_launches, 1.0000001); };
  auto launch_b = [&]() {
    for (int i = 0; i < copies; i++) {
      if (b_memcpy) (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, sb);
      else hipLaunchKernelGGL(copy_k, dim3(2048), dim3(256), 0, sb, (const u64x2 *)a, (u64x2 *)b, bytes / 16);
    }
  };
  launch_a(); launch_b(); CHECK(hipDeviceSynchronize());
  float ta, tb, tab;
  CHECK(hipEventRecord(e0, sa)); launch_a(); CHECK(hipEventRecord(e1, sa)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ta, e0, e1));
  CHECK(hipEventRecord(e0, sb)); launch_b(); CHECK(hipEventRecord(e1, sb)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&tb, e0, e1));
  // both: A first (it takes the CUs), B right behind it on the other stream; wall time until both are done
  hipEvent_t ea, eb;
  CHECK(hipEventCreate(&ea)); CHECK(hipEventCreate(&eb));
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, sa));
  launch_a(); CHECK(hipEventRecord(ea, sa));
  launch_b(); CHECK(hipEventRecord(eb, sb));
  CHECK(hipEventSynchronize(ea)); CHECK(hipEventSynchronize(eb));
  float t1, t2;
  CHECK(hipEventElapsedTime(&t1, e0, ea));
  CHECK(hipEventElapsedTime(&t2, e0, eb));
  tab = t1 > t2 ? t1 : t2;
  printf("%-44s A alone %7.2f ms | B alone %7.2f ms (%.0f GB/s) | together %7.2f ms (A done at %.2f, B at %.2f) -> %.2f of the sum, %.2f of the max\n",
         label, ta, tb, copies * 2.0 * bytes / tb / 1e6, tab, t1, t2, tab / (ta + tb), tab / (ta > tb ? ta : tb));
  hipFree(out); hipFree(a); hipFree(b);
  return 0;
}
int main() {
  if (run<124>("A: 254 VGPRs, 1 launch; B: kernel", 74 * 1024)) return 1;
  if (run<124>("A: 254 VGPRs, 20 launches; B: kernel", 74 * 1024, 20)) return 1;
  if (run<124>("A: 254 VGPRs, 200 launches; B: kernel", 74 * 1024, 200)) return 1;
  if (run<124>("A: 254 VGPRs, 1 launch; B: hipMemcpyAsync", 74 * 1024, 1, true)) return 1;
  if (run<124>("A: 254 VGPRs, 200 launches; B: hipMemcpyAsync", 74 * 1024, 200, true)) return 1;
  if (run<104>("A: 214 VGPRs, 200 launches; B: kernel", 74 * 1024, 200)) return 1;
  if (run<88>("A: 182 VGPRs, 200 launches; B: kernel", 74 * 1024, 200)) return 1;
  return 0;
}
