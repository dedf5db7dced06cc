// store_shape.hip -- does a lane-strided 16-byte store pattern (each lane fills its own 256 contiguous bytes with sixteen
// stores, lanes 256 bytes apart) reach the bandwidth of the coalesced pattern (64 lanes x 16 contiguous bytes per store)?
// The question behind it: may the last round of the wide transforms (32 consecutive elements per lane) store straight to
// global memory instead of parking its results in LDS for a coalesced flush.   hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// one "polynomial" = 16384 x 8 bytes = 128 KiB per workgroup iteration, 512 threads, 256 bytes per thread
template <int MODE, bool NT, bool LOAD>
__global__ void __launch_bounds__(512) k(u64x2 *dst, const u64x2 *src, unsigned long long batch) {
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  for (unsigned long long p = blockIdx.x; p < batch; p += gridDim.x) {
    u64x2 *d = dst + p * 8192;
    const u64x2 *s = src + p * 8192;
    u64x2 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (LOAD) v[i] = __builtin_nontemporal_load(s + t + 512 * i);  // coalesced loads, as the forward transform's
      else v[i] = u64x2{(unsigned long long)t + i, p};
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
      u64x2 *q = MODE == 0 ? d + wave * 1024 + lane + 64 * i    // coalesced: a wave writes 1 KiB per store
                           : d + t * 16 + i;                     // lane-strided: lane t owns bytes [256 t, 256 t + 256)
      if (NT) __builtin_nontemporal_store(v[i], q); else *q = v[i];
    }
  }
}
template <int MODE, bool NT, bool LOAD>
static void run(const char *name, u64x2 *dst, u64x2 *src, unsigned long long batch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k<MODE, NT, LOAD>), dim3(256), dim3(512), 0, 0, dst, src, batch);
  CK(hipEventRecord(e0));
  for (int r = 0; r < 10; r++) hipLaunchKernelGGL((k<MODE, NT, LOAD>), dim3(256), dim3(512), 0, 0, dst, src, batch);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)batch * 131072.0 * (LOAD ? 2 : 1) * 10;
  printf("%-44s %8.1f GB/s\n", name, bytes / ms / 1e6);
}
int main() {
  const unsigned long long batch = 16384;  // 2 GiB
  u64x2 *dst, *src;
  CK(hipMalloc(&dst, batch * 131072)); CK(hipMalloc(&src, batch * 131072));
  CK(hipMemset(src, 1, batch * 131072));
  run<0, true, false>("store only, coalesced, non-temporal", dst, src, batch);
  run<1, true, false>("store only, lane-strided, non-temporal", dst, src, batch);
  run<0, false, false>("store only, coalesced", dst, src, batch);
  run<1, false, false>("store only, lane-strided", dst, src, batch);
  run<0, true, true>("copy, coalesced stores, non-temporal", dst, src, batch);
  run<1, true, true>("copy, lane-strided stores, non-temporal", dst, src, batch);
  run<0, false, true>("copy, coalesced stores", dst, src, batch);
  run<1, false, true>("copy, lane-strided stores", dst, src, batch);
  return 0;
}
