// Probe: does a global_load_lds (LDS-DMA) wave-instruction hold LGKM_CNT until its data has landed in LDS?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const float* src, unsigned long long* out) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const float* g = src + (size_t)threadIdx.x * 1024 * 64 + blockIdx.x * 4;  // every lane a different, cold line
  unsigned long long t0, t1, t2;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
  if (threadIdx.x == 0) {
    out[0] = t1 - t0;
    out[1] = t2 - t0;
    out[2] = (unsigned long long)lds[5];
  }
}
int main() {
  float* src; unsigned long long* out;
  hipMalloc(&src, (size_t)64 * 1024 * 64 * 4 + 4096); hipMemset(src, 0, (size_t)64 * 1024 * 64 * 4 + 4096);
  hipMalloc(&out, 64);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src + rep * 32, out);
    unsigned long long h[3]; hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
    printf("rep %d: issue -> lgkmcnt(0) returns after %llu cycles; -> vmcnt(0) after %llu cycles\n", rep, h[0], h[1]);
  }
  return 0;
}
