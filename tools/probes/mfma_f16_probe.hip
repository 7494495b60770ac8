// Probe: does v_mfma_f32_32x32x16_f16 flush fp16 subnormal inputs?  How many cycles per MFMA back to back?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void probe(float* out, unsigned long long* cyc) {
  h8 a, b;
  const _Float16 sub = (_Float16)9.5367431640625e-07f;  // 2^-20: subnormal in fp16
  for (int j = 0; j < 8; ++j) { a[j] = sub; b[j] = (_Float16)1.0f; }
  f16v acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  out[threadIdx.x] = acc[0];
  // products below the f32 normal range of an fp16*fp16? (2^-24 * 2^-24 = 2^-48: fine in f32)
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)5.9604644775390625e-08f; b[j] = (_Float16)5.9604644775390625e-08f; }
  f16v acc2 = {0};
  acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
  out[64 + threadIdx.x] = acc2[0];
  // timing: 256 back-to-back MFMAs on 4 accumulators
  f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.5f + 0.01f * j); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[128 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 192 * 4); hipMalloc(&cyc, 8);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc);
  float h[192]; unsigned long long c;
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("subnormal(2^-20) x 1.0 summed over K=16: got %.10e expect %.10e  -> inputs %s\n", h[0], 16 * 9.5367431640625e-07, h[0] > 0 ? "NOT flushed" : "FLUSHED");
  printf("2^-24 x 2^-24 x16: got %.10e expect %.10e\n", h[64], 16 * 5.9604644775390625e-08 * 5.9604644775390625e-08);
  printf("256 MFMAs 32x32x16 f16 in %llu cycles -> %.1f cyc/MFMA\n", c, c / 256.0);
  return 0;
}
