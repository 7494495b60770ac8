// Which fp32 MFMA shape does the chip clock higher under load?  Bare loops, operands in registers, random data:
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_f32_shape_probe tools/probes/mfma_f32_shape_probe.hip && tools/probes/mfma_f32_shape_probe [waves_per_simd]
// Prints TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime) of v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 at the same register footprint (128 accumulator registers per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// KIND of the filler: 0 v_fma_f32, 1 v_pk_fma_f32 (two fp32 per lane), 2 v_add_f32, 3 ds_read_b128 (LDS), 4 v_pk_add_f32
template <int SHAPE, int NV = 0, int WPS = 1, int KIND = 0>
__global__ __launch_bounds__(256, WPS) void probe(const float* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ clk, int iters) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = in[(tid * 16 + i) & 0xFFFFF];
    b[i] = in[(tid * 16 + 8 + i) & 0xFFFFF];
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
  float fill[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 fill2[4], pa[4], pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    fill2[i] = f32x2{(float)i, 1.f};
    pa[i] = f32x2{a[i], a[i + 4]};
    pb[i] = f32x2{b[i], b[i + 4]};
  }
  __shared__ float ldsbuf[4096];
  ldsbuf[threadIdx.x] = a[0];
  f32x4 ldsv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const unsigned ldsaddr = (unsigned)(threadIdx.x & 63) * 16u;
  __syncthreads();
  if (SHAPE == 32) {
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + k) & 7], b[(i * 3 + k) & 7], acc[i], 0, 0, 0);  // 32 MFMAs x 4096 FLOP
#pragma unroll
          for (int v = 0; v < NV; ++v) {  // NV independent filler instructions per MFMA
            if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(fill[v & 7]) : "v"(a[v & 7]), "v"(b[(v + 1) & 7]));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(fill2[v & 3]) : "v"(pa[v & 3]), "v"(pb[(v + 1) & 3]));
            if (KIND == 2) asm volatile("v_add_f32 %0, %1, %0" : "+v"(fill[v & 7]) : "v"(a[v & 7]));
            if (KIND == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(ldsv[v & 1]) : "v"(ldsaddr));
            if (KIND == 4) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(fill2[v & 3]) : "v"(pa[v & 3]));
          }
          if (KIND == 3 && NV > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
#pragma unroll
    for (int v = 0; v < 8; ++v) sum += fill[v];
#pragma unroll
    for (int v = 0; v < 4; ++v) sum += fill2[v][0] + fill2[v][1];
    sum += ldsv[0][0] + ldsv[1][3];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += acc[i][r];
  } else {
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + k) & 7], b[(i * 3 + k) & 7], acc[i], 0, 0, 0);  // 64 MFMAs x 2048 FLOP
    }
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) sum += acc[i][r];
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[tid] = sum;
  if (threadIdx.x == 0) {
    clk[blockIdx.x * 2] = c1 - c0;
    clk[blockIdx.x * 2 + 1] = r1 - r0;
  }
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 1;  // waves per SIMD
  const int blocks = 256 * wps, iters = 20000;
  float *in, *out;
  unsigned long long* clk;
  hipMalloc(&in, (1 << 20) * 4); hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&clk, (size_t)blocks * 16);
  std::vector<float> h(1 << 20);
  srand(3);
  for (auto& v : h) v = (rand() & 0xffff) / 32768.f - 1.f;
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep)
    for (int shape : {32, 16}) {
      for (int w = 0; w < 2; ++w) {  // warm-up launch, then the timed one (each ~0.1 - 0.2 s)
        hipEventRecord(e0);
        if (shape == 32) {
          if (wps == 1) hipLaunchKernelGGL((probe<32, 0, 1>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
          else if (wps == 2) hipLaunchKernelGGL((probe<32, 0, 2>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
          else hipLaunchKernelGGL((probe<32, 0, 3>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
        } else {
          if (wps == 1) hipLaunchKernelGGL((probe<16, 0, 1>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
          else if (wps == 2) hipLaunchKernelGGL((probe<16, 0, 2>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
          else hipLaunchKernelGGL((probe<16, 0, 3>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> c(blocks * 2);
      hipMemcpy(c.data(), clk, c.size() * 8, hipMemcpyDeviceToHost);
      double cyc = 0, rt = 0;
      for (int i = 0; i < blocks; ++i) { cyc += c[2 * i]; rt += c[2 * i + 1]; }
      const double flop = (double)blocks * 4 * iters * 32 * 4096.0;
      printf("%s, %d wave(s) per SIMD: %.2f ms  %.1f TFLOP/s  in-kernel clock %.3f GHz  cycles per MFMA and SIMD %.1f\n", shape == 32 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_16x16x4_f32", wps, ms,
             flop / ms / 1e9, cyc / rt * 0.1, cyc / blocks / ((double)iters * (shape == 32 ? 32 : 64)) / wps);  // (per SIMD: a block is one wave per SIMD, wps blocks share a CU -- provided they are co-resident: launch_bounds(256, wps))
    }
  // do vector instructions overlap with the fp32 MFMA?  NV v_fma_f32 (independent of the MFMAs, eight separate chains) behind every v_mfma_f32_32x32x2_f32: overlapped, the time stays;
  // added, it grows by ~4 cycles (one wave per SIMD) or ~2 (two) per instruction
  auto fillrun = [&](auto nv_tag, auto kind_tag) {
    constexpr int NVv = decltype(nv_tag)::value;
    constexpr int KD = decltype(kind_tag)::value;
    const char* kn[5] = {"v_fma_f32", "v_pk_fma_f32", "v_add_f32", "ds_read_b128", "v_pk_add_f32"};
    for (int w = 0; w < 2; ++w) {
      hipEventRecord(e0);
      if (wps == 1) hipLaunchKernelGGL((probe<32, NVv, 1, KD>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
      else if (wps == 2) hipLaunchKernelGGL((probe<32, NVv, 2, KD>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
      else hipLaunchKernelGGL((probe<32, NVv, 3, KD>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks * 2);
    hipMemcpy(c.data(), clk, c.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < blocks; ++i) { cyc += c[2 * i]; rt += c[2 * i + 1]; }
    printf("32x32x2 + %2d %s per MFMA, %d wave(s) per SIMD: %.2f ms  clock %.3f GHz\n", NVv, kn[KD], wps, ms, cyc / rt * 0.1);
  };
  fillrun(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  fillrun(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
  fillrun(std::integral_constant<int, 4>{}, std::integral_constant<int, 0>{});
  fillrun(std::integral_constant<int, 8>{}, std::integral_constant<int, 0>{});
  fillrun(std::integral_constant<int, 12>{}, std::integral_constant<int, 0>{});
  fillrun(std::integral_constant<int, 6>{}, std::integral_constant<int, 1>{});   // six packed fmas = the arithmetic of twelve scalar ones
  fillrun(std::integral_constant<int, 12>{}, std::integral_constant<int, 1>{});
  fillrun(std::integral_constant<int, 12>{}, std::integral_constant<int, 2>{});
  fillrun(std::integral_constant<int, 6>{}, std::integral_constant<int, 4>{});
  fillrun(std::integral_constant<int, 12>{}, std::integral_constant<int, 4>{});
  // (KIND 3, ds_read_b128, waits for its data after every MFMA and so measures the LDS latency, not the issue cost: not run by default)
  return 0;
}
