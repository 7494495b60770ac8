// Stand-alone probe: what a read-only stream achieves on this GPU, by footprint (fits the 256-MB Infinity Cache or not), bytes in flight per wave and grid size.
// The peak-finding kernel (csrc/post_kernels.hip) reads 109 MB once per batch of 32 cfg3 frames: this is the ceiling it is measured against.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_read_probe hbm_read_probe.hip && ./hbm_read_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int U>
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ src, size_t n_vec, float* __restrict__ out) {
  // a block walks a contiguous range; each thread has U independent 16-byte loads in flight per trip
  const size_t per_block = (n_vec + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per_block, hi = lo + per_block < n_vec ? lo + per_block : n_vec;
  float m = -1e30f;
  for (size_t i = lo + threadIdx.x; i < hi; i += (size_t)256 * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t j = i + (size_t)256 * u;
      v[u] = j < hi ? src[j] : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) m = fmaxf(m, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
  }
  if (m > 1e29f) out[blockIdx.x] = m;  // (never: keeps the loads alive)
}

template <int U>
static void run(const float4* src, size_t bytes, int blocks, float* out, const char* what) {
  const size_t n_vec = bytes / 16;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(read_kernel<U>, dim3(blocks), dim3(256), 0, 0, src, n_vec, out);
  hipEventRecord(e0, 0);
  const int reps = 50;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(read_kernel<U>, dim3(blocks), dim3(256), 0, 0, src, n_vec, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = 1e3 * ms / reps;
  printf("%-28s %7.1f MB  blocks %6d  %2d x 16 B per thread in flight: %8.1f us  %7.0f GB/s (%.2f of 8 TB/s)\n", what, bytes / 1e6, blocks, U, us, bytes / us / 1e3, bytes / us / 1e3 / 8000);
}

int main() {
  const size_t big = (size_t)2 << 30;
  float4* src;
  float* out;
  hipMalloc(&src, big);
  hipMalloc(&out, 1 << 20);
  hipMemset(src, 0, big);
  const size_t sizes[] = {(size_t)109051904, (size_t)872415232, big};
  const char* names[] = {"cfg3 maps (cache-resident)", "8 x cfg3 maps", "2 GiB"};
  for (int s = 0; s < 3; ++s)
    for (int blocks : {1024, 2048, 4096, 16384}) {
      run<2>(src, sizes[s], blocks, out, names[s]);
      run<4>(src, sizes[s], blocks, out, names[s]);
      run<8>(src, sizes[s], blocks, out, names[s]);
    }
  return 0;
}
