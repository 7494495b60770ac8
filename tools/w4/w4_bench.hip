// Stand-alone timing harness of conv3x3_wino4_kernel (unity build of the kernel file; variants by -DW4_EXP=n):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o tools/w4/w4_bench tools/w4/w4_bench.hip && tools/w4/w4_bench [B H W c0 c1 cout lowres iters]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../sleap_nn_amd/csrc/wino4_kernels.hip"

namespace ph {
static thread_local char g_err[512];
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
int device_cu_count(int* out) {
  int dev = 0;
  hipGetDevice(&dev);
  return hipDeviceGetAttribute(out, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess ? PH_OK : PH_E_HIP;
}
}  // namespace ph

int main(int argc, char** argv) {
  int B = argc > 1 ? atoi(argv[1]) : 32, H = argc > 2 ? atoi(argv[2]) : 64, W = argc > 3 ? atoi(argv[3]) : 64;
  int c0 = argc > 4 ? atoi(argv[4]) : 256, c1 = argc > 5 ? atoi(argv[5]) : 512, cout = argc > 6 ? atoi(argv[6]) : 256;
  int lowres = argc > 7 ? atoi(argv[7]) : 1, iters = argc > 8 ? atoi(argv[8]) : 20;
  using namespace ph;
  if (prepare_wino4_kernels() != PH_OK) return printf("prepare failed: %s\n", g_err), 1;
  const size_t n0 = (size_t)B * H * W * c0, n1 = (size_t)B * (lowres ? H / 2 : H) * (lowres ? W / 2 : W) * c1, no = (size_t)B * H * W * cout;
  const int ntiles = (cout + 63) / 64, nchunks = (c0 + c1) / 16;
  const size_t nw = (size_t)wino4_pack_floats(ntiles, nchunks);
  std::vector<float> h(std::max(std::max(n0, n1), nw));
  srand(1);
  float *d0, *d1, *dw, *dout, *dbias;
  hipMalloc(&d0, n0 * 4); hipMalloc(&d1, std::max<size_t>(n1, 1) * 4); hipMalloc(&dw, nw * 4); hipMalloc(&dout, no * 4); hipMalloc(&dbias, 4096);
  auto fill = [&](float* d, size_t n, float s) { for (size_t i = 0; i < n; ++i) h[i] = s * ((rand() & 0xffff) / 32768.f - 1.f); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); };
  fill(d0, n0, 1.f); if (n1) fill(d1, n1, 1.f); fill(dw, nw, 0.05f); hipMemset(dbias, 0, 4096);
  ConvArgs a{};
  a.src0 = d0; a.src1 = c1 ? d1 : nullptr; a.c0p = c0; a.c1p = c1; a.coutp = cout; a.B = B; a.H = H; a.W = W; a.relu = 1; a.bn = 64;
  a.bias = dbias; a.dst = dout; a.wpack_wino4 = dw; a.src1_lowres = c1 ? lowres : 0; a.use_wino4 = 2;
  if (!wino4_fits(a)) return printf("shape does not fit\n"), 1;
#ifdef W4_STAMP
  unsigned long long* probe;
  hipMalloc(&probe, 256 * 12 * 8 * 8);
  hipMemset(probe, 0, 256 * 12 * 8 * 8);
  a.clock_probe = probe;
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) if (launch_conv3x3_wino4(a, 0) != PH_OK) return printf("launch failed: %s\n", g_err), 1;
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) launch_conv3x3_wino4(a, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
  const double direct = 2.0 * (c0 + c1) * cout * 9.0 * H * W * B;
  // determinism: two more launches into cleared buffers must agree bit for bit
  std::vector<float> o1(no), o2(no);
  hipMemset(dout, 0xff, no * 4); launch_conv3x3_wino4(a, 0); hipMemcpy(o1.data(), dout, no * 4, hipMemcpyDeviceToHost);
  hipMemset(dout, 0xff, no * 4); launch_conv3x3_wino4(a, 0); hipMemcpy(o2.data(), dout, no * 4, hipMemcpyDeviceToHost);
  size_t ndiff = 0, nnan = 0; double sum = 0;
  for (size_t i = 0; i < no; ++i) { ndiff += memcmp(&o1[i], &o2[i], 4) != 0; nnan += o1[i] != o1[i]; sum += o1[i]; }
  printf("   determinism: %zu of %zu elements differ, %zu NaN (unwritten), checksum %.6e\n", ndiff, no, nnan, sum);
#ifdef W4_STAMP
  {
    std::vector<unsigned long long> pr(256 * 12 * 8);
    hipMemcpy(pr.data(), probe, pr.size() * 8, hipMemcpyDeviceToHost);
    const int Q = (c0 + c1) / 4;
    const char* names[8] = {"frags+MFMA1", "t_rows", "MFMA2", "DMA issue", "t_cols+store", "MFMA3", "waitcnt", "barrier"};
    for (int slot = 0; slot < 3; ++slot) {
      double acc[8] = {0}; int n = 0;
      for (int blk = 0; blk < 256; ++blk) for (int w = slot * 4; w < slot * 4 + 4; ++w) { for (int i = 0; i < 8; ++i) acc[i] += (double)pr[((size_t)blk * 12 + w) * 8 + i]; ++n; }
      double tot = 0; for (int i = 0; i < 8; ++i) tot += acc[i];
      printf("   waves %d-%d (first tile of each workgroup, cycles per quarter):", slot * 4, slot * 4 + 3);
      for (int i = 0; i < 8; ++i) printf(" %s %.0f", names[i], acc[i] / n / Q);
      printf("  | total %.0f\n", tot / n / Q);
    }
  }
#endif
  hipMemcpy(h.data(), dout, 64 * 4, hipMemcpyDeviceToHost);
  printf("B %d %dx%d %d+%d->%d lowres %d: %.4f ms  executed %.1f TFLOP/s (%.3f of 157.3)  direct-equivalent %.1f  out[0..3] %g %g %g %g\n", B, H, W, c0, c1, cout, a.src1_lowres, ms,
         direct / 4 / ms / 1e9, direct / 4 / ms / 1e9 / 157.3, direct / ms / 1e9, h[0], h[1], h[2], h[3]);
  return 0;
}
