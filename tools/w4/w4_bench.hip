// Stand-alone timing harness of conv3x3_wino4_kernel (unity build of the kernel file; ablations by -DW4_EXP=n, in-kernel stamps by -DW4_STAMP [-DW4_STSEL=mask] [-DW4_STTILE],
// whole-workgroup cycles and clock by -DW4_CLOCK: tools/w4/build_variants.sh):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o tools/w4/w4_bench tools/w4/w4_bench.hip && tools/w4/w4_bench [B H W c0 c1 cout lowres iters]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include <algorithm>
#include "../../sleap_nn_amd/csrc/wino4_kernels.hip"
#ifdef W4S
#include "wino4s_experiment.inc"
#endif

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
// (split K is not exercised here: a.split_scratch stays null)
int wino2d_ksplit_shape(int, int, int, int, int, int, int) { return 1; }
int launch_splitk_reduce(const float*, long long, int, const float*, float*, float*, int, int, int, int, int, hipStream_t) { return PH_E_HIP; }
}  // namespace ph

int main(int argc, char** argv) {
  int B = argc > 1 ? atoi(argv[1]) : 32, H = argc > 2 ? atoi(argv[2]) : 64, W = argc > 3 ? atoi(argv[3]) : 64;
  int c0 = argc > 4 ? atoi(argv[4]) : 256, c1 = argc > 5 ? atoi(argv[5]) : 512, cout = argc > 6 ? atoi(argv[6]) : 256;
  int lowres = argc > 7 ? atoi(argv[7]) : 1, iters = argc > 8 ? atoi(argv[8]) : 20;
  using namespace ph;
  if (prepare_wino4_kernels() != PH_OK) return printf("prepare failed: %s\n", g_err), 1;
  const size_t n0 = (size_t)B * H * W * c0, n1 = (size_t)B * (lowres ? H / 2 : H) * (lowres ? W / 2 : W) * c1, no = (size_t)B * H * W * cout;
  const int ntiles = (cout + 63) / 64, nchunks = (c0 + c1) / 16;
  const size_t nw = (size_t)wino4_pack_floats(ntiles, nchunks), nwp = (size_t)ntiles * nchunks * 9 * 64 * 16;
  std::vector<float> h(std::max(std::max(n0, n1), nwp));
  srand(1);
  float *d0, *d1, *dwp, *dw, *dout, *dbias;
  hipMalloc(&d0, n0 * 4); hipMalloc(&d1, std::max<size_t>(n1, 1) * 4); hipMalloc(&dwp, nwp * 4); hipMalloc(&dw, nw * 4); hipMalloc(&dout, no * 4); hipMalloc(&dbias, 4096);
  auto fill = [&](float* d, size_t n, float s) { for (size_t i = 0; i < n; ++i) h[i] = s * ((rand() & 0xffff) / 32768.f - 1.f); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); };
  fill(d0, n0, 1.f); if (n1) fill(d1, n1, 1.f); fill(dwp, nwp, 0.05f); fill(dbias, 1024, 0.1f);
  if (launch_wino4_pack(dwp, dw, ntiles, nchunks, 0) != PH_OK) return printf("pack failed\n"), 1;  // random direct weights -> the kernel's order
  hipDeviceSynchronize();
  ConvArgs a{};
  a.src0 = d0; a.src1 = c1 ? d1 : nullptr; a.c0p = c0; a.c1p = c1; a.coutp = cout; a.B = B; a.H = H; a.W = W; a.relu = 1; a.bn = 64;
  a.bias = dbias; a.dst = dout; a.src1_lowres = c1 ? lowres : 0; a.use_wino4 = 2; a.wpack_wino4 = dw;
  if (!wino4_fits(a)) return printf("shape does not fit\n"), 1;
#if defined(W4_STAMP) || defined(W4_CLOCK)
  unsigned long long* probe;
  hipMalloc(&probe, 256 * 12 * 8 * 8);
  hipMemset(probe, 0, 256 * 12 * 8 * 8);
  a.clock_probe = probe;
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double direct = 2.0 * (c0 + c1) * cout * 9.0 * H * W * B;
  std::vector<float> o1(no), o2(no);
#ifdef W4S  // the sixteen-tile kernel on the same problem: its own weight order, its result compared with the eight-wave kernel's, then timed in its place
  std::vector<float> ref(no);
  {
    float* dws;
    hipMalloc(&dws, nw * 4);
    if (prepare_wino4s() != PH_OK || launch_wino4s_pack(dwp, dws, ntiles, nchunks, 0) != PH_OK) return printf("pack (s) failed\n"), 1;
    a.wpack_wino4s = dws;
    hipMemset(dout, 0xff, no * 4);
    if (launch_conv3x3_wino4(a, 0) != PH_OK) return printf("launch failed: %s\n", g_err), 1;
    hipMemcpy(ref.data(), dout, no * 4, hipMemcpyDeviceToHost);
    hipMemset(dout, 0xff, no * 4);
    if (launch_conv3x3_wino4s(a, 0) != PH_OK) return printf("launch (s) failed: %s\n", g_err), 1;
    if (hipDeviceSynchronize() != hipSuccess) return printf("kernel (s) failed: %s\n", hipGetErrorString(hipGetLastError())), 1;
    std::vector<float> got(no);
    hipMemcpy(got.data(), dout, no * 4, hipMemcpyDeviceToHost);
    double md = 0, mx = 0; size_t nn = 0;
    for (size_t i = 0; i < no; ++i) { if (got[i] != got[i]) { ++nn; continue; } md = std::max(md, (double)fabsf(got[i] - ref[i])); mx = std::max(mx, (double)fabsf(ref[i])); }
    printf("sixteen-tile kernel vs eight-wave kernel: max |diff| %.3e (scale %.3e), %zu NaN\n", md, mx, nn);
  }
#define launch_conv3x3_wino4 launch_conv3x3_wino4s
#endif
  for (int i = 0; i < 3; ++i) if (launch_conv3x3_wino4(a, 0) != PH_OK) return printf("launch failed: %s\n", g_err), 1;
  if (hipDeviceSynchronize() != hipSuccess) return printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())), 1;
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) launch_conv3x3_wino4(a, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
  // determinism: two more launches into cleared buffers must agree bit for bit (correctness against the oracle is tests/'s business)
  hipMemset(dout, 0xff, no * 4); launch_conv3x3_wino4(a, 0); hipMemcpy(o1.data(), dout, no * 4, hipMemcpyDeviceToHost);
  hipMemset(dout, 0xff, no * 4); launch_conv3x3_wino4(a, 0); hipMemcpy(o2.data(), dout, no * 4, hipMemcpyDeviceToHost);
  size_t ndiff = 0, nnan = 0; double sum = 0;
  for (size_t i = 0; i < no; ++i) { ndiff += memcmp(&o1[i], &o2[i], 4) != 0; nnan += o1[i] != o1[i]; sum += o1[i]; }
  printf("B %d %dx%d %d+%d->%d lowres %d: %.4f ms  executed %.1f TFLOP/s (%.3f of 157.3)  direct-equivalent %.1f | determinism: %zu differ, %zu NaN (unwritten), checksum %.6e\n",
         B, H, W, c0, c1, cout, a.src1_lowres, ms, direct / 4 / ms / 1e9, direct / 4 / ms / 1e9 / 157.3, direct / ms / 1e9, ndiff, nnan, sum);
#ifdef W4_STAMP
  {
    std::vector<unsigned long long> pr(256 * 12 * 8);
    hipMemcpy(pr.data(), probe, pr.size() * 8, hipMemcpyDeviceToHost);
    const int Q = (c0 + c1) / 4;
    for (int slot = 0; slot < 4; ++slot) {
      double acc[8] = {0}; int n = 0;
      for (int blk = 0; blk < 256; ++blk) for (int w = slot * 2; w < slot * 2 + 2; ++w) { for (int i = 0; i < 8; ++i) acc[i] += (double)pr[((size_t)blk * 12 + w) * 8 + i]; ++n; }
      double tot = 0; for (int i = 0; i < 8; ++i) tot += acc[i];
      printf("   waves %d-%d (first tile of each workgroup, cycles per quarter):", slot * 2, slot * 2 + 1);
      for (int i = 0; i < 8; ++i) printf(" s%d %.0f", i, acc[i] / n / Q);  // (segments between P4_ST(i) and the next stamp; -DW4_STTILE: the tile's phases, see the kernel)
      printf("  | total %.0f\n", tot / n / Q);
    }
  }
#endif
#ifdef W4_CLOCK
  {
    std::vector<unsigned long long> pr(512);
    hipMemcpy(pr.data(), probe, pr.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int blk = 0; blk < 256; ++blk) { cyc += (double)pr[blk * 2]; rt += (double)pr[blk * 2 + 1]; }
    const double quarters = (double)((c0 + c1) / 4) * (((H + 15) / 16) * ((W + 31) / 32) * B * ntiles) / 256.0;
    printf("   in-kernel: %.0f shader cycles per workgroup (%.0f per quarter incl. prologue and epilogue), clock %.3f GHz\n", cyc / 256, cyc / 256 / quarters, cyc / rt * 0.1);
  }
#endif
  return 0;
}
