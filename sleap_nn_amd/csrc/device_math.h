// Device math shared by the forward and backward ConvNeXt kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace ph {

// erf as ATen's vectorised CPU kernels compute it (Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7):
// branch-free, and the same approximation the reference's GELU runs through on the CPU.  The
// reciprocal and the exponential use the hardware v_rcp_f32 / v_exp_f32 (1 ulp; the exponent's
// argument rounding adds <= 3e-7 absolute to erf for |x| <= 3 and less beyond): 14 instructions
// instead of ~45 with IEEE division and libm expf, which matters in a GEMM epilogue.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x);
  const float r = fmaf(-p * t, e, 1.0f);
  return copysignf(r, x);
}

// GELU (erf form, nn.GELU default) and its derivative  d/dx [x * Phi(x)] = Phi(x) + x * phi(x).
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f));
  const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);  // exp(-x^2 / 2) / sqrt(2 pi)
  return fmaf(x, pdf, cdf);
}

}  // namespace ph
