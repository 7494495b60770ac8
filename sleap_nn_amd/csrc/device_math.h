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
// Two values at a time on the packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two results per lane and issue slot): the same operations in the same
// order as gelu_f -- same bits -- at ~9 instead of ~19 issue slots per value (the reciprocal and the exponential stay one per value).
typedef float ph_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ph_f32x2 gelu_f2(ph_f32x2 x) {
  const ph_f32x2 y = x * ph_f32x2{0.70710678118654752440f, 0.70710678118654752440f};
  const ph_f32x2 ay = __builtin_elementwise_abs(y);
  const ph_f32x2 d = __builtin_elementwise_fma(ph_f32x2{0.3275911f, 0.3275911f}, ay, ph_f32x2{1.0f, 1.0f});
  const ph_f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  ph_f32x2 p = __builtin_elementwise_fma(ph_f32x2{1.061405429f, 1.061405429f}, t, ph_f32x2{-1.453152027f, -1.453152027f});
  p = __builtin_elementwise_fma(p, t, ph_f32x2{1.421413741f, 1.421413741f});
  p = __builtin_elementwise_fma(p, t, ph_f32x2{-0.284496736f, -0.284496736f});
  p = __builtin_elementwise_fma(p, t, ph_f32x2{0.254829592f, 0.254829592f});
  const ph_f32x2 q = (ph_f32x2{-1.4426950408889634f, -1.4426950408889634f} * y) * y;
  const ph_f32x2 e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
  const ph_f32x2 r = __builtin_elementwise_fma(-p * t, e, ph_f32x2{1.0f, 1.0f});
  const ph_f32x2 erf = {copysignf(r[0], y[0]), copysignf(r[1], y[1])};
  return (ph_f32x2{0.5f, 0.5f} * x) * (ph_f32x2{1.0f, 1.0f} + erf);
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f));
  const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);  // exp(-x^2 / 2) / sqrt(2 pi)
  return fmaf(x, pdf, cdf);
}

}  // namespace ph
