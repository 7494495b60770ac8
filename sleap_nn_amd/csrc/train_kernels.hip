// Training-path kernels for gfx950: MSE(+OHKM) loss and its gradient, backward of the encoder-
// decoder ops (ReLU mask, 3x3 conv weight gradient on MFMA, bias gradient, 2x2 max-pool and
// bilinear-x2 backward, 1x1 head backward, first-conv weight gradient), Adam, and the gather that
// re-packs the canonical (state_dict-ordered) parameter arena into the kernels' weight layouts.
// Data gradients of the 3x3 convolutions reuse the forward MFMA kernels with flipped,
// in/out-swapped weights (net_kernels.hip).
//
// Reference semantics (paths relative to talmolab/sleap-nn):
//   nn.MSELoss per head, total = sum_h loss_weight_h * loss_h ... training/lightning_modules.py:526,1850-1895
//   compute_ohkm_loss ........................................ training/losses.py:8-63
//   autograd of Conv2d / ReLU / max_pool2d / Upsample(bilinear) = ATen backward semantics
//   torch.optim.Adam (lr, betas (0.9, 0.999), eps 1e-8, amsgrad optional) lightning_modules.py:752-763
//
// All reductions are two-stage with a fixed order (no float atomics): results are bitwise
// reproducible from run to run.
#include <algorithm>

#include "common.h"
#include "train_kernels.h"

namespace ph {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float block_sum_256(float v, float* red /* >= 4 floats */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  const float t = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return t;
}

// ---------------------------------------------------------------------------------------
// Loss: per-channel sum of squared errors (NCHW), two stage.
// ---------------------------------------------------------------------------------------
constexpr int SSE_SLICES = 64;

// sample_w (B floats or nullptr): per-sample weights of the MSE term (negative-sample weighting,
// lightning_modules.py:526-545); the weighted sums go to partial_w, the plain ones (OHKM ranks channels by them) to partial.
__global__ __launch_bounds__(256) void sse_partial_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, const float* __restrict__ sample_w, int B, int C,
                                                          int HW, float* __restrict__ partial /* C x SSE_SLICES */, float* __restrict__ partial_w /* same, or unused */) {
  __shared__ float red[4];
  const int c = blockIdx.x, sl = blockIdx.y;
  const size_t n = (size_t)B * HW;
  const size_t per = (n + SSE_SLICES - 1) / SSE_SLICES;
  const size_t lo = (size_t)sl * per, hi = std::min(n, lo + per);
  float acc = 0.f, acc_w = 0.f;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
    const size_t b = i / HW, hw = i - b * HW;
    const size_t o = (b * C + c) * HW + hw;
    const float d = pred[o] - tgt[o];
    acc += d * d;
    if (sample_w) acc_w += sample_w[b] * (d * d);
  }
  const float t = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[c * SSE_SLICES + sl] = t;
  if (sample_w) {  // workgroup-uniform
    const float tw = block_sum_256(acc_w, red);
    if (threadIdx.x == 0) partial_w[c * SSE_SLICES + sl] = tw;
  }
}

// One thread: per-channel SSE -> head loss (MSE + optional OHKM) and the gradient coefficients: the MSE term's
// coeff[C] = loss_weight * 2/N (multiplied by the sample's weight in mse_grad_kernel) and the per-channel OHKM extra
// coeff[c] = loss_weight * [c is hard] * 2*loss_scale/n_elements.
__global__ void loss_coeff_kernel(const float* __restrict__ partial, const float* __restrict__ partial_w /* weighted sums or nullptr */, int C,
                                  double n_total /* B*C*H*W */, double n_per_channel /* B*H*W */,
                                  float loss_weight, int ohkm, float hard_to_easy_ratio, int min_hard, int max_hard, float loss_scale,
                                  float* __restrict__ coeff /* C + 1 */, float* __restrict__ loss_out /* [0] = this head's loss */) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float sse[256];
  float total = 0.f;
  for (int c = 0; c < C; ++c) {
    float s = 0.f, sw = 0.f;
    for (int k = 0; k < SSE_SLICES; ++k) {
      s += partial[c * SSE_SLICES + k];
      if (partial_w) sw += partial_w[c * SSE_SLICES + k];
    }
    sse[c] = s;
    total += partial_w ? sw : s;
  }
  float loss = (float)((double)total / n_total);
  const float base = (float)(2.0 / n_total);
  coeff[C] = loss_weight * base;
  for (int c = 0; c < C; ++c) coeff[c] = 0.f;
  if (ohkm) {
    float best = sse[0];
    for (int c = 1; c < C; ++c) best = fminf(best, sse[c]);
    int n_hard = 0;
    for (int c = 0; c < C; ++c) n_hard += ((sse[c] / best) >= hard_to_easy_ratio) ? 1 : 0;
    int mx = max_hard < 0 ? C : min(max_hard, C);
    const int k = min(max(n_hard, min_hard), mx);
    // top-k channels by SSE (ties: lowest channel index first)
    bool used[256];
    for (int c = 0; c < C; ++c) used[c] = false;
    float ksum = 0.f;
    const double n_el = n_per_channel * (double)k;
    for (int j = 0; j < k; ++j) {
      int bi = -1;
      for (int c = 0; c < C; ++c)
        if (!used[c] && (bi < 0 || sse[c] > sse[bi])) bi = c;
      used[bi] = true;
      ksum += sse[bi] * loss_scale;
      coeff[bi] += loss_weight * (float)(2.0 * (double)loss_scale / n_el);
    }
    loss += (float)((double)ksum / n_el);
  }
  loss_out[0] = loss;
}

struct LossWeights {
  float w[PH_MAX_OUTPUTS];
};
// the few loss weights travel by value as a kernel argument: no host->device copy on the step's stream (capturable)
__global__ void total_loss_kernel(const float* __restrict__ head_loss, LossWeights w, int n, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < n; ++i) t += w.w[i] * head_loss[i];
    out[0] = t;
  }
}

__global__ __launch_bounds__(256) void mse_grad_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, const float* __restrict__ coeff,
                                                       const float* __restrict__ sample_w, int C, int HW, size_t n, float* __restrict__ dy) {
  const float base = coeff[C];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t bc = i / HW;
    const int c = (int)(bc % C);
    const float k = (sample_w ? base * sample_w[bc / C] : base) + coeff[c];
    dy[i] = k * (pred[i] - tgt[i]);
  }
}

int launch_loss(const float* pred, const float* tgt, const float* sample_w, int B, int C, int H, int W, float loss_weight, const OhkmParams& ok,
                float* scratch /* loss_scratch_floats(C) */, float* dy, float* loss_out, hipStream_t s) {
  if (C > 256) {
    set_error("loss: more than 256 channels");
    return PH_E_INVALID;
  }
  float* partial = scratch;
  float* partial_w = scratch + (size_t)C * SSE_SLICES;
  float* coeff = scratch + (size_t)2 * C * SSE_SLICES;
  const int HW = H * W;
  hipLaunchKernelGGL(sse_partial_kernel, dim3(C, SSE_SLICES), dim3(256), 0, s, pred, tgt, sample_w, B, C, HW, partial, partial_w);
  hipLaunchKernelGGL(loss_coeff_kernel, dim3(1), dim3(1), 0, s, partial, sample_w ? partial_w : nullptr, C, (double)B * C * HW, (double)B * HW, loss_weight, ok.enabled,
                     ok.hard_to_easy_ratio, ok.min_hard, ok.max_hard, ok.loss_scale, coeff, loss_out);
  const size_t n = (size_t)B * C * HW;
  hipLaunchKernelGGL(mse_grad_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 16384)), dim3(256), 0, s, pred, tgt, coeff, sample_w, C, HW, n, dy);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

int64_t loss_scratch_floats(int C) { return (int64_t)2 * C * SSE_SLICES + C + 1; }

int launch_total_loss(const float* head_loss, const float* w_host, int n, float* out, hipStream_t s) {
  if (n > PH_MAX_OUTPUTS) {
    set_error("loss: more than %d heads", PH_MAX_OUTPUTS);
    return PH_E_INVALID;
  }
  LossWeights w{};
  for (int i = 0; i < n; ++i) w.w[i] = w_host[i];
  hipLaunchKernelGGL(total_loss_kernel, dim3(1), dim3(1), 0, s, head_loss, w, n, out);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

constexpr int BIAS_SLICES = 1024;  // rows of the partial table of a bias gradient (bias_partial_kernel / pool_bwd_kernel / head_bwd_dx4_kernel -> bias_final_kernel)
__global__ __launch_bounds__(256) void bias_final_kernel(const float* __restrict__ partial, int cp, int cout, float* __restrict__ gb);

// ---------------------------------------------------------------------------------------
// 1x1 head backward.  dY: NCHW (B, Cout, HW); X: NHWC (cp); W: [Cout][cp].
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w, int cout, int cp, int HW, size_t npix,
                                                          const float* __restrict__ y_out /* NCHW post-activation, or nullptr */, int sigmoid, int accumulate,
                                                          float* __restrict__ dx) {
  // thread = (pixel, 4-channel group)
  const int groups = cp >> 2;
  const size_t total = npix * groups;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int g = (int)(idx % groups);
    const size_t p = idx / groups;
    const size_t b = p / HW, hw = p - b * HW;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int co = 0; co < cout; ++co) {
      float d = dy[(b * cout + co) * HW + hw];
      if (sigmoid) {
        const float yv = y_out[(b * cout + co) * HW + hw];
        d *= yv * (1.f - yv);
      }
      acc += d * *reinterpret_cast<const f32x4*>(w + (size_t)co * cp + g * 4);
    }
    f32x4* o = reinterpret_cast<f32x4*>(dx + p * cp + g * 4);
    if (accumulate) acc += *o;
    *o = acc;
  }
}

// The same for FOUR consecutive pixels per thread (HW % 4 == 0: a quad never straddles two images): one 16-byte load of dY per output channel serves four
// pixels and the weight quad is loaded once per four outputs -- 6.5 instead of 26 load instructions per 16 bytes stored (the one-pixel form is bound by
// load issue: 0.47 ms for a 537-MB gradient).  x_mask != nullptr: x is the output of a conv + ReLU and this launch completes its gradient -- the mask
// (x > 0) is applied to the finished value as it is stored; bias_partial != nullptr (needs x_mask, 256 % (cp / 4) == 0, BIAS_SLICES blocks): the stored
// values are summed per channel into row blockIdx.x of the partial table (the producer conv's bias gradient), as pool_bwd_kernel does.
__global__ __launch_bounds__(256) void head_bwd_dx4_kernel(const float* __restrict__ dy, const float* __restrict__ w, int cout, int cp, int HW, size_t npix,
                                                           const float* __restrict__ y_out, int sigmoid, int accumulate, const float* __restrict__ x_mask,
                                                           float* __restrict__ dx, float* __restrict__ bias_partial);

// dY (NCHW, optionally times sigmoid'(y)) -> rows [pixel][cop] with zero pad channels, the operand layout of the
// row weight-gradient GEMM and of the bias reduction.  Tile transpose through LDS: 64 pixels x cop channels.
__global__ __launch_bounds__(256) void head_dy_rows_kernel(const float* __restrict__ dy, const float* __restrict__ y_out, int sigmoid, int cout, int cop, int HW,
                                                           size_t npix, float* __restrict__ rows) {
  extern __shared__ float tile[];  // [cop][65]
  const size_t p0 = (size_t)blockIdx.x * 64;
  for (int i = threadIdx.x; i < cop * 64; i += 256) {
    const int co = i >> 6, pp = i & 63;
    const size_t p = p0 + pp;
    float d = 0.f;
    if (co < cout && p < npix) {
      const size_t b = p / HW, hw = p - b * HW;
      d = dy[(b * cout + co) * HW + hw];
      if (sigmoid) {
        const float yv = y_out[(b * cout + co) * HW + hw];
        d *= yv * (1.f - yv);
      }
    }
    tile[co * 65 + pp] = d;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < cop * 64; i += 256) {
    const int pp = i / cop, co = i - pp * cop;
    if (p0 + pp < npix) rows[(p0 + pp) * cop + co] = tile[co * 65 + pp];
  }
}

__global__ __launch_bounds__(256) void head_bwd_dx4_kernel(const float* __restrict__ dy, const float* __restrict__ w, int cout, int cp, int HW, size_t npix,
                                                           const float* __restrict__ y_out, int sigmoid, int accumulate, const float* __restrict__ x_mask,
                                                           float* __restrict__ dx, float* __restrict__ bias_partial) {
  __shared__ f32x4 red[256];
  const int groups = cp >> 2;
  const size_t total = (npix >> 2) * groups;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int g = (int)(idx % groups);
    const size_t p = (idx / groups) << 2;
    const size_t b = p / HW, hw = p - b * HW;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int co = 0; co < cout; ++co) {
      f32x4 d = *reinterpret_cast<const f32x4*>(dy + (b * cout + co) * HW + hw);
      if (sigmoid) {
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y_out + (b * cout + co) * HW + hw);
        d *= yv * (1.f - yv);
      }
      const f32x4 wq = *reinterpret_cast<const f32x4*>(w + (size_t)co * cp + g * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += d[j] * wq;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t o = (p + j) * cp + g * 4;
      f32x4 v = acc[j];
      if (accumulate) v += *reinterpret_cast<const f32x4*>(dx + o);
      if (x_mask) {
        const f32x4 f = *reinterpret_cast<const f32x4*>(x_mask + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = f[e] > 0.f ? v[e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(dx + o) = v;
      bsum += v;
    }
  }
  if (bias_partial) {  // kernel-uniform
    red[threadIdx.x] = bsum;
    __syncthreads();
    if ((int)threadIdx.x < groups) {
      f32x4 t = red[threadIdx.x];
      for (int k = groups; k < 256; k += groups) t += red[k + threadIdx.x];  // fixed order
      *reinterpret_cast<f32x4*>(bias_partial + (size_t)blockIdx.x * cp + 4 * threadIdx.x) = t;
    }
  }
}

// the mask / bias-gradient fusion of head_bwd_dx4_kernel applies to this shape
bool head_bwd_can_fold(int cp, int HW) { return (HW & 3) == 0 && cp >= 4 && 256 % (cp / 4) == 0; }

// Head weight / bias gradients through the row weight-gradient GEMM (MFMA, K = pixels) and the column-sum reduction.
// x_mask (with head_bwd_can_fold): the ReLU mask of the conv that produced x, applied to the finished dx; conv_gb (needs x_mask): that conv's bias gradient
// (conv_cout channels), summed from the stored values.
int launch_head_bwd(const float* dy, const float* y_out, int sigmoid, const float* x, const float* w_packed, int B, int HW, int cin, int cp, int cout,
                    int accumulate, float* dx, float* gw, float* gb, float* scratch, hipStream_t s, const float* x_mask, float* conv_gb, int conv_cout) {
  const size_t npix = (size_t)B * HW;
  const size_t total = npix * (cp / 4);
  PH_REQUIRE(!x_mask || head_bwd_can_fold(cp, HW), "head backward: the folded ReLU mask needs HW %% 4 == 0 and 256 %% (cp / 4) == 0");
  PH_REQUIRE(!conv_gb || x_mask, "head backward: the fused bias gradient needs the folded mask");
  if ((HW & 3) == 0) {
    if (conv_gb) {
      hipLaunchKernelGGL(head_bwd_dx4_kernel, dim3(BIAS_SLICES), dim3(256), 0, s, dy, w_packed, cout, cp, HW, npix, y_out, sigmoid, accumulate, x_mask, dx, scratch);
      hipLaunchKernelGGL(bias_final_kernel, dim3((conv_cout + 3) / 4), dim3(256), 0, s, scratch, cp, conv_cout, conv_gb);
    } else {
      hipLaunchKernelGGL(head_bwd_dx4_kernel, dim3((unsigned)std::min<size_t>((total / 4 + 255) / 256, 16384)), dim3(256), 0, s, dy, w_packed, cout, cp, HW, npix, y_out,
                         sigmoid, accumulate, x_mask, dx, static_cast<float*>(nullptr));
    }
  } else {
    hipLaunchKernelGGL(head_bwd_dx_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 16384)), dim3(256), 0, s, dy, w_packed, cout, cp, HW, npix, y_out,
                       sigmoid, accumulate, dx);
  }
  const int cop = pad16(cout);
  float* rows = scratch;
  float* rest = scratch + align_up((int64_t)npix * cop, 64);
  hipLaunchKernelGGL(head_dy_rows_kernel, dim3((unsigned)((npix + 63) / 64)), dim3(256), (size_t)cop * 65 * sizeof(float), s, dy, y_out, sigmoid, cout, cop, HW, npix, rows);
  PH_HIP_CHECK(hipGetLastError());
  RowWgradArgs a{};
  a.dy = rows;
  a.x = x;
  a.slab = rest;
  a.np = cop;
  a.kp = cp;
  a.M = (int)npix;
  int rc = launch_row_wgrad(a, cout, cin, 1, gw, s);  // canonical (cout, cin, 1, 1)
  if (rc != PH_OK) return rc;
  if (gb) rc = launch_bias_grad(rows, npix, cop, cout, gb, rest, s);
  return rc;
}

int64_t head_bwd_scratch_floats(int cp, int cout, int64_t npix) {
  const int cop = pad16(cout);
  return align_up(npix * cop, 64) + std::max<int64_t>(row_wgrad_slab_floats((int)npix, cout, cp), bias_scratch_floats(cop));
}

// ---------------------------------------------------------------------------------------
// Elementwise backward pieces (NHWC, float4)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relu_mask_kernel(float* __restrict__ g, const float* __restrict__ y, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 gv = reinterpret_cast<f32x4*>(g)[i];
    const f32x4 yv = reinterpret_cast<const f32x4*>(y)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) gv[k] = yv[k] > 0.f ? gv[k] : 0.f;
    reinterpret_cast<f32x4*>(g)[i] = gv;
  }
}

int launch_relu_mask(float* g, const float* y, size_t n, hipStream_t s) {
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 16384)), dim3(256), 0, s, g, y, n4);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// max_pool2d(2,2) backward: the gradient goes to the FIRST maximum of each window in row-major
// order (ATen); windows reaching past an odd edge see the zero pad last, so a real element wins ties.
// relu_mask: x is the output of a conv + ReLU and this is the LAST contribution to its gradient: the ReLU mask (x > 0) is applied to the
// finished sum here, on values this kernel reads anyway, instead of in a separate read-modify-write pass over the gradient tensor.
// bias_partial != nullptr (needs relu_mask, 256 % (cp / 4) == 0 and a grid of BIAS_SLICES blocks): the finished, masked gradient is also summed per
// channel -- the bias gradient of the conv that produced x -- into row blockIdx.x of the BIAS_SLICES x cp partial table that bias_final_kernel adds
// up; a thread keeps one channel quad for its whole walk (its stride is a multiple of cp / 4), so its sum stays in four registers.
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ gp, const float* __restrict__ x, int B, int H, int W, int cp, int accumulate,
                                                       int relu_mask, float* __restrict__ gx, float* __restrict__ bias_partial) {
  __shared__ f32x4 red[256];
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, groups = cp >> 2;
  const size_t total = (size_t)B * Ho * Wo * groups;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int gq = (int)(idx % groups);
    size_t p = idx / groups;
    const int xo = (int)(p % Wo);
    p /= Wo;
    const int yo = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gp + idx * 4);
    f32x4 v[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = 2 * yo + (k >> 1), xx = 2 * xo + (k & 1);
      ok[k] = yy < H && xx < W;
      v[k] = ok[k] ? *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + yy) * W + xx) * cp + gq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // first max among the in-image elements; the zero pad (if any) can only win when all real elements are < 0
      int best = -1;
      float bv = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (ok[k] && (best < 0 || v[k][e] > bv)) {
          best = k;
          bv = v[k][e];
        }
      const bool pad_wins = (!(ok[1] && ok[2] && ok[3])) && (bv < 0.f);
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k][e] = (k == best && !pad_wins) ? g[e] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (ok[k]) {
        const int yy = 2 * yo + (k >> 1), xx = 2 * xo + (k & 1);
        f32x4* dst = reinterpret_cast<f32x4*>(gx + (((size_t)b * H + yy) * W + xx) * cp + gq * 4);
        f32x4 r = accumulate ? (*dst + o[k]) : o[k];
        if (relu_mask) {
#pragma unroll
          for (int e = 0; e < 4; ++e) r[e] = v[k][e] > 0.f ? r[e] : 0.f;
        }
        *dst = r;
        bsum += r;
      }
  }
  if (bias_partial) {  // kernel-uniform
    red[threadIdx.x] = bsum;
    __syncthreads();
    if ((int)threadIdx.x < groups) {
      f32x4 t = red[threadIdx.x];
      for (int k = groups; k < 256; k += groups) t += red[k + threadIdx.x];  // fixed order
      *reinterpret_cast<f32x4*>(bias_partial + (size_t)blockIdx.x * cp + 4 * threadIdx.x) = t;
    }
  }
}

// the bias-gradient fusion of pool_bwd_kernel applies to this shape
bool pool_bwd_can_sum_bias(int cp) { return cp >= 4 && 256 % (cp / 4) == 0; }

// gb != nullptr (with relu_mask, pool_bwd_can_sum_bias(cp)): also the bias gradient of the conv whose output x is (cout channels), via scratch (>= bias_scratch_floats(cp))
int launch_pool_bwd(const float* gp, const float* x, int B, int H, int W, int cp, int accumulate, int relu_mask, float* gx, float* gb, int cout, float* scratch, hipStream_t s) {
  const size_t total = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (cp / 4);
  if (gb) {
    PH_REQUIRE(relu_mask && pool_bwd_can_sum_bias(cp) && scratch, "pool_bwd: the fused bias gradient needs the ReLU-mask form and 256 %% (cp / 4) == 0");
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(BIAS_SLICES), dim3(256), 0, s, gp, x, B, H, W, cp, accumulate, relu_mask, gx, scratch);
    hipLaunchKernelGGL(bias_final_kernel, dim3((cout + 3) / 4), dim3(256), 0, s, scratch, cp, cout, gb);
  } else {
    hipLaunchKernelGGL(pool_bwd_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 16384)), dim3(256), 0, s, gp, x, B, H, W, cp, accumulate, relu_mask, gx,
                       static_cast<float*>(nullptr));
  }
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// bilinear x2 (align_corners=False) backward, gather form: input i receives from outputs
// 2i-1 (w 0.25), 2i (0.75), 2i+1 (0.75), 2i+2 (0.25); at the borders the clamped taps fold onto
// the edge input (output 0 and output 2n-1 give weight 1.0 to inputs 0 and n-1).
__device__ __forceinline__ void up_taps(int i, int n, int* o, float* w) {
  // outputs 2i-1..2i+2; weight of input i in output o: forward weights
  o[0] = 2 * i - 1;
  o[1] = 2 * i;
  o[2] = 2 * i + 1;
  o[3] = 2 * i + 2;
  w[0] = 0.25f;
  w[1] = 0.75f;
  w[2] = 0.75f;
  w[3] = 0.25f;
  if (i == 0) {
    w[0] = 0.f;   // no output -1
    w[1] = 1.0f;  // output 0: src clamps to 0 -> all weight on input 0
  }
  if (i == n - 1) {
    w[3] = 0.f;   // no output 2n
    w[2] = 1.0f;  // output 2n-1: i1 clamps to n-1 -> 0.75 + 0.25
  }
}

__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ gy, int B, int H, int W, int cp, int accumulate, float* __restrict__ gx) {
  const int groups = cp >> 2, Ho = 2 * H, Wo = 2 * W;
  const size_t total = (size_t)B * H * W * groups;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int gq = (int)(idx % groups);
    size_t p = idx / groups;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const int b = (int)(p / H);
    int oy[4], ox[4];
    float wy[4], wx[4];
    up_taps(y, H, oy, wy);
    up_taps(x, W, ox, wx);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (wy[j] == 0.f || oy[j] < 0 || oy[j] >= Ho) continue;
      f32x4 row = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (wx[i] == 0.f || ox[i] < 0 || ox[i] >= Wo) continue;
        row += wx[i] * *reinterpret_cast<const f32x4*>(gy + (((size_t)b * Ho + oy[j]) * Wo + ox[i]) * cp + gq * 4);
      }
      acc += wy[j] * row;
    }
    f32x4* dst = reinterpret_cast<f32x4*>(gx + idx * 4);
    *dst = accumulate ? (*dst + acc) : acc;
  }
}

int launch_upsample_bwd(const float* gy, int B, int H, int W, int cp, int accumulate, float* gx, hipStream_t s) {
  const size_t total = (size_t)B * H * W * (cp / 4);
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 16384)), dim3(256), 0, s, gy, B, H, W, cp, accumulate, gx);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// Bias gradient: column sums of an NHWC tensor, two stage.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bias_partial_kernel(float* __restrict__ g, const float* __restrict__ y_mask, size_t npix, int cp,
                                                           float* __restrict__ partial /* BIAS_SLICES x cp */) {
  // y_mask != nullptr: the ReLU mask (g = 0 where the forward output y is 0) is applied on the way, in place.
  // thread = (pixel row r, channel quad c): 256/min(cp/4,256) pixels are summed side by side, four pixels
  // per trip with all loads issued before the first store (the pass is HBM-bound, not latency-bound);
  // the rows are combined in a fixed order through LDS (deterministic).
  __shared__ f32x4 red[256];
  const int sl = blockIdx.x;
  const size_t per = (npix + BIAS_SLICES - 1) / BIAS_SLICES;
  const size_t lo = (size_t)sl * per, hi = std::min(npix, lo + per);
  const int cq = cp >> 2;  // cp is a multiple of 16
  const int cw = min(cq, 256), rows = 256 / cw;
  const int c0 = threadIdx.x % cw, r = threadIdx.x / cw;
  f32x4* g4 = reinterpret_cast<f32x4*>(g);
  const f32x4* y4 = reinterpret_cast<const f32x4*>(y_mask);
  auto masked = [](f32x4 v, f32x4 y) __attribute__((always_inline)) {
    return f32x4{y[0] > 0.f ? v[0] : 0.f, y[1] > 0.f ? v[1] : 0.f, y[2] > 0.f ? v[2] : 0.f, y[3] > 0.f ? v[3] : 0.f};
  };
  for (int cb = 0; cb < cq; cb += cw) {
    const int c = cb + c0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (r < rows && c < cq) {
      size_t p = lo + r;
      for (; p + 3 * (size_t)rows < hi; p += 4 * (size_t)rows) {
        const size_t i0 = p * cq + c, st = (size_t)rows * cq;
        f32x4 v0 = g4[i0], v1 = g4[i0 + st], v2 = g4[i0 + 2 * st], v3 = g4[i0 + 3 * st];
        if (y_mask) {
          const f32x4 m0 = y4[i0], m1 = y4[i0 + st], m2 = y4[i0 + 2 * st], m3 = y4[i0 + 3 * st];
          v0 = masked(v0, m0), v1 = masked(v1, m1), v2 = masked(v2, m2), v3 = masked(v3, m3);
          g4[i0] = v0, g4[i0 + st] = v1, g4[i0 + 2 * st] = v2, g4[i0 + 3 * st] = v3;
        }
        acc += (v0 + v1) + (v2 + v3);
      }
      for (; p < hi; p += rows) {
        f32x4 v = g4[p * cq + c];
        if (y_mask) {
          v = masked(v, y4[p * cq + c]);
          g4[p * cq + c] = v;
        }
        acc += v;
      }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (r == 0 && c < cq) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < rows; ++k) s += red[k * cw + c0];
      *reinterpret_cast<f32x4*>(partial + (size_t)sl * cp + 4 * c) = s;
    }
    __syncthreads();
  }
}
// one wave per channel: lane l adds slices l, l + 64, ... and the 64 partial sums are combined by a fixed butterfly
__global__ __launch_bounds__(256) void bias_final_kernel(const float* __restrict__ partial, int cp, int cout, float* __restrict__ gb) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  float s = 0.f;
  if (c < cout)
    for (int k = lane; k < BIAS_SLICES; k += 64) s += partial[(size_t)k * cp + c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (c < cout && lane == 0) gb[c] = s;
}
int launch_bias_grad(const float* g, size_t npix, int cp, int cout, float* gb, float* scratch, hipStream_t s) {
  return launch_relu_mask_bias_grad(const_cast<float*>(g), nullptr, npix, cp, cout, gb, scratch, s);
}
// ReLU mask (in place, y_mask = forward output) + bias gradient in one pass over the gradient tensor
int launch_relu_mask_bias_grad(float* g, const float* y_mask, size_t npix, int cp, int cout, float* gb, float* scratch, hipStream_t s) {
  hipLaunchKernelGGL(bias_partial_kernel, dim3(BIAS_SLICES), dim3(256), 0, s, g, y_mask, npix, cp, scratch);
  hipLaunchKernelGGL(bias_final_kernel, dim3((cout + 3) / 4), dim3(256), 0, s, scratch, cp, cout, gb);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}
int64_t bias_scratch_floats(int cp) { return (int64_t)BIAS_SLICES * cp; }

// ---------------------------------------------------------------------------------------
// 3x3 conv weight gradient on MFMA.
//   dW[tap][ci][co] = sum_pixels X[p + tap][ci] * dY[p][co]  ->  GEMM with M = ci, N = co, K = pixels.
//   Workgroup (256 thr) owns a 32(ci) x 32(co) tile for all 9 taps over a K slice of pixel
//   tiles (4 x TW pixels each).  A[i = ci][k = pixel pair], B[k][j = co]: both fragments are
//   single dwords read from LDS images [pixel][32 channels] (lanes run over the channel: no bank
//   conflicts).  Waves split the taps {0,4,8} {1,5} {2,6} {3,7}.  Partial sums go to a slab per
//   K slice; wgrad_reduce_kernel adds the slabs in a fixed order into the canonical OIHW gradient.
// ---------------------------------------------------------------------------------------
constexpr int WG_TW = 32, WG_HW = WG_TW + 2;  // wgrad16_kernel: 8 x 32 pixel tiles
constexpr int W16_TH = 8, W16_HH = W16_TH + 2;

template <int TW>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradArgs a) {
  // Every wave accumulates all nine taps over its row of each 4 x TW pixel tile (TW in {32, 24, 16, 12}, picked
  // per layer so that narrow feature maps are not mostly tile padding): nine accumulators
  // per wave, perfectly balanced (a split of the taps over four waves is 3+2+2+2); the four partial sets
  // are added through LDS once, after the K loop.  The next tile's X halo and dY travel global -> registers
  // while this tile's MFMAs run (fetch / commit below): the global latency never sits between two tiles.
  constexpr int TH = 4, HH = TH + 2, HW = TW + 2;
  constexpr int NXQ = HH * HW * 8, NYQ = TH * TW * 8;  // float4 quads per tile image
  constexpr int NX = (NXQ + 255) / 256, NY = (NYQ + 255) / 256;
  __shared__ float sX[(NXQ * 4 > 4096 ? NXQ * 4 : 4096)];
  __shared__ float sY[NYQ * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, lh = lane >> 5;
  const int n_co_t = (a.coutp + 31) / 32;
  const int ci_t = blockIdx.x / n_co_t, co_t = blockIdx.x - ci_t * n_co_t;
  const int slice = blockIdx.y, n_slices = gridDim.y;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  const int n_tiles = tiles_x * tiles_y * a.B;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  f32x4 rx[NX], ry[NY];
  unsigned okx = 0, oky = 0;
  const int cxq = min(ci_t * 32 + (tid & 7) * 4, a.cxp - 4), cyq = min(co_t * 32 + (tid & 7) * 4, a.coutp - 4);
  const bool cx_ok = ci_t * 32 + (tid & 7) * 4 < a.cxp, cy_ok = co_t * 32 + (tid & 7) * 4 < a.coutp;
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    int t = tile;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    okx = 0, oky = 0;
#pragma unroll
    for (int j = 0; j < NX; ++j) {  // X halo [pix][32 ci]; outside the image / beyond the channels -> 0 at commit
      const int pix = min((tid + 256 * j) >> 3, HH * HW - 1);
      const int hy = pix / HW, hx = pix - hy * HW;
      const int gy = y0 + hy - 1, gx = x0 + hx - 1;
      const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && cx_ok;
      const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
      rx[j] = *reinterpret_cast<const f32x4*>(a.x + ((size_t)(b * a.H + cy) * a.W + cx) * a.cxp + cxq);
      okx |= ok ? (1u << j) : 0u;
    }
#pragma unroll
    for (int j = 0; j < NY; ++j) {  // dY [pix][32 co]
      const int pix = min((tid + 256 * j) >> 3, TH * TW - 1);
      const int py = pix / TW, px = pix - py * TW;
      const int gy = y0 + py, gx = x0 + px;
      const bool ok = gy < a.H && gx < a.W && cy_ok;
      const int cy = min(gy, a.H - 1), cx = min(gx, a.W - 1);
      ry[j] = *reinterpret_cast<const f32x4*>(a.dy + ((size_t)(b * a.H + cy) * a.W + cx) * a.coutp + cyq);
      oky |= ok ? (1u << j) : 0u;
    }
  };
  auto commit = [&]() __attribute__((always_inline)) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NX; ++j)
      if (tid + 256 * j < NXQ) *reinterpret_cast<f32x4*>(sX + (tid + 256 * j) * 4) = ((okx >> j) & 1u) ? rx[j] : z;
#pragma unroll
    for (int j = 0; j < NY; ++j)
      if (tid + 256 * j < NYQ) *reinterpret_cast<f32x4*>(sY + (tid + 256 * j) * 4) = ((oky >> j) & 1u) ? ry[j] : z;
  };

  if (slice < n_tiles) fetch(slice);
  for (int tile = slice; tile < n_tiles; tile += n_slices) {
    commit();
    __syncthreads();
    if (tile + n_slices < n_tiles) fetch(tile + n_slices);
#pragma unroll 2
    for (int s2 = 0; s2 < TW / 2; ++s2) {  // this wave's row of the tile, two pixels per MFMA
      const int py = wave, px = 2 * s2 + lh;
      const float bv = sY[(py * TW + px) * 32 + lx];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const float av = sX[((py + ky) * HW + px + kx) * 32 + lx];
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tap], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // add the four waves' partial sets through LDS (sX is free now: 4 waves x 1024 floats per tap fit),
  // fixed order, then slab[slice][block][tap][ci 32][co 32];  D: row(ci) = (r&3) + 8*(r>>2) + 4*lh, col(co) = lx
  float* slab = a.slab + (((size_t)slice * gridDim.x + blockIdx.x) * 9) * 1024;
  for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sX[wave * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + lx] = acc[tap][r];
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) slab[(size_t)tap * 1024 + i] = (sX[i] + sX[1024 + i]) + (sX[2048 + i] + sX[3072 + i]);
    __syncthreads();
  }
}

// 16-channel layers (the full-resolution first encoder block): the same computation on v_mfma_f32_16x16x4_f32,
// a 16(ci) x 16(co) tile with four pixels per MFMA -- a 32x32 tile would be three quarters padding there.
__global__ __launch_bounds__(256, 2) void wgrad16_kernel(WgradArgs a) {
  constexpr int NXQ = W16_HH * WG_HW * 4, NYQ = W16_TH * WG_TW * 4;  // float4 quads per tile image
  constexpr int NX = (NXQ + 255) / 256, NY = NYQ / 256;
  __shared__ float sX[NXQ * 4];
  __shared__ float sY[NYQ * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int slice = blockIdx.y, n_slices = gridDim.y;
  const int tiles_x = (a.W + WG_TW - 1) / WG_TW, tiles_y = (a.H + W16_TH - 1) / W16_TH;
  const int n_tiles = tiles_x * tiles_y * a.B;
  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // next tile: global -> registers while this tile's MFMAs run (fetch), registers -> LDS afterwards (commit)
  f32x4 rx[NX], ry[NY];
  unsigned okx = 0, oky = 0;
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    int t = tile;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = tx * WG_TW, y0 = ty * W16_TH;
    okx = 0, oky = 0;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int i = min(tid + 256 * j, NXQ - 1);
      const int pix = i >> 2, q = i & 3;
      const int hy = pix / WG_HW, hx = pix - hy * WG_HW;
      const int gy = y0 + hy - 1, gx = x0 + hx - 1;
      const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
      rx[j] = *reinterpret_cast<const f32x4*>(a.x + ((size_t)(b * a.H + cy) * a.W + cx) * 16 + q * 4);
      okx |= ok ? (1u << j) : 0u;
    }
#pragma unroll
    for (int j = 0; j < NY; ++j) {
      const int i = tid + 256 * j;
      const int pix = i >> 2, q = i & 3;
      const int py = pix / WG_TW, px = pix - py * WG_TW;
      const int gy = y0 + py, gx = x0 + px;
      const bool ok = gy < a.H && gx < a.W;
      const int cy = min(gy, a.H - 1), cx = min(gx, a.W - 1);
      ry[j] = *reinterpret_cast<const f32x4*>(a.dy + ((size_t)(b * a.H + cy) * a.W + cx) * 16 + q * 4);
      oky |= ok ? (1u << j) : 0u;
    }
  };
  auto commit = [&]() __attribute__((always_inline)) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NX; ++j)
      if (tid + 256 * j < NXQ) *reinterpret_cast<f32x4*>(sX + (tid + 256 * j) * 4) = ((okx >> j) & 1u) ? rx[j] : z;
#pragma unroll
    for (int j = 0; j < NY; ++j) *reinterpret_cast<f32x4*>(sY + (tid + 256 * j) * 4) = ((oky >> j) & 1u) ? ry[j] : z;
  };
  if (slice < n_tiles) fetch(slice);
  for (int tile = slice; tile < n_tiles; tile += n_slices) {
    commit();
    __syncthreads();
    if (tile + n_slices < n_tiles) fetch(tile + n_slices);
#pragma unroll 2
    for (int s2 = 0; s2 < 16; ++s2) {  // this wave's 64 pixels (rows 2w, 2w+1), four per MFMA
      const int py = 2 * wave + (s2 >> 3), px = 4 * (s2 & 7) + lg;
      const float bv = sY[(py * WG_TW + px) * 16 + li];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const float av = sX[((py + ky) * WG_HW + px + kx) * 16 + li];
        acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[tap], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // D: row(ci) = 4*lg + r, col(co) = li; the slab slot keeps the 32x32 layout of wgrad_reduce_kernel
  float* slab = a.slab + (((size_t)slice * gridDim.x + blockIdx.x) * 9) * 1024;
  for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
    for (int r = 0; r < 4; ++r) sX[wave * 256 + (4 * lg + r) * 16 + li] = acc[tap][r];
    __syncthreads();
    {
      const int ci = tid >> 4, co = tid & 15;
      slab[(size_t)tap * 1024 + ci * 32 + co] = (sX[tid] + sX[256 + tid]) + (sX[512 + tid] + sX[768 + tid]);
    }
    __syncthreads();
  }
}

// grad[(co*cin_total + ci_off + ci)*9 + tap] = sum_slices slab[...].  One workgroup per (tile, tap) slab slot
// (1024 floats = 256 quads); 1024 threads = 256 quads x 4 slice parts, so the reads are whole 4 KiB slots
// and a thread walks only a quarter of the slices, four loads in flight; the parts meet in LDS in a fixed order.
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ slab, int n_slices, int n_blocks, int n_co_t, int cin, int cout, int cin_total,
                                                            int ci_off, float* __restrict__ grad) {
  __shared__ f32x4 red[3 * 256];
  const int blk = blockIdx.x / 9, tap = blockIdx.x - blk * 9;
  const int q = threadIdx.x & 255, part = threadIdx.x >> 8;
  const f32x4* src = reinterpret_cast<const f32x4*>(slab) + ((size_t)blk * 9 + tap) * 256 + q;
  const size_t st = (size_t)n_blocks * 9 * 256;  // quads per slice
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int k = part;
  for (; k + 12 < n_slices; k += 16) {
    const f32x4 v0 = src[(size_t)k * st], v1 = src[(size_t)(k + 4) * st], v2 = src[(size_t)(k + 8) * st], v3 = src[(size_t)(k + 12) * st];
    s += (v0 + v1) + (v2 + v3);
  }
  for (; k < n_slices; k += 4) s += src[(size_t)k * st];
  if (part) red[(part - 1) * 256 + q] = s;
  __syncthreads();
  if (part) return;
  s = (s + red[q]) + (red[256 + q] + red[512 + q]);
  const int ci = (blk / n_co_t) * 32 + (q >> 3);
  if (ci >= cin) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int co = (blk % n_co_t) * 32 + (q & 7) * 4 + j;
    if (co < cout) grad[((size_t)co * cin_total + ci_off + ci) * 9 + tap] = s[j];
  }
}

// pixel-tile width of wgrad_kernel for a W-wide feature map: the candidate with the least padded columns (ties: wider)
static int wgrad_tile_w(int W) {
  const int cand[4] = {32, 24, 16, 12};
  int best = 32, best_pad = (W + 31) / 32 * 32;
  for (int i = 1; i < 4; ++i) {
    const int pad = (W + cand[i] - 1) / cand[i] * cand[i];
    if (pad < best_pad) best = cand[i], best_pad = pad;
  }
  return best;
}

// K slices per (ci, co) tile: enough workgroups to fill 256 CUs x 2, never more than pixel tiles
int wgrad_slices(int B, int H, int W, int blocks) {
  const int tw = wgrad_tile_w(W);
  const int n_tiles = ((W + tw - 1) / tw) * ((H + 3) / 4) * B;
  const int want = (1024 + blocks - 1) / blocks;
  return std::max(1, std::min(n_tiles, std::max(want, 4)));
}
static int64_t wgw_slab_floats(int cxp, int coutp, int B, int H, int W);
int64_t wgrad_slab_floats(int cin_part, int cout, int B, int H, int W) {
  const int blocks = ((pad16(cin_part) + 31) / 32) * ((pad16(cout) + 31) / 32);
  return std::max((int64_t)wgrad_slices(B, H, W, blocks) * blocks * 9 * 1024, wgw_slab_floats(pad16(cin_part), pad16(cout), B, H, W));
}

int launch_wgrad(const WgradArgs& a0, int cin_part, int cout, int cin_total, int ci_off, float* grad, hipStream_t s) {
  WgradArgs a = a0;
  const int n_ci_t = (a.cxp + 31) / 32, n_co_t = (a.coutp + 31) / 32;
  const int n_slices = wgrad_slices(a.B, a.H, a.W, n_ci_t * n_co_t);
  if (a.cxp == 16 && a.coutp == 16)
    hipLaunchKernelGGL(wgrad16_kernel, dim3(1, n_slices), dim3(256), 0, s, a);
  else
    switch (wgrad_tile_w(a.W)) {
      case 32: hipLaunchKernelGGL(wgrad_kernel<32>, dim3(n_ci_t * n_co_t, n_slices), dim3(256), 0, s, a); break;
      case 24: hipLaunchKernelGGL(wgrad_kernel<24>, dim3(n_ci_t * n_co_t, n_slices), dim3(256), 0, s, a); break;
      case 16: hipLaunchKernelGGL(wgrad_kernel<16>, dim3(n_ci_t * n_co_t, n_slices), dim3(256), 0, s, a); break;
      default: hipLaunchKernelGGL(wgrad_kernel<12>, dim3(n_ci_t * n_co_t, n_slices), dim3(256), 0, s, a); break;
    }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(n_ci_t * n_co_t * 9), dim3(1024), 0, s, a.slab, n_slices, n_ci_t * n_co_t, n_co_t, cin_part, cout, cin_total, ci_off, grad);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// 3x3 conv weight gradient in the Winograd F(2x2,3x3) domain (the same identity the forward / data-gradient kernels use,
// differentiated):  Y = A^T [(G g G^T) (.) (B^T d B)] A  =>  dL/dg = G^T [ sum_tiles (B^T d B) (.) (A dY A^T) ] G.
//   Per Winograd position (xi, nu) one GEMM  U[pos][ci][co] = sum over 2x2 output tiles of V[pos][tile][ci] * Z[pos][tile][co]
//   with V = B^T d B of the tile's 4x4 input patch and Z = A dY A^T of its 2x2 output gradient: 16 MFMA products per tile
//   and (ci, co) instead of the 36 of the direct form (4/9 of the matrix work).  Both operands are transformed on the fly
//   from the raw LDS images (lanes run over the channel: conflict-free dword reads, every address an immediate offset of
//   one per-lane base): wave xi combines the two patch rows of "its" row of B^T d (4 VALU), takes the four column
//   combinations (4 VALU) and forms its row of A dY (0 - 2 VALU per output-channel tile) and the column combinations (2);
//   the signs of row / column 3 (-dY) are applied by the finishing kernel instead of in the loop.
//   Workgroup (256 thr) = a 32(ci) x 32 NCO(co) tile, wave = xi, 4 x NCO accumulators per wave, over a K slice of 4 x TW
//   pixel tiles (two tile rows; a K step of the 32x32x2 MFMA is a PAIR of tiles, lane half lh = left / right half of the row).  The waves own
//   disjoint positions: no cross-wave reduction; the slab keeps [slice][block][pos 16][ci 32][co]; wgrad_wino_reduce_kernel
//   adds the slices in a fixed order, wgrad_wino_finish_kernel applies G^T . G and writes the canonical OIHW gradient.
// ---------------------------------------------------------------------------------------
template <int XI, int TW, int NCO>
__device__ __forceinline__ void wgw_tile_steps(const float* __restrict__ sX, const float* __restrict__ sY, int lx, int lh, f32x16 (&acc)[4][NCO]) {
  // A K step is a PAIR of tiles: lane half lh = 0 walks the left half of a tile row, lh = 1 the right half, so that a lane's
  // consecutive steps are ADJACENT tiles -- their 4-column patches overlap by two columns and the row combinations t[2], t[3] of one
  // step are t[0], t[1] of the next (two new columns per step: 4 LDS reads + 2 VALU instead of 8 + 4).
  constexpr int HW = TW + 2, CO = 32 * NCO, HTC = TW / 4;  // HTC: K steps per tile row (TW / 2 tile columns, two per step)
  constexpr int RA = XI == 0 ? 0 : XI == 2 ? 2 : 1, RB = XI == 0 ? 2 : XI == 1 ? 2 : XI == 2 ? 1 : 3;
  const float* bx = sX + (2 * HTC * lh) * 32 + lx;
  const float* by = sY + (2 * HTC * lh) * CO + lx;
  auto tcol = [&](int tr, int col) __attribute__((always_inline)) {
    const float da = bx[((2 * tr + RA) * HW + col) * 32], db = bx[((2 * tr + RB) * HW + col) * 32];
    return XI == 1 ? da + db : da - db;
  };
#pragma unroll
  for (int tr = 0; tr < 2; ++tr) {
    float t0 = tcol(tr, 0), t1 = tcol(tr, 1);
#pragma unroll
    for (int s = 0; s < HTC; ++s) {
      const int c0 = 2 * s;  // compile-time after unrolling
      const float t2 = tcol(tr, c0 + 2), t3 = tcol(tr, c0 + 3);
      const float v[4] = {t0 - t2, t1 + t2, t2 - t1, t1 - t3};
      t0 = t2, t1 = t3;
#pragma unroll
      for (int n = 0; n < NCO; ++n) {
        float r[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const float y0 = XI == 3 ? 0.f : by[((2 * tr) * TW + c0 + b) * CO + n * 32];
          const float y1 = XI == 0 ? 0.f : by[((2 * tr + 1) * TW + c0 + b) * CO + n * 32];
          r[b] = XI == 0 ? y0 : XI == 1 ? y0 + y1 : XI == 2 ? y0 - y1 : y1;  // row 3 of A dY is -dY[1]: sign applied by the finishing kernel
        }
        const float z[4] = {r[0], r[0] + r[1], r[0] - r[1], r[1]};  // column 3 is -r[1]: sign applied by the finishing kernel
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) acc[nu][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[nu], z[nu], acc[nu][n], 0, 0, 0);
      }
    }
  }
}

template <int TW, int NCO>
__global__ __launch_bounds__(256, 2) void wgrad_wino_kernel(WgradArgs a) {
  constexpr int TH = 4, HH = TH + 2, HW = TW + 2, CO = 32 * NCO, QY = CO / 4;
  constexpr int NXQ = HH * HW * 8, NYQ = TH * TW * QY;  // float4 quads per tile image
  // fetch passes: X = one halo row of <= 32 columns x 8 quads per pass (+ one pass for columns 32, 33 of all rows when TW = 32);
  // dY = one tile row of 256 / QY pixels x QY quads per pass
  constexpr int NXE = HW > 32 ? 1 : 0, NX = HH + NXE, YPP = 256 / QY, YH = (TW + YPP - 1) / YPP, NY = TH * YH;
  static_assert(HH * (HW - 32) * 8 <= 256, "the extra halo columns must fit one pass");
  __shared__ float sX[NXQ * 4];
  __shared__ float sY[NYQ * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, lh = lane >> 5;
  const int n_co_t = (a.coutp + CO - 1) / CO;
  const int ci_t = blockIdx.x / n_co_t, co_t = blockIdx.x - ci_t * n_co_t;
  const int slice = blockIdx.y, n_slices = gridDim.y;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  const int n_tiles = tiles_x * tiles_y * a.B;

  f32x16 acc[4][NCO];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int n = 0; n < NCO; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nu][n][r] = 0.f;

  // Next tile: global -> registers while this tile's MFMAs run (fetch), registers -> LDS afterwards (commit).  The loads go through
  // a per-frame buffer descriptor: a quad's address is one v_add of a wave-uniform row offset to a per-thread constant, and quads
  // outside the image ("same" padding; tiles cut by the edge) or beyond the channels get an out-of-range offset -- the hardware's
  // range check returns zeros (no per-quad index arithmetic, no select on the data).
  constexpr unsigned OOB = 0xFFFFFF00u;
  u32x4 rx[NX], ry[NY];
  const int xq = tid & 7, xc = tid >> 3;                       // X pass: quad, halo column
  const int xe_r = tid >> 4, xe_c = 32 + ((tid >> 3) & 1);     // extra pass (TW = 32): halo row, halo column 32 / 33
  const int yq = tid & (QY - 1), yc = tid / QY;                // dY pass: quad, pixel column within the pass
  const bool cx_ok = ci_t * 32 + xq * 4 < a.cxp, cy_ok = co_t * CO + yq * 4 < a.coutp;
  const unsigned x_thr = (unsigned)(xc * a.cxp + ci_t * 32 + xq * 4) * 4u, y_thr = (unsigned)(yc * a.coutp + co_t * CO + yq * 4) * 4u;
  const unsigned xe_thr = (unsigned)((xe_r * a.W + xe_c) * a.cxp + ci_t * 32 + xq * 4) * 4u;
  const unsigned x_frame = (unsigned)(a.H * a.W) * (unsigned)(a.cxp * 4), y_frame = (unsigned)(a.H * a.W) * (unsigned)(a.coutp * 4);  // bytes per frame (< 4 GiB: checked at launch)
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    int t = tile;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)b * (x_frame / 4), 0, (int)x_frame, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy) + (size_t)b * (y_frame / 4), 0, (int)y_frame, 0x00020000);
    const int gx = x0 + xc - 1;
    const bool x_ok = cx_ok && xc < HW && gx >= 0 && gx < a.W;
#pragma unroll
    for (int j = 0; j < HH; ++j) {  // halo row j, columns 0 .. 31
      const int gy = y0 + j - 1;                                                  // wave-uniform
      const unsigned row = (unsigned)((gy * a.W + x0 - 1) * a.cxp) * 4u;          // wave-uniform; wraps harmlessly when the row is not used
      rx[j] = __builtin_amdgcn_raw_buffer_load_b128(xr, (x_ok && gy >= 0 && gy < a.H) ? x_thr + row : OOB, 0, 0);
    }
    if constexpr (NXE) {  // halo columns 32, 33 of all six rows
      const int gy = y0 + xe_r - 1, gxe = x0 + xe_c - 1;
      const bool ok = cx_ok && xe_r < HH && gy >= 0 && gy < a.H && gxe < a.W;
      rx[HH] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? xe_thr + (unsigned)(((y0 - 1) * a.W + x0 - 1) * a.cxp) * 4u : OOB, 0, 0);
    }
#pragma unroll
    for (int h = 0; h < YH; ++h) {
      const bool y_ok = cy_ok && h * YPP + yc < TW && x0 + h * YPP + yc < a.W;
#pragma unroll
      for (int py = 0; py < TH; ++py) {
        const unsigned row = (unsigned)(((y0 + py) * a.W + x0 + h * YPP) * a.coutp) * 4u;  // wave-uniform
        ry[py * YH + h] = __builtin_amdgcn_raw_buffer_load_b128(yr, (y_ok && y0 + py < a.H) ? y_thr + row : OOB, 0, 0);
      }
    }
  };
  auto commit = [&]() __attribute__((always_inline)) {
    if (xc < HW) {
#pragma unroll
      for (int j = 0; j < HH; ++j) *reinterpret_cast<u32x4*>(sX + ((j * HW + xc) * 8 + xq) * 4) = rx[j];
    }
    if constexpr (NXE) {
      if (xe_r < HH) *reinterpret_cast<u32x4*>(sX + ((xe_r * HW + xe_c) * 8 + xq) * 4) = rx[HH];
    }
#pragma unroll
    for (int h = 0; h < YH; ++h)
      if (h * YPP + yc < TW) {
#pragma unroll
        for (int py = 0; py < TH; ++py) *reinterpret_cast<u32x4*>(sY + ((py * TW + h * YPP + yc) * QY + yq) * 4) = ry[py * YH + h];
      }
  };

  if (slice < n_tiles) fetch(slice);
  for (int tile = slice; tile < n_tiles; tile += n_slices) {
    commit();
    __syncthreads();
    if (tile + n_slices < n_tiles) fetch(tile + n_slices);
    switch (wave) {  // wave-uniform: wave xi owns row xi of the transformed patch
      case 0: wgw_tile_steps<0, TW, NCO>(sX, sY, lx, lh, acc); break;
      case 1: wgw_tile_steps<1, TW, NCO>(sX, sY, lx, lh, acc); break;
      case 2: wgw_tile_steps<2, TW, NCO>(sX, sY, lx, lh, acc); break;
      default: wgw_tile_steps<3, TW, NCO>(sX, sY, lx, lh, acc); break;
    }
    __syncthreads();
  }
  // slab[slice][block][pos = 4 xi + nu][ci 32][co CO];  D: row(ci) = (r&3) + 8*(r>>2) + 4*lh, col(co) = lx
  float* slab = a.slab + (((size_t)slice * gridDim.x + blockIdx.x) * 16 + wave * 4) * (32 * CO);
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int n = 0; n < NCO; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[(size_t)nu * (32 * CO) + ((r & 3) + 8 * (r >> 2) + 4 * lh) * CO + n * 32 + lx] = acc[nu][n][r];
}

// usum[slot][256 quads] = sum over slices of slab[slice][slot][...] in a fixed order; slot = 256-quad piece of a (block, position) image.
// 1024 threads = 256 quads x 4 slice parts (four loads in flight per thread), the parts meet in LDS.
__global__ __launch_bounds__(1024) void wgrad_wino_reduce_kernel(const float* __restrict__ slab, int n_slices, size_t slice_quads, float* __restrict__ usum) {
  __shared__ f32x4 red[3 * 256];
  const int q = threadIdx.x & 255, part = threadIdx.x >> 8;
  const f32x4* src = reinterpret_cast<const f32x4*>(slab) + (size_t)blockIdx.x * 256 + q;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int k = part;
  for (; k + 12 < n_slices; k += 16) {
    const f32x4 v0 = src[(size_t)k * slice_quads], v1 = src[(size_t)(k + 4) * slice_quads], v2 = src[(size_t)(k + 8) * slice_quads], v3 = src[(size_t)(k + 12) * slice_quads];
    s += (v0 + v1) + (v2 + v3);
  }
  for (; k < n_slices; k += 4) s += src[(size_t)k * slice_quads];
  if (part) red[(part - 1) * 256 + q] = s;
  __syncthreads();
  if (part) return;
  reinterpret_cast<f32x4*>(usum)[(size_t)blockIdx.x * 256 + q] = (s + red[q]) + (red[256 + q] + red[512 + q]);
}

// dg = G^T U' G per (ci, co), U'[xi][nu] = s(xi) s(nu) U[xi][nu] with s(3) = -1 (the loop used +dY where A dY A^T has -dY);
// grad[(co * cin_total + ci_off + ci) * 9 + ky * 3 + kx].  One thread per (block, ci, co).
__global__ __launch_bounds__(256) void wgrad_wino_finish_kernel(const float* __restrict__ usum, int n_blocks, int n_co_t, int ci_w /* 32, or 16 */, int co_w /* 32 NCO, or 16 */, int cin, int cout,
                                                                int cin_total, int ci_off, float* __restrict__ grad) {
  const int per = ci_w * co_w;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_blocks * per) return;
  const int blk = idx / per, e = idx - blk * per;
  const int ci = (blk / n_co_t) * ci_w + e / co_w, co = (blk % n_co_t) * co_w + e % co_w;
  if (ci >= cin || co >= cout) return;
  float u[4][4];
#pragma unroll
  for (int xi = 0; xi < 4; ++xi)
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      const float v = usum[((size_t)blk * 16 + xi * 4 + nu) * per + e];
      u[xi][nu] = ((xi == 3) != (nu == 3)) ? -v : v;
    }
  float w[3][4];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu) {
    const float h = 0.5f * (u[1][nu] + u[2][nu]);
    w[0][nu] = u[0][nu] + h;
    w[1][nu] = 0.5f * (u[1][nu] - u[2][nu]);
    w[2][nu] = h + u[3][nu];
  }
  float* g = grad + ((size_t)co * cin_total + ci_off + ci) * 9;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const float h = 0.5f * (w[ky][1] + w[ky][2]);
    g[ky * 3 + 0] = w[ky][0] + h;
    g[ky * 3 + 1] = 0.5f * (w[ky][1] - w[ky][2]);
    g[ky * 3 + 2] = h + w[ky][3];
  }
}

// 16 -> 16 layers (the first encoder block at full resolution): the same Winograd-domain gradient on v_mfma_f32_16x16x4_f32 -- a K step is
// FOUR tiles (lane quarter lg walks its quarter of a tile row, so consecutive steps are adjacent tiles and share two patch columns),
// A[i = ci][k = lg], B[k = lg][j = co], four positions (row xi) x one 16 x 16 accumulator per wave.  Pixel tile 4 x 32, whole-row fetch
// passes through per-frame buffer descriptors as in wgrad_wino_kernel.  Slab [slice][pos 16][ci 16][co 16].
template <int XI>
__device__ __forceinline__ void wgw16_tile_steps(const float* __restrict__ sX, const float* __restrict__ sY, int li, int lg, f32x4 (&acc)[4]) {
  constexpr int TW = 32, HW = TW + 2, Q = TW / 8;  // Q: K steps per tile row (16 tile columns, four per step)
  constexpr int RA = XI == 0 ? 0 : XI == 2 ? 2 : 1, RB = XI == 0 ? 2 : XI == 1 ? 2 : XI == 2 ? 1 : 3;
  const float* bx = sX + (2 * Q * lg) * 16 + li;
  const float* by = sY + (2 * Q * lg) * 16 + li;
  auto tcol = [&](int tr, int col) __attribute__((always_inline)) {
    const float da = bx[((2 * tr + RA) * HW + col) * 16], db = bx[((2 * tr + RB) * HW + col) * 16];
    return XI == 1 ? da + db : da - db;
  };
#pragma unroll
  for (int tr = 0; tr < 2; ++tr) {
    float t0 = tcol(tr, 0), t1 = tcol(tr, 1);
#pragma unroll
    for (int s = 0; s < Q; ++s) {
      const int c0 = 2 * s;
      const float t2 = tcol(tr, c0 + 2), t3 = tcol(tr, c0 + 3);
      const float v[4] = {t0 - t2, t1 + t2, t2 - t1, t1 - t3};
      t0 = t2, t1 = t3;
      float r[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float y0 = XI == 3 ? 0.f : by[((2 * tr) * TW + c0 + b) * 16];
        const float y1 = XI == 0 ? 0.f : by[((2 * tr + 1) * TW + c0 + b) * 16];
        r[b] = XI == 0 ? y0 : XI == 1 ? y0 + y1 : XI == 2 ? y0 - y1 : y1;
      }
      const float z[4] = {r[0], r[0] + r[1], r[0] - r[1], r[1]};
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) acc[nu] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[nu], z[nu], acc[nu], 0, 0, 0);
    }
  }
}

__global__ __launch_bounds__(256, 2) void wgrad16_wino_kernel(WgradArgs a) {
  constexpr int TW = 32, TH = 4, HH = TH + 2, HW = TW + 2;
  __shared__ float sX[HH * HW * 16];
  __shared__ float sY[TH * TW * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int slice = blockIdx.y, n_slices = gridDim.y;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  const int n_tiles = tiles_x * tiles_y * a.B;
  f32x4 acc[4];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu) acc[nu] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr unsigned OOB = 0xFFFFFF00u;
  // fetch passes: 256 threads = 64 pixels x 4 quads: X = one halo row (34 columns) per pass, dY = two tile rows (2 x 32 pixels) per pass
  u32x4 rx[HH], ry[TH / 2];
  const int q = tid & 3, pc = tid >> 2;  // quad, pixel column of the pass
  const int yrow = pc >> 5, ycol = pc & 31;
  const unsigned x_thr = (unsigned)(pc * 16 + q * 4) * 4u, y_thr = (unsigned)((yrow * a.W + ycol) * 16 + q * 4) * 4u;
  const unsigned frame = (unsigned)(a.H * a.W) * 64u;  // bytes per frame of a 16-channel tensor
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    int t = tile;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)b * (frame / 4), 0, (int)frame, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy) + (size_t)b * (frame / 4), 0, (int)frame, 0x00020000);
    const int gx = x0 + pc - 1;
    const bool x_ok = pc < HW && gx >= 0 && gx < a.W;
#pragma unroll
    for (int j = 0; j < HH; ++j) {
      const int gy = y0 + j - 1;
      const unsigned row = (unsigned)((gy * a.W + x0 - 1) * 16) * 4u;
      rx[j] = __builtin_amdgcn_raw_buffer_load_b128(xr, (x_ok && gy >= 0 && gy < a.H) ? x_thr + row : OOB, 0, 0);
    }
    const bool y_ok = x0 + ycol < a.W;
#pragma unroll
    for (int h = 0; h < TH / 2; ++h) {
      const int gy = y0 + 2 * h + yrow;
      const unsigned row = (unsigned)(((y0 + 2 * h) * a.W + x0) * 16) * 4u;
      ry[h] = __builtin_amdgcn_raw_buffer_load_b128(yr, (y_ok && gy < a.H) ? y_thr + row : OOB, 0, 0);
    }
  };
  auto commit = [&]() __attribute__((always_inline)) {
    if (pc < HW) {
#pragma unroll
      for (int j = 0; j < HH; ++j) *reinterpret_cast<u32x4*>(sX + ((j * HW + pc) * 4 + q) * 4) = rx[j];
    }
#pragma unroll
    for (int h = 0; h < TH / 2; ++h) *reinterpret_cast<u32x4*>(sY + (((2 * h + yrow) * TW + ycol) * 4 + q) * 4) = ry[h];
  };
  if (slice < n_tiles) fetch(slice);
  for (int tile = slice; tile < n_tiles; tile += n_slices) {
    commit();
    __syncthreads();
    if (tile + n_slices < n_tiles) fetch(tile + n_slices);
    switch (wave) {
      case 0: wgw16_tile_steps<0>(sX, sY, li, lg, acc); break;
      case 1: wgw16_tile_steps<1>(sX, sY, li, lg, acc); break;
      case 2: wgw16_tile_steps<2>(sX, sY, li, lg, acc); break;
      default: wgw16_tile_steps<3>(sX, sY, li, lg, acc); break;
    }
    __syncthreads();
  }
  // slab[slice][pos = 4 xi + nu][ci 16][co 16];  D: row(ci) = 4 lg + r, col(co) = li
  float* slab = a.slab + ((size_t)slice * 16 + wave * 4) * 256;
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int r = 0; r < 4; ++r) slab[nu * 256 + (4 * lg + r) * 16 + li] = acc[nu][r];
}

int launch_wgrad16_wino(const WgradArgs& a, int cin_part, int cout, int cin_total, int ci_off, float* grad, hipStream_t s) {
  PH_REQUIRE(a.cxp == 16 && a.coutp == 16, "wgrad16_wino: 16 padded channels on both sides");
  PH_REQUIRE((uint64_t)a.H * a.W * 64 < 0xFFFFFF00ull, "wgrad16_wino: a %d x %d frame does not fit a 32-bit buffer descriptor", a.H, a.W);
  const int n_slices = wgrad_slices(a.B, a.H, a.W, 1);
  hipLaunchKernelGGL(wgrad16_wino_kernel, dim3(1, n_slices), dim3(256), 0, s, a);
  const size_t slice_quads = 16 * 256 / 4;  // 1024 quads per slice
  float* usum = a.slab + (size_t)n_slices * slice_quads * 4;
  hipLaunchKernelGGL(wgrad_wino_reduce_kernel, dim3((unsigned)(slice_quads / 256)), dim3(1024), 0, s, a.slab, n_slices, slice_quads, usum);
  hipLaunchKernelGGL(wgrad_wino_finish_kernel, dim3(1), dim3(256), 0, s, usum, 1, 1, 16, 16, cin_part, cout, cin_total, ci_off, grad);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

static int wgw_nco(int coutp) { return coutp >= 64 ? 2 : 1; }
static int64_t wgw_slab_floats(int cxp, int coutp, int B, int H, int W) {
  const int co_w = 32 * wgw_nco(coutp);
  const int blocks = ((cxp + 31) / 32) * ((coutp + co_w - 1) / co_w);
  return ((int64_t)wgrad_slices(B, H, W, blocks) + 1) * blocks * 16 * 32 * co_w;  // + 1: the slice sums (usum)
}

int launch_wgrad_wino(const WgradArgs& a, int cin_part, int cout, int cin_total, int ci_off, float* grad, hipStream_t s) {
  PH_REQUIRE((uint64_t)a.H * a.W * std::max(a.cxp, a.coutp) * 4 < 0xFFFFFF00ull, "wgrad: a %d x %d x %d frame does not fit a 32-bit buffer descriptor", a.H, a.W, std::max(a.cxp, a.coutp));
  const int nco = wgw_nco(a.coutp), co_w = 32 * nco;
  const int n_ci_t = (a.cxp + 31) / 32, n_co_t = (a.coutp + co_w - 1) / co_w, blocks = n_ci_t * n_co_t;
  const int n_slices = wgrad_slices(a.B, a.H, a.W, blocks);
  const dim3 grid(blocks, n_slices);
#define PH_WGW(T, N) hipLaunchKernelGGL((wgrad_wino_kernel<T, N>), grid, dim3(256), 0, s, a)
  switch (wgrad_tile_w(a.W) * 4 + nco) {
    case 32 * 4 + 2: PH_WGW(32, 2); break;
    case 32 * 4 + 1: PH_WGW(32, 1); break;
    case 24 * 4 + 2: PH_WGW(24, 2); break;
    case 24 * 4 + 1: PH_WGW(24, 1); break;
    case 16 * 4 + 2: PH_WGW(16, 2); break;
    case 16 * 4 + 1: PH_WGW(16, 1); break;
    case 12 * 4 + 2: PH_WGW(12, 2); break;
    default: PH_WGW(12, 1); break;
  }
#undef PH_WGW
  const size_t slice_quads = (size_t)blocks * 16 * 8 * co_w;  // quads per slice = blocks * 16 * 32 * co_w / 4
  float* usum = a.slab + (size_t)n_slices * slice_quads * 4;
  hipLaunchKernelGGL(wgrad_wino_reduce_kernel, dim3((unsigned)(slice_quads / 256)), dim3(1024), 0, s, a.slab, n_slices, slice_quads, usum);
  const int n_el = blocks * 32 * co_w;
  hipLaunchKernelGGL(wgrad_wino_finish_kernel, dim3((n_el + 255) / 256), dim3(256), 0, s, usum, blocks, n_co_t, 32, co_w, cin_part, cout, cin_total, ci_off, grad);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// First-conv weight gradient (Cin 1 or 3 image, NCHW uint8/float):
//   dW[co][ci][tap] = sum_{b,p} dY[b,p,co] * img[b,ci,p+tap] (normalised).  Persistent blocks walk
//   pixel tiles; thread = (co, tap-ci) pair keeps one partial in a register.
// ---------------------------------------------------------------------------------------
constexpr int ICW_BLOCKS = 1024;
// with_bias: column n_comb of the product (free: 9 or 27 of the 16 / 32 columns are taps) multiplies dY by 1 -- the conv's bias gradient, written after
// the weight partials of a block (row pitch n_out + cout), so that the bias needs no pass of its own over the first layer's gradient, the largest tensor
// of the backward.
__global__ __launch_bounds__(256) void input_wgrad_partial_kernel(const void* __restrict__ img, int dtype, const float* __restrict__ dy, int B, int cin, int H, int W,
                                                                  int coutp, int cout, int with_bias, float* __restrict__ partial /* ICW_BLOCKS x (cout*cin*9 [+ cout]) */) {
  // A tiny GEMM per pixel tile on v_mfma_f32_16x16x4_f32: D[co 16][(ci, tap) <= 32] += dY^T[co][4 pixels] * patch[4 pixels][(ci, tap)].
  // Wave w owns pixel rows 2w, 2w+1 of the 8 x 32 tile; a lane's A operand is dY[pixel 4s + kg][co = lane & 15] (64 consecutive
  // floats per wave), its B operand the image value under tap (lane & 15) of pixel 4s + kg.
  __shared__ float sImg[3 * 10 * 34 + 4];
  __shared__ __attribute__((aligned(16))) float sDy[8 * 32 * 16];
  __shared__ float sRed[4 * 2 * 256];
  const int n_out = cout * cin * 9, n_comb = cin * 9;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 15, kg = lane >> 4;
  const int tiles_x = (W + 31) / 32, tiles_y = (H + 7) / 8;
  const int n_tiles = tiles_x * tiles_y * B;
  const int pitch = n_out + (with_bias ? cout : 0);
  int boff[2];
  bool bok[2];
  float bone[2];  // the operand of a lane that is no tap: 1 in the bias column, 0 elsewhere
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int comb = nb * 16 + ln;
    bok[nb] = comb < n_comb;
    bone[nb] = (with_bias && comb == n_comb) ? 1.f : 0.f;
    const int cc = bok[nb] ? comb : 0;
    const int ci = cc / 9, tap = cc - ci * 9;
    boff[nb] = ci * 340 + (tap / 3) * 34 + tap % 3;
  }
  for (int cbase = 0; cbase < cout; cbase += 16) {  // 16 output channels at a time
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
      int t = tile;
      const int tx = t % tiles_x;
      t /= tiles_x;
      const int ty = t % tiles_y;
      const int b = t / tiles_y;
      const int x0 = tx * 32, y0 = ty * 8;
      for (int i = tid; i < cin * 340; i += 256) {
        const int c = i / 340, r = i - c * 340;
        const int hy = r / 34, hx = r - hy * 34;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        float v = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
          const size_t o = (((size_t)b * cin + c) * H + gy) * W + gx;
          if (dtype == 0)
            v = (float)reinterpret_cast<const uint8_t*>(img)[o] / 255.0f;
          else {
            v = reinterpret_cast<const float*>(img)[o];
            if (dtype == 2) v = v / 255.0f;
          }
        }
        sImg[i] = v;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // dY tile: 256 pixels x 16 channels, one float4 per thread per trip
        const int i = tid + 256 * j;
        const int pix = i >> 2, q = i & 3;
        const int gy = y0 + (pix >> 5), gx = x0 + (pix & 31);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gy < H && gx < W) v = *reinterpret_cast<const f32x4*>(dy + (((size_t)b * H + gy) * W + gx) * coutp + cbase + q * 4);  // coutp is a multiple of 16
        *reinterpret_cast<f32x4*>(sDy + pix * 16 + q * 4) = v;
      }
      __syncthreads();
#pragma unroll 4
      for (int s4 = 0; s4 < 16; ++s4) {
        const int p = 4 * s4 + kg;
        const int py = 2 * wave + (p >> 5), px = p & 31;
        const float av = sDy[(py * 32 + px) * 16 + ln];
        const int ib = py * 34 + px;
        const float b0 = bok[0] ? sImg[boff[0] + ib] : bone[0];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc[0], 0, 0, 0);
        if (n_comb >= 16) {  // (cin = 3: taps 16 .. 26, the bias column 27)
          const float b1 = bok[1] ? sImg[boff[1] + ib] : bone[1];
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc[1], 0, 0, 0);
        }
      }
      __syncthreads();
    }
    // D[i = co = 4 * kg + r][j = comb = ln]; the four waves' partial sums meet in LDS in a fixed order
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) sRed[(wave * 2 + nb) * 256 + (4 * kg + r) * 16 + ln] = acc[nb][r];
    __syncthreads();
    for (int i = tid; i < 512; i += 256) {
      const int nb = i >> 8, co = (i >> 4) & 15, comb = nb * 16 + (i & 15);
      if (cbase + co < cout && (comb < n_comb || (with_bias && comb == n_comb))) {
        const int e = (i & 255);
        const float sum = (sRed[(0 * 2 + nb) * 256 + e] + sRed[(1 * 2 + nb) * 256 + e]) + (sRed[(2 * 2 + nb) * 256 + e] + sRed[(3 * 2 + nb) * 256 + e]);
        partial[(size_t)blockIdx.x * pitch + (comb < n_comb ? (size_t)(cbase + co) * n_comb + comb : (size_t)n_out + cbase + co)] = sum;
      }
    }
    __syncthreads();
  }
}
// 1024 threads = 256 outputs x 4 block parts (four loads in flight per thread); the parts meet in LDS in a fixed order
__global__ __launch_bounds__(1024) void input_wgrad_final_kernel(const float* __restrict__ partial, int n_blocks, int n_out, int pitch, float* __restrict__ gw) {
  // The first layer's weight gradient is the worst-conditioned sum of the network (every pixel of the batch contributes a
  // signed term to each of the 9 * cin * cout outputs and they largely cancel): the per-tile partials are combined in double
  // (a few hundred adds per output, free) so that the result is as close to the exact gradient as ATen's own fp32 reduction.
  __shared__ double red[3 * 256];
  const int e = threadIdx.x & 255, part = threadIdx.x >> 8;
  const int i = blockIdx.x * 256 + e;
  const bool ok = i < n_out;
  const float* src = partial + (ok ? i : 0);
  double s = 0.0;
  int k = part;
  for (; k + 12 < n_blocks; k += 16)
    s += ((double)src[(size_t)k * pitch] + (double)src[(size_t)(k + 4) * pitch]) + ((double)src[(size_t)(k + 8) * pitch] + (double)src[(size_t)(k + 12) * pitch]);
  for (; k < n_blocks; k += 4) s += (double)src[(size_t)k * pitch];
  if (part) red[(part - 1) * 256 + e] = s;
  __syncthreads();
  if (part || !ok) return;
  gw[i] = (float)((s + red[e]) + (red[256 + e] + red[512 + e]));
}
// gb != nullptr: dy is complete (masked) -- its per-channel sums, the conv's bias gradient, come out of the same launch
int launch_input_wgrad(const void* img, int dtype, const float* dy, int B, int cin, int H, int W, int coutp, int cout, float* gw, float* scratch, hipStream_t s, float* gb) {
  if (cin > 3) {
    set_error("input wgrad: more than 3 input channels");
    return PH_E_INVALID;
  }
  const int n_tiles = ((W + 31) / 32) * ((H + 7) / 8) * B;
  const int blocks = std::min(n_tiles, ICW_BLOCKS);
  const int n_out = cout * cin * 9, pitch = n_out + (gb ? cout : 0);
  hipLaunchKernelGGL(input_wgrad_partial_kernel, dim3(blocks), dim3(256), 0, s, img, dtype, dy, B, cin, H, W, coutp, cout, gb ? 1 : 0, scratch);
  hipLaunchKernelGGL(input_wgrad_final_kernel, dim3((n_out + 255) / 256), dim3(1024), 0, s, scratch, blocks, n_out, pitch, gw);
  if (gb) hipLaunchKernelGGL(input_wgrad_final_kernel, dim3((cout + 255) / 256), dim3(1024), 0, s, scratch + n_out, blocks, cout, pitch, gb);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}
int64_t input_wgrad_scratch_floats(int cin, int cout) { return (int64_t)ICW_BLOCKS * (cout * cin * 9 + cout); }

// ---------------------------------------------------------------------------------------
// Adam (torch.optim.Adam, weight_decay 0) and the canonical -> packed weight gather.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   float* __restrict__ vmax, size_t n, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                   float grad_scale, float decay /* 1 - lr * weight_decay (AdamW), 1 for Adam */) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float gi = g[i] * grad_scale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float vv = vi;
    if (vmax) {
      vv = fmaxf(vmax[i], vi);
      vmax[i] = vv;
    }
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = p[i] * decay - (lr / bc1) * (mi / denom);  // AdamW: param.mul_(1 - lr * weight_decay) first, then the Adam update
  }
}

__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ canon, const int* __restrict__ map, size_t n, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int j = map[i];
    out[i] = j >= 0 ? canon[j] : 0.f;
  }
}

// All packed buffers of a model in ONE launch: block -> segment by binary search over the segments' first blocks (1024 elements per block).
__global__ __launch_bounds__(256) void gather_multi_kernel(const float* __restrict__ canon, const GatherSegment* __restrict__ seg, int n_seg) {
  int lo = 0, hi = n_seg - 1;
  while (lo < hi) {  // last segment whose first_block <= blockIdx.x (block-uniform)
    const int mid = (lo + hi + 1) >> 1;
    if (seg[mid].first_block <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const GatherSegment sg = seg[lo];
  const size_t base = (size_t)(blockIdx.x - sg.first_block) * 1024;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const size_t i = base + k * 256 + threadIdx.x;
    if (i < sg.n) {
      const int j = sg.map[i];
      sg.dst[i] = j >= 0 ? canon[j] : 0.f;
    }
  }
}

int launch_gather_multi(const float* canon, const GatherSegment* seg_dev, int n_seg, unsigned total_blocks, hipStream_t s) {
  if (n_seg == 0 || total_blocks == 0) return PH_OK;
  hipLaunchKernelGGL(gather_multi_kernel, dim3(total_blocks), dim3(256), 0, s, canon, seg_dev, n_seg);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

int launch_gather(const float* canon, const int* map, size_t n, float* out, hipStream_t s) {
  if (n == 0) return PH_OK;
  hipLaunchKernelGGL(gather_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 16384)), dim3(256), 0, s, canon, map, n, out);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// On-device training targets (the reference renders them per sample on the CPU in the Dataset).
//   confidence maps: per node, max over instances of exp(-((x-px)^2 + (y-py)^2) / (2 (sigma*stride)^2)),
//     NaN points contribute 0            (data/confidence_maps.py:36-41,96-166)
//   part-affinity fields: per edge, sum over instances of exp(-(d2)^2 / (2 sigma^2)) * unit(dst-src), with
//     d2 = SQUARED distance to the segment (the reference feeds the squared distance to its Gaussian,
//     data/edge_maps.py:15-78,120-220 -- kept as is); instances with no node strictly inside
//     (0, grid_max) are dropped, NaN contributions are zeroed (edge_maps.py:286-290,213).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void render_confmaps_kernel(const float* __restrict__ pts /* B,I,N,2 */, int B, int I, int N, int h, int w, int stride, float sigma,
                                                              float* __restrict__ out /* B,N,h,w */) {
  const size_t total = (size_t)B * N * h * w;
  const float inv = 1.0f / (2.0f * sigma * sigma);
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int x = (int)(idx % w);
    size_t r = idx / w;
    const int y = (int)(r % h);
    r /= h;
    const int n = (int)(r % N);
    const int b = (int)(r / N);
    const float gx = (float)(x * stride), gy = (float)(y * stride);
    float best = 0.f;
    for (int i = 0; i < I; ++i) {
      const float px = pts[(((size_t)b * I + i) * N + n) * 2], py = pts[(((size_t)b * I + i) * N + n) * 2 + 1];
      const float dx = gx - px, dy = gy - py;
      const float v = expf(-(dx * dx + dy * dy) * inv);
      if (v == v) best = fmaxf(best, v);  // NaN -> 0 (nan_to_num), then maximum with the zero init
    }
    out[idx] = best;
  }
}

__global__ __launch_bounds__(256) void render_pafs_kernel(const float* __restrict__ pts /* B,I,N,2 */, const int* __restrict__ edges /* E,2 */, int B, int I, int N, int E,
                                                          int h, int w, int stride, float sigma, float* __restrict__ out /* B,2E,h,w */) {
  const size_t total = (size_t)B * E * h * w;
  const float inv = 1.0f / (2.0f * sigma * sigma);
  const float xmax = (float)((w - 1) * stride), ymax = (float)((h - 1) * stride);
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int x = (int)(idx % w);
    size_t r = idx / w;
    const int y = (int)(r % h);
    r /= h;
    const int e = (int)(r % E);
    const int b = (int)(r / E);
    const float gx = (float)(x * stride), gy = (float)(y * stride);
    const int sn = edges[2 * e], dn = edges[2 * e + 1];
    float ax = 0.f, ay = 0.f;
    for (int i = 0; i < I; ++i) {
      const float* ip = pts + ((size_t)b * I + i) * N * 2;
      bool any_in = false;
      for (int n = 0; n < N; ++n) {
        const float qx = ip[2 * n], qy = ip[2 * n + 1];
        any_in = any_in || (qx > 0.f && qy > 0.f && qx < xmax && qy < ymax);
      }
      if (!any_in) continue;
      const float sx = ip[2 * sn], sy = ip[2 * sn + 1], tx = ip[2 * dn], ty = ip[2 * dn + 1];
      const float vx = tx - sx, vy = ty - sy;
      const float len2 = fmaxf(vx * vx + vy * vy, 1.0f);
      const float rx = gx - sx, ry = gy - sy;
      float t = (rx * vx + ry * vy) / len2;
      t = fminf(fmaxf(t, 0.f), 1.f);
      const float ex = t * vx - rx, ey = t * vy - ry;
      const float d2 = ex * ex + ey * ey;
      const float g = expf(-(d2 * d2) * inv);
      const float norm = sqrtf(vx * vx + vy * vy);
      const float px = g * (vx / norm), py = g * (vy / norm);
      if (px == px) ax += px;
      if (py == py) ay += py;
    }
    const size_t plane = (size_t)h * w;
    float* o = out + ((size_t)b * 2 * E + 2 * e) * plane + (size_t)y * w + x;
    o[0] = ax;
    o[plane] = ay;
  }
}

}  // namespace ph

using namespace ph;

extern "C" {

int ph_render_confmaps(const float* points_dev, int32_t B, int32_t I, int32_t N, int32_t img_h, int32_t img_w, int32_t stride, float sigma, float* out_dev,
                       void* stream) {
  PH_REQUIRE(points_dev && out_dev && B > 0 && I >= 0 && N > 0 && img_h > 0 && img_w > 0 && stride > 0, "ph_render_confmaps: bad arguments");
  const int h = (img_h + stride - 1) / stride, w = (img_w + stride - 1) / stride;  // len(arange(0, size, stride))
  const size_t total = (size_t)B * N * h * w;
  hipLaunchKernelGGL(render_confmaps_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 16384)), dim3(256), 0, static_cast<hipStream_t>(stream), points_dev, B,
                     I, N, h, w, stride, sigma * (float)stride, out_dev);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

int ph_render_pafs(const float* points_dev, const int32_t* edges_dev, int32_t B, int32_t I, int32_t N, int32_t E, int32_t img_h, int32_t img_w, int32_t stride,
                   float sigma, float* out_dev, void* stream) {
  PH_REQUIRE(points_dev && edges_dev && out_dev && B > 0 && I >= 0 && N > 0 && E > 0 && img_h > 0 && img_w > 0 && stride > 0, "ph_render_pafs: bad arguments");
  const int h = (img_h + stride - 1) / stride, w = (img_w + stride - 1) / stride;
  const size_t total = (size_t)B * E * h * w;
  hipLaunchKernelGGL(render_pafs_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 16384)), dim3(256), 0, static_cast<hipStream_t>(stream), points_dev,
                     edges_dev, B, I, N, E, h, w, stride, sigma, out_dev);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

int ph_adamw_step(float* params_dev, const float* grads_dev, float* exp_avg_dev, float* exp_avg_sq_dev, float* max_exp_avg_sq_dev, int64_t n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_scale, void* stream) {
  PH_REQUIRE(params_dev && grads_dev && exp_avg_dev && exp_avg_sq_dev && n >= 0 && step >= 1 && weight_decay >= 0.f, "ph_adamw_step: bad arguments");
  if (n == 0) return PH_OK;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 16384)), dim3(256), 0, static_cast<hipStream_t>(stream), params_dev, grads_dev,
                     exp_avg_dev, exp_avg_sq_dev, max_exp_avg_sq_dev, (size_t)n, lr, beta1, beta2, eps, bc1, sqrtf(bc2), grad_scale, 1.f - lr * weight_decay);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

int ph_adam_step(float* params_dev, const float* grads_dev, float* exp_avg_dev, float* exp_avg_sq_dev, float* max_exp_avg_sq_dev, int64_t n, float lr,
                 float beta1, float beta2, float eps, int32_t step, float grad_scale, void* stream) {
  return ph_adamw_step(params_dev, grads_dev, exp_avg_dev, exp_avg_sq_dev, max_exp_avg_sq_dev, n, lr, beta1, beta2, eps, 0.f, step, grad_scale, stream);
}

}  // extern "C"
