// Backward / training kernels of the ConvNeXt encoder ops for gfx950 (MI355X, CDNA4).
//
// Autograd semantics restated (the reference trains these modules through torch autograd,
// training/lightning_modules.py:1850-1922 over architectures/convnext.py:19-130):
//   GELU (erf) ................ dx = dy * (Phi(x) + x * phi(x))
//   layer_scale * u + x ....... du = dy * s;  dx += dy;  ds = sum_pixels dy * u
//   LayerNorm over channels ... dx = rstd * (g*dy - mean_c(g*dy) - xhat * mean_c(g*dy*xhat)),
//                               dg = sum_pixels dy * xhat,  db = sum_pixels dy
//   depthwise 7x7 ............. dx = dwconv(dy, flipped kernel);  dw[c][tap] = sum_pixels dy[p] * x[p + tap]
//   Linear / Conv2d(k2,s2) .... dx: the forward row GEMM on transposed weights;
//                               dW[n][k] = sum_rows dY[m][n] * A[m][k]  (row-wgrad kernel, MFMA, K = rows)
//   patch stem ................ dW[co][ci][tap] = sum_pixels dy[p][co] * image[ci, p*stride + tap - 1]
// All reductions over pixels are two-stage with a fixed combination order (deterministic).
#include <algorithm>

#include "common.h"
#include "device_math.h"
#include "train_kernels.h"

namespace ph {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float LN_EPS_T = 1e-6f;
constexpr int RED_SLICES = 1024;

// ---------------------------------------------------------------------------------------
// Elementwise forward ops of the training (unfused) program.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = gelu_f(v[e]);
    reinterpret_cast<f32x4*>(y)[i] = r;
  }
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x, float* __restrict__ gx, int accumulate, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    const f32x4 g = reinterpret_cast<const f32x4*>(gy)[i];
    f32x4 r = accumulate ? reinterpret_cast<const f32x4*>(gx)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] += g[e] * gelu_grad_f(v[e]);
    reinterpret_cast<f32x4*>(gx)[i] = r;
  }
}

// y = scale[c] * u + x      (CNBlock: layer_scale * block(x) + x)
__global__ __launch_bounds__(256) void scale_add_fwd_kernel(const float* __restrict__ u, const float* __restrict__ x, const float* __restrict__ scale, float* __restrict__ y,
                                                            int cp4, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 s = reinterpret_cast<const f32x4*>(scale)[i % cp4];
    const f32x4 a = reinterpret_cast<const f32x4*>(u)[i], b = reinterpret_cast<const f32x4*>(x)[i];
    reinterpret_cast<f32x4*>(y)[i] = a * s + b;
  }
}

// du = dy * scale;  dx (+)= dy
__global__ __launch_bounds__(256) void scale_add_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ scale, float* __restrict__ gu, float* __restrict__ gx,
                                                            int acc_x, int cp4, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 s = reinterpret_cast<const f32x4*>(scale)[i % cp4];
    const f32x4 g = reinterpret_cast<const f32x4*>(gy)[i];
    reinterpret_cast<f32x4*>(gu)[i] = g * s;
    reinterpret_cast<f32x4*>(gx)[i] = acc_x ? reinterpret_cast<const f32x4*>(gx)[i] + g : g;
  }
}

static inline unsigned ew_blocks(size_t n4) { return (unsigned)std::min<size_t>((n4 + 255) / 256, 256 * 64); }

int launch_gelu_fwd(const float* x, float* y, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, s, x, y, n / 4);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}
int launch_gelu_bwd(const float* gy, const float* x, float* gx, int accumulate, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, s, gy, x, gx, accumulate, n / 4);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}
int launch_scale_add_fwd(const float* u, const float* x, const float* scale, float* y, int cp, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(scale_add_fwd_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, s, u, x, scale, y, cp / 4, n / 4);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}
int launch_scale_add_bwd(const float* gy, const float* scale, float* gu, float* gx, int acc_x, int cp, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(scale_add_bwd_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, s, gy, scale, gu, gx, acc_x, cp / 4, n / 4);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// Per-channel reductions over pixels: out[c] = sum_p a[p][c] * b[p][c]   (b == nullptr: sum of a)
// or, with per-pixel statistics, sum_p a[p][c] * (b[p][c] - mean[p]) * rstd[p]  (LayerNorm weight grad).
// Two stage, fixed order.  Layout of `partial`: RED_SLICES x cp.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chan_reduce_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ stats /* npix x 2 or null */,
                                                                  size_t npix, int cp, float* __restrict__ partial) {
  // thread = (pixel row r, channel quad c): float4 loads, four pixels per trip (all loads issued before the adds)
  __shared__ f32x4 red[256];
  const int sl = blockIdx.x;
  const size_t per = (npix + RED_SLICES - 1) / RED_SLICES;
  const size_t lo = (size_t)sl * per, hi = std::min(npix, lo + per);
  const int cq = cp >> 2;  // cp is a multiple of 16
  const int cw = min(cq, 256), rows = 256 / cw;
  const int c0 = threadIdx.x % cw, r = threadIdx.x / cw;
  const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
  const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
  auto term = [&](size_t p, f32x4 av, f32x4 bv) __attribute__((always_inline)) {
    if (stats) {
      const float mean = stats[2 * p], rstd = stats[2 * p + 1];
      bv = f32x4{(bv[0] - mean) * rstd, (bv[1] - mean) * rstd, (bv[2] - mean) * rstd, (bv[3] - mean) * rstd};
    }
    return av * bv;
  };
  const f32x4 one = {1.f, 1.f, 1.f, 1.f};
  for (int cb = 0; cb < cq; cb += cw) {
    const int c = cb + c0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (r < rows && c < cq) {
      size_t p = lo + r;
      const size_t st = (size_t)rows;
      for (; p + 3 * st < hi; p += 4 * st) {
        const f32x4 a0 = a4[p * cq + c], a1 = a4[(p + st) * cq + c], a2 = a4[(p + 2 * st) * cq + c], a3 = a4[(p + 3 * st) * cq + c];
        f32x4 b0 = one, b1 = one, b2 = one, b3 = one;
        if (b) b0 = b4[p * cq + c], b1 = b4[(p + st) * cq + c], b2 = b4[(p + 2 * st) * cq + c], b3 = b4[(p + 3 * st) * cq + c];
        acc += (term(p, a0, b0) + term(p + st, a1, b1)) + (term(p + 2 * st, a2, b2) + term(p + 3 * st, a3, b3));
      }
      for (; p < hi; p += st) acc += term(p, a4[p * cq + c], b ? b4[p * cq + c] : one);
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
__global__ __launch_bounds__(256) void chan_reduce_final_kernel(const float* __restrict__ partial, int cp, int c, float* __restrict__ out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  float s = 0.f;
  if (i < c)
    for (int k = lane; k < RED_SLICES; k += 64) s += partial[(size_t)k * cp + i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (i < c && lane == 0) out[i] = s;
}
int launch_chan_reduce(const float* a, const float* b, const float* stats, size_t npix, int cp, int c, float* out, float* scratch, hipStream_t s) {
  hipLaunchKernelGGL(chan_reduce_partial_kernel, dim3(RED_SLICES), dim3(256), 0, s, a, b, stats, npix, cp, scratch);
  hipLaunchKernelGGL(chan_reduce_final_kernel, dim3((c + 3) / 4), dim3(256), 0, s, scratch, cp, c, out);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}
int64_t chan_reduce_scratch_floats(int cp) { return (int64_t)RED_SLICES * cp; }

// ---------------------------------------------------------------------------------------
// LayerNorm backward (data gradient).  16 lanes per pixel as in the forward; mean / rstd are
// recomputed from the saved input and also written to `stats` (npix x 2) for the weight gradient.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ gamma,
                                                            float* __restrict__ gx, float* __restrict__ stats, int accumulate, int c, int cp, size_t npix) {
  const int sub = threadIdx.x & 15;
  const int quads = cp >> 2;
  const float inv_c = 1.0f / (float)c;
  const size_t stride = (size_t)gridDim.x * 16;
  const size_t rounds = (npix + stride - 1) / stride;
  size_t pix = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  auto red16 = [](float v) {
    v += __shfl_xor(v, 8, 16);
    v += __shfl_xor(v, 4, 16);
    v += __shfl_xor(v, 2, 16);
    v += __shfl_xor(v, 1, 16);
    return v;
  };
  for (size_t it = 0; it < rounds; ++it, pix += stride) {
    const bool live = pix < npix;
    const size_t pp = live ? pix : npix - 1;
    const float* px = x + pp * cp;
    const float* pg = gy + pp * cp;
    float s = 0.f;
    for (int q = sub; q < quads; q += 16) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(px + q * 4);
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    const float mean = red16(s) * inv_c;
    float ss = 0.f;
    for (int q = sub; q < quads; q += 16) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(px + q * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[e] - mean;
        ss += (q * 4 + e < c) ? d * d : 0.f;
      }
    }
    const float rstd = 1.0f / sqrtf(red16(ss) * inv_c + LN_EPS_T);
    float s1 = 0.f, s2 = 0.f;  // sum_c g*dy, sum_c g*dy*xhat   (gamma is zero-padded: pad channels drop out)
    for (int q = sub; q < quads; q += 16) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(px + q * 4);
      const f32x4 g = *reinterpret_cast<const f32x4*>(pg + q * 4);
      const f32x4 w = *reinterpret_cast<const f32x4*>(gamma + q * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = g[e] * w[e];
        s1 += a;
        s2 += a * ((v[e] - mean) * rstd);
      }
    }
    const float m1 = red16(s1) * inv_c, m2 = red16(s2) * inv_c;
    if (live) {
      if (sub == 0) {
        stats[2 * pix] = mean;
        stats[2 * pix + 1] = rstd;
      }
      float* o = gx + pix * cp;
      for (int q = sub; q < quads; q += 16) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(px + q * 4);
        const f32x4 g = *reinterpret_cast<const f32x4*>(pg + q * 4);
        const f32x4 w = *reinterpret_cast<const f32x4*>(gamma + q * 4);
        f32x4 r = accumulate ? *reinterpret_cast<const f32x4*>(o + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (v[e] - mean) * rstd;
          const float d = rstd * (g[e] * w[e] - m1 - xh * m2);
          r[e] += (q * 4 + e < c) ? d : 0.f;
        }
        *reinterpret_cast<f32x4*>(o + q * 4) = r;
      }
    }
  }
}
int launch_layernorm_bwd(const float* x, const float* gy, const float* gamma, float* gx, float* stats, int accumulate, int c, int cp, size_t npix, hipStream_t s) {
  const int blocks = (int)std::min<size_t>((npix + 15) / 16, 256 * 32);
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, s, x, gy, gamma, gx, stats, accumulate, c, cp, npix);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// Depthwise 7x7 weight gradient.  One thread = one channel, marching along one image row with
// the 7x7 input window in registers (7 new loads + 1 gradient load per 49 FMAs; lanes run over
// channels, so every load is coalesced).  Workgroup = 64 channels x 4 rows; partial[slice][tap][cp].
// ---------------------------------------------------------------------------------------
constexpr int DWG_SLICES = 512;
__global__ __launch_bounds__(256) void dwconv7_wgrad_partial_kernel(const float* __restrict__ x, const float* __restrict__ gy, int B, int H, int W, int cp,
                                                                    float* __restrict__ partial) {
  __shared__ float red[4][64];
  const int lane_c = threadIdx.x & 63, rsub = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (wave-uniform, and known to be: the row descriptors live in scalar registers)
  const int c = blockIdx.y * 64 + lane_c;
  const bool cok = c < cp;
  const int rows = B * H;
  float acc[49];
#pragma unroll
  for (int t = 0; t < 49; ++t) acc[t] = 0.f;
  // The 7 x 7 window lives in registers as a RING over columns: column x + d sits in slot (x + d) mod 7, so moving one pixel to the right replaces one
  // slot per kernel row and moves nothing (the shifting window of round 2 spent 42 of its ~126 vector instructions per pixel on register moves).  The x
  // loop is unrolled by seven so that every slot index is a compile-time constant.  Rows are addressed through per-row buffer descriptors (a wave = one
  // output row: wave-uniform): columns right of the image and rows outside it are out of range = zeros, no clamps or selects.
  const unsigned vc = cok ? (unsigned)c * 4u : 0x80000000u;
  const int row_bytes = W * cp * 4, px_bytes = cp * 4;
  for (int row = blockIdx.x * 4 + rsub; row < rows; row += gridDim.x * 4) {
    const int b = row / H, y = row - b * H;
    __amdgpu_buffer_rsrc_t rs[7];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      const int iy = y + ky - 3;
      const bool ok = iy >= 0 && iy < H;
      rs[ky] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + ((size_t)b * H + (ok ? iy : 0)) * W * cp), 0, ok ? row_bytes : 0, 0x00020000);
    }
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gy + (size_t)row * W * cp), 0, row_bytes, 0x00020000);
    float win[7][7];  // win[ky][slot]
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
      for (int j = 0; j < 7; ++j) win[ky][j] = 0.f;  // columns -3 .. -1 (slots 4 .. 6) stay zero until they are replaced; slot 3 is loaded at x = 0
#pragma unroll
      for (int j = 0; j < 3; ++j) win[ky][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs[ky], vc + (unsigned)(j * px_bytes), 0, 0));  // columns 0 .. 2 (beyond a narrow image: zeros)
    }
    for (int x0 = 0; x0 < W; x0 += 7) {
#pragma unroll
      for (int ph = 0; ph < 7; ++ph) {
        const int xx = x0 + ph;
        if (xx < W) {  // wave-uniform
          const unsigned vnew = vc + (unsigned)((xx + 3) * px_bytes);  // column xx + 3 -> slot (ph + 3) % 7, the one column xx - 4 held
#pragma unroll
          for (int ky = 0; ky < 7; ++ky) win[ky][(ph + 3) % 7] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs[ky], vnew, 0, 0));
          const float g = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, vc + (unsigned)(xx * px_bytes), 0, 0));
#pragma unroll
          for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
              acc[ky * 7 + kx] = fmaf(g, win[ky][(ph + kx + 4) % 7], acc[ky * 7 + kx]);  // column xx + kx - 3
              asm volatile("" : "+v"(acc[ky * 7 + kx]));  // one scalar fma per tap: paired (v_pk_fma_f32) the ring's rotating slots cost 53 register moves per pixel
            }
        }
      }
    }
  }
  // combine the four row streams of the workgroup in a fixed order
  for (int t = 0; t < 49; ++t) {
    red[rsub][lane_c] = acc[t];
    __syncthreads();
    if (rsub == 0 && cok) partial[((size_t)blockIdx.x * 49 + t) * cp + c] = (red[0][lane_c] + red[1][lane_c]) + (red[2][lane_c] + red[3][lane_c]);
    __syncthreads();
  }
}
// (thread = (tap, channel) with the channel fastest: a slice's 49 x cp table is read along its rows -- the tap-fastest order this kernel had read every
// element from a different cache line, 252 us per launch at 1.3 MB of partials; four sums in flight, fixed order)
__global__ void dwconv7_wgrad_final_kernel(const float* __restrict__ partial, int n_slices, int cp, int c, float* __restrict__ gw /* (C,1,7,7) */) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c * 49) return;
  const int t = i / c, ch = i - t * c;
  const float* src = partial + (size_t)t * cp + ch;
  const size_t st = (size_t)49 * cp;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
  for (; k + 3 < n_slices; k += 4) {
    s0 += src[(size_t)k * st];
    s1 += src[(size_t)(k + 1) * st];
    s2 += src[(size_t)(k + 2) * st];
    s3 += src[(size_t)(k + 3) * st];
  }
  for (; k < n_slices; ++k) s0 += src[(size_t)k * st];
  gw[ch * 49 + t] = (s0 + s1) + (s2 + s3);
}
static int dwg_slices(int B, int H) { return std::max(1, std::min(DWG_SLICES, (B * H + 3) / 4)); }
int launch_dwconv7_wgrad(const float* x, const float* gy, int B, int H, int W, int cp, int c, float* gw, float* scratch, hipStream_t s) {
  const int slices = dwg_slices(B, H);
  hipLaunchKernelGGL(dwconv7_wgrad_partial_kernel, dim3(slices, (cp + 63) / 64), dim3(256), 0, s, x, gy, B, H, W, cp, scratch);
  hipLaunchKernelGGL(dwconv7_wgrad_final_kernel, dim3((c * 49 + 255) / 256), dim3(256), 0, s, scratch, slices, cp, c, gw);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}
int64_t dwconv7_wgrad_scratch_floats(int B, int H, int cp) { return (int64_t)dwg_slices(B, H) * 49 * cp; }

// ---------------------------------------------------------------------------------------
// Row weight gradient on MFMA:  dW[n][k] = sum_m dY[m][n] * A[m][k]   (Linear: A row m = pixel m;
// Conv2d k2 s2: A row m = pixel (b, 2oy + dy, 2ox + dx) of the input, one launch per tap).
//   GEMM with M = n (dY channel), N = k (A channel), K = rows.  Both operands are read the way they
//   lie in memory (row-major over channels), staged per 32-row chunk as [row][128 channels] in LDS;
//   an MFMA fragment is ONE dword per lane (lane = channel, lane >> 5 = which of the two rows of the
//   k-pair): ds_read_b32 with the lanes running over channels is conflict-free.
//   Workgroup = 256 threads, tile 128 (n) x 128 (k), wave = 64 x 64; the row range is split over
//   blockIdx.y slices so the grid fills the chip; partial tiles go to a slab, a second kernel adds
//   the slices in a fixed order into the canonical gradient.
// ---------------------------------------------------------------------------------------
constexpr int RW_ROWS = 32;
// LDS layout of one operand chunk: TRANSPOSED, [128 channels][32 rows], so that the four rows a lane contracts in four consecutive MFMAs are one
// ds_read_b128 (a dword per MFMA and operand before: as many LDS instructions as MFMAs).  Row group g (4 rows = 16 B) of channel ch sits at
// ch * 32 + 4 * (g ^ ((ch >> 1) & 7)) floats: 16 consecutive channels reading one row group touch 16 distinct 16-B slots (conflict-free).
__device__ __forceinline__ int rw_slot(int ch, int g) { return ch * RW_ROWS + 4 * (g ^ ((ch >> 1) & 7)); }
// TK = 2: tile 128 (n) x 128 (k), waves 2 x 2, wave tile 64 x 64.  TK = 3: tile 128 x 96, waves 4 x 1, wave tile 32 x 96 -- for operands whose channel
// count is a multiple of 96 but not of 128 (ConvNeXt widths 96 and 192: a 128-wide tile multiplies 25 % padding).
template <int TK>
__global__ __launch_bounds__(256, 2) void row_wgrad_kernel(RowWgradArgs a) {
  constexpr int TN = TK == 2 ? 2 : 1, KTILE = TK == 2 ? 128 : 96;
  __shared__ __attribute__((aligned(16))) float sY[2][RW_ROWS * 128];
  __shared__ __attribute__((aligned(16))) float sA[2][RW_ROWS * 128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, lh = lane >> 5;
  const int n_kt = (a.kp + KTILE - 1) / KTILE;
  const int nt = blockIdx.x / n_kt, kt = blockIdx.x - nt * n_kt;
  const int slice = blockIdx.y, n_slices = gridDim.y;
  const int wn = TK == 2 ? wave >> 1 : wave, wk = TK == 2 ? wave & 1 : 0;  // wave tile: n in [wn * 32 TN, + 32 TN), k in [wk * 32 TK, + 32 TK)
  const int chunks = (a.M + RW_ROWS - 1) / RW_ROWS;
  const int per = (chunks + n_slices - 1) / n_slices;
  const int c_lo = slice * per, c_hi = min(chunks, c_lo + per);

  f32x16 acc[TN][TK];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: thread t owns the 4 x 4 block (rows 4 g .. 4 g + 3, channels 4 sq .. 4 sq + 3) of both operands, g = t & 7, sq = t >> 3: one 16-B load per
  // row (a wave's load instruction covers 8 rows x 128 B, whole cache lines), and the block leaves for LDS transposed -- register naming only -- as
  // one 16-B write per channel.  The next chunk travels global -> registers (fetch) while this chunk's MFMAs run and lands in the other LDS buffer
  // afterwards (commit): the global latency is never exposed between two chunks.
  const int sg = tid & 7, sq = tid >> 3;
  f32x4 rys[1][4], ras[1][4];
  auto fetch = [&](int chunk, f32x4 (&ry)[4], f32x4 (&ra)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * sg + i;
      const int m = chunk * RW_ROWS + row;
      const bool ok = m < a.M;
      const int mm = ok ? m : a.M - 1;
      const int cy = nt * 128 + sq * 4;
      int ck = sq * 4 < KTILE ? kt * KTILE + sq * 4 : a.kp;
      f32x4 vy = {0.f, 0.f, 0.f, 0.f}, va = {0.f, 0.f, 0.f, 0.f};
      if (ok && cy < a.np) vy = *reinterpret_cast<const f32x4*>(a.dy + (size_t)mm * a.np + cy);
      if (ok && ck < a.kp) {
        size_t pix = (size_t)mm;
        if (a.patch == 2) {  // k x k "same" conv: input pixel = output pixel + (ky - k / 2, kx - k / 2), zero outside the image
          const int x = mm % a.W, y = (mm / a.W) % a.H;
          const int yy = y + a.tap / a.ksize - (a.ksize >> 1), xx = x + a.tap % a.ksize - (a.ksize >> 1);
          const bool in = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
          pix = in ? (size_t)((long long)mm + (a.tap / a.ksize - (a.ksize >> 1)) * a.W + (a.tap % a.ksize - (a.ksize >> 1))) : (size_t)mm;
          if (!in) ck = a.kp;  // -> zeros
        } else if (a.patch == 1) {  // output pixel (b, oy, ox) -> input pixel (b, 2oy + dy, 2ox + dx)
          const int ow = a.W >> 1, oh = a.H >> 1;
          const int ox = mm % ow;
          const int r2 = mm / ow;
          const int oy = r2 % oh;
          pix = ((size_t)(r2 / oh) * a.H + 2 * oy + (a.tap >> 1)) * a.W + 2 * ox + (a.tap & 1);
        } else if (a.patch == 3) {  // 3x3 stride-2 pad-1 gather: row (b, oy, ox) on the (H/2 x W/2) grid -> pixel (2oy + ky - 1, 2ox + kx - 1) of the H x W map
          const int ow = a.W >> 1, oh = a.H >> 1;
          const int ox = mm % ow;
          const int r2 = mm / ow;
          const int oy = r2 % oh;
          const int yy = 2 * oy + a.tap / 3 - 1, xx = 2 * ox + a.tap % 3 - 1;
          const bool in = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
          pix = in ? ((size_t)(r2 / oh) * a.H + yy) * a.W + xx : 0;
          if (!in) ck = a.kp;  // -> zeros
        }
        if (ck < a.kp) va = *reinterpret_cast<const f32x4*>(a.x + pix * a.kp + ck);
      }
      ry[i] = vy;
      ra[i] = va;
    }
  };
  // Linear layers (patch 0, nearly all of the work): the rows of a slice are addressed through buffer descriptors based at the slice's first row -- a
  // per-lane byte offset fixed for the whole launch plus the chunk's offset in a scalar register, no vector arithmetic per chunk (measured: the
  // 64-bit row addressing and the range selects of the general path cost 16 % of the kernel); channels past np / kp get an offset beyond the
  // descriptor's range and read as zeros.  (Range checking is not relied on for rows: whole chunks only.)
  const bool fast = a.patch == 0 && (size_t)per * RW_ROWS * (size_t)max(a.np, a.kp) * 4 < 0x40000000ull;
  __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, 0, 0x00020000), rs_a = rs_y;
  unsigned vo_y[4], vo_a[4];
  if (fast) {
    const size_t row0 = (size_t)min(c_lo, chunks) * RW_ROWS;
    const size_t left = (size_t)a.M - min(row0, (size_t)a.M);  // rows from the slice's first row to the end of the tensor
    const size_t span = min(left, (size_t)per * RW_ROWS);
    rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dy + row0 * a.np), 0, (int)(span * a.np * 4), 0x00020000);
    rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + row0 * a.kp), 0, (int)(span * a.kp * 4), 0x00020000);
    const int cy = nt * 128 + sq * 4, ck = sq * 4 < KTILE ? kt * KTILE + sq * 4 : a.kp;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      vo_y[i] = cy < a.np ? (unsigned)((4 * sg + i) * a.np + cy) * 4u : 0x80000000u;
      vo_a[i] = ck < a.kp ? (unsigned)((4 * sg + i) * a.kp + ck) * 4u : 0x80000000u;
    }
  }
  auto fetch_fast = [&](int chunk, f32x4 (&ry)[4], f32x4 (&ra)[4]) __attribute__((always_inline)) {
    const int so_y = (chunk - c_lo) * RW_ROWS * a.np * 4, so_a = (chunk - c_lo) * RW_ROWS * a.kp * 4;  // scalar
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ry[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_y, vo_y[i], so_y, 0));
      ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, vo_a[i], so_a, 0));
    }
  };
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};  // column sums of the operand that is the layer's dY, this thread's channel quad (a.bias_sum)
  const int bias_op = a.bias_sum == 1 && kt == 0 ? 1 : (a.bias_sum == 2 && nt == 0 ? 2 : 0);  // workgroup-uniform
  auto commit = [&](int buf, const f32x4 (&ry)[4], const f32x4 (&ra)[4]) __attribute__((always_inline)) {
    if (bias_op == 1) bsum += (ry[0] + ry[1]) + (ry[2] + ry[3]);
    if (bias_op == 2) bsum += (ra[0] + ra[1]) + (ra[2] + ra[3]);
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int o = rw_slot(4 * sq + cc, sg);
      *reinterpret_cast<f32x4*>(&sY[buf][o]) = f32x4{ry[0][cc], ry[1][cc], ry[2][cc], ry[3][cc]};
      *reinterpret_cast<f32x4*>(&sA[buf][o]) = f32x4{ra[0][cc], ra[1][cc], ra[2][cc], ra[3][cc]};
    }
  };

  // fragment of MFMA step j of row-group pair q: element j of the 16 bytes at row group 2 q + lh, i.e. K rows {8 q + j, 8 q + 4 + j} over the two
  // lane halves -- the same pairing in both operands, which is all the contraction needs
  int chy[TN], cha[TK];
#pragma unroll
  for (int i = 0; i < TN; ++i) chy[i] = (wn * TN + i) * 32 + lx;
#pragma unroll
  for (int j = 0; j < TK; ++j) cha[j] = (wk * TK + j) * 32 + lx;
  auto multiply = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < RW_ROWS / 8; ++q) {
      f32x4 fy[TN], fa[TK];
#pragma unroll
      for (int i = 0; i < TN; ++i) fy[i] = *reinterpret_cast<const f32x4*>(&sY[buf][rw_slot(chy[i], 2 * q + lh)]);
#pragma unroll
      for (int j = 0; j < TK; ++j) fa[j] = *reinterpret_cast<const f32x4*>(&sA[buf][rw_slot(cha[j], 2 * q + lh)]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fy[i][e], fa[j][e], acc[i][j], 0, 0, 0);
    }
  };
  // Whole chunks of a Linear layer, [c_lo, c_full): the next chunk travels global -> registers while this chunk's MFMAs run and lands in the other LDS
  // buffer afterwards.  (Fetching two chunks ahead with a second register set measured 10 % slower: 256 VGPRs, spills.)
  const int c_full = fast ? max(c_lo, min(c_hi, a.M / RW_ROWS)) : c_lo;
  if (c_lo < c_full) {
    fetch_fast(c_lo, rys[0], ras[0]);
    commit(0, rys[0], ras[0]);
    __syncthreads();
    for (int ch = c_lo; ch < c_full; ch += 2) {
      if (ch + 1 < c_full) fetch_fast(ch + 1, rys[0], ras[0]);
      multiply(0);
      if (ch + 1 < c_full) commit(1, rys[0], ras[0]);
      __syncthreads();
      if (ch + 1 >= c_full) break;
      if (ch + 2 < c_full) fetch_fast(ch + 2, rys[0], ras[0]);
      multiply(1);
      if (ch + 2 < c_full) commit(0, rys[0], ras[0]);
      __syncthreads();
    }
  }
  // Everything else (gathered rows of the patch modes, a partial last chunk): the general fetch, one chunk ahead
  if (c_full < c_hi) {
    fetch(c_full, rys[0], ras[0]);
    commit(0, rys[0], ras[0]);
    __syncthreads();
    for (int ch = c_full; ch < c_hi; ++ch) {
      const int buf = (ch - c_full) & 1;
      if (ch + 1 < c_hi) fetch(ch + 1, rys[0], ras[0]);
      if (buf)
        multiply(1);
      else
        multiply(0);
      if (ch + 1 < c_hi) commit(buf ^ 1, rys[0], ras[0]);
      __syncthreads();
    }
  }
  if (bias_op) {  // the eight row groups of a channel quad meet in LDS (free after the loop's last barrier) in a fixed order
    f32x4* red = reinterpret_cast<f32x4*>(&sY[0][0]);
    red[tid] = bsum;
    __syncthreads();
    if (sg == 0) {
      f32x4 t = red[tid];
#pragma unroll
      for (int k = 1; k < 8; ++k) t += red[tid + k];
      const int tiles = bias_op == 1 ? (int)gridDim.x / n_kt : n_kt, tile = bias_op == 1 ? nt : kt;
      *reinterpret_cast<f32x4*>(a.bias_slab + ((size_t)slice * tiles + tile) * 128 + 4 * sq) = t;
    }
  }
  // slab[slice][block][128 n][128 k];  D: row(n) = (r&3) + 8*(r>>2) + 4*lh, col(k) = lx
  float* slab = a.slab + ((size_t)slice * gridDim.x + blockIdx.x) * (128 * 128);
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[(size_t)((wn * TN + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 128 + (wk * TK + j) * 32 + lx] = acc[i][j][r];
}

// grad[(n * k_total + k) * taps + tap] = sum_slices slab[...]; 1024 threads = 256 elements x 4 slice parts (a thread
// walks a quarter of the slices, four loads in flight), the parts meet in LDS in a fixed order
// (ktile: width of a block's k range, 128 or 96; swapped: the GEMM ran with the operands exchanged -- slab rows are k, columns n)
__global__ __launch_bounds__(1024) void row_wgrad_reduce_kernel(const float* __restrict__ slab, int n_slices, int n_blocks, int n_kt, int n, int k, int k_total, int k_off,
                                                                int taps, int tap, int ktile, int swapped, float* __restrict__ grad) {
  __shared__ float red[3 * 256];
  const int e = threadIdx.x & 255, part = threadIdx.x >> 8;
  const int i = blockIdx.x * 256 + e;
  const bool ok = i < n * k;
  const int ii = ok ? i : 0;
  const int kk = ii % k, nn = ii / k;
  const int row = swapped ? kk : nn, col = swapped ? nn : kk;  // position in the GEMM that ran: row = its n axis (128-tiles), col = its k axis (ktile-tiles)
  const int blk = (row >> 7) * n_kt + col / ktile;
  const float* src = slab + (size_t)blk * (128 * 128) + (size_t)(row & 127) * 128 + col % ktile;
  const size_t st = (size_t)n_blocks * (128 * 128);
  float s = 0.f;
  int sl = part;
  for (; sl + 12 < n_slices; sl += 16) {
    const float v0 = src[(size_t)sl * st], v1 = src[(size_t)(sl + 4) * st], v2 = src[(size_t)(sl + 8) * st], v3 = src[(size_t)(sl + 12) * st];
    s += (v0 + v1) + (v2 + v3);
  }
  for (; sl < n_slices; sl += 4) s += src[(size_t)sl * st];
  if (part) red[(part - 1) * 256 + e] = s;
  __syncthreads();
  if (part || !ok) return;
  grad[((size_t)nn * k_total + k_off + kk) * taps + tap] = (s + red[e]) + (red[256 + e] + red[512 + e]);
}

// gb[c] = sum over slices of bias_slab[slice][tile c / width][c % width] (width: channels of a tile that are real, 128 or 96); one thread per channel
__global__ __launch_bounds__(256) void row_wgrad_bias_reduce_kernel(const float* __restrict__ slab, int n_slices, int tiles, int width, int n, float* __restrict__ gb) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  const float* src = slab + (size_t)(c / width) * 128 + c % width;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
  for (; k + 3 < n_slices; k += 4) {
    s0 += src[(size_t)k * tiles * 128];
    s1 += src[(size_t)(k + 1) * tiles * 128];
    s2 += src[(size_t)(k + 2) * tiles * 128];
    s3 += src[(size_t)(k + 3) * tiles * 128];
  }
  for (; k < n_slices; ++k) s0 += src[(size_t)k * tiles * 128];
  gb[c] = (s0 + s1) + (s2 + s3);
}

// Tile plan of one launch: 96-wide k tiles when the k operand's (padded) channel count is a multiple of 96 but not of 128; a Linear layer whose
// n operand is the one like that runs with the operands exchanged (the patch modes gather rows of x only, they are never exchanged).
struct RwPlan {
  int swapped, ktile, n_nt, n_kt;
};
static RwPlan rw_plan(int np, int kp, int patch) {
  auto is96 = [](int c) { return c % 128 != 0 && c % 96 == 0; };
  RwPlan p;
  p.swapped = patch == 0 && !is96(kp) && is96(np) ? 1 : 0;
  const int rn = p.swapped ? kp : np, rk = p.swapped ? np : kp;
  p.ktile = is96(rk) ? 96 : 128;
  p.n_nt = (rn + 127) / 128;
  p.n_kt = (rk + p.ktile - 1) / p.ktile;
  return p;
}
static int rw_slices(int M, int blocks) {
  const int chunks = (M + RW_ROWS - 1) / RW_ROWS;
  const int want = (1024 + blocks - 1) / blocks;  // ~2 workgroups per CU x 2 rounds
  return std::max(1, std::min(chunks, want));
}
int64_t row_wgrad_slab_floats(int M, int n, int k) {
  // (the larger of the plans a patch mode and a Linear layer would take for these widths)
  int64_t best = 0;
  for (int patch = 0; patch < 2; ++patch) {
    const RwPlan p = rw_plan(pad16(n), pad16(k), patch);
    const int blocks = p.n_nt * p.n_kt;
    const int64_t slices = rw_slices(M, blocks);
    best = std::max(best, slices * blocks * 128 * 128 + slices * std::max(p.n_nt, p.n_kt) * 128);  // + the bias-sum slab of a Linear layer
  }
  return best;
}
int launch_row_wgrad(const RowWgradArgs& a0, int n, int k, int taps, float* grad, hipStream_t s) { return launch_row_wgrad_part(a0, n, k, k, 0, taps, grad, s); }

// k_total / k_off: the weight's input-channel axis is k_total wide and this operand covers [k_off, k_off + k) (concat sources)
int launch_row_wgrad_part(const RowWgradArgs& a0, int n, int k, int k_total, int k_off, int taps, float* grad, hipStream_t s) {
  RowWgradArgs a = a0;
  const RwPlan p = rw_plan(a.np, a.kp, a.patch);
  if (p.swapped) {
    std::swap(a.dy, a.x);
    std::swap(a.np, a.kp);
  }
  const int blocks = p.n_nt * p.n_kt;
  const int slices = rw_slices(a.M, blocks);
  const bool with_bias = a.gb && a.patch == 0;
  if (with_bias) {
    a.bias_sum = p.swapped ? 2 : 1;
    a.bias_slab = a.slab + (size_t)slices * blocks * 128 * 128;
  } else {
    a.bias_sum = 0;
  }
  if (p.ktile == 96)
    hipLaunchKernelGGL(row_wgrad_kernel<3>, dim3(blocks, slices), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(row_wgrad_kernel<2>, dim3(blocks, slices), dim3(256), 0, s, a);
  hipLaunchKernelGGL(row_wgrad_reduce_kernel, dim3((n * k + 255) / 256), dim3(1024), 0, s, a.slab, slices, blocks, p.n_kt, n, k, k_total, k_off, taps, a.tap, p.ktile, p.swapped, grad);
  if (with_bias)  // dY is the n operand (128-wide tiles) or, exchanged, the k operand (ktile-wide tiles)
    hipLaunchKernelGGL(row_wgrad_bias_reduce_kernel, dim3((a.gb_n + 255) / 256), dim3(256), 0, s, a.bias_slab, slices, p.swapped ? p.n_kt : p.n_nt, p.swapped ? p.ktile : 128,
                       a.gb_n, a.gb);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// Patch-stem / first k x k conv weight gradient: dW[co][ci][ky][kx] = sum_{b,oy,ox} dy[b,oy,ox,co] * img[b,ci,oy*s+ky-pad,ox*s+kx-pad].
// The image patches are written once as rows [pixel][ci*k*k + tap] (im2col, <= 48 floats per pixel, padded
// to a multiple of 16) and the gradient is one row-wgrad GEMM over them.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patch_im2col_kernel(const void* __restrict__ img, int dtype, int B, int cin, int H, int W, int OH, int OW, int k, int stride, int pad,
                                                           int kp, float* __restrict__ out) {
  const int kk = k * k, n_item = cin * kk;
  const size_t total = (size_t)B * OH * OW * kp;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int item = (int)(idx % kp);
    const size_t p = idx / kp;
    float v = 0.f;
    if (item < n_item) {
      const int ci = item / kk, tap = item - ci * kk;
      const int ky = tap / k, kx = tap - ky * k;
      const int ox = (int)(p % OW);
      const size_t r = p / OW;
      const int oy = (int)(r % OH), b = (int)(r / OH);
      const int yy = oy * stride + ky - pad, xx = ox * stride + kx - pad;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const size_t o = (((size_t)b * cin + ci) * H + yy) * W + xx;
        if (dtype == 0)
          v = (float)reinterpret_cast<const uint8_t*>(img)[o] / 255.0f;
        else {
          v = reinterpret_cast<const float*>(img)[o];
          if (dtype == 2) v = v / 255.0f;
        }
      }
    }
    out[idx] = v;
  }
}
int launch_patch_stem_wgrad(const void* img, int dtype, const float* dy, int B, int cin, int H, int W, int OH, int OW, int k, int stride, int pad, int coutp, int cout, float* gw,
                            float* scratch, hipStream_t s) {
  const int n_item = cin * k * k, kp = pad16(n_item);
  const size_t npix = (size_t)B * OH * OW;
  float* patches = scratch;
  float* slab = scratch + align_up((int64_t)npix * kp, 64);
  const size_t total = npix * kp;
  hipLaunchKernelGGL(patch_im2col_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 256 * 64)), dim3(256), 0, s, img, dtype, B, cin, H, W, OH, OW, k, stride, pad, kp, patches);
  PH_HIP_CHECK(hipGetLastError());
  RowWgradArgs a{};
  a.dy = dy;
  a.x = patches;
  a.slab = slab;
  a.np = coutp;
  a.kp = kp;
  a.M = (int)npix;
  return launch_row_wgrad(a, cout, n_item, 1, gw, s);  // canonical (cout, cin, k, k) == [co][ci*k*k + tap]
}
int64_t patch_stem_wgrad_scratch_floats(int cin, int cout, int k, int64_t npix) {
  const int kp = pad16(cin * k * k);
  return align_up(npix * kp, 64) + row_wgrad_slab_floats((int)npix, cout, cin * k * k);
}

// ---------------------------------------------------------------------------------------
// Class-vector head loss (lightning_modules.py:2655-2677): nn.CrossEntropyLoss()(p, y) where p is ALREADY the
// head's softmax output (the reference applies log_softmax on top of it) and y holds class probabilities:
//   loss = mean_b( -sum_c y[b,c] * log_softmax(p[b,:])[c] ).
// Gradient wrt the head's pre-softmax logits z:  dp = (softmax(p) - y) * w / B,  dz = p * (dp - sum_c dp[c] p[c]).
// One thread per row; rows are few (crops of a batch) and classes a handful.
// ---------------------------------------------------------------------------------------
__global__ void class_ce_kernel(const float* __restrict__ p, const float* __restrict__ y, int B, int C, float weight, float* __restrict__ dz, float* __restrict__ loss_out) {
  __shared__ float red[256];
  float lsum = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float* pr = p + (size_t)b * C;
    const float* yr = y + (size_t)b * C;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, pr[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(pr[c] - m);
    const float lse = m + logf(se);
    float dot = 0.f, l = 0.f;
    for (int c = 0; c < C; ++c) {
      const float q = expf(pr[c] - lse);
      l -= yr[c] * (pr[c] - lse);
      const float dp = (q - yr[c]) * weight / (float)B;
      dot += dp * pr[c];
    }
    for (int c = 0; c < C; ++c) {
      const float q = expf(pr[c] - lse);
      const float dp = (q - yr[c]) * weight / (float)B;
      dz[(size_t)b * C + c] = pr[c] * (dp - dot);
    }
    lsum += l;
  }
  red[threadIdx.x] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (int)blockDim.x; ++i) t += red[i];
    loss_out[0] = t / (float)B;
  }
}
int launch_class_ce(const float* p, const float* y, int B, int C, float weight, float* dz, float* loss_out, hipStream_t s) {
  hipLaunchKernelGGL(class_ce_kernel, dim3(1), dim3(256), 0, s, p, y, B, C, weight, dz, loss_out);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// Global max pool backward: the gradient of channel c goes to the FIRST position holding the maximum (ATen's argmax).
__global__ __launch_bounds__(256) void global_maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, int HW, int cp, int accumulate, float* __restrict__ gx) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < cp; c += 256) {
    const float* xp = x + (size_t)b * HW * cp + c;
    float m = -INFINITY;
    int arg = 0;
    for (int p = 0; p < HW; ++p) {
      const float v = xp[(size_t)p * cp];
      if (v > m) {
        m = v;
        arg = p;
      }
    }
    const float g = gy[(size_t)b * cp + c];
    float* gp = gx + (size_t)b * HW * cp + c;
    for (int p = 0; p < HW; ++p) {
      const float add = p == arg ? g : 0.f;
      gp[(size_t)p * cp] = accumulate ? gp[(size_t)p * cp] + add : add;
    }
  }
}
int launch_global_maxpool_bwd(const float* x, const float* gy, int B, int HW, int cp, int accumulate, float* gx, hipStream_t s) {
  hipLaunchKernelGGL(global_maxpool_bwd_kernel, dim3(B), dim3(256), 0, s, x, gy, HW, cp, accumulate, gx);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

}  // namespace ph

extern "C" {

// Diagnostic: time (and spot-check) the row weight-gradient GEMM on synthetic operands.  Not a product path; tools/row_wgrad_bench.py.
int ph_debug_row_wgrad_bench(int32_t M, int32_t n, int32_t k, int32_t iters, float* ms_out, float* max_rel_err_out) {
  using namespace ph;
  PH_REQUIRE(ms_out && max_rel_err_out && iters > 0 && M > 0 && n > 0 && k > 0 && n % 16 == 0 && k % 16 == 0, "ph_debug_row_wgrad_bench: bad arguments");
  const size_t n_dy = (size_t)M * n, n_x = (size_t)M * k, n_slab = (size_t)row_wgrad_slab_floats(M, n, k), n_g = (size_t)n * k;
  float *dy = nullptr, *x = nullptr, *slab = nullptr, *g = nullptr;
  PH_HIP_CHECK(hipMalloc(&dy, n_dy * 4));
  PH_HIP_CHECK(hipMalloc(&x, n_x * 4));
  PH_HIP_CHECK(hipMalloc(&slab, n_slab * 4));
  PH_HIP_CHECK(hipMalloc(&g, n_g * 4));
  std::vector<float> hdy(n_dy), hx(n_x);
  unsigned st = 777u;
  auto rnd = [&]() {
    st = st * 1664525u + 1013904223u;
    return ((st >> 8) & 0xffff) / 65536.0f - 0.5f;
  };
  for (auto& v : hdy) v = rnd();
  for (auto& v : hx) v = rnd();
  PH_HIP_CHECK(hipMemcpy(dy, hdy.data(), n_dy * 4, hipMemcpyHostToDevice));
  PH_HIP_CHECK(hipMemcpy(x, hx.data(), n_x * 4, hipMemcpyHostToDevice));
  RowWgradArgs a{};
  a.dy = dy;
  a.x = x;
  a.slab = slab;
  a.np = n;
  a.kp = k;
  a.M = M;
  hipEvent_t e0, e1;
  PH_HIP_CHECK(hipEventCreate(&e0));
  PH_HIP_CHECK(hipEventCreate(&e1));
  int rc = PH_OK;
  for (int i = 0; i < 2 && rc == PH_OK; ++i) rc = launch_row_wgrad(a, n, k, 1, g, nullptr);
  PH_HIP_CHECK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters && rc == PH_OK; ++i) rc = launch_row_wgrad(a, n, k, 1, g, nullptr);
  PH_HIP_CHECK(hipEventRecord(e1, nullptr));
  PH_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  PH_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / iters;
  std::vector<float> hg(n_g);
  PH_HIP_CHECK(hipMemcpy(hg.data(), g, n_g * 4, hipMemcpyDeviceToHost));
  double worst = 0.0, scale = 0.0;
  for (int t = 0; t < 64; ++t) {
    const int nn = (int)(((unsigned)t * 2654435761u) % (unsigned)n), kk = (int)(((unsigned)t * 40503u + 17u) % (unsigned)k);
    double ref = 0.0;
    for (int m = 0; m < M; ++m) ref += (double)hdy[(size_t)m * n + nn] * (double)hx[(size_t)m * k + kk];
    worst = std::max(worst, std::fabs(ref - (double)hg[(size_t)nn * k + kk]));
    scale = std::max(scale, std::fabs(ref));
  }
  *max_rel_err_out = (float)(worst / std::max(scale, 1e-30));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(dy);
  (void)hipFree(x);
  (void)hipFree(slab);
  (void)hipFree(g);
  return rc;
}

}  // extern "C"
