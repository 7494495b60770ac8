// ConvNeXt encoder kernels for gfx950 (MI355X, CDNA4).
//
// Reference semantics (paths relative to talmolab/sleap-nn; CNBlock / LayerNorm2d /
// Conv2dNormActivation are torchvision's, instantiated at architectures/convnext.py:67-110):
//   stem ............ Conv2d(k = stem_patch_kernel, stride = stem_patch_stride, padding = 1, bias)
//                     + LayerNorm2d(eps 1e-6)                                  convnext.py:73-84
//   CNBlock ......... dwconv 7x7 (pad 3, groups = C, bias) -> LayerNorm(C, eps 1e-6) ->
//                     Linear(C, 4C) -> GELU (erf) -> Linear(4C, C) -> * layer_scale -> + input
//   downsample ...... LayerNorm2d(eps 1e-6) + Conv2d(k 2, stride 2, bias)      convnext.py:101-110
//
// Layout is the same NHWC-with-padded-channels fp32 as the UNet kernels, so the permutes of
// the reference vanish: LayerNorm is a reduction over the innermost axis and both Linear layers
// and the 2x2/stride-2 convolution are plain GEMMs over pixel rows.  The GEMMs run on
// v_mfma_f32_32x32x2_f32 (exact fp32 products) with LDS-DMA staging (gemm_mfma_dma_kernel).
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "common.h"
#include "device_math.h"
#include "net_kernels.h"

namespace ph {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float LN_EPS = 1e-6f;

__device__ __forceinline__ void decode_block_1d(int tiles, int nt_count, int* tile, int* ntile) {
  // same XCD-aware dealing as the 3x3 convolution: the N tiles of one M tile get ids 8 apart,
  // i.e. they run on the same XCD at the same time and share the A rows through its L2.
  const int id = blockIdx.x;
  if ((tiles & 7) == 0) {
    const int xcd = id & 7, j = id >> 3;
    *ntile = j % nt_count;
    *tile = (j / nt_count) * 8 + xcd;
  } else {
    *ntile = id % nt_count;
    *tile = id / nt_count;
  }
}

// ---------------------------------------------------------------------------------------
// K20: patch stem.  One thread = one output pixel x 4 output channels, straight from the NCHW
// image (uint8 / float, normalisation fused as in K0).  K = k*k*Cin <= 48: VALU work, HBM-bound
// on the NHWC store.  Weights: [tap][ci][Cp].
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patch_stem_kernel(PatchStemArgs a) {
  const int groups = a.coutp >> 2;
  const size_t total = (size_t)a.B * a.OH * a.OW * groups;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int gq = (int)(idx % groups);
    size_t p = idx / groups;
    const int ox = (int)(p % a.OW);
    p /= a.OW;
    const int oy = (int)(p % a.OH);
    const int b = (int)(p / a.OH);
    f32x4 acc = *reinterpret_cast<const f32x4*>(a.bias + gq * 4);
    for (int ci = 0; ci < a.cin; ++ci) {
      const size_t plane = ((size_t)b * a.cin + ci) * a.H * a.W;
      for (int ky = 0; ky < a.k; ++ky) {
        const int yy = oy * a.stride + ky - 1;
        if (yy < 0 || yy >= a.H) continue;
        for (int kx = 0; kx < a.k; ++kx) {
          const int xx = ox * a.stride + kx - 1;
          if (xx < 0 || xx >= a.W) continue;
          float v;
          if (a.dtype == 0)
            v = (float)reinterpret_cast<const uint8_t*>(a.src)[plane + (size_t)yy * a.W + xx] / 255.0f;
          else {
            v = reinterpret_cast<const float*>(a.src)[plane + (size_t)yy * a.W + xx];
            if (a.dtype == 2) v = v / 255.0f;
          }
          const f32x4 w = *reinterpret_cast<const f32x4*>(a.w + ((size_t)(ky * a.k + kx) * a.cin + ci) * a.coutp + gq * 4);
          acc += v * w;
        }
      }
    }
    *reinterpret_cast<f32x4*>(a.dst + (((size_t)b * a.OH + oy) * a.OW + ox) * a.coutp + gq * 4) = acc;
  }
}

// The common stem (one input channel, K x K taps, <= 128 padded output channels) without the generic kernel's per-tap global loads,
// divisions and branches: the normalised image patch of an 8 x 32 output tile goes to LDS once, a thread keeps the K x K weight quads of
// ITS four output channels in registers, and a pixel's channel quads are the 32 lanes of a half wave -- so an optional LayerNorm2d over
// the channels (the op that follows the stem conv) is two 5-step shuffles away and the un-normalised tensor never reaches HBM.
template <int K>
__global__ __launch_bounds__(256) void patch_stem_c1_kernel(PatchStemArgs a) {
  constexpr int TOH = 8, TOW = 32;
  extern __shared__ float sPatch[];  // ((TOH - 1) * stride + K) x ((TOW - 1) * stride + K)
  const int PH_ = (TOH - 1) * a.stride + K, PW = (TOW - 1) * a.stride + K;
  const int tid = threadIdx.x, gq = tid & 31, ps = tid >> 5;
  const int groups = a.coutp >> 2;
  const bool act = gq < groups;
  f32x4 wreg[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) wreg[t] = act ? *reinterpret_cast<const f32x4*>(a.w + (size_t)t * a.coutp + gq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};  // [tap][ci = 1][Cp]
  const f32x4 b0 = act ? *reinterpret_cast<const f32x4*>(a.bias + gq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 g = {0.f, 0.f, 0.f, 0.f}, bt = {0.f, 0.f, 0.f, 0.f};
  if (a.ln_gamma && act) {
    g = *reinterpret_cast<const f32x4*>(a.ln_gamma + gq * 4);
    bt = *reinterpret_cast<const f32x4*>(a.ln_beta + gq * 4);
  }
  const float inv_c = a.ln_c > 0 ? 1.0f / (float)a.ln_c : 0.f;
  const int tiles_x = (a.OW + TOW - 1) / TOW, tiles_y = (a.OH + TOH - 1) / TOH;
  for (int tile = blockIdx.x; tile < tiles_x * tiles_y * a.B; tile += gridDim.x) {
    int t = tile;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int ox0 = tx * TOW, oy0 = ty * TOH;
    __syncthreads();  // the previous tile's reads are done
    for (int i = tid; i < PH_ * PW; i += 256) {
      const int iy = i / PW, ix = i - iy * PW;
      const int gy = oy0 * a.stride + iy - 1, gx = ox0 * a.stride + ix - 1;  // padding 1 (convnext.py:73-84)
      float v = 0.f;
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
        const size_t o = ((size_t)b * a.H + gy) * a.W + gx;
        if (a.dtype == 0)
          v = (float)reinterpret_cast<const uint8_t*>(a.src)[o] / 255.0f;
        else {
          v = reinterpret_cast<const float*>(a.src)[o];
          if (a.dtype == 2) v = v / 255.0f;
        }
      }
      sPatch[i] = v;
    }
    __syncthreads();
    for (int j = 0; j < TOW; ++j) {  // half wave ps walks row ps of the tile
      const int oy = oy0 + ps, ox = ox0 + j;
      f32x4 acc = b0;
#pragma unroll
      for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) acc += sPatch[(ps * a.stride + ky) * PW + j * a.stride + kx] * wreg[ky * K + kx];
      if (a.ln_gamma) {  // LayerNorm2d over the channels: two-pass moments as layernorm_kernel (pad channels are exact zeros / excluded)
        float sm = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) sm += __shfl_xor(sm, d, 32);
        const float mean = sm * inv_c;
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dd = acc[e] - mean;
          ss += (gq * 4 + e < a.ln_c) ? dd * dd : 0.f;
        }
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) ss += __shfl_xor(ss, d, 32);
        const float rstd = 1.0f / sqrtf(ss * inv_c + LN_EPS);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = (acc[e] - mean) * rstd * g[e] + bt[e];
      }
      if (act && oy < a.OH && ox < a.OW) *reinterpret_cast<f32x4*>(a.dst + (((size_t)b * a.OH + oy) * a.OW + ox) * a.coutp + gq * 4) = acc;
    }
  }
}

int launch_patch_stem(const PatchStemArgs& a, hipStream_t s) {
  if (a.cin == 1 && a.coutp <= 128 && (a.k == 4 || a.k == 2 || a.k == 3) && a.stride >= 1 && a.stride <= 4) {
    const int tiles = ((a.OW + 31) / 32) * ((a.OH + 7) / 8) * a.B;
    const dim3 grid(std::min(tiles, 256 * 8));
    const size_t lds = (size_t)(7 * a.stride + a.k) * (31 * a.stride + a.k) * sizeof(float);
    if (a.k == 4)
      hipLaunchKernelGGL(patch_stem_c1_kernel<4>, grid, dim3(256), lds, s, a);
    else if (a.k == 3)
      hipLaunchKernelGGL(patch_stem_c1_kernel<3>, grid, dim3(256), lds, s, a);
    else
      hipLaunchKernelGGL(patch_stem_c1_kernel<2>, grid, dim3(256), lds, s, a);
    PH_HIP_CHECK(hipGetLastError());
    return PH_OK;
  }
  PH_REQUIRE(!a.ln_gamma, "the fused LayerNorm needs the one-channel patch stem kernel");
  const size_t total = (size_t)a.B * a.OH * a.OW * (a.coutp / 4);
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(patch_stem_kernel, dim3(blocks), dim3(256), 0, s, a);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// K21: depthwise 7x7 "same" convolution, NHWC.  One thread = 4 channels x a strip of 8 output
// pixels of one row: per kernel row it loads the 14 input float4s of the strip once and reuses
// each for up to 7 outputs (1568 FMAs per 147 float4 loads); lanes run along channels, so every
// load/store instruction is a contiguous run of Cp floats per pixel.  Out-of-image taps are
// read from a clamped address and zeroed with a select (no divergent branches around loads).
// Weights: [tap][Cp].
// ---------------------------------------------------------------------------------------
constexpr int DW_STRIP = 8, DW_ROWS = 2;

// One thread = 4 channels x a block of DW_ROWS x DW_STRIP output pixels.  It walks the DW_ROWS + 6 input rows
// once, keeps the 14 input float4s of the current row in registers and applies them to every output row they
// contribute to (1,568 float4 FMAs per 140 input + 196 weight loads; the one-row version needed 4 x 147 loads).
__global__ __launch_bounds__(256) void dwconv7_kernel(DwConvArgs a) {
  const int groups = a.cp >> 2;
  const int strips = (a.W + DW_STRIP - 1) / DW_STRIP;
  const int rblocks = (a.H + DW_ROWS - 1) / DW_ROWS;
  const size_t total = (size_t)a.B * rblocks * strips * groups;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int gq = (int)(idx % groups);
    size_t p = idx / groups;
    const int st = (int)(p % strips);
    p /= strips;
    const int rb = (int)(p % rblocks);
    const int b = (int)(p / rblocks);
    const int x0 = st * DW_STRIP, y0 = rb * DW_ROWS;
    const f32x4 bias = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + gq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc[DW_ROWS][DW_STRIP];
#pragma unroll
    for (int r = 0; r < DW_ROWS; ++r)
#pragma unroll
      for (int o = 0; o < DW_STRIP; ++o) acc[r][o] = bias;
    const float* wq = a.w + gq * 4;
#pragma unroll
    for (int ir = 0; ir < DW_ROWS + 6; ++ir) {
      const int iy = y0 + ir - 3;
      if (iy < 0 || iy >= a.H) continue;
      const float* row = a.src + ((size_t)(b * a.H + iy) * a.W) * a.cp + gq * 4;
      f32x4 in[DW_STRIP + 6];
#pragma unroll
      for (int i = 0; i < DW_STRIP + 6; ++i) {
        const int ix = x0 + i - 3;
        const int cx = min(max(ix, 0), a.W - 1);
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + (size_t)cx * a.cp);
        const bool ok = ix >= 0 && ix < a.W;
        in[i][0] = ok ? v[0] : 0.f;
        in[i][1] = ok ? v[1] : 0.f;
        in[i][2] = ok ? v[2] : 0.f;
        in[i][3] = ok ? v[3] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < DW_ROWS; ++r) {
        const int ky = ir - r;  // this input row is kernel row ky of output row r
        if (ky < 0 || ky > 6) continue;
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(wq + (size_t)(ky * 7 + kx) * a.cp);
#pragma unroll
          for (int o = 0; o < DW_STRIP; ++o) acc[r][o] += in[o + kx] * w;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < DW_ROWS; ++r) {
      if (y0 + r >= a.H) continue;
      float* drow = a.dst + ((size_t)(b * a.H + y0 + r) * a.W) * a.cp + gq * 4;
#pragma unroll
      for (int o = 0; o < DW_STRIP; ++o)
        if (x0 + o < a.W) {
          f32x4* dp = reinterpret_cast<f32x4*>(drow + (size_t)(x0 + o) * a.cp);
          *dp = a.accumulate ? *dp + acc[r][o] : acc[r][o];
        }
    }
  }
}

// dwconv7 + LayerNorm in one pass (CNBlock's first two ops; the depthwise output never reaches HBM).  A workgroup owns whole pixels:
// 256 / (Cp / 4) strips of DW_ROWS x DW_STRIP pixels x all channel quads; after the depthwise accumulation (as dwconv7_kernel) the
// per-pixel moments are reduced through LDS -- 16 lanes per pixel, two passes (mean, then the biased variance about it, pad channels
// excluded) exactly like layernorm_kernel -- and the normalised, affine-transformed values are stored.
__global__ __launch_bounds__(256) void dwconv7_ln_kernel(DwConvArgs a) {
  constexpr int NPIX = DW_ROWS * DW_STRIP;  // pixels per strip block
  __shared__ float red[16 * 256];           // [local strip][pixel][channel quad]: spb * NPIX * groups <= 16 * 256
  __shared__ float stat[2 * 16 * 16];       // mean, rstd per (local strip, pixel): spb * NPIX <= 10 * 16
  const int groups = a.cp >> 2;
  const int spb = 256 / groups;             // strips per workgroup (>= 1: cp <= 1024)
  const int strips = (a.W + DW_STRIP - 1) / DW_STRIP;
  const int sgroups = (strips + spb - 1) / spb;
  const int rblocks = (a.H + DW_ROWS - 1) / DW_ROWS;
  const int tid = threadIdx.x;
  const int ls = tid / groups, gq = tid - ls * groups;
  int t = blockIdx.x;
  const int sg = t % sgroups;
  t /= sgroups;
  const int rb = t % rblocks;
  const int b = t / rblocks;
  const int st = sg * spb + ls;
  const bool active = ls < spb && st < strips;
  const int x0 = st * DW_STRIP, y0 = rb * DW_ROWS;
  f32x4 acc[DW_ROWS][DW_STRIP];
  {
    const f32x4 bias = (active && a.bias) ? *reinterpret_cast<const f32x4*>(a.bias + gq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < DW_ROWS; ++r)
#pragma unroll
      for (int o = 0; o < DW_STRIP; ++o) acc[r][o] = bias;
  }
  if (active) {
    const float* wq = a.w + gq * 4;
#pragma unroll
    for (int ir = 0; ir < DW_ROWS + 6; ++ir) {
      const int iy = y0 + ir - 3;
      if (iy < 0 || iy >= a.H) continue;
      const float* row = a.src + ((size_t)(b * a.H + iy) * a.W) * a.cp + gq * 4;
      f32x4 in[DW_STRIP + 6];
#pragma unroll
      for (int i = 0; i < DW_STRIP + 6; ++i) {
        const int ix = x0 + i - 3;
        const int cx = min(max(ix, 0), a.W - 1);
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + (size_t)cx * a.cp);
        const bool ok = ix >= 0 && ix < a.W;
        in[i][0] = ok ? v[0] : 0.f;
        in[i][1] = ok ? v[1] : 0.f;
        in[i][2] = ok ? v[2] : 0.f;
        in[i][3] = ok ? v[3] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < DW_ROWS; ++r) {
        const int ky = ir - r;  // this input row is kernel row ky of output row r
        if (ky < 0 || ky > 6) continue;
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(wq + (size_t)(ky * 7 + kx) * a.cp);
#pragma unroll
          for (int o = 0; o < DW_STRIP; ++o) acc[r][o] += in[o + kx] * w;
        }
      }
    }
  }
  // ---- moments per pixel: item = (local strip, pixel), 16 lanes per item
  const float inv_c = 1.0f / (float)a.ln_c;
  const int n_items = spb * NPIX;
  auto reduce_items = [&](float* out /* stat + 0 or + 256 */, bool second) __attribute__((always_inline)) {
    for (int item = tid >> 4; item < n_items; item += 16) {
      const int sub = tid & 15;
      float s = 0.f;
      for (int g = sub; g < groups; g += 16) s += red[item * groups + g];
      s += __shfl_xor(s, 8, 16);
      s += __shfl_xor(s, 4, 16);
      s += __shfl_xor(s, 2, 16);
      s += __shfl_xor(s, 1, 16);
      if (sub == 0) out[item] = second ? 1.0f / sqrtf(s * inv_c + LN_EPS) : s * inv_c;
    }
  };
  if (active) {
#pragma unroll
    for (int r = 0; r < DW_ROWS; ++r)
#pragma unroll
      for (int o = 0; o < DW_STRIP; ++o) red[(ls * NPIX + r * DW_STRIP + o) * groups + gq] = (acc[r][o][0] + acc[r][o][1]) + (acc[r][o][2] + acc[r][o][3]);  // pad channels are exact zeros
  }
  __syncthreads();
  reduce_items(stat, false);
  __syncthreads();
  if (active) {
#pragma unroll
    for (int r = 0; r < DW_ROWS; ++r)
#pragma unroll
      for (int o = 0; o < DW_STRIP; ++o) {
        const float mean = stat[ls * NPIX + r * DW_STRIP + o];
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = acc[r][o][e] - mean;
          ss += (gq * 4 + e < a.ln_c) ? d * d : 0.f;
        }
        red[(ls * NPIX + r * DW_STRIP + o) * groups + gq] = ss;
      }
  }
  __syncthreads();
  reduce_items(stat + 256, true);
  __syncthreads();
  if (active) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(a.ln_gamma + gq * 4), bt = *reinterpret_cast<const f32x4*>(a.ln_beta + gq * 4);
#pragma unroll
    for (int r = 0; r < DW_ROWS; ++r) {
      if (y0 + r >= a.H) continue;
      float* drow = a.dst + ((size_t)(b * a.H + y0 + r) * a.W) * a.cp + gq * 4;
#pragma unroll
      for (int o = 0; o < DW_STRIP; ++o)
        if (x0 + o < a.W) {
          const float mean = stat[ls * NPIX + r * DW_STRIP + o], rstd = stat[256 + ls * NPIX + r * DW_STRIP + o];
          f32x4 y;
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = (acc[r][o][e] - mean) * rstd * g[e] + bt[e];
          *reinterpret_cast<f32x4*>(drow + (size_t)(x0 + o) * a.cp) = y;
        }
    }
  }
}

int launch_dwconv7(const DwConvArgs& a, hipStream_t s) {
  if (a.ln_gamma) {
    PH_REQUIRE(a.ln_beta && a.ln_c > 0 && !a.accumulate && a.cp <= 1024, "fused dwconv + LayerNorm: bad arguments");
    const int groups = a.cp >> 2, spb = 256 / groups, strips = (a.W + DW_STRIP - 1) / DW_STRIP;
    const size_t blocks = (size_t)a.B * ((a.H + DW_ROWS - 1) / DW_ROWS) * ((strips + spb - 1) / spb);
    hipLaunchKernelGGL(dwconv7_ln_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    PH_HIP_CHECK(hipGetLastError());
    return PH_OK;
  }
  const size_t total = (size_t)a.B * ((a.H + DW_ROWS - 1) / DW_ROWS) * ((a.W + DW_STRIP - 1) / DW_STRIP) * (a.cp / 4);
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 64);
  hipLaunchKernelGGL(dwconv7_kernel, dim3(blocks), dim3(256), 0, s, a);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// K22: LayerNorm over the channel axis of an NHWC tensor (LayerNorm2d and CNBlock's
// nn.LayerNorm are the same reduction in this layout).  16 lanes per pixel; two-pass
// mean / biased variance like ATen's RowwiseMoments result, y = (x - mean) * rstd * g + b.
// Pad channels hold zeros on input (so they do not disturb the sum), are excluded from the
// variance, and are written as zeros (gamma / beta are zero-padded).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ src, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ dst, int c, int cp, size_t npix) {
  const int sub = threadIdx.x & 15;
  const int quads = cp >> 2;
  const float inv_c = 1.0f / (float)c;
  const size_t stride = (size_t)gridDim.x * 16;
  const size_t rounds = (npix + stride - 1) / stride;
  size_t pix = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  for (size_t it = 0; it < rounds; ++it, pix += stride) {
    const bool live = pix < npix;
    const float* p = src + (live ? pix : npix - 1) * cp;
    float s = 0.f;
    for (int q = sub; q < quads; q += 16) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + q * 4);
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    s += __shfl_xor(s, 8, 16);
    s += __shfl_xor(s, 4, 16);
    s += __shfl_xor(s, 2, 16);
    s += __shfl_xor(s, 1, 16);
    const float mean = s * inv_c;
    float ss = 0.f;
    for (int q = sub; q < quads; q += 16) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + q * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[e] - mean;
        ss += (q * 4 + e < c) ? d * d : 0.f;
      }
    }
    ss += __shfl_xor(ss, 8, 16);
    ss += __shfl_xor(ss, 4, 16);
    ss += __shfl_xor(ss, 2, 16);
    ss += __shfl_xor(ss, 1, 16);
    const float rstd = 1.0f / sqrtf(ss * inv_c + LN_EPS);
    if (live) {
      float* o = dst + pix * cp;
      for (int q = sub; q < quads; q += 16) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + q * 4);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + q * 4);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + q * 4);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (v[e] - mean) * rstd * g[e] + bt[e];
        *reinterpret_cast<f32x4*>(o + q * 4) = r;
      }
    }
  }
}

int launch_layernorm(const float* src, const float* gamma, const float* beta, float* dst, int c, int cp, size_t npix, hipStream_t s) {
  const int blocks = (int)std::min<size_t>((npix + 15) / 16, 256 * 32);
  hipLaunchKernelGGL(layernorm_kernel, dim3(blocks), dim3(256), 0, s, src, gamma, beta, dst, c, cp, npix);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// K23: row GEMM on MFMA with LDS-DMA staging: dst[m, n] = epilogue(sum_k A[m, k] * Wt[n, k] + bias[n]).
//   A rows are pixels of one or two NHWC tensors (a channel concat is two K panels); K runs
//   over sources x 16-channel slices x taps, the taps innermost so that the re-read of a row's
//   neighbours is one 16-KiB step away (L2 / L1 hits):
//     mode 0  Linear / 1x1 conv ....... 1 tap, row m = pixel m
//     mode 1  Conv2d k2 s2 ............ 4 taps (dy, dx), row m = output pixel (b, oy, ox)
//     mode 2  Conv2d 3x3 "same" ....... 9 taps, out-of-image taps read a zero page.  Used for
//             feature maps too small to fill the 16x32-pixel tiles of the halo-tiled kernel K1d.
//     mode 3  one output phase (oy & 1, ox & 1) of ConvTranspose2d(k3, s2, p1, op1): 1, 2, 2 or 4 taps over the INPUT
//             grid, rows = input pixels, results scattered to output pixel (2y + py, 2x + px) (encoder_decoder.py:439-461);
//             the four phases together do 9 taps per input pixel = 2.25 per output pixel, no zero-stuffed tensor.
//     mode 4  Conv2d 3x3 stride 2 pad 1 (the data gradient of that transposed conv): 9 taps gathered from a 2H x 2W map.
//     mode 5  Conv2d k x k "same", odd k <= 9 (kernel_size 5 / 7 backbones, the 7x7 convs of a UNet stem block,
//             encoder_decoder.py:144-225): k^2 taps, validity computed per tap from the row's (y, x).
//   512 threads = 8 waves, tile 256 rows x BN columns (BN = 32 * NT_TOTAL); waves are arranged
//   (8 / WN) x WN and each owns WN 32-row tiles x (NT_TOTAL / WN) 32-column tiles.
//   Pipeline: a ring of three LDS stages of 16 K-values each (256 x 16 A slice + BN x 16 weight
//   slice, quad-major 1-KiB pieces written by global_load_lds_dwordx4).  The DMA for stage s+2
//   is issued while stage s is multiplied; the only synchronisation per stage is
//   s_waitcnt vmcnt(3) (this wave's three pieces of stage s+1 have landed; the three of stage
//   s+2 stay in flight) + s_barrier.  72 KiB of LDS at BN = 128, so two workgroups share a CU
//   and one's epilogue overlaps the other's main loop.
//   Epilogues: bias; + ReLU; + GELU (erf form); layer_scale * (acc + bias) + residual.
// ---------------------------------------------------------------------------------------
// One (M tile, N tile) per workgroup: kept for the 9-tap convolution form, where the persistent kernel's two tile plans
// push the register file over the edge (256 VGPRs + spills: 107 -> 102 TFLOP/s measured).
template <int MODE, int MT, int NTW, int WM, int WN, int NSTAGE, int MINW>
__global__ __launch_bounds__(WM* WN * 64, MINW) void gemm_mfma_dma_tile_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WAVES = WM * WN;
  constexpr int TM = WM * MT * 32;
  constexpr int BN = WN * NTW * 32;
  constexpr int A_PIECES = TM / 8, B_PIECES = BN / 8;  // 1-KiB pieces of 8 rows x 32 channels per stage
  constexpr int A_SLOTS = (A_PIECES + WAVES - 1) / WAVES, B_SLOTS = (B_PIECES + WAVES - 1) / WAVES;
  constexpr int NDMA = A_SLOTS + B_SLOTS;  // DMA instructions per wave per stage
  constexpr int STAGE_FLOATS = (A_PIECES + B_PIECES) * 256;
  static_assert((NSTAGE - 2) * NDMA < 64 && NSTAGE >= 2 && WAVES % 2 == 0, "vmcnt field overflow / odd wave count");
  constexpr int NTAPS_C = MODE == 0 ? 1 : (MODE == 1 ? 4 : 9);
  // mode 3 with all_phases: the four output phases of a transposed conv in ONE launch (blockIdx.y = phase: its tap count, weight pack and output offset) -- at small batches
  // the four phase launches were four launch floors (the reference's fixture bottom-up model at 320 x 560 x 4: 76 + 54 us of a 454-us forward)
  const int phase = (MODE == 3 && a.all_phases) ? (int)blockIdx.y : a.out_tap;
  const float* const wpack = (MODE == 3 && a.all_phases) ? a.wpack_ph[phase] : a.wpack;
  const int NTAPS = MODE == 3 ? (1 + (phase >> 1)) * (1 + (phase & 1)) : (MODE == 5 ? a.ksize * a.ksize : NTAPS_C);  // mode 3: 1, 2 or 4 taps by output phase; mode 5: k x k

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int mtiles = (a.M + TM - 1) / TM;
  int t, ntile;
  decode_block_1d(mtiles, (a.coutp + BN - 1) / BN, &t, &ntile);
  const int m0 = t * TM;

  f32x16 acc[MT][NTW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int nstages = ((a.c0p + 31) / 32 + (a.c1p + 31) / 32) * NTAPS;  // 32-channel slices x taps, per source
  const float* wbase = wpack + (size_t)ntile * nstages * (B_PIECES * 256);

  // ---- DMA plan.  A 1-KiB piece = 8 rows x 128 B (32 channels): every row contributes one whole
  // 128-B cache line per stage (64-B half lines cost twice the L2 -> L1 traffic: each half is
  // evicted from the 32-KiB L1 before its other half is wanted).  Lane L of piece p moves row
  // p * 8 + (L & 7), channel quad (L >> 3) ^ (p & 1): the XOR alternates the two 128-B halves of
  // the 256-B LDS bank row between consecutive pieces, so a ds_read_b128 over 16 consecutive rows
  // touches 16 distinct 16-B slots (conflict-free, no padding).  Everything inside the K loop is
  // branch-free (selects only): a branch would split the scheduling region and serialise address
  // arithmetic, DMA issue and MFMAs.
  const int dr = lane & 7, dquad = (lane >> 3) ^ (wave & 1);  // p & 1 == wave & 1 (WAVES is even)
  unsigned long long a_base[2][A_SLOTS];  // byte address of (row's pixel, channel quad dquad) in each source
  unsigned a_mask[A_SLOTS];               // mode 5: (y << 16) | x of the row's pixel instead of a tap mask (k x k taps do not fit 32 bits)
#pragma unroll
  for (int s = 0; s < A_SLOTS; ++s) {
    const int row = min(m0 + min(wave + WAVES * s, A_PIECES - 1) * 8 + dr, a.M - 1);
    unsigned mask = 0x1ffu;
    long long pix = row;
    if (MODE == 5) mask = ((unsigned)((row / a.W) % a.H) << 16) | (unsigned)(row % a.W);
    if (MODE == 1) {
      const int ow = a.W >> 1, oh = a.H >> 1;
      const int ox = row % ow;
      const int r2 = row / ow;
      const int oy = r2 % oh;
      pix = ((long long)(r2 / oh) * a.H + 2 * oy) * a.W + 2 * ox;
    } else if (MODE == 2) {
      const int x = row % a.W;
      const int y = (row / a.W) % a.H;
      mask = 0;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        mask |= (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? (1u << tap) : 0u;
      }
    } else if (MODE == 3) {  // transposed-conv phase: taps reach (y + dy, x + dx), dy, dx in {0, 1}; beyond the image = zero
      const int x = row % a.W;
      const int y = (row / a.W) % a.H;
      const int py = phase >> 1, px = phase & 1;
      mask = 0;
#pragma unroll
      for (int tap = 0; tap < 4; ++tap) {
        const int ty = px ? tap >> 1 : tap, tx = px ? tap & 1 : 0;
        const int dy = py ? 1 - ty : 0, dx = px ? 1 - tx : 0;
        mask |= (y + dy < a.H && x + dx < a.W) ? (1u << tap) : 0u;
      }
    } else if (MODE == 4) {  // 3x3 stride-2 pad-1 gather over an a.H x a.W map; row = (b, oy, ox) on the (a.H / 2) x (a.W / 2) grid
      const int ow = a.W >> 1, oh = a.H >> 1;
      const int ox = row % ow;
      const int r2 = row / ow;
      const int oy = r2 % oh;
      pix = ((long long)(r2 / oh) * a.H + 2 * oy) * a.W + 2 * ox;
      mask = 0;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int yy = 2 * oy + tap / 3 - 1, xx = 2 * ox + tap % 3 - 1;
        mask |= (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? (1u << tap) : 0u;
      }
    }
    a_base[0][s] = (unsigned long long)(a.src0 + pix * a.c0p + dquad * 4);
    a_base[1][s] = (unsigned long long)((a.src1 ? a.src1 : a.src0) + pix * a.c1p + dquad * 4);
    a_mask[s] = mask;
  }
  const unsigned long long zero_addr = (unsigned long long)(a.zeros + (lane >> 3) * 4);
  const unsigned long long w_lane = (unsigned long long)(wbase + lane * 4);
  // fetch cursor (wave-uniform scalars): index of the next stage, its tap, channel offset, source
  int f_idx = 0, f_tap = 0, f_coff = 0, f_src = 0;
  int f_ty = 0, f_tx = 0;  // mode 5: the tap's kernel row / column
  auto issue_stage = [&](float* buf) {
    const int cp = f_src ? a.c1p : a.c0p;
    int toff;  // pixel offset of the tap
    if (MODE == 0) {
      toff = 0;
    } else if (MODE == 5) {
      toff = (f_ty - (a.ksize >> 1)) * a.W + (f_tx - (a.ksize >> 1));
    } else if (MODE == 1) {
      toff = (f_tap >> 1) * a.W + (f_tap & 1);
    } else if (MODE == 3) {
      const int py = phase >> 1, px = phase & 1;
      const int ty = px ? f_tap >> 1 : f_tap, tx = px ? f_tap & 1 : 0;
      toff = (py ? 1 - ty : 0) * a.W + (px ? 1 - tx : 0);
    } else {
      const int ty = (f_tap * 11) >> 5;  // f_tap / 3 for 0..8
      toff = (ty - 1) * a.W + (f_tap - 3 * ty - 1);
    }
    const long long soff = ((long long)toff * cp + f_coff) * 4;  // bytes, wave-uniform
    // a slice may be half empty (Cp is a multiple of 16, not 32): quads past Cp come from the zero page
    const unsigned live = ((f_idx < nstages) & (f_coff + dquad * 4 < cp)) ? 1u : 0u;
#pragma unroll
    for (int k = 0; k < A_SLOTS; ++k) {
      const unsigned long long real = (f_src ? a_base[1][k] : a_base[0][k]) + (unsigned long long)soff;
      unsigned tap_ok;
      if (MODE == 5) {
        const int yy = (int)(a_mask[k] >> 16) + f_ty - (a.ksize >> 1), xx = (int)(a_mask[k] & 0xffffu) + f_tx - (a.ksize >> 1);
        tap_ok = (((unsigned)yy < (unsigned)a.H) & ((unsigned)xx < (unsigned)a.W)) ? 1u : 0u;
      } else {
        tap_ok = (a_mask[k] >> f_tap) & 1u;
      }
      const unsigned long long sel = 0ull - (unsigned long long)(live & tap_ok);
      const unsigned long long g = (real & sel) | (zero_addr & ~sel);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(buf + min(wave + WAVES * k, A_PIECES - 1) * 256), 16, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < B_SLOTS; ++k) {
      const int pb = min(wave + WAVES * k, B_PIECES - 1);
      const unsigned long long g = w_lane + ((unsigned long long)(min(f_idx, nstages - 1) * B_PIECES + pb) << 10);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(buf + (A_PIECES + pb) * 256), 16, 0, 0);
    }
    // advance (integer arithmetic only): taps innermost, then 32-channel slices, then the second source
    f_idx += 1;
    const int wrap = (f_tap + 1 == NTAPS) ? 1 : 0;
    if (MODE == 5) {
      const int xw = (f_tx + 1 == a.ksize) ? 1 : 0;
      f_tx = (f_tx + 1) * (1 - xw);
      f_ty = (f_ty + xw) * (1 - wrap);
    }
    f_tap = (f_tap + 1) * (1 - wrap);
    f_coff += 32 * wrap;
    const int sw = wrap & (f_coff >= cp ? 1 : 0) & (f_src == 0 ? 1 : 0) & (a.c1p > 0 ? 1 : 0);
    f_coff *= (1 - sw);
    f_src |= sw;
  };

  // ---- fragment read offsets (floats, stage-relative); step g adds g * 64 (two quads)
  const int lx = lane & 31, lh = lane >> 5;
  const int fsw = (lh ^ (lx >> 3)) & 1;
  int offA[MT], offB[NTW];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int row = (wm * MT + m) * 32 + lx;
    offA[m] = (row >> 3) * 256 + fsw * 32 + (row & 7) * 4;
  }
#pragma unroll
  for (int n = 0; n < NTW; ++n) {
    const int col = (wn * NTW + n) * 32 + lx;
    offB[n] = (A_PIECES + (col >> 3)) * 256 + fsw * 32 + (col & 7) * 4;
  }

  // ---- pipeline.  Ring of NSTAGE = 3 stages; ONE barrier per stage, in its middle:
  //   steps 0,1 of stage s | vmcnt(0) + barrier | issue DMA of stage s+2 | steps 2,3 of stage s
  // The barrier proves (a) stage s+1 has landed for every wave (its pieces were issued a whole
  // stage earlier, so the wait is free) and (b) every wave is done reading stage s-1, whose buffer
  // the DMA issued right after it refills.  Nothing synchronises at the stage boundary itself, so
  // the first fragments of stage s+1 are prefetched during the last step of stage s and the MFMA
  // stream never restarts from an empty pipe.
  static_assert(NSTAGE == 3, "the mid-stage barrier schedule needs a ring of three");
  f32x4 af[2][MT], bf[2][NTW];
  auto load_frags = [&](const float* buf, int step, int fb) {
#pragma unroll
    for (int m = 0; m < MT; ++m) af[fb][m] = *reinterpret_cast<const f32x4*>(buf + step * 64 + offA[m]);
#pragma unroll
    for (int n = 0; n < NTW; ++n) bf[fb][n] = *reinterpret_cast<const f32x4*>(buf + step * 64 + offB[n]);
  };
  auto mfma_step = [&](int fb) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[fb][n][j], af[fb][m][j], acc[m][n], 0, 0, 0);  // transposed tile: D[channel][pixel]
  };
  auto sched_step = [&](bool with_dma) {
    __builtin_amdgcn_sched_group_barrier(0x100, MT + NTW, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, MT * NTW, 0);
    if (with_dma) {
      __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, NDMA, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 3 * MT * NTW, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // The two waves that share a SIMD must not issue their DMA pieces at the same time: an LDS-DMA
  // wave-instruction occupies its wave for 60-185 cycles (MI355X guide, per-instruction table), six
  // of them ~1,000 -- if both partners do that right after the barrier the MFMA pipe idles for that
  // long in every stage (measured: 9,670 instead of 8,192 cycles per stage).  So half of the waves
  // ("late") issue one step after the other half; while one partner feeds the DMA engine the
  // other one feeds the MFMA pipe.
  auto stage = [&](const float* cur, const float* nxt, float* fill, auto late) {
    load_frags(cur, 1, 1);
    mfma_step(0);
    sched_step(false);
    load_frags(cur, 2, 0);
    mfma_step(1);
    sched_step(false);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): only the pieces of the next stage are outstanding, issued most of a stage ago
    __builtin_amdgcn_s_barrier();
    if (!decltype(late)::value) issue_stage(fill);
    load_frags(cur, 3, 1);
    mfma_step(0);
    sched_step(!decltype(late)::value);
    if (decltype(late)::value) issue_stage(fill);
    load_frags(nxt, 0, 0);
    mfma_step(1);
    sched_step(decltype(late)::value);
  };

  float* b0 = lds;
  float* b1 = lds + STAGE_FLOATS;
  float* b2 = lds + 2 * STAGE_FLOATS;
  issue_stage(b0);
  issue_stage(b1);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __builtin_amdgcn_s_barrier();
  load_frags(b0, 0, 0);
  auto k_loop = [&](auto late) {
    for (int st = 0; st < nstages; st += 3) {
      stage(b0, b1, b2, late);
      if (st + 1 < nstages) stage(b1, b2, b0, late);
      if (st + 2 < nstages) stage(b2, b0, b1, late);
    }
  };
  // a.late_split: 0 = waves 4-7 (and 12-15) are late: waves i, i+4, i+8, ... share SIMD i % 4 (measured); 1 = odd waves; 2 = nobody
  const bool is_late = WAVES >= 8 && (a.late_split == 0 ? ((wave >> 2) & 1) != 0 : (a.late_split == 1 ? (wave & 1) != 0 : false));
  if (is_late)  // wave-uniform; both paths execute the same number of barriers
    k_loop(std::true_type{});
  else
    k_loop(std::false_type{});
  __builtin_amdgcn_s_waitcnt(0x0F70);  // drain the (dummy) tail fetches before the LDS is released

  // ---- epilogue.  The product is accumulated transposed (weights as the MFMA's A operand), so in
  // the C/D map of v_mfma_f32_32x32x2_f32 a lane owns ONE pixel (lane & 31) and its registers run
  // over output channels (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5): each aligned register quad is
  // four consecutive channels -> bias / scale / residual loads and the stores are 16-B vectors,
  // four store instructions per 32x32 tile instead of sixteen.
  const bool interior = (m0 + TM <= a.M) && ((ntile + 1) * BN <= a.coutp);
#pragma unroll
  for (int n = 0; n < NTW; ++n) {
    const int cbase = ntile * BN + (wn * NTW + n) * 32 + 4 * lh;  // + 8 * q
    f32x4 bias4[4], scale4[4], shift4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bias4[q] = *reinterpret_cast<const f32x4*>(a.bias + cbase + 8 * q);
      scale4[q] = a.scale ? *reinterpret_cast<const f32x4*>(a.scale + cbase + 8 * q) : f32x4{1.f, 1.f, 1.f, 1.f};
      shift4[q] = a.shift ? *reinterpret_cast<const f32x4*>(a.shift + cbase + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int row = m0 + (wm * MT + m) * 32 + lx;
      size_t opix = (size_t)min(row, a.M - 1);
      if (a.out_patch) {
        const int ow = a.out_W >> 1, oh = a.out_H >> 1;
        const int rr = (int)opix;
        const int ox = rr % ow;
        const int r2 = rr / ow;
        const int oy = r2 % oh;
        opix = ((size_t)(r2 / oh) * a.out_H + 2 * oy + (phase >> 1)) * a.out_W + 2 * ox + (phase & 1);
      }
      const size_t rofs = opix * a.coutp;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = acc[m][n][4 * q + e] + bias4[q][e];
          if (a.affine_first) x = x * scale4[q][e] + shift4[q][e];  // folded BatchNorm (eval): before the activation
          if (a.act == 1) x = fmaxf(x, 0.f);
          if (a.act == 2) x = gelu_f(x);
          if (a.act == 4) x = x / (1.f + expf(-x));                 // SiLU
          v[e] = a.affine_first ? x : x * scale4[q][e];
        }
        const int col = cbase + 8 * q;
        if (interior) {
          if (a.residual) v += *reinterpret_cast<const f32x4*>(a.residual + rofs + col);
          *reinterpret_cast<f32x4*>(a.dst + rofs + col) = v;
        } else if (row < a.M && col < a.coutp) {  // coutp is a multiple of 16: a quad is inside or outside as a whole
          if (a.residual) v += *reinterpret_cast<const f32x4*>(a.residual + rofs + col);
          *reinterpret_cast<f32x4*>(a.dst + rofs + col) = v;
        }
      }
    }
  }
}

// EPI = 1: the training epilogues (act 3, dst_pre) are compiled in; the inference kernels (EPI = 0) do not carry them
template <int MODE, int MT, int NTW, int WM, int WN, int NSTAGE, int MINW, int EPI = 0>
__global__ __launch_bounds__(WM* WN * 64, MINW) void gemm_mfma_dma_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WAVES = WM * WN;
  constexpr int TM = WM * MT * 32;
  constexpr int BN = WN * NTW * 32;
  constexpr int A_PIECES = TM / 8, B_PIECES = BN / 8;  // 1-KiB pieces of 8 rows x 32 channels per stage
  constexpr int A_SLOTS = (A_PIECES + WAVES - 1) / WAVES, B_SLOTS = (B_PIECES + WAVES - 1) / WAVES;
  constexpr int NDMA = A_SLOTS + B_SLOTS;  // DMA instructions per wave per stage
  constexpr int STAGE_FLOATS = (A_PIECES + B_PIECES) * 256;
  static_assert(NSTAGE == 3 && WAVES % 2 == 0, "the mid-stage barrier schedule needs a ring of three and an even wave count");
  constexpr int NTAPS = MODE == 0 ? 1 : (MODE == 1 ? 4 : 9);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int mtiles = (a.M + TM - 1) / TM;
  const int ntc = (a.coutp + BN - 1) / BN;
  const int total = mtiles * ntc;  // (M tile, N tile) pairs; this workgroup takes blockIdx.x, + gridDim.x, ...
  const int nstages = ((a.c0p + 31) / 32 + (a.c1p + 31) / 32) * NTAPS;  // 32-channel slices x taps, per source
  const int nse = max(nstages, 2);  // a one-stage tile is padded with a dummy stage so the fetch never runs two tiles ahead

  // ---- DMA plan.  A 1-KiB piece = 8 rows x 128 B (32 channels): every row contributes one whole
  // 128-B cache line per stage (64-B half lines cost twice the L2 -> L1 traffic: each half is
  // evicted from the 32-KiB L1 before its other half is wanted).  Lane L of piece p moves row
  // p * 8 + (L & 7), channel quad (L >> 3) ^ (p & 1): the XOR alternates the two 128-B halves of
  // the 256-B LDS bank row between consecutive pieces, so a ds_read_b128 over 16 consecutive rows
  // touches 16 distinct 16-B slots (conflict-free, no padding).  Everything inside the K loop is
  // branch-free (selects only): a branch would split the scheduling region and serialise address
  // arithmetic, DMA issue and MFMAs.
  const int dr = lane & 7, dquad = (lane >> 3) ^ (wave & 1);  // p & 1 == wave & 1 (WAVES is even)
  // The transfers go through buffer descriptors (as conv3x3_wino2d_kernel's): a row's address is a 32-bit offset from a per-tile
  // pixel base that is fixed when the tile is set up (one per concat source: the row pitch differs), the stage's tap / channel-slice
  // offset rides in the descriptor's (scalar) base, and masked rows (taps outside the image, quads past Cp, dummy stages) get an
  // out-of-range offset -- the hardware's range check returns the zeros.  A stage issues its pieces with ~2 vector instructions per
  // piece instead of ~10 (64-bit address arithmetic and selects), and unlike global_load_lds these loads do not make the
  // compiler's LDS-read waits lgkmcnt(0).
  constexpr unsigned OOB = 0xFFFFFF00u, RANGE = 0x80000000u;
  struct Plan {
    unsigned a_off0[A_SLOTS], a_off1[A_SLOTS];  // byte offset of the slot's row from the tile's pixel base, per source (+ the lane's quad)
    unsigned a_mask[A_SLOTS];
    long long base_pix;  // wave-uniform
    int m0, ntile;
  };
  auto setup = [&](int vid, Plan& P) __attribute__((always_inline)) {
    int t, ntile;  // XCD-aware dealing over virtual workgroup ids (see decode_block_1d)
    if ((mtiles & 7) == 0) {
      const int xcd = vid & 7, j = vid >> 3;
      ntile = j % ntc;
      t = (j / ntc) * 8 + xcd;
    } else {
      ntile = vid % ntc;
      t = vid / ntc;
    }
    P.m0 = t * TM;
    P.ntile = ntile;
    auto pix_of = [&](int row) -> long long {
      if (MODE == 1) {
        const int ow = a.W >> 1, oh = a.H >> 1;
        const int ox = row % ow;
        const int r2 = row / ow;
        const int oy = r2 % oh;
        return ((long long)(r2 / oh) * a.H + 2 * oy) * a.W + 2 * ox;
      }
      return row;
    };
    // every row of the tile lies at or after its first row's pixel; the 3x3 taps reach one row + one pixel back
    P.base_pix = pix_of(min(P.m0, a.M - 1)) - (MODE == 2 ? a.W + 1 : 0);
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) {
      const int row = min(P.m0 + min(wave + WAVES * s, A_PIECES - 1) * 8 + dr, a.M - 1);
      unsigned mask = 0x1ffu;
      if (MODE == 2) {
        const int x = row % a.W;
        const int y = (row / a.W) % a.H;
        mask = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
          mask |= (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? (1u << tap) : 0u;
        }
      }
      const unsigned rel = (unsigned)(pix_of(row) - P.base_pix);
      P.a_off0[s] = rel * (unsigned)(a.c0p * 4) + (unsigned)dquad * 16u;
      P.a_off1[s] = rel * (unsigned)(a.c1p * 4) + (unsigned)dquad * 16u;
      P.a_mask[s] = mask;
    }
  };
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpack), 0, (int)((unsigned)ntc * (unsigned)nstages * (unsigned)(B_PIECES * 1024)), 0x00020000);

  // PERSISTENT workgroup: it walks its tiles back to back and the stage stream (ring of three LDS stages,
  // fetch two stages ahead) simply continues into the next tile, so a tile starts with its first stages already
  // in LDS and the epilogue's stores overlap the next tile's fetch.  Pc = plan of the tile being multiplied,
  // Pn = plan of the one after it; f_next says which of the two the fetch cursor is in.
  Plan Pc, Pn;
  int vid = blockIdx.x, nvid = blockIdx.x + gridDim.x;
  int next_ok = nvid < total ? 1 : 0;
  setup(vid, Pc);
  setup(next_ok ? nvid : vid, Pn);
  // fetch cursor (wave-uniform scalars): stage index inside its tile, tap, channel offset, source, tile selector
  int f_idx = 0, f_tap = 0, f_coff = 0, f_src = 0, f_next = 0;
  auto issue_stage = [&](float* buf) __attribute__((always_inline)) {
    const int cp = f_src ? a.c1p : a.c0p;
    int toff;  // pixel offset of the tap
    if (MODE == 0) {
      toff = 0;
    } else if (MODE == 1) {
      toff = (f_tap >> 1) * a.W + (f_tap & 1);
    } else {
      const int ty = (f_tap * 11) >> 5;  // f_tap / 3 for 0..8
      toff = (ty - 1) * a.W + (f_tap - 3 * ty - 1);
    }
    // descriptor base (wave-uniform): the tile's pixel base moved by the tap, at this stage's channel slice of the source
    const long long bpix = (f_next ? Pn.base_pix : Pc.base_pix) + toff;
    const float* sb = (f_src ? a.src1 : a.src0) + bpix * cp + f_coff;
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sb), 0, (int)RANGE, 0x00020000);
    // dummy fetches (padding stage, or no tile left) and quads past Cp (a slice may be half empty: Cp is a
    // multiple of 16, not 32) read out of range = zeros
    const bool tile_ok = f_next ? next_ok != 0 : true;
    const bool live = (f_idx < nstages) && tile_ok && (f_coff + dquad * 4 < cp);
#pragma unroll
    for (int k = 0; k < A_SLOTS; ++k) {
      const unsigned o0 = f_next ? Pn.a_off0[k] : Pc.a_off0[k], o1 = f_next ? Pn.a_off1[k] : Pc.a_off1[k];
      const unsigned mk = f_next ? Pn.a_mask[k] : Pc.a_mask[k];
      const bool ok = MODE == 2 ? (live && ((mk >> f_tap) & 1u)) : live;
      const unsigned vo = ok ? (f_src ? o1 : o0) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (__attribute__((address_space(3))) void*)(buf + min(wave + WAVES * k, A_PIECES - 1) * 256), 16, vo, 0, 0, 0);
    }
    const int wn_tile = f_next ? Pn.ntile : Pc.ntile;
#pragma unroll
    for (int k = 0; k < B_SLOTS; ++k) {
      const int pb = min(wave + WAVES * k, B_PIECES - 1);
      const int so = ((wn_tile * nstages + min(f_idx, nstages - 1)) * B_PIECES + pb) << 10;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(buf + (A_PIECES + pb) * 256), 16, (unsigned)lane * 16u, so, 0, 0);
    }
    // advance (integer arithmetic only): taps innermost, then 32-channel slices, then the second source, then the next tile
    f_idx += 1;
    const int wrap = (f_tap + 1 == NTAPS) ? 1 : 0;
    f_tap = (f_tap + 1) * (1 - wrap);
    f_coff += 32 * wrap;
    const int sw = wrap & (f_coff >= cp ? 1 : 0) & (f_src == 0 ? 1 : 0) & (a.c1p > 0 ? 1 : 0);
    f_coff *= (1 - sw);
    f_src |= sw;
    const int tw = (f_idx == nse) ? 1 : 0;  // tile finished: the cursor moves on to the next tile's stage 0
    f_idx *= (1 - tw);
    f_tap *= (1 - tw);
    f_coff *= (1 - tw);
    f_src *= (1 - tw);
    f_next += tw;
  };

  // ---- fragment read offsets (floats, stage-relative); step g adds g * 64 (two quads)
  const int lx = lane & 31, lh = lane >> 5;
  const int fsw = (lh ^ (lx >> 3)) & 1;
  int offA[MT], offB[NTW];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int row = (wm * MT + m) * 32 + lx;
    offA[m] = (row >> 3) * 256 + fsw * 32 + (row & 7) * 4;
  }
#pragma unroll
  for (int n = 0; n < NTW; ++n) {
    const int col = (wn * NTW + n) * 32 + lx;
    offB[n] = (A_PIECES + (col >> 3)) * 256 + fsw * 32 + (col & 7) * 4;
  }

  f32x16 acc[MT][NTW];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  };
  zero_acc();

  // ---- pipeline.  Ring of three stages; ONE barrier per stage, in its middle:
  //   steps 0,1 of stage s | vmcnt(0) + barrier | issue DMA of stage s+2 | steps 2,3 of stage s
  // The barrier proves (a) stage s+1 has landed for every wave (its pieces were issued a whole
  // stage earlier, so the wait is free) and (b) every wave is done reading stage s-1, whose buffer
  // the DMA issued right after it refills.  Nothing synchronises at the stage boundary itself, so
  // the first fragments of stage s+1 are prefetched during the last step of stage s and the MFMA
  // stream never restarts from an empty pipe.
  f32x4 af[2][MT], bf[2][NTW];
  auto load_frags = [&](const float* buf, int step, int fb) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < MT; ++m) af[fb][m] = *reinterpret_cast<const f32x4*>(buf + step * 64 + offA[m]);
#pragma unroll
    for (int n = 0; n < NTW; ++n) bf[fb][n] = *reinterpret_cast<const f32x4*>(buf + step * 64 + offB[n]);
  };
  auto mfma_step = [&](int fb) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[fb][n][j], af[fb][m][j], acc[m][n], 0, 0, 0);  // transposed tile: D[channel][pixel]
  };
  auto sched_step = [&](bool with_dma) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_group_barrier(0x100, MT + NTW, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, MT * NTW, 0);
    if (with_dma) {
      __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, NDMA, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 3 * MT * NTW, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // The two waves that share a SIMD must not issue their DMA pieces at the same time: an LDS-DMA
  // wave-instruction occupies its wave for 60-185 cycles (MI355X guide, per-instruction table), six
  // of them ~1,000 -- if both partners do that right after the barrier the MFMA pipe idles for that
  // long in every stage (measured: 9,670 instead of 8,192 cycles per stage).  So half of the waves
  // ("late") issue one step after the other half; while one partner feeds the DMA engine the
  // other one feeds the MFMA pipe.
  auto stage = [&](const float* cur, const float* nxt, float* fill, auto late) __attribute__((always_inline)) {
    load_frags(cur, 1, 1);
    mfma_step(0);
    sched_step(false);
    load_frags(cur, 2, 0);
    mfma_step(1);
    sched_step(false);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): only the pieces of the next stage are outstanding, issued most of a stage ago
    __builtin_amdgcn_s_barrier();
    if (!decltype(late)::value) issue_stage(fill);
    load_frags(cur, 3, 1);
    mfma_step(0);
    sched_step(!decltype(late)::value);
    if (decltype(late)::value) issue_stage(fill);
    load_frags(nxt, 0, 0);
    mfma_step(1);
    sched_step(decltype(late)::value);
  };

  // ---- epilogue of one tile.  The product is accumulated transposed (weights as the MFMA's A operand), so
  // in the C/D map of v_mfma_f32_32x32x2_f32 a lane owns ONE pixel (lane & 31) and its registers run over
  // output channels (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5): each aligned register quad is four consecutive
  // channels -> bias / scale / residual loads and the stores are 16-B vectors.
  auto epilogue = [&](int m0, int ntile) __attribute__((always_inline)) {
    const bool interior = (m0 + TM <= a.M) && ((ntile + 1) * BN <= a.coutp);
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      const int cbase = ntile * BN + (wn * NTW + n) * 32 + 4 * lh;  // + 8 * q
      f32x4 bias4[4], scale4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bias4[q] = *reinterpret_cast<const f32x4*>(a.bias + cbase + 8 * q);
        scale4[q] = a.scale ? *reinterpret_cast<const f32x4*>(a.scale + cbase + 8 * q) : f32x4{1.f, 1.f, 1.f, 1.f};
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int row = m0 + (wm * MT + m) * 32 + lx;
        size_t opix = (size_t)min(row, a.M - 1);
        if (a.out_patch) {
          const int ow = a.out_W >> 1, oh = a.out_H >> 1;
          const int rr = (int)opix;
          const int ox = rr % ow;
          const int r2 = rr / ow;
          const int oy = r2 % oh;
          opix = ((size_t)(r2 / oh) * a.out_H + 2 * oy + (a.out_tap >> 1)) * a.out_W + 2 * ox + (a.out_tap & 1);
        }
        const size_t rofs = opix * a.coutp;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = cbase + 8 * q;
          if constexpr (EPI == 0) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = acc[m][n][4 * q + e] + bias4[q][e];
              if (a.act == 1) x = fmaxf(x, 0.f);
              if (a.act == 2) x = gelu_f(x);
              v[e] = x * scale4[q][e];
            }
            if (interior) {
              if (a.residual) v += *reinterpret_cast<const f32x4*>(a.residual + rofs + col);
              *reinterpret_cast<f32x4*>(a.dst + rofs + col) = v;
            } else if (row < a.M && col < a.coutp) {  // coutp is a multiple of 16: a quad is inside or outside as a whole
              if (a.residual) v += *reinterpret_cast<const f32x4*>(a.residual + rofs + col);
              *reinterpret_cast<f32x4*>(a.dst + rofs + col) = v;
            }
          } else {  // training epilogues: GELU' multiplier from aux (act 3), second output with the pre-activation (dst_pre)
            if (!interior && !(row < a.M && col < a.coutp)) continue;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (a.act == 3) z = *reinterpret_cast<const f32x4*>(a.aux + rofs + col);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = acc[m][n][4 * q + e] + bias4[q][e];
              if (a.act != 3) z[e] = x;
              if (a.act == 1) x = fmaxf(x, 0.f);
              if (a.act == 2) x = gelu_f(x);
              if (a.act == 3) x *= gelu_grad_f(z[e]);  // data gradient straight through the GELU that produced this GEMM's input
              v[e] = x * scale4[q][e];
            }
            if (a.residual) v += *reinterpret_cast<const f32x4*>(a.residual + rofs + col);
            *reinterpret_cast<f32x4*>(a.dst + rofs + col) = v;
            if (a.dst_pre) *reinterpret_cast<f32x4*>(a.dst_pre + rofs + col) = z;
          }
        }
      }
    }
  };

  float* b0 = lds;
  float* b1 = lds + STAGE_FLOATS;
  float* b2 = lds + 2 * STAGE_FLOATS;
  issue_stage(b0);
  issue_stage(b1);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __builtin_amdgcn_s_barrier();
  load_frags(b0, 0, 0);

  auto run = [&](auto late) __attribute__((always_inline)) {
    int st = 0;
    bool more = true;
    auto post = [&]() __attribute__((always_inline)) {  // after every stage: finish the tile if that was its last stage
      st += 1;
      if (st == nse) {
        epilogue(Pc.m0, Pc.ntile);
        zero_acc();
        st = 0;
        vid = nvid;
        nvid += gridDim.x;
        more = next_ok != 0;
        Pc = Pn;
        f_next -= 1;
        next_ok = nvid < total ? 1 : 0;
        if (more && next_ok) setup(nvid, Pn);
      }
    };
    while (more) {
      stage(b0, b1, b2, late);
      post();
      if (!more) break;
      stage(b1, b2, b0, late);
      post();
      if (!more) break;
      stage(b2, b0, b1, late);
      post();
    }
  };
  // a.late_split: 0 = waves 4-7 (and 12-15) are late: waves i, i+4, i+8, ... share SIMD i % 4 (measured); 1 = odd waves; 2 = nobody
  const bool is_late = WAVES >= 8 && (a.late_split == 0 ? ((wave >> 2) & 1) != 0 : (a.late_split == 1 ? (wave & 1) != 0 : false));
  if (is_late)  // wave-uniform; both paths execute the same number of barriers
    run(std::true_type{});
  else
    run(std::false_type{});
  __builtin_amdgcn_s_waitcnt(0x0F70);  // drain the (dummy) tail fetches before the LDS is released
}

int gemm_choose_bn(int coutp) {
  const int cand[4] = {128, 96, 64, 32};
  int best = 128, best_pad = 1 << 30;
  for (int bn : cand) {
    const int padded = (coutp + bn - 1) / bn * bn;
    if (padded < best_pad) {
      best_pad = padded;
      best = bn;
    }
  }
  return best;
}

// Kernel variants: <MT, NTW, WM, WN, NSTAGE, MINW> = wave tile (32-row x 32-col units), wave grid, LDS ring
// depth (stages of 32 K-values), min waves / SIMD.
#define PH_GEMM_VARIANTS(X)     \
  X(0, 2, 2, 4, 2, 3, 2) /* 256x128, 8 waves, ring of 3 = 144 KiB: one workgroup per CU (default, BN = 128) */ \
  X(1, 1, 3, 8, 1, 3, 2) /* 256x96  */ \
  X(2, 2, 1, 4, 2, 3, 2) /* 256x64  */ \
  X(3, 1, 1, 8, 1, 3, 2) /* 256x32  */ \
  X(4, 2, 2, 2, 2, 3, 2) /* 128x128, 4 waves, 96 KiB; for grids too small for 256-row tiles */

template <int MT, int NTW, int WM, int WN, int NSTAGE, int MINW>
struct GemmCfg {
  static constexpr int TM = WM * MT * 32, BN = WN * NTW * 32, THREADS = WM * WN * 64;
  static constexpr size_t LDS = (size_t)NSTAGE * (TM / 8 + BN / 8) * 1024;
};

int prepare_convnext_kernels() {
#define X(id, MT, NTW, WM, WN, S, W)                                                                                                        \
  PH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_dma_kernel<0, MT, NTW, WM, WN, S, W>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)GemmCfg<MT, NTW, WM, WN, S, W>::LDS));                                                              \
  PH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_dma_kernel<1, MT, NTW, WM, WN, S, W>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)GemmCfg<MT, NTW, WM, WN, S, W>::LDS));                                                              \
  PH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_dma_tile_kernel<2, MT, NTW, WM, WN, S, W>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)GemmCfg<MT, NTW, WM, WN, S, W>::LDS));                                                              \
  PH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_dma_tile_kernel<3, MT, NTW, WM, WN, S, W>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)GemmCfg<MT, NTW, WM, WN, S, W>::LDS));                                                              \
  PH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_dma_tile_kernel<4, MT, NTW, WM, WN, S, W>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)GemmCfg<MT, NTW, WM, WN, S, W>::LDS));                                                              \
  PH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_dma_tile_kernel<5, MT, NTW, WM, WN, S, W>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)GemmCfg<MT, NTW, WM, WN, S, W>::LDS));                                                              \
  PH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_dma_kernel<2, MT, NTW, WM, WN, S, W>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)GemmCfg<MT, NTW, WM, WN, S, W>::LDS));                                                              \
  PH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_dma_kernel<0, MT, NTW, WM, WN, S, W, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)GemmCfg<MT, NTW, WM, WN, S, W>::LDS));
  PH_GEMM_VARIANTS(X)
#undef X
  return PH_OK;
}

int gemm_variant_bn(int variant) {
  switch (variant) {
#define X(id, MT, NTW, WM, WN, S, W) \
  case id: return GemmCfg<MT, NTW, WM, WN, S, W>::BN;
    PH_GEMM_VARIANTS(X)
#undef X
    default: return -1;
  }
}

int launch_gemm_variant(int variant, const GemmArgs& a_in, hipStream_t s) {
  GemmArgs a = a_in;
  const int persist2 = a.persist2;  // handle option: persistent workgroups for the 9-tap mode too (measured 3-4 % slower than one tile per workgroup)
  PH_REQUIRE(a.M > 0 && a.c0p > 0 && a.c0p % 16 == 0 && a.c1p % 16 == 0 && a.coutp % 16 == 0 && a.mode >= 0 && a.mode <= 5, "launch_gemm: bad shape");
  PH_REQUIRE(a.mode != 5 || ((a.ksize & 1) && a.ksize >= 1 && a.ksize <= 9 && a.M % (a.H * a.W) == 0 && a.H < 65536 && a.W < 65536), "launch_gemm: k x k conv needs an odd kernel <= 9 and whole images");
  PH_REQUIRE(a.mode != 3 || (((a.all_phases && a.wpack_ph[0] && a.wpack_ph[1] && a.wpack_ph[2] && a.wpack_ph[3]) || ((a.ntaps == 1 || a.ntaps == 2 || a.ntaps == 4) && a.ntaps == (1 + (a.out_tap >> 1)) * (1 + (a.out_tap & 1)))) && a.out_patch && a.M % (a.H * a.W) == 0),
             "launch_gemm: transposed-conv phase needs ntaps matching the phase, out_patch and whole images");
  PH_REQUIRE(a.mode != 4 || (a.H % 2 == 0 && a.W % 2 == 0 && a.M % ((a.H / 2) * (a.W / 2)) == 0), "launch_gemm: stride-2 gather needs even map sizes and whole images");
  PH_REQUIRE(a.mode >= 3 || (!a.affine_first && a.act != 4 && !a.shift), "launch_gemm: affine-first / SiLU epilogues exist for modes 3 and 4 only");
  PH_REQUIRE(a.c1p == 0 || a.src1, "launch_gemm: second source missing");
  PH_REQUIRE((a.act != 3 && !a.dst_pre) || a.mode == 0, "launch_gemm: the training epilogues exist for the Linear mode only");
  PH_REQUIRE(a.act != 3 || a.aux, "launch_gemm: act 3 needs aux");
  PH_REQUIRE(a.mode != 1 || (a.H >= 2 && a.W >= 2), "launch_gemm: 2x2 patches need H, W >= 2");
  PH_REQUIRE(a.mode != 2 || a.M % (a.H * a.W) == 0, "launch_gemm: conv rows must be whole images");
  PH_REQUIRE(gemm_variant_bn(variant) == a.bn, "launch_gemm: variant %d does not match the N tile %d of the packed weights", variant, a.bn);
  int n_cu = 0;  // persistent workgroups: one per CU of the current device
  {
    const int rc_cu = device_cu_count(&n_cu);
    if (rc_cu != PH_OK) return rc_cu;
  }
  switch (variant) {
#define X(id, MT, NTW, WM, WN, S, W)                                                                                           \
  case id: {                                                                                                                   \
    using C = GemmCfg<MT, NTW, WM, WN, S, W>;                                                                                  \
    const dim3 grid((unsigned)std::min<long>((long)((a.M + C::TM - 1) / C::TM) * ((a.coutp + C::BN - 1) / C::BN), (long)n_cu)); \
    if (a.mode == 0 && (a.act == 3 || a.dst_pre))                                                                              \
      hipLaunchKernelGGL((gemm_mfma_dma_kernel<0, MT, NTW, WM, WN, S, W, 1>), grid, dim3(C::THREADS), C::LDS, s, a);           \
    else if (a.mode == 0)                                                                                                      \
      hipLaunchKernelGGL((gemm_mfma_dma_kernel<0, MT, NTW, WM, WN, S, W>), grid, dim3(C::THREADS), C::LDS, s, a);              \
    else if (a.mode == 1)                                                                                                      \
      hipLaunchKernelGGL((gemm_mfma_dma_kernel<1, MT, NTW, WM, WN, S, W>), grid, dim3(C::THREADS), C::LDS, s, a);              \
    else if (a.mode == 2 && persist2)                                                                                          \
      hipLaunchKernelGGL((gemm_mfma_dma_kernel<2, MT, NTW, WM, WN, S, W>), grid, dim3(C::THREADS), C::LDS, s, a);              \
    else {                                                                                                                     \
      const dim3 full((unsigned)(((a.M + C::TM - 1) / C::TM) * ((a.coutp + C::BN - 1) / C::BN)));                              \
      if (a.mode == 2)                                                                                                         \
        hipLaunchKernelGGL((gemm_mfma_dma_tile_kernel<2, MT, NTW, WM, WN, S, W>), full, dim3(C::THREADS), C::LDS, s, a);       \
      else if (a.mode == 3)                                                                                                    \
        hipLaunchKernelGGL((gemm_mfma_dma_tile_kernel<3, MT, NTW, WM, WN, S, W>), dim3(full.x, a.all_phases ? 4 : 1), dim3(C::THREADS), C::LDS, s, a); \
      else if (a.mode == 5)                                                                                                    \
        hipLaunchKernelGGL((gemm_mfma_dma_tile_kernel<5, MT, NTW, WM, WN, S, W>), full, dim3(C::THREADS), C::LDS, s, a);       \
      else                                                                                                                     \
        hipLaunchKernelGGL((gemm_mfma_dma_tile_kernel<4, MT, NTW, WM, WN, S, W>), full, dim3(C::THREADS), C::LDS, s, a);       \
    }                                                                                                                          \
    break;                                                                                                                     \
  }
    PH_GEMM_VARIANTS(X)
#undef X
    default: set_error("launch_gemm: unknown variant %d", variant); return PH_E_INVALID;
  }
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

int launch_gemm(const GemmArgs& a, hipStream_t s) {
  int variant;
  switch (a.bn) {
    case 128: {
      // 256-row tiles (variant 0) or 128-row tiles (variant 4, ~5 % less efficient per tile): whichever wastes
      // less of the chip in its last, partially filled round of workgroups (one workgroup per CU, 256 CUs)
      const long nt = (a.coutp + 127) / 128;
      const long b0 = (long)((a.M + 255) / 256) * nt, b4 = (long)((a.M + 127) / 128) * nt;
      const double e0 = (double)b0 / (double)((b0 + 255) / 256 * 256);
      const double e4 = 0.95 * (double)b4 / (double)((b4 + 255) / 256 * 256);
      variant = e4 > e0 ? 4 : 0;
      break;
    }
    case 96: variant = 1; break;
    case 64: variant = 2; break;
    case 32: variant = 3; break;
    default: set_error("launch_gemm: unsupported N tile %d", a.bn); return PH_E_INVALID;
  }
  return launch_gemm_variant(variant, a, s);
}

}  // namespace ph

extern "C" {

// Diagnostic: time one row-GEMM kernel variant on synthetic (random) operands.  Not a product path;
// used by tools/gemm_bench.py to choose tile shapes.  mode / H / W as in GemmArgs (mode 2: M = B*H*W).
int ph_debug_gemm_bench(int32_t variant, int32_t M, int32_t K, int32_t N, int32_t mode, int32_t H, int32_t W, int32_t act, int32_t iters, float* ms_out) {
  using namespace ph;
  PH_REQUIRE(ms_out && iters > 0 && M > 0 && K % 16 == 0 && N % 16 == 0, "ph_debug_gemm_bench: bad arguments");
  if (prepare_convnext_kernels() != PH_OK) return PH_E_HIP;
  const int bn = gemm_variant_bn(variant);
  PH_REQUIRE(bn > 0, "unknown variant %d", variant);
  const int taps = mode == 0 ? 1 : (mode == 1 ? 4 : 9);
  const int ntiles = (N + bn - 1) / bn;
  const size_t in_rows = mode == 1 ? (size_t)M * 4 : (size_t)M;
  const size_t n_src = in_rows * K, n_w = (size_t)ntiles * ((K + 31) / 32) * taps * (bn / 8) * 256, n_b = (size_t)ntiles * bn, n_dst = (size_t)M * N;
  float *src = nullptr, *w = nullptr, *bias = nullptr, *dst = nullptr, *zeros = nullptr;
  PH_HIP_CHECK(hipMalloc(&src, n_src * 4));
  PH_HIP_CHECK(hipMalloc(&w, n_w * 4));
  PH_HIP_CHECK(hipMalloc(&bias, n_b * 4));
  PH_HIP_CHECK(hipMalloc(&dst, n_dst * 4));
  PH_HIP_CHECK(hipMalloc(&zeros, 256));
  PH_HIP_CHECK(hipMemset(zeros, 0, 256));
  {
    std::vector<float> h(std::max(std::max(n_src, n_w), n_b));
    unsigned st = 12345u;
    const int mode_d = (act >> 8) & 3;  // operand pattern in bits 8-9 of `act`: 0 rand | 1 relu (half zeros) | 2 zeros
    act &= 0xff;
    for (auto& v : h) {
      st = st * 1664525u + 1013904223u;
      v = ((st >> 8) & 0xffff) / 65536.0f - 0.5f;
      if (mode_d == 1) v = v > 0.f ? v : 0.f;
      if (mode_d == 2) v = 0.f;
    }
    PH_HIP_CHECK(hipMemcpy(src, h.data(), n_src * 4, hipMemcpyHostToDevice));
    PH_HIP_CHECK(hipMemcpy(w, h.data(), n_w * 4, hipMemcpyHostToDevice));
    PH_HIP_CHECK(hipMemcpy(bias, h.data(), n_b * 4, hipMemcpyHostToDevice));
  }
  GemmArgs a{};
  a.src0 = src;
  a.wpack = w;
  a.bias = bias;
  a.dst = dst;
  a.zeros = zeros;
  a.c0p = K;
  a.coutp = N;
  a.bn = bn;
  a.M = M;
  a.mode = mode;
  a.H = H;
  a.W = W;
  a.act = act;
  hipEvent_t e0, e1;
  PH_HIP_CHECK(hipEventCreate(&e0));
  PH_HIP_CHECK(hipEventCreate(&e1));
  int rc = PH_OK;
  for (int i = 0; i < 2 && rc == PH_OK; ++i) rc = launch_gemm_variant(variant, a, nullptr);
  PH_HIP_CHECK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters && rc == PH_OK; ++i) rc = launch_gemm_variant(variant, a, nullptr);
  PH_HIP_CHECK(hipEventRecord(e1, nullptr));
  PH_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  PH_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(src);
  (void)hipFree(w);
  (void)hipFree(bias);
  (void)hipFree(dst);
  (void)hipFree(zeros);
  return rc;
}

}  // extern "C"
