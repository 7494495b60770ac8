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
#include "common.h"
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

int launch_patch_stem(const PatchStemArgs& a, hipStream_t s) {
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
constexpr int DW_STRIP = 8;

__global__ __launch_bounds__(256) void dwconv7_kernel(DwConvArgs a) {
  const int groups = a.cp >> 2;
  const int strips = (a.W + DW_STRIP - 1) / DW_STRIP;
  const size_t total = (size_t)a.B * a.H * strips * groups;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int gq = (int)(idx % groups);
    size_t p = idx / groups;
    const int st = (int)(p % strips);
    p /= strips;
    const int y = (int)(p % a.H);
    const int b = (int)(p / a.H);
    const int x0 = st * DW_STRIP;
    const f32x4 bias = *reinterpret_cast<const f32x4*>(a.bias + gq * 4);
    f32x4 acc[DW_STRIP];
#pragma unroll
    for (int o = 0; o < DW_STRIP; ++o) acc[o] = bias;
    for (int ky = 0; ky < 7; ++ky) {
      const int iy = y + ky - 3;
      if (iy < 0 || iy >= a.H) continue;  // uniform across the lanes of a pixel row
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
      for (int kx = 0; kx < 7; ++kx) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(a.w + (size_t)(ky * 7 + kx) * a.cp + gq * 4);
#pragma unroll
        for (int o = 0; o < DW_STRIP; ++o) acc[o] += in[o + kx] * w;
      }
    }
    float* drow = a.dst + ((size_t)(b * a.H + y) * a.W) * a.cp + gq * 4;
#pragma unroll
    for (int o = 0; o < DW_STRIP; ++o)
      if (x0 + o < a.W) *reinterpret_cast<f32x4*>(drow + (size_t)(x0 + o) * a.cp) = acc[o];
  }
}

int launch_dwconv7(const DwConvArgs& a, hipStream_t s) {
  const size_t total = (size_t)a.B * a.H * ((a.W + DW_STRIP - 1) / DW_STRIP) * (a.cp / 4);
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
// erf as ATen's vectorised CPU kernels compute it (Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7):
// branch-free, and the same approximation the reference's GELU runs through on the CPU.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = 1.0f / (1.0f + 0.3275911f * ax);
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * expf(-x * x);
  return copysignf(r, x);
}

template <int NT_TOTAL, int WN>
__global__ __launch_bounds__(512, 4) void gemm_mfma_dma_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int BN = NT_TOTAL * 32;
  constexpr int MT = WN;
  constexpr int NTW = NT_TOTAL / WN;
  constexpr int B_PIECES = BN / 16;
  constexpr int STAGE_FLOATS = (16 + B_PIECES) * 256;
  static_assert(NT_TOTAL % WN == 0 && (8 / WN) * MT * 32 == 256, "wave arrangement must tile 256 x BN");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int mtiles = (a.M + 255) / 256;
  int t, ntile;
  decode_block_1d(mtiles, (a.coutp + BN - 1) / BN, &t, &ntile);
  const int m0 = t * 256;

  f32x16 acc[MT][NTW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int ntaps = a.mode == 0 ? 1 : (a.mode == 1 ? 4 : 9);
  const int total = ((a.c0p + a.c1p) / 16) * ntaps;  // stages of 16 K-values
  const float* wbase = a.wpack + (size_t)ntile * total * (B_PIECES * 256);

  // ---- DMA plan: lane L of piece p moves quad q = L >> 4 of row p * 16 + (L & 15)
  const int dq = lane >> 4, dr = lane & 15;
  long long a_pix[2];
  unsigned a_mask[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int row = min(m0 + (wave + 8 * s) * 16 + dr, a.M - 1);
    unsigned mask = 0x1ffu;
    long long pix = row;
    if (a.mode == 1) {
      const int ow = a.W >> 1, oh = a.H >> 1;
      const int ox = row % ow;
      const int r2 = row / ow;
      const int oy = r2 % oh;
      pix = ((long long)(r2 / oh) * a.H + 2 * oy) * a.W + 2 * ox;
    } else if (a.mode == 2) {
      const int x = row % a.W;
      const int y = (row / a.W) % a.H;
      mask = 0;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        mask |= (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? (1u << tap) : 0u;
      }
    }
    a_pix[s] = pix;
    a_mask[s] = mask;
  }
  // fetch cursor (wave-uniform): which source / 16-channel slice / tap the next stage to fetch is
  int f_idx = 0, f_tap = 0, f_coff = 0, f_src = 0;
  auto issue_stage = [&](float* buf) {
    const float* sp = f_src ? a.src1 : a.src0;
    const int cp = f_src ? a.c1p : a.c0p;
    int toff;  // pixel offset of the tap
    if (a.mode == 0)
      toff = 0;
    else if (a.mode == 1)
      toff = (f_tap >> 1) * a.W + (f_tap & 1);
    else
      toff = (f_tap / 3 - 1) * a.W + (f_tap % 3 - 1);
    const bool live = f_idx < total;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float* real = sp + (a_pix[k] + toff) * cp + f_coff + dq * 4;
      const float* g = (live && ((a_mask[k] >> f_tap) & 1u)) ? real : a.zeros + dq * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(buf + (wave + 8 * k) * 256), 16, 0, 0);
    }
    {
      const int pb = min(wave, B_PIECES - 1);
      const float* g = wbase + ((size_t)min(f_idx, total - 1) * B_PIECES + pb) * 256 + lane * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(buf + (16 + pb) * 256),
                                       16, 0, 0);
    }
    // advance: taps innermost, then 16-channel slices, then the second source
    f_idx += 1;
    f_tap += 1;
    if (f_tap == ntaps) {
      f_tap = 0;
      f_coff += 16;
      if (f_coff >= cp && f_src == 0 && a.c1p > 0) {
        f_coff = 0;
        f_src = 1;
      }
    }
  };

  // ---- fragment read offsets (floats, stage-relative)
  const int lx = lane & 31, lh = lane >> 5;
  int offA[MT], offB[NTW];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int row = (wm * MT + m) * 32 + lx;
    offA[m] = (row >> 4) * 256 + lh * 64 + (row & 15) * 4;
  }
#pragma unroll
  for (int n = 0; n < NTW; ++n) {
    const int col = (wn * NTW + n) * 32 + lx;
    offB[n] = (16 + (col >> 4)) * 256 + lh * 64 + (col & 15) * 4;
  }

  auto stage = [&](const float* cur, float* fill) {
    f32x4 af[2][MT], bf[2][NTW];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int m = 0; m < MT; ++m) af[g][m] = *reinterpret_cast<const f32x4*>(cur + g * 128 + offA[m]);
#pragma unroll
      for (int n = 0; n < NTW; ++n) bf[g][n] = *reinterpret_cast<const f32x4*>(cur + g * 128 + offB[n]);
    }
    issue_stage(fill);
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NTW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][m][j], bf[g][n][j], acc[m][n], 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F73);  // vmcnt(3): everything but the three pieces just issued has landed
    __builtin_amdgcn_s_barrier();
  };

  float* b0 = lds;
  float* b1 = lds + STAGE_FLOATS;
  float* b2 = lds + 2 * STAGE_FLOATS;
  issue_stage(b0);
  issue_stage(b1);
  __builtin_amdgcn_s_waitcnt(0x0F73);
  __builtin_amdgcn_s_barrier();
  for (int st = 0; st < total; st += 3) {
    stage(b0, b2);
    if (st + 1 < total) stage(b1, b0);
    if (st + 2 < total) stage(b2, b1);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // drain the (dummy) tail fetches before the LDS is released

  // ---- epilogue.  C/D map of v_mfma_f32_32x32x2_f32: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const bool interior = (m0 + 256 <= a.M) && ((ntile + 1) * BN <= a.coutp);
#pragma unroll
  for (int n = 0; n < NTW; ++n) {
    const int col = ntile * BN + (wn * NTW + n) * 32 + lx;
    const float bias = a.bias[col];
    const float scale = a.scale ? a.scale[col] : 1.0f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int rbase = m0 + (wm * MT + m) * 32 + 4 * lh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[m][n][r] + bias;
        if (a.act == 1) v = fmaxf(v, 0.f);
        if (a.act == 2) v = 0.5f * v * (1.0f + erf_as(v * 0.70710678118654752440f));
        acc[m][n][r] = v * scale;
      }
      if (interior) {
        float* drow = a.dst + (size_t)rbase * a.coutp + col;
        if (a.residual) {
          const float* rrow = a.residual + (size_t)rbase * a.coutp + col;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][n][r] += rrow[(size_t)((r & 3) + 8 * (r >> 2)) * a.coutp];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) drow[(size_t)((r & 3) + 8 * (r >> 2)) * a.coutp] = acc[m][n][r];
      } else {
        const int cc = min(col, a.coutp - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (row < a.M && col < a.coutp) {
            const size_t o = (size_t)row * a.coutp + cc;
            a.dst[o] = acc[m][n][r] + (a.residual ? a.residual[o] : 0.f);
          }
        }
      }
    }
  }
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

static size_t gemm_lds_bytes(int bn) { return (size_t)3 * (16 + bn / 16) * 1024; }

int prepare_convnext_kernels() {
  PH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_dma_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds_bytes(128)));
  PH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_dma_kernel<3, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds_bytes(96)));
  PH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_dma_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds_bytes(64)));
  PH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_dma_kernel<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds_bytes(32)));
  return PH_OK;
}

int launch_gemm(const GemmArgs& a, hipStream_t s) {
  PH_REQUIRE(a.M > 0 && a.c0p > 0 && a.c0p % 16 == 0 && a.c1p % 16 == 0 && a.coutp % 16 == 0 && a.mode >= 0 && a.mode <= 2, "launch_gemm: bad shape");
  PH_REQUIRE(a.c1p == 0 || a.src1, "launch_gemm: second source missing");
  PH_REQUIRE(a.mode != 1 || (a.H >= 2 && a.W >= 2), "launch_gemm: 2x2 patches need H, W >= 2");
  PH_REQUIRE(a.mode != 2 || a.M % (a.H * a.W) == 0, "launch_gemm: conv rows must be whole images");
  const int mtiles = (a.M + 255) / 256;
  const int ntiles = (a.coutp + a.bn - 1) / a.bn;
  const dim3 grid((unsigned)(mtiles * ntiles));
  const size_t lds = gemm_lds_bytes(a.bn);
  switch (a.bn) {
    case 128: hipLaunchKernelGGL((gemm_mfma_dma_kernel<4, 2>), grid, dim3(512), lds, s, a); break;
    case 96: hipLaunchKernelGGL((gemm_mfma_dma_kernel<3, 1>), grid, dim3(512), lds, s, a); break;
    case 64: hipLaunchKernelGGL((gemm_mfma_dma_kernel<2, 2>), grid, dim3(512), lds, s, a); break;
    case 32: hipLaunchKernelGGL((gemm_mfma_dma_kernel<1, 1>), grid, dim3(512), lds, s, a); break;
    default: set_error("launch_gemm: unsupported N tile %d", a.bn); return PH_E_INVALID;
  }
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

}  // namespace ph
