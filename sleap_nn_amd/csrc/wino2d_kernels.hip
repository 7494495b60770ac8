// 3x3 "same" convolution as Winograd F(2x2, 3x3) on the fp32 matrix cores (gfx950).
//
// Reference semantics: nn.Conv2d(k3, padding "same") + bias + ReLU of SimpleConvBlock (architectures/encoder_decoder.py:108-121,
// 494-510), the decoder's torch.concat((skip, x)) as two K panels (encoder_decoder.py:545,556) and the encoder's 2x2 max pool
// (architectures/common.py:69-107) as a fused epilogue -- the same contract as conv3x3_wino_persist_kernel (net_kernels.hip),
// which transforms along x only.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A     per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// 16 multiplications per four outputs instead of 36: the matrix cores execute 4/9 of the direct convolution's FLOPs (the
// one-dimensional form: 2/3).  All arithmetic is fp32; the transforms have coefficients 0, +-1, 1/2 only.
//
//   * GEMM view: M = 2x2 output tiles, N = output channels, K = input channels, once per Winograd position (xi, nu) -- sixteen
//     independent GEMMs that share nothing but the raw input patch.  A register-resident accumulator set for all sixteen would be
//     512 VGPRs per wave at M 32 x N 64, so the positions are split over the waves of a workgroup: wave w owns row xi = w & 3 of
//     the transformed patch (all four nu: 4 x 2 accumulators of 16 registers, as many as the 1-D kernel holds) for M half
//     w >> 2.  A workgroup tile is 8 x 8 Winograd tiles = 16 x 16 output pixels x 64 channels.
//   * Input transform on the fly from the raw 18 x 18 halo tile that LDS-DMA stages: a lane reads the two patch rows its xi
//     combines (four ds_read_b128 each), one add/sub per column gives row xi of B^T d, four more give the four A fragments
//     (row xi of B^T d B); each feeds 8 MFMAs.
//   * Weights are transformed on the device (wino2d_pack_kernel) into LDS order [g][xi][nu][n tile][lh][lx][4]: 64 KiB per
//     16-channel chunk.  As in the 1-D kernel they live in three half-chunk slots of 32 KiB (a chunk reads slots (h, h+1); the
//     next chunk's first half streams into h+2, its second half into h after a mid-chunk barrier).
//   * Epilogue: a wave holds, for its xi, P[b] = sum_nu M[xi][nu] A[nu][b]; output row a = 0 is P0 + P1 + P2 and row a = 1 is
//     P1 - P2 - P3 over xi, i.e. over waves: waves xi = 1, 2 pass their P through LDS (the free weight slot and the free halo
//     buffer), wave 0 finishes the even output rows and wave 3 the odd ones (bias, ReLU, stores); the fused 2x2 max pool takes
//     one more hop (wave 0's row maxima to wave 3).
#include <type_traits>

#include "common.h"
#include "net_kernels.h"

namespace ph {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int W2_T = 16;                                   // output tile edge
constexpr int W2_HW = W2_T + 2;                            // halo edge
constexpr int W2_NPIX = W2_HW * W2_HW;                     // 324 halo pixels
constexpr int W2_A_PIECES = (W2_NPIX + 15) / 16;           // 21 pieces of 16 pixels x 16 channels (1 KiB)
constexpr int W2_A_FLOATS = W2_A_PIECES * 256;             // 5376
constexpr int W2_A1_OFF = 8192;                            // second halo buffer; [5376, 8192) is spare, so that either halo buffer
                                                           // plus the spare is a contiguous 32-KiB exchange area
constexpr int W2_B_OFF = W2_A1_OFF + W2_A_FLOATS;          // 13568
constexpr int W2_BH_FLOATS = 8192;                         // half a chunk of weights (N tile 64): 32 pieces
constexpr int W2_LDS_FLOATS = W2_B_OFF + 3 * W2_BH_FLOATS; // 38144 floats = 149 KiB

__device__ __forceinline__ void w2_deal_tile(int id, int tiles, int nt_count, int* tile, int* ntile) {
  // same dealing as deal_tile (net_kernels.hip): every XCD walks a contiguous range of pixel tiles
  if ((tiles & 7) == 0) {
    const int xcd = id & 7, j = id >> 3;
    *ntile = j % nt_count;
    *tile = xcd * (tiles >> 3) + j / nt_count;
  } else {
    *ntile = id % nt_count;
    *tile = id / nt_count;
  }
}

// wpack [panel][tap 9][bn][16] (pack_conv) -> U = G g G^T in LDS order [panel][g 2][xi 4][nu 4][n tile][lh 2][lx 32][4]
__global__ __launch_bounds__(256) void wino2d_pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int panels, int bn) {
  const int nt = bn / 32;
  const int per_panel = 32 * nt * 256;
  const size_t total = (size_t)panels * per_panel;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int panel = (int)(i / per_panel);
    int r = (int)(i - (size_t)panel * per_panel);
    const int e = r & 3, lx = (r >> 2) & 31, lh = (r >> 7) & 1;
    r >>= 8;
    const int n = r % nt;
    r /= nt;
    const int nu = r & 3, xi = (r >> 2) & 3, g = r >> 4;
    const int row = n * 32 + lx, kc = (2 * g + lh) * 4 + e;
    const float* w = src + ((size_t)panel * 9 * bn + row) * 16 + kc;
    const size_t ts = (size_t)bn * 16;  // tap stride
    float h[3];                         // row xi of G g, per kx
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const float g0 = w[(0 * 3 + kx) * ts], g1 = w[(1 * 3 + kx) * ts], g2 = w[(2 * 3 + kx) * ts];
      h[kx] = xi == 0 ? g0 : (xi == 1 ? 0.5f * ((g0 + g2) + g1) : (xi == 2 ? 0.5f * ((g0 + g2) - g1) : g2));
    }
    dst[i] = nu == 0 ? h[0] : (nu == 1 ? 0.5f * ((h[0] + h[2]) + h[1]) : (nu == 2 ? 0.5f * ((h[0] + h[2]) - h[1]) : h[2]));
  }
}
int64_t wino2d_pack_floats(int panels, int bn) { return (int64_t)panels * 32 * (bn / 32) * 256; }
int launch_wino2d_pack(const float* wpack, float* wino, int panels, int bn, hipStream_t s) {
  const int64_t n = wino2d_pack_floats(panels, bn);
  hipLaunchKernelGGL(wino2d_pack_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, wpack, wino, panels, bn);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

template <int BN>
__global__ __launch_bounds__(512, 2) void conv3x3_wino2d_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NT = BN / 32;
  static_assert(BN == 64, "N tile 64");
  constexpr int A_SLOTS = (W2_A_PIECES + 7) / 8;  // 3 halo pieces per wave
  constexpr int B_SLOTS = 32 * NT / 2 / 8;        // 4 weight pieces per wave and half chunk

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xi = wave & 3, mh = wave >> 2;
  const int tiles_x = (a.W + W2_T - 1) / W2_T;
  const int tiles_y = (a.H + W2_T - 1) / W2_T;
  const int tiles = tiles_x * tiles_y * a.B;
  const int ntc = (a.coutp + BN - 1) / BN;
  const int total = tiles * ntc;
  const int chunks0 = a.c0p / 16;
  const int chunks1 = a.c1p / 16;
  const int nchunks = chunks0 + chunks1;
  const int dq = lane >> 4, dr = lane & 15;
  const int lx = lane & 31, lh = lane >> 5;

  struct Plan {
    int a_pix[A_SLOTS];
    unsigned a_ok;
    int b, x0, y0, ntile;
  };
  auto setup = [&](int vid, Plan& P) {
    int t, ntile;
    w2_deal_tile(vid, tiles, ntc, &t, &ntile);
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    P.b = t / tiles_y;
    P.x0 = tx * W2_T;
    P.y0 = ty * W2_T;
    P.ntile = ntile;
    P.a_ok = 0;
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) {
      const int p = min(wave + 8 * s, W2_A_PIECES - 1);
      const int pix = p * 16 + dr;
      const int hy = pix / W2_HW, hx = pix - hy * W2_HW;
      const int gy = P.y0 + hy - 1, gx = P.x0 + hx - 1;
      const bool in = (pix < W2_NPIX) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      P.a_ok |= (in ? 1u : 0u) << s;
      const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
      P.a_pix[s] = (P.b * a.H + cy) * a.W + cx;
    }
  };

  const float* p_src = a.src0;
  const float* p_w = a.wpack_wino2;
  int p_cp = a.c0p, p_coff = 0;
  int f_pix[A_SLOTS];
  unsigned f_ok = 0;
  auto select_fetch = [&](const Plan& P, int ch) {
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) f_pix[s] = P.a_pix[s];
    f_ok = P.a_ok;
    if (ch < chunks0) {
      p_src = a.src0;
      p_cp = a.c0p;
      p_coff = ch * 16;
    } else {
      p_src = a.src1;
      p_cp = a.c1p;
      p_coff = (ch - chunks0) * 16;
    }
    p_w = a.wpack_wino2 + ((size_t)P.ntile * nchunks + ch) * (2 * W2_BH_FLOATS);
  };
  auto dma_a = [&](int s, float* abuf) {
    const int p = min(wave + 8 * s, W2_A_PIECES - 1);
    const float* real = p_src + (size_t)f_pix[s] * p_cp + p_coff + dq * 4;
    const float* zero = a.zeros + dq * 4;
    const float* g = ((f_ok >> s) & 1u) ? real : zero;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(abuf + p * 256), 16, 0, 0);
  };
  auto dma_b = [&](int s, int half, float* bhalf) {  // wave w moves pieces 4w .. 4w+3 of a half chunk
    const int pb = wave * B_SLOTS + s;
    const float* g = p_w + half * W2_BH_FLOATS + pb * 256 + lane * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(bhalf + pb * 256), 16, 0, 0);
  };

  // A side: lane (lx, lh) is Winograd tile (ty = 4 mh + (lx >> 3), tx = lx & 7); wave xi combines patch rows (ra, rb):
  // xi 0: d0 - d2, xi 1: d1 + d2, xi 2: d2 - d1, xi 3: d1 - d3.  Quad lh (+ 2g) of the pixel.
  const int ra = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
  const int rb = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
  const float sgn = xi == 1 ? 1.f : -1.f;
  int offD[2][4];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int pix = (2 * (4 * mh + (lx >> 3)) + (rr ? rb : ra)) * W2_HW + 2 * (lx & 7) + c;
      offD[rr][c] = (pix >> 4) * 256 + lh * 64 + (pix & 15) * 4;
    }
  const int offB = xi * 4 * NT * 256 + lh * 128 + lx * 4;
  float* const bbuf = lds + W2_B_OFF;

  Plan P, Pn;
  int vid = blockIdx.x;
  setup(vid, P);
  select_fetch(P, 0);
#pragma unroll
  for (int s = 0; s < A_SLOTS; ++s) dma_a(s, lds);
#pragma unroll
  for (int s = 0; s < B_SLOTS; ++s) dma_b(s, 0, bbuf);
#pragma unroll
  for (int s = 0; s < B_SLOTS; ++s) dma_b(s, 1, bbuf + W2_BH_FLOATS);
  __syncthreads();
  int apar = 0;  // halo buffer of the running chunk
  int hb = 0;    // weight half-slot of the running chunk's first half; the second half sits in (hb + 1) % 3
  while (true) {
    const int nvid = vid + gridDim.x;
    const bool has_next = nvid < total;  // workgroup-uniform
    if (has_next) setup(nvid, Pn);
    f32x16 acc[4][NT];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][n][r] = 0.f;
    for (int ch = 0; ch < nchunks; ++ch) {
      const float* acur = lds + (apar ? W2_A1_OFF : 0);
      float* anxt = lds + (apar ? 0 : W2_A1_OFF);
      const int h1 = hb == 2 ? 0 : hb + 1, h2 = h1 == 2 ? 0 : h1 + 1;
      const float* b_first = bbuf + hb * W2_BH_FLOATS;
      const float* b_second = bbuf + h1 * W2_BH_FLOATS;
      float* bn_first = bbuf + h2 * W2_BH_FLOATS;   // next chunk, first half: free since the previous chunk ended
      float* bn_second = bbuf + hb * W2_BH_FLOATS;  // next chunk, second half: free after this chunk's mid barrier
      if (ch + 1 < nchunks)
        select_fetch(P, ch + 1);
      else if (has_next)
        select_fetch(Pn, 0);  // the next tile's first chunk rides under this tile's last one
      else
        select_fetch(P, ch);  // nothing left: refetch (harmless)
      f32x4 av[2][4], bf[2][NT];
      auto make_a = [&](int g, int slot) {
        f32x4 t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 da = *reinterpret_cast<const f32x4*>(acur + offD[0][c] + g * 128);
          const f32x4 db = *reinterpret_cast<const f32x4*>(acur + offD[1][c] + g * 128);
          t[c] = da + sgn * db;
        }
        av[slot][0] = t[0] - t[2];
        av[slot][1] = t[1] + t[2];
        av[slot][2] = t[2] - t[1];
        av[slot][3] = t[1] - t[3];
      };
      auto load_b = [&](int q, int fb) {
        const float* base = ((q >> 2) == 0 ? b_first : b_second) + (q & 3) * NT * 256 + offB;
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[fb][n] = *reinterpret_cast<const f32x4*>(base + n * 256);
      };
      make_a(0, 0);
      load_b(0, 0);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int g = q >> 2, nu = q & 3, fcur = q & 1;
        if (q + 1 < 8) load_b(q + 1, fcur ^ 1);
        if (q == 1) make_a(1, 1);
        // DMA: next chunk's halo and first weight half during the first half, its second weight half after the mid barrier
        if (q < A_SLOTS) dma_a(q, anxt);
        if (q < B_SLOTS) dma_b(q, 0, bn_first);
        if (q >= 4 && q < 4 + B_SLOTS) dma_b(q - 4, 1, bn_second);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[nu][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][nu][j], bf[fcur][n][j], acc[nu][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (q == 3) {
          __builtin_amdgcn_s_barrier();  // every wave is done reading the first weight half: its slot may be refilled
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
      apar ^= 1;
      hb = hb == 0 ? 2 : hb - 1;  // (hb + 2) % 3: the slot the next chunk's first half was fetched into
    }
    // ---- epilogue.  In lane (lx, lh) accumulator register r is tile (ty = 4 mh + (r >> 2), tx = (r & 3) + 4 lh), channel n * 32 + lx.
    // Free LDS now: weight slot (hb + 2) % 3 and the halo buffer of the finished chunk (+ the spare next to it).
    const int b = P.b, x0 = P.x0, y0 = P.y0, ntile = P.ntile;
    const bool interior = (x0 + W2_T <= a.W) && (y0 + W2_T <= a.H) && ((ntile + 1) * BN <= a.coutp);
    float* const r1 = bbuf + (hb == 0 ? 2 : hb - 1) * W2_BH_FLOATS;  // (hb + 2) % 3
    float* const r2 = lds + (apar ? 0 : W2_A_FLOATS);                 // finished buffer A0 -> [0, 8192); A1 -> [5376, 13568)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float m0 = acc[0][n][r], m1 = acc[1][n][r], m2 = acc[2][n][r], m3 = acc[3][n][r];
        acc[0][n][r] = (m0 + m1) + m2;
        acc[1][n][r] = (m1 - m2) - m3;
      }
    if (xi == 1 || xi == 2) {
      float* const rg = (xi == 1 ? r1 : r2) + lane * 4;
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            f32x4 v;
            v[0] = acc[bb][n][4 * k + 0];
            v[1] = acc[bb][n][4 * k + 1];
            v[2] = acc[bb][n][4 * k + 2];
            v[3] = acc[bb][n][4 * k + 3];
            *reinterpret_cast<f32x4*>(rg + (((mh * 2 + bb) * NT + n) * 4 + k) * 256) = v;
          }
    }
    __syncthreads();
    if (xi == 0 || xi == 3) {
      const float* const g1 = r1 + lane * 4;
      const float* const g2 = r2 + lane * 4;
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int o = (((mh * 2 + bb) * NT + n) * 4 + k) * 256;
            const f32x4 p1 = *reinterpret_cast<const f32x4*>(g1 + o);
            const f32x4 p2 = *reinterpret_cast<const f32x4*>(g2 + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float mine = acc[bb][n][4 * k + e];
              acc[bb][n][4 * k + e] = xi == 0 ? mine + (p1[e] + p2[e]) : (p1[e] - p2[e]) - mine;
            }
          }
    }
    __syncthreads();
    const int arow = xi == 3 ? 1 : 0;  // output row of the tile this wave finishes
    if (xi == 0 || xi == 3) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int co = ntile * BN + n * 32 + lx;
        const float bias = a.bias[co];
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[bb][n][r] + bias;
            acc[bb][n][r] = a.relu ? fmaxf(v, 0.f) : v;
          }
      }
    }
    if (a.dst_pool) {  // fused 2x2/2 max pool ("same" padding: zeros beyond the image)
      float* const rp = r1 + lane * 4;
      f32x16 rm[NT];
      if (xi == 0 || xi == 3) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v0 = acc[0][n][r], v1 = acc[1][n][r];
            if (!interior) {
              const int y = y0 + 2 * (4 * mh + (r >> 2)) + arow, x = x0 + 2 * ((r & 3) + 4 * lh);
              v0 = (y < a.H && x < a.W) ? v0 : 0.f;
              v1 = (y < a.H && x + 1 < a.W) ? v1 : 0.f;
            }
            rm[n][r] = fmaxf(v0, v1);
          }
      }
      if (xi == 0) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            f32x4 v;
            v[0] = rm[n][4 * k + 0];
            v[1] = rm[n][4 * k + 1];
            v[2] = rm[n][4 * k + 2];
            v[3] = rm[n][4 * k + 3];
            *reinterpret_cast<f32x4*>(rp + ((mh * NT + n) * 4 + k) * 256) = v;
          }
      }
      __syncthreads();
      if (xi == 3) {
        const int Hp = (a.H + 1) >> 1, Wp = (a.W + 1) >> 1;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int co = ntile * BN + n * 32 + lx;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const f32x4 top = *reinterpret_cast<const f32x4*>(rp + ((mh * NT + n) * 4 + k) * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * k + e;
              const int py = (y0 >> 1) + 4 * mh + (r >> 2), px = (x0 >> 1) + (r & 3) + 4 * lh;
              if (interior || (py < Hp && px < Wp && co < a.coutp)) a.dst_pool[((size_t)(b * Hp + py) * Wp + px) * a.coutp + co] = fmaxf(top[e], rm[n][r]);
            }
          }
        }
      }
      __syncthreads();  // the pool hop's LDS is the next chunk's first DMA target
    }
    if (xi == 0 || xi == 3) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int co = ntile * BN + n * 32 + lx;
        if (interior) {
          float* const dp0 = a.dst + ((size_t)(b * a.H + y0 + 8 * mh + arow) * a.W + x0 + 8 * lh) * a.coutp + co;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float* dp = dp0 + ((size_t)(2 * (r >> 2)) * a.W + 2 * (r & 3)) * a.coutp;
            dp[0] = acc[0][n][r];
            dp[a.coutp] = acc[1][n][r];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int y = y0 + 2 * (4 * mh + (r >> 2)) + arow, x = x0 + 2 * ((r & 3) + 4 * lh);
            if (y < a.H && co < a.coutp) {
              float* dp = a.dst + ((size_t)(b * a.H + y) * a.W + x) * a.coutp + co;
              if (x < a.W) dp[0] = acc[0][n][r];
              if (x + 1 < a.W) dp[a.coutp] = acc[1][n][r];
            }
          }
        }
      }
    }
    if (!has_next) break;
    vid = nvid;
    P = Pn;
  }
}

static int w2_cu_count(int* out) {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    PH_HIP_CHECK(hipGetDevice(&dev));
    PH_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  }
  *out = n_cu;
  return PH_OK;
}

int prepare_wino2d_kernels() {
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wino2d_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) {
    set_error("hipFuncSetAttribute(wino2d) failed: %s", hipGetErrorString(e));
    return PH_E_HIP;
  }
  return PH_OK;
}

// 3x3 conv, N tile 64, no accumulation into dst: the caller (launch_conv3x3_dma) checks those.
int launch_conv3x3_wino2d(const ConvArgs& a, hipStream_t s) {
  PH_REQUIRE(a.bn == 64 && a.wpack_wino2 && !a.accumulate, "wino2d: N tile 64, transformed weights, no accumulate");
  int n_cu = 0;
  const int rc = w2_cu_count(&n_cu);
  if (rc != PH_OK) return rc;
  const int tiles = ((a.W + W2_T - 1) / W2_T) * ((a.H + W2_T - 1) / W2_T) * a.B;
  const int ntc = (a.coutp + 63) / 64;
  hipLaunchKernelGGL(conv3x3_wino2d_kernel<64>, dim3(std::min(tiles * ntc, n_cu)), dim3(512), (size_t)W2_LDS_FLOATS * sizeof(float), s, a);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

}  // namespace ph
