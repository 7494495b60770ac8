// 3x3 "same" convolution in the plain-fp16 precision (the reference's autocast mode, torch_backend.py:113-143) on v_mfma_f32_16x16x32_f16,
// re-decomposed for the layers conv3x3_f16_persist_kernel (f16_kernels.hip) runs furthest from the pipe's rate: the 96 x 96 ... 24 x 24 levels of
// the UNet at 16 frames (BASELINE cfg5) and every decoder concat conv.
//
// Reference semantics: SimpleConvBlock's Conv2d(k3, "same") + bias + ReLU and the 2x2 max pool behind it (architectures/encoder_decoder.py:108-121,
// architectures/common.py:69-107); the decoder's concat((skip, x)) as two K panels (encoder_decoder.py:545,556) and -- `src1_lowres` -- the bilinear x2 in
// front of it (encoder_decoder.py:339-420: F.interpolate(scale_factor=2, mode="bilinear", align_corners=False)) inside the loader; a 1x1 head
// (architectures/heads.py:58-67) in the epilogue.
//
// What bounds the 16 x 32-pixel x 64-channel persistent kernel on these layers: every wave stages 75 KiB per 32-channel chunk by
// LDS-DMA between its own MFMAs (an LDS-DMA wave-instruction holds its wave 60 - 185 cycles), half of those bytes are weights, the fixed tile pads a 48 x 48
// map by 25 % and a 24 x 24 one by 44 %, and 288 / 576 units of one size leave the last round of the 256 CUs half empty.  Here
//   * the M tile is R whole rows x Wt columns, LINEARISED: M tile m of a wave is 16 consecutive tile pixels wherever the row ends, so any (R, Wt) with
//     R Wt <= 16 MT MG works and the host picks the pair whose unit count divides the chip best (f16_rows_plan): 12 x 48 pixels x 64 channels makes a 48 x 48 map
//     with 256 channels at 16 frames exactly 256 units of 576 pixels, no padding;
//   * twelve waves: 8 MFMA waves as MG pixel groups x SN = 8 / MG slices of 32 output channels (wave tile 16 MT pixels x 32 channels: MT x 2 accumulators of
//     v_mfma_f32_16x16x32_f16, transposed product D[channel][pixel]) and 4 LOADER waves, one per SIMD, which alone issue the LDS-DMA of the halo -- one
//     32-channel chunk ahead, double-buffered, one workgroup barrier per chunk -- so no MFMA wave ever waits on a DMA issue or on a vmcnt(0);
//   * weights never pass through LDS: a wave's two A operands of a tap (32 channels x 32 k: 2 KiB) are 16-byte-per-lane loads from the packed fp16 panels
//     (f16_weight_pack_kernel's [n tile][chunk][tap][64 rows] pieces ARE the 16x16x32 operand, 1 KiB contiguous), two taps ahead in a three-deep register ring;
//   * a half-resolution second source is staged as it lies in memory (a (R/2 + 2) x (Wt/2 + 2) tile, clamped at the image border, two chunks ahead) and the
//     loader waves blend it into the halo buffer cell by cell -- one low-resolution cell (i .. i+1, j .. j+1) gives the 2 x 2 output pixels between its
//     corners, the arithmetic of upsample2x_fmt_kernel, fp32, rounded to fp16 once -- so the up-sampled tensor never exists in HBM;
//   * epilogue: bias, ReLU, fp16, the tile laid out in LDS as it lies in memory, 16-byte stores of whole channel rows; the fused 2x2 max pool and a 1x1
//     head (fp32 weights as fp16 (hi, lo') pairs, NCHW fp32 out) read that staged tile.
#include <algorithm>
#include <cmath>
#include <type_traits>

#include "act_format.h"
#include "common.h"
#include "f16_kernels.h"

namespace ph {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef PH_ROWS_PD
#define PH_ROWS_PD 4  // fragment reads in flight behind the MFMAs (steps ahead)
#endif
#ifndef PH_ROWS_WD
#define PH_ROWS_WD 2  // taps the weight loads run ahead of their MFMAs
#endif
#ifndef PH_ROWS_PRIO
#define PH_ROWS_PRIO 0  // 1: loader waves at s_setprio 1, 2: MFMA waves at s_setprio 1
#endif
#ifndef PH_ROWS_EXP
#define PH_ROWS_EXP 0  // timing experiments (wrong results): 1 no halo DMA, 2 no weight loads in the K loop, 4 no fragment reads, 8 no MFMAs
#endif

namespace {
constexpr int ROWS_MAX_PIECES = 60;        // 16-pixel pieces of one halo chunk (1 KiB each)
constexpr int ROWS_LDS_LIMIT = 160 * 1024;
constexpr int ROWS_MT_UNALIGNED = 5;       // M tiles per wave of the per-lane-address form (register budget of twelve waves: 168)

__device__ __forceinline__ f16x8 h8(const f32x4& v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ void barrier_after_dma() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void barrier_plain() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
}  // namespace

// LDS map (bytes): [B0: NP KiB][X: xb][B1: NP KiB][low0 | low1: 2 x lowp KiB][head weights]; the epilogue's staging tile starts at 0 when the tile's last
// chunk lay in B0 and at NP KiB (X, then B1) when it lay in B1 -- the other buffer already holds the next tile's first chunk.
// ALIGNED: the tile width is a multiple of 16, so an M tile is 16 consecutive pixels of ONE tile row that start at a multiple-of-16 column: its fragment address is a
// wave-uniform part (row, column group: scalar registers) + one of three per-lane values (kx = 0, 1, 2).  Otherwise every (M tile, kx) has a per-lane address (3 MT registers).
template <int MT, int MG, bool ALIGNED>
__global__ __launch_bounds__(768) void conv3x3_f16_rows_kernel(ConvF16Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int SN = 8 / MG;          // slices of 32 output channels
  constexpr int BN = 32 * SN;         // 64 (MG 4) or 128 (MG 2)
  constexpr int SP = BN * 2 + 16;     // staged bytes per pixel (+ 16: bank spread of the 8-byte writes)
  constexpr int PER = BN / 8;         // 16-byte units per staged pixel
  constexpr int WCH = 9 * 64 * 16;    // floats per (64-channel weight tile, chunk)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int R = a.rows_r, Wt = a.rows_wt, HP16 = a.rows_hp16;
  const int NP = (R + 2) * HP16, BUF = NP * 1024;
  const int off_b1 = BUF + a.rows_xb;
  const int off_low = off_b1 + BUF;
  const int LC16 = a.rows_lc16, LOWB = a.rows_lowp * 1024;
  const int off_hw = off_low + 2 * LOWB;
  const int tiles_x = (a.W + Wt - 1) / Wt, tiles_y = (a.H + R - 1) / R, tiles = tiles_x * tiles_y * a.B;
  const int ntc = (a.coutp + BN - 1) / BN, total = tiles * ntc, nch = a.chunks0 + a.chunks1;
  const int npx = R * Wt;

  auto decode = [&](int vid, int& b, int& y0, int& x0, int& ntile) {
    int t;  // XCD-aware dealing (deal_tile in net_kernels.hip): every XCD walks a contiguous range of pixel tiles, the N tile is the fastest index
    if ((tiles & 7) == 0) {
      const int xcd = vid & 7, j = vid >> 3;
      ntile = j % ntc;
      t = xcd * (tiles >> 3) + j / ntc;
    } else {
      ntile = vid % ntc;
      t = vid / ntc;
    }
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    b = t / tiles_y;
    y0 = ty * R;
    x0 = tx * Wt;
  };

  if (wave >= 8) {
    // ============================ loader waves ============================
    const int lw = wave - 8;
    const int lpx = lane & 15, lq = lane >> 4;
    if (PH_ROWS_PRIO == 1) __builtin_amdgcn_s_setprio(1);
    // full-resolution chunk -> halo buffer by LDS-DMA: piece p = (halo row, 16-pixel column group), lane = (quad, pixel)
    auto fill_dma = [&](const float* src, int rs, int coff, int b, int y0, int x0, int bufb) {
      int hr = 0, cg = lw;
      while (cg >= HP16) {
        cg -= HP16;
        ++hr;
      }
      for (int p = lw; p < NP && !(PH_ROWS_EXP & 1); p += 4) {
        const int hx = cg * 16 + lpx, gy = y0 + hr - 1, gx = x0 + hx - 1;
        const bool in = hx < Wt + 2 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const float* g = in ? src + ((size_t)(b * a.H + gy) * a.W + gx) * rs + coff + lq * 4 : a.zeros + lq * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(smem + bufb + p * 1024), 16, 0, 0);
        cg += 4;
        while (cg >= HP16) {
          cg -= HP16;
          ++hr;
        }
      }
    };
    // half-resolution chunk -> low buffer by LDS-DMA: rows i0 .. i0 + R/2 + 1, columns j0 .. j0 + Wt/2 + 1 of the source, clamped to the image (the clamp IS
    // ATen's border rule: the second tap of the last row / column is that row / column again, and a cell in front of the first one repeats the first)
    auto fill_low = [&](int coff, int b, int y0, int x0, int lowb) {
      const int Hl = a.H >> 1, Wl = a.W >> 1, LR = (R >> 1) + 2, LNP = LR * LC16;
      const int i0 = (y0 >> 1) - 1, j0 = (x0 >> 1) - 1;
      int lr = 0, cg = lw;
      while (cg >= LC16) {
        cg -= LC16;
        ++lr;
      }
      for (int p = lw; p < LNP; p += 4) {
        const int li = min(max(i0 + lr, 0), Hl - 1), lj = min(max(j0 + cg * 16 + lpx, 0), Wl - 1);
        const float* g = a.src1 + ((size_t)(b * Hl + li) * Wl + lj) * a.rs1 + coff + lq * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(smem + lowb + p * 1024), 16, 0, 0);
        cg += 4;
        while (cg >= LC16) {
          cg -= LC16;
          ++lr;
        }
      }
    };
    // low buffer -> halo buffer: item = (cell (ci, cj), quad): four 16-byte reads, the 2 x 2 output pixels (halo rows 2 ci, 2 ci + 1; columns 2 cj, 2 cj + 1)
    auto blend = [&](int y0, int x0, int lowb, int bufb) {
      const int CR = (R >> 1) + 1, CC = (Wt >> 1) + 1, items = CR * CC * 4;
      const int lt = tid - 512;
      for (int it = lt; it < items; it += 256) {
        const int q = it & 3, cell = it >> 2;
        const int ci = (cell * a.rows_inv_cc) >> 20, cj = cell - ci * CC;
        auto lowat = [&](int li, int lj) { return *reinterpret_cast<const f16x8*>(smem + lowb + (li * LC16 + (lj >> 4)) * 1024 + q * 256 + (lj & 15) * 16); };
        const f16x8 h00 = lowat(ci, cj), h01 = lowat(ci, cj + 1), h10 = lowat(ci + 1, cj), h11 = lowat(ci + 1, cj + 1);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const int gy = y0 - 1 + 2 * ci + dy;
          const float sy = fmaxf(((float)gy + 0.5f) * 0.5f - 0.5f, 0.f);
          const float ly = sy - (float)(int)sy, hy = 1.f - ly;
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const int hxp = 2 * cj + dx, gx = x0 - 1 + hxp;
            const float sx = fmaxf(((float)gx + 0.5f) * 0.5f - 0.5f, 0.f);
            const float lx = sx - (float)(int)sx, hx = 1.f - lx;
            const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            f16x8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const float v = hy * (hx * (float)h00[k] + lx * (float)h01[k]) + ly * (hx * (float)h10[k] + lx * (float)h11[k]);
              o[k] = in ? (_Float16)v : (_Float16)0.f;
            }
            if (hxp < HP16 * 16) *reinterpret_cast<f16x8*>(smem + bufb + ((2 * ci + dy) * HP16 + (hxp >> 4)) * 1024 + q * 256 + (hxp & 15) * 16) = o;
          }
        }
      }
    };
    // the same in packed fp16 arithmetic (`rows_blend16`; v_pk_fma_f16, two channels per instruction, no conversions: 64 instead of ~180 vector operations per
    // item): with ATen's scale-2 weights -- 3/4 of the nearer source pixel, 1/4 of the farther one; at the image border both taps are the SAME clamped pixel, so the
    // constant weights reproduce the border rule -- a one-dimensional blend fma(a, 3/4, b / 4) is the exact value rounded ONCE (b / 4 is exact in fp16).  The
    // horizontal result is rounded to fp16 before the vertical blend, where the fp32 form above rounds once at the end: at most one more half-ulp on these tensors.
    auto blend16 = [&](int y0, int x0, int lowb, int bufb) {
      const int CR = (R >> 1) + 1, CC = (Wt >> 1) + 1, items = CR * CC * 4;
      const int lt = tid - 512;
      f16x8 c75, c25, zero8;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        c75[k] = (_Float16)0.75f;
        c25[k] = (_Float16)0.25f;
        zero8[k] = (_Float16)0.f;
      }
      for (int it = lt; it < items; it += 256) {
        const int q = it & 3, cell = it >> 2;
        const int ci = (cell * a.rows_inv_cc) >> 20, cj = cell - ci * CC;
        auto lowat = [&](int li, int lj) { return *reinterpret_cast<const f16x8*>(smem + lowb + (li * LC16 + (lj >> 4)) * 1024 + q * 256 + (lj & 15) * 16); };
        const f16x8 h00 = lowat(ci, cj), h01 = lowat(ci, cj + 1), h10 = lowat(ci + 1, cj), h11 = lowat(ci + 1, cj + 1);
        // horizontal: halo column 2 cj (dx 0) is nearer to source column cj + 1?  gx = x0 - 1 + 2 cj + dx, source coordinate (gx + 0.5) / 2 - 0.5 = (j0 + cj) + 0.25 (dx 0) / + 0.75 (dx 1)
        // with j0 + cj the cell's left source column: dx 0 takes 3/4 of the LEFT pixel, dx 1 3/4 of the RIGHT one
        const f16x8 q00 = h00 * c25, q01 = h01 * c25, q10 = h10 * c25, q11 = h11 * c25;
        const f16x8 t0[2] = {__builtin_elementwise_fma(h00, c75, q01), __builtin_elementwise_fma(h01, c75, q00)};  // upper source row, dx 0 / 1
        const f16x8 t1[2] = {__builtin_elementwise_fma(h10, c75, q11), __builtin_elementwise_fma(h11, c75, q10)};  // lower source row
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int hxp = 2 * cj + dx, gx = x0 - 1 + hxp;
          const f16x8 o0 = __builtin_elementwise_fma(t0[dx], c75, t1[dx] * c25);  // dy 0: 3/4 of the upper row
          const f16x8 o1 = __builtin_elementwise_fma(t1[dx], c75, t0[dx] * c25);  // dy 1: 3/4 of the lower row
#pragma unroll
          for (int dy = 0; dy < 2; ++dy) {
            const int gy = y0 - 1 + 2 * ci + dy;
            const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            if (hxp < HP16 * 16) *reinterpret_cast<f16x8*>(smem + bufb + ((2 * ci + dy) * HP16 + (hxp >> 4)) * 1024 + q * 256 + (hxp & 15) * 16) = in ? (dy ? o1 : o0) : zero8;
          }
        }
      }
    };
    auto is_low = [&](int ch) { return a.src1_lowres && ch >= a.chunks0; };
    // chunk `ch` of a tile into halo buffer `bufb` (a low-resolution chunk must already lie in its low buffer)
    auto fill = [&](int ch, int b, int y0, int x0, int bufb) {
      if (is_low(ch)) {
        if (a.rows_blend16)
          blend16(y0, x0, off_low + ((ch - a.chunks0) & 1) * LOWB, bufb);
        else
          blend(y0, x0, off_low + ((ch - a.chunks0) & 1) * LOWB, bufb);
      }
      else if (ch < a.chunks0)
        fill_dma(a.src0, a.rs0, ch * 16, b, y0, x0, bufb);
      else
        fill_dma(a.src1, a.rs1, (ch - a.chunks0) * 16, b, y0, x0, bufb);
    };
    auto stage_low = [&](int ch, int b, int y0, int x0) {
      if (is_low(ch)) fill_low((ch - a.chunks0) * 16, b, y0, x0, off_low + ((ch - a.chunks0) & 1) * LOWB);
    };

    int vid = blockIdx.x;
    int b, y0, x0, ntile;
    decode(vid, b, y0, x0, ntile);
    int g = 0;
    // prologue: chunk 0 -> B0; the low buffers are two chunks ahead of the halo buffers (chunks 0 and 1 staged here)
    stage_low(0, b, y0, x0);
    if (nch > 1) stage_low(1, b, y0, x0);
    if (a.src1_lowres) barrier_after_dma();  // (workgroup-uniform) the blend below reads what every loader wave staged
    fill(0, b, y0, x0, 0);
    barrier_after_dma();
    while (true) {
      const int nvid = vid + gridDim.x;
      const bool has_next = nvid < total;
      int nb = b, ny0 = y0, nx0 = x0, nnt = ntile;
      if (has_next) decode(nvid, nb, ny0, nx0, nnt);
      for (int ch = 0; ch < nch; ++ch) {
        const int nbuf = ((g + 1) & 1) ? off_b1 : 0;
        // during the MFMAs of chunk ch: chunk ch + 1 goes to the other halo buffer, the low-resolution bytes of chunk ch + 2 to the low buffer chunk ch used
        if (ch + 1 < nch)
          fill(ch + 1, b, y0, x0, nbuf);
        else if (has_next)
          fill(0, nb, ny0, nx0, nbuf);
        if (ch + 2 < nch)
          stage_low(ch + 2, b, y0, x0);
        else if (has_next && ch + 2 - nch < nch)
          stage_low(ch + 2 - nch, nb, ny0, nx0);
        barrier_after_dma();
        ++g;
      }
      barrier_plain();  // E1: the MFMA waves have staged the output tile
      barrier_plain();  // E2: ... and read it back; the buffer under it may take DMA pieces again
      if (!has_next) break;
      vid = nvid;
      b = nb, y0 = ny0, x0 = nx0, ntile = nnt;
    }
    return;
  }

  // ============================ MFMA waves ============================
  const int mg = wave / SN, sn = wave % SN;
  if (PH_ROWS_PRIO == 2) __builtin_amdgcn_s_setprio(1);
  const int lpx = lane & 15, lq = lane >> 4;
  // fused head: fp32 weights [o][c] -> (hi, lo' = (w - hi) 2^11) A operands [part][o tile 2][k step BN / 32][lane][8]
  _Float16* const hw_lds = reinterpret_cast<_Float16*>(smem + off_hw);
  if (a.head_w) {
    constexpr int KC = BN / 32;
    for (int i = tid; i < 2 * KC * 64; i += 512) {
      const int l = i & 63, kc = (i >> 6) % KC, ot = (i >> 6) / KC;
      const int o = ot * 16 + (l & 15), k0 = kc * 32 + 8 * (l >> 4);
      f16x8 hi, lo;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float w = o < a.head_cout ? a.head_w[(size_t)o * a.head_wcp + k0 + k] : 0.f;
        hi[k] = split_hi(w);
        lo[k] = split_lo(w, hi[k]);
      }
      *reinterpret_cast<f16x8*>(hw_lds + ((ot * KC + kc) * 64 + l) * 8) = hi;
      *reinterpret_cast<f16x8*>(hw_lds + 2 * KC * 512 + ((ot * KC + kc) * 64 + l) * 8) = lo;
    }
  }

  int vid = blockIdx.x;
  int b, y0, x0, ntile;
  decode(vid, b, y0, x0, ntile);
  // weights of (N tile, chunk): BN 64 -> 64-channel tile `ntile`, slice sn; BN 128 -> tiles 2 ntile + (sn >> 1), slice sn & 1
  auto wtile = [&](int nt) { return a.wpack + (size_t)(BN == 64 ? nt : 2 * nt + (sn >> 1)) * nch * WCH + ((sn & (BN == 64 ? 3 : 1)) * 2) * 256 + lane * 4; };
  const float* wcur = wtile(ntile);
  constexpr int WD = PH_ROWS_WD, WR = WD == 2 ? 3 : 9;  // ring slots: a divisor of 9 > WD, so that a tap's slot is the same in every chunk
  static_assert(WD >= 1 && WD <= 8 && (WD < 3 || WR == 9), "weight prefetch depth");
  f32x4 wr[WR][2];
#pragma unroll
  for (int t = 0; t < WD; ++t) {
    wr[t][0] = *reinterpret_cast<const f32x4*>(wcur + t * 1024);
    wr[t][1] = *reinterpret_cast<const f32x4*>(wcur + t * 1024 + 256);
  }
  if (a.src1_lowres) barrier_plain();  // (the loaders' extra prologue barrier)
  barrier_lds();                       // chunk 0 landed (and the head weights are in LDS)
  int g = 0;
  while (true) {
    const int nvid = vid + gridDim.x;
    const bool has_next = nvid < total;
    int nb = b, ny0 = y0, nx0 = x0, nnt = ntile;
    if (has_next) decode(nvid, nb, ny0, nx0, nnt);
    const float* const wnext_tile = has_next ? wtile(nnt) : wcur;
    // fragment addresses: tile pixel p = 16 (MT mg + m) + lane % 16 -> (row, column) -> halo pixel (row + ky, column + kx)
    int ak[ALIGNED ? 1 : MT][3];  // per-lane byte offsets
    int sm[ALIGNED ? MT : 1];     // ALIGNED: wave-uniform byte offset of M tile m (its row and 16-pixel column group)
    if constexpr (ALIGNED) {
      const int w16 = Wt >> 4;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) ak[0][kx] = ((lpx + kx) >> 4) * 1024 + ((lpx + kx) & 15) * 16 + lq * 256;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        int mt = mg * MT + m;
        mt = mt * 16 < npx ? mt : 0;  // (M tiles beyond the tile compute tile 0 again; nothing of theirs is stored)
        const int r = mt / w16, cgm = mt - r * w16;
        sm[m] = (r * HP16 + cgm) * 1024;
      }
    } else {
      sm[0] = 0;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        int p = (mg * MT + m) * 16 + lpx;
        p = p < npx ? p : npx - 1;  // (lanes beyond the tile compute some valid pixel again; nothing of theirs is stored)
        const int r = (p * a.rows_inv_wt) >> 20, c = p - r * Wt;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) ak[m][kx] = (r * HP16 + ((c + kx) >> 4)) * 1024 + ((c + kx) & 15) * 16 + lq * 256;
      }
    }
    f32x4 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m][0] = acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < nch; ++ch) {
      const int bufb = (g & 1) ? off_b1 : 0;
      const float* const wnxt = ch + 1 < nch ? wcur + WCH : wnext_tile;
      // Flat software pipeline over the 9 MT (tap, M tile) steps of the chunk: a step is one ds_read_b128 (the pixel fragment, B operand) + two MFMAs; the read of
      // step s + PD is issued in front of the MFMAs of step s into a ring of PD + 1 fragments, so a wave has PD reads in flight behind its matrix work (left to
      // itself the compiler reads a pair, waits for it and issues four MFMAs: the LDS latency shows in every pair).
      constexpr int NS = 9 * MT, PD = PH_ROWS_PD, RING = PD + 1;
      f16x8 ring[RING];
      auto rd = [&](int st) __attribute__((always_inline)) {
        const int tap = st / MT, m = st - tap * MT, ky = tap / 3, kx = tap - 3 * ky;
        int so = sm[ALIGNED ? m : 0] + bufb + ky * HP16 * 1024;
        asm("" : "+s"(so));  // keep the wave-uniform part a scalar operand of ONE v_add per read (un-hidden, the 9 MT sums are hoisted out of the chunk loop into 9 MT vector registers)
        return *reinterpret_cast<const f16x8*>(smem + (ak[ALIGNED ? 0 : m][kx] + so));
      };
#pragma unroll
      for (int st = 0; st < PD; ++st)
        if (!(PH_ROWS_EXP & 4) || ch == 0) ring[st % RING] = rd(st);
      auto step = [&](int st, auto more) __attribute__((always_inline)) {
        const int tap = st / MT, m = st - tap * MT;
        if (m == 0 && !(PH_ROWS_EXP & 2)) {  // the weights of the tap after next (of the next chunk / the next tile's first chunk behind taps 7 and 8)
          const float* const wp = tap + WD < 9 ? wcur + (tap + WD) * 1024 : wnxt + (tap + WD - 9) * 1024;
          wr[(tap + WD) % WR][0] = *reinterpret_cast<const f32x4*>(wp);
          wr[(tap + WD) % WR][1] = *reinterpret_cast<const f32x4*>(wp + 256);
        }
        if constexpr (decltype(more)::value && !(PH_ROWS_EXP & 4)) ring[(st + PD) % RING] = rd(st + PD);
        if constexpr (!(PH_ROWS_EXP & 8)) {
          acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h8(wr[tap % WR][0]), ring[st % RING], acc[m][0], 0, 0, 0);
          acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h8(wr[tap % WR][1]), ring[st % RING], acc[m][1], 0, 0, 0);
        } else {
          acc[m][0][0] += wr[tap % WR][0][0] + (float)ring[st % RING][0];  // (keeps the loads alive)
        }
        __builtin_amdgcn_sched_barrier(0);  // the order as written: left alone the scheduler hoists all 18 weight loads of a chunk to its top (72 registers) and sinks every read to its use
      };
#pragma unroll
      for (int st = 0; st < NS - PD; ++st) step(st, std::true_type{});
#pragma unroll
      for (int st = NS - PD; st < NS; ++st) step(st, std::false_type{});
      wcur = wnxt;
      barrier_plain();  // every MFMA wave is out of this buffer, the loaders' next chunk has landed
      ++g;
    }
    // ---- epilogue: D row = channel 4 lq + r of the 16-channel tile, column = pixel lpx of M tile m -> staged [tile pixel][BN channels] fp16
    const int sb = ((g - 1) & 1) ? BUF : 0;
    char* const stage = smem + sb;
    {
      // (bias added AFTER the accumulation, in fp32, like every other fp16 conv kernel: the plans that differ only in which kernel runs a layer stay bit-identical)
      f32x4 bias4[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bias4[nt] = *reinterpret_cast<const f32x4*>(a.bias + ntile * BN + sn * 32 + nt * 16 + 4 * lq);
      const float lo = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int p = (mg * MT + m) * 16 + lpx;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          f32x4 t;
#pragma unroll
          for (int r = 0; r < 4; ++r) t[r] = fmaxf(acc[m][nt][r] + bias4[nt][r], lo);
          const f16x4 h = __builtin_convertvector(t, f16x4);  // (v_cvt_pk_f16_f32: two values per instruction, round to nearest even)
          if (p < npx) *reinterpret_cast<f16x4*>(stage + p * SP + (sn * 32 + nt * 16 + 4 * lq) * 2) = h;
        }
      }
    }
    barrier_lds();  // E1
    const size_t rsb = (size_t)a.rs_dst * 4;
    if (!a.skip_dst) {
      char* const dbase = reinterpret_cast<char*>(a.dst) + (size_t)ntile * (BN * 2);
      for (int idx = tid; idx < npx * PER; idx += 512) {
        const int px = idx / PER, j = idx - px * PER;
        const int r = (px * a.rows_inv_wt) >> 20, c = px - r * Wt;
        const int y = y0 + r, x = x0 + c;
        if (y < a.H && x < a.W && ntile * BN + j * 8 < a.coutp) {
          const f32x4 piece = *reinterpret_cast<const f32x4*>(stage + px * SP + j * 16);
          // non-temporal: not read again by this launch; kept in L2 it would evict the weight panels every workgroup re-reads
          __builtin_nontemporal_store(piece, reinterpret_cast<f32x4*>(dbase + ((size_t)(b * a.H + y) * a.W + x) * rsb + j * 16));
        }
      }
    }
    if (a.dst_pool) {  // fused 2x2/2 max pool, "same": zeros beyond the image (common.py:93-96)
      const int Hp = (a.H + 1) >> 1, Wp = (a.W + 1) >> 1, PW = Wt >> 1, npp = (R >> 1) * PW;
      char* const pbase = reinterpret_cast<char*>(a.dst_pool) + (size_t)ntile * (BN * 2);
      for (int idx = tid; idx < npp * PER; idx += 512) {
        const int pp = idx / PER, j = idx - pp * PER;
        const int pr = (pp * a.rows_inv_pw) >> 20, pc = pp - pr * PW;
        const int y = y0 + 2 * pr, x = x0 + 2 * pc;
        if (y < a.H && x < a.W && ntile * BN + j * 8 < a.coutp) {
          const char* const s00 = stage + ((2 * pr) * Wt + 2 * pc) * SP + j * 16;
          const bool xin = x + 1 < a.W, yin = y + 1 < a.H;
          f16x8 v = *reinterpret_cast<const f16x8*>(s00);
          f16x8 zero;
#pragma unroll
          for (int k = 0; k < 8; ++k) zero[k] = (_Float16)0.f;
          const f16x8 v01 = xin ? *reinterpret_cast<const f16x8*>(s00 + SP) : zero;
          const f16x8 v10 = yin ? *reinterpret_cast<const f16x8*>(s00 + Wt * SP) : zero;
          const f16x8 v11 = (xin && yin) ? *reinterpret_cast<const f16x8*>(s00 + Wt * SP + SP) : zero;
          v = __builtin_elementwise_max(__builtin_elementwise_max(v, v01), __builtin_elementwise_max(v10, v11));
          *reinterpret_cast<f16x8*>(pbase + ((size_t)(b * Hp + (y >> 1)) * Wp + (x >> 1)) * rsb + j * 16) = v;
        }
      }
    }
    if (a.head_w) {  // (workgroup-uniform; BN == coutp: the staged tile holds every channel of its pixels)
      constexpr int KC = BN / 32;
      const int ngroups = (npx + 15) >> 4;
      for (int gq = wave; gq < ngroups; gq += 8) {
        int p = gq * 16 + lpx;
        const bool pv = p < npx;
        p = pv ? p : npx - 1;
        f32x4 hacc[2], haccx[2];
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) hacc[ot] = haccx[ot] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const f16x8 yv = *reinterpret_cast<const f16x8*>(stage + p * SP + (kc * 32 + 8 * lq) * 2);
#pragma unroll
          for (int ot = 0; ot < 2; ++ot) {
            if (ot * 16 < a.head_cout) {
              const f16x8 hw = *reinterpret_cast<const f16x8*>(hw_lds + ((ot * KC + kc) * 64 + lane) * 8);
              const f16x8 hwl = *reinterpret_cast<const f16x8*>(hw_lds + 2 * KC * 512 + ((ot * KC + kc) * 64 + lane) * 8);
              hacc[ot] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hw, yv, hacc[ot], 0, 0, 0);
              haccx[ot] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hwl, yv, haccx[ot], 0, 0, 0);
            }
          }
        }
        const int r = (p * a.rows_inv_wt) >> 20, c = p - r * Wt;
        const int y = y0 + r, x = x0 + c;
        if (pv && y < a.H && x < a.W) {
#pragma unroll
          for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int o = ot * 16 + 4 * lq + i;
              if (o < a.head_cout) {
                float hv = hacc[ot][i] + haccx[ot][i] * SPLIT_INV + a.head_b[o];
                if (a.head_sigmoid) hv = 1.f / (1.f + expf(-hv));
                a.head_dst[(((size_t)b * a.head_cout + o) * a.H + y) * a.W + x] = hv;
              }
            }
        }
      }
    }
    barrier_lds();  // E2
    if (!has_next) break;
    vid = nvid;
    b = nb, y0 = ny0, x0 = nx0, ntile = nnt;
  }
}

// ---------------------------------------------------------------------------------------
// Host side: which (R, Wt, MT, MG) a layer runs with
// ---------------------------------------------------------------------------------------
namespace {
int inv20(int d) { return (int)(((1u << 20) + (unsigned)d - 1) / (unsigned)d); }
bool inv20_exact(int d, int limit) {
  const int inv = inv20(d);
  for (int p = 0; p < limit; ++p)
    if ((int)(((long long)p * inv) >> 20) != p / d) return false;
  return true;
}
}  // namespace

// Estimated launch body in shader cycles, or a negative number when the kernel does not take the shape.  The choice minimises
//   rounds of the chip x (chunks x max(MFMA time of a chunk, loader time of a chunk) + prologue + epilogue)
// over MG in {4 (N tile 64), 2 (N tile 128)}, the tile width (the whole row where it fits seven 16-pixel pieces, else strips) and the rows per tile.
double f16_rows_plan(ConvF16Args& a, int n_cu) {
  if (a.prec != 1 || a.bn != 64 || a.coutp < 64 || (a.coutp & 31) || a.chunks0 <= 0) return -1.0;
  if (a.chunks1 > 0 && !a.src1) return -1.0;
  if (a.src1_lowres && (!a.src1 || (a.H & 1) || (a.W & 1) || a.chunks1 <= 0)) return -1.0;
  if (a.head_w && !(a.head_cout >= 1 && a.head_cout <= 32 && a.head_wcp == a.coutp && (a.coutp == 64 || a.coutp == 128) && a.head_b && a.head_dst && !a.dst_pool)) return -1.0;
  if ((uint64_t)a.B * a.H * a.W >= 0x7FFFFFFFull) return -1.0;
  const bool even = a.dst_pool || a.src1_lowres;
  const int nch = a.chunks0 + a.chunks1;
  double best = -1.0;
  for (int MG = 4; MG >= 2; MG -= 2) {
    const int BN = 256 / MG;
    if (MG == 2 && (a.coutp % 128)) continue;
    if (a.head_w && a.coutp != BN) continue;
    const int ntc = (a.coutp + BN - 1) / BN;
    int widths[8];
    int nw = 0;
    if (a.W + 2 <= 7 * 16) widths[nw++] = a.W;
    for (int wt : {96, 64, 48, 32, 16})
      if (wt < a.W) widths[nw++] = wt;
    for (int wi = 0; wi < nw; ++wi) {
      const int Wt = widths[wi];
      if (even && (Wt & 1)) continue;
      const int HP16 = (Wt + 2 + 15) / 16;
      for (int R = even ? 2 : 1; R <= a.H + (even ? 1 : 0) && R <= 64; R += even ? 2 : 1) {
        const int px = R * Wt;
        int MT = (px + 16 * MG - 1) / (16 * MG);
        if (MT > ((Wt & 15) ? ROWS_MT_UNALIGNED : 9)) break;
        if (MT < 3) MT = 3;
        const int NP = (R + 2) * HP16;
        if (NP > ROWS_MAX_PIECES) break;
        const int SP = BN * 2 + 16;
        const int xb = std::max(0, (px * SP + 1023) / 1024 * 1024 - NP * 1024);
        const int lc16 = a.src1_lowres ? ((Wt >> 1) + 2 + 15) / 16 : 0, lowp = a.src1_lowres ? ((R >> 1) + 2) * lc16 : 0;
        const int hwb = a.head_w ? 2 * 2 * (BN / 32) * 1024 : 0;
        const int lds = 2 * NP * 1024 + xb + 2 * lowp * 1024 + hwb;
        if (lds > ROWS_LDS_LIMIT) continue;
        if (!inv20_exact(Wt, 16 * MT * MG + 16) || !inv20_exact(std::max(Wt >> 1, 1), px / 4 + 16) || !inv20_exact((Wt >> 1) + 1, ((R >> 1) + 1) * ((Wt >> 1) + 1) + 16)) continue;
        const double tiles = (double)a.B * ((a.H + R - 1) / R) * ((a.W + Wt - 1) / Wt);
        const double units = tiles * ntc;
        const double rounds = std::ceil(units / n_cu);
        // calibrated on the cfg5 forward (profiles/r6_f16_rows_*): a chunk takes 1.45 x its MFMA issue time (two MFMA waves per SIMD) or the loaders' time, whichever is
        // longer -- LDS-DMA ~130 cycles of issue per piece and loader wave; a pass of the blend over 256 items ~1,800 cycles in packed fp16, ~4,500 in fp32 beside
        // the MFMA waves --; per tile ~3,000 cycles + its stores at ~10 B per cycle and CU; ~8,000 cycles per launch
        const double mfma = 1.45 * 9.0 * MT * 2 * 16 * 2;
        const double load_full = NP / 4.0 * 130;
        const double items = ((R >> 1) + 1.0) * ((Wt >> 1) + 1.0) * 4;
        const double load_low = std::ceil(items / 256.0) * (a.rows_blend16 ? 1800.0 : 4500.0) + lowp / 4.0 * 130;
        const int n_low = a.src1_lowres ? a.chunks1 : 0;
        const double tile = 3000.0 + px * BN * 2.0 / 10.0 * (a.skip_dst ? 0.3 : 1.0) + (a.dst_pool ? px * BN * 0.5 / 10.0 : 0.0);
        const double pxl = (double)a.B * a.H * a.W;
        const double bytes = pxl * 64.0 * (a.chunks0 + (a.src1_lowres ? 0.25 : 1.0) * a.chunks1) + (a.skip_dst ? 0.0 : pxl * a.coutp * 2.0) + (a.dst_pool ? pxl * a.coutp * 0.5 : 0.0);
        // (tiles of one or two chunks: the un-overlapped prologue / epilogue of a tile weighs more than the model's tile term says -- measured on cfg5's 192 x 192 layers,
        // where the round-2 kernel is 2 - 13 % faster: enc2 39 / 68 us against 45 / 71, the last decoder conv 80 against 87)
        const double short_k = nch <= 2 ? 1.25 : 1.0;
        const double cost = std::max(8000.0 + short_k * rounds * ((nch - n_low) * std::max(mfma, load_full) + n_low * std::max(mfma, load_low) + tile), bytes / 2500.0);
        if (best < 0 || cost < best) {
          best = cost;
          a.rows_r = R;
          a.rows_wt = Wt;
          a.rows_mt = MT;
          a.rows_mg = MG;
          a.rows_hp16 = HP16;
          a.rows_xb = xb;
          a.rows_lc16 = lc16;
          a.rows_lowp = lowp;
          a.rows_inv_wt = inv20(Wt);
          a.rows_inv_pw = inv20(std::max(Wt >> 1, 1));
          a.rows_inv_cc = inv20((Wt >> 1) + 1);
          a.rows_lds = lds;
        }
      }
    }
  }
  return best;
}

template <int MT, int MG, bool ALIGNED>
static int launch_rows_inst(const ConvF16Args& a, dim3 grid, hipStream_t s) {
  static bool attr_done = false;  // (per instantiation; the attribute is idempotent, a race only repeats the call)
  if (!attr_done) {
    PH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_f16_rows_kernel<MT, MG, ALIGNED>), hipFuncAttributeMaxDynamicSharedMemorySize, ROWS_LDS_LIMIT));
    attr_done = true;
  }
  hipLaunchKernelGGL((conv3x3_f16_rows_kernel<MT, MG, ALIGNED>), grid, dim3(768), (size_t)a.rows_lds, s, a);
  return PH_OK;
}

int launch_conv3x3_f16_rows(const ConvF16Args& a, hipStream_t s) {
  PH_REQUIRE(a.rows_r > 0 && a.rows_wt > 0 && a.rows_mt >= 3 && a.rows_mt <= 9 && (a.rows_mg == 2 || a.rows_mg == 4), "conv3x3_f16_rows: no tile plan (call f16_rows_plan first)");
  PH_REQUIRE(a.rows_r * a.rows_wt <= 16 * a.rows_mt * a.rows_mg && (a.rows_r + 2) * a.rows_hp16 <= ROWS_MAX_PIECES && a.rows_lds <= ROWS_LDS_LIMIT, "conv3x3_f16_rows: inconsistent tile plan");
  int n_cu = 0;
  {
    const int rc_cu = device_cu_count(&n_cu);
    if (rc_cu != PH_OK) return rc_cu;
  }
  const int BN = 256 / a.rows_mg;
  const long tiles = (long)a.B * ((a.H + a.rows_r - 1) / a.rows_r) * ((a.W + a.rows_wt - 1) / a.rows_wt);
  const long total = tiles * ((a.coutp + BN - 1) / BN);
  PH_REQUIRE(total < 0x7FFFFFFFl, "conv3x3_f16_rows: too many units");
  const dim3 grid((unsigned)std::min<long>(total, n_cu));
  int rc = PH_OK;
  const bool aligned = (a.rows_wt & 15) == 0;
  PH_REQUIRE(aligned || a.rows_mt <= ROWS_MT_UNALIGNED, "conv3x3_f16_rows: a tile width that is not a multiple of 16 takes at most %d M tiles per wave", ROWS_MT_UNALIGNED);
#define PH_ROWS_CASE(MTV)                                                     \
  case MTV:                                                                   \
    rc = a.rows_mg == 4 ? launch_rows_inst<MTV, 4, true>(a, grid, s) : launch_rows_inst<MTV, 2, true>(a, grid, s); \
    break;
#define PH_ROWS_CASE_U(MTV)                                                   \
  case MTV:                                                                   \
    rc = a.rows_mg == 4 ? launch_rows_inst<MTV, 4, false>(a, grid, s) : launch_rows_inst<MTV, 2, false>(a, grid, s); \
    break;
  if (aligned) {
    switch (a.rows_mt) {
#ifndef PH_ROWS_FEW  // (diagnostic builds on the GPU box: only the forms the cfg5 forward runs)
      PH_ROWS_CASE(3)
      PH_ROWS_CASE(4)
      PH_ROWS_CASE(5)
      PH_ROWS_CASE(7)
      PH_ROWS_CASE(8)
#endif
      PH_ROWS_CASE(6)
      PH_ROWS_CASE(9)
      default:
        set_error("conv3x3_f16_rows: M tile count %d not built", a.rows_mt);
        return PH_E_INVALID;
    }
  } else {
    switch (a.rows_mt) {
#ifndef PH_ROWS_FEW
      PH_ROWS_CASE_U(3)
      PH_ROWS_CASE_U(4)
#endif
      PH_ROWS_CASE_U(5)
      default:
        set_error("conv3x3_f16_rows: M tile count %d not built", a.rows_mt);
        return PH_E_INVALID;
    }
  }
#undef PH_ROWS_CASE
#undef PH_ROWS_CASE_U
  if (rc != PH_OK) return rc;
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

}  // namespace ph
