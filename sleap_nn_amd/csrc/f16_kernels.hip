// 3x3 convolution on the fp16 matrix pipe of gfx950 (v_mfma_f32_32x32x16_f16, fp32 accumulate) and the format-generic
// companions of the encoder-decoder forward (bilinear x2, 2x2 max pool, 1x1 heads, slot read-back).
//
// Two precisions share one kernel (activation formats: act_format.h):
//   PREC 3  "split fp16": every operand is the pair (hi, lo' = (x - hi) 2^11) of fp16 numbers; a product is
//           a_hi b_hi + (a_hi b_lo' + a_lo' b_hi) 2^-11 -- three MFMAs into two fp32 accumulators (main, cross) -- i.e.
//           22-bit products with fp32 accumulation.  On the cfg3 network the heads differ from the exact-fp32 path by
//           ~1e-6 of their scale (the same distance the exact path has from ATen's CPU kernels), far inside the 1e-4
//           parity bar, at 16/3 of the fp32-MFMA rate: the convolution stack turns from MFMA-bound into
//           LDS-fill / HBM-bound.
//   PREC 1  plain fp16 operands (the reference's autocast mode; its tolerance is 5e-3): one MFMA per product, fp16 storage.
//
// Reference semantics: conv3x3 "same" + bias + ReLU (+ fused 2x2 max pool), concat(skip, x) as two K panels
// (sleap_nn/architectures/encoder_decoder.py:108-121,494-510,545,556; common.py:69-107).
//
// Kernel structure = conv3x3_mfma_dma_persist_kernel (net_kernels.hip): persistent 512-thread workgroups, 16 x 32 pixel
// tiles, LDS-DMA staging of the (16+2) x (32+2) halo and the 9 x BN weight rows of one K chunk into a double-buffered
// ring, one barrier per chunk.  A K chunk is 64 B per pixel / weight row: 16 channels x (hi, lo') in split mode, 32
// channels in plain mode, as four 16-B quads; lanes 0-31 read quad g*2, lanes 32-63 quad g*2 + 1 -- exactly the fragment
// v_mfma_f32_32x32x16_f16 wants (lane l: row l & 31, k = 8 (l >> 5) .. + 7), so the conflict-free quad-major LDS image
// of the fp32 kernel serves unchanged.  The product is accumulated TRANSPOSED (weights are the A operand): a lane owns
// one pixel and each accumulator register quad is four consecutive output channels, which the epilogue converts and
// stores as one 8-byte piece of the destination's format.
#include <algorithm>
#include <cmath>
#include <type_traits>

#include "act_format.h"
#include "common.h"
#include "f16_kernels.h"
#include "net_kernels.h"

namespace ph {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int TW = 32, TH = 16, HALO_W = TW + 2, HALO_H = TH + 2;
constexpr int NPIX = HALO_H * HALO_W;        // 612
constexpr int A_PIECES = (NPIX + 15) / 16;   // 39
}  // namespace

__device__ __forceinline__ f16x8 as_h8(const f32x4& v) { return __builtin_bit_cast(f16x8, v); }

// quad_perm(1, 0, 3, 2): the value of the neighbouring lane (x ^ 1)
__device__ __forceinline__ float lane_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}

template <int BN, int PREC, bool HEAD = false>
__global__ __launch_bounds__(512, 2) void conv3x3_f16_persist_kernel(ConvF16Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NT = BN / 32;
  constexpr int B_PIECES = 9 * BN / 16;
  constexpr int PIECES = A_PIECES + B_PIECES;
  constexpr int A_SLOTS = (A_PIECES + 7) / 8;   // 5
  constexpr int B_SLOTS = (B_PIECES + 7) / 8;   // 5 (BN 64) / 3 (BN 32)
  constexpr int SLOTS = A_SLOTS + B_SLOTS;      // <= 10: one DMA piece per tap, the surplus ones ride in tap 0
  constexpr int BUF_FLOATS = PIECES * 256;
  constexpr int DST_FMT = PREC == 3 ? FMT_SPLIT : FMT_F16;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (a.W + TW - 1) / TW;
  const int tiles_y = (a.H + TH - 1) / TH;
  const int tiles = tiles_x * tiles_y * a.B;
  const int ntc = (a.coutp + BN - 1) / BN;
  const int total = tiles * ntc;
  const int nchunks = a.chunks0 + a.chunks1;
  const int dq = lane >> 4, dr = lane & 15;
  const int lx = lane & 31, lh = lane >> 5;
  unsigned long long probe_t0 = 0, probe_r0 = 0;
  if (a.clock_probe) {
    probe_t0 = __builtin_amdgcn_s_memtime();
    probe_r0 = __builtin_amdgcn_s_memrealtime();
  }

  struct Plan {
    int a_pix[A_SLOTS];
    unsigned a_ok;
    int b, x0, y0, ntile;
  };
  auto setup = [&](int vid, Plan& P) {
    int t, ntile;  // XCD-aware dealing (see deal_tile in net_kernels.hip): every XCD walks a contiguous range of pixel tiles
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
    P.b = t / tiles_y;
    P.x0 = tx * TW;
    P.y0 = ty * TH;
    P.ntile = ntile;
    P.a_ok = 0;
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) {
      const int p = min(wave + 8 * s, A_PIECES - 1);
      const int pix = p * 16 + dr;
      const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
      const int gy = P.y0 + hy - 1, gx = P.x0 + hx - 1;
      const bool in = (pix < NPIX) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      P.a_ok |= (in ? 1u : 0u) << s;
      const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
      P.a_pix[s] = (P.b * a.H + cy) * a.W + cx;
    }
  };

  int f_pix[A_SLOTS];
  unsigned f_ok = 0;
  const float* p_src = a.src0;
  const float* p_w = a.wpack;
  int p_rs = a.rs0, p_coff = 0;
  auto select_fetch = [&](const Plan& P, int ch) {
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) f_pix[s] = P.a_pix[s];
    f_ok = P.a_ok;
    if (ch < a.chunks0) {
      p_src = a.src0;
      p_rs = a.rs0;
      p_coff = ch * 16;
    } else {
      p_src = a.src1;
      p_rs = a.rs1;
      p_coff = (ch - a.chunks0) * 16;
    }
    p_w = a.wpack + ((size_t)P.ntile * nchunks + ch) * (9 * BN * 16);
  };
  auto dma_slot = [&](int s, float* buf) {
    const float* g;
    int p;
    if (s < A_SLOTS) {  // compile-time after unrolling
      p = min(wave + 8 * s, A_PIECES - 1);
      const float* real = p_src + (size_t)f_pix[s] * p_rs + p_coff + dq * 4;
      const float* zero = a.zeros + dq * 4;
      g = ((f_ok >> s) & 1u) ? real : zero;
    } else {
      const int pb = min(wave + 8 * (s - A_SLOTS), B_PIECES - 1);
      p = A_PIECES + pb;
      g = p_w + pb * 256 + lane * 4;  // weights are packed in LDS order
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(buf + p * 256), 16, 0, 0);
  };

  // fused head: its weights as fp16 (hi, lo' = (w - hi) 2^11) fragments behind the two ring buffers, [hi | lo'][K step 4][K block lh 2][head channel 32][8]: 8 KiB, written once
  // (the barrier in front of the K loop publishes them)
  _Float16* const hw_lds = reinterpret_cast<_Float16*>(lds + 2 * BUF_FLOATS);
  if constexpr (HEAD) {
    if (tid < 256) {
      const int ks = tid >> 6, hh = (tid >> 5) & 1, o = tid & 31;
      f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0;
      if (o < a.head_cout) {
        w0 = *reinterpret_cast<const f32x4*>(a.head_w + (size_t)o * a.head_wcp + 16 * ks + 8 * hh);
        w1 = *reinterpret_cast<const f32x4*>(a.head_w + (size_t)o * a.head_wcp + 16 * ks + 8 * hh + 4);
      }
      f16x8 hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        hi[k] = split_hi(w0[k]);
        lo[k] = split_lo(w0[k], hi[k]);
        hi[4 + k] = split_hi(w1[k]);
        lo[4 + k] = split_lo(w1[k], hi[4 + k]);
      }
      *reinterpret_cast<f16x8*>(hw_lds + tid * 8) = hi;
      *reinterpret_cast<f16x8*>(hw_lds + 2048 + tid * 8) = lo;
    }
  }

  // fragment read offsets (floats, buffer-relative): pixel (row 2w + rr, x + kx) quad lh (+ 2 g)
  int offA[4][3];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int pix = (2 * wave + rr) * HALO_W + lx + kx;
      offA[rr][kx] = (pix >> 4) * 256 + lh * 64 + (pix & 15) * 4;
    }
  const int offB = A_PIECES * 256 + (lx >> 4) * 256 + lh * 64 + (lx & 15) * 4;  // + tap*(BN/16)*256 + n*512 + g*128
  float* buf0 = lds;
  float* buf1 = lds + BUF_FLOATS;

  // Waves i and i + 4 share SIMD i, and an LDS-DMA wave-instruction blocks its wave for 60-450 cycles (the CU's address path
  // takes ~56 cycles per 1-KiB piece): partners must not issue theirs at the same point of a tap, or the SIMD's matrix pipe
  // idles meanwhile.  The upper four waves ("late") place the tap's piece after 2/3 of its MFMAs, the lower four after 1/6.
  auto run = [&](auto late) {
  constexpr bool LATE = decltype(late)::value;
  Plan P, Pn;
  int vid = blockIdx.x;
  setup(vid, P);
  select_fetch(P, 0);
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) dma_slot(s, buf0);
  __syncthreads();
  int parity = 0;
  while (true) {
    const int nvid = vid + gridDim.x;
    const bool has_next = nvid < total;  // workgroup-uniform
    if (has_next) setup(nvid, Pn);
    f32x16 acc[2][NT];   // main:  hi x hi   (plain mode: the only accumulator)
    f32x16 accx[PREC == 3 ? 2 : 1][PREC == 3 ? NT : 1];  // cross: hi x lo' + lo' x hi, scaled by 2^11
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[m][n][r] = 0.f;
          if (PREC == 3) accx[m][n][r] = 0.f;
        }
    for (int ch = 0; ch < nchunks; ++ch) {
      float* cur = ((parity + ch) & 1) ? buf1 : buf0;
      float* nxt = ((parity + ch) & 1) ? buf0 : buf1;
      if (ch + 1 < nchunks)
        select_fetch(P, ch + 1);
      else if (has_next)
        select_fetch(Pn, 0);  // the next tile's first chunk rides under this tile's last one
      else
        select_fetch(P, ch);  // nothing left: refetch (harmless, keeps the loop branch-free inside)
      f32x4 af[2][2][2], bf[2][NT][2];  // [buffer][m | n][g]: g = 0 hi / first 16 channels, g = 1 lo' / second 16 channels
      auto load_frags = [&](int tap, int fb) {
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
          for (int m = 0; m < 2; ++m) af[fb][m][g] = *reinterpret_cast<const f32x4*>(cur + offA[m + ky][kx] + g * 128);
#pragma unroll
          for (int n = 0; n < NT; ++n) bf[fb][n][g] = *reinterpret_cast<const f32x4*>(cur + offB + tap * (BN / 16) * 256 + n * 512 + g * 128);
        }
      };
      load_frags(0, 0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int fc = tap & 1;
        if (tap + 1 < 9) load_frags(tap + 1, fc ^ 1);
        if (tap < SLOTS) dma_slot(tap, nxt);
        if (tap == 0 && SLOTS > 9) dma_slot(9, nxt);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            // transposed product: A = weight rows (output channels), B = pixels
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(bf[fc][n][0]), as_h8(af[fc][m][0]), acc[m][n], 0, 0, 0);
            if (PREC == 3) {
              accx[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(bf[fc][n][0]), as_h8(af[fc][m][1]), accx[m][n], 0, 0, 0);
              accx[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(bf[fc][n][1]), as_h8(af[fc][m][0]), accx[m][n], 0, 0, 0);
            } else {
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(bf[fc][n][1]), as_h8(af[fc][m][1]), acc[m][n], 0, 0, 0);
            }
          }
        // pinned order: a first MFMA, then the next tap's LDS reads (the wait in front of a tap's first MFMA is lgkmcnt(0)
        // in a kernel that issues LDS-DMA, so reads issued before it would be drained on the spot), the DMA piece in the middle
        constexpr int MF = (PREC == 3 ? 3 : 2) * 2 * NT;
        constexpr int BEFORE = LATE ? (2 * MF) / 3 : MF / 6;  // MFMAs between the LDS reads and the DMA piece
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4 + 2 * NT, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, BEFORE, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, MF - 1 - BEFORE, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();  // vmcnt(0) + barrier: the next buffer landed everywhere, this one is free again
    }
    // ---- epilogue: D row = channel (r & 3) + 8 (r >> 2) + 4 lh of the N tile, column = pixel lx of the wave's row.
    // A lane holds 4-channel pieces of ONE pixel; stored directly, a wave-instruction would touch 32 different cache lines with
    // 16 B each (measured: 3.1 of the 8.5 ms conv time at cfg3 was this store pattern).  The pieces are therefore first laid
    // out in LDS exactly as the tile lies in memory (the K loop's last buffer is free; every wave has a private 32-pixel region,
    // rows padded by 16 B against bank conflicts), read back 16 B per lane and stored with 64 lanes covering whole 128-B lines.
    const int b = P.b, x0 = P.x0, y0 = P.y0, ntile = P.ntile;
    const int x = x0 + lx;
    const size_t rsb = (size_t)a.rs_dst * 4;  // destination bytes per pixel
    const bool x_ok = x < a.W;
    constexpr int SEG = DST_FMT == FMT_SPLIT ? BN * 4 : BN * 2;  // bytes of this N tile per pixel
    constexpr int SEGP = SEG + 16;
    constexpr int PER = SEG / 16;                                // 16-B units per pixel
    static_assert(8 * 32 * SEGP <= PIECES * 1024, "epilogue staging must fit one ring buffer");
    char* const my = reinterpret_cast<char*>(((parity + nchunks - 1) & 1) ? buf1 : buf0) + wave * (32 * SEGP);
    float v[2][NT][16];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int cb = ntile * BN + n * 32 + 4 * lh;  // + 8 q: the lane's channel quads
      f32x4 bias4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bias4[q] = *reinterpret_cast<const f32x4*>(a.bias + cb + 8 * q);  // bias is padded to a multiple of BN
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float t = acc[m][n][r];
          if (PREC == 3) t += accx[m][n][r] * SPLIT_INV;
          t += bias4[r >> 2][r & 3];
          v[m][n][r] = a.relu ? fmaxf(t, 0.f) : t;
        }
    }
    auto convert4 = [&](const float* src4, f16x4& hi, f16x4& lo) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const _Float16 h = split_hi(src4[k]);
        hi[k] = h;
        if constexpr (DST_FMT == FMT_SPLIT) lo[k] = split_lo(src4[k], h);
      }
    };
    // fused 1x1 head (plain fp16, one N tile of 64 = every channel of a pixel in this workgroup): the staged fp16 row [32 pixels][64 channels] of the wave is the B operand of eight
    // v_mfma_f32_32x32x16_f16 (lane = pixel lx, K block lh: one ds_read_b128 per K step), the head's fp32 weights [o][c] -- converted on the way, rows beyond head_cout zero -- the A
    // operand; D = lane (pixel, head channels 8 (i >> 2) + 4 lh + (i & 3)): NCHW stores, 32 consecutive pixels per head channel
    constexpr bool fused_head = HEAD && PREC == 1 && BN == 64;  // (its own instantiation: the sixteen accumulator registers of the head cost the plain kernel eleven spills)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if (a.skip_dst && !fused_head) break;  // nobody reads the full-resolution output: only the pooled one below
      const int y = y0 + 2 * wave + m;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f16x4 hi, lo;
          convert4(&v[m][n][4 * q], hi, lo);
          if constexpr (DST_FMT == FMT_SPLIT) {
            char* p = my + lx * SEGP + (2 * n + (q >> 1)) * 64 + (q & 1) * 16 + lh * 8;
            *reinterpret_cast<f16x4*>(p) = hi;
            *reinterpret_cast<f16x4*>(p + 32) = lo;
          } else {
            *reinterpret_cast<f16x4*>(my + lx * SEGP + (n * 32 + 8 * q + 4 * lh) * 2) = hi;
          }
        }
      const bool y_ok = y < a.H;
      if constexpr (fused_head) {
        {
          f32x16 hacc, haccx;  // weights as (hi, lo' = (w - hi) 2^11): the separate head kernel multiplies the same fp16-rounded activations by fp32 weights
#pragma unroll
          for (int r = 0; r < 16; ++r) hacc[r] = haccx[r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const f16x8 yv = *reinterpret_cast<const f16x8*>(my + lx * SEGP + (16 * ks + 8 * lh) * 2);
            const f16x8 hw = *reinterpret_cast<const f16x8*>(hw_lds + ((ks * 2 + lh) * 32 + lx) * 8);
            const f16x8 hwl = *reinterpret_cast<const f16x8*>(hw_lds + 2048 + ((ks * 2 + lh) * 32 + lx) * 8);
            hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hw, yv, hacc, 0, 0, 0);
            haccx = __builtin_amdgcn_mfma_f32_32x32x16_f16(hwl, yv, haccx, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = 8 * (r >> 2) + 4 * lh + (r & 3);
            if (o < a.head_cout && y_ok && x_ok) {
              float hv = hacc[r] + haccx[r] * SPLIT_INV + a.head_b[o];
              if (a.head_sigmoid) hv = 1.f / (1.f + expf(-hv));
              a.head_dst[(((size_t)b * a.head_cout + o) * a.H + y) * a.W + x] = hv;
            }
          }
        }
      }
      if (a.skip_dst) continue;
      char* const drow = reinterpret_cast<char*>(a.dst) + ((size_t)(b * a.H + (y_ok ? y : 0)) * a.W + x0) * rsb + (size_t)ntile * SEG;
#pragma unroll
      for (int it = 0; it < 32 * PER / 64; ++it) {
        const int idx = it * 64 + lane;
        const int pp = idx / PER, j = idx - pp * PER;
        const f32x4 piece = *reinterpret_cast<const f32x4*>(my + pp * SEGP + j * 16);
        const int c_first = ntile * BN + (DST_FMT == FMT_SPLIT ? (j >> 2) * 16 : j * 8);  // first channel the 16-B unit belongs to
        // non-temporal: the tile's output is not read again by this launch, and kept in L2 it would evict the weight panels the K loop re-reads
        if (y_ok && x0 + pp < a.W && c_first < a.coutp) __builtin_nontemporal_store(piece, reinterpret_cast<f32x4*>(drow + (size_t)pp * rsb + j * 16));
      }
    }
    if (!fused_head && a.dst_pool) {  // (a fused head excludes the pool: launch_conv3x3_f16)  fused 2x2/2 max pool ("same": zeros beyond the image; values are >= 0 after the ReLU)
      const int Hp = (a.H + 1) >> 1, Wp = (a.W + 1) >> 1;
      const int yt = y0 + 2 * wave;
      const bool y1_ok = yt + 1 < a.H;
      const bool p_ok = !(lx & 1) && yt < a.H && x_ok;
      char* const prow = reinterpret_cast<char*>(a.dst_pool) + ((size_t)(b * Hp + (yt >> 1)) * Wp + (x >> 1)) * rsb;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float pv[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float c0 = x_ok ? v[0][n][4 * q + k] : 0.f, c1 = (x_ok && y1_ok) ? v[1][n][4 * q + k] : 0.f;
            const float col = fmaxf(c0, c1);
            pv[k] = fmaxf(col, lane_xor1(col));
          }
          const int c = ntile * BN + n * 32 + 8 * q + 4 * lh;
          if (p_ok && c < a.coutp) {
            f16x4 hi, lo;
            convert4(pv, hi, lo);
            if constexpr (DST_FMT == FMT_SPLIT) {
              char* p = prow + (c >> 4) * 64 + ((c >> 3) & 1) * 16 + (c & 7) * 2;
              *reinterpret_cast<f16x4*>(p) = hi;
              *reinterpret_cast<f16x4*>(p + 32) = lo;
            } else {
              *reinterpret_cast<f16x4*>(prow + c * 2) = hi;
            }
          }
        }
    }
    __builtin_amdgcn_s_barrier();  // every wave has read its staging region back: the buffer may take the next tile's DMA pieces
    if (!has_next) break;
    parity = (parity + nchunks) & 1;
    vid = nvid;
    P = Pn;
  }
  };
  if (wave >= 4)  // wave-uniform; both variants execute the same barriers
    run(std::true_type{});
  else
    run(std::false_type{});
  if (a.clock_probe && tid == 0) {  // shader clock under this load = d memtime / d memrealtime * 100 MHz
    a.clock_probe[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - probe_t0;
    a.clock_probe[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - probe_r0;
  }
}

// fp32 quad-major weight panels (the LDS-DMA layout of the fp32 kernels: [panel][piece = row / 16][quad k / 4][row % 16][4]) ->
// fp16 panels in the same piece structure, quads = [hi k0-7][hi k8-15][lo' k0-7][lo' k8-15] (split) of one 16-channel
// source panel, or [k0-7][k8-15][k16-23][k24-31] (plain) of two consecutive ones (`pair` = 1; an odd tail is zero filled).
__global__ __launch_bounds__(256) void f16_weight_pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int n_tiles, int src_chunks0, int src_chunks1, int rows,
                                                              int plain) {
  const int pieces = rows / 16;
  const int src_chunks = src_chunks0 + src_chunks1;
  const int d0 = plain ? (src_chunks0 + 1) / 2 : src_chunks0, d1 = plain ? (src_chunks1 + 1) / 2 : src_chunks1;
  const int dst_chunks = d0 + d1;
  const size_t total = (size_t)n_tiles * dst_chunks * pieces * 64;  // (piece row, quad) items
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int q = (int)(i & 3), r = (int)((i >> 2) & 15);
    size_t t = i >> 6;
    const int piece = (int)(t % pieces);
    t /= pieces;
    const int dch = (int)(t % dst_chunks), nt = (int)(t / dst_chunks);
    // which fp32 source chunk and which 8 of its 16 channels feed this quad
    int sch, k0, lo = 0;
    bool valid = true;
    if (!plain) {
      sch = dch;
      k0 = (q & 1) * 8;
      lo = q >> 1;
    } else {
      const bool second = dch >= d0;
      const int local = second ? dch - d0 : dch, base = second ? src_chunks0 : 0, cnt = second ? src_chunks1 : src_chunks0;
      const int sl = 2 * local + (q >> 1);
      valid = sl < cnt;
      sch = base + min(sl, cnt - 1);
      k0 = (q & 1) * 8;
    }
    const float* s = src + (((size_t)nt * src_chunks + sch) * pieces + piece) * 256 + r * 4;
    f16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int kk = k0 + k;
      const float v = valid ? s[(kk >> 2) * 64 + (kk & 3)] : 0.f;
      const _Float16 hi = split_hi(v);
      o[k] = lo ? split_lo(v, hi) : hi;
    }
    *reinterpret_cast<f16x8*>(dst + (((size_t)nt * dst_chunks + dch) * pieces + piece) * 256 + q * 64 + r * 4) = o;
  }
}

int64_t f16_weight_pack_floats(int n_tiles, int chunks0, int chunks1, int bn, int plain) {
  const int d = plain ? (chunks0 + 1) / 2 + (chunks1 + 1) / 2 : chunks0 + chunks1;
  return (int64_t)n_tiles * d * 9 * bn * 16;
}

int launch_f16_weight_pack(const float* w_dma_f32, float* dst, int n_tiles, int chunks0, int chunks1, int bn, int plain, hipStream_t s) {
  const int64_t items = f16_weight_pack_floats(n_tiles, chunks0, chunks1, bn, plain) / 4;
  hipLaunchKernelGGL(f16_weight_pack_kernel, dim3((unsigned)std::min<int64_t>((items + 255) / 256, 4096)), dim3(256), 0, s, w_dma_f32, dst, n_tiles, chunks0, chunks1, 9 * bn,
                     plain);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// Estimated launch body of conv3x3_f16_persist_kernel in shader cycles (the routing weighs conv3x3_f16_rows_kernel's plan against it): rounds of the chip x (fixed
// tile cost + cycles per K chunk), calibrated on the per-op table of the cfg5 forward (profiles/r4z_f16_cfg5_per_op.txt: 2 chunks 12.5 k, 8 chunks 68 k per unit),
// and never below the time the layer's algorithmic bytes take at ~5 TB/s (2,500 bytes per shader cycle chip-wide at 2 GHz)
double f16_conv_cost(const ConvF16Args& a, int n_cu) {
  const double tiles = (double)((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH) * a.B;
  const double units = tiles * ((a.coutp + a.bn - 1) / a.bn);
  const double rounds = std::ceil(units / n_cu);
  const int nch = a.chunks0 + a.chunks1;
  const double px = (double)a.B * a.H * a.W;
  const double bytes = px * 64.0 * nch + (a.skip_dst ? 0.0 : px * a.coutp * 2.0) + (a.dst_pool ? px * a.coutp * 0.5 : 0.0);
  return std::max(rounds * (6000.0 + 7800.0 * nch * (a.bn == 64 ? 1.0 : 0.6)), bytes / 2500.0);
}
// ... of upsample2x_fmt_kernel on a (B, H, W, cp) fp16 source: 1.25 x the output bytes at ~5 TB/s + the launch boundary
double f16_upsample_cost(int B, int H, int W, int cp, int n_cu) {
  (void)n_cu;
  return (double)B * H * W * cp * 2.0 * 5.0 / 2500.0 + 3000.0;
}

int launch_conv3x3_f16(const ConvF16Args& a, hipStream_t s) {
  PH_REQUIRE(a.prec == 1 || a.prec == 3, "conv3x3_f16: precision must be 1 (fp16) or 3 (split fp16)");
  PH_REQUIRE(a.bn == 32 || a.bn == 64, "conv3x3_f16: N tile must be 32 or 64");
  PH_REQUIRE(a.chunks0 > 0 && (a.chunks1 == 0 || a.src1), "conv3x3_f16: bad sources");
  PH_REQUIRE(!a.head_w || (a.prec == 1 && a.bn == 64 && a.coutp == 64 && !a.dst_pool && a.head_cout >= 1 && a.head_cout <= 32 && a.head_wcp == 64 && a.head_b && a.head_dst),
             "conv3x3_f16: a fused head needs the plain fp16 precision on ONE N tile of 64 output channels, at most 32 head channels, no fused pool");
  int n_cu = 0;  // persistent workgroups: one per CU of the current device
  {
    const int rc_cu = device_cu_count(&n_cu);
    if (rc_cu != PH_OK) return rc_cu;
  }
  const int tiles = ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH) * a.B;
  const int total = tiles * ((a.coutp + a.bn - 1) / a.bn);
  const size_t lds = (size_t)2 * (A_PIECES + 9 * a.bn / 16) * 1024 + (a.head_w ? 8192 : 0);  // (+ the fused head's weight fragments)
  const dim3 grid(std::min(total, n_cu));
  if (a.bn == 64 && a.prec == 3)
    hipLaunchKernelGGL((conv3x3_f16_persist_kernel<64, 3>), grid, dim3(512), lds, s, a);
  else if (a.bn == 32 && a.prec == 3)
    hipLaunchKernelGGL((conv3x3_f16_persist_kernel<32, 3>), grid, dim3(512), lds, s, a);
  else if (a.bn == 64 && a.head_w)
    hipLaunchKernelGGL((conv3x3_f16_persist_kernel<64, 1, true>), grid, dim3(512), lds, s, a);
  else if (a.bn == 64)
    hipLaunchKernelGGL((conv3x3_f16_persist_kernel<64, 1>), grid, dim3(512), lds, s, a);
  else
    hipLaunchKernelGGL((conv3x3_f16_persist_kernel<32, 1>), grid, dim3(512), lds, s, a);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// Format-generic companions (8 channels per thread through load8 / store8)
// ---------------------------------------------------------------------------------------

// bilinear x2, align_corners=False (ATen upsample_bilinear2d: src = max(0, (dst + 0.5) * 0.5 - 0.5)); one thread = 8 channels x
// the 2x2 output pixels between input pixels (i..i+1, j..j+1), i, j from -1 (see upsample2x_kernel in net_kernels.hip)
// `f16math` (plain fp16 only; handle option "upsample_f16math"): the blend in packed fp16 arithmetic with ATen's scale-2 weights as constants -- 3/4 of the nearer
// source pixel, 1/4 of the farther one (at the border both taps are the same clamped pixel, so the constants reproduce the border rule); fma(a, 3/4, b / 4) is the
// exact one-dimensional blend rounded once, the horizontal result is rounded to fp16 before the vertical blend.  The arithmetic, operation for operation, of the
// blend conv3x3_f16_rows_kernel's loader waves do when the up-sampling is folded into the conv (f16_rows_kernels.hip: blend16): either plan gives the same bits.
template <int FMT>
__global__ __launch_bounds__(256) void upsample2x_fmt_kernel(const void* __restrict__ src, void* __restrict__ dst, int B, int H, int W, int cp, int f16math) {
  const int Ho = 2 * H, Wo = 2 * W, groups = cp >> 3;
  const int nI = H + 1, nJ = W + 1;
  const size_t total = (size_t)B * nI * nJ * groups;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(idx % groups);
    size_t p = idx / groups;
    const int j = (int)(p % nJ) - 1;
    p /= nJ;
    const int i = (int)(p % nI) - 1;
    const int b = (int)(p / nI);
    const int r0 = max(i, 0), r1 = min(i + 1, H - 1), c0 = max(j, 0), c1 = min(j + 1, W - 1);
    const size_t base = (size_t)b * H * W;
    if constexpr (FMT == FMT_F16) {
      if (f16math) {
        auto at = [&](int r, int c) { return *reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(src) + ((base + (size_t)r * W + c) * cp + 8 * g) * 2); };
        const f16x8 h00 = at(r0, c0), h01 = at(r0, c1), h10 = at(r1, c0), h11 = at(r1, c1);
        f16x8 c75, c25;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          c75[k] = (_Float16)0.75f;
          c25[k] = (_Float16)0.25f;
        }
        const f16x8 q00 = h00 * c25, q01 = h01 * c25, q10 = h10 * c25, q11 = h11 * c25;
        const f16x8 t0[2] = {__builtin_elementwise_fma(h00, c75, q01), __builtin_elementwise_fma(h01, c75, q00)};
        const f16x8 t1[2] = {__builtin_elementwise_fma(h10, c75, q11), __builtin_elementwise_fma(h11, c75, q10)};
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int x = 2 * j + 1 + dx;
          const f16x8 o0 = __builtin_elementwise_fma(t0[dx], c75, t1[dx] * c25), o1 = __builtin_elementwise_fma(t1[dx], c75, t0[dx] * c25);
#pragma unroll
          for (int dy = 0; dy < 2; ++dy) {
            const int y = 2 * i + 1 + dy;
            if (y >= 0 && y < Ho && x >= 0 && x < Wo)
              *reinterpret_cast<f16x8*>(reinterpret_cast<char*>(dst) + ((((size_t)b * Ho + y) * Wo + x) * cp + 8 * g) * 2) = dy ? o1 : o0;
          }
        }
        continue;
      }
    }
    float v00[8], v01[8], v10[8], v11[8];
    load8<FMT>(src, base + (size_t)r0 * W + c0, cp, g, v00);
    load8<FMT>(src, base + (size_t)r0 * W + c1, cp, g, v01);
    load8<FMT>(src, base + (size_t)r1 * W + c0, cp, g, v10);
    load8<FMT>(src, base + (size_t)r1 * W + c1, cp, g, v11);
#pragma unroll
    for (int dy = 1; dy <= 2; ++dy) {
      const int y = 2 * i + dy;
      if (y < 0 || y >= Ho) continue;
      const float sy = fmaxf((y + 0.5f) * 0.5f - 0.5f, 0.f);
      const float ly = sy - (float)(int)sy, hy = 1.f - ly;
#pragma unroll
      for (int dx = 1; dx <= 2; ++dx) {
        const int x = 2 * j + dx;
        if (x < 0 || x >= Wo) continue;
        const float sx = fmaxf((x + 0.5f) * 0.5f - 0.5f, 0.f);
        const float lx = sx - (float)(int)sx, hx = 1.f - lx;
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = hy * (hx * v00[k] + lx * v01[k]) + ly * (hx * v10[k] + lx * v11[k]);
        store8<FMT>(dst, ((size_t)b * Ho + y) * Wo + x, cp, g, o);
      }
    }
  }
}

// 2x2/2 max pool, zero pad bottom/right when odd (common.py:93-96)
template <int FMT>
__global__ __launch_bounds__(256) void pool2x2_fmt_kernel(const void* __restrict__ src, void* __restrict__ dst, int B, int H, int W, int cp) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, groups = cp >> 3;
  const size_t total = (size_t)B * Ho * Wo * groups;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(idx % groups);
    size_t p = idx / groups;
    const int x = (int)(p % Wo);
    p /= Wo;
    const int y = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const int y1 = 2 * y + 1, x1 = 2 * x + 1;
    const size_t base = (size_t)b * H * W;
    float m[8], t[8];
    load8<FMT>(src, base + (size_t)(2 * y) * W + 2 * x, cp, g, m);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int yy = c == 0 ? 2 * y : y1, xx = c == 1 ? 2 * x : x1;
      if (yy < H && xx < W) {
        load8<FMT>(src, base + (size_t)yy * W + xx, cp, g, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], t[k]);
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], 0.f);
      }
    }
    store8<FMT>(dst, ((size_t)b * Ho + y) * Wo + x, cp, g, m);
  }
}

// 1x1 head convolution on the fp32 matrix pipe (exact fp32 products whatever the activation format), activations in FMT ->
// NCHW fp32 (+ optional sigmoid) (architectures/heads.py:58-67).  Transposed product D[channel][pixel] = sum_c W[channel][c]
// X[pixel][c]: the weights (A operand) sit in LDS, a lane's B operand is 16 B of "its" pixel (channels 8g + 4 lh .. + 3) read
// straight from the activation tensor -- every 128-B line of a pixel is touched by four consecutive K groups of the same
// lanes, L1 hits -- and the result lands with a lane per pixel and a register per channel: each register is stored as 32
// consecutive floats of one NCHW channel plane.  HBM-bound: Cp * bytes/channel read + Cout * 4 B written per pixel.
//   256 threads = 4 waves; a wave owns 64 pixels (two 32-pixel N tiles) per trip and all ceil(cout / 32) M tiles.
template <int FMT, int MT>
__global__ __launch_bounds__(256) void head1x1_mfma_kernel(const void* __restrict__ src, const float* __restrict__ w /* [cout][wcp] */, const float* __restrict__ bias,
                                                           float* __restrict__ dst, int B, int HW, int cp, int wcp, int cout, int sigmoid) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // MT * 32 rows x (wcp + 4): the row pad keeps the b128 fragment reads conflict-free
  const int ldw = wcp + 4;
  for (int i = threadIdx.x; i < MT * 32 * (wcp / 4); i += 256) {
    const int row = i / (wcp / 4), q = i - row * (wcp / 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < cout) v = *reinterpret_cast<const f32x4*>(w + (size_t)row * wcp + q * 4);
    *reinterpret_cast<f32x4*>(lds + row * ldw + q * 4) = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lx = lane & 31, lh = lane >> 5;
  const size_t npix = (size_t)B * HW;
  const size_t n_groups = (npix + 63) / 64;
  for (size_t grp = (size_t)blockIdx.x * 4 + wave; grp < n_groups; grp += (size_t)gridDim.x * 4) {
    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    size_t pix[2];
    bool ok[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      pix[n] = grp * 64 + n * 32 + lx;
      ok[n] = pix[n] < npix;
      pix[n] = ok[n] ? pix[n] : npix - 1;
    }
    // K in chunks of four groups = 32 channels = ONE 128-byte line per pixel (fp32): the eight loads of a chunk (two pixels per lane) are issued together, one chunk ahead of
    // the MFMAs that read them.  (One load per group and pixel, each waited for before its four MFMAs, left a wave with one request in flight: the launch ran at the memory
    // LATENCY -- 0.10 ms for cfg3's PAF head -- and every line was fetched four times, 16 bytes per visit, from whatever level still held it.)
    const int n_g = wcp / 8;
    auto load_chunk = [&](int g0, f32x4 (&dst)[4][2]) __attribute__((always_inline)) {
#pragma unroll
      for (int gi = 0; gi < 4; ++gi)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int g = min(g0 + gi, n_g - 1);  // (a chunk past the end re-reads the last group: its MFMAs are skipped)
          const int c0 = 8 * g + 4 * lh;
          if constexpr (FMT == FMT_F32) {
            dst[gi][n] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(src) + pix[n] * cp + c0);
          } else if constexpr (FMT == FMT_SPLIT) {
            const char* p = reinterpret_cast<const char*>(src) + (pix[n] * cp + (size_t)(c0 >> 4) * 16) * 4 + ((c0 >> 3) & 1) * 16 + (c0 & 7) * 2;
            const f16x4 hi = *reinterpret_cast<const f16x4*>(p), lo = *reinterpret_cast<const f16x4*>(p + 32);
#pragma unroll
            for (int k = 0; k < 4; ++k) dst[gi][n][k] = (float)hi[k] + (float)lo[k] * SPLIT_INV;
          } else {
            const f16x4 h = *reinterpret_cast<const f16x4*>(reinterpret_cast<const char*>(src) + (pix[n] * cp + c0) * 2);
#pragma unroll
            for (int k = 0; k < 4; ++k) dst[gi][n][k] = (float)h[k];
          }
        }
    };
    auto mfma_chunk = [&](int g0, const f32x4 (&xb)[4][2]) __attribute__((always_inline)) {
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        const int g = g0 + gi;
        if (g < n_g) {  // (wave-uniform)
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const f32x4 wa = *reinterpret_cast<const f32x4*>(lds + (m * 32 + lx) * ldw + 8 * g + 4 * lh);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j], xb[gi][n][j], acc[m][n], 0, 0, 0);
          }
        }
      }
    };
    f32x4 xa[4][2], xc[4][2];
    load_chunk(0, xa);
    for (int g0 = 0; g0 < n_g; g0 += 8) {
      if (g0 + 4 < n_g) load_chunk(g0 + 4, xc);
      mfma_chunk(g0, xa);
      if (g0 + 8 < n_g) load_chunk(g0 + 8, xa);
      if (g0 + 4 < n_g) mfma_chunk(g0 + 4, xc);
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      if (!ok[n]) continue;
      const size_t b = pix[n] / HW, hw = pix[n] - b * HW;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (co < cout) {
            float v = acc[m][n][r] + bias[co];
            if (sigmoid) v = 1.f / (1.f + expf(-v));
            dst[(b * cout + co) * HW + hw] = v;
          }
        }
    }
  }
}

template <int FMT>
__global__ void slot_to_nchw_fmt_kernel(const void* __restrict__ src, float* __restrict__ dst, int B, int HW, int cp, int c) {
  const int groups = (c + 7) / 8;
  const size_t total = (size_t)B * HW * groups;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = idx % ((size_t)B * HW);
    const int g = (int)(idx / ((size_t)B * HW));
    float v[8];
    load8<FMT>(src, pix, cp, g, v);
    const size_t b = pix / HW, hw = pix - b * HW;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (8 * g + k < c) dst[(b * c + 8 * g + k) * HW + hw] = v[k];
  }
}

#define PH_FMT_DISPATCH(fmt, CALL)                     \
  switch (fmt) {                                       \
    case FMT_SPLIT: { constexpr int F = FMT_SPLIT; CALL; break; } \
    case FMT_F16: { constexpr int F = FMT_F16; CALL; break; }     \
    default: { constexpr int F = FMT_F32; CALL; break; }          \
  }

int launch_upsample_fmt(int fmt, const void* src, void* dst, int B, int H, int W, int cp, hipStream_t s, int f16math) {
  const size_t total = (size_t)B * (H + 1) * (W + 1) * (cp / 8);
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 32);
  PH_FMT_DISPATCH(fmt, hipLaunchKernelGGL(upsample2x_fmt_kernel<F>, dim3(blocks), dim3(256), 0, s, src, dst, B, H, W, cp, f16math));
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

int launch_pool_fmt(int fmt, const void* src, void* dst, int B, int H, int W, int cp, hipStream_t s) {
  const size_t total = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (cp / 8);
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 32);
  PH_FMT_DISPATCH(fmt, hipLaunchKernelGGL(pool2x2_fmt_kernel<F>, dim3(blocks), dim3(256), 0, s, src, dst, B, H, W, cp));
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

int launch_head_fmt(int fmt, const void* src, const float* w, const float* bias, float* dst, int B, int HW, int cp, int wcp, int cout, int sigmoid, hipStream_t s) {
  PH_REQUIRE(wcp % 8 == 0 && wcp <= cp && cout >= 1 && cout <= 128, "head1x1: unsupported shape (cp=%d, weight row=%d, cout=%d)", cp, wcp, cout);
  const int mt = (cout + 31) / 32;
  const size_t lds = (size_t)mt * 32 * (wcp + 4) * sizeof(float);
  PH_REQUIRE(lds <= 160 * 1024, "head1x1: weights do not fit the LDS (cin=%d cout=%d)", wcp, cout);
  const size_t n_groups = ((size_t)B * HW + 63) / 64;
  const dim3 grid((unsigned)std::min<size_t>((n_groups + 3) / 4, 256 * 8));
#define PH_HEAD_LAUNCH(F, M)                                                                                                              \
  {                                                                                                                                       \
    if (lds > 64 * 1024)                                                                                                                  \
      PH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(head1x1_mfma_kernel<F, M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((head1x1_mfma_kernel<F, M>), grid, dim3(256), lds, s, src, w, bias, dst, B, HW, cp, wcp, cout, sigmoid);            \
  }
#define PH_HEAD_MT(F)                 \
  switch (mt) {                       \
    case 1: PH_HEAD_LAUNCH(F, 1) break; \
    case 2: PH_HEAD_LAUNCH(F, 2) break; \
    case 3: PH_HEAD_LAUNCH(F, 3) break; \
    default: PH_HEAD_LAUNCH(F, 4) break; \
  }
  PH_FMT_DISPATCH(fmt, PH_HEAD_MT(F));
#undef PH_HEAD_MT
#undef PH_HEAD_LAUNCH
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

int launch_slot_to_nchw_fmt(int fmt, const void* src, float* dst, int B, int HW, int cp, int c, hipStream_t s) {
  const size_t total = (size_t)B * HW * ((c + 7) / 8);
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
  PH_FMT_DISPATCH(fmt, hipLaunchKernelGGL(slot_to_nchw_fmt_kernel<F>, dim3(blocks), dim3(256), 0, s, src, dst, B, HW, cp, c));
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// Fused first encoder block in the plain-fp16 precision, both convolutions on the matrix cores (handle option "stem_f16mfma"; stem_fused_kernel<CIN, 3> of
// net_kernels.hip computes the first conv with ~36 vector FMAs per pixel and 4-channel group and converts its fp32 result as the second conv reads it:
// ~620 vector instructions around 36 MFMAs per 64 pixels, 219 us of BASELINE cfg5's 1.39-ms forward).
//   uint8 / float image tile -> / 255 -> fp16 (autocast's cast of the conv input, torch_backend.py:113-143) -> conv3x3(Cin -> 16) as an im2col product on
//   v_mfma_f32_16x16x16_f16 (K = 9 Cin taps padded to 16 / 32: a lane gathers its four taps of its pixel from the LDS image) + bias + ReLU -> fp16 halo tile in
//   LDS (zeros outside the image: the second conv's padding) -> conv3x3(16 -> 16) on v_mfma_f32_16x16x32_f16, two taps per MFMA (K = 2 x 16 channels) + bias +
//   ReLU -> 2x2 max pool in registers -> pooled NHWC store (+ the full-resolution tensor if somebody reads it).
// Reference: encoder_decoder.py:108-121 (SimpleConvBlock), common.py:69-107 (MaxPool2dWithSamePadding), lightning_modules.py:1840-1848 (/ 255).
// Tile 8 rows x 32 columns per 256-thread workgroup as stem_fused_kernel; wave w owns output rows 2w, 2w + 1 (four 16-pixel M tiles).
// ---------------------------------------------------------------------------------------
namespace {
constexpr int S_TH = 8, S_TW = 32, S_HW = S_TW + 2, S_HH = S_TH + 2, S_IW = S_TW + 4, S_IH = S_TH + 4;
}

// Persistent: a workgroup walks tiles with stride gridDim.x; the weight operands, the lane's tap offsets and the halo geometry of its M tiles are set up once; the image
// bytes of the NEXT tile are requested before the current tile's first conv and land in registers behind its matrix work (a tile is ~1 us of instructions behind ~2 us of
// cold-load latency otherwise).  Two barriers per tile: image patch written / halo tile written; the halo tile is double-buffered so that a wave may start the next tile's
// first conv while another still reads this tile's.
__device__ __forceinline__ void stem_barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

#ifndef PH_STEM_OCC
#define PH_STEM_OCC 4  // waves per SIMD the register budget allows = workgroups per CU (cfg5, tools/f16_ab.sh: 3 -> 180 us, 4 -> 161, 5 -> 225: the tile loop spills; with the biases and the
                       // per-M-tile pixel offsets in LDS tables instead of 20 registers the kernel needs 95 registers and 5 -> 155 us, 4 -> 168, 6 -> 194: a 3 % gain at the noise level, not kept)
#endif
template <int CIN>
__global__ __launch_bounds__(256, PH_STEM_OCC) void stem_f16_kernel(StemArgs a) {
  constexpr int NK = (9 * CIN + 15) / 16;      // K steps of 16 of the first conv
  constexpr int NP0 = S_HH * S_HW;             // 340 halo pixels of the first conv's output
  constexpr int NMT = (NP0 + 15) / 16;         // 22 M tiles of the first conv
  constexpr int MT_W = (NMT + 3) / 4;          // per wave
  constexpr int NIMG = CIN * S_IH * S_IW;      // image patch elements
  constexpr int IMG_IT = (NIMG + 255) / 256;
  __shared__ _Float16 sImg[NIMG + 8];
  __shared__ __attribute__((aligned(16))) _Float16 sA[2][(NP0 + 4) * 16];
  __shared__ _Float16 sLut[256];  // uint8 frames: fp16(v / 255), the correctly rounded division done ONCE per value (a v_div sequence per pixel was ~10 % of the tile's vector instructions)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int tiles_x = (a.W + S_TW - 1) / S_TW, tiles_y = (a.H + S_TH - 1) / S_TH, tiles = tiles_x * tiles_y * a.B;
  const int per_xcd = (tiles + 7) >> 3;  // every XCD gets a contiguous range of tiles (neighbouring tiles share image lines)
  const int nvirt = 8 * per_xcd;
  sLut[tid] = (_Float16)((float)tid / 255.0f);  // (published by the first barrier of the tile loop)

  // ---- operands that do not depend on the tile: first conv A[i = li (output channel)][k = 16 s + 4 lg + j], k = tap * CIN + c (zero beyond 9 CIN); the lane's tap
  // offsets into the image patch; second conv A of tap pair pp: lanes lg 0, 1 hold tap 2 pp (input channels 8 (lg & 1) ..), lg 2, 3 tap 2 pp + 1
  f16x4 wa0[NK];
  int koff[NK][4];
#pragma unroll
  for (int s = 0; s < NK; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 16 * s + 4 * lg + j;
      const bool real = k < 9 * CIN;
      const int tap = real ? k / CIN : 0, c = real ? k - tap * CIN : 0;
      wa0[s][j] = real ? (_Float16)a.w0[(size_t)(tap * CIN + c) * 16 + li] : (_Float16)0.f;
      koff[s][j] = c * (S_IH * S_IW) + (tap / 3) * S_IW + (tap % 3);
    }
  f16x8 wb[5];
  int tapoff[5];
#pragma unroll
  for (int pp = 0; pp < 5; ++pp) {
    const int tap = 2 * pp + (lg >> 1);
    const bool real = tap < 9;
    const int tp = real ? tap : 8;
    const float* w = a.w1 + (size_t)(tp * 16 + li) * 16 + 8 * (lg & 1);
    const f32x4 w_lo = *reinterpret_cast<const f32x4*>(w), w_hi = *reinterpret_cast<const f32x4*>(w + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      wb[pp][k] = real ? (_Float16)w_lo[k] : (_Float16)0.f;
      wb[pp][4 + k] = real ? (_Float16)w_hi[k] : (_Float16)0.f;
    }
    tapoff[pp] = ((tp / 3) * S_HW + (tp % 3)) * 16 + 8 * (lg & 1);  // in fp16 elements of the halo tile
  }
  const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.b0 + 4 * lg);
  const f32x4 bias4 = *reinterpret_cast<const f32x4*>(a.b1 + 4 * lg);
  // the lane's pixel in each of its wave's M tiles of the first conv: image-patch offset and halo coordinates (hy << 8 | hx), the same for every tile
  int m_base[MT_W], m_yx[MT_W];
#pragma unroll
  for (int i = 0; i < MT_W; ++i) {
    const int p = (wave + 4 * i) * 16 + li;
    const int pc = p < NP0 ? p : NP0 - 1;
    const int hy = pc / S_HW, hx = pc - hy * S_HW;
    m_base[i] = hy * S_IW + hx;
    m_yx[i] = (hy << 8) | hx;
  }
  // image patch elements of this thread (element e = tid + 256 it): patch coordinates
  int e_yx[IMG_IT];
#pragma unroll
  for (int it = 0; it < IMG_IT; ++it) {
    const int e = min(tid + 256 * it, NIMG - 1);
    const int c = e / (S_IH * S_IW), r = e - c * (S_IH * S_IW);
    const int iy = r / S_IW, ix = r - iy * S_IW;
    e_yx[it] = (c << 16) | (iy << 8) | ix;
  }
  auto tile_of = [&](int vid, int& b, int& y0, int& x0) {
    int t = (vid & 7) * per_xcd + (vid >> 3);
    const bool ok = vid < nvirt && t < tiles;
    t = ok ? t : 0;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    b = t / tiles_y;
    x0 = tx * S_TW;
    y0 = ty * S_TH;
    return ok;
  };
  float img[IMG_IT];  // float frames: the value; uint8 frames: the byte (as an int's bits)
  auto fetch = [&](int b, int y0, int x0) {
#pragma unroll
    for (int it = 0; it < IMG_IT; ++it) {
      const int c = e_yx[it] >> 16, iy = (e_yx[it] >> 8) & 255, ix = e_yx[it] & 255;
      const int gy = y0 + iy - 2, gx = x0 + ix - 2;
      float v = 0.f;  // (the bits of integer 0 too)
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
        const size_t o = (((size_t)b * CIN + c) * a.H + gy) * a.W + gx;
        if (a.dtype == 0)
          v = __builtin_bit_cast(float, (int)reinterpret_cast<const uint8_t*>(a.src)[o]);
        else
          v = reinterpret_cast<const float*>(a.src)[o];
      }
      img[it] = v;
    }
  };
  const int Hp = (a.H + 1) / 2, Wp = (a.W + 1) / 2;
  f16x4 zero4;
#pragma unroll
  for (int r = 0; r < 4; ++r) zero4[r] = (_Float16)0.f;

  int vid = blockIdx.x;
  int b, y0, x0;
  bool ok = tile_of(vid, b, y0, x0);  // (workgroup-uniform)
  if (ok) fetch(b, y0, x0);
  int par = 0;
  while (ok) {
    // ---- image patch -> fp16 in LDS (/ 255 for integer-valued inputs)
    if (a.dtype == 0) {  // (workgroup-uniform)
      if (vid != (int)blockIdx.x) {
#pragma unroll
        for (int it = 0; it < IMG_IT; ++it)
          if (tid + 256 * it < NIMG) sImg[tid + 256 * it] = sLut[__builtin_bit_cast(int, img[it])];
      } else {  // first tile: the table is not published yet
#pragma unroll
        for (int it = 0; it < IMG_IT; ++it)
          if (tid + 256 * it < NIMG) sImg[tid + 256 * it] = (_Float16)((float)__builtin_bit_cast(int, img[it]) / 255.0f);
      }
    } else {
#pragma unroll
      for (int it = 0; it < IMG_IT; ++it)
        if (tid + 256 * it < NIMG) sImg[tid + 256 * it] = (_Float16)(a.dtype == 1 ? img[it] : img[it] / 255.0f);
    }
    stem_barrier_lds();  // (not __syncthreads(): its vmcnt(0) would wait for the next tile's image loads here)
    const int nvid = vid + gridDim.x;
    int nb, ny0, nx0;
    const bool nok = tile_of(nvid, nb, ny0, nx0);
    if (nok) fetch(nb, ny0, nx0);  // lands behind this tile's matrix work
    _Float16* const A = sA[par];
    const bool interior = y0 >= 1 && y0 + S_TH + 1 <= a.H && x0 >= 1 && x0 + S_TW + 1 <= a.W;  // the whole halo lies inside the image (workgroup-uniform)

    // ---- first conv: M tile = 16 consecutive halo pixels (row-major over the 10 x 34 halo), transposed product D[channel 4 lg + r][pixel li]
#pragma unroll
    for (int i = 0; i < MT_W; ++i) {
      const int mt = wave + 4 * i;
      if (mt < NMT) {  // (wave-uniform)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NK; ++s) {
          f16x4 xb;
#pragma unroll
          for (int j = 0; j < 4; ++j) xb[j] = sImg[m_base[i] + koff[s][j]];
          acc = __builtin_amdgcn_mfma_f32_16x16x16f16(wa0[s], xb, acc, 0, 0, 0);
        }
        f32x4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = fmaxf(acc[r] + b0[r], 0.f);
        f16x4 o = __builtin_convertvector(t, f16x4);
        if (!interior) {  // (workgroup-uniform: only tiles at the image border test their halo pixels) outside the image = the second conv's zero padding
          const int gy = y0 + (m_yx[i] >> 8) - 1, gx = x0 + (m_yx[i] & 255) - 1;
          if (!(gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)) o = zero4;
        }
        const int p = mt * 16 + li;
        if (p < NP0) *reinterpret_cast<f16x4*>(A + p * 16 + 4 * lg) = o;
      }
    }
    stem_barrier_lds();  // (not __syncthreads(): its vmcnt(0) would wait for the next tile's image loads here)

    // ---- second conv: two taps per MFMA
    f32x4 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int h = 0; h < 2; ++h) acc[m][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pp = 0; pp < 5; ++pp)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f16x8 xb = *reinterpret_cast<const f16x8*>(A + ((2 * wave + m) * S_HW + h * 16 + li) * 16 + tapoff[pp]);
          acc[m][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[pp], xb, acc[m][h], 0, 0, 0);
        }
    const bool whole = y0 + S_TH <= a.H && x0 + S_TW <= a.W;  // no output pixel beyond the image (workgroup-uniform)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int x = x0 + h * 16 + li;
      float pooled[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) pooled[r] = 0.f;  // values are >= 0 after the ReLU; out-of-image elements count as the reference's zero pad
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int y = y0 + 2 * wave + m;
        const bool in = whole || ((y < a.H) && (x < a.W));
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = fmaxf(acc[m][h][r] + bias4[r], 0.f);
        if (a.dst_full && in) {  // 16 logical channels in the 32-channel fp16 format: the upper 16 are written as zeros (nobody else would)
          const f16x4 oh = __builtin_convertvector(f32x4{o[0], o[1], o[2], o[3]}, f16x4);
          _Float16* d = reinterpret_cast<_Float16*>(a.dst_full) + (((size_t)b * a.H + y) * a.W + x) * 32 + 4 * lg;
          *reinterpret_cast<f16x4*>(d) = oh;
          *reinterpret_cast<f16x4*>(d + 16) = zero4;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) pooled[r] = fmaxf(pooled[r], in ? o[r] : 0.f);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {  // the x neighbour of the pool window sits in lane li ^ 1
        const int pi = __builtin_bit_cast(int, pooled[r]);
        pooled[r] = fmaxf(pooled[r], __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(pi, pi, 0xB1, 0xF, 0xF, false)));
      }
      const f16x4 ph = __builtin_convertvector(f32x4{pooled[0], pooled[1], pooled[2], pooled[3]}, f16x4);
      const int py = (y0 >> 1) + wave, px = (x0 >> 1) + h * 8 + (li >> 1);
      if (!(li & 1) && (whole || (py < Hp && px < Wp))) {
        _Float16* d = reinterpret_cast<_Float16*>(a.dst_pool) + (((size_t)b * Hp + py) * Wp + px) * 32 + 4 * lg;
        *reinterpret_cast<f16x4*>(d) = ph;
        *reinterpret_cast<f16x4*>(d + 16) = zero4;
      }
    }
    par ^= 1;
    vid = nvid;
    ok = nok;
    b = nb, y0 = ny0, x0 = nx0;
  }
}

int launch_stem_f16(const StemArgs& a, hipStream_t s) {
  PH_REQUIRE(a.out_fmt == FMT_F16 && (a.cin == 1 || a.cin == 3), "stem_f16_kernel: plain fp16 outputs, 1 or 3 input channels");
  int n_cu = 0;
  {
    const int rc_cu = device_cu_count(&n_cu);
    if (rc_cu != PH_OK) return rc_cu;
  }
  const int tiles = ((a.W + S_TW - 1) / S_TW) * ((a.H + S_TH - 1) / S_TH) * a.B;
  const int grid = std::min(8 * ((tiles + 7) / 8), PH_STEM_OCC * n_cu);  // persistent: four workgroups per CU -- 128 registers: at six the tile loop spills -- (a multiple of 8: the XCD dealing of the virtual tile ids)
  if (a.cin == 1)
    hipLaunchKernelGGL((stem_f16_kernel<1>), dim3(grid), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((stem_f16_kernel<3>), dim3(grid), dim3(256), 0, s, a);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// Two 3x3 convolutions of a 32-channel encoder block in ONE launch, plain fp16 (handle option "block_fuse"): conv(<= 16 -> 32) + ReLU -> the fp16 halo tile stays in LDS
// -> conv(32 -> 32) + ReLU (+ 2x2 max pool).  BASELINE cfg5's second encoder block (384 x 384, 16 frames) is 75 + 83 us as two launches of conv3x3_f16_persist_kernel --
// 4 x the MFMA time and 2 x the time of their algorithmic bytes: one K chunk per tile leaves nothing to overlap the tile's loads with -- and the 32-channel intermediate
// (151 MB at cfg5, 537 MB at cfg3's 32 frames) is written and read back in between.  Here it never reaches HBM.
// Reference: encoder_decoder.py:108-121 (SimpleConvBlock: two Conv2d + ReLU), common.py:69-107 (MaxPool2dWithSamePadding).
//   Persistent 256-thread workgroups, tile 8 rows x 32 columns (as the fused stem).  Per tile: the 12 x 36-pixel input halo (the 16 real channels: 32 B per pixel) arrives by
//   LDS-DMA one tile ahead (double-buffered); first conv = 22 M tiles of 16 consecutive halo pixels x two 16-channel N tiles x 9 taps on v_mfma_f32_16x16x16_f16 (weights in
//   registers), + bias + ReLU, zeros outside the image (the second conv's padding), fp16 -> LDS [pixel][32 channels] at a 96-byte pitch (conflict-free b128 fragment reads);
//   second conv = wave w's rows 2w, 2w + 1 (four M tiles) x two N tiles x 9 taps on v_mfma_f32_16x16x32_f16, weights from LDS (f16_weight_pack_kernel's pieces ARE the A
//   operand), + bias + ReLU, the 2x2 max pool in registers, pooled and (if anybody reads it) full-resolution stores.
// ---------------------------------------------------------------------------------------
namespace {
constexpr int B2_IN_PX = S_IH * S_IW;        // 432 input halo pixels, 32 B each (channels 0 .. 15 of the 32-channel fp16 pixel)
constexpr int B2_IN_BYTES = ((B2_IN_PX * 32 + 1023) / 1024) * 1024;  // 14 KiB (whole 1-KiB DMA pieces)
constexpr int B2_MID_PX = S_HH * S_HW;       // 340
constexpr int B2_MID_PITCH = 96;             // bytes per intermediate pixel (64 + 32 pad)
constexpr int B2_WB_BYTES = 18 * 1024;
constexpr int B2_LDS = 2 * B2_IN_BYTES + (B2_MID_PX + 2) * B2_MID_PITCH + B2_WB_BYTES;
}  // namespace


__global__ __launch_bounds__(256, 2) void block2_c32_f16_kernel(Block2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sIn = smem;
  char* const sMid = smem + 2 * B2_IN_BYTES;
  char* const sWB = sMid + (B2_MID_PX + 2) * B2_MID_PITCH;
  constexpr int NMT = (B2_MID_PX + 15) / 16;  // 22
  constexpr int MT_W = (NMT + 3) / 4;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int tiles_x = (a.W + S_TW - 1) / S_TW, tiles_y = (a.H + S_TH - 1) / S_TH, tiles = tiles_x * tiles_y * a.B;
  const int per_xcd = (tiles + 7) >> 3;
  const int nvirt = 8 * per_xcd;
  auto tile_of = [&](int vid, int& b, int& y0, int& x0) {
    int t = (vid & 7) * per_xcd + (vid >> 3);
    const bool ok = vid < nvirt && t < tiles;
    t = ok ? t : 0;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    b = t / tiles_y;
    x0 = tx * S_TW;
    y0 = ty * S_TH;
    return ok;
  };
  // input halo by LDS-DMA: piece p (1 KiB) = 32 consecutive halo pixels x 32 B; lane -> (pixel p 32 + lane / 2, 16-byte half lane % 2)
  auto stage_in = [&](int b, int y0, int x0, int buf) {
    for (int p = wave; p * 32 < B2_IN_PX; p += 4) {
      const int pix = p * 32 + (lane >> 1);
      const int iy = pix / S_IW, ix = pix - iy * S_IW;
      const int gy = y0 + iy - 2, gx = x0 + ix - 2;
      const bool in = pix < B2_IN_PX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      const char* g = in ? reinterpret_cast<const char*>(a.src) + (((size_t)b * a.H + gy) * a.W + gx) * 64 + (lane & 1) * 16 : reinterpret_cast<const char*>(a.zeros);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(sIn + buf * B2_IN_BYTES + p * 1024), 16, 0, 0);
    }
  };
  // ---- operands that do not depend on the tile
  for (int i = tid; i < B2_WB_BYTES / 16; i += 256) reinterpret_cast<f32x4*>(sWB)[i] = reinterpret_cast<const f32x4*>(a.wb)[i];
  // first conv on v_mfma_f32_16x16x32_f16 (v_mfma_f32_16x16x16_f16 takes the same 16 cycles for half the K): two taps per MFMA -- lanes lg 0, 1 hold tap 2 pp
  // (input channels 8 (lg & 1) ..), lg 2, 3 tap 2 pp + 1 (zero weights for the missing tenth tap); A[i = li (output channel 16 n + li)][k]
  f16x8 wa[5][2];
#pragma unroll
  for (int pp = 0; pp < 5; ++pp)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int tap = 2 * pp + (lg >> 1);
      const f16x8 w = *reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(a.wa) + ((tap < 9 ? tap : 8) * 2 + n) * 1024 + (lg & 1) * 256 + li * 16);
#pragma unroll
      for (int k = 0; k < 8; ++k) wa[pp][n][k] = tap < 9 ? w[k] : (_Float16)0.f;
    }
  int tapA[5], tapB[9];
#pragma unroll
  for (int pp = 0; pp < 5; ++pp) {
    const int tap = min(2 * pp + (lg >> 1), 8);
    tapA[pp] = ((tap / 3) * S_IW + (tap % 3)) * 32 + (lg & 1) * 16;
  }
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) tapB[tap] = ((tap / 3) * S_HW + (tap % 3)) * B2_MID_PITCH + lg * 16;
  f32x4 ba4[2], bb4[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    ba4[n] = *reinterpret_cast<const f32x4*>(a.ba + 16 * n + 4 * lg);
    bb4[n] = *reinterpret_cast<const f32x4*>(a.bb + 16 * n + 4 * lg);
  }
  int m_base[MT_W], m_yx[MT_W];
#pragma unroll
  for (int i = 0; i < MT_W; ++i) {
    const int p = (wave + 4 * i) * 16 + li;
    const int pc = p < B2_MID_PX ? p : B2_MID_PX - 1;
    const int hy = pc / S_HW, hx = pc - hy * S_HW;
    m_base[i] = (hy * S_IW + hx) * 32;
    m_yx[i] = (hy << 8) | hx;
  }
  const int Hp = (a.H + 1) / 2, Wp = (a.W + 1) / 2;
  const float lo_a = a.relu_a ? 0.f : -__builtin_inff(), lo_b = a.relu_b ? 0.f : -__builtin_inff();
  f16x4 zero4;
#pragma unroll
  for (int r = 0; r < 4; ++r) zero4[r] = (_Float16)0.f;

  int vid = blockIdx.x;
  int b, y0, x0;
  bool ok = tile_of(vid, b, y0, x0);  // (workgroup-uniform)
  if (ok) stage_in(b, y0, x0, 0);
  int par = 0;
  while (ok) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this tile's input (and, first time round, the weights) landed
    __builtin_amdgcn_s_barrier();                                // ... everywhere; every wave is out of the previous tile's intermediate
    asm volatile("" ::: "memory");
    const int nvid = vid + gridDim.x;
    int nb, ny0, nx0;
    const bool nok = tile_of(nvid, nb, ny0, nx0);
    if (nok) stage_in(nb, ny0, nx0, par ^ 1);  // (the other input buffer: its readers passed the barrier above a tile ago)
    const char* const In = sIn + par * B2_IN_BYTES;
    const bool interior = y0 >= 1 && y0 + S_TH + 1 <= a.H && x0 >= 1 && x0 + S_TW + 1 <= a.W;

    // ---- first conv: D[channel 16 n + 4 lg + r][pixel li] of M tile mt = 16 consecutive halo pixels.  The nine fragment reads of M tile i + 1 are issued in front of the
    // MFMAs of M tile i (left to itself the compiler reads a pair of taps, waits, issues four MFMAs: the LDS latency shows eighteen times per tile and wave)
    f16x8 xa[2][5];
    auto read_a = [&](int i, int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int pp = 0; pp < 5; ++pp) xa[buf][pp] = *reinterpret_cast<const f16x8*>(In + m_base[i] + tapA[pp]);
    };
    // (order per M tile: its 18 MFMAs; the wait for the NEXT tile's fragments -- requested a whole tile ago --; the requests of the tile after next into the registers just
    // consumed; this tile's epilogue.  hipcc waits lgkmcnt(0) at a fragment's first use, i.e. also for whatever was requested just before: requests must not sit there)
    read_a(0, 0);
    if (MT_W > 1) read_a(1, 1);
#pragma unroll
    for (int i = 0; i < MT_W; ++i) {
      const int mt = wave + 4 * i;
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int pp = 0; pp < 5; ++pp) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[pp][0], xa[i & 1][pp], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[pp][1], xa[i & 1][pp], acc[1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (i + 1 < MT_W) {
#pragma unroll
        for (int pp = 0; pp < 5; ++pp) asm volatile("" ::"v"(xa[(i + 1) & 1][pp]));
      }
      if (i + 2 < MT_W) read_a(i + 2, i & 1);
      __builtin_amdgcn_sched_barrier(0);
      if (mt < NMT) {  // (wave-uniform)
        const int p = mt * 16 + li;
        f16x4 o[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          f32x4 t;
#pragma unroll
          for (int r = 0; r < 4; ++r) t[r] = fmaxf(acc[n][r] + ba4[n][r], lo_a);  // (lo: 0 with a ReLU, -inf without)
          o[n] = __builtin_convertvector(t, f16x4);
        }
        if (!interior) {  // (workgroup-uniform: only tiles at the image border test their halo pixels) outside the image = the second conv's zero padding
          const int gy = y0 + (m_yx[i] >> 8) - 1, gx = x0 + (m_yx[i] & 255) - 1;
          if (!(gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)) o[0] = o[1] = zero4;
        }
        if (p < B2_MID_PX) {
          *reinterpret_cast<f16x4*>(sMid + p * B2_MID_PITCH + (4 * lg) * 2) = o[0];
          *reinterpret_cast<f16x4*>(sMid + p * B2_MID_PITCH + (16 + 4 * lg) * 2) = o[1];
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- second conv
    f32x4 acc[2][2][2];  // [row m][half h][n]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][h][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (the six fragments of tap t + 1 -- two weight, four pixel -- are requested in front of the eight MFMAs of tap t)
    f16x8 wq[2][2], xq[2][4];
    auto read_b = [&](int tap, int buf) __attribute__((always_inline)) {
      wq[buf][0] = *reinterpret_cast<const f16x8*>(sWB + (tap * 2 + 0) * 1024 + lane * 16);
      wq[buf][1] = *reinterpret_cast<const f16x8*>(sWB + (tap * 2 + 1) * 1024 + lane * 16);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h) xq[buf][2 * m + h] = *reinterpret_cast<const f16x8*>(sMid + ((2 * wave + m) * S_HW + h * 16 + li) * B2_MID_PITCH + tapB[tap]);
    };
    read_b(0, 0);
    read_b(1, 1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          acc[m][h][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[tap & 1][0], xq[tap & 1][2 * m + h], acc[m][h][0], 0, 0, 0);
          acc[m][h][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[tap & 1][1], xq[tap & 1][2 * m + h], acc[m][h][1], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
      if (tap + 1 < 9) {
        asm volatile("" ::"v"(wq[(tap + 1) & 1][0]), "v"(wq[(tap + 1) & 1][1]));
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("" ::"v"(xq[(tap + 1) & 1][k]));
      }
      if (tap + 2 < 9) read_b(tap + 2, tap & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    const bool whole = y0 + S_TH <= a.H && x0 + S_TW <= a.W;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int x = x0 + h * 16 + li;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        float pooled[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pooled[r] = -__builtin_inff();
        bool any_out = false;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int y = y0 + 2 * wave + m;
          const bool in = whole || ((y < a.H) && (x < a.W));
          any_out |= !in;
          float o[4];
          f32x4 t;
#pragma unroll
          for (int r = 0; r < 4; ++r) t[r] = fmaxf(acc[m][h][n][r] + bb4[n][r], lo_b);
          const f16x4 oh = __builtin_convertvector(t, f16x4);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (float)oh[r];
          if (a.dst_full && in) *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(a.dst_full) + (((size_t)b * a.H + y) * a.W + x) * 32 + 16 * n + 4 * lg) = oh;
#pragma unroll
          for (int r = 0; r < 4; ++r) pooled[r] = in ? fmaxf(pooled[r], o[r]) : pooled[r];
        }
        if (a.dst_pool) {  // "same" pooling: a window that reaches past the image includes the zero pad (common.py:93-96)
          const bool xn_in = whole || (x + ((li & 1) ? -1 : 1) < a.W);  // (the x neighbour of the window, lane li ^ 1)
          f16x4 ph;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = any_out ? fmaxf(pooled[r], 0.f) : pooled[r];
            const int pi = __builtin_bit_cast(int, v);
            const float nb_v = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(pi, pi, 0xB1, 0xF, 0xF, false));
            v = fmaxf(v, xn_in ? nb_v : 0.f);
            ph[r] = (_Float16)v;
          }
          const int py = (y0 >> 1) + wave, px = (x0 >> 1) + h * 8 + (li >> 1);
          if (!(li & 1) && (whole || (py < Hp && px < Wp && x < a.W)))
            *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(a.dst_pool) + (((size_t)b * Hp + py) * Wp + px) * 32 + 16 * n + 4 * lg) = ph;
        }
      }
    }
    par ^= 1;
    vid = nvid;
    ok = nok;
    b = nb, y0 = ny0, x0 = nx0;
  }
}

int launch_block2_c32_f16(const Block2Args& a, hipStream_t s) {
  PH_REQUIRE(a.src && a.wa && a.wb && a.ba && a.bb && (a.dst_full || a.dst_pool) && a.zeros && a.B > 0 && a.H > 0 && a.W > 0, "block2_c32_f16_kernel: bad arguments");
  int n_cu = 0;
  {
    const int rc_cu = device_cu_count(&n_cu);
    if (rc_cu != PH_OK) return rc_cu;
  }
  static bool attr_done = false;
  if (!attr_done) {
    PH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(block2_c32_f16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, B2_LDS));
    attr_done = true;
  }
  const int tiles = ((a.W + S_TW - 1) / S_TW) * ((a.H + S_TH - 1) / S_TH) * a.B;
  const int grid = std::min(8 * ((tiles + 7) / 8), 2 * n_cu);
  hipLaunchKernelGGL(block2_c32_f16_kernel, dim3(grid), dim3(256), (size_t)B2_LDS, s, a);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

int prepare_f16_kernels() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_f16_persist_kernel<64, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_f16_persist_kernel<32, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_f16_persist_kernel<64, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_f16_persist_kernel<32, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_f16_persist_kernel<64, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) {
    set_error("hipFuncSetAttribute(conv f16) failed: %s", hipGetErrorString(e));
    return PH_E_HIP;
  }
  return PH_OK;
}

}  // namespace ph
