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
//     512 VGPRs per wave at M 32 x N 64, so the positions are split over the waves of a workgroup: a wave owns row xi of the
//     transformed patch (all four nu: 4 x 2 accumulators of 16 registers, as many as the 1-D kernel holds) for one M half.
//     A workgroup tile is 8 x 8 Winograd tiles = 16 x 16 output pixels x 64 channels; 512 threads, one workgroup per CU, persistent.
//   * Input transform on the fly from the raw 18 x 18 halo tile that LDS-DMA stages: a lane reads the two patch rows its xi
//     combines (four ds_read_b128 each, columns at immediate offsets), one fma per column gives row xi of B^T d, four more give
//     the four A fragments (row xi of B^T d B); each feeds 8 MFMAs -- one VALU instruction per MFMA, the floor of this form.
//   * Weights are transformed on the device (wino2d_pack_kernel) into LDS order [g][xi][nu][n tile][lh][lx][4]: 64 KiB per
//     16-channel chunk, handled in HALVES (8 channels): three weight half-slots (32 KiB) and four halo half-slots (12 KiB) form
//     the software pipeline described at the kernel; every transfer goes through a buffer descriptor (no per-transfer address
//     arithmetic, out-of-image pixels come back as zeros from the range check).
//   * Epilogue: a wave holds, for its xi, P[b] = sum_nu M[xi][nu] A[nu][b]; output row a = 0 is P0 + P1 + P2 and row a = 1 is
//     P1 - P2 - P3 over xi, i.e. over waves: waves xi = 1, 2 pass their P through LDS (the free weight slot, the two free halo
//     slots and a 16-KiB spare), wave xi = 0 finishes the even output rows and wave xi = 3 the odd ones (ReLU, stores; the bias
//     rides in on wave xi = 1's P); the fused 2x2 max pool takes one more hop (wave 0's row maxima to wave 3).  The product is
//     accumulated transposed (weights = A operand): a lane is a tile, a register quad four consecutive channels -- 16-byte stores.
//   Measured (MI355X, cfg3, 32 frames): the 14 N-tile-64 layers take 10.7 ms instead of 13.8 ms with the 1-D kernel; the 768 -> 256
//   layer executes 128 TFLOP/s = 0.81 of the fp32 MFMA peak (288 TFLOP/s in direct-convolution FLOPs).  In-kernel stamps
//   (-DPH_W2_STAMP, tools/w2_stamp.py): a half costs ~4850 cycles for 4096 cycles of MFMA work per SIMD; a tile's epilogue ~9000.
#include <type_traits>

#include "common.h"
#include "net_kernels.h"

namespace ph {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int W2_T = 16;                                   // output tile edge
constexpr int W2_HW = W2_T + 2;                            // halo edge
constexpr int W2_PP_PIECES = (W2_HW * W2_HW / 2 + 63) / 64; // 3 DMA pieces (64 pixels x one 4-channel quad, 1 KiB) per column-parity plane
constexpr int W2_PP_FLOATS = W2_PP_PIECES * 256;           // 768
constexpr int W2_PL_PIECES = 2 * W2_PP_PIECES;             // 6 per quad plane: a half-slot is [quad][column parity][hy * 9 + hx / 2][4] --
constexpr int W2_PL_FLOATS = W2_PL_PIECES * 256;           // 1536   lanes tx = 0..7 of a patch read then hit eight consecutive 16-B bank groups
constexpr int W2_AH_PIECES = 2 * W2_PL_PIECES;             // 12 pieces per half chunk (8 channels)
constexpr int W2_AH_FLOATS = W2_AH_PIECES * 256;           // 3072: one halo half-slot
constexpr int W2_BH_FLOATS = 8192;                         // half a chunk of weights (N tile 64): 32 pieces
constexpr int W2_B_OFF = 4 * W2_AH_FLOATS;                 // 12288: four halo half-slots, then three weight half-slots
constexpr int W2_S_OFF = W2_B_OFF + 3 * W2_BH_FLOATS;      // 36864: 16 KiB spare (epilogue exchange overflow)
constexpr int W2_LDS_FLOATS = W2_S_OFF + 4096;             // 40960 floats = 160 KiB

// rot: added to the N tile (mod nt_count); the kernel passes the round (id / workgroups) when the layer's last N tile is half empty and a round covers
// whole pixel tiles, so that the cheap half tiles go round the workgroups instead of always to the same ones (with an even workgroup count and two N
// tiles a workgroup would otherwise see one N tile only: the ones with the full tiles would set the launch's time)
__device__ __forceinline__ void w2_deal_tile(int id, int tiles, int nt_count, int rot, int* tile, int* ntile) {
  // same dealing as deal_tile (net_kernels.hip): every XCD walks a contiguous range of pixel tiles
  if ((tiles & 7) == 0) {
    const int xcd = id & 7, j = id >> 3;
    *ntile = (j % nt_count + rot) % nt_count;
    *tile = xcd * (tiles >> 3) + j / nt_count;
  } else {
    *ntile = (id % nt_count + rot) % nt_count;
    *tile = id / nt_count;
  }
}

// wpack [panel][tap 9][bn][16] (pack_conv) -> U = G g G^T in LDS order [panel][g 2][xi 4][nu 4][n tile][lh 2][lx 32][4]
__device__ __forceinline__ float wino2d_pack_element(const float* __restrict__ src, int bn, size_t i) {
  const int nt = bn / 32;
  const int per_panel = 32 * nt * 256;
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
  return nu == 0 ? h[0] : (nu == 1 ? 0.5f * ((h[0] + h[2]) + h[1]) : (nu == 2 ? 0.5f * ((h[0] + h[2]) - h[1]) : h[2]));
}
__global__ __launch_bounds__(256) void wino2d_pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int panels, int bn) {
  const size_t total = (size_t)panels * 32 * (bn / 32) * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) dst[i] = wino2d_pack_element(src, bn, i);
}
// every F(2x2,3x3) weight buffer of a model in one launch (ph_model_set_params): block -> segment by binary search, 1024 elements per block
__global__ __launch_bounds__(256) void wino2d_pack_multi_kernel(const PackSegment* __restrict__ seg, int n_seg) {
  int lo = 0, hi = n_seg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (seg[mid].first_block <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackSegment sg = seg[lo];
  const size_t base = (size_t)(blockIdx.x - sg.first_block) * 1024;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const size_t i = base + k * 256 + threadIdx.x;
    if (i < sg.total) sg.dst[i] = wino2d_pack_element(sg.src, sg.bn, i);
  }
}
int launch_wino2d_pack_multi(const PackSegment* seg_dev, int n_seg, unsigned total_blocks, hipStream_t s) {
  if (n_seg == 0 || total_blocks == 0) return PH_OK;
  hipLaunchKernelGGL(wino2d_pack_multi_kernel, dim3(total_blocks), dim3(256), 0, s, seg_dev, n_seg);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}
int64_t wino2d_pack_floats(int panels, int bn) { return (int64_t)panels * 32 * (bn / 32) * 256; }
int launch_wino2d_pack(const float* wpack, float* wino, int panels, int bn, hipStream_t s) {
  const int64_t n = wino2d_pack_floats(panels, bn);
  hipLaunchKernelGGL(wino2d_pack_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, wpack, wino, panels, bn);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

#ifndef PH_W2_STORE_EXP
#define PH_W2_STORE_EXP 0  // diagnostic builds only (tools/w2_store_exp.sh); 0 = the product
#endif
// The K loop runs in HALF chunks (8 input channels).  Half k of the workgroup's running count reads weight half-slot k % 3 and
// the A fragments that were transformed during half k - 1 from halo half-slot k % 4; during half k the waves transform the
// fragments of half k + 1 (slot (k + 1) % 4) and issue the LDS-DMA of the weights of half k + 2 (slot (k + 2) % 3, read
// last in half k - 1) and of the halo of half k + 3 (slot (k + 3) % 4, read last in half k - 2): six DMA instructions per
// wave and half.  A half ends with s_waitcnt vmcnt(6) + s_barrier: every wave has then retired what it issued during the
// PREVIOUS half, so whatever half k + 1 reads is in LDS and visible -- each transfer has between one and two halves
// (1.7 - 3.4 us) to land, and nothing issued recently is ever waited for.  The counters run on across tiles (persistent
// workgroup): the fetch cursors walk into the next tile while the current one is still being multiplied.
// HT: the layer's last N tile is half empty (padded channel count 32 mod 64): that tile skips its missing half's MFMAs and the N tiles rotate over the
// workgroups by round.  A separate instantiation: the second K-loop body and the rotation cost the common case 2 % (code size, scalar registers).
// BWD: the backward's store options (accumulate into dst, the folded ReLU mask) are compiled in; the forward instantiations do not carry their branches
// (~740 cycles per tile of a 64 -> 64 layer, stamp build).
// dst = act(bias + (((p_0 + p_1) + p_2) + ...)) of one (pixel, channel quad) over the K slices' planes, in slice order; the planes' loads are independent (four in flight),
// only the additions are chained
__device__ __forceinline__ f32x4 splitk_sum(const float* __restrict__ p, size_t stride, int ks) {
  f32x4 v = *reinterpret_cast<const f32x4*>(p);
  int k = 1;
  for (; k + 4 <= ks; k += 4) {
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(p + (size_t)k * stride), a1 = *reinterpret_cast<const f32x4*>(p + (size_t)(k + 1) * stride);
    const f32x4 a2 = *reinterpret_cast<const f32x4*>(p + (size_t)(k + 2) * stride), a3 = *reinterpret_cast<const f32x4*>(p + (size_t)(k + 3) * stride);
    v = (((v + a0) + a1) + a2) + a3;
  }
  for (; k < ks; ++k) v += *reinterpret_cast<const f32x4*>(p + (size_t)k * stride);
  return v;
}
// one 2x2 pool window (or, without a pool, just its pixels) x four channels of the second stage: sums, bias, ReLU, full-resolution stores, pooled store ("same" padding: zeros)
__device__ __forceinline__ void splitk_finish_window(const float* __restrict__ part, size_t stride, int ks, const f32x4 bv, float* __restrict__ dst, float* __restrict__ dst_pool,
                                                     int b, int py, int px, int c0, int H, int W, int coutp, int relu) {
  f32x4 pooled = {0.f, 0.f, 0.f, 0.f};
  bool first = true, padded = false;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int oy = 2 * py + dy, ox = 2 * px + dx;
      if (oy >= H || ox >= W) {
        padded = true;
        continue;
      }
      const size_t idx = ((size_t)(b * H + oy) * W + ox) * coutp + c0;
      f32x4 v = splitk_sum(part + idx, stride, ks) + bv;
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (dst) *reinterpret_cast<f32x4*>(dst + idx) = v;
      if (first) {
        pooled = v;
        first = false;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) pooled[e] = fmaxf(pooled[e], v[e]);
      }
    }
  if (dst_pool && !first) {
    if (padded) {
#pragma unroll
      for (int e = 0; e < 4; ++e) pooled[e] = fmaxf(pooled[e], 0.f);
    }
    const int Hp = (H + 1) >> 1, Wp = (W + 1) >> 1;
    *reinterpret_cast<f32x4*>(dst_pool + ((size_t)(b * Hp + py) * Wp + px) * coutp + c0) = pooled;
  }
}

// KS (split K, the small-batch regime): a work unit is (pixel tile, N tile, K slice) -- slice ks of a.ksplit covers halves [ks NH / ksplit, (ks + 1) NH / ksplit) of the NH = 2 nchunks
// halves (both concat sources counted through) -- and its epilogue stores the slice's RAW partial sums (no bias, no ReLU, no pool, no head) to plane ks of a scratch tensor
// [ksplit][B][H][W][coutp]; splitk_reduce_kernel then adds the planes in the fixed order 0, 1, ... and applies bias / ReLU / pool.  A layer whose (pixel tiles x N tiles) fill a
// fraction of the CUs (cfg1: one 16 x 16 tile x 4 N tiles for 256 CUs, 32 - 48 serial halves of ~2 us) runs on ksplit times as many CUs with ksplit times shorter K loops.
template <int BN, bool HT = false, bool BWD = false, bool KS = false>
__global__ __launch_bounds__(512, 2) void conv3x3_wino2d_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NT = BN / 32;
  static_assert(BN == 64, "N tile 64");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave w runs on SIMD w & 3.  The two waves that own the same row xi (M halves 0 and 1) sit on DIFFERENT SIMDs (xi = (w + 2 mh) & 3):
  // the epilogue's finishing work (waves xi = 0 and 3 of both halves) then spreads over all four SIMDs instead of two.
  const int mh = wave >> 2, xi = (wave + 2 * mh) & 3;
  const int tiles_x = (a.W + W2_T - 1) / W2_T;
  const int tiles_y = (a.H + W2_T - 1) / W2_T;
  const int tiles = tiles_x * tiles_y * a.B;
  const int ntc = (a.coutp + BN - 1) / BN;
  const int units = tiles * ntc;
  const int total = KS ? units * a.ksplit : units;
  const int chunks0 = a.c0p / 16;
  const int chunks1 = a.c1p / 16;
  const int nchunks = chunks0 + chunks1;
  const int lx = lane & 31, lh = lane >> 5;

  struct Plan {  // workgroup-uniform
    int b, x0, y0, ntile;
    int ks, h0, nh;  // KS: K slice, its first half and its number of halves
  };
  // (a round = gridDim.x consecutive ids; it covers whole pixel tiles when its id count -- per XCD, where the dealing is XCD-aware -- is a multiple of ntc)
  const int G = (int)gridDim.x;
  const bool rotate = HT && ntc > 1 && (((tiles & 7) == 0) ? ((G & 7) == 0 && (G >> 3) % ntc == 0) : (G % ntc == 0));
  auto setup = [&](int vid, Plan& P) {
    int t, ntile;
    if (KS) {
      const int ks = vid / units, nh_all = 2 * nchunks;
      vid -= ks * units;
      P.ks = ks;
      P.h0 = ks * nh_all / a.ksplit;
      P.nh = (ks + 1) * nh_all / a.ksplit - P.h0;
    }
    w2_deal_tile(vid, tiles, ntc, rotate ? vid / G : 0, &t, &ntile);
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    P.b = t / tiles_y;
    P.x0 = tx * W2_T;
    P.y0 = ty * W2_T;
    P.ntile = ntile;
  };

  Plan P, Pn;
  bool has_next = false;

  // ---- halo fetch cursor (three halves ahead of the multiplication).  Pieces go through a buffer descriptor of the source
  // tensor: the per-lane part of the address is one 32-bit byte offset per piece, fixed for a (tile, source); the half's channel
  // offset rides in the scalar offset, so a half costs no vector instruction; out-of-image pixels (and the unused tail of a
  // plane) get an offset beyond the descriptor's range and the hardware returns zeros ("same" padding).
  unsigned fa_off[2];
  int fa_soff = 0, fa_left = 0, fa_src = 0, fa_rem1 = 0;
  bool fa_more = false;
  Plan FA;
  __amdgpu_buffer_rsrc_t fa_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, (int)((unsigned)(a.B * a.H * a.W) * (unsigned)(a.c0p * 4)), 0x00020000);
  auto a_point = [&](const float* src, int cp) {  // this lane's halo pixel in DMA pieces (wave, min(wave + 8, 11)) of tile FA
    fa_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)((unsigned)(a.B * a.H * a.W) * (unsigned)(cp * 4)), 0x00020000);
    fa_soff = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int p = s == 0 ? wave : min(wave + 8, W2_AH_PIECES - 1);
      const int q2 = p >= W2_PL_PIECES ? 1 : 0;
      const int pp = p - q2 * W2_PL_PIECES;                 // piece within the quad plane: parity plane pp / 3, entries (pp % 3) * 64 + lane
      const int par = pp >= W2_PP_PIECES ? 1 : 0;
      const int e = (pp - par * W2_PP_PIECES) * 64 + lane;  // entry hy * 9 + hx / 2
      const int hy = e / 9, hx = 2 * (e - hy * 9) + par;
      const int gy = FA.y0 + hy - 1, gx = FA.x0 + hx - 1;
      const bool in = cp > 0 && (hy < W2_HW) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      fa_off[s] = in ? ((unsigned)((FA.b * a.H + gy) * a.W + gx) * (unsigned)cp + q2 * 4) * 4u : 0xFFFFFF00u;
    }
  };
  auto a_enter_tile = [&](const Plan& Q) {
    FA = Q;
    if (KS) {  // the slice starts in either source and may run over from the first into the second
      const int n0 = 2 * chunks0, h1 = Q.h0 + Q.nh;
      if (Q.h0 < n0) {
        fa_src = 0;
        fa_left = min(h1, n0) - Q.h0;
        fa_rem1 = max(h1 - n0, 0);
        a_point(a.src0, a.c0p);
        fa_soff = Q.h0 * 32;
      } else {
        fa_src = 1;
        fa_left = Q.nh;
        fa_rem1 = 0;
        a_point(a.src1, a.c1p);
        fa_soff = (Q.h0 - n0) * 32;
      }
      return;
    }
    fa_src = 0;
    fa_left = 2 * chunks0;
    a_point(a.src0, a.c0p);
  };
  auto a_advance = [&]() {  // scalar only
    fa_soff += 32;
    if (--fa_left == 0) {  // workgroup-uniform
      if (fa_src == 0 && (KS ? fa_rem1 > 0 : chunks1 > 0)) {
        fa_src = 1;
        fa_left = KS ? fa_rem1 : 2 * chunks1;
        a_point(a.src1, a.c1p);
      } else if (fa_more) {
        fa_more = false;
        a_enter_tile(Pn);
      } else {  // nothing left to fetch: keep issuing (the vmcnt bookkeeping wants six per half), every lane out of range
        fa_left = 0x40000000;
        a_point(a.src0, 0);
      }
    }
  };
  auto a_issue = [&](int s, float* slot) {
    const int p = s == 0 ? wave : min(wave + 8, W2_AH_PIECES - 1);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(fa_rsrc, (__attribute__((address_space(3))) void*)(slot + p * 256), 16, fa_off[s], fa_soff, 0, 0);
  };
  // ---- weight fetch cursor (two halves ahead): wave w moves pieces 4w .. 4w+3 of a half (one lane offset, immediate offsets)
  const __amdgpu_buffer_rsrc_t fb_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpack_wino2, 0, (int)((unsigned)ntc * (unsigned)nchunks * (unsigned)(2 * W2_BH_FLOATS * 4)), 0x00020000);
  int fb_soff = 0, fb_left = 0;
  bool fb_more = false;
  const unsigned fb_lane = (unsigned)(wave * 1024 + lane * 4) * 4u;
  auto b_enter_tile = [&](const Plan& Q) {
    fb_soff = Q.ntile * nchunks * (2 * W2_BH_FLOATS * 4) + (KS ? Q.h0 * (W2_BH_FLOATS * 4) : 0);
    fb_left = KS ? Q.nh : 2 * nchunks;
  };
  auto b_advance = [&]() {
    fb_soff += W2_BH_FLOATS * 4;
    if (--fb_left == 0) {
      if (fb_more) {
        fb_more = false;
        b_enter_tile(Pn);
      } else {
        fb_soff = 0;
        fb_left = 0x40000000;
      }
    }
  };
  auto b_issue2 = [&](int pair, float* slot) {  // pieces 4w + 2 pair, + 1
    __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)(slot + wave * 1024);
    if (pair == 0) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(fb_rsrc, l, 16, fb_lane, fb_soff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(fb_rsrc, l, 16, fb_lane, fb_soff, 1024, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(fb_rsrc, l, 16, fb_lane, fb_soff, 2048, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(fb_rsrc, l, 16, fb_lane, fb_soff, 3072, 0);
    }
  };

  // A side: lane (lx, lh) is Winograd tile (ty = 4 mh + (lx >> 3), tx = lx & 7); wave xi combines patch rows (ra, rb):
  // xi 0: d0 - d2, xi 1: d1 + d2, xi 2: d2 - d1, xi 3: d1 - d3.  Channel quad lh of the half = plane lh of the slot; the four
  // columns of a patch row are 16 B apart (immediate offsets of the ds_read).
  const int ra = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
  const int rb = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
  float sgn = xi == 1 ? 1.f : -1.f;
  float m1 = -1.f;
  asm volatile("" : "+v"(m1));  // opaque: keeps x - y as fma(y, m1, x), which packs two lanes per instruction (v_pk_fma_f32)
  const int lbase = lh * W2_PL_FLOATS + ((2 * (4 * mh + (lx >> 3))) * 9 + (lx & 7)) * 4;
  const int offRA = lbase + ra * 9 * 4, offRB = lbase + rb * 9 * 4;
  const int offB = xi * 4 * NT * 256 + lh * 128 + lx * 4;
  float* const abuf = lds;
  float* const bbuf = lds + W2_B_OFF;
  float* const spare = lds + W2_S_OFF;

  f32x4 av[4], tt[4], da[2], db[2];  // fragments of the running half; row xi of B^T d of the next one; raw patch rows in flight
  auto aread = [&](const float* aslot, int i, int c) {
    const int co = (c & 1) * W2_PP_FLOATS + (c >> 1) * 4;  // column 2 tx + c: parity plane c & 1, entry tx + (c >> 1)
    da[i] = *reinterpret_cast<const f32x4*>(aslot + offRA + co);
    db[i] = *reinterpret_cast<const f32x4*>(aslot + offRB + co);
  };
  auto ypass = [&](int i, int c) {
    tt[c] = da[i] + sgn * db[i];
    asm volatile("" : "+v"(tt[c]));  // pin: computed here, not sunk to its use in the next half
  };
  auto xpass = [&](int nu) {  // fragment nu of the next half replaces the one whose MFMAs have just been issued
    if (nu == 0)
      av[0] = tt[2] * m1 + tt[0];
    else if (nu == 1)
      av[1] = tt[1] + tt[2];
    else if (nu == 2)
      av[2] = tt[1] * m1 + tt[2];
    else
      av[3] = tt[3] * m1 + tt[1];
    asm volatile("" : "+v"(av[nu]));
  };

#ifdef PH_W2_STAGGER  // diagnostic (tools/w2_store_exp.sh): spread the workgroups' phases over PH_W2_STAGGER per cent of a tile period, so that their store bursts do not coincide chip-wide
  {
    const int period = 2 * nchunks * 4850 + 12000;
    const int delay = (int)(((blockIdx.x >> 3) & 31) * (unsigned)period / 32u) * PH_W2_STAGGER / 100;
    for (int i = 0; i < delay / 6400; ++i) __builtin_amdgcn_s_sleep(100);
  }
#endif
  // ---- prologue: halo halves 0..2 and weight halves 0..1 of the first tile, fragments of half 0
  int vid = blockIdx.x;
  setup(vid, P);
  {
    const int nvid = vid + gridDim.x;
    has_next = nvid < total;
    if (has_next) setup(nvid, Pn);
    fa_more = fb_more = has_next;
  }
  a_enter_tile(P);
  b_enter_tile(P);
#pragma unroll
  for (int h = 0; h < 3; ++h) {
    a_issue(0, abuf + h * W2_AH_FLOATS);
    a_issue(1, abuf + h * W2_AH_FLOATS);
    a_advance();
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    b_issue2(0, bbuf + h * W2_BH_FLOATS);
    b_issue2(1, bbuf + h * W2_BH_FLOATS);
    b_advance();
  }
  // fused head (see the epilogue): its [32 o][64 c] weight matrix lives in the pool half of the spare (a fused head excludes a fused pool),
  // 16-byte chunks XOR-swizzled by the row so that the A-fragment reads (lane = o, row pitch 256 B) are conflict-free; rows >= cout are zeros
  float* const headw = spare + 8 * 256;
  if (a.head_w) {
    const int o = tid >> 4, chunk = tid & 15;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (o < a.head_cout) v = *reinterpret_cast<const f32x4*>(a.head_w + (size_t)o * a.head_wcp + chunk * 4);
    *reinterpret_cast<f32x4*>(headw + (o * 16 + (chunk ^ (o & 15))) * 4) = v;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    aread(abuf, 0, c);
    ypass(0, c);
  }
#pragma unroll
  for (int nu = 0; nu < 4; ++nu) xpass(nu);
#ifdef PH_W2_STAMP
  unsigned long long st_loop = 0, st_e1 = 0, st_e2 = 0, st_e3 = 0, st_tiles = 0;
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_mark = st_t0;
#define W2_STAMP(acc)                                              \
  {                                                                \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();    \
    acc += t_ - st_mark;                                           \
    st_mark = t_;                                                  \
  }
#else
#define W2_STAMP(acc)
#endif
  int ka = 0;  // halo half-slot of the running half (its fragments are already in av)
  int kb = 0;  // weight half-slot of the running half
  while (true) {
    f32x16 acc[4][NT];
    f32x4 bf[2][NT];
    // NV: 32-channel halves of this N tile that hold real output channels (1 for the last tile of a layer whose padded channel count is 32 mod 64 --
    // ConvNeXt's 96-wide decoder level: the other half's MFMAs and weight reads, a quarter of that layer's matrix work, are skipped)
    auto half = [&](auto first_tag, auto nv_tag, bool last) {
      constexpr bool FIRST = decltype(first_tag)::value;
      constexpr int NV = decltype(nv_tag)::value;
      {
        const float* bcur = bbuf + kb * W2_BH_FLOATS + offB;
        const float* bnxt = bbuf + (kb == 2 ? 0 : kb + 1) * W2_BH_FLOATS + offB;
        float* bdst = bbuf + (kb == 0 ? 2 : kb - 1) * W2_BH_FLOATS;   // (kb + 2) % 3
        const float* atr = abuf + ((ka + 1) & 3) * W2_AH_FLOATS;      // halo of the next half
        float* adst = abuf + ((ka + 3) & 3) * W2_AH_FLOATS;
        auto load_b = [&](const float* bslot, int nu, int fbuf) {
#pragma unroll
          for (int n = 0; n < NV; ++n) bf[fbuf][n] = *reinterpret_cast<const f32x4*>(bslot + (nu * NT + n) * 256);
        };
        if (FIRST) load_b(bcur, 0, 0);  // a tile's first half reads its first fragments after the epilogue's barrier; later halves get them in step 3 of the half before
        // Pinned order inside a step: its first MFMAs, THEN the LDS reads of the next step (the compiler's wait before a step's
        // first MFMA is lgkmcnt(0): reads issued before it would be drained on the spot), then the transform arithmetic of patch
        // rows read one step earlier and the DMA issue, then the remaining MFMAs.
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
          const int fcur = nu & 1;
          if (nu == 3) {
            // The half's barrier sits BEFORE its last step, not after it: by now this wave has issued (and, after the lgkmcnt
            // wait, received) every LDS read of the half, so once all waves are here the slots it read may be refilled -- which
            // only happens from the next half's first step on -- and everything any wave issued during the PREVIOUS half has
            // landed (vmcnt(5): all but this half's five transfers so far).  The next half's first fragments are then read under
            // this step's MFMAs instead of after a barrier with the matrix pipe drained.  A tile's first half skips the vmcnt
            // wait: the epilogue before it retired everything, and waiting here would drain the epilogue's 64 global stores
            // (vmcnt counts them in the same queue).
            if (FIRST)
              __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
            else
              __builtin_amdgcn_s_waitcnt(0x0075);  // vmcnt(5) lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int n = 0; n < NV; ++n) {
            if (FIRST) {
              f32x16 z;
#pragma unroll
              for (int r = 0; r < 16; ++r) z[r] = 0.f;
              acc[nu][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[fcur][n][0], av[nu][0], z, 0, 0, 0);
            } else {
              acc[nu][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[fcur][n][0], av[nu][0], acc[nu][n], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (nu + 1 < 4) load_b(bcur, nu + 1, fcur ^ 1);
          if (nu == 3 && !last) load_b(bnxt, 0, 0);  // visible since the barrier above
          if (nu == 0) {
            aread(atr, 0, 0);
            aread(atr, 1, 2);
          } else if (nu == 1) {
            ypass(0, 0);
            ypass(1, 2);
            __builtin_amdgcn_sched_barrier(0);
            aread(atr, 0, 1);
            aread(atr, 1, 3);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int n = 0; n < NV; ++n) acc[nu][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[fcur][n][1], av[nu][1], acc[nu][n], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (nu == 0) {
            b_issue2(0, bdst);
          } else if (nu == 1) {
            xpass(0);
            b_issue2(1, bdst);
          } else if (nu == 2) {
            ypass(0, 1);
            ypass(1, 3);
            xpass(1);
            a_issue(0, adst);
          } else {
            xpass(2);
            a_issue(1, adst);
            a_advance();
            b_advance();
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 2; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < NV; ++n) acc[nu][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[fcur][n][j], av[nu][j], acc[nu][n], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        xpass(3);
        __builtin_amdgcn_sched_barrier(0);
        ka = (ka + 1) & 3;
        kb = kb == 2 ? 0 : kb + 1;
      }
    };
    if (HT && P.ntile * BN + 32 >= a.coutp) {  // workgroup-uniform: only the first 32 channels of this N tile exist
      half(std::true_type{}, std::integral_constant<int, 1>{}, false);
      for (int h = 1; h < 2 * nchunks; ++h) half(std::false_type{}, std::integral_constant<int, 1>{}, h + 1 == 2 * nchunks);
#pragma unroll
      for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nu][1][r] = 0.f;  // (the epilogue's arithmetic runs over both halves; its stores skip channels past coutp)
    } else if (KS) {
      const int nh = P.nh;
      half(std::true_type{}, std::integral_constant<int, NT>{}, nh == 1);
      for (int h = 1; h < nh; ++h) half(std::false_type{}, std::integral_constant<int, NT>{}, h + 1 == nh);
    } else {
      half(std::true_type{}, std::integral_constant<int, NT>{}, false);
      for (int h = 1; h < 2 * nchunks; ++h) half(std::false_type{}, std::integral_constant<int, NT>{}, h + 1 == 2 * nchunks);
    }
    W2_STAMP(st_loop)
    // ---- epilogue.  The product is accumulated TRANSPOSED (weights = A operand): lane (lx, lh) is tile (ty = 4 mh + (lx >> 3), tx = lx & 7)
    // and accumulator register r is channel n * 32 + (r & 3) + 8 (r >> 2) + 4 lh -- four consecutive channels per register quad, so an
    // output pixel leaves as 16-byte stores (16 per finishing wave instead of 64 four-byte ones).
    // Free LDS until the next half's DMA: weight slot (kb + 2) % 3, halo slots (ka + 3) % 4 ("X") and ka ("Y": the next tile's
    // first half, already transformed) and the spare.  Slot X and the weight slot are DMA targets of the NEXT half, slot Y of
    // the one after.
    const int b = P.b, x0 = P.x0, y0 = P.y0, ntile = P.ntile;
    const bool interior = (x0 + W2_T <= a.W) && (y0 + W2_T <= a.H) && ((ntile + 1) * BN <= a.coutp);
    float* const r1 = bbuf + (kb == 0 ? 2 : kb - 1) * W2_BH_FLOATS;
    float* const slot_x = abuf + ((ka + 3) & 3) * W2_AH_FLOATS;
    float* const slot_y = abuf + ka * W2_AH_FLOATS;
    // exchange area of wave xi = 2: 16 groups of 1 KiB per M half -- 12 in a halo slot, 4 in the spare
    float* const r2a = mh ? slot_y : slot_x;
    float* const r2b = spare + mh * 4 * 256;
    auto r2 = [&](int idx) -> float* { return idx < 12 ? r2a + idx * 256 : r2b + (idx - 12) * 256; };
#pragma unroll
    for (int n = 0; n < NT; ++n) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (!KS && xi == 1) bias = *reinterpret_cast<const f32x4*>(a.bias + ntile * BN + n * 32 + 8 * q + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * q + e;
          const float m0 = acc[0][n][r], m1 = acc[1][n][r], m2 = acc[2][n][r], m3 = acc[3][n][r];
          acc[0][n][r] = ((m0 + m1) + m2) + bias[e];
          acc[1][n][r] = ((m1 - m2) - m3) + bias[e];
        }
      }
    }
    if (xi == 1 || xi == 2) {
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
            const int idx = (bb * NT + n) * 4 + k;
            float* dstp = xi == 1 ? r1 + (mh * 16 + idx) * 256 : r2(idx);
            *reinterpret_cast<f32x4*>(dstp + lane * 4) = v;
          }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    W2_STAMP(st_e1)
    if (xi == 0 || xi == 3) {
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int idx = (bb * NT + n) * 4 + k;
            const f32x4 p1 = *reinterpret_cast<const f32x4*>(r1 + (mh * 16 + idx) * 256 + lane * 4);
            const f32x4 p2 = *reinterpret_cast<const f32x4*>(r2(idx) + lane * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float mine = acc[bb][n][4 * k + e];
              acc[bb][n][4 * k + e] = xi == 0 ? mine + (p1[e] + p2[e]) : (p1[e] - p2[e]) - mine;
            }
          }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // The transfers issued during the tile's last half (the next tile's halves 1 and 2) retire HERE, a few thousand cycles after
    // their issue and before the stores enter the queue, so that the next tile's first half needs no vmcnt wait at all.
    __builtin_amdgcn_s_waitcnt(0xF70);  // vmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
    W2_STAMP(st_e2)
    const int arow = xi == 3 ? 1 : 0;  // output row of the tile this wave finishes
    if (xi == 0 || xi == 3) {
      if (a.relu) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bb][n][r] = fmaxf(acc[bb][n][r], 0.f);
      }
    }
    if (a.head_w && (xi == 0 || xi == 3)) {
      // Fused 1x1 head (Cout = this one N tile: the lane pair (lx, lh) holds all 64 channels of its two pixels).  The accumulator layout IS
      // the B operand of v_mfma_f32_32x32x2_f32 -- lane (n = tile lx, k = lh), register r = channel n * 32 + 8 (r >> 2) + 4 lh + (r & 3) --
      // so head[o][tile] = sum_c W[o][c] out[c][tile] is 32 MFMAs per output column with A[i = o = lx][k = lh] = W[o][channel(r, lh)]:
      // four consecutive r are four consecutive channels, one 16-byte LDS read of the head's [o][c] weight rows (staged once per workgroup).
      // The matrix pipe is idle in the epilogue anyway; the 64-channel tensor need not reach HBM (skip_dst) and the head launch is gone.
      // (the head accumulators live in acc[2][0 .. 1]: positions 2 and 3 are dead once P is formed, and the next tile's first MFMAs take C = 0)
      static_assert(NT == 2, "the fused head borrows acc[2][0 .. 1]");
      f32x16(&hd)[NT] = acc[2];
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int r = 0; r < 16; ++r) hd[bb][r] = 0.f;
      int lo = lx;
      asm volatile("" : "+v"(lo));  // (opaque: the eight read addresses are formed here, not hoisted out of the tile loop and spilled)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 wq = *reinterpret_cast<const f32x4*>(headw + (lo * 16 + ((n * 8 + 2 * q + lh) ^ (lo & 15))) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) hd[bb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[e], acc[bb][n][4 * q + e], hd[bb], 0, 0, 0);
        }
      const int oy = y0 + 2 * (4 * mh + (lx >> 3)) + arow, ox = x0 + 2 * (lx & 7);
      if (oy < a.H && ox < a.W) {
        const bool two = ox + 1 < a.W, pair_ok = (a.W & 1) == 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int o = (r & 3) + 8 * (r >> 2) + 4 * lh;  // D row of this register
          asm volatile("" : "+v"(o));                // (opaque: keeps the sixteen output addresses from being hoisted out of the tile loop and spilled)
          if (o < a.head_cout) {
            const float hb = a.head_b[o];
            float v0 = hd[0][r] + hb, v1 = hd[1][r] + hb;
            if (a.head_sigmoid) {
              v0 = 1.f / (1.f + expf(-v0));
              v1 = 1.f / (1.f + expf(-v1));
            }
            float* const hp = a.head_dst + (((size_t)b * a.head_cout + o) * a.H + oy) * a.W + ox;
            if (two && pair_ok) {  // ox is even: with an even W the pixel pair is one aligned 8-byte store
              f32x2 v;
              v[0] = v0;
              v[1] = v1;
              *reinterpret_cast<f32x2*>(hp) = v;
            } else {
              hp[0] = v0;
              if (two) hp[1] = v1;
            }
          }
        }
      }
    }
    if (a.dst_pool) {  // fused 2x2/2 max pool ("same" padding: zeros beyond the image): wave 0's row maxima hop to wave 3
      float* const rp = (mh ? spare + 8 * 256 : slot_y) + lane * 4;  // 8 groups per M half; neither is a DMA target of the next half
      f32x16 rm[NT];
      const int oy = y0 + 2 * (4 * mh + (lx >> 3)) + arow, ox = x0 + 2 * (lx & 7);  // this lane's output pixel pair (ox, ox + 1) in row oy
      if (xi == 0 || xi == 3) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v0 = acc[0][n][r], v1 = acc[1][n][r];
            if (!interior) {
              v0 = (oy < a.H && ox < a.W) ? v0 : 0.f;
              v1 = (oy < a.H && ox + 1 < a.W) ? v1 : 0.f;
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
            *reinterpret_cast<f32x4*>(rp + (n * 4 + k) * 256) = v;
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (xi == 3) {
        const int Hp = (a.H + 1) >> 1, Wp = (a.W + 1) >> 1;
        const int py = oy >> 1, px = ox >> 1;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const f32x4 top = *reinterpret_cast<const f32x4*>(rp + (n * 4 + k) * 256);
            const int co = ntile * BN + n * 32 + 8 * k + 4 * lh;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(top[e], rm[n][4 * k + e]);
            if (interior || (py < Hp && px < Wp && co < a.coutp)) *reinterpret_cast<f32x4*>(a.dst_pool + ((size_t)(b * Hp + py) * Wp + px) * a.coutp + co) = v;
          }
      }
    }
    if ((xi == 0 || xi == 3) && !a.skip_dst) {
      const int oy = y0 + 2 * (4 * mh + (lx >> 3)) + arow, ox = x0 + 2 * (lx & 7);
      if (interior || oy < a.H) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int co = ntile * BN + n * 32 + 8 * k + 4 * lh;
            if (!interior && co >= a.coutp) continue;
            float* dp = a.dst + ((size_t)(b * a.H + oy) * a.W + ox) * a.coutp + co;
            if (KS) dp += (size_t)P.ks * a.split_stride;  // the K slice's plane of the scratch tensor
#if PH_W2_STORE_EXP == 2  // diagnostic: perfectly coalesced (WRONG) addresses, 1 KiB contiguous per store instruction
            dp = a.dst + ((size_t)(b * a.H + y0) * a.W + x0) * a.coutp + (size_t)(wave * 16 + n * 8 + k * 2) * 256 + lane * 4;
#endif
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
#if PH_W2_STORE_EXP == 1  // diagnostic: half the stores
              if (bb) continue;
#elif PH_W2_STORE_EXP == 2
              dp += bb * 256 - bb * a.coutp;
#endif
              if (!interior && ox + bb >= a.W) continue;
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = acc[bb][n][4 * k + e];
              if (BWD && a.accumulate) v += *reinterpret_cast<const f32x4*>(dp + bb * a.coutp);  // gradient accumulation (the second data-gradient launch into a concat source's gradient)
              if (BWD && a.relu_mask_src) {  // backward: this launch completes the gradient of a conv + ReLU output -- that ReLU's mask rides in the (lane-local) store
                const f32x4 f = *reinterpret_cast<const f32x4*>(a.relu_mask_src + (dp - a.dst) + bb * a.coutp);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = f[e] > 0.f ? v[e] : 0.f;
              }
#if PH_W2_STORE_EXP == 3  // diagnostics: cache-policy bits on the store
              asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dp + bb * a.coutp), "v"(v) : "memory");
#elif PH_W2_STORE_EXP == 4
              asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dp + bb * a.coutp), "v"(v) : "memory");
#elif PH_W2_STORE_EXP == 5
              asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dp + bb * a.coutp), "v"(v) : "memory");
#elif PH_W2_STORE_EXP == 6
              asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(dp + bb * a.coutp), "v"(v) : "memory");
#else
              *reinterpret_cast<f32x4*>(dp + bb * a.coutp) = v;
#endif
            }
          }
      }
    }
    if (KS && a.split_counters) {
      // In-kernel second stage (no second launch).  Release / acquire through device-scope counters: every workgroup fences its stores before its increment, whoever reads other
      // workgroups' planes fences again before it does (first touched by this CU in this launch: the L1 is invalidated at kernel start).  The planes are added in slice order
      // whoever adds them: the result does not depend on who does which part.
      //   splitk_finish 1: the workgroup that stores the LAST K slice of a (pixel tile, N tile) does the whole second stage (up to 24 planes of a 64-KiB tile at ONE CU's memory rate:
      //     measured slower than the separate launch).
      //   splitk_finish 2 (round 5): the unit's ksplit workgroups SHARE it.  A workgroup that has stored its slice waits -- bounded: ~25 us -- until all slices have arrived, then
      //     claims share `its slice index` of the tile's items (an atomic bit per share) and reduces it; the last arriver, which never waits, also sweeps every share nobody has
      //     claimed by then (a workgroup that gave up waiting, e.g. because its siblings were not resident yet beside another stream's kernel, leaves without claiming: no
      //     deadlock, every share is done exactly once).  The last workgroup to LEAVE the unit zeroes its three words for the next launch (hipGraph replays included).
      __syncthreads();  // (s_waitcnt vmcnt(0): this workgroup's partial sums have left the CU)
      int* const flags = reinterpret_cast<int*>(spare + 4092);  // [0] all slices arrived, [1] this workgroup is the last arriver, [2] share claimed
      const int unit = ((P.b * tiles_y + P.y0 / W2_T) * tiles_x + P.x0 / W2_T) * ntc + P.ntile;
      const bool shared = a.splitk_finish >= 2 && a.ksplit <= 32 && total <= (int)gridDim.x;
      unsigned* const cnt = a.split_counters + unit;
      unsigned* const claimed = a.split_counters + a.split_counters_n + unit;
      unsigned* const departed = a.split_counters + 2 * a.split_counters_n + unit;
      if (tid == 0) {
        __threadfence();
        const unsigned old = atomicAdd(cnt, 1u);
        const bool last = old + 1u == (unsigned)a.ksplit;
        bool complete = last;
        if (shared && !last) {
          const long long t0 = (long long)__builtin_amdgcn_s_memtime();
          while (true) {
            if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)a.ksplit) {
              complete = true;
              break;
            }
            if ((long long)__builtin_amdgcn_s_memtime() - t0 > 50000ll) break;
            __builtin_amdgcn_s_sleep(8);
          }
        }
        if (!shared && last) *cnt = 0u;
        __threadfence();
        flags[0] = complete ? 1 : 0;
        flags[1] = last ? 1 : 0;
      }
      __syncthreads();
      const bool complete = flags[0] != 0, last = flags[1] != 0;
      const int cq = min(BN, a.coutp - P.ntile * BN) >> 2;  // channel quads of this N tile that exist
      const float* part = a.dst;                             // (the launch's dst IS the scratch tensor)
      // items [i0, i1) of the unit's second stage: 2 x 2 pool windows (or pixels) x channel quads
      auto reduce_items = [&](int i0, int i1) __attribute__((always_inline)) {
        if (a.fin_dst_pool) {
          for (int i = i0 + tid; i < i1; i += 512) {
            const int c4 = i % cq, w = i / cq;
            const int py = (P.y0 >> 1) + (w >> 3), px = (P.x0 >> 1) + (w & 7);
            if (2 * py < a.H && 2 * px < a.W) {
              const int c0 = P.ntile * BN + 4 * c4;
              splitk_finish_window(part, (size_t)a.split_stride, a.ksplit, *reinterpret_cast<const f32x4*>(a.fin_bias + c0), a.fin_dst, a.fin_dst_pool, P.b, py, px, c0, a.H, a.W, a.coutp,
                                   a.fin_relu);
            }
          }
        } else {
          for (int i = i0 + tid; i < i1; i += 512) {
            const int c4 = i % cq, pxl = i / cq;
            const int oy = P.y0 + (pxl >> 4), ox = P.x0 + (pxl & 15);
            if (oy < a.H && ox < a.W) {
              const int c0 = P.ntile * BN + 4 * c4;
              const size_t idx = ((size_t)(P.b * a.H + oy) * a.W + ox) * a.coutp + c0;
              f32x4 v = splitk_sum(part + idx, (size_t)a.split_stride, a.ksplit) + *reinterpret_cast<const f32x4*>(a.fin_bias + c0);
              if (a.fin_relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
              }
              *reinterpret_cast<f32x4*>(a.fin_dst + idx) = v;
            }
          }
        }
      };
      const int n_items = (a.fin_dst_pool ? 64 : 256) * cq;
      if (!shared) {
        if (last) reduce_items(0, n_items);
      } else {
        if (complete) {
          for (int sh = 0; sh < a.ksplit; ++sh) {  // own share first; the last arriver then offers itself for every other one
            const int share = (P.ks + sh) % a.ksplit;
            if (sh > 0 && !last) break;
            __syncthreads();
            if (tid == 0) flags[2] = ((atomicOr(claimed, 1u << share) >> share) & 1u) ? 0 : 1;
            __syncthreads();
            if (flags[2]) reduce_items((int)((long long)share * n_items / a.ksplit), (int)((long long)(share + 1) * n_items / a.ksplit));
          }
        }
        __syncthreads();
        if (tid == 0) {
          __threadfence();
          if (atomicAdd(departed, 1u) + 1u == (unsigned)a.ksplit) {  // the last one out zeroes the unit's words
            *cnt = 0u;
            *claimed = 0u;
            *departed = 0u;
            __threadfence();
          }
        }
      }
      __syncthreads();  // (the flag words are rewritten by the next unit)
    }
    W2_STAMP(st_e3)
#ifdef PH_W2_STAMP
    ++st_tiles;
#endif
    if (!has_next) break;
    vid += gridDim.x;
    P = Pn;
    {
      const int nvid = vid + gridDim.x;
      has_next = nvid < total;
      if (has_next) setup(nvid, Pn);
      fa_more = fb_more = has_next;
    }
  }
#ifdef PH_W2_STAMP
  if (a.clock_probe && lane == 0) {
    unsigned long long* o = a.clock_probe + ((size_t)blockIdx.x * 8 + wave) * 8;
    o[0] = st_loop;
    o[1] = st_e1;
    o[2] = st_e2;
    o[3] = st_e3;
    o[4] = __builtin_amdgcn_s_memtime() - st_t0;
    o[5] = st_tiles;
    o[6] = __builtin_amdgcn_s_memrealtime() - st_r0;
    o[7] = 2 * nchunks;
  }
#endif
}

static int w2_cu_count(int* out) { return device_cu_count(out); }

// Second stage of a split-K launch as a kernel of its own (used when the in-kernel finish below is not available: no counters, or more work units than counters):
// POOL: one thread = one pool window x four channels; otherwise one pixel x four channels.
template <bool POOL>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, size_t stride, int ks, const float* __restrict__ bias, float* __restrict__ dst,
                                                            float* __restrict__ dst_pool, int B, int H, int W, int coutp, int relu) {
  const int cq = coutp >> 2;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int c4 = (int)(i % cq);
  size_t r = i / cq;
  const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c4 * 4);
  if (!POOL) {
    if (r >= (size_t)B * H * W) return;
    const size_t idx = r * coutp + c4 * 4;
    f32x4 v = splitk_sum(part + idx, stride, ks) + bv;
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    *reinterpret_cast<f32x4*>(dst + idx) = v;
    return;
  }
  const int Hp = (H + 1) >> 1, Wp = (W + 1) >> 1;
  if (r >= (size_t)B * Hp * Wp) return;
  const int px = (int)(r % Wp);
  r /= Wp;
  splitk_finish_window(part, stride, ks, bv, dst, dst_pool, (int)(r / Hp), (int)(r % Hp), px, c4 * 4, H, W, coutp, relu);
}

// the second stage of a split-K launch (either Winograd kernel): dst (unless nullptr) and dst_pool (unless nullptr: then dst may be nullptr) from the ks planes of `part`
int launch_splitk_reduce(const float* part, long long stride, int ks, const float* bias, float* dst, float* dst_pool, int B, int H, int W, int coutp, int relu, hipStream_t s) {
  if (dst_pool) {
    const size_t threads = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (coutp / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, part, (size_t)stride, ks, bias, dst, dst_pool, B, H, W, coutp, relu);
  } else {
    const size_t threads = (size_t)B * H * W * (coutp / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, part, (size_t)stride, ks, bias, dst, nullptr, B, H, W, coutp, relu);
  }
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// K slices a layer of this shape takes on the F(2x2,3x3) kernel (1 = no split).  splitk: the handle option "conv_splitk" -- 0 never, 1 where the layer's work units fill less than half
// of the CUs and a slice keeps at least three halves (a unit costs ~8 us of prologue + epilogue + a ~4-us second stage against ~2 us per half), n >= 2 forces n slices (tests).
int wino2d_ksplit_shape(int B, int H, int W, int cinp, int coutp, int splitk, int n_cu) {
  if (splitk <= 0 || cinp < 32) return 1;
  if ((coutp & 63) != 0 && (coutp & 63) <= 32) return 1;  // the half-empty-N-tile instantiation has no split form
  const long units = (long)((W + W2_T - 1) / W2_T) * ((H + W2_T - 1) / W2_T) * B * ((coutp + 63) / 64);
  const int nh = cinp / 8;
  if (splitk >= 2) return std::max(1, std::min(splitk, nh));
  if (units * 2 > n_cu || nh < 8) return 1;
  return (int)std::max<long>(1, std::min<long>(n_cu / units, nh / 3));
}
int64_t wino2d_split_scratch_bytes(int B, int H, int W, int cinp, int coutp, int splitk, int n_cu) {
  const int ks = wino2d_ksplit_shape(B, H, W, cinp, coutp, splitk, n_cu);
  return ks > 1 ? (int64_t)ks * B * H * W * coutp * 4 : 0;
}
int wino2d_ksplit(const ConvArgs& a) {
  if (!a.split_scratch || a.accumulate || a.relu_mask_src || a.head_w) return 1;
  int n_cu = 0;
  if (device_cu_count(&n_cu) != PH_OK || n_cu <= 0) return 1;
  const int ks = wino2d_ksplit_shape(a.B, a.H, a.W, a.c0p + a.c1p, a.coutp, a.splitk, n_cu);
  return (ks > 1 && (int64_t)ks * a.B * a.H * a.W * a.coutp * 4 <= a.split_scratch_bytes) ? ks : 1;
}

int prepare_wino2d_kernels() {
  const void* ks[5] = {reinterpret_cast<const void*>(conv3x3_wino2d_kernel<64, false, false>), reinterpret_cast<const void*>(conv3x3_wino2d_kernel<64, true, false>),
                       reinterpret_cast<const void*>(conv3x3_wino2d_kernel<64, false, true>), reinterpret_cast<const void*>(conv3x3_wino2d_kernel<64, true, true>),
                       reinterpret_cast<const void*>(conv3x3_wino2d_kernel<64, false, false, true>)};
  hipError_t e = hipSuccess;
  for (const void* k : ks)
    if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) {
    set_error("hipFuncSetAttribute(wino2d) failed: %s", hipGetErrorString(e));
    return PH_E_HIP;
  }
  return PH_OK;
}

// The halo DMA addresses a source through a 32-bit buffer descriptor: each source tensor must stay below 4 GiB.
bool wino2d_fits(const ConvArgs& a) {
  const uint64_t px = (uint64_t)a.B * a.H * a.W;
  return px * (uint64_t)a.c0p * 4 < 0xFFFFFF00ull && px * (uint64_t)a.c1p * 4 < 0xFFFFFF00ull && px < 0x7FFFFFFFull;
}
// 3x3 conv, N tile 64: the caller (launch_conv3x3_dma) checks the preconditions.
int launch_conv3x3_wino2d(const ConvArgs& a, hipStream_t s) {
  PH_REQUIRE(a.bn == 64 && a.wpack_wino2 && a.c0p + a.c1p >= 32 && wino2d_fits(a), "wino2d: N tile 64, transformed weights, at least 32 input channels, sources below 4 GiB");
  int n_cu = 0;
  const int rc = w2_cu_count(&n_cu);
  if (rc != PH_OK) return rc;
  const int tiles = ((a.W + W2_T - 1) / W2_T) * ((a.H + W2_T - 1) / W2_T) * a.B;
  const int ntc = (a.coutp + 63) / 64;
  const bool ht = (a.coutp & 63) != 0 && (a.coutp & 63) <= 32, bwd = a.accumulate || a.relu_mask_src;
  const size_t lds = (size_t)W2_LDS_FLOATS * sizeof(float);
  const int ksplit = wino2d_ksplit(a);
  if (ksplit > 1) {  // split K: raw partial sums per slice, then the fixed-order second stage (bias, ReLU, pool)
    ConvArgs k = a;
    k.dst = a.split_scratch;
    k.relu = 0;
    k.dst_pool = nullptr;
    k.skip_dst = 0;
    k.ksplit = ksplit;
    k.split_stride = (long long)a.B * a.H * a.W * a.coutp;
    const bool in_kernel = a.split_counters && tiles * ntc <= a.split_counters_n && a.splitk_finish;  // (split_counters holds 3 x split_counters_n words: arrivals, claimed shares, departures)
    if (in_kernel) {  // the last-arriving workgroup of a (pixel tile, N tile) finishes it: one launch
      k.fin_dst = a.skip_dst ? nullptr : a.dst;
      k.fin_dst_pool = a.dst_pool;
      k.fin_bias = a.bias;
      k.fin_relu = a.relu;
    } else {
      k.split_counters = nullptr;
    }
    hipLaunchKernelGGL((conv3x3_wino2d_kernel<64, false, false, true>), dim3(std::min(tiles * ntc * ksplit, n_cu)), dim3(512), lds, s, k);
    PH_HIP_CHECK(hipGetLastError());
    if (in_kernel) return PH_OK;
    const int rc2 = launch_splitk_reduce(a.split_scratch, k.split_stride, ksplit, a.bias, a.skip_dst ? nullptr : a.dst, a.dst_pool, a.B, a.H, a.W, a.coutp, a.relu, s);
    if (rc2 != PH_OK) return rc2;
    return PH_OK;
  }
  const dim3 grid(std::min(tiles * ntc, n_cu));
  if (ht && bwd)
    hipLaunchKernelGGL((conv3x3_wino2d_kernel<64, true, true>), grid, dim3(512), lds, s, a);
  else if (ht)
    hipLaunchKernelGGL((conv3x3_wino2d_kernel<64, true, false>), grid, dim3(512), lds, s, a);
  else if (bwd)
    hipLaunchKernelGGL((conv3x3_wino2d_kernel<64, false, true>), grid, dim3(512), lds, s, a);
  else
    hipLaunchKernelGGL((conv3x3_wino2d_kernel<64, false, false>), grid, dim3(512), lds, s, a);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

}  // namespace ph
