// 3x3 "same" convolution for SMALL feature maps (the stride >= 4 levels of a UNet at batch 1 - 8: BASELINE cfg1 is ONE 256 x 256 frame) as Winograd
// F(2x2,3x3) on v_mfma_f32_16x16x4_f32, one launch per layer, no split K.
//
// Reference semantics: SimpleConvBlock's Conv2d(k3, "same") + bias + ReLU and the 2x2 max pool behind it (architectures/encoder_decoder.py:108-121,
// architectures/common.py:69-107); the decoder's bilinear x2 (encoder_decoder.py:339-420, F.interpolate(scale_factor=2, mode="bilinear",
// align_corners=False)) and concat((skip, x)) (encoder_decoder.py:545,556) ride in the loader.
//
// Why another kernel: conv3x3_wino2d_kernel / conv3x3_wino4_kernel are persistent whole-CU workgroups on 16 x 16 / 32 x 16-pixel x 64-channel tiles.  A 16 x 16
// map with 256 output channels is FOUR such units; they split K over workgroups to fill the chip and pay a second launch for the reduction (and a third for the
// bilinear): 17.5 + 5.4 (+ 5.7) us per layer whatever its size (profiles/r5_cfg1_trace_before.txt), 5.2 us of each launch being the dispatch floor.  Here a work
// unit is (8 x 8 output pixels = 16 Winograd tiles, 16 output channels): the 16 x 16 map with 256 channels is 64 workgroups, a 32 x 32 one with 128 is 128, and
// nobody shares an accumulator, so there is no reduction, the result is bitwise repeatable, and the bilinear / pool / bias / ReLU are part of the one launch.
//   * Workgroup = 4 MFMA waves + 4 loader waves (one of each per SIMD).  MFMA wave w owns Winograd row xi = w: positions (w, nu), nu = 0..3, i.e. 4 accumulators of
//     4 registers; D[m = output channel][n = tile]: A[m = lane & 15][k = lane >> 4] is a transformed weight, B[k = lane >> 4][n = lane & 15] a transformed input
//     value of tile n.
//   * K runs in chunks of 32 input channels of the concatenated (src0 | src1) channel axis.  The loader waves bring the raw 10 x 10-pixel halo of a chunk with
//     16-byte loads (out-of-image pixels: zeros; a half-resolution src1: four loads and the bilinear weights of ATen's upsample_bilinear2d) into LDS as
//     [row][12][34] floats (the pads make the fragment reads below conflict-free), double-buffered; their loads are requested two chunks before the store, on
//     their own vmcnt, so the memory latency (every chunk's bytes are first touched by all workgroups at once: ~2 us against ~0.6 us of MFMAs) runs beside the
//     MFMA waves instead of in front of them.  One workgroup barrier per chunk.
//   * Per "pair" of K steps (8 channels: lane group g holds channels 8j + 2g, 8j + 2g + 1) a lane reads the two patch rows its Winograd row combines
//     (8 ds_read_b64), 8 packed adds give B operands of the 4 positions x 2 K steps = 8 MFMAs.  Weights come straight from L2 into registers (32 bytes per lane
//     and pair, laid out [N block][chunk][pair][wave][half][lane][4] by sm_pack_kernel), two chunks ahead in three register sets.
//   * Epilogue: the 4 x 16 accumulator registers of the four waves meet in LDS ([position][tile][16 channels], 16 KiB over the halo buffers); thread (tile,
//     channel) applies A^T M A, bias, ReLU, stores its 2 x 2 pixels (64-byte runs of channels) and their max (the tile IS a pool window).
#include "common.h"
#include "net_kernels.h"

namespace ph {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int SM_RS = 12;                         // halo row stride in pixels (10 + 2 pad)
constexpr int SM_CS = 34;                         // floats per halo pixel (32 channels + 2 pad): 2 * 34 = 4 (mod 64), 2 * 12 * 34 = 48 (mod 64) -> the 16 tiles x 2 lane groups of a half-wave hit 64 different banks
constexpr int SM_BUF = 10 * SM_RS * SM_CS;        // 4080 floats per halo buffer
constexpr int SM_PAIR_FLOATS = 4 * 2 * 64 * 4;    // [wave][half][lane][4]
constexpr int SM_CHUNK_FLOATS = 4 * SM_PAIR_FLOATS;  // 8192 floats (32 KiB) per (N block, chunk)

// wpack [n tile][chunk16][tap 9][bn][16] (pack_conv) -> U = G g G^T as [N block][chunk32][pair j][wave = xi][half h][lane][4]; element q = 4 h + r of a lane is
// (nu = q >> 1, e = q & 1): position (xi, nu), output channel 16 nb + (lane & 15), input channel 32 c + 8 j + 2 (lane >> 4) + e (zero beyond the layer's channels)
__global__ __launch_bounds__(256) void sm_pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int nblocks, int nch, int chunks16, int bn) {
  const long total = (long)nblocks * nch * SM_CHUNK_FLOATS;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = (int)(i & 3), l = (int)(i >> 2) & 63, h = (int)(i >> 8) & 1, xi = (int)(i >> 9) & 3, j = (int)(i >> 11) & 3;
    const long rest = i >> 13;
    const int c = (int)(rest % nch), nb = (int)(rest / nch);
    const int q = 4 * h + r, nu = q >> 1, e = q & 1;
    const int co = 16 * nb + (l & 15), k = 32 * c + 8 * j + 2 * (l >> 4) + e;
    float v = 0.f;
    if (k < 16 * chunks16) {
      const int nt = co / bn, row = co - nt * bn;
      const float* w = src + ((((size_t)nt * chunks16 + (k >> 4)) * 9) * bn + row) * 16 + (k & 15);
      const int ts = bn * 16;  // tap stride
      float hh[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float g0 = w[(0 * 3 + kx) * ts], g1 = w[(1 * 3 + kx) * ts], g2 = w[(2 * 3 + kx) * ts];
        hh[kx] = xi == 0 ? g0 : (xi == 1 ? 0.5f * ((g0 + g2) + g1) : (xi == 2 ? 0.5f * ((g0 + g2) - g1) : g2));
      }
      v = nu == 0 ? hh[0] : (nu == 1 ? 0.5f * ((hh[0] + hh[2]) + hh[1]) : (nu == 2 ? 0.5f * ((hh[0] + hh[2]) - hh[1]) : hh[2]));
    }
    dst[i] = v;
  }
}
int64_t sm_pack_floats(int nblocks, int chunks16) { return (int64_t)nblocks * ((chunks16 + 1) / 2) * SM_CHUNK_FLOATS; }
int launch_sm_pack(const float* wpack, float* dst, int nblocks, int chunks16, int bn, hipStream_t s) {
  PH_REQUIRE(nblocks >= 1 && chunks16 >= 1 && (bn == 32 || bn == 64), "small-map pack: N tile 32 or 64");
  const int nch = (chunks16 + 1) / 2;
  hipLaunchKernelGGL(sm_pack_kernel, dim3(std::min(nblocks * nch * (SM_CHUNK_FLOATS / 256), 2048)), dim3(256), 0, s, wpack, dst, nblocks, nch, chunks16, bn);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// NB: N blocks of 16 output channels per workgroup.  2 (Cout a multiple of 32, more units than CUs: the 128 x 128 levels of cfg1) halves the units and shares the halo and its
// transform between the blocks; the weights of a chunk are then 64 registers, kept one chunk ahead in two sets instead of two ahead in three.
template <bool LOWRES, int NB>
__global__ __launch_bounds__(512) void conv3x3_sm_kernel(ConvArgs a) {
  constexpr int EXCH = 4096 * NB > 2 * SM_BUF ? 4096 * NB : 2 * SM_BUF;
  __shared__ __attribute__((aligned(16))) float lds[EXCH + 64 * (16 * NB + 1)];  // two halo buffers (the exchange tile of the epilogue, 4096 NB floats, lies over them) + the output tile a fused head reads
  constexpr int NSETS = NB == 2 ? 2 : 3;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblocks = (a.coutp >> 4) / NB;
  const int tiles_x = (a.W + 7) >> 3, tiles_y = (a.H + 7) >> 3;
  // consecutive workgroup ids are dealt over the 8 XCDs: with the N block as the fastest index an XCD's L2 holds the weights of one or two N blocks
  int u = blockIdx.x;
  const int nb = u % nblocks;
  u /= nblocks;
  const int tx = u % tiles_x;
  u /= tiles_x;
  const int ty = u % tiles_y;
  const int b = u / tiles_y;
  const int y0 = ty * 8, x0 = tx * 8;
  const int Kp = a.c0p + a.c1p, nch = (Kp + 31) >> 5;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;

  if (wave >= 4) {
    // ================= loader waves: global -> registers (two chunks in flight) -> LDS halo buffer, one chunk ahead of the MFMA waves =================
    // piece p = lt + 256 s of a chunk is (halo pixel p >> 3, channel quad q = lt & 7: the same for the thread's four pieces)
    const int lt = tid - 256;
    const int q = lt & 7;
    int pix_off[4];  // pixel index in the full-resolution sources, -1 = outside the image (zeros), -2 = no such piece
    int lds_off[4];
    int gyx[4];      // LOWRES: (gy << 16) | gx of an in-image piece
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int p = lt + 256 * s;
      const int pix = p >> 3;
      const int hy = pix / 10, hx = pix - hy * 10;
      const int gy = y0 + hy - 1, gx = x0 + hx - 1;
      const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      pix_off[s] = p >= 800 ? -2 : (in ? (b * a.H + gy) * a.W + gx : -1);
      lds_off[s] = (hy * SM_RS + hx) * SM_CS + 4 * q;
      gyx[s] = in ? (gy << 16) | gx : 0;
    }
    constexpr int NT = LOWRES ? 4 : 1;  // registers per piece: the four bilinear taps of a half-resolution source
    f32x4 st[2][4][NT];
    bool st_low[2] = {false, false};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // ATen upsample_bilinear2d(align_corners=False), scale 2: source coordinate max((dst + 0.5) / 2 - 0.5, 0), second tap clamped to the last row / column
    auto tap = [&](int gcoord, int nlow, int& i0, int& i1, float& l) {
      const float sc = fmaxf(((float)gcoord + 0.5f) * 0.5f - 0.5f, 0.f);
      i0 = (int)sc;
      i1 = min(i0 + 1, nlow - 1);
      l = sc - (float)i0;
    };
    // every piece loads from a valid address (an out-of-image / missing piece from pixel 0 of the frame) and is zeroed when it goes to LDS: no branches around the loads
    auto fetch = [&](int c, auto S) {
      constexpr int r = decltype(S)::value;
      const int k = 32 * c + 4 * q;
      st_low[r] = false;
      if (k < a.c0p) {
#pragma unroll
        for (int s = 0; s < 4; ++s) st[r][s][0] = *reinterpret_cast<const f32x4*>(a.src0 + (size_t)max(pix_off[s], b * a.H * a.W) * a.c0p + k);
      } else if (k < Kp) {
        const int k1 = k - a.c0p;
        if (LOWRES) {
          st_low[r] = true;
          const int Hl = a.H >> 1, Wl = a.W >> 1;
          const float* const p = a.src1 + (size_t)b * Hl * Wl * a.c1p + k1;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            int iy0, iy1, ix0, ix1;
            float ly, lx;
            tap(gyx[s] >> 16, Hl, iy0, iy1, ly);
            tap(gyx[s] & 0xFFFF, Wl, ix0, ix1, lx);
            st[r][s][0] = *reinterpret_cast<const f32x4*>(p + (size_t)(iy0 * Wl + ix0) * a.c1p);
            st[r][s][NT > 1 ? 1 : 0] = *reinterpret_cast<const f32x4*>(p + (size_t)(iy0 * Wl + ix1) * a.c1p);
            st[r][s][NT > 1 ? 2 : 0] = *reinterpret_cast<const f32x4*>(p + (size_t)(iy1 * Wl + ix0) * a.c1p);
            st[r][s][NT > 1 ? 3 : 0] = *reinterpret_cast<const f32x4*>(p + (size_t)(iy1 * Wl + ix1) * a.c1p);
          }
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) st[r][s][0] = *reinterpret_cast<const f32x4*>(a.src1 + (size_t)max(pix_off[s], b * a.H * a.W) * a.c1p + k1);
        }
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) st[r][s][0] = zero4;
      }
    };
    auto store = [&](int buf, auto S) {
      constexpr int r = decltype(S)::value;
      float* const base = lds + buf * SM_BUF;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        f32x4 v = st[r][s][0];
        if (LOWRES && st_low[r]) {  // the arithmetic of upsample2x_kernel
          int i0, i1;
          float ly, lx;
          tap(gyx[s] >> 16, a.H >> 1, i0, i1, ly);
          tap(gyx[s] & 0xFFFF, a.W >> 1, i0, i1, lx);
          const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = hy * (hx * st[r][s][0][e] + lx * st[r][s][NT > 1 ? 1 : 0][e]) + ly * (hx * st[r][s][NT > 1 ? 2 : 0][e] + lx * st[r][s][NT > 1 ? 3 : 0][e]);
        }
        if (pix_off[s] < 0) v = zero4;
        if (pix_off[s] != -2) {
          *reinterpret_cast<f32x2*>(base + lds_off[s]) = f32x2{v[0], v[1]};
          *reinterpret_cast<f32x2*>(base + lds_off[s] + 2) = f32x2{v[2], v[3]};
        }
      }
    };
    // chunk x travels in register set x & 1: requested two iterations before the one that stores it
    fetch(0, I0{});
    if (nch > 1) fetch(1, I1{});
    store(0, I0{});
    if (nch > 2) fetch(2, I0{});
    __syncthreads();
    auto step = [&](int c, auto Snext) {  // during the MFMAs of chunk c: chunk c + 1 (set Snext) goes to LDS, chunk c + 3 is requested into the freed set
      if (c + 1 < nch) store((c + 1) & 1, Snext);
      if (c + 3 < nch) fetch(c + 3, Snext);
      __syncthreads();
    };
    for (int c = 0; c < nch; c += 2) {
      step(c, I1{});
      if (c + 1 < nch) step(c + 1, I0{});
    }
  } else {
  // ================= MFMA waves =================
  const int n = lane & 15, g = lane >> 4;
  // ---- weights: NSETS register sets, NSETS - 1 chunks ahead
  f32x4 wr[NSETS][NB][4][2];
  const float* const wbase = a.wpack_sm + (size_t)nb * NB * nch * SM_CHUNK_FLOATS + wave * 512 + lane * 4;
  auto fetch_w = [&](int c, auto S) {
    constexpr int s = decltype(S)::value;
#pragma unroll
    for (int ib = 0; ib < NB; ++ib) {
      const float* p = wbase + ((size_t)ib * nch + c) * SM_CHUNK_FLOATS;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        wr[s][ib][j][0] = *reinterpret_cast<const f32x4*>(p + j * SM_PAIR_FLOATS);
        wr[s][ib][j][1] = *reinterpret_cast<const f32x4*>(p + j * SM_PAIR_FLOATS + 256);
      }
    }
  };

  // ---- wave = Winograd row xi combines patch rows (ra, rb): t = d[ra] + sg d[rb]
  const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
  const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
  const float sg = wave == 1 ? 1.f : -1.f;
  const int tty = n >> 2, ttx = n & 3;
  const int offa = ((2 * tty + ra) * SM_RS + 2 * ttx) * SM_CS + 2 * g;
  const int offb = ((2 * tty + rb) * SM_RS + 2 * ttx) * SM_CS + 2 * g;
  f32x4 acc[NB][4];
#pragma unroll
  for (int ib = 0; ib < NB; ++ib)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[ib][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](int buf, auto S) {
    constexpr int s = decltype(S)::value;
    const float* const base = lds + buf * SM_BUF;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x2 t[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x2 da = *reinterpret_cast<const f32x2*>(base + offa + c * SM_CS + 8 * j);
        const f32x2 db = *reinterpret_cast<const f32x2*>(base + offb + c * SM_CS + 8 * j);
        t[c] = da + sg * db;
      }
      const f32x2 v0 = t[0] - t[2], v1 = t[1] + t[2], v2 = t[2] - t[1], v3 = t[1] - t[3];
#pragma unroll
      for (int ib = 0; ib < NB; ++ib) {
        const f32x4 w0 = wr[s][ib][j][0], w1 = wr[s][ib][j][1];
        acc[ib][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[0], v0[0], acc[ib][0], 0, 0, 0);
        acc[ib][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[2], v1[0], acc[ib][1], 0, 0, 0);
        acc[ib][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[0], v2[0], acc[ib][2], 0, 0, 0);
        acc[ib][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[2], v3[0], acc[ib][3], 0, 0, 0);
        acc[ib][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[1], v0[1], acc[ib][0], 0, 0, 0);
        acc[ib][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[3], v1[1], acc[ib][1], 0, 0, 0);
        acc[ib][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[1], v2[1], acc[ib][2], 0, 0, 0);
        acc[ib][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[3], v3[1], acc[ib][3], 0, 0, 0);
      }
    }
  };

  fetch_w(0, I0{});
  if (NSETS == 3 && nch > 1) fetch_w(1, I1{});
  __syncthreads();
  auto body = [&](int c, auto S, auto S2) {  // chunk c from register set S; the weights of chunk c + NSETS - 1 go to set S2 = (S + NSETS - 1) % NSETS
    if (c + NSETS - 1 < nch) fetch_w(c + NSETS - 1, S2);
    compute(c & 1, S);
    __syncthreads();
  };
  if constexpr (NSETS == 3) {
    for (int c = 0; c < nch; c += 3) {
      body(c, I0{}, I2{});
      if (c + 1 < nch) body(c + 1, I1{}, I0{});
      if (c + 2 < nch) body(c + 2, I2{}, I1{});
    }
  } else {
    for (int c = 0; c < nch; c += 2) {
      body(c, I0{}, I1{});
      if (c + 1 < nch) body(c + 1, I1{}, I0{});
    }
  }

  // ---- exchange [position][tile][16 NB channels] (the barrier that closed the last chunk has every wave out of the halo buffers)
#pragma unroll
  for (int ib = 0; ib < NB; ++ib)
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) *reinterpret_cast<f32x4*>(lds + ((4 * wave + nu) * 16 + n) * (16 * NB) + 16 * ib + 4 * g) = acc[ib][nu];
  }
  __syncthreads();
  // ---- epilogue, all eight waves: thread (tile, channel) of the 16 tiles x 16 NB channels
  if (tid < 256 * NB) {
    constexpr int CW = 16 * NB;
    const int c = tid & (CW - 1), tile = tid / CW;
    float m[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) m[p] = lds[(p * 16 + tile) * CW + c];
    float s0[4], s1[4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      s0[nu] = (m[0 * 4 + nu] + m[1 * 4 + nu]) + m[2 * 4 + nu];
      s1[nu] = (m[1 * 4 + nu] - m[2 * 4 + nu]) - m[3 * 4 + nu];
    }
    const float bias = a.bias[CW * nb + c];
    float y[2][2];
    y[0][0] = (s0[0] + s0[1]) + s0[2] + bias;
    y[0][1] = (s0[1] - s0[2]) - s0[3] + bias;
    y[1][0] = (s1[0] + s1[1]) + s1[2] + bias;
    y[1][1] = (s1[1] - s1[2]) - s1[3] + bias;
    const int py = y0 + 2 * (tile >> 2), px = x0 + 2 * (tile & 3);
    float pooled = 0.f;  // MaxPool2dWithSamePadding pads with zeros (common.py:93-96): a window that reaches past the image includes a 0
    bool any_out = false, first = true;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jx = 0; jx < 2; ++jx) {
        float v = y[i][jx];
        if (a.relu) v = fmaxf(v, 0.f);
        const bool in = py + i < a.H && px + jx < a.W;
        if (a.head_w) lds[EXCH + ((2 * (tile >> 2) + i) * 8 + 2 * (tile & 3) + jx) * (CW + 1) + c] = v;
        if (in) {
          if (!a.skip_dst) a.dst[((size_t)(b * a.H + py + i) * a.W + px + jx) * a.coutp + CW * nb + c] = v;
          pooled = first ? v : fmaxf(pooled, v);
          first = false;
        } else {
          any_out = true;
        }
      }
    if (a.dst_pool && py < a.H && px < a.W) {
      if (any_out) pooled = fmaxf(pooled, 0.f);
      const int Hp = (a.H + 1) >> 1, Wp = (a.W + 1) >> 1;
      a.dst_pool[((size_t)(b * Hp + (py >> 1)) * Wp + (px >> 1)) * a.coutp + CW * nb + c] = pooled;
    }
  }
  // ---- fused 1x1 head (the workgroup holds every channel of its 64 pixels: Cout = 16 NB): head_dst[b][o][y][x] = (sigmoid)(sum_c head_w[o][c] out[y][x][c] + head_b[o]), NCHW;
  // wave = head channel (its weights are wave-uniform scalar loads), lane = pixel
  if (a.head_w) {  // (workgroup-uniform)
    constexpr int CW = 16 * NB;
    __syncthreads();
    const int pl = tid & 63;
    const int py = y0 + (pl >> 3), px = x0 + (pl & 7);
    const float* const yrow = lds + EXCH + pl * (CW + 1);
    for (int o = wave; o < a.head_cout; o += 8) {
      const float* const hw = a.head_w + (size_t)o * a.head_wcp;
      float sum = 0.f;
#pragma unroll
      for (int cc = 0; cc < CW; ++cc) sum = fmaf(hw[cc], yrow[cc], sum);
      sum += a.head_b[o];
      if (a.head_sigmoid) sum = 1.f / (1.f + expf(-sum));
      if (py < a.H && px < a.W) a.head_dst[(((size_t)b * a.head_cout + o) * a.H + py) * a.W + px] = sum;
    }
  }
}

// Shapes the kernel takes: exact fp32, any H x W (whole 2 x 2 tiles are computed, stores are masked), channel counts padded to 16; fused pool, unread full-resolution
// output, two sources, a half-resolution second source (even H, W), a fused 1x1 head behind a conv of at most 32 output channels.  No accumulate, no ReLU mask (inference plans only).
bool sm_fits(const ConvArgs& a) {
  if (a.head_w && !(a.coutp <= 32 && a.head_wcp == a.coutp && a.head_cout >= 1 && a.head_b && a.head_dst)) return false;  // a fused head needs every channel of a pixel in one workgroup
  if (!a.wpack_sm || a.accumulate || a.relu_mask_src || (a.coutp & 15) || (a.c0p & 15) || (a.c1p & 15) || a.coutp < 16 || a.c0p < 16) return false;
  if (a.src1_lowres && (!a.src1 || (a.H & 1) || (a.W & 1))) return false;
  if (!a.src1 && a.c1p) return false;
  const uint64_t px = (uint64_t)a.B * a.H * a.W;
  if (px >= 0x7FFFFFFFull) return false;
  const uint64_t units = (uint64_t)a.B * ((a.H + 7) / 8) * ((a.W + 7) / 8) * (a.coutp / 16);
  return units < 0x7FFFFFFFull;
}
// Estimated launch body in microseconds (the dispatch floor excluded, as in wino4_fits): rounds of the chip x (prologue + epilogue ~2.5 us + ~0.8 us per 32-channel chunk)
static int sm_nb(const ConvArgs& a, int n_cu) {  // N blocks per workgroup
  const long units1 = (long)a.B * ((a.H + 7) / 8) * ((a.W + 7) / 8) * (a.coutp / 16);
  if (a.head_w) return a.coutp / 16;  // (sm_fits: Cout <= 32)
  return ((a.coutp & 31) == 0 && units1 > n_cu) ? 2 : 1;
}
// Estimated launch body in microseconds (the dispatch floor excluded, as in wino4_fits): rounds of the chip x (prologue + epilogue ~2.3 us + ~0.95 us per 32-channel chunk, ~1.5 us
// for a chunk of the half-resolution source; two N blocks: 1.3 / 1.9); calibrated on cfg1's eleven layers (profiles/r5_cfg1_smallmap.txt)
double sm_cost_us(const ConvArgs& a, int n_cu) {
  const int nb = sm_nb(a, n_cu);
  const double units = (double)a.B * ((a.H + 7) / 8) * ((a.W + 7) / 8) * (a.coutp / 16 / nb);
  const int nch = (a.c0p + a.c1p + 31) / 32;
  const int nlow = a.src1_lowres ? nch - a.c0p / 32 : 0;
  return std::max(1.0, units / (double)n_cu) * (2.3 + (nb == 2 ? 1.3 : 0.95) * (nch - nlow) + (nb == 2 ? 1.9 : 1.5) * nlow);
}
int launch_conv3x3_sm(const ConvArgs& a, hipStream_t s) {
  PH_REQUIRE(sm_fits(a), "conv3x3_sm_kernel does not take this shape (ask sm_fits first)");
  int n_cu = 0;
  if (device_cu_count(&n_cu) != PH_OK || n_cu <= 0) n_cu = 256;
  const int nb = sm_nb(a, n_cu);
  const unsigned units = (unsigned)a.B * ((a.H + 7) / 8) * ((a.W + 7) / 8) * (a.coutp / 16 / nb);
  if (a.src1_lowres && nb == 2)
    hipLaunchKernelGGL((conv3x3_sm_kernel<true, 2>), dim3(units), dim3(512), 0, s, a);
  else if (a.src1_lowres)
    hipLaunchKernelGGL((conv3x3_sm_kernel<true, 1>), dim3(units), dim3(512), 0, s, a);
  else if (nb == 2)
    hipLaunchKernelGGL((conv3x3_sm_kernel<false, 2>), dim3(units), dim3(512), 0, s, a);
  else
    hipLaunchKernelGGL((conv3x3_sm_kernel<false, 1>), dim3(units), dim3(512), 0, s, a);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

}  // namespace ph
