// 3x3 "same" convolution with 32 (NB = 2) or 16 (NB = 1) output channels and 16 or 32 input channels (the second encoder block of a
// filters = 16 UNet at half resolution: 2 of cfg3's 16 conv launches; in training also the first block's 16 -> 16 conv and the data
// gradients of both) as Winograd F(2x2, 3x3) on v_mfma_f32_16x16x4_f32.
//
// Reference semantics: SimpleConvBlock's Conv2d(k3, "same") + bias + ReLU and the 2x2 max pool behind it
// (architectures/encoder_decoder.py:108-121, architectures/common.py:69-107), as conv3x3_wino2d_kernel (wino2d_kernels.hip).
//
// With K this short (one or two 16-channel chunks per tile) the wave-split form of conv3x3_wino2d_kernel would spend more
// time in its cross-wave epilogue than in the K loop.  Here a wave keeps ALL sixteen Winograd positions of its own tiles:
// the 16x16x4 MFMA's accumulator is 4 registers, so 16 positions x 2 N blocks of 16 channels are the same 128 accumulator
// registers, for M = 16 Winograd tiles (8 x 2 tiles = 16 x 4 pixels) per wave.  The output transform, bias, ReLU and the 2x2
// max pool (a Winograd tile IS a pool window) are then lane-local register arithmetic -- no exchange, no epilogue barrier.
//   * Workgroup = 8 waves = 16 x 32 pixels; the transformed weights (32 KiB per chunk) stay in LDS for the whole launch
//     (persistent workgroups); the raw halo (18 x 34 pixels x 16 channels, 40 KiB) is double-buffered per (tile, chunk) unit
//     and arrives by buffer-descriptor LDS-DMA one unit ahead (out-of-image pixels: the range check's zeros).
//   * Halo layout [column parity][hy * 9 + hx / 2][16 channels]: the 64 lanes of a fragment read (tile, channel quad) cover
//     contiguous 512-byte runs.  Per Winograd row xi a lane reads the two patch rows it combines (8 ds_read_b128), 8 VALU ops
//     give the four A fragments, each feeding 8 MFMAs (2 N blocks x 4 channel groups).
#include <type_traits>

#include "common.h"
#include "net_kernels.h"

namespace ph {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int V_TW = 16, V_TH = 32;                     // workgroup tile in pixels
constexpr int V_HH = V_TH + 2;                           // halo 18 x 34 pixels (V_TW + 2 columns: nine column pairs per parity plane)
constexpr int V_PLANE_E = V_HH * 9;                     // 306 entries per column-parity plane
constexpr int V_PLANE_PIECES = (V_PLANE_E + 15) / 16;   // 20 DMA pieces (16 entries x 64 B) per plane
constexpr int V_PLANE_FLOATS = V_PLANE_PIECES * 256;    // 5120
constexpr int V_HALO_FLOATS = 2 * V_PLANE_FLOATS;       // 10240 per buffer (40 KiB)
constexpr int v_w_floats(int nb) { return 16 * nb * 256; }  // per chunk: [position][N block][lane][4] (32 KiB for two N blocks of 16 channels)

// wpack [chunk][tap 9][32][16] (pack_conv, N tile 32) -> U = G g G^T as [chunk][position xi * 4 + nu][N block < nbs][kq][n][4]
__global__ __launch_bounds__(256) void w16_pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int chunks, int nbs) {
  const int per = v_w_floats(nbs), total = chunks * per;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int chunk = i / per, r = i - chunk * per;
    const int e = r & 3, n = (r >> 2) & 15, kq = (r >> 6) & 3, nb = (r >> 8) % nbs, pos = (r >> 8) / nbs;
    const int xi = pos >> 2, nu = pos & 3;
    const int row = nb * 16 + n, kc = 4 * kq + e;
    const float* w = src + ((size_t)chunk * 9 * 32 + row) * 16 + kc;
    const int ts = 32 * 16;  // tap stride
    float h[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const float g0 = w[(0 * 3 + kx) * ts], g1 = w[(1 * 3 + kx) * ts], g2 = w[(2 * 3 + kx) * ts];
      h[kx] = xi == 0 ? g0 : (xi == 1 ? 0.5f * ((g0 + g2) + g1) : (xi == 2 ? 0.5f * ((g0 + g2) - g1) : g2));
    }
    dst[i] = nu == 0 ? h[0] : (nu == 1 ? 0.5f * ((h[0] + h[2]) + h[1]) : (nu == 2 ? 0.5f * ((h[0] + h[2]) - h[1]) : h[2]));
  }
}
int64_t w16_pack_floats(int chunks, int nbs) { return (int64_t)chunks * v_w_floats(nbs); }
int launch_w16_pack(const float* wpack, float* w16, int chunks, int nbs, hipStream_t s) {
  PH_REQUIRE(nbs == 1 || nbs == 2, "w16 pack: one or two N blocks of 16 channels");
  hipLaunchKernelGGL(w16_pack_kernel, dim3(std::min(chunks * v_w_floats(nbs) / 256, 256)), dim3(256), 0, s, wpack, w16, chunks, nbs);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// TWO: the decoder's concat((skip, x)) (encoder_decoder.py:545,556) as two K panels: chunks [0, c0p / 16) come from src0, the rest from src1 (its own descriptor and
// pixel stride) -- the full-resolution decoder level of an output-stride-1 filters-16 UNet (16 + 32 -> 16 channels), or the last level of a filters-8 one; they ran on the
// N-tile-32 F(2,3) kernel with half of its N tile empty before.  (Resident weights bound the chunk count: 16 KiB per chunk and N block next to 80 KiB of halo buffers.)
// HEAD: a 1x1 head (<= 16 output channels) on this conv's output rides in the epilogue: the lane-local result y[pixel][e] = channel 16 nb + 4 kq + e of tile li IS the B operand
// of v_mfma_f32_16x16x4_f32 (lane = (n = tile, k = kq)), so head[o][tile] = sum_c W[o][c] y[c][tile] is 4 NB MFMAs per pixel with A[i = o][k = kq] = W[o][16 nb + 4 kq + e]
// (one 16-byte read of the head's own [o][c] weights per N block, kept in registers); D = lane (tile, o quad kq), registers o = 4 kq + r: NCHW pixel-pair stores.  The
// final conv of an output-stride-1 / -2 UNet (16 -> 16, 32 -> 32) then needs no head launch, and in inference plans its output never reaches HBM (skip_dst).
template <int CHUNKS, int NB, bool TWO = false, bool HEAD = false>
__global__ __launch_bounds__(512, 2) void conv3x3_w16_kernel(ConvArgs a) {
  constexpr int V_W_FLOATS = v_w_floats(NB);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const wl = lds;                           // transformed weights, resident
  float* const hbuf = lds + CHUNKS * V_W_FLOATS;   // two halo buffers
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  float m1 = -1.f;
  asm volatile("" : "+v"(m1));  // opaque: keeps x - y as fma(y, m1, x) (v_pk_fma_f32; a plain vector subtraction is compiled to one v_sub_f32 per element)
  const int tiles_x = (a.W + V_TW - 1) / V_TW;
  const int tiles_y = (a.H + V_TH - 1) / V_TH;
  const int total = tiles_x * tiles_y * a.B;

  // ---- halo DMA: piece p = wave + 8 s (s = 0..4) of a buffer; lane -> entry (p % 20) * 16 + (lane >> 2) of parity plane p / 20, quad lane & 3
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, (int)((unsigned)(a.B * a.H * a.W) * (unsigned)(a.c0p * 4)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc((void*)(TWO ? a.src1 : a.src0), 0, (int)((unsigned)(a.B * a.H * a.W) * (unsigned)((TWO ? a.c1p : a.c0p) * 4)), 0x00020000);
  const int chunks0 = TWO ? a.c0p / 16 : CHUNKS;
  int hyx[5];  // the lane's halo pixel (hy << 8) | hx in DMA slot s, or -1 beyond the plane
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int p = wave + 8 * s;
    const int par = p >= V_PLANE_PIECES ? 1 : 0;
    const int e = (p - par * V_PLANE_PIECES) * 16 + (lane >> 2);
    const int hy = e / 9, hx = 2 * (e - hy * 9) + par;
    hyx[s] = e < V_PLANE_E ? (hy << 8) | hx : -1;
  }
  unsigned fvo[5], fvo1[5];  // (fvo1: dead unless TWO; a template-dependent array bound here makes this clang drop the host-side instantiations)
  // workgroup ids are dealt round-robin over the 8 XCDs: every XCD walks a contiguous range of tiles (neighbours' halo overlap meets in its L2)
  auto tile_of = [&](int vid) { return (total & 7) == 0 ? (vid & 7) * (total >> 3) + (vid >> 3) : vid; };
  auto point = [&](int vid) {  // per-lane offsets of tile vid's halo
    int t = tile_of(vid);
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = tx * V_TW, y0 = ty * V_TH;
    const unsigned base = (unsigned)((b * a.H + y0) * a.W + x0) * (unsigned)(a.c0p * 4);  // < 4 GiB (w16_fits); the halo terms below wrap consistently
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int hy = hyx[s] >> 8, hx = hyx[s] & 255;
      const int gy = y0 + hy - 1, gx = x0 + hx - 1;
      const bool in = hyx[s] >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      fvo[s] = in ? base + (unsigned)(((hy - 1) * a.W + (hx - 1)) * a.c0p * 4 + (lane & 3) * 16) : 0xFFFFFF00u;
      if (TWO) fvo1[s] = in ? (unsigned)(((b * a.H + gy) * a.W + gx) * a.c1p * 4 + (lane & 3) * 16) : 0xFFFFFF00u;
    }
  };
  auto dma = [&](float* buf, int chunk) {
    if (TWO && chunk >= chunks0) {  // workgroup-uniform
#pragma unroll
      for (int s = 0; s < 5; ++s)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, (__attribute__((address_space(3))) void*)(buf + (wave + 8 * s) * 256), 16, fvo1[s], (chunk - chunks0) * 64, 0, 0);
      return;
    }
#pragma unroll
    for (int s = 0; s < 5; ++s)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(buf + (wave + 8 * s) * 256), 16, fvo[s], chunk * 64, 0, 0);
  };

  // ---- A side: lane (li, kq) is Winograd tile (tile row 2 wave + (li >> 3), column li & 7), channel quad kq of the chunk
  const int lbase = ((4 * wave + 2 * (li >> 3)) * 9 + (li & 7)) * 16 + kq * 4;
  auto a_off = [&](int r, int c) { return lbase + r * 9 * 16 + (c >> 1) * 16 + (c & 1) * V_PLANE_FLOATS; };
  const int boff = lane * 4;

  // ---- prologue: weights (CHUNKS * 16 NB pieces of 1 KiB), first unit's halo
  {
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpack_w16, 0, CHUNKS * V_W_FLOATS * 4, 0x00020000);
    constexpr int WP = CHUNKS * 2 * NB;  // 1-KiB pieces per wave
#pragma unroll
    for (int s = 0; s < WP; ++s)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(wl + (wave * WP + s) * 256), 16, (unsigned)lane * 16u, (wave * WP + s) * 1024, 0, 0);
  }
  f32x4 hw[HEAD ? NB : 1];  // HEAD: row li of the head's weights, columns 16 nb + 4 kq .. + 3 (zeros for rows past head_cout)
  if (HEAD) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      hw[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (li < a.head_cout) hw[nb] = *reinterpret_cast<const f32x4*>(a.head_w + (size_t)li * a.head_wcp + nb * 16 + 4 * kq);
    }
  }
  int vid = blockIdx.x;
  point(vid);
  dma(hbuf, 0);
  __syncthreads();
  int par = 0;  // halo buffer of the running unit
  while (true) {
    const int nvid = vid + gridDim.x;
    const bool has_next = nvid < total;
    f32x4 acc[16][NB];
    if constexpr (CHUNKS > 1) {  // (one chunk: a position's first MFMA takes C = 0 instead -- 64 NB register moves per tile that the matrix pipe waits for)
#pragma unroll
      for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[p][nb][e] = 0.f;
    }
#pragma unroll 1
    for (int ch = 0; ch < CHUNKS; ++ch) {  // (one copy of the body: unrolled, the two-chunk variant spills)
      // the next unit's halo streams into the other buffer under this unit's MFMAs
      if (ch + 1 < CHUNKS) {
        dma(hbuf + (par ^ 1) * V_HALO_FLOATS, ch + 1);
      } else if (has_next) {
        point(nvid);
        dma(hbuf + (par ^ 1) * V_HALO_FLOATS, 0);
      }
      const float* hb = hbuf + par * V_HALO_FLOATS;
      const float* wc = wl + ch * V_W_FLOATS + boff;
      f32x4 da[4], db[4], av[4], bf[2][NB];
      auto read_rows = [&](int xi) {  // the two patch rows Winograd row xi combines
        const int ra = xi == 0 ? 0 : (xi == 2 ? 2 : 1), rb = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          da[c] = *reinterpret_cast<const f32x4*>(hb + a_off(ra, c));
          db[c] = *reinterpret_cast<const f32x4*>(hb + a_off(rb, c));
        }
      };
      auto load_b = [&](int pos, int fb) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bf[fb][nb] = *reinterpret_cast<const f32x4*>(wc + (pos * NB + nb) * 256);
      };
      read_rows(0);
      load_b(0, 0);
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        f32x4 t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) t[c] = xi == 1 ? da[c] + db[c] : db[c] * m1 + da[c];  // (x - y as fma(y, -1, x), exact: packs two lanes per instruction)
        av[0] = t[2] * m1 + t[0];
        av[1] = t[1] + t[2];
        av[2] = t[1] * m1 + t[2];
        av[3] = t[3] * m1 + t[1];
#pragma unroll
        for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(av[c]));
        // Pinned order of a step (one position, 8 MFMAs): its first MFMAs, then the LDS reads of the next step (and, in a row's first
        // step, the next row's patch rows), then the rest -- a read is never waited for in the step that issues it.
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
          const int pos = xi * 4 + nu, cur = pos & 1;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[pos][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[cur][nb][0], av[nu][0], CHUNKS > 1 ? acc[pos][nb] : f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);  // D[channel][tile]
          __builtin_amdgcn_sched_barrier(0);
          if (pos + 1 < 16) load_b(pos + 1, cur ^ 1);
          if (nu == 0 && xi + 1 < 4) read_rows(xi + 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 1; j < 4; ++j)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[pos][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[cur][nb][j], av[nu][j], acc[pos][nb], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();  // the next unit's halo has landed (vmcnt(0)) and every wave is done with this buffer
      par ^= 1;
    }
    // ---- epilogue, lane-local.  The product is accumulated transposed (weights = A operand): lane (li, kq) holds tile li of the
    // wave's 16 (tile row li >> 3, column li & 7) and register e is channel nb * 16 + 4 kq + e -- four consecutive channels per
    // lane, so an output pixel's 16 channels of an N block leave as one 16-byte store per lane (64 contiguous bytes per tile).
    {
      int t = tile_of(vid);
      const int tx = t % tiles_x;
      t /= tiles_x;
      const int ty = t % tiles_y;
      const int b = t / tiles_y;
      const int x0 = tx * V_TW, y0 = ty * V_TH + 4 * wave;
      const bool interior = (x0 + V_TW <= a.W) && (y0 + 4 <= a.H);
      const int Hp = (a.H + 1) >> 1, Wp = (a.W + 1) >> 1;
      const int oy = y0 + 2 * (li >> 3), ox = x0 + 2 * (li & 7);
      f32x4 hd[2][2];
      if (HEAD) {
#pragma unroll
        for (int aa = 0; aa < 2; ++aa)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) hd[aa][bb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int co = nb * 16 + 4 * kq;
        const f32x4 bias = *reinterpret_cast<const f32x4*>(a.bias + co);
        f32x4 y[2][2];
        {  // A^T M A on register quads (four channels at a time), the subtractions as packed fmas
          f32x4 P[4][2];
#pragma unroll
          for (int xi = 0; xi < 4; ++xi) {
            P[xi][0] = (acc[xi * 4 + 0][nb] + acc[xi * 4 + 1][nb]) + acc[xi * 4 + 2][nb];
            P[xi][1] = acc[xi * 4 + 3][nb] * m1 + (acc[xi * 4 + 2][nb] * m1 + acc[xi * 4 + 1][nb]);
          }
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            const f32x4 v0 = ((P[0][bb] + P[1][bb]) + P[2][bb]) + bias, v1 = (P[3][bb] * m1 + (P[2][bb] * m1 + P[1][bb])) + bias;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              y[0][bb][e] = a.relu ? fmaxf(v0[e], 0.f) : v0[e];
              y[1][bb][e] = a.relu ? fmaxf(v1[e], 0.f) : v1[e];
            }
          }
        }
        if (HEAD) {
#pragma unroll
          for (int aa = 0; aa < 2; ++aa)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
              for (int e = 0; e < 4; ++e) hd[aa][bb] = __builtin_amdgcn_mfma_f32_16x16x4f32(hw[nb][e], y[aa][bb][e], hd[aa][bb], 0, 0, 0);
        }
        if (!a.skip_dst) {
#pragma unroll
          for (int aa = 0; aa < 2; ++aa)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
              if (interior || (oy + aa < a.H && ox + bb < a.W)) {
                const size_t o = ((size_t)(b * a.H + oy + aa) * a.W + ox + bb) * a.coutp + co;
                f32x4 v = y[aa][bb];
                if (a.relu_mask_src) {  // backward: this launch completes the gradient of a conv + ReLU output -- that ReLU's mask rides in the (lane-local) epilogue
                  const f32x4 f = *reinterpret_cast<const f32x4*>(a.relu_mask_src + o);
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = f[e] > 0.f ? v[e] : 0.f;
                }
                *reinterpret_cast<f32x4*>(a.dst + o) = v;
              }
        }
        if (a.dst_pool) {  // "same" padding: zeros beyond the image
          f32x4 pm;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v00 = y[0][0][e], v01 = y[0][1][e], v10 = y[1][0][e], v11 = y[1][1][e];
            if (!interior) {
              v00 = (oy < a.H && ox < a.W) ? v00 : 0.f;
              v01 = (oy < a.H && ox + 1 < a.W) ? v01 : 0.f;
              v10 = (oy + 1 < a.H && ox < a.W) ? v10 : 0.f;
              v11 = (oy + 1 < a.H && ox + 1 < a.W) ? v11 : 0.f;
            }
            pm[e] = fmaxf(fmaxf(v00, v01), fmaxf(v10, v11));
          }
          const int py = oy >> 1, px = ox >> 1;
          if (interior || (py < Hp && px < Wp)) *reinterpret_cast<f32x4*>(a.dst_pool + ((size_t)(b * Hp + py) * Wp + px) * a.coutp + co) = pm;
        }
      }
      if (HEAD) {  // NCHW fp32: this lane holds head channels 4 kq + r of its tile's 2 x 2 pixels; a pixel pair (ox even) is one aligned 8-byte store when W is even
        const bool pair_ok = (a.W & 1) == 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 4 * kq + r;
          if (o < a.head_cout) {
            const float hb = a.head_b[o];
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) {
              if (oy + aa >= a.H || ox >= a.W) continue;
              float v0 = hd[aa][0][r] + hb, v1 = hd[aa][1][r] + hb;
              if (a.head_sigmoid) {
                v0 = 1.f / (1.f + expf(-v0));
                v1 = 1.f / (1.f + expf(-v1));
              }
              float* const hp = a.head_dst + (((size_t)b * a.head_cout + o) * a.H + oy + aa) * a.W + ox;
              const bool two = ox + 1 < a.W;
              if (two && pair_ok) {
                typedef float f32x2_t __attribute__((ext_vector_type(2)));
                f32x2_t v;
                v[0] = v0;
                v[1] = v1;
                *reinterpret_cast<f32x2_t*>(hp) = v;
              } else {
                hp[0] = v0;
                if (two) hp[1] = v1;
              }
            }
          }
        }
      }
    }
    if (!has_next) break;
    vid = nvid;
  }
}

static int v_cu_count(int* out) { return device_cu_count(out); }

int prepare_w16_kernels() {
  const void* ks[10] = {reinterpret_cast<const void*>(conv3x3_w16_kernel<1, 1, false, true>), reinterpret_cast<const void*>(conv3x3_w16_kernel<2, 2, false, true>),
                        reinterpret_cast<const void*>(conv3x3_w16_kernel<1, 1>), reinterpret_cast<const void*>(conv3x3_w16_kernel<2, 1>),
                       reinterpret_cast<const void*>(conv3x3_w16_kernel<1, 2>), reinterpret_cast<const void*>(conv3x3_w16_kernel<2, 2>),
                       reinterpret_cast<const void*>(conv3x3_w16_kernel<2, 1, true>), reinterpret_cast<const void*>(conv3x3_w16_kernel<3, 1, true>),
                       reinterpret_cast<const void*>(conv3x3_w16_kernel<4, 1, true>), reinterpret_cast<const void*>(conv3x3_w16_kernel<2, 2, true>)};
  for (const void* k : ks) {
    const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(w16) failed: %s", hipGetErrorString(e));
      return PH_E_HIP;
    }
  }
  return PH_OK;
}

// chunk budget: the transformed weights stay resident in LDS next to the two 40-KiB halo buffers -- 16 KiB per chunk and N block of 16 channels
bool w16_shape_ok(int c0p, int c1p, int coutp) {
  if (coutp != 16 && coutp != 32) return false;
  const int chunks = (c0p + c1p) / 16;
  if (c1p == 0) return c0p == 16 || c0p == 32;
  return (c0p % 16) == 0 && (c1p % 16) == 0 && c0p > 0 && chunks >= 2 && chunks <= (coutp == 16 ? 4 : 2);
}
bool w16_fits(const ConvArgs& a) {
  const uint64_t px = (uint64_t)a.B * a.H * a.W;
  return a.wpack_w16 && a.bn == 32 && w16_shape_ok(a.c0p, a.src1 ? a.c1p : 0, a.coutp) && (a.src1 != nullptr) == (a.c1p > 0) && !a.src1_lowres && !a.accumulate &&
         px * (uint64_t)a.c0p * 4 < 0xFFFFFF00ull && px * (uint64_t)a.c1p * 4 < 0xFFFFFF00ull && px < 0x7FFFFFFFull;
}

// shapes whose 1x1 head can ride in the epilogue (the HEAD instantiations)
bool w16_takes_head(const ConvArgs& a) {
  return w16_fits(a) && !a.src1 && !a.dst_pool && !a.relu_mask_src && ((a.c0p == 16 && a.coutp == 16) || (a.c0p == 32 && a.coutp == 32));
}

int launch_conv3x3_w16(const ConvArgs& a, hipStream_t s) {
  PH_REQUIRE(w16_fits(a), "w16: Cout 16 or 32 (one N tile), 16 or 32 input channels from one source or 2 - 4 chunks from two, no accumulate, sources below 4 GiB");
  int n_cu = 0;
  const int rc = v_cu_count(&n_cu);
  if (rc != PH_OK) return rc;
  const int tiles = ((a.W + V_TW - 1) / V_TW) * ((a.H + V_TH - 1) / V_TH) * a.B;
  const int chunks = (a.c0p + a.c1p) / 16, nbs = a.coutp / 16;
  const size_t ldsb = (size_t)(chunks * v_w_floats(nbs) + 2 * V_HALO_FLOATS) * sizeof(float);
  const dim3 grid(std::min(tiles, n_cu));
  if (a.head_w) {
    PH_REQUIRE(w16_takes_head(a) && a.head_cout >= 1 && a.head_cout <= 16 && a.head_wcp == a.coutp && a.head_b && a.head_dst, "w16: a fused head needs 16 -> 16 or 32 -> 32 channels from one source, no pool, and at most 16 head channels");
    if (nbs == 1)
      hipLaunchKernelGGL((conv3x3_w16_kernel<1, 1, false, true>), grid, dim3(512), ldsb, s, a);
    else
      hipLaunchKernelGGL((conv3x3_w16_kernel<2, 2, false, true>), grid, dim3(512), ldsb, s, a);
  } else if (a.src1 && nbs == 2)
    hipLaunchKernelGGL((conv3x3_w16_kernel<2, 2, true>), grid, dim3(512), ldsb, s, a);
  else if (a.src1 && chunks == 2)
    hipLaunchKernelGGL((conv3x3_w16_kernel<2, 1, true>), grid, dim3(512), ldsb, s, a);
  else if (a.src1 && chunks == 3)
    hipLaunchKernelGGL((conv3x3_w16_kernel<3, 1, true>), grid, dim3(512), ldsb, s, a);
  else if (a.src1)
    hipLaunchKernelGGL((conv3x3_w16_kernel<4, 1, true>), grid, dim3(512), ldsb, s, a);
  else if (chunks == 1 && nbs == 1)
    hipLaunchKernelGGL((conv3x3_w16_kernel<1, 1>), grid, dim3(512), ldsb, s, a);
  else if (chunks == 2 && nbs == 1)
    hipLaunchKernelGGL((conv3x3_w16_kernel<2, 1>), grid, dim3(512), ldsb, s, a);
  else if (chunks == 1)
    hipLaunchKernelGGL((conv3x3_w16_kernel<1, 2>), grid, dim3(512), ldsb, s, a);
  else
    hipLaunchKernelGGL((conv3x3_w16_kernel<2, 2>), grid, dim3(512), ldsb, s, a);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

}  // namespace ph
