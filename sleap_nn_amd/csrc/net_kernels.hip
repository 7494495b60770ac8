// Encoder-decoder forward kernels for gfx950 (MI355X, CDNA4).
//
// Activations live in HBM as NHWC fp32 with the channel count padded to a multiple of 16
// ("Cp"); pad channels are always written as exact zeros (zero weights, zero bias), so no
// buffer ever needs clearing.  All dense convolutions run as implicit GEMMs on the matrix
// cores through v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate), which is
// what lets the confidence maps stay within 1e-4 of the reference's ATen-CPU results.
//
// Reference semantics implemented here (paths relative to talmolab/sleap-nn):
//   conv3x3 + bias + ReLU ............ architectures/encoder_decoder.py:108-121,494-510
//   concat(skip, upsampled) .......... encoder_decoder.py:545,556 (two K-panels, no copy)
//   2x2 max pool "same" .............. architectures/common.py:69-107 (zero pad if odd)
//   bilinear x2 (align_corners=False)  encoder_decoder.py:431-435
//   ConvTranspose2d(k3,s2,p1,op1) .... encoder_decoder.py:439-461
//   1x1 head conv .................... architectures/heads.py:58-67 (head1x1_mfma_kernel, f16_kernels.hip)
//   uint8 -> float /255 .............. data/normalization.py:7-35
#include <type_traits>

#include "act_format.h"
#include "common.h"
#include "net_kernels.h"
#include "f16_kernels.h"

namespace ph {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------
// K1: 3x3 "same" convolution as implicit GEMM on MFMA.
//   GEMM view: M = pixels (tile 8 rows x 32 cols), N = BN output channels, K = 9 * Cin_p.
//   256 threads = 4 waves; wave w owns image rows 2w, 2w+1 of the tile (two 32-pixel M
//   tiles) times all BN/32 N tiles -> 2*(BN/32) accumulators of 16 VGPRs.
//   K loop: 16-channel chunks; per chunk the (8+2)x(32+2) input halo and the 9 x BN x 16
//   weight panel are staged in LDS with rows padded from 16 to 20 dwords, which makes the
//   ds_read_b128 fragment reads bank-conflict free (20*i mod 64 hits 16 distinct 16-B slots).
//   One ds_read_b128 feeds four MFMAs: lanes 0-31 hold channels c..c+3 of "their" pixel /
//   output channel, lanes 32-63 channels c+4..c+7, i.e. MFMA j contracts channels
//   {c+j, c+4+j} -- the same pairing on the A and the B side.
// ---------------------------------------------------------------------------------------
constexpr int TH = 8, TW = 32, KC = 16, LROW = 20;
constexpr int HALO_W = TW + 2, HALO_H = TH + 2;

// Work item `id` (a workgroup id, or a persistent workgroup's virtual id) -> (pixel tile, N tile).  Workgroups are dealt
// round-robin over the 8 XCDs, each with a private L2 (ids b and b + 8 share an XCD).  Every XCD gets a CONTIGUOUS range of pixel
// tiles, walked in order with the N tiles of one pixel tile on consecutive local ids: at any time an XCD works on neighbouring
// tiles, so the N tiles' identical halo reads AND the one-to-two-pixel halo overlap between neighbouring tiles hit in its L2
// instead of going out to HBM once per tile (round 1 dealt tile t to XCD t % 8: neighbours never met; 1.56x over-fetch on the
// 512x512 layers).  (Speed only; any placement is correct.)
__device__ __forceinline__ void deal_tile(int id, int tiles, int nt_count, int* tile, int* ntile) {
  if ((tiles & 7) == 0) {
    const int xcd = id & 7, j = id >> 3;
    *ntile = j % nt_count;
    *tile = xcd * (tiles >> 3) + j / nt_count;
  } else {
    *ntile = id % nt_count;
    *tile = id / nt_count;
  }
}
__device__ __forceinline__ void decode_block(int tiles, int nt_count, int* tile, int* ntile) { deal_tile(blockIdx.x, tiles, nt_count, tile, ntile); }

// Fused 2x2/2 max pool of a wave's two accumulator rows (rows 2w, 2w+1 of the tile; C/D map
// x = (r&3) + 8*(r>>2) + 4*(lane>>5)): x pairs are register pairs, y pairs the two M tiles.
// Out-of-image elements count as 0 = the reference's zero pad (values are >= 0 after ReLU).
template <int NT>
__device__ __forceinline__ void store_pooled(const ConvArgs& a, f32x16 (&acc)[2][NT], int b, int x0, int y_top, int ntile_base, int lx, int lh, bool interior) {
  const int Hp = (a.H + 1) >> 1, Wp = (a.W + 1) >> 1;
  const int py = y_top >> 1;
  if (py >= Hp) return;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int co = ntile_base + n * 32 + lx;
    if (co >= a.coutp) continue;
    float* prow = a.dst_pool + ((size_t)(b * Hp + py) * Wp) * a.coutp + co;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int px = (x0 >> 1) + 4 * rq + 2 * lh + pr;
        const int r0 = 4 * rq + 2 * pr;
        float v00 = acc[0][n][r0], v01 = acc[0][n][r0 + 1], v10 = acc[1][n][r0], v11 = acc[1][n][r0 + 1];
        if (!interior) {
          const int x = x0 + 2 * pr + 8 * rq + 4 * lh;
          const bool xa = x < a.W, xb = x + 1 < a.W, ya = y_top < a.H, yb = y_top + 1 < a.H;
          v00 = (xa && ya) ? v00 : 0.f;
          v01 = (xb && ya) ? v01 : 0.f;
          v10 = (xa && yb) ? v10 : 0.f;
          v11 = (xb && yb) ? v11 : 0.f;
        }
        if (px < Wp) prow[(size_t)px * a.coutp] = fmaxf(fmaxf(v00, v01), fmaxf(v10, v11));
      }
  }
}

template <int BN>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldsA = lds;                            // HALO_H*HALO_W rows of LROW
  float* ldsB = lds + HALO_H * HALO_W * LROW;   // 9*BN rows of LROW
  constexpr int NT = BN / 32;
  constexpr int A_ITEMS = HALO_H * HALO_W * 4;        // 16-B pieces of the input halo per chunk
  constexpr int A_SLOTS = (A_ITEMS + 255) / 256;      // 6
  constexpr int B_ITEMS = 9 * BN * 4;                 // 16-B pieces of the weight panel per chunk
  constexpr int B_SLOTS = (B_ITEMS + 255) / 256;      // 9 (BN=64) / 5 (BN=32)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int tiles_x = (a.W + TW - 1) / TW;
  const int tiles_y = (a.H + TH - 1) / TH;
  int t, ntile;
  decode_block(tiles_x * tiles_y * a.B, (a.coutp + BN - 1) / BN, &t, &ntile);
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;

  f32x16 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int chunks0 = a.c0p / KC;
  const int chunks1 = a.c1p / KC;
  const int nchunks = chunks0 + chunks1;
  const float* wbase = a.wpack + (size_t)ntile * nchunks * (9 * BN * KC);

  // ---- per-thread staging plan, fixed for the whole K loop.  Item i = tid + 256*s is the
  // 16-B piece q = i&3 (= tid&3) of halo pixel i>>2.  Out-of-image pixels load a clamped
  // (valid) address and are zeroed by a select afterwards: no branch around any load, so all
  // loads of a chunk are in flight together (a branch per load would serialise them).
  const int q4 = (tid & 3) * 4;
  int a_pix[A_SLOTS];      // clamped pixel index (b*H + y)*W + x
  unsigned a_ok = 0;       // bit s: pixel inside the image
#pragma unroll
  for (int s = 0; s < A_SLOTS; ++s) {
    const int pix = (tid >> 2) + 64 * s;
    const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
    const int gy = y0 + hy - 1, gx = x0 + hx - 1;
    const bool in = (pix < HALO_H * HALO_W) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    a_ok |= (in ? 1u : 0u) << s;
    const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
    a_pix[s] = (b * a.H + cy) * a.W + cx;
  }

  f32x4 ra[A_SLOTS], rb[B_SLOTS];
  // source of chunk `ch`: (pointer, padded channel count, channel offset) + its weight panel
  const float* p_src;
  const float* p_w;
  int p_cp, p_coff;
  auto select_chunk = [&](int ch) {
    if (ch < chunks0) {
      p_src = a.src0;
      p_cp = a.c0p;
      p_coff = ch * KC;
    } else {
      p_src = a.src1;
      p_cp = a.c1p;
      p_coff = (ch - chunks0) * KC;
    }
    p_w = wbase + (size_t)ch * (9 * BN * KC);
  };
  auto load_slot = [&](int slot) {  // slot < A_SLOTS: input halo piece, else weight piece
    if (slot < A_SLOTS) {
      ra[slot] = *reinterpret_cast<const f32x4*>(p_src + (size_t)a_pix[slot] * p_cp + p_coff + q4);
    } else {
      const int sb = slot - A_SLOTS;
      const int i = min(tid + 256 * sb, B_ITEMS - 1);
      rb[sb] = *reinterpret_cast<const f32x4*>(p_w + i * 4);
    }
  };
  auto commit = [&]() {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) {
      const int pix = (tid >> 2) + 64 * s;
      if (A_ITEMS % 256 == 0 || pix < HALO_H * HALO_W)
        *reinterpret_cast<f32x4*>(ldsA + pix * LROW + q4) = ((a_ok >> s) & 1u) ? ra[s] : z;
    }
#pragma unroll
    for (int s = 0; s < B_SLOTS; ++s) {
      const int row = (tid >> 2) + 64 * s;
      if (B_ITEMS % 256 == 0 || row < 9 * BN) *reinterpret_cast<f32x4*>(ldsB + row * LROW + q4) = rb[s];
    }
  };

  const int lx = lane & 31;
  const int lh = lane >> 5;

  unsigned long long t0 = 0, r0 = 0;
  if (a.clock_probe) {
    t0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  select_chunk(0);
#pragma unroll
  for (int sl = 0; sl < A_SLOTS + B_SLOTS; ++sl) load_slot(sl);
#ifdef PH_STAMP
  unsigned long long st_commit = 0, st_bar1 = 0, st_mfma = 0, st_bar2 = 0, st_prev = __builtin_amdgcn_s_memtime();
#define PH_STAMP_AT(acc)                                         \
  {                                                              \
    __builtin_amdgcn_sched_barrier(0);                           \
    const unsigned long long _t = __builtin_amdgcn_s_memtime();  \
    acc += _t - st_prev;                                         \
    st_prev = _t;                                                \
    __builtin_amdgcn_sched_barrier(0);                           \
  }
#else
#define PH_STAMP_AT(acc)
#endif
  for (int ch = 0; ch < nchunks; ++ch) {
    commit();          // waits for the chunk's loads, fills LDS
    PH_STAMP_AT(st_commit)
    __syncthreads();
    PH_STAMP_AT(st_bar1)
    // Next chunk's loads are issued ONE PER MFMA STEP below (a burst of 15 loads per wave would
    // block the wave in the vector-memory issue queue while the matrix pipe idles).  On the last
    // chunk the same chunk is fetched again (harmless, keeps the loop branch-free).
    select_chunk(min(ch + 1, nchunks - 1));
    // ---- 18 steps (9 taps x 2 channel groups) x 4 MFMA k-steps.  Fragments are double
    // buffered in registers: the ds_read_b128s of step i+1 are issued before the MFMAs of
    // step i, so a wave that has its SIMD to itself still issues MFMAs back to back.
    f32x4 af[2][2], bf[2][NT];
    auto load_frags = [&](int step, int buf) {
      const int tap = step >> 1, g = step & 1;
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int m = 0; m < 2; ++m)
        af[buf][m] = *reinterpret_cast<const f32x4*>(ldsA + ((2 * wave + m + ky) * HALO_W + lx + kx) * LROW + g * 8 + lh * 4);
#pragma unroll
      for (int n = 0; n < NT; ++n)
        bf[buf][n] = *reinterpret_cast<const f32x4*>(ldsB + (tap * BN + n * 32 + lx) * LROW + g * 8 + lh * 4);
    };
    load_frags(0, 0);
#pragma unroll
    for (int step = 0; step < 18; ++step) {
      const int cur = step & 1;
      if (step + 1 < 18) load_frags(step + 1, cur ^ 1);
      if (step < A_SLOTS + B_SLOTS) load_slot(step);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][m][j], bf[cur][n][j], acc[m][n], 0, 0, 0);
      // pin the step boundary: next step's fragment reads + one prefetch load, then this step's
      // MFMAs (without the fence the scheduler sinks all prefetch loads behind the last MFMA)
      __builtin_amdgcn_sched_group_barrier(0x100, 2 + NT, 0);   // next step's fragment reads first
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT, 0);   // a few MFMAs
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);        // address arithmetic of the prefetch load
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);        // one prefetch load
      __builtin_amdgcn_sched_group_barrier(0x008, 6 * NT, 0);   // rest of the MFMAs
      __builtin_amdgcn_sched_barrier(0);
    }
    PH_STAMP_AT(st_mfma)
    __syncthreads();
    PH_STAMP_AT(st_bar2)
  }
#ifdef PH_STAMP
  if (a.clock_probe && lane == 0) {
    const size_t bi = ((size_t)blockIdx.x) * 4 + wave;
    a.clock_probe[4 * bi + 0] = st_commit;
    a.clock_probe[4 * bi + 1] = st_bar1;
    a.clock_probe[4 * bi + 2] = st_mfma;
    a.clock_probe[4 * bi + 3] = st_bar2;
  }
#else
  unsigned long long t_loop = 0;
  if (a.clock_probe) t_loop = __builtin_amdgcn_s_memtime();
#endif
  // ---- epilogue: bias (+ReLU) in place in the accumulators, then NHWC stores straight from
  // them.  C/D map: col = lane&31 (channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pixel x
  // inside the 32-wide tile).  Interior tiles take a branch-free path: 64 stores issued back
  // to back (a guarded store per element makes hipcc wait vmcnt(0) between stores, which
  // serialises ~64 HBM round trips per wave).
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const float bias = a.bias[ntile * BN + n * 32 + lx];  // bias is padded to a multiple of BN
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[m][n][r] + bias;
        acc[m][n][r] = a.relu ? fmaxf(v, 0.f) : v;
      }
  }
  const bool interior = (x0 + TW <= a.W) && (y0 + TH <= a.H) && ((ntile + 1) * BN <= a.coutp);
  if (a.dst_pool) store_pooled<NT>(a, acc, b, x0, y0 + 2 * wave, ntile * BN, lx, lh, interior);
  if (interior) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float* drow = a.dst + ((size_t)(b * a.H + y0 + 2 * wave + m) * a.W + x0 + 4 * lh) * a.coutp + ntile * BN + n * 32 + lx;
        if (!a.accumulate) {
#pragma unroll
          for (int r = 0; r < 16; ++r) drow[(size_t)((r & 3) + 8 * (r >> 2)) * a.coutp] = acc[m][n][r];
        } else {  // backward pass: add to the gradient already in dst
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][n][r] += drow[(size_t)((r & 3) + 8 * (r >> 2)) * a.coutp];
#pragma unroll
          for (int r = 0; r < 16; ++r) drow[(size_t)((r & 3) + 8 * (r >> 2)) * a.coutp] = acc[m][n][r];
        }
      }
  } else {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int co = ntile * BN + n * 32 + lx;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int y = y0 + 2 * wave + m;
        float* drow = a.dst + ((size_t)(b * a.H + min(y, a.H - 1)) * a.W) * a.coutp + min(co, a.coutp - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (x < a.W && y < a.H && co < a.coutp) {
            float* dp = drow + (size_t)x * a.coutp;
            *dp = a.accumulate ? (*dp + acc[m][n][r]) : acc[m][n][r];
          }
        }
      }
    }
  }
#ifndef PH_STAMP
  if (a.clock_probe && lane == 0) {  // shader clock = dt_memtime / dt_memrealtime * 100 MHz
    __builtin_amdgcn_s_waitcnt(0);  // include the store drain
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    const size_t bi = ((size_t)blockIdx.x) * 4 + wave;
    a.clock_probe[4 * bi + 0] = t_loop - t0;   // kernel entry -> end of K loop
    a.clock_probe[4 * bi + 1] = t1 - t0;       // kernel entry -> stores drained
    a.clock_probe[4 * bi + 2] = r1 - r0;       // same span on the 100 MHz counter
    a.clock_probe[4 * bi + 3] = t0;            // absolute start (for occupancy timelines)
  }
#endif
}

int launch_conv3x3(const ConvArgs& a, hipStream_t s) {
  const int tiles = ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH) * a.B;
  if (a.bn == 64) {
    const size_t lds = (HALO_H * HALO_W + 9 * 64) * LROW * sizeof(float);
    dim3 grid(tiles * ((a.coutp + 63) / 64));
    hipLaunchKernelGGL(conv3x3_mfma_kernel<64>, grid, dim3(256), lds, s, a);
  } else {
    const size_t lds = (HALO_H * HALO_W + 9 * 32) * LROW * sizeof(float);
    dim3 grid(tiles * ((a.coutp + 31) / 32));
    hipLaunchKernelGGL(conv3x3_mfma_kernel<32>, grid, dim3(256), lds, s, a);
  }
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// K1d: the same implicit GEMM with LDS-DMA staging and a double-buffered LDS ring.
//   512 threads = 8 waves (two per SIMD), tile 16 rows x 32 cols x BN channels; wave w owns rows
//   2w, 2w+1.  Per 16-channel chunk the (16+2)x(32+2) halo and the 9 x BN x 16 weight panel go
//   global -> LDS directly (global_load_lds_dwordx4: no VGPR staging, no ds_write pass), one
//   1-KiB piece per MFMA step, into the buffer that is NOT being read; one barrier per chunk.
//   LDS image ("quad-major pieces"): rows (halo pixels / weight rows) are grouped in pieces of
//   16; inside a 1-KiB piece the four 16-B channel quads are stored quad-major:
//       byte(row r, quad q) = (r >> 4) * 1024 + q * 256 + (r & 15) * 16.
//   A DMA wave-instruction writes 64 x 16 B contiguously, so the permutation costs nothing (it
//   is applied to the per-lane SOURCE address), and a ds_read_b128 of 16 rows that are distinct
//   mod 16 (any 16 consecutive pixels; the b128 lane groups {0-3,12-15,20-27} too) touches all
//   16 slots of the 256-B bank row exactly once: conflict-free without padding.
// ---------------------------------------------------------------------------------------
constexpr int D_TH = 16, D_HALO_H = D_TH + 2;
constexpr int D_NPIX = D_HALO_H * HALO_W;            // 612
constexpr int D_A_PIECES = (D_NPIX + 15) / 16;       // 39

template <int BN>
__global__ __launch_bounds__(512, 2) void conv3x3_mfma_dma_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NT = BN / 32;
  constexpr int B_PIECES = 9 * BN / 16;
  constexpr int PIECES = D_A_PIECES + B_PIECES;
  constexpr int A_SLOTS = (D_A_PIECES + 7) / 8;   // 5: input pieces  p = min(w + 8 s, last)
  constexpr int B_SLOTS = (B_PIECES + 7) / 8;     // 5 / 3: weight pieces
  constexpr int SLOTS = A_SLOTS + B_SLOTS;
  constexpr int BUF_FLOATS = PIECES * 256;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (a.W + TW - 1) / TW;
  const int tiles_y = (a.H + D_TH - 1) / D_TH;
  int t, ntile;
  decode_block(tiles_x * tiles_y * a.B, (a.coutp + BN - 1) / BN, &t, &ntile);
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int x0 = tx * TW, y0 = ty * D_TH;

  f32x16 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int chunks0 = a.c0p / KC;
  const int chunks1 = a.c1p / KC;
  const int nchunks = chunks0 + chunks1;
  const float* wbase = a.wpack_dma + (size_t)ntile * nchunks * (9 * BN * KC);

  // ---- DMA plan (branch-free: slot kinds are compile-time, surplus slots repeat the last
  // piece, out-of-image pixels read a zero page).  Lane L of a piece holds quad q = L >> 4 of
  // row (p*16 + (L & 15)).
  const int dq = lane >> 4, dr = lane & 15;
  int a_pix[A_SLOTS];
  unsigned a_ok = 0;
#pragma unroll
  for (int s = 0; s < A_SLOTS; ++s) {
    const int p = min(wave + 8 * s, D_A_PIECES - 1);
    const int pix = p * 16 + dr;
    const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
    const int gy = y0 + hy - 1, gx = x0 + hx - 1;
    const bool in = (pix < D_NPIX) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    a_ok |= (in ? 1u : 0u) << s;
    const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
    a_pix[s] = (b * a.H + cy) * a.W + cx;
  }
  const float* p_src;
  const float* p_w;
  int p_cp, p_coff;
  auto select_chunk = [&](int ch) {
    if (ch < chunks0) {
      p_src = a.src0;
      p_cp = a.c0p;
      p_coff = ch * KC;
    } else {
      p_src = a.src1;
      p_cp = a.c1p;
      p_coff = (ch - chunks0) * KC;
    }
    p_w = wbase + (size_t)ch * (9 * BN * KC);
  };
  auto dma_slot = [&](int s, float* buf) {
    const float* g;
    int p;
    if (s < A_SLOTS) {  // compile-time after unrolling
      p = min(wave + 8 * s, D_A_PIECES - 1);
      const float* real = p_src + (size_t)a_pix[s] * p_cp + p_coff + dq * 4;
      const float* zero = a.zeros + dq * 4;
      g = ((a_ok >> s) & 1u) ? real : zero;
    } else {
      const int pb = min(wave + 8 * (s - A_SLOTS), B_PIECES - 1);
      p = D_A_PIECES + pb;
      g = p_w + pb * 256 + lane * 4;  // weights are packed in LDS order
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(buf + p * 256), 16, 0, 0);
  };

  // ---- fragment read offsets (floats, buffer-relative)
  const int lx = lane & 31, lh = lane >> 5;
  int offA[4][3];  // [row 2w + rr][kx] for q = lh
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int pix = (2 * wave + rr) * HALO_W + lx + kx;
      offA[rr][kx] = (pix >> 4) * 256 + lh * 64 + (pix & 15) * 4;
    }
  const int offB = D_A_PIECES * 256 + (lx >> 4) * 256 + lh * 64 + (lx & 15) * 4;  // + tap*(BN/16)*256 + nt*512 + g*128

  float* buf0 = lds;
  float* buf1 = lds + BUF_FLOATS;

  select_chunk(0);
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) dma_slot(s, buf0);
  __syncthreads();  // drains the DMA (vmcnt(0)) and publishes the buffer

  // Waves i and i + 4 share SIMD i (measured); an LDS-DMA wave-instruction occupies its wave for
  // 60-185 cycles, so the two partners must not issue theirs at the same point of a step: the
  // upper four waves ("late") place the step's DMA piece after 6*NT of its MFMAs, the lower four
  // after 2*NT -- while one partner feeds the DMA engine the other feeds the MFMA pipe.
  auto k_loop = [&](auto late) {
    constexpr bool LATE = decltype(late)::value;
    for (int ch = 0; ch < nchunks; ++ch) {
      float* cur = (ch & 1) ? buf1 : buf0;
      float* nxt = (ch & 1) ? buf0 : buf1;
      select_chunk(min(ch + 1, nchunks - 1));  // last chunk: refetch itself into the idle buffer (harmless, branch-free)
      f32x4 af[2][2], bf[2][NT];
      auto load_frags = [&](int step, int fb) {
        const int tap = step >> 1, g = step & 1;
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int m = 0; m < 2; ++m) af[fb][m] = *reinterpret_cast<const f32x4*>(cur + offA[m + ky][kx] + g * 128);
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[fb][n] = *reinterpret_cast<const f32x4*>(cur + offB + tap * (BN / 16) * 256 + n * 512 + g * 128);
      };
      load_frags(0, 0);
#pragma unroll
      for (int step = 0; step < 18; ++step) {
        const int fcur = step & 1;
        if (step + 1 < 18) load_frags(step + 1, fcur ^ 1);
        if (step < SLOTS) dma_slot(step, nxt);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[fcur][m][j], bf[fcur][n][j], acc[m][n], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 + NT, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, LATE ? 6 * NT : 2 * NT, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, LATE ? 2 * NT : 6 * NT, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();  // vmcnt(0) + barrier: next buffer landed everywhere, this one is free again
    }
  };
  if (a.dma_stagger && wave >= 4)  // wave-uniform; both paths execute the same barriers
    k_loop(std::true_type{});
  else
    k_loop(std::false_type{});

  // ---- epilogue (same C/D map as the register-staged kernel)
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const float bias = a.bias[ntile * BN + n * 32 + lx];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[m][n][r] + bias;
        acc[m][n][r] = a.relu ? fmaxf(v, 0.f) : v;
      }
  }
  const bool interior = (x0 + TW <= a.W) && (y0 + D_TH <= a.H) && ((ntile + 1) * BN <= a.coutp);
  if (a.dst_pool) store_pooled<NT>(a, acc, b, x0, y0 + 2 * wave, ntile * BN, lx, lh, interior);
  if (interior) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float* drow = a.dst + ((size_t)(b * a.H + y0 + 2 * wave + m) * a.W + x0 + 4 * lh) * a.coutp + ntile * BN + n * 32 + lx;
        if (!a.accumulate) {
#pragma unroll
          for (int r = 0; r < 16; ++r) drow[(size_t)((r & 3) + 8 * (r >> 2)) * a.coutp] = acc[m][n][r];
        } else {  // backward pass: add to the gradient already in dst
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][n][r] += drow[(size_t)((r & 3) + 8 * (r >> 2)) * a.coutp];
#pragma unroll
          for (int r = 0; r < 16; ++r) drow[(size_t)((r & 3) + 8 * (r >> 2)) * a.coutp] = acc[m][n][r];
        }
      }
  } else {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int co = ntile * BN + n * 32 + lx;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int y = y0 + 2 * wave + m;
        float* drow = a.dst + ((size_t)(b * a.H + min(y, a.H - 1)) * a.W) * a.coutp + min(co, a.coutp - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (x < a.W && y < a.H && co < a.coutp) {
            float* dp = drow + (size_t)x * a.coutp;
            *dp = a.accumulate ? (*dp + acc[m][n][r]) : acc[m][n][r];
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// K1p: the same kernel as a PERSISTENT workgroup (one per CU) that walks its share of the
// (pixel tile, N tile) list.  During the last K chunk of a tile the idle LDS buffer receives the
// first chunk of the NEXT tile, so a new tile starts with its data already in LDS (the one-tile
// kernel pays an exposed ~2 us DMA round trip per tile: 5-10 % on layers with few input channels)
// and the epilogue's stores overlap that fetch.
// ---------------------------------------------------------------------------------------
template <int BN>
__global__ __launch_bounds__(512, 2) void conv3x3_mfma_dma_persist_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NT = BN / 32;
  constexpr int B_PIECES = 9 * BN / 16;
  constexpr int PIECES = D_A_PIECES + B_PIECES;
  constexpr int A_SLOTS = (D_A_PIECES + 7) / 8;
  constexpr int B_SLOTS = (B_PIECES + 7) / 8;
  constexpr int SLOTS = A_SLOTS + B_SLOTS;
  constexpr int BUF_FLOATS = PIECES * 256;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (a.W + TW - 1) / TW;
  const int tiles_y = (a.H + D_TH - 1) / D_TH;
  const int tiles = tiles_x * tiles_y * a.B;
  const int ntc = (a.coutp + BN - 1) / BN;
  const int total = tiles * ntc;
  const int chunks0 = a.c0p / KC;
  const int chunks1 = a.c1p / KC;
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
    deal_tile(vid, tiles, ntc, &t, &ntile);  // XCD-aware dealing over virtual workgroup ids
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    P.b = t / tiles_y;
    P.x0 = tx * TW;
    P.y0 = ty * D_TH;
    P.ntile = ntile;
    P.a_ok = 0;
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) {
      const int p = min(wave + 8 * s, D_A_PIECES - 1);
      const int pix = p * 16 + dr;
      const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
      const int gy = P.y0 + hy - 1, gx = P.x0 + hx - 1;
      const bool in = (pix < D_NPIX) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      P.a_ok |= (in ? 1u : 0u) << s;
      const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
      P.a_pix[s] = (P.b * a.H + cy) * a.W + cx;
    }
  };

  // fetch state: which tile plan / chunk the DMA slots of the running chunk loop pull
  int f_pix[A_SLOTS];
  unsigned f_ok = 0;
  const float* p_src = a.src0;
  const float* p_w = a.wpack_dma;
  int p_cp = a.c0p, p_coff = 0;
  auto select_fetch = [&](const Plan& P, int ch) {
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) f_pix[s] = P.a_pix[s];
    f_ok = P.a_ok;
    if (ch < chunks0) {
      p_src = a.src0;
      p_cp = a.c0p;
      p_coff = ch * KC;
    } else {
      p_src = a.src1;
      p_cp = a.c1p;
      p_coff = (ch - chunks0) * KC;
    }
    p_w = a.wpack_dma + ((size_t)P.ntile * nchunks + ch) * (9 * BN * KC);
  };
  auto dma_slot = [&](int s, float* buf) {
    const float* g;
    int p;
    if (s < A_SLOTS) {  // compile-time after unrolling
      p = min(wave + 8 * s, D_A_PIECES - 1);
      const float* real = p_src + (size_t)f_pix[s] * p_cp + p_coff + dq * 4;
      const float* zero = a.zeros + dq * 4;
      g = ((f_ok >> s) & 1u) ? real : zero;
    } else {
      const int pb = min(wave + 8 * (s - A_SLOTS), B_PIECES - 1);
      p = D_A_PIECES + pb;
      g = p_w + pb * 256 + lane * 4;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(buf + p * 256), 16, 0, 0);
  };

  int offA[4][3];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int pix = (2 * wave + rr) * HALO_W + lx + kx;
      offA[rr][kx] = (pix >> 4) * 256 + lh * 64 + (pix & 15) * 4;
    }
  const int offB = D_A_PIECES * 256 + (lx >> 4) * 256 + lh * 64 + (lx & 15) * 4;
  float* buf0 = lds;
  float* buf1 = lds + BUF_FLOATS;

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
      f32x16 acc[2][NT];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
      for (int ch = 0; ch < nchunks; ++ch) {
        float* cur = ((parity + ch) & 1) ? buf1 : buf0;
        float* nxt = ((parity + ch) & 1) ? buf0 : buf1;
        if (ch + 1 < nchunks)
          select_fetch(P, ch + 1);
        else if (has_next)
          select_fetch(Pn, 0);  // the next tile's first chunk rides under this tile's last one
        else
          select_fetch(P, ch);  // nothing left: refetch (harmless, keeps the loop branch-free inside)
        f32x4 af[2][2], bf[2][NT];
        auto load_frags = [&](int step, int fb) {
          const int tap = step >> 1, g = step & 1;
          const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
          for (int m = 0; m < 2; ++m) af[fb][m] = *reinterpret_cast<const f32x4*>(cur + offA[m + ky][kx] + g * 128);
#pragma unroll
          for (int n = 0; n < NT; ++n) bf[fb][n] = *reinterpret_cast<const f32x4*>(cur + offB + tap * (BN / 16) * 256 + n * 512 + g * 128);
        };
        load_frags(0, 0);
#pragma unroll
        for (int step = 0; step < 18; ++step) {
          const int fcur = step & 1;
          if (step + 1 < 18) load_frags(step + 1, fcur ^ 1);
          if (step < SLOTS) dma_slot(step, nxt);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int n = 0; n < NT; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[fcur][m][j], bf[fcur][n][j], acc[m][n], 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2 + NT, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, LATE ? 6 * NT : 2 * NT, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, LATE ? 2 * NT : 6 * NT, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
      }
      // ---- epilogue of this tile
      const int b = P.b, x0 = P.x0, y0 = P.y0, ntile = P.ntile;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float bias = a.bias[ntile * BN + n * 32 + lx];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[m][n][r] + bias;
            acc[m][n][r] = a.relu ? fmaxf(v, 0.f) : v;
          }
      }
      const bool interior = (x0 + TW <= a.W) && (y0 + D_TH <= a.H) && ((ntile + 1) * BN <= a.coutp);
      if (a.dst_pool) store_pooled<NT>(a, acc, b, x0, y0 + 2 * wave, ntile * BN, lx, lh, interior);
      if (interior) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            float* drow = a.dst + ((size_t)(b * a.H + y0 + 2 * wave + m) * a.W + x0 + 4 * lh) * a.coutp + ntile * BN + n * 32 + lx;
            if (!a.accumulate) {
#pragma unroll
              for (int r = 0; r < 16; ++r) drow[(size_t)((r & 3) + 8 * (r >> 2)) * a.coutp] = acc[m][n][r];
            } else {
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[m][n][r] += drow[(size_t)((r & 3) + 8 * (r >> 2)) * a.coutp];
#pragma unroll
              for (int r = 0; r < 16; ++r) drow[(size_t)((r & 3) + 8 * (r >> 2)) * a.coutp] = acc[m][n][r];
            }
          }
      } else {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int co = ntile * BN + n * 32 + lx;
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const int y = y0 + 2 * wave + m;
            float* drow = a.dst + ((size_t)(b * a.H + min(y, a.H - 1)) * a.W) * a.coutp + min(co, a.coutp - 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
              if (x < a.W && y < a.H && co < a.coutp) {
                float* dp = drow + (size_t)x * a.coutp;
                *dp = a.accumulate ? (*dp + acc[m][n][r]) : acc[m][n][r];
              }
            }
          }
        }
      }
      if (!has_next) break;
      parity = (parity + nchunks) & 1;
      vid = nvid;
      P = Pn;
    }
  };
  if (a.dma_stagger && wave >= 4)
    run(std::true_type{});
  else
    run(std::false_type{});
}

// ---------------------------------------------------------------------------------------
// K1w: the persistent LDS-DMA kernel with Winograd F(2,3) along x.  Two neighbouring outputs of a row,
//   y0 = d0 g0 + d1 g1 + d2 g2,  y1 = d1 g0 + d2 g1 + d3 g2     (d = four input pixels, g = the row's three taps)
// are  y0 = m0 + m1 + m2,  y1 = m1 - m2 - m3  with
//   m0 = (d0 - d2) g0,  m1 = (d1 + d2)(g0 + g1 + g2)/2,  m2 = (d2 - d1)(g0 - g1 + g2)/2,  m3 = (d1 - d3) g2:
// four multiplications instead of six -> 12 instead of 18 MFMA K-steps per (ky, channel): 1.5x less matrix work.
//   * The GEMM's M index is an output PAIR: a wave's two rows x 32 pixels are ONE 32-row M tile (row = lane >> 4 & 1,
//     pair t = lane & 15) with four accumulator sets (m0..m3); the output transform is register arithmetic in the epilogue.
//   * The input transform happens on the fly: the raw halo tile is staged by LDS-DMA exactly as in K1p; per (ky, 8-channel
//     group) a lane reads its four pixels d0..d3 (ds_read_b128 each) and forms the four A fragments with v_sub / v_add.
//     LDS read bandwidth has ~8x headroom in K1p, the 2-way bank conflict of the stride-2 pixel reads is affordable.
//   * The weights are pre-transformed (wino_pack_kernel) and stored in step order: step q = (ky, g, xi) reads pieces
//     q * NT + n, each 32 rows x 2 quads, lane offset lh * 128 + lx * 4 floats (1 KiB contiguous per wave: conflict-free).
//   * LDS: 12 weight panels per chunk are 48 KiB, twice that plus two 39-KiB halo buffers do not fit 160 KiB.  The weights
//     live in THREE half-chunk slots (24 KiB each): a chunk reads (h, h+1), the next chunk's first half is fetched into h+2
//     during steps 0..7, and after a mid-chunk barrier (every wave is done with h) its second half goes into h.  150 KiB.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float wino_pack_element(const float* __restrict__ src /* [panel][tap 9][bn][16] */, int bn, size_t i) {
  const int nt = bn / 32;
  const int per_panel = 24 * nt * 256;
  const int panel = (int)(i / per_panel);
  int r = (int)(i - (size_t)panel * per_panel);
  const int e = r & 3, lx = (r >> 2) & 31, lh = (r >> 7) & 1;
  r >>= 8;  // piece index q * nt + n
  const int n = r % nt, q = r / nt;
  const int ky = q >> 3, g = (q >> 2) & 1, xi = q & 3;
  const int row = n * 32 + lx, kc = (2 * g + lh) * 4 + e;
  const float* w = src + (((size_t)panel * 9 + ky * 3) * bn + row) * 16 + kc;
  const float g0 = w[0], g1 = w[(size_t)bn * 16], g2 = w[(size_t)2 * bn * 16];
  if (xi == 0) return g0;
  if (xi == 1) return 0.5f * ((g0 + g2) + g1);
  if (xi == 2) return 0.5f * ((g0 + g2) - g1);
  return g2;
}
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int panels, int bn) {
  const size_t total = (size_t)panels * 24 * (bn / 32) * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) dst[i] = wino_pack_element(src, bn, i);
}
// every F(2,3) weight buffer of a model in one launch (ph_model_set_params): block -> segment by binary search, 1024 elements per block
__global__ __launch_bounds__(256) void wino_pack_multi_kernel(const PackSegment* __restrict__ seg, int n_seg) {
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
    if (i < sg.total) sg.dst[i] = wino_pack_element(sg.src, sg.bn, i);
  }
}
int launch_wino_pack_multi(const PackSegment* seg_dev, int n_seg, unsigned total_blocks, hipStream_t s) {
  if (n_seg == 0 || total_blocks == 0) return PH_OK;
  hipLaunchKernelGGL(wino_pack_multi_kernel, dim3(total_blocks), dim3(256), 0, s, seg_dev, n_seg);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}
// fused stem: w1 [tap][co 16][ci 16] -> [kernel row][m index 4][co][ci]
__global__ __launch_bounds__(256) void stem_wino_pack_kernel(const float* __restrict__ w1, float* __restrict__ w1w) {
  for (int i = threadIdx.x; i < 12 * 256; i += 256) {
    const int e = i & 255, p = i >> 8;
    const int ky = p >> 2, xi = p & 3;
    const float g0 = w1[(ky * 3 + 0) * 256 + e], g1 = w1[(ky * 3 + 1) * 256 + e], g2 = w1[(ky * 3 + 2) * 256 + e];
    w1w[i] = xi == 0 ? g0 : (xi == 1 ? 0.5f * ((g0 + g2) + g1) : (xi == 2 ? 0.5f * ((g0 + g2) - g1) : g2));
  }
}
// fused stem, F(2x2,3x3): w1 [tap][co 16][ci 16] -> U = G g G^T as [position xi * 4 + nu][co][ci]
__global__ __launch_bounds__(256) void stem_wino2d_pack_kernel(const float* __restrict__ w1, float* __restrict__ w2) {
  for (int i = threadIdx.x; i < 16 * 256; i += 256) {
    const int e = i & 255, pos = i >> 8;
    const int xi = pos >> 2, nu = pos & 3;
    float h[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const float g0 = w1[(0 * 3 + kx) * 256 + e], g1 = w1[(1 * 3 + kx) * 256 + e], g2 = w1[(2 * 3 + kx) * 256 + e];
      h[kx] = xi == 0 ? g0 : (xi == 1 ? 0.5f * ((g0 + g2) + g1) : (xi == 2 ? 0.5f * ((g0 + g2) - g1) : g2));
    }
    w2[i] = nu == 0 ? h[0] : (nu == 1 ? 0.5f * ((h[0] + h[2]) + h[1]) : (nu == 2 ? 0.5f * ((h[0] + h[2]) - h[1]) : h[2]));
  }
}
int launch_stem_wino2d_pack(const float* w1, float* w2, hipStream_t s) {
  hipLaunchKernelGGL(stem_wino2d_pack_kernel, dim3(1), dim3(256), 0, s, w1, w2);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}
int launch_stem_wino_pack(const float* w1, float* w1w, hipStream_t s) {
  hipLaunchKernelGGL(stem_wino_pack_kernel, dim3(1), dim3(256), 0, s, w1, w1w);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}
int64_t wino_pack_floats(int panels, int bn) { return (int64_t)panels * 24 * (bn / 32) * 256; }
int launch_wino_pack(const float* wpack, float* wino, int panels, int bn, hipStream_t s) {
  const int64_t n = wino_pack_floats(panels, bn);
  hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, wpack, wino, panels, bn);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// WAVES = 8: one 512-thread workgroup per CU, tile 16 x 32 pixels -- the only instantiation (a WAVES = 4 variant with two
// 256-thread workgroups per CU was measured 2 % slower in round 1 and is no longer built).
template <int BN, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void conv3x3_wino_persist_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NT = BN / 32;
  constexpr int BH_PIECES = 12 * NT;  // weight pieces of half a chunk (12 steps)
  constexpr int W_TH = 2 * WAVES;                      // tile rows: two per wave
  constexpr int W_NPIX = (W_TH + 2) * HALO_W;          // halo pixels
  constexpr int W_A_PIECES = (W_NPIX + 15) / 16;
  constexpr int A_SLOTS = (W_A_PIECES + WAVES - 1) / WAVES;
  constexpr int BH_SLOTS = (BH_PIECES + WAVES - 1) / WAVES;
  constexpr int A_FLOATS = W_A_PIECES * 256, BH_FLOATS = BH_PIECES * 256;
  static_assert(A_SLOTS + BH_SLOTS <= 12 && BH_SLOTS <= 12, "one DMA piece per step");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (a.W + TW - 1) / TW;
  const int tiles_y = (a.H + W_TH - 1) / W_TH;
  const int tiles = tiles_x * tiles_y * a.B;
  const int ntc = (a.coutp + BN - 1) / BN;
  const int total = tiles * ntc;
  const int chunks0 = a.c0p / KC;
  const int chunks1 = a.c1p / KC;
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
    deal_tile(vid, tiles, ntc, &t, &ntile);
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    P.b = t / tiles_y;
    P.x0 = tx * TW;
    P.y0 = ty * W_TH;
    P.ntile = ntile;
    P.a_ok = 0;
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) {
      const int p = min(wave + WAVES * s, W_A_PIECES - 1);
      const int pix = p * 16 + dr;
      const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
      const int gy = P.y0 + hy - 1, gx = P.x0 + hx - 1;
      const bool in = (pix < W_NPIX) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      P.a_ok |= (in ? 1u : 0u) << s;
      const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
      P.a_pix[s] = (P.b * a.H + cy) * a.W + cx;
    }
  };

  const float* p_src = a.src0;
  const float* p_w = a.wpack_wino;
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
      p_coff = ch * KC;
    } else {
      p_src = a.src1;
      p_cp = a.c1p;
      p_coff = (ch - chunks0) * KC;
    }
    p_w = a.wpack_wino + ((size_t)P.ntile * nchunks + ch) * (2 * BH_FLOATS);
  };
  auto dma_a = [&](int s, float* abuf) {
    const int p = min(wave + WAVES * s, W_A_PIECES - 1);
    const float* real = p_src + (size_t)f_pix[s] * p_cp + p_coff + dq * 4;
    const float* zero = a.zeros + dq * 4;
    const float* g = ((f_ok >> s) & 1u) ? real : zero;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(abuf + p * 256), 16, 0, 0);
  };
  auto dma_b = [&](int s, int half, float* bhalf) {
    const int pb = min(wave + WAVES * s, BH_PIECES - 1);
    const float* g = p_w + half * BH_FLOATS + pb * 256 + lane * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(bhalf + pb * 256), 16, 0, 0);
  };

  // A-side fragment offsets: lane (lx, lh) is output pair t = lx & 15 of the wave's row rr = lx >> 4; it needs halo pixels
  // (2w + rr + ky, 2t + c), c = 0..3, quad lh (+ 2g)
  int offD[3][4];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int pix = (2 * wave + (lx >> 4) + ky) * HALO_W + 2 * (lx & 15) + c;
      offD[ky][c] = (pix >> 4) * 256 + lh * 64 + (pix & 15) * 4;
    }
  const int offB = lh * 128 + lx * 4;
  float* const abuf = lds;
  float* const bbuf = lds + 2 * A_FLOATS;

  auto run = [&](auto late) {
    constexpr bool LATE = decltype(late)::value;
    Plan P, Pn;
    int vid = blockIdx.x;
    setup(vid, P);
    select_fetch(P, 0);
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) dma_a(s, abuf);
#pragma unroll
    for (int s = 0; s < BH_SLOTS; ++s) dma_b(s, 0, bbuf);
#pragma unroll
    for (int s = 0; s < BH_SLOTS; ++s) dma_b(s, 1, bbuf + BH_FLOATS);
    __syncthreads();
    int apar = 0;  // A buffer of the running chunk
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
        const float* acur = abuf + apar * A_FLOATS;
        float* anxt = abuf + (apar ^ 1) * A_FLOATS;
        const int h1 = hb == 2 ? 0 : hb + 1, h2 = h1 == 2 ? 0 : h1 + 1;
        const float* b_first = bbuf + hb * BH_FLOATS;
        const float* b_second = bbuf + h1 * BH_FLOATS;
        float* bn_first = bbuf + h2 * BH_FLOATS;   // next chunk, first half: free since the previous chunk ended
        float* bn_second = bbuf + hb * BH_FLOATS;  // next chunk, second half: free after this chunk's mid barrier
        if (ch + 1 < nchunks)
          select_fetch(P, ch + 1);
        else if (has_next)
          select_fetch(Pn, 0);  // the next tile's first chunk rides under this tile's last one
        else
          select_fetch(P, ch);  // nothing left: refetch (harmless)
        f32x4 dd[2][4], bf[2][NT], af[2];
        auto load_d = [&](int kyg, int db) {  // the four pixels of group (ky, g)
          const int ky = kyg >> 1, g = kyg & 1;
#pragma unroll
          for (int c = 0; c < 4; ++c) dd[db][c] = *reinterpret_cast<const f32x4*>(acur + offD[ky][c] + g * 128);
        };
        auto load_b = [&](int q, int fb) {
          const float* base = (q < 12 ? b_first : b_second) + (q % 12) * NT * 256 + offB;
#pragma unroll
          for (int n = 0; n < NT; ++n) bf[fb][n] = *reinterpret_cast<const f32x4*>(base + n * 256);
        };
        auto make_a = [&](int q, int fa) {
          const int db = (q >> 2) & 1, xi = q & 3;
          if (xi == 0)
            af[fa] = dd[db][0] - dd[db][2];
          else if (xi == 1)
            af[fa] = dd[db][1] + dd[db][2];
          else if (xi == 2)
            af[fa] = dd[db][2] - dd[db][1];
          else
            af[fa] = dd[db][1] - dd[db][3];
        };
        load_d(0, 0);
        load_b(0, 0);
        make_a(0, 0);
#pragma unroll
        for (int q = 0; q < 24; ++q) {
          const int fcur = q & 1;
          if (q + 1 < 24) load_b(q + 1, fcur ^ 1);
          if ((q & 3) == 1 && (q >> 2) + 1 < 6) load_d((q >> 2) + 1, ((q >> 2) + 1) & 1);
          // one DMA piece per step: the next chunk's halo and first weight half in steps 0.., its second weight half after the mid barrier
          if (q < A_SLOTS)
            dma_a(q, anxt);
          else if (q < A_SLOTS + BH_SLOTS)
            dma_b(q - A_SLOTS, 0, bn_first);
          else if (q >= 12 && q < 12 + BH_SLOTS)
            dma_b(q - 12, 1, bn_second);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[q & 3][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[fcur][j], bf[fcur][n][j], acc[q & 3][n], 0, 0, 0);
          if (q + 1 < 24) make_a(q + 1, fcur ^ 1);
          const bool has_dma = q < A_SLOTS + BH_SLOTS || (q >= 12 && q < 12 + BH_SLOTS);
          // Pinned order: the step's first MFMAs, THEN the LDS reads of the next step.  The compiler's wait before the first MFMA is
          // always lgkmcnt(0) here (an LDS-DMA instruction marks the counter out-of-order for its scoreboard), so reads issued
          // before that MFMA would be drained on the spot -- their latency exposed in every step.
          __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
          if ((q & 3) == 1 && (q >> 2) + 1 < 6)
            __builtin_amdgcn_sched_group_barrier(0x100, NT + 4, 0);
          else
            __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
          if (has_dma) {
            if (LATE) __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, LATE ? NT : 3 * NT, 0);
          } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * NT, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (q == 11) __builtin_amdgcn_s_barrier();  // every wave is done reading the first weight half: its slot may be refilled
        }
        __syncthreads();
        apar ^= 1;
        hb = hb == 0 ? 2 : hb - 1;  // (hb + 2) % 3: the slot the next chunk's first half was fetched into
      }
      // ---- epilogue of this tile: output transform, bias, ReLU, stores.  D row (r & 3) + 8 (r >> 2) + 4 lh = pair index:
      // r < 8 is the wave's first row, r >= 8 its second; the pair's outputs are pixels 2t, 2t + 1.
      const int b = P.b, x0 = P.x0, y0 = P.y0, ntile = P.ntile;
      const bool interior = (x0 + TW <= a.W) && (y0 + W_TH <= a.H) && ((ntile + 1) * BN <= a.coutp);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int co = ntile * BN + n * 32 + lx;
        const float bias = a.bias[co];
        f32x16 ya, yb;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float m1 = acc[1][n][r], m2 = acc[2][n][r];
          float va = (acc[0][n][r] + m1) + m2 + bias;
          float vb = (m1 - m2) - acc[3][n][r] + bias;
          ya[r] = a.relu ? fmaxf(va, 0.f) : va;
          yb[r] = a.relu ? fmaxf(vb, 0.f) : vb;
        }
        if (interior) {
          // Interior tile (all but the image's last row / column of tiles): straight-line stores.  One lane-dependent base
          // pointer; every store adds a wave-uniform offset -- no per-store bounds test, no exec-mask branch (the general
          // path below costs ~30 cycles of branching per store, ~5k cycles per tile and wave).
          if (a.dst_pool) {
            const int Hp = (a.H + 1) >> 1, Wp = (a.W + 1) >> 1;
            float* const pp = a.dst_pool + ((size_t)(b * Hp + ((y0 + 2 * wave) >> 1)) * Wp + (x0 >> 1) + 4 * lh) * a.coutp + co;
#pragma unroll
            for (int r = 0; r < 8; ++r)
              pp[(size_t)((r & 3) + 8 * (r >> 2)) * a.coutp] = fmaxf(fmaxf(ya[r], yb[r]), fmaxf(ya[r + 8], yb[r + 8]));
          }
          if (a.skip_dst) continue;  // only the pooled output is read
          float* const dp0 = a.dst + ((size_t)(b * a.H + y0 + 2 * wave) * a.W + x0 + 8 * lh) * a.coutp + co;
          if (!a.accumulate) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float* dp = dp0 + (size_t)((r >> 3) * a.W + 2 * (r & 3) + 16 * ((r >> 2) & 1)) * a.coutp;
              dp[0] = ya[r];
              dp[a.coutp] = yb[r];
            }
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float* dp = dp0 + (size_t)((r >> 3) * a.W + 2 * (r & 3) + 16 * ((r >> 2) & 1)) * a.coutp;
              ya[r] += dp[0];
              yb[r] += dp[a.coutp];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float* dp = dp0 + (size_t)((r >> 3) * a.W + 2 * (r & 3) + 16 * ((r >> 2) & 1)) * a.coutp;
              dp[0] = ya[r];
              dp[a.coutp] = yb[r];
            }
          }
          continue;
        }
        if (a.dst_pool) {  // fused 2x2/2 max pool ("same" padding: zeros beyond the image)
          const int Hp = (a.H + 1) >> 1, Wp = (a.W + 1) >> 1;
          const int yt = y0 + 2 * wave, py = yt >> 1;
          if (py < Hp && co < a.coutp) {
            float* prow = a.dst_pool + ((size_t)(b * Hp + py) * Wp) * a.coutp + co;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const int t = (r & 3) + 8 * (r >> 2) + 4 * lh;
              const int x = x0 + 2 * t, px = x >> 1;
              float v00 = ya[r], v01 = yb[r], v10 = ya[r + 8], v11 = yb[r + 8];
              if (!interior) {
                const bool xa = x < a.W, xb = x + 1 < a.W, yya = yt < a.H, yyb = yt + 1 < a.H;
                v00 = (xa && yya) ? v00 : 0.f;
                v01 = (xb && yya) ? v01 : 0.f;
                v10 = (xa && yyb) ? v10 : 0.f;
                v11 = (xb && yyb) ? v11 : 0.f;
              }
              if (px < Wp)
                prow[(size_t)px * a.coutp] = fmaxf(fmaxf(v00, v01), fmaxf(v10, v11));
            }
          }
        }
        if (a.skip_dst) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int y = y0 + 2 * wave + (r >> 3);
          const int t = (r & 3) + 8 * ((r >> 2) & 1) + 4 * lh;
          const int x = x0 + 2 * t;
          if (interior || (y < a.H && co < a.coutp)) {
            float* dp = a.dst + ((size_t)(b * a.H + y) * a.W + x) * a.coutp + co;
            if (interior || x < a.W) dp[0] = a.accumulate ? dp[0] + ya[r] : ya[r];
            if (interior || x + 1 < a.W) dp[a.coutp] = a.accumulate ? dp[a.coutp] + yb[r] : yb[r];
          }
        }
      }
      if (!has_next) break;
      vid = nvid;
      P = Pn;
    }
  };
  if (WAVES == 8 && a.dma_stagger && wave >= 4)
    run(std::true_type{});
  else
    run(std::false_type{});
}

// ---------------------------------------------------------------------------------------
// 16 -> 16 channel 3x3 conv (the full-resolution layer of a filters=16 encoder and its data gradient): the
// 32/64-wide N tiles of the kernels above are half padding there.  v_mfma_f32_16x16x4_f32 computes the
// transposed product D[co][pixel] = sum_k W[co][k] X[k][pixel]: the weights (A operand) live in 36 registers
// for the whole launch, a lane's B operand for the four K steps of a tap is ONE ds_read_b128 (its pixel,
// channels 4kg..4kg+3; 64 lanes read 1 KiB contiguous: no bank conflicts), and the accumulator is four
// consecutive output channels of one pixel: bias, ReLU and a float4 NHWC store, 1 KiB contiguous per wave.
//   Tile 8 x 32 pixels per 256-thread workgroup (halo 10 x 34 x 16 channels = 21.25 KiB LDS, four workgroups
//   per CU cover each other's staging); wave w owns rows 2w, 2w+1 as four independent 16-pixel chains.
//   Workgroups walk tiles with a grid stride so the weight registers are loaded once.
// ---------------------------------------------------------------------------------------
constexpr int C16_TH = 8, C16_TW = 32, C16_HH = C16_TH + 2, C16_HW = C16_TW + 2;
__global__ __launch_bounds__(256, 4) void conv3x3_c16_kernel(ConvArgs a) {
  __shared__ float sX[C16_HH * C16_HW * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, kg = lane >> 4;
  float wr[9][4];  // A[i = co = n][k = kg] of K step ks <-> input channel 4*kg + ks
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wr[tap][ks] = a.w16[(tap * 16 + 4 * kg + ks) * 16 + n];
  const f32x4 bias4 = *reinterpret_cast<const f32x4*>(a.bias + 4 * kg);
  const int tiles_x = (a.W + C16_TW - 1) / C16_TW, tiles_y = (a.H + C16_TH - 1) / C16_TH;
  const int n_tiles = tiles_x * tiles_y * a.B;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    int t = tile;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = tx * C16_TW, y0 = ty * C16_TH;
    for (int i = tid; i < C16_HH * C16_HW * 4; i += 256) {
      const int pix = i >> 2, q = i & 3;
      const int hy = pix / C16_HW, hx = pix - hy * C16_HW;
      const int gy = y0 + hy - 1, gx = x0 + hx - 1;
      const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
      const f32x4 v = *reinterpret_cast<const f32x4*>(a.src0 + ((size_t)(b * a.H + cy) * a.W + cx) * 16 + q * 4);
      *reinterpret_cast<f32x4*>(sX + pix * 16 + q * 4) = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = bias4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int py = 2 * wave + (g >> 1), px = (g & 1) * 16 + n;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(sX + ((py + ky) * C16_HW + px + kx) * 16 + 4 * kg);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[tap][ks], xv[ks], acc[g], 0, 0, 0);
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int gy = y0 + 2 * wave + (g >> 1), gx = x0 + (g & 1) * 16 + n;
      if (gy < a.H && gx < a.W) {
        f32x4 v = acc[g];
        if (a.relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
        f32x4* d = reinterpret_cast<f32x4*>(a.dst + ((size_t)(b * a.H + gy) * a.W + gx) * 16 + 4 * kg);
        if (a.accumulate) v += *d;
        *d = v;
      }
    }
    __syncthreads();
  }
}

static int cu_count(int* out);
static int launch_conv3x3_c16(const ConvArgs& a, hipStream_t s) {
  int n_cu = 0;
  const int rc = cu_count(&n_cu);
  if (rc != PH_OK) return rc;
  const int tiles = ((a.W + C16_TW - 1) / C16_TW) * ((a.H + C16_TH - 1) / C16_TH) * a.B;
  hipLaunchKernelGGL(conv3x3_c16_kernel, dim3(std::min(tiles, n_cu * 4)), dim3(256), 0, s, a);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

static int cu_count(int* out) { return device_cu_count(out); }

// Kernel choice for a 3x3 "same" conv on the LDS-DMA family.  Which variant runs is decided by fields of `a` that the
// model runtime fills from the handle's options (ph_model_set_option); there is no process-global state here.
// which kernel launch_conv3x3_dma picks: 4 F(4x4,3x3), 1 wave-private F(2x2,3x3), 0 conv3x3_c16, 2 wave-split F(2x2,3x3), 3 the F(2,3) / direct halo kernels
static int conv3x3_dma_route(const ConvArgs& a) {
  const bool wino = a.use_wino && a.wpack_wino && (a.bn == 64 || a.use_wino != 2);  // use_wino 2: Winograd for the N-tile-64 layers only
  if (wino && a.persist && a.use_wino4 && a.c0p + a.c1p >= a.wino4_min_cin && wino4_fits(a)) return 4;
  if (wino && a.persist && a.use_w16 && w16_fits(a)) return 1;
  if (a.use_c16 && a.w16 && a.c0p == 16 && a.coutp == 16 && !a.src1 && !a.dst_pool) return 0;
  if (wino && a.persist && a.use_wino2d && a.wpack_wino2 && a.bn == 64 && a.c0p + a.c1p >= 32 && wino2d_fits(a)) return 2;
  return 3;
}
// The small-map kernel (conv3x3_sm_kernel) against the kernel launch_conv3x3_dma would run, both priced in microseconds of launch body as in wino4_fits (rounds of the chip x unit time,
// + ~8 us per extra launch: a split-K second stage, a bilinear x2 the other kernel cannot fold).  use_sm = 2 forces it wherever the shape fits (tests, A/B).
bool conv3x3_takes_sm(const ConvArgs& a) {
  if (!a.use_sm || !sm_fits(a)) return false;
  if (a.use_sm >= 2) return true;
  int n_cu = 0;
  if (device_cu_count(&n_cu) != PH_OK || n_cu <= 0) n_cu = 256;
  ConvArgs hi = a;  // what runs otherwise takes the up-sampled tensor, unless it is the F(4x4,3x3) kernel
  hi.src1_lowres = 0;
  const int r = a.src1_lowres && conv3x3_dma_route(a) == 4 ? 4 : conv3x3_dma_route(hi);
  const double extra = (a.src1_lowres && r != 4) ? 6.0 : 0.0;
  const double Qn = (a.c0p + a.c1p) / 4.0;
  const long ntc = (a.coutp + 63) / 64;
  const bool may_split = a.split_scratch != nullptr;
  double other;
  if (r == 4) {
    const int ks = may_split ? wino4_ksplit_shape(a.B, a.H, a.W, a.c0p + a.c1p, a.coutp, a.splitk, n_cu) : 1;
    const long t4 = (long)((a.H + 15) / 16) * ((a.W + 31) / 32) * a.B * ntc;
    other = (double)((t4 * ks + n_cu - 1) / n_cu) * (11.0 + 1.5 * Qn / ks) + (ks > 1 ? 8.0 : 0.0);
  } else if (r == 2) {
    const int ks = may_split ? wino2d_ksplit_shape(a.B, a.H, a.W, a.c0p + a.c1p, a.coutp, a.splitk, n_cu) : 1;
    const long t2 = (long)((a.H + 15) / 16) * ((a.W + 15) / 16) * a.B * ntc;
    other = (double)((t2 * ks + n_cu - 1) / n_cu) * (8.0 + 1.0 * Qn / ks) + (ks > 1 ? 8.0 : 0.0);
  } else if (r == 1) {  // the wave-private kernel: 16 x 32-pixel tiles, ~9 us + 3 us per 16 input channels (cfg1's 128 x 128 levels: 32 workgroups, 9 - 15 us)
    const long t1 = (long)((a.H + 31) / 32) * ((a.W + 15) / 16) * a.B;
    other = (double)((t1 + n_cu - 1) / n_cu) * (6.0 + 3.0 * (a.c0p + a.c1p) / 16.0);
  } else {
    return false;  // the 16 -> 16 / F(2,3) / direct kernels keep their layers
  }
  if ((r == 4 || r == 2) && (a.coutp & 63) != 0 && (a.coutp & 63) <= 32) other *= 0.6;  // the half-empty N tile: the waves of the missing half skip their MFMAs (published workload, 96 -> 32 at 160 x 280 x 4: 55 us measured, 94 modelled)
  return sm_cost_us(a, n_cu) < other + extra;
}
bool conv3x3_dma_honours_mask(const ConvArgs& a) {  // the three F(2x2,3x3) / F(4x4,3x3) kernels store lane-locally; the fused pool / head epilogues are forward-only
  const int r = conv3x3_dma_route(a);
  return (r == 1 || r == 2 || r == 4) && !a.dst_pool && !a.head_w && !a.skip_dst;
}
bool conv3x3_dma_is_f2x2(const ConvArgs& a) { const int r = conv3x3_dma_route(a); return r == 1 || r == 2; }
bool conv3x3_dma_is_wino2d(const ConvArgs& a) { return conv3x3_dma_route(a) == 2; }
bool conv3x3_dma_is_wino4(const ConvArgs& a) { return conv3x3_dma_route(a) == 4; }
bool conv3x3_dma_is_w16_head(const ConvArgs& a) { return conv3x3_dma_route(a) == 1 && w16_takes_head(a); }
// PH_KV_* code of the kernel launch_conv3x3_dma would run (ph_model_last_kernels)
int conv3x3_dma_variant(const ConvArgs& a) {
  switch (conv3x3_dma_route(a)) {
    case 0: return PH_KV_C16;
    case 1: return PH_KV_W16;
    case 2: return wino2d_ksplit(a) > 1 ? PH_KV_WINO2D_KS : PH_KV_WINO2D;
    case 4: return PH_KV_WINO4;
    default: break;
  }
  const bool wino = a.use_wino && a.wpack_wino && (a.bn == 64 || a.use_wino != 2);
  return (a.persist && wino) ? PH_KV_WINO1D : PH_KV_DIRECT;
}

int launch_conv3x3_dma(const ConvArgs& a, hipStream_t s) {
  const int route = conv3x3_dma_route(a);
  PH_REQUIRE(!a.relu_mask_src || ((route == 1 || route == 2 || route == 4) && !a.dst_pool && !a.head_w && !a.skip_dst),
             "relu_mask_src is applied by the F(2x2,3x3) / F(4x4,3x3) kernels' plain stores only (ask conv3x3_dma_honours_mask first)");
  PH_REQUIRE(!a.head_w || (route == 2 && !a.dst_pool && a.coutp == 64 && a.bn == 64 && a.head_cout >= 1 && a.head_cout <= 32 && a.head_wcp == 64 && a.head_b && a.head_dst) ||
                 (route == 1 && w16_takes_head(a) && a.head_cout >= 1 && a.head_cout <= 16 && a.head_wcp == a.coutp && a.head_b && a.head_dst),
             "a fused head needs the F(2x2,3x3) kernel on 64 output channels (<= 32 head channels) or the wave-private kernel on 16 -> 16 / 32 -> 32 channels (<= 16 head channels): ask conv3x3_dma_is_wino2d / _is_w16_head first");
  if (route == 1) return launch_conv3x3_w16(a, s);
  if (route == 0) return launch_conv3x3_c16(a, s);
  if (route == 2) return launch_conv3x3_wino2d(a, s);
  if (route == 4) return launch_conv3x3_wino4(a, s);
  PH_REQUIRE(!a.src1_lowres, "a half-resolution second source is only taken by the F(4x4,3x3) kernel (ask conv3x3_dma_is_wino4 first)");
  const int tiles = ((a.W + TW - 1) / TW) * ((a.H + D_TH - 1) / D_TH) * a.B;
  const int ntc = (a.coutp + a.bn - 1) / a.bn;
  const bool wino = a.use_wino && a.wpack_wino && (a.bn == 64 || a.use_wino != 2);
  int n_cu = 0;
  if (a.persist) {
    const int rc = cu_count(&n_cu);
    if (rc != PH_OK) return rc;
  }
  const dim3 grid_p(std::min(tiles * ntc, n_cu)), grid_1(tiles * ntc);
  const size_t lds_w = (size_t)(2 * D_A_PIECES + 3 * 12 * (a.bn / 32)) * 1024;
  const size_t lds_d = (size_t)2 * (D_A_PIECES + 9 * a.bn / 16) * 1024;
  if (a.bn == 64) {
    if (a.persist && wino)
      hipLaunchKernelGGL((conv3x3_wino_persist_kernel<64, 8>), grid_p, dim3(512), lds_w, s, a);
    else if (a.persist)
      hipLaunchKernelGGL(conv3x3_mfma_dma_persist_kernel<64>, grid_p, dim3(512), lds_d, s, a);
    else
      hipLaunchKernelGGL(conv3x3_mfma_dma_kernel<64>, grid_1, dim3(512), lds_d, s, a);
  } else {
    if (a.persist && wino)
      hipLaunchKernelGGL((conv3x3_wino_persist_kernel<32, 8>), grid_p, dim3(512), lds_w, s, a);
    else if (a.persist)
      hipLaunchKernelGGL(conv3x3_mfma_dma_persist_kernel<32>, grid_p, dim3(512), lds_d, s, a);
    else
      hipLaunchKernelGGL(conv3x3_mfma_dma_kernel<32>, grid_1, dim3(512), lds_d, s, a);
  }
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// K0f: fused first encoder block (the HBM-heaviest part of the network, 16 channels at full
// resolution): image tile -> /255 -> conv3x3(Cin->16)+ReLU computed straight into the LDS A
// tile (it never exists in HBM) -> conv3x3(16->16) on v_mfma_f32_16x16x4_f32 (N = 16 exactly:
// no padded half tile) -> bias + ReLU -> optional full-res store -> 2x2 max pool in registers
// -> pooled NHWC store.  Per frame this reads 1 B/px and writes 16 B/px instead of the
// unfused 64+64+64+64+16 B/px.
//   Tile: 8 rows x 32 cols of conv outputs per 256-thread block; wave w owns rows 2w, 2w+1 as
//   four 16-pixel M tiles.  16x16x4 maps: A[i = lane&15][k = lane>>4], B[k = lane>>4][n = lane&15],
//   D: col = lane&15 (channel), row = 4*(lane>>4) + reg (pixel).  One ds_read_b128 per operand
//   feeds 4 MFMAs (lane group g reads channels 4g..4g+3; MFMA j contracts {j, 4+j, 8+j, 12+j}).
// ---------------------------------------------------------------------------------------
// WINO: conv1 as Winograd F(2,3) along x (see K1w): an M row of the 16x16x4 MFMA is an output PAIR, a wave's two rows are two M
// tiles x four accumulators (m0..m3); 96 instead of 144 MFMAs per wave, the transform is four f32x4 adds per (kernel row, image row).
// One output channel of one pixel in the stem's output format (16 logical channels; the plain-fp16 format pads to 32
// channels and nobody else would write the upper 16, so they are written as zeros here).
__device__ __forceinline__ void stem_store(const StemArgs& a, void* base, size_t pix, int c, float v) {
  if (a.out_fmt == FMT_SPLIT)
    store1<FMT_SPLIT>(base, pix, 16, c, v);
  else if (a.out_fmt == FMT_F16) {
    store1<FMT_F16>(base, pix, 32, c, v);
    store1<FMT_F16>(base, pix, 32, c + 16, 0.f);
  } else
    store1<FMT_F32>(base, pix, 16, c, v);
}

// four consecutive channels (c0 a multiple of 4) of one pixel in the stem's output format
__device__ __forceinline__ void stem_store4(const StemArgs& a, void* base, size_t pix, int c0, const float (&v)[4]) {
  if (a.out_fmt == FMT_SPLIT)
    store4<FMT_SPLIT>(base, pix, 16, c0, v);
  else if (a.out_fmt == FMT_F16) {
    const float z[4] = {0.f, 0.f, 0.f, 0.f};
    store4<FMT_F16>(base, pix, 32, c0, v);
    store4<FMT_F16>(base, pix, 32, c0 + 16, z);
  } else
    store4<FMT_F32>(base, pix, 16, c0, v);
}

template <int CIN, int WINO>  // WINO: 0 direct second conv, 1 Winograd F(2,3) along x, 2 Winograd F(2x2,3x3)
// (five workgroups per CU -- what the 29.5 KiB of LDS allow -- for the gray-image F(2x2,3x3) instantiation of the headline: 96 registers with nine spills, measured 1.095 -> 1.064 ms on cfg3's
// 32 frames; the kernel is a chain of cold loads and two barriers per 8 x 32-pixel tile, i.e. bound by latency x occupancy rather than by its vector instructions)
__global__ __launch_bounds__(256, (CIN == 1 && WINO == 2) ? 5 : 4) void stem_fused_kernel(StemArgs a) {
  constexpr int IMG_W = TW + 4, IMG_H = TH + 4;   // image patch incl. both halos
  __shared__ __attribute__((aligned(16))) float sA[HALO_H * HALO_W * LROW];   // conv0 output (conv1 input halo)
  __shared__ float sImg[CIN * IMG_H * IMG_W];
  __shared__ float sW0[9 * CIN * 16 + 16];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  // Workgroup ids are dealt round-robin over the 8 XCDs; give every XCD a CONTIGUOUS range of tiles so that neighbouring tiles --
  // which share image rows (a 36-byte row segment costs a whole 128-B line) -- meet in one L2 (round 1: 193 MB fetched for 32 MB of frames)
  const int per_xcd = (tiles_x * tiles_y * a.B + 7) >> 3;
  int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (t >= tiles_x * tiles_y * a.B) return;  // workgroup-uniform
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;

  // ---- stage: image patch (normalised, zero outside), conv0 weights+bias, conv1 weights
  for (int i = tid; i < CIN * IMG_H * IMG_W; i += 256) {
    const int c = i / (IMG_H * IMG_W), r = i - c * (IMG_H * IMG_W);
    const int iy = r / IMG_W, ix = r - iy * IMG_W;
    const int gy = y0 + iy - 2, gx = x0 + ix - 2;
    float v = 0.f;
    if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
      const size_t o = (((size_t)b * CIN + c) * a.H + gy) * a.W + gx;
      if (a.dtype == 0)
        v = (float)reinterpret_cast<const uint8_t*>(a.src)[o] / 255.0f;
      else {
        v = reinterpret_cast<const float*>(a.src)[o];
        if (a.dtype == 2) v = v / 255.0f;
      }
    }
    sImg[i] = v;
  }
  for (int i = tid; i < 9 * CIN * 16; i += 256) sW0[i] = a.w0[i];
  if (tid < 16) sW0[9 * CIN * 16 + tid] = a.b0[tid];
  __syncthreads();

  // ---- conv0 + ReLU into the A tile; item = (halo pixel, 4-channel group q = tid&3)
  {
    const int q4 = (tid & 3) * 4;
    // a thread's four output channels are fixed: for a gray image its 9 weight quads live in registers
    // (the inner loop is then one LDS broadcast read + 4 FMAs per tap instead of two LDS reads)
    f32x4 wreg[CIN == 1 ? 9 : 1];
    if (CIN == 1) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) wreg[tap] = *reinterpret_cast<const f32x4*>(sW0 + tap * 16 + q4);
    }
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(sW0 + 9 * CIN * 16 + q4);
    // (a tile whose whole halo lies inside the image -- nearly all of them -- needs no per-pixel test: the loop is specialised on that, workgroup-uniform)
    const bool halo_inside = y0 >= 1 && y0 + TH + 1 <= a.H && x0 >= 1 && x0 + TW + 1 <= a.W;
    auto conv0 = [&](auto inside_tag) __attribute__((always_inline)) {
    constexpr bool INSIDE = decltype(inside_tag)::value;
    for (int pix = tid >> 2; pix < HALO_H * HALO_W; pix += 64) {
      const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
      const int gy = y0 + hy - 1, gx = x0 + hx - 1;
      const bool in = INSIDE || (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W);  // outside the image = conv1's zero padding
      f32x4 acc = b0;
#pragma unroll
      for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float v = sImg[(c * IMG_H + hy + ky) * IMG_W + hx + kx];
            const f32x4 w = CIN == 1 ? wreg[ky * 3 + kx] : *reinterpret_cast<const f32x4*>(sW0 + ((ky * 3 + kx) * CIN + c) * 16 + q4);
            acc += v * w;
          }
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = in ? fmaxf(acc[k], 0.f) : 0.f;
      *reinterpret_cast<f32x4*>(sA + pix * LROW + q4) = acc;
    }
    };
    if (halo_inside)
      conv0(std::true_type{});
    else
      conv0(std::false_type{});
  }
  __syncthreads();

  // ---- conv1 on the matrix cores
  const int li = lane & 15, lg = lane >> 4;
  if constexpr (WINO == 3) {
    // Plain-fp16 precision (the reference's autocast mode, output format FMT_F16): the second conv on v_mfma_f32_16x16x16_f16 -- all 16
    // input channels of a tap are ONE MFMA per 16 pixels (36 MFMAs per wave instead of 64 fp32 ones plus their Winograd transforms).
    // The first conv's fp32 output is rounded to fp16 as it is read (autocast's conv-to-conv tensor is fp16 too); fp32 accumulation.
    // Transposed product: D[channel 4 lg + e][pixel li] per (row m, half h of the 32 columns).
    f16x4 wa[9];  // A[i = li (output channel)][k = 4 lg .. 4 lg + 3 (input channels)] per tap
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(a.w1 + (size_t)(tap * 16 + li) * 16 + lg * 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) wa[tap][k] = (_Float16)w[k];
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int h = 0; h < 2; ++h) acc[m][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(sA + ((2 * wave + m + ky) * HALO_W + h * 16 + li + kx) * LROW + lg * 4);
          f16x4 xb;
#pragma unroll
          for (int k = 0; k < 4; ++k) xb[k] = (_Float16)x[k];
          acc[m][h] = __builtin_amdgcn_mfma_f32_16x16x16f16(wa[tap], xb, acc[m][h], 0, 0, 0);
        }
    }
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(a.b1 + 4 * lg);
    const int Hp = (a.H + 1) / 2, Wp = (a.W + 1) / 2;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int x = x0 + h * 16 + li;
      float pooled[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) pooled[r] = 0.f;  // values are >= 0 after the ReLU; out-of-image elements count as the reference's zero pad
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int y = y0 + 2 * wave + m;
        const bool in = (y < a.H) && (x < a.W);
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = fmaxf(acc[m][h][r] + bias4[r], 0.f);
        if (in && a.dst_full) stem_store4(a, a.dst_full, ((size_t)b * a.H + y) * a.W + x, 4 * lg, o);
#pragma unroll
        for (int r = 0; r < 4; ++r) pooled[r] = fmaxf(pooled[r], in ? o[r] : 0.f);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {  // the x neighbour of the pool window sits in lane li ^ 1
        const int pi = __builtin_bit_cast(int, pooled[r]);
        pooled[r] = fmaxf(pooled[r], __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(pi, pi, 0xB1, 0xF, 0xF, false)));
      }
      const int py = (y0 >> 1) + wave, px = (x0 >> 1) + h * 8 + (li >> 1);
      if (!(li & 1) && py < Hp && px < Wp) stem_store4(a, a.dst_pool, ((size_t)b * Hp + py) * Wp + px, 4 * lg, pooled);
    }
    return;
  }
  if constexpr (WINO == 2) {
    // F(2x2,3x3), wave-private (as conv3x3_w16_kernel): the wave's two image rows are ONE row of 16 Winograd tiles (tile li = output
    // columns 2 li, 2 li + 1), all sixteen positions in the wave (16 accumulators of 4 registers), 64 MFMAs instead of the 96 of the
    // 1-D form; transposed product (weights = A operand): lane (li, lg) ends with channels 4 lg .. 4 lg + 3 of its tile's four pixels,
    // so bias, ReLU, the stores and the 2x2 max pool are lane-local.
    f32x4 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m1 = -1.f;
    asm volatile("" : "+v"(m1));  // opaque: keeps x - y as fma(y, m1, x), which packs two lanes per instruction (v_pk_fma_f32; a plain subtraction stays four v_sub_f32)
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      constexpr int RA[4] = {0, 1, 2, 1}, RB[4] = {2, 2, 1, 3};
      f32x4 t[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 da = *reinterpret_cast<const f32x4*>(sA + ((2 * wave + RA[xi]) * HALO_W + 2 * li + c) * LROW + lg * 4);
        const f32x4 db = *reinterpret_cast<const f32x4*>(sA + ((2 * wave + RB[xi]) * HALO_W + 2 * li + c) * LROW + lg * 4);
        t[c] = xi == 1 ? da + db : db * m1 + da;
      }
      const f32x4 av[4] = {t[2] * m1 + t[0], t[1] + t[2], t[1] * m1 + t[2], t[3] * m1 + t[1]};
      f32x4 bw[4];  // A[i = li (output channel)][k = lg], element j = input channel 4 lg + j
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) bw[nu] = *reinterpret_cast<const f32x4*>(a.w1w2 + (size_t)((xi * 4 + nu) * 16 + li) * 16 + lg * 4);
      // the four positions of a row take turns: an accumulator is touched every fourth MFMA (a dependent v_mfma_f32_16x16x4_f32 pair is 40 cycles apart, an independent one 32)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) acc[xi * 4 + nu] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[nu][j], av[nu][j], acc[xi * 4 + nu], 0, 0, 0);
    }
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(a.b1 + 4 * lg);
    const int Hp = (a.H + 1) / 2, Wp = (a.W + 1) / 2;
    const int x = x0 + 2 * li, y = y0 + 2 * wave;
    // A^T M A on register quads (four channels at a time; the subtractions as fma(y, -1, x), exact, so that they pack)
    float o[2][2][4];
    {
      f32x4 P[4][2];
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        P[xi][0] = (acc[xi * 4 + 0] + acc[xi * 4 + 1]) + acc[xi * 4 + 2];
        P[xi][1] = acc[xi * 4 + 3] * m1 + (acc[xi * 4 + 2] * m1 + acc[xi * 4 + 1]);
      }
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const f32x4 o0 = ((P[0][bb] + P[1][bb]) + P[2][bb]) + bias4;
        const f32x4 o1 = (P[3][bb] * m1 + (P[2][bb] * m1 + P[1][bb])) + bias4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[0][bb][r] = fmaxf(o0[r], 0.f);
          o[1][bb][r] = fmaxf(o1[r], 0.f);
        }
      }
    }
    float pooled[4] = {0.f, 0.f, 0.f, 0.f};  // values are >= 0 after the ReLU; out-of-image elements count as the reference's zero pad
#pragma unroll
    for (int aa = 0; aa < 2; ++aa)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const bool in = (y + aa < a.H) && (x + bb < a.W);
        if (in && a.dst_full) stem_store4(a, a.dst_full, ((size_t)b * a.H + y + aa) * a.W + x + bb, 4 * lg, o[aa][bb]);
#pragma unroll
        for (int r = 0; r < 4; ++r) pooled[r] = fmaxf(pooled[r], in ? o[aa][bb][r] : 0.f);
      }
    const int py = (y0 >> 1) + wave, px = (x0 >> 1) + li;
    if (py < Hp && px < Wp) stem_store4(a, a.dst_pool, ((size_t)b * Hp + py) * Wp + px, 4 * lg, pooled);
    return;
  }
  if constexpr (WINO == 1) {
    // The product is accumulated TRANSPOSED (the transformed weights are the A operand, the pixel pairs the B operand): D row
    // 4 lg + r = output channel, column li = output pair t (pixels 2t, 2t + 1), so a lane owns four consecutive channels of
    // its pair -- one 16-byte (fp32) or 8-byte (fp16 formats) store per pixel, and the 2x2 max pool is in-lane arithmetic.
    f32x4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) acc[m][xi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      f32x4 bw[4];  // transformed weights of this kernel row: A[i = li (output channel)][k = lg], element j = input channel 4 lg + j
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) bw[xi] = *reinterpret_cast<const f32x4*>(a.w1w + (size_t)((ky * 4 + xi) * 16 + li) * 16 + lg * 4);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        f32x4 d[4];  // the pair's four input pixels (halo columns 2t .. 2t+3), channels 4 lg .. 4 lg + 3
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = *reinterpret_cast<const f32x4*>(sA + ((2 * wave + m + ky) * HALO_W + 2 * li + c) * LROW + lg * 4);
        const f32x4 af[4] = {d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int xi = 0; xi < 4; ++xi) acc[m][xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[xi][j], af[xi][j], acc[m][xi], 0, 0, 0);
      }
    }
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(a.b1 + 4 * lg);
    const int Hp = (a.H + 1) / 2, Wp = (a.W + 1) / 2;
    const int x = x0 + 2 * li;
    float pooled[4] = {0.f, 0.f, 0.f, 0.f};  // values are >= 0 after the ReLU; out-of-image elements count as the reference's zero pad
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int y = y0 + 2 * wave + m;
      float oa[4], ob[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float m1 = acc[m][1][r], m2 = acc[m][2][r];
        oa[r] = fmaxf((acc[m][0][r] + m1) + m2 + bias4[r], 0.f);
        ob[r] = fmaxf((m1 - m2) - acc[m][3][r] + bias4[r], 0.f);
      }
      const bool ina = (y < a.H) && (x < a.W), inb = (y < a.H) && (x + 1 < a.W);
      if (a.dst_full) {
        const size_t pix = ((size_t)b * a.H + y) * a.W + x;
        if (ina) stem_store4(a, a.dst_full, pix, 4 * lg, oa);
        if (inb) stem_store4(a, a.dst_full, pix + 1, 4 * lg, ob);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) pooled[r] = fmaxf(pooled[r], fmaxf(ina ? oa[r] : 0.f, inb ? ob[r] : 0.f));
    }
    const int py = (y0 >> 1) + wave, px = (x0 >> 1) + li;
    if (py < Hp && px < Wp) stem_store4(a, a.dst_pool, ((size_t)b * Hp + py) * Wp + px, 4 * lg, pooled);
    return;
  }
  // conv1's B fragments are the same for every tile: nine 16-B loads per lane straight into registers
  // (no LDS copy of the weights: 11.5 KiB less LDS per workgroup, five workgroups per CU instead of three)
  f32x4 bw[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) bw[tap] = *reinterpret_cast<const f32x4*>(a.w1 + (size_t)(tap * 16 + li) * 16 + lg * 4);
  f32x4 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int h = 0; h < 2; ++h) acc[m][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const f32x4 bf = bw[tap];
    f32x4 af[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        af[m][h] = *reinterpret_cast<const f32x4*>(sA + ((2 * wave + m + ky) * HALO_W + h * 16 + li + kx) * LROW + lg * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[m][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][h][j], bf[j], acc[m][h], 0, 0, 0);
  }

  // ---- epilogue: bias + ReLU, optional full-res store, 2x2 max pool (out-of-image = 0, the
  // reference's zero pad; values are >= 0 after ReLU so max() is unaffected)
  const float bias = a.b1[li];
  const int Hp = (a.H + 1) / 2, Wp = (a.W + 1) / 2;
  float v[2][2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int y = y0 + 2 * wave + m;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int x = x0 + h * 16 + 4 * lg + r;
        float o = fmaxf(acc[m][h][r] + bias, 0.f);
        const bool in = (y < a.H) && (x < a.W);
        if (in && a.dst_full) stem_store(a, a.dst_full, ((size_t)b * a.H + y) * a.W + x, li, o);
        v[m][h][r] = in ? o : 0.f;
      }
  }
  const int py = (y0 >> 1) + wave;
  if (py < Hp) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int px = (x0 >> 1) + h * 8 + 2 * lg + pr;
        if (px < Wp) {
          const float o = fmaxf(fmaxf(v[0][h][2 * pr], v[0][h][2 * pr + 1]), fmaxf(v[1][h][2 * pr], v[1][h][2 * pr + 1]));
          stem_store(a, a.dst_pool, ((size_t)b * Hp + py) * Wp + px, li, o);
        }
      }
  }
}

int launch_stem(const StemArgs& a, hipStream_t s) {
  const int tiles = ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH) * a.B;
  const int grid = 8 * ((tiles + 7) / 8);  // the kernel deals tiles to XCDs in contiguous ranges
  const int wino = (a.out_fmt == FMT_F16 && a.wino >= 2) ? 3 : (a.wino >= 2 && a.w1w2) ? 2 : ((a.wino && a.w1w) ? 1 : 0);
  if (wino == 3 && a.f16_mfma && (a.cin == 1 || a.cin == 3)) return launch_stem_f16(a, s);
  if (a.cin == 1 && wino == 3)
    hipLaunchKernelGGL((stem_fused_kernel<1, 3>), dim3(grid), dim3(256), 0, s, a);
  else if (a.cin == 3 && wino == 3)
    hipLaunchKernelGGL((stem_fused_kernel<3, 3>), dim3(grid), dim3(256), 0, s, a);
  else if (a.cin == 1 && wino == 2)
    hipLaunchKernelGGL((stem_fused_kernel<1, 2>), dim3(grid), dim3(256), 0, s, a);
  else if (a.cin == 1 && wino == 1)
    hipLaunchKernelGGL((stem_fused_kernel<1, 1>), dim3(grid), dim3(256), 0, s, a);
  else if (a.cin == 1)
    hipLaunchKernelGGL((stem_fused_kernel<1, 0>), dim3(grid), dim3(256), 0, s, a);
  else if (a.cin == 3 && wino == 2)
    hipLaunchKernelGGL((stem_fused_kernel<3, 2>), dim3(grid), dim3(256), 0, s, a);
  else if (a.cin == 3 && wino == 1)
    hipLaunchKernelGGL((stem_fused_kernel<3, 1>), dim3(grid), dim3(256), 0, s, a);
  else if (a.cin == 3)
    hipLaunchKernelGGL((stem_fused_kernel<3, 0>), dim3(grid), dim3(256), 0, s, a);
  else {
    set_error("fused stem supports 1 or 3 input channels, got %d", a.cin);
    return PH_E_INVALID;
  }
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// K0: first convolution straight from the NCHW image (uint8 or float), normalisation fused.
// One thread = one output pixel x 4 output channels (16-B store; a wave writes 1 KiB
// contiguous when Cp == 16).  Cin is 1 or 3 so K = 9..27: VALU work, HBM-bound.
// Weights: [tap][ci][Cp] fp32.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void input_conv3x3_kernel(InputConvArgs a) {
  const int groups = a.coutp >> 2;
  const size_t total = (size_t)a.B * a.H * a.W * groups;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int gq = (int)(idx % groups);
    size_t p = idx / groups;
    const int x = (int)(p % a.W);
    p /= a.W;
    const int y = (int)(p % a.H);
    const int b = (int)(p / a.H);
    f32x4 acc = *reinterpret_cast<const f32x4*>(a.bias + gq * 4);
    const int k = a.ksize, kh = a.ksize >> 1;
    for (int ci = 0; ci < a.cin; ++ci) {
      const size_t plane = ((size_t)b * a.cin + ci) * a.H * a.W;
      for (int ky = 0; ky < k; ++ky) {
        const int yy = y + ky - kh;
        if (yy < 0 || yy >= a.H) continue;
        for (int kx = 0; kx < k; ++kx) {
          const int xx = x + kx - kh;
          if (xx < 0 || xx >= a.W) continue;
          float v;
          if (a.dtype == 0)
            v = (float)reinterpret_cast<const uint8_t*>(a.src)[plane + (size_t)yy * a.W + xx] / 255.0f;
          else {
            v = reinterpret_cast<const float*>(a.src)[plane + (size_t)yy * a.W + xx];
            if (a.dtype == 2) v = v / 255.0f;
          }
          const f32x4 w = *reinterpret_cast<const f32x4*>(a.w + ((size_t)(ky * k + kx) * a.cin + ci) * a.coutp + gq * 4);
          acc += v * w;
        }
      }
    }
    if (a.relu) {
      acc[0] = fmaxf(acc[0], 0.f);
      acc[1] = fmaxf(acc[1], 0.f);
      acc[2] = fmaxf(acc[2], 0.f);
      acc[3] = fmaxf(acc[3], 0.f);
    }
    const size_t pix = ((size_t)b * a.H + y) * a.W + x;
    if (a.out_fmt == FMT_F32)
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.dst) + pix * a.coutp + gq * 4) = acc;
    else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (a.out_fmt == FMT_SPLIT)
          store1<FMT_SPLIT>(a.dst, pix, a.coutp, gq * 4 + k, acc[k]);
        else
          store1<FMT_F16>(a.dst, pix, a.dst_cp, gq * 4 + k, acc[k]);
      }
      if (a.out_fmt == FMT_F16 && a.dst_cp > a.coutp && gq < (a.dst_cp - a.coutp) / 4) {  // zero the pad channels [coutp, dst_cp)
#pragma unroll
        for (int k = 0; k < 4; ++k) store1<FMT_F16>(a.dst, pix, a.dst_cp, a.coutp + gq * 4 + k, 0.f);
      }
    }
  }
}

// The common first conv (3x3, 1 or 3 input channels, 16 output channels, fp32 slots) without the generic kernel's per-tap global
// loads, divisions and branches: an 8 x 32 pixel tile's normalised image patch goes to LDS once, a thread keeps the 9 x CIN weight
// quads of ITS four output channels in registers (item = (pixel, channel quad): four lanes share a pixel -- LDS broadcast reads --
// and a wave's stores are one contiguous KiB).  The training program's forward runs this (the inference plans fuse the first block).
template <int CIN>
__global__ __launch_bounds__(256) void input_conv3_c16_kernel(InputConvArgs a) {
  constexpr int PW = TW + 2, PH_ = TH + 2;
  __shared__ float sImg[CIN * PH_ * PW];
  const int tid = threadIdx.x;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  const int q4 = (tid & 3) * 4;
  f32x4 wreg[9 * CIN];
#pragma unroll
  for (int t = 0; t < 9 * CIN; ++t) wreg[t] = *reinterpret_cast<const f32x4*>(a.w + t * 16 + q4);  // [tap][cin][16]
  const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + q4);
  for (int tile = blockIdx.x; tile < tiles_x * tiles_y * a.B; tile += gridDim.x) {
    int t = tile;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    __syncthreads();  // the previous tile's reads are done
    for (int i = tid; i < CIN * PH_ * PW; i += 256) {
      const int c = i / (PH_ * PW), r = i - c * (PH_ * PW);
      const int iy = r / PW, ix = r - iy * PW;
      const int gy = y0 + iy - 1, gx = x0 + ix - 1;
      float v = 0.f;
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
        const size_t o = (((size_t)b * CIN + c) * a.H + gy) * a.W + gx;
        if (a.dtype == 0)
          v = (float)reinterpret_cast<const uint8_t*>(a.src)[o] / 255.0f;
        else {
          v = reinterpret_cast<const float*>(a.src)[o];
          if (a.dtype == 2) v = v / 255.0f;
        }
      }
      sImg[i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TH * TW / 64; ++j) {
      const int pix = (tid >> 2) + 64 * j;
      const int py = pix / TW, px = pix - py * TW;
      f32x4 acc = b0;
#pragma unroll
      for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) acc += sImg[(c * PH_ + py + ky) * PW + px + kx] * wreg[(ky * 3 + kx) * CIN + c];
      if (a.relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = fmaxf(acc[k], 0.f);
      }
      const int y = y0 + py, x = x0 + px;
      if (y < a.H && x < a.W) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.dst) + (((size_t)b * a.H + y) * a.W + x) * 16 + q4) = acc;
    }
  }
}

int launch_input_conv(const InputConvArgs& a, hipStream_t s) {
  if (a.ksize == 3 && a.coutp == 16 && a.out_fmt == FMT_F32 && (a.cin == 1 || a.cin == 3)) {
    const int tiles = ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH) * a.B;
    const dim3 grid(std::min(tiles, 256 * 8));
    if (a.cin == 1)
      hipLaunchKernelGGL(input_conv3_c16_kernel<1>, grid, dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL(input_conv3_c16_kernel<3>, grid, dim3(256), 0, s, a);
    PH_HIP_CHECK(hipGetLastError());
    return PH_OK;
  }
  const size_t total = (size_t)a.B * a.H * a.W * (a.coutp / 4);
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(input_conv3x3_kernel, dim3(blocks), dim3(256), 0, s, a);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// K3: 2x2 stride-2 max pool, NHWC, float4 per thread.  Odd sizes: the missing right/bottom
// element is the reference's zero pad (common.py:93-96), i.e. max(v, 0).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool2x2_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H, int W, int cp) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, groups = cp >> 2;
  const size_t total = (size_t)B * Ho * Wo * groups;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int gq = (int)(idx % groups);
    size_t p = idx / groups;
    const int x = (int)(p % Wo);
    p /= Wo;
    const int y = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const int y1 = 2 * y + 1, x1 = 2 * x + 1;
    const float* base = src + ((size_t)b * H * W) * cp + gq * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(base + ((size_t)(2 * y) * W + 2 * x) * cp);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 v01 = (x1 < W) ? *reinterpret_cast<const f32x4*>(base + ((size_t)(2 * y) * W + x1) * cp) : z;
    f32x4 v10 = (y1 < H) ? *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * W + 2 * x) * cp) : z;
    f32x4 v11 = (y1 < H && x1 < W) ? *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * W + x1) * cp) : z;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = fmaxf(fmaxf(v[k], v01[k]), fmaxf(v10[k], v11[k]));
    *reinterpret_cast<f32x4*>(dst + (((size_t)b * Ho + y) * Wo + x) * cp + gq * 4) = v;
  }
}

int launch_pool(const float* src, float* dst, int B, int H, int W, int cp, hipStream_t s) {
  const size_t total = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (cp / 4);
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(pool2x2_kernel, dim3(blocks), dim3(256), 0, s, src, dst, B, H, W, cp);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// K4: bilinear x2 upsample, align_corners=False (ATen upsample_bilinear2d semantics:
// src = max(0, (dst + 0.5) * 0.5 - 0.5), i1 = min(i0 + 1, in - 1), lambda = src - i0).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H, int W, int cp) {
  // One thread = 4 channels x the 2x2 output pixels (2i+1..2i+2, 2j+1..2j+2) that lie between input pixels
  // (i..i+1, j..j+1): four 16-B loads feed four 16-B stores (one load set per output pixel cost 4x the loads).
  // i, j run from -1 so that output row / column 0 is produced too; every output evaluates the reference's own
  // source-coordinate formula (half-pixel centres, clamped at 0), so the border weights are exactly its weights.
  const int Ho = 2 * H, Wo = 2 * W, groups = cp >> 2;
  const int nI = H + 1, nJ = W + 1;
  const size_t total = (size_t)B * nI * nJ * groups;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int gq = (int)(idx % groups);
    size_t p = idx / groups;
    const int j = (int)(p % nJ) - 1;
    p /= nJ;
    const int i = (int)(p % nI) - 1;
    const int b = (int)(p / nI);
    const int r0 = max(i, 0), r1 = min(i + 1, H - 1), c0 = max(j, 0), c1 = min(j + 1, W - 1);
    const float* base = src + ((size_t)b * H * W) * cp + gq * 4;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((size_t)r0 * W + c0) * cp);
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((size_t)r0 * W + c1) * cp);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((size_t)r1 * W + c0) * cp);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((size_t)r1 * W + c1) * cp);
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
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = hy * (hx * v00[k] + lx * v01[k]) + ly * (hx * v10[k] + lx * v11[k]);
        *reinterpret_cast<f32x4*>(dst + (((size_t)b * Ho + y) * Wo + x) * cp + gq * 4) = o;
      }
    }
  }
}

int launch_upsample(const float* src, float* dst, int B, int H, int W, int cp, hipStream_t s) {
  const size_t total = (size_t)B * (H + 1) * (W + 1) * (cp / 4);
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(upsample2x_kernel, dim3(blocks), dim3(256), 0, s, src, dst, B, H, W, cp);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// K5a: zero-stuff for the transposed convolution: Z[2i,2j] = X[i,j], zeros elsewhere
// (size 2H x 2W: the trailing zero row/col is output_padding=1).  The transposed conv is
// then the 3x3 "same" MFMA conv of Z with the flipped, in/out-transposed kernel.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void zero_stuff_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H, int W, int cp) {
  const int Ho = 2 * H, Wo = 2 * W, groups = cp >> 2;
  const size_t total = (size_t)B * Ho * Wo * groups;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int gq = (int)(idx % groups);
    size_t p = idx / groups;
    const int x = (int)(p % Wo);
    p /= Wo;
    const int y = (int)(p % Ho);
    const int b = (int)(p / Ho);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!(x & 1) && !(y & 1)) v = *reinterpret_cast<const f32x4*>(src + (((size_t)b * H + (y >> 1)) * W + (x >> 1)) * cp + gq * 4);
    *reinterpret_cast<f32x4*>(dst + idx * 4) = v;
  }
}

int launch_zero_stuff(const float* src, float* dst, int B, int H, int W, int cp, hipStream_t s) {
  const size_t total = (size_t)B * 4 * H * W * (cp / 4);
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(zero_stuff_kernel, dim3(blocks), dim3(256), 0, s, src, dst, B, H, W, cp);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// Global max pool over H x W per channel (nn.AdaptiveMaxPool2d(1)): one workgroup per (image, 64 channels).
__global__ __launch_bounds__(256) void global_maxpool_kernel(const float* __restrict__ src, float* __restrict__ dst, int HW, int cp) {
  __shared__ float red[4][64];
  const int b = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
  float m = -INFINITY;
  if (c < cp)
    for (int p = r; p < HW; p += 4) m = fmaxf(m, src[((size_t)b * HW + p) * cp + c]);
  red[r][threadIdx.x & 63] = m;
  __syncthreads();
  if (r == 0 && c < cp) dst[(size_t)b * cp + c] = fmaxf(fmaxf(red[0][c & 63], red[1][c & 63]), fmaxf(red[2][c & 63], red[3][c & 63]));
}
int launch_global_maxpool(const float* src, float* dst, int B, int HW, int cp, hipStream_t s) {
  hipLaunchKernelGGL(global_maxpool_kernel, dim3(B, (cp + 63) / 64), dim3(256), 0, s, src, dst, HW, cp);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// Row softmax in place (nn.Softmax(dim=-1) of the class-vector head): one thread per row, max-subtracted.
__global__ void softmax_rows_kernel(float* __restrict__ x, int rows, int n) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float* p = x + (size_t)r * n;
  float m = -INFINITY;
  for (int i = 0; i < n; ++i) m = fmaxf(m, p[i]);
  float s = 0.f;
  for (int i = 0; i < n; ++i) {
    const float e = expf(p[i] - m);
    p[i] = e;
    s += e;
  }
  for (int i = 0; i < n; ++i) p[i] = p[i] / s;
}
int launch_softmax_rows(float* x, int rows, int n, hipStream_t s) {
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 63) / 64), dim3(64), 0, s, x, rows, n);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// NHWC(Cp) -> NCHW(C) copy for the debug/parity read-back of intermediate slots.
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int HW, int cp, int c) {
  const size_t total = (size_t)B * c * HW;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t hw = idx % HW;
    size_t r = idx / HW;
    const int ch = (int)(r % c);
    const size_t b = r / c;
    dst[idx] = src[(b * HW + hw) * cp + ch];
  }
}

int launch_nhwc_to_nchw(const float* src, float* dst, int B, int HW, int cp, int c, hipStream_t s) {
  const size_t total = (size_t)B * c * HW;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(blocks), dim3(256), 0, s, src, dst, B, HW, cp, c);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

int conv_lds_bytes(int bn) { return (HALO_H * HALO_W + 9 * bn) * LROW * (int)sizeof(float); }

int prepare_kernels() {
  // 73 KiB of dynamic LDS for the BN=64 variant exceeds the 64 KiB default cap.
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_mfma_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) {
    set_error("hipFuncSetAttribute(conv64) failed: %s", hipGetErrorString(e));
    return PH_E_HIP;
  }
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_mfma_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, conv_lds_bytes(32));
  if (e != hipSuccess) {
    set_error("hipFuncSetAttribute(conv32) failed: %s", hipGetErrorString(e));
    return PH_E_HIP;
  }
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_mfma_dma_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_mfma_dma_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_mfma_dma_persist_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_mfma_dma_persist_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wino_persist_kernel<64, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wino_persist_kernel<32, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) {
    set_error("hipFuncSetAttribute(conv dma) failed: %s", hipGetErrorString(e));
    return PH_E_HIP;
  }
  const int rc2 = prepare_wino2d_kernels();
  if (rc2 != PH_OK) return rc2;
  const int rc3 = prepare_wino4_kernels();
  return rc3 != PH_OK ? rc3 : prepare_w16_kernels();
}

}  // namespace ph
