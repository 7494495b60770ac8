// 3x3 "same" convolution as Winograd F(4x4, 3x3) on the fp32 matrix cores (gfx950): the N-tile-64 layers from 64 input channels on.
//
// Reference semantics: nn.Conv2d(k3, padding "same") + bias + ReLU of SimpleConvBlock (architectures/encoder_decoder.py:108-121,
// 494-510) over torch.concat((skip, x)) (encoder_decoder.py:545,556) where x may be nn.Upsample(x2, bilinear, align_corners=False)
// of a half-resolution tensor (encoder_decoder.py:431-435): the up-sampling is folded into the input transform, the
// up-sampled tensor never exists.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A     per 4x4 output tile, 6x6 input patch d, 3x3 filter g
//
// 36 multiplications per sixteen outputs instead of 144: the matrix cores execute 1/4 of the direct convolution's FLOPs
// (F(2x2,3x3): 4/9).  fp32 throughout; the transforms have coefficients up to 8 (A), 5 (B) and 1/24 (G): measured against the fp64
// convolution the result is ~5x further off than F(2x2,3x3) (1e-5 relative max at K = 768), inside the heads' 1e-4 bar;
// which layers run here is decided per layer by wino4_fits' time model (a tile costs ~11 us outside its K loop).
//
//   * GEMM view: per Winograd position p = xi * 6 + nu one GEMM with M = 4x4-pixel tiles, N = output channels, K = input channels.  A workgroup
//     tile is 8 x 4 Winograd tiles (32 x 16 output pixels) x 64 channels; all 36 positions x (32 tiles x 64 channels) accumulators (288 KiB)
//     live in the registers of EIGHT waves, two per SIMD: wave (nt, pg) owns positions 9 pg .. 9 pg + 8 of the 32-channel N half nt: nine 32x32
//     accumulators (144 registers), 18 MFMAs per quarter.  The product is accumulated transposed (weights = A operand): a lane is a tile, a
//     register quad four consecutive channels.  SIMD s holds waves (0, s) and (1, s), which read the same V fragments.
//   * K runs in QUARTERS (4 input channels = two K steps of v_mfma_f32_32x32x2_f32).  Per quarter the workgroup transforms the raw 34 x 18 halo
//     of the NEXT quarter into V[36 positions][32 tiles][4 channels] in LDS while it multiplies the current one.  The fp32 MFMA runs at the
//     vector ALUs' rate and vector instructions do not overlap with it (measured, DESIGN.md 4.1c: MFMA cycles + VALU cycles add), so the design
//     rule is FEW vector instructions, and every other kind -- LDS, global memory, scalar -- written between the MFMAs of one wave's stream:
//       - weights come straight from L2 into registers, one quarter ahead (five buffer loads per wave and quarter, each into the registers the
//         MFMAs before it just read): no LDS ring, no LDS traffic, no LDS-DMA issue cost (~150 cycles per transfer beside MFMAs);
//       - the halo goes through registers too (buffer load in one quarter, ds_write_b128 in the next): two loads per wave and quarter;
//       - the transform is dealt as (16 tiles) x (xi group): waves 0..3 take an xi pair ({1, 2} or {3, 4}: the pair shares its two row
//         differences), waves 4..7 one of xi 0 / 5, a SIMD one of each; LDS reads are issued one MFMA group ahead of their use;
//       - the MFMAs of positions 6..8 of a quarter are issued at the top of the NEXT quarter, behind the barrier (their operands wait in
//         registers), so that the matrix pipe has work while the reads that follow the barrier are in flight.
//     One s_barrier per quarter.  LDS in the loop: two raw halo slots + two V slots = 60 KiB.
//   * Second source at half resolution (ConvArgs::src1_lowres): its slot holds the 18 x 10 LOW-resolution halo (index-clamped by the
//     loader like the bilinear kernel clamps), the row pass applies B^T U (U = the 6 x 4 bilinear matrix of the patch rows, rows
//     outside the image zeroed: the conv's padding) as four per-lane coefficients, the column pass forms the six up-sampled column
//     samples first: 16 instead of 24 LDS reads per thread.
//   * Epilogue: two rounds through LDS (144 KiB each), every wave with half of its registers per round (channel quads 0, 1, then 2, 3, of both N
//     halves); all eight waves apply A^T . A: one thread per (tile, channel quad, row half) reads the 36 positions and stores two rows of the 4 x 4
//     tile: bias, ReLU, the backward pass' ReLU mask, the fused 2 x 2 max pool (its windows are inside a thread's rows), 16-byte stores, eight lanes =
//     two 64-byte runs of a pixel.  (round 3's form: twelve waves, three per SIMD, weights and halo by LDS-DMA, ran the same arithmetic 6 - 8 % slower.)
#include <type_traits>

#include "common.h"
#include "net_kernels.h"

#ifndef W4_EXP
#define W4_EXP 0  // timing experiments of tools/w4/w4_bench.hip only (results are wrong with any bit set): 1 no weight loads, 2 no input transform, 4 no per-quarter barrier, 8 no halo transfers, 16 no MFMAs, 32 weights always of the first quarter (cache-hot), 64 halo always of the first quarter, 128 transform without its arithmetic (LDS traffic only), 256 weights loaded once per tile, 512 no output stores.  Ablations that zero the operands (1, 8, 16) also raise the clock: compare cycles (-DW4_CLOCK), not milliseconds
#endif

namespace ph {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int W4_TX = 8, W4_TY = 4;             // Winograd tiles per workgroup tile
constexpr int W4_PW = 4 * W4_TX;                // 32 output columns
constexpr int W4_PH = 4 * W4_TY;                // 16 output rows
constexpr int W4_RAW_FLOATS = 12 * 256;         // one raw halo slot: 12 pieces of 64 entries x 16 B (648 entries used: [hy 18][x & 3][x >> 2: 9]; 11 pieces are moved)
constexpr int W4_V_FLOATS = 36 * 128;           // one V slot: [pg 4][four 16-byte fragment slots + one 8-byte slot][lane 64]
constexpr int W4_LDS_FLOATS = 36 * 1024;        // the epilogue's exchange: 36 positions x 32 tiles x 8 channel quads x 4 = 144 KiB (the loop uses the first 60 KiB)
constexpr int W4_Q_FLOATS = 36 * 64 * 4;        // transformed weights of one (N tile, quarter): 36 positions x 64 channels x 4 input channels
constexpr int P4_WAVE_FLOATS = 4 * 256 + 128;   // one wave's fragments of a quarter (nine positions: four position pairs of 16 B per lane + one of 8 B)

// wpack [panel][tap 9][bn 64][16] (pack_conv) -> U = G g G^T in the order the waves load it:
// [n tile][quarter][wave nt * 4 + pg][four 16-byte slots [lane (lh, lx)][ks * 2 + (pl & 1)] of the position pairs pl = (0, 1) .. (6, 7) | one 8-byte slot [lane][ks] of pl = 8]:
// output channel n tile * 64 + nt * 32 + lx, input channel quarter * 4 + 2 lh + ks, position p = xi * 6 + nu = 9 pg + pl
__device__ __forceinline__ float wino4_g_row(int k, float g0, float g1, float g2) {  // row k of G applied to (g0, g1, g2)
  switch (k) {
    case 0: return 0.25f * g0;
    case 1: return (-1.f / 6.f) * ((g0 + g2) + g1);
    case 2: return (-1.f / 6.f) * ((g0 + g2) - g1);
    case 3: return (1.f / 24.f) * g0 + ((1.f / 12.f) * g1 + (1.f / 6.f) * g2);
    case 4: return (1.f / 24.f) * g0 + ((-1.f / 12.f) * g1 + (1.f / 6.f) * g2);
    default: return g2;
  }
}
// One workgroup = one (n tile, quarter) unit of the destination: 64 output channels x 4 input channels x 36 positions = 9,216 floats.  Thread (row, kc)
// loads its nine taps ONCE, forms all 36 positions in registers and drops them into LDS at their place in the unit; the unit then leaves as 36 coalesced
// 1-KiB rows.  (One thread per destination element re-read the nine taps 36 times through 4-byte loads 64 B apart: 0.48 ms for the 2304 -> 768 layer's
// 255 MB, 3.6 ms per ConvNeXt training step, where the weights change every step.)
__global__ __launch_bounds__(256) void wino4_pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int ntiles, int nchunks) {
  __shared__ float unit[W4_Q_FLOATS];
  const int quarter = blockIdx.x % (4 * nchunks), ntile = blockIdx.x / (4 * nchunks);
  const int tid = threadIdx.x;
  const int row = tid >> 2, kcl = tid & 3;
  const int nt = row >> 5, lx = row & 31, lh = kcl >> 1, j = kcl & 1;
  const float* w = src + (((size_t)ntile * nchunks + (quarter >> 2)) * 9 * 64 + row) * 16 + (quarter & 3) * 4 + kcl;
  const size_t ts = (size_t)64 * 16;  // tap stride
  float g[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) g[t] = w[t * ts];
#pragma unroll
  for (int xi = 0; xi < 6; ++xi) {
    float h[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) h[kx] = wino4_g_row(xi, g[0 * 3 + kx], g[1 * 3 + kx], g[2 * 3 + kx]);
#pragma unroll
    for (int nu = 0; nu < 6; ++nu) {
      const float u = wino4_g_row(nu, h[0], h[1], h[2]);
      const int p = xi * 6 + nu, pgp = p / 9, pl = p % 9;
      unit[(nt * 4 + pgp) * P4_WAVE_FLOATS + (pl < 8 ? (((pl >> 1) * 64 + lh * 32 + lx) << 2) + j * 2 + (pl & 1) : 1024 + ((lh * 32 + lx) << 1) + j)] = u;
    }
  }
  __syncthreads();
  float* out = dst + (size_t)blockIdx.x * W4_Q_FLOATS;
#pragma unroll 4
  for (int i = tid; i < W4_Q_FLOATS; i += 256) out[i] = unit[i];
}
int64_t wino4_pack_floats(int ntiles, int nchunks) { return (int64_t)ntiles * 4 * nchunks * W4_Q_FLOATS; }
int launch_wino4_pack(const float* wpack, float* wino, int ntiles, int nchunks, hipStream_t s) {
  hipLaunchKernelGGL(wino4_pack_kernel, dim3((unsigned)(ntiles * 4 * nchunks)), dim3(256), 0, s, wpack, wino, ntiles, nchunks);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

__device__ __forceinline__ void w4_deal_tile(int id, int tiles, int nt_count, int rot, int* tile, int* ntile) {
  // as w2_deal_tile: every XCD walks a contiguous range of pixel tiles, the N tiles of one pixel tile sit on one XCD; rot (the round, when the layer's
  // last N tile is half empty and a round covers whole pixel tiles) sends the cheap half tiles round the workgroups
  if ((tiles & 7) == 0) {
    const int xcd = id & 7, j = id >> 3;
    *ntile = (j % nt_count + rot) % nt_count;
    *tile = xcd * (tiles >> 3) + j / nt_count;
  } else {
    *ntile = (id % nt_count + rot) % nt_count;
    *tile = id / nt_count;
  }
}

// v[nu] = sum_c B^T[nu][c] t[c] (shared sub-expressions: 14 operations)
__device__ __forceinline__ void w4_col_pass(const float (&t)[6], float (&v)[6]) {
  const float a = fmaf(-4.f, t[2], t[4]);
  const float b = fmaf(-4.f, t[1], t[3]);
  const float c = t[4] - t[2];
  const float d0 = t[3] - t[1];
  const float d = d0 + d0;
  v[0] = fmaf(4.f, t[0], fmaf(-5.f, t[2], t[4]));
  v[1] = a + b;
  v[2] = a - b;
  v[3] = c + d;
  v[4] = c - d;
  v[5] = fmaf(4.f, t[1], fmaf(-5.f, t[3], t[5]));
}

template <int P, int N>
__device__ __forceinline__ void p4_vput(float* slot, int vwq, int vwd, const float* v) {  // N transformed values of positions P .. P + N - 1 into a V slot
  if constexpr (N > 0) {
    constexpr int pgc = P / 9, pl = P % 9;
    if constexpr (pl == 8) {
      slot[pgc * P4_WAVE_FLOATS + vwd] = v[0];
      p4_vput<P + 1, N - 1>(slot, vwq, vwd, v + 1);
    } else if constexpr ((pl & 1) == 0 && N >= 2) {
      *reinterpret_cast<float2*>(slot + pgc * P4_WAVE_FLOATS + (pl >> 1) * 256 + vwq) = make_float2(v[0], v[1]);
      p4_vput<P + 2, N - 2>(slot, vwq, vwd, v + 2);
    } else {
      slot[pgc * P4_WAVE_FLOATS + (pl >> 1) * 256 + vwq + (pl & 1)] = v[0];
      p4_vput<P + 1, N - 1>(slot, vwq, vwd, v + 1);
    }
  }
}

template <bool LOWRES, bool KS = false>
__global__ __launch_bounds__(512) void conv3x3_wino4_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int pw = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave 0..7
  // consumer role: positions 9 pg .. 9 pg + 8 of N tile nt
  const int nt = pw >> 2, pg = pw & 3;
  const int lx = lane & 31, lh = lane >> 5;
  // producer role: xi group xg (0: xi 1 and 2, 1: xi 3 and 4, 2: xi 0, 3: xi 5) for the sixteen tiles typ = 2 (pw & 1) + (lane >> 5), txp; channel cp of the quarter
  const int xg = pw < 4 ? (pw >> 1) : 2 + ((pw - 4) >> 1);
  const int typ = 2 * (pw & 1) + (lane >> 5), txp = (lane >> 2) & 7, cp = lane & 3;

  const int tiles_x = (a.W + W4_PW - 1) / W4_PW;
  const int tiles_y = (a.H + W4_PH - 1) / W4_PH;
  const int tiles = tiles_x * tiles_y * a.B;
  const int ntc = (a.coutp + 63) / 64;
  const int units = tiles * ntc;
  const int total = KS ? units * a.ksplit : units;
  const int Q0 = a.c0p / 4, Q1 = a.c1p / 4, Q = Q0 + Q1;
  constexpr int lowres = LOWRES ? 1 : 0;
  const int Hl = a.H >> 1, Wl = a.W >> 1;

#ifdef W4_CLOCK  // diagnostic build: shader cycles and 100-MHz ticks of the whole workgroup into ConvArgs::clock_probe[block * 2 ..]
  const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  float* const rawbuf = lds;                      // two raw halo slots
  float* const vbuf = lds + 2 * W4_RAW_FLOATS;    // two V slots

  // ---- producer constants (t[c] = fma(beta, fma(alpha, d1, d3), fma(alpha, d2, d4)) for xi 1..4, fma(4, P, fma(-5, R, S)) for xi 0 and 5)
  const float alpha = xg == 0 ? -4.f : -1.f, beta = xg == 0 ? 1.f : 2.f, nbeta = -beta;
  const int pbase_f = ((4 * typ + (xg == 2 ? 0 : 1)) * 36 + txp) * 4 + cp;  // first patch row this thread reads: 1 (xi 1..4: rows 1..4; xi 5: rows 1, 3, 5) or 0 (xi 0: rows 0, 2, 4)
  const int pbase_l = ((2 * typ) * 18 + txp) * 4 + cp;
  const int vwq = ((cp >> 1) * 32 + typ * 8 + txp) * 4 + (cp & 1) * 2;  // V write: 16-byte slots [lane (c >> 1, tile)][ks = c & 1][position parity]
  const int vwd = 1024 + ((cp >> 1) * 32 + typ * 8 + txp) * 2 + (cp & 1);
  const int vrq = pg * P4_WAVE_FLOATS + lane * 4, vrd = pg * P4_WAVE_FLOATS + 1024 + lane * 2;

  struct Plan {
    int b, x0, y0, ntile;
  };
  const int G = (int)gridDim.x;
  const bool rotate = (a.coutp & 63) != 0 && (a.coutp & 63) <= 32 && ntc > 1 && (((tiles & 7) == 0) ? ((G & 7) == 0 && (G >> 3) % ntc == 0) : (G % ntc == 0));
  auto setup = [&](int vid, Plan& P) {
    int t, ntile;
    w4_deal_tile(vid, tiles, ntc, rotate ? vid / G : 0, &t, &ntile);
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    P.b = t / tiles_y;
    P.x0 = tx * W4_PW;
    P.y0 = ty * W4_PH;
    P.ntile = ntile;
  };

  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpack_wino4, 0, (int)((unsigned)ntc * (unsigned)Q * (unsigned)(W4_Q_FLOATS * 4)), 0x00020000);
  const unsigned w_lane = (unsigned)(pw * P4_WAVE_FLOATS + lane * 4) * 4u;
  const unsigned w_lane9 = (unsigned)(pw * P4_WAVE_FLOATS + 1024 + lane * 2) * 4u;

  for (int vid = blockIdx.x; vid < total; vid += gridDim.x) {
#ifdef W4_STAMP
    const unsigned long long tile_t0 = __builtin_amdgcn_s_memtime();
#endif
    Plan P;
    const int ksl = KS ? vid / units : 0;
    const int qbeg = KS ? ksl * Q / a.ksplit : 0;
    const int qend = KS ? (ksl + 1) * Q / a.ksplit : Q;
    setup(KS ? vid - ksl * units : vid, P);
    // ---- loader state: this wave moves pieces pw and pw + 8 of a raw slot (pieces 0..10 hold the 648 halo entries; the transfers aimed at piece 11 are out of range = zeros)
    unsigned off0[2], off1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = (pw + 8 * i) * 64 + lane;
      const int hy = e / 36, rem = e - hy * 36;
      const int pl = rem / 9, k = rem - pl * 9;
      const int hx = 4 * k + pl;
      const int gy = P.y0 + hy - 1, gx = P.x0 + hx - 1;
      const bool in = pw + 8 * i < 11 && hy < W4_PH + 2 && hx < W4_PW + 2 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      off0[i] = in ? (unsigned)((P.b * a.H + gy) * a.W + gx) * (unsigned)(a.c0p * 4) : 0xFFFFFF00u;
      if (!lowres) {
        off1[i] = (in && a.c1p > 0) ? (unsigned)((P.b * a.H + gy) * a.W + gx) * (unsigned)(a.c1p * 4) : 0xFFFFFF00u;
      } else {
        const int ly = e / 18, rem2 = e - ly * 18;
        const int par = rem2 / 9, kk = rem2 - par * 9;
        const int lc = 2 * kk + par;
        const int sy = min(max((P.y0 >> 1) - 1 + ly, 0), Hl - 1), sx = min(max((P.x0 >> 1) - 1 + lc, 0), Wl - 1);
        off1[i] = (ly < W4_PH / 2 + 2) ? (unsigned)((P.b * Hl + sy) * Wl + sx) * (unsigned)(a.c1p * 4) : 0xFFFFFF00u;
      }
    }
    const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, (int)((unsigned)(a.B * a.H * a.W) * (unsigned)(a.c0p * 4)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src1 ? a.src1 : a.src0), 0,
                                                                           (int)((unsigned)(lowres ? a.B * Hl * Wl : a.B * a.H * a.W) * (unsigned)(a.c1p * 4)), 0x00020000);
    // The halo goes through registers (buffer load one quarter, ds_write the next): an LDS-DMA transfer costs its wave ~150 cycles of issue beside MFMAs, a register load + ds_write_b128 ~30
    f32x4 hreg[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // (initialised: left undefined, the register allocator spills 144 registers of the folded-bilinear instantiation instead of 20)
    const bool has1 = pw + 8 < 11;  // (wave-uniform) this wave owns a second piece of the full-resolution layout
    auto load_raw = [&](int k) {  // halo of quarter k into this wave's registers
      const bool s1 = k >= Q0;
      const int so = k < qend ? ((W4_EXP & 64) ? 0 : (s1 ? (k - Q0) * 16 : k * 16)) : 0;
      if (W4_EXP & 8) return;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i == 1 && !has1) continue;
        const unsigned off = k < qend ? (s1 ? off1[i] : off0[i]) : 0xFFFFFF00u;
        if (s1)
          hreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc1, off, so, 0));
        else
          hreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc0, off, so, 0));
      }
    };
    auto store_raw = [&](int k) {  // ... and from there into raw slot k & 1
      float* slot = rawbuf + (k & 1) * W4_RAW_FLOATS + lane * 4;
      if (W4_EXP & 8) return;
      *reinterpret_cast<f32x4*>(slot + pw * 256) = hreg[0];
      if (has1) *reinterpret_cast<f32x4*>(slot + (pw + 8) * 256) = hreg[1];
    };
    const int w_tile = P.ntile * Q * (W4_Q_FLOATS * 4);
    f32x4 wf[4];
    float wf9[2];
    // Software pipeline: the MFMAs of positions 6..8 (fragment slots 3 and 4) of quarter q are issued at the top of quarter q + 1, behind the barrier; their operands wait in these
    // registers.  Zeros before a tile's first quarter.
    f32x4 vf3 = {0.f, 0.f, 0.f, 0.f};
    float2 vf9 = make_float2(0.f, 0.f);
    auto load_w = [&](int k, int j) {  // fragment slot j (0..3: 16 bytes, 4: 8 bytes) of quarter k's weights into its registers
      const bool live = k < qend && !(W4_EXP & 1);
      if ((W4_EXP & 256) && k != qbeg) return;
      const int so = live ? w_tile + ((W4_EXP & 32) ? qbeg : k) * (W4_Q_FLOATS * 4) : 0;
      if (j < 4) {
        wf[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, live ? w_lane + (unsigned)j * 1024u : 0xFFFFFF00u, so, 0));
      } else {
        const f32x2 r = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(wrsrc, live ? w_lane9 : 0xFFFFFF00u, so, 0));  // (bit_cast of the whole vector: of one element it reads element 0)
        wf9[0] = r[0];
        wf9[1] = r[1];
      }
    };

    // low-resolution row coefficients of this thread's tile row for its one or two xi and the zero masks of patch columns 0 and 5
    float cl[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float zx0 = 1.f, zx5 = 1.f;
    if (lowres) {
      const float bt[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0}, {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int xi = xg == 0 ? 1 + s : (xg == 1 ? 3 + s : (xg == 2 ? 0 : 5));
        float c4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          const int Y = P.y0 + 4 * typ - 1 + r;
          float btv = 0.f;
#pragma unroll
          for (int x = 0; x < 6; ++x) btv = xi == x ? bt[x][r] : btv;
          if (Y < 0 || Y >= a.H) btv = 0.f;
          const float w0 = (r & 1) ? 0.25f : 0.75f;
          c4[r >> 1] += btv * w0;
          c4[(r >> 1) + 1] += btv * (1.f - w0);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) cl[s][n] = c4[n];
      }
      zx0 = (P.x0 + 4 * txp - 1 < 0) ? 0.f : 1.f;
      zx5 = (P.x0 + 4 * txp + 4 >= a.W) ? 0.f : 1.f;
    }

    // ---- the input transform of quarter k (raw slot k % 3 -> V slot k & 1) in pieces: reads of patch columns [c0, c1), their row pass, then per xi the column pass + stores
    float dd[6][4];   // raw values: [patch column][row]
    float ta[6], tb[6];
    auto t_read = [&](auto xg_tag, auto lr_tag, const float* raw, int c0, int c1) __attribute__((always_inline)) {
      constexpr int XG = decltype(xg_tag)::value;
      constexpr bool LR = decltype(lr_tag)::value;
      if (W4_EXP & 2) return;
      if constexpr (LR) {  // (columns = the four low-resolution columns; rows = the four low-resolution rows)
        const float* p = raw + pbase_l;
#pragma unroll
        for (int n = 0; n < 4; ++n)
          if (n >= c0 && n < c1) {
            const int co = ((n & 1) * 9 + (n >> 1)) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) dd[n][r] = p[co + r * 18 * 4];
          }
      } else {
        const float* p = raw + pbase_f;
#pragma unroll
        for (int c = 0; c < 6; ++c)
          if (c >= c0 && c < c1) {
            const int co = ((c & 3) * 9 + (c >> 2)) * 4;
            if constexpr (XG < 2) {
#pragma unroll
              for (int r = 0; r < 4; ++r) dd[c][r] = p[co + r * 144];
            } else {
#pragma unroll
              for (int r = 0; r < 3; ++r) dd[c][r] = p[co + 2 * r * 144];
            }
          }
      }
    };
    float tla[4], tlb[4];
    auto t_rows = [&](auto xg_tag, auto lr_tag, int c0, int c1) __attribute__((always_inline)) {
      constexpr int XG = decltype(xg_tag)::value;
      constexpr bool LR = decltype(lr_tag)::value;
      if (W4_EXP & 2) return;
      if constexpr (LR) {
#pragma unroll
        for (int n = 0; n < 4; ++n)
          if (n >= c0 && n < c1) {
            tla[n] = fmaf(cl[0][0], dd[n][0], fmaf(cl[0][1], dd[n][1], fmaf(cl[0][2], dd[n][2], cl[0][3] * dd[n][3])));
            asm volatile("" : "+v"(tla[n]));
            if constexpr (XG < 2) {
              tlb[n] = fmaf(cl[1][0], dd[n][0], fmaf(cl[1][1], dd[n][1], fmaf(cl[1][2], dd[n][2], cl[1][3] * dd[n][3])));
              asm volatile("" : "+v"(tlb[n]));
            }
          }
      } else {
#pragma unroll
        for (int c = 0; c < 6; ++c)
          if (c >= c0 && c < c1) {
            if (W4_EXP & 128) {
              ta[c] = dd[c][0];
              tb[c] = dd[c][1];
            } else if constexpr (XG < 2) {
              const float A = fmaf(alpha, dd[c][1], dd[c][3]);
              const float Bv = fmaf(alpha, dd[c][0], dd[c][2]);
              ta[c] = fmaf(beta, Bv, A);
              tb[c] = fmaf(nbeta, Bv, A);
              asm volatile("" : "+v"(ta[c]), "+v"(tb[c]));  // scalar chains: packed-fp32 VALU (and the moves that feed it) is a loss beside MFMAs
            } else {
              ta[c] = fmaf(4.f, dd[c][0], fmaf(-5.f, dd[c][1], dd[c][2]));
              asm volatile("" : "+v"(ta[c]));
            }
          }
      }
    };
    auto t_mix = [&](const float (&tl)[4], float (&tt)[6]) __attribute__((always_inline)) {  // low resolution: the six up-sampled column samples of a row-transformed patch
      tt[0] = zx0 * fmaf(0.75f, tl[0], 0.25f * tl[1]);
      tt[1] = fmaf(0.25f, tl[0], 0.75f * tl[1]);
      tt[2] = fmaf(0.75f, tl[1], 0.25f * tl[2]);
      tt[3] = fmaf(0.25f, tl[1], 0.75f * tl[2]);
      tt[4] = fmaf(0.75f, tl[2], 0.25f * tl[3]);
      tt[5] = zx5 * fmaf(0.25f, tl[2], 0.75f * tl[3]);
#pragma unroll
      for (int c = 0; c < 6; ++c) asm volatile("" : "+v"(tt[c]));
    };
    auto t_cols = [&](auto xg_tag, auto lr_tag, auto which_tag, float* vslot) __attribute__((always_inline)) {  // which = 0: the first (or only) xi of the group, 1: the second
      constexpr int XG = decltype(xg_tag)::value;
      constexpr bool LR = decltype(lr_tag)::value;
      constexpr int WH = decltype(which_tag)::value;
      if (W4_EXP & 2) return;
      if constexpr (LR) t_mix(WH ? tlb : tla, WH ? tb : ta);
      float v[6];
      if (W4_EXP & 128) {
#pragma unroll
        for (int n = 0; n < 6; ++n) v[n] = (WH ? tb : ta)[n];
      } else {
        w4_col_pass(WH ? tb : ta, v);
      }
      constexpr int xi = XG == 0 ? 1 + WH : (XG == 1 ? 3 + WH : (XG == 2 ? 0 : 5));
      p4_vput<xi * 6, 6>(vslot, vwq, vwd, v);
    };

    // ---- prologue: raw of the slice's first three quarters, weights of its first; transform the first quarter
    load_raw(qbeg);
    store_raw(qbeg);
    load_raw(qbeg + 1);
#pragma unroll
    for (int j = 0; j < 3; ++j) load_w(qbeg, j);
    wf[3] = f32x4{0.f, 0.f, 0.f, 0.f};
    wf9[0] = wf9[1] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    auto transform_all = [&](auto xg_tag, int k) __attribute__((always_inline)) {
      const float* raw = rawbuf + (k & 1) * W4_RAW_FLOATS;
      float* vslot = vbuf + (k & 1) * W4_V_FLOATS;
      constexpr int XG = decltype(xg_tag)::value;
      if (lowres && k >= Q0) {
        t_read(xg_tag, std::true_type{}, raw, 0, 4);
        t_rows(xg_tag, std::true_type{}, 0, 4);
        t_cols(xg_tag, std::true_type{}, std::integral_constant<int, 0>{}, vslot);
        if constexpr (XG < 2) t_cols(xg_tag, std::true_type{}, std::integral_constant<int, 1>{}, vslot);
      } else {
        t_read(xg_tag, std::false_type{}, raw, 0, 6);
        t_rows(xg_tag, std::false_type{}, 0, 6);
        t_cols(xg_tag, std::false_type{}, std::integral_constant<int, 0>{}, vslot);
        if constexpr (XG < 2) t_cols(xg_tag, std::false_type{}, std::integral_constant<int, 1>{}, vslot);
      }
    };
    switch (xg) {
      case 0: transform_all(std::integral_constant<int, 0>{}, qbeg); break;
      case 1: transform_all(std::integral_constant<int, 1>{}, qbeg); break;
      case 2: transform_all(std::integral_constant<int, 2>{}, qbeg); break;
      default: transform_all(std::integral_constant<int, 3>{}, qbeg); break;
    }
    f32x16 acc[9];
#pragma unroll
    for (int x = 0; x < 9; ++x)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    store_raw(qbeg + 1);
    load_raw(qbeg + 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // V of the first quarter and the halo of the second are written
    __builtin_amdgcn_s_barrier();

#if W4_EXP & 16
#define P4_MFMA(a_, b_, c_) (c_)
#else
#define P4_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x2f32(a_, b_, c_, 0, 0, 0)
#endif
#define P4_SB __builtin_amdgcn_sched_barrier(0);
#ifdef W4_STAMP  // diagnostic build (tools/w4/w4_bench.hip -DW4_STAMP): cycles per segment of a quarter, per wave, into ConvArgs::clock_probe.  The s_memtime results are only
    // read after the quarter's own lgkmcnt(0), so the stamps add no wait of their own (an s_memtime read on the spot waits for every LDS operation in flight)
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tile_t1 = __builtin_amdgcn_s_memtime();  // prologue done
#ifndef W4_STSEL
#define W4_STSEL 0xFF  // which stamps are taken (few at a time: every pending result occupies an SGPR pair the compiler must not have to move)
#endif
#define P4_ST(i) if constexpr ((W4_STSEL >> i) & 1) { asm volatile("s_memtime %0" : "=s"(tm[i])); } P4_SB
#define P4_ST_END                                      \
  asm volatile("s_memtime %0" : "=s"(tm[8]));          \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(tm[0]), "+s"(tm[1]), "+s"(tm[2]), "+s"(tm[3]), "+s"(tm[4]), "+s"(tm[5]), "+s"(tm[6]), "+s"(tm[7]), "+s"(tm[8])::"memory");   \
  {                                                    \
    unsigned long long prev_ = 0; bool have_ = false;  \
    _Pragma("unroll") for (int i_ = 0; i_ < 9; ++i_) if (i_ == 8 || ((W4_STSEL >> i_) & 1)) { if (have_) st[i_ - 1] += tm[i_] - prev_; prev_ = tm[i_]; have_ = true; } \
  }
#else
#define P4_ST(i)
#define P4_ST_END
#endif
    // One quarter: 18 MFMAs of quarter q with the transform of quarter q + 1 (kind LR), the weight loads of q + 1 and the halo transfers of q + 3 between them.
    // MM = false: this wave's 32 output channels do not exist (last N tile of a layer with 32 channels mod 64): producer duties only.
    auto quarters = [&](auto xg_tag, auto mm_tag, auto lr_tag, int q0, int q1) __attribute__((always_inline)) {
      constexpr int XG = decltype(xg_tag)::value;
      constexpr bool MM = decltype(mm_tag)::value;
      constexpr bool LR = decltype(lr_tag)::value;
      constexpr bool PAIR = XG < 2;
      constexpr int NC = LR ? 4 : 6, H1 = NC / 2;  // raw columns, read in two halves
      const std::integral_constant<int, 0> first{};
      const std::integral_constant<int, 1> second{};
      for (int q = q0; q < q1; ++q) {
        const float* raw = rawbuf + ((q + 1) & 1) * W4_RAW_FLOATS;
        float* vslot = vbuf + ((q + 1) & 1) * W4_V_FLOATS;
        const float* vq = vbuf + (q & 1) * W4_V_FLOATS + vrq;
        const float* vd = vbuf + (q & 1) * W4_V_FLOATS + vrd;
        f32x4 vf0, vf1, vf2;
        // the previous quarter's last six MFMAs come first (their operands have been in registers since before the barrier), the quarter's memory instructions between them
        P4_ST(0)
        if constexpr (MM) {
          acc[6] = P4_MFMA(wf[3][0], vf3[0], acc[6]);
          acc[7] = P4_MFMA(wf[3][1], vf3[1], acc[7]);
        }
        P4_SB
        store_raw(q + 2);  // (loaded during quarter q - 1; slot q & 1 was transformed during quarter q - 1)
        t_read(xg_tag, lr_tag, raw, 0, H1);
        P4_SB
        P4_ST(1)
        if constexpr (MM) {
          acc[8] = P4_MFMA(wf9[0], vf9.x, acc[8]);
          vf0 = *reinterpret_cast<const f32x4*>(vq);
          acc[6] = P4_MFMA(wf[3][2], vf3[2], acc[6]);
        }
        P4_SB
        load_raw(q + 3);
        P4_SB
        P4_ST(2)
        if constexpr (MM) {
          acc[7] = P4_MFMA(wf[3][3], vf3[3], acc[7]);
          load_w(q, 3);
          vf1 = *reinterpret_cast<const f32x4*>(vq + 256);
          P4_ST(3)
          acc[8] = P4_MFMA(wf9[1], vf9.y, acc[8]);
          load_w(q, 4);
        }
        t_rows(xg_tag, lr_tag, 0, H1);
        t_read(xg_tag, lr_tag, raw, H1, NC);
        P4_SB
        P4_ST(4)
        if constexpr (MM) {
          acc[0] = P4_MFMA(wf[0][0], vf0[0], acc[0]);
          acc[1] = P4_MFMA(wf[0][1], vf0[1], acc[1]);
          vf2 = *reinterpret_cast<const f32x4*>(vq + 512);
          acc[0] = P4_MFMA(wf[0][2], vf0[2], acc[0]);
          acc[1] = P4_MFMA(wf[0][3], vf0[3], acc[1]);
          load_w(q + 1, 0);
        }
        t_rows(xg_tag, lr_tag, H1, NC);
        P4_SB
        P4_ST(5)
        if constexpr (MM) {
          acc[2] = P4_MFMA(wf[1][0], vf1[0], acc[2]);
          acc[3] = P4_MFMA(wf[1][1], vf1[1], acc[3]);
          vf3 = *reinterpret_cast<const f32x4*>(vq + 768);
        }
        t_cols(xg_tag, lr_tag, first, vslot);
        if constexpr (MM) {
          acc[2] = P4_MFMA(wf[1][2], vf1[2], acc[2]);
          acc[3] = P4_MFMA(wf[1][3], vf1[3], acc[3]);
          load_w(q + 1, 1);
        }
        P4_SB
        P4_ST(6)
        if constexpr (MM) {
          acc[4] = P4_MFMA(wf[2][0], vf2[0], acc[4]);
          acc[5] = P4_MFMA(wf[2][1], vf2[1], acc[5]);
          vf9 = *reinterpret_cast<const float2*>(vd);
        }
        if constexpr (PAIR) t_cols(xg_tag, lr_tag, second, vslot);
        if constexpr (MM) {
          acc[4] = P4_MFMA(wf[2][2], vf2[2], acc[4]);
          acc[5] = P4_MFMA(wf[2][3], vf2[3], acc[5]);
          load_w(q + 1, 2);
        }
        P4_SB
        P4_ST(7)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // V[(q + 1) & 1] and raw[q & 1] are written, the fragments of positions 6..8 read
        if (!(W4_EXP & 4)) __builtin_amdgcn_s_barrier();
        P4_SB
        P4_ST_END
      }
    };
    // (the transform of quarter q + 1 is the low-resolution one from q = Q0 - 1 on)
    auto run = [&](auto xg_tag, auto mm_tag) __attribute__((always_inline)) {
      if constexpr (LOWRES) {
        const int qs = min(max(Q0 - 1, qbeg), qend);
        quarters(xg_tag, mm_tag, std::false_type{}, qbeg, qs);
        quarters(xg_tag, mm_tag, std::true_type{}, qs, qend);
      } else {
        quarters(xg_tag, mm_tag, std::false_type{}, qbeg, qend);
      }
    };
    auto run_mm = [&](auto mm_tag) __attribute__((always_inline)) {
      switch (xg) {
        case 0: run(std::integral_constant<int, 0>{}, mm_tag); break;
        case 1: run(std::integral_constant<int, 1>{}, mm_tag); break;
        case 2: run(std::integral_constant<int, 2>{}, mm_tag); break;
        default: run(std::integral_constant<int, 3>{}, mm_tag); break;
      }
    };
    if (nt == 1 && P.ntile * 64 + 32 >= a.coutp) {  // wave-uniform
      run_mm(std::false_type{});
    } else {
      run_mm(std::true_type{});
      // the last quarter's positions 6..8
      acc[6] = P4_MFMA(wf[3][0], vf3[0], acc[6]);
      acc[7] = P4_MFMA(wf[3][1], vf3[1], acc[7]);
      acc[8] = P4_MFMA(wf9[0], vf9.x, acc[8]);
      acc[6] = P4_MFMA(wf[3][2], vf3[2], acc[6]);
      acc[7] = P4_MFMA(wf[3][3], vf3[3], acc[7]);
      acc[8] = P4_MFMA(wf9[1], vf9.y, acc[8]);
    }

#ifdef W4_STAMP
    const unsigned long long tile_t2 = __builtin_amdgcn_s_memtime();  // loop done
#endif
    // ---- epilogue (the loop's last barrier has passed: every LDS read of the tile is done)
    float* const exch = lds;  // [position 36][tile 32][channel quad 8, XOR-swizzled by the tile][4]: 144 KiB
    // Two rounds through LDS, each with HALF of every wave's registers: round r = channel quads k = 2 r, 2 r + 1 of both N halves (144 KiB), so that the accumulators a wave still holds
    // during the first output transform are 72, not 144.  LDS slot of (N half nt, quad k, lane half lh): nt * 4 + (k & 1) * 2 + lh, XOR-swizzled by the tile.
    auto dump = [&](int r) __attribute__((always_inline)) {
      // position p = 9 pg + i at p * 1024 floats (an immediate offset per i); the slot's XOR makes two base addresses per round, made here
      // (hoisted out of the tile loop the 36 addresses were spilled and came back one scratch load at a time: 22 thousand cycles per dump)
      int d0 = ((pg * 9 * 32 + lx) * 8 + ((nt * 4 + lh) ^ (lx & 7))) << 2;
      asm volatile("" : "+v"(d0));
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        float* const dk = exch + (d0 ^ (kk << 3));
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          f32x4 v;
          if (r == 0) {
            v[0] = acc[i][4 * kk + 0];
            v[1] = acc[i][4 * kk + 1];
            v[2] = acc[i][4 * kk + 2];
            v[3] = acc[i][4 * kk + 3];
          } else {
            v[0] = acc[i][8 + 4 * kk + 0];
            v[1] = acc[i][8 + 4 * kk + 1];
            v[2] = acc[i][8 + 4 * kk + 2];
            v[3] = acc[i][8 + 4 * kk + 3];
          }
          *reinterpret_cast<f32x4*>(dk + i * 1024) = v;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // The output transform of a round, all eight waves: one thread per (tile, slot, channel PAIR of the slot): it reads the 36 positions of its two channels (8 bytes each: every
    // value of the exchange is read exactly once -- the round-4 form, one thread per (tile, slot, row half), read every value twice and formed the shared row differences in both
    // halves) and stores all four rows of the 4 x 4 output tile.
    auto finish = [&](int round) __attribute__((always_inline)) {
      const int tile = tid >> 4, cq = (tid >> 1) & 7, hc = tid & 1;
      const int ty = tile >> 3, tx = tile & 7;
      int z0 = ((tile * 8 + (cq ^ (tile & 7))) << 2) + 2 * hc;
      asm volatile("" : "+v"(z0));
      const float* zp = exch + z0;
      // row pass (over xi) per nu: Z[a][nu] for the four output rows a
      f32x2 z[4][6];
#pragma unroll
      for (int n = 0; n < 6; ++n) {
        f32x2 m[6];
#pragma unroll
        for (int x = 0; x < 6; ++x) m[x] = *reinterpret_cast<const f32x2*>(zp + (x * 6 + n) * 1024);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float s12 = m[1][e] + m[2][e], d12 = m[1][e] - m[2][e], s34 = m[3][e] + m[4][e], d34 = m[3][e] - m[4][e];
          z[0][n][e] = (m[0][e] + s12) + s34;
          z[1][n][e] = fmaf(2.f, d34, d12);
          z[2][n][e] = fmaf(4.f, s34, s12);
          z[3][n][e] = fmaf(8.f, d34, d12) + m[5][e];
        }
      }
      const int co = P.ntile * 64 + (cq >> 2) * 32 + round * 16 + (cq & 3) * 4 + 2 * hc;
      const f32x2 bias = KS ? f32x2{0.f, 0.f} : *reinterpret_cast<const f32x2*>(a.bias + co);
      const int oy0 = P.y0 + 4 * ty, ox = P.x0 + 4 * tx;
      const bool live = oy0 < a.H && ox < a.W && co < a.coutp;  // (H and W are multiples of 4: a 4 x 4 tile is inside the image or outside; channel counts are even)
      f32x2 prow[2];  // fused 2 x 2 / 2 max pool (ConvArgs::dst_pool; even H and W: no padded windows): the column-pair maxima of the window's first row
#pragma unroll
      for (int aa = 0; aa < 4; ++aa) {
        const f32x2 s12 = z[aa][1] + z[aa][2], d12 = z[aa][1] - z[aa][2], s34 = z[aa][3] + z[aa][4], d34 = z[aa][3] - z[aa][4];
        f32x2 y[4];
        y[0] = ((z[aa][0] + s12) + s34) + bias;
        y[1] = (d12 + 2.f * d34) + bias;
        y[2] = (s12 + 4.f * s34) + bias;
        y[3] = ((d12 + 8.f * d34) + z[aa][5]) + bias;
        if (live) {
          float* const dp = a.dst + ((size_t)(P.b * a.H + oy0 + aa) * a.W + ox) * a.coutp + co + (KS ? (size_t)ksl * a.split_stride : (size_t)0);
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) {
            f32x2 v = y[bb];
            if (a.relu) {
#pragma unroll
              for (int e = 0; e < 2; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (a.relu_mask_src) {  // backward: this launch completes the gradient of a conv + ReLU output -- that ReLU's mask rides in the (lane-local) store
              const f32x2 f = *reinterpret_cast<const f32x2*>(a.relu_mask_src + (dp - a.dst) + (size_t)bb * a.coutp);
#pragma unroll
              for (int e = 0; e < 2; ++e) v[e] = f[e] > 0.f ? v[e] : 0.f;
            }
            y[bb] = v;
            if ((KS || !a.skip_dst) && !(W4_EXP & 512)) *reinterpret_cast<f32x2*>(dp + (size_t)bb * a.coutp) = v;  // (non-temporal stores: the output stage 2x slower)
          }
          if (!KS && a.dst_pool) {
            f32x2 m0, m1;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              m0[e] = fmaxf(y[0][e], y[1][e]);
              m1[e] = fmaxf(y[2][e], y[3][e]);
            }
            if (aa & 1) {
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                m0[e] = fmaxf(m0[e], prow[0][e]);
                m1[e] = fmaxf(m1[e], prow[1][e]);
              }
              float* const pp = a.dst_pool + ((size_t)(P.b * (a.H >> 1) + ((oy0 + aa) >> 1)) * (a.W >> 1) + (ox >> 1)) * a.coutp + co;
              *reinterpret_cast<f32x2*>(pp) = m0;
              *reinterpret_cast<f32x2*>(pp + a.coutp) = m1;
            } else {
              prow[0] = m0;
              prow[1] = m1;
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
#ifdef W4_STAMP
    unsigned long long e1 = 0, e2 = 0, e3 = 0;
#define P4_ET(x) x = __builtin_amdgcn_s_memtime();
#else
#define P4_ET(x)
#endif
    dump(0);
    __builtin_amdgcn_s_barrier();
    P4_ET(e1)
    finish(0);
    __builtin_amdgcn_s_barrier();
    P4_ET(e2)
    dump(1);
    __builtin_amdgcn_s_barrier();
    P4_ET(e3)
    finish(1);
    __builtin_amdgcn_s_barrier();
#ifdef W4_STAMP
    if (a.clock_probe && lane == 0 && vid == (int)blockIdx.x) {
      unsigned long long* o = a.clock_probe + ((size_t)blockIdx.x * 12 + pw) * 8;
#ifdef W4_STTILE  // the tile's phases (the harness divides by the quarter count): prologue, one quarter, first dump, first output transform, second dump, second output transform
      o[0] = (tile_t1 - tile_t0) * (unsigned long long)(qend - qbeg);
      o[1] = st[0] + st[1] + st[2] + st[3] + st[4] + st[5] + st[6] + st[7];
      o[2] = (e1 - tile_t2) * (unsigned long long)(qend - qbeg);
      o[3] = (e2 - e1) * (unsigned long long)(qend - qbeg);
      o[4] = (e3 - e2) * (unsigned long long)(qend - qbeg);
      o[5] = (__builtin_amdgcn_s_memtime() - e3) * (unsigned long long)(qend - qbeg);
      o[6] = o[7] = 0;
#else
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = st[i];
#endif
    }
#endif
  }
#ifdef W4_CLOCK
  if (a.clock_probe && tid == 0) {
    a.clock_probe[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - clk_c0;
    a.clock_probe[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
  }
#endif
}

int prepare_wino4_kernels() {
  hipError_t e = hipSuccess;
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wino4_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_FLOATS * (int)sizeof(float));
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wino4_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_FLOATS * (int)sizeof(float));
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wino4_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_FLOATS * (int)sizeof(float));
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wino4_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_FLOATS * (int)sizeof(float));
  if (e != hipSuccess) {
    set_error("hipFuncSetAttribute(wino4) failed: %s", hipGetErrorString(e));
    return PH_E_HIP;
  }
  return PH_OK;
}

// Shapes the kernel takes: N tile 64, whole 4x4 Winograd tiles (H and W multiples of 4), channel counts in quarters, every tensor
// addressable through a 32-bit buffer descriptor; no fused pool / head / accumulate / ReLU mask (those stay on the F(2x2,3x3) kernel).
bool wino4_fits(const ConvArgs& a) {
  if (!a.wpack_wino4 || a.bn != 64 || (a.H & 3) || (a.W & 3) || a.head_w || a.accumulate) return false;
  if (a.src1_lowres && (!a.src1 || (a.H & 1) || (a.W & 1))) return false;
  const uint64_t px = (uint64_t)a.B * a.H * a.W;
  if (px * (uint64_t)a.c0p * 4 >= 0xFFFFFF00ull || px * (uint64_t)a.c1p * 4 >= 0xFFFFFF00ull || px >= 0x7FFFFFFFull) return false;
  const uint64_t wbytes = (uint64_t)((a.coutp + 63) / 64) * (uint64_t)((a.c0p + a.c1p) / 4) * W4_Q_FLOATS * 4;
  if (wbytes >= 0xFFFFFF00ull) return false;
  // Against the F(2x2,3x3) kernel, which would run the layer otherwise.  Both kernels run one persistent workgroup per CU, so a launch costs rounds-of-the-chip x time per unit;
  // measured unit times (cfg3 layers, K from 128 to 768 channels): F(2x2,3x3) ~8 us of prologue + epilogue + 2.0 us per half (8 channels) for a 16 x 16-pixel tile, this kernel
  // ~8 us + 1.6 us per quarter (4 channels) for a 32 x 16 one (768 -> 256 at 64 x 64, 32 frames: 4 rounds x 315 us = 1.26 ms measured 1.24; F(2x2,3x3): 8 x 200 = 1.60, measured 1.62),
  // a split-K second stage ~8 us.  Each kernel is priced at the K split it would take (wino4_ksplit_shape / wino2d_ksplit_shape); large batches reduce to the old 2 / 1.3 rule.
  if (a.use_wino4 >= 2) return true;  // forced (handle option conv_wino4 = 3: tests, the timing harness)
  int n_cu = 0;
  if (device_cu_count(&n_cu) != PH_OK || n_cu <= 0) n_cu = 256;
  const bool may_split = a.split_scratch != nullptr;
  const int ks4 = may_split ? wino4_ksplit_shape(a.B, a.H, a.W, a.c0p + a.c1p, a.coutp, a.splitk, n_cu) : 1;
  const int ks2 = may_split ? wino2d_ksplit_shape(a.B, a.H, a.W, a.c0p + a.c1p, a.coutp, a.splitk, n_cu) : 1;
  const long ntc = (a.coutp + 63) / 64;
  const long t4 = (long)((a.H + W4_PH - 1) / W4_PH) * ((a.W + W4_PW - 1) / W4_PW) * a.B * ntc;
  const long t2 = (long)((a.H + 15) / 16) * ((a.W + 15) / 16) * a.B * ntc;
  const double Qn = (a.c0p + a.c1p) / 4.0;
  const double cost4 = (double)((t4 * ks4 + n_cu - 1) / n_cu) * (11.0 + 1.5 * Qn / ks4) + (ks4 > 1 ? 8.0 : 0.0);  // (the pipelined kernel: 3 200 cycles per quarter, 23 000 per tile outside its loop)
  const double cost2 = (double)((t2 * ks2 + n_cu - 1) / n_cu) * (8.0 + 1.0 * Qn / ks2) + (ks2 > 1 ? 8.0 : 0.0);
  return cost4 <= cost2;
}

// K slices a layer of this shape takes on this kernel (1 = no split): where its units fill less than half of the CUs; a slice keeps at least 16 quarters (64 channels)
int wino4_ksplit_shape(int B, int H, int W, int cinp, int coutp, int splitk, int n_cu) {
  if (splitk <= 0 || cinp < 128) return 1;
  const long units = (long)((H + W4_PH - 1) / W4_PH) * ((W + W4_PW - 1) / W4_PW) * B * ((coutp + 63) / 64);
  const int q = cinp / 4;
  if (splitk >= 2) return std::max(1, std::min(splitk, q / 4));
  if (units * 2 > n_cu) return 1;
  return (int)std::max<long>(1, std::min<long>(n_cu / units, q / 16));
}
int64_t wino4_split_scratch_bytes(int B, int H, int W, int cinp, int coutp, int splitk, int n_cu) {
  if ((H & 3) || (W & 3)) return 0;
  const int ks = wino4_ksplit_shape(B, H, W, cinp, coutp, splitk, n_cu);
  return ks > 1 ? (int64_t)ks * B * H * W * coutp * 4 : 0;
}
static int wino4_ksplit(const ConvArgs& a) {
  if (!a.split_scratch || a.relu_mask_src) return 1;
  int n_cu = 0;
  if (device_cu_count(&n_cu) != PH_OK || n_cu <= 0) return 1;
  const int ks = wino4_ksplit_shape(a.B, a.H, a.W, a.c0p + a.c1p, a.coutp, a.splitk, n_cu);
  return (ks > 1 && (int64_t)ks * a.B * a.H * a.W * a.coutp * 4 <= a.split_scratch_bytes) ? ks : 1;
}

int launch_conv3x3_wino4(const ConvArgs& a, hipStream_t s) {
  PH_REQUIRE(wino4_fits(a), "wino4: N tile 64, H and W multiples of 4, no fused head / accumulate, tensors below 4 GiB");
  int n_cu = 0;
  const int rc = device_cu_count(&n_cu);
  if (rc != PH_OK) return rc;
  const int tiles = ((a.W + W4_PW - 1) / W4_PW) * ((a.H + W4_PH - 1) / W4_PH) * a.B;
  const int ntc = (a.coutp + 63) / 64;
  const int ksplit = wino4_ksplit(a);
  if (ksplit > 1) {  // split K: raw partial sums per slice, then the fixed-order second stage (bias, ReLU)
    ConvArgs k = a;
    k.dst = a.split_scratch;
    k.relu = 0;
    k.dst_pool = nullptr;
    k.skip_dst = 0;
    k.ksplit = ksplit;
    k.split_stride = (long long)a.B * a.H * a.W * a.coutp;
    const dim3 grid(std::min(tiles * ntc * ksplit, n_cu));
    if (a.src1_lowres)
      hipLaunchKernelGGL((conv3x3_wino4_kernel<true, true>), grid, dim3(512), (size_t)W4_LDS_FLOATS * sizeof(float), s, k);
    else
      hipLaunchKernelGGL((conv3x3_wino4_kernel<false, true>), grid, dim3(512), (size_t)W4_LDS_FLOATS * sizeof(float), s, k);
    PH_HIP_CHECK(hipGetLastError());
    return launch_splitk_reduce(a.split_scratch, k.split_stride, ksplit, a.bias, a.skip_dst ? nullptr : a.dst, a.dst_pool, a.B, a.H, a.W, a.coutp, a.relu, s);
  }
  if (a.src1_lowres)
    hipLaunchKernelGGL((conv3x3_wino4_kernel<true, false>), dim3(std::min(tiles * ntc, n_cu)), dim3(512), (size_t)W4_LDS_FLOATS * sizeof(float), s, a);
  else
    hipLaunchKernelGGL((conv3x3_wino4_kernel<false, false>), dim3(std::min(tiles * ntc, n_cu)), dim3(512), (size_t)W4_LDS_FLOATS * sizeof(float), s, a);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

}  // namespace ph
