// CNBlock's MLP in ONE launch on the fp32 matrix cores (gfx950): y = residual + layer_scale * (W2 gelu(W1 x + b1) + b2), per pixel, the 4C-wide hidden
// activation never leaves the registers.
//
// Reference semantics: torchvision CNBlock (convnext.py:96 builds the encoder from it): ... LayerNorm -> Linear(C, 4C) -> GELU -> Linear(4C, C) -> layer_scale * . + x.
// As two row GEMMs (gemm_mfma_dma_kernel) the stage-0 block (C = 96, 2.4 M pixels at cfg4) writes and re-reads a 3.6-GB hidden tensor and runs its K = 96 product at
// 82 TFLOP/s (three K stages per tile: prologue, epilogue and the output stream dominate), 3.9 ms per block against 2.2 ms of MFMA time.
//
//   * CHAINED products.  A wave owns 32 pixels.  First product, transposed (weights = A operand): D1[hidden i][pixel j] over a block of 32 hidden channels.  In the
//     accumulator layout of v_mfma_f32_32x32x2_f32 lane (j, h = lane / 32) holds pixel j and, in register r, hidden channel 8 (r / 4) + 4 h + (r % 4) -- which is
//     exactly a B operand of the same instruction (lane (j, h) supplies k = h of a two-deep K step for pixel j) if the K order of the second product is chosen to
//     match: K step r of the second product = hidden channels (8 (r / 4) + (r % 4), 8 (r / 4) + 4 + (r % 4)).  The order of a sum is free, the weights of the second
//     Linear are packed in that order, and GELU(D1 + b1) goes from the accumulator registers straight into the second product: no LDS, no HBM, no shuffle.
//   * x stays in registers for the whole tile (lane (j, h) holds channels [h C / 2, (h + 1) C / 2) of pixel j: C / 2 registers; the first product's K order is chosen
//     for THAT: step s = channels (s, C / 2 + s)), the output accumulators (C / 32 blocks of 16 registers) too.
//   * Weights come straight from L2 into the A-operand registers: per block of 32 hidden channels 24 pieces of 1 KiB ([4 K steps][lane][4 floats]: one 16-byte load
//     per lane feeds four MFMAs) in the order the wave consumes them, a ring of eight pieces (32 registers) ahead of their use -- ~2 000 MFMA cycles of cover.  No LDS
//     for the weights, no barrier anywhere in the loop: the eight waves of a workgroup drift apart and fill each other's gaps.  (First version: the pieces by LDS-DMA
//     into a double-buffered LDS image shared by the workgroup, one barrier per hidden block: 3.37 ms per stage-0 block at cfg4 against 3.92 for the two GEMMs.)
//     Two accumulator chains in the first product (alternate K steps) instead of one: the same 3.06 ms per 96-channel block.
//   * Persistent workgroups (one per CU, eight waves = 256 pixels per tile); the epilogue adds b2, applies the layer scale and the residual with 16-byte accesses
//     (a lane's register quad = four consecutive channels of its pixel).
// Summation order differs from the two-GEMM path (K permuted): fp32 rounding only; the ConvNeXt parity tests hold both at the same tolerance.
#include "common.h"
#include "device_math.h"
#include "net_kernels.h"

namespace ph {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef PH_MLP_EXP
#define PH_MLP_EXP 0  // timing experiments only (wrong results): 1 = no GELU arithmetic, 2 = the weight ring is never refilled
#endif

// C: channels (a multiple of 32; x and the output are (M, C) rows).  Hidden = 4 C.  NW: waves per workgroup.  PXB: 32-pixel blocks per wave (2: every weight piece feeds eight
// MFMAs on two independent accumulator sets instead of four on one: half the weight traffic per FLOP, and a wave that is alone on its SIMD for a while still has two chains to issue).
template <int C, int NW, int PXB>
__global__ __launch_bounds__(64 * NW, NW / 4) void cnblock_mlp_kernel(MlpArgs a) {
  constexpr int NHB = 4 * C / 32;        // hidden blocks of 32 channels
  constexpr int KS1 = C / 2;             // K steps of the first product
  constexpr int NB = C / 32;             // output blocks
  constexpr int P1 = C / 8, P2 = C / 8;  // 1-KiB pieces per hidden block: W1 part, W2 part
  constexpr int NP = P1 + P2;            // pieces a wave consumes per hidden block, in this order: W1 q = 0 .. P1 - 1, then W2 (g, nb) = (0, 0), (0, 1), ...
  constexpr int RING = (C > 96 || NW > 8 || PXB > 1) ? 4 : 8;   // (192 channels / two pixel blocks: x and the output accumulators take 192 registers; four pieces ahead is what is left)
  static_assert(NP % RING == 0 && P1 % 4 == 0, "ring slots must be static");
  __shared__ float sB1[4 * C];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  for (int i = tid; i < 4 * C; i += 64 * NW) sB1[i] = a.b1[i];
  __syncthreads();

  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w1img, 0, NHB * P1 * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2img, 0, NHB * P2 * 1024, 0x00020000);
  const unsigned lane_off = (unsigned)lane * 16u;
  // piece u (consumption order) of hidden block hb
  auto fetch = [&](int u, int hb) __attribute__((always_inline)) -> f32x4 {
    if (u < P1) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, lane_off, (hb * P1 + u) * 1024, 0));
    const int g = (u - P1) / NB, nb = (u - P1) % NB;
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2, lane_off, (hb * P2 + nb * 4 + g) * 1024, 0));
  };
  f32x4 ring[RING];
#pragma unroll
  for (int u = 0; u < RING; ++u) ring[u] = fetch(u, 0);

  constexpr int TR = 32 * NW * PXB;  // rows per tile
  const int tiles = (a.M + TR - 1) / TR;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    int row[PXB];
    // x of this wave's pixels: lane (j, h) takes channels [h C / 2, (h + 1) C / 2) of pixel j of each of its blocks
    float x[PXB][KS1];
#pragma unroll
    for (int pb = 0; pb < PXB; ++pb) {
      row[pb] = tile * TR + (wave * PXB + pb) * 32 + j;
      const int rowc = row[pb] < a.M ? row[pb] : a.M - 1;
      const f32x4* xp = reinterpret_cast<const f32x4*>(a.x + (size_t)rowc * C + h * KS1);
#pragma unroll
      for (int q = 0; q < KS1 / 4; ++q) {
        const f32x4 t = xp[q];
        x[pb][4 * q] = t[0]; x[pb][4 * q + 1] = t[1]; x[pb][4 * q + 2] = t[2]; x[pb][4 * q + 3] = t[3];
      }
    }
    f32x16 acc2[PXB][NB];
#pragma unroll
    for (int pb = 0; pb < PXB; ++pb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[pb][nb][r] = 0.f;

#pragma unroll 1
    for (int hb = 0; hb < NHB; ++hb) {
      const int nhb = hb + 1 < NHB ? hb + 1 : 0;  // (the ring runs on into the next tile's first block)
      // ---- first product: D1[hidden 8 (r / 4) + 4 h + (r % 4)][pixel j], K step s = channels (s, C / 2 + s)
      f32x16 acc1[PXB];
#pragma unroll
      for (int pb = 0; pb < PXB; ++pb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[pb][r] = 0.f;
#pragma unroll
      for (int u = 0; u < P1; ++u) {
        const f32x4 w = ring[u % RING];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int pb = 0; pb < PXB; ++pb) acc1[pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t], x[pb][4 * u + t], acc1[pb], 0, 0, 0);
        if (!(PH_MLP_EXP & 2)) ring[u % RING] = u + RING < NP ? fetch(u + RING, hb) : fetch(u + RING - NP, nhb);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- GELU(D1 + b1): the accumulator registers become the second product's B operands
      float hid[PXB][16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(sB1 + hb * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int pb = 0; pb < PXB; ++pb)
#pragma unroll
          for (int t = 0; t < 4; t += 2) {  // (packed fp32: two values per instruction, gelu_f's bits)
            const ph_f32x2 s2 = ph_f32x2{acc1[pb][4 * g + t], acc1[pb][4 * g + t + 1]} + ph_f32x2{b[t], b[t + 1]};
            const ph_f32x2 v = (PH_MLP_EXP & 1) ? s2 : gelu_f2(s2);
            hid[pb][4 * g + t] = v[0];
            hid[pb][4 * g + t + 1] = v[1];
          }
      }
      // ---- second product: D2[out channel][pixel j] += W2[out][hidden] hid, K step r = the hidden channels register r holds in the two lane halves
#pragma unroll
      for (int u = P1; u < NP; ++u) {
        const int g = (u - P1) / NB, nb = (u - P1) % NB;
        const f32x4 w = ring[u % RING];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int pb = 0; pb < PXB; ++pb) acc2[pb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t], hid[pb][4 * g + t], acc2[pb][nb], 0, 0, 0);
        if (!(PH_MLP_EXP & 2)) ring[u % RING] = u + RING < NP ? fetch(u + RING, hb) : fetch(u + RING - NP, nhb);
        if (nb == NB - 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- epilogue: y = residual + scale * (D2 + b2); lane (j, h), block nb, register quad g = channels nb 32 + 8 g + 4 h .. + 3 of pixel j
#pragma unroll
    for (int pb = 0; pb < PXB; ++pb)
      if (row[pb] < a.M) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int c0 = nb * 32 + 8 * g + 4 * h;
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(a.b2 + c0);
            f32x4 v;
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = acc2[pb][nb][4 * g + t] + b2[t];
            if (a.scale) {
              const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + c0);
#pragma unroll
              for (int t = 0; t < 4; ++t) v[t] *= sc[t];
            }
            if (a.residual) {
              const f32x4 rs = *reinterpret_cast<const f32x4*>(a.residual + (size_t)row[pb] * C + c0);
#pragma unroll
              for (int t = 0; t < 4; ++t) v[t] += rs[t];
            }
            *reinterpret_cast<f32x4*>(a.dst + (size_t)row[pb] * C + c0) = v;
          }
      }
  }
}

bool cnblock_mlp_fits(int c, int cp, int hidden) { return c == cp && hidden == 4 * c && (c == 96 || c == 192); }
int64_t cnblock_mlp_w_floats(int c) { return (int64_t)4 * c * c; }  // each of the two images holds its 4 C x C matrix once

int launch_cnblock_mlp(const MlpArgs& a, hipStream_t s) {
  PH_REQUIRE(a.x && a.w1img && a.w2img && a.b1 && a.b2 && a.dst && a.M > 0, "cnblock_mlp_kernel: bad arguments");
  PH_REQUIRE(cnblock_mlp_fits(a.C, a.C, 4 * a.C), "cnblock_mlp_kernel: unsupported width");
  int n_cu = 0;
  {
    const int rc_cu = device_cu_count(&n_cu);
    if (rc_cu != PH_OK) return rc_cu;
  }
#ifndef PH_MLP_PXB96
#define PH_MLP_PXB96 1  // (2 = 64 pixels per wave, ring of four, 19 registers spilled: 3.15 - 3.18 ms per 96-channel block against 3.06)
#endif
  if (a.C == 96) {
    const int tiles = (a.M + 256 * PH_MLP_PXB96 - 1) / (256 * PH_MLP_PXB96);
    hipLaunchKernelGGL((cnblock_mlp_kernel<96, 8, PH_MLP_PXB96>), dim3(std::min(tiles, n_cu)), dim3(512), 0, s, a);
  } else {
    const int tiles = (a.M + 255) >> 8;
    hipLaunchKernelGGL((cnblock_mlp_kernel<192, 8, 1>), dim3(std::min(tiles, n_cu)), dim3(512), 0, s, a);
  }
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

}  // namespace ph
