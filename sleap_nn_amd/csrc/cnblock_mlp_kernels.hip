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
//   * Weights: per block of 32 hidden channels one LDS image [W1 part: C / 8 pieces | W2 part: C / 8 pieces] of 1-KiB pieces that ARE the A operands
//     ([4 K steps][lane][4 floats]: one ds_read_b128 feeds four MFMAs), brought by LDS-DMA one hidden block ahead (double-buffered), shared by the eight waves of the
//     workgroup: C = 96: 24 KiB per hidden block and buffer.  One barrier per hidden block (96 MFMAs per wave).
//   * Persistent workgroups (one per CU, eight waves = 256 pixels per tile); the epilogue adds b2, applies the layer scale and the residual with 16-byte accesses
//     (a lane's register quad = four consecutive channels of its pixel).
// Summation order differs from the two-GEMM path (K permuted): fp32 rounding only; the ConvNeXt parity tests hold both at the same tolerance.
#include "common.h"
#include "device_math.h"
#include "net_kernels.h"

namespace ph {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// C: channels (a multiple of 32; x and the output are (M, C) rows).  Hidden = 4 C.
template <int C>
__global__ __launch_bounds__(512, 2) void cnblock_mlp_kernel(MlpArgs a) {
  constexpr int NHB = 4 * C / 32;        // hidden blocks of 32 channels
  constexpr int KS1 = C / 2;             // K steps of the first product
  constexpr int NB = C / 32;             // output blocks
  constexpr int P1 = C / 8, P2 = C / 8;  // 1-KiB pieces per hidden block: W1 part, W2 part
  constexpr int IMG = (P1 + P2) * 1024;  // bytes per hidden block
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][IMG] weights | b1 [4 C floats]
  float* const sB1 = reinterpret_cast<float*>(smem + 2 * IMG);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  for (int i = tid; i < 4 * C; i += 512) sB1[i] = a.b1[i];

  auto stage = [&](int hb, int buf) {  // hidden block hb's image into buffer buf: pieces wave, wave + 8, ...
    char* const dst = smem + buf * IMG;
#pragma unroll
    for (int p = wave; p < P1 + P2; p += 8) {
      const char* g = p < P1 ? reinterpret_cast<const char*>(a.w1img) + ((size_t)hb * P1 + p) * 1024 : reinterpret_cast<const char*>(a.w2img) + ((size_t)hb * P2 + (p - P1)) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + lane * 16), (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
    }
  };

  const int tiles = (a.M + 255) >> 8;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int row = tile * 256 + wave * 32 + j;
    const int rowc = row < a.M ? row : a.M - 1;
    stage(0, 0);
    // x of this wave's 32 pixels: lane (j, h) takes channels [h C / 2, (h + 1) C / 2) of pixel j
    float x[KS1];
    {
      const f32x4* xp = reinterpret_cast<const f32x4*>(a.x + (size_t)rowc * C + h * KS1);
#pragma unroll
      for (int q = 0; q < KS1 / 4; ++q) {
        const f32x4 t = xp[q];
        x[4 * q] = t[0]; x[4 * q + 1] = t[1]; x[4 * q + 2] = t[2]; x[4 * q + 3] = t[3];
      }
    }
    f32x16 acc2[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[nb][r] = 0.f;

#pragma unroll 1
    for (int hb = 0; hb < NHB; ++hb) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of block hb (and x, first time round) landed
      __syncthreads();                                   // ... everybody's; every wave is out of the other buffer
      if (hb + 1 < NHB) stage(hb + 1, (hb + 1) & 1);
      const char* const img = smem + (hb & 1) * IMG;
      // ---- first product: D1[hidden 8 (r / 4) + 4 h + (r % 4)][pixel j], K step s = channels (s, C / 2 + s)
      f32x16 acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
      f32x4 wq[2];
      wq[0] = *reinterpret_cast<const f32x4*>(img + lane * 16);
#pragma unroll
      for (int q = 0; q < KS1 / 4; ++q) {
        if (q + 1 < KS1 / 4) wq[(q + 1) & 1] = *reinterpret_cast<const f32x4*>(img + (q + 1) * 1024 + lane * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[q & 1][t], x[4 * q + t], acc1, 0, 0, 0);
      }
      // ---- GELU(D1 + b1): the accumulator registers become the second product's B operands
      float hid[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(sB1 + hb * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int t = 0; t < 4; ++t) hid[4 * g + t] = gelu_f(acc1[4 * g + t] + b[t]);
      }
      // ---- second product: D2[out channel][pixel j] += W2[out][hidden] hid, K step r = the hidden channels register r holds in the two lane halves
      const char* const img2 = img + P1 * 1024;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 w2[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) w2[nb] = *reinterpret_cast<const f32x4*>(img2 + (nb * 4 + g) * 1024 + lane * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[nb][t], hid[4 * g + t], acc2[nb], 0, 0, 0);
      }
    }
    // ---- epilogue: y = residual + scale * (D2 + b2); lane (j, h), block nb, register quad g = channels nb 32 + 8 g + 4 h .. + 3 of pixel j
    if (row < a.M) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = nb * 32 + 8 * g + 4 * h;
          const f32x4 b2 = *reinterpret_cast<const f32x4*>(a.b2 + c0);
          f32x4 v;
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = acc2[nb][4 * g + t] + b2[t];
          if (a.scale) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + c0);
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] *= sc[t];
          }
          if (a.residual) {
            const f32x4 rs = *reinterpret_cast<const f32x4*>(a.residual + (size_t)row * C + c0);
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] += rs[t];
          }
          *reinterpret_cast<f32x4*>(a.dst + (size_t)row * C + c0) = v;
        }
    }
    __syncthreads();  // (the next tile's stage(0, 0) refills buffer 0: every wave must be out of the last hidden blocks' buffers)
  }
}

bool cnblock_mlp_fits(int c, int cp, int hidden) { return c == cp && hidden == 4 * c && c == 96; }
int64_t cnblock_mlp_w_floats(int c) { return (int64_t)4 * c * c; }  // each of the two images holds its 4 C x C matrix once

int launch_cnblock_mlp(const MlpArgs& a, hipStream_t s) {
  PH_REQUIRE(a.x && a.w1img && a.w2img && a.b1 && a.b2 && a.dst && a.M > 0, "cnblock_mlp_kernel: bad arguments");
  PH_REQUIRE(cnblock_mlp_fits(a.C, a.C, 4 * a.C), "cnblock_mlp_kernel: unsupported width");
  int n_cu = 0;
  {
    const int rc_cu = device_cu_count(&n_cu);
    if (rc_cu != PH_OK) return rc_cu;
  }
  constexpr int C = 96;
  const size_t lds = 2 * (C / 4) * 1024 + 4 * C * 4;
  static bool attr_done = false;
  if (!attr_done) {
    PH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(cnblock_mlp_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  const int tiles = (a.M + 255) >> 8;
  hipLaunchKernelGGL(cnblock_mlp_kernel<C>, dim3(std::min(tiles, n_cu)), dim3(512), lds, s, a);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

}  // namespace ph
