// Activation storage formats of the network runtime and the 8-channel load / store helpers every
// format-generic kernel goes through.
//
//   FMT_F32    NHWC fp32, Cp = channels padded to 16, 4 B per channel (training, ConvNeXt, exact-fp32 inference).
//   FMT_SPLIT  "split fp16": a value x is kept as the pair (hi, lo') of fp16 numbers with
//                  hi = rn_f16(x),  lo' = rn_f16((x - hi) * 2^11),   x ~= hi + lo' * 2^-11   (|error| <= 2^-22 |x|),
//              stored per pixel and 16-channel chunk as 64 B = [hi c0-7][hi c8-15][lo' c0-7][lo' c8-15]: four 16-B quads,
//              each one MFMA-ready (v_mfma_f32_32x32x16_f16 wants 8 consecutive k per lane).  Same footprint as fp32, so
//              slot sizes and every 16-B-quad based LDS-DMA plan are those of FMT_F32.  A product a*b is evaluated as
//              a_hi b_hi + (a_hi b_lo' + a_lo' b_hi) 2^-11 on the fp16 matrix pipe with fp32 accumulation: 22-bit
//              products at 16/3 of the fp32-MFMA rate.  lo' is pre-scaled so that it never underflows fp16 when hi is
//              normal; fp16 subnormals are NOT flushed by the gfx950 MFMA (tools/probes/mfma_f16_probe.hip).
//   FMT_F16    plain fp16 (the reference's autocast mode, torch_backend.py:113-143): Cp = channels padded to 32,
//              2 B per channel, 64 B per 32-channel chunk = [c0-7][c8-15][c16-23][c24-31].
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ph {

enum ActFmt { FMT_F32 = 0, FMT_SPLIT = 1, FMT_F16 = 2 };

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr float SPLIT_SCALE = 2048.0f, SPLIT_INV = 1.0f / 2048.0f;

static inline int fmt_cpad(int fmt, int c) { return fmt == FMT_F16 ? (c + 31) / 32 * 32 : (c + 15) / 16 * 16; }
static inline int fmt_bytes_per_channel(int fmt) { return fmt == FMT_F16 ? 2 : 4; }

__device__ __forceinline__ _Float16 split_hi(float v) { return (_Float16)v; }  // v_cvt_f16_f32, round to nearest even
__device__ __forceinline__ _Float16 split_lo(float v, _Float16 hi) { return (_Float16)((v - (float)hi) * SPLIT_SCALE); }

// channels [8g, 8g + 8) of pixel `pix` of a tensor with `cp` padded channels
template <int FMT>
__device__ __forceinline__ void load8(const void* __restrict__ base, size_t pix, int cp, int g, float (&v)[8]) {
  if constexpr (FMT == FMT_F32) {
    const f32x4_t* p = reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(base) + pix * cp + 8 * g);
    const f32x4_t a = p[0], b = p[1];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = a[k];
      v[4 + k] = b[k];
    }
  } else if constexpr (FMT == FMT_SPLIT) {
    const char* p = reinterpret_cast<const char*>(base) + (pix * cp + (size_t)(g >> 1) * 16) * 4 + (g & 1) * 16;
    const f16x8 hi = *reinterpret_cast<const f16x8*>(p), lo = *reinterpret_cast<const f16x8*>(p + 32);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (float)hi[k] + (float)lo[k] * SPLIT_INV;
  } else {
    const f16x8 h = *reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(base) + (pix * cp + 8 * g) * 2);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (float)h[k];
  }
}

template <int FMT>
__device__ __forceinline__ void store8(void* __restrict__ base, size_t pix, int cp, int g, const float (&v)[8]) {
  if constexpr (FMT == FMT_F32) {
    f32x4_t* p = reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(base) + pix * cp + 8 * g);
    p[0] = f32x4_t{v[0], v[1], v[2], v[3]};
    p[1] = f32x4_t{v[4], v[5], v[6], v[7]};
  } else if constexpr (FMT == FMT_SPLIT) {
    char* p = reinterpret_cast<char*>(base) + (pix * cp + (size_t)(g >> 1) * 16) * 4 + (g & 1) * 16;
    f16x8 hi, lo;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const _Float16 h = split_hi(v[k]);
      hi[k] = h;
      lo[k] = split_lo(v[k], h);
    }
    *reinterpret_cast<f16x8*>(p) = hi;
    *reinterpret_cast<f16x8*>(p + 32) = lo;
  } else {
    f16x8 h;
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = (_Float16)v[k];
    *reinterpret_cast<f16x8*>(reinterpret_cast<char*>(base) + (pix * cp + 8 * g) * 2) = h;
  }
}

// four consecutive channels c0 .. c0 + 3 (c0 a multiple of 4) of one pixel: one 16-byte (fp32) or 8-byte piece per half
template <int FMT>
__device__ __forceinline__ void store4(void* __restrict__ base, size_t pix, int cp, int c0, const float (&v)[4]) {
  if constexpr (FMT == FMT_F32) {
    *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(base) + pix * cp + c0) = f32x4_t{v[0], v[1], v[2], v[3]};
  } else if constexpr (FMT == FMT_SPLIT) {
    char* p = reinterpret_cast<char*>(base) + (pix * cp + (size_t)(c0 >> 4) * 16) * 4 + ((c0 >> 3) & 1) * 16 + (c0 & 7) * 2;
    f16x4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const _Float16 h = split_hi(v[k]);
      hi[k] = h;
      lo[k] = split_lo(v[k], h);
    }
    *reinterpret_cast<f16x4*>(p) = hi;
    *reinterpret_cast<f16x4*>(p + 32) = lo;
  } else {
    f16x4 h;
#pragma unroll
    for (int k = 0; k < 4; ++k) h[k] = (_Float16)v[k];
    *reinterpret_cast<f16x4*>(reinterpret_cast<char*>(base) + (pix * cp + c0) * 2) = h;
  }
}

// one channel of one pixel (scalar access for kernels whose lanes run along channels)
template <int FMT>
__device__ __forceinline__ void store1(void* __restrict__ base, size_t pix, int cp, int c, float v) {
  if constexpr (FMT == FMT_F32) {
    reinterpret_cast<float*>(base)[pix * cp + c] = v;
  } else if constexpr (FMT == FMT_SPLIT) {
    _Float16* p = reinterpret_cast<_Float16*>(reinterpret_cast<char*>(base) + (pix * cp + (size_t)(c >> 4) * 16) * 4) + (c & 15);
    const _Float16 hi = split_hi(v);
    p[0] = hi;
    p[16] = split_lo(v, hi);
  } else {
    reinterpret_cast<_Float16*>(base)[pix * cp + c] = (_Float16)v;
  }
}

}  // namespace ph
