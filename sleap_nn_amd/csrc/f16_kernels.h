// Launcher declarations of f16_kernels.hip: the 3x3 convolution on the fp16 matrix pipe (split-fp16 and plain fp16
// precisions) and the format-generic bilinear / pool / head / read-back kernels.  Formats: act_format.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ph {

struct ConvF16Args {
  const float* src0 = nullptr;   // activations in FMT_SPLIT / FMT_F16 (addressed in 4-byte units)
  const float* src1 = nullptr;   // second concat source or nullptr
  int rs0 = 0, rs1 = 0;          // pixel stride of the sources in 4-byte units (split: Cp, plain: Cp / 2)
  int chunks0 = 0, chunks1 = 0;  // K chunks (64 B per pixel each) of the sources
  const float* wpack = nullptr;  // launch_f16_weight_pack output
  const float* bias = nullptr;   // fp32, padded to a multiple of bn
  void* dst = nullptr;           // FMT_SPLIT (prec 3) / FMT_F16 (prec 1)
  void* dst_pool = nullptr;      // optional fused 2x2/2 max pool output, same format
  int skip_dst = 0;              // the full-resolution output is never read (inference plan): only dst_pool is written
  int rs_dst = 0;                // pixel stride of dst / dst_pool in 4-byte units
  int coutp = 0;                 // padded output channels (split: multiple of 16, plain: of 32)
  int B = 0, H = 0, W = 0;
  int relu = 0;
  int bn = 64;                   // N tile (= the tile the weights were packed for)
  int prec = 3;                  // 3 split fp16, 1 plain fp16
  const float* zeros = nullptr;  // >= 64 B of zeros in HBM
  // fused 1x1 head (prec 1, bn 64, coutp 64, no pool): head_dst[b][o][y][x] = (sigmoid)(sum_c head_w[o][c] out[y][x][c] + head_b[o]) on the fp16-rounded output, NCHW fp32
  const float* head_w = nullptr;  // [head_cout][head_wcp] fp32 (the head op's own weights), or nullptr
  const float* head_b = nullptr;
  float* head_dst = nullptr;
  int head_cout = 0, head_wcp = 0, head_sigmoid = 0;
  unsigned long long* clock_probe = nullptr;  // diagnostic (ph_model_set_clock_probe): per workgroup {d s_memtime, d s_memrealtime}; nullptr = off
  // conv3x3_f16_rows_kernel (f16_rows_kernels.hip; prec 1, bn 64 weights): src1 may be the HALF-resolution tensor (the bilinear x2 rides in the loader);
  // the tile plan below is filled by f16_rows_plan
  int src1_lowres = 0;
  int rows_blend16 = 0;  // the folded bilinear x2 in packed fp16 arithmetic (one more fp16 rounding than upsample2x_fmt_kernel's fp32 form)
  int rows_r = 0, rows_wt = 0, rows_mt = 0, rows_mg = 0;   // tile = rows_r rows x rows_wt columns <= 16 rows_mt rows_mg pixels; rows_mg pixel groups x 8 / rows_mg channel slices
  int rows_hp16 = 0, rows_xb = 0, rows_lc16 = 0, rows_lowp = 0, rows_lds = 0;  // halo pitch (16-pixel pieces), staging extension (bytes), low tile pitch / pieces, LDS bytes
  int rows_inv_wt = 0, rows_inv_pw = 0, rows_inv_cc = 0;   // ceil(2^20 / d) for d = rows_wt, rows_wt / 2, rows_wt / 2 + 1
};

struct StemArgs;
int launch_stem_f16(const StemArgs& a, hipStream_t s);  // stem_f16_kernel: the fused first encoder block, both convs on the fp16 matrix pipe (plain fp16 outputs)
// the two-conv encoder block kernel's arguments (block2_c32_f16_kernel, f16_kernels.hip)
struct Block2Args {
  const void* src;      // FMT_F16, 32-channel pixels (the upper 16 channels are not read), B x H x W
  const float* wa;      // first conv: f16_weight_pack_kernel pieces for bn 32, one chunk: [tap 9][n 2][quad 4][row 16][8 f16]
  const float* wb;      // second conv: the same layout
  const float* ba;      // fp32 bias, 32
  const float* bb;
  void* dst_full;       // FMT_F16 32 channels, B x H x W, or nullptr
  void* dst_pool;       // FMT_F16 32 channels, ceil(H/2) x ceil(W/2), or nullptr
  int B, H, W;
  int relu_a, relu_b;
  const float* zeros;
};
int launch_block2_c32_f16(const Block2Args& a, hipStream_t s);  // block2_c32_f16_kernel: conv(<= 16 -> 32) + conv(32 -> 32) (+ pool) of an encoder block in one launch
int prepare_f16_kernels();
int launch_conv3x3_f16(const ConvF16Args& a, hipStream_t s);
double f16_conv_cost(const ConvF16Args& a, int n_cu);   // estimated launch body of conv3x3_f16_persist_kernel, shader cycles
double f16_upsample_cost(int B, int H, int W, int cp, int n_cu);  // ... of upsample2x_fmt_kernel producing the (2H, 2W) tensor from (B, H, W, cp)
double f16_rows_plan(ConvF16Args& a, int n_cu);         // fills the rows_* plan; estimated launch body in shader cycles, < 0: the shape is not taken
int launch_conv3x3_f16_rows(const ConvF16Args& a, hipStream_t s);
int64_t f16_weight_pack_floats(int n_tiles, int chunks0, int chunks1, int bn, int plain);
int launch_f16_weight_pack(const float* w_dma_f32, float* dst, int n_tiles, int chunks0, int chunks1, int bn, int plain, hipStream_t s);
int launch_upsample_fmt(int fmt, const void* src, void* dst, int B, int H, int W, int cp, hipStream_t s, int f16math = 0);  // f16math: plain fp16 only, see upsample2x_fmt_kernel
int launch_pool_fmt(int fmt, const void* src, void* dst, int B, int H, int W, int cp, hipStream_t s);
int launch_head_fmt(int fmt, const void* src, const float* w, const float* bias, float* dst, int B, int HW, int cp, int wcp, int cout, int sigmoid, hipStream_t s);
int launch_slot_to_nchw_fmt(int fmt, const void* src, float* dst, int B, int HW, int cp, int c, hipStream_t s);

}  // namespace ph
