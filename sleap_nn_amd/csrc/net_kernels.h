// Launcher declarations for the encoder-decoder kernels (net_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

namespace ph {

struct ConvArgs {
  const float* src0;   // NHWC, c0p channels
  const float* src1;   // NHWC, c1p channels (nullptr / 0 when single-source)
  const float* wpack;  // [n_tile][chunk][tap][bn][16]
  const float* bias;   // [coutp_rounded_to_bn]
  float* dst;          // NHWC, coutp channels
  int c0p, c1p, coutp;
  int B, H, W;
  int relu;
  int bn;              // N tile: 32 or 64
  unsigned long long* clock_probe = nullptr;  // diagnostic: per block {d s_memtime, d s_memrealtime}; nullptr = off
  const float* wpack_dma = nullptr;  // weights in the LDS-DMA layout (quad-major 16-row pieces), or nullptr
  const float* zeros = nullptr;      // >= 64 B of zeros in HBM (source of out-of-image halo pixels for LDS-DMA)
  int accumulate = 0;      // epilogue adds the existing dst value (gradient accumulation in the backward pass)
  int dma_stagger = 1;     // LDS-DMA kernel: SIMD-partner waves issue their DMA piece at different points of a step
  const float* wpack_wino = nullptr;  // Winograd F(2,3)-along-x weights in LDS order (conv3x3_wino_persist_kernel), or nullptr
  const float* wpack_wino2 = nullptr; // Winograd F(2x2,3x3) weights in LDS order (conv3x3_wino2d_kernel, N tile 64), or nullptr
  const float* wpack_w16 = nullptr;   // wave-private F(2x2,3x3) weights (conv3x3_w16_kernel: Cout 16 / 32, Cin 16 / 32), or nullptr
  int use_w16 = 1;     // handle option "conv_w16"
  const float* wpack_sm = nullptr;    // small-map F(2x2,3x3) weights (conv3x3_sm_kernel, smallmap_kernels.hip), or nullptr
  int use_sm = 0;      // filled from the handle option "conv_smallmap" and the kind of plan: 1 = where estimated faster than the kernel that would run otherwise, 2 = wherever the shape fits
  int use_wino2d = 1;  // N-tile-64 layers on the F(2x2,3x3) kernel where wpack_wino2 exists (handle option "conv_wino2d")
  const float* w16 = nullptr;  // [tap][ci 16][co 16] weights for conv3x3_c16_kernel (16 -> 16 channel layers), or nullptr
  int skip_dst = 0;           // the full-resolution output is never read (inference plan, fused pool): only dst_pool is written (Winograd kernels; others ignore it)
  const float* relu_mask_src = nullptr;  // backward: NHWC tensor shaped like dst (the forward activation this gradient belongs to); the stored value is zeroed where it is <= 0 (the Winograd kernels' plain stores: ask conv3x3_dma_honours_mask)
  // fused 1x1 head (conv3x3_wino2d_kernel, Cout = one N tile of 64): head_dst[b][o][y][x] = (sigmoid)(sum_c head_w[o][c] * out[y][x][c] + head_b[o]), NCHW fp32
  const float* head_w = nullptr;  // [head_cout][head_wcp] row-major (the head op's own weight layout), or nullptr = no fused head
  const float* head_b = nullptr;  // [head_cout]
  float* head_dst = nullptr;
  int head_cout = 0, head_wcp = 0, head_sigmoid = 0;
  float* dst_pool = nullptr;  // optional fused 2x2/2 max pool of the (ReLU'd) output, NHWC ceil(H/2) x ceil(W/2); nullptr = off
  // kernel selection, filled from the model handle's options (ph_model_set_option)
  int use_wino = 1;   // 1: Winograd F(2,3) kernel where wpack_wino exists; 2: only for the N-tile-64 layers; 0: direct 9-tap kernel
  int persist = 1;    // persistent workgroups (one per CU) instead of one tile per workgroup
  int use_c16 = 1;    // 16 -> 16 channel layers on conv3x3_c16_kernel
  // Winograd F(4x4,3x3) (conv3x3_wino4_kernel, wino4_kernels.hip): layers with at least wino4_min_cin padded input channels
  const float* wpack_wino4 = nullptr;  // transformed weights in the kernel's private-ring order, or nullptr
  const float* wpack_wino4s = nullptr; // (tools/w4/wino4s_experiment.inc only: weights in the order of the sixteen-tile experiment kernel)
  int use_wino4 = 0;          // filled from the handle option "conv_wino4" and the kind of plan: 1 = where it is estimated faster than F(2x2,3x3), 2 = wherever the shape fits
  int wino4_min_cin = 64;     // handle option "conv_wino4_min_cin"
  int src1_lowres = 0;        // src1 is (B, H/2, W/2, c1p): bilinear x2 (align_corners = False) is folded into the input transform (wino4 only)
  // split K on the F(2x2,3x3) kernel (small batches: fewer work units than CUs), see conv3x3_wino2d_kernel<.., KS>
  int splitk = 0;                     // handle option "conv_splitk": 0 never, 1 where estimated faster, n >= 2 force n slices
  float* split_scratch = nullptr;     // [ksplit][B][H][W][coutp] partial sums (the plan's tmp range), or nullptr = no split
  long long split_scratch_bytes = 0;
  int ksplit = 1;                     // filled by the launcher for the kernel
  long long split_stride = 0;         // floats per scratch plane
  unsigned* split_counters = nullptr; // zeroed device counters, one per (pixel tile, N tile) of a split launch (owned by the model handle): the workgroup that stores a unit's last
  int split_counters_n = 0;           //   K slice runs the second stage itself; nullptr = the second stage is a launch of its own (splitk_reduce_kernel)
  int splitk_finish = 0;              // handle option "conv_splitk_finish": 0 = the two-launch form (default, fastest), 1 = in-kernel second stage by the last arriver, 2 = shared by the unit's workgroups (both measured slower)
  float* fin_dst = nullptr;           // filled by the launcher: the layer's real outputs / parameters for the in-kernel second stage
  float* fin_dst_pool = nullptr;
  const float* fin_bias = nullptr;
  int fin_relu = 0;
};

struct InputConvArgs {
  const void* src;  // NCHW uint8 / float
  const float* w;   // [tap][cin][coutp]
  const float* bias;
  void* dst;        // NHWC coutp (format out_fmt)
  int dtype;        // 0 u8 (/255), 1 f32 as-is, 2 f32 (/255)
  int cin, coutp, B, H, W, relu;
  int ksize = 3;    // odd kernel size, "same" padding; weights [tap k*k][cin][coutp]
  int out_fmt = 0;  // ActFmt of dst (act_format.h)
  int dst_cp = 0;   // padded channels of dst in its format (FMT_F16 pads to 32)
};

struct StemArgs {
  const void* src;     // NCHW uint8 / float image
  const float* w0;     // [tap][cin][16] first conv (channels padded to 16)
  const float* b0;     // [16]
  const float* w1;     // [tap][n=16][16] second conv, k contiguous
  const float* b1;     // [16]
  const float* w1w = nullptr;  // second conv, Winograd F(2,3) along x: [kernel row][m index 4][n=16][16], or nullptr
  const float* w1w2 = nullptr; // second conv, Winograd F(2x2,3x3): [position 16][n=16][16], or nullptr
  void* dst_full;      // NHWC 16 full resolution or nullptr (format out_fmt)
  void* dst_pool;      // NHWC 16, ceil(H/2) x ceil(W/2)
  int dtype, cin, B, H, W;
  int out_fmt = 0;     // ActFmt of the outputs (act_format.h)
  int wino = 2;        // second conv: 2 Winograd F(2x2,3x3), 1 F(2,3) along x, 0 direct (handle option "stem_wino")
  int f16_mfma = 1;    // plain fp16 outputs: stem_f16_kernel (both convs on the fp16 matrix pipe; handle option "stem_f16mfma") instead of stem_fused_kernel<CIN, 3>
};

struct PatchStemArgs {
  const void* src;  // NCHW uint8 / float image
  const float* w;   // [tap][cin][coutp]
  const float* bias;
  float* dst;       // NHWC coutp, OH x OW
  int dtype, cin, coutp, B, H, W, OH, OW, k, stride;
  // fused LayerNorm2d over the channels of the result (the ConvNeXt stem: conv -> LayerNorm2d; inference plans), as DwConvArgs; nullptr = off
  const float* ln_gamma = nullptr;
  const float* ln_beta = nullptr;
  int ln_c = 0;
};

struct DwConvArgs {
  const float* src;   // NHWC cp
  const float* w;     // [49][cp]
  const float* bias;  // [cp] or nullptr
  float* dst;
  int cp, B, H, W;
  int accumulate = 0;  // dst += result (data gradient into a slot that already holds the residual branch's gradient)
  // fused LayerNorm over the channels of the result (CNBlock: dwconv -> LayerNorm; inference plans): dst = LN(dwconv(src)); nullptr = off
  const float* ln_gamma = nullptr;  // [cp], zero-padded
  const float* ln_beta = nullptr;   // [cp], zero-padded
  int ln_c = 0;                     // true channel count (pad channels are excluded from the moments and written as zeros)
};

struct GemmArgs {
  const float* src0 = nullptr;      // NHWC activations, c0p channels
  const float* src1 = nullptr;      // second concat source (c1p channels) or nullptr
  const float* wpack = nullptr;     // pack_gemm layout
  const float* bias = nullptr;      // padded to n_tiles * bn
  const float* scale = nullptr;     // layer_scale (padded) or nullptr
  const float* residual = nullptr;  // (M, coutp) or nullptr
  float* dst = nullptr;             // (M, coutp)
  const float* zeros = nullptr;     // >= 64 B of zeros in HBM
  int c0p = 0, c1p = 0, coutp = 0, bn = 0;
  int M = 0;                        // output rows
  int mode = 0;                     // 0: row = pixel (Linear); 1: 2x2/stride-2 patches; 2: 3x3 "same" conv; 3: transposed-conv phase; 4: 3x3 stride-2 gather; 5: k x k "same" conv
  int H = 0, W = 0;                 // input spatial size (modes 1, 2)
  int act = 0;                      // 0 none, 1 ReLU, 2 GELU, 3 multiply by GELU'(aux) (Linear modes only), 4 SiLU (modes 3, 4)
  float* dst_pre = nullptr;         // optional second output (M, coutp): the value before the activation (training keeps it)
  const float* aux = nullptr;       // act 3: (M, coutp) pre-activation the GELU derivative is taken at
  unsigned long long* probe = nullptr;  // diagnostic builds (PH_GEMM_STAMP) only
  int late_split = 0;               // which waves issue their DMA pieces one step late (see gemm_mfma_dma_kernel)
  int persist2 = 0;                 // persistent workgroups for the 9-tap mode too (handle option "gemm_persist2")
  int ntaps = 0;                    // mode 3: taps of this output phase (1, 2 or 4)
  int all_phases = 0;               // mode 3: ONE launch for the four output phases (grid.y = phase); wpack_ph holds the four phases' weight packs, ntaps / out_tap are ignored
  const float* wpack_ph[4] = {nullptr, nullptr, nullptr, nullptr};
  int ksize = 0;                    // mode 5: odd kernel size (k x k "same" conv)
  const float* shift = nullptr;     // modes 3, 4 with affine_first: per-channel shift (folded BatchNorm), padded like bias
  int affine_first = 0;             // modes 3, 4: dst = act(scale * (acc + bias) + shift) instead of act(acc + bias) * scale
  // output row mapping: 0 = row m; 1 = row m is output pixel (b, oy, ox) of a 2x2/stride-2 conv and the result is
  // written to input pixel (b, 2oy + (out_tap >> 1), 2ox + (out_tap & 1)) of an out_H x out_W map (its data gradient)
  int out_patch = 0, out_tap = 0, out_H = 0, out_W = 0;
};

int launch_patch_stem(const PatchStemArgs& a, hipStream_t s);
int launch_dwconv7(const DwConvArgs& a, hipStream_t s);
int launch_layernorm(const float* src, const float* gamma, const float* beta, float* dst, int c, int cp, size_t npix, hipStream_t s);
int launch_gemm(const GemmArgs& a, hipStream_t s);
// CNBlock's MLP in one launch (cnblock_mlp_kernels.hip): dst = residual + scale * (W2 gelu(W1 x + b1) + b2) over (M, C) rows
struct MlpArgs {
  const float* x = nullptr;         // (M, C): the LayerNorm output
  const float* w1img = nullptr;     // Linear(C, 4C) weight as [hidden block][C / 8 pieces of [lane][4]] (model.hip: pack_mlp_w1)
  const float* w2img = nullptr;     // Linear(4C, C) weight as [hidden block][C / 8 pieces of [lane][4]] in the K order of the chained product (pack_mlp_w2)
  const float* b1 = nullptr;        // >= 4C
  const float* b2 = nullptr;        // >= C
  const float* scale = nullptr;     // layer scale (>= C) or nullptr
  const float* residual = nullptr;  // (M, C) or nullptr
  float* dst = nullptr;             // (M, C)
  int M = 0, C = 0;
};
bool cnblock_mlp_fits(int c, int cp, int hidden);
int launch_cnblock_mlp(const MlpArgs& a, hipStream_t s);
int gemm_choose_bn(int coutp);
int prepare_convnext_kernels();
int launch_stem(const StemArgs& a, hipStream_t s);
int prepare_kernels();
int conv_lds_bytes(int bn);
int launch_conv3x3(const ConvArgs& a, hipStream_t s);
// wpack [panel][tap 9][bn][16] -> Winograd weights [panel][step 24][n tile][lh][lx][4] (see conv3x3_wino_persist_kernel)
int launch_wino_pack(const float* wpack, float* wino, int panels, int bn, hipStream_t s);
int64_t wino_pack_floats(int panels, int bn);
struct PackSegment {  // one derived weight buffer in a multi-buffer pack launch
  const float* src;
  float* dst;
  unsigned long long total;  // elements of dst
  int bn;
  unsigned first_block;      // prefix sum of ceil(total / 1024) over the segments before this one
};
int launch_wino_pack_multi(const PackSegment* seg_dev, int n_seg, unsigned total_blocks, hipStream_t s);    // F(2,3) buffers
int launch_wino2d_pack_multi(const PackSegment* seg_dev, int n_seg, unsigned total_blocks, hipStream_t s);  // F(2x2,3x3) buffers
int launch_stem_wino_pack(const float* w1, float* w1w, hipStream_t s);
int launch_stem_wino2d_pack(const float* w1, float* w2, hipStream_t s);  // [tap][co][ci] -> [position 16][co][ci]  // [tap][co][ci] -> [kernel row][m index][co][ci]
int launch_conv3x3_dma(const ConvArgs& a, hipStream_t s);
bool conv3x3_dma_honours_mask(const ConvArgs& a);  // launch_conv3x3_dma would run a kernel that applies relu_mask_src
bool conv3x3_dma_is_f2x2(const ConvArgs& a);       // launch_conv3x3_dma would run one of the two F(2x2,3x3) kernels
bool conv3x3_dma_is_wino2d(const ConvArgs& a);     // ... the wave-split one (takes a fused head on a 64-channel output)
bool conv3x3_dma_is_w16_head(const ConvArgs& a);   // ... the wave-private one on a shape whose 1x1 head (<= 16 channels) can ride in its epilogue
int conv3x3_dma_variant(const ConvArgs& a);        // PH_KV_* code (posehip.h) of the kernel launch_conv3x3_dma would run
// wpack [panel][tap 9][bn][16] -> F(2x2,3x3) weights [panel][g 2][xi 4][nu 4][n tile][lh][lx][4] (see conv3x3_wino2d_kernel)
int launch_wino2d_pack(const float* wpack, float* wino, int panels, int bn, hipStream_t s);
int64_t wino2d_pack_floats(int panels, int bn);
int launch_conv3x3_wino2d(const ConvArgs& a, hipStream_t s);
// wpack [panel][tap 9][64][16] -> F(4x4,3x3) weights [n tile][quarter][wave][fragment slot][lane] (see conv3x3_wino4_kernel)
int launch_wino4_pack(const float* wpack, float* wino, int ntiles, int nchunks, hipStream_t s);
int64_t wino4_pack_floats(int ntiles, int nchunks);
int launch_conv3x3_wino4(const ConvArgs& a, hipStream_t s);
bool wino4_fits(const ConvArgs& a);
bool conv3x3_dma_is_wino4(const ConvArgs& a);      // launch_conv3x3_dma would run the F(4x4,3x3) kernel (the only one that folds a half-resolution src1)
int prepare_wino4_kernels();
bool wino2d_fits(const ConvArgs& a);  // sources addressable through the kernel's 32-bit buffer descriptors
int wino2d_ksplit(const ConvArgs& a);  // K slices launch_conv3x3_wino2d would use for this launch (1 = the one-stage kernel)
int launch_splitk_reduce(const float* part, long long stride, int ks, const float* bias, float* dst, float* dst_pool, int B, int H, int W, int coutp, int relu, hipStream_t s);
int wino4_ksplit_shape(int B, int H, int W, int cinp, int coutp, int splitk, int n_cu);            // K slices the F(4x4,3x3) kernel would take
int64_t wino4_split_scratch_bytes(int B, int H, int W, int cinp, int coutp, int splitk, int n_cu);
int wino2d_ksplit_shape(int B, int H, int W, int cinp, int coutp, int splitk, int n_cu);
int64_t wino2d_split_scratch_bytes(int B, int H, int W, int cinp, int coutp, int splitk, int n_cu);  // scratch a plan reserves for such a layer
int prepare_wino2d_kernels();
// wpack [chunk][tap 9][32][16] -> wave-private F(2x2,3x3) weights [chunk][position][N block][kq][n][4] (see conv3x3_w16_kernel)
int launch_w16_pack(const float* wpack, float* w16, int chunks, int nbs, hipStream_t s);  // nbs: N blocks of 16 output channels (1 or 2)
int64_t w16_pack_floats(int chunks, int nbs);
int launch_conv3x3_w16(const ConvArgs& a, hipStream_t s);
bool w16_fits(const ConvArgs& a);
bool w16_takes_head(const ConvArgs& a);  // the wave-private kernel would run this launch and can carry a fused 1x1 head (<= 16 head channels, head_wcp = coutp)
bool w16_shape_ok(int c0p, int c1p, int coutp);  // channel counts the wave-private kernel takes (c1p = 0: one source)
int prepare_w16_kernels();
// wpack [n tile][chunk][tap 9][bn][16] -> small-map F(2x2,3x3) weights [N block of 16][chunk of 32][pair][wave][half][lane][4] (see conv3x3_sm_kernel)
int launch_sm_pack(const float* wpack, float* dst, int nblocks, int chunks16, int bn, hipStream_t s);
int64_t sm_pack_floats(int nblocks, int chunks16);
int launch_conv3x3_sm(const ConvArgs& a, hipStream_t s);
bool sm_fits(const ConvArgs& a);
double sm_cost_us(const ConvArgs& a, int n_cu);
bool conv3x3_takes_sm(const ConvArgs& a);  // the small-map kernel runs this launch (inference plans: where its estimate beats the kernel launch_conv3x3_dma would pick); it also folds a half-resolution src1
int launch_input_conv(const InputConvArgs& a, hipStream_t s);
int launch_pool(const float* src, float* dst, int B, int H, int W, int cp, hipStream_t s);
int launch_upsample(const float* src, float* dst, int B, int H, int W, int cp, hipStream_t s);
int launch_zero_stuff(const float* src, float* dst, int B, int H, int W, int cp, hipStream_t s);
int launch_global_maxpool(const float* src, float* dst, int B, int HW, int cp, hipStream_t s);
int launch_softmax_rows(float* x, int rows, int n, hipStream_t s);
int launch_nhwc_to_nchw(const float* src, float* dst, int B, int HW, int cp, int c, hipStream_t s);

}  // namespace ph
