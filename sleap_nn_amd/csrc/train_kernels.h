// Launcher declarations for the training-path kernels (train_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ph {

struct OhkmParams {
  int enabled = 0;
  float hard_to_easy_ratio = 2.0f;
  int min_hard = 2;
  int max_hard = -1;
  float loss_scale = 5.0f;
};

struct WgradArgs {
  const float* x;    // NHWC source activation, cxp channels
  const float* dy;   // NHWC output gradient (already ReLU-masked), coutp channels
  float* slab;       // scratch: [slices][blocks][9][32][32]
  int cxp, coutp, B, H, W;
};

int launch_loss(const float* pred, const float* tgt, const float* sample_w, int B, int C, int H, int W, float loss_weight, const OhkmParams& ok, float* scratch,
                float* dy, float* loss_out, hipStream_t s);
int64_t loss_scratch_floats(int C);
constexpr int PH_MAX_OUTPUTS = 8;
int launch_total_loss(const float* head_loss, const float* w_host, int n, float* out, hipStream_t s);
int launch_head_bwd(const float* dy, const float* y_out, int sigmoid, const float* x, const float* w_packed, int B, int HW, int cin, int cp, int cout,
                    int accumulate, float* dx, float* gw, float* gb, float* scratch, hipStream_t s, const float* x_mask = nullptr, float* conv_gb = nullptr, int conv_cout = 0);
bool head_bwd_can_fold(int cp, int HW);
int64_t head_bwd_scratch_floats(int cp, int cout, int64_t npix);
int launch_relu_mask(float* g, const float* y, size_t n, hipStream_t s);
int launch_pool_bwd(const float* gp, const float* x, int B, int H, int W, int cp, int accumulate, int relu_mask, float* gx, float* gb, int cout, float* scratch, hipStream_t s);
bool pool_bwd_can_sum_bias(int cp);
int launch_upsample_bwd(const float* gy, int B, int H, int W, int cp, int accumulate, float* gx, hipStream_t s);
int launch_bias_grad(const float* g, size_t npix, int cp, int cout, float* gb, float* scratch, hipStream_t s);
int launch_relu_mask_bias_grad(float* g, const float* y_mask, size_t npix, int cp, int cout, float* gb, float* scratch, hipStream_t s);
int64_t bias_scratch_floats(int cp);
int wgrad_slices(int B, int H, int W, int blocks);
int64_t wgrad_slab_floats(int cin_part, int cout, int B, int H, int W);
int launch_wgrad(const WgradArgs& a, int cin_part, int cout, int cin_total, int ci_off, float* grad, hipStream_t s);
// the same gradient in the Winograd F(2x2,3x3) domain (4/9 of the matrix work; layers with >= 32 padded channels on both sides)
int launch_wgrad_wino(const WgradArgs& a, int cin_part, int cout, int cin_total, int ci_off, float* grad, hipStream_t s);
int launch_wgrad16_wino(const WgradArgs& a, int cin_part, int cout, int cin_total, int ci_off, float* grad, hipStream_t s);  // 16 -> 16 layers
int launch_input_wgrad(const void* img, int dtype, const float* dy, int B, int cin, int H, int W, int coutp, int cout, float* gw, float* scratch, hipStream_t s, float* gb = nullptr);
int64_t input_wgrad_scratch_floats(int cin, int cout);
int launch_gather(const float* canon, const int* map, size_t n, float* out, hipStream_t s);
struct GatherSegment {  // one packed buffer of the model: out[i] = map[i] >= 0 ? canon[map[i]] : 0
  const int* map;
  float* dst;
  size_t n;
  unsigned first_block;  // prefix sum of ceil(n / 1024) over the segments before this one
};
int launch_gather_multi(const float* canon, const GatherSegment* seg_dev, int n_seg, unsigned total_blocks, hipStream_t s);

// ---- ConvNeXt encoder ops (convnext_train_kernels.hip)
struct RowWgradArgs {
  const float* dy;  // (M, np) output gradient rows
  const float* x;   // input activations, kp channels (rows = pixels; patch mode gathers 2x2/stride-2 taps)
  float* slab;      // scratch: [slices][blocks][128][128]
  int np, kp, M;
  int patch = 0, tap = 0, H = 0, W = 0;  // patch 1: 2x2/stride-2 taps (dy, dx); patch 2: k x k "same" conv taps (ksize); patch 3: 3x3 stride-2 pad-1 taps (rows on the H/2 x W/2 grid); H, W = size of the gathered map
  int ksize = 3;                         // patch 2: odd kernel size, tap = ky * ksize + kx
  // Linear layers (patch 0): gb != nullptr -> the column sums of dy (the layer's bias gradient, gb_n channels) come out of the same launch: the workgroups
  // of the first k tile (or, operands exchanged, of the first n tile) add up the dy rows they stage anyway.  bias_sum / bias_slab are set by the launcher.
  float* gb = nullptr;
  int gb_n = 0;
  int bias_sum = 0;            // 0 off, 1 the n operand's staging registers (blocks kt == 0), 2 the k operand's (blocks nt == 0)
  float* bias_slab = nullptr;  // [slices][tiles of that operand][128]
};
int launch_gelu_fwd(const float* x, float* y, size_t n, hipStream_t s);
int launch_gelu_bwd(const float* gy, const float* x, float* gx, int accumulate, size_t n, hipStream_t s);
int launch_scale_add_fwd(const float* u, const float* x, const float* scale, float* y, int cp, size_t n, hipStream_t s);
int launch_scale_add_bwd(const float* gy, const float* scale, float* gu, float* gx, int acc_x, int cp, size_t n, hipStream_t s);
int launch_chan_reduce(const float* a, const float* b, const float* stats, size_t npix, int cp, int c, float* out, float* scratch, hipStream_t s);
int64_t chan_reduce_scratch_floats(int cp);
int launch_layernorm_bwd(const float* x, const float* gy, const float* gamma, float* gx, float* stats, int accumulate, int c, int cp, size_t npix, hipStream_t s);
int launch_dwconv7_wgrad(const float* x, const float* gy, int B, int H, int W, int cp, int c, float* gw, float* scratch, hipStream_t s);
int64_t dwconv7_wgrad_scratch_floats(int B, int H, int cp);
int launch_row_wgrad(const RowWgradArgs& a, int n, int k, int taps, float* grad, hipStream_t s);
int launch_row_wgrad_part(const RowWgradArgs& a, int n, int k, int k_total, int k_off, int taps, float* grad, hipStream_t s);
int64_t row_wgrad_slab_floats(int M, int n, int k);
int launch_patch_stem_wgrad(const void* img, int dtype, const float* dy, int B, int cin, int H, int W, int OH, int OW, int k, int stride, int pad, int coutp, int cout, float* gw,
                            float* scratch, hipStream_t s);  // pad 1: the ConvNeXt stem; pad k / 2 with stride 1: the first k x k "same" conv of a UNet
int64_t patch_stem_wgrad_scratch_floats(int cin, int cout, int k, int64_t npix);
int launch_class_ce(const float* p, const float* y, int B, int C, float weight, float* dz, float* loss_out, hipStream_t s);
int launch_global_maxpool_bwd(const float* x, const float* gy, int B, int HW, int cp, int accumulate, float* gx, hipStream_t s);

}  // namespace ph
