// Internal structures of the network runtime shared by model.hip (forward) and train.hip (backward).
#pragma once
#include <string>
#include <vector>

#include "act_format.h"
#include "common.h"
#include "f16_kernels.h"
#include "net_kernels.h"

namespace ph {

struct PackedOp {
  ph_op_desc d;
  float* w_dev = nullptr;
  float* b_dev = nullptr;
  float* w2_dev = nullptr;
  float* b2_dev = nullptr;
  float* w_dma_dev = nullptr;  // conv weights in the LDS-DMA (quad-major piece) layout
  float* w_mlp_dev = nullptr;  // Linear of a CNBlock MLP pair (inference programs): its weight as cnblock_mlp_kernel's LDS images
  int bn = 0;
  float* wd_gemm_dev[4] = {nullptr, nullptr, nullptr, nullptr};  // Linear / 2x2 conv: transposed weights (per tap) for the data gradient
  int bn_dg = 0;
  float* dw_flip_dev = nullptr;  // depthwise conv: spatially flipped taps (data gradient)
  float* w_gemm_dev = nullptr;  // 3x3 conv weights in the row-GEMM layout (small feature maps)
  float* b_gemm_dev = nullptr;
  int bn_g = 0;
  // data-gradient weights of a 3x3 conv (flipped taps, in/out swapped), one set per concat source
  float* wd_dev[2] = {nullptr, nullptr};
  float* wd_dma_dev[2] = {nullptr, nullptr};
  float* zero_bias_dev = nullptr;
  float* wdk_gemm_dev[2] = {nullptr, nullptr};  // k x k conv (k != 3): data-gradient weights (flipped taps, in/out swapped) in the row-GEMM layout, per concat source
  int bn_dk[2] = {0, 0};
  int bn_d[2] = {0, 0};
  float* w_wino_dev = nullptr;             // 3x3 conv, N tile 64: Winograd F(2,3) weights derived on the device from w_dev
  float* wd_wino_dev[2] = {nullptr, nullptr};  // ... and from wd_dev (data gradient)
  float* w_stem2_dev = nullptr;            // fused stem: F(2x2,3x3) weights of the second conv
  float* w_w16_dev = nullptr;              // 3x3 conv, Cout 16 / 32, Cin 16 / 32: wave-private F(2x2,3x3) weights derived from w_dev
  float* wd_w16_dev[2] = {nullptr, nullptr};  // ... and from wd_dev (data gradient)
  float* w_wino2_dev = nullptr;            // 3x3 conv, N tile 64: Winograd F(2x2,3x3) weights derived from w_dev
  float* wd_wino2_dev[2] = {nullptr, nullptr};  // ... and from wd_dev (data gradient)
  float* w_n64_dev = nullptr;              // 3x3 conv with 32 output channels and >= 64 input channels: the weights packed once more for an N tile of 64 (upper half zeros) ...
  float* b_n64_dev = nullptr;              // ... with a 64-entry bias, so that the layer can run on the F(2x2,3x3) kernel's half-empty-N-tile form (w_wino2_dev is derived from w_n64_dev)
  float* w_sm_dev = nullptr;               // 3x3 conv: small-map F(2x2,3x3) weights derived from w_dev (conv3x3_sm_kernel)
  float* w_wino4_dev = nullptr;            // 3x3 conv, N tile 64, >= 128 padded input channels: Winograd F(4x4,3x3) weights derived from w_dev
  float* wd_wino4_dev[2] = {nullptr, nullptr};  // ... and from wd_dev (data gradient: Cout input channels)
  // ConvTranspose2d(k3, s2, p1, op1) as four output-phase row GEMMs (mode 3) + its data gradient (mode 4)
  float* wt_phase_dev[4] = {nullptr, nullptr, nullptr, nullptr};
  float* bt_dev = nullptr;       // bias padded to the GEMM's N tiles
  float* wt_dgrad_dev = nullptr; // 3x3 stride-2 conv weights (out = cin0, in = cout) in the row-GEMM layout
  float* wt_scale_dev = nullptr; // optional folded-BatchNorm scale / shift (weight2 / bias2), padded like bt_dev
  float* wt_shift_dev = nullptr;
  int bn_t = 0, bn_td = 0;
  float* w_f16_dev[2] = {nullptr, nullptr};  // 3x3 conv / transposed conv on the fp16 matrix pipe: [0] split-fp16, [1] plain fp16 weights (derived lazily from w_dma_dev)
  float* w16_dev = nullptr;   // 16 -> 16 channel 3x3 conv: [tap][ci][co] for conv3x3_c16_kernel
  float* wd16_dev = nullptr;  // ... and of its data gradient
};

// One packed device buffer and the gather map that rebuilds it from the canonical parameter arena
// (map[i] = index into the arena, -1 = structural zero).
struct PackedBuffer {
  float* dst = nullptr;
  int* map = nullptr;
  size_t n = 0;
};

// A buffer computed on the device from a packed buffer (not a pure gather of parameters): Winograd conv weights.
struct DerivedBuffer {
  const float* src = nullptr;
  float* dst = nullptr;
  int panels = 0, bn = 0;  // bn == 0: the fused stem's second conv (launch_stem_wino_pack)
  int kind = 0;            // 0: Winograd transform of src; 1: fp16 weight pack of the LDS-DMA panels (launch_f16_weight_pack); 2: F(2x2,3x3) transform; 3: wave-private F(2x2,3x3) transform (panels = chunks); 4: the fused stem's F(2x2,3x3) transform; 5: F(4x4,3x3) transform (panels = N tiles, bn = 16-channel chunks); 6: small-map F(2x2,3x3) transform (panels = N blocks of 16, bn = 16-channel chunks, n_tiles = the source packing's N tile)
  int n_tiles = 0, chunks0 = 0, chunks1 = 0, plain = 0;  // kind 1
};

struct SlotShape {
  int c = 0, cp = 0, h = 0, w = 0;  // cp: channels padded for the plan's format
  int64_t offset = -1;
  int def_op = -1;                  // index of the op that writes the slot
};

struct Plan {
  std::vector<SlotShape> slots;
  int64_t tmp_offset = 0, tmp_bytes = 0, total = 0;
  bool reuse = false;  // slots share memory by lifetime (handle option workspace_reuse)
  std::vector<char> unread;  // reuse plans: slot has no reader in the program (a fused conv+pool's full-resolution output the decoder never taps)
  int fmt = FMT_F32;  // activation format of every slot (act_format.h)
  int bpc = 4;        // bytes per channel in that format
};

}  // namespace ph

struct ph_model {
  std::vector<ph::PackedOp> ops;
  int n_slots = 0, n_outputs = 0;
  std::vector<void*> allocs;
  void* gather_table_dev = nullptr;           // GatherSegment table of `packed` (ph_model_set_params: one gather launch for all of them)
  int gather_segments = 0;
  unsigned gather_blocks = 0;
  bool pack_tables_built = false;             // PackSegment tables of the F(2,3) [0] / F(2x2,3x3) [1] buffers in `derived`
  void* pack_table_dev[2] = {nullptr, nullptr};
  int pack_segments[2] = {0, 0};
  unsigned pack_blocks[2] = {0, 0};
  // last forward (for ph_model_read_slot)
  ph::Plan last_plan;
  char* last_ws = nullptr;
  int last_batch = 0;
  std::vector<int> last_variant;  // PH_KV_* code of the kernel each op of the last forward ran (ph_model_last_kernels)
  // optional per-op HIP-event timing (ph_model_set_profiling)
  bool profiling = false;
  bool events_pending = false;
  std::vector<hipEvent_t> ev;      // n_ops + 1 events: ev[i] before op i, ev[n_ops] after the last
  std::vector<double> op_ms;       // accumulated per op
  int profiled_forwards = 0;
  unsigned long long* clock_probe = nullptr;  // diagnostic buffer (ph_model_set_clock_probe)
  ph_comm* comm = nullptr;                    // ph_model_set_comm: ph_model_backward exchanges the gradient arena over it (two buckets on comm_stream)
  hipStream_t comm_stream = nullptr;
  hipEvent_t comm_ev[3] = {nullptr, nullptr, nullptr};  // tail final / sweep done (caller's stream) / exchange done (side stream)
  std::vector<hipEvent_t> comm_events_owned;
  hipEvent_t bucket_event = nullptr;          // recorded mid-backward when the arena tail is final (ph_model_set_bucket_event)
  float* zeros_dev = nullptr;                 // zero page for LDS-DMA halo padding
  unsigned* split_counters_dev = nullptr;     // PH_SPLIT_COUNTERS zeroed counters of the split-K launches' in-kernel second stage (a handle's forwards are serialised, as its plan / profiling state already requires)
  int conv_splitk_finish = 0;                 // "conv_splitk_finish": 0 (default) = splitk_reduce_kernel as a launch of its own; 1 = the last-arriving workgroup of a split-K unit runs the second stage in the same launch; 2 (round 5) = the unit's workgroups wait (bounded) for all slices and SHARE the second stage (claimed by atomic bits; the last arriver sweeps unclaimed shares: no deadlock).  All three bit-identical.  Measured (cfg1 forward): 299 us (0) / 391 (1: one workgroup adds up to 24 planes of its 64-KiB tile at one CU's memory rate) / 370 (2: the wait for the slowest slice + device-scope fences + remote reads of the other XCDs' planes cost more than the ~6-us launch that spreads the same bytes over the chip)
  std::vector<int64_t> weight_offset;          // canonical arena offset of weights[i] (ph_model_create order)
  std::vector<int64_t> weight_numel;
  int64_t n_params = 0;
  bool wino4_stale = false;                    // ph_model_set_params skipped the F(4x4,3x3) weights (training handle): refresh before their next use
  std::vector<ph::DerivedBuffer> derived;      // refreshed after the gathers of ph_model_set_params
  std::vector<ph::PackedBuffer> packed;        // every packed weight buffer, for ph_model_set_params
  // ---- per-handle options (ph_model_set_option); the library reads no environment variables
  int use_dma = 1;                            // "conv_dma": 0 selects the register-staged 3x3 kernel
  int dma32 = 1;                              // "conv_dma32": Cout <= 48 layers on the LDS-DMA kernel (BN = 32) instead of the register-staged one
  int conv_wino = 1;                          // "conv_wino": Winograd F(2,3) 3x3 kernels (2: N-tile-64 layers only, 0: direct 9-tap kernels)
  int conv_w16 = 1;                           // "conv_w16": Cout-32 / Cin-16-or-32 3x3 convs on the wave-private F(2x2,3x3) kernel
  int conv_wino2d = 1;                        // "conv_wino2d": N-tile-64 3x3 convs on the F(2x2,3x3) kernel (0: the F(2,3)-along-x kernel)
  int conv_wino4 = 1;                         // "conv_wino4": K-heavy N-tile-64 3x3 convs on the F(4x4,3x3) kernel: 1 in inference plans (workspace_reuse), 2 in every plan -- both where wino4_fits estimates it faster than F(2x2,3x3) --, 3 every plan and wherever the shape fits, 0 never
  int conv_wino4_min_cin = 64;                // "conv_wino4_min_cin": padded input channels (both sources) from which a layer takes that kernel
  int conv_n32_wino2d = 1;                    // "conv_n32_wino2d" (1: inference plans, 2: every plan, 0: never): Cout-32 layers with >= 64 input channels (the last decoder level of an output-stride-2 UNet: 96 -> 32) on the F(2x2,3x3) kernel with a half-empty N tile of 64 instead of the N-tile-32 F(2,3) kernel
  int block_fuse = 1;                         // "block_fuse"
  int mlp_fuse = 1;                           // "mlp_fuse": CNBlock's Linear -> GELU -> Linear -> scale + residual in one launch (cnblock_mlp_kernel) where the width fits
  int stem_f16mfma = 1;                       // "stem_f16mfma"
  int upsample_f16math = 1;                   // "upsample_f16math"
  int conv_f16_rows = 1;                      // "conv_f16_rows" (plain fp16: the row-tile kernel of f16_rows_kernels.hip; 1 where estimated faster, 2 wherever it fits, 0 never)
  int conv_smallmap = 1;                      // "conv_smallmap" (1: inference plans with conv_splitk = 1 (the automatic small-batch routing), where estimated faster; 2: every fp32 3x3 conv whose shape fits, any plan; 0: never): conv3x3_sm_kernel
  int conv_splitk = 1;                        // "conv_splitk" (1: where estimated faster, inference plans only; n >= 2: force n slices in every plan; 0: never): F(2x2,3x3) layers with fewer (pixel tile, N tile) units than half the CUs split K over workgroups + a fixed-order second stage (small batches); n >= 2 forces n slices, 0 never
  int upsample_fold = 1;                      // "upsample_fold": a bilinear x2 whose only reader is the next conv's second source is folded into that conv's F(4x4,3x3) input transform (inference plans)
  int dgrad_wino = 1;                         // "dgrad_wino": 0 = direct 9-tap kernels for the backward's data-gradient convs (A/B: ~7 % slower cfg3 step, same gradients to the digit)
  int stem_wino = 2;                          // "stem_wino"
  int conv_persist = 1;                       // "conv_persist"
  int conv_c16 = 1;                           // "conv_c16"
  int dma_stagger = 1;                        // "conv_dma_stagger"
  int gemm_late_split = 0;                    // "gemm_late_split"
  int gemm_persist2 = 0;                      // "gemm_persist2"
  int fuse_gelu_fwd = 1;                      // Linear -> GELU op pair: one GEMM whose epilogue writes both tensors (0: two kernels)
  int fuse_gelu_bwd = 1;                      // Linear data gradient multiplies by GELU' in its epilogue (0: separate kernel)
  int wgrad_wino = 1;                         // "wgrad_wino": 3x3 weight gradients in the Winograd F(2x2,3x3) domain (layers with >= 32 padded channels on both sides)
  int dw_ln_fuse = 1;                         // "dw_ln_fuse"
  int head_fuse = 1;                          // "head_fuse"
  int pool_peephole = 1;                      // "pool_peephole"
  int mask_fold = 1;                          // "mask_fold": max-pool backward applies the ReLU mask of the conv it closes (no separate mask pass there)
  int wgrad_rows = 0;                         // 3x3 weight gradients of wide layers as nine row-wgrad GEMMs (off by default: measured slower than the 32x32-tile kernel; 1 auto, 2 always)
  double gemm_fill_threshold_wino2d = 0.4;    // ... and when it is the F(2x2,3x3) kernel (fill counted on its 16x16-pixel tiles)
  double gemm_fill_threshold_wino = 0.5;      // ... and the (lower) break-even when the halo kernel is the Winograd one
  double gemm_fill_threshold = 0.8;           // 3x3 convs whose maps fill the 16x32 tiles less than this run as row GEMMs
  int workspace_reuse = 0;                    // "workspace_reuse": 1 = slots of an inference program share memory by lifetime (no read-back, no backward)
  int convt_one_launch = 1;                   // "convt_one_launch": the four output-phase GEMMs of a transposed conv as ONE launch (grid.y = phase) instead of four (four launch floors at small batches)
  int convt_phase = 1;                        // "convt_phase": transposed convs as four phase GEMMs (0: zero-stuffing + 3x3 conv, 4x the FLOPs; A/B)
  int conv_precision = 0;                     // "conv_precision": 0 exact fp32 MFMA; 1 split-fp16 MFMA (22-bit products, fp32 accumulate); 2 plain fp16 (autocast-equivalent)
};


namespace ph {
int build_plan(const ph_model* m, int B, int H, int W, Plan& plan, int fmt = FMT_F32);
int forward_format(const ph_model* m);
int ensure_f16_weights(ph_model* m, int plain, hipStream_t s);  // format the forward of this program runs in under the handle's conv_precision
int upload(ph_model* m, const std::vector<float>& host, float** dev);
int upload_ints(ph_model* m, const std::vector<int>& host, int** dev);
int choose_bn(int coutp);
void apply_conv_options(const ph_model* m, ConvArgs& a);
}  // namespace ph
