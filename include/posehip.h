/*
 * posehip.h -- C ABI of libposehip.so, the MI355X (gfx950) hot path that replaces the
 * ATen/SciPy arithmetic behind sleap-nn's ModelBackend / inference ops seam.
 *
 * Every entry point is `extern "C"`, takes plain pointers + sizes + a hipStream_t passed
 * as void*, returns 0 on success or a negative PH_E_* code (never throws); the text of
 * the last error of the calling thread is available from ph_last_error().
 * Device pointers ("dev") must be HIP device memory of the current device; "host"
 * pointers are ordinary host memory.  The caller owns all inputs and outputs.
 *
 * Which reference interface each group replaces (paths relative to talmolab/sleap-nn):
 *   ph_model_*          sleap_nn/inference/layers/backends/torch_backend.py:113-153
 *                       (TorchBackend.__call__) -> training/lightning_modules.py:1840-1848
 *                       (squeeze + normalize_on_gpu) -> architectures/model.py:237-261
 *                       (Model.forward) -> architectures/unet.py:260-299 and
 *                       encoder_decoder.py:130-141,318-336,522-558,705-730.
 *   ph_local_peaks      inference/ops/peaks.py:184-259 (find_local_peaks[_rough]) with
 *                       ops/crops.py:31-124 + data/instance_cropping.py:129-171.
 *   ph_global_peaks     inference/ops/peaks.py:89-181 (find_global_peaks[_rough]).
 *   ph_paf_score        inference/ops/paf.py:84-497 (get_connection_candidates,
 *                       make_line_subs, get_paf_lines, score_paf_lines[_batch]) and
 *                       inference/utils.py:29-130 (interp1d).
 *   ph_lsap             scipy.optimize.linear_sum_assignment as called at
 *                       inference/ops/paf.py:589.
 *   ph_group_batch      inference/ops/paf.py:500-1149 (match_candidates_*,
 *                       assign_connections_to_instances, make_predicted_instances,
 *                       group_instances_*) + inference/streaming.py:147-255 padding.
 *   ph_toposort_edges   inference/ops/paf.py:890-912.
 *   ph_group_packed     inference/layers/bottomup.py:126-195 (hand-off) + inference/streaming.py:147-255.
 *   ph_centroid_select  inference/layers/centroid.py:195-261 + layers/topdown.py:183-235.
 *   ph_topdown_scatter  inference/layers/topdown.py:236-260.
 */
#ifndef POSEHIP_H
#define POSEHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PH_VERSION 108

/* error codes */
#define PH_OK 0
#define PH_E_INVALID -1     /* bad argument / unsupported configuration            */
#define PH_E_HIP -2         /* a HIP runtime call failed                           */
#define PH_E_CAPACITY -3    /* caller-provided output capacity too small           */
#define PH_E_INFEASIBLE -4  /* assignment problem has no finite-cost solution      */
#define PH_E_WORKSPACE -5   /* workspace too small                                 */

const char* ph_last_error(void);
int ph_version(void);

/* ------------------------------------------------------------------------------------
 * Network (encoder-decoder forward).  The host describes the network as a flat program of
 * ops over numbered activation slots; weights are handed over in the reference's own
 * state_dict layout (Conv2d: OIHW, ConvTranspose2d: IOHW, fp32, host memory) and are
 * re-packed for the MFMA kernels at creation.
 * ---------------------------------------------------------------------------------- */

enum ph_op_kind {
  PH_OP_INPUT_CONV = 1, /* uint8|float NCHW image -> /255 -> conv kxk "same" + bias + ReLU -> NHWC slot */
  PH_OP_CONV = 2,       /* conv kxk "same" over concat(src0, src1) + bias (+ReLU) -> NHWC slot          */
  PH_OP_POOL = 3,       /* 2x2/2 max pool, zero pad bottom/right when odd (architectures/common.py)     */
  PH_OP_UPSAMPLE = 4,   /* bilinear x2, align_corners=False (encoder_decoder.py:431-435)                */
  PH_OP_CONVT = 5,      /* ConvTranspose2d(k3,s2,p1,op1) + bias [-> folded BatchNorm: * weight2[c] + bias2[c]] (+ReLU | SiLU)
                           (encoder_decoder.py:439-461; the reference's decoder builds it with batch_norm=False, ReLU) */
  PH_OP_HEAD = 6,       /* 1x1 conv + bias (+sigmoid) -> NCHW fp32 output #out_index (heads.py:58-67)   */
  PH_OP_STEM = 7,       /* fused first encoder block: image -> /255 -> conv3x3+ReLU (weight/bias) ->
                           conv3x3+ReLU (weight2/bias2, cout <= 16) -> [full-res NHWC slot dst, if >= 0]
                           -> 2x2 max pool -> NHWC slot dst2.  One launch, the two full-resolution
                           activations never touch HBM unless dst >= 0.                               */
  /* ConvNeXt encoder (architectures/convnext.py:19-130; CNBlock / LayerNorm2d are torchvision's) */
  PH_OP_PATCH_STEM = 8, /* image -> /255 -> Conv2d(k = ksize, stride = cmid, padding 1) + bias -> NHWC slot
                           (convnext.py:73-84; the LayerNorm2d that follows is a PH_OP_LAYERNORM)       */
  PH_OP_DWCONV = 9,     /* depthwise 7x7 "same" conv + bias, weight (C,1,7,7)  (CNBlock.block[0])       */
  PH_OP_LAYERNORM = 10, /* LayerNorm over channels, eps 1e-6, weight/bias = affine (LayerNorm2d, block[2]) */
  PH_OP_LINEAR = 11,    /* per-pixel Linear(cin0 -> cout), weight (cout, cin0) + bias; PH_FLAG_GELU: erf-GELU
                           epilogue (block[3..4]); PH_FLAG_SCALE_RESIDUAL: dst = weight2[c] * (acc + bias) +
                           src1 (block[5], layer_scale, residual add)                                    */
  PH_OP_PATCH_CONV = 12, /* Conv2d(k2, s2) + bias, weight (cout, cin0, 2, 2)  (convnext.py:101-110)     */
  /* the two epilogues of PH_OP_LINEAR as ops of their own: the training program keeps the
     pre-activation tensors that autograd needs (GELU input, un-scaled block output)              */
  PH_OP_GELU = 13,      /* dst = GELU_erf(src0)                                                          */
  PH_OP_SCALE_ADD = 14, /* dst = weight[c] * src0 + src1   (layer_scale * block(x) + x)                  */
  PH_OP_GLOBAL_MAXPOOL = 15 /* dst (1x1) = max over H x W of src0 (nn.AdaptiveMaxPool2d(1), heads.py:519-520) */
};

#define PH_FLAG_RELU 1
#define PH_FLAG_SIGMOID 2
#define PH_FLAG_GELU 4
#define PH_FLAG_SCALE_RESIDUAL 8
#define PH_FLAG_SOFTMAX 16 /* PH_OP_HEAD: softmax over the output channels (ClassVectorsHead, heads.py:536-537) */
#define PH_FLAG_SILU 32    /* PH_OP_CONVT: SiLU instead of ReLU (the activation is an epilogue parameter of the phase GEMMs) */

typedef struct ph_op_desc {
  int32_t kind;      /* enum ph_op_kind                                              */
  int32_t src0;      /* input slot (-1 = the network input image)                    */
  int32_t src1;      /* second concat source slot or -1                              */
  int32_t dst;       /* output slot (PH_OP_HEAD: ignored)                            */
  int32_t cin0;      /* logical channels of src0                                     */
  int32_t cin1;      /* logical channels of src1 (0 if none)                         */
  int32_t cout;      /* logical output channels                                      */
  int32_t ksize;     /* kernel size (PH_OP_CONV / INPUT_CONV: odd, <= 7)             */
  int32_t flags;     /* PH_FLAG_*                                                    */
  int32_t weight;    /* index into the weights[] array of ph_model_create, or -1     */
  int32_t bias;      /* index into the weights[] array, or -1                        */
  int32_t out_index; /* PH_OP_HEAD: which output pointer receives the result         */
  int32_t dst2;      /* PH_OP_STEM: pooled output slot; PH_OP_CONV (ReLU): optional fused 2x2 max-pool slot, -1 = none */
  int32_t weight2;   /* PH_OP_STEM: second conv weight index; PH_OP_LINEAR: layer_scale index; PH_OP_CONVT: folded-BN scale or -1 */
  int32_t bias2;     /* PH_OP_STEM: second conv bias index; PH_OP_CONVT: folded-BN shift or -1 */
  int32_t cmid;      /* PH_OP_STEM: channels between the two convs (<= 16); PH_OP_PATCH_STEM: stride */
} ph_op_desc;

/* sizeof(ph_op_desc) as the library was compiled: a binding checks its own struct against it
 * before handing arrays to ph_model_create (16 int32 fields = 64 bytes in PH_VERSION 101). */
int32_t ph_op_desc_size(void);

typedef struct ph_model ph_model;

/* Create a model on the current HIP device.  `weights[i]` are host fp32 arrays (layouts
 * above); they are copied, the caller may free them afterwards.  Returns NULL on error. */
ph_model* ph_model_create(const ph_op_desc* ops, int32_t n_ops, const float* const* weights,
                          const int64_t* weight_numel, int32_t n_weights, int32_t n_slots,
                          int32_t n_outputs);
void ph_model_destroy(ph_model* m);

/* Bytes of device workspace ph_model_forward needs for a (B, C, H, W) input. */
int64_t ph_model_workspace_bytes(const ph_model* m, int32_t batch, int32_t height, int32_t width);

/* Shape of output #i for an (H, W) input: writes channels/height/width. */
int ph_model_output_shape(const ph_model* m, int32_t out_index, int32_t height, int32_t width,
                          int32_t* c, int32_t* h, int32_t* w);

/* Forward.  input_dev: (B, C, H, W) NCHW, dtype 0 = uint8 (divided by 255),
 * 1 = float32 already in [0,1], 2 = float32 in [0,255] (divided by 255)
 * (data/normalization.py:7-35; the data-dependent max()>1 test is resolved by the caller).
 * out_dev[i]: (B, c_i, h_i, w_i) NCHW fp32.  Everything is enqueued on `stream`. */
int ph_model_forward(ph_model* m, const void* input_dev, int32_t in_dtype, int32_t batch,
                     int32_t in_channels, int32_t height, int32_t width, void* workspace_dev,
                     int64_t workspace_bytes, float* const* out_dev, void* stream);

/* ------------------------------------------------------------------------------------
 * Training step pieces (forward = ph_model_forward on the UNFUSED program with bilinear
 * up-sampling; reference: training/lightning_modules.py:1850-1922, training/losses.py:8-63,
 * torch.optim.Adam as configured at lightning_modules.py:752-763).
 * The "canonical parameter arena" is weights[] of ph_model_create concatenated in order
 * (each in the reference's state_dict layout), fp32, on the device.
 * ---------------------------------------------------------------------------------- */
int64_t ph_model_num_params(const ph_model* m);

/* Rebuild every packed weight buffer of the model from the canonical arena (device gather). */
int ph_model_set_params(ph_model* m, const float* params_flat_dev, void* stream);

int64_t ph_model_backward_workspace_bytes(const ph_model* m, int32_t batch, int32_t height, int32_t width);

/* Loss + backward of the last ph_model_forward (same input, shapes and activation workspace).
 *   head_out_dev[i] / target_dev[i]: (B, c_i, h_i, w_i) NCHW fp32 predictions and targets.
 *   loss = sum_i loss_weights[i] * (MSE_i [+ OHKM_i]);  loss_dev: float[1 + n_outputs] = {total, per head}.
 *   sample_weights_dev: NULL (plain nn.MSELoss) or float[B] on the device: MSE_i becomes mean_b(w_b * mean_chw(diff^2)),
 *   the negative-sample weighting of training/lightning_modules.py:526-545 (w_b = negative_loss_weight for negative
 *   frames, 1 otherwise; train stage only -- the caller passes NULL for validation).  OHKM is not weighted (as there).
 *   grads_flat_dev: d loss / d params in the canonical arena layout (every entry is written).
 * Deterministic (fixed-order reductions, no float atomics). */
int ph_model_backward(ph_model* m, const void* input_dev, int32_t in_dtype, int32_t batch, int32_t in_channels,
                      int32_t height, int32_t width, const void* act_workspace_dev, void* grad_workspace_dev,
                      int64_t grad_workspace_bytes, const float* const* head_out_dev, const float* const* target_dev,
                      const float* loss_weights_host, const float* sample_weights_dev, int32_t ohkm_enabled, float hard_to_easy_ratio,
                      int32_t min_hard_keypoints, int32_t max_hard_keypoints, float ohkm_loss_scale,
                      float* loss_dev, float* grads_flat_dev, void* stream);

/* Two gradient buckets for overlapping the data-parallel all-reduce with the backward sweep (DDP's bucketing,
 * training/model_trainer.py:1751-1813 runs the reference under Lightning's DDP strategy).  The sweep runs heads -> decoder ->
 * middle -> encoder and the arena is in program order, so its tail becomes final first: ph_model_grad_bucket_split returns the
 * offset B (closest to the middle at which the arena splits cleanly between two ops; n_params if it never does), and an event
 * handed to ph_model_set_bucket_event (a hipEvent_t; NULL = off) is recorded on the backward's stream as soon as every gradient
 * in [B, n_params) is final -- a collective on another stream can wait on it while the rest of the backward still runs. */
int64_t ph_model_grad_bucket_split(const ph_model* m);
int ph_model_set_bucket_event(ph_model* m, void* hip_event);

/* ------------------------------------------------------------------------------------
 * Data-parallel gradient exchange over RCCL (what Lightning's DDP strategy does for the reference: training/model_trainer.py:1751-1813,
 * docs/guides/multi-gpu.md:67-81).  One process per GPU.  librccl is opened at run time (dlopen): ph_comm_available() says whether it could be.
 *   ph_comm_unique_id   rank 0 draws 128 opaque bytes (ncclGetUniqueId) and hands them to every rank over the host language's own channel
 *   ph_comm_create      collective over the ranks, on the CURRENT HIP device (ncclCommInitRank); NULL + ph_last_error() on failure
 *   ph_allreduce        in-place SUM of `count` floats over the ranks, enqueued on `stream`
 *   ph_model_set_comm   ph_model_backward then exchanges the gradient arena itself: the tail bucket [ph_model_grad_bucket_split, n_params) on
 *                       `comm_stream` as soon as the sweep has finished it, the head bucket behind the sweep; the stream ph_model_backward runs on
 *                       waits for both, so the ph_adam_step enqueued next (grad_scale = 1 / world: DDP's mean) reads the sum.  comm = NULL: off.
 * ---------------------------------------------------------------------------------- */
typedef struct ph_comm ph_comm;
int ph_comm_available(void);
int ph_comm_unique_id(void* out_id128);
ph_comm* ph_comm_create(const void* id128, int32_t world, int32_t rank);
void ph_comm_destroy(ph_comm* comm);
int32_t ph_comm_world(const ph_comm* comm);
int ph_allreduce(ph_comm* comm, float* buf_dev, int64_t count, void* stream);
int ph_model_set_comm(ph_model* m, ph_comm* comm, void* comm_stream);

/* torch.optim.Adam step (weight_decay = 0) on flat device arrays; max_exp_avg_sq_dev != NULL enables
 * amsgrad.  grad_scale multiplies the gradient first (1/world_size after a sum all-reduce). */
int ph_adam_step(float* params_dev, const float* grads_dev, float* exp_avg_dev, float* exp_avg_sq_dev,
                 float* max_exp_avg_sq_dev, int64_t n, float lr, float beta1, float beta2, float eps,
                 int32_t step, float grad_scale, void* stream);
/* torch.optim.AdamW step (the reference's other optimizer choice, lightning_modules.py:752-755): decoupled weight decay
 * param *= 1 - lr * weight_decay, then the Adam update above (torch's default weight_decay is 0.01). */
int ph_adamw_step(float* params_dev, const float* grads_dev, float* exp_avg_dev, float* exp_avg_sq_dev,
                  float* max_exp_avg_sq_dev, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int32_t step, float grad_scale, void* stream);

/* On-device rendering of the training targets (reference: data/confidence_maps.py:96-166
 * generate_multiconfmaps; data/edge_maps.py:15-78,120-220,250-323 generate_pafs).
 * points_dev: (B, I, N, 2) fp32 (x, y) image coordinates, NaN = missing.  Grids are
 * arange(0, size, stride).  Confmaps: (B, N, h, w), Gaussian sigma*stride, max over instances.
 * PAFs: (B, 2E, h, w) channel 2e = x, 2e+1 = y, summed over instances; edges_dev int32 (E, 2). */
int ph_render_confmaps(const float* points_dev, int32_t B, int32_t I, int32_t N, int32_t img_h, int32_t img_w,
                       int32_t stride, float sigma, float* out_dev, void* stream);
int ph_render_pafs(const float* points_dev, const int32_t* edges_dev, int32_t B, int32_t I, int32_t N, int32_t E,
                   int32_t img_h, int32_t img_w, int32_t stride, float sigma, float* out_dev, void* stream);

/* Diagnostic / test hook (pure host arithmetic, no GPU needed): the split-K plan the 3x3 kernels would take for a layer of this shape on a chip of n_cu CUs
 * under handle option conv_splitk = splitk (padded channel counts; section 4.1d of DESIGN.md).  out[0] = K slices on the F(2x2,3x3) kernel, out[1] = on the
 * F(4x4,3x3) kernel (1 = no split), out[2], out[3] = KiB of partial-sum scratch each would need (0 when it does not split). */
int ph_debug_split_plan(int32_t B, int32_t H, int32_t W, int32_t cin_padded, int32_t cout_padded, int32_t splitk, int32_t n_cu, int64_t* out4);

/* Diagnostic (tools/gemm_bench.py): average milliseconds of one row-GEMM kernel variant on synthetic operands. */
int ph_debug_gemm_bench(int32_t variant, int32_t M, int32_t K, int32_t N, int32_t mode, int32_t H, int32_t W,
                        int32_t act, int32_t iters, float* ms_out);

/* Diagnostic (tools/row_wgrad_bench.py): average milliseconds of the row weight-gradient GEMM dW[n][k] = sum_m dY[m][n] X[m][k]
 * (Linear layers of the training step, train_kernels.h launch_row_wgrad) on synthetic operands, and the largest error of 64
 * sampled entries against a float64 sum on the host, relative to the largest of them. */
int ph_debug_row_wgrad_bench(int32_t M, int32_t n, int32_t k, int32_t iters, float* ms_out, float* max_rel_err_out);

/* Per-handle options (the library reads no environment variables and keeps no process-global
 * tunables).  Every key selects between kernel variants that compute the same result; defaults
 * are the measured-best ones.  Keys: "conv_wino" (1 Winograd F(2,3) 3x3 kernels | 2 only the
 * N-tile-64 layers | 0 direct 9-tap kernels), "stem_wino", "dgrad_wino" (0: direct kernels for the backward's data-gradient convs), "conv_dma", "conv_dma32",
 * "conv_persist", "conv_c16", "conv_dma_stagger", "fuse_gelu_fwd", "fuse_gelu_bwd", "wgrad_rows", "convt_phase", "workspace_reuse" (1: activation slots of an inference forward share memory by lifetime),
 * "conv_precision" (0 exact fp32 MFMA | 1 split-fp16 MFMA, 22-bit products | 2 plain fp16, the reference's autocast mode),
 * "gemm_late_split", "gemm_persist2", "conv_gemm_fill", "conv_gemm_fill_wino", "conv_gemm_fill_wino2d",
 * "conv_wino2d" / "conv_w16" (the Winograd F(2x2,3x3) kernels), "conv_wino4" (K-heavy 3x3 convs on the Winograd F(4x4,3x3) kernel: 1 inference
 * plans | 2 every plan | 0 never), "conv_wino4_min_cin", "upsample_fold" (a bilinear x2 read only by the next conv's second source rides in that kernel's input transform), "head_fuse" (a 1x1 head computed in its producer conv's epilogue),
 * "conv_splitk" / "conv_splitk_finish" / "conv_n32_wino2d" / "conv_smallmap" (the small-batch routings: K split over workgroups, Cout-32 layers on the N-tile-64 kernels,
 * conv3x3_sm_kernel on (8 x 8 pixels, 16 channels) units for small maps: 1 inference plans where estimated faster | 2 wherever the shape fits | 0 never),
 * "pool_peephole" (unfused programs: a conv writes the next op's 2x2 max pool), "dw_ln_fuse" (ConvNeXt: LayerNorm inside the
 * depthwise / stem kernels), "wgrad_wino" (3x3 weight gradients in the Winograd domain), "mask_fold" (ReLU masks applied by the kernel
 * that completes a gradient), "conv_f16_rows" (plain fp16: conv3x3_f16_rows_kernel 1 where its plan is estimated faster | 2 wherever the shape fits | 0 never), "upsample_f16math"
 * (plain fp16: the bilinear x2 blended in packed fp16 arithmetic, folded or standalone: same bits), "stem_f16mfma" (plain fp16: stem_f16_kernel), "block_fuse" (plain fp16, inference plans:
 * the two convs of a 32-channel encoder block in one launch), "mlp_fuse" (inference plans: CNBlock's Linear + GELU + Linear + layer scale + residual in one launch at 96 / 192 channels).
 * Unknown key -> PH_E_INVALID. */
int ph_model_set_option(ph_model* m, const char* key, double value);
int ph_model_get_option(const ph_model* m, const char* key, double* value);

/* Per-op timing with HIP events recorded on the forward's own stream (used by bench.py for
 * the roofline object).  While enabled every forward records one event before each op and
 * one after the last; ph_model_profile_read waits for the last recorded forward, returns
 * the milliseconds accumulated per op since profiling was enabled and the number of
 * forwards they cover.  enabled: 1 = start (clears the accumulated times), 0 = stop recording
 * (what was accumulated stays readable), 2 = resume without clearing -- so a caller can sample
 * every n-th forward of a timed region instead of paying ~25 event records in each. */
int ph_model_set_profiling(ph_model* m, int32_t enabled);
int ph_model_profile_read(ph_model* m, double* op_ms, int32_t n_ops, int32_t* n_forwards);

/* Which kernel family each op of the LAST ph_model_forward ran (codes[i] for op i, PH_KV_*; 0 for ops that are not matrix
 * work or were folded into their producer).  bench.py prices a launch's executed MFMA FLOPs from this instead of restating
 * the dispatch rules: direct kernels execute the direct-convolution count 2*Cin*Cout*k*k*H*W, Winograd F(2,3) 2/3 of it,
 * F(2x2,3x3) 4/9, F(4x4,3x3) 1/4. */
#define PH_KV_NONE 0
#define PH_KV_DIRECT 1    /* 9-tap halo kernels on v_mfma_f32_32x32x2_f32 (conv3x3_mfma_*), conv3x3_c16 excluded */
#define PH_KV_WINO1D 2    /* conv3x3_wino_persist_kernel: Winograd F(2,3) along x                                */
#define PH_KV_WINO2D 3    /* conv3x3_wino2d_kernel: wave-split Winograd F(2x2,3x3)                               */
#define PH_KV_W16 4       /* conv3x3_w16_kernel: wave-private Winograd F(2x2,3x3) on v_mfma_f32_16x16x4_f32      */
#define PH_KV_C16 5       /* conv3x3_c16_kernel (16 -> 16 channels, direct)                                      */
#define PH_KV_ROWGEMM 6   /* gemm_mfma_dma_*_kernel (row GEMM: Linear, 2x2/s2, k x k taps, transposed-conv phases) */
#define PH_KV_WINO4 7     /* conv3x3_wino4_kernel: Winograd F(4x4,3x3)                                           */
#define PH_KV_F16 8       /* conv3x3_f16_persist_kernel (fp16 matrix pipe; split precision = 3 MFMAs per product) */
#define PH_KV_STEM 9      /* stem_fused_kernel (second conv on v_mfma_f32_16x16x4_f32; "stem_wino" picks its form) */
#define PH_KV_WINO2D_KS 10 /* conv3x3_wino2d_kernel, K split over workgroups + splitk_reduce_kernel (small batches)       */
#define PH_KV_SMALLMAP 12 /* conv3x3_sm_kernel: Winograd F(2x2,3x3) on 8 x 8-pixel x 16-channel units (small maps, small batches)      */
#define PH_KV_F16_ROWS 13 /* conv3x3_f16_rows_kernel: plain fp16 on v_mfma_f32_16x16x32_f16, row tiles, loader waves, folded bilinear x2 */
#define PH_KV_F16_BLOCK 14 /* block2_c32_f16_kernel: conv(<= 16 -> 32) + conv(32 -> 32) (+ pool) of an encoder block in one launch (plain fp16); the second conv reports PH_KV_FUSED */
#define PH_KV_MLP 15      /* cnblock_mlp_kernel: Linear(C, 4C) + GELU + Linear(4C, C) + layer scale + residual of a CNBlock in one launch (both Linear ops report it; the second has no launch of its own) */
#define PH_KV_FUSED 11    /* no launch of its own: a 1x1 head computed in the epilogue of the conv that produces its input */
int ph_model_last_kernels(const ph_model* m, int32_t* codes, int32_t n_ops);

/* Diagnostic: when buf_dev != NULL the conv3x3 kernels that carry a probe write into the record block of
 * their op -- 32768 x uint64 per op of the program, op i at buf_dev + 32768 i words (the buffer must hold
 * 32768 x n_ops words) -- e.g. {delta s_memtime, delta s_memrealtime} per workgroup (fp16 pipe; the F(4x4,3x3)
 * kernel in -DW4_CLOCK builds): the shader clock under load is d_memtime / d_memrealtime * 100 MHz
 * (tools/clockprobe_f16.py, tools/clockprobe_layers.py).  NULL turns it off (default). */
int ph_model_set_clock_probe(ph_model* m, void* buf_dev);

/* Debug/parity helper: copy activation slot `slot` of the last forward (NHWC, padded
 * channels) into an NCHW fp32 device buffer of the logical channel count. */
int ph_model_read_slot(ph_model* m, int32_t slot, float* out_dev, int64_t out_numel, void* stream);

/* ------------------------------------------------------------------------------------
 * Peak finding
 * ---------------------------------------------------------------------------------- */

/* Local maxima (strict > over the 8 neighbours, -inf outside) above `threshold`, emitted in
 * the reference's (sample, y, x, channel) order, optionally refined by the integral
 * (first-moment) offset over a patch x patch zero-padded window (refine != 0).
 * cms_dev: (B, C, H, W) fp32.  Outputs (device, capacity `cap` rows):
 *   out_xy (cap,2) f32 [x,y]; out_val (cap) f32; out_sample (cap) i32; out_channel (cap) i32
 *   out_count: device int32[2 + 2B]: [0] = total peaks found (may exceed cap: then only the
 *   first cap rows are valid and the caller should retry), [1+b] = peaks of sample b,
 *   [1+B+b] = exclusive offset of sample b (b = 0..B, last entry = total).
 * xy_scale: out_xy = refined (x, y) * xy_scale in one fp32 multiply -- 1 for find_local_peaks itself, the confidence
 *   maps' output stride for callers that go on in image coordinates (`peaks * cms_output_stride`, layers/bottomup.py:111).
 * scratch_dev: >= 4*(2*B*H + 2) bytes runs the three-pass kernels (count rows, scan, emit: the maps are read twice); with
 *   ph_local_peaks_scratch_bytes(B, C, H, W) bytes the maps are read ONCE (one block per sample and eight rows finds, refines and stages its peaks in order;
 *   a placement pass over the staged peaks puts them at their final offsets) -- same outputs bit for bit, two launches, no atomics. */
int ph_local_peaks(const float* cms_dev, int32_t B, int32_t C, int32_t H, int32_t W,
                   float threshold, int32_t refine, int32_t patch, float* out_xy, float* out_val,
                   int32_t* out_sample, int32_t* out_channel, int32_t* out_count, int32_t cap,
                   float xy_scale, void* scratch_dev, int64_t scratch_bytes, void* stream);

int64_t ph_local_peaks_scratch_bytes(int32_t B, int32_t C, int32_t H, int32_t W);

/* Global peak per (sample, channel): value = max; x = first column containing the max,
 * y = first row containing the max (independent, as the reference); below `threshold`
 * -> NaN coords and value 0.  out_xy (B,C,2), out_val (B,C). */
int ph_global_peaks(const float* cms_dev, int32_t B, int32_t C, int32_t H, int32_t W,
                    float threshold, int32_t refine, int32_t patch, float* out_xy, float* out_val,
                    void* stream);

/* Crop gather (top-down stage 2 pick-up; inference/ops/crops.py:31-124): n crops of
 * crop_h x crop_w from (B, C, H, W) images (dtype 0 = uint8, 1 = float32), zero outside the image.
 * topleft_xy_dev: (n, 2) float32 bbox top-left corners (x, y) as produced by make_centered_bboxes;
 * the integer origin is trunc(tl + half) - half with half = (crop_w/2, crop_h/2) (crops.py:85-90).
 * out_dev: (n, C, crop_h, crop_w), same dtype as the images. */
int ph_crop_bboxes(const void* images_dev, int32_t dtype, int32_t B, int32_t C, int32_t H, int32_t W,
                   const float* topleft_xy_dev, const int32_t* sample_inds_dev, int32_t n,
                   int32_t crop_h, int32_t crop_w, void* out_dev, void* stream);

/* Top-down glue, device side.  ph_centroid_select = CentroidLayer.postprocess after the peak finding (inference/layers/centroid.py:195-261:
 * per frame keep the peaks in order, or the max_instances largest values in descending order (torch.topk) when there are more; NaN-pad to
 * (B, max_instances, 2) / (B, max_instances); undo input scale and eff_scale (inference/ops/coord.py:40-70)) plus what TopDownLayer.predict
 * (inference/layers/topdown.py:127-150, 183-267) derives from the valid centroids IN SIZED SPACE (centroid * eff_scale: the crops are cut from the
 * sizematched frame): make_centered_bboxes (data/instance_cropping.py:129-171) around the sized centroid -- out_bboxes (B, max_instances, 4, 2; NaN where
 * empty) holds those boxes / eff_scale (image space, as the reference stores them) --, and the stage-2 lists in torch.nonzero order of the valid mask
 * (frame, then slot): list_sample int32[n_valid], list_topleft float[n_valid, 2] (sized space: the ph_crop_bboxes inputs), list_slot int32[n_valid] = frame * max_instances + slot,
 * pos_of_slot int32[B * max_instances] = list position or -1, out_n_valid int32[1].  peaks / counts are ph_local_peaks' outputs (coordinates
 * already multiplied by the output stride through its xy_scale).  eff_scale_dev float[B] or NULL; every list pointer may be NULL (centroid-only use). */
int ph_centroid_select(const float* peaks_xy_dev, const float* peak_vals_dev, const int32_t* counts_dev, int32_t B, int32_t max_instances,
                       int32_t cap, float input_scale, const float* eff_scale_dev, float crop_h, float crop_w, float* out_centroids_dev,
                       float* out_vals_dev, float* out_bboxes_dev, int32_t* list_sample_dev, float* list_topleft_dev, int32_t* list_slot_dev,
                       int32_t* pos_of_slot_dev, int32_t* out_n_valid_dev, void* stream);

/* ... and the way back (topdown.py:236-267): crop-local keypoints (n_valid, n_nodes, 2) / values (n_valid, n_nodes) of stage 2 into
 * out_keypoints = (crop keypoints + the crop's top-left (add_crop_offset, ops/coord.py:73-90)) / eff_scale of the slot's frame (sized space ->
 * image space; eff_scale_dev float[slots / max_instances] or NULL = 1), out_crop_keypoints, out_vals, all (slots = B * max_instances,
 * n_nodes[, 2]) and NaN where pos_of_slot is -1. */
int ph_topdown_scatter(const float* crop_xy_dev, const float* crop_vals_dev, const float* list_topleft_dev, const int32_t* pos_of_slot_dev,
                       int32_t slots, int32_t n_nodes, const float* eff_scale_dev, int32_t max_instances, float* out_keypoints_dev,
                       float* out_crop_keypoints_dev, float* out_vals_dev, void* stream);

/* Antialiased bilinear resize of planes x H x W -> planes x OH x OW (uint8: dtype 0, float32: dtype 1), NCHW planes.
 * Replaces torchvision.transforms.v2.functional.resize as called by resize_image (data/resizing.py:70-84, the input-scale step)
 * and apply_sizematcher (data/resizing.py:136-175): for tensors that is torch's interpolate(mode="bilinear",
 * align_corners=False, antialias=True).  uint8 results are bit-exact with the CPU operator (int16 fixed-point weights,
 * horizontal pass first into a uint8 intermediate); float32 matches to fp32 rounding.  tmp_dev: planes * H * OW elements of the
 * same dtype (only read/written when both axes change; may be NULL otherwise). */
int ph_resize_bilinear_aa(const void* src_dev, int32_t dtype, int32_t planes, int32_t H, int32_t W,
                          void* dst_dev, int32_t OH, int32_t OW, void* tmp_dev, void* stream);

/* Class-map sampling for multi-class bottom-up (inference/ops/identity.py:86-101): for each peak
 * gather the K class probabilities at (round-half-even(y), round-half-even(x)) clamped to the map.
 * class_maps_dev (B, K, H, W) fp32; peaks_xy_dev (n, 2) in class-map pixels; out_probs_dev (n, K). */
int ph_sample_class_maps(const float* class_maps_dev, int32_t B, int32_t K, int32_t H, int32_t W,
                         const float* peaks_xy_dev, const int32_t* sample_inds_dev, int32_t n,
                         float* out_probs_dev, void* stream);

/* ------------------------------------------------------------------------------------
 * PAF line-integral scoring (device)
 * ---------------------------------------------------------------------------------- */

/* pafs_dev: (B, 2E, H, W) NCHW fp32 (the head output; the reference's permute to
 * (B,H,W,2E) is folded into the indexing).  peaks are the ph_local_peaks outputs ALREADY
 * multiplied by the confmap stride (image-space x,y), grouped by sample:
 * peak_offsets_dev int32[B+1].  edges_dev int32[E*2] (src node, dst node).
 * t_dev: float[n_points] = torch.linspace(0,1,n_points) values.
 * Outputs (device): cand_edge (cap) i32, cand_src/cand_dst (cap) i32 = peak indices LOCAL
 * to the sample, cand_score (cap) f32, cand_offsets int32[B+1] (+ total in [B]).
 * Candidates of one sample are ordered by edge, then src peak, then dst peak (ascending).
 * scratch_dev: >= 4*(n_peaks_total + B*(n_nodes+1) + B*(E+1) + 16) bytes. */
int ph_paf_score(const float* pafs_dev, int32_t B, int32_t E2, int32_t H, int32_t W,
                 const float* peaks_xy_dev, const int32_t* peak_channel_dev,
                 const int32_t* peak_offsets_dev, int32_t n_peaks_total, int32_t n_nodes,
                 const int32_t* edges_dev, int32_t n_edges, const float* t_dev, int32_t n_points,
                 int32_t pafs_stride, float max_edge_length, float dist_penalty_weight,
                 int32_t* cand_edge, int32_t* cand_src, int32_t* cand_dst, float* cand_score,
                 int32_t* cand_offsets, int32_t cap, void* scratch_dev, int64_t scratch_bytes,
                 void* stream);

/* ------------------------------------------------------------------------------------
 * Host (CPU) stage: assignment + instance assembly.  Pure host code, no HIP calls.
 * ---------------------------------------------------------------------------------- */

/* Rectangular linear sum assignment (minimise), row-major cost (nr x nc) doubles; +inf =
 * forbidden.  Writes min(nr,nc) pairs sorted by row.  Tie behaviour follows SciPy's
 * rectangular_lsap (shortest augmenting path).  PH_E_INFEASIBLE if no finite solution. */
int ph_lsap(const double* cost, int32_t nr, int32_t nc, int32_t* rows, int32_t* cols);

/* BFS edge order from the topological root; out_order has room for n_edges entries;
 * returns the number of entries written (>=0) or a negative error. */
int ph_toposort_edges(const int32_t* edges, int32_t n_edges, int32_t* out_order);

/* Match + assemble a batch (all host arrays).  Inputs mirror ScoredBatch
 * (inference/streaming.py:43-112) in flattened form.  Outputs are NaN-padded:
 *   out_kpts (B, max_inst, n_nodes, 2), out_vals (B, max_inst, n_nodes), out_scores (B, max_inst),
 *   out_n_inst int32[B] = instances found per sample before truncation.
 * If an instance count exceeds max_inst: truncate_by_score != 0 keeps the top max_inst by
 * instance score (numpy argsort()[::-1] order), else keeps the first max_inst. */
int ph_group_batch(int32_t B, int32_t n_nodes, const int32_t* edges, int32_t n_edges,
                   const float* peaks_xy, const float* peak_vals, const int32_t* peak_channel,
                   const int32_t* peak_offsets, const int32_t* cand_edge, const int32_t* cand_src,
                   const int32_t* cand_dst, const float* cand_score, const int32_t* cand_offsets,
                   float min_line_score, double min_instance_peaks, int32_t min_instance_peaks_is_fraction,
                   int32_t max_inst, int32_t truncate_by_score, float* out_kpts, float* out_vals,
                   float* out_scores, int32_t* out_n_inst);

/* The whole CPU stage of a bottom-up batch from the packed D2H arena of the GPU stage, in one call: unpack + capacity check + the
 * max_peaks_per_node guard (inference/layers/bottomup.py:126-161), ph_group_batch, then the input-scale / eff_scale undo and the output sizing of
 * group_scored_batch (inference/streaming.py:147-255).  arena (host, pinned or not) = float32 / int32 words
 *   [counts 2+2B (ph_local_peaks' out_count) | cand offsets B+1 | xy 2*peak_cap | vals peak_cap | cand score cand_cap | channel peak_cap |
 *    cand edge cand_cap | cand src cand_cap | cand dst cand_cap].
 * max_instances < 0 = None (keep every instance); max_peaks_per_node < 0 = no guard; eff_scale float[B] or NULL.
 * Outputs (B, out_cap, n_nodes, 2) / (B, out_cap, n_nodes) / (B, out_cap) NaN-padded (row stride out_cap, or max(1, max_instances) when that is given
 * and <= out_cap), out_n_inst int32[B].  status int32[4]: [0] peaks, [1] candidates, [2] flags -- 1: arena capacity exceeded (nothing grouped: re-run
 * the GPU stage with at least status[0] / status[1] entries), 2: out_cap too small (nothing grouped: come back with status[3]), 4: guard fired (all NaN)
 * --, [3] instances per frame the caller should keep (max over frames, >= 1; max_instances when given). */
int ph_group_packed(const float* arena, int32_t B, int32_t n_nodes, int32_t peak_cap, int32_t cand_cap, const int32_t* edges, int32_t n_edges,
                    float min_line_score, double min_instance_peaks, int32_t min_instance_peaks_is_fraction, int32_t max_instances,
                    int32_t max_peaks_per_node, float input_scale, const float* eff_scale, int32_t out_cap, float* out_kpts, float* out_vals,
                    float* out_scores, int32_t* out_n_inst, int32_t* status);

/* Host: Hungarian matching of peaks to classes per (sample, channel)
 * (inference/ops/identity.py:13-76).  probs (n, K) fp32, sample/channel (n) int32.
 * Writes matched (peak index, class index) pairs that also are the peak's arg-max class
 * (is_best filter, identity.py:61-66) in (sample, channel) order; returns the pair count. */
int ph_group_class_peaks(const float* probs, const int32_t* sample_inds, const int32_t* channel_inds,
                         int32_t n, int32_t n_samples, int32_t n_channels, int32_t K,
                         int32_t* out_peak_inds, int32_t* out_class_inds);

#ifdef __cplusplus
}
#endif
#endif /* POSEHIP_H */
