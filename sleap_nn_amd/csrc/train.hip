// Backward pass + parameter management of the network runtime (C ABI: ph_model_backward,
// ph_model_set_params, ph_model_num_params).  Together with ph_model_forward and ph_adam_step this
// is one training step of the reference's LightningModule.training_step
// (sleap_nn/training/lightning_modules.py:1850-1922: forward, per-head MSE (+OHKM), weighted sum,
// autograd backward) for UNet models with bilinear up-sampling; gradients come out in the
// reference's state_dict layout (OIHW arena) so that a flat RCCL all-reduce + Adam apply directly.
#include <algorithm>

#include "model_internal.h"
#include "train_kernels.h"

using namespace ph;

namespace {

struct BwdPlan {
  Plan act;                         // activation slots (same layout as the forward workspace)
  std::vector<int64_t> head_dy_off; // per output: NCHW dY buffer offset in the grad workspace
  int64_t scratch_off = 0, scratch_bytes = 0, total = 0;
};

int build_bwd_plan(const ph_model* m, int B, int H, int W, BwdPlan& bp) {
  int rc = build_plan(m, B, H, W, bp.act);
  if (rc != PH_OK) return rc;
  int64_t off = bp.act.total;  // gradient slots mirror the activation slots one to one
  bp.head_dy_off.assign(m->n_outputs, -1);
  int64_t scratch = 0;
  for (const PackedOp& op : m->ops) {
    const ph_op_desc& d = op.d;
    if (d.kind == PH_OP_STEM || (d.kind == PH_OP_CONV && d.dst2 >= 0)) {
      set_error("backward needs the unfused program (no stem / conv+pool fusion)");
      return PH_E_INVALID;
    }
    if (d.kind == PH_OP_CONV && d.ksize != 3 && !op.wdk_gemm_dev[0]) {
      set_error("backward of a %d x %d conv needs its data-gradient GEMM weights (model built by an older ph_model_create?)", d.ksize, d.ksize);
      return PH_E_INVALID;
    }
    if (d.kind == PH_OP_CONVT) {
      if (!op.wt_dgrad_dev || op.wt_scale_dev || (d.flags & PH_FLAG_SILU)) {
        set_error("backward of a transposed conv supports bias + ReLU (the reference's decoder); folded BatchNorm / SiLU are inference-only");
        return PH_E_INVALID;
      }
      const SlotShape& si = bp.act.slots[d.src0];
      scratch = std::max<int64_t>(scratch, row_wgrad_slab_floats(B * si.h * si.w, d.cin0, d.cout));
      scratch = std::max<int64_t>(scratch, bias_scratch_floats(pad16(d.cout)));
      continue;
    }
    if (d.kind == PH_OP_LINEAR && (d.flags & (PH_FLAG_GELU | PH_FLAG_SCALE_RESIDUAL))) {
      set_error("backward needs the unfused ConvNeXt program (GELU / layer-scale as ops of their own)");
      return PH_E_INVALID;
    }
    if (d.kind >= PH_OP_PATCH_STEM) {
      const SlotShape& so = bp.act.slots[d.dst];
      const int64_t npix = (int64_t)B * so.h * so.w;
      int64_t need = bias_scratch_floats(so.cp);
      switch (d.kind) {
        case PH_OP_PATCH_STEM: need = std::max(need, patch_stem_wgrad_scratch_floats(d.cin0, d.cout, d.ksize, npix)); break;
        case PH_OP_DWCONV: need = std::max(need, dwconv7_wgrad_scratch_floats(B, so.h, so.cp)); break;
        case PH_OP_LAYERNORM: need = std::max(need, 2 * npix + chan_reduce_scratch_floats(so.cp)); break;
        case PH_OP_LINEAR:
        case PH_OP_PATCH_CONV: need = std::max(need, row_wgrad_slab_floats((int)npix, d.cout, d.cin0)); break;
        case PH_OP_SCALE_ADD: need = std::max(need, chan_reduce_scratch_floats(so.cp)); break;
        default: break;
      }
      scratch = std::max(scratch, need);
      continue;
    }
    if (d.kind == PH_OP_HEAD) {
      const SlotShape& s0 = bp.act.slots[d.src0];
      bp.head_dy_off[d.out_index] = off;
      off += align_up((int64_t)B * d.cout * s0.h * s0.w * 4, 256);
      scratch = std::max<int64_t>(scratch, head_bwd_scratch_floats(s0.cp, d.cout, (int64_t)B * s0.h * s0.w));
      scratch = std::max<int64_t>(scratch, loss_scratch_floats(d.cout) + 64);
    } else if (d.kind == PH_OP_CONV) {
      const SlotShape& s0 = bp.act.slots[d.src0];
      scratch = std::max<int64_t>(scratch, wgrad_slab_floats(d.cin0, d.cout, B, s0.h, s0.w));
      if (d.cin1 > 0) scratch = std::max<int64_t>(scratch, wgrad_slab_floats(d.cin1, d.cout, B, s0.h, s0.w));
      scratch = std::max<int64_t>(scratch, row_wgrad_slab_floats(B * s0.h * s0.w, d.cout, std::max(d.cin0, d.cin1)));
      scratch = std::max<int64_t>(scratch, bias_scratch_floats(pad16(d.cout)));
    } else if (d.kind == PH_OP_INPUT_CONV) {
      if (d.ksize != 3) scratch = std::max<int64_t>(scratch, patch_stem_wgrad_scratch_floats(d.cin0, d.cout, d.ksize, (int64_t)B * H * W));  // im2col + one row-wgrad GEMM
      scratch = std::max<int64_t>(scratch, input_wgrad_scratch_floats(d.cin0, d.cout));
      scratch = std::max<int64_t>(scratch, bias_scratch_floats(pad16(d.cout)));
    }
  }
  bp.scratch_off = off;
  bp.scratch_bytes = align_up(scratch * 4, 256);
  bp.total = off + bp.scratch_bytes;
  return PH_OK;
}

}  // namespace

extern "C" {

int64_t ph_model_num_params(const ph_model* m) {
  if (!m) {
    set_error("ph_model_num_params: null model");
    return PH_E_INVALID;
  }
  return m->n_params;
}

int ph_model_set_params(ph_model* m, const float* params_flat_dev, void* stream) {
  PH_REQUIRE(m && params_flat_dev, "ph_model_set_params: null argument");
  if (!m->gather_table_dev && !m->packed.empty()) {  // built once: every packed buffer of the model is a segment of one launch
    std::vector<GatherSegment> seg;
    unsigned blocks = 0;
    for (const PackedBuffer& pb : m->packed) {
      if (pb.n == 0) continue;
      seg.push_back(GatherSegment{pb.map, pb.dst, pb.n, blocks});
      blocks += (unsigned)((pb.n + 1023) / 1024);
    }
    void* t = nullptr;
    PH_HIP_CHECK(hipMalloc(&t, seg.size() * sizeof(GatherSegment)));
    m->allocs.push_back(t);
    PH_HIP_CHECK(hipMemcpy(t, seg.data(), seg.size() * sizeof(GatherSegment), hipMemcpyHostToDevice));
    m->gather_table_dev = t;
    m->gather_segments = (int)seg.size();
    m->gather_blocks = blocks;
  }
  {
    int rc = launch_gather_multi(params_flat_dev, static_cast<const GatherSegment*>(m->gather_table_dev), m->gather_segments, m->gather_blocks, static_cast<hipStream_t>(stream));
    if (rc != PH_OK) return rc;
  }
  // derived buffers: the F(2,3) and F(2x2,3x3) transforms (most of them) each go in ONE table-driven launch, the rest one by one
  if (!m->pack_tables_built) {
    for (int kind = 0; kind < 2; ++kind) {
      std::vector<PackSegment> seg;
      unsigned blocks = 0;
      for (const DerivedBuffer& db : m->derived) {
        const bool mine = kind == 0 ? (db.kind == 0 && db.bn != 0) : db.kind == 2;
        if (!mine) continue;
        const unsigned long long total = (unsigned long long)(kind == 0 ? wino_pack_floats(db.panels, db.bn) : wino2d_pack_floats(db.panels, db.bn));
        if (total == 0) continue;
        seg.push_back(PackSegment{db.src, db.dst, total, db.bn, blocks});
        blocks += (unsigned)((total + 1023) / 1024);
      }
      void* t = nullptr;
      if (!seg.empty()) {
        PH_HIP_CHECK(hipMalloc(&t, seg.size() * sizeof(PackSegment)));
        m->allocs.push_back(t);
        PH_HIP_CHECK(hipMemcpy(t, seg.data(), seg.size() * sizeof(PackSegment), hipMemcpyHostToDevice));
      }
      m->pack_table_dev[kind] = t;
      m->pack_segments[kind] = (int)seg.size();
      m->pack_blocks[kind] = blocks;
    }
    m->pack_tables_built = true;
  }
  {
    int rc = launch_wino_pack_multi(static_cast<const PackSegment*>(m->pack_table_dev[0]), m->pack_segments[0], m->pack_blocks[0], static_cast<hipStream_t>(stream));
    if (rc == PH_OK) rc = launch_wino2d_pack_multi(static_cast<const PackSegment*>(m->pack_table_dev[1]), m->pack_segments[1], m->pack_blocks[1], static_cast<hipStream_t>(stream));
    if (rc != PH_OK) return rc;
  }
  for (const DerivedBuffer& db : m->derived) {
    if ((db.kind == 0 && db.bn != 0) || db.kind == 2) continue;  // in the two launches above
    if (db.kind == 5 || db.kind == 6) {  // F(4x4,3x3) / small-map weights: only inference plans read them (conv_wino4 = conv_smallmap = 1) -- a training step does not pay for the re-derivation,
      if ((db.kind == 5 ? m->conv_wino4 : m->conv_smallmap) < 2 && !m->workspace_reuse) {  // the next forward that wants them refreshes them (ph_model_forward)
        m->wino4_stale = true;
        continue;
      }
      const int rc5 = db.kind == 5 ? launch_wino4_pack(db.src, db.dst, db.panels, db.bn, static_cast<hipStream_t>(stream)) : launch_sm_pack(db.src, db.dst, db.panels, db.bn, db.n_tiles, static_cast<hipStream_t>(stream));
      if (rc5 != PH_OK) return rc5;
      continue;
    }
    int rc = db.kind == 1 ? launch_f16_weight_pack(db.src, db.dst, db.n_tiles, db.chunks0, db.chunks1, db.bn, db.plain, static_cast<hipStream_t>(stream))
             : db.kind == 3 ? launch_w16_pack(db.src, db.dst, db.panels, db.bn, static_cast<hipStream_t>(stream))
             : db.kind == 4 ? launch_stem_wino2d_pack(db.src, db.dst, static_cast<hipStream_t>(stream))
                            : launch_stem_wino_pack(db.src, db.dst, static_cast<hipStream_t>(stream));
    if (rc != PH_OK) return rc;
  }
  return PH_OK;
}

// Smallest / largest arena offset an op's parameters start at (-1: the op has none).
static int64_t op_param_offset(const ph_model* m, const ph_op_desc& d, bool highest = false) {
  int64_t v = -1;
  for (int idx : {d.weight, d.bias, d.weight2, d.bias2})
    if (idx >= 0 && idx < (int)m->weight_offset.size()) v = v < 0 ? m->weight_offset[idx] : (highest ? std::max(v, m->weight_offset[idx]) : std::min(v, m->weight_offset[idx]));
  return v;
}

// The op (forward order) whose completion in the reverse sweep makes the arena tail [offset, n_params) final: the program's
// parameters lie in op order, so the candidates are the ops at which the arena splits cleanly; the one closest to the middle wins.
static int bucket_split_op(const ph_model* m, int64_t* offset_out) {
  const int n = (int)m->ops.size();
  std::vector<int64_t> off(n), hi(n);
  for (int i = 0; i < n; ++i) {
    off[i] = op_param_offset(m, m->ops[i].d);
    hi[i] = op_param_offset(m, m->ops[i].d, true);
  }
  int best = -1;
  int64_t best_off = m->n_params;
  for (int k = 1; k < n; ++k) {
    if (off[k] <= 0) continue;
    bool clean = true;
    for (int i = 0; i < n && clean; ++i)
      if (off[i] >= 0) clean = i < k ? hi[i] < off[k] : off[i] >= off[k];  // EVERY parameter of an earlier op (weight2 / bias2 too) lies below the split
    if (!clean) continue;
    if (best < 0 || std::llabs(2 * off[k] - m->n_params) < std::llabs(2 * best_off - m->n_params)) {
      best = k;
      best_off = off[k];
    }
  }
  if (offset_out) *offset_out = best_off;
  return best;
}

int64_t ph_model_grad_bucket_split(const ph_model* m) {
  if (!m) {
    set_error("ph_model_grad_bucket_split: null model");
    return PH_E_INVALID;
  }
  int64_t off = m->n_params;
  bucket_split_op(m, &off);
  return off;
}

int ph_model_set_bucket_event(ph_model* m, void* hip_event) {
  PH_REQUIRE(m, "ph_model_set_bucket_event: null model");
  m->bucket_event = static_cast<hipEvent_t>(hip_event);
  return PH_OK;
}

// Data-parallel training behind the C ABI: with a communicator attached, ph_model_backward exchanges the gradient arena itself -- the tail bucket [split, n_params) on
// `comm_stream` as soon as the sweep has finished it (the encoder's gradients are still being computed on the caller's stream), the head bucket behind the sweep -- and the
// caller's stream waits for both before ph_model_backward's work on it counts as done: the next thing enqueued there (ph_adam_step with grad_scale = 1 / world) sees the
// SUM over the ranks.  What DDP's bucketed all-reduce does in the reference (training/model_trainer.py:1751-1813 hands the model to Lightning's DDP strategy).
int ph_model_set_comm(ph_model* m, ph_comm* comm, void* comm_stream) {
  PH_REQUIRE(m, "ph_model_set_comm: null model");
  PH_REQUIRE(!comm || comm_stream, "ph_model_set_comm: a communicator needs a side stream of its own (not the stream ph_model_backward runs on)");
  m->comm = comm;
  m->comm_stream = static_cast<hipStream_t>(comm_stream);
  if (comm && !m->comm_ev[0]) {
    for (auto& e : m->comm_ev) PH_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : m->comm_ev) m->comm_events_owned.push_back(e);
  }
  return PH_OK;
}

int64_t ph_model_backward_workspace_bytes(const ph_model* m, int32_t batch, int32_t height, int32_t width) {
  if (!m || batch <= 0 || height <= 0 || width <= 0) {
    set_error("ph_model_backward_workspace_bytes: bad arguments");
    return PH_E_INVALID;
  }
  BwdPlan bp;
  int rc = build_bwd_plan(m, batch, height, width, bp);
  if (rc != PH_OK) return rc;
  return bp.total;
}

int ph_model_backward(ph_model* m, const void* input_dev, int32_t in_dtype, int32_t batch, int32_t in_channels, int32_t height, int32_t width,
                      const void* act_workspace_dev, void* grad_workspace_dev, int64_t grad_workspace_bytes, const float* const* head_out_dev,
                      const float* const* target_dev, const float* loss_weights_host, const float* sample_weights_dev, int32_t ohkm_enabled, float hard_to_easy_ratio,
                      int32_t min_hard_keypoints, int32_t max_hard_keypoints, float ohkm_loss_scale, float* loss_dev, float* grads_flat_dev, void* stream) {
  PH_REQUIRE(m && input_dev && act_workspace_dev && grad_workspace_dev && head_out_dev && target_dev && loss_weights_host && loss_dev && grads_flat_dev,
             "ph_model_backward: null argument");
  PH_REQUIRE(((uintptr_t)grad_workspace_dev & 255) == 0, "gradient workspace must be 256-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  PH_REQUIRE(m->last_plan.fmt == FMT_F32, "backward needs the activations of an exact-fp32 forward (handle option conv_precision = 0)");
  PH_REQUIRE(!m->last_plan.reuse, "backward needs every activation of the forward (handle option workspace_reuse = 0)");
  BwdPlan bp;
  int rc = build_bwd_plan(m, batch, height, width, bp);
  if (rc != PH_OK) return rc;
  if (bp.total > grad_workspace_bytes) {
    set_error("gradient workspace too small: need %lld bytes, got %lld", (long long)bp.total, (long long)grad_workspace_bytes);
    return PH_E_WORKSPACE;
  }
  const char* aws = static_cast<const char*>(act_workspace_dev);
  char* gws = static_cast<char*>(grad_workspace_dev);
  auto A = [&](int slot) { return reinterpret_cast<const float*>(aws + bp.act.slots[slot].offset); };
  auto G = [&](int slot) { return reinterpret_cast<float*>(gws + bp.act.slots[slot].offset); };
  float* scratch = reinterpret_cast<float*>(gws + bp.scratch_off);
  std::vector<char> init(m->n_slots, 0);
  std::vector<char> fused_away(m->ops.size(), 0);
  // ReLU-mask folding: the gradient of a conv + ReLU output is complete when its FIRST consumer (program order = last in this sweep)
  // has added its share; when that consumer is a max pool, pool_bwd applies the mask itself (it reads the activation anyway) and the
  // conv's own step only reduces the bias gradient (handle option "mask_fold")
  std::vector<int> first_consumer(m->n_slots, -1), producer(m->n_slots, -1);
  std::vector<char> masked(m->n_slots, 0), bias_done(m->n_slots, 0);
  for (int oi = (int)m->ops.size() - 1; oi >= 0; --oi) {
    const ph_op_desc& d = m->ops[oi].d;
    if (d.src0 >= 0) first_consumer[d.src0] = oi;
    if (d.src1 >= 0) first_consumer[d.src1] = oi;
    if (d.dst >= 0) producer[d.dst] = oi;
  }
  OhkmParams ok;
  ok.enabled = ohkm_enabled;
  ok.hard_to_easy_ratio = hard_to_easy_ratio;
  ok.min_hard = min_hard_keypoints;
  ok.max_hard = max_hard_keypoints;
  ok.loss_scale = ohkm_loss_scale;

  // ---- losses and head-output gradients (loss_dev: [0] = total, [1 + i] = head i)
  std::vector<float> lw(loss_weights_host, loss_weights_host + m->n_outputs);
  for (const PackedOp& op : m->ops) {
    const ph_op_desc& d = op.d;
    if (d.kind != PH_OP_HEAD) continue;
    const SlotShape& s0 = bp.act.slots[d.src0];
    PH_REQUIRE(head_out_dev[d.out_index] && target_dev[d.out_index], "head %d: null output/target", d.out_index);
    float* dy = reinterpret_cast<float*>(gws + bp.head_dy_off[d.out_index]);
    if (d.flags & PH_FLAG_SOFTMAX)  // class-vector head: cross entropy on the softmax output, gradient wrt the logits
      rc = launch_class_ce(head_out_dev[d.out_index], target_dev[d.out_index], batch, d.cout, lw[d.out_index], dy, loss_dev + 1 + d.out_index, s);
    else
      rc = launch_loss(head_out_dev[d.out_index], target_dev[d.out_index], sample_weights_dev, batch, d.cout, s0.h, s0.w, lw[d.out_index], ok, scratch, dy,
                       loss_dev + 1 + d.out_index, s);
    if (rc != PH_OK) return rc;
  }
  // total = sum_h w_h * loss_h (the weights are a by-value kernel argument)
  rc = launch_total_loss(loss_dev + 1, lw.data(), m->n_outputs, loss_dev, s);
  if (rc != PH_OK) return rc;

  // ---- reverse sweep
  int64_t split_off = m->n_params;
  const bool exchange = m->comm != nullptr;  // (a communicator of ONE rank runs the same schedule: two in-place sums that change nothing -- what the single-GPU test exercises)
  const int split_op = (m->bucket_event || exchange) ? bucket_split_op(m, &split_off) : -1;
  if (split_op < 0) split_off = m->n_params;
  for (int oi = (int)m->ops.size() - 1; oi >= 0; --oi) {
    const PackedOp& op = m->ops[oi];
    const ph_op_desc& d = op.d;
    switch (d.kind) {
      case PH_OP_HEAD: {
        const SlotShape& s0 = bp.act.slots[d.src0];
        const float* dy = reinterpret_cast<const float*>(gws + bp.head_dy_off[d.out_index]);
        // this head completes the gradient of a conv + ReLU output (its first consumer): that ReLU's mask, and the conv's bias gradient, ride in the stores
        const int pr = producer[d.src0];
        const bool fold = m->mask_fold && first_consumer[d.src0] == oi && pr >= 0 && m->ops[pr].d.kind == PH_OP_CONV && (m->ops[pr].d.flags & PH_FLAG_RELU) &&
                          head_bwd_can_fold(s0.cp, s0.h * s0.w);
        const bool sum_bias = fold && m->ops[pr].d.bias >= 0;
        rc = launch_head_bwd(dy, head_out_dev[d.out_index], (d.flags & PH_FLAG_SIGMOID) ? 1 : 0, A(d.src0), op.w_dev, batch, s0.h * s0.w, d.cin0, s0.cp, d.cout,
                             init[d.src0], G(d.src0), grads_flat_dev + m->weight_offset[d.weight], d.bias >= 0 ? grads_flat_dev + m->weight_offset[d.bias] : nullptr,
                             scratch, s, fold ? A(d.src0) : nullptr, sum_bias ? grads_flat_dev + m->weight_offset[m->ops[pr].d.bias] : nullptr,
                             sum_bias ? m->ops[pr].d.cout : 0);
        init[d.src0] = 1;
        if (fold) masked[d.src0] = 1;
        bias_done[d.src0] = sum_bias ? 1 : 0;
        break;
      }
      case PH_OP_CONV: {
        const SlotShape& so = bp.act.slots[d.dst];
        PH_REQUIRE(init[d.dst], "conv output slot %d received no gradient", d.dst);
        const size_t npix = (size_t)batch * so.h * so.w;
        if (masked[d.dst]) {  // the last contributor applied the ReLU mask already (and, if it was a max pool's backward, summed the bias gradient)
          if (d.bias >= 0 && !bias_done[d.dst]) {
            rc = launch_bias_grad(G(d.dst), npix, so.cp, d.cout, grads_flat_dev + m->weight_offset[d.bias], scratch, s);
            if (rc != PH_OK) return rc;
          }
        } else if ((d.flags & PH_FLAG_RELU) && d.bias >= 0) {  // mask + bias gradient in one pass
          rc = launch_relu_mask_bias_grad(G(d.dst), A(d.dst), npix, so.cp, d.cout, grads_flat_dev + m->weight_offset[d.bias], scratch, s);
          if (rc != PH_OK) return rc;
        } else {
          if (d.flags & PH_FLAG_RELU) {
            rc = launch_relu_mask(G(d.dst), A(d.dst), npix * so.cp, s);
            if (rc != PH_OK) return rc;
          }
          if (d.bias >= 0) {
            rc = launch_bias_grad(G(d.dst), npix, so.cp, d.cout, grads_flat_dev + m->weight_offset[d.bias], scratch, s);
            if (rc != PH_OK) return rc;
          }
        }
        const int srcs[2] = {d.src0, d.src1}, parts[2] = {d.cin0, d.cin1}, offs[2] = {0, d.cin0};
        if (d.ksize != 3) {
          // k x k "same" conv (kernel_size 5 / 7 / 9, the 7 x 7 convs of a stem block; encoder_decoder.py:38-141,144-225): the weight gradient is
          // k^2 row-wgrad GEMMs (one per tap: rows = pixels, the x operand gathered at the tap's offset), the data gradient the same k^2-tap row
          // GEMM as the forward (mode 5) on the flipped, in/out-swapped weights, accumulating where the source already has a gradient
          for (int part = 0; part < 2; ++part) {
            if (parts[part] <= 0 || srcs[part] < 0) continue;
            const SlotShape& si = bp.act.slots[srcs[part]];
            for (int tap = 0; tap < d.ksize * d.ksize; ++tap) {
              RowWgradArgs w{};
              w.dy = G(d.dst);
              w.x = A(srcs[part]);
              w.slab = scratch;
              w.np = so.cp;
              w.kp = si.cp;
              w.M = (int)npix;
              w.patch = 2;
              w.ksize = d.ksize;
              w.tap = tap;
              w.H = so.h;
              w.W = so.w;
              rc = launch_row_wgrad_part(w, d.cout, parts[part], d.cin0 + d.cin1, offs[part], d.ksize * d.ksize, grads_flat_dev + m->weight_offset[d.weight], s);
              if (rc != PH_OK) return rc;
            }
            GemmArgs g{};
            g.src0 = G(d.dst);
            g.c0p = so.cp;
            g.wpack = op.wdk_gemm_dev[part];
            g.bias = op.zero_bias_dev;
            g.dst = G(srcs[part]);
            g.residual = init[srcs[part]] ? G(srcs[part]) : nullptr;
            g.zeros = m->zeros_dev;
            g.coutp = si.cp;
            g.bn = op.bn_dk[part];
            g.M = (int)npix;
            g.mode = 5;
            g.ksize = d.ksize;
            g.H = so.h;
            g.W = so.w;
            g.late_split = m->gemm_late_split;
            rc = launch_gemm(g, s);
            if (rc != PH_OK) return rc;
            init[srcs[part]] = 1;
          }
          break;
        }
        for (int part = 0; part < 2; ++part) {
          if (parts[part] <= 0 || srcs[part] < 0) continue;
          const SlotShape& si = bp.act.slots[srcs[part]];
          const double tile_fill = ((double)si.cp / ((si.cp + 127) / 128 * 128)) * ((double)so.cp / ((so.cp + 127) / 128 * 128));
          if (m->wgrad_rows == 2 || (m->wgrad_rows == 1 && si.cp >= 128 && so.cp >= 128 && tile_fill >= 0.8)) {
            // nine row-wgrad GEMMs (one per tap): 128x128 tiles on the MFMA instead of 32x32, for wide layers
            for (int tap = 0; tap < 9; ++tap) {
              RowWgradArgs w{};
              w.dy = G(d.dst);
              w.x = A(srcs[part]);
              w.slab = scratch;
              w.np = so.cp;
              w.kp = si.cp;
              w.M = (int)npix;
              w.patch = 2;
              w.tap = tap;
              w.H = so.h;
              w.W = so.w;
              rc = launch_row_wgrad_part(w, d.cout, parts[part], d.cin0 + d.cin1, offs[part], 9, grads_flat_dev + m->weight_offset[d.weight], s);
              if (rc != PH_OK) return rc;
            }
          } else {
            WgradArgs w{};
            w.x = A(srcs[part]);
            w.dy = G(d.dst);
            w.slab = scratch;
            w.cxp = si.cp;
            w.coutp = so.cp;
            w.B = batch;
            w.H = so.h;
            w.W = so.w;
            if (m->wgrad_wino && w.cxp == 16 && w.coutp == 16)
              rc = launch_wgrad16_wino(w, parts[part], d.cout, d.cin0 + d.cin1, offs[part], grads_flat_dev + m->weight_offset[d.weight], s);
            else if (m->wgrad_wino && w.coutp >= 32)
              rc = launch_wgrad_wino(w, parts[part], d.cout, d.cin0 + d.cin1, offs[part], grads_flat_dev + m->weight_offset[d.weight], s);
            else
              rc = launch_wgrad(w, parts[part], d.cout, d.cin0 + d.cin1, offs[part], grads_flat_dev + m->weight_offset[d.weight], s);
            if (rc != PH_OK) return rc;
          }
          // data gradient: conv3x3 of the masked output gradient with the flipped, swapped weights
          ConvArgs a{};
          a.src0 = G(d.dst);
          a.c0p = so.cp;
          a.src1 = nullptr;
          a.c1p = 0;
          a.wpack = op.wd_dev[part];
          a.wpack_dma = op.wd_dma_dev[part];
          a.wpack_wino = op.wd_wino_dev[part];
          a.wpack_wino2 = op.wd_wino2_dev[part];
          a.wpack_w16 = op.wd_w16_dev[part];
          a.wpack_wino4 = op.wd_wino4_dev[part];
          a.use_wino4 = m->conv_wino4 == 3 ? 2 : (m->conv_wino4 == 2 ? 1 : 0);  // (training plans keep every slot: the F(4x4,3x3) kernel runs in them only on request)
          a.w16 = op.wd16_dev;
          a.bias = op.zero_bias_dev;
          a.dst = G(srcs[part]);
          a.coutp = si.cp;
          a.B = batch;
          a.H = so.h;
          a.W = so.w;
          a.relu = 0;
          a.bn = op.bn_d[part];
          a.clock_probe = nullptr;
          a.zeros = m->zeros_dev;
          a.dst_pool = nullptr;
          apply_conv_options(m, a);
          if (!m->dgrad_wino) a.use_wino = 0;
          a.accumulate = init[srcs[part]];
          const bool dma = m->use_dma && (a.bn == 64 || m->dma32);
          {  // this launch completes the gradient of a conv + ReLU output and its kernel has a lane-local epilogue: that ReLU's mask rides there
            const int pr = producer[srcs[part]];
            const int pk = pr >= 0 ? m->ops[pr].d.kind : -1;
            if (m->mask_fold && dma && first_consumer[srcs[part]] == oi && (pk == PH_OP_CONV || pk == PH_OP_INPUT_CONV) && (m->ops[pr].d.flags & PH_FLAG_RELU) &&
                d.src0 != d.src1 && conv3x3_dma_honours_mask(a)) {
              a.relu_mask_src = A(srcs[part]);
              masked[srcs[part]] = 1;
            }
          }
          rc = dma ? launch_conv3x3_dma(a, s) : launch_conv3x3(a, s);
          if (rc != PH_OK) return rc;
          init[srcs[part]] = 1;
        }
        break;
      }
      case PH_OP_CONVT: {  // y = ReLU(ConvTranspose2d(x; Wt) + b), encoder_decoder.py:439-461
        const SlotShape& so = bp.act.slots[d.dst];
        const SlotShape& si = bp.act.slots[d.src0];
        PH_REQUIRE(init[d.dst], "transposed conv output slot %d received no gradient", d.dst);
        const size_t npix = (size_t)batch * so.h * so.w;
        if (d.flags & PH_FLAG_RELU)
          rc = launch_relu_mask_bias_grad(G(d.dst), A(d.dst), npix, so.cp, d.cout, grads_flat_dev + m->weight_offset[d.bias], scratch, s);
        else
          rc = launch_bias_grad(G(d.dst), npix, so.cp, d.cout, grads_flat_dev + m->weight_offset[d.bias], scratch, s);
        // dWt[ci][co][ky][kx] = sum over input pixels (y, x) of x[y, x, ci] * dY[2y - 1 + ky, 2x - 1 + kx, co]: one row-wgrad GEMM per tap,
        // rows = input pixels, the dY operand gathered with stride 2 (patch 3); the result lands in the IOHW layout directly
        for (int tap = 0; tap < 9 && rc == PH_OK; ++tap) {
          RowWgradArgs w{};
          w.dy = A(d.src0);  // the "n" operand: input channels
          w.np = si.cp;
          w.x = G(d.dst);    // the gathered "k" operand: output channels
          w.kp = so.cp;
          w.slab = scratch;
          w.M = batch * si.h * si.w;
          w.patch = 3;
          w.tap = tap;
          w.H = so.h;
          w.W = so.w;
          rc = launch_row_wgrad(w, d.cin0, d.cout, 9, grads_flat_dev + m->weight_offset[d.weight], s);
        }
        if (rc != PH_OK) return rc;
        // dX = Conv2d(dY, Wt viewed as (out = cin0, in = cout, 3, 3), stride 2, pad 1): row GEMM mode 4
        GemmArgs g{};
        g.src0 = G(d.dst);
        g.c0p = so.cp;
        g.wpack = op.wt_dgrad_dev;
        g.bias = op.zero_bias_dev;
        g.dst = G(d.src0);
        g.residual = init[d.src0] ? G(d.src0) : nullptr;  // accumulate into a gradient that is already there
        g.zeros = m->zeros_dev;
        g.coutp = si.cp;
        g.bn = op.bn_td;
        g.M = batch * si.h * si.w;
        g.mode = 4;
        g.H = so.h;
        g.W = so.w;
        g.late_split = m->gemm_late_split;
        rc = launch_gemm(g, s);
        init[d.src0] = 1;
        break;
      }
      case PH_OP_INPUT_CONV: {
        const SlotShape& so = bp.act.slots[d.dst];
        PH_REQUIRE(init[d.dst], "first conv output received no gradient");
        PH_REQUIRE(d.cin0 == in_channels, "input has %d channels, network expects %d", in_channels, d.cin0);
        const size_t npix = (size_t)batch * so.h * so.w;
        // (a masked gradient and a 3 x 3 kernel: the weight-gradient launch below reads every element anyway and sums the bias gradient in a spare column)
        const bool bias_in_wgrad = masked[d.dst] && d.bias >= 0 && d.ksize == 3 && !bias_done[d.dst];
        if (masked[d.dst]) {  // the last contributor applied the ReLU mask already
          if (d.bias >= 0 && !bias_done[d.dst] && !bias_in_wgrad) {
            rc = launch_bias_grad(G(d.dst), npix, so.cp, d.cout, grads_flat_dev + m->weight_offset[d.bias], scratch, s);
            if (rc != PH_OK) return rc;
          }
        } else if ((d.flags & PH_FLAG_RELU) && d.bias >= 0) {  // mask + bias gradient in one pass
          rc = launch_relu_mask_bias_grad(G(d.dst), A(d.dst), npix, so.cp, d.cout, grads_flat_dev + m->weight_offset[d.bias], scratch, s);
          if (rc != PH_OK) return rc;
        } else {
          if (d.flags & PH_FLAG_RELU) {
            rc = launch_relu_mask(G(d.dst), A(d.dst), npix * so.cp, s);
            if (rc != PH_OK) return rc;
          }
          if (d.bias >= 0) {
            rc = launch_bias_grad(G(d.dst), npix, so.cp, d.cout, grads_flat_dev + m->weight_offset[d.bias], scratch, s);
            if (rc != PH_OK) return rc;
          }
        }
        if (d.ksize != 3)  // first k x k "same" conv: im2col of the image + one row-wgrad GEMM (as the ConvNeXt patch stem, padding k / 2, stride 1)
          rc = launch_patch_stem_wgrad(input_dev, in_dtype, G(d.dst), batch, d.cin0, height, width, so.h, so.w, d.ksize, 1, d.ksize / 2, so.cp, d.cout,
                                       grads_flat_dev + m->weight_offset[d.weight], scratch, s);
        else
          rc = launch_input_wgrad(input_dev, in_dtype, G(d.dst), batch, d.cin0, height, width, so.cp, d.cout, grads_flat_dev + m->weight_offset[d.weight], scratch, s,
                                  bias_in_wgrad ? grads_flat_dev + m->weight_offset[d.bias] : nullptr);
        break;
      }
      case PH_OP_POOL: {
        const SlotShape& si = bp.act.slots[d.src0];
        PH_REQUIRE(init[d.dst], "pool output slot %d received no gradient", d.dst);
        const int pr = producer[d.src0];
        const bool fold = m->mask_fold && first_consumer[d.src0] == oi && pr >= 0 && m->ops[pr].d.kind == PH_OP_CONV && (m->ops[pr].d.flags & PH_FLAG_RELU);
        // ... and, the gradient being complete and masked as it is stored, it is summed per channel on the way: the producer conv's bias gradient,
        // which its own step would otherwise read the whole tensor again for
        const bool sum_bias = fold && m->ops[pr].d.bias >= 0 && pool_bwd_can_sum_bias(si.cp);
        rc = launch_pool_bwd(G(d.dst), A(d.src0), batch, si.h, si.w, si.cp, init[d.src0], fold ? 1 : 0, G(d.src0),
                             sum_bias ? grads_flat_dev + m->weight_offset[m->ops[pr].d.bias] : nullptr, sum_bias ? m->ops[pr].d.cout : 0, scratch, s);
        init[d.src0] = 1;
        masked[d.src0] = fold ? 1 : 0;
        bias_done[d.src0] = sum_bias ? 1 : 0;
        break;
      }
      case PH_OP_UPSAMPLE: {
        const SlotShape& si = bp.act.slots[d.src0];
        PH_REQUIRE(init[d.dst], "upsample output slot %d received no gradient", d.dst);
        rc = launch_upsample_bwd(G(d.dst), batch, si.h, si.w, si.cp, init[d.src0], G(d.src0), s);
        init[d.src0] = 1;
        break;
      }
      case PH_OP_SCALE_ADD: {  // y = s * u + x
        const SlotShape& so = bp.act.slots[d.dst];
        PH_REQUIRE(init[d.dst], "scale-add output slot %d received no gradient", d.dst);
        const size_t npix = (size_t)batch * so.h * so.w;
        rc = launch_chan_reduce(G(d.dst), A(d.src0), nullptr, npix, so.cp, d.cout, grads_flat_dev + m->weight_offset[d.weight], scratch, s);
        if (rc != PH_OK) return rc;
        PH_REQUIRE(!init[d.src0], "scale-add input slot %d already has a gradient", d.src0);
        rc = launch_scale_add_bwd(G(d.dst), op.w_dev, G(d.src0), G(d.src1), init[d.src1], so.cp, npix * so.cp, s);
        init[d.src0] = 1;
        init[d.src1] = 1;
        break;
      }
      case PH_OP_GELU: {
        if (fused_away[oi]) break;  // its consumer's data-gradient GEMM already wrote G(src0) through the GELU derivative
        const SlotShape& so = bp.act.slots[d.dst];
        PH_REQUIRE(init[d.dst], "GELU output slot %d received no gradient", d.dst);
        rc = launch_gelu_bwd(G(d.dst), A(d.src0), G(d.src0), init[d.src0], (size_t)batch * so.h * so.w * so.cp, s);
        init[d.src0] = 1;
        break;
      }
      case PH_OP_LINEAR:
      case PH_OP_PATCH_CONV: {
        const SlotShape& so = bp.act.slots[d.dst];
        const SlotShape& si = bp.act.slots[d.src0];
        PH_REQUIRE(init[d.dst], "GEMM output slot %d received no gradient", d.dst);
        PH_REQUIRE(!init[d.src0], "GEMM input slot %d already has a gradient (accumulation is not supported here)", d.src0);
        const bool patch = d.kind == PH_OP_PATCH_CONV;
        PH_REQUIRE(!patch || (si.h % 2 == 0 && si.w % 2 == 0), "2x2/stride-2 conv backward needs even input sizes");
        const int M = batch * so.h * so.w;
        const int taps = patch ? 4 : 1;
        // Linear fed by a GELU whose output nobody else reads (CNBlock: Linear -> GELU -> Linear): the data-gradient
        // GEMM multiplies by GELU'(pre-activation) in its epilogue and writes the GELU input's gradient directly,
        // so the GELU output's gradient never exists in HBM
        int gelu_op = -1;
        if (!patch && m->fuse_gelu_bwd) {
          int readers = 0;
          for (size_t j = 0; j < m->ops.size(); ++j) {
            const ph_op_desc& e = m->ops[j].d;
            if (e.src0 == d.src0 || e.src1 == d.src0) readers += 1;
            if ((int)j < oi && e.dst == d.src0 && e.kind == PH_OP_GELU) gelu_op = (int)j;
          }
          if (gelu_op >= 0 && (readers != 1 || init[m->ops[gelu_op].d.src0])) gelu_op = -1;
        }
        // a Linear layer's bias gradient (column sums of dY) comes out of its weight-gradient GEMM, which stages every dY row anyway
        const bool bias_in_wgrad = !patch && !(d.flags & PH_FLAG_RELU) && d.bias >= 0;
        if (d.flags & PH_FLAG_RELU)
          rc = launch_relu_mask_bias_grad(G(d.dst), A(d.dst), (size_t)M, so.cp, d.cout, grads_flat_dev + m->weight_offset[d.bias], scratch, s);
        else if (!bias_in_wgrad)
          rc = launch_bias_grad(G(d.dst), (size_t)M, so.cp, d.cout, grads_flat_dev + m->weight_offset[d.bias], scratch, s);
        for (int tap = 0; tap < taps && rc == PH_OK; ++tap) {
          RowWgradArgs w{};
          w.dy = G(d.dst);
          w.x = A(d.src0);
          w.slab = scratch;
          w.np = so.cp;
          w.kp = si.cp;
          w.M = M;
          if (bias_in_wgrad) {
            w.gb = grads_flat_dev + m->weight_offset[d.bias];
            w.gb_n = d.cout;
          }
          w.patch = patch ? 1 : 0;
          w.tap = tap;
          w.H = si.h;
          w.W = si.w;
          rc = launch_row_wgrad(w, d.cout, d.cin0, taps, grads_flat_dev + m->weight_offset[d.weight], s);
          if (rc != PH_OK) break;
          GemmArgs g{};
          g.src0 = G(d.dst);
          g.c0p = so.cp;
          g.wpack = op.wd_gemm_dev[tap];
          g.bias = op.zero_bias_dev;
          g.dst = G(d.src0);
          g.zeros = m->zeros_dev;
          g.coutp = si.cp;
          g.bn = op.bn_dg;
          g.M = M;
          g.mode = 0;
          g.out_patch = patch ? 1 : 0;
          g.out_tap = tap;
          g.out_H = si.h;
          g.out_W = si.w;
          if (gelu_op >= 0) {
            g.dst = G(m->ops[gelu_op].d.src0);
            g.act = 3;
            g.aux = A(m->ops[gelu_op].d.src0);
          }
          rc = launch_gemm(g, s);
        }
        init[d.src0] = 1;
        if (gelu_op >= 0) {
          init[m->ops[gelu_op].d.src0] = 1;
          fused_away[gelu_op] = 1;
        }
        break;
      }
      case PH_OP_LAYERNORM: {
        const SlotShape& so = bp.act.slots[d.dst];
        PH_REQUIRE(init[d.dst], "LayerNorm output slot %d received no gradient", d.dst);
        const size_t npix = (size_t)batch * so.h * so.w;
        float* stats = scratch;
        float* red = scratch + 2 * npix;
        rc = launch_layernorm_bwd(A(d.src0), G(d.dst), op.w_dev, G(d.src0), stats, init[d.src0], so.c, so.cp, npix, s);
        if (rc != PH_OK) return rc;
        rc = launch_chan_reduce(G(d.dst), A(d.src0), stats, npix, so.cp, so.c, grads_flat_dev + m->weight_offset[d.weight], red, s);
        if (rc != PH_OK) return rc;
        rc = launch_chan_reduce(G(d.dst), nullptr, nullptr, npix, so.cp, so.c, grads_flat_dev + m->weight_offset[d.bias], red, s);
        init[d.src0] = 1;
        break;
      }
      case PH_OP_DWCONV: {
        const SlotShape& so = bp.act.slots[d.dst];
        PH_REQUIRE(init[d.dst], "depthwise conv output slot %d received no gradient", d.dst);
        rc = launch_bias_grad(G(d.dst), (size_t)batch * so.h * so.w, so.cp, d.cout, grads_flat_dev + m->weight_offset[d.bias], scratch, s);
        if (rc != PH_OK) return rc;
        rc = launch_dwconv7_wgrad(A(d.src0), G(d.dst), batch, so.h, so.w, so.cp, d.cout, grads_flat_dev + m->weight_offset[d.weight], scratch, s);
        if (rc != PH_OK) return rc;
        DwConvArgs a{};
        a.src = G(d.dst);
        a.w = op.dw_flip_dev;
        a.bias = nullptr;
        a.dst = G(d.src0);
        a.cp = so.cp;
        a.B = batch;
        a.H = so.h;
        a.W = so.w;
        a.accumulate = init[d.src0];
        rc = launch_dwconv7(a, s);
        init[d.src0] = 1;
        break;
      }
      case PH_OP_GLOBAL_MAXPOOL: {
        const SlotShape& si = bp.act.slots[d.src0];
        PH_REQUIRE(init[d.dst], "global pool output slot %d received no gradient", d.dst);
        rc = launch_global_maxpool_bwd(A(d.src0), G(d.dst), batch, si.h * si.w, si.cp, init[d.src0], G(d.src0), s);
        init[d.src0] = 1;
        break;
      }
      case PH_OP_PATCH_STEM: {
        const SlotShape& so = bp.act.slots[d.dst];
        PH_REQUIRE(init[d.dst], "stem output received no gradient");
        PH_REQUIRE(d.cin0 == in_channels, "input has %d channels, network expects %d", in_channels, d.cin0);
        rc = launch_bias_grad(G(d.dst), (size_t)batch * so.h * so.w, so.cp, d.cout, grads_flat_dev + m->weight_offset[d.bias], scratch, s);
        if (rc != PH_OK) return rc;
        rc = launch_patch_stem_wgrad(input_dev, in_dtype, G(d.dst), batch, d.cin0, height, width, so.h, so.w, d.ksize, d.cmid, 1, so.cp, d.cout,
                                     grads_flat_dev + m->weight_offset[d.weight], scratch, s);
        break;
      }
      default:
        set_error("backward: unsupported op kind %d", d.kind);
        rc = PH_E_INVALID;
    }
    if (rc != PH_OK) return rc;
    if (oi == split_op) {  // gradients [bucket split, n_params) are final from here on
      if (m->bucket_event) PH_HIP_CHECK(hipEventRecord(m->bucket_event, s));
      if (exchange && split_off < m->n_params) {  // tail bucket: on the side stream, under the rest of the sweep
        PH_HIP_CHECK(hipEventRecord(m->comm_ev[0], s));
        PH_HIP_CHECK(hipStreamWaitEvent(m->comm_stream, m->comm_ev[0], 0));
        rc = ph_allreduce(m->comm, grads_flat_dev + split_off, m->n_params - split_off, m->comm_stream);
        if (rc != PH_OK) return rc;
      }
    }
  }
  if (m->bucket_event && split_op < 0) PH_HIP_CHECK(hipEventRecord(m->bucket_event, s));
  if (exchange) {  // head bucket (the whole arena when it does not split), then the caller's stream joins the side stream
    PH_HIP_CHECK(hipEventRecord(m->comm_ev[1], s));
    PH_HIP_CHECK(hipStreamWaitEvent(m->comm_stream, m->comm_ev[1], 0));
    rc = ph_allreduce(m->comm, grads_flat_dev, split_off, m->comm_stream);
    if (rc != PH_OK) return rc;
    PH_HIP_CHECK(hipEventRecord(m->comm_ev[2], m->comm_stream));
    PH_HIP_CHECK(hipStreamWaitEvent(s, m->comm_ev[2], 0));
  }
  return PH_OK;
}

}  // extern "C"
