// Host runtime of the network path: op program, weight re-packing, workspace planning and
// the forward launch sequence (C ABI: ph_model_*).  Replaces TorchBackend.__call__ ->
// LightningModule.forward -> Model.forward -> UNet.forward of the reference
// (sleap_nn/inference/layers/backends/torch_backend.py:113-153,
//  training/lightning_modules.py:1840-1848, architectures/model.py:237-261,
//  architectures/unet.py:260-299).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

#include "model_internal.h"
#include "train_kernels.h"

// diagnostic clock probe (ph_model_set_clock_probe): 64-bit words reserved per op of the program -- the largest writer (the F(2x2,3x3) kernel's stamps) takes 256 x 8 x 8
static constexpr size_t PH_PROBE_WORDS_PER_OP = 32768;

namespace ph {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

}  // namespace ph

namespace ph {

int device_cu_count(int* out) {
  static std::atomic<int> cache[64];  // zero-initialised; a count is a device property, racing writers store the same value
  int dev = 0;
  PH_HIP_CHECK(hipGetDevice(&dev));
  int n = (dev >= 0 && dev < 64) ? cache[dev].load(std::memory_order_relaxed) : 0;
  if (!n) {
    PH_HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    if (dev >= 0 && dev < 64) cache[dev].store(n, std::memory_order_relaxed);
  }
  *out = n;
  return PH_OK;
}

int upload(ph_model* m, const std::vector<float>& host, float** dev) {
  void* p = nullptr;
  PH_HIP_CHECK(hipMalloc(&p, std::max<size_t>(host.size(), 4) * sizeof(float)));
  m->allocs.push_back(p);
  PH_HIP_CHECK(hipMemcpy(p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  *dev = static_cast<float*>(p);
  return PH_OK;
}

int upload_ints(ph_model* m, const std::vector<int>& host, int** dev) {
  void* p = nullptr;
  PH_HIP_CHECK(hipMalloc(&p, std::max<size_t>(host.size(), 4) * sizeof(int)));
  m->allocs.push_back(p);
  PH_HIP_CHECK(hipMemcpy(p, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice));
  *dev = static_cast<int*>(p);
  return PH_OK;
}

// Upload a packed value array together with its gather map.  `packed_idx` is the SAME packing
// applied to an array holding (canonical index + 1) as doubles (exact for any arena size).
static int upload_packed(ph_model* m, const std::vector<float>& packed_val, const std::vector<double>& packed_idx, float** dev) {
  int rc = upload(m, packed_val, dev);
  if (rc != PH_OK) return rc;
  std::vector<int> map(packed_idx.size());
  for (size_t i = 0; i < packed_idx.size(); ++i) map[i] = (int)((int64_t)packed_idx[i] - 1);
  PackedBuffer pb;
  pb.dst = *dev;
  pb.n = packed_val.size();
  rc = upload_ints(m, map, &pb.map);
  if (rc != PH_OK) return rc;
  m->packed.push_back(pb);
  return PH_OK;
}

// Run one packing routine twice -- on the weight values and on their canonical indices -- and upload both.
template <typename PackFn>
static int pack_upload(ph_model* m, PackFn&& pack, const float* values, const std::vector<double>& indices, float** dev) {
  std::vector<float> v;
  std::vector<double> x;
  pack(values, v);
  pack(indices.data(), x);
  return upload_packed(m, v, x, dev);
}

// OIHW (cout, cin_total, 3, 3) -> the conv weight of the data gradient wrt channels
// [ci_off, ci_off + cin_part): Wd[o = ci][i = co][ky][kx] = W[co][ci_off + ci][2-ky][2-kx].
template <typename T>
static void dgrad_weight(const T* w, int cout, int cin_total, int ci_off, int cin_part, std::vector<T>& out, int k = 3) {
  out.assign((size_t)cin_part * cout * k * k, T(0));
  for (int ci = 0; ci < cin_part; ++ci)
    for (int co = 0; co < cout; ++co)
      for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx)
          out[(((size_t)ci * cout + co) * k + ky) * k + kx] = w[(((size_t)co * cin_total + ci_off + ci) * k + (k - 1 - ky)) * k + (k - 1 - kx)];
}

// Conv2d OIHW (cout, cin0+cin1, 3, 3) or ConvTranspose2d IOHW (cin0, cout, 3, 3) ->
// [n_tile][chunk][tap][bn][16] with channel/row padding zeros.
template <typename T>
static void pack_conv(const T* w, bool transposed, int cin0, int cin1, int cout, int bn, std::vector<T>& out) {
  const int c0p = pad16(cin0), c1p = cin1 > 0 ? pad16(cin1) : 0;
  const int coutp = pad16(cout);
  const int ntiles = (coutp + bn - 1) / bn;
  const int ch0 = c0p / 16, ch1 = c1p / 16, nch = ch0 + ch1;
  const int cin = cin0 + cin1;
  out.assign((size_t)ntiles * nch * 9 * bn * 16, T(0));
  for (int nt = 0; nt < ntiles; ++nt)
    for (int ch = 0; ch < nch; ++ch)
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap % 3;
        for (int n = 0; n < bn; ++n) {
          const int co = nt * bn + n;
          if (co >= cout) continue;
          for (int kc = 0; kc < 16; ++kc) {
            int ci;
            if (ch < ch0) {
              const int c = ch * 16 + kc;
              if (c >= cin0) continue;
              ci = c;
            } else {
              const int c = (ch - ch0) * 16 + kc;
              if (c >= cin1) continue;
              ci = cin0 + c;
            }
            T v;
            if (!transposed)
              v = w[(((size_t)co * cin + ci) * 3 + ky) * 3 + kx];
            else  // flipped, in/out swapped: Wc[co][ci][ky][kx] = Wt[ci][co][2-ky][2-kx]
              v = w[(((size_t)ci * cout + co) * 3 + (2 - ky)) * 3 + (2 - kx)];
            out[((((size_t)nt * nch + ch) * 9 + tap) * bn + n) * 16 + kc] = v;
          }
        }
      }
}

// [n_tile][chunk][tap][bn][16] -> [n_tile][chunk][piece = (tap*bn + n)/16][quad q][row r = (tap*bn+n)%16][4]
// (the order in which one LDS-DMA wave-instruction writes a 1-KiB piece, see conv3x3_mfma_dma_kernel)
// [tap][ci 16][co 16] (zero padded) from OIHW, for conv3x3_c16_kernel
template <typename T>
static void pack_c16(const T* w, int cout, int cin, std::vector<T>& out) {
  out.assign(9 * 256, T(0));
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int tap = 0; tap < 9; ++tap) out[(size_t)(tap * 16 + ci) * 16 + co] = w[((size_t)co * cin + ci) * 9 + tap];
}

template <typename T>
static void repack_dma(const std::vector<T>& in, int bn, std::vector<T>& out) {
  out.resize(in.size());
  const size_t panel = (size_t)9 * bn * 16;
  for (size_t base = 0; base < in.size(); base += panel)
    for (int row = 0; row < 9 * bn; ++row)
      for (int k = 0; k < 16; ++k) {
        const int piece = row >> 4, r = row & 15, q = k >> 2, e = k & 3;
        out[base + (size_t)piece * 256 + q * 64 + r * 4 + e] = in[base + (size_t)row * 16 + k];
      }
}

// Row-GEMM weights (gemm_mfma_dma_kernel): Linear (cout, cin) [taps = 1], Conv2d k2 s2 (cout, cin, 2, 2)
// [taps = 4] or Conv2d 3x3 over concat(cin0, cin1) (cout, cin0 + cin1, 3, 3) [taps = 9] ->
// [n_tile][stage][piece = n / 8][slot][row n % 8][4]; a stage is 32 channels of one source at one tap,
// stage order = source, 32-channel slice, tap (taps innermost); slot = channel quad ^ (piece & 1)
// (the LDS image the kernel's DMA writes); zero-padded channels / rows.
// CNBlock MLP, first Linear (4C, C) -> [hidden block hb][piece q: C / 8][lane (i = lane & 31, h = lane >> 5)][t]: W1[hb 32 + i][h C / 2 + 4 q + t]
// (the A operand of K step s = 4 q + t of cnblock_mlp_kernel's first product: channels (s, C / 2 + s) in the two lane halves)
template <typename T>
static void pack_mlp_w1(const T* w, int c, std::vector<T>& out) {
  const int nhb = 4 * c / 32, p1 = c / 8;
  out.assign((size_t)nhb * p1 * 256, T(0));
  for (int hb = 0; hb < nhb; ++hb)
    for (int q = 0; q < p1; ++q)
      for (int lane = 0; lane < 64; ++lane)
        for (int t = 0; t < 4; ++t) out[(((size_t)hb * p1 + q) * 64 + lane) * 4 + t] = w[(size_t)(hb * 32 + (lane & 31)) * c + (lane >> 5) * (c / 2) + 4 * q + t];
}
// ... second Linear (C, 4C) -> [hb][piece nb 4 + g][lane (n = lane & 31, h)][t]: W2[nb 32 + n][hb 32 + 8 g + 4 h + t]: K step r = 4 g + t of the chained product is
// the pair of hidden channels that accumulator register r of the first product holds in the two lane halves
template <typename T>
static void pack_mlp_w2(const T* w, int c, std::vector<T>& out) {
  const int nhb = 4 * c / 32, nb_ = c / 32;
  out.assign((size_t)nhb * nb_ * 4 * 256, T(0));
  for (int hb = 0; hb < nhb; ++hb)
    for (int nb = 0; nb < nb_; ++nb)
      for (int g = 0; g < 4; ++g)
        for (int lane = 0; lane < 64; ++lane)
          for (int t = 0; t < 4; ++t)
            out[((((size_t)hb * nb_ + nb) * 4 + g) * 64 + lane) * 4 + t] = w[(size_t)(nb * 32 + (lane & 31)) * (4 * c) + hb * 32 + 8 * g + 4 * (lane >> 5) + t];
}

template <typename T>
static void pack_gemm(const T* w, int cout, int cin0, int cin1, int taps, int bn, std::vector<T>& out) {
  const int c0p = pad16(cin0), c1p = cin1 > 0 ? pad16(cin1) : 0, coutp = pad16(cout);
  const int ntiles = (coutp + bn - 1) / bn;
  const int sl0 = (c0p + 31) / 32, sl1 = (c1p + 31) / 32;
  const int total = (sl0 + sl1) * taps, bp = bn / 8, cin = cin0 + cin1;
  out.assign((size_t)ntiles * total * bp * 256, T(0));
  for (int nt = 0; nt < ntiles; ++nt)
    for (int st = 0; st < total; ++st) {
      const int slice = st / taps, tap = st % taps;
      const bool second = slice >= sl0;
      const int c0 = (second ? slice - sl0 : slice) * 32;
      for (int pb = 0; pb < bp; ++pb)
        for (int slot = 0; slot < 8; ++slot)
          for (int r = 0; r < 8; ++r)
            for (int e = 0; e < 4; ++e) {
              const int co = nt * bn + pb * 8 + r, c = c0 + (slot ^ (pb & 1)) * 4 + e;
              if (co >= cout || c >= (second ? cin1 : cin0)) continue;
              const int ci = second ? cin0 + c : c;
              out[(((size_t)nt * total + st) * bp + pb) * 256 + slot * 32 + r * 4 + e] = w[((size_t)co * cin + ci) * taps + tap];
            }
    }
}

int choose_bn(int coutp) { return coutp >= 64 ? 64 : 32; }

void apply_conv_options(const ph_model* m, ConvArgs& a) {
  a.use_wino = m->conv_wino;
  a.use_wino2d = m->conv_wino2d;
  a.wino4_min_cin = m->conv_wino4_min_cin;
  a.use_w16 = m->conv_w16;
  a.persist = m->conv_persist;
  a.use_c16 = m->conv_c16;
  a.dma_stagger = m->dma_stagger;
  a.splitk = m->conv_splitk;  // (ph_model_forward clears it for a training plan unless a slice count is forced)
  a.split_counters = m->split_counters_dev;
  a.split_counters_n = m->split_counters_dev ? 4096 : 0;  // (units; the buffer holds three words per unit)
  a.splitk_finish = m->conv_splitk_finish;
  a.use_sm = 0;  // (ph_model_forward: the kind of plan decides)
}

// Programs made only of the UNet-style ops can run on the fp16 matrix pipe (handle option "conv_precision"); anything
// else (ConvNeXt blocks, class-vector heads, non-3x3 kernels) and every training program stays in exact fp32.
int forward_format(const ph_model* m) {
  if (m->conv_precision != 1 && m->conv_precision != 2) return FMT_F32;
  for (const PackedOp& op : m->ops) {
    const ph_op_desc& d = op.d;
    switch (d.kind) {
      case PH_OP_STEM:
      case PH_OP_INPUT_CONV:
      case PH_OP_POOL:
      case PH_OP_UPSAMPLE:
        break;
      case PH_OP_CONV:
      case PH_OP_CONVT:
        if (d.ksize != 3) return FMT_F32;
        break;
      case PH_OP_HEAD:
        if (d.flags & PH_FLAG_SOFTMAX) return FMT_F32;
        break;
      default:
        return FMT_F32;
    }
  }
  return m->conv_precision == 1 ? FMT_SPLIT : FMT_F16;
}

int build_plan(const ph_model* m, int B, int H, int W, Plan& plan, int fmt) {
  plan.slots.assign(m->n_slots, SlotShape());
  plan.fmt = fmt;
  plan.bpc = fmt_bytes_per_channel(fmt);
  const int bpc = plan.bpc;
  int64_t off = 0, tmp = 0;
  // Slot allocation.  Default: every slot gets its own range (training keeps every activation; ph_model_read_slot can read any of
  // them back).  With the handle option "workspace_reuse" (inference programs) a slot's range returns to a free list after its
  // last reader and later slots take the best-fitting free block: 9.0 -> 3.6 GB at cfg3 x 32 frames, and a layer's output is
  // written where an older tensor has just left the caches.
  const bool reuse = m->workspace_reuse != 0;
  plan.reuse = reuse;
  const int n_ops = (int)m->ops.size();
  std::vector<int> last_use(m->n_slots, -1);
  std::vector<int64_t> slot_bytes(m->n_slots, 0);
  std::vector<char> released(m->n_slots, 0);
  if (reuse)
    for (int j = 0; j < n_ops; ++j) {
      const ph_op_desc& e = m->ops[j].d;
      if (e.src0 >= 0 && e.src0 < m->n_slots) last_use[e.src0] = j;
      if (e.src1 >= 0 && e.src1 < m->n_slots) last_use[e.src1] = j;
    }
  if (reuse)  // a bilinear op the next conv may fold into its input transform (ph_model_forward, PH_OP_UPSAMPLE): that conv then reads the op's SOURCE
    for (int j = 0; j + 1 < n_ops; ++j) {
      const ph_op_desc& e = m->ops[j].d;
      const ph_op_desc& nx = m->ops[j + 1].d;
      if (e.kind == PH_OP_UPSAMPLE && nx.kind == PH_OP_CONV && nx.src1 == e.dst && e.src0 >= 0 && e.src0 < m->n_slots) last_use[e.src0] = std::max(last_use[e.src0], j + 1);
    }
  plan.unread.assign(m->n_slots, 0);
  if (reuse)
    for (int sl = 0; sl < m->n_slots; ++sl) plan.unread[sl] = last_use[sl] < 0;
  std::vector<std::pair<int64_t, int64_t>> free_list;  // (offset, bytes), kept sorted and coalesced
  auto release = [&](int slot) {
    free_list.emplace_back(plan.slots[slot].offset, slot_bytes[slot]);
    std::sort(free_list.begin(), free_list.end());
    for (size_t k = 0; k + 1 < free_list.size();)
      if (free_list[k].first + free_list[k].second == free_list[k + 1].first) {
        free_list[k].second += free_list[k + 1].second;
        free_list.erase(free_list.begin() + k + 1);
      } else
        ++k;
    released[slot] = 1;
  };
  auto alloc = [&](int slot, int64_t bytes) {
    bytes = align_up(bytes, 256);
    slot_bytes[slot] = bytes;
    int best = -1;
    for (size_t k = 0; k < free_list.size(); ++k)
      if (free_list[k].second >= bytes && (best < 0 || free_list[k].second < free_list[best].second)) best = (int)k;
    if (best >= 0) {
      const int64_t o = free_list[best].first;
      free_list[best].first += bytes;
      free_list[best].second -= bytes;
      if (free_list[best].second == 0) free_list.erase(free_list.begin() + best);
      return o;
    }
    if (!free_list.empty() && free_list.back().first + free_list.back().second == off) {  // grow the trailing free block
      const int64_t o = free_list.back().first;
      free_list.pop_back();
      off = o + bytes;
      return o;
    }
    const int64_t o = off;
    off += bytes;
    return o;
  };
  int op_i = -1;
  for (const PackedOp& op : m->ops) {
    const ph_op_desc& d = op.d;
    ++op_i;
    // Run-time fusions let op i write op i + 1's dst one op early (pool_peephole: a conv's epilogue writes the pool that follows it;
    // fuse_gelu_fwd: a Linear writes its GELU or its layer-scale + residual; dw_ln_fuse: a depthwise / stem conv writes its LayerNorm).  Such a dst must not land on
    // a range op i is still reading, so the releases due before a pool / GELU / LayerNorm of the previous op's output wait one op.
    const bool forwarded = op_i > 0 && (d.kind == PH_OP_POOL || d.kind == PH_OP_GELU || d.kind == PH_OP_LAYERNORM || d.kind == PH_OP_SCALE_ADD) && d.src0 >= 0 && d.src0 == m->ops[op_i - 1].d.dst;
    if (reuse && !forwarded)  // slots whose last reader ran before this op (a slot nobody reads is released right after the op that wrote it)
      for (int sl = 0; sl < m->n_slots; ++sl)
        if (!released[sl] && plan.slots[sl].offset >= 0 && slot_bytes[sl] > 0 && std::max(last_use[sl], plan.slots[sl].def_op) < op_i) release(sl);
    int h, w;
    if (d.src0 < 0) {
      h = H;
      w = W;
    } else {
      PH_REQUIRE(d.src0 < m->n_slots && plan.slots[d.src0].offset >= 0, "op reads slot %d before it is written", d.src0);
      h = plan.slots[d.src0].h;
      w = plan.slots[d.src0].w;
    }
    if (d.kind == PH_OP_HEAD) continue;
    if (d.kind == PH_OP_STEM) {
      PH_REQUIRE(d.src0 < 0, "stem op must read the network input");
      PH_REQUIRE(d.dst2 >= 0 && d.dst2 < m->n_slots && d.dst < m->n_slots, "bad stem slots");
      if (d.dst >= 0) {
        SlotShape& f = plan.slots[d.dst];
        f.c = d.cout; f.cp = fmt_cpad(fmt, 16); f.h = h; f.w = w; f.def_op = op_i;
        f.offset = alloc(d.dst, (int64_t)B * h * w * f.cp * bpc);
      }
      SlotShape& p = plan.slots[d.dst2];
      p.c = d.cout; p.cp = fmt_cpad(fmt, 16); p.h = (h + 1) / 2; p.w = (w + 1) / 2; p.def_op = op_i;
      p.offset = alloc(d.dst2, (int64_t)B * p.h * p.w * p.cp * bpc);
      continue;
    }
    int oh = h, ow = w;
    if (d.kind == PH_OP_PATCH_STEM) {
      PH_REQUIRE(d.src0 < 0, "patch stem must read the network input");
      oh = (h + 2 - d.ksize) / d.cmid + 1;
      ow = (w + 2 - d.ksize) / d.cmid + 1;
      PH_REQUIRE(oh > 0 && ow > 0, "input %dx%d is too small for the %dx%d patch stem", h, w, d.ksize, d.ksize);
    } else if (d.kind == PH_OP_PATCH_CONV) {
      oh = h / 2;
      ow = w / 2;
      PH_REQUIRE(oh > 0 && ow > 0, "input of a 2x2/stride-2 convolution is smaller than 2x2");
    } else if ((d.kind == PH_OP_LINEAR && (d.flags & PH_FLAG_SCALE_RESIDUAL)) || d.kind == PH_OP_SCALE_ADD) {
      PH_REQUIRE(d.src1 >= 0 && d.src1 < m->n_slots && plan.slots[d.src1].offset >= 0, "residual slot %d is not written yet", d.src1);
      const SlotShape& s1 = plan.slots[d.src1];
      PH_REQUIRE(s1.h == h && s1.w == w && s1.c == d.cout, "residual shape mismatch");
    }
    if (d.kind == PH_OP_GLOBAL_MAXPOOL) {
      oh = 1;
      ow = 1;
    }
    if (d.kind == PH_OP_POOL) {
      oh = (h + 1) / 2;
      ow = (w + 1) / 2;
    } else if (d.kind == PH_OP_UPSAMPLE || d.kind == PH_OP_CONVT) {
      oh = 2 * h;
      ow = 2 * w;
    }
    if (d.kind == PH_OP_CONV && d.src1 >= 0) {
      const SlotShape& s1 = plan.slots[d.src1];
      PH_REQUIRE(s1.offset >= 0, "op reads slot %d before it is written", d.src1);
      PH_REQUIRE(s1.h == h && s1.w == w,
                 "concat sources differ in size (%dx%d vs %dx%d): input H,W must be multiples of the model max stride",
                 h, w, s1.h, s1.w);
    }
    if (d.kind == PH_OP_CONVT) tmp = std::max<int64_t>(tmp, (int64_t)B * oh * ow * fmt_cpad(fmt, d.cin0) * bpc);
    if (d.kind == PH_OP_CONV && d.ksize == 3 && fmt == FMT_F32 && (m->conv_splitk >= 2 || (m->conv_splitk == 1 && reuse))) {  // (auto: inference plans only, as the kernel choice below)  // partial-sum planes of a split-K launch (conv3x3_wino2d_kernel<.., KS>)
      int n_cu = 0;
      if (device_cu_count(&n_cu) != PH_OK || n_cu <= 0) n_cu = 256;
      tmp = std::max<int64_t>(tmp, wino2d_split_scratch_bytes(B, h, w, pad16(d.cin0) + (d.src1 >= 0 ? pad16(d.cin1) : 0), pad16(d.cout), m->conv_splitk, n_cu));
      if (m->conv_wino4) tmp = std::max<int64_t>(tmp, wino4_split_scratch_bytes(B, h, w, pad16(d.cin0) + (d.src1 >= 0 ? pad16(d.cin1) : 0), pad16(d.cout), m->conv_splitk, n_cu));
    }
    PH_REQUIRE(d.dst >= 0 && d.dst < m->n_slots, "bad dst slot %d", d.dst);
    SlotShape& s = plan.slots[d.dst];
    s.c = (d.kind == PH_OP_POOL || d.kind == PH_OP_UPSAMPLE || d.kind == PH_OP_GLOBAL_MAXPOOL) ? d.cin0 : d.cout;
    s.cp = fmt_cpad(fmt, s.c);
    s.h = oh;
    s.w = ow;
    s.def_op = op_i;
    s.offset = alloc(d.dst, (int64_t)B * oh * ow * s.cp * bpc);
    if (d.kind == PH_OP_CONV && d.dst2 >= 0) {  // fused 2x2 max pool of the conv output
      PH_REQUIRE(d.dst2 < m->n_slots && (d.flags & PH_FLAG_RELU), "fused pool needs a valid slot and a ReLU conv");
      SlotShape& p = plan.slots[d.dst2];
      p.c = d.cout;
      p.cp = fmt_cpad(fmt, d.cout);
      p.h = (oh + 1) / 2;
      p.w = (ow + 1) / 2;
      p.def_op = op_i;
      p.offset = alloc(d.dst2, (int64_t)B * p.h * p.w * p.cp * bpc);
    }
  }
  plan.tmp_offset = off;
  plan.tmp_bytes = align_up(tmp, 256);
  plan.total = off + plan.tmp_bytes;
  return PH_OK;
}

// fp16 weight packs of every 3x3 conv / transposed conv, derived on the device from the fp32 LDS-DMA panels the first time a
// forward needs them (and again after every ph_model_set_params, through m->derived).  Allocates: not capturable -- a
// hipGraph user runs one eager forward first (HipBackend does).
int ensure_f16_weights(ph_model* m, int plain, hipStream_t s) {
  for (PackedOp& op : m->ops) {
    const ph_op_desc& d = op.d;
    if ((d.kind != PH_OP_CONV && d.kind != PH_OP_CONVT) || op.w_f16_dev[plain]) continue;
    PH_REQUIRE(op.w_dma_dev && (op.bn == 32 || op.bn == 64), "conv op has no LDS-DMA weight panels");
    const int chunks0 = pad16(d.cin0) / 16, chunks1 = d.cin1 > 0 ? pad16(d.cin1) / 16 : 0;
    const int n_tiles = (pad16(d.cout) + op.bn - 1) / op.bn;
    float* w = nullptr;
    PH_HIP_CHECK(hipMalloc(&w, (size_t)f16_weight_pack_floats(n_tiles, chunks0, chunks1, op.bn, plain) * sizeof(float)));
    m->allocs.push_back(w);
    int rc = launch_f16_weight_pack(op.w_dma_dev, w, n_tiles, chunks0, chunks1, op.bn, plain, s);
    if (rc != PH_OK) return rc;
    DerivedBuffer db;
    db.kind = 1;
    db.src = op.w_dma_dev;
    db.dst = w;
    db.n_tiles = n_tiles;
    db.chunks0 = chunks0;
    db.chunks1 = chunks1;
    db.bn = op.bn;
    db.plain = plain;
    m->derived.push_back(db);
    op.w_f16_dev[plain] = w;
  }
  return PH_OK;
}

static int drain_events(ph_model* m) {
  if (!m->events_pending) return PH_OK;
  const size_t n = m->ops.size();
  PH_HIP_CHECK(hipEventSynchronize(m->ev[n]));
  for (size_t i = 0; i < n; ++i) {
    float ms = 0.f;
    PH_HIP_CHECK(hipEventElapsedTime(&ms, m->ev[i], m->ev[i + 1]));
    m->op_ms[i] += ms;
  }
  m->profiled_forwards += 1;
  m->events_pending = false;
  return PH_OK;
}

}  // namespace ph

using namespace ph;

extern "C" {

const char* ph_last_error(void) { return ph::g_err.c_str(); }
int ph_version(void) { return PH_VERSION; }

ph_model* ph_model_create(const ph_op_desc* ops, int32_t n_ops, const float* const* weights, const int64_t* weight_numel,
                          int32_t n_weights, int32_t n_slots, int32_t n_outputs) {
  if (!ops || n_ops <= 0 || n_slots <= 0) {
    set_error("ph_model_create: empty program");
    return nullptr;
  }
  if (prepare_kernels() != PH_OK || prepare_convnext_kernels() != PH_OK || prepare_f16_kernels() != PH_OK) return nullptr;
  ph_model* m = new ph_model();
  m->n_slots = n_slots;
  m->n_outputs = n_outputs;
  {
    std::vector<float> z(64, 0.f);
    if (upload(m, z, &m->zeros_dev) != PH_OK) {
      ph_model_destroy(m);
      return nullptr;
    }
    std::vector<float> zc(3 * 4096, 0.f);  // (uploaded as floats: all-zero bits either way; arrivals | claimed shares | departures per split-K unit)
    float* cdev = nullptr;
    if (upload(m, zc, &cdev) != PH_OK) {
      ph_model_destroy(m);
      return nullptr;
    }
    m->split_counters_dev = reinterpret_cast<unsigned*>(cdev);
  }
  auto fail = [&](const char* msg, int i) -> ph_model* {
    set_error("ph_model_create: op %d: %s", i, msg);
    ph_model_destroy(m);
    return nullptr;
  };
  // canonical parameter arena = weights[] concatenated in the given order
  m->weight_offset.resize(n_weights);
  m->weight_numel.assign(weight_numel, weight_numel + n_weights);
  for (int i = 0; i < n_weights; ++i) {
    m->weight_offset[i] = m->n_params;
    m->n_params += weight_numel[i];
  }
  if (m->n_params >= (int64_t)1 << 31) return (set_error("models with >= 2^31 parameters are not supported"), ph_model_destroy(m), nullptr);
  auto index_array = [&](int wi) {
    std::vector<double> v((size_t)(wi >= 0 ? weight_numel[wi] : 0));
    for (size_t k = 0; k < v.size(); ++k) v[k] = (double)(m->weight_offset[wi] + (int64_t)k + 1);
    return v;
  };
  // zero-padded copy of a per-channel vector (bias, LayerNorm affine, layer scale)
  auto pad_vec = [](size_t padded, int n) {
    return [padded, n](const auto* src, auto& out) {
      out.assign(padded, 0);
      for (int k = 0; k < n; ++k) out[k] = src[k];
    };
  };
  for (int i = 0; i < n_ops; ++i) {
    PackedOp op;
    op.d = ops[i];
    const ph_op_desc& d = op.d;
    auto widx_ok = [&](int k) { return k >= 0 && k < n_weights; };
    bool ok = true;
    switch (d.kind) {
      case PH_OP_POOL:
      case PH_OP_UPSAMPLE:
      case PH_OP_GELU:
      case PH_OP_GLOBAL_MAXPOOL:
        break;
      case PH_OP_SCALE_ADD: {
        if (!widx_ok(d.weight) || weight_numel[d.weight] != d.cout || d.cin0 != d.cout || d.src1 < 0) return fail("scale-add needs a (C) scale and two sources", i);
        ok = pack_upload(m, pad_vec(pad16(d.cout), d.cout), weights[d.weight], index_array(d.weight), &op.w_dev) == PH_OK;
        break;
      }
      case PH_OP_STEM: {
        if (d.ksize != 3 || d.cmid < 1 || d.cmid > 16 || d.cout < 1 || d.cout > 16 || (d.cin0 != 1 && d.cin0 != 3))
          return fail("fused stem needs kernel 3, 1 or 3 input channels and <= 16 filters", i);
        if (!widx_ok(d.weight) || !widx_ok(d.bias) || !widx_ok(d.weight2) || !widx_ok(d.bias2)) return fail("weight index out of range", i);
        if (weight_numel[d.weight] != (int64_t)d.cmid * d.cin0 * 9 || weight_numel[d.bias] != d.cmid ||
            weight_numel[d.weight2] != (int64_t)d.cout * d.cmid * 9 || weight_numel[d.bias2] != d.cout)
          return fail("stem weight size mismatch", i);
        auto pack_w0 = [&](const auto* w0, auto& p0) {
          p0.assign((size_t)9 * d.cin0 * 16, 0);
          for (int co = 0; co < d.cmid; ++co)
            for (int ci = 0; ci < d.cin0; ++ci)
              for (int tap = 0; tap < 9; ++tap) p0[((size_t)tap * d.cin0 + ci) * 16 + co] = w0[((size_t)co * d.cin0 + ci) * 9 + tap];
        };
        auto pack_w1 = [&](const auto* w1, auto& p1) {
          p1.assign((size_t)9 * 16 * 16, 0);
          for (int co = 0; co < d.cout; ++co)
            for (int ci = 0; ci < d.cmid; ++ci)
              for (int tap = 0; tap < 9; ++tap) p1[((size_t)tap * 16 + co) * 16 + ci] = w1[((size_t)co * d.cmid + ci) * 9 + tap];
        };
        ok = pack_upload(m, pack_w0, weights[d.weight], index_array(d.weight), &op.w_dev) == PH_OK &&
             pack_upload(m, pad_vec(16, d.cmid), weights[d.bias], index_array(d.bias), &op.b_dev) == PH_OK &&
             pack_upload(m, pack_w1, weights[d.weight2], index_array(d.weight2), &op.w2_dev) == PH_OK &&
             pack_upload(m, pad_vec(16, d.cout), weights[d.bias2], index_array(d.bias2), &op.b2_dev) == PH_OK;
        if (ok) {  // Winograd form of the second conv, derived on the device (refreshed by ph_model_set_params)
          float* w = nullptr;
          ok = hipMalloc(&w, 12 * 256 * sizeof(float)) == hipSuccess;
          if (ok) {
            m->allocs.push_back(w);
            ok = launch_stem_wino_pack(op.w2_dev, w, nullptr) == PH_OK;
            DerivedBuffer db;
            db.src = op.w2_dev;
            db.dst = w;
            m->derived.push_back(db);
            op.w_wino_dev = w;
          }
        }
        if (ok) {  // ... and its F(2x2,3x3) form (the default)
          float* w = nullptr;
          ok = hipMalloc(&w, 16 * 256 * sizeof(float)) == hipSuccess;
          if (ok) {
            m->allocs.push_back(w);
            ok = launch_stem_wino2d_pack(op.w2_dev, w, nullptr) == PH_OK;
            DerivedBuffer db;
            db.src = op.w2_dev;
            db.dst = w;
            db.kind = 4;
            m->derived.push_back(db);
            op.w_stem2_dev = w;
          }
        }
        break;
      }
      case PH_OP_CONV:
      case PH_OP_CONVT: {
        if (!widx_ok(d.weight) || !widx_ok(d.bias)) return fail("weight index out of range", i);
        if (weight_numel[d.bias] != d.cout) return fail("bias size mismatch", i);
        if (d.kind == PH_OP_CONVT && d.ksize != 3) return fail("transposed conv: only kernel 3 (stride 2, padding 1, output padding 1)", i);
        if (d.ksize != 3) {
          // kernel_size 5 / 7 / 9 (and the 7x7 convs of a stem block): k x k "same" conv as a k^2-tap row GEMM (mode 5); inference only
          if (!(d.ksize & 1) || d.ksize < 1 || d.ksize > 9) return fail("kernel_size must be odd and <= 9", i);
          if (weight_numel[d.weight] != (int64_t)(d.cin0 + d.cin1) * d.cout * d.ksize * d.ksize) return fail("weight size mismatch", i);
          const int coutp_k = pad16(d.cout);
          op.bn_g = gemm_choose_bn(coutp_k);
          auto pack_k = [&](const auto* w, auto& out) { pack_gemm(w, d.cout, d.cin0, d.cin1, d.ksize * d.ksize, op.bn_g, out); };
          ok = pack_upload(m, pack_k, weights[d.weight], index_array(d.weight), &op.w_gemm_dev) == PH_OK &&
               pack_upload(m, pad_vec((size_t)((coutp_k + op.bn_g - 1) / op.bn_g) * op.bn_g, d.cout), weights[d.bias], index_array(d.bias), &op.b_gemm_dev) == PH_OK;
          if (ok && d.kind == PH_OP_CONV) {  // data-gradient weights (training): the same k x k row GEMM on the flipped, in/out-swapped taps, one per concat source
            const int cin_total = d.cin0 + d.cin1;
            const int parts[2] = {d.cin0, d.cin1}, offs[2] = {0, d.cin0};
            int max_bn = 32;
            const std::vector<double> iwk = index_array(d.weight);
            for (int part = 0; part < 2 && ok; ++part) {
              if (parts[part] <= 0) continue;
              op.bn_dk[part] = gemm_choose_bn(pad16(parts[part]));
              max_bn = std::max(max_bn, op.bn_dk[part]);
              auto pack_dk = [&](const auto* w, auto& out) {
                std::remove_reference_t<decltype(out)> dw;
                dgrad_weight(w, d.cout, cin_total, offs[part], parts[part], dw, d.ksize);
                pack_gemm(dw.data(), parts[part], d.cout, 0, d.ksize * d.ksize, op.bn_dk[part], out);
              };
              ok = pack_upload(m, pack_dk, weights[d.weight], iwk, &op.wdk_gemm_dev[part]) == PH_OK;
            }
            if (ok) {
              std::vector<float> zb((size_t)pad16(std::max(d.cin0, d.cin1)) + 2 * max_bn, 0.f);
              ok = upload(m, zb, &op.zero_bias_dev) == PH_OK;
            }
          }
          break;
        }
        if (d.kind == PH_OP_CONVT && d.cin1 != 0) return fail("transposed conv takes one source", i);
        if (weight_numel[d.weight] != (int64_t)(d.cin0 + d.cin1) * d.cout * 9) return fail("weight size mismatch", i);
        const int coutp = pad16(d.cout);
        op.bn = choose_bn(coutp);
        const bool tr = d.kind == PH_OP_CONVT;
        const std::vector<double> iw = index_array(d.weight), ib = index_array(d.bias);
        auto pack_reg = [&](const auto* w, auto& out) { pack_conv(w, tr, d.cin0, d.cin1, d.cout, op.bn, out); };
        auto pack_dma = [&](const auto* w, auto& out) {
          std::remove_reference_t<decltype(out)> tmp;
          pack_conv(w, tr, d.cin0, d.cin1, d.cout, op.bn, tmp);
          repack_dma(tmp, op.bn, out);
        };
        ok = pack_upload(m, pack_reg, weights[d.weight], iw, &op.w_dev) == PH_OK && pack_upload(m, pack_dma, weights[d.weight], iw, &op.w_dma_dev) == PH_OK &&
             pack_upload(m, pad_vec((size_t)((coutp + op.bn - 1) / op.bn) * op.bn, d.cout), weights[d.bias], ib, &op.b_dev) == PH_OK;
        // Winograd F(2,3) weights of an N-tile-64 conv: a linear combination of taps, so not a gather of parameters --
        // computed on the device from the packed [tap] panels, again after every ph_model_set_params
        auto derive_wino = [&](const float* src, int cin_a, int cin_b, int cout_, int bn, float** dst) {
          if (bn != 64 && bn != 32) return true;
          const int panels = ((pad16(cout_) + bn - 1) / bn) * (pad16(cin_a) / 16 + (cin_b > 0 ? pad16(cin_b) / 16 : 0));
          float* w = nullptr;
          if (hipMalloc(&w, (size_t)wino_pack_floats(panels, bn) * sizeof(float)) != hipSuccess) return false;
          m->allocs.push_back(w);
          if (launch_wino_pack(src, w, panels, bn, nullptr) != PH_OK) return false;
          DerivedBuffer db;
          db.src = src;
          db.dst = w;
          db.panels = panels;
          db.bn = bn;
          m->derived.push_back(db);
          *dst = w;
          return true;
        };
        if (ok && !tr) ok = derive_wino(op.w_dev, d.cin0, d.cin1, d.cout, op.bn, &op.w_wino_dev);
        // F(2x2,3x3) weights (conv3x3_wino2d_kernel, N tile 64): the forward conv's, and below the data-gradient convs'
        auto derive_wino2 = [&](const float* src, int cin_a, int cin_b, int cout_, int bn, float** dst) {
          if (bn != 64) return true;
          const int panels = ((pad16(cout_) + 63) / 64) * (pad16(cin_a) / 16 + (cin_b > 0 ? pad16(cin_b) / 16 : 0));
          float* w = nullptr;
          if (hipMalloc(&w, (size_t)wino2d_pack_floats(panels, 64) * sizeof(float)) != hipSuccess) return false;
          m->allocs.push_back(w);
          if (launch_wino2d_pack(src, w, panels, 64, nullptr) != PH_OK) return false;
          DerivedBuffer db;
          db.src = src;
          db.dst = w;
          db.panels = panels;
          db.bn = 64;
          db.kind = 2;
          m->derived.push_back(db);
          *dst = w;
          return true;
        };
        if (ok && !tr) ok = derive_wino2(op.w_dev, d.cin0, d.cin1, d.cout, op.bn, &op.w_wino2_dev);
        if (ok && !tr && op.bn == 32 && coutp == 32 && pad16(d.cin0) + (d.cin1 > 0 ? pad16(d.cin1) : 0) >= 64) {
          // 32 output channels, K >= 64: a second packing with an N tile of 64 (rows 32 .. 63 zero) for the F(2x2,3x3) kernel's half-empty-tile form (conv_n32_wino2d)
          auto pack_64 = [&](const auto* w, auto& out) { pack_conv(w, false, d.cin0, d.cin1, d.cout, 64, out); };
          ok = pack_upload(m, pack_64, weights[d.weight], iw, &op.w_n64_dev) == PH_OK && pack_upload(m, pad_vec(64, d.cout), weights[d.bias], ib, &op.b_n64_dev) == PH_OK &&
               derive_wino2(op.w_n64_dev, d.cin0, d.cin1, d.cout, 64, &op.w_wino2_dev);
        }
        // F(4x4,3x3) weights (conv3x3_wino4_kernel) of the layers with many input channels: 4x the direct weights' bytes
        auto derive_wino4 = [&](const float* src, int cin_a, int cin_b, int cout_, int bn, float** dst) {
          if (bn != 64 || pad16(cin_a) + (cin_b > 0 ? pad16(cin_b) : 0) < 64) return true;  // (derived from 64 input channels on; which layers take the kernel is conv_wino4_min_cin + the time model)
          const int ntiles = (pad16(cout_) + 63) / 64, nchunks = pad16(cin_a) / 16 + (cin_b > 0 ? pad16(cin_b) / 16 : 0);
          float* w = nullptr;
          if (hipMalloc(&w, (size_t)wino4_pack_floats(ntiles, nchunks) * sizeof(float)) != hipSuccess) return false;
          m->allocs.push_back(w);
          if (launch_wino4_pack(src, w, ntiles, nchunks, nullptr) != PH_OK) return false;
          DerivedBuffer db;
          db.src = src;
          db.dst = w;
          db.panels = ntiles;
          db.bn = nchunks;
          db.kind = 5;
          m->derived.push_back(db);
          *dst = w;
          return true;
        };
        if (ok && !tr) ok = derive_wino4(op.w_dev, d.cin0, d.cin1, d.cout, op.bn, &op.w_wino4_dev);
        if (ok && !tr && op.w_n64_dev) ok = derive_wino4(op.w_n64_dev, d.cin0, d.cin1, d.cout, 64, &op.w_wino4_dev);  // (Cout 32, K >= 64: the N-tile-64 packing above; the kernel's waves of the empty N half skip their MFMAs)
        // wave-private F(2x2,3x3) weights (conv3x3_w16_kernel: 16 / 32 output channels, 16 / 32 input channels, one source): the forward
        // conv's, and below the data-gradient convs'
        auto derive_w16 = [&](const float* src, int cin_, int cout_, int bn, float** dst, int cin_b = 0) {
          const int cip = pad16(cin_), cibp = cin_b > 0 ? pad16(cin_b) : 0, cop = pad16(cout_);
          if (bn != 32 || !w16_shape_ok(cip, cibp, cop)) return true;
          const int chunks = (cip + cibp) / 16, nbs = cop / 16;
          float* w = nullptr;
          if (hipMalloc(&w, (size_t)w16_pack_floats(chunks, nbs) * sizeof(float)) != hipSuccess) return false;
          m->allocs.push_back(w);
          if (launch_w16_pack(src, w, chunks, nbs, nullptr) != PH_OK) return false;
          DerivedBuffer db;
          db.src = src;
          db.dst = w;
          db.panels = chunks;
          db.bn = nbs;
          db.kind = 3;
          m->derived.push_back(db);
          *dst = w;
          return true;
        };
        if (ok && !tr) ok = derive_w16(op.w_dev, d.cin0, d.cout, op.bn, &op.w_w16_dev, d.cin1);
        if (ok && !tr) {  // small-map F(2x2,3x3) weights (conv3x3_sm_kernel): 16/9 of the direct weights' bytes, every 3x3 conv
          const int nblocks = coutp / 16, chunks16 = pad16(d.cin0) / 16 + (d.cin1 > 0 ? pad16(d.cin1) / 16 : 0);
          float* w = nullptr;
          ok = hipMalloc(&w, (size_t)sm_pack_floats(nblocks, chunks16) * sizeof(float)) == hipSuccess;
          if (ok) {
            m->allocs.push_back(w);
            ok = launch_sm_pack(op.w_dev, w, nblocks, chunks16, op.bn, nullptr) == PH_OK;
            DerivedBuffer db;
            db.src = op.w_dev;
            db.dst = w;
            db.panels = nblocks;
            db.bn = chunks16;
            db.n_tiles = op.bn;
            db.kind = 6;
            m->derived.push_back(db);
            op.w_sm_dev = w;
          }
        }
        if (ok && !tr) {  // row-GEMM form for feature maps too small for the 16x32-pixel tiles
          op.bn_g = gemm_choose_bn(coutp);
          auto pack_g = [&](const auto* w, auto& out) { pack_gemm(w, d.cout, d.cin0, d.cin1, 9, op.bn_g, out); };
          ok = pack_upload(m, pack_g, weights[d.weight], iw, &op.w_gemm_dev) == PH_OK &&
               pack_upload(m, pad_vec((size_t)((coutp + op.bn_g - 1) / op.bn_g) * op.bn_g, d.cout), weights[d.bias], ib, &op.b_gemm_dev) == PH_OK;
        }
        const bool c16 = d.kind == PH_OP_CONV && d.cin1 == 0 && d.cin0 <= 16 && d.cout <= 16;
        if (ok && c16) {
          auto pack_16 = [&](const auto* w, auto& out) { pack_c16(w, d.cout, d.cin0, out); };
          auto pack_16d = [&](const auto* w, auto& out) {
            std::remove_reference_t<decltype(out)> dw;
            dgrad_weight(w, d.cout, d.cin0, 0, d.cin0, dw);
            pack_c16(dw.data(), d.cin0, d.cout, out);
          };
          ok = pack_upload(m, pack_16, weights[d.weight], iw, &op.w16_dev) == PH_OK && pack_upload(m, pack_16d, weights[d.weight], iw, &op.wd16_dev) == PH_OK;
        }
        if (ok && tr) {
          // ConvTranspose2d(k3, s2, p1, op1): output (oy, ox) = (2y - 1 + ky, 2x - 1 + kx).  Even output rows see only ky = 1
          // (input row y), odd rows ky = 0 (input row y + 1) and ky = 2 (row y) -- likewise in x: four output phases with
          // 1, 2, 2, 4 taps.  Each phase is one row GEMM over the input grid (mode 3) whose tap t reads input pixel
          // (y + dy, x + dx); weights per phase: wp[co][ci][t] = Wt[ci][co][ky(t)][kx(t)].
          op.bn_t = gemm_choose_bn(coutp);
          const size_t npad_t = (size_t)((coutp + op.bn_t - 1) / op.bn_t) * op.bn_t;
          for (int ph = 0; ph < 4 && ok; ++ph) {
            const int py = ph >> 1, px = ph & 1, nt = (1 + py) * (1 + px);
            auto pack_p = [&](const auto* w, auto& out) {
              std::remove_reference_t<decltype(out)> wp((size_t)d.cout * d.cin0 * nt);
              for (int t = 0; t < nt; ++t) {
                const int ty = px ? t >> 1 : t, tx = px ? t & 1 : 0;
                const int ky = py ? 2 * ty : 1, kx = px ? 2 * tx : 1;  // tap t reads (y + 1 - ty, x + 1 - tx): ky = 0 <-> dy = 1
                for (int co = 0; co < d.cout; ++co)
                  for (int ci = 0; ci < d.cin0; ++ci) wp[((size_t)co * d.cin0 + ci) * nt + t] = w[(((size_t)ci * d.cout + co) * 3 + ky) * 3 + kx];
              }
              pack_gemm(wp.data(), d.cout, d.cin0, 0, nt, op.bn_t, out);
            };
            ok = pack_upload(m, pack_p, weights[d.weight], iw, &op.wt_phase_dev[ph]) == PH_OK;
          }
          if (ok) ok = pack_upload(m, pad_vec(npad_t, d.cout), weights[d.bias], ib, &op.bt_dev) == PH_OK;
          if (ok && d.weight2 >= 0) {  // folded BatchNorm (eval): y = act(scale * (conv + bias) + shift)
            if (!widx_ok(d.weight2) || !widx_ok(d.bias2) || weight_numel[d.weight2] != d.cout || weight_numel[d.bias2] != d.cout) return fail("transposed conv: scale / shift must be (cout)", i);
            ok = pack_upload(m, pad_vec(npad_t, d.cout), weights[d.weight2], index_array(d.weight2), &op.wt_scale_dev) == PH_OK &&
                 pack_upload(m, pad_vec(npad_t, d.cout), weights[d.bias2], index_array(d.bias2), &op.wt_shift_dev) == PH_OK;
          }
          if (ok) {  // data gradient = Conv2d(dY, Wt as (out = cin0, in = cout, 3, 3), stride 2, pad 1): the ConvTranspose2d weight as it lies in memory
            const int cinp = pad16(d.cin0);
            op.bn_td = gemm_choose_bn(cinp);
            auto pack_dg = [&](const auto* w, auto& out) { pack_gemm(w, d.cin0, d.cout, 0, 9, op.bn_td, out); };
            std::vector<float> zb((size_t)cinp + 128, 0.f);
            ok = pack_upload(m, pack_dg, weights[d.weight], iw, &op.wt_dgrad_dev) == PH_OK && upload(m, zb, &op.zero_bias_dev) == PH_OK;
          }
        }
        if (ok && d.kind == PH_OP_CONV) {  // data-gradient weights (training)
          const int cin_total = d.cin0 + d.cin1;
          const int parts[2] = {d.cin0, d.cin1}, offs[2] = {0, d.cin0};
          int max_bn = 32;
          for (int part = 0; part < 2 && ok; ++part) {
            if (parts[part] <= 0) continue;
            op.bn_d[part] = choose_bn(pad16(parts[part]));
            max_bn = std::max(max_bn, op.bn_d[part]);
            auto pack_d = [&](const auto* w, auto& out) {
              std::remove_reference_t<decltype(out)> dw;
              dgrad_weight(w, d.cout, cin_total, offs[part], parts[part], dw);
              pack_conv(dw.data(), false, d.cout, 0, parts[part], op.bn_d[part], out);
            };
            auto pack_dd = [&](const auto* w, auto& out) {
              std::remove_reference_t<decltype(out)> tmp;
              pack_d(w, tmp);
              repack_dma(tmp, op.bn_d[part], out);
            };
            ok = pack_upload(m, pack_d, weights[d.weight], iw, &op.wd_dev[part]) == PH_OK &&
                 pack_upload(m, pack_dd, weights[d.weight], iw, &op.wd_dma_dev[part]) == PH_OK &&
                 derive_wino(op.wd_dev[part], d.cout, 0, parts[part], op.bn_d[part], &op.wd_wino_dev[part]) &&
                 derive_wino2(op.wd_dev[part], d.cout, 0, parts[part], op.bn_d[part], &op.wd_wino2_dev[part]) &&
                 derive_w16(op.wd_dev[part], d.cout, parts[part], op.bn_d[part], &op.wd_w16_dev[part]) &&
                 derive_wino4(op.wd_dev[part], d.cout, 0, parts[part], op.bn_d[part], &op.wd_wino4_dev[part]);
          }
          if (ok) {
            std::vector<float> zb((size_t)pad16(std::max(d.cin0, d.cin1)) + max_bn, 0.f);
            ok = upload(m, zb, &op.zero_bias_dev) == PH_OK;
          }
        }
        break;
      }
      case PH_OP_INPUT_CONV:
      case PH_OP_PATCH_STEM: {
        // [tap][ci][Cp] from OIHW (cout, cin, k, k); the patch stem (k x k, stride = cmid, padding 1) shares the layout
        if (!widx_ok(d.weight) || !widx_ok(d.bias)) return fail("weight index out of range", i);
        if (weight_numel[d.bias] != d.cout) return fail("bias size mismatch", i);
        if (d.kind == PH_OP_INPUT_CONV && (!(d.ksize & 1) || d.ksize < 1 || d.ksize > 9)) return fail("kernel_size must be odd and <= 9", i);
        if (d.kind == PH_OP_PATCH_STEM && (d.ksize < 2 || d.ksize > 8 || d.cmid < 1 || d.cmid > d.ksize)) return fail("patch stem needs 2 <= kernel <= 8 and 1 <= stride <= kernel", i);
        const int kk = d.ksize * d.ksize;
        if (weight_numel[d.weight] != (int64_t)d.cin0 * d.cout * kk) return fail("weight size mismatch", i);
        const int coutp = pad16(d.cout);
        auto pack_in = [&](const auto* w, auto& out) {
          out.assign((size_t)kk * d.cin0 * coutp, 0);
          for (int co = 0; co < d.cout; ++co)
            for (int ci = 0; ci < d.cin0; ++ci)
              for (int tap = 0; tap < kk; ++tap) out[((size_t)tap * d.cin0 + ci) * coutp + co] = w[((size_t)co * d.cin0 + ci) * kk + tap];
        };
        ok = pack_upload(m, pack_in, weights[d.weight], index_array(d.weight), &op.w_dev) == PH_OK &&
             pack_upload(m, pad_vec(coutp, d.cout), weights[d.bias], index_array(d.bias), &op.b_dev) == PH_OK;
        break;
      }
      case PH_OP_HEAD: {
        if (!widx_ok(d.weight) || !widx_ok(d.bias)) return fail("weight index out of range", i);
        if (weight_numel[d.bias] != d.cout) return fail("bias size mismatch", i);
        if (weight_numel[d.weight] != (int64_t)d.cin0 * d.cout) return fail("weight size mismatch", i);
        if (d.out_index < 0 || d.out_index >= n_outputs) return fail("bad out_index", i);
        const int cp = pad16(d.cin0);
        auto pack_head = [&](const auto* w, auto& out) {
          out.assign((size_t)d.cout * cp, 0);
          for (int co = 0; co < d.cout; ++co)
            for (int ci = 0; ci < d.cin0; ++ci) out[(size_t)co * cp + ci] = w[(size_t)co * d.cin0 + ci];
        };
        ok = pack_upload(m, pack_head, weights[d.weight], index_array(d.weight), &op.w_dev) == PH_OK &&
             pack_upload(m, pad_vec(d.cout, d.cout), weights[d.bias], index_array(d.bias), &op.b_dev) == PH_OK;
        break;
      }
      case PH_OP_DWCONV: {
        // (C, 1, 7, 7) -> [tap][Cp]
        if (!widx_ok(d.weight) || !widx_ok(d.bias)) return fail("weight index out of range", i);
        if (d.ksize != 7 || d.cin0 != d.cout) return fail("depthwise conv must be 7x7 with cin == cout", i);
        if (weight_numel[d.weight] != (int64_t)d.cout * 49 || weight_numel[d.bias] != d.cout) return fail("weight size mismatch", i);
        const int cp = pad16(d.cout);
        auto pack_dw = [&](const auto* w, auto& out) {
          out.assign((size_t)49 * cp, 0);
          for (int c = 0; c < d.cout; ++c)
            for (int tap = 0; tap < 49; ++tap) out[(size_t)tap * cp + c] = w[(size_t)c * 49 + tap];
        };
        auto pack_dw_flip = [&](const auto* w, auto& out) {  // data gradient = the same kernel on the spatially flipped taps
          out.assign((size_t)49 * cp, 0);
          for (int c = 0; c < d.cout; ++c)
            for (int tap = 0; tap < 49; ++tap) out[(size_t)(48 - tap) * cp + c] = w[(size_t)c * 49 + tap];
        };
        ok = pack_upload(m, pack_dw, weights[d.weight], index_array(d.weight), &op.w_dev) == PH_OK &&
             pack_upload(m, pad_vec(cp, d.cout), weights[d.bias], index_array(d.bias), &op.b_dev) == PH_OK &&
             pack_upload(m, pack_dw_flip, weights[d.weight], index_array(d.weight), &op.dw_flip_dev) == PH_OK;
        break;
      }
      case PH_OP_LAYERNORM: {
        if (!widx_ok(d.weight) || !widx_ok(d.bias)) return fail("weight index out of range", i);
        if (d.cin0 != d.cout || weight_numel[d.weight] != d.cout || weight_numel[d.bias] != d.cout) return fail("LayerNorm affine size mismatch", i);
        const int cp = pad16(d.cout);
        ok = pack_upload(m, pad_vec(cp, d.cout), weights[d.weight], index_array(d.weight), &op.w_dev) == PH_OK &&
             pack_upload(m, pad_vec(cp, d.cout), weights[d.bias], index_array(d.bias), &op.b_dev) == PH_OK;
        break;
      }
      case PH_OP_LINEAR:
      case PH_OP_PATCH_CONV: {
        if (!widx_ok(d.weight) || !widx_ok(d.bias)) return fail("weight index out of range", i);
        const int segs = d.kind == PH_OP_PATCH_CONV ? 4 : 1;
        if (d.kind == PH_OP_PATCH_CONV && d.ksize != 2) return fail("patch conv must be 2x2 stride 2", i);
        if (weight_numel[d.weight] != (int64_t)d.cin0 * d.cout * segs || weight_numel[d.bias] != d.cout) return fail("weight size mismatch", i);
        const int coutp = pad16(d.cout);
        op.bn = gemm_choose_bn(coutp);
        const size_t npad = (size_t)((coutp + op.bn - 1) / op.bn) * op.bn;
        auto pack_g = [&](const auto* w, auto& out) { pack_gemm(w, d.cout, d.cin0, 0, segs, op.bn, out); };
        ok = pack_upload(m, pack_g, weights[d.weight], index_array(d.weight), &op.w_dma_dev) == PH_OK &&
             pack_upload(m, pad_vec(npad, d.cout), weights[d.bias], index_array(d.bias), &op.b_dev) == PH_OK;
        if (ok && d.kind == PH_OP_LINEAR && (d.flags & PH_FLAG_GELU) && cnblock_mlp_fits(d.cin0, pad16(d.cin0), d.cout)) {  // inference programs: the first Linear of a CNBlock MLP ...
          auto pack_1 = [&](const auto* w, auto& out) { pack_mlp_w1(w, d.cin0, out); };
          ok = pack_upload(m, pack_1, weights[d.weight], index_array(d.weight), &op.w_mlp_dev) == PH_OK;
        } else if (ok && d.kind == PH_OP_LINEAR && (d.flags & PH_FLAG_SCALE_RESIDUAL) && cnblock_mlp_fits(d.cout, pad16(d.cout), d.cin0)) {  // ... and the second
          auto pack_2 = [&](const auto* w, auto& out) { pack_mlp_w2(w, d.cout, out); };
          ok = pack_upload(m, pack_2, weights[d.weight], index_array(d.weight), &op.w_mlp_dev) == PH_OK;
        }
        if (ok) {  // data-gradient weights (training): dX = dY * W, one transposed matrix per tap, + a zero bias
          const int cinp = pad16(d.cin0);
          op.bn_dg = gemm_choose_bn(cinp);
          for (int tap = 0; tap < segs && ok; ++tap) {
            auto pack_t = [&](const auto* w, auto& out) {
              std::remove_reference_t<decltype(out)> wt((size_t)d.cin0 * d.cout);
              for (int co = 0; co < d.cout; ++co)
                for (int ci = 0; ci < d.cin0; ++ci) wt[(size_t)ci * d.cout + co] = w[((size_t)co * d.cin0 + ci) * segs + tap];
              pack_gemm(wt.data(), d.cin0, d.cout, 0, 1, op.bn_dg, out);
            };
            ok = pack_upload(m, pack_t, weights[d.weight], index_array(d.weight), &op.wd_gemm_dev[tap]) == PH_OK;
          }
          if (ok) {
            std::vector<float> zb((size_t)cinp + 128, 0.f);
            ok = upload(m, zb, &op.zero_bias_dev) == PH_OK;
          }
        }
        if (ok && (d.flags & PH_FLAG_SCALE_RESIDUAL)) {
          if (!widx_ok(d.weight2) || weight_numel[d.weight2] != d.cout || d.src1 < 0) return fail("layer-scale epilogue needs weight2 (C) and a residual source", i);
          ok = pack_upload(m, pad_vec(npad, d.cout), weights[d.weight2], index_array(d.weight2), &op.w2_dev) == PH_OK;
        }
        break;
      }
      default:
        return fail("unknown op kind", i);
    }
    if (!ok) {
      ph_model_destroy(m);
      return nullptr;
    }
    m->ops.push_back(op);
  }
  if (!m->derived.empty() && hipDeviceSynchronize() != hipSuccess) {  // the derived buffers were computed on the null stream
    set_error("ph_model_create: device synchronisation failed");
    ph_model_destroy(m);
    return nullptr;
  }
  return m;
}

void ph_model_destroy(ph_model* m) {
  if (!m) return;
  for (void* p : m->allocs) (void)hipFree(p);
  for (hipEvent_t e : m->ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : m->comm_events_owned) (void)hipEventDestroy(e);
  delete m;
}

int64_t ph_model_workspace_bytes(const ph_model* m, int32_t batch, int32_t height, int32_t width) {
  if (!m || batch <= 0 || height <= 0 || width <= 0) {
    set_error("ph_model_workspace_bytes: bad arguments");
    return PH_E_INVALID;
  }
  Plan plan;
  int rc = build_plan(m, batch, height, width, plan, forward_format(m));
  if (rc != PH_OK) return rc;
  return plan.total;
}

int ph_model_output_shape(const ph_model* m, int32_t out_index, int32_t height, int32_t width, int32_t* c, int32_t* h, int32_t* w) {
  PH_REQUIRE(m && c && h && w, "ph_model_output_shape: null argument");
  Plan plan;
  int rc = build_plan(m, 1, height, width, plan);
  if (rc != PH_OK) return rc;
  for (const PackedOp& op : m->ops)
    if (op.d.kind == PH_OP_HEAD && op.d.out_index == out_index) {
      *c = op.d.cout;
      *h = plan.slots[op.d.src0].h;
      *w = plan.slots[op.d.src0].w;
      return PH_OK;
    }
  set_error("no head writes output %d", out_index);
  return PH_E_INVALID;
}

int ph_model_forward(ph_model* m, const void* input_dev, int32_t in_dtype, int32_t batch, int32_t in_channels, int32_t height,
                     int32_t width, void* workspace_dev, int64_t workspace_bytes, float* const* out_dev, void* stream) {
  PH_REQUIRE(m && input_dev && workspace_dev && out_dev, "ph_model_forward: null argument");
  PH_REQUIRE(in_dtype >= 0 && in_dtype <= 2, "ph_model_forward: bad in_dtype %d", in_dtype);
  PH_REQUIRE(((uintptr_t)workspace_dev & 255) == 0, "workspace must be 256-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  Plan plan;
  const int fmt = forward_format(m);
  int rc = build_plan(m, batch, height, width, plan, fmt);
  if (rc != PH_OK) return rc;
  if (fmt != FMT_F32) {
    rc = ensure_f16_weights(m, fmt == FMT_F16 ? 1 : 0, s);
    if (rc != PH_OK) return rc;
  }
  const int rs_div = 4 / plan.bpc;  // pixel stride in 4-byte units = cp / rs_div
  if (plan.total > workspace_bytes) {
    set_error("workspace too small: need %lld bytes, got %lld", (long long)plan.total, (long long)workspace_bytes);
    return PH_E_WORKSPACE;
  }
  char* ws = static_cast<char*>(workspace_dev);
  auto slot_ptr = [&](int sidx) { return reinterpret_cast<float*>(ws + plan.slots[sidx].offset); };
  if (m->profiling) {
    rc = drain_events(m);
    if (rc != PH_OK) return rc;
  }
  if (m->wino4_stale && (m->conv_wino4 >= 2 || (m->conv_wino4 == 1 && plan.reuse) || m->conv_smallmap >= 2 || (m->conv_smallmap == 1 && plan.reuse))) {  // (the inference-only weight forms: F(4x4,3x3), small-map)
    for (const DerivedBuffer& db : m->derived)
      if (db.kind == 5 || db.kind == 6) {
        rc = db.kind == 5 ? launch_wino4_pack(db.src, db.dst, db.panels, db.bn, s) : launch_sm_pack(db.src, db.dst, db.panels, db.bn, db.n_tiles, s);
        if (rc != PH_OK) return rc;
      }
    m->wino4_stale = false;
  }
  size_t op_index = 0;
  m->last_variant.assign(m->ops.size(), PH_KV_NONE);
  int* const kv = m->last_variant.data();
  bool skip_next_gelu = false, skip_next_scale_add = false, skip_next_linear = false;
  int pooled_by_conv = -1;  // slot whose 2x2 max pool the producing conv's epilogue already wrote (run-time fusion of an unfused program, see PH_OP_CONV)
  std::vector<char> head_done(m->ops.size(), 0);  // head ops the producing conv's epilogue already computed (see PH_OP_CONV)
  std::vector<char> conv_done(m->ops.size(), 0);  // conv ops the conv in front of them already computed (two-conv block kernel, see PH_OP_CONV)
  int deferred_up = -1;     // index of a bilinear op left to the conv that follows it (see PH_OP_UPSAMPLE / PH_OP_CONV)
  int fused_ln = -1;        // index of a LayerNorm op the producing depthwise conv already applied (see PH_OP_DWCONV)
  for (const PackedOp& op : m->ops) {
    const ph_op_desc& d = op.d;
    if (m->profiling) PH_HIP_CHECK(hipEventRecord(m->ev[op_index], s));
    ++op_index;
    switch (d.kind) {
      case PH_OP_INPUT_CONV: {
        PH_REQUIRE(d.cin0 == in_channels, "input has %d channels, network expects %d", in_channels, d.cin0);
        InputConvArgs a{};
        a.src = input_dev;
        a.w = op.w_dev;
        a.bias = op.b_dev;
        a.dst = slot_ptr(d.dst);
        a.dtype = in_dtype;
        a.cin = d.cin0;
        a.coutp = pad16(d.cout);
        a.B = batch;
        a.H = height;
        a.W = width;
        a.relu = (d.flags & PH_FLAG_RELU) ? 1 : 0;
        a.ksize = d.ksize;
        a.out_fmt = fmt;
        a.dst_cp = plan.slots[d.dst].cp;
        rc = launch_input_conv(a, s);
        break;
      }
      case PH_OP_STEM: {
        PH_REQUIRE(d.cin0 == in_channels, "input has %d channels, network expects %d", in_channels, d.cin0);
        StemArgs a{};
        a.src = input_dev;
        a.w0 = op.w_dev;
        a.b0 = op.b_dev;
        a.w1 = op.w2_dev;
        a.w1w = op.w_wino_dev;
        a.w1w2 = op.w_stem2_dev;
        a.b1 = op.b2_dev;
        a.dst_full = d.dst >= 0 ? slot_ptr(d.dst) : nullptr;
        a.dst_pool = slot_ptr(d.dst2);
        a.dtype = in_dtype;
        a.cin = d.cin0;
        a.B = batch;
        a.H = height;
        a.W = width;
        a.wino = m->stem_wino;
        a.f16_mfma = m->stem_f16mfma;
        a.out_fmt = fmt;
        kv[op_index - 1] = PH_KV_STEM;
        rc = launch_stem(a, s);
        break;
      }
      case PH_OP_CONV: {
        const SlotShape& s0 = plan.slots[d.src0];
        if (conv_done[op_index - 1]) {  // computed by the conv in front of it (block2_c32_f16_kernel, below)
          kv[op_index - 1] = PH_KV_FUSED;
          break;
        }
        if (fmt != FMT_F32) {
          PH_REQUIRE(s0.c == d.cin0 && (d.src1 < 0 || plan.slots[d.src1].c == d.cin1), "conv channel mismatch");
          const SlotShape& so = plan.slots[d.dst];
          if (fmt == FMT_F16 && m->block_fuse && plan.reuse && d.src1 < 0 && d.ksize == 3 && d.dst2 < 0 && op.bn == 32 && so.cp == 32 && s0.cp == 32 && d.cin0 <= 16 &&
              op_index < m->ops.size()) {
            // inference plans, plain fp16: conv(<= 16 -> 32) + ReLU whose only reader is the next op, a conv(32 -> 32) (+ ReLU, + pool): both in ONE launch, the intermediate
            // tensor stays in LDS (block2_c32_f16_kernel)
            const PackedOp& nxo = m->ops[op_index];
            const ph_op_desc& nx = nxo.d;
            bool fuse = nx.kind == PH_OP_CONV && nx.ksize == 3 && nx.src0 == d.dst && nx.src1 < 0 && nxo.bn == 32 && plan.slots[nx.dst].cp == 32 && nx.cin0 == d.cout && nxo.w_f16_dev[1] &&
                        op.w_f16_dev[1];
            for (size_t k = 0; fuse && k < m->ops.size(); ++k)
              if (k != op_index && (m->ops[k].d.src0 == d.dst || m->ops[k].d.src1 == d.dst)) fuse = false;
            if (fuse) {
              Block2Args b2{};
              b2.src = slot_ptr(d.src0);
              b2.wa = op.w_f16_dev[1];
              b2.wb = nxo.w_f16_dev[1];
              b2.ba = op.b_dev;
              b2.bb = nxo.b_dev;
              const bool full_unread = nx.dst2 >= 0 && !plan.unread.empty() && plan.unread[nx.dst];
              b2.dst_full = full_unread ? nullptr : slot_ptr(nx.dst);
              b2.dst_pool = nx.dst2 >= 0 ? slot_ptr(nx.dst2) : nullptr;
              b2.B = batch;
              b2.H = s0.h;
              b2.W = s0.w;
              b2.relu_a = (d.flags & PH_FLAG_RELU) ? 1 : 0;
              b2.relu_b = (nx.flags & PH_FLAG_RELU) ? 1 : 0;
              b2.zeros = m->zeros_dev;
              conv_done[op_index] = 1;
              kv[op_index - 1] = PH_KV_F16_BLOCK;
              rc = launch_block2_c32_f16(b2, s);
              break;
            }
          }
          const int cdiv = fmt == FMT_F16 ? 32 : 16;
          ConvF16Args f{};
          f.src0 = slot_ptr(d.src0);
          f.rs0 = s0.cp / rs_div;
          f.chunks0 = s0.cp / cdiv;
          if (d.src1 >= 0) {
            f.src1 = slot_ptr(d.src1);
            f.rs1 = plan.slots[d.src1].cp / rs_div;
            f.chunks1 = plan.slots[d.src1].cp / cdiv;
          }
          f.wpack = op.w_f16_dev[fmt == FMT_F16 ? 1 : 0];
          f.bias = op.b_dev;
          f.dst = slot_ptr(d.dst);
          f.dst_pool = d.dst2 >= 0 ? slot_ptr(d.dst2) : nullptr;
          f.skip_dst = (f.dst_pool && plan.reuse && !plan.unread.empty() && plan.unread[d.dst]) ? 1 : 0;
          f.rs_dst = so.cp / rs_div;
          f.coutp = so.cp;
          f.B = batch;
          f.H = s0.h;
          f.W = s0.w;
          f.relu = (d.flags & PH_FLAG_RELU) ? 1 : 0;
          f.bn = op.bn;
          f.prec = fmt == FMT_F16 ? 1 : 3;
          f.zeros = m->zeros_dev;
          f.clock_probe = m->clock_probe ? m->clock_probe + PH_PROBE_WORDS_PER_OP * (op_index - 1) : nullptr;  // one record block per op
          if (m->head_fuse && fmt == FMT_F16 && !f.dst_pool && op.bn == 64 && (so.cp == 64 || (so.cp == 128 && m->conv_f16_rows))) {
            // plain fp16: a 1x1 head that reads this conv's 64-channel output rides in its epilogue (four MFMAs on the staged fp16 row); when nothing else reads the tensor it never reaches HBM
            // (128 channels: only conv3x3_f16_rows_kernel holds both N tiles of a pixel in one workgroup -- undone below if that kernel does not take the layer)
            for (size_t j = op_index; j < m->ops.size(); ++j) {
              const ph_op_desc& hx = m->ops[j].d;
              if (hx.kind != PH_OP_HEAD || hx.src0 != d.dst) continue;
              if (!(hx.flags & PH_FLAG_SOFTMAX) && out_dev[hx.out_index] && hx.cout <= 32 && fmt_cpad(FMT_F16, hx.cin0) == so.cp) {
                f.head_w = m->ops[j].w_dev;
                f.head_b = m->ops[j].b_dev;
                f.head_dst = out_dev[hx.out_index];
                f.head_cout = hx.cout;
                f.head_wcp = so.cp;
                f.head_sigmoid = (hx.flags & PH_FLAG_SIGMOID) ? 1 : 0;
                head_done[j] = 1;
                if (plan.reuse) {
                  bool other = false;
                  for (size_t k = 0; k < m->ops.size(); ++k)
                    if (k != j && (m->ops[k].d.src0 == d.dst || m->ops[k].d.src1 == d.dst)) other = true;
                  if (!other) f.skip_dst = 1;
                }
              }
              break;
            }
          }
          if (deferred_up >= 0 || (fmt == FMT_F16 && m->conv_f16_rows)) {
            // conv3x3_f16_rows_kernel (row tiles, loader waves, weights L2 -> registers; f16_rows_kernels.hip) where its plan is estimated faster ("conv_f16_rows" 2: wherever
            // the shape fits).  A bilinear x2 in front of this conv that only feeds its second source was not launched (PH_OP_UPSAMPLE): the kernel reads the
            // half-resolution tensor itself; if it does not take the layer, the tensor is produced now.
            int n_cu = 0;
            if (device_cu_count(&n_cu) != PH_OK || n_cu <= 0) n_cu = 256;
            bool on_rows = false;
            if (fmt == FMT_F16 && m->conv_f16_rows) {
              ConvF16Args r = f;
              if (deferred_up >= 0) {
                const ph_op_desc& up = m->ops[deferred_up].d;
                r.src1 = slot_ptr(up.src0);
                r.rs1 = plan.slots[up.src0].cp / rs_div;
                r.src1_lowres = 1;
                r.rows_blend16 = m->upsample_f16math ? 1 : 0;
              }
              const double c_rows = f16_rows_plan(r, n_cu);
              double c_old = f16_conv_cost(f, n_cu);
              if (deferred_up >= 0) c_old += f16_upsample_cost(batch, s0.h / 2, s0.w / 2, plan.slots[d.src1].cp, n_cu);
              if (f.head_w && !(f.bn == 64 && f.coutp == 64))  // a head only the row-tile kernel fuses: the other way costs head1x1_mfma_kernel's launch (its bytes at ~5 TB/s + the boundary)
                c_old += (double)batch * s0.h * s0.w * (f.coutp * 2.0 + f.head_cout * 4.0) / 2500.0 + 3000.0;
              if (c_rows >= 0 && (m->conv_f16_rows >= 2 || c_rows < 0.95 * c_old)) {  // (the estimates are good to a few per cent: a tie stays with the round-2 kernel, measured 6 - 8 % faster on the 2-chunk 192 x 192 layers of cfg5)
                f = r;
                on_rows = true;
              }
            }
            if (deferred_up >= 0 && !on_rows) {
              const ph_op_desc& up = m->ops[deferred_up].d;
              const SlotShape& sl = plan.slots[up.src0];
              rc = launch_upsample_fmt(fmt, slot_ptr(up.src0), slot_ptr(up.dst), batch, sl.h, sl.w, sl.cp, s, (fmt == FMT_F16 && m->upsample_f16math) ? 1 : 0);
              if (rc != PH_OK) return rc;
            }
            deferred_up = -1;
            if (on_rows) {
              kv[op_index - 1] = PH_KV_F16_ROWS;
              rc = launch_conv3x3_f16_rows(f, s);
              break;
            }
            if (f.head_w && (f.bn != 64 || f.coutp != 64)) {  // (a head only the row-tile kernel fuses: leave it to its own launch)
              for (size_t j = op_index; j < m->ops.size(); ++j)
                if (m->ops[j].d.kind == PH_OP_HEAD && m->ops[j].d.src0 == d.dst && head_done[j]) head_done[j] = 0;
              f.head_w = nullptr;
              f.skip_dst = (f.dst_pool && plan.reuse && !plan.unread.empty() && plan.unread[d.dst]) ? 1 : 0;
            }
          }
          kv[op_index - 1] = PH_KV_F16;
          rc = launch_conv3x3_f16(f, s);
          break;
        }
        if (d.ksize != 3) {  // k x k "same" conv: k^2-tap row GEMM
          PH_REQUIRE(s0.c == d.cin0 && (d.src1 < 0 || plan.slots[d.src1].c == d.cin1), "conv channel mismatch");
          GemmArgs g{};
          g.src0 = slot_ptr(d.src0);
          g.c0p = s0.cp;
          g.src1 = d.src1 >= 0 ? slot_ptr(d.src1) : nullptr;
          g.c1p = d.src1 >= 0 ? plan.slots[d.src1].cp : 0;
          g.wpack = op.w_gemm_dev;
          g.bias = op.b_gemm_dev;
          g.dst = slot_ptr(d.dst);
          g.zeros = m->zeros_dev;
          g.coutp = pad16(d.cout);
          g.bn = op.bn_g;
          g.M = batch * s0.h * s0.w;
          g.mode = 5;
          g.ksize = d.ksize;
          g.H = s0.h;
          g.W = s0.w;
          g.act = (d.flags & PH_FLAG_RELU) ? 1 : 0;
          g.late_split = m->gemm_late_split;
          kv[op_index - 1] = PH_KV_ROWGEMM;
          rc = launch_gemm(g, s);
          if (rc == PH_OK && d.dst2 >= 0) rc = launch_pool(slot_ptr(d.dst), slot_ptr(d.dst2), batch, s0.h, s0.w, g.coutp, s);  // the fused-pool flag of the plan, unfused here
          break;
        }
        ConvArgs a{};
        a.src0 = slot_ptr(d.src0);
        a.c0p = s0.cp;
        a.src1 = d.src1 >= 0 ? slot_ptr(d.src1) : nullptr;
        a.c1p = d.src1 >= 0 ? plan.slots[d.src1].cp : 0;
        PH_REQUIRE(s0.c == d.cin0 && (d.src1 < 0 || plan.slots[d.src1].c == d.cin1), "conv channel mismatch");
        a.wpack = op.w_dev;
        a.bias = op.b_dev;
        a.dst = slot_ptr(d.dst);
        a.coutp = pad16(d.cout);
        a.B = batch;
        a.H = s0.h;
        a.W = s0.w;
        a.relu = (d.flags & PH_FLAG_RELU) ? 1 : 0;
        a.bn = op.bn;
        a.clock_probe = m->clock_probe ? m->clock_probe + PH_PROBE_WORDS_PER_OP * (op_index - 1) : nullptr;  // one record block per op (diagnostic builds / kernels that stamp)
        a.dst_pool = d.dst2 >= 0 ? slot_ptr(d.dst2) : nullptr;
        a.skip_dst = (a.dst_pool && plan.reuse && !plan.unread.empty() && plan.unread[d.dst]) ? 1 : 0;  // e.g. cfg3's second encoder block: 1 GiB per 32 frames nobody reads
        a.wpack_dma = op.w_dma_dev;
        a.wpack_wino = op.w_wino_dev;
        a.wpack_wino2 = op.w_wino2_dev;
        a.wpack_w16 = op.w_w16_dev;
        a.w16 = op.w16_dev;
        a.zeros = m->zeros_dev;
        apply_conv_options(m, a);
        a.wpack_sm = op.w_sm_dev;
        a.use_sm = m->conv_smallmap >= 2 ? 2 : ((m->conv_smallmap == 1 && m->conv_splitk == 1 && plan.reuse && m->use_dma) ? 1 : 0);
        a.wpack_wino4 = op.w_wino4_dev;
        a.use_wino4 = m->conv_wino4 == 3 ? 2 : ((m->conv_wino4 == 2 || (m->conv_wino4 == 1 && plan.reuse)) ? 1 : 0);
        // (the two round-4 routings below -- Cout-32 layers on the N-tile-64 kernels, split K -- apply to inference plans only: a training program's forward keeps the kernels its
        // gradient tests were pinned with.  The wave-private kernel's two-source (16 + 32 -> 16) and fused-head forms are NOT gated that way: they are shapes of conv_w16 itself and
        // run in every plan; tests/test_gpu_training.py's default network (filters 8, output stride 2: concat 16 + 32 -> 16) trains through the two-source form, its gradients are
        // compared with autograd's)
        if (op.w_n64_dev && op.w_wino2_dev && op.w_wino_dev && (m->conv_n32_wino2d >= 2 || (m->conv_n32_wino2d == 1 && plan.reuse)) && m->use_dma && m->conv_wino && m->conv_wino2d && m->conv_persist && !w16_fits(a) && wino2d_fits(a)) {
          // Cout 32, K >= 64 (and not a shape of the wave-private kernel): N tile 64 with its upper half empty -- the F(2x2,3x3) kernel skips the missing half's MFMAs
          a.bn = 64;
          a.wpack = op.w_n64_dev;
          a.bias = op.b_n64_dev;
          a.wpack_dma = nullptr;
        }
        if (a.splitk == 1 && !plan.reuse) a.splitk = 0;
        if (plan.tmp_bytes > 0) {
          a.split_scratch = reinterpret_cast<float*>(ws + plan.tmp_offset);
          a.split_scratch_bytes = plan.tmp_bytes;
        }
        if (deferred_up >= 0) {
          // The op before this one is a bilinear x2 that only feeds this conv's second source and was not launched: the F(4x4,3x3) kernel
          // reads the half-resolution tensor itself (the up-sampling rides in its input transform); any other kernel gets the tensor now.
          const ph_op_desc& up = m->ops[deferred_up].d;
          const SlotShape& sl = plan.slots[up.src0];
          ConvArgs f = a;
          f.src1 = slot_ptr(up.src0);
          f.src1_lowres = 1;
          if ((m->use_dma && conv3x3_dma_is_wino4(f)) || conv3x3_takes_sm(f)) {
            a = f;
          } else {
            rc = launch_upsample(slot_ptr(up.src0), slot_ptr(up.dst), batch, sl.h, sl.w, sl.cp, s);
            if (rc != PH_OK) return rc;
          }
          deferred_up = -1;
        }
        const bool on_sm = conv3x3_takes_sm(a);  // small maps at small batches: (8 x 8 pixels, 16 channels) units, one launch, no split K (conv3x3_sm_kernel)
        const double fill = (double)s0.h * s0.w / ((double)((s0.h + 15) / 16 * 16) * ((s0.w + 31) / 32 * 32));
        // the halo kernel pads Cout to a multiple of its N tile (64): e.g. Cout = 96 does 33 % extra MFMA work there,
        // none in the row GEMM (N tiles of 96 / 128)
        const double n_fill = (double)a.coutp / ((a.coutp + op.bn - 1) / op.bn * op.bn);
        const int bn_g = op.bn_g > 0 ? op.bn_g : 128;
        const double n_fill_g = (double)a.coutp / ((a.coutp + bn_g - 1) / bn_g * bn_g);
        // the halo kernel runs Winograd F(2,3) (2/3 of the MFMA work) where its weights exist: it wins down to ~0.5 tile fill (measured on the ConvNeXt decoder: 48x48 and 24x24 maps)
        double halo_gain = (op.w_wino_dev && m->use_dma) ? m->gemm_fill_threshold / m->gemm_fill_threshold_wino : 1.0;
        double halo_fill = fill;
        if (m->use_dma && m->conv_wino && m->conv_wino2d && m->conv_persist && op.w_wino2_dev && a.bn == 64 && a.c0p + a.c1p >= 32 && wino2d_fits(a)) {
          // the F(2x2,3x3) kernel: 16x16-pixel tiles and 4/9 of the MFMA work -- it beats the 9-tap row GEMM down to ~0.4 tile fill
          halo_fill = (double)s0.h * s0.w / ((double)((s0.h + 15) / 16 * 16) * ((s0.w + 15) / 16 * 16));
          halo_gain = m->gemm_fill_threshold / m->gemm_fill_threshold_wino2d;
        }
        if (!on_sm && m->use_dma && d.dst2 < 0 && op.w_gemm_dev && !a.src1_lowres && (halo_fill * halo_gain < m->gemm_fill_threshold || n_fill * halo_fill * halo_gain < 0.8 * n_fill_g)) {
          // small feature map (the 16x32-pixel tiles of the halo kernel would be mostly padding) or a Cout that
          // fits the halo kernel's N tile badly -> 9-tap row GEMM
          GemmArgs g{};
          g.src0 = a.src0;
          g.src1 = a.src1;
          g.c0p = a.c0p;
          g.c1p = a.c1p;
          g.wpack = op.w_gemm_dev;
          g.bias = op.b_gemm_dev;
          g.dst = a.dst;
          g.zeros = m->zeros_dev;
          g.coutp = a.coutp;
          g.bn = op.bn_g;
          g.M = batch * s0.h * s0.w;
          g.mode = 2;
          g.H = s0.h;
          g.W = s0.w;
          g.act = a.relu ? 1 : 0;
          g.late_split = m->gemm_late_split;
          g.persist2 = m->gemm_persist2;
          kv[op_index - 1] = PH_KV_ROWGEMM;
          rc = launch_gemm(g, s);
          break;
        }
        if (m->head_fuse && !a.dst_pool && ((a.coutp == 64 && a.bn == 64) || ((a.coutp == 16 || a.coutp == 32) && a.bn == 32)) && m->use_dma) {
          // A 1x1 head that reads this conv's 64-channel output rides in the F(2x2,3x3) kernel's epilogue (the accumulator layout is the
          // MFMA's B operand: conv3x3_wino2d_kernel); when nothing else reads the tensor (inference plans) it never reaches HBM.
          for (size_t j = op_index; j < m->ops.size(); ++j) {
            const ph_op_desc& hx = m->ops[j].d;
            if (hx.kind != PH_OP_HEAD || hx.src0 != d.dst) continue;
            ConvArgs no4 = a;  // (a fused head beats F(4x4,3x3) + a head launch on the 64-channel layers both could take: ask what runs without that kernel; head_w then keeps it off)
            no4.use_wino4 = 0;
            const bool on_w2d = a.coutp == 64 && hx.cout <= 32 && pad16(hx.cin0) == 64 && conv3x3_dma_is_wino2d(no4) && wino2d_ksplit(no4) <= 1;
            const bool on_w16 = a.coutp <= 32 && hx.cout <= 16 && pad16(hx.cin0) == a.coutp && conv3x3_dma_is_w16_head(a);  // (conv3x3_w16_kernel<.., HEAD>)
            const bool sm_head = on_sm && a.coutp <= 32 && pad16(hx.cin0) == a.coutp;  // (conv3x3_sm_kernel: the workgroup holds every channel of its pixels)
            if (!(hx.flags & PH_FLAG_SOFTMAX) && out_dev[hx.out_index] && (sm_head || (!on_sm && (on_w2d || on_w16)))) {
              a.head_w = m->ops[j].w_dev;
              a.head_b = m->ops[j].b_dev;
              a.head_dst = out_dev[hx.out_index];
              a.head_cout = hx.cout;
              a.head_wcp = a.coutp;
              a.head_sigmoid = (hx.flags & PH_FLAG_SIGMOID) ? 1 : 0;
              head_done[j] = 1;
              if (plan.reuse) {  // training keeps every activation
                bool other = false;
                for (size_t k = 0; k < m->ops.size(); ++k)
                  if (k != j && (m->ops[k].d.src0 == d.dst || m->ops[k].d.src1 == d.dst)) other = true;
                if (!other) a.skip_dst = 1;
              }
            }
            break;
          }
        }
        if (m->pool_peephole && !a.dst_pool && !a.head_w && a.relu && m->use_dma && (a.bn == 64 || m->dma32) && op_index < m->ops.size()) {
          // An UNFUSED program (the training forward: the backward walks one op per activation) still gets the conv kernels' fused
          // 2x2 max pool: when the next op is the pool of this conv's output, the epilogue writes both tensors and the pool op is skipped.
          const ph_op_desc& nx = m->ops[op_index].d;
          if (nx.kind == PH_OP_POOL && nx.src0 == d.dst) {
            a.dst_pool = slot_ptr(nx.dst);
            if (conv3x3_dma_is_f2x2(a))  // the F(2x2,3x3) kernels (a Winograd tile is a pool window); other kernels keep the separate pool
              pooled_by_conv = d.dst;
            else
              a.dst_pool = nullptr;
          }
        }
        if (on_sm) {
          kv[op_index - 1] = PH_KV_SMALLMAP;
          rc = launch_conv3x3_sm(a, s);
          break;
        }
        kv[op_index - 1] = (m->use_dma && (a.bn == 64 || m->dma32)) ? conv3x3_dma_variant(a) : PH_KV_DIRECT;
        rc = (m->use_dma && (a.bn == 64 || m->dma32)) ? launch_conv3x3_dma(a, s) : launch_conv3x3(a, s);
        break;
      }
      case PH_OP_POOL: {
        if (pooled_by_conv == d.src0) {  // written by the producing conv's epilogue
          pooled_by_conv = -1;
          break;
        }
        const SlotShape& s0 = plan.slots[d.src0];
        rc = fmt == FMT_F32 ? launch_pool(slot_ptr(d.src0), slot_ptr(d.dst), batch, s0.h, s0.w, s0.cp, s)
                            : launch_pool_fmt(fmt, slot_ptr(d.src0), slot_ptr(d.dst), batch, s0.h, s0.w, s0.cp, s);
        break;
      }
      case PH_OP_UPSAMPLE: {
        const SlotShape& s0 = plan.slots[d.src0];
        if (fmt == FMT_F32 && plan.reuse && m->upsample_fold && (m->conv_wino4 || m->conv_smallmap) && m->use_dma && op_index < m->ops.size()) {
          // inference plans: when the NEXT op is a 3x3 conv that takes this tensor as its second source and nothing else reads it, the conv
          // decides (it may run the F(4x4,3x3) kernel, which up-samples in its input transform) -- see PH_OP_CONV
          const PackedOp& nxo = m->ops[op_index];
          const ph_op_desc& nx = nxo.d;
          bool only = nx.kind == PH_OP_CONV && nx.ksize == 3 && nx.src1 == d.dst && nx.src0 != d.dst && nx.dst2 < 0 && ((m->conv_wino4 && nxo.w_wino4_dev != nullptr) || (m->conv_smallmap && nxo.w_sm_dev != nullptr));  // (a form that is switched off does not defer the launch)
          for (size_t k = 0; only && k < m->ops.size(); ++k)
            if (k != op_index && (m->ops[k].d.src0 == d.dst || m->ops[k].d.src1 == d.dst)) only = false;
          if (only) {
            deferred_up = (int)op_index - 1;
            break;
          }
        }
        if (fmt == FMT_F16 && plan.reuse && m->upsample_fold && m->conv_f16_rows && op_index < m->ops.size()) {
          // the same on the fp16 pipe: conv3x3_f16_rows_kernel's loader waves blend the half-resolution tensor into the halo
          const PackedOp& nxo = m->ops[op_index];
          const ph_op_desc& nx = nxo.d;
          bool only = nx.kind == PH_OP_CONV && nx.ksize == 3 && nx.src1 == d.dst && nx.src0 != d.dst && nxo.bn == 64 && plan.slots[nx.src0].h == 2 * s0.h && plan.slots[nx.src0].w == 2 * s0.w;
          for (size_t k = 0; only && k < m->ops.size(); ++k)
            if (k != op_index && (m->ops[k].d.src0 == d.dst || m->ops[k].d.src1 == d.dst)) only = false;
          if (only) {
            deferred_up = (int)op_index - 1;
            break;
          }
        }
        rc = fmt == FMT_F32 ? launch_upsample(slot_ptr(d.src0), slot_ptr(d.dst), batch, s0.h, s0.w, s0.cp, s)
                            : launch_upsample_fmt(fmt, slot_ptr(d.src0), slot_ptr(d.dst), batch, s0.h, s0.w, s0.cp, s, (fmt == FMT_F16 && m->upsample_f16math) ? 1 : 0);
        break;
      }
      case PH_OP_CONVT: {
        const SlotShape& s0 = plan.slots[d.src0];
        PH_REQUIRE(s0.c == d.cin0, "convT channel mismatch");
        if (fmt == FMT_F32 && m->convt_phase && op.wt_phase_dev[0]) {  // four output-phase GEMMs: 9 taps per INPUT pixel, no zero-stuffed tensor
          const SlotShape& so = plan.slots[d.dst];
          for (int ph = 0; ph < (m->convt_one_launch ? 1 : 4) && rc == PH_OK; ++ph) {  // one launch for the four phases (grid.y), or four launches
            GemmArgs g{};
            g.src0 = slot_ptr(d.src0);
            g.c0p = s0.cp;
            g.wpack = op.wt_phase_dev[ph];
            g.bias = op.bt_dev;
            g.dst = slot_ptr(d.dst);
            g.zeros = m->zeros_dev;
            g.coutp = so.cp;
            g.bn = op.bn_t;
            g.M = batch * s0.h * s0.w;
            g.mode = 3;
            g.ntaps = (1 + (ph >> 1)) * (1 + (ph & 1));
            g.H = s0.h;
            g.W = s0.w;
            g.out_patch = 1;
            g.out_tap = ph;
            if (m->convt_one_launch) {
              g.all_phases = 1;
              for (int q = 0; q < 4; ++q) g.wpack_ph[q] = op.wt_phase_dev[q];
            }
            g.out_H = 2 * s0.h;
            g.out_W = 2 * s0.w;
            g.act = (d.flags & PH_FLAG_SILU) ? 4 : ((d.flags & PH_FLAG_RELU) ? 1 : 0);
            g.scale = op.wt_scale_dev;
            g.shift = op.wt_shift_dev;
            g.affine_first = op.wt_scale_dev ? 1 : 0;
            g.late_split = m->gemm_late_split;
            rc = launch_gemm(g, s);
          }
          kv[op_index - 1] = PH_KV_ROWGEMM;
          break;
        }
        PH_REQUIRE(!op.wt_scale_dev && !(d.flags & PH_FLAG_SILU), "folded BatchNorm / SiLU on a transposed conv need the phase GEMMs (exact precision, convt_phase = 1)");
        float* tmp = reinterpret_cast<float*>(ws + plan.tmp_offset);
        rc = launch_zero_stuff(slot_ptr(d.src0), tmp, batch, s0.h, s0.w, s0.cp / rs_div, s);  // 16-B quads: format-agnostic
        if (rc != PH_OK) break;
        if (fmt != FMT_F32) {
          const SlotShape& so = plan.slots[d.dst];
          ConvF16Args f{};
          f.src0 = tmp;
          f.rs0 = s0.cp / rs_div;
          f.chunks0 = s0.cp / (fmt == FMT_F16 ? 32 : 16);
          f.wpack = op.w_f16_dev[fmt == FMT_F16 ? 1 : 0];
          f.bias = op.b_dev;
          f.dst = slot_ptr(d.dst);
          f.rs_dst = so.cp / rs_div;
          f.coutp = so.cp;
          f.B = batch;
          f.H = 2 * s0.h;
          f.W = 2 * s0.w;
          f.relu = (d.flags & PH_FLAG_RELU) ? 1 : 0;
          f.bn = op.bn;
          f.prec = fmt == FMT_F16 ? 1 : 3;
          f.zeros = m->zeros_dev;
          f.clock_probe = m->clock_probe ? m->clock_probe + PH_PROBE_WORDS_PER_OP * (op_index - 1) : nullptr;  // one record block per op
          if (m->head_fuse && fmt == FMT_F16 && !f.dst_pool && op.bn == 64 && so.cp == 64) {
            // plain fp16: a 1x1 head that reads this conv's 64-channel output rides in its epilogue (four MFMAs on the staged fp16 row); when nothing else reads the tensor it never reaches HBM
            for (size_t j = op_index; j < m->ops.size(); ++j) {
              const ph_op_desc& hx = m->ops[j].d;
              if (hx.kind != PH_OP_HEAD || hx.src0 != d.dst) continue;
              if (!(hx.flags & PH_FLAG_SOFTMAX) && out_dev[hx.out_index] && hx.cout <= 32 && pad16(hx.cin0) == 64) {
                f.head_w = m->ops[j].w_dev;
                f.head_b = m->ops[j].b_dev;
                f.head_dst = out_dev[hx.out_index];
                f.head_cout = hx.cout;
                f.head_wcp = 64;
                f.head_sigmoid = (hx.flags & PH_FLAG_SIGMOID) ? 1 : 0;
                head_done[j] = 1;
                if (plan.reuse) {
                  bool other = false;
                  for (size_t k = 0; k < m->ops.size(); ++k)
                    if (k != j && (m->ops[k].d.src0 == d.dst || m->ops[k].d.src1 == d.dst)) other = true;
                  if (!other) f.skip_dst = 1;
                }
              }
              break;
            }
          }
          kv[op_index - 1] = PH_KV_F16;
          rc = launch_conv3x3_f16(f, s);
          break;
        }
        ConvArgs a{};
        a.src0 = tmp;
        a.c0p = s0.cp;
        a.src1 = nullptr;
        a.c1p = 0;
        a.wpack = op.w_dev;
        a.bias = op.b_dev;
        a.dst = slot_ptr(d.dst);
        a.coutp = pad16(d.cout);
        a.B = batch;
        a.H = 2 * s0.h;
        a.W = 2 * s0.w;
        a.relu = (d.flags & PH_FLAG_RELU) ? 1 : 0;
        a.bn = op.bn;
        a.clock_probe = nullptr;
        a.dst_pool = nullptr;
        a.wpack_dma = op.w_dma_dev;
        a.zeros = m->zeros_dev;
        apply_conv_options(m, a);
        kv[op_index - 1] = (m->use_dma && (a.bn == 64 || m->dma32)) ? conv3x3_dma_variant(a) : PH_KV_DIRECT;
        rc = (m->use_dma && (a.bn == 64 || m->dma32)) ? launch_conv3x3_dma(a, s) : launch_conv3x3(a, s);
        break;
      }
      case PH_OP_PATCH_STEM: {
        PH_REQUIRE(d.cin0 == in_channels, "input has %d channels, network expects %d", in_channels, d.cin0);
        const SlotShape& so = plan.slots[d.dst];
        PatchStemArgs a{};
        a.src = input_dev;
        a.w = op.w_dev;
        a.bias = op.b_dev;
        a.dst = slot_ptr(d.dst);
        a.dtype = in_dtype;
        a.cin = d.cin0;
        a.coutp = so.cp;
        a.B = batch;
        a.H = height;
        a.W = width;
        a.OH = so.h;
        a.OW = so.w;
        a.k = d.ksize;
        a.stride = d.cmid;
        if (m->dw_ln_fuse && plan.reuse && fmt == FMT_F32 && op_index < m->ops.size() && a.cin == 1 && a.coutp <= 128 && a.k >= 2 && a.k <= 4 && a.stride >= 1 && a.stride <= 4) {
          // the stem's LayerNorm2d rides in the patch kernel (inference plans: nothing else reads the un-normalised tensor)
          const ph_op_desc& nx = m->ops[op_index].d;
          if (nx.kind == PH_OP_LAYERNORM && nx.src0 == d.dst && nx.cin0 == d.cout) {
            bool other = false;
            for (size_t k = 0; k < m->ops.size(); ++k)
              if (k != op_index && (m->ops[k].d.src0 == d.dst || m->ops[k].d.src1 == d.dst)) other = true;
            if (!other) {
              a.ln_gamma = m->ops[op_index].w_dev;
              a.ln_beta = m->ops[op_index].b_dev;
              a.ln_c = d.cout;
              a.dst = slot_ptr(nx.dst);
              fused_ln = (int)op_index;
            }
          }
        }
        rc = launch_patch_stem(a, s);
        break;
      }
      case PH_OP_DWCONV: {
        const SlotShape& s0 = plan.slots[d.src0];
        PH_REQUIRE(s0.c == d.cin0, "depthwise conv channel mismatch");
        DwConvArgs a{};
        a.src = slot_ptr(d.src0);
        a.w = op.w_dev;
        a.bias = op.b_dev;
        a.dst = slot_ptr(d.dst);
        a.cp = s0.cp;
        a.B = batch;
        a.H = s0.h;
        a.W = s0.w;
        if (m->dw_ln_fuse && plan.reuse && fmt == FMT_F32 && op_index < m->ops.size()) {
          // CNBlock: the LayerNorm that follows rides in the depthwise kernel (inference plans: nothing else reads the depthwise output)
          const ph_op_desc& nx = m->ops[op_index].d;
          if (nx.kind == PH_OP_LAYERNORM && nx.src0 == d.dst && nx.cin0 == d.cout) {
            bool other = false;
            for (size_t k = 0; k < m->ops.size(); ++k)
              if (k != op_index && (m->ops[k].d.src0 == d.dst || m->ops[k].d.src1 == d.dst)) other = true;
            if (!other) {
              a.ln_gamma = m->ops[op_index].w_dev;
              a.ln_beta = m->ops[op_index].b_dev;
              a.ln_c = d.cout;
              a.dst = slot_ptr(nx.dst);
              fused_ln = (int)op_index;
            }
          }
        }
        rc = launch_dwconv7(a, s);
        break;
      }
      case PH_OP_LAYERNORM: {
        if (fused_ln == (int)op_index - 1) {  // applied by the depthwise conv before it
          fused_ln = -1;
          break;
        }
        const SlotShape& s0 = plan.slots[d.src0];
        PH_REQUIRE(s0.c == d.cin0, "LayerNorm channel mismatch");
        rc = launch_layernorm(slot_ptr(d.src0), op.w_dev, op.b_dev, slot_ptr(d.dst), s0.c, s0.cp, (size_t)batch * s0.h * s0.w, s);
        break;
      }
      case PH_OP_LINEAR:
      case PH_OP_PATCH_CONV: {
        if (skip_next_linear) {  // the second Linear of a CNBlock MLP that cnblock_mlp_kernel computed with the first
          skip_next_linear = false;
          kv[op_index - 1] = PH_KV_MLP;
          break;
        }
        const SlotShape& s0 = plan.slots[d.src0];
        const SlotShape& so = plan.slots[d.dst];
        PH_REQUIRE(s0.c == d.cin0, "GEMM channel mismatch");
        // CNBlock's MLP in one launch (inference plans that recycle slots: the 4C-wide hidden tensor is nobody else's business): Linear + GELU whose only reader is a
        // Linear with layer scale + residual, both at a width cnblock_mlp_kernel takes
        if (m->mlp_fuse && d.kind == PH_OP_LINEAR && (d.flags & PH_FLAG_GELU) && op.w_mlp_dev && plan.reuse && op_index < m->ops.size()) {
          const PackedOp& nxo = m->ops[op_index];
          const ph_op_desc& nx = nxo.d;
          bool pair = nx.kind == PH_OP_LINEAR && (nx.flags & PH_FLAG_SCALE_RESIDUAL) && nxo.w_mlp_dev && nx.src0 == d.dst && nx.src1 >= 0 && nx.src1 != d.dst && nx.cin0 == d.cout && nx.cout == d.cin0 &&
                      s0.cp == d.cin0 && plan.slots[nx.dst].cp == nx.cout && plan.slots[nx.src1].cp == nx.cout;
          for (size_t k = 0; pair && k < m->ops.size(); ++k)
            if (k != op_index && (m->ops[k].d.src0 == d.dst || m->ops[k].d.src1 == d.dst)) pair = false;
          if (pair) {
            MlpArgs f{};
            f.x = slot_ptr(d.src0);
            f.w1img = op.w_mlp_dev;
            f.w2img = nxo.w_mlp_dev;
            f.b1 = op.b_dev;
            f.b2 = nxo.b_dev;
            f.scale = nxo.w2_dev;
            f.residual = slot_ptr(nx.src1);
            f.dst = slot_ptr(nx.dst);
            f.M = batch * so.h * so.w;
            f.C = d.cin0;
            kv[op_index - 1] = PH_KV_MLP;
            skip_next_linear = true;
            rc = launch_cnblock_mlp(f, s);
            break;
          }
        }
        GemmArgs a{};
        a.src0 = slot_ptr(d.src0);
        a.wpack = op.w_dma_dev;
        a.bias = op.b_dev;
        a.dst = slot_ptr(d.dst);
        a.zeros = m->zeros_dev;
        a.c0p = s0.cp;
        a.coutp = so.cp;
        a.bn = op.bn;
        a.M = batch * so.h * so.w;
        a.mode = d.kind == PH_OP_PATCH_CONV ? 1 : 0;
        a.H = s0.h;
        a.W = s0.w;
        a.act = (d.flags & PH_FLAG_GELU) ? 2 : ((d.flags & PH_FLAG_RELU) ? 1 : 0);
        if (d.flags & PH_FLAG_SCALE_RESIDUAL) {
          a.scale = op.w2_dev;
          a.residual = slot_ptr(d.src1);
        }
        a.late_split = m->gemm_late_split;
        // training program (Linear and GELU as separate ops, the pre-activation is kept for the backward pass): the
        // GEMM epilogue writes both the pre-activation and GELU(pre-activation), the GELU op after it becomes a no-op
        if (m->fuse_gelu_fwd && a.mode == 0 && a.act == 0 && !a.residual && op_index < m->ops.size()) {
          const ph_op_desc& nx = m->ops[op_index].d;
          if (nx.kind == PH_OP_GELU && nx.src0 == d.dst && nx.dst != d.dst) {
            a.act = 2;
            a.dst_pre = a.dst;
            a.dst = slot_ptr(nx.dst);
            skip_next_gelu = true;
          }
        }
        // ... and (CNBlock's second Linear -> layer scale + residual, separate ops in a training program because the backward needs the un-scaled
        // output): the epilogue writes the un-scaled output AND scale * output + residual, the scale-add op after it becomes a no-op
        if (m->fuse_gelu_fwd && a.mode == 0 && a.act == 0 && !a.residual && !a.dst_pre && op_index < m->ops.size() && so.cp % a.bn == 0) {
          const ph_op_desc& nx = m->ops[op_index].d;
          if (nx.kind == PH_OP_SCALE_ADD && nx.src0 == d.dst && nx.dst != d.dst && nx.src1 != d.dst && nx.src1 >= 0 && nx.cin0 == d.cout) {
            a.scale = m->ops[op_index].w_dev;
            a.residual = slot_ptr(nx.src1);
            a.dst_pre = a.dst;
            a.dst = slot_ptr(nx.dst);
            skip_next_scale_add = true;
          }
        }
        kv[op_index - 1] = PH_KV_ROWGEMM;
        rc = launch_gemm(a, s);
        break;
      }
      case PH_OP_GLOBAL_MAXPOOL: {
        const SlotShape& s0 = plan.slots[d.src0];
        PH_REQUIRE(s0.c == d.cin0, "global pool channel mismatch");
        rc = launch_global_maxpool(slot_ptr(d.src0), slot_ptr(d.dst), batch, s0.h * s0.w, s0.cp, s);
        break;
      }
      case PH_OP_GELU: {
        if (skip_next_gelu) {  // already produced by the preceding Linear's epilogue
          skip_next_gelu = false;
          break;
        }
        const SlotShape& s0 = plan.slots[d.src0];
        rc = launch_gelu_fwd(slot_ptr(d.src0), slot_ptr(d.dst), (size_t)batch * s0.h * s0.w * s0.cp, s);
        break;
      }
      case PH_OP_SCALE_ADD: {
        if (skip_next_scale_add) {  // already produced by the preceding Linear's epilogue
          skip_next_scale_add = false;
          break;
        }
        const SlotShape& s0 = plan.slots[d.src0];
        PH_REQUIRE(s0.c == d.cin0, "scale-add channel mismatch");
        rc = launch_scale_add_fwd(slot_ptr(d.src0), slot_ptr(d.src1), op.w_dev, slot_ptr(d.dst), s0.cp, (size_t)batch * s0.h * s0.w * s0.cp, s);
        break;
      }
      case PH_OP_HEAD: {
        if (head_done[op_index - 1]) {  // computed by its producer's epilogue
          kv[op_index - 1] = PH_KV_FUSED;
          break;
        }
        const SlotShape& s0 = plan.slots[d.src0];
        PH_REQUIRE(s0.c == d.cin0, "head channel mismatch");
        PH_REQUIRE(out_dev[d.out_index] != nullptr, "output %d is null", d.out_index);
        rc = launch_head_fmt(fmt, slot_ptr(d.src0), op.w_dev, op.b_dev, out_dev[d.out_index], batch, s0.h * s0.w, s0.cp, pad16(d.cin0), d.cout,
                             (d.flags & PH_FLAG_SIGMOID) ? 1 : 0, s);
        if (rc == PH_OK && (d.flags & PH_FLAG_SOFTMAX)) {
          PH_REQUIRE(s0.h == 1 && s0.w == 1, "softmax head expects a pooled (1x1) feature");
          rc = launch_softmax_rows(out_dev[d.out_index], batch, d.cout, s);
        }
        break;
      }
      default:
        set_error("unknown op kind %d", d.kind);
        rc = PH_E_INVALID;
    }
    if (rc != PH_OK) return rc;
  }
  if (m->profiling) {
    PH_HIP_CHECK(hipEventRecord(m->ev[m->ops.size()], s));
    m->events_pending = true;
  }
  m->last_plan = plan;
  m->last_ws = ws;
  m->last_batch = batch;
  return PH_OK;
}

int32_t ph_op_desc_size(void) { return (int32_t)sizeof(ph_op_desc); }

int ph_debug_split_plan(int32_t B, int32_t H, int32_t W, int32_t cin_padded, int32_t cout_padded, int32_t splitk, int32_t n_cu, int64_t* out4) {
  PH_REQUIRE(out4 && B > 0 && H > 0 && W > 0 && cin_padded > 0 && cout_padded > 0 && n_cu > 0 && (cin_padded % 16) == 0 && (cout_padded % 16) == 0, "ph_debug_split_plan: bad arguments");
  out4[0] = wino2d_ksplit_shape(B, H, W, cin_padded, cout_padded, splitk, n_cu);
  out4[1] = ((H & 3) || (W & 3)) ? 1 : wino4_ksplit_shape(B, H, W, cin_padded, cout_padded, splitk, n_cu);
  out4[2] = wino2d_split_scratch_bytes(B, H, W, cin_padded, cout_padded, splitk, n_cu) / 1024;
  out4[3] = wino4_split_scratch_bytes(B, H, W, cin_padded, cout_padded, splitk, n_cu) / 1024;
  return PH_OK;
}

namespace {
struct OptionRef {
  const char* key;
  int* i;
  double* d;
};
// every tunable of a handle; defaults are the measured-best variants (DESIGN.md, appendix "handle options")
std::vector<OptionRef> option_table(ph_model* m) {
  return {
      {"conv_wino", &m->conv_wino, nullptr},            // 1 Winograd F(2,3) 3x3 kernels, 2 only N-tile-64 layers, 0 direct 9-tap kernels
      {"conv_w16", &m->conv_w16, nullptr},              // 1: Cout-32 / Cin-16-or-32 3x3 convs on the wave-private F(2x2,3x3) kernel; 0: F(2,3) along x
      {"conv_wino2d", &m->conv_wino2d, nullptr},        // 1: N-tile-64 3x3 convs on the F(2x2,3x3) kernel; 0: F(2,3) along x only
      {"conv_wino4", &m->conv_wino4, nullptr},          // K-heavy 3x3 convs on the F(4x4,3x3) kernel: 1 inference plans, 2 every plan (both: where estimated faster), 3 every plan wherever it fits, 0 never
      {"conv_wino4_min_cin", &m->conv_wino4_min_cin, nullptr},  // padded input channels from which a layer takes it
      {"conv_n32_wino2d", &m->conv_n32_wino2d, nullptr},  // Cout-32 / K >= 64 layers on the F(2x2,3x3) kernel with a half-empty N tile (0: the N-tile-32 F(2,3) kernel)
      {"conv_splitk_finish", &m->conv_splitk_finish, nullptr},  // 0: the split-K second stage as a launch of its own (A/B, tests)
      {"mlp_fuse", &m->mlp_fuse, nullptr},                // inference plans: CNBlock's two Linears (+ GELU, layer scale, residual) in one launch (cnblock_mlp_kernel)
      {"block_fuse", &m->block_fuse, nullptr},          // plain fp16, inference plans: the two convs of a 32-channel encoder block in one launch (block2_c32_f16_kernel)
      {"stem_f16mfma", &m->stem_f16mfma, nullptr},      // plain fp16: the fused first block with BOTH convs on the fp16 matrix pipe (stem_f16_kernel) instead of stem_fused_kernel<CIN, 3>
      {"upsample_f16math", &m->upsample_f16math, nullptr},  // a bilinear x2 folded into conv3x3_f16_rows_kernel blends in packed fp16 arithmetic (1) or in fp32 as upsample2x_fmt_kernel does (0)
      {"conv_f16_rows", &m->conv_f16_rows, nullptr},    // plain-fp16 precision: conv3x3_f16_rows_kernel 0 never, 1 where its plan is estimated faster than conv3x3_f16_persist_kernel, 2 wherever the shape fits
      {"conv_smallmap", &m->conv_smallmap, nullptr},    // conv3x3_sm_kernel for small maps at small batches: 0 never, 1 where estimated faster (inference plans, conv_splitk = 1), 2 wherever the shape fits
      {"conv_splitk", &m->conv_splitk, nullptr},        // split K on the F(2x2,3x3) kernel for layers with fewer work units than CUs: 0 never, 1 where estimated faster, n >= 2 force n slices
      {"upsample_fold", &m->upsample_fold, nullptr},    // bilinear x2 folded into the F(4x4,3x3) input transform of the conv that consumes it
      {"stem_wino", &m->stem_wino, nullptr},            // second conv of the fused stem in Winograd form
      {"dgrad_wino", &m->dgrad_wino, nullptr},          // 0: direct 9-tap kernels for the backward's data-gradient convs
      {"conv_dma", &m->use_dma, nullptr},               // 0: register-staged 3x3 kernel instead of the LDS-DMA family
      {"conv_dma32", &m->dma32, nullptr},               // 0: Cout <= 48 layers on the register-staged kernel
      {"conv_persist", &m->conv_persist, nullptr},      // 0: one pixel tile per workgroup
      {"conv_c16", &m->conv_c16, nullptr},              // 0: 16 -> 16 channel layers on the generic kernel
      {"conv_dma_stagger", &m->dma_stagger, nullptr},   // 0: SIMD-partner waves issue DMA pieces at the same step
      {"fuse_gelu_fwd", &m->fuse_gelu_fwd, nullptr},
      {"fuse_gelu_bwd", &m->fuse_gelu_bwd, nullptr},
      {"dw_ln_fuse", &m->dw_ln_fuse, nullptr},          // ConvNeXt inference: LayerNorm fused into the depthwise 7x7 kernel
      {"head_fuse", &m->head_fuse, nullptr},            // a 1x1 head on a 64-channel conv output is computed in that conv's F(2x2,3x3) epilogue
      {"pool_peephole", &m->pool_peephole, nullptr},    // unfused programs: a conv whose next op pools its output writes the pooled tensor from its epilogue
      {"mask_fold", &m->mask_fold, nullptr},            // ReLU mask applied by the pool backward that completes a conv output's gradient
      {"wgrad_wino", &m->wgrad_wino, nullptr},          // 3x3 weight gradients: 1 Winograd F(2x2,3x3) domain, 0 direct nine-tap kernel
      {"wgrad_rows", &m->wgrad_rows, nullptr},          // 0 32x32-tile wgrad kernel, 1 auto, 2 nine row-wgrad GEMMs
      {"workspace_reuse", &m->workspace_reuse, nullptr},  // 1: activation slots share memory once their last reader has run (inference programs only)
      {"convt_one_launch", &m->convt_one_launch, nullptr},  // 1: the four output-phase GEMMs of a transposed conv in one launch (grid.y = phase); 0: four launches
      {"convt_phase", &m->convt_phase, nullptr},        // 0: transposed convs by zero-stuffing + 3x3 conv (4x the FLOPs; A/B reference)
      {"conv_precision", &m->conv_precision, nullptr},  // 0 exact fp32 MFMA, 1 split-fp16 MFMA (22-bit products), 2 plain fp16 (autocast-equivalent)
      {"gemm_late_split", &m->gemm_late_split, nullptr},
      {"gemm_persist2", &m->gemm_persist2, nullptr},
      {"conv_gemm_fill", nullptr, &m->gemm_fill_threshold},            // tile fill below which a 3x3 conv runs as a row GEMM (0 disables)
      {"conv_gemm_fill_wino", nullptr, &m->gemm_fill_threshold_wino},  // ... when the halo kernel is the Winograd one
      {"conv_gemm_fill_wino2d", nullptr, &m->gemm_fill_threshold_wino2d},  // ... when it is the F(2x2,3x3) kernel (16x16-pixel tiles)
  };
}
}  // namespace

int ph_model_set_option(ph_model* m, const char* key, double value) {
  PH_REQUIRE(m && key, "ph_model_set_option: null argument");
  for (const OptionRef& o : option_table(m))
    if (!strcmp(o.key, key)) {
      if (o.i) *o.i = (int)value;
      if (o.d) *o.d = (!strcmp(key, "conv_gemm_fill_wino") || !strcmp(key, "conv_gemm_fill_wino2d")) ? std::max(value, 1e-3) : value;
      return PH_OK;
    }
  set_error("ph_model_set_option: unknown option '%s'", key);
  return PH_E_INVALID;
}

int ph_model_get_option(const ph_model* m, const char* key, double* value) {
  PH_REQUIRE(m && key && value, "ph_model_get_option: null argument");
  for (const OptionRef& o : option_table(const_cast<ph_model*>(m)))
    if (!strcmp(o.key, key)) {
      *value = o.i ? (double)*o.i : *o.d;
      return PH_OK;
    }
  set_error("ph_model_get_option: unknown option '%s'", key);
  return PH_E_INVALID;
}

int ph_model_set_profiling(ph_model* m, int32_t enabled) {
  PH_REQUIRE(m, "ph_model_set_profiling: null model");
  if (enabled && m->ev.empty()) {
    m->ev.resize(m->ops.size() + 1);
    for (auto& e : m->ev) PH_HIP_CHECK(hipEventCreate(&e));
  }
  if (enabled == 1 || m->op_ms.size() != m->ops.size()) {
    m->op_ms.assign(m->ops.size(), 0.0);
    m->profiled_forwards = 0;
    m->events_pending = false;
  }
  m->profiling = enabled != 0;
  return PH_OK;
}

int ph_model_profile_read(ph_model* m, double* op_ms, int32_t n_ops, int32_t* n_forwards) {
  PH_REQUIRE(m && op_ms && n_forwards, "ph_model_profile_read: null argument");
  PH_REQUIRE(n_ops == (int32_t)m->ops.size(), "ph_model_profile_read: model has %d ops", (int)m->ops.size());
  int rc = drain_events(m);
  if (rc != PH_OK) return rc;
  for (int i = 0; i < n_ops; ++i) op_ms[i] = i < (int)m->op_ms.size() ? m->op_ms[i] : 0.0;
  *n_forwards = m->profiled_forwards;
  return PH_OK;
}

int ph_model_last_kernels(const ph_model* m, int32_t* codes, int32_t n_ops) {
  PH_REQUIRE(m && codes, "ph_model_last_kernels: null argument");
  PH_REQUIRE(n_ops == (int32_t)m->ops.size(), "ph_model_last_kernels: model has %d ops", (int)m->ops.size());
  for (int i = 0; i < n_ops; ++i) codes[i] = i < (int)m->last_variant.size() ? m->last_variant[i] : PH_KV_NONE;
  return PH_OK;
}

int ph_model_set_clock_probe(ph_model* m, void* buf_dev) {
  PH_REQUIRE(m, "ph_model_set_clock_probe: null model");
  m->clock_probe = static_cast<unsigned long long*>(buf_dev);
  return PH_OK;
}

int ph_model_read_slot(ph_model* m, int32_t slot, float* out_dev, int64_t out_numel, void* stream) {
  PH_REQUIRE(m && out_dev && m->last_ws, "ph_model_read_slot: no forward has run");
  PH_REQUIRE(slot >= 0 && slot < m->n_slots && m->last_plan.slots[slot].offset >= 0, "bad slot %d", slot);
  PH_REQUIRE(!m->last_plan.reuse, "activations are recycled in this plan: set the handle option workspace_reuse to 0 before the forward to read a slot back");
  const SlotShape& s = m->last_plan.slots[slot];
  PH_REQUIRE(out_numel == (int64_t)m->last_batch * s.c * s.h * s.w, "slot %d has %d x %d x %d x %d elements", slot, m->last_batch, s.c, s.h, s.w);
  if (m->last_plan.fmt != FMT_F32)
    return launch_slot_to_nchw_fmt(m->last_plan.fmt, m->last_ws + s.offset, out_dev, m->last_batch, s.h * s.w, s.cp, s.c, static_cast<hipStream_t>(stream));
  return launch_nhwc_to_nchw(reinterpret_cast<float*>(m->last_ws + s.offset), out_dev, m->last_batch, s.h * s.w, s.cp, s.c,
                             static_cast<hipStream_t>(stream));
}

}  // extern "C"
