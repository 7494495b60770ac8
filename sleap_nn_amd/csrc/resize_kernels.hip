// Antialiased bilinear resize (the sizematcher / input-scale step of the preprocessing chain).
//
// Replaces torchvision.transforms.v2.functional.resize as called at sleap_nn/data/resizing.py:83 (resize_image) and :158
// (apply_sizematcher): for tensors that is torch.nn.functional.interpolate(mode="bilinear", align_corners=False,
// antialias=True).  torchvision is not vendored in the reference tree; the arithmetic restated here is ATen's separable
// antialias kernel (aten/src/ATen/native/cpu/UpSampleKernel.cpp, HelperInterpLinear::aa_filter and
// _compute_indices_int16_weights_aa), pinned by tests against torch's own operator:
//   * per output index: centre = scale (i + 0.5), support = max(scale, 1), taps [xmin, xmin + xsize), triangle weights
//     normalised to sum 1;
//   * uint8 frames (what the reference's layers keep, base.py:212-253): weights as int16 fixed point with the largest precision
//     whose maximum weight stays below 2^15, accumulate from 2^(precision - 1), shift, clamp -- horizontal pass first into a
//     uint8 intermediate, then vertical (bit-exact with the CPU operator);
//   * float frames: float weights and accumulation in tap order (matches to fp32 rounding).
// HBM-bound, one thread per output element; the per-axis tables are built on the host once per (in, out) size and cached.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "common.h"

namespace ph {

struct ResizeTable {
  int32_t* start = nullptr;  // [out] first tap
  int32_t* count = nullptr;  // [out] number of taps
  int32_t* wi = nullptr;     // [out][max_taps] int16 fixed-point weights (as int32)
  float* wf = nullptr;       // [out][max_taps] float weights
  int max_taps = 0, precision = 0;
};

static int build_table(int in_size, int out_size, ResizeTable& t) {
  // double arithmetic for the fixed-point table (ATen computes it in double), float for the float table (scalar_t = float)
  const double scale = (double)in_size / (double)out_size;
  const double support = scale >= 1.0 ? scale : 1.0;
  const double invscale = scale >= 1.0 ? 1.0 / scale : 1.0;
  const int max_taps = (int)std::ceil(support) * 2 + 1;
  std::vector<int32_t> start(out_size), count(out_size), wi((size_t)out_size * max_taps, 0);
  std::vector<double> w((size_t)out_size * max_taps, 0.0);
  std::vector<float> wf((size_t)out_size * max_taps, 0.f);
  const float scale_f = (float)in_size / (float)out_size;
  const float support_f = scale_f >= 1.0f ? scale_f : 1.0f;
  const float invscale_f = scale_f >= 1.0f ? 1.0f / scale_f : 1.0f;
  double w_max = 0.0;
  for (int i = 0; i < out_size; ++i) {
    const double center = scale * (i + 0.5);
    const int lo = std::max((int)(int64_t)(center - support + 0.5), 0);
    const int n = std::min((int)(int64_t)(center + support + 0.5), in_size) - lo;
    if (n < 1 || n > max_taps) {
      set_error("resize: unexpected tap count %d (in %d, out %d)", n, in_size, out_size);
      return PH_E_INVALID;
    }
    start[i] = lo;
    count[i] = n;
    double total = 0.0;
    for (int j = 0; j < n; ++j) {
      const double x = std::fabs((j + lo - center + 0.5) * invscale);
      const double v = x < 1.0 ? 1.0 - x : 0.0;
      w[(size_t)i * max_taps + j] = v;
      total += v;
    }
    for (int j = 0; j < n; ++j) {
      if (total != 0.0) w[(size_t)i * max_taps + j] /= total;
      w_max = std::max(w_max, w[(size_t)i * max_taps + j]);
    }
    // float table: the same recipe in float arithmetic, with its own tap range (kept inside [lo, lo + n) of the double table)
    const float center_f = scale_f * ((float)i + 0.5f);
    float total_f = 0.f;
    for (int j = 0; j < n; ++j) {
      const float x = std::fabs(((float)(j + lo) - center_f + 0.5f) * invscale_f);
      const float v = x < 1.0f ? 1.0f - x : 0.f;
      wf[(size_t)i * max_taps + j] = v;
      total_f += v;
    }
    if (total_f != 0.f)
      for (int j = 0; j < n; ++j) wf[(size_t)i * max_taps + j] /= total_f;
  }
  int precision = 0;
  for (precision = 0; precision < 22; ++precision) {
    const int next_value = (int)(0.5 + w_max * (double)(1 << (precision + 1)));
    if (next_value >= (1 << 15)) break;
  }
  for (size_t k = 0; k < w.size(); ++k) {
    const double v = w[k] * (double)(1 << precision);
    wi[k] = v < 0 ? (int)(-0.5 + v) : (int)(0.5 + v);
  }
  t.max_taps = max_taps;
  t.precision = precision;
  PH_HIP_CHECK(hipMalloc(&t.start, out_size * sizeof(int32_t)));
  PH_HIP_CHECK(hipMalloc(&t.count, out_size * sizeof(int32_t)));
  PH_HIP_CHECK(hipMalloc(&t.wi, wi.size() * sizeof(int32_t)));
  PH_HIP_CHECK(hipMalloc(&t.wf, wf.size() * sizeof(float)));
  PH_HIP_CHECK(hipMemcpy(t.start, start.data(), out_size * sizeof(int32_t), hipMemcpyHostToDevice));
  PH_HIP_CHECK(hipMemcpy(t.count, count.data(), out_size * sizeof(int32_t), hipMemcpyHostToDevice));
  PH_HIP_CHECK(hipMemcpy(t.wi, wi.data(), wi.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  PH_HIP_CHECK(hipMemcpy(t.wf, wf.data(), wf.size() * sizeof(float), hipMemcpyHostToDevice));
  return PH_OK;
}

// tables are cached per (device, in, out) for the life of the process (a few KiB each)
static int get_table(int in_size, int out_size, const ResizeTable** out) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, int>, ResizeTable> cache;
  int dev = 0;
  PH_HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_tuple(dev, in_size, out_size);
  auto it = cache.find(key);
  if (it == cache.end()) {
    ResizeTable t;
    int rc = build_table(in_size, out_size, t);
    if (rc != PH_OK) return rc;
    it = cache.emplace(key, t).first;
  }
  *out = &it->second;
  return PH_OK;
}

// One axis of the separable resize.  The tensor is seen as [outer][len][inner]: inner = 1 for the horizontal pass (len = W),
// inner = W for the vertical one (len = H).  thread = one output element.
template <typename T>
__global__ __launch_bounds__(256) void resize_axis_kernel(const T* __restrict__ src, T* __restrict__ dst, size_t outer, int in_len, int out_len, size_t inner,
                                                          const int32_t* __restrict__ start, const int32_t* __restrict__ count, const int32_t* __restrict__ wi,
                                                          const float* __restrict__ wf, int max_taps, int precision) {
  const size_t total = outer * (size_t)out_len * inner;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const size_t in_i = idx % inner;
    const size_t r = idx / inner;
    const int o = (int)(r % out_len);
    const size_t ou = r / out_len;
    const int lo = start[o], n = count[o];
    const T* p = src + (ou * in_len + lo) * inner + in_i;
    if constexpr (sizeof(T) == 1) {
      const int32_t* w = wi + (size_t)o * max_taps;
      int acc = 1 << (precision - 1);
      for (int j = 0; j < n; ++j) acc += (int)p[(size_t)j * inner] * w[j];
      acc >>= precision;
      dst[idx] = (T)min(max(acc, 0), 255);
    } else {
      const float* w = wf + (size_t)o * max_taps;
      float acc = (float)p[0] * w[0];
      for (int j = 1; j < n; ++j) acc += (float)p[(size_t)j * inner] * w[j];
      dst[idx] = (T)acc;
    }
  }
}

template <typename T>
static int run_resize(const T* src, int planes, int H, int W, T* dst, int OH, int OW, T* tmp, hipStream_t s) {
  const T* cur = src;
  if (OW != W) {
    const ResizeTable* t = nullptr;
    int rc = get_table(W, OW, &t);
    if (rc != PH_OK) return rc;
    T* out = (OH != H) ? tmp : dst;
    const size_t total = (size_t)planes * H * OW;
    hipLaunchKernelGGL(resize_axis_kernel<T>, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0, s, cur, out, (size_t)planes * H, W, OW, (size_t)1,
                       t->start, t->count, t->wi, t->wf, t->max_taps, t->precision);
    cur = out;
  }
  if (OH != H) {
    const ResizeTable* t = nullptr;
    int rc = get_table(H, OH, &t);
    if (rc != PH_OK) return rc;
    const size_t total = (size_t)planes * OH * OW;
    hipLaunchKernelGGL(resize_axis_kernel<T>, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0, s, cur, dst, (size_t)planes, H, OH, (size_t)OW,
                       t->start, t->count, t->wi, t->wf, t->max_taps, t->precision);
  } else if (OW == W) {
    PH_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)planes * H * W * sizeof(T), hipMemcpyDeviceToDevice, s));
  }
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

}  // namespace ph

extern "C" int ph_resize_bilinear_aa(const void* src_dev, int32_t dtype, int32_t planes, int32_t H, int32_t W, void* dst_dev, int32_t OH, int32_t OW, void* tmp_dev,
                                     void* stream) {
  using namespace ph;
  PH_REQUIRE(src_dev && dst_dev, "ph_resize_bilinear_aa: null pointer");
  PH_REQUIRE(planes > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "ph_resize_bilinear_aa: bad shape %d x %d x %d -> %d x %d", planes, H, W, OH, OW);
  PH_REQUIRE(dtype == 0 || dtype == 1, "ph_resize_bilinear_aa: dtype must be 0 (uint8) or 1 (float32)");
  PH_REQUIRE(tmp_dev || OH == H || OW == W, "ph_resize_bilinear_aa: a two-axis resize needs the planes*H*OW intermediate buffer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == 0) return run_resize(static_cast<const uint8_t*>(src_dev), planes, H, W, static_cast<uint8_t*>(dst_dev), OH, OW, static_cast<uint8_t*>(tmp_dev), s);
  return run_resize(static_cast<const float*>(src_dev), planes, H, W, static_cast<float*>(dst_dev), OH, OW, static_cast<float*>(tmp_dev), s);
}
