// Post-processing kernels for gfx950: local / global peak finding with integral refinement
// and PAF line-integral scoring.  HBM-bound integer/index work: coalesced row reads,
// wavefront ballots/prefix sums for the ordered compaction, no atomics (deterministic).
//
// Reference semantics (paths relative to talmolab/sleap-nn):
//   find_local_peaks[_rough] ..... sleap_nn/inference/ops/peaks.py:184-259
//   morphological_dilation ....... peaks.py:26-63  (8 neighbours, -inf outside)
//   integral_regression .......... peaks.py:66-86 + ops/crops.py:85-124 (zero padded crop)
//   find_global_peaks[_rough] .... peaks.py:89-181
//   get_connection_candidates .... sleap_nn/inference/ops/paf.py:84-130
//   make_line_subs / interp1d .... paf.py:133-234, sleap_nn/inference/utils.py:29-130
//   score_paf_lines .............. paf.py:290-410
#include <algorithm>

#include "common.h"

namespace ph {

// ---------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int block_exclusive_scan_256(int v, int* total, int* lds /* >= 8 ints */) {
  // inclusive scan inside the wave via shuffles, then across the 4 waves through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int n = __shfl_up(inc, d, 64);
    if (lane >= d) inc += n;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int s = lds[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__device__ __forceinline__ bool is_local_peak(const float* __restrict__ plane, int H, int W, int y, int x, float thr) {
  const float v = plane[(size_t)y * W + x];
  if (!(v > thr)) return false;
  bool ok = true;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    const int yy = y + dy;
    if (yy < 0 || yy >= H) continue;
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      if (dx == 0 && dy == 0) continue;
      const int xx = x + dx;
      if (xx < 0 || xx >= W) continue;
      ok = ok && (v > plane[(size_t)yy * W + xx]);
    }
  }
  return ok;
}

// first moments over a patch x patch zero padded window centred on the integer peak
__device__ __forceinline__ void integral_offset(const float* __restrict__ plane, int H, int W, int px, int py, int patch, float* dx, float* dy) {
  const int half = patch / 2;
  // top-left = trunc((p - (patch-1)/2) + half) - half  (crops.py:85-90; exact for integer p)
  const float tlx = (float)px - (patch - 1) * 0.5f, tly = (float)py - (patch - 1) * 0.5f;
  const int x0 = (int)(tlx + half) - half, y0 = (int)(tly + half) - half;
  const float g0 = -(patch - 1) * 0.5f;
  float z = 0.f, sx = 0.f, sy = 0.f;
  for (int j = 0; j < patch; ++j) {
    const int yy = y0 + j;
    const float gy = g0 + j;
    for (int i = 0; i < patch; ++i) {
      const int xx = x0 + i;
      float v = 0.f;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = plane[(size_t)yy * W + xx];
      z = __fadd_rn(z, v);
      sx = __fadd_rn(sx, __fmul_rn(g0 + i, v));
      sy = __fadd_rn(sy, __fmul_rn(gy, v));
    }
  }
  *dx = sx / z;
  *dy = sy / z;
}

// ---------------------------------------------------------------------------------------
// K9: local peaks.  Pass 1: one block per (sample, row) counts the peaks of the row.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void peaks_count_kernel(const float* __restrict__ cms, int C, int H, int W, float thr, int* __restrict__ row_count) {
  __shared__ int red[8];
  const int row = blockIdx.x;  // b*H + y
  const int b = row / H, y = row - b * H;
  int cnt = 0;
  for (int x = threadIdx.x; x < W; x += 256)
    for (int c = 0; c < C; ++c) cnt += is_local_peak(cms + ((size_t)b * C + c) * H * W, H, W, y, x, thr) ? 1 : 0;
  int tot;
  block_exclusive_scan_256(cnt, &tot, red);
  if (threadIdx.x == 0) row_count[row] = tot;
}

// Pass 2: exclusive scan of the B*H row counts (single block), per-sample totals.
__global__ __launch_bounds__(256) void peaks_scan_kernel(const int* __restrict__ row_count, int* __restrict__ row_offset, int n_rows, int H, int B, int* __restrict__ out_count) {
  __shared__ int red[8];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n_rows; base += 256) {
    const int i = base + threadIdx.x;
    const int v = i < n_rows ? row_count[i] : 0;
    int tot;
    const int ex = block_exclusive_scan_256(v, &tot, red);
    if (i < n_rows) row_offset[i] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    row_offset[n_rows] = carry;
    out_count[0] = carry;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += 256) {
    const int s = row_offset[b * H];
    const int e = (b + 1 < B) ? row_offset[(b + 1) * H] : row_offset[n_rows];
    out_count[1 + b] = e - s;
    out_count[1 + B + b] = s;  // exclusive offsets, [1+B+B] = total
  }
  if (threadIdx.x == 0) out_count[1 + 2 * B] = row_offset[n_rows];
}

// Pass 3: emit in (sample, y, x, channel) order, refine.
__global__ __launch_bounds__(256) void peaks_emit_kernel(const float* __restrict__ cms, int C, int H, int W, float thr, int refine, int patch,
                                                         const int* __restrict__ row_offset, float* __restrict__ out_xy, float* __restrict__ out_val,
                                                         int* __restrict__ out_sample, int* __restrict__ out_channel, int cap, float xy_scale) {
  __shared__ int red[8];
  const int row = blockIdx.x;
  const int b = row / H, y = row - b * H;
  int base = row_offset[row];
  if (row_offset[row + 1] == base) return;  // empty row (uniform per block)
  for (int xb = 0; xb < W; xb += 256) {
    const int x = xb + threadIdx.x;
    int cnt = 0;
    if (x < W)
      for (int c = 0; c < C; ++c) cnt += is_local_peak(cms + ((size_t)b * C + c) * H * W, H, W, y, x, thr) ? 1 : 0;
    int tot;
    int off = base + block_exclusive_scan_256(cnt, &tot, red);
    if (cnt > 0) {
      for (int c = 0; c < C; ++c) {
        const float* plane = cms + ((size_t)b * C + c) * H * W;
        if (!is_local_peak(plane, H, W, y, x, thr)) continue;
        if (off < cap) {
          float fx = (float)x, fy = (float)y;
          if (refine) {
            float dx, dy;
            integral_offset(plane, H, W, x, y, patch, &dx, &dy);
            fx += dx;
            fy += dy;
          }
          out_xy[2 * (size_t)off] = fx * xy_scale;  // one fp32 multiply after the refinement: `peaks * output_stride` of the callers
          out_xy[2 * (size_t)off + 1] = fy * xy_scale;
          out_val[off] = plane[(size_t)y * W + x];
          out_sample[off] = b;
          out_channel[off] = c;
        }
        ++off;
      }
    }
    base += tot;
  }
}

int launch_local_peaks(const float* cms, int B, int C, int H, int W, float thr, int refine, int patch, float* out_xy, float* out_val,
                       int* out_sample, int* out_channel, int* out_count, int cap, float xy_scale, int* scratch, hipStream_t s) {
  const int n_rows = B * H;
  int* row_count = scratch;              // n_rows
  int* row_offset = scratch + n_rows;    // n_rows + 1
  hipLaunchKernelGGL(peaks_count_kernel, dim3(n_rows), dim3(256), 0, s, cms, C, H, W, thr, row_count);
  hipLaunchKernelGGL(peaks_scan_kernel, dim3(1), dim3(256), 0, s, row_count, row_offset, n_rows, H, B, out_count);
  hipLaunchKernelGGL(peaks_emit_kernel, dim3(n_rows), dim3(256), 0, s, cms, C, H, W, thr, refine, patch, row_offset, out_xy, out_val, out_sample,
                     out_channel, cap, xy_scale);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// K11: global peaks.  One block per (sample, channel) plane: max value, then the first
// column and (independently) the first row that contain it.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void global_peaks_kernel(const float* __restrict__ cms, int H, int W, float thr, int refine, int patch,
                                                           float* __restrict__ out_xy, float* __restrict__ out_val) {
  __shared__ float smax[4];
  __shared__ int sminx[4], sminy[4];
  const int pc = blockIdx.x;
  const float* plane = cms + (size_t)pc * H * W;
  const int n = H * W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, plane[i]);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  if (lane == 0) smax[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  int mx = 0x7fffffff, my = 0x7fffffff;
  for (int i = threadIdx.x; i < n; i += 256)
    if (plane[i] == m) {
      const int y = i / W, x = i - y * W;
      mx = min(mx, x);
      my = min(my, y);
    }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    mx = min(mx, __shfl_xor(mx, d, 64));
    my = min(my, __shfl_xor(my, d, 64));
  }
  if (lane == 0) {
    sminx[wave] = mx;
    sminy[wave] = my;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = min(min(sminx[0], sminx[1]), min(sminx[2], sminx[3]));
    my = min(min(sminy[0], sminy[1]), min(sminy[2], sminy[3]));
    if (m < thr || mx == 0x7fffffff) {
      const float nanv = __builtin_nanf("");
      out_xy[2 * (size_t)pc] = (m < thr) ? nanv : 0.f;
      out_xy[2 * (size_t)pc + 1] = (m < thr) ? nanv : 0.f;
      out_val[pc] = (m < thr) ? 0.f : m;
    } else {
      float fx = (float)mx, fy = (float)my;
      if (refine) {
        float dx, dy;
        integral_offset(plane, H, W, mx, my, patch, &dx, &dy);
        fx += dx;
        fy += dy;
      }
      out_xy[2 * (size_t)pc] = fx;
      out_xy[2 * (size_t)pc + 1] = fy;
      out_val[pc] = m;
    }
  }
}

int launch_global_peaks(const float* cms, int B, int C, int H, int W, float thr, int refine, int patch, float* out_xy, float* out_val, hipStream_t s) {
  hipLaunchKernelGGL(global_peaks_kernel, dim3(B * C), dim3(256), 0, s, cms, H, W, thr, refine, patch, out_xy, out_val);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// K12: PAF scoring.
//  paf_index_kernel (1 block / sample): per-node peak lists (stable = ascending peak index),
//     per-edge candidate counts n_src*n_dst and their exclusive offsets.
//  paf_scan_kernel: exclusive scan of the per-sample candidate totals.
//  paf_score_kernel: one thread per candidate: n_points samples on the src->dst segment,
//     nearest-pixel (round-half-even) gather from the NCHW PAFs, dot with the unit vector,
//     mean + distance penalty.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void paf_index_kernel(const int* __restrict__ chan, const int* __restrict__ peak_off, int n_nodes,
                                                        const int* __restrict__ edges, int n_edges, int* __restrict__ sorted_idx,
                                                        int* __restrict__ node_start /* B*(n_nodes+1) */, int* __restrict__ edge_start /* B*(E+1) */,
                                                        int* __restrict__ cand_count /* B */, int n_peaks_cap) {
  const int b = blockIdx.x;
  // offsets come from the peak kernel's TRUE counts; rows beyond the caller's capacity were
  // never written, so clamp (the caller sees count > cap and retries with more room)
  const int p0 = min(peak_off[b], n_peaks_cap), n = min(peak_off[b + 1], n_peaks_cap) - p0;
  int* ns = node_start + (size_t)b * (n_nodes + 1);
  int* es = edge_start + (size_t)b * (n_edges + 1);
  // node histogram -> starts (serial over nodes; n_nodes is small)
  for (int k = threadIdx.x; k <= n_nodes; k += 256) ns[k] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    const int k = chan[p0 + i];
    if (k >= 0 && k < n_nodes) atomicAdd(&ns[k + 1], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int k = 0; k < n_nodes; ++k) {
      const int c = ns[k + 1];
      ns[k] = run;
      run += c;
    }
    ns[n_nodes] = run;
  }
  __syncthreads();
  // stable rank inside the node = number of earlier peaks on the same node
  for (int i = threadIdx.x; i < n; i += 256) {
    const int k = chan[p0 + i];
    if (k < 0 || k >= n_nodes) continue;
    int r = 0;
    for (int j = 0; j < i; ++j) r += (chan[p0 + j] == k) ? 1 : 0;
    sorted_idx[p0 + ns[k] + r] = i;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int e = 0; e < n_edges; ++e) {
      const int s = edges[2 * e], d = edges[2 * e + 1];
      es[e] = run;
      run += (ns[s + 1] - ns[s]) * (ns[d + 1] - ns[d]);
    }
    es[n_edges] = run;
    cand_count[b] = run;
  }
}

__global__ void paf_scan_kernel(const int* __restrict__ cand_count, int B, int* __restrict__ cand_offsets) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int run = 0;
    for (int b = 0; b < B; ++b) {
      cand_offsets[b] = run;
      run += cand_count[b];
    }
    cand_offsets[B] = run;
  }
}

struct PafScoreArgs {
  const float* pafs;
  int B, E2, H, W;
  const float* peaks_xy;
  const int* peak_off;
  const int* sorted_idx;
  const int* node_start;
  const int* edge_start;
  const int* edges;
  int n_nodes, n_edges;
  const float* t;
  int n_points, stride;
  float max_edge_length, dist_penalty_weight;
  const int* cand_offsets;
  int *cand_edge, *cand_src, *cand_dst;
  float* cand_score;
  int cap;
  int n_peaks_cap;
};

__global__ __launch_bounds__(256) void paf_score_kernel(PafScoreArgs a) {
  const int b = blockIdx.y;
  const int* ns = a.node_start + (size_t)b * (a.n_nodes + 1);
  const int* es = a.edge_start + (size_t)b * (a.n_edges + 1);
  const int total = es[a.n_edges];
  const int p0 = min(a.peak_off[b], a.n_peaks_cap);
  const int out0 = a.cand_offsets[b];
  const float den = 1.0f + 1.1920928955078125e-07f;  // eps + (x1 - x0), fp32 (utils.py:48,120-125)
  const size_t plane = (size_t)a.H * a.W;
  for (int q = blockIdx.x * 256 + threadIdx.x; q < total; q += gridDim.x * 256) {
    // edge of candidate q (edge offsets are non-decreasing; E is small)
    int e = 0;
    while (e + 1 < a.n_edges && es[e + 1] <= q) ++e;
    const int sn = a.edges[2 * e], dn = a.edges[2 * e + 1];
    const int nd = ns[dn + 1] - ns[dn];
    const int local = q - es[e];
    const int si = local / nd, di = local - si * nd;
    const int src = a.sorted_idx[p0 + ns[sn] + si];
    const int dst = a.sorted_idx[p0 + ns[dn] + di];
    const float sx = a.peaks_xy[2 * (size_t)(p0 + src)], sy = a.peaks_xy[2 * (size_t)(p0 + src) + 1];
    const float dx = a.peaks_xy[2 * (size_t)(p0 + dst)], dy = a.peaks_xy[2 * (size_t)(p0 + dst) + 1];
    const float vx = __fsub_rn(dx, sx), vy = __fsub_rn(dy, sy);
    const float slx = vx / den, sly = vy / den;
    const float len = sqrtf(__fadd_rn(__fmul_rn(vx, vx), __fmul_rn(vy, vy)));
    const float ux = vx / len, uy = vy / len;
    const float* fxp = a.pafs + ((size_t)b * a.E2 + 2 * e) * plane;
    const float* fyp = fxp + plane;
    float acc = 0.f;
    for (int k = 0; k < a.n_points; ++k) {
      const float t = a.t[k];
      const float px = __fadd_rn(sx, __fmul_rn(slx, t));
      const float py = __fadd_rn(sy, __fmul_rn(sly, t));
      int col = (int)rintf(px / (float)a.stride);
      int row = (int)rintf(py / (float)a.stride);
      col = min(max(col, 0), a.W - 1);
      row = min(max(row, 0), a.H - 1);
      const size_t o = (size_t)row * a.W + col;
      acc = __fadd_rn(acc, __fadd_rn(__fmul_rn(fxp[o], ux), __fmul_rn(fyp[o], uy)));
    }
    const float pen = fminf(a.max_edge_length / len - 1.f, 0.f) * a.dist_penalty_weight;
    const float score = acc / (float)a.n_points + pen;
    const int o = out0 + q;
    if (o < a.cap) {
      a.cand_edge[o] = e;
      a.cand_src[o] = src;
      a.cand_dst[o] = dst;
      a.cand_score[o] = score;
    }
  }
}

int launch_paf_score(PafScoreArgs a, const int* chan, int n_peaks_total, int* scratch, int* cand_offsets, hipStream_t s) {
  int* sorted_idx = scratch;                                       // n_peaks_total
  int* node_start = sorted_idx + n_peaks_total;                    // B*(n_nodes+1)
  int* edge_start = node_start + (size_t)a.B * (a.n_nodes + 1);    // B*(E+1)
  int* cand_count = edge_start + (size_t)a.B * (a.n_edges + 1);    // B
  hipLaunchKernelGGL(paf_index_kernel, dim3(a.B), dim3(256), 0, s, chan, a.peak_off, a.n_nodes, a.edges, a.n_edges, sorted_idx, node_start, edge_start,
                     cand_count, n_peaks_total);
  a.n_peaks_cap = n_peaks_total;
  hipLaunchKernelGGL(paf_scan_kernel, dim3(1), dim3(64), 0, s, cand_count, a.B, cand_offsets);
  a.sorted_idx = sorted_idx;
  a.node_start = node_start;
  a.edge_start = edge_start;
  a.cand_offsets = cand_offsets;
  hipLaunchKernelGGL(paf_score_kernel, dim3(8, a.B), dim3(256), 0, s, a);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

// ---------------------------------------------------------------------------------------
// Crop gather + class-map sampling (top-down / multi-class glue)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void crop_kernel(const T* __restrict__ img, int C, int H, int W, const float* __restrict__ tl, const int* __restrict__ sidx,
                                                   int n, int ch, int cw, T* __restrict__ out) {
  const size_t total = (size_t)n * C * ch * cw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % cw);
    size_t r = i / cw;
    const int y = (int)(r % ch);
    r /= ch;
    const int c = (int)(r % C);
    const int k = (int)(r / C);
    const int hx = cw / 2, hy = ch / 2;
    const int ox = (int)(tl[2 * k] + (float)hx) - hx;      // trunc toward zero, as .to(torch.long)
    const int oy = (int)(tl[2 * k + 1] + (float)hy) - hy;
    const int gx = ox + x, gy = oy + y;
    T v = (T)0;
    if (gx >= 0 && gx < W && gy >= 0 && gy < H) v = img[(((size_t)sidx[k] * C + c) * H + gy) * W + gx];
    out[i] = v;
  }
}

__global__ __launch_bounds__(256) void sample_class_maps_kernel(const float* __restrict__ maps, int K, int H, int W, const float* __restrict__ xy,
                                                                const int* __restrict__ sidx, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * K) return;
  const int p = i / K, k = i - p * K;
  int col = (int)rintf(xy[2 * p]), row = (int)rintf(xy[2 * p + 1]);
  col = min(max(col, 0), W - 1);
  row = min(max(row, 0), H - 1);
  out[i] = maps[(((size_t)sidx[p] * K + k) * H + row) * W + col];
}

}  // namespace ph

using namespace ph;

extern "C" {

int ph_crop_bboxes(const void* images_dev, int32_t dtype, int32_t B, int32_t C, int32_t H, int32_t W, const float* topleft_xy_dev,
                   const int32_t* sample_inds_dev, int32_t n, int32_t crop_h, int32_t crop_w, void* out_dev, void* stream) {
  PH_REQUIRE(images_dev && out_dev && (n == 0 || (topleft_xy_dev && sample_inds_dev)), "ph_crop_bboxes: null argument");
  PH_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && n >= 0 && crop_h > 0 && crop_w > 0 && (dtype == 0 || dtype == 1), "ph_crop_bboxes: bad shape/dtype");
  if (n == 0) return PH_OK;
  const size_t total = (size_t)n * C * crop_h * crop_w;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 65535);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == 0)
    hipLaunchKernelGGL(crop_kernel<uint8_t>, dim3(blocks), dim3(256), 0, s, static_cast<const uint8_t*>(images_dev), C, H, W, topleft_xy_dev,
                       sample_inds_dev, n, crop_h, crop_w, static_cast<uint8_t*>(out_dev));
  else
    hipLaunchKernelGGL(crop_kernel<float>, dim3(blocks), dim3(256), 0, s, static_cast<const float*>(images_dev), C, H, W, topleft_xy_dev,
                       sample_inds_dev, n, crop_h, crop_w, static_cast<float*>(out_dev));
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

int ph_sample_class_maps(const float* class_maps_dev, int32_t B, int32_t K, int32_t H, int32_t W, const float* peaks_xy_dev,
                         const int32_t* sample_inds_dev, int32_t n, float* out_probs_dev, void* stream) {
  PH_REQUIRE(class_maps_dev && (n == 0 || (peaks_xy_dev && sample_inds_dev && out_probs_dev)), "ph_sample_class_maps: null argument");
  PH_REQUIRE(B > 0 && K > 0 && H > 0 && W > 0 && n >= 0, "ph_sample_class_maps: bad shape");
  if (n == 0) return PH_OK;
  hipLaunchKernelGGL(sample_class_maps_kernel, dim3((n * K + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), class_maps_dev, K, H, W,
                     peaks_xy_dev, sample_inds_dev, n, out_probs_dev);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

int ph_local_peaks(const float* cms_dev, int32_t B, int32_t C, int32_t H, int32_t W, float threshold, int32_t refine, int32_t patch, float* out_xy,
                   float* out_val, int32_t* out_sample, int32_t* out_channel, int32_t* out_count, int32_t cap, float xy_scale,
                   void* scratch_dev, int64_t scratch_bytes, void* stream) {
  PH_REQUIRE(cms_dev && out_xy && out_val && out_sample && out_channel && out_count && scratch_dev, "ph_local_peaks: null argument");
  PH_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && cap >= 0, "ph_local_peaks: bad shape");
  PH_REQUIRE(patch >= 1 && (patch & 1), "ph_local_peaks: integral patch must be odd");
  PH_REQUIRE(scratch_bytes >= (int64_t)4 * (2 * (int64_t)B * H + 2), "ph_local_peaks: scratch too small");
  return launch_local_peaks(cms_dev, B, C, H, W, threshold, refine, patch, out_xy, out_val, out_sample, out_channel, out_count, cap, xy_scale,
                            static_cast<int*>(scratch_dev), static_cast<hipStream_t>(stream));
}

int ph_global_peaks(const float* cms_dev, int32_t B, int32_t C, int32_t H, int32_t W, float threshold, int32_t refine, int32_t patch, float* out_xy,
                    float* out_val, void* stream) {
  PH_REQUIRE(cms_dev && out_xy && out_val, "ph_global_peaks: null argument");
  PH_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "ph_global_peaks: bad shape");
  PH_REQUIRE(patch >= 1 && (patch & 1), "ph_global_peaks: integral patch must be odd");
  return launch_global_peaks(cms_dev, B, C, H, W, threshold, refine, patch, out_xy, out_val, static_cast<hipStream_t>(stream));
}

int ph_paf_score(const float* pafs_dev, int32_t B, int32_t E2, int32_t H, int32_t W, const float* peaks_xy_dev, const int32_t* peak_channel_dev,
                 const int32_t* peak_offsets_dev, int32_t n_peaks_total, int32_t n_nodes, const int32_t* edges_dev, int32_t n_edges,
                 const float* t_dev, int32_t n_points, int32_t pafs_stride, float max_edge_length, float dist_penalty_weight, int32_t* cand_edge,
                 int32_t* cand_src, int32_t* cand_dst, float* cand_score, int32_t* cand_offsets, int32_t cap, void* scratch_dev,
                 int64_t scratch_bytes, void* stream) {
  PH_REQUIRE(pafs_dev && peaks_xy_dev && peak_channel_dev && peak_offsets_dev && edges_dev && t_dev && cand_edge && cand_src && cand_dst &&
                 cand_score && cand_offsets && scratch_dev,
             "ph_paf_score: null argument");
  PH_REQUIRE(B > 0 && n_edges > 0 && E2 == 2 * n_edges && n_nodes > 0 && n_points > 0 && pafs_stride > 0, "ph_paf_score: bad shape");
  const int64_t need = 4 * ((int64_t)n_peaks_total + (int64_t)B * (n_nodes + 1) + (int64_t)B * (n_edges + 1) + B);
  PH_REQUIRE(scratch_bytes >= need, "ph_paf_score: scratch too small (%lld < %lld)", (long long)scratch_bytes, (long long)need);
  PafScoreArgs a;
  a.pafs = pafs_dev;
  a.B = B;
  a.E2 = E2;
  a.H = H;
  a.W = W;
  a.peaks_xy = peaks_xy_dev;
  a.peak_off = peak_offsets_dev;
  a.edges = edges_dev;
  a.n_nodes = n_nodes;
  a.n_edges = n_edges;
  a.t = t_dev;
  a.n_points = n_points;
  a.stride = pafs_stride;
  a.max_edge_length = max_edge_length;
  a.dist_penalty_weight = dist_penalty_weight;
  a.cand_edge = cand_edge;
  a.cand_src = cand_src;
  a.cand_dst = cand_dst;
  a.cand_score = cand_score;
  a.cap = cap;
  return launch_paf_score(a, peak_channel_dev, n_peaks_total, static_cast<int*>(scratch_dev), cand_offsets, static_cast<hipStream_t>(stream));
}

}  // extern "C"
