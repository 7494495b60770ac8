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

// the same test with the eight neighbours requested together (clamped addresses, out-of-image ones masked afterwards): one memory latency per candidate
__device__ __forceinline__ bool is_local_peak_of_candidate(const float* __restrict__ plane, int H, int W, int y, int x, float v) {
  float nb[3][3];
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) nb[dy + 1][dx + 1] = plane[(size_t)min(max(y + dy, 0), H - 1) * W + min(max(x + dx, 0), W - 1)];
  bool ok = true;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      if (dx == 0 && dy == 0) continue;
      const bool in = y + dy >= 0 && y + dy < H && x + dx >= 0 && x + dx < W;
      ok = ok & (!in | (v > nb[dy + 1][dx + 1]));  // (v > NaN is false: a value next to a NaN is no peak)
    }
  return ok;
}

// The neighbourhood test AND the integral refinement of a candidate from ONE window of loads (PATCH x PATCH >= 3 x 3 around the pixel, clamped addresses, all requested
// together): the test reads the window's centre 3 x 3 with out-of-image neighbours skipped, the refinement its zero-padded values -- the same arithmetic, in the same order,
// as is_local_peak_of_candidate + integral_offset_fixed<PATCH>.
template <int PATCH>
__device__ __forceinline__ bool test_and_refine(const float* __restrict__ plane, int H, int W, int px, int py, float* val, float* dx, float* dy) {
  constexpr int half = PATCH / 2;
  const float g0 = -(PATCH - 1) * 0.5f;
  float w[PATCH][PATCH];
#pragma unroll
  for (int j = 0; j < PATCH; ++j)
#pragma unroll
    for (int i = 0; i < PATCH; ++i) w[j][i] = plane[(size_t)min(max(py - half + j, 0), H - 1) * W + min(max(px - half + i, 0), W - 1)];
  const float v = w[half][half];
  bool ok = true;
#pragma unroll
  for (int dj = -1; dj <= 1; ++dj)
#pragma unroll
    for (int di = -1; di <= 1; ++di) {
      if (di == 0 && dj == 0) continue;
      const bool in = py + dj >= 0 && py + dj < H && px + di >= 0 && px + di < W;
      ok = ok & (!in | (v > w[half + dj][half + di]));
    }
  float z = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
  for (int j = 0; j < PATCH; ++j)
#pragma unroll
    for (int i = 0; i < PATCH; ++i) {
      const int yy = py - half + j, xx = px - half + i;
      const float t = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? w[j][i] : 0.f;
      z = __fadd_rn(z, t);
      sx = __fadd_rn(sx, __fmul_rn(g0 + i, t));
      sy = __fadd_rn(sy, __fmul_rn(g0 + j, t));
    }
  *val = v;
  *dx = sx / z;
  *dy = sy / z;
  return ok;
}

// first moments over a patch x patch zero padded window centred on the integer peak.  PATCH > 0: compile-time size -- the window's loads are issued together (independent
// addresses) and only the sums are chained, in the same (row, column) order as the run-time loop: same bits, one memory latency instead of patch^2 of them (the loop form
// cost ~25 us per peak under load and set the time of the whole peak kernel).
template <int PATCH>
__device__ __forceinline__ void integral_offset_fixed(const float* __restrict__ plane, int H, int W, int px, int py, float* dx, float* dy) {
  constexpr int half = PATCH / 2;
  const float tlx = (float)px - (PATCH - 1) * 0.5f, tly = (float)py - (PATCH - 1) * 0.5f;
  const int x0 = (int)(tlx + half) - half, y0 = (int)(tly + half) - half;
  const float g0 = -(PATCH - 1) * 0.5f;
  float w[PATCH][PATCH];
#pragma unroll
  for (int j = 0; j < PATCH; ++j)
#pragma unroll
    for (int i = 0; i < PATCH; ++i) {
      const int yy = y0 + j, xx = x0 + i;
      w[j][i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? plane[(size_t)yy * W + xx] : 0.f;
    }
  float z = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
  for (int j = 0; j < PATCH; ++j)
#pragma unroll
    for (int i = 0; i < PATCH; ++i) {
      z = __fadd_rn(z, w[j][i]);
      sx = __fadd_rn(sx, __fmul_rn(g0 + i, w[j][i]));
      sy = __fadd_rn(sy, __fmul_rn(g0 + j, w[j][i]));
    }
  *dx = sx / z;
  *dy = sy / z;
}
__device__ __forceinline__ void integral_offset(const float* __restrict__ plane, int H, int W, int px, int py, int patch, float* dx, float* dy) {
  if (patch == 5) return integral_offset_fixed<5>(plane, H, W, px, py, dx, dy);
  if (patch == 3) return integral_offset_fixed<3>(plane, H, W, px, py, dx, dy);
  if (patch == 7) return integral_offset_fixed<7>(plane, H, W, px, py, dx, dy);
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

// ---------------------------------------------------------------------------------------
// K9b: local peaks in ONE pass over the confidence maps + a placement pass over the (few) peaks found.
//   peaks_onepass_kernel: one block per (sample, group of OP_R rows).  Every map value is loaded exactly once by the lane that owns its column quad; the few values
//     that are above the threshold and beat the neighbours their wave holds take the strict 3x3 test (peaks.py:26-63,184-259: v > every in-image neighbour) and the
//     integral refinement from one window of loads; the survivors are ranked into the reference's (y, x, channel) order in the block's staging area and the block's
//     count is stored.
//   peaks_place_kernel: block i sums the counts of the blocks before it (fixed order: no atomics), copies its staged peaks to their final place and, in the rare case
//     that a block found more peaks than its staging area holds, recomputes that block's rows straight into the output (the legacy emit logic) -- the result never depends
//     on the staging capacity.  Block 0 also writes the totals and the per-sample counts / offsets.
// Two launches, the maps read once (the legacy path: count + scan + emit, two full reads with nine loads per value each).
// ---------------------------------------------------------------------------------------
#ifndef PH_PEAKS_EXP
#define PH_PEAKS_EXP 0  // timing experiments only (results wrong with any bit set): 1 = a block ends behind its loads + threshold test + candidate push, 16 = no refinement, 64 = the loads and the threshold test alone (nothing is pushed)
#endif
constexpr int OP_R = 8;      // rows per block
constexpr int OP_HALF = 4;   // rows per streaming item (a wave keeps two items = 8 KiB in flight)
constexpr int OP_STAGE = 512;  // staged peaks per block (16 B each)
constexpr int OP_CAND = 2048;  // candidates a block lists (more: its peaks are counted and the placement kernel lists them)
constexpr int OP_RECOMPUTE = 1 << 30;  // flag on a block's count: nothing is staged, the placement kernel recomputes the block's rows

// NCH: 256-column chunks of a row (W <= 256 NCH).  VEC: rows are 16-byte aligned (W a multiple of 4): one 16-byte load per lane and row.
//   streaming -- the block's work items are (channel, upper / lower OP_HALF rows); wave w takes items w, w + 4, ... (13 channels: 7 / 7 / 6 / 6 items) and has the NEXT
//     item's four 16-byte loads in flight while it tests one.  Every map value is read exactly once.  A window without a value above the threshold -- three in four -- is
//     done after 16 compares and a ballot; in the others the 3x3 neighbourhood maximum is formed in registers and only the values that beat it go on the (unordered) LDS
//     list: one or two per Gaussian blob instead of every pixel of it above the threshold;
//   listing -- one candidate per thread: its eight neighbours (peaks.py:26-63,184-259: v > every in-image neighbour, `v > NaN` false) and its refinement window are requested
//     together, ONE memory latency for the block's whole list; a surviving peak's place in the reference's (row, column, channel) order is the number of survivors with a
//     smaller key (a few compares); its thread stages it.
// Where the time goes (round 6, PH_PEAKS_EXP builds + `tools/peaks_bench.py zeros|sparse`, cfg3's 32 x 13 x 256 x 256 maps, rocprofv3 kernel durations): the loads and
// the threshold test alone 17.1 us (a bare read of the same bytes: 15.9), + 3.0 us for the windows that hold a blob, + 2.0 us for the listing (5.0 before the register
// filter: every pixel of a blob above the threshold had its 5x5 window loaded), + 4.8 us for the placement launch.  The + 3.0 is vector-instruction THROUGHPUT, not a
// tail: with blobs in one frame of eight the kernel takes what it takes on all-zero maps (20.8 vs 20.7 us per call), with blobs in every frame the four blocks of a CU all
// execute window code on the same four SIMDs.  The placement launch costs 4.8 us whatever it does (1 024 blocks summing 1 024 counts, the same with one memory latency
// less, or 33 blocks: 4.8 - 5.0 us) and folding it into a last-block-done merge was built and measured: slots of `epoch << 32 | count` that every finishing block
// publishes and scans (no spin, no shared counter, safe on uninitialised scratch) -- 49 us: a million cache-bypassing 8-byte loads on 8 KiB of slots, and the merging
// block's chain (publish, scan, acquire, copy) is as long as the launch it replaces.  Kept: two launches.
template <int NCH, bool VEC>
__global__ __launch_bounds__(256) void peaks_onepass_kernel(const float* __restrict__ cms, int C, int H, int W, float thr, int refine, int patch, float xy_scale,
                                                            int groups, int* __restrict__ blk_count, float* __restrict__ stg_xy, float* __restrict__ stg_val,
                                                            int* __restrict__ stg_ch) {
  __shared__ unsigned cand[OP_CAND];  // candidates, unordered: key = row << 16 | column << 6 | channel
  __shared__ unsigned pkey[OP_STAGE];  // surviving peaks, unordered: key
  __shared__ float4 pres[OP_STAGE];    // ... and (x, y, value, -)
  __shared__ int n_cand, n_pass;
  // Workgroups go round the eight XCDs: block id -> (sample, row group) so that an XCD walks CONSECUTIVE row groups (the candidates' neighbour rows in the next group are
  // then in that XCD's L2).  Everything the block writes is indexed by blk, the logical id.
  const int n_blk = (int)gridDim.x;
#ifdef PH_PEAKS_NOREMAP
  const int blk = (int)blockIdx.x;
#else
  const int blk = (n_blk & 7) == 0 ? ((int)blockIdx.x & 7) * (n_blk >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
#endif
  const int b = blk / groups, y0 = (blk - b * groups) * OP_R;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rows = min(OP_R, H - y0);
  if (threadIdx.x == 0) {
    n_cand = 0;
    n_pass = 0;
  }
  __syncthreads();
  const int n_items = 2 * C;  // (channel, upper / lower OP_HALF rows)
  const float* const sample = cms + (size_t)b * C * H * W;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int xb = k * 256;
    const int x = xb + 4 * lane;
    if (xb < W) {  // block-uniform
      const int xc = x < W ? x : 0;
      // rows y0 + OP_HALF (it & 1) .. + OP_HALF - 1 of channel it >> 1, from clamped addresses (a row / column outside the image repeats the nearest inside one: masked below)
      auto fetch = [&](int it, float (&q)[OP_HALF][4]) {
        const float* plane = sample + (size_t)(it >> 1) * H * W;
        const int r0 = y0 + (it & 1) * OP_HALF;
#pragma unroll
        for (int r = 0; r < OP_HALF; ++r) {
          const float* rp = plane + (size_t)min(r0 + r, H - 1) * W;
          if (VEC) {
            const float4 t = *reinterpret_cast<const float4*>(rp + xc);
            q[r][0] = t.x; q[r][1] = t.y; q[r][2] = t.z; q[r][3] = t.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) q[r][e] = rp[min(x + e, W - 1)];
          }
        }
      };
      auto test = [&](int it, const float (&q)[OP_HALF][4]) {
        bool hit = false;
#pragma unroll
        for (int r = 0; r < OP_HALF; ++r)
#pragma unroll
          for (int e = 0; e < 4; ++e) hit = hit | (q[r][e] > thr);
        if (PH_PEAKS_EXP & 64) {  // (timing: the loads stay alive, nothing is pushed)
          if (__ballot(hit) == 0x123456789ull) blk_count[0] = 1;
        } else if (__ballot(hit) != 0ull) {  // (wave-uniform; most windows of a confidence map hold no value above the threshold)
          // A value with a larger-or-equal neighbour among the ones this wave holds cannot be a peak: the 3x3 neighbourhood maximum of the window is formed from
          // registers (columns x - 1 / x + 4 from the neighbour lanes; what the wave does not hold -- the rows above / below the item's four, the columns beyond the
          // 256-column chunk -- and what is outside the image counts as -inf, and v_max skips a NaN: the filter only ever keeps too many, the listing below applies the
          // reference's exact test to whatever it keeps).  ~15 - 60 values per Gaussian blob are above the threshold, one or two survive: the list, the listing's
          // window loads and its latency shrink by that factor.  ~120 vector instructions, in the windows that hold a blob only, with the next item's loads in flight.
          // (Four in-lane compares per value and a branch per survivor instead: 23.4 us against 22.0.)
          const int rb = (it & 1) * OP_HALF, c = it >> 1;
          const float ninf = -__builtin_inff();
          const bool has_l = lane > 0, has_r = lane < 63 && x + 4 < W;
          float fp[4], fc[4], fn[4], hc[4], hn[4];
          auto hrow = [&](const float (&row)[4], bool in_rows, float (&f)[4], float (&h)[4]) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (in_rows & (x + e < W)) ? row[e] : ninf;
            float l = __shfl_up(v[3], 1, 64), rt = __shfl_down(v[0], 1, 64);
            l = has_l ? l : ninf;
            rt = has_r ? rt : ninf;
            h[0] = fmaxf(l, v[1]);
            h[1] = fmaxf(v[0], v[2]);
            h[2] = fmaxf(v[1], v[3]);
            h[3] = fmaxf(v[2], rt);
#pragma unroll
            for (int e = 0; e < 4; ++e) f[e] = fmaxf(h[e], v[e]);
          };
#pragma unroll
          for (int e = 0; e < 4; ++e) fp[e] = ninf;
          hrow(q[0], rb < rows, fc, hc);
          unsigned msk = 0u;
#pragma unroll
          for (int r = 0; r < OP_HALF; ++r) {
            if (r + 1 < OP_HALF) {
              hrow(q[r + 1], rb + r + 1 < rows, fn, hn);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) fn[e] = ninf;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float nb = fmaxf(fmaxf(hc[e], fp[e]), fn[e]);
              msk |= ((q[r][e] > fmaxf(thr, nb)) & (rb + r < rows) & (x + e < W)) ? (1u << (4 * r + e)) : 0u;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              fp[e] = fc[e];
              fc[e] = fn[e];
              hc[e] = hn[e];
            }
          }
          // the list is unordered: a lane with candidates reserves its places with its own LDS atomic (a lane or two per blob)
          int at = 0;
          if (msk) at = atomicAdd(&n_cand, __popc(msk));
          while (msk) {
            const int bit = __ffs((int)msk) - 1;
            msk &= msk - 1;
            if (at < OP_CAND) cand[at] = ((unsigned)(rb + (bit >> 2)) << 16) | ((unsigned)(x + (bit & 3)) << 6) | (unsigned)c;
            ++at;
          }
        }
      };
      // wave w takes items w, w + 4, ... (13 channels: 7 / 7 / 6 / 6 -- by whole channels it was 4 / 4 / 4 / 1) and has the NEXT item's loads in flight while it tests one
      float qa[OP_HALF][4], qb[OP_HALF][4];
      int it = wave;
      if (it < n_items) fetch(it, qa);
      for (; it < n_items; it += 8) {
        const bool more = it + 4 < n_items;
        if (more) fetch(it + 4, qb);
        test(it, qa);
        if (more) {
          if (it + 8 < n_items) fetch(it + 8, qa);
          test(it + 4, qb);
        }
      }
    }
  }
  __syncthreads();
  const int nc = n_cand;
  if ((PH_PEAKS_EXP & 1) || nc == 0) {  // a block without a candidate -- most of them -- is done here
    if (threadIdx.x == 0) blk_count[blk] = 0;
    return;
  }
  if (nc > OP_CAND) {
    // More candidates than the list holds (a map that is above the threshold nearly everywhere): count this block's peaks exactly -- every thread walks its column --
    // and leave the listing to the placement kernel's recompute path (flagged count).
    int mine = 0;
    for (int r = 0; r < rows; ++r)
      for (int xx = threadIdx.x; xx < W; xx += 256)
        for (int c = 0; c < C; ++c) mine += is_local_peak(cms + ((size_t)b * C + c) * H * W, H, W, y0 + r, xx, thr) ? 1 : 0;
    if (mine) atomicAdd(&n_pass, mine);
    __syncthreads();
    if (threadIdx.x == 0) blk_count[blk] = n_pass | OP_RECOMPUTE;
    return;
  }
  // (a short list -- the usual case -- is one wave's work: wave blk & 3 takes it, so that the four blocks of a CU list on four different SIMDs)
#pragma unroll 1
  for (int i = (threadIdx.x + 64 * (blk & 3)) & 255; i < nc; i += 256) {
    const unsigned key = cand[i];
    const int x = (key >> 6) & 0x3FF, y = y0 + (int)(key >> 16), c = key & 63;
    const float* plane = cms + ((size_t)b * C + c) * H * W;
    float val, dx = 0.f, dy = 0.f;
    bool pass;
    if (refine && patch == 5 && !(PH_PEAKS_EXP & 16)) {  // (the refinement of a candidate that turns out to be no peak is a few wasted loads)
      pass = test_and_refine<5>(plane, H, W, x, y, &val, &dx, &dy);
    } else if (refine && patch == 3 && !(PH_PEAKS_EXP & 16)) {
      pass = test_and_refine<3>(plane, H, W, x, y, &val, &dx, &dy);
    } else if (refine && patch == 7 && !(PH_PEAKS_EXP & 16)) {
      pass = test_and_refine<7>(plane, H, W, x, y, &val, &dx, &dy);
    } else {
      val = plane[(size_t)y * W + x];
      pass = is_local_peak_of_candidate(plane, H, W, y, x, val);
      if (refine && !(PH_PEAKS_EXP & 16)) integral_offset(plane, H, W, x, y, patch, &dx, &dy);
    }
    const float fx = (float)x + dx, fy = (float)y + dy;
    if (pass) {
      const int j = atomicAdd(&n_pass, 1);
      if (j < OP_STAGE) {
        pkey[j] = key;
        pres[j] = make_float4(fx * xy_scale, fy * xy_scale, val, 0.f);
      }
    }
  }
  __syncthreads();
  const int np = n_pass;
  if (np <= OP_STAGE) {  // (more: nothing is staged, the placement kernel recomputes the block from its count)
    const size_t sbase = (size_t)blk * OP_STAGE;
    for (int j = threadIdx.x; j < np; j += 256) {
      const unsigned key = pkey[j];
      int rank = 0;
      for (int t = 0; t < np; ++t) rank += pkey[t] < key ? 1 : 0;  // (keys are distinct: one per (pixel, channel))
      const float4 p = pres[j];
      stg_xy[2 * (sbase + rank)] = p.x;
      stg_xy[2 * (sbase + rank) + 1] = p.y;
      stg_val[sbase + rank] = p.z;
      stg_ch[sbase + rank] = (int)(key & 63);
    }
  }
  if (threadIdx.x == 0) blk_count[blk] = np;
}

__global__ __launch_bounds__(256) void peaks_place_kernel(const float* __restrict__ cms, int B, int C, int H, int W, float thr, int refine, int patch, float xy_scale, int groups,
                                                          const int* __restrict__ blk_count, const float* __restrict__ stg_xy, const float* __restrict__ stg_val,
                                                          const int* __restrict__ stg_ch, float* __restrict__ out_xy, float* __restrict__ out_val,
                                                          int* __restrict__ out_sample, int* __restrict__ out_channel, int* __restrict__ out_count, int cap) {
  __shared__ int red[8];
  __shared__ int s_off[2];
  const int blk = blockIdx.x;
  // exclusive offset of this block = sum of the counts of the blocks before it (integers: the order of the partial sums does not matter)
  int part = 0;
  for (int i = threadIdx.x; i < blk; i += 256) part += blk_count[i] & (OP_RECOMPUTE - 1);
  int tot;
  block_exclusive_scan_256(part, &tot, red);
  const int my_off = tot, my_cnt = blk_count[blk] & (OP_RECOMPUTE - 1);
  const bool staged = !(blk_count[blk] & OP_RECOMPUTE) && my_cnt <= OP_STAGE;
  const int b = blk / groups, y0 = (blk - b * groups) * OP_R;
  if (staged) {
    const size_t sbase = (size_t)blk * OP_STAGE;
    for (int i = threadIdx.x; i < my_cnt; i += 256) {
      const int o = my_off + i;
      if (o < cap) {
        out_xy[2 * (size_t)o] = stg_xy[2 * (sbase + i)];
        out_xy[2 * (size_t)o + 1] = stg_xy[2 * (sbase + i) + 1];
        out_val[o] = stg_val[sbase + i];
        out_sample[o] = b;
        out_channel[o] = stg_ch[sbase + i];
      }
    }
  } else {  // the staging area was too small for this block: recompute its rows into their final place (peaks_emit_kernel's logic)
    int base = my_off;
    for (int y = y0; y < min(y0 + OP_R, H); ++y)
      for (int xb = 0; xb < W; xb += 256) {
        const int x = xb + threadIdx.x;
        int cnt = 0;
        if (x < W)
          for (int c = 0; c < C; ++c) cnt += is_local_peak(cms + ((size_t)b * C + c) * H * W, H, W, y, x, thr) ? 1 : 0;
        int t2;
        int off = base + block_exclusive_scan_256(cnt, &t2, red);
        if (cnt > 0)
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
              out_xy[2 * (size_t)off] = fx * xy_scale;
              out_xy[2 * (size_t)off + 1] = fy * xy_scale;
              out_val[off] = plane[(size_t)y * W + x];
              out_sample[off] = b;
              out_channel[off] = c;
            }
            ++off;
          }
        base += t2;
      }
  }
  if (blk == 0) {  // totals: out_count[0] = n, [1 + b] = peaks of sample b, [1 + B + b] = exclusive offsets (B + 1 entries); thread = sample, one block scan per 256 samples
    __syncthreads();
    if (threadIdx.x == 0) s_off[0] = 0;
    __syncthreads();
    for (int sb0 = 0; sb0 < B; sb0 += 256) {
      const int bb = sb0 + threadIdx.x;
      int p = 0;
      if (bb < B)
        for (int i = 0; i < groups; ++i) p += blk_count[bb * groups + i] & (OP_RECOMPUTE - 1);
      int t3;
      const int ex = block_exclusive_scan_256(p, &t3, red);
      const int carry = s_off[0];
      if (bb < B) {
        out_count[1 + bb] = p;
        out_count[1 + B + bb] = carry + ex;
      }
      __syncthreads();
      if (threadIdx.x == 0) s_off[0] = carry + t3;
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      out_count[0] = s_off[0];
      out_count[1 + 2 * B] = s_off[0];
    }
  }
}

// scratch that lets ph_local_peaks take the one-pass path (0: the shape is outside it -- more than 64 channels or maps wider than 512 -- and the three-pass kernels run)
int64_t local_peaks_onepass_scratch_bytes(int B, int C, int H, int W) {
  if (C > 64 || W > 512) return 0;
  const int64_t n_blk = (int64_t)B * ((H + OP_R - 1) / OP_R);
  return 4 * (n_blk + 4 * n_blk * OP_STAGE);
}

int launch_local_peaks(const float* cms, int B, int C, int H, int W, float thr, int refine, int patch, float* out_xy, float* out_val,
                       int* out_sample, int* out_channel, int* out_count, int cap, float xy_scale, int* scratch, int64_t scratch_bytes, hipStream_t s) {
  const int64_t need = local_peaks_onepass_scratch_bytes(B, C, H, W);
  if (need > 0 && scratch_bytes >= need) {
    const int groups = (H + OP_R - 1) / OP_R, n_blk = B * groups;
    int* blk_count = scratch;
    float* stg_xy = reinterpret_cast<float*>(scratch + n_blk);
    float* stg_val = stg_xy + 2 * (size_t)n_blk * OP_STAGE;
    int* stg_ch = reinterpret_cast<int*>(stg_val + (size_t)n_blk * OP_STAGE);
#define PH_ONEPASS(NCH_, VEC_) \
  hipLaunchKernelGGL((peaks_onepass_kernel<NCH_, VEC_>), dim3(n_blk), dim3(256), 0, s, cms, C, H, W, thr, refine, patch, xy_scale, groups, blk_count, stg_xy, stg_val, stg_ch)
    const bool vec = (W & 3) == 0 && (reinterpret_cast<uintptr_t>(cms) & 15) == 0;
    if (W <= 256) {
      if (vec) PH_ONEPASS(1, true); else PH_ONEPASS(1, false);
    } else {
      if (vec) PH_ONEPASS(2, true); else PH_ONEPASS(2, false);
    }
#undef PH_ONEPASS
    hipLaunchKernelGGL(peaks_place_kernel, dim3(n_blk), dim3(256), 0, s, cms, B, C, H, W, thr, refine, patch, xy_scale, groups, blk_count, stg_xy, stg_val, stg_ch, out_xy, out_val,
                       out_sample, out_channel, out_count, cap);
    PH_HIP_CHECK(hipGetLastError());
    return PH_OK;
  }
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
// Round 4: ONE pass with 16-byte loads.  A thread keeps (max, first column holding it, first row holding it) over its elements -- a larger value replaces all three, an equal one
// lowers the column and the row independently -- and the (associative) combination of those triples over the block is the same "value = max, x = first column containing the max,
// y = first row containing the max" as two passes (peaks.py:89-181); NaN compares false both ways and is skipped, as fmaxf skipped it.  Four quads per thread are in flight per trip
// (the two-pass scalar loop streamed a 256-KiB plane at ~3 GB/s per block: 80 us for cfg2's 104 planes).
struct GPeak {
  float m;
  int x, y;
};
__device__ __forceinline__ void gpeak_take(GPeak& p, float v, int x, int y) {
  if (v > p.m) {
    p.m = v;
    p.x = x;
    p.y = y;
  } else if (v == p.m) {
    p.x = min(p.x, x);
    p.y = min(p.y, y);
  }
}
__device__ __forceinline__ void gpeak_merge(GPeak& p, float m, int x, int y) {
  if (m > p.m) {
    p.m = m;
    p.x = x;
    p.y = y;
  } else if (m == p.m) {
    p.x = min(p.x, x);
    p.y = min(p.y, y);
  }
}
__global__ __launch_bounds__(256) void global_peaks_kernel(const float* __restrict__ cms, int H, int W, float thr, int refine, int patch,
                                                           float* __restrict__ out_xy, float* __restrict__ out_val) {
  __shared__ float smax[4];
  __shared__ int sminx[4], sminy[4];
  const int pc = blockIdx.x;
  const float* plane = cms + (size_t)pc * H * W;
  const int n = H * W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  GPeak p{-INFINITY, 0x7fffffff, 0x7fffffff};
  if ((W & 3) == 0 && (reinterpret_cast<uintptr_t>(plane) & 15) == 0) {
    const int nq = n >> 2, wq = W >> 2;
    const float4* q4 = reinterpret_cast<const float4*>(plane);
    int i = threadIdx.x;
    for (; i + 3 * 256 < nq; i += 4 * 256) {
      float4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = q4[i + k * 256];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int q = i + k * 256, y = q / wq, x = 4 * (q - y * wq);
        gpeak_take(p, v[k].x, x, y);
        gpeak_take(p, v[k].y, x + 1, y);
        gpeak_take(p, v[k].z, x + 2, y);
        gpeak_take(p, v[k].w, x + 3, y);
      }
    }
    for (; i < nq; i += 256) {
      const float4 v = q4[i];
      const int y = i / wq, x = 4 * (i - y * wq);
      gpeak_take(p, v.x, x, y);
      gpeak_take(p, v.y, x + 1, y);
      gpeak_take(p, v.z, x + 2, y);
      gpeak_take(p, v.w, x + 3, y);
    }
  } else {
    for (int i = threadIdx.x; i < n; i += 256) {
      const int y = i / W;
      gpeak_take(p, plane[i], i - y * W, y);
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) gpeak_merge(p, __shfl_xor(p.m, d, 64), __shfl_xor(p.x, d, 64), __shfl_xor(p.y, d, 64));
  if (lane == 0) {
    smax[wave] = p.m;
    sminx[wave] = p.x;
    sminy[wave] = p.y;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    GPeak t{smax[0], sminx[0], sminy[0]};
#pragma unroll
    for (int w = 1; w < 4; ++w) gpeak_merge(t, smax[w], sminx[w], sminy[w]);
    const float m = t.m;
    const int mx = t.x, my = t.y;
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

// ---------------------------------------------------------------------------------------
// Top-down glue on the device (layers/centroid.py:195-261 CentroidLayer.postprocess: per-frame top-k + NaN pad + coordinate ladder;
// layers/topdown.py:183-260: valid centroids -> centred boxes -> crops -> crop offset + scatter back to (B, I, ...)).  The reference and the
// round-4 build walk the frames of a batch in Python with a host read per step; here one wave per frame selects, pads and lists, and one launch scatters.
// ---------------------------------------------------------------------------------------
// One wave per frame b.  Peaks of frame b are rows [off_b, off_b + n_b) of (xy, vals) in the reference's order (counts = ph_local_peaks' out_count).
// n_b <= I: kept in order; else the I largest values, descending (torch.topk; equal values: the earlier peak first).  Slots beyond are NaN.
// Lists for stage 2 (frames ascending, slots ascending = torch.nonzero order of the valid mask): sample index, bbox top-left, flat slot; pos_of_slot = inverse (-1: empty).
__global__ __launch_bounds__(64) void centroid_select_kernel(const float* __restrict__ xy, const float* __restrict__ vals, const int* __restrict__ counts, int B, int I, int cap,
                                                             float input_scale, const float* __restrict__ eff_scale, float half_w, float half_h,
                                                             float* __restrict__ out_cp, float* __restrict__ out_cv, float* __restrict__ out_bbox,
                                                             int* __restrict__ list_sample, float* __restrict__ list_tl, int* __restrict__ list_slot,
                                                             int* __restrict__ pos_of_slot, int* __restrict__ out_n_valid) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float NANF = __builtin_nanf("");
  const int total = min(counts[0], cap);
  int off = min(counts[1 + B + b], total);
  int n = min(counts[1 + b], total - off);
  int base = 0, all = 0;  // valid entries of the frames before this one / of all frames
  for (int j = 0; j < B; ++j) {
    const int oj = min(counts[1 + B + j], total);
    const int nj = min(min(counts[1 + j], total - oj), I);
    if (j < b) base += nj;
    all += nj;
  }
  if (b == 0 && lane == 0 && out_n_valid) *out_n_valid = all;
  const int keep = min(n, I);
  const float eff = eff_scale ? eff_scale[b] : 1.f;
  auto emit = [&](int slot, int src) {  // (one lane)
    float x = xy[2 * (size_t)(off + src)], y = xy[2 * (size_t)(off + src) + 1];
    if (input_scale != 1.f) {
      x = x / input_scale;
      y = y / input_scale;
    }
    x = x / eff;
    y = y / eff;
    const size_t s = (size_t)b * I + slot;
    out_cp[2 * s] = x;
    out_cp[2 * s + 1] = y;
    out_cv[s] = vals[off + src];
    if (out_bbox) {
      // TopDownLayer.predict (layers/topdown.py:127-150, 262-267): stage 2 works in SIZED space -- the centroid (original-image space above) times eff_scale is the box centre,
      // the crops are cut from the sizematched frame at that box, and boxes / keypoints are divided by eff_scale afterwards (ph_topdown_scatter).  make_centered_bboxes:
      // corners (x -/+ w/2, y -/+ h/2) + (+/-0.5): TL, TR, BR, BL
      const float sx = x * eff, sy = y * eff;
      float sb[8];
      sb[0] = (sx - half_w) + 0.5f;
      sb[1] = (sy - half_h) + 0.5f;
      sb[2] = (sx + half_w) - 0.5f;
      sb[3] = (sy - half_h) + 0.5f;
      sb[4] = (sx + half_w) - 0.5f;
      sb[5] = (sy + half_h) - 0.5f;
      sb[6] = (sx - half_w) + 0.5f;
      sb[7] = (sy + half_h) - 0.5f;
      float* bb = out_bbox + 8 * s;
      for (int e = 0; e < 8; ++e) bb[e] = sb[e] / eff;  // (debug output: image space)
      if (list_sample) {
        const int pos = base + slot;
        list_sample[pos] = b;
        list_tl[2 * pos] = sb[0];   // sized space: what ph_crop_bboxes cuts the sizematched frame with, and add_crop_offset's offset
        list_tl[2 * pos + 1] = sb[1];
        list_slot[pos] = (int)s;
      }
    }
    if (pos_of_slot) pos_of_slot[s] = base + slot;
  };
  if (n <= I) {
    for (int k = lane; k < keep; k += 64) emit(k, k);
  } else {
    // I rounds of a wave-wide arg-max over the peaks not taken yet (taken peaks are remembered as a bit per peak in registers: 64 x 64 = up to 4096 peaks per frame;
    // more than that and the tail beyond 4096 is not considered -- CentroidLayer's row capacity max(1024, 256 B) is far below unless one frame holds them all)
    unsigned long long taken = 0ull;  // bit j: peak lane + 64 j
    n = min(n, 4096);
    for (int k = 0; k < I; ++k) {
      float best = -__builtin_inff();
      int bi = 0x7FFFFFFF;
      for (int j = 0; lane + 64 * j < n; ++j) {
        if ((taken >> j) & 1ull) continue;
        const float v = vals[off + lane + 64 * j];
        if (v > best || (v == best && lane + 64 * j < bi)) {
          best = v;
          bi = lane + 64 * j;
        }
      }
      for (int m = 32; m >= 1; m >>= 1) {
        const float ov = __shfl_xor(best, m);
        const int oi = __shfl_xor(bi, m);
        if (ov > best || (ov == best && oi < bi)) {
          best = ov;
          bi = oi;
        }
      }
      if (bi != 0x7FFFFFFF && (bi & 63) == lane) {
        taken |= 1ull << (bi >> 6);
        emit(k, bi);
      }
    }
  }
  for (int k = keep + lane; k < I; k += 64) {
    const size_t s = (size_t)b * I + k;
    out_cp[2 * s] = NANF;
    out_cp[2 * s + 1] = NANF;
    out_cv[s] = NANF;
    if (out_bbox)
      for (int e = 0; e < 8; ++e) out_bbox[8 * s + e] = NANF;
    if (pos_of_slot) pos_of_slot[s] = -1;
  }
}

// (B I, N) threads: slot s of the padded outputs takes crop pos_of_slot[s]'s keypoints (+ its box's top-left: add_crop_offset) or NaN.
__global__ __launch_bounds__(256) void topdown_scatter_kernel(const float* __restrict__ crop_xy, const float* __restrict__ crop_vals, const float* __restrict__ list_tl,
                                                              const int* __restrict__ pos_of_slot, int slots, int N, const float* __restrict__ eff_scale, int I,
                                                              float* __restrict__ out_k, float* __restrict__ out_c, float* __restrict__ out_v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= slots * N) return;
  const int s = i / N, k = i - s * N;
  const int pos = pos_of_slot[s];
  const float NANF = __builtin_nanf("");
  float cx = NANF, cy = NANF, v = NANF, gx = NANF, gy = NANF;
  if (pos >= 0) {
    cx = crop_xy[2 * ((size_t)pos * N + k)];
    cy = crop_xy[2 * ((size_t)pos * N + k) + 1];
    v = crop_vals[(size_t)pos * N + k];
    const float eff = eff_scale ? eff_scale[s / I] : 1.f;  // sized space -> image space (topdown.py:262-264)
    gx = (cx + list_tl[2 * pos]) / eff;
    gy = (cy + list_tl[2 * pos + 1]) / eff;
  }
  out_k[2 * (size_t)i] = gx;
  out_k[2 * (size_t)i + 1] = gy;
  out_c[2 * (size_t)i] = cx;
  out_c[2 * (size_t)i + 1] = cy;
  out_v[i] = v;
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
                            static_cast<int*>(scratch_dev), scratch_bytes, static_cast<hipStream_t>(stream));
}

int64_t ph_local_peaks_scratch_bytes(int32_t B, int32_t C, int32_t H, int32_t W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  return std::max<int64_t>((int64_t)4 * (2 * (int64_t)B * H + 2), local_peaks_onepass_scratch_bytes(B, C, H, W));
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

int ph_centroid_select(const float* peaks_xy_dev, const float* peak_vals_dev, const int32_t* counts_dev, int32_t B, int32_t max_instances, int32_t cap,
                       float input_scale, const float* eff_scale_dev, float crop_h, float crop_w, float* out_centroids_dev, float* out_vals_dev,
                       float* out_bboxes_dev, int32_t* list_sample_dev, float* list_topleft_dev, int32_t* list_slot_dev, int32_t* pos_of_slot_dev,
                       int32_t* out_n_valid_dev, void* stream) {
  PH_REQUIRE(peaks_xy_dev && peak_vals_dev && counts_dev && out_centroids_dev && out_vals_dev, "ph_centroid_select: null argument");
  PH_REQUIRE(B > 0 && max_instances > 0 && cap >= 0, "ph_centroid_select: bad shape");
  PH_REQUIRE(!list_sample_dev || (out_bboxes_dev && list_topleft_dev && list_slot_dev), "ph_centroid_select: the stage-2 lists need the boxes");
  hipLaunchKernelGGL(centroid_select_kernel, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), peaks_xy_dev, peak_vals_dev, counts_dev, B, max_instances, cap,
                     input_scale, eff_scale_dev, crop_w / 2, crop_h / 2, out_centroids_dev, out_vals_dev, out_bboxes_dev, list_sample_dev, list_topleft_dev, list_slot_dev,
                     pos_of_slot_dev, out_n_valid_dev);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

int ph_topdown_scatter(const float* crop_xy_dev, const float* crop_vals_dev, const float* list_topleft_dev, const int32_t* pos_of_slot_dev, int32_t slots,
                       int32_t n_nodes, const float* eff_scale_dev, int32_t max_instances, float* out_keypoints_dev, float* out_crop_keypoints_dev, float* out_vals_dev,
                       void* stream) {
  PH_REQUIRE(crop_xy_dev && crop_vals_dev && list_topleft_dev && pos_of_slot_dev && out_keypoints_dev && out_crop_keypoints_dev && out_vals_dev, "ph_topdown_scatter: null argument");
  PH_REQUIRE(slots > 0 && n_nodes > 0 && (!eff_scale_dev || (max_instances > 0 && slots % max_instances == 0)), "ph_topdown_scatter: bad shape");
  hipLaunchKernelGGL(topdown_scatter_kernel, dim3((slots * n_nodes + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), crop_xy_dev, crop_vals_dev, list_topleft_dev,
                     pos_of_slot_dev, slots, n_nodes, eff_scale_dev, max_instances > 0 ? max_instances : 1, out_keypoints_dev, out_crop_keypoints_dev, out_vals_dev);
  PH_HIP_CHECK(hipGetLastError());
  return PH_OK;
}

}  // extern "C"
