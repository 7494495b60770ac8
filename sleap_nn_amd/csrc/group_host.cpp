// Host (CPU) stage of bottom-up grouping: rectangular assignment + greedy instance assembly.
// Pure C++, no HIP.  Replaces, for a whole batch in one call, the per-sample / per-edge Python
// loops of the reference:
//   match_candidates_sample ........... sleap_nn/inference/ops/paf.py:500-619
//   scipy.optimize.linear_sum_assignment (called at paf.py:589; SciPy's rectangular_lsap is
//       the shortest-augmenting-path algorithm of D. F. Crouse, "On implementing 2D
//       rectangular assignment algorithms", IEEE TAES 52(4), 2016 -- restated here from the
//       paper's pseudocode, with SciPy's tie rules: candidate columns scanned from the
//       highest index down, ties on the shortest path broken towards unassigned columns)
//   assign_connections_to_instances ... paf.py:705-820
//   make_predicted_instances .......... paf.py:823-887
//   toposort_edges .................... paf.py:890-912
//   group_instances_sample/_batch ..... paf.py:915-1149
//   NaN-padding / top-N truncation .... sleap_nn/inference/streaming.py:196-241
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>

#include "../../include/posehip.h"

namespace ph {
void set_error(const char* fmt, ...);
}

namespace {

const double INF = std::numeric_limits<double>::infinity();

// Shortest augmenting path from row i; returns the sink column or -1 if none is reachable.
int augment(int nc, const double* cost, const std::vector<double>& u, const std::vector<double>& v, std::vector<int>& path,
            const std::vector<int>& row4col, std::vector<double>& dist, int i, std::vector<char>& SR, std::vector<char>& SC,
            std::vector<int>& remaining, double* p_min) {
  double min_val = 0;
  int num_remaining = nc;
  for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;  // reverse fill: constant matrices solve to identity
  std::fill(SR.begin(), SR.end(), 0);
  std::fill(SC.begin(), SC.end(), 0);
  std::fill(dist.begin(), dist.end(), INF);
  int sink = -1;
  while (sink == -1) {
    int index = -1;
    double lowest = INF;
    SR[i] = 1;
    for (int it = 0; it < num_remaining; ++it) {
      const int j = remaining[it];
      const double r = min_val + cost[(size_t)i * nc + j] - u[i] - v[j];
      if (r < dist[j]) {
        path[j] = i;
        dist[j] = r;
      }
      if (dist[j] < lowest || (dist[j] == lowest && row4col[j] == -1)) {
        lowest = dist[j];
        index = it;
      }
    }
    min_val = lowest;
    if (min_val == INF) return -1;
    const int j = remaining[index];
    if (row4col[j] == -1)
      sink = j;
    else
      i = row4col[j];
    SC[j] = 1;
    remaining[index] = remaining[--num_remaining];
  }
  *p_min = min_val;
  return sink;
}

int lsap(const double* cost_in, int nr, int nc, int* rows, int* cols) {
  if (nr == 0 || nc == 0) return 0;
  const bool transpose = nc < nr;
  std::vector<double> tmp;
  const double* cost = cost_in;
  if (transpose) {
    tmp.resize((size_t)nr * nc);
    for (int i = 0; i < nr; ++i)
      for (int j = 0; j < nc; ++j) tmp[(size_t)j * nr + i] = cost_in[(size_t)i * nc + j];
    std::swap(nr, nc);
    cost = tmp.data();
  }
  for (size_t k = 0; k < (size_t)nr * nc; ++k)
    if (cost[k] != cost[k] || cost[k] == -INF) return PH_E_INVALID;
  std::vector<double> u(nr, 0), v(nc, 0), dist(nc);
  std::vector<int> path(nc, -1), col4row(nr, -1), row4col(nc, -1), remaining(nc);
  std::vector<char> SR(nr), SC(nc);
  for (int cur = 0; cur < nr; ++cur) {
    double min_val;
    const int sink = augment(nc, cost, u, v, path, row4col, dist, cur, SR, SC, remaining, &min_val);
    if (sink < 0) return PH_E_INFEASIBLE;
    u[cur] += min_val;
    for (int i = 0; i < nr; ++i)
      if (SR[i] && i != cur) u[i] += min_val - dist[col4row[i]];
    for (int j = 0; j < nc; ++j)
      if (SC[j]) v[j] -= min_val - dist[j];
    int j = sink;
    while (true) {
      const int i = path[j];
      row4col[j] = i;
      std::swap(col4row[i], j);
      if (i == cur) break;
    }
  }
  if (transpose) {
    std::vector<int> order(nr);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return col4row[a] < col4row[b]; });
    for (int k = 0; k < nr; ++k) {
      rows[k] = col4row[order[k]];
      cols[k] = order[k];
    }
  } else {
    for (int i = 0; i < nr; ++i) {
      rows[i] = i;
      cols[i] = col4row[i];
    }
  }
  return nr;
}

// BFS edge order from the first zero-in-degree node (insertion order), neighbours in
// insertion order; an edge to an already-visited node is skipped (tree edges only).
int toposort(const int32_t* edges, int n_edges, int32_t* out) {
  std::vector<int> nodes;  // insertion order
  auto idx_of = [&](int n) {
    for (size_t i = 0; i < nodes.size(); ++i)
      if (nodes[i] == n) return (int)i;
    nodes.push_back(n);
    return (int)nodes.size() - 1;
  };
  std::vector<std::vector<int>> adj;
  std::vector<int> indeg;
  for (int e = 0; e < n_edges; ++e) {
    const int s = idx_of(edges[2 * e]);
    const int d = idx_of(edges[2 * e + 1]);
    adj.resize(nodes.size());
    indeg.resize(nodes.size(), 0);
    bool dup = false;
    for (int x : adj[s]) dup = dup || (x == d);
    if (!dup) {
      adj[s].push_back(d);
      indeg[d] += 1;
    }
  }
  if (nodes.empty()) return 0;
  int root = -1;
  for (size_t i = 0; i < nodes.size(); ++i)
    if (indeg[i] == 0) {
      root = (int)i;
      break;
    }
  if (root < 0) return PH_E_INVALID;
  std::vector<char> seen(nodes.size(), 0);
  std::vector<int> q{root};
  seen[root] = 1;
  int n_out = 0;
  for (size_t h = 0; h < q.size(); ++h) {
    const int uu = q[h];
    for (int vv : adj[uu]) {
      if (seen[vv]) continue;
      seen[vv] = 1;
      for (int e = 0; e < n_edges; ++e)
        if (edges[2 * e] == nodes[uu] && edges[2 * e + 1] == nodes[vv]) {
          out[n_out++] = e;
          break;
        }
      q.push_back(vv);
    }
  }
  return n_out;
}

struct Conn {
  int src, dst;  // node-local peak indices
  float score;
};

}  // namespace

extern "C" {

int ph_lsap(const double* cost, int32_t nr, int32_t nc, int32_t* rows, int32_t* cols) {
  if (nr < 0 || nc < 0 || (nr > 0 && nc > 0 && (!cost || !rows || !cols))) {
    ph::set_error("ph_lsap: bad arguments");
    return PH_E_INVALID;
  }
  const int rc = lsap(cost, nr, nc, rows, cols);
  if (rc == PH_E_INFEASIBLE) ph::set_error("cost matrix is infeasible");
  if (rc == PH_E_INVALID) ph::set_error("matrix contains invalid numeric entries");
  return rc < 0 ? rc : PH_OK;
}

int ph_toposort_edges(const int32_t* edges, int32_t n_edges, int32_t* out_order) {
  if (n_edges < 0 || (n_edges > 0 && (!edges || !out_order))) {
    ph::set_error("ph_toposort_edges: bad arguments");
    return PH_E_INVALID;
  }
  const int rc = toposort(edges, n_edges, out_order);
  if (rc < 0) ph::set_error("skeleton graph has no root (cycle)");
  return rc;
}

int ph_group_class_peaks(const float* probs, const int32_t* sample_inds, const int32_t* channel_inds, int32_t n, int32_t n_samples,
                         int32_t n_channels, int32_t K, int32_t* out_peak_inds, int32_t* out_class_inds) {
  if (n < 0 || n_samples <= 0 || n_channels <= 0 || K <= 0 || (n > 0 && (!probs || !sample_inds || !channel_inds || !out_peak_inds || !out_class_inds))) {
    ph::set_error("ph_group_class_peaks: bad arguments");
    return PH_E_INVALID;
  }
  int n_out = 0;
  std::vector<int> members;
  std::vector<double> cost;
  std::vector<int> rows, cols;
  for (int smp = 0; smp < n_samples; ++smp)
    for (int chn = 0; chn < n_channels; ++chn) {
      members.clear();
      for (int i = 0; i < n; ++i)
        if (sample_inds[i] == smp && channel_inds[i] == chn) members.push_back(i);
      if (members.empty()) continue;
      const int nr = (int)members.size();
      cost.resize((size_t)nr * K);
      for (int r = 0; r < nr; ++r)
        for (int k = 0; k < K; ++k) cost[(size_t)r * K + k] = -(double)probs[(size_t)members[r] * K + k];
      const int nm = std::min(nr, (int)K);
      rows.resize(nm);
      cols.resize(nm);
      const int rc = lsap(cost.data(), nr, K, rows.data(), cols.data());
      if (rc < 0) {
        ph::set_error(rc == PH_E_INFEASIBLE ? "cost matrix is infeasible" : "matrix contains invalid numeric entries");
        return rc;
      }
      for (int k = 0; k < nm; ++k) {
        const int pi = members[rows[k]];
        const float matched = probs[(size_t)pi * K + cols[k]];
        float best = probs[(size_t)pi * K];
        for (int c = 1; c < K; ++c) best = std::max(best, probs[(size_t)pi * K + c]);
        if (matched == best) {  // keep only matches that are the peak's arg-max class (identity.py:61-66)
          out_peak_inds[n_out] = pi;
          out_class_inds[n_out] = cols[k];
          ++n_out;
        }
      }
    }
  return n_out;
}

int ph_group_batch(int32_t B, int32_t n_nodes, const int32_t* edges, int32_t n_edges, const float* peaks_xy, const float* peak_vals,
                   const int32_t* peak_channel, const int32_t* peak_offsets, const int32_t* cand_edge, const int32_t* cand_src,
                   const int32_t* cand_dst, const float* cand_score, const int32_t* cand_offsets, float min_line_score,
                   double min_instance_peaks, int32_t min_instance_peaks_is_fraction, int32_t max_inst, int32_t truncate_by_score,
                   float* out_kpts, float* out_vals, float* out_scores, int32_t* out_n_inst) {
  if (B <= 0 || n_nodes <= 0 || n_edges < 0 || max_inst <= 0 || !peak_offsets || !cand_offsets || !out_kpts || !out_vals || !out_scores ||
      !out_n_inst) {
    ph::set_error("ph_group_batch: bad arguments");
    return PH_E_INVALID;
  }
  const float NANF = std::numeric_limits<float>::quiet_NaN();
  std::fill(out_kpts, out_kpts + (size_t)B * max_inst * n_nodes * 2, NANF);
  std::fill(out_vals, out_vals + (size_t)B * max_inst * n_nodes, NANF);
  std::fill(out_scores, out_scores + (size_t)B * max_inst, NANF);

  std::vector<int32_t> order(n_edges);
  const int n_sorted = n_edges > 0 ? toposort(edges, n_edges, order.data()) : 0;
  if (n_sorted < 0) {
    ph::set_error("skeleton graph has no root (cycle)");
    return PH_E_INVALID;
  }

  std::vector<std::vector<int>> node_peaks(n_nodes);     // node -> peak indices (sample-local), ascending
  std::vector<int> node_rank;                            // peak -> rank inside its node
  std::vector<std::vector<Conn>> conns(n_edges);
  std::vector<double> cost;
  std::vector<int> rows, cols;
  std::vector<std::vector<int>> assign(n_nodes);         // node -> per node-local peak: instance id or -1

  for (int b = 0; b < B; ++b) {
    const int p0 = peak_offsets[b], np = peak_offsets[b + 1] - p0;
    const int c0 = cand_offsets[b], ncand = cand_offsets[b + 1] - c0;
    for (auto& v : node_peaks) v.clear();
    node_rank.assign(np, -1);
    for (int i = 0; i < np; ++i) {
      const int k = peak_channel[p0 + i];
      if (k < 0 || k >= n_nodes) continue;
      node_rank[i] = (int)node_peaks[k].size();
      node_peaks[k].push_back(i);
    }
    // ---- matching per edge (paf.py:558-611).  The cost matrix rows/cols are the sorted unique
    // src/dst peak indices that occur among the edge's candidates.
    for (auto& c : conns) c.clear();
    std::vector<int> cand_of_edge_start(n_edges + 1, 0);
    for (int q = 0; q < ncand; ++q) {
      const int e = cand_edge[c0 + q];
      if (e < 0 || e >= n_edges) {
        ph::set_error("candidate edge index out of range");
        return PH_E_INVALID;
      }
      cand_of_edge_start[e + 1] += 1;
    }
    for (int e = 0; e < n_edges; ++e) cand_of_edge_start[e + 1] += cand_of_edge_start[e];
    std::vector<int> cand_sorted(ncand);
    {
      std::vector<int> fill(cand_of_edge_start.begin(), cand_of_edge_start.end() - 1);
      for (int q = 0; q < ncand; ++q) cand_sorted[fill[cand_edge[c0 + q]]++] = q;
    }
    for (int e = 0; e < n_edges; ++e) {
      const int qs = cand_of_edge_start[e], qe = cand_of_edge_start[e + 1];
      if (qe == qs) continue;
      std::vector<int> su, du;
      for (int t = qs; t < qe; ++t) {
        su.push_back(cand_src[c0 + cand_sorted[t]]);
        du.push_back(cand_dst[c0 + cand_sorted[t]]);
      }
      std::sort(su.begin(), su.end());
      su.erase(std::unique(su.begin(), su.end()), su.end());
      std::sort(du.begin(), du.end());
      du.erase(std::unique(du.begin(), du.end()), du.end());
      const int nr = (int)su.size(), nc = (int)du.size();
      cost.assign((size_t)nr * nc, INF);
      for (int t = qs; t < qe; ++t) {
        const int q = cand_sorted[t];
        const int r = (int)(std::lower_bound(su.begin(), su.end(), cand_src[c0 + q]) - su.begin());
        const int c = (int)(std::lower_bound(du.begin(), du.end(), cand_dst[c0 + q]) - du.begin());
        const float sc = cand_score[c0 + q];
        cost[(size_t)r * nc + c] = (sc != sc) ? INF : -(double)sc;  // NaN -> +inf (paf.py:586)
      }
      const int nm = std::min(nr, nc);
      rows.resize(nm);
      cols.resize(nm);
      const int rc = lsap(cost.data(), nr, nc, rows.data(), cols.data());
      if (rc < 0) {
        ph::set_error(rc == PH_E_INFEASIBLE ? "cost matrix is infeasible" : "matrix contains invalid numeric entries");
        return rc;
      }
      // matched (row, col) index the edge-grouped peaks == node-local peak ranks
      for (int k = 0; k < nm; ++k) {
        const float sc = (float)(-cost[(size_t)rows[k] * nc + cols[k]]);
        if (sc >= min_line_score) conns[e].push_back({rows[k], cols[k], sc});
      }
    }
    // ---- greedy assembly in toposorted edge order (paf.py:743-789)
    for (int k = 0; k < n_nodes; ++k) assign[k].assign(node_peaks[k].size(), -1);
    int max_id = -1;
    auto recompute_max = [&]() {
      max_id = -1;
      for (auto& a : assign)
        for (int v : a) max_id = std::max(max_id, v);
    };
    for (int oi = 0; oi < n_sorted; ++oi) {
      const int e = order[oi];
      const int sn = edges[2 * e], dn = edges[2 * e + 1];
      for (const Conn& c : conns[e]) {
        if (c.src >= (int)assign[sn].size() || c.dst >= (int)assign[dn].size()) continue;
        const int ia = assign[sn][c.src], ib = assign[dn][c.dst];
        if (ia < 0 && ib < 0) {
          recompute_max();
          assign[sn][c.src] = max_id + 1;
          assign[dn][c.dst] = max_id + 1;
        } else if (ia >= 0 && ib < 0) {
          assign[dn][c.dst] = ia;
        } else if (ia >= 0 && ib >= 0) {
          assign[dn][c.dst] = ia;
          bool overlap = false;
          for (int k = 0; k < n_nodes && !overlap; ++k) {
            bool ha = false, hb = false;
            for (int v : assign[k]) {
              ha = ha || v == ia;
              hb = hb || v == ib;
            }
            overlap = ha && hb;
          }
          if (!overlap)
            for (auto& a : assign)
              for (int& v : a)
                if (v == ib) v = ia;
        }
        // src unassigned, dst assigned: nothing happens (reference behaviour)
      }
    }
    // ---- min_instance_peaks filter (paf.py:791-818)
    if (min_instance_peaks > 0) {
      const int mip = min_instance_peaks_is_fraction ? (int)(min_instance_peaks * n_nodes) : (int)min_instance_peaks;
      recompute_max();
      std::vector<int> counts(max_id + 1, 0);
      for (auto& a : assign)
        for (int v : a)
          if (v >= 0) counts[v] += 1;
      for (auto& a : assign)
        for (int& v : a)
          if (v >= 0 && counts[v] < mip) v = -1;
    }
    // ---- contiguous renumbering by ascending id (np.unique, paf.py:845-850)
    recompute_max();
    std::vector<int> remap(max_id + 1, -1);
    for (auto& a : assign)
      for (int v : a)
        if (v >= 0) remap[v] = 0;
    int n_inst = 0;
    for (int& r : remap)
      if (r == 0) r = n_inst++;
    out_n_inst[b] = n_inst;
    // ---- instance scores: sum of edge scores in toposorted order, fp32 (paf.py:853-865)
    std::vector<float> iscore(n_inst, 0.f);
    for (int oi = 0; oi < n_sorted; ++oi) {
      const int e = order[oi];
      const int sn = edges[2 * e];
      for (const Conn& c : conns[e]) {
        if (c.src >= (int)assign[sn].size()) continue;
        const int v = assign[sn][c.src];
        if (v >= 0) iscore[remap[v]] += c.score;
      }
    }
    // ---- which instances are kept, in which output order (streaming.py:221-241)
    std::vector<int> keep(n_inst);
    std::iota(keep.begin(), keep.end(), 0);
    if (n_inst > max_inst && truncate_by_score) {
      // numpy argsort (quicksort kind, but stable for these sizes is not guaranteed): emulate
      // np.argsort(scores)[::-1] with a stable ascending sort, NaN last, then reverse.
      std::stable_sort(keep.begin(), keep.end(), [&](int a, int c) {
        const float x = iscore[a], y = iscore[c];
        if (x != x) return false;
        if (y != y) return true;
        return x < y;
      });
      std::reverse(keep.begin(), keep.end());
    }
    const int n_keep = std::min(n_inst, max_inst);
    std::vector<int> slot_of(n_inst, -1);
    for (int k = 0; k < n_keep; ++k) slot_of[keep[k]] = k;
    for (int k = 0; k < n_nodes; ++k)
      for (size_t r = 0; r < assign[k].size(); ++r) {
        const int v = assign[k][r];
        if (v < 0) continue;
        const int slot = slot_of[remap[v]];
        if (slot < 0) continue;
        const int pi = p0 + node_peaks[k][r];
        float* kp = out_kpts + (((size_t)b * max_inst + slot) * n_nodes + k) * 2;
        kp[0] = peaks_xy[2 * (size_t)pi];
        kp[1] = peaks_xy[2 * (size_t)pi + 1];
        out_vals[((size_t)b * max_inst + slot) * n_nodes + k] = peak_vals[pi];
      }
    for (int k = 0; k < n_keep; ++k) out_scores[(size_t)b * max_inst + k] = iscore[keep[k]];
  }
  return PH_OK;
}

// One call for the CPU stage of a bottom-up batch, straight from the packed D2H arena BottomUpLayer's GPU stage fills
// (layers/bottomup.py _enqueue_scoring: [counts 2+2B | cand offsets B+1 | xy 2P | vals P | score Q | channel P | cand edge Q | src Q | dst Q],
// int32 rows bit-cast): what _finish_scoring (unpack, capacity check, max_peaks_per_node guard: sleap_nn layers/bottomup.py:126-161) and
// group_scored_batch (sleap_nn inference/streaming.py:147-255: matching + assembly + input-scale / eff-scale undo + NaN pad) do, without a
// numpy copy or a Python loop in between.  status[0] = peaks, [1] = candidates, [2] = flags (1: the arena's capacities were exceeded -- nothing
// was grouped, re-run the GPU stage with larger ones; 2: out_cap is smaller than the instance bound -- nothing was grouped, status[3] = the
// capacity to come back with; 4: the max_peaks_per_node guard fired, outputs are all NaN), [3] = instances to keep per sample
// (max over samples, at least 1; max_instances when that is given).  max_instances < 0 = None.
int ph_group_packed(const float* arena, int32_t B, int32_t n_nodes, int32_t peak_cap, int32_t cand_cap, const int32_t* edges, int32_t n_edges,
                    float min_line_score, double min_instance_peaks, int32_t min_instance_peaks_is_fraction, int32_t max_instances,
                    int32_t max_peaks_per_node, float input_scale, const float* eff_scale, int32_t out_cap, float* out_kpts, float* out_vals,
                    float* out_scores, int32_t* out_n_inst, int32_t* status) {
  if (!arena || B <= 0 || n_nodes <= 0 || peak_cap <= 0 || cand_cap <= 0 || out_cap <= 0 || !out_kpts || !out_vals || !out_scores || !out_n_inst || !status) {
    ph::set_error("ph_group_packed: bad arguments");
    return PH_E_INVALID;
  }
  const int32_t* ihead = reinterpret_cast<const int32_t*>(arena);
  const int n_head = (2 + 2 * B) + (B + 1);
  const int32_t n_peaks = ihead[0];
  const int32_t* peak_offsets = ihead + 1 + B;
  const int32_t* cand_offsets = ihead + 2 + 2 * B;
  const int32_t n_cand = cand_offsets[B];
  status[0] = n_peaks;
  status[1] = n_cand;
  status[2] = 0;
  status[3] = max_instances >= 0 ? std::max(1, max_instances) : 1;
  if (n_peaks < 0 || n_cand < 0 || n_peaks > peak_cap || n_cand > cand_cap) {
    status[2] = 1;
    return PH_OK;
  }
  const float* xy = arena + n_head;
  const float* vals = xy + 2 * (size_t)peak_cap;
  const float* score = vals + peak_cap;
  const int32_t* channel = reinterpret_cast<const int32_t*>(score + cand_cap);
  const int32_t* cand_edge = channel + peak_cap;
  const int32_t* cand_src = cand_edge + cand_cap;
  const int32_t* cand_dst = cand_src + cand_cap;
  const float NANF = std::numeric_limits<float>::quiet_NaN();
  bool skip = false;
  int bound = 0;
  std::vector<int> per_node(n_nodes);
  for (int b = 0; b < B; ++b) {
    const int p0 = peak_offsets[b], np = peak_offsets[b + 1] - p0;
    bound = std::max(bound, np);
    if (max_peaks_per_node >= 0 && !skip) {
      std::fill(per_node.begin(), per_node.end(), 0);
      for (int i = 0; i < np; ++i) {
        const int k = channel[p0 + i];
        if (k >= 0 && k < n_nodes && ++per_node[k] > max_peaks_per_node) skip = true;
      }
    }
  }
  if (skip) {  // combinatorial-blow-up guard: a batch of NaNs, (B, max_instances or 1, ...)
    status[2] = 4;
    std::fill(out_kpts, out_kpts + (size_t)B * out_cap * n_nodes * 2, NANF);
    std::fill(out_vals, out_vals + (size_t)B * out_cap * n_nodes, NANF);
    std::fill(out_scores, out_scores + (size_t)B * out_cap, NANF);
    std::fill(out_n_inst, out_n_inst + B, 0);
    return PH_OK;
  }
  // an instance needs at least one matched edge, so the per-sample instance count is bounded by the number of peaks (group_scored_batch)
  const int cap = max_instances >= 0 ? std::max(1, max_instances) : std::max(1, bound);
  if (cap > out_cap) {
    status[2] = 2;
    status[3] = cap;
    return PH_OK;
  }
  // (grouping with max_inst = out_cap >= cap: with max_instances None nothing is truncated, the layout stride is the caller's)
  const int rc = ph_group_batch(B, n_nodes, edges, n_edges, xy, vals, channel, peak_offsets, cand_edge, cand_src, cand_dst, score, cand_offsets, min_line_score,
                                min_instance_peaks, min_instance_peaks_is_fraction, max_instances >= 0 ? cap : out_cap, max_instances >= 0 ? 1 : 0, out_kpts, out_vals,
                                out_scores, out_n_inst);
  if (rc != PH_OK) return rc;
  const int stride = max_instances >= 0 ? cap : out_cap;
  if (max_instances < 0) {
    int mi = 1;
    for (int b = 0; b < B; ++b) mi = std::max(mi, (int)out_n_inst[b]);
    status[3] = mi;
  }
  const bool by_input = input_scale != 1.0f;
  bool by_eff = false;
  if (eff_scale)
    for (int b = 0; b < B; ++b) by_eff = by_eff || eff_scale[b] != 1.0f;
  if (by_input || by_eff)
    for (int b = 0; b < B; ++b) {
      float* kp = out_kpts + (size_t)b * stride * n_nodes * 2;
      for (size_t i = 0; i < (size_t)stride * n_nodes * 2; ++i) {
        float v = kp[i];
        if (by_input) v = v / input_scale;
        if (by_eff) v = v / eff_scale[b];
        kp[i] = v;
      }
    }
  return PH_OK;
}

}  // extern "C"
