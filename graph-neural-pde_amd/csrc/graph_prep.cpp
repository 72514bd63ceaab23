// Host-side graph preparation: COO (any order, duplicates allowed) -> CSR + stable permutation,
// CSC view, long-row chunk list; balanced k-way row partition for the multi-GPU path.
//
// Replaces the implicit COO handling inside torch_sparse.spmm / torch_scatter that the reference
// relies on (reference src/function_transformer_attention.py:35,190-191,213;
// src/function_laplacian_diffusion.py:31-35).  The stable counting sort keeps duplicates and the
// within-row edge order of the caller, so `perm` maps every CSR position back to exactly one edge
// of the reference's edge list (attention is returned in that order).
#include <algorithm>
#include <cstring>
#include <numeric>
#include <queue>
#include <vector>
#include "common.h"

using namespace gnpde;

extern "C" int gnpde_graph_count_long(const int64_t* row, const int64_t* col, int64_t n_edges, int32_t n_nodes,
                                      int32_t* n_long_rows, int32_t* n_long_chunks, int32_t* n_long_cols) {
  GNPDE_CHECK_ARG(n_nodes >= 0 && n_edges >= 0 && n_edges < (int64_t(1) << 31), GNPDE_EINVAL,
                  "graph_count_long: bad sizes n=%d e=%lld", n_nodes, (long long)n_edges);
  GNPDE_CHECK_ARG((row || n_edges == 0) && n_long_rows && n_long_chunks && n_long_cols, GNPDE_EINVAL,
                  "graph_count_long: null pointer");
  std::vector<int32_t> deg(static_cast<size_t>(n_nodes), 0), cdeg(col ? static_cast<size_t>(n_nodes) : 0, 0);
  for (int64_t e = 0; e < n_edges; ++e) {
    const int64_t r = row[e];
    GNPDE_CHECK_ARG(r >= 0 && r < n_nodes, GNPDE_EINVAL, "graph_count_long: row index %lld out of range",
                    (long long)r);
    ++deg[r];
    if (col) {
      const int64_t c = col[e];
      GNPDE_CHECK_ARG(c >= 0 && c < n_nodes, GNPDE_EINVAL, "graph_count_long: column index %lld out of range",
                      (long long)c);
      ++cdeg[c];
    }
  }
  int32_t lr = 0, lc = 0, lcol = 0;
  for (int32_t i = 0; i < n_nodes; ++i) {
    if (deg[i] > GNPDE_LONG_ROW) {
      ++lr;
      lc += (deg[i] + GNPDE_LONG_ROW - 1) / GNPDE_LONG_ROW;
    }
    if (col && cdeg[i] > GNPDE_LONG_ROW) ++lcol;
  }
  *n_long_rows = lr;
  *n_long_chunks = lc;
  *n_long_cols = lcol;
  return 0;
}

extern "C" int gnpde_graph_build(const int64_t* row, const int64_t* col, int64_t n_edges, int32_t n_nodes,
                                 int32_t* rowptr, int32_t* colidx, int32_t* perm, int32_t* rowidx,
                                 int32_t* cscptr, int32_t* cscpos, int32_t* long_rows,
                                 int32_t* long_chunk_ptr, int32_t* long_chunk_row,
                                 int32_t* long_chunk_begin, int32_t* long_chunk_end, int32_t* long_cols,
                                 int32_t* bin_rows, int32_t* bin_counts, int32_t* long_chunk_first) {
  GNPDE_CHECK_ARG(n_nodes >= 0 && n_edges >= 0 && n_edges < (int64_t(1) << 31), GNPDE_EINVAL,
                  "graph_build: bad sizes n=%d e=%lld", n_nodes, (long long)n_edges);
  GNPDE_CHECK_ARG(rowptr && (n_edges == 0 || (row && col && colidx && perm && rowidx)), GNPDE_EINVAL,
                  "graph_build: null pointer");
  const size_t n = static_cast<size_t>(n_nodes);
  std::fill(rowptr, rowptr + n + 1, 0);
  for (int64_t e = 0; e < n_edges; ++e) {
    const int64_t r = row[e], c = col[e];
    GNPDE_CHECK_ARG(r >= 0 && r < n_nodes && c >= 0 && c < n_nodes, GNPDE_EINVAL,
                    "graph_build: edge %lld = (%lld,%lld) outside [0,%d)", (long long)e, (long long)r,
                    (long long)c, n_nodes);
    ++rowptr[r + 1];
  }
  for (size_t i = 0; i < n; ++i) rowptr[i + 1] += rowptr[i];
  {
    std::vector<int32_t> cur(rowptr, rowptr + n);
    for (int64_t e = 0; e < n_edges; ++e) {
      const int32_t r = static_cast<int32_t>(row[e]);
      const int32_t p = cur[r]++;
      colidx[p] = static_cast<int32_t>(col[e]);
      perm[p] = static_cast<int32_t>(e);
      rowidx[p] = r;
    }
  }
  if (cscptr && cscpos) {
    std::fill(cscptr, cscptr + n + 1, 0);
    for (int64_t p = 0; p < n_edges; ++p) ++cscptr[colidx[p] + 1];
    for (size_t i = 0; i < n; ++i) cscptr[i + 1] += cscptr[i];
    std::vector<int32_t> cur(cscptr, cscptr + n);
    for (int64_t p = 0; p < n_edges; ++p) cscpos[cur[colidx[p]]++] = static_cast<int32_t>(p);
  }
  int32_t lr = 0, lc = 0;
  for (int32_t i = 0; i < n_nodes; ++i) {
    const int32_t b = rowptr[i], e = rowptr[i + 1];
    if (e - b > GNPDE_LONG_ROW) {
      GNPDE_CHECK_ARG(long_rows && long_chunk_ptr && long_chunk_row && long_chunk_begin && long_chunk_end,
                      GNPDE_EINVAL, "graph_build: long rows present but chunk arrays are null");
      long_rows[lr] = i;
      long_chunk_ptr[lr] = lc;
      for (int32_t s = b; s < e; s += GNPDE_LONG_ROW) {
        long_chunk_row[lc] = i;
        if (long_chunk_first) long_chunk_first[lc] = long_chunk_ptr[lr];
        long_chunk_begin[lc] = s;
        long_chunk_end[lc] = std::min(e, s + GNPDE_LONG_ROW);
        ++lc;
      }
      ++lr;
    }
  }
  if (long_chunk_ptr) long_chunk_ptr[lr] = lc;
  if (cscptr && long_cols) {
    int32_t k = 0;
    for (int32_t i = 0; i < n_nodes; ++i)
      if (cscptr[i + 1] - cscptr[i] > GNPDE_LONG_ROW) long_cols[k++] = i;
  }
  if (bin_rows && bin_counts) {
    int32_t n16 = 0, n64 = 0;
    for (int32_t i = 0; i < n_nodes; ++i) {
      const int32_t dg = rowptr[i + 1] - rowptr[i];
      if (dg >= 1 && dg <= 16) ++n16;
      else if (dg > 16 && dg <= GNPDE_LONG_ROW) ++n64;
    }
    int32_t p16 = 0, p64 = n16;
    auto put = [&](int32_t slot, int32_t i, int32_t dg) {
      bin_rows[4 * slot + 0] = i;
      bin_rows[4 * slot + 1] = rowptr[i];
      bin_rows[4 * slot + 2] = dg;
      bin_rows[4 * slot + 3] = 0;
    };
    // Rows of 17..GNPDE_LONG_ROW entries: LONGEST FIRST (ties in row order; a counting sort over the length).  A wavefront
    // takes one such row and needs one pair of dependent round trips per 64 entries, so a 512-entry row runs ~8x as long as
    // the typical one: in row order the long rows that happen to start late form the tail of the row-attention launch,
    // longest-first they start at time 0 and the short rows fill in behind them (ogbn-arxiv shape, wave-slot model of the
    // launch: 30.6 -> 21.9 us, tools/attention_slot_model.py).  The records are processed independently of one another, so their
    // order does not enter any result.
    std::vector<int32_t> start(GNPDE_LONG_ROW + 2, 0);
    for (int32_t i = 0; i < n_nodes; ++i) {
      const int32_t dg = rowptr[i + 1] - rowptr[i];
      if (dg > 16 && dg <= GNPDE_LONG_ROW) ++start[dg];
    }
    {
      int32_t pos = n16;                              // slots of length GNPDE_LONG_ROW first, then downwards
      for (int32_t len = GNPDE_LONG_ROW; len > 16; --len) {
        const int32_t cnt = start[len];
        start[len] = pos;
        pos += cnt;
      }
    }
    for (int32_t i = 0; i < n_nodes; ++i) {
      const int32_t dg = rowptr[i + 1] - rowptr[i];
      if (dg >= 1 && dg <= 16) put(p16++, i, dg);
      else if (dg > 16 && dg <= GNPDE_LONG_ROW) put(start[dg]++, i, dg);
    }
    (void)p64;
    bin_counts[0] = n16;
    bin_counts[1] = n64;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// k-way row partition (METIS is not available offline).  Work of a row = its nnz + row_weight; row_weight = 1 balances on
// EDGES, which is what the aggregation time follows.
//   A. size-constrained label propagation: every node joins the cluster most of its neighbours are in,
//      clusters capped at a part / cluster_div (default a quarter)  -> communities (hubs do not glue everything together
//      because a full cluster stops accepting members);
//   B. clusters, heaviest first, are packed into the part they are most connected to that still has room;
//   C. balance-constrained label propagation on single nodes across parts.
// Deterministic for a given seed.
// ------------------------------------------------------------------------------------------------
extern "C" int gnpde_partition_rows(const int32_t* rowptr, const int32_t* colidx, int32_t n_nodes,
                                    int32_t n_parts, int32_t refine_iters, uint64_t seed, int32_t* part) {
  return gnpde_partition_rows_ex(rowptr, colidx, n_nodes, n_parts, refine_iters, seed, 1, 4, part);
}

extern "C" int gnpde_partition_rows_ex(const int32_t* rowptr, const int32_t* colidx, int32_t n_nodes, int32_t n_parts,
                                       int32_t refine_iters, uint64_t seed, int32_t row_weight_arg, int32_t cluster_div_arg,
                                       int32_t* part) {
  GNPDE_CHECK_ARG(rowptr && part && n_nodes >= 0 && n_parts >= 1 && row_weight_arg >= 1 && cluster_div_arg >= 1, GNPDE_EINVAL,
                  "partition_rows: bad args");
  GNPDE_CHECK_ARG(colidx || rowptr[n_nodes] == 0, GNPDE_EINVAL, "partition_rows: null colidx");
  const int32_t n = n_nodes, P = n_parts;
  if (P == 1 || n == 0) {
    std::fill(part, part + n, 0);
    return 0;
  }
  const int64_t row_weight = row_weight_arg;
  auto work = [&](int32_t v) -> int64_t { return int64_t(rowptr[v + 1] - rowptr[v]) + row_weight; };
  int64_t total = 0, maxw = 0;
  for (int32_t v = 0; v < n; ++v) {
    total += work(v);
    maxw = std::max(maxw, work(v));
  }
  const double avg = double(total) / P;
  const int64_t hi = static_cast<int64_t>(avg * 1.03) + maxw / 8 + 1, lo = static_cast<int64_t>(avg * 0.97) - maxw / 8;

  // visiting order: a fixed pseudo-random permutation (stride coprime with n)
  std::vector<int32_t> order(n);
  {
    uint64_t stride = (seed * 2654435761ULL + 40503ULL) % std::max<int64_t>(n, 1);
    if (stride == 0) stride = 1;
    auto gcd = [](uint64_t a, uint64_t b) { while (b) { uint64_t t = a % b; a = b; b = t; } return a; };
    while (gcd(stride, static_cast<uint64_t>(n)) != 1) ++stride;
    uint64_t cur = seed % n;
    for (int32_t i = 0; i < n; ++i) {
      order[i] = static_cast<int32_t>(cur);
      cur = (cur + stride) % n;
    }
  }

  // ---- A. clusters
  std::vector<int32_t> label(n);
  std::vector<int64_t> cw(n);
  for (int32_t v = 0; v < n; ++v) {
    label[v] = v;
    cw[v] = work(v);
  }
  const int cluster_div = cluster_div_arg;
  const int64_t cap = std::max<int64_t>(static_cast<int64_t>(avg / cluster_div), maxw);
  std::vector<int32_t> cnt(n, 0), touched;
  touched.reserve(256);
  const int cluster_iters = std::max(4, refine_iters);
  for (int it = 0; it < cluster_iters; ++it) {
    int64_t moved = 0;
    for (int32_t oi = 0; oi < n; ++oi) {
      const int32_t v = order[oi];
      const int32_t own = label[v];
      touched.clear();
      for (int32_t e = rowptr[v]; e < rowptr[v + 1]; ++e) {
        const int32_t u = colidx[e];
        if (u == v) continue;
        const int32_t lu = label[u];
        if (cnt[lu]++ == 0) touched.push_back(lu);
      }
      const int64_t w = work(v);
      int32_t best = own;
      int32_t best_cnt = cnt[own];
      for (int32_t lb : touched) {
        if (lb == own) continue;
        if (cnt[lb] > best_cnt && cw[lb] + w <= cap) {
          best = lb;
          best_cnt = cnt[lb];
        }
      }
      for (int32_t lb : touched) cnt[lb] = 0;
      if (best != own) {
        cw[own] -= w;
        cw[best] += w;
        label[v] = best;
        ++moved;
      }
    }
    if (moved == 0) break;
  }
  // compact cluster ids, member lists
  std::vector<int32_t> cid(n, -1);
  int32_t nc = 0;
  for (int32_t v = 0; v < n; ++v)
    if (cid[label[v]] < 0) cid[label[v]] = nc++;
  std::vector<int64_t> cweight(nc, 0);
  std::vector<int32_t> cptr(nc + 1, 0), members(n);
  for (int32_t v = 0; v < n; ++v) {
    label[v] = cid[label[v]];
    cweight[label[v]] += work(v);
    ++cptr[label[v] + 1];
  }
  for (int32_t c = 0; c < nc; ++c) cptr[c + 1] += cptr[c];
  {
    std::vector<int32_t> cur(cptr.begin(), cptr.end() - 1);
    for (int32_t v = 0; v < n; ++v) members[cur[label[v]]++] = v;
  }
  // ---- B. pack clusters into parts
  std::vector<int32_t> corder(nc);
  std::iota(corder.begin(), corder.end(), 0);
  std::stable_sort(corder.begin(), corder.end(), [&](int32_t a, int32_t b) { return cweight[a] > cweight[b]; });
  std::vector<int32_t> cpart(nc, -1);
  std::vector<int64_t> load(P, 0), conn(P, 0);
  for (int32_t c : corder) {
    std::fill(conn.begin(), conn.end(), 0);
    for (int32_t i = cptr[c]; i < cptr[c + 1]; ++i) {
      const int32_t v = members[i];
      for (int32_t e = rowptr[v]; e < rowptr[v + 1]; ++e) {
        const int32_t pc = cpart[label[colidx[e]]];
        if (pc >= 0) ++conn[pc];
      }
    }
    int32_t best = -1;
    for (int32_t p = 0; p < P; ++p) {
      if (load[p] + cweight[c] > hi) continue;
      if (best < 0 || conn[p] > conn[best] || (conn[p] == conn[best] && load[p] < load[best])) best = p;
    }
    if (best < 0) best = static_cast<int32_t>(std::min_element(load.begin(), load.end()) - load.begin());
    cpart[c] = best;
    load[best] += cweight[c];
  }
  for (int32_t v = 0; v < n; ++v) part[v] = cpart[label[v]];

  // ---- C. node-level refinement under the balance constraint
  std::vector<int32_t> pcnt(P, 0), ptouched;
  ptouched.reserve(P);
  for (int32_t it = 0; it < refine_iters; ++it) {
    int64_t moved = 0;
    for (int32_t oi = 0; oi < n; ++oi) {
      const int32_t v = order[oi];
      const int32_t from = part[v];
      ptouched.clear();
      for (int32_t e = rowptr[v]; e < rowptr[v + 1]; ++e) {
        const int32_t u = colidx[e];
        if (u == v) continue;
        if (pcnt[part[u]]++ == 0) ptouched.push_back(part[u]);
      }
      int32_t best = from, best_cnt = pcnt[from];
      for (int32_t p : ptouched) {
        if (p != from && pcnt[p] > best_cnt) {
          best = p;
          best_cnt = pcnt[p];
        }
      }
      for (int32_t p : ptouched) pcnt[p] = 0;
      if (best != from) {
        const int64_t w = work(v);
        if (load[best] + w <= hi && load[from] - w >= lo) {
          part[v] = best;
          load[best] += w;
          load[from] -= w;
          ++moved;
        }
      }
    }
    if (moved == 0) break;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Communication refinement of a k-way row partition (round 3).  What a partitioned evaluation pays for is not the edge cut
// but the ROWS each rank receives: M[r][q] = number of distinct nodes of part q that rows of part r reference = the rows that
// cross the xGMI link q -> r in every evaluation; the exchange ends when the busiest link has delivered.  The label-propagation
// partition above minimises cut edges only.  This pass moves single nodes between parts when that lowers the TOTAL number of
// received rows (sum of M) without raising the busiest link, keeping the parts inside the balance window of the partitioner
// (3 % on entries + row_weight per row); then a second phase works on the busiest links alone and accepts moves that lower
// sum_{links} max(0, M - 0.9 max M) even when the total does not fall.  Both phases are plain hill climbing with exact
// bookkeeping: ref[u][p] = rows of part p that reference node u (self references never travel and are ignored), from which M
// follows; a move v: a -> b changes (i) who receives v (the parts whose rows reference v: from a's link to b's link) and (ii)
// what a and b receive (the nodes row v references).  Deterministic.  stats (nullable): [0] busiest link before, [1] after,
// [2] total rows received before, [3] after, [4] moves applied.
// ------------------------------------------------------------------------------------------------
extern "C" int gnpde_partition_refine_links(const int32_t* rowptr, const int32_t* colidx, int32_t n_nodes, int32_t n_parts,
                                            int32_t row_weight_arg, int32_t max_passes, int32_t* part, int64_t* stats) {
  GNPDE_CHECK_ARG(rowptr && part && n_nodes >= 0 && n_parts >= 1 && row_weight_arg >= 1 && max_passes >= 0, GNPDE_EINVAL,
                  "partition_refine_links: bad args");
  GNPDE_CHECK_ARG(colidx || rowptr[n_nodes] == 0, GNPDE_EINVAL, "partition_refine_links: null colidx");
  const int32_t n = n_nodes, P = n_parts;
  if (stats) std::fill(stats, stats + 5, 0);
  if (P == 1 || n == 0 || max_passes == 0) return 0;
  GNPDE_CHECK_ARG(static_cast<int64_t>(n) * P <= (int64_t(1) << 31), GNPDE_ESHAPE,
                  "partition_refine_links: %d nodes x %d parts exceeds the dense reference table", n, P);
  for (int32_t v = 0; v < n; ++v)
    GNPDE_CHECK_ARG(part[v] >= 0 && part[v] < P, GNPDE_EINVAL, "partition_refine_links: part[%d] = %d", v, part[v]);
  const int64_t row_weight = row_weight_arg;
  auto work = [&](int32_t v) -> int64_t { return int64_t(rowptr[v + 1] - rowptr[v]) + row_weight; };
  std::vector<int64_t> load(P, 0);
  int64_t total = 0, maxw = 0;
  for (int32_t v = 0; v < n; ++v) {
    load[part[v]] += work(v);
    total += work(v);
    maxw = std::max(maxw, work(v));
  }
  const double avg = double(total) / P;
  const int64_t hi = std::max<int64_t>(static_cast<int64_t>(avg * 1.03) + maxw / 8 + 1, *std::max_element(load.begin(), load.end()));
  const int64_t lo = std::min<int64_t>(static_cast<int64_t>(avg * 0.97) - maxw / 8, *std::min_element(load.begin(), load.end()));

  // ref[u * P + p]: rows of part p referencing node u
  std::vector<int32_t> ref(static_cast<size_t>(n) * P, 0);
  for (int32_t v = 0; v < n; ++v)
    for (int32_t e = rowptr[v]; e < rowptr[v + 1]; ++e)
      if (colidx[e] != v) ++ref[static_cast<size_t>(colidx[e]) * P + part[v]];
  std::vector<int64_t> M(static_cast<size_t>(P) * P, 0);   // M[r * P + q]
  for (int32_t u = 0; u < n; ++u)
    for (int32_t r = 0; r < P; ++r)
      if (r != part[u] && ref[static_cast<size_t>(u) * P + r] > 0) ++M[static_cast<size_t>(r) * P + part[u]];
  auto max_link = [&]() { return *std::max_element(M.begin(), M.end()); };
  auto volume = [&]() { int64_t s = 0; for (int64_t x : M) s += x; return s; };
  const int64_t link0 = max_link(), vol0 = volume();

  // apply v: a -> b with exact bookkeeping; returns the change of the total volume.  `inc` collects the links that grew.
  std::vector<int32_t> inc;
  auto apply = [&](int32_t v, int32_t a, int32_t b) -> int64_t {
    int64_t dv = 0;
    const int32_t* rv = &ref[static_cast<size_t>(v) * P];
    for (int32_t r = 0; r < P; ++r) {          // (i) who receives v
      if (rv[r] <= 0) continue;
      if (r != a) { --M[static_cast<size_t>(r) * P + a]; --dv; }
      if (r != b) { ++M[static_cast<size_t>(r) * P + b]; ++dv; inc.push_back(r * P + b); }
    }
    for (int32_t e = rowptr[v]; e < rowptr[v + 1]; ++e) {   // (ii) what a and b receive
      const int32_t u = colidx[e];
      if (u == v) continue;
      const int32_t pu = part[u];              // (u != v, so pu is u's current part)
      int32_t* ru = &ref[static_cast<size_t>(u) * P];
      if (--ru[a] == 0 && pu != a) { --M[static_cast<size_t>(a) * P + pu]; --dv; }
      if (ru[b]++ == 0 && pu != b) { ++M[static_cast<size_t>(b) * P + pu]; ++dv; inc.push_back(b * P + pu); }
    }
    part[v] = b;
    const int64_t w = work(v);
    load[a] -= w;
    load[b] += w;
    return dv;
  };

  // visiting order: a fixed stride permutation
  std::vector<int32_t> order(n);
  {
    uint64_t stride = (2654435761ULL + 40503ULL) % std::max<int64_t>(n, 1);
    if (stride == 0) stride = 1;
    auto gcd = [](uint64_t x, uint64_t y) { while (y) { uint64_t t = x % y; x = y; y = t; } return x; };
    while (gcd(stride, static_cast<uint64_t>(n)) != 1) ++stride;
    uint64_t cur = 0;
    for (int32_t i = 0; i < n; ++i) { order[i] = static_cast<int32_t>(cur); cur = (cur + stride) % n; }
  }
  std::vector<int32_t> cand;
  std::vector<char> seen(P, 0);
  int64_t moves = 0;
  // excess over a threshold, summed over the links: the quantity phase 2 lowers
  auto excess = [&](int64_t thr) { int64_t s = 0; for (int64_t x : M) s += std::max<int64_t>(0, x - thr); return s; };

  for (int phase = 0; phase < 2; ++phase) {
    for (int32_t pass = 0; pass < max_passes; ++pass) {
      const int64_t cap = max_link();
      const int64_t thr = phase == 0 ? 0 : cap - cap / 10;
      int64_t pass_moves = 0;
      for (int32_t oi = 0; oi < n; ++oi) {
        const int32_t v = order[oi];
        const int32_t a = part[v];
        // candidate parts: those of the nodes v references and of the rows that reference v
        cand.clear();
        for (int32_t e = rowptr[v]; e < rowptr[v + 1]; ++e) {
          const int32_t p = part[colidx[e]];
          if (p != a && !seen[p]) { seen[p] = 1; cand.push_back(p); }
        }
        const int32_t* rv = &ref[static_cast<size_t>(v) * P];
        for (int32_t r = 0; r < P; ++r)
          if (r != a && rv[r] > 0 && !seen[r]) { seen[r] = 1; cand.push_back(r); }
        for (int32_t p : cand) seen[p] = 0;
        if (cand.empty()) continue;
        if (phase == 1) {   // only nodes that sit on a link above the threshold
          bool hot = false;
          for (int32_t r = 0; r < P && !hot; ++r)
            if (r != a && rv[r] > 0 && M[static_cast<size_t>(r) * P + a] > thr) hot = true;
          for (int32_t e = rowptr[v]; e < rowptr[v + 1] && !hot; ++e) {
            const int32_t pu = part[colidx[e]];
            if (pu != a && M[static_cast<size_t>(a) * P + pu] > thr) hot = true;
          }
          if (!hot) continue;
        }
        const int64_t w = work(v);
        int32_t best = -1;
        int64_t best_gain = 0;
        for (int32_t b : cand) {
          if (load[b] + w > hi || load[a] - w < lo) continue;
          inc.clear();
          const int64_t ex0 = phase == 1 ? excess(thr) : 0;
          const int64_t dv = apply(v, a, b);
          bool ok = true;
          for (int32_t l : inc)
            if (M[l] > cap) { ok = false; break; }
          int64_t gain = 0;
          if (ok) gain = phase == 0 ? -dv : (ex0 - excess(thr)) * 4 - dv;   // phase 2: excess first, volume as the tie-break
          inc.clear();
          (void)apply(v, b, a);     // undo
          if (ok && gain > best_gain) { best_gain = gain; best = b; }
        }
        if (best >= 0) {
          inc.clear();
          (void)apply(v, a, best);
          ++pass_moves;
        }
      }
      moves += pass_moves;
      if (pass_moves == 0) break;
    }
  }
  if (stats) {
    stats[0] = link0; stats[1] = max_link(); stats[2] = vol0; stats[3] = volume(); stats[4] = moves;
  }
  return 0;
}
