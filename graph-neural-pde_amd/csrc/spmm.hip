// Row-parallel CSR aggregation  ax[i] = sum_e w[e] * u[col_e]  fused with the diffusion epilogue
// f = alpha' (ax - u_i) + beta x0_i  and the fixed-step solver's stage algebra (gnpde.h).
//
// Replaces torch_sparse.spmm (index_select -> mul -> scatter_add_ with an [E,d] temporary) plus
// the elementwise tail of ODEFunc.forward (reference src/function_laplacian_diffusion.py:31-51,
// src/function_transformer_attention.py:35,46-53) and torchdiffeq's per-stage AXPY chain.
//
// Mapping (HBM/L2-gather bound, no MFMA): one 64-lane wavefront per row.  L = lanes that cover one
// neighbour's feature row (VEC floats each, K column tiles), G = 64/L neighbours are gathered by one
// wave instruction, U independent gathers are issued before the first FMA so that >= 8 x 16 B loads
// are in flight per lane.  The G partial sums are combined with an xor butterfly (deterministic, no
// atomics).  Rows longer than GNPDE_LONG_ROW are processed as independent chunks into a partial
// buffer and summed by a second small kernel, so a 13k-degree hub does not serialise on one wave.
#include "common.h"
#include "epilogue.h"

namespace gnpde {
namespace {

struct SpmmArgs {
  int item_base, item_end;   // work items [item_base, item_end): < n_long_chunks are long-row chunks, then the rows
  int n, n_long_chunks;
  const int* __restrict__ rowptr;
  const int* __restrict__ colidx;
  const int* __restrict__ lc_row;
  const int* __restrict__ lc_begin;
  const int* __restrict__ lc_end;
  const float* __restrict__ w;
  const float* __restrict__ u;
  int d, ld;
  float* plain_out;      // != nullptr: write ax only (no epilogue)
  float* partial;        // [n_long_chunks, ldp]
  int ldp;
  gnpde_epilogue_t ep;
};

template <int VEC, int L, int K, int U, bool NTI, bool NT>
__global__ __launch_bounds__(kBlock) void spmm_rows_kernel(const SpmmArgs a) {
  constexpr int G = kWave / L;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const unsigned blk = xcd_swizzle(blockIdx.x, gridDim.x);
  const int item = __builtin_amdgcn_readfirstlane(a.item_base + static_cast<int>(blk) * kWavesPerBlock + wave);
  if (item >= a.item_end) return;
  const int sub = lane / L;   // neighbour slot
  const int cl = lane % L;    // column lane

  // Work items: first the long-row chunks (512 edges each = the longest-running waves, so they start at time 0
  // and overlap with everything else instead of forming the kernel's tail), then the rows.
  int row, e0, e1;
  int chunk = -1;
  if (item >= a.n_long_chunks) {
    row = item - a.n_long_chunks;
    if (row >= a.n) return;
    e0 = a.rowptr[row];
    e1 = a.rowptr[row + 1];
    if (e1 - e0 > GNPDE_LONG_ROW) return;  // processed as chunks
  } else {
    chunk = item;
    row = a.lc_row[chunk];
    e0 = a.lc_begin[chunk];
    e1 = a.lc_end[chunk];
  }

  float acc[K][VEC];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[k][v] = 0.0f;

  for (int j = e0; j < e1; j += G * U) {
    float vals[U][K][VEC];
    float ww[U];
#pragma unroll
    for (int t = 0; t < U; ++t) {
      const int e = j + t * G + sub;
      ww[t] = 0.0f;
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int v = 0; v < VEC; ++v) vals[t][k][v] = 0.0f;
      if (e < e1) {  // masked lanes issue no memory request
        const int c = NTI ? __builtin_nontemporal_load(a.colidx + e) : a.colidx[e];
        ww[t] = NTI ? __builtin_nontemporal_load(a.w + e) : a.w[e];
        const float* src = a.u + static_cast<size_t>(c) * a.ld;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int col = (k * L + cl) * VEC;
          if (col < a.d) load_vec<VEC>(src + col, vals[t][k]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < U; ++t)
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[k][v] = fmaf(ww[t], vals[t][k][v], acc[k][v]);
  }

  // combine the G neighbour slots (lanes with equal column lane)
#pragma unroll
  for (int off = L; off < kWave; off <<= 1)
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[k][v] += __shfl_xor(acc[k][v], off, kWave);

  if (sub != 0) return;

  if (chunk >= 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int col = (k * L + cl) * VEC;
      if (col < a.d) store_vec<VEC>(a.partial + static_cast<size_t>(chunk) * a.ldp + col, acc[k]);
    }
    return;  // spmm_long_reduce_kernel folds the chunks of a row in chunk order
  }
  if (a.plain_out != nullptr) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int col = (k * L + cl) * VEC;
      if (col < a.d) store_vec<VEC>(a.plain_out + static_cast<size_t>(row) * a.ld + col, acc[k]);
    }
    return;
  }
  const float alpha = alpha_of(a.ep);
  const float beta = a.ep.x0 != nullptr ? *a.ep.beta : 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int col = (k * L + cl) * VEC;
    if (col < a.d) {
      const size_t off = static_cast<size_t>(row) * a.ld + col;
      float ui[VEC];
      load_vec<VEC>(a.u + off, ui);
      epilogue<VEC, NT>(a.ep, alpha, beta, off, acc[k], ui);
    }
  }
}

// One block per long row: sum its chunk partials in chunk order, then the epilogue.
__global__ __launch_bounds__(kBlock) void spmm_long_reduce_kernel(const SpmmArgs a, const int* __restrict__ long_rows,
                                                                 const int* __restrict__ long_chunk_ptr) {
  const int lr = blockIdx.x;
  const int row = long_rows[lr];
  const int c0 = long_chunk_ptr[lr], c1 = long_chunk_ptr[lr + 1];
  const bool plain = a.plain_out != nullptr;
  const float alpha = plain ? 0.0f : alpha_of(a.ep);
  const float beta = (!plain && a.ep.x0 != nullptr) ? *a.ep.beta : 0.0f;
  for (int col = threadIdx.x; col < a.d; col += blockDim.x) {
    float s = 0.0f;
    for (int c = c0; c < c1; ++c) s += a.partial[static_cast<size_t>(c) * a.ldp + col];
    const size_t off = static_cast<size_t>(row) * a.ld + col;
    if (plain) {
      a.plain_out[off] = s;
    } else {
      const float ax[1] = {s};
      const float ui[1] = {a.u[off]};
      epilogue<1, false>(a.ep, alpha, beta, off, ax, ui);
    }
  }
}

template <int VEC, int L, int K, int U, bool NTI, bool NT = NTI>
void launch_rows(const SpmmArgs& a, hipStream_t s) {
  const long long items = static_cast<long long>(a.item_end) - a.item_base;
  const unsigned grid = xcd_grid((items + kWavesPerBlock - 1) / kWavesPerBlock);
  hipLaunchKernelGGL((spmm_rows_kernel<VEC, L, K, U, NTI, NT>), dim3(grid), dim3(kBlock), 0, s, a);
}

template <int VEC>
int dispatch_rows(const SpmmArgs& a, hipStream_t s) {
  const int slots = (a.d + VEC - 1) / VEC;
  if (slots <= 8) launch_rows<VEC, 8, 1, 4, false, true>(a, s);
  else if (slots <= 16) launch_rows<VEC, 16, 1, 4, false, true>(a, s);
  else if (slots <= 32) {
    if constexpr (VEC == 4) {  // the headline shape (d = 128): tunable for A/B runs
      const int v = g_tune[GNPDE_TUNE_SPMM_VARIANT];
      switch (v) {
        case 1: launch_rows<4, 32, 1, 8, false>(a, s); break;
        case 2: launch_rows<4, 16, 2, 2, false>(a, s); break;
        case 3: launch_rows<4, 32, 1, 4, true>(a, s); break;
        case 4: launch_rows<4, 32, 1, 8, true>(a, s); break;
        case 5: launch_rows<4, 16, 2, 4, false>(a, s); break;
        case 6: launch_rows<4, 32, 1, 2, false>(a, s); break;
        case 7: launch_rows<4, 16, 2, 4, true>(a, s); break;
        case 8: launch_rows<4, 16, 2, 8, true>(a, s); break;
        case 9: launch_rows<4, 8, 4, 2, true>(a, s); break;
        case 10: launch_rows<4, 8, 4, 4, true>(a, s); break;
        case 11: launch_rows<4, 16, 2, 4, true, false>(a, s); break;
        case 12: launch_rows<4, 16, 2, 4, false, true>(a, s); break;
        case 13: launch_rows<4, 16, 2, 2, true>(a, s); break;
        case 14: launch_rows<4, 32, 1, 4, false, true>(a, s); break;
        case 15: launch_rows<4, 32, 1, 4, true, false>(a, s); break;
        case 16: launch_rows<4, 32, 1, 4, false>(a, s); break;
        default: launch_rows<4, 16, 2, 4, false, true>(a, s); break;  // measured best on MI355X (tools/spmm_ab.py)
      }
    } else {
      launch_rows<VEC, 16, 2, 4, false, true>(a, s);
    }
  }
  else if (slots <= 64) launch_rows<VEC, 32, 2, 4, false, true>(a, s);
  else if (slots <= 128) launch_rows<VEC, 64, 2, 2, false, true>(a, s);
  else if (slots <= 192) launch_rows<VEC, 64, 3, 1, false, true>(a, s);
  else if (slots <= 256) launch_rows<VEC, 64, 4, 1, false, true>(a, s);
  else {
    set_error("spmm: feature width d=%d too large for VEC=%d (max %d)", a.d, VEC, 256 * VEC);
    return GNPDE_ESHAPE;
  }
  return 0;
}


// ------------------------------------------------------------------------------------------------
// SDDMM: out[p] = scale * a[row_p] . b[col_p] for every stored entry -- the gradient of the aggregation
// w.r.t. the edge weights (d w_e = alpha' g_row . x_col), needed when attention_weights carry gradients.
// Same work items as the aggregation (512-entry chunks of the hub rows first, then one wavefront per row, XCD-aware
// block order), a_row slice kept in registers, L lanes per neighbour, U independent gathers in flight,
// xor-butterfly dot.  Entries are independent, so the hub chunks need no second pass.
// ------------------------------------------------------------------------------------------------
struct SddmmArgs {
  int n, n_long_chunks;
  const int* __restrict__ rowptr;
  const int* __restrict__ colidx;
  const int* __restrict__ lc_row;
  const int* __restrict__ lc_begin;
  const int* __restrict__ lc_end;
  const float* __restrict__ a;
  const float* __restrict__ b;
  int d, lda, ldb;
  const float* __restrict__ scale_ptr;
  int scale_sigmoid;
  float* __restrict__ out;
};

template <int VEC, int L, int K, int U>
__global__ __launch_bounds__(kBlock) void sddmm_kernel(const SddmmArgs s) {
  constexpr int G = kWave / L;
  const int lane = threadIdx.x & (kWave - 1);
  const unsigned blk = xcd_swizzle(blockIdx.x, gridDim.x);
  const int item = __builtin_amdgcn_readfirstlane(static_cast<int>(blk) * kWavesPerBlock + static_cast<int>(threadIdx.x >> 6));
  int row, e0, e1;
  if (item >= s.n_long_chunks) {
    row = item - s.n_long_chunks;
    if (row >= s.n) return;
    e0 = s.rowptr[row];
    e1 = s.rowptr[row + 1];
    if (e1 - e0 > GNPDE_LONG_ROW) return;  // processed as chunks
  } else {
    row = s.lc_row[item];
    e0 = s.lc_begin[item];
    e1 = s.lc_end[item];
  }
  const int sub = lane / L, cl = lane % L;
  float scale = 1.0f;
  if (s.scale_ptr != nullptr) {
    scale = *s.scale_ptr;
    if (s.scale_sigmoid) scale = 1.0f / (1.0f + expf(-scale));
  }
  float av[K][VEC];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int col = (k * L + cl) * VEC;
#pragma unroll
    for (int v = 0; v < VEC; ++v) av[k][v] = 0.0f;
    if (col < s.d) load_vec<VEC>(s.a + static_cast<size_t>(row) * s.lda + col, av[k]);
  }
  for (int j = e0; j < e1; j += G * U) {
    float bv[U][K][VEC];
#pragma unroll
    for (int t = 0; t < U; ++t) {
      const int e = j + t * G + sub;
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int v = 0; v < VEC; ++v) bv[t][k][v] = 0.0f;
      if (e < e1) {
        const float* src = s.b + static_cast<size_t>(s.colidx[e]) * s.ldb;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int col = (k * L + cl) * VEC;
          if (col < s.d) load_vec<VEC>(src + col, bv[t][k]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < U; ++t) {
      float p = 0.0f;
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int v = 0; v < VEC; ++v) p = fmaf(av[k][v], bv[t][k][v], p);
#pragma unroll
      for (int off = 1; off < L; off <<= 1) p += __shfl_xor(p, off, kWave);
      const int e = j + t * G + sub;
      if (cl == 0 && e < e1) s.out[e] = scale * p;
    }
  }
}

template <int VEC>
int dispatch_sddmm(const gnpde_graph_t* g, const float* a, const float* b, int d, int lda, int ldb, const float* scale,
                   int scale_sigmoid, float* out, hipStream_t st) {
  SddmmArgs s;
  s.n = g->n; s.n_long_chunks = g->n_long_chunks;
  s.rowptr = g->rowptr; s.colidx = g->colidx;
  s.lc_row = g->long_chunk_row; s.lc_begin = g->long_chunk_begin; s.lc_end = g->long_chunk_end;
  s.a = a; s.b = b; s.d = d; s.lda = lda; s.ldb = ldb; s.scale_ptr = scale; s.scale_sigmoid = scale_sigmoid; s.out = out;
  const long long items = static_cast<long long>(g->n) + g->n_long_chunks;
  const unsigned grid = xcd_grid((items + kWavesPerBlock - 1) / kWavesPerBlock);
  const int slots = (d + VEC - 1) / VEC;
#define GNPDE_SDDMM(LL, KK, UU) hipLaunchKernelGGL((sddmm_kernel<VEC, LL, KK, UU>), dim3(grid), dim3(kBlock), 0, st, s)
  if (slots <= 16) GNPDE_SDDMM(16, 1, 4);
  else if (slots <= 32) GNPDE_SDDMM(16, 2, 4);
  else if (slots <= 64) GNPDE_SDDMM(32, 2, 2);
  else if (slots <= 128) GNPDE_SDDMM(64, 2, 2);
  else if (slots <= 256) GNPDE_SDDMM(64, 4, 1);
  else {
    set_error("sddmm: feature width d=%d too large for VEC=%d", d, VEC);
    return GNPDE_ESHAPE;
  }
#undef GNPDE_SDDMM
  return 0;
}

inline bool aligned(const void* p, size_t a) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace

int launch_spmm_rhs(const gnpde_graph_t* g, const float* w_csr, const float* u, int d, int ld,
                    const gnpde_epilogue_t* epi, float* plain_out, void* ws, size_t ws_bytes, hipStream_t stream,
                    const Fork* fork) {
  GNPDE_CHECK_ARG(g && u && (w_csr || g->e == 0), GNPDE_EINVAL, "spmm: null pointer");
  GNPDE_CHECK_ARG(d >= 1 && ld >= d, GNPDE_EINVAL, "spmm: bad d=%d ld=%d", d, ld);
  GNPDE_CHECK_ARG((epi != nullptr) != (plain_out != nullptr), GNPDE_EINVAL, "spmm: need exactly one of epilogue / plain output");
  if (g->n == 0) return 0;
  SpmmArgs a{};
  a.n = g->n;
  a.n_long_chunks = g->n_long_chunks;
  a.rowptr = g->rowptr;
  a.colidx = g->colidx;
  a.lc_row = g->long_chunk_row;
  a.lc_begin = g->long_chunk_begin;
  a.lc_end = g->long_chunk_end;
  a.w = w_csr;
  a.u = u;
  a.d = d;
  a.ld = ld;
  a.plain_out = plain_out;
  a.ldp = static_cast<int>(align_up(static_cast<size_t>(d), 4));
  a.partial = static_cast<float*>(ws);
  if (g->n_long_chunks > 0) {
    const size_t need = static_cast<size_t>(g->n_long_chunks) * a.ldp * sizeof(float);
    GNPDE_CHECK_ARG(ws != nullptr && ws_bytes >= need, GNPDE_EWS, "spmm: workspace %zu < %zu bytes", ws_bytes, need);
    GNPDE_CHECK_ARG(g->long_rows && g->long_chunk_ptr && g->long_chunk_row && g->long_chunk_begin && g->long_chunk_end,
                    GNPDE_EINVAL, "spmm: long-row arrays missing");
  }
  bool a16 = (d % 4 == 0) && (ld % 4 == 0) && aligned(u, 16) && aligned(plain_out, 16) && aligned(ws, 16);
  bool a8 = (d % 2 == 0) && (ld % 2 == 0) && aligned(u, 8) && aligned(plain_out, 8) && aligned(ws, 8);
  if (epi) {
    a.ep = *epi;
    const gnpde_epilogue_t& e = a.ep;
    GNPDE_CHECK_ARG(e.alpha != nullptr, GNPDE_EINVAL, "spmm: alpha pointer is null");
    GNPDE_CHECK_ARG(e.x0 == nullptr || e.beta != nullptr, GNPDE_EINVAL, "spmm: x0 given without beta");
    GNPDE_CHECK_ARG(e.stage >= GNPDE_STAGE_RHS && e.stage <= GNPDE_STAGE_LINCOMB, GNPDE_EINVAL, "spmm: bad stage %d", e.stage);
    const int st = e.stage;
    if (st == GNPDE_STAGE_LINCOMB) {
      GNPDE_CHECK_ARG(e.n_prev >= 0 && e.n_prev <= GNPDE_MAX_PREV, GNPDE_EINVAL, "spmm: bad n_prev %d", e.n_prev);
      GNPDE_CHECK_ARG(e.out_k != nullptr || e.out_y != nullptr, GNPDE_EINVAL, "spmm: LINCOMB without output");
      GNPDE_CHECK_ARG(e.out_y == nullptr || e.y != nullptr, GNPDE_EINVAL, "spmm: LINCOMB needs y");
      for (int j = 0; j < e.n_prev; ++j) {
        GNPDE_CHECK_ARG(e.prev[j] != nullptr, GNPDE_EINVAL, "spmm: LINCOMB prev[%d] is null", j);
        a16 = a16 && aligned(e.prev[j], 16);
        a8 = a8 && aligned(e.prev[j], 8);
      }
    }
    const bool needs_k = st == GNPDE_STAGE_RHS || (st >= GNPDE_STAGE_RK1 && st <= GNPDE_STAGE_RK3);
    const bool needs_y = st == GNPDE_STAGE_EULER || (st >= GNPDE_STAGE_RK1 && st <= GNPDE_STAGE_RK4) ||
                         st == GNPDE_STAGE_RK2C || st == GNPDE_STAGE_RK4C;
    const bool needs_k1 = (st >= GNPDE_STAGE_RK2 && st <= GNPDE_STAGE_RK4) || st == GNPDE_STAGE_RK3C || st == GNPDE_STAGE_RK4C;
    GNPDE_CHECK_ARG(!needs_k || e.out_k != nullptr, GNPDE_EINVAL, "spmm: out_k is null");
    GNPDE_CHECK_ARG(st == GNPDE_STAGE_RHS || st == GNPDE_STAGE_LINCOMB || e.out_y != nullptr, GNPDE_EINVAL, "spmm: out_y is null");
    GNPDE_CHECK_ARG(!needs_y || e.y != nullptr, GNPDE_EINVAL, "spmm: y is null");
    GNPDE_CHECK_ARG(!needs_k1 || e.k1 != nullptr, GNPDE_EINVAL, "spmm: k1 is null");
    GNPDE_CHECK_ARG(!(st == GNPDE_STAGE_RK3 || st == GNPDE_STAGE_RK4) || e.k2 != nullptr, GNPDE_EINVAL, "spmm: k2 is null");
    GNPDE_CHECK_ARG(st != GNPDE_STAGE_RK4 || e.k3 != nullptr, GNPDE_EINVAL, "spmm: k3 is null");
    // every row gathers from u while other rows run their epilogue: outputs must not alias u
    GNPDE_CHECK_ARG(e.out_k != u && e.out_y != u, GNPDE_EINVAL, "spmm: output aliases the gathered operand");
    const void* ptrs[] = {e.x0, e.y, e.k1, e.k2, e.k3, e.out_k, e.out_y};
    for (const void* p : ptrs) {
      a16 = a16 && aligned(p, 16);
      a8 = a8 && aligned(p, 8);
    }
  } else {
    GNPDE_CHECK_ARG(plain_out != u, GNPDE_EINVAL, "spmm: output aliases the gathered operand");
  }
  auto run = [&](const SpmmArgs& arg, hipStream_t st) -> int {
    if (arg.item_end <= arg.item_base) return 0;
    if (a16) return dispatch_rows<4>(arg, st);
    if (a8) return dispatch_rows<2>(arg, st);
    return dispatch_rows<1>(arg, st);
  };
  const bool forked = fork != nullptr && fork->aux != nullptr && g->n_long_rows > 0;
  // rows and long-row chunks in one launch, then the per-row reduction of the chunk partials (a last-arriver
  // reduction inside the kernel was measured 1.7x slower: every agent-scope release fence writes back the
  // XCD's dirty L2 lines); with a fork the chunks + reduction run as a parallel branch
  a.item_base = forked ? g->n_long_chunks : 0;
  a.item_end = g->n_long_chunks + g->n;
  if (g->row_begin > 0) {  // boundary pass of a partitioned graph: rows [row_begin, n) only (chunks belong to them)
    GNPDE_CHECK_ARG(!forked && g->row_begin <= g->n, GNPDE_EINVAL, "spmm: bad row_begin %d", g->row_begin);
    if (g->n_long_chunks > 0) {
      SpmmArgs c = a;
      c.item_base = 0;
      c.item_end = g->n_long_chunks;
      const int rc0 = run(c, stream);
      if (rc0 != 0) return rc0;
      GNPDE_LAUNCH_CHECK();
    }
    a.item_base = g->n_long_chunks + g->row_begin;
  }
  int rc = run(a, stream);
  if (rc != 0) return rc;
  GNPDE_LAUNCH_CHECK();
  if (g->n_long_rows > 0) {
    hipStream_t br = forked ? fork_begin(fork, stream) : stream;
    if (forked) {
      SpmmArgs c = a;
      c.item_base = 0;
      c.item_end = g->n_long_chunks;
      rc = run(c, br);
      if (rc != 0) return rc;
      GNPDE_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(spmm_long_reduce_kernel, dim3(g->n_long_rows), dim3(kBlock), 0, br, a, g->long_rows,
                       g->long_chunk_ptr);
    GNPDE_LAUNCH_CHECK();
    if (forked) fork_end(fork, stream, br);
  }
  return 0;
}

}  // namespace gnpde

extern "C" size_t gnpde_spmm_workspace_bytes(const gnpde_graph_t* g, int32_t d) {
  if (!g || d < 1) return 0;
  return static_cast<size_t>(g->n_long_chunks) * gnpde::align_up(static_cast<size_t>(d), 4) * sizeof(float);
}

extern "C" int gnpde_spmm_rhs(const gnpde_graph_t* g, const float* w_csr, const float* u, int32_t d, int32_t ld,
                              const gnpde_epilogue_t* epi, void* workspace, size_t workspace_bytes, void* stream) {
  GNPDE_CHECK_ARG(epi != nullptr, GNPDE_EINVAL, "spmm_rhs: epilogue is null");
  return gnpde::launch_spmm_rhs(g, w_csr, u, d, ld, epi, nullptr, workspace, workspace_bytes,
                                static_cast<hipStream_t>(stream));
}

extern "C" int gnpde_spmm(const gnpde_graph_t* g, const float* w_csr, const float* u, int32_t d, int32_t ld,
                          float* out, void* workspace, size_t workspace_bytes, void* stream) {
  GNPDE_CHECK_ARG(out != nullptr, GNPDE_EINVAL, "spmm: out is null");
  return gnpde::launch_spmm_rhs(g, w_csr, u, d, ld, nullptr, out, workspace, workspace_bytes,
                                static_cast<hipStream_t>(stream));
}

extern "C" int gnpde_sddmm(const gnpde_graph_t* g, const float* a, int32_t lda, const float* b, int32_t ldb, int32_t d,
                           const float* scale, int32_t scale_sigmoid, float* out_csr, void* stream) {
  using namespace gnpde;
  GNPDE_CHECK_ARG(g && a && b && out_csr && d >= 1 && lda >= d && ldb >= d, GNPDE_EINVAL, "sddmm: bad arguments");
  if (g->n == 0 || g->e == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc;
  if (d % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && aligned(a, 16) && aligned(b, 16)) rc = dispatch_sddmm<4>(g, a, b, d, lda, ldb, scale, scale_sigmoid, out_csr, s);
  else if (d % 2 == 0 && lda % 2 == 0 && ldb % 2 == 0 && aligned(a, 8) && aligned(b, 8)) rc = dispatch_sddmm<2>(g, a, b, d, lda, ldb, scale, scale_sigmoid, out_csr, s);
  else rc = dispatch_sddmm<1>(g, a, b, d, lda, ldb, scale, scale_sigmoid, out_csr, s);
  if (rc) return rc;
  GNPDE_LAUNCH_CHECK();
  return 0;
}
