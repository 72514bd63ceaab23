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
  // work of one launch: long-row chunks [chunk_begin, chunk_end) and rows [row_begin, row_end)
  int chunk_begin, chunk_end, row_begin, row_end;
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
  int short_rows;        // host-side hint: at least half of the rows have <= 16 entries (-> row pairs for d = 68..128)
  int row_shift;         // rows are dealt to the XCDs in blocks of 2^row_shift rows (xcd_row); < 0: contiguous eighths
  gnpde_epilogue_t ep;
};


// Work item of wave `lw` (index local to the XCD) of a block that runs on XCD x = blockIdx % 8.  Every XCD gets every
// 8th long-row chunk FIRST (512 entries each = the longest-running waves: they start at time 0 and overlap with
// everything else instead of forming the kernel's tail) and then its share of the rows.  (Handing the chunks out
// contiguously put all of them -- 58 % of the entries of the R-MAT graph, 10 % at the ogbn-arxiv shape -- on XCD 0: 42 ms
// instead of 18.)
//
// Rows -> XCDs (gnpde_graph_t.xcd_deal).  CONTIGUOUS: every XCD takes a contiguous eighth of the rows, so that rows that are
// neighbours in a locality-ordered graph share that XCD's 4 MiB L2.  That is only balanced when the row length does not
// depend on the row id: in an R-MAT graph the expected degree falls by a factor 0.32 with every set bit of the id, so
// the eighth with the top bits 000 holds 7.5x the entries of the eighth with 111 (8.9 M vs 1.2 M non-hub entries at the 2^21
// shape) and the launch lasts as long as XCD 0 needs: 1.46x the balanced time by the work model (entries + 3 per row, hub
// chunks included; tools/xcd_balance.py).  Real graphs have the same trait (ids in order of publication, crawl or degree).
// HASHED (chosen by the graph builder when the contiguous deal is more than 3 % out of balance): the rows are dealt in BLOCKS
// of B = 2^row_shift consecutive rows (16 .. 128: the rows of a pair, and a stretch of neighbouring rows, stay on one XCD),
// eight consecutive blocks form a group, and the eight XCDs take the blocks of group j in an order rotated by a hash of j:
// block(x, j) = 8 j + ((x + hash(j)) mod 8).  A fixed rotation (plain round robin) would keep the R-MAT skew -- it only
// selects three OTHER id bits -- the hashed one leaves 0.1 - 0.5 % (same tool).  A bijection for any hash, so every row is
// still taken exactly once; which XCD takes it does not enter the arithmetic (bit-identical results, tools/xcd_check.py).
struct Item { int row, e0, e1, chunk; bool valid; };

__device__ __forceinline__ int chunks_of_xcd(const SpmmArgs& a, int x) {
  return (a.chunk_end - a.chunk_begin + kXcds - 1 - x) / kXcds;
}
__device__ __forceinline__ int rows_per_xcd(const SpmmArgs& a) { return xcd_rows_per(a.row_end - a.row_begin, a.row_shift); }
__device__ __forceinline__ int xcd_row(const SpmmArgs& a, int x, int r) { return xcd_row_of(a.row_begin, a.row_end, a.row_shift, x, r); }

__device__ __forceinline__ Item item_of(const SpmmArgs& a, int x, int lw) {
  Item it;
  it.valid = false;
  it.chunk = -1;
  it.row = it.e0 = it.e1 = 0;
  const int cx = chunks_of_xcd(a, x);
  if (lw < cx) {
    it.chunk = a.chunk_begin + lw * kXcds + x;
    it.row = __builtin_amdgcn_readfirstlane(a.lc_row[it.chunk]);
    it.e0 = __builtin_amdgcn_readfirstlane(a.lc_begin[it.chunk]);
    it.e1 = __builtin_amdgcn_readfirstlane(a.lc_end[it.chunk]);
    it.valid = true;
    return it;
  }
  const int row = xcd_row(a, x, lw - cx);
  if (row < 0) return it;
  it.row = row;
  it.e0 = __builtin_amdgcn_readfirstlane(a.rowptr[it.row]);
  it.e1 = __builtin_amdgcn_readfirstlane(a.rowptr[it.row + 1]);
  it.valid = it.e1 - it.e0 <= GNPDE_LONG_ROW;   // longer rows are processed as chunks
  return it;
}

// Block size of the row -> XCD deal: blocks of 16 .. 128 rows, >= 8192 of them where the graph is large enough (the more
// blocks an XCD draws, the closer the hash brings heavy-tailed row lengths to the mean: 0.2 % at the R-MAT shape with
// 16 384 blocks of 128, 5 % with 2 048 blocks of 1 024); always even (row pairs stay together).
inline int choose_row_shift(long long rn, int graph_deal) {
  const int knob = g_tune[GNPDE_TUNE_XCD_ROWS];           // A/B: 1 = contiguous eighths, 2 = hashed blocks, whatever the graph says
  const bool hashed = knob == 2 || (knob != 1 && graph_deal == GNPDE_XCD_HASHED);
  if (!hashed) return -1;
  int s = 4;
  while (s < 7 && (rn >> (s + 1)) >= 8192) ++s;
  return s;
}

// blocks of WPB waves so that every XCD can reach the end of its local list
inline unsigned balanced_grid(const SpmmArgs& a, int wpb) {
  const long long cn = a.chunk_end - a.chunk_begin, rn = a.row_end - a.row_begin;
  const long long per_xcd = (cn + kXcds - 1) / kXcds + xcd_rows_per(static_cast<int>(rn), a.row_shift);
  long long blocks = (per_xcd + wpb - 1) / wpb;
  if (blocks < 1) blocks = 1;
  return static_cast<unsigned>(blocks * kXcds);
}

template <int VEC, int L, int K, int U, bool NTI, bool NT, int BLK = kBlock>
__global__ __launch_bounds__(BLK) void spmm_rows_kernel(const SpmmArgs a) {
  constexpr int G = kWave / L;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int xcd = static_cast<int>(blockIdx.x % kXcds);
  const int lw = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x / kXcds) * (BLK / kWave) + wave);
  const Item it = item_of(a, xcd, lw);
  if (!it.valid) return;
  const int sub = lane / L;   // neighbour slot
  const int cl = lane % L;    // column lane
  const int row = it.row, e0 = it.e0, e1 = it.e1, chunk = it.chunk;

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


// ------------------------------------------------------------------------------------------------
// Wide-row variant (state widths whose row fits ONE wave instruction: d <= 64 VEC).  Differences from
// spmm_rows_kernel, all aimed at the HBM-bound shapes (RMAT, d = 256: 1-KB rows, table >> Infinity Cache):
//  * the column ids and weights of up to 64 entries are fetched by ONE coalesced load per wave (lane = entry)
//    and handed to the gather loop by v_readlane (G == 1: the neighbour is wave-uniform, so the row base
//    address lives in SGPRs and the gather is `global_load_dwordx4 v, v_off, s[base]`) or by ds_bpermute
//    (G > 1), instead of every lane group re-loading them: the dependent chain per 64 entries is
//    index load -> U gathers, not (index load -> gather) x 64/(G U);
//  * the per-row streaming operands of the epilogue (u_i, x0_i, y_i, k1_i) are requested BEFORE the gather
//    loop, so their latency overlaps the gathers instead of extending the wave's life;
//  * BLK = 64: one wavefront per workgroup, so a finished wave frees its slot at once (rows of a power-law
//    graph differ by 100x in length; with 4 waves per workgroup the slots of the short ones idle until the
//    longest is done);  PERSIST: a resident grid strides over the work items instead of one launch slot per row.
// Same work items, same summation order inside a row (entries in CSR order, G slots combined by the xor
// butterfly), same epilogue arithmetic.
// ------------------------------------------------------------------------------------------------
template <int VEC, bool NT>
struct Pre {   // epilogue operands requested ahead of the gather loop
  float ui[VEC], x0[VEC], y[VEC], k1[VEC];
};

template <int VEC, bool NT>
__device__ __forceinline__ bool stage_prefetchable(int stage) {
  return stage == GNPDE_STAGE_RHS || stage == GNPDE_STAGE_EULER || (stage >= GNPDE_STAGE_RK1C && stage <= GNPDE_STAGE_RK4C);
}

// k = alpha (ax - u_i) + beta x0_i and the compact / euler / plain stages from operands already in registers
template <int VEC, bool NT>
__device__ __forceinline__ void epilogue_pre(const gnpde_epilogue_t& ep, float alpha, float beta, size_t off,
                                             const float (&ax)[VEC], const Pre<VEC, NT>& p) {
  auto st = [](float* q, const float (&v)[VEC]) { if constexpr (NT) store_vec_nt<VEC>(q, v); else store_vec<VEC>(q, v); };
  constexpr float kThird = 1.0f / 3.0f;
  float k[VEC], o[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) k[v] = alpha * (ax[v] - p.ui[v]);
  if (ep.x0 != nullptr) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) k[v] = k[v] + beta * p.x0[v];
  }
  const float dt = ep.dt;
  switch (ep.stage) {
    case GNPDE_STAGE_RHS:
      st(ep.out_k + off, k);
      break;
    case GNPDE_STAGE_EULER:
#pragma unroll
      for (int v = 0; v < VEC; ++v) o[v] = p.y[v] + dt * k[v];
      st(ep.out_y + off, o);
      break;
    case GNPDE_STAGE_RK1C:
#pragma unroll
      for (int v = 0; v < VEC; ++v) o[v] = p.ui[v] + (dt * k[v]) * kThird;
      st(ep.out_y + off, o);
      break;
    case GNPDE_STAGE_RK2C:
#pragma unroll
      for (int v = 0; v < VEC; ++v) o[v] = (2.0f * p.y[v] - p.ui[v]) + dt * k[v];
      st(ep.out_y + off, o);
      break;
    case GNPDE_STAGE_RK3C:
#pragma unroll
      for (int v = 0; v < VEC; ++v) o[v] = (2.0f * p.k1[v] - p.ui[v]) + dt * k[v];
      st(ep.out_y + off, o);
      break;
    case GNPDE_STAGE_RK4C:
#pragma unroll
      for (int v = 0; v < VEC; ++v) o[v] = (((6.0f * p.k1[v] + 3.0f * p.ui[v]) - p.y[v]) + dt * k[v]) * 0.125f;
      st(ep.out_y + off, o);
      break;
    default:
      break;
  }
}


// The long-row fold on 16-byte lanes: one wavefront per long row, the epilogue operands requested before the chunk loop and the
// chunk partials fetched eight at a time (the scalar kernel above walks a chain of dependent-latency loads: 8.4 us for the 204
// hub rows of the ogbn-arxiv shape, 99 us for the 27 896 of the R-MAT shape).  Same summation: the chunks of a row in chunk order.
__global__ __launch_bounds__(kWave) void spmm_long_reduce4_kernel(const SpmmArgs a, const int* __restrict__ long_rows,
                                                                 const int* __restrict__ long_chunk_ptr) {
  constexpr int VEC = 4;
  const int lr = blockIdx.x;
  const int row = long_rows[lr];
  const int c0 = long_chunk_ptr[lr], c1 = long_chunk_ptr[lr + 1];
  const bool plain = a.plain_out != nullptr;
  const float alpha = plain ? 0.0f : alpha_of(a.ep);
  const float beta = (!plain && a.ep.x0 != nullptr) ? *a.ep.beta : 0.0f;
  const bool pre_ok = !plain && stage_prefetchable<VEC, false>(a.ep.stage);
  for (int col = static_cast<int>(threadIdx.x) * VEC; col < a.d; col += kWave * VEC) {
    const size_t off = static_cast<size_t>(row) * a.ld + col;
    Pre<VEC, false> pre;
    if (pre_ok) {
      load_vec<VEC>(a.u + off, pre.ui);
      if (a.ep.x0 != nullptr) load_vec<VEC>(a.ep.x0 + off, pre.x0);
      const int st = a.ep.stage;
      if (st == GNPDE_STAGE_EULER || st == GNPDE_STAGE_RK2C || st == GNPDE_STAGE_RK4C) load_vec<VEC>(a.ep.y + off, pre.y);
      if (st == GNPDE_STAGE_RK3C || st == GNPDE_STAGE_RK4C) load_vec<VEC>(a.ep.k1 + off, pre.k1);
    }
    float acc[VEC] = {0.0f, 0.0f, 0.0f, 0.0f};
    const float* p = a.partial + static_cast<size_t>(c0) * a.ldp + col;
    int c = c0;
    for (; c + 8 <= c1; c += 8, p += 8 * static_cast<size_t>(a.ldp)) {
      float v[8][VEC];
#pragma unroll
      for (int t = 0; t < 8; ++t) load_vec<VEC>(p + t * static_cast<size_t>(a.ldp), v[t]);
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] += v[t][q];
    }
    for (; c < c1; ++c, p += a.ldp) {
      float v[VEC];
      load_vec<VEC>(p, v);
#pragma unroll
      for (int q = 0; q < VEC; ++q) acc[q] += v[q];
    }
    if (plain) {
      store_vec<VEC>(a.plain_out + off, acc);
    } else if (pre_ok) {
      epilogue_pre<VEC, false>(a.ep, alpha, beta, off, acc, pre);
    } else {
      float ui[VEC];
      load_vec<VEC>(a.u + off, ui);
      epilogue<VEC, false>(a.ep, alpha, beta, off, acc, ui);
    }
  }
}

// U gathers of one wave (G neighbours each) from the 64 (column id, weight) pairs held one per lane in cv / wv.
// G == 1: the entry is wave-uniform -> v_readlane, row base address in SGPRs.  TAIL: entries >= cnt are skipped.
template <int VEC, int L, int U, bool TAIL, bool FULL>
__device__ __forceinline__ void gather_batch(const SpmmArgs& a, int cv, float wv, int t0, int cnt, int sub, int col,
                                             bool col_ok_rt, float (&acc)[VEC]) {
  const bool col_ok = FULL ? true : col_ok_rt;   // FULL: d == L * VEC, no column predicate
  constexpr int G = kWave / L;
  float vals[U][VEC];
  float ww[U];
  int cs[U];
  bool oks[U];
  // ids / weights of the whole batch first, then the row loads (one loop made the predicated tail a chain of LDS round trips)
#pragma unroll
  for (int t = 0; t < U; ++t) {
    bool ok = col_ok;
    if constexpr (G == 1) {
      const int idx = t0 + t;
      cs[t] = __builtin_amdgcn_readlane(cv, idx & (kWave - 1));
      ww[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wv), idx & (kWave - 1)));   // bit pattern, not a numeric conversion
      if constexpr (TAIL) ok = ok && idx < cnt;
    } else {
      const int idx = t0 + t * G + sub;
      cs[t] = __shfl(cv, idx & (kWave - 1), kWave);
      ww[t] = __shfl(wv, idx & (kWave - 1), kWave);
      if constexpr (TAIL) ok = ok && idx < cnt;
    }
    if constexpr (TAIL) ww[t] = ok ? ww[t] : 0.0f;
    oks[t] = ok;
  }
#pragma unroll
  for (int t = 0; t < U; ++t) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) vals[t][v] = 0.0f;
    const float* rowp = a.u + static_cast<size_t>(cs[t]) * a.ld;
    if (oks[t]) load_vec<VEC>(rowp + col, vals[t]);
  }
#pragma unroll
  for (int t = 0; t < U; ++t)
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = fmaf(ww[t], vals[t][v], acc[v]);
}

template <int VEC, int L, int U, int BLK, bool PERSIST, bool NT, bool FULL>
__global__ __launch_bounds__(BLK) void spmm_wide_kernel(const SpmmArgs a) {
  constexpr int G = kWave / L;
  constexpr int WPB = BLK / kWave;
  const int lane = threadIdx.x & (kWave - 1);
  const int sub = lane / L;   // neighbour slot
  const int cl = lane % L;    // column lane
  const int col = cl * VEC;
  const bool col_ok = FULL ? true : col < a.d;
  const int xcd = static_cast<int>(blockIdx.x % kXcds);
  int lw = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x / kXcds) * WPB + static_cast<int>(threadIdx.x >> 6));
  const int stride = static_cast<int>(gridDim.x / kXcds) * WPB;   // waves per XCD of this launch
  const int lw_end = chunks_of_xcd(a, xcd) + rows_per_xcd(a);
  const bool pre_ok = a.plain_out == nullptr && stage_prefetchable<VEC, NT>(a.ep.stage);
  const float alpha = a.plain_out == nullptr ? alpha_of(a.ep) : 0.0f;
  const float beta = (a.plain_out == nullptr && a.ep.x0 != nullptr) ? *a.ep.beta : 0.0f;

  for (; lw < lw_end; lw += stride) {
    const Item it = item_of(a, xcd, lw);
    if (!it.valid) {
      if constexpr (PERSIST) continue; else break;
    }
    const int row = it.row, e0 = it.e0, e1 = it.e1, chunk = it.chunk;
    const size_t off = static_cast<size_t>(row) * a.ld + col;

    // epilogue operands: in flight while the neighbours are gathered
    Pre<VEC, NT> pre;
    const bool do_pre = pre_ok && chunk < 0 && sub == 0 && col_ok;
    if (do_pre) {
      auto ldp = [](const float* q, float (&v)[VEC]) { if constexpr (NT) load_vec_nt<VEC>(q, v); else load_vec<VEC>(q, v); };
      load_vec<VEC>(a.u + off, pre.ui);
      if (a.ep.x0 != nullptr) ldp(a.ep.x0 + off, pre.x0);
      const int st = a.ep.stage;
      if (st == GNPDE_STAGE_EULER || st == GNPDE_STAGE_RK2C || st == GNPDE_STAGE_RK4C) ldp(a.ep.y + off, pre.y);
      if (st == GNPDE_STAGE_RK3C || st == GNPDE_STAGE_RK4C) ldp(a.ep.k1 + off, pre.k1);
    }

    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.0f;

    for (int base = e0; base < e1; base += kWave) {
      const int me = base + lane;
      const bool in = me < e1;
      const int cv = in ? a.colidx[me] : 0;     // ONE coalesced load of 64 column ids ...
      const float wv = in ? a.w[me] : 0.0f;     // ... and weights per wave
      const int cnt = (e1 - base) < kWave ? (e1 - base) : kWave;   // wave-uniform
      // full batches of G*U entries without predicates (all U gathers issue back to back), then one predicated tail
      int t0 = 0;
      for (; t0 + G * U <= cnt; t0 += G * U) gather_batch<VEC, L, U, false, FULL>(a, cv, wv, t0, cnt, sub, col, col_ok, acc);
      if (t0 < cnt) gather_batch<VEC, L, U, true, FULL>(a, cv, wv, t0, cnt, sub, col, col_ok, acc);
    }

    // combine the G neighbour slots (lanes with equal column lane)
#pragma unroll
    for (int o = L; o < kWave; o <<= 1)
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] += __shfl_xor(acc[v], o, kWave);

    if (sub != 0 || !col_ok) continue;
    if (chunk >= 0) {
      store_vec<VEC>(a.partial + static_cast<size_t>(chunk) * a.ldp + col, acc);
      continue;  // spmm_long_reduce_kernel folds the chunks of a row in chunk order
    }
    if (a.plain_out != nullptr) {
      store_vec<VEC>(a.plain_out + off, acc);
      continue;
    }
    if (pre_ok) {
      epilogue_pre<VEC, NT>(a.ep, alpha, beta, off, acc, pre);
    } else {
      float ui[VEC];
      load_vec<VEC>(a.u + off, ui);
      epilogue<VEC, NT>(a.ep, alpha, beta, off, acc, ui);
    }
    if constexpr (!PERSIST) break;
  }
}


// ------------------------------------------------------------------------------------------------
// Row-PAIR variant of the wide-row kernel for rows of 17..32 16-byte lanes (d = 68..128, the ogbn-arxiv shape).
// There a row occupies half a wavefront, and the wide kernel lets the two halves share ONE row (two neighbour slots):
// with a median of 8 entries per row a wave then has 4 KB of neighbour rows in flight and runs its epilogue on 32 of its 64
// lanes.  Here the two halves of a wave take two CONSECUTIVE rows when both have <= 32 entries (one coalesced 32-entry index
// load each): twice the bytes in flight per wave for the same registers, half the waves, every lane busy in the epilogue.
// Each half keeps the wide kernel's summation order -- even entries into one accumulator, odd entries into a second, in
// CSR order, then even + odd -- so the result is bit-identical to the shared-row mode whatever row a row is paired with.
// A pair with a longer row falls back to the shared-row mode for its two rows in turn; hub chunks are shared-row items.
template <int VEC, int U, bool NT, bool FULL>
__device__ __forceinline__ void shared_row_item(const SpmmArgs& a, int row, int e0, int e1, int chunk, int lane, int sub, int col,
                                                bool col_ok, bool pre_ok, float alpha, float beta) {
  constexpr int L = 32, G = 2;
  const size_t off = static_cast<size_t>(row) * a.ld + col;
  Pre<VEC, NT> pre;
  const bool do_pre = pre_ok && chunk < 0 && sub == 0 && col_ok;
  if (do_pre) {
    auto ldp = [](const float* q, float (&v)[VEC]) { if constexpr (NT) load_vec_nt<VEC>(q, v); else load_vec<VEC>(q, v); };
    load_vec<VEC>(a.u + off, pre.ui);
    if (a.ep.x0 != nullptr) ldp(a.ep.x0 + off, pre.x0);
    const int st = a.ep.stage;
    if (st == GNPDE_STAGE_EULER || st == GNPDE_STAGE_RK2C || st == GNPDE_STAGE_RK4C) ldp(a.ep.y + off, pre.y);
    if (st == GNPDE_STAGE_RK3C || st == GNPDE_STAGE_RK4C) ldp(a.ep.k1 + off, pre.k1);
  }
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.0f;
  for (int base = e0; base < e1; base += kWave) {
    const int me = base + lane;
    const bool in = me < e1;
    const int cv = in ? a.colidx[me] : 0;
    const float wv = in ? a.w[me] : 0.0f;
    const int cnt = (e1 - base) < kWave ? (e1 - base) : kWave;
    int t0 = 0;
    for (; t0 + G * U <= cnt; t0 += G * U) gather_batch<VEC, L, U, false, FULL>(a, cv, wv, t0, cnt, sub, col, col_ok, acc);
    if (t0 < cnt) gather_batch<VEC, L, U, true, FULL>(a, cv, wv, t0, cnt, sub, col, col_ok, acc);
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] += __shfl_xor(acc[v], L, kWave);
  if (sub != 0 || !col_ok) return;
  if (chunk >= 0) {
    store_vec<VEC>(a.partial + static_cast<size_t>(chunk) * a.ldp + col, acc);
    return;
  }
  if (a.plain_out != nullptr) {
    store_vec<VEC>(a.plain_out + off, acc);
    return;
  }
  if (pre_ok) {
    epilogue_pre<VEC, NT>(a.ep, alpha, beta, off, acc, pre);
  } else {
    float ui[VEC];
    load_vec<VEC>(a.u + off, ui);
    epilogue<VEC, NT>(a.ep, alpha, beta, off, acc, ui);
  }
}

template <int VEC, int U, int BLK, bool NT, bool FULL>
__global__ __launch_bounds__(BLK) void spmm_pair_kernel(const SpmmArgs a) {
  constexpr int L = 32;
  constexpr int WPB = BLK / kWave;
  const int lane = threadIdx.x & (kWave - 1);
  const int half = lane >> 5;
  const int cl = lane & (L - 1);
  const int col = cl * VEC;
  const bool col_ok = FULL ? true : col < a.d;
  const int xcd = static_cast<int>(blockIdx.x % kXcds);
  const int lw = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x / kXcds) * WPB + static_cast<int>(threadIdx.x >> 6));
  const bool pre_ok = a.plain_out == nullptr && stage_prefetchable<VEC, NT>(a.ep.stage);
  const float alpha = a.plain_out == nullptr ? alpha_of(a.ep) : 0.0f;
  const float beta = (a.plain_out == nullptr && a.ep.x0 != nullptr) ? *a.ep.beta : 0.0f;

  const int cx = chunks_of_xcd(a, xcd);
  if (lw < cx) {   // hub chunk first (as item_of)
    const int chunk = a.chunk_begin + lw * kXcds + xcd;
    const int row = __builtin_amdgcn_readfirstlane(a.lc_row[chunk]);
    const int e0 = __builtin_amdgcn_readfirstlane(a.lc_begin[chunk]);
    const int e1 = __builtin_amdgcn_readfirstlane(a.lc_end[chunk]);
    shared_row_item<VEC, U, NT, FULL>(a, row, e0, e1, chunk, lane, half, col, col_ok, pre_ok, alpha, beta);
    return;
  }
  const int per = rows_per_xcd(a);
  const int r0 = 2 * (lw - cx);
  if (r0 >= per) return;
  int row = xcd_row(a, xcd, r0 + half);         // r0 is even and the blocks have an even size: both rows of a pair are consecutive
  bool have = row >= 0;
  if (!have) row = 0;
  int e0 = 0, e1 = 0;
  if (have) {
    e0 = a.rowptr[row];
    e1 = a.rowptr[row + 1];
  }
  if (e1 - e0 > GNPDE_LONG_ROW) have = false;   // processed as chunks
  const int len = have ? e1 - e0 : 0;
  const int len_a = __builtin_amdgcn_readlane(len, 0), len_b = __builtin_amdgcn_readlane(len, L);
  if (len_a > L || len_b > L) {                 // a longer row in the pair: both halves share each row in turn
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int hv = __builtin_amdgcn_readlane(static_cast<int>(have), h * L);
      if (!hv) continue;
      const int hr = __builtin_amdgcn_readlane(row, h * L);
      const int h0 = __builtin_amdgcn_readlane(e0, h * L), h1 = __builtin_amdgcn_readlane(e1, h * L);
      shared_row_item<VEC, U, NT, FULL>(a, hr, h0, h1, -1, lane, half, col, col_ok, pre_ok, alpha, beta);
    }
    return;
  }

  const size_t off = static_cast<size_t>(row) * a.ld + col;
  Pre<VEC, NT> pre;
  const bool mine = have && col_ok;
  if (pre_ok && mine) {
    auto ldp = [](const float* q, float (&v)[VEC]) { if constexpr (NT) load_vec_nt<VEC>(q, v); else load_vec<VEC>(q, v); };
    load_vec<VEC>(a.u + off, pre.ui);
    if (a.ep.x0 != nullptr) ldp(a.ep.x0 + off, pre.x0);
    const int st = a.ep.stage;
    if (st == GNPDE_STAGE_EULER || st == GNPDE_STAGE_RK2C || st == GNPDE_STAGE_RK4C) ldp(a.ep.y + off, pre.y);
    if (st == GNPDE_STAGE_RK3C || st == GNPDE_STAGE_RK4C) ldp(a.ep.k1 + off, pre.k1);
  }
  const bool in = cl < len;
  const int cv = in ? a.colidx[e0 + cl] : 0;     // the whole row: one coalesced load per half
  const float wv = in ? a.w[e0 + cl] : 0.0f;
  const int cmax = len_a > len_b ? len_a : len_b;
  float acc0[VEC], acc1[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc0[v] = acc1[v] = 0.0f;
  for (int t0 = 0; t0 < cmax; t0 += U) {
    float vals[U][VEC];
    float ww[U];
    // (round 3) all ids / weights of the batch are handed out FIRST, then the row loads: written as one loop, the compiler
    // emitted bpermute, bpermute, wait, predicated load per entry -- sixteen LDS round trips between the first and the last
    // gather of a wave
    int cs[U];
#pragma unroll
    for (int t = 0; t < U; ++t) {
      const int idx = t0 + t;                    // < 32 (cmax <= 32, U divides 32)
      cs[t] = __shfl(cv, (half << 5) + idx, kWave);
      const float w = __shfl(wv, (half << 5) + idx, kWave);
      ww[t] = (col_ok && idx < len) ? w : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < U; ++t) {
      const bool ok = col_ok && t0 + t < len;
#pragma unroll
      for (int v = 0; v < VEC; ++v) vals[t][v] = 0.0f;
      if (ok) load_vec<VEC>(a.u + static_cast<size_t>(cs[t]) * a.ld + col, vals[t]);
    }
#pragma unroll
    for (int t = 0; t < U; ++t) {
      if (t % 2 == 0) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc0[v] = fmaf(ww[t], vals[t][v], acc0[v]);
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc1[v] = fmaf(ww[t], vals[t][v], acc1[v]);
      }
    }
  }
  if (!mine) return;
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = acc0[v] + acc1[v];
  if (a.plain_out != nullptr) {
    store_vec<VEC>(a.plain_out + off, acc);
    return;
  }
  if (pre_ok) {
    epilogue_pre<VEC, NT>(a.ep, alpha, beta, off, acc, pre);
  } else {
    float ui[VEC];
    load_vec<VEC>(a.u + off, ui);
    epilogue<VEC, NT>(a.ep, alpha, beta, off, acc, ui);
  }
}

// ------------------------------------------------------------------------------------------------
// Row attention + aggregation in ONE kernel (scaled-dot scores, softmax over the row, head mean): what
// row_attention_sd_kernel (two launches) + spmm_wide_kernel did, per row and per wave, without the [E] weight round trip,
// the second read of the column ids and the two launch boundaries -- the attention's dependent gathers (ids -> k rows)
// now overlap with the neighbouring waves' x-row gathers instead of running as a latency-bound kernel of their own.
//   scores : lane = (entry slot, head), head fastest: the H lanes of an entry read ONE contiguous A-float row of k
//            (DK4 16-byte loads each), 64 / H entries per pass; max / sum over the slots by xor butterflies with stride H
//   weights: w_e = (1/H) sum_h exp(s_eh - m_h) / (l_h + 1e-16), summed over the H lanes of the entry, then redistributed
//            so that lane e of the wave holds the weight of entry e of the 64-entry chunk (what gather_batch expects)
//   rows with <= 64 entries keep their scores in registers; longer rows (<= 512) take an online-softmax pass over their
//   chunks (running max / sum per lane) and RECOMPUTE the scores chunk by chunk in the aggregation pass (the k rows are
//   64 B against 512 B of x per entry); hub chunks (rows > 512) take their weights from memory, written by the two small
//   hub launches of attention.hip (launch_hub_attention) before this kernel.
// ------------------------------------------------------------------------------------------------
struct AttnSpmmArgs {
  SpmmArgs s;
  const float* __restrict__ q;       // [n, ldqk] row-side projection
  const float* __restrict__ k;       // [n_cols, ldqk] column-side projection
  int ldqk;
  float inv_sqrt_dk;
  const float* __restrict__ edge_w;  // CSR order or null (reweight_attention)
};

// scores of the 64-entry chunk starting at `base` for this lane's (slot, head): PASSES = H registers, entry = p * (64/H) + slot
template <int H, int DK4>
__device__ __forceinline__ void chunk_scores(const AttnSpmmArgs& fa, int cv, int base, int cnt, int slot, int head,
                                             const float (&qv)[DK4 * 4], float (&sc)[H]) {
  constexpr int ES = kWave / H;
#pragma unroll
  for (int p = 0; p < H; ++p) {
    const int idx = p * ES + slot;
    const int c = __shfl(cv, idx, kWave);
    sc[p] = -INFINITY;
    if (idx < cnt) {
      const float* kp = fa.k + static_cast<size_t>(c) * fa.ldqk + head * (DK4 * 4);
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < DK4; ++j) {
        const float4 kv = *reinterpret_cast<const float4*>(kp + 4 * j);
        dot = fmaf(qv[4 * j + 0], kv.x, dot);
        dot = fmaf(qv[4 * j + 1], kv.y, dot);
        dot = fmaf(qv[4 * j + 2], kv.z, dot);
        dot = fmaf(qv[4 * j + 3], kv.w, dot);
      }
      float sv = dot * fa.inv_sqrt_dk;
      if (fa.edge_w != nullptr) sv *= fa.edge_w[base + idx];
      sc[p] = sv;
    }
  }
}

template <int H>
__device__ __forceinline__ float slots_max(float v) {
#pragma unroll
  for (int off = H; off < kWave; off <<= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
  return v;
}
template <int H>
__device__ __forceinline__ float slots_sum(float v) {
#pragma unroll
  for (int off = H; off < kWave; off <<= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// head-mean weights from the scores and the row statistics of this lane's head; returns the weight of entry `lane`
template <int H>
__device__ __forceinline__ float chunk_weights(const float (&sc)[H], float m, float inv_l, int lane) {
  constexpr int ES = kWave / H;
  float wv = 0.f;
#pragma unroll
  for (int p = 0; p < H; ++p) {
    float a = expf(sc[p] - m) * inv_l;                 // exp(-inf) = 0 for the absent entries
#pragma unroll
    for (int off = 1; off < H; off <<= 1) a += __shfl_xor(a, off, kWave);   // sum over the H lanes of the entry
    a *= (1.0f / static_cast<float>(H));
    // entry p * ES + slot lives in the lanes slot * H ..: lane e of the wave wants entry e
    const float got = __shfl(a, (lane % ES) * H, kWave);
    if (lane / ES == p) wv = got;
  }
  return wv;
}

template <int VEC, int L, int U, int H, int DK4, bool NT, bool FULL>
__global__ __launch_bounds__(kWave) void attn_spmm_kernel(const AttnSpmmArgs fa) {
  const SpmmArgs& a = fa.s;
  constexpr int G = kWave / L;
  const int lane = threadIdx.x & (kWave - 1);
  const int sub = lane / L, cl = lane % L;
  const int col = cl * VEC;
  const bool col_ok = FULL ? true : col < a.d;
  const int head = lane % H, slot = lane / H;
  const int xcd = static_cast<int>(blockIdx.x % kXcds);
  const int lw = static_cast<int>(blockIdx.x / kXcds);
  const Item it = item_of(a, xcd, lw);
  if (!it.valid) return;
  const int row = it.row, e0 = it.e0, e1 = it.e1, chunk = it.chunk;
  const size_t off = static_cast<size_t>(row) * a.ld + col;
  const bool pre_ok = stage_prefetchable<VEC, NT>(a.ep.stage);

  Pre<VEC, NT> pre;
  const bool do_pre = pre_ok && chunk < 0 && sub == 0 && col_ok;
  if (do_pre) {
    auto ldp = [](const float* q, float (&v)[VEC]) { if constexpr (NT) load_vec_nt<VEC>(q, v); else load_vec<VEC>(q, v); };
    load_vec<VEC>(a.u + off, pre.ui);
    if (a.ep.x0 != nullptr) ldp(a.ep.x0 + off, pre.x0);
    const int st = a.ep.stage;
    if (st == GNPDE_STAGE_EULER || st == GNPDE_STAGE_RK2C || st == GNPDE_STAGE_RK4C) ldp(a.ep.y + off, pre.y);
    if (st == GNPDE_STAGE_RK3C || st == GNPDE_STAGE_RK4C) ldp(a.ep.k1 + off, pre.k1);
  }

  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.0f;

  auto aggregate = [&](int cv, float wv, int cnt) {
    int t0 = 0;
    for (; t0 + G * U <= cnt; t0 += G * U) gather_batch<VEC, L, U, false, FULL>(a, cv, wv, t0, cnt, sub, col, col_ok, acc);
    if (t0 < cnt) gather_batch<VEC, L, U, true, FULL>(a, cv, wv, t0, cnt, sub, col, col_ok, acc);
  };

  if (chunk >= 0) {                               // hub chunk: weights were written by the hub launches
    for (int base = e0; base < e1; base += kWave) {
      const int me = base + lane;
      const bool in = me < e1;
      aggregate(in ? a.colidx[me] : 0, in ? a.w[me] : 0.0f, (e1 - base) < kWave ? (e1 - base) : kWave);
    }
  } else if (e1 > e0) {
    float qv[DK4 * 4];
    {
      const float* qp = fa.q + static_cast<size_t>(row) * fa.ldqk + head * (DK4 * 4);
#pragma unroll
      for (int j = 0; j < DK4; ++j) {
        const float4 t = *reinterpret_cast<const float4*>(qp + 4 * j);
        qv[4 * j] = t.x; qv[4 * j + 1] = t.y; qv[4 * j + 2] = t.z; qv[4 * j + 3] = t.w;
      }
    }
    float sc[H];
    if (e1 - e0 <= kWave) {                       // one chunk: scores stay in registers
      const int cnt = e1 - e0;
      const int cv = lane < cnt ? a.colidx[e0 + lane] : 0;
      chunk_scores<H, DK4>(fa, cv, e0, cnt, slot, head, qv, sc);
      float m = sc[0];
#pragma unroll
      for (int p = 1; p < H; ++p) m = fmaxf(m, sc[p]);
      m = slots_max<H>(m);
      float l = 0.f;
#pragma unroll
      for (int p = 0; p < H; ++p) l += expf(sc[p] - m);
      l = slots_sum<H>(l) + 1e-16f;
      aggregate(cv, chunk_weights<H>(sc, m, 1.0f / l, lane), cnt);
    } else {                                      // 65 .. 512 entries: online statistics, then recompute per chunk
      float m = -INFINITY, l = 0.f;
      for (int base = e0; base < e1; base += kWave) {
        const int cnt = (e1 - base) < kWave ? (e1 - base) : kWave;
        const int cv = lane < cnt ? a.colidx[base + lane] : 0;
        chunk_scores<H, DK4>(fa, cv, base, cnt, slot, head, qv, sc);
        float mc = sc[0];
#pragma unroll
        for (int p = 1; p < H; ++p) mc = fmaxf(mc, sc[p]);
        if (mc > m) {                             // (mc == -inf only for lanes without entries in this chunk)
          l *= expf(m - mc);
          m = mc;
        }
        if (m > -INFINITY) {
#pragma unroll
          for (int p = 0; p < H; ++p) l += expf(sc[p] - m);
        }
      }
      const float mrow = slots_max<H>(m);
      l = slots_sum<H>(m > -INFINITY ? l * expf(m - mrow) : 0.f) + 1e-16f;
      const float inv_l = 1.0f / l;
      for (int base = e0; base < e1; base += kWave) {
        const int cnt = (e1 - base) < kWave ? (e1 - base) : kWave;
        const int cv = lane < cnt ? a.colidx[base + lane] : 0;
        chunk_scores<H, DK4>(fa, cv, base, cnt, slot, head, qv, sc);
        aggregate(cv, chunk_weights<H>(sc, mrow, inv_l, lane), cnt);
      }
    }
  }

#pragma unroll
  for (int o = L; o < kWave; o <<= 1)
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] += __shfl_xor(acc[v], o, kWave);

  if (sub != 0 || !col_ok) return;
  if (chunk >= 0) {
    store_vec<VEC>(a.partial + static_cast<size_t>(chunk) * a.ldp + col, acc);
    return;
  }
  const float alpha = alpha_of(a.ep);
  const float beta = a.ep.x0 != nullptr ? *a.ep.beta : 0.0f;
  if (pre_ok) {
    epilogue_pre<VEC, NT>(a.ep, alpha, beta, off, acc, pre);
  } else {
    float ui[VEC];
    load_vec<VEC>(a.u + off, ui);
    epilogue<VEC, NT>(a.ep, alpha, beta, off, acc, ui);
  }
}

template <int H, int DK4>
bool launch_attn_spmm_hd(const AttnSpmmArgs& fa, hipStream_t st) {
  const SpmmArgs& a = fa.s;
  const int slots = (a.d + 3) / 4;
  const unsigned grid = balanced_grid(a, 1);
#define GNPDE_AS(LL)                                                                                                     \
  if (a.d == LL * 4) hipLaunchKernelGGL((attn_spmm_kernel<4, LL, 8, H, DK4, true, true>), dim3(grid), dim3(kWave), 0, st, fa); \
  else hipLaunchKernelGGL((attn_spmm_kernel<4, LL, 8, H, DK4, true, false>), dim3(grid), dim3(kWave), 0, st, fa);
  if (slots <= 16) return false;
  if (slots <= 32) { GNPDE_AS(32) return true; }
  if (slots <= 64) { GNPDE_AS(64) return true; }
#undef GNPDE_AS
  return false;
}

// resident grid for the persistent variants: every CU filled with the waves its registers admit
inline unsigned persistent_grid(int wpb) { return xcd_grid(256LL * 32 / wpb); }

template <int VEC, int L, int U, int BLK, bool PERSIST>
void launch_wide(const SpmmArgs& a, hipStream_t s) {
  constexpr int WPB = BLK / kWave;
  unsigned grid = balanced_grid(a, WPB);
  if (PERSIST && grid > persistent_grid(WPB)) grid = persistent_grid(WPB);
  if (a.d == L * VEC) hipLaunchKernelGGL((spmm_wide_kernel<VEC, L, U, BLK, PERSIST, true, true>), dim3(grid), dim3(BLK), 0, s, a);
  else hipLaunchKernelGGL((spmm_wide_kernel<VEC, L, U, BLK, PERSIST, true, false>), dim3(grid), dim3(BLK), 0, s, a);
}

// Software-pipelined, resident form of the row-pair kernel.  A wave of the one-item-per-wave kernels pays three dependent memory
// round trips per row (row pointer -> column ids / weights -> neighbour rows) and has data in flight during one of them only.
// Here a resident wave strides over the pairs of its XCD and keeps three pairs in different stages: while the neighbour rows of
// pair A are gathered, the column ids / weights of pair B and the row pointers of pair C are already on their way, so an
// iteration waits for ONE round trip.  Same per-row arithmetic and summation order as spmm_pair_kernel (bit-identical).
template <int VEC, int U, int BLK, bool NT, bool FULL>
__global__ __launch_bounds__(BLK) void spmm_pair_pipe_kernel(const SpmmArgs a) {
  constexpr int L = 32;
  constexpr int WPB = BLK / kWave;
  const int lane = threadIdx.x & (kWave - 1);
  const int half = lane >> 5;
  const int cl = lane & (L - 1);
  const int col = cl * VEC;
  const bool col_ok = FULL ? true : col < a.d;
  const int xcd = static_cast<int>(blockIdx.x % kXcds);
  const int lw = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x / kXcds) * WPB + static_cast<int>(threadIdx.x >> 6));
  const int wpx = static_cast<int>(gridDim.x / kXcds) * WPB;       // waves of this XCD
  const bool pre_ok = a.plain_out == nullptr && stage_prefetchable<VEC, NT>(a.ep.stage);
  const float alpha = a.plain_out == nullptr ? alpha_of(a.ep) : 0.0f;
  const float beta = (a.plain_out == nullptr && a.ep.x0 != nullptr) ? *a.ep.beta : 0.0f;

  const int cx = chunks_of_xcd(a, xcd);
  int j = lw;
  for (; j < cx; j += wpx) {   // hub chunks first
    const int chunk = a.chunk_begin + j * kXcds + xcd;
    const int row = __builtin_amdgcn_readfirstlane(a.lc_row[chunk]);
    const int e0 = __builtin_amdgcn_readfirstlane(a.lc_begin[chunk]);
    const int e1 = __builtin_amdgcn_readfirstlane(a.lc_end[chunk]);
    shared_row_item<VEC, U, NT, FULL>(a, row, e0, e1, chunk, lane, half, col, col_ok, pre_ok, alpha, beta);
  }
  const int per = rows_per_xcd(a);
  const int first = a.row_begin + xcd * per;
  int last = first + per;
  if (last > a.row_end) last = a.row_end;
  const int n_pairs = last > first ? (last - first + 1) / 2 : 0;
  int p = j - cx;
  if (p >= n_pairs) return;

  // stage 1: row pointers of a pair (half h: row first + 2 p + h); rows longer than GNPDE_LONG_ROW are chunk items
  auto header = [&](int pp, int& row, int& e0, int& len) {
    row = first + 2 * pp + half;
    const bool have = pp < n_pairs && row < last;
    int b = 0, e = 0;
    if (have) {
      b = a.rowptr[row];
      e = a.rowptr[row + 1];
    }
    e0 = b;
    len = (have && e - b <= GNPDE_LONG_ROW) ? e - b : -1;      // -1: nothing to do for this half
  };
  // stage 2: the row's first 32 column ids / weights, one coalesced load per half
  auto columns = [&](int e0, int len, int& cv, float& wv) {
    const bool in = cl < len;
    cv = in ? a.colidx[e0 + cl] : 0;
    wv = in ? a.w[e0 + cl] : 0.0f;
  };
  int row_a, e0_a, len_a, row_b, e0_b, len_b, row_c, e0_c, len_c;
  int cv_a, cv_b;
  float wv_a, wv_b;
  header(p, row_a, e0_a, len_a);
  columns(e0_a, len_a, cv_a, wv_a);
  header(p + wpx, row_b, e0_b, len_b);

  for (;;) {
    columns(e0_b, len_b, cv_b, wv_b);              // pair B: ids / weights on their way ...
    header(p + 2 * wpx, row_c, e0_c, len_c);       // pair C: row pointers on their way ...
    const int la = __builtin_amdgcn_readlane(len_a, 0), lb = __builtin_amdgcn_readlane(len_a, L);
    if (la > L || lb > L) {                        // a longer row in the pair: both halves share each row in turn
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int hl = h == 0 ? la : lb;
        if (hl < 0) continue;
        const int hr = __builtin_amdgcn_readlane(row_a, h * L), h0 = __builtin_amdgcn_readlane(e0_a, h * L);
        shared_row_item<VEC, U, NT, FULL>(a, hr, h0, h0 + hl, -1, lane, half, col, col_ok, pre_ok, alpha, beta);
      }
    } else {                                       // ... while pair A gathers its neighbour rows
      const size_t off = static_cast<size_t>(row_a) * a.ld + col;
      Pre<VEC, NT> pre;
      const bool mine = len_a >= 0 && col_ok;
      if (pre_ok && mine) {
        auto ldp = [](const float* q, float (&v)[VEC]) { if constexpr (NT) load_vec_nt<VEC>(q, v); else load_vec<VEC>(q, v); };
        load_vec<VEC>(a.u + off, pre.ui);
        if (a.ep.x0 != nullptr) ldp(a.ep.x0 + off, pre.x0);
        const int st = a.ep.stage;
        if (st == GNPDE_STAGE_EULER || st == GNPDE_STAGE_RK2C || st == GNPDE_STAGE_RK4C) ldp(a.ep.y + off, pre.y);
        if (st == GNPDE_STAGE_RK3C || st == GNPDE_STAGE_RK4C) ldp(a.ep.k1 + off, pre.k1);
      }
      const int cmax = la > lb ? la : lb;
      float acc0[VEC], acc1[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc0[v] = acc1[v] = 0.0f;
      for (int t0 = 0; t0 < cmax; t0 += U) {
        float vals[U][VEC];
        float ww[U];
#pragma unroll
        for (int t = 0; t < U; ++t) {
          const int idx = t0 + t;
          const int c = __shfl(cv_a, (half << 5) + idx, kWave);
          const float w = __shfl(wv_a, (half << 5) + idx, kWave);
          const bool ok = col_ok && idx < len_a;
          ww[t] = ok ? w : 0.0f;
#pragma unroll
          for (int v = 0; v < VEC; ++v) vals[t][v] = 0.0f;
          if (ok) load_vec<VEC>(a.u + static_cast<size_t>(c) * a.ld + col, vals[t]);
        }
#pragma unroll
        for (int t = 0; t < U; ++t) {
          if (t % 2 == 0) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc0[v] = fmaf(ww[t], vals[t][v], acc0[v]);
          } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc1[v] = fmaf(ww[t], vals[t][v], acc1[v]);
          }
        }
      }
      if (mine) {
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = acc0[v] + acc1[v];
        if (a.plain_out != nullptr) {
          store_vec<VEC>(a.plain_out + off, acc);
        } else if (pre_ok) {
          epilogue_pre<VEC, NT>(a.ep, alpha, beta, off, acc, pre);
        } else {
          float ui[VEC];
          load_vec<VEC>(a.u + off, ui);
          epilogue<VEC, NT>(a.ep, alpha, beta, off, acc, ui);
        }
      }
    }
    p += wpx;
    if (p >= n_pairs) break;
    row_a = row_b; e0_a = e0_b; len_a = len_b; cv_a = cv_b; wv_a = wv_b;
    row_b = row_c; e0_b = e0_c; len_b = len_c;
  }
}

// row pairs: one wave per two consecutive rows (hub chunks first, as balanced_grid)
template <int VEC, int U, int BLK>
void launch_pair(const SpmmArgs& a, hipStream_t s) {
  constexpr int WPB = BLK / kWave;
  const long long cn = a.chunk_end - a.chunk_begin, rn = a.row_end - a.row_begin;
  const long long per = xcd_rows_per(static_cast<int>(rn), a.row_shift);
  const long long per_xcd = (cn + kXcds - 1) / kXcds + (per + 1) / 2;
  long long blocks = (per_xcd + WPB - 1) / WPB;
  if (blocks < 1) blocks = 1;
  const unsigned grid = static_cast<unsigned>(blocks * kXcds);
  if (a.d == 32 * VEC) hipLaunchKernelGGL((spmm_pair_kernel<VEC, U, BLK, true, true>), dim3(grid), dim3(BLK), 0, s, a);
  else hipLaunchKernelGGL((spmm_pair_kernel<VEC, U, BLK, true, false>), dim3(grid), dim3(BLK), 0, s, a);
}

// A/B (gnpde_tune(0, 135)): the same kernel with ordinary (cached) loads and stores of the per-row streaming operands
template <int VEC, int U, int BLK>
void launch_pair_cached(const SpmmArgs& a, hipStream_t s) {
  constexpr int WPB = BLK / kWave;
  const long long cn = a.chunk_end - a.chunk_begin, rn = a.row_end - a.row_begin;
  const long long per = xcd_rows_per(static_cast<int>(rn), a.row_shift);
  const long long per_xcd = (cn + kXcds - 1) / kXcds + (per + 1) / 2;
  long long blocks = (per_xcd + WPB - 1) / WPB;
  if (blocks < 1) blocks = 1;
  const unsigned grid = static_cast<unsigned>(blocks * kXcds);
  if (a.d == 32 * VEC) hipLaunchKernelGGL((spmm_pair_kernel<VEC, U, BLK, false, true>), dim3(grid), dim3(BLK), 0, s, a);
  else hipLaunchKernelGGL((spmm_pair_kernel<VEC, U, BLK, false, false>), dim3(grid), dim3(BLK), 0, s, a);
}

// resident grid of the pipelined row-pair kernel: `waves_per_cu` waves on every CU, shrunk when there is less work than that
template <int VEC, int U, int BLK>
void launch_pair_pipe(const SpmmArgs& a, hipStream_t s, int waves_per_cu) {
  constexpr int WPB = BLK / kWave;
  const long long cn = a.chunk_end - a.chunk_begin, rn = a.row_end - a.row_begin;
  const long long per = (rn + kXcds - 1) / kXcds;
  const long long per_xcd = (cn + kXcds - 1) / kXcds + (per + 1) / 2;
  long long blocks = (per_xcd + WPB - 1) / WPB;             // per XCD, one item per wave
  const long long resident = 256LL * waves_per_cu / WPB / kXcds;
  if (blocks > resident) blocks = resident;
  if (blocks < 1) blocks = 1;
  const unsigned grid = static_cast<unsigned>(blocks * kXcds);
  SpmmArgs c = a;
  c.row_shift = -1;   // this (A/B only) kernel walks a contiguous range of row pairs per XCD
  if (a.d == 32 * VEC) hipLaunchKernelGGL((spmm_pair_pipe_kernel<VEC, U, BLK, true, true>), dim3(grid), dim3(BLK), 0, s, c);
  else hipLaunchKernelGGL((spmm_pair_pipe_kernel<VEC, U, BLK, true, false>), dim3(grid), dim3(BLK), 0, s, c);
}

// tune codes >= 100 (tools/spmm_ab.py): 100 + 10 * {0: U4, 1: U8, 2: U16} + {0: 256 thr, 1: 64 thr, 2: 256 thr persistent,
// 3: 64 thr persistent}
template <int VEC, int L>
bool dispatch_wide(const SpmmArgs& a, hipStream_t s, int code) {
  switch (code) {
    case 100: launch_wide<VEC, L, 4, 256, false>(a, s); return true;
    case 101: launch_wide<VEC, L, 4, 64, false>(a, s); return true;
    case 102: launch_wide<VEC, L, 4, 256, true>(a, s); return true;
    case 103: launch_wide<VEC, L, 4, 64, true>(a, s); return true;
    case 110: launch_wide<VEC, L, 8, 256, false>(a, s); return true;
    case 111: launch_wide<VEC, L, 8, 64, false>(a, s); return true;
    case 112: launch_wide<VEC, L, 8, 256, true>(a, s); return true;
    case 113: launch_wide<VEC, L, 8, 64, true>(a, s); return true;
    case 120: launch_wide<VEC, L, 16, 256, false>(a, s); return true;
    case 121: launch_wide<VEC, L, 16, 64, false>(a, s); return true;
    case 122: launch_wide<VEC, L, 16, 256, true>(a, s); return true;
    case 123: launch_wide<VEC, L, 16, 64, true>(a, s); return true;
    default: return false;
  }
}

template <int VEC, int L, int K, int U, bool NTI, bool NT = NTI>
void launch_rows(const SpmmArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((spmm_rows_kernel<VEC, L, K, U, NTI, NT>), dim3(balanced_grid(a, kWavesPerBlock)), dim3(kBlock), 0, s, a);
}

template <int VEC, int L, int K, int U>
void launch_rows_b64(const SpmmArgs& a, hipStream_t s) {   // one wavefront per workgroup (A/B: intra-block imbalance)
  hipLaunchKernelGGL((spmm_rows_kernel<VEC, L, K, U, false, true, 64>), dim3(balanced_grid(a, 1)), dim3(64), 0, s, a);
}

template <int VEC>
int dispatch_rows(const SpmmArgs& a, hipStream_t s) {
  const int slots = (a.d + VEC - 1) / VEC;
  int code = g_tune[GNPDE_TUNE_SPMM_VARIANT];
  // default for 16-byte lanes and rows of 17..64 lanes (d = 68..256): the wide-row kernel, 8 gathers in flight, one
  // wavefront per workgroup -- measured best at the ogbn-arxiv shape (178 vs 221 us) and within 1 % of the best at the
  // R-MAT d = 256 shape (16.6 vs 16.8 ms), tools/spmm_ab.py, profiles/r02_ab_*.log
  // rows of 17..32 lanes (d = 68..128) in graphs whose rows are mostly short (the ogbn-arxiv shape: median 8 entries): two rows
  // per wave, 16 gathers in flight per half (spmm_pair_kernel; bit-identical results): 171 vs 184 us at the ogbn-arxiv shape,
  // 193 vs 230 us on a uniform degree-8 graph, but 2 % slower at degree 24 -- hence the hint (profiles/r02_ab_row_pairs.txt)
  if (code == 0 && VEC == 4 && slots > 16 && slots <= 32 && a.short_rows) code = 133;
  if (code == 0 && VEC == 4 && slots > 16 && slots <= 64) code = 111;
  if (code == 99) code = 0;   // A/B: force the round-1 kernel
  if (code >= 130 && code <= 149 && VEC == 4 && slots > 16 && slots <= 32) {   // row pairs (A/B)
    if constexpr (VEC == 4) {
      switch (code) {
        case 140: launch_pair_pipe<4, 8, 64>(a, s, 16); return 0;
        case 141: launch_pair_pipe<4, 16, 64>(a, s, 12); return 0;
        case 142: launch_pair_pipe<4, 8, 256>(a, s, 16); return 0;
        case 143: launch_pair_pipe<4, 16, 256>(a, s, 12); return 0;
        case 144: launch_pair_pipe<4, 8, 64>(a, s, 20); return 0;
        case 145: launch_pair_pipe<4, 16, 64>(a, s, 8); return 0;
        case 146: launch_pair_pipe<4, 8, 64>(a, s, 32); return 0;
        case 147: launch_pair_pipe<4, 16, 64>(a, s, 16); return 0;
        case 130: launch_pair<4, 8, 64>(a, s); return 0;
        case 131: launch_pair<4, 8, 256>(a, s); return 0;
        case 132: launch_pair<4, 4, 64>(a, s); return 0;
        case 133: launch_pair<4, 16, 64>(a, s); return 0;
        case 134: launch_pair<4, 4, 256>(a, s); return 0;
        case 135: launch_pair_cached<4, 16, 64>(a, s); return 0;
        default: launch_pair<4, 16, 256>(a, s); return 0;
      }
    }
  }
  if (code >= 100) {
    bool done = false;
    if (slots <= 16) done = dispatch_wide<VEC, 16>(a, s, code);
    else if (slots <= 32) done = dispatch_wide<VEC, 32>(a, s, code);
    else if (slots <= 64) done = dispatch_wide<VEC, 64>(a, s, code);
    if (done) return 0;
  }
  if (code == 50) {
    if (slots <= 32 && slots > 16) { launch_rows_b64<VEC, 16, 2, 4>(a, s); return 0; }
    if (slots <= 64 && slots > 32) { launch_rows_b64<VEC, 32, 2, 4>(a, s); return 0; }
  }
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
  int row_shift;         // rows -> XCDs as in the aggregation (xcd_row_of): < 0 contiguous eighths, else hashed blocks of 2^row_shift rows
};

template <int VEC, int L, int K, int U>
__global__ __launch_bounds__(kBlock) void sddmm_kernel(const SddmmArgs s) {
  constexpr int G = kWave / L;
  const int lane = threadIdx.x & (kWave - 1);
  // same XCD-balanced work list as the aggregation (item_of): every 8th chunk first, then this XCD's share of the rows
  const int x = static_cast<int>(blockIdx.x % kXcds);
  const int lw = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x / kXcds) * kWavesPerBlock + static_cast<int>(threadIdx.x >> 6));
  const int cx = (s.n_long_chunks + kXcds - 1 - x) / kXcds;
  int row, e0, e1;
  if (lw >= cx) {
    row = xcd_row_of(0, s.n, s.row_shift, x, lw - cx);
    if (row < 0) return;
    e0 = s.rowptr[row];
    e1 = s.rowptr[row + 1];
    if (e1 - e0 > GNPDE_LONG_ROW) return;  // processed as chunks
  } else {
    const int item = lw * kXcds + x;
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
  s.row_shift = choose_row_shift(g->n, g->xcd_deal);
  const long long per_xcd = (static_cast<long long>(g->n_long_chunks) + kXcds - 1) / kXcds + xcd_rows_per(g->n, s.row_shift);
  const unsigned grid = static_cast<unsigned>(((per_xcd + kWavesPerBlock - 1) / kWavesPerBlock) * kXcds);
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

// ------------------------------------------------------------------------------------------------
// One stage of the native adjoint solve, row side (csrc/adjoint.hip): the aggregation F = alpha (A u - u) + beta x0 with its
// LINCOMB stage epilogue, the SDDMM r_e = g[row] . u[col] of the SAME gathered rows, and the two dot products the scalar
// gradients need (sum g . F, sum g . x0) -- what ran as three kernels, two of which gathered every neighbour row
// (spmm_pair_kernel 187 us + sddmm_kernel 181 us + a 53-us streaming pass at the ogbn-arxiv shape).  Work items and summation
// order of the aggregation as in spmm_wide_kernel (L lanes of 16 bytes per neighbour row, G = 64 / L neighbour slots, U gathers
// in flight, slots combined by the xor butterfly).  The dot of a gathered row with g_i needs a sum over the L lanes of its slot;
// the U partial dots of a batch are reduced TOGETHER by a transposing butterfly -- at every step a lane keeps half of its values
// and hands the other half to its partner, so U values cost U - 1 + log2(L / U) shuffles instead of U log2(L), and lane
// cl ends up with the dot of entry cl / (L / U) of the batch.  r is complete per entry, so hub chunks need no second pass for it.
// ------------------------------------------------------------------------------------------------
struct AdjRowsArgs {
  SpmmArgs s;                      // graph, w, u (the state-side stage input), LINCOMB epilogue, hub-chunk partials
  const float* __restrict__ g;     // [n, ld] adjoint-side stage input
  float* __restrict__ r;           // [e] out
  float* __restrict__ dots;        // [gridDim.x + n_long_rows][2] out: per-wave sums of g . F and g . x0
  int r_accumulate;                // != 0: r[e] += (every entry belongs to exactly one lane of one launch: no race); the recorded
                                   // dopri5 backward sums the edge products of all its evaluations this way
  float r_scale;                   // weight of this launch's products in the sum (the recorded fixed-grid backward: the stage's b_j h)
  const int* __restrict__ wpos;    // or null: the weight of entry p is w[wpos[p]] (weights stored in ANOTHER graph's order: the cotangent-side
                                   // sweep runs on the transposed graph with the weights as the attention wrote them, no permutation pass)
};

template <int N, int MASK>
struct TransposeReduce {   // N values per lane -> 1, over the lane pairs (cl ^ MASK), then recursively MASK / 2
  static __device__ __forceinline__ void run(float* p, int cl) {
    constexpr int H = N / 2;
    const bool upper = (cl & MASK) != 0;
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const float send = upper ? p[j] : p[j + H];
      const float keep = upper ? p[j + H] : p[j];
      p[j] = keep + __shfl_xor(send, MASK, kWave);
    }
    if constexpr (H > 1) TransposeReduce<H, MASK / 2>::run(p, cl);
  }
};

// k = alpha (ax - u_i) + beta x0_i: this lane's share of g_i . k and g_i . x0_i, then the stage algebra of the shared epilogue
// (compact rk4 / euler / LINCOMB: epilogue.h; it forms k by the same expression, the source row comes from L1 the second time)
template <int VEC>
__device__ __forceinline__ void adjoint_epilogue(const gnpde_epilogue_t& ep, float alpha, float beta, size_t off, const float (&ax)[VEC],
                                                 const float (&ui)[VEC], const float (&gi)[VEC], const float (&cmask)[VEC],
                                                 float& d1, float& d2) {
  float k[VEC], s[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) { k[v] = alpha * (ax[v] - ui[v]); s[v] = 0.f; }
  if (ep.x0 != nullptr) {
    load_vec<VEC>(ep.x0 + off, s);
#pragma unroll
    for (int v = 0; v < VEC; ++v) k[v] = k[v] + beta * s[v];
  }
  // d1: with alpha' = sigmoid(alpha_train) the dot with k (d alpha_train = (1 - sigma)(sum g . k - beta sum g . x0)); with the raw
  // alpha (opt['no_alpha_sigmoid']) the dot with A u - u itself, which IS d k / d alpha_train
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    d1 = fmaf(gi[v] * cmask[v], ep.alpha_sigmoid ? k[v] : ax[v] - ui[v], d1);
    d2 = fmaf(gi[v] * cmask[v], s[v], d2);
  }
  epilogue<VEC, true>(ep, alpha, beta, off, ax, ui);
}

// one work item (a row, or a 512-entry chunk of a hub row) by a whole wavefront: G = 64 / L neighbour slots share the row
template <int L, int U>
__device__ __forceinline__ void adjoint_item(const AdjRowsArgs& fa, int row, int e0, int e1, int chunk, int lane, float& d1, float& d2) {
  constexpr int VEC = 4;
  constexpr int G = kWave / L;
  constexpr int EPB = G * U;          // entries per batch
  constexpr int LPE = L / U;          // lanes that end up with the same entry's dot
  static_assert(U <= L && (L % U) == 0, "transposing butterfly needs U <= L");
  const SpmmArgs& a = fa.s;
  const int sub = lane / L, cl = lane % L;
  const int col = cl * VEC;
  const bool col_ok = col < a.d;
  const size_t off = static_cast<size_t>(row) * a.ld + col;
  float gi[VEC], ui[VEC], cmask[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) { gi[v] = 0.f; ui[v] = 0.f; cmask[v] = col + v < a.d ? 1.0f : 0.0f; }   // padded rows: columns [d, ld) are not data
  if (col_ok) {
    load_vec<VEC>(fa.g + off, gi);
    if (chunk < 0 && sub == 0) load_vec<VEC>(a.u + off, ui);
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) gi[v] *= cmask[v];
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.0f;
  for (int base = e0; base < e1; base += kWave) {
    const int me = base + lane;
    const bool in = me < e1;
    const int cv = in ? a.colidx[me] : 0;          // one coalesced load of 64 column ids and weights per wave
    const float wv = in ? a.w[fa.wpos != nullptr ? fa.wpos[me] : me] : 0.0f;
    const int cnt = (e1 - base) < kWave ? (e1 - base) : kWave;
    for (int t0 = 0; t0 < cnt; t0 += EPB) {
      float vals[U][VEC], ww[U], p[U];
      int cs[U];
      bool oks[U];
#pragma unroll
      for (int t = 0; t < U; ++t) {
        const int idx = t0 + t * G + sub;
        cs[t] = __shfl(cv, idx & (kWave - 1), kWave);
        const float w = __shfl(wv, idx & (kWave - 1), kWave);
        oks[t] = col_ok && idx < cnt;
        ww[t] = oks[t] ? w : 0.0f;
      }
#pragma unroll
      for (int t = 0; t < U; ++t) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) vals[t][v] = 0.0f;
        if (oks[t]) load_vec<VEC>(a.u + static_cast<size_t>(cs[t]) * a.ld + col, vals[t]);
      }
#pragma unroll
      for (int t = 0; t < U; ++t) {
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          acc[v] = fmaf(ww[t], vals[t][v], acc[v]);
          q = fmaf(gi[v], vals[t][v], q);
        }
        p[t] = q;
      }
      TransposeReduce<U, L / 2>::run(p, cl);
#pragma unroll
      for (int m = LPE / 2; m >= 1; m >>= 1) p[0] += __shfl_xor(p[0], m, kWave);
      const int idx = t0 + (cl / LPE) * G + sub;
      if ((cl % LPE) == 0 && idx < cnt) {
        if (fa.r_accumulate) fa.r[base + idx] = fmaf(fa.r_scale, p[0], fa.r[base + idx]);
        else fa.r[base + idx] = p[0];
      }
    }
  }
#pragma unroll
  for (int o = L; o < kWave; o <<= 1)
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] += __shfl_xor(acc[v], o, kWave);
  if (sub == 0 && col_ok) {
    if (chunk >= 0) {
      store_vec<VEC>(a.partial + static_cast<size_t>(chunk) * a.ldp + col, acc);   // adjoint_long_reduce4_kernel folds the chunks of a row
    } else {
      const float alpha = alpha_of(a.ep);
      const float beta = a.ep.x0 != nullptr ? *a.ep.beta : 0.0f;
      adjoint_epilogue<VEC>(a.ep, alpha, beta, off, acc, ui, gi, cmask, d1, d2);
    }
  }
}

__device__ __forceinline__ void write_wave_dots(const AdjRowsArgs& fa, int lane, float d1, float d2) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    d1 += __shfl_xor(d1, o, kWave);
    d2 += __shfl_xor(d2, o, kWave);
  }
  if (lane == 0) {
    fa.dots[2 * static_cast<size_t>(blockIdx.x)] = d1;
    fa.dots[2 * static_cast<size_t>(blockIdx.x) + 1] = d2;
  }
}

template <int L, int U>
__global__ __launch_bounds__(kWave) void adjoint_rows_kernel(const AdjRowsArgs fa) {
  const SpmmArgs& a = fa.s;
  const int lane = threadIdx.x;
  const int xcd = static_cast<int>(blockIdx.x % kXcds);
  const int lw = static_cast<int>(blockIdx.x / kXcds);
  const Item it = item_of(a, xcd, lw);
  float d1 = 0.f, d2 = 0.f;
  if (it.valid) adjoint_item<L, U>(fa, it.row, it.e0, it.e1, it.chunk, lane, d1, d2);
  write_wave_dots(fa, lane, d1, d2);
}

// hub rows of the adjoint stage: chunk partials in chunk order (as spmm_long_reduce4_kernel), then the same epilogue and dots
__global__ __launch_bounds__(kWave) void adjoint_long_reduce4_kernel(const AdjRowsArgs fa, const int* __restrict__ long_rows,
                                                                    const int* __restrict__ long_chunk_ptr, int dots_base) {
  constexpr int VEC = 4;
  const SpmmArgs& a = fa.s;
  const int lr = blockIdx.x;
  const int row = long_rows[lr];
  const int c0 = long_chunk_ptr[lr], c1 = long_chunk_ptr[lr + 1];
  const float alpha = alpha_of(a.ep);
  const float beta = a.ep.x0 != nullptr ? *a.ep.beta : 0.0f;
  float d1 = 0.f, d2 = 0.f;
  for (int col = static_cast<int>(threadIdx.x) * VEC; col < a.d; col += kWave * VEC) {
    const size_t off = static_cast<size_t>(row) * a.ld + col;
    float gi[VEC], ui[VEC], cmask[VEC];
    load_vec<VEC>(fa.g + off, gi);
    load_vec<VEC>(a.u + off, ui);
#pragma unroll
    for (int v = 0; v < VEC; ++v) cmask[v] = col + v < a.d ? 1.0f : 0.0f;
    float acc[VEC] = {0.0f, 0.0f, 0.0f, 0.0f};
    const float* p = a.partial + static_cast<size_t>(c0) * a.ldp + col;
    int c = c0;
    for (; c + 8 <= c1; c += 8, p += 8 * static_cast<size_t>(a.ldp)) {
      float v[8][VEC];
#pragma unroll
      for (int t = 0; t < 8; ++t) load_vec<VEC>(p + t * static_cast<size_t>(a.ldp), v[t]);
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] += v[t][q];
    }
    for (; c < c1; ++c, p += a.ldp) {
      float v[VEC];
      load_vec<VEC>(p, v);
#pragma unroll
      for (int q = 0; q < VEC; ++q) acc[q] += v[q];
    }
    adjoint_epilogue<VEC>(a.ep, alpha, beta, off, acc, ui, gi, cmask, d1, d2);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    d1 += __shfl_xor(d1, o, kWave);
    d2 += __shfl_xor(d2, o, kWave);
  }
  if (threadIdx.x == 0) {
    fa.dots[2 * (static_cast<size_t>(dots_base) + lr)] = d1;
    fa.dots[2 * (static_cast<size_t>(dots_base) + lr) + 1] = d2;
  }
}

}  // namespace

namespace {
// grid of the adjoint row kernel: one wavefront per work item.  (A row-PAIR form as spmm_pair_kernel -- two rows of <= 32 entries per
// wave, 16 gathers in flight per half -- was built and measured SLOWER at the ogbn-arxiv shape, 263 vs 248 us: with the 16 partial
// dots and g_i next to the 16 gathered float4s it needs 140 VGPRs, three waves per SIMD against five here; profiles/r04_train_*.)
inline unsigned adjoint_rows_grid(const gnpde_graph_t* g) {
  SpmmArgs a{};
  a.chunk_begin = 0; a.chunk_end = g->n_long_chunks; a.row_begin = 0; a.row_end = g->n;
  a.row_shift = choose_row_shift(g->n, g->xcd_deal);
  return balanced_grid(a, 1);
}
}  // namespace

// number of per-wave dot slots launch_adjoint_rows writes (its grid + one per long row)
int adjoint_rows_dot_slots(const gnpde_graph_t* g, int d) {
  (void)d;
  return static_cast<int>(adjoint_rows_grid(g)) + g->n_long_rows;
}

int launch_adjoint_rows(const gnpde_graph_t* g, const float* w_csr, const float* u, const float* gvec, int d, int ld,
                        const gnpde_epilogue_t* epi, float* r_out, float* dots, void* ws, size_t ws_bytes, hipStream_t stream,
                        bool padded_rows, bool accumulate_r, float r_scale, const int* wpos) {
  GNPDE_CHECK_ARG(g && u && gvec && epi && r_out && dots && (w_csr || g->e == 0), GNPDE_EINVAL, "adjoint_rows: null pointer");
  const int stg = epi->stage;
  GNPDE_CHECK_ARG((stg == GNPDE_STAGE_LINCOMB || stg == GNPDE_STAGE_EULER || (stg >= GNPDE_STAGE_RK1C && stg <= GNPDE_STAGE_RK4C)) &&
                  epi->alpha != nullptr && (epi->x0 == nullptr || epi->beta != nullptr), GNPDE_EINVAL,
                  "adjoint_rows: needs a LINCOMB, euler or compact rk4 epilogue");
  GNPDE_CHECK_ARG(stg != GNPDE_STAGE_LINCOMB || (epi->n_prev >= 0 && epi->n_prev <= GNPDE_MAX_PREV && (epi->out_y == nullptr || epi->y != nullptr)),
                  GNPDE_EINVAL, "adjoint_rows: bad LINCOMB epilogue");
  GNPDE_CHECK_ARG(stg == GNPDE_STAGE_LINCOMB || epi->out_y != nullptr, GNPDE_EINVAL, "adjoint_rows: out_y is null");
  GNPDE_CHECK_ARG(!(stg == GNPDE_STAGE_EULER || stg == GNPDE_STAGE_RK2C || stg == GNPDE_STAGE_RK4C) || epi->y != nullptr, GNPDE_EINVAL, "adjoint_rows: y is null");
  GNPDE_CHECK_ARG(!(stg == GNPDE_STAGE_RK3C || stg == GNPDE_STAGE_RK4C) || epi->k1 != nullptr, GNPDE_EINVAL, "adjoint_rows: k1 is null");
  GNPDE_CHECK_ARG(epi->out_k != u && epi->out_y != u, GNPDE_EINVAL, "adjoint_rows: output aliases the gathered operand");
  GNPDE_CHECK_ARG(g->row_begin == 0 && d >= 1 && d <= 256 && ld >= d && ld % 4 == 0 && (d % 4 == 0 || padded_rows), GNPDE_ESHAPE,
                  "adjoint_rows: whole graphs, rows of up to 256 floats in 16-byte lanes");
  if (g->n == 0) return 0;
  AdjRowsArgs fa{};
  SpmmArgs& a = fa.s;
  a.n = g->n; a.n_long_chunks = g->n_long_chunks;
  a.rowptr = g->rowptr; a.colidx = g->colidx;
  a.lc_row = g->long_chunk_row; a.lc_begin = g->long_chunk_begin; a.lc_end = g->long_chunk_end;
  a.w = w_csr; a.u = u; a.d = d; a.ld = ld; a.plain_out = nullptr;
  a.ldp = static_cast<int>(align_up(static_cast<size_t>(d), 4));
  a.partial = static_cast<float*>(ws);
  a.ep = *epi;
  a.chunk_begin = 0; a.chunk_end = g->n_long_chunks; a.row_begin = 0; a.row_end = g->n;
  a.row_shift = choose_row_shift(g->n, g->xcd_deal);
  if (g->n_long_chunks > 0) {
    const size_t need = static_cast<size_t>(g->n_long_chunks) * a.ldp * sizeof(float);
    GNPDE_CHECK_ARG(ws != nullptr && ws_bytes >= need, GNPDE_EWS, "adjoint_rows: workspace %zu < %zu bytes", ws_bytes, need);
  }
  const void* ptrs[] = {u, gvec, ws, epi->x0, epi->y, epi->k1, epi->out_k, epi->out_y, stg == GNPDE_STAGE_LINCOMB ? epi->prev[0] : nullptr,
                        stg == GNPDE_STAGE_LINCOMB ? epi->prev[1] : nullptr, stg == GNPDE_STAGE_LINCOMB ? epi->prev[2] : nullptr};
  for (const void* p : ptrs) GNPDE_CHECK_ARG(aligned(p, 16), GNPDE_EINVAL, "adjoint_rows: operands must be 16-byte aligned");
  fa.g = gvec; fa.r = r_out; fa.dots = dots; fa.r_accumulate = accumulate_r ? 1 : 0; fa.r_scale = r_scale; fa.wpos = wpos;
  const unsigned grid = adjoint_rows_grid(g);
  const int slots = (d + 3) / 4;
  if (slots <= 16) hipLaunchKernelGGL((adjoint_rows_kernel<16, 8>), dim3(grid), dim3(kWave), 0, stream, fa);
  else if (slots <= 32) hipLaunchKernelGGL((adjoint_rows_kernel<32, 8>), dim3(grid), dim3(kWave), 0, stream, fa);
  else hipLaunchKernelGGL((adjoint_rows_kernel<64, 8>), dim3(grid), dim3(kWave), 0, stream, fa);
  GNPDE_LAUNCH_CHECK();
  if (g->n_long_rows > 0) {
    hipLaunchKernelGGL(adjoint_long_reduce4_kernel, dim3(g->n_long_rows), dim3(kWave), 0, stream, fa, g->long_rows, g->long_chunk_ptr,
                       static_cast<int>(grid));
    GNPDE_LAUNCH_CHECK();
  }
  return 0;
}

namespace {
}  // namespace

int launch_spmm_rhs(const gnpde_graph_t* g, const float* w_csr, const float* u, int d, int ld,
                    const gnpde_epilogue_t* epi, float* plain_out, void* ws, size_t ws_bytes, hipStream_t stream,
                    const Fork* fork, bool padded_rows) {
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
  // padded_rows: the caller guarantees that columns [d, ld) of every operand are padding (GNPDE_RHS_PADDED_ROWS), so a
  // width that is not a multiple of 4 still takes 16-byte lanes (the last lane reads / writes up to 3 padding columns)
  bool a16 = (d % 4 == 0 || padded_rows) && (ld % 4 == 0) && aligned(u, 16) && aligned(plain_out, 16) && aligned(ws, 16);
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
    if (arg.chunk_end <= arg.chunk_begin && arg.row_end <= arg.row_begin) return 0;
    if (a16) return dispatch_rows<4>(arg, st);
    if (a8) return dispatch_rows<2>(arg, st);
    return dispatch_rows<1>(arg, st);
  };
  const bool forked = fork != nullptr && fork->aux != nullptr && g->n_long_rows > 0;
  // rows and long-row chunks in one launch, then the per-row reduction of the chunk partials (a last-arriver
  // reduction inside the kernel was measured 1.7x slower: every agent-scope release fence writes back the
  // XCD's dirty L2 lines); with a fork the chunks + reduction run as a parallel branch.  The boundary pass of a
  // partitioned graph covers rows [row_begin, n) only (its chunks belong to those rows).
  GNPDE_CHECK_ARG(g->row_begin >= 0 && g->row_begin <= g->n, GNPDE_EINVAL, "spmm: bad row_begin %d", g->row_begin);
  GNPDE_CHECK_ARG(!(forked && g->row_begin > 0), GNPDE_EINVAL, "spmm: fork with row_begin");
  a.chunk_begin = 0;
  a.chunk_end = forked ? 0 : g->n_long_chunks;
  a.short_rows = 2LL * g->n_bin16 >= g->n ? 1 : 0;
  a.row_begin = g->row_begin;
  a.row_end = g->n;
  a.row_shift = choose_row_shift(static_cast<long long>(a.row_end) - a.row_begin, g->xcd_deal);
  if (g_tune[GNPDE_TUNE_SPMM_PART] == 1) a.row_end = a.row_begin;       // (timing the two kinds of work items separately)
  if (g_tune[GNPDE_TUNE_SPMM_PART] == 2) a.chunk_end = 0;
  int rc = run(a, stream);
  if (rc != 0) return rc;
  GNPDE_LAUNCH_CHECK();
  if (g->n_long_rows > 0) {
    hipStream_t br = stream;
    if (forked) { const int frc = fork_begin(fork, stream, &br); if (frc) return frc; }
    if (forked) {
      SpmmArgs c = a;
      c.chunk_end = g->n_long_chunks;
      c.row_begin = c.row_end = 0;
      rc = run(c, br);
      if (rc != 0) return rc;
      GNPDE_LAUNCH_CHECK();
    }
    if (a16)
      hipLaunchKernelGGL(spmm_long_reduce4_kernel, dim3(g->n_long_rows), dim3(kWave), 0, br, a, g->long_rows, g->long_chunk_ptr);
    else
      hipLaunchKernelGGL(spmm_long_reduce_kernel, dim3(g->n_long_rows), dim3(kBlock), 0, br, a, g->long_rows,
                         g->long_chunk_ptr);
    GNPDE_LAUNCH_CHECK();
    if (forked) { const int frc = fork_end(fork, stream, br); if (frc) return frc; }
  }
  return 0;
}


bool attn_spmm_supported(const gnpde_graph_t* g, const gnpde_attention_t& at, int d, int ld, const float* u,
                         const gnpde_epilogue_t& e) {
  // Row attention inside the aggregation kernel: measured SLOWER than the separate kernels where the kernels are bound by memory
  // (ogbn-arxiv and up: DESIGN.md section 4) -- opt-in there (gnpde_tune(6, 1)); where an evaluation is bound by its LAUNCHES (a
  // state that fits one XCD's L2: Cora, 0.87 MB) one launch instead of three is the point: 7.8 us vs 5.2 + 5.2 (+ boundaries),
  // profiles/r04_cora_*.  gnpde_tune(6, 2) keeps the separate kernels there too (A/B).
  const bool small = static_cast<long long>(g->n) * ld * 4 <= (4LL << 20) && g->n_long_rows == 0;
  if (!(g_tune[GNPDE_TUNE_ROW_FUSION] == 1 || (small && g_tune[GNPDE_TUNE_ROW_FUSION] != 2))) return false;
  if (at.type != GNPDE_ATT_SCALED_DOT || at.norm_idx != 0 || at.square_plus) return false;
  const int dk = at.att_dim / at.heads;
  if (!((at.heads == 4 && dk == 4) || (at.heads == 8 && dk == 16) || (at.heads == 4 && dk == 16) || (at.heads == 2 && dk == 16)))
    return false;
  const int slots = (d + 3) / 4;
  if (ld % 4 != 0 || slots <= 16 || slots > 64 || g->row_begin != 0) return false;
  if (e.stage == GNPDE_STAGE_LINCOMB) {
    for (int j = 0; j < e.n_prev; ++j) if (!aligned(e.prev[j], 16)) return false;
  }
  const void* ptrs[] = {u, e.x0, e.y, e.k1, e.k2, e.k3, e.out_k, e.out_y};
  for (const void* p : ptrs) if (!aligned(p, 16)) return false;
  return true;
}

// rows: attention + aggregation in one kernel; hub chunks aggregate with the weights in w_hub_csr (written before by
// launch_hub_attention); then the per-row fold of the chunk partials
int launch_attn_spmm(const gnpde_graph_t* g, const gnpde_attention_t* at, const float* w_hub_csr, const float* u, int d, int ld,
                     const gnpde_epilogue_t* epi, void* ws, size_t ws_bytes, hipStream_t stream, bool padded_rows) {
  GNPDE_CHECK_ARG(g && at && u && epi && at->q && at->k, GNPDE_EINVAL, "attn_spmm: null argument");
  GNPDE_CHECK_ARG(d % 4 == 0 || padded_rows, GNPDE_ESHAPE, "attn_spmm: width %d needs padded rows", d);
  GNPDE_CHECK_ARG(epi->alpha != nullptr && (epi->x0 == nullptr || epi->beta != nullptr), GNPDE_EINVAL, "attn_spmm: bad epilogue");
  GNPDE_CHECK_ARG(epi->out_k != u && epi->out_y != u, GNPDE_EINVAL, "attn_spmm: output aliases the gathered operand");
  if (g->n == 0) return 0;
  AttnSpmmArgs fa{};
  SpmmArgs& a = fa.s;
  a.n = g->n; a.n_long_chunks = g->n_long_chunks;
  a.rowptr = g->rowptr; a.colidx = g->colidx;
  a.lc_row = g->long_chunk_row; a.lc_begin = g->long_chunk_begin; a.lc_end = g->long_chunk_end;
  a.w = w_hub_csr; a.u = u; a.d = d; a.ld = ld; a.plain_out = nullptr;
  a.ldp = static_cast<int>(align_up(static_cast<size_t>(d), 4));
  a.partial = static_cast<float*>(ws);
  a.ep = *epi;
  a.chunk_begin = 0; a.chunk_end = g->n_long_chunks; a.row_begin = 0; a.row_end = g->n;
  a.row_shift = choose_row_shift(g->n, g->xcd_deal);
  if (g->n_long_chunks > 0) {
    const size_t need = static_cast<size_t>(g->n_long_chunks) * a.ldp * sizeof(float);
    GNPDE_CHECK_ARG(ws != nullptr && ws_bytes >= need && w_hub_csr != nullptr, GNPDE_EWS, "attn_spmm: workspace %zu < %zu bytes", ws_bytes, need);
  }
  fa.q = at->q; fa.k = at->k; fa.ldqk = at->ldqk;
  const int dk = at->att_dim / at->heads;
  fa.inv_sqrt_dk = 1.0f / sqrtf(static_cast<float>(dk));
  fa.edge_w = at->edge_w_csr;
  GNPDE_CHECK_ARG(at->ldqk % 4 == 0 && aligned(at->q, 16) && aligned(at->k, 16), GNPDE_EINVAL, "attn_spmm: q / k must be 16-byte aligned rows");
  bool ok = false;
  if (at->heads == 4 && dk == 4) ok = launch_attn_spmm_hd<4, 1>(fa, stream);
  else if (at->heads == 8 && dk == 16) ok = launch_attn_spmm_hd<8, 4>(fa, stream);
  else if (at->heads == 4 && dk == 16) ok = launch_attn_spmm_hd<4, 4>(fa, stream);
  else if (at->heads == 2 && dk == 16) ok = launch_attn_spmm_hd<2, 4>(fa, stream);
  GNPDE_CHECK_ARG(ok, GNPDE_ESHAPE, "attn_spmm: configuration not covered");
  GNPDE_LAUNCH_CHECK();
  if (g->n_long_rows > 0) {
    hipLaunchKernelGGL(spmm_long_reduce_kernel, dim3(g->n_long_rows), dim3(kBlock), 0, stream, a, g->long_rows, g->long_chunk_ptr);
    GNPDE_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace gnpde

extern "C" int gnpde_xcd_row_map(int32_t row_begin, int32_t row_end, int32_t deal, int32_t* row_shift, int32_t* rows_per_xcd,
                                 int32_t* map) {
  GNPDE_CHECK_ARG(row_begin >= 0 && row_end >= row_begin && row_shift && rows_per_xcd &&
                  (deal == GNPDE_XCD_CONTIGUOUS || deal == GNPDE_XCD_HASHED), GNPDE_EINVAL, "xcd_row_map: bad arguments");
  const int shift = gnpde::choose_row_shift(static_cast<long long>(row_end) - row_begin, deal);
  const int per = gnpde::xcd_rows_per(row_end - row_begin, shift);
  *row_shift = shift;
  *rows_per_xcd = per;
  if (map != nullptr)
    for (int x = 0; x < gnpde::kXcds; ++x)
      for (int r = 0; r < per; ++r) map[static_cast<size_t>(x) * per + r] = gnpde::xcd_row_of(row_begin, row_end, shift, x, r);
  return 0;
}

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
