// GRAND-nl right-hand side in ONE pass over the graph:  scaled-dot multi-head edge attention with a
// softmax over every row, head-mean aggregation  A(x) x,  diffusion epilogue and solver stage algebra.
//
// Replaces, per evaluation, the whole of ODEFuncTransformerAtt.forward (reference
// src/function_transformer_attention.py:38-53): the Q/K Linear layers (:174-175), the q[edge[0]] /
// k[edge[1]] gathers and per-head products (:190-196), torch_geometric softmax (:213), the head mean and
// torch_sparse.spmm (:34-35) and the elementwise tail (:46-52).
//
// Why one gather suffices.  The score of edge (i,j), head h is
//     s = q_i,h . k_j,h / sqrt(d_k),   q = W_q x_i + b_q,   k = W_k x_j + b_k
//       = ( (W_k,h^T q_i,h) . x_j  +  q_i,h . b_k,h ) / sqrt(d_k)  =  ( g_i,h . x_j + c_i,h ) / sqrt(d_k)
// so once the wavefront that owns row i has formed g_i,h in registers (d floats per head, from x_i and
// the two weight matrices kept in LDS), every score is a dot product with the SAME neighbour row x_j
// that the aggregation needs anyway.  x_j is gathered once (512 B at d = 128, four full 128-B lines),
// no [N,2A] projection, no [E,h] scores and no [E] weights ever touch HBM, and the 64-B k_j rows (half
// of every fetched line wasted) are never gathered.  The softmax is evaluated online per batch of
// G*U neighbours (running max m_h, denominator l_h and one accumulator per head, rescaled when the max
// moves), which is exact up to rounding:  out_i = (1/H) sum_h acc_h / (l_h + 1e-16).
//
// Mapping: as the aggregation kernel -- one wavefront per row, L lanes span a neighbour's feature row
// (VEC floats x K tiles each), G = 64/L neighbours per wave instruction, U instructions per batch.  The
// dot products are reduced over the L lanes with an xor butterfly; maxima / sums over the G neighbour
// slots likewise.  Rows longer than GNPDE_LONG_ROW are processed as chunks that leave (m, l, acc)
// partials, folded by a small second kernel.  The grid is persistent (a few blocks per CU, each staging
// the weights in LDS once and looping over rows).
#include <cmath>
#include "common.h"
#include "epilogue.h"

namespace gnpde {
namespace {

struct FusedArgs {
  int n, n_long_chunks;
  const int* __restrict__ rowptr;
  const int* __restrict__ colidx;
  const int* __restrict__ lc_row;
  const int* __restrict__ lc_begin;
  const int* __restrict__ lc_end;
  const float* __restrict__ u;
  int d, ld;
  const float* __restrict__ proj_w;  // [2A, d]: rows 0..A-1 = W_q, A..2A-1 = W_k
  const float* __restrict__ proj_b;  // [2A]
  int A, dk;
  float inv_sqrt_dk;
  const float* __restrict__ edge_w;  // CSR order or null (reweight_attention)
  float* partial;                    // [n_long_chunks][H][ldp + 4]
  int ldp;
  gnpde_epilogue_t ep;
};

// ---- cross-lane helpers.  __shfl_xor lowers to ds_bpermute_b32 (an LDS-pipe round trip per call); inside
// a 16-lane row the same butterflies are single VALU instructions with DPP modifiers.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
// sum / max over the L lanes of a group (L = 16, 32 or 64); every lane of the group ends with the result
template <int L>
__device__ __forceinline__ float group_sum(float x) {
  x += dpp_mov<0xB1>(x);   // quad_perm [1,0,3,2]
  x += dpp_mov<0x4E>(x);   // quad_perm [2,3,0,1]
  x += dpp_mov<0x141>(x);  // row_half_mirror
  x += dpp_mov<0x140>(x);  // row_mirror
  if constexpr (L >= 32) x += __shfl_xor(x, 16, kWave);
  if constexpr (L >= 64) x += __shfl_xor(x, 32, kWave);
  return x;
}
// exp for arguments <= 0 on the transcendental unit: 2^(x log2 e); the product's rounding costs a relative
// error of |x| 6e-8, negligible where the weight is not (e^x < 1e-7 beyond |x| = 16)
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

// WPS = wavefronts per SIMD the register allocator must leave room for (launch bound)
template <int H, int VEC, int L, int K, int U, int WPS>
__global__ __launch_bounds__(kBlock, WPS) void attn_rhs_fused_kernel(const FusedArgs a) {
  constexpr int G = kWave / L;
  extern __shared__ __align__(16) float smem[];
  float* sWq = smem;                                   // [A][d]
  float* sWk = smem + static_cast<size_t>(a.A) * a.d;  // [A][d]
  float* sB = smem + 2 * static_cast<size_t>(a.A) * a.d;  // [2A]
  {
    const int n4 = 2 * a.A * a.d / 4;
    const float4* src = reinterpret_cast<const float4*>(a.proj_w);
    float4* dst = reinterpret_cast<float4*>(smem);
    for (int i = threadIdx.x; i < n4; i += kBlock) dst[i] = src[i];
    for (int i = threadIdx.x; i < 2 * a.A; i += kBlock) sB[i] = a.proj_b != nullptr ? a.proj_b[i] : 0.0f;
  }
  __syncthreads();

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int sub = lane / L;
  const int cl = lane % L;
  const int n_items = a.n + a.n_long_chunks;
  const int stride = gridDim.x * kWavesPerBlock;
  const float alpha = alpha_of(a.ep);
  const float beta = a.ep.x0 != nullptr ? *a.ep.beta : 0.0f;

  int cols[K];
  bool colok[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    cols[k] = (k * L + cl) * VEC;
    colok[k] = cols[k] < a.d;
  }

  for (int item = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * kWavesPerBlock + wave); item < n_items;
       item += stride) {
    int row, e0, e1, chunk = -1;
    if (item < a.n) {
      row = item;
      e0 = a.rowptr[row];
      e1 = a.rowptr[row + 1];
      if (e1 - e0 > GNPDE_LONG_ROW) continue;  // processed as chunks
    } else {
      chunk = item - a.n;
      row = a.lc_row[chunk];
      e0 = a.lc_begin[chunk];
      e1 = a.lc_end[chunk];
    }

    // ---- row prologue: x_i slice, then g_h = W_k,h^T (W_q,h x_i + b_q,h) and c_h = q_h . b_k,h
    float xi[K][VEC];
    const size_t roff = static_cast<size_t>(row) * a.ld;
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) xi[k][v] = 0.0f;
      if (colok[k]) load_vec<VEC>(a.u + roff + cols[k], xi[k]);
    }
    float g[H][K][VEC];
    float cst[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      cst[h] = 0.0f;
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int v = 0; v < VEC; ++v) g[h][k][v] = 0.0f;
      for (int t = 0; t < a.dk; ++t) {
        const int mrow = h * a.dk + t;
        float qm = 0.0f;
        float wk[K][VEC];
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) wk[k][v] = 0.0f;
          if (colok[k]) {
            float wq[VEC];
            load_vec<VEC>(sWq + static_cast<size_t>(mrow) * a.d + cols[k], wq);
            load_vec<VEC>(sWk + static_cast<size_t>(mrow) * a.d + cols[k], wk[k]);
#pragma unroll
            for (int v = 0; v < VEC; ++v) qm = fmaf(wq[v], xi[k][v], qm);
          }
        }
        qm = group_sum<L>(qm);
        qm += sB[mrow];
        cst[h] = fmaf(qm, sB[a.A + mrow], cst[h]);
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int v = 0; v < VEC; ++v) g[h][k][v] = fmaf(qm, wk[k][v], g[h][k][v]);
      }
    }

    // ---- edges: online softmax per head
    float m[H], l[H], acc[H][K][VEC];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      m[h] = -INFINITY;
      l[h] = 0.0f;
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[h][k][v] = 0.0f;
    }
    for (int j = e0; j < e1; j += G * U) {
      float vals[U][K][VEC];
      float ew[U];
      bool ok[U];
#pragma unroll
      for (int t = 0; t < U; ++t) {
        const int e = j + t * G + sub;
        ok[t] = e < e1;
        ew[t] = 1.0f;
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int v = 0; v < VEC; ++v) vals[t][k][v] = 0.0f;
        if (ok[t]) {
          const int c = a.colidx[e];
          if (a.edge_w != nullptr) ew[t] = a.edge_w[e];
          const float* src = a.u + static_cast<size_t>(c) * a.ld;
#pragma unroll
          for (int k = 0; k < K; ++k)
            if (colok[k]) load_vec<VEC>(src + cols[k], vals[t][k]);
        }
      }
      float sc[U][H];
#pragma unroll
      for (int t = 0; t < U; ++t)
#pragma unroll
        for (int h = 0; h < H; ++h) {
          float p = 0.0f;
#pragma unroll
          for (int k = 0; k < K; ++k)
#pragma unroll
            for (int v = 0; v < VEC; ++v) p = fmaf(g[h][k][v], vals[t][k][v], p);
          sc[t][h] = p;
        }
#pragma unroll
      for (int t = 0; t < U; ++t)
#pragma unroll
        for (int h = 0; h < H; ++h) sc[t][h] = group_sum<L>(sc[t][h]);
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float bm = -INFINITY;
#pragma unroll
        for (int t = 0; t < U; ++t) {
          float s = (sc[t][h] + cst[h]) * a.inv_sqrt_dk;
          if (a.edge_w != nullptr) s = s * ew[t];
          sc[t][h] = ok[t] ? s : -INFINITY;
          bm = fmaxf(bm, sc[t][h]);
        }
#pragma unroll
        for (int off = L; off < kWave; off <<= 1) bm = fmaxf(bm, __shfl_xor(bm, off, kWave));
        const float mn = fmaxf(m[h], bm);
        const float scale = fast_exp(m[h] - mn);  // first batch: exp(-inf) = 0
        m[h] = mn;
        l[h] *= scale;
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[h][k][v] *= scale;
#pragma unroll
        for (int t = 0; t < U; ++t) {
          const float p = fast_exp(sc[t][h] - mn);  // masked slots: exp(-inf) = 0
          l[h] += p;
#pragma unroll
          for (int k = 0; k < K; ++k)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[h][k][v] = fmaf(p, vals[t][k][v], acc[h][k][v]);
        }
      }
    }
    // combine the G neighbour slots (they share m, so partial l / acc are on one scale)
#pragma unroll
    for (int off = L; off < kWave; off <<= 1)
#pragma unroll
      for (int h = 0; h < H; ++h) {
        l[h] += __shfl_xor(l[h], off, kWave);
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[h][k][v] += __shfl_xor(acc[h][k][v], off, kWave);
      }
    if (sub != 0) continue;

    if (chunk >= 0) {
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float* base = a.partial + (static_cast<size_t>(chunk) * H + h) * (a.ldp + 4);
        if (cl == 0) {
          base[0] = m[h];
          base[1] = l[h];
        }
#pragma unroll
        for (int k = 0; k < K; ++k)
          if (colok[k]) store_vec<VEC>(base + 4 + cols[k], acc[h][k]);
      }
      continue;
    }
    float rl[H];
#pragma unroll
    for (int h = 0; h < H; ++h) rl[h] = (1.0f / static_cast<float>(H)) / (l[h] + 1e-16f);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (!colok[k]) continue;
      float ax[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        float s = 0.0f;
#pragma unroll
        for (int h = 0; h < H; ++h) s = fmaf(acc[h][k][v], rl[h], s);
        ax[v] = s;
      }
      epilogue<VEC, true>(a.ep, alpha, beta, roff + cols[k], ax, xi[k]);
    }
  }
}

// one block per long row: fold the chunk partials, then the epilogue
template <int H>
__global__ __launch_bounds__(kBlock) void attn_long_reduce_kernel(const FusedArgs a, const int* __restrict__ long_rows,
                                                                  const int* __restrict__ long_chunk_ptr) {
  const int lr = blockIdx.x;
  const int row = long_rows[lr];
  const int c0 = long_chunk_ptr[lr], c1 = long_chunk_ptr[lr + 1];
  const size_t hs = static_cast<size_t>(a.ldp + 4);
  float M[H], Lh[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    float mx = -INFINITY;
    for (int c = c0; c < c1; ++c) mx = fmaxf(mx, a.partial[(static_cast<size_t>(c) * H + h) * hs]);
    float ls = 0.0f;
    for (int c = c0; c < c1; ++c) {
      const float* b = a.partial + (static_cast<size_t>(c) * H + h) * hs;
      ls += b[1] * expf(b[0] - mx);
    }
    M[h] = mx;
    Lh[h] = ls + 1e-16f;
  }
  const float alpha = alpha_of(a.ep);
  const float beta = a.ep.x0 != nullptr ? *a.ep.beta : 0.0f;
  for (int col = threadIdx.x; col < a.d; col += blockDim.x) {
    float s = 0.0f;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float ah = 0.0f;
      for (int c = c0; c < c1; ++c) {
        const float* b = a.partial + (static_cast<size_t>(c) * H + h) * hs;
        ah += b[4 + col] * expf(b[0] - M[h]);
      }
      s += ah / Lh[h];
    }
    const size_t off = static_cast<size_t>(row) * a.ld + col;
    const float ax[1] = {s / static_cast<float>(H)};
    const float ui[1] = {a.u[off]};
    epilogue<1, false>(a.ep, alpha, beta, off, ax, ui);
  }
}

int g_num_cus = 0;

template <int H, int L, int K, int U, int WPS>
int launch_fused(const FusedArgs& a, const gnpde_graph_t* g, hipStream_t s) {
  const size_t lds = (2 * static_cast<size_t>(a.A) * a.d + 2 * a.A) * sizeof(float);
  auto kern = attn_rhs_fused_kernel<H, 4, L, K, U, WPS>;
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    GNPDE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(lds)));
    attr_set = true;
  }
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_num_cus = prop.multiProcessorCount;
    if (g_num_cus <= 0) g_num_cus = 256;
  }
  static int occ = 0;  // per instantiation; queried once (not a stream operation)
  if (occ == 0) {
    int q = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, kern, kBlock, lds) != hipSuccess || q < 1) q = 1;
    occ = q;
  }
  int per_cu = occ;
  const int tune = g_tune[GNPDE_TUNE_FUSED_BLOCKS_PER_CU];
  if (tune > 0) per_cu = tune;
  const long long items = static_cast<long long>(a.n) + a.n_long_chunks;
  long long blocks = static_cast<long long>(g_num_cus) * per_cu;
  const long long need = (items + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > need) blocks = need;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), lds, s, a);
  GNPDE_LAUNCH_CHECK();
  if (g->n_long_rows > 0) {
    hipLaunchKernelGGL((attn_long_reduce_kernel<H>), dim3(g->n_long_rows), dim3(kBlock), 0, s, a, g->long_rows,
                       g->long_chunk_ptr);
    GNPDE_LAUNCH_CHECK();
  }
  return 0;
}

template <int H>
int dispatch_fused(const FusedArgs& a, const gnpde_graph_t* g, hipStream_t s) {
  const int slots = a.d / 4;
  if (slots <= 32) {
    if constexpr (H <= 4) {
      switch (g_tune[GNPDE_TUNE_ONE_PASS_VARIANT]) {
        case 1: return launch_fused<H, 16, 2, 4, 2>(a, g, s);
        case 2: return launch_fused<H, 16, 2, 2, 3>(a, g, s);
        case 3: return launch_fused<H, 16, 2, 2, 4>(a, g, s);
        case 4: return launch_fused<H, 16, 2, 1, 4>(a, g, s);
        default: return launch_fused<H, 16, 2, 4, 3>(a, g, s);
      }
    } else {
      return launch_fused<H, 16, 2, 2, 2>(a, g, s);
    }
  }
  if (slots <= 64) return launch_fused<H, 32, 2, 2, 2>(a, g, s);
  if (slots <= 128) return launch_fused<H, 64, 2, 2, 2>(a, g, s);
  return GNPDE_ESHAPE;
}

}  // namespace

size_t fused_attn_workspace_bytes(const gnpde_graph_t* g, int d, int heads) {
  const size_t ldp = align_up(static_cast<size_t>(d), 4);
  return static_cast<size_t>(g->n_long_chunks) * heads * (ldp + 4) * sizeof(float);
}

static thread_local bool g_force = false;

// true when the one-pass kernel covers this configuration (otherwise the multi-kernel path runs)
bool fused_attn_supported(const gnpde_attention_t& at, int d, int ld, const void* u, const gnpde_epilogue_t* epi) {
  if (g_tune[GNPDE_TUNE_ONE_PASS] != 1 && !g_force) return false;  // opt-in: see DESIGN.md (VALU-bound today)
  if (at.type != GNPDE_ATT_SCALED_DOT || at.norm_idx != 0 || at.square_plus) return false;
  if (!(at.heads == 1 || at.heads == 2 || at.heads == 4 || at.heads == 8)) return false;
  if (d % 4 != 0 || ld % 4 != 0 || d > 512) return false;
  if (at.heads == 8 && d > 256) return false;
  const size_t lds = (2 * static_cast<size_t>(at.att_dim) * d + 2 * at.att_dim) * sizeof(float);
  if (lds > 150 * 1024) return false;
  auto al = [](const void* p) { return p == nullptr || reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  if (!al(u)) return false;
  if (epi && !(al(epi->x0) && al(epi->y) && al(epi->k1) && al(epi->k2) && al(epi->k3) && al(epi->out_k) && al(epi->out_y)))
    return false;
  return true;
}

int launch_attn_rhs_fused(const gnpde_graph_t* g, const gnpde_attention_t* at, const float* proj_w, const float* proj_b,
                          const float* u, int d, int ld, const gnpde_epilogue_t* epi, void* ws, size_t ws_bytes,
                          hipStream_t stream) {
  GNPDE_CHECK_ARG(g && at && proj_w && u && epi, GNPDE_EINVAL, "attn_rhs_fused: null argument");
  g_force = true;  // an explicit call always runs the one-pass kernel if the shape is covered
  const bool covered = fused_attn_supported(*at, d, ld, u, epi);
  g_force = false;
  GNPDE_CHECK_ARG(covered, GNPDE_ESHAPE, "attn_rhs_fused: configuration not covered");
  GNPDE_CHECK_ARG(reinterpret_cast<uintptr_t>(proj_w) % 16 == 0, GNPDE_EINVAL, "attn_rhs_fused: weights must be 16-byte aligned");
  GNPDE_CHECK_ARG(epi->alpha != nullptr && (epi->x0 == nullptr || epi->beta != nullptr), GNPDE_EINVAL, "attn_rhs_fused: bad epilogue");
  GNPDE_CHECK_ARG(epi->out_k != u && epi->out_y != u, GNPDE_EINVAL, "attn_rhs_fused: output aliases the gathered operand");
  if (g->n == 0) return 0;
  FusedArgs a{};
  a.n = g->n;
  a.n_long_chunks = g->n_long_chunks;
  a.rowptr = g->rowptr;
  a.colidx = g->colidx;
  a.lc_row = g->long_chunk_row;
  a.lc_begin = g->long_chunk_begin;
  a.lc_end = g->long_chunk_end;
  a.u = u;
  a.d = d;
  a.ld = ld;
  a.proj_w = proj_w;
  a.proj_b = proj_b;
  a.A = at->att_dim;
  a.dk = at->att_dim / at->heads;
  a.inv_sqrt_dk = static_cast<float>(1.0 / std::sqrt(static_cast<double>(a.dk)));
  a.edge_w = at->edge_w_csr;
  a.ldp = static_cast<int>(align_up(static_cast<size_t>(d), 4));
  a.partial = static_cast<float*>(ws);
  a.ep = *epi;
  if (g->n_long_chunks > 0) {
    const size_t need = fused_attn_workspace_bytes(g, d, at->heads);
    GNPDE_CHECK_ARG(ws && ws_bytes >= need && reinterpret_cast<uintptr_t>(ws) % 16 == 0, GNPDE_EWS,
                    "attn_rhs_fused: workspace %zu < %zu bytes", ws_bytes, need);
  }
  switch (at->heads) {
    case 1: return dispatch_fused<1>(a, g, stream);
    case 2: return dispatch_fused<2>(a, g, stream);
    case 4: return dispatch_fused<4>(a, g, stream);
    default: return dispatch_fused<8>(a, g, stream);
  }
}

}  // namespace gnpde

extern "C" size_t gnpde_attn_rhs_fused_workspace_bytes(const gnpde_graph_t* g, int32_t d, int32_t heads) {
  if (!g || d < 1 || heads < 1) return 0;
  return gnpde::fused_attn_workspace_bytes(g, d, heads);
}

extern "C" int gnpde_attn_rhs_fused_supported(const gnpde_attention_t* att, int32_t d, int32_t ld) {
  if (att == nullptr) return 0;
  gnpde::g_force = true;
  const bool ok = gnpde::fused_attn_supported(*att, d, ld, nullptr, nullptr);
  gnpde::g_force = false;
  return ok ? 1 : 0;
}

extern "C" int gnpde_attn_rhs_fused(const gnpde_graph_t* g, const gnpde_attention_t* att, const float* proj_w,
                                    const float* proj_b, const float* u, int32_t d, int32_t ld,
                                    const gnpde_epilogue_t* epi, void* workspace, size_t workspace_bytes, void* stream) {
  return gnpde::launch_attn_rhs_fused(g, att, proj_w, proj_b, u, d, ld, epi, workspace, workspace_bytes,
                                      static_cast<hipStream_t>(stream));
}
