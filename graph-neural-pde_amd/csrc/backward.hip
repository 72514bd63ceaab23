// Backward (VJP) pieces of the GRAND-nl right-hand side for scaled-dot attention with a softmax over the
// row -- what autograd needs to differentiate ODEFuncTransformerAtt.forward (reference
// src/function_transformer_attention.py:38-53) without materialising [E,d_k,h] temporaries:
//   w_e = (1/H) sum_h a_eh,  a_eh = softmax_row(s_eh),  s_eh = q_row,h . k_col,h / sqrt(d_k)
//   given dL/dw_e (from gnpde_sddmm):
//     ds_eh = scale (a_eh / H) (dw_e - sum_{e' in row} a_e'h dw_e')      gnpde_softmax_rows_bwd
//     dq_i,h = (1/sqrt d_k) sum_{e in row i} ds_eh k_col(e),h            gnpde_head_spmm (rows)
//     dk_j,h = (1/sqrt d_k) sum_{e in col j} ds_eh q_row(e),h            gnpde_head_spmm (columns, CSC view)
// All reductions are segment-local (no atomics, deterministic).
#include "common.h"

namespace gnpde {
namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// Segments (rows, or columns through the CSC view) longer than GNPDE_LONG_ROW are "hub" segments: in the kernels
// below the first n_long blocks of the grid take one hub each with all four wavefronts (partial sums through the
// LDS, fixed order), the remaining blocks take one ordinary segment per wavefront and skip the hubs.  Without
// this a power-law graph's 13 K-entry rows serialise in single wavefronts and set the kernel time.
__device__ __forceinline__ float block_sum(float v, float* red) {   // all 256 threads participate
  v = wsum(v);
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ds[p, :] = scale * a[p, :] / H * (dw[p] - sum_{p' in row} a[p', head] dw[p'])   [* edge_w[p]]
// att is in the CALLER's edge order (indexed through perm).  HT = compile-time head count (0: run-time loop).
template <int HT>
__global__ __launch_bounds__(kBlock) void softmax_rows_bwd_kernel(const int* __restrict__ rowptr, const int* __restrict__ perm,
                                                                 const float* __restrict__ att_edge, const float* __restrict__ dw,
                                                                 const float* __restrict__ edge_w, const float* __restrict__ scale_ptr,
                                                                 int scale_sigmoid, int n, int h, const int* __restrict__ long_rows,
                                                                 int n_long, float* __restrict__ ds) {
  __shared__ float red[kWavesPerBlock];
  constexpr int HC = HT > 0 ? HT : 1;
  const int lane = threadIdx.x & (kWave - 1);
  const bool hub = static_cast<int>(blockIdx.x) < n_long;
  int row;
  if (hub) {
    row = long_rows[blockIdx.x];
  } else {
    row = (static_cast<int>(blockIdx.x) - n_long) * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= n) return;
  }
  const int b = rowptr[row], e = rowptr[row + 1];
  if (!hub && e - b > GNPDE_LONG_ROW) return;
  float scale = 1.0f / static_cast<float>(h);
  if (scale_ptr != nullptr) {
    float sc = *scale_ptr;
    if (scale_sigmoid) sc = 1.0f / (1.0f + expf(-sc));
    scale *= sc;
  }
  const int first = hub ? static_cast<int>(threadIdx.x) : lane;
  const int step = hub ? kBlock : kWave;
  if constexpr (HT > 0) {
    float c[HC];
#pragma unroll
    for (int i = 0; i < HC; ++i) c[i] = 0.f;
    for (int p = b + first; p < e; p += step) {
      const float* ar = att_edge + static_cast<size_t>(perm[p]) * HC;
      const float d = dw[p];
#pragma unroll
      for (int i = 0; i < HC; ++i) c[i] = fmaf(ar[i], d, c[i]);
    }
#pragma unroll
    for (int i = 0; i < HC; ++i) c[i] = hub ? block_sum(c[i], red) : wsum(c[i]);
    for (int p = b + first; p < e; p += step) {
      const float* ar = att_edge + static_cast<size_t>(perm[p]) * HC;
      const float d = dw[p];
      const float ew = edge_w != nullptr ? edge_w[p] : 1.0f;
#pragma unroll
      for (int i = 0; i < HC; ++i) {
        float v = ar[i] * scale * (d - c[i]);
        if (edge_w != nullptr) v *= ew;
        ds[static_cast<size_t>(p) * HC + i] = v;
      }
    }
  } else {
    for (int head = 0; head < h; ++head) {
      float c = 0.f;
      for (int p = b + first; p < e; p += step) c = fmaf(att_edge[static_cast<size_t>(perm[p]) * h + head], dw[p], c);
      c = hub ? block_sum(c, red) : wsum(c);
      for (int p = b + first; p < e; p += step) {
        float v = att_edge[static_cast<size_t>(perm[p]) * h + head] * scale * (dw[p] - c);
        if (edge_w != nullptr) v *= edge_w[p];
        ds[static_cast<size_t>(p) * h + head] = v;
      }
    }
  }
}

// out[seg, :] = scale * sum_{p in seg} ds[pos(p), head(col)] * feat[other(p), :]   (A = h * dk columns, dk % 4 == 0)
// lane = (edge slot, float4 column); A4 = A / 4 lanes per edge, a power of two <= 64
template <int A4>
__global__ __launch_bounds__(kBlock) void head_spmm_kernel(const int* __restrict__ segptr, const int* __restrict__ segpos,
                                                          const int* __restrict__ other_of_pos, const float* __restrict__ ds,
                                                          int h, int dk, const float* __restrict__ feat, int ldf, float scale,
                                                          int n, const int* __restrict__ long_segs, int n_long,
                                                          float* __restrict__ out, int ldo) {
  __shared__ float part[kWavesPerBlock][A4][4];
  constexpr int ES = kWave / A4;  // edges per wavefront iteration
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const bool hub = static_cast<int>(blockIdx.x) < n_long;
  int seg;
  if (hub) {
    seg = long_segs[blockIdx.x];
  } else {
    seg = (static_cast<int>(blockIdx.x) - n_long) * kWavesPerBlock + wave;
    if (seg >= n) return;
  }
  const int es = lane / A4, a4 = lane % A4;
  const int head = (a4 * 4) / dk;
  const int b = segptr[seg], e = segptr[seg + 1];
  if (!hub && e - b > GNPDE_LONG_ROW) return;
  const int first = hub ? wave * ES + es : es;
  const int step = hub ? ES * kWavesPerBlock : ES;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int U = 4;     // independent (position -> other end -> feature row) load chains in flight
  for (int t0 = b + first; t0 < e; t0 += U * step) {
    float w[U];
    float4 f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u * step;
      w[u] = 0.f;
      f[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < e) {
        const int p = segpos != nullptr ? segpos[t] : t;
        const int o = other_of_pos[p];
        w[u] = ds[static_cast<size_t>(p) * h + head];
        f[u] = *reinterpret_cast<const float4*>(feat + static_cast<size_t>(o) * ldf + a4 * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc[0] = fmaf(w[u], f[u].x, acc[0]); acc[1] = fmaf(w[u], f[u].y, acc[1]);
      acc[2] = fmaf(w[u], f[u].z, acc[2]); acc[3] = fmaf(w[u], f[u].w, acc[3]);
    }
  }
#pragma unroll
  for (int off = A4; off < kWave; off <<= 1)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += __shfl_xor(acc[i], off, kWave);
  if (hub) {
    if (es == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) part[wave][a4][i] = acc[i];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (part[0][a4][i] + part[1][a4][i]) + (part[2][a4][i] + part[3][a4][i]);
  }
  if (es == 0)
    *reinterpret_cast<float4*>(out + static_cast<size_t>(seg) * ldo + a4 * 4) =
        make_float4(scale * acc[0], scale * acc[1], scale * acc[2], scale * acc[3]);
}

}  // namespace
}  // namespace gnpde

extern "C" int gnpde_softmax_rows_bwd(const gnpde_graph_t* g, const float* att_edge, int32_t heads, const float* dw_csr,
                                      const float* edge_w_csr, const float* scale, int32_t scale_sigmoid, float* ds_csr,
                                      void* stream) {
  using namespace gnpde;
  GNPDE_CHECK_ARG(g && g->perm && att_edge && dw_csr && ds_csr && heads >= 1, GNPDE_EINVAL, "softmax_rows_bwd: bad arguments");
  GNPDE_CHECK_ARG(g->n_long_rows == 0 || g->long_rows, GNPDE_EINVAL, "softmax_rows_bwd: graph has long rows but no list of them");
  if (g->n == 0 || g->e == 0) return 0;
  const int nl = g->n_long_rows;
  const dim3 grid(static_cast<unsigned>(nl + (g->n + kWavesPerBlock - 1) / kWavesPerBlock));
  hipStream_t s = static_cast<hipStream_t>(stream);
#define GNPDE_SB(HH) \
  hipLaunchKernelGGL((softmax_rows_bwd_kernel<HH>), grid, dim3(kBlock), 0, s, g->rowptr, g->perm, att_edge, dw_csr, edge_w_csr, \
                     scale, scale_sigmoid, g->n, heads, g->long_rows, nl, ds_csr)
  switch (heads) {
    case 1: GNPDE_SB(1); break;
    case 2: GNPDE_SB(2); break;
    case 4: GNPDE_SB(4); break;
    case 8: GNPDE_SB(8); break;
    default: GNPDE_SB(0); break;
  }
#undef GNPDE_SB
  GNPDE_LAUNCH_CHECK();
  return 0;
}

extern "C" int gnpde_head_spmm(const gnpde_graph_t* g, int32_t by_column, const float* ds_csr, int32_t heads, int32_t dk,
                               const float* feat, int32_t ldf, float scale, float* out, int32_t ldo, void* stream) {
  using namespace gnpde;
  GNPDE_CHECK_ARG(g && ds_csr && feat && out && heads >= 1 && dk >= 4 && dk % 4 == 0, GNPDE_EINVAL, "head_spmm: bad arguments");
  const int A = heads * dk, a4 = A / 4;
  GNPDE_CHECK_ARG(a4 <= 64 && (a4 & (a4 - 1)) == 0, GNPDE_ESHAPE, "head_spmm: attention_dim/4 = %d must be a power of two <= 64", a4);
  GNPDE_CHECK_ARG(ldf >= A && ldo >= A && ldf % 4 == 0 && ldo % 4 == 0 && reinterpret_cast<uintptr_t>(feat) % 16 == 0 &&
                      reinterpret_cast<uintptr_t>(out) % 16 == 0, GNPDE_EINVAL, "head_spmm: operands must be 16-byte aligned");
  GNPDE_CHECK_ARG(!by_column || (g->cscptr && g->cscpos && g->rowidx), GNPDE_EINVAL, "head_spmm: column mode needs the CSC view");
  if (g->n == 0) return 0;
  const int* segptr = by_column ? g->cscptr : g->rowptr;
  const int* segpos = by_column ? g->cscpos : nullptr;
  const int* other = by_column ? g->rowidx : g->colidx;
  const int* long_segs = by_column ? g->long_cols : g->long_rows;
  const int nl = by_column ? g->n_long_cols : g->n_long_rows;
  GNPDE_CHECK_ARG(nl == 0 || long_segs, GNPDE_EINVAL, "head_spmm: graph has long segments but no list of them");
  const unsigned grid = static_cast<unsigned>(nl + (g->n + kWavesPerBlock - 1) / kWavesPerBlock);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define GNPDE_HS(N4) \
  hipLaunchKernelGGL((head_spmm_kernel<N4>), dim3(grid), dim3(kBlock), 0, s, segptr, segpos, other, ds_csr, heads, dk, feat, ldf, \
                     scale, g->n, long_segs, nl, out, ldo)
  switch (a4) {
    case 1: GNPDE_HS(1); break;
    case 2: GNPDE_HS(2); break;
    case 4: GNPDE_HS(4); break;
    case 8: GNPDE_HS(8); break;
    case 16: GNPDE_HS(16); break;
    case 32: GNPDE_HS(32); break;
    default: GNPDE_HS(64); break;
  }
#undef GNPDE_HS
  GNPDE_LAUNCH_CHECK();
  return 0;
}
