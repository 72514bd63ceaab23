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

// The gradient of the head-mean weights per entry, read through an optional position map: r[p] = data[pos[p]] when the producer left it in
// ANOTHER graph's order (the cotangent-side sweep of a recorded solve forms the products on the transposed graph: reading them through
// the map here replaces a 26-us permutation pass; csrc/adjoint.hip).
struct RRef {
  const float* __restrict__ data;
  const int* __restrict__ pos;
  __device__ __forceinline__ float operator[](long long p) const { return pos != nullptr ? data[pos[p]] : data[p]; }
};


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
// lane = (edge slot, float4 column); A4 = A / 4 lanes per edge, a power of two <= 64.  GL lanes work on one segment:
// GL = 16 (four segments per wavefront) for the segments with len_lo < length <= len_hi = 64 -- most rows of a
// power-law graph --, GL = 64 for the longer ones; each launch skips the segments outside its length class, and the
// GL = 64 launch also takes the hub segments (HUBS) with whole blocks.
template <int A4, int GL, bool HUBS>
__global__ __launch_bounds__(kBlock) void head_spmm_kernel(const int* __restrict__ segptr, const int* __restrict__ segpos,
                                                          const int* __restrict__ other_of_pos, const float* __restrict__ ds,
                                                          int h, int dk, const float* __restrict__ feat, int ldf, float scale,
                                                          int n, int len_lo, int len_hi, const int* __restrict__ long_segs,
                                                          int n_long, float* __restrict__ out, int ldo) {
  __shared__ float part[kWavesPerBlock][A4][4];
  constexpr int ES = GL / A4;        // edges per group iteration
  constexpr int SPW = kWave / GL;    // segments per wavefront
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const bool hub = HUBS && static_cast<int>(blockIdx.x) < n_long;
  const int gl = lane % GL;
  const int es = gl / A4, a4 = gl % A4;
  const int head = (a4 * 4) / dk;
  int seg, b = 0, e = 0;
  if (hub) {
    seg = long_segs[blockIdx.x];
    b = segptr[seg]; e = segptr[seg + 1];
  } else {
    seg = ((static_cast<int>(blockIdx.x) - (HUBS ? n_long : 0)) * kWavesPerBlock + wave) * SPW + lane / GL;
    if (seg < n) {
      b = segptr[seg]; e = segptr[seg + 1];
      if (e - b <= len_lo || e - b > len_hi) e = b;      // not this launch's length class: nothing to do
    } else {
      seg = -1;
    }
  }
  const bool mine = e > b || (seg >= 0 && !hub && len_lo == 0 && e == b && segptr[seg + 1] == segptr[seg]);  // empty segments: zero row, written by the first class
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
  for (int off = A4; off < GL; off <<= 1)
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
  if (es == 0 && (mine || hub))
    *reinterpret_cast<float4*>(out + static_cast<size_t>(seg) * ldo + a4 * 4) =
        make_float4(scale * acc[0], scale * acc[1], scale * acc[2], scale * acc[3]);
}

// ------------------------------------------------------------------------------------------------
// Row attention backward in one pass: recomputes the scaled-dot scores and their row softmax from q, k and writes
//   ds[p, h] = scale (a[p,h] / H) (r[p] - sum_{p' in row} a[p',h] r[p'])   [* edge_w[p]]
// i.e. scores + segment statistics + normalise + softmax backward of the general path (four launches and an [E,h]
// attention array in edge order) as ONE launch that never materialises the attention.  lane = one entry of the
// row at a time (its k row is A = H * DK floats, gathered as float4s), scores of up to 8 entries per lane kept in
// registers for rows <= 512; hub rows are taken by whole blocks in three passes (max, sum and c, write).
// ------------------------------------------------------------------------------------------------
template <int H, int DK>
__device__ __forceinline__ void row_scores(const float* __restrict__ qrow, const float* __restrict__ krow, float inv, float ew,
                                           float (&sc)[H]) {
#pragma unroll
  for (int hh = 0; hh < H; ++hh) {
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < DK; c += 4) {
      const float4 qv = *reinterpret_cast<const float4*>(qrow + hh * DK + c);
      const float4 kv = *reinterpret_cast<const float4*>(krow + hh * DK + c);
      dot = fmaf(qv.x, kv.x, dot); dot = fmaf(qv.y, kv.y, dot); dot = fmaf(qv.z, kv.z, dot); dot = fmaf(qv.w, kv.w, dot);
    }
    sc[hh] = dot * inv * ew;
  }
}

__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
  return v;
}

__device__ __forceinline__ float block_max(float v, float* red) {
  v = wmax(v);
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// group reductions over GL consecutive lanes (GL = 16 or 64)
template <int GL>
__device__ __forceinline__ float gsum(float v) {
#pragma unroll
  for (int off = GL / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}
template <int GL>
__device__ __forceinline__ float gmax(float v) {
#pragma unroll
  for (int off = GL / 2; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
  return v;
}

template <int N, int MASK>
struct LaneTranspose {   // N values per lane -> 1 over the lane pairs (l ^ MASK), then MASK / 2, ...; lane l ends with value index given by its top bits
  static __device__ __forceinline__ void run(float* p, int l) {
    constexpr int Hh = N / 2;
    const bool upper = (l & MASK) != 0;
#pragma unroll
    for (int j = 0; j < Hh; ++j) {
      const float send = upper ? p[j] : p[j + Hh];
      const float keep = upper ? p[j + Hh] : p[j];
      p[j] = keep + __shfl_xor(send, MASK, kWave);
    }
    if constexpr (Hh > 1) LaneTranspose<Hh, MASK / 2>::run(p, l);
  }
};

// Ordinary rows come from the degree-binned records {row, begin, len, 0} (graph_prep): rows with <= 16 entries
// take GL = 16 lanes (four rows per wavefront, one entry per lane), rows with 17..512 entries a whole wavefront
// (GL = 64, up to PER = 8 entries per lane, scores kept in registers).  With HUBS the first n_long blocks of the
// grid take one hub row each.
// DQ: the kernel also forms d q[row, :] = (1 / sqrt d_k) sum_{p in row} ds[p, head] k[col_p, :] -- the row-side head sum of the adjoint
// stage (csrc/adjoint.hip), from the ds values it has just computed and the k rows it has just read, instead of a second launch that
// re-reads both (A = H DK <= 32 columns: A accumulators per lane, summed over the row's lanes by the transposing butterfly).
template <int H, int DK, int GL, int PER, bool HUBS, bool DQ>
__global__ __launch_bounds__(kBlock) void attention_rows_bwd_kernel(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                                                   const int* __restrict__ bin_rows, int first_rec, int n_rec,
                                                                   const float* __restrict__ q, const float* __restrict__ k, int ldqk,
                                                                   const RRef r, const float* __restrict__ edge_w,
                                                                   const float* __restrict__ scale_ptr, int scale_sigmoid,
                                                                   const int* __restrict__ long_rows, int n_long,
                                                                   float* __restrict__ ds, float* __restrict__ dq, int lddq) {
  __shared__ float red[kWavesPerBlock];
  constexpr int A = H * DK;
  constexpr int AQ = DQ ? A : 1;
  __shared__ float qpart[DQ ? kWavesPerBlock : 1][AQ];
  float accq[AQ];
#pragma unroll
  for (int c = 0; c < AQ; ++c) accq[c] = 0.f;
  auto add_dq = [&](const float (&dsv)[H], const float* __restrict__ krow) {
    if constexpr (DQ) {
#pragma unroll
      for (int c = 0; c < A; c += 4) {
        const float4 kv = *reinterpret_cast<const float4*>(krow + c);
        const float wv = dsv[c / DK];
        accq[c + 0] = fmaf(wv, kv.x, accq[c + 0]); accq[c + 1] = fmaf(wv, kv.y, accq[c + 1]);
        accq[c + 2] = fmaf(wv, kv.z, accq[c + 2]); accq[c + 3] = fmaf(wv, kv.w, accq[c + 3]);
      }
    }
  };
  constexpr int RPW = kWave / GL;               // rows per wavefront
  const int lane = threadIdx.x & (kWave - 1);
  const bool hub = HUBS && static_cast<int>(blockIdx.x) < n_long;
  float scale = 1.0f / static_cast<float>(H);
  if (scale_ptr != nullptr) {
    float sc = *scale_ptr;
    if (scale_sigmoid) sc = 1.0f / (1.0f + expf(-sc));
    scale *= sc;
  }
  const float inv = 1.0f / sqrtf(static_cast<float>(DK));

  if (!hub) {
    const int gl = lane % GL;
    const long long rec = (static_cast<long long>(blockIdx.x) - (HUBS ? n_long : 0)) * (kWavesPerBlock * RPW) +
                          (threadIdx.x >> 6) * RPW + lane / GL;
    const bool live_row = rec < n_rec;
    int4 info = make_int4(0, 0, 0, 0);
    if (live_row) info = reinterpret_cast<const int4*>(bin_rows)[first_rec + rec];
    const int b = info.y, e = info.y + info.z;
    const float* qrow = q + static_cast<size_t>(info.x) * ldqk;
    float s[PER][H];
    float rr[PER], ew[PER];
    float mx[H];
#pragma unroll
    for (int hh = 0; hh < H; ++hh) mx[hh] = -INFINITY;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int p = b + gl + i * GL;
      rr[i] = 0.f; ew[i] = 1.f;
#pragma unroll
      for (int hh = 0; hh < H; ++hh) s[i][hh] = -INFINITY;
      if (p < e) {
        ew[i] = edge_w != nullptr ? edge_w[p] : 1.f;
        rr[i] = r[p];
        row_scores<H, DK>(qrow, k + static_cast<size_t>(colidx[p]) * ldqk, inv, ew[i], s[i]);
#pragma unroll
        for (int hh = 0; hh < H; ++hh) mx[hh] = fmaxf(mx[hh], s[i][hh]);
      }
    }
    float den[H], c[H];
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
      mx[hh] = gmax<GL>(mx[hh]);
      den[hh] = 0.f; c[hh] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const bool live = b + gl + i * GL < e;
#pragma unroll
      for (int hh = 0; hh < H; ++hh) {
        const float ex = live ? expf(s[i][hh] - mx[hh]) : 0.f;
        s[i][hh] = ex;
        den[hh] += ex;
        c[hh] = fmaf(ex, rr[i], c[hh]);
      }
    }
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
      den[hh] = gsum<GL>(den[hh]) + 1e-16f;
      c[hh] = gsum<GL>(c[hh]) / den[hh];
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int p = b + gl + i * GL;
      if (p < e) {
        float out[H];
#pragma unroll
        for (int hh = 0; hh < H; ++hh) out[hh] = (s[i][hh] / den[hh]) * scale * (rr[i] - c[hh]) * ew[i];
        if constexpr (H == 4) {
          *reinterpret_cast<float4*>(ds + static_cast<size_t>(p) * 4) = make_float4(out[0], out[1], out[2], out[3]);
        } else {
#pragma unroll
          for (int hh = 0; hh < H; ++hh) ds[static_cast<size_t>(p) * H + hh] = out[hh];
        }
        add_dq(out, k + static_cast<size_t>(colidx[p]) * ldqk);
      }
    }
    if constexpr (DQ) {
      constexpr int C = A < GL ? A : GL;
      constexpr int LPE = GL / C;
#pragma unroll
      for (int ch = 0; ch < A / C; ++ch) {
        if constexpr (C > 1) LaneTranspose<C, GL / 2>::run(accq + ch * C, gl);
        float v = accq[ch * C];
#pragma unroll
        for (int m = LPE / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
        if (live_row && (gl % LPE) == 0) dq[static_cast<size_t>(info.x) * lddq + ch * C + gl / LPE] = inv * v;
      }
    }
    return;
  }

  // hub row: every pass recomputes the scores (three gathers of the k rows, hub entries only)
  const int row = long_rows[blockIdx.x];
  const int b = rowptr[row], e = rowptr[row + 1];
  const float* qrow = q + static_cast<size_t>(row) * ldqk;
  float mx[H], den[H], c[H];
#pragma unroll
  for (int hh = 0; hh < H; ++hh) { mx[hh] = -INFINITY; den[hh] = 0.f; c[hh] = 0.f; }
  for (int p = b + static_cast<int>(threadIdx.x); p < e; p += kBlock) {
    float sc[H];
    row_scores<H, DK>(qrow, k + static_cast<size_t>(colidx[p]) * ldqk, inv, edge_w != nullptr ? edge_w[p] : 1.f, sc);
#pragma unroll
    for (int hh = 0; hh < H; ++hh) mx[hh] = fmaxf(mx[hh], sc[hh]);
  }
#pragma unroll
  for (int hh = 0; hh < H; ++hh) mx[hh] = block_max(mx[hh], red);
  for (int p = b + static_cast<int>(threadIdx.x); p < e; p += kBlock) {
    float sc[H];
    row_scores<H, DK>(qrow, k + static_cast<size_t>(colidx[p]) * ldqk, inv, edge_w != nullptr ? edge_w[p] : 1.f, sc);
    const float rp = r[p];
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
      const float ex = expf(sc[hh] - mx[hh]);
      den[hh] += ex;
      c[hh] = fmaf(ex, rp, c[hh]);
    }
  }
#pragma unroll
  for (int hh = 0; hh < H; ++hh) {
    den[hh] = block_sum(den[hh], red) + 1e-16f;
    c[hh] = block_sum(c[hh], red) / den[hh];
  }
  for (int p = b + static_cast<int>(threadIdx.x); p < e; p += kBlock) {
    float sc[H];
    const float ewp = edge_w != nullptr ? edge_w[p] : 1.f;
    row_scores<H, DK>(qrow, k + static_cast<size_t>(colidx[p]) * ldqk, inv, ewp, sc);
    const float rp = r[p];
    float out[H];
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
      out[hh] = (expf(sc[hh] - mx[hh]) / den[hh]) * scale * (rp - c[hh]) * ewp;
      ds[static_cast<size_t>(p) * H + hh] = out[hh];
    }
    add_dq(out, k + static_cast<size_t>(colidx[p]) * ldqk);
  }
  if constexpr (DQ) {
    const int lane_ = threadIdx.x & (kWave - 1);
    constexpr int C = A < kWave ? A : kWave;
    constexpr int LPE = kWave / C;
#pragma unroll
    for (int ch = 0; ch < A / C; ++ch) {
      if constexpr (C > 1) LaneTranspose<C, kWave / 2>::run(accq + ch * C, lane_);
      float v = accq[ch * C];
#pragma unroll
      for (int m = LPE / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
      if ((lane_ % LPE) == 0) qpart[threadIdx.x >> 6][ch * C + lane_ / LPE] = v;
    }
    __syncthreads();
    if (threadIdx.x < A)
      dq[static_cast<size_t>(row) * lddq + threadIdx.x] =
          inv * ((qpart[0][threadIdx.x] + qpart[1][threadIdx.x]) + (qpart[2][threadIdx.x] + qpart[3][threadIdx.x]));
  }
}

// one entry of a head-wise row sum: acc[c] += ds[pp, head(c)] * feat[o, c] (head_rowsum_kernel / head_rowsum_hub_kernel below)
template <int H, int DK4>
__device__ __forceinline__ void entry_fma(const float* __restrict__ ds, const float* __restrict__ feat, int ldf, int h_rt, long long pp, int o,
                                          float (&acc)[H * DK4 * 4]) {
  float dsv[H];
  if constexpr (H == 4) {
    const float4 t = *reinterpret_cast<const float4*>(ds + pp * 4);
    dsv[0] = t.x; dsv[1] = t.y; dsv[2] = t.z; dsv[3] = t.w;
  } else if constexpr (H == 2) {
    const float2 t = *reinterpret_cast<const float2*>(ds + pp * 2);
    dsv[0] = t.x; dsv[1] = t.y;
  } else {
#pragma unroll
    for (int hh = 0; hh < H; ++hh) dsv[hh] = ds[pp * H + hh];
  }
  const float* fr = feat + static_cast<size_t>(o) * ldf;
  float4 f[H * DK4];
#pragma unroll
  for (int c4 = 0; c4 < H * DK4; ++c4) f[c4] = *reinterpret_cast<const float4*>(fr + 4 * c4);
#pragma unroll
  for (int c4 = 0; c4 < H * DK4; ++c4) {
    const float w = dsv[c4 / DK4];
    acc[4 * c4 + 0] = fmaf(w, f[c4].x, acc[4 * c4 + 0]);
    acc[4 * c4 + 1] = fmaf(w, f[c4].y, acc[4 * c4 + 1]);
    acc[4 * c4 + 2] = fmaf(w, f[c4].z, acc[4 * c4 + 2]);
    acc[4 * c4 + 3] = fmaf(w, f[c4].w, acc[4 * c4 + 3]);
  }
}

// ------------------------------------------------------------------------------------------------
// Hub rows (> GNPDE_LONG_ROW entries) of the two backward row kernels, 1024 threads per row.  Inside attention_rows_bwd_kernel /
// head_rowsum_kernel a hub is ONE 256-thread block walking the row three times (once) with one dependent gather per step: the 13 k-entry
// hub of the ogbn-arxiv shape alone is 150 serialised round trips and sets the duration of the whole launch (127 us for 3.6 k rows + 204
// hubs, against 28 us for the 23.6 k rows of 17..64 entries).  Here a hub gets 16 wavefronts and every lane keeps UH entries in flight;
// same arithmetic per entry, block-wide reductions in a fixed order (wave butterflies, then the 16 wave results in wave order).
// ------------------------------------------------------------------------------------------------
constexpr int kHubThreads = 1024;
constexpr int kHubWaves = kHubThreads / kWave;

__device__ __forceinline__ float hub_block_sum(float v, float* red) {   // all 1024 threads take part
  v = wsum(v);
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < kHubWaves; ++w) t += red[w];
  return t;
}
__device__ __forceinline__ float hub_block_max(float v, float* red) {
  v = wmax(v);
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int w = 1; w < kHubWaves; ++w) t = fmaxf(t, red[w]);
  return t;
}

template <int H, int DK, bool DQ>
__global__ __launch_bounds__(kHubThreads) void attention_hub_bwd_kernel(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                                                       const float* __restrict__ q, const float* __restrict__ k, int ldqk,
                                                                       const RRef r, const float* __restrict__ edge_w,
                                                                       const float* __restrict__ scale_ptr, int scale_sigmoid,
                                                                       const int* __restrict__ long_rows, float* __restrict__ ds,
                                                                       float* __restrict__ dq, int lddq) {
  constexpr int A = H * DK;
  constexpr int AQ = DQ ? A : 1;
  constexpr int UH = 4;
  __shared__ float red[kHubWaves];
  __shared__ float qpart[DQ ? kHubWaves : 1][AQ];
  float scale = 1.0f / static_cast<float>(H);
  if (scale_ptr != nullptr) {
    float sc = *scale_ptr;
    if (scale_sigmoid) sc = 1.0f / (1.0f + expf(-sc));
    scale *= sc;
  }
  const float inv = 1.0f / sqrtf(static_cast<float>(DK));
  const int row = long_rows[blockIdx.x];
  const int b = rowptr[row], e = rowptr[row + 1];
  const float* qrow = q + static_cast<size_t>(row) * ldqk;
  float mx[H], den[H], c[H];
#pragma unroll
  for (int hh = 0; hh < H; ++hh) { mx[hh] = -INFINITY; den[hh] = 0.f; c[hh] = 0.f; }
  // pass 1: maxima (UH entries per lane per step: their column ids, then their key rows, are all in flight together)
  for (int p0 = b + static_cast<int>(threadIdx.x); p0 < e; p0 += UH * kHubThreads) {
    int cols[UH];
#pragma unroll
    for (int u = 0; u < UH; ++u) { const int p = p0 + u * kHubThreads; cols[u] = p < e ? colidx[p] : -1; }
#pragma unroll
    for (int u = 0; u < UH; ++u) {
      if (cols[u] < 0) continue;
      const int p = p0 + u * kHubThreads;
      float sc[H];
      row_scores<H, DK>(qrow, k + static_cast<size_t>(cols[u]) * ldqk, inv, edge_w != nullptr ? edge_w[p] : 1.f, sc);
#pragma unroll
      for (int hh = 0; hh < H; ++hh) mx[hh] = fmaxf(mx[hh], sc[hh]);
    }
  }
#pragma unroll
  for (int hh = 0; hh < H; ++hh) mx[hh] = hub_block_max(mx[hh], red);
  // pass 2: denominators and c = sum a r
  for (int p0 = b + static_cast<int>(threadIdx.x); p0 < e; p0 += UH * kHubThreads) {
    int cols[UH];
#pragma unroll
    for (int u = 0; u < UH; ++u) { const int p = p0 + u * kHubThreads; cols[u] = p < e ? colidx[p] : -1; }
#pragma unroll
    for (int u = 0; u < UH; ++u) {
      if (cols[u] < 0) continue;
      const int p = p0 + u * kHubThreads;
      float sc[H];
      row_scores<H, DK>(qrow, k + static_cast<size_t>(cols[u]) * ldqk, inv, edge_w != nullptr ? edge_w[p] : 1.f, sc);
      const float rp = r[p];
#pragma unroll
      for (int hh = 0; hh < H; ++hh) {
        const float ex = expf(sc[hh] - mx[hh]);
        den[hh] += ex;
        c[hh] = fmaf(ex, rp, c[hh]);
      }
    }
  }
#pragma unroll
  for (int hh = 0; hh < H; ++hh) {
    den[hh] = hub_block_sum(den[hh], red) + 1e-16f;
    c[hh] = hub_block_sum(c[hh], red) / den[hh];
  }
  // pass 3: ds (and d q)
  float accq[AQ];
#pragma unroll
  for (int cc = 0; cc < AQ; ++cc) accq[cc] = 0.f;
  for (int p0 = b + static_cast<int>(threadIdx.x); p0 < e; p0 += UH * kHubThreads) {
    int cols[UH];
#pragma unroll
    for (int u = 0; u < UH; ++u) { const int p = p0 + u * kHubThreads; cols[u] = p < e ? colidx[p] : -1; }
#pragma unroll
    for (int u = 0; u < UH; ++u) {
      if (cols[u] < 0) continue;
      const int p = p0 + u * kHubThreads;
      const float ewp = edge_w != nullptr ? edge_w[p] : 1.f;
      const float* krow = k + static_cast<size_t>(cols[u]) * ldqk;
      float sc[H], out[H];
      row_scores<H, DK>(qrow, krow, inv, ewp, sc);
      const float rp = r[p];
#pragma unroll
      for (int hh = 0; hh < H; ++hh) {
        out[hh] = (expf(sc[hh] - mx[hh]) / den[hh]) * scale * (rp - c[hh]) * ewp;
        ds[static_cast<size_t>(p) * H + hh] = out[hh];
      }
      if constexpr (DQ) {
#pragma unroll
        for (int cc = 0; cc < A; cc += 4) {
          const float4 kv = *reinterpret_cast<const float4*>(krow + cc);
          const float wv = out[cc / DK];
          accq[cc + 0] = fmaf(wv, kv.x, accq[cc + 0]); accq[cc + 1] = fmaf(wv, kv.y, accq[cc + 1]);
          accq[cc + 2] = fmaf(wv, kv.z, accq[cc + 2]); accq[cc + 3] = fmaf(wv, kv.w, accq[cc + 3]);
        }
      }
    }
  }
  if constexpr (DQ) {
    const int lane_ = threadIdx.x & (kWave - 1);
    constexpr int C = A < kWave ? A : kWave;
    constexpr int LPE = kWave / C;
#pragma unroll
    for (int ch = 0; ch < A / C; ++ch) {
      if constexpr (C > 1) LaneTranspose<C, kWave / 2>::run(accq + ch * C, lane_);
      float v = accq[ch * C];
#pragma unroll
      for (int m = LPE / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
      if ((lane_ % LPE) == 0) qpart[threadIdx.x >> 6][ch * C + lane_ / LPE] = v;
    }
    __syncthreads();
    if (threadIdx.x < A) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < kHubWaves; ++w) t += qpart[w][threadIdx.x];
      dq[static_cast<size_t>(row) * lddq + threadIdx.x] = inv * t;
    }
  }
}

template <int H, int DK4>
__global__ __launch_bounds__(kHubThreads) void head_rowsum_hub_kernel(const int* __restrict__ rowptr, const int* __restrict__ other,
                                                                     const int* __restrict__ pos, const float* __restrict__ ds,
                                                                     const float* __restrict__ feat, int ldf, float scale,
                                                                     const int* __restrict__ long_rows, float* __restrict__ out, int ldo) {
  constexpr int A = H * DK4 * 4;
  constexpr int UH = 4;
  __shared__ float part[kHubWaves][A];
  const int lane = threadIdx.x & (kWave - 1);
  const int seg = long_rows[blockIdx.x];
  const int b = rowptr[seg], e = rowptr[seg + 1];
  float acc[A];
#pragma unroll
  for (int c = 0; c < A; ++c) acc[c] = 0.f;
  for (int t0 = b + static_cast<int>(threadIdx.x); t0 < e; t0 += UH * kHubThreads) {
    int pp[UH], oo[UH];
#pragma unroll
    for (int u = 0; u < UH; ++u) {
      const int t = t0 + u * kHubThreads;
      pp[u] = -1; oo[u] = 0;
      if (t < e) { pp[u] = pos != nullptr ? pos[t] : t; oo[u] = other[t]; }
    }
#pragma unroll
    for (int u = 0; u < UH; ++u)
      if (pp[u] >= 0) entry_fma<H, DK4>(ds, feat, ldf, H, pp[u], oo[u], acc);
  }
  constexpr int C = A < kWave ? A : kWave;
  constexpr int LPE = kWave / C;
#pragma unroll
  for (int ch = 0; ch < A / C; ++ch) {
    if constexpr (C > 1) LaneTranspose<C, kWave / 2>::run(acc + ch * C, lane);
    float v = acc[ch * C];
#pragma unroll
    for (int m = LPE / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
    if ((lane % LPE) == 0) part[threadIdx.x >> 6][ch * C + lane / LPE] = v;
  }
  __syncthreads();
  if (threadIdx.x < A) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < kHubWaves; ++w) t += part[w][threadIdx.x];
    out[static_cast<size_t>(seg) * ldo + threadIdx.x] = scale * t;
  }
}

// ------------------------------------------------------------------------------------------------
// Hub rows as 512-entry CHUNKS spread over the chip (the graph's long_chunk_* lists, as the forward's hub phases).  One workgroup
// per hub -- even with 1024 threads -- is bound by what ONE CU's memory pipeline delivers: the 10.8 k-entry hub of the ogbn-arxiv
// shape pulls 3 MB of key rows / ds vectors through one CU, 127 us for the softmax backward and 45 us for a head sum while 250 CUs
// idle.  Chunked: (a) per chunk, scores + online-softmax partials (m_c, l_c = sum e^{s-m_c}, t_c = sum e^{s-m_c} r); (b) per chunk,
// fold the row's partials (<= 22 of them, every chunk block redoes the fold), then ds for its entries and its partial d q; (c) per
// hub row, the chunks' d q partials in chunk order.  The head sums (d k, or d q when it is not fused) are (b')/(c) alone.
// Needs scratch: [n_long_chunks] x (3 H + A) floats (hub_bwd_workspace_floats).
// ------------------------------------------------------------------------------------------------
template <int H, int DK>
__global__ __launch_bounds__(kBlock) void attention_hub_stats_kernel(const int* __restrict__ colidx, const int* __restrict__ lc_row,
                                                                    const int* __restrict__ lc_begin, const int* __restrict__ lc_end,
                                                                    const float* __restrict__ q, const float* __restrict__ k, int ldqk,
                                                                    const RRef r, const float* __restrict__ edge_w,
                                                                    float* __restrict__ part) {
  constexpr int PER = GNPDE_LONG_ROW / kBlock;      // entries per thread
  __shared__ float red[kWavesPerBlock];
  const int c = blockIdx.x;
  const int row = lc_row[c], b = lc_begin[c], e = lc_end[c];
  const float inv = 1.0f / sqrtf(static_cast<float>(DK));
  const float* qrow = q + static_cast<size_t>(row) * ldqk;
  int cols[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) { const int p = b + i * kBlock + static_cast<int>(threadIdx.x); cols[i] = p < e ? colidx[p] : -1; }
  float sc[PER][H], rr[PER];
  float mx[H];
#pragma unroll
  for (int hh = 0; hh < H; ++hh) mx[hh] = -INFINITY;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int p = b + i * kBlock + static_cast<int>(threadIdx.x);
    rr[i] = 0.f;
#pragma unroll
    for (int hh = 0; hh < H; ++hh) sc[i][hh] = -INFINITY;
    if (cols[i] >= 0) {
      rr[i] = r[p];
      row_scores<H, DK>(qrow, k + static_cast<size_t>(cols[i]) * ldqk, inv, edge_w != nullptr ? edge_w[p] : 1.f, sc[i]);
#pragma unroll
      for (int hh = 0; hh < H; ++hh) mx[hh] = fmaxf(mx[hh], sc[i][hh]);
    }
  }
  float* out = part + static_cast<size_t>(c) * 3 * H;
#pragma unroll
  for (int hh = 0; hh < H; ++hh) {
    const float m = block_max(mx[hh], red);
    float l = 0.f, t = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const float ex = cols[i] >= 0 ? expf(sc[i][hh] - m) : 0.f;
      l += ex;
      t = fmaf(ex, rr[i], t);
    }
    l = block_sum(l, red);
    t = block_sum(t, red);
    if (threadIdx.x == 0) { out[hh] = m; out[H + hh] = l; out[2 * H + hh] = t; }
  }
}

template <int H, int DK, bool DQ>
__global__ __launch_bounds__(kBlock) void attention_hub_ds_kernel(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                                                 const int* __restrict__ lc_row, const int* __restrict__ lc_begin,
                                                                 const int* __restrict__ lc_end, const int* __restrict__ lc_first,
                                                                 const float* __restrict__ q, const float* __restrict__ k, int ldqk,
                                                                 const RRef r, const float* __restrict__ edge_w,
                                                                 const float* __restrict__ scale_ptr, int scale_sigmoid,
                                                                 const float* __restrict__ part, float* __restrict__ ds,
                                                                 float* __restrict__ dqpart) {
  constexpr int A = H * DK;
  constexpr int AQ = DQ ? A : 1;
  constexpr int PER = GNPDE_LONG_ROW / kBlock;
  __shared__ float st[3 * H];                       // row maximum, 1 / denominator, c per head
  __shared__ float qpart[DQ ? kWavesPerBlock : 1][AQ];
  const int c = blockIdx.x;
  const int row = lc_row[c], b = lc_begin[c], e = lc_end[c];
  const int c0 = lc_first[c];
  const int nch = (rowptr[row + 1] - rowptr[row] + GNPDE_LONG_ROW - 1) / GNPDE_LONG_ROW;
  if (threadIdx.x < H) {                            // the row's statistics from its chunks' partials, chunk order
    const int hh = threadIdx.x;
    float m = -INFINITY;
    for (int i = 0; i < nch; ++i) m = fmaxf(m, part[static_cast<size_t>(c0 + i) * 3 * H + hh]);
    float l = 0.f, t = 0.f;
    for (int i = 0; i < nch; ++i) {
      const float* pc = part + static_cast<size_t>(c0 + i) * 3 * H;
      const float w = expf(pc[hh] - m);
      l = fmaf(pc[H + hh], w, l);
      t = fmaf(pc[2 * H + hh], w, t);
    }
    const float den = l + 1e-16f;
    st[hh] = m; st[H + hh] = den; st[2 * H + hh] = t / den;
  }
  __syncthreads();
  float scale = 1.0f / static_cast<float>(H);
  if (scale_ptr != nullptr) {
    float sv = *scale_ptr;
    if (scale_sigmoid) sv = 1.0f / (1.0f + expf(-sv));
    scale *= sv;
  }
  const float inv = 1.0f / sqrtf(static_cast<float>(DK));
  const float* qrow = q + static_cast<size_t>(row) * ldqk;
  float accq[AQ];
#pragma unroll
  for (int cc = 0; cc < AQ; ++cc) accq[cc] = 0.f;
  int cols[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) { const int p = b + i * kBlock + static_cast<int>(threadIdx.x); cols[i] = p < e ? colidx[p] : -1; }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    if (cols[i] < 0) continue;
    const int p = b + i * kBlock + static_cast<int>(threadIdx.x);
    const float ewp = edge_w != nullptr ? edge_w[p] : 1.f;
    const float* krow = k + static_cast<size_t>(cols[i]) * ldqk;
    float sc[H], out[H];
    row_scores<H, DK>(qrow, krow, inv, ewp, sc);
    const float rp = r[p];
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
      out[hh] = (expf(sc[hh] - st[hh]) / st[H + hh]) * scale * (rp - st[2 * H + hh]) * ewp;
      ds[static_cast<size_t>(p) * H + hh] = out[hh];
    }
    if constexpr (DQ) {
#pragma unroll
      for (int cc = 0; cc < A; cc += 4) {
        const float4 kv = *reinterpret_cast<const float4*>(krow + cc);
        const float wv = out[cc / DK];
        accq[cc + 0] = fmaf(wv, kv.x, accq[cc + 0]); accq[cc + 1] = fmaf(wv, kv.y, accq[cc + 1]);
        accq[cc + 2] = fmaf(wv, kv.z, accq[cc + 2]); accq[cc + 3] = fmaf(wv, kv.w, accq[cc + 3]);
      }
    }
  }
  if constexpr (DQ) {
    const int lane_ = threadIdx.x & (kWave - 1);
    constexpr int C = A < kWave ? A : kWave;
    constexpr int LPE = kWave / C;
#pragma unroll
    for (int ch = 0; ch < A / C; ++ch) {
      if constexpr (C > 1) LaneTranspose<C, kWave / 2>::run(accq + ch * C, lane_);
      float v = accq[ch * C];
#pragma unroll
      for (int m = LPE / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
      if ((lane_ % LPE) == 0) qpart[threadIdx.x >> 6][ch * C + lane_ / LPE] = v;
    }
    __syncthreads();
    if (threadIdx.x < A)
      dqpart[static_cast<size_t>(c) * A + threadIdx.x] = (qpart[0][threadIdx.x] + qpart[1][threadIdx.x]) + (qpart[2][threadIdx.x] + qpart[3][threadIdx.x]);
  }
}

// partial head sums of one 512-entry chunk of a hub segment: rowpart[c][0:A]
template <int H, int DK4>
__global__ __launch_bounds__(kBlock) void head_rowsum_chunk_kernel(const int* __restrict__ other, const int* __restrict__ pos,
                                                                  const int* __restrict__ lc_begin, const int* __restrict__ lc_end,
                                                                  const float* __restrict__ ds, const float* __restrict__ feat, int ldf,
                                                                  float* __restrict__ rowpart) {
  constexpr int A = H * DK4 * 4;
  constexpr int PER = GNPDE_LONG_ROW / kBlock;
  __shared__ float part[kWavesPerBlock][A];
  const int lane = threadIdx.x & (kWave - 1);
  const int c = blockIdx.x;
  const int b = lc_begin[c], e = lc_end[c];
  float acc[A];
#pragma unroll
  for (int a = 0; a < A; ++a) acc[a] = 0.f;
  int pp[PER], oo[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int t = b + i * kBlock + static_cast<int>(threadIdx.x);
    pp[i] = -1; oo[i] = 0;
    if (t < e) { pp[i] = pos != nullptr ? pos[t] : t; oo[i] = other[t]; }
  }
#pragma unroll
  for (int i = 0; i < PER; ++i)
    if (pp[i] >= 0) entry_fma<H, DK4>(ds, feat, ldf, H, pp[i], oo[i], acc);
  constexpr int C = A < kWave ? A : kWave;
  constexpr int LPE = kWave / C;
#pragma unroll
  for (int ch = 0; ch < A / C; ++ch) {
    if constexpr (C > 1) LaneTranspose<C, kWave / 2>::run(acc + ch * C, lane);
    float v = acc[ch * C];
#pragma unroll
    for (int m = LPE / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
    if ((lane % LPE) == 0) part[threadIdx.x >> 6][ch * C + lane / LPE] = v;
  }
  __syncthreads();
  if (threadIdx.x < A)
    rowpart[static_cast<size_t>(c) * A + threadIdx.x] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// out[hub row, a] = scale * sum of its chunks' partials, chunk order (one wavefront per hub row, lanes = columns)
__global__ __launch_bounds__(kWave) void hub_rowpart_fold_kernel(const int* __restrict__ long_rows, const int* __restrict__ long_chunk_ptr,
                                                                const float* __restrict__ rowpart, int A, float scale,
                                                                float* __restrict__ out, int ldo) {
  const int lr = blockIdx.x;
  const int row = long_rows[lr];
  const int c0 = long_chunk_ptr[lr], c1 = long_chunk_ptr[lr + 1];
  for (int a = threadIdx.x; a < A; a += kWave) {
    float t = 0.f;
    for (int c = c0; c < c1; ++c) t += rowpart[static_cast<size_t>(c) * A + a];
    out[static_cast<size_t>(row) * ldo + a] = scale * t;
  }
}

template <int H, int DK, bool DQ>
int launch_attention_rows_bwd_v(const gnpde_graph_t* g, const gnpde_attention_t* att, const float* r_csr, const float* scale,
                                int scale_sigmoid, float* ds_csr, float* dq, int lddq, float* hub_ws, hipStream_t s, const int* rpos) {
  const int n16 = g->n_bin16, n64 = g->n_bin64, nl = g->n_long_rows;
  // second class (17..512 entries, longest first): its trailing n_bin_le64 records (17..64 entries) take a ONE-pass launch (an entry per
  // lane, 1/8 of the score registers: twice the waves per SIMD and no predicated dead passes), the leading ones the 8-pass launch
  const int n_le64 = (g->n_bin_le64 > 0 && g->n_bin_le64 <= n64) ? g->n_bin_le64 : 0;
  const int n_gt64 = n64 - n_le64;
  if (n16 > 0) {
    constexpr int rows_per_block = kWavesPerBlock * (kWave / 16);
    const unsigned grid = static_cast<unsigned>((n16 + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL((attention_rows_bwd_kernel<H, DK, 16, 1, false, DQ>), dim3(grid), dim3(kBlock), 0, s, g->rowptr, g->colidx,
                       g->bin_rows, 0, n16, att->q, att->k, att->ldqk, RRef{r_csr, rpos}, att->edge_w_csr, scale, scale_sigmoid, g->long_rows, 0,
                       ds_csr, dq, lddq);
    GNPDE_LAUNCH_CHECK();
  }
  if (nl > 0 && hub_ws != nullptr && g->long_chunk_first != nullptr && g->n_long_chunks > 0) {
    // hubs as 512-entry chunks over the whole chip (scratch: [chunks][3 H] statistics, then [chunks][A] d q partials)
    const int nc = g->n_long_chunks;
    float* stats = hub_ws;
    float* dqpart = hub_ws + static_cast<size_t>(nc) * 3 * H;
    hipLaunchKernelGGL((attention_hub_stats_kernel<H, DK>), dim3(nc), dim3(kBlock), 0, s, g->colidx, g->long_chunk_row, g->long_chunk_begin,
                       g->long_chunk_end, att->q, att->k, att->ldqk, RRef{r_csr, rpos}, att->edge_w_csr, stats);
    GNPDE_LAUNCH_CHECK();
    hipLaunchKernelGGL((attention_hub_ds_kernel<H, DK, DQ>), dim3(nc), dim3(kBlock), 0, s, g->rowptr, g->colidx, g->long_chunk_row,
                       g->long_chunk_begin, g->long_chunk_end, g->long_chunk_first, att->q, att->k, att->ldqk, RRef{r_csr, rpos}, att->edge_w_csr, scale,
                       scale_sigmoid, stats, ds_csr, dqpart);
    GNPDE_LAUNCH_CHECK();
    if constexpr (DQ) {
      hipLaunchKernelGGL(hub_rowpart_fold_kernel, dim3(nl), dim3(kWave), 0, s, g->long_rows, g->long_chunk_ptr, dqpart, H * DK,
                         1.0f / sqrtf(static_cast<float>(DK)), dq, lddq);
      GNPDE_LAUNCH_CHECK();
    }
  } else if (nl > 0) {       // no scratch: one 1024-thread workgroup per hub
    hipLaunchKernelGGL((attention_hub_bwd_kernel<H, DK, DQ>), dim3(nl), dim3(kHubThreads), 0, s, g->rowptr, g->colidx, att->q, att->k,
                       att->ldqk, RRef{r_csr, rpos}, att->edge_w_csr, scale, scale_sigmoid, g->long_rows, ds_csr, dq, lddq);
    GNPDE_LAUNCH_CHECK();
  }
  if (n_gt64 > 0) {
    const unsigned grid = static_cast<unsigned>((n_gt64 + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL((attention_rows_bwd_kernel<H, DK, kWave, GNPDE_LONG_ROW / kWave, false, DQ>), dim3(grid), dim3(kBlock), 0, s,
                       g->rowptr, g->colidx, g->bin_rows, n16, n_gt64, att->q, att->k, att->ldqk, RRef{r_csr, rpos}, att->edge_w_csr, scale,
                       scale_sigmoid, g->long_rows, 0, ds_csr, dq, lddq);
    GNPDE_LAUNCH_CHECK();
  }
  if (n_le64 > 0) {
    const unsigned grid = static_cast<unsigned>((n_le64 + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL((attention_rows_bwd_kernel<H, DK, kWave, 1, false, DQ>), dim3(grid), dim3(kBlock), 0, s, g->rowptr, g->colidx,
                       g->bin_rows, n16 + n_gt64, n_le64, att->q, att->k, att->ldqk, RRef{r_csr, rpos}, att->edge_w_csr, scale, scale_sigmoid,
                       g->long_rows, 0, ds_csr, dq, lddq);
    GNPDE_LAUNCH_CHECK();
  }
  return 0;
}

template <int H, int DK>
int launch_attention_rows_bwd(const gnpde_graph_t* g, const gnpde_attention_t* att, const float* r_csr, const float* scale,
                              int scale_sigmoid, float* ds_csr, float* dq, int lddq, float* hub_ws, hipStream_t s, const int* rpos) {
  if constexpr (H * DK <= 32) {
    if (dq != nullptr) return launch_attention_rows_bwd_v<H, DK, true>(g, att, r_csr, scale, scale_sigmoid, ds_csr, dq, lddq, hub_ws, s, rpos);
  }
  return launch_attention_rows_bwd_v<H, DK, false>(g, att, r_csr, scale, scale_sigmoid, ds_csr, nullptr, 0, hub_ws, s, rpos);
}

// ------------------------------------------------------------------------------------------------
// Head-wise weighted segment sum with one LANE PER ENTRY (the adjoint stage's d q and d k; csrc/adjoint.hip):
//   out[seg, c] = scale * sum_{t in seg} ds[pos(t), head(c)] * feat[other(t), c]        c < A = H * DK4 * 4
// head_spmm_kernel above gives a wavefront to every segment and A / 4 lanes to every entry -- at A = 16 and a median of 8 entries
// per row one iteration of 16 entry slots, most of them empty, behind a chain of four dependent loads: 139 us per launch at the
// ogbn-arxiv shape for 220 MB of traffic.  Here the segments come from the degree-binned records (as in
// attention_rows_bwd_kernel): rows of <= 16 entries take 16 lanes (four rows per wavefront), a lane fetches its entry's ds vector
// and whole feature row (A floats in registers) with all loads of the wavefront in flight at once, and the A partial columns are
// summed over the lanes of the row by a transposing butterfly (A - 1 + log2(GL / A) shuffles instead of A log2 GL), after which
// lane gl holds column gl / (GL / A) and the row is written by one coalesced store.  Hub segments: one block each.
// pos == nullptr: ds is indexed by the segment's own positions (d q, rows of the graph); else through pos (d k: the segments
// are rows of the transposed graph and pos maps its positions to the CSR positions ds is stored at).
// ------------------------------------------------------------------------------------------------
template <int H, int DK4, int GL, int PER, bool HUBS>
__global__ __launch_bounds__(kBlock) void head_rowsum_kernel(const int* __restrict__ rowptr, const int* __restrict__ other,
                                                            const int* __restrict__ pos, const int* __restrict__ bin_rows, int first_rec,
                                                            int n_rec, const float* __restrict__ ds, const float* __restrict__ feat, int ldf,
                                                            float scale, const int* __restrict__ long_rows, int n_long,
                                                            float* __restrict__ out, int ldo) {
  constexpr int A = H * DK4 * 4;
  constexpr int RPW = kWave / GL;
  __shared__ float part[kWavesPerBlock][A];
  const int lane = threadIdx.x & (kWave - 1);
  const bool hub = HUBS && static_cast<int>(blockIdx.x) < n_long;
  float acc[A];
#pragma unroll
  for (int c = 0; c < A; ++c) acc[c] = 0.f;
  if (!hub) {
    const int gl = lane % GL;
    const long long rec = (static_cast<long long>(blockIdx.x) - (HUBS ? n_long : 0)) * (kWavesPerBlock * RPW) +
                          (threadIdx.x >> 6) * RPW + lane / GL;
    int4 info = make_int4(0, 0, 0, 0);
    if (rec < n_rec) info = reinterpret_cast<const int4*>(bin_rows)[first_rec + rec];
    const int b = info.y, e = info.y + info.z;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int t = b + gl + i * GL;
      if (t < e) entry_fma<H, DK4>(ds, feat, ldf, H, pos != nullptr ? pos[t] : t, other[t], acc);
    }
    // sum over the GL lanes of the row, C columns at a time
    constexpr int C = A < GL ? A : GL;
    constexpr int LPE = GL / C;
#pragma unroll
    for (int ch = 0; ch < A / C; ++ch) {
      if constexpr (C > 1) LaneTranspose<C, GL / 2>::run(acc + ch * C, gl);
      float v = acc[ch * C];
#pragma unroll
      for (int m = LPE / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
      if (rec < n_rec && (gl % LPE) == 0) out[static_cast<size_t>(info.x) * ldo + ch * C + gl / LPE] = scale * v;
    }
    return;
  }
  // hub segment: the block's 256 threads stride over its entries, then wave butterflies and a fold through the LDS
  const int seg = long_rows[blockIdx.x];
  const int b = rowptr[seg], e = rowptr[seg + 1];
  for (int t = b + static_cast<int>(threadIdx.x); t < e; t += kBlock)
    entry_fma<H, DK4>(ds, feat, ldf, H, pos != nullptr ? pos[t] : t, other[t], acc);
  constexpr int C = A < kWave ? A : kWave;
  constexpr int LPE = kWave / C;
#pragma unroll
  for (int ch = 0; ch < A / C; ++ch) {
    if constexpr (C > 1) LaneTranspose<C, kWave / 2>::run(acc + ch * C, lane);
    float v = acc[ch * C];
#pragma unroll
    for (int m = LPE / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
    if ((lane % LPE) == 0) part[threadIdx.x >> 6][ch * C + lane / LPE] = v;
  }
  __syncthreads();
  if (threadIdx.x < A)
    out[static_cast<size_t>(seg) * ldo + threadIdx.x] =
        scale * ((part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
}

template <int H, int DK4>
int launch_head_rowsum_hd(const gnpde_graph_t* g, const int* pos, const float* ds, const float* feat, int ldf, float scale, float* out,
                          int ldo, float* hub_ws, hipStream_t s) {
  const int n16 = g->n_bin16, n64 = g->n_bin64, nl = g->n_long_rows;
  const int n_le64 = (g->n_bin_le64 > 0 && g->n_bin_le64 <= n64) ? g->n_bin_le64 : 0;     // as launch_attention_rows_bwd_v
  const int n_gt64 = n64 - n_le64;
  if (n16 > 0) {
    constexpr int rows_per_block = kWavesPerBlock * (kWave / 16);
    const unsigned grid = static_cast<unsigned>((n16 + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL((head_rowsum_kernel<H, DK4, 16, 1, false>), dim3(grid), dim3(kBlock), 0, s, g->rowptr, g->colidx, pos, g->bin_rows, 0,
                       n16, ds, feat, ldf, scale, g->long_rows, 0, out, ldo);
    GNPDE_LAUNCH_CHECK();
  }
  if (nl > 0 && hub_ws != nullptr && g->n_long_chunks > 0) {      // hub segments as 512-entry chunks over the whole chip
    hipLaunchKernelGGL((head_rowsum_chunk_kernel<H, DK4>), dim3(g->n_long_chunks), dim3(kBlock), 0, s, g->colidx, pos, g->long_chunk_begin,
                       g->long_chunk_end, ds, feat, ldf, hub_ws);
    GNPDE_LAUNCH_CHECK();
    hipLaunchKernelGGL(hub_rowpart_fold_kernel, dim3(nl), dim3(kWave), 0, s, g->long_rows, g->long_chunk_ptr, hub_ws, H * DK4 * 4, scale, out, ldo);
    GNPDE_LAUNCH_CHECK();
  } else if (nl > 0) {
    hipLaunchKernelGGL((head_rowsum_hub_kernel<H, DK4>), dim3(nl), dim3(kHubThreads), 0, s, g->rowptr, g->colidx, pos, ds, feat, ldf, scale,
                       g->long_rows, out, ldo);
    GNPDE_LAUNCH_CHECK();
  }
  if (n_gt64 > 0) {
    const unsigned grid = static_cast<unsigned>((n_gt64 + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL((head_rowsum_kernel<H, DK4, kWave, GNPDE_LONG_ROW / kWave, false>), dim3(grid), dim3(kBlock), 0, s, g->rowptr,
                       g->colidx, pos, g->bin_rows, n16, n_gt64, ds, feat, ldf, scale, g->long_rows, 0, out, ldo);
    GNPDE_LAUNCH_CHECK();
  }
  if (n_le64 > 0) {
    const unsigned grid = static_cast<unsigned>((n_le64 + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL((head_rowsum_kernel<H, DK4, kWave, 1, false>), dim3(grid), dim3(kBlock), 0, s, g->rowptr, g->colidx, pos, g->bin_rows,
                       n16 + n_gt64, n_le64, ds, feat, ldf, scale, g->long_rows, 0, out, ldo);
    GNPDE_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace

// out[seg, 0:A] = scale * sum over the rows of `g` (entry t: ds[pos ? pos[t] : t, head], feat[colidx[t], :]); rows without entries are
// NOT written (the caller zeroes `out`).  Returns GNPDE_ESHAPE for head shapes without a kernel (the caller falls back to
// gnpde_head_spmm).
bool head_rowsum_supported(int heads, int dk) {
  if (dk % 4 != 0) return false;
  const int dk4 = dk / 4, a4 = heads * dk4;
  return (heads == 1 || heads == 2 || heads == 4 || heads == 8) && (dk4 == 1 || dk4 == 2 || dk4 == 4) && a4 <= 8;
}

size_t hub_bwd_workspace_floats(const gnpde_graph_t* g, int heads, int att_dim) {
  return static_cast<size_t>(g->n_long_chunks) * (3 * static_cast<size_t>(heads) + static_cast<size_t>(att_dim));
}

int launch_head_rowsum(const gnpde_graph_t* g, const int* pos, const float* ds, int heads, int dk, const float* feat, int ldf,
                       float scale, float* out, int ldo, float* hub_ws, hipStream_t s) {
  GNPDE_CHECK_ARG(g && ds && feat && out && head_rowsum_supported(heads, dk), GNPDE_ESHAPE, "head_rowsum: heads=%d d_k=%d has no kernel", heads, dk);
  GNPDE_CHECK_ARG(ldf % 4 == 0 && reinterpret_cast<uintptr_t>(feat) % 16 == 0 && reinterpret_cast<uintptr_t>(ds) % 16 == 0, GNPDE_EINVAL,
                  "head_rowsum: operands must be 16-byte aligned");
  GNPDE_CHECK_ARG(g->bin_rows != nullptr && (g->n_long_rows == 0 || g->long_rows), GNPDE_EINVAL, "head_rowsum: graph without degree bins");
  if (g->n == 0 || g->e == 0) return 0;
  const int dk4 = dk / 4;
#define GNPDE_HR(HH, DD) if (heads == HH && dk4 == DD) return launch_head_rowsum_hd<HH, DD>(g, pos, ds, feat, ldf, scale, out, ldo, hub_ws, s);
  GNPDE_HR(1, 1) GNPDE_HR(1, 2) GNPDE_HR(1, 4) GNPDE_HR(2, 1) GNPDE_HR(2, 2) GNPDE_HR(2, 4) GNPDE_HR(4, 1) GNPDE_HR(4, 2) GNPDE_HR(8, 1)
#undef GNPDE_HR
  return GNPDE_ESHAPE;
}

namespace {
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
  hipStream_t s = static_cast<hipStream_t>(stream);
  // one launch: a wavefront per segment, hub segments by whole blocks.  (A split into a 16-lane class for segments of
  // <= 64 entries plus a wavefront class for the rest was measured slower: 60 + 110 us against 136 us at the
  // ogbn-arxiv shape -- both launches walk all segment pointers; the template keeps the option.)
  const unsigned grid64 = static_cast<unsigned>(nl + (g->n + kWavesPerBlock - 1) / kWavesPerBlock);
#define GNPDE_HS_ARGS(LO, HI, NL) segptr, segpos, other, ds_csr, heads, dk, feat, ldf, scale, g->n, LO, HI, long_segs, NL, out, ldo
#define GNPDE_HS(N4) \
  hipLaunchKernelGGL((head_spmm_kernel<N4, 64, true>), dim3(grid64), dim3(kBlock), 0, s, GNPDE_HS_ARGS(0, GNPDE_LONG_ROW, nl));
  switch (a4) {
    case 1: GNPDE_HS(1); break;
    case 2: GNPDE_HS(2); break;
    case 4: GNPDE_HS(4); break;
    case 8: GNPDE_HS(8); break;
    case 16: GNPDE_HS(16); break;
    case 32: GNPDE_HS(32); break;
    default: GNPDE_HS(64); break;
  }
#undef GNPDE_HS_ARGS
#undef GNPDE_HS
  GNPDE_LAUNCH_CHECK();
  return 0;
}

namespace gnpde {
// dq != nullptr (head shapes with heads * d_k <= 32 only, see attention_rows_bwd_dq_supported): also d q[row, 0:A] into dq (rows
// without entries are not written)
bool attention_rows_bwd_dq_supported(int heads, int dk) {
  return (heads == 1 || heads == 2 || heads == 4 || heads == 8) && (dk == 4 || dk == 8 || dk == 16) && heads * dk <= 32;
}

int launch_attention_rows_bwd_dq(const gnpde_graph_t* g, const gnpde_attention_t* att, const float* r_csr, const float* scale,
                                 int32_t scale_sigmoid, float* ds_csr, float* dq, int lddq, float* hub_ws, hipStream_t s, const int* rpos) {
  GNPDE_CHECK_ARG(g && att && r_csr && ds_csr && att->q && att->k, GNPDE_EINVAL, "attention_rows_bwd: null argument");
  GNPDE_CHECK_ARG(att->type == GNPDE_ATT_SCALED_DOT && att->norm_idx == 0 && !att->square_plus, GNPDE_ESHAPE,
                  "attention_rows_bwd: only scaled-dot attention with a softmax over the row");
  GNPDE_CHECK_ARG(att->heads >= 1 && att->att_dim % att->heads == 0, GNPDE_EINVAL, "attention_rows_bwd: bad head count");
  GNPDE_CHECK_ARG(att->ldqk % 4 == 0 && reinterpret_cast<uintptr_t>(att->q) % 16 == 0 && reinterpret_cast<uintptr_t>(att->k) % 16 == 0,
                  GNPDE_EINVAL, "attention_rows_bwd: q / k must be 16-byte aligned with ld %% 4 == 0");
  GNPDE_CHECK_ARG(g->n_long_rows == 0 || g->long_rows, GNPDE_EINVAL, "attention_rows_bwd: graph has long rows but no list of them");
  if (g->n == 0 || g->e == 0) return 0;
  GNPDE_CHECK_ARG(g->bin_rows != nullptr, GNPDE_EINVAL, "attention_rows_bwd: graph has no degree-binned row records");
  const int h = att->heads, dk = att->att_dim / att->heads;
  GNPDE_CHECK_ARG(dq == nullptr || attention_rows_bwd_dq_supported(h, dk), GNPDE_ESHAPE, "attention_rows_bwd: d q fusion needs heads * d_k <= 32");
#define GNPDE_AB(HH, DD) \
  if (h == HH && dk == DD) return launch_attention_rows_bwd<HH, DD>(g, att, r_csr, scale, scale_sigmoid, ds_csr, dq, lddq, hub_ws, s, rpos);
  GNPDE_AB(1, 4) GNPDE_AB(1, 8) GNPDE_AB(1, 16) GNPDE_AB(2, 4) GNPDE_AB(2, 8) GNPDE_AB(2, 16) GNPDE_AB(4, 4) GNPDE_AB(4, 8)
  GNPDE_AB(4, 16) GNPDE_AB(8, 4) GNPDE_AB(8, 8) GNPDE_AB(8, 16)
#undef GNPDE_AB
  set_error("attention_rows_bwd: no kernel for heads=%d, d_k=%d (heads in {1,2,4,8}, d_k in {4,8,16})", h, dk);
  return GNPDE_ESHAPE;
}
}  // namespace gnpde

extern "C" int gnpde_attention_rows_bwd(const gnpde_graph_t* g, const gnpde_attention_t* att, const float* r_csr, const float* scale,
                                        int32_t scale_sigmoid, float* ds_csr, void* stream) {
  return gnpde::launch_attention_rows_bwd_dq(g, att, r_csr, scale, scale_sigmoid, ds_csr, nullptr, 0, nullptr, static_cast<hipStream_t>(stream), nullptr);
}
