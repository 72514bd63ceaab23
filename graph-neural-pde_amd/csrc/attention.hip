// Edge attention: per-edge multi-head scores -> per-node (row or column) softmax / squareplus ->
// head-mean weights in CSR order (+ the reference's [E,h] outputs in the caller's edge order).
//
// Replaces SpGraphTransAttentionLayer.forward (reference src/function_transformer_attention.py:
// 190-213), SpGraphAttentionLayer.forward (src/function_GAT_attention.py:111-114),
// torch_geometric.utils.softmax [3P] and utils.squareplus (src/utils.py:179-208).  The reference
// materialises q[edge[0]] and k[edge[1]] as two [E,d_k,h] tensors, their product, and runs
// scatter_max / scatter_add (atomics) per head; here a lane owns one (edge, head) pair, gathers its
// d_k-slice of q and k straight from the [N,A] projections (16-byte loads), and the segment
// statistics are wave reductions over the CSR (norm_idx=0) or CSC (norm_idx=1) segment -- no atomics
// except one order-independent max per block for squareplus' global maximum.
//
// General path (any attention_norm_idx, squareplus, [E,h] outputs):
//   Pass 1  scores[p,h]   (CSR order)            edge-parallel, balanced regardless of degree skew
//   Pass 2  seg stats     m[seg,h], den[seg,h]   one wavefront per segment, one block per long segment
//   Pass 3  normalise     w[p] = mean_h att      edge-parallel; optional scatter to edge order
// Fused path (the solver's hot case: softmax over the ROW, only the head-mean weights wanted):
//   one launch; a 16-lane group (rows with <= 16 entries, 4 rows per wavefront) or a whole wavefront
//   (<= GNPDE_LONG_ROW entries) owns a row, lanes own edges, the h scores of up to 8 passes stay in
//   registers, max / sum are DPP-style xor reductions inside the group, w is written once.  Long
//   rows run the general passes restricted to their chunks.
#include <cmath>
#include "common.h"
#include "rhs.h"

namespace gnpde {
namespace {

constexpr int kGmaxSlots = 64;
constexpr int kGmaxPartCap = 8192;      // waves of a "small" squareplus maximum sweep (AttArgs::gmax_part)

struct AttArgs {
  int n, e, h, dk, type, norm_idx, square_plus;
  float inv_sqrt_dk_den;  // sqrt(d_k), divisor of the scaled dot product
  float scale_mul;        // fl32(1 / sqrt(d_k)): the row kernels multiply (exactly the quotient when sqrt(d_k) is a power of two: d_k = 4, 16, 64; within 1 ulp of it otherwise)
  float leaky_slope;
  const int* __restrict__ rowidx;
  const int* __restrict__ colidx;
  const int* __restrict__ perm;
  const int* __restrict__ segptr;   // rowptr or cscptr
  const int* __restrict__ segpos;   // nullptr (rows: positions are contiguous) or cscpos
  const float* __restrict__ q;
  const float* __restrict__ k;
  int ldqk;
  const float* __restrict__ gat_terms;  // [n, 2h]: src terms then dst terms
  const float* __restrict__ output_var;
  const float* __restrict__ lengthscale;
  const float* __restrict__ edge_w;
  float* scores;      // [e,h]
  float* seg_m;       // [n,h]
  float* seg_den;     // [n,h]
  unsigned* gmax;     // ordered-uint encoding of the global max score
  unsigned* gmax_slots;   // squareplus maximum sweep of the fused path: kGmaxSlots partial maxima, 128 bytes apart (a block updates slot
                          // blockIdx % kGmaxSlots; gmax_fold_kernel folds them into *gmax).  A small graph's whole grid is resident at once,
                          // so every wave sees the initial value and issues its atomic: on ONE address they serialise at ~10 ns each --
                          // 24-36 us of a 48-us Cora evaluation (profiles/r05_c2_as_run_sequence_before.txt).  nullptr: straight into *gmax.
  float* w_mean;      // [e] or null
  float* att_edge;    // [E,h] or null
  float* prods_edge;  // [E,h] or null
  // when non-null the edge-parallel passes work on these CSR ranges only (one block per chunk)
  const int* __restrict__ chunk_begin;
  const int* __restrict__ chunk_end;
  // long segments handled by seg_stats_long_kernel (segment ids); others by the wave-per-segment kernel
  const int* __restrict__ long_segs;
  const int* __restrict__ rowptr;
  const int* __restrict__ bin_rows;
  int hub_fold_lds;   // default on; gnpde_tune(11, 2) folds straight from memory: see hub_normalise_body
  // fused scaled-dot segment kernels beyond the plain row softmax (row_attention_sd_kernel and its hub phases):
  const int* __restrict__ out_pos;   // nullptr, or position e of the walked graph -> position of the weight in w_mean (the walked graph
                                     // is the TRANSPOSED one when the normalisation runs over columns)
  int sp_mode;                       // 0 softmax; 1 squareplus with the global maximum in *gmax; 2 only form that maximum
  // squareplus on a SMALL grid (at most kGmaxPartCap waves in the maximum sweep, no hub rows): every wave of the sweep stores its
  // maximum in gmax_part[gmax_base + wave] (no atomic, nothing to clear beforehand) and every block of the second sweep folds the
  // gmax_count entries itself (block 0 also publishes *gmax for the kernels behind) -- no memset node and no fold launch, two of
  // the eight launches of such an evaluation (Cora as run_GNN.py runs it: ~4.5 us each)
  unsigned* gmax_part;
  int gmax_base, gmax_count;
};

__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
  const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

// the partial maxima of the squareplus sweep (AttArgs::gmax_slots) -> *gmax; one wavefront
__global__ __launch_bounds__(kWave) void gmax_fold_kernel(const unsigned* __restrict__ slots, unsigned* __restrict__ gmax) {
  unsigned v = slots[32 * static_cast<int>(threadIdx.x)];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const unsigned w = static_cast<unsigned>(__shfl_xor(static_cast<int>(v), o, kWave));
    v = w > v ? w : v;
  }
  if (threadIdx.x == 0 && v > *gmax) *gmax = v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// ---- pass 0 (GAT): per-node terms  src[i,h] = sum_c a[c] h_i[c,head],  dst[i,h] = sum_c a[d_k+c] h_i[c,head]
// `table` (optional, [n, 8h]): the same terms as 4-wide query / key vectors for the scaled-dot row kernels --
//   q_i[head] = (2 src, 2, 0, 0),  k_j[head] = (1, dst, 0, 0):  q . k / sqrt(4) = src_i + dst_j  (bit for bit: 2 src and 2 dst are
//   exact, their sum is rounded once, halving is exact), so GAT's row softmax rides row_attention_sd_kernel (MODE 3: leaky ReLU on the
//   score) instead of the generic row kernel + separate hub launches: 103 -> 47 us of attention per evaluation at the ogbn-arxiv shape
__global__ __launch_bounds__(kBlock) void gat_terms_kernel(const float* __restrict__ wx, int ld, const float* __restrict__ a,
                                                          int n, int h, int dk, float* __restrict__ terms, float* __restrict__ table) {
  const long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= static_cast<long long>(n) * h) return;
  const int i = static_cast<int>(idx / h), head = static_cast<int>(idx % h);
  const float* row = wx + static_cast<size_t>(i) * ld + head * dk;
  float s = 0.f, t = 0.f;
  for (int c = 0; c < dk; ++c) {
    const float v = row[c];
    s = fmaf(a[c], v, s);
    t = fmaf(a[dk + c], v, t);
  }
  terms[static_cast<size_t>(i) * 2 * h + head] = s;
  terms[static_cast<size_t>(i) * 2 * h + h + head] = t;
  if (table != nullptr) {
    float* row = table + static_cast<size_t>(i) * 8 * h;
    *reinterpret_cast<float4*>(row + 4 * head) = make_float4(2.0f * s, 2.0f, 0.f, 0.f);
    *reinterpret_cast<float4*>(row + 4 * h + 4 * head) = make_float4(1.0f, t, 0.f, 0.f);
  }
}

// score of one (edge, head): q-slice of the row node against k-slice of the column node
template <int TYPE, bool VEC4>
__device__ __forceinline__ float edge_score(const AttArgs& a, int p, int r, int c, int head) {
  float s;
  if constexpr (TYPE == GNPDE_ATT_GAT) {
    const float v = a.gat_terms[static_cast<size_t>(r) * 2 * a.h + head] +
                    a.gat_terms[static_cast<size_t>(c) * 2 * a.h + a.h + head];
    s = v > 0.f ? v : v * a.leaky_slope;
  } else {
    const float* qp = a.q + static_cast<size_t>(r) * a.ldqk + head * a.dk;
    const float* kp = a.k + static_cast<size_t>(c) * a.ldqk + head * a.dk;
    float mq = 0.f, mk = 0.f;
    if constexpr (TYPE == GNPDE_ATT_PEARSON) {
      for (int j = 0; j < a.dk; ++j) { mq += qp[j]; mk += kp[j]; }
      mq = mq / static_cast<float>(a.dk);
      mk = mk / static_cast<float>(a.dk);
    }
    float dot = 0.f, nq = 0.f, nk = 0.f;
    auto term = [&](float qv, float kv) {
      if constexpr (TYPE == GNPDE_ATT_SCALED_DOT) {
        dot = fmaf(qv, kv, dot);
      } else if constexpr (TYPE == GNPDE_ATT_EXP_KERNEL) {
        const float df = qv - kv;
        dot = fmaf(df, df, dot);
      } else {
        qv -= mq; kv -= mk;
        dot = fmaf(qv, kv, dot);
        nq = fmaf(qv, qv, nq);
        nk = fmaf(kv, kv, nk);
      }
    };
    if constexpr (VEC4) {
      for (int j = 0; j < a.dk; j += 4) {
        const float4 qv = *reinterpret_cast<const float4*>(qp + j);
        const float4 kv = *reinterpret_cast<const float4*>(kp + j);
        term(qv.x, kv.x); term(qv.y, kv.y); term(qv.z, kv.z); term(qv.w, kv.w);
      }
    } else {
      for (int j = 0; j < a.dk; ++j) term(qp[j], kp[j]);
    }
    if constexpr (TYPE == GNPDE_ATT_SCALED_DOT) {
      s = dot * a.scale_mul;      // fl32(1 / sqrt(d_k)): exactly the quotient for d_k = 4, 16, 64, within 1 ulp otherwise
    } else if constexpr (TYPE == GNPDE_ATT_EXP_KERNEL) {
      const float ov = *a.output_var, ls = *a.lengthscale;
      s = (ov * ov) * expf(-(dot / (2.0f * (ls * ls))));
    } else {
      // cosine / pearson: x1.x2 / (max(|x1|, eps) max(|x2|, eps)), eps = 1e-5 -- EACH norm clamped, torch >= 1.12's cosine_similarity (the
      // torch the golden vectors were recorded with) and what the per-vector normalisation of the fused path expresses (misc.hip
      // normalise_heads_kernel), so that the layer's [E,h] attention and ODEFunc.forward agree for degenerate rows too.  torch 1.8 (the
      // reference's pin) clamps the PRODUCT at eps instead: the two differ only when |x1| |x2| < 1e-5 or one norm alone is below 1e-5
      // (tests/test_properties_cpu.py::test_cosine_clamp_against_torch_1_8_product_clamp).  Until round 5 this kernel
      // used the product clamp and disagreed with the fused path by up to 1e-3 on such rows.
      s = dot / (fmaxf(sqrtf(nq), 1e-5f) * fmaxf(sqrtf(nk), 1e-5f));
    }
  }
  if (a.edge_w != nullptr) s = s * a.edge_w[p];
  return s;
}

// ---- pass 1: one lane per (CSR position, head); over all edges (grid-stride) or one block per chunk
template <int TYPE, bool VEC4>
__global__ __launch_bounds__(kBlock) void scores_kernel(const AttArgs a) {
  long long first, total, stride;
  if (a.chunk_begin != nullptr) {
    first = static_cast<long long>(a.chunk_begin[blockIdx.x]) * a.h + threadIdx.x;
    total = static_cast<long long>(a.chunk_end[blockIdx.x]) * a.h;
    stride = blockDim.x;
  } else {
    first = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    total = static_cast<long long>(a.e) * a.h;
    stride = static_cast<long long>(gridDim.x) * blockDim.x;
  }
  float lmax = -INFINITY;
  for (long long idx = first; idx < total; idx += stride) {
    const int p = static_cast<int>(idx / a.h);
    const int head = static_cast<int>(idx - static_cast<long long>(p) * a.h);
    const float s = edge_score<TYPE, VEC4>(a, p, a.rowidx[p], a.colidx[p], head);
    a.scores[idx] = s;
    lmax = fmaxf(lmax, s);
  }
  if (a.square_plus) {  // block max -> one order-independent atomic
    __shared__ float smax[kWavesPerBlock];
    lmax = wave_max(lmax);
    if ((threadIdx.x & (kWave - 1)) == 0) smax[threadIdx.x >> 6] = lmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = smax[0];
      for (int i = 1; i < kWavesPerBlock; ++i) m = fmaxf(m, smax[i]);
      atomicMax(a.gmax, f2ord(m));
    }
  }
}

__device__ __forceinline__ float squareplus_num(float s, float gmax) {
  const float z = s - gmax;
  return (z + sqrtf(z * z + 4.0f)) / 2.0f;
}
// the same with v_sqrt_f32 (1 ulp) and an exact halving: the fused row kernel is bound by VALU issue, and the IEEE square root is a
// dozen instructions per (entry, head)
__device__ __forceinline__ float squareplus_num_fast(float s, float gmax) {
  const float z = s - gmax;
  return (z + __builtin_amdgcn_sqrtf(fmaf(z, z, 4.0f))) * 0.5f;
}

// ---- pass 2: one wavefront per segment (heads in an outer loop); segments longer than GNPDE_LONG_ROW
// are left to seg_stats_long_kernel so that a hub does not serialise on one wave
__device__ __forceinline__ float stat_term(const AttArgs& a, int t, int head, float gmax, float m) {
  const int p = a.segpos ? a.segpos[t] : t;
  const float s = a.scores[static_cast<size_t>(p) * a.h + head];
  if (a.square_plus) return squareplus_num(s, gmax);
  return expf(s - m);
}

__global__ __launch_bounds__(kBlock) void seg_stats_kernel(const AttArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int seg = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6));
  if (seg >= a.n) return;
  const int b = a.segptr[seg], e = a.segptr[seg + 1];
  if (e - b > GNPDE_LONG_ROW && a.long_segs != nullptr) return;
  const float gmax = a.square_plus ? ord2f(*a.gmax) : 0.f;
  for (int head = 0; head < a.h; ++head) {
    float m = 0.f;
    if (!a.square_plus) {
      float mx = -INFINITY;
      for (int t = b + lane; t < e; t += kWave) {
        const int p = a.segpos ? a.segpos[t] : t;
        mx = fmaxf(mx, a.scores[static_cast<size_t>(p) * a.h + head]);
      }
      mx = wave_max(mx);
      m = (e > b) ? mx : 0.f;
    }
    float sum = 0.f;
    for (int t = b + lane; t < e; t += kWave) sum += stat_term(a, t, head, gmax, m);
    const float den = wave_sum(sum) + 1e-16f;
    if (lane == 0) {
      a.seg_m[static_cast<size_t>(seg) * a.h + head] = m;
      a.seg_den[static_cast<size_t>(seg) * a.h + head] = den;
    }
  }
}

// Head-parallel variant for h in {1,2,4,8}: lane = (entry slot, head) with the head fastest, so the h scores of
// an entry are one contiguous 4h-byte read (instead of h strided passes over the segment) and both sweeps
// cover all heads at once.  Reductions over the entry slots are xor butterflies with stride H.
template <int H>
__global__ __launch_bounds__(kBlock) void seg_stats_heads_kernel(const AttArgs a) {
  constexpr int ES = kWave / H;  // entries per wave pass
  const int lane = threadIdx.x & (kWave - 1);
  const int seg = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6));
  if (seg >= a.n) return;
  const int b = a.segptr[seg], e = a.segptr[seg + 1];
  if (e - b > GNPDE_LONG_ROW && a.long_segs != nullptr) return;
  const int es = lane / H, head = lane % H;
  const float gmax = a.square_plus ? ord2f(*a.gmax) : 0.f;
  float m = 0.f;
  if (!a.square_plus) {
    float mx = -INFINITY;
    for (int t = b + es; t < e; t += ES) {
      const int p = a.segpos ? a.segpos[t] : t;
      mx = fmaxf(mx, a.scores[static_cast<size_t>(p) * H + head]);
    }
#pragma unroll
    for (int off = H; off < kWave; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, kWave));
    m = (e > b) ? mx : 0.f;
  }
  float sum = 0.f;
  for (int t = b + es; t < e; t += ES) {
    const int p = a.segpos ? a.segpos[t] : t;
    const float sv = a.scores[static_cast<size_t>(p) * H + head];
    sum += a.square_plus ? squareplus_num(sv, gmax) : expf(sv - m);
  }
#pragma unroll
  for (int off = H; off < kWave; off <<= 1) sum += __shfl_xor(sum, off, kWave);
  if (es == 0) {
    a.seg_m[static_cast<size_t>(seg) * H + head] = m;
    a.seg_den[static_cast<size_t>(seg) * H + head] = sum + 1e-16f;
  }
}

// Long segments: blockIdx.x = long segment, blockIdx.y = 512-entry chunk of it.  Every chunk leaves an
// online-softmax partial (m_c, l_c) per head (squareplus: just the partial sum); seg_stats_long_combine
// folds them:  m = max_c m_c,  den = sum_c l_c exp(m_c - m) + 1e-16.
__global__ __launch_bounds__(kBlock) void seg_stats_long_partial_kernel(const AttArgs a, float* __restrict__ part, int max_chunks) {
  __shared__ float red[kWavesPerBlock];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int seg = a.long_segs[blockIdx.x];
  const int b = a.segptr[seg] + static_cast<int>(blockIdx.y) * GNPDE_LONG_ROW;
  const int e = min(a.segptr[seg + 1], b + GNPDE_LONG_ROW);
  if (b >= a.segptr[seg + 1]) return;
  const float gmax = a.square_plus ? ord2f(*a.gmax) : 0.f;
  float* out = part + (static_cast<size_t>(blockIdx.x) * max_chunks + blockIdx.y) * 2 * a.h;
  for (int head = 0; head < a.h; ++head) {
    float m = 0.f;
    if (!a.square_plus) {
      float mx = -INFINITY;
      for (int t = b + threadIdx.x; t < e; t += kBlock) {
        const int p = a.segpos ? a.segpos[t] : t;
        mx = fmaxf(mx, a.scores[static_cast<size_t>(p) * a.h + head]);
      }
      mx = wave_max(mx);
      if (lane == 0) red[wave] = mx;
      __syncthreads();
      m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      __syncthreads();
    }
    float sum = 0.f;
    for (int t = b + threadIdx.x; t < e; t += kBlock) sum += stat_term(a, t, head, gmax, m);
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
      out[head] = m;
      out[a.h + head] = (red[0] + red[1]) + (red[2] + red[3]);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kWave) void seg_stats_long_combine_kernel(const AttArgs a, const float* __restrict__ part,
                                                                      int max_chunks) {
  const int seg = a.long_segs[blockIdx.x];
  const int len = a.segptr[seg + 1] - a.segptr[seg];
  const int nch = (len + GNPDE_LONG_ROW - 1) / GNPDE_LONG_ROW;
  const float* base = part + static_cast<size_t>(blockIdx.x) * max_chunks * 2 * a.h;
  for (int head = threadIdx.x; head < a.h; head += blockDim.x) {
    float m = -INFINITY;
    if (a.square_plus) m = 0.f;
    else
      for (int c = 0; c < nch; ++c) m = fmaxf(m, base[static_cast<size_t>(c) * 2 * a.h + head]);
    float l = 0.f;
    for (int c = 0; c < nch; ++c) {
      const float lc = base[static_cast<size_t>(c) * 2 * a.h + a.h + head];
      l += a.square_plus ? lc : lc * expf(base[static_cast<size_t>(c) * 2 * a.h + head] - m);
    }
    a.seg_m[static_cast<size_t>(seg) * a.h + head] = m;
    a.seg_den[static_cast<size_t>(seg) * a.h + head] = l + 1e-16f;
  }
}

// ---- pass 3: one lane per CSR position, heads serial (same order as attention.mean(dim=1))
__global__ __launch_bounds__(kBlock) void normalise_kernel(const AttArgs a) {
  long long first, last, stride;
  if (a.chunk_begin != nullptr) {
    first = static_cast<long long>(a.chunk_begin[blockIdx.x]) + threadIdx.x;
    last = a.chunk_end[blockIdx.x];
    stride = blockDim.x;
  } else {
    first = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    last = a.e;
    stride = static_cast<long long>(gridDim.x) * blockDim.x;
  }
  const float gmax = a.square_plus ? ord2f(*a.gmax) : 0.f;
  for (long long p = first; p < last; p += stride) {
    const int seg = a.norm_idx == 0 ? a.rowidx[p] : a.colidx[p];
    const long long dst = (a.att_edge || a.prods_edge) ? static_cast<long long>(a.perm[p]) * a.h : 0;
    float acc = 0.f;
    for (int head = 0; head < a.h; ++head) {
      const float s = a.scores[p * a.h + head];
      const float den = a.seg_den[static_cast<size_t>(seg) * a.h + head];
      float v;
      if (a.square_plus) v = squareplus_num(s, gmax) / den;
      else v = expf(s - a.seg_m[static_cast<size_t>(seg) * a.h + head]) / den;
      acc += v;
      if (a.att_edge) a.att_edge[dst + head] = v;
      if (a.prods_edge) a.prods_edge[dst + head] = s;
    }
    if (a.w_mean) a.w_mean[p] = acc / static_cast<float>(a.h);
  }
}

// ---- hub rows of the fused path (softmax over the row, head-mean weights only), two launches:
// (a) one block per 512-entry chunk: scores -> scratch, chunk maximum and sum of exponentials per head
// (b) one block per chunk: fold the row's chunk partials (<= max_chunks of them) and write the weights
template <int TYPE, bool VEC4>
__device__ __forceinline__ void hub_scores_partial_body(const AttArgs& a, float* __restrict__ part, int chunk) {
  __shared__ float red[kWavesPerBlock];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int b = a.chunk_begin[chunk], e = a.chunk_end[chunk];
  const int row = a.rowidx[b];
  float* out = part + static_cast<size_t>(chunk) * 2 * a.h;
  for (int head = 0; head < a.h; ++head) {
    float sv[GNPDE_LONG_ROW / kBlock];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < GNPDE_LONG_ROW / kBlock; ++i) {
      const int p = b + i * kBlock + threadIdx.x;
      sv[i] = -INFINITY;
      if (p < e) {
        sv[i] = edge_score<TYPE, VEC4>(a, p, row, a.colidx[p], head);
        a.scores[static_cast<size_t>(p) * a.h + head] = sv[i];
        mx = fmaxf(mx, sv[i]);
      }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < GNPDE_LONG_ROW / kBlock; ++i) sum += expf(sv[i] - m);
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
      out[head] = m;
      out[a.h + head] = (red[0] + red[1]) + (red[2] + red[3]);
    }
    __syncthreads();
  }
}

// Head-parallel form for H in {1,2,4,8} (the hot case): lane = (entry, head) with the head fastest, so the H lanes of an entry
// read ONE contiguous A-float row of k (round 1 looped over the heads and gathered every k row H times, 64 B at a time: 6.9 ms
// for the 99.5 k hub chunks of the R-MAT graph); the scores of the PER passes stay in registers, the per-head maximum / sum
// are xor butterflies over the lanes with equal head and a fold over the four waves in LDS.
template <int TYPE, bool VEC4, int H, int MODE = 0>
__device__ __forceinline__ void hub_scores_partial_heads(const AttArgs& a, float* __restrict__ part, int chunk) {
  constexpr int EPB = kBlock / H;                 // entries per block pass
  constexpr int PER = GNPDE_LONG_ROW / EPB;       // passes
  __shared__ float red[kWavesPerBlock][H];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int head = threadIdx.x % H, slot = threadIdx.x / H;
  const int b = a.chunk_begin[chunk], e = a.chunk_end[chunk];
  const int row = a.rowidx[b];
  float* out = part + static_cast<size_t>(chunk) * 2 * H;
  int cols[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {                 // all column ids first, then all k gathers
    const int p = b + i * EPB + slot;
    cols[i] = p < e ? a.colidx[p] : -1;
  }
  float sv[PER];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int p = b + i * EPB + slot;
    sv[i] = -INFINITY;
    if (cols[i] >= 0) {
      sv[i] = edge_score<TYPE, VEC4>(a, p, row, cols[i], head);
      if constexpr (MODE == 3) sv[i] = sv[i] > 0.f ? sv[i] : sv[i] * a.leaky_slope;
      a.scores[static_cast<size_t>(p) * H + head] = sv[i];
      mx = fmaxf(mx, sv[i]);
    }
  }
#pragma unroll
  for (int off = H; off < kWave; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, kWave));
  if (lane < H) red[wave][lane] = mx;
  __syncthreads();
  const float m = fmaxf(fmaxf(red[0][head], red[1][head]), fmaxf(red[2][head], red[3][head]));
  __syncthreads();
  if constexpr (MODE == 2) {     // squareplus, first sweep: only the global maximum of the scores (utils.py:196 `src.max()`)
    if (threadIdx.x < H && m > -INFINITY) {
      const unsigned mine = f2ord(m);
      unsigned* tgt = a.gmax_slots != nullptr ? a.gmax_slots + 32 * (blockIdx.x % kGmaxSlots) : a.gmax;
      if (mine > __hip_atomic_load(tgt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(tgt, mine);
    }
    return;
  }
  float sum = 0.f;
  if constexpr (MODE == 1) {     // squareplus: the chunk's share of the segment sum of u = (z + sqrt(z^2 + 4)) / 2, z = s - max
    const float gm = ord2f(*a.gmax);
#pragma unroll
    for (int i = 0; i < PER; ++i) sum += cols[i] >= 0 ? squareplus_num(sv[i], gm) : 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < PER; ++i) sum += __builtin_amdgcn_exp2f((sv[i] - m) * 1.44269504088896341f);   // v_exp_f32; exp(-inf) = 0 for the absent entries
  }
#pragma unroll
  for (int off = H; off < kWave; off <<= 1) sum += __shfl_xor(sum, off, kWave);
  if (lane < H) red[wave][lane] = sum;
  __syncthreads();
  if (threadIdx.x < H) {
    out[threadIdx.x] = m;
    out[H + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}

template <int TYPE, bool VEC4>
__global__ __launch_bounds__(kBlock) void hub_scores_partial_kernel(const AttArgs a, float* __restrict__ part) {
  hub_scores_partial_body<TYPE, VEC4>(a, part, blockIdx.x);
}

template <int MODE = 0, bool SCATTER = false>
__device__ __forceinline__ void hub_normalise_body(const AttArgs& a, const float* __restrict__ part,
                                                   const int* __restrict__ long_chunk_row_first, int c) {
  // long_chunk_row_first[c] = index of the first chunk of the row chunk c belongs to
  __shared__ float st[2 * 64];  // [2h]: row maximum and denominator per head
  constexpr int kStage = 2048;  // floats of chunk partials staged through LDS (a row of up to 2048 / 2h chunks)
  __shared__ float sp_[kStage];
  const int b = a.chunk_begin[c], e = a.chunk_end[c];
  const int row = a.rowidx[b];
  const int c0 = long_chunk_row_first[c];
  const int nch = (a.rowptr[row + 1] - a.rowptr[row] + GNPDE_LONG_ROW - 1) / GNPDE_LONG_ROW;
  // The fold below is a chain of 2 nch DEPENDENT-latency loads issued by h threads (the compiler keeps two in flight): 22
  // chunks of the largest hub of the ogbn-arxiv shape = ~20 round trips before any of the block's 512 weights can be written,
  // repeated by every chunk block of the row.  Staged form (the default; measured round 3, profiles/r03_v0_blind_changes_measured.txt: equal at the ogbn-arxiv shape,
  // 4.85 -> 4.68 ms per attention at the R-MAT shape; gnpde_tune(11, 2) selects the fold from memory): the whole
  // block fetches the row's partials in ONE round trip into LDS and the h folding threads read them from there -- the same
  // values combined in the same order, so the weights are bit-identical.
  const int nval = nch * 2 * a.h;                   // block-uniform
  const bool staged = a.hub_fold_lds != 0 && nval <= kStage;
  if (staged) {
    const float* src = part + static_cast<size_t>(c0) * 2 * a.h;
    for (int t = threadIdx.x; t < nval; t += kBlock) sp_[t] = src[t];
    __syncthreads();
  }
  constexpr bool sp = MODE == 1;
  if (threadIdx.x < a.h && sp) {       // squareplus: the statistics are plain sums of the chunks' shares
    const int head = threadIdx.x;
    float l = 0.f;
    for (int i = 0; i < nch; ++i) l += staged ? sp_[i * 2 * a.h + a.h + head] : part[static_cast<size_t>(c0 + i) * 2 * a.h + a.h + head];
    st[head] = ord2f(*a.gmax);
    st[a.h + head] = __builtin_amdgcn_rcpf(l + 1e-16f);
  } else if (threadIdx.x < a.h) {
    const int head = threadIdx.x;
    float m = -INFINITY;
    float l = 0.f;
    if (staged) {
      for (int i = 0; i < nch; ++i) m = fmaxf(m, sp_[i * 2 * a.h + head]);
      for (int i = 0; i < nch; ++i) {
        const float* q = sp_ + i * 2 * a.h;
        l += q[a.h + head] * expf(q[head] - m);
      }
    } else {
      for (int i = 0; i < nch; ++i) m = fmaxf(m, part[static_cast<size_t>(c0 + i) * 2 * a.h + head]);
      for (int i = 0; i < nch; ++i) {
        const float* q = part + static_cast<size_t>(c0 + i) * 2 * a.h;
        l += q[a.h + head] * expf(q[head] - m);
      }
    }
    st[head] = m;
    st[a.h + head] = __builtin_amdgcn_rcpf(l + 1e-16f);     // one reciprocal per (row, head), a multiplication per entry
  }
  __syncthreads();
  for (int p = b + threadIdx.x; p < e; p += kBlock) {
    float acc = 0.f;
    if (sp) {
      for (int head = 0; head < a.h; ++head) acc += squareplus_num(a.scores[static_cast<size_t>(p) * a.h + head], st[head]) * st[a.h + head];
    } else {
      for (int head = 0; head < a.h; ++head)
        acc += __builtin_amdgcn_exp2f((a.scores[static_cast<size_t>(p) * a.h + head] - st[head]) * 1.44269504088896341f) * st[a.h + head];
    }
    if constexpr (SCATTER) a.w_mean[a.out_pos[p]] = acc / static_cast<float>(a.h);
    else a.w_mean[p] = acc / static_cast<float>(a.h);
  }
}

__global__ __launch_bounds__(kBlock) void hub_normalise_kernel(const AttArgs a, const float* __restrict__ part,
                                                              const int* __restrict__ long_chunk_row_first) {
  hub_normalise_body<0, false>(a, part, long_chunk_row_first, blockIdx.x);
}

// ---- fused path: softmax over the row, head-mean weights only.
// GL lanes own one row; inside a group lane = (edge slot, head) with the head fastest, so the H lanes
// of an edge read one contiguous A-float row of k (one 16-byte load each when d_k = 4) and a pass covers
// GE = GL / H edges.  Scores of up to P passes stay in registers (one float per pass per lane).
template <int TYPE, int H, int GL, int P, bool VEC4>
__global__ __launch_bounds__(kBlock) void row_attention_kernel(const AttArgs a, int first_row, int n_rows) {
  constexpr int RPW = kWave / GL;  // rows per wavefront
  constexpr int GE = GL / H;       // edges per pass
  const int lane = threadIdx.x & (kWave - 1);
  const int gi = lane % GL;
  const int slot = gi / H, head = gi % H;
  const long long ridx = (static_cast<long long>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6)) * RPW + lane / GL;
  const bool live = ridx < n_rows;
  int4 info = make_int4(0, 0, 0, 0);  // {row, first CSR position, length, -}
  if (live) info = reinterpret_cast<const int4*>(a.bin_rows)[first_row + ridx];
  int row = info.x, e0 = info.y, e1 = info.y + info.z;
  if constexpr (GL == kWave) {
    row = __builtin_amdgcn_readfirstlane(row);
    e0 = __builtin_amdgcn_readfirstlane(e0);
    e1 = __builtin_amdgcn_readfirstlane(e1);
  }
  int npass = P;  // passes that hold any edge (wave-uniform when a wavefront owns one row)
  if constexpr (GL == kWave) npass = (e1 - e0 + GE - 1) / GE;

  float s[P];
  float m = -INFINITY;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    s[p] = -INFINITY;
    if (p < npass) {
      const int e = e0 + p * GE + slot;
      if (e < e1) {
        s[p] = edge_score<TYPE, VEC4>(a, e, row, a.colidx[e], head);
        m = fmaxf(m, s[p]);
      }
    }
  }
#pragma unroll
  for (int off = H; off < GL; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
  float l = 0.f;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    if (p < npass) {
      s[p] = expf(s[p] - m);  // empty slots: exp(-inf) = 0
      l += s[p];
    }
  }
#pragma unroll
  for (int off = H; off < GL; off <<= 1) l += __shfl_xor(l, off, kWave);
  const float den = l + 1e-16f;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    if (p < npass) {
      float v = s[p] / den;
#pragma unroll
      for (int off = 1; off < H; off <<= 1) v += __shfl_xor(v, off, kWave);
      const int e = e0 + p * GE + slot;
      if (head == 0 && e < e1) a.w_mean[e] = v / static_cast<float>(H);
    }
  }
}

// ---- scaled-dot specialisation of the fused row kernel with batched gathers.
// The generic kernel above issues, per pass, colidx -> k -> use, i.e. two dependent memory round trips
// that the scheduler cannot overlap across passes (or rows).  Here all index loads of a batch (RI rows x
// PB passes) are issued first, then all k gathers, then the arithmetic -- 2 round trips per BATCH.
// Out-of-range slots read a clamped (valid, cached) position and are masked to -inf afterwards, so the
// batch is branch-free.  d_k = 4 * DK4.
// The first `n_hub` blocks of the launch do the hub-row work instead (phase 0: chunk scores + partial
// statistics, phase 1: fold partials + normalise), so the long rows ride along with the row kernels rather
// than costing two extra serialised launches.
// MODE 0: softmax over the segment.  MODE 1: squareplus with the global maximum already in *a.gmax (utils.py:179-208: u = (z +
// sqrt(z^2 + 4)) / 2 with z = s - max over ALL scores, normalised by the segment sum).  MODE 2: the sweep that forms that maximum
// (scores recomputed, nothing stored).  The segments are the rows of whatever graph a.bin_rows / a.colidx describe -- the
// transposed graph, with a.q / a.k exchanged and a.out_pos mapping its positions to the CSR positions of the weights, when the
// reference normalises over edge[1] (attention_norm_idx = 1, function_transformer_attention.py:210-213).
template <int H, int DK4, int GL, int RI, int PB, int NB, int MODE, bool SCATTER>
__global__ __launch_bounds__(kBlock) void row_attention_sd_kernel(const AttArgs a, int first_row, int n_rows, int n_hub,
                                                                 int hub_phase, float* __restrict__ part,
                                                                 const int* __restrict__ chunk_first) {
  if (static_cast<int>(blockIdx.x) < n_hub) {
    if (hub_phase == 0) hub_scores_partial_heads<(MODE == 4 ? GNPDE_ATT_EXP_KERNEL : GNPDE_ATT_SCALED_DOT), true, H, MODE>(a, part, blockIdx.x);
    else hub_normalise_body<MODE, SCATTER>(a, part, chunk_first, blockIdx.x);
    return;
  }
  constexpr int RPW = kWave / GL;
  constexpr int GE = GL / H;
  const int lane = threadIdx.x & (kWave - 1);
  const int gi = lane % GL;
  const int slot = gi / H, head = gi % H;
  const long long rbase = ((static_cast<long long>(blockIdx.x - n_hub) * kWavesPerBlock + (threadIdx.x >> 6)) * RPW + lane / GL) * RI;

  int row[RI], e0[RI], e1[RI];
  bool live[RI];
  // all RI records are requested before the first one is used (round 3: with the load, its wait and the readfirstlane inside
  // one loop body the four record fetches of a wave were four dependent round trips); a slot past the end re-reads the
  // class's last record -- valid addresses, nothing written (live[r])
  int4 info[RI];
#pragma unroll
  for (int r = 0; r < RI; ++r) {
    live[r] = rbase + r < n_rows;
    const long long idx = live[r] ? rbase + r : static_cast<long long>(n_rows) - 1;
    info[r] = reinterpret_cast<const int4*>(a.bin_rows)[first_row + (idx < 0 ? 0 : idx)];
  }
#pragma unroll
  for (int r = 0; r < RI; ++r) {
    row[r] = info[r].x; e0[r] = info[r].y; e1[r] = info[r].y + info[r].z;
    if constexpr (GL == kWave) {
      row[r] = __builtin_amdgcn_readfirstlane(row[r]);
      e0[r] = __builtin_amdgcn_readfirstlane(e0[r]);
      e1[r] = __builtin_amdgcn_readfirstlane(e1[r]);
    }
  }
  float s[RI][NB * PB];
  float m[RI];
#pragma unroll
  for (int r = 0; r < RI; ++r) m[r] = -INFINITY;
  int nbatch = NB;
  if constexpr (GL == kWave && RI == 1) nbatch = (e1[0] - e0[0] + PB * GE - 1) / (PB * GE);  // wave-uniform
  if constexpr (MODE == 1) {
    // squareplus, second sweep: the scores were stored by the maximum sweep (MODE 2) -- one coalesced 4 H-byte read per entry
    // instead of the k-row gather
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < RI; ++r)
#pragma unroll
        for (int i = 0; i < PB; ++i) {
          const int e = e0[r] + (nb * PB + i) * GE + slot;
          s[r][nb * PB + i] = (nb < nbatch && e < e1[r]) ? a.scores[static_cast<size_t>(e) * H + head] : -INFINITY;
        }
  } else {
  float4 qv[RI][DK4];
#pragma unroll
  for (int r = 0; r < RI; ++r)
#pragma unroll
    for (int j = 0; j < DK4; ++j)
      qv[r][j] = *reinterpret_cast<const float4*>(a.q + static_cast<size_t>(row[r]) * a.ldqk + head * a.dk + 4 * j);
  float exp_ov2 = 0.f, exp_den = 1.f;
  if constexpr (MODE == 4) {
    const float ov = *a.output_var, ls = *a.lengthscale;
    exp_ov2 = ov * ov;
    exp_den = 2.0f * (ls * ls);
  }

  // column ids of the batch AFTER the current one are requested before the current batch's k rows are used, so that a row of
  // several batches pays one dependent round trip per batch (k rows), not two (ids -> k rows)
  int cn[RI][PB];
  auto load_ids = [&](int nb) {
#pragma unroll
    for (int r = 0; r < RI; ++r)
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int e = e0[r] + (nb * PB + i) * GE + slot;
        cn[r][i] = a.colidx[e < e1[r] ? e : e1[r] - 1];
      }
  };
  load_ids(0);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    if (nb < nbatch) {
      int c[RI][PB];
#pragma unroll
      for (int r = 0; r < RI; ++r)
#pragma unroll
        for (int i = 0; i < PB; ++i) c[r][i] = cn[r][i];
      if (NB > 1 && nb + 1 < nbatch) load_ids(nb + 1);
      float4 kv[RI][PB][DK4];
#pragma unroll
      for (int r = 0; r < RI; ++r)
#pragma unroll
        for (int i = 0; i < PB; ++i)
#pragma unroll
          for (int j = 0; j < DK4; ++j)
            kv[r][i][j] = *reinterpret_cast<const float4*>(a.k + static_cast<size_t>(c[r][i]) * a.ldqk + head * a.dk + 4 * j);
#pragma unroll
      for (int r = 0; r < RI; ++r)
#pragma unroll
        for (int i = 0; i < PB; ++i) {
          float dot = 0.f;
#pragma unroll
          for (int j = 0; j < DK4; ++j) {
            if constexpr (MODE == 4) {   // exp kernel: |q - k|^2, the same chain as edge_score<GNPDE_ATT_EXP_KERNEL>
              float t;
              t = qv[r][j].x - kv[r][i][j].x; dot = fmaf(t, t, dot);
              t = qv[r][j].y - kv[r][i][j].y; dot = fmaf(t, t, dot);
              t = qv[r][j].z - kv[r][i][j].z; dot = fmaf(t, t, dot);
              t = qv[r][j].w - kv[r][i][j].w; dot = fmaf(t, t, dot);
            } else {
              dot = fmaf(qv[r][j].x, kv[r][i][j].x, dot);
              dot = fmaf(qv[r][j].y, kv[r][i][j].y, dot);
              dot = fmaf(qv[r][j].z, kv[r][i][j].z, dot);
              dot = fmaf(qv[r][j].w, kv[r][i][j].w, dot);
            }
          }
          const int e = e0[r] + (nb * PB + i) * GE + slot;
          // (this kernel is bound by VALU issue, not by memory: ~580 wave instructions per 4 rows, a fifth of them the IEEE
          //  division sequences and the full-range expf -- round 3: exact power-of-two scale, one reciprocal per row, v_exp)
          float sv = dot * a.scale_mul;
          if constexpr (MODE == 3) sv = sv > 0.f ? sv : sv * a.leaky_slope;      // GAT: the vectors are gat_terms_kernel's table
          if constexpr (MODE == 4) sv = exp_ov2 * expf(-(dot / exp_den));        // ov^2 exp(-|q - k|^2 / 2 l^2) (reference :193-196)
          if (a.edge_w != nullptr) sv = sv * a.edge_w[e < e1[r] ? e : e1[r] - 1];
          if constexpr (MODE == 2) {
            if (live[r] && e < e1[r]) a.scores[static_cast<size_t>(e) * H + head] = sv;     // kept for the second sweep
          }
          sv = e < e1[r] ? sv : -INFINITY;
          s[r][nb * PB + i] = sv;
          m[r] = fmaxf(m[r], sv);
        }
    } else {
#pragma unroll
      for (int r = 0; r < RI; ++r)
#pragma unroll
        for (int i = 0; i < PB; ++i) s[r][nb * PB + i] = -INFINITY;
    }
  }
  }   // MODE != 1
  // The reductions run LEVEL by level over all RI rows of the wave, not row by row: every xor step is a dependent LDS round
  // trip (ds_bpermute + wait), and written row by row the 10 steps of a row x RI rows formed one chain of 40 (round 3, from the
  // ISA); level-synchronous there are RI independent exchanges in flight per step.  Same operations per row, same order.
  if constexpr (MODE == 2) {     // only the maximum over every score (slots past the end hold -inf, rows past the end repeat a listed row)
    float mw = m[0];
#pragma unroll
    for (int r = 1; r < RI; ++r) mw = fmaxf(mw, m[r]);
    mw = wave_max(mw);
    // one atomic per wave at most, and only while the wave's maximum beats the value it can see (a stale read only costs an atomic:
    // tens of thousands of waves hammering ONE address serialise -- 0.7 ms per sweep at the ogbn-arxiv shape when every wave did)
    if (a.gmax_part != nullptr) {      // small grid: one plain store per wave (0 = below every encoded float)
      if (lane == 0)
        a.gmax_part[a.gmax_base + (static_cast<int>(blockIdx.x) - n_hub) * kWavesPerBlock + static_cast<int>(threadIdx.x >> 6)] =
            mw > -INFINITY ? f2ord(mw) : 0u;
      return;
    }
    if (lane == 0 && mw > -INFINITY) {
      const unsigned mine = f2ord(mw);
      unsigned* tgt = a.gmax_slots != nullptr ? a.gmax_slots + 32 * (blockIdx.x % kGmaxSlots) : a.gmax;
      if (mine > __hip_atomic_load(tgt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(tgt, mine);
    }
    return;
  }
  float l[RI];
  if constexpr (MODE == 1) {
    float gm;
    if (a.gmax_part != nullptr) {      // (block-uniform) fold the sweep's per-wave maxima: gmax_count / 256 coalesced loads per thread
      __shared__ unsigned s_gmax[kWavesPerBlock];
      unsigned v = 0u;
      for (int i = threadIdx.x; i < a.gmax_count; i += kBlock) {
        const unsigned w = a.gmax_part[i];
        v = w > v ? w : v;
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        const unsigned w = static_cast<unsigned>(__shfl_xor(static_cast<int>(v), o, kWave));
        v = w > v ? w : v;
      }
      if (lane == 0) s_gmax[threadIdx.x >> 6] = v;
      __syncthreads();
      v = s_gmax[0];
#pragma unroll
      for (int w = 1; w < kWavesPerBlock; ++w) v = s_gmax[w] > v ? s_gmax[w] : v;
      if (blockIdx.x == 0 && threadIdx.x == 0 && first_row == 0) *a.gmax = v;     // (for the kernels behind this launch: the backward passes)
      gm = ord2f(v);
    } else {
      gm = ord2f(*a.gmax);
    }
#pragma unroll
    for (int r = 0; r < RI; ++r) {
      l[r] = 0.f;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        if (nb < nbatch) {
#pragma unroll
          for (int i = 0; i < PB; ++i) {
            const float sv = s[r][nb * PB + i];
            s[r][nb * PB + i] = sv > -INFINITY ? squareplus_num_fast(sv, gm) : 0.f;
            l[r] += s[r][nb * PB + i];
          }
        }
    }
  } else {
#pragma unroll
    for (int off = H; off < GL; off <<= 1)
#pragma unroll
      for (int r = 0; r < RI; ++r) m[r] = fmaxf(m[r], __shfl_xor(m[r], off, kWave));
#pragma unroll
    for (int r = 0; r < RI; ++r) {
      l[r] = 0.f;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        if (nb < nbatch) {
#pragma unroll
          for (int i = 0; i < PB; ++i) {
            s[r][nb * PB + i] = __builtin_amdgcn_exp2f((s[r][nb * PB + i] - m[r]) * 1.44269504088896341f);   // v_exp_f32: arguments are <= 0, -inf -> 0
            l[r] += s[r][nb * PB + i];
          }
        }
    }
  }
#pragma unroll
  for (int off = H; off < GL; off <<= 1)
#pragma unroll
    for (int r = 0; r < RI; ++r) l[r] += __shfl_xor(l[r], off, kWave);
  float rden[RI];
#pragma unroll
  for (int r = 0; r < RI; ++r) rden[r] = __builtin_amdgcn_rcpf(l[r] + 1e-16f);   // v_rcp_f32 (1 ulp), one per (row, head); a multiplication per entry
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
    if (nb < nbatch) {
      float v[RI][PB];
#pragma unroll
      for (int r = 0; r < RI; ++r)
#pragma unroll
        for (int i = 0; i < PB; ++i) v[r][i] = s[r][nb * PB + i] * rden[r];
#pragma unroll
      for (int off = 1; off < H; off <<= 1)
#pragma unroll
        for (int r = 0; r < RI; ++r)
#pragma unroll
          for (int i = 0; i < PB; ++i) v[r][i] += __shfl_xor(v[r][i], off, kWave);
#pragma unroll
      for (int r = 0; r < RI; ++r)
#pragma unroll
        for (int i = 0; i < PB; ++i) {
          const int e = e0[r] + (nb * PB + i) * GE + slot;
          if (head == 0 && live[r] && e < e1[r]) {
            if constexpr (SCATTER) a.w_mean[a.out_pos[e]] = v[r][i] / static_cast<float>(H);     // (a template switch: the plain row softmax keeps its store as it was)
            else a.w_mean[e] = v[r][i] / static_cast<float>(H);
          }
        }
    }
}

__global__ __launch_bounds__(kBlock) void edge_to_csr_mean_kernel(const int* __restrict__ perm, const float* __restrict__ src,
                                                                 int h, int e, float* __restrict__ w) {
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long p = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; p < e; p += stride) {
    const float* s = src + static_cast<size_t>(perm[p]) * h;
    if (h == 1) {
      w[p] = s[0];
    } else {
      float acc = 0.f;
      for (int j = 0; j < h; ++j) acc += s[j];
      w[p] = acc / static_cast<float>(h);
    }
  }
}

inline unsigned stream_grid(long long work_items) {
  long long blocks = (work_items + kBlock - 1) / kBlock;
  const long long cap = 256LL * 8;  // 256 CUs x 8 blocks, grid-stride beyond
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return static_cast<unsigned>(blocks);
}

inline size_t long_slots_of(const gnpde_graph_t* g) {
  const size_t rc = static_cast<size_t>(g->n_long_rows) * ((g->max_row_len + GNPDE_LONG_ROW - 1) / GNPDE_LONG_ROW);
  const size_t cc = static_cast<size_t>(g->n_long_cols) * ((g->max_col_len + GNPDE_LONG_ROW - 1) / GNPDE_LONG_ROW);
  const size_t lc = static_cast<size_t>(g->n_long_chunks);
  const size_t m = rc > cc ? rc : cc;
  return m > lc ? m : lc;
}

struct AttLayout {
  size_t scores, seg_m, seg_den, gmax, gat, gat_table, part, total;
};
// `long_slots` = (number of long segments) x (max chunks of one segment), an upper bound is fine
AttLayout att_layout(int n, int e, int h, bool gat, size_t long_slots, int key_rows = 0) {
  AttLayout L{};
  size_t off = 0;
  L.scores = off; off += align_up(static_cast<size_t>(e) * h * 4, 256);
  L.seg_m = off;  off += align_up(static_cast<size_t>(n) * h * 4, 256);
  L.seg_den = off; off += align_up(static_cast<size_t>(n) * h * 4, 256);
  L.gmax = off; off += 256 + (kGmaxSlots * 128 > kGmaxPartCap * 4 ? kGmaxSlots * 128 : kGmaxPartCap * 4);   // the maximum, then its partial slots
                                                                                                             // (AttArgs::gmax_slots / gmax_part)
  L.gat = off; if (gat) off += align_up(static_cast<size_t>(key_rows > n ? key_rows : n) * 2 * h * 4, 256);   // (halo rows of a partitioned graph)
  L.gat_table = off; if (gat) off += align_up(static_cast<size_t>(key_rows > n ? key_rows : n) * 8 * h * 4, 256);
  L.part = off; off += align_up(long_slots * 2 * h * 4, 256);
  L.total = off;
  return L;
}

template <int TYPE>
void launch_scores(const AttArgs& a, bool vec4, unsigned grid, hipStream_t s) {
  if (vec4) hipLaunchKernelGGL((scores_kernel<TYPE, true>), dim3(grid), dim3(kBlock), 0, s, a);
  else hipLaunchKernelGGL((scores_kernel<TYPE, false>), dim3(grid), dim3(kBlock), 0, s, a);
}

void launch_scores_any(const AttArgs& a, bool vec4, unsigned grid, hipStream_t s) {
  switch (a.type) {
    case GNPDE_ATT_SCALED_DOT: launch_scores<GNPDE_ATT_SCALED_DOT>(a, vec4, grid, s); break;
    case GNPDE_ATT_COSINE: launch_scores<GNPDE_ATT_COSINE>(a, vec4, grid, s); break;
    case GNPDE_ATT_PEARSON: launch_scores<GNPDE_ATT_PEARSON>(a, vec4, grid, s); break;
    case GNPDE_ATT_EXP_KERNEL: launch_scores<GNPDE_ATT_EXP_KERNEL>(a, vec4, grid, s); break;
    default: launch_scores<GNPDE_ATT_GAT>(a, false, grid, s); break;
  }
}

// rows per workgroup of the <= 16-entry class of row_attention_sd_kernel (launch_rows_sd_mode's first launch)
// (4 heads, d_k 4 or 16: a QUARTER of a wave per row, see launch_rows_sd_mode)
constexpr bool sd_quarter_rows(int h, int dk4) { return h == 4 && (dk4 == 1 || dk4 == 4); }
constexpr int sd_rows_per_block16(int h, int dk4) {
  const int gl16 = sd_quarter_rows(h, dk4) ? 16 : ((16 * h < kWave) ? 16 * h : kWave);
  const int p16 = (16 * h + gl16 - 1) / gl16;
  const int ri16 = sd_quarter_rows(h, dk4) ? 2 : ((dk4 == 1 && p16 == 1) ? 4 : (p16 == 1 ? 2 : 1));
  return (kWave / gl16) * ri16 * kWavesPerBlock;
}
// waves of the two launches of a sweep without hub blocks: [0] the <= 16-entry class, [1] the others (AttArgs::gmax_part)
inline void sd_sweep_waves(int h, int dk4, int n16, int n64, long long (&w)[2]) {
  const long long rpb = sd_rows_per_block16(h, dk4);
  w[0] = n16 > 0 ? (n16 + rpb - 1) / rpb * kWavesPerBlock : 0;
  w[1] = n64 > 0 ? static_cast<long long>(n64 + kWavesPerBlock - 1) / kWavesPerBlock * kWavesPerBlock : 0;
}

template <int H, int DK4, int MODE, bool SCATTER>
void launch_rows_sd_mode(const AttArgs& a_in, int n16, int n64, hipStream_t s, int n_hub, float* part, const int* chunk_first) {
  AttArgs a = a_in;
  // Rows of <= 16 entries.  4 heads (round 6): a QUARTER of a wave per row -- 16 lanes = 4 entries x 4 heads per pass, four passes, two
  // rows interleaved per group: 8 rows per wave instead of 4 on whole waves, where a median row of 8 entries left half of the lanes without
  // an entry.  Measured at the ogbn-arxiv shape: row attention 41.2 -> 37.3 us per evaluation (1074 -> 1089 steps/s at T = 100), R-MAT
  // (d_k = 16) 4.48 -> 3.98 ms; the other packings tried on the row softmax (half a wave x 2 passes with 8 / 4 rows per wave: 38.2; a quarter
  // with 16 rows per wave: 40.4, with 4: 37.3; half with 16 rows: 43.0; an eighth x 8 passes with 16 / 8 rows: 40.4 / 37.2) are in DESIGN.md,
  // the whole-wave form stays behind gnpde_tune(17, 9) for the row softmax (A/B).  Other head counts: 16 H lanes (at most a wave) per row.
  constexpr bool QUARTER = sd_quarter_rows(H, DK4);
  constexpr int GL16 = QUARTER ? 16 : ((16 * H < kWave) ? 16 * H : kWave);   // lanes per row for rows with <= 16 entries
  constexpr int P16 = (16 * H + GL16 - 1) / GL16;           // passes to cover 16 entries
  constexpr int RPW16 = kWave / GL16;
  constexpr int RI16 = QUARTER ? 2 : ((DK4 == 1 && P16 == 1) ? 4 : (P16 == 1 ? 2 : 1));  // rows interleaved per group
  constexpr int P64 = GNPDE_LONG_ROW / (kWave / H);         // passes to cover GNPDE_LONG_ROW entries
  constexpr int PB64 = (DK4 == 1) ? 4 : 2;
  // launch 1: hub phase 0 + rows with <= 16 entries; launch 2: hub phase 1 + rows with 17..512 entries (the maximum sweep of
  // squareplus, MODE 2, has no hub phase 1)
  bool first_done = false;
  if constexpr (QUARTER && MODE == 0 && !SCATTER) {
    if (g_tune[GNPDE_TUNE_ATT_ROWS16] == 9 && (n16 > 0 || n_hub > 0)) {      // A/B: a whole wave per row, four rows interleaved (until round 6)
      constexpr int RIW = DK4 == 1 ? 4 : 2;
      const long long rpb = static_cast<long long>(RIW) * kWavesPerBlock;
      const unsigned grid = static_cast<unsigned>((n16 + rpb - 1) / rpb) + n_hub;
      hipLaunchKernelGGL((row_attention_sd_kernel<H, DK4, kWave, RIW, 1, 1, MODE, SCATTER>), dim3(grid), dim3(kBlock), 0, s, a, 0, n16, n_hub, 0, part,
                         chunk_first);
      first_done = true;
    }
  }
  if (!first_done && (n16 > 0 || n_hub > 0)) {
    const long long rows_per_block = static_cast<long long>(RPW16) * RI16 * kWavesPerBlock;
    static_assert(RPW16 * RI16 * kWavesPerBlock == sd_rows_per_block16(H, DK4), "sd_sweep_waves counts this launch's waves");
    const unsigned grid = static_cast<unsigned>((n16 + rows_per_block - 1) / rows_per_block) + n_hub;
    hipLaunchKernelGGL((row_attention_sd_kernel<H, DK4, GL16, RI16, P16, 1, MODE, SCATTER>), dim3(grid), dim3(kBlock), 0, s, a, 0, n16, n_hub,
                       0, part, chunk_first);
    if (MODE == 2 && a.gmax_part != nullptr) a.gmax_base += static_cast<int>(grid - n_hub) * kWavesPerBlock;   // the second launch's waves follow
  }
  const int n_hub2 = MODE == 2 ? 0 : n_hub;
  if (n64 > 0 || n_hub2 > 0) {
    const unsigned grid = static_cast<unsigned>((n64 + kWavesPerBlock - 1) / kWavesPerBlock) + n_hub2;
    hipLaunchKernelGGL((row_attention_sd_kernel<H, DK4, kWave, 1, PB64, P64 / PB64, MODE, SCATTER>), dim3(grid), dim3(kBlock), 0, s, a, n16,
                       n64, n_hub2, 1, part, chunk_first);
  }
}

template <int H, int DK4>
void launch_rows_sd(const AttArgs& a, int n16, int n64, hipStream_t s, int n_hub = 0, float* part = nullptr,
                    const int* chunk_first = nullptr) {
  const bool sc = a.out_pos != nullptr;
  if (a.sp_mode == 4) { launch_rows_sd_mode<H, DK4, 4, false>(a, n16, n64, s, n_hub, part, chunk_first); return; }   // exp kernel scores
  if (a.sp_mode == 3) {              // GAT scores through the scaled-dot kernels (4-wide vectors only)
    if constexpr (DK4 == 1) launch_rows_sd_mode<H, DK4, 3, false>(a, n16, n64, s, n_hub, part, chunk_first);
    return;
  }
  if (a.sp_mode == 2) launch_rows_sd_mode<H, DK4, 2, false>(a, n16, n64, s, n_hub, part, chunk_first);      // (writes no weights)
  else if (a.sp_mode == 1) {
    if (sc) launch_rows_sd_mode<H, DK4, 1, true>(a, n16, n64, s, n_hub, part, chunk_first);
    else launch_rows_sd_mode<H, DK4, 1, false>(a, n16, n64, s, n_hub, part, chunk_first);
  } else {
    if (sc) launch_rows_sd_mode<H, DK4, 0, true>(a, n16, n64, s, n_hub, part, chunk_first);
    else launch_rows_sd_mode<H, DK4, 0, false>(a, n16, n64, s, n_hub, part, chunk_first);
  }
}

// scaled-dot rows + hub chunks in two launches; false if this (heads, d_k) has no specialised kernel
bool launch_sd_with_hubs(const AttArgs& c, int n16, int n64, int n_hub, float* part, const int* chunk_first, hipStream_t s) {
  if (g_tune[GNPDE_TUNE_ATT_GENERIC_ROWS] != 0) return false;
#define GNPDE_SD(HH)                                                                              \
  case HH:                                                                                        \
    if (c.dk == 4) { launch_rows_sd<HH, 1>(c, n16, n64, s, n_hub, part, chunk_first); return true; }  \
    if (c.dk == 8) { launch_rows_sd<HH, 2>(c, n16, n64, s, n_hub, part, chunk_first); return true; }  \
    if (c.dk == 16) { launch_rows_sd<HH, 4>(c, n16, n64, s, n_hub, part, chunk_first); return true; } \
    return false;
  switch (c.h) {
    GNPDE_SD(1) GNPDE_SD(2) GNPDE_SD(4) GNPDE_SD(8)
    default: return false;
  }
#undef GNPDE_SD
}

template <int TYPE, int H, bool VEC4>
void launch_rows_th(const AttArgs& a, int n16, int n64, hipStream_t s) {
  if constexpr (TYPE == GNPDE_ATT_SCALED_DOT && VEC4) {
    if (g_tune[GNPDE_TUNE_ATT_GENERIC_ROWS] == 0) {
      if (a.dk == 4) return launch_rows_sd<H, 1>(a, n16, n64, s);
      if (a.dk == 8) return launch_rows_sd<H, 2>(a, n16, n64, s);
      if (a.dk == 16) return launch_rows_sd<H, 4>(a, n16, n64, s);
    }
  }
  // rows with <= 16 entries: 16*H lanes cover them in one pass when H <= 4, else a whole wave in H/4 passes
  constexpr int GL16 = (16 * H < kWave) ? 16 * H : kWave;
  constexpr int P16 = (16 * H + GL16 - 1) / GL16;
  constexpr int RPW16 = kWave / GL16;
  constexpr int P64 = GNPDE_LONG_ROW / (kWave / H);
  if (n16 > 0) {
    const unsigned grid = static_cast<unsigned>((n16 + RPW16 * kWavesPerBlock - 1) / (RPW16 * kWavesPerBlock));
    hipLaunchKernelGGL((row_attention_kernel<TYPE, H, GL16, P16, VEC4>), dim3(grid), dim3(kBlock), 0, s, a, 0, n16);
  }
  if (n64 > 0) {
    const unsigned grid = static_cast<unsigned>((n64 + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL((row_attention_kernel<TYPE, H, kWave, P64, VEC4>), dim3(grid), dim3(kBlock), 0, s, a, n16, n64);
  }
}

template <int TYPE, bool VEC4>
bool launch_rows_t(const AttArgs& a, int n16, int n64, hipStream_t s) {
  switch (a.h) {
    case 1: launch_rows_th<TYPE, 1, VEC4>(a, n16, n64, s); return true;
    case 2: launch_rows_th<TYPE, 2, VEC4>(a, n16, n64, s); return true;
    case 4: launch_rows_th<TYPE, 4, VEC4>(a, n16, n64, s); return true;
    case 8: launch_rows_th<TYPE, 8, VEC4>(a, n16, n64, s); return true;
    default: return false;
  }
}

template <int H>
__global__ __launch_bounds__(kBlock) void hub_scores_partial_sd_kernel(const AttArgs a, float* __restrict__ part) {
  hub_scores_partial_heads<GNPDE_ATT_SCALED_DOT, true, H>(a, part, blockIdx.x);
}

template <int TYPE>
void launch_hub_a(const AttArgs& c, bool vec4, float* part, int n_chunks, hipStream_t s) {
  if constexpr (TYPE == GNPDE_ATT_SCALED_DOT) {
    if (vec4) {
      switch (c.h) {
        case 1: hipLaunchKernelGGL(hub_scores_partial_sd_kernel<1>, dim3(n_chunks), dim3(kBlock), 0, s, c, part); return;
        case 2: hipLaunchKernelGGL(hub_scores_partial_sd_kernel<2>, dim3(n_chunks), dim3(kBlock), 0, s, c, part); return;
        case 4: hipLaunchKernelGGL(hub_scores_partial_sd_kernel<4>, dim3(n_chunks), dim3(kBlock), 0, s, c, part); return;
        case 8: hipLaunchKernelGGL(hub_scores_partial_sd_kernel<8>, dim3(n_chunks), dim3(kBlock), 0, s, c, part); return;
        default: break;
      }
    }
  }
  if (vec4) hipLaunchKernelGGL((hub_scores_partial_kernel<TYPE, true>), dim3(n_chunks), dim3(kBlock), 0, s, c, part);
  else hipLaunchKernelGGL((hub_scores_partial_kernel<TYPE, false>), dim3(n_chunks), dim3(kBlock), 0, s, c, part);
}

void launch_hub_a_any(const AttArgs& c, bool vec4, float* part, int n_chunks, hipStream_t s) {
  switch (c.type) {
    case GNPDE_ATT_SCALED_DOT: launch_hub_a<GNPDE_ATT_SCALED_DOT>(c, vec4, part, n_chunks, s); break;
    case GNPDE_ATT_COSINE: launch_hub_a<GNPDE_ATT_COSINE>(c, vec4, part, n_chunks, s); break;
    case GNPDE_ATT_PEARSON: launch_hub_a<GNPDE_ATT_PEARSON>(c, vec4, part, n_chunks, s); break;
    case GNPDE_ATT_EXP_KERNEL: launch_hub_a<GNPDE_ATT_EXP_KERNEL>(c, vec4, part, n_chunks, s); break;
    default: launch_hub_a<GNPDE_ATT_GAT>(c, false, part, n_chunks, s); break;
  }
}

// fused row kernels exist for these (type, heads, alignment) combinations
bool fused_supported(const AttArgs& a, bool vec4) {
  if (!(a.h == 1 || a.h == 2 || a.h == 4 || a.h == 8)) return false;
  if (a.type == GNPDE_ATT_GAT) return true;
  return vec4 && (a.type == GNPDE_ATT_SCALED_DOT || a.type == GNPDE_ATT_COSINE || a.type == GNPDE_ATT_PEARSON ||
                  a.type == GNPDE_ATT_EXP_KERNEL);
}

void launch_rows_any(const AttArgs& a, int n16, int n64, hipStream_t s) {
  switch (a.type) {
    case GNPDE_ATT_SCALED_DOT: launch_rows_t<GNPDE_ATT_SCALED_DOT, true>(a, n16, n64, s); break;
    case GNPDE_ATT_COSINE: launch_rows_t<GNPDE_ATT_COSINE, true>(a, n16, n64, s); break;
    case GNPDE_ATT_PEARSON: launch_rows_t<GNPDE_ATT_PEARSON, true>(a, n16, n64, s); break;
    case GNPDE_ATT_EXP_KERNEL: launch_rows_t<GNPDE_ATT_EXP_KERNEL, true>(a, n16, n64, s); break;
    default: launch_rows_t<GNPDE_ATT_GAT, false>(a, n16, n64, s); break;
  }
}

}  // namespace

size_t attention_workspace_bytes(const gnpde_graph_t* g, int h, bool gat);

// stats_only: run the general passes 1-2 only (scores + segment statistics stay in the workspace) and hand back the
// kernel arguments -- the backward pass continues from there
static int edge_attention_impl(const gnpde_graph_t* g, const gnpde_attention_t* at, float* w_mean_csr, float* att_edge,
                               float* prods_edge, void* ws, size_t ws_bytes, hipStream_t stream, const Fork* fork,
                               bool stats_only, AttArgs* args_out, bool hubs_only = false, int pass_only = 0) {
  GNPDE_CHECK_ARG(g && at, GNPDE_EINVAL, "edge_attention: null descriptor");
  GNPDE_CHECK_ARG(stats_only || w_mean_csr || att_edge || prods_edge, GNPDE_EINVAL, "edge_attention: no output requested");
  GNPDE_CHECK_ARG(at->heads >= 1 && at->att_dim >= at->heads && at->att_dim % at->heads == 0, GNPDE_EINVAL,
                  "edge_attention: heads (%d) must be a factor of the attention dimension (%d)", at->heads, at->att_dim);
  GNPDE_CHECK_ARG(at->type >= GNPDE_ATT_SCALED_DOT && at->type <= GNPDE_ATT_GAT, GNPDE_EINVAL, "edge_attention: bad type %d", at->type);
  GNPDE_CHECK_ARG(at->norm_idx == 0 || at->norm_idx == 1, GNPDE_EINVAL, "edge_attention: attention_norm_idx must be 0 or 1");
  GNPDE_CHECK_ARG(at->q && at->k && at->ldqk >= at->att_dim, GNPDE_EINVAL, "edge_attention: bad q/k");
  GNPDE_CHECK_ARG(at->type != GNPDE_ATT_GAT || at->gat_a, GNPDE_EINVAL, "edge_attention: GAT needs the vector a");
  GNPDE_CHECK_ARG(at->type != GNPDE_ATT_EXP_KERNEL || (at->output_var && at->lengthscale), GNPDE_EINVAL,
                  "edge_attention: exp_kernel needs output_var and lengthscale");
  GNPDE_CHECK_ARG(at->norm_idx == 0 || (g->cscptr && g->cscpos), GNPDE_EINVAL, "edge_attention: norm_idx=1 needs the CSC view");
  GNPDE_CHECK_ARG(g->rowidx && g->perm, GNPDE_EINVAL, "edge_attention: graph lacks rowidx/perm");
  if (g->e == 0 || g->n == 0) return 0;
  const bool gat = at->type == GNPDE_ATT_GAT;
  const int n_long = at->norm_idx == 0 ? g->n_long_rows : g->n_long_cols;
  const int* long_list = at->norm_idx == 0 ? g->long_rows : g->long_cols;
  const int max_len = at->norm_idx == 0 ? g->max_row_len : g->max_col_len;
  const int max_chunks = (max_len + GNPDE_LONG_ROW - 1) / GNPDE_LONG_ROW;
  const AttLayout L = att_layout(g->n, g->e, at->heads, gat, long_slots_of(g), at->n_key_rows);
  GNPDE_CHECK_ARG(ws && ws_bytes >= L.total, GNPDE_EWS, "edge_attention: workspace %zu < %zu bytes", ws_bytes, L.total);
  GNPDE_CHECK_ARG(reinterpret_cast<uintptr_t>(ws) % 16 == 0, GNPDE_EINVAL, "edge_attention: workspace must be 16-byte aligned");
  char* base = static_cast<char*>(ws);

  AttArgs a{};
  a.n = g->n; a.e = g->e; a.h = at->heads; a.dk = at->att_dim / at->heads;
  a.type = at->type; a.norm_idx = at->norm_idx; a.square_plus = at->square_plus ? 1 : 0;
  a.inv_sqrt_dk_den = static_cast<float>(std::sqrt(static_cast<double>(a.dk)));
  a.scale_mul = static_cast<float>(1.0 / std::sqrt(static_cast<double>(a.dk)));
  a.leaky_slope = at->leaky_slope;
  a.rowidx = g->rowidx; a.colidx = g->colidx; a.perm = g->perm;
  a.rowptr = g->rowptr; a.bin_rows = g->bin_rows;
  a.segptr = at->norm_idx == 0 ? g->rowptr : g->cscptr;
  a.segpos = at->norm_idx == 0 ? nullptr : g->cscpos;
  a.q = at->q; a.k = at->k; a.ldqk = at->ldqk;
  a.output_var = at->output_var; a.lengthscale = at->lengthscale; a.edge_w = at->edge_w_csr;
  a.scores = reinterpret_cast<float*>(base + L.scores);
  a.seg_m = reinterpret_cast<float*>(base + L.seg_m);
  a.seg_den = reinterpret_cast<float*>(base + L.seg_den);
  a.gmax = reinterpret_cast<unsigned*>(base + L.gmax);
  a.gat_terms = reinterpret_cast<float*>(base + L.gat);
  a.w_mean = w_mean_csr; a.att_edge = att_edge; a.prods_edge = prods_edge;
  a.hub_fold_lds = g_tune[GNPDE_TUNE_HUB_FOLD] == 2 ? 0 : 1;   // default since round 3 (bit-identical; -3.4 % on the R-MAT launch)
  float* part = reinterpret_cast<float*>(base + L.part);

  // GAT's row softmax on the scaled-dot row kernels (gat_terms_kernel's table): the fused row path without edge weights
  const bool gat_sd = gat && !stats_only && pass_only == 0 && !hubs_only && a.norm_idx == 0 && !a.square_plus && att_edge == nullptr &&
                      prods_edge == nullptr && w_mean_csr != nullptr && g->bin_rows != nullptr && a.edge_w == nullptr &&
                      (a.h == 1 || a.h == 2 || a.h == 4 || a.h == 8) && (fork == nullptr || fork->aux == nullptr) &&
                      (g->n_long_rows == 0 || g->long_chunk_first != nullptr) && g_tune[GNPDE_TUNE_ATT_GENERIC_ROWS] == 0;
  if (gat) {
    const int term_rows = at->n_key_rows > g->n ? at->n_key_rows : g->n;      // every row a column may address
    const long long items = static_cast<long long>(term_rows) * a.h;
    hipLaunchKernelGGL(gat_terms_kernel, dim3(static_cast<unsigned>((items + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                       at->q, at->ldqk, at->gat_a, term_rows, a.h, a.dk, reinterpret_cast<float*>(base + L.gat),
                       gat_sd ? reinterpret_cast<float*>(base + L.gat_table) : nullptr);
    GNPDE_LAUNCH_CHECK();
  }
  const bool vec4 = (a.dk % 4 == 0) && (a.ldqk % 4 == 0) && (reinterpret_cast<uintptr_t>(a.q) % 16 == 0) &&
                    (reinterpret_cast<uintptr_t>(a.k) % 16 == 0);

  const bool fused = !stats_only && pass_only == 0 && a.norm_idx == 0 && !a.square_plus && att_edge == nullptr &&
                     prods_edge == nullptr && g->bin_rows != nullptr && fused_supported(a, vec4);
  GNPDE_CHECK_ARG(g->row_begin == 0 || fused, GNPDE_EINVAL, "edge_attention: a row sub-range needs the fused row path");
  if (fused) {
    AttArgs c = a;
    c.chunk_begin = g->long_chunk_begin;
    c.chunk_end = g->long_chunk_end;
    c.long_segs = g->long_rows;
    if (hubs_only) {   // head-mean weights of the entries of the long rows only (the short rows are attended elsewhere)
      if (g->n_long_rows > 0) {
        launch_hub_a_any(c, vec4, part, g->n_long_chunks, stream);
        GNPDE_LAUNCH_CHECK();
        hipLaunchKernelGGL(hub_normalise_kernel, dim3(g->n_long_chunks), dim3(kBlock), 0, stream, c, part, g->long_chunk_first);
        GNPDE_LAUNCH_CHECK();
      }
      return 0;
    }
    if (gat_sd) {
      AttArgs c2 = c;
      c2.type = GNPDE_ATT_SCALED_DOT;
      c2.q = reinterpret_cast<const float*>(base + L.gat_table);
      c2.k = c2.q + 4 * a.h;
      c2.ldqk = 8 * a.h;
      c2.dk = 4;
      c2.scale_mul = 0.5f;
      c2.inv_sqrt_dk_den = 2.0f;
      c2.sp_mode = 3;
      if (launch_sd_with_hubs(c2, g->n_bin16, g->n_bin64, g->n_long_rows > 0 ? g->n_long_chunks : 0, part, g->long_chunk_first, stream)) {
        GNPDE_LAUNCH_CHECK();
        return 0;
      }
    }
    if (a.type == GNPDE_ATT_SCALED_DOT && vec4 && (fork == nullptr || fork->aux == nullptr) &&
        (g->n_long_rows == 0 || g->long_chunk_first != nullptr) &&
        launch_sd_with_hubs(c, g->n_bin16, g->n_bin64, g->n_long_rows > 0 ? g->n_long_chunks : 0, part, g->long_chunk_first,
                            stream)) {
      GNPDE_LAUNCH_CHECK();
      return 0;
    }
    if (a.type == GNPDE_ATT_EXP_KERNEL && vec4 && (fork == nullptr || fork->aux == nullptr) &&
        (g->n_long_rows == 0 || g->long_chunk_first != nullptr)) {
      // the exp kernel through the same row kernels (MODE 4: squared distance instead of the dot product, per-entry formula as
      // edge_score<GNPDE_ATT_EXP_KERNEL>): 815 -> ~1000 steps/s at the ogbn-arxiv shape
      AttArgs c4 = c;
      c4.sp_mode = 4;
      if (launch_sd_with_hubs(c4, g->n_bin16, g->n_bin64, g->n_long_rows > 0 ? g->n_long_chunks : 0, part, g->long_chunk_first, stream)) {
        GNPDE_LAUNCH_CHECK();
        return 0;
      }
    }
    hipStream_t br = stream;
    if (g->n_long_rows > 0) {  // hubs: chunk passes, as a parallel branch when a fork stream is given
      { const int frc = fork_begin(fork, stream, &br); if (frc) return frc; }
      launch_hub_a_any(c, vec4, part, g->n_long_chunks, br);
      GNPDE_LAUNCH_CHECK();
      hipLaunchKernelGGL(hub_normalise_kernel, dim3(g->n_long_chunks), dim3(kBlock), 0, br, c, part, g->long_chunk_first);
      GNPDE_LAUNCH_CHECK();
    }
    launch_rows_any(a, g->n_bin16, g->n_bin64, stream);
    GNPDE_LAUNCH_CHECK();
    if (g->n_long_rows > 0) { const int frc = fork_end(fork, stream, br); if (frc) return frc; }
    return 0;
  }

  // scaled-dot scores with the OTHER normalisers in fused segment passes (the three passes below round-trip an [E,h] score array
  // and walk the segments twice: 2 - 2.5 x the time of the row softmax at the ogbn-arxiv shape).  squareplus: one sweep for the
  // global maximum, then the fused kernel with it; normalisation over the columns: the same kernels over the rows of the
  // transposed graph with q and k exchanged, the weights scattered to their CSR positions.
  {
    const gnpde_graph_t* sg = a.norm_idx == 0 ? g : at->graph_t;
    const bool sd_fast = !stats_only && pass_only == 0 && !hubs_only && a.type == GNPDE_ATT_SCALED_DOT && vec4 && att_edge == nullptr &&
                         prods_edge == nullptr && w_mean_csr != nullptr && (fork == nullptr || fork->aux == nullptr) &&
                         (a.h == 1 || a.h == 2 || a.h == 4 || a.h == 8) && (a.dk == 4 || a.dk == 8 || a.dk == 16) &&
                         g_tune[GNPDE_TUNE_ATT_GENERIC_ROWS] == 0 && g->row_begin == 0 && sg != nullptr && sg->bin_rows != nullptr &&
                         sg->rowidx != nullptr && sg->n == g->n && sg->e == g->e && sg->row_begin == 0 &&
                         (sg->n_long_rows == 0 || (sg->long_chunk_first != nullptr && sg->long_chunk_begin != nullptr)) &&
                         (a.norm_idx == 0 || (at->t_from_csr != nullptr && a.edge_w == nullptr));
    if (sd_fast) {
      AttArgs c = a;
      c.rowidx = sg->rowidx; c.colidx = sg->colidx; c.rowptr = sg->rowptr; c.bin_rows = sg->bin_rows;
      c.chunk_begin = sg->long_chunk_begin; c.chunk_end = sg->long_chunk_end; c.long_segs = sg->long_rows;
      if (a.norm_idx == 1) {
        c.q = a.k; c.k = a.q;            // the segment's own vector is k_j, the gathered one q_i
        c.out_pos = at->t_from_csr;
      }
      const int n_hub = sg->n_long_rows > 0 ? sg->n_long_chunks : 0;
      if (a.square_plus) {
        c.sp_mode = 2;
        long long sw[2];
        sd_sweep_waves(a.h, a.dk / 4, sg->n_bin16, sg->n_bin64, sw);
        if (n_hub == 0 && sw[0] + sw[1] <= kGmaxPartCap && g_tune[GNPDE_TUNE_GMAX_SMALL] == 0) {
          // small grid: per-wave maxima, folded by the blocks of the second sweep (AttArgs::gmax_part) -- nothing to clear, no fold launch
          c.gmax_part = a.gmax + 64;
          c.gmax_base = 0;
          c.gmax_count = static_cast<int>(sw[0] + sw[1]);
          if (!launch_sd_with_hubs(c, sg->n_bin16, sg->n_bin64, n_hub, part, sg->long_chunk_first, stream)) return GNPDE_ESHAPE;
          GNPDE_LAUNCH_CHECK();
        } else {
          GNPDE_HIP(hipMemsetAsync(a.gmax, 0, 256 + kGmaxSlots * 128, stream));
          c.gmax_slots = a.gmax + 64;          // (256 bytes behind the maximum itself)
          if (!launch_sd_with_hubs(c, sg->n_bin16, sg->n_bin64, n_hub, part, sg->long_chunk_first, stream)) return GNPDE_ESHAPE;
          GNPDE_LAUNCH_CHECK();
          hipLaunchKernelGGL(gmax_fold_kernel, dim3(1), dim3(kWave), 0, stream, c.gmax_slots, a.gmax);
          GNPDE_LAUNCH_CHECK();
          c.gmax_slots = nullptr;
        }
        c.sp_mode = 1;
      }
      if (!launch_sd_with_hubs(c, sg->n_bin16, sg->n_bin64, n_hub, part, sg->long_chunk_first, stream)) return GNPDE_ESHAPE;
      GNPDE_LAUNCH_CHECK();
      return 0;
    }
  }
  if (a.square_plus && pass_only <= 1) GNPDE_HIP(hipMemsetAsync(a.gmax, 0, 256 + kGmaxSlots * 128, stream));    // (scores_kernel: atomic maxima)
  if (pass_only == 0 || pass_only == 1) {
    launch_scores_any(a, vec4, stream_grid(static_cast<long long>(a.e) * a.h), stream);
    GNPDE_LAUNCH_CHECK();
  }
  a.long_segs = (n_long > 0 && long_list != nullptr) ? long_list : nullptr;
  if (pass_only == 0 || pass_only == 2) {
    const dim3 sgrid((g->n + kWavesPerBlock - 1) / kWavesPerBlock);
    switch (a.h) {
      case 1: hipLaunchKernelGGL(seg_stats_heads_kernel<1>, sgrid, dim3(kBlock), 0, stream, a); break;
      case 2: hipLaunchKernelGGL(seg_stats_heads_kernel<2>, sgrid, dim3(kBlock), 0, stream, a); break;
      case 4: hipLaunchKernelGGL(seg_stats_heads_kernel<4>, sgrid, dim3(kBlock), 0, stream, a); break;
      case 8: hipLaunchKernelGGL(seg_stats_heads_kernel<8>, sgrid, dim3(kBlock), 0, stream, a); break;
      default: hipLaunchKernelGGL(seg_stats_kernel, sgrid, dim3(kBlock), 0, stream, a); break;
    }
  }
  GNPDE_LAUNCH_CHECK();
  if (a.long_segs != nullptr && (pass_only == 0 || pass_only == 2)) {
    hipLaunchKernelGGL(seg_stats_long_partial_kernel, dim3(n_long, max_chunks), dim3(kBlock), 0, stream, a, part, max_chunks);
    GNPDE_LAUNCH_CHECK();
    hipLaunchKernelGGL(seg_stats_long_combine_kernel, dim3(n_long), dim3(kWave), 0, stream, a, part, max_chunks);
    GNPDE_LAUNCH_CHECK();
  }
  if (args_out != nullptr) *args_out = a;
  if (stats_only) return 0;
  if (pass_only == 0 || pass_only == 3) {
    hipLaunchKernelGGL(normalise_kernel, dim3(stream_grid(a.e)), dim3(kBlock), 0, stream, a);
    GNPDE_LAUNCH_CHECK();
  }
  return 0;
}

// One pass of the general path at a time (the row-partitioned solver exchanges between them: the global maximum of squareplus
// after pass 1, the per-column partial statistics of attention_norm_idx = 1 after pass 2)
int launch_edge_attention_pass(const gnpde_graph_t* g, const gnpde_attention_t* at, int pass, float* w_mean_csr, void* ws,
                               size_t ws_bytes, hipStream_t stream) {
  GNPDE_CHECK_ARG(pass >= 1 && pass <= 3, GNPDE_EINVAL, "edge_attention_pass: pass must be 1 (scores), 2 (segment statistics) or 3 (normalise)");
  GNPDE_CHECK_ARG(pass != 3 || w_mean_csr != nullptr, GNPDE_EINVAL, "edge_attention_pass: pass 3 writes w_mean_csr");
  float dummy_out = 0.f;   // (the impl wants an output to be named; passes 1 and 2 write none)
  return edge_attention_impl(g, at, pass == 3 ? w_mean_csr : &dummy_out, nullptr, nullptr, ws, ws_bytes, stream, nullptr, false, nullptr,
                             false, pass);
}

// merge of segment statistics (m, den) <- (m, den) (+) (m_in, den_in) for the rows listed: the online-softmax combination,
// or the plain sum for squareplus (its statistics carry no maximum)
__global__ __launch_bounds__(kBlock) void stats_merge_kernel(float* __restrict__ m, float* __restrict__ den, const int* __restrict__ rows,
                                                            int n_rows, int h, const float* __restrict__ m_in,
                                                            const float* __restrict__ den_in, int square_plus) {
  const long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= static_cast<long long>(n_rows) * h) return;
  const int i = static_cast<int>(idx / h), head = static_cast<int>(idx % h);
  const size_t o = static_cast<size_t>(rows != nullptr ? rows[i] : i) * h + head;
  if (square_plus) {
    den[o] = den[o] + den_in[idx];
    return;
  }
  const float ma = m[o], mb = m_in[idx];
  const float mm = fmaxf(ma, mb);
  // (both maxima -inf = a segment without entries on either side: exp(-inf - -inf) would be NaN; its statistics stay (-inf, sum))
  den[o] = mm == -INFINITY ? den[o] + den_in[idx] : den[o] * expf(ma - mm) + den_in[idx] * expf(mb - mm);
  m[o] = mm;
}

int launch_edge_attention(const gnpde_graph_t* g, const gnpde_attention_t* at, float* w_mean_csr, float* att_edge,
                          float* prods_edge, void* ws, size_t ws_bytes, hipStream_t stream, const Fork* fork) {
  return edge_attention_impl(g, at, w_mean_csr, att_edge, prods_edge, ws, ws_bytes, stream, fork, false, nullptr);
}

// w_mean_csr[p] for the entries of the rows longer than GNPDE_LONG_ROW only (row softmax path); the caller attends the
// other rows itself (attn_spmm_kernel)
int launch_hub_attention(const gnpde_graph_t* g, const gnpde_attention_t* at, float* w_mean_csr, void* ws, size_t ws_bytes,
                         hipStream_t stream) {
  return edge_attention_impl(g, at, w_mean_csr, nullptr, nullptr, ws, ws_bytes, stream, nullptr, false, nullptr, true);
}

// ------------------------------------------------------------------------------------------------
// Backward of the normalisation + head mean for EVERY normaliser (softmax or squareplus, over rows or columns):
//   given dw[p] = dL/d(mean_h att[p,h]) / scale  (CSR order; the aggregation's SDDMM r = g_row . x_col)
//   returns ds[p,h] = dL/d(prods[p,h])  -- the scores BEFORE the optional edge re-weighting.
// softmax    a = e^{s-m}/den :  ds = c a (dw - t),            t[seg,h] = sum_{seg} a dw,   c = scale / H
// squareplus a = u/den, u = (z + sqrt(z^2+4))/2, z = s - M :   dz = c (dw - t)/den * u / sqrt(z^2+4),  ds = dz, and the
//            global maximum M (reference src/utils.py:196, `src.max()`) takes -sum(dz), which autograd hands EVENLY to the
//            entries equal to M (evenly_distribute_backward): ds -= [s == M] sum(dz) / #{s == M}.
// Passes: scores + segment statistics (the forward's passes 1-2, recomputed), t by one wavefront per segment (a block per
// long segment), one edge-parallel pass for ds with block partials of (sum dz, #max) summed in a fixed order, and for
// squareplus a second edge-parallel pass that applies the maximum's share.  No atomics.
// ------------------------------------------------------------------------------------------------
namespace {

struct BwdArgs {
  const float* __restrict__ dw;      // [e] CSR order: gradient of the HEAD MEAN (shared by the heads), or
  const float* __restrict__ datt;    // [E,h] caller's edge order: gradient of every head's attention (then dw == nullptr)
  const int* __restrict__ perm;      // CSR position -> edge id (for datt)
  int post;                          // 0: ds w.r.t. the raw scores (times edge_w); 1: (ds w.r.t. the score) * score  -- the
                                     //    common factor of every derivative of an exp kernel; 2: ds * leaky'(score) (GAT)
  float leaky_slope;
  float* t;                          // [n,h]
  float* ds;                         // [e,h]
  float* partial;                    // [grid,2]: sum dz, count of maxima
  const float* __restrict__ scale_ptr;
  int scale_sigmoid;
};

__device__ __forceinline__ float grad_in(const BwdArgs& b, long long p, int head, int h) {
  return b.dw != nullptr ? b.dw[p] : b.datt[static_cast<long long>(b.perm[p]) * h + head];
}

__device__ __forceinline__ float att_value(const AttArgs& a, float s, int seg, int head, float gmax) {
  const float den = a.seg_den[static_cast<size_t>(seg) * a.h + head];
  if (a.square_plus) return squareplus_num(s, gmax) / den;
  return expf(s - a.seg_m[static_cast<size_t>(seg) * a.h + head]) / den;
}

// t[seg,h] = sum over the segment of att * dw; BLOCK = false: a wavefront per segment (long ones skipped),
// BLOCK = true: a block per long segment
template <bool BLOCK>
__global__ __launch_bounds__(kBlock) void seg_dot_kernel(const AttArgs a, const BwdArgs b) {
  __shared__ float red[kWavesPerBlock];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  int seg;
  if (BLOCK) {
    seg = a.long_segs[blockIdx.x];
  } else {
    seg = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * kWavesPerBlock + wave);
    if (seg >= a.n) return;
  }
  const int s0 = a.segptr[seg], s1 = a.segptr[seg + 1];
  if (!BLOCK && s1 - s0 > GNPDE_LONG_ROW && a.long_segs != nullptr) return;
  const float gmax = a.square_plus ? ord2f(*a.gmax) : 0.f;
  const int first = BLOCK ? static_cast<int>(threadIdx.x) : lane;
  const int step = BLOCK ? kBlock : kWave;
  for (int head = 0; head < a.h; ++head) {
    float sum = 0.f;
    for (int t = s0 + first; t < s1; t += step) {
      const int p = a.segpos ? a.segpos[t] : t;
      sum += att_value(a, a.scores[static_cast<size_t>(p) * a.h + head], seg, head, gmax) * grad_in(b, p, head, a.h);
    }
    sum = wave_sum(sum);
    if (BLOCK) {
      if (lane == 0) red[wave] = sum;
      __syncthreads();
      if (threadIdx.x == 0) b.t[static_cast<size_t>(seg) * a.h + head] = (red[0] + red[1]) + (red[2] + red[3]);
      __syncthreads();
    } else if (lane == 0) {
      b.t[static_cast<size_t>(seg) * a.h + head] = sum;
    }
  }
}

// what multiplies d L / d score: edge_w (raw-score gradient), the score itself (exp kernels), or LeakyReLU' (GAT)
__device__ __forceinline__ float post_factor(const BwdArgs& b, float s, float ew) {
  if (b.post == 1) return s;
  if (b.post == 2) return s > 0.f ? 1.0f : b.leaky_slope;
  return ew;
}

__global__ __launch_bounds__(kBlock) void att_bwd_edge_kernel(const AttArgs a, const BwdArgs b) {
  __shared__ float rs[kWavesPerBlock], rc[kWavesPerBlock];
  float scale = 1.0f;
  if (b.scale_ptr != nullptr) {
    scale = *b.scale_ptr;
    if (b.scale_sigmoid) scale = 1.0f / (1.0f + expf(-scale));
  }
  const float c = b.dw != nullptr ? scale / static_cast<float>(a.h) : scale;   // the head mean's 1/H only for a shared dw
  const float gmax = a.square_plus ? ord2f(*a.gmax) : 0.f;
  float lsum = 0.f, lcnt = 0.f;
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long p = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; p < a.e; p += stride) {
    const int seg = a.norm_idx == 0 ? a.rowidx[p] : a.colidx[p];
    const float ew = a.edge_w != nullptr ? a.edge_w[p] : 1.0f;
    for (int head = 0; head < a.h; ++head) {
      const float s = a.scores[p * a.h + head];
      const float v = grad_in(b, p, head, a.h) - b.t[static_cast<size_t>(seg) * a.h + head];
      float out;
      if (a.square_plus) {
        const float z = s - gmax;
        const float den = a.seg_den[static_cast<size_t>(seg) * a.h + head];
        const float root = sqrtf(z * z + 4.0f);
        out = c * v / den * (0.5f * (z + root)) / root;    // du/dz = u / sqrt(z^2 + 4)
        lsum += out;
        lcnt += (s == gmax) ? 1.0f : 0.0f;
      } else {
        out = c * att_value(a, s, seg, head, gmax) * v;
      }
      b.ds[p * a.h + head] = out * post_factor(b, s, ew);
    }
  }
  if (a.square_plus) {   // block partials, folded in block order by att_bwd_max_share_kernel
    lsum = wave_sum(lsum);
    lcnt = wave_sum(lcnt);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    if (lane == 0) { rs[wave] = lsum; rc[wave] = lcnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
      b.partial[2 * blockIdx.x] = (rs[0] + rs[1]) + (rs[2] + rs[3]);
      b.partial[2 * blockIdx.x + 1] = (rc[0] + rc[1]) + (rc[2] + rc[3]);
    }
  }
}

// squareplus: ds -= [s == M] sum(dz) / #{s == M}   (every block folds the partials in the same order)
__global__ __launch_bounds__(kBlock) void att_bwd_max_share_kernel(const AttArgs a, const BwdArgs b, int n_partials) {
  __shared__ float share;
  if (threadIdx.x == 0) {
    float tot = 0.f, cnt = 0.f;
    for (int i = 0; i < n_partials; ++i) { tot += b.partial[2 * i]; cnt += b.partial[2 * i + 1]; }
    share = cnt > 0.f ? tot / cnt : 0.f;
  }
  __syncthreads();
  const float gmax = ord2f(*a.gmax);
  const long long total = static_cast<long long>(a.e) * a.h;
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
    if (a.scores[i] == gmax) {
      const float ew = a.edge_w != nullptr ? a.edge_w[i / a.h] : 1.0f;
      b.ds[i] -= share * post_factor(b, a.scores[i], ew);
    }
  }
}

constexpr int kBwdPartials = 2048;

}  // namespace

size_t attention_bwd_workspace_bytes(const gnpde_graph_t* g, int h, bool gat) {
  return align_up(attention_workspace_bytes(g, h, gat), 256) + align_up(static_cast<size_t>(g->n) * h * 4, 256) + kBwdPartials * 2 * 4;
}

int launch_edge_attention_bwd(const gnpde_graph_t* g, const gnpde_attention_t* at, const float* dw_csr, const float* datt_edge,
                              int post, const float* scale, int scale_sigmoid, float* ds_csr, void* ws, size_t ws_bytes,
                              hipStream_t stream) {
  GNPDE_CHECK_ARG(g && at && ds_csr && ((dw_csr != nullptr) != (datt_edge != nullptr)) && post >= 0 && post <= 2, GNPDE_EINVAL,
                  "edge_attention_bwd: need exactly one of dw_csr / datt_edge, post in 0..2");
  if (g->e == 0 || g->n == 0) return 0;
  const bool gat = at->type == GNPDE_ATT_GAT;
  const size_t fwd = align_up(attention_workspace_bytes(g, at->heads, gat), 256);
  const size_t need = attention_bwd_workspace_bytes(g, at->heads, gat);
  GNPDE_CHECK_ARG(ws && ws_bytes >= need, GNPDE_EWS, "edge_attention_bwd: workspace %zu < %zu bytes", ws_bytes, need);
  AttArgs a{};
  int rc = edge_attention_impl(g, at, nullptr, nullptr, nullptr, ws, fwd, stream, nullptr, true, &a);
  if (rc) return rc;
  BwdArgs b{};
  b.dw = dw_csr;
  b.datt = datt_edge;
  b.perm = g->perm;
  b.post = post;
  b.leaky_slope = at->leaky_slope;
  b.ds = ds_csr;
  b.t = reinterpret_cast<float*>(static_cast<char*>(ws) + fwd);
  b.partial = reinterpret_cast<float*>(static_cast<char*>(ws) + fwd + align_up(static_cast<size_t>(g->n) * at->heads * 4, 256));
  b.scale_ptr = scale;
  b.scale_sigmoid = scale_sigmoid;
  hipLaunchKernelGGL(seg_dot_kernel<false>, dim3((g->n + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, stream, a, b);
  GNPDE_LAUNCH_CHECK();
  if (a.long_segs != nullptr) {
    const int n_long = at->norm_idx == 0 ? g->n_long_rows : g->n_long_cols;
    hipLaunchKernelGGL(seg_dot_kernel<true>, dim3(n_long), dim3(kBlock), 0, stream, a, b);
    GNPDE_LAUNCH_CHECK();
  }
  unsigned grid = stream_grid(a.e);
  if (grid > kBwdPartials) grid = kBwdPartials;
  hipLaunchKernelGGL(att_bwd_edge_kernel, dim3(grid), dim3(kBlock), 0, stream, a, b);
  GNPDE_LAUNCH_CHECK();
  if (a.square_plus) {
    hipLaunchKernelGGL(att_bwd_max_share_kernel, dim3(stream_grid(static_cast<long long>(a.e) * a.h)), dim3(kBlock), 0, stream,
                       a, b, static_cast<int>(grid));
    GNPDE_LAUNCH_CHECK();
  }
  return 0;
}

size_t attention_workspace_bytes_rows(const gnpde_graph_t* g, int h, bool gat, int key_rows) {
  return att_layout(g->n, g->e, h, gat, long_slots_of(g), key_rows).total;
}

size_t attention_workspace_bytes(const gnpde_graph_t* g, int h, bool gat) {
  return att_layout(g->n, g->e, h, gat, long_slots_of(g)).total;
}

}  // namespace gnpde

extern "C" size_t gnpde_attention_workspace_bytes(const gnpde_graph_t* g, const gnpde_attention_t* a) {
  if (!g || !a || a->heads < 1) return 0;
  return gnpde::attention_workspace_bytes_rows(g, a->heads, a->type == GNPDE_ATT_GAT, a->n_key_rows);
}

extern "C" int gnpde_edge_attention_pass(const gnpde_graph_t* g, const gnpde_attention_t* a, int32_t pass, float* w_mean_csr,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  return gnpde::launch_edge_attention_pass(g, a, pass, w_mean_csr, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

namespace gnpde {
AttLayoutView att_layout_view(const gnpde_graph_t* g, const gnpde_attention_t* a) {
  const AttLayout L = att_layout(g->n, g->e, a->heads, a->type == GNPDE_ATT_GAT, long_slots_of(g), a->n_key_rows);
  return AttLayoutView{L.seg_m, L.seg_den, L.gmax, L.total};
}
}  // namespace gnpde

extern "C" int gnpde_attention_workspace_regions(const gnpde_graph_t* g, const gnpde_attention_t* a, size_t* offsets) {
  GNPDE_CHECK_ARG(g && a && offsets && a->heads >= 1, GNPDE_EINVAL, "attention_workspace_regions: bad arguments");
  const gnpde::AttLayout L = gnpde::att_layout(g->n, g->e, a->heads, a->type == GNPDE_ATT_GAT, gnpde::long_slots_of(g), a->n_key_rows);
  offsets[0] = L.scores; offsets[1] = L.seg_m; offsets[2] = L.seg_den; offsets[3] = L.gmax;
  return 0;
}

extern "C" int gnpde_segment_stats_merge(float* seg_m, float* seg_den, const int32_t* rows, int32_t n_rows, int32_t heads,
                                         const float* m_in, const float* den_in, int32_t square_plus, void* stream) {
  GNPDE_CHECK_ARG(n_rows >= 0 && heads >= 1, GNPDE_EINVAL, "segment_stats_merge: bad shape");
  if (n_rows == 0) return 0;
  GNPDE_CHECK_ARG(seg_den && den_in && (square_plus || (seg_m && m_in)), GNPDE_EINVAL, "segment_stats_merge: null pointer");
  const long long items = static_cast<long long>(n_rows) * heads;
  hipLaunchKernelGGL(gnpde::stats_merge_kernel, dim3(static_cast<unsigned>((items + gnpde::kBlock - 1) / gnpde::kBlock)),
                     dim3(gnpde::kBlock), 0, static_cast<hipStream_t>(stream), seg_m, seg_den, rows, n_rows, heads, m_in, den_in, square_plus);
  GNPDE_LAUNCH_CHECK();
  return 0;
}

extern "C" int gnpde_edge_attention(const gnpde_graph_t* g, const gnpde_attention_t* a, float* w_mean_csr, float* att_edge,
                                    float* prods_edge, void* workspace, size_t workspace_bytes, void* stream) {
  return gnpde::launch_edge_attention(g, a, w_mean_csr, att_edge, prods_edge, workspace, workspace_bytes,
                                      static_cast<hipStream_t>(stream), nullptr);
}

extern "C" size_t gnpde_attention_bwd_workspace_bytes(const gnpde_graph_t* g, const gnpde_attention_t* a) {
  if (!g || !a || a->heads < 1) return 0;
  return gnpde::attention_bwd_workspace_bytes(g, a->heads, a->type == GNPDE_ATT_GAT);
}

extern "C" int gnpde_edge_attention_bwd(const gnpde_graph_t* g, const gnpde_attention_t* a, const float* dw_csr, const float* scale,
                                        int32_t scale_sigmoid, float* ds_csr, void* workspace, size_t workspace_bytes, void* stream) {
  return gnpde::launch_edge_attention_bwd(g, a, dw_csr, nullptr, 0, scale, scale_sigmoid, ds_csr, workspace, workspace_bytes,
                                          static_cast<hipStream_t>(stream));
}

extern "C" int gnpde_edge_attention_bwd_heads(const gnpde_graph_t* g, const gnpde_attention_t* a, const float* datt_edge,
                                              int32_t post, float* ds_csr, void* workspace, size_t workspace_bytes, void* stream) {
  return gnpde::launch_edge_attention_bwd(g, a, nullptr, datt_edge, post, nullptr, 0, ds_csr, workspace, workspace_bytes,
                                          static_cast<hipStream_t>(stream));
}

extern "C" int gnpde_edge_to_csr_mean(const gnpde_graph_t* g, const float* src_edge, int32_t h, float* w_csr, void* stream) {
  GNPDE_CHECK_ARG(g && g->perm && src_edge && w_csr && h >= 1, GNPDE_EINVAL, "edge_to_csr_mean: bad arguments");
  if (g->e == 0) return 0;
  hipLaunchKernelGGL(gnpde::edge_to_csr_mean_kernel, dim3(gnpde::stream_grid(g->e)), dim3(gnpde::kBlock), 0,
                     static_cast<hipStream_t>(stream), g->perm, src_edge, h, g->e, w_csr);
  GNPDE_LAUNCH_CHECK();
  return 0;
}
