// Native adjoint solve: the backward pass of ODEblock.forward when opt['adjoint'] is set (reference src/base_classes.py:44-47,
// src/block_constant.py:45-55 hand the integration to torchdiffeq.odeint_adjoint) for the fixed-grid adjoint methods
// (`adjoint_method` euler / rk4 == 3/8 rule, `adjoint_step_size`).  torchdiffeq integrates the augmented system
//     y' = f(y),   a' = -a^T df/dy,   g' = -a^T df/dtheta
// backwards in time through the substitution s = -t and evaluates f and its vector-Jacobian products with autograd -- per stage
// one forward of ODEFunc.forward (reference src/function_transformer_attention.py:38-53, src/function_laplacian_diffusion.py:38-51),
// one autograd backward through index_select / scatter_add / softmax, and a dozen elementwise launches of stage algebra.
// Here one stage is a fixed sequence of this library's kernels and the whole backward solve is ONE captured hipGraph:
//   q||k = u_y [Wq;Wk]^T + b                     (fp32 MFMA projection, linear.hip)
//   w    = head mean of the normalised attention (attention.hip, the inference kernels)
//   F    = alpha (A u_y - u_y) + beta x0         (aggregation, spmm.hip) -- its epilogue forms the NEXT stage input u_y
//   r_e  = u_a[row] . u_y[col]                   (SDDMM)
//   ds   = normaliser backward (alpha / H folded in); dq = sum_row ds k, dk = sum_col ds q  (backward.hip)
//   P    = [dq dk] [Wq;Wk]                       (projection kernel on the transposed weights)
//   V    = alpha (A^T u_a - u_a) + P             (aggregation on the transposed CSR with the weights permuted) -- its epilogue
//                                                  forms the NEXT stage input u_a
//   g   += c ([dq dk]^T u_y, colsum [dq dk], d alpha, d beta)   (one pass over the rows + a fold, below)
// No PyTorch op runs inside the solve.  GRAND-l (constant weights) is the same without the attention steps.
#include <vector>
#include "common.h"
#include "rhs.h"
#include "epilogue.h"

namespace gnpde {

int launch_linear_any(const float* x, int n, int d, int ldx, const float* W, int m, int ldw, const float* b, float* out,
                      int ldo, hipStream_t s, int relu = 0);
int launch_edge_attention(const gnpde_graph_t* g, const gnpde_attention_t* at, float* w_mean_csr, float* att_edge,
                          float* prods_edge, void* ws, size_t ws_bytes, hipStream_t stream, const Fork* fork);
size_t attention_workspace_bytes(const gnpde_graph_t* g, int h, bool gat);
int launch_normalise_heads(float* qk, long long n, int ld, int att_dim, int heads, bool centre, hipStream_t s, float* inv_out = nullptr);
int launch_normalise_heads_bwd(const float* out, float* g, long long n, int ld, int att_dim, int heads, bool centre, const float* inv, hipStream_t s);
bool normalise_heads_bwd_supported(int att_dim, int heads);
int launch_edge_attention_bwd(const gnpde_graph_t* g, const gnpde_attention_t* at, const float* dw_csr, const float* datt_edge,
                              int post, const float* scale, int scale_sigmoid, float* ds_csr, void* ws, size_t ws_bytes,
                              hipStream_t stream);

namespace {

constexpr int kParamBlocks = 512;      // row slabs of the parameter-gradient pass
constexpr int kGramTile = 32;          // rows of the Gram block (m) one workgroup accumulates

struct ParamArgs {
  const float* __restrict__ dqk;   // [n, M] or null (GRAND-l)
  const float* __restrict__ uy;    // [n, ld] stage input of the state
  const float* __restrict__ ua;    // [n, ld] stage input of the adjoint
  const float* __restrict__ F;     // [n, ld] f(u_y) of this stage
  const float* __restrict__ x0;    // [n, ld] or null
  int n, d, ld, M;
  int rows_per_block;
  int stride;                      // floats per block partial: M d + M + 2
  float* __restrict__ partial;     // [gridDim.x, stride]
};

// partial[b][m, c] = sum_{i in slab b} dqk[i, m] u_y[i, c]   and   partial[b][M d + m] = sum_i dqk[i, m]
// The weight gradients d[Wq;Wk] = [dq dk]^T u_y are a [M, n] x [n, d] product with n >> M, d: every workgroup takes a slab of rows
// and a tile of kGramTile m's; its four wavefronts walk the slab's rows interleaved.  A lane owns VC consecutive columns of the
// state row (one coalesced load per row and wave); the row of dqk is wave-uniform, so its kGramTile values arrive by SCALAR loads
// and feed the FMAs as SGPR operands: kGramTile x VC accumulators per lane, no cross-lane traffic until the four waves fold their
// blocks through the LDS (fixed order).  Lanes 0..kGramTile-1 also keep the column sums of dqk (the bias gradients).
template <int VC>
__global__ __launch_bounds__(kBlock) void adjoint_gram_kernel(const ParamArgs p) {
  __shared__ float fold[3][8][kWave * VC];
  __shared__ float bfold[3][kGramTile];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int col = lane * VC;
  const int m0 = static_cast<int>(blockIdx.y) * kGramTile;
  const int mt = p.M - m0 < kGramTile ? p.M - m0 : kGramTile;     // live m's of this tile (a multiple of 4)
  const int r0 = static_cast<int>(blockIdx.x) * p.rows_per_block;
  int r1 = r0 + p.rows_per_block;
  if (r1 > p.n) r1 = p.n;
  float acc[kGramTile][VC];
#pragma unroll
  for (int m = 0; m < kGramTile; ++m)
#pragma unroll
    for (int c = 0; c < VC; ++c) acc[m][c] = 0.f;
  float bsum = 0.f;
  float cm[VC];
#pragma unroll
  for (int c = 0; c < VC; ++c) cm[c] = col + c < p.d ? 1.0f : 0.0f;   // padded rows: columns [d, ld) are not data
  const bool col_ok = col < p.d;
  // R rows per iteration, every load (the rows of u_y, and the rows of dqk one value per lane) issued before the first FMA; the
  // wave-uniform dqk[i, m] is handed to the FMAs by v_readlane (an SGPR operand): no scalar-memory round trip per row
  constexpr int R = 8;
  for (int i0 = r0 + wave; i0 < r1; i0 += R * kWavesPerBlock) {
    float x[R][VC], ql[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int i = i0 + j * kWavesPerBlock;
      const bool live = i < r1;                    // wave-uniform
#pragma unroll
      for (int c = 0; c < VC; ++c) x[j][c] = 0.f;
      ql[j] = 0.f;
      if (live) {
        if (col_ok) load_vec<VC>(p.uy + static_cast<size_t>(i) * p.ld + col, x[j]);
        if (lane < mt) ql[j] = p.dqk[static_cast<size_t>(i) * p.M + m0 + lane];
      }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
#pragma unroll
      for (int c = 0; c < VC; ++c) x[j][c] *= cm[c];
      bsum += ql[j];
#pragma unroll
      for (int m = 0; m < kGramTile; ++m) {
        const float q = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ql[j]), m));
#pragma unroll
        for (int c = 0; c < VC; ++c) acc[m][c] = fmaf(q, x[j][c], acc[m][c]);
      }
    }
  }
  // fold the four waves in slices of 8 m's
  float* out = p.partial + static_cast<size_t>(blockIdx.x) * p.stride;
#pragma unroll
  for (int s = 0; s < kGramTile / 8; ++s) {
    if (wave > 0) {
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int c = 0; c < VC; ++c) fold[wave - 1][m][c * kWave + lane] = acc[s * 8 + m][c];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int mm = s * 8 + m;
#pragma unroll
        for (int c = 0; c < VC; ++c) {
          const float v = ((acc[mm][c] + fold[0][m][c * kWave + lane]) + fold[1][m][c * kWave + lane]) + fold[2][m][c * kWave + lane];
          if (mm < mt && col + c < p.d) out[static_cast<size_t>(m0 + mm) * p.d + col + c] = v;
        }
      }
    }
    __syncthreads();
  }
  if (wave > 0 && lane < kGramTile) bfold[wave - 1][lane] = bsum;
  __syncthreads();
  if (wave == 0 && lane < mt) out[static_cast<size_t>(p.M) * p.d + m0 + lane] = ((bsum + bfold[0][lane]) + bfold[1][lane]) + bfold[2][lane];
}

// The same Gram block on the fp32 matrix cores (exact fp32 products and sums, as the projection of csrc/linear.hip).  The product
// [M, n] x [n, d] has its K index along the ROWS of both operands, so the v_mfma_f32_16x16x4_f32 operand shapes are row-major memory
// as it lies: lane (j = l & 15, kq = l >> 4) of a K step over rows i0..i0+3 reads row i0 + kq -- MV consecutive floats of dqk at
// m = MV j (A operand of M tile jm: its component jm, i.e. m = MV j + jm) and one float4 of u_y per 64 columns at c = 64 g + 4 j
// (B operand of N tile (g, jn): component jn, i.e. c = 64 g + 4 j + jn).  The tiles are therefore strided sets of m's / columns --
// a permutation of the output, undone by the store.  Three loads feed 16 MFMAs (M = 32, d = 128); the VALU kernel above spends
// 4096 FMAs per row on operands it reads once (42.7 us at the ogbn-arxiv shape; this one 28 us, profiles/r04_train_v8_*).
// A wavefront owns every fourth K step of its slab; the four waves fold through the LDS in wave order.
typedef float gram_f32x4 __attribute__((ext_vector_type(4)));

template <int MV, int NG>
__global__ __launch_bounds__(kBlock) void adjoint_gram_mfma_kernel(const ParamArgs p) {
  constexpr int NT = NG * 4;                       // N tiles (16 columns each)
  constexpr int KU = 4;                            // K steps (of 4 rows) in flight per wave
  __shared__ gram_f32x4 fold[kWavesPerBlock][MV * NT][kWave];
  __shared__ float bfold[kWavesPerBlock][MV][16];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int j = lane & 15, kq = lane >> 4;
  const int r0 = static_cast<int>(blockIdx.x) * p.rows_per_block;
  int r1 = r0 + p.rows_per_block;
  if (r1 > p.n) r1 = p.n;
  gram_f32x4 acc[MV][NT];
#pragma unroll
  for (int t = 0; t < MV; ++t)
#pragma unroll
    for (int u = 0; u < NT; ++u) acc[t][u] = gram_f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[MV];
#pragma unroll
  for (int t = 0; t < MV; ++t) bsum[t] = 0.f;
  for (int i0 = r0 + 4 * wave; i0 < r1; i0 += 4 * KU * kWavesPerBlock) {
    float av[KU][MV];
    gram_f32x4 bv[KU][NG];
#pragma unroll
    for (int s = 0; s < KU; ++s) {
      const int i = i0 + 4 * kWavesPerBlock * s + kq;
      const bool live = i < r1;
#pragma unroll
      for (int t = 0; t < MV; ++t) av[s][t] = 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g) bv[s][g] = gram_f32x4{0.f, 0.f, 0.f, 0.f};
      if (live) {
        const float* qa = p.dqk + static_cast<size_t>(i) * p.M + MV * j;
        if constexpr (MV == 1) av[s][0] = qa[0];
        else if constexpr (MV == 2) { const float2 v = *reinterpret_cast<const float2*>(qa); av[s][0] = v.x; av[s][1] = v.y; }
        else { const float4 v = *reinterpret_cast<const float4*>(qa); av[s][0] = v.x; av[s][1] = v.y; av[s][2] = v.z; av[s][3] = v.w; }
        const float* xa = p.uy + static_cast<size_t>(i) * p.ld + 4 * j;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const float4 v = *reinterpret_cast<const float4*>(xa + 64 * g);
          bv[s][g] = gram_f32x4{v.x, v.y, v.z, v.w};
        }
      }
    }
#pragma unroll
    for (int s = 0; s < KU; ++s) {
#pragma unroll
      for (int t = 0; t < MV; ++t) {
        bsum[t] += av[s][t];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int jn = 0; jn < 4; ++jn)
            acc[t][g * 4 + jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][t], bv[s][g][jn], acc[t][g * 4 + jn], 0, 0, 0);
      }
    }
  }
  // column sums of dqk: the four row groups of a wave, then the waves
#pragma unroll
  for (int t = 0; t < MV; ++t) {
    bsum[t] += __shfl_xor(bsum[t], 16, kWave);
    bsum[t] += __shfl_xor(bsum[t], 32, kWave);
    if (lane < 16) bfold[wave][t][lane] = bsum[t];
  }
#pragma unroll
  for (int t = 0; t < MV; ++t)
#pragma unroll
    for (int u = 0; u < NT; ++u) fold[wave][t * NT + u][lane] = acc[t][u];
  __syncthreads();
  float* out = p.partial + static_cast<size_t>(blockIdx.x) * p.stride;
  // wave w folds the tiles q = w, w + 4, ...: C layout of a tile = (row 4 kq + i, column j)
  for (int q = wave; q < MV * NT; q += kWavesPerBlock) {
    const gram_f32x4 v = ((fold[0][q][lane] + fold[1][q][lane]) + fold[2][q][lane]) + fold[3][q][lane];
    const int t = q / NT, u = q % NT;
    const int c = 64 * (u / 4) + 4 * j + (u % 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = MV * (4 * kq + i) + t;
      out[static_cast<size_t>(m) * p.d + c] = v[i];
    }
  }
  if (wave == 0 && lane < 16) {
#pragma unroll
    for (int t = 0; t < MV; ++t)
      out[static_cast<size_t>(p.M) * p.d + MV * lane + t] = ((bfold[0][t][lane] + bfold[1][t][lane]) + bfold[2][t][lane]) + bfold[3][t][lane];
  }
}

// partial[b][M d + M + {0, 1}] = sum of the per-wave dots w = b, b + nb, b + 2 nb, ... written by the row kernel of the stage
// (launch_adjoint_rows): first level of the fold of sum u_a . F and sum u_a . x0, fixed order
__global__ __launch_bounds__(kBlock) void adjoint_dots_fold_kernel(const float* __restrict__ dots, int n_dots, int nb, float* __restrict__ partial,
                                                                  int stride, int slot) {
  // (the per-row dots are sums of opposite signs -- <u, A v> against <u, v> -- and there are n of them: this level is summed in double,
  //  round 6; the nb block results go on as floats)
  __shared__ double red[kWavesPerBlock][2];
  double d1 = 0.0, d2 = 0.0;
  for (long long w = static_cast<long long>(blockIdx.x) + static_cast<long long>(threadIdx.x) * nb; w < n_dots; w += static_cast<long long>(kBlock) * nb) {
    d1 += static_cast<double>(dots[2 * w]);
    d2 += static_cast<double>(dots[2 * w + 1]);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    d1 += __shfl_xor(d1, off, kWave);
    d2 += __shfl_xor(d2, off, kWave);
  }
  if ((threadIdx.x & (kWave - 1)) == 0) { red[threadIdx.x >> 6][0] = d1; red[threadIdx.x >> 6][1] = d2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* out = partial + static_cast<size_t>(blockIdx.x) * stride + slot;
    out[0] = static_cast<float>((red[0][0] + red[1][0]) + (red[2][0] + red[3][0]));
    out[1] = static_cast<float>((red[0][1] + red[1][1]) + (red[2][1] + red[3][1]));
  }
}

// grads[idx] += coef * sum_b partial[b][idx]; the two dot slots become  d alpha_train = (1 - sigma)(sum u_a.F - beta sum u_a.x0)
// (f - beta x0 = sigma (A x - x), d sigma / d alpha_train = sigma (1 - sigma))  and  d beta_train = sum u_a.x0.
// Block = 64 outputs x 4 slices of the slabs, slices folded through the LDS in order.
__global__ __launch_bounds__(kBlock) void adjoint_param_fold_kernel(const float* __restrict__ partial, int nb, int stride, int n_plain,
                                                                   float coef, const float* __restrict__ alpha,
                                                                   const float* __restrict__ beta, int has_source, int alpha_sigmoid,
                                                                   int source_in_s1, float* __restrict__ grads) {
  constexpr int OUTS = 32, SL = kBlock / OUTS;
  __shared__ float part[SL][OUTS][2];
  const int o = threadIdx.x % OUTS, sl = threadIdx.x / OUTS;
  const int idx = static_cast<int>(blockIdx.x) * OUTS + o;
  const bool dot_slot = idx == n_plain;                       // this thread folds BOTH dot sums
  const bool live = idx < n_plain || dot_slot;
  float s1 = 0.f, s2 = 0.f;
  if (live) {
    const int per = (nb + SL - 1) / SL;
    int b0 = sl * per, b1 = b0 + per;
    if (b1 > nb) b1 = nb;
    const float* p = partial + static_cast<size_t>(b0) * stride + idx;
    int b = b0;
    for (; b + 8 <= b1; b += 8, p += 8 * static_cast<size_t>(stride)) {
      float v[8], w[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        v[t] = p[t * static_cast<size_t>(stride)];
        w[t] = dot_slot ? p[t * static_cast<size_t>(stride) + 1] : 0.f;
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) { s1 += v[t]; s2 += w[t]; }
    }
    for (; b < b1; ++b, p += stride) {
      s1 += p[0];
      if (dot_slot) s2 += p[1];
    }
  }
  part[sl][o][0] = s1;
  part[sl][o][1] = s2;
  __syncthreads();
  if (sl != 0 || !live) return;
  s1 = 0.f; s2 = 0.f;
#pragma unroll
  for (int t = 0; t < SL; ++t) { s1 += part[t][o][0]; s2 += part[t][o][1]; }
  if (!dot_slot) {
    grads[idx] = fmaf(coef, s1, grads[idx]);
    return;
  }
  if (!alpha_sigmoid) {        // opt['no_alpha_sigmoid']: alpha' = alpha_train, and the row kernel summed u_a . (A u - u) itself
    grads[n_plain] = fmaf(coef, s1, grads[n_plain]);
    grads[n_plain + 1] = fmaf(coef, has_source ? s2 : 0.f, grads[n_plain + 1]);
    return;
  }
  const float a = *alpha;
  const float sig = 1.0f / (1.0f + expf(-a));
  // (source_in_s1: the row kernel summed u_a . F with the source term inside F; the cotangent-side sweep sums u_y . S, which has none --
  //  nothing to subtract, and no cancellation against beta <u_a, x0>)
  const float b = (has_source && source_in_s1) ? *beta : 0.f;
  grads[n_plain] = fmaf(coef, (1.0f - sig) * (s1 - b * s2), grads[n_plain]);
  grads[n_plain + 1] = fmaf(coef, has_source ? s2 : 0.f, grads[n_plain + 1]);
}

// dst[i] = src[idx[i]].  Eight elements per thread: two 16-byte loads of the map, all eight gathers issued before the first use, two 16-byte
// stores -- the one-element-per-thread form of round 4 kept one 4-byte gather in flight per lane and ran at 0.4 TB/s on a table that sits
// in the L2s (25 us for the 2.48 M weights of the ogbn-arxiv shape).
__global__ __launch_bounds__(kBlock) void permute_f32_kernel(const float* __restrict__ src, const int* __restrict__ idx, int n,
                                                            float* __restrict__ dst) {
  const long long base = (static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x) * 8;
  if (base + 8 <= n) {
    const int4 a = *reinterpret_cast<const int4*>(idx + base), b = *reinterpret_cast<const int4*>(idx + base + 4);
    const float v0 = src[a.x], v1 = src[a.y], v2 = src[a.z], v3 = src[a.w], v4 = src[b.x], v5 = src[b.y], v6 = src[b.z], v7 = src[b.w];
    *reinterpret_cast<float4*>(dst + base) = make_float4(v0, v1, v2, v3);
    *reinterpret_cast<float4*>(dst + base + 4) = make_float4(v4, v5, v6, v7);
  } else {
    for (long long i = base; i < n; ++i) dst[i] = src[idx[i]];
  }
}

// ---- reverse sweep over a recorded solve, cotangent side only (round 6).  The stage's row kernel runs on the TRANSPOSED graph with the
// roles exchanged (gathered operand = the cotangent u_a, own row = the recorded stage input u_y): one pass over the gathered rows gives
// S = alpha' (A^T u_a - u_a), the edge products r = u_y[row'] . u_a[col'] in the transposed order and <u_y, S> = alpha' <u_a, A u_y - u_y>
// (the alpha gradient) -- the state side needs no gather of its own, because nothing integrates it.  GRAND-nl / GAT then need the
// attention backward before the stage algebra can close: combine = stage algebra with k = S + P, and <u_a, x0> for d beta on the way.
struct CombineArgs {
  const float* S; const float* P;      // [n, ld]; P may be null
  const float* ua;                     // the row's own stage input of the cotangent recursion
  const float* x0;                     // or null
  const float* beta;                   // device scalar (with x0)
  int alpha_sigmoid;
  long long n4;                        // float4s of a state buffer (n * ld / 4)
  gnpde_epilogue_t ep;                 // stage, dt, y, k1, out_y / out_k
  float* pairs;                        // [gridDim.x][2]: (0, <u_a, x0>) per block, appended to the row kernel's dots
};

__global__ __launch_bounds__(kBlock) void stage_combine_kernel(const CombineArgs a) {
  __shared__ float red[kWavesPerBlock];
  float t = 0.f;
  for (long long i = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; i < a.n4; i += static_cast<long long>(gridDim.x) * kBlock) {
    const size_t off = static_cast<size_t>(i) * 4;
    float k[4], ui[4];
    load_vec<4>(a.S + off, k);
    load_vec<4>(a.ua + off, ui);
    if (a.P != nullptr) {
      float pv[4];
      load_vec<4>(a.P + off, pv);
#pragma unroll
      for (int v = 0; v < 4; ++v) k[v] = k[v] + 1.0f * pv[v];          // (as the aggregation epilogue formed it: k + beta s with beta = 1)
    }
    if (a.x0 != nullptr) {
      float xv[4];
      load_vec<4>(a.x0 + off, xv);
#pragma unroll
      for (int v = 0; v < 4; ++v) t = fmaf(ui[v], xv[v], t);          // (padding columns hold zeros on both sides)
    }
    stage_store<4, false>(a.ep, off, k, ui);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    a.pairs[2 * static_cast<size_t>(blockIdx.x)] = 0.f;
    a.pairs[2 * static_cast<size_t>(blockIdx.x) + 1] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}
constexpr int kCombineBlocks = 1024;

// ---- GAT (reference src/function_GAT_attention.py:105-115): score_eh = LeakyReLU(ts_ih + td_jh), ts_ih = sum_c a[c] wx[i, h, c],
// td_jh = sum_c a[d_k + c] wx[j, h, c], wx = u W.  With c_eh = dL/d(score) LeakyReLU'(score) from the normaliser backward:
//     d ts_ih = sum over row i of c,   d td_jh = sum over column j of c,
//     d wx[i, h, c] = d ts_ih a[c] + d td_ih a[d_k + c],   d a[c] = sum_ih d ts_ih wx[i, h, c],   d a[d_k + c] = sum_ih d td_ih wx[i, h, c].
// out[i, h] = sum over the entries p of row i of c[pos ? pos[p] : p, h]: a wavefront per row, lanes over the entries, fixed fold
constexpr int kGatMaxHeads = 8;
__global__ __launch_bounds__(kBlock) void gat_head_sums_kernel(const int* __restrict__ rowptr, const int* __restrict__ pos,
                                                              const float* __restrict__ c, int n, int h, float* __restrict__ out) {
  const int lane = threadIdx.x & (kWave - 1);
  const int row = static_cast<int>(blockIdx.x) * kWavesPerBlock + static_cast<int>(threadIdx.x >> 6);
  if (row >= n) return;
  const int e0 = rowptr[row], e1 = rowptr[row + 1];
  float acc[kGatMaxHeads];
#pragma unroll
  for (int j = 0; j < kGatMaxHeads; ++j) acc[j] = 0.f;
  for (int e = e0 + lane; e < e1; e += kWave) {
    const size_t p = static_cast<size_t>(pos != nullptr ? pos[e] : e) * h;
#pragma unroll
    for (int j = 0; j < kGatMaxHeads; ++j)
      if (j < h) acc[j] += c[p + j];
  }
#pragma unroll
  for (int j = 0; j < kGatMaxHeads; ++j) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc[j] += __shfl_xor(acc[j], off, kWave);
  }
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < kGatMaxHeads; ++j)
      if (j < h) out[static_cast<size_t>(row) * h + j] = acc[j];
  }
}

// d wx [n, A] from the head sums
__global__ __launch_bounds__(kBlock) void gat_dwx_kernel(const float* __restrict__ dts, const float* __restrict__ dtd, const float* __restrict__ a,
                                                        long long n, int A, int h, float* __restrict__ dwx) {
  const int dk = A / h;
  const long long total = n * A;
  for (long long i = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; i < total; i += static_cast<long long>(gridDim.x) * kBlock) {
    const long long row = i / A;
    const int col = static_cast<int>(i - row * A);
    const int hh = col / dk, cc = col - hh * dk;
    dwx[i] = fmaf(dts[row * h + hh], a[cc], dtd[row * h + hh] * a[dk + cc]);
  }
}

// ---- exp kernel (reference src/function_transformer_attention.py:193-194): score_eh = ov^2 exp(-|q_ih - k_jh|^2 / (2 l^2)).  With
// c_eh = dL/d(score) score from the normaliser backward, S1_ih = sum_row c, S2_jh = sum_col c, T_ih = sum_row c k_jh, U_jh = sum_col c q_ih:
//     d q_ih = (T_ih - S1_ih q_ih) / l^2,   d k_jh = (U_jh - S2_jh k_jh) / l^2,
//     d ov = 2 sum_ih S1_ih / ov,   d l = sum_e c_e |q_i - k_j|^2 / l^3 = sum_ih (S1_ih |q_ih|^2 - 2 q_ih . T_ih + S2_ih |k_ih|^2) / l^3.
// In place on dqk = [T | U] (interleaved rows [n, 2A]); block b of the slab grid also leaves its share of (d ov, d l) in
// partial[b][slot], partial[b][slot + 1].  One thread per column of the 2A-wide row, rows of the slab in order.
__global__ __launch_bounds__(kBlock) void exp_node_bwd_kernel(const float* __restrict__ s1, const float* __restrict__ s2, const float* __restrict__ q,
                                                             const float* __restrict__ k, int ldq, float* __restrict__ dqk, int n, int A, int h,
                                                             const float* __restrict__ lengthscale, const float* __restrict__ output_var,
                                                             int rows_per_block, float* __restrict__ partial, int stride, int slot) {
  __shared__ float red[2][kWavesPerBlock];
  const int dk = A / h;
  const int col = threadIdx.x;                 // [0, 2A): q side then k side
  const int r0 = static_cast<int>(blockIdx.x) * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > n) r1 = n;
  const float l = *lengthscale, ov = *output_var;
  const float inv_l2 = 1.0f / (l * l);
  float acc_l = 0.f, acc_ov = 0.f;
  if (col < 2 * A) {
    const bool kside = col >= A;
    const int c = kside ? col - A : col;
    const int hh = c / dk;
    for (int i = r0; i < r1; ++i) {
      const float sv = (kside ? s2 : s1)[static_cast<size_t>(i) * h + hh];
      const float v = (kside ? k : q)[static_cast<size_t>(i) * ldq + c];
      const float t = dqk[static_cast<size_t>(i) * 2 * A + col];
      acc_l += kside ? sv * v * v : sv * v * v - 2.0f * v * t;
      if (!kside && c == hh * dk) acc_ov += sv;          // (one column per head carries S1_ih)
      dqk[static_cast<size_t>(i) * 2 * A + col] = (t - sv * v) * inv_l2;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { acc_l += __shfl_xor(acc_l, off, kWave); acc_ov += __shfl_xor(acc_ov, off, kWave); }
  if ((threadIdx.x & (kWave - 1)) == 0) { red[0][threadIdx.x >> 6] = acc_l; red[1][threadIdx.x >> 6] = acc_ov; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* out = partial + static_cast<size_t>(blockIdx.x) * stride + slot;
    out[0] = 2.0f * ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / ov;
    out[1] = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (l * l * l);
  }
}

// partial[b][slot + {c, d_k + c}] = this slab's share of d a; the rest of the M slots behind `slot` (where the Gram kernels left the
// column sums of d wx -- GAT's projection has no bias) is zeroed.  One thread per column of wx, rows of the slab in order.
__global__ __launch_bounds__(kBlock) void gat_da_partial_kernel(const float* __restrict__ dts, const float* __restrict__ dtd, const float* __restrict__ wx,
                                                               int n, int A, int h, int rows_per_block, float* __restrict__ partial, int stride, int slot) {
  __shared__ float ps[kBlock], pd[kBlock];
  const int dk = A / h;
  const int col = threadIdx.x;
  const int r0 = static_cast<int>(blockIdx.x) * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > n) r1 = n;
  float s1 = 0.f, s2 = 0.f;
  if (col < A) {
    const int hh = col / dk;
    for (int i = r0; i < r1; ++i) {
      const float v = wx[static_cast<size_t>(i) * A + col];
      s1 = fmaf(dts[static_cast<size_t>(i) * h + hh], v, s1);
      s2 = fmaf(dtd[static_cast<size_t>(i) * h + hh], v, s2);
    }
  }
  ps[threadIdx.x] = s1;
  pd[threadIdx.x] = s2;
  __syncthreads();
  float* out = partial + static_cast<size_t>(blockIdx.x) * stride + slot;
  if (col < A) {
    float v = 0.f;
    if (col < dk) {
      for (int hh = 0; hh < h; ++hh) v += ps[hh * dk + col];
    } else if (col < 2 * dk) {
      for (int hh = 0; hh < h; ++hh) v += pd[hh * dk + (col - dk)];
    }
    out[col] = v;
  }
}

}  // namespace
}  // namespace gnpde

using namespace gnpde;

struct gnpde_adjoint {
  gnpde_rhs_t rhs;
  gnpde_graph_t graph, graph_t;
  const int32_t* t_from_csr;
  const float* proj_wt;
  const float* w_t_fixed;
  int method;
  std::vector<float> dts;
  char* ws;
  size_t ws_bytes;
  // workspace regions
  size_t state_bytes;
  float *uy[2], *ua[2], *F[4], *V[3], *P, *qk, *dqk, *w, *w_t, *r, *ds, *partial, *one, *dots, *hub_ws, *qk_inv;
  float *gts = nullptr, *gtd = nullptr;      // GAT: [n, h] head sums of c over the rows / over the columns
  int extra = 0;             // gradient slots behind the Gram block and the bias sums: 2 for the exp kernel (d output_var, d lengthscale)
  int unit_heads = 0;        // 1: cosine_sim, 2: pearson -- scores are the scaled dot product of normalised (mean-centred) head vectors
  int n_dots;
  char *ws_att, *ws_attbwd, *ws_spmm, *ws_spmm_t;
  size_t att_bytes, attbwd_bytes, spmm_bytes, spmm_t_bytes;
  int M, stride;
  bool rows_bwd;             // one-pass row softmax backward (else the general normaliser backward)
  bool zeroed = false;
  hipStream_t cap_stream = nullptr;
  hipGraph_t graph_obj = nullptr;
  hipGraphExec_t exec = nullptr;
  float *cap_y = nullptr, *cap_a = nullptr, *cap_g = nullptr;
  int n_evals = 0;
  // recorded forward solve (gnpde_adjoint_set_tape): the stage inputs of the FORWARD solve in evaluation order; the run is then the
  // reverse sweep through those evaluations (what autograd does through torchdiffeq's fixed-grid loop when opt['adjoint'] is off)
  const float* tape = nullptr;
  const int32_t* csr_from_t = nullptr;   // [e] position in graph_t of the entry at CSR position q (inverse of t_from_csr), or null
  bool swapped = false;              // the sweep gathers cotangent rows only (see stage_combine_kernel)
  float* r_t = nullptr;              // [e] edge products in the transposed graph's order
  int dots_capacity = 0;
  const float* tape_rec = nullptr;   // one RhsRecord (q||k, head-mean weights) per forward evaluation, behind the stage inputs (rhs.h), or null
  float* r_acc = nullptr;    // [e] or null: sum over the evaluations of (b_j h) u_a[row] . u_y[col] in CSR order (GRAND-l weight gradients)
};

namespace {

bool rows_bwd_shape(const gnpde_attention_t& at) {
  if ((at.type != GNPDE_ATT_SCALED_DOT && at.type != GNPDE_ATT_COSINE && at.type != GNPDE_ATT_PEARSON) || at.norm_idx != 0 || at.square_plus) return false;
  const int h = at.heads, dk = at.att_dim / at.heads;
  return (h == 1 || h == 2 || h == 4 || h == 8) && (dk == 4 || dk == 8 || dk == 16);
}

int check_adjoint(const gnpde_rhs_t* rhs, const gnpde_graph_t* gt, int method) {
  int rc = check_rhs(rhs);
  if (rc) return rc;
  GNPDE_CHECK_ARG(gt != nullptr && gt->n == rhs->graph->n && gt->e == rhs->graph->e, GNPDE_EINVAL,
                  "adjoint: the transposed graph does not match the descriptor's graph");
  GNPDE_CHECK_ARG(method == GNPDE_METHOD_EULER || method == GNPDE_METHOD_RK4 || method == GNPDE_METHOD_MIDPOINT, GNPDE_EINVAL,
                  "adjoint: bad method %d", method);     // (midpoint: as the reverse sweep of a recorded solve only)
  GNPDE_CHECK_ARG(rhs->kind == GNPDE_RHS_LAPLACIAN || rhs->kind == GNPDE_RHS_TRANSFORMER || rhs->kind == GNPDE_RHS_GAT, GNPDE_ESHAPE,
                  "adjoint: GRAND-l, GRAND-nl and the GAT function");
  if (rhs->kind == GNPDE_RHS_GAT) {
    const gnpde_attention_t& at = rhs->att;
    GNPDE_CHECK_ARG(at.heads >= 2 && at.heads <= kGatMaxHeads && at.att_dim % at.heads == 0 && at.att_dim % 4 == 0 && at.att_dim <= kBlock &&
                    rhs->proj_m == at.att_dim && at.gat_a != nullptr, GNPDE_ESHAPE,
                    "adjoint (GAT): 2..8 heads, attention_dim a multiple of 4 and <= 256");
  }
  GNPDE_CHECK_ARG(rhs->ld % 4 == 0 && rhs->d <= 256 && (rhs->d % 4 == 0 || (rhs->flags & GNPDE_RHS_PADDED_ROWS)), GNPDE_ESHAPE,
                  "adjoint: state rows of up to 256 floats in 16-byte lanes (d %% 4 == 0 or padded rows)");
  GNPDE_CHECK_ARG(rhs->n_state_rows <= rhs->graph->n && rhs->proj_row_end == 0 && rhs->graph->row_begin == 0, GNPDE_ESHAPE,
                  "adjoint: whole-graph descriptors only");
  if (rhs->kind == GNPDE_RHS_TRANSFORMER) {
    const gnpde_attention_t& at = rhs->att;
    const int a4 = at.att_dim / 4;
    const bool unit = at.type == GNPDE_ATT_COSINE || at.type == GNPDE_ATT_PEARSON;
    const bool expk = at.type == GNPDE_ATT_EXP_KERNEL;
    GNPDE_CHECK_ARG(at.type == GNPDE_ATT_SCALED_DOT || unit || expk, GNPDE_ESHAPE, "adjoint: scaled-dot, cosine_sim, pearson and exp_kernel scores");
    GNPDE_CHECK_ARG(!expk || (at.output_var && at.lengthscale && 2 * at.att_dim <= kBlock && at.heads <= kGatMaxHeads), GNPDE_ESHAPE,
                    "adjoint (exp kernel): scalars missing, attention_dim > 128 or more than 8 heads");
    GNPDE_CHECK_ARG(!unit || (at.heads >= 1 && normalise_heads_bwd_supported(at.att_dim, at.heads)), GNPDE_ESHAPE,
                    "adjoint: cosine_sim / pearson need d_k in {4, 8, 16}");
    GNPDE_CHECK_ARG(at.att_dim % at.heads == 0 && (at.att_dim / at.heads) % 4 == 0 && a4 <= 64 && (a4 & (a4 - 1)) == 0, GNPDE_ESHAPE,
                    "adjoint: attention_dim / 4 must be a power of two <= 64 and d_k a multiple of 4");
  }
  return 0;
}

size_t adjoint_layout(const gnpde_rhs_t& r, const gnpde_graph_t& gt, int method, gnpde_adjoint* s) {
  const gnpde_graph_t& g = *r.graph;
  const size_t state = align_up(static_cast<size_t>(g.n) * r.ld * 4, 256);
  const bool nl = r.kind != GNPDE_RHS_LAPLACIAN;        // an attention per evaluation (GRAND-nl, GAT)
  const bool gat = r.kind == GNPDE_RHS_GAT;
  const int M = nl ? r.proj_m : 0;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes, 256); return o; };
  size_t o_uy[2], o_ua[2], o_F[4], o_V[3], o_P = 0, o_qk = 0, o_dqk = 0, o_w = 0, o_wt = 0, o_r = 0, o_ds = 0, o_hub = 0, o_inv = 0, o_dts = 0, o_dtd = 0;
  for (int i = 0; i < 2; ++i) { o_uy[i] = take(state); o_ua[i] = take(state); }
  const int nF = method == GNPDE_METHOD_RK4 ? 1 : 0, nV = method == GNPDE_METHOD_RK4 ? 1 : 0;     // (u4 of the state / of the adjoint)
  for (int i = 0; i < 4; ++i) o_F[i] = i < nF ? take(state) : 0;
  for (int i = 0; i < 3; ++i) o_V[i] = i < nV ? take(state) : 0;
  const size_t o_one = take(256);
  const size_t e4 = static_cast<size_t>(g.e > 0 ? g.e : 1) * 4;
  size_t att_b = 0, attbwd_b = 0;
  o_r = take(e4);
  const int n_dots = adjoint_rows_dot_slots(&g, r.d);
  const int n_dots_t = adjoint_rows_dot_slots(&gt, r.d);
  const int dots_cap = (n_dots > n_dots_t ? n_dots : n_dots_t) + kCombineBlocks;
  const size_t o_dots = take(static_cast<size_t>(dots_cap) * 8);
  const size_t o_rt = take(e4);
  if (nl) {
    o_P = take(state);
    o_qk = take(static_cast<size_t>(g.n) * M * 4);
    o_dqk = take(static_cast<size_t>(g.n) * M * 4);
    o_w = take(e4); o_wt = take(e4);
    o_inv = take(static_cast<size_t>(g.n) * 2 * r.att.heads * 4);
    o_ds = take(e4 * r.att.heads);
    const size_t hub_f = hub_bwd_workspace_floats(&g, r.att.heads, r.att.att_dim), hub_ft = hub_bwd_workspace_floats(&gt, r.att.heads, r.att.att_dim);
    o_hub = take((hub_f > hub_ft ? hub_f : hub_ft) * 4 + 256);
    att_b = attention_workspace_bytes(&g, r.att.heads, gat);
    attbwd_b = gnpde_attention_bwd_workspace_bytes(&g, &r.att);
    if (gat || r.att.type == GNPDE_ATT_EXP_KERNEL) { o_dts = take(static_cast<size_t>(g.n) * r.att.heads * 4); o_dtd = take(static_cast<size_t>(g.n) * r.att.heads * 4); }
  }
  const int extra = (r.kind == GNPDE_RHS_TRANSFORMER && r.att.type == GNPDE_ATT_EXP_KERNEL) ? 2 : 0;
  const size_t o_att = take(att_b), o_attbwd = take(attbwd_b);
  const size_t spmm_b = gnpde_spmm_workspace_bytes(&g, r.d), spmm_t_b = gnpde_spmm_workspace_bytes(&gt, r.d);
  const size_t o_spmm = take(spmm_b), o_spmm_t = take(spmm_t_b);
  const int stride = M * r.d + M + extra + 2;
  const size_t o_part = take(static_cast<size_t>(kParamBlocks) * stride * 4);
  if (s) {
    char* b = s->ws;
    auto f = [&](size_t o) { return reinterpret_cast<float*>(b + o); };
    for (int i = 0; i < 2; ++i) { s->uy[i] = f(o_uy[i]); s->ua[i] = f(o_ua[i]); }
    for (int i = 0; i < 4; ++i) s->F[i] = i < nF ? f(o_F[i]) : nullptr;
    for (int i = 0; i < 3; ++i) s->V[i] = i < nV ? f(o_V[i]) : nullptr;
    s->one = f(o_one);
    s->P = nl ? f(o_P) : nullptr; s->qk = nl ? f(o_qk) : nullptr; s->dqk = nl ? f(o_dqk) : nullptr;
    s->w = nl ? f(o_w) : nullptr; s->w_t = nl ? f(o_wt) : nullptr; s->r = f(o_r); s->ds = nl ? f(o_ds) : nullptr;
    s->dots = f(o_dots); s->n_dots = n_dots; s->dots_capacity = dots_cap;
    s->r_t = f(o_rt);
    s->hub_ws = nl ? f(o_hub) : nullptr;
    s->qk_inv = nl ? f(o_inv) : nullptr;
    s->gts = (gat || extra) ? f(o_dts) : nullptr; s->gtd = (gat || extra) ? f(o_dtd) : nullptr;
    s->extra = extra;
    s->ws_att = b + o_att; s->ws_attbwd = b + o_attbwd; s->ws_spmm = b + o_spmm; s->ws_spmm_t = b + o_spmm_t;
    s->att_bytes = att_b; s->attbwd_bytes = attbwd_b; s->spmm_bytes = spmm_b; s->spmm_t_bytes = spmm_t_b;
    s->partial = f(o_part);
    s->state_bytes = state;
    s->M = M; s->stride = stride;
  }
  return off;
}

// One stage: F = f(uy) with epilogue eF, V = (df/dy)^T ua with epilogue eV, parameter gradients accumulated with weight pcoef.
int enqueue_stage(gnpde_adjoint* s, const float* uy, const float* ua, float* Fout, gnpde_epilogue_t eF, float* Vout,
                  gnpde_epilogue_t eV, float pcoef, float* grads, hipStream_t st, const RhsRecord* recorded = nullptr) {
  const gnpde_rhs_t& r = s->rhs;
  const gnpde_graph_t* g = &s->graph;
  const gnpde_graph_t* gt = &s->graph_t;
  const int n = g->n, d = r.d, ld = r.ld;
  const bool padded = (r.flags & GNPDE_RHS_PADDED_ROWS) != 0 && ld % 4 == 0;
  const bool nl = r.kind != GNPDE_RHS_LAPLACIAN;
  const bool gat = r.kind == GNPDE_RHS_GAT;
  const float* w = r.w_csr;
  const float* wt = s->w_t_fixed;
  int rc;
  gnpde_attention_t at = r.att;
  const int A = at.att_dim, M = s->M;
  const float* qk = s->qk;
  const float* kk = nl && !gat ? s->qk + A : s->qk;      // the key side of q||k and the row stride of both (interleaved unless recorded as two tables)
  int ldq = M;
  const float* wfwd = s->w;
  if (nl && recorded != nullptr) {
    // the forward solve left this evaluation's q||k and weights on the tape: neither the projection nor the attention runs again
    qk = recorded->proj;
    wfwd = recorded->wmean;
    if (rhs_key_table(r, uy)) { kk = qk + static_cast<size_t>(n) * A; ldq = A; }     // (as enqueue_rhs wrote them)
    else { kk = qk + A; ldq = M; }
    at.q = qk; at.k = kk; at.ldqk = ldq;
    w = wfwd;
  } else if (nl) {
    rc = launch_linear_any(uy, n, d, ld, r.proj_w, M, d, r.proj_b, s->qk, M, st);
    if (rc) return rc;
    at.q = s->qk; at.k = gat ? s->qk : s->qk + A; at.ldqk = M;
    if (s->unit_heads) {
      // cosine_sim / pearson (reference src/function_transformer_attention.py:197-206) = the scaled dot product of unit (mean-centred) head
      // vectors, as in the forward solver (csrc/solver.hip enqueue_rhs): normalised in place, the scale of every vector kept for the backward
      rc = launch_normalise_heads(s->qk, n, M, A, at.heads, s->unit_heads == 2, st, s->qk_inv);
      if (rc) return rc;
      at.type = GNPDE_ATT_SCALED_DOT;
    }
    rc = launch_edge_attention(g, &at, s->w, nullptr, nullptr, s->ws_att, s->att_bytes, st, nullptr);
    if (rc) return rc;
    w = s->w;
  }
  const bool swapped = s->tape != nullptr && s->swapped;
  bool r_through_map = false;
  int n_pairs = s->n_dots;                     // (d1, d2) pairs the dots fold below has to sum
  if (swapped) {
    // Recorded solve, cotangent side only: the row kernel on the TRANSPOSED graph with the roles exchanged (gathered = u_a, own row = the
    // recorded u_y): S = alpha' (A^T u_a - u_a), r in the transposed order, <u_y, S> -- see stage_combine_kernel.
    const int n_dots_t = adjoint_rows_dot_slots(gt, d);
    if (nl) {
      // (the weights stay in the order the attention wrote them: the row kernel takes w[t_from_csr[p]] itself -- the 4-byte gathers of the
      //  26-us permutation pass now ride under the row gathers; gnpde_tune(15, 2): the separate pass, for A/B)
      const bool fused_w = g_tune[GNPDE_TUNE_SWEEP_UNSWAPPED] != 2;
      if (g->e > 0 && !fused_w) {
        hipLaunchKernelGGL(permute_f32_kernel, dim3((g->e + 8 * kBlock - 1) / (8 * kBlock)), dim3(kBlock), 0, st, w, s->t_from_csr, g->e, s->w_t);
        GNPDE_LAUNCH_CHECK();
      }
      gnpde_epilogue_t eS{};
      eS.stage = GNPDE_STAGE_LINCOMB; eS.alpha = r.alpha; eS.alpha_sigmoid = r.alpha_sigmoid; eS.out_k = s->uy[0];
      rc = launch_adjoint_rows(gt, fused_w ? w : s->w_t, ua, uy, d, ld, &eS, s->r_t, s->dots, s->ws_spmm_t, s->spmm_t_bytes, st, padded, false, 1.0f,
                               fused_w ? s->t_from_csr : nullptr);
      if (rc) return rc;
      // the edge products in the order the attention backward walks them: the row-softmax kernels read r_t through the position map
      // themselves (RRef, csrc/backward.hip); the general normaliser backward (columns / squareplus / GAT / exp kernel) takes a permuted copy
      r_through_map = fused_w && s->rows_bwd && !gat && at.type == GNPDE_ATT_SCALED_DOT;
      if (g->e > 0 && !r_through_map) {
        hipLaunchKernelGGL(permute_f32_kernel, dim3((g->e + 8 * kBlock - 1) / (8 * kBlock)), dim3(kBlock), 0, st, s->r_t, s->csr_from_t, g->e, s->r);
        GNPDE_LAUNCH_CHECK();
      }
      n_pairs = n_dots_t + kCombineBlocks;       // (the combine pass appends its <u_a, x0> pairs)
    } else {
      // GRAND-l: nothing stands between the aggregation and the stage algebra -- the whole stage is this launch (+ the source dot)
      gnpde_epilogue_t e2 = eV;
      e2.alpha = r.alpha; e2.beta = nullptr; e2.x0 = nullptr; e2.alpha_sigmoid = r.alpha_sigmoid; e2.out_k = Vout;
      rc = launch_adjoint_rows(gt, s->w_t_fixed, ua, uy, d, ld, &e2, s->r_acc != nullptr ? s->r_acc : s->r, s->dots, s->ws_spmm_t, s->spmm_t_bytes, st,
                               padded, s->r_acc != nullptr, pcoef);
      if (rc) return rc;
      n_pairs = n_dots_t;
      if (r.x0 != nullptr) {
        CombineArgs ca{};
        ca.S = ua; ca.P = nullptr; ca.ua = ua; ca.x0 = r.x0; ca.beta = r.beta; ca.alpha_sigmoid = r.alpha_sigmoid;
        ca.n4 = static_cast<long long>(n) * ld / 4;
        ca.ep.stage = -1;                          // (no stage output: the dot only)
        ca.pairs = s->dots + 2 * static_cast<size_t>(n_dots_t);
        hipLaunchKernelGGL(stage_combine_kernel, dim3(kCombineBlocks), dim3(kBlock), 0, st, ca);
        GNPDE_LAUNCH_CHECK();
        n_pairs += kCombineBlocks;
      }
    }
  } else {
    // F with the next stage input in its epilogue + r_e = ua[row] . uy[col] + the per-wave dots, one kernel over the gathered rows
    eF.alpha = r.alpha; eF.beta = r.beta; eF.x0 = r.x0; eF.alpha_sigmoid = r.alpha_sigmoid;
    eF.out_k = Fout;
    if (s->tape != nullptr && s->r_acc != nullptr && !nl)
      rc = launch_adjoint_rows(g, w, uy, ua, d, ld, &eF, s->r_acc, s->dots, s->ws_spmm, s->spmm_bytes, st, padded, true, pcoef);
    else
      rc = launch_adjoint_rows(g, w, uy, ua, d, ld, &eF, s->r, s->dots, s->ws_spmm, s->spmm_bytes, st, padded);
    if (rc) return rc;
  }
  const float* source = nullptr;
  const float* source_scale = nullptr;
  int rows_per_block = (n + kParamBlocks - 1) / kParamBlocks;      // slab grid of the parameter-gradient passes
  if (rows_per_block < 4) rows_per_block = 4;
  const int nb = (n + rows_per_block - 1) / rows_per_block;
  if (nl && !gat && at.type == GNPDE_ATT_EXP_KERNEL) {
    const int h = at.heads, dk = A / h;
    rc = launch_edge_attention_bwd(g, &at, s->r, nullptr, 1, r.alpha, r.alpha_sigmoid, s->ds, s->ws_attbwd, s->attbwd_bytes, st);
    if (rc) return rc;
    const unsigned gr = static_cast<unsigned>((n + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(gat_head_sums_kernel, dim3(gr), dim3(kBlock), 0, st, g->rowptr, static_cast<const int*>(nullptr), s->ds, n, h, s->gts);
    GNPDE_LAUNCH_CHECK();
    hipLaunchKernelGGL(gat_head_sums_kernel, dim3(gr), dim3(kBlock), 0, st, gt->rowptr, s->t_from_csr, s->ds, n, h, s->gtd);
    GNPDE_LAUNCH_CHECK();
    rc = gnpde_head_spmm(g, 0, s->ds, h, dk, kk, ldq, 1.0f, s->dqk, M, st);            // T = sum_row c k
    if (rc) return rc;
    rc = gnpde_head_spmm(g, 1, s->ds, h, dk, qk, ldq, 1.0f, s->dqk + A, M, st);        // U = sum_col c q
    if (rc) return rc;
    hipLaunchKernelGGL(exp_node_bwd_kernel, dim3(nb), dim3(kBlock), 0, st, s->gts, s->gtd, qk, kk, ldq, s->dqk, n, A, h, at.lengthscale, at.output_var,
                       rows_per_block, s->partial, s->stride, M * d + M);
    GNPDE_LAUNCH_CHECK();
    rc = launch_linear_any(s->dqk, n, M, M, s->proj_wt, d, M, nullptr, s->P, ld, st);
    if (rc) return rc;
    if (g->e > 0 && !swapped) {
      hipLaunchKernelGGL(permute_f32_kernel, dim3((g->e + 8 * kBlock - 1) / (8 * kBlock)), dim3(kBlock), 0, st, wfwd, s->t_from_csr, g->e, s->w_t);
      GNPDE_LAUNCH_CHECK();
    }
    wt = s->w_t;
    source = s->P;
    source_scale = s->one;
  } else if (gat) {
    const int h = at.heads;
    rc = launch_edge_attention_bwd(g, &at, s->r, nullptr, 2, r.alpha, r.alpha_sigmoid, s->ds, s->ws_attbwd, s->attbwd_bytes, st);
    if (rc) return rc;
    const unsigned gr = static_cast<unsigned>((n + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(gat_head_sums_kernel, dim3(gr), dim3(kBlock), 0, st, g->rowptr, static_cast<const int*>(nullptr), s->ds, n, h, s->gts);
    GNPDE_LAUNCH_CHECK();
    hipLaunchKernelGGL(gat_head_sums_kernel, dim3(gr), dim3(kBlock), 0, st, gt->rowptr, s->t_from_csr, s->ds, n, h, s->gtd);
    GNPDE_LAUNCH_CHECK();
    long long blocks = (static_cast<long long>(n) * M + kBlock - 1) / kBlock;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gat_dwx_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, st, s->gts, s->gtd, at.gat_a, static_cast<long long>(n), M, h, s->dqk);
    GNPDE_LAUNCH_CHECK();
    rc = launch_linear_any(s->dqk, n, M, M, s->proj_wt, d, M, nullptr, s->P, ld, st);
    if (rc) return rc;
    if (g->e > 0 && !swapped) {
      hipLaunchKernelGGL(permute_f32_kernel, dim3((g->e + 8 * kBlock - 1) / (8 * kBlock)), dim3(kBlock), 0, st, wfwd, s->t_from_csr, g->e, s->w_t);
      GNPDE_LAUNCH_CHECK();
    }
    wt = s->w_t;
    source = s->P;
    source_scale = s->one;
  } else if (nl) {
    const int h = at.heads, dk = A / h;
    const float inv = 1.0f / sqrtf(static_cast<float>(dk));
    const bool lanes = head_rowsum_supported(h, dk);            // lane-per-entry row sums (rows without entries stay zero)
    const bool dq_fused = s->rows_bwd && lanes && attention_rows_bwd_dq_supported(h, dk);
    if (lanes) GNPDE_HIP(hipMemsetAsync(s->dqk, 0, static_cast<size_t>(n) * M * 4, st));
    if (s->rows_bwd) rc = launch_attention_rows_bwd_dq(g, &at, r_through_map ? s->r_t : s->r, r.alpha, r.alpha_sigmoid, s->ds, dq_fused ? s->dqk : nullptr, M,
                                                       s->hub_ws, st, r_through_map ? s->csr_from_t : nullptr);
    else rc = gnpde_edge_attention_bwd(g, &at, s->r, r.alpha, r.alpha_sigmoid, s->ds, s->ws_attbwd, s->attbwd_bytes, st);
    if (rc) return rc;
    if (lanes) {
      // d q over the rows (unless the backward kernel formed it), d k over the rows of the transposed graph
      if (!dq_fused) {
        rc = launch_head_rowsum(g, nullptr, s->ds, h, dk, kk, ldq, inv, s->dqk, M, s->hub_ws, st);
        if (rc) return rc;
      }
      rc = launch_head_rowsum(gt, s->t_from_csr, s->ds, h, dk, qk, ldq, inv, s->dqk + A, M, s->hub_ws, st);
      if (rc) return rc;
    } else {
      rc = gnpde_head_spmm(g, 0, s->ds, h, dk, kk, ldq, inv, s->dqk, M, st);
      if (rc) return rc;
      rc = gnpde_head_spmm(g, 1, s->ds, h, dk, qk, ldq, inv, s->dqk + A, M, st);
      if (rc) return rc;
    }
    if (s->unit_heads) {         // through the normalisation: d (q||k) from the gradient of the unit vectors, in place
      rc = launch_normalise_heads_bwd(s->qk, s->dqk, n, M, A, h, s->unit_heads == 2, s->qk_inv, st);
      if (rc) return rc;
    }
    rc = launch_linear_any(s->dqk, n, M, M, s->proj_wt, d, M, nullptr, s->P, ld, st);
    if (rc) return rc;
    // (measured and not adopted, profiles/r04_train_v7_* / v9_*: scattering w_t from the row kernel and ds in the transposed order from the
    //  softmax backward -- 4- / 16-byte scattered stores cost more than these gathers, adjoint_rows +50 us, softmax backward +35 us against
    //  -25 / -30 us; and running this permutation + the NEXT stage's attention as a parallel hipGraph branch beside the backward chain --
    //  the overlapped kernels slow each other down by what the overlap hides, 0.82 ms per f + VJP either way)
    if (g->e > 0 && !swapped) {
      hipLaunchKernelGGL(permute_f32_kernel, dim3((g->e + 8 * kBlock - 1) / (8 * kBlock)), dim3(kBlock), 0, st, wfwd, s->t_from_csr, g->e, s->w_t);
      GNPDE_LAUNCH_CHECK();
    }
    wt = s->w_t;
    source = s->P;
    source_scale = s->one;
  }
  if (swapped && nl) {
    // V = S + P and the stage algebra of the cotangent recursion, one pass over the rows (+ the <u_a, x0> pairs)
    CombineArgs ca{};
    ca.S = s->uy[0]; ca.P = source; ca.ua = ua; ca.x0 = r.x0; ca.beta = r.beta; ca.alpha_sigmoid = r.alpha_sigmoid;
    ca.n4 = static_cast<long long>(n) * ld / 4;
    ca.ep = eV;
    ca.ep.out_k = Vout;
    ca.pairs = s->dots + 2 * static_cast<size_t>(n_pairs - kCombineBlocks);
    hipLaunchKernelGGL(stage_combine_kernel, dim3(kCombineBlocks), dim3(kBlock), 0, st, ca);
    GNPDE_LAUNCH_CHECK();
  } else if (!swapped) {
    // V = alpha (A^T ua - ua) [+ 1 * P]
    eV.alpha = r.alpha; eV.beta = source_scale; eV.x0 = source; eV.alpha_sigmoid = r.alpha_sigmoid;
    eV.out_k = Vout;
    rc = launch_spmm_rhs(gt, wt, ua, d, ld, &eV, nullptr, s->ws_spmm_t, s->spmm_t_bytes, st, nullptr, padded);
    if (rc) return rc;
  }
  // parameter gradients
  ParamArgs p{};
  p.dqk = nl ? s->dqk : nullptr; p.uy = uy; p.ua = ua; p.F = Fout; p.x0 = r.x0;
  p.n = n; p.d = d; p.ld = ld; p.M = M;
  p.rows_per_block = rows_per_block;
  p.stride = s->stride; p.partial = s->partial;
  const bool gram_mfma = nl && (M == 16 || M == 32 || M == 64) && d % 64 == 0 && d <= 256 && d * (M / 16) <= 512 && ld % 4 == 0 &&
                         reinterpret_cast<uintptr_t>(uy) % 16 == 0 && g_tune[GNPDE_TUNE_ADJOINT_GRAM] != 1;
  if (gram_mfma) {
#define GNPDE_GM(MVV, NGV) hipLaunchKernelGGL((adjoint_gram_mfma_kernel<MVV, NGV>), dim3(nb), dim3(kBlock), 0, st, p)
    const int mv = M / 16, ng = d / 64;
    if (mv == 1 && ng == 1) GNPDE_GM(1, 1); else if (mv == 1 && ng == 2) GNPDE_GM(1, 2); else if (mv == 1 && ng == 3) GNPDE_GM(1, 3);
    else if (mv == 1 && ng == 4) GNPDE_GM(1, 4); else if (mv == 2 && ng == 1) GNPDE_GM(2, 1); else if (mv == 2 && ng == 2) GNPDE_GM(2, 2);
    else if (mv == 2 && ng == 3) GNPDE_GM(2, 3); else if (mv == 2 && ng == 4) GNPDE_GM(2, 4); else if (mv == 4 && ng == 1) GNPDE_GM(4, 1);
    else GNPDE_GM(4, 2);
#undef GNPDE_GM
    GNPDE_LAUNCH_CHECK();
  } else if (nl) {
    const unsigned gy = static_cast<unsigned>((M + kGramTile - 1) / kGramTile);
    if (d <= 64) hipLaunchKernelGGL(adjoint_gram_kernel<1>, dim3(nb, gy), dim3(kBlock), 0, st, p);
    else if (d <= 128) hipLaunchKernelGGL(adjoint_gram_kernel<2>, dim3(nb, gy), dim3(kBlock), 0, st, p);
    else hipLaunchKernelGGL(adjoint_gram_kernel<4>, dim3(nb, gy), dim3(kBlock), 0, st, p);
    GNPDE_LAUNCH_CHECK();
  }
  if (gat) {       // d a into the (otherwise unused: no bias) M slots behind the Gram block
    hipLaunchKernelGGL(gat_da_partial_kernel, dim3(nb), dim3(kBlock), 0, st, s->gts, s->gtd, qk, n, M, at.heads, p.rows_per_block, s->partial, s->stride, M * d);
    GNPDE_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(adjoint_dots_fold_kernel, dim3(nb), dim3(kBlock), 0, st, s->dots, n_pairs, nb, s->partial, s->stride, M * d + M + s->extra);
  GNPDE_LAUNCH_CHECK();
  const int n_plain = M * d + M + s->extra;
  hipLaunchKernelGGL(adjoint_param_fold_kernel, dim3((n_plain + 1 + 31) / 32), dim3(kBlock), 0, st, s->partial, nb, s->stride, n_plain,
                     pcoef, r.alpha, r.beta, r.x0 != nullptr ? 1 : 0, r.alpha_sigmoid, swapped ? 0 : 1, grads);
  GNPDE_LAUNCH_CHECK();
  return 0;
}

int enqueue_adjoint(gnpde_adjoint* s, float* y, float* a, float* grads, hipStream_t st) {
  const gnpde_rhs_t& r = s->rhs;
  const size_t nbytes = static_cast<size_t>(r.graph->n) * r.ld * 4;
  GNPDE_HIP(hipMemsetAsync(grads, 0, static_cast<size_t>(s->stride) * 4, st));
  if (s->method == GNPDE_METHOD_EULER) {
    float* cy = y; float* ca = a;
    int flip = 0;
    for (float dt : s->dts) {
      gnpde_epilogue_t eF{}, eV{};
      eF.stage = GNPDE_STAGE_EULER; eF.dt = -dt; eF.y = cy; eF.out_y = s->uy[flip];       // y' = -F in s = -t
      eV.stage = GNPDE_STAGE_EULER; eV.dt = dt; eV.y = ca; eV.out_y = s->ua[flip];        // a' = +V
      int rc = enqueue_stage(s, cy, ca, nullptr, eF, nullptr, eV, dt, grads, st);
      if (rc) return rc;
      cy = s->uy[flip]; ca = s->ua[flip];
      flip ^= 1;
    }
    if (cy != y) {
      GNPDE_HIP(hipMemcpyAsync(y, cy, nbytes, hipMemcpyDeviceToDevice, st));
      GNPDE_HIP(hipMemcpyAsync(a, ca, nbytes, hipMemcpyDeviceToDevice, st));
    }
    return 0;
  }
  // rk4 (3/8 rule) in the COMPACT form of the forward solver (gnpde.h: stage states expressed through the previous stage inputs, so no
  // stage derivative F_j / V_j is ever stored or re-read -- the identities hold for any sequence of derivatives, here -F for the state
  // and +V for the adjoint): per stage each kernel reads one or two state-sized operands besides its own row instead of up to five.
  // u2 = uy[0] / ua[0], u3 = uy[1] / ua[1], u4 = F[0] / V[0] (free buffers).
  float* y4 = s->F[0]; float* a4 = s->V[0];
  for (float dtf : s->dts) {
    const double dt = dtf;
    const float c8 = static_cast<float>(dt * 0.125), c38 = static_cast<float>(3.0 * (dt * 0.125));
    gnpde_epilogue_t eF{}, eV{};
    eF.stage = GNPDE_STAGE_RK1C; eF.dt = -dtf; eF.out_y = s->uy[0];
    eV.stage = GNPDE_STAGE_RK1C; eV.dt = dtf; eV.out_y = s->ua[0];
    int rc = enqueue_stage(s, y, a, nullptr, eF, nullptr, eV, c8, grads, st);
    if (rc) return rc;
    eF = gnpde_epilogue_t{}; eV = gnpde_epilogue_t{};
    eF.stage = GNPDE_STAGE_RK2C; eF.dt = -dtf; eF.y = y; eF.out_y = s->uy[1];
    eV.stage = GNPDE_STAGE_RK2C; eV.dt = dtf; eV.y = a; eV.out_y = s->ua[1];
    rc = enqueue_stage(s, s->uy[0], s->ua[0], nullptr, eF, nullptr, eV, c38, grads, st);
    if (rc) return rc;
    eF = gnpde_epilogue_t{}; eV = gnpde_epilogue_t{};
    eF.stage = GNPDE_STAGE_RK3C; eF.dt = -dtf; eF.k1 = s->uy[0]; eF.out_y = y4;
    eV.stage = GNPDE_STAGE_RK3C; eV.dt = dtf; eV.k1 = s->ua[0]; eV.out_y = a4;
    rc = enqueue_stage(s, s->uy[1], s->ua[1], nullptr, eF, nullptr, eV, c38, grads, st);
    if (rc) return rc;
    eF = gnpde_epilogue_t{}; eV = gnpde_epilogue_t{};
    eF.stage = GNPDE_STAGE_RK4C; eF.dt = -dtf; eF.y = y; eF.k1 = s->uy[1]; eF.out_y = y;     // in place: y is not gathered in this stage
    eV.stage = GNPDE_STAGE_RK4C; eV.dt = dtf; eV.y = a; eV.k1 = s->ua[1]; eV.out_y = a;
    rc = enqueue_stage(s, y4, a4, nullptr, eF, nullptr, eV, c8, grads, st);
    if (rc) return rc;
  }
  return 0;
}

// Reverse sweep through a RECORDED fixed-grid solve (s->tape: the forward's stage inputs, csrc/solver.hip gnpde_solver_set_tape).
// With the cotangents of the stage derivatives normalised by their weights, c_j = (b_j h) chat_j, the sweep through one rk4 (3/8 rule)
// step reads
//     chat_4 = g,  chat_3 = g + (h/3) J_4^T chat_4,  chat_2 = 2 g - chat_3 + h J_3^T chat_3,  chat_1 = 2 chat_3 - chat_2 + h J_2^T chat_2,
//     g_new  = (6 chat_2 + 3 chat_1 - g + h J_1^T chat_1) / 8
// -- the compact stage formulas of the forward solver with (g, chat_3, chat_2, chat_1) in the places of (y, u2, u3, u4): the same V
// epilogues as the continuous adjoint above, the state-side operand read from the tape (u4, u3, u2, u1) instead of integrated
// backwards, parameter gradients weighted b_j h.  F is formed for the scalar gradients' dots only (no stage output).
int enqueue_taped(gnpde_adjoint* s, float* a, float* grads, hipStream_t st) {
  const gnpde_rhs_t& r = s->rhs;
  const size_t nbytes = static_cast<size_t>(r.graph->n) * r.ld * 4;
  const size_t stride = s->state_bytes / 4;
  auto slot = [&](size_t i) { return s->tape + i * stride; };
  GNPDE_HIP(hipMemsetAsync(grads, 0, static_cast<size_t>(s->stride) * 4, st));
  if (s->r_acc != nullptr && r.graph->e > 0) GNPDE_HIP(hipMemsetAsync(s->r_acc, 0, static_cast<size_t>(r.graph->e) * 4, st));
  const int S = static_cast<int>(s->dts.size());
  auto no_output = []() { gnpde_epilogue_t e{}; e.stage = GNPDE_STAGE_LINCOMB; return e; };
  RhsRecord rec_store{};
  auto rec = [&](size_t eval) -> const RhsRecord* {
    if (s->tape_rec == nullptr) return nullptr;
    rec_store = rhs_record_at(r, const_cast<float*>(s->tape_rec), eval);
    return &rec_store;
  };
  if (s->method == GNPDE_METHOD_EULER) {
    float* ca = a;
    int flip = 0;
    for (int n = S - 1; n >= 0; --n) {
      const float h = s->dts[n];
      gnpde_epilogue_t eV{};
      eV.stage = GNPDE_STAGE_EULER; eV.dt = h; eV.y = ca; eV.out_y = s->ua[flip];
      int rc = enqueue_stage(s, slot(n), ca, nullptr, no_output(), nullptr, eV, h, grads, st, rec(n));
      if (rc) return rc;
      ca = s->ua[flip];
      flip ^= 1;
    }
    if (ca != a) GNPDE_HIP(hipMemcpyAsync(a, ca, nbytes, hipMemcpyDeviceToDevice, st));
    return 0;
  }
  if (s->method == GNPDE_METHOD_MIDPOINT) {
    // y_mid = y + (h/2) f(y), y' = y + h f(y_mid):  W = J_mid^T g,  g_new = g + h W + (h^2 / 2) J_y^T W
    for (int n = S - 1; n >= 0; --n) {
      const float h = s->dts[n];
      gnpde_epilogue_t eV{};
      eV.stage = GNPDE_STAGE_LINCOMB; eV.y = a; eV.n_prev = 0; eV.coef[0] = h; eV.out_y = s->ua[1];
      int rc = enqueue_stage(s, slot(2 * static_cast<size_t>(n) + 1), a, nullptr, no_output(), s->ua[0], eV, h, grads, st, rec(2 * static_cast<size_t>(n) + 1));
      if (rc) return rc;
      const float hh = 0.5f * h * h;
      eV = gnpde_epilogue_t{};
      eV.stage = GNPDE_STAGE_LINCOMB; eV.y = s->ua[1]; eV.n_prev = 0; eV.coef[0] = hh; eV.out_y = a;
      rc = enqueue_stage(s, slot(2 * static_cast<size_t>(n)), s->ua[0], nullptr, no_output(), nullptr, eV, hh, grads, st, rec(2 * static_cast<size_t>(n)));
      if (rc) return rc;
    }
    return 0;
  }
  float* a4 = s->V[0];
  for (int n = S - 1; n >= 0; --n) {
    const float dtf = s->dts[n];
    const double dt = dtf;
    const float c8 = static_cast<float>(dt * 0.125), c38 = static_cast<float>(3.0 * (dt * 0.125));
    const float *u1 = slot(4 * static_cast<size_t>(n)), *u2 = u1 + stride, *u3 = u2 + stride, *u4 = u3 + stride;
    gnpde_epilogue_t eV{};
    eV.stage = GNPDE_STAGE_RK1C; eV.dt = dtf; eV.out_y = s->ua[0];
    const size_t e0 = 4 * static_cast<size_t>(n);
    int rc = enqueue_stage(s, u4, a, nullptr, no_output(), nullptr, eV, c8, grads, st, rec(e0 + 3));
    if (rc) return rc;
    eV = gnpde_epilogue_t{};
    eV.stage = GNPDE_STAGE_RK2C; eV.dt = dtf; eV.y = a; eV.out_y = s->ua[1];
    rc = enqueue_stage(s, u3, s->ua[0], nullptr, no_output(), nullptr, eV, c38, grads, st, rec(e0 + 2));
    if (rc) return rc;
    eV = gnpde_epilogue_t{};
    eV.stage = GNPDE_STAGE_RK3C; eV.dt = dtf; eV.k1 = s->ua[0]; eV.out_y = a4;
    rc = enqueue_stage(s, u2, s->ua[1], nullptr, no_output(), nullptr, eV, c38, grads, st, rec(e0 + 1));
    if (rc) return rc;
    eV = gnpde_epilogue_t{};
    eV.stage = GNPDE_STAGE_RK4C; eV.dt = dtf; eV.y = a; eV.k1 = s->ua[1]; eV.out_y = a;      // in place: a is not gathered in this stage
    rc = enqueue_stage(s, u1, a4, nullptr, no_output(), nullptr, eV, c8, grads, st, rec(e0));
    if (rc) return rc;
  }
  return 0;
}

void drop_adjoint_graph(gnpde_adjoint* s) {
  if (s->exec) { (void)hipGraphExecDestroy(s->exec); s->exec = nullptr; }
  if (s->graph_obj) { (void)hipGraphDestroy(s->graph_obj); s->graph_obj = nullptr; }
  s->cap_y = s->cap_a = s->cap_g = nullptr;
}

}  // namespace

extern "C" int gnpde_adjoint_grad_floats(const gnpde_rhs_t* rhs) {
  if (!rhs) return 0;
  const int M = rhs->kind != GNPDE_RHS_LAPLACIAN ? rhs->proj_m : 0;
  const int extra = (rhs->kind == GNPDE_RHS_TRANSFORMER && rhs->att.type == GNPDE_ATT_EXP_KERNEL) ? 2 : 0;
  return M * rhs->d + M + extra + 2;
}

extern "C" size_t gnpde_adjoint_workspace_bytes(const gnpde_rhs_t* rhs, const gnpde_graph_t* graph_t, int32_t method) {
  if (check_adjoint(rhs, graph_t, method)) return 0;
  return adjoint_layout(*rhs, *graph_t, method, nullptr);
}

extern "C" int gnpde_adjoint_create(gnpde_adjoint_t** out, const gnpde_rhs_t* rhs, const gnpde_graph_t* graph_t,
                                    const int32_t* t_from_csr, const float* proj_wt, const float* w_t_csr, int32_t method,
                                    const float* dts, int32_t n_steps, void* workspace, size_t workspace_bytes) {
  GNPDE_CHECK_ARG(out != nullptr, GNPDE_EINVAL, "adjoint_create: out is null");
  *out = nullptr;
  int rc = check_adjoint(rhs, graph_t, method);
  if (rc) return rc;
  GNPDE_CHECK_ARG(n_steps >= 0 && (dts || n_steps == 0), GNPDE_EINVAL, "adjoint_create: bad time grid");
  if (rhs->kind != GNPDE_RHS_LAPLACIAN)
    GNPDE_CHECK_ARG(t_from_csr != nullptr && proj_wt != nullptr && reinterpret_cast<uintptr_t>(proj_wt) % 16 == 0, GNPDE_EINVAL,
                    "adjoint_create: GRAND-nl needs the position map of the transposed graph and the transposed projection weights");
  else
    GNPDE_CHECK_ARG(w_t_csr != nullptr || rhs->graph->e == 0, GNPDE_EINVAL, "adjoint_create: GRAND-l needs the weights in the transposed graph's order");
  gnpde_adjoint* s = new gnpde_adjoint();
  s->rhs = *rhs;
  s->graph = *rhs->graph;
  s->rhs.graph = &s->graph;
  s->graph_t = *graph_t;
  if (s->rhs.att.graph_t != nullptr) s->rhs.att.graph_t = &s->graph_t;     // the same transposed graph, owned here
  s->t_from_csr = t_from_csr;
  s->proj_wt = proj_wt;
  s->w_t_fixed = w_t_csr;
  s->method = method;
  s->dts.assign(dts, dts + n_steps);
  s->rows_bwd = rhs->kind == GNPDE_RHS_TRANSFORMER && rows_bwd_shape(rhs->att) && rhs->proj_m % 4 == 0;
  if (rhs->kind == GNPDE_RHS_TRANSFORMER)
    s->unit_heads = rhs->att.type == GNPDE_ATT_COSINE ? 1 : rhs->att.type == GNPDE_ATT_PEARSON ? 2 : 0;
  const size_t need = adjoint_layout(s->rhs, s->graph_t, method, nullptr);
  if (!(workspace && workspace_bytes >= need && reinterpret_cast<uintptr_t>(workspace) % 256 == 0)) {
    set_error("adjoint_create: workspace %zu bytes (need %zu, 256-byte aligned)", workspace_bytes, need);
    delete s;
    return GNPDE_EWS;
  }
  s->ws = static_cast<char*>(workspace);
  s->ws_bytes = workspace_bytes;
  adjoint_layout(s->rhs, s->graph_t, method, s);
  s->n_evals = n_steps * (method == GNPDE_METHOD_RK4 ? 4 : method == GNPDE_METHOD_MIDPOINT ? 2 : 1);
  *out = s;
  return 0;
}

extern "C" int gnpde_adjoint_run(gnpde_adjoint_t* s, float* y, float* a, float* grads, int32_t use_graph, void* stream) {
  GNPDE_CHECK_ARG(s && y && a && grads && y != a, GNPDE_EINVAL, "adjoint_run: null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!s->zeroed) {
    // stage buffers start as zeros (the padding columns of padded rows are never written by the projection kernel) and the
    // unit scale of the V epilogue's source term is set once; ordered before everything else on the caller's stream
    const size_t head = static_cast<size_t>(reinterpret_cast<char*>(s->partial) - s->ws);
    GNPDE_HIP(hipMemsetAsync(s->ws, 0, head, st));
    const float one = 1.0f;
    GNPDE_HIP(hipMemcpyAsync(s->one, &one, sizeof(float), hipMemcpyHostToDevice, st));
    GNPDE_HIP(hipStreamSynchronize(st));   // (the host scalar above must not go out of scope before the copy ran)
    s->zeroed = true;
  }
  GNPDE_CHECK_ARG(s->tape != nullptr || s->method != GNPDE_METHOD_MIDPOINT, GNPDE_EINVAL,
                  "adjoint_run: midpoint runs as the reverse sweep of a recorded solve only (gnpde_adjoint_set_tape)");
  if (!use_graph) return s->tape ? enqueue_taped(s, a, grads, st) : enqueue_adjoint(s, y, a, grads, st);
  if (s->exec == nullptr || s->cap_y != y || s->cap_a != a || s->cap_g != grads) {
    drop_adjoint_graph(s);
    if (s->cap_stream == nullptr) GNPDE_HIP(hipStreamCreateWithFlags(&s->cap_stream, hipStreamNonBlocking));
    GNPDE_HIP(hipStreamBeginCapture(s->cap_stream, hipStreamCaptureModeThreadLocal));
    const int rc = s->tape ? enqueue_taped(s, a, grads, s->cap_stream) : enqueue_adjoint(s, y, a, grads, s->cap_stream);
    hipGraph_t gobj = nullptr;
    const hipError_t ec = hipStreamEndCapture(s->cap_stream, &gobj);
    if (rc != 0) {
      if (gobj) (void)hipGraphDestroy(gobj);
      return rc;
    }
    if (ec != hipSuccess) {
      set_error("adjoint_run: hipStreamEndCapture failed: %s", hipGetErrorString(ec));
      return static_cast<int>(ec);
    }
    s->graph_obj = gobj;
    GNPDE_HIP(hipGraphInstantiate(&s->exec, s->graph_obj, nullptr, nullptr, 0));
    s->cap_y = y; s->cap_a = a; s->cap_g = grads;
  }
  GNPDE_HIP(hipGraphLaunch(s->exec, st));
  return 0;
}

extern "C" int gnpde_adjoint_tape_swapped(const gnpde_adjoint_t* s) { return s != nullptr && s->tape != nullptr && s->swapped ? 1 : 0; }

extern "C" int gnpde_adjoint_set_tape(gnpde_adjoint_t* s, const void* tape, size_t tape_bytes, float* r_acc, const int32_t* csr_from_t) {
  GNPDE_CHECK_ARG(s != nullptr, GNPDE_EINVAL, "adjoint_set_tape: solver is null");
  drop_adjoint_graph(s);
  s->tape = nullptr;
  s->tape_rec = nullptr;
  s->r_acc = nullptr;
  s->csr_from_t = nullptr;
  s->swapped = false;
  if (tape == nullptr) return 0;
  const size_t per = s->method == GNPDE_METHOD_RK4 ? 4 : s->method == GNPDE_METHOD_MIDPOINT ? 2 : 1;
  const size_t rec_floats = rhs_record_stride(s->rhs);
  const size_t need = (per * s->dts.size() + 1) * s->state_bytes + per * s->dts.size() * rec_floats * 4;
  GNPDE_CHECK_ARG(reinterpret_cast<uintptr_t>(tape) % 256 == 0 && tape_bytes >= need, GNPDE_EWS,
                  "adjoint_set_tape: %zu bytes (need %zu, 256-byte aligned: the tape of gnpde_solver_set_tape for the same method and grid)",
                  tape_bytes, need);
  GNPDE_CHECK_ARG(r_acc == nullptr || s->rhs.kind == GNPDE_RHS_LAPLACIAN, GNPDE_EINVAL,
                  "adjoint_set_tape: edge-weight gradients are GRAND-l's (GRAND-nl forms its weights from the state)");
  s->csr_from_t = csr_from_t;
  // the cotangent-side form of the sweep: always for GRAND-l, for the functions with an attention per evaluation when the caller hands over
  // the inverse position map (gnpde_tune(15, 1): the first form, which gathers the recorded state rows again)
  s->swapped = g_tune[GNPDE_TUNE_SWEEP_UNSWAPPED] != 1 && (s->rhs.kind == GNPDE_RHS_LAPLACIAN || csr_from_t != nullptr);
  s->tape = static_cast<const float*>(tape);
  s->tape_rec = rec_floats > 0 ? s->tape + (per * s->dts.size() + 1) * (s->state_bytes / 4) : nullptr;
  s->r_acc = r_acc;
  return 0;
}

extern "C" int gnpde_adjoint_num_rhs_evals(const gnpde_adjoint_t* s) { return s ? s->n_evals : 0; }

extern "C" int gnpde_adjoint_destroy(gnpde_adjoint_t* s) {
  if (!s) return 0;
  drop_adjoint_graph(s);
  if (s->cap_stream) (void)hipStreamDestroy(s->cap_stream);
  delete s;
  return 0;
}
