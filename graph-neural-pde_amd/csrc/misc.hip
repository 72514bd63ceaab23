// Small data-movement kernels: row gather (halo packing for the multi-GPU exchange) and the
// stand-alone diffusion epilogue (GAT mix_features path, where the aggregate passes through Wout
// before alpha (ax - x) + beta x0 is applied; reference src/function_GAT_attention.py:33-38,56-64).
#include "common.h"

namespace gnpde {
namespace {

__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const float* __restrict__ src, int ld_src,
                                                            const int* __restrict__ idx, int count, int d,
                                                            float* __restrict__ dst, int ld_dst) {
  // one wavefront per row, lanes stride over the columns (coalesced on both sides)
  const int lane = threadIdx.x & (kWave - 1);
  const long long r = static_cast<long long>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  if (r >= count) return;
  const float* s = src + static_cast<size_t>(idx[r]) * ld_src;
  float* o = dst + static_cast<size_t>(r) * ld_dst;
  for (int c = lane; c < d; c += kWave) o[c] = s[c];
}

struct LinArgs {
  const float* base;
  const float* v[GNPDE_MAX_PREV];
  float c[GNPDE_MAX_PREV];
  int n_v;
  long long n;
  float* out;
  const float* scale;   // NULL or device scalar multiplied into every c[j] (fl32 product)
};

// out = base + sum_j c_j v_j over a flat array (stage algebra of host-driven Runge-Kutta loops in ONE pass)
template <bool VEC4>
__global__ __launch_bounds__(kBlock) void lincomb_kernel(LinArgs a) {
  // The argument struct is only READ (pointers and coefficients stay in the kernel-argument segment, fetched by scalar loads
  // with a wave-uniform index); scaling the coefficients in place would force the whole struct into scratch memory and every
  // v[j] / c[j] of the inner loops with it.  c[j] * s is the same fl32 product wherever it is formed.
  const float s = a.scale != nullptr ? *a.scale : 1.f;
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  if constexpr (VEC4) {
    const long long n4 = a.n / 4;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
      float4 r = reinterpret_cast<const float4*>(a.base)[i];
      for (int j = 0; j < a.n_v; ++j) {
        const float4 t = reinterpret_cast<const float4*>(a.v[j])[i];
        const float c = a.c[j] * s;
        r.x = fmaf(c, t.x, r.x); r.y = fmaf(c, t.y, r.y); r.z = fmaf(c, t.z, r.z); r.w = fmaf(c, t.w, r.w);
      }
      reinterpret_cast<float4*>(a.out)[i] = r;
    }
    for (long long i = n4 * 4 + static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < a.n; i += stride) {
      float r = a.base[i];
      for (int j = 0; j < a.n_v; ++j) r = fmaf(a.c[j] * s, a.v[j][i], r);
      a.out[i] = r;
    }
  } else {
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < a.n; i += stride) {
      float r = a.base[i];
      for (int j = 0; j < a.n_v; ++j) r = fmaf(a.c[j] * s, a.v[j][i], r);
      a.out[i] = r;
    }
  }
}

struct ErrArgs {
  const float* y0;
  const float* y1;
  const float* k[GNPDE_MAX_PREV];
  float coef[GNPDE_MAX_PREV];
  int n_k;
  float atol, rtol;
  long long n;
  int d, ld;
  const float* scale;   // NULL or device scalar multiplied into every coef[j] (fl32 product)
};

// block partial sums of (err / tol)^2 in a fixed order -> ws[blockIdx].  VEC4: rows with a stride that is a multiple of 4
// floats and 16-byte aligned operands are read as float4 (the padding columns [d, ld) are masked out).
template <bool VEC4>
// The squares are summed in DOUBLE (per lane, per wave, per block): the norm is then independent of the order of the rows to ~1e-16,
// far below the float32 the ratio is rounded to -- a solve on a relabelled graph (graph.LocalityView: same rows, permuted) takes the
// same accept / reject decisions and step sizes as on the graph as given.
__global__ __launch_bounds__(kBlock) void rk_error_partial_kernel(ErrArgs a, double* __restrict__ ws) {
  __shared__ double red[kWavesPerBlock];
  const float s = a.scale != nullptr ? *a.scale : 1.f;   // (applied at the use: the argument struct stays read-only, see lincomb_kernel)
  double acc = 0.0;
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  if constexpr (VEC4) {
    const int q = a.ld / 4;                       // float4 slots per row
    const long long total = a.n * q;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
      const long long r = i / q;
      const int c = static_cast<int>(i - r * q) * 4;
      if (c >= a.d) continue;
      float4 err = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int j = 0; j < a.n_k; ++j) {
        const float4 k = reinterpret_cast<const float4*>(a.k[j])[i];
        const float cj = a.coef[j] * s;
        err.x = fmaf(k.x, cj, err.x); err.y = fmaf(k.y, cj, err.y);
        err.z = fmaf(k.z, cj, err.z); err.w = fmaf(k.w, cj, err.w);
      }
      const float4 u = reinterpret_cast<const float4*>(a.y0)[i];
      const float4 v = reinterpret_cast<const float4*>(a.y1)[i];
      const float e[4] = {err.x, err.y, err.z, err.w};
      const float m[4] = {fmaxf(fabsf(u.x), fabsf(v.x)), fmaxf(fabsf(u.y), fabsf(v.y)), fmaxf(fabsf(u.z), fabsf(v.z)),
                          fmaxf(fabsf(u.w), fabsf(v.w))};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (c + t < a.d) {
          const float qv = e[t] / (a.atol + a.rtol * m[t]);
          acc += static_cast<double>(qv * qv);
        }
      }
    }
  } else {
    const long long total = a.n * a.d;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
      const long long r = i / a.d;
      const size_t off = static_cast<size_t>(r) * a.ld + static_cast<size_t>(i - r * a.d);
      float err = 0.f;
      for (int j = 0; j < a.n_k; ++j) err = fmaf(a.k[j][off], a.coef[j] * s, err);
      const float tol = a.atol + a.rtol * fmaxf(fabsf(a.y0[off]), fabsf(a.y1[off]));
      const float qv = err / tol;
      acc += static_cast<double>(qv * qv);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(kBlock) void rk_error_final_kernel(const double* __restrict__ ws, int nblocks, double count,
                                                               float* __restrict__ ratio) {
  __shared__ double red[kBlock];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += kBlock) acc += ws[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *ratio = static_cast<float>(sqrt(red[0] / count));
}

struct InterpArgs {
  const float* y0;
  const float* y1;
  const float* k[GNPDE_MAX_PREV];
  float mid[GNPDE_MAX_PREV];   // fl32(c_mid_j) * fl32(dt)
  float h, x;
  long long n;
  int d, ld;
  float* out;
};

// torchdiffeq's quartic end-point interpolation (_interp_fit + _interp_evaluate of the dopri5 solver) in one pass:
// y_mid = y0 + sum_j mid_j k_j;  a, b, c, d from (y0, y1, y_mid, f0 = k_0, f1 = k_6, h);  out = y0 + x d + x^2 c + x^3 b + x^4 a
__global__ __launch_bounds__(kBlock) void dopri5_interp_kernel(const InterpArgs a) {
  const long long total = a.n * a.d;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const long long r = i / a.d;
    const size_t off = static_cast<size_t>(r) * a.ld + static_cast<size_t>(i - r * a.d);
    const float ya = a.y0[off], yb = a.y1[off];
    float ym = ya;
#pragma unroll
    for (int j = 0; j < GNPDE_MAX_PREV; ++j)
      if (a.mid[j] != 0.0f) ym += a.k[j][off] * a.mid[j];
    const float fa = a.k[0][off], fb = a.k[6][off], h = a.h, x = a.x;
    const float ca = 2.0f * h * (fb - fa) - 8.0f * (yb + ya) + 16.0f * ym;
    const float cb = h * (5.0f * fa - 3.0f * fb) + 18.0f * ya + 14.0f * yb - 32.0f * ym;
    const float cc = h * (fb - 4.0f * fa) - 11.0f * ya - 5.0f * yb + 16.0f * ym;
    const float cd = h * fa;
    float tot = ya + x * cd;
    float xp = x * x;
    tot += xp * cc;
    xp *= x;
    tot += xp * cb;
    xp *= x;
    tot += xp * ca;
    a.out[off] = tot;
  }
}

}  // namespace

int launch_dopri5_interp(const float* y0, const float* y1, const float* const* k, const float* mid_coef, float h, float x,
                         int64_t n, int32_t d, int32_t ld, float* out, hipStream_t stream) {
  GNPDE_CHECK_ARG(y0 && y1 && k && mid_coef && out && n >= 1 && d >= 1 && ld >= d, GNPDE_EINVAL, "dopri5_interp: bad arguments");
  InterpArgs a{};
  a.y0 = y0; a.y1 = y1; a.h = h; a.x = x; a.n = n; a.d = d; a.ld = ld; a.out = out;
  for (int j = 0; j < GNPDE_MAX_PREV; ++j) {
    GNPDE_CHECK_ARG(k[j] != nullptr, GNPDE_EINVAL, "dopri5_interp: k[%d] is null", j);
    a.k[j] = k[j];
    a.mid[j] = mid_coef[j];
  }
  long long blocks = (n * d + kBlock - 1) / kBlock;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dopri5_interp_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, a);
  GNPDE_LAUNCH_CHECK();
  return 0;
}

int launch_rk_error_ratio(const float* y0, const float* y1, const float* const* k, const float* coef, int32_t n_k, float atol,
                          float rtol, int64_t n, int32_t d, int32_t ld, float* ratio, float* workspace, hipStream_t s,
                          const float* scale, int* partial_blocks) {
  GNPDE_CHECK_ARG(reinterpret_cast<uintptr_t>(workspace) % 8 == 0, GNPDE_EINVAL, "rk_error_ratio: workspace must be 8-byte aligned");
  GNPDE_CHECK_ARG(y0 && y1 && k && coef && (ratio || partial_blocks) && workspace && n_k >= 1 && n_k <= GNPDE_MAX_PREV && n >= 1 && d >= 1 && ld >= d,
                  GNPDE_EINVAL, "rk_error_ratio: bad arguments");
  ErrArgs a{};
  a.y0 = y0; a.y1 = y1; a.atol = atol; a.rtol = rtol; a.n = n; a.d = d; a.ld = ld; a.scale = scale;
  bool vec = ld % 4 == 0 && reinterpret_cast<uintptr_t>(y0) % 16 == 0 && reinterpret_cast<uintptr_t>(y1) % 16 == 0;
  for (int j = 0; j < n_k; ++j) {
    GNPDE_CHECK_ARG(k[j] != nullptr, GNPDE_EINVAL, "rk_error_ratio: k[%d] is null", j);
    if (coef[j] == 0.0f) continue;              // (dopri5: the second stage has no weight in the error estimate)
    a.k[a.n_k] = k[j];
    a.coef[a.n_k] = coef[j];
    a.n_k += 1;
    vec = vec && reinterpret_cast<uintptr_t>(k[j]) % 16 == 0;
  }
  const long long items = vec ? n * (ld / 4) : n * d;
  long long blocks = (items + kBlock - 1) / kBlock;
  if (blocks > 2048) blocks = 2048;
  if (vec) hipLaunchKernelGGL((rk_error_partial_kernel<true>), dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, s, a, reinterpret_cast<double*>(workspace));
  else hipLaunchKernelGGL((rk_error_partial_kernel<false>), dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, s, a, reinterpret_cast<double*>(workspace));
  GNPDE_LAUNCH_CHECK();
  if (partial_blocks != nullptr) {   // the caller folds workspace[0 .. blocks) itself (sqrt(sum / (n d)))
    *partial_blocks = static_cast<int>(blocks);
    return 0;
  }
  hipLaunchKernelGGL(rk_error_final_kernel, dim3(1), dim3(kBlock), 0, s, reinterpret_cast<const double*>(workspace), static_cast<int>(blocks),
                     static_cast<double>(n) * d, ratio);
  GNPDE_LAUNCH_CHECK();
  return 0;
}

}  // namespace gnpde

extern "C" int gnpde_dopri5_interp(const float* y0, const float* y1, const float* const* k, const float* mid_coef, float h,
                                   float x, int64_t n, int32_t d, int32_t ld, float* out, void* stream) {
  return gnpde::launch_dopri5_interp(y0, y1, k, mid_coef, h, x, n, d, ld, out, static_cast<hipStream_t>(stream));
}

extern "C" int gnpde_rk_error_ratio(const float* y0, const float* y1, const float* const* k, const float* coef, int32_t n_k,
                                    float atol, float rtol, int64_t n, int32_t d, int32_t ld, float* ratio, float* workspace,
                                    void* stream) {
  return gnpde::launch_rk_error_ratio(y0, y1, k, coef, n_k, atol, rtol, n, d, ld, ratio, workspace,
                                      static_cast<hipStream_t>(stream), nullptr, nullptr);
}

extern "C" int gnpde_gather_rows(const float* src, int32_t ld_src, const int32_t* idx, int32_t count, int32_t d,
                                 float* dst, int32_t ld_dst, void* stream) {
  GNPDE_CHECK_ARG(count >= 0 && d >= 1 && ld_src >= d && ld_dst >= d, GNPDE_EINVAL, "gather_rows: bad shape");
  if (count == 0) return 0;
  GNPDE_CHECK_ARG(src && idx && dst, GNPDE_EINVAL, "gather_rows: null pointer");
  hipLaunchKernelGGL(gnpde::gather_rows_kernel, dim3((count + gnpde::kWavesPerBlock - 1) / gnpde::kWavesPerBlock),
                     dim3(gnpde::kBlock), 0, static_cast<hipStream_t>(stream), src, ld_src, idx, count, d, dst, ld_dst);
  GNPDE_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Measurement aid (bench.py's roofline): what the memory system gives a perfectly BALANCED gather of whole state rows.
// out[i] = sum_{t < k} table[idx[i * k + t]]: every output row gathers exactly k rows of d floats by index, adds them (one VALU
// add per loaded value, nothing else: no weights, no epilogue operands, no stage algebra) and is stored once.  Same row width,
// same table and the launch geometry of the aggregation kernels -- LPR = 32 or 64 lanes of 16 bytes own an output row, one
// wavefront per workgroup, XCD x takes the x-th eighth of the rows -- with all k (<= 16 per batch) row loads of a lane in
// flight before the first add.  Uniform k: no degree skew, no hub rows, no tail.  The aggregation cannot gather faster than
// this; how close it comes is roofline.frac when the table is cache-resident and HBM's 8 TB/s is not the ceiling.
namespace gnpde {
// (variant 2, experiment: ids with the top bit set are fetched with the nontemporal hint -- rows referenced once should not
//  evict the often-referenced ones from the XCD's L2)
template <int LPR, bool SHUFFLE, bool HINT = false>
__global__ __launch_bounds__(kWave) void gather_ceiling_kernel(const float* __restrict__ table, int ld, int d,
                                                               const int* __restrict__ idx, int k, float* __restrict__ out,
                                                               int n_out) {
  constexpr int RPW = kWave / LPR;          // output rows per wavefront
  const int lane = threadIdx.x;
  const int sub = lane / LPR, cl = lane % LPR;
  const int col = cl * 4;
  const unsigned nb = gridDim.x;
  const unsigned item = xcd_swizzle(blockIdx.x, nb);
  const long long row = static_cast<long long>(item) * RPW + sub;
  if (row >= n_out || col >= d) return;
  const int* my = idx + row * k;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int t0 = 0; t0 < k; t0 += 16) {
    float4 v[16];
    // SHUFFLE: the 16 ids of the batch by ONE coalesced load per row group (lane cl < 16 holds id t0 + cl), handed out by
    // ds_bpermute; otherwise every lane of the group loads each id itself (same address: one request)
    int mine = 0;
    if constexpr (SHUFFLE) mine = (cl < 16 && t0 + cl < k) ? my[t0 + cl] : 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
      int c;
      if constexpr (SHUFFLE) c = __shfl(mine, sub * LPR + t, kWave); else c = t0 + t < k ? my[t0 + t] : 0;
      if constexpr (HINT) {
        if (t0 + t < k) {
          const float* src = table + static_cast<size_t>(c & 0x7fffffff) * ld + col;
          if (c < 0) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 q = __builtin_nontemporal_load(reinterpret_cast<const f4*>(src));
            v[t] = make_float4(q[0], q[1], q[2], q[3]);
          } else {
            v[t] = *reinterpret_cast<const float4*>(src);
          }
        }
      } else {
        if (t0 + t < k) v[t] = *reinterpret_cast<const float4*>(table + static_cast<size_t>(c) * ld + col);
      }
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      acc.x += v[t].x; acc.y += v[t].y; acc.z += v[t].z; acc.w += v[t].w;
    }
  }
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 o = {acc.x, acc.y, acc.z, acc.w};
  __builtin_nontemporal_store(o, reinterpret_cast<f4*>(out + static_cast<size_t>(row) * ld + col));
}
}  // namespace gnpde

extern "C" int gnpde_gather_ceiling(const float* table, int32_t n_rows, int32_t d, int32_t ld, const int32_t* idx, int32_t k,
                                    float* out, int32_t n_out, int32_t variant, void* stream) {
  GNPDE_CHECK_ARG(table && idx && out && n_rows >= 1 && n_out >= 1 && k >= 1, GNPDE_EINVAL, "gather_ceiling: bad arguments");
  GNPDE_CHECK_ARG(d >= 4 && d % 4 == 0 && d <= 256 && ld >= d && ld % 4 == 0 && reinterpret_cast<uintptr_t>(table) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(out) % 16 == 0, GNPDE_ESHAPE,
                  "gather_ceiling: rows of 4..256 floats in 16-byte lanes (d %% 4 == 0, aligned)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  GNPDE_CHECK_ARG(variant >= 0 && variant <= 2, GNPDE_EINVAL, "gather_ceiling: variant 0 (ids loaded per lane), 1 (coalesced + shuffle) or 2 (0 with the nontemporal hint on ids whose top bit is set)");
  if (variant == 2) {
    GNPDE_CHECK_ARG(d <= 128, GNPDE_ESHAPE, "gather_ceiling: variant 2 covers d <= 128");
    const unsigned grid = gnpde::xcd_grid((static_cast<long long>(n_out) + 1) / 2);
    hipLaunchKernelGGL((gnpde::gather_ceiling_kernel<32, false, true>), dim3(grid), dim3(gnpde::kWave), 0, s, table, ld, d, idx, k, out, n_out);
    GNPDE_LAUNCH_CHECK();
    return 0;
  }
  if (d <= 128) {
    const unsigned grid = gnpde::xcd_grid((static_cast<long long>(n_out) + 1) / 2);
    if (variant == 0) hipLaunchKernelGGL((gnpde::gather_ceiling_kernel<32, false>), dim3(grid), dim3(gnpde::kWave), 0, s, table, ld, d, idx, k, out, n_out);
    else hipLaunchKernelGGL((gnpde::gather_ceiling_kernel<32, true>), dim3(grid), dim3(gnpde::kWave), 0, s, table, ld, d, idx, k, out, n_out);
  } else {
    const unsigned grid = gnpde::xcd_grid(n_out);
    if (variant == 0) hipLaunchKernelGGL((gnpde::gather_ceiling_kernel<64, false>), dim3(grid), dim3(gnpde::kWave), 0, s, table, ld, d, idx, k, out, n_out);
    else hipLaunchKernelGGL((gnpde::gather_ceiling_kernel<64, true>), dim3(grid), dim3(gnpde::kWave), 0, s, table, ld, d, idx, k, out, n_out);
  }
  GNPDE_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Measurement aid (bench.py's roofline, `stream_read_probe`): a coalesced streaming READ of a table -- every 16-byte lane of
// the grid walks the table with a grid stride, eight loads in flight, one add per loaded value, one float per workgroup
// written.  On a table that exceeds the L2s (32 MiB) but fits the Infinity Cache this is the rate the memory side delivers
// L2-cold lines at: a hardware number for the cache-resident regime, not a gather.
namespace gnpde {
__global__ __launch_bounds__(kBlock) void stream_read_kernel(const float4* __restrict__ src, long long n4, int passes,
                                                             float* __restrict__ sink) {
  const long long stride = static_cast<long long>(gridDim.x) * kBlock;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p = 0; p < passes; ++p) {   // whole-table passes inside ONE launch: the ramp of a 15-us kernel is not in the rate
    long long i = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
      float4 v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = src[i + t * stride];
#pragma unroll
      for (int t = 0; t < 8; ++t) { acc.x += v[t].x; acc.y += v[t].y; acc.z += v[t].z; acc.w += v[t].w; }
    }
    for (; i < n4; i += stride) {
      const float4 v = src[i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    asm volatile("" ::: "memory");     // the next pass re-reads memory
  }
  float s = (acc.x + acc.y) + (acc.z + acc.w);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, kWave);
  __shared__ float part[kWavesPerBlock];
  if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kWavesPerBlock; ++w) t += part[w];
    sink[blockIdx.x] = t;
  }
}
}  // namespace gnpde

namespace gnpde {
namespace {
// q||k rows -> unit (optionally mean-centred) head vectors, in place; the query side also takes the factor sqrt(d_k) the scaled-dot
// kernels divide by.  One thread per (row, side, head).
__global__ __launch_bounds__(kBlock) void normalise_heads_kernel(float* __restrict__ qk, long long n, int ld, int att_dim, int heads,
                                                                int centre, float q_scale) {
  const long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= n * 2 * heads) return;
  const int head = static_cast<int>(idx % heads);
  const int side = static_cast<int>((idx / heads) % 2);
  const long long row = idx / (2 * heads);
  const int dk = att_dim / heads;
  float* v = qk + row * ld + side * att_dim + head * dk;
  float mean = 0.f;
  if (centre) {
    for (int j = 0; j < dk; ++j) mean += v[j];
    mean = mean / static_cast<float>(dk);
  }
  float nn = 0.f;
  for (int j = 0; j < dk; ++j) {
    const float c = v[j] - mean;
    nn = fmaf(c, c, nn);
  }
  // torch.nn.functional.cosine_similarity(eps = 1e-5) (reference src/function_transformer_attention.py:197-206) divides by
  // max(|x1|, eps) max(|x2|, eps) (torch >= 1.12; the fixtures were recorded with this image's torch): a clamp per vector, which is
  // exactly what a normalisation per vector can express
  const float inv = (side == 0 ? q_scale : 1.0f) / fmaxf(sqrtf(nn), 1e-5f);
  for (int j = 0; j < dk; ++j) v[j] = (v[j] - mean) * inv;
}
}  // namespace

namespace {
// the same with the head vector in registers (d_k = 4 DK4, 16-byte aligned rows): one 16-byte load and store per 4 columns
template <int DK4>
__global__ __launch_bounds__(kBlock) void normalise_heads_vec_kernel(float* __restrict__ qk, long long n, int ld, int att_dim, int heads,
                                                                    int centre, float q_scale, float* __restrict__ inv_out) {
  constexpr int DK = 4 * DK4;
  const long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= n * 2 * heads) return;
  const int head = static_cast<int>(idx % heads);
  const int side = static_cast<int>((idx / heads) % 2);
  const long long row = idx / (2 * heads);
  float4* v = reinterpret_cast<float4*>(qk + row * ld + side * att_dim + head * DK);
  float4 x[DK4];
#pragma unroll
  for (int j = 0; j < DK4; ++j) x[j] = v[j];
  float mean = 0.f;
  if (centre) {
#pragma unroll
    for (int j = 0; j < DK4; ++j) { mean += x[j].x; mean += x[j].y; mean += x[j].z; mean += x[j].w; }
    mean = mean / static_cast<float>(DK);
  }
  float nn = 0.f;
#pragma unroll
  for (int j = 0; j < DK4; ++j) {
    x[j].x -= mean; x[j].y -= mean; x[j].z -= mean; x[j].w -= mean;
    nn = fmaf(x[j].x, x[j].x, nn); nn = fmaf(x[j].y, x[j].y, nn); nn = fmaf(x[j].z, x[j].z, nn); nn = fmaf(x[j].w, x[j].w, nn);
  }
  const float inv = (side == 0 ? q_scale : 1.0f) / fmaxf(sqrtf(nn), 1e-5f);      // (as the scalar kernel above)
  if (inv_out != nullptr) inv_out[idx] = sqrtf(nn) >= 1e-5f ? inv : -inv;       // (negative: the clamp was active -- no projection term in the backward)
#pragma unroll
  for (int j = 0; j < DK4; ++j) v[j] = make_float4(x[j].x * inv, x[j].y * inv, x[j].z * inv, x[j].w * inv);
}

// Backward of the normalisation above, in place on the gradient g = dL/d(out) of the SAME table layout: with out = s c / max(|c|, eps),
// c = v - mean (centre) or v, s = sqrt(d_k) on the query side:
//     dL/dc = inv (g - out (out . g) / s^2)      (|c| >= eps;  inv g where the clamp was active),      dL/dv = dL/dc - mean(dL/dc) (centre)
// (torch autograd through F.cosine_similarity of the mean-centred head vectors, reference src/function_transformer_attention.py:197-206).
template <int DK4>
__global__ __launch_bounds__(kBlock) void normalise_heads_bwd_kernel(const float* __restrict__ out, float* __restrict__ g, long long n, int ld,
                                                                    int att_dim, int heads, int centre, float q_scale,
                                                                    const float* __restrict__ inv_in) {
  constexpr int DK = 4 * DK4;
  const long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= n * 2 * heads) return;
  const int head = static_cast<int>(idx % heads);
  const int side = static_cast<int>((idx / heads) % 2);
  const long long row = idx / (2 * heads);
  const size_t off = row * ld + side * att_dim + head * DK;
  const float4* o4 = reinterpret_cast<const float4*>(out + off);
  float4* g4 = reinterpret_cast<float4*>(g + off);
  float o[DK], gg[DK];
#pragma unroll
  for (int j = 0; j < DK4; ++j) {
    const float4 a = o4[j], b = g4[j];
    o[4 * j] = a.x; o[4 * j + 1] = a.y; o[4 * j + 2] = a.z; o[4 * j + 3] = a.w;
    gg[4 * j] = b.x; gg[4 * j + 1] = b.y; gg[4 * j + 2] = b.z; gg[4 * j + 3] = b.w;
  }
  const float inv_s = inv_in[idx];
  const float inv = fabsf(inv_s);
  const float s2 = side == 0 ? q_scale * q_scale : 1.0f;
  float dot = 0.f;
  if (inv_s > 0.f) {
#pragma unroll
    for (int j = 0; j < DK; ++j) dot = fmaf(o[j], gg[j], dot);
    dot = dot / s2;
  }
  float mean = 0.f;
#pragma unroll
  for (int j = 0; j < DK; ++j) {
    gg[j] = inv * (gg[j] - o[j] * dot);
    mean += gg[j];
  }
  mean = centre ? mean / static_cast<float>(DK) : 0.f;
#pragma unroll
  for (int j = 0; j < DK4; ++j) g4[j] = make_float4(gg[4 * j] - mean, gg[4 * j + 1] - mean, gg[4 * j + 2] - mean, gg[4 * j + 3] - mean);
}
}  // namespace

// cosine_sim / pearson scores as scaled-dot scores of normalised vectors (csrc/solver.hip enqueue_rhs): rows [0, n) of the q||k table
bool normalise_heads_bwd_supported(int att_dim, int heads) {
  const int dk = heads > 0 ? att_dim / heads : 0;
  return heads > 0 && att_dim % heads == 0 && (dk == 4 || dk == 8 || dk == 16);
}

// in place on g [n, ld] (the gradient of the normalised q||k table `out`); inv: what launch_normalise_heads recorded for the same rows
int launch_normalise_heads_bwd(const float* out, float* g, long long n, int ld, int att_dim, int heads, bool centre, const float* inv,
                               hipStream_t s) {
  if (n <= 0) return 0;
  GNPDE_CHECK_ARG(out && g && inv && normalise_heads_bwd_supported(att_dim, heads) && ld % 4 == 0 &&
                  reinterpret_cast<uintptr_t>(out) % 16 == 0 && reinterpret_cast<uintptr_t>(g) % 16 == 0, GNPDE_ESHAPE,
                  "normalise_heads_bwd: d_k in {4, 8, 16}, 16-byte aligned rows");
  const int dk = att_dim / heads;
  const long long items = n * 2 * heads;
  const dim3 grid(static_cast<unsigned>((items + kBlock - 1) / kBlock));
  const float qs = sqrtf(static_cast<float>(dk));
  if (dk == 4) hipLaunchKernelGGL(normalise_heads_bwd_kernel<1>, grid, dim3(kBlock), 0, s, out, g, n, ld, att_dim, heads, centre ? 1 : 0, qs, inv);
  else if (dk == 8) hipLaunchKernelGGL(normalise_heads_bwd_kernel<2>, grid, dim3(kBlock), 0, s, out, g, n, ld, att_dim, heads, centre ? 1 : 0, qs, inv);
  else hipLaunchKernelGGL(normalise_heads_bwd_kernel<4>, grid, dim3(kBlock), 0, s, out, g, n, ld, att_dim, heads, centre ? 1 : 0, qs, inv);
  GNPDE_LAUNCH_CHECK();
  return 0;
}

int launch_normalise_heads(float* qk, long long n, int ld, int att_dim, int heads, bool centre, hipStream_t s, float* inv_out) {
  if (n <= 0) return 0;
  const int dk = att_dim / heads;
  const long long items = n * 2 * heads;
  if (ld % 4 == 0 && reinterpret_cast<uintptr_t>(qk) % 16 == 0 && (dk == 4 || dk == 8 || dk == 16)) {
    const dim3 grid(static_cast<unsigned>((items + kBlock - 1) / kBlock));
    const float qs = sqrtf(static_cast<float>(dk));
    if (dk == 4) hipLaunchKernelGGL(normalise_heads_vec_kernel<1>, grid, dim3(kBlock), 0, s, qk, n, ld, att_dim, heads, centre ? 1 : 0, qs, inv_out);
    else if (dk == 8) hipLaunchKernelGGL(normalise_heads_vec_kernel<2>, grid, dim3(kBlock), 0, s, qk, n, ld, att_dim, heads, centre ? 1 : 0, qs, inv_out);
    else hipLaunchKernelGGL(normalise_heads_vec_kernel<4>, grid, dim3(kBlock), 0, s, qk, n, ld, att_dim, heads, centre ? 1 : 0, qs, inv_out);
    GNPDE_LAUNCH_CHECK();
    return 0;
  }
  GNPDE_CHECK_ARG(inv_out == nullptr, GNPDE_ESHAPE, "normalise_heads: the recorded scale needs d_k in {4, 8, 16} and 16-byte aligned rows");
  hipLaunchKernelGGL(normalise_heads_kernel, dim3(static_cast<unsigned>((items + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, qk, n, ld, att_dim,
                     heads, centre ? 1 : 0, sqrtf(static_cast<float>(dk)));
  GNPDE_LAUNCH_CHECK();
  return 0;
}
}  // namespace gnpde

extern "C" int gnpde_stream_read(const float* table, int64_t n_floats, int32_t passes, float* sink, int32_t n_sink, void* stream) {
  GNPDE_CHECK_ARG(table && sink && passes >= 1 && n_floats >= 4 && n_floats % 4 == 0 && reinterpret_cast<uintptr_t>(table) % 16 == 0, GNPDE_EINVAL,
                  "stream_read: a 16-byte aligned table of a multiple of 4 floats");
  long long blocks = (n_floats / 4 + gnpde::kBlock * 8 - 1) / (gnpde::kBlock * 8);
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (blocks < 1) blocks = 1;
  GNPDE_CHECK_ARG(n_sink >= blocks, GNPDE_EWS, "stream_read: sink of %d floats < %lld workgroups", n_sink, blocks);
  hipLaunchKernelGGL(gnpde::stream_read_kernel, dim3(static_cast<unsigned>(blocks)), dim3(gnpde::kBlock), 0,
                     static_cast<hipStream_t>(stream), reinterpret_cast<const float4*>(table), static_cast<long long>(n_floats / 4), passes, sink);
  GNPDE_LAUNCH_CHECK();
  return 0;
}

namespace gnpde {
int launch_lincomb(const float* base, const float* const* v, const float* coef, int32_t n_v, int64_t n, float* out,
                   hipStream_t s, const float* scale) {
  GNPDE_CHECK_ARG(base && out && n >= 0 && n_v >= 0 && n_v <= GNPDE_MAX_PREV && (n_v == 0 || (v && coef)), GNPDE_EINVAL,
                  "lincomb: bad arguments");
  if (n == 0) return 0;
  LinArgs a;
  a.base = base; a.n_v = n_v; a.n = n; a.out = out; a.scale = scale;
  bool al = reinterpret_cast<uintptr_t>(base) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0;
  for (int j = 0; j < GNPDE_MAX_PREV; ++j) {
    a.v[j] = j < n_v ? v[j] : nullptr;
    a.c[j] = j < n_v ? coef[j] : 0.f;
    if (j < n_v) {
      GNPDE_CHECK_ARG(v[j] != nullptr, GNPDE_EINVAL, "lincomb: vector %d is null", j);
      al = al && reinterpret_cast<uintptr_t>(v[j]) % 16 == 0;
    }
  }
  long long blocks = (n / 4 + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  if (al) hipLaunchKernelGGL((lincomb_kernel<true>), dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, s, a);
  else hipLaunchKernelGGL((lincomb_kernel<false>), dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, s, a);
  GNPDE_LAUNCH_CHECK();
  return 0;
}
}  // namespace gnpde

extern "C" int gnpde_lincomb(const float* base, const float* const* v, const float* coef, int32_t n_v, int64_t n, float* out,
                             void* stream) {
  return gnpde::launch_lincomb(base, v, coef, n_v, n, out, static_cast<hipStream_t>(stream), nullptr);
}
