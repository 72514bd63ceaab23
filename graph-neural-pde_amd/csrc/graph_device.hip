// The arrays of gnpde_graph_t built ON THE DEVICE from an edge list that already lives there: what csrc/graph_prep.cpp does on the host
// (and graph.build_arrays_on_device did with ~60 torch launches and half a dozen host reads per graph).  Blocks that hand over a NEW edge
// set every training forward -- hard attention, rewiring: reference src/block_transformer_hard_attention.py:55-61,
// src/block_transformer_rewiring.py -- pay this once per forward and once more for the transposed graph of the backward pass.
//
// Element for element the result of the other two builders (tests/test_kernels_gpu.py): both orderings are STABLE sorts of the caller's
// edge list (by row: the CSR positions, perm; by column of the CSR order: cscpos), the row records are listed by class (1..16 entries in
// row order, then 17..GNPDE_LONG_ROW entries longest first with ties in row order), long rows / columns in index order.
//   phase 1  gnpde_graph_build_device       everything whose size is known (e, n) + the counts (one host read by the caller)
//   phase 2  gnpde_graph_build_device_long  the lists of the long rows / columns and their 512-entry chunks
#include <cstring>
#include <algorithm>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include "common.h"

namespace gnpde {
namespace {

constexpr int kCountWords = 16;
enum { C_ERR = 0, C_N16, C_N64, C_NLE64, C_NLONGR, C_NLONGC, C_NCHUNKS, C_MAXROW, C_MAXCOL };
constexpr int kKeyNone = 1000;      // sort key of the rows that are not listed (no entries, or longer than GNPDE_LONG_ROW)

__global__ __launch_bounds__(kBlock) void gd_convert_kernel(const long long* __restrict__ row, const long long* __restrict__ col, long long e, int n,
                                                           int* __restrict__ row32, int* __restrict__ col32, int* __restrict__ iota,
                                                           int* __restrict__ counts) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= e) return;
  const long long r = row[i], c = col[i];
  const bool bad = r < 0 || r >= n || c < 0 || c >= n;
  if (bad) atomicOr(counts + C_ERR, 1);
  row32[i] = bad ? 0 : static_cast<int>(r);
  col32[i] = bad ? 0 : static_cast<int>(c);
  iota[i] = static_cast<int>(i);
}

// segment pointers from SORTED keys, no atomics (a hub row of 10^5 entries would serialise 10^5 atomic adds on one counter): position p
// writes ptr[r] = p for every r in (key[p - 1], key[p]] -- the first position of key[p] and of the empty segments in front of it
__global__ __launch_bounds__(kBlock) void gd_ptr_kernel(const int* __restrict__ sorted, long long e, int n, int* __restrict__ ptr) {
  const long long p = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p > e) return;
  const int prev = p > 0 ? sorted[p - 1] : -1;
  const int cur = p < e ? sorted[p] : n;
  for (int r = prev + 1; r <= cur; ++r) ptr[r] = static_cast<int>(p);
}

__global__ __launch_bounds__(kBlock) void gd_gather_kernel(const int* __restrict__ src, const int* __restrict__ idx, long long e, int* __restrict__ out) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < e) out[i] = src[idx[i]];
}

__global__ __launch_bounds__(kBlock) void gd_classify_kernel(const int* __restrict__ rowptr, const int* __restrict__ cscptr, int n, int* __restrict__ keys,
                                                            int* __restrict__ rows, int* __restrict__ counts) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  int n16 = 0, n64 = 0, nle = 0, nlr = 0, nlc = 0, nch = 0, mr = 0, mc = 0;
  if (r < n) {
    const int d = rowptr[r + 1] - rowptr[r], cd = cscptr[r + 1] - cscptr[r];
    int key = kKeyNone;
    if (d >= 1 && d <= 16) { key = 0; n16 = 1; }
    else if (d > 16 && d <= GNPDE_LONG_ROW) { key = 1 + (GNPDE_LONG_ROW - d); n64 = 1; nle = d <= 64 ? 1 : 0; }
    else if (d > GNPDE_LONG_ROW) { nlr = 1; nch = (d + GNPDE_LONG_ROW - 1) / GNPDE_LONG_ROW; }
    if (cd > GNPDE_LONG_ROW) nlc = 1;
    keys[r] = key;
    rows[r] = r;
    mr = d; mc = cd;
  }
  // wave-level sums / maxima first: one atomic per wave and counter
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    n16 += __shfl_xor(n16, o, kWave); n64 += __shfl_xor(n64, o, kWave); nle += __shfl_xor(nle, o, kWave);
    nlr += __shfl_xor(nlr, o, kWave); nlc += __shfl_xor(nlc, o, kWave); nch += __shfl_xor(nch, o, kWave);
    mr = max(mr, __shfl_xor(mr, o, kWave)); mc = max(mc, __shfl_xor(mc, o, kWave));
  }
  // ... then the block's four waves through the LDS: one atomic per BLOCK and counter (thousands of waves on eight addresses serialise)
  __shared__ int part[kWavesPerBlock][8];
  if ((threadIdx.x & (kWave - 1)) == 0) {
    int* p = part[threadIdx.x >> 6];
    p[0] = n16; p[1] = n64; p[2] = nle; p[3] = nlr; p[4] = nlc; p[5] = nch; p[6] = mr; p[7] = mc;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int c = threadIdx.x;
    int v = part[0][c];
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; ++w) v = c < 6 ? v + part[w][c] : max(v, part[w][c]);
    const int slot[8] = {C_N16, C_N64, C_NLE64, C_NLONGR, C_NLONGC, C_NCHUNKS, C_MAXROW, C_MAXCOL};
    if (v != 0) {
      if (c < 6) atomicAdd(counts + slot[c], v);
      else atomicMax(counts + slot[c], v);
    }
  }
}

__global__ __launch_bounds__(kBlock) void gd_bins_kernel(const int* __restrict__ keys_sorted, const int* __restrict__ listed, const int* __restrict__ rowptr,
                                                        int n, int* __restrict__ bin_rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 rec = make_int4(0, 0, 0, 0);
  if (keys_sorted[i] < kKeyNone) {
    const int r = listed[i];
    rec = make_int4(r, rowptr[r], rowptr[r + 1] - rowptr[r], 0);
  }
  reinterpret_cast<int4*>(bin_rows)[i] = rec;
}

struct LongPred {
  const int* ptr;
  __device__ bool operator()(int r) const { return ptr[r + 1] - ptr[r] > GNPDE_LONG_ROW; }
};

__global__ __launch_bounds__(kBlock) void gd_chunks_per_kernel(const int* __restrict__ long_rows, const int* __restrict__ rowptr, int n_long,
                                                              int* __restrict__ per) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_long) return;
  if (i == n_long) { per[i] = 0; return; }
  const int r = long_rows[i];
  per[i] = (rowptr[r + 1] - rowptr[r] + GNPDE_LONG_ROW - 1) / GNPDE_LONG_ROW;
}

__global__ __launch_bounds__(kWave) void gd_chunks_fill_kernel(const int* __restrict__ long_rows, const int* __restrict__ chunk_ptr, const int* __restrict__ rowptr,
                                                              int* __restrict__ chunk_row, int* __restrict__ chunk_begin, int* __restrict__ chunk_end,
                                                              int* __restrict__ chunk_first) {
  const int i = blockIdx.x;
  const int r = long_rows[i], c0 = chunk_ptr[i], c1 = chunk_ptr[i + 1];
  const int b = rowptr[r], e = rowptr[r + 1];
  for (int c = c0 + static_cast<int>(threadIdx.x); c < c1; c += kWave) {
    const int cb = b + (c - c0) * GNPDE_LONG_ROW;
    chunk_row[c] = r;
    chunk_begin[c] = cb;
    chunk_end[c] = cb + GNPDE_LONG_ROW < e ? cb + GNPDE_LONG_ROW : e;
    chunk_first[c] = c0;
  }
}

inline int bits_for(int n) {
  int b = 1;
  while (b < 31 && (1LL << b) < static_cast<long long>(n)) ++b;
  return b;
}

struct Phase1Layout {
  size_t row32, col32, iota, keys_tmp, deg, cdeg, rkeys, rkeys_out, rows, listed, temp, temp_bytes, total;
};

Phase1Layout phase1_layout(long long e, int n) {
  Phase1Layout L{};
  const size_t ee = static_cast<size_t>(e > 0 ? e : 1), nn = static_cast<size_t>(n > 0 ? n : 1);
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes, 256); return o; };
  L.row32 = take(ee * 4); L.col32 = take(ee * 4); L.iota = take(ee * 4); L.keys_tmp = take(ee * 4);
  L.deg = take((nn + 1) * 4); L.cdeg = take((nn + 1) * 4);
  L.rkeys = take(nn * 4); L.rkeys_out = take(nn * 4); L.rows = take(nn * 4); L.listed = take(nn * 4);
  size_t t1 = 0, t2 = 0, t3 = 0;
  int* p = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, t1, p, p, p, p, ee, 0, 32, nullptr);
  (void)rocprim::radix_sort_pairs(nullptr, t2, p, p, p, p, nn, 0, 32, nullptr);
  (void)rocprim::exclusive_scan(nullptr, t3, p, p, 0, nn + 1, rocprim::plus<int>(), nullptr);
  size_t t4 = 0;
  (void)rocprim::select(nullptr, t4, rocprim::counting_iterator<int>(0), p, p, nn, LongPred{nullptr}, nullptr);
  L.temp_bytes = align_up(std::max(std::max(t1, t2), std::max(t3, t4)) + 256, 256);
  L.temp = take(L.temp_bytes);
  L.total = off;
  return L;
}

}  // namespace
}  // namespace gnpde

extern "C" size_t gnpde_graph_build_device_workspace_bytes(int64_t e, int32_t n) {
  if (e < 0 || n < 0) return 0;
  return gnpde::phase1_layout(e, n).total;
}

extern "C" int gnpde_graph_build_device(const int64_t* row, const int64_t* col, int64_t e, int32_t n, int32_t* rowptr, int32_t* colidx,
                                        int32_t* perm, int32_t* rowidx, int32_t* cscptr, int32_t* cscpos, int32_t* bin_rows,
                                        int32_t* counts, void* workspace, size_t workspace_bytes, void* stream) {
  using namespace gnpde;
  GNPDE_CHECK_ARG(e >= 0 && n >= 0 && e < (1LL << 31) && rowptr && cscptr && bin_rows && counts && workspace && (e == 0 || (row && col && colidx && perm && rowidx && cscpos)),
                  GNPDE_EINVAL, "graph_build_device: bad arguments");
  const Phase1Layout L = phase1_layout(e, n);
  GNPDE_CHECK_ARG(workspace_bytes >= L.total && reinterpret_cast<uintptr_t>(workspace) % 256 == 0, GNPDE_EWS,
                  "graph_build_device: workspace %zu bytes (need %zu, 256-byte aligned)", workspace_bytes, L.total);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  int* row32 = reinterpret_cast<int*>(ws + L.row32); int* col32 = reinterpret_cast<int*>(ws + L.col32);
  int* iota = reinterpret_cast<int*>(ws + L.iota); int* keys_tmp = reinterpret_cast<int*>(ws + L.keys_tmp);
  int* rkeys = reinterpret_cast<int*>(ws + L.rkeys); int* rkeys_out = reinterpret_cast<int*>(ws + L.rkeys_out);
  int* rows = reinterpret_cast<int*>(ws + L.rows); int* listed = reinterpret_cast<int*>(ws + L.listed);
  void* temp = ws + L.temp;
  size_t tb = L.temp_bytes;
  GNPDE_HIP(hipMemsetAsync(counts, 0, kCountWords * 4, st));
  auto grid = [](long long items) { return dim3(static_cast<unsigned>(items > 0 ? (items + kBlock - 1) / kBlock : 1)); };
  if (e > 0) {
    hipLaunchKernelGGL(gd_convert_kernel, grid(e), dim3(kBlock), 0, st, reinterpret_cast<const long long*>(row), reinterpret_cast<const long long*>(col),
                       static_cast<long long>(e), n, row32, col32, iota, counts);
    GNPDE_LAUNCH_CHECK();
    const int bits = bits_for(n);
    // CSR order: the stable sort of the edge list by row (radix sort is stable); the row pointers from the sorted rows
    tb = L.temp_bytes;
    GNPDE_HIP(rocprim::radix_sort_pairs(temp, tb, row32, rowidx, iota, perm, static_cast<size_t>(e), 0, bits, st));
    hipLaunchKernelGGL(gd_ptr_kernel, grid(e + 1), dim3(kBlock), 0, st, rowidx, static_cast<long long>(e), n, rowptr);
    GNPDE_LAUNCH_CHECK();
    hipLaunchKernelGGL(gd_gather_kernel, grid(e), dim3(kBlock), 0, st, col32, perm, static_cast<long long>(e), colidx);
    GNPDE_LAUNCH_CHECK();
    // CSC positions: the stable sort of the CSR positions by column; the column pointers from the sorted columns
    tb = L.temp_bytes;
    GNPDE_HIP(rocprim::radix_sort_pairs(temp, tb, colidx, keys_tmp, iota, cscpos, static_cast<size_t>(e), 0, bits, st));
    hipLaunchKernelGGL(gd_ptr_kernel, grid(e + 1), dim3(kBlock), 0, st, keys_tmp, static_cast<long long>(e), n, cscptr);
    GNPDE_LAUNCH_CHECK();
  } else {
    GNPDE_HIP(hipMemsetAsync(rowptr, 0, (static_cast<size_t>(n) + 1) * 4, st));
    GNPDE_HIP(hipMemsetAsync(cscptr, 0, (static_cast<size_t>(n) + 1) * 4, st));
  }
  if (n > 0) {
    hipLaunchKernelGGL(gd_classify_kernel, grid(n), dim3(kBlock), 0, st, rowptr, cscptr, n, rkeys, rows, counts);
    GNPDE_LAUNCH_CHECK();
    tb = L.temp_bytes;
    GNPDE_HIP(rocprim::radix_sort_pairs(temp, tb, rkeys, rkeys_out, rows, listed, static_cast<size_t>(n), 0, 10, st));
    hipLaunchKernelGGL(gd_bins_kernel, grid(n), dim3(kBlock), 0, st, rkeys_out, listed, rowptr, n, bin_rows);
    GNPDE_LAUNCH_CHECK();
  } else {
    GNPDE_HIP(hipMemsetAsync(bin_rows, 0, 16, st));
  }
  return 0;
}

extern "C" int gnpde_graph_build_device_long(const int32_t* rowptr, const int32_t* cscptr, int32_t n, int32_t n_long_rows, int32_t n_long_chunks,
                                             int32_t n_long_cols, int32_t* long_rows, int32_t* long_chunk_ptr, int32_t* long_chunk_row,
                                             int32_t* long_chunk_begin, int32_t* long_chunk_end, int32_t* long_cols, int32_t* long_chunk_first,
                                             int32_t* scratch_count, void* workspace, size_t workspace_bytes, void* stream) {
  using namespace gnpde;
  GNPDE_CHECK_ARG(rowptr && cscptr && n >= 0 && n_long_rows >= 0 && n_long_chunks >= 0 && n_long_cols >= 0 && long_chunk_ptr && scratch_count && workspace,
                  GNPDE_EINVAL, "graph_build_device_long: bad arguments");
  const Phase1Layout L = phase1_layout(0, n);
  GNPDE_CHECK_ARG(workspace_bytes >= L.total, GNPDE_EWS, "graph_build_device_long: workspace %zu bytes (need %zu)", workspace_bytes, L.total);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  void* temp = ws + L.temp;
  int* per = reinterpret_cast<int*>(ws + L.deg);        // [n_long_rows + 1] chunk counts (n_long_rows <= n: the region holds n + 1)
  size_t tb = L.temp_bytes;
  if (n_long_rows > 0) {
    GNPDE_CHECK_ARG(long_rows && long_chunk_row && long_chunk_begin && long_chunk_end && long_chunk_first, GNPDE_EINVAL, "graph_build_device_long: null list");
    GNPDE_HIP(rocprim::select(temp, tb, rocprim::counting_iterator<int>(0), long_rows, scratch_count, static_cast<size_t>(n), LongPred{rowptr}, st));
    hipLaunchKernelGGL(gd_chunks_per_kernel, dim3((n_long_rows + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, st, long_rows, rowptr, n_long_rows, per);
    GNPDE_LAUNCH_CHECK();
    tb = L.temp_bytes;
    GNPDE_HIP(rocprim::exclusive_scan(temp, tb, per, long_chunk_ptr, 0, static_cast<size_t>(n_long_rows) + 1, rocprim::plus<int>(), st));
    hipLaunchKernelGGL(gd_chunks_fill_kernel, dim3(n_long_rows), dim3(kWave), 0, st, long_rows, long_chunk_ptr, rowptr, long_chunk_row, long_chunk_begin,
                       long_chunk_end, long_chunk_first);
    GNPDE_LAUNCH_CHECK();
  } else {
    GNPDE_HIP(hipMemsetAsync(long_chunk_ptr, 0, 4, st));
  }
  if (n_long_cols > 0) {
    GNPDE_CHECK_ARG(long_cols != nullptr, GNPDE_EINVAL, "graph_build_device_long: long_cols is null");
    tb = L.temp_bytes;
    GNPDE_HIP(rocprim::select(temp, tb, rocprim::counting_iterator<int>(0), long_cols, scratch_count, static_cast<size_t>(n), LongPred{cscptr}, st));
  }
  (void)n_long_chunks;
  return 0;
}
