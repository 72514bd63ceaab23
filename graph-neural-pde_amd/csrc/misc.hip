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

}  // namespace
}  // namespace gnpde

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
