// Issue rate of v_mfma_f32_16x16x4_f32 on MI355X as a function of how the accumulator chains are interleaved -- the q||k projection
// (csrc/linear.hip) spends 68 % of its wave cycles stalled at issue (profiles/r04_pmc_sq_forward.txt) with two chains alternating
// (A B A B ...).  Patterns over 64 MFMAs per iteration:  1 chain (A A A A ...), 2 chains alternating, 2 chains in runs of 4
// (A A A A B B B B), 4 chains alternating, 8 chains alternating.  One to four waves per SIMD.  No memory traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f32_probe.hip -o tools/probes/mfma_f32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, int RUN>
__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters, float a0, float b0) {
  f32x4 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      const int c = (m / RUN) % CHAINS;
      acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int c = 1; c < CHAINS; ++c) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int CHAINS, int RUN>
void run(const char* name, float* out) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 2000;
  for (int waves_per_simd = 1; waves_per_simd <= 4; ++waves_per_simd) {
    // 256 CUs x 4 SIMDs; a 256-thread workgroup puts one wave on every SIMD of a CU
    const int blocks = 256 * waves_per_simd;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL((mfma_kernel<CHAINS, RUN>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double mfma_per_simd = 64.0 * iters * waves_per_simd;
    const double ns_per_mfma = 1e6 * best / mfma_per_simd;
    const double tflops = 2048.0 * mfma_per_simd * 1024.0 / (best * 1e-3) / 1e12;
    printf("%-34s %d wave(s)/SIMD  %7.2f ns per MFMA per SIMD  (%5.1f cycles at 2.4 GHz)  %6.1f TFLOP/s chip-wide\n", name, waves_per_simd, ns_per_mfma,
           ns_per_mfma * 2.4, tflops);
  }
}

int main() {
  float* out; CHECK(hipMalloc(&out, 256 * 4 * 256 * 4 * sizeof(float)));
  run<1, 1>("1 chain  (A A A A)", out);
  run<2, 1>("2 chains (A B A B)", out);
  run<2, 4>("2 chains (A A A A B B B B)", out);
  run<4, 1>("4 chains (A B C D)", out);
  run<8, 1>("8 chains", out);
  return 0;
}
