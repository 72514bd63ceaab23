// Cost of a grid-wide barrier on MI355X, for the small-graph (Cora) regime: would ONE resident kernel per rk4 step with barriers
// between its four evaluations beat four dependent launches inside a hipGraph (4.9 us per evaluation today)?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/barrier_probe.hip -o tools/probes/barrier_probe && tools/probes/barrier_probe
// Variants: participants = every CU (256 workgroups) or the CUs of ONE XCD (the 32 workgroups with blockIdx % 8 == 0: workgroups are
// dealt to the XCDs round robin); ordering = relaxed agent-scope atomics only (the barrier's own latency: an atomic add + a polled
// load, both served by the L2 / the fabric) or release / acquire at agent scope (what data exchanged across XCDs needs: the release
// writes back the XCD's dirty L2 lines, the acquire invalidates).  Spins are bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool FENCED>
__global__ __launch_bounds__(256) void barrier_kernel(unsigned* counter, int xcd_only, unsigned participants, int iters, float* data,
                                                      int payload_floats, int* timed_out) {
  if (xcd_only && (blockIdx.x & 7) != 0) return;
  const unsigned me = xcd_only ? blockIdx.x >> 3 : blockIdx.x;
  unsigned target = 0;
  for (int it = 0; it < iters; ++it) {
    // a little work whose result the other workgroups read after the barrier (payload_floats per workgroup)
    for (int i = threadIdx.x; i < payload_floats; i += blockDim.x) data[static_cast<size_t>(me) * payload_floats + i] += 1.0f;
    __syncthreads();
    target += participants;
    if (threadIdx.x == 0) {
      if (FENCED) __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);
      else __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long long spins = 0;
      for (;;) {
        const unsigned v = FENCED ? __atomic_load_n(counter, __ATOMIC_ACQUIRE) : __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= target) break;
        if (++spins > (1ll << 24)) { *timed_out = 1; break; }
      }
    }
    __syncthreads();
  }
}

int main() {
  unsigned* counter; int* timed_out; float* data;
  const int payloads[3] = {0, 1024, 16384};
  CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&timed_out, 4)); CHECK(hipMalloc(&data, 256ull * 16384 * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 2000;
  for (int payload : payloads)
    for (int xcd_only = 0; xcd_only < 2; ++xcd_only)
      for (int fenced = 0; fenced < 2; ++fenced) {
        float best = 1e30f;
        int to = 0;
        for (int rep = 0; rep < 3; ++rep) {
          CHECK(hipMemset(counter, 0, 4)); CHECK(hipMemset(timed_out, 0, 4)); CHECK(hipMemset(data, 0, 256ull * 16384 * 4));
          CHECK(hipDeviceSynchronize());
          CHECK(hipEventRecord(e0));
          const unsigned participants = xcd_only ? 32 : 256;
          if (fenced) hipLaunchKernelGGL(barrier_kernel<true>, dim3(256), dim3(256), 0, 0, counter, xcd_only, participants, iters, data, payload, timed_out);
          else hipLaunchKernelGGL(barrier_kernel<false>, dim3(256), dim3(256), 0, 0, counter, xcd_only, participants, iters, data, payload, timed_out);
          CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
          float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
          CHECK(hipMemcpy(&to, timed_out, 4, hipMemcpyDeviceToHost));
        }
        printf("payload %6d floats/workgroup  %-22s %-28s %7.2f us per iteration%s\n", payload, xcd_only ? "one XCD (32 workgroups)" : "256 workgroups",
               fenced ? "release/acquire (agent)" : "relaxed atomics only", 1e3f * best / iters, to ? "  [SPIN LIMIT HIT]" : "");
      }
  return 0;
}
