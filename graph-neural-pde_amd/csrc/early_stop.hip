// Early-stopping evaluator of the test-time integrators, on device.
//
// The reference (src/early_stop_solver.py:163-218 for rk4, :30-128 for dopri5) leaves the solver after EVERY
// step to run  relu -> F.linear(m2) -> argmax -> masked accuracy  with three `.item()` host reads, and keeps the
// (train, val, test, time) of the step with the best validation accuracy.  Here that is two launches appended to
// the step inside the solver's hipGraph and no host traffic at all:
//
//   decode_count_kernel   logits = relu(y[:, :d_dec]) W^T + b on the fp32 matrix cores (16 nodes x 16 classes per
//                         v_mfma_f32_16x16x4_f32 tile, same operand scheme as linear.hip), arg-max over the
//                         classes (first maximum wins, like torch.max), compared with the label and counted per
//                         split with integer atomics (order independent, so deterministic);
//   early_stop_update     one thread: strict "val > best_val" (early_stop_solver.py:156-157 / :79-80) on the
//                         integer counts -- the denominators are equal, so this is the reference's float
//                         comparison without the division -- then clears the running counters.
//
// The ogbn-arxiv branch of the reference applies log_softmax before the arg-max (:192-193); it does not change
// the arg-max and the loss computed next to it is discarded (:196-198), so neither is evaluated here.
#include "common.h"

namespace gnpde {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct DecArgs {
  const float* __restrict__ y;
  const float* __restrict__ weight;
  const float* __restrict__ bias;
  const int* __restrict__ labels;
  const unsigned char* __restrict__ split;
  int* __restrict__ state;
  int n, ld, d, c;
  int d16, dpad;
  bool vec;
};

template <int CT>
__global__ __launch_bounds__(kBlock) void decode_count_kernel(const DecArgs a) {
  extern __shared__ float w_lds[];   // [16 * CT][dpad], zero padded in both directions
  for (int idx = threadIdx.x; idx < 16 * CT * a.dpad; idx += kBlock) {
    const int row = idx / a.dpad, k = idx - row * a.dpad;
    w_lds[idx] = (row < a.c && k < a.d) ? a.weight[static_cast<size_t>(row) * a.d + k] : 0.f;
  }
  __syncthreads();

  const int lane = threadIdx.x & (kWave - 1);
  const int j = lane & 15, kq = lane >> 4;
  const int n_tiles = (a.n + 15) / 16;
  int hit_train = 0, hit_val = 0, hit_test = 0;

  for (int tile = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); tile < n_tiles; tile += gridDim.x * kWavesPerBlock) {
    const int node = tile * 16 + j;                  // A operand row of this lane
    const bool live = node < a.n;
    const float* yrow = a.y + static_cast<size_t>(live ? node : 0) * a.ld;
    f32x4 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < a.d16; kb += 16) {
      const int k = kb + 4 * kq;
      float av[4];
      if (a.vec && k + 3 < a.d) {
        const float4 v = *reinterpret_cast<const float4*>(yrow + k);
        av[0] = v.x; av[1] = v.y; av[2] = v.z; av[3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = (k + i < a.d) ? yrow[k + i] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = live ? fmaxf(av[i], 0.f) : 0.f;     // F.relu
      float4 bv[CT];
#pragma unroll
      for (int t = 0; t < CT; ++t) bv[t] = *reinterpret_cast<const float4*>(w_lds + (t * 16 + j) * a.dpad + k);
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[t].x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[t].y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[t].z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[t].w, acc[t], 0, 0, 0);
      }
    }
    // C/D layout: class = t * 16 + j, node = tile * 16 + 4 * kq + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float best = -INFINITY;
      int arg = 0x7fffffff;
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const int cls = t * 16 + j;
        if (cls < a.c) {
          const float v = acc[t][r] + (a.bias ? a.bias[cls] : 0.f);
          if (v > best || arg == 0x7fffffff) { best = v; arg = cls; }
        }
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) {
        const float ov = __shfl_xor(best, off, kWave);
        const int oa = __shfl_xor(arg, off, kWave);
        if (oa != 0x7fffffff && (arg == 0x7fffffff || ov > best || (ov == best && oa < arg))) { best = ov; arg = oa; }
      }
      const int vnode = tile * 16 + 4 * kq + r;
      if (j == r && vnode < a.n) {
        const int ok = (arg == a.labels[vnode]) ? 1 : 0;
        const unsigned sp = a.split[vnode];
        hit_train += ok & static_cast<int>(sp & 1u);
        hit_val += ok & static_cast<int>((sp >> 1) & 1u);
        hit_test += ok & static_cast<int>((sp >> 2) & 1u);
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    hit_train += __shfl_xor(hit_train, off, kWave);
    hit_val += __shfl_xor(hit_val, off, kWave);
    hit_test += __shfl_xor(hit_test, off, kWave);
  }
  if (lane == 0) {
    if (hit_train) atomicAdd(a.state + 0, hit_train);
    if (hit_val) atomicAdd(a.state + 1, hit_val);
    if (hit_test) atomicAdd(a.state + 2, hit_test);
  }
}

__global__ void early_stop_update_kernel(int* __restrict__ state, int step, int* __restrict__ trace, int trace_capacity) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int tr = state[0], va = state[1], te = state[2];
  const int slot = state[7];
  if (trace != nullptr && slot < trace_capacity) {
    trace[4 * slot + 0] = tr; trace[4 * slot + 1] = va; trace[4 * slot + 2] = te; trace[4 * slot + 3] = step;
  }
  if (va > state[4]) {
    state[3] = tr; state[4] = va; state[5] = te; state[6] = step;
  }
  state[0] = 0; state[1] = 0; state[2] = 0;
  state[7] = slot + 1;
}

template <int CT>
int launch_decode(const DecArgs& a, hipStream_t st) {
  const size_t lds = static_cast<size_t>(16) * CT * a.dpad * sizeof(float);
  GNPDE_CHECK_ARG(lds <= 160 * 1024, GNPDE_ESHAPE, "early_stop_eval: decoder %d x %d does not fit the LDS", a.c, a.d);
  if (lds > 48 * 1024) {
    GNPDE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_count_kernel<CT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(lds)));
  }
  const int n_tiles = (a.n + 15) / 16;
  int grid = (n_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
  if (grid > 1024) grid = 1024;     // persistent beyond that: the decoder is staged into the LDS once per block
  hipLaunchKernelGGL((decode_count_kernel<CT>), dim3(grid), dim3(kBlock), lds, st, a);
  GNPDE_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int enqueue_early_stop_eval(const gnpde_decoder_t& dec, const float* y, int ld, int n, int step, int* state, int* trace,
                            int trace_capacity, hipStream_t st) {
  if (n > 0) {
    DecArgs a;
    a.y = y; a.weight = dec.weight; a.bias = dec.bias; a.labels = dec.labels; a.split = dec.split; a.state = state;
    a.n = n; a.ld = ld; a.d = dec.d_dec; a.c = dec.n_classes;
    a.d16 = (dec.d_dec + 15) / 16 * 16;
    a.dpad = a.d16 + 4;
    a.vec = (ld % 4 == 0) && (reinterpret_cast<uintptr_t>(y) % 16 == 0);
    int rc;
    const int ct = (dec.n_classes + 15) / 16;
    switch (ct) {
      case 1: rc = launch_decode<1>(a, st); break;
      case 2: rc = launch_decode<2>(a, st); break;
      case 3: rc = launch_decode<3>(a, st); break;
      case 4: rc = launch_decode<4>(a, st); break;
      default:
        set_error("early_stop_eval: %d classes (at most 64 supported)", dec.n_classes);
        return GNPDE_ESHAPE;
    }
    if (rc) return rc;
  }
  hipLaunchKernelGGL(early_stop_update_kernel, dim3(1), dim3(1), 0, st, state, step, trace, trace_capacity);
  GNPDE_LAUNCH_CHECK();
  return 0;
}

int check_decoder(const gnpde_decoder_t* dec, int d_state) {
  GNPDE_CHECK_ARG(dec != nullptr, GNPDE_EINVAL, "decoder is null");
  GNPDE_CHECK_ARG(dec->weight && dec->labels && dec->split, GNPDE_EINVAL, "decoder: weight / labels / split must be set");
  GNPDE_CHECK_ARG(dec->n_classes >= 1 && dec->d_dec >= 1, GNPDE_EINVAL, "decoder: bad shape %d x %d", dec->n_classes, dec->d_dec);
  GNPDE_CHECK_ARG(dec->d_dec <= d_state, GNPDE_EINVAL, "decoder reads %d columns of a %d-wide state", dec->d_dec, d_state);
  GNPDE_CHECK_ARG(dec->n_classes <= 64, GNPDE_ESHAPE, "decoder: %d classes (at most 64 supported)", dec->n_classes);
  return 0;
}

}  // namespace gnpde

using namespace gnpde;

extern "C" int gnpde_early_stop_reset(int32_t* state, void* stream) {
  GNPDE_CHECK_ARG(state != nullptr, GNPDE_EINVAL, "early_stop_reset: state is null");
  GNPDE_HIP(hipMemsetAsync(state, 0, GNPDE_EARLY_STATE_INTS * sizeof(int32_t), static_cast<hipStream_t>(stream)));
  return 0;
}

extern "C" int gnpde_early_stop_eval(const gnpde_decoder_t* dec, const float* y, int32_t d, int32_t ld, int32_t n, int32_t step,
                                     int32_t* state, int32_t* trace, int32_t trace_capacity, void* stream) {
  int rc = check_decoder(dec, d);
  if (rc) return rc;
  GNPDE_CHECK_ARG(y && state && n >= 0 && ld >= d, GNPDE_EINVAL, "early_stop_eval: bad argument");
  GNPDE_CHECK_ARG(trace != nullptr || trace_capacity == 0, GNPDE_EINVAL, "early_stop_eval: trace capacity without a trace");
  return enqueue_early_stop_eval(*dec, y, ld, n, step, state, trace, trace_capacity, static_cast<hipStream_t>(stream));
}
