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
//                         split; one partial count per block, no atomics (several thousand same-address atomics
//                         serialise in one L2 channel: 180 us at the ogbn-arxiv shape, measured);
//   early_stop_update     one block: sums the partial counts, then strict "val > best_val"
//                         (early_stop_solver.py:156-157 / :79-80) on the integer counts -- the denominators are
//                         equal, so this is the reference's float comparison without the division.
//
// The ogbn-arxiv branch of the reference applies log_softmax before the arg-max (:192-193); it does not change
// the arg-max and the loss computed next to it is discarded (:196-198), so neither is evaluated here.
#include "common.h"

namespace gnpde {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kStateHead = 8;                                     // result fields; per-block partial counts follow
constexpr int kMaxParts = (GNPDE_EARLY_STATE_INTS - kStateHead) / 3;

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
  const int* gate;   // NULL, or a device flag: 0 = this evaluation is skipped (device-controlled dopri5: rejected trial steps)
};

template <int CT>
__global__ __launch_bounds__(kBlock) void decode_count_kernel(const DecArgs a) {
  if (a.gate != nullptr && *a.gate == 0) return;   // (uniform over the grid)
  extern __shared__ float w_lds[];   // [16 * CT][dpad], zero padded in both directions
  for (int row = threadIdx.x >> 6; row < 16 * CT; row += kWavesPerBlock)
    for (int k = threadIdx.x & (kWave - 1); k < a.dpad; k += kWave)
      w_lds[row * a.dpad + k] = (row < a.c && k < a.d) ? a.weight[static_cast<size_t>(row) * a.d + k] : 0.f;
  __syncthreads();

  constexpr int KB = 8;              // 16-column blocks per batch: their loads are issued together (128 columns)
  const int lane = threadIdx.x & (kWave - 1);
  const int j = lane & 15, kq = lane >> 4;
  const int n_tiles = (a.n + 15) / 16;
  const int n_batch = (a.d16 + 16 * KB - 1) / (16 * KB);
  const int stride = gridDim.x * kWavesPerBlock;
  int hit_train = 0, hit_val = 0, hit_test = 0;

  // batch `bt` of tile `tile`: this lane's 8 float4 of the A operand (row tile * 16 + j)
  auto load_batch = [&](float4 (&av)[KB], int tile, int bt) {
    const int node = tile * 16 + j;
    const bool live = node < a.n;
    const float* yrow = a.y + static_cast<size_t>(live ? node : 0) * a.ld;
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const int k = (bt * KB + u) * 16 + 4 * kq;
      av[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live && k < a.d) {
        if (a.vec && k + 3 < a.d) {
          av[u] = *reinterpret_cast<const float4*>(yrow + k);
        } else {
          av[u].x = yrow[k];
          if (k + 1 < a.d) av[u].y = yrow[k + 1];
          if (k + 2 < a.d) av[u].z = yrow[k + 2];
          if (k + 3 < a.d) av[u].w = yrow[k + 3];
        }
      }
    }
  };

  int tile = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  int bt = 0;
  float4 cur[KB], nxt[KB];
  if (tile < n_tiles) load_batch(cur, tile, 0);
  f32x4 acc[CT];
  while (tile < n_tiles) {
    // software pipeline: the next batch (same tile, or the wave's next tile) is in flight during the MFMAs
    int ntile = tile, nbt = bt + 1;
    if (nbt == n_batch) { nbt = 0; ntile = tile + stride; }
    if (ntile < n_tiles) load_batch(nxt, ntile, nbt);
    if (bt == 0) {
#pragma unroll
      for (int t = 0; t < CT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const int kb = (bt * KB + u) * 16;
      if (kb < a.d16) {                                                  // wave-uniform
        const int k = kb + 4 * kq;
        const float x[4] = {fmaxf(cur[u].x, 0.f), fmaxf(cur[u].y, 0.f), fmaxf(cur[u].z, 0.f), fmaxf(cur[u].w, 0.f)};  // F.relu
        float bv[CT][4];
#pragma unroll
        for (int t = 0; t < CT; ++t) {
          const float4 w4 = *reinterpret_cast<const float4*>(w_lds + (t * 16 + j) * a.dpad + k);
          bv[t][0] = w4.x; bv[t][1] = w4.y; bv[t][2] = w4.z; bv[t][3] = w4.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < CT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[i], bv[t][i], acc[t], 0, 0, 0);
      }
    }
    if (bt == n_batch - 1) {
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
    for (int u = 0; u < KB; ++u) cur[u] = nxt[u];
    tile = ntile;
    bt = nbt;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    hit_train += __shfl_xor(hit_train, off, kWave);
    hit_val += __shfl_xor(hit_val, off, kWave);
    hit_test += __shfl_xor(hit_test, off, kWave);
  }
  __shared__ int red[kWavesPerBlock][3];
  if (lane == 0) {
    red[threadIdx.x >> 6][0] = hit_train; red[threadIdx.x >> 6][1] = hit_val; red[threadIdx.x >> 6][2] = hit_test;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    int* part = a.state + kStateHead + 3 * blockIdx.x;
    part[threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}

__global__ __launch_bounds__(kBlock) void early_stop_update_kernel(int* __restrict__ state, int n_parts, int step,
                                                                   int* __restrict__ trace, int trace_capacity,
                                                                   const int* __restrict__ gate, const int* __restrict__ tag) {
  if (gate != nullptr && *gate == 0) return;
  if (tag != nullptr) step = *tag;               // the step tag lives on the device (accepted-step count of the controller)
  __shared__ int red[kBlock][3];
  int s0 = 0, s1 = 0, s2 = 0;
  for (int i = threadIdx.x; i < n_parts; i += kBlock) {
    const int* part = state + kStateHead + 3 * i;
    s0 += part[0]; s1 += part[1]; s2 += part[2];
  }
  red[threadIdx.x][0] = s0; red[threadIdx.x][1] = s1; red[threadIdx.x][2] = s2;
  __syncthreads();
  for (int off = kBlock / 2; off >= 1; off >>= 1) {
    if (static_cast<int>(threadIdx.x) < off) {
      red[threadIdx.x][0] += red[threadIdx.x + off][0];
      red[threadIdx.x][1] += red[threadIdx.x + off][1];
      red[threadIdx.x][2] += red[threadIdx.x + off][2];
    }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const int tr = red[0][0], va = red[0][1], te = red[0][2];
  const int slot = state[7];
  if (trace != nullptr && slot < trace_capacity) {
    trace[4 * slot + 0] = tr; trace[4 * slot + 1] = va; trace[4 * slot + 2] = te; trace[4 * slot + 3] = step;
  }
  if (va > state[4]) {
    state[3] = tr; state[4] = va; state[5] = te; state[6] = step;
  }
  state[0] = tr; state[1] = va; state[2] = te;    // counts of the latest evaluation
  state[7] = slot + 1;
}

template <int CT>
int launch_decode(const DecArgs& a, hipStream_t st, int* n_parts) {
  const size_t lds = static_cast<size_t>(16) * CT * a.dpad * sizeof(float);
  GNPDE_CHECK_ARG(lds <= 160 * 1024, GNPDE_ESHAPE, "early_stop_eval: decoder %d x %d does not fit the LDS", a.c, a.d);
  if (lds > 48 * 1024) {
    GNPDE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_count_kernel<CT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(lds)));
  }
  // persistent waves with an equal number of tiles each (the decoder is staged into the LDS once per block)
  const int n_tiles = (a.n + 15) / 16;
  const int max_waves = 256 * 3 * kWavesPerBlock;
  const int per_wave = (n_tiles + max_waves - 1) / max_waves;
  const int waves = (n_tiles + per_wave - 1) / per_wave;
  int grid = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
  if (grid > kMaxParts) grid = kMaxParts;
  *n_parts = grid;
  hipLaunchKernelGGL((decode_count_kernel<CT>), dim3(grid), dim3(kBlock), lds, st, a);
  GNPDE_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int enqueue_early_stop_eval(const gnpde_decoder_t& dec, const float* y, int ld, int n, int step, int* state, int* trace,
                            int trace_capacity, hipStream_t st, const int* gate, const int* tag) {
  int n_parts = 0;
  if (n > 0) {
    DecArgs a;
    a.gate = gate;
    a.y = y; a.weight = dec.weight; a.bias = dec.bias; a.labels = dec.labels; a.split = dec.split; a.state = state;
    a.n = n; a.ld = ld; a.d = dec.d_dec; a.c = dec.n_classes;
    a.d16 = (dec.d_dec + 15) / 16 * 16;
    a.dpad = a.d16 + 4;
    a.vec = (ld % 4 == 0) && (reinterpret_cast<uintptr_t>(y) % 16 == 0);
    int rc;
    const int ct = (dec.n_classes + 15) / 16;
    switch (ct) {
      case 1: rc = launch_decode<1>(a, st, &n_parts); break;
      case 2: rc = launch_decode<2>(a, st, &n_parts); break;
      case 3: rc = launch_decode<3>(a, st, &n_parts); break;
      case 4: rc = launch_decode<4>(a, st, &n_parts); break;
      default:
        set_error("early_stop_eval: %d classes (at most 64 supported)", dec.n_classes);
        return GNPDE_ESHAPE;
    }
    if (rc) return rc;
  }
  hipLaunchKernelGGL(early_stop_update_kernel, dim3(1), dim3(kBlock), 0, st, state, n_parts, step, trace, trace_capacity, gate, tag);
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
  GNPDE_HIP(hipMemsetAsync(state, 0, kStateHead * sizeof(int32_t), static_cast<hipStream_t>(stream)));
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
