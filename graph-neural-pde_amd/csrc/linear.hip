// out[n, m] = x[n, d] * W[m, d]^T + b[m]  on the fp32 matrix cores.
//
// The only GEMM-shaped work on the path: nn.Linear Q and K of SpGraphTransAttentionLayer
// (reference src/function_transformer_attention.py:174-175, fused into one [N,d]x[d,2A] product;
// V is skipped, its result is dead when mix_features is false, :34-35) and torch.mm(x, W) of the
// GAT layer (src/function_GAT_attention.py:106).  Tall-skinny: N = 1e5..1e6 rows, d and m <= 256,
// so it is bound by reading x once; v_mfma_f32_16x16x4_f32 is exact fp32 (bit-equal to an fmaf
// chain) and keeps the VALU free.
//
// One wavefront owns 16 consecutive rows and all m output columns (m/16 accumulators).  Every lane
// loads ONE float4 of x per 16-wide K block -- x[row0 + (l&15)][kb + 4*(l>>4) .. +3] -- and uses its
// i-th component as the A operand of the i-th MFMA of that block; the matching B operand is the i-th
// component of the float4 W[col0 + (l&15)][kb + 4*(l>>4) .. +3].  The k index of an MFMA step is thus
// {kb+i, kb+4+i, kb+8+i, kb+12+i}: a permutation of the K order, which a dot product does not care
// about, and it makes both operand loads 16-byte vector loads straight from row-major memory with no
// LDS transpose.
#include "common.h"

namespace gnpde {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// relu on the A operand (the decoder of the reference's GNN.forward applies F.relu to the state before m2)
__device__ __forceinline__ f32x4 relu_if(f32x4 v, int relu) {
  if (relu) {
    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
  }
  return v;
}

template <bool ALIGNED>
__device__ __forceinline__ f32x4 load4_guard(const float* __restrict__ base, int k, int kmax, bool row_ok) {
  f32x4 r = {0.f, 0.f, 0.f, 0.f};
  if (!row_ok) return r;
  if constexpr (ALIGNED) {
    if (k + 3 < kmax) {
      const float4 t = *reinterpret_cast<const float4*>(base + k);
      r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
      return r;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (k + i < kmax) r[i] = base[k + i];
  return r;
}

// MT = number of 16-column output tiles kept in registers by one wave.
// FULL: every column tile is complete and d % 16 == 0 -> no guards on the operand loads (rows past the end
// are clamped to the last row and only their stores are masked), so the loads of a K batch are
// straight-line code the scheduler can issue back to back, and one launch covers the ragged last tile.
template <int MT, bool ALIGNED, bool FULL, int KUV = (MT <= 2 ? 4 : 2)>
__global__ __launch_bounds__(kBlock) void linear_kernel(const float* __restrict__ x, int n, int d, int ldx,
                                                        const float* __restrict__ W, int m, int ldw,
                                                        const float* __restrict__ b, float* __restrict__ out,
                                                        int ldo, int col_base, int row_base, int relu) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const long long tile = static_cast<long long>(blockIdx.x) * kWavesPerBlock + wave;
  const int row0 = row_base + static_cast<int>(tile * 16);
  if (row0 >= n) return;
  col_base += static_cast<int>(blockIdx.y) * (MT * 16);     // 2-D launches: grid.y walks the column groups (small n, below)
  const int r = lane & 15;       // row within the tile (A operand) / column within a 16-col tile (B)
  const int kq = lane >> 4;      // which 4-wide K quarter of the 16-wide K block
  const int arow = row0 + r;
  const bool arow_ok = arow < n;
  const float* xrow = x + static_cast<size_t>(arow_ok ? arow : (FULL ? n - 1 : 0)) * ldx;

  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // KU 16-wide K blocks per iteration: all operand loads of an iteration are issued before the first
  // MFMA so that several HBM round trips overlap (a wave has only d/16 dependent-free blocks to hide
  // ~2 us of latency behind)
  constexpr int KU = KUV;
  for (int kb = 0; kb < d; kb += 16 * KU) {
    f32x4 av[KU];
    f32x4 bv[KU][MT];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int k = kb + 16 * u + 4 * kq;
      if constexpr (FULL) {
        if (kb + 16 * u < d) {  // d % 16 == 0: a K block is complete or absent -- a scalar (wave-uniform) test
          const float4 ta = *reinterpret_cast<const float4*>(xrow + k);
          av[u] = relu_if(f32x4{ta.x, ta.y, ta.z, ta.w}, relu);
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            const float4 tb = *reinterpret_cast<const float4*>(W + static_cast<size_t>(col_base + t * 16 + r) * ldw + k);
            bv[u][t] = f32x4{tb.x, tb.y, tb.z, tb.w};
          }
        } else {
          av[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < MT; ++t) bv[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      } else {
        av[u] = relu_if(load4_guard<ALIGNED>(xrow, k, d, arow_ok), relu);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const int col = col_base + t * 16 + r;
          const bool ok = col < m;
          bv[u][t] = load4_guard<ALIGNED>(W + static_cast<size_t>(ok ? col : 0) * ldw, k, d, ok);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < KU; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < MT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][i], bv[u][t][i], acc[t], 0, 0, 0);
  }

  // C/D layout of the 16x16 tile: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int col = col_base + t * 16 + r;
    if constexpr (!FULL) {
      if (col >= m) continue;
    }
    const float bias = b != nullptr ? b[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int orow = row0 + 4 * kq + i;
      if (orow < n) out[static_cast<size_t>(orow) * ldo + col] = acc[t][i] + bias;
    }
  }
}

// Persistent variant for the projection shapes of the hot path (d = 16*KB <= 128 wide, m = 16*MT <= 32 columns):
// the whole B operand (all of W) stays in registers for the life of the wavefront, wavefronts stride over the
// 16-row tiles, and the A operand of the NEXT tile is in flight while the MFMAs of the current one issue, so
// the kernel streams x at memory speed instead of paying one HBM round trip per tile.
template <int MT, int KB>
__global__ __launch_bounds__(kBlock) void linear_persistent_kernel(const float* __restrict__ x, int n, int ldx,
                                                                   const float* __restrict__ W, int ldw,
                                                                   const float* __restrict__ b, float* __restrict__ out,
                                                                   int ldo, int col_base, int relu) {
  const int lane = threadIdx.x & (kWave - 1);
  const int r = lane & 15, kq = lane >> 4;
  const long long n_tiles = (static_cast<long long>(n) + 15) / 16;
  const long long stride = static_cast<long long>(gridDim.x) * kWavesPerBlock;
  long long tile = static_cast<long long>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  if (tile >= n_tiles) return;

  f32x4 bv[KB][MT];
#pragma unroll
  for (int u = 0; u < KB; ++u)
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const float4 tb = *reinterpret_cast<const float4*>(W + static_cast<size_t>(col_base + t * 16 + r) * ldw + 16 * u + 4 * kq);
      bv[u][t] = f32x4{tb.x, tb.y, tb.z, tb.w};
    }
  float bias[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) bias[t] = b != nullptr ? b[col_base + t * 16 + r] : 0.0f;

  auto load_tile = [&](long long tl, f32x4 (&av)[KB]) {
    long long row = tl * 16 + r;
    if (row >= n) row = n - 1;  // ragged last tile: clamp reads, mask writes
    const float* xr = x + static_cast<size_t>(row) * ldx + 4 * kq;
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const float4 ta = *reinterpret_cast<const float4*>(xr + 16 * u);
      av[u] = relu_if(f32x4{ta.x, ta.y, ta.z, ta.w}, relu);
    }
  };

  f32x4 cur[KB], nxt[KB];
  load_tile(tile, cur);
  while (true) {
    const long long next = tile + stride;
    const bool more = next < n_tiles;
    if (more) load_tile(next, nxt);
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < KB; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[u][i], bv[u][t][i], acc[t], 0, 0, 0);
    const long long row0 = tile * 16;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long orow = row0 + 4 * kq + i;
        if (orow < n) out[static_cast<size_t>(orow) * ldo + col_base + t * 16 + r] = acc[t][i] + bias[t];
      }
    if (!more) break;
#pragma unroll
    for (int u = 0; u < KB; ++u) cur[u] = nxt[u];
    tile = next;
  }
}


// Wide projections (m >= 64 output columns, e.g. the R-MAT configuration's [N,256] x [256,128]): linear_kernel re-reads all of
// W (128 KB) from L2 for every 16-row tile -- 8x the bytes of the tile itself, 2.84 ms where the fp32 MFMA time is 0.9 ms.
// Here W is staged ONCE per workgroup in LDS (rows padded by 4 floats, so the 16 lanes of a ds_read_b128 group hit 16
// different 16-byte slots) and the 8 waves of the workgroup stride over the row tiles: per 16-K block a wave issues MT
// B-operand reads from LDS and 4 MT MFMAs, with the A operands of the next K batch in flight from HBM.
template <int MT, int DMAX>
__global__ __launch_bounds__(512) void linear_lds_kernel(const float* __restrict__ x, int n, int d, int ldx,
                                                         const float* __restrict__ W, int ldw, const float* __restrict__ b,
                                                         float* __restrict__ out, int ldo, int col_base, int relu) {
  constexpr int LDL = DMAX + 4;
  constexpr int KU = 4;                                  // 16-wide K blocks per batch
  __shared__ float lds[MT * 16 * LDL];
  const int d4 = d >> 2;
  for (int idx = threadIdx.x; idx < MT * 16 * d4; idx += 512) {
    const int row = idx / d4, c4 = idx - row * d4;
    *reinterpret_cast<float4*>(&lds[row * LDL + 4 * c4]) =
        *reinterpret_cast<const float4*>(W + static_cast<size_t>(col_base + row) * ldw + 4 * c4);
  }
  __syncthreads();
  const int lane = threadIdx.x & (kWave - 1);
  const int r = lane & 15, kq = lane >> 4;
  float bias[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) bias[t] = b != nullptr ? b[col_base + t * 16 + r] : 0.0f;
  const long long n_tiles = (static_cast<long long>(n) + 15) / 16;
  const long long stride = static_cast<long long>(gridDim.x) * 8;
  const int nkb = d >> 4;
  for (long long tile = static_cast<long long>(blockIdx.x) * 8 + (threadIdx.x >> 6); tile < n_tiles; tile += stride) {
    long long row = tile * 16 + r;
    if (row >= n) row = n - 1;                           // ragged last tile: clamp reads, mask writes
    const float* xr = x + static_cast<size_t>(row) * ldx + 4 * kq;
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 cur[KU], nxt[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const float4 ta = u < nkb ? *reinterpret_cast<const float4*>(xr + 16 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
      cur[u] = relu_if(f32x4{ta.x, ta.y, ta.z, ta.w}, relu);
    }
    for (int u0 = 0; u0 < nkb; u0 += KU) {
#pragma unroll
      for (int u = 0; u < KU; ++u) {                     // next batch's A operands (wave-uniform guard)
        const int kb = u0 + KU + u;
        const float4 ta = kb < nkb ? *reinterpret_cast<const float4*>(xr + 16 * kb) : make_float4(0.f, 0.f, 0.f, 0.f);
        nxt[u] = relu_if(f32x4{ta.x, ta.y, ta.z, ta.w}, relu);
      }
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        if (u0 + u < nkb) {
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            const float4 tb = *reinterpret_cast<const float4*>(&lds[(t * 16 + r) * LDL + 16 * (u0 + u) + 4 * kq]);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[u][0], tb.x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[u][1], tb.y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[u][2], tb.z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[u][3], tb.w, acc[t], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < KU; ++u) cur[u] = nxt[u];
    }
    const long long row0 = tile * 16;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long orow = row0 + 4 * kq + i;
        if (orow < n) out[static_cast<size_t>(orow) * ldo + col_base + t * 16 + r] = acc[t][i] + bias[t];
      }
  }
}

// Round 3: the same persistent kernel with the A operand fetched in FULL LINES and transposed through LDS.  The kernel above
// loads fragment-shaped -- lane (r, kq) reads 16 bytes of row r, so one load instruction touches 16 rows x 64 B, half a cache
// line each -- and streams x at 3.7 TB/s with the MFMA pipe 30 % busy.  Here lane l of load `it` reads chunk it * 64 + l of the
// tile's 16 x d floats (two whole rows, 1 KB contiguous, per instruction), the wave writes the chunks into its own LDS slab
// (chunk (r, j) at r * d/4 + (j ^ (r & 7)): the xor keeps both the row-linear writes and the column-strided fragment reads
// free of bank conflicts) and reads the fragments back as ds_read_b128.  Wave-private slabs: no barrier, LDS operations of a
// wave execute in order.  Same MFMA sequence, same k order, same results bit for bit.
template <int MT, int KB>
__global__ __launch_bounds__(kBlock) void linear_staged_kernel(const float* __restrict__ x, int n, int ldx,
                                                               const float* __restrict__ W, int ldw,
                                                               const float* __restrict__ b, float* __restrict__ out,
                                                               int ldo, int col_base, int relu, int variant) {
  // variant (A/B knob gnpde_tune(8, 3 | 4 | 5), bit 0: nontemporal loads of x, bit 1: every wave walks a CONTIGUOUS range of tiles
  // instead of striding over the table by the grid; results are bit-identical in every variant)
  constexpr int CPR = KB * 4;                 // 16-byte chunks per row (d = 16 KB floats)
  constexpr int CHUNKS = 16 * CPR;            // per tile
  constexpr int PER_LANE = CHUNKS / kWave;    // = KB
  __shared__ f32x4 slab[kWavesPerBlock][CHUNKS];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const long long n_tiles = (static_cast<long long>(n) + 15) / 16;
  const long long n_waves = static_cast<long long>(gridDim.x) * kWavesPerBlock;
  const bool contiguous = (variant & 2) != 0;
  const bool nt = (variant & 1) != 0;
  const long long per_wave = (n_tiles + n_waves - 1) / n_waves;
  const long long wid = static_cast<long long>(blockIdx.x) * kWavesPerBlock + wave;
  const long long stride = contiguous ? 1 : n_waves;
  long long tile = contiguous ? wid * per_wave : wid;
  const long long tile_end = contiguous ? ((wid + 1) * per_wave < n_tiles ? (wid + 1) * per_wave : n_tiles) : n_tiles;
  if (tile >= tile_end) return;

  f32x4 bv[KB][MT];
#pragma unroll
  for (int u = 0; u < KB; ++u)
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const float4 tb = *reinterpret_cast<const float4*>(W + static_cast<size_t>(col_base + t * 16 + r) * ldw + 16 * u + 4 * kq);
      bv[u][t] = f32x4{tb.x, tb.y, tb.z, tb.w};
    }
  float bias[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) bias[t] = b != nullptr ? b[col_base + t * 16 + r] : 0.0f;

  // chunk c = it * 64 + lane of a tile: row c / CPR, column chunk c % CPR
  auto load_tile = [&](long long tl, f32x4 (&g)[PER_LANE]) {
#pragma unroll
    for (int it = 0; it < PER_LANE; ++it) {
      const int c = it * kWave + lane;
      long long row = tl * 16 + c / CPR;
      if (row >= n) row = n - 1;               // ragged last tile: clamp reads, mask writes
      const f32x4* src = reinterpret_cast<const f32x4*>(x + static_cast<size_t>(row) * ldx + 4 * (c % CPR));
      g[it] = nt ? __builtin_nontemporal_load(src) : *src;
    }
  };
  f32x4* my = slab[wave];
  f32x4 g[PER_LANE];
  load_tile(tile, g);
  while (true) {
#pragma unroll
    for (int it = 0; it < PER_LANE; ++it) {
      const int c = it * kWave + lane;
      const int rr = c / CPR, j = c % CPR;
      my[rr * CPR + (j ^ (rr & 7))] = g[it];
    }
    const long long next = tile + stride;
    const bool more = next < tile_end;
    if (more) load_tile(next, g);              // in flight during the MFMAs of this tile
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragments two K blocks at a time: the reads of the next pair are in flight during the MFMAs of this one
#pragma unroll
    for (int u = 0; u < KB; u += 2) {
      f32x4 cu[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) cu[q] = relu_if(my[r * CPR + ((4 * (u + q) + kq) ^ (r & 7))], relu);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cu[q][i], bv[u + q][t][i], acc[t], 0, 0, 0);
    }
    const long long row0 = tile * 16;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long orow = row0 + 4 * kq + i;
        if (orow < n) out[static_cast<size_t>(orow) * ldo + col_base + t * 16 + r] = acc[t][i] + bias[t];
      }
    if (!more) break;
    __builtin_amdgcn_wave_barrier();           // the fragment reads above precede the next tile's slab writes
    tile = next;
  }
}

// Round 6 variants of the staged kernel for the q||k projection shape (MT = 2: m = 32 output columns), selected by
// gnpde_tune(8, 6 | 7 | 8) for A/B runs and by default where they measured faster:
//   PAIRC  column tile t of lane r is output column 2 r + t, so a lane's two accumulators of an output row are ADJACENT columns and leave
//          as one 8-byte store: a store instruction writes four complete 128-byte output rows (the kernel above writes every output
//          line as two 64-byte halves by two different instructions);
//   DEPTH2 two row tiles of x in flight per wave (registers) instead of one.
// Same MFMA sequence and k order per output element: bit-identical results.
template <int KB, bool PAIRC, bool DEPTH2, int DIAG = 0>
__global__ __launch_bounds__(kBlock) void linear_staged2_kernel(const float* __restrict__ x, int n, int ldx,
                                                                const float* __restrict__ W, int ldw,
                                                                const float* __restrict__ b, float* __restrict__ out,
                                                                int ldo, int col_base, int relu, float* __restrict__ out2 = nullptr,
                                                                int split = 0) {
  // out2 / split (round 6): output columns >= split go to out2[row * ldo + col - split] -- the q||k projection as TWO tables [n, A] when a
  // key row is shorter than a cache line (A <= 16: the attention's gathers of 64-byte k rows then fetch no q halves of 128-byte lines)
  // DIAG (A/B diagnostics, gnpde_tune(13, v)): 1 = no output stores (unless a result is NaN), 2 = no LDS / MFMA work (the loads alone)
  constexpr int diag = DIAG;
  constexpr int MT = 2;
  constexpr int CPR = KB * 4;
  constexpr int CHUNKS = 16 * CPR;
  constexpr int PER_LANE = CHUNKS / kWave;
  __shared__ f32x4 slab[kWavesPerBlock][CHUNKS];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const long long n_tiles = (static_cast<long long>(n) + 15) / 16;
  const long long stride = static_cast<long long>(gridDim.x) * kWavesPerBlock;
  long long tile = static_cast<long long>(blockIdx.x) * kWavesPerBlock + wave;
  if (tile >= n_tiles) return;

  int wcol[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) wcol[t] = col_base + (PAIRC ? 2 * r + t : t * 16 + r);
  f32x4 bv[KB][MT];
#pragma unroll
  for (int u = 0; u < KB; ++u)
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const float4 tb = *reinterpret_cast<const float4*>(W + static_cast<size_t>(wcol[t]) * ldw + 16 * u + 4 * kq);
      bv[u][t] = f32x4{tb.x, tb.y, tb.z, tb.w};
    }
  float bias[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) bias[t] = b != nullptr ? b[wcol[t]] : 0.0f;

  // chunk c = it * 64 + lane of a tile: row c / CPR = it * (64 / CPR) + lane / CPR, column chunk j = lane % CPR (CPR = 16 or 32 divides 64):
  // one per-lane offset, the rest are compile-time multiples of the row stride
  constexpr int RPI = kWave / CPR;               // rows per load instruction
  const int lr = lane / CPR, j = lane % CPR;
  auto load_tile = [&](long long tl, f32x4 (&g)[PER_LANE]) {
    const long long row0 = tl * 16;
    const int rmax = static_cast<int>(n - 1 - row0 < 15 ? n - 1 - row0 : 15);       // (wave-uniform) ragged last tile: clamp reads, mask writes
    const float* base = x + static_cast<size_t>(row0) * ldx + 4 * j;
#pragma unroll
    for (int it = 0; it < PER_LANE; ++it) {
      const int rr = it * RPI + lr;
      g[it] = *reinterpret_cast<const f32x4*>(base + static_cast<size_t>(rr < rmax ? rr : rmax) * ldx);
    }
  };
  f32x4* my = slab[wave];
  auto consume = [&](long long tl, const f32x4 (&g)[PER_LANE]) {
#pragma unroll
    for (int it = 0; it < PER_LANE; ++it) {
      const int rr = it * RPI + lr;
      if (diag == 2) { if (g[it][0] != g[it][0]) out[lane] = g[it][1]; continue; }
      my[rr * CPR + (j ^ (rr & 7))] = g[it];
    }
  };
  auto compute = [&](long long tl) {
    if (diag == 2) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < KB; u += 2) {
      f32x4 cu[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) cu[q] = relu_if(my[r * CPR + ((4 * (u + q) + kq) ^ (r & 7))], relu);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cu[q][i], bv[u + q][t][i], acc[t], 0, 0, 0);
    }
    const long long row0 = tl * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long orow = row0 + 4 * kq + i;
      if (orow < n && (diag != 1 || acc[0][i] != acc[0][i])) {
        if constexpr (PAIRC) {
          const int c0 = col_base + 2 * r;
          float* dst = (split > 0 && c0 >= split) ? out2 + static_cast<size_t>(orow) * ldo + (c0 - split) : out + static_cast<size_t>(orow) * ldo + c0;
          *reinterpret_cast<float2*>(dst) = make_float2(acc[0][i] + bias[0], acc[1][i] + bias[1]);
        } else {
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            float* dst = (split > 0 && wcol[t] >= split) ? out2 + static_cast<size_t>(orow) * ldo + (wcol[t] - split)
                                                         : out + static_cast<size_t>(orow) * ldo + wcol[t];
            *dst = acc[t][i] + bias[t];
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();           // the fragment reads above precede the next tile's slab writes
  };
  if constexpr (!DEPTH2) {
    f32x4 g[PER_LANE];
    load_tile(tile, g);
    while (true) {
      consume(tile, g);
      const long long next = tile + stride;
      const bool more = next < n_tiles;
      if (more) load_tile(next, g);
      compute(tile);
      if (!more) break;
      tile = next;
    }
  } else {
    f32x4 ga[PER_LANE], gb[PER_LANE];
    load_tile(tile, ga);
    if (tile + stride < n_tiles) load_tile(tile + stride, gb);
    while (true) {
      consume(tile, ga);
      if (tile + 2 * stride < n_tiles) load_tile(tile + 2 * stride, ga);
      compute(tile);
      tile += stride;
      if (tile >= n_tiles) break;
      consume(tile, gb);
      if (tile + 2 * stride < n_tiles) load_tile(tile + 2 * stride, gb);
      compute(tile);
      tile += stride;
      if (tile >= n_tiles) break;
    }
  }
}

template <int MT, bool ALIGNED>
void launch_tile(const float* x, int n, int d, int ldx, const float* W, int m, int ldw, const float* b, float* out,
                 int ldo, int col, hipStream_t s, int relu) {
  // complete 16-row tiles with complete column tiles and d % 16 == 0 take the unguarded kernel
  const bool full_cols = ALIGNED && (d % 16 == 0) && (col + 16 * MT <= m);
  const long long tiles = (static_cast<long long>(n) + 15) / 16;
  if constexpr (MT <= 2) {
    if (full_cols && d <= 128 && g_tune[GNPDE_TUNE_LINEAR_STREAMING] != 1) {     // (values >= 3: variants of the staged kernel)
      // persistent grid: enough wavefronts to fill the chip (8 per SIMD at these register counts would be
      // ideal; W fragments + double-buffered A cost ~150 VGPRs -> 3 per SIMD), each striding over tiles
      long long blocks = 256LL * 3;
      const long long need = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
      if (blocks > need) blocks = need;
      const unsigned pg = static_cast<unsigned>(blocks);
      // d = 64 / 128 (4 or 8 chunks per lane and tile): full-line loads transposed through LDS (linear_staged_kernel);
      // gnpde_tune(8, 2) keeps the fragment-shaped loads for A/B runs
      if ((d == 128 || d == 64) && g_tune[GNPDE_TUNE_LINEAR_STREAMING] != 2) {
        const int knob = g_tune[GNPDE_TUNE_LINEAR_STREAMING];
        if constexpr (MT == 2) {
          if ((knob == 0 || (knob >= 6 && knob <= 11)) && ldo % 2 == 0 && col % 2 == 0 && reinterpret_cast<uintptr_t>(out) % 8 == 0) {
            // 0 (default since round 6) = 9: paired columns on a grid of 4 workgroups per CU (3 resident: the fourth starts as one ends) --
            // 24.4 -> 24.0 us in the headline solve, 26.9 -> 25.8 us stand-alone (profiles/r06_linear_ab.txt); 6: paired columns;
            // 7: two tiles in flight; 8: both; 10 / 11: 7 / 8 on grids of 4 / 2 workgroups per CU; 12: the round-3 kernel
            long long pg2 = (knob == 0 || knob >= 9) ? 256LL * (knob == 11 ? 2 : 4) : blocks;
            if (pg2 > need) pg2 = need;
            const unsigned g2 = static_cast<unsigned>(pg2);
            const int kv = knob == 0 ? 6 : (knob >= 9 ? knob - 3 : knob);
#define GNPDE_LS2(KBV, PC, D2) hipLaunchKernelGGL((linear_staged2_kernel<KBV, PC, D2>), dim3(g2), dim3(kBlock), 0, s, x, n, ldx, W, ldw, b, out, ldo, col, relu)
            const int diag = g_tune[GNPDE_TUNE_LINEAR_DIAG];
            if (d == 128 && diag == 1) hipLaunchKernelGGL((linear_staged2_kernel<8, true, false, 1>), dim3(g2), dim3(kBlock), 0, s, x, n, ldx, W, ldw, b, out, ldo, col, relu);
            else if (d == 128 && diag == 2) hipLaunchKernelGGL((linear_staged2_kernel<8, true, false, 2>), dim3(g2), dim3(kBlock), 0, s, x, n, ldx, W, ldw, b, out, ldo, col, relu);
            else if (d == 128) { if (kv == 6) GNPDE_LS2(8, true, false); else if (kv == 7) GNPDE_LS2(8, false, true); else GNPDE_LS2(8, true, true); }
            else { if (kv == 6) GNPDE_LS2(4, true, false); else if (kv == 7) GNPDE_LS2(4, false, true); else GNPDE_LS2(4, true, true); }
#undef GNPDE_LS2
            return;
          }
        }
        const int variant = knob >= 3 && knob <= 5 ? knob - 2 : 0;     // 3 / 4 / 5 -> 1 / 2 / 3
        if (d == 128)
          hipLaunchKernelGGL((linear_staged_kernel<MT, 8>), dim3(pg), dim3(kBlock), 0, s, x, n, ldx, W, ldw, b, out, ldo, col, relu, variant);
        else
          hipLaunchKernelGGL((linear_staged_kernel<MT, 4>), dim3(pg), dim3(kBlock), 0, s, x, n, ldx, W, ldw, b, out, ldo, col, relu, variant);
        return;
      }
#define GNPDE_LP(KBV) \
  hipLaunchKernelGGL((linear_persistent_kernel<MT, KBV>), dim3(pg), dim3(kBlock), 0, s, x, n, ldx, W, ldw, b, out, ldo, col, relu)
      switch (d / 16) {
        case 1: GNPDE_LP(1); return;
        case 2: GNPDE_LP(2); return;
        case 3: GNPDE_LP(3); return;
        case 4: GNPDE_LP(4); return;
        case 5: GNPDE_LP(5); return;
        case 6: GNPDE_LP(6); return;
        case 7: GNPDE_LP(7); return;
        case 8: GNPDE_LP(8); return;
        default: break;
      }
#undef GNPDE_LP
    }
  }
  if constexpr (MT >= 4 && ALIGNED) {
    if (full_cols && d <= 256 && g_tune[GNPDE_TUNE_LINEAR_STREAMING] == 0) {   // W staged in LDS, persistent workgroups
      const long long need = (tiles + 7) / 8;
      if (d <= 128) {
        long long blocks = 256LL * (MT == 4 ? 4 : 2);
        if (blocks > need) blocks = need;
        hipLaunchKernelGGL((linear_lds_kernel<MT, 128>), dim3(static_cast<unsigned>(blocks)), dim3(512), 0, s, x, n, d, ldx, W, ldw, b,
                           out, ldo, col, relu);
      } else {
        long long blocks = 256LL * (MT == 4 ? 2 : 1);
        if (blocks > need) blocks = need;
        hipLaunchKernelGGL((linear_lds_kernel<MT, 256>), dim3(static_cast<unsigned>(blocks)), dim3(512), 0, s, x, n, d, ldx, W, ldw, b,
                           out, ldo, col, relu);
      }
      return;
    }
  }
  const unsigned grid = static_cast<unsigned>((tiles + kWavesPerBlock - 1) / kWavesPerBlock);
  if (full_cols)
    hipLaunchKernelGGL((linear_kernel<MT, ALIGNED, true>), dim3(grid), dim3(kBlock), 0, s, x, n, d, ldx, W, m, ldw, b, out, ldo,
                       col, 0, relu);
  else
    hipLaunchKernelGGL((linear_kernel<MT, ALIGNED, false>), dim3(grid), dim3(kBlock), 0, s, x, n, d, ldx, W, m, ldw, b, out,
                       ldo, col, 0, relu);
}

// Small n (Cora: 2 708 rows, [n,80] x [80,256]): the persistent / LDS-staged kernels above are built to stream a tall x -- at 170 row
// tiles they put 22 workgroups on the chip, each staging 40 KB of W first, and need one launch per 128 output columns: 2 x 9.8 us
// per evaluation, two thirds of a launch-bound GRAND-nl evaluation on such a graph.  Here ONE launch covers all of [n, m]: a
// wavefront per (16-row tile, 32-column group), operands straight from L2, EIGHT 16-wide K blocks in flight per batch (d <= 128:
// every operand load of the wave is issued before its first MFMA -- the launch is a few dependent round trips long, nothing
// else).  Same MFMA sequence and k order per output element as
// every other variant (bit-identical results).
constexpr long long kSmallLinearTiles = 2048;      // n <= 32 768 rows

template <bool ALIGNED>
void launch_linear(const float* x, int n, int d, int ldx, const float* W, int m, int ldw, const float* b, float* out,
                   int ldo, hipStream_t s, int relu) {
  if constexpr (ALIGNED) {
    const long long tiles = (static_cast<long long>(n) + 15) / 16;
    if (tiles <= kSmallLinearTiles && d % 16 == 0 && m % 16 == 0 && g_tune[GNPDE_TUNE_LINEAR_STREAMING] == 0) {
      const unsigned gx = static_cast<unsigned>((tiles + kWavesPerBlock - 1) / kWavesPerBlock);
      if (m % 32 == 0)
        hipLaunchKernelGGL((linear_kernel<2, true, true, 8>), dim3(gx, m / 32), dim3(kBlock), 0, s, x, n, d, ldx, W, m, ldw, b, out, ldo, 0, 0, relu);
      else
        hipLaunchKernelGGL((linear_kernel<1, true, true, 8>), dim3(gx, m / 16), dim3(kBlock), 0, s, x, n, d, ldx, W, m, ldw, b, out, ldo, 0, 0, relu);
      return;
    }
  }
  int col = 0;
  while (col < m) {
    const int rem = (m - col + 15) / 16;
    if (rem >= 8) {
      launch_tile<8, ALIGNED>(x, n, d, ldx, W, m, ldw, b, out, ldo, col, s, relu);
      col += 128;
    } else if (rem >= 4) {
      launch_tile<4, ALIGNED>(x, n, d, ldx, W, m, ldw, b, out, ldo, col, s, relu);
      col += 64;
    } else if (rem >= 2) {
      launch_tile<2, ALIGNED>(x, n, d, ldx, W, m, ldw, b, out, ldo, col, s, relu);
      col += 32;
    } else {
      launch_tile<1, ALIGNED>(x, n, d, ldx, W, m, ldw, b, out, ldo, col, s, relu);
      col += 16;
    }
  }
}

}  // namespace

// The q||k projection as two tables (q [n, A], k [n, A]): offered where a key row is shorter than a 128-byte line and the launch would
// take the staged kernel anyway (tall x, d = 64 / 128, 2A = 32 output columns, 16-byte aligned operands).
bool linear_split_supported(const float* x, long long n, int d, int ldx, const float* W, int m, int ldw, int split) {
  const long long tiles = (n + 15) / 16;
  return m == 32 && split == 16 && (d == 64 || d == 128) && tiles > kSmallLinearTiles && ldx % 4 == 0 && ldw % 4 == 0 &&
         reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(W) % 16 == 0 && g_tune[GNPDE_TUNE_LINEAR_STREAMING] == 0 &&
         g_tune[GNPDE_TUNE_KEY_TABLE] != 1;
}

int launch_linear_split(const float* x, int n, int d, int ldx, const float* W, int m, int ldw, const float* b, float* out_q, float* out_k,
                        int split, hipStream_t s) {
  GNPDE_CHECK_ARG(x && W && out_q && out_k && linear_split_supported(x, n, d, ldx, W, m, ldw, split) &&
                  reinterpret_cast<uintptr_t>(out_q) % 8 == 0 && reinterpret_cast<uintptr_t>(out_k) % 8 == 0, GNPDE_ESHAPE,
                  "linear_split: shape without a two-table kernel (see linear_split_supported)");
  const long long tiles = (static_cast<long long>(n) + 15) / 16;
  long long blocks = 256LL * 4;
  const long long need = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > need) blocks = need;
  const unsigned g2 = static_cast<unsigned>(blocks);
  if (d == 128) hipLaunchKernelGGL((linear_staged2_kernel<8, true, false>), dim3(g2), dim3(kBlock), 0, s, x, n, ldx, W, ldw, b, out_q, split, 0, 0, out_k, split);
  else hipLaunchKernelGGL((linear_staged2_kernel<4, true, false>), dim3(g2), dim3(kBlock), 0, s, x, n, ldx, W, ldw, b, out_q, split, 0, 0, out_k, split);
  GNPDE_LAUNCH_CHECK();
  return 0;
}

int launch_linear_any(const float* x, int n, int d, int ldx, const float* W, int m, int ldw, const float* b, float* out,
                      int ldo, hipStream_t s, int relu) {
  GNPDE_CHECK_ARG(x && W && out, GNPDE_EINVAL, "linear: null pointer");
  GNPDE_CHECK_ARG(n >= 0 && d >= 1 && m >= 1 && ldx >= d && ldw >= d && ldo >= m, GNPDE_EINVAL,
                  "linear: bad shape n=%d d=%d m=%d ldx=%d ldw=%d ldo=%d", n, d, m, ldx, ldw, ldo);
  if (n == 0) return 0;
  const bool al = (ldx % 4 == 0) && (ldw % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                  (reinterpret_cast<uintptr_t>(W) % 16 == 0);
  if (al) launch_linear<true>(x, n, d, ldx, W, m, ldw, b, out, ldo, s, relu);
  else launch_linear<false>(x, n, d, ldx, W, m, ldw, b, out, ldo, s, relu);
  GNPDE_LAUNCH_CHECK();
  return 0;
}

}  // namespace gnpde

extern "C" int gnpde_linear(const float* x, int32_t n, int32_t d, int32_t ldx, const float* W, int32_t m, int32_t ldw,
                            const float* b, float* out, int32_t ldo, void* stream) {
  return gnpde::launch_linear_any(x, n, d, ldx, W, m, ldw, b, out, ldo, static_cast<hipStream_t>(stream), 0);
}

extern "C" int gnpde_linear_split_supported(const float* x, int64_t n, int32_t d, int32_t ldx, const float* W, int32_t m, int32_t ldw, int32_t split) {
  return gnpde::linear_split_supported(x, n, d, ldx, W, m, ldw, split) ? 1 : 0;
}

extern "C" int gnpde_linear_split(const float* x, int32_t n, int32_t d, int32_t ldx, const float* W, int32_t m, int32_t ldw, const float* b,
                                  float* out_q, float* out_k, int32_t split, void* stream) {
  return gnpde::launch_linear_split(x, n, d, ldx, W, m, ldw, b, out_q, out_k, split, static_cast<hipStream_t>(stream));
}

extern "C" int gnpde_relu_linear(const float* x, int32_t n, int32_t d, int32_t ldx, const float* W, int32_t m, int32_t ldw,
                                 const float* b, float* out, int32_t ldo, void* stream) {
  return gnpde::launch_linear_any(x, n, d, ldx, W, m, ldw, b, out, ldo, static_cast<hipStream_t>(stream), 1);
}
