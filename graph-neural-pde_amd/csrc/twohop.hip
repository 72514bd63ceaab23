// Two-hop densification of the rewiring block, once per TRAINING forward with new_edges = 'k_hop_att' (SURVEY 8f row 3):
//   S = coalesce( A  ++  offdiag(A A) ) / 2        (reference src/block_transformer_rewiring.py:68-86:
//       torch_sparse.spspmm(A, A, coalesced=True) -> remove_self_loops -> cat with A -> / 2 -> torch_sparse.coalesce(op='add')).
// The reference goes through a general SpGEMM, a filter, a concatenation and a sort-based coalesce (four [nnz]-sized COO
// temporaries); round 1 of this package expanded every product into a list and sorted it.  Here one row-wise (Gustavson) kernel
// does the whole expression without materialising the products:
//   * one wavefront owns one output row i at a time (rows handed out by an atomic counter) and a private dense accumulator of n
//     floats plus a bitmap of touched columns (one bit per column, one byte per 2048-column group) in its workspace slab;
//   * the entries of row i of A go in first, then for every stored (i,k) the 64 lanes walk row k of A and add a_ik a_kj into
//     acc[j], j != i, with fire-and-forget L2 atomics (no returns, so nothing waits on memory inside the walk).  Within one
//     instruction the lanes hit distinct columns (row k has distinct columns unless the caller's edge list has duplicates) and a
//     wavefront's atomics to one address retire in program order, so every sum is formed in CSR order: run-to-run identical;
//   * the row is emitted by scanning the group bytes, then the 64 bitmap words of each touched group: popcount + wave prefix sum
//     give the positions, the columns come out ascending (the (row, col) order of coalesce) and atomicExch reads and re-zeroes
//     bitmap and accumulator in the same operation, leaving the slab clean for the wavefront's next row.
// Two passes of the same kernel: structure only (row counts -> exclusive scan = output row pointer; the caller reads the total and
// allocates), then numbers.  Workspace: min(2048, 16 GiB / slab) slabs of ~4.2 n bytes.
#include "common.h"

namespace gnpde {
namespace {

constexpr int kGroupCols = 2048;             // columns per group: 64 bitmap words, one per lane
constexpr size_t kSlabBudget = size_t(4) << 30;   // (was 16 GiB: 1.4 GB at the ogbn-arxiv size either way, 4 GB instead of 17 at 2 M nodes)

struct TwoHopLayout {
  int words;        // bitmap words per slab (multiple of 64)
  int groups;       // = words / 64
  int flag_words;   // group flags, 4 one-byte flags per 32-bit word, padded to a multiple of 64 words
  size_t slab_bytes;
  int waves;
  size_t counter_off, counts_off, slabs_off, total;
};

TwoHopLayout two_hop_layout(int n) {
  TwoHopLayout l;
  l.groups = (n + kGroupCols - 1) / kGroupCols;
  l.words = l.groups * 64;
  l.flag_words = ((l.groups + 3) / 4 + 63) / 64 * 64;
  l.slab_bytes = align_up(size_t(l.words) * 4 + size_t(l.flag_words) * 4 + size_t(l.words) * 32 * 4, 256);
  // slabs = wavefronts in flight: enough to fill the 256 CUs a few times over, inside the byte budget -- the budget wins over
  // the floor (2 M nodes: 8.4-MB slabs; 64 of them would be 0.5 GB, the old floor of 64 x 4.1 n bytes passed 16 GB there)
  size_t w = kSlabBudget / l.slab_bytes;
  if (w > 2048) w = 2048;
  if (w < 1) w = 1;
  if (w > size_t(n)) w = size_t(n > 0 ? n : 1);
  l.waves = int(w);
  l.counter_off = 0;
  l.counts_off = 256;
  l.slabs_off = l.counts_off + align_up(size_t(n > 0 ? n : 1) * 4, 256);
  l.total = l.slabs_off + l.slab_bytes * size_t(l.waves);
  return l;
}

struct TwoHopArgs {
  const int* rowptr;
  const int* col;
  const float* w;
  int n;
  unsigned* counter;
  int* counts;                   // pass 1 out
  const long long* out_rowptr;   // pass 2 in
  long long* out_row;
  long long* out_col;
  float* out_w;
  char* slabs;
  size_t slab_bytes;
  int words, flag_words;
};

__device__ __forceinline__ int wave_exclusive_scan(int v, int lane, int* total) {
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  *total = __shfl(incl, 63, 64);
  return incl - v;
}

template <bool NUMERIC>
__global__ __launch_bounds__(64) void two_hop_kernel(TwoHopArgs a) {
  const int lane = threadIdx.x;
  char* slab = a.slabs + a.slab_bytes * blockIdx.x;
  unsigned* bitmap = reinterpret_cast<unsigned*>(slab);
  unsigned* flag_words = bitmap + a.words;
  unsigned char* flags = reinterpret_cast<unsigned char*>(flag_words);
  float* acc = reinterpret_cast<float*>(flag_words + a.flag_words);

  __shared__ int ticket;

  auto touch = [&](int j, float v) {
    atomicOr(&bitmap[j >> 5], 1u << (j & 31));     // result unused: no-return atomic
    flags[j >> 11] = 1;                            // every lane stores the same byte
    if (NUMERIC) unsafeAtomicAdd(&acc[j], v);
  };

  for (;;) {
    // lane 0 draws the ticket, the block (= one wavefront) reads it back through LDS behind a barrier, which also keeps
    // the compiler from threading the other lanes around the draw
    if (lane == 0) ticket = static_cast<int>(atomicAdd(a.counter, 1u));
    __syncthreads();
    const int i = __builtin_amdgcn_readfirstlane(ticket);
    __syncthreads();
    if (i >= a.n) break;
    const int rb = a.rowptr[i], re = a.rowptr[i + 1];
    // the row of A itself
    for (int e = rb + lane; e < re; e += 64) touch(a.col[e], NUMERIC ? a.w[e] : 0.f);
    // row i of A A: for each stored (i, k), row k of A
    for (int e0 = rb; e0 < re; e0 += 64) {
      const int e = e0 + lane;
      int kb = 0, ke = 0;
      float aik = 0.f;
      if (e < re) {
        const int k = a.col[e];
        kb = a.rowptr[k];
        ke = a.rowptr[k + 1];
        if (NUMERIC) aik = a.w[e];
      }
      const int cnt = min(64, re - e0);
      for (int t = 0; t < cnt; ++t) {
        const int fb = __shfl(kb, t, 64), fe = __shfl(ke, t, 64);
        const float av = __shfl(aik, t, 64);
        for (int f = fb + lane; f < fe; f += 64) {
          const int j = a.col[f];
          if (j != i) touch(j, NUMERIC ? av * a.w[f] : 0.f);
        }
      }
    }
    __threadfence();   // every store / atomic of the walk has reached L2 before the row is read back
    // emit (and re-zero): groups ascending, words ascending across the lanes, bits ascending inside a lane
    long long pos = NUMERIC ? a.out_rowptr[i] : 0;
    int count = 0;
    for (int f0 = 0; f0 < a.flag_words; f0 += 64) {
      const unsigned fw = atomicExch(&flag_words[f0 + lane], 0u);
      unsigned long long live = __ballot(fw != 0);
      while (live) {
        const int src = __ffsll(static_cast<long long>(live)) - 1;
        live &= live - 1;
        unsigned four = __shfl(fw, src, 64);
        while (four) {
          const int byte = (__ffs(static_cast<int>(four)) - 1) >> 3;
          four &= ~(0xffu << (byte * 8));
          const int group = (f0 + src) * 4 + byte;
          unsigned word = atomicExch(&bitmap[group * 64 + lane], 0u);
          int total;
          const int before = wave_exclusive_scan(__popc(word), lane, &total);
          if (NUMERIC) {
            long long p = pos + count + before;
            while (word) {
              const int bit = __ffs(static_cast<int>(word)) - 1;
              word &= word - 1;
              const int j = (group * 64 + lane) * 32 + bit;
              const float v = atomicExch(&acc[j], 0.f);
              a.out_row[p] = i;
              a.out_col[p] = j;
              a.out_w[p] = 0.5f * v;
              ++p;
            }
          }
          count += total;
        }
      }
    }
    if (!NUMERIC) a.counts[i] = count;   // same value from every lane
  }
}

// out[0] = 0, out[i + 1] = counts[0] + ... + counts[i]   (one block)
__global__ __launch_bounds__(kBlock) void row_pointer_kernel(const int* __restrict__ counts, int n, long long* __restrict__ out) {
  __shared__ long long buf[kBlock];
  __shared__ long long carry;
  if (threadIdx.x == 0) {
    carry = 0;
    out[0] = 0;
  }
  __syncthreads();
  for (int b0 = 0; b0 < n; b0 += kBlock) {
    const int i = b0 + threadIdx.x;
    buf[threadIdx.x] = i < n ? counts[i] : 0;
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {
      const long long add = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
      __syncthreads();
      buf[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < n) out[i + 1] = carry + buf[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == kBlock - 1) carry += buf[threadIdx.x];
    __syncthreads();
  }
}

int check_two_hop(const int32_t* rowptr, const int32_t* col, int32_t n, void* ws, size_t ws_bytes, const char* who) {
  GNPDE_CHECK_ARG(rowptr && col && n >= 1, GNPDE_EINVAL, "%s: bad arguments", who);
  GNPDE_CHECK_ARG(ws && ws_bytes >= two_hop_layout(n).total, GNPDE_EWS, "%s: workspace too small", who);
  return 0;
}

TwoHopArgs two_hop_args(const int32_t* rowptr, const int32_t* col, const float* w, int32_t n, void* ws, const TwoHopLayout& l) {
  TwoHopArgs a{};
  char* base = static_cast<char*>(ws);
  a.rowptr = rowptr;
  a.col = col;
  a.w = w;
  a.n = n;
  a.counter = reinterpret_cast<unsigned*>(base + l.counter_off);
  a.counts = reinterpret_cast<int*>(base + l.counts_off);
  a.slabs = base + l.slabs_off;
  a.slab_bytes = l.slab_bytes;
  a.words = l.words;
  a.flag_words = l.flag_words;
  return a;
}

}  // namespace
}  // namespace gnpde

using namespace gnpde;

extern "C" size_t gnpde_two_hop_workspace_bytes(int32_t n_nodes) { return two_hop_layout(n_nodes > 0 ? n_nodes : 1).total; }

extern "C" int gnpde_two_hop_count(const int32_t* rowptr, const int32_t* col, int32_t n_nodes, int64_t* out_rowptr, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  if (int rc = check_two_hop(rowptr, col, n_nodes, workspace, workspace_bytes, "two_hop_count")) return rc;
  GNPDE_CHECK_ARG(out_rowptr, GNPDE_EINVAL, "two_hop_count: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const TwoHopLayout l = two_hop_layout(n_nodes);
  TwoHopArgs a = two_hop_args(rowptr, col, nullptr, n_nodes, workspace, l);
  // counter + every slab clean (each wavefront re-zeroes what it touched, row by row)
  GNPDE_HIP(hipMemsetAsync(workspace, 0, l.total, s));
  hipLaunchKernelGGL(two_hop_kernel<false>, dim3(l.waves), dim3(64), 0, s, a);
  GNPDE_LAUNCH_CHECK();
  hipLaunchKernelGGL(row_pointer_kernel, dim3(1), dim3(kBlock), 0, s, a.counts, n_nodes, reinterpret_cast<long long*>(out_rowptr));
  GNPDE_LAUNCH_CHECK();
  return 0;
}

extern "C" int gnpde_two_hop_fill(const int32_t* rowptr, const int32_t* col, const float* w, int32_t n_nodes,
                                  const int64_t* out_rowptr, int64_t* out_edge_index, int64_t out_ld, float* out_weight,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_two_hop(rowptr, col, n_nodes, workspace, workspace_bytes, "two_hop_fill")) return rc;
  GNPDE_CHECK_ARG(w && out_rowptr && out_edge_index && out_weight && out_ld >= 0, GNPDE_EINVAL, "two_hop_fill: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const TwoHopLayout l = two_hop_layout(n_nodes);
  TwoHopArgs a = two_hop_args(rowptr, col, w, n_nodes, workspace, l);
  a.out_rowptr = reinterpret_cast<const long long*>(out_rowptr);
  a.out_row = reinterpret_cast<long long*>(out_edge_index);
  a.out_col = a.out_row + out_ld;
  a.out_w = out_weight;
  GNPDE_HIP(hipMemsetAsync(workspace, 0, l.total, s));
  hipLaunchKernelGGL(two_hop_kernel<true>, dim3(l.waves), dim3(64), 0, s, a);
  GNPDE_LAUNCH_CHECK();
  return 0;
}
