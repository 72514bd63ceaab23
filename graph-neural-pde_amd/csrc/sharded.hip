// Row-partitioned fixed-step solve: one process per GPU, boundary rows exchanged between the ranks once per
// evaluation of f, everything -- pack kernel, the grouped RCCL send/recv, interior rows, boundary rows, all stages of
// all steps -- enqueued by this library and captured per rank in ONE hipGraph per solve.
//
// The reference is single-device (nn.DataParallel replicas in its HPO scripts only, reference src/ray_tune.py:65-66);
// this is the multi-GPU form of the solver loop of src/block_constant.py:57-62 that BASELINE.json's north star names
// ("partitioned across the 8 GPUs of one node with RCCL halo-exchange of boundary node features over xGMI between
// ODE steps").  The first version drove the exchange from Python through torch.distributed (pack launch +
// all_to_all_single + two library calls per evaluation: ~125 us of host time against ~47 us of GPU work per evaluation
// at 8 GPUs).  Here the host issues one hipGraphLaunch per solve.
//
// Per evaluation of f at stage input u ([n_own + n_halo, d] rows: own rows, then the halo rows grouped by owner):
//   main stream : pack (gather own rows the peers need into the send buffer)          -> event
//   comm stream : wait; ncclGroupStart; per peer ncclSend(send slice) / ncclRecv(halo slice of u); ncclGroupEnd -> event
//   main stream : f on the INTERIOR rows (no halo neighbour) while the exchange is in flight; wait event;
//                 f on the BOUNDARY rows (projects the halo rows that just arrived, then attends / aggregates).
// Point-to-point sends use every xGMI link of the rank at once (no ring); nothing is unpacked: the halo region of the
// stage buffer IS the receive buffer.
//
// Two transports for the exchange:
//  * P2P (default): the stage buffers of every rank live in IPC-shared device memory (hipIpcGetMemHandle /
//    hipIpcOpenMemHandle); a push kernel on the side stream stores this rank's boundary rows STRAIGHT INTO the peers'
//    halo regions over xGMI and then raises an epoch flag in each peer's (fine-grained) flag array; the receiving rank's
//    main stream runs a one-block wait kernel in front of its boundary pass.  No library call, no host involvement,
//    nothing but kernel nodes in the graph -- and several ranks can share ONE GPU (processes map each other's memory
//    the same way), which is how tests/test_sharded_gpu.py runs real 2- and 4-rank partitions on a single-GPU box.
//  * RCCL (ncclSend / ncclRecv grouped per evaluation): bound at run time (dlopen / dlsym), so the library has no
//    link-time dependency on it; eager launches only -- capturing the grouped send/recv on a forked stream sends
//    hip::Stream::EndCapture of the HIP runtime bundled with torch 2.10 (7.0.51831) into unbounded recursion
//    (profiles/r02_rccl_capture_segfault.txt).
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstring>
#include <vector>
#include "gnpde.h"
#include "common.h"
#include "rhs.h"

namespace gnpde {
int launch_linear_any(const float* x, int n, int d, int ldx, const float* W, int m, int ldw, const float* b, float* out,
                      int ldo, hipStream_t s, int relu = 0);
int launch_edge_attention_pass(const gnpde_graph_t* g, const gnpde_attention_t* at, int pass, float* w_mean_csr, void* ws,
                               size_t ws_bytes, hipStream_t stream);
namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl(const char* path) {
  if (g_rccl.handle != nullptr) return 0;
  const char* cands[] = {path, "librccl.so.1", "librccl.so"};
  void* h = nullptr;
  for (const char* c : cands) {
    if (c == nullptr || c[0] == 0) continue;
    h = dlopen(c, RTLD_NOW | RTLD_NOLOAD);          // already in the process (e.g. torch's copy)?
    if (h == nullptr) h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
    if (h != nullptr) break;
  }
  GNPDE_CHECK_ARG(h != nullptr, GNPDE_ESTATE, "RCCL could not be loaded (%s)", dlerror());
  Rccl r;
  r.handle = h;
#define GNPDE_SYM(field, name)                                                       \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                     \
  GNPDE_CHECK_ARG(r.field != nullptr, GNPDE_ESTATE, "RCCL symbol %s is missing", name)
  GNPDE_SYM(GetUniqueId, "ncclGetUniqueId");
  GNPDE_SYM(CommInitRank, "ncclCommInitRank");
  GNPDE_SYM(CommDestroy, "ncclCommDestroy");
  GNPDE_SYM(Send, "ncclSend");
  GNPDE_SYM(Recv, "ncclRecv");
  GNPDE_SYM(GroupStart, "ncclGroupStart");
  GNPDE_SYM(GroupEnd, "ncclGroupEnd");
  GNPDE_SYM(GetErrorString, "ncclGetErrorString");
#undef GNPDE_SYM
  g_rccl = r;
  return 0;
}

#define GNPDE_NCCL(call)                                                                         \
  do {                                                                                           \
    ncclResult_t _r = (call);                                                                    \
    if (_r != ncclSuccess) {                                                                     \
      ::gnpde::set_error("%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
      return 1000 + static_cast<int>(_r);                                                        \
    }                                                                                            \
  } while (0)

}  // namespace
}  // namespace gnpde

using namespace gnpde;

struct gnpde_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  hipStream_t stream = nullptr;     // the exchange runs beside the interior rows
};

// IPC-shared exchange memory of one rank.  data: n_buffers stage buffers (coarse-grained: read through L2 by the
// kernels launched after the wait kernel; a kernel boundary invalidates stale lines).  ctl (fine-grained, never cached):
// [0, 128) epoch flags written by the peers, [128] pushes done, [129] waits done, [130] push blocks finished, [131] error.
// Behind them, the all-reduce of the adaptive solver's error norm (p2p_sum_kernel): [132] sums done, [256, 384) the peers' sum epochs,
// [512, 1024) two banks (epoch parity) of one double per rank.
constexpr int kCtlWords = 1024, kCtlPush = 128, kCtlWait = 129, kCtlDone = 130, kCtlErr = 131, kCtlSum = 132, kCtlSumFlags = 256,
              kCtlSumVals = 512, kMaxWorld = 128;

struct gnpde_p2p {
  int rank = 0, world = 1;
  size_t buffer_bytes = 0;
  int n_buffers = 0;
  char* data = nullptr;
  uint32_t* ctl = nullptr;
  std::vector<char*> peer_data;
  std::vector<uint32_t*> peer_ctl;
  hipStream_t stream = nullptr;
};

namespace {

struct PushArgs {
  const float* src;            // local stage buffer
  int ld, d, n_send, rank, world;
  const int32_t* send_idx;     // [n_send] local rows, grouped by destination
  const int32_t* order;        // [n_send] or NULL: the w-th wavefront of the launch copies send slot order[w] (push_order)
  const int32_t* seg;          // [world+1] prefix of the send counts
  float* const* dst;           // [world] base of the SAME stage buffer on every peer
  const long long* dst_row0;   // [world] first halo row of this rank's rows on peer p
  uint32_t* const* peer_ctl;   // [world]
  uint32_t* ctl;               // local
  int w_begin;                 // this launch copies the slots order[w_begin .. w_begin + n_send) (a chunk of the walk)
  int publish;                 // 1: the last block raises the epoch flags (the last chunk of an evaluation's push), 0: rows only
  unsigned long long* stamp_start;   // NULL or where block 0 stamps the wall clock when the push starts
  unsigned long long* stamp_end;     // NULL or where the publishing block stamps it after the flags are up
};

// One wavefront per row: copy it into the owner-side halo slot on the peer (xGMI stores), then -- last block -- publish
// the new epoch in every peer's flag array (system-scope release after all row stores of all blocks).
__global__ __launch_bounds__(kBlock) void push_rows_kernel(const PushArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * kWavesPerBlock + static_cast<int>(threadIdx.x >> 6));
  if (a.stamp_start != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *a.stamp_start = wall_clock64();
  if (w < a.n_send) {
    // The send list is grouped by destination.  Walked in that order, everything in flight at any moment targets ONE peer:
    // one xGMI link carries the whole push while the other six idle, and the exchange takes the SUM of the per-link times.
    // order[] interleaves the destinations (in proportion to their row counts, so that all links finish together).
    const int i = a.order != nullptr ? __builtin_amdgcn_readfirstlane(a.order[a.w_begin + w]) : a.w_begin + w;
    int p = 0;
    while (p + 1 < a.world && i >= a.seg[p + 1]) ++p;
    const float* src = a.src + static_cast<size_t>(a.send_idx != nullptr ? a.send_idx[i] : i) * a.ld;     // (NULL: the rows in list order)
    float* dst = a.dst[p] + static_cast<size_t>(a.dst_row0[p] + (i - a.seg[p])) * a.ld;
    if ((a.d & 3) == 0 && (a.ld & 3) == 0) {
      for (int c = lane * 4; c < a.d; c += kWave * 4)
        *reinterpret_cast<float4*>(dst + c) = *reinterpret_cast<const float4*>(src + c);
    } else {
      for (int c = lane; c < a.d; c += kWave) dst[c] = src[c];
    }
  }
  __threadfence_system();
  if (!a.publish) return;      // rows only: a later launch on the same stream publishes (its blocks start after these stores)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = __hip_atomic_fetch_add(a.ctl + kCtlDone, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (done == gridDim.x - 1) {
      __hip_atomic_store(a.ctl + kCtlDone, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned e = __hip_atomic_load(a.ctl + kCtlPush, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
      __hip_atomic_store(a.ctl + kCtlPush, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      for (int p = 0; p < a.world; ++p)
        if (p != a.rank) __hip_atomic_store(a.peer_ctl[p] + a.rank, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      if (a.stamp_end != nullptr) *a.stamp_end = wall_clock64();
    }
  }
}

// In front of the boundary pass: wait until every peer has published the epoch of THIS evaluation.  Bounded spin: a
// peer that never arrives sets the error word instead of hanging the GPU (gnpde_sharded_solver_status).
__global__ __launch_bounds__(kMaxWorld) void wait_flags_kernel(uint32_t* ctl, int rank, int world, long long max_spins,
                                                               unsigned long long* stamp) {
  __shared__ unsigned expect;
  if (stamp != nullptr && threadIdx.x == 0) stamp[0] = wall_clock64();   // the main stream arrives (interior pass done)
  if (threadIdx.x == 0) {
    expect = __hip_atomic_load(ctl + kCtlWait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    __hip_atomic_store(ctl + kCtlWait, expect, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const int p = threadIdx.x;
  // once a wait has timed out the solve is lost (gnpde_sharded_solver_status reports it): the remaining waits of the queued
  // evaluations return at once instead of each spinning to the bound -- a K-step graph then ends in milliseconds, not minutes
  const bool lost = __hip_atomic_load(ctl + kCtlErr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
  if (p < world && p != rank && !lost) {
    long long n = 0;
    // (signed distance: epochs wrap after 2^32 evaluations)
    while (static_cast<int>(__hip_atomic_load(ctl + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - expect) < 0) {
      __builtin_amdgcn_s_sleep(8);
      if (++n > max_spins) {
        __hip_atomic_store(ctl + kCtlErr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __threadfence_system();
  __syncthreads();
  if (stamp != nullptr && threadIdx.x == 0) stamp[1] = wall_clock64();   // every peer's rows have landed
}

// All-reduce (sum) of one double over the ranks, one block: fold this rank's partials (the fold of dopri5.hip's control kernel), store
// the result into bank (epoch & 1) of every peer, raise the epoch in every peer's flag array, wait for every peer's epoch, add the
// bank in RANK order -- every rank adds the same numbers in the same order, so every rank's controller sees the same bits and takes
// the same accept / reject decisions.  A bank is rewritten two sums later: by then every peer has passed the sum in between, which
// needed this rank's contribution to it, which this rank made after reading the bank.  Bounded wait like wait_flags_kernel.
// MAXWORD: the same exchange for the MAXIMUM of one unsigned word per rank (`value` points at it: the order-preserving encoding of
// squareplus' global maximum score, csrc/attention.hip f2ord) -- exact, so no order to keep.
template <bool MAXWORD>
__global__ __launch_bounds__(kBlock) void p2p_sum_kernel(double* __restrict__ value, int n_partials, uint32_t* ctl, uint32_t* const* peer_ctl,
                                                        int rank, int world, long long max_spins) {
  __shared__ double red[kBlock];
  __shared__ unsigned epoch;
  if constexpr (MAXWORD) {
    if (world == 1) return;
    if (threadIdx.x == 0) red[0] = __longlong_as_double(static_cast<long long>(*reinterpret_cast<const uint32_t*>(value)));
    __syncthreads();
  } else {
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_partials; i += kBlock) acc += value[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (world == 1) {
    if (threadIdx.x == 0) value[0] = red[0];
    return;
  }
  }
  if (threadIdx.x == 0) {
    epoch = __hip_atomic_load(ctl + kCtlSum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    __hip_atomic_store(ctl + kCtlSum, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const unsigned e = epoch;
  const int bank = static_cast<int>(e & 1u) * kMaxWorld;
  const unsigned long long mine = static_cast<unsigned long long>(__double_as_longlong(red[0]));
  const int p = threadIdx.x;
  const bool lost = __hip_atomic_load(ctl + kCtlErr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
  if (p < world) {
    unsigned long long* slot = reinterpret_cast<unsigned long long*>(peer_ctl[p] + kCtlSumVals) + bank + rank;
    __hip_atomic_store(slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    if (p != rank) __hip_atomic_store(peer_ctl[p] + kCtlSumFlags + rank, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (p < world && p != rank && !lost) {
    long long n = 0;
    while (static_cast<int>(__hip_atomic_load(ctl + kCtlSumFlags + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
      __builtin_amdgcn_s_sleep(8);
      if (++n > max_spins) {
        __hip_atomic_store(ctl + kCtlErr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long* vals = reinterpret_cast<const unsigned long long*>(ctl + kCtlSumVals) + bank;
    if constexpr (MAXWORD) {
      uint32_t best = 0u;
      for (int q = 0; q < world; ++q) {
        const uint32_t w = static_cast<uint32_t>(__hip_atomic_load(vals + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        best = w > best ? w : best;
      }
      *reinterpret_cast<uint32_t*>(value) = best;
    } else {
      double total = 0.0;
      for (int q = 0; q < world; ++q)
        total += __longlong_as_double(static_cast<long long>(__hip_atomic_load(vals + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)));
      value[0] = total;
    }
  }
}

// ---- statistics of the segments of a normaliser that is not row-local (attention_norm_idx = 1: the segments are the COLUMNS, and a
// column's entries lie on every rank that holds a row pointing at it) between the tables of the attention passes ([n, h] maxima and
// [n, h] sums) and the rows that travel ([.., 2 h]: maxima then sums)
__global__ __launch_bounds__(kBlock) void stats_pack_kernel(const float* __restrict__ m, const float* __restrict__ den, int row0, int n_rows, int h,
                                                           float* __restrict__ out) {
  const long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= static_cast<long long>(n_rows) * 2 * h) return;
  const int i = static_cast<int>(idx / (2 * h)), c = static_cast<int>(idx % (2 * h));
  const size_t o = static_cast<size_t>(row0 + i) * h;
  out[idx] = c < h ? m[o + c] : den[o + c - h];
}
__global__ __launch_bounds__(kBlock) void stats_unpack_kernel(const float* __restrict__ in, int row0, int n_rows, int h, float* __restrict__ m,
                                                             float* __restrict__ den) {
  const long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= static_cast<long long>(n_rows) * 2 * h) return;
  const int i = static_cast<int>(idx / (2 * h)), c = static_cast<int>(idx % (2 * h));
  const size_t o = static_cast<size_t>(row0 + i) * h;
  if (c < h) m[o + c] = in[idx]; else den[o + c - h] = in[idx];
}
// (m, den)[rows[i]] <- (m, den)[rows[i]] (+) in[i]: the arithmetic of stats_merge_kernel (csrc/attention.hip) on travelling rows
__global__ __launch_bounds__(kBlock) void stats_merge_rows_kernel(float* __restrict__ m, float* __restrict__ den, const int* __restrict__ rows,
                                                                 int n_rows, int h, const float* __restrict__ in, int square_plus) {
  const long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= static_cast<long long>(n_rows) * h) return;
  const int i = static_cast<int>(idx / h), head = static_cast<int>(idx % h);
  const size_t o = static_cast<size_t>(rows[i]) * h + head;
  const float m_in = in[static_cast<size_t>(i) * 2 * h + head], den_in = in[static_cast<size_t>(i) * 2 * h + h + head];
  if (square_plus) {
    den[o] = den[o] + den_in;
    return;
  }
  const float ma = m[o], mb = m_in;
  const float mm = fmaxf(ma, mb);
  den[o] = mm == -INFINITY ? den[o] + den_in : den[o] * expf(ma - mm) + den_in * expf(mb - mm);
  m[o] = mm;
}

}  // namespace

struct gnpde_sharded_solver {
  gnpde_comm* comm;
  gnpde_p2p* p2p = nullptr;            // != nullptr: P2P transport, stage buffers are p2p buffers 0..3
  int32_t* d_seg = nullptr;            // [world+1] on device (inside the workspace)
  long long* d_dst_row0 = nullptr;     // [world]
  float** d_dst[4] = {nullptr, nullptr, nullptr, nullptr};   // per stage buffer: [world] peer bases
  uint32_t** d_peer_ctl = nullptr;     // [world]
  int32_t* d_order = nullptr;          // [n_send] destination-interleaved walk of the send list (P2P)
  size_t off_tables = 0;
  gnpde_rhs_t rhs_int, rhs_bnd;
  gnpde_graph_t g_int, g_bnd;
  RhsLayout L_int, L_bnd;
  int method;
  std::vector<float> dts;
  int n_own, n_halo, n_send, d, ld;
  std::vector<int> send_counts, recv_counts;
  const int32_t* send_idx;
  char* ws;
  size_t off_ua, off_ub, off_uc, off_send, off_rhs;
  hipStream_t cap_stream = nullptr;
  hipEvent_t e_pack = nullptr, e_recv = nullptr;
  hipGraph_t graph_obj = nullptr;
  hipGraphExec_t exec = nullptr;
  float* captured_y = nullptr;
  bool exchanges = false;           // an exchange (and, with P2P, an epoch handshake with every peer) per evaluation
  long long max_spins = 1LL << 24;  // bound of the wait kernel's poll loop (~1 us per poll): a lost peer cannot hang the GPU
  int n_evals = 0;
  // P2P: wall-clock stamps of every evaluation of the last run, [n_evals + 1][4] = push start, epoch published, main stream
  // at the wait, all peers' rows landed (gnpde_sharded_solver_timing); the extra row is the end-of-solve rendezvous
  unsigned long long* d_stamps = nullptr;
  int eval_cursor = 0;              // evaluation being enqueued
  size_t ws_bytes = 0;
  // Boundary pass in chunks with the push of each chunk's rows right behind it (gnpde_sharded_solver_set_boundary_chunks): the
  // rows of the NEXT stage input travel while the remaining boundary rows are still being computed
  std::vector<gnpde_rhs_t> rhs_chunk;
  std::vector<gnpde_graph_t> g_chunk;
  std::vector<RhsLayout> L_chunk;
  std::vector<int> push_chunk_ptr;  // [n_chunks + 1] into the walk order
  // Normalisers that are not row-local (gnpde_sharded_solver_set_general): the evaluation is projection + the three attention passes
  // with their exchanges in between + aggregation, all inside the stream
  struct General {
    bool on = false;
    gnpde_graph_t g_att, g_spmm;
    gnpde_attention_t att;
    float* qk = nullptr; float* w = nullptr; float* stats_send = nullptr;
    char* att_ws = nullptr; size_t att_ws_bytes = 0; char* spmm_ws = nullptr; size_t spmm_ws_bytes = 0;
    size_t off_m = 0, off_den = 0, off_gmax = 0;
    float* table = nullptr;          // this rank's [S: n_local x 2h | in_part: n_send x 2h] inside the shared block
    float* in_part = nullptr;
    char* dev = nullptr;             // device tables below (one allocation)
    int32_t* d_rev_seg = nullptr; long long* d_rev_row0 = nullptr; float** d_rev_dst = nullptr; float** d_s_dst = nullptr;
  } gen;
};

namespace {

// Workspace: [ua | ub | uc | send]  (RCCL transport only; with P2P the stage buffers live in the shared block)
//            [device tables of the P2P transport] [scratch of one evaluation: max of the two passes]
size_t sharded_layout(const gnpde_rhs_t& ri, const gnpde_rhs_t& rb, int method, int n_local, int n_send, int world, bool p2p,
                      gnpde_sharded_solver* s) {
  const size_t state = align_up(static_cast<size_t>(n_local) * ri.ld * 4, 256);
  size_t off = 0;
  size_t ua = 0, ub = 0, uc = 0, send = 0;
  if (!p2p) {
    ua = off; off += state;
    if (method == GNPDE_METHOD_RK4) {
      ub = off; off += state;
      uc = off; off += state;
    }
    send = off; off += align_up(static_cast<size_t>(n_send > 0 ? n_send : 1) * ri.d * 4, 256);
  }
  const size_t tables = off;
  if (p2p) off += align_up(static_cast<size_t>(world + 1) * 4, 256) + 6 * align_up(static_cast<size_t>(world) * 8, 256) +
                  align_up(static_cast<size_t>(n_send > 0 ? n_send : 1) * 4, 256);
  const size_t rhs_off = off;
  const size_t ti = rhs_layout(ri).total, tb = rhs_layout(rb).total;
  off += ti > tb ? ti : tb;
  if (s) {
    s->off_ua = ua; s->off_ub = ub; s->off_uc = uc; s->off_send = send; s->off_rhs = rhs_off; s->off_tables = tables;
  }
  return off;
}

float* stage_buffer(gnpde_sharded_solver* s, int b) {   // 0: y (P2P only), 1: ua, 2: ub, 3: uc
  if (s->p2p) return reinterpret_cast<float*>(s->p2p->data + static_cast<size_t>(b) * s->p2p->buffer_bytes);
  const size_t offs[4] = {0, s->off_ua, s->off_ub, s->off_uc};
  return reinterpret_cast<float*>(s->ws + offs[b]);
}

int buffer_index(gnpde_sharded_solver* s, const float* u) {
  for (int b = 0; b < 4; ++b)
    if (stage_buffer(s, b) == u) return b;
  return -1;
}

// RCCL: pack + grouped send / recv of the halo rows of `u`; leaves e_recv recorded on the comm stream
int enqueue_exchange_rccl(gnpde_sharded_solver* s, float* u, hipStream_t st) {
  gnpde_comm* c = s->comm;
  float* send = reinterpret_cast<float*>(s->ws + s->off_send);
  if (s->n_send > 0) {
    const int rc = gnpde_gather_rows(u, s->ld, s->send_idx, s->n_send, s->d, send, s->d, st);
    if (rc) return rc;
  }
  GNPDE_HIP(hipEventRecord(s->e_pack, st));
  GNPDE_HIP(hipStreamWaitEvent(c->stream, s->e_pack, 0));
  GNPDE_NCCL(g_rccl.GroupStart());
  size_t so = 0, ro = 0;
  for (int p = 0; p < c->world; ++p) {
    const size_t ns = static_cast<size_t>(s->send_counts[p]), nr = static_cast<size_t>(s->recv_counts[p]);
    if (ns > 0) GNPDE_NCCL(g_rccl.Send(send + so * s->d, ns * s->d, ncclFloat, p, c->comm, c->stream));
    if (nr > 0)
      GNPDE_NCCL(g_rccl.Recv(u + (static_cast<size_t>(s->n_own) + ro) * s->ld, nr * s->d, ncclFloat, p, c->comm, c->stream));
    so += ns;
    ro += nr;
  }
  GNPDE_NCCL(g_rccl.GroupEnd());
  GNPDE_HIP(hipEventRecord(s->e_recv, c->stream));
  return 0;
}

// P2P: push kernel on the side stream (rows into the peers' halo regions, then the epoch flags); e_recv = push issued.
// u == nullptr: publish the epoch only (the end-of-solve rendezvous).  [w_begin, w_end): the part of the walk to copy (whole
// list: 0, n_send); publish: raise the epoch flags after it; stamp_row: evaluation whose INPUT this push delivers (timing).
int enqueue_push_p2p(gnpde_sharded_solver* s, float* u, int w_begin, int w_end, bool publish, int stamp_row, bool stamp_start,
                     hipStream_t st) {
  gnpde_p2p* x = s->p2p;
  const int b = u != nullptr ? buffer_index(s, u) : 0;
  GNPDE_CHECK_ARG(b >= 0, GNPDE_ESTATE, "sharded solver: stage input is not a shared stage buffer");
  GNPDE_HIP(hipEventRecord(s->e_pack, st));
  GNPDE_HIP(hipStreamWaitEvent(x->stream, s->e_pack, 0));
  PushArgs a;
  a.src = u; a.ld = s->ld; a.d = s->d; a.n_send = u != nullptr ? w_end - w_begin : 0; a.rank = x->rank; a.world = x->world;
  a.send_idx = s->send_idx; a.order = s->d_order; a.seg = s->d_seg; a.dst = s->d_dst[b]; a.dst_row0 = s->d_dst_row0;
  a.peer_ctl = s->d_peer_ctl; a.ctl = x->ctl;
  a.w_begin = w_begin; a.publish = publish ? 1 : 0;
  unsigned long long* row = (s->d_stamps != nullptr && stamp_row >= 0 && stamp_row <= s->n_evals)
                                ? s->d_stamps + 4 * static_cast<size_t>(stamp_row) : nullptr;
  a.stamp_start = (row != nullptr && stamp_start) ? row : nullptr;
  a.stamp_end = (row != nullptr && publish) ? row + 1 : nullptr;
  if (a.n_send > 0 || publish) {
    const unsigned grid = static_cast<unsigned>(a.n_send > 0 ? (a.n_send + kWavesPerBlock - 1) / kWavesPerBlock : 1);
    hipLaunchKernelGGL(push_rows_kernel, dim3(grid), dim3(kBlock), 0, x->stream, a);
    GNPDE_LAUNCH_CHECK();
  }
  GNPDE_HIP(hipEventRecord(s->e_recv, x->stream));
  return 0;
}

int enqueue_exchange_p2p(gnpde_sharded_solver* s, float* u, hipStream_t st) {
  return enqueue_push_p2p(s, u, 0, s->n_send, true, s->eval_cursor, true, st);
}

// push of the rows of a [.., w] table (list order when idx is NULL) into the peers' copies + the epoch; joins the side stream back
int enqueue_push_table(gnpde_sharded_solver* s, const float* src, int w, int n_send, const int32_t* idx, const int32_t* seg, float* const* dst,
                       const long long* dst_row0, hipStream_t st) {
  gnpde_p2p* x = s->p2p;
  GNPDE_HIP(hipEventRecord(s->e_pack, st));
  GNPDE_HIP(hipStreamWaitEvent(x->stream, s->e_pack, 0));
  PushArgs a;
  a.src = src; a.ld = w; a.d = w; a.n_send = n_send; a.rank = x->rank; a.world = x->world;
  a.send_idx = idx; a.order = nullptr; a.seg = seg; a.dst = dst; a.dst_row0 = dst_row0;
  a.peer_ctl = s->d_peer_ctl; a.ctl = x->ctl; a.w_begin = 0; a.publish = 1; a.stamp_start = nullptr; a.stamp_end = nullptr;
  const unsigned grid = static_cast<unsigned>(n_send > 0 ? (n_send + kWavesPerBlock - 1) / kWavesPerBlock : 1);
  hipLaunchKernelGGL(push_rows_kernel, dim3(grid), dim3(kBlock), 0, x->stream, a);
  GNPDE_LAUNCH_CHECK();
  GNPDE_HIP(hipEventRecord(s->e_recv, x->stream));
  GNPDE_HIP(hipStreamWaitEvent(st, s->e_recv, 0));
  hipLaunchKernelGGL(wait_flags_kernel, dim3(1), dim3(kMaxWorld), 0, st, x->ctl, x->rank, x->world, s->max_spins, nullptr);
  GNPDE_LAUNCH_CHECK();
  return 0;
}

// One evaluation with a normaliser that is not row-local (reference src/function_transformer_attention.py:190-213 with
// attention_norm_idx = 1 and / or squareplus, src/utils.py:179-208), everything inside the stream -- the sequence of
// distributed.NativeBackend.rhs_stage_general, which drove it from Python with torch.distributed exchanges:
//   state rows to the peers; projection of own + halo rows; pass 1 (scores + the local maximum) [squareplus: MAX over the ranks];
//   pass 2 (statistics of every local segment) [over columns: the halo columns' partial statistics to their owners, merged there peer
//   by peer in rank order, the totals back with the state's pattern]; pass 3 (normalise + head mean); aggregation with the stage.
// Every rank issues the same pushes in the same order, so ONE epoch sequence orders all three exchanges.
int enqueue_eval_general(gnpde_sharded_solver* s, float* u, const gnpde_epilogue_t& e, hipStream_t st) {
  auto& G = s->gen;
  gnpde_p2p* x = s->p2p;
  const int n_local = s->n_own + s->n_halo, h = G.att.heads, A = G.att.att_dim;
  const bool exch = s->exchanges;
  if (exch) {
    if (int rc = enqueue_exchange_p2p(s, u, st)) return rc;
    GNPDE_HIP(hipStreamWaitEvent(st, s->e_recv, 0));
    hipLaunchKernelGGL(wait_flags_kernel, dim3(1), dim3(kMaxWorld), 0, st, x->ctl, x->rank, x->world, s->max_spins, nullptr);
    GNPDE_LAUNCH_CHECK();
  }
  const gnpde_rhs_t& r = s->rhs_int;
  if (int rc = launch_linear_any(u, n_local, s->d, s->ld, r.proj_w, r.proj_m, s->d, r.proj_b, G.qk, r.proj_m, st)) return rc;
  gnpde_attention_t at = G.att;
  at.q = G.qk;
  at.k = r.kind == GNPDE_RHS_GAT ? G.qk : G.qk + A;
  at.ldqk = r.proj_m;
  float* m = reinterpret_cast<float*>(G.att_ws + G.off_m);
  float* den = reinterpret_cast<float*>(G.att_ws + G.off_den);
  if (int rc = launch_edge_attention_pass(&G.g_att, &at, 1, nullptr, G.att_ws, G.att_ws_bytes, st)) return rc;
  if (at.square_plus && exch) {
    hipLaunchKernelGGL(p2p_sum_kernel<true>, dim3(1), dim3(kBlock), 0, st, reinterpret_cast<double*>(G.att_ws + G.off_gmax), 1, x->ctl,
                       s->d_peer_ctl, x->rank, x->world, s->max_spins);
    GNPDE_LAUNCH_CHECK();
  }
  if (int rc = launch_edge_attention_pass(&G.g_att, &at, 2, nullptr, G.att_ws, G.att_ws_bytes, st)) return rc;
  if (at.norm_idx == 1 && exch) {
    auto blocks = [](long long items) { return dim3(static_cast<unsigned>(items > 0 ? (items + kBlock - 1) / kBlock : 1)); };
    // (i) partial statistics of the halo columns -> their owners' in_part rows
    if (s->n_halo > 0) {
      hipLaunchKernelGGL(stats_pack_kernel, blocks(static_cast<long long>(s->n_halo) * 2 * h), dim3(kBlock), 0, st, m, den, s->n_own, s->n_halo, h, G.stats_send);
      GNPDE_LAUNCH_CHECK();
    }
    if (int rc = enqueue_push_table(s, G.stats_send, 2 * h, s->n_halo, nullptr, G.d_rev_seg, G.d_rev_dst, G.d_rev_row0, st)) return rc;
    // (ii) merge at the owner, peer by peer in rank order
    int pos = 0;
    for (int p = 0; p < x->world; ++p) {
      const int cnt = s->send_counts[p];
      if (cnt > 0) {
        hipLaunchKernelGGL(stats_merge_rows_kernel, blocks(static_cast<long long>(cnt) * h), dim3(kBlock), 0, st, m, den, s->send_idx + pos, cnt, h,
                           G.in_part + static_cast<size_t>(pos) * 2 * h, at.square_plus);
        GNPDE_LAUNCH_CHECK();
      }
      pos += cnt;
    }
    // (iii) the totals travel back like the state's boundary rows (same send list, same halo rows on the peers)
    if (s->n_own > 0) {
      hipLaunchKernelGGL(stats_pack_kernel, blocks(static_cast<long long>(s->n_own) * 2 * h), dim3(kBlock), 0, st, m, den, 0, s->n_own, h, G.table);
      GNPDE_LAUNCH_CHECK();
    }
    if (int rc = enqueue_push_table(s, G.table, 2 * h, s->n_send, s->send_idx, s->d_seg, G.d_s_dst, s->d_dst_row0, st)) return rc;
    if (s->n_halo > 0) {
      hipLaunchKernelGGL(stats_unpack_kernel, blocks(static_cast<long long>(s->n_halo) * 2 * h), dim3(kBlock), 0, st,
                         G.table + static_cast<size_t>(s->n_own) * 2 * h, s->n_own, s->n_halo, h, m, den);
      GNPDE_LAUNCH_CHECK();
    }
  }
  if (int rc = launch_edge_attention_pass(&G.g_att, &at, 3, G.w, G.att_ws, G.att_ws_bytes, st)) return rc;
  if (int rc = launch_spmm_rhs(&G.g_spmm, G.w, u, s->d, s->ld, &e, nullptr, G.spmm_ws, G.spmm_ws_bytes, st)) return rc;
  ++s->eval_cursor;
  return 0;
}

// exchange + f(u) with the fused stage: interior rows overlap the exchange, boundary rows follow it
int enqueue_eval(gnpde_sharded_solver* s, float* u, gnpde_epilogue_t e, hipStream_t st) {
  if (s->gen.on) return enqueue_eval_general(s, u, e, st);
  char* rws = s->ws + s->off_rhs;
  const bool exch = s->exchanges;
  const bool chunked = s->p2p != nullptr && exch && !s->rhs_chunk.empty();
  const bool first = s->eval_cursor == 0, last = s->eval_cursor == s->n_evals - 1;
  if (exch && (!chunked || first)) {      // (chunked: the input of every later evaluation was pushed during the previous one)
    const int rc = s->p2p ? enqueue_exchange_p2p(s, u, st) : enqueue_exchange_rccl(s, u, st);
    if (rc) return rc;
  }
  if (s->g_int.n > 0) {
    const int rc = enqueue_rhs(s->rhs_int, u, e, rws, s->L_int, st);
    if (rc) return rc;
  }
  if (exch) {
    GNPDE_HIP(hipStreamWaitEvent(st, s->e_recv, 0));
    if (s->p2p) {   // the peers' rows have landed once every peer has published this evaluation's epoch
      hipLaunchKernelGGL(wait_flags_kernel, dim3(1), dim3(kMaxWorld), 0, st, s->p2p->ctl, s->p2p->rank, s->p2p->world,
                         s->max_spins, s->d_stamps != nullptr ? s->d_stamps + 4 * static_cast<size_t>(s->eval_cursor) + 2 : nullptr);
      GNPDE_LAUNCH_CHECK();
    }
  }
  if (chunked) {
    // boundary rows chunk by chunk; behind each chunk its rows of the stage OUTPUT (= the next evaluation's input) go to the
    // peers on the side stream while the next chunk is computed; the last chunk's push publishes the next evaluation's epoch
    float* out = e.out_y;
    const int K = static_cast<int>(s->rhs_chunk.size());
    for (int c = 0; c < K; ++c) {
      if (s->g_chunk[c].n > s->g_chunk[c].row_begin) {
        const int rc = enqueue_rhs(s->rhs_chunk[c], u, e, rws, s->L_chunk[c], st);
        if (rc) return rc;
      }
      if (!last) {
        const int rc = enqueue_push_p2p(s, out, s->push_chunk_ptr[c], s->push_chunk_ptr[c + 1], c == K - 1, s->eval_cursor + 1, c == 0, st);
        if (rc) return rc;
      }
    }
  } else if (s->g_bnd.n > s->g_bnd.row_begin) {
    const int rc = enqueue_rhs(s->rhs_bnd, u, e, rws, s->L_bnd, st);
    if (rc) return rc;
  }
  ++s->eval_cursor;
  return 0;
}

// y_work: the stage buffer that holds the state ([n_own + n_halo, ld]); the caller's y with RCCL, shared buffer 0 with P2P
int enqueue_sharded_solve(gnpde_sharded_solver* s, float* y, hipStream_t st) {
  float* ua = stage_buffer(s, 1);
  gnpde_epilogue_t base = base_epilogue(s->rhs_int);
  if (s->method == GNPDE_METHOD_EULER) {
    float* cur = y;
    float* nxt = ua;
    for (float dt : s->dts) {
      gnpde_epilogue_t e = base;
      e.stage = GNPDE_STAGE_EULER; e.dt = dt; e.y = cur; e.out_y = nxt;
      const int rc = enqueue_eval(s, cur, e, st);
      if (rc) return rc;
      float* t = cur; cur = nxt; nxt = t;
    }
    if (cur != y)
      GNPDE_HIP(hipMemcpyAsync(y, cur, static_cast<size_t>(s->n_own) * s->ld * 4, hipMemcpyDeviceToDevice, st));
    return 0;
  }
  float* ub = stage_buffer(s, 2);
  float* uc = stage_buffer(s, 3);
  for (float dt : s->dts) {   // compact rk4 stages (gnpde.h): stage states from the previous stage inputs
    gnpde_epilogue_t e = base;
    e.dt = dt;
    e.stage = GNPDE_STAGE_RK1C; e.out_y = ua;
    int rc = enqueue_eval(s, y, e, st);
    if (rc) return rc;
    e.stage = GNPDE_STAGE_RK2C; e.y = y; e.out_y = ub;
    rc = enqueue_eval(s, ua, e, st);
    if (rc) return rc;
    e.stage = GNPDE_STAGE_RK3C; e.k1 = ua; e.out_y = uc;
    rc = enqueue_eval(s, ub, e, st);
    if (rc) return rc;
    e.stage = GNPDE_STAGE_RK4C; e.k1 = ub; e.out_y = y;
    rc = enqueue_eval(s, uc, e, st);
    if (rc) return rc;
  }
  return 0;
}

// the whole solve of the caller's y: with P2P the state is copied into / out of the shared stage buffer 0
int enqueue_run(gnpde_sharded_solver* s, float* y, hipStream_t st) {
  s->eval_cursor = 0;
  if (!s->p2p) return enqueue_sharded_solve(s, y, st);
  float* y0 = stage_buffer(s, 0);
  const size_t bytes = static_cast<size_t>(s->n_own) * s->ld * 4;
  GNPDE_HIP(hipMemcpyAsync(y0, y, bytes, hipMemcpyDeviceToDevice, st));
  int rc = enqueue_sharded_solve(s, y0, st);
  if (rc) return rc;
  GNPDE_HIP(hipMemcpyAsync(y, y0, bytes, hipMemcpyDeviceToDevice, st));
  if (s->exchanges) {
    // End-of-solve rendezvous (one more epoch, no rows).  Inside a solve a halo region is rewritten at the earliest two
    // evaluations after it was read, and the per-evaluation handshake keeps every peer within one evaluation of this
    // rank, so no acknowledgement is needed; but the LAST evaluation of a solve and the FIRST of the next one may use the
    // same stage buffer (euler with an odd number of steps), and then a fast rank would push into a halo region its peer
    // is still reading.  After this epoch every peer has finished its last boundary pass.
    rc = enqueue_exchange_p2p(s, nullptr, st);
    if (rc) return rc;
    GNPDE_HIP(hipStreamWaitEvent(st, s->e_recv, 0));
    hipLaunchKernelGGL(wait_flags_kernel, dim3(1), dim3(kMaxWorld), 0, st, s->p2p->ctl, s->p2p->rank, s->p2p->world, s->max_spins,
                       s->d_stamps != nullptr ? s->d_stamps + 4 * static_cast<size_t>(s->eval_cursor) + 2 : nullptr);
    GNPDE_LAUNCH_CHECK();
  }
  return 0;
}

void drop_sharded_graph(gnpde_sharded_solver* s) {
  if (s->exec) { (void)hipGraphExecDestroy(s->exec); s->exec = nullptr; }
  if (s->graph_obj) { (void)hipGraphDestroy(s->graph_obj); s->graph_obj = nullptr; }
  s->captured_y = nullptr;
}

int check_sharded_args(const gnpde_halo_t* h, const gnpde_rhs_t* ri, const gnpde_rhs_t* rb) {
  GNPDE_CHECK_ARG(h && ri && rb, GNPDE_EINVAL, "sharded solver: null argument");
  int rc = check_rhs(ri);
  if (rc) return rc;
  rc = check_rhs(rb);
  if (rc) return rc;
  GNPDE_CHECK_ARG(h->world >= 1 && h->world <= kMaxWorld && h->n_own >= 0 && h->n_halo >= 0 && h->send_counts && h->recv_counts,
                  GNPDE_EINVAL, "sharded solver: bad halo description");
  GNPDE_CHECK_ARG(ri->d == rb->d && ri->ld == rb->ld && ri->kind == rb->kind, GNPDE_EINVAL,
                  "sharded solver: interior / boundary descriptors disagree");
  GNPDE_CHECK_ARG(ri->ld == ri->d, GNPDE_ESHAPE, "sharded solver: the state must be dense (ld == d): halo rows are received in place");
  GNPDE_CHECK_ARG(ri->graph->n <= h->n_own && rb->graph->n == h->n_own && rb->graph->row_begin == ri->graph->n, GNPDE_EINVAL,
                  "sharded solver: interior rows [0,%d) / boundary rows [%d,%d) do not tile the %d owned rows", ri->graph->n,
                  rb->graph->row_begin, rb->graph->n, h->n_own);
  return 0;
}

long long sum_counts(const int32_t* c, int world) {
  long long t = 0;
  for (int p = 0; p < world; ++p) t += c[p];
  return t;
}

// common part of the two constructors
int create_common(gnpde_sharded_solver** out, gnpde_comm* comm, gnpde_p2p* p2p, const gnpde_halo_t* halo,
                  const gnpde_rhs_t* rhs_interior, const gnpde_rhs_t* rhs_boundary, int method, const float* dts, int n_steps,
                  void* workspace, size_t workspace_bytes) {
  GNPDE_CHECK_ARG(out != nullptr, GNPDE_EINVAL, "sharded_solver_create: out is null");
  *out = nullptr;
  int rc = check_sharded_args(halo, rhs_interior, rhs_boundary);
  if (rc) return rc;
  GNPDE_CHECK_ARG(method == GNPDE_METHOD_EULER || method == GNPDE_METHOD_RK4, GNPDE_EINVAL, "sharded_solver_create: bad method %d", method);
  GNPDE_CHECK_ARG(n_steps >= 0 && (dts || n_steps == 0), GNPDE_EINVAL, "sharded_solver_create: bad time grid");
  for (int p = 0; p < halo->world; ++p)
    GNPDE_CHECK_ARG(halo->send_counts[p] >= 0 && halo->recv_counts[p] >= 0, GNPDE_EINVAL, "sharded_solver_create: negative count");
  const long long n_send = sum_counts(halo->send_counts, halo->world), n_recv = sum_counts(halo->recv_counts, halo->world);
  GNPDE_CHECK_ARG(n_recv == halo->n_halo, GNPDE_EINVAL, "sharded_solver_create: recv counts sum to %lld, halo has %d rows", n_recv, halo->n_halo);
  GNPDE_CHECK_ARG(n_send == 0 || halo->send_idx != nullptr, GNPDE_EINVAL, "sharded_solver_create: send_idx is null");
  gnpde_sharded_solver* s = new gnpde_sharded_solver();
  s->comm = comm;
  s->p2p = p2p;
  // P2P: every rank signals every peer each evaluation (also with nothing to send), so world > 1 always "exchanges"
  s->exchanges = p2p ? halo->world > 1 : (n_send > 0 || n_recv > 0);
  s->rhs_int = *rhs_interior;
  s->rhs_bnd = *rhs_boundary;
  // (row-range views: the column normaliser keeps its three exchangeable passes, never the transposed-graph form)
  s->rhs_int.att.graph_t = nullptr; s->rhs_int.att.t_from_csr = nullptr;
  s->rhs_bnd.att.graph_t = nullptr; s->rhs_bnd.att.t_from_csr = nullptr;
  s->g_int = *rhs_interior->graph;
  s->g_bnd = *rhs_boundary->graph;
  s->rhs_int.graph = &s->g_int;
  s->rhs_bnd.graph = &s->g_bnd;
  s->L_int = rhs_layout(s->rhs_int);
  s->L_bnd = rhs_layout(s->rhs_bnd);
  s->method = method;
  s->dts.assign(dts, dts + n_steps);
  s->n_own = halo->n_own;
  s->n_halo = halo->n_halo;
  s->n_send = static_cast<int>(n_send);
  s->d = rhs_interior->d;
  s->ld = rhs_interior->ld;
  s->send_counts.assign(halo->send_counts, halo->send_counts + halo->world);
  s->recv_counts.assign(halo->recv_counts, halo->recv_counts + halo->world);
  s->send_idx = halo->send_idx;
  const size_t need = sharded_layout(s->rhs_int, s->rhs_bnd, method, s->n_own + s->n_halo, s->n_send, halo->world, p2p != nullptr, s);
  if (!(workspace && workspace_bytes >= need && reinterpret_cast<uintptr_t>(workspace) % 256 == 0)) {
    set_error("sharded_solver_create: workspace %zu bytes (need %zu, 256-byte aligned)", workspace_bytes, need);
    delete s;
    return GNPDE_EWS;
  }
  s->ws = static_cast<char*>(workspace);
  s->ws_bytes = workspace_bytes;
  s->n_evals = n_steps * (method == GNPDE_METHOD_RK4 ? 4 : 1);
  hipError_t e = hipEventCreateWithFlags(&s->e_pack, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->e_recv, hipEventDisableTiming);
  if (e != hipSuccess) {
    set_error("hipEventCreate failed: %s", hipGetErrorString(e));
    gnpde_sharded_solver_destroy(s);
    return static_cast<int>(e);
  }
  *out = s;
  return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ RCCL communicator
extern "C" int gnpde_comm_load_library(const char* path) { return load_rccl(path); }

extern "C" int gnpde_comm_get_unique_id(void* id_out) {
  GNPDE_CHECK_ARG(id_out != nullptr, GNPDE_EINVAL, "comm_get_unique_id: null output");
  static_assert(sizeof(ncclUniqueId) <= GNPDE_COMM_ID_BYTES, "unique id does not fit");
  int rc = load_rccl(nullptr);
  if (rc) return rc;
  ncclUniqueId id;
  GNPDE_NCCL(g_rccl.GetUniqueId(&id));
  std::memset(id_out, 0, GNPDE_COMM_ID_BYTES);
  std::memcpy(id_out, &id, sizeof(id));
  return 0;
}

extern "C" int gnpde_comm_create(gnpde_comm_t** out, const void* id, int32_t rank, int32_t world) {
  GNPDE_CHECK_ARG(out && id && world >= 1 && rank >= 0 && rank < world, GNPDE_EINVAL, "comm_create: bad arguments");
  *out = nullptr;
  int rc = load_rccl(nullptr);
  if (rc) return rc;
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  gnpde_comm* c = new gnpde_comm();
  c->rank = rank;
  c->world = world;
  const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, uid, rank);   // collective over the ranks
  if (r != ncclSuccess) {
    set_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    delete c;
    return 1000 + static_cast<int>(r);
  }
  const hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
    (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return static_cast<int>(e);
  }
  *out = c;
  return 0;
}

extern "C" int gnpde_comm_destroy(gnpde_comm_t* c) {
  if (!c) return 0;
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  delete c;
  return 0;
}

// ------------------------------------------------------------------------------------------------ push order
// order[w] = slot of the destination-grouped send list that the w-th wavefront of the push copies: the slots of all
// destinations merged by their relative position (j + 1/2) / count_p inside their own segment (ties: lower rank first), so
// that at any moment the rows in flight go to ALL peers in proportion to what each is owed and every link finishes at the
// same time.  A permutation of [0, sum counts); each destination's own rows keep their order.  Host only.
extern "C" int gnpde_push_order(const int32_t* send_counts, int32_t world, int32_t* order) {
  GNPDE_CHECK_ARG(send_counts && order && world >= 1 && world <= kMaxWorld, GNPDE_EINVAL, "push_order: bad arguments");
  std::vector<long long> next(world, 0), base(world, 0);
  long long total = 0;
  for (int p = 0; p < world; ++p) {
    GNPDE_CHECK_ARG(send_counts[p] >= 0, GNPDE_EINVAL, "push_order: negative count");
    base[p] = total;
    total += send_counts[p];
  }
  for (long long w = 0; w < total; ++w) {
    int best = -1;
    for (int p = 0; p < world; ++p) {
      if (next[p] >= send_counts[p]) continue;
      if (best < 0) { best = p; continue; }
      // (2 j_p + 1) / count_p < (2 j_best + 1) / count_best, in integers
      const long long lhs = (2 * next[p] + 1) * static_cast<long long>(send_counts[best]);
      const long long rhs = (2 * next[best] + 1) * static_cast<long long>(send_counts[p]);
      if (lhs < rhs) best = p;
    }
    order[w] = static_cast<int32_t>(base[best] + next[best]);
    ++next[best];
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ P2P shared memory
extern "C" int gnpde_p2p_create(gnpde_p2p_t** out, int32_t rank, int32_t world, size_t buffer_bytes, int32_t n_buffers) {
  GNPDE_CHECK_ARG(out && world >= 1 && world <= kMaxWorld && rank >= 0 && rank < world && n_buffers >= 1 && n_buffers <= 5 &&
                  buffer_bytes > 0, GNPDE_EINVAL, "p2p_create: bad arguments");
  *out = nullptr;
  static_assert(2 * sizeof(hipIpcMemHandle_t) <= GNPDE_P2P_HANDLE_BYTES, "IPC handles do not fit");
  gnpde_p2p* x = new gnpde_p2p();
  x->rank = rank;
  x->world = world;
  x->buffer_bytes = align_up(buffer_bytes, 256);
  x->n_buffers = n_buffers;
  x->peer_data.assign(world, nullptr);
  x->peer_ctl.assign(world, nullptr);
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&x->data), x->buffer_bytes * n_buffers);
  // flags other devices / processes write while a kernel of this rank polls them: fine-grained (never cached in L2)
  if (e == hipSuccess) e = hipExtMallocWithFlags(reinterpret_cast<void**>(&x->ctl), kCtlWords * sizeof(uint32_t), hipDeviceMallocFinegrained);
  if (e == hipSuccess) e = hipMemset(x->data, 0, x->buffer_bytes * n_buffers);
  if (e == hipSuccess) e = hipMemset(x->ctl, 0, kCtlWords * sizeof(uint32_t));
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    set_error("p2p_create failed: %s", hipGetErrorString(e));
    gnpde_p2p_destroy(x);
    return static_cast<int>(e);
  }
  x->peer_data[rank] = x->data;
  x->peer_ctl[rank] = x->ctl;
  *out = x;
  return 0;
}

extern "C" int gnpde_p2p_get_handle(gnpde_p2p_t* x, void* handle_out) {
  GNPDE_CHECK_ARG(x && handle_out, GNPDE_EINVAL, "p2p_get_handle: null argument");
  hipIpcMemHandle_t h[2];
  GNPDE_HIP(hipIpcGetMemHandle(&h[0], x->data));
  GNPDE_HIP(hipIpcGetMemHandle(&h[1], x->ctl));
  std::memset(handle_out, 0, GNPDE_P2P_HANDLE_BYTES);
  std::memcpy(handle_out, h, sizeof(h));
  return 0;
}

extern "C" int gnpde_p2p_connect(gnpde_p2p_t* x, const void* handles) {
  GNPDE_CHECK_ARG(x && handles, GNPDE_EINVAL, "p2p_connect: null argument");
  const char* base = static_cast<const char*>(handles);
  for (int p = 0; p < x->world; ++p) {
    if (p == x->rank || x->peer_data[p] != nullptr) continue;
    hipIpcMemHandle_t h[2];
    std::memcpy(h, base + static_cast<size_t>(p) * GNPDE_P2P_HANDLE_BYTES, sizeof(h));
    void* pd = nullptr;
    void* pc = nullptr;
    GNPDE_HIP(hipIpcOpenMemHandle(&pd, h[0], hipIpcMemLazyEnablePeerAccess));
    GNPDE_HIP(hipIpcOpenMemHandle(&pc, h[1], hipIpcMemLazyEnablePeerAccess));
    x->peer_data[p] = static_cast<char*>(pd);
    x->peer_ctl[p] = static_cast<uint32_t*>(pc);
  }
  return 0;
}

extern "C" void* gnpde_p2p_buffer(gnpde_p2p_t* x, int32_t b) {
  if (!x || b < 0 || b >= x->n_buffers) return nullptr;
  return x->data + static_cast<size_t>(b) * x->buffer_bytes;
}

extern "C" int gnpde_p2p_destroy(gnpde_p2p_t* x) {
  if (!x) return 0;
  for (int p = 0; p < x->world; ++p) {
    if (p == x->rank) continue;
    if (x->peer_data[p]) (void)hipIpcCloseMemHandle(x->peer_data[p]);
    if (x->peer_ctl[p]) (void)hipIpcCloseMemHandle(x->peer_ctl[p]);
  }
  if (x->stream) (void)hipStreamDestroy(x->stream);
  if (x->data) (void)hipFree(x->data);
  if (x->ctl) (void)hipFree(x->ctl);
  delete x;
  return 0;
}

// ------------------------------------------------------------------------------------------------ solver
extern "C" size_t gnpde_sharded_solver_workspace_bytes(const gnpde_halo_t* halo, const gnpde_rhs_t* rhs_interior,
                                                       const gnpde_rhs_t* rhs_boundary, int32_t method, int32_t p2p) {
  if (check_sharded_args(halo, rhs_interior, rhs_boundary)) return 0;
  if (method != GNPDE_METHOD_EULER && method != GNPDE_METHOD_RK4) return 0;
  return sharded_layout(*rhs_interior, *rhs_boundary, method, halo->n_own + halo->n_halo,
                        static_cast<int>(sum_counts(halo->send_counts, halo->world)), halo->world, p2p != 0, nullptr);
}

extern "C" int gnpde_sharded_solver_create(gnpde_sharded_solver_t** out, gnpde_comm_t* comm, const gnpde_halo_t* halo,
                                           const gnpde_rhs_t* rhs_interior, const gnpde_rhs_t* rhs_boundary, int32_t method,
                                           const float* dts, int32_t n_steps, void* workspace, size_t workspace_bytes) {
  if (halo && halo->send_counts && halo->recv_counts && halo->world >= 1 && halo->world <= kMaxWorld) {
    const bool exch = sum_counts(halo->send_counts, halo->world) > 0 || sum_counts(halo->recv_counts, halo->world) > 0;
    GNPDE_CHECK_ARG(!exch || (comm != nullptr && comm->world == halo->world && comm->rank == halo->rank), GNPDE_EINVAL,
                    "sharded_solver_create: communicator does not match the halo description");
  }
  return create_common(out, comm, nullptr, halo, rhs_interior, rhs_boundary, method, dts, n_steps, workspace, workspace_bytes);
}

extern "C" int gnpde_sharded_solver_create_p2p(gnpde_sharded_solver_t** out, gnpde_p2p_t* p2p, const gnpde_halo_t* halo,
                                               const gnpde_rhs_t* rhs_interior, const gnpde_rhs_t* rhs_boundary, int32_t method,
                                               const float* dts, int32_t n_steps, const int64_t* peer_halo_row0,
                                               const int64_t* peer_buffer_bytes, void* workspace, size_t workspace_bytes) {
  GNPDE_CHECK_ARG(p2p && halo && peer_halo_row0 && peer_buffer_bytes, GNPDE_EINVAL, "sharded_solver_create_p2p: null argument");
  GNPDE_CHECK_ARG(p2p->world == halo->world && p2p->rank == halo->rank, GNPDE_EINVAL, "sharded_solver_create_p2p: rank / world mismatch");
  const int nbuf = method == GNPDE_METHOD_RK4 ? 4 : 2;
  GNPDE_CHECK_ARG(p2p->n_buffers >= nbuf, GNPDE_EINVAL, "sharded_solver_create_p2p: %d shared buffers, need %d", p2p->n_buffers, nbuf);
  const size_t state = static_cast<size_t>(halo->n_own + halo->n_halo) * (rhs_interior ? rhs_interior->ld : 0) * 4;
  GNPDE_CHECK_ARG(p2p->buffer_bytes >= state, GNPDE_EINVAL, "sharded_solver_create_p2p: shared buffers of %zu bytes, state needs %zu",
                  p2p->buffer_bytes, state);
  for (int p = 0; p < p2p->world; ++p)
    GNPDE_CHECK_ARG(p2p->peer_data[p] && p2p->peer_ctl[p], GNPDE_ESTATE, "sharded_solver_create_p2p: peer %d is not connected", p);
  gnpde_sharded_solver* s = nullptr;
  int rc = create_common(&s, nullptr, p2p, halo, rhs_interior, rhs_boundary, method, dts, n_steps, workspace, workspace_bytes);
  if (rc) return rc;
  if (hipMalloc(reinterpret_cast<void**>(&s->d_stamps), (static_cast<size_t>(s->n_evals) + 1) * 4 * sizeof(unsigned long long)) != hipSuccess ||
      hipMemset(s->d_stamps, 0, (static_cast<size_t>(s->n_evals) + 1) * 4 * sizeof(unsigned long long)) != hipSuccess) {
    (void)hipGetLastError();
    if (s->d_stamps) (void)hipFree(s->d_stamps);
    s->d_stamps = nullptr;        // timing is an extra: the solver works without it
  }
  // device tables of the push kernel
  const int W = halo->world;
  char* t = s->ws + s->off_tables;
  const size_t seg_bytes = align_up(static_cast<size_t>(W + 1) * 4, 256), tab = align_up(static_cast<size_t>(W) * 8, 256);
  s->d_seg = reinterpret_cast<int32_t*>(t);
  s->d_dst_row0 = reinterpret_cast<long long*>(t + seg_bytes);
  s->d_peer_ctl = reinterpret_cast<uint32_t**>(t + seg_bytes + tab);
  std::vector<int32_t> seg(W + 1, 0);
  std::vector<long long> row0(W);
  std::vector<uint32_t*> pctl(W);
  for (int p = 0; p < W; ++p) {
    seg[p + 1] = seg[p] + halo->send_counts[p];
    row0[p] = peer_halo_row0[p];
    pctl[p] = p2p->peer_ctl[p];
  }
  hipError_t e = hipMemcpy(s->d_seg, seg.data(), (W + 1) * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(s->d_dst_row0, row0.data(), W * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(s->d_peer_ctl, pctl.data(), W * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess && s->n_send > 0) {
    s->d_order = reinterpret_cast<int32_t*>(t + seg_bytes + 6 * tab);
    std::vector<int32_t> order(static_cast<size_t>(s->n_send));
    const int orc = gnpde_push_order(halo->send_counts, W, order.data());
    if (orc != 0) {
      gnpde_sharded_solver_destroy(s);
      return orc;
    }
    e = hipMemcpy(s->d_order, order.data(), static_cast<size_t>(s->n_send) * 4, hipMemcpyHostToDevice);
  }
  for (int b = 0; b < 4 && e == hipSuccess; ++b) {
    s->d_dst[b] = reinterpret_cast<float**>(t + seg_bytes + (2 + b) * tab);
    std::vector<float*> dst(W);
    for (int p = 0; p < W; ++p)
      dst[p] = reinterpret_cast<float*>(p2p->peer_data[p] + static_cast<size_t>(b) * static_cast<size_t>(peer_buffer_bytes[p]));
    e = hipMemcpy(s->d_dst[b], dst.data(), W * 8, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    set_error("sharded_solver_create_p2p: table upload failed: %s", hipGetErrorString(e));
    gnpde_sharded_solver_destroy(s);
    return static_cast<int>(e);
  }
  *out = s;
  return 0;
}

extern "C" int gnpde_sharded_solver_set_general(gnpde_sharded_solver_t* s, const gnpde_general_t* g) {
  GNPDE_CHECK_ARG(s != nullptr && g != nullptr, GNPDE_EINVAL, "sharded_solver_set_general: null argument");
  GNPDE_CHECK_ARG(s->p2p != nullptr, GNPDE_ESTATE, "sharded_solver_set_general: the P2P transport only");
  GNPDE_CHECK_ARG(s->rhs_chunk.empty(), GNPDE_ESTATE, "sharded_solver_set_general: no chunked boundary pass");
  GNPDE_CHECK_ARG(g->att_graph && g->spmm_graph && g->att && g->qk && g->w && g->att_ws && g->peer_in_offset && g->peer_rev_row0 &&
                  g->peer_buffer_bytes,
                  GNPDE_EINVAL, "sharded_solver_set_general: null member");
  GNPDE_CHECK_ARG(s->rhs_int.kind == GNPDE_RHS_TRANSFORMER || s->rhs_int.kind == GNPDE_RHS_GAT, GNPDE_EINVAL,
                  "sharded_solver_set_general: an attention function");
  const int n_local = s->n_own + s->n_halo, W = s->p2p->world, h = g->att->heads;
  GNPDE_CHECK_ARG(g->att_graph->n == n_local && g->spmm_graph->n == s->n_own && g->spmm_graph->row_begin == 0, GNPDE_EINVAL,
                  "sharded_solver_set_general: the attention graph has every local node as a segment, the aggregation graph the owned rows");
  GNPDE_CHECK_ARG(h >= 1 && g->att->att_dim % h == 0 && (g->att->norm_idx == 0 || g->att->norm_idx == 1), GNPDE_EINVAL,
                  "sharded_solver_set_general: bad attention description");
  const bool columns = g->att->norm_idx == 1 && W > 1;
  GNPDE_CHECK_ARG(!columns || (g->stats_send != nullptr && g->stats_buffer >= 0 && g->stats_buffer < s->p2p->n_buffers && g->stats_buffer >= 4),
                  GNPDE_EINVAL, "sharded_solver_set_general: the column statistics need a shared buffer behind the four stage buffers");
  const size_t row_bytes = static_cast<size_t>(2 * h) * 4;
  const size_t in_off = static_cast<size_t>(g->in_offset);
  GNPDE_CHECK_ARG(!columns || (in_off % 256 == 0 && in_off >= static_cast<size_t>(n_local) * row_bytes &&
                               in_off + static_cast<size_t>(s->n_send) * row_bytes <= s->p2p->buffer_bytes),
                  GNPDE_EWS, "sharded_solver_set_general: [S | in_part] does not fit the shared buffer (%zu + %zu of %zu bytes)", in_off,
                  static_cast<size_t>(s->n_send) * row_bytes, s->p2p->buffer_bytes);
  const gnpde::AttLayoutView lv = gnpde::att_layout_view(g->att_graph, g->att);
  GNPDE_CHECK_ARG(g->att_ws_bytes >= lv.total, GNPDE_EWS, "sharded_solver_set_general: attention workspace %zu < %zu bytes", g->att_ws_bytes, lv.total);
  drop_sharded_graph(s);
  auto& G = s->gen;
  if (G.dev) { (void)hipFree(G.dev); G.dev = nullptr; }
  G.g_att = *g->att_graph; G.g_spmm = *g->spmm_graph; G.att = *g->att;
  G.att.graph_t = nullptr; G.att.t_from_csr = nullptr;
  G.qk = g->qk; G.w = g->w; G.stats_send = g->stats_send;
  G.att_ws = static_cast<char*>(g->att_ws); G.att_ws_bytes = g->att_ws_bytes;
  G.spmm_ws = static_cast<char*>(g->spmm_ws); G.spmm_ws_bytes = g->spmm_ws_bytes;
  G.off_m = lv.seg_m; G.off_den = lv.seg_den; G.off_gmax = lv.gmax;
  if (columns) {
    G.table = reinterpret_cast<float*>(s->p2p->data + static_cast<size_t>(g->stats_buffer) * s->p2p->buffer_bytes);
    G.in_part = reinterpret_cast<float*>(reinterpret_cast<char*>(G.table) + in_off);
    const size_t seg_b = align_up(static_cast<size_t>(W + 1) * 4, 256), tab = align_up(static_cast<size_t>(W) * 8, 256);
    if (hipMalloc(reinterpret_cast<void**>(&G.dev), seg_b + 3 * tab) != hipSuccess) {
      (void)hipGetLastError();
      set_error("sharded_solver_set_general: table allocation failed");
      return GNPDE_EINVAL;
    }
    G.d_rev_seg = reinterpret_cast<int32_t*>(G.dev);
    G.d_rev_row0 = reinterpret_cast<long long*>(G.dev + seg_b);
    G.d_rev_dst = reinterpret_cast<float**>(G.dev + seg_b + tab);
    G.d_s_dst = reinterpret_cast<float**>(G.dev + seg_b + 2 * tab);
    std::vector<int32_t> seg(W + 1, 0);
    std::vector<long long> row0(W);
    std::vector<float*> rdst(W), sdst(W);
    for (int p = 0; p < W; ++p) {
      seg[p + 1] = seg[p] + s->recv_counts[p];
      row0[p] = g->peer_rev_row0[p];
      char* base = s->p2p->peer_data[p] + static_cast<size_t>(g->stats_buffer) * static_cast<size_t>(g->peer_buffer_bytes[p]);
      sdst[p] = reinterpret_cast<float*>(base);
      rdst[p] = reinterpret_cast<float*>(base + static_cast<size_t>(g->peer_in_offset[p]));
    }
    hipError_t e = hipMemcpy(G.d_rev_seg, seg.data(), (W + 1) * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(G.d_rev_row0, row0.data(), W * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(G.d_rev_dst, rdst.data(), W * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(G.d_s_dst, sdst.data(), W * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      set_error("sharded_solver_set_general: table upload failed: %s", hipGetErrorString(e));
      return static_cast<int>(e);
    }
  }
  G.on = true;
  return 0;
}

extern "C" int gnpde_sharded_solver_run(gnpde_sharded_solver_t* s, float* y, int32_t use_graph, void* stream) {
  GNPDE_CHECK_ARG(s && y, GNPDE_EINVAL, "sharded_solver_run: null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!use_graph) return enqueue_run(s, y, st);
  GNPDE_CHECK_ARG(s->p2p != nullptr || !s->exchanges, GNPDE_ESTATE,
                  "sharded_solver_run: the RCCL transport cannot be captured with this HIP runtime (hip::Stream::EndCapture "
                  "recurses without bound on the forked send/recv stream); run it eagerly or use the P2P transport");
  if (s->exec == nullptr || s->captured_y != y) {
    drop_sharded_graph(s);
    if (s->cap_stream == nullptr) GNPDE_HIP(hipStreamCreateWithFlags(&s->cap_stream, hipStreamNonBlocking));
    GNPDE_HIP(hipStreamBeginCapture(s->cap_stream, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue_run(s, y, s->cap_stream);
    hipGraph_t gobj = nullptr;
    const hipError_t ec = hipStreamEndCapture(s->cap_stream, &gobj);
    if (rc != 0) {
      if (gobj) (void)hipGraphDestroy(gobj);
      return rc;
    }
    if (ec != hipSuccess) {
      set_error("hipStreamEndCapture failed: %s", hipGetErrorString(ec));
      return static_cast<int>(ec);
    }
    s->graph_obj = gobj;
    GNPDE_HIP(hipGraphInstantiate(&s->exec, s->graph_obj, nullptr, nullptr, 0));
    s->captured_y = y;
  }
  GNPDE_HIP(hipGraphLaunch(s->exec, st));
  return 0;
}

extern "C" int gnpde_sharded_solver_status(gnpde_sharded_solver_t* s, int32_t* timed_out, int64_t* epochs) {
  GNPDE_CHECK_ARG(s != nullptr, GNPDE_EINVAL, "sharded_solver_status: null solver");
  uint32_t w[4] = {0, 0, 0, 0};
  if (s->p2p) GNPDE_HIP(hipMemcpy(w, s->p2p->ctl + kCtlPush, sizeof(w), hipMemcpyDeviceToHost));   // synchronises
  if (timed_out) *timed_out = static_cast<int32_t>(w[kCtlErr - kCtlPush]);
  if (epochs) *epochs = static_cast<int64_t>(w[0]);
  return 0;
}

namespace {
// the proportional whole-list walk of the send slots (what create() installs): restored whenever the chunked order goes away
int restore_default_push_order(gnpde_sharded_solver* s) {
  if (s->p2p == nullptr || s->n_send == 0 || s->d_order == nullptr) return 0;
  std::vector<int32_t> order(static_cast<size_t>(s->n_send));
  const int rc = gnpde_push_order(s->send_counts.data(), static_cast<int32_t>(s->send_counts.size()), order.data());
  if (rc) return rc;
  GNPDE_HIP(hipMemcpy(s->d_order, order.data(), static_cast<size_t>(s->n_send) * 4, hipMemcpyHostToDevice));
  return 0;
}
}  // namespace

extern "C" int gnpde_sharded_solver_set_boundary_chunks(gnpde_sharded_solver_t* s, const gnpde_rhs_t* const* rhs_chunks,
                                                        int32_t n_chunks, const int32_t* push_order,
                                                        const int32_t* push_chunk_ptr) {
  GNPDE_CHECK_ARG(s != nullptr && n_chunks >= 0, GNPDE_EINVAL, "sharded_solver_set_boundary_chunks: bad arguments");
  drop_sharded_graph(s);
  // Whatever happens below, the solver is first put back into the one-pass state (no chunk descriptors, the default push
  // order): a call that fails validation leaves THAT state, never a half-filled chunk list the next run would index past.
  const bool had_chunks = !s->rhs_chunk.empty();
  s->rhs_chunk.clear(); s->g_chunk.clear(); s->L_chunk.clear(); s->push_chunk_ptr.clear();
  if (had_chunks)
    if (int rc = restore_default_push_order(s)) return rc;
  if (n_chunks == 0) return 0;                     // back to one boundary pass + one push per evaluation
  GNPDE_CHECK_ARG(s->p2p != nullptr, GNPDE_ESTATE, "sharded_solver_set_boundary_chunks: P2P transport only");
  GNPDE_CHECK_ARG(rhs_chunks && push_chunk_ptr && (push_order || s->n_send == 0), GNPDE_EINVAL,
                  "sharded_solver_set_boundary_chunks: null argument");
  GNPDE_CHECK_ARG(push_chunk_ptr[0] == 0 && push_chunk_ptr[n_chunks] == s->n_send, GNPDE_EINVAL,
                  "sharded_solver_set_boundary_chunks: the chunks must tile the %d send slots", s->n_send);
  // everything is validated and built in locals; the solver takes it over only when every check has passed
  std::vector<gnpde_rhs_t> rhs_new;
  std::vector<gnpde_graph_t> g_new(static_cast<size_t>(n_chunks));   // (sized first: the descriptors point into it)
  std::vector<RhsLayout> L_new;
  std::vector<int> ptr_new;
  int row = s->g_bnd.row_begin;
  for (int c = 0; c < n_chunks; ++c) {
    GNPDE_CHECK_ARG(rhs_chunks[c] != nullptr && push_chunk_ptr[c] <= push_chunk_ptr[c + 1], GNPDE_EINVAL,
                    "sharded_solver_set_boundary_chunks: bad chunk %d", c);
    if (int rc = check_rhs(rhs_chunks[c])) return rc;
    const gnpde_rhs_t& r = *rhs_chunks[c];
    GNPDE_CHECK_ARG(r.kind == s->rhs_bnd.kind && r.d == s->d && r.ld == s->ld && r.graph->row_begin == row &&
                    r.graph->n >= row && r.graph->n <= s->n_own, GNPDE_EINVAL,
                    "sharded_solver_set_boundary_chunks: chunk %d must continue the boundary rows at row %d", c, row);
    row = r.graph->n;
    g_new[c] = *r.graph;
    rhs_new.push_back(r);
    rhs_new.back().att.graph_t = nullptr; rhs_new.back().att.t_from_csr = nullptr;
    L_new.push_back(rhs_layout(rhs_new.back()));
    GNPDE_CHECK_ARG(s->off_rhs + L_new.back().total <= s->ws_bytes, GNPDE_EWS,
                    "sharded_solver_set_boundary_chunks: chunk %d needs %zu bytes of scratch, the workspace has %zu", c,
                    L_new.back().total, s->ws_bytes - s->off_rhs);
    ptr_new.push_back(push_chunk_ptr[c]);
  }
  ptr_new.push_back(push_chunk_ptr[n_chunks]);
  GNPDE_CHECK_ARG(row == s->n_own, GNPDE_EINVAL, "sharded_solver_set_boundary_chunks: the chunks end at row %d of %d", row, s->n_own);
  if (s->n_send > 0) {
    std::vector<char> seen(static_cast<size_t>(s->n_send), 0);
    for (int w = 0; w < s->n_send; ++w) {
      GNPDE_CHECK_ARG(push_order[w] >= 0 && push_order[w] < s->n_send && !seen[push_order[w]], GNPDE_EINVAL,
                      "sharded_solver_set_boundary_chunks: push_order is not a permutation of the send slots");
      seen[push_order[w]] = 1;
    }
    GNPDE_HIP(hipMemcpy(s->d_order, push_order, static_cast<size_t>(s->n_send) * 4, hipMemcpyHostToDevice));
  }
  s->g_chunk.swap(g_new);
  s->rhs_chunk.swap(rhs_new);
  for (int c = 0; c < n_chunks; ++c) s->rhs_chunk[c].graph = &s->g_chunk[c];
  s->L_chunk.swap(L_new);
  s->push_chunk_ptr.swap(ptr_new);
  return 0;
}

extern "C" int gnpde_sharded_solver_timing(gnpde_sharded_solver_t* s, int64_t* stamps, int32_t capacity_evals, int32_t* n_evals,
                                           int64_t* ticks_per_second) {
  GNPDE_CHECK_ARG(s != nullptr && (stamps != nullptr || capacity_evals == 0) && capacity_evals >= 0, GNPDE_EINVAL,
                  "sharded_solver_timing: bad arguments");
  if (n_evals) *n_evals = s->d_stamps != nullptr && s->exchanges ? s->n_evals : 0;
  if (ticks_per_second) {
    int khz = 0, dev = 0;
    GNPDE_HIP(hipGetDevice(&dev));
    GNPDE_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    *ticks_per_second = static_cast<int64_t>(khz) * 1000;
  }
  if (s->d_stamps == nullptr || !s->exchanges || capacity_evals == 0) return 0;
  const int n = s->n_evals < capacity_evals ? s->n_evals : capacity_evals;
  GNPDE_HIP(hipMemcpy(stamps, s->d_stamps, static_cast<size_t>(n) * 4 * sizeof(int64_t), hipMemcpyDeviceToHost));   // synchronises
  return 0;
}

// ------------------------------------------------------------------------------------------------ the engine, for dopri5.hip (rhs.h)
namespace gnpde {

ShardedShape sharded_shape(gnpde_sharded_solver* s) {
  ShardedShape h{};
  h.n_own = s->n_own; h.n_local = s->n_own + s->n_halo; h.d = s->d; h.ld = s->ld;
  h.world = s->p2p ? s->p2p->world : (s->comm ? s->comm->world : 1);
  h.p2p = s->p2p != nullptr;
  h.n_buffers = s->p2p ? s->p2p->n_buffers : 0;
  h.buffer_bytes = s->p2p ? s->p2p->buffer_bytes : 0;
  h.rhs = &s->rhs_int;
  return h;
}

float* sharded_stage_buffer(gnpde_sharded_solver* s, int b) { return stage_buffer(s, b); }

int sharded_prepare_adaptive(gnpde_sharded_solver* s) {
  GNPDE_CHECK_ARG(s->p2p != nullptr, GNPDE_ESTATE, "adaptive partitioned solve: the P2P transport only");
  GNPDE_CHECK_ARG(s->rhs_chunk.empty(), GNPDE_ESTATE, "adaptive partitioned solve: no chunked boundary pass");
  drop_sharded_graph(s);
  if (s->d_stamps) (void)hipFree(s->d_stamps);    // (per-evaluation stamps are sized by a fixed grid's evaluation count)
  s->d_stamps = nullptr;
  s->n_evals = 0;
  s->eval_cursor = 0;
  return 0;
}

int sharded_enqueue_eval(gnpde_sharded_solver* s, float* u, const gnpde_epilogue_t& e, hipStream_t st) {
  s->eval_cursor = 0;      // (first / last only matter to the chunked boundary pass and the stamps, both off here)
  return enqueue_eval(s, u, e, st);
}

int sharded_enqueue_sum(gnpde_sharded_solver* s, double* value, int n_partials, hipStream_t st) {
  GNPDE_CHECK_ARG(s->p2p != nullptr && value != nullptr && n_partials >= 1, GNPDE_EINVAL, "sharded sum: bad arguments");
  hipLaunchKernelGGL(p2p_sum_kernel<false>, dim3(1), dim3(kBlock), 0, st, value, n_partials, s->p2p->ctl, s->d_peer_ctl, s->p2p->rank,
                     s->p2p->world, s->max_spins);
  GNPDE_LAUNCH_CHECK();
  return 0;
}

int sharded_lost_peer(gnpde_sharded_solver* s, int* lost) {
  uint32_t w = 0;
  if (s->p2p) GNPDE_HIP(hipMemcpy(&w, s->p2p->ctl + kCtlErr, sizeof(w), hipMemcpyDeviceToHost));
  *lost = w != 0u;
  return 0;
}

}  // namespace gnpde

extern "C" int gnpde_sharded_solver_set_spin_limit(gnpde_sharded_solver_t* s, int64_t max_spins) {
  GNPDE_CHECK_ARG(s != nullptr && max_spins > 0, GNPDE_EINVAL, "sharded_solver_set_spin_limit: bad argument");
  drop_sharded_graph(s);
  s->max_spins = max_spins;
  return 0;
}

extern "C" int gnpde_sharded_solver_num_rhs_evals(const gnpde_sharded_solver_t* s) { return s ? s->n_evals : 0; }

extern "C" int gnpde_sharded_solver_destroy(gnpde_sharded_solver_t* s) {
  if (!s) return 0;
  drop_sharded_graph(s);
  if (s->cap_stream) (void)hipStreamDestroy(s->cap_stream);
  if (s->e_pack) (void)hipEventDestroy(s->e_pack);
  if (s->e_recv) (void)hipEventDestroy(s->e_recv);
  if (s->d_stamps) (void)hipFree(s->d_stamps);
  if (s->gen.dev) (void)hipFree(s->gen.dev);
  delete s;
  return 0;
}
