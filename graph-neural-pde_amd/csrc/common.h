// Internal helpers shared by the translation units of libgnpde_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "gnpde.h"

namespace gnpde {

void set_error(const char* fmt, ...);

#define GNPDE_CHECK_ARG(cond, code, ...)            \
  do {                                              \
    if (!(cond)) {                                  \
      ::gnpde::set_error(__VA_ARGS__);              \
      return (code);                                \
    }                                               \
  } while (0)

#define GNPDE_HIP(call)                                                                   \
  do {                                                                                    \
    hipError_t _e = (call);                                                               \
    if (_e != hipSuccess) {                                                               \
      ::gnpde::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, \
                         __LINE__);                                                       \
      return static_cast<int>(_e);                                                        \
    }                                                                                     \
  } while (0)

// launch check that does not synchronise (valid inside stream capture)
#define GNPDE_LAUNCH_CHECK()                                                              \
  do {                                                                                    \
    hipError_t _e = hipGetLastError();                                                    \
    if (_e != hipSuccess) {                                                               \
      ::gnpde::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),       \
                         __FILE__, __LINE__);                                             \
      return static_cast<int>(_e);                                                        \
    }                                                                                     \
  } while (0)

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kBlock = 256;        // 4 wavefronts, one per SIMD
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kXcds = 8;           // MI355X: 8 XCDs, block b is dispatched to XCD b % 8

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Grid rounded to a multiple of the XCD count so that the swizzle below is a bijection.
inline unsigned xcd_grid(long long blocks) {
  if (blocks < 1) blocks = 1;
  return static_cast<unsigned>((blocks + kXcds - 1) / kXcds * kXcds);
}

// Blocks that land on one XCD (b % 8 equal) get a CONTIGUOUS range of work items, so rows that
// are neighbours in the (locality-ordered) graph share that XCD's 4 MiB L2.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned b, unsigned nblocks) {
  const unsigned per = nblocks / kXcds;
  return (b % kXcds) * per + (b / kXcds);
}

// Rows -> XCDs of the aggregation launches (csrc/spmm.hip explains why): the rows [row_begin, row_end) are dealt in blocks of
// 2^shift consecutive rows; eight consecutive blocks form a group and the eight XCDs take the blocks of group j in an order
// rotated by a hash of j:  block(x, j) = 8 j + ((x + hash(j)) mod 8)  -- a bijection for any hash.  shift < 0: contiguous
// eighths.  Host + device, so that the host tests exercise the code the kernels run
// (gnpde_xcd_row_map).
static_assert((kXcds & (kXcds - 1)) == 0, "xcd_row_of rotates inside groups of kXcds blocks");

// length of an XCD's row list (an upper bound with hashed blocks: entries past the last row are invalid)
__host__ __device__ __forceinline__ int xcd_rows_per(int rn, int shift) {
  if (shift < 0) return (rn + kXcds - 1) / kXcds;
  const int nb = (rn + (1 << shift) - 1) >> shift;
  return ((nb + kXcds - 1) / kXcds) << shift;
}

// r-th row of the list of XCD x, or -1
__host__ __device__ __forceinline__ int xcd_row_of(int row_begin, int row_end, int shift, int x, int r) {
  if (shift < 0) {
    const int per = (row_end - row_begin + kXcds - 1) / kXcds;
    if (r >= per) return -1;
    const int row = row_begin + x * per + r;
    return row < row_end ? row : -1;
  }
  const int j = r >> shift, i = r & ((1 << shift) - 1);
  const unsigned rot = (static_cast<unsigned>(j) * 2654435761u) >> 29;                 // top bits of a multiplicative hash: 0..7
  const int b = j * kXcds + static_cast<int>((static_cast<unsigned>(x) + rot) & static_cast<unsigned>(kXcds - 1));
  const long long row = static_cast<long long>(row_begin) + (static_cast<long long>(b) << shift) + i;
  return row < row_end ? static_cast<int>(row) : -1;
}

// kernel-variant knobs for A/B measurements (gnpde_tune); 0 = the default variant
enum {
  GNPDE_TUNE_SPMM_VARIANT = 0,         // aggregation kernel <L,K,U,nontemporal> variant (tools/spmm_ab.py)
  GNPDE_TUNE_FUSED_BLOCKS_PER_CU = 1,  // persistent grid of the one-pass kernel
  GNPDE_TUNE_ONE_PASS = 2,             // 1: GRAND-nl evaluations use the one-pass kernel
  GNPDE_TUNE_FORK = 3,                 // 1: hub-row work on a second stream (fork / join)
  GNPDE_TUNE_ATT_GENERIC_ROWS = 4,     // 1: generic row-attention kernel instead of the scaled-dot specialisation
  GNPDE_TUNE_RK4_CLASSIC = 5,          // 1: torchdiffeq-order rk4 stages (k1..k3 stored) instead of the compact form
  GNPDE_TUNE_ROW_FUSION = 6,           // row attention + aggregation in one kernel (attn_spmm_kernel): 0 = only for graphs whose state fits an XCD's L2
                                       // (launch-bound evaluations), 1 = always (slower at scale, A/B), 2 = never
  GNPDE_TUNE_ONE_PASS_VARIANT = 7,     // register / unroll variants of the one-pass kernel (tools/onepass_ab.py)
  GNPDE_TUNE_LINEAR_STREAMING = 8,     // 1: one-tile-per-wave projection kernel instead of the persistent one
  GNPDE_TUNE_SPMM_PART = 9,            // measurement only: 1 = hub chunks only, 2 = rows only (results are then incomplete)
  GNPDE_TUNE_XCD_ROWS = 10,            // 0: as gnpde_graph_t.xcd_deal says; 1: contiguous eighths for every graph; 2: hashed blocks for every graph
  GNPDE_TUNE_HUB_FOLD = 11,            // 2: phase 1 of the hub-row attention folds the row's chunk partials straight from memory instead of staging them through LDS (default: staged)
  GNPDE_TUNE_ADJOINT_GRAM = 12,        // 1: weight-gradient Gram blocks of the adjoint stage on the VALU (scalar-operand kernel) instead of the matrix cores
  GNPDE_TUNE_SWEEP_UNSWAPPED = 15,     // 1: the reverse sweep over a recorded solve gathers the STATE rows again (round-6 first form) instead of the cotangent rows only (A/B)
  GNPDE_TUNE_KEY_TABLE = 14,           // 1: keep q||k interleaved [n, 2A] where the solver would write two tables (A/B)
  GNPDE_TUNE_LINEAR_DIAG = 13,         // A/B diagnostics of the staged projection kernel (1: no stores, 2: loads alone); never set in production
  GNPDE_TUNE_GMAX_SMALL = 16,          // 1: squareplus on a small grid keeps the slot atomics + memset + fold launch (A/B against the per-wave maxima folded by the second sweep)
  GNPDE_TUNE_ATT_ROWS16 = 17,          // 9: the row softmax of the scaled-dot row kernel (4 heads) keeps a whole wave per row of <= 16 entries (A/B against the quarter-wave packing)
  GNPDE_TUNE_COUNT = 18
};
extern int g_tune[GNPDE_TUNE_COUNT];

// Optional second stream for the hub-row work of a launch sequence.  The long-row passes touch rows
// disjoint from the main kernels', so they run as a parallel branch: fork_begin makes `aux` wait for
// everything queued on `s`, fork_end makes `s` wait for the branch.  Inside stream capture this becomes
// a fork/join in the hipGraph; with aux == nullptr everything stays on one stream.
struct Fork {
  hipStream_t aux = nullptr;
  hipEvent_t e_fork = nullptr;
  hipEvent_t e_join = nullptr;
};

// Both report failures: a dropped record / wait would silently serialise the branch or -- worse -- let it race with
// the main stream inside a captured graph.
inline int fork_begin(const Fork* f, hipStream_t s, hipStream_t* branch) {
  *branch = s;
  if (f == nullptr || f->aux == nullptr) return 0;
  GNPDE_HIP(hipEventRecord(f->e_fork, s));
  GNPDE_HIP(hipStreamWaitEvent(f->aux, f->e_fork, 0));
  *branch = f->aux;
  return 0;
}

inline int fork_end(const Fork* f, hipStream_t s, hipStream_t branch) {
  if (branch == s) return 0;
  GNPDE_HIP(hipEventRecord(f->e_join, branch));
  GNPDE_HIP(hipStreamWaitEvent(s, f->e_join, 0));
  return 0;
}

// internal launchers used by the solver (defined in the kernel translation units)
int launch_spmm_rhs(const gnpde_graph_t* g, const float* w_csr, const float* u, int d, int ld,
                    const gnpde_epilogue_t* epi, float* plain_out, void* ws, size_t ws_bytes,
                    hipStream_t stream, const Fork* fork = nullptr, bool padded_rows = false);

// adjoint stage, row side (spmm.hip): F with its LINCOMB epilogue + r_e = g[row] . u[col] + per-wave dots of g . F and g . x0
int adjoint_rows_dot_slots(const gnpde_graph_t* g, int d);
int launch_adjoint_rows(const gnpde_graph_t* g, const float* w_csr, const float* u, const float* gvec, int d, int ld,
                        const gnpde_epilogue_t* epi, float* r_out, float* dots, void* ws, size_t ws_bytes, hipStream_t stream,
                        bool padded_rows, bool accumulate_r = false, float r_scale = 1.0f, const int* wpos = nullptr);

// head-wise weighted row sums with a lane per entry (backward.hip): the adjoint stage's d q (pos == nullptr) and d k (rows of the
// transposed graph, pos = its positions -> CSR positions of ds); rows without entries are not written
bool head_rowsum_supported(int heads, int dk);
int launch_head_rowsum(const gnpde_graph_t* g, const int* pos, const float* ds, int heads, int dk, const float* feat, int ldf,
                       float scale, float* out, int ldo, float* hub_ws, hipStream_t s);
// scratch (floats) that lets the backward row passes spread the hub rows of `g` over the chip as 512-entry chunks (hub_ws; nullptr: one
// workgroup per hub)
size_t hub_bwd_workspace_floats(const gnpde_graph_t* g, int heads, int att_dim);

// row softmax backward (ds) with the row-side head sum d q formed in the same kernel (backward.hip; heads * d_k <= 32)
bool attention_rows_bwd_dq_supported(int heads, int dk);
int launch_attention_rows_bwd_dq(const gnpde_graph_t* g, const gnpde_attention_t* att, const float* r_csr, const float* scale,
                                 int32_t scale_sigmoid, float* ds_csr, float* dq, int lddq, float* hub_ws, hipStream_t s, const int* rpos = nullptr);

// attention + aggregation of the short rows in one kernel (spmm.hip) and the hub-row weights it needs (attention.hip)
bool attn_spmm_supported(const gnpde_graph_t* g, const gnpde_attention_t& at, int d, int ld, const float* u,
                         const gnpde_epilogue_t& e);
int launch_attn_spmm(const gnpde_graph_t* g, const gnpde_attention_t* at, const float* w_hub_csr, const float* u, int d, int ld,
                     const gnpde_epilogue_t* epi, void* ws, size_t ws_bytes, hipStream_t stream, bool padded_rows);
int launch_hub_attention(const gnpde_graph_t* g, const gnpde_attention_t* at, float* w_mean_csr, void* ws, size_t ws_bytes,
                         hipStream_t stream);

// early_stop.hip
// gate / tag (nullable device ints): *gate == 0 skips the evaluation when the kernels run, *tag replaces `step` (the
// device-controlled dopri5 decides both per trial step)
int enqueue_early_stop_eval(const gnpde_decoder_t& dec, const float* y, int ld, int n, int step, int* state, int* trace,
                            int trace_capacity, hipStream_t st, const int* gate = nullptr, const int* tag = nullptr);
int check_decoder(const gnpde_decoder_t* dec, int d_state);

// Stage algebra and error norm of the adaptive solver with the step size optionally read from DEVICE memory when the kernel runs
// (misc.hip): *scale multiplies every coefficient (NULL = the plain C entry points).  partial_blocks != NULL: only the block
// partial sums are formed (workspace[0 .. *partial_blocks)), the caller folds them.
int launch_lincomb(const float* base, const float* const* v, const float* coef, int32_t n_v, int64_t n, float* out,
                   hipStream_t s, const float* scale);
int launch_rk_error_ratio(const float* y0, const float* y1, const float* const* k, const float* coef, int32_t n_k, float atol,
                          float rtol, int64_t n, int32_t d, int32_t ld, float* ratio, float* workspace, hipStream_t s,
                          const float* scale, int* partial_blocks);
int launch_dopri5_interp(const float* y0, const float* y1, const float* const* k, const float* mid_coef, float h, float x,
                         int64_t n, int32_t d, int32_t ld, float* out, hipStream_t stream);

}  // namespace gnpde
