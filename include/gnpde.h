/*
 * gnpde.h -- C ABI of libgnpde_hip.so: the MI355X (gfx950) implementation of the GRAND / BLEND
 * ODE right-hand side f(t,x) = alpha * (A(x) x - x) + beta * x0 and of the fixed-step solver loop
 * around it.
 *
 * The reference (twitter-research/graph-neural-pde) has no FFI: its boundary for this path is the
 * Python protocol ODEFunc.forward(t, x) / ODEblock.forward(x) (src/base_classes.py:32-95).  The
 * entry points below are what a binding for that path binds; each cites the reference code it
 * replaces (paths relative to the reference root).  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - every pointer marked "device" is HBM memory owned by the caller; nothing here allocates
 *     device memory, synchronises the device or touches the default stream (safe inside hipGraph
 *     capture), except the gnpde_solver_* objects which own their hipGraphExec;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - every function returns 0 on success, a hipError_t (> 0) for runtime failures, or a negative
 *     GNPDE_E* code for bad arguments; gnpde_last_error() returns a thread-local message;
 *   - all floating point is IEEE fp32 (no fast-math), all device indices int32;
 *   - learnable scalars (alpha_train, beta_train, ...) are read from DEVICE pointers so that no
 *     host synchronisation is needed between optimiser steps and solves.
 */
#ifndef GNPDE_H
#define GNPDE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNPDE_ABI_VERSION 7   /* 7: gnpde_dopri5_create_sharded (device controller over the row partition), gnpde_dopri5_set_pair, gnpde_sharded_solver_set_general, gnpde_graph_build_device;  6: gnpde_adjoint_set_tape takes csr_from_t, gnpde_adjoint_tape_swapped, gnpde_linear_split;  5: gnpde_solver_set_tape / gnpde_adjoint_set_tape (recorded fixed-grid solve);  4: gnpde_dopri5_set_tape / _tape_backward, gnpde_adjoint_adaptive_*, GNPDE_METHOD_MIDPOINT;  2: gnpde_graph_t.xcd_deal appended, gnpde_xcd_row_map; 3: gnpde_attention_t.graph_t / t_from_csr appended,
                                 gnpde_adjoint_*, gnpde_stream_read; gnpde_graph_t.n_bin_le64 and gnpde_attention_t.n_key_rows in what was
                                 padding (struct sizes unchanged) */

#define GNPDE_EINVAL   (-1)  /* bad argument                                  */
#define GNPDE_ESHAPE   (-2)  /* shape not supported by any kernel variant     */
#define GNPDE_EWS      (-3)  /* workspace too small                           */
#define GNPDE_ESTATE   (-4)  /* object used in the wrong state                */

/* Rows with more than GNPDE_LONG_ROW non-zeros are split into chunks of that many edges so that a
 * hub node (ogbn-arxiv: degree 13k) is spread over many wavefronts. */
#define GNPDE_LONG_ROW 512

int         gnpde_abi_version(void);
const char* gnpde_last_error(void);
/* Select a kernel variant for A/B measurements (key 0: aggregation kernel variant); 0 = default. */
int         gnpde_tune(int32_t key, int32_t value);

/* ------------------------------------------------------------------------------------------------
 * Graph preparation (host, C++).  Replaces the implicit COO handling of torch_sparse.spmm /
 * torch_scatter (reference call sites: src/function_transformer_attention.py:35,190-191,213;
 * src/function_laplacian_diffusion.py:31-35): COO int64 [2,E] in ANY order, duplicates allowed
 * -> CSR with a stable permutation, a CSC view, and the long-row chunk list.
 * ---------------------------------------------------------------------------------------------- */

/* Number of long rows / long-row chunks / long columns the edge list produces (sizes of the arrays
 * below).  `col` may be NULL (then *n_long_cols = 0). */
int gnpde_graph_count_long(const int64_t* row, const int64_t* col, int64_t n_edges, int32_t n_nodes,
                           int32_t* n_long_rows, int32_t* n_long_chunks, int32_t* n_long_cols);

/* All arrays are HOST memory, caller allocated:
 *   rowptr[n+1], colidx[E], perm[E] (CSR position -> index into the caller's edge list; stable, so
 *   duplicates keep their relative order), rowidx[E] (row of each CSR position),
 *   cscptr[n+1], cscpos[E] (for every column, the CSR positions of its entries, ascending)  -- may
 *   both be NULL;  long_rows[n_long_rows], long_chunk_ptr[n_long_rows+1],
 *   long_chunk_row/begin/end/first[n_long_chunks] (first = index of the first chunk of the same row),
 *   long_cols[n_long_cols] (NULL without the CSC view);
 *   bin_rows[4n]: one int32x4 record {row, first CSR position, length, 0} per listed row, grouped by
 *   degree class -- first the rows with 1..16 entries, then those with 17..GNPDE_LONG_ROW (empty and
 *   long rows are not listed); bin_counts[2] = sizes of the two classes.  The first class is in ascending row
 *   order, the second LONGEST FIRST (ties in row order): a wavefront takes one such row, and the long ones should start the
 *   row-attention launch, not end it.  (One 16-byte load replaces the rowptr indirection in the row-attention kernels.)
 * Returns GNPDE_EINVAL if an index is outside [0, n_nodes). */
int gnpde_graph_build(const int64_t* row, const int64_t* col, int64_t n_edges, int32_t n_nodes,
                      int32_t* rowptr, int32_t* colidx, int32_t* perm, int32_t* rowidx,
                      int32_t* cscptr, int32_t* cscpos,
                      int32_t* long_rows, int32_t* long_chunk_ptr, int32_t* long_chunk_row,
                      int32_t* long_chunk_begin, int32_t* long_chunk_end, int32_t* long_cols,
                      int32_t* bin_rows, int32_t* bin_counts, int32_t* long_chunk_first);

/* The same arrays built ON THE DEVICE from an edge list that already lives there (blocks that hand over a new edge set every training
 * forward: reference src/block_transformer_hard_attention.py:55-61, src/block_transformer_rewiring.py), element for element what
 * gnpde_graph_build gives.  Two calls around ONE host read:
 *   gnpde_graph_build_device       row / col: device int64 [E]; fills rowptr[n+1], colidx / perm / rowidx / cscpos [E], cscptr[n+1],
 *                                  bin_rows[4n] and counts[16] (device): [0] != 0: an index outside [0, n); [1] rows with 1..16 entries,
 *                                  [2] with 17..GNPDE_LONG_ROW, [3] of those with <= 64, [4] long rows, [5] long columns, [6] 512-entry
 *                                  chunks of the long rows, [7] / [8] longest row / column.
 *   gnpde_graph_build_device_long  (after the caller read counts and allocated the lists) long_rows[n_long_rows],
 *                                  long_chunk_ptr[n_long_rows+1], long_chunk_row / begin / end / first[n_long_chunks],
 *                                  long_cols[n_long_cols]; scratch_count: one device word.
 * Workspace: gnpde_graph_build_device_workspace_bytes(E, n) device bytes, 256-byte aligned, the same block for both calls. */
size_t gnpde_graph_build_device_workspace_bytes(int64_t n_edges, int32_t n_nodes);
int gnpde_graph_build_device(const int64_t* row, const int64_t* col, int64_t n_edges, int32_t n_nodes, int32_t* rowptr, int32_t* colidx,
                             int32_t* perm, int32_t* rowidx, int32_t* cscptr, int32_t* cscpos, int32_t* bin_rows, int32_t* counts,
                             void* workspace, size_t workspace_bytes, void* stream);
int gnpde_graph_build_device_long(const int32_t* rowptr, const int32_t* cscptr, int32_t n_nodes, int32_t n_long_rows, int32_t n_long_chunks,
                                  int32_t n_long_cols, int32_t* long_rows, int32_t* long_chunk_ptr, int32_t* long_chunk_row,
                                  int32_t* long_chunk_begin, int32_t* long_chunk_end, int32_t* long_cols, int32_t* long_chunk_first,
                                  int32_t* scratch_count, void* workspace, size_t workspace_bytes, void* stream);

/* Balanced k-way row partition for the multi-GPU path (no METIS offline; no reference equivalent): size-capped
 * label-propagation clusters packed into parts, then node-level refinement under the balance constraint.  The parts are
 * balanced (3 %) on  entries + row_weight  per row -- 1 follows the aggregation time, larger values even out the node counts --
 * and the clusters are capped at a part's work / cluster_div.  part[n] receives values in [0, n_parts).  Host only,
 * deterministic for given arguments.  gnpde_partition_rows = row_weight 1, cluster_div 4.  The outcome of the heuristic moves
 * by tens of per cent with these two; the Python layer scores a few combinations by the busiest xGMI link and the busiest
 * rank they produce and keeps the cheapest (distributed.PartitionPlan). */
int gnpde_partition_rows(const int32_t* rowptr, const int32_t* colidx, int32_t n_nodes,
                         int32_t n_parts, int32_t refine_iters, uint64_t seed, int32_t* part);
int gnpde_partition_rows_ex(const int32_t* rowptr, const int32_t* colidx, int32_t n_nodes, int32_t n_parts,
                            int32_t refine_iters, uint64_t seed, int32_t row_weight, int32_t cluster_div, int32_t* part);

/* Communication refinement of a row partition (no reference equivalent).  A partitioned evaluation waits for the rows each rank
 * RECEIVES, not for cut edges: M[r][q] = distinct nodes of part q referenced by rows of part r = rows on the link q -> r per
 * evaluation.  Hill climbing on single-node moves with exact bookkeeping: phase 1 lowers the total of M without raising the
 * busiest link, phase 2 lowers the excess of the links above 90 % of the busiest; the parts stay inside the balance window of
 * gnpde_partition_rows_ex (entries + row_weight per row).  part[n] in / out.  stats (nullable, int64[5]): busiest link before /
 * after, total received rows before / after, moves.  Host only, deterministic. */
int gnpde_partition_refine_links(const int32_t* rowptr, const int32_t* colidx, int32_t n_nodes, int32_t n_parts,
                                 int32_t row_weight, int32_t max_passes, int32_t* part, int64_t* stats);

/* Device view of a prepared graph (all pointers device memory). */
typedef struct gnpde_graph {
  int32_t n;                         /* nodes (square operator)                              */
  int32_t e;                         /* stored entries                                       */
  const int32_t* rowptr;             /* [n+1]                                                */
  const int32_t* colidx;             /* [e]  CSR order                                       */
  const int32_t* rowidx;             /* [e]  row of each CSR position                        */
  const int32_t* perm;               /* [e]  CSR position -> caller edge id                  */
  const int32_t* cscptr;             /* [n+1] or NULL                                        */
  const int32_t* cscpos;             /* [e]   or NULL                                        */
  int32_t n_long_rows;
  int32_t n_long_chunks;
  const int32_t* long_rows;          /* [n_long_rows]                                        */
  const int32_t* long_chunk_ptr;     /* [n_long_rows+1]                                      */
  const int32_t* long_chunk_row;     /* [n_long_chunks]                                      */
  const int32_t* long_chunk_begin;   /* [n_long_chunks]                                      */
  const int32_t* long_chunk_end;     /* [n_long_chunks]                                      */
  int32_t n_long_cols;               /* columns with more than GNPDE_LONG_ROW entries        */
  int32_t n_bin16;                   /* rows with 1..16 entries                              */
  int32_t n_bin64;                   /* rows with 17..GNPDE_LONG_ROW entries                 */
  int32_t max_row_len;               /* longest row                                          */
  int32_t max_col_len;               /* longest column (0 without the CSC view)              */
  int32_t row_begin;                 /* aggregation covers rows [row_begin, n) (0 normally; the boundary
                                        pass of a row-partitioned graph starts after the interior rows) */
  const int32_t* long_cols;          /* [n_long_cols] or NULL                                */
  const int32_t* bin_rows;           /* [(n_bin16 + n_bin64) * 4] records {row, begin, len, 0} */
  const int32_t* long_chunk_first;   /* [n_long_chunks] index of the first chunk of the same row */
  int32_t xcd_deal;                  /* how the aggregation launches deal the rows to the 8 XCDs (gnpde_xcd_row_map):
                                        GNPDE_XCD_CONTIGUOUS or GNPDE_XCD_HASHED; chosen per graph by whoever builds it */
  int32_t n_bin_le64;                /* (ABI 3) how many records of the second degree class (17..GNPDE_LONG_ROW entries, longest
                                        first) have at most 64 entries: they TRAIL that class, and the backward kernels give them a
                                        one-pass launch of their own (one entry per lane); 0: the whole class in one launch */
} gnpde_graph_t;

/* gnpde_graph_t.xcd_deal.  CONTIGUOUS: XCD x takes the x-th eighth of the rows -- neighbouring rows share an XCD's L2, and the
 * deal is balanced as long as the row length does not depend on the row id.  HASHED: blocks of 16 .. 128 consecutive rows, the
 * eight blocks of a group taken by the eight XCDs in an order rotated by a hash of the group -- balanced whatever the ids mean
 * (R-MAT: the expected degree falls 3x with every set id bit, contiguous eighths leave the launch 1.46x out of balance; ids in
 * order of time, crawl or degree do the same to real graphs).  The Python layer measures the contiguous deal's imbalance when it
 * builds a graph (work model: entries + 3 per row) and switches to HASHED above 3 %. */
#define GNPDE_XCD_CONTIGUOUS 0
#define GNPDE_XCD_HASHED     1

/* ------------------------------------------------------------------------------------------------
 * Epilogue of one right-hand-side evaluation, optionally fused with the solver's stage algebra.
 *   k      = alpha' * (A u - u_i) [+ beta * x0_i]      alpha' = sigmoid(*alpha) or *alpha
 *            (src/function_laplacian_diffusion.py:43-51 == function_transformer_attention.py:46-53
 *             == function_GAT_attention.py:56-64)
 * and then, per `stage` (torchdiffeq 0.2.1 fixed-grid solvers, called from
 * src/block_constant.py:57-62; rk4 == 3/8-rule rk4_alt_step_func, cf. src/early_stop_solver.py:150-155):
 *   GNPDE_STAGE_RHS    out_k = k
 *   GNPDE_STAGE_EULER  out_y = y + dt * k                       (u == y here, so out_y must differ)
 *   GNPDE_STAGE_RK1    out_k = k1 ; out_y = y + dt*k1*(1/3)
 *   GNPDE_STAGE_RK2    out_k = k2 ; out_y = y + dt*(k2 - k1*(1/3))
 *   GNPDE_STAGE_RK3    out_k = k3 ; out_y = y + dt*(k1 - k2 + k3)
 *   GNPDE_STAGE_RK4               ; out_y = y + (k1 + 3*(k2+k3) + k4)*dt*0.125   (in place on y)
 *   Compact rk4 (same step, stage states expressed through the previous stage INPUTS so that k1..k3 are
 *   never stored or re-read: 16 instead of 24 state-sized streams per step; u2 = y + dt k1/3 etc.):
 *   GNPDE_STAGE_RK1C   out_y = u + dt*k1*(1/3)                          (u == y)
 *   GNPDE_STAGE_RK2C   out_y = (2*y - u) + dt*k2                        (u == u2;  == y + dt*(k2 - k1/3))
 *   GNPDE_STAGE_RK3C   out_y = (2*a - u) + dt*k3      a = u2  (field k1)  (u == u3;  == y + dt*(k1 - k2 + k3))
 *   GNPDE_STAGE_RK4C   out_y = ((6*a + 3*u - y) + dt*k4)*0.125   a = u3  (u == u4; in place on y)
 *   General explicit Runge-Kutta stage (adaptive solvers: dopri5 of torchdiffeq, reference default method):
 *   GNPDE_STAGE_LINCOMB  out_k = k (if non-NULL) ;
 *                        out_y = y + sum_{j < n_prev} coef[j] * prev[j] + coef[n_prev] * k   (if non-NULL)
 *                        with coef[j] = fl32(beta_j) * fl32(dt) prepared by the caller (torchdiffeq's
 *                        k.matmul(beta * dt) rounding), or coef[j] = fl32(beta_j) and coef_scale -> fl32(dt) on the device.
 * `u` is the stage input the operator is applied to, `y` the state at the start of the step.
 * ---------------------------------------------------------------------------------------------- */
enum {
  GNPDE_STAGE_RHS = 0, GNPDE_STAGE_EULER = 1,
  GNPDE_STAGE_RK1 = 2, GNPDE_STAGE_RK2 = 3, GNPDE_STAGE_RK3 = 4, GNPDE_STAGE_RK4 = 5,
  GNPDE_STAGE_RK1C = 6, GNPDE_STAGE_RK2C = 7, GNPDE_STAGE_RK3C = 8, GNPDE_STAGE_RK4C = 9,
  GNPDE_STAGE_LINCOMB = 10
};
#define GNPDE_MAX_PREV 7

typedef struct gnpde_epilogue {
  const float* alpha;      /* device scalar (alpha_train)                                   */
  const float* beta;       /* device scalar (beta_train); used iff x0 != NULL               */
  const float* x0;         /* [n, ld] device or NULL (opt['add_source'] false)              */
  int32_t alpha_sigmoid;   /* 1: alpha' = sigmoid(alpha)  (opt['no_alpha_sigmoid'] false)   */
  int32_t stage;           /* GNPDE_STAGE_*                                                 */
  float   dt;              /* step size for the fused stage algebra                         */
  const float* y;          /* [n, ld] state at step start (stages != RHS)                   */
  const float* k1;         /* [n, ld] (RK2..RK4)                                            */
  const float* k2;         /* [n, ld] (RK3, RK4)                                            */
  const float* k3;         /* [n, ld] (RK4)                                                 */
  float* out_k;            /* [n, ld] (RHS, RK1..RK3)                                       */
  float* out_y;            /* [n, ld] (all but RHS)                                         */
  int32_t n_prev;          /* LINCOMB: number of earlier stage derivatives                  */
  int32_t pad_;
  const float* prev[GNPDE_MAX_PREV];   /* LINCOMB: [n, ld] each                             */
  float coef[GNPDE_MAX_PREV + 1];      /* LINCOMB: weights of prev[0..n_prev-1], then of k  */
  const float* coef_scale; /* LINCOMB: NULL, or a DEVICE scalar every coef[] is multiplied by when the kernel runs
                              (fl32 product): coef[] = fl32(beta_j), *coef_scale = fl32(dt) kept on the device by the
                              adaptive controller (gnpde_dopri5_*), so that a captured step serves every step size */
} gnpde_epilogue_t;

/* Host-side query of the work distribution of the aggregation launches (no reference equivalent; used by the host tests and
 * by tools/xcd_balance.py).  Workgroup b of a launch runs on XCD b % 8; every XCD walks its own list of rows.  deal =
 * GNPDE_XCD_HASHED: the rows [row_begin, row_end) are dealt in blocks of 2^*row_shift consecutive rows, eight consecutive blocks
 * form a group and XCD x takes block 8 j + ((x + hash(j)) mod 8) of group j; GNPDE_XCD_CONTIGUOUS: *row_shift = -1, XCD x takes
 * the x-th eighth (gnpde_tune(10, 1 / 2) overrides the argument with CONTIGUOUS / HASHED, as it overrides gnpde_graph_t.xcd_deal
 * in the launches).  *rows_per_xcd: length of every XCD's list; map (NULL, or [8 * *rows_per_xcd]): map[x * *rows_per_xcd + r]
 * = the r-th row of XCD x, or -1 past the end.  Every row appears exactly once; the two rows of an even-aligned pair (r, r + 1)
 * are consecutive. */
int gnpde_xcd_row_map(int32_t row_begin, int32_t row_end, int32_t deal, int32_t* row_shift, int32_t* rows_per_xcd, int32_t* map);

/* Bytes of scratch gnpde_spmm_rhs needs for the long-row partial sums (0 if no long rows). */
size_t gnpde_spmm_workspace_bytes(const gnpde_graph_t* g, int32_t d);

/* K7+K8 of SURVEY.md: ax[i] = sum_{e in row i} w[e] * u[col_e]  (torch_sparse.spmm, src/
 * function_laplacian_diffusion.py:31-35, function_transformer_attention.py:35) followed by the
 * epilogue above.  w is in CSR order.  u, x0, y, k*, out_* are [n, ld] row-major with ld >= d.
 * Deterministic (no atomics). */
int gnpde_spmm_rhs(const gnpde_graph_t* g, const float* w_csr, const float* u, int32_t d, int32_t ld,
                   const gnpde_epilogue_t* epi, void* workspace, size_t workspace_bytes, void* stream);

/* Plain aggregation out[i] = sum_e w[e] * u[col_e] without the diffusion epilogue (GAT
 * mix_features path, src/function_GAT_attention.py:33-36). */
int gnpde_spmm(const gnpde_graph_t* g, const float* w_csr, const float* u, int32_t d, int32_t ld,
               float* out, void* workspace, size_t workspace_bytes, void* stream);

/* Sampled dense-dense product on the graph pattern: out_csr[p] = s * a[row_p] . b[col_p], with
 * s = 1 (scale NULL), *scale, or sigmoid(*scale).  This is the gradient of gnpde_spmm_rhs w.r.t. the
 * per-edge weights (d w_e = alpha' g_row . u_col) that autograd needs when the reference's blocks hand
 * attention_weights with gradients (training, src/block_transformer_attention.py:38-39). */
int gnpde_sddmm(const gnpde_graph_t* g, const float* a, int32_t lda, const float* b, int32_t ldb, int32_t d,
                const float* scale, int32_t scale_sigmoid, float* out_csr, void* stream);

/* Backward of the row softmax + head mean of gnpde_edge_attention (scaled-dot, attention_norm_idx 0):
 *   ds[p,h] = s (a[p,h] / H) (dw[p] - sum_{p' in row} a[p',h] dw[p'])   [* edge_w[p] if given]
 * with s = 1 (scale NULL), *scale, or sigmoid(*scale) (device scalar: the alpha of the epilogue, so that dw can be
 * the UNSCALED g_row . x_col, which also yields d alpha).  att_edge is the [E,h] attention in the caller's edge
 * order, dw_csr / ds_csr are in CSR order.  Rows longer than GNPDE_LONG_ROW are taken by whole blocks. */
int gnpde_softmax_rows_bwd(const gnpde_graph_t* g, const float* att_edge, int32_t heads, const float* dw_csr,
                           const float* edge_w_csr, const float* scale, int32_t scale_sigmoid, float* ds_csr,
                           void* stream);

/* Head-wise weighted segment sum over the graph pattern (d q / d k of the attention scores):
 *   out[i, c] = scale * sum_{p in row i} ds[p, head(c)] * feat[col_p, c]           (by_column = 0)
 *   out[j, c] = scale * sum_{p in column j} ds[p, head(c)] * feat[row_p, c]        (by_column = 1, CSC view)
 * c < heads*dk, head(c) = c / dk; dk % 4 == 0 and heads*dk/4 a power of two <= 64. */
int gnpde_head_spmm(const gnpde_graph_t* g, int32_t by_column, const float* ds_csr, int32_t heads, int32_t dk,
                    const float* feat, int32_t ldf, float scale, float* out, int32_t ldo, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dense feature mixing on the fp32 matrix cores (v_mfma_f32_16x16x4_f32):
 *   out[n, m] = x[n, d] * W[m, d]^T + b[m]          (b may be NULL)
 * Replaces nn.Linear Q and K of SpGraphTransAttentionLayer (src/function_transformer_attention.py:
 * 174-175; pass W = [Q.weight; K.weight] to get q||k in one pass, V is skipped because its result
 * is unused when mix_features is false, :34-35) and torch.mm(x, W) of the GAT layer
 * (src/function_GAT_attention.py:106; pass W^T).
 * ---------------------------------------------------------------------------------------------- */
int gnpde_linear(const float* x, int32_t n, int32_t d, int32_t ldx, const float* W, int32_t m,
                 int32_t ldw, const float* b, float* out, int32_t ldo, void* stream);

/* out = relu(x) W^T + b: the decoder of the reference's GNN.forward (F.relu -> [dropout, identity at test time] -> m2,
 * src/GNN.py:61-71) in one pass; same kernels as gnpde_linear with the activation applied to the A operand. */
int gnpde_relu_linear(const float* x, int32_t n, int32_t d, int32_t ldx, const float* W, int32_t m, int32_t ldw,
                      const float* b, float* out, int32_t ldo, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Edge attention: per-edge scores -> per-node normalisation -> head-mean weights.
 * Replaces SpGraphTransAttentionLayer.forward (src/function_transformer_attention.py:190-213),
 * SpGraphAttentionLayer.forward (src/function_GAT_attention.py:111-114),
 * torch_geometric.utils.softmax and utils.squareplus (src/utils.py:179-208).
 * ---------------------------------------------------------------------------------------------- */
enum {
  GNPDE_ATT_SCALED_DOT = 0,  /* sum_c q k / sqrt(d_k)                     (:196)               */
  GNPDE_ATT_COSINE     = 1,  /* cosine similarity, eps 1e-5               (:198-199)           */
  GNPDE_ATT_PEARSON    = 2,  /* mean-centred cosine                       (:201-206)           */
  GNPDE_ATT_EXP_KERNEL = 3,  /* var^2 exp(-|q-k|^2 / (2 l^2))             (:194)               */
  GNPDE_ATT_GAT        = 4   /* LeakyReLU(a_src.h_i + a_dst.h_j)          (GAT :111-113)       */
};

typedef struct gnpde_attention {
  int32_t type;              /* GNPDE_ATT_*                                                    */
  int32_t heads;             /* h                                                              */
  int32_t att_dim;           /* A = h * d_k                                                    */
  int32_t norm_idx;          /* opt['attention_norm_idx']: 0 normalise over row, 1 over column */
  int32_t square_plus;       /* opt['square_plus']                                             */
  float   leaky_slope;       /* GAT only                                                       */
  const float* q;            /* [n, ldqk] device: row-side projection (GAT: h = x W)           */
  const float* k;            /* [n, ldqk] device: column-side projection (GAT: same as q)      */
  int32_t ldqk;
  int32_t n_key_rows;        /* rows of q / k the COLUMNS may address when that is more than the graph's n (a row-partitioned
                              * graph's halo rows lie behind its row range); 0: n.  ABI 3, in what was padding.  The GAT node
                              * terms (:111-114) are formed for all of them                    */
  const float* gat_a;        /* [2*d_k] device (GAT only)                                      */
  const float* output_var;   /* device scalar (exp_kernel)                                     */
  const float* lengthscale;  /* device scalar (exp_kernel)                                     */
  const float* edge_w_csr;   /* [e] CSR order or NULL: opt['reweight_attention'] (:208-209)    */
  /* optional (norm_idx == 1, scaled-dot scores): the transposed graph over the SAME edge list and, for every position of it, the
   * CSR position of that entry in the graph the call is made on.  With them the normalisation over the COLUMNS runs as the fused
   * row kernel over the rows of graph_t (scores, statistics and weights in one pass, weights scattered to their CSR positions)
   * instead of three passes with an [E,h] score round trip; NULL: the three passes. */
  const struct gnpde_graph* graph_t;
  const int32_t* t_from_csr;
} gnpde_attention_t;

/* Scratch: scores [e,h] + segment statistics [n,2h] + global max + GAT node terms [n,2h]. */
size_t gnpde_attention_workspace_bytes(const gnpde_graph_t* g, const gnpde_attention_t* a);

/* Outputs (all optional except at least one):
 *   w_mean_csr [e]    mean over heads of the normalised attention, CSR order (feeds gnpde_spmm_rhs)
 *   att_edge   [E,h]  attention in the CALLER's edge order (what the reference returns, :214)
 *   prods_edge [E,h]  un-normalised scores in the caller's edge order (:214)                  */
int gnpde_edge_attention(const gnpde_graph_t* g, const gnpde_attention_t* a,
                         float* w_mean_csr, float* att_edge, float* prods_edge,
                         void* workspace, size_t workspace_bytes, void* stream);

/* The general attention path one pass at a time, for callers that must exchange between the passes (the row-partitioned solver,
 * SURVEY 8e):  pass 1 scores [e,h] into the workspace (+ the global maximum for squareplus, an order-preserving uint32 word),
 * pass 2 segment statistics m[n,h], den[n,h] (per row for attention_norm_idx 0, per column for 1; den includes the + 1e-16),
 * pass 3 normalise with the statistics / maximum CURRENTLY in the workspace and write w_mean_csr.  Same kernels and arithmetic
 * as gnpde_edge_attention's general path (reference src/function_transformer_attention.py:190-213, src/utils.py:179-208).
 * gnpde_attention_workspace_regions: byte offsets {scores, seg_m, seg_den, gmax} inside the workspace.
 * gnpde_segment_stats_merge: (m, den)[rows[i]] <- merge with (m_in, den_in)[i], i < n_rows (rows NULL: i itself) -- softmax:
 * M = max(m, m_in), den = den e^(m-M) + den_in e^(m_in-M); squareplus: den += den_in.  Partial statistics of a column whose
 * entries live on several ranks are combined with it, peer by peer in a fixed order. */
int gnpde_edge_attention_pass(const gnpde_graph_t* g, const gnpde_attention_t* a, int32_t pass, float* w_mean_csr,
                              void* workspace, size_t workspace_bytes, void* stream);
int gnpde_attention_workspace_regions(const gnpde_graph_t* g, const gnpde_attention_t* a, size_t* offsets);
int gnpde_segment_stats_merge(float* seg_m, float* seg_den, const int32_t* rows, int32_t n_rows, int32_t heads,
                              const float* m_in, const float* den_in, int32_t square_plus, void* stream);

/* Backward of the normalisation + head mean of gnpde_edge_attention for EVERY normaliser the reference has (softmax or
 * squareplus, opt['attention_norm_idx'] 0 or 1; reference src/function_transformer_attention.py:210-213, src/utils.py:179-208,
 * torch_geometric.utils.softmax):  dw_csr[p] = g_row . x_col (gnpde_sddmm without scale, CSR order);
 *   ds_csr[p,h] = dL / d prods[p,h]   (the raw scores, before att->edge_w_csr is applied), scaled by s / H with
 * s = 1 (scale NULL), *scale, or sigmoid(*scale).  squareplus: the global maximum's gradient goes evenly to the entries that
 * attain it, as autograd's `src.max()` does.  The scores and segment statistics are recomputed from att->q / att->k
 * (any att->type).  Feed ds to gnpde_head_spmm for the scaled-dot score's d q / d k. */
size_t gnpde_attention_bwd_workspace_bytes(const gnpde_graph_t* g, const gnpde_attention_t* a);
int gnpde_edge_attention_bwd(const gnpde_graph_t* g, const gnpde_attention_t* a, const float* dw_csr, const float* scale,
                             int32_t scale_sigmoid, float* ds_csr, void* workspace, size_t workspace_bytes, void* stream);

/* The same backward for a gradient that arrives PER HEAD in the caller's edge order (datt_edge [E,h]: what autograd hands back
 * for the [E,h] attention a block computes once per forward, reference src/block_transformer_attention.py:38-39), with the
 * factor that turns d L / d score into what the score function's own derivative needs:
 *   post 0: ds = d L / d raw score (times att->edge_w_csr)                      -- scaled dot, cosine, pearson
 *   post 1: ds = (d L / d score) * score  -- the common factor of every derivative of var^2 exp(-|q-k|^2 / 2 l^2)
 *   post 2: ds = (d L / d score) * LeakyReLU'(pre-activation)                   -- GAT */
int gnpde_edge_attention_bwd_heads(const gnpde_graph_t* g, const gnpde_attention_t* a, const float* datt_edge, int32_t post,
                                   float* ds_csr, void* workspace, size_t workspace_bytes, void* stream);

/* The same ds as gnpde_softmax_rows_bwd, but computed from q and k in ONE pass (scores, row softmax and its
 * backward; no [E,h] attention array): att->type must be GNPDE_ATT_SCALED_DOT with norm_idx 0 and no squareplus,
 * heads in {1,2,4,8}, d_k in {4,8,16} (GNPDE_ESHAPE otherwise: use gnpde_edge_attention + gnpde_softmax_rows_bwd).
 * r_csr[p] = g_row . x_col (gnpde_sddmm without scale); att->edge_w_csr is honoured. */
int gnpde_attention_rows_bwd(const gnpde_graph_t* g, const gnpde_attention_t* att, const float* r_csr, const float* scale,
                             int32_t scale_sigmoid, float* ds_csr, void* stream);


/* ------------------------------------------------------------------------------------------------
 * GRAND-nl evaluation in ONE pass (the solver's hot case): scaled-dot attention, softmax over the row
 * (attention_norm_idx 0, no squareplus), head-mean aggregation, epilogue and solver stage.
 * Replaces the whole of ODEFuncTransformerAtt.forward (src/function_transformer_attention.py:38-53,
 * incl. the Q/K Linear layers :174-175) with a single gather of every neighbour row: the score is
 * evaluated as ((W_k,h^T q_i,h) . x_j + q_i,h . b_k,h) / sqrt(d_k), so neither the [N,2A] projection nor
 * [E,h] scores nor [E] weights are materialised.  proj_w = [Q.weight; K.weight] ([2A, d], 16-byte
 * aligned), proj_b = [Q.bias; K.bias] or NULL.  att->q / att->k are ignored; att->edge_w_csr is honoured.
 * gnpde_attn_rhs_fused_supported tells whether the configuration is covered (else use gnpde_linear +
 * gnpde_edge_attention + gnpde_spmm_rhs, which compute the same function).
 * ---------------------------------------------------------------------------------------------- */
size_t gnpde_attn_rhs_fused_workspace_bytes(const gnpde_graph_t* g, int32_t d, int32_t heads);
int gnpde_attn_rhs_fused_supported(const gnpde_attention_t* att, int32_t d, int32_t ld);
int gnpde_attn_rhs_fused(const gnpde_graph_t* g, const gnpde_attention_t* att, const float* proj_w,
                         const float* proj_b, const float* u, int32_t d, int32_t ld,
                         const gnpde_epilogue_t* epi, void* workspace, size_t workspace_bytes, void* stream);

/* w_csr[p] = mean_h src[perm[p], :]   (src [E,h] in the caller's edge order; h = 1: plain gather).
 * Used when a block hands attention_weights / edge_weight in edge order
 * (src/function_laplacian_diffusion.py:29-35). */
int gnpde_edge_to_csr_mean(const gnpde_graph_t* g, const float* src_edge, int32_t h, float* w_csr,
                           void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fixed-step solver (torchdiffeq `euler` / `rk4` fixed grid, src/block_constant.py:57-62,
 * src/block_transformer_attention.py:58-63) with the whole time loop captured in one hipGraph.
 * ---------------------------------------------------------------------------------------------- */
enum { GNPDE_RHS_LAPLACIAN = 0, GNPDE_RHS_TRANSFORMER = 1, GNPDE_RHS_GAT = 2 };
/* flags of gnpde_rhs_t.  PADDED_ROWS: ld is a multiple of 4 and the columns [d, ld) of EVERY state-shaped operand (stage
 * input, x0, y, k*, outputs) are padding that may be read and overwritten -- lets a state width that is not a multiple of
 * 4 (BLEND ogbn-arxiv: d = 162, reference best_params feat_hidden_dim 64 + pos_enc_hidden_dim 98) use 16-byte lanes. */
#define GNPDE_RHS_PADDED_ROWS 1
/* MIDPOINT (torchdiffeq fixed_grid.py Midpoint, advertised by the reference at src/run_GNN.py:330): y + dt f(y + (dt / 2) f(y)),
 * two evaluations per step, both as GNPDE_STAGE_LINCOMB stages; gnpde_solver_* only. */
enum { GNPDE_METHOD_EULER = 0, GNPDE_METHOD_RK4 = 1, GNPDE_METHOD_MIDPOINT = 2 };

typedef struct gnpde_rhs {
  int32_t kind;               /* GNPDE_RHS_*                                                   */
  const gnpde_graph_t* graph;
  int32_t d, ld;              /* state width and leading dimension                             */
  int32_t n_state_rows;       /* rows of the state incl. halo rows of a row-partitioned graph (0: graph->n);
                                 the projection covers all of them, the aggregation only graph->n rows */
  int32_t proj_row_begin;     /* with proj_row_end > 0: project only rows [proj_row_begin, proj_row_end)
                                 (interior / boundary passes overlapping the halo exchange) */
  int32_t proj_row_end;
  int32_t flags;              /* GNPDE_RHS_PADDED_ROWS or 0 */
  /* epilogue scalars */
  const float* alpha; const float* beta; const float* x0; int32_t alpha_sigmoid;
  /* LAPLACIAN: fixed weights, CSR order */
  const float* w_csr;
  /* TRANSFORMER / GAT: projection W [m, d] (+ bias), m = 2A (q||k) or A (GAT), and the attention
   * descriptor (its q/k/ldqk fields are filled by the solver from its workspace)               */
  const float* proj_w; const float* proj_b; int32_t proj_m;
  gnpde_attention_t att;
} gnpde_rhs_t;

typedef struct gnpde_solver gnpde_solver_t;

/* Workspace (device) the solver needs: stage buffers k1..k3, two stage inputs, projections,
 * attention scratch, spmm scratch. */
size_t gnpde_solver_workspace_bytes(const gnpde_rhs_t* rhs, int32_t method);

/* dts[n_steps] are the step sizes of the time grid (host array; the last one may be short).
 * The descriptor, and the device arrays it points to, must stay alive as long as the solver. */
int gnpde_solver_create(gnpde_solver_t** out, const gnpde_rhs_t* rhs, int32_t method,
                        const float* dts, int32_t n_steps, void* workspace, size_t workspace_bytes);

/* Integrates y (in place, [n, ld]) over the whole grid.  With use_graph != 0 the launches are
 * captured once into a hipGraph (keyed on the y pointer) and replayed. */
int gnpde_solver_run(gnpde_solver_t* s, float* y, int32_t use_graph, void* stream);

/* Recorded solve -- training with opt['adjoint'] off through a fixed-grid method (the reference's default: src/base_classes.py:44-47
 * picks torchdiffeq.odeint, run_GNN.py:62-96 calls loss.backward() through its Python loop; src/block_constant.py:45-62).  With a tape
 * attached every stage input of the following runs is written to a slot of its own (n_evals + 1 state-sized slots, 256-byte aligned
 * strides: slot 0 = y0, slot i = the input of evaluation i, the last slot = y(T)) instead of a recycled stage buffer, so the record
 * costs no extra pass over the state.  GRAND-nl with scaled-dot scores also records, behind the slots, every evaluation's q||k
 * projection [n, 2A] and head-mean weights [e] (the evaluation writes them there instead of into its scratch regions), so that the
 * reverse sweep runs neither the projection nor the attention again.  The caller hands over ZERO-FILLED memory (padding columns are
 * read by 16-byte lanes).
 * tape == NULL detaches.  The reverse sweep is gnpde_adjoint_set_tape + gnpde_adjoint_run below. */
size_t gnpde_solver_tape_bytes(const gnpde_rhs_t* rhs, int32_t method, int32_t n_steps);
int    gnpde_solver_set_tape(gnpde_solver_t* s, void* tape, size_t tape_bytes);

/* The q||k projection of SpGraphTransAttentionLayer (nn.Linear Q and K, reference src/function_transformer_attention.py:174-175) as TWO
 * tables q [n, A], k [n, A] instead of interleaved rows [n, 2A]: when a key row is shorter than a 128-byte cache line (A <= 16) the
 * attention's gathers of k rows otherwise fetch the q half of every line.  Offered (gnpde_linear_split_supported != 0) for tall x,
 * d = 64 / 128, m = 2A = 32, 16-byte aligned operands; the fixed-step / dopri5 solvers use it by themselves for whole-graph GRAND-nl
 * descriptors with scaled-dot scores (gnpde_attention_t.q / k / ldqk already describe either layout). */
int gnpde_linear_split_supported(const float* x, int64_t n, int32_t d, int32_t ldx, const float* W, int32_t m, int32_t ldw, int32_t split);
int gnpde_linear_split(const float* x, int32_t n, int32_t d, int32_t ldx, const float* W, int32_t m, int32_t ldw, const float* b,
                       float* out_q, float* out_k, int32_t split, void* stream);

/* One un-fused evaluation out = f(u) of the same descriptor (what ODEFunc.forward returns). */
int gnpde_rhs_eval(const gnpde_rhs_t* rhs, const float* u, float* out, void* workspace,
                   size_t workspace_bytes, void* stream);
size_t gnpde_rhs_workspace_bytes(const gnpde_rhs_t* rhs);

/* ------------------------------------------------------------------------------------------------
 * Native adjoint solve: the backward pass of ODEblock.forward under opt['adjoint'] (reference src/base_classes.py:44-47,
 * src/block_constant.py:45-55 -> torchdiffeq.odeint_adjoint) for the fixed-grid adjoint methods (opt['adjoint_method'] euler /
 * rk4 == 3/8 rule, opt['adjoint_step_size']).  torchdiffeq integrates  y' = f,  a' = -a^T df/dy,  g' = -a^T df/dtheta  backwards
 * in time (s = -t) and gets f and the vector-Jacobian products from autograd through ODEFunc.forward
 * (src/function_transformer_attention.py:38-53, src/function_laplacian_diffusion.py:38-51); here one stage is a fixed sequence
 * of this library's kernels (projection, attention, aggregation with the NEXT stage input formed in its epilogue, SDDMM,
 * normaliser backward, head-SpMMs, aggregation on the transposed CSR, one pass for the parameter gradients) and the whole
 * backward solve is one captured hipGraph.  rhs: GNPDE_RHS_LAPLACIAN (constant weights) or GNPDE_RHS_TRANSFORMER with
 * scaled-dot scores (any normaliser), alpha_sigmoid = 1, d <= 256 in 16-byte lanes.
 *   graph_t      CSR of the transposed operator over the SAME edge list;
 *   t_from_csr   [e] device (GRAND-nl): CSR position in rhs->graph of the entry stored at position p of graph_t;
 *   proj_wt      [d, 2A] device (GRAND-nl): rhs->proj_w transposed;   w_t_csr  [e] device (GRAND-l): the weights in graph_t's order.
 * gnpde_adjoint_run: y [n, ld] in: y(t1), out: the state integrated back to t0;  a [n, ld] in: dL/dy(t1), out: dL/dy(t0);
 * grads [gnpde_adjoint_grad_floats] out: GRAND-nl  d[Wq;Wk] [2A, d], d[bq;bk] [2A], d alpha_train, d beta_train;
 *                                        GRAND-l   d alpha_train, d beta_train.
 * dts[n_steps]: the step sizes of the reversed-time grid (torchdiffeq's fixed grid over [-t1, -t0]). */
typedef struct gnpde_adjoint gnpde_adjoint_t;
int    gnpde_adjoint_grad_floats(const gnpde_rhs_t* rhs);
size_t gnpde_adjoint_workspace_bytes(const gnpde_rhs_t* rhs, const gnpde_graph_t* graph_t, int32_t method);
int    gnpde_adjoint_create(gnpde_adjoint_t** out, const gnpde_rhs_t* rhs, const gnpde_graph_t* graph_t, const int32_t* t_from_csr,
                            const float* proj_wt, const float* w_t_csr, int32_t method, const float* dts, int32_t n_steps,
                            void* workspace, size_t workspace_bytes);
int    gnpde_adjoint_run(gnpde_adjoint_t* s, float* y, float* a, float* grads, int32_t use_graph, void* stream);
/* Reverse sweep through a RECORDED forward solve (gnpde_solver_set_tape; what autograd does through torchdiffeq's fixed-grid loop,
 * fixed_grid.py / rk_common.py rk4_alt_step_func, when opt['adjoint'] is off).  The object is created with the FORWARD method
 * (euler, midpoint or rk4) and the forward grid's dts; with a tape attached gnpde_adjoint_run ignores y's contents and maps
 * a = dL/dy(T) to dL/dy0, grads as above.  r_acc (GRAND-l, nullable) [e]: receives sum over the evaluations of
 * (b_j h) u_a[row] . u_y[col] in the CSR order of rhs->graph -- times alpha' the gradient of the edge weights (attention block).
 * tape == NULL detaches. */
int    gnpde_adjoint_set_tape(gnpde_adjoint_t* s, const void* tape, size_t tape_bytes, float* r_acc, const int32_t* csr_from_t);
/* csr_from_t (nullable) [e] device: position in graph_t of the entry stored at CSR position q of rhs->graph (the inverse of t_from_csr).  With
 * it -- and always for GRAND-l -- the sweep gathers the COTANGENT rows only: per recorded evaluation one row kernel on graph_t with the roles
 * exchanged (S = alpha' (A^T u_a - u_a), the edge products in graph_t's order, <u_y, S>), the attention backward, and one pass that closes
 * the stage algebra with S + P; the recorded state rows are read as own rows, never gathered.  gnpde_adjoint_tape_swapped: 1 when that
 * form runs -- r_acc then holds the products in GRAPH_T's order. */
int    gnpde_adjoint_tape_swapped(const gnpde_adjoint_t* s);
int    gnpde_adjoint_num_rhs_evals(const gnpde_adjoint_t* s);
int    gnpde_adjoint_destroy(gnpde_adjoint_t* s);

/* f(u) of a descriptor with an arbitrary epilogue / stage (building block of host-controlled adaptive
 * solvers); the epilogue's alpha / beta / x0 / alpha_sigmoid fields are taken from the descriptor. */
int gnpde_rhs_stage(const gnpde_rhs_t* rhs, const float* u, const gnpde_epilogue_t* epi, void* workspace,
                    size_t workspace_bytes, void* stream);

/* Error ratio of an embedded Runge-Kutta step, on device (torchdiffeq _compute_error_ratio with the rms norm):
 *   err = sum_j coef[j] * k[j],  tol = atol + rtol * max(|y0|, |y1|),  *ratio = sqrt(mean((err / tol)^2)).
 * Deterministic two-level reduction with the squares summed in double (the norm does not depend on the order of the rows beyond
 * ~1e-16, so a solve on a relabelled graph takes the same decisions); workspace: 4096 floats (16 KB), 8-byte aligned. */
int gnpde_rk_error_ratio(const float* y0, const float* y1, const float* const* k, const float* coef, int32_t n_k,
                         float atol, float rtol, int64_t n, int32_t d, int32_t ld, float* ratio, float* workspace,
                         void* stream);

/* End-point interpolation of an accepted Dormand-Prince step (torchdiffeq _interp_fit / _interp_evaluate, the quartic
 * through y0, y1, the mid-point estimate and the two end slopes) in one pass:
 *   y_mid = y0 + sum_j mid_coef[j] * k[j]   (7 stage derivatives; mid_coef[j] = fl32(c_mid_j) * fl32(dt))
 *   out   = value of the quartic at fraction x = (t_out - t0) / (t1 - t0) of the step, h = fl32(dt). */
int gnpde_dopri5_interp(const float* y0, const float* y1, const float* const* k, const float* mid_coef, float h, float x,
                        int64_t n, int32_t d, int32_t ld, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Early-stopping evaluator  [replaces EarlyStopRK4.evaluate / EarlyStopDopri5.evaluate + test / test_OGB + the
 * best-validation bookkeeping, reference src/early_stop_solver.py:100-128, :156-157, :178-218]
 *
 * After a solver step: logits = relu(y[:, 0:d_dec]) W^T + b, prediction = first arg-max over the classes,
 * hits counted per split; the step with the strictly largest validation count so far is remembered.  No host
 * synchronisation: the counters live in a device `state` of GNPDE_EARLY_STATE_INTS int32:
 *   [0..2] train / val / test hits of the latest evaluation
 *   [3..5] train / val / test hits of the best step     [6] its `step` tag     [7] evaluations done
 *   [8..]  scratch: per-block partial counts (summed in a fixed order, no atomics)
 * Accuracies are hits / split size (the host knows the sizes).  `trace` (nullable, [trace_capacity][4] int32)
 * receives {train, val, test, step} of every evaluation, in order.
 * ---------------------------------------------------------------------------------------------- */
#define GNPDE_EARLY_STATE_INTS (8 + 3 * 2048)

typedef struct {
  const float* weight;    /* device [n_classes, d_dec] row-major: nn.Linear weight of the decoder m2              */
  const float* bias;      /* device [n_classes] or NULL                                                            */
  const int32_t* labels;  /* device [n]                                                                            */
  const uint8_t* split;   /* device [n]: bit 0 train_mask, bit 1 val_mask, bit 2 test_mask                         */
  int32_t n_classes;      /* <= 64                                                                                 */
  int32_t d_dec;          /* decoder input width; < d when the state is augmented (early_stop_solver.py:180-181)   */
} gnpde_decoder_t;

int gnpde_early_stop_reset(int32_t* state, void* stream);
int gnpde_early_stop_eval(const gnpde_decoder_t* dec, const float* y, int32_t d, int32_t ld, int32_t n, int32_t step,
                          int32_t* state, int32_t* trace, int32_t trace_capacity, void* stream);

/* Attach the evaluator to a fixed-step solver: every gnpde_solver_run then resets `state`, and evaluates the
 * state after each step (step tag = 1-based index into the time grid), all inside the same hipGraph.  Call
 * before the first run or between runs (drops a captured graph); dec == NULL detaches. */
int gnpde_solver_set_early_stop(gnpde_solver_t* s, const gnpde_decoder_t* dec, int32_t* state, int32_t* trace,
                                int32_t trace_capacity);

/* out[i] = base[i] + sum_j coef[j] * v[j][i]  for i < n, n_v <= GNPDE_MAX_PREV (host arrays of device pointers /
 * coefficients): the stage algebra of host-driven Runge-Kutta loops (adjoint solve) in one pass.  out may alias
 * base or any v[j]. */
int gnpde_lincomb(const float* base, const float* const* v, const float* coef, int32_t n_v, int64_t n, float* out,
                  void* stream);

int gnpde_solver_num_rhs_evals(const gnpde_solver_t* s);
int gnpde_solver_destroy(gnpde_solver_t* s);

/* ------------------------------------------------------------------------------------------------
 * dopri5 with the step-size controller ON THE DEVICE  [replaces torchdiffeq 0.2.1 Dopri5Solver / RKAdaptiveStepsizeODESolver as
 * reached from ODEblock.forward with the reference's default opt['method'] = 'dopri5' (src/block_constant.py:57-62,
 * src/block_transformer_attention.py:58-63; the loop the reference re-states in src/early_stop_solver.py:30-128)]
 *
 * One TRIAL step -- first stage input, five evaluations of f whose epilogues form the next stage input, f(y1) (first-same-
 * as-last), the error ratio, the controller (accept / reject, step-size update in float64, end-point test), the quartic
 * end-point interpolation and the commit of the accepted state -- is captured ONCE as a hipGraph: the step size, the time, the
 * accept flag and the interpolation fraction live in device memory, the stage kernels read fl32(dt) through
 * gnpde_epilogue_t.coef_scale, interpolation and commit are predicated on device flags.  gnpde_dopri5_run replays that graph
 * `trials_per_sync` times between two reads of the controller state (one 64-byte copy), so the host neither launches kernels nor
 * decides anything per step; trial steps replayed after the end point has been reached leave every result untouched (they are
 * not counted).  Same rule as torchdiffeq: rms error norm, safety 0.9, factor limits 0.2 / 10, no shrinking after an accepted
 * step, Shampine's error weights, initial step of Hairer's heuristic as torchdiffeq selects it.
 *   y0 [n, ld_y0] in, y_out [n, ld_out] out (may alias y0); integrates from t0 to t1 > t0.
 *   max_evals > 0: stop launching once more evaluations than that were spent (reference opt['max_nfe']); *finished tells.
 * Synchronises the stream. */
typedef struct gnpde_dopri5 gnpde_dopri5_t;
size_t gnpde_dopri5_workspace_bytes(const gnpde_rhs_t* rhs);
int gnpde_dopri5_create(gnpde_dopri5_t** out, const gnpde_rhs_t* rhs, float rtol, float atol, void* workspace,
                        size_t workspace_bytes);
enum { GNPDE_ADAPTIVE_HEUN = 0, GNPDE_ADAPTIVE_DOPRI5 = 1 };
/* The embedded pair of the following runs: GNPDE_ADAPTIVE_DOPRI5 (default) or GNPDE_ADAPTIVE_HEUN -- torchdiffeq 0.2.1's
 * `adaptive_heun` (adaptive_heun.py; reference `--method adaptive_heun`, src/run_GNN.py): one evaluation per trial step, k1 = f(y + h k0),
 * y1 = y + h (k0 + k1) / 2, error h (k0 - k1) / 2, order 2 in the step-size rule and the initial step, the quartic end-point
 * interpolant through y + h k0 / 2, and k1 as the next step's first derivative (rk_common.py takes f1 = k[..., -1] for every pair).
 * Same controller record, batching and early stopping; the recorded solve (gnpde_dopri5_set_tape) is Dormand-Prince's only. */
int gnpde_dopri5_set_pair(gnpde_dopri5_t* s, int32_t pair);
int gnpde_dopri5_run(gnpde_dopri5_t* s, const float* y0, int32_t ld_y0, double t0, double t1, float* y_out, int32_t ld_out,
                     int32_t trials_per_sync, int32_t max_evals, int32_t* finished, void* stream);
/* Early stopping on the device controller  [replaces EarlyStopDopri5.advance + evaluate, reference src/early_stop_solver.py:
 * 82-128: the decoder, arg-max and split accuracies after EVERY trial step with three .item() reads, and the max_test_steps
 * cut].  The evaluator kernels (gnpde_early_stop_eval) are appended to the captured trial step and gated by the controller
 * record: an accepted step evaluates the new state with tag = number of accepted steps and its time goes to times[tag]
 * (times[0] = t0); trial steps rejected before the first accept evaluate the initial state once (tag 0); other rejected steps
 * re-evaluate nothing (the unchanged state cannot win the strict `val > best`).  After max_trial_steps trial steps the solve
 * stops and y_out is the state reached (reference :93-98), not an interpolation.  The host still only replays graphs and reads
 * the 96-byte record once per batch.  state / trace as in gnpde_solver_set_early_stop; dec == NULL detaches.  Call between
 * runs (drops the captured trial steps). */
int gnpde_dopri5_set_early_stop(gnpde_dopri5_t* s, const gnpde_decoder_t* dec, int32_t* state, int32_t* trace,
                                int32_t trace_capacity, double* times, int32_t times_capacity, int32_t max_trial_steps);
/* Node relabelling folded into the solve's own copies (graph.LocalityView of the Python layer: the fused solves run on the graph with
 * its nodes relabelled; no reference equivalent): order[r] = the caller's row that solver row r holds (device, int32 [n], must outlive the
 * runs).  gnpde_dopri5_run then reads y0[order[r]] into its row r and writes its row r to y_out[order[r]] (y_out must not alias y0
 * then).  NULL restores the plain copies. */
int gnpde_dopri5_set_row_order(gnpde_dopri5_t* s, const int32_t* order);
/* of the last run: evaluations of f, accepted and rejected steps, graph launches, host synchronisations */
int gnpde_dopri5_stats(const gnpde_dopri5_t* s, int32_t* n_evals, int32_t* n_accepted, int32_t* n_rejected, int32_t* n_launches,
                       int32_t* n_syncs);
int gnpde_dopri5_destroy(gnpde_dopri5_t* s);
/* Recorded solve + reverse sweep  [replaces what torch autograd does when the reference trains with opt['adjoint'] = False -- its
 * default, and its Cora / Citeseer best_params (src/base_classes.py:44-47, src/best_params.py) -- i.e. `loss.backward()` through
 * every accepted step of torchdiffeq's dopri5 (rk_common.py _runge_kutta_step, interp.py), with the step sizes constants of the
 * backward pass (misc.py _optimal_step_size runs under torch.no_grad)].  GRAND-l descriptors only (f linear in the state; the
 * edge weights are constants of the solve that carry gradients: src/block_transformer_attention.py:36-72).
 *   gnpde_dopri5_set_tape       ZERO-FILLED device memory of gnpde_dopri5_tape_bytes(rhs, capacity_steps) bytes, 256-byte aligned;
 *                               every ACCEPTED trial step of the following runs leaves its stage inputs u_0..u_5 (and y1 as u_0 of
 *                               the next slot) and its step size there (one copy kernel inside the captured trial step, gated by the
 *                               controller record).  A run that accepts more than capacity_steps steps returns GNPDE_EWS.  NULL
 *                               detaches.  Call between runs (drops the captured trial steps).
 *   gnpde_dopri5_tape_steps     accepted steps of the last recorded run that reached t1 (0: nothing to differentiate)
 *   gnpde_dopri5_tape_backward  given dL/d(y_out) of the LAST run (grad_out [n, ld_go]): dL/dy0 -> grad_y0 [n, ld_gy0];
 *                               r_t [e] = sum over all 6 S + 1 evaluations k = f(u) of  u[row'] . G[col']  in the CSR order of
 *                               graph_t (G = gradient reaching k): dL/dw_e = alpha' r; sum_g [n, ld] = sum of the G (dL/dbeta =
 *                               <sum_g, x0>); dot_out[0] = sum <G, k - beta x0> (dL/dalpha_train = that * (1 - alpha') under the
 *                               sigmoid).  graph_t / w_t: the transposed graph (gnpde_graph_t of the flipped edge list) and the weights
 *                               of the descriptor in ITS CSR order.  One launch of the fused row kernel per evaluation + two small
 *                               combinations per step, no host synchronisation; the algebra is written out in oracle/tape_reverse.py. */
size_t gnpde_dopri5_tape_bytes(const gnpde_rhs_t* rhs, int32_t capacity_steps);
int gnpde_dopri5_set_tape(gnpde_dopri5_t* s, void* tape, size_t tape_bytes, int32_t capacity_steps);
int gnpde_dopri5_tape_steps(const gnpde_dopri5_t* s);
/* host arrays: the float32 step sizes of the accepted steps of the last recorded run and the fraction of the last step at which t1 lies */
int gnpde_dopri5_tape_record(const gnpde_dopri5_t* s, float* h_out, int32_t capacity, float* x_out);
size_t gnpde_dopri5_tape_backward_workspace_bytes(const gnpde_dopri5_t* s, const gnpde_graph_t* graph_t);
int gnpde_dopri5_tape_backward(gnpde_dopri5_t* s, const gnpde_graph_t* graph_t, const float* w_t, const float* grad_out,
                               int32_t ld_go, float* grad_y0, int32_t ld_gy0, float* r_t, float* sum_g, float* dot_out,
                               void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Adjoint with an ADAPTIVE adjoint method, the controller ON THE DEVICE  [replaces torchdiffeq 0.2.1's odeint_adjoint backward pass
 * with `adjoint_method` 'adaptive_heun' -- the reference's DEFAULT (src/run_GNN.py:334; best_params Pubmed) -- or 'dopri5' (best_params
 * CoauthorCS, Computers) as reached from src/base_classes.py:44-47: the augmented state (vjp_t, y, a, g_theta) integrated backwards as one
 * flat vector through autograd (adjoint.py augmented_dynamics), AdaptiveHeunSolver / Dopri5Solver with the mixed norm of misc.py]
 *
 * GRAND-l descriptors only: for f(u) = alpha' (A u - u) + beta x0 the system decouples (reversed time s = -t) into
 *   y' = -f(y),  a' = alpha' (A^T a - a),  g_alpha' = (1 - alpha') <a, f(y) - beta x0>,  g_beta' = <a, x0>.
 * One TRIAL step of the embedded pair = one hipGraph: per stage the fused row kernel on the graph (f at the stage input, the next stage
 * input in its epilogue, the dots) and the aggregation on the transposed graph, then two error norms, control, finish
 * (csrc/adjoint_adaptive.hip); accept / reject, step size (float64), mixed norm (largest of rms(y), rms(a), |g_alpha|, |g_beta| errors
 * over their tolerances), end-point interpolation decided by kernels from a 112-byte record the host reads once per batch of trial steps.
 * torchdiffeq's rule throughout (safety 0.9, factors 0.2 / 10, the pair's order, the last stage derivative carried over as the next
 * step's first).
 *   rhs: f on the graph (kind GNPDE_RHS_LAPLACIAN; x0 / beta optional); graph_t / w_t: the transposed graph and the weights in ITS CSR
 *   order; method: GNPDE_ADAPTIVE_*; rtol / atol: the adjoint tolerances.
 *   gnpde_adjoint_adaptive_run: y [n, ld_y] = the state at the LATER time (read), a [n, ld_a] in: dL/dy there, out: at the earlier time;
 *   g (device float[2]) in / out: accumulated (g_alpha, g_beta); integrates s from s0 to s1 > s0 (= -t) starting with step dt0 (the
 *   caller selects it: misc.py _select_initial_step, two evaluations).  max_evals / *finished as gnpde_dopri5_run.  Synchronises the stream
 *   while it runs (one read per batch); the final copies are queued behind. */
/* (GNPDE_ADAPTIVE_HEUN / GNPDE_ADAPTIVE_DOPRI5: declared with gnpde_dopri5_set_pair above) */
typedef struct gnpde_adjoint_adaptive gnpde_adjoint_adaptive_t;
size_t gnpde_adjoint_adaptive_workspace_bytes(const gnpde_rhs_t* rhs, const gnpde_graph_t* graph_t, int32_t method);
int gnpde_adjoint_adaptive_create(gnpde_adjoint_adaptive_t** out, const gnpde_rhs_t* rhs, const gnpde_graph_t* graph_t, const float* w_t,
                                  int32_t method, float rtol, float atol, void* workspace, size_t workspace_bytes);
int gnpde_adjoint_adaptive_run(gnpde_adjoint_adaptive_t* s, const float* y, int32_t ld_y, float* a, int32_t ld_a, float* g, double s0,
                               double s1, double dt0, int32_t trials_per_sync, int32_t max_evals, int32_t* finished, void* stream);
int gnpde_adjoint_adaptive_stats(const gnpde_adjoint_adaptive_t* s, int32_t* n_evals, int32_t* n_accepted, int32_t* n_rejected,
                                 int32_t* n_launches, int32_t* n_syncs);
int gnpde_adjoint_adaptive_destroy(gnpde_adjoint_adaptive_t* s);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU halo exchange helpers (row-partitioned graph, one process per GPU, RCCL between).
 * ---------------------------------------------------------------------------------------------- */
/* dst[i, 0:d] = src[idx[i], 0:d]  for i < count  (pack boundary rows into a send buffer)        */
int gnpde_gather_rows(const float* src, int32_t ld_src, const int32_t* idx, int32_t count, int32_t d,
                      float* dst, int32_t ld_dst, void* stream);

/* Measurement aid, no reference equivalent (bench.py's `roofline.ceiling`): out[i] = sum_{t<k} table[idx[i*k + t]] for
 * i < n_out -- a perfectly balanced gather of whole rows of d floats (d % 4 == 0, <= 256) with the row width, table and launch
 * geometry of the aggregation kernels and no arithmetic but the adds.  Its rate on the table and mean degree of a workload is
 * the ceiling the aggregation is reported against when the gathered table is cache-resident (torch_sparse.spmm's gather,
 * src/function_transformer_attention.py:25-36, stripped of everything else). */
int gnpde_gather_ceiling(const float* table, int32_t n_rows, int32_t d, int32_t ld, const int32_t* idx, int32_t k,
                         float* out, int32_t n_out, int32_t variant /* 0: ids loaded per lane, 1: coalesced + shuffle */, void* stream);

/* Measurement aid, no reference equivalent (bench.py's `roofline.stream_read_probe`): a coalesced streaming read of
 * n_floats floats, `passes` times inside one launch (16-byte lanes, grid stride, eight loads in flight per lane), one float
 * per workgroup written to sink
 * (n_sink >= 2048).  The rate the memory side delivers L2-cold lines of a table at -- the hardware ceiling next to which the
 * aggregation's gather of the same table (torch_sparse.spmm's index_select, src/function_transformer_attention.py:25-36) is
 * reported when that table is resident in the Infinity Cache. */
int gnpde_stream_read(const float* table, int64_t n_floats, int32_t passes, float* sink, int32_t n_sink, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Edge-set bookkeeping of the hard-attention / rewiring blocks (once per training forward):
 *   threshold = torch.quantile(score, q);  mask = score > threshold;  edge_index[:, mask];  kept scores renormalised by their
 *   sum per endpoint (reference src/block_transformer_hard_attention.py:48-66, src/block_transformer_rewiring.py:40-50,154-183).
 * gnpde_quantile: *out (device) = the q-quantile of v[0..n) with torch.quantile's default linear interpolation, evaluated in
 *   float32 exactly as torch does (rank = fl32(q) * fl32(n-1)); a radix select, no sort; n is not limited to 16 M.
 * gnpde_threshold_edges: stable compaction of the columns of edge_index ([2, n_edges] int64, row-major) whose score exceeds
 *   *threshold (device scalar) into out_edge_index ([2, n_edges] capacity, same row stride), out_weight[i] = score / (sum of the
 *   kept scores with the same endpoint edge_index[norm_idx] + 1e-16); *out_count (device int64) = number of kept edges.
 * ---------------------------------------------------------------------------------------------- */
size_t gnpde_quantile_workspace_bytes(void);
int gnpde_quantile(const float* v, int64_t n, double q, float* out, void* workspace, size_t workspace_bytes, void* stream);
size_t gnpde_threshold_edges_workspace_bytes(int64_t n_edges, int32_t n_nodes);
int gnpde_threshold_edges(const int64_t* edge_index, const float* score, int64_t n_edges, const float* threshold,
                          int32_t norm_idx, int32_t n_nodes, int64_t* out_edge_index, float* out_weight, int64_t* out_count,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Two-hop densification of the rewiring block (new_edges = 'k_hop_att', reference src/block_transformer_rewiring.py:68-86):
 *   S = coalesce(A ++ offdiag(A A)) / 2, i.e. torch_sparse.spspmm(A, A) -> remove_self_loops -> cat with A -> / 2 ->
 *   torch_sparse.coalesce(op='add'), in one row-wise kernel without materialising the products.  A is given in CSR (rowptr [n+1],
 *   col, w in CSR order; duplicates allowed and summed).  Two phases, as the output size is data dependent:
 *   gnpde_two_hop_count -> out_rowptr [n+1] (device int64; out_rowptr[n] = nnz(S)); the caller reads the total and allocates
 *   gnpde_two_hop_fill  -> out_edge_index ([2, out_ld] int64, row-major, first nnz columns written) and out_weight, entries ordered
 *                          by (row, col) as coalesce orders them.  Sums are formed in CSR order: run-to-run identical. */
size_t gnpde_two_hop_workspace_bytes(int32_t n_nodes);
int gnpde_two_hop_count(const int32_t* rowptr, const int32_t* col, int32_t n_nodes, int64_t* out_rowptr, void* workspace,
                        size_t workspace_bytes, void* stream);
int gnpde_two_hop_fill(const int32_t* rowptr, const int32_t* col, const float* w, int32_t n_nodes, const int64_t* out_rowptr,
                       int64_t* out_edge_index, int64_t out_ld, float* out_weight, void* workspace, size_t workspace_bytes,
                       void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row-partitioned solve over the GPUs of one node: one process per GPU, RCCL point-to-point halo exchange once per
 * evaluation of f, the whole solve (pack, grouped send/recv on a second stream, interior rows, boundary rows, every
 * stage of every step) captured per rank in ONE hipGraph.  The reference has no counterpart (single device;
 * nn.DataParallel replicas in src/ray_tune.py:65-66); this is the multi-GPU form of the solver loop of
 * src/block_constant.py:57-62 that BASELINE.json's north star asks for.
 *
 * RCCL is bound at run time: gnpde_comm_load_library(path) names the librccl to use (NULL / never called: the first
 * of "librccl.so.1", "librccl.so" that loads; a library already in the process is reused).
 * Bootstrap: rank 0 calls gnpde_comm_get_unique_id, ships the GNPDE_COMM_ID_BYTES to the other ranks by any means
 * (the Python host broadcasts them through torch.distributed), every rank calls gnpde_comm_create (collective).
 * ---------------------------------------------------------------------------------------------- */
#define GNPDE_COMM_ID_BYTES 128
typedef struct gnpde_comm gnpde_comm_t;
int gnpde_comm_load_library(const char* path);
int gnpde_comm_get_unique_id(void* id_out /* GNPDE_COMM_ID_BYTES */);
int gnpde_comm_create(gnpde_comm_t** out, const void* id, int32_t rank, int32_t world);
int gnpde_comm_destroy(gnpde_comm_t* comm);

/* What this rank exchanges per evaluation.  Local row numbering of the state: [interior rows | boundary rows | halo rows
 * grouped by owning rank, ascending]; rows [0, n_own) are owned.  send_idx (device) lists, grouped by destination rank,
 * the owned rows each peer needs (send_counts[p] of them for rank p); recv_counts[p] rows arrive from rank p and land,
 * in rank order, in the halo region [n_own, n_own + n_halo) of the stage buffer -- nothing is unpacked. */
typedef struct gnpde_halo {
  int32_t world, rank;
  int32_t n_own, n_halo;
  const int32_t* send_idx;      /* device [sum(send_counts)]  */
  const int32_t* send_counts;   /* host [world]               */
  const int32_t* recv_counts;   /* host [world]               */
} gnpde_halo_t;

/* P2P transport (the default of the Python layer): the stage buffers of every rank live in IPC-shared device memory; a
 * push kernel stores this rank's boundary rows straight into the peers' halo regions over xGMI and raises an epoch flag
 * in each peer's flag array, a one-block wait kernel in front of the boundary pass polls the flags (bounded spin).  No
 * library call and no host involvement per evaluation: the per-rank hipGraph holds kernel and memcpy nodes only.
 * Several ranks may share one device (the processes map each other's memory the same way).
 *   gnpde_p2p_create      allocates n_buffers (4 for rk4, 2 for euler; a fifth for gnpde_sharded_solver_set_general) stage buffers of buffer_bytes >= (n_own + n_halo) * d * 4
 *                         and the flag array (fine-grained memory)
 *   gnpde_p2p_get_handle  -> GNPDE_P2P_HANDLE_BYTES to be all-gathered over the ranks by the host
 *   gnpde_p2p_connect     handles of ALL ranks, [world][GNPDE_P2P_HANDLE_BYTES]; maps the peers' memory */
#define GNPDE_P2P_HANDLE_BYTES 128
typedef struct gnpde_p2p gnpde_p2p_t;
int   gnpde_p2p_create(gnpde_p2p_t** out, int32_t rank, int32_t world, size_t buffer_bytes, int32_t n_buffers);
int   gnpde_p2p_get_handle(gnpde_p2p_t* p2p, void* handle_out);
int   gnpde_p2p_connect(gnpde_p2p_t* p2p, const void* handles);
void* gnpde_p2p_buffer(gnpde_p2p_t* p2p, int32_t b);      /* device pointer of local stage buffer b */
int   gnpde_p2p_destroy(gnpde_p2p_t* p2p);

typedef struct gnpde_sharded_solver gnpde_sharded_solver_t;

/* rhs_interior: descriptor over the graph view holding the rows without halo neighbours (graph->n = their count;
 * projection rows [0, n_own)); rhs_boundary: the view holding the other owned rows (graph->n = n_own, graph->row_begin =
 * number of interior rows; projection rows [n_own, n_own + n_halo)).  Both with n_state_rows = n_own + n_halo,
 * ld == d, the same kind / scalars / x0 ([n_own, d]).
 * gnpde_sharded_solver_create: RCCL transport (comm may be NULL when nothing is exchanged, world 1); EAGER launches only:
 * capturing the grouped send/recv on a forked stream crashes the HIP 7.0 runtime bundled with torch 2.10.
 * gnpde_sharded_solver_create_p2p: P2P transport; peer_halo_row0[p] = row of peer p's stage buffers where THIS rank's rows
 * start (= n_own of p + the recv counts of p for ranks below this one), peer_buffer_bytes[p] = p's buffer_bytes (aligned
 * up to 256). */
size_t gnpde_sharded_solver_workspace_bytes(const gnpde_halo_t* halo, const gnpde_rhs_t* rhs_interior,
                                            const gnpde_rhs_t* rhs_boundary, int32_t method, int32_t p2p);
int gnpde_sharded_solver_create(gnpde_sharded_solver_t** out, gnpde_comm_t* comm, const gnpde_halo_t* halo,
                                const gnpde_rhs_t* rhs_interior, const gnpde_rhs_t* rhs_boundary, int32_t method,
                                const float* dts, int32_t n_steps, void* workspace, size_t workspace_bytes);
int gnpde_sharded_solver_create_p2p(gnpde_sharded_solver_t** out, gnpde_p2p_t* p2p, const gnpde_halo_t* halo,
                                    const gnpde_rhs_t* rhs_interior, const gnpde_rhs_t* rhs_boundary, int32_t method,
                                    const float* dts, int32_t n_steps, const int64_t* peer_halo_row0,
                                    const int64_t* peer_buffer_bytes, void* workspace, size_t workspace_bytes);
/* y: device state whose first n_own rows are integrated in place over the whole grid (RCCL transport: [n_own + n_halo, d],
 * the halo rows are scratch; P2P: [>= n_own, d], copied into / out of the shared stage buffer).  use_graph != 0: captured
 * once per y pointer and replayed (P2P transport only).  Every rank must call it (the exchange is collective). */
int gnpde_sharded_solver_run(gnpde_sharded_solver_t* s, float* y, int32_t use_graph, void* stream);
/* Host-side: the order in which the P2P push walks a destination-grouped send list of send_counts[0..world) rows --
 * order[w] = slot copied by the w-th wavefront; the destinations are interleaved in proportion to their counts so that every
 * xGMI link of the rank is busy for the whole push (walked group by group, one link would carry it all while six idle).  A
 * permutation of [0, sum send_counts) that keeps each destination's rows in order.  What gnpde_sharded_solver_create_p2p uses;
 * exported for the host tests. */
int gnpde_push_order(const int32_t* send_counts, int32_t world, int32_t* order);
/* P2P transport: compute the boundary rows in n_chunks consecutive row ranges and push each range's rows of the stage OUTPUT --
 * the next evaluation's input -- into the peers' halo regions right behind it, on the side stream, while the next range is
 * computed; the last range's push raises the next evaluation's epoch.  The first evaluation of a solve still pushes its whole
 * input, the last one pushes nothing.  What is left to hide behind the next interior pass is one range's push instead of the
 * whole exchange.  rhs_chunks[c]: descriptor of range c (same kind / d / ld as the boundary descriptor; graph rows
 * [row_begin, n) continue where range c-1 ended, the ranges tile [n_interior, n_own); for the transformer kind range 0
 * projects the halo rows and the later ones an empty slice).  push_order: permutation of the send slots grouped by range --
 * slots [push_chunk_ptr[c], push_chunk_ptr[c+1]) hold the rows range c computes (rows of the interior pass that peers read
 * ride with range 0).  Same arithmetic in the same order as the unchunked solve: results are bit-identical.  n_chunks = 0
 * restores the single boundary pass.  GNPDE_EWS when the workspace has no room for a range's scratch
 * (gnpde_rhs_workspace_bytes of the range descriptors beyond the interior / boundary ones).  Drops a captured graph.
 * No reference equivalent (the reference is single-device). */
int gnpde_sharded_solver_set_boundary_chunks(gnpde_sharded_solver_t* s, const gnpde_rhs_t* const* rhs_chunks, int32_t n_chunks,
                                             const int32_t* push_order, const int32_t* push_chunk_ptr);
/* Exchange timing of the LAST run (P2P transport): for every evaluation four wall-clock stamps (s_memrealtime ticks, rate in
 * *ticks_per_second) written by the kernels themselves inside the hipGraph -- [0] the push kernel starts, [1] its last block has
 * published the epoch (all boundary rows stored into the peers' halo regions), [2] the main stream reaches the wait (interior
 * rows done), [3] every peer's rows have landed.  [1]-[0] = push duration (xGMI stores), [3]-[2] = time the boundary pass waited
 * for the exchange, i.e. the part of the exchange that compute did not hide.  With gnpde_sharded_solver_set_boundary_chunks the
 * rows of evaluation s travel during evaluation s - 1: [0] = the push behind the FIRST row range starts, [1] = the push behind
 * the last range has published the epoch, both stamped in the row of the evaluation whose INPUT they deliver.  stamps:
 * int64[capacity_evals][4]; *n_evals: the evaluations of a run (0 when the solver does not exchange).  Synchronises.  No reference
 * equivalent (measurement). */
int gnpde_sharded_solver_timing(gnpde_sharded_solver_t* s, int64_t* stamps, int32_t capacity_evals, int32_t* n_evals,
                                int64_t* ticks_per_second);

/* Synchronises.  *timed_out != 0: a wait kernel gave up polling a peer's epoch flag (results are invalid);
 * *epochs = evaluations this rank has published so far.  After the first time-out the waits of the evaluations still
 * queued return without polling (the flag is sticky for the lifetime of the gnpde_p2p_t), so a lost solve ends quickly. */
int gnpde_sharded_solver_status(gnpde_sharded_solver_t* s, int32_t* timed_out, int64_t* epochs);
int gnpde_sharded_solver_set_spin_limit(gnpde_sharded_solver_t* s, int64_t max_spins);
int gnpde_sharded_solver_num_rhs_evals(const gnpde_sharded_solver_t* s);
int gnpde_sharded_solver_destroy(gnpde_sharded_solver_t* s);

/* Normalisers that are not row-local over the row partition, inside the per-rank graph  [replaces the Python-driven evaluation that ran
 * until ABI 6: reference src/function_transformer_attention.py:190-213 with attention_norm_idx = 1 and / or squareplus
 * (src/utils.py:179-208), src/function_GAT_attention.py with attention_norm_idx = 1].  After this call every evaluation of the solver is:
 * push of the boundary rows; projection of own + halo rows into qk; pass 1 of gnpde_edge_attention_pass on att_graph (every local node
 * a segment) [squareplus: the MAXIMUM of the global score maximum over the ranks, one word exchanged through the flag blocks];
 * pass 2 [normalised over columns: the partial statistics of the halo columns are pushed into their owners' in_part rows, merged
 * there peer by peer in rank order (the arithmetic of gnpde_segment_stats_merge), and the totals are pushed back with the state's
 * pattern]; pass 3 into w; the aggregation over spmm_graph (the owned rows) with the solver's stage epilogue.  Every rank issues the
 * same pushes in the same order, so the one epoch sequence of the transport orders all of them.
 *   att: type / heads / att_dim / norm_idx / square_plus / leaky_slope / gat_a / output_var / lengthscale / edge_w_csr (att_graph's CSR
 *        order); q, k, ldqk are filled per evaluation (transformer: q = qk, k = qk + att_dim, ldqk = 2 att_dim; GAT: q = k = qk).
 *   qk [n_own + n_halo, proj_m], w [entries of att_graph], stats_send [n_halo, 2 heads], the two workspaces: device memory of the caller.
 *   stats_buffer: index (>= 4) of the shared buffer of gnpde_p2p_create that holds [S: (n_own + n_halo) x 2 heads | in_part: n_send x
 *        2 heads] on every rank; in_offset / peer_in_offset[p]: byte offset of in_part on this rank / on peer p (multiples of 256);
 *        peer_buffer_bytes[p] as in gnpde_sharded_solver_create_p2p; peer_rev_row0[p]: row of peer p's in_part where THIS rank's rows
 *        start (= the send counts of p for the ranks below this one).  Only read when att->norm_idx == 1 and world > 1. */
typedef struct gnpde_general {
  const gnpde_graph_t* att_graph;
  const gnpde_graph_t* spmm_graph;
  const gnpde_attention_t* att;
  float* qk; float* w; float* stats_send;
  void* att_ws; size_t att_ws_bytes;
  void* spmm_ws; size_t spmm_ws_bytes;
  int32_t stats_buffer;
  int64_t in_offset;
  const int64_t* peer_in_offset;
  const int64_t* peer_buffer_bytes;
  const int64_t* peer_rev_row0;
} gnpde_general_t;
int gnpde_sharded_solver_set_general(gnpde_sharded_solver_t* s, const gnpde_general_t* g);

/* dopri5 over the row partition with the controller on the device  [replaces, per rank, the Python controller that ran over
 * torch.distributed until ABI 6: reference src/block_constant.py:57-62 with opt['method'] = 'dopri5' on a graph no single GPU holds].
 * `engine`: a gnpde_sharded_solver_t created with gnpde_sharded_solver_create_p2p (method GNPDE_METHOD_RK4 -- four shared stage
 * buffers --, n_steps = 0): it stays the caller's, must outlive the dopri5 object and is used by nothing else meanwhile.  The trial
 * step of gnpde_dopri5_create is captured per rank with every evaluation as push + interior rows + wait + boundary rows, and the
 * error norm (and the three norms of the initial step size) as ONE double summed over the ranks inside the stream: every rank stores
 * its partial into every peer's fine-grained flag block, waits for the peers' and adds them in rank order, so every rank's
 * controller record holds the same bits and every rank takes the same accept / reject decisions and step sizes -- no host read per
 * trial step, no collective call.  n_rows_total: rows of the WHOLE graph (the mean of the norm).  gnpde_dopri5_run then takes and
 * returns the OWNED rows ([n_own, d]); early stopping, the tape and the row order are single-GPU features (GNPDE_ESTATE here). */
size_t gnpde_dopri5_sharded_workspace_bytes(gnpde_sharded_solver_t* engine);
int gnpde_dopri5_create_sharded(gnpde_dopri5_t** out, gnpde_sharded_solver_t* engine, float rtol, float atol, int64_t n_rows_total,
                                void* workspace, size_t workspace_bytes);

#ifdef __cplusplus
}
#endif
#endif /* GNPDE_H */
