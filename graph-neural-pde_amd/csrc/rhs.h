// One right-hand-side evaluation composed from the kernels of this library (defined in solver.hip); shared by the
// single-GPU solver and the row-partitioned solver (sharded.hip).  Internal, not part of the C ABI.
#pragma once
#include "common.h"

namespace gnpde {

// Offsets of the scratch regions of one evaluation inside its workspace.  The projection comes FIRST so that two
// descriptors over the same state rows (interior / boundary pass of a partitioned graph) that are handed the same
// workspace see the SAME q||k buffer; the regions behind it are scratch that each pass overwrites.
struct RhsLayout {
  size_t proj, wmean, att, spmm, fused, total;
  size_t att_bytes, spmm_bytes, fused_bytes;
};

// Recorded solve (gnpde_solver_set_tape) of GRAND-nl: the evaluation writes its q||k projection and its head-mean weights into buffers
// of their own (the record of that evaluation) instead of the shared scratch regions, and keeps to the separate kernels (no attention
// inside the aggregation kernel), so that the reverse sweep finds both and runs neither the projection nor the attention again.
struct RhsRecord {
  float* proj;     // [n, proj_m]
  float* wmean;    // [e]
};
// floats of one evaluation's record (0: this descriptor records nothing besides its stage inputs)
inline size_t rhs_record_stride(const gnpde_rhs_t& r) {
  if (r.kind != GNPDE_RHS_TRANSFORMER || r.att.type != GNPDE_ATT_SCALED_DOT || r.graph == nullptr) return 0;
  return (align_up(static_cast<size_t>(r.graph->n) * r.proj_m * 4, 256) + align_up(static_cast<size_t>(r.graph->e > 0 ? r.graph->e : 1) * 4, 256)) / 4;
}
inline RhsRecord rhs_record_at(const gnpde_rhs_t& r, float* base, size_t eval) {
  float* p = base + eval * rhs_record_stride(r);
  return RhsRecord{p, p + align_up(static_cast<size_t>(r.graph->n) * r.proj_m * 4, 256) / 4};
}

// Where q and k of an evaluation's projection lie: interleaved rows [n, 2A] (q = base, k = base + A, stride 2A) or -- GRAND-nl with
// scaled-dot scores and key rows shorter than a cache line, whole-graph descriptors -- two tables (q = base, k = base + n A, stride A).
bool linear_split_supported(const float* x, long long n, int d, int ldx, const float* W, int m, int ldw, int split);
int launch_linear_split(const float* x, int n, int d, int ldx, const float* W, int m, int ldw, const float* b, float* out_q, float* out_k,
                        int split, hipStream_t s);
inline bool rhs_key_table(const gnpde_rhs_t& r, const float* u) {
  return r.kind == GNPDE_RHS_TRANSFORMER && r.att.type == GNPDE_ATT_SCALED_DOT && r.proj_row_end == 0 && r.n_state_rows <= r.graph->n &&
         r.proj_m == 2 * r.att.att_dim && linear_split_supported(u, r.graph->n, r.d, r.ld, r.proj_w, r.proj_m, r.d, r.att.att_dim);
}

// where the segment statistics and the squareplus maximum of the attention passes lie inside their workspace (attention.hip)
struct AttLayoutView { size_t seg_m, seg_den, gmax, total; };
AttLayoutView att_layout_view(const gnpde_graph_t* g, const gnpde_attention_t* a);

RhsLayout rhs_layout(const gnpde_rhs_t& r);
int check_rhs(const gnpde_rhs_t* r);
// Enqueue f(u) with the given epilogue; `ws` follows rhs_layout(r).
int enqueue_rhs(const gnpde_rhs_t& r, const float* u, const gnpde_epilogue_t& epi, char* ws, const RhsLayout& L,
                hipStream_t s, const Fork* fork = nullptr, const RhsRecord* record = nullptr);
// epilogue with the descriptor's alpha / beta / x0 filled in
gnpde_epilogue_t base_epilogue(const gnpde_rhs_t& r);

// The exchange engine of the row-partitioned solvers (a gnpde_sharded_solver_t with the P2P transport, sharded.hip) as the adaptive
// solver drives it (dopri5.hip, gnpde_dopri5_create_sharded): one evaluation = push of the boundary rows of `u` + interior pass +
// wait + boundary pass; one sum = the all-reduce of a double over the ranks, inside the stream (no host, no library call).
struct ShardedShape {
  int n_own, n_local, d, ld, world, n_buffers;
  bool p2p;
  size_t buffer_bytes;
  const gnpde_rhs_t* rhs;      // the interior descriptor: alpha / beta / x0 / widths of the owned rows
};
ShardedShape sharded_shape(gnpde_sharded_solver_t* s);
float* sharded_stage_buffer(gnpde_sharded_solver_t* s, int b);
int sharded_prepare_adaptive(gnpde_sharded_solver_t* s);        // no per-evaluation stamps, no chunked boundary pass
int sharded_enqueue_eval(gnpde_sharded_solver_t* s, float* u, const gnpde_epilogue_t& e, hipStream_t st);
// value[0] <- sum over the ranks (in rank order: the same bits on every rank) of (value[0] + ... + value[n_partials - 1])
int sharded_enqueue_sum(gnpde_sharded_solver_t* s, double* value, int n_partials, hipStream_t st);
int sharded_lost_peer(gnpde_sharded_solver_t* s, int* lost);    // synchronising read of the error word

}  // namespace gnpde
