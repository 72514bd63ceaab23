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

RhsLayout rhs_layout(const gnpde_rhs_t& r);
int check_rhs(const gnpde_rhs_t* r);
// Enqueue f(u) with the given epilogue; `ws` follows rhs_layout(r).
int enqueue_rhs(const gnpde_rhs_t& r, const float* u, const gnpde_epilogue_t& epi, char* ws, const RhsLayout& L,
                hipStream_t s, const Fork* fork = nullptr);
// epilogue with the descriptor's alpha / beta / x0 filled in
gnpde_epilogue_t base_epilogue(const gnpde_rhs_t& r);

}  // namespace gnpde
