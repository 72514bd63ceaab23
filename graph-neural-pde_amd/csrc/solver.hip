// One right-hand-side evaluation f(u) composed from the kernels of this library, and the
// fixed-step solver loop (torchdiffeq 0.2.1 `euler` / `rk4` == 3/8 rule) with the stage algebra
// fused into the aggregation epilogue and the whole time grid captured in ONE hipGraph.
//
// Replaces ODEFunc.forward (reference src/function_laplacian_diffusion.py:38-51,
// src/function_transformer_attention.py:38-53, src/function_GAT_attention.py:45-65) and the Python
// solver loop reached from ODEblock.forward (src/block_constant.py:57-62,
// src/block_transformer_attention.py:58-63): ~30 launches + ~10 elementwise launches per stage there,
// 1 (GRAND-l) or 5 (GRAND-nl) launches per stage here and a single graph launch per forward pass.
#include <vector>
#include "common.h"
#include "rhs.h"

namespace gnpde {

int launch_linear_any(const float* x, int n, int d, int ldx, const float* W, int m, int ldw, const float* b, float* out,
                      int ldo, hipStream_t s, int relu = 0);
int launch_edge_attention(const gnpde_graph_t* g, const gnpde_attention_t* at, float* w_mean_csr, float* att_edge,
                          float* prods_edge, void* ws, size_t ws_bytes, hipStream_t stream, const Fork* fork);
int launch_normalise_heads(float* qk, long long n, int ld, int att_dim, int heads, bool centre, hipStream_t s, float* inv_out = nullptr);
size_t attention_workspace_bytes(const gnpde_graph_t* g, int h, bool gat);
size_t attention_workspace_bytes_rows(const gnpde_graph_t* g, int h, bool gat, int key_rows);
size_t fused_attn_workspace_bytes(const gnpde_graph_t* g, int d, int heads);
bool fused_attn_supported(const gnpde_attention_t& at, int d, int ld, const void* u, const gnpde_epilogue_t* epi);
int launch_attn_rhs_fused(const gnpde_graph_t* g, const gnpde_attention_t* at, const float* proj_w, const float* proj_b,
                          const float* u, int d, int ld, const gnpde_epilogue_t* epi, void* ws, size_t ws_bytes,
                          hipStream_t stream);


// The projection comes FIRST so that two descriptors over the same state rows (interior / boundary pass of a
// partitioned graph) that are handed the same workspace see the SAME q||k buffer; the regions behind it are
// scratch that each pass overwrites (the passes are ordered on one stream).
RhsLayout rhs_layout(const gnpde_rhs_t& r) {
  RhsLayout L{};
  const gnpde_graph_t& g = *r.graph;
  size_t off = 0;
  if (r.kind != GNPDE_RHS_LAPLACIAN) {
    const size_t prow = r.n_state_rows > g.n ? r.n_state_rows : g.n;
    L.proj = off;  off += align_up(prow * r.proj_m * 4, 256);
  }
  L.spmm = off;
  L.spmm_bytes = gnpde_spmm_workspace_bytes(&g, r.d);
  off += align_up(L.spmm_bytes, 256);
  if (r.kind != GNPDE_RHS_LAPLACIAN) {
    L.wmean = off; off += align_up(static_cast<size_t>(g.e) * 4, 256);
    L.att = off;
    L.att_bytes = attention_workspace_bytes_rows(&g, r.att.heads, r.kind == GNPDE_RHS_GAT, r.n_state_rows);
    off += align_up(L.att_bytes, 256);
    if (r.kind == GNPDE_RHS_TRANSFORMER) {
      L.fused = off;
      L.fused_bytes = fused_attn_workspace_bytes(&g, r.d, r.att.heads);
      off += align_up(L.fused_bytes, 256);
    }
  }
  L.total = off;
  return L;
}

int check_rhs(const gnpde_rhs_t* r) {
  GNPDE_CHECK_ARG(r && r->graph, GNPDE_EINVAL, "rhs: null descriptor");
  GNPDE_CHECK_ARG(r->kind >= GNPDE_RHS_LAPLACIAN && r->kind <= GNPDE_RHS_GAT, GNPDE_EINVAL, "rhs: bad kind %d", r->kind);
  GNPDE_CHECK_ARG(r->d >= 1 && r->ld >= r->d, GNPDE_EINVAL, "rhs: bad d/ld");
  GNPDE_CHECK_ARG(r->alpha != nullptr, GNPDE_EINVAL, "rhs: alpha is null");
  GNPDE_CHECK_ARG(r->x0 == nullptr || r->beta != nullptr, GNPDE_EINVAL, "rhs: x0 without beta");
  if (r->kind == GNPDE_RHS_LAPLACIAN) {
    GNPDE_CHECK_ARG(r->w_csr != nullptr || r->graph->e == 0, GNPDE_EINVAL, "rhs: laplacian needs w_csr");
  } else {
    GNPDE_CHECK_ARG(r->proj_w != nullptr && r->proj_m >= 1, GNPDE_EINVAL, "rhs: projection weights missing");
    const int need = r->kind == GNPDE_RHS_TRANSFORMER ? 2 * r->att.att_dim : r->att.att_dim;
    GNPDE_CHECK_ARG(r->proj_m == need, GNPDE_EINVAL, "rhs: proj_m=%d but attention needs %d", r->proj_m, need);
  }
  return 0;
}

// Enqueue f(u) with the given epilogue.  `ws` follows rhs_layout.
int enqueue_rhs(const gnpde_rhs_t& r, const float* u, const gnpde_epilogue_t& epi, char* ws, const RhsLayout& L,
                hipStream_t s, const Fork* fork, const RhsRecord* record) {
  const gnpde_graph_t* g = r.graph;
  const float* w = r.w_csr;
  if (record != nullptr && rhs_record_stride(r) == 0) record = nullptr;
  if (record == nullptr && r.kind == GNPDE_RHS_TRANSFORMER && fused_attn_supported(r.att, r.d, r.ld, u, &epi) &&
      reinterpret_cast<uintptr_t>(r.proj_w) % 16 == 0) {
    // scaled-dot attention, softmax over rows: projection + attention + aggregation + epilogue in one pass
    return launch_attn_rhs_fused(g, &r.att, r.proj_w, r.proj_b, u, r.d, r.ld, &epi, ws + L.fused, L.fused_bytes, s);
  }
  if (r.kind != GNPDE_RHS_LAPLACIAN) {
    float* proj = record ? record->proj : reinterpret_cast<float*>(ws + L.proj);
    float* wmean = record ? record->wmean : reinterpret_cast<float*>(ws + L.wmean);
    int p0 = 0, p1 = r.n_state_rows > g->n ? r.n_state_rows : g->n;   // keys of halo rows are recomputed locally
    if (r.proj_row_end > 0) {  // this pass projects only a slice of the state rows
      p0 = r.proj_row_begin;
      p1 = r.proj_row_end;
    }
    const bool key_table = rhs_key_table(r, u);       // q and k as two tables: the k gathers of the attention fetch whole lines of keys
    int rc = 0;
    if (key_table) rc = launch_linear_split(u, g->n, r.d, r.ld, r.proj_w, r.proj_m, r.d, r.proj_b, proj, proj + static_cast<size_t>(g->n) * r.att.att_dim,
                                            r.att.att_dim, s);
    else if (p1 > p0) rc = launch_linear_any(u + static_cast<size_t>(p0) * r.ld, p1 - p0, r.d, r.ld, r.proj_w, r.proj_m, r.d,
                                             r.proj_b, proj + static_cast<size_t>(p0) * r.proj_m, r.proj_m, s);
    if (rc) return rc;
    gnpde_attention_t at = r.att;
    at.ldqk = key_table ? r.att.att_dim : r.proj_m;
    at.q = proj;
    at.k = key_table ? proj + static_cast<size_t>(g->n) * r.att.att_dim : (r.kind == GNPDE_RHS_TRANSFORMER ? proj + r.att.att_dim : proj);
    if (r.kind == GNPDE_RHS_TRANSFORMER && (at.type == GNPDE_ATT_COSINE || at.type == GNPDE_ATT_PEARSON) &&
        at.heads >= 1 && at.att_dim % at.heads == 0) {
      // cosine_sim / pearson (reference src/function_transformer_attention.py:198-206) = the scaled dot product of unit (mean-centred)
      // head vectors: the rows just projected are normalised in place (the query side times sqrt(d_k)) and everything behind --
      // fused row kernels with their hub phases, the fused normalisers over columns / squareplus, attention inside the aggregation
      // kernel on small graphs -- is the scaled-dot path
      rc = launch_normalise_heads(proj + static_cast<size_t>(p0) * r.proj_m, p1 > p0 ? p1 - p0 : 0, r.proj_m, at.att_dim, at.heads,
                                  at.type == GNPDE_ATT_PEARSON, s);
      if (rc) return rc;
      at.type = GNPDE_ATT_SCALED_DOT;
    }
    at.n_key_rows = r.n_state_rows > g->n ? r.n_state_rows : 0;     // halo rows: the GAT node terms cover them
    const bool padded = (r.flags & GNPDE_RHS_PADDED_ROWS) != 0 && r.ld % 4 == 0;
    if (record == nullptr && r.kind == GNPDE_RHS_TRANSFORMER && fork == nullptr && r.proj_row_end == 0 && r.n_state_rows <= g->n &&
        (r.d % 4 == 0 || padded) && attn_spmm_supported(g, at, r.d, r.ld, u, epi)) {
      // scaled-dot row softmax: the short rows are attended inside the aggregation kernel; only the hub rows' weights are
      // computed ahead (two small launches)
      rc = launch_hub_attention(g, &at, wmean, ws + L.att, L.att_bytes, s);
      if (rc) return rc;
      return launch_attn_spmm(g, &at, wmean, u, r.d, r.ld, &epi, ws + L.spmm, L.spmm_bytes, s, padded);
    }
    rc = launch_edge_attention(g, &at, wmean, nullptr, nullptr, ws + L.att, L.att_bytes, s, fork);
    if (rc) return rc;
    w = wmean;
  }
  return launch_spmm_rhs(g, w, u, r.d, r.ld, &epi, nullptr, ws + L.spmm, L.spmm_bytes, s, fork,
                         (r.flags & GNPDE_RHS_PADDED_ROWS) != 0 && r.ld % 4 == 0);
}

gnpde_epilogue_t base_epilogue(const gnpde_rhs_t& r) {
  gnpde_epilogue_t e{};
  e.alpha = r.alpha;
  e.beta = r.beta;
  e.x0 = r.x0;
  e.alpha_sigmoid = r.alpha_sigmoid;
  return e;
}

}  // namespace gnpde

using namespace gnpde;

struct gnpde_solver {
  gnpde_rhs_t rhs;
  gnpde_graph_t graph;
  gnpde_graph_t graph_t;
  int method;
  std::vector<float> dts;
  char* ws;
  size_t ws_bytes;
  RhsLayout L;
  size_t off_k1, off_k2, off_k3, off_ua, off_ub, off_rhs;
  hipStream_t cap_stream = nullptr;
  Fork fork;                 // second stream + events for the hub-row branch
  hipGraph_t graph_obj = nullptr;
  hipGraphExec_t exec = nullptr;
  float* captured_y = nullptr;
  int n_evals = 0;
  bool early = false;        // early-stopping evaluator after every step
  gnpde_decoder_t dec{};
  int* early_state = nullptr;
  int* early_trace = nullptr;
  int early_trace_capacity = 0;
  float* tape = nullptr;     // recorded solve (gnpde_solver_set_tape): n_evals + 1 state-sized slots, the stage inputs in evaluation order
  float* tape_rec = nullptr; // ... followed by one RhsRecord per evaluation (GRAND-nl with scaled-dot scores: q||k and the weights)
};

namespace {

size_t solver_layout(const gnpde_rhs_t& r, int method, gnpde_solver* s) {
  const size_t state = align_up(static_cast<size_t>(r.graph->n) * r.ld * 4, 256);
  size_t off = 0;
  size_t k1 = 0, k2 = 0, k3 = 0, ua = 0, ub = 0;
  ua = off; off += state;
  if (method == GNPDE_METHOD_RK4) {
    ub = off; off += state;
    k1 = off; off += state;
    k2 = off; off += state;
    k3 = off; off += state;
  }
  const size_t rhs_off = off;
  off += rhs_layout(r).total;
  if (s) {
    s->off_ua = ua; s->off_ub = ub; s->off_k1 = k1; s->off_k2 = k2; s->off_k3 = k3; s->off_rhs = rhs_off;
  }
  return off;
}

int ensure_fork(gnpde_solver* s) {
  if (s->fork.aux != nullptr) return 0;
  GNPDE_HIP(hipStreamCreateWithFlags(&s->fork.aux, hipStreamNonBlocking));
  GNPDE_HIP(hipEventCreateWithFlags(&s->fork.e_fork, hipEventDisableTiming));
  GNPDE_HIP(hipEventCreateWithFlags(&s->fork.e_join, hipEventDisableTiming));
  return 0;
}

int enqueue_solve(gnpde_solver* s, float* y, hipStream_t st) {
  const gnpde_rhs_t& r = s->rhs;
  const Fork* fk = (s->rhs.graph->n_long_rows > 0 && g_tune[GNPDE_TUNE_FORK] == 1) ? &s->fork : nullptr;  // opt-in: cross-stream joins cost more than they hide
  if (fk != nullptr) {
    const int frc = ensure_fork(s);
    if (frc) return frc;
  }
  char* rws = s->ws + s->off_rhs;
  float* ua = reinterpret_cast<float*>(s->ws + s->off_ua);
  int step = 0;
  auto evaluate = [&](const float* state) -> int {
    ++step;
    if (!s->early) return 0;
    return enqueue_early_stop_eval(s->dec, state, r.ld, r.graph->n, step, s->early_state,
                                   s->early_trace, s->early_trace_capacity, st);
  };
  if (s->early) GNPDE_HIP(hipMemsetAsync(s->early_state, 0, 8 * sizeof(int32_t), st));
  if (s->tape != nullptr) {
    // Recorded solve: the same launches, every stage input written to a slot of its own instead of a recycled buffer -- the record
    // costs no extra pass over the state.  Slot 0 = y0, slot i = the input of evaluation i (rk4: u1..u4 of step n at 4n..4n+3),
    // the last slot = y(T), copied back into y.
    const size_t stride = align_up(static_cast<size_t>(r.graph->n) * r.ld * 4, 256) / 4;
    const size_t nbytes = static_cast<size_t>(r.graph->n) * r.ld * 4;
    auto slot = [&](size_t i) { return s->tape + i * stride; };
    GNPDE_HIP(hipMemcpyAsync(slot(0), y, nbytes, hipMemcpyDeviceToDevice, st));
    size_t at = 0;
    RhsRecord rec_store{};
    auto rec = [&](size_t eval) -> const RhsRecord* {
      if (s->tape_rec == nullptr) return nullptr;
      rec_store = rhs_record_at(r, s->tape_rec, eval);
      return &rec_store;
    };
    for (float dt : s->dts) {
      gnpde_epilogue_t e = base_epilogue(r);
      e.dt = dt;
      int rc = 0;
      if (s->method == GNPDE_METHOD_EULER) {
        e.stage = GNPDE_STAGE_EULER; e.y = slot(at); e.out_y = slot(at + 1);
        rc = enqueue_rhs(r, slot(at), e, rws, s->L, st, fk, rec(at));
        at += 1;
      } else if (s->method == GNPDE_METHOD_MIDPOINT) {
        e.stage = GNPDE_STAGE_LINCOMB; e.y = slot(at); e.n_prev = 0; e.out_k = nullptr;
        e.coef[0] = 0.5f * dt; e.out_y = slot(at + 1);
        rc = enqueue_rhs(r, slot(at), e, rws, s->L, st, fk, rec(at));
        if (rc) return rc;
        e.coef[0] = dt; e.out_y = slot(at + 2);
        rc = enqueue_rhs(r, slot(at + 1), e, rws, s->L, st, fk, rec(at + 1));
        at += 2;
      } else {
        float *u1 = slot(at), *u2 = slot(at + 1), *u3 = slot(at + 2), *u4 = slot(at + 3);
        e.stage = GNPDE_STAGE_RK1C; e.out_y = u2;
        rc = enqueue_rhs(r, u1, e, rws, s->L, st, fk, rec(at));
        if (rc) return rc;
        e.stage = GNPDE_STAGE_RK2C; e.y = u1; e.out_y = u3;
        rc = enqueue_rhs(r, u2, e, rws, s->L, st, fk, rec(at + 1));
        if (rc) return rc;
        e.stage = GNPDE_STAGE_RK3C; e.k1 = u2; e.out_y = u4;
        rc = enqueue_rhs(r, u3, e, rws, s->L, st, fk, rec(at + 2));
        if (rc) return rc;
        e.stage = GNPDE_STAGE_RK4C; e.k1 = u3; e.out_y = slot(at + 4);
        rc = enqueue_rhs(r, u4, e, rws, s->L, st, fk, rec(at + 3));
        at += 4;
      }
      if (rc) return rc;
      rc = evaluate(slot(at));
      if (rc) return rc;
    }
    GNPDE_HIP(hipMemcpyAsync(y, slot(at), nbytes, hipMemcpyDeviceToDevice, st));
    return 0;
  }
  if (s->method == GNPDE_METHOD_EULER) {
    float* cur = y;
    float* nxt = ua;
    for (float dt : s->dts) {
      gnpde_epilogue_t e = base_epilogue(r);
      e.stage = GNPDE_STAGE_EULER; e.dt = dt; e.y = cur; e.out_y = nxt;
      int rc = enqueue_rhs(r, cur, e, rws, s->L, st, fk);
      if (rc) return rc;
      float* t = cur; cur = nxt; nxt = t;
      rc = evaluate(cur);
      if (rc) return rc;
    }
    if (cur != y)
      GNPDE_HIP(hipMemcpyAsync(y, cur, static_cast<size_t>(r.graph->n) * r.ld * 4, hipMemcpyDeviceToDevice, st));
    return 0;
  }
  if (s->method == GNPDE_METHOD_MIDPOINT) {
    // torchdiffeq Midpoint._step_func: y_mid = y + f(y) * (dt / 2);  y += dt * f(y_mid)  (second stage row-local in place on y)
    for (float dt : s->dts) {
      gnpde_epilogue_t e = base_epilogue(r);
      e.stage = GNPDE_STAGE_LINCOMB; e.y = y; e.n_prev = 0; e.out_k = nullptr;
      e.coef[0] = 0.5f * dt; e.out_y = ua;
      int rc = enqueue_rhs(r, y, e, rws, s->L, st, fk);
      if (rc) return rc;
      e.coef[0] = dt; e.out_y = y;
      rc = enqueue_rhs(r, ua, e, rws, s->L, st, fk);
      if (rc) return rc;
      rc = evaluate(y);
      if (rc) return rc;
    }
    return 0;
  }
  float* ub = reinterpret_cast<float*>(s->ws + s->off_ub);
  float* k1 = reinterpret_cast<float*>(s->ws + s->off_k1);
  float* k2 = reinterpret_cast<float*>(s->ws + s->off_k2);
  float* k3 = reinterpret_cast<float*>(s->ws + s->off_k3);
  if (g_tune[GNPDE_TUNE_RK4_CLASSIC] == 0) {
    // compact stages: y, u2 (ua), u3 (ub), u4 (the k1 slot); k1..k3 are never materialised
    float* uc = k1;
    for (float dt : s->dts) {
      gnpde_epilogue_t e = base_epilogue(r);
      e.dt = dt;
      e.stage = GNPDE_STAGE_RK1C; e.out_y = ua;
      int rc = enqueue_rhs(r, y, e, rws, s->L, st, fk);
      if (rc) return rc;
      e.stage = GNPDE_STAGE_RK2C; e.y = y; e.out_y = ub;
      rc = enqueue_rhs(r, ua, e, rws, s->L, st, fk);
      if (rc) return rc;
      e.stage = GNPDE_STAGE_RK3C; e.k1 = ua; e.out_y = uc;
      rc = enqueue_rhs(r, ub, e, rws, s->L, st, fk);
      if (rc) return rc;
      e.stage = GNPDE_STAGE_RK4C; e.k1 = ub; e.out_y = y;
      rc = enqueue_rhs(r, uc, e, rws, s->L, st, fk);
      if (rc) return rc;
      rc = evaluate(y);
      if (rc) return rc;
    }
    return 0;
  }
  for (float dt : s->dts) {
    gnpde_epilogue_t e = base_epilogue(r);
    e.dt = dt; e.y = y;
    e.stage = GNPDE_STAGE_RK1; e.out_k = k1; e.out_y = ua;
    int rc = enqueue_rhs(r, y, e, rws, s->L, st, fk);
    if (rc) return rc;
    e.stage = GNPDE_STAGE_RK2; e.k1 = k1; e.out_k = k2; e.out_y = ub;
    rc = enqueue_rhs(r, ua, e, rws, s->L, st, fk);
    if (rc) return rc;
    e.stage = GNPDE_STAGE_RK3; e.k2 = k2; e.out_k = k3; e.out_y = ua;
    rc = enqueue_rhs(r, ub, e, rws, s->L, st, fk);
    if (rc) return rc;
    e.stage = GNPDE_STAGE_RK4; e.k3 = k3; e.out_k = nullptr; e.out_y = y;
    rc = enqueue_rhs(r, ua, e, rws, s->L, st, fk);
    if (rc) return rc;
    rc = evaluate(y);
    if (rc) return rc;
  }
  return 0;
}

void drop_graph(gnpde_solver* s) {
  if (s->exec) { (void)hipGraphExecDestroy(s->exec); s->exec = nullptr; }
  if (s->graph_obj) { (void)hipGraphDestroy(s->graph_obj); s->graph_obj = nullptr; }
  s->captured_y = nullptr;
}

}  // namespace

extern "C" size_t gnpde_rhs_workspace_bytes(const gnpde_rhs_t* rhs) {
  if (check_rhs(rhs)) return 0;
  return rhs_layout(*rhs).total;
}

extern "C" int gnpde_rhs_eval(const gnpde_rhs_t* rhs, const float* u, float* out, void* workspace, size_t workspace_bytes,
                              void* stream) {
  int rc = check_rhs(rhs);
  if (rc) return rc;
  GNPDE_CHECK_ARG(u && out && u != out, GNPDE_EINVAL, "rhs_eval: bad u/out");
  const RhsLayout L = rhs_layout(*rhs);
  GNPDE_CHECK_ARG(L.total == 0 || (workspace && workspace_bytes >= L.total), GNPDE_EWS, "rhs_eval: workspace %zu < %zu bytes",
                  workspace_bytes, L.total);
  gnpde_epilogue_t e = base_epilogue(*rhs);
  e.stage = GNPDE_STAGE_RHS;
  e.out_k = out;
  return enqueue_rhs(*rhs, u, e, static_cast<char*>(workspace), L, static_cast<hipStream_t>(stream));
}

extern "C" int gnpde_rhs_stage(const gnpde_rhs_t* rhs, const float* u, const gnpde_epilogue_t* epi, void* workspace,
                               size_t workspace_bytes, void* stream) {
  int rc = check_rhs(rhs);
  if (rc) return rc;
  GNPDE_CHECK_ARG(u && epi, GNPDE_EINVAL, "rhs_stage: null argument");
  const RhsLayout L = rhs_layout(*rhs);
  GNPDE_CHECK_ARG(L.total == 0 || (workspace && workspace_bytes >= L.total), GNPDE_EWS, "rhs_stage: workspace %zu < %zu bytes",
                  workspace_bytes, L.total);
  gnpde_epilogue_t e = *epi;
  e.alpha = rhs->alpha;
  e.beta = rhs->beta;
  e.x0 = rhs->x0;
  e.alpha_sigmoid = rhs->alpha_sigmoid;
  return enqueue_rhs(*rhs, u, e, static_cast<char*>(workspace), L, static_cast<hipStream_t>(stream));
}

extern "C" size_t gnpde_solver_workspace_bytes(const gnpde_rhs_t* rhs, int32_t method) {
  if (check_rhs(rhs)) return 0;
  if (method != GNPDE_METHOD_EULER && method != GNPDE_METHOD_RK4 && method != GNPDE_METHOD_MIDPOINT) return 0;
  return solver_layout(*rhs, method, nullptr);
}

extern "C" int gnpde_solver_create(gnpde_solver_t** out, const gnpde_rhs_t* rhs, int32_t method, const float* dts,
                                   int32_t n_steps, void* workspace, size_t workspace_bytes) {
  GNPDE_CHECK_ARG(out != nullptr, GNPDE_EINVAL, "solver_create: out is null");
  *out = nullptr;
  int rc = check_rhs(rhs);
  if (rc) return rc;
  GNPDE_CHECK_ARG(method == GNPDE_METHOD_EULER || method == GNPDE_METHOD_RK4 || method == GNPDE_METHOD_MIDPOINT, GNPDE_EINVAL,
                  "solver_create: bad method %d", method);
  GNPDE_CHECK_ARG(n_steps >= 0 && (dts || n_steps == 0), GNPDE_EINVAL, "solver_create: bad time grid");
  gnpde_solver* s = new gnpde_solver();
  s->rhs = *rhs;
  s->graph = *rhs->graph;
  s->rhs.graph = &s->graph;
  if (s->rhs.att.graph_t != nullptr) {   // (the descriptor's transposed graph is copied as well: the caller's structs may go away)
    s->graph_t = *s->rhs.att.graph_t;
    s->rhs.att.graph_t = &s->graph_t;
  }
  s->method = method;
  s->dts.assign(dts, dts + n_steps);
  s->L = rhs_layout(s->rhs);
  const size_t need = solver_layout(s->rhs, method, s);
  if (!(workspace && workspace_bytes >= need && reinterpret_cast<uintptr_t>(workspace) % 256 == 0)) {
    set_error("solver_create: workspace %zu bytes (need %zu, 256-byte aligned)", workspace_bytes, need);
    delete s;
    return GNPDE_EWS;
  }
  s->ws = static_cast<char*>(workspace);
  s->ws_bytes = workspace_bytes;
  s->n_evals = n_steps * (method == GNPDE_METHOD_RK4 ? 4 : method == GNPDE_METHOD_MIDPOINT ? 2 : 1);
  *out = s;
  return 0;
}

extern "C" int gnpde_solver_run(gnpde_solver_t* s, float* y, int32_t use_graph, void* stream) {
  GNPDE_CHECK_ARG(s && y, GNPDE_EINVAL, "solver_run: null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!use_graph) return enqueue_solve(s, y, st);
  if (s->exec == nullptr || s->captured_y != y) {
    drop_graph(s);
    if (s->cap_stream == nullptr) GNPDE_HIP(hipStreamCreateWithFlags(&s->cap_stream, hipStreamNonBlocking));
    GNPDE_HIP(hipStreamBeginCapture(s->cap_stream, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue_solve(s, y, s->cap_stream);
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

extern "C" size_t gnpde_solver_tape_bytes(const gnpde_rhs_t* rhs, int32_t method, int32_t n_steps) {
  if (check_rhs(rhs) || n_steps < 0) return 0;
  if (method != GNPDE_METHOD_EULER && method != GNPDE_METHOD_RK4 && method != GNPDE_METHOD_MIDPOINT) return 0;
  const size_t state = align_up(static_cast<size_t>(rhs->graph->n) * rhs->ld * 4, 256);
  const size_t per = method == GNPDE_METHOD_RK4 ? 4 : method == GNPDE_METHOD_MIDPOINT ? 2 : 1;
  return (per * static_cast<size_t>(n_steps) + 1) * state + per * static_cast<size_t>(n_steps) * rhs_record_stride(*rhs) * 4;
}

extern "C" int gnpde_solver_set_tape(gnpde_solver_t* s, void* tape, size_t tape_bytes) {
  GNPDE_CHECK_ARG(s != nullptr, GNPDE_EINVAL, "solver_set_tape: solver is null");
  drop_graph(s);
  s->tape = nullptr;
  s->tape_rec = nullptr;
  if (tape == nullptr) return 0;
  const size_t need = gnpde_solver_tape_bytes(&s->rhs, s->method, static_cast<int32_t>(s->dts.size()));
  GNPDE_CHECK_ARG(reinterpret_cast<uintptr_t>(tape) % 256 == 0 && tape_bytes >= need, GNPDE_EWS,
                  "solver_set_tape: %zu bytes (need %zu, 256-byte aligned, zero-filled)", tape_bytes, need);
  s->tape = static_cast<float*>(tape);
  const size_t state = align_up(static_cast<size_t>(s->rhs.graph->n) * s->rhs.ld * 4, 256);
  s->tape_rec = rhs_record_stride(s->rhs) > 0 ? s->tape + (static_cast<size_t>(s->n_evals) + 1) * (state / 4) : nullptr;
  return 0;
}

extern "C" int gnpde_solver_set_early_stop(gnpde_solver_t* s, const gnpde_decoder_t* dec, int32_t* state, int32_t* trace,
                                           int32_t trace_capacity) {
  GNPDE_CHECK_ARG(s != nullptr, GNPDE_EINVAL, "solver_set_early_stop: solver is null");
  drop_graph(s);
  if (dec == nullptr) {
    s->early = false;
    return 0;
  }
  int rc = check_decoder(dec, s->rhs.d);
  if (rc) return rc;
  GNPDE_CHECK_ARG(state != nullptr, GNPDE_EINVAL, "solver_set_early_stop: state is null");
  GNPDE_CHECK_ARG(trace != nullptr || trace_capacity == 0, GNPDE_EINVAL, "solver_set_early_stop: trace capacity without a trace");
  s->dec = *dec;
  s->early_state = state;
  s->early_trace = trace;
  s->early_trace_capacity = trace_capacity;
  s->early = true;
  return 0;
}

extern "C" int gnpde_solver_num_rhs_evals(const gnpde_solver_t* s) { return s ? s->n_evals : 0; }

extern "C" int gnpde_solver_destroy(gnpde_solver_t* s) {
  if (!s) return 0;
  drop_graph(s);
  if (s->cap_stream) (void)hipStreamDestroy(s->cap_stream);
  if (s->fork.aux) (void)hipStreamDestroy(s->fork.aux);
  if (s->fork.e_fork) (void)hipEventDestroy(s->fork.e_fork);
  if (s->fork.e_join) (void)hipEventDestroy(s->fork.e_join);
  delete s;
  return 0;
}
