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
// stage buffer IS the receive buffer.  RCCL is bound at run time (dlopen / dlsym): the library has no link-time
// dependency on it, single-GPU users never load it, and a Python host can hand over the very librccl its
// torch.distributed already loaded.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstring>
#include <vector>
#include "common.h"
#include "rhs.h"

namespace gnpde {
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

struct gnpde_sharded_solver {
  gnpde_comm* comm;
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
  bool warmed = false;              // connections to the peers exist (first exchange must run outside capture)
  int n_evals = 0;
};

namespace {

size_t sharded_layout(const gnpde_rhs_t& ri, const gnpde_rhs_t& rb, int method, int n_local, int n_send,
                      gnpde_sharded_solver* s) {
  const size_t state = align_up(static_cast<size_t>(n_local) * ri.ld * 4, 256);
  size_t off = 0;
  const size_t ua = off; off += state;
  size_t ub = 0, uc = 0;
  if (method == GNPDE_METHOD_RK4) {
    ub = off; off += state;
    uc = off; off += state;
  }
  const size_t send = off; off += align_up(static_cast<size_t>(n_send > 0 ? n_send : 1) * ri.d * 4, 256);
  const size_t rhs_off = off;
  const size_t ti = rhs_layout(ri).total, tb = rhs_layout(rb).total;
  off += ti > tb ? ti : tb;
  if (s) {
    s->off_ua = ua; s->off_ub = ub; s->off_uc = uc; s->off_send = send; s->off_rhs = rhs_off;
  }
  return off;
}

// pack + grouped send / recv of the halo rows of `u`; leaves e_recv recorded on the comm stream
int enqueue_exchange(gnpde_sharded_solver* s, float* u, hipStream_t st) {
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

// exchange + f(u) with the fused stage: interior rows overlap the exchange, boundary rows follow it
int enqueue_eval(gnpde_sharded_solver* s, float* u, gnpde_epilogue_t e, hipStream_t st) {
  char* rws = s->ws + s->off_rhs;
  const bool exch = s->n_send > 0 || s->n_halo > 0;
  if (exch) {
    const int rc = enqueue_exchange(s, u, st);
    if (rc) return rc;
  }
  if (s->g_int.n > 0) {
    const int rc = enqueue_rhs(s->rhs_int, u, e, rws, s->L_int, st);
    if (rc) return rc;
  }
  if (exch) GNPDE_HIP(hipStreamWaitEvent(st, s->e_recv, 0));
  if (s->g_bnd.n > s->g_bnd.row_begin) {
    const int rc = enqueue_rhs(s->rhs_bnd, u, e, rws, s->L_bnd, st);
    if (rc) return rc;
  }
  return 0;
}

int enqueue_sharded_solve(gnpde_sharded_solver* s, float* y, hipStream_t st) {
  float* ua = reinterpret_cast<float*>(s->ws + s->off_ua);
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
  float* ub = reinterpret_cast<float*>(s->ws + s->off_ub);
  float* uc = reinterpret_cast<float*>(s->ws + s->off_uc);
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

void drop_sharded_graph(gnpde_sharded_solver* s) {
  if (s->exec) { (void)hipGraphExecDestroy(s->exec); s->exec = nullptr; }
  if (s->graph_obj) { (void)hipGraphDestroy(s->graph_obj); s->graph_obj = nullptr; }
  s->captured_y = nullptr;
}

}  // namespace

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

static int check_sharded_args(const gnpde_halo_t* h, const gnpde_rhs_t* ri, const gnpde_rhs_t* rb) {
  GNPDE_CHECK_ARG(h && ri && rb, GNPDE_EINVAL, "sharded solver: null argument");
  int rc = check_rhs(ri);
  if (rc) return rc;
  rc = check_rhs(rb);
  if (rc) return rc;
  GNPDE_CHECK_ARG(h->world >= 1 && h->n_own >= 0 && h->n_halo >= 0 && h->send_counts && h->recv_counts, GNPDE_EINVAL,
                  "sharded solver: bad halo description");
  GNPDE_CHECK_ARG(ri->d == rb->d && ri->ld == rb->ld && ri->kind == rb->kind, GNPDE_EINVAL,
                  "sharded solver: interior / boundary descriptors disagree");
  GNPDE_CHECK_ARG(ri->ld == ri->d, GNPDE_ESHAPE, "sharded solver: the state must be dense (ld == d): halo rows are received in place");
  GNPDE_CHECK_ARG(ri->graph->n <= h->n_own && rb->graph->n == h->n_own && rb->graph->row_begin == ri->graph->n, GNPDE_EINVAL,
                  "sharded solver: interior rows [0,%d) / boundary rows [%d,%d) do not tile the %d owned rows", ri->graph->n,
                  rb->graph->row_begin, rb->graph->n, h->n_own);
  return 0;
}

extern "C" size_t gnpde_sharded_solver_workspace_bytes(const gnpde_halo_t* halo, const gnpde_rhs_t* rhs_interior,
                                                       const gnpde_rhs_t* rhs_boundary, int32_t method) {
  if (check_sharded_args(halo, rhs_interior, rhs_boundary)) return 0;
  if (method != GNPDE_METHOD_EULER && method != GNPDE_METHOD_RK4) return 0;
  long long n_send = 0;
  for (int p = 0; p < halo->world; ++p) n_send += halo->send_counts[p];
  return sharded_layout(*rhs_interior, *rhs_boundary, method, halo->n_own + halo->n_halo, static_cast<int>(n_send), nullptr);
}

extern "C" int gnpde_sharded_solver_create(gnpde_sharded_solver_t** out, gnpde_comm_t* comm, const gnpde_halo_t* halo,
                                           const gnpde_rhs_t* rhs_interior, const gnpde_rhs_t* rhs_boundary, int32_t method,
                                           const float* dts, int32_t n_steps, void* workspace, size_t workspace_bytes) {
  GNPDE_CHECK_ARG(out != nullptr, GNPDE_EINVAL, "sharded_solver_create: out is null");
  *out = nullptr;
  int rc = check_sharded_args(halo, rhs_interior, rhs_boundary);
  if (rc) return rc;
  GNPDE_CHECK_ARG(method == GNPDE_METHOD_EULER || method == GNPDE_METHOD_RK4, GNPDE_EINVAL, "sharded_solver_create: bad method %d", method);
  GNPDE_CHECK_ARG(n_steps >= 0 && (dts || n_steps == 0), GNPDE_EINVAL, "sharded_solver_create: bad time grid");
  long long n_send = 0, n_recv = 0;
  for (int p = 0; p < halo->world; ++p) {
    GNPDE_CHECK_ARG(halo->send_counts[p] >= 0 && halo->recv_counts[p] >= 0, GNPDE_EINVAL, "sharded_solver_create: negative count");
    n_send += halo->send_counts[p];
    n_recv += halo->recv_counts[p];
  }
  GNPDE_CHECK_ARG(n_recv == halo->n_halo, GNPDE_EINVAL, "sharded_solver_create: recv counts sum to %lld, halo has %d rows", n_recv, halo->n_halo);
  GNPDE_CHECK_ARG(n_send == 0 || halo->send_idx != nullptr, GNPDE_EINVAL, "sharded_solver_create: send_idx is null");
  const bool exch = n_send > 0 || n_recv > 0;
  GNPDE_CHECK_ARG(!exch || (comm != nullptr && comm->world == halo->world && comm->rank == halo->rank), GNPDE_EINVAL,
                  "sharded_solver_create: communicator does not match the halo description");
  gnpde_sharded_solver* s = new gnpde_sharded_solver();
  s->comm = comm;
  s->rhs_int = *rhs_interior;
  s->rhs_bnd = *rhs_boundary;
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
  const size_t need = sharded_layout(s->rhs_int, s->rhs_bnd, method, s->n_own + s->n_halo, s->n_send, s);
  if (!(workspace && workspace_bytes >= need && reinterpret_cast<uintptr_t>(workspace) % 256 == 0)) {
    set_error("sharded_solver_create: workspace %zu bytes (need %zu, 256-byte aligned)", workspace_bytes, need);
    delete s;
    return GNPDE_EWS;
  }
  s->ws = static_cast<char*>(workspace);
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

extern "C" int gnpde_sharded_solver_run(gnpde_sharded_solver_t* s, float* y, int32_t use_graph, void* stream) {
  GNPDE_CHECK_ARG(s && y, GNPDE_EINVAL, "sharded_solver_run: null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool exch = s->n_send > 0 || s->n_halo > 0;
  if (!use_graph) {
    s->warmed = true;
    return enqueue_sharded_solve(s, y, st);
  }
  if (s->exec == nullptr || s->captured_y != y) {
    drop_sharded_graph(s);
    if (s->cap_stream == nullptr) GNPDE_HIP(hipStreamCreateWithFlags(&s->cap_stream, hipStreamNonBlocking));
    if (exch && !s->warmed) {
      // RCCL sets up its peer connections (allocations, IPC mappings) on first use: that cannot happen inside stream
      // capture.  One exchange of the scratch stage buffer outside capture, then wait for it.
      float* ua = reinterpret_cast<float*>(s->ws + s->off_ua);
      GNPDE_HIP(hipMemsetAsync(ua, 0, static_cast<size_t>(s->n_own + s->n_halo) * s->ld * 4, s->cap_stream));
      const int rc = enqueue_exchange(s, ua, s->cap_stream);
      if (rc) return rc;
      GNPDE_HIP(hipStreamWaitEvent(s->cap_stream, s->e_recv, 0));
      GNPDE_HIP(hipStreamSynchronize(s->cap_stream));
      s->warmed = true;
    }
    // relaxed mode: RCCL may call capture-unsafe runtime functions on this thread while it records its kernels
    GNPDE_HIP(hipStreamBeginCapture(s->cap_stream, hipStreamCaptureModeRelaxed));
    const int rc = enqueue_sharded_solve(s, y, s->cap_stream);
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

extern "C" int gnpde_sharded_solver_num_rhs_evals(const gnpde_sharded_solver_t* s) { return s ? s->n_evals : 0; }

extern "C" int gnpde_sharded_solver_destroy(gnpde_sharded_solver_t* s) {
  if (!s) return 0;
  drop_sharded_graph(s);
  if (s->cap_stream) (void)hipStreamDestroy(s->cap_stream);
  if (s->e_pack) (void)hipEventDestroy(s->e_pack);
  if (s->e_recv) (void)hipEventDestroy(s->e_recv);
  delete s;
  return 0;
}
