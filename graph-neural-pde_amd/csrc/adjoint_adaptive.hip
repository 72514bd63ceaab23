// torchdiffeq's adjoint with an ADAPTIVE adjoint method -- 'adaptive_heun', the reference's DEFAULT (src/run_GNN.py:334; best_params
// Pubmed), and 'dopri5' (best_params CoauthorCS, Computers: src/best_params.py) -- for the Laplacian function, with the step-size
// controller ON THE DEVICE (see gnpde.h, gnpde_adjoint_adaptive_*).
//
// torchdiffeq integrates the augmented state (vjp_t, y, a, g_theta) backwards as ONE flat vector through autograd (adjoint.py
// augmented_dynamics under misc.py _ReverseFunc; reference src/base_classes.py:44-47).  For f(u) = alpha' (A u - u) + beta x0 the system
// decouples, in reversed time s = -t:
//     y' = -f(y),   a' = alpha' (A^T a - a),   g_alpha' = (1 - alpha') <a, f(y) - beta x0>,   g_beta' = <a, x0>,   every other component 0.
// One TRIAL step of the embedded pair (adaptive_heun.py: 1 stage; dopri5.py: 6 stages, first-same-as-last; rk_common.py
// _runge_kutta_step / _adaptive_step, incl. its habit of carrying the LAST STAGE derivative over as the next step's first) is one hipGraph:
//     per stage r     fused row kernel on the graph: F_r = f(u_y) at the stage input, epilogue: the next stage input (or y1) of y,
//                     per-wave dots <u_a, F_r>, <u_a, x0>;  aggregation on the transposed graph: V_r = alpha' (A^T u_a - u_a), epilogue: the same for a
//     2 x error norm  block partial sums of ((sum_j e_j h K_j) / (atol + rtol max(|.|, |.1|)))^2 for y and for a
//     control         folds the partial sums and every stage's dots; the scalars' step g1 = g + h sum_j c_j Ks_j, error and mid point; mixed
//                     norm = the largest component rms (misc.py _mixed_norm: y, a and each scalar a component of its own); accept / reject; t, dt
//                     (float64, misc.py _optimal_step_size); end point reached -> interpolation fraction, the scalars interpolated
//     finish          commit of a rejected step (an accepted one costs nothing: two buffer parities alternate), the quartic end-point
//                     interpolation of a (interp.py), the next trial step's first stage inputs
// The host replays the graph and reads a 128-byte record once per batch of trial steps, exactly like gnpde_dopri5_run.
#include <cmath>
#include "common.h"
#include "rhs.h"

namespace gnpde {
namespace {

constexpr int kMaxStages = 6;

// embedded pairs of torchdiffeq 0.2.1 (adaptive_heun.py, dopri5.py): stage weights a[r] = weights of the input of stage r + 1 (row S - 1 of a
// first-same-as-last pair is the solution), solution / error / mid-point weights over K_0..K_S
struct Tableau {
  int S, fsal, order;
  double a[kMaxStages][kMaxStages];
  double c_sol[kMaxStages + 1], c_err[kMaxStages + 1], c_mid[kMaxStages + 1];
};

const Tableau kHeun = {1, 0, 2, {{1.0}}, {0.5, 0.5}, {0.5, -0.5}, {0.5, 0.0}};
const Tableau kDopri5 = {
  6, 1, 5,
  {{1.0 / 5},
   {3.0 / 40, 9.0 / 40},
   {44.0 / 45, -56.0 / 15, 32.0 / 9},
   {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729},
   {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656},
   {35.0 / 384, 0.0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84}},
  {35.0 / 384, 0.0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84, 0.0},
  {35.0 / 384 - 1951.0 / 21600, 0.0, 500.0 / 1113 - 22642.0 / 50085, 125.0 / 192 - 451.0 / 720, -2187.0 / 6784 + 12231.0 / 42400,
   11.0 / 84 - 649.0 / 6300, -1.0 / 60},
  {6025192743.0 / 30085553152.0 / 2, 0.0, 51252292925.0 / 65400821598.0 / 2, -2691868925.0 / 45128329728.0 / 2,
   187940372067.0 / 1594534317056.0 / 2, -1776094331.0 / 19743644256.0 / 2, 11237099.0 / 235043384.0 / 2}};

struct HCtl {             // controller record (device; copied to the host once per batch)
  double t, dt, t1;
  float h[2];             // fl32(dt) of the trial step in flight (slot = its parity) and of the next one
  float ratio, x;
  int accept, interp, done;
  int trials, accepted, rejected;
  float g[2][2];          // (g_alpha, g_beta) at the start of the trial step, per parity
  float ks[2][2];         // their derivatives carried over (the previous step's last stage), per parity
  float g_out[2];         // interpolated at t1
  int pad_[2];
};
static_assert(sizeof(HCtl) == 112, "controller record");

struct HCtlArgs {
  const double* part;                        // [kErrBlocks][2 + 2 S]: adaptive_err_kernel's block partial sums
  double count;
  const float* alpha; const float* beta; int alpha_sigmoid;
  float atol, rtol;
  int use_alpha, use_beta;
  int S, order;
  float c_sol[kMaxStages + 1], c_err[kMaxStages + 1], c_mid[kMaxStages + 1];
  const HCtl* c;                             // the record this trial step started from ...
  HCtl* c_next;                              // ... and where the next one's goes (the other of the two records)
  int parity;
};

__device__ __forceinline__ float quartic(float y0, float y1, float f0, float f1, float ym, float h, float x) {
  const float ca = 2.0f * h * (f1 - f0) - 8.0f * (y1 + y0) + 16.0f * ym;
  const float cb = h * (5.0f * f0 - 3.0f * f1) + 18.0f * y0 + 14.0f * y1 - 32.0f * ym;
  const float cc = h * (f1 - 4.0f * f0) - 11.0f * y0 - 5.0f * y1 + 16.0f * ym;
  const float cd = h * f0;
  float tot = y0 + x * cd;
  float xp = x * x;
  tot += xp * cc;
  xp *= x;
  tot += xp * cb;
  xp *= x;
  tot += xp * ca;
  return tot;
}

// sum over the block, every thread gets it
__device__ __forceinline__ double block_sum(double v, double* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const double out = red[0];
  __syncthreads();
  return out;
}

// Ks of one evaluation from the per-wave dots: ((1 - alpha') (<u_a, F> - beta <u_a, x0>), <u_a, x0>)
__device__ __forceinline__ void scalars_of(double d1, double d2, const float* alpha, int sig, const float* beta, float (&ks)[2]) {
  const float a = sig ? 1.0f / (1.0f + expf(-*alpha)) : *alpha;
  const float b = beta != nullptr ? *beta : 0.0f;
  const float f1 = static_cast<float>(d1), f2 = static_cast<float>(d2);
  ks[0] = (1.0f - a) * (f1 - b * f2);
  ks[1] = f2;
}

__global__ __launch_bounds__(kBlock) void adaptive_init_kernel(const float* __restrict__ dots, long long n_pairs, const float* alpha, int sig,
                                                              const float* beta, HCtl* c, double t0, double t1, double dt0, const float* g_in) {
  __shared__ double red[kBlock];
  double a1 = 0.0, a2 = 0.0;
  for (long long i = threadIdx.x; i < n_pairs; i += kBlock) { a1 += dots[2 * i]; a2 += dots[2 * i + 1]; }
  const double d1 = block_sum(a1, red), d2 = block_sum(a2, red);
  if (threadIdx.x != 0) return;
  float ks[2];
  scalars_of(d1, d2, alpha, sig, beta, ks);
  c->t = t0; c->t1 = t1; c->dt = dt0;
  c->h[0] = static_cast<float>(dt0); c->h[1] = 0.f;
  c->ratio = 0.f; c->x = 0.f;
  c->accept = c->interp = c->done = 0;
  c->trials = c->accepted = c->rejected = 0;
  for (int j = 0; j < 2; ++j) { c->g[0][j] = g_in[j]; c->g[1][j] = g_in[j]; c->ks[0][j] = ks[j]; c->ks[1][j] = ks[j]; c->g_out[j] = g_in[j]; }
}

// The controller's arithmetic for one trial step (torchdiffeq rk_common.py _adaptive_step / misc.py _optimal_step_size and _mixed_norm over
// the components y, a and each scalar of its own), from the folded sums `tot` = (sum err_y^2, sum err_a^2, then per new stage the dots
// <u_a, F>, <u_a, x0>).  L: the record this trial step started from; returns the record of the next one.  Registers and compile-time
// indices only (round 5: the first version read and wrote the record field by field through its pointer and cost 32 us).
template <int S>
__device__ __forceinline__ HCtl control_decide(const HCtl& L, const double* tot, const HCtlArgs& a, float alpha_raw, float beta) {
  const bool p1 = a.parity != 0;         // slot of the trial in flight; the next one takes the other
  HCtl N = L;
  const float h = p1 ? L.h[1] : L.h[0];
  const float g0[2] = {p1 ? L.g[1][0] : L.g[0][0], p1 ? L.g[1][1] : L.g[0][1]};
  const float k0[2] = {p1 ? L.ks[1][0] : L.ks[0][0], p1 ? L.ks[1][1] : L.ks[0][1]};
  float h_next, g_next[2], k_next[2];
  if (L.done) {                       // replayed past the end point: change nothing
    N.accept = 0;
    N.interp = 0;
    h_next = h;
    g_next[0] = g0[0]; g_next[1] = g0[1];
    k_next[0] = k0[0]; k_next[1] = k0[1];
  } else {
    const float al = a.alpha_sigmoid ? 1.0f / (1.0f + expf(-alpha_raw)) : alpha_raw;
    float ks[S + 1][2];                // Ks of every evaluation from the folded dots: ((1 - alpha') (<u_a, F> - beta <u_a, x0>), <u_a, x0>)
    ks[0][0] = k0[0]; ks[0][1] = k0[1];
#pragma unroll
    for (int r = 0; r < S; ++r) {
      const float f1 = static_cast<float>(tot[2 + 2 * r]), f2 = static_cast<float>(tot[3 + 2 * r]);
      ks[r + 1][0] = (1.0f - al) * (f1 - beta * f2);
      ks[r + 1][1] = f2;
    }
    float g1[2], gm[2], rs = 0.f;
    const int use[2] = {a.use_alpha, a.use_beta};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float acc = 0.f, err = 0.f, mid = 0.f;
#pragma unroll
      for (int m = 0; m <= S; ++m) {          // fl32(c) * fl32(dt) per weight, summed in stage order, as on the flat vector
        if (a.c_sol[m] != 0.f) acc += ks[m][j] * (a.c_sol[m] * h);
        if (a.c_err[m] != 0.f) err += ks[m][j] * (a.c_err[m] * h);
        if (a.c_mid[m] != 0.f) mid += ks[m][j] * (a.c_mid[m] * h);
      }
      g1[j] = g0[j] + acc;
      gm[j] = g0[j] + mid;
      const float tol = a.atol + a.rtol * fmaxf(fabsf(g0[j]), fabsf(g1[j]));
      if (use[j]) rs = fmaxf(rs, fabsf(err / tol));
    }
    const float ry = static_cast<float>(sqrt(tot[0] / a.count)), ra = static_cast<float>(sqrt(tot[1] / a.count));
    const float ratio32 = fmaxf(fmaxf(ry, ra), rs);
    N.ratio = ratio32;
    const double ratio = static_cast<double>(ratio32);
    const double dt = L.dt;
    N.trials = L.trials + 1;
    int accept = 0, interp = 0;
    if (ratio <= 1.0) {
      const double t_next = L.t + dt;
      if (t_next >= L.t1) {
        N.x = static_cast<float>((L.t1 - L.t) / (t_next - L.t));
        interp = 1;
        N.done = 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) N.g_out[j] = quartic(g0[j], g1[j], ks[0][j], ks[S][j], gm[j], h, N.x);
      }
      N.t = t_next;
      accept = 1;
      N.accepted = L.accepted + 1;
    } else {
      N.rejected = L.rejected + 1;
    }
    double factor;
    if (ratio == 0.0) {
      factor = 10.0;
    } else {
      const double lo = ratio < 1.0 ? 1.0 : 0.2;
      factor = fmin(10.0, fmax(0.9 / pow(ratio, 1.0 / a.order), lo));
    }
    N.dt = dt * factor;
    h_next = static_cast<float>(N.dt);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      g_next[j] = accept ? g1[j] : g0[j];
      k_next[j] = accept ? ks[S][j] : k0[j];
    }
    N.accept = accept;
    N.interp = interp;
  }
  if (p1) {
    N.h[0] = h_next; N.g[0][0] = g_next[0]; N.g[0][1] = g_next[1]; N.ks[0][0] = k_next[0]; N.ks[0][1] = k_next[1];
  } else {
    N.h[1] = h_next; N.g[1][0] = g_next[0]; N.g[1][1] = g_next[1]; N.ks[1][0] = k_next[0]; N.ks[1][1] = k_next[1];
  }
  return N;
}

// ---- round 6: the tail of a trial step in TWO launches instead of four (error norm of y, error norm of a, controller, finish), and no
// launch whose only work is one wave of scalar code.
//   adaptive_err_kernel     kErrBlocks blocks: block partial sums (double) of BOTH error norms and of every new stage's dot pairs
//   adaptive_finish2_kernel every block folds those kErrBlocks partial rows itself (same code, same order: the same doubles in every
//                           block) and its first thread takes the controller's decision; block 0 writes the record of the NEXT trial step
//                           -- into the OTHER of two records, so that no block can read a record that has already been advanced --
//                           and all of them go on to the finish pass with the decision in LDS.
// Element arithmetic and the controller's float sequence are unchanged (rk_error_partial_kernel, control_decide); the squares and the dots
// are summed in double, so the grouping of the partial sums does not reach the float32 ratio.
constexpr int kErrBlocks = 1024;             // (256 blocks read the 8 operands of a Pubmed-size state in 41 us, 1024 in ~13)

struct HErrArgs {
  const float* y0; const float* y1; const float* kf[kMaxStages + 1];
  const float* a0; const float* a1; const float* kv[kMaxStages + 1];
  float ce[kMaxStages + 1];
  float atol, rtol;
  long long n;
  int d, ld;
  const float* h;                 // fl32(dt) of this trial step (device)
  const float* dots; long long n_pairs;
  double* part;                   // [kErrBlocks][2 + 2 S]
};

// this thread's share of sum ((sum_j ce_j h k_j) / (atol + rtol max(|y0|, |y1|)))^2 for the state (-> ey) and the adjoint (-> ea): the element
// arithmetic of rk_error_partial_kernel (misc.hip), both components in one loop so that their loads overlap
template <int S>
__device__ __forceinline__ void err_both(const HErrArgs& a, float s, double& ey, double& ea) {
  const int q = a.ld / 4;
  const long long total = a.n * q;
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long r = i / q;
    const int c = static_cast<int>(i - r * q) * 4;
    if (c >= a.d) continue;
    float4 e1 = make_float4(0.f, 0.f, 0.f, 0.f), e2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j <= S; ++j) {
      if (a.ce[j] == 0.0f) continue;              // (dopri5: the second stage has no weight in the error estimate)
      const float4 k1 = reinterpret_cast<const float4*>(a.kf[j])[i];
      const float4 k2 = reinterpret_cast<const float4*>(a.kv[j])[i];
      const float cj = a.ce[j] * s;
      e1.x = fmaf(k1.x, cj, e1.x); e1.y = fmaf(k1.y, cj, e1.y); e1.z = fmaf(k1.z, cj, e1.z); e1.w = fmaf(k1.w, cj, e1.w);
      e2.x = fmaf(k2.x, cj, e2.x); e2.y = fmaf(k2.y, cj, e2.y); e2.z = fmaf(k2.z, cj, e2.z); e2.w = fmaf(k2.w, cj, e2.w);
    }
    const float4 u1 = reinterpret_cast<const float4*>(a.y0)[i], v1 = reinterpret_cast<const float4*>(a.y1)[i];
    const float4 u2 = reinterpret_cast<const float4*>(a.a0)[i], v2 = reinterpret_cast<const float4*>(a.a1)[i];
    const float ee1[4] = {e1.x, e1.y, e1.z, e1.w}, ee2[4] = {e2.x, e2.y, e2.z, e2.w};
    const float m1[4] = {fmaxf(fabsf(u1.x), fabsf(v1.x)), fmaxf(fabsf(u1.y), fabsf(v1.y)), fmaxf(fabsf(u1.z), fabsf(v1.z)), fmaxf(fabsf(u1.w), fabsf(v1.w))};
    const float m2[4] = {fmaxf(fabsf(u2.x), fabsf(v2.x)), fmaxf(fabsf(u2.y), fabsf(v2.y)), fmaxf(fabsf(u2.z), fabsf(v2.z)), fmaxf(fabsf(u2.w), fabsf(v2.w))};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (c + t < a.d) {
        const float q1 = ee1[t] / (a.atol + a.rtol * m1[t]);
        const float q2 = ee2[t] / (a.atol + a.rtol * m2[t]);
        ey += static_cast<double>(q1 * q1);
        ea += static_cast<double>(q2 * q2);
      }
    }
  }
}

// sum of one double per thread over the block, fixed order (lanes by xor butterfly, the four waves in order); thread 0 holds it
__device__ __forceinline__ double block_fold(double v, double* red4) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) red4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red4[0] + red4[1]) + (red4[2] + red4[3]);
}

template <int S>
__global__ __launch_bounds__(kBlock) void adaptive_err_kernel(const HErrArgs a) {
  __shared__ double red4[kWavesPerBlock];
  constexpr int kVals = 2 + 2 * S;
  const float s = *a.h;
  double* out = a.part + static_cast<size_t>(blockIdx.x) * kVals;
  double ey = 0.0, ea = 0.0;
  err_both<S>(a, s, ey, ea);
  ey = block_fold(ey, red4);
  ea = block_fold(ea, red4);
  if (threadIdx.x == 0) { out[0] = ey; out[1] = ea; }
  const float2* __restrict__ pairs = reinterpret_cast<const float2*>(a.dots);
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
#pragma unroll
  for (int r = 0; r < S; ++r) {
    double d1 = 0.0, d2 = 0.0;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < a.n_pairs; i += stride) {
      const float2 v = pairs[a.n_pairs * r + i];
      d1 += v.x;
      d2 += v.y;
    }
    d1 = block_fold(d1, red4);
    d2 = block_fold(d2, red4);
    if (threadIdx.x == 0) { out[2 + 2 * r] = d1; out[3 + 2 * r] = d2; }
  }
}

struct HFinishArgs {
  const float* y; float* y1; const float* f; float* f1;       // state: (y, y1), the carried derivative and the last stage's (F_0, F_S); y' = -F
  const float* a; float* a1; const float* v[kMaxStages + 1];  // adjoint: (a, a1), V_0..V_S (v[S] is written on a rejected step); a' = +V
  float* vS;
  float* uy; float* ua; float* a_out;
  float c_mid[kMaxStages + 1];
  float a00;
  int S;
  long long n4;
  const HCtl* c;
  int parity;
};

template <int S>
__global__ __launch_bounds__(kBlock) void adaptive_finish2_kernel(const HCtlArgs a, const HFinishArgs p) {
  constexpr int kVals = 2 + 2 * S;
  __shared__ double red4[kWavesPerBlock];
  __shared__ double tot[kVals];
  __shared__ float dec_f[3];       // h, h of the next trial step, interpolation fraction
  __shared__ int dec_i[2];         // accept, interp
  // every block: the kErrBlocks partial rows, kErrBlocks / kBlock per thread in row order, folded in the same order everywhere
  static_assert(kErrBlocks % kBlock == 0, "partial rows per thread");
  double mine[kVals];
#pragma unroll
  for (int j = 0; j < kVals; ++j) mine[j] = 0.0;
#pragma unroll
  for (int rr = 0; rr < kErrBlocks / kBlock; ++rr) {
    const double* row = a.part + (static_cast<size_t>(rr) * kBlock + threadIdx.x) * kVals;
#pragma unroll
    for (int j = 0; j < kVals; ++j) mine[j] += row[j];
  }
#pragma unroll
  for (int j = 0; j < kVals; ++j) {
    const double t = block_fold(mine[j], red4);
    if (threadIdx.x == 0) tot[j] = t;
  }
  if (threadIdx.x == 0) {
    const HCtl L = *a.c;
    const float alpha_raw = *a.alpha;
    const float beta = a.beta != nullptr ? *a.beta : 0.0f;
    const HCtl N = control_decide<S>(L, tot, a, alpha_raw, beta);
    dec_f[0] = a.parity ? L.h[1] : L.h[0];
    dec_f[1] = a.parity ? N.h[0] : N.h[1];
    dec_f[2] = N.x;
    dec_i[0] = N.accept;
    dec_i[1] = N.interp;
    if (blockIdx.x == 0) *a.c_next = N;
  }
  __syncthreads();
  const int accept = dec_i[0], interp = dec_i[1];
  const float h = dec_f[0], hn = dec_f[1], x = dec_f[2];
  const float cn = p.a00 * hn;
  typedef float f4 __attribute__((ext_vector_type(4)));
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < p.n4;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    // (the decision is the same for every thread of the grid: an accepted step that does not contain the end point -- the common case --
    //  needs the new state, adjoint and last derivatives only, a rejected one the old ones only: four of the eight streams)
    f4 ya = {0.f, 0.f, 0.f, 0.f}, yb = ya, fa = ya, fb = ya, aa = ya, ab = ya, va = ya, vb = ya;
    if (interp || !accept) {
      ya = reinterpret_cast<const f4*>(p.y)[i];
      fa = reinterpret_cast<const f4*>(p.f)[i];
      aa = reinterpret_cast<const f4*>(p.a)[i];
      va = reinterpret_cast<const f4*>(p.v[0])[i];
    }
    if (interp || accept) {
      yb = reinterpret_cast<const f4*>(p.y1)[i];
      fb = reinterpret_cast<const f4*>(p.f1)[i];
      ab = reinterpret_cast<const f4*>(p.a1)[i];
      vb = reinterpret_cast<const f4*>(p.v[S])[i];
    }
    if (interp) {
      f4 am = aa;
#pragma unroll
      for (int j = 0; j <= S; ++j) {
        const float m = p.c_mid[j] * h;
        if (m != 0.0f) am += reinterpret_cast<const f4*>(p.v[j])[i] * m;
      }
      f4 o;
#pragma unroll
      for (int t = 0; t < 4; ++t) o[t] = quartic(aa[t], ab[t], va[t], vb[t], am[t], h, x);
      reinterpret_cast<f4*>(p.a_out)[i] = o;
    }
    const f4 yn = accept ? yb : ya, fn = accept ? fb : fa, an = accept ? ab : aa, vn = accept ? vb : va;
    if (!accept) {
      reinterpret_cast<f4*>(p.y1)[i] = yn;
      reinterpret_cast<f4*>(p.f1)[i] = fn;
      reinterpret_cast<f4*>(p.a1)[i] = an;
      reinterpret_cast<f4*>(p.vS)[i] = vn;
    }
    f4 uy, ua;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uy[t] = fmaf(-cn, fn[t], yn[t]);
      ua[t] = fmaf(cn, vn[t], an[t]);
    }
    reinterpret_cast<f4*>(p.uy)[i] = uy;
    reinterpret_cast<f4*>(p.ua)[i] = ua;
  }
}

__global__ __launch_bounds__(kBlock) void adaptive_copy_rows_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst,
                                                                   long long n, int d) {
  const long long total = n * d;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const long long r = i / d;
    const int c = static_cast<int>(i - r * d);
    dst[static_cast<size_t>(r) * ld_dst + c] = src[static_cast<size_t>(r) * ld_src + c];
  }
}

__global__ void adaptive_store_g_kernel(const HCtl* c, float* g) {
  g[0] = c->g_out[0];
  g[1] = c->g_out[1];
}

}  // namespace
}  // namespace gnpde

using namespace gnpde;

struct gnpde_adjoint_adaptive {
  gnpde_rhs_t rhs;            // f on the graph
  gnpde_rhs_t rhs_b;          // alpha' (A^T u - u) on the transposed graph
  gnpde_graph_t graph, graph_t;
  RhsLayout Lb;
  const Tableau* tab;
  float rtol, atol;
  char* ws;
  size_t ws_bytes, state_bytes;
  size_t off_ctl, off_err_y, off_err_a, off_dots, off_part, off_rhs_b, off_r, off_state;
  int slots, n_state;
  size_t part_bytes;
  hipStream_t cap_stream = nullptr;
  hipGraph_t graph_obj[4] = {nullptr, nullptr, nullptr, nullptr};      // one trial step of parity 0 / 1; TWO trial steps starting with parity 0 / 1
  hipGraphExec_t exec[4] = {nullptr, nullptr, nullptr, nullptr};
  bool cleared = false;
  HCtl* host_ctl = nullptr;
  int n_evals = 0, n_accepted = 0, n_rejected = 0, n_launches = 0, n_syncs = 0;
  float* Y[2]; float* A[2]; float* KFp[2]; float* KVp[2]; float* KFm[kMaxStages]; float* KVm[kMaxStages]; float* UY[2]; float* UA[2]; float* AOUT;
  HCtl* ctl; float* dots; float* r_dummy;
};

namespace {

const Tableau* tableau_of(int method) {
  return method == GNPDE_ADAPTIVE_HEUN ? &kHeun : method == GNPDE_ADAPTIVE_DOPRI5 ? &kDopri5 : nullptr;
}

size_t adaptive_layout(const gnpde_rhs_t& r, const gnpde_rhs_t& rb, const Tableau& tab, gnpde_adjoint_adaptive* s) {
  const size_t state = align_up(static_cast<size_t>(r.graph->n) * r.ld * 4, 256);
  const int slots = adjoint_rows_dot_slots(r.graph, r.d);
  const size_t part = static_cast<size_t>(r.graph->n_long_chunks) * align_up(static_cast<size_t>(r.d), 4) * sizeof(float);
  const int n_state = 4 + 4 + 2 * (tab.S - 1) + 4 + 1;        // Y, A pairs; carried / last derivatives; middle stages; stage inputs; a_out
  size_t off = 0;
  const size_t off_ctl = off;   off += 512;                  // two controller records (a trial step reads one and writes the other)
  const size_t off_err_y = off; off += align_up(static_cast<size_t>(kErrBlocks) * (2 + 2 * kMaxStages) * sizeof(double), 256);
  const size_t off_err_a = off; off += 256;                  // (unused since round 6)
  const size_t off_dots = off;  off += align_up(static_cast<size_t>(slots) * 2 * sizeof(float) * tab.S, 256);
  const size_t off_part = off;  off += align_up(part, 256);
  const RhsLayout Lb = rhs_layout(rb);
  const size_t off_rhs_b = off; off += align_up(Lb.total, 256);
  const size_t off_r = off;     off += align_up(static_cast<size_t>(r.graph->e > 0 ? r.graph->e : 1) * sizeof(float), 256);
  const size_t off_state = off; off += static_cast<size_t>(n_state) * state;
  if (s) {
    s->state_bytes = state; s->slots = slots; s->part_bytes = part; s->Lb = Lb; s->n_state = n_state;
    s->off_ctl = off_ctl; s->off_err_y = off_err_y; s->off_err_a = off_err_a; s->off_dots = off_dots; s->off_part = off_part;
    s->off_rhs_b = off_rhs_b; s->off_r = off_r; s->off_state = off_state;
  }
  return off;
}

// One evaluation of the augmented right-hand side at the stage inputs (uy, ua): F -> f_out with the dots into region `dots_region`, V -> v_out.
// n_prev >= 0: the LINCOMB epilogues form  dst_y = y - h (sum_j w_j F_j + w_new F)  and  dst_a = a + h (sum_j w_j V_j + w_new V)
// from the earlier stage derivatives; n_prev < 0: plain evaluation.
int enqueue_eval(gnpde_adjoint_adaptive* s, const float* uy, const float* ua, float* f_out, float* v_out, int dots_region, int n_prev,
                 const float* const* pf, const float* const* pv, const float* w, const float* y, float* dst_y, const float* a, float* dst_a,
                 const float* h, hipStream_t st) {
  const gnpde_rhs_t& r = s->rhs;
  const bool padded = (r.flags & GNPDE_RHS_PADDED_ROWS) != 0;
  gnpde_epilogue_t e = base_epilogue(r);
  e.stage = GNPDE_STAGE_LINCOMB;
  e.out_k = f_out;
  gnpde_epilogue_t eb = base_epilogue(s->rhs_b);
  eb.stage = GNPDE_STAGE_LINCOMB;
  eb.out_k = v_out;
  if (n_prev >= 0) {
    // (earlier derivatives with a zero weight -- the second stage in Dormand-Prince's solution row -- are not streamed)
    int np = 0;
    for (int j = 0; j < n_prev; ++j) {
      if (w[j] == 0.0f) continue;
      e.prev[np] = pf[j]; e.coef[np] = -w[j]; eb.prev[np] = pv[j]; eb.coef[np] = w[j];
      ++np;
    }
    e.y = y; e.out_y = dst_y; e.n_prev = np; e.coef_scale = h;
    eb.y = a; eb.out_y = dst_a; eb.n_prev = np; eb.coef_scale = h;
    e.coef[np] = -w[n_prev];
    eb.coef[np] = w[n_prev];
  }
  float* dots = s->dots + 2 * static_cast<size_t>(s->slots) * dots_region;
  if (int rc = launch_adjoint_rows(&s->graph, r.w_csr, uy, ua, r.d, r.ld, &e, s->r_dummy, dots, s->part_bytes ? s->ws + s->off_part : nullptr,
                                   s->part_bytes, st, padded, false))
    return rc;
  return enqueue_rhs(s->rhs_b, ua, eb, s->ws + s->off_rhs_b, s->Lb, st);
}

int enqueue_trial(gnpde_adjoint_adaptive* s, int parity, hipStream_t st) {
  const gnpde_rhs_t& r = s->rhs;
  const Tableau& tab = *s->tab;
  const int S = tab.S;
  const long long n = r.graph->n;
  const int p = parity, q = 1 - parity;
  const float* h = &s->ctl[p].h[p];            // (record p is the one the trial step of parity p reads)
  float* kf[kMaxStages + 1];
  float* kv[kMaxStages + 1];
  kf[0] = s->KFp[p]; kv[0] = s->KVp[p];
  for (int j = 1; j < S; ++j) { kf[j] = s->KFm[j - 1]; kv[j] = s->KVm[j - 1]; }
  kf[S] = s->KFp[q]; kv[S] = s->KVp[q];
  const float* cur_y = s->UY[0];
  const float* cur_a = s->UA[0];
  for (int rr = 1; rr <= S; ++rr) {
    float w[kMaxStages + 1];
    int n_prev = -1;
    float* dst_y = nullptr;
    float* dst_a = nullptr;
    if (rr < S) {                       // the input of stage rr + 1 (of a first-same-as-last pair the last one IS the solution)
      n_prev = rr;
      for (int j = 0; j <= rr; ++j) w[j] = static_cast<float>(tab.a[rr][j]);
      const bool last = tab.fsal && rr == S - 1;
      dst_y = last ? s->Y[q] : s->UY[rr % 2];
      dst_a = last ? s->A[q] : s->UA[rr % 2];
    } else if (!tab.fsal) {             // the solution from every stage derivative
      n_prev = S;
      for (int j = 0; j <= S; ++j) w[j] = static_cast<float>(tab.c_sol[j]);
      dst_y = s->Y[q];
      dst_a = s->A[q];
    }
    if (int rc = enqueue_eval(s, cur_y, cur_a, kf[rr], kv[rr], rr - 1, n_prev, kf, kv, w, s->Y[p], dst_y, s->A[p], dst_a, h, st)) return rc;
    if (dst_y != nullptr) { cur_y = dst_y; cur_a = dst_a; }
  }
  const HCtl* rec = s->ctl + p;                 // this trial step's record; the next one's goes to the other
  HErrArgs ea{};
  ea.y0 = s->Y[p]; ea.y1 = s->Y[q]; ea.a0 = s->A[p]; ea.a1 = s->A[q];
  for (int j = 0; j <= S; ++j) { ea.kf[j] = kf[j]; ea.kv[j] = kv[j]; ea.ce[j] = static_cast<float>(tab.c_err[j]); }
  ea.atol = s->atol; ea.rtol = s->rtol; ea.n = n; ea.d = r.d; ea.ld = r.ld;
  ea.h = &rec->h[p];
  ea.dots = s->dots; ea.n_pairs = s->slots;
  ea.part = reinterpret_cast<double*>(s->ws + s->off_err_y);
  HCtlArgs ca{};
  ca.part = ea.part;
  ca.count = static_cast<double>(n) * r.d;
  ca.alpha = r.alpha; ca.beta = r.x0 != nullptr ? r.beta : nullptr; ca.alpha_sigmoid = r.alpha_sigmoid;
  ca.atol = s->atol; ca.rtol = s->rtol;
  ca.use_alpha = 1; ca.use_beta = r.x0 != nullptr ? 1 : 0;
  ca.S = S; ca.order = tab.order;
  for (int j = 0; j <= S; ++j) {
    ca.c_sol[j] = static_cast<float>(tab.c_sol[j]); ca.c_err[j] = static_cast<float>(tab.c_err[j]); ca.c_mid[j] = static_cast<float>(tab.c_mid[j]);
  }
  ca.c = rec; ca.c_next = s->ctl + q; ca.parity = p;
  if (S != 1 && S != 6) {
    set_error("adjoint_adaptive: no controller kernel for a pair of %d stages", S);
    return GNPDE_EINVAL;
  }
  if (S == 1) hipLaunchKernelGGL(adaptive_err_kernel<1>, dim3(kErrBlocks), dim3(kBlock), 0, st, ea);
  else hipLaunchKernelGGL(adaptive_err_kernel<6>, dim3(kErrBlocks), dim3(kBlock), 0, st, ea);
  GNPDE_LAUNCH_CHECK();
  HFinishArgs fa{};
  fa.y = s->Y[p]; fa.y1 = s->Y[q]; fa.f = kf[0]; fa.f1 = kf[S];
  fa.a = s->A[p]; fa.a1 = s->A[q];
  for (int j = 0; j <= S; ++j) { fa.v[j] = kv[j]; fa.c_mid[j] = static_cast<float>(tab.c_mid[j]); }
  fa.vS = kv[S];
  fa.uy = s->UY[0]; fa.ua = s->UA[0]; fa.a_out = s->AOUT;
  fa.a00 = static_cast<float>(tab.a[0][0]);
  fa.S = S;
  fa.n4 = n * r.ld / 4; fa.c = rec; fa.parity = p;
  long long blocks = (fa.n4 + kBlock - 1) / kBlock;
  if (blocks > 1024) blocks = 1024;         // (every block folds the partial rows and takes the decision itself: 32 KB of L2 reads per block)
  if (blocks < 1) blocks = 1;
  if (S == 1) hipLaunchKernelGGL(adaptive_finish2_kernel<1>, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, st, ca, fa);
  else hipLaunchKernelGGL(adaptive_finish2_kernel<6>, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, st, ca, fa);
  GNPDE_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" size_t gnpde_adjoint_adaptive_workspace_bytes(const gnpde_rhs_t* rhs, const gnpde_graph_t* graph_t, int32_t method) {
  const Tableau* tab = tableau_of(method);
  if (check_rhs(rhs) || graph_t == nullptr || rhs->kind != GNPDE_RHS_LAPLACIAN || tab == nullptr) return 0;
  gnpde_rhs_t rb = *rhs;
  rb.graph = graph_t;
  rb.beta = nullptr;
  rb.x0 = nullptr;
  return adaptive_layout(*rhs, rb, *tab, nullptr);
}

extern "C" int gnpde_adjoint_adaptive_create(gnpde_adjoint_adaptive_t** out, const gnpde_rhs_t* rhs, const gnpde_graph_t* graph_t, const float* w_t,
                                             int32_t method, float rtol, float atol, void* workspace, size_t workspace_bytes) {
  GNPDE_CHECK_ARG(out != nullptr, GNPDE_EINVAL, "adjoint_adaptive_create: out is null");
  *out = nullptr;
  if (int rc = check_rhs(rhs)) return rc;
  const Tableau* tab = tableau_of(method);
  GNPDE_CHECK_ARG(tab != nullptr, GNPDE_EINVAL, "adjoint_adaptive_create: method must be GNPDE_ADAPTIVE_HEUN or GNPDE_ADAPTIVE_DOPRI5");
  GNPDE_CHECK_ARG(rhs->kind == GNPDE_RHS_LAPLACIAN, GNPDE_EINVAL, "adjoint_adaptive_create: the Laplacian function only (f linear in the state)");
  GNPDE_CHECK_ARG(graph_t != nullptr && graph_t->n == rhs->graph->n && graph_t->e == rhs->graph->e && (w_t != nullptr || graph_t->e == 0),
                  GNPDE_EINVAL, "adjoint_adaptive_create: transposed graph / weights missing or of another size");
  GNPDE_CHECK_ARG(rtol >= 0.f && atol >= 0.f && rtol + atol > 0.f, GNPDE_EINVAL, "adjoint_adaptive_create: bad tolerances");
  GNPDE_CHECK_ARG(rhs->ld % 4 == 0 && rhs->d <= 256 && rhs->graph->row_begin == 0, GNPDE_ESHAPE,
                  "adjoint_adaptive_create: whole graphs, rows of up to 256 floats with a stride that is a multiple of 4");
  gnpde_adjoint_adaptive* s = new gnpde_adjoint_adaptive();
  s->rhs = *rhs;
  s->graph = *rhs->graph;
  s->rhs.graph = &s->graph;
  s->graph_t = *graph_t;
  s->rhs_b = s->rhs;
  s->rhs_b.graph = &s->graph_t;
  s->rhs_b.w_csr = w_t;
  s->rhs_b.beta = nullptr;
  s->rhs_b.x0 = nullptr;
  s->tab = tab;
  s->rtol = rtol;
  s->atol = atol;
  const size_t need = adaptive_layout(s->rhs, s->rhs_b, *tab, s);
  if (!(workspace && workspace_bytes >= need && reinterpret_cast<uintptr_t>(workspace) % 256 == 0)) {
    set_error("adjoint_adaptive_create: workspace %zu bytes (need %zu, 256-byte aligned)", workspace_bytes, need);
    delete s;
    return GNPDE_EWS;
  }
  s->ws = static_cast<char*>(workspace);
  s->ws_bytes = workspace_bytes;
  s->ctl = reinterpret_cast<HCtl*>(s->ws + s->off_ctl);
  s->dots = reinterpret_cast<float*>(s->ws + s->off_dots);
  s->r_dummy = reinterpret_cast<float*>(s->ws + s->off_r);
  float* base = reinterpret_cast<float*>(s->ws + s->off_state);
  const size_t stride = s->state_bytes / 4;
  int b = 0;
  auto next = [&]() { return base + static_cast<size_t>(b++) * stride; };
  s->Y[0] = next(); s->Y[1] = next(); s->A[0] = next(); s->A[1] = next();
  s->KFp[0] = next(); s->KFp[1] = next(); s->KVp[0] = next(); s->KVp[1] = next();
  for (int j = 0; j < tab->S - 1; ++j) { s->KFm[j] = next(); s->KVm[j] = next(); }
  s->UY[0] = next(); s->UY[1] = next(); s->UA[0] = next(); s->UA[1] = next();
  s->AOUT = next();
  if (b != s->n_state) {
    set_error("adjoint_adaptive_create: internal buffer count %d != %d", b, s->n_state);
    delete s;
    return GNPDE_EINVAL;
  }
  if (hipHostMalloc(reinterpret_cast<void**>(&s->host_ctl), sizeof(HCtl), hipHostMallocDefault) != hipSuccess) {
    set_error("adjoint_adaptive_create: pinned allocation failed");
    delete s;
    return GNPDE_EINVAL;
  }
  *out = s;
  return 0;
}

extern "C" int gnpde_adjoint_adaptive_run(gnpde_adjoint_adaptive_t* s, const float* y, int32_t ld_y, float* a, int32_t ld_a, float* g, double s0,
                                          double s1, double dt0, int32_t trials_per_sync, int32_t max_evals, int32_t* finished, void* stream) {
  GNPDE_CHECK_ARG(s && y && a && g, GNPDE_EINVAL, "adjoint_adaptive_run: null argument");
  const gnpde_rhs_t& r = s->rhs;
  const Tableau& tab = *s->tab;
  GNPDE_CHECK_ARG(ld_y >= r.d && ld_a >= r.d && s1 > s0 && dt0 > 0.0, GNPDE_EINVAL, "adjoint_adaptive_run: bad strides, time span or first step");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (trials_per_sync < 1) trials_per_sync = 1;
  const long long n = r.graph->n;
  s->n_evals = s->n_accepted = s->n_rejected = s->n_launches = s->n_syncs = 0;
  if (finished) *finished = 0;
  if (!s->cleared) {     // once: the padding columns [d, ld) are never written with anything but what they hold
    GNPDE_HIP(hipMemsetAsync(s->ws + s->off_state, 0, static_cast<size_t>(s->n_state) * s->state_bytes, st));
    s->cleared = true;
  }
  long long cb = (n * r.d + kBlock - 1) / kBlock;
  if (cb > 8192) cb = 8192;
  if (cb < 1) cb = 1;
  hipLaunchKernelGGL(adaptive_copy_rows_kernel, dim3(static_cast<unsigned>(cb)), dim3(kBlock), 0, st, y, ld_y, s->Y[0], r.ld, n, r.d);
  GNPDE_LAUNCH_CHECK();
  hipLaunchKernelGGL(adaptive_copy_rows_kernel, dim3(static_cast<unsigned>(cb)), dim3(kBlock), 0, st, a, ld_a, s->A[0], r.ld, n, r.d);
  GNPDE_LAUNCH_CHECK();
  // the derivative at the start (torchdiffeq: f0 of _before_integrate), the controller record, the first stage inputs
  if (int rc = enqueue_eval(s, s->Y[0], s->A[0], s->KFp[0], s->KVp[0], 0, -1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, st))
    return rc;
  hipLaunchKernelGGL(adaptive_init_kernel, dim3(1), dim3(kBlock), 0, st, s->dots, static_cast<long long>(s->slots), r.alpha, r.alpha_sigmoid,
                     r.x0 != nullptr ? r.beta : nullptr, s->ctl, s0, s1, dt0, g);
  GNPDE_LAUNCH_CHECK();
  const long long flat = n * r.ld;
  {
    const float a00 = static_cast<float>(tab.a[0][0]);
    const float* v[1] = {s->KFp[0]};
    const float c[1] = {-a00};
    if (int rc = launch_lincomb(s->Y[0], v, c, 1, flat, s->UY[0], st, &s->ctl->h[0])) return rc;
    const float* w[1] = {s->KVp[0]};
    const float c2[1] = {a00};
    if (int rc = launch_lincomb(s->A[0], w, c2, 1, flat, s->UA[0], st, &s->ctl->h[0])) return rc;
  }
  for (int which = 0; which < 4; ++which) {
    if (s->exec[which] != nullptr) continue;
    const int parity = which & 1;
    if (s->cap_stream == nullptr) GNPDE_HIP(hipStreamCreateWithFlags(&s->cap_stream, hipStreamNonBlocking));
    GNPDE_HIP(hipStreamBeginCapture(s->cap_stream, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_trial(s, parity, s->cap_stream);
    if (rc == 0 && which >= 2) rc = enqueue_trial(s, 1 - parity, s->cap_stream);      // two trial steps per launch: half the launch gaps
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
    s->graph_obj[which] = gobj;
    GNPDE_HIP(hipGraphInstantiate(&s->exec[which], gobj, nullptr, nullptr, 0));
  }
  bool have_record = false;
  for (;;) {
    // as gnpde_dopri5_run: queue as many trial steps as cannot pass s1 even if each were accepted with the largest growth (x 10)
    const HCtl& hc = *s->host_ctl;
    int batch = 1;
    double reach = hc.dt, step = hc.dt;
    while (have_record && batch < trials_per_sync && hc.t + reach < hc.t1) {
      step *= 10.0;
      reach += step;
      ++batch;
    }
    for (int b = 0; b < batch;) {
      const int par = (s->n_launches + b) & 1;
      if (batch - b >= 2) { GNPDE_HIP(hipGraphLaunch(s->exec[2 + par], st)); b += 2; }
      else { GNPDE_HIP(hipGraphLaunch(s->exec[par], st)); b += 1; }
    }
    s->n_launches += batch;
    // (trial step k, parity k & 1, leaves the next record in slot (k + 1) & 1)
    GNPDE_HIP(hipMemcpyAsync(s->host_ctl, s->ctl + (s->n_launches & 1), sizeof(HCtl), hipMemcpyDeviceToHost, st));
    GNPDE_HIP(hipStreamSynchronize(st));
    s->n_syncs += 1;
    have_record = true;
    s->n_evals = hc.trials * tab.S;
    s->n_accepted = hc.accepted;
    s->n_rejected = hc.rejected;
    if (hc.done) break;
    GNPDE_CHECK_ARG(hc.t + hc.dt > hc.t, GNPDE_EINVAL, "adjoint_adaptive_run: underflow in dt %g at s %g", hc.dt, hc.t);
    if (max_evals > 0 && s->n_evals > max_evals) return 0;   // *finished stays 0
  }
  hipLaunchKernelGGL(adaptive_copy_rows_kernel, dim3(static_cast<unsigned>(cb)), dim3(kBlock), 0, st, s->AOUT, r.ld, a, ld_a, n, r.d);
  GNPDE_LAUNCH_CHECK();
  hipLaunchKernelGGL(adaptive_store_g_kernel, dim3(1), dim3(1), 0, st, s->ctl + (s->n_launches & 1), g);
  GNPDE_LAUNCH_CHECK();
  if (finished) *finished = 1;
  return 0;
}

extern "C" int gnpde_adjoint_adaptive_stats(const gnpde_adjoint_adaptive_t* s, int32_t* n_evals, int32_t* n_accepted, int32_t* n_rejected,
                                            int32_t* n_launches, int32_t* n_syncs) {
  GNPDE_CHECK_ARG(s != nullptr, GNPDE_EINVAL, "adjoint_adaptive_stats: solver is null");
  if (n_evals) *n_evals = s->n_evals;
  if (n_accepted) *n_accepted = s->n_accepted;
  if (n_rejected) *n_rejected = s->n_rejected;
  if (n_launches) *n_launches = s->n_launches;
  if (n_syncs) *n_syncs = s->n_syncs;
  return 0;
}

extern "C" int gnpde_adjoint_adaptive_destroy(gnpde_adjoint_adaptive_t* s) {
  if (!s) return 0;
  for (int p = 0; p < 4; ++p) {
    if (s->exec[p]) (void)hipGraphExecDestroy(s->exec[p]);
    if (s->graph_obj[p]) (void)hipGraphDestroy(s->graph_obj[p]);
  }
  if (s->cap_stream) (void)hipStreamDestroy(s->cap_stream);
  if (s->host_ctl) (void)hipHostFree(s->host_ctl);
  delete s;
  return 0;
}
