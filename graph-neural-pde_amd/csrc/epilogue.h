// Device helpers shared by the aggregation kernels: vector loads/stores (plain and nontemporal) and
// the diffusion epilogue + fixed-step stage algebra (see gnpde.h, gnpde_epilogue_t).
#pragma once
#include "common.h"

namespace gnpde {

template <int VEC>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else if constexpr (VEC == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  } else {
    v[0] = *p;
  }
}

// streaming (touched once per launch) variants: nontemporal hint keeps the gathered rows in L2
template <int VEC>
__device__ __forceinline__ void load_vec_nt(const float* __restrict__ p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  } else if constexpr (VEC == 2) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 t = __builtin_nontemporal_load(reinterpret_cast<const f2*>(p));
    v[0] = t[0]; v[1] = t[1];
  } else {
    v[0] = __builtin_nontemporal_load(p);
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec_nt(float* __restrict__ p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 t = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<f4*>(p));
  } else if constexpr (VEC == 2) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 t = {v[0], v[1]};
    __builtin_nontemporal_store(t, reinterpret_cast<f2*>(p));
  } else {
    __builtin_nontemporal_store(v[0], p);
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* __restrict__ p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  } else {
    *p = v[0];
  }
}

__device__ __forceinline__ float alpha_of(const gnpde_epilogue_t& ep) {
  const float a = *ep.alpha;
  return ep.alpha_sigmoid ? 1.0f / (1.0f + expf(-a)) : a;
}

template <int VEC, bool NT>
__device__ __forceinline__ void stage_store(const gnpde_epilogue_t& ep, size_t off, const float (&k)[VEC], const float (&ui)[VEC]);

// k = alpha (ax - u_i) + beta x0_i, then the stage algebra in torchdiffeq's operation order.
template <int VEC, bool NT>
__device__ __forceinline__ void epilogue(const gnpde_epilogue_t& ep, float alpha, float beta, size_t off,
                                         const float (&ax)[VEC], const float (&ui)[VEC]) {
  // per-row streaming operands (y, k1..k3, x0 in; k, y out) are touched once per launch
  auto ld = [](const float* p, float (&v)[VEC]) { if constexpr (NT) load_vec_nt<VEC>(p, v); else load_vec<VEC>(p, v); };
  float k[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) k[v] = alpha * (ax[v] - ui[v]);
  if (ep.x0 != nullptr) {
    float s[VEC];
    ld(ep.x0 + off, s);
#pragma unroll
    for (int v = 0; v < VEC; ++v) k[v] = k[v] + beta * s[v];
  }
  stage_store<VEC, NT>(ep, off, k, ui);
}

// The stage algebra alone: k = the derivative of this row (however it was formed), u_i = the row's own stage input.
template <int VEC, bool NT>
__device__ __forceinline__ void stage_store(const gnpde_epilogue_t& ep, size_t off, const float (&k)[VEC], const float (&ui)[VEC]) {
  auto ld = [](const float* p, float (&v)[VEC]) { if constexpr (NT) load_vec_nt<VEC>(p, v); else load_vec<VEC>(p, v); };
  auto st = [](float* p, const float (&v)[VEC]) { if constexpr (NT) store_vec_nt<VEC>(p, v); else store_vec<VEC>(p, v); };
  constexpr float kThird = 1.0f / 3.0f;
  const float dt = ep.dt;
  float y[VEC], a[VEC], b[VEC], c[VEC], o[VEC];
  switch (ep.stage) {
    case GNPDE_STAGE_RHS:
      st(ep.out_k + off, k);
      break;
    case GNPDE_STAGE_EULER:
      ld(ep.y + off, y);
#pragma unroll
      for (int v = 0; v < VEC; ++v) o[v] = y[v] + dt * k[v];
      st(ep.out_y + off, o);
      break;
    case GNPDE_STAGE_RK1:
      ld(ep.y + off, y);
      st(ep.out_k + off, k);
#pragma unroll
      for (int v = 0; v < VEC; ++v) o[v] = y[v] + (dt * k[v]) * kThird;
      st(ep.out_y + off, o);
      break;
    case GNPDE_STAGE_RK2:
      ld(ep.y + off, y);
      ld(ep.k1 + off, a);
      st(ep.out_k + off, k);
#pragma unroll
      for (int v = 0; v < VEC; ++v) o[v] = y[v] + dt * (k[v] - a[v] * kThird);
      st(ep.out_y + off, o);
      break;
    case GNPDE_STAGE_RK3:
      ld(ep.y + off, y);
      ld(ep.k1 + off, a);
      ld(ep.k2 + off, b);
      st(ep.out_k + off, k);
#pragma unroll
      for (int v = 0; v < VEC; ++v) o[v] = y[v] + dt * ((a[v] - b[v]) + k[v]);
      st(ep.out_y + off, o);
      break;
    case GNPDE_STAGE_RK4:
      ld(ep.y + off, y);
      ld(ep.k1 + off, a);
      ld(ep.k2 + off, b);
      ld(ep.k3 + off, c);
#pragma unroll
      for (int v = 0; v < VEC; ++v) o[v] = y[v] + (((a[v] + 3.0f * (b[v] + c[v])) + k[v]) * dt) * 0.125f;
      st(ep.out_y + off, o);
      break;
    case GNPDE_STAGE_RK1C:  // u == y
#pragma unroll
      for (int v = 0; v < VEC; ++v) o[v] = ui[v] + (dt * k[v]) * kThird;
      st(ep.out_y + off, o);
      break;
    case GNPDE_STAGE_RK2C:
      ld(ep.y + off, y);
#pragma unroll
      for (int v = 0; v < VEC; ++v) o[v] = (2.0f * y[v] - ui[v]) + dt * k[v];
      st(ep.out_y + off, o);
      break;
    case GNPDE_STAGE_RK3C:
      ld(ep.k1 + off, a);  // u2
#pragma unroll
      for (int v = 0; v < VEC; ++v) o[v] = (2.0f * a[v] - ui[v]) + dt * k[v];
      st(ep.out_y + off, o);
      break;
    case GNPDE_STAGE_RK4C:
      ld(ep.y + off, y);
      ld(ep.k1 + off, a);  // u3
#pragma unroll
      for (int v = 0; v < VEC; ++v) o[v] = (((6.0f * a[v] + 3.0f * ui[v]) - y[v]) + dt * k[v]) * 0.125f;
      st(ep.out_y + off, o);
      break;
    case GNPDE_STAGE_LINCOMB: {
      if (ep.out_k != nullptr) st(ep.out_k + off, k);
      if (ep.out_y != nullptr) {
        ld(ep.y + off, y);
#pragma unroll
        for (int v = 0; v < VEC; ++v) o[v] = 0.0f;
        const float cs = ep.coef_scale != nullptr ? *ep.coef_scale : 1.0f;   // fl32(dt) kept on the device (gnpde_dopri5_*)
        for (int j = 0; j < ep.n_prev; ++j) {
          ld(ep.prev[j] + off, a);
          const float cj = ep.coef[j] * cs;
#pragma unroll
          for (int v = 0; v < VEC; ++v) o[v] = fmaf(a[v], cj, o[v]);
        }
        const float ck = ep.coef[ep.n_prev] * cs;
#pragma unroll
        for (int v = 0; v < VEC; ++v) o[v] = y[v] + fmaf(k[v], ck, o[v]);
        st(ep.out_y + off, o);
      }
      break;
    }
    default:
      break;
  }
}


}  // namespace gnpde
