"""Integrator entry point with torchdiffeq's calling convention
(`odeint(func, y0, t, method=, options=, atol=, rtol=)` -> tensor [len(t), *y0.shape]), as used by
the reference's blocks (src/block_constant.py:57-62, src/block_transformer_attention.py:58-63).

Fixed-grid methods (`euler`, `rk4` == torchdiffeq 0.2.1's 3/8-rule rk4_alt_step_func) on one of this
package's ODEFunc objects run as ONE native solver object whose whole time loop is a captured
hipGraph (csrc/solver.hip).  Everything else (dopri5, foreign callables, several output times) runs
the host loops below, which still evaluate f through the native kernels."""
import math
import os

import torch

from . import _lib

_THIRD = 1 / 3
_TRIAL_TRACE = None       # tests / diagnostics: a list that receives (t, dt, error ratio) of every trial step of the host-controlled adaptive solves
_EVALS_PER_STEP = {'euler': 1, 'midpoint': 2, 'rk4': 4}


def time_grid(t, step_size):
  """torchdiffeq FixedGridODESolver grid: niters = ceil((t1 - t0)/h + 1) points t0 + i h, the last
  one replaced by t1 (so the final step may be short).  Computed in t's dtype like the original."""
  t0, t1 = t[0], t[-1]
  niters = int(torch.ceil((t1 - t0) / step_size + 1).item())
  grid = torch.arange(0, niters, dtype=t.dtype, device=t.device) * step_size + t0
  grid[-1] = t1
  return grid


_GRID_CACHE = {}


def step_sizes(t, step_size):
  """The step sizes of torchdiffeq's fixed grid over `t` as host floats.  The blocks pass the SAME device tensor `block.t` on every forward
  pass: reading it back costs a device -> host synchronisation per solve (~25 us of a 150-us Cora forward), so the result is kept per
  (tensor object, version, storage address, dtype, step size); tensors without a version counter are read every time."""
  try:
    key = (id(t), t._version, t.data_ptr(), t.dtype, float(step_size))
  except RuntimeError:
    key = None
  hit = _GRID_CACHE.get(key) if key is not None else None
  if hit is not None and hit[0] is t:
    return hit[1]
  grid = time_grid(t.detach().to('cpu'), step_size)
  dts = tuple((grid[1:] - grid[:-1]).tolist())
  if key is not None:
    if len(_GRID_CACHE) >= 64:
      _GRID_CACHE.clear()
    _GRID_CACHE[key] = (t, dts)        # (holds `t`: its id cannot be handed to another tensor while it is a key)
  return dts


def end_points(t):
  """(float(t[0]), float(t[-1])) with the same cache as step_sizes: the adaptive device solves read them once per `block.t`."""
  try:
    key = (id(t), t._version, t.data_ptr(), t.dtype, 'ends')
  except RuntimeError:
    key = None
  hit = _GRID_CACHE.get(key) if key is not None else None
  if hit is not None and hit[0] is t:
    return hit[1]
  ends = (float(t[0]), float(t[-1]))
  if key is not None:
    if len(_GRID_CACHE) >= 64:
      _GRID_CACHE.clear()
    _GRID_CACHE[key] = (t, ends)
  return ends


# --------------------------------------------------------------------------------------------------
# native fixed-step path
# --------------------------------------------------------------------------------------------------
def _native_ok(func, y0, t):
  return (hasattr(func, '_descriptor') and y0.is_cuda and y0.dim() == 2 and y0.dtype == torch.float32
          and len(t) == 2 and not func._needs_grad(y0))


def _solve_native(func, y0, t, method, step_size, use_graph=True, evaluator=None):
  from . import ops
  dts = list(step_sizes(t, step_size))
  n_evals = len(dts) * _EVALS_PER_STEP[method]
  # NFE guard with the reference's semantics (raise at the first evaluation that finds nfe > max_nfe)
  room = func.opt['max_nfe'] + 1 - func.nfe
  if n_evals > room:
    func.nfe += max(room, 0)
    from .utils import MaxNFEException
    raise MaxNFEException
  st = func.__dict__.setdefault('_solver_state', {})
  # relabelled graph (graph.LocalityView): the state enters as y0[order] and leaves as y[inv]; the in-graph early-stopping
  # evaluator gets its labels and split masks in the same order (EarlyStopEvaluator.relabelled: shared counters and trace)
  view = func._locality_view(y0) if hasattr(func, '_locality_view') else None
  if evaluator is not None and view is not None:
    evaluator = evaluator.relabelled(view)
  key = (method, tuple(dts), tuple(y0.shape), str(y0.device), id(view))
  ent = st.get(key)
  y0c = y0.detach()
  if ent is None:
    # (rows padded to a multiple of 4 floats when the width is not one: 16-byte lanes for d = 162 etc.)
    ent = {'y': _lib.alloc_state(y0c.shape[0], y0c.shape[1], y0c.device),
           'x0': _lib.alloc_state(y0c.shape[0], y0c.shape[1], y0c.device) if func.opt['add_source'] else None,
           'solver': None, 'sig': None, 'view': view}
    for old in st.values():     # one live solver per function object: its buffers are state-sized
      if old.get('solver') is not None:
        old['solver'].close()
    st.clear()
    dd = func.__dict__.get('_dopri5_device')      # ... shared with the adaptive path: an eval that alternates methods
    if dd:                                        # must not hold a fixed-step AND a 12-buffer dopri5 workspace
      for old in dd.values():
        if old.get('solver') is not None:
          old['solver'].close()
      dd.clear()
    st[key] = ent
  if view is None:
    ent['y'].copy_(y0c)
  else:
    view.enter(y0c, out=ent['y'])
  if ent['x0'] is not None:
    if func.x0 is None:
      raise _lib.GnpdeError('add_source is set but x0 was never assigned (call ODEblock.set_x0)')
    # the solver's copy of the source term is refreshed only when the caller's tensor is another one or was written to (keyed on
    # the tensor OBJECT, kept alive here so that its address cannot be handed to another tensor, and its version counter)
    src = func.x0
    # (inference tensors have no version counter -- reading it raises -- and writes through .data / set_() do not bump it: the
    #  key also holds the storage address, and anything that cannot be keyed is copied every solve)
    try:
      stamp = (src._version, src.data_ptr())
    except RuntimeError:
      stamp = None
    if stamp is None or ent.get('x0_src') is not src or ent.get('x0_version') != stamp:
      if view is None:
        ent['x0'].copy_(src)
      else:
        view.enter(src.detach(), out=ent['x0'])
      ent['x0_src'], ent['x0_version'] = src, stamp
  desc = func._descriptor(ent['y'], x0_override=ent['x0'], graph=None if view is None else view.graph)
  sig = func._descriptor_signature(desc)
  if ent['solver'] is None or ent['sig'] != sig:
    if ent['solver'] is not None:
      ent['solver'].close()
    ent['solver'] = ops.FixedStepSolver(desc, method, dts, y0.device)
    ent['sig'] = sig
  if getattr(ent['solver'], 'evaluator', None) is not evaluator:
    ent['solver'].set_early_stop(evaluator)     # per-step early-stopping evaluation inside the same hipGraph
  ent['solver'].run(ent['y'], use_graph=use_graph)
  func.nfe += n_evals
  out = torch.empty((2,) + tuple(y0.shape), dtype=y0.dtype, device=y0.device)
  out[0].copy_(y0c)
  if view is None:
    out[1].copy_(ent['y'])
  else:
    view.leave(ent['y'], out=out[1])
  return out


# --------------------------------------------------------------------------------------------------
# host loops (any callable)
# --------------------------------------------------------------------------------------------------
def _rk4_38_step(func, t0, dt, t1, y0):
  k1 = func(t0, y0)
  k2 = func(t0 + dt * _THIRD, y0 + dt * k1 * _THIRD)
  k3 = func(t0 + dt * (2 * _THIRD), y0 + dt * (k2 - k1 * _THIRD))
  k4 = func(t1, y0 + dt * (k1 - k2 + k3))
  return (k1 + 3 * (k2 + k3) + k4) * dt * 0.125


def _solve_fixed_host(func, y0, t, method, step_size, on_step=None):
  grid = time_grid(t, step_size)
  out = torch.empty((len(t),) + tuple(y0.shape), dtype=y0.dtype, device=y0.device)
  out[0] = y0
  j = 1
  y = y0
  for i in range(len(grid) - 1):
    ta, tb = grid[i], grid[i + 1]
    dt = tb - ta
    if method == 'euler':
      dy = dt * func(ta, y)
    elif method == 'midpoint':      # torchdiffeq fixed_grid.py Midpoint._step_func
      half = 0.5 * dt
      dy = dt * func(ta + half, y + func(ta, y) * half)
    else:
      dy = _rk4_38_step(func, ta, dt, tb, y)
    y_next = y + dy
    while j < len(t) and tb >= t[j]:
      if t[j] == tb:
        out[j] = y_next
      elif t[j] == ta:
        out[j] = y
      else:
        out[j] = y + ((t[j] - ta) / (tb - ta)) * (y_next - y)
      j += 1
    y = y_next
    if on_step is not None:
      on_step(y, i + 1)
  return out


# Dormand-Prince 5(4) coefficients (Shampine's error weights), as used by torchdiffeq's dopri5
_DP_A = (1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0)
_DP_B = ((1 / 5,),
         (3 / 40, 9 / 40),
         (44 / 45, -56 / 15, 32 / 9),
         (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
         (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656),
         (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84))
_DP_E = (35 / 384 - 1951 / 21600, 0.0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
         -2187 / 6784 + 12231 / 42400, 11 / 84 - 649 / 6300, -1 / 60)
_DP_MID = (6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
           187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2)


_DP_SOL = (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0.0)

# embedded pairs of torchdiffeq 0.2.1 (dopri5.py, adaptive_heun.py): stage nodes, stage weights, solution and error
# weights, mid-point weights of the quartic interpolant, order; `fsal`: the last stage input is the solution
_TABLEAUS = {
  'dopri5': dict(alpha=_DP_A, beta=_DP_B, c_sol=_DP_SOL, c_err=_DP_E, c_mid=_DP_MID, order=5, fsal=True),
  'adaptive_heun': dict(alpha=(1.0,), beta=((1.0,),), c_sol=(0.5, 0.5), c_err=(0.5, -0.5), c_mid=(0.5, 0.0), order=2,
                        fsal=False),
}


def _rms(v):
  return v.pow(2).mean().sqrt()


def _mixed_norm(shapes):
  """torchdiffeq's default norm of a tuple state (misc.py _mixed_linf_rms_norm): the largest component rms."""
  sizes = [int(torch.Size(sh).numel()) for sh in shapes]

  def norm(v):
    return max(_rms(part) for part in torch.split(v, sizes))
  return norm


def _combine(y0, ks, coeffs, dt):
  """y0 + sum_j (c_j * dt) k_j with the coefficients rounded to the state dtype first, as
  torchdiffeq's k.matmul(beta * dt) does."""
  acc = None
  for kj, c in zip(ks, coeffs):
    if c == 0.0:
      continue
    term = kj * (torch.tensor(c, dtype=kj.dtype, device=kj.device) * dt)
    acc = term if acc is None else acc + term
  return acc if y0 is None else y0 + acc


def _solve_dopri5(func, y0, t, rtol, atol, max_num_steps=2 ** 31 - 1, safety=0.9, ifactor=10.0, dfactor=0.2,
                  on_accept=None, stop_after=None, tableau='dopri5', norm=None, on_reject=None):
  """Adaptive embedded Runge-Kutta (Dormand-Prince 5(4) by default, `adaptive_heun` 2(1)) with torchdiffeq 0.2.1's
  controller: time and step size in float64, state in y0's dtype, rms error norm (or `norm`), the last stage
  derivative reused as the next step's first (rk_common.py: f1 = k[..., -1], also for the non-FSAL Heun pair),
  quartic-interpolated output."""
  tab = _TABLEAUS[tableau]
  order = tab['order']
  nrm = norm if norm is not None else _rms
  dev = y0.device
  f64 = dict(dtype=torch.float64, device=dev)
  rtol_t, atol_t = torch.as_tensor(rtol, **f64), torch.as_tensor(atol, **f64)
  tt = t.to(torch.float64)
  out = torch.empty((len(t),) + tuple(y0.shape), dtype=y0.dtype, device=dev)
  out[0] = y0
  f0 = func(tt[0], y0)
  # initial step (Hairer, Norsett & Wanner), order p = 4
  scale = atol_t + torch.abs(y0) * rtol_t
  d0, d1 = nrm(y0 / scale), nrm(f0 / scale)
  h0 = torch.tensor(1e-6, dtype=y0.dtype, device=dev) if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
  f1 = func(tt[0].to(y0.dtype) + h0, y0 + h0 * f0)
  d2 = nrm((f1 - f0) / scale) / h0
  if d1 <= 1e-15 and d2 <= 1e-15:
    h1 = torch.max(torch.tensor(1e-6, dtype=y0.dtype, device=dev), h0 * 1e-3)
  else:
    h1 = (0.01 / max(d1, d2)) ** (1.0 / order)
  dt = torch.min(100 * h0, h1).to(torch.float64)
  y, f, t_prev, t_cur = y0, f0, tt[0], tt[0]
  interp = None
  for i in range(1, len(t)):
    n_steps = 0
    while tt[i] > t_cur and (stop_after is None or n_steps < stop_after):
      assert n_steps < max_num_steps, 'max_num_steps exceeded'
      assert t_cur + dt > t_cur, 'underflow in dt {}'.format(float(dt))
      dty = dt.to(y.dtype)
      ks = [f]
      yi = None
      for a_i, b_i in zip(tab['alpha'], tab['beta']):
        yi = _combine(y, ks, b_i, dty)
        ks.append(func(t_cur + dt if a_i == 1.0 else t_cur + a_i * dt, yi))
      y1 = yi if tab['fsal'] else _combine(y, ks, tab['c_sol'], dty)
      f1 = ks[-1]
      err = _combine(None, ks, tab['c_err'], dty)
      tol = atol_t + rtol_t * torch.max(y.abs(), y1.abs())
      ratio = nrm(err / tol)
      accept = bool(ratio <= 1)
      if _TRIAL_TRACE is not None:
        _TRIAL_TRACE.append((float(t_cur), float(dt), float(ratio)))
      if accept:
        y_mid = _combine(y, ks, tab['c_mid'], dty)
        interp = (y, y1, y_mid, ks[0], ks[-1], dty, t_cur, t_cur + dt)
        t_prev, t_cur = t_cur, t_cur + dt
        y, f = y1, f1
        if on_accept is not None:
          on_accept(y, float(t_cur))
      elif on_reject is not None:
        on_reject(y, float(t_cur))
      # step-size controller: torchdiffeq 0.2.1 computes the next step size under torch.no_grad (misc.py _optimal_step_size), so a
      # differentiated solve (opt['adjoint'] = False) treats every step size after the first as a constant of the backward pass
      with torch.no_grad():
        if ratio == 0:
          dt = dt * ifactor
        else:
          lo = 1.0 if ratio < 1 else dfactor
          r = ratio.to(torch.float64)
          factor = torch.clamp(safety / r ** (1.0 / order), min=lo, max=ifactor)
          dt = dt * factor
      n_steps += 1
    if stop_after is not None and n_steps >= stop_after:   # EarlyStopDopri5.advance: the state where it stopped
      out[i] = y
      continue
    ya, yb, ym, fa, fb, h, ta, tb = interp
    xfrac = ((tt[i] - ta) / (tb - ta)).to(y0.dtype)
    ca = 2 * h * (fb - fa) - 8 * (yb + ya) + 16 * ym
    cb = h * (5 * fa - 3 * fb) + 18 * ya + 14 * yb - 32 * ym
    cc = h * (fb - 4 * fa) - 11 * ya - 5 * yb + 16 * ym
    cd = h * fa
    total = ya + xfrac * cd
    xp = xfrac
    for coef in (cc, cb, ca):
      xp = xp * xfrac
      total = total + xp * coef
    out[i] = total
  return out


def _solve_dopri5_native(func, y0, t, rtol, atol, safety=0.9, ifactor=10.0, dfactor=0.2, on_accept=None,
                         stop_after=None, on_reject=None):
  """dopri5 with torchdiffeq 0.2.1's controller on the host (one scalar read per trial step) and everything
  state-sized on the device: each stage is ONE right-hand-side launch whose epilogue also forms the next stage
  input  y + sum_j (beta_ij dt) k_j  (GNPDE_STAGE_LINCOMB), the error ratio is a device reduction.  Same
  accept / reject rule, step-size update (float64), FSAL and quartic end-point interpolation as the host loop
  `_solve_dopri5` (which stays the path for foreign callables)."""
  import numpy as np
  from . import ops
  dev = y0.device
  f32 = np.float32
  T0, T1 = float(t[0]), float(t[-1])
  # state-sized buffers (rows padded when d % 4 != 0) are kept on the function object between solves
  cache = func.__dict__.setdefault('_dopri5_buffers', {})
  bkey = (tuple(y0.shape), str(dev))
  if cache.get('key') != bkey:
    cache.clear()
    cache['key'] = bkey
    cache['bufs'] = [_lib.alloc_state(y0.shape[0], y0.shape[1], dev) for _ in range(12)]
  bufs = cache['bufs']
  y, y1, u, K, y_out = bufs[0], bufs[1], bufs[2:4], bufs[4:11], bufs[11]
  y.copy_(y0.detach())
  ratio_dev = torch.zeros(1, dtype=torch.float32, device=dev)
  err_ws = torch.empty(4096, dtype=torch.float32, device=dev)
  desc = func._descriptor(y)
  L = _lib.lib()
  ws = desc.graph.workspace('rhs%d_%d' % (desc.struct.kind, desc.struct.d), L.gnpde_rhs_workspace_bytes(desc.ref()))

  def feval(src, out_k=None, out_y=None, ybase=None, prev=(), coef=()):
    func._check_nfe()
    ops.rhs_stage(desc, src, _lib.STAGE_LINCOMB, ws=ws, y=ybase, out_k=out_k, out_y=out_y, prev=prev, coef=coef)

  rtol64, atol64 = float(rtol), float(atol)
  feval(y, out_k=K[0])
  # initial step (order 4 estimate), a handful of reductions once per solve
  def scaled_rms(terms, coefs):
    """rms(sum_j c_j v_j / (atol + rtol |y|)) in one fused pass (the error-ratio kernel with y0 = y1 = y)."""
    ops.rk_error_ratio(y, y, terms, coefs, atol64, rtol64, ratio_dev, err_ws)
    return float(ratio_dev.item())

  d0, d1 = scaled_rms([y], [1.0]), scaled_rms([K[0]], [1.0])
  h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else float(f32(0.01) * f32(d0) / f32(d1))
  torch.add(y, K[0], alpha=h0, out=u[0])
  feval(u[0], out_k=K[1])
  d2 = scaled_rms([K[1], K[0]], [1.0, -1.0]) / h0
  h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else float(f32(f32(0.01) / f32(max(d1, d2))) ** f32(1.0 / 5.0))
  dt = float(min(100 * h0, h1))
  t_cur = T0
  out = torch.empty((2,) + tuple(y0.shape), dtype=y0.dtype, device=dev)
  out[0].copy_(y0)
  n_steps = 0
  while T1 > t_cur and (stop_after is None or n_steps < stop_after):
    assert t_cur + dt > t_cur, 'underflow in dt {}'.format(dt)
    dty = f32(dt)
    w = [[f32(b) * dty for b in row] for row in _DP_B]
    torch.add(y, K[0], alpha=float(w[0][0]), out=u[0])              # first stage input (k0 is FSAL)
    for i in range(1, 6):                                            # k_i = f(u_i);  u_{i+1} in the same launch
      dst = y1 if i == 5 else u[i % 2]
      feval(u[(i - 1) % 2], out_k=K[i], out_y=dst, ybase=y, prev=K[:i], coef=w[i])
    feval(y1, out_k=K[6])                                            # f(y1): next step's k0 if accepted
    ops.rk_error_ratio(y, y1, K, [f32(e) * dty for e in _DP_E], atol64, rtol64, ratio_dev, err_ws)
    ratio = float(ratio_dev.item())                                  # the one host sync of the step
    if ratio <= 1:
      t_next = t_cur + dt
      if t_next >= T1:   # end point inside this step: quartic interpolation (torchdiffeq _interp_fit / _interp_evaluate)
        xf = float(f32((T1 - t_cur) / (t_next - t_cur)))
        ops.dopri5_interp(y, y1, K, [f32(c) * dty for c in _DP_MID], float(dty), xf, y_out)
        out[1].copy_(y_out)
      y, y1 = y1, y
      K[0], K[6] = K[6], K[0]
      t_cur = t_next
      if on_accept is not None:
        on_accept(y, t_cur)
    elif on_reject is not None:
      on_reject(y, t_cur)
    if ratio == 0:
      dt = dt * ifactor
    else:
      lo = 1.0 if ratio < 1 else dfactor
      dt = dt * min(ifactor, max(safety / ratio ** (1.0 / 5.0), lo))
    n_steps += 1
  if stop_after is not None and n_steps >= stop_after:   # EarlyStopDopri5.advance: the state where it stopped
    out[1].copy_(y)
  return out


def _solve_dopri5_device(func, y0, t, rtol, atol, trials_per_sync=None, evaluator=None, stop_after=None, pair='dopri5'):
  """dopri5 with the controller on the device (csrc/dopri5.hip): a trial step is one hipGraph replay; accept / reject, the
  step-size update, the end-point interpolation and the commit are decided by kernels from a record in device memory, which the
  host reads once per `trials_per_sync` trial steps.  Same arithmetic as `_solve_dopri5_native` (which stays for callers that
  hook into every step).  evaluator + stop_after: the early-stopping test integrator -- the decoder / arg-max / split counts run
  as kernels behind every trial step, gated by the controller, and the solve gives up after `stop_after` trial steps
  (gnpde_dopri5_set_early_stop); returns (out, times of the accepted steps) then."""
  from . import ops
  from .utils import MaxNFEException
  room = func.opt['max_nfe'] + 1 - func.nfe          # evaluations the reference would still allow before it raises
  if room <= 0:
    raise MaxNFEException
  st = func.__dict__.setdefault('_dopri5_device', {})
  # Relabelled graph (graph.LocalityView) under the same rule as the fixed-step solves (opt['gnpde_reorder'] / GNPDE_REORDER, 'auto' by
  # default).  The error norm of a trial step is a sum over the rows; its squares are accumulated in double (csrc/misc.hip), so the
  # float32 ratio -- and with it every accept / reject decision and step size -- is the same on the relabelled graph as on the graph as
  # given (the states themselves are bit-identical row by row: the entries of a row keep their order)
  view = func._locality_view(y0) if hasattr(func, '_locality_view') else None
  if evaluator is not None and view is not None:
    evaluator = evaluator.relabelled(view)
  key = (tuple(y0.shape), str(y0.device), float(rtol), float(atol), id(view))
  ent = st.get(key)
  y0c = y0.detach()
  if ent is None:
    ent = {'y': _lib.alloc_state(y0c.shape[0], y0c.shape[1], y0c.device),
           'x0': _lib.alloc_state(y0c.shape[0], y0c.shape[1], y0c.device) if func.opt['add_source'] else None,
           'solver': None, 'sig': None, 'view': view}
    for old in st.values():
      if old['solver'] is not None:
        old['solver'].close()
    st.clear()   # one live solver per function object: its workspace is 12 state-sized buffers
    fs = func.__dict__.get('_solver_state')       # ... shared with the fixed-step path
    if fs:
      for old in fs.values():
        if old.get('solver') is not None:
          old['solver'].close()
      fs.clear()
    st[key] = ent
  if ent['x0'] is not None:
    if func.x0 is None:
      raise _lib.GnpdeError('add_source is set but x0 was never assigned (call ODEblock.set_x0)')
    if view is None:
      ent['x0'].copy_(func.x0)
    else:
      view.enter(func.x0.detach(), out=ent['x0'])
  # (ent['y'] only fixes the row stride; the solver owns its buffers)
  desc = func._descriptor(ent['y'], x0_override=ent['x0'], graph=None if view is None else view.graph)
  sig = func._descriptor_signature(desc)
  if ent['solver'] is None or ent['sig'] != sig:
    if ent['solver'] is not None:
      ent['solver'].close()
    ent['solver'] = ops.Dopri5Solver(desc, rtol, atol, y0.device)
    ent['sig'] = sig
  sol = ent['solver']
  if getattr(sol, 'pair', 'dopri5') != pair:      # (`pair`: 'dopri5' or torchdiffeq's 'adaptive_heun' -- the same controller, another trial step)
    sol.set_pair(pair)
  if evaluator is not None:
    if getattr(sol, 'evaluator', None) is not evaluator or getattr(sol, 'max_trial_steps', None) != int(stop_after):
      sol.set_early_stop(evaluator, int(stop_after))
  elif getattr(sol, 'evaluator', None) is not None:
    sol.set_early_stop(None)
  if trials_per_sync is None:
    # launch-bound sizes: keep the queue full between reads; large states: a read per trial step costs nothing next to six
    # evaluations and nothing is replayed past the end point
    trials_per_sync = 8 if y0c.numel() < (1 << 22) else 1
    if os.environ.get('GNPDE_DOPRI5_TRIALS_PER_SYNC'):          # A/B runs
      trials_per_sync = max(1, int(os.environ['GNPDE_DOPRI5_TRIALS_PER_SYNC']))
  out = torch.empty((2,) + tuple(y0.shape), dtype=y0.dtype, device=y0.device)
  out[0].copy_(y0c)
  # relabelled graph: the permutation rides in the solver's own first and last copy (gnpde_dopri5_set_row_order) instead of two
  # index_select launches around it
  order32 = None
  if view is not None:
    order32 = view.__dict__.get('order32')
    if order32 is None:
      order32 = view.__dict__['order32'] = view.order.to(torch.int32).contiguous()
  if getattr(sol, '_row_order', None) is not order32:
    sol.set_row_order(order32)
  t0_, t1_ = end_points(t)
  finished = ent['solver'].run(y0c, t0_, t1_, out[1], trials_per_sync=trials_per_sync, max_evals=room)
  spent = ent['solver'].stats()['evals']
  func._dopri5_stats = ent['solver'].stats()
  if not finished or spent > room:
    func.nfe += min(spent, room)
    raise MaxNFEException
  func.nfe += spent
  if evaluator is not None:
    n_acc = ent['solver'].stats()['accepted']
    return out, ent['solver'].times[:n_acc + 1].tolist()
  return out


# --------------------------------------------------------------------------------------------------
# recorded dopri5: training WITHOUT the adjoint method (reference default opt['adjoint'] = False; best_params Cora / Citeseer)
# --------------------------------------------------------------------------------------------------
def _tape_busy(func, name):
  """True while the record of an earlier forward pass of this function is still waiting for its backward (the autograd node of that pass
  is alive and has not run): the next differentiated solve then takes the host loop instead of overwriting it -- two forwards before
  one backward (siamese losses, gradient accumulation over two graphs) stay differentiable, as in the reference."""
  ref = func.__dict__.get(name)
  ctx = ref() if ref is not None else None
  return ctx is not None and not getattr(ctx, 'gnpde_consumed', False)


def _tape_claim(func, name, ctx):
  import weakref
  func.__dict__[name] = weakref.ref(ctx)


_TAPE_GROWTH_CAP_BYTES = 96 << 30     # a recorded dopri5 solve whose tape would outgrow this runs the host loop instead (round-5 advisor item)


class _TapeTooLong(_lib.GnpdeError):
  pass


_TAPE_BUDGET_BYTES = 8 << 30     # first tape allocation (it grows when a solve accepts more steps than it holds)


def _recorded_ok(func, y0, t, options):
  """The differentiated dopri5 solve runs as ONE recorded device solve + ONE native reverse sweep (csrc/dopri5.hip,
  gnpde_dopri5_set_tape / _tape_backward) for the Laplacian function -- f is linear in the state, its weights are constants of
  the solve that may carry gradients (attention block) -- with alpha' = sigmoid(alpha_train).  Everything else keeps the host
  loop `_solve_dopri5` over the kernel-backed autograd Functions of autograd.py."""
  if func.__class__.__name__ != 'LaplacianODEFunc' or not hasattr(func, '_descriptor'):
    return False
  if not (y0.is_cuda and y0.dim() == 2 and y0.dtype == torch.float32 and y0.shape[1] <= 256 and len(t) == 2 and t.dtype == torch.float32):
    return False
  opt = func.opt
  if opt.get('no_alpha_sigmoid') or opt.get('gnpde_host_dopri5_training') or opt.get('gnpde_composite_backward'):
    return False
  if os.environ.get('GNPDE_HOST_DOPRI5_TRAINING', '0') == '1':      # A/B runs (bench.py --config cora-epoch)
    return False
  if options.get('norm') is not None or options.get('host_controller') or options.get('eager_stages'):
    return False
  return torch.is_grad_enabled() and func._needs_grad(y0)


class _RecordedDopri5(torch.autograd.Function):
  """Forward: the device-controlled dopri5 solve (one hipGraph per trial step) that leaves the stage inputs of every ACCEPTED step
  on a tape.  Backward: what autograd does through torchdiffeq's accepted steps (step sizes constants: misc.py _optimal_step_size
  is decorated with torch.no_grad), as one native reverse sweep -- one fused row-kernel launch per evaluation, no PyTorch op and no
  host synchronisation inside (oracle/tape_reverse.py states the algebra; tests/test_tape_gpu.py)."""

  @staticmethod
  def forward(ctx, y0, edge_values, alpha_train, beta_train, func, t0, t1, rtol, atol):
    from . import ops
    from .utils import MaxNFEException
    room = func.opt['max_nfe'] + 1 - func.nfe
    if room <= 0:
      raise MaxNFEException
    y0c = _lib.f32c(y0.detach())
    n, d = y0c.shape
    st = func.__dict__.setdefault('_tape_state', {})
    key = (n, d, str(y0c.device), float(rtol), float(atol))
    ent = st.get(key)
    if ent is None:
      for old in st.values():
        if old.get('solver') is not None:
          old['solver'].close()
      st.clear()
      ent = st[key] = {'y': _lib.alloc_state(n, d, y0c.device),
                       'x0': _lib.alloc_state(n, d, y0c.device) if func.opt['add_source'] else None, 'solver': None, 'sig': None}
    if ent['x0'] is not None:
      if func.x0 is None:
        raise _lib.GnpdeError('add_source is set but x0 was never assigned (call ODEblock.set_x0)')
      ent['x0'].copy_(func.x0.detach())
    graph = func._graph(y0c)
    desc = func._descriptor(ent['y'], x0_override=ent['x0'], graph=graph)      # (refreshes the CSR-ordered weights in place)
    sig = func._descriptor_signature(desc)
    if ent['solver'] is None or ent['sig'] != sig:
      if ent['solver'] is not None:
        ent['solver'].close()
      ent['solver'] = ops.Dopri5Solver(desc, rtol, atol, y0c.device)
      ent['sig'] = sig
      state_bytes = max(n * desc.struct.ld * 4, 1)
      ent['solver'].set_tape(int(max(8, min(64, _TAPE_BUDGET_BYTES // (6 * state_bytes)))))
    sol = ent['solver']
    out = torch.empty((2, n, d), dtype=torch.float32, device=y0c.device)
    out[0].copy_(y0c)
    tps = 8 if y0c.numel() < (1 << 22) else 1
    while True:
      try:
        finished = sol.run(y0c, t0, t1, out[1], trials_per_sync=tps, max_evals=room)
        break
      except _lib.GnpdeError as exc:
        if 'do not fit the tape' not in str(exc):
          raise
        torch.cuda.synchronize(y0c.device)
        state_bytes = max(n * desc.struct.ld * 4, 1)
        if 2 * sol.tape_capacity * 6 * state_bytes > _TAPE_GROWTH_CAP_BYTES:
          raise _TapeTooLong('recorded dopri5: %d accepted steps do not fit a tape of %.0f GB' % (sol.tape_capacity, _TAPE_GROWTH_CAP_BYTES / 1e9))
        sol.set_tape(2 * sol.tape_capacity)           # more accepted steps than slots: a longer tape, and the solve again
    stats = sol.stats()
    func._dopri5_stats = dict(stats, recorded=True)
    spent = stats['evals']
    if not finished or spent > room:
      func.nfe += min(spent, room)
      raise MaxNFEException
    func.nfe += spent
    sol.tape_generation += 1
    gt, t_from_csr = graph.transposed_positions()
    w_csr = func._weights_csr(graph)
    w_t = torch.index_select(w_csr[:graph.e], 0, t_from_csr[:graph.e].long()) if graph.e > 0 else w_csr
    ctx.func, ctx.sol, ctx.gen, ctx.gt, ctx.e = func, sol, sol.tape_generation, gt, graph.e
    ctx.x0 = ent['x0']
    ctx.x0_version = None if ent['x0'] is None else ent['x0']._version
    ctx.heads = edge_values.shape[1] if edge_values.dim() == 2 else 0
    ctx.save_for_backward(w_t, alpha_train, beta_train)
    func._last_train_solve = 'native recorded-tape dopri5'
    _tape_claim(func, '_tape_live_dopri5', ctx)
    return out

  @staticmethod
  def backward(ctx, grad_out):
    w_t, alpha_train, beta_train = ctx.saved_tensors
    sol, gt, E = ctx.sol, ctx.gt, ctx.e
    ctx.gnpde_consumed = True
    if sol.handle is None or sol.tape_generation != ctx.gen:
      raise _lib.GnpdeError('recorded dopri5: the tape of this forward pass was overwritten by a later solve of the same function '
                            '(backward must run before the next training forward; opt["gnpde_host_dopri5_training"] = True keeps a tape per forward)')
    need = ctx.needs_input_grad
    with torch.no_grad():
      g1 = grad_out[1].contiguous()
      gy0, r_t, sum_g, dot = sol.tape_backward(gt, w_t, g1)
      a = torch.sigmoid(alpha_train.detach().reshape(()))
      dy0 = gy0 + grad_out[0] if need[0] else None
      dw = None
      if need[1] and E > 0:
        dw_e = torch.empty(E, dtype=torch.float32, device=g1.device)
        dw_e[gt.perm_long] = a * r_t[:E]
        dw = (dw_e / ctx.heads).unsqueeze(1).expand(E, ctx.heads).contiguous() if ctx.heads else dw_e
      dalpha = (dot.reshape(()) * (1 - a)).reshape(alpha_train.shape) if need[2] else None
      dbeta = None
      if need[3] and ctx.x0 is not None:
        if ctx.x0._version != ctx.x0_version:
          raise _lib.GnpdeError('recorded dopri5: the source term of this forward pass was overwritten before its backward')
        dbeta = (sum_g * ctx.x0).sum().reshape(beta_train.shape)
    return dy0, dw, dalpha, dbeta, None, None, None, None, None


def _solve_dopri5_recorded(func, y0, t, rtol, atol):
  if _tape_busy(func, '_tape_live_dopri5'):
    func._last_train_solve = 'differentiable host loop (the record of an earlier forward pass still awaits its backward)'
    return _solve_dopri5(func, y0, t, rtol, atol)
  t0_, t1_ = end_points(t)
  nfe0 = func.nfe
  try:
    return _RecordedDopri5.apply(y0, func._edge_values(), func.alpha_train, func.beta_train, func, t0_, t1_, float(rtol),
                                 float(atol))
  except _TapeTooLong as exc:
    func.nfe = nfe0
    func._last_train_solve = 'differentiable host loop (%s)' % exc
    return _solve_dopri5(func, y0, t, rtol, atol)


# --------------------------------------------------------------------------------------------------
# recorded fixed-grid solve: training WITHOUT the adjoint method through euler / midpoint / rk4 -- `python run_GNN.py --function
# transformer --block constant --method rk4` (reference run_GNN.py:336 `--adjoint` is store_true, base_classes.py:44-47 then picks
# torchdiffeq.odeint and loss.backward() runs through its Python loop, run_GNN.py:62-96)
# --------------------------------------------------------------------------------------------------
def _transformer_stage_native(func):
  """GRAND-nl per-evaluation attention the native VJP stage of csrc/adjoint.hip covers: scaled-dot scores, and (round 6) cosine_sim /
  pearson -- the scaled dot product of unit (mean-centred) head vectors, d_k in {4, 8, 16} -- and exp_kernel, with any normaliser."""
  lay, opt = func.multihead_att_layer, func.opt
  a4 = lay.attention_dim // 4
  if opt['mix_features'] or getattr(lay, 'split_kernel', False):
    return False
  if not (lay.d_k % 4 == 0 and lay.attention_dim % 4 == 0 and a4 <= 64 and (a4 & (a4 - 1)) == 0):
    return False
  if opt['attention_type'] == 'scaled_dot':
    return True
  if opt['attention_type'] == 'exp_kernel':       # (round 6; the BLEND split kernel -- two exp kernels multiplied -- keeps the stage loop)
    return lay.attention_dim <= 128 and lay.h <= 8
  return opt['attention_type'] in ('cosine_sim', 'pearson') and lay.d_k in (4, 8, 16)


def _gat_stage_native(func):
  """The GAT function (reference src/function_GAT_attention.py) on the native VJP stage: 2..8 heads, attention_dim a multiple of 4."""
  lay, opt = func.multihead_att_layer, func.opt
  return (not opt['mix_features']) and 2 <= lay.h <= 8 and lay.attention_dim % 4 == 0 and lay.attention_dim <= 256


def _stage_projection_t(func, ex, dev):
  """[d, m] projection weights of the per-evaluation attention, transposed for P = d(q||k) W, in a buffer with a persistent address."""
  if func.__class__.__name__ == 'ODEFuncAtt':
    src = func.multihead_att_layer.W.detach()                    # [d, A] as the reference stores it
  else:
    wqk, _ = func.multihead_att_layer.qk_weights()
    src = wqk.t()
  if ex.get('proj_wt') is None or ex['proj_wt'].shape != tuple(src.shape):
    ex['proj_wt'] = torch.empty(tuple(src.shape), dtype=torch.float32, device=dev)
  ex['proj_wt'].copy_(src)                 # refreshed in place: the captured graph keeps the pointer
  return ex['proj_wt']


def _recorded_fixed_ok(func, y0, t, method):
  """The differentiated fixed-grid solve runs as ONE recorded native solve (csrc/solver.hip, gnpde_solver_set_tape: the captured
  hipGraph of the inference solve with every stage input written to a slot of its own) + ONE native reverse sweep over the record
  (csrc/adjoint.hip, gnpde_adjoint_set_tape: the VJP stage kernels of the adjoint solve, one hipGraph) for GRAND-l (weights may carry
  gradients: attention block) and GRAND-nl with scaled-dot scores, alpha' = sigmoid(alpha_train).  Everything else keeps the host
  loop `_solve_fixed_host` over the kernel-backed autograd Functions of autograd.py."""
  if method not in ('euler', 'rk4', 'midpoint') or not hasattr(func, '_descriptor'):
    return False
  if not (y0.is_cuda and y0.dim() == 2 and y0.dtype == torch.float32 and y0.shape[1] <= 256 and len(t) == 2):
    return False
  opt = func.opt
  if opt.get('gnpde_composite_backward') or opt.get('gnpde_host_fixed_training') or opt.get('gnpde_shard'):
    return False
  if os.environ.get('GNPDE_HOST_FIXED_TRAINING', '0') == '1':      # A/B runs (bench.py --train --no-adjoint)
    return False
  if not (torch.is_grad_enabled() and func._needs_grad(y0)):
    return False
  kind = func.__class__.__name__
  if kind == 'LaplacianODEFunc':
    return True
  if kind == 'ODEFuncTransformerAtt':
    return _transformer_stage_native(func)
  if kind == 'ODEFuncAtt':
    return _gat_stage_native(func)
  return False


def _grad_vector_by_param(func, g, d):
  """{id(parameter): its slice of the native gradient vector} (gnpde_adjoint_run: d[Wq;Wk], d[bq;bk], d alpha_train, d beta_train)."""
  by_param = {}
  tail = 0
  if func.__class__.__name__ == 'ODEFuncTransformerAtt':
    lay = func.multihead_att_layer
    A = lay.attention_dim
    gram = g[:2 * A * d].view(2 * A, d)
    gb = g[2 * A * d:2 * A * d + 2 * A]
    by_param = {id(lay.Q.weight): gram[:A], id(lay.K.weight): gram[A:], id(lay.Q.bias): gb[:A], id(lay.K.bias): gb[A:]}
    tail = 2 * A * d + 2 * A
    if func.opt['attention_type'] == 'exp_kernel':      # two more slots: d output_var, d lengthscale
      by_param[id(lay.output_var)] = g[tail].reshape(lay.output_var.shape)
      by_param[id(lay.lengthscale)] = g[tail + 1].reshape(lay.lengthscale.shape)
      tail += 2
  elif func.__class__.__name__ == 'ODEFuncAtt':      # d W^T [A, d], then d a in the first 2 d_k of the A slots behind it
    lay = func.multihead_att_layer
    A = lay.attention_dim
    by_param = {id(lay.W): g[:A * d].view(A, d).t(), id(lay.a): g[A * d:A * d + 2 * lay.d_k].reshape(lay.a.shape)}
    tail = A * d + A
  by_param[id(func.alpha_train)] = g[tail].reshape(func.alpha_train.shape)
  if func.opt['add_source']:
    by_param[id(func.beta_train)] = g[tail + 1].reshape(func.beta_train.shape)
  return by_param


class _RecordedFixedGrid(torch.autograd.Function):
  """Forward: the native fixed-grid solver with a tape (no extra pass: the stage epilogues write their outputs into the tape's slots).
  Backward: the reverse sweep through the recorded evaluations -- per evaluation the VJP stage of the native adjoint solve (projection,
  attention, row kernel with the edge products, normaliser backward, d q / d k, aggregation on the transposed graph whose epilogue forms
  the next cotangent, parameter-gradient pass), all steps in one hipGraph, no PyTorch op and no host synchronisation inside."""

  @staticmethod
  def forward(ctx, y0, edge_values, func, method, dts, *params):
    from . import ops
    from .utils import MaxNFEException
    n_evals = len(dts) * _EVALS_PER_STEP[method]
    room = func.opt['max_nfe'] + 1 - func.nfe
    if n_evals > room:
      func.nfe += max(room, 0)
      raise MaxNFEException
    y0c = _lib.f32c(y0.detach())
    n, d = y0c.shape
    view = func._locality_view(y0c) if hasattr(func, '_locality_view') else None
    st = func.__dict__.setdefault('_fixed_tape_state', {})
    key = (method, tuple(dts), n, d, str(y0c.device), id(view))
    ent = st.get(key)
    if ent is None:
      for old in st.values():
        for name in ('solver', 'sweep'):
          if old.get(name) is not None:
            old[name].close()
      st.clear()
      ent = st[key] = {'y': _lib.alloc_state(n, d, y0c.device), 'a': _lib.alloc_state(n, d, y0c.device),
                       'x0': _lib.alloc_state(n, d, y0c.device) if func.opt['add_source'] else None,
                       'solver': None, 'sweep': None, 'sig': None, 'sweep_sig': None, 'view': view, 'extra': {}}
    if view is None:
      ent['y'].copy_(y0c)
    else:
      view.enter(y0c, out=ent['y'])
    if ent['x0'] is not None:
      if func.x0 is None:
        raise _lib.GnpdeError('add_source is set but x0 was never assigned (call ODEblock.set_x0)')
      if view is None:
        ent['x0'].copy_(func.x0.detach())
      else:
        view.enter(func.x0.detach(), out=ent['x0'])
    graph = func._graph(y0c) if view is None else view.graph
    desc = func._descriptor(ent['y'], x0_override=ent['x0'], graph=graph)      # (refreshes the CSR-ordered weights in place)
    sig = func._descriptor_signature(desc)
    if ent['solver'] is None or ent['sig'] != sig:
      for name in ('solver', 'sweep'):
        if ent[name] is not None:
          ent[name].close()
      ent['sweep'] = None
      ent['solver'] = ops.FixedStepSolver(desc, method, dts, y0c.device)
      ent['solver'].set_tape(True)
      ent['sig'] = sig
    sol = ent['solver']
    sol.run(ent['y'])
    func.nfe += n_evals
    sol.tape_generation += 1
    out = torch.empty((2, n, d), dtype=torch.float32, device=y0c.device)
    out[0].copy_(y0c)
    if view is None:
      out[1].copy_(ent['y'])
    else:
      view.leave(ent['y'], out=out[1])
    ctx.func, ctx.ent, ctx.gen, ctx.desc, ctx.graph, ctx.view = func, ent, sol.tape_generation, desc, graph, view
    ctx.method, ctx.dts, ctx.params = method, dts, params
    ctx.heads = 0 if edge_values is None else (edge_values.shape[1] if edge_values.dim() == 2 else 0)
    ctx.has_edge_values = edge_values is not None
    func._last_train_solve = 'native recorded fixed-grid %s' % method
    _tape_claim(func, '_tape_live_fixed', ctx)
    return out

  @staticmethod
  def backward(ctx, grad_out):
    from . import ops
    func, ent, graph, view, desc = ctx.func, ctx.ent, ctx.graph, ctx.view, ctx.desc
    ctx.gnpde_consumed = True
    sol = ent['solver']
    if sol is None or sol.handle is None or sol.tape_generation != ctx.gen:
      raise _lib.GnpdeError('recorded fixed-grid solve: the tape of this forward pass was overwritten by a later solve of the same function '
                            '(backward must run before the next training forward; opt["gnpde_host_fixed_training"] = True keeps a graph per forward)')
    need = ctx.needs_input_grad
    nl = func.__class__.__name__ in ('ODEFuncTransformerAtt', 'ODEFuncAtt')
    with torch.no_grad():
      n, d = grad_out.shape[1], grad_out.shape[2]
      dev = grad_out.device
      gt, t_from_csr = graph.transposed_positions()
      ex = ent['extra']
      proj_wt = w_t = None
      if nl:
        proj_wt = _stage_projection_t(func, ex, dev)
      else:
        w_csr = func._weights_csr(graph)
        if ex.get('w_t') is None or ex['w_t'].numel() != max(graph.e, 1):
          ex['w_t'] = torch.empty(max(graph.e, 1), dtype=torch.float32, device=dev)
          ex['r_acc'] = torch.zeros(max(graph.e, 1), dtype=torch.float32, device=dev)
        if graph.e > 0:
          torch.index_select(w_csr[:graph.e], 0, t_from_csr.long(), out=ex['w_t'][:graph.e])
        w_t = ex['w_t']
      want_dw = (not nl) and ctx.has_edge_values and need[1] and graph.e > 0
      sweep_sig = (ent['sig'], id(gt), bool(want_dw), id(sol))
      if ent['sweep'] is None or ent['sweep_sig'] != sweep_sig:
        if ent['sweep'] is not None:
          ent['sweep'].close()
        ent['sweep'] = ops.AdjointSolver(desc, gt, t_from_csr if nl else None, proj_wt, w_t, ctx.method, ctx.dts, dev)
        csr_from_t = None
        if nl and graph.e > 0:       # the inverse position map: lets the sweep gather the cotangent rows only (csrc/adjoint.hip)
          csr_from_t = graph.__dict__.get('_csr_from_t')
          if csr_from_t is None:
            csr_from_t = torch.empty_like(t_from_csr)
            csr_from_t[t_from_csr.long()] = torch.arange(graph.e, dtype=t_from_csr.dtype, device=dev)
            graph.__dict__['_csr_from_t'] = csr_from_t
        ent['sweep'].set_tape(sol.tape, ex['r_acc'] if want_dw else None, csr_from_t)
        ent['grads'] = torch.zeros(ent['sweep'].n_grad, dtype=torch.float32, device=dev)
        ent['sweep_sig'] = sweep_sig
      ab = ent['a']
      g1 = grad_out[1]
      if view is None:
        ab.copy_(g1)
      else:
        view.enter(g1.contiguous(), out=ab)
      ent['sweep'].run(ent['y'], ab, ent['grads'])      # (y is not read in the taped mode: the state comes from the tape)
      dy0 = None
      if need[0]:
        dy0 = torch.empty((n, d), dtype=torch.float32, device=dev)
        if view is None:
          dy0.copy_(ab)
        else:
          view.leave(ab, out=dy0)
        dy0 += grad_out[0]
      dw = None
      if want_dw:
        a = func.alpha_train.detach().reshape(())
        if not func.opt['no_alpha_sigmoid']:
          a = torch.sigmoid(a)
        E = graph.e
        dw_e = torch.empty(E, dtype=torch.float32, device=dev)
        order = gt.perm_long if ent['sweep'].swapped else graph.perm_long      # (the products lie in the order of the graph the row kernel ran on)
        dw_e[order] = a * ex['r_acc'][:E]
        dw = (dw_e / ctx.heads).unsqueeze(1).expand(E, ctx.heads).contiguous() if ctx.heads else dw_e
      by_param = _grad_vector_by_param(func, ent['grads'], d)
      gparams = []
      for i, p in enumerate(ctx.params):
        g = by_param.get(id(p)) if need[5 + i] else None
        gparams.append(None if g is None else g.clone(memory_format=torch.contiguous_format))
    return (dy0, dw, None, None, None) + tuple(gparams)


_FIXED_TAPE_BUDGET_BYTES = 96 << 30     # a recorded fixed-grid solve that would need more than this (or more than 70 % of the free device memory) runs the host loop


def _fixed_tape_estimate(func, y0, n_evals):
  """Bytes of the record of a fixed-grid solve: n_evals + 1 stage inputs and, for GRAND-nl with scaled-dot scores, q||k and the weights
  of every evaluation (csrc/solver.hip gnpde_solver_tape_bytes; an upper estimate without building the descriptor)."""
  n, d = y0.shape
  ld = (d + 3) // 4 * 4
  total = (n_evals + 1) * n * ld * 4
  lay = getattr(func, 'multihead_att_layer', None)
  if func.__class__.__name__ == 'ODEFuncTransformerAtt' and lay is not None and func.edge_index is not None:
    total += n_evals * (n * 2 * lay.attention_dim * 4 + int(func.edge_index.shape[1]) * 4)
  return total


def _solve_fixed_recorded(func, y0, t, method, step_size):
  dts = step_sizes(t, step_size)
  need = _fixed_tape_estimate(func, y0, len(dts) * _EVALS_PER_STEP[method])
  try:
    free = torch.cuda.mem_get_info(y0.device)[0]
  except Exception:   # noqa: BLE001
    free = _FIXED_TAPE_BUDGET_BYTES
  if _tape_busy(func, '_tape_live_fixed'):
    func._last_train_solve = 'differentiable host loop (the record of an earlier forward pass still awaits its backward)'
    return _solve_fixed_host(func, y0, t, method, step_size)
  cached = func.__dict__.get('_fixed_tape_state')          # (a tape of this shape that already exists is not allocated again)
  if not cached and need > min(_FIXED_TAPE_BUDGET_BYTES, 0.7 * free):
    func._last_train_solve = 'differentiable host loop (the record of %d evaluations would take %.1f GB)' % (len(dts) * _EVALS_PER_STEP[method], need / 1e9)
    return _solve_fixed_host(func, y0, t, method, step_size)
  params = tuple(p for p in func.parameters() if p.requires_grad)
  edge_values = None
  if func.__class__.__name__ == 'LaplacianODEFunc':
    ev = func._edge_values()
    edge_values = ev if ev.requires_grad else None
  return _RecordedFixedGrid.apply(y0, edge_values, func, method, dts, *params)


class _TupleFunc(object):
  """A function of a tuple state seen as a function of the flattened concatenation (torchdiffeq misc.py _TupleFunc):
  the regularised training state (x, r_1, ..., r_k) of reference src/block_constant.py:40-43."""

  def __init__(self, func, shapes):
    self.func, self.shapes = func, shapes

  def __call__(self, t, flat):
    out = self.func(t, tuple(_unflatten(flat, self.shapes)))
    return _flatten(out)

  def parameters(self):
    return self.func.parameters() if isinstance(self.func, torch.nn.Module) else iter(())


def _solve_tuple(solver, func, y0, t, kw):
  shapes = [p.shape for p in y0]
  if kw.get('method') in (None, 'dopri5', 'adaptive_heun'):     # torchdiffeq's default norm of a tuple state
    kw = dict(kw, options=dict(kw.get('options') or {}))
    kw['options'].setdefault('norm', _mixed_norm(shapes))
  flat = solver(_TupleFunc(func, shapes), _flatten(y0), t, **kw)          # [len(t), total]
  outs, pos = [], 0
  for sh in shapes:
    cnt = int(torch.Size(sh).numel())
    outs.append(flat[:, pos:pos + cnt].reshape((flat.shape[0],) + tuple(sh)))
    pos += cnt
  return tuple(outs)


def odeint(func, y0, t, *, rtol=1e-7, atol=1e-9, method=None, options=None, use_graph=True, **adjoint_kwargs):
  """Drop-in for torchdiffeq.odeint on this path.  Unknown options (e.g. `max_iters`, which the
  reference passes and torchdiffeq ignores with a warning) are ignored."""
  if isinstance(y0, tuple):
    return _solve_tuple(odeint, func, y0, t, dict(rtol=rtol, atol=atol, method=method, options=options, use_graph=use_graph))
  options = dict(options or {})
  method = 'dopri5' if method is None else method
  if method in ('euler', 'rk4', 'midpoint'):
    step_size = options.get('step_size', None)
    if step_size is None:
      raise ValueError('fixed-grid methods need options["step_size"]')
    if _native_ok(func, y0, t):
      from . import distributed as D
      if D.shard_requested(func):     # one process per GPU, rows partitioned over the ranks (distributed.solve_sharded)
        if method == 'midpoint':
          raise _lib.GnpdeError('gnpde_shard: the row-partitioned solver runs euler, rk4, dopri5 and adaptive_heun; unset gnpde_shard for midpoint')
        return D.solve_sharded(func, y0, t, method, step_size, use_graph=use_graph)
      return _solve_native(func, y0, t, method, step_size, use_graph=use_graph)
    if _recorded_fixed_ok(func, y0, t, method):      # training without the adjoint method: recorded solve + native reverse sweep
      return _solve_fixed_recorded(func, y0, t, method, step_size)
    return _solve_fixed_host(func, y0, t, method, step_size)
  if method in ('dopri5', 'adaptive_heun') and hasattr(func, '_descriptor'):
    from . import distributed as D
    if D.shard_requested(func):
      # rows partitioned over the ranks; the controller runs on every rank over an all-reduced error norm
      if _native_ok(func, y0, t):
        return D.solve_sharded(func, y0, t, method, None, rtol=rtol, atol=atol)
      # (a replicated single-GPU solve on every rank would be a silent waste of N - 1 GPUs)
      raise _lib.GnpdeError('gnpde_shard is set but this %s solve cannot run row-partitioned (float32 [n, d] state on a HIP device, two '
                            'output times, no autograd) -- unset gnpde_shard for it' % method)
  if method == 'dopri5':
    if _recorded_ok(func, y0, t, options):      # training without the adjoint method: recorded solve + native reverse sweep
      return _solve_dopri5_recorded(func, y0, t, rtol, atol)
    if _native_ok(func, y0, t) and t.dtype == torch.float32 and not options.get('host_controller', False):
      if options.get('eager_stages', False):      # controller on the host, one scalar read per trial step
        return _solve_dopri5_native(func, y0, t, rtol, atol)
      return _solve_dopri5_device(func, y0, t, rtol, atol, trials_per_sync=options.get('trials_per_sync'))
    return _solve_dopri5(func, y0, t, rtol, atol, norm=options.get('norm'))
  if method == 'adaptive_heun':
    if (hasattr(func, '_descriptor') and _native_ok(func, y0, t) and t.dtype == torch.float32 and not options.get('host_controller', False)
        and options.get('norm') is None):
      # the device controller with the Heun pair's trial step (one evaluation per trial step; csrc/dopri5.hip, gnpde_dopri5_set_pair)
      return _solve_dopri5_device(func, y0, t, rtol, atol, trials_per_sync=options.get('trials_per_sync'), pair='adaptive_heun')
    return _solve_dopri5(func, y0, t, rtol, atol, tableau='adaptive_heun', norm=options.get('norm'))
  raise ValueError('unsupported method %r (euler, midpoint, rk4, dopri5, adaptive_heun)' % (method,))


# --------------------------------------------------------------------------------------------------
# adjoint sensitivity (torchdiffeq 0.2.1 adjoint.py, as selected by opt['adjoint'], reference
# src/base_classes.py:44-47, src/block_constant.py:45-55)
# --------------------------------------------------------------------------------------------------
def _flatten(parts):
  return torch.cat([p.reshape(-1) for p in parts])


def _unflatten(v, shapes):
  out, pos = [], 0
  for sh in shapes:
    cnt = int(torch.Size(sh).numel())
    out.append(v[pos:pos + cnt].view(sh))
    pos += cnt
  return out


def _adjoint_fixed_grid(func, params, y, a, gparams, span, method, step_size):
  """euler / rk4 (3/8 rule) on the augmented system over the reversed-time grid s = -t (torchdiffeq's grid: the
  short step lands next to the earlier time), written on the separate components (y, a, g_theta) with fused
  axpy updates instead of one flat vector: per stage one evaluation F = f(u_y) and one vector-Jacobian product
  (V_y, V_theta) = u_a^T df/d(y, theta); in s the derivatives are  y' = -F,  a' = +V_y,  g' = +V_theta.
  Same stage formulas as rk_common.rk4_alt_step_func; agrees with the flat-vector path to rounding (the products
  dt * k / 3 are formed as k * (dt / 3))."""
  grid = time_grid(span, step_size)
  dts = (grid[1:] - grid[:-1]).tolist()
  y, a = y.clone(), a.clone()
  gparams = [g.clone() for g in gparams]

  def stage(uy, ua, s_now):
    with torch.enable_grad():
      yy = uy.detach().requires_grad_(True)
      F = func(-s_now, yy)
      grads = torch.autograd.grad(F, (yy,) + tuple(params), ua, allow_unused=True)
    V = grads[0] if grads[0] is not None else torch.zeros_like(uy)
    return F.detach(), V, grads[1:]

  def comb(base, terms, inplace=False):
    """base + sum_j c_j v_j: one fused pass on the device (gnpde_lincomb), chained axpys elsewhere."""
    if base.is_cuda and base.dtype == torch.float32 and base.is_contiguous() and all(
        v.is_contiguous() and v.dtype == torch.float32 for v, _ in terms):
      from . import ops
      return ops.lincomb(base, terms, out=base if inplace else None)
    out = base if inplace else None
    for v, c in terms:
      out = torch.add(base, v, alpha=c) if out is None else out.add_(v, alpha=c)
    return out

  def acc_params(vps, coef):
    for g, v in zip(gparams, vps):
      if v is not None:
        g.add_(v, alpha=coef)

  for step, dt in enumerate(dts):
    s0, s1 = grid[step], grid[step + 1]
    if method == 'euler':
      F1, V1, P1 = stage(y, a, s0)
      y = comb(y, [(F1, -dt)], inplace=True)
      a = comb(a, [(V1, dt)], inplace=True)
      acc_params(P1, dt)
      continue
    third, eighth = dt / 3.0, dt * 0.125
    F1, V1, P1 = stage(y, a, s0)
    F2, V2, P2 = stage(comb(y, [(F1, -third)]), comb(a, [(V1, third)]), s0 + third)
    F3, V3, P3 = stage(comb(y, [(F2, -dt), (F1, third)]), comb(a, [(V2, dt), (V1, -third)]), s0 + 2 * third)
    F4, V4, P4 = stage(comb(y, [(F1, -dt), (F2, dt), (F3, -dt)]), comb(a, [(V1, dt), (V2, -dt), (V3, dt)]), s1)
    y = comb(y, [(F1, -eighth), (F2, -3 * eighth), (F3, -3 * eighth), (F4, -eighth)], inplace=True)
    a = comb(a, [(V1, eighth), (V2, 3 * eighth), (V3, 3 * eighth), (V4, eighth)], inplace=True)
    for P, c in ((P1, eighth), (P2, 3 * eighth), (P3, 3 * eighth), (P4, eighth)):
      acc_params(P, c)
  return a, gparams


def _adjoint_native_ok(func, y, method):
  """The fixed-grid adjoint solve runs as ONE native object (csrc/adjoint.hip) for GRAND-l and for GRAND-nl with scaled-dot, cosine_sim
  or pearson scores (any normaliser), alpha' = sigmoid(alpha_train) or the raw alpha_train; everything else (GAT, exp_kernel, the BLEND
  split kernel) keeps the stage-by-stage loop above."""
  if method not in ('euler', 'rk4') or not hasattr(func, '_descriptor'):
    return False
  if not (y.is_cuda and y.dim() == 2 and y.dtype == torch.float32 and y.shape[1] <= 256):
    return False
  opt = func.opt
  if opt.get('gnpde_composite_backward') or opt.get('gnpde_host_adjoint'):
    return False
  kind = func.__class__.__name__
  if kind == 'LaplacianODEFunc':
    return True       # the weights are constants of the solve: torchdiffeq's adjoint returns gradients for y0 and func.parameters() only
  if kind == 'ODEFuncTransformerAtt':
    return _transformer_stage_native(func)     # (round 6: cosine_sim / pearson scores and the raw alpha of opt['no_alpha_sigmoid'] as well)
  if kind == 'ODEFuncAtt':
    return _gat_stage_native(func)             # (round 6)
  return False


def _adjoint_native(func, params, y, a, span, method, step_size):
  """(a at the earlier time, [gradient contribution per entry of `params`]) of one backward interval, by the native solver."""
  from . import ops
  from .utils import MaxNFEException
  grid = time_grid(span.detach().to('cpu'), step_size)
  dts = (grid[1:] - grid[:-1]).tolist()
  n_evals = len(dts) * (4 if method == 'rk4' else 1)
  room = func.opt['max_nfe'] + 1 - func.nfe
  if n_evals > room:
    func.nfe += max(room, 0)
    raise MaxNFEException
  st = func.__dict__.setdefault('_adjoint_state', {})
  view = func._locality_view(y, forward_solve=False) if hasattr(func, '_locality_view') else None
  key = (method, tuple(dts), tuple(y.shape), str(y.device), id(view))
  ent = st.get(key)
  if ent is None:
    for old in st.values():
      if old.get('solver') is not None:
        old['solver'].close()
    st.clear()
    n, d = y.shape
    ent = st[key] = {'y': _lib.alloc_state(n, d, y.device), 'a': _lib.alloc_state(n, d, y.device),
                     'x0': _lib.alloc_state(n, d, y.device) if func.opt['add_source'] else None,
                     'solver': None, 'sig': None, 'view': view, 'extra': {}}
  yb, ab = ent['y'], ent['a']
  if view is None:
    yb.copy_(y.detach())
    ab.copy_(a.detach())
  else:
    view.enter(y.detach(), out=yb)
    view.enter(a.detach(), out=ab)
  if ent['x0'] is not None:
    if func.x0 is None:
      raise _lib.GnpdeError('add_source is set but x0 was never assigned (call ODEblock.set_x0)')
    if view is None:
      ent['x0'].copy_(func.x0.detach())
    else:
      view.enter(func.x0.detach(), out=ent['x0'])
  graph = func._graph(y) if view is None else view.graph
  desc = func._descriptor(yb, x0_override=ent['x0'], graph=graph)
  gt, t_from_csr = graph.transposed_positions()
  nl = func.__class__.__name__ in ('ODEFuncTransformerAtt', 'ODEFuncAtt')
  ex = ent['extra']
  proj_wt = w_t = None
  if nl:
    proj_wt = _stage_projection_t(func, ex, y.device)
  else:
    w_csr = func._weights_csr(graph)
    if ex.get('w_t') is None or ex['w_t'].numel() != max(graph.e, 1):
      ex['w_t'] = torch.empty(max(graph.e, 1), dtype=torch.float32, device=y.device)
    if graph.e > 0:
      torch.index_select(w_csr[:graph.e], 0, t_from_csr.long(), out=ex['w_t'][:graph.e])
    w_t = ex['w_t']
  sig = (func._descriptor_signature(desc), id(gt))
  if ent['solver'] is None or ent['sig'] != sig:
    if ent['solver'] is not None:
      ent['solver'].close()
    ent['solver'] = ops.AdjointSolver(desc, gt, t_from_csr if nl else None, proj_wt, w_t, method, dts, y.device)
    ent['grads'] = torch.zeros(ent['solver'].n_grad, dtype=torch.float32, device=y.device)
    ent['sig'] = sig
  sol = ent['solver']
  sol.run(yb, ab, ent['grads'])
  func.nfe += n_evals
  a_out = torch.empty_like(a, memory_format=torch.contiguous_format)
  if view is None:
    a_out.copy_(ab)
  else:
    view.leave(ab, out=a_out)
  by_param = _grad_vector_by_param(func, ent['grads'], y.shape[1])
  return a_out, [by_param.get(id(p)) for p in params]


def _adjoint_adaptive_ok(func, y, method):
  """The ADAPTIVE adjoint methods (`adjoint_method` adaptive_heun -- the reference's default, run_GNN.py:334; best_params Pubmed -- and
  dopri5 -- CoauthorCS, Computers) on the Laplacian function run with native stages: no autograd graph, no flat vector, two aggregation
  launches per stage (see _adjoint_adaptive_native).  Everything else keeps the flat host loop."""
  if method not in ('dopri5', 'adaptive_heun') or func.__class__.__name__ != 'LaplacianODEFunc' or not hasattr(func, '_descriptor'):
    return False
  if not (y.is_cuda and y.dim() == 2 and y.dtype == torch.float32):
    return False
  opt = func.opt
  return not (opt.get('no_alpha_sigmoid') or opt.get('gnpde_composite_backward') or opt.get('gnpde_host_adjoint'))


def _adjoint_adaptive_native(func, params, y, a, gparams, span, method, rtol, atol, safety=0.9, ifactor=10.0, dfactor=0.2):
  """One backward interval of torchdiffeq's adjoint with an ADAPTIVE adjoint method, for f(u) = alpha' (A u - u) + beta x0: the augmented
  system (vjp_t, y, a, g_theta) in reversed time s = -t is

      y' = -f(y),   a' = alpha' (A^T a - a),   g_alpha' = (1 - alpha') <a, f(y) - beta x0>,   g_beta' = <a, x0>,   everything else 0,

  integrated by the embedded pair with torchdiffeq 0.2.1's controller (host, float64; mixed norm = the largest component rms, the
  scalars being components of their own) exactly as `_solve_dopri5` does on the flat vector -- but on the components: per stage ONE
  launch of the aggregation on the graph (f, its epilogue forming the next stage input of y) and ONE on the transposed graph (the
  same for a), two dot products, no autograd graph; per trial step two error-norm launches and one host read.  Returns (a at the earlier
  time, new accumulated gradients of alpha_train / beta_train as a dict by parameter id)."""
  import numpy as np
  from . import ops
  f32 = np.float32
  tab = _TABLEAUS[method]
  order, stages = tab['order'], len(tab['alpha'])
  dev = y.device
  n, d = y.shape
  graph = func._graph(y)
  gt, t_from_csr = graph.transposed_positions()
  cache = func.__dict__.setdefault('_adjoint_adaptive', {})
  key = (n, d, str(dev), method)
  if cache.get('key') != key:
    cache.clear()
    cache['key'] = key
    cache['bufs'] = [_lib.alloc_state(n, d, dev) for _ in range(2 * (stages + 1) + 9)]
  bufs = cache['bufs']
  KF, KV = bufs[:stages + 1], bufs[stages + 1:2 * (stages + 1)]
  Y, Y1, A_, A1, UY0, UY1, UA0, UA1, X0 = bufs[2 * (stages + 1):]
  Y.copy_(y.detach())
  A_.copy_(a.detach())
  has_src = bool(func.opt['add_source'])
  if has_src:
    if func.x0 is None:
      raise _lib.GnpdeError('add_source is set but x0 was never assigned (call ODEblock.set_x0)')
    X0.copy_(func.x0.detach())
  desc_f = func._descriptor(Y, x0_override=X0 if has_src else None, graph=graph)
  w_csr = func._weights_csr(graph)
  w_t = torch.index_select(w_csr[:graph.e], 0, t_from_csr[:graph.e].long()) if graph.e > 0 else w_csr
  alpha_d = ops._scalar_dev(func.alpha_train, Y)
  desc_b = ops.RhsDescriptor(_lib.RHS_LAPLACIAN, gt, d, Y.stride(0), alpha_d, None, None, True, w_csr=w_t, padded_rows=_lib.is_padded(Y))
  one_minus_a = 1 - torch.sigmoid(func.alpha_train.detach().reshape(()))
  beta = func.beta_train.detach().reshape(()) if has_src else None
  ids = [id(p) for p in params]
  ia = ids.index(id(func.alpha_train)) if id(func.alpha_train) in ids else None
  ib = ids.index(id(func.beta_train)) if (has_src and id(func.beta_train) in ids) else None
  g = torch.zeros(2, dtype=torch.float32, device=dev)          # (g_alpha, g_beta) accumulated so far: they scale the tolerance of their components
  if ia is not None:
    g[0] = gparams[ia].reshape(())
  if ib is not None:
    g[1] = gparams[ib].reshape(())
  ratio_dev = torch.zeros(1, dtype=torch.float32, device=dev)
  err_ws = torch.empty(4096, dtype=torch.float32, device=dev)
  rt, at = float(rtol), float(atol)

  def dot(u, v):
    return (u * v).sum()

  def feval(uy, ua, j, nxt=None):
    """K_j of every component at the stage inputs (uy, ua); nxt = (out_uy, out_ua, coefficient row) forms the next stage inputs in the
    epilogues: u + sum_m (row_m dt) K_m with K_y = -F."""
    func._check_nfe()
    if nxt is None:
      ops.rhs_stage(desc_f, uy, _lib.STAGE_LINCOMB, out_k=KF[j])
      ops.rhs_stage(desc_b, ua, _lib.STAGE_LINCOMB, out_k=KV[j])
    else:
      out_uy, out_ua, row = nxt
      ops.rhs_stage(desc_f, uy, _lib.STAGE_LINCOMB, y=Y, out_k=KF[j], out_y=out_uy, prev=KF[:j], coef=[-c for c in row])
      ops.rhs_stage(desc_b, ua, _lib.STAGE_LINCOMB, y=A_, out_k=KV[j], out_y=out_ua, prev=KV[:j], coef=list(row))
    dx0 = dot(ua, X0) if has_src else torch.zeros((), device=dev)
    dF = dot(ua, KF[j])
    ka = one_minus_a * (dF - beta * dx0) if has_src else one_minus_a * dF
    return torch.stack([ka, dx0])

  def comp_rms(base0, base1, ks, coefs):
    ops.rk_error_ratio(base0, base1, ks, coefs, at, rt, ratio_dev, err_ws)
    return ratio_dev.clone()

  def scal_ratio(v, g0, g1):
    tol = at + rt * torch.max(g0.abs(), g1.abs())
    r = (v / tol).abs()
    if ia is None:
      r = r * torch.tensor([0.0, 1.0], device=dev)
    if ib is None:
      r = r * torch.tensor([1.0, 0.0], device=dev)
    return r.max().reshape(1)

  def mixed(parts):
    return float(torch.cat(parts).max().item())

  T0, T1 = float(span[0]), float(span[-1])
  Ks = [None] * (stages + 1)
  Ks[0] = feval(Y, A_, 0)
  # initial step (Hairer, Norsett & Wanner), as misc.py _select_initial_step over the mixed norm
  d0 = mixed([comp_rms(Y, Y, [Y], [1.0]), comp_rms(A_, A_, [A_], [1.0]), scal_ratio(g, g, g)])
  d1 = mixed([comp_rms(Y, Y, [KF[0]], [1.0]), comp_rms(A_, A_, [KV[0]], [1.0]), scal_ratio(Ks[0], g, g)])
  h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else float(f32(0.01) * f32(d0) / f32(d1))
  torch.add(Y, KF[0], alpha=-h0, out=UY0)
  torch.add(A_, KV[0], alpha=h0, out=UA0)
  Ks[1] = feval(UY0, UA0, 1)
  d2 = mixed([comp_rms(Y, Y, [KF[1], KF[0]], [1.0, -1.0]), comp_rms(A_, A_, [KV[1], KV[0]], [1.0, -1.0]), scal_ratio(Ks[1] - Ks[0], g, g)]) / h0
  h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else float(f32(f32(0.01) / f32(max(d1, d2))) ** f32(1.0 / order))
  dt = float(min(100 * h0, h1))
  if (ia is not None and (ib is not None or not has_src) and not func.opt.get('gnpde_host_controller_adjoint')
      and Y.stride(0) % 4 == 0 and d <= 256):
    # the rest of the interval with the controller on the device (csrc/adjoint_adaptive.hip: one hipGraph replay per trial step of the
    # embedded pair; the host reads a record once per batch)
    from .utils import MaxNFEException
    if cache.get('w_t') is None or cache['w_t'].numel() != max(graph.e, 1):
      cache['w_t'] = torch.empty(max(graph.e, 1), dtype=torch.float32, device=dev)
    if graph.e > 0:
      cache['w_t'][:graph.e].copy_(w_t)          # persistent address: the captured trial steps keep the pointer
    sig = (func._descriptor_signature(desc_f), id(gt), float(rt), float(at))
    if cache.get('solver') is None or cache.get('solver_sig') != sig:
      if cache.get('solver') is not None:
        cache['solver'].close()
      cache['solver'] = ops.AdjointAdaptiveSolver(desc_f, gt, cache['w_t'], method, rt, at, dev)
      cache['solver_sig'] = sig
    sol = cache['solver']
    room = func.opt['max_nfe'] + 1 - func.nfe
    if room <= 0:
      raise MaxNFEException
    finished = sol.run(Y, A_, g, T0, T1, dt, trials_per_sync=8 if Y.numel() < (1 << 22) else 1, max_evals=room)
    spent = sol.stats()['evals']
    func._adjoint_adaptive_stats = sol.stats()
    if not finished or spent > room:
      func.nfe += min(spent, room)
      raise MaxNFEException
    func.nfe += spent
    new = {ia: g[0].reshape(gparams[ia].shape)}
    if ib is not None:
      new[ib] = g[1].reshape(gparams[ib].shape)
    return A_[:, :d].clone(memory_format=torch.contiguous_format), new
  t_cur = T0
  a_out = None
  g_out = g
  while T1 > t_cur:
    assert t_cur + dt > t_cur, 'underflow in dt {}'.format(dt)
    dty = f32(dt)
    rows = [[float(f32(b) * dty) for b in row] for row in tab['beta']]
    # first stage input from the derivative carried over (rk_common: f1 = k[..., -1], also for the non-FSAL Heun pair)
    torch.add(Y, KF[0], alpha=-rows[0][0], out=UY0)
    torch.add(A_, KV[0], alpha=rows[0][0], out=UA0)
    uy, ua = [UY0, UY1], [UA0, UA1]
    src_y, src_a = UY0, UA0
    for r in range(1, stages + 1):               # K_r at stage input u_r; its epilogue forms u_{r+1}
      if r < stages:
        last = r == stages - 1 and tab['fsal']    # (first-same-as-last: the last stage input IS the solution)
        dst_y = Y1 if last else uy[r % 2]
        dst_a = A1 if last else ua[r % 2]
        Ks[r] = feval(src_y, src_a, r, (dst_y, dst_a, rows[r]))
        src_y, src_a = dst_y, dst_a
      else:
        Ks[r] = feval(src_y, src_a, r)

    def axpys(base, ks, cs, sign, out):
      """out = base + sign * sum_j cs_j ks_j (views of padded rows are fine: plain torch ops)"""
      first = True
      for kj, c in zip(ks, cs):
        if c == 0.0:
          continue
        if first:
          torch.add(base, kj, alpha=sign * c, out=out)
          first = False
        else:
          out.add_(kj, alpha=sign * c)
      if first:
        out.copy_(base)
      return out
    if tab['fsal']:
      sol = rows[stages - 1]
    else:
      sol = [float(f32(c) * dty) for c in tab['c_sol']]
      axpys(Y, KF, sol, -1.0, Y1)
      axpys(A_, KV, sol, 1.0, A1)
    g1 = g + sum(Ks[j] * sol[j] for j in range(len(sol)) if sol[j] != 0.0)
    cerr = [float(f32(e) * dty) for e in tab['c_err']]
    err_s = sum(Ks[j] * cerr[j] for j in range(stages + 1) if cerr[j] != 0.0)
    ratio = mixed([comp_rms(Y, Y1, KF, cerr), comp_rms(A_, A1, KV, cerr), scal_ratio(err_s, g, g1)])
    if _TRIAL_TRACE is not None:
      _TRIAL_TRACE.append((t_cur, dt, ratio))
    if ratio <= 1:
      t_next = t_cur + dt
      if t_next >= T1:     # the end time lies in this step: torchdiffeq's quartic interpolation (only a and the scalars are returned)
        xf = float(f32((T1 - t_cur) / (t_next - t_cur)))
        cmid = [float(f32(c) * dty) for c in tab['c_mid']]
        h = float(dty)

        def interp(v0, v1, k0, k1, mid):
          ca = 2 * h * (k1 - k0) - 8 * (v1 + v0) + 16 * mid
          cb = h * (5 * k0 - 3 * k1) + 18 * v0 + 14 * v1 - 32 * mid
          cc = h * (k1 - 4 * k0) - 11 * v0 - 5 * v1 + 16 * mid
          cd = h * k0
          return v0 + xf * cd + xf ** 2 * cc + xf ** 3 * cb + xf ** 4 * ca
        a_mid = axpys(A_, KV, cmid, 1.0, torch.empty_like(A_))
        a_out = interp(A_, A1, KV[0], KV[stages], a_mid)[:, :d].contiguous()
        g_mid = g + sum(Ks[j] * cmid[j] for j in range(stages + 1) if cmid[j] != 0.0)
        g_out = interp(g, g1, Ks[0], Ks[stages], g_mid)
      Y, Y1 = Y1, Y
      A_, A1 = A1, A_
      KF[0], KF[stages] = KF[stages], KF[0]
      KV[0], KV[stages] = KV[stages], KV[0]
      Ks[0] = Ks[stages]
      g = g1
      t_cur = t_next
      # (the descriptor of f reads y-independent operands only: alpha, beta, x0, the weights -- its `ld` is every buffer's)
    if ratio == 0:
      dt = dt * ifactor
    else:
      lo = 1.0 if ratio < 1 else dfactor
      dt = dt * min(ifactor, max(safety / ratio ** (1.0 / order), lo))
  new = {}
  if ia is not None:
    new[ia] = g_out[0].reshape(gparams[ia].shape)
  if ib is not None:
    new[ib] = g_out[1].reshape(gparams[ib].shape)
  return a_out, new


class _AdjointSolve(torch.autograd.Function):
  """Forward: the plain solve WITHOUT a tape -- on this package's functions that is the native hipGraph solver, so
  the training forward runs at inference speed and stores two states, not the trajectory.  Backward: the augmented
  system (vjp_t, y, a, g_theta) with  dy/dt = f,  da/dt = -a^T df/dy,  dg/dt = -a^T df/dtheta  is integrated from
  t[i] back to t[i-1] with the ADJOINT method / step size / tolerances; like torchdiffeq, the time reversal is the
  substitution s = -t (so a fixed grid puts its short step next to t[i-1]), the tuple is integrated as one flat
  vector (mixed linf-rms norm for adaptive methods), y is reset to the stored forward value and the incoming
  gradient is added at every output time.  f and its vector-Jacobian products run through the native kernels."""

  @staticmethod
  def forward(ctx, func, y0, t, fwd, adj, *params):
    ctx.func, ctx.adj = func, adj
    ans = odeint(func, y0.detach(), t, **fwd)
    ctx.save_for_backward(t, ans, *params)
    return ans

  @staticmethod
  def backward(ctx, grad_out):
    func, adj = ctx.func, ctx.adj
    t, ans, *params = ctx.saved_tensors
    params = tuple(params)
    with torch.no_grad():
      state = [torch.zeros((), dtype=ans.dtype, device=ans.device), ans[-1], grad_out[-1].contiguous()]
      state.extend(torch.zeros_like(p) for p in params)
      shapes = [s_.shape for s_ in state]

      def reversed_flat_dynamics(s, flat):
        """-(vjp_t, f, vjp_y, vjp_params) at time t = -s, vjp = grad(f, ., -a)  (adjoint.py augmented_dynamics
        under misc.py _ReverseFunc and _TupleFunc)."""
        parts = _unflatten(flat, shapes)
        with torch.enable_grad():
          # fresh allocations, not views into the flat vector: a view starts one float after the vjp_t slot, which
          # would push every kernel onto its unaligned (scalar-load) variant
          y = parts[1].detach().clone().requires_grad_(True)
          f_eval = func(-s, y)
          grads = torch.autograd.grad(f_eval, (y,) + params, -parts[2], allow_unused=True)
        outs = [torch.zeros_like(parts[0]), f_eval.detach()]
        outs.append(torch.zeros_like(y) if grads[0] is None else grads[0])
        for p, g in zip(params, grads[1:]):
          outs.append(torch.zeros_like(p) if g is None else g)
        return -_flatten(outs)

      options = dict(adj['options'])
      # (a caller's own adjoint norm -- torchdiffeq's adjoint_options['norm'], e.g. 'seminorm' -- is honoured by the flat loop only: the
      # native controller's norm is the default mixed one)
      user_norm = options.get('norm') is not None
      if adj['method'] in ('dopri5', 'adaptive_heun') and 'norm' not in options:
        options['norm'] = _mixed_norm(shapes)
      fixed = adj['method'] in ('euler', 'rk4')
      if fixed and options.get('step_size') is None:
        raise ValueError('fixed-grid adjoint methods need adjoint_options["step_size"]')
      for i in range(len(t) - 1, 0, -1):
        span = -t[i - 1:i + 1].flip(0)
        if fixed and _adjoint_native_ok(func, state[1], adj['method']):
          # the whole interval as one native object (one hipGraph; no PyTorch op between the stages)
          state[2], contrib = _adjoint_native(func, params, state[1], state[2], span, adj['method'], options['step_size'])
          state = state[:3] + [gp if c is None else gp + c.reshape(gp.shape) for gp, c in zip(state[3:], contrib)]
        elif fixed:
          state[2], gp = _adjoint_fixed_grid(func, params, state[1], state[2], state[3:], span, adj['method'],
                                             options['step_size'])
          state = state[:3] + list(gp)
        elif _adjoint_adaptive_ok(func, state[1], adj['method']) and not options.get('host_flat', False) and not user_norm:
          # adaptive adjoint method on the Laplacian function: native stages on the components (no autograd graph, no flat vector)
          state[2], new = _adjoint_adaptive_native(func, params, state[1], state[2], state[3:], span, adj['method'], adj['rtol'], adj['atol'])
          for idx, val in new.items():
            state[3 + idx] = val
        else:
          flat = odeint(reversed_flat_dynamics, _flatten(state), span, rtol=adj['rtol'], atol=adj['atol'],
                        method=adj['method'], options=options)[1]
          state = [p.clone() for p in _unflatten(flat, shapes)]
        state[1] = ans[i - 1]
        state[2] = state[2] + grad_out[i - 1]
    return (None, state[2], None, None, None) + tuple(state[3:])


def odeint_adjoint(func, y0, t, *, rtol=1e-7, atol=1e-9, method=None, options=None, adjoint_rtol=None, adjoint_atol=None,
                   adjoint_method=None, adjoint_options=None, adjoint_params=None, use_graph=True):
  """Drop-in for torchdiffeq.odeint_adjoint: same values as `odeint`; gradients with respect to y0 and the
  function's parameters come from solving the adjoint ODE backwards (see _AdjointSolve), not from a tape."""
  if isinstance(y0, tuple):
    if adjoint_params is None:
      adjoint_params = tuple(func.parameters()) if isinstance(func, torch.nn.Module) else ()
    return _solve_tuple(odeint_adjoint, func, y0, t,
                        dict(rtol=rtol, atol=atol, method=method, options=options, adjoint_rtol=adjoint_rtol,
                             adjoint_atol=adjoint_atol, adjoint_method=adjoint_method, adjoint_options=adjoint_options,
                             adjoint_params=tuple(adjoint_params), use_graph=use_graph))
  options = dict(options or {})
  fwd = dict(rtol=rtol, atol=atol, method=method, options=options, use_graph=use_graph)
  if adjoint_params is None:
    adjoint_params = tuple(func.parameters()) if isinstance(func, torch.nn.Module) else ()
  params = tuple(p for p in adjoint_params if p.requires_grad)
  if not torch.is_grad_enabled() or not (y0.requires_grad or params):
    return odeint(func, y0, t, **fwd)
  adj = dict(rtol=rtol if adjoint_rtol is None else adjoint_rtol, atol=atol if adjoint_atol is None else adjoint_atol,
             method=(method or 'dopri5') if adjoint_method is None else adjoint_method,
             options={k: v for k, v in options.items() if k != 'norm'} if adjoint_options is None else dict(adjoint_options))
  return _AdjointSolve.apply(func, y0, t, fwd, adj, *params)
