"""TEST INFRASTRUCTURE (checker only -- imported by tests/, never by the product).

What autograd does when the reference trains with opt['adjoint'] = False and method = 'dopri5' (reference src/base_classes.py:44-47,
src/block_transformer_attention.py:58-63): `loss.backward()` runs back through every ACCEPTED step of torchdiffeq 0.2.1's
Dopri5Solver (rk_common.py _runge_kutta_step / _adaptive_step, interp.py _interp_fit / _interp_evaluate); rejected trial steps
leave no trace in the result, and the step sizes are constants of the backward pass (misc.py _optimal_step_size runs under
torch.no_grad; the first step size is differentiable in torchdiffeq but changes the gradients by ~1e-6 relative at the
reference's tolerances, tests/test_tape_reverse_cpu.py).  Written out as an explicit reverse sweep over a recorded tape of stage
inputs, for the GRAND-l right-hand side f(u) = a (A u - u) + b x0, which is linear in u:

  step  u_0 = y,  u_i = y + h sum_{j<i} a_ij k_j (i = 1..5),  u_6 = y1 = y + h sum_j b_j k_j,  k_i = f(u_i),  k_0 = k_6 of the step before
  end   out = quartic through (y, y1, y_mid, k_0, k_6) at x = (T - t) / h of the last step

  reverse, per step, given (G_y1, G_k6):   W_6 = G_y1 + a (A^T G_k6 - G_k6)
                                            G_ki = h sum_{m>i} a_mi W_m (+ the interpolation's direct terms in the last step),  a_6j = b_j
                                            W_i  = a (A^T G_ki - G_ki)                      i = 5..1
                                            G_y  = sum_m W_m,  G_k0 -> the step before as its G_k6
  parameters, summed over every evaluation k_i = f(u_i):   r_e += G_ki[row] . u_i[col]   (d w_e = a r_e)
                                                           s_a += <G_ki, a (A u_i - u_i)> = <u_i, a (A^T G_ki - G_ki)>
                                                           s_b += <G_ki, x0>

This is the algebra csrc/dopri5.hip's gnpde_dopri5_tape_backward runs with native kernels; tests/test_tape_reverse_cpu.py holds it
against torch autograd through the restated solver in float64, tests/test_tape_gpu.py holds the native path against both."""
import torch

# Dormand-Prince 5(4), torchdiffeq dopri5.py (same numbers as graph-neural-pde_amd/odeint.py and oracle/shims)
A = ((1 / 5,),
     (3 / 40, 9 / 40),
     (44 / 45, -56 / 15, 32 / 9),
     (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
     (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656),
     (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84))
E = (35 / 384 - 1951 / 21600, 0.0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720, -2187 / 6784 + 12231 / 42400,
     11 / 84 - 649 / 6300, -1 / 60)
MID = (6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
       187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2)


def _rms(v):
  return v.pow(2).mean().sqrt()


@torch.no_grad()
def dopri5_record(f, y0, T, rtol, atol, safety=0.9, ifactor=10.0, dfactor=0.2):
  """torchdiffeq 0.2.1 dopri5 from 0 to T (controller in float64, state in y0's dtype) that RECORDS, per accepted step, the stage
  inputs u_0..u_6 and the step size; returns (out, tape) with tape = {'steps': [{'u': [u0..u6], 'h': h}], 'x': fraction of the
  last step at which T lies, 'nfe': evaluations}."""
  dt_ = y0.dtype
  nfe = [0]

  def F(u):
    nfe[0] += 1
    return f(u)
  f0 = F(y0)
  scale = atol + y0.abs() * rtol
  d0, d1 = _rms(y0 / scale), _rms(f0 / scale)
  h0 = torch.tensor(1e-6, dtype=dt_) if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
  f1 = F(y0 + h0 * f0)
  d2 = _rms((f1 - f0) / scale) / h0
  h1 = torch.max(torch.tensor(1e-6, dtype=dt_), h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1.0 / 5)
  dt = torch.min(100 * h0, h1).to(torch.float64)
  t = torch.zeros((), dtype=torch.float64)
  T = torch.as_tensor(T, dtype=torch.float64)
  y, k0 = y0, f0
  steps = []
  last = None
  while T > t:
    h = dt.to(dt_)
    ks, us = [k0], [y]
    for row in A:
      u = y
      acc = None
      for kj, c in zip(ks, row):
        if c != 0.0:
          term = kj * (torch.tensor(c, dtype=dt_) * h)
          acc = term if acc is None else acc + term
      u = y + acc
      us.append(u)
      ks.append(F(u))
    y1 = us[6]
    err = None
    for kj, c in zip(ks, E):
      if c != 0.0:
        term = kj * (torch.tensor(c, dtype=dt_) * h)
        err = term if err is None else err + term
    ratio = _rms(err / (atol + rtol * torch.max(y.abs(), y1.abs())))
    if ratio <= 1:
      steps.append({'u': us, 'h': h.clone()})
      last = (y, y1, ks, h, t.clone(), t + dt)
      t = t + dt
      y, k0 = y1, ks[6]
    if ratio == 0:
      dt = dt * ifactor
    else:
      lo = 1.0 if ratio < 1 else dfactor
      dt = dt * torch.clamp(safety / ratio.to(torch.float64) ** 0.2, min=lo, max=ifactor)
  ya, yb, ks, h, ta, tb = last
  ym = ya
  acc = None
  for kj, c in zip(ks, MID):
    if c != 0.0:
      term = kj * (torch.tensor(c, dtype=dt_) * h)
      acc = term if acc is None else acc + term
  ym = ya + acc
  x = ((T - ta) / (tb - ta)).to(dt_)
  fa, fb = ks[0], ks[6]
  ca = 2 * h * (fb - fa) - 8 * (yb + ya) + 16 * ym
  cb = h * (5 * fa - 3 * fb) + 18 * ya + 14 * yb - 32 * ym
  cc = h * (fb - 4 * fa) - 11 * ya - 5 * yb + 16 * ym
  cd = h * fa
  out = ya + x * cd + x ** 2 * cc + x ** 3 * cb + x ** 4 * ca
  return out, {'steps': steps, 'x': x, 'nfe': nfe[0]}


@torch.no_grad()
def dopri5_replay(f, y0, hs, x):
  """The accepted steps of a dopri5 solve replayed with GIVEN step sizes hs (no controller) and end-point fraction x of the last step:
  (out, tape) as dopri5_record returns them.  Lets a float64 check follow the accept / reject decisions a float32 solve took."""
  dt_ = y0.dtype
  y, k0 = y0, f(y0)
  steps = []
  ks = None
  for h in hs:
    h = torch.as_tensor(h, dtype=dt_)
    ks, us = [k0], [y]
    for row in A:
      acc = None
      for kj, c in zip(ks, row):
        if c != 0.0:
          term = kj * (torch.tensor(c, dtype=dt_) * h)
          acc = term if acc is None else acc + term
      us.append(y + acc)
      ks.append(f(us[-1]))
    steps.append({'u': us, 'h': h})
    ya, y, k0 = y, us[6], ks[6]
  yb, h = y, torch.as_tensor(hs[-1], dtype=dt_)
  acc = None
  for kj, c in zip(ks, MID):
    if c != 0.0:
      term = kj * (torch.tensor(c, dtype=dt_) * h)
      acc = term if acc is None else acc + term
  ym = ya + acc
  x = torch.as_tensor(x, dtype=dt_)
  fa, fb = ks[0], ks[6]
  ca = 2 * h * (fb - fa) - 8 * (yb + ya) + 16 * ym
  cb = h * (5 * fa - 3 * fb) + 18 * ya + 14 * yb - 32 * ym
  cc = h * (fb - 4 * fa) - 11 * ya - 5 * yb + 16 * ym
  cd = h * fa
  out = ya + x * cd + x ** 2 * cc + x ** 3 * cb + x ** 4 * ca
  return out, {'steps': steps, 'x': x, 'nfe': 1 + 6 * len(steps)}


def interp_weights(x, h):
  """Partial derivatives of the quartic end-point interpolation (torchdiffeq interp.py) with respect to what it is built from:
  out = p_y y + p_y1 y1 + p_ym y_mid + p_k0 k_0 + p_k6 k_6  (every operand enters linearly)."""
  x2, x3, x4 = x * x, x * x * x, x * x * x * x
  p_ym = 16 * x2 - 32 * x3 + 16 * x4
  p_y = 1 - 11 * x2 + 18 * x3 - 8 * x4
  p_y1 = -5 * x2 + 14 * x3 - 8 * x4
  p_k0 = h * (x - 4 * x2 + 5 * x3 - 2 * x4)
  p_k6 = h * (x2 - 3 * x3 + 2 * x4)
  return p_y, p_y1, p_ym, p_k0, p_k6


@torch.no_grad()
def dopri5_tape_reverse(tape, gout, vjp_u, on_eval):
  """Reverse sweep.  vjp_u(g) = (df/du)^T g = a (A^T g - g); on_eval(g, u, w) is called once per evaluation k = f(u) with the
  gradient g that reaches k and w = vjp_u(g) (the caller accumulates its parameter gradients from them).  Returns dL/dy0."""
  steps, x = tape['steps'], tape['x']
  S = len(steps)
  zero = torch.zeros_like(gout)
  Gy1, Gk6 = None, None
  for s in range(S - 1, -1, -1):
    u, h = steps[s]['u'], steps[s]['h']
    D = [zero] * 7      # direct terms on k_0..k_6 (last step: the interpolation)
    Dy = zero
    if s == S - 1:
      p_y, p_y1, p_ym, p_k0, p_k6 = interp_weights(x, h)
      Dy = (p_y + p_ym) * gout
      Gy1 = p_y1 * gout
      D = [(p_ym * h * MID[j]) * gout for j in range(7)]
      D[0] = D[0] + p_k0 * gout
      Gk6 = D[6] + p_k6 * gout
    W = [None] * 7
    w6 = vjp_u(Gk6)
    on_eval(Gk6, u[6], w6)
    W[6] = Gy1 + w6
    for i in range(5, 0, -1):
      g = D[i]
      for m in range(i + 1, 7):
        c = A[m - 1][i] if i < len(A[m - 1]) else 0.0
        if c != 0.0:
          g = g + (torch.tensor(c, dtype=gout.dtype) * h) * W[m]
      W[i] = vjp_u(g)
      on_eval(g, u[i], W[i])
    g0 = D[0]
    for m in range(1, 7):
      c = A[m - 1][0]
      if c != 0.0:
        g0 = g0 + (torch.tensor(c, dtype=gout.dtype) * h) * W[m]
    Gy = Dy
    for m in range(1, 7):
      Gy = Gy + W[m]
    Gy1, Gk6 = Gy, g0            # the step before: its y1 is this y, its k_6 is this k_0
  w0 = vjp_u(Gk6)
  on_eval(Gk6, steps[0]['u'][0], w0)
  return Gy1 + w0
