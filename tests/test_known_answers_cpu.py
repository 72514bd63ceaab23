"""Known-answer tests of the solvers that do NOT rest on any restatement of torchdiffeq by this repository.

torchdiffeq 0.2.1 is not in /root/reference and cannot be installed offline, so `odeint(method='dopri5')`, `adaptive_heun`
and `odeint_adjoint` were pinned in round 1 only by agreement between two restatements written here.  These tests tie them
to facts established elsewhere:

* closed forms: y' = A y  =>  y(T) = expm(A T) y0, and for L = c . y(T): dL/dy0 = expm(A T)^T c, dL/dA by autograd through
  torch.matrix_exp (float64);
* an INDEPENDENT implementation of the same published method: SciPy's RK45 is Dormand-Prince 5(4) with the controller
  constants torchdiffeq documents (safety 0.9, step factor in [0.2, 10], rms error norm against atol + rtol max(|y0|,|y1|),
  Hairer's initial-step rule with order 4).  On problems where no trial step is rejected the two must take the SAME
  accepted steps (SciPy only differs after a rejection, where it caps the next growth at 1, and at the end point, which
  it clamps to while torchdiffeq steps past it and interpolates);
* classical orders of convergence (euler 1, rk4 3/8-rule 4, adaptive Heun-Euler within its tolerance).
"""
import importlib
import math

import numpy as np
import pytest
import torch
from scipy.integrate import solve_ivp

O = importlib.import_module('gnpde_amd.odeint')
F64 = torch.float64


def _stable_matrix(n, seed):
  g = torch.Generator().manual_seed(seed)
  m = torch.randn(n, n, generator=g, dtype=F64) * 0.4
  return m - m.t() - 0.3 * torch.eye(n, dtype=F64) + 0.1 * torch.randn(n, n, generator=g, dtype=F64)


class Linear(torch.nn.Module):
  def __init__(self, A):
    super(Linear, self).__init__()
    self.A = torch.nn.Parameter(A.clone())
    self.calls = 0

  def forward(self, t, y):
    self.calls += 1
    return y @ self.A.t()


# --------------------------------------------------------------------------------------------------
# closed forms
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('method,opts,tol', [
  ('dopri5', {}, 2e-9),
  ('adaptive_heun', {}, 5e-6),
  ('rk4', {'step_size': 0.02}, 1e-8),
])
def test_linear_system_against_matrix_exponential(method, opts, tol):
  A = _stable_matrix(6, 0)
  y0 = torch.randn(3, 6, generator=torch.Generator().manual_seed(1), dtype=F64)
  T = 2.5
  f = Linear(A)
  with torch.no_grad():
    out = O.odeint(f, y0, torch.tensor([0.0, T], dtype=F64), method=method, options=opts, rtol=1e-10, atol=1e-12)
  exact = y0 @ torch.matrix_exp(A * T).t()
  err = float((out[1] - exact).abs().max() / exact.abs().max())
  assert err < tol, (method, err)
  assert torch.equal(out[0], y0)


def test_fixed_step_orders_of_convergence():
  A = _stable_matrix(5, 2)
  y0 = torch.randn(5, generator=torch.Generator().manual_seed(3), dtype=F64)
  T = 1.0
  exact = torch.matrix_exp(A * T) @ y0

  def err(method, h):
    with torch.no_grad():
      out = O.odeint(Linear(A), y0, torch.tensor([0.0, T], dtype=F64), method=method, options={'step_size': h})
    return float((out[1] - exact).norm())

  r_euler = err('euler', 0.01) / err('euler', 0.005)
  r_rk4 = err('rk4', 0.1) / err('rk4', 0.05)
  assert 1.9 < r_euler < 2.1, r_euler          # first order
  assert 14.0 < r_rk4 < 18.0, r_rk4            # fourth order (3/8 rule)


def test_fixed_grid_short_last_step_lands_on_T():
  """T not a multiple of the step: torchdiffeq's grid replaces the last point by T (C2: T = 18.2948 -> 19 steps)."""
  y0 = torch.tensor([1.0], dtype=F64)
  T = 1.23
  with torch.no_grad():
    out = O.odeint(lambda t, y: -y, y0, torch.tensor([0.0, T], dtype=F64), method='rk4', options={'step_size': 0.5})
  # three steps 0.5, 0.5, 0.23 of the 3/8 rule on y' = -y: amplification factor R(h) = 1 - h + h^2/2 - h^3/6 + h^4/24
  R = lambda h: 1 - h + h ** 2 / 2 - h ** 3 / 6 + h ** 4 / 24   # noqa: E731
  assert abs(float(out[1]) - R(0.5) * R(0.5) * R(0.23)) < 1e-14


# --------------------------------------------------------------------------------------------------
# dopri5 controller against SciPy's RK45 (independent implementation of Dormand-Prince 5(4))
# --------------------------------------------------------------------------------------------------
def _our_accepted_times(rhs, y0, T, rtol, atol):
  acc, rej = [], []
  calls = [0]

  def f(t, y):
    calls[0] += 1
    return rhs(t, y)

  with torch.no_grad():
    out = O._solve_dopri5(f, y0, torch.tensor([0.0, T], dtype=F64), rtol, atol,
                          on_accept=lambda y, t: acc.append(t), on_reject=lambda y, t: rej.append(t))
  return out[1], acc, rej, calls[0]


@pytest.mark.parametrize('name', ['decay', 'linear6', 'oscillator'])
@pytest.mark.parametrize('rtol,atol', [(1e-3, 1e-6), (1e-6, 1e-9), (1e-9, 1e-12)])
def test_dopri5_accepted_steps_match_scipy_rk45(name, rtol, atol):
  if name == 'decay':
    A = torch.tensor([[-1.0]], dtype=F64)
    y0 = torch.tensor([1.0], dtype=F64)
    T = 5.0
  elif name == 'linear6':
    A = _stable_matrix(6, 4)
    y0 = torch.randn(6, generator=torch.Generator().manual_seed(5), dtype=F64)
    T = 3.0
  else:
    A = torch.tensor([[0.0, 1.0], [-4.0, -0.1]], dtype=F64)
    y0 = torch.tensor([1.0, 0.0], dtype=F64)
    T = 6.0
  An = A.numpy()
  y1, acc, rej, calls = _our_accepted_times(lambda t, y: A @ y, y0, T, rtol, atol)
  sol = solve_ivp(lambda t, y: An @ y, (0.0, T), y0.numpy(), method='RK45', rtol=rtol, atol=atol)
  assert sol.success
  assert len(rej) == 0, 'pick problems without rejected steps: SciPy limits growth after a rejection, torchdiffeq does not'
  st = sol.t[1:]
  # same number of steps; all accepted times but the last agree (SciPy clamps the last step to T, torchdiffeq steps past
  # T and interpolates back)
  assert len(acc) == len(st), (len(acc), len(st))
  np.testing.assert_allclose(np.array(acc[:-1]), st[:-1], rtol=1e-9, atol=0)
  assert acc[-1] >= T
  # evaluations: f(y0) + one for the initial step + 6 per trial step (first-same-as-last)
  assert calls == 2 + 6 * len(acc) == sol.nfev
  exact = torch.matrix_exp(A * T) @ y0
  assert float((y1 - exact).abs().max()) < 50 * (atol + rtol * float(exact.abs().max())) * max(len(acc), 1) ** 0.5
  np.testing.assert_allclose(y1.numpy(), sol.y[:, -1], rtol=0, atol=20 * (atol + rtol))


def test_dopri5_rejections_shrink_by_the_documented_factor():
  """A kink forces rejections; every rejected trial is followed by a step no longer than 0.9 * h (safety) and no shorter
  than 0.2 * h (dfactor), and accepted steps never grow by more than 10x (ifactor)."""
  events = []

  def rhs(t, y):
    return torch.where(torch.as_tensor(t) < 1.0, -y, -50.0 * y + 3.0)

  with torch.no_grad():
    O._solve_dopri5(rhs, torch.tensor([1.0], dtype=F64), torch.tensor([0.0, 3.0], dtype=F64), 1e-6, 1e-9,
                    on_accept=lambda y, t: events.append(('a', t)), on_reject=lambda y, t: events.append(('r', t)))
  assert sum(1 for k, _ in events if k == 'r') >= 1
  t_prev, h_prev = 0.0, None
  for kind, t in events:
    if kind == 'a':
      h = t - t_prev
      if h_prev is not None:
        assert h <= 10.0 * h_prev * (1 + 1e-12)
      t_prev, h_prev = t, h
  assert t_prev >= 3.0


# --------------------------------------------------------------------------------------------------
# adjoint gradients against closed forms
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('method,adj,opts,tol', [
  ('dopri5', 'dopri5', {}, 2e-7),
  ('rk4', 'rk4', {'step_size': 0.02}, 1e-7),
  ('dopri5', 'rk4', {'step_size': 0.02}, 1e-7),
  ('euler', 'euler', {'step_size': 0.001}, 5e-3),
  ('dopri5', 'adaptive_heun', {}, 1e-4),
])
def test_adjoint_gradients_against_matrix_exponential(method, adj, opts, tol):
  A0 = _stable_matrix(5, 6)
  y0 = torch.randn(4, 5, generator=torch.Generator().manual_seed(7), dtype=F64)
  c = torch.randn(4, 5, generator=torch.Generator().manual_seed(8), dtype=F64)
  T = 1.5
  # closed form by autograd through the matrix exponential
  Ae = A0.clone().requires_grad_(True)
  ye = y0.clone().requires_grad_(True)
  (c * (ye @ torch.matrix_exp(Ae * T).t())).sum().backward()
  f = Linear(A0)
  yy = y0.clone().requires_grad_(True)
  out = O.odeint_adjoint(f, yy, torch.tensor([0.0, T], dtype=F64), method=method, options=opts, rtol=1e-10, atol=1e-12,
                         adjoint_method=adj, adjoint_options=opts, adjoint_rtol=1e-10, adjoint_atol=1e-12)
  (c * out[1]).sum().backward()
  e_y = float((yy.grad - ye.grad).abs().max() / ye.grad.abs().max())
  e_a = float((f.A.grad - Ae.grad).abs().max() / Ae.grad.abs().max())
  assert e_y < tol and e_a < tol, (method, adj, e_y, e_a)


def test_adjoint_with_intermediate_output_times():
  """Gradients arriving at several output times (torchdiffeq adds grad_y[i] when the backward pass crosses t[i])."""
  A0 = _stable_matrix(4, 9)
  y0 = torch.randn(4, generator=torch.Generator().manual_seed(10), dtype=F64)
  ts = torch.tensor([0.0, 0.4, 1.1, 2.0], dtype=F64)
  cs = torch.randn(4, 4, generator=torch.Generator().manual_seed(11), dtype=F64)
  Ae = A0.clone().requires_grad_(True)
  ye = y0.clone().requires_grad_(True)
  sum((cs[i] * (torch.matrix_exp(Ae * float(ts[i])) @ ye)).sum() for i in range(4)).backward()
  f = Linear(A0)
  yy = y0.clone().requires_grad_(True)
  out = O.odeint_adjoint(f, yy, ts, method='dopri5', rtol=1e-10, atol=1e-12, adjoint_method='dopri5',
                         adjoint_rtol=1e-10, adjoint_atol=1e-12)
  (cs * out).sum().backward()
  assert float((yy.grad - ye.grad).abs().max() / ye.grad.abs().max()) < 5e-7
  assert float((f.A.grad - Ae.grad).abs().max() / Ae.grad.abs().max()) < 5e-7


def test_float32_adjoint_gradient_error_is_rounding_not_method():
  """The fp32 gradient of a 40-step rk4 solve differs from the float64 closed form by a few 1e-6 relative: the tolerance
  the GPU gradient tests use (tests/test_autograd_gpu.py, GTOL) has to cover THIS rounding plus the device's different
  reduction orders, not a methodological error."""
  A0 = _stable_matrix(6, 12)
  y0 = torch.randn(32, 6, generator=torch.Generator().manual_seed(13), dtype=F64)
  c = torch.randn(32, 6, generator=torch.Generator().manual_seed(14), dtype=F64)
  T, h = 4.0, 0.1
  res = {}
  for dt in (F64, torch.float32):
    f = Linear(A0.to(dt))
    yy = y0.to(dt).requires_grad_(True)
    out = O.odeint_adjoint(f, yy, torch.tensor([0.0, T], dtype=dt), method='rk4', options={'step_size': h},
                           adjoint_method='rk4', adjoint_options={'step_size': h})
    (c.to(dt) * out[1]).sum().backward()
    res[dt] = (yy.grad.double(), f.A.grad.double())
  e_y = float((res[torch.float32][0] - res[F64][0]).abs().max() / res[F64][0].abs().max())
  e_a = float((res[torch.float32][1] - res[F64][1]).abs().max() / res[F64][1].abs().max())
  assert e_y < 2e-5 and e_a < 2e-5, (e_y, e_a)
  assert max(e_y, e_a) > 1e-8, 'fp32 cannot be exact'
