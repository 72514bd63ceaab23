"""Operator surface kept from the reference's src/base_classes.py so that GNN.py / run_GNN.py can
use these classes unchanged: ODEFunc (:77-95) and ODEblock (:32-74) with the same constructor
signatures, attributes written from outside (edge_index, edge_weight, attention_weights, x0, nfe),
parameter names (alpha_train, beta_train, alpha_sc, beta_sc) and the two-function layout
(`odefunc` + `reg_odefunc.odefunc`) that the reference's state_dicts contain."""
import torch
from torch import nn

from . import _lib
from .graph import graph_of
from .utils import MaxNFEException
from .odeint import odeint, odeint_adjoint


class RegularizedODEfunc(nn.Module):
  """Holder matching reference src/regularized_ODE_function.py:8-33 (state_dict prefix
  `reg_odefunc.odefunc.`).  The autograd-based regularisers themselves are training-only and outside
  this hot path: with an empty list this is a pass-through."""

  def __init__(self, odefunc, regularization_fns):
    super(RegularizedODEfunc, self).__init__()
    self.odefunc = odefunc
    self.regularization_fns = regularization_fns

  def forward(self, t, state):
    if len(self.regularization_fns) > 0:
      raise NotImplementedError('kinetic / Jacobian regularisers are not part of the MI355X hot path')
    x = state[0] if isinstance(state, tuple) else state
    return self.odefunc(t, x)


class ODEFunc(nn.Module):
  """Base of the three right-hand sides.  Subclasses implement `_descriptor(x)` (the native
  gnpde_rhs_t for the current attributes); `forward(t, x)` = nfe guard + one native evaluation."""

  def __init__(self, opt, data, device):
    super(ODEFunc, self).__init__()
    self.opt = opt
    self.device = device
    self.edge_index = None
    self.edge_weight = None
    self.attention_weights = None
    self.alpha_train = nn.Parameter(torch.tensor(0.0))
    self.beta_train = nn.Parameter(torch.tensor(0.0))
    self.x0 = None
    self.nfe = 0
    self.alpha_sc = nn.Parameter(torch.ones(1))
    self.beta_sc = nn.Parameter(torch.ones(1))
    self._cache = {}

  # ---- shared plumbing -------------------------------------------------------------------------
  def _graph(self, x):
    if self.edge_index is None:
      raise _lib.GnpdeError('%s.edge_index has not been set' % self.__class__.__name__)
    ei = self.edge_index
    if ei.device != x.device:
      raise _lib.GnpdeError('edge_index is on %s but the state is on %s' % (ei.device, x.device))
    return graph_of(ei, x.shape[0], x.device)

  def _check_nfe(self):
    if self.nfe > self.opt["max_nfe"]:
      raise MaxNFEException
    self.nfe += 1

  def _needs_grad(self, x):
    if not torch.is_grad_enabled():
      return False
    return x.requires_grad or any(p.requires_grad for p in self.parameters())

  def _source(self, x):
    """x0 term of the epilogue (opt['add_source'])."""
    if not self.opt['add_source']:
      return None
    if self.x0 is None:
      raise _lib.GnpdeError('add_source is set but x0 was never assigned (call ODEblock.set_x0)')
    return self.x0

  def _memo(self, key, tensors, make):
    """Cache derived device buffers on (identity, version) of their source tensors."""
    sig = tuple((id(t), t._version) if t is not None else None for t in tensors)
    hit = self._cache.get(key)
    if hit is not None and hit[0] == sig:
      return hit[2]
    val = make()
    self._cache[key] = (sig, tensors, val)  # `tensors` kept alive so that ids stay unique
    return val

  def forward(self, t, x):
    from . import ops
    self._check_nfe()
    if self._needs_grad(x):
      from .autograd import rhs_with_grad
      return rhs_with_grad(self, x)
    with torch.no_grad():
      x = _lib.f32c(x)   # descriptor (leading dimension) and kernels must see the same, contiguous, operand
      return ops.rhs_eval(self._descriptor(x), x)

  def __repr__(self):
    return self.__class__.__name__


class ODEblock(nn.Module):
  def __init__(self, odefunc, regularization_fns, opt, data, device, t):
    super(ODEblock, self).__init__()
    self.opt = opt
    self.t = t
    self.device = device
    self.aug_dim = 2 if opt['augment'] else 1
    self.odefunc = odefunc(self.aug_dim * opt['hidden_dim'], self.aug_dim * opt['hidden_dim'], opt, data, device)
    self.nreg = len(regularization_fns)
    self.reg_odefunc = RegularizedODEfunc(self.odefunc, regularization_fns)
    self.train_integrator = odeint_adjoint if opt['adjoint'] else odeint
    self.test_integrator = None
    self.set_tol()

  def set_x0(self, x0):
    self.odefunc.x0 = x0.clone().detach()
    self.reg_odefunc.odefunc.x0 = x0.clone().detach()

  def set_tol(self):
    self.atol = self.opt['tol_scale'] * 1e-7
    self.rtol = self.opt['tol_scale'] * 1e-9
    if self.opt['adjoint']:
      self.atol_adjoint = self.opt['tol_scale_adjoint'] * 1e-7
      self.rtol_adjoint = self.opt['tol_scale_adjoint'] * 1e-9

  def reset_tol(self):
    self.atol = 1e-7
    self.rtol = 1e-9
    self.atol_adjoint = 1e-7
    self.rtol_adjoint = 1e-9

  def set_time(self, time):
    self.t = torch.tensor([0, time]).to(self.device)

  def _second_function(self, odefunc, opt, data, device):
    """The reference's blocks build a SECOND function object and leave the first inside reg_odefunc
    (state_dicts therefore hold both); same here."""
    width = self.aug_dim * opt['hidden_dim']
    self.odefunc = odefunc(width, width, opt, data, device)

  def _share_graph(self, edge_index, edge_weight, device):
    """Hand the block's normalised adjacency to both function objects."""
    self.odefunc.edge_index = edge_index.to(device)
    self.odefunc.edge_weight = edge_weight.to(device)
    inner = self.reg_odefunc.odefunc
    inner.edge_index, inner.edge_weight = self.odefunc.edge_index, self.odefunc.edge_weight

  def _rw_graph(self, data, opt, device):
    from .utils import get_rw_adj
    ei, ew = get_rw_adj(data.edge_index, edge_weight=data.edge_attr, norm_dim=1, fill_value=opt['self_loop_weight'],
                        num_nodes=data.num_nodes, dtype=data.x.dtype)
    self._share_graph(ei, ew, device)
    return self.odefunc.edge_index, self.odefunc.edge_weight

  def _use_default_integrators(self, opt):
    self.train_integrator = odeint_adjoint if opt['adjoint'] else odeint
    self.test_integrator = odeint
    self.set_tol()

  def _integrate(self, x, options):
    """Common tail of the block forwards (reference src/block_constant.py:35-70)."""
    t = self.t.type_as(x)
    integrator = self.train_integrator if self.training else self.test_integrator
    if self.training and self.nreg > 0:
      raise NotImplementedError('regularised training states are not part of the MI355X hot path')
    kw = dict(method=self.opt['method'], options=options, atol=self.atol, rtol=self.rtol)
    if self.opt['adjoint'] and self.training:
      kw.update(adjoint_method=self.opt['adjoint_method'],
                adjoint_options=dict(step_size=self.opt['adjoint_step_size']),
                adjoint_atol=self.atol_adjoint, adjoint_rtol=self.rtol_adjoint)
    state_dt = integrator(self.odefunc, x, t, **kw)
    return state_dt[1]

  def __repr__(self):
    return self.__class__.__name__ + '( Time Interval ' + str(self.t[0].item()) + ' -> ' + str(self.t[1].item()) + ")"
