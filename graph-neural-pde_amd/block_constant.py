"""Block with a fixed, normalised adjacency (reference src/block_constant.py:6-70): builds the
rw / sym-normalised weights once, hands them to both function instances, and integrates with the
native fixed-step solver (euler / rk4) or the host dopri5 loop."""
import torch

from .base_classes import ODEblock
from .odeint import odeint, odeint_adjoint
from .utils import get_rw_adj, gcn_norm_fill_val


class ConstantODEblock(ODEblock):
  def __init__(self, odefunc, regularization_fns, opt, data, device, t=torch.tensor([0, 1])):
    super(ConstantODEblock, self).__init__(odefunc, regularization_fns, opt, data, device, t)
    # the reference builds a second function object here and leaves the first inside reg_odefunc
    self.odefunc = odefunc(self.aug_dim * opt['hidden_dim'], self.aug_dim * opt['hidden_dim'], opt, data, device)
    normalise = get_rw_adj if opt['data_norm'] == 'rw' else gcn_norm_fill_val
    kw = dict(norm_dim=1) if opt['data_norm'] == 'rw' else {}
    edge_index, edge_weight = normalise(data.edge_index, edge_weight=data.edge_attr, fill_value=opt['self_loop_weight'],
                                        num_nodes=data.num_nodes, dtype=data.x.dtype, **kw)
    self.odefunc.edge_index = edge_index.to(device)
    self.odefunc.edge_weight = edge_weight.to(device)
    self.reg_odefunc.odefunc.edge_index, self.reg_odefunc.odefunc.edge_weight = self.odefunc.edge_index, self.odefunc.edge_weight
    self.train_integrator = odeint_adjoint if opt['adjoint'] else odeint
    self.test_integrator = odeint
    self.set_tol()

  def forward(self, x):
    return self._integrate(x, dict(step_size=self.opt['step_size'], max_iters=self.opt['max_iters']))
