"""Block with a fixed, normalised adjacency (reference src/block_constant.py:6-70): builds the
rw / sym-normalised weights once, hands them to both function instances, and integrates with the
native fixed-step solver (euler / rk4) or the host dopri5 loop."""
import torch

from .base_classes import ODEblock
from .utils import gcn_norm_fill_val


class ConstantODEblock(ODEblock):
  def __init__(self, odefunc, regularization_fns, opt, data, device, t=torch.tensor([0, 1])):
    super(ConstantODEblock, self).__init__(odefunc, regularization_fns, opt, data, device, t)
    self._second_function(odefunc, opt, data, device)
    if opt['data_norm'] == 'rw':
      self._rw_graph(data, opt, device)
    else:
      ei, ew = gcn_norm_fill_val(data.edge_index, edge_weight=data.edge_attr, fill_value=opt['self_loop_weight'],
                                 num_nodes=data.num_nodes, dtype=data.x.dtype)
      self._share_graph(ei, ew, device)
    self._use_default_integrators(opt)

  def forward(self, x):
    return self._integrate(x, dict(step_size=self.opt['step_size'], max_iters=self.opt['max_iters']))
