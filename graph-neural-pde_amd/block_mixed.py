"""Block that integrates a Laplacian function with weights gamma' * rw + (1 - gamma') * mean attention,
computed once per forward pass (reference src/block_mixed.py:8-67)."""
import torch
from torch import nn

from .base_classes import ODEblock
from .function_transformer_attention import SpGraphTransAttentionLayer
from .odeint import odeint, odeint_adjoint
from .utils import get_rw_adj


class MixedODEblock(ODEblock):
  def __init__(self, odefunc, regularization_fns, opt, data, device, t=torch.tensor([0, 1]), gamma=0.):
    super(MixedODEblock, self).__init__(odefunc, regularization_fns, opt, data, device, t)
    self.odefunc = odefunc(self.aug_dim * opt['hidden_dim'], self.aug_dim * opt['hidden_dim'], opt, data, device)
    edge_index, edge_weight = get_rw_adj(data.edge_index, edge_weight=data.edge_attr, norm_dim=1,
                                         fill_value=opt['self_loop_weight'], num_nodes=data.num_nodes,
                                         dtype=data.x.dtype)
    self.odefunc.edge_index = edge_index.to(device)
    self.odefunc.edge_weight = edge_weight.to(device)
    self.reg_odefunc.odefunc.edge_index, self.reg_odefunc.odefunc.edge_weight = self.odefunc.edge_index, self.odefunc.edge_weight
    self.train_integrator = odeint_adjoint if opt['adjoint'] else odeint
    self.test_integrator = odeint
    self.set_tol()
    # parameter trading off between attention and the Laplacian
    self.gamma = nn.Parameter(gamma * torch.ones(1))
    self.multihead_att_layer = SpGraphTransAttentionLayer(opt['hidden_dim'], opt['hidden_dim'], opt, device).to(device)

  def get_attention_weights(self, x):
    attention, values = self.multihead_att_layer(x, self.odefunc.edge_index)
    return attention

  def get_mixed_attention(self, x):
    gamma = torch.sigmoid(self.gamma)
    attention = self.get_attention_weights(x)
    return attention.mean(dim=1) * (1 - gamma) + self.odefunc.edge_weight * gamma

  def forward(self, x):
    self.odefunc.attention_weights = self.get_mixed_attention(x)
    return self._integrate(x, {'step_size': self.opt['step_size']})
