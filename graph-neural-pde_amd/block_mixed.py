"""Block that integrates a Laplacian function with weights gamma' * rw + (1 - gamma') * mean attention,
computed once per forward pass (reference src/block_mixed.py:8-67)."""
import torch
from torch import nn

from .base_classes import ODEblock
from .function_transformer_attention import SpGraphTransAttentionLayer


class MixedODEblock(ODEblock):
  def __init__(self, odefunc, regularization_fns, opt, data, device, t=torch.tensor([0, 1]), gamma=0.):
    super(MixedODEblock, self).__init__(odefunc, regularization_fns, opt, data, device, t)
    self._second_function(odefunc, opt, data, device)
    self._rw_graph(data, opt, device)
    self._use_default_integrators(opt)
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
