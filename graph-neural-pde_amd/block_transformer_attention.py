"""Block that computes multi-head attention ONCE from x(0) and integrates a Laplacian function
with it (reference src/block_transformer_attention.py:7-72)."""
import torch

from .base_classes import ODEblock
from .function_transformer_attention import SpGraphTransAttentionLayer


class AttODEblock(ODEblock):
  def __init__(self, odefunc, regularization_fns, opt, data, device, t=torch.tensor([0, 1]), gamma=0.5):
    super(AttODEblock, self).__init__(odefunc, regularization_fns, opt, data, device, t)
    self._second_function(odefunc, opt, data, device)
    self._rw_graph(data, opt, device)
    self._use_default_integrators(opt)
    self.multihead_att_layer = SpGraphTransAttentionLayer(opt['hidden_dim'], opt['hidden_dim'], opt, device,
                                                          edge_weights=self.odefunc.edge_weight).to(device)

  def get_attention_weights(self, x):
    attention, values = self.multihead_att_layer(x, self.odefunc.edge_index)
    return attention

  def forward(self, x):
    self.odefunc.attention_weights = self.get_attention_weights(x)
    self.reg_odefunc.odefunc.attention_weights = self.odefunc.attention_weights
    return self._integrate(x, {'step_size': self.opt['step_size']})
