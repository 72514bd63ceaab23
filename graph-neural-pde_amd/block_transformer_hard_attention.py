"""Block that keeps only the strongest att_samp_pct of the edges while training and integrates a
function over the sampled, renormalised attention (reference src/block_transformer_hard_attention.py:7-103;
`block` of the ogbn-arxiv / Computers / Photo best_params).  The edge selection (quantile, mask, segment
sum) is host-side bookkeeping done once per forward with device tensor ops; attention and f stay native.
The function's graph is rebuilt lazily whenever `edge_index` is swapped."""
import torch

from .base_classes import ODEblock
from .function_transformer_attention import SpGraphTransAttentionLayer


class HardAttODEblock(ODEblock):
  def __init__(self, odefunc, regularization_fns, opt, data, device, t=torch.tensor([0, 1]), gamma=0.5):
    super(HardAttODEblock, self).__init__(odefunc, regularization_fns, opt, data, device, t)
    assert opt['att_samp_pct'] > 0 and opt['att_samp_pct'] <= 1, "attention sampling threshold must be in (0,1]"
    self.opt = opt
    self._second_function(odefunc, opt, data, device)
    self.num_nodes = data.num_nodes
    self.data_edge_index, _ = self._rw_graph(data, opt, device)   # odefunc.edge_index is swapped while training
    self._use_default_integrators(opt)
    if opt['function'] not in {'GAT', 'transformer'}:
      self.multihead_att_layer = SpGraphTransAttentionLayer(opt['hidden_dim'], opt['hidden_dim'], opt, device,
                                                            edge_weights=self.odefunc.edge_weight).to(device)

  def get_attention_weights(self, x):
    if self.opt['function'] not in {'GAT', 'transformer'}:
      attention, values = self.multihead_att_layer(x, self.data_edge_index)
    else:
      attention, values = self.odefunc.multihead_att_layer(x, self.data_edge_index)
    return attention

  def renormalise_attention(self, attention):
    index = self.odefunc.edge_index[self.opt['attention_norm_idx']]
    sums = torch.zeros(self.num_nodes, dtype=attention.dtype, device=attention.device).index_add_(0, index, attention)
    return attention / (sums[index] + 1e-16)

  def forward(self, x):
    attention_weights = self.get_attention_weights(x)
    if self.training:
      with torch.no_grad():
        mean_att = attention_weights.mean(dim=1, keepdim=False)
        if self.opt['use_flux']:
          delta = torch.linalg.norm(x[self.data_edge_index[0, :], :] - x[self.data_edge_index[1, :], :], dim=1)
          mean_att = mean_att * delta
        threshold = torch.quantile(mean_att, 1 - self.opt['att_samp_pct'])
        mask = mean_att > threshold
        self.odefunc.edge_index = self.data_edge_index[:, mask]
        self.odefunc.attention_weights = self.renormalise_attention(mean_att[mask])
    else:
      self.odefunc.edge_index = self.data_edge_index
      self.odefunc.attention_weights = attention_weights.mean(dim=1, keepdim=False)
    self.reg_odefunc.odefunc.edge_index, self.reg_odefunc.odefunc.edge_weight = self.odefunc.edge_index, self.odefunc.edge_weight
    self.reg_odefunc.odefunc.attention_weights = self.odefunc.attention_weights
    return self._integrate(x, {'step_size': self.opt['step_size']})
