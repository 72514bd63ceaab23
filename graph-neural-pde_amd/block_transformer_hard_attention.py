"""Hard attention: while training, only the strongest `att_samp_pct` share of the edges (ranked by head-mean
attention, optionally times the feature distance across the edge) carries the diffusion, with the kept attention
renormalised per node; in evaluation mode every edge is used (reference
src/block_transformer_hard_attention.py:7-103; `block` of the ogbn-arxiv / Computers / Photo best_params).

The selection (quantile by radix select, stable compaction, per-node renormalisation: csrc/rewire.hip) and the attention
itself and every evaluation of f run on the native kernels.  Swapping `odefunc.edge_index` makes the function rebuild
its CSR lazily."""
import torch

from .base_classes import ODEblock
from .function_transformer_attention import SpGraphTransAttentionLayer


class HardAttODEblock(ODEblock):
  def __init__(self, odefunc, regularization_fns, opt, data, device, t=torch.tensor([0, 1]), gamma=0.5):
    super(HardAttODEblock, self).__init__(odefunc, regularization_fns, opt, data, device, t)
    assert opt['att_samp_pct'] > 0 and opt['att_samp_pct'] <= 1, "attention sampling threshold must be in (0,1]"
    self.opt = opt
    self.num_nodes = data.num_nodes
    self._second_function(odefunc, opt, data, device)
    self.data_edge_index, _ = self._rw_graph(data, opt, device)   # the full edge set; odefunc.edge_index is a subset of it while training
    self._use_default_integrators(opt)
    self._own_layer = opt['function'] not in {'GAT', 'transformer'}
    if self._own_layer:
      self.multihead_att_layer = SpGraphTransAttentionLayer(opt['hidden_dim'], opt['hidden_dim'], opt, device,
                                                            edge_weights=self.odefunc.edge_weight).to(device)

  def get_attention_weights(self, x):
    layer = self.multihead_att_layer if self._own_layer else self.odefunc.multihead_att_layer
    return layer(x, self.data_edge_index)[0]

  def _edge_scores(self, x, attention):
    score = attention.mean(dim=1)
    if self.opt['use_flux']:
      src, dst = self.data_edge_index
      score = score * torch.linalg.norm(x[src] - x[dst], dim=1)
    return score

  def _sample_edges(self, x, attention):
    """Keep the edges whose score exceeds the (1 - att_samp_pct) quantile (reference :48-66)."""
    from . import ops
    from .block_transformer_rewiring import _device_f32
    score = _device_f32(self._edge_scores(x, attention), 'hard attention edge scores')
    # radix-select quantile (same float32 rank arithmetic as torch.quantile), stable compaction, renormalisation
    thr = ops.quantile(score, 1 - self.opt['att_samp_pct'])
    self.odefunc.edge_index, self.odefunc.attention_weights = ops.threshold_edges(
      self.data_edge_index, score, thr, self.opt['attention_norm_idx'], self.num_nodes)

  def forward(self, x):
    if not self.training and not torch.is_grad_enabled() and self._own_layer:
      # evaluation: every edge carries the diffusion with the head-mean attention -- straight out of the fused row kernels
      # (SpGraphTransAttentionLayer.mean_attention) instead of [E,h] attention and products that are only averaged
      self.odefunc.edge_index = self.data_edge_index
      self.odefunc.attention_weights = self.multihead_att_layer.mean_attention(x, self.data_edge_index)
      twin = self.reg_odefunc.odefunc
      twin.edge_index, twin.edge_weight, twin.attention_weights = (self.odefunc.edge_index, self.odefunc.edge_weight,
                                                                   self.odefunc.attention_weights)
      return self._integrate(x, {'step_size': self.opt['step_size']})
    if self.training:
      # (the reference forms the attention with autograd history and then uses it under no_grad only, src/block_transformer_hard_
      #  attention.py:48-66: no gradient reaches the layer either way; formed without the history it is the same values, bit for bit)
      with torch.no_grad():
        attention = self.get_attention_weights(x)
        self._sample_edges(x, attention)
    else:
      attention = self.get_attention_weights(x)
      self.odefunc.edge_index = self.data_edge_index
      self.odefunc.attention_weights = attention.mean(dim=1)
    twin = self.reg_odefunc.odefunc
    twin.edge_index, twin.edge_weight, twin.attention_weights = (self.odefunc.edge_index, self.odefunc.edge_weight,
                                                                 self.odefunc.attention_weights)
    return self._integrate(x, {'step_size': self.opt['step_size']})
