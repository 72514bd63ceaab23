"""Block that rewires the graph with its own attention while training -- adds edges (random pairs, or the two-hop
neighbourhood of the current transition matrix), drops the weakest ones, renormalises -- and always integrates over
head-mean attention recomputed on the current edge set (reference src/block_transformer_rewiring.py:10-260,
`--block rewire_attention`).

The rewiring is once-per-forward bookkeeping on the device: quantile, compaction + renormalisation (csrc/rewire.hip) and
the two-hop densification (csrc/twohop.hip) are native kernels, random-edge de-duplication is torch.unique as in the
reference; the attention and every evaluation of f inside the solver run on the native kernels.  The function's CSR is rebuilt lazily because
`edge_index` is a new tensor after each rewiring."""
import numpy as np
import torch

from .base_classes import ODEblock
from .function_transformer_attention import SpGraphTransAttentionLayer


def _device_f32(t, what):
  """The rewiring bookkeeping has no host / PyTorch path: anything but a float32 HIP tensor is refused, like every
  other entry of the package (GnpdeError)."""
  if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32):
    from ._lib import GnpdeError
    raise GnpdeError('%s: needs a float32 tensor on a HIP device, got %s on %s (there is no CPU / PyTorch fallback)'
                     % (what, getattr(t, 'dtype', type(t)), getattr(t, 'device', 'host')))
  return t


class RewireAttODEblock(ODEblock):
  def __init__(self, odefunc, regularization_fns, opt, data, device, t=torch.tensor([0, 1]), gamma=0.5):
    super(RewireAttODEblock, self).__init__(odefunc, regularization_fns, opt, data, device, t)
    assert opt['att_samp_pct'] > 0 and opt['att_samp_pct'] <= 1, "attention sampling threshold must be in (0,1]"
    self.opt = opt
    self._second_function(odefunc, opt, data, device)
    self.num_nodes = data.num_nodes
    self.data_edge_index, _ = self._rw_graph(data, opt, device)   # changed by the rewiring
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

  def add_random_edges(self):
    """M uniformly random (ordered) node pairs, duplicates of existing edges dropped (reference :52-66; same numpy
    generator call, so a seeded run draws the same pairs)."""
    M = int(self.num_nodes * (1 / (1 - (self.opt['rw_addD'])) - 1))
    with torch.no_grad():
      new_edges = torch.tensor(np.random.choice(self.num_nodes, size=(2, M), replace=True, p=None))
      cat = torch.cat([self.data_edge_index, new_edges.to(self.data_edge_index)], dim=1)
      self.data_edge_index = torch.unique(cat, sorted=False, return_inverse=False, return_counts=False, dim=1)
      self.odefunc.edge_index = self.data_edge_index

  def add_khop_edges(self, k=2, rm_self_loops=True):
    """(A + A^2 without its diagonal) / 2 on the current transition matrix (reference :68-86)."""
    n = self.num_nodes
    for _ in range(k - 1):
      ei, ew = self.odefunc.edge_index, _device_f32(self.odefunc.edge_weight, 'add_khop_edges')
      # one row-wise kernel for spspmm -> remove_self_loops -> cat -> / 2 -> coalesce (csrc/twohop.hip)
      from . import ops
      from .graph import graph_of
      ei, ew = ops.two_hop(graph_of(ei, n), ew)
      self.data_edge_index = ei
      self.odefunc.edge_index = self.data_edge_index
      self.odefunc.attention_weights = ew

  def densify_edges(self):
    kind = self.opt['new_edges']
    if kind == 'random':
      self.add_random_edges()
    elif kind == 'random_walk':
      raise NotImplementedError("new_edges='random_walk' is not implemented in the reference either "
                                '(its add_rw_edges is commented out)')
    elif kind == 'k_hop_lap':
      pass
    elif kind == 'k_hop_att':
      self.add_khop_edges(k=2)

  def threshold_edges(self, x, threshold):
    if self.opt['new_edges'] == 'k_hop_att' and self.opt['sparsify'] == 'S_hat':   # sparsify on (A + A^2) / 2
      mean_att = self.odefunc.attention_weights
    else:                                                                           # on recomputed attention
      mean_att = self.get_attention_weights(x).mean(dim=1, keepdim=False)
    if self.opt['use_flux']:
      delta = torch.linalg.norm(x[self.data_edge_index[0, :], :] - x[self.data_edge_index[1, :], :], dim=1)
      mean_att = mean_att * delta
    total = self.data_edge_index.shape[1]
    from . import ops          # stable compaction + renormalisation (csrc/rewire.hip)
    if not torch.is_tensor(threshold):
      threshold = torch.tensor(float(threshold), dtype=torch.float32, device=mean_att.device)
    kept, sampled_attention_weights = ops.threshold_edges(self.data_edge_index, _device_f32(mean_att, 'threshold_edges'),
                                                          threshold, self.opt['attention_norm_idx'], self.num_nodes)
    self.odefunc.edge_index = kept
    print('retaining {} of {} edges'.format(self.odefunc.edge_index.shape[1], total))
    self.data_edge_index = kept
    self.odefunc.edge_weight = sampled_attention_weights
    self.odefunc.attention_weights = sampled_attention_weights

  def forward(self, x):
    if self.training:
      with torch.no_grad():
        attention_weights = self.get_attention_weights(x)
        self.odefunc.attention_weights = attention_weights.mean(dim=1, keepdim=False)
        pre_count = self.odefunc.edge_index.shape[1]
        self.densify_edges()
        post_count = self.odefunc.edge_index.shape[1]
        pc_change = post_count / pre_count - 1
        q = 1 / (pc_change - self.opt['rw_addD'])
        from . import ops
        threshold = ops.quantile(_device_f32(self.odefunc.edge_weight, 'rewiring quantile'), q)   # radix select, same float32 rank arithmetic as torch.quantile
        self.threshold_edges(x, threshold)
    self.odefunc.edge_index = self.data_edge_index
    if not torch.is_grad_enabled() and self.opt['function'] not in {'GAT', 'transformer'}:
      mean_att = self.multihead_att_layer.mean_attention(x, self.data_edge_index)     # (evaluation: the head mean alone, fused row kernels)
    else:
      mean_att = self.get_attention_weights(x).mean(dim=1, keepdim=False)
    self.odefunc.edge_weight = mean_att
    self.odefunc.attention_weights = mean_att
    self.reg_odefunc.odefunc.edge_index, self.reg_odefunc.odefunc.edge_weight = self.odefunc.edge_index, self.odefunc.edge_weight
    self.reg_odefunc.odefunc.attention_weights = self.odefunc.attention_weights
    return self._integrate(x, {'step_size': self.opt['step_size']})
