"""GRAND-l right-hand side f = alpha' (A x - x) + beta x0 with fixed weights A
(reference src/function_laplacian_diffusion.py:15-51).  One gnpde_spmm_rhs launch per evaluation."""
import torch
from torch import nn

from . import _lib, ops
from .base_classes import ODEFunc


class LaplacianODEFunc(ODEFunc):

  # currently requires in_features = out_features
  def __init__(self, in_features, out_features, opt, data, device):
    super(LaplacianODEFunc, self).__init__(opt, data, device)
    self.in_features = in_features
    self.out_features = out_features
    # registered but unused by the reference's forward as well (kept for state_dict parity)
    self.w = nn.Parameter(torch.eye(opt['hidden_dim']))
    self.d = nn.Parameter(torch.zeros(opt['hidden_dim']) + 1)
    self.alpha_sc = nn.Parameter(torch.ones(1))
    self.beta_sc = nn.Parameter(torch.ones(1))

  def _edge_values(self):
    """Which per-edge weights define A (reference :28-36): the attention block hands [E,h]
    attention (head mean taken here), mixed / hard_attention hand [E], everything else uses
    edge_weight."""
    block = self.opt['block']
    if block in ['attention', 'mixed', 'hard_attention']:
      src, what = self.attention_weights, 'attention_weights'
    else:
      src, what = self.edge_weight, 'edge_weight'
    if src is None:
      raise _lib.GnpdeError('LaplacianODEFunc.%s has not been set for block=%r' % (what, block))
    return src

  def _weights_csr(self, graph):
    src = self._edge_values()
    # one buffer per graph OBJECT, the two newest kept (the solver may run on the locality view of the graph direct calls use)
    ents = self._cache.setdefault('w_csr', [])
    ent = next((e for e in ents if e['graph'] is graph), None)
    if ent is None:
      ent = {'graph': graph, 'buf': torch.empty(max(graph.e, 1), dtype=torch.float32, device=graph.device),
             'sig': None, 'src': None, 'gen': 0}
      ents[:] = ents[-1:] + [ent]
    sig = (id(src), src._version)
    if ent['sig'] != sig:
      ops.edge_to_csr_mean(graph, src, out=ent['buf'])  # in place: captured graphs keep the pointer
      ent['sig'], ent['src'] = sig, src
      # generation of the buffer's CONTENTS: ids of freed tensors are reused by Python, this counter is not
      self._w_generation = getattr(self, '_w_generation', 0) + 1
      ent['gen'] = self._w_generation
    return ent['buf']

  def sparse_multiply(self, x):
    """A x alone (reference :28-36)."""
    graph = self._graph(x)
    with torch.no_grad():
      return ops.spmm(graph, self._weights_csr(graph), _lib.f32c(x))

  def _descriptor(self, x, x0_override=None, graph=None):
    graph = self._graph(x) if graph is None else graph
    x0 = x0_override if x0_override is not None else self._source(x)
    alpha = ops._scalar_dev(self.alpha_train, x)
    beta = ops._scalar_dev(self.beta_train, x) if x0 is not None else None
    return ops.RhsDescriptor(_lib.RHS_LAPLACIAN, graph, x.shape[1], x.stride(0), alpha, beta,
                             None if x0 is None else self._match_rows(x0, x), not self.opt['no_alpha_sigmoid'],
                             w_csr=self._weights_csr(graph), padded_rows=_lib.is_padded(x))

  def _descriptor_signature(self, desc):
    s = desc.struct
    return (id(desc.graph), s.alpha, s.beta, s.x0, s.alpha_sigmoid, s.w_csr, s.d, s.ld)
