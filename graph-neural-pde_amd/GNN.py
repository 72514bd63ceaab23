"""The model around the ODE block (reference src/GNN.py:8-72 on top of BaseGNN, src/base_classes.py:97-145): encoder
Linear(s) -> ODE block -> relu -> decoder Linear, with the reference's options (beltrami feature / positional encoders,
use_mlp, use_labels, batch_norm, augment, fc_out), parameter names (m1, m2, mx, mp, m11, m12, fc, bn_in, bn_out) and
`reg_states` / `regularization_coeffs` hand-off to run_GNN.py's train().

At TEST time on a HIP device the encoder is one launch of the fp32-MFMA projection kernel and `relu -> m2` is one launch of
the same kernel with the activation on its A operand (gnpde_relu_linear), so a forward of the launch-bound small-graph
configurations is: encoder launch + ONE graph launch of the solver + decoder launch.  Training keeps the PyTorch layers
(autograd) around the native block."""
import torch
import torch.nn.functional as F
from torch import nn

from . import _lib, ops
from .base_classes import create_regularization_fns
from .model_configurations import set_block, set_function
from .utils import Meter


class BaseGNN(nn.Module):
  def __init__(self, opt, dataset, device=torch.device('cpu')):
    super(BaseGNN, self).__init__()
    self.opt = opt
    self.T = opt['time']
    self.num_classes = dataset.num_classes
    self.num_features = dataset.data.num_features
    self.num_nodes = dataset.data.num_nodes
    self.device = device
    self.fm = Meter()   # forward / backward NFE tallies that run_GNN.py's train() updates (src/base_classes.py:107-108)
    self.bm = Meter()
    if opt['beltrami']:
      self.mx = nn.Linear(self.num_features, opt['feat_hidden_dim'])
      self.mp = nn.Linear(opt['pos_enc_dim'], opt['pos_enc_hidden_dim'])
      opt['hidden_dim'] = opt['feat_hidden_dim'] + opt['pos_enc_hidden_dim']   # (the reference mutates opt the same way)
    else:
      self.m1 = nn.Linear(self.num_features, opt['hidden_dim'])
    if opt['use_mlp']:
      self.m11 = nn.Linear(opt['hidden_dim'], opt['hidden_dim'])
      self.m12 = nn.Linear(opt['hidden_dim'], opt['hidden_dim'])
    if opt['use_labels']:
      opt['hidden_dim'] = opt['hidden_dim'] + dataset.num_classes
    else:
      self.hidden_dim = opt['hidden_dim']
    if opt['fc_out']:
      self.fc = nn.Linear(opt['hidden_dim'], opt['hidden_dim'])
    self.m2 = nn.Linear(opt['hidden_dim'], dataset.num_classes)
    if opt['batch_norm']:
      self.bn_in = nn.BatchNorm1d(opt['hidden_dim'])
      self.bn_out = nn.BatchNorm1d(opt['hidden_dim'])
    self.regularization_fns, self.regularization_coeffs = create_regularization_fns(opt)

  def getNFE(self):
    return self.odeblock.odefunc.nfe + self.odeblock.reg_odefunc.odefunc.nfe

  def resetNFE(self):
    self.odeblock.odefunc.nfe = 0
    self.odeblock.reg_odefunc.odefunc.nfe = 0

  def reset(self):
    self.m1.reset_parameters()
    self.m2.reset_parameters()

  def __repr__(self):
    return self.__class__.__name__


class GNN(BaseGNN):
  def __init__(self, opt, dataset, device=torch.device('cpu')):
    super(GNN, self).__init__(opt, dataset, device)
    self.f = set_function(opt)
    block = set_block(opt)
    time_tensor = torch.tensor([0, self.T]).to(device)
    self.odeblock = block(self.f, self.regularization_fns, opt, dataset.data, device, t=time_tensor).to(device)

  # ---- pieces ----------------------------------------------------------------------------------
  def _native(self, x):
    return (not self.training) and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()

  def _lin(self, layer, x, relu_input=False):
    # the projection kernel is built for the hot path's shapes (K a multiple of 16, >= 16 outputs); a bag-of-words encoder (Cora: K = 1433)
    # or a 7-class decoder would take its guarded scalar-load variant -- 158 / 64 us against ~20 for the vendor GEMM, measured on Cora
    # (profiles/r05_cora_epoch_kernel_stats.csv) -- so those plain GEMMs outside the ODE go to the library
    if self._native(x) and x.shape[1] % 16 == 0 and layer.weight.shape[0] >= 16:
      return ops.linear(x, layer.weight.detach(), None if layer.bias is None else layer.bias.detach(), relu_input=relu_input)
    return layer(F.relu(x) if relu_input else x)

  def encode(self, x, pos_encoding=None):
    opt = self.opt
    y = None
    if opt['use_labels']:
      y = x[:, -self.num_classes:]
      x = x[:, :-self.num_classes]
    drop = lambda t, p: F.dropout(t, p, training=self.training)   # noqa: E731
    if opt['beltrami']:
      x = torch.cat([self._lin(self.mx, drop(x, opt['input_dropout'])),
                     self._lin(self.mp, drop(pos_encoding, opt['input_dropout']))], dim=1)
    else:
      x = self._lin(self.m1, drop(x, opt['input_dropout']))
    if opt['use_mlp']:
      x = drop(x, opt['dropout'])
      x = drop(x + self._lin(self.m11, x, relu_input=True), opt['dropout'])
      x = drop(x + self._lin(self.m12, x, relu_input=True), opt['dropout'])
    if y is not None:
      x = torch.cat([x, y], dim=-1)
    if opt['batch_norm']:
      x = self.bn_in(x)
    if opt['augment']:
      x = torch.cat([x, torch.zeros(x.shape).to(self.device)], dim=1)
    return x

  def decode(self, z, width):
    opt = self.opt
    if opt['augment']:
      z = torch.split(z, width // 2, dim=1)[0]
    if self._native(z):       # relu -> [fc -> relu] -> (dropout: identity at test time) -> m2, one launch per Linear
      if opt['fc_out']:
        z = self._lin(self.fc, z, relu_input=True)
      return self._lin(self.m2, z, relu_input=True)
    z = F.relu(z)
    if opt['fc_out']:
      z = F.relu(self.fc(z))
    z = F.dropout(z, opt['dropout'], training=self.training)
    return self.m2(z)

  def forward(self, x, pos_encoding=None):
    x = self.encode(x, pos_encoding)
    self.odeblock.set_x0(x)
    if self.training and self.odeblock.nreg > 0:
      z, self.reg_states = self.odeblock(x)
    else:
      z = self.odeblock(x)
    return self.decode(z, x.shape[1])
