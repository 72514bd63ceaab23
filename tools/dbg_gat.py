import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
import gnpde_amd as G
from helpers import Fixture, Data
fx = Fixture('adjoint_constant_gat_rk4_rk4')
dev = torch.device('cuda:0')
x = fx.t('x', dev)
block = G.ConstantODEblock(G.ODEFuncAtt, [], fx.opt, Data(x, fx.t('edge_index', dev)), dev, t=torch.tensor([0, fx.opt['time']])).to(dev)
block.load_state_dict(fx.params, strict=True)
block.train()
xin = x.clone().requires_grad_(True)
block.set_x0(xin)
z = block(xin)
try:
  (z * fx.t('c', dev)).sum().backward()
  torch.cuda.synchronize()
  print('ok', float(xin.grad.abs().max()))
except Exception as e:
  print('ERR', repr(e)[:300])
  print('last error:', G.lib().gnpde_last_error())
