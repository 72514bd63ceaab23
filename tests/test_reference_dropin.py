"""Container-only: the reference's own GNN.py / GNN_early.py / model_configurations.py pick up the MI355X classes once
gnpde_amd.dropin.install() answers the reference's module names, and the resulting model's state_dict interchanges
with the fixture recorded from the pure reference model.  Skipped where /root/reference is absent (the GPU box)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_env  # noqa: E402

SCRIPT = r'''
import sys, os, json
ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import ref_env
import gnpde_amd
import gnpde_amd.dropin as dropin
ref_env.activate()                      # stand-ins + reference src on sys.path
served = dropin.install(native_gnn=True)
assert 'base_classes' in served and 'block_constant' in served and dropin.installed()
import torch
from helpers import Fixture
import importlib.util
spec = importlib.util.spec_from_file_location('_reference_GNN', os.path.join(ref_env.REFERENCE_ROOT, 'src', 'GNN.py'))
ref_gnn = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref_gnn)
from GNN import GNN as NativeGNN        # install(native_gnn=True): fused encoder / decoder launches
assert NativeGNN.__module__ == 'gnpde_amd.GNN'
GNN = ref_gnn.GNN                       # the reference's model, unmodified, over the drop-in blocks / functions
from utils import DummyDataset          # the reference's utils
from torch_geometric.data import Data
for name in ('gnn_constant_transformer_rk4', 'gnn_attention_laplacian_euler'):
  fx = Fixture(name)
  data = Data(x=fx.t('x'), edge_index=fx.t('edge_index'), edge_attr=None)
  model = GNN(dict(fx.opt), DummyDataset(data, int(fx.arr['num_classes'])), torch.device('cpu'))
  f = model.odeblock.odefunc
  assert type(f).__module__.startswith('gnpde_amd'), type(f).__module__
  assert type(model.odeblock).__module__.startswith('gnpde_amd')
  ours = {k: tuple(v.shape) for k, v in model.state_dict().items()}
  theirs = {k: tuple(v.shape) for k, v in fx.params.items()}
  assert ours == theirs, set(ours) ^ set(theirs)
  model.load_state_dict(fx.params, strict=True)
  print(model)                          # the reference's print(model) raises for --function transformer
  native = NativeGNN(dict(fx.opt), DummyDataset(data, int(fx.arr['num_classes'])), torch.device('cpu'))
  assert {k: tuple(v.shape) for k, v in native.state_dict().items()} == theirs
  native.load_state_dict(model.state_dict(), strict=True)
  assert native.regularization_coeffs == model.regularization_coeffs and repr(native) == repr(model)
# the reference's early-stopping model (GNN_early.py) picks up the device evaluator, its registry the rewiring block
from GNN_early import GNNEarly
fx = Fixture('gnn_constant_transformer_rk4')
data = Data(x=fx.t('x'), edge_index=fx.t('edge_index'), edge_attr=None)
opt = dict(fx.opt, earlystopxT=3, max_test_steps=100)
early = GNNEarly(opt, DummyDataset(data, int(fx.arr['num_classes'])), torch.device('cpu'))
integ = early.odeblock.test_integrator
assert type(integ).__module__.startswith('gnpde_amd'), type(integ).__module__
assert integ.data is data and abs(float(integ.t[1]) - 3 * opt['time']) < 1e-6
early.load_state_dict(fx.params, strict=True)
early.set_solver_m2()
assert integ.m2_weight.shape == early.m2.weight.shape
from model_configurations import set_block
assert set_block(dict(opt, block='rewire_attention')).__module__.startswith('gnpde_amd')
import base_classes                      # merged: the reference's BaseGNN / registry, this package's hot-path types
assert base_classes.ODEFunc is gnpde_amd.base_classes.ODEFunc and base_classes.ODEblock is gnpde_amd.base_classes.ODEblock
assert base_classes.BaseGNN.__module__ == '_reference_base_classes' and base_classes.__gnpde_reference__.endswith('src/base_classes.py')
assert issubclass(GNNEarly, base_classes.BaseGNN)
print('DROPIN_OK')
'''


@pytest.mark.skipif(not ref_env.available(), reason='reference tree not present')
def test_reference_gnn_builds_on_native_classes():
  res = subprocess.run([sys.executable, '-c', SCRIPT, ROOT], capture_output=True, text=True, timeout=300)
  assert res.returncode == 0 and 'DROPIN_OK' in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


@pytest.mark.skipif(not ref_env.available(), reason='reference tree not present')
def test_reference_unit_tests_pass_over_the_stand_ins():
  """The reference's whole test directory (24 tests: attention known answers, normalisations, blocks, GNN, early
  stopping) passes over oracle/shims -- what pins the stand-ins the golden vectors were generated with."""
  res = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'run_reference_tests.py')], capture_output=True,
                       text=True, timeout=600)
  assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
  assert 'ran 24, failures 0, errors 0' in res.stdout


KHOP_SCRIPT = r'''
import sys, os, types
ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
from oracle import ref_env
ref_env.activate()
import torch
from oracle import restate as R
from block_transformer_rewiring import RewireAttODEblock       # the reference's block, unmodified
g = torch.Generator().manual_seed(3)
n = 60
ei = torch.randint(0, n, (2, 300), generator=g)
ei = torch.cat([ei, ei[:, :7], torch.tensor([[4, 9], [4, 9]])], dim=1)     # duplicates and self loops
w = torch.rand(ei.shape[1], generator=g)
fake = types.SimpleNamespace(num_nodes=n, odefunc=types.SimpleNamespace(edge_index=ei, edge_weight=w, attention_weights=None))
RewireAttODEblock.add_khop_edges(fake, k=2)
idx, val = R.two_hop(ei, w, n)
assert torch.equal(fake.data_edge_index, idx), (fake.data_edge_index.shape, idx.shape)
assert torch.allclose(fake.odefunc.attention_weights.double(), val, rtol=1e-5, atol=1e-7)
print('KHOP_OK', idx.shape[1])
'''


@pytest.mark.skipif(not ref_env.available(), reason='reference tree not present')
def test_oracle_two_hop_is_the_reference_blocks_add_khop_edges():
  """Pins oracle.restate.two_hop (what the native two-hop kernel is tested against) on the reference's own
  RewireAttODEblock.add_khop_edges run over the stand-ins."""
  res = subprocess.run([sys.executable, '-c', KHOP_SCRIPT, ROOT], capture_output=True, text=True, timeout=300)
  assert res.returncode == 0 and 'KHOP_OK' in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
