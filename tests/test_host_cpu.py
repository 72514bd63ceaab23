"""Host-side logic that needs no GPU: C-ABI surface, graph preparation, partitioner, normalisation
helpers, the host integrator loops, parameter-name parity and loud failure without a HIP device."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

import gnpde_amd as G
from gnpde_amd import _lib
from oracle import restate as R
from helpers import Fixture, fixtures, Data, assert_parity, random_graph
from test_oracle_golden import _block_rhs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
  header = open(os.path.join(ROOT, 'include', 'gnpde.h')).read()
  declared = set(re.findall(r'\b(gnpde_[a-z_0-9]+)\s*\(', header))
  assert len(declared) >= 20
  lib = ctypes.CDLL(_lib.LIB_PATH)
  for name in sorted(declared):
    assert hasattr(lib, name), 'libgnpde_hip.so does not export %s' % name
  assert declared == set(_lib.PROTOTYPES), 'ctypes prototypes out of sync with gnpde.h'
  # one ABI number in three places: the header, the library built from it, the Python side's struct layouts
  in_header = int(re.search(r'#define\s+GNPDE_ABI_VERSION\s+(\d+)', header).group(1))
  assert G.lib().gnpde_abi_version() == in_header == _lib.ABI_VERSION


def test_driver_build_hook_passes():
  """__graft_entry__.build() -- the driver's "does it build" check -- compiles what changed, loads the library and agrees with it
  on the ABI version (a hard-coded number there went stale once)."""
  import __graft_entry__ as entry
  entry.build()
  src = open(os.path.join(ROOT, '__graft_entry__.py')).read()
  assert not re.search(r'gnpde_abi_version\(\)\s*==\s*\d', src), 'compare with _lib.ABI_VERSION, not a literal'


def test_every_entry_point_is_documented():
  """INTEGRATION.md's table names every symbol of the C ABI (what it replaces in the reference)."""
  header = open(os.path.join(ROOT, 'include', 'gnpde.h')).read()
  doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
  missing = [n for n in sorted(set(re.findall(r'\b(gnpde_[a-z_0-9]+)\s*\(', header))) if n not in doc]
  assert not missing, 'undocumented entry points: %s' % missing


def test_structs_match_header_layout():
  # pointer-sized fields and int32s only: sizes are what the C compiler produces on x86-64
  assert ctypes.sizeof(_lib.GraphStruct) == 8 + 6 * 8 + 8 + 5 * 8 + 24 + 3 * 8 + 8   # (+ xcd_deal and its padding)
  assert _lib.GraphStruct.xcd_deal.offset == 8 + 6 * 8 + 8 + 5 * 8 + 24 + 3 * 8
  assert _lib.GraphStruct.n_bin_le64.offset == _lib.GraphStruct.xcd_deal.offset + 4     # (ABI 3: in what was padding)
  assert ctypes.sizeof(_lib.EpilogueStruct) == 3 * 8 + 4 + 4 + 4 + 4 + 6 * 8 + 8 + 7 * 8 + 8 * 4 + 8
  assert ctypes.sizeof(_lib.AttentionStruct) == 24 + 8 + 8 + 8 + 4 * 8 + 2 * 8     # (+ graph_t, t_from_csr: ABI 3)
  assert _lib.AttentionStruct.n_key_rows.offset == _lib.AttentionStruct.ldqk.offset + 4   # (ABI 3: in what was padding)
  assert ctypes.sizeof(_lib.DecoderStruct) == 4 * 8 + 2 * 4


@pytest.mark.parametrize('seed', [0, 1])
def test_graph_build_matches_numpy(seed):
  n = 700
  ei = random_graph(n, 5, seed=seed, hubs=2, hub_deg=900, isolated=4, dup=30)
  g = G.CSRGraph(ei, n)
  row, col = ei[0].numpy(), ei[1].numpy()
  order = np.argsort(row, kind='stable')
  assert np.array_equal(g.perm.numpy(), order)
  assert np.array_equal(g.colidx.numpy(), col[order])
  assert np.array_equal(g.t['rowidx'][:g.e].numpy(), row[order])
  assert np.array_equal(g.rowptr.numpy(), np.concatenate([[0], np.cumsum(np.bincount(row, minlength=n))]))
  # CSC view: positions of each column's entries, ascending
  cpos = g.t['cscpos'][:g.e].numpy()
  cptr = g.t['cscptr'].numpy()
  assert np.array_equal(cptr, np.concatenate([[0], np.cumsum(np.bincount(col, minlength=n))]))
  ccol = g.colidx.numpy()[cpos]
  assert np.all(np.diff(ccol) >= 0)
  for c in (0, 5, n - 5):
    seg = cpos[cptr[c]:cptr[c + 1]]
    assert np.all(np.diff(seg) > 0) and np.all(g.colidx.numpy()[seg] == c)
  # long rows
  deg = np.diff(g.rowptr.numpy())
  long_rows = np.nonzero(deg > _lib.LONG_ROW)[0]
  assert g.n_long_rows == len(long_rows) >= 2
  assert np.array_equal(g.t['long_rows'][:g.n_long_rows].numpy(), long_rows)
  b, e_ = g.t['long_chunk_begin'][:g.n_long_chunks].numpy(), g.t['long_chunk_end'][:g.n_long_chunks].numpy()
  assert np.all(e_ - b <= _lib.LONG_ROW) and (e_ - b).sum() == deg[long_rows].sum()
  # degree classes
  rec = g.t['bin_rows'].numpy().reshape(-1, 4)[:g.n_bin16 + g.n_bin64]
  assert np.array_equal(rec[:, 1], g.rowptr.numpy()[rec[:, 0]]) and np.array_equal(rec[:, 2], deg[rec[:, 0]])
  b16, b64 = rec[:g.n_bin16, 0], rec[g.n_bin16:, 0]
  assert np.array_equal(b16, np.nonzero((deg >= 1) & (deg <= 16))[0])
  # rows of 17..512 entries: longest first, ties in row order (the long rows start the row-attention launch, not its tail)
  in64 = np.nonzero((deg > 16) & (deg <= _lib.LONG_ROW))[0]
  assert np.array_equal(b64, in64[np.argsort(-deg[in64], kind='stable')])
  cdeg = np.bincount(col, minlength=n)
  assert np.array_equal(g.t['long_cols'][:g.n_long_cols].numpy(), np.nonzero(cdeg > _lib.LONG_ROW)[0])


def test_graph_build_rejects_bad_index():
  with pytest.raises(G.GnpdeError):
    G.CSRGraph(torch.tensor([[0, 5], [1, 2]]), 3)


def test_graph_empty_and_cache():
  g = G.CSRGraph(torch.zeros(2, 0, dtype=torch.long), 4)
  assert g.e == 0 and g.rowptr.tolist() == [0, 0, 0, 0, 0]
  ei = random_graph(50, 3, seed=2)
  a = G.graph_of(ei, 50)
  assert G.graph_of(ei, 50) is a
  ei[0, 0] = (ei[0, 0] + 1) % 50  # in-place edit bumps the version -> rebuilt
  assert G.graph_of(ei, 50) is not a


@pytest.mark.parametrize('parts', [2, 4, 8])
def test_partition_balanced(parts):
  ei, n = G.synthetic.make_graph('arxiv', seed=1, scale=0.05)
  g = G.CSRGraph(ei, n)
  part = G.partition_rows(g, parts)
  assert part.min() == 0 and part.max() == parts - 1
  work = (g.rowptr[1:] - g.rowptr[:-1]).long() + 1
  load = torch.zeros(parts, dtype=torch.long).index_add_(0, part.long(), work)
  assert load.max().item() <= 1.1 * load.float().mean().item()
  # the partition must beat a random assignment on edge cut
  cut = (part[ei[0]] != part[ei[1]]).float().mean().item()
  rnd = torch.randint(0, parts, (n,), generator=torch.Generator().manual_seed(0))
  assert cut < (rnd[ei[0]] != rnd[ei[1]]).float().mean().item()


def test_normalisation_helpers_match_golden():
  fx = Fixture('norms')
  ei, w = fx.t('edge_index'), fx.t('edge_weight')
  for fill in (0.0, 0.3, 1.0, 3.2):
    for nd in (0, 1):
      e2, w2 = G.get_rw_adj(ei, edge_weight=w, norm_dim=nd, fill_value=fill, num_nodes=50, dtype=torch.float32)
      assert torch.equal(e2, fx.t('rw_ei_f%g_n%d' % (fill, nd)))
      assert_parity(w2, fx.t('rw_w_f%g_n%d' % (fill, nd)), 2e-6, 'rw')
    e2, w2 = G.gcn_norm_fill_val(ei, edge_weight=w, fill_value=fill, num_nodes=50, dtype=torch.float32)
    assert torch.equal(e2, fx.t('gcn_ei_f%g' % fill))
    assert_parity(w2, fx.t('gcn_w_f%g' % fill), 2e-6, 'gcn')
  e2, w2 = G.get_rw_adj(ei, None, norm_dim=1, fill_value=1.0, num_nodes=50, dtype=torch.float32)
  assert torch.equal(e2, fx.t('rw_ei_unweighted'))
  assert_parity(w2, fx.t('rw_w_unweighted'), 2e-6, 'rw unweighted')


@pytest.mark.parametrize('name', fixtures('block_'))
def test_host_integrator_loops(name):
  """The host euler / rk4 / dopri5 loops (used for foreign callables and the adaptive method)
  reproduce the reference's block output when f is the CPU oracle."""
  fx = Fixture(name)
  opt = fx.opt
  f = _block_rhs(fx)
  calls = [0]

  def counted(t, y):
    calls[0] += 1
    return f(t, y)

  z = G.odeint(counted, fx.t('x'), torch.tensor([0, opt['time']], dtype=torch.float32), method=opt['method'],
               options=dict(step_size=opt['step_size'], max_iters=opt['max_iters']),
               atol=opt['tol_scale'] * 1e-7, rtol=opt['tol_scale'] * 1e-9)[1]
  assert_parity(z, fx.t('z'), 5e-6, name)
  assert calls[0] == int(fx.arr['nfe'])


def test_time_grid():
  t = torch.tensor([0, 18.294754260552843])
  g = G.time_grid(t, 1.0)
  assert torch.equal(g, R.time_grid(18.294754260552843, 1.0))
  assert len(G.time_grid(torch.tensor([0., 2.2]), 0.5)) == 6


@pytest.mark.parametrize('name', fixtures('block_') + fixtures('func_'))
def test_state_dict_names_match_reference(name):
  """Parameter names / shapes interchange with the reference's state_dicts."""
  fx = Fixture(name)
  x = fx.t('x')
  data = Data(x, fx.t('edge_index'))
  fcls = {'laplacian': G.LaplacianODEFunc, 'transformer': G.ODEFuncTransformerAtt, 'GAT': G.ODEFuncAtt}[fx.opt['function']]
  if name.startswith('block_'):
    bcls = {'constant': G.ConstantODEblock, 'attention': G.AttODEblock, 'mixed': G.MixedODEblock,
            'hard_attention': G.HardAttODEblock}[fx.opt['block']]
    mod = bcls(fcls, [], fx.opt, data, torch.device('cpu'), t=torch.tensor([0, fx.opt['time']]))
  else:
    mod = fcls(x.shape[1], x.shape[1], fx.opt, data, torch.device('cpu'))
  ours = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
  theirs = {k: tuple(v.shape) for k, v in fx.params.items()}
  assert ours == theirs
  mod.load_state_dict(fx.params, strict=True)


def test_registry():
  assert G.set_function({'function': 'transformer'}) is G.ODEFuncTransformerAtt
  assert G.set_function({'function': 'GAT'}) is G.ODEFuncAtt
  assert G.set_function({'function': 'laplacian'}) is G.LaplacianODEFunc
  assert G.set_block({'block': 'constant'}) is G.ConstantODEblock
  assert G.set_block({'block': 'attention'}) is G.AttODEblock
  assert G.set_block({'block': 'mixed'}) is G.MixedODEblock
  assert G.set_block({'block': 'hard_attention'}) is G.HardAttODEblock
  with pytest.raises(G.FunctionNotDefined):
    G.set_function({'function': 'nope'})
  with pytest.raises(G.BlockNotDefined):
    G.set_block({'block': 'nope'})


def test_heads_must_divide_attention_dim():
  """reference test_gnn.py: heads not dividing attention_dim raises AssertionError."""
  fx = Fixture('func_transformer_sd_softmax_n0')
  opt = dict(fx.opt, heads=3, attention_dim=16)
  with pytest.raises(AssertionError):
    G.ODEFuncTransformerAtt(24, 24, opt, Data(fx.t('x'), fx.t('edge_index')), torch.device('cpu'))


def test_cpu_tensors_fail_loudly():
  """There is no CPU fallback: the product path refuses host tensors instead of computing on them."""
  fx = Fixture('func_laplacian_constant')
  x = fx.t('x')
  func = G.LaplacianODEFunc(x.shape[1], x.shape[1], fx.opt, Data(x, fx.t('edge_index')), torch.device('cpu'))
  func.edge_index, func.edge_weight, func.x0 = fx.t('func_edge_index'), fx.t('edge_weight'), fx.t('x0')
  with torch.no_grad(), pytest.raises(G.GnpdeError):
    func(0.0, x)
  assert not os.path.exists(os.path.join(ROOT, 'graph-neural-pde_amd', 'fallback.py'))
  # the rewiring / hard-attention bookkeeping has no PyTorch branch either: host tensors are refused
  fx = Fixture('rewire_khop_shat')
  x = fx.t('x')
  block = G.RewireAttODEblock(G.LaplacianODEFunc, [], dict(fx.opt), Data(x, fx.t('edge_index')), torch.device('cpu'),
                              t=torch.tensor([0, fx.opt['time']]))
  with pytest.raises(G.GnpdeError):
    block.add_khop_edges(k=2)
  block.odefunc.attention_weights = block.odefunc.edge_weight
  with pytest.raises(G.GnpdeError):
    block.threshold_edges(x, 0.5)
  hard = G.HardAttODEblock(G.LaplacianODEFunc, [], dict(fx.opt, att_samp_pct=0.5), Data(x, fx.t('edge_index')),
                           torch.device('cpu'), t=torch.tensor([0, fx.opt['time']]))
  with pytest.raises(G.GnpdeError):
    hard._sample_edges(x, torch.rand(hard.data_edge_index.shape[1], 4))
  import inspect
  from gnpde_amd import block_transformer_rewiring as B
  src = inspect.getsource(B)
  for gone in ('_spspmm', '_coalesce', 'torch.quantile(', 'index_add_'):
    assert gone not in src, 'PyTorch branch %r is back in the rewiring block' % gone


def test_early_stop_integrator_surface():
  """EarlyStopInt keeps the reference's constructor, time span and the fields GNNEarly writes / run_GNN reads
  (reference early_stop_solver.py:234-245, GNN_early.py:28-40, run_GNN.py:266-271); no CPU evaluation path."""
  fx = Fixture('early_rk4_transformer')
  integ = G.EarlyStopInt(fx.opt['time'], fx.opt, torch.device('cpu'))
  assert integ.solver is None and integ.data is None and integ.m2_weight is None and integ.m2_bias is None
  assert integ.max_test_steps == fx.opt['max_test_steps']
  assert integ.t.dtype == torch.float32 and integ.t.tolist() == pytest.approx([0.0, 3 * fx.opt['time']])
  x = fx.t('x')
  with pytest.raises(G.GnpdeError):           # data / decoder never assigned
    integ(lambda t, y: y, x, integ.t, method='rk4', options={'step_size': 1.0})
  data = Data(x, fx.t('edge_index'))
  data.y = fx.t('labels')
  for k in ('train_mask', 'val_mask', 'test_mask'):
    setattr(data, k, fx.t(k).bool())
  integ.data, integ.m2_weight, integ.m2_bias = data, fx.t('m2_weight'), fx.t('m2_bias')
  with pytest.raises(G.GnpdeError):           # host tensors: refused, not computed on the CPU
    integ(lambda t, y: y, x, integ.t, method='rk4', options={'step_size': 1.0})
  bad = G.EarlyStopInt(1.0, dict(fx.opt, method='euler'), torch.device('cpu'))
  with pytest.raises(AssertionError):
    bad(lambda t, y: y, x, bad.t, method='euler', options={'step_size': 1.0})
  from gnpde_amd import dropin
  assert dropin.MODULES['early_stop_solver'] == 'gnpde_amd.early_stop_solver'


def test_product_does_not_import_oracle():
  """The oracle is test infrastructure: no module of the product may import it (comments may name it)."""
  import re
  pkg = os.path.join(ROOT, 'graph-neural-pde_amd')
  for dirpath, _, files in os.walk(pkg):
    for fn in files:
      if fn.endswith('.py'):
        src = open(os.path.join(dirpath, fn)).read()
        assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), '%s imports the oracle' % fn
        assert 'restate' not in src and 'ref_env' not in src, '%s references oracle modules' % fn


def test_rewiring_sparse_helpers_match_dense():
  """Test-side composite of the rewiring block's two-hop step (tests/sparse_composite.py): COO product through the row pointer and duplicate-summing coalesce,
  against dense matrices (what torch_sparse.spspmm / coalesce compute for the reference)."""
  from sparse_composite import _spspmm, _coalesce
  g = torch.Generator().manual_seed(0)
  n = 23
  ia = torch.randint(0, n, (2, 90), generator=g)
  va = torch.rand(90, generator=g)
  ib = torch.randint(0, n, (2, 70), generator=g)
  vb = torch.rand(70, generator=g)
  dense = lambda i, v: torch.zeros(n, n).index_put_((i[0], i[1]), v, accumulate=True)  # noqa: E731
  ic, vc = _spspmm(ia, va, ib, vb, n)
  assert torch.allclose(dense(ic, vc), dense(ia, va) @ dense(ib, vb), atol=1e-6)
  key = ic[0] * n + ic[1]
  assert torch.all(key[1:] > key[:-1]), 'product is not coalesced in row-major order'
  i2, v2 = _coalesce(torch.cat([ia, ia[:, :10]], 1), torch.cat([va, va[:10]]), n)
  assert torch.allclose(dense(i2, v2), dense(ia, va) + dense(ia[:, :10], va[:10]), atol=1e-6)
  empty_i, empty_v = _spspmm(ia[:, :0], va[:0], ib, vb, n)
  assert empty_i.shape == (2, 0) and empty_v.numel() == 0


def test_header_is_valid_c_and_links(tmp_path):
  """include/gnpde.h is a C header (not C++): a C99 translation unit that takes the address of every declared entry
  point compiles with -Wall -Werror and links against libgnpde_hip.so (no device needed to link)."""
  import shutil
  import subprocess
  if shutil.which('gcc') is None:
    pytest.skip('no gcc')
  header = open(os.path.join(ROOT, 'include', 'gnpde.h')).read()
  names = sorted(set(re.findall(r'\b(gnpde_[a-z_0-9]+)\s*\(', header)))
  src = tmp_path / 'abi.c'
  src.write_text('#include "gnpde.h"\n#include <stdio.h>\n'
                 'typedef void (*fn_t)(void);\nint main(void) {\n  fn_t table[] = {\n' +
                 ''.join('    (fn_t)%s,\n' % n for n in names) +
                 '  };\n  gnpde_graph_t g; gnpde_epilogue_t e; gnpde_rhs_t r; gnpde_decoder_t d;\n'
                 '  printf("%zu %zu %zu %zu %zu\\n", sizeof table / sizeof table[0], sizeof g, sizeof e, sizeof r, sizeof d);\n'
                 '  return 0;\n}\n')
  libdir = os.path.dirname(_lib.LIB_PATH)
  exe = tmp_path / 'abi'
  cmd = ['gcc', '-std=c99', '-Wall', '-Werror', '-pedantic', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe),
         '-L', libdir, '-lgnpde_hip', '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib', '-L', '/opt/rocm/lib',
         '-Wl,--allow-shlib-undefined']
  res = subprocess.run(cmd, capture_output=True, text=True)
  assert res.returncode == 0, res.stderr[-3000:]
  out = subprocess.run([str(exe)], capture_output=True, text=True)
  if out.returncode == 0:      # runs wherever the HIP runtime library can be loaded (it only prints sizes)
    n, sg, se, sr, sd = (int(v) for v in out.stdout.split())
    assert n == len(names)
    assert (sg, se, sd) == (ctypes.sizeof(_lib.GraphStruct), ctypes.sizeof(_lib.EpilogueStruct), ctypes.sizeof(_lib.DecoderStruct))
    assert sr == ctypes.sizeof(_lib.RhsStruct)


@pytest.mark.parametrize('seed,hubs', [(0, 0), (1, 3), (2, 1)])
def test_device_graph_build_equals_host_builder(seed, hubs):
  """graph.build_arrays_on_device (sort / scan ops, run here on CPU tensors) produces element for element the arrays
  of the C++ builder: CSR, stable permutation, CSC view, hub chunk lists, degree-class records, counts."""
  from gnpde_amd.graph import build_arrays_on_device
  n = 700
  ei = random_graph(n, 6, seed, hubs=hubs, hub_deg=1300, isolated=4, dup=15)
  host = G.CSRGraph(ei, n, device=torch.device('cpu'))
  arrays, counts = build_arrays_on_device(ei, n)
  for k in ('n_long_rows', 'n_long_chunks', 'n_long_cols', 'n_bin16', 'n_bin64', 'max_row_len', 'max_col_len'):
    assert counts[k] == getattr(host, k), k
  used = {'rowptr': n + 1, 'colidx': host.e, 'perm': host.e, 'rowidx': host.e, 'cscptr': n + 1, 'cscpos': host.e,
          'long_rows': host.n_long_rows, 'long_chunk_ptr': host.n_long_rows + 1, 'long_chunk_row': host.n_long_chunks,
          'long_chunk_begin': host.n_long_chunks, 'long_chunk_end': host.n_long_chunks, 'long_cols': host.n_long_cols,
          'bin_rows': 4 * (host.n_bin16 + host.n_bin64), 'long_chunk_first': host.n_long_chunks}
  for k, m in used.items():
    assert torch.equal(arrays[k][:m].cpu(), host.t[k][:m].cpu()), k
  empty, c0 = build_arrays_on_device(torch.zeros(2, 0, dtype=torch.long), 5)
  assert c0['n_bin16'] == 0 and empty['rowptr'].tolist() == [0] * 6
  with pytest.raises(G.GnpdeError):
    build_arrays_on_device(torch.tensor([[0], [9]]), 5)


def test_new_entry_points_validate_their_arguments():
  """Argument checks of the round-2 entry points run before anything touches a device: bad calls come back with an error
  code and a message (gnpde_last_error), size queries are consistent."""
  import ctypes
  L = _lib.lib()
  # two-hop densification
  small, big = L.gnpde_two_hop_workspace_bytes(1), L.gnpde_two_hop_workspace_bytes(200_000)
  assert 0 < small < big
  assert L.gnpde_two_hop_workspace_bytes(2_000_000) <= (17 << 30)        # slabs are capped at 16 GiB
  assert L.gnpde_two_hop_count(None, None, 10, None, None, 0, None) != 0
  assert b'two_hop_count' in L.gnpde_last_error()
  rowptr = (ctypes.c_int32 * 3)(0, 1, 2)
  col = (ctypes.c_int32 * 2)(1, 0)
  out = (ctypes.c_int64 * 3)()
  dummy = (ctypes.c_char * 64)()
  assert L.gnpde_two_hop_count(rowptr, col, 2, out, dummy, 64, None) != 0    # workspace too small
  assert b'workspace' in L.gnpde_last_error()
  # device-controlled dopri5
  assert L.gnpde_dopri5_workspace_bytes(None) == 0
  handle = ctypes.c_void_p()
  assert L.gnpde_dopri5_create(ctypes.byref(handle), None, 1e-7, 1e-9, None, 0) != 0 and not handle.value
  assert L.gnpde_dopri5_run(None, None, 0, 0.0, 1.0, None, 0, 1, 0, None, None) != 0
  assert L.gnpde_dopri5_stats(None, None, None, None, None, None) != 0
  assert L.gnpde_dopri5_destroy(None) == 0
  # decoder projection, quantile / compaction
  assert L.gnpde_relu_linear(None, 4, 4, 4, None, 4, 4, None, None, 4, None) != 0
  assert L.gnpde_quantile(None, 5, 0.5, None, None, 0, None) != 0
  assert L.gnpde_threshold_edges(None, None, 5, None, 0, 4, None, None, None, None, 0, None) != 0


def test_dropin_launcher_serves_the_reference_module_names(tmp_path):
  """python -m gnpde_amd.dropin SCRIPT: the script's imports of the reference's module names get this package's classes;
  `base_classes` is the script directory's own file with ODEFunc / ODEblock / RegularizedODEfunc replaced (no reference
  tree needed: a stand-in base_classes.py plays its part)."""
  import subprocess
  src = tmp_path / 'src'
  src.mkdir()
  (src / 'base_classes.py').write_text(
    'class ODEFunc(object):\n  pass\nclass ODEblock(object):\n  pass\nclass BaseGNN(object):\n  marker = 41\nREGISTRY = {"k": 1}\n')
  (src / 'run_it.py').write_text(
    'import sys\n'
    'from base_classes import BaseGNN, ODEFunc, ODEblock, RegularizedODEfunc, REGISTRY\n'
    'from block_constant import ConstantODEblock\n'
    'from function_transformer_attention import ODEFuncTransformerAtt, SpGraphTransAttentionLayer\n'
    'from early_stop_solver import EarlyStopInt, SOLVERS\n'
    'import gnpde_amd\n'
    'assert ODEFunc is gnpde_amd.base_classes.ODEFunc and issubclass(ODEFuncTransformerAtt, ODEFunc)\n'
    'assert ODEblock is gnpde_amd.base_classes.ODEblock and issubclass(ConstantODEblock, ODEblock)\n'
    'assert BaseGNN.marker == 41 and REGISTRY == {"k": 1} and BaseGNN.__module__ == "_reference_base_classes"\n'
    'assert "GNN" not in sys.modules or not sys.modules["GNN"].__name__.startswith("gnpde_amd")\n'
    'assert __name__ == "__main__" and sys.argv[1:] == ["--dataset", "Cora"]\n'
    'print("LAUNCHED_OK")\n')
  env = dict(os.environ, PYTHONPATH=ROOT)
  res = subprocess.run([sys.executable, '-m', 'gnpde_amd.dropin', str(src / 'run_it.py'), '--dataset', 'Cora'], env=env,
                       capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
  assert res.returncode == 0 and 'LAUNCHED_OK' in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
  res = subprocess.run([sys.executable, '-m', 'gnpde_amd.dropin', '--bogus', str(src / 'run_it.py')], env=env,
                       capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
  assert res.returncode != 0 and 'unknown option' in res.stderr


def test_dropin_install_refuses_after_a_foreign_import(tmp_path, monkeypatch):
  """install() after a reference module of the same name was already imported from elsewhere: refused, not half-applied."""
  import types
  from gnpde_amd import dropin
  assert not dropin.installed()
  monkeypatch.setitem(sys.modules, 'block_constant', types.ModuleType('block_constant'))
  with pytest.raises(ImportError):
    dropin.install()
  assert not dropin.installed()
  monkeypatch.delitem(sys.modules, 'block_constant')
  try:
    served = dropin.install(native_gnn=True)
    assert 'GNN' in served and sys.modules['block_mixed'] is G.block_mixed
    import base_classes                                   # no reference file on sys.path: this package's types alone
    assert base_classes.ODEblock is G.ODEblock and base_classes.__gnpde_reference__ is None
  finally:
    dropin.uninstall()
  assert not dropin.installed() and 'block_mixed' not in sys.modules and 'base_classes' not in sys.modules


def _xcd_map(row_begin, row_end, deal):
  shift, per = ctypes.c_int32(0), ctypes.c_int32(0)
  L = _lib.lib()
  _lib.check(L.gnpde_xcd_row_map(row_begin, row_end, deal, ctypes.byref(shift), ctypes.byref(per), None))
  m = np.full((8, max(per.value, 1)), -7, dtype=np.int32)
  _lib.check(L.gnpde_xcd_row_map(row_begin, row_end, deal, ctypes.byref(shift), ctypes.byref(per), m.ctypes.data))
  return shift.value, per.value, m[:, :per.value]


@pytest.mark.parametrize('contiguous', [False, True])
def test_xcd_row_map_takes_every_row_exactly_once(contiguous):
  """The rows -> XCD deals of the aggregation kernels (the same host + device function the kernels call): every row of
  [row_begin, row_end) exactly once whatever the size, lists padded with -1 only, even-aligned pairs consecutive (what
  spmm_pair_kernel relies on), hashed deal: blocks of consecutive rows."""
  deal = _lib.XCD_CONTIGUOUS if contiguous else _lib.XCD_HASHED
  for rb, re_ in [(0, 0), (0, 1), (0, 7), (0, 16), (3, 130), (0, 2708), (1000, 2708), (0, 169343), (54321, 169343),
                  (0, 2 ** 21), (5, 2 ** 21 + 77), (0, 40_000)]:
    shift, per, m = _xcd_map(rb, re_, deal)
    assert (shift < 0) == contiguous
    rows = m[m >= 0]
    assert rows.size == re_ - rb and np.array_equal(np.sort(rows), np.arange(rb, re_)), (rb, re_)
    assert np.all((m >= rb) | (m == -1))
    if per >= 2:
      a, b = m[:, 0:per - per % 2:2], m[:, 1:per - per % 2 + 1:2]
      both = (a >= 0) & (b >= 0)
      assert np.all(b[both] == a[both] + 1)
      if not contiguous:
        assert np.all(b[a < 0] < 0)                 # a pair never starts with a hole
    if not contiguous and re_ - rb > 0:
      assert 4 <= shift <= 7
      B = 1 << shift
      blk = m.reshape(8, -1, B)                     # blocks of consecutive rows (the last block of the range may be cut)
      head = blk[:, :, :1]
      assert np.all((blk < 0) | (blk == head + np.arange(B)))
  # the A/B knob overrides the argument (as it overrides gnpde_graph_t.xcd_deal in the launches)
  L = _lib.lib()
  try:
    _lib.check(L.gnpde_tune(_lib.TUNE_XCD_ROWS, 1))
    assert _xcd_map(0, 5000, _lib.XCD_HASHED)[0] < 0
    _lib.check(L.gnpde_tune(_lib.TUNE_XCD_ROWS, 2))
    assert _xcd_map(0, 5000, _lib.XCD_CONTIGUOUS)[0] >= 4
  finally:
    _lib.check(L.gnpde_tune(_lib.TUNE_XCD_ROWS, 0))
  with pytest.raises(G.GnpdeError):
    _xcd_map(0, 10, 7)


def test_xcd_deal_is_chosen_per_graph():
  """Row length that depends on the bits of the row id, as in an R-MAT graph (expected degree x 0.32 per set bit): contiguous
  eighths -- and any fixed round robin of blocks -- leave the XCDs 1.5 - 2x out of balance, the hashed deal within a few per cent (0.2 % at the R-MAT size); the graph
  builder measures the contiguous deal and picks the hashed one for such a graph, and keeps the contiguous one for a graph
  whose row lengths do not depend on the ids (both builders, and the narrowed row range of a partitioned graph)."""
  n = 1 << 15
  ids = np.arange(n)
  pop = np.zeros(n, dtype=np.int64)
  for b in range(15):
    pop += (ids >> b) & 1
  deg = np.minimum(np.maximum((600.0 * 0.316 ** pop).astype(np.int64), 1), 700)       # a few rows above GNPDE_LONG_ROW
  gen = np.random.default_rng(0)
  row = np.repeat(ids, deg)
  col = gen.integers(0, n, row.size)
  skewed = torch.from_numpy(np.stack([row, col]))
  flat = torch.from_numpy(np.stack([gen.permutation(n)[row], col]))                    # same lengths, ids shuffled
  work = np.where(deg <= 512, deg + 3.0, 0.0)

  def imbalance(deal):
    _, _, m = _xcd_map(0, n, deal)
    per_xcd = np.array([work[r[r >= 0]].sum() for r in m]) + deg[deg > 512].sum() / 8.0
    return per_xcd.max() / per_xcd.mean()
  assert imbalance(_lib.XCD_CONTIGUOUS) > 1.3 and imbalance(_lib.XCD_HASHED) < 1.06     # (2 048 blocks only at this size)
  g_skewed, g_flat = G.CSRGraph(skewed, n, device='cpu'), G.CSRGraph(flat, n, device='cpu')
  assert g_skewed.struct.xcd_deal == _lib.XCD_HASHED and g_flat.struct.xcd_deal == _lib.XCD_CONTIGUOUS
  assert abs(g_skewed.xcd_imbalance_contiguous - imbalance(_lib.XCD_CONTIGUOUS)) < 1e-9
  assert g_flat.xcd_imbalance_contiguous < 1.03
  # the device builder computes the same arrays with torch ops (here on the CPU) and has to reach the same decision
  from gnpde_amd.graph import build_arrays_on_device, contiguous_deal_imbalance
  arrays, _ = build_arrays_on_device(skewed, n)
  assert abs(contiguous_deal_imbalance(arrays['rowptr'], 0, n) - g_skewed.xcd_imbalance_contiguous) < 1e-9
  # a narrowed row range (boundary pass of a partitioned graph) is judged on ITS rows: the upper half of the ids alone is
  # still skewed, a range of 8 rows is not worth the arithmetic
  g_skewed.set_row_range(n // 2, n)
  assert g_skewed.struct.row_begin == n // 2 and g_skewed.struct.n == n and g_skewed.struct.xcd_deal == _lib.XCD_HASHED
  g_skewed.set_row_range(0, 0)
  assert g_skewed.struct.xcd_deal == _lib.XCD_CONTIGUOUS and g_skewed.xcd_imbalance_contiguous == 1.0
  with pytest.raises(G.GnpdeError):
    g_skewed.set_row_range(5, n + 1)


def test_row_records_of_the_wave_per_row_class_are_longest_first():
  """Rows of 17..512 entries in a heavy-tailed graph: both builders (host C++ and the torch ops of the device builder) list
  them by descending length, ties in row order; rows of 1..16 entries stay in row order; every row appears once."""
  from gnpde_amd.graph import build_arrays_on_device
  n = 4000
  gen = np.random.default_rng(3)
  deg = np.minimum((gen.pareto(1.2, n) * 6 + 1).astype(np.int64), 600)
  deg[:5] = 0
  row = np.repeat(np.arange(n), deg)
  ei = torch.from_numpy(np.stack([row, gen.integers(0, n, row.size)]))
  ei = ei[:, torch.from_numpy(gen.permutation(row.size))]
  g = G.CSRGraph(ei, n, device='cpu')
  rec = g.t['bin_rows'].numpy().reshape(-1, 4)[:g.n_bin16 + g.n_bin64]
  r16, r64 = rec[:g.n_bin16], rec[g.n_bin16:]
  assert np.all(np.diff(r16[:, 0]) > 0) and np.all((r16[:, 2] >= 1) & (r16[:, 2] <= 16))
  assert np.all((r64[:, 2] > 16) & (r64[:, 2] <= _lib.LONG_ROW)) and len(np.unique(r64[:, 2])) > 20
  assert np.all(np.diff(r64[:, 2]) <= 0)                                       # descending length ...
  same = np.diff(r64[:, 2]) == 0
  assert np.all(np.diff(r64[:, 0])[same] > 0)                                  # ... ties in row order
  assert np.array_equal(np.sort(rec[:, 0]), np.nonzero((deg >= 1) & (deg <= _lib.LONG_ROW))[0])
  assert np.array_equal(rec[:, 2], deg[rec[:, 0]]) and np.array_equal(rec[:, 1], g.rowptr.numpy()[rec[:, 0]])
  arrays, counts = build_arrays_on_device(ei, n)
  assert counts['n_bin16'] == g.n_bin16 and counts['n_bin64'] == g.n_bin64
  assert np.array_equal(arrays['bin_rows'].numpy()[:4 * len(rec)].reshape(-1, 4), rec)


def test_meter_matches_the_reference_meter():
  """gnpde_amd.Meter = src/utils.py:212-233 (what run_GNN.py's train() calls on model.fm / model.bm); the native BaseGNN
  carries both.  Compared with the reference's own class where the reference tree is present."""
  import gnpde_amd as G
  m = G.Meter()
  assert (m.val, m.sum, m.cnt, m.get_average(), m.get_value()) == (None, 0, 0, 0, None)
  for v in (16, 8, 0):
    m.update(v)
  assert (m.val, m.sum, m.cnt, m.get_average(), m.get_value()) == (0, 24, 3, 8.0, 0)
  m.reset()
  assert (m.val, m.sum, m.cnt) == (None, 0, 0)
  path = '/root/reference/src/utils.py'
  if os.path.exists(path):
    src = open(path).read()
    body = src[src.index('class Meter(object):'):src.index('class DummyDataset')]
    ns = {}
    exec(body, ns)
    ref = ns['Meter']()
    ours = G.Meter()
    for v in (3, 5.5, 0):
      ref.update(v)
      ours.update(v)
      assert vars(ref) == vars(ours) and ref.get_average() == ours.get_average() and ref.get_value() == ours.get_value()
  import inspect
  assert 'self.fm = Meter()' in inspect.getsource(G.BaseGNN.__init__) and 'self.bm = Meter()' in inspect.getsource(G.BaseGNN.__init__)


@pytest.mark.parametrize('parts', [2, 3, 8])
def test_partition_link_refinement_invariants(parts):
  """gnpde_partition_refine_links: the rows a rank RECEIVES (distinct referenced nodes per peer) are what an evaluation waits
  for.  The refinement must (i) report the same busiest link / total as an independent recount (distributed.pair_traffic),
  (ii) never raise the busiest link or the total, (iii) keep every part inside the partitioner's balance window, (iv) keep
  part ids valid, (v) be deterministic."""
  from gnpde_amd import distributed as D
  from gnpde_amd.graph import CSRGraph, partition_rows
  ei, n = G.synthetic.make_graph('arxiv', scale=0.05)
  ei2, _ = G.add_remaining_self_loops(ei, None, 1.0, n)
  g = CSRGraph(ei2, n, device='cpu')
  before = partition_rows(g, parts, refine_links=0).long()
  st = {}
  after = partition_rows(g, parts, refine_links=6, stats=st).long()
  again = partition_rows(g, parts, refine_links=6).long()
  assert torch.equal(after, again)
  assert int(after.min()) >= 0 and int(after.max()) < parts
  m0, m1 = D.pair_traffic(ei2, before, parts), D.pair_traffic(ei2, after, parts)
  assert st['max_link_before'] == int(m0.max()) and st['received_rows_before'] == int(m0.sum())
  assert st['max_link_after'] == int(m1.max()) and st['received_rows_after'] == int(m1.sum())
  assert int(m1.max()) <= int(m0.max()) and st['moves'] > 0
  assert int(m1.max()) < int(m0.max()) or int(m1.sum()) < int(m0.sum()), 'the refinement found nothing to improve on a community graph'
  deg = torch.bincount(ei2[0], minlength=n) + 1
  load = torch.zeros(parts, dtype=torch.long).index_add_(0, after, deg)
  load0 = torch.zeros(parts, dtype=torch.long).index_add_(0, before, deg)
  avg = float(deg.sum()) / parts
  hi = max(int(avg * 1.03) + int(deg.max()) // 8 + 1, int(load0.max()))
  lo = min(int(avg * 0.97) - int(deg.max()) // 8, int(load0.min()))
  assert int(load.max()) <= hi and int(load.min()) >= lo, (load.tolist(), lo, hi)


def test_partition_link_refinement_handles_directed_graphs_and_loops():
  """Self references never travel (a node's own row reads it locally); a directed graph's received rows follow the row ->
  column direction only."""
  from gnpde_amd import distributed as D
  from gnpde_amd.graph import CSRGraph, partition_rows
  g0 = torch.Generator().manual_seed(5)
  n = 400
  ei = torch.cat([torch.randint(0, n, (2, 2400), generator=g0), torch.arange(n).repeat(2, 1)], dim=1)   # directed + loops
  g = CSRGraph(ei, n, device='cpu')
  st = {}
  part = partition_rows(g, 4, refine_links=4, stats=st).long()
  m = D.pair_traffic(ei, part, 4)
  assert st['max_link_after'] == int(m.max()) and st['received_rows_after'] == int(m.sum())
  assert int(torch.diagonal(m).sum()) == 0


def test_locality_view_relabels_without_touching_the_order_inside_a_row():
  """graph.LocalityView (host parts of it: clustering, relabelled CSR): `order` is a permutation laid out part by part, the
  relabelled CSR holds the caller's edges in the caller's order inside every row (same `perm` semantics, so per-edge arrays
  and row sums carry over unchanged), and enter / leave are inverse permutations of the rows."""
  from gnpde_amd import synthetic
  n = 5000
  ei_np, _, _ = synthetic.community_powerlaw_graph(n, 30000, seed=2, n_comm=10)
  ei = torch.as_tensor(ei_np)
  base = G.CSRGraph(ei, n, device='cpu')
  assert base.locality_view(4 * 128) is None                 # automatic rule: small table (and no device to time on)
  view = base.locality_view(4 * 128, '1')
  assert view is base.locality_view(4 * 128, '1') and view.stats['n_parts'] == 8
  order, inv = view.order, view.inv
  assert torch.equal(torch.sort(order).values, torch.arange(n)) and torch.equal(inv[order], torch.arange(n))
  g = view.graph
  assert g.e == base.e and g.n == n
  rp_b, rp_v = base.rowptr.long(), g.rowptr.long()
  perm_b, perm_v = base.perm.long(), g.perm.long()
  col_b, col_v = base.colidx.long(), g.colidx.long()
  for i in range(0, n, 37):
    v = int(order[i])
    assert torch.equal(perm_v[rp_v[i]:rp_v[i + 1]], perm_b[rp_b[v]:rp_b[v + 1]])           # same edges, same order
    assert torch.equal(order[col_v[rp_v[i]:rp_v[i + 1]]], col_b[rp_b[v]:rp_b[v + 1]])       # pointing at the same nodes
  x = torch.randn(n, 7)
  assert torch.equal(view.leave(view.enter(x)), x) and torch.equal(view.enter(x)[inv], x)
  inside = view.stats['entries_inside_a_part']
  assert 0.3 < inside <= 1.0                                   # ten planted communities, 65 % of the edges inside one


def test_bench_headline_is_the_last_line_and_fits_the_drivers_tail(capsys):
  """The driver keeps an 8-KB tail of bench.py's stdout and parses its LAST line (round 5's single 26-KB line left `parsed: null`).  The
  line builder on a recorded full object + configs block: the last line is the headline, < 4 KB, with `roofline` and `cpu_baseline`."""
  import json
  import os
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  sys.path.insert(0, root)
  import bench
  full = json.loads(open(os.path.join(root, 'profiles', 'r05_bench_default_steps20.json')).read().strip().splitlines()[-1])
  configs = full.pop('configs')
  bench.emit(full, configs)
  lines = capsys.readouterr().out.splitlines()
  assert lines[0].startswith('{"bench_detail"') and lines[1].startswith('{"bench_configs"') and len(lines) == 3
  last = lines[-1]
  assert len(last) < bench.HEADLINE_MAX_BYTES
  head = json.loads(last)
  assert head['metric'] == full['metric'] and head['value'] == full['value'] and head['unit'] == 'steps/s'
  for key in ('n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data'):
    assert key in head
  assert 'workload' in head['config'] and 'model' not in head['config']
  for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'avg_launch_us', 'algorithmic_bytes_per_launch'):
    assert key in head['roofline']
  assert head['roofline']['frac'] == full['roofline']['frac']
  for key in ('value', 'unit', 'cores', 'kind', 'sample'):
    assert key in head['cpu_baseline']
  assert set(head['configs_summary']) == set(k for k in configs if not k.startswith('_'))
  # a pathologically large configs block is shed, never the contract keys
  big = dict(configs, **{'filler_%d' % i: {'value': 1.0, 'unit': 'x' * 40, 'roofline': None} for i in range(200)})
  head2 = bench.headline(full, big)
  assert len(json.dumps(head2)) < bench.HEADLINE_MAX_BYTES and 'roofline' in head2 and 'cpu_baseline' in head2


def test_bench_gpus_n_without_a_launcher_refuses_only_for_lack_of_devices(monkeypatch):
  """`python bench.py --gpus 8` the way the driver runs `--gpus 1` spawns its ranks itself (bench.launch_ranks); on a host with fewer
  devices it says why instead of dying at argument parsing."""
  import os
  import sys
  import pytest
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  sys.path.insert(0, root)
  import bench
  monkeypatch.delenv('GNPDE_RANKS_SHARE_DEVICE', raising=False)
  monkeypatch.setattr(bench.torch.cuda, 'device_count', lambda: 1)
  with pytest.raises(SystemExit) as err:
    bench.launch_ranks(8)
  assert 'GNPDE_RANKS_SHARE_DEVICE' in str(err.value)
  seen = {}

  class _Done(object):
    returncode = 0

  def fake_run(cmd, env=None, **kw):
    seen['cmd'], seen['env'] = cmd, env
    return _Done()
  import subprocess
  monkeypatch.setattr(subprocess, 'run', fake_run)
  monkeypatch.setattr(bench.torch.cuda, 'device_count', lambda: 8)
  monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '20', '--warmup', '5'])
  bench.launch_ranks(8)
  cmd = seen['cmd']
  assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '8'
  assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-6:] == ['--gpus', '8', '--steps', '20', '--warmup', '5']
  assert 'RANK' not in seen['env'] and seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
