"""gnpde_amd -- MI355X-native ODE right-hand side for GRAND / BLEND graph neural diffusion.

Same operator surface as the reference's src/ modules of the same names (ODEFunc / ODEblock and the
function_* / block_* classes); every arithmetic step of f(t,x) and of the fixed-step solver loop
runs in csrc/libgnpde_hip.so (hand-written gfx950 kernels, C ABI in include/gnpde.h)."""
from . import _lib
from ._lib import GnpdeError, build, lib
from .graph import CSRGraph, graph_of, partition_rows
from . import ops
from .utils import MaxNFEException, get_rw_adj, gcn_norm_fill_val, add_remaining_self_loops, DummyData, DummyDataset, Meter
from .odeint import odeint, odeint_adjoint, time_grid
from .base_classes import ODEFunc, ODEblock, RegularizedODEfunc, REGULARIZATION_FNS, create_regularization_fns
from .function_laplacian_diffusion import LaplacianODEFunc
from .function_transformer_attention import ODEFuncTransformerAtt, SpGraphTransAttentionLayer
from .function_GAT_attention import ODEFuncAtt, SpGraphAttentionLayer
from .block_constant import ConstantODEblock
from .block_transformer_attention import AttODEblock
from .block_mixed import MixedODEblock
from .block_transformer_hard_attention import HardAttODEblock
from .block_transformer_rewiring import RewireAttODEblock
from .early_stop_solver import EarlyStopInt, EarlyStopRK4, EarlyStopDopri5
from .model_configurations import set_block, set_function, BlockNotDefined, FunctionNotDefined
from .GNN import GNN, BaseGNN
from . import synthetic

__all__ = ['GnpdeError', 'build', 'lib', 'CSRGraph', 'graph_of', 'partition_rows', 'ops', 'MaxNFEException',
           'get_rw_adj', 'gcn_norm_fill_val', 'add_remaining_self_loops', 'odeint', 'odeint_adjoint', 'time_grid',
           'ODEFunc', 'ODEblock', 'LaplacianODEFunc', 'ODEFuncTransformerAtt', 'SpGraphTransAttentionLayer',
           'ODEFuncAtt', 'SpGraphAttentionLayer', 'ConstantODEblock', 'AttODEblock', 'MixedODEblock', 'HardAttODEblock', 'RewireAttODEblock', 'EarlyStopInt', 'EarlyStopRK4', 'EarlyStopDopri5', 'set_block', 'set_function',
           'synthetic']
