"""A/B timing of one-pass kernel variants (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnpde_amd as G
from gnpde_amd import ops, _lib
dev = torch.device('cuda:0')
ei, n = G.synthetic.make_graph('arxiv'); d, A, h = 128, 16, 4
ei2, _ = G.add_remaining_self_loops(ei, None, 1.0, n)
graph = G.CSRGraph(ei2.to(dev), n)
x = torch.randn(n, d, device=dev); x0 = torch.randn_like(x); out = torch.empty_like(x)
wqk = (torch.randn(2 * A, d, device=dev) / d ** 0.5).contiguous(); bqk = torch.zeros(2 * A, device=dev)
att = ops.attention_struct(_lib.ATT_SCALED_DOT, h, A, 0, False)
alpha, beta = torch.tensor([0.0], device=dev), torch.tensor([0.1], device=dev)
for v in [0, 1, 2, 3, 4]:
  ops.tune(_lib.TUNE_ONE_PASS_VARIANT, v)
  for bpc in ([0] if v else [0, 2, 4]):
    ops.tune(_lib.TUNE_FUSED_BLOCKS_PER_CU, bpc)
    for _ in range(2): ops.attn_rhs_fused(graph, att, wqk, bqk, x, alpha, beta, x0, True, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.attn_rhs_fused(graph, att, wqk, bqk, x, alpha, beta, x0, True, out=out)
    e1.record(); torch.cuda.synchronize()
    print('variant', v, 'blocks/cu', bpc, '%.1f us' % (e0.elapsed_time(e1) * 100), flush=True)
