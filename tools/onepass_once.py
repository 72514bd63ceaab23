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
ops.tune(_lib.TUNE_ONE_PASS_VARIANT, int(sys.argv[1]) if len(sys.argv) > 1 else 3)
for _ in range(3): ops.attn_rhs_fused(graph, att, wqk, bqk, x, alpha, beta, x0, True, out=out)
torch.cuda.synchronize()
