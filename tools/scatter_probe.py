"""The DCN col2im scatter alone and the one-kernel data gradient alone at 256 x 64 x 56 x 56 (zero offsets), a few launches each:
the probe of tools/scatter_pmc.sh (SQ counters: what bounds the LDS scatter)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops
dev = torch.device('cuda')
N, C, S, dg = 256, 64, 56, 2
g = torch.Generator().manual_seed(3)
x = torch.randn(N, C, S, S, generator=g).to(dev)
off = torch.zeros(N, 18 * dg, S, S, device=dev)
go = torch.randn(N, C, S, S, generator=g).to(dev)
w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).to(dev)
cgr = torch.randn(N, 9 * C, S, S, generator=g).to(dev)
wf = ops.pack_dcn_bwd_weight(w, dg)
for _ in range(int(os.environ.get('PROBE_ITERS', '4'))):
    ops.deform_col2im(cgr, off, (N, C, S, S), dg)
    ops.deform_conv_backward_data_fused(x, off, go, wf, dg)
torch.cuda.synchronize()
print('done')
