"""The three DCN forward kernels alone for rocprofv3 --pmc SQ_* passes (see sq_probe.py / sq_pmc.sh)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops
dev = torch.device('cuda')
torch.manual_seed(0)
it = int(os.environ.get('PROBE_ITERS', '4'))
for C, S in ((256, 14), (128, 28), (64, 56)):
    N = 512
    x = torch.randn(N, C, S, S, device=dev)
    off = torch.randn(N, 36, S, S, device=dev) * 0.5
    w = torch.randn(C, C, 3, 3, device=dev) / (9 * C) ** 0.5
    wq = ops.pack_conv_weight(w)
    for _ in range(it):
        ops.deform_conv(x, off, wq, C, 2, relu=True)
    torch.cuda.synchronize()
print('done')
