"""conv2d_wgrad vs an fp64 reference at the training step's shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops
torch.manual_seed(0)
for (NB, Co, Ci, S, k) in [(256, 16, 128, 28, 3), (4, 16, 128, 28, 3), (64, 16, 128, 28, 3), (256, 256, 256, 14, 3), (256, 36, 64, 56, 3),
                           (256, 36, 256, 14, 3), (256, 128, 256, 56, 1), (256, 30, 64, 56, 1), (256, 16, 128, 28, 1), (255, 16, 128, 28, 3), (256, 32, 128, 28, 3), (256, 16, 64, 28, 3)]:
    dy = torch.randn(NB, Co, S, S, device='cuda')
    x = torch.randn(NB, Ci, S, S, device='cuda')
    dw = ops.conv2d_wgrad(dy, x, k)
    ref = torch.nn.grad.conv2d_weight(x.double(), (Co, Ci, k, k), dy.double(), padding=k // 2)
    err = (dw.double() - ref).abs().max().item()
    print(f'NB {NB} Cout {Co} Cin {Ci} {S}x{S} k{k}: max|dw| {ref.abs().max().item():.3e} max err {err:.3e} rel {err / ref.abs().max().item():.2e}', flush=True)
