"""DCN backward pieces at the training step's shapes (256 RoIs): im2col, coordinate gradient + col2im, GB/s of
the column matrix they stream."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops
dev = torch.device('cuda')
def t(fn, iters=10, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
N = int(os.environ.get('DP_N', 256))
for C, S in ((256, 14), (128, 28), (64, 56)):
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(N, C, S, S, device=dev, generator=g)
    off = torch.randn(N, 36, S, S, device=dev, generator=g) * 0.7
    colgrad = torch.randn(N, 9 * C, S, S, device=dev, generator=g)
    gb = colgrad.numel() * 4 / 1e9
    ms_i = t(lambda: ops.deform_im2col(x, off, 2))
    ms_c = t(lambda: ops.deform_col2im_coord(colgrad, x, off, 2))
    print(f'C={C} {S}x{S} N={N}: column matrix {gb:.2f} GB | im2col {ms_i:.3f} ms ({gb / ms_i:.2f} TB/s written) | '
          f'coord+col2im {ms_c:.3f} ms ({2 * gb / ms_c:.2f} TB/s read)', flush=True)
