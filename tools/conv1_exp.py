"""1x1 implicit-GEMM tilings on the memory-bound shapes of the 56x56 stage (DM_CONV1_VARIANT is read at every call):
the DCN forward over the column matrix (576 -> 64), the column-gradient GEMM (64 -> 576), the fusion conv
([64, 64, 2] -> 64), a 64 -> 64 data gradient, the 28x28 counterparts."""
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


shapes = [('dcn fwd over col 576->64 @56 x128', [576], 64, 56, 128), ('colgrad 64->576 @56 x256', [64], 576, 56, 256),
          ('fuse [64,64,2]->64 @56 x128', [64, 64, 2], 64, 56, 128), ('dgrad 64->64 @56 x256', [64], 64, 56, 256),
          ('dcn fwd over col 1152->128 @28 x256', [1152], 128, 28, 256), ('colgrad 128->1152 @28 x256', [128], 1152, 28, 256),
          ('out1x1 64->30 @56 x128', [64], 30, 56, 128)]
for name, cins, cout, S, N in shapes:
    xs = [torch.randn(N, c, S, S, device=dev) for c in cins]
    w = torch.randn(cout, sum(cins), 1, 1, device=dev) / sum(cins) ** 0.5
    wq = ops.pack_conv_weight(w, src_channels=cins)
    out = torch.empty(N, cout, S, S, device=dev)
    fl = 2.0 * N * S * S * sum(cins) * cout
    mb = (sum(x.numel() for x in xs) + out.numel()) * 4 / 1e6
    base = None
    t(lambda: ops.conv2d(xs, wq, None, cout, 1, relu=True, out=out), iters=20)      # clocks up before the first variant is timed
    row = f'{name:40s} ({mb:6.0f} MB)'
    for v in ('0', '5', '1', '3', '4'):
        os.environ['DM_CONV1_VARIANT'] = v
        try:
            ms = t(lambda: ops.conv2d(xs, wq, None, cout, 1, relu=True, out=out))
        except RuntimeError as e:
            row += f' | v{v} n/a'
            continue
        ref = out.clone() if base is None else base
        base = ref
        row += f' | v{v} {ms:.3f} ms {fl / ms / 1e9:5.1f} TF/s {mb / ms / 1e3:4.2f} TB/s{"" if torch.equal(out, ref) else " BITS DIFFER"}'
    print(row, flush=True)
