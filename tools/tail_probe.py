"""How much of a conv3x3 / DCN launch at 14x14 is the last, nearly empty round of workgroups?
Times the kernels at RoI counts around the 512 of the benchmark: 501 RoIs = 1536 workgroups =
exactly 3 per slot (256 CUs x 2), 512 RoIs = 1568 = 3.06.  DM_CONV_TAIL=0 disables the
small-tile tail launch of the conv (read once per process)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops
from kbench import t
dev = torch.device('cuda')
w = torch.randn(256, 256, 3, 3, device=dev) / 48
b = torch.randn(256, device=dev)
wq = ops.pack_conv_weight(w)
ns = [int(v) for v in sys.argv[1:]] or [334, 336, 400, 480, 501, 502, 512, 640, 668, 670, 768, 1002, 1024]
xs = torch.randn(max(ns), 256, 14, 14, device=dev)
offs = torch.randn(max(ns), 36, 14, 14, device=dev)
ref = ops.conv2d(xs[:64], wq, b, 256, 3, relu=True)      # 64 RoIs: one launch of 128 x 128 tiles
for n in ns:
    x, off = xs[:n], offs[:n]
    ms = t(lambda: ops.conv2d(x, wq, b, 256, 3, relu=True), iters=20, warmup=3)
    md = t(lambda: ops.deform_conv(x, off, wq, 256, 2, relu=True), iters=20, warmup=3)
    y = ops.conv2d(x, wq, b, 256, 3, relu=True)
    k = min(n, 64)
    same = torch.equal(y[:k], ref[:k]) and torch.equal(y[n - 8:], ops.conv2d(x[n - 8:], wq, b, 256, 3, relu=True))
    wgs = 2 * ((n * 196 + 127) // 128)
    print(f'N={n:5d} wgs={wgs:5d} ({wgs / 512:5.2f} rounds)  conv {ms:6.3f} ms = {ms / n * 1e3:6.3f} us/RoI  bits {"same" if same else "DIFFER"}   dcn {md:6.3f} ms = {md / n * 1e3:6.3f} us/RoI')
