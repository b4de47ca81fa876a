"""The mask head at the RoI counts of real inference (a few to 100 detections): time to the 28x28 exit and to
112x112, eager and as a replayed HIP graph, plus the big kernels alone.  Launches that do not fill the
chip cost a lone workgroup's time however small the batch."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import bench
from dynamask_amd import ops
from kbench import t
dev = torch.device('cuda')
head, sd = bench.build_head(dev)
feats_c, rois_c, labels_c = bench.make_inputs(0, dev)
feats = [f.to(dev) for f in feats_c]; rois = rois_c.to(dev); labels = labels_c.to(dev)
def graphed(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return t(g.replay, iters=30)
w3 = ops.pack_conv_weight(torch.randn(256, 256, 3, 3, device=dev) / 48)
for n in [int(v) for v in sys.argv[1:]] or (8, 16, 32, 64, 100):
    r, l = rois[:n].contiguous(), labels[:n].contiguous()
    with torch.no_grad():
        a = graphed(lambda: head._mask_forward(feats, r, l, last_stage=1))
        b = graphed(lambda: head._mask_forward(feats, r, l))
    row = f'N={n:4d}: exit28 {a:.3f} ms  full112 {b:.3f} ms |'
    for C, S in ((256, 14), (128, 28), (64, 56)):
        x = torch.randn(n, C, S, S, device=dev); off = torch.randn(n, 36, S, S, device=dev)
        wq = ops.pack_conv_weight(torch.randn(C, C, 3, 3, device=dev) / (9 * C) ** 0.5)
        row += f' dcn{S} {t(lambda: ops.deform_conv(x, off, wq, C, 2, relu=True)):.3f}'
        w1 = ops.pack_conv_weight(torch.randn(C, 2 * C + 2, 1, 1, device=dev), src_channels=[C, C, 2])
        srcs = [x, torch.randn_like(x), torch.randn(n, 2, S, S, device=dev)]
        row += f' fuse{S} {t(lambda: ops.conv2d(srcs, w1, None, C, 1, relu=True)):.3f}'
        wo = ops.pack_conv_weight(torch.randn(36, C, 3, 3, device=dev) / (9 * C) ** 0.5)
        row += f' off{S} {t(lambda: ops.conv2d(x, wo, None, 36, 3)):.3f}'
    x = torch.randn(n, 256, 14, 14, device=dev)
    row += f' conv14 {t(lambda: ops.conv2d(x, w3, None, 256, 3, relu=True)):.3f}'
    print(row, flush=True)
