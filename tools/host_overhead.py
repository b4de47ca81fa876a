"""Host-side cost of the Python->C-ABI launch path: per-op wrapper time and the 100-detection
inference path (host enqueue time vs device time)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import bench
from dynamask_amd import ops, synth
dev = torch.device('cuda')
head, sd = bench.build_head(dev)
feats_c, rois_c, labels_c = bench.make_inputs(0, dev)
feats = [f.to(dev) for f in feats_c]
rois, labels = rois_c.to(dev), labels_c.to(dev)

def host_time(fn, n=200):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6

x = torch.randn(4, 8, 14, 14, device=dev)
print('upsample2x wrapper      %.1f us/call' % host_time(lambda: ops.upsample2x(x)))
print('roi_align wrapper (8)   %.1f us/call' % host_time(lambda: ops.roi_align(feats[:4], rois[:8], 14, [1 / 4, 1 / 8, 1 / 16, 1 / 32])))
w = torch.randn(16, 8, 3, 3, device=dev); b = torch.randn(16, device=dev)
wq = ops.pack_conv_weight(w)
print('conv2d wrapper          %.1f us/call' % host_time(lambda: ops.conv2d(x, wq, b, 16, 3, relu=True)))
print('torch.empty             %.1f us/call' % host_time(lambda: torch.empty((4, 8, 14, 14), device=dev)))
det = rois[:100, 1:].contiguous(); dl = labels[:100]
for n in (100,):
    with torch.no_grad():
        h = host_time(lambda: head.simple_test_mask_logits(feats, det, dl), n=20)
        ms = bench.time_kernel(lambda: head.simple_test_mask_logits(feats, det, dl), iters=20, warmup=3)
    print('simple_test_mask_logits(%d dets): host enqueue %.0f us, device-timed %.0f us' % (n, h, ms * 1e3))
import cProfile, pstats
pr = cProfile.Profile()
with torch.no_grad():
    pr.enable()
    for _ in range(20):
        head.simple_test_mask_logits(feats, det, dl)
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(14)
