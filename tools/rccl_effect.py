"""Does an initialised RCCL communicator change kernel timings?  RoIAlign14 graph replay
before / after init_process_group('nccl', world 1) / after a forced all-reduce."""
import os, socket, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist
from dynamask_amd import ops, synth
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
feats = [f.to(dev) for f in synth.make_fpn(1, 800, 1333, 256, seed=0)]
rois = synth.make_rois(1, 512, 800, 1333, seed=1).to(dev)
call = lambda: ops.roi_align(feats[:4], rois, 14, [1 / 4, 1 / 8, 1 / 16, 1 / 32])


def t(fn, iters=5, warmup=2):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def graphed(tag):
    call(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            call()
    print(f'{tag}: graph {t(g.replay) / 20 * 1e3:.1f} us, eager {t(call, 30, 5) * 1e3:.1f} us', flush=True)
    return g


g0 = graphed('before rccl')
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
print(f'old graph after init: {t(g0.replay) / 20 * 1e3:.1f} us')
graphed('after init')
x = torch.zeros(4162462, device=dev)
dist.all_reduce(x); torch.cuda.synchronize()
print(f'old graph after all_reduce: {t(g0.replay) / 20 * 1e3:.1f} us')
graphed('after all_reduce')
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    w = dist.all_reduce(x, async_op=True)
w.wait(); torch.cuda.synchronize()
graphed('after side-stream all_reduce')
dist.destroy_process_group()
graphed('after destroy')
