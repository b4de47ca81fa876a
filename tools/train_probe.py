"""Training-step timing probe: per-step wall time (host) and event time (device) for a
few consecutive steps, to separate launch-side stalls from kernel time."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import bench
from dynamask_amd import synth
from dynamask_amd.dist import FlatParamGroup, mask_path_parameters
dev = torch.device('cuda')
head, sd = bench.build_head(dev)
B, per = 2, 128
feats = [f.to(dev) for f in synth.make_fpn(B, bench.IMG_H, bench.IMG_W, 256, seed=10)]
rois = synth.make_rois(B, per, bench.IMG_H, bench.IMG_W, seed=11).to(dev)
labels = synth.make_labels(B * per, seed=12).to(dev)
targets = [t.to(dev) for t in synth.make_targets(B * per, seed=13)]
noise = synth.make_gumbel_noise(B * per, seed=14).to(dev)
head.train()
grp = FlatParamGroup(mask_path_parameters(head))
if os.environ.get('TP_NO_DIRECT_GRAD'):          # A/B: gradients through autograd's AccumulateGrad as in round 1
    for p in grp.params:
        p._dm_direct_grad = False
for i in range(int(os.environ.get('TP_STEPS', 12))):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    grp.zero_grad()
    res = head._mask_forward_train(feats, rois, labels, targets, noise=noise)
    t1 = time.perf_counter()
    res['loss_mask']['loss_masks'].backward()
    t2 = time.perf_counter()
    grp.all_reduce_async()
    grp.sgd_step(lr=0.02, momentum=0.9, weight_decay=1e-4)
    e1.record()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f'step {i}: wall {1e3*(t4-t0):7.2f} ms  device {e0.elapsed_time(e1):7.2f} ms  host fwd {1e3*(t1-t0):6.2f} bwd {1e3*(t2-t1):6.2f} opt {1e3*(t3-t2):5.2f}  '
          f'mem {torch.cuda.memory_allocated()/2**30:.2f}/{torch.cuda.memory_reserved()/2**30:.2f} GiB', flush=True)

from dynamask_amd import ops as _ops
print(f'pack plan: {len(_ops.PACK_PLAN.entries)} packs, {_ops.PACK_PLAN.launches} batch launches, {_ops.PACK_PLAN.uploads} table uploads', flush=True)

# windows of 4 steps without a sync in between (what bench.py times): the host runs ahead of the GPU
if os.environ.get('TP_WINDOWS'):
    import gc
    def stats():
        s = torch.cuda.memory_stats()
        return s['num_device_alloc'], s['num_device_free'], s['reserved_bytes.all.current'] / 2**30, s['active_bytes.all.peak'] / 2**30
    from dynamask_amd import ops
    import contextlib
    hint = ops.overlapped_streams() if os.environ.get('TP_OVERLAPPED') else contextlib.nullcontext()
    hint.__enter__()
    if os.environ.get('TP_HIGH_PRIO'):       # the whole step on a high-priority stream (the side streams keep the default)
        print('priority range', torch.cuda.Stream.priority_range(), flush=True)
        hp = torch.cuda.Stream(priority=int(os.environ['TP_HIGH_PRIO']))
        hp.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(hp)
    for w in range(int(os.environ['TP_WINDOWS'])):
        torch.cuda.synchronize()
        a0 = stats(); g0 = gc.get_count()
        t0 = time.perf_counter()
        for _ in range(4):
            grp.zero_grad()
            res = head._mask_forward_train(feats, rois, labels, targets, noise=noise)
            res['loss_mask']['loss_masks'].backward()
            grp.all_reduce_async()
            grp.sgd_step(lr=0.02, momentum=0.9, weight_decay=1e-4)
        th = time.perf_counter()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        a1 = stats()
        print(f'window {w}: {(t1 - t0) / 4 * 1e3:6.2f} ms/step (host issue {(th - t0) / 4 * 1e3:6.2f})  device allocs +{a1[0] - a0[0]} frees +{a1[1] - a0[1]}  '
              f'reserved {a1[2]:.2f} GiB peak active {a1[3]:.2f} GiB  gc {g0}', flush=True)

# A/B of a Python-level switch inside one process: TP_AB=ops.WGRAD_CAT alternates windows of 8 steps with the flag
# (a one-element list in dynamask_amd.<module>) True and False, and prints the median of each side
if os.environ.get('TP_AB'):
    import importlib, statistics
    modname, attr = os.environ['TP_AB'].rsplit('.', 1)
    flag = getattr(importlib.import_module('dynamask_amd.' + modname), attr)
    def window(k=8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            grp.zero_grad()
            res = head._mask_forward_train(feats, rois, labels, targets, noise=noise)
            res['loss_mask']['loss_masks'].backward()
            grp.all_reduce_async()
            grp.sgd_step(lr=0.02, momentum=0.9, weight_decay=1e-4)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e3
    sides = {True: [], False: []}
    for r in range(int(os.environ.get('TP_AB_ROUNDS', 6))):
        for v in (True, False):
            flag[0] = v
            window(2)
            sides[v].append(window())
    for v in (True, False):
        print(f'{os.environ["TP_AB"]}={v}: median {statistics.median(sides[v]):.3f} ms/step  ({" ".join(f"{x:.2f}" for x in sides[v])})', flush=True)
    flag[0] = True
