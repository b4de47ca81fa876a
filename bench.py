#!/usr/bin/env python
"""Benchmark of the DynaMask mask-head hot path on MI355X.

Workload (BASELINE.json configs[1], SURVEY section 8d reading (ii)): one
1333x800 image (FPN P2..P6 of R-50-FPN shape), 512 RoIs, fixed 28x28 mask exit
of DynaMaskHead: RoIAlign 14x14 over P2..P5 -> 2 x conv3x3 -> SFM stage 0
(semantic 1x1 on P4, point sample, class logits, fuse 1x1, DCN 3x3, 1x1, x2
upsample) -> stage-1 class logits at 28x28.  A "step" is one such RoI batch;
inputs are resident in HBM before the timed region.

One JSON line on stdout (rank 0).  `value` = images (RoI batches) per second
over all ranks; `ms_per_step` = ms per RoI batch; `roofline` prices the
dominant kernel (conv3x3 256->256 implicit GEMM, fp32 MFMA) and
`roofline_roialign` the RoIAlign kernel, both timed live with events on the
launch stream; `cpu_baseline` times the oracle (CPU restatement of the
reference) on a bounded sample of the same RoIs on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

IMG_H, IMG_W = 800, 1333
ROIS_PER_IMG = 512
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec


def build_head(dev):
    from dynamask_amd import losses, mask_heads, registry, roi_extractors, roi_head, synth  # noqa: F401
    cfg = dict(type='DynaMaskRoIHead',
               mask_roi_extractor=dict(type='SingleRoIExtractor', **synth.MASK_ROI_EXTRACTOR_CFG),
               mask_head=dict(type='DynaMaskHead', **synth.MASK_HEAD_CFG))
    m = registry.build_head(cfg)
    sd = {**synth.init_dynamask_head_state(seed=5, test_mode=True), **synth.init_mask_pre_state(seed=6)}
    m.load_state_dict(sd, strict=True)
    return m.to(dev).eval(), sd


def make_inputs(rank, dev):
    from dynamask_amd import synth
    feats = synth.make_fpn(1, IMG_H, IMG_W, 256, seed=0 + 1000 * rank)
    rois = synth.make_rois(1, ROIS_PER_IMG, IMG_H, IMG_W, seed=1 + 1000 * rank)
    labels = synth.make_labels(ROIS_PER_IMG, seed=2 + 1000 * rank)
    return feats, rois, labels


def bbox2roi_(boxes):
    from dynamask_amd.roi_head import bbox2roi
    return bbox2roi(boxes).contiguous()


def time_kernel(fn, iters=20, warmup=3):
    """Average duration (ms) of `fn` (launches on torch's current stream)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def time_kernel_median(fn, iters=7, warmup=2):
    """Median duration (ms) of `fn` over `iters` individually timed calls: for the multi-kernel
    extras, where one allocator refill inside a 5-call average would triple the figure."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for e0, e1 in ev:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    return sorted(e0.elapsed_time(e1) for e0, e1 in ev)[iters // 2]


def time_kernel_graphed(fn, reps=20, iters=5, warmup=2):
    """Average duration (ms) of one `fn` when `reps` of them are replayed back to back as a
    HIP graph: for kernels shorter than the Python/ctypes call that launches them (RoIAlign:
    ~70 us of kernel behind ~80 us of wrapper) event timing of eager launches measures the
    host.  Falls back to eager timing if capture is unavailable."""
    try:
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        # median over individually timed replays: one disturbed replay (seen once: 4x) must not
        # set the figure
        return time_kernel_median(g.replay, iters=max(iters, 7), warmup=warmup) / reps
    except Exception as e:      # noqa: BLE001
        print(f'[bench] graph capture failed ({e}); eager kernel timing', file=sys.stderr)
        return time_kernel(fn)


def time_kernel_single_in_graph(fn, iters=15):
    """Duration (ms) of ONE `fn` launch: replay time of a HIP graph holding two launches minus that of a graph holding
    one (median of `iters` individually timed replays each) -- the second launch's time with exactly one predecessor
    instead of nineteen.  (Event-record nodes between the launches would time it directly; on this ROCm torch refuses
    timing events inside a capture and hipEventElapsedTime rejects events recorded by a graph.)"""
    try:
        fn()
        torch.cuda.synchronize()
        t = []
        for reps in (1, 2):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(reps):
                    fn()
            t.append(time_kernel_median(g.replay, iters=iters, warmup=4))
        return max(t[1] - t[0], 0.0)
    except Exception as e:      # noqa: BLE001
        print(f'[bench] single-launch timing unavailable: {e}', file=sys.stderr)
        return None


def time_kernel_cold(fn, iters=9, evict_mib=512, dirty=True):
    """Duration (ms) of ONE `fn` launch from COLD caches: before every timed launch a read-modify-write sweep over
    `evict_mib` MiB of another buffer pushes the feature maps and the previous output out of the 256 MiB Infinity Cache and
    the L2s; the events bracket only the launch (a one-launch HIP graph, enqueued while the sweep is still running, so no
    host latency sits between the events).  Median of `iters`.  ``dirty=False``: the sweep only READS its buffer (a sum), so
    the lines the timed launch displaces are clean -- with the read-modify-write sweep every line the launch allocates in
    the Infinity Cache first sends a modified line of the sweep back to HBM, traffic that is the evictor's, not the launch's."""
    try:
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        sweep = torch.zeros(evict_mib * (1 << 20) // 4, device='cuda', dtype=torch.float32)
        ts = []
        for i in range(iters + 2):
            if dirty:
                sweep.add_(1.0)
            else:
                sink = sweep.sum()      # noqa: F841
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            if i >= 2:
                ts.append(e0.elapsed_time(e1))
        del sweep
        return sorted(ts)[len(ts) // 2]
    except Exception as e:      # noqa: BLE001
        print(f'[bench] cold timing unavailable: {e}', file=sys.stderr)
        return None


def roialign_algorithmic_bytes(rois, levels, feat_shapes, C=256, P=14):
    """output write + rois + per-RoI footprint read (capped per level), overall read capped
    by the size of the levels touched.  ``levels`` = the FPN level of each RoI as the HIP kernel itself
    reports it (levels_out of dm_roi_align_fwd): the denominator does not depend on the checker."""
    N = rois.shape[0]
    write = N * C * P * P * 4 + N * 20
    strides = (4, 8, 16, 32)
    read = 0
    touched = 0
    for l in range(4):
        sel = rois[levels == l]
        if len(sel) == 0:
            continue
        H, W = feat_shapes[l]
        touched += C * 4 * H * W
        w = torch.ceil((sel[:, 3] - sel[:, 1]) / strides[l]) + 2
        h = torch.ceil((sel[:, 4] - sel[:, 2]) / strides[l]) + 2
        read += int((C * 4 * torch.minimum(w * h, torch.tensor(float(H * W)))).sum().item())
    return write + min(read, touched)


def host_core_share():
    """Cores this process may actually use: min(affinity, cgroup CPU quota).  Running the
    oracle with more threads than that is slower, not faster (measured: 128 threads on a
    16-core quota = 10x slower than 16 threads)."""
    n = len(os.sched_getaffinity(0))
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(sd, feats, rois, labels, sample, reps=2):
    from oracle import ref_model
    n_threads = torch.get_num_threads()
    r, l = rois[:sample].contiguous(), labels[:sample].contiguous()
    kw = dict(stage_sup_size=(14, 28, 56, 112))

    def run():
        # fixed 28x28 exit: stage 0 in full, stage-1 logits only (the oracle computes all
        # of stage 1 as the reference does; only its head-of-stage logits are needed, so
        # time the restatement of exactly the measured work)
        import torch.nn.functional as F
        from oracle import ref_ops
        ins = ref_ops.single_roi_extractor(feats[:4], r, 14, (4, 8, 16, 32))
        x = ins
        for i in range(2):
            x = F.relu(F.conv2d(x, sd[f'mask_head.instance_convs.{i}.conv.weight'],
                                sd[f'mask_head.instance_convs.{i}.conv.bias'], padding=1))
        ip0, dp0, x = ref_model.sfm_stage(sd, 'mask_head.stages.0.', x, feats[-3], r, l, 14, 0.25, True)
        ar = torch.arange(len(r))
        ip1 = F.conv2d(x, sd['mask_head.stages.1.instance_logits.weight'], sd['mask_head.stages.1.instance_logits.bias'])[ar, l]
        dp1 = F.conv2d(x, sd['mask_head.stages.1.detail_logits.weight'], sd['mask_head.stages.1.detail_logits.bias'])[ar, l]
        return ip1, dp1
    with torch.no_grad():
        run()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = run()
        dt = (time.perf_counter() - t0) / reps
    return dt, n_threads, out


def _max_over_ranks(x, world):
    """Max of a host float over the ranks (default group = gloo at N > 1: a CPU tensor)."""
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def train_step_bench(head, dev, rank, world, steps=4, warmup=4, rehearsal=False):
    """BASELINE configs[2]/[3]: training step of the mask path, 2 images/GPU x 128
    positive RoIs, dynamic 14/28/56/112 selection + BCE backward + RCCL all-reduce of
    the flat mask-head gradient + fused SGD.  Returns ms per step (max over ranks).

    The RCCL communicator exists only inside this function: at N > 1 the default process group
    is gloo (host barriers and the max over ranks) and the data-path collective gets an `nccl`
    subgroup that is destroyed again before the function returns, so that the headline leg runs
    without an idle communicator in the process at every N (its watchdog thread costs eager
    multi-stream launch sequences ~10 %)."""
    from dynamask_amd import synth
    from dynamask_amd.dist import FlatParamGroup, mask_path_parameters
    import torch.distributed as dist
    B, per = 2, 128
    feats = [f.to(dev) for f in synth.make_fpn(B, IMG_H, IMG_W, 256, seed=10 + 1000 * rank)]
    rois = synth.make_rois(B, per, IMG_H, IMG_W, seed=11 + 1000 * rank).to(dev)
    labels = synth.make_labels(B * per, seed=12 + 1000 * rank).to(dev)
    targets = [t.to(dev) for t in synth.make_targets(B * per, seed=13 + 1000 * rank)]
    noise = synth.make_gumbel_noise(B * per, seed=14 + 1000 * rank).to(dev)
    head.train()
    sub = None
    if world > 1 and not rehearsal:
        sub = dist.new_group(backend='nccl', device_id=dev)
    grp = FlatParamGroup(mask_path_parameters(head), process_group=sub)
    saved = grp.flat_param.clone()          # the SGD steps below must not leak into later legs

    reduce_on = [True]

    def step():
        grp.zero_grad()
        res = head._mask_forward_train(feats, rois, labels, targets, noise=noise)
        res['loss_mask']['loss_masks'].backward()
        if reduce_on[0]:
            grp.all_reduce_async(force=True)        # world 1: nothing to reduce unless a one-rank communicator exists (below)
        grp.sgd_step(lr=0.02, momentum=0.9, weight_decay=1e-4)
        return res

    for _ in range(warmup):
        step()

    issue = []          # host time to issue a step's launches (ms): the step is close to host-bound

    def window():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = step()
        issue.append((time.perf_counter() - t0) / steps * 1e3)
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()
        w = _max_over_ranks(time.perf_counter() - t0, world)
        return w, r
    # median of three windows of `steps` steps: a caching-allocator growth (hipMalloc) inside one
    # window otherwise shows up as a 20-30 % outlier of this secondary figure
    wins = [window() for _ in range(3)]
    res = wins[-1][1]
    dt = sorted(w for w, _ in wins)[1]
    print('[bench] train-step windows (ms/step): ' + ', '.join(f'{w / steps * 1e3:.2f}' for w, _ in wins)
          + '; host issue ' + ', '.join(f'{v:.2f}' for v in issue), file=sys.stderr)
    # communication alone: the flat-gradient all-reduce (16.65 MB) timed by itself, so that
    # the scaling curve can be read with and without it; 0 at world size 1
    comm_ms = 0.0
    forced_ms = None
    own_group = False
    # what the collective costs the STEP (the curve with the communication isolated): one more window of
    # the same steps without the all-reduce (the ranks then drift apart -- timing only, parameters are restored below);
    # exposed = step with - step without.  At world 1 the pair is (forced one-rank all-reduce, plain step), below.
    noreduce_ms = None
    if world > 1:
        reduce_on[0] = False
        step()
        noreduce_ms = window()[0] / steps * 1e3
        reduce_on[0] = True
    if world == 1 and not dist.is_initialized() and os.environ.get('DM_BENCH_NO_RCCL', '0') != '1':
        # World size 1: the step above ran without a collective (a single-GPU job has nothing to reduce: this
        # is the N = 1 point of the scaling curve).  So that the RCCL call, its side stream and the 1/world
        # scaling of the fused SGD step still run on hardware in EVERY round, a one-rank communicator is
        # created now, one window is timed with the all-reduce forced through it, and it is torn down again
        # (left alive, its watchdog thread slowed eager multi-stream launch sequences by 10 %).
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(_free_port()))
        try:
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
            own_group = True
            step()
            forced_ms = window()[0] / steps * 1e3
        except Exception as e:      # noqa: BLE001  (recorded in the JSON; the headline needs no collective)
            print(f'[bench] RCCL world-1 group unavailable: {e}', file=sys.stderr)
    if dist.is_initialized():
        for _ in range(2):
            grp.all_reduce_async(force=True); grp.wait()
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            grp.all_reduce_async(force=True); grp.wait()
        torch.cuda.synchronize()
        comm_ms = _max_over_ranks((time.perf_counter() - t0) / 5 * 1e3, world)
    from dynamask_amd import ops
    grp.flat_param.copy_(saved)
    ops.WEIGHT_EPOCH[0] += 1                # packed-weight caches follow the restored parameters
    head.eval()
    collective = (f'{dist.get_backend(sub)} all-reduce over {world} rank(s), executed' if dist.is_initialized()
                  else 'none (no process group)')
    if own_group:
        torch.cuda.synchronize()
        dist.destroy_process_group()
    if sub is not None:
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group(sub)
    step_ms = dt / steps * 1e3
    if world > 1:
        exposed_ms = step_ms - noreduce_ms
    else:
        exposed_ms = (forced_ms - step_ms) if forced_ms is not None else None
    return step_ms, float(res['loss_mask']['loss_masks'].detach()), grp.numel, B, comm_ms, collective, forced_ms, exposed_ms


def count_host_syncs(fn):
    """Number of synchronising calls `fn` makes on the host (torch's sync debug mode warns on every .item() / .tolist()
    / .cpu() / blocking copy / explicit synchronize of the CURRENT device that waits for the GPU); None if unavailable."""
    import warnings
    try:
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode('warn')
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            fn()
        return sum(1 for x in w if 'synchronizing' in str(x.message).lower())      # (not the mode's own "prototype feature" notice)
    except Exception as e:      # noqa: BLE001
        print(f'[bench] sync counting unavailable: {e}', file=sys.stderr)
        return None
    finally:
        torch.cuda.set_sync_debug_mode('default')


def entry_points_bench(sd, dev, iters=7):
    """The entry points the reference's detector actually calls, at the reference's shapes, each as a whole (rows f1, f2, f4
    of SURVEY 8f beside the path): ``DynaMaskRoIHead.forward_train`` (two_stage.py:161-164 -> dynamask_roi_head.py:21-46:
    assigner + sampler per image, bbox branch, device mask targets, mask path, losses) + backward; the mask targets alone
    (dynamask_head.py:246-271); paste + threshold + RLE of 100 detections (dynamask_head.py:279-342 + core/mask/utils.py:36-63)."""
    from dynamask_amd import bbox_heads, registry, synth  # noqa: F401
    from dynamask_amd.registry import ConfigDict
    B = 2
    rh = registry.build_head(dict(
        type='DynaMaskRoIHead',
        bbox_roi_extractor=dict(type='SingleRoIExtractor', **synth.BBOX_ROI_EXTRACTOR_CFG),
        bbox_head=dict(type='Shared2FCBBoxHead', **synth.BBOX_HEAD_CFG),
        mask_roi_extractor=dict(type='SingleRoIExtractor', **synth.MASK_ROI_EXTRACTOR_CFG),
        mask_head=dict(type='DynaMaskHead', **synth.MASK_HEAD_CFG),
        train_cfg=registry._to_cfgdict(synth.RCNN_TRAIN_CFG), test_cfg=ConfigDict(**synth.RCNN_TEST_CFG)))
    rh.load_state_dict({**sd, **synth.init_bbox_head_state(seed=8)}, strict=True)
    rh = rh.to(dev).train()
    feats = [f.to(dev) for f in synth.make_fpn(B, IMG_H, IMG_W, 256, seed=40)]
    tb = synth.make_train_batch(B, IMG_H, IMG_W, seed=41)
    props = [p.to(dev) for p in tb['proposals']]
    gtb = [t.to(dev) for t in tb['gt_bboxes']]
    gtl = [t.to(dev) for t in tb['gt_labels']]
    gtm = [t.to(dev) for t in tb['gt_masks']]
    kept = {}

    def ft():
        for p_ in rh.parameters():
            p_.grad = None
        losses = rh.forward_train(feats, tb['img_metas'], props, gtb, gtl, None, gtm)
        sum(v for k, v in losses.items() if 'loss' in k).backward()
        kept['losses'] = losses
    for _ in range(3):
        ft()
    out = {'forward_train_ms': time_kernel_median(ft, iters=iters, warmup=1),
           'forward_train_host_syncs': count_host_syncs(ft)}
    # the mask targets alone, for the positives the sampler of the last call kept (<= 128 per image, four sizes)
    with torch.no_grad():
        srs = []
        for i in range(B):
            ar = rh.bbox_assigner.assign(props[i], gtb[i], None, gtl[i])
            srs.append(rh.bbox_sampler.sample(ar, props[i], gtb[i], gtl[i]))
        pb, pi = [r.pos_bboxes for r in srs], [r.pos_assigned_gt_inds for r in srs]
        gt_call = lambda: rh.mask_head.get_targets(pb, pi, gtm)      # noqa: E731
        out['get_targets_ms'] = time_kernel_median(gt_call, iters=iters, warmup=2)
        out['get_targets_host_syncs'] = count_host_syncs(gt_call)
        n_pos = sum(int(b_.shape[0]) for b_ in pb)
        # paste + threshold + RLE of 100 detections at 112 x 112 into the 800 x 1333 canvas; the RLE strings on the host
        rois = synth.make_rois(1, 100, IMG_H, IMG_W, seed=42).to(dev)
        det = torch.cat([rois[:, 1:], torch.ones(100, 1, device=dev)], 1)
        logits = ((synth.make_targets(100, sizes=(112,), seed=43)[0] * 2 - 1) * 3).unsqueeze(1).to(dev)
        lab = torch.zeros(100, dtype=torch.long, device=dev)
        cfg = ConfigDict(mask_thr_binary=0.5)
        rle_call = lambda: rh.mask_head.get_seg_rles(logits, det, lab, cfg, (IMG_H, IMG_W, 3), 1.0, True)      # noqa: E731
        bit_call = lambda: rh.mask_head.get_seg_masks(logits, det, lab, cfg, (IMG_H, IMG_W, 3), 1.0, True)     # noqa: E731
        out['paste_rle_ms'] = time_kernel_median(rle_call, iters=iters, warmup=2)
        out['paste_rle_host_syncs'] = count_host_syncs(rle_call)
        out['paste_bitmaps_ms'] = time_kernel_median(bit_call, iters=iters, warmup=2)
    out['what'] = (f'forward_train: {B} images x 1000 proposals (15 / 7 GT boxes with 800x1333 bitmaps), MaxIoUAssigner + RandomSampler(512, 0.25) '
                   f'-> {n_pos} positives, RoIAlign7 + Shared2FC + CE / L1 losses, device mask targets at 14/28/56/112, RoIAlign14 + '
                   'RoIAlign56 + MaskPre + ST-Gumbel + DynaMaskHead + DynaCrossEntropyLoss, then backward of the summed losses (no '
                   'optimiser step); get_targets: the same positives, four sizes; paste_rle: 100 detections, 112x112 logits -> sigmoid '
                   '-> paste into 800x1333 -> >= 0.5 -> COCO RLE dicts on the host (paste_bitmaps: bool arrays on the host instead); '
                   'median of individually event-timed calls incl. their host work; *_host_syncs = blocking host<-device waits per call')
    out['losses'] = {k: float(v) for k, v in kept['losses'].items()}
    del rh, feats
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this same
    script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, as
    torch.distributed.run would), wait for all of them, return the worst exit code.  The parent
    never touches the GPU and execs nothing: rank 0's JSON line goes straight to our stdout."""
    import subprocess
    port = os.environ.get('MASTER_PORT') or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        live = list(procs)
        while live:
            time.sleep(0.2)
            for p in list(live):
                c = p.poll()
                if c is None:
                    continue
                live.remove(p)
                rc = max(rc, abs(c))
            if rc:                 # a rank that died must not leave the others waiting in a collective
                break
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--leg', choices=('infer', 'train'), default='infer',
                    help="which leg is the headline `value`: 'infer' = BASELINE configs[1] (default); 'train' = the "
                         "configs[2]/[3] training step, timed over exactly --steps steps (the other leg's figures stay "
                         "top-level keys of the same line)")
    ap.add_argument('--no-graph', action='store_true', help='time eager launches instead of a HIP graph replay')
    ap.add_argument('--end-to-end', action='store_true', help='(default since round 4; kept for old command lines)')
    ap.add_argument('--no-end-to-end', action='store_true',
                    help='skip the whole-detector context leg (stock MIOpen ResNet-50-FPN + the mask path, rank 0 only)')
    ap.add_argument('--cpu-sample', type=int, default=512, help='RoIs of the batch timed on the host cores (0 = skip)')
    args = ap.parse_args()

    # ---- ranks: one process per GPU.  The driver starts N>1 ranks itself (torch.distributed.run
    # sets WORLD_SIZE); a bare `python bench.py --gpus N` starts them here, as fresh children,
    # BEFORE this process has touched the GPU (nothing above initialises HIP). ----
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    # stdout carries exactly ONE line, the JSON: libraries that print banners from C (RCCL's
    # version block at communicator creation) are sent to stderr by pointing fd 1 there; the
    # result line is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU '
                 f'(python bench.py --gpus N, or torch.distributed.run --nproc-per-node N bench.py --gpus N)')
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('DM_BENCH_LAUNCH_ONLY', '0') == '1':
        # launcher check without a GPU (tests/test_dist_cpu.py): the ranks meet over gloo, sum
        # their rank numbers and rank 0 prints what the launcher gave them
        import torch.distributed as dist
        dist.init_process_group('gloo')
        t = torch.tensor([float(rank)])
        dist.all_reduce(t)
        if rank == 0:
            os.write(json_fd, (json.dumps({'n_gpus': world, 'rank_sum': float(t.item()), 'launcher': 'ok'}) + '\n').encode())
        dist.barrier()
        dist.destroy_process_group()
        return
    # rehearsal on a box with fewer GPUs than ranks: DM_BENCH_REHEARSAL=1 puts every rank on
    # cuda:0 and uses gloo (RCCL refuses two ranks on one device)
    rehearsal = os.environ.get('DM_BENCH_REHEARSAL', '0') == '1'
    if rehearsal:
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() < world:
        sys.exit(f'bench.py: {world} ranks need {world} GPUs, this node shows {torch.cuda.device_count()} '
                 f'(DM_BENCH_REHEARSAL=1 rehearses the launcher on one device over gloo)')
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if world > 1:
        # default group = gloo: host-side barriers and the max over ranks.  The RCCL communicator of the data-path
        # collective lives only inside the training leg (train_step_bench) at every N, so that each rank's headline
        # leg runs in the same process state as the N = 1 bench (an idle communicator's watchdog thread slows eager
        # multi-stream launch sequences by ~10 %, measured on full_head_112_ms: 10.0 -> 11.45 ms).
        dist.init_process_group('gloo')

    head, sd = build_head(dev)
    feats_c, rois_c, labels_c = make_inputs(rank, dev)
    feats = [f.to(dev) for f in feats_c]
    rois, labels = rois_c.to(dev), labels_c.to(dev)

    # training step (configs[2]/[3]) on every rank (parameters are restored afterwards); reported in `extra`, not
    # the headline.  It runs FIRST: its four streams must each sit on a hardware queue of their own, and the
    # HIP-graph capture of the headline leg below brings streams of its own that end up sharing queues with them
    # (measured: 23.6 ms per step before the capture, 24.3 ms after it).  The headline leg is not affected by the
    # order (the training leg leaves nothing behind but cached allocator memory).
    t_steps, t_warm = (args.steps, args.warmup) if args.leg == 'train' else (4, 4)
    train_ms, train_loss, n_flat, train_b, comm_ms, collective, forced_ms, exposed_ms = train_step_bench(
        head, dev, rank, world, steps=t_steps, warmup=t_warm, rehearsal=rehearsal)

    entry = None
    if rank == 0 and os.environ.get('DM_BENCH_NO_ENTRY', '0') != '1':
        # the reference's real entry points as a whole (rank 0): before the headline's graph capture, like the training leg
        entry = entry_points_bench(sd, dev)
        torch.cuda.empty_cache()

    def step():
        with torch.no_grad():
            return head._mask_forward(feats, rois, labels, last_stage=1)
    eager_step = step

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # The step is ~25 dependent launches; replay it as one HIP graph (no tracing
    # compiler involved: the graph holds exactly the C-ABI launches of step()).
    graph = None
    if not args.no_graph:
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                graph_out = step()
            g.replay()
            torch.cuda.synchronize()
            ref_out = step()
            if not torch.equal(graph_out['stage_instance_preds'][1], ref_out['stage_instance_preds'][1]):
                raise RuntimeError('graph replay differs from eager')
            graph = g

            def step():          # noqa: F811
                graph.replay()
                return graph_out
        except Exception as e:      # capture not available: stay eager (recorded in the JSON)
            print(f'[bench] HIP graph capture failed, timing eager launches: {e}', file=sys.stderr)
            graph = None
    def timed_window():
        """EXACTLY args.steps steps between barrier + synchronize on both sides; max over ranks."""
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        return _max_over_ranks(time.perf_counter() - t0, world)
    # three such windows, the median is reported (one window of 20 graph replays is 80 ms: a single
    # sample of that length is at the mercy of the clock ramp)
    windows = sorted(timed_window() for _ in range(3))
    dt = windows[1]
    ms_per_step = dt / args.steps * 1e3
    value = world * 1.0 / (dt / args.steps)          # images (RoI batches of 512) per second, all ranks

    result = {
        'metric': 'img/s (DynaMask mask-head path, 512 RoIs/img, fixed 28x28 exit; ms_per_step = ms per RoI batch)',
        'value': value, 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[1]: DynaMask R-50-FPN mask head inference, 1333x800 FPN shapes, '
                               '512 RoIs/img, fixed 28x28 exit (RoIAlign14 + 2 conv3x3 + SFM stage 0 + stage-1 logits)',
                   'rois_per_img': ROIS_PER_IMG, 'imgs_per_gpu': 1, 'parallelism': f'images sharded x{world}, no collective',
                   'cpu_baseline': 'rank 0 at N = 1 only (omitted from the N > 1 lines)',
                   'launch': 'hip graph replay' if graph is not None else 'eager',
                   'timing': f'median of 3 windows of {args.steps} steps; windows ms/step = '
                             + ', '.join(f'{w / args.steps * 1e3:.3f}' for w in windows)},
    }
    # The training step (BASELINE configs[2] at N = 1, configs[3] at N = 8) is the unit of the north star's scaling
    # curve: its figures are TOP-LEVEL keys of the line at every N, whichever leg is the headline.
    result.update({
        'train_ms_per_step': train_ms, 'train_img_per_s': world * train_b / (train_ms * 1e-3),
        'train_imgs_per_gpu': train_b, 'train_steps_timed': t_steps,
        'allreduce_alone_ms': comm_ms, 'allreduce_exposed_ms': exposed_ms, 'collective': collective,
        'infer_ms_per_roi_batch': ms_per_step, 'infer_img_per_s': value,
    })
    if entry is not None:
        for k in ('forward_train_ms', 'get_targets_ms', 'paste_rle_ms'):
            result[k] = entry[k]
        result['entry_points'] = entry
    if args.leg == 'train':
        result.update({
            'metric': 'img/s (DynaMask mask-path training step: 2 img/GPU x 128 positive RoIs, fwd + loss + bwd + '
                      'flat-gradient all-reduce + fused SGD; ms_per_step = ms per training step)',
            'value': result['train_img_per_s'], 'ms_per_step': train_ms,
        })
        result['config'] = {
            'workload': ('BASELINE configs[2]: DynaMask R-50-FPN training step, 1xMI355X' if world == 1 else
                         f'BASELINE configs[3]: DynaMask R-50-FPN training, {world}xMI355X, 2 img/GPU, RCCL grad all-reduce') +
                        ', dynamic 14/28/56/112 resolution selection + BCE backward (mask path: RoIAlign14 + RoIAlign56 + '
                        'MaskPre + ST-Gumbel + DynaMaskHead + loss + backward)',
            'imgs_per_gpu': train_b, 'pos_rois_per_img': 128, 'parallelism': f'images sharded x{world}, {collective}',
            'launch': 'eager (four streams)', 'timing': f'median of 3 windows of {t_steps} steps'}

    if rank == 0:
        from dynamask_amd import ops
        # ---- roofline of the dominant kernel: conv3x3 256->256 on [512,256,14,14] ----
        x = torch.randn(ROIS_PER_IMG, 256, 14, 14, device=dev)
        conv = head.mask_head.instance_convs[0].conv
        wp, b = conv.packed([256]), conv.bias.detach()
        conv_call = lambda: ops.conv2d(x, wp, b, 256, 3, relu=True)      # noqa: E731
        ms = sorted(time_kernel(conv_call, iters=10, warmup=2) for _ in range(5))[2]     # median of 5 x 10 calls
        flops = 2.0 * ROIS_PER_IMG * 196 * 256 * 256 * 9
        ach = flops / (ms * 1e-3) / 1e12
        result['roofline'] = {'kernel': 'conv_igemm_kernel<3,2,2,2,2,8> + its <3,4,1,1,1,8> last-round launch (conv3x3 256->256 '
                                        '+bias+ReLU, 512 RoIs; ms_per_launch = both launches of one dm_conv2d_fwd call)',
                              'bound': 'mfma', 'achieved': ach, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                              'frac': ach / PEAK_FP32_MFMA_TFLOPS, 'traffic': None, 'ms_per_launch': ms,
                              'flops_per_launch': flops}
        # ---- the second-largest kernel of the step: DCN 3x3 256->256 (deform_groups 2) on [512,256,14,14] ----
        dcn = head.mask_head.stages[0].fuse_conv[1]
        off = torch.randn(ROIS_PER_IMG, 36, 14, 14, device=dev) * 0.5
        wd = dcn._pk.get('w', dcn.weight, ops.pack_conv_weight, job=(False, None, None, None))
        dcn_call = lambda: ops.deform_conv(x, off, wd, 256, 2, relu=True)      # noqa: E731
        ms_d = sorted(time_kernel(dcn_call, iters=10, warmup=2) for _ in range(5))[2]
        ach_d = flops / (ms_d * 1e-3) / 1e12
        result['roofline_dcn'] = {'kernel': 'deform_conv_lds_kernel (+ its last-round launch): DCNv1 3x3 256->256, deform_groups 2, +ReLU, 512 RoIs '
                                            '@14x14, offsets N(0, 0.5) px; the bilinear gather is the MFMA B-operand producer (no column matrix); '
                                            'flops = the dense contraction only (2 N 196 256 256 9), the gather arithmetic is not counted',
                                  'bound': 'mfma', 'achieved': ach_d, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                  'frac': ach_d / PEAK_FP32_MFMA_TFLOPS, 'traffic': None, 'ms_per_launch': ms_d, 'flops_per_launch': flops}
        # ---- RoIAlign 14x14 multi-level (the north star's HBM-roofline kernel) ----
        ext = head.mask_roi_extractor
        ms_r = time_kernel_graphed(lambda: ext(feats[:4], rois))
        ms_r1 = time_kernel_single_in_graph(lambda: ext(feats[:4], rois))
        ms_rc = time_kernel_cold(lambda: ext(feats[:4], rois))
        ms_rcc = time_kernel_cold(lambda: ext(feats[:4], rois), dirty=False)
        _, lv = ops.roi_align(feats[:4], rois, 14, [1 / 4, 1 / 8, 1 / 16, 1 / 32], return_levels=True)
        nbytes = roialign_algorithmic_bytes(rois_c, lv.cpu().long(), [tuple(f.shape[2:]) for f in feats_c[:4]])
        # `frac` is the HBM figure: the call from COLD caches (194 MB of maps + output fit the 256 MiB Infinity
        # Cache, so launches replayed back to back measure that cache, not HBM); the cache-warm figures stay beside it.
        ms_hbm = ms_rc if ms_rc else ms_r
        ach_r = nbytes / (ms_hbm * 1e-3) / 1e9
        ach_w = nbytes / (ms_r * 1e-3) / 1e9
        result['roofline_roialign'] = {'kernel': 'roi_order_kernel + roi_align_tile_kernel (one dm_roi_align_fwd_ws call: RoIs ranked by level and position on the device, then LDS-staged channel-quad tiles, merged stencils; P2..P5 -> [512,256,14,14]); '
                                                 'ms_per_launch / achieved / frac = ONE call from cold caches (a 512 MiB read-modify-write sweep of another buffer before every '
                                                 'timed call, events around a one-call HIP graph, median of 9); ms_per_launch_cold_clean / frac_cold_clean = the same with a sweep that only '
                                                 'READS its 512 MiB (the lines the call displaces are clean: no write-back of the evictor\'s data inside the timed region); ms_per_launch_warm / frac_warm = 20 calls replayed back to back '
                                                 'as one HIP graph / 20 (maps + output stay in the Infinity Cache), median of 7 replays; ms_per_launch_single = replay of a '
                                                 'graph of two calls minus a graph of one (medians of 15): one warm call with one predecessor',
                                       'bound': 'hbm', 'achieved': ach_r, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                       'frac': ach_r / PEAK_HBM_GBS, 'traffic': None, 'ms_per_launch': ms_hbm,
                                       'cache_state': 'cold' if ms_rc else 'warm (cold timing unavailable)',
                                       'ms_per_launch_cold_clean': ms_rcc,
                                       'frac_cold_clean': (nbytes / (ms_rcc * 1e-3) / 1e9 / PEAK_HBM_GBS) if ms_rcc else None,
                                       'ms_per_launch_warm': ms_r, 'achieved_warm': ach_w, 'frac_warm': ach_w / PEAK_HBM_GBS,
                                       'ms_per_launch_single': ms_r1,
                                       'frac_single': (nbytes / (ms_r1 * 1e-3) / 1e9 / PEAK_HBM_GBS) if ms_r1 else None,
                                       'bytes_per_launch': nbytes}
        pmc = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(pmc):
            try:
                t = json.load(open(pmc))
                result['roofline']['traffic'] = t.get('conv3x3_bytes_per_launch')
                result['roofline_dcn']['traffic'] = t.get('dcn3x3_bytes_per_launch')
                result['roofline_roialign']['traffic'] = t.get('roialign_bytes_per_launch')
                src = ('profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of '
                       'tools/collect_profiles.sh (' + str(t.get('collected', 'date n/a')) + '), not measured in this run')
                result['roofline']['traffic_source'] = src
                result['roofline_dcn']['traffic_source'] = src
                result['roofline_roialign']['traffic_source'] = src
            except Exception:
                pass
        # the committed rocprofv3 per-kernel average of the same kernel (tools/pmc_probe.py alternates it with the
        # convolution: the maps are not cache-warm as in the graph of 20) -- a number from a file, labelled as such
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_roofline_kernel_stats.csv')))
        stats = cands[-1] if cands else ''          # the latest round's collection
        if stats:
            try:
                import csv
                parts, mm = {}, {'conv_igemm_kernel<3': {}, 'deform_conv': {}}
                for row in csv.DictReader(open(stats)):
                    for kname in ('roi_align_tile_kernel', 'roi_order_kernel'):
                        if kname in row['Name']:
                            parts[kname] = float(row['AverageNs']) / 1e3
                    for kname in mm:
                        if kname in row['Name']:
                            mm[kname][row['Name'].split('(anonymous namespace)::')[1].split('(')[0]] = float(row['AverageNs']) / 1e3
                rsrc = ('profiles/' + os.path.basename(stats) + ' (rocprofv3 --kernel-trace --stats over tools/pmc_probe.py), '
                        'not measured in this run')
                if 'roi_align_tile_kernel' in parts:
                    us = sum(parts.values())               # the call = the ordering kernel (if it ran) + the extraction
                    result['roofline_roialign']['us_per_launch_rocprof_committed'] = us
                    result['roofline_roialign']['us_per_kernel_rocprof_committed'] = parts
                    result['roofline_roialign']['frac_rocprof_committed'] = nbytes / (us * 1e-6) / 1e9 / PEAK_HBM_GBS
                    result['roofline_roialign']['rocprof_source'] = rsrc
                # the same bookkeeping for the two MFMA kernels: a call = its main launch + its last-round launch, summed
                for key, kname in (('roofline', 'conv_igemm_kernel<3'), ('roofline_dcn', 'deform_conv')):
                    if mm[kname]:
                        us = sum(mm[kname].values())
                        result[key]['ms_per_launch_rocprof_committed'] = us / 1e3
                        result[key]['us_per_kernel_rocprof_committed'] = mm[kname]
                        result[key]['frac_rocprof_committed'] = flops / (us * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS
                        result[key]['rocprof_source'] = rsrc
            except Exception as e:      # noqa: BLE001
                print(f'[bench] committed rocprof figures unavailable: {e}', file=sys.stderr)
        extra = {}
        # ---- the reference's real inference shape (dynamask_roi_head.py:117-158, tools/benchmark.py:63-89): <= 100
        # detections per image, every exit to 112x112 + boundary merge, through the product's bucketed HIP-graph
        # replay (graphs.py); the eager figures beside it
        with torch.no_grad():
            inf = {}
            for nd in (100, 16):
                det, dl = rois[:nd, 1:].contiguous(), labels[:nd].contiguous()
                call = lambda: head.simple_test_mask_logits(feats, det, dl)      # noqa: E731
                head.enable_inference_graphs(False)
                inf[f'eager_{nd}dets_ms'] = time_kernel_median(call, iters=9, warmup=2)
                gl = head.enable_inference_graphs(True)
                ref_out = call().clone()
                inf[f'graph_{nd}dets_ms'] = time_kernel_median(call, iters=15, warmup=3)
                head.enable_inference_graphs(False)
                inf[f'graph_{nd}dets_equals_eager'] = bool(torch.equal(ref_out, call()))
                inf[f'graph_{nd}dets_captures'] = gl.captures
            result['inference_100dets_ms'] = inf['graph_100dets_ms']
            result['inference_16dets_ms'] = inf['graph_16dets_ms']
            inf['what'] = ('simple_test_mask_logits: RoIAlign14 + DynaMaskHead to 112x112 + boundary merge for the first N RoIs '
                           'of the image; graph = DynaMaskRoIHead.enable_inference_graphs() (buckets 16/24/32/48/64/80/100)')
            extra['inference'] = inf
        # ---- other exits, for context (not the headline) ----
        with torch.no_grad():
            extra['full_head_112_ms'] = time_kernel(lambda: head._mask_forward(feats, rois, labels), iters=5, warmup=1)
        # ---- per-RoI early exit: same 512 RoIs, exits uniform over 14/28/56/112 ----
        ex = torch.arange(ROIS_PER_IMG, device=dev) % 4
        det_boxes = rois[:, 1:].contiguous()
        extra['dynamic_inference'] = {
            'uniform_exits_ms': time_kernel(lambda: head.dynamic_mask_logits(feats, det_boxes, labels, exits=ex),
                                            iters=5, warmup=2),
            'all_exit_112_ms': time_kernel(lambda: head.dynamic_mask_logits(feats, det_boxes, labels,
                                                                            exits=torch.full_like(ex, 3)), iters=5, warmup=2),
            'with_selector_ms': time_kernel(lambda: head.dynamic_mask_logits(feats, det_boxes, labels), iters=5, warmup=2),
            'selector_exit_histogram': torch.bincount(head.dynamic_mask_logits(feats, det_boxes, labels)['exits'],
                                                      minlength=4).tolist(),
            'what': 'RoIAlign14 + DynaMaskHead with each RoI leaving at its own exit (+ boundary merge up to it); '
                    'with_selector adds RoIAlign56(P2) + MaskPre + argmax (random-init selector: histogram is arbitrary)'}
        # ---- SURVEY 8d reading (i) of cfg-2 and cfg-5: the fixed-28x28 FCN producers ----
        from dynamask_amd import registry, synth
        gi = synth          # the reference's config values

        def fcn_ms(up, fpn_feats, fpn_rois):
            cfg = dict(type='FCNMaskHead', **gi.FCN_HEAD_CFG)
            cfg.pop('loss_mask')
            if up == 'carafe':
                cfg['upsample_cfg'] = dict(type='carafe', scale_factor=2, up_kernel=5, up_group=1, encoder_kernel=3,
                                           encoder_dilation=1, compressed_channels=64)
            fcn = registry.build_head(cfg)
            fsd = synth.init_fcn_head_state(seed=7, upsample=up, test_mode=True)
            fcn.load_state_dict({k[len('mask_head.'):]: v for k, v in fsd.items()}, strict=True)
            fcn = fcn.to(dev).eval()
            ext = head.mask_roi_extractor
            with torch.no_grad():
                return time_kernel_median(lambda: fcn(ext(fpn_feats[:4], fpn_rois)))
        extra['fcn_deconv_28_ms'] = fcn_ms('deconv', feats, rois)
        f5 = [f.to(dev) for f in synth.make_fpn(1, 1024, 2048, 256, seed=20)]
        r5 = synth.make_rois(1, ROIS_PER_IMG, 1024, 2048, seed=21).to(dev)
        extra['fcn_carafe_cfg5_2048x1024_ms'] = fcn_ms('carafe', f5, r5)
        # configs[4] as an inference call of the reference (standard_roi_head.simple_test_mask -> FCNMaskHead.forward ->
        # get_seg_masks, fcn_mask_head.py:151-237, -> encode_mask_results): 100 detections on the 2048x1024 image,
        # RoIAlign14 + 4 conv3x3 + CARAFE x2 + 80-class logits + class select + sigmoid + paste + threshold + RLE on the host
        from dynamask_amd.registry import ConfigDict as _CD
        cfg5 = dict(type='FCNMaskHead', **gi.FCN_HEAD_CFG)
        cfg5.pop('loss_mask')
        cfg5['upsample_cfg'] = dict(type='carafe', scale_factor=2, up_kernel=5, up_group=1, encoder_kernel=3, encoder_dilation=1,
                                    compressed_channels=64)
        fcn5 = registry.build_head(cfg5)
        fcn5.load_state_dict({k[len('mask_head.'):]: v for k, v in synth.init_fcn_head_state(seed=7, upsample='carafe', test_mode=True).items()},
                             strict=True)
        fcn5 = fcn5.to(dev).eval()
        det5 = torch.cat([r5[:100, 1:], torch.ones(100, 1, device=dev)], 1)
        lab5 = labels[:100].contiguous()

        def fcn5_infer(rle=True):
            with torch.no_grad():
                mp = fcn5(head.mask_roi_extractor(f5[:4], r5[:100].contiguous()))
                fn_ = fcn5.get_seg_rles if rle else fcn5.get_seg_masks
                return fn_(mp, det5, lab5, _CD(mask_thr_binary=0.5), (1024, 2048, 3), 1.0, True)
        result['fcn_carafe_cfg5_infer100_rle_ms'] = time_kernel_median(fcn5_infer, iters=7, warmup=2)
        extra['fcn_carafe_cfg5_infer100'] = {
            'rle_ms': result['fcn_carafe_cfg5_infer100_rle_ms'],
            'bitmaps_ms': time_kernel_median(lambda: fcn5_infer(False), iters=7, warmup=2),
            'what': 'BASELINE configs[4] workload as ONE inference call on one GPU: 2048x1024 FPN maps, 100 detections, RoIAlign14 -> '
                    'FCNMaskHead (4 conv3x3, CARAFE x2, 80-class 1x1) -> FCNMaskHead.get_seg_rles (class select, sigmoid, paste into '
                    '1024x2048, >= 0.5, COCO RLE dicts on the host) / get_seg_masks (bool arrays on the host); event-timed incl. host work'}
        del f5, r5, fcn5
        # ---- measured device-to-device copy rate (second denominator for the HBM-bound kernels) ----
        a_ = torch.empty(1 << 28, device=dev, dtype=torch.float32)
        b_ = torch.empty_like(a_)
        ms_c = time_kernel(lambda: b_.copy_(a_), iters=10, warmup=2)
        copy_gbs = 2 * a_.numel() * 4 / (ms_c * 1e-3) / 1e9
        del a_, b_
        extra['hbm_copy_GBps_read_plus_write'] = copy_gbs
        result['roofline_roialign']['frac_of_measured_copy'] = result['roofline_roialign']['achieved'] / copy_gbs
        result['extra'] = extra
        # ---- CPU baseline: the oracle on this box's host cores ----
        if args.cpu_sample > 0 and world == 1:
            default_threads = torch.get_num_threads()
            torch.set_num_threads(min(default_threads, host_core_share()))
            dt_cpu, cores, out_cpu = cpu_baseline(sd, feats_c, rois_c, labels_c, args.cpu_sample, reps=6)
            gpu = eager_step()      # not the graph: its packed-weight buffers predate the training leg
            err = float((gpu['stage_instance_preds'][1][:args.cpu_sample, 0].cpu() - out_cpu[0]).abs().max())
            torch.set_num_threads(1)
            n1 = min(8, args.cpu_sample)
            dt1, _, _ = cpu_baseline(sd, feats_c, rois_c, labels_c, n1)
            torch.set_num_threads(default_threads)
            extra['cpu_baseline_1thread'] = {'value': (n1 / ROIS_PER_IMG) / dt1, 'unit': 'img/s', 'cores': 1,
                                             'sample': f'first {n1} RoIs, {dt1:.2f} s per pass'}
            result['cpu_baseline'] = {'value': (args.cpu_sample / ROIS_PER_IMG) / dt_cpu, 'unit': 'img/s', 'cores': cores,
                                      'kind': 'port',
                                      'sample': f'first {args.cpu_sample} of the 512 RoIs of the same image through the same '
                                                f'28x28 exit (PyTorch-CPU oracle, {cores} threads = this process\'s CPU quota), '
                                                f'{dt_cpu:.2f} s per pass, 1 warm-up + 6 timed passes; '
                                                f'scaled to 512 RoIs/img', 'max_abs_err_vs_gpu': err}

        if not args.no_end_to_end:
            # Context only: the (out-of-scope) stock MIOpen backbone next to the mask path at
            # the reference's inference shape: <=100 detections, all exits to 112x112, boundary
            # merge, paste.  RPN / bbox branch / NMS are not part of this repo and not timed.
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            from stock_backbone import ResNet50FPN
            from dynamask_amd.registry import ConfigDict
            bb = ResNet50FPN().to(dev).eval()
            img = torch.randn(1, 3, 800, 1344, device=dev)
            head.test_cfg = ConfigDict(mask_thr_binary=0.5)
            det = torch.cat([rois[:100, 1:], torch.ones(100, 1, device=dev)], 1)
            dl = labels[:100]
            meta = [dict(ori_shape=(800, 1333, 3), scale_factor=1.0)]
            with torch.no_grad():
                t_bb = time_kernel(lambda: bb(img), iters=10, warmup=3)

                def e2e():
                    f = bb(img)
                    return head.simple_test_mask([t.contiguous() for t in f], meta, det, dl)
                for _ in range(2):
                    e2e()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    e2e()            # includes the device->host copy of the pasted masks, as the reference
                torch.cuda.synchronize()
                t_e2e = (time.perf_counter() - t0) / 5 * 1e3
                t_mask = time_kernel(lambda: head.simple_test_mask_logits(feats, det, dl), iters=10, warmup=2)

                def e2e_rle():
                    f = bb(img)
                    return head.simple_test_mask([t.contiguous() for t in f], meta, det, dl, encode=True)
                for _ in range(2):
                    e2e_rle()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    e2e_rle()        # masks leave the device as COCO RLE (device encoder), not as bitmaps
                torch.cuda.synchronize()
                t_e2e_rle = (time.perf_counter() - t0) / 5 * 1e3
            # whole RoI head at the reference's test shape: 1000 proposals -> bbox branch
            # (RoIAlign 7x7, 2 FC + predictors, decode, NMS) -> <= 100 detections -> mask branch -> RLE
            from dynamask_amd import bbox_heads  # noqa: F401
            gi = synth
            rh = registry.build_head(dict(
                type='DynaMaskRoIHead',
                bbox_roi_extractor=dict(type='SingleRoIExtractor', **gi.BBOX_ROI_EXTRACTOR_CFG),
                bbox_head=dict(type='Shared2FCBBoxHead', **gi.BBOX_HEAD_CFG),
                mask_roi_extractor=dict(type='SingleRoIExtractor', **gi.MASK_ROI_EXTRACTOR_CFG),
                mask_head=dict(type='DynaMaskHead', **gi.MASK_HEAD_CFG), test_cfg=ConfigDict(**gi.RCNN_TEST_CFG)))
            rh.load_state_dict({**sd, **synth.init_bbox_head_state(seed=8)}, strict=True)
            rh = rh.to(dev).eval()
            props = synth.make_rois(1, 1000, IMG_H, IMG_W, seed=31)[:, 1:].contiguous().to(dev)
            meta2 = [dict(img_shape=(IMG_H, IMG_W, 3), ori_shape=(IMG_H, IMG_W, 3), scale_factor=1.0)]

            def full_head():
                return rh.simple_test(feats, [props], meta2, rescale=False, encode=True)
            for _ in range(2):
                full_head()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                out_fh = full_head()
            torch.cuda.synchronize()
            t_fh = (time.perf_counter() - t0) / 5 * 1e3
            n_det = sum(len(b) for b in out_fh[0])
            with torch.no_grad():
                t_bbox = time_kernel(lambda: rh._bbox_forward(feats, bbox2roi_([props])), iters=10, warmup=2)
            result['extra']['roi_head_simple_test'] = {
                'ms': t_fh, 'detections': n_det, 'bbox_forward_1000_props_ms': t_bbox,
                'what': 'DynaMaskRoIHead.simple_test on resident FPN maps: 1000 proposals -> RoIAlign7 + Shared2FC (dm_fc_fwd) '
                        '+ softmax/decode + NMS -> masks of the kept detections -> RLE (random-init heads)'}
            # the north star's whole-detector figure (tools/benchmark.py protocol: synchronise around each forward, data
            # loading excluded) as top-level keys: context -- the backbone is stock PyTorch-ROCm / MIOpen, out of scope
            result['end_to_end_img_per_s'] = 1e3 / t_e2e
            result['end_to_end_img_per_s_rle'] = 1e3 / t_e2e_rle
            result['backbone_fpn_ms'] = t_bb
            result['end_to_end_what'] = ('context, not the headline: stock MIOpen ResNet-50+FPN fp32 (out of scope, random weights) + this '
                                         'repo\'s mask path at 100 detections incl. merge, paste and D2H of the masks (bitmaps / device RLE); '
                                         'RPN, bbox head and NMS not included')
            result['extra']['end_to_end'] = {
                'backbone_fpn_ms': t_bb, 'mask_path_100dets_ms': t_mask, 'backbone_plus_mask_path_ms': t_e2e,
                'backbone_plus_mask_path_rle_ms': t_e2e_rle, 'img_per_s_rle': 1e3 / t_e2e_rle,
                'img_per_s': 1e3 / t_e2e,
                'what': 'stock PyTorch-ROCm/MIOpen ResNet-50+FPN fp32 (random weights, out of scope) + this repo\'s mask path '
                        'for 100 detections incl. merge, paste and D2H of the bool masks; RPN / bbox head / NMS not included'}

    if rank == 0:
        extra = result['extra']
        extra['train_step'] = {'ms_per_step': train_ms, 'img_per_s': world * train_b / (train_ms * 1e-3),
                               'imgs_per_gpu': train_b, 'pos_rois_per_img': 128, 'loss': train_loss,
                               'allreduce_floats': n_flat, 'allreduce_alone_ms': comm_ms,
                               'collective': collective, 'ms_per_step_with_forced_one_rank_allreduce': forced_ms,
                               'allreduce_exposed_ms': exposed_ms,
                               'allreduce_exposed_how': 'N > 1: step with the all-reduce - the same window without it; N = 1: step with the '
                                                        'forced one-rank all-reduce - plain step',
                               'what': 'fwd + loss + bwd (head, MaskPre, RoIAlign) + RCCL all-reduce of the flat '
                                       'mask-path gradient + fused SGD; BASELINE configs[2] (N=1) / configs[3] (N=8)'}
        os.write(json_fd, (json.dumps(result) + '\n').encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
