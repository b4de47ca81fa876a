"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: flat gradient buffer
all-reduce and image sharding."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from dynamask_amd.dist import FlatParamGroup, shard_images
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(5, 7), nn.Linear(7, 3))
    ref = [p.detach().clone() for p in net.parameters()]
    grp = FlatParamGroup(net.parameters())
    assert grp.numel == sum(p.numel() for p in net.parameters())
    for p, r in zip(net.parameters(), ref):          # re-homing keeps values
        assert torch.equal(p.detach(), r)
    grp.zero_grad()
    x = torch.full((4, 5), float(rank + 1))
    net(x).sum().backward()                            # autograd accumulates into the flat views
    local = grp.flat_grad.clone()
    grp.all_reduce_async()
    grp.wait()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    assert torch.allclose(grp.flat_grad, sum(gathered))
    # every parameter's .grad is a view of the reduced flat buffer
    off = 0
    for p in net.parameters():
        assert torch.equal(p.grad.reshape(-1), grp.flat_grad[off:off + p.numel()])
        off += p.numel()
    imgs = shard_images(5, rank, world)
    q.put((rank, imgs, float(grp.flat_grad.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_allreduce_world2_gloo():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    outs.sort()
    assert outs[0][1] + outs[1][1] == [0, 1, 2, 3, 4]           # images partition, no overlap
    assert abs(outs[0][2] - outs[1][2]) < 1e-5                 # identical reduced gradients on both ranks


def test_shard_images_balanced():
    from dynamask_amd.dist import shard_images
    for n, w in ((16, 8), (5, 2), (3, 4)):
        parts = [shard_images(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


# ---------------------------------------------------------------- bench.py rank launcher
def _run_bench(args, env_extra, timeout=180):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        if k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` (no launcher) must start 2 ranks that form one process
    group (--gpus was parsed and ignored)."""
    import json
    r = _run_bench(['--gpus', '2'], {'DM_BENCH_LAUNCH_ONLY': '1'})
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(line) == 1                       # rank 0 alone prints
    out = json.loads(line[0])
    assert out['n_gpus'] == 2 and out['rank_sum'] == 1.0


def test_bench_refuses_gpus_world_size_mismatch():
    r = _run_bench(['--gpus', '4'], {'DM_BENCH_LAUNCH_ONLY': '1', 'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0',
                                     'MASTER_PORT': str(_free_port())})
    assert r.returncode != 0
    assert 'WORLD_SIZE=2' in r.stderr


# ---------------------------------------------------------------- .grad replaced by autograd
def _torch_sgd_(params, grads, mom, lr, momentum, wd, grad_scale, first_step=False):
    """CPU stand-in of dm_sgd_momentum_step for these tests (torch.optim.SGD arithmetic)."""
    g = grads * grad_scale + wd * params
    if first_step:
        mom.copy_(g)
    else:
        mom.mul_(momentum).add_(g)
    params.sub_(lr * mom)


def _grad_none_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from dynamask_amd import ops
    from dynamask_amd.dist import FlatParamGroup
    ops.sgd_momentum_step_ = _torch_sgd_
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(3, 4), nn.Linear(4, 2))
    ref = nn.Sequential(nn.Linear(3, 4), nn.Linear(4, 2))
    ref.load_state_dict(net.state_dict())
    opt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    grp = FlatParamGroup(net.parameters())
    for it in range(3):
        # reference: mean over ranks of the per-rank gradients
        opt.zero_grad()
        for r in range(world):
            (ref(torch.full((2, 3), float(r + 1 + it))).sum() / world).backward()
        opt.step()
        # product: the caller uses torch's default zero_grad (set_to_none=True) in odd steps and
        # the group's own in even ones
        if it % 2:
            net.zero_grad()                      # p.grad = None: autograd will create fresh tensors
        else:
            grp.zero_grad()
        net(torch.full((2, 3), float(rank + 1 + it))).sum().backward()
        grp.all_reduce_async()
        grp.sgd_step(lr=0.1, momentum=0.9, weight_decay=1e-4)
        # after the step every .grad is a view of the flat buffer again and holds THIS step's sum
        exp = torch.cat([p.grad.reshape(-1) for p in ref.parameters()]) * world
        assert torch.allclose(grp.flat_grad, exp, atol=1e-5), (it, grp.flat_grad, exp)
        for p, pr in zip(net.parameters(), ref.parameters()):
            assert torch.allclose(p.detach(), pr.detach(), atol=1e-6), it
    q.put(rank)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_flat_group_survives_grad_set_to_none_world1_and_world2():
    """module.zero_grad(set_to_none=True) between steps must neither drop the
    gradient (world 1) nor accumulate the previous step's (zero_grad + rebind copy)."""
    ctx = mp.get_context('spawn')
    for world in (1, 2):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_grad_none_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0


# ---------------------------------------------------------------- what the N > 1 bench line carries
TOP_LEVEL_TRAIN_KEYS = ('train_ms_per_step', 'train_img_per_s', 'train_imgs_per_gpu', 'allreduce_alone_ms', 'allreduce_exposed_ms', 'collective',
                        'infer_ms_per_roi_batch', 'infer_img_per_s', 'inference_100dets_ms')


def test_two_rank_rehearsal_line_carries_the_training_step_at_top_level():
    """the driver's SCALE record keeps only the top-level keys of the line, so the training-step
    figures (the unit of the north star's scaling curve) must be there at every N.  profiles/rNN_bench_rehearsal_gpus2.json
    (the latest round's) is the line `DM_BENCH_REHEARSAL=1 python bench.py --gpus 2` printed on the one-GPU box
    (tools/collect_profiles.sh); bench.py's source must name the same keys."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import glob
    latest = sorted(glob.glob(os.path.join(root, 'profiles', 'r[0-9][0-9]_bench_rehearsal_gpus2.json')))[-1]
    with open(latest) as f:
        line = [l for l in f.read().splitlines() if l.startswith('{')]
    assert len(line) == 1
    out = json.loads(line[0])
    assert out['n_gpus'] == 2 and out['scaling'] == 'weak'
    for k in TOP_LEVEL_TRAIN_KEYS:
        assert k in out, k
    assert out['train_ms_per_step'] > 0 and out['train_img_per_s'] > 0
    assert 'all-reduce over 2 rank(s), executed' in out['collective']
    assert abs(out['train_img_per_s'] - 2 * out['train_imgs_per_gpu'] / (out['train_ms_per_step'] * 1e-3)) < 1e-6 * out['train_img_per_s']
    src = open(os.path.join(root, 'bench.py')).read()
    for k in TOP_LEVEL_TRAIN_KEYS:
        assert f"'{k}'" in src, k


def test_bench_leg_train_is_an_option_of_the_command_line():
    r = _run_bench(['--help'], {})
    assert r.returncode == 0 and '--leg' in r.stdout and 'train' in r.stdout


# ---------------------------------------------------------------- the whole mask-path step over two ranks
def _whole_step_worker(rank, world, port, q, steps):
    """shard -> forward + loss + backward -> all-reduce of the flat gradient -> SGD, with the product's own plumbing
    (registry-built DynaMaskRoIHead as the parameter holder, dist.mask_path_parameters, FlatParamGroup, shard_images)
    and the ORACLE as the arithmetic: the product's kernels are HIP-only and refuse CPU tensors, the collective and the
    optimiser bookkeeping around them are the same code on either device (the fused SGD kernel is replaced by its torch
    restatement, as in the test above)."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(3)
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tests', 'golden'))
    sys.path.insert(0, root)
    import golden_inputs as gi
    from oracle import ref_model
    from dynamask_amd import ops, registry, roi_head, losses, mask_heads, roi_extractors  # noqa: F401
    from dynamask_amd.dist import FlatParamGroup, mask_path_parameters, shard_images
    ops.sgd_momentum_step_ = _torch_sgd_

    def build():
        cfg = dict(type='DynaMaskRoIHead', mask_roi_extractor=dict(type='SingleRoIExtractor', **gi.MASK_ROI_EXTRACTOR_CFG),
                   mask_head=dict(type='DynaMaskHead', **gi.MASK_HEAD_CFG),
                   train_cfg=registry.ConfigDict(flops=[0.23, 0.62, 1.01, 1.4], Lambda=0.3, mask_size=28),
                   test_cfg=registry.ConfigDict(mask_thr_binary=0.5))
        m = registry.build_head(cfg)
        m.load_state_dict({**gi.head_state(), **gi.mask_pre_state()}, strict=True)
        return m.train()

    hi = gi.head_inputs()
    n = hi['rois'].shape[0]
    U = torch.rand(n, 4, generator=torch.Generator().manual_seed(9))
    tg = gi.head_targets(n)

    def shard_loss(model, images):
        """the mask loss of one rank's images: its own RoIs, its own BatchNorm statistics, its own normaliser
        (mmdet/apis/train.py:75-79: one DDP replica)"""
        sd = {**dict(model.named_buffers()), **dict(model.named_parameters())}
        keep = torch.isin(hi['rois'][:, 0].long(), torch.tensor(images))
        rois = hi['rois'][keep].clone()
        remap = {b: i for i, b in enumerate(images)}
        rois[:, 0] = torch.tensor([remap[int(b)] for b in rois[:, 0]], dtype=torch.float32)
        feats = [f[images] for f in hi['feats']]
        loss, _, ind, _ = ref_model.mask_forward_train(sd, feats, rois, hi['labels'][keep], [t[keep] for t in tg], U[keep])
        return loss, ind

    model = build()
    grp = FlatParamGroup(mask_path_parameters(model))
    mine = shard_images(2, rank, world)
    assert len(mine) == 2 // world
    # the reference run of this test: ONE process, both shards one after the other, gradients averaged, torch.optim.SGD
    ref = build()
    opt = torch.optim.SGD(mask_path_parameters(ref), lr=0.02, momentum=0.9, weight_decay=1e-4)
    for it in range(steps):
        grp.zero_grad()
        loss, ind = shard_loss(model, mine)
        loss.backward()
        grp.all_reduce_async()
        grp.sgd_step(lr=0.02, momentum=0.9, weight_decay=1e-4)
        opt.zero_grad()
        for r in range(2):
            lr_, _ = shard_loss(ref, [r])
            (lr_ / 2).backward()
        opt.step()
        if world == 2:
            worst = max(float((p.detach() - pr.detach()).abs().max()) for p, pr in zip(mask_path_parameters(model), mask_path_parameters(ref)))
            scale = max(float(pr.detach().abs().max()) for pr in mask_path_parameters(ref))
            assert worst <= 1e-6 * max(scale, 1.0), (it, worst)
    flat = grp.flat_param.detach().clone()
    if world > 1:
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1])          # the same bits on both ranks after every update
    q.put((rank, float(flat.double().sum()), [int(i) for i in ind]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_whole_mask_path_step_world2_gloo_equals_one_process_over_both_shards():
    """Two gloo ranks, one image each: forward + DynaCrossEntropyLoss + backward through extractor, MaskPre, selector and
    the four-stage head, all-reduce of the flat 4.16 M-float gradient, momentum SGD -- for two steps.  After every step
    the parameters equal (1e-6) those of one process that ran both shards with their own BatchNorm statistics and loss
    normalisers and averaged the gradients (the DDP semantics of mmdet/apis/train.py:75-79), and both ranks hold the
    same bits.  The arithmetic is the oracle's (tests may use it); every line of dist.py on the path is the product's."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_whole_step_worker, args=(r, 2, port, q, 2)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=900) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert outs[0][1] == outs[1][1]
