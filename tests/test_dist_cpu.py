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
