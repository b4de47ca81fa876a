"""Multi-GPU coupling of the mask-head path: ONE flat fp32 gradient buffer for
the mask-head + MaskPre parameters (4 162 462 floats = 16.65 MB at the DynaMask
config), all-reduced with RCCL over xGMI (``torch.distributed`` backend "nccl")
on a side stream, then a fused SGD step on the flat parameter buffer.

Replaces, for this path, the reference's ``MMDistributedDataParallel`` bucketed
all-reduce (mmdet/apis/train.py:75-79) + mmcv OptimizerHook: same arithmetic
(sum over ranks, divide by world size), but one collective per step instead of
25 MB buckets over the whole detector.  Images shard across ranks; BatchNorm
statistics, the class-balance term and the per-stage loss normalisers stay
per-rank exactly as in the reference (``broadcast_buffers=False``, no SyncBN).

Device-agnostic plumbing (tested on CPU with the gloo backend); the optimiser
step itself is a HIP kernel and runs only on the GPU.
"""
import os

import torch
import torch.distributed as dist

from . import hazard


class FlatParamGroup:
    """Re-homes ``params`` into one flat buffer (``p.data`` and ``p.grad`` become
    views), so that the all-reduce and the optimiser touch a single tensor.

    The parameters are marked ``_dm_direct_grad``: the backward kernels of this package then ADD their gradients
    straight into ``p.grad`` (the flat view) and hand autograd ``None``, so no AccumulateGrad node runs for them.
    Consequence: tensor hooks, post-accumulate-grad hooks and DistributedDataParallel's reducer hooks never fire on
    these parameters -- this class IS the gradient reduction for them (``all_reduce_async``); do not wrap the same
    parameters in DDP.  ``release()`` (also run when the group is garbage-collected) clears the marks, after which
    the parameters behave like any others again (their storage stays in the flat buffer)."""

    def __init__(self, params, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.numel = n
        self.flat_param = torch.empty(n, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_momentum = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat_param[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[off:off + k].view_as(p)
                p.grad = self.flat_grad[off:off + k].view_as(p)
                p._dm_direct_grad = id(self)   # train_path._direct: backward kernels may accumulate into the view (owner: this group)
                off += k
        self.group = process_group
        self.steps = 0
        self._work = None
        self._synced = False
        from . import streams
        # shared pool (hardware queues are few): slot 2 is also the training step's coordinate-gradient stream, which is
        # idle by the time the collective is issued (after the whole backward)
        self._stream = streams.side(dev, 2) if dev.type == 'cuda' else None

    def release(self):
        """Clear the direct-gradient marks of this group's parameters (the mark used to outlive the group)."""
        for p in getattr(self, 'params', ()):
            if getattr(p, '_dm_direct_grad', False) == id(self):      # (a later group may own the parameter by now)
                p._dm_direct_grad = False

    def __del__(self):
        try:
            self.release()
        except Exception:      # noqa: BLE001  (interpreter shutdown)
            pass

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def _views(self):
        off = 0
        for p in self.params:
            k = p.numel()
            yield p, self.flat_grad[off:off + k].view_as(p)
            off += k

    def zero_grad(self):
        """Zero the flat gradient and point every ``p.grad`` at its (zeroed) view.  Nothing is
        copied: whatever tensor autograd or ``module.zero_grad(set_to_none=True)`` left in
        ``p.grad`` belongs to the step that has just ended."""
        self.flat_grad.zero_()
        hazard.touch('FlatParamGroup.zero_grad', writes=[self.flat_grad])
        for p, view in self._views():
            if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                p.grad = view
        self._synced = False

    def sync_grads(self):
        """Make ``flat_grad`` hold this step's gradients.  Normally ``p.grad`` IS the view and
        there is nothing to do.  If ``p.grad`` was set to None before the backward
        (``module.zero_grad()`` with torch's default ``set_to_none=True``), autograd has created a
        fresh tensor: it is the truth for that parameter and overwrites the view.  A parameter
        that received no gradient at all (``p.grad is None``) contributes zeros (unlike
        ``torch.optim.SGD`` it still gets weight decay and momentum: the step is one fused
        kernel over the flat buffer)."""
        with torch.no_grad():
            for p, view in self._views():
                if p.grad is None:
                    view.zero_()
                elif p.grad.data_ptr() != view.data_ptr():
                    view.copy_(p.grad)
                p.grad = view
        self._synced = True

    def all_reduce_async(self, force=None):
        """Sum the flat gradient over ranks; on GPUs the collective runs on a side
        stream so that it overlaps whatever the caller enqueues next (the backbone
        backward in a full detector).  At world size 1 there is nothing to sum and the
        collective is skipped unless ``force`` (or ``DM_FORCE_COLLECTIVE=1``) asks for it --
        used by the single-GPU tests and bench so that the RCCL call, the side stream and the
        1/world scaling of the fused step run on hardware."""
        self.sync_grads()
        if force is None:
            force = os.environ.get('DM_FORCE_COLLECTIVE', '0') == '1'
        initialised = dist.is_available() and dist.is_initialized()
        if self.world_size == 1 and not (force and initialised):
            return
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                hazard.touch('all_reduce', writes=[self.flat_grad])
                self._work = dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self._work = dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)

    def grad_sumsq(self):
        """Sum of squares of the (reduced) flat gradient as a device scalar: this group's share of
        the model-wide norm of ``clip_grad_norm_``."""
        from . import ops
        if not self._synced:
            self.sync_grads()
        self.wait()
        return ops.sumsq(self.flat_grad)

    def clip_grad_norm_(self, max_norm, other_sumsq=None):
        """``optimizer_config = dict(grad_clip=dict(max_norm=35, norm_type=2))``
        (configs/dynamask/coco/r50-dynamask-1x.py:274 -> OptimizerHook.clip_grads ->
        ``clip_grad_norm_``): scale the flat gradient by min(1, max_norm / (||g|| + 1e-6)).  The
        norm is taken over this group plus ``other_sumsq`` (device scalar: the squared norm of the
        detector's remaining parameters, which the caller owns).  After the all-reduce the buffer
        holds the SUM over ranks, so the mean's norm is ||sum|| / world.  No host sync.
        Returns the total squared norm (of the mean gradient), device scalar."""
        from . import ops
        ss = self.grad_sumsq()
        w = float(self.world_size)
        total = ss / (w * w) if w != 1.0 else ss
        if other_sumsq is not None:
            total = total + other_sumsq
        ops.clip_scale_(self.flat_grad, total.reshape(1).contiguous(), max_norm)
        return total

    def scale_grads_(self, params, factor):
        """Multiply the gradients of ``params`` (a subset of this group) by ``factor``, e.g. the reference's optional
        ``OptimizerHook_`` (OptimizerHook.py:27-29): ``roi_head.mask_predictor`` gradients x 0.05 between the
        clipping and the optimizer step.  Adjacent parameters are scaled as one run of the flat buffer."""
        from . import ops
        if not self._synced:
            self.sync_grads()
        self.wait()
        want = {id(p) for p in params}
        off, runs = 0, []
        for p in self.params:
            k = p.numel()
            if id(p) in want:
                if runs and runs[-1][1] == off:
                    runs[-1][1] = off + k
                else:
                    runs.append([off, off + k])
            off += k
        for lo, hi in runs:
            ops.scale_(self.flat_grad[lo:hi], factor)

    def sgd_step(self, lr=0.02, momentum=0.9, weight_decay=1e-4, grad_scale=1.0):
        """Fused SGD on the flat buffer; the 1/world averaging (times ``grad_scale``, e.g. the
        clip coefficient of ``clip_grad_norm``) rides in the kernel."""
        from . import ops
        if not self._synced:                 # no all_reduce_async() this step (single process)
            self.sync_grads()
        self.wait()
        ops.sgd_momentum_step_(self.flat_param, self.flat_grad, self.flat_momentum, lr, momentum, weight_decay,
                               grad_scale / self.world_size, first_step=(self.steps == 0))
        self.steps += 1
        self._synced = False


def mask_path_parameters(roi_head):
    """The parameters whose gradients this path owns."""
    return list(roi_head.mask_head.parameters()) + list(roi_head.mask_predictor.parameters())


def shard_images(num_images, rank, world_size):
    """Images shard across ranks (the reference's DistributedSampler role for this
    path): contiguous, balanced slices."""
    per = num_images // world_size
    rem = num_images % world_size
    start = rank * per + min(rank, rem)
    return list(range(start, start + per + (1 if rank < rem else 0)))
