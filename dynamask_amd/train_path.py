"""Training step of the mask-head path: forward that keeps what the backward
needs, and a hand-sequenced backward over the C-ABI kernels.

One ``torch.autograd.Function`` per block of the reference graph (RoI extractor,
DynaMaskHead) so that ``loss.backward()`` of an mmdet-style training loop reaches
the parameters, the RoI features and the FPN maps -- autograd only routes
tensors; no arithmetic runs in PyTorch.

Reference graph: ``DynaMaskHead.forward`` / ``SFMStage.forward``
(mask_heads/dynamask_head.py:102-125,220-244), ``SingleRoIExtractor.forward``
(roi_extractors/single_level_roi_extractor.py:53-81); gradients are what
autograd derives for them plus mmcv's DeformConv2d / RoIAlign backward
(mmdet/ops/dcn/src/deform_conv_cuda.cpp:262-486).
"""
import os

import torch

from . import hazard, ops

# maps of at least this many pixels keep the DCN column matrix from the forward for the weight gradient (56 x 56 and, since the
# leaf work was dealt over three side streams, 28 x 28: 20.09 -> 20.00 ms per step over three runs each; all three: 20.2)
_KEEP_COL_MIN_PIXELS = 784
# the training forward splits the RoIs in two halves on two streams from this many RoIs on
_FWD_SPLIT_MIN_ROIS = 128
_SIDE_STREAMS = {'leaf': 0, 'selector': 1, 'bbox': 1, 'coord': 2}      # slots of the shared pool (streams.py)


def side_stream(dev, which='leaf'):
    """A side stream of the training path on ``dev`` (None on the CPU, or with DM_TRAIN_SIDE_STREAM=0):
    'leaf' carries the leaf work of the mask head's passes, 'selector' the resolution-selector branch, 'coord'
    the second of two kernels of the chain that only share their inputs, 'bbox' the bbox branch of ``forward_train``."""
    if dev.type != 'cuda' or os.environ.get('DM_TRAIN_SIDE_STREAM', '1') == '0':
        return None
    from . import streams
    st = streams.side(dev, _SIDE_STREAMS[which])
    if hazard.ENABLED[0]:
        hazard.name_stream(st, f'side{_SIDE_STREAMS[which]}')
    return st


def _join_caller_after_backward(dev):
    """Called from a backward that runs on a side stream (autograd replays a node on the stream of its forward)
    and adds into gradient buffers in place: the stream that called ``backward()`` must not read those buffers
    before this stream is done.  The engine joins the streams of its AccumulateGrad nodes only, so the wait is
    queued as a final callback of the running backward pass (it runs on the caller's stream, before
    ``backward()`` returns)."""
    if dev.type != 'cuda':
        return
    cur = torch.cuda.current_stream(dev)
    if cur == torch.cuda.default_stream(dev):
        return
    torch.autograd.Variable._execution_engine.queue_callback(lambda: torch.cuda.current_stream(dev).wait_stream(cur))


class _SideWork:
    """Leaf work of a backward pass on a second HIP stream.

    Weight / bias gradients and the semantic branch (point-sample adjoint -> 1x1 conv on the FPN map) feed
    nothing further down the chain of data gradients, and the chain's own kernels are partly bound by LDS
    atomics, global atomics or HBM (DCN col2im / coordinate gradient, upsample and ReLU adjoints) while the
    leaves are MFMA GEMMs: issued on a second stream they fill each other's idle units.  ``run`` orders the
    side stream after everything issued so far on the main stream and keeps the tensors it is given alive
    until ``join`` (the caching allocator would otherwise hand a tensor the main stream has dropped to the
    main stream's next allocation while the side stream still reads it).  ``DM_TRAIN_SIDE_STREAM=0`` runs
    everything on the caller's stream."""

    def __init__(self, dev, alt=None):
        """``alt``: name of a second side stream for the calls that ask for it (``run(..., alt=True)``).  The backward
        passes 'coord': the leaf queue is what the step ends on (its backlog of weight gradients kept the optimizer
        waiting ~1 ms after the chain was done) and the coordinate-gradient stream idles between its three kernels.  What
        goes there is what is issued AFTER a stage's coordinate gradient (the offset convolution's weight gradient, which
        reads that kernel's output anyway, and the semantic branch): a long launch queued in front of the coordinate
        gradient would hold up the chain, which waits for it.  ``run(..., alt='selector')`` names another stream of the
        pool: the DCN weight gradient and the 1x1 output convolution's, issued before the coordinate gradient, go to the
        selector / bbox stream, which is idle in the middle of the backward, and so does the offset convolution's
        (20.8 -> 20.45 -> 20.15 -> 20.05 ms per step; the fuse convolution's weight gradients there as well: 20.6, on the
        coordinate-gradient stream: no change; the semantic branch on the selector stream: 20.6; everything on the one leaf
        stream: 20.9; re-measured in round 5 with MaskPre on the map, profiles/r05_train_experiments.txt: the same ranking)."""
        self.side = side_stream(dev)
        self.enabled = self.side is not None
        self.keep = []
        self.alt = None
        self.extra = set()
        if self.enabled:
            self.main = torch.cuda.current_stream(dev)
            self.enabled = self.side != self.main
            if self.enabled and alt is not None:
                self.alt = side_stream(dev, alt)
                if self.alt is None or self.alt == self.main or self.alt == self.side:
                    self.alt = None

    def run(self, fn, *tensors, after=None, alt=False):
        """``after``: an event the side stream waits for INSTEAD of everything issued so far on the main stream
        (work whose inputs were ready long before, issued late so that the host feeds the main stream first)."""
        if not self.enabled:
            return fn()
        self.keep.extend(t for t in tensors if t is not None)
        side = self.side
        if alt and self.alt is not None:                     # (self.alt is None: one leaf stream for everything)
            side = self.alt if alt is True else (side_stream(self.side.device, alt) or self.side)
            if side == self.main:
                side = self.side
            if side is not self.side and side is not self.alt:
                self.extra.add(side)
        if after is not None:
            side.wait_event(after)
        else:
            side.wait_stream(self.main)
        with torch.cuda.stream(side):
            return fn()

    def join(self, *outs):
        """The main stream waits for the side stream; ``outs`` = tensors the side stream produced that live on."""
        if self.enabled:
            self.main.wait_stream(self.side)
            if self.alt is not None:
                self.main.wait_stream(self.alt)
            for e in self.extra:
                self.main.wait_stream(e)
            for t in outs:
                if t is not None:
                    t.record_stream(self.main)
        self.keep.clear()


class RoIExtractFn(torch.autograd.Function):
    """Multi-level RoIAlign with its scatter-add backward (K1/K3)."""

    @staticmethod
    def forward(ctx, rois, output_size, scales, sampling_ratio, finest_scale, *feats):
        feats = [f.contiguous() for f in feats]
        ctx.save_for_backward(rois)
        ctx.cfg = (output_size, list(scales), sampling_ratio, finest_scale, [tuple(f.shape) for f in feats])
        return ops.roi_align(feats, rois, output_size, scales, sampling_ratio, finest_scale)

    @staticmethod
    @hazard.backward_node
    def backward(ctx, g):
        (rois,) = ctx.saved_tensors
        hazard.engine_handoff(g)
        output_size, scales, sr, fs, shapes = ctx.cfg
        grads = ops.roi_align_backward(g.contiguous(), shapes, rois, output_size, scales, sr, fs)
        return (None, None, None, None, None, *grads)


def _pack_dcn_colmajor(w):
    """DCN weight [Cout, C, 3, 3] packed as the 1x1 conv over the tap-major column matrix (9C -> Cout)."""
    cout, c = w.shape[0], w.shape[1]
    return ops.pack_conv_weight(ops.dcn_weight_permute(w.detach().contiguous(), cout, c, True), transpose_flip=True)


def _flip_pack(conv, lo, hi):
    """Packed weights of d/d(input channels lo:hi) of ``conv`` (forward kernel, transposed + rotated)."""
    return conv._pk.get(('flip', lo, hi), conv.weight,
                        lambda t: ops.pack_conv_weight(t[:, lo:hi].contiguous(), transpose_flip=True),
                        job=(True, None, lo, hi))


def _prepack_backward(head, feats):
    for i, cm in enumerate(head.instance_convs):
        _flip_pack(cm.conv, 0, cm.conv.in_channels)
    for idx, stage in enumerate(head.stages):
        c, dcn = stage.instance_in_channel, stage.fuse_conv[1]
        _flip_pack(stage.fuse_transform_out, 0, c)
        _flip_pack(dcn.conv_offset, 0, c)
        for lo, hi in ((0, c), (c, 2 * c), (2 * c, 2 * c + 2)):
            _flip_pack(stage.fuse_conv[0], lo, hi)
        _flip_pack(stage.semantic_transform_in, 0, feats[len(feats) - idx - 3].shape[1])
        dcn._pk.get('colgrad', dcn.weight, ops.pack_dcn_colgrad_weight)


# event after which the step's inputs (FPN maps, RoIs, labels) are ready on the device: recorded by the caller at the
# very start of the step (roi_head._mask_forward_train_tensors) so that branches issued late do not wait for the main
# stream's queue; None = record one where it is needed
_INPUTS_READY = [None]
# launches already issued for the next MaskHeadFn.apply (mask_head_forward_train ``between``)
_PRECOMPUTED = [None]


def _direct(p):
    """The tensor a parameter's gradient may be accumulated into IN PLACE, or None.

    ``FlatParamGroup`` (dist.py) keeps every ``p.grad`` as a zero-filled view of its flat gradient
    buffer and marks the parameter ``_dm_direct_grad``.  The weight-gradient kernels accumulate
    (split-K atomics) anyway, so they add straight into that view and the backward returns None for
    the parameter: no zero-filled temporary and no AccumulateGrad ``add_`` kernel per parameter
    (about 120 small launches per training step).  Without the mark the gradient is returned to
    autograd as usual."""
    g = p.grad
    if g is not None and getattr(p, '_dm_direct_grad', False) and g.is_contiguous() and g.dtype == torch.float32:
        return g
    return None


def params_grad(weight, bias, dy, srcs, ks, pg, wshape=None):
    """Weight and bias gradient of a conv / FC layer from ONE pass over ``dy`` (the bias gradient is the row sums of
    the weight-gradient GEMM's A operand: dm_conv2d_wgrad ``db``; round 2 read dy a second time through
    dm_channel_sum).  Accumulated into the flat-buffer views where the parameters have them (``_direct``), else
    handed back through ``pg``."""
    tw = _direct(weight)
    tb = _direct(bias) if bias is not None else None
    dw = tw.view(wshape) if (tw is not None and wshape is not None) else tw
    if bias is None:
        r = ops.conv2d_wgrad(dy, srcs, ks, dw=dw)
        if tw is None:
            pg[weight] = r.view_as(weight)
        return
    gw, gb = ops.conv2d_wgrad(dy, srcs, ks, dw=dw, db=tb, want_bias=tb is None)
    if tw is None:
        pg[weight] = gw.view_as(weight)
    if tb is None:
        pg[bias] = gb


def _flipped(conv, key, w):
    """Packed weights of the data-gradient convolution, cached on the module."""
    return conv._pk.get(('flip',) + key, w, lambda t: ops.pack_conv_weight(t, transpose_flip=True),
                        job=(True, None, None, None))


class MaskHeadFn(torch.autograd.Function):
    """DynaMaskHead forward/backward.  Inputs: (head, rois, labels, ins_feats,
    n_feats, *fpn_feats, *head.parameters())."""

    @staticmethod
    def forward(ctx, head, rois, labels, ins_feats, n_feats, *tensors):
        pre = _PRECOMPUTED[0]
        _PRECOMPUTED[0] = None
        if pre is None:
            pre = MaskHeadFn.issue(head, rois, labels, ins_feats, tensors[:n_feats])
        ips, dps, saved, feats, rois, labels = pre
        ctx.head, ctx.saved, ctx.feats, ctx.rois, ctx.labels = head, saved, feats, rois, labels
        ctx.n_feats = n_feats
        return (*ips, *dps)

    @staticmethod
    def issue(head, rois, labels, ins_feats, fpn_feats, after_convs=None):
        """The launches of the forward (no autograd state): -> (ips, dps, saved, feats, rois, labels).
        ``after_convs``: called once the instance convolutions of both halves have been issued (the GPU then has
        ~2 ms of work queued): the place for a branch that must finish by the loss but feeds nothing before it."""
        feats = [t.contiguous() for t in fpn_feats]
        labels = labels.long().contiguous()
        rois = rois.contiguous()
        saved = {}
        x = ins_feats.contiguous()
        # semantic branch of every stage (1x1 conv on the FPN map + point sample): independent of the instance
        # chain until the stage's fusion conv, so it runs beside the four instance convs
        sw = _SideWork(rois.device)
        sems, isfs = [], []

        # HOST ISSUE ORDER (round 3).  A step is ~350 launches; the host needs 3.5 ms to issue the forward's.  Round 2
        # issued them in program order -- selector branch, semantic branches, the ~45 weight packs of forward and
        # backward, the second stream's half of the head -- and the main stream's first convolution reached the GPU
        # 3.9 ms into the step (round 3's timeline before this change, docs/HISTORY.md: the chain's queue empty for 3.4 of its first
        # 3.9 ms).  Now what the chain needs first is issued first: the forward's packs, the instance convs of both
        # halves, then the semantic branches (needed at the first fusion conv), the stages, and only then what has
        # slack until the loss or the backward (selector branch in roi_head.py, the backward's packs below).  Work
        # issued late waits for the event of its inputs, not for the main stream's queue.
        if _INPUTS_READY[0] is not None:
            ready = _INPUTS_READY[0]
        else:
            # called without the RoI head's up-front refresh (a direct _mask_forward / mask_head_forward_train under
            # grad, after a weight update): the packs are refreshed here, on the main stream, BEFORE the event the side
            # work waits for (ops.PackPlan.get additionally orders any stream behind the refresh it reads from)
            ops.PACK_PLAN.refresh(rois.device)
            ready = torch.cuda.current_stream(rois.device).record_event() if rois.is_cuda else None

        def semantic_branches():
            for idx, stage in enumerate(head.stages):
                sem = stage.semantic_transform_in.run(feats[len(feats) - idx - 3], relu=True)
                sems.append(sem)
                isfs.append(ops.point_sample(sem, rois, stage.out_size, stage.spatial_scale))
        # Everything below is per RoI: the buffers are allocated for the whole batch (the backward sees whole
        # tensors) and filled by rows -- two halves on two streams when the batch is large enough, so that the
        # memory-bound kernels of one half (logit gathers, deformable im2col, upsampling) run beside the GEMMs of the
        # other.  (The backward is not split: it already shares the GPU between the chain and its leaves, and halving
        # its launches as well was measured slower.)
        dev = x.device
        n = x.shape[0]

        def buf(*shape):
            return torch.empty(shape, device=dev, dtype=torch.float32)
        conv_in = [x]
        for conv in head.instance_convs:
            conv_in.append(buf(n, conv.conv.out_channels, x.shape[2], x.shape[3]))
        n_stages = len(head.stages)
        stbuf = []
        xin = conv_in[-1]
        for idx, stage in enumerate(head.stages):
            up_flag = head.pre_upsample_last_stage or idx < n_stages - 1
            c, s, co = stage.instance_in_channel, stage.out_size, stage.instance_out_channel
            dcn = stage.fuse_conv[1]
            st = dict(xin=xin, tail=buf(n, co, s, s), f1=buf(n, c, s, s), off=buf(n, dcn.conv_offset.out_channels, s, s),
                      f2=buf(n, dcn.out_channels, s, s), ip=buf(n, 1, s, s), dp=buf(n, 1, s, s),
                      col=buf(n, 9 * c, s, s) if s * s >= _KEEP_COL_MIN_PIXELS else None,
                      up=buf(n, co, 2 * s, 2 * s) if up_flag else None, feat_idx=len(feats) - idx - 3)
            stbuf.append(st)
            xin = st['up'] if up_flag else st['tail']
        x_last = xin
        S = x_last.shape[-1]
        fin_ip, fin_dp = buf(n, 1, S, S), buf(n, 1, S, S)
        fin_up = None if head.pre_upsample_last_stage else (buf(n, 1, 2 * S, 2 * S), buf(n, 1, 2 * S, 2 * S))
        lab_last = labels.clamp(max=0) if head.stage_num_classes[-1] == 1 else labels

        def convs(lo, hi):
            for i, conv in enumerate(head.instance_convs):
                conv.conv.run([conv_in[i][lo:hi]], relu=True, out=conv_in[i + 1][lo:hi])

        def stages(lo, hi):
            r, lab = rois[lo:hi], labels[lo:hi]
            for idx, stage in enumerate(head.stages):
                st = stbuf[idx]
                c, s, co, nc = stage.instance_in_channel, stage.out_size, stage.instance_out_channel, stage.num_classes
                x_, tail = st['xin'][lo:hi], st['tail'][lo:hi]
                ops.class_logits(x_, stage.instance_logits.weight.detach().view(nc, c), stage.instance_logits.bias.detach(),
                                 stage.detail_logits.weight.detach().view(nc, c), stage.detail_logits.bias.detach(), lab,
                                 sig_out=tail, sig_ch_offset=co - 2, out=(st['ip'][lo:hi], st['dp'][lo:hi]))
                f1 = stage.fuse_conv[0].run([x_, isfs[idx][lo:hi], tail[:, co - 2:]], relu=True, out=st['f1'][lo:hi])
                dcn = stage.fuse_conv[1]
                off = dcn.conv_offset.run(f1, out=st['off'][lo:hi])
                if st['col'] is not None:
                    # 56 x 56: training keeps the deformable column matrix -- the weight gradient needs it anyway, and
                    # im2col + a 1x1 GEMM over it (0.61 + 0.68 ms, 256 RoIs) cost the chain less than the fused kernel
                    # (0.76 ms) plus an im2col in the backward (0.61 ms, beside the chain on the leaf stream): 23.45 vs
                    # 23.5 ms per step, and 23.75 with the 28 x 28 stage kept too (then; kept now, see _KEEP_COL_MIN_PIXELS).
                    # 1.85 + 0.92 GB live until the backward.
                    col = ops.deform_im2col(f1, off, dcn.deform_groups, out=st['col'][lo:hi])
                    f2 = ops.conv2d([col], dcn._pk.get('w_cm', dcn.weight, _pack_dcn_colmajor), None, dcn.out_channels, 1,
                                    relu=True, out=st['f2'][lo:hi])
                else:
                    f2 = ops.deform_conv(f1, off, dcn._pk.get('w', dcn.weight, ops.pack_conv_weight, job=(False, None, None, None)),
                                         dcn.out_channels,
                                         dcn.deform_groups, relu=True, out=st['f2'][lo:hi])
                stage.fuse_transform_out.run(f2, relu=True, out=tail, out_ch_offset=0)
                if st['up'] is not None:
                    ops.upsample2x(tail, align_corners=False, relu=True, out=st['up'][lo:hi])
            nc = head.stage_num_classes[-1]
            c = head.final_instance_logits.in_channels
            ops.class_logits(x_last[lo:hi], head.final_instance_logits.weight.detach().view(nc, c),
                             head.final_instance_logits.bias.detach(), head.final_detail_logits.weight.detach().view(nc, c),
                             head.final_detail_logits.bias.detach(), lab_last[lo:hi], out=(fin_ip[lo:hi], fin_dp[lo:hi]))
            if fin_up is not None:
                ops.upsample2x(fin_ip[lo:hi], align_corners=True, out=fin_up[0][lo:hi])
                ops.upsample2x(fin_dp[lo:hi], align_corners=True, out=fin_up[1][lo:hi])

        second = side_stream(dev, 'coord') if n >= _FWD_SPLIT_MIN_ROIS else None       # idle during the forward
        if second is not None and sw.enabled:
            main = torch.cuda.current_stream(dev)
            h = (n + 1) // 2            # (0.44 .. 0.60 of the RoIs in the first half: 22.6-22.9 ms all the same, round 3)
            # kernel-layout weights are cached per module and refreshed after every optimizer step by whoever asks
            # first: refresh them here, on the main stream, before the fork (packed by one stream and read by the
            # other without a dependency would be a race)
            head.prepack(fused_dcn=[st['col'] is None for st in stbuf])
            for stage, st in zip(head.stages, stbuf):
                if st['col'] is not None:
                    stage.fuse_conv[1]._pk.get('w_cm', stage.fuse_conv[1].weight, _pack_dcn_colmajor)
            packed = main.record_event()            # ins_feats and the forward's packs are ready
            convs(0, h)
            second.wait_event(packed)
            with torch.cuda.stream(second):
                convs(h, n)
            if after_convs is not None:
                after_convs()
            sw.run(semantic_branches, after=ready)
            sw.join(*sems, *isfs)
            stages(0, h)
            with torch.cuda.stream(second):
                second.wait_stream(sw.side)         # the semantic branches
                stages(h, n)
            main.wait_stream(second)
        else:
            if after_convs is not None:
                after_convs()
            sw.run(semantic_branches, after=ready)
            convs(0, n)
            sw.join(*sems, *isfs)
            stages(0, n)
        # the backward's kernel-layout weights (transposed / rotated packs of every data gradient, the DCN
        # column-gradient GEMM's): beside the rest of the forward on the side stream, issued last; the main stream
        # joins it (the packs are read by the backward's chain and leaves)
        sw.run(lambda: _prepack_backward(head, feats), after=ready)
        sw.join()
        saved['conv_in'] = conv_in[:-1]
        saved['stages'] = []
        ips, dps = [], []
        for idx, st in enumerate(stbuf):
            saved['stages'].append(dict(xin=st['xin'], sem=sems[idx], isf=isfs[idx], tail=st['tail'], f1=st['f1'], off=st['off'],
                                        f2=st['f2'], up=st['up'], col=st['col'], feat_idx=st['feat_idx']))
            ips.append(st['ip'])
            dps.append(st['dp'])
        saved['x_last'] = x_last
        saved['lab_last'] = lab_last
        ips.append(fin_ip if fin_up is None else fin_up[0])
        dps.append(fin_dp if fin_up is None else fin_up[1])
        return ips, dps, saved, feats, rois, labels

    @staticmethod
    @hazard.backward_node
    def backward(ctx, *grads):
        head, sv, feats, rois, labels = ctx.head, ctx.saved, ctx.feats, ctx.rois, ctx.labels
        hazard.engine_handoff(*grads)
        dev = rois.device
        n_st = len(head.stages) + 1
        g_ips = [g.contiguous() if g is not None else None for g in grads[:n_st]]
        g_dps = [g.contiguous() if g is not None else None for g in grads[n_st:]]
        pgrad = {}                      # parameter -> gradient tensor
        g_feats = [None] * len(feats)
        sw = _SideWork(dev, alt='coord')

        def zeros(shape):
            return torch.zeros(shape, device=dev, dtype=torch.float32)

        def zl(g, like):
            return g if g is not None else torch.zeros_like(like)

        def conv_params_bwd(conv, dy, srcs, ks):
            params_grad(conv.weight, conv.bias, dy, srcs, ks, pgrad)

        def logit_grads(inst, det, nc, c):
            """Accumulation targets of class_logits_backward (zero-filled temporaries, or the flat views)."""
            outs, keep = [], []
            for p_, shape in ((inst.weight, (nc, c)), (inst.bias, (nc,)), (det.weight, (nc, c)), (det.bias, (nc,))):
                t = _direct(p_)
                if t is None:
                    t = zeros(shape)
                    keep.append((p_, t))
                outs.append(t.view(shape))
            return outs, keep

        def data_grad(conv, dy, lo, hi, ks, out=None, accumulate=False, mask=None):
            """d/d(input channels lo:hi) of a conv: forward kernel, transposed+rotated weights.  ``mask``: the ReLU
            output this gradient flows into -- its adjoint is applied in the conv's epilogue."""
            wq = _flip_pack(conv, lo, hi)
            if out is None and mask is not None:
                out = torch.empty_like(mask)
            return ops.conv2d(dy, wq, None, hi - lo, ks, out=out, accumulate=accumulate, mask=mask)

        # ---------------- final logits + x2 (align_corners=True) upsample
        x_last = sv['x_last']
        n = x_last.shape[0]
        S = x_last.shape[-1]
        if head.pre_upsample_last_stage:
            g_ip, g_dp = zl(g_ips[-1], x_last[:, :1]), zl(g_dps[-1], x_last[:, :1])
        else:
            full = (n, 1, 2 * S, 2 * S)
            g_ip = ops.upsample2x_backward(g_ips[-1] if g_ips[-1] is not None else zeros(full), None, (n, 1, S, S), True)
            g_dp = ops.upsample2x_backward(g_dps[-1] if g_dps[-1] is not None else zeros(full), None, (n, 1, S, S), True)
        fi, fd = head.final_instance_logits, head.final_detail_logits
        nc, c = head.stage_num_classes[-1], fi.in_channels
        g_x = torch.empty_like(x_last)
        (gwi, gbi, gwd, gbd), keep = logit_grads(fi, fd, nc, c)
        ops.class_logits_backward(x_last, fi.weight.detach().view(nc, c), fd.weight.detach().view(nc, c), sv['lab_last'],
                                  g_ip, g_dp, g_x, False, gwi, gbi, gwd, gbd)
        for p_, t in keep:
            pgrad[p_] = t.view_as(p_)

        # ---------------- SFM stages, last to first; g_x = grad wrt the stage's output
        for idx in reversed(range(len(head.stages))):
            stage, st = head.stages[idx], sv['stages'][idx]
            xin, sem, isf, tail, f1, off, f2, up = (st[k] for k in ('xin', 'sem', 'isf', 'tail', 'f1', 'off', 'f2', 'up'))
            c, co, s = stage.instance_in_channel, stage.instance_out_channel, stage.out_size
            if up is not None:
                g_tail = ops.upsample2x_backward(g_x, up, tuple(tail.shape), False)      # ReLU mask fused
            else:
                g_tail = g_x
            # tail = [relu(fuse_transform_out(f2)) | sigmoid(ip) | sigmoid(dp)]; sigmoids are > 0 so
            # masking the whole tensor by tail > 0 only touches the conv channels
            ops.relu_backward_(g_tail, tail)
            dy = g_tail[:, :co - 2]
            sw.run(lambda: conv_params_bwd(stage.fuse_transform_out, dy, f2, 1), g_tail, alt='selector')
            g_f2 = data_grad(stage.fuse_transform_out, dy, 0, c, 1, mask=f2)
            dcn = stage.fuse_conv[1]

            col = st.pop('col', None)          # allocated by the forward on the main stream: kept alive until the join

            def dcn_weight_grad():
                gw_dcn = ops.deform_conv_backward_weight(f1, off, g_f2, dcn.deform_groups, gw_accum=_direct(dcn.weight), col=col)
                if gw_dcn is not None:
                    pgrad[dcn.weight] = gw_dcn
            sw.run(dcn_weight_grad, g_f2, col, alt='selector')
            g_f1, g_off = ops.deform_conv_backward_data(f1, off, dcn.weight.detach(), g_f2, dcn.deform_groups,
                                                        side=side_stream(dev, 'coord'),
                                                        w_colgrad=dcn._pk.get('colgrad', dcn.weight, ops.pack_dcn_colgrad_weight))
            sw.run(lambda: conv_params_bwd(dcn.conv_offset, g_off, f1, 3), g_off, alt='selector')
            data_grad(dcn.conv_offset, g_off, 0, c, 3, out=g_f1, accumulate=True, mask=f1)
            f0 = stage.fuse_conv[0]
            sw.run(lambda: conv_params_bwd(f0, g_f1, [xin, isf, tail[:, co - 2:]], 1), g_f1)
            g_xin = data_grad(f0, g_f1, 0, c, 1)
            g_isf = data_grad(f0, g_f1, c, 2 * c, 1)
            g_sig = data_grad(f0, g_f1, 2 * c, 2 * c + 2, 1)
            # logits: direct loss gradient + through the two sigmoid channels (tail + fuse input)
            g_ip = zl(g_ips[idx], g_sig[:, :1]).clone() if g_ips[idx] is not None else zeros((n, 1, s, s))
            g_dp = zl(g_dps[idx], g_sig[:, :1]).clone() if g_dps[idx] is not None else zeros((n, 1, s, s))
            ops.sigmoid_backward(tail[:, co - 2:co - 1], g_tail[:, co - 2:co - 1], g_sig[:, 0:1], g_logit=g_ip)
            ops.sigmoid_backward(tail[:, co - 1:co], g_tail[:, co - 1:co], g_sig[:, 1:2], g_logit=g_dp)
            il, dl = stage.instance_logits, stage.detail_logits
            nc = stage.num_classes
            (gwi, gbi, gwd, gbd), keep = logit_grads(il, dl, nc, c)
            ops.class_logits_backward(xin, il.weight.detach().view(nc, c), dl.weight.detach().view(nc, c), labels, g_ip, g_dp,
                                      g_xin, True, gwi, gbi, gwd, gbd)
            for p_, t in keep:
                pgrad[p_] = t.view_as(p_)
            # semantic branch: point sample adjoint -> relu -> 1x1 conv on the FPN map
            def semantic_branch(stage=stage, st=st, sem=sem, g_isf=g_isf):
                g_sem = ops.point_sample_backward(g_isf, tuple(sem.shape), rois, stage.spatial_scale)
                ops.relu_backward_(g_sem, sem)
                fidx = st['feat_idx']
                feat = feats[fidx]
                conv_params_bwd(stage.semantic_transform_in, g_sem, feat, 1)
                if g_feats[fidx] is None:
                    g_feats[fidx] = data_grad(stage.semantic_transform_in, g_sem, 0, feat.shape[1], 1)
                else:
                    data_grad(stage.semantic_transform_in, g_sem, 0, feat.shape[1], 1, out=g_feats[fidx], accumulate=True)
            sw.run(semantic_branch, g_isf, alt=True)
            g_x = g_xin

        # ---------------- instance convs
        for i in reversed(range(len(head.instance_convs))):
            conv = head.instance_convs[i].conv
            x_in = sv['conv_in'][i]
            y = sv['conv_in'][i + 1] if i + 1 < len(sv['conv_in']) else sv['stages'][0]['xin']
            if i == len(head.instance_convs) - 1:
                ops.relu_backward_(g_x, y)          # (the earlier convs' masks ride in the data gradient below)
            g_y = g_x
            if i > 0:
                sw.run(lambda conv=conv, g_y=g_y, x_in=x_in: conv_params_bwd(conv, g_y, x_in, conv.kernel_size), g_y)
            # x_in of conv i > 0 is the ReLU output of conv i - 1: its adjoint is fused into this data gradient
            g_x = data_grad(conv, g_y, 0, conv.in_channels, conv.kernel_size, mask=x_in if i > 0 else None)
            if i == 0:
                # the chain ends here: its stream takes the last weight gradient itself instead of queueing it behind
                # the leaf stream's backlog (the step used to end with the chain's queue empty for 1.4 ms)
                conv_params_bwd(conv, g_y, x_in, conv.kernel_size)

        sw.join(*g_feats, *pgrad.values())
        params = list(head.parameters())
        out_p = [pgrad.get(p) for p in params]
        for p, g in zip(params, out_p):
            if g is not None and g.shape != p.shape:
                raise RuntimeError('gradient shape mismatch')
        return (None, None, None, g_x, None, *g_feats, *out_p)


def mask_head_forward_train(head, ins_feats, feats, rois, labels, between=None):
    """Differentiable ``DynaMaskHead.forward`` -> (stage_instance_preds, stage_detail_preds).

    ``between``: a callable run while the head's launches are being issued (right after its instance convolutions:
    the GPU has work queued, and the branch gets the rest of the forward to finish in) and BEFORE the head's autograd
    node is created.
    The autograd engine runs ready nodes newest first, and it issues one node's launches at a time: whatever is
    created inside ``between`` (the selector branch) gets older sequence numbers than the head's node, so in the
    backward the head -- the chain that decides the length of the step -- is issued first and the branch with slack
    after it, while in the forward the head's launches were issued first as well."""
    feats = list(feats)
    if between is not None:
        def branch():
            with torch.enable_grad():
                between()
        with torch.no_grad():
            _PRECOMPUTED[0] = None
            pre = MaskHeadFn.issue(head, rois, labels, ins_feats.detach(), [f.detach() for f in feats], after_convs=branch)
        _PRECOMPUTED[0] = pre
    else:
        _PRECOMPUTED[0] = None          # (never hand a stale hand-off to this call)
    # (Two half-batches on two streams -- the inference path's arrangement -- were measured here too: 25.8 ms
    # against 23.8 ms per step at 256 RoIs.  The backward already shares the GPU between the chain and its leaves;
    # halving every launch on top of that only adds tails.)
    try:
        outs = MaskHeadFn.apply(head, rois, labels, ins_feats, len(feats), *feats, *list(head.parameters()))
    finally:
        _PRECOMPUTED[0] = None          # consumed by forward(); cleared here too if the apply never reached it
    n = len(head.stages) + 1
    return list(outs[:n]), list(outs[n:])


def roi_extract_train(extractor, feats, rois):
    lay = extractor.roi_layers[0]
    feats = list(feats)[:extractor.num_inputs]
    scales = [l.spatial_scale for l in extractor.roi_layers]
    return RoIExtractFn.apply(rois, lay.output_size[0], scales, lay.sampling_ratio, float(extractor.finest_scale), *feats)


def _maskpre_tail_forward(mp, p1):
    """MaskPre behind its first pooling (base_roi_head.py:17-26): conv2 -> BN -> ReLU -> pool -> fc1 -> ReLU -> fc2."""
    y2 = mp.conv2.run(p1)
    m2, v2 = ops.bn_stats(y2, mp.bn2.running_mean, mp.bn2.running_var, mp.bn2.momentum)
    p2 = ops.bn_relu_maxpool(y2, m2, v2, mp.bn2.weight.detach(), mp.bn2.bias.detach(), mp.bn2.eps)
    mp.bn1.num_batches_tracked += 1
    mp.bn2.num_batches_tracked += 1
    h = mp.fc1.run(p2.reshape(p2.size(0), 3136), relu=True)
    return y2, m2, v2, p2, h, mp.fc2.run(h)


def _maskpre_tail_backward(mp, g, y1, m1, v1, p1, y2, m2, v2, p2, h, pg):
    """-> d loss / d y1 (the first BatchNorm's input); parameter gradients of everything behind conv1 into ``pg`` or,
    where the parameters have them, straight into the flat views."""
    n = y1.shape[0]

    def fc_bwd(fc, gy, xin):
        gy4 = gy.contiguous().view(n, fc.out_features, 1, 1)
        x4 = xin.contiguous().view(n, fc.in_features, 1, 1)
        params_grad(fc.weight, fc.bias, gy4, x4, 1, pg, (fc.out_features, fc.in_features, 1, 1))
        wq = fc._pk.get('flip', fc.weight, lambda t: ops.pack_conv_weight(
            t.view(fc.out_features, fc.in_features, 1, 1), transpose_flip=True))
        return ops.conv2d(gy4, wq, None, fc.in_features, 1).view(n, fc.in_features)

    g_h = fc_bwd(mp.fc2, g, h)
    ops.relu_backward_(g_h, h)
    g_p2 = fc_bwd(mp.fc1, g_h, p2.reshape(n, 3136)).view_as(p2).contiguous()
    g_y2, gg2, gb2 = ops.bn_relu_maxpool_backward(y2, m2, v2, mp.bn2.weight.detach(), mp.bn2.bias.detach(), g_p2, mp.bn2.eps)
    pg[mp.bn2.weight], pg[mp.bn2.bias] = gg2, gb2
    conv = mp.conv2
    params_grad(conv.weight, conv.bias, g_y2, p1, conv.kernel_size, pg, tuple(conv.weight.shape))
    wq = conv._pk.get(('flip', 0, conv.in_channels), conv.weight,
                      lambda t: ops.pack_conv_weight(t, transpose_flip=True), job=(True, None, None, None))
    g_p1 = ops.conv2d(g_y2, wq, None, conv.in_channels, conv.kernel_size)
    g_y1, gg1, gb1 = ops.bn_relu_maxpool_backward(y1, m1, v1, mp.bn1.weight.detach(), mp.bn1.bias.detach(), g_p1, mp.bn1.eps)
    pg[mp.bn1.weight], pg[mp.bn1.bias] = gg1, gb1
    return g_y1


class MaskPreFn(torch.autograd.Function):
    """MaskPre (base_roi_head.py:10-27) in train mode: forward + backward.
    Inputs: (mask_pre_module, x, *mask_pre.parameters()); x is the detached 56x56
    RoI feature (dynamask_roi_head.py:59), so no data gradient leaves the block.
    (The default training path since round 5 is MaskPreMapFn below; this form -- conv1 on the extracted [N, 256, 56, 56]
    tensor, as the reference computes it -- is what the deterministic mode and direct callers of
    ``get_mask_label(ins_semantic_feats)`` use.)"""

    @staticmethod
    def forward(ctx, mp, x, *params):
        x = x.contiguous()
        y1 = mp.conv1.run(x)
        m1, v1 = ops.bn_stats(y1, mp.bn1.running_mean, mp.bn1.running_var, mp.bn1.momentum)
        p1 = ops.bn_relu_maxpool(y1, m1, v1, mp.bn1.weight.detach(), mp.bn1.bias.detach(), mp.bn1.eps)
        y2, m2, v2, p2, h, logits = _maskpre_tail_forward(mp, p1)
        ctx.mp = mp
        ctx.sv = (x, y1, m1, v1, p1, y2, m2, v2, p2, h)
        return logits

    @staticmethod
    @hazard.backward_node
    def backward(ctx, g):
        mp = ctx.mp
        hazard.engine_handoff(g)
        x, y1, m1, v1, p1, y2, m2, v2, p2, h = ctx.sv
        pg = {}
        g_y1 = _maskpre_tail_backward(mp, g, y1, m1, v1, p1, y2, m2, v2, p2, h, pg)
        params_grad(mp.conv1.weight, mp.conv1.bias, g_y1, x, mp.conv1.kernel_size, pg, tuple(mp.conv1.weight.shape))
        _join_caller_after_backward(x.device)
        return (None, None, *[pg.get(p) for p in mp.parameters()])


class MaskPreMapFn(torch.autograd.Function):
    """MaskPre in train mode with its first convolution applied to the P2 MAP (round 5; SURVEY 2.3 K2: "best fused with
    K9 so the 3.2 MB / RoI tensor never hits HBM").

    ``conv1`` is 1x1 and the RoIAlign a linear interpolation: conv1(RoIAlign56(x)) = RoIAlign56(W1 x) + b1
    (base_roi_head.py:13, dynamask_roi_head.py:59; samples outside the map are 0 on both sides).  So W1 is applied once to
    [B, 256, 200, 336] (8.8 GFLOP instead of 52.6 on 256 x 3136 extracted pixels) and 128 channels are extracted, not 256:
    the [N, 256, 56, 56] tensor (822 MB at the training shape, written once and read twice) no longer exists.  Train-mode
    BatchNorm cancels b1, which therefore only enters the running mean (``mean_shift`` of dm_bn_stats).  Backward: conv1's
    weight gradient is taken on the map -- G = adjoint of the extraction applied to d loss / d y1 (dm_roi_align_bwd's
    gather form: one atomic per footprint cell), dW1 = G . x^T -- its bias gradient is the channel sums of d loss / d y1
    (a rounding residue, as in the reference: BatchNorm's backward sums to zero).  x[0] is detached in the reference, so no
    data gradient leaves the block.  Sums differ from MaskPreFn's by association only; the deterministic mode
    (ops.DETERMINISTIC) keeps MaskPreFn (the adjoint's float atomics land in arrival order).
    Inputs: (mask_pre, p2_map, rois, output_size, spatial_scale, sampling_ratio, *mask_pre.parameters())."""

    @staticmethod
    def forward(ctx, mp, feat_map, rois, output_size, spatial_scale, sampling_ratio, *params):
        feat_map = feat_map.contiguous()
        rois = rois.contiguous()
        c1 = mp.conv1
        y_map = ops.conv2d([feat_map], c1.packed([feat_map.shape[1]]), None, c1.out_channels, 1)      # W1 x, no bias
        y1 = ops.roi_align([y_map], rois, output_size, [spatial_scale], sampling_ratio)               # = conv1(RoIAlign(x)) - b1
        map_shape = tuple(y_map.shape)
        del y_map
        m1, v1 = ops.bn_stats(y1, mp.bn1.running_mean, mp.bn1.running_var, mp.bn1.momentum, mean_shift=c1.bias.detach())
        p1 = ops.bn_relu_maxpool(y1, m1, v1, mp.bn1.weight.detach(), mp.bn1.bias.detach(), mp.bn1.eps)
        y2, m2, v2, p2, h, logits = _maskpre_tail_forward(mp, p1)
        ctx.mp = mp
        ctx.sv = (feat_map, rois, y1, m1, v1, p1, y2, m2, v2, p2, h)
        ctx.cfg = (output_size, spatial_scale, sampling_ratio, map_shape)
        return logits

    @staticmethod
    @hazard.backward_node
    def backward(ctx, g):
        mp = ctx.mp
        hazard.engine_handoff(g)
        feat_map, rois, y1, m1, v1, p1, y2, m2, v2, p2, h = ctx.sv
        output_size, spatial_scale, sampling_ratio, map_shape = ctx.cfg
        pg = {}
        g_y1 = _maskpre_tail_backward(mp, g, y1, m1, v1, p1, y2, m2, v2, p2, h, pg)
        c1 = mp.conv1
        tb = _direct(c1.bias)
        if tb is not None:
            ops.channel_sum(g_y1, out=tb)
        else:
            pg[c1.bias] = ops.channel_sum(g_y1)
        (g_map,) = ops.roi_align_backward(g_y1, [map_shape], rois, output_size, [spatial_scale], sampling_ratio)
        del g_y1
        tw = _direct(c1.weight)
        gw = ops.conv2d_wgrad(g_map, feat_map, 1, dw=tw)
        if tw is None:
            pg[c1.weight] = gw.view_as(c1.weight)
        _join_caller_after_backward(feat_map.device)
        return (None, None, None, None, None, None, *[pg.get(p) for p in mp.parameters()])


class GumbelSelectFn(torch.autograd.Function):
    """Straight-through Gumbel-softmax (hard), dynamask_roi_head.py:97-114."""

    @staticmethod
    def forward(ctx, logits, noise, temperature):
        y, hot, idx = ops.gumbel_select(logits.contiguous(), noise.contiguous(), temperature)
        ctx.save_for_backward(y)
        ctx.t = temperature
        ctx.mark_non_differentiable(idx)
        return hot, idx

    @staticmethod
    @hazard.backward_node
    def backward(ctx, g_hot, _g_idx):
        (y,) = ctx.saved_tensors
        hazard.engine_handoff(g_hot)
        return ops.gumbel_select_backward(y, g_hot.contiguous(), ctx.t), None, None


class FCNMaskHeadFn(torch.autograd.Function):
    """FCNMaskHead.forward (fcn_mask_head.py:117-126) with a hand-sequenced backward: conv stack,
    upsample (deconv / CARAFE / bilinear / nearest) and the 1x1 logits conv.  Inputs: (head, x, *head.parameters())."""

    @staticmethod
    def forward(ctx, head, x, *params):
        acts = [x.detach().contiguous()]
        for conv in head.convs:
            acts.append(conv(acts[-1]))
        h = acts[-1]
        up = head.upsample
        aux = None
        if up is None:
            u = h
        elif head.upsample_method == 'carafe':
            comp = up.channel_compressor.run(h)
            enc = up.content_encoder.run(comp)
            u = ops.carafe(h, enc, up.up_kernel, up.up_group, up.scale_factor)
            aux = (comp, enc)
        else:
            u = up(h, relu=(head.upsample_method == 'deconv'))
        out = head.conv_logits.run(u)
        ctx.head, ctx.acts, ctx.u, ctx.aux, ctx.need_x = head, acts, u, aux, x.requires_grad
        return out

    @staticmethod
    @hazard.backward_node
    def backward(ctx, g):
        head, acts, u, aux = ctx.head, ctx.acts, ctx.u, ctx.aux
        hazard.engine_handoff(g)
        pg = {}

        def params_bwd(conv, dy, xin, ks):
            params_grad(conv.weight, conv.bias, dy, xin, ks, pg)

        def data_grad(conv, dy, ks, out=None, accumulate=False):
            wq = conv._pk.get(('flip', 0, conv.in_channels), conv.weight,
                              lambda t: ops.pack_conv_weight(t, transpose_flip=True), job=(True, None, None, None))
            return ops.conv2d(dy, wq, None, conv.in_channels, ks, out=out, accumulate=accumulate)

        g = g.contiguous()
        params_bwd(head.conv_logits, g, u, 1)
        g_u = data_grad(head.conv_logits, g, 1)
        h = acts[-1]
        up = head.upsample
        m = head.upsample_method
        if up is None:
            g_h = g_u
        elif m == 'deconv':
            ops.relu_backward_(g_u, u)
            gyu = ops.pixel_unshuffle2x(g_u)                                   # [N, 4*Cout, H, W]
            cin, cout = up.in_channels, up.out_channels
            dwp = ops.conv2d_wgrad(gyu, h, 1)                                  # [(d, co), ci, 1, 1]
            pg[up.weight] = dwp.view(2, 2, cout, cin).permute(3, 2, 0, 1).contiguous()
            pg[up.bias] = ops.channel_sum(g_u)
            wb = up._pk.get('bwd', up.weight, lambda t: ops.pack_conv_weight(
                t.permute(0, 2, 3, 1).reshape(cin, 4 * cout, 1, 1).contiguous()))
            g_h = ops.conv2d(gyu, wb, None, cin, 1)
        elif m == 'carafe':
            comp, enc = aux
            g_h, g_enc = ops.carafe_backward(h, enc, g_u, up.up_kernel, up.up_group, up.scale_factor)
            params_bwd(up.content_encoder, g_enc, comp, 3)
            g_comp = data_grad(up.content_encoder, g_enc, 3)
            params_bwd(up.channel_compressor, g_comp, h, 1)
            data_grad(up.channel_compressor, g_comp, 1, out=g_h, accumulate=True)
        elif m == 'bilinear':
            g_h = ops.upsample2x_backward(g_u, None, tuple(h.shape), False)
        else:
            g_h = ops.upsample2x_nearest_backward(g_u)
        for i in reversed(range(len(head.convs))):
            conv = head.convs[i].conv
            ops.relu_backward_(g_h, acts[i + 1])
            params_bwd(conv, g_h, acts[i], conv.kernel_size)
            if i > 0 or ctx.need_x:
                g_h = data_grad(conv, g_h, conv.kernel_size)
        g_x = g_h if ctx.need_x else None
        return (None, g_x, *[pg.get(p) for p in head.parameters()])
