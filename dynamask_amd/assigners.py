"""RoI assignment and sampling in front of the mask path at training time -- what
``DynaMaskRoIHead.forward_train`` (dynamask_roi_head.py:21-38) runs before the heads.

Mirrors, under the reference's registry names and kwargs:
``BboxOverlaps2D`` (core/bbox/iou_calculators/iou2d_calculator.py:6-34), ``MaxIoUAssigner``
(core/bbox/assigners/max_iou_assigner.py:9-212), ``AssignResult``
(assigners/assign_result.py), ``RandomSampler`` / ``BaseSampler``
(samplers/random_sampler.py:7-78, base_sampler.py:8-101) and ``SamplingResult``
(samplers/sampling_result.py:6-56).  The IoU matrix and the assignment run in
``dm_bbox_overlaps`` / ``dm_max_iou_assign``; the sampling (quota arithmetic, the random subset, the
ordered compaction and every gather of ``SamplingResult``) is one launch of ``dm_random_sample``: each
candidate carries a random key and a class over its quota keeps the smallest keys.
"""
import torch

from . import ops
from .registry import Registry, build_from_cfg

BBOX_ASSIGNERS = Registry('bbox_assigner')
BBOX_SAMPLERS = Registry('bbox_sampler')
IOU_CALCULATORS = Registry('IoU calculator')


def build_assigner(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_ASSIGNERS, default_args)


def build_sampler(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_SAMPLERS, default_args)


def build_iou_calculator(cfg, default_args=None):
    return build_from_cfg(cfg, IOU_CALCULATORS, default_args)


@IOU_CALCULATORS.register_module()
class BboxOverlaps2D:
    """iou2d_calculator.py:6-34: accepts [n, 4] or [n, 5] (score column dropped)."""

    def __call__(self, bboxes1, bboxes2, mode='iou', is_aligned=False):
        assert bboxes1.size(-1) in [0, 4, 5] and bboxes2.size(-1) in [0, 4, 5]
        if is_aligned:
            raise NotImplementedError('the RoI head uses the full [n1, n2] matrix')
        if bboxes2.size(-1) == 5:
            bboxes2 = bboxes2[..., :4]
        if bboxes1.size(-1) == 5:
            bboxes1 = bboxes1[..., :4]
        return ops.bbox_overlaps(bboxes1.float().contiguous(), bboxes2.float().contiguous(), mode)


class AssignResult:
    """assign_result.py: num_gts, gt_inds (0 = negative, -1 = ignore, i+1 = gt i), max_overlaps, labels."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts = num_gts
        self.gt_inds = gt_inds
        self.max_overlaps = max_overlaps
        self.labels = labels

    @property
    def num_preds(self):
        return len(self.gt_inds)

    def add_gt_(self, gt_labels):
        """assign_result.py:190-204."""
        self_inds = torch.arange(1, len(gt_labels) + 1, dtype=torch.long, device=gt_labels.device)
        self.gt_inds = torch.cat([self_inds, self.gt_inds])
        self.max_overlaps = torch.cat([self.max_overlaps.new_ones(len(gt_labels)), self.max_overlaps])
        if self.labels is not None:
            self.labels = torch.cat([gt_labels, self.labels])


@BBOX_ASSIGNERS.register_module()
class MaxIoUAssigner:
    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True, match_low_quality=True, gpu_assign_thr=-1,
                 iou_calculator=dict(type='BboxOverlaps2D')):
        self.pos_iou_thr = pos_iou_thr
        self.neg_iou_thr = neg_iou_thr
        self.min_pos_iou = min_pos_iou
        self.gt_max_assign_all = gt_max_assign_all
        self.ignore_iof_thr = ignore_iof_thr
        self.ignore_wrt_candidates = ignore_wrt_candidates
        self.gpu_assign_thr = gpu_assign_thr          # accepted; the assignment always stays on the device
        self.match_low_quality = match_low_quality
        self.iou_calculator = build_iou_calculator(iou_calculator)

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        """max_iou_assigner.py:58-127."""
        overlaps = self.iou_calculator(gt_bboxes, bboxes)
        if (self.ignore_iof_thr > 0 and gt_bboxes_ignore is not None and gt_bboxes_ignore.numel() > 0
                and bboxes.numel() > 0):
            # :107-118: candidates inside an ignore region (IoF over the candidate, or over the region) are taken out
            if self.ignore_wrt_candidates:
                iof = self.iou_calculator(bboxes, gt_bboxes_ignore, mode='iof')
            else:
                iof = self.iou_calculator(gt_bboxes_ignore, bboxes, mode='iof')
            if overlaps.numel() > 0:
                ops.ignore_columns_(overlaps, iof, self.ignore_iof_thr, boxes_major=self.ignore_wrt_candidates)
        return self.assign_wrt_overlaps(overlaps, gt_labels)

    def assign_wrt_overlaps(self, overlaps, gt_labels=None):
        """max_iou_assigner.py:129-212."""
        num_gts, num_bboxes = overlaps.size(0), overlaps.size(1)
        if num_gts == 0 or num_bboxes == 0:
            assigned_gt_inds = overlaps.new_full((num_bboxes,), -1, dtype=torch.long)
            max_overlaps = overlaps.new_zeros((num_bboxes,))
            if num_gts == 0:
                assigned_gt_inds[:] = 0
            labels = None if gt_labels is None else overlaps.new_full((num_bboxes,), -1, dtype=torch.long)
            return AssignResult(num_gts, assigned_gt_inds, max_overlaps, labels=labels)
        gt_inds, max_overlaps, labels = ops.max_iou_assign(
            overlaps.contiguous(), self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou, self.match_low_quality,
            self.gt_max_assign_all, None if gt_labels is None else gt_labels.long().contiguous())
        return AssignResult(num_gts, gt_inds, max_overlaps, labels=labels)


class SamplingResult:
    """What the heads read from a sampled image (the fields of sampling_result.py:21-56): views of
    the fixed-capacity buffers ``dm_random_sample`` filled, cut to the kept counts."""

    __slots__ = ('pos_inds', 'neg_inds', 'pos_bboxes', 'neg_bboxes', 'pos_is_gt', 'num_gts', 'pos_assigned_gt_inds',
                 'pos_gt_bboxes', 'pos_gt_labels')

    def __init__(self, buffers, n_pos, n_neg, num_gts):
        for name in ('pos_inds', 'pos_bboxes', 'pos_is_gt', 'pos_assigned_gt_inds', 'pos_gt_bboxes'):
            setattr(self, name, buffers[name][:n_pos])
        for name in ('neg_inds', 'neg_bboxes'):
            setattr(self, name, buffers[name][:n_neg])
        lab = buffers['pos_gt_labels']
        self.pos_gt_labels = None if lab is None else lab[:n_pos]
        self.num_gts = num_gts

    @property
    def bboxes(self):
        return torch.cat([self.pos_bboxes, self.neg_bboxes])


class _PendingSample:
    """Buffers of a ``random_sample`` launch whose counts have not been read yet (RandomSampler.sample_deferred)."""

    def __init__(self, buffers, num_gts):
        self.buffers, self.num_gts = buffers, num_gts


@BBOX_SAMPLERS.register_module()
class RandomSampler:
    """``RandomSampler(num, pos_fraction, neg_pos_ub=-1, add_gt_as_proposals=True)`` of the reference's config
    (random_sampler.py:7-30, base_sampler.py:11-22); ``sample`` has the signature of base_sampler.py:35-40.

    The random subset is drawn as a selection by key on the device.  By default the keys are one
    ``torch.rand`` tensor over the candidates (no host round trip before the launch).  ``cpu_rng=True`` is the
    test hook: the keys are the inverse of ``torch.randperm(class size)`` drawn on torch's CPU generator --
    once per class that exceeds its quota, positives first, exactly the draws a CPU run of the reference makes --
    so that the kept indices equal the reference's under the same seed (golden g11)."""

    def __init__(self, num, pos_fraction, neg_pos_ub=-1, add_gt_as_proposals=True, cpu_rng=False, **kwargs):
        self.num = int(num)
        self.pos_fraction = pos_fraction
        self.neg_pos_ub = neg_pos_ub
        self.add_gt_as_proposals = add_gt_as_proposals
        self.cpu_rng = cpu_rng

    def _quota_neg(self, kept_pos):
        q = self.num - kept_pos
        if self.neg_pos_ub >= 0:
            q = min(q, int(self.neg_pos_ub * max(1, kept_pos)))
        return q

    def _reference_order_keys(self, gt_inds, quota_pos):
        """Test hook (one host sync): per class, key[r] = position of the class's r-th member in the permutation
        the reference would draw; a class within its quota draws nothing."""
        dev = gt_inds.device
        n_pos, n_neg = torch.stack([(gt_inds > 0).sum(), (gt_inds == 0).sum()]).tolist()
        keys = []
        for size, quota in ((n_pos, quota_pos), (n_neg, self._quota_neg(min(n_pos, quota_pos)))):
            k = torch.zeros(max(size, 1), dtype=torch.float32)
            if size > quota:
                k[torch.randperm(size)] = torch.arange(size, dtype=torch.float32)
            keys.append(k.to(dev))
        return keys

    def sample(self, assign_result, bboxes, gt_bboxes, gt_labels=None, _defer=False, **kwargs):
        boxes = bboxes.reshape(-1, bboxes.shape[-1])[:, :4]
        n_prepended = 0
        if self.add_gt_as_proposals and len(gt_bboxes) > 0:
            if gt_labels is None:
                raise ValueError('gt_labels must be given when add_gt_as_proposals is True')
            assign_result.add_gt_(gt_labels)          # base_sampler.py:78: the caller's AssignResult grows too
            n_prepended = gt_bboxes.shape[0]
            boxes = torch.cat([gt_bboxes, boxes], dim=0)
        boxes = boxes.float().contiguous()
        gt_inds = assign_result.gt_inds.contiguous()
        quota_pos = int(self.num * self.pos_fraction)
        if gt_inds.numel() == 0:
            empty = dict(pos_inds=gt_inds.new_zeros((0,)), neg_inds=gt_inds.new_zeros((0,)),
                         pos_bboxes=boxes.new_zeros((0, 4)), neg_bboxes=boxes.new_zeros((0, 4)),
                         pos_gt_bboxes=boxes.new_zeros((0, 4)), pos_assigned_gt_inds=gt_inds.new_zeros((0,)),
                         pos_gt_labels=None if assign_result.labels is None else gt_inds.new_zeros((0,)),
                         pos_is_gt=boxes.new_zeros((0,), dtype=torch.uint8))
            return SamplingResult(empty, 0, 0, gt_bboxes.shape[0])
        if self.cpu_rng:
            pos_keys, neg_keys = self._reference_order_keys(gt_inds, quota_pos)
        else:
            pos_keys = neg_keys = torch.rand(gt_inds.shape[0], device=gt_inds.device)
        labels = None if assign_result.labels is None else assign_result.labels.contiguous()
        buffers = ops.random_sample(gt_inds, boxes, n_prepended, gt_bboxes.float().reshape(-1, 4).contiguous(), labels,
                                    pos_keys, neg_keys, self.cpu_rng, self.num, quota_pos, float(self.neg_pos_ub))
        if _defer:
            return _PendingSample(buffers, gt_bboxes.shape[0])
        n_pos, n_neg = buffers['counts'][:2].tolist()          # the one host sync: the heads' tensors are sized by it
        return SamplingResult(buffers, n_pos, n_neg, gt_bboxes.shape[0])

    def sample_deferred(self, assign_result, bboxes, gt_bboxes, gt_labels=None, **kwargs):
        """``sample`` without its host sync: the kernels are enqueued, the result is ``finish_samples``' to size.  A caller
        with several images (``DynaMaskRoIHead.forward_train``) defers them all and waits ONCE for all the counts."""
        return self.sample(assign_result, bboxes, gt_bboxes, gt_labels, _defer=True, **kwargs)

    @staticmethod
    def finish_samples(pending):
        """SamplingResults of ``sample_deferred`` calls: one device -> host read for the (positives, negatives) of all."""
        live = [p for p in pending if isinstance(p, _PendingSample)]
        counts = torch.stack([p.buffers['counts'][:2] for p in live]).tolist() if live else []
        it = iter(counts)
        out = []
        for p in pending:
            if isinstance(p, _PendingSample):
                n_pos, n_neg = next(it)
                out.append(SamplingResult(p.buffers, n_pos, n_neg, p.num_gts))
            else:
                out.append(p)                                   # (an image without proposals: already a SamplingResult)
        return out
