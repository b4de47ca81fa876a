"""RoI assignment and sampling in front of the mask path at training time -- what
``DynaMaskRoIHead.forward_train`` (dynamask_roi_head.py:21-38) runs before the heads.

Mirrors, under the reference's registry names and kwargs:
``BboxOverlaps2D`` (core/bbox/iou_calculators/iou2d_calculator.py:6-34), ``MaxIoUAssigner``
(core/bbox/assigners/max_iou_assigner.py:9-212), ``AssignResult``
(assigners/assign_result.py), ``RandomSampler`` / ``BaseSampler``
(samplers/random_sampler.py:7-78, base_sampler.py:8-101) and ``SamplingResult``
(samplers/sampling_result.py:6-56).  The IoU matrix and the assignment run in
``dm_bbox_overlaps`` / ``dm_max_iou_assign``; the sampler only permutes and gathers indices
(torch index plumbing, as in the reference).
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
    """sampling_result.py:21-56."""

    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags):
        self.pos_inds = pos_inds
        self.neg_inds = neg_inds
        self.pos_bboxes = bboxes[pos_inds]
        self.neg_bboxes = bboxes[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        if gt_bboxes.numel() == 0:
            assert self.pos_assigned_gt_inds.numel() == 0
            self.pos_gt_bboxes = torch.empty_like(gt_bboxes).view(-1, 4)
        else:
            if len(gt_bboxes.shape) < 2:
                gt_bboxes = gt_bboxes.view(-1, 4)
            self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds, :]
        self.pos_gt_labels = assign_result.labels[pos_inds] if assign_result.labels is not None else None

    @property
    def bboxes(self):
        return torch.cat([self.pos_bboxes, self.neg_bboxes])


@BBOX_SAMPLERS.register_module()
class RandomSampler:
    """random_sampler.py + base_sampler.py.  ``cpu_rng=True`` draws the permutations on torch's
    CPU generator and copies the indices to the device (reproducible against a CPU run of the
    reference with the same seed; the reference draws on the boxes' device)."""

    def __init__(self, num, pos_fraction, neg_pos_ub=-1, add_gt_as_proposals=True, cpu_rng=False, **kwargs):
        self.num = num
        self.pos_fraction = pos_fraction
        self.neg_pos_ub = neg_pos_ub
        self.add_gt_as_proposals = add_gt_as_proposals
        self.cpu_rng = cpu_rng
        self.pos_sampler = self
        self.neg_sampler = self

    def random_choice(self, gallery, num):
        assert len(gallery) >= num
        if self.cpu_rng:
            perm = torch.randperm(gallery.numel())[:num].to(gallery.device)
        else:
            perm = torch.randperm(gallery.numel(), device=gallery.device)[:num]
        return gallery[perm]

    def _sample_pos(self, assign_result, num_expected, **kwargs):
        pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False)
        if pos_inds.numel() != 0:
            pos_inds = pos_inds.squeeze(1)
        if pos_inds.numel() <= num_expected:
            return pos_inds
        return self.random_choice(pos_inds, num_expected)

    def _sample_neg(self, assign_result, num_expected, **kwargs):
        neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False)
        if neg_inds.numel() != 0:
            neg_inds = neg_inds.squeeze(1)
        if len(neg_inds) <= num_expected:
            return neg_inds
        return self.random_choice(neg_inds, num_expected)

    def sample(self, assign_result, bboxes, gt_bboxes, gt_labels=None, **kwargs):
        """base_sampler.py:35-101."""
        if len(bboxes.shape) < 2:
            bboxes = bboxes[None, :]
        bboxes = bboxes[:, :4]
        gt_flags = bboxes.new_zeros((bboxes.shape[0],), dtype=torch.uint8)
        if self.add_gt_as_proposals and len(gt_bboxes) > 0:
            if gt_labels is None:
                raise ValueError('gt_labels must be given when add_gt_as_proposals is True')
            bboxes = torch.cat([gt_bboxes, bboxes], dim=0)
            assign_result.add_gt_(gt_labels)
            gt_ones = bboxes.new_ones(gt_bboxes.shape[0], dtype=torch.uint8)
            gt_flags = torch.cat([gt_ones, gt_flags])
        num_expected_pos = int(self.num * self.pos_fraction)
        pos_inds = self.pos_sampler._sample_pos(assign_result, num_expected_pos, bboxes=bboxes, **kwargs)
        pos_inds = pos_inds.unique()
        num_sampled_pos = pos_inds.numel()
        num_expected_neg = self.num - num_sampled_pos
        if self.neg_pos_ub >= 0:
            _pos = max(1, num_sampled_pos)
            neg_upper_bound = int(self.neg_pos_ub * _pos)
            if num_expected_neg > neg_upper_bound:
                num_expected_neg = neg_upper_bound
        neg_inds = self.neg_sampler._sample_neg(assign_result, num_expected_neg, bboxes=bboxes, **kwargs)
        neg_inds = neg_inds.unique()
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags)
