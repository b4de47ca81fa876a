"""The reference's own MaxIoUAssigner tests (/root/reference/tests/test_assigner.py:14-152: inputs and expected
assignments, copied as data) run against the device assigner: the plain case, ignore regions, and the corner cases
an image or a network produces -- no ground truth, no boxes, neither."""
import pytest
import torch

pytestmark = pytest.mark.gpu

BBOXES = [[0, 0, 10, 10], [10, 10, 20, 20], [5, 5, 15, 15], [32, 32, 38, 42]]
GT = [[0, 0, 10, 9], [0, 10, 10, 19]]


@pytest.fixture(scope='module')
def Assigner():
    from dynamask_amd.assigners import MaxIoUAssigner
    return MaxIoUAssigner


def _t(rows, dtype=torch.float32):
    return torch.tensor(rows, dtype=dtype).cuda() if rows else torch.empty((0,), dtype=dtype).cuda()


def test_max_iou_assigner(Assigner):
    self = Assigner(pos_iou_thr=0.5, neg_iou_thr=0.5)
    r = self.assign(_t(BBOXES), _t(GT), gt_labels=_t([2, 3], torch.long))
    assert len(r.gt_inds) == 4 and len(r.labels) == 4
    assert torch.all(r.gt_inds.cpu() == torch.LongTensor([1, 0, 2, 0]))


def test_max_iou_assigner_with_ignore(Assigner):
    self = Assigner(pos_iou_thr=0.5, neg_iou_thr=0.5, ignore_iof_thr=0.5, ignore_wrt_candidates=False)
    bboxes = [[0, 0, 10, 10], [10, 10, 20, 20], [5, 5, 15, 15], [30, 32, 40, 42]]
    r = self.assign(_t(bboxes), _t(GT), gt_bboxes_ignore=_t([[30, 30, 40, 40]]))
    assert torch.all(r.gt_inds.cpu() == torch.LongTensor([1, 0, 2, -1]))


def test_max_iou_assigner_with_empty_gt(Assigner):
    self = Assigner(pos_iou_thr=0.5, neg_iou_thr=0.5)
    r = self.assign(_t(BBOXES), _t([]))
    assert torch.all(r.gt_inds.cpu() == torch.LongTensor([0, 0, 0, 0]))


def test_max_iou_assigner_with_empty_boxes(Assigner):
    self = Assigner(pos_iou_thr=0.5, neg_iou_thr=0.5)
    bboxes = torch.empty((0, 4)).cuda()
    r = self.assign(bboxes, _t(GT), gt_labels=_t([2, 3], torch.long))
    assert len(r.gt_inds) == 0 and tuple(r.labels.shape) == (0,)
    r = self.assign(bboxes, _t(GT), gt_labels=None)
    assert len(r.gt_inds) == 0 and r.labels is None


def test_max_iou_assigner_with_empty_boxes_and_ignore(Assigner):
    self = Assigner(pos_iou_thr=0.5, neg_iou_thr=0.5, ignore_iof_thr=0.5)
    bboxes = torch.empty((0, 4)).cuda()
    ign = _t([[30, 30, 40, 40]])
    r = self.assign(bboxes, _t(GT), gt_labels=_t([2, 3], torch.long), gt_bboxes_ignore=ign)
    assert len(r.gt_inds) == 0 and tuple(r.labels.shape) == (0,)
    r = self.assign(bboxes, _t(GT), gt_labels=None, gt_bboxes_ignore=ign)
    assert len(r.gt_inds) == 0 and r.labels is None


def test_max_iou_assigner_with_empty_boxes_and_gt(Assigner):
    self = Assigner(pos_iou_thr=0.5, neg_iou_thr=0.5)
    r = self.assign(torch.empty((0, 4)).cuda(), torch.empty((0, 4)).cuda())
    assert len(r.gt_inds) == 0


def test_bbox_overlaps_docstring_examples():
    """iou2d_calculator.py:55-79 (the reference's docstring: inputs and printed values as data), incl. empty inputs."""
    from dynamask_amd.assigners import BboxOverlaps2D
    iou = BboxOverlaps2D()
    b1 = _t([[0, 0, 10, 10], [10, 10, 20, 20], [32, 32, 38, 42]])
    b2 = _t([[0, 0, 10, 20], [0, 10, 10, 19], [10, 10, 20, 20]])
    torch.testing.assert_close(iou(b1, b2).cpu(), torch.tensor([[0.5, 0., 0.], [0., 0., 1.], [0., 0., 0.]]), atol=1e-4, rtol=0)
    empty, nonempty = torch.empty((0, 4)).cuda(), _t([[0, 0, 10, 9]])
    assert tuple(iou(empty, nonempty).shape) == (0, 1)
    assert tuple(iou(nonempty, empty).shape) == (1, 0)
    assert tuple(iou(empty, empty).shape) == (0, 0)


# ---------------------------------------------------------------- dm_random_sample (key-ranked RoI sampling)
def _select_by_key_numpy(gt_inds, keys, num, quota_pos, neg_pos_ub):
    """Brute force of the kernel's contract: per class keep everything within the quota, else the quota's
    smallest keys (ties: lower index); kept boxes in ascending index order."""
    import numpy as np
    pos = np.nonzero(gt_inds > 0)[0]
    neg = np.nonzero(gt_inds == 0)[0]

    def pick(cand, quota):
        if len(cand) <= quota:
            return cand
        order = np.lexsort((cand, keys[cand]))          # key, then index
        return np.sort(cand[order[:quota]])
    p = pick(pos, quota_pos)
    q = num - len(p)
    if neg_pos_ub >= 0:
        q = min(q, int(neg_pos_ub * max(1, len(p))))
    return p, pick(neg, max(q, 0))


@pytest.mark.parametrize('M,G,num,frac,ub', [(1000, 7, 512, 0.25, -1), (2500, 30, 512, 0.25, 3), (300, 4, 512, 0.25, -1),
                                              (1027, 1, 64, 0.5, 0.7), (5, 2, 8, 0.25, -1), (4097, 3, 256, 0.1, -1)])
def test_random_sample_is_the_selection_by_key(M, G, num, frac, ub):
    import numpy as np
    from dynamask_amd import ops
    g = torch.Generator().manual_seed(M + num)
    gt_inds = torch.randint(-1, G + 1, (M,), generator=g)
    gt_inds[torch.rand(M, generator=g) < 0.5] = 0
    boxes = torch.rand(M, 4, generator=g) * 100
    gtb = torch.rand(G, 4, generator=g) * 100
    labels = torch.randint(0, 80, (M,), generator=g)
    keys = torch.rand(M, generator=g)
    keys[::7] = keys[3]                                   # ties: the lower index wins
    n_pre = min(G, M)
    quota_pos = int(num * frac)
    out = ops.random_sample(gt_inds.cuda(), boxes.cuda(), n_pre, gtb.cuda(), labels.cuda(), keys.cuda(), keys.cuda(), False,
                            num, quota_pos, float(ub))
    n_pos, n_neg, c_pos, c_neg = out['counts'].tolist()
    p, q = _select_by_key_numpy(gt_inds.numpy(), keys.numpy(), num, quota_pos, ub)
    assert (c_pos, c_neg) == (int((gt_inds > 0).sum()), int((gt_inds == 0).sum()))
    np.testing.assert_array_equal(out['pos_inds'][:n_pos].cpu().numpy(), p)
    np.testing.assert_array_equal(out['neg_inds'][:n_neg].cpu().numpy(), q)
    pi = torch.from_numpy(p)
    assert torch.equal(out['pos_bboxes'][:n_pos].cpu(), boxes[pi])
    assert torch.equal(out['neg_bboxes'][:n_neg].cpu(), boxes[torch.from_numpy(q)])
    assert torch.equal(out['pos_assigned_gt_inds'][:n_pos].cpu(), gt_inds[pi] - 1)
    assert torch.equal(out['pos_gt_bboxes'][:n_pos].cpu(), gtb[gt_inds[pi] - 1])
    assert torch.equal(out['pos_gt_labels'][:n_pos].cpu(), labels[pi])
    assert torch.equal(out['pos_is_gt'][:n_pos].cpu(), (pi < n_pre).to(torch.uint8))


def test_random_sampler_device_noise_is_a_valid_uniform_sample():
    """Product mode (keys = torch.rand on the device): sizes follow the quotas, indices are sorted members of
    their class, and over repeated draws every positive is kept about equally often."""
    from dynamask_amd.assigners import AssignResult, RandomSampler
    M, G = 600, 5
    g = torch.Generator().manual_seed(3)
    gt_inds = torch.zeros(M, dtype=torch.long)
    gt_inds[torch.randperm(M, generator=g)[:200]] = torch.randint(1, G + 1, (200,), generator=g)
    props = (torch.rand(M, 4, generator=g) * 50).cuda()
    gtb = (torch.rand(G, 4, generator=g) * 50).cuda()
    smp = RandomSampler(num=128, pos_fraction=0.25, add_gt_as_proposals=True)
    hits = torch.zeros(M + G)
    for _ in range(200):
        ar = AssignResult(G, gt_inds.cuda(), torch.zeros(M).cuda(), labels=torch.zeros(M, dtype=torch.long).cuda())
        sr = smp.sample(ar, props, gtb, torch.arange(G).cuda())
        assert len(sr.pos_inds) == 32 and len(sr.neg_inds) == 96 and sr.bboxes.shape == (128, 4)
        pi = sr.pos_inds.cpu()
        assert torch.all(pi[1:] > pi[:-1]) and torch.all(ar.gt_inds.cpu()[pi] > 0)
        assert torch.all(ar.gt_inds.cpu()[sr.neg_inds.cpu()] == 0)
        assert torch.equal(sr.pos_is_gt.cpu().bool(), pi < G)
        hits[pi] += 1
    cand = torch.cat([torch.ones(G, dtype=torch.bool), gt_inds > 0])
    rate = hits[cand] / 200                     # expected 32 / 205 = 0.156 each, sd 0.026
    assert hits[~cand].sum() == 0 and rate.min() > 0.05 and rate.max() < 0.30
