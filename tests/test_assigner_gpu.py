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
