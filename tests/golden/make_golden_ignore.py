"""Golden vectors for the ``gt_bboxes_ignore`` branch of the REFERENCE's MaxIoUAssigner
(mmdet/core/bbox/assigners/max_iou_assigner.py:107-118): boxes whose IoF with an ignore region exceeds
``ignore_iof_thr`` get overlap -1 (assignment -1), with the IoF taken over the candidate (``ignore_wrt_candidates``)
or over the ignore region.  The reference's own assigner and ``BboxOverlaps2D`` are loaded by path as in
make_golden_train.py.  Run ONLY in the authoring container:  python tests/golden/make_golden_ignore.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_train as mgt  # noqa: E402

SEED = 977


def inputs():
    g = torch.Generator().manual_seed(SEED)
    n = 96
    xy = torch.rand(n, 2, generator=g) * torch.tensor([300.0, 200.0])
    wh = torch.rand(n, 2, generator=g) * 90 + 4
    bboxes = torch.cat([xy, xy + wh], 1)
    gts = torch.tensor([[20., 30., 120., 110.], [150., 40., 260., 150.], [60., 100., 200., 190.], [250., 10., 300., 60.]])
    ign = torch.tensor([[0., 0., 80., 80.], [200., 120., 330., 230.], [140., 60., 170., 90.]])
    labels = torch.tensor([3, 7, 1, 5])
    return bboxes, gts, ign, labels


def main():
    R = mgt.load_train_reference()
    bboxes, gts, ign, labels = inputs()
    out = dict(bboxes=bboxes.numpy(), gts=gts.numpy(), ign=ign.numpy(), labels=labels.numpy())
    for name, wrt in (('cand', True), ('region', False)):
        asg = R['assigner'].MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, ignore_iof_thr=0.5,
                                           ignore_wrt_candidates=wrt)
        ar = asg.assign(bboxes.clone(), gts.clone(), ign.clone(), labels.clone())
        out[f'{name}_gt_inds'] = ar.gt_inds.numpy()
        out[f'{name}_max_overlaps'] = ar.max_overlaps.numpy()
        out[f'{name}_labels'] = ar.labels.numpy()
    # the same assigner without ignore regions, so that a test can see that the branch changes something
    asg0 = R['assigner'].MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, ignore_iof_thr=0.5)
    out['noign_gt_inds'] = asg0.assign(bboxes.clone(), gts.clone(), None, labels.clone()).gt_inds.numpy()
    path = os.path.join(HERE, 'g13_assign_ignore.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: (v.shape, int((v == -1).sum()) if v.dtype.kind == 'i' else None) for k, v in out.items() if 'gt_inds' in k})


if __name__ == '__main__':
    main()
