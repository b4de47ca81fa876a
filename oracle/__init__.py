"""CPU oracle for the DynaMask mask-head hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``dynamask_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / timed CPU baseline.

Parity status (see DESIGN.md "Oracle"):
  * reference-owned arithmetic (losses, DetailTarget, generate_block_target,
    MaskPre, gumbel selector, head control flow, boundary merge): PINNED by
    golden vectors generated from the reference's own modules
    (tests/golden/make_golden.py -> tests/golden/*.npz).
  * mmcv==1.0.5 operators (RoIAlign, SimpleRoIAlign/point_sample,
    DeformConv2dPack, CARAFEPack): mmcv is a third-party dependency absent from
    /root/reference and from this image -> "parity unpinned" for those ops.
    They are restated from mmcv's published algorithm and pinned only by
    known-answer tests (constant/ramp maps, zero-offset DCN == conv2d,
    grid_sample identity, float64 brute force).
  * likewise third-party and unpinned: mmcv.ops.nms / batched_nms (restated greedy
    NMS) and pycocotools' RLE (cocoapi maskApi.c restated; round trips + hand vectors).
  * bbox branch (Shared2FCBBoxHead, get_bboxes, delta2bbox): PINNED by g10
    (tests/golden/make_golden_bbox.py).
"""
