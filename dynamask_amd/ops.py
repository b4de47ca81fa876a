"""Operator bindings: torch device tensors -> raw pointers -> C ABI.

PyTorch is plumbing here (device memory, streams); all arithmetic runs in
libdynamask_hip.so.  Every function requires contiguous fp32 tensors on a HIP
device and raises otherwise -- no eager fallback.
"""
import ctypes

import torch

from . import hazard
from ._lib import check, lib


import os

# bumped whenever a kernel rewrites parameter memory behind torch's back (the fused
# SGD step), so that packed-weight caches refresh
WEIGHT_EPOCH = [0]

# DM_DETERMINISTIC=1 (or setting this flag): every cross-workgroup accumulation of the training step goes through
# the *_fx entry points (64-bit fixed-point cells, order-independent) and is converted once: two runs of a training
# loop give bit-identical parameters.  Costs one zero-fill and one conversion per accumulated tensor.
DETERMINISTIC = [os.environ.get('DM_DETERMINISTIC', '0') == '1']


def _fx_like(t):
    return torch.zeros(t.shape, device=t.device, dtype=torch.int64)


def _fx_finish(fx, out, accumulate):
    check(lib().dm_fx_to_float(_p(fx), fx.numel(), _p(out), 1 if accumulate else 0, 0, _stream()), 'dm_fx_to_float')
    return out


# the current stream's handle straight from the binding (0.4 us; torch.cuda.current_stream() builds a Stream object and
# resolves the device twice: 8 us x 160 calls of a training step), with the public path as the fallback
_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_CUR_DEVICE = getattr(torch._C, '_cuda_getDevice', None)


def _stream():
    if _RAW_STREAM is not None and _CUR_DEVICE is not None:
        return ctypes.c_void_p(_RAW_STREAM(_CUR_DEVICE()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'{name}: expected a tensor')
    if not t.is_cuda:
        raise RuntimeError(f'{name}: dynamask_amd operators run on the MI355X only (got a {t.device} tensor); '
                           'there is no CPU fallback')
    if t.dtype != dtype:
        raise TypeError(f'{name}: expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name}: tensor must be contiguous (NCHW)')
    return t


def _chk_src(t):
    """A conv source may be a channel slice of a wider NCHW tensor: channels,
    rows and columns dense, arbitrary batch stride."""
    if isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and \
            not t.is_contiguous():
        _, C, H, W = t.shape
        if t.stride(3) == 1 and t.stride(2) == W and t.stride(1) == H * W and t.stride(0) >= C * H * W:
            return t
    return _chk(t, 'src')


def _p(t):
    if t is None:
        return ctypes.c_void_p(0)
    if hazard.ENABLED[0]:
        hazard.note_ptr(t)          # (DM_HAZARD: the tracker learns which tensor the pointer came from)
    return ctypes.c_void_p(t.data_ptr())


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    if hazard.ENABLED[0]:
        hazard.note_ptr_array(arr, tensors)
    return arr


def _int_array(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def _float_array(vals):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


# ------------------------------------------------------------------ RoIAlign
# 14x14 / 7x7 extractions of 192 RoIs or more go through dm_roi_align_fwd_ws with a scratch buffer (ROI_WORKSPACE below):
# the library orders the RoIs by level and position on the device first (DM_ROI_SORT, default on: 57 -> 50.7 us for 512 RoIs,
# the same bits)
ROI_WORKSPACE = True          # (tests compare with the unordered kernel by clearing it)
ROI_WORKSPACE_MIN = int(os.environ.get('DM_ROI_SORT_MIN', '192'))


def roi_align(feats, rois, output_size, spatial_scales, sampling_ratio=0, finest_scale=56.0, return_levels=False):
    feats = [_chk(f, 'feat') for f in feats]
    _chk(rois, 'rois')
    B, C = feats[0].shape[:2]
    N = rois.shape[0]
    out = torch.empty((N, C, output_size, output_size), device=rois.device, dtype=torch.float32)
    levels = torch.zeros((N,), device=rois.device, dtype=torch.int32) if return_levels else None
    if ROI_WORKSPACE and N >= ROI_WORKSPACE_MIN:
        # 14x14 / 7x7 extraction with a workspace: the RoIs are first ordered by level and position (one extra launch:
        # neighbouring workgroups then share their footprints in the L2; same results).
        wsb = int(lib().dm_roi_align_workspace_bytes(N, output_size))
        ws = torch.empty(((wsb + 15) // 16 * 4,), device=rois.device, dtype=torch.int32) if wsb > 0 else None
        rc = lib().dm_roi_align_fwd_ws(_ptr_array(feats), _int_array([f.shape[2] for f in feats]),
                                       _int_array([f.shape[3] for f in feats]), _float_array(spatial_scales), len(feats),
                                       B, C, _p(rois), N, output_size, sampling_ratio, finest_scale, _p(out), _p(levels),
                                       _p(ws), wsb, _stream())
        check(rc, 'dm_roi_align_fwd_ws')
        return (out, levels) if return_levels else out
    rc = lib().dm_roi_align_fwd(_ptr_array(feats), _int_array([f.shape[2] for f in feats]),
                                _int_array([f.shape[3] for f in feats]), _float_array(spatial_scales), len(feats),
                                B, C, _p(rois), N, output_size, sampling_ratio, finest_scale, _p(out), _p(levels),
                                _stream())
    check(rc, 'dm_roi_align_fwd')
    return (out, levels) if return_levels else out


def roi_align_backward(grad_out, feat_shapes, rois, output_size, spatial_scales, sampling_ratio=0, finest_scale=56.0):
    _chk(grad_out, 'grad_out')
    _chk(rois, 'rois')
    grads = [torch.zeros(s, device=grad_out.device, dtype=torch.float32) for s in feat_shapes]
    B, C = feat_shapes[0][:2]
    N = rois.shape[0]
    rc = lib().dm_roi_align_bwd(_p(grad_out), _ptr_array(grads), _int_array([s[2] for s in feat_shapes]),
                                _int_array([s[3] for s in feat_shapes]), _float_array(spatial_scales),
                                len(feat_shapes), B, C, _p(rois), N, output_size, sampling_ratio, finest_scale,
                                _stream())
    check(rc, 'dm_roi_align_bwd')
    return grads


# --------------------------------------------------------------- RLE (after the path, 8f rank 2)
def _rle_collect(N, img_h, img_w, runs, start, positions, launch, capacity):
    """Shared tail of the two encoders: read the run totals, re-run once with a larger
    buffer if the boundaries did not fit, copy exactly the used part of `positions`, build
    the COCO dicts ({'size': [h, w], 'counts': bytes}, what pycocotools' encode returns)."""
    import ctypes as C
    start_h = start.cpu()                        # synchronises the stream
    total = int(start_h[N])
    if total > capacity:
        positions = torch.empty((total,), device=runs.device, dtype=torch.int32)
        launch(positions, total)
        start_h = start.cpu()
    pos_h = torch.empty((max(total, 1),), dtype=torch.int32, pin_memory=True)
    if total > 0:
        pos_h[:total].copy_(positions[:total], non_blocking=True)
        torch.cuda.current_stream().synchronize()
    base = pos_h.data_ptr()
    L = lib()
    out = []
    cap = 64
    buf = C.create_string_buffer(cap)
    for n in range(N):
        s0, s1 = int(start_h[n]), int(start_h[n + 1])
        need = (s1 - s0 + 1) * 7             # <= 7 characters per count (31-bit values)
        if need > cap:
            cap = need
            buf = C.create_string_buffer(cap)
        ln = L.dm_rle_string(C.c_void_p(base + 4 * s0), s1 - s0, img_h * img_w, buf, cap)
        if ln < 0:
            raise RuntimeError('dm_rle_string: buffer too small')
        out.append({'size': [int(img_h), int(img_w)], 'counts': buf.raw[:ln]})
    return out


def rle_encode(canvas):
    """uint8/bool [N, h, w] device bitmaps -> list of COCO RLE dicts (device encoder)."""
    if canvas.dtype == torch.bool:
        canvas = canvas.view(torch.uint8)
    _chk(canvas, 'canvas', torch.uint8)
    N, h, w = canvas.shape
    if N == 0:
        return []
    dev = canvas.device
    scratch = torch.empty((lib().dm_rle_scratch_ints(N, h, w),), device=dev, dtype=torch.int32)
    runs = torch.empty((N,), device=dev, dtype=torch.int32)
    start = torch.empty((N + 1,), device=dev, dtype=torch.int32)
    capacity = max(4096, N * 4 * (h + w))

    def launch(positions, cap):
        check(lib().dm_rle_encode_canvas(_p(canvas), N, h, w, _p(scratch), _p(runs), _p(start), _p(positions), cap,
                                         _stream()), 'dm_rle_encode_canvas')
    positions = torch.empty((capacity,), device=dev, dtype=torch.int32)
    launch(positions, capacity)
    return _rle_collect(N, h, w, runs, start, positions, launch, capacity)


def paste_rle(masks, boxes, img_h, img_w, threshold=0.5, apply_sigmoid=False):
    """Paste + threshold + RLE in one go: the [N, img_h, img_w] canvas is never written and only
    the run boundaries cross PCIe.  Same pixel arithmetic as ``paste_masks``."""
    _chk(masks, 'masks')
    _chk(boxes, 'boxes')
    N = masks.shape[0]
    if N == 0:
        return []
    mh, mw = masks.shape[-2:]
    dev = masks.device
    img_h, img_w = int(img_h), int(img_w)
    scratch = torch.empty((lib().dm_rle_scratch_ints(N, img_h, img_w),), device=dev, dtype=torch.int32)
    runs = torch.empty((N,), device=dev, dtype=torch.int32)
    start = torch.empty((N + 1,), device=dev, dtype=torch.int32)
    capacity = max(4096, N * 4 * (img_h + img_w))

    def launch(positions, cap):
        check(lib().dm_paste_rle(_p(masks), _p(boxes), N, mh, mw, img_h, img_w, float(threshold),
                                 1 if apply_sigmoid else 0, _p(scratch), _p(runs), _p(start), _p(positions), cap,
                                 _stream()), 'dm_paste_rle')
    positions = torch.empty((capacity,), device=dev, dtype=torch.int32)
    launch(positions, capacity)
    return _rle_collect(N, img_h, img_w, runs, start, positions, launch, capacity)


# --------------------------------------------------------------- bbox branch (8f rank 4)
def fc(x, weight, bias=None, relu=False):
    """nn.Linear forward: x [N, K], weight [M, K], bias [M] -> [N, M] (fp32 MFMA, deterministic split-K)."""
    _chk(x, 'x')
    _chk(weight, 'weight')
    if bias is not None:
        _chk(bias, 'bias')
    N, K = x.shape
    M = weight.shape[0]
    assert weight.shape[1] == K
    out = torch.empty((N, M), device=x.device, dtype=torch.float32)
    ns = int(lib().dm_fc_scratch_floats(N, K, M))
    scratch = torch.empty((ns,), device=x.device, dtype=torch.float32) if ns > 0 else None
    check(lib().dm_fc_fwd(_p(x), _p(weight), _p(bias), N, K, M, 1 if relu else 0, _p(out), _p(scratch), _stream()),
          'dm_fc_fwd')
    return out


def bbox_decode(rois, cls_score, bbox_pred, num_classes, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.),
                wh_ratio_clip=16 / 1000, max_shape=None, scale=(1.0, 1.0), class_agnostic=False):
    """softmax(cls_score), delta2bbox(rois, bbox_pred) clipped to ``max_shape`` (h, w) and
    divided by ``scale`` (sx, sy).  rois [N, 5] (batch column first) or [N, 4]."""
    _chk(rois, 'rois')
    N = rois.shape[0]
    dev = rois.device
    nb = 1 if class_agnostic else num_classes
    scores = None
    if cls_score is not None:
        _chk(cls_score, 'cls_score')
        assert cls_score.shape == (N, num_classes + 1)
        scores = torch.empty_like(cls_score)
    if bbox_pred is not None:
        _chk(bbox_pred, 'bbox_pred')
        assert bbox_pred.shape == (N, 4 * nb)
    bboxes = torch.empty((N, 4 * nb), device=dev, dtype=torch.float32)
    ch, cw = (float(max_shape[0]), float(max_shape[1])) if max_shape is not None else (0.0, 0.0)
    check(lib().dm_bbox_decode(_p(rois), rois.shape[1], rois.shape[1] - 4, _p(cls_score), _p(bbox_pred), N, num_classes,
                               1 if class_agnostic else 0, _float_array(means), _float_array(stds), float(wh_ratio_clip),
                               ch, cw, float(scale[0]), float(scale[1]), _p(scores), _p(bboxes), _stream()),
          'dm_bbox_decode')
    return bboxes, scores


def nms(boxes, scores, iou_threshold, offset=0, max_num=-1):
    """mmcv.ops.nms: (dets [k, 5] in descending score order, keep indices [k] into the input).
    Suppression matrix on the device, greedy pass on the host."""
    import ctypes as C
    _chk(boxes, 'boxes')
    _chk(scores, 'scores')
    M = boxes.shape[0]
    if M == 0:
        return boxes.new_zeros((0, 5)), torch.zeros((0,), dtype=torch.long, device=boxes.device)
    order = torch.sort(scores, descending=True, stable=True)[1]
    sb = boxes[order].contiguous()
    words = (M + 63) // 64
    mask = torch.empty((M, words), device=boxes.device, dtype=torch.int64)
    check(lib().dm_nms_mask(_p(sb), M, float(iou_threshold), int(offset), _p(mask), _stream()), 'dm_nms_mask')
    mask_h = mask.cpu()                                   # synchronises
    keep_h = torch.empty((M,), dtype=torch.int32)
    n = lib().dm_nms_reduce(C.c_void_p(mask_h.data_ptr()), M, C.c_void_p(keep_h.data_ptr()), int(max_num))
    keep_sorted = keep_h[:n].to(device=boxes.device, dtype=torch.long)
    keep = order[keep_sorted]
    dets = torch.cat([boxes[keep], scores[keep][:, None]], 1)
    return dets, keep


# --------------------------------------------------------------- convolutions
def packed_cout(cout):
    return lib().dm_conv_packed_cout(int(cout))


def packed_floats(cout, ksize, src_channels):
    n = lib().dm_conv_packed_floats(int(cout), int(ksize), len(src_channels), _int_array(src_channels))
    if n < 0:
        raise ValueError('bad conv packing request')
    return int(n)


def pack_conv_weight(w, transpose_flip=False, src_channels=None):
    """OIHW -> [k*k][KQ][CoutP][4] (see include/dynamask_hip.h).  ``src_channels``:
    how the input channels split over the concat sources (default: one source)."""
    _chk(w, 'weight')
    cout, cin, kh, kw = w.shape
    assert kh == kw and kh in (1, 3)
    rows = cout if transpose_flip else cin
    cols = cin if transpose_flip else cout
    if src_channels is None:
        src_channels = [rows]
    assert sum(src_channels) == rows
    wp = torch.empty((packed_floats(cols, kh, src_channels),), device=w.device, dtype=torch.float32)
    check(lib().dm_conv_pack_weight(_p(w), cout, cin, kh, 1 if transpose_flip else 0, len(src_channels),
                                    _int_array(src_channels), _p(wp), _stream()), 'dm_conv_pack_weight')
    return wp


class _DevicePlan:
    """The packs of ONE device: entries, the device-resident job table, and the event of the last refresh."""

    def __init__(self, device):
        self.device = device
        self.entries = []          # dicts: param (weakref), out, job fields, ver, used
        self.table = None
        self.table_ids = None
        self.old_tables = []       # previous job tables stay alive for two more refreshes: a batch launch on another
                                   # stream may still be reading one when the next is uploaded
        self.event = None          # recorded behind the last refresh; streams that read packs wait for it (get)
        self.generation = 0
        self.waited = {}           # stream id -> generation it has waited for


class PackPlan:
    """Kernel-layout weights of a training step, refreshed by ONE launch (dm_conv_pack_weight_batch) per device.

    Every optimizer step changes ~45 weight tensors, each of which the kernels read in a packed layout; packing them
    one launch at a time cost the host 1.3 ms of the 3.7 ms it needs to issue a forward pass, and the forward is the
    part of the step where the GPU waits for the host.  A pack registered here (``get``) owns a persistent output
    buffer; the first request that finds its pack stale refreshes, in one launch, every registered pack of that device
    that was used since the previous refresh (a pack nobody asked for in a whole step is left alone and refreshed on
    demand).  The job table lives on the device and is rebuilt only when the set of jobs changes.

    Streams: a refresh runs on the stream that asked for it and records an event; ``get`` makes any OTHER stream wait
    for that event before it hands out a buffer (once per stream and refresh), so a pack refreshed on the main stream
    and read by a side stream -- or the other way round -- is ordered whoever triggers the refresh."""

    def __init__(self):
        self.plans = {}            # device index -> _DevicePlan
        self.launches = 0
        self.uploads = 0

    @property
    def entries(self):
        return [e for p in self.plans.values() for e in p.entries]

    @staticmethod
    def _ver(param):
        return (param.data_ptr(), param._version, WEIGHT_EPOCH[0])

    def _plan(self, device):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        p = self.plans.get(idx)
        if p is None:
            p = self.plans[idx] = _DevicePlan(torch.device('cuda', idx))
        return p

    def register(self, param, transpose_flip, src_channels, lo, hi):
        import weakref
        cout, cin_total, kh, kw = param.shape
        lo = 0 if lo is None else lo
        hi = cin_total if hi is None else hi
        cin = hi - lo
        rows = cout if transpose_flip else cin
        cols = cin if transpose_flip else cout
        src_channels = [rows] if src_channels is None else list(src_channels)
        assert sum(src_channels) == rows and kh == kw and kh in (1, 3) and len(src_channels) <= 4
        out = torch.empty((packed_floats(cols, kh, src_channels),), device=param.device, dtype=torch.float32)
        plan = self._plan(param.device)
        e = dict(param=weakref.ref(param), out=out, cout=cout, cin=cin, ks=kh, flip=1 if transpose_flip else 0,
                 srcs=src_channels, ld=cin_total, c0=lo, ver=None, used=True, plan=plan)
        plan.entries.append(e)
        return e

    def get(self, e):
        p = e['param']()
        if p is None:
            raise RuntimeError('PackPlan.get: the parameter of this pack no longer exists')
        plan = e['plan']
        e['used'] = True
        if e['ver'] != self._ver(p):
            self._refresh(plan)
            e['used'] = True          # (refresh clears the flag of what it packed; this one is in use now)
        if plan.event is not None:
            st = torch.cuda.current_stream(plan.device)
            if plan.waited.get(st.cuda_stream) != plan.generation:
                st.wait_event(plan.event)          # (a no-op for the stream that recorded it)
                plan.waited[st.cuda_stream] = plan.generation
        return e['out']

    def refresh(self, device=None):
        """Refresh the stale packs of ``device`` (default: the current device) on its current stream."""
        idx = torch.cuda.current_device() if device is None else torch.device(device).index
        plan = self.plans.get(idx)
        if plan is not None:
            self._refresh(plan)

    def _refresh(self, plan):
        from ._lib import PackJob
        live = []
        for e in plan.entries:
            p = e['param']()
            if p is not None:
                live.append((e, p))
        plan.entries = [e for e, _ in live]
        todo = [(e, p) for e, p in live if e['used'] and e['ver'] != self._ver(p)]
        if not todo:
            return
        with torch.cuda.device(plan.device):
            ids = tuple((id(e), p.data_ptr()) for e, p in todo)
            if ids != plan.table_ids and torch.cuda.is_current_stream_capturing():
                # (with the job table already on the device the refresh is one launch and is captured like any other:
                # a captured training step repacks its weights on every replay)
                raise RuntimeError('PackPlan: a kernel-layout weight is stale inside a HIP-graph capture and its job table '
                                   'is not on the device yet (the upload cannot be captured: run the path once eagerly first)')
            if ids != plan.table_ids:
                arr = (PackJob * len(todo))()
                for j, (e, p) in zip(arr, todo):
                    j.w, j.w_packed = p.data_ptr(), e['out'].data_ptr()
                    j.Cout, j.Cin, j.ksize, j.transpose_flip = e['cout'], e['cin'], e['ks'], e['flip']
                    j.num_srcs = len(e['srcs'])
                    for k, c in enumerate(e['srcs']):
                        j.src_channels[k] = c
                    j.ld, j.c0 = e['ld'], e['c0']
                host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
                if plan.table is not None:
                    plan.old_tables = (plan.old_tables + [plan.table])[-2:]
                plan.table = host.to(plan.device)
                plan.table_ids = ids
                self.uploads += 1
            if hazard.ENABLED[0]:       # the device-side job table hides what this launch reads and writes
                hazard.touch('dm_conv_pack_weight_batch', reads=[p for _, p in todo], writes=[e['out'] for e, _ in todo])
            check(lib().dm_conv_pack_weight_batch(_p(plan.table), len(todo), _stream()), 'dm_conv_pack_weight_batch')
            self.launches += 1
            plan.event = torch.cuda.current_stream(plan.device).record_event()
            plan.generation += 1
            plan.waited = {torch.cuda.current_stream(plan.device).cuda_stream: plan.generation}
        for e, p in todo:
            e['ver'] = self._ver(p)
            e['used'] = False
        # (a pack requested later in the same step sets ``used`` again and finds its version current)


PACK_PLAN = PackPlan()


_overlapped = 0


class overlapped_streams:
    """Context: launches inside run next to other work on a second stream (DynaMaskRoIHead splits
    the RoIs over two streams).  Passed to dm_conv2d_fwd as flag bit 3, a scheduling hint."""

    def __enter__(self):
        global _overlapped
        _overlapped += 1

    def __exit__(self, *exc):
        global _overlapped
        _overlapped -= 1


# DM_CONV_SPLITK=0: no split-K for small inference launches (the sums of one launch, in one order, whatever the RoI count)
CONV_SPLITK = [os.environ.get('DM_CONV_SPLITK', '1') == '1']
_SPLITK_DEPTH = [0]


class splitk_scope:
    """Inside this scope ``conv2d`` may split the K loop of a launch that would leave most of the chip idle
    (dm_conv2d_fwd_ws): entered by the inference entry points of the RoI head only -- a training forward keeps one
    association of its sums whatever the RoI count (its golden tests pin gradients behind ReLU kinks and pool ties)."""

    def __enter__(self):
        _SPLITK_DEPTH[0] += 1
        return self

    def __exit__(self, *exc):
        _SPLITK_DEPTH[0] -= 1
        return False


def conv2d(srcs, w_packed, bias, cout, ksize, relu=False, out=None, out_ch_offset=0, accumulate=False, mask=None):
    """Fused concat(srcs) -> conv(ksize, same) -> +bias -> ReLU.  ``mask`` (same shape as ``out``): outputs where it
    is not > 0 are stored as 0 -- the ReLU adjoint of a data gradient, fused into the epilogue."""
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    srcs = [_chk_src(s) for s in srcs]
    _chk(w_packed, 'w_packed')
    if bias is not None:
        _chk(bias, 'bias')
    NB, _, H, W = srcs[0].shape
    for s in srcs:
        assert s.shape[0] == NB and s.shape[2] == H and s.shape[3] == W
    cin = sum(s.shape[1] for s in srcs)
    assert w_packed.numel() == packed_floats(cout, ksize, [s.shape[1] for s in srcs]), 'weights packed for other sources'
    if out is None:
        out = torch.empty((NB, cout, H, W), device=srcs[0].device, dtype=torch.float32)
    else:
        _chk(out, 'out')
        assert out.shape[0] == NB and out.shape[2] == H and out.shape[3] == W
    strides = (ctypes.c_longlong * len(srcs))(*[int(s.stride(0)) for s in srcs])
    flags = (1 if relu else 0) | (2 if accumulate else 0) | (8 if _overlapped else 0)
    if mask is not None:
        _chk(mask, 'mask')
        assert mask.shape == out.shape
        rc = lib().dm_conv2d_fwd_masked(_ptr_array(srcs), _int_array([s.shape[1] for s in srcs]), strides, len(srcs), NB, H, W,
                                        _p(w_packed), _p(bias), cout, ksize, flags, _p(out), out.shape[1], out_ch_offset,
                                        _p(mask), _stream())
        check(rc, 'dm_conv2d_fwd_masked')
        return out
    if CONV_SPLITK[0] and _SPLITK_DEPTH[0] > 0 and not accumulate:
        # the <= 100-RoI inference calls: a launch of few workgroups splits its K loop (dm_conv2d_fwd_ws)
        nws = int(lib().dm_conv2d_splitk_floats(NB, H, W, cout, ksize))
        if nws > 0:
            ws = torch.empty((nws,), device=out.device, dtype=torch.float32)
            rc = lib().dm_conv2d_fwd_ws(_ptr_array(srcs), _int_array([s.shape[1] for s in srcs]), strides, len(srcs), NB, H, W,
                                        _p(w_packed), _p(bias), cout, ksize, flags, _p(out), out.shape[1], out_ch_offset,
                                        _p(ws), nws, _stream())
            check(rc, 'dm_conv2d_fwd_ws')
            return out
    rc = lib().dm_conv2d_fwd(_ptr_array(srcs), _int_array([s.shape[1] for s in srcs]), strides, len(srcs), NB, H, W,
                             _p(w_packed), _p(bias), cout, ksize, flags, _p(out), out.shape[1], out_ch_offset, _stream())
    check(rc, 'dm_conv2d_fwd')
    return out


def conv1x1_group(xs, w_packeds, biases, couts, relu=False, outs=None):
    """Up to three independent single-source 1x1 convolutions (+ bias, + ReLU) as ONE launch (dm_conv1x1_group_fwd): the
    FPN-wide semantic convolutions of the SFM stages.  Same bits as ``conv2d`` per problem."""
    k = len(xs)
    assert 1 <= k <= 3 and len(w_packeds) == k and len(biases) == k and len(couts) == k
    NB = xs[0].shape[0]
    for x, w in zip(xs, w_packeds):
        _chk(x, 'x')
        _chk(w, 'w_packed')
        assert x.shape[0] == NB
    for x, w, c in zip(xs, w_packeds, couts):
        assert w.numel() == packed_floats(c, 1, [x.shape[1]]), 'weights packed for other sources'
    if outs is None:
        outs = [torch.empty((NB, c, x.shape[2], x.shape[3]), device=x.device, dtype=torch.float32) for x, c in zip(xs, couts)]
    for o, x, c in zip(outs, xs, couts):
        _chk(o, 'out')
        assert tuple(o.shape) == (NB, c, x.shape[2], x.shape[3])
    bias_arr = (ctypes.c_void_p * k)(*[0 if b is None else _chk(b, 'bias').data_ptr() for b in biases])
    if hazard.ENABLED[0]:
        hazard.note_ptr_array(bias_arr, [b for b in biases if b is not None])
    check(lib().dm_conv1x1_group_fwd(k, _ptr_array(xs), _int_array([x.shape[1] for x in xs]), _int_array([x.shape[2] for x in xs]),
                                     _int_array([x.shape[3] for x in xs]), NB, _ptr_array(w_packeds), bias_arr, _int_array(couts),
                                     1 if relu else 0, _ptr_array(outs), _stream()), 'dm_conv1x1_group_fwd')
    return outs


def deform_conv(x, offset, w_packed, cout, deform_groups, relu=False, out=None):
    _chk(x, 'x')
    _chk(offset, 'offset')
    _chk(w_packed, 'w_packed')
    NB, C, H, W = x.shape
    assert offset.shape == (NB, deform_groups * 18, H, W)
    assert w_packed.numel() == packed_floats(cout, 3, [C])
    if out is None:
        out = torch.empty((NB, cout, H, W), device=x.device, dtype=torch.float32)
    else:
        _chk(out, 'out')
        assert tuple(out.shape) == (NB, cout, H, W)
    if CONV_SPLITK[0] and _SPLITK_DEPTH[0] > 0:
        nws = int(lib().dm_deform_conv_splitk_floats(NB, C, H, W, cout))
        if nws > 0:
            ws = torch.empty((nws,), device=out.device, dtype=torch.float32)
            check(lib().dm_deform_conv_fwd_ws(_p(x), _p(offset), NB, C, H, W, _p(w_packed), cout, deform_groups,
                                              (1 if relu else 0) | (8 if _overlapped else 0), _p(out), _p(ws), nws, _stream()),
                  'dm_deform_conv_fwd_ws')
            return out
    check(lib().dm_deform_conv_fwd(_p(x), _p(offset), NB, C, H, W, _p(w_packed), cout, deform_groups,
                                   (1 if relu else 0) | (8 if _overlapped else 0), _p(out), _stream()), 'dm_deform_conv_fwd')
    return out


def deform_conv_tout_supported(x, cout, m2):
    """Can ``deform_conv_tout`` take this shape (else: ``deform_conv`` + ``conv2d``)?"""
    NB, C, H, W = x.shape
    return bool(lib().dm_deform_conv_tout_supported(NB, C, H, W, cout, m2))


def pack_tout_weight(w):
    """The 1x1 weight [M2, C, 1, 1] behind a DCN, as ``deform_conv_tout`` reads it: transposed [C][M2 padded to 32], zeros
    in the padding."""
    m2, c = w.shape[0], w.shape[1]
    m2p = (m2 + 31) // 32 * 32
    out = torch.zeros((c, m2p), device=w.device, dtype=torch.float32)
    out[:, :m2] = w.detach().reshape(m2, c).t()
    return out


def deform_conv_tout(x, offset, w_packed, cout, deform_groups, w2t, b2, m2, out2, keep_dcn=False, dcn_out=None):
    """relu(DCN 3x3) -> 1x1 conv + bias + ReLU into channels [0, m2) of ``out2`` in ONE launch (dm_deform_conv_tout_fwd:
    the second GEMM runs on the DCN's accumulators in registers; same bits as ``deform_conv(relu=True)`` followed by
    ``conv2d(relu=True, out=out2)``).  ``keep_dcn``: also return relu(DCN) (else it is never written)."""
    _chk(x, 'x')
    _chk(offset, 'offset')
    _chk(w_packed, 'w_packed')
    _chk(w2t, 'w2t')
    _chk(b2, 'b2')
    _chk(out2, 'out2')
    NB, C, H, W = x.shape
    assert offset.shape == (NB, deform_groups * 18, H, W)
    assert w_packed.numel() == packed_floats(cout, 3, [C])
    assert tuple(w2t.shape) == (cout, (m2 + 31) // 32 * 32) and b2.numel() == m2
    assert out2.shape[0] == NB and tuple(out2.shape[2:]) == (H, W) and out2.shape[1] >= m2
    dcn = None
    if dcn_out is not None:                 # (a caller's buffer for relu(DCN): implies keep_dcn)
        dcn = _chk(dcn_out, 'dcn_out')
        assert tuple(dcn.shape) == (NB, cout, H, W)
    elif keep_dcn:
        dcn = torch.empty((NB, cout, H, W), device=x.device, dtype=torch.float32)
    check(lib().dm_deform_conv_tout_fwd(_p(x), _p(offset), NB, C, H, W, _p(w_packed), cout, deform_groups, _p(w2t), _p(b2), m2,
                                        _p(out2), out2.shape[1], _p(dcn), _stream()), 'dm_deform_conv_tout_fwd')
    return dcn


def pack_deconv_weight(w):
    _chk(w, 'weight')
    cin, cout, kh, kw = w.shape
    assert kh == 2 and kw == 2
    wp = torch.empty((packed_floats(4 * cout, 1, [cin]),), device=w.device, dtype=torch.float32)
    check(lib().dm_deconv_pack_weight(_p(w), cin, cout, _p(wp), _stream()), 'dm_deconv_pack_weight')
    return wp


def deconv2x2(x, w_packed, bias, cout, relu=False):
    _chk(x, 'x')
    _chk(w_packed, 'w_packed')
    NB, C, H, W = x.shape
    out = torch.empty((NB, cout, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
    check(lib().dm_deconv2x2_fwd(_p(x), NB, C, H, W, _p(w_packed), _p(bias), cout, 1 if relu else 0, _p(out),
                                 _stream()), 'dm_deconv2x2_fwd')
    return out


def carafe(x, enc, up_kernel=5, group=1, scale=2):
    _chk(x, 'x')
    _chk(enc, 'enc')
    NB, C, H, W = x.shape
    assert enc.shape == (NB, up_kernel * up_kernel * group * scale * scale, H, W)
    out = torch.empty((NB, C, H * scale, W * scale), device=x.device, dtype=torch.float32)
    check(lib().dm_carafe_fwd(_p(x), _p(enc), NB, C, H, W, up_kernel, group, scale, _p(out), _stream()),
          'dm_carafe_fwd')
    return out


# ------------------------------------------------------------ bandwidth kernels
def point_sample(feat, rois, output_size, spatial_scale):
    _chk(feat, 'feat')
    _chk(rois, 'rois')
    B, C, H, W = feat.shape
    N = rois.shape[0]
    out = torch.empty((N, C, output_size, output_size), device=feat.device, dtype=torch.float32)
    check(lib().dm_point_sample_fwd(_p(feat), B, C, H, W, _p(rois), N, output_size, spatial_scale, _p(out),
                                    _stream()), 'dm_point_sample_fwd')
    return out


def class_logits(x, w_inst, b_inst, w_det, b_det, labels, sig_out=None, sig_ch_offset=0, out=None):
    _chk(x, 'x')
    for t, n in ((w_inst, 'w_inst'), (b_inst, 'b_inst'), (w_det, 'w_det'), (b_det, 'b_det')):
        _chk(t, n)
    _chk(labels, 'labels', torch.int64)
    N, C, H, W = x.shape
    nc = w_inst.shape[0]
    if out is None:
        inst = torch.empty((N, 1, H, W), device=x.device, dtype=torch.float32)
        det = torch.empty((N, 1, H, W), device=x.device, dtype=torch.float32)
    else:                                   # (inst, det): e.g. row slices of the buffers of a chunked launch sequence
        inst, det = out
        for t, nm in ((inst, 'out[0]'), (det, 'out[1]')):
            _chk(t, nm)
            assert tuple(t.shape) == (N, 1, H, W)
    sig_ct = 0
    if sig_out is not None:
        _chk(sig_out, 'sig_out')
        assert sig_out.shape[0] == N and sig_out.shape[2:] == x.shape[2:]
        sig_ct = sig_out.shape[1]
    check(lib().dm_class_logits_fwd(_p(x), N, C, H * W, _p(w_inst), _p(b_inst), _p(w_det), _p(b_det), nc,
                                    _p(labels), _p(inst), _p(det), _p(sig_out), sig_ct, sig_ch_offset, _stream()),
          'dm_class_logits_fwd')
    return inst, det


def class_logits_up2x_supported(x):
    return x.shape[3] % 2 == 0 and x.shape[2] >= 2 and x.shape[3] >= 2


def class_logits_up2x(x, w_inst, b_inst, w_det, b_det, labels, out=None):
    """``class_logits(upsample2x(x, align_corners=False, relu=True), ...)`` in one kernel, without the upsampled tensor
    (the stage before an exit).  x [N, C, H, W] -> (inst, det) [N, 1, 2H, 2W]."""
    _chk(x, 'x')
    for t, n in ((w_inst, 'w_inst'), (b_inst, 'b_inst'), (w_det, 'w_det'), (b_det, 'b_det')):
        _chk(t, n)
    _chk(labels, 'labels', torch.int64)
    N, C, H, W = x.shape
    nc = w_inst.shape[0]
    if out is None:
        inst = torch.empty((N, 1, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
        det = torch.empty((N, 1, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
    else:
        inst, det = out
        for t, nm in ((inst, 'out[0]'), (det, 'out[1]')):
            _chk(t, nm)
            assert tuple(t.shape) == (N, 1, 2 * H, 2 * W)
    check(lib().dm_class_logits_up2x_fwd(_p(x), N, C, H, W, _p(w_inst), _p(b_inst), _p(w_det), _p(b_det), nc, _p(labels),
                                         _p(inst), _p(det), _stream()), 'dm_class_logits_up2x_fwd')
    return inst, det


def upsample2x(x, align_corners=False, relu=False, out=None):
    _chk(x, 'x')
    N, C, H, W = x.shape
    if out is None:
        out = torch.empty((N, C, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
    else:
        _chk(out, 'out')
        assert tuple(out.shape) == (N, C, 2 * H, 2 * W)
    check(lib().dm_upsample2x_bilinear_fwd(_p(x), N * C, H, W, 1 if align_corners else 0, 1 if relu else 0,
                                           _p(out), _stream()), 'dm_upsample2x_bilinear_fwd')
    return out


def boundary_merge_(coarse, fine):
    """In place on ``fine`` (as the reference, dynamask_roi_head.py:148)."""
    _chk(coarse, 'coarse')
    _chk(fine, 'fine')
    n, S = coarse.shape[0], coarse.shape[-1]
    assert fine.shape[0] == n and fine.shape[-1] == 2 * S and fine.shape[-2] == 2 * S
    check(lib().dm_boundary_merge(_p(coarse), _p(fine), n, S, _stream()), 'dm_boundary_merge')
    return fine


def boundary_merge_chain(p_s, p_2s, fine_or_final, out=None):
    """The inference tail in one launch (dm_boundary_merge_chain): ``merge(merge(p_s -> p_2s) -> fine)``.
    ``fine_or_final``: the [n, 1, 4S, 4S] fine logits (merged in place and returned), or the last stage's [n, 1, 2S, 2S]
    logits -- then the fine logits are their align_corners x2 upsample, computed inside the kernel, and the result goes
    to ``out`` (allocated if None).  ``p_2s`` is not modified."""
    _chk(p_s, 'p_s')
    _chk(p_2s, 'p_2s')
    _chk(fine_or_final, 'fine_or_final')
    n, S = p_s.shape[0], p_s.shape[-1]
    assert p_s.shape[-2] == S and tuple(p_2s.shape[-2:]) == (2 * S, 2 * S) and p_2s.shape[0] == n == fine_or_final.shape[0]
    if n == 0:
        return fine_or_final if fine_or_final.shape[-1] == 4 * S else p_s.new_zeros((0, 1, 4 * S, 4 * S))
    if fine_or_final.shape[-1] == 4 * S:
        assert out is None or out is fine_or_final
        check(lib().dm_boundary_merge_chain(_p(p_s), _p(p_2s), None, _p(fine_or_final), n, S, _stream()), 'dm_boundary_merge_chain')
        return fine_or_final
    assert tuple(fine_or_final.shape[-2:]) == (2 * S, 2 * S)
    if out is None:
        out = torch.empty((n, 1, 4 * S, 4 * S), device=p_s.device, dtype=torch.float32)
    else:
        _chk(out, 'out')
        assert out.shape[0] == n and tuple(out.shape[-2:]) == (4 * S, 4 * S)
    check(lib().dm_boundary_merge_chain(_p(p_s), _p(p_2s), _p(fine_or_final), _p(out), n, S, _stream()), 'dm_boundary_merge_chain')
    return out


def stage_head(sem, rois, output_size, spatial_scale, x, w_inst, b_inst, w_det, b_det, labels, sig_out=None, sig_ch_offset=0, out=None):
    """``point_sample(sem, rois, output_size, spatial_scale)`` and ``class_logits(x, ..., sig_out, sig_ch_offset, out)`` of
    one SFM stage as ONE launch (dm_stage_head_fwd) -> (sampled, inst, det)."""
    _chk(sem, 'sem')
    _chk(rois, 'rois')
    _chk(x, 'x')
    for t, n_ in ((w_inst, 'w_inst'), (b_inst, 'b_inst'), (w_det, 'w_det'), (b_det, 'b_det')):
        _chk(t, n_)
    _chk(labels, 'labels', torch.int64)
    B, Cs, H, W = sem.shape
    N, C, S, S2 = x.shape
    assert S == S2 == output_size and rois.shape[0] == N
    nc = w_inst.shape[0]
    sampled = torch.empty((N, Cs, S, S), device=x.device, dtype=torch.float32)
    if out is None:
        inst = torch.empty((N, 1, S, S), device=x.device, dtype=torch.float32)
        det = torch.empty((N, 1, S, S), device=x.device, dtype=torch.float32)
    else:
        inst, det = out
        for t, nm in ((inst, 'out[0]'), (det, 'out[1]')):
            _chk(t, nm)
            assert tuple(t.shape) == (N, 1, S, S)
    sig_ct = 0
    if sig_out is not None:
        _chk(sig_out, 'sig_out')
        assert sig_out.shape[0] == N and sig_out.shape[2:] == x.shape[2:]
        sig_ct = sig_out.shape[1]
    check(lib().dm_stage_head_fwd(_p(sem), B, Cs, H, W, _p(rois), N, S, spatial_scale, _p(sampled), _p(x), C, _p(w_inst), _p(b_inst),
                                  _p(w_det), _p(b_det), nc, _p(labels), _p(inst), _p(det), _p(sig_out), sig_ct, sig_ch_offset,
                                  _stream()), 'dm_stage_head_fwd')
    return sampled, inst, det


def gumbel_select(logits, U, temperature=0.5):
    _chk(logits, 'logits')
    _chk(U, 'U')
    N, K = logits.shape
    y = torch.empty_like(logits)
    hot = torch.empty_like(logits)
    idx = torch.empty((N,), device=logits.device, dtype=torch.int32)
    check(lib().dm_gumbel_select_fwd(_p(logits), _p(U), N, K, temperature, _p(y), _p(hot), _p(idx), _stream()),
          'dm_gumbel_select_fwd')
    return y, hot, idx


def detail_target(masks, fuse=(0.7, 0.3)):
    """fuse: two host floats, or a device tensor holding them (read by the kernel: no sync)."""
    _chk(masks, 'masks')
    N, S = masks.shape[0], masks.shape[-1]
    out = torch.empty((N, S, S), device=masks.device, dtype=torch.float32)
    if isinstance(fuse, torch.Tensor):
        fd = _chk(fuse.detach().reshape(-1), 'fuse')
        assert fd.numel() == 2
        rc = lib().dm_detail_target(_p(masks), N, S, 0.0, 0.0, _p(fd), _p(out), _stream())
    else:
        rc = lib().dm_detail_target(_p(masks), N, S, float(fuse[0]), float(fuse[1]), None, _p(out), _stream())
    check(rc, 'dm_detail_target')
    return out


def mask_loss(inst_pred, det_pred, inst_tgt, det_tgt, weight, need_grad=True):
    """Returns (sums[2], per_roi_det[N], grad_inst, grad_det) -- see the header."""
    for t, n in ((inst_pred, 'inst_pred'), (det_pred, 'det_pred'), (inst_tgt, 'inst_tgt'), (det_tgt, 'det_tgt'),
                 (weight, 'weight')):
        _chk(t, n)
    N = inst_pred.shape[0]
    HW = inst_pred.numel() // max(N, 1)
    sums = torch.zeros((2,), device=inst_pred.device, dtype=torch.float32)
    per_roi = torch.zeros((N,), device=inst_pred.device, dtype=torch.float32)
    gi = torch.empty_like(inst_pred) if need_grad else None
    gd = torch.empty_like(det_pred) if need_grad else None
    scratch = torch.empty((max(int(lib().dm_mask_loss_scratch_floats(N)), 1),), device=inst_pred.device, dtype=torch.float32)
    check(lib().dm_mask_loss_fwd_bwd(_p(inst_pred), _p(det_pred), _p(inst_tgt), _p(det_tgt), _p(weight), N, HW,
                                     _p(sums), _p(per_roi), _p(gi), _p(gd), _p(scratch), _stream()), 'dm_mask_loss_fwd_bwd')
    return sums, per_roi, gi, gd


def mask_loss_stage(inst_pred, det_pred, inst_tgt, det_tgt, mask_labels, stage, detail_weight, loss_terms, grad_ml,
                    want_inst_grad):
    """dm_mask_loss_stage: one stage of DynaCrossEntropyLoss, normalisers applied in the kernel.  ``loss_terms`` [2]
    and ``grad_ml`` [N, K] are accumulated into / written in place; returns (grad_inst | None, grad_det)."""
    for t, n in ((inst_pred, 'inst_pred'), (det_pred, 'det_pred'), (inst_tgt, 'inst_tgt'), (det_tgt, 'det_tgt'),
                 (mask_labels, 'mask_labels'), (loss_terms, 'loss_terms'), (grad_ml, 'grad_ml')):
        _chk(t, n)
    N, K = mask_labels.shape
    HW = inst_pred.numel() // max(N, 1)
    gi = torch.empty_like(inst_pred) if want_inst_grad else None
    gd = torch.empty_like(det_pred)
    scratch = torch.empty((max(int(lib().dm_mask_loss_scratch_floats(N)), 1),), device=inst_pred.device, dtype=torch.float32)
    check(lib().dm_mask_loss_stage(_p(inst_pred), _p(det_pred), _p(inst_tgt), _p(det_tgt), _p(mask_labels), K, int(stage), N,
                                   HW, float(detail_weight), _p(loss_terms), _p(grad_ml), _p(gi), _p(gd), _p(scratch),
                                   _stream()), 'dm_mask_loss_stage')
    return gi, gd


def gumbel_select_backward(y_soft, grad_y, temperature=0.5):
    _chk(y_soft, 'y_soft')
    _chk(grad_y, 'grad_y')
    N, K = y_soft.shape
    g = torch.empty_like(y_soft)
    check(lib().dm_gumbel_select_bwd(_p(y_soft), _p(grad_y), N, K, temperature, _p(g), _stream()),
          'dm_gumbel_select_bwd')
    return g


def class_balance(mask_labels):
    """(cb loss scalar tensor, d cb / d mask_labels)."""
    _chk(mask_labels, 'mask_labels')
    N, K = mask_labels.shape
    loss = torch.empty((1,), device=mask_labels.device, dtype=torch.float32)
    grad = torch.empty_like(mask_labels)
    check(lib().dm_class_balance_fwd_bwd(_p(mask_labels), N, K, _p(loss), _p(grad), _stream()),
          'dm_class_balance_fwd_bwd')
    return loss[0], grad


_BN_SCRATCH = {}


def _bn_scratch(device, C):
    """Partial-sum scratch of the split BatchNorm kernels, one buffer per (device, stream): no
    allocation on the step path (kernels that share it are ordered by their stream)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    need = lib().dm_bn_scratch_floats(C)
    buf = _BN_SCRATCH.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.empty((max(need, 1 << 15),), device=device, dtype=torch.float32)
        _BN_SCRATCH[key] = buf
    return buf


def bn_stats(x, running_mean=None, running_var=None, momentum=0.1, mean_shift=None):
    """``mean_shift`` [C]: the running mean moves towards mean + mean_shift (x lacks a per-channel bias, see the header)."""
    _chk(x, 'x')
    if mean_shift is not None:
        _chk(mean_shift, 'mean_shift')
    NB, C, H, W = x.shape
    mean = torch.empty((C,), device=x.device, dtype=torch.float32)
    var = torch.empty((C,), device=x.device, dtype=torch.float32)
    scratch = _bn_scratch(x.device, C)
    check(lib().dm_bn_stats(_p(x), NB, C, H * W, _p(mean), _p(var), _p(running_mean), _p(running_var), momentum,
                            _p(mean_shift), _p(scratch), _stream()), 'dm_bn_stats')
    return mean, var


def bn_relu_maxpool(x, mean, var, gamma, beta, eps=1e-5):
    for t, n in ((x, 'x'), (mean, 'mean'), (var, 'var'), (gamma, 'gamma'), (beta, 'beta')):
        _chk(t, n)
    NB, C, H, W = x.shape
    out = torch.empty((NB, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), device=x.device, dtype=torch.float32)
    check(lib().dm_bn_relu_maxpool_fwd(_p(x), NB, C, H, W, _p(mean), _p(var), _p(gamma), _p(beta), eps, _p(out),
                                       _stream()), 'dm_bn_relu_maxpool_fwd')
    return out


def bn_relu_maxpool_argmax(x, mean, var, gamma, beta, eps=1e-5):
    """Plane index of the tap each pooled output took (diagnostic; = max_pool2d(return_indices=True))."""
    for t, n in ((x, 'x'), (mean, 'mean'), (var, 'var'), (gamma, 'gamma'), (beta, 'beta')):
        _chk(t, n)
    NB, C, H, W = x.shape
    arg = torch.empty((NB, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), device=x.device, dtype=torch.int32)
    check(lib().dm_bn_relu_maxpool_argmax(_p(x), NB, C, H, W, _p(mean), _p(var), _p(gamma), _p(beta), eps, _p(arg),
                                          _stream()), 'dm_bn_relu_maxpool_argmax')
    return arg


# ------------------------------------------------------------------- backward ops
def relu_backward_(grad, out):
    _chk(grad, 'grad')
    _chk(out, 'out')
    assert grad.numel() == out.numel()
    check(lib().dm_relu_bwd(_p(grad), _p(out), grad.numel(), _stream()), 'dm_relu_bwd')
    return grad


def sigmoid_backward(sig, ga, gb, g_logit=None):
    """sig / ga / gb: [N, 1, H, W] views (possibly channel slices); returns / accumulates
    d/dlogit = (ga + gb) * sig * (1 - sig)."""
    N, _, H, W = sig.shape
    for t in (sig, ga) + ((gb,) if gb is not None else ()):
        assert t.is_cuda and t.dtype == torch.float32 and t.shape[1] == 1 and t.stride(3) == 1 and t.stride(2) == W
    acc = g_logit is not None
    if g_logit is None:
        g_logit = torch.empty((N, 1, H, W), device=sig.device, dtype=torch.float32)
    check(lib().dm_sigmoid_bwd(_p(sig), sig.stride(0), _p(ga), ga.stride(0), _p(gb), gb.stride(0) if gb is not None else 0,
                               N, H * W, _p(g_logit), 1 if acc else 0, _stream()), 'dm_sigmoid_bwd')
    return g_logit


def channel_sum(g, out=None):
    g = _chk_src(g)
    NB, C, H, W = g.shape
    acc = out is not None
    if out is None:
        out = torch.empty((C,), device=g.device, dtype=torch.float32)
    if DETERMINISTIC[0]:
        fx = _fx_like(out)
        check(lib().dm_channel_sum_fx(_p(g), g.stride(0), NB, C, H * W, _p(fx), _stream()), 'dm_channel_sum_fx')
        return _fx_finish(fx, out, acc)
    check(lib().dm_channel_sum(_p(g), g.stride(0), NB, C, H * W, _p(out), 1 if acc else 0, _stream()), 'dm_channel_sum')
    return out


CLB_SLAB = [True]      # class-logit parameter gradients without contended atomics (False: per-RoI float atomics)
WGRAD_SLAB = [True]    # weight gradients by slab reduce (no atomics); False: float atomics
_WGRAD_SCRATCH = [0]


def _wgrad_scratch_floats():
    if not _WGRAD_SCRATCH[0]:
        _WGRAD_SCRATCH[0] = int(lib().dm_conv2d_wgrad_scratch_floats())
    return _WGRAD_SCRATCH[0]


def conv2d_wgrad(dy, srcs, ksize, dw=None, db=None, want_bias=False):
    """dW [Cout, sum(Cs), k, k] of conv(cat(srcs)) given dy [NB, Cout, H, W].  ``db`` ([Cout], accumulated into) or
    ``want_bias`` (a fresh tensor): the bias gradient from the same pass over dy; returns dW, or (dW, db) with a bias."""
    dy = _chk_src(dy)
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    srcs = [_chk_src(s) for s in srcs]
    NB, Cout, H, W = dy.shape
    cin = sum(s.shape[1] for s in srcs)
    det = DETERMINISTIC[0]
    if dw is None:
        dw = (torch.empty if det else torch.zeros)((Cout, cin, ksize, ksize), device=dy.device, dtype=torch.float32)
        acc = False
    else:
        _chk(dw, 'dw')
        assert dw.shape == (Cout, cin, ksize, ksize)
        acc = True
    bias_acc = db is not None
    if db is None and want_bias:
        db = (torch.empty if det else torch.zeros)((Cout,), device=dy.device, dtype=torch.float32)
    if db is not None:
        _chk(db, 'db')
        assert db.shape == (Cout,)
    if WGRAD_SLAB[0]:
        # split-K partial tiles into slabs, added in a fixed order: no atomics, the same bits every run (in either mode)
        if det:
            if not acc:
                dw.zero_()
            if db is not None and not bias_acc:
                db.zero_()
        nfl = _wgrad_scratch_floats()
        scratch = torch.empty((nfl,), device=dy.device, dtype=torch.float32)      # from the calling stream's pool
        base = 0
        for i, s in enumerate(srcs):
            check(lib().dm_conv2d_wgrad_slab(_p(dy), dy.stride(0), Cout, _p(s), s.stride(0), s.shape[1], NB, H, W, ksize, _p(dw),
                                             cin * ksize * ksize, base * ksize * ksize, _p(db if i == 0 else None), _p(scratch),
                                             nfl, _stream()), 'dm_conv2d_wgrad_slab')
            base += s.shape[1]
        return (dw, db) if db is not None else dw
    fx = _fx_like(dw) if det else None
    bfx = _fx_like(db) if det and db is not None else None
    fn = lib().dm_conv2d_wgrad_fx if det else lib().dm_conv2d_wgrad
    base = 0
    for i, s in enumerate(srcs):
        bias_ptr = (bfx if det else db) if i == 0 else None
        check(fn(_p(dy), dy.stride(0), Cout, _p(s), s.stride(0), s.shape[1], NB, H, W, ksize, _p(fx if det else dw),
                 cin * ksize * ksize, base * ksize * ksize, _p(bias_ptr), _stream()), 'dm_conv2d_wgrad')
        base += s.shape[1]
    if det:
        _fx_finish(fx, dw, acc)
        if db is not None:
            _fx_finish(bfx, db, bias_acc)
    return (dw, db) if db is not None else dw


def upsample2x_backward(grad_out, fwd_out, in_shape, align_corners=False):
    _chk(grad_out, 'grad_out')
    if fwd_out is not None:
        _chk(fwd_out, 'fwd_out')
    N, C, H, W = in_shape
    gin = torch.empty(in_shape, device=grad_out.device, dtype=torch.float32)
    check(lib().dm_upsample2x_bilinear_bwd(_p(grad_out), _p(fwd_out), N * C, H, W, 1 if align_corners else 0, _p(gin),
                                           _stream()), 'dm_upsample2x_bilinear_bwd')
    return gin


def point_sample_backward(grad_out, feat_shape, rois, spatial_scale, grad_feat=None):
    _chk(grad_out, 'grad_out')
    _chk(rois, 'rois')
    B, C, H, W = feat_shape
    N, _, S, _ = grad_out.shape
    if DETERMINISTIC[0]:
        acc = grad_feat is not None
        if grad_feat is None:
            grad_feat = torch.empty(feat_shape, device=grad_out.device, dtype=torch.float32)
        fx = _fx_like(grad_feat)
        check(lib().dm_point_sample_bwd_fx(_p(grad_out), B, C, H, W, _p(rois), N, S, spatial_scale, _p(fx), _stream()),
              'dm_point_sample_bwd_fx')
        return _fx_finish(fx, grad_feat, acc)
    if grad_feat is None:
        grad_feat = torch.zeros(feat_shape, device=grad_out.device, dtype=torch.float32)
    check(lib().dm_point_sample_bwd(_p(grad_out), B, C, H, W, _p(rois), N, S, spatial_scale, _p(grad_feat), _stream()),
          'dm_point_sample_bwd')
    return grad_feat


def class_logits_backward(x, w_inst, w_det, labels, g_inst, g_det, grad_x, accumulate_x, gw_inst, gb_inst, gw_det, gb_det):
    for t, n in ((x, 'x'), (w_inst, 'w_inst'), (w_det, 'w_det'), (g_inst, 'g_inst'), (g_det, 'g_det'), (grad_x, 'grad_x'),
                 (gw_inst, 'gw_inst'), (gb_inst, 'gb_inst'), (gw_det, 'gw_det'), (gb_det, 'gb_det')):
        _chk(t, n)
    _chk(labels, 'labels', torch.int64)
    N, C, H, W = x.shape
    nc = w_inst.shape[0]
    if CLB_SLAB[0] and N > 0:
        # per-RoI sums to a scratch, added per class in RoI order by a second launch: no contended atomics (the RoIs of an
        # image share a few classes) and a fixed order of additions -- also what the deterministic mode uses
        scratch = torch.empty((int(lib().dm_class_logits_bwd_scratch_floats(N, C)),), device=x.device, dtype=torch.float32)
        check(lib().dm_class_logits_bwd_slab(_p(x), N, C, H * W, _p(w_inst), _p(w_det), nc, _p(labels), _p(g_inst), _p(g_det),
                                             _p(grad_x), 1 if accumulate_x else 0, _p(gw_inst), _p(gb_inst), _p(gw_det), _p(gb_det),
                                             _p(scratch), scratch.numel(), _stream()), 'dm_class_logits_bwd_slab')
        return grad_x
    if DETERMINISTIC[0]:
        outs = (gw_inst, gb_inst, gw_det, gb_det)
        fxs = [_fx_like(t) for t in outs]
        check(lib().dm_class_logits_bwd_fx(_p(x), N, C, H * W, _p(w_inst), _p(w_det), nc, _p(labels), _p(g_inst), _p(g_det),
                                           _p(grad_x), 1 if accumulate_x else 0, *[_p(f) for f in fxs], _stream()),
              'dm_class_logits_bwd_fx')
        for f, t in zip(fxs, outs):
            _fx_finish(f, t, True)
        return grad_x
    check(lib().dm_class_logits_bwd(_p(x), N, C, H * W, _p(w_inst), _p(w_det), nc, _p(labels), _p(g_inst), _p(g_det),
                                    _p(grad_x), 1 if accumulate_x else 0, _p(gw_inst), _p(gb_inst), _p(gw_det), _p(gb_det),
                                    _stream()), 'dm_class_logits_bwd')
    return grad_x


def deform_im2col(x, offset, deform_groups, out=None):
    _chk(x, 'x')
    _chk(offset, 'offset')
    NB, C, H, W = x.shape
    if out is None:
        col = torch.empty((NB, 9 * C, H, W), device=x.device, dtype=torch.float32)
    else:
        col = out
        _chk(col, 'out')
        assert tuple(col.shape) == (NB, 9 * C, H, W)
    check(lib().dm_deform_im2col(_p(x), _p(offset), NB, C, H, W, deform_groups, _p(col), _stream()), 'dm_deform_im2col')
    return col


def deform_col2im_coord(colgrad, x, offset, deform_groups):
    _chk(colgrad, 'colgrad')
    _chk(x, 'x')
    _chk(offset, 'offset')
    NB, C, H, W = x.shape
    gx = torch.empty_like(x)
    goff = torch.empty_like(offset)
    check(lib().dm_deform_col2im_coord(_p(colgrad), _p(x), _p(offset), NB, C, H, W, deform_groups, _p(gx), _p(goff),
                                       _stream()), 'dm_deform_col2im_coord')
    return gx, goff


def deform_coord_grad(colgrad, x, offset, deform_groups, out=None):
    """Offset gradient of DCNv1 from the column gradient (first half of deform_col2im_coord)."""
    _chk(colgrad, 'colgrad')
    _chk(x, 'x')
    _chk(offset, 'offset')
    NB, C, H, W = x.shape
    goff = torch.empty_like(offset) if out is None else out
    check(lib().dm_deform_coord_grad(_p(colgrad), _p(x), _p(offset), NB, C, H, W, deform_groups, _p(goff), _stream()),
          'dm_deform_coord_grad')
    return goff


def deform_col2im(colgrad, offset, x_shape, deform_groups, out=None):
    """Input gradient of DCNv1 from the column gradient (second half of deform_col2im_coord)."""
    _chk(colgrad, 'colgrad')
    _chk(offset, 'offset')
    NB, C, H, W = x_shape
    gx = torch.empty(x_shape, device=colgrad.device, dtype=torch.float32) if out is None else out
    check(lib().dm_deform_col2im(_p(colgrad), _p(offset), NB, C, H, W, deform_groups, _p(gx), _stream()), 'dm_deform_col2im')
    return gx


def dcn_weight_permute(src, cout, c, to_colmajor, dst=None, accumulate=False):
    _chk(src, 'src')
    if dst is None:
        shape = (9 * c, cout, 1, 1) if to_colmajor else (cout, c, 3, 3)
        dst = torch.empty(shape, device=src.device, dtype=torch.float32)
    check(lib().dm_dcn_weight_permute(_p(src), _p(dst), cout, c, 1 if to_colmajor else 0, 1 if accumulate else 0,
                                      _stream()), 'dm_dcn_weight_permute')
    return dst


def pack_dcn_colgrad_weight(weight):
    """DCN weight [Cout, C, 3, 3] packed for the column-gradient GEMM W^T . dY (a 1x1 conv Cout -> 9C)."""
    cout, c = weight.shape[0], weight.shape[1]
    return pack_conv_weight(dcn_weight_permute(weight.contiguous(), cout, c, True))        # [(tap,ci)][co]


def deform_conv_backward_data(x, offset, weight, grad_out, deform_groups, side=None, w_colgrad=None):
    """(grad_x, grad_offset) of DCNv1 3x3: column gradient = W^T . dY as a 1x1 conv, then the coordinate gradient and
    col2im over it (``w_colgrad``: ``pack_dcn_colgrad_weight(weight)``); ``side``: a second stream -- the coordinate
    gradient (bound by its gathers) then runs there, beside col2im (bound by LDS atomics) on the caller's stream; both
    only read the column gradient.  (Round 4's one-kernel form measured at parity -- both are bound by the LDS scatter --
    and was removed in round 5: docs/HISTORY.md.)"""
    NB, C, H, W = x.shape
    if w_colgrad is None:
        w_colgrad = pack_dcn_colgrad_weight(weight)
    colgrad = conv2d(grad_out, w_colgrad, None, 9 * C, 1)                      # W^T . dY
    if side is None:
        return deform_col2im_coord(colgrad, x, offset, deform_groups)
    main = torch.cuda.current_stream(x.device)
    goff = torch.empty_like(offset)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        deform_coord_grad(colgrad, x, offset, deform_groups, out=goff)
    gx = deform_col2im(colgrad, offset, tuple(x.shape), deform_groups)
    main.wait_stream(side)             # colgrad (main-stream memory) is released after this point
    return gx, goff


def deform_conv_backward_weight(x, offset, grad_out, deform_groups, gw_accum=None, col=None):
    """grad_weight [Cout, C, 3, 3] of DCNv1 3x3 = dY . col^T as a 1x1 weight gradient over the tap-major column
    matrix.  ``gw_accum``: tensor the gradient is ADDED to (then None is returned); ``col``: the column matrix of
    (x, offset) if the forward kept it."""
    NB, C, H, W = x.shape
    cout = grad_out.shape[1]
    if col is None:
        col = deform_im2col(x, offset, deform_groups)
    gw_cm = conv2d_wgrad(grad_out, col, 1)                                     # [co][(tap,ci)]
    del col
    if gw_accum is not None:
        dcn_weight_permute(gw_cm, cout, C, False, dst=gw_accum, accumulate=True)
        return None
    return dcn_weight_permute(gw_cm, cout, C, False)


def deform_conv_backward(x, offset, weight, grad_out, deform_groups, gw_accum=None, col=None):
    """(grad_x, grad_offset, grad_weight) of DCNv1 3x3.  The two GEMMs of the
    reference's backward run as 1x1 convs over the tap-major column matrix."""
    gx, goff = deform_conv_backward_data(x, offset, weight, grad_out, deform_groups)
    gw = deform_conv_backward_weight(x, offset, grad_out, deform_groups, gw_accum=gw_accum, col=col)
    return gx, goff, gw


def bn_relu_maxpool_backward(x, mean, var, gamma, beta, grad_out, eps=1e-5):
    for t, n in ((x, 'x'), (mean, 'mean'), (var, 'var'), (gamma, 'gamma'), (beta, 'beta'), (grad_out, 'grad_out')):
        _chk(t, n)
    NB, C, H, W = x.shape
    gx = torch.empty_like(x)
    gg = torch.empty((C,), device=x.device, dtype=torch.float32)
    gb = torch.empty((C,), device=x.device, dtype=torch.float32)
    scratch = _bn_scratch(x.device, C)
    check(lib().dm_bn_relu_maxpool_bwd(_p(x), NB, C, H, W, _p(mean), _p(var), _p(gamma), _p(beta), eps, _p(grad_out),
                                       _p(gx), _p(gg), _p(gb), _p(scratch), _stream()), 'dm_bn_relu_maxpool_bwd')
    return gx, gg, gb


def sgd_momentum_step_(params_flat, grads_flat, momentum_flat, lr, momentum=0.9, weight_decay=1e-4, grad_scale=1.0,
                       first_step=False):
    for t, n in ((params_flat, 'params'), (grads_flat, 'grads'), (momentum_flat, 'momentum')):
        _chk(t, n)
    check(lib().dm_sgd_momentum_step(_p(params_flat), _p(grads_flat), _p(momentum_flat), params_flat.numel(), lr, momentum,
                                     weight_decay, grad_scale, 1 if first_step else 0, _stream()), 'dm_sgd_momentum_step')
    WEIGHT_EPOCH[0] += 1


# ------------------------------------------------------- callers either side of the path
def mask_target_rois(boxes, gt_inds, max_w, max_h):
    _chk(boxes, 'boxes')
    _chk(gt_inds, 'gt_inds', torch.int64)
    N = boxes.shape[0]
    rois = torch.empty((N, 5), device=boxes.device, dtype=torch.float32)
    check(lib().dm_mask_target_rois(_p(boxes), _p(gt_inds), N, float(max_w), float(max_h), _p(rois), _stream()),
          'dm_mask_target_rois')
    return rois


def pack_polygons(masks, device):
    """``PolygonMasks.masks`` (list over objects of lists of float arrays x0, y0, x1, y1, ...) -> the three device
    arrays of ``polygon_mask_targets``: vertices [V, 2] float64, first vertex of each polygon [P + 1] int32, first
    polygon of each object [G + 1] int32.  One upload per image."""
    import numpy as np
    verts, poly_start, inst_start = [], [0], [0]
    for obj in masks:
        for p in obj:
            p = np.asarray(p, dtype=np.float64).reshape(-1)
            assert p.size % 2 == 0, 'a polygon is a flat (x, y) list'
            verts.append(p.reshape(-1, 2))
            poly_start.append(poly_start[-1] + p.size // 2)
        inst_start.append(len(poly_start) - 1)
    v = np.concatenate(verts, 0) if verts else np.zeros((0, 2), np.float64)
    return (torch.from_numpy(np.ascontiguousarray(v)).to(device), torch.tensor(poly_start, dtype=torch.int32, device=device),
            torch.tensor(inst_start, dtype=torch.int32, device=device))


def polygon_mask_targets(packed, boxes, gt_inds, size):
    """[N, size, size] 0 / 1 targets of the objects ``gt_inds`` in the frames of ``boxes`` (clipped to the image)."""
    verts, poly_start, inst_start = packed
    _chk(verts, 'verts', torch.float64)
    _chk(poly_start, 'poly_start', torch.int32)
    _chk(inst_start, 'inst_start', torch.int32)
    _chk(boxes, 'boxes')
    _chk(gt_inds, 'gt_inds', torch.int64)
    N = boxes.shape[0]
    out = torch.empty((N, size, size), device=boxes.device, dtype=torch.float32)
    check(lib().dm_polygon_mask_targets(_p(verts), _p(poly_start), _p(inst_start), inst_start.numel() - 1, _p(boxes), _p(gt_inds),
                                        N, size, _p(out), _stream()), 'dm_polygon_mask_targets')
    return out


def threshold_ge(x, thr):
    _chk(x, 'x')
    out = torch.empty_like(x)
    check(lib().dm_threshold_ge(_p(x), x.numel(), float(thr), _p(out), _stream()), 'dm_threshold_ge')
    return out


def paste_masks(masks, boxes, img_h, img_w, threshold=0.5, apply_sigmoid=False, out=None):
    """masks [N, 1, h, w] or [N, h, w], boxes [N, 4] -> bool [N, img_h, img_w].
    ``out``: optional contiguous uint8 [N, img_h, img_w] slab to paste into (a slice of a
    larger canvas when detections are pasted bucket by bucket)."""
    _chk(masks, 'masks')
    _chk(boxes, 'boxes')
    N = masks.shape[0]
    mh, mw = masks.shape[-2:]
    if out is None:
        out = torch.empty((N, img_h, img_w), device=masks.device, dtype=torch.uint8)
    else:
        _chk(out, 'out', torch.uint8)
        assert tuple(out.shape) == (N, img_h, img_w)
    check(lib().dm_paste_masks(_p(masks), _p(boxes), N, mh, mw, int(img_h), int(img_w), float(threshold),
                               1 if apply_sigmoid else 0, _p(out), _stream()), 'dm_paste_masks')
    return out.view(torch.bool) if N > 0 else out.bool()


# ------------------------------------------------------- RoI sampling + bbox-branch training
def bbox_overlaps(bboxes1, bboxes2, mode='iou', eps=1e-6):
    """core/bbox/iou_calculators/iou2d_calculator.py:37-131 (is_aligned=False) -> [n1, n2]."""
    assert mode in ('iou', 'iof')
    n1, n2 = bboxes1.shape[0], bboxes2.shape[0]
    out = torch.empty((n1, n2), device=bboxes1.device, dtype=torch.float32)
    if n1 * n2 == 0:
        return out
    _chk(bboxes1, 'bboxes1')
    _chk(bboxes2, 'bboxes2')
    check(lib().dm_bbox_overlaps(_p(bboxes1), n1, _p(bboxes2), n2, 1 if mode == 'iof' else 0, float(eps), _p(out),
                                 _stream()), 'dm_bbox_overlaps')
    return out


def ignore_columns_(overlaps, iof, thr, boxes_major=True):
    """max_iou_assigner.py:107-118, in place: overlaps[:, n] = -1 where box n's largest IoF with an ignore region is
    > thr.  ``iof``: [N, I] (boxes vs regions, ``boxes_major``) or [I, N] (regions vs boxes)."""
    _chk(overlaps, 'overlaps')
    _chk(iof, 'iof')
    G, N = overlaps.shape
    I = iof.shape[1] if boxes_major else iof.shape[0]
    assert (iof.shape[0] if boxes_major else iof.shape[1]) == N
    check(lib().dm_ignore_columns(_p(overlaps), G, N, _p(iof), I, 1 if boxes_major else 0, float(thr), _stream()),
          'dm_ignore_columns')
    return overlaps


def max_iou_assign(overlaps, pos_iou_thr, neg_iou_thr, min_pos_iou=0.0, match_low_quality=True, gt_max_assign_all=True,
                   gt_labels=None):
    """max_iou_assigner.py:129-212 for non-empty inputs -> (gt_inds, max_overlaps, labels|None)."""
    _chk(overlaps, 'overlaps')
    k, n = overlaps.shape
    dev = overlaps.device
    lo, hi = (0.0, float(neg_iou_thr)) if isinstance(neg_iou_thr, float) else (float(neg_iou_thr[0]), float(neg_iou_thr[1]))
    gt_inds = torch.empty((n,), device=dev, dtype=torch.int64)
    max_ov = torch.empty((n,), device=dev, dtype=torch.float32)
    labels = None
    if gt_labels is not None:
        _chk(gt_labels, 'gt_labels', torch.int64)
        labels = torch.empty((n,), device=dev, dtype=torch.int64)
    scratch = torch.empty((2 * k,), device=dev, dtype=torch.float32)
    check(lib().dm_max_iou_assign(_p(overlaps), k, n, float(pos_iou_thr), lo, hi, float(min_pos_iou),
                                  1 if match_low_quality else 0, 1 if gt_max_assign_all else 0, _p(gt_labels), _p(scratch),
                                  _p(gt_inds), _p(max_ov), _p(labels), _stream()), 'dm_max_iou_assign')
    return gt_inds, max_ov, labels


def random_sample(gt_inds, bboxes, n_prepended, gt_bboxes, labels, pos_keys, neg_keys, keys_by_class_rank, num, quota_pos,
                  neg_pos_ub=-1.0):
    """dm_random_sample: key-ranked RoI sampling -> dict of [num]-row device buffers + ``counts`` [4] int32
    (positives kept, negatives kept, positive candidates, negative candidates).  No host sync."""
    _chk(gt_inds, 'gt_inds', torch.int64)
    _chk(bboxes, 'bboxes')
    _chk(pos_keys, 'pos_keys')
    _chk(neg_keys, 'neg_keys')
    M = gt_inds.shape[0]
    assert bboxes.shape == (M, 4)
    if not keys_by_class_rank:
        assert pos_keys.numel() >= M and neg_keys.numel() >= M
    G = int(gt_bboxes.shape[0])
    if G:
        _chk(gt_bboxes, 'gt_bboxes')
    if labels is not None:
        _chk(labels, 'labels', torch.int64)
        assert labels.shape[0] == M
    dev = gt_inds.device
    i64 = dict(device=dev, dtype=torch.int64)
    f32 = dict(device=dev, dtype=torch.float32)
    out = dict(pos_inds=torch.empty((num,), **i64), neg_inds=torch.empty((num,), **i64),
               counts=torch.empty((4,), device=dev, dtype=torch.int32),
               pos_bboxes=torch.empty((num, 4), **f32), neg_bboxes=torch.empty((num, 4), **f32),
               pos_gt_bboxes=torch.empty((num, 4), **f32), pos_assigned_gt_inds=torch.empty((num,), **i64),
               pos_gt_labels=torch.empty((num,), **i64) if labels is not None else None,
               pos_is_gt=torch.empty((num,), device=dev, dtype=torch.uint8))
    scratch = torch.empty((3 * M,), device=dev, dtype=torch.int32)
    check(lib().dm_random_sample(_p(gt_inds), _p(bboxes), M, int(n_prepended), _p(gt_bboxes) if G else _p(None), G,
                                 _p(labels), _p(pos_keys), _p(neg_keys), 1 if keys_by_class_rank else 0, int(num),
                                 int(quota_pos), float(neg_pos_ub), _p(scratch), _p(out['pos_inds']), _p(out['neg_inds']),
                                 _p(out['counts']), _p(out['pos_bboxes']), _p(out['neg_bboxes']), _p(out['pos_gt_bboxes']),
                                 _p(out['pos_assigned_gt_inds']), _p(out['pos_gt_labels']), _p(out['pos_is_gt']),
                                 _stream()), 'dm_random_sample')
    return out


def bbox_encode(proposals, gt, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.)):
    """bbox2delta, delta_xywh_bbox_coder.py:74-116."""
    _chk(proposals, 'proposals')
    _chk(gt, 'gt')
    assert proposals.shape == gt.shape
    out = torch.empty_like(proposals)
    check(lib().dm_bbox_encode(_p(proposals), _p(gt), proposals.shape[0], _float_array(means), _float_array(stds), _p(out),
                               _stream()), 'dm_bbox_encode')
    return out


def softmax_ce(cls_score, labels, weight, scale, need_grad=True):
    """-> (loss [1] = scale * sum_i w_i CE_i, acc [1] percent, grad [N, C] or None)."""
    _chk(cls_score, 'cls_score')
    _chk(labels, 'labels', torch.int64)
    if weight is not None:
        _chk(weight, 'weight')
    N, C = cls_score.shape
    dev = cls_score.device
    out = torch.empty((2,), device=dev, dtype=torch.float32)
    grad = torch.empty_like(cls_score) if need_grad else None
    scratch = torch.empty((2 * N,), device=dev, dtype=torch.float32)
    check(lib().dm_softmax_ce_fwd_bwd(_p(cls_score), _p(labels), _p(weight), N, C, float(scale), _p(scratch), _p(out[0:1]),
                                      _p(out[1:2]), _p(grad), _stream()), 'dm_softmax_ce_fwd_bwd')
    return out[0], out[1], grad


def l1_loss_pos(bbox_pred, labels, targets, weights, num_classes, scale, need_grad=True):
    """-> (loss [1], grad [N, NB*4] or None); NB = bbox_pred.shape[1] // 4."""
    _chk(bbox_pred, 'bbox_pred')
    _chk(labels, 'labels', torch.int64)
    _chk(targets, 'targets')
    _chk(weights, 'weights')
    N = bbox_pred.shape[0]
    nb = bbox_pred.shape[1] // 4
    dev = bbox_pred.device
    loss = torch.empty((1,), device=dev, dtype=torch.float32)
    grad = torch.empty_like(bbox_pred) if need_grad else None
    scratch = torch.empty((N,), device=dev, dtype=torch.float32)
    check(lib().dm_l1_loss_fwd_bwd(_p(bbox_pred), _p(labels), _p(targets), _p(weights), N, nb, int(num_classes), float(scale),
                                   _p(scratch), _p(loss), _p(grad), _stream()), 'dm_l1_loss_fwd_bwd')
    return loss[0], grad


def sumsq(x, out=None):
    """Sum of squares of a flat fp32 buffer (fixed-order) -> device scalar [1]."""
    _chk(x, 'x')
    out = torch.empty((1,), device=x.device, dtype=torch.float32) if out is None else out
    scratch = torch.empty((int(lib().dm_sumsq_scratch_floats()),), device=x.device, dtype=torch.float32)
    check(lib().dm_sumsq(_p(x), x.numel(), _p(scratch), _p(out), _stream()), 'dm_sumsq')
    return out


def clip_scale_(x, sumsq_total, max_norm):
    """x *= min(1, max_norm / (sqrt(sumsq_total) + 1e-6)); the coefficient is taken on the device."""
    _chk(x, 'x')
    _chk(sumsq_total, 'sumsq')
    check(lib().dm_clip_scale(_p(x), x.numel(), _p(sumsq_total), float(max_norm), _stream()), 'dm_clip_scale')
    return x


def scale_(x, factor):
    """x *= factor (x: a contiguous run of fp32, e.g. a slice of a flat gradient buffer)."""
    _chk(x, 'x')
    check(lib().dm_scale(_p(x), x.numel(), float(factor), _stream()), 'dm_scale')
    return x


# ------------------------------------------------------- FCNMaskHead upsample layers: backward
def carafe_backward(x, enc, grad_out, up_kernel=5, group=1, scale=2):
    """-> (grad_x, grad_enc) of ``carafe`` (mask-head shape: scale 2, H*W <= 256)."""
    _chk(x, 'x')
    _chk(enc, 'enc')
    _chk(grad_out, 'grad_out')
    NB, C, H, W = x.shape
    assert grad_out.shape == (NB, C, H * scale, W * scale)
    gx, genc = torch.empty_like(x), torch.empty_like(enc)
    scratch = torch.empty((int(lib().dm_carafe_bwd_scratch_floats(NB, H, W, up_kernel, group)),), device=x.device,
                          dtype=torch.float32)
    check(lib().dm_carafe_bwd(_p(x), _p(enc), _p(grad_out), NB, C, H, W, up_kernel, group, scale, _p(gx), _p(genc),
                              _p(scratch), _stream()), 'dm_carafe_bwd')
    return gx, genc


def upsample2x_nearest(x):
    _chk(x, 'x')
    NB, C, H, W = x.shape
    out = torch.empty((NB, C, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
    check(lib().dm_upsample2x_nearest_fwd(_p(x), NB * C, H, W, _p(out), _stream()), 'dm_upsample2x_nearest_fwd')
    return out


def upsample2x_nearest_backward(grad_out):
    _chk(grad_out, 'grad_out')
    NB, C, OH, OW = grad_out.shape
    gin = torch.empty((NB, C, OH // 2, OW // 2), device=grad_out.device, dtype=torch.float32)
    check(lib().dm_upsample2x_nearest_bwd(_p(grad_out), NB * C, OH // 2, OW // 2, _p(gin), _stream()),
          'dm_upsample2x_nearest_bwd')
    return gin


def pixel_unshuffle2x(x):
    """[N, C, 2H, 2W] -> [N, 4C, H, W] with channel (dy*2+dx)*C + c."""
    _chk(x, 'x')
    NB, C, OH, OW = x.shape
    out = torch.empty((NB, 4 * C, OH // 2, OW // 2), device=x.device, dtype=torch.float32)
    check(lib().dm_pixel_unshuffle2x(_p(x), NB, C, OH // 2, OW // 2, _p(out), _stream()), 'dm_pixel_unshuffle2x')
    return out
