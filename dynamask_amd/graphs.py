"""HIP-graph replay of the inference mask path for the detection counts of real inference.

The reference infers at most 100 detections per image, every exit run to 112x112 and merged
(roi_heads/dynamask_roi_head.py:117-158; tools/benchmark.py:63-89 is the protocol).  At those sizes the mask path is
~60 dependent launches of 10-100 us each: the Python / ctypes issue of a launch costs about as much as the kernel, so the
literal C-ABI launch sequence of ``simple_test_mask_logits`` is captured once per BUCKET of detection counts
(16 / 24 / 32 / 48 / 64 / 80 / 100, RoIs padded with empty boxes, which the kernels turn into zero rows) and replayed.  No tracing
compiler: a graph holds exactly the launches the eager call makes.

A graph is tied to the addresses it was captured with: the FPN maps' storage, the packed weights (``ops.WEIGHT_EPOCH``
and the (address, version) of every head parameter) and its own static RoI / label / output buffers.  The cache is keyed on all of that; a backbone that hands over its
maps in the same buffers every image (the caching allocator does, for a fixed input size) replays, anything else
captures again.  The returned logits are a view of the graph's static output: consume them before the next call with
the same bucket.
"""
import torch

from . import ops

# (round 6: 24 / 48 / 80 added -- a call is padded up to its bucket, and the call time grows almost linearly with the RoI
# count above 16: 0.61 / 0.79 / 0.90 / 1.21 / 1.39 / 1.93 ms at 16 / 24 / 32 / 48 / 64 / 100 detections, so 33 detections
# in the 48-bucket cost 1.21 ms instead of the 64-bucket's 1.39)
BUCKETS = (16, 24, 32, 48, 64, 80, 100)


class GraphedMaskLogits:
    def __init__(self, roi_head, buckets=BUCKETS, max_graphs=16):
        self.head = roi_head
        self.buckets = tuple(sorted(buckets))
        self.max_graphs = max_graphs
        self._graphs = {}          # key -> (graph, rois_static, labels_static, out_static, [filled rows, column 0 dirty])
        self.captures = 0
        self.replays = 0
        self._params = None

    def bucket_for(self, n):
        for b in self.buckets:
            if n <= b:
                return b
        return None

    def _weights_key(self):
        """(address, version) of every parameter the captured launches read, directly or through a kernel-layout pack:
        an in-place update (optimizer step, load_state_dict, copy_) bumps ``_version`` without touching
        ops.WEIGHT_EPOCH, and a replay would combine stale packed weights with the new class-logit weights.
        The module tree is walked once into slots -- (owner module, parameter name) and (parent, child name, child) --
        and every call re-reads the CURRENT object of each slot (a dict lookup each, ~15 us for the head against ~110 us
        for ``parameters()``): a Parameter that was replaced (``load_state_dict(..., assign=True)``, ``m.weight =
        nn.Parameter(...)``, prune / parametrize re-registration) changes the key through its address, a replaced,
        added or removed submodule or parameter slot rebuilds the slot lists."""
        head = self.head.mask_head
        for _ in range(2):
            if self._params is None:
                mods = list(head.modules())
                self._params = ([(m, n) for m in mods for n, p in m._parameters.items() if p is not None],
                                [(m, n, c) for m in mods for n, c in m._modules.items()],
                                [(m, len(m._parameters), len(m._modules)) for m in mods])
            slots, edges, counts = self._params
            try:
                ok = (all(m._modules[n] is c for m, n, c in edges)
                      and all(len(m._parameters) == a and len(m._modules) == b for m, a, b in counts))
                ps = [m._parameters[n] for m, n in slots] if ok else None
            except KeyError:
                ps = None
            if ps is not None and not any(p is None for p in ps):
                return tuple([p.data_ptr() for p in ps]), sum(p._version for p in ps)
            self._params = None
        raise RuntimeError('the mask head changed while its graph key was built')

    def _key(self, bucket, x):
        return (bucket, tuple(int(t.data_ptr()) for t in x), tuple(tuple(t.shape) for t in x), ops.WEIGHT_EPOCH[0],
                self._weights_key(), torch.cuda.current_device())

    def _capture(self, key, bucket, x):
        dev = x[0].device
        rois = torch.zeros((bucket, 5), device=dev, dtype=torch.float32)          # empty boxes: zero rows
        labels = torch.zeros((bucket,), device=dev, dtype=torch.int64)
        head = self.head

        def run():
            return head._merged_logits(x, rois, labels)
        with torch.no_grad():
            # once eagerly on a side stream: packs the weights and sizes the allocator before the capture
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                run()
            torch.cuda.current_stream(dev).wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = run()
        if len(self._graphs) >= self.max_graphs:
            self._graphs.pop(next(iter(self._graphs)))
        self._graphs[key] = (g, rois, labels, out, [0, False])
        self.captures += 1
        return self._graphs[key]

    def __call__(self, x, mask_rois, det_labels, boxes=None):
        """``mask_rois`` [n, 5], ``det_labels`` [n] -> merged logits [n, 1, 112, 112] (None: no bucket holds n).
        ``boxes`` [n, >= 4] instead of ``mask_rois`` (None): the boxes of ONE image -- written straight into columns 1..4 of
        the graph's static RoI buffer (column 0, the batch index, stays 0), one strided copy instead of bbox2roi's fill + cat
        and a copy of their result."""
        src = mask_rois if boxes is None else boxes
        n = src.shape[0]
        bucket = self.bucket_for(n)
        if bucket is None or not src.is_cuda:
            return None
        x = list(x)
        key = self._key(bucket, x)
        entry = self._graphs.get(key) or self._capture(key, bucket, x)
        g, rois, labels, out = entry[:4]
        if boxes is None:
            rois[:n].copy_(mask_rois)
        else:
            if entry[4][1]:                    # the last call left batch indices in column 0
                rois[:, 0].zero_()
            rois[:n, 1:5].copy_(boxes[:, :4])
        filled = entry[4]                      # [rows that may hold a box, column 0 may be non-zero]
        if n < filled[0]:
            rois[n:filled[0]].zero_()          # empty boxes: zero rows (rows past `filled` are zero already)
        filled[0], filled[1] = n, boxes is None
        labels[:n].copy_(det_labels)
        g.replay()
        self.replays += 1
        return out[:n]
