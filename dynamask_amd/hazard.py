"""DM_HAZARD=1: host-side cross-stream hazard tracker for the C-ABI launches (a debugging aid, off by default).

The training step runs on four HIP streams (train_path.py); the reference runs on one
(mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:265), so it never needed this.  Every launch of the library goes
through ``ops._p`` (tensor -> pointer) and ``_lib.lib()``; with the tracker on, ``lib()`` hands out a proxy that
knows, from the prototypes of include/dynamask_hip.h, which pointer arguments a call reads (``const T*``) and which it
writes (``T*``: treated as read-modify-write), and this module keeps

  * a vector clock per stream (what each stream is ordered behind), advanced by launches and merged by the event /
    stream waits of ``torch.cuda`` (patched while the tracker is on),
  * per device storage the accesses not yet ordered before every stream,

and reports a launch whose memory overlaps an access of ANOTHER stream that it is not ordered behind
(read-after-write, write-after-read, write-after-write), as well as a storage that takes over the addresses of an
earlier one while a stream the new owner is not ordered behind may still be using it (a tensor recycled by the
caching allocator under a side stream: allocator pools are per stream, ``record_stream`` is what makes that safe).

What it cannot see: kernels torch launches itself (zeros, clone, add_ ...), unless told (``touch``), and the stream
waits the autograd engine inserts between nodes -- ``engine_handoff`` models those at the entry of a custom backward
(the engine makes the node's stream wait for the producers of its incoming gradients).  A missing sight only ever
hides a hazard or -- for an unseen wait -- invents one; reports name both launches so either can be checked by hand.

The core (``Tracker``) is plain Python over integer stream ids and byte ranges: tests/test_hazard_cpu.py drives it
without a GPU; tests/test_hazard_gpu.py runs the training step under it.
"""
import os
import re
import weakref

ENABLED = [os.environ.get('DM_HAZARD', '0') not in ('', '0')]
RAISE = [os.environ.get('DM_HAZARD', '0') == '2']

_HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'dynamask_hip.h')


class HazardError(RuntimeError):
    pass


# --------------------------------------------------------------------------- prototypes -> argument roles
def parse_header(path=_HEADER):
    """{function: [role, ...]} with role in 'in' (const T*), 'out' (T*), 'in[]' / 'out[]' (host array of device
    pointers), 'stream', 'scalar'."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = {}
    for name, args in re.findall(r'\b(?:int|long long|const char\*)\s+(dm_\w+)\s*\((.*?)\)\s*;', src, flags=re.S):
        roles = []
        for a in [' '.join(x.split()) for x in args.split(',')]:
            if a in ('void', ''):
                continue
            if a.startswith('dm_stream_t'):
                roles.append('stream')
            elif '*' not in a:
                roles.append('scalar')
            elif a.count('*') == 2:
                roles.append('in[]' if a.startswith('const') else 'out[]')
            else:
                roles.append('in' if a.startswith('const') else 'out')
        out[name] = roles
    return out


# --------------------------------------------------------------------------- access patterns
class Pattern:
    """``n`` runs of ``inner`` bytes, ``stride`` bytes apart, from ``lo`` (n == 1: one dense run)."""
    __slots__ = ('lo', 'n', 'stride', 'inner')

    def __init__(self, lo, inner, n=1, stride=0):
        self.lo, self.inner, self.n, self.stride = int(lo), int(inner), int(n), int(stride)
        if self.n <= 1 or self.stride <= self.inner:          # dense after all
            self.inner = self.inner if self.n <= 1 else self.stride * (self.n - 1) + self.inner
            self.n, self.stride = 1, 0

    @property
    def hi(self):
        return self.lo + (self.stride * (self.n - 1) if self.n > 1 else 0) + self.inner

    def overlaps(self, o):
        if self.inner <= 0 or o.inner <= 0 or self.hi <= o.lo or o.hi <= self.lo:
            return False
        if self.n == 1 and o.n == 1:
            return True
        if self.n > 1 and o.n > 1 and self.stride == o.stride:
            # two row-strided views of one buffer (channel slices of an NCHW tensor): compare inside a period
            d = (o.lo - self.lo) % self.stride
            return d < self.inner or (self.stride - d) < o.inner
        a, b = (self, o) if self.n > 1 else (o, self)           # a strided, b dense (or another stride: its bounding run)
        blo, bhi = b.lo, b.hi
        k0 = max(0, (blo - a.lo - a.inner) // a.stride)
        for k in range(k0, min(a.n, k0 + 3 + (bhi - blo) // a.stride)):
            s = a.lo + k * a.stride
            if s < bhi and blo < s + a.inner:
                return True
        return False

    def covers(self, o):
        return self.n == 1 and self.lo <= o.lo and o.hi <= self.hi

    def __repr__(self):
        return f'[{self.lo:#x}+{self.inner}]' if self.n == 1 else f'[{self.lo:#x}+{self.inner} x{self.n} /{self.stride}]'


def tensor_pattern(t):
    """Bytes a kernel may touch through ``t``: dense tensors one run; a batch-strided channel slice n runs."""
    if t.numel() == 0:
        return Pattern(t.data_ptr(), 0)
    es = t.element_size()
    if t.is_contiguous():
        return Pattern(t.data_ptr(), t.numel() * es)
    if t.dim() >= 2 and t[0].is_contiguous():
        return Pattern(t.data_ptr(), t[0].numel() * es, t.shape[0], t.stride(0) * es)
    span = sum((s - 1) * st for s, st in zip(t.shape, t.stride())) + 1
    return Pattern(t.data_ptr(), span * es)


class Access:
    __slots__ = ('pat', 'stream', 'tick', 'clock', 'write', 'what', 'key')

    def __init__(self, pat, stream, tick, clock, write, what, key):
        self.pat, self.stream, self.tick, self.clock, self.write, self.what, self.key = pat, stream, tick, clock, write, what, key


# --------------------------------------------------------------------------- the tracker
class Tracker:
    def __init__(self):
        self.clock = {}            # stream -> {stream: tick}
        self.host = {}             # what the host has synchronised with: merged into a stream at its next launch
        self.records = {}          # storage key -> [Access]
        self.spans = {}            # storage key -> (lo, hi)
        self.guard = {}            # storage key -> {stream}: record_stream() was called (the allocator orders reuse)
        self.reports = []
        self.returned = {}         # storage key -> (stream, clock snapshot) of the custom backward that returned it
        self.launches = 0
        self.names = {}            # stream id -> label for reports

    # ---- clocks
    def _clk(self, s):
        c = self.clock.get(s)
        if c is None:
            c = self.clock[s] = {s: 0}
        return c

    @staticmethod
    def _merge(dst, src):
        for k, v in src.items():
            if dst.get(k, 0) < v:
                dst[k] = v

    def record(self, s):
        """An event recorded on ``s``: a snapshot of what ``s`` is ordered behind (itself included)."""
        return dict(self._clk(s))

    def wait(self, s, snapshot):
        if snapshot:
            self._merge(self._clk(s), snapshot)

    def host_sync(self, s=None, snapshot=None):
        """The host has waited for stream ``s`` (None: the whole device) or for an event's ``snapshot``."""
        if snapshot is not None:
            self._merge(self.host, snapshot)
        elif s is None:
            for c in list(self.clock.values()):
                self._merge(self.host, c)
        else:
            self._merge(self.host, self._clk(s))

    def ordered(self, s, acc):
        """Is access ``acc`` (of another stream) complete before what ``s`` launches next?"""
        return self._clk(s).get(acc.stream, 0) >= acc.tick or self.host.get(acc.stream, 0) >= acc.tick

    # ---- storages
    def _adopt(self, key, lo, hi, s, what):
        """First sight of a storage: it may have taken over the addresses of earlier ones."""
        self.spans[key] = (lo, hi)
        new = Pattern(lo, hi - lo)
        for k, (klo, khi) in list(self.spans.items()):
            if k == key or khi <= lo or hi <= klo:
                continue
            keep = []
            for r in self.records.get(k, ()):
                if not r.pat.overlaps(new):
                    keep.append(r)
                    continue
                if r.stream != s and not self.ordered(s, r) and r.stream not in self.guard.get(k, ()):
                    self._report('recycled', what, s, r,
                                 'the new tensor occupies memory an earlier tensor was using on another stream, '
                                 'with no record_stream() and no wait between them')
            if keep:
                self.records[k] = keep
            else:
                self.records.pop(k, None)
                self.spans.pop(k, None)
                self.guard.pop(k, None)

    def launch(self, s, name, reads=(), writes=()):
        """One kernel launch on stream ``s``.  reads / writes: iterables of (key, storage_lo, storage_hi, Pattern, label)."""
        self.launches += 1
        c = self._clk(s)
        if self.host:
            self._merge(c, self.host)
        c[s] = tick = c.get(s, 0) + 1
        snap = None
        for is_write, group in ((False, reads), (True, writes)):
            for key, lo, hi, pat, label in group:
                what = f'{name}({label})'
                if key not in self.spans:
                    self._adopt(key, lo, hi, s, what)
                recs = self.records.setdefault(key, [])
                keep = []
                for r in recs:
                    if r.stream != s and (is_write or r.write) and r.pat.overlaps(pat) and not self.ordered(s, r):
                        kind = 'write-after-write' if (is_write and r.write) else ('read-after-write' if r.write else 'write-after-read')
                        self._report(kind, what, s, r, '')
                    # drop what this access supersedes: an ordered (or same-stream) earlier access it covers
                    if (r.stream == s or self.ordered(s, r)) and ((is_write and pat.covers(r.pat)) or
                                                                 (not is_write and not r.write and r.stream == s and pat.covers(r.pat))):
                        continue
                    keep.append(r)
                if snap is None:
                    snap = dict(c)
                keep.append(Access(pat, s, tick, snap, is_write, what, key))
                if len(keep) > 48:
                    keep = self._gc(keep)
                self.records[key] = keep
        return tick

    def _gc(self, recs):
        """Forget accesses every known stream is already ordered behind (they can never conflict again)."""
        streams = list(self.clock)
        return [r for r in recs if not all(st == r.stream or self.ordered(st, r) for st in streams)]

    def returned_from(self, s, key):
        """A custom backward on stream ``s`` hands a gradient (storage ``key``) back to the autograd engine."""
        self.returned[key] = (s, dict(self._clk(s)))

    def handoff(self, s, key, pat):
        """The autograd engine hands a tensor to a node on stream ``s``: it makes ``s`` wait for the PRODUCER NODE's
        stream as of that node's return (torch/csrc/autograd/input_buffer.cpp) -- not for every stream that ever wrote
        the tensor.  Where the producing backward told us its return (``returned_from``), only that snapshot is merged,
        and a write on another stream that the producer had not joined by then is what the tracker exists to find: a
        gradient finished on a side stream behind the engine's back.  Gradients of torch's own nodes (no
        return record) keep the lenient rule: wait for whoever wrote them."""
        ret = self.returned.pop(key, None)
        if ret is not None:
            rs, snap = ret
            for r in self.records.get(key, ()):
                if r.write and r.pat.overlaps(pat) and r.stream != rs and snap.get(r.stream, 0) < r.tick \
                        and self.host.get(r.stream, 0) < r.tick:
                    self._report('unjoined gradient', f'autograd hand-off of {r.pat}', s, r,
                                 f'the backward that produced it returned on stream {self._label(rs)} without waiting for '
                                 f'that write: the engine orders the consumer behind stream {self._label(rs)} only')
            self._merge(self._clk(s), snap)
            return
        for r in self.records.get(key, ()):
            if r.write and r.stream != s and r.pat.overlaps(pat):
                self._merge(self._clk(s), r.clock)

    def guard_stream(self, key, s):
        self.guard.setdefault(key, set()).add(s)

    def _label(self, s):
        return self.names.get(s, hex(s) if isinstance(s, int) else str(s))

    def _report(self, kind, what, s, r, note):
        msg = (f'{kind}: {what} on stream {self._label(s)} touches {r.pat} which {r.what} on stream '
               f'{self._label(r.stream)} (launch #{r.tick} there) {"wrote" if r.write else "read"}; '
               f'stream {self._label(s)} is ordered behind launch #{self._clk(s).get(r.stream, 0)} of that stream only'
               + (f' -- {note}' if note else ''))
        if msg not in self.reports:
            self.reports.append(msg)
        if RAISE[0]:
            raise HazardError(msg)

    def reset(self):
        self.__init__()


TRACKER = Tracker()

# --------------------------------------------------------------------------- torch / ctypes glue
_PENDING = {}          # pointer value -> tensor, filled by ops._p / ops._ptr_array between two library calls
_PENDING_ARRAYS = {}   # id(ctypes array) -> [tensors]
_ROLES = [None]
_PATCHED = [False]


def note_ptr(t):
    """ops._p: remember which tensor a pointer argument came from (the widest view wins on equal addresses)."""
    p = t.data_ptr()
    o = _PENDING.get(p)
    if o is None or tensor_pattern(o).hi < tensor_pattern(t).hi:
        _PENDING[p] = t


def note_ptr_array(arr, tensors):
    _PENDING_ARRAYS[id(arr)] = list(tensors)


def _entry(t, label):
    st = t.untyped_storage()
    lo = st.data_ptr()
    return (lo, lo, lo + st.nbytes(), tensor_pattern(t), label)


def _stream_id(stream_arg=None):
    import torch
    if stream_arg is not None:
        v = getattr(stream_arg, 'value', stream_arg)
        return int(v or 0)
    return int(torch.cuda.current_stream().cuda_stream)


def _capturing():
    import torch
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def touch(name, reads=(), writes=(), stream=None):
    """Tell the tracker about a launch it cannot see (a torch op on tracked memory, a collective, the device-side job
    table of dm_conv_pack_weight_batch).  ``stream``: a torch stream (default: the current one)."""
    if not ENABLED[0] or _capturing():
        return
    install()
    s = _stream_id() if stream is None else int(stream.cuda_stream)
    TRACKER.launch(s, name, [_entry(t, f'r{i}') for i, t in enumerate(reads) if t is not None and t.is_cuda],
                   [_entry(t, f'w{i}') for i, t in enumerate(writes) if t is not None and t.is_cuda])


def engine_handoff(*tensors):
    """At the entry of a custom backward: the autograd engine has made the current stream wait for the streams
    that produced these gradients (torch/csrc/autograd/input_buffer.cpp); model that wait."""
    if not ENABLED[0] or _capturing():
        return
    s = _stream_id()
    for t in tensors:
        if t is not None and getattr(t, 'is_cuda', False):
            TRACKER.handoff(s, t.untyped_storage().data_ptr(), tensor_pattern(t))


def engine_return(outs):
    """At the exit of a custom backward: the gradients it hands back, and the stream (with everything that stream is
    ordered behind) the engine will make their consumers wait for."""
    if not ENABLED[0] or _capturing():
        return
    s = _stream_id()
    for t in (outs if isinstance(outs, (tuple, list)) else (outs,)):
        if t is not None and getattr(t, 'is_cuda', False):
            TRACKER.returned_from(s, t.untyped_storage().data_ptr())


def backward_node(fn):
    """Decorator for the ``backward`` staticmethods of the package's autograd Functions: reports the return to the
    tracker (a plain call when the tracker is off)."""
    import functools

    @functools.wraps(fn)
    def wrapped(ctx, *grads):
        out = fn(ctx, *grads)
        if ENABLED[0]:
            engine_return(out)
        return out
    return wrapped


class _LibProxy:
    """What ``_lib.lib()`` returns while the tracker is on: every dm_* call is reported before it is made."""

    def __init__(self, real):
        self._real = real
        if _ROLES[0] is None:
            _ROLES[0] = parse_header()

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        roles = _ROLES[0].get(name)
        if roles is None or 'stream' not in roles:
            def plain(*args):
                _PENDING.clear()
                _PENDING_ARRAYS.clear()
                return fn(*args)
            return plain

        def call(*args):
            try:
                if not _capturing():
                    reads, writes, s = [], [], 0
                    for i, (role, a) in enumerate(zip(roles, args)):
                        if role == 'stream':
                            s = _stream_id(a)
                        elif role in ('in', 'out'):
                            v = getattr(a, 'value', None)
                            t = _PENDING.get(v) if v else None
                            if t is not None:
                                (writes if role == 'out' else reads).append(_entry(t, f'arg{i}'))
                        elif role in ('in[]', 'out[]'):
                            for j, t in enumerate(_PENDING_ARRAYS.get(id(a), ())):
                                (writes if role == 'out[]' else reads).append(_entry(t, f'arg{i}[{j}]'))
                    TRACKER.launch(s, name, reads, writes)
            finally:
                _PENDING.clear()
                _PENDING_ARRAYS.clear()
            return fn(*args)
        return call


def wrap_lib(real):
    install()
    return _LibProxy(real)


def install():
    """Patch the torch.cuda synchronisation calls the package uses so that the tracker sees them."""
    if _PATCHED[0]:
        return
    import torch
    _PATCHED[0] = True
    ev_record, ev_wait, ev_sync = torch.cuda.Event.record, torch.cuda.Event.wait, torch.cuda.Event.synchronize
    st_sync, dev_sync, rec_stream = torch.cuda.Stream.synchronize, torch.cuda.synchronize, torch.Tensor.record_stream
    snaps = weakref.WeakKeyDictionary()

    def record(self, stream=None):
        if stream is None:
            stream = torch.cuda.current_stream()
        if ENABLED[0] and not _capturing():
            snaps[self] = TRACKER.record(int(stream.cuda_stream))
        return ev_record(self, stream)

    def wait(self, stream=None):
        if stream is None:
            stream = torch.cuda.current_stream()
        if ENABLED[0] and not _capturing():
            TRACKER.wait(int(stream.cuda_stream), snaps.get(self))
        return ev_wait(self, stream)

    def event_synchronize(self):
        r = ev_sync(self)
        if ENABLED[0]:
            TRACKER.host_sync(snapshot=snaps.get(self) or {})
        return r

    def stream_synchronize(self):
        r = st_sync(self)
        if ENABLED[0]:
            TRACKER.host_sync(int(self.cuda_stream))
        return r

    def device_synchronize(device=None):
        r = dev_sync(device)
        if ENABLED[0]:
            TRACKER.host_sync()
        return r

    def record_stream(self, stream):
        if ENABLED[0] and self.is_cuda:
            TRACKER.guard_stream(self.untyped_storage().data_ptr(), int(stream.cuda_stream))
        return rec_stream(self, stream)

    torch.cuda.Event.record, torch.cuda.Event.wait, torch.cuda.Event.synchronize = record, wait, event_synchronize
    torch.cuda.Stream.synchronize, torch.cuda.synchronize, torch.Tensor.record_stream = stream_synchronize, device_synchronize, record_stream


def name_stream(stream, label):
    if ENABLED[0] and stream is not None:
        TRACKER.names[int(stream.cuda_stream)] = label


def reports():
    return list(TRACKER.reports)


def reset():
    TRACKER.reset()
    _PENDING.clear()
    _PENDING_ARRAYS.clear()


if ENABLED[0]:          # DM_HAZARD set in the environment: see every wait from the first one on
    try:
        install()
    except Exception:      # noqa: BLE001  (torch without CUDA bindings: nothing to patch)
        pass
