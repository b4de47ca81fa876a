"""The stream-hazard tracker's core (dynamask_amd/hazard.py) on synthetic launches: no GPU needed.

SURVEY 5.2 asked for race evidence; the reference runs one stream (deform_conv_cuda_kernel.cu:265) and has no
equivalent.  tests/test_hazard_gpu.py runs the real training step under the tracker."""
import os

import pytest

from dynamask_amd import hazard
from dynamask_amd.hazard import Pattern, Tracker

MAIN, SIDE, OTHER = 1, 2, 3


def acc(key, lo, n, label='t', span=None):
    return (key, key, key + (span or max(n, 1)), Pattern(lo, n), label)


def test_header_roles_cover_every_prototype_and_mark_const_pointers_as_reads():
    from dynamask_amd import _lib
    roles = hazard.parse_header()
    assert set(roles) == set(_lib.SIGNATURES)
    for name, r in roles.items():
        assert len(r) == len(_lib.SIGNATURES[name][0]), name
        for role, ct in zip(r, _lib.SIGNATURES[name][0]):
            if role in ('in', 'out', 'in[]', 'out[]', 'stream'):
                assert ct in (_lib._vp, _lib.ctypes.c_char_p), (name, role, ct)
    r = roles['dm_conv2d_fwd']
    assert r[0] == 'in[]' and r[7] == 'in' and r[12] == 'out' and r[-1] == 'stream'
    assert roles['dm_roi_align_bwd'][1] == 'out[]'


def test_read_after_write_on_another_stream_needs_a_wait():
    t = Tracker()
    t.launch(MAIN, 'produce', writes=[acc(0x1000, 0x1000, 256)])
    t.launch(SIDE, 'consume', reads=[acc(0x1000, 0x1000, 256)])
    assert len(t.reports) == 1 and t.reports[0].startswith('read-after-write')
    t = Tracker()
    t.launch(MAIN, 'produce', writes=[acc(0x1000, 0x1000, 256)])
    t.wait(SIDE, t.record(MAIN))
    t.launch(SIDE, 'consume', reads=[acc(0x1000, 0x1000, 256)])
    assert t.reports == []


def test_an_event_recorded_before_the_write_does_not_order_it():
    t = Tracker()
    ev = t.record(MAIN)
    t.launch(MAIN, 'produce', writes=[acc(0x1000, 0x1000, 256)])
    t.wait(SIDE, ev)
    t.launch(SIDE, 'consume', reads=[acc(0x1000, 0x1000, 256)])
    assert len(t.reports) == 1


def test_write_after_read_and_transitive_order():
    t = Tracker()
    t.launch(MAIN, 'produce', writes=[acc(0x1000, 0x1000, 256)])
    t.wait(SIDE, t.record(MAIN))
    t.launch(SIDE, 'consume', reads=[acc(0x1000, 0x1000, 256)])
    t.launch(MAIN, 'overwrite', writes=[acc(0x1000, 0x1000, 256)])
    assert len(t.reports) == 1 and t.reports[0].startswith('write-after-read')
    # ordered through a third stream: main -> side -> other
    t = Tracker()
    t.launch(MAIN, 'produce', writes=[acc(0x1000, 0x1000, 256)])
    t.wait(SIDE, t.record(MAIN))
    t.launch(SIDE, 'middle', writes=[acc(0x9000, 0x9000, 16)])
    t.wait(OTHER, t.record(SIDE))
    t.launch(OTHER, 'consume', reads=[acc(0x1000, 0x1000, 256)])
    assert t.reports == []


def test_disjoint_rows_and_interleaved_channel_slices_do_not_conflict():
    t = Tracker()
    # two halves of one buffer written by two streams (the training forward fills its buffers by rows)
    t.launch(MAIN, 'rows0', writes=[acc(0x1000, 0x1000, 512, span=1024)])
    t.launch(SIDE, 'rows1', writes=[acc(0x1000, 0x1200, 512, span=1024)])
    assert t.reports == []
    # channel slices [:, :6] and [:, 6:] of an [N, 8, ...] tensor: same stride, interleaved
    a = (0x8000, 0x8000, 0x8000 + 4 * 800, Pattern(0x8000, 600, 4, 800), 'a')
    b = (0x8000, 0x8000, 0x8000 + 4 * 800, Pattern(0x8000 + 600, 200, 4, 800), 'b')
    t.launch(MAIN, 'lo', writes=[a])
    t.launch(SIDE, 'hi', writes=[b])
    assert t.reports == []
    c = (0x8000, 0x8000, 0x8000 + 4 * 800, Pattern(0x8000 + 500, 200, 4, 800), 'c')
    t.launch(OTHER, 'straddle', reads=[c])
    assert len(t.reports) == 2           # overlaps both slices


def test_host_synchronisation_orders_everything_before_it():
    t = Tracker()
    t.launch(SIDE, 'produce', writes=[acc(0x1000, 0x1000, 256)])
    t.host_sync()
    t.launch(MAIN, 'consume', reads=[acc(0x1000, 0x1000, 256)])
    assert t.reports == []


def test_recycled_storage_under_a_side_stream_is_reported_unless_guarded():
    t = Tracker()
    t.launch(MAIN, 'produce', writes=[acc(0x1000, 0x1000, 256)])
    t.wait(SIDE, t.record(MAIN))
    t.launch(SIDE, 'slow reader', reads=[acc(0x1000, 0x1000, 256)])
    # the tensor is dropped; the allocator hands its block to a new tensor of the main stream
    t.launch(MAIN, 'new tenant', writes=[acc(0x1080, 0x1080, 64)])
    assert len(t.reports) == 1 and t.reports[0].startswith('recycled')
    t = Tracker()
    t.launch(MAIN, 'produce', writes=[acc(0x1000, 0x1000, 256)])
    t.wait(SIDE, t.record(MAIN))
    t.launch(SIDE, 'slow reader', reads=[acc(0x1000, 0x1000, 256)])
    t.guard_stream(0x1000, SIDE)          # record_stream(side): the allocator waits for the side stream
    t.launch(MAIN, 'new tenant', writes=[acc(0x1080, 0x1080, 64)])
    assert t.reports == []


def test_engine_handoff_models_the_wait_autograd_inserts():
    t = Tracker()
    t.launch(SIDE, 'grad producer', writes=[acc(0x1000, 0x1000, 256)])
    t.handoff(MAIN, 0x1000, Pattern(0x1000, 256))
    t.launch(MAIN, 'backward node', reads=[acc(0x1000, 0x1000, 256)])
    assert t.reports == []


def test_records_do_not_grow_without_bound():
    t = Tracker()
    for i in range(2000):
        t.launch(MAIN if i % 2 else SIDE, 'k', reads=[acc(0x1000, 0x1000, 256)])
        if i % 2:
            t.wait(SIDE, t.record(MAIN))
        else:
            t.wait(MAIN, t.record(SIDE))
    assert len(t.records[0x1000]) < 64


def test_handoff_waits_for_the_producer_node_only_and_reports_unjoined_side_stream_gradients():
    """the autograd engine orders a consumer behind the PRODUCER NODE's stream as of its return, not behind
    every stream that wrote the gradient.  A backward that finishes a gradient on a side stream and returns without
    joining it is reported; one that joins first is not, and its consumer ends up ordered behind the side stream too."""
    from dynamask_amd.hazard import Pattern, Tracker
    pat = Pattern(0, 1024)
    ent = lambda key: (key, key, key + 1024, pat, 'g')          # noqa: E731
    # -- producer node on stream 1 writes the gradient on side stream 2 and returns WITHOUT a join
    t = Tracker()
    t.launch(1, 'producer_main')
    t.wait(2, t.record(1))
    t.launch(2, 'producer_side', writes=[ent(4096)])
    t.returned_from(1, 4096)
    t.handoff(3, 4096, pat)
    assert len(t.reports) == 1 and 'unjoined gradient' in t.reports[0] and 'producer_side' in t.reports[0]
    t.launch(3, 'consumer', reads=[ent(4096)])                   # ... and the read itself is a hazard as well
    assert any('read-after-write' in r for r in t.reports)
    # -- the same with the join in front of the return: clean, and the consumer is ordered behind the side stream
    t = Tracker()
    t.launch(1, 'producer_main')
    t.wait(2, t.record(1))
    t.launch(2, 'producer_side', writes=[ent(4096)])
    t.wait(1, t.record(2))
    t.returned_from(1, 4096)
    t.handoff(3, 4096, pat)
    t.launch(3, 'consumer', reads=[ent(4096)])
    assert t.reports == []
    # -- a gradient of one of torch's own nodes (no return record): the lenient rule of before
    t = Tracker()
    t.launch(2, 'torch_node', writes=[ent(8192)])
    t.handoff(3, 8192, pat)
    t.launch(3, 'consumer', reads=[ent(8192)])
    assert t.reports == []
