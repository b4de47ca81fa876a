"""Standalone attempt at the packed-fp32 miscompute of profiles/r05_race_hunt.txt: the shipped
class_logits backward (the wave kernel behind dm_class_logits_bwd_slab, whose `v_pk_fma_f32 ... op_sel:[0,1,0]` dropped a
product in lane 48 inside the four-stream training step) at the step's 14 x 14 shape, launched again and again on one stream
and compared BIT FOR BIT with a run alone on the GPU, while three other streams run the step's real neighbours -- the DCN
col2im (LDS atomics), a 3x3 weight-gradient GEMM (MFMA) and the gather-form RoIAlign adjoint (float atomics) -- not
synthetic busy loops.  Run it against a library built WITH packed fp32 and against the product:

  python -m dynamask_amd.build --packed-fp32                   # -> dynamask_amd/libdynamask_hip_pk.so
  DM_ALLOW_PACKED_FP32=1 DYNAMASK_HIP_LIB=dynamask_amd/libdynamask_hip_pk.so python tools/pk_repro.py [iterations]
  python tools/pk_repro.py [iterations]
"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import _lib, ops, synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device('cuda')
g = torch.Generator().manual_seed(7)
N, C, S, NC = 256, 256, 14, 80
x = torch.randn(N, C, S, S, generator=g).to(dev)
wi, wd = torch.randn(NC, C, generator=g).to(dev), torch.randn(NC, C, generator=g).to(dev)
labels = torch.randint(0, 6, (N,), generator=g).to(dev)            # the RoIs of an image share a few classes
gi_, gd_ = torch.randn(N, 1, S, S, generator=g).to(dev), torch.randn(N, 1, S, S, generator=g).to(dev)


def clb():
    gx = torch.zeros_like(x)
    outs = [torch.zeros(NC, C, device=dev), torch.zeros(NC, device=dev), torch.zeros(NC, C, device=dev), torch.zeros(NC, device=dev)]
    ops.class_logits_backward(x, wi, wd, labels, gi_, gd_, gx, False, *outs)
    return [gx] + outs


# the neighbours, at the shapes they have in the step (256 RoIs)
colgrad = torch.randn(N, 9 * 64, 56, 56, generator=g).to(dev) * 0.1
off56 = (torch.randn(N, 36, 56, 56, generator=g) * 0.5).to(dev)
dy = torch.randn(N, 256, S, S, generator=g).to(dev)
feat_shape = (2, 128, 200, 336)
rois = synth.make_rois(2, 128, 800, 1333, seed=11).to(dev)
go56 = torch.randn(N, 128, 56, 56, generator=g).to(dev) * 0.1
side = [torch.cuda.Stream() for _ in range(3)]
neigh = [lambda: ops.deform_col2im(colgrad, off56, (N, 64, 56, 56), 2),
         lambda: ops.conv2d_wgrad(dy, [x], 3),
         lambda: ops.roi_align_backward(go56, [feat_shape], rois, 56, [0.25])]

torch.cuda.synchronize()
ref = clb()
torch.cuda.synchronize()
again = clb()
torch.cuda.synchronize()
assert all(torch.equal(a, b) for a, b in zip(ref, again)), 'the call alone is not reproducible: nothing to compare with'
info = _lib.lib().dm_build_info().decode()
print('library:', _lib.LIB_PATH)
print('build  :', info)
bad, worst, t0 = 0, 0.0, time.time()
names = ['grad_x', 'gw_inst', 'gb_inst', 'gw_det', 'gb_det']
for it in range(iters):
    main = torch.cuda.current_stream()
    for s, fn in zip(side, neigh):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            for _ in range(2):
                fn()
    got = [clb() for _ in range(6)]           # six calls beside the neighbours' launches
    torch.cuda.synchronize()
    for outs in got:
        for nm, a, b in zip(names, outs, ref):
            if not torch.equal(a, b):
                d = (a - b).abs()
                bad += 1
                worst = max(worst, float(d.max()))
                if bad <= 8:
                    idx = int(d.flatten().argmax())
                    print(f'iteration {it}: {nm} differs in {int((d > 0).sum())} elements, max |diff| {float(d.max()):.3e} at flat index {idx}')
    if (it + 1) % 100 == 0:
        print(f'{it + 1} iterations ({6 * (it + 1)} calls beside busy queues): {bad} differing outputs, {time.time() - t0:.0f} s', flush=True)
print(f'RESULT: {bad} differing outputs in {6 * iters} calls (max |diff| {worst:.3e}); packed fp32 in this build: {"-packed-fp32-ops" not in info}')
