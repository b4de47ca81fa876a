"""Does the DCN backward's column gradient have to travel through HBM?  The 56x56 stage writes 1.85 GB of it
(W^T . dY), then reads it twice (coordinate gradient, col2im).  Walked in chunks of RoIs through ONE reused buffer
that fits the 256 MB Infinity Cache, the three kernels of a chunk could meet it there.  Times the chain for several
chunk sizes (same kernels, same results), and the forward pair im2col -> GEMM the same way."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops
dev = torch.device('cuda')


def t(fn, iters=5, warmup=2):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for C, S, N in ((64, 56, 256), (128, 28, 256), (256, 14, 256)):
    torch.manual_seed(0)
    x = torch.randn(N, C, S, S, device=dev)
    off = torch.randn(N, 18, S, S, device=dev) * 1.5
    w = torch.randn(C, C, 3, 3, device=dev) / (9 * C) ** 0.5
    dy = torch.randn(N, C, S, S, device=dev)
    wcg = ops.pack_dcn_colgrad_weight(w)
    wf = ops.pack_conv_weight(ops.dcn_weight_permute(w, C, C, True).reshape(9 * C, C).t().contiguous().reshape(C, 9 * C, 1, 1))
    side = torch.cuda.Stream()
    ref = None
    for chunk in (N, 128, 64, 32, 16, 8):
        buf = torch.empty(chunk, 9 * C, S, S, device=dev)
        gx, goff = torch.empty_like(x), torch.empty_like(off)

        def bwd(two_streams):
            main = torch.cuda.current_stream()
            for i in range(0, N, chunk):
                sl = slice(i, i + chunk)
                if two_streams:
                    main.wait_stream(side)           # the buffer is free again
                ops.conv2d(dy[sl], wcg, None, 9 * C, 1, out=buf)
                if two_streams:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        ops.deform_coord_grad(buf, x[sl], off[sl], 1, out=goff[sl])
                else:
                    ops.deform_coord_grad(buf, x[sl], off[sl], 1, out=goff[sl])
                ops.deform_col2im(buf, off[sl], tuple(x[sl].shape), 1, out=gx[sl])
            if two_streams:
                main.wait_stream(side)
        out = torch.empty(N, C, S, S, device=dev)

        def fwd():
            for i in range(0, N, chunk):
                sl = slice(i, i + chunk)
                ops.deform_im2col(x[sl], off[sl], 1, out=buf)
                ops.conv2d(buf, wf, None, C, 1, relu=True, out=out[sl])
        gw = torch.zeros(C, 9 * C, 1, 1, device=dev)

        def wgrad():
            for i in range(0, N, chunk):
                sl = slice(i, i + chunk)
                ops.deform_im2col(x[sl], off[sl], 1, out=buf)
                ops.conv2d_wgrad(dy[sl], buf, 1, dw=gw)
        t1, t2, t3, t4 = t(lambda: bwd(False)), t(lambda: bwd(True)), t(fwd), t(wgrad)
        bwd(False); torch.cuda.synchronize()
        if ref is None:
            ref = (gx.clone(), goff.clone(), out.clone())
        same = torch.equal(gx, ref[0]) and torch.equal(goff, ref[1]) and torch.equal(out, ref[2])
        print(f'C {C:3d} {S}x{S} x{N}  chunk {chunk:3d} ({buf.numel() * 4 / 2**20:6.0f} MiB)  backward-data {t1:.3f} ms, two streams {t2:.3f} ms | '
              f'im2col+GEMM forward {t3:.3f} ms | im2col+wgrad {t4:.3f} ms | same bits {same}', flush=True)
        del buf
