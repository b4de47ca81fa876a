"""Where each RoI chain of the 100-detection inference call is at every launch, WITHOUT the profiler (rocprofv3's kernel
trace serialises queues differently from a plain run): eager mode, a torch event behind every launch of every chain
(DynaMaskRoIHead._launch_hook), times relative to the fork.  Prints one line per launch and chain.
  python tools/chain_probe.py [detections] [reps]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device('cuda')
head, sd = bench.build_head(dev)
feats_c, rois_c, labels_c = bench.make_inputs(0, dev)
feats = [f.to(dev) for f in feats_c]; rois = rois_c.to(dev); labels = labels_c.to(dev)
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 100
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
det, dl = rois[:nd, 1:].contiguous(), labels[:nd].contiguous()
head.enable_inference_graphs(False)
call = lambda: head.simple_test_mask_logits(feats, det, dl)
with torch.no_grad():
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    print(f'{nd} detections, eager: {bench.time_kernel_median(call, iters=9, warmup=2):.4f} ms without events')
    runs = []
    for _ in range(reps):
        log = []
        def hook(st, log=log):
            e = torch.cuda.Event(enable_timing=True)
            e.record(st)
            log.append((st.cuda_stream, e))
        head._launch_hook = hook
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        call()
        t1.record()
        torch.cuda.synchronize()
        head._launch_hook = None
        runs.append((t0.elapsed_time(t1), [(s, t0.elapsed_time(e)) for s, e in log]))
    runs.sort(key=lambda r: r[0])
    total, log = runs[len(runs) // 2]
    print(f'median run with events: {total:.4f} ms (all: {", ".join(f"{r[0]:.3f}" for r in runs)})')
    ids = sorted({s for s, _ in log})
    per = {s: [t for s2, t in log if s2 == s] for s in ids}
    n = max(len(v) for v in per.values())
    print('launch   ' + '   '.join(f'chain{i} done at (dt)' for i in range(len(ids))))
    for j in range(n):
        cells = []
        for s in ids:
            v = per[s]
            cells.append(f'{v[j] * 1e3:8.1f} ({(v[j] - (v[j - 1] if j else 0.0)) * 1e3:6.1f})' if j < len(v) else ' ' * 17)
        print(f'{j:4d}   ' + '   '.join(cells))
