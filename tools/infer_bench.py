"""Inference legs only, quickly: the headline step (512 RoIs, 28x28 exit, graph replay) and simple_test_mask_logits at
100 / 64 / 32 / 16 detections, eager and as the bucketed HIP graph (bench.py's own timers).  One line per figure.
  python tools/infer_bench.py [reps]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device('cuda')
head, sd = bench.build_head(dev)
feats_c, rois_c, labels_c = bench.make_inputs(0, dev)
feats = [f.to(dev) for f in feats_c]; rois = rois_c.to(dev); labels = labels_c.to(dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with torch.no_grad():
    step = lambda: head._mask_forward(feats, rois, labels, last_stage=1)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    hs = sorted(bench.time_kernel(g.replay, iters=20, warmup=3) for _ in range(reps))
    print(f'headline_512rois_28exit_graph_ms {hs[len(hs) // 2]:.4f}   (all: {", ".join(f"{v:.4f}" for v in hs)})')
    print(f'full_head_112_eager_ms {bench.time_kernel(lambda: head._mask_forward(feats, rois, labels), iters=5, warmup=2):.4f}')
    for nd in (100, 64, 32, 16):
        det, dl = rois[:nd, 1:].contiguous(), labels[:nd].contiguous()
        call = lambda: head.simple_test_mask_logits(feats, det, dl)
        head.enable_inference_graphs(False)
        ref = call().clone()
        e = sorted(bench.time_kernel_median(call, iters=9, warmup=2) for _ in range(reps))
        head.enable_inference_graphs(True)
        got = call()
        # (eager calls split their RoIs over two streams from 80 detections on, captured ones from 32: where the chunking differs the
        # split-K choices of the launches differ, and with them the association of the channel sums -- rounding only)
        same = 'True' if bool(torch.equal(ref, got)) else f'no (max |d| {float((ref - got).abs().max()):.1e}: other chunking, other split-K sums)'
        gr = sorted(bench.time_kernel_median(call, iters=15, warmup=3) for _ in range(reps))
        head.enable_inference_graphs(False)
        print(f'infer_{nd}dets  eager {e[len(e) // 2]:.4f} ms   graph {gr[len(gr) // 2]:.4f} ms   graph==eager {same}   '
              f'(eager all: {", ".join(f"{v:.3f}" for v in e)}; graph all: {", ".join(f"{v:.3f}" for v in gr)})')
