"""Three convolutions alone, a few launches each, for rocprofv3 --pmc SQ_* passes (what are the waves of the
memory-bound 1x1 GEMMs doing?):  python3 tools/sq_probe.py  under  rocprofv3 --pmc <counters> --output-format csv -d <dir> --
   a: 576 -> 64 @56x56 x128 (+ReLU)   b: 64 -> 576 @56x56 x256   c: conv3x3 256 -> 256 @14x14 x512 (+bias, ReLU)
The kernels are told apart by their template arguments: <1,2,2,1,2,16> / <1,2,2,2,2,16> / <3,2,2,2,2,8>."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops
dev = torch.device('cuda')
torch.manual_seed(0)
it = int(os.environ.get('PROBE_ITERS', '4'))


def run(N, cin, cout, S, ks, relu, bias):
    x = torch.randn(N, cin, S, S, device=dev)
    w = torch.randn(cout, cin, ks, ks, device=dev) / (cin * ks * ks) ** 0.5
    wq = ops.pack_conv_weight(w)
    b = torch.randn(cout, device=dev) if bias else None
    out = torch.empty(N, cout, S, S, device=dev)
    for _ in range(it):
        ops.conv2d([x], wq, b, cout, ks, relu=relu, out=out)
    torch.cuda.synchronize()


run(128, 576, 64, 56, 1, True, False)
run(256, 64, 576, 56, 1, False, False)
run(512, 256, 256, 14, 3, True, True)
print('done')
