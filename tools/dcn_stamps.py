"""Phase timeline of the fused DCN data gradient (workgroup 0, first channel block): needs the stamp build,
  hipcc ... -DDM_DCN_STAMPS (tools/dcn_stamps.sh builds a second library beside the product's and points DYNAMASK_HIP_LIB at it)."""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops, _lib

N, C, S, dg = (int(v) for v in (sys.argv[1:5] + ['256', '64', '56', '2'][len(sys.argv) - 1:]))
dev = torch.device('cuda')
g = torch.Generator().manual_seed(3)
x = torch.randn(N, C, S, S, generator=g).to(dev)
off = torch.zeros(N, 18 * dg, S, S, device=dev)
go = torch.randn(N, C, S, S, generator=g).to(dev)
w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).to(dev)
wf = ops.pack_dcn_bwd_weight(w, dg)
for _ in range(3):
    ops.deform_conv_backward_data_fused(x, off, go, wf, dg)
torch.cuda.synchronize()
lib = _lib.lib()
lib.dm_dcn_stamps.argtypes = [ctypes.c_void_p]
buf = np.zeros(64 * 10 * 8, dtype=np.uint64)
assert lib.dm_dcn_stamps(buf.ctypes.data) == 0
st = buf.reshape(64, 10, 8).astype(np.int64)
bands = (S + 3) // 4
t0 = st[0, :, 0].min()
names = ['units', 'wait barrier 1', 'flush', 'stage next band', 'reload A', 'wait barrier 2']
print(f'{N}x{C}x{S}x{S}: workgroup 0, block 0; s_memtime ticks (100 MHz -> x 24 = shader cycles at 2.4 GHz); per band, mean over the ten waves [min..max]')
tot = np.zeros(6)
for b in range(bands):
    d = np.diff(st[b, :, :7], axis=1)
    tot += d.mean(axis=0)
    print(f'band {b:2d} starts {int(st[b, :, 0].min() - t0):6d}: ' + '  '.join(f'{names[k]} {d[:, k].mean():6.0f} [{d[:, k].min()}..{d[:, k].max()}]' for k in range(6)))
print('sum over bands (ticks):', '  '.join(f'{names[k]} {tot[k]:.0f}' for k in range(6)), ' total', int(st[bands - 1, :, 6].max() - t0))
