"""Host-core probe for the CPU baseline: visible cores, cgroup quota, oracle throughput vs threads."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import bench
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'torch threads', torch.get_num_threads())
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
    if os.path.exists(f):
        print(f, open(f).read().strip())
from dynamask_amd import synth
sd = {**synth.init_dynamask_head_state(seed=5, test_mode=True), **synth.init_mask_pre_state(seed=6)}
feats, rois, labels = bench.make_inputs(0, None)
for nt in (1, 4, 8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    n = 64 if nt > 1 else 8
    dt, _, _ = bench.cpu_baseline(sd, feats, rois, labels, n)
    print(f'threads {nt:4d}: {n} RoIs in {dt:.3f} s -> {dt / n * 1e3:.1f} ms/RoI', flush=True)
