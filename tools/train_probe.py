import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import bench
dev = torch.device('cuda')
head, sd = bench.build_head(dev)
ms, loss, n, b = bench.train_step_bench(head, dev, 0, 1, steps=2, warmup=1)
print('train ms', ms)
