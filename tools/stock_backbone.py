"""Stock PyTorch-ROCm (MIOpen) ResNet-50 + FPN with random weights: NOT part of the
product.  Only used by `bench.py --end-to-end` to put the mask path next to the
(out-of-scope) backbone for an end-to-end img/s figure."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class Bottleneck(nn.Module):
    def __init__(self, cin, mid, stride):
        super().__init__()
        cout = mid * 4
        self.c1 = nn.Conv2d(cin, mid, 1, bias=False)
        self.b1 = nn.BatchNorm2d(mid)
        self.c2 = nn.Conv2d(mid, mid, 3, stride, 1, bias=False)
        self.b2 = nn.BatchNorm2d(mid)
        self.c3 = nn.Conv2d(mid, cout, 1, bias=False)
        self.b3 = nn.BatchNorm2d(cout)
        self.down = None
        if stride != 1 or cin != cout:
            self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.down is None else self.down(x)
        x = F.relu(self.b1(self.c1(x)))
        x = F.relu(self.b2(self.c2(x)))
        return F.relu(self.b3(self.c3(x)) + idt)


class ResNet50FPN(nn.Module):
    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(3, 2, 1))
        cfg, cin, layers = [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)], 64, []
        for mid, n, s in cfg:
            blocks = []
            for i in range(n):
                blocks.append(Bottleneck(cin, mid, s if i == 0 else 1))
                cin = mid * 4
            layers.append(nn.Sequential(*blocks))
        self.layers = nn.ModuleList(layers)
        self.lat = nn.ModuleList([nn.Conv2d(c, 256, 1) for c in (256, 512, 1024, 2048)])
        self.out = nn.ModuleList([nn.Conv2d(256, 256, 3, padding=1) for _ in range(4)])

    def forward(self, img):
        x = self.stem(img)
        cs = []
        for l in self.layers:
            x = l(x)
            cs.append(x)
        lats = [l(c) for l, c in zip(self.lat, cs)]
        for i in range(3, 0, -1):
            lats[i - 1] = lats[i - 1] + F.interpolate(lats[i], size=lats[i - 1].shape[-2:], mode='nearest')
        outs = [o(l) for o, l in zip(self.out, lats)]
        outs.append(F.max_pool2d(outs[-1], 1, stride=2))
        return outs
