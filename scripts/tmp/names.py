import os, sys, tempfile, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from mdcv.yolo.models import Darknet
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
g = torch.Generator().manual_seed(1000)
x, tg = torch.rand(32, 3, 416, 416, generator=g).to(dev), bench.synth_targets(32, 16, g).to(dev)
cfg = bench.write_yolo_cfg(tmp); os.chdir(tmp)
net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True).to(dev).train()
net(x, tg)[0].sum().backward()
for plan in net._plans.values():
    for nm in ("pre", "fwd", "bwd"):
        c = collections.Counter(getattr(f, "__name__", str(f)) for f, a in getattr(plan, nm))
        print(nm, dict(c))
