"""How many 1x1 blocks of yolo_baseline 416^2 B=32 take the fused forms, and the per-kernel time of one serial instrumented step.
usage: pw_diag.py [0|1]   (engine.Plan.pw_fuse)"""
import os, sys, tempfile, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdcv import engine
from mdcv.yolo.models import Darknet
from mdcv.optim import FusedAdam
if len(sys.argv) > 1:
    engine.Plan.pw_fuse = bool(int(sys.argv[1]))
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
cfg = bench.write_yolo_cfg(tmp)
os.chdir(tmp)
g = torch.Generator().manual_seed(1000)
x, tg = torch.rand(32, 3, 416, 416, generator=g).to(dev), bench.synth_targets(32, 16, g).to(dev)
torch.manual_seed(0)
net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True).to(dev).train()
opt = FusedAdam(net, lr=1e-3)
for _ in range(3):
    opt.zero_grad(); net(x, tg)[0].sum().backward(); opt.step()
torch.cuda.synchronize()
plan = [p for p in net._plans.values() if p.has_bwd][0]
print("pw_fwd", getattr(plan, "pw_fwd_count", 0), "pw_bwd", getattr(plan, "pw_bwd_count", 0), "fused_bn", plan.fused_bn, "fwd launches", len(plan.fwd), "bwd", len(plan.bwd))
for name, lst in (("fwd", plan.fwd), ("bwd", plan.bwd)):
    recs = engine.run_timed(plan, lst, kernels=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in recs:
        for kn, ms in r[3]:
            k = kn[:90]
            agg[k][0] += 1; agg[k][1] += ms * 1e3
    tot = sum(v[1] for v in agg.values())
    print(name, "serial kernel time %.1f us" % tot)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   %6.1f us  x%3d  %s" % (v[1], v[0], k))
