"""How long does the side stream (weight gradients) run on after the main stream's last backward kernel?  (MDCV_TAIL_PROBE=1)"""
import os, sys, tempfile
os.environ["MDCV_TAIL_PROBE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mdcv.yolo.models import Darknet
from mdcv.optim import FusedAdam
tmp = tempfile.mkdtemp(); cfg = bench.write_yolo_cfg(tmp); os.chdir(tmp)
torch.manual_seed(0)
net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision="bf16").cuda().train()
opt = FusedAdam(net, lr=1e-3)
g = torch.Generator().manual_seed(1)
x = torch.rand(32, 3, 416, 416, generator=g).cuda(); tg = bench.synth_targets(32, 16, g).cuda()
for i in range(25):
    opt.zero_grad(); out = net(x, tg); out[0].sum().backward(); opt.step()
torch.cuda.synchronize()
plan = [p for p in net._plans.values() if p.has_bwd][0]
t = [a.elapsed_time(b) for a, b in plan.tail_events[8:]]
print("side-stream tail after main backward (ms): mean %.3f min %.3f max %.3f" % (sum(t) / len(t), min(t), max(t)))
