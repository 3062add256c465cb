"""Same-box A/B of which data gradients carry the fused BatchNorm-backward sums (engine.Plan.fuse_skip bit mask, see Plan._fuse_pays):
one process, a fresh model per mask, 3 rounds.  usage: ab_fuse.py 13,141,..."""
import os, sys, tempfile, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdcv import engine
from mdcv.yolo.models import Darknet
from mdcv.optim import FusedAdam

masks = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "13,141").split(",")]
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
cfg = bench.write_yolo_cfg(tmp)
g = torch.Generator().manual_seed(1000)
x, tg = torch.rand(32, 3, 416, 416, generator=g).to(dev), bench.synth_targets(32, 16, g).to(dev)
for rnd in range(3):
    for m in masks:
        engine.Plan.fuse_skip = m
        os.chdir(tmp)
        torch.manual_seed(0)
        net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True).to(dev).train()
        opt = FusedAdam(net, lr=1e-3)

        def step():
            opt.zero_grad()
            net(x, tg)[0].sum().backward()
            opt.step()
        for _ in range(8): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30): step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 30
        plan = [p for p in net._plans.values() if p.has_bwd][0]
        print("round %d  fuse_skip %4d  fused layers %2d  %.3f ms  %.1f img/s" % (rnd, m, plan.fused_bn, dt * 1e3, 32 / dt), flush=True)
        del net, opt, plan
        torch.cuda.empty_cache()
