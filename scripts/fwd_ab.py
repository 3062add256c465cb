"""The forward launch list of the YOLOv3 training plan alone (BASELINE config 3), with the forward statistics as partial rows + finalize launches
(Y0) and through exact accumulators (Y1): list time back to back, and the per-kernel time of one serial instrumented pass.   usage: fwd_ab.py"""
import os, sys, tempfile, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdcv import engine
from mdcv.yolo.models import Darknet
from mdcv.optim import FusedAdam
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
cfg = bench.write_yolo_cfg(tmp)
os.chdir(tmp)
g = torch.Generator().manual_seed(1000)
x, tg = torch.rand(32, 3, 416, 416, generator=g).to(dev), bench.synth_targets(32, 16, g).to(dev)
plans = {}
for mode in (0, 1):
    engine.Plan.stats_xacc = bool(mode)
    torch.manual_seed(0)
    net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True).to(dev).train()
    opt = FusedAdam(net, lr=1e-3)
    for _ in range(3):
        opt.zero_grad(); net(x, tg)[0].sum().backward(); opt.step()
    torch.cuda.synchronize()
    plans[mode] = ([p for p in net._plans.values() if p.has_bwd][0], net, opt)
for rnd in range(3):
    for mode in (0, 1):
        plan = plans[mode][0]
        for _ in range(5): plan.run(plan.fwd)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30): plan.run(plan.fwd)
        b.record(); torch.cuda.synchronize()
        print("round %d  Y%d  forward list %.3f ms  (%d launches, %d layers through accumulators)" % (rnd, mode, a.elapsed_time(b) / 30, len(plan.fwd), getattr(plan, "stats_xfolded", 0)), flush=True)
for mode in (0, 1):
    plan = plans[mode][0]
    recs = engine.run_timed(plan, plan.fwd, kernels=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in recs:
        for kn, ms in r[3]:
            k = kn[:100]
            agg[k][0] += 1; agg[k][1] += ms * 1e3
    tot = sum(v[1] for v in agg.values())
    print("Y%d serial kernel time %.1f us" % (mode, tot))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
        print("   %6.1f us  x%3d  %s" % (v[1], v[0], k))
