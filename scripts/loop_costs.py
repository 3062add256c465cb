#!/usr/bin/env python3
"""Where the gap between bench.py's headline step and the reference's UNCHANGED training-loop statements comes from: the same model
and batch (YOLOv3 416^2, batch 32, bf16) stepped with the loop's three host-side habits switched on one at a time --
  stock torch.optim.Adam over model.parameters() (CVC-YOLOv3/train.py:180-187) instead of mdcv.optim.FusedAdam,
  the batch copied from pinned host memory every step (train.py:60-61),
  the sixteen blocking .item() / .to('cpu') reads per step (train.py:63, 75, 85-89).
usage: loop_costs.py [steps]"""
import os, sys, tempfile, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdcv.yolo.models import Darknet
from mdcv.optim import FusedAdam

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
cfg = bench.write_yolo_cfg(tmp)
os.chdir(tmp)
g = torch.Generator().manual_seed(1000)
xh, th = torch.rand(32, 3, 416, 416, generator=g).pin_memory(), bench.synth_targets(32, 16, g).pin_memory()
xd, td = xh.to(dev), th.to(dev)


def run(name, optim, h2d, items, fused_kw=None):
    torch.manual_seed(0)
    net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True).to(dev).train()
    if optim == "fused":
        opt = FusedAdam(net, lr=1e-3)
    else:
        opt = torch.optim.Adam(filter(lambda p: p.requires_grad, net.parameters()), lr=1e-3, weight_decay=0.0, **(fused_kw or {}))

    def step():
        imgs, targets = (xh.to(dev, non_blocking=True), th.to(dev, non_blocking=True)) if h2d else (xd, td)
        if items:
            n = ((targets[:, :, 1:5] > 0).sum(dim=2) > 1).sum().item() + 1e-12
        opt.zero_grad()
        losses = net(imgs, targets)
        losses[0].sum().backward()
        opt.step()
        if items:
            logged = [l.sum().to("cpu").item() for l in losses]
            tot = losses[0].item() / n
            t0 = losses[0].item()
            pct = [l.item() / t0 * 100 for l in losses[1:]]
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print("%-78s %7.3f ms  %6.0f img/s" % (name, ms, 32 / ms * 1e3), flush=True)
    del net, opt
    torch.cuda.empty_cache()
    return ms


base = run("FusedAdam, resident batch, no host reads (bench.py's headline step)", "fused", False, False)
a = run("stock torch.optim.Adam (foreach), resident batch, no host reads", "stock", False, False)
a2 = run("stock torch.optim.Adam(fused=True), resident batch, no host reads", "stock", False, False, {"fused": True})
b = run("FusedAdam, batch from pinned host memory every step", "fused", True, False)
c = run("FusedAdam, resident batch, the loop's sixteen .item() reads", "fused", False, True)
d = run("stock Adam + host batch + .item() reads (the unchanged loop)", "stock", True, True)
e = run("FusedAdam + host batch + .item() reads", "fused", True, True)
print("optimizer %+.3f ms (fused=True: %+.3f) ; host batch %+.3f ms ; host reads %+.3f ms ; all three %+.3f ms" % (a - base, a2 - base, b - base, c - base, d - base))
