#!/usr/bin/env python3
"""How far ahead of the GPU is the host's launch loop?  The YOLOv3 step (416^2, batch 32, bf16, FusedAdam) with an artificial host delay (busy wait)
inserted at one place per step: while the step time does not move, the host had at least that much slack THERE; where it moves one for one, the GPU
was already waiting for the host.   usage: host_slack.py [steps]"""
import os, sys, tempfile, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdcv.yolo.models import Darknet
from mdcv.optim import FusedAdam

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
cfg = bench.write_yolo_cfg(tmp)
os.chdir(tmp)
g = torch.Generator().manual_seed(1000)
x, tg = torch.rand(32, 3, 416, 416, generator=g).to(dev), bench.synth_targets(32, 16, g).to(dev)
torch.manual_seed(0)
net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True).to(dev).train()
opt = FusedAdam(net, lr=1e-3)


def spin(us):
    t = time.perf_counter() + us * 1e-6
    while time.perf_counter() < t:
        pass


def run(where, us):
    def step():
        if where == "before_forward": spin(us)
        opt.zero_grad()
        out = net(x, tg)
        if where == "before_backward": spin(us)
        loss = out[0].sum()
        loss.backward()
        if where == "before_optimizer": spin(us)
        opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


base = run("none", 0)
print("no delay: %.3f ms" % base, flush=True)
for where in ("before_forward", "before_backward", "before_optimizer"):
    print(where, " ".join("%4d us: %+.3f ms" % (us, run(where, us) - base) for us in (100, 300, 1000, 3000)), flush=True)
print("no delay again: %.3f ms" % run("none", 0))
# host time of the phases (GPU drained first, so these are pure launch-loop times)
for name in ("forward", "backward", "optimizer"):
    ts = []
    for _ in range(5):
        opt.zero_grad(); torch.cuda.synchronize()
        t0 = time.perf_counter(); out = net(x, tg); t1 = time.perf_counter()
        loss = out[0].sum(); torch.cuda.synchronize()
        t2 = time.perf_counter(); loss.backward(); t3 = time.perf_counter(); torch.cuda.synchronize()
        t4 = time.perf_counter(); opt.step(); t5 = time.perf_counter(); torch.cuda.synchronize()
        ts.append((t1 - t0, t3 - t2, t5 - t4))
print("host launch-loop time per phase (ms, GPU idle at entry): forward %.2f backward %.2f optimizer %.2f" % tuple(1e3 * min(t[i] for t in ts) for i in range(3)))
