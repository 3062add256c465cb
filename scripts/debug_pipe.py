"""Debug: FusedAdam(pipeline) vs plain on the full YOLOv3: first diverging parameter per step."""
import os, sys, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdcv.yolo.models import Darknet
from mdcv.optim import FusedAdam
tmp = tempfile.mkdtemp(); cfg = bench.write_yolo_cfg(tmp)
def run(pipe, steps=int(os.environ.get("DBG_STEPS", "6")), B=int(os.environ.get("DBG_B", "8"))):
    cwd = os.getcwd(); os.chdir(tmp)
    torch.manual_seed(0)
    net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision="bf16")
    os.chdir(cwd)
    net = net.cuda().train()
    opt = FusedAdam(net, lr=1e-3, pipeline=pipe)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(B, 3, 416, 416, generator=g).cuda(); tg = bench.synth_targets(B, 16, g).cuda()
    snaps = []
    for i in range(steps):
        opt.zero_grad()
        out = net(x, tg); out[0].sum().backward(); opt.step()
        if os.environ.get("DBG_SYNC", "0") == "1":
            torch.cuda.synchronize(); net._param_sync(); torch.cuda.synchronize()
        snaps.append(out[0].detach().clone())
    torch.cuda.synchronize()
    snaps = [(float(s), None, None) for s in snaps]
    snaps[-1] = (snaps[-1][0], net.flat_parameters()[0].clone(), net.flat_parameters()[1].clone())
    plan = net._last_train_plan
    return snaps, plan, net
modes = [m == '1' for m in os.environ.get('DBG_MODES', '0,1').split(',')]
a, pa, na = run(modes[0]); b, pb, nb = run(modes[1])
print("groups", pb.param_groups if pb.param_groups else pa.param_groups)
names = [(n, na._goff[id(p)]) for n, p in na.named_parameters()]
for i, ((la, pa_, ga), (lb, pb_, gb)) in enumerate(zip(a, b)):
    if pa_ is None:
        print("step", i, "loss", la, lb); continue
    dp = (pa_ != pb_).nonzero().flatten(); dg = (ga != gb).nonzero().flatten()
    print("step", i, "loss", la, lb, "param diffs", dp.numel(), "grad diffs", dg.numel())
    if dp.numel():
        first = int(dp[0]); last = int(dp[-1])
        fn = [n for n, (o, c) in names if o <= first < o + c]; ln = [n for n, (o, c) in names if o <= last < o + c]
        print("   first diff at", first, fn, "last", last, ln)
