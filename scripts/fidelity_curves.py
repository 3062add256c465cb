"""Loss curves of the fp32-kernel mode, the bf16 mode and an fp32 run whose initial weights were perturbed by one bf16 rounding, on the
batches of tests/test_gpu_fidelity.py: how far apart two trajectories drift from rounding-sized perturbations alone."""
import contextlib, io, os, sys, tempfile
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdcv.yolo.models import Darknet
from mdcv.rektnet.keypoint_net import KeypointNet
from mdcv.rektnet.cross_ratio_loss import CrossRatioLoss
from mdcv.optim import FusedAdam
from mdcv.data import SyntheticCones, SyntheticConeCrops

which = sys.argv[1] if len(sys.argv) > 1 else "both"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
lr_y = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
lr_r = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-2


def perturb(net):
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(p.to(torch.bfloat16).float())


def show(name, curves):
    a = curves["fp32"]
    for k, v in curves.items():
        rel = np.abs(v - a) / np.abs(a)
        print(name, k, "first5", np.round(v[:5], 4).tolist(), "last", round(float(v[-10:].mean()), 4), "rel first", f"{rel[0]:.2e}", "head", f"{rel[:5].max():.3f}",
              "median", f"{np.median(rel):.3f}", "max", f"{rel.max():.3f}")
    print(name, "curve fp32", np.round(a, 3).tolist())
    print(name, "curve bf16", np.round(curves["bf16"], 3).tolist())


if which in ("both", "yolo"):
    tmp = tempfile.mkdtemp()
    cfg = bench.write_yolo_cfg(tmp)
    data = SyntheticCones(8, 416, 416, 16, 1, batches=8, seed=21, device="cuda")
    batches = [data.batch(i) for i in range(8)]
    curves = {}
    for tag, prec, pert in (("fp32", "fp32", False), ("bf16", "bf16", False), ("fp32+w_bf16_rounded", "fp32", True)):
        os.chdir(tmp)
        torch.manual_seed(0)
        net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision=prec).cuda().train()
        if pert:
            perturb(net)
        opt = FusedAdam(net, lr=lr_y)
        ls = []
        for i in range(steps):
            _, x, tg = batches[i % 8]
            opt.zero_grad()
            out = net(x, tg)
            out[0].sum().backward()
            opt.step()
            ls.append(out[0].detach())
        curves[tag] = torch.stack(ls).cpu().double().numpy()
        del net, opt
        torch.cuda.empty_cache()
    show("yolo", curves)
if which in ("both", "rektnet"):
    with contextlib.redirect_stdout(io.StringIO()):
        crit = CrossRatioLoss("l1_softargmax", True, 0.05, 0.05)
    data = SyntheticConeCrops(64, 80, batches=8, seed=5, device="cuda")
    batches = [data.batch(i) for i in range(8)]
    curves = {}
    for tag, prec, pert in (("fp32", "fp32", False), ("bf16", "bf16", False), ("fp32+w_bf16_rounded", "fp32", True)):
        torch.manual_seed(0)
        net = KeypointNet(7, (80, 80), precision=prec).cuda().train()
        if pert:
            perturb(net)
        opt = FusedAdam(net, lr=lr_r)
        ls = []
        for i in range(steps):
            x, thm, tp = batches[i % 8][:3]
            opt.zero_grad()
            hm, pts = net(x)
            loss = crit(hm, pts, thm, tp)[2]
            loss.backward()
            opt.step()
            ls.append(loss.detach())
        curves[tag] = torch.stack(ls).cpu().double().numpy()
    show("rektnet", curves)
