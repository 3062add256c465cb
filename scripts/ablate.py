#!/usr/bin/env python3
"""Which launches is a training step's time sensitive to?  Re-times the YOLOv3 (or RektNet) step with one family of launches
dropped from the plan's launch lists.  Results are WRONG by construction (timing only), which is why this lives in a script that
edits the lists of a plan it built itself and not behind an environment variable of the product.
    python scripts/ablate.py [yolo|rektnet]
The difference to the full step is what removing / hiding that family could buy at most."""
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

FAMILIES = ["", "conv2d_wgrad", "mdcv_bn_act_fwd", "mdcv_bn_act_bwd_apply", "mdcv_bn_stats_finalize,mdcv_bn_bwd_finalize_rows",
            "mdcv_bn_act_bwd_reduce_finalize", "mdcv_pack_weights_batched", "mdcv_conv2d_dgrad_bnsums",
            "conv2d_wgrad,mdcv_bn_act_fwd,mdcv_bn_act_bwd_apply,mdcv_bn_stats_finalize,mdcv_bn_bwd_finalize_rows,mdcv_bn_act_bwd_reduce_finalize", ""]


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "yolo"
    from mdcv.optim import FusedAdam
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1000)
    if wl == "yolo":
        from mdcv.yolo.models import Darknet
        tmp = tempfile.mkdtemp()
        cfg = bench.write_yolo_cfg(tmp)
        os.chdir(tmp)
        torch.manual_seed(0)
        net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True).to(dev).train()
        B = 32
        x, tg = torch.rand(B, 3, 416, 416, generator=g).to(dev), bench.synth_targets(B, 16, g).to(dev)
        opt = FusedAdam(net, lr=1e-3)

        def step():
            opt.zero_grad()
            net(x, tg)[0].sum().backward()
            opt.step()
    else:
        from mdcv.rektnet.keypoint_net import KeypointNet
        from mdcv.rektnet.cross_ratio_loss import CrossRatioLoss
        torch.manual_seed(0)
        net = KeypointNet(7, (80, 80)).to(dev).train()
        crit = CrossRatioLoss("l1_softargmax", True, 0.05, 0.05)
        B = 256
        x, tp = torch.rand(B, 3, 80, 80, generator=g).to(dev), (torch.rand(B, 7, 2, generator=g) * (79 / 80)).to(dev)
        opt = FusedAdam(net, lr=0.1)

        def step():
            opt.zero_grad()
            hm, pts = net(x)
            crit(hm, pts, None, tp)[2].backward()
            opt.step()
    step()
    plan = [p for p in net._plans.values() if p.has_bwd][0]
    fwd0, bwd0 = list(plan.fwd), list(plan.bwd)
    for fam in FAMILIES:
        drop = set(v for v in fam.split(",") if v)
        plan.fwd = [(f, a) for f, a in fwd0 if getattr(f, "__name__", "") not in drop]
        plan.bwd = [(f, a) for f, a in bwd0 if getattr(f, "__name__", "") not in drop]
        for _ in range(6):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print("%-60s %8.1f img/s %7.3f ms" % (fam or "full", B / dt, dt * 1e3), flush=True)


if __name__ == "__main__":
    main()
