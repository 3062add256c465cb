"""-m gpu: bf16 (production) against the fp32-kernel mode over training, at the BASELINE model sizes ("loss-curve
tracking over >= 50 steps vs fp32 mode").  Same seeds, same batch sequence, FusedAdam; the only difference between the two runs is the
storage / MFMA input precision of activations and packed weights (statistics, heads, master weights and optimizer are fp32 in both).

Training from a random initialisation with Adam is chaotic in its first steps (the YOLOv3 loss goes 71 -> 370 -> 104 in three steps at the
reference's lr 1e-3; RektNet's reference lr is 0.1): two fp32 runs that differ by ONE rounding of the initial weights drift apart by
2 % (YOLOv3, median over 60 steps) to 36 % (RektNet at lr 1e-2).  "Tracks" is therefore judged against that control: a third run in the
fp32-kernel mode whose initial weights were rounded to bf16 once.  The bf16 run may deviate from the fp32 run by at most 2.5x what the
control does (with small absolute floors), its very first loss (same weights, same batch) must be within 1e-2, and all runs must learn
(`scripts/fidelity_curves.py` prints the three curves)."""
import contextlib
import io
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _curve_stats(a, b):
    rel = np.abs(a - b) / np.abs(a)
    k = 10
    ma = np.convolve(a, np.ones(k) / k, "valid")
    mb = np.convolve(b, np.ones(k) / k, "valid")
    return dict(first=float(rel[0]), head=float(rel[:5].max()), median=float(np.median(rel)), max=float(rel.max()),
                smooth_max=float((np.abs(ma - mb) / ma).max()), fp32=(float(a[0]), float(a[-k:].mean())), bf16=(float(b[0]), float(b[-k:].mean())))


def _round_weights_to_bf16_once(net):
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(p.to(torch.bfloat16).float())


def _judge(curves, learn):
    st, ctl = _curve_stats(curves["fp32"], curves["bf16"]), _curve_stats(curves["fp32"], curves["control"])
    msg = dict(bf16=st, control=ctl)
    for k in ("fp32", "bf16"):
        assert st[k][1] < learn * st[k][0], msg                                        # both learn
    assert st["first"] < 1e-2, msg                                                      # same weights, same batch: one step of bf16 arithmetic
    assert st["median"] <= max(2.5 * ctl["median"], 0.03), msg
    assert st["smooth_max"] <= max(2.5 * ctl["smooth_max"], 0.10), msg


RUNS = (("fp32", "fp32", False), ("bf16", "bf16", False), ("control", "fp32", True))


def test_yolo_baseline_416_b8_bf16_tracks_fp32_over_60_steps(tmp_path):
    """CVC-YOLOv3 yolo_baseline 416x416 classes=80, 8 images per step from the on-device synthetic cone stream (8 distinct batches,
    cycled), 60 optimizer steps (Adam 1e-3, train.py:180-187)."""
    import bench
    from mdcv.yolo.models import Darknet
    from mdcv.optim import FusedAdam
    from mdcv.data import SyntheticCones
    cfg = bench.write_yolo_cfg(str(tmp_path))
    data = SyntheticCones(8, 416, 416, 16, 1, batches=8, seed=21, device="cuda")
    batches = [data.batch(i) for i in range(8)]
    curves = {}
    for tag, prec, control in RUNS:
        cwd = os.getcwd()
        os.chdir(tmp_path)
        try:
            torch.manual_seed(0)
            net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision=prec).cuda().train()
        finally:
            os.chdir(cwd)
        if control:
            _round_weights_to_bf16_once(net)
        opt = FusedAdam(net, lr=1e-3)
        ls = []
        for i in range(60):
            _, x, tg = batches[i % 8]
            opt.zero_grad()
            out = net(x, tg)
            out[0].sum().backward()
            opt.step()
            ls.append(out[0].detach())
        curves[tag] = torch.stack(ls).cpu().double().numpy()
        del net, opt
        torch.cuda.empty_cache()
    _judge(curves, learn=0.25)


def test_rektnet_b64_bf16_tracks_fp32_over_60_steps():
    """RektNet KeypointNet 80x80, 64 crops per step from the synthetic crop stream (8 distinct batches, cycled), l1_softargmax + geometric
    loss, Adam (train_eval.py:263) at lr 1e-2, 60 steps."""
    from mdcv.rektnet.keypoint_net import KeypointNet
    from mdcv.rektnet.cross_ratio_loss import CrossRatioLoss
    from mdcv.optim import FusedAdam
    from mdcv.data import SyntheticConeCrops
    with contextlib.redirect_stdout(io.StringIO()):
        crit = CrossRatioLoss("l1_softargmax", True, 0.05, 0.05)
    data = SyntheticConeCrops(64, 80, batches=8, seed=5, device="cuda")
    batches = [data.batch(i) for i in range(8)]
    curves = {}
    for tag, prec, control in RUNS:
        torch.manual_seed(0)
        net = KeypointNet(7, (80, 80), precision=prec).cuda().train()
        if control:
            _round_weights_to_bf16_once(net)
        opt = FusedAdam(net, lr=1e-2)
        ls = []
        for i in range(60):
            x, thm, tp = batches[i % 8][:3]
            opt.zero_grad()
            hm, pts = net(x)
            loss = crit(hm, pts, thm, tp)[2]
            loss.backward()
            opt.step()
            ls.append(loss.detach())
        curves[tag] = torch.stack(ls).cpu().double().numpy()
    _judge(curves, learn=0.5)
