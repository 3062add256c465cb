import os, sys, contextlib, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mdcv.rektnet.keypoint_net import KeypointNet
from mdcv.rektnet.cross_ratio_loss import CrossRatioLoss
from mdcv.optim import FusedAdam
from mdcv.data import SyntheticConeCrops
for f32 in (True, False, True, False):
    KeypointNet.f32_logits = f32
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        crit = CrossRatioLoss("l1_softargmax", True, 0.05, 0.05)
    kp = KeypointNet(7, (80, 80), precision="bf16").cuda().train()
    opt = FusedAdam(kp, lr=1e-2)
    losses = []
    for x, hm_t, pts_t, _, _ in SyntheticConeCrops(256, 80, batches=300, seed=5):
        opt.zero_grad()
        hm, pts = kp(x)
        loss = crit(hm, pts, hm_t, pts_t)[2]
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    ls = torch.stack(losses).flatten().cpu()
    print("f32_logits", f32, "loss0 %.4f  @50 %.4f @100 %.4f @150 %.4f @200 %.4f @300 %.4f" % (ls[0], ls[40:50].mean(), ls[90:100].mean(), ls[140:150].mean(), ls[190:200].mean(), ls[290:300].mean()), flush=True)
