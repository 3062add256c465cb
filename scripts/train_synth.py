"""Sustained training on the on-device synthetic cone data: YOLOv3 (classes=1, 416x416, batch 32) and RektNet (batch 256), bf16.
Prints the loss every 25 steps.  usage: train_synth.py [steps]"""
import contextlib, io, os, sys, tempfile, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdcv.yolo.models import Darknet
from mdcv.rektnet.keypoint_net import KeypointNet
from mdcv.rektnet.cross_ratio_loss import CrossRatioLoss
from mdcv.optim import FusedAdam
from mdcv.data.synth import SyntheticCones, SyntheticConeCrops
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tmp = tempfile.mkdtemp(); cfg = bench.write_yolo_cfg(tmp, classes=1)
cwd = os.getcwd(); os.chdir(tmp); torch.manual_seed(0)
net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision=os.environ.get("TS_PREC", "bf16")).cuda().train(); os.chdir(cwd)
opt = FusedAdam(net, lr=1e-3)
data = SyntheticCones(32, 416, 416, 16, 1, batches=steps, seed=3)
t0 = time.perf_counter(); hist = []
for i, (_, x, tg) in enumerate(data):
    opt.zero_grad(); out = net(x, tg); out[0].sum().backward(); opt.step()
    if i % 25 == 0 or i == steps - 1:
        hist.append(float(out[0])); print("yolo step %4d loss %.4f parts %s" % (i, hist[-1], ["%.3f" % float(v) for v in out[1:]]), flush=True)
torch.cuda.synchronize(); print("yolo: %d steps in %.1f s (%.0f img/s incl. data generation)" % (steps, time.perf_counter() - t0, 32 * steps / (time.perf_counter() - t0)))
assert all(h == h for h in hist) and hist[-1] < 0.5 * hist[0], hist
from mdcv.yolo.validate import validate                     # the reference's train.py runs validate() on its validation loader
val = SyntheticCones(32, 416, 416, 16, 1, batches=8, seed=99)
m = validate(dataloader=val, model=net, device=torch.device("cuda"))
print("validate on 256 held-out synthetic images: mAP %.3f recall %.3f precision %.3f  (%.2f ms/img)" % (m[0], m[1], m[2], 1e3 * m[3]))
net.train()
with contextlib.redirect_stdout(io.StringIO()):
    crit = CrossRatioLoss("l1_softargmax", True, 0.05, 0.05)
kp = KeypointNet(7, (80, 80), precision="bf16").cuda().train()
opt = FusedAdam(kp, lr=1e-2)
data = SyntheticConeCrops(256, 80, batches=steps, seed=5)
t0 = time.perf_counter(); hist = []
for i, (x, hm_t, pts_t, _, _) in enumerate(data):
    opt.zero_grad(); hm, pts = kp(x); loss = crit(hm, pts, hm_t, pts_t)[2]; loss.backward(); opt.step()
    if i % 25 == 0 or i == steps - 1:
        hist.append(float(loss)); print("rektnet step %4d loss %.4f" % (i, hist[-1]), flush=True)
torch.cuda.synchronize(); print("rektnet: %d steps in %.1f s (%.0f img/s incl. data generation)" % (steps, time.perf_counter() - t0, 256 * steps / (time.perf_counter() - t0)))
assert all(h == h for h in hist) and hist[-1] < 0.5 * hist[0], hist
