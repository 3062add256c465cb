"""Run-to-run reproducibility over MANY steps: two YOLOv3 models from the same seed take the same batches step by step; after every step the
flat parameter buffers must be bit-identical.  Reports the first step at which they differ (a race shows up as a rare divergence that
training then amplifies).  usage: repro_probe.py [steps=200] [fork_scope=1] [batch=32] [yolo|rektnet]"""
import os, sys, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdcv.yolo import models as M
from mdcv.yolo.models import Darknet
from mdcv.optim import FusedAdam
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
M._NetPlan.fork_device_scope = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
if os.environ.get("PROBE_NOFUSE"):
    from mdcv import engine
    engine.Plan.fuse_bn = False
tmp = tempfile.mkdtemp(); cfg = bench.write_yolo_cfg(tmp)
os.chdir(tmp)
from mdcv import engine as _eng
_orig_new_act = _eng.Plan.new_act


def _new_act(self, *a, **k):
    act = _orig_new_act(self, *a, **k)
    self.__dict__.setdefault("_acts", []).append(act)
    return act


_eng.Plan.new_act = _new_act
NET = sys.argv[4] if len(sys.argv) > 4 else "yolo"
nets, opts = [], []
g = torch.Generator().manual_seed(21)
if NET == "yolo":
    for i in range(2):
        torch.manual_seed(0)
        n = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision=os.environ.get("PROBE_PREC", "bf16")).cuda().train()
        nets.append(n); opts.append(FusedAdam(n, lr=1e-3))
    xs = [torch.rand(B, 3, 416, 416, generator=g).cuda() for _ in range(4)]
    tg = [bench.synth_targets(B, 16, g).cuda() for _ in range(4)]

    def one(n, o, s):
        o.zero_grad()
        out = n(xs[s % 4], tg[s % 4]); out[0].sum().backward(); o.step()
        return float(out[0].detach().sum())
else:
    import contextlib
    from mdcv.rektnet.keypoint_net import KeypointNet
    from mdcv.rektnet.cross_ratio_loss import CrossRatioLoss
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    for i in range(2):
        torch.manual_seed(0)
        n = KeypointNet(7, (80, 80), precision=os.environ.get("PROBE_PREC", "bf16")).cuda().train()
        nets.append(n); opts.append(FusedAdam(n, lr=1e-2))
    with contextlib.redirect_stdout(sys.stderr):
        crit = CrossRatioLoss("l1_softargmax", True, 0.05, 0.05)
    xs = [torch.rand(B, 3, 80, 80, generator=g).cuda() for _ in range(4)]
    tg = [(torch.rand(B, 7, 2, generator=g) * (79 / 80)).cuda() for _ in range(4)]

    def one(n, o, s):
        o.zero_grad()
        hm, pts = n(xs[s % 4])
        loss = crit(hm, pts, None, tg[s % 4])[2]
        loss.backward(); o.step()
        return float(loss.detach())
bad = 0
for s in range(steps):
    losses = []
    for n, o in zip(nets, opts):
        losses.append(one(n, o, s))
    a, b = nets[0].flat_parameters()[0], nets[1].flat_parameters()[0]
    if not torch.equal(a, b):
        d = (a != b)
        ga, gb = nets[0].flat_parameters()[1], nets[1].flat_parameters()[1]
        dg = (ga != gb)
        idx = torch.nonzero(dg).flatten()
        print("step %d: DIFFER params %d elements, grads %d elements (first grad index %s, last %s) losses %r" %
              (s, int(d.sum()), int(dg.sum()), idx[:1].tolist(), idx[-1:].tolist(), losses), flush=True)
        bad += 1
        pa, pb = nets[0]._last_train_plan, nets[1]._last_train_plan
        rows = []
        for k, (x, y) in enumerate(zip(pa._acts, pb._acts)):
            dx = x.dense().float() != y.dense().float()
            nd = int(dx.sum())
            if nd:
                idx = torch.nonzero(dx.reshape(-1, x.C))
                rows.append((nd, k, (x.B, x.H, x.W, x.C), int(idx[:, 0].min()), int(idx[:, 0].max()), int(idx[:, 1].min()), int(idx[:, 1].max()),
                             len(torch.unique(idx[:, 0])), len(torch.unique(idx[:, 1]))))
        rows.sort()
        for r in rows[:6]:
            print("    act #%d %s: %d elements differ; pixel rows %d..%d (%d distinct), channels %d..%d (%d distinct)" % (r[1], r[2], r[0], r[3], r[4], r[7], r[5], r[6], r[8]), flush=True)
        print("    (%d activation / gradient buffers differ in all, of %d)" % (len(rows), len(pa._acts)), flush=True)
        nets[1].load_state_dict(nets[0].state_dict())          # resynchronise and keep looking
        opts[1].load_state_dict(opts[0].state_dict())
print("steps %d, divergences %d, fork_device_scope %d" % (steps, bad, M._NetPlan.fork_device_scope))
