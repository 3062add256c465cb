"""Where the bf16 mode loses gradient direction: per BatchNorm layer of yolo_baseline (fp32-kernel mode = truth), the ratio between the STORED
activation gradient g and what survives BatchNorm backward, and the error a bf16 rounding of g leaves in dy -- plain, and with g stored
relative to a per-channel offset (its mean).  usage: grad_precision_probe.py [batch=8]"""
import os, sys, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdcv import engine
from mdcv.yolo.models import Darknet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rec = []
orig = engine.Plan.emit_bn_act_bwd


def spy(self, dout, y1, bs1, act, slope, y2=None, bs2=None):
    r = orig(self, dout, y1, bs1, act, slope, y2, bs2)
    if y2 is None:
        rec.append((dout, y1, bs1, act, slope, r))
    return r


engine.Plan.emit_bn_act_bwd = spy
tmp = tempfile.mkdtemp(); cfg = bench.write_yolo_cfg(tmp)
cwd = os.getcwd(); os.chdir(tmp)
torch.manual_seed(3)
net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision="fp32").cuda().train()
os.chdir(cwd)
g = torch.Generator().manual_seed(21)
x = torch.rand(B, 3, 416, 416, generator=g).cuda(); tg = bench.synth_targets(B, 16, g).cuda()
out = net(x, tg); out[0].sum().backward(); torch.cuda.synchronize()
print("loss", float(out[0]), "BN layers recorded (backward order):", len(rec))
print("%3s %5s %5s | %9s %9s | %8s %8s %8s" % ("#", "M/1k", "C", "|g|/|dy'|", "|gbar|/|g|", "err bf16", "err ctr", "err ctr+y"))
for i, (dout, y, bs, act, slope, dy) in enumerate(rec):
    gg = dout.dense().reshape(-1, dout.C).double()
    yy = y.dense().reshape(-1, y.C).double()
    C = bs.C
    gg, yy = gg[:, :C], yy[:, :C]
    sc, sh, mu, istd = (t[:C].double() for t in (bs.scale, bs.shift, bs.mean, bs.invstd))
    pre = yy * sc + sh
    l = torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, slope)) if act else torch.ones_like(pre)
    yh = (yy - mu) * istd

    def proj(gx):
        dz = gx * l
        return dz - dz.mean(0) - yh * (dz * yh).mean(0)
    r0 = proj(gg)
    gb = gg.mean(0)
    e_plain = proj(gg.float().bfloat16().double()) - r0
    ctr = (gg - gb).float().bfloat16().double() + gb
    e_ctr = proj(ctr) - r0
    # offset affine in yhat: g ~ a + b*yhat per channel (least squares), stored relative to it
    b = (gg * yh).mean(0) / (yh * yh).mean(0).clamp_min(1e-30)
    fit = gb + b * yh
    ctr2 = (gg - fit).float().bfloat16().double() + fit
    e_ctr2 = proj(ctr2) - r0
    n = lambda t: float(t.norm())  # noqa: E731
    print("%3d %5d %5d | %9.1f %9.3f | %8.4f %8.4f %8.4f" % (i, gg.shape[0] // 1000, C, n(gg) / max(n(r0), 1e-30), n(gb) * gg.shape[0] ** 0.5 / n(gg),
                                                          n(e_plain) / n(r0), n(e_ctr) / n(r0), n(e_ctr2) / n(r0)), flush=True)
