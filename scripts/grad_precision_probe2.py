"""bf16 mode against fp32-kernel mode on the same weights / batch: cosine of the activation gradient g entering each BatchNorm backward, of the
dy leaving it, and of the conv weight gradients, in backward order.  usage: grad_precision_probe2.py [batch=8]"""
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
        rec.append((dout, y1, bs1, r))
    return r


engine.Plan.emit_bn_act_bwd = spy
tmp = tempfile.mkdtemp(); cfg = bench.write_yolo_cfg(tmp)
g = torch.Generator().manual_seed(21)
x = torch.rand(B, 3, 416, 416, generator=g).cuda(); tg = bench.synth_targets(B, 16, g).cuda()
runs = {}
sd = None
for prec in ("fp32", "bf16"):
    rec.clear()
    cwd = os.getcwd(); os.chdir(tmp)
    torch.manual_seed(3)
    net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision=prec).cuda().train()
    os.chdir(cwd)
    if sd is None: sd = {k: v.clone() for k, v in net.state_dict().items()}
    else: net.load_state_dict(sd)
    out = net(x, tg); out[0].sum().backward(); torch.cuda.synchronize()
    runs[prec] = dict(loss=float(out[0].detach()),
                      g=[d.dense().float().reshape(-1, d.C)[:, :bs.C].clone() for d, y, bs, r in rec],
                      y=[y.dense().float().reshape(-1, y.C)[:, :bs.C].clone() for d, y, bs, r in rec],
                      dy=[r.dense().float().reshape(-1, r.C)[:, :bs.C].clone() for d, y, bs, r in rec],
                      wg={n: p.grad.detach().float().clone() for n, p in net.named_parameters() if ".conv_" in n and n.endswith("weight")})
    del net
print("loss", runs["fp32"]["loss"], runs["bf16"]["loss"])
cos = lambda a, b: float((a.double().reshape(-1) @ b.double().reshape(-1)) / (a.double().norm() * b.double().norm() + 1e-30))  # noqa: E731
print("%3s %6s %5s | %8s %8s %8s | %s" % ("#", "M", "C", "cos y", "cos g", "cos dy", "|g| bf16/fp32"))
for i in range(len(runs["fp32"]["g"])):
    a, b = runs["fp32"], runs["bf16"]
    print("%3d %6d %5d | %8.5f %8.5f %8.5f | %.3f" % (i, a["g"][i].shape[0], a["g"][i].shape[1], cos(a["y"][i], b["y"][i]), cos(a["g"][i], b["g"][i]),
                                                 cos(a["dy"][i], b["dy"][i]), float(b["g"][i].norm() / a["g"][i].norm())), flush=True)
print("conv weight gradients (module order):")
for n in runs["fp32"]["wg"]:
    print("  %-40s cos %.4f" % (n, cos(runs["fp32"]["wg"][n], runs["bf16"]["wg"][n])))
