"""Per-layer cosine between the HIP path's weight gradients and the CPU oracle's (fp32), full YOLOv3 (batch from argv), bf16 and fp32, next to
the same cosine of the reference arithmetic under torch.autocast(bfloat16) (tests/golden/yolo_autocast_bf16_cos.json, batch 32)."""
import json, os, sys, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oracle import yolo_oracle as yo
from mdcv.yolo.models import Darknet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
precs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["bf16"]
tmp = tempfile.mkdtemp(); cfg = bench.write_yolo_cfg(tmp)
cwd = os.getcwd(); os.chdir(tmp)
orc = yo.DarknetOracle(cfg, anchors=yo.VANILLA_ANCHORS, seed=3)
os.chdir(cwd)
g = torch.Generator().manual_seed(21)
x = torch.rand(B, 3, 416, 416, generator=g); tg = bench.synth_targets(B, 16, g)
torch.set_num_threads(min(os.cpu_count() or 1, 32))
for k in orc.trainable(): orc.params[k].requires_grad_(True)
ref = orc.forward(x, tg); ref[0].sum().backward()
golden = json.load(open(os.path.join(ROOT, "tests", "golden", "yolo_autocast_bf16_cos.json")))["cos"]
for prec in precs:
    os.chdir(tmp); net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision=prec); os.chdir(cwd)
    sd = net.state_dict()
    for k in list(sd.keys()):
        i = k.split(".")[1]; leaf = k.split(".", 3)[3]
        if leaf == "num_batches_tracked": continue
        sd[k] = orc.params[(f"conv{i}." if ".conv_" in k else f"bn{i}.") + leaf].detach().clone()
    net.load_state_dict(sd); net = net.cuda().train()
    out = net(x.cuda(), tg.cuda()); out[0].sum().backward()
    print(prec, "loss", float(out[0]), float(ref[0]))
    named = dict(net.named_parameters())
    for n, p in named.items():
        if ".conv_" in n and n.endswith("weight"):
            i = n.split(".")[1]
            a = p.grad.detach().cpu().double().reshape(-1); b = orc.params[f"conv{i}.weight"].grad.double().reshape(-1)
            cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
            print("  layer %3s %-22s cos %.4f  (reference under autocast: %.4f)  |g| %.3e ref %.3e" % (i, tuple(p.shape), cos, golden.get(f"conv{i}.weight", float("nan")), float(a.norm()), float(b.norm())))
