import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import yolo_oracle as yo
from mdcv.yolo.models import Darknet
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
def key(n):
    _, i, mod, leaf = n.split("."); return ("conv" if mod.startswith("conv") else "bn") + f"{i}.{leaf}"
def run(B, T, prec, seed):
    os.chdir(os.path.join(G, "mini"))
    orc = yo.DarknetOracle("mini.cfg", anchors=yo.read_anchor_row("dataset/train.csv")); orc.load_weights("mini.weights", [18, 18])
    net = Darknet("mini.cfg", 2.0, 1.6, 25.0, 0.1, False, precision=prec); net.load_weights("mini.weights", [18, 18]); net = net.cuda().train()
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, 64, 64, generator=g); tg = torch.zeros(B, T, 5)
    for b in range(B):
        n = 1 + b % T
        tg[b, :n, 1:3] = torch.rand(n, 2, generator=g) * 0.9 + 0.05; tg[b, :n, 3:5] = torch.rand(n, 2, generator=g) * 0.28 + 0.02
    for k in orc.trainable(): orc.params[k].requires_grad_(True)
    ref = orc.forward(x, tg); ref[0].sum().backward()
    for it in range(2):
        for p in net.parameters(): p.grad = None
        out = net(x.cuda(), tg.cuda()); out[0].sum().backward()
        torch.cuda.synchronize()
        print(f"B={B} {prec} it={it} loss {float(out[0]):.6f} ref {float(ref[0]):.6f}")
        if it == 0:
            for n, p in net.named_parameters():
                r = orc.params[key(n)].grad
                e = float((p.grad.cpu() - r).abs().max() / max(float(r.abs().max()), 1e-12))
                print(f"   {n:45s} relerr {e:.2e}  |ref| {float(r.abs().max()):.3e}")
        # only the first iteration is comparable (running stats do not affect train-mode grads, so it=1 should match too)
        else:
            worst = max(float((p.grad.cpu() - orc.params[key(n)].grad).abs().max() / max(float(orc.params[key(n)].grad.abs().max()), 1e-12)) for n, p in net.named_parameters())
            print("   second iteration worst relerr", worst)
run(2, 4, "fp32", 1)
run(5, 6, "fp32", 2)
