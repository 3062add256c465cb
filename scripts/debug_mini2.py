import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import yolo_oracle as yo
from mdcv.yolo.models import Darknet
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
def run(B, T, prec, seed):
    os.chdir(os.path.join(G, "mini"))
    orc = yo.DarknetOracle("mini.cfg", anchors=yo.read_anchor_row("dataset/train.csv")); orc.load_weights("mini.weights", [18, 18]); orc.keep_outs = True
    net = Darknet("mini.cfg", 2.0, 1.6, 25.0, 0.1, False, precision=prec); net.load_weights("mini.weights", [18, 18]); net = net.cuda().train()
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, 64, 64, generator=g); tg = torch.zeros(B, T, 5)
    for b in range(B):
        n = 1 + b % T
        tg[b, :n, 1:3] = torch.rand(n, 2, generator=g) * 0.9 + 0.05; tg[b, :n, 3:5] = torch.rand(n, 2, generator=g) * 0.28 + 0.02
    for k in orc.trainable(): orc.params[k].requires_grad_(True)
    ref = orc.forward(x, tg); ref[0].sum().backward()
    outs_ref = orc.last_outs
    for it in range(2):
        for p in net.parameters(): p.grad = None
        out = net(x.cuda(), tg.cuda()); out[0].sum().backward()
        torch.cuda.synchronize()
        plan = list(net._plans.values())[0]
        print(f"--- B={B} it={it}")
        for i, node in enumerate(plan.outs):
            if node is None or net.module_defs[i]["type"] == "yolo": continue
            r = outs_ref[i]
            a = node.act.dense().float().permute(0, 3, 1, 2)[:, :r.shape[1]].cpu()
            ea = float((a - r.detach()).abs().max() / max(float(r.detach().abs().max()), 1e-12))
            msg = f"  out[{i:2d}] {net.module_defs[i]['type'][:5]:5s} {node.name:9s} act relerr {ea:.1e}"
            if node.grad is not None and r.grad is not None:
                gq = node.grad.dense().float().permute(0, 3, 1, 2)[:, :r.shape[1]].cpu()
                eg = float((gq - r.grad).abs().max() / max(float(r.grad.abs().max()), 1e-12))
                msg += f"  grad relerr {eg:.1e} state={node.gstate} ldc={node.grad.ldc}"
            print(msg)
run(2, 4, "fp32", 1)
run(5, 6, "fp32", 2)
