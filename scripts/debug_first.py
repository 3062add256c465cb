"""Debug: first vs second model instance in one process, step-0 forward only: loss parts and per-layer activation checksums."""
import os, sys, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdcv.yolo.models import Darknet
tmp = tempfile.mkdtemp(); cfg = bench.write_yolo_cfg(tmp)
B = int(os.environ.get("DBG_B", "32"))
def run():
    cwd = os.getcwd(); os.chdir(tmp)
    torch.manual_seed(0)
    net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision="bf16")
    os.chdir(cwd)
    net = net.cuda().train()
    g = torch.Generator().manual_seed(1)
    x = torch.rand(B, 3, 416, 416, generator=g).cuda(); tg = bench.synth_targets(B, 16, g).cuda()
    out = net(x, tg)
    torch.cuda.synchronize()
    plan = [p for p in net._plans.values()][0]
    bufs = [b for b in plan.keep if torch.is_tensor(b)]
    sums = [float(b.float().double().sum()) if b.dtype in (torch.float32, torch.bfloat16) else None for b in bufs]
    return [float(v) for v in out], sums, [tuple(b.shape) for b in bufs]
a, sa, sh = run(); b, sb, _ = run(); c, sc, _ = run()
print("parts A", a[:3]); print("parts B", b[:3]); print("parts C", c[:3])
print("B==C buffers:", sum(1 for u, v in zip(sb, sc) if u is not None and u != v), "differ")
n = 0
for i, (u, v) in enumerate(zip(sa, sb)):
    if u is not None and u != v and not (u != u and v != v):
        print("buffer", i, sh[i], u, v); n += 1
        if n > 12: break
