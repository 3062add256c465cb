"""Step time of identical YOLOv3 models built one after another in ONE process (the caching allocator hands every model's plan different
addresses): is the step time bimodal, and which kernels carry the difference?  usage: bimodal_probe.py [n]"""
import os, sys, tempfile, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdcv.yolo.models import Darknet
from mdcv.optim import FusedAdam

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
cfg = bench.write_yolo_cfg(tmp)
g = torch.Generator().manual_seed(1000)
x, tg = torch.rand(32, 3, 416, 416, generator=g).to(dev), bench.synth_targets(32, 16, g).to(dev)
res = []
for it in range(n):
    os.chdir(tmp)
    torch.manual_seed(0)
    net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True).to(dev).train()
    opt = FusedAdam(net, lr=1e-3)

    def step():
        opt.zero_grad()
        net(x, tg)[0].sum().backward()
        opt.step()
    for _ in range(8): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    ks = bench.in_step_kernel_times(step)
    plan = [p for p in net._plans.values() if p.has_bwd][0]
    ptrs = sorted(t.data_ptr() for t in plan.keep if isinstance(t, torch.Tensor))
    print("model %d: %.3f ms   sum of kernel ms %.3f   lowest plan buffer 0x%x" % (it, dt * 1e3, sum(v[1] for v in ks.values()), ptrs[0] if ptrs else 0), flush=True)
    res.append((dt, ks))
    del net, opt, plan
    torch.cuda.empty_cache()
fast = min(res, key=lambda r: r[0]); slow = max(res, key=lambda r: r[0])
print("fastest %.3f ms, slowest %.3f ms; kernels that differ most (ms per step, slow - fast):" % (fast[0] * 1e3, slow[0] * 1e3))
diff = sorted(((slow[1].get(k, [0, 0])[1] - fast[1].get(k, [0, 0])[1], k) for k in set(fast[1]) | set(slow[1])), reverse=True)
for d, k in diff[:12]:
    print("  %+7.3f  %s" % (d, k[:110]))
