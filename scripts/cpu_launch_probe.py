"""How far ahead of the GPU does the Python launch loop run?  CPU time to ENQUEUE K train steps vs time until the GPU has finished them."""
import os, sys, time, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdcv.yolo.models import Darknet
from mdcv.optim import FusedAdam
tmp = tempfile.mkdtemp(); cfg = bench.write_yolo_cfg(tmp)
cwd = os.getcwd(); os.chdir(tmp); torch.manual_seed(0)
net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision="bf16").cuda().train(); os.chdir(cwd)
opt = FusedAdam(net, lr=1e-3)
g = torch.Generator().manual_seed(1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.rand(B, 3, 416, 416, generator=g).cuda(); tg = bench.synth_targets(B, 16, g).cuda()
def step():
    opt.zero_grad(); out = net(x, tg); out[0].sum().backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("B=%d  CPU enqueue %.2f ms/step   GPU-complete %.2f ms/step   (CPU ahead by %.1f ms at the end)" % (B, 1e3 * (t1 - t0) / K, 1e3 * (t2 - t0) / K, 1e3 * (t2 - t1)))
