import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rektnet_oracle as ro
print("cpu_count", os.cpu_count())
sd = ro.init_state(0)
params = [v.requires_grad_(True) for k, v in sd.items() if "running" not in k]
x = torch.rand(8, 3, 80, 80); tp = torch.rand(8, 7, 2)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    ts = []
    for it in range(3):
        t0 = time.perf_counter()
        for p in params: p.grad = None
        hm, pts = ro.keypoint_forward(x, sd, train=True)
        ro.cross_ratio_loss(hm, pts, None, tp, "l1_softargmax", True, 0.05, 0.05)[2].backward()
        ts.append(time.perf_counter() - t0)
    print(nt, ["%.3f" % t for t in ts], flush=True)
