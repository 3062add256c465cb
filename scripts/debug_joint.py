import os, sys, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mdcv.yolo.models import Darknet
from mdcv.yolo.postprocess import detect_postprocess
tmp = tempfile.mkdtemp(); cfg = bench.write_yolo_cfg(tmp); os.chdir(tmp)
torch.manual_seed(0)
net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision="bf16").cuda().eval()
x = torch.rand(4, 3, 608, 608).cuda()
with torch.no_grad():
    o = net(x)
print(o.shape, o.dtype, o[..., 4].min().item(), o[..., 4].max().item(), torch.isnan(o).sum().item())
thr = float(torch.quantile(o[0, :, 4].float(), 0.99)); print("thr", thr, (o[..., 4] > thr).sum(1))
d = detect_postprocess(o, None, thr, 0.25, 0.5, 608, 608)
print(d.count, d.boxes[0, :3], d.prob[0, :3])
print(o[0, :3, :6])
