"""Debug: which geometries leave promised partial rows unwritten in mdcv_conv2d_dgrad_bnsums (dtype from argv: 0 fp32, 1 bf16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
dt = int(sys.argv[1]) if len(sys.argv) > 1 else 0
TD = {0: torch.float32, 1: torch.bfloat16}[dt]
# (B, Cin_of_conv = channels of dx, H, W of dx, Cout_of_conv = channels of dy, k, stride, pad)
GEOMS = []
for B in (3, 16, 31, 32):
    GEOMS += [(B, 32, 416, 416, 64, 3, 2, 1), (B, 64, 208, 208, 32, 1, 1, 0), (B, 32, 208, 208, 64, 3, 1, 1), (B, 64, 208, 208, 128, 3, 2, 1),
              (B, 128, 104, 104, 64, 1, 1, 0), (B, 64, 104, 104, 128, 3, 1, 1), (B, 128, 104, 104, 256, 3, 2, 1), (B, 256, 52, 52, 128, 1, 1, 0),
              (B, 128, 52, 52, 256, 3, 1, 1), (B, 256, 52, 52, 512, 3, 2, 1), (B, 512, 26, 26, 256, 1, 1, 0), (B, 256, 26, 26, 512, 3, 1, 1),
              (B, 512, 26, 26, 1024, 3, 2, 1), (B, 1024, 13, 13, 512, 1, 1, 0), (B, 512, 13, 13, 1024, 3, 1, 1)]
for (B, Ci, H, W, Co, k, s, p) in GEOMS:
    Ho = (H + 2 * p - k) // s + 1
    rows = L.conv2d_dgrad_bnsums_rows(dt, B, Ho, Ho, Co, H, W, Ci, k, k, s, p, 1, Co)
    if rows <= 0:
        print((B, Ci, H, Co, k, s), "no fused path"); continue
    dy = torch.randn(B * Ho * Ho * Co, device="cuda").to(TD); wd = (torch.randn(Ci * k * k * Co, device="cuda") * 0.05).to(TD)
    y = torch.randn(B * H * W * Ci, device="cuda").to(TD); dx = torch.empty(B * H * W * Ci, device="cuda", dtype=TD)
    sc = torch.ones(Ci, device="cuda"); sh = torch.zeros(Ci, device="cuda"); mean = torch.zeros(Ci, device="cuda")
    part = torch.full((rows + 64, 2, Ci), float("nan"), device="cuda")
    rc = L.conv2d_dgrad_bnsums(dt, dy.data_ptr(), Co, wd.data_ptr(), dx.data_ptr(), Ci, None, 0, B, Ho, Ho, Co, H, W, Ci, k, k, s, p, 1,
                               y.data_ptr(), Ci, sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), 1, 0.1, part.data_ptr(), st)
    torch.cuda.synchronize()
    unwritten = int(torch.isnan(part[:rows]).any(dim=2).any(dim=1).sum()); beyond = int((~torch.isnan(part[rows:])).sum())
    flag = "  <-- UNWRITTEN" if unwritten else ("  <-- OVERFLOW" if beyond else "")
    print((B, Ci, H, Co, k, s), "rc", rc, "rows", rows, "unwritten rows", unwritten, "written beyond", beyond, flag)
