"""Debug: does mdcv_conv2d write more BatchNorm partial rows than mdcv_conv2d_stats_rows_geom reports?  (every YOLOv3 / RektNet conv geometry)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
GEOMS = [(32, 416, 8, 32, 3, 1, 1), (32, 416, 32, 64, 3, 2, 1), (32, 208, 64, 32, 1, 1, 1), (32, 208, 32, 64, 3, 1, 1), (32, 208, 64, 128, 3, 2, 1),
         (32, 104, 128, 64, 1, 1, 1), (32, 104, 64, 128, 3, 1, 1), (32, 104, 128, 256, 3, 2, 1), (32, 52, 256, 128, 1, 1, 1), (32, 52, 128, 256, 3, 1, 1),
         (32, 52, 256, 512, 3, 2, 1), (32, 26, 512, 256, 1, 1, 1), (32, 26, 256, 512, 3, 1, 1), (32, 26, 512, 1024, 3, 2, 1), (32, 13, 1024, 512, 1, 1, 1),
         (32, 13, 512, 1024, 3, 1, 1), (32, 13, 1024, 256, 1, 1, 1), (32, 26, 768, 256, 1, 1, 1), (32, 52, 384, 128, 1, 1, 1), (32, 26, 256, 128, 1, 1, 1), (32, 13, 512, 256, 1, 1, 1),
         (256, 80, 8, 16, 7, 1, 1), (256, 80, 16, 16, 3, 1, 2), (256, 80, 16, 16, 3, 1, 1), (256, 80, 16, 16, 1, 1, 1), (256, 80, 16, 32, 3, 1, 2), (256, 80, 32, 32, 3, 1, 1),
         (256, 80, 32, 64, 3, 1, 2), (256, 80, 64, 64, 3, 1, 1), (256, 80, 64, 128, 3, 1, 2), (256, 80, 128, 128, 3, 1, 1), (256, 80, 64, 128, 1, 1, 1),
         (31, 52, 256, 128, 1, 1, 1), (31, 104, 128, 64, 1, 1, 1), (33, 26, 512, 256, 1, 1, 1), (31, 208, 64, 128, 3, 2, 1), (31, 104, 128, 256, 3, 2, 1), (31, 52, 128, 256, 3, 1, 1), (33, 13, 512, 1024, 3, 1, 1), (31, 26, 256, 512, 3, 1, 1), (255, 80, 64, 64, 3, 1, 1), (255, 80, 128, 128, 3, 1, 1), (255, 80, 16, 16, 1, 1, 1),
         (8, 416, 8, 32, 3, 1, 1), (8, 208, 64, 128, 3, 2, 1), (8, 52, 128, 256, 3, 1, 1), (2, 64, 8, 16, 3, 1, 1), (4, 13, 512, 1024, 3, 1, 1)]
SENT = 12345.0
for (B, H, Ci, Co, k, s, dil) in GEOMS:
    pad = dil * (k - 1) // 2; Ho = (H + 2 * pad - dil * (k - 1) - 1) // s + 1
    x = torch.randn(B * H * H * Ci, device="cuda").to(torch.bfloat16)
    wf = (torch.randn(Co * k * k * Ci, device="cuda") * 0.05).to(torch.bfloat16)
    rows = L.conv2d_stats_rows_geom(1, B, Ho, Ho, Ci, Co, k, k, s, pad, dil, Ci)
    guard = 4096
    stt = torch.full(((rows + guard) * 2 * Co,), SENT, device="cuda")
    y = torch.zeros(B * Ho * Ho * Co, device="cuda", dtype=torch.bfloat16)
    rc = L.conv2d(1, 0, x.data_ptr(), Ci, wf.data_ptr(), y.data_ptr(), Co, None, None, 0, stt.data_ptr(), B, H, H, Ci, Ho, Ho, Co, k, k, s, pad, dil, st)
    torch.cuda.synchronize()
    body, tail = stt[:rows * 2 * Co], stt[rows * 2 * Co:]
    over = int((tail != SENT).sum()); unwritten = int((body == SENT).sum())
    flag = "  <-- OVERFLOW" if over else ("  <-- rows not written" if unwritten else "")
    print((B, H, Ci, Co, k, s, dil), "rc", rc, "rows", rows, "written beyond:", over, "floats; unwritten inside:", unwritten, flag)
