"""Micro-benchmark of mdcv_conv2d_wgrad on YOLOv3 / RektNet layer shapes (bf16)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
SHAPES = [(32, 52, 128, 256, 3, 1), (32, 26, 256, 512, 3, 1), (32, 13, 512, 1024, 3, 1), (32, 52, 256, 128, 1, 1),
          (256, 80, 128, 128, 3, 1), (256, 80, 16, 16, 3, 1), (32, 416, 8, 32, 3, 1)]
only = int(sys.argv[1]) if len(sys.argv) > 1 else -1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L.conv2d_wgrad_set_variant(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
def ev():
    e = ctypes.c_void_p(); L.event_create(ctypes.byref(e)); return e
for idx, (B, H, Ci, Co, k, s) in enumerate(SHAPES):
    if only >= 0 and idx != only: continue
    pad = (k - 1) // 2
    Ho = (H + 2 * pad - k) // s + 1
    x = torch.randn(B * H * H * Ci, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B * Ho * Ho * Co, device="cuda").to(torch.bfloat16)
    M, ktot = B * Ho * Ho, k * k * Ci
    sp = L.conv2d_wgrad_splits(1, M, Co, ktot)
    ws = torch.empty(sp * Co * ktot, device="cuda")
    dw = torch.empty(Co * Ci * k * k, device="cuda")
    def call():
        return L.conv2d_wgrad(1, dy.data_ptr(), Co, x.data_ptr(), Ci, ws.data_ptr(), sp, dw.data_ptr(), 0, B, H, H, Ci, Ci, Ho, Ho, Co, Co, k, k, s, pad, 1, st)
    for _ in range(3): assert call() == 0
    e0, e1 = ev(), ev()
    L.event_record(e0, st)
    for _ in range(iters): call()
    L.event_record(e1, st); L.event_sync(e1)
    ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms))
    t = ms.value / iters
    print((B, H, Ci, Co, k, s), "splits", sp, "%.3f ms  %.0f TF" % (t, 2.0 * M * Co * ktot / t / 1e9), flush=True)
