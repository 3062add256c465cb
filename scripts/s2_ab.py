"""Stride-2 3x3 data gradients of yolo_baseline at batch 32, alone: four class launches (variant 14) vs one launch with two workgroups per
tile (15) for the sparse grids; the dense ones for reference.  usage: s2_ab.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p(); L.event_create(ctypes.byref(e0)); L.event_create(ctypes.byref(e1))
for (B, H, Co, Ci) in [(32, 13, 1024, 512), (32, 26, 512, 256), (32, 52, 256, 128), (32, 104, 128, 64), (32, 208, 64, 32)]:
    dy = [torch.randn(B * H * H * Co, device="cuda").to(torch.bfloat16) for _ in range(3)]
    wd = torch.randn(Ci * 9 * Co, device="cuda").to(torch.bfloat16)
    dx = torch.empty(B * 4 * H * H * Ci, device="cuda", dtype=torch.bfloat16)
    res = []
    for v in (14, 15, 16, 17):
        L.conv2d_set_variant(v)
        def call(i):
            return L.conv2d(1, 1, dy[i % 3].data_ptr(), Co, wd.data_ptr(), dx.data_ptr(), Ci, None, None, 0, None, B, H, H, Co, 2 * H, 2 * H, Ci, 3, 3, 2, 1, 1, st)
        for i in range(5): assert call(i) == 0
        L.event_record(e0, st)
        for i in range(30): call(i)
        L.event_record(e1, st); L.event_sync(e1)
        ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms))
        res.append("v%d %.1f us" % (v, 1e3 * ms.value / 30))
    L.conv2d_set_variant(15); L.conv2d_set_variant(17)
    fl = 2.0 * B * H * H * Co * 9 * Ci
    print((B, H, Co, Ci), " | ".join(res), "| floor %.1f us" % (fl / 1.5e15 * 1e6), flush=True)
