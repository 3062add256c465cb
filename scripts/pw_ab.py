"""1x1 layers of yolo_baseline @416 batch 32 alone: forward (with statistics) under every forced tile configuration.
usage: pw_ab.py"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
B = 32
e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p(); L.event_create(ctypes.byref(e0)); L.event_create(ctypes.byref(e1))
NAMES = {-1: "auto", 6: "128x128/2", 9: "128x128/3", 7: "128x64/2", 10: "128x64/3", 8: "256x128/2", 11: "256x128/3"}
for (H, Ci, Co) in [(52, 256, 128), (52, 128, 256), (26, 512, 256), (26, 256, 512), (13, 1024, 512), (13, 512, 1024), (104, 128, 64)]:
    xs = [torch.randn(B * H * H * Ci, device="cuda").to(torch.bfloat16) for _ in range(4)]
    ys = [torch.randn(B * H * H * Co, device="cuda").to(torch.bfloat16) for _ in range(4)]
    wf = (torch.randn(Co * Ci, device="cuda") * 0.05).to(torch.bfloat16)
    out = []
    for v in NAMES:
        L.cdll.mdcv_conv2d_set_variant(v if v >= 0 else -1)
        rows = L.conv2d_stats_rows_geom(1, B, H, H, Ci, Co, 1, 1, 1, 0, 1, Ci)
        stt = torch.zeros(rows * 2 * Co * 2, device="cuda")
        def call(i):
            return L.conv2d(1, 0, xs[i % 4].data_ptr(), Ci, wf.data_ptr(), ys[i % 4].data_ptr(), Co, None, None, 0, stt.data_ptr(), B, H, H, Ci, H, H, Co, 1, 1, 1, 0, 1, st)
        ok = all(call(i) == 0 for i in range(4))
        if not ok:
            out.append("%s: n/a" % NAMES[v]); continue
        L.event_record(e0, st)
        for i in range(40): call(i)
        L.event_record(e1, st); L.event_sync(e1)
        ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms))
        out.append("%s: %.1f" % (NAMES[v], ms.value / 40 * 1e3))
    L.cdll.mdcv_conv2d_set_variant(-1)
    byt = 2.0 * B * H * H * (Ci + Co)
    print("%3d^2 %4d->%4d (%.0f MB, %.1f GFLOP) us: %s" % (H, Ci, Co, byt / 1e6, 2.0 * B * H * H * Ci * Co / 1e9, "  ".join(out)))
