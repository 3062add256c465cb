"""A/B timing + agreement of the weight-gradient kernels on the 3x3 stride-1 layer shapes.
variants (mdcv_conv2d_wgrad_set_variant): 0 default dispatch ; 9 generic im2col kernel ; 8 kw-shared kernel wherever eligible ; 1800 default without the
tiled LDS-ring kernel.  usage: wgrad_ab.py [variants] [shapes: yolo|rekt|all]"""
import ctypes, os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,9").split(",")]
which = sys.argv[2] if len(sys.argv) > 2 else "all"
iters, rounds = 50, 3
YOLO = [(32, 52, 128, 256), (32, 26, 256, 512), (32, 13, 512, 1024), (32, 104, 64, 128), (16, 52, 128, 256), (8, 76, 128, 256)]
REKT = [(256, 80, 128, 128), (64, 80, 128, 128)]
BIG = [(128, 52, 128, 256), (128, 26, 256, 512), (128, 13, 512, 1024), (256, 52, 128, 256), (256, 13, 512, 1024)]
SHAPES = YOLO if which == "yolo" else REKT if which == "rekt" else BIG if which == "big" else YOLO + REKT
e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p(); L.event_create(ctypes.byref(e0)); L.event_create(ctypes.byref(e1))
for (B, H, Ci, Co) in SHAPES:
    xs = [torch.randn(B * H * H * Ci, device="cuda").to(torch.bfloat16) for _ in range(3)]
    dys = [torch.randn(B * H * H * Co, device="cuda").to(torch.bfloat16) for _ in range(3)]
    res, outs = {}, {}
    for v in variants:
        L.conv2d_wgrad_set_variant(v)
        splits = L.conv2d_wgrad_splits_geom(1, B, H, H, Ci, H, H, Co, 3, 3, 1, 1, 1, Co, Ci)
        ws = torch.empty(splits * Co * 9 * Ci, device="cuda")
        dw = torch.full((Co * Ci * 9,), float("nan"), device="cuda")
        def call(i):
            return L.conv2d_wgrad(1, dys[i % 3].data_ptr(), Co, xs[i % 3].data_ptr(), Ci, ws.data_ptr(), splits, dw.data_ptr(), 0, B, H, H, Ci, Ci, H, H, Co, Co, 3, 3, 1, 1, 1, st)
        assert call(0) == 0
        torch.cuda.synchronize()
        outs[v] = dw.clone()
        for i in range(5): assert call(i) == 0
        ts = []
        for r in range(rounds):
            L.event_record(e0, st)
            for i in range(iters): call(i)
            L.event_record(e1, st); L.event_sync(e1)
            ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms)); ts.append(ms.value / iters)
        res[v] = (statistics.median(ts), splits)
    fl = 2.0 * B * H * H * Co * 9 * Ci
    ref = outs[variants[-1]]
    agree = " ".join("v%d:maxrel=%.1e" % (v, float((outs[v] - ref).abs().max() / ref.abs().max())) for v in variants[:-1])
    print((B, H, Ci, Co), " | ".join("v%d: %.1f us %4.0f TF (splits %d)" % (v, 1e3 * t, fl / t / 1e9, s) for v, (t, s) in res.items()), "|", agree, flush=True)
L.conv2d_wgrad_set_variant(0)
