"""A/B timing of the weight-gradient kernels (variant 8: kw-shared kernel wherever eligible; 0: default dispatch; 9: generic) on the 3x3 stride-1 layer shapes."""
import ctypes, os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "8,9,0").split(",")]
iters, rounds = 100, 3
SHAPES = [(32, 52, 128, 256), (32, 26, 256, 512), (32, 13, 512, 1024), (256, 80, 128, 128)]
e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p(); L.event_create(ctypes.byref(e0)); L.event_create(ctypes.byref(e1))
for (B, H, Ci, Co) in SHAPES:
    xs = [torch.randn(B * H * H * Ci, device="cuda").to(torch.bfloat16) for _ in range(3)]
    dys = [torch.randn(B * H * H * Co, device="cuda").to(torch.bfloat16) for _ in range(3)]
    dw = torch.empty(Co * Ci * 9, device="cuda")
    res = {}
    for v in variants:
        L.conv2d_wgrad_set_variant(v)
        splits = L.conv2d_wgrad_splits_geom(1, B, H, H, Ci, H, H, Co, 3, 3, 1, 1, 1, Co, Ci)
        ws = torch.empty(splits * Co * 9 * Ci, device="cuda")
        def call(i):
            return L.conv2d_wgrad(1, dys[i % 3].data_ptr(), Co, xs[i % 3].data_ptr(), Ci, ws.data_ptr(), splits, dw.data_ptr(), 0, B, H, H, Ci, Ci, H, H, Co, Co, 3, 3, 1, 1, 1, st)
        for i in range(10): assert call(i) == 0
        ts = []
        for r in range(rounds):
            L.event_record(e0, st)
            for i in range(iters): call(i)
            L.event_record(e1, st); L.event_sync(e1)
            ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms)); ts.append(ms.value / iters)
        res[v] = (statistics.median(ts), splits)
    fl = 2.0 * B * H * H * Co * 9 * Ci
    print((B, H, Ci, Co), " | ".join("v%d: %.1f us %4.0f TF (splits %d)" % (v, 1e3 * t, fl / t / 1e9, s) for v, (t, s) in res.items()), flush=True)
L.conv2d_wgrad_set_variant(0)
