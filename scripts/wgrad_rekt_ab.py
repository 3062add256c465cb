import ctypes, os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
iters = 20
codes = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "0,34020").split(",")]
e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p(); L.event_create(ctypes.byref(e0)); L.event_create(ctypes.byref(e1))
SH = [(80, 16, 16, 1), (80, 16, 32, 1), (80, 32, 32, 1), (80, 32, 32, 2), (80, 32, 64, 1), (80, 64, 64, 1), (80, 64, 64, 2), (80, 64, 128, 1), (80, 128, 128, 1), (80, 128, 128, 2)]
for (H, Ci, Co, dil) in SH:
    B = 256
    M = B * H * H
    g = torch.Generator(device="cuda").manual_seed(1)
    dys = [torch.randn(M, Co, device="cuda", generator=g).to(torch.bfloat16) for _ in range(2)]
    xs = [torch.randn(M, Ci, device="cuda", generator=g).to(torch.bfloat16) for _ in range(2)]
    dw = torch.zeros(Co, Ci, 3, 3, device="cuda")
    out = []
    for code in codes:
        dt = _lib.tuned(1, code)
        splits = L.conv2d_wgrad_splits_geom(dt, B, H, H, Ci, H, H, Co, 3, 3, 1, dil, dil, Co, Ci)
        ws = torch.empty(splits * Co * 9 * Ci, device="cuda")
        def call(i):
            rc = L.conv2d_wgrad(dt, dys[i % 2].data_ptr(), Co, xs[i % 2].data_ptr(), Ci, ws.data_ptr(), splits, dw.data_ptr(), 0, B, H, H, Ci, Ci,
                                H, H, Co, Co, 3, 3, 1, dil, dil, st)
            assert rc == 0, rc
        for i in range(3): call(i)
        torch.cuda.synchronize()
        ts = []
        for r in range(3):
            L.event_record(e0, st)
            for i in range(iters): call(i)
            L.event_record(e1, st); L.event_sync(e1)
            ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms))
            ts.append(ms.value / iters * 1e3)
        t = statistics.median(ts)
        out.append("code %6d: splits %3d  %7.1f us  %5.0f TF" % (code, splits, t, 2.0 * M * Co * 9 * Ci / t / 1e6))
    print("H=%3d %4d->%4d d%d | " % (H, Ci, Co, dil) + " | ".join(out), flush=True)
