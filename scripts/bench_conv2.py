"""Why are convs slower in-network than in a tight loop?  (a) sustained clocks, (b) cold caches (rotating buffer sets)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
def ev():
    e = ctypes.c_void_p(); L.event_create(ctypes.byref(e)); return e
def run(shape, variant, iters, nsets, stats=False):
    B, H, Ci, Co, k, s, mode = shape
    pad = (k - 1) // 2
    Ho = (H + 2 * pad - k) // s + 1
    xs = [torch.randn(B * H * H * Ci, device="cuda").to(torch.bfloat16) for _ in range(nsets)]
    ys = [torch.randn(B * Ho * Ho * Co, device="cuda").to(torch.bfloat16) for _ in range(nsets)]
    wf = (torch.randn(Co * k * k * Ci, device="cuda") * 0.05).to(torch.bfloat16)
    stt = torch.zeros(L.conv2d_stats_rows_geom(1, B, Ho, Ho, Ci, Co, k, k, s, pad, 1, Ci) * 2 * Co, device="cuda") if stats else None
    L.conv2d_set_variant(variant)
    def call(i):
        x, y = xs[i % nsets], ys[i % nsets]
        if mode == 0:
            return L.conv2d(1, 0, x.data_ptr(), Ci, wf.data_ptr(), y.data_ptr(), Co, None, None, 0, stt.data_ptr() if stats else None, B, H, H, Ci, Ho, Ho, Co, k, k, s, pad, 1, st)
        return L.conv2d(1, 1, y.data_ptr(), Co, wf.data_ptr(), x.data_ptr(), Ci, None, None, 0, None, B, Ho, Ho, Co, H, H, Ci, k, k, s, pad, 1, st)
    for i in range(3): assert call(i) == 0
    e0, e1 = ev(), ev()
    L.event_record(e0, st)
    for i in range(iters): call(i)
    L.event_record(e1, st); L.event_sync(e1)
    ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms))
    t = ms.value / iters
    return 2.0 * B * Ho * Ho * Co * k * k * Ci / t / 1e9
for sh in [(32, 26, 256, 512, 3, 1, 0), (32, 26, 256, 512, 3, 1, 1), (32, 13, 512, 1024, 3, 1, 0), (32, 52, 128, 256, 3, 1, 0)]:
    print(sh, "burst10/1set %.0f | sustained400/1set %.0f | sustained400/24sets(cold) %.0f | +stats %.0f" % (
        run(sh, -1, 10, 1), run(sh, -1, 400, 1), run(sh, -1, 400, 24), run(sh, -1, 400, 24, True)), flush=True)
