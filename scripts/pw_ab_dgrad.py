"""1x1 data gradients with the fused BatchNorm-backward sums (and the shortcut add) of yolo_baseline @416 batch 32 alone,
under every forced tile configuration; plus the plain data gradient and the apply-like lower bound (bytes / 5 TB/s).
usage: pw_ab_dgrad.py"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
B = 32
e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p(); L.event_create(ctypes.byref(e0)); L.event_create(ctypes.byref(e1))
NAMES = {-1: "auto", 6: "128x128/2", 9: "128x128/3", 7: "128x64/2", 10: "128x64/3", 8: "256x128/2", 11: "256x128/3"}
def timeit(call, n=40):
    for i in range(4):
        if call(i) != 0: return None
    L.event_record(e0, st)
    for i in range(n): call(i)
    L.event_record(e1, st); L.event_sync(e1)
    ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms))
    return ms.value / n * 1e3
# conv forward Ci -> Co ; its data gradient: in = dY [Co], out = dX [Ci]; fused sums belong to the BatchNorm that produced X (Ci channels)
for (H, Ci, Co, add) in [(52, 256, 128, True), (26, 512, 256, True), (13, 1024, 512, True), (52, 256, 256, False), (104, 128, 64, True)]:
    dys = [torch.randn(B * H * H * Co, device="cuda").to(torch.bfloat16) for _ in range(4)]
    dxs = [torch.empty(B * H * H * Ci, device="cuda", dtype=torch.bfloat16) for _ in range(4)]
    adds = [torch.randn(B * H * H * Ci, device="cuda").to(torch.bfloat16) for _ in range(4)]
    ys = [torch.randn(B * H * H * Ci, device="cuda").to(torch.bfloat16) for _ in range(4)]
    wd = (torch.randn(Co * Ci, device="cuda") * 0.05).to(torch.bfloat16)
    sc = torch.rand(Ci, device="cuda") + 0.5; sh = torch.randn(Ci, device="cuda"); mu = torch.randn(Ci, device="cuda")
    geom = (B, H, H, Co, H, H, Ci, 1, 1, 1, 0, 1)
    out = []
    for v in NAMES:
        L.cdll.mdcv_conv2d_set_variant(v)
        rows = L.conv2d_dgrad_bnsums_rows(1, *geom, Co)
        part = torch.zeros(max(rows, 1) * 2 * Ci * 2, device="cuda")
        def fused(i):
            return L.conv2d_dgrad_bnsums(1, dys[i % 4].data_ptr(), Co, wd.data_ptr(), dxs[i % 4].data_ptr(), Ci, adds[i % 4].data_ptr() if add else None, Ci,
                                         *geom, ys[i % 4].data_ptr(), Ci, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), 1, 0.1, part.data_ptr(), st)
        def plain(i):
            return L.conv2d(1, 1, dys[i % 4].data_ptr(), Co, wd.data_ptr(), dxs[i % 4].data_ptr(), Ci, None, adds[i % 4].data_ptr() if add else None, Ci, None,
                            *geom, st)
        tf = timeit(fused) if rows else None
        tp = timeit(plain)
        out.append("%s: %s / %s" % (NAMES[v], "%.1f" % tf if tf else "n/a", "%.1f" % tp if tp else "n/a"))
    L.cdll.mdcv_conv2d_set_variant(-1)
    byt = 2.0 * B * H * H * (Co + Ci * (3 if add else 2))
    print("%3d^2 dY %4d -> dX %4d add=%d (%.0f MB fused, %.1f us at 5 TB/s)  fused / plain us: %s" % (H, Co, Ci, add, byt / 1e6, byt / 5e6, "  ".join(out)))
