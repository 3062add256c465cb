"""A/B micro-benchmark of mdcv_conv2d tile variants on the layer shapes of yolo_baseline@416 (B=32) and RektNet (B=256)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
SHAPES = [  # B, H, Cin, Cout, k, stride, mode
    (32, 13, 512, 1024, 3, 1, 0), (32, 13, 512, 1024, 3, 1, 1), (32, 13, 1024, 512, 1, 1, 0),
    (32, 26, 256, 512, 3, 1, 0), (32, 26, 256, 512, 3, 1, 1), (32, 26, 512, 256, 1, 1, 0),
    (32, 52, 128, 256, 3, 1, 0), (32, 52, 128, 256, 3, 1, 1), (32, 52, 256, 128, 1, 1, 0),
    (32, 104, 64, 128, 3, 1, 0), (32, 104, 64, 128, 3, 1, 1),
    (32, 52, 256, 512, 3, 2, 0), (32, 52, 256, 512, 3, 2, 1),
    (256, 80, 128, 128, 3, 1, 0), (256, 80, 128, 128, 3, 1, 1), (256, 80, 64, 128, 3, 1, 0),
]
def ev():
    e = ctypes.c_void_p(); L.event_create(ctypes.byref(e)); return e
def run(shape, variant, iters=10):
    B, H, Ci, Co, k, s, mode = shape
    pad = (k - 1) // 2
    Ho = (H + 2 * pad - k) // s + 1
    x = torch.randn(B * H * H * Ci, device="cuda").to(torch.bfloat16)
    y = torch.randn(B * Ho * Ho * Co, device="cuda").to(torch.bfloat16)
    wf = (torch.randn(Co * k * k * Ci, device="cuda") * 0.05).to(torch.bfloat16)
    L.conv2d_set_variant(variant)
    def call():
        if mode == 0:
            return L.conv2d(1, 0, x.data_ptr(), Ci, wf.data_ptr(), y.data_ptr(), Co, None, None, 0, None, B, H, H, Ci, Ho, Ho, Co, k, k, s, pad, 1, st)
        return L.conv2d(1, 1, y.data_ptr(), Co, wf.data_ptr(), x.data_ptr(), Ci, None, None, 0, None, B, Ho, Ho, Co, H, H, Ci, k, k, s, pad, 1, st)
    for _ in range(3):
        rc = call(); assert rc == 0, rc
    e0, e1 = ev(), ev()
    L.event_record(e0, st)
    for _ in range(iters): call()
    L.event_record(e1, st); L.event_sync(e1)
    ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms))
    t = ms.value / iters
    fl = 2.0 * B * Ho * Ho * Co * k * k * Ci
    return t, fl / t / 1e9
names = {106: "g128x128gen", 6: "g128x128UT", 9: "g128x128s3UT", 107: "g128x64gen", 7: "g128x64UT", 108: "g256x128gen", 8: "g256x128UT", 11: "g256x128s3UT"}
print("shape".ljust(40), "  ".join(n.rjust(12) for n in names.values()))
for sh in SHAPES:
    row = []
    for v in names:
        t, tf = run(sh, v)
        row.append(f"{tf:6.0f}TF")
    print(str(sh).ljust(40), "  ".join(r.rjust(12) for r in row), flush=True)
L.conv2d_set_variant(-1)
