"""One conv shape in a loop (for rocprofv3 --pmc / timing).  usage: conv_one.py B H Cin Cout k stride mode variant iters"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
B, H, Ci, Co, k, s, mode, variant, iters = (int(v) for v in sys.argv[1:10])
pad = (k - 1) // 2
Ho = (H + 2 * pad - k) // s + 1
nsets = 8
xs = [torch.randn(B * H * H * Ci, device="cuda").to(torch.bfloat16) for _ in range(nsets)]
ys = [torch.randn(B * Ho * Ho * Co, device="cuda").to(torch.bfloat16) for _ in range(nsets)]
wf = (torch.randn(Co * k * k * Ci, device="cuda") * 0.05).to(torch.bfloat16)
stt = torch.zeros(L.conv2d_stats_rows_geom(1, B, Ho, Ho, Ci, Co, k, k, s, pad, 1, Ci) * 2 * Co, device="cuda")
L.conv2d_set_variant(variant)
def call(i):
    x, y = xs[i % nsets], ys[i % nsets]
    if mode == 0:
        return L.conv2d(1, 0, x.data_ptr(), Ci, wf.data_ptr(), y.data_ptr(), Co, None, None, 0, stt.data_ptr(), B, H, H, Ci, Ho, Ho, Co, k, k, s, pad, 1, st)
    return L.conv2d(1, 1, y.data_ptr(), Co, wf.data_ptr(), x.data_ptr(), Ci, None, None, 0, None, B, Ho, Ho, Co, H, H, Ci, k, k, s, pad, 1, st)
for i in range(3): assert call(i) == 0
torch.cuda.synchronize()
e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p(); L.event_create(ctypes.byref(e0)); L.event_create(ctypes.byref(e1))
L.event_record(e0, st)
for i in range(iters): call(i)
L.event_record(e1, st); L.event_sync(e1)
ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms))
t = ms.value / iters
print("shape", sys.argv[1:9], "ms %.4f TF/s %.0f" % (t, 2.0 * B * Ho * Ho * Co * k * k * Ci / t / 1e9))
