"""One weight-gradient shape in a loop (for rocprofv3 --pmc / timing).  usage: wgrad_one.py B H Cin Cout k dil variant iters"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
B, H, Ci, Co, k, dil, variant, iters = (int(v) for v in sys.argv[1:9])
pad = dil * (k - 1) // 2
L.conv2d_wgrad_set_variant(variant)
xs = [torch.randn(B * H * H * Ci, device="cuda").to(torch.bfloat16) for _ in range(3)]
dys = [torch.randn(B * H * H * Co, device="cuda").to(torch.bfloat16) for _ in range(3)]
sp = L.conv2d_wgrad_splits_geom(1, B, H, H, Ci, H, H, Co, k, k, 1, pad, dil, Co, Ci)
ws = torch.empty(sp * Co * k * k * Ci, device="cuda"); dw = torch.empty(Co * Ci * k * k, device="cuda")
def call(i):
    return L.conv2d_wgrad(1, dys[i % 3].data_ptr(), Co, xs[i % 3].data_ptr(), Ci, ws.data_ptr(), sp, dw.data_ptr(), 0, B, H, H, Ci, Ci, H, H, Co, Co, k, k, 1, pad, dil, st)
for i in range(3): assert call(i) == 0
torch.cuda.synchronize()
e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p(); L.event_create(ctypes.byref(e0)); L.event_create(ctypes.byref(e1))
L.event_record(e0, st)
for i in range(iters): call(i)
L.event_record(e1, st); L.event_sync(e1)
ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms)); t = ms.value / iters
print("wgrad", sys.argv[1:8], "splits", sp, "us %.1f TF/s %.0f  operand GB/s %.0f" % (t * 1e3, 2.0 * B * H * H * Co * k * k * Ci / t / 1e9, 2.0 * B * H * H * (Ci + Co) / t / 1e6))
