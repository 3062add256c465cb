"""The first conv -> BatchNorm -> activation of yolo_baseline (416^2 x 32 images, 3 -> 32 channels) alone: the two streaming passes of csrc/first_conv.hip against
mdcv_conv2d (+ statistics rows) and mdcv_bn_act_fwd, rotating buffer sets (us per launch, GB/s of the launch's own traffic).   usage: first_conv_ab.py [iters]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B, H, W, Co = 32, 416, 416, 32
NS = 3
xs = [torch.rand(B, H, W, 8, device="cuda").to(torch.bfloat16) for _ in range(NS)]
ys = [torch.empty(B, H, W, Co, dtype=torch.bfloat16, device="cuda") for _ in range(NS)]
zs = [torch.empty(B, H, W, Co, dtype=torch.bfloat16, device="cuda") for _ in range(NS)]
wf = (torch.randn(Co * 72, device="cuda") * 0.2).to(torch.bfloat16)
sc, sh = torch.rand(Co, device="cuda") + 0.5, torch.randn(Co, device="cuda") * 0.1
rows = L.first_conv_rows(B, H)
part = torch.empty(rows * 2 * Co, device="cuda")
rows0 = L.conv2d_stats_rows_geom(1, B, H, W, 8, Co, 3, 3, 1, 1, 1, 8)
part0 = torch.empty(rows0 * 2 * Co, device="cuda")
e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p(); L.event_create(ctypes.byref(e0)); L.event_create(ctypes.byref(e1))


def timeit(fn):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    L.event_record(e0, st)
    for i in range(iters): fn(i)
    L.event_record(e1, st); L.event_sync(e1)
    ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms))
    return ms.value / iters * 1e3


xb, yb = B * H * W * 16, B * H * W * Co * 2
cases = [
    ("pass 1: statistics from x", lambda i: L.first_conv_stats(1, xs[i % NS].data_ptr(), 8, wf.data_ptr(), part.data_ptr(), B, H, W, st), xb),
    ("pass 2: x -> y and z", lambda i: L.first_conv_bn_act(1, xs[i % NS].data_ptr(), 8, wf.data_ptr(), sc.data_ptr(), sh.data_ptr(), 1, 0.1, ys[i % NS].data_ptr(), Co,
                                                            zs[i % NS].data_ptr(), Co, B, H, W, st), xb + 2 * yb),
    ("mdcv_conv2d + statistics rows", lambda i: L.conv2d(1, 0, xs[i % NS].data_ptr(), 8, wf.data_ptr(), ys[i % NS].data_ptr(), Co, None, None, 0, part0.data_ptr(),
                                                         B, H, W, 8, H, W, Co, 3, 3, 1, 1, 1, st), xb + yb),
    ("mdcv_bn_act_fwd", lambda i: L.bn_act_fwd(1, ys[i % NS].data_ptr(), Co, sc.data_ptr(), sh.data_ptr(), None, 0, None, None, None, 0, zs[i % NS].data_ptr(), Co,
                                               B * H * W, Co, 1, 0.1, st), 2 * yb),
]
for name, fn, nbytes in cases:
    assert fn(0) == 0
    us = timeit(fn)
    print("%-34s %8.1f us  %6.0f GB/s" % (name, us, nbytes / us / 1e3), flush=True)
