"""mdcv_head1x1_f32 alone on KeypointNet's tensor (256 images x 80^2 pixels, 128 bf16 channels -> 7 fp32 logits): us per call and bytes / s.
usage: head_ab.py [iters]    MDCV_LIB=<other build> times another build of the library."""
import ctypes, os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
M, C, K = 256 * 6400, 128, 7
g = torch.Generator(device="cuda").manual_seed(1)
xs = [torch.randn(M, C, device="cuda", generator=g).to(torch.bfloat16) for _ in range(3)]
w = torch.randn(K, C, device="cuda", generator=g) * 0.1
b = torch.randn(K, device="cuda", generator=g)
out = torch.empty(M, 8, device="cuda")
e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p(); L.event_create(ctypes.byref(e0)); L.event_create(ctypes.byref(e1))
def call(i):
    rc = L.head1x1_f32(xs[i % 3].data_ptr(), C, w.data_ptr(), b.data_ptr(), out.data_ptr(), M, C, K, st); assert rc == 0
for i in range(5): call(i)
torch.cuda.synchronize()
ts = []
for r in range(3):
    L.event_record(e0, st)
    for i in range(iters): call(i)
    L.event_record(e1, st); L.event_sync(e1)
    ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms)); ts.append(ms.value / iters * 1e3)
t = statistics.median(ts)
ref = xs[0].float() @ w.t() + b
call(0); torch.cuda.synchronize()
print("head1x1_f32 %s: %.1f us  %.2f TB/s  max |err| %.2e" % (os.environ.get("MDCV_LIB", "tree"), t, (M * C * 2 + M * 32) / t / 1e6, float((out[:, :K] - ref).abs().max())))
