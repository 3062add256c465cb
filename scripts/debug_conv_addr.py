"""Debug: does mdcv_conv2d give the same output for the same data at different addresses?  (208,64->128,3x3,s2), B=32."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
B, H, Ci, Co, k, s = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (32, 208, 64, 128, 3, 2))]
pad = (k - 1) // 2; Ho = (H + 2 * pad - k) // s + 1
torch.manual_seed(0)
x = torch.randn(B * H * H * Ci, device="cuda").to(torch.bfloat16)
wf = (torch.randn(Co * k * k * Ci, device="cuda") * 0.05).to(torch.bfloat16)
rows = L.conv2d_stats_rows_geom(1, B, Ho, Ho, Ci, Co, k, k, s, pad, 1, Ci)
def run(x, wf):
    y = torch.zeros(B * Ho * Ho * Co, device="cuda", dtype=torch.bfloat16)
    stt = torch.zeros(rows * 2 * Co, device="cuda")
    assert L.conv2d(1, 0, x.data_ptr(), Ci, wf.data_ptr(), y.data_ptr(), Co, None, None, 0, stt.data_ptr(), B, H, H, Ci, Ho, Ho, Co, k, k, s, pad, 1, st) == 0
    torch.cuda.synchronize()
    return y, stt
y0, s0 = run(x, wf)
y1, s1 = run(x, wf)
print("same addresses twice: equal", torch.equal(y0, y1), torch.equal(s0, s1))
junk = [torch.full((int(3e6 + 777 * i),), float(i), device="cuda") for i in range(5)]
x2 = x.clone(); w2 = wf.clone()
for name, (xx, ww) in {"x moved": (x2, wf), "w moved": (x, w2), "both": (x2, w2)}.items():
    y, ss = run(xx, ww)
    d = (y.float() - y0.float()).abs()
    print(name, "equal", torch.equal(y, y0), "max diff", float(d.max()), "n diff", int((d > 0).sum()), "stats equal", torch.equal(ss, s0))
    if not torch.equal(y, y0):
        idx = (d > 0).nonzero().flatten()[:5].tolist()
        print("   first diffs at", [(i // Co // (Ho * Ho), (i // Co) % (Ho * Ho) // Ho, (i // Co) % Ho, i % Co) for i in idx])
