import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
B, H, Ci, Co, k, s, mode, variant, iters = [int(v) for v in sys.argv[1:10]]
pad = (k - 1) // 2; Ho = (H + 2 * pad - k) // s + 1
x = torch.randn(B * H * H * Ci, device="cuda").to(torch.bfloat16); y = torch.randn(B * Ho * Ho * Co, device="cuda").to(torch.bfloat16)
wf = (torch.randn(Co * k * k * Ci, device="cuda") * 0.05).to(torch.bfloat16)
L.conv2d_set_variant(variant)
for _ in range(iters):
    if mode == 0: L.conv2d(1, 0, x.data_ptr(), Ci, wf.data_ptr(), y.data_ptr(), Co, None, None, 0, None, B, H, H, Ci, Ho, Ho, Co, k, k, s, pad, 1, st)
    else: L.conv2d(1, 1, y.data_ptr(), Co, wf.data_ptr(), x.data_ptr(), Ci, None, None, 0, None, B, Ho, Ho, Co, H, H, Ci, k, k, s, pad, 1, st)
torch.cuda.synchronize()
