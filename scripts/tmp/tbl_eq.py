import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
SH = [(13, 512, 1024, 512), (26, 256, 512, 256), (52, 128, 256, 128), (104, 64, 128, 64), (208, 32, 64, 32), (26, 256, 512, 768), (52,128,256,384)]
for (H, Ci, Co, xl) in SH:
    B = 32
    M = B * H * H
    g = torch.Generator(device="cuda").manual_seed(1)
    dy = torch.randn(M, Co, device="cuda", generator=g).to(torch.bfloat16)
    xw = torch.randn(M, xl, device="cuda", generator=g).to(torch.bfloat16)
    xp = xw.data_ptr() + (xl - Ci) * 2
    res = []
    for code in (0, 34040, 0, 34040):
        dt = _lib.tuned(1, code)
        splits = L.conv2d_wgrad_splits_geom(dt, B, H, H, Ci, H, H, Co, 3, 3, 1, 1, 1, Co, xl)
        ws = torch.empty(splits * Co * 9 * Ci, device="cuda")
        dw = torch.zeros(Co, Ci, 3, 3, device="cuda")
        rc = L.conv2d_wgrad(dt, dy.data_ptr(), Co, xp, xl, ws.data_ptr(), splits, dw.data_ptr(), 0, B, H, H, Ci, Ci, H, H, Co, Co, 3, 3, 1, 1, 1, st)
        assert rc == 0
        torch.cuda.synchronize()
        res.append(dw.clone())
    print(H, Ci, Co, xl, "splits", splits, "tbl==step", torch.equal(res[0], res[1]), "tbl==tbl", torch.equal(res[0], res[2]), "step==step", torch.equal(res[1], res[3]),
          "maxdiff", (res[0] - res[1]).abs().max().item(), flush=True)
