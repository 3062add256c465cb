"""mdcv_bn_act_fwd (after a finalize launch) against mdcv_bn_act_fwd_statsfold, alone: us per launch.   usage: fold_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib()
BF16 = _lib.BF16
st = lambda: torch.cuda.current_stream().cuda_stream


def bench(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (M, C, ng) in [(21632, 512, 8), (5408, 1024, 4), (86528, 256, 16), (86528, 128, 16)]:
    y = torch.randn(M, C, device="cuda").bfloat16(); z = torch.empty_like(y)
    sup = torch.rand(ng, 2, C, device="cuda") * 100
    sup[:, 1] += 1e4
    gam, bet = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    co = [torch.zeros(C, device="cuda") for _ in range(4)]
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    rows = ng * 22
    part = torch.rand(rows, 2, C, device="cuda"); part[:, 1] += 100
    scr = torch.zeros(3 * C, dtype=torch.float64, device="cuda")
    plain = lambda: L.bn_act_fwd(BF16, y.data_ptr(), C, co[0].data_ptr(), co[1].data_ptr(), None, 0, None, None, None, 0, z.data_ptr(), C, M, C, 1, 0.1, st())
    fin = lambda: L.bn_stats_finalize(part.data_ptr(), rows, scr.data_ptr(), float(M), gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5,
                                      *[c.data_ptr() for c in co], C, st())
    def pair(): fin(); plain()
    res = ["M %6d C %4d groups %2d: plain %6.1f  finalize+plain %6.1f " % (M, C, ng, bench(plain), bench(pair))]
    for blocks in (2048, 1024, 512, 256):
        L.bn_act_fwd_statsfold_blocks(blocks)
        fold = lambda: L.bn_act_fwd_statsfold(BF16, y.data_ptr(), C, sup.data_ptr(), ng, float(M), gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                              0.1, 1e-5, *[c.data_ptr() for c in co], None, 0, z.data_ptr(), C, M, C, 1, 0.1, st())
        res.append(" fold@%d %6.1f" % (blocks, bench(fold)))
    print("".join(res), flush=True)
