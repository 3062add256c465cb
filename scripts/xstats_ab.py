"""Forward BatchNorm statistics through exact accumulators (csrc/exact_acc.h), alone: us per launch.
  consumer: mdcv_bn_act_fwd / mdcv_bn_stats_finalize + mdcv_bn_act_fwd / mdcv_bn_act_fwd_xstats at 1..8 replicas and 512..2048 workgroups
  producer: mdcv_conv2d with partial rows / mdcv_conv2d_xstats at 1, 2, 8 replicas
usage: xstats_ab.py"""
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


for (M, C) in [(5408, 1024), (21632, 512), (86528, 256), (86528, 128), (346112, 128), (346112, 64), (1384448, 64), (1384448, 32)]:
    y = torch.randn(M, C, device="cuda").bfloat16(); z = torch.empty_like(y)
    gam, bet = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    co = [torch.zeros(C, device="cuda") for _ in range(4)]
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    rows = (M + 127) // 128
    part = torch.rand(rows, 2, C, device="cuda"); part[:, 1] += 100
    scr = torch.zeros(3 * C, dtype=torch.float64, device="cuda")
    plain = lambda: L.bn_act_fwd(BF16, y.data_ptr(), C, co[0].data_ptr(), co[1].data_ptr(), None, 0, None, None, None, 0, z.data_ptr(), C, M, C, 1, 0.1, st())
    fin = lambda: L.bn_stats_finalize(part.data_ptr(), rows, scr.data_ptr(), float(M), gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5,
                                      *[c.data_ptr() for c in co], C, st())
    def pair(): fin(); plain()
    res = ["M %7d C %4d: plain %6.1f  finalize+plain %6.1f " % (M, C, bench(plain), bench(pair))]
    for reps in (1, 2, 4, 8, 32):
        if reps > L.xstats_reps(1 << 30, C):
            continue
        acc = torch.randint(0, 1 << 30, (L.xstats_words(reps, C),), dtype=torch.int64, device="cuda")
        for blocks in (512,):
            f = lambda: L.bn_act_fwd_xstats(BF16, y.data_ptr(), C, acc.data_ptr(), reps, float(M), gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                            0.1, 1e-5, *[c.data_ptr() for c in co], None, 0, z.data_ptr(), C, M, C, 1, 0.1, st())
            res.append(" x%d@%d %6.1f" % (reps, blocks, bench(f)))
    print("".join(res), flush=True)

print()
for (B, Ci, H, W, Co, k, s) in [(32, 128, 52, 52, 256, 3, 1), (32, 256, 26, 26, 512, 3, 1), (32, 512, 13, 13, 1024, 3, 1), (32, 64, 104, 104, 128, 3, 1),
                                (32, 256, 52, 52, 128, 1, 1), (32, 512, 26, 26, 256, 1, 1), (32, 1024, 13, 13, 512, 1, 1), (32, 128, 104, 104, 64, 1, 1),
                                (32, 32, 416, 416, 64, 3, 2), (32, 64, 208, 208, 128, 3, 2), (32, 8, 416, 416, 32, 3, 1), (32, 64, 208, 208, 32, 1, 1)]:
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = torch.randn(B, H, W, Ci, device="cuda").bfloat16()
    w = (torch.randn(Co * k * k * Ci, device="cuda") / (Ci * k * k) ** 0.5).bfloat16()      # (any packed layout: timing only)
    y = torch.empty(B, Ho, Wo, Co, device="cuda", dtype=torch.bfloat16)
    geom = (B, H, W, Ci, Ho, Wo, Co, k, k, s, pad, 1)
    rows = L.conv2d_stats_rows_geom(BF16, B, Ho, Wo, Ci, Co, k, k, s, pad, 1, Ci)
    part = torch.zeros(rows, 2, Co, device="cuda")
    base = lambda: L.conv2d(BF16, 0, x.data_ptr(), Ci, w.data_ptr(), y.data_ptr(), Co, None, None, 0, part.data_ptr(), *geom, st())
    nost = lambda: L.conv2d(BF16, 0, x.data_ptr(), Ci, w.data_ptr(), y.data_ptr(), Co, None, None, 0, None, *geom, st())
    res = ["%-34s rows %6d: no stats %6.1f  rows %6.1f " % (str((B, Ci, H, W, Co, k, s)), rows, bench(nost), bench(base))]
    for reps in (1, 2, 8, 32):
        acc = torch.zeros(L.xstats_words(reps, Co), dtype=torch.int64, device="cuda")
        f = lambda: L.conv2d_xstats(BF16, x.data_ptr(), Ci, w.data_ptr(), y.data_ptr(), Co, None, acc.data_ptr(), reps, *geom, st())
        res.append(" x%d %6.1f" % (reps, bench(f)))
    print("".join(res), flush=True)
