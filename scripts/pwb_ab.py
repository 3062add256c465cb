"""One-launch 1x1 backward (csrc/pw_bwd.hip) against the launches it replaces, on the 1x1 layers of yolo_baseline at batch 32:
data gradient (+ addsrc, + fused BatchNorm-backward sums) and weight gradient (kernel + slab reduce), one stream, rotating operand sets.
usage: pwb_ab.py [iters]"""
import ctypes, os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev, bf = "cuda", torch.bfloat16
e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p(); L.event_create(ctypes.byref(e0)); L.event_create(ctypes.byref(e1))


def timeit(fn, n=iters, rounds=3):
    for _ in range(3): fn(0)
    torch.cuda.synchronize()
    ts = []
    for r in range(rounds):
        L.event_record(e0, st)
        for i in range(n): fn(i)
        L.event_record(e1, st); L.event_sync(e1)
        ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms))
        ts.append(ms.value / n * 1e3)
    return statistics.median(ts)


# (H, Cin of the layer = channels of x / dx, Cout = channels of dy, count in yolo_baseline)
SH = [(26, 512, 256, 11), (52, 256, 128, 10), (13, 1024, 512, 7), (52, 256, 256, 1), (26, 768, 256, 1), (52, 384, 128, 1), (13, 512, 256, 1),
      (13, 1024, 256, 1), (26, 256, 128, 1), (104, 128, 64, 2)]
NSET = 3
tot_old = tot_new = 0.0
for (H, N, K, cnt) in SH:
    B = 32
    M = B * H * H
    g = torch.Generator(device=dev).manual_seed(1)
    dys = [torch.randn(M, K, device=dev, generator=g).to(bf) for _ in range(NSET)]
    xs = [torch.randn(M, N, device=dev, generator=g).to(bf) for _ in range(NSET)]
    adds = [torch.randn(M, N, device=dev, generator=g).to(bf) for _ in range(NSET)]
    fys = [torch.randn(M, N, device=dev, generator=g).to(bf) for _ in range(NSET)]
    wd = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(bf)
    fsc = torch.rand(N, device=dev, generator=g) + 0.5; fsh = torch.randn(N, device=dev, generator=g) * 0.3; fmean = torch.randn(N, device=dev, generator=g) * 0.1
    dx0 = torch.zeros(M, N, device=dev, dtype=bf); dx1 = torch.zeros_like(dx0)
    dw0 = torch.zeros(K, N, device=dev); dw1 = torch.zeros_like(dw0)
    rows0 = L.conv2d_dgrad_bnsums_rows(1, B, H, H, K, H, H, N, 1, 1, 1, 0, 1, K)
    p0 = torch.zeros(max(rows0, 1), 2, N, device=dev)
    splits = L.conv2d_wgrad_splits_geom(1, B, H, H, N, H, H, K, 1, 1, 1, 0, 1, K, N)
    ws0 = torch.empty(splits * K * N, device=dev)
    slabs = L.pw_bwd_slabs(1, M, N, K, K, N, N, N, N)
    if slabs < 1:
        print("H=%d %d->%d: not eligible" % (H, N, K)); continue
    ws1 = torch.empty(slabs * K * N, device=dev)
    p1 = torch.zeros(slabs, 2, N, device=dev)

    def old(i, fused=True):
        dy, x, ad, fy = dys[i % NSET], xs[i % NSET], adds[i % NSET], fys[i % NSET]
        if fused and rows0 > 0:
            rc = L.conv2d_dgrad_bnsums(1, dy.data_ptr(), K, wd.data_ptr(), dx0.data_ptr(), N, ad.data_ptr(), N, B, H, H, K, H, H, N, 1, 1, 1, 0, 1,
                                       fy.data_ptr(), N, fsc.data_ptr(), fsh.data_ptr(), fmean.data_ptr(), 1, 0.1, p0.data_ptr(), st)
        else:
            rc = L.conv2d(1, 1, dy.data_ptr(), K, wd.data_ptr(), dx0.data_ptr(), N, None, ad.data_ptr(), N, None, B, H, H, K, H, H, N, 1, 1, 1, 0, 1, st)
        assert rc == 0, rc
        rc = L.conv2d_wgrad(1, dy.data_ptr(), K, x.data_ptr(), N, ws0.data_ptr(), splits, dw0.data_ptr(), 0, B, H, H, N, N, H, H, K, K, 1, 1, 1, 0, 1, st)
        assert rc == 0, rc

    def new(i, fused=True):
        dy, x, ad, fy = dys[i % NSET], xs[i % NSET], adds[i % NSET], fys[i % NSET]
        rc = L.pw_bwd(1, dy.data_ptr(), K, x.data_ptr(), N, wd.data_ptr(), dx1.data_ptr(), N, ad.data_ptr(), N, ws1.data_ptr(), slabs,
                      fy.data_ptr() if fused else None, N, fsc.data_ptr(), fsh.data_ptr(), fmean.data_ptr(), 1, 0.1, p1.data_ptr(), M, N, K, st)
        assert rc == 0, rc
        rc = L.wgrad_reduce(ws1.data_ptr(), slabs, dw1.data_ptr(), 0, K, K, N, N, 1, st)
        assert rc == 0, rc

    def new_k(i):
        dy, x, ad, fy = dys[i % NSET], xs[i % NSET], adds[i % NSET], fys[i % NSET]
        L.pw_bwd(1, dy.data_ptr(), K, x.data_ptr(), N, wd.data_ptr(), dx1.data_ptr(), N, ad.data_ptr(), N, ws1.data_ptr(), slabs,
                 fy.data_ptr(), N, fsc.data_ptr(), fsh.data_ptr(), fmean.data_ptr(), 1, 0.1, p1.data_ptr(), M, N, K, st)

    old(0); new(0); torch.cuda.synchronize()
    ddx = float((dx0.float() - dx1.float()).abs().max() / dx0.float().abs().max())
    ddw = float((dw0 - dw1).abs().max() / dw0.abs().max())
    ds = float((p0[:, 0].sum(0) - p1[:, 0].sum(0)).abs().max() / p0[:, 0].sum(0).abs().max()) if rows0 > 0 else float("nan")
    t_old, t_new, t_newk = timeit(old), timeit(new), timeit(new_k)
    t_old_nf, t_new_nf = timeit(lambda i: old(i, False)), timeit(lambda i: new(i, False))
    gb = (M * K + 4 * M * N) * 2 / 1e9 + slabs * K * N * 4 / 1e9
    print("H=%3d %4d->%3d x%2d  slabs %3d (old splits %3d) | fused-sums: old %.1f us  new %.1f us (kernel %.1f us = %.2f TB/s) | plain: old %.1f  new %.1f | "
          "dx %.1e dw %.1e sums %.1e" % (H, N, K, cnt, slabs, splits, t_old, t_new, t_newk, gb / t_newk * 1e3, t_old_nf, t_new_nf, ddx, ddw, ds), flush=True)
    tot_old += cnt * t_old; tot_new += cnt * t_new
print("sum over the step's 1x1 layers: old %.2f ms  new %.2f ms" % (tot_old / 1e3, tot_new / 1e3))
