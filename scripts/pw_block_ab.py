"""Fused 1x1 blocks (csrc/pw_block.hip) against the launch pairs they replace: results and timing, one process.
usage: pw_block_ab.py [iters]"""
import ctypes, os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = "cuda"
bf = torch.bfloat16
e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p(); L.event_create(ctypes.byref(e0)); L.event_create(ctypes.byref(e1))


def timeit(fn, n=iters, rounds=3):
    for _ in range(5): fn(0)
    torch.cuda.synchronize()
    ts = []
    for r in range(rounds):
        L.event_record(e0, st)
        for i in range(n): fn(i)
        L.event_record(e1, st); L.event_sync(e1)
        ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms))
        ts.append(ms.value / n * 1e3)
    return statistics.median(ts)


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


# forward shapes of yolo_baseline (K = channels of the BatchNorm output, N = the 1x1 conv's outputs), then the backward shapes of the same layers (K <-> N)
SH = [(32 * 52 * 52, 256, 128, 52), (32 * 26 * 26, 512, 256, 26), (32 * 13 * 13, 1024, 512, 13), (32 * 104 * 104, 128, 64, 104),
      (32 * 52 * 52, 128, 256, 52), (32 * 26 * 26, 256, 512, 26), (32 * 104 * 104, 64, 128, 104)]
NS = 4
for (M, K, N, H) in SH:
    B = 32
    g = torch.Generator(device=dev).manual_seed(1)
    ys = [torch.randn(M, K, device=dev, generator=g).to(bf) for _ in range(NS)]
    rs = [torch.randn(M, K, device=dev, generator=g).to(bf) for _ in range(NS)]
    scale = torch.rand(K, device=dev, generator=g) + 0.5
    shift = torch.randn(K, device=dev, generator=g) * 0.3
    w = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(bf)
    z_ref = torch.zeros(M, K, device=dev, dtype=bf); z = torch.zeros_like(z_ref)
    o_ref = torch.zeros(M, N, device=dev, dtype=bf); o = torch.zeros_like(o_ref)
    rows_ref = L.conv2d_stats_rows_geom(1, B, H, H, K, N, 1, 1, 1, 0, 1, K)
    rows = L.pw_rows(M, K)
    s_ref = torch.zeros(rows_ref, 2, N, device=dev); s = torch.zeros(rows, 2, N, device=dev)

    def ref_fwd(i):
        y, r = ys[i % NS], rs[i % NS]
        assert L.bn_act_fwd(1, y.data_ptr(), K, scale.data_ptr(), shift.data_ptr(), None, 0, None, None, r.data_ptr(), K, z_ref.data_ptr(), K, M, K, 1, 0.1, st) == 0
        assert L.conv2d(1, 0, z_ref.data_ptr(), K, w.data_ptr(), o_ref.data_ptr(), N, None, None, 0, s_ref.data_ptr(), B, H, H, K, H, H, N, 1, 1, 1, 0, 1, st) == 0

    def fus_fwd(i):
        y, r = ys[i % NS], rs[i % NS]
        assert L.pw_conv_fwd(1, y.data_ptr(), K, scale.data_ptr(), shift.data_ptr(), r.data_ptr(), K, 1, 0.1, z.data_ptr(), K, w.data_ptr(), None, o.data_ptr(), N, s.data_ptr(), M, K, N, st) == 0
    ref_fwd(0); fus_fwd(0); torch.cuda.synchronize()
    zf = torch.nn.functional.leaky_relu(ys[0].float() * scale + shift, 0.1) + rs[0].float()
    of = zf.to(bf).float() @ w.float().t()
    print("fwd M=%d K=%d N=%d: z equal %s | out vs pair %.2e, vs fp32 %.2e (pair vs fp32 %.2e) | colsum rel %.2e sq %.2e" % (
        M, K, N, bool(torch.equal(z, z_ref)), rel(o, o_ref), rel(o, of), rel(o_ref, of),
        rel(s[:, 0].sum(0), s_ref[:, 0].sum(0)), rel(s[:, 1].sum(0), s_ref[:, 1].sum(0))), flush=True)
    t_ref, t_fus = timeit(ref_fwd), timeit(fus_fwd)
    print("    time: pair %.1f us   fused %.1f us" % (t_ref, t_fus), flush=True)

    # ---- backward: layer conv1x1 (N -> K) -> BN -> act; dz [M,K], y [M,K]; dx [M,N]
    dzs = [torch.randn(M, K, device=dev, generator=g).to(bf) for _ in range(NS)]
    cA = torch.rand(K, device=dev, generator=g) + 0.5; cB = torch.randn(K, device=dev, generator=g) * 0.01; cC = torch.randn(K, device=dev, generator=g) * 0.01
    wd = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(bf)
    adds = [torch.randn(M, N, device=dev, generator=g).to(bf) for _ in range(2)]
    fys = [torch.randn(M, N, device=dev, generator=g).to(bf) for _ in range(2)]
    fsc = torch.rand(N, device=dev, generator=g) + 0.5; fsh = torch.randn(N, device=dev, generator=g) * 0.3; fmean = torch.randn(N, device=dev, generator=g) * 0.1
    dy_ref = torch.zeros(M, K, device=dev, dtype=bf); dy = torch.zeros_like(dy_ref)
    dx_ref = torch.zeros(M, N, device=dev, dtype=bf); dx = torch.zeros_like(dx_ref)
    prow_ref = L.conv2d_dgrad_bnsums_rows(1, B, H, H, K, H, H, N, 1, 1, 1, 0, 1, K)
    p_ref = torch.zeros(max(prow_ref, 1), 2, N, device=dev); pp = torch.zeros(rows, 2, N, device=dev)
    for fused_sums in (True, False):
        def ref_bwd(i):
            dzz, y = dzs[i % NS], ys[i % NS]
            assert L.bn_act_bwd_apply(1, dzz.data_ptr(), K, y.data_ptr(), K, scale.data_ptr(), shift.data_ptr(), cA.data_ptr(), cB.data_ptr(), cC.data_ptr(),
                                      dy_ref.data_ptr(), K, None, 0, None, None, None, None, None, None, 0, M, K, 1, 0.1, st) == 0
            if fused_sums:
                assert L.conv2d_dgrad_bnsums(1, dy_ref.data_ptr(), K, wd.data_ptr(), dx_ref.data_ptr(), N, adds[i % 2].data_ptr(), N, B, H, H, K, H, H, N, 1, 1, 1, 0, 1,
                                             fys[i % 2].data_ptr(), N, fsc.data_ptr(), fsh.data_ptr(), fmean.data_ptr(), 1, 0.1, p_ref.data_ptr(), st) == 0
            else:
                assert L.conv2d(1, 1, dy_ref.data_ptr(), K, wd.data_ptr(), dx_ref.data_ptr(), N, None, adds[i % 2].data_ptr(), N, None, B, H, H, K, H, H, N, 1, 1, 1, 0, 1, st) == 0

        def fus_bwd(i):
            dzz, y = dzs[i % NS], ys[i % NS]
            assert L.pw_conv_bwd(1, dzz.data_ptr(), K, y.data_ptr(), K, scale.data_ptr(), shift.data_ptr(), cA.data_ptr(), cB.data_ptr(), cC.data_ptr(), 1, 0.1,
                                 dy.data_ptr(), K, wd.data_ptr(), dx.data_ptr(), N, adds[i % 2].data_ptr(), N,
                                 fys[i % 2].data_ptr() if fused_sums else None, N, fsc.data_ptr(), fsh.data_ptr(), fmean.data_ptr(), 1, 0.1, pp.data_ptr(), M, K, N, st) == 0
        ref_bwd(0); fus_bwd(0); torch.cuda.synchronize()
        msg = "bwd%s: dy equal %s | dx vs pair %.2e" % (" +sums" if fused_sums else "      ", bool(torch.equal(dy, dy_ref)), rel(dx, dx_ref))
        if fused_sums and prow_ref > 0:
            msg += " | sum g rel %.2e  sum g(y-m) rel %.2e" % (rel(pp[:, 0].sum(0), p_ref[:, 0].sum(0)), rel(pp[:, 1].sum(0), p_ref[:, 1].sum(0)))
        print("    " + msg, flush=True)
        t_ref, t_fus = timeit(ref_bwd), timeit(fus_bwd)
        print("    time: pair %.1f us   fused %.1f us" % (t_ref, t_fus), flush=True)
