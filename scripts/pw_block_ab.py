"""Fused 1x1 forward blocks (csrc/pw_block.hip) against the launch pairs they replace: results and timing, one process (the 1x1 backward: scripts/pwb_ab.py).
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


# forward shapes of yolo_baseline (K = channels of the BatchNorm output, N = the 1x1 conv's outputs)
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

