"""-m gpu: every HIP kernel through the C ABI against a torch-CPU fp32 reference of the same op.

fp32 mode (exact-f32 MFMA) pins indexing/layout tightly; bf16 mode is checked against the same reference computed on
bf16-rounded inputs with a tolerance that covers output rounding only.
"""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from mdcv import _lib  # noqa: E402
from mdcv.engine import pad8  # noqa: E402

F32, BF16 = 0, 1
TD = {F32: torch.float32, BF16: torch.bfloat16}


def st():
    return torch.cuda.current_stream().cuda_stream


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
from variant_lib import VariantLib  # noqa: E402


def to_nhwc(x, dtype, cpad=None):
    """NCHW fp32 (cpu) -> device NHWC buffer [B,H,W,cpad] of dtype via the library kernel."""
    L = _lib.lib()
    B, C, H, W = x.shape
    cp = cpad or pad8(C)
    src = x.contiguous().cuda()
    dst = torch.empty(B, H, W, cp, dtype=TD[dtype], device="cuda")
    L.check(L.nchw_to_nhwc(dtype, src.data_ptr(), dst.data_ptr(), B, C, H, W, cp, cp, st()))
    return dst


def to_nchw(buf, dtype, C):
    L = _lib.lib()
    B, H, W, ldc = buf.shape
    out = torch.empty(B, C, H, W, dtype=torch.float32, device="cuda")
    L.check(L.nhwc_to_nchw(dtype, buf.data_ptr(), ldc, out.data_ptr(), B, C, H, W, st()))
    return out.cpu()


def rnd(dtype, t):
    return t.to(torch.bfloat16).float() if dtype == BF16 else t


def pack(dtype, w, cin_pad=None, need_d=True):
    L = _lib.lib()
    co, ci, kh, kw = w.shape
    cop, cip = pad8(co), cin_pad or pad8(ci)
    wf = torch.zeros(cop * kh * kw * cip, dtype=TD[dtype], device="cuda")
    wd = torch.zeros(cip * kh * kw * cop, dtype=TD[dtype], device="cuda") if need_d else None
    wg = w.contiguous().cuda()
    L.check(L.pack_weights(dtype, wg.data_ptr(), wf.data_ptr(), wd.data_ptr() if need_d else None, co, ci, kh, kw, cop, cip, st()))
    return wf, wd


def test_layout_roundtrip():
    x = torch.randn(2, 5, 7, 9)
    for dt in (F32, BF16):
        buf = to_nhwc(x, dt)
        assert buf.shape[-1] == 8
        ref = rnd(dt, x)
        assert torch.equal(buf.float().cpu()[..., :5], ref.permute(0, 2, 3, 1))
        assert float(buf.float()[..., 5:].abs().max()) == 0.0
        assert torch.equal(to_nchw(buf, dt, 5), ref)


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, dil, bias
    (2, 3, 20, 20, 16, 3, 1, 1, 1, False),      # input layer (Cin padded 3->8)
    (2, 16, 17, 19, 32, 3, 2, 1, 1, False),     # stride-2 downsample, odd sizes
    (3, 32, 13, 13, 64, 1, 1, 0, 1, False),     # 1x1
    (2, 64, 13, 13, 255, 1, 1, 0, 1, True),     # preyolo head: ragged Cout, bias
    (2, 3, 24, 24, 16, 7, 1, 3, 1, True),       # RektNet stem 7x7
    (2, 16, 20, 20, 32, 3, 1, 2, 2, True),      # dilated 3x3
    (1, 128, 9, 9, 7, 1, 1, 0, 1, True),        # RektNet head Cout=7
    (2, 48, 11, 11, 136, 3, 1, 1, 1, False),    # Cout > 128 (two N tiles), Cin not a multiple of 32
    (1, 256, 13, 13, 512, 3, 1, 1, 1, False),   # big K (bf16: shift-GEMM kernel, 8 channel chunks)
    (3, 64, 26, 20, 128, 3, 1, 1, 1, True),     # shift-GEMM: non-square, bias, tiles straddling images
    (2, 32, 52, 52, 256, 3, 1, 1, 1, False),    # shift-GEMM: a single channel chunk, two N tiles
    (5, 96, 8, 9, 128, 3, 1, 1, 1, False),      # shift-GEMM: tiny images (several per tile), 3 chunks
    (1, 128, 104, 104, 128, 3, 1, 1, 1, False), # shift-GEMM: wide rows (4 KiB-chunks per wave for the halo)
]


@pytest.mark.parametrize("dt", [F32, BF16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(case, dt):
    L = _lib.lib()
    B, Ci, H, W, Co, k, s, p, d, has_bias = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    b = torch.randn(Co, generator=g) if has_bias else None
    xr, wr = rnd(dt, x).requires_grad_(True), rnd(dt, w).requires_grad_(True)
    yref = F.conv2d(xr, wr, b, stride=s, padding=p, dilation=d)
    Ho, Wo = yref.shape[2], yref.shape[3]
    cip, cop = pad8(Ci), pad8(Co)
    xb = to_nhwc(x, dt)
    wf, wd = pack(dt, w)
    y = torch.empty(B, Ho, Wo, cop, dtype=TD[dt], device="cuda")
    rows = L.conv2d_stats_rows_geom(dt, B, Ho, Wo, cip, cop, k, k, s, p, d, cip)
    stats = torch.zeros(rows, 2, cop, dtype=torch.float32, device="cuda")
    bp = None
    if has_bias:
        bp = torch.zeros(cop, device="cuda")
        bp[:Co] = b.cuda()
    L.check(L.conv2d(dt, 0, xb.data_ptr(), cip, wf.data_ptr(), y.data_ptr(), cop, bp.data_ptr() if bp is not None else None, None, 0,
                     stats.data_ptr(), B, H, W, cip, Ho, Wo, cop, k, k, s, p, d, st()), "conv fwd")
    got = to_nchw(y, dt, Co)
    tol = dict(rtol=1e-4, atol=1e-4) if dt == F32 else dict(rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(got.numpy(), yref.detach().numpy(), **tol)
    if cop > Co:
        assert float(y.float()[..., Co:].abs().max()) == 0.0
    # BN statistics epilogue (computed from the fp32 accumulators)
    ssum = stats[:, 0, :Co].sum(0).cpu()
    ssq = stats[:, 1, :Co].sum(0).cpu()
    np.testing.assert_allclose(ssum.numpy(), yref.detach().sum((0, 2, 3)).numpy(), rtol=2e-3, atol=2e-2 * (B * Ho * Wo) ** 0.5)
    np.testing.assert_allclose(ssq.numpy(), (yref.detach() ** 2).sum((0, 2, 3)).numpy(), rtol=5e-3, atol=1e-2)

    # backward: random dY (pad channels zero)
    dy = torch.randn(B, Co, Ho, Wo, generator=g)
    dyr = rnd(dt, dy)
    yref.backward(dyr)
    dyb = to_nhwc(dy, dt)
    # dgrad (+ addsrc)
    add = torch.randn(B, Ci, H, W, generator=g)
    addb = to_nhwc(add, dt)
    dx = torch.empty(B, H, W, cip, dtype=TD[dt], device="cuda")
    L.check(L.conv2d(dt, 1, dyb.data_ptr(), cop, wd.data_ptr(), dx.data_ptr(), cip, None, addb.data_ptr(), cip, None,
                     B, Ho, Wo, cop, H, W, cip, k, k, s, p, d, st()), "conv dgrad")
    gotdx = to_nchw(dx, dt, Ci)
    np.testing.assert_allclose(gotdx.numpy(), (xr.grad + rnd(dt, add)).numpy(), **(dict(rtol=1e-4, atol=1e-4) if dt == F32 else dict(rtol=3e-2, atol=3e-2)))
    # wgrad
    M = B * Ho * Wo
    ktot = k * k * cip
    splits = L.conv2d_wgrad_splits(dt, M, cop, ktot)
    ws = torch.empty(splits * cop * ktot, dtype=torch.float32, device="cuda")
    dw = torch.full((Co, Ci, k, k), 7.0, dtype=torch.float32, device="cuda")
    L.check(L.conv2d_wgrad(dt, dyb.data_ptr(), cop, xb.data_ptr(), cip, ws.data_ptr(), splits, dw.data_ptr(), 0, B, H, W, cip, Ci,
                           Ho, Wo, cop, Co, k, k, s, p, d, st()), "conv wgrad")
    ref = wr.grad.numpy()
    scale = max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(dw.cpu().numpy(), ref, rtol=1e-4 if dt == F32 else 2e-2, atol=(1e-4 if dt == F32 else 2e-2) * scale)


@pytest.mark.parametrize("case", [(8, 128, 208, 208, 256, False), (8, 32, 416, 416, 64, False), (7, 32, 416, 408, 64, False), (6, 128, 208, 160, 256, True)], ids=str)
def test_stride2_dgrad_all_classes_in_one_launch(case):
    """Large stride-2 data gradients (>= 512 tiles, even sizes, bf16) run the four output-parity classes inside ONE launch (each workgroup
    walks the classes of its tile of dY positions: conv_glds_kernel ALLCLS); set_variant(16) restores the four launches.  Both == torch,
    and the two forms agree bit for bit (same products, same K order per class); with the fused BatchNorm sums too."""
    L = VariantLib()
    dt = BF16
    B, Ci, H, W, Co, fused = case
    g = torch.Generator().manual_seed(B + Ci + H)
    Ho, Wo = H // 2, W // 2
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    dy = torch.randn(B, Co, Ho, Wo, generator=g)
    add = torch.randn(B, Ci, H, W, generator=g)
    xr = torch.zeros(B, Ci, H, W, requires_grad=True)
    torch.set_num_threads(16)
    F.conv2d(xr, rnd(dt, w), None, stride=2, padding=1).backward(rnd(dt, dy))
    ref = (xr.grad + rnd(dt, add)).numpy()
    _, wd = pack(dt, w)
    dyb, addb = to_nhwc(dy, dt), to_nhwc(add, dt)
    outs, parts = {}, {}
    yb = to_nhwc(torch.randn(B, Ci, H, W, generator=g), dt) if fused else None
    coef = [torch.rand(Ci, generator=g).cuda() + 0.5 for _ in range(3)] if fused else None
    for variant in (17, 16):
        L.check(L.conv2d_set_variant(variant))
        try:
            dx = torch.full((B, H, W, Ci), 7.0, dtype=TD[dt], device="cuda")
            if fused:
                rows = L.conv2d_dgrad_bnsums_rows(dt, B, Ho, Wo, Co, H, W, Ci, 3, 3, 2, 1, 1, Co)
                assert rows > 0
                part = torch.full((rows, 2, Ci), float("nan"), dtype=torch.float32, device="cuda")
                L.check(L.conv2d_dgrad_bnsums(dt, dyb.data_ptr(), Co, wd.data_ptr(), dx.data_ptr(), Ci, addb.data_ptr(), Ci, B, Ho, Wo, Co, H, W, Ci,
                                              3, 3, 2, 1, 1, yb.data_ptr(), Ci, coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(), 1, 0.1,
                                              part.data_ptr(), st()), "dgrad bnsums")
                parts[variant] = part.sum(0).cpu().numpy()
                assert np.isfinite(parts[variant]).all()
            else:
                L.check(L.conv2d(dt, 1, dyb.data_ptr(), Co, wd.data_ptr(), dx.data_ptr(), Ci, None, addb.data_ptr(), Ci, None,
                                 B, Ho, Wo, Co, H, W, Ci, 3, 3, 2, 1, 1, st()), "dgrad")
            outs[variant] = to_nchw(dx, dt, Ci).numpy()
        finally:
            L.conv2d_set_variant(17)
    np.testing.assert_allclose(outs[17], ref, rtol=3e-2, atol=3e-2)
    np.testing.assert_array_equal(outs[17], outs[16])
    if fused:
        np.testing.assert_allclose(parts[17], parts[16], rtol=1e-4, atol=1e-2)


VARIANT_CASES = [(2, 72, 15, 17, 255, 3, 1, 1, 1, True), (2, 136, 14, 14, 144, 3, 2, 1, 1, False), (3, 256, 9, 9, 160, 1, 1, 0, 1, False),
                 (2, 96, 12, 12, 192, 3, 1, 2, 2, True)]


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11])
@pytest.mark.parametrize("dt", [F32, BF16], ids=["fp32", "bf16"])
def test_conv_tile_variants(variant, dt):
    """Every tile configuration of the wide-layer dispatch (register-staged and LDS-DMA kernels) on fwd + dgrad."""
    L = VariantLib()
    L.conv2d_set_variant(variant)
    try:
        for case in VARIANT_CASES:
            B, Ci, H, W, Co, k, s, p, d, has_bias = case
            g = torch.Generator().manual_seed(11 + Ci)
            x = torch.randn(B, Ci, H, W, generator=g)
            w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
            b = torch.randn(Co, generator=g) if has_bias else None
            xr, wr = rnd(dt, x).requires_grad_(True), rnd(dt, w)
            yref = F.conv2d(xr, wr, b, stride=s, padding=p, dilation=d)
            Ho, Wo = yref.shape[2], yref.shape[3]
            cip, cop = pad8(Ci), pad8(Co)
            xb = to_nhwc(x, dt)
            wf, wd = pack(dt, w)
            y = torch.empty(B, Ho, Wo, cop, dtype=TD[dt], device="cuda")
            stats = torch.zeros(L.conv2d_stats_rows_geom(dt, B, Ho, Wo, cip, cop, k, k, s, p, d, cip), 2, cop, dtype=torch.float32, device="cuda")
            bp = None
            if has_bias:
                bp = torch.zeros(cop, device="cuda")
                bp[:Co] = b.cuda()
            L.check(L.conv2d(dt, 0, xb.data_ptr(), cip, wf.data_ptr(), y.data_ptr(), cop, bp.data_ptr() if bp is not None else None, None, 0,
                             stats.data_ptr(), B, H, W, cip, Ho, Wo, cop, k, k, s, p, d, st()), "conv fwd")
            tol = dict(rtol=1e-4, atol=1e-4) if dt == F32 else dict(rtol=2e-2, atol=2e-2)
            np.testing.assert_allclose(to_nchw(y, dt, Co).numpy(), yref.detach().numpy(), **tol)
            np.testing.assert_allclose(stats[:, 0, :Co].sum(0).cpu().numpy(), yref.detach().sum((0, 2, 3)).numpy(), rtol=2e-3,
                                       atol=2e-2 * (B * Ho * Wo) ** 0.5)
            dy = torch.randn(B, Co, Ho, Wo, generator=g)
            yref.backward(rnd(dt, dy))
            dyb = to_nhwc(dy, dt)
            add = torch.randn(B, Ci, H, W, generator=g)
            addb = to_nhwc(add, dt)
            dx = torch.empty(B, H, W, cip, dtype=TD[dt], device="cuda")
            L.check(L.conv2d(dt, 1, dyb.data_ptr(), cop, wd.data_ptr(), dx.data_ptr(), cip, None, addb.data_ptr(), cip, None,
                             B, Ho, Wo, cop, H, W, cip, k, k, s, p, d, st()), "conv dgrad")
            np.testing.assert_allclose(to_nchw(dx, dt, Ci).numpy(), (xr.grad + rnd(dt, add)).numpy(),
                                       **(dict(rtol=1e-4, atol=1e-4) if dt == F32 else dict(rtol=3e-2, atol=3e-2)))
    finally:
        L.conv2d_set_variant(-1)


def test_conv_wgrad_many_splits_and_tiles():
    """M large enough for many pixel splits; Ktot > 128 and Cout > 128 -> several output tiles."""
    L = _lib.lib()
    dt = F32
    B, Ci, H, W, Co, k = 4, 16, 40, 40, 144, 3
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, Ci, H, W, generator=g)
    dy = torch.randn(B, Co, H, W, generator=g)
    w = torch.zeros(Co, Ci, k, k, requires_grad=True)
    F.conv2d(x, w, padding=1).backward(dy)
    xb, dyb = to_nhwc(x, dt), to_nhwc(dy, dt)
    ktot = 9 * Ci
    splits = L.conv2d_wgrad_splits(dt, B * H * W, Co, ktot)
    assert splits > 8
    ws = torch.empty(splits * Co * ktot, dtype=torch.float32, device="cuda")
    dw = torch.empty(Co, Ci, k, k, dtype=torch.float32, device="cuda")
    L.check(L.conv2d_wgrad(dt, dyb.data_ptr(), Co, xb.data_ptr(), Ci, ws.data_ptr(), splits, dw.data_ptr(), 0, B, H, W, Ci, Ci, H, W, Co, Co,
                           k, k, 1, 1, 1, st()))
    np.testing.assert_allclose(dw.cpu().numpy(), w.grad.numpy(), rtol=2e-4, atol=2e-3)


def test_conv_strided_channel_views():
    """ldc > C on input and output (route-concat slices)."""
    L = _lib.lib()
    dt = F32
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 16, 10, 10, generator=g)
    w = torch.randn(24, 16, 3, 3, generator=g) * 0.1
    ref = F.conv2d(x, w, padding=1)
    big_in = torch.zeros(2, 10, 10, 40, device="cuda")
    big_in[..., 8:24] = x.permute(0, 2, 3, 1).cuda()
    big_out = torch.full((2, 10, 10, 64), -3.0, device="cuda")
    wf, _ = pack(dt, w, need_d=False)
    L.check(L.conv2d(dt, 0, big_in.data_ptr() + 8 * 4, 40, wf.data_ptr(), big_out.data_ptr() + 16 * 4, 64, None, None, 0, None,
                     2, 10, 10, 16, 10, 10, 24, 3, 3, 1, 1, 1, st()))
    got = big_out[..., 16:40].permute(0, 3, 1, 2).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
    assert float((big_out[..., :16] + 3).abs().max()) == 0 and float((big_out[..., 40:] + 3).abs().max()) == 0


@pytest.mark.parametrize("dt", [F32, BF16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("C,act,dual,resid", [(16, 1, False, False), (64, 1, False, True), (32, 2, True, False), (128, 2, False, False), (1024, 1, False, True)])
def test_bn_act_fwd_bwd(dt, C, act, dual, resid):
    L = _lib.lib()
    B, H, W = 3, 9, 7
    M = B * H * W
    g = torch.Generator().manual_seed(C + act)
    slope = 0.1
    y1 = torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3
    y2 = torch.randn(B, C, H, W, generator=g) if dual else None
    rs = torch.randn(B, C, H, W, generator=g) if resid else None
    gam1, bet1 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    gam2, bet2 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    dout = torch.randn(B, C, H, W, generator=g)
    # reference on (rounded) inputs
    a1 = rnd(dt, y1).requires_grad_(True)
    a2 = rnd(dt, y2).requires_grad_(True) if dual else None
    p1g, p1b = gam1.clone().requires_grad_(True), bet1.clone().requires_grad_(True)
    p2g, p2b = gam2.clone().requires_grad_(True), bet2.clone().requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    pre = F.batch_norm(a1, rm, rv, p1g, p1b, training=True, momentum=0.1, eps=1e-5)
    if dual:
        pre = pre + F.batch_norm(a2, torch.zeros(C), torch.ones(C), p2g, p2b, training=True, momentum=0.1, eps=1e-5)
    z = F.leaky_relu(pre, slope) if act == 1 else F.relu(pre)
    if resid:
        z = z + rnd(dt, rs)
    z.backward(rnd(dt, dout))

    def stats_for(y, gam, bet, rmean, rvar):
        yb = to_nhwc(y, dt)
        yf = yb.float().reshape(M, C)
        partial = torch.stack((yf.sum(0), (yf * yf).sum(0))).reshape(1, 2, C).contiguous()
        accum = torch.zeros(3 * C, dtype=torch.float64, device="cuda")
        bufs = [torch.zeros(C, device="cuda") for _ in range(4)]
        gd, bd = gam.cuda(), bet.cuda()          # keep alive: the kernels are asynchronous
        L.check(L.partial_reduce(partial.data_ptr(), 1, 2, C, accum.data_ptr(), st()))
        L.check(L.bn_finalize(accum.data_ptr(), float(M), gd.data_ptr(), bd.data_ptr(), rmean.data_ptr(), rvar.data_ptr(), 0.1, 1e-5,
                              *[b.data_ptr() for b in bufs], C, st()))
        assert float(accum.abs().max()) == 0.0
        return yb, accum, bufs
    rmd, rvd = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    y1b, acc, (s1, b1, m1, i1) = stats_for(y1, gam1, bet1, rmd, rvd)
    np.testing.assert_allclose(rmd.cpu().numpy(), rm.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rvd.cpu().numpy(), rv.numpy(), rtol=1e-4, atol=1e-5)
    if dual:
        y2b, _, (s2, b2, m2, i2) = stats_for(y2, gam2, bet2, torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"))
    rsb = to_nhwc(rs, dt) if resid else None
    out = torch.empty(B, H, W, C, dtype=TD[dt], device="cuda")
    P = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    L.check(L.bn_act_fwd(dt, y1b.data_ptr(), C, s1.data_ptr(), b1.data_ptr(), P(y2b) if dual else None, C, P(s2) if dual else None,
                         P(b2) if dual else None, P(rsb), C, out.data_ptr(), C, M, C, act, slope, st()))
    tol = dict(rtol=1e-4, atol=1e-4) if dt == F32 else dict(rtol=2e-2, atol=3e-2)
    np.testing.assert_allclose(to_nchw(out, dt, C).numpy(), z.detach().numpy(), **tol)
    # backward
    db = to_nhwc(dout, dt)
    pws = torch.empty(L.bn_act_bwd_reduce_ws_floats(dt, M, C, 3 if dual else 2), device="cuda")
    L.check(L.bn_act_bwd_reduce(dt, db.data_ptr(), C, y1b.data_ptr(), C, s1.data_ptr(), b1.data_ptr(), m1.data_ptr(), i1.data_ptr(),
                                P(y2b) if dual else None, C, P(s2) if dual else None, P(b2) if dual else None, P(m2) if dual else None,
                                P(i2) if dual else None, acc.data_ptr(), pws.data_ptr(), M, C, act, slope, st()))
    outs = {}
    nsums = 3 if dual else 2
    order = [(2, gam2, m2, i2, 0)] if dual else []
    order.append((1, gam1, m1, i1, 1))
    for kx, gam, mm, ii, zero in order:
        bufs = [torch.zeros(C, device="cuda") for _ in range(5)]
        gd = gam.cuda()
        L.check(L.bn_bwd_finalize(acc.data_ptr(), kx, nsums, zero, float(M), gd.data_ptr(), mm.data_ptr(), ii.data_ptr(),
                                  *[b.data_ptr() for b in bufs], C, st()))
        torch.cuda.synchronize()
        outs[kx] = bufs
    assert float(acc.abs().max()) == 0.0
    dy1 = torch.empty(B, H, W, C, dtype=TD[dt], device="cuda")
    dy2 = torch.empty(B, H, W, C, dtype=TD[dt], device="cuda") if dual else None
    o1 = outs[1]
    o2 = outs.get(2)
    L.check(L.bn_act_bwd_apply(dt, db.data_ptr(), C, y1b.data_ptr(), C, s1.data_ptr(), b1.data_ptr(), o1[2].data_ptr(), o1[3].data_ptr(),
                               o1[4].data_ptr(), dy1.data_ptr(), C, P(y2b) if dual else None, C, P(s2) if dual else None, P(b2) if dual else None,
                               o2[2].data_ptr() if dual else None, o2[3].data_ptr() if dual else None, o2[4].data_ptr() if dual else None,
                               P(dy2), C, M, C, act, slope, st()))
    btol = dict(rtol=2e-3, atol=2e-4) if dt == F32 else dict(rtol=5e-2, atol=5e-2)
    np.testing.assert_allclose(to_nchw(dy1, dt, C).numpy(), a1.grad.numpy(), **btol)
    np.testing.assert_allclose(o1[0].cpu().numpy(), p1g.grad.numpy(), rtol=2e-3 if dt == F32 else 3e-2, atol=1e-3 if dt == F32 else 1e-1)
    np.testing.assert_allclose(o1[1].cpu().numpy(), p1b.grad.numpy(), rtol=2e-3 if dt == F32 else 3e-2, atol=1e-3 if dt == F32 else 1e-1)
    if dual:
        np.testing.assert_allclose(to_nchw(dy2, dt, C).numpy(), a2.grad.numpy(), **btol)
        np.testing.assert_allclose(o2[0].cpu().numpy(), p2g.grad.numpy(), rtol=2e-3 if dt == F32 else 3e-2, atol=1e-3 if dt == F32 else 1e-1)
    # the fused entry points (what the launch plans call) give the same vectors as the two-step path above
    rows = 37
    yf = y1b.float().reshape(M, C)
    part = torch.zeros(rows, 2, C, device="cuda")
    idx = torch.arange(M, device="cuda") % rows
    part[:, 0].index_add_(0, idx, yf)
    part[:, 1].index_add_(0, idx, yf * yf)
    fb = [torch.zeros(C, device="cuda") for _ in range(4)]
    rm2, rv2 = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    g1d, b1d, g2d = gam1.cuda(), bet1.cuda(), gam2.cuda()
    scratch = torch.zeros(3 * C, dtype=torch.float64, device="cuda")
    L.check(L.bn_stats_finalize(part.data_ptr(), rows, scratch.data_ptr(), float(M), g1d.data_ptr(), b1d.data_ptr(), rm2.data_ptr(),
                                rv2.data_ptr(), 0.1, 1e-5, *[b.data_ptr() for b in fb], C, st()))
    for got, ref in zip(fb, (s1, b1, m1, i1)):
        np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(rm2.cpu().numpy(), rmd.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv2.cpu().numpy(), rvd.cpu().numpy(), rtol=1e-5, atol=1e-6)
    f1 = [torch.zeros(C, device="cuda") for _ in range(5)]
    f2 = [torch.zeros(C, device="cuda") for _ in range(5)]
    L.check(L.bn_act_bwd_reduce_finalize(dt, db.data_ptr(), C, y1b.data_ptr(), C, s1.data_ptr(), b1.data_ptr(), m1.data_ptr(), i1.data_ptr(),
                                         P(y2b) if dual else None, C, P(s2) if dual else None, P(b2) if dual else None,
                                         P(m2) if dual else None, P(i2) if dual else None, pws.data_ptr(), M, C, act, slope, float(M),
                                         g1d.data_ptr(), *[b.data_ptr() for b in f1],
                                         g2d.data_ptr() if dual else None, *[(b.data_ptr() if dual else None) for b in f2], st()))
    for got, ref in zip(f1, o1):
        np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-6)
    if dual:
        for got, ref in zip(f2, o2):
            np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dt", [F32, BF16], ids=["fp32", "bf16"])
def test_upsample_and_colsum(dt):
    L = _lib.lib()
    x = torch.randn(2, 16, 5, 6)
    xb = to_nhwc(x, dt)
    out = torch.zeros(2, 10, 12, 40, dtype=TD[dt], device="cuda")          # into a concat slice at channel 8
    L.check(L.upsample2x_fwd(dt, xb.data_ptr(), 16, out.data_ptr() + 8 * out.element_size(), 40, 2, 5, 6, 16, st()))
    ref = F.interpolate(rnd(dt, x), scale_factor=2, mode="nearest")
    assert torch.equal(out[..., 8:24].float().permute(0, 3, 1, 2).cpu(), ref)
    d = torch.randn(2, 16, 10, 12)
    dbuf = torch.zeros(2, 10, 12, 40, dtype=TD[dt], device="cuda")
    dbuf[..., 8:24] = rnd(dt, d).permute(0, 2, 3, 1).to(TD[dt]).cuda()
    din = torch.empty(2, 5, 6, 16, dtype=TD[dt], device="cuda")
    L.check(L.upsample2x_bwd(dt, dbuf.data_ptr() + 8 * dbuf.element_size(), 40, din.data_ptr(), 16, 2, 5, 6, 16, st()))
    refd = rnd(dt, d).view(2, 16, 5, 2, 6, 2).sum((3, 5))
    np.testing.assert_allclose(to_nchw(din, dt, 16).numpy(), refd.numpy(), rtol=1e-5 if dt == F32 else 1e-2, atol=1e-5 if dt == F32 else 2e-2)
    # column sums with C = 24 (3 bf16 vectors / 6 fp32 vectors per pixel: the non-power-of-two strip path)
    t = torch.randn(3, 24, 7, 5)
    tb = to_nhwc(t, dt)
    acc = torch.zeros(24, dtype=torch.float64, device="cuda")
    o = torch.zeros(24, device="cuda")
    L.check(L.colsum(dt, tb.data_ptr(), 24, 3 * 7 * 5, 24, acc.data_ptr(), st()))
    L.check(L.accum_to_f32(acc.data_ptr(), o.data_ptr(), 24, 1, st()))
    np.testing.assert_allclose(o.cpu().numpy(), rnd(dt, t).sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)
    # atomics-free variant (partial rows + column-owner reduce), also on a tall input (many partial rows)
    for shape in ((3, 24, 7, 5), (4, 264, 40, 37)):
        t = torch.randn(*shape)
        tb = to_nhwc(t, dt)
        Cc, Mm = shape[1], shape[0] * shape[2] * shape[3]
        pws = torch.empty(L.colsum_ws_floats(dt, Mm, Cc), device="cuda")
        o2 = torch.zeros(Cc, device="cuda")
        L.check(L.colsum_f32(dt, tb.data_ptr(), Cc, Mm, Cc, pws.data_ptr(), o2.data_ptr(), st()))
        np.testing.assert_allclose(o2.cpu().numpy(), rnd(dt, t).sum((0, 2, 3)).numpy(), rtol=1e-4, atol=2e-3 * (Mm / 105) ** 0.5)


def test_adam_sgd_match_torch():
    L = _lib.lib()
    n = 10007
    g = torch.Generator().manual_seed(1)
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    for wd in (0.0, 0.01):
        p = torch.nn.Parameter(p0.clone())
        opt = torch.optim.Adam([p], lr=1e-3, weight_decay=wd)
        pd, m, v = p0.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        gd = gr.cuda()
        for step in (1, 2, 3):
            p.grad = gr.clone()
            opt.step()
            L.check(L.adam_step(pd.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, step, 1e-3, 0.9, 0.999, 1e-8, wd, 1.0, st()))
        np.testing.assert_allclose(pd.cpu().numpy(), p.detach().numpy(), rtol=1e-5, atol=1e-6)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([p], lr=1e-2, momentum=0.9)
    pd, buf = p0.clone().cuda(), torch.zeros(n, device="cuda")
    for step in (1, 2, 3):
        p.grad = gr.clone()
        opt.step()
        L.check(L.sgd_step(pd.data_ptr(), gd.data_ptr(), buf.data_ptr(), n, step, 1e-2, 0.9, 0.0, 1.0, st()))
    np.testing.assert_allclose(pd.cpu().numpy(), p.detach().numpy(), rtol=1e-5, atol=1e-6)


def test_device_is_gfx950():
    L = _lib.lib()
    cu, wave = ctypes.c_int(), ctypes.c_int()
    hbm = ctypes.c_longlong()
    arch = ctypes.create_string_buffer(64)
    L.check(L.device_info(ctypes.byref(cu), ctypes.byref(wave), ctypes.byref(hbm), arch, 64))
    assert wave.value == 64 and cu.value >= 200 and arch.value.decode().startswith("gfx950")


@pytest.mark.parametrize("dt", [F32, BF16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("stride,H,W", [(2, 26, 26), (2, 8, 12), (1, 13, 13), (1, 6, 9)])
def test_maxpool2x2(dt, stride, H, W):
    L = _lib.lib()
    B, C = 2, 24
    g = torch.Generator().manual_seed(H * 10 + stride)
    x = torch.randn(B, C, H, W, generator=g)
    xr = rnd(dt, x).requires_grad_(True)
    ref = F.max_pool2d(F.pad(xr, (0, 1, 0, 1)), 2, 1, 0) if stride == 1 else F.max_pool2d(xr, 2, 2, 0)
    Ho, Wo = ref.shape[2], ref.shape[3]
    xb = to_nhwc(x, dt)
    out = torch.empty(B, Ho, Wo, C, dtype=TD[dt], device="cuda")
    idx = torch.empty(B * Ho * Wo * C, dtype=torch.uint8, device="cuda")
    L.check(L.maxpool2x2_fwd(dt, xb.data_ptr(), C, out.data_ptr(), C, idx.data_ptr(), B, H, W, C, stride, st()))
    assert torch.equal(to_nchw(out, dt, C), ref.detach())
    d = torch.randn(B, C, Ho, Wo, generator=g)
    ref.backward(rnd(dt, d))
    db = to_nhwc(d, dt)
    din = torch.empty(B, H, W, C, dtype=TD[dt], device="cuda")
    L.check(L.maxpool2x2_bwd(dt, db.data_ptr(), C, idx.data_ptr(), din.data_ptr(), C, B, H, W, C, stride, st()))
    np.testing.assert_allclose(to_nchw(din, dt, C).numpy(), xr.grad.numpy(), rtol=1e-6 if dt == F32 else 1e-2, atol=1e-6 if dt == F32 else 2e-2)


FIRST_CONV_CASES = [(2, 3, 416, 416, 1), (3, 3, 37, 53, 1), (1, 3, 5, 16, 0), (2, 1, 30, 95, 2), (1, 3, 64, 640, 1)]   # B, Cin, H, W, activation


@pytest.mark.parametrize("case", FIRST_CONV_CASES, ids=[str(c) for c in FIRST_CONV_CASES])
def test_first_conv_two_streaming_passes(case):
    """mdcv_first_conv_stats + mdcv_bn_stats_finalize + mdcv_first_conv_bn_act (the first conv -> BatchNorm -> activation without re-reading the layer's output)
    against F.conv2d on the bf16-rounded operands (y within one bf16 step of the float64 result, z likewise through the same scale / shift), against the
    generic path mdcv_conv2d(+statistics) + finalize + mdcv_bn_act_fwd (same statistics to fp32 summation order, same y / z up to bf16 roundings of
    accumulators that differ in the last fp32 bits), ragged widths / heights (not multiples of 16 / 4), every activation."""
    L = _lib.lib()
    dt = BF16
    B, Ci, H, W, act = case
    Co = 32
    g = torch.Generator().manual_seed(B * 7 + H + W)
    x = torch.rand(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    xb = to_nhwc(x, dt)                                              # [B, H, W, 8]
    wf, _ = pack(dt, w, need_d=False)
    assert L.first_conv_ok(dt, B, H, W, 8, Co, 3, 3, 1, 1, 1, 8) == 1
    assert L.first_conv_ok(dt, B, H, W, 16, Co, 3, 3, 1, 1, 1, 16) == 0 and L.first_conv_ok(F32, B, H, W, 8, Co, 3, 3, 1, 1, 1, 8) == 0
    M = B * H * W
    gamma = (torch.rand(Co, generator=g) + 0.5).cuda(); beta = (torch.randn(Co, generator=g) * 0.3).cuda()
    slope = 0.1

    def finalize(partial, rows):
        acc = torch.zeros(2 * Co, dtype=torch.float64, device="cuda")
        rm, rv = torch.zeros(Co, device="cuda"), torch.ones(Co, device="cuda")
        out = [torch.zeros(Co, device="cuda") for _ in range(4)]     # scale, shift, mean, invstd
        L.check(L.bn_stats_finalize(partial.data_ptr(), rows, acc.data_ptr(), float(M), gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                    0.1, 1e-5, *[o.data_ptr() for o in out], Co, st()))
        return out
    # the two-pass form
    rows = L.first_conv_rows(B, H)
    part = torch.full((rows, 2, Co), float("nan"), device="cuda")
    L.check(L.first_conv_stats(dt, xb.data_ptr(), 8, wf.data_ptr(), part.data_ptr(), B, H, W, st()), "first_conv_stats")
    sc1, sh1, mean1, is1 = finalize(part, rows)
    ldz = 40                                                         # z inside a wider buffer (a concat slice)
    y1 = torch.zeros(B, H, W, Co, dtype=TD[dt], device="cuda")
    z1 = torch.zeros(B, H, W, ldz, dtype=TD[dt], device="cuda")
    L.check(L.first_conv_bn_act(dt, xb.data_ptr(), 8, wf.data_ptr(), sc1.data_ptr(), sh1.data_ptr(), act, slope, y1.data_ptr(), Co, z1.data_ptr(), ldz,
                                B, H, W, st()), "first_conv_bn_act")
    # the generic path
    rows0 = L.conv2d_stats_rows_geom(dt, B, H, W, 8, Co, 3, 3, 1, 1, 1, 8)
    part0 = torch.zeros(rows0, 2, Co, device="cuda")
    y0 = torch.zeros(B, H, W, Co, dtype=TD[dt], device="cuda")
    L.check(L.conv2d(dt, 0, xb.data_ptr(), 8, wf.data_ptr(), y0.data_ptr(), Co, None, None, 0, part0.data_ptr(), B, H, W, 8, H, W, Co, 3, 3, 1, 1, 1, st()))
    sc0, sh0, mean0, is0 = finalize(part0, rows0)
    z0 = torch.zeros(B, H, W, Co, dtype=TD[dt], device="cuda")
    L.check(L.bn_act_fwd(dt, y0.data_ptr(), Co, sc0.data_ptr(), sh0.data_ptr(), None, 0, None, None, None, 0, z0.data_ptr(), Co, M, Co, act, slope, st()))
    torch.cuda.synchronize()
    assert not bool(torch.isnan(part).any())
    # float64 reference on the rounded operands
    ref = F.conv2d(rnd(dt, x).double(), rnd(dt, w).double(), None, 1, 1).permute(0, 2, 3, 1)          # [B, H, W, Co]
    yf1 = y1.double().cpu()
    tol = 2 ** -8 * np.maximum(np.abs(ref.numpy()), 2 ** -6)                                            # one bf16 rounding step of the result
    assert bool((np.abs(yf1.numpy() - ref.numpy()) <= tol).all()), float(np.abs(yf1.numpy() - ref.numpy()).max())
    mean_ref, var_ref = ref.reshape(-1, Co).mean(0), ref.reshape(-1, Co).var(0, unbiased=False)
    np.testing.assert_allclose(mean1.cpu().numpy(), mean_ref.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(is1.cpu().numpy(), (1.0 / np.sqrt(var_ref.numpy() + 1e-5)), rtol=1e-4)
    for a_, b_, name in ((sc1, sc0, "scale"), (sh1, sh0, "shift"), (mean1, mean0, "mean"), (is1, is0, "invstd")):
        np.testing.assert_allclose(a_.cpu().numpy(), b_.cpu().numpy(), rtol=2e-5, atol=2e-6, err_msg=name)
    # y / z against the generic path: the accumulators differ in their last fp32 bits (K order), so a bf16 rounding may flip on a few elements
    dy = (y1.float() - y0.float()).abs()
    assert float(dy.max()) <= 2 ** -7 * max(1e-6, float(y0.float().abs().max())) and float((dy > 0).float().mean()) < 2e-2
    zz = z1[..., :Co].float()
    assert float(z1[..., Co:].float().abs().max()) == 0.0                                             # nothing written beside the 32 channels
    pre = y1.float() * sc1 + sh1
    zr = pre if act == 0 else torch.where(pre > 0, pre, pre * (slope if act == 1 else 0.0))
    assert float((zz - zr).abs().max()) <= 2 ** -7 * max(1e-6, float(zr.abs().max()))                 # z is act(scale * y_as_stored + shift), rounded to bf16
    dz = (zz - z0.float()).abs()
    assert float(dz.max()) <= 2 ** -6 * max(1e-6, float(z0.float().abs().max())) and float((dz > 0).float().mean()) < 5e-2


FUSE_CASES = [
    # B, Cin(conv input = channels of the BatchNorm), H, W, Cout, k, stride, pad
    (2, 64, 26, 20, 128, 3, 1, 1),      # shift kernel, 256-row tiles
    (5, 128, 13, 13, 256, 3, 1, 1),     # shift kernel, 128-row tiles, images straddling tiles
    (2, 256, 14, 14, 128, 1, 1, 0),     # 1x1, 128x128 tiles
    (2, 64, 17, 19, 128, 3, 2, 1),      # stride-2 data gradient: four parity-class launches
    (3, 32, 20, 20, 48, 3, 1, 1),       # narrow conv input: 128x32 tile
    (2, 16, 24, 24, 32, 3, 1, 2),       # dilated (pad 2, dil 2 below), 128x16 tile
    (2, 64, 15, 17, 72, 3, 1, 1),       # K (= Cout_pad 72) not a multiple of 32: generic address path
]


@pytest.mark.parametrize("dt", [F32, BF16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", FUSE_CASES, ids=[str(c) for c in FUSE_CASES])
@pytest.mark.parametrize("act,with_add", [(1, True), (0, False), (2, True)])
def test_dgrad_with_fused_bn_sums(case, dt, act, with_add):
    """mdcv_conv2d_dgrad_bnsums == mdcv_conv2d(mode 1) + mdcv_bn_act_bwd_reduce + mdcv_bn_bwd_finalize (same dz, same vectors)."""
    L = _lib.lib()
    B, Ci, H, W, Co, k, s, p = case
    d = 2 if (k == 3 and p == 2) else 1
    g = torch.Generator().manual_seed(Ci * 7 + Co + act)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    Ho = (H + 2 * p - d * (k - 1) - 1) // s + 1
    Wo = (W + 2 * p - d * (k - 1) - 1) // s + 1
    cip, cop = pad8(Ci), pad8(Co)
    wf, wd = pack(dt, w)
    dyb = to_nhwc(torch.randn(B, Co, Ho, Wo, generator=g), dt)
    addb = to_nhwc(torch.randn(B, Ci, H, W, generator=g), dt) if with_add else None
    yb = to_nhwc(torch.randn(B, Ci, H, W, generator=g) * 1.3 + 0.2, dt)          # raw conv output of the producer layer
    M = B * H * W
    scale = (torch.rand(cip, generator=g) + 0.5).cuda(); shift = (torch.randn(cip, generator=g) * 0.3).cuda()
    mean = (torch.randn(cip, generator=g) * 0.2 + 0.2).cuda(); invstd = (torch.rand(cip, generator=g) + 0.5).cuda()
    gamma = (torch.rand(cip, generator=g) + 0.5).cuda()
    slope = 0.1
    rows = L.conv2d_dgrad_bnsums_rows(dt, B, Ho, Wo, cop, H, W, cip, k, k, s, p, d, cop)
    if dt == F32:                                   # bf16 only: the query says so and the entry point refuses
        assert rows == 0
        return
    assert rows > 0
    P = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    # reference path
    dx0 = torch.empty(B, H, W, cip, dtype=TD[dt], device="cuda")
    L.check(L.conv2d(dt, 1, dyb.data_ptr(), cop, wd.data_ptr(), dx0.data_ptr(), cip, None, P(addb), cip, None,
                     B, Ho, Wo, cop, H, W, cip, k, k, s, p, d, st()), "dgrad")
    acc = torch.zeros(3 * cip, dtype=torch.float64, device="cuda")
    pws = torch.empty(L.bn_act_bwd_reduce_ws_floats(dt, M, cip, 2), device="cuda")
    L.check(L.bn_act_bwd_reduce(dt, dx0.data_ptr(), cip, yb.data_ptr(), cip, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                invstd.data_ptr(), None, 0, None, None, None, None, acc.data_ptr(), pws.data_ptr(), M, cip, act, slope, st()))
    ref = [torch.zeros(cip, device="cuda") for _ in range(5)]
    L.check(L.bn_bwd_finalize(acc.data_ptr(), 1, 2, 1, float(M), gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                              *[b.data_ptr() for b in ref], cip, st()))
    # fused path
    dx1 = torch.empty(B, H, W, cip, dtype=TD[dt], device="cuda")
    part = torch.full((rows, 2, cip), float("nan"), device="cuda")
    L.check(L.conv2d_dgrad_bnsums(dt, dyb.data_ptr(), cop, wd.data_ptr(), dx1.data_ptr(), cip, P(addb), cip, B, Ho, Wo, cop, H, W, cip,
                                  k, k, s, p, d, yb.data_ptr(), cip, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), act, slope,
                                  part.data_ptr(), st()), "fused dgrad")
    got = [torch.zeros(cip, device="cuda") for _ in range(5)]
    L.check(L.bn_bwd_finalize_rows(part.data_ptr(), rows, cip, float(M), gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                   *[b.data_ptr() for b in got], st()))
    torch.cuda.synchronize()
    assert torch.equal(dx0, dx1)                                   # same kernel arithmetic, same stored gradient
    assert not bool(torch.isnan(part[:, :, :Ci]).any())            # every row the query promised was written
    for a, b, name in zip(got, ref, ("dgamma", "dbeta", "cA", "cB", "cC")):
        an, bn_ = a.cpu().numpy()[:Ci], b.cpu().numpy()[:Ci]
        scale_ = max(1.0, float(np.abs(bn_).max()))
        np.testing.assert_allclose(an, bn_, rtol=2e-4, atol=2e-4 * scale_, err_msg=name)


@pytest.mark.parametrize("variant", [-8, -9, -12, -30, -31, -201])
def test_shift_tile_plans(variant):
    """Every tuning plan of the 3x3 shift kernel (256 / 128 / mixed rows, 192-row tiles; K loop in lockstep (-30) or with ping-pong wave
    groups on every forward launch (-31); 384-row ping-pong tiles forced (-201)) gives the forward result and the same BatchNorm
    statistics as the default plan."""
    L = VariantLib()
    dt = BF16
    g = torch.Generator().manual_seed(5)
    outs = {}
    for v in (-7, variant):
        L.conv2d_set_variant(v)
        try:
            for case in ((4, 64, 26, 26, 256), (32, 128, 13, 13, 128), (9, 32, 30, 17, 128)):
                B, Ci, H, W, Co = case
                gg = torch.Generator().manual_seed(B + Ci)
                x = torch.randn(B, Ci, H, W, generator=gg)
                w = torch.randn(Co, Ci, 3, 3, generator=gg) / (Ci * 9) ** 0.5
                xb = to_nhwc(x, dt)
                wf, wd = pack(dt, w)
                y = torch.empty(B, H, W, Co, dtype=TD[dt], device="cuda")
                rows = L.conv2d_stats_rows_geom(dt, B, H, W, Ci, Co, 3, 3, 1, 1, 1, Ci)
                stats = torch.zeros(rows, 2, Co, device="cuda")
                L.check(L.conv2d(dt, 0, xb.data_ptr(), Ci, wf.data_ptr(), y.data_ptr(), Co, None, None, 0, stats.data_ptr(),
                                 B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 1, st()), "conv")
                dx = torch.empty(B, H, W, Ci, dtype=TD[dt], device="cuda")
                dyb = to_nhwc(torch.randn(B, Co, H, W, generator=gg), dt)
                L.check(L.conv2d(dt, 1, dyb.data_ptr(), Co, wd.data_ptr(), dx.data_ptr(), Ci, None, None, 0, None,
                                 B, H, W, Co, H, W, Ci, 3, 3, 1, 1, 1, st()), "dgrad")
                torch.cuda.synchronize()
                outs.setdefault(v, []).append((y.clone(), stats.sum(0).clone(), dx.clone()))
        finally:
            L.conv2d_set_variant(-7); L.conv2d_set_variant(-32); L.conv2d_set_variant(-200)
    for (y0, s0, d0), (y1, s1, d1) in zip(outs[-7], outs[variant]):
        assert torch.equal(y0, y1) and torch.equal(d0, d1)
        np.testing.assert_allclose(s1.cpu().numpy(), s0.cpu().numpy(), rtol=1e-4, atol=1e-2)


PW_CASES = [  # (M, K, N, extra channel stride, resid, act)
    (64 * 21 + 17, 256, 128, 0, True, 1), (64 * 9, 128, 64, 8, True, 1), (3000, 64, 128, 0, False, 1), (32 * 40 + 5, 512, 256, 16, True, 1),
    (16 * 70 + 3, 1024, 512, 0, True, 1), (2000, 256, 24, 0, False, 2), (1100, 128, 256, 8, True, 0), (70000, 256, 128, 0, True, 1)]


@pytest.mark.parametrize("case", PW_CASES, ids=[str(c) for c in PW_CASES])
def test_pw_block_forward(case):
    """mdcv_pw_conv_fwd (BatchNorm-apply + activation (+ residual) folded into the operand load of a 1x1 conv) == the pair mdcv_bn_act_fwd +
    mdcv_conv2d bit for bit (z, out), == a torch fp32 reference within bf16 rounding; its partial statistics sum to the pair's."""
    L = _lib.lib()
    M, K, N, xs, with_r, act = case
    g = torch.Generator().manual_seed(M + K + N)
    ldy, ldz, ldo = K + xs, K + 2 * xs, N + xs
    y = (torch.randn(M, ldy, generator=g) * 1.5).to(torch.bfloat16).cuda()
    r = torch.randn(M, ldy, generator=g).to(torch.bfloat16).cuda() if with_r else None
    scale = (torch.rand(K, generator=g) + 0.5).cuda(); shift = (torch.randn(K, generator=g) * 0.3).cuda()
    w = torch.randn(N, K, 1, 1, generator=g) / K ** 0.5
    wf, _ = pack(BF16, w, need_d=False)
    bias = (torch.randn(pad8(N), generator=g) * 0.1).cuda() if act == 2 else None
    P = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    z0 = torch.zeros(M, ldz, dtype=torch.bfloat16, device="cuda"); z1 = torch.zeros_like(z0)
    o0 = torch.zeros(M, ldo, dtype=torch.bfloat16, device="cuda"); o1 = torch.zeros_like(o0)
    rows0 = L.conv2d_stats_rows_geom(BF16, 1, M, 1, K, N, 1, 1, 1, 0, 1, ldz)
    rows1 = L.pw_rows(M, K)
    s0 = torch.zeros(rows0, 2, N, device="cuda"); s1 = torch.full((rows1, 2, N), float("nan"), device="cuda")
    L.check(L.bn_act_fwd(BF16, y.data_ptr(), ldy, scale.data_ptr(), shift.data_ptr(), None, 0, None, None, P(r), ldy, z0.data_ptr(), ldz, M, K,
                         act, 0.1, st()))
    L.check(L.conv2d(BF16, 0, z0.data_ptr(), ldz, wf.data_ptr(), o0.data_ptr(), ldo, P(bias), None, 0, s0.data_ptr(), 1, M, 1, K, M, 1, N,
                     1, 1, 1, 0, 1, st()))
    L.check(L.pw_conv_fwd(BF16, y.data_ptr(), ldy, scale.data_ptr(), shift.data_ptr(), P(r), ldy, act, 0.1, z1.data_ptr(), ldz, wf.data_ptr(),
                          P(bias), o1.data_ptr(), ldo, s1.data_ptr(), M, K, N, st()), "pw_conv_fwd")
    torch.cuda.synchronize()
    assert torch.equal(z0, z1)
    assert torch.equal(o0[:, :N], o1[:, :N]) and float(o1[:, N:].abs().max() if ldo > N else 0) == 0
    assert not bool(torch.isnan(s1).any())
    np.testing.assert_allclose(s1.sum(0).cpu().numpy(), s0.sum(0).cpu().numpy(), rtol=2e-4, atol=2e-2)
    zf = y[:, :K].float().cpu() * scale.cpu() + shift.cpu()
    zf = zf if act == 0 else torch.where(zf > 0, zf, zf * (0.1 if act == 1 else 0.0))
    if with_r:
        zf = zf + r[:, :K].float().cpu()
    assert float((z1[:, :K].float().cpu() - zf).abs().max()) <= 2 ** -7 * float(zf.abs().max())
    of = z1[:, :K].float().cpu() @ w.reshape(N, K).to(torch.bfloat16).float().t() + (bias[:N].cpu() if bias is not None else 0)
    assert float((o1[:, :N].float().cpu() - of).abs().max()) <= 1e-2 * float(of.abs().max())
    # the same launch with its statistics added to exact accumulators (mdcv_pw_conv_fwd_xstats): out / z bit-identical, the digits hold the exact
    # sum of the rows the launch above wrote (float64 of <= 2^17 fp32 rows is exact enough to compare at 1e-12 relative)
    if N % 8 == 0:
        reps = L.xstats_reps(rows1, N)
        acc = torch.zeros(L.xstats_words(reps, N), dtype=torch.int64, device="cuda")
        z2 = torch.zeros_like(z0); o2 = torch.zeros_like(o0)
        L.check(L.pw_conv_fwd_xstats(BF16, y.data_ptr(), ldy, scale.data_ptr(), shift.data_ptr(), P(r), ldy, act, 0.1, z2.data_ptr(), ldz, wf.data_ptr(),
                                     P(bias), o2.data_ptr(), ldo, acc.data_ptr(), reps, M, K, N, st()), "pw_conv_fwd_xstats")
        torch.cuda.synchronize()
        assert torch.equal(z2, z1) and torch.equal(o2, o1)
        d = acc.reshape(reps, 3, 2, N).sum(0).double()
        tot = d[0] * 2.0 ** -70 + d[1] * 2.0 ** -30 + d[2] * 2.0 ** 10
        np.testing.assert_allclose(tot.cpu().numpy(), s1.double().sum(0).cpu().numpy(), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("case", [(2, 3, 40, 56, 32, 3, 1, 1, 1), (3, 3, 33, 47, 16, 3, 1, 1, 1), (2, 5, 30, 30, 32, 3, 2, 1, 1), (1, 8, 64, 40, 24, 1, 1, 0, 0)], ids=str)
def test_wgrad_with_bn_apply_in_the_operand_load(case):
    """mdcv_conv2d_wgrad_bnapply (a first layer's weight gradient straight from dz and y: the BatchNorm-backward apply formed in LDS, rounded to bf16
    like the apply pass) == mdcv_bn_act_bwd_apply + mdcv_conv2d_wgrad BIT FOR BIT, and == F.conv2d autograd on the float64 dy within bf16 rounding;
    ragged channel counts (16 / 24 real of 16 / 24 / 32 padded), ragged last pixel tiles and splits, stride 2, a 1x1 kernel, ReLU / leaky / none."""
    L = _lib.lib()
    dt = BF16
    B, Ci, H, W, Co, k, stride, pad, act = case
    g = torch.Generator().manual_seed(B + Ci + W + Co)
    cip, cop = pad8(Ci), pad8(Co)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    M = B * Ho * Wo
    x = torch.randn(B, Ci, H, W, generator=g)
    xb = to_nhwc(x, dt)
    dz = to_nhwc(torch.randn(B, Co, Ho, Wo, generator=g), dt)
    y = to_nhwc(torch.randn(B, Co, Ho, Wo, generator=g) * 1.5, dt)
    scale = (torch.rand(Co, generator=g) + 0.5).cuda(); shift = (torch.randn(Co, generator=g) * 0.3).cuda()
    cA = (torch.rand(Co, generator=g) + 0.5).cuda(); cB = (torch.randn(Co, generator=g) * 0.05).cuda(); cC = (torch.randn(Co, generator=g) * 0.05).cuda()
    geom = (B, H, W, cip, Ho, Wo, cop, k, k, stride, pad, 1)
    assert L.conv2d_wgrad_bnapply_ok(dt, *geom, cop, cop, cip) == 1
    splits = L.conv2d_wgrad_splits_geom(dt, *geom, cop, cip)
    # the two-launch form
    dy = torch.zeros(B, Ho, Wo, cop, dtype=torch.bfloat16, device="cuda")
    L.check(L.bn_act_bwd_apply(dt, dz.data_ptr(), cop, y.data_ptr(), cop, scale.data_ptr(), shift.data_ptr(), cA.data_ptr(), cB.data_ptr(), cC.data_ptr(),
                               dy.data_ptr(), cop, None, 0, None, None, None, None, None, None, 0, M, Co, act, 0.1, st()))
    ws0 = torch.empty(splits * cop * k * k * cip, device="cuda")
    dw0 = torch.full((Co, Ci, k, k), 7.0, device="cuda")
    L.check(L.conv2d_wgrad(dt, dy.data_ptr(), cop, xb.data_ptr(), cip, ws0.data_ptr(), splits, dw0.data_ptr(), 0, B, H, W, cip, Ci, Ho, Wo, cop, Co,
                           k, k, stride, pad, 1, st()), "wgrad")
    # one launch
    ws1 = torch.full((splits * cop * k * k * cip,), float("nan"), device="cuda")
    dw1 = torch.full((Co, Ci, k, k), 7.0, device="cuda")
    L.check(L.conv2d_wgrad_bnapply(dt, dz.data_ptr(), cop, y.data_ptr(), cop, scale.data_ptr(), shift.data_ptr(), cA.data_ptr(), cB.data_ptr(),
                                   cC.data_ptr(), act, 0.1, xb.data_ptr(), cip, ws1.data_ptr(), splits, dw1.data_ptr(), 0, B, H, W, cip, Ci, Ho, Wo,
                                   cop, Co, k, k, stride, pad, 1, st()), "wgrad_bnapply")
    torch.cuda.synchronize()
    assert torch.equal(dw0, dw1), float((dw0 - dw1).abs().max() / dw0.abs().max())
    # float64 reference from the bf16-rounded operands
    dzf, yf = dz[..., :Co].double().cpu(), y[..., :Co].double().cpu()
    pre = y[..., :Co].float().cpu() * scale.cpu() + shift.cpu()
    slope = {0: 1.0, 1: 0.1, 2: 0.0}[act]
    gg = dzf * torch.where(pre > 0, torch.ones_like(dzf), torch.full_like(dzf, float(np.float32(slope)))) if act else dzf
    dyf = (cA.double().cpu() * gg + cB.double().cpu() * yf + cC.double().cpu()).permute(0, 3, 1, 2).contiguous()
    wz = torch.zeros(Co, Ci, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(xb[..., :Ci].double().cpu().permute(0, 3, 1, 2), wz, None, stride, pad).backward(dyf)
    ref = wz.grad.numpy()
    np.testing.assert_allclose(dw1.cpu().numpy(), ref, rtol=0, atol=3e-3 * max(1.0, float(np.abs(ref).max())))
    assert L.conv2d_wgrad_bnapply_ok(dt, B, H, W, 64, Ho, Wo, 64, k, k, stride, pad, 1, 64, 64, 64) == 0        # wide layers keep their apply pass


PWB_CASES = [  # (M, Cout = channels of dy, real Cout, Cin = channels of x / dx, extra channel stride, addsrc)
    (32 * 21 + 17, 256, 256, 512, 0, True), (5408, 512, 512, 1024, 0, True), (3000, 128, 128, 256, 8, False), (2703, 256, 255, 256, 0, False),
    (700, 64, 64, 128, 16, True), (86528, 128, 128, 256, 0, True), (1100, 256, 256, 768, 8, True), (96, 128, 128, 64, 0, False)]


@pytest.mark.parametrize("fused", [False, True], ids=["plain", "bnsums"])
@pytest.mark.parametrize("case", PWB_CASES, ids=[str(c) for c in PWB_CASES])
def test_pw_bwd_one_launch(case, fused):
    """mdcv_pw_bwd (data gradient + weight-gradient slabs of a 1x1 conv in one launch, optional addsrc and fused BatchNorm-backward sums of the
    producer layer) + mdcv_wgrad_reduce against F.conv2d autograd on the bf16-rounded operands (dx: bf16 output rounding, 2e-2 of the
    largest element; dW: fp32 accumulation, 1e-4 of the largest element) and against the launches it replaces (mdcv_conv2d mode 1,
    mdcv_conv2d_wgrad); ragged Cout (255 of 256), channel strides wider than the tensors, ragged last tiles and slabs."""
    L = _lib.lib()
    M, K, Kr, N, xs, with_add = case
    g = torch.Generator().manual_seed(M + K + N + 7)
    ldk, ldn = K + xs, N + xs
    dyf = torch.randn(M, ldk, generator=g)
    dyf[:, Kr:] = 0.0                                                # pad lanes of the activations are exact zeros
    dy = dyf.to(torch.bfloat16).cuda()
    x = torch.randn(M, ldn, generator=g).to(torch.bfloat16).cuda()
    w = torch.randn(Kr, N, 1, 1, generator=g) / K ** 0.5             # the layer's weight [Cout][Cin]
    _, wd = pack(BF16, w)
    add = torch.randn(M, ldn, generator=g).to(torch.bfloat16).cuda() if with_add else None
    fy = (torch.randn(M, ldn, generator=g) * 1.3 + 0.2).to(torch.bfloat16).cuda()
    fsc = (torch.rand(N, generator=g) + 0.5).cuda(); fsh = (torch.randn(N, generator=g) * 0.3).cuda()
    fmean = (torch.randn(N, generator=g) * 0.2 + 0.2).cuda(); finv = (torch.rand(N, generator=g) + 0.5).cuda(); gamma = (torch.rand(N, generator=g) + 0.5).cuda()
    P = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    slabs = L.pw_bwd_slabs(BF16, M, N, K, ldk, ldn, ldn, ldn if with_add else 8, ldn if fused else 8)
    assert slabs >= 1
    ws = torch.full((slabs, K, N), float("nan"), device="cuda")
    part = torch.full((slabs, 2, N), float("nan"), device="cuda")
    dx1 = torch.zeros(M, ldn, dtype=torch.bfloat16, device="cuda")
    dw1 = torch.full((Kr, N, 1, 1), 7.0, device="cuda")
    L.check(L.pw_bwd(BF16, dy.data_ptr(), ldk, x.data_ptr(), ldn, wd.data_ptr(), dx1.data_ptr(), ldn, P(add), ldn, ws.data_ptr(), slabs,
                     fy.data_ptr() if fused else None, ldn, fsc.data_ptr(), fsh.data_ptr(), fmean.data_ptr(), 1, 0.1, part.data_ptr(), M, N, K, st()), "pw_bwd")
    L.check(L.wgrad_reduce(ws.data_ptr(), slabs, dw1.data_ptr(), 0, K, Kr, N, N, 1, st()), "wgrad_reduce")
    torch.cuda.synchronize()
    assert not bool(torch.isnan(ws).any())
    # the launches it replaces
    dx0 = torch.zeros(M, ldn, dtype=torch.bfloat16, device="cuda")
    L.check(L.conv2d(BF16, 1, dy.data_ptr(), ldk, wd.data_ptr(), dx0.data_ptr(), ldn, None, P(add), ldn, None, 1, M, 1, K, M, 1, N, 1, 1, 1, 0, 1, st()))
    splits = L.conv2d_wgrad_splits_geom(BF16, 1, M, 1, N, M, 1, K, 1, 1, 1, 0, 1, ldk, ldn)
    ws0 = torch.empty(splits * K * N, device="cuda")
    dw0 = torch.zeros(Kr, N, 1, 1, device="cuda")
    L.check(L.conv2d_wgrad(BF16, dy.data_ptr(), ldk, x.data_ptr(), ldn, ws0.data_ptr(), splits, dw0.data_ptr(), 0, 1, M, 1, N, N, M, 1, K, Kr, 1, 1, 1, 0, 1, st()))
    torch.cuda.synchronize()
    # torch autograd on the same (bf16-rounded) operands
    xt = x[:, :N].float().cpu().t().reshape(1, N, 1, M)
    wt = w.to(torch.bfloat16).float().requires_grad_(True)
    xt.requires_grad_(True)
    F.conv2d(xt, wt).backward(dy[:, :Kr].float().cpu().t().reshape(1, Kr, 1, M))
    dxf = xt.grad.reshape(N, M).t() + (add[:, :N].float().cpu() if with_add else 0)
    wz = torch.zeros(Kr, N, 1, 1, requires_grad=True)
    F.conv2d(xt.detach(), wz).backward(dy[:, :Kr].float().cpu().t().reshape(1, Kr, 1, M))
    assert float((dx1[:, :N].float().cpu() - dxf).abs().max()) <= 2e-2 * float(dxf.abs().max())
    assert float((dx1[:, :N].float() - dx0[:, :N].float()).abs().max()) <= 2 ** -7 * float(dxf.abs().max())      # (one bf16 rounding step apart at most)
    if xs:
        assert float(dx1[:, N:].float().abs().max()) == 0.0                          # nothing written between the rows
    ref = wz.grad.numpy()
    tol = 1e-4 * max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(dw1.cpu().numpy(), ref, rtol=0, atol=tol)
    np.testing.assert_allclose(dw1.cpu().numpy(), dw0.cpu().numpy(), rtol=0, atol=tol)
    if fused:
        assert not bool(torch.isnan(part).any())
        # the sums over the rows as stored: g = dz * act'(scale * y + shift) ; sum g, sum g (y - mean), in float64 on the host
        dz = dx1[:, :N].double().cpu(); yv = fy[:, :N].double().cpu()
        pre = fy[:, :N].float().cpu() * fsc.cpu() + fsh.cpu()                       # (fp32, as the kernel forms it)
        gg = dz * torch.where(pre > 0, torch.ones_like(dz), torch.full_like(dz, float(np.float32(0.1))))
        s_g, s_x = gg.sum(0), (gg * (yv - fmean.double().cpu())).sum(0)
        got_g, got_x = part[:, 0].double().sum(0).cpu(), part[:, 1].double().sum(0).cpu()
        scale_g = float(gg.abs().sum(0).max())
        assert float((got_g - s_g).abs().max()) <= 2e-6 * scale_g and float((got_x - s_x).abs().max()) <= 4e-6 * scale_g * 3
        if N <= 512:                                                                 # ... and through the finalize, against the stand-alone reduce pass
            acc = torch.zeros(3 * N, dtype=torch.float64, device="cuda")
            pws = torch.empty(L.bn_act_bwd_reduce_ws_floats(BF16, M, N, 2), device="cuda")
            dxc = dx1[:, :N].contiguous(); fyc = fy[:, :N].contiguous()
            L.check(L.bn_act_bwd_reduce(BF16, dxc.data_ptr(), N, fyc.data_ptr(), N, fsc.data_ptr(), fsh.data_ptr(), fmean.data_ptr(), finv.data_ptr(),
                                        None, 0, None, None, None, None, acc.data_ptr(), pws.data_ptr(), M, N, 1, 0.1, st()))
            refc = [torch.zeros(N, device="cuda") for _ in range(5)]
            L.check(L.bn_bwd_finalize(acc.data_ptr(), 1, 2, 1, float(M), gamma.data_ptr(), fmean.data_ptr(), finv.data_ptr(), *[b.data_ptr() for b in refc], N, st()))
            got = [torch.zeros(N, device="cuda") for _ in range(5)]
            L.check(L.bn_bwd_finalize_rows(part.data_ptr(), slabs, N, float(M), gamma.data_ptr(), fmean.data_ptr(), finv.data_ptr(),
                                           *[b.data_ptr() for b in got], st()))
            torch.cuda.synchronize()
            for a_, b_, name in zip(got, refc, ("dgamma", "dbeta", "cA", "cB", "cC")):
                an, bn_ = a_.cpu().numpy(), b_.cpu().numpy()
                np.testing.assert_allclose(an, bn_, rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(bn_).max())), err_msg=name)


WGRAD_SHIFT_CASES = [(2, 128, 13, 13, 128), (3, 128, 26, 20, 256), (1, 256, 52, 52, 128), (5, 128, 9, 8, 128), (32, 128, 13, 13, 256),
                     (8, 128, 80, 80, 128)]


@pytest.mark.parametrize("case", WGRAD_SHIFT_CASES, ids=[str(c) for c in WGRAD_SHIFT_CASES])
def test_wgrad_shift_kernel(case):
    """3x3 stride-1 weight gradient with the kw taps sharing one activation tile == torch reference == generic kernel."""
    L = VariantLib()
    dt = BF16
    B, Ci, H, W, Co = case
    g = torch.Generator().manual_seed(Ci + Co + H)
    x = torch.randn(B, Ci, H, W, generator=g)
    dy = torch.randn(B, Co, H, W, generator=g)
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    xr = rnd(dt, x)
    F.conv2d(xr, w, None, stride=1, padding=1).backward(rnd(dt, dy))
    ref = w.grad.numpy()
    xb, dyb = to_nhwc(x, dt), to_nhwc(dy, dt)
    M, ktot = B * H * W, 9 * Ci
    outs = {}
    for variant in (8, 0, 9):                   # 8: kw-shared kernel wherever eligible ; 0: default dispatch ; 9: generic kernel
        L.conv2d_wgrad_set_variant(variant)
        try:
            splits = L.conv2d_wgrad_splits_geom(dt, B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 1, Co, Ci)
            ws = torch.full((splits * Co * ktot,), float("nan"), dtype=torch.float32, device="cuda")
            dw = torch.full((Co, Ci, 3, 3), 7.0, dtype=torch.float32, device="cuda")
            L.check(L.conv2d_wgrad(dt, dyb.data_ptr(), Co, xb.data_ptr(), Ci, ws.data_ptr(), splits, dw.data_ptr(), 0, B, H, W, Ci, Ci,
                                   H, W, Co, Co, 3, 3, 1, 1, 1, st()), "wgrad")
            torch.cuda.synchronize()
            outs[variant] = (dw.cpu().numpy(), splits)
        finally:
            L.conv2d_wgrad_set_variant(0)
    scale = max(1.0, float(np.abs(ref).max()))
    for v, (got, splits) in outs.items():
        assert np.isfinite(got).all(), v
        np.testing.assert_allclose(got, ref, rtol=2e-2, atol=2e-2 * scale, err_msg=f"variant {v} splits {splits}")
    np.testing.assert_allclose(outs[8][0], outs[9][0], rtol=1e-3, atol=1e-3 * scale)      # same bf16 products, fp32 sums in another order


WGRAD_STREAM_CASES = [(2, 16, 16, 12, 9, 1), (3, 16, 16, 80, 80, 2), (2, 16, 32, 30, 17, 2), (4, 32, 32, 80, 80, 1), (2, 32, 64, 40, 33, 2),
                      (3, 64, 64, 26, 20, 1), (2, 64, 128, 80, 80, 2), (1, 64, 128, 104, 104, 1), (1, 32, 64, 208, 208, 1), (5, 64, 64, 5, 4, 1),
                      # channel-tiled instantiation (128 co x 64 ci tiles): the wide layers of YOLOv3 / RektNet's 128 -> 128
                      (2, 128, 256, 26, 26, 1), (3, 256, 128, 13, 13, 1), (1, 128, 128, 80, 80, 1), (2, 192, 384, 9, 11, 1), (2, 128, 128, 20, 17, 2),
                      (4, 512, 1024, 13, 13, 1)]


@pytest.mark.parametrize("case", WGRAD_STREAM_CASES, ids=[str(c) for c in WGRAD_STREAM_CASES])
def test_wgrad_stream_kernel(case):
    """3x3 stride-1 weight gradient (dilation 1 / 2) with the nine taps reading one LDS activation ring == torch reference
    == generic kernel."""
    L = VariantLib()
    dt = BF16
    B, Ci, Co, H, W, dil = case
    g = torch.Generator().manual_seed(Ci + Co + H + dil)
    x = torch.randn(B, Ci, H, W, generator=g)
    dy = torch.randn(B, Co, H, W, generator=g)
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(rnd(dt, x), w, None, stride=1, padding=dil, dilation=dil).backward(rnd(dt, dy))
    ref = w.grad.numpy()
    xb, dyb = to_nhwc(x, dt), to_nhwc(dy, dt)
    ktot = 9 * Ci
    outs = {}
    for variant in (0, 34050, 34020, 9):        # 0: default dispatch (stream kernel, DMA addresses from the LDS table; slab-free where one split fills the chip) ;
                                                # 34050: the slab form everywhere ; 34020: its stepping-lane form ; 9: generic kernel
        L.conv2d_wgrad_set_variant(variant)
        try:
            splits = L.conv2d_wgrad_splits_geom(dt, B, H, W, Ci, H, W, Co, 3, 3, 1, dil, dil, Co, Ci)
            ws = torch.full((splits * Co * ktot,), float("nan"), dtype=torch.float32, device="cuda")
            dw = torch.full((Co, Ci, 3, 3), 7.0, dtype=torch.float32, device="cuda")
            L.check(L.conv2d_wgrad(dt, dyb.data_ptr(), Co, xb.data_ptr(), Ci, ws.data_ptr(), splits, dw.data_ptr(), 0, B, H, W, Ci, Ci,
                                   H, W, Co, Co, 3, 3, 1, dil, dil, st()), "wgrad")
            torch.cuda.synchronize()
            outs[variant] = (dw.cpu().numpy(), splits)
        finally:
            L.conv2d_wgrad_set_variant(0)
    scale = max(1.0, float(np.abs(ref).max()))
    for v, (got, splits) in outs.items():
        assert np.isfinite(got).all(), v
        np.testing.assert_allclose(got, ref, rtol=2e-2, atol=2e-2 * scale, err_msg=f"variant {v} splits {splits}")
    np.testing.assert_allclose(outs[0][0], outs[9][0], rtol=1e-3, atol=1e-3 * scale)      # same bf16 products, fp32 sums in another order
    assert outs[34050][1] == outs[34020][1] and np.array_equal(outs[34050][0], outs[34020][0])     # same lanes, same ring image: the table changes addresses' COST only
    np.testing.assert_allclose(outs[0][0], outs[34050][0], rtol=1e-3, atol=1e-3 * scale)   # slab-free form (where taken: splits == 1): same products, another summation order


WGRAD_S2_CASES = [  # B, Cin, Cout, Hout, Wout  (input 2 Hout x 2 Wout); last entry: x / dy are channel slices of wider buffers
    (2, 32, 64, 13, 13, 0), (3, 64, 128, 9, 14, 0), (2, 128, 256, 26, 26, 0), (1, 32, 64, 52, 52, 0), (4, 96, 192, 5, 4, 0), (2, 64, 64, 40, 7, 0),
    (5, 32, 128, 13, 11, 8), (32, 512, 1024, 13, 13, 0)]


@pytest.mark.parametrize("case", WGRAD_S2_CASES, ids=[str(c) for c in WGRAD_S2_CASES])
def test_wgrad_stride2_parity_plane_kernel(case):
    """3x3 / stride-2 / pad-1 weight gradient (Darknet-53's down-sampling layers, reference models.py create_modules) on the parity-plane LDS ring
    (csrc/wgrad_stream_s2.hip) == torch reference == generic kernel; image and row boundaries, several splits and table windows, many channel
    tiles, operands that are channel slices of wider NHWC buffers; bit-identical from run to run."""
    L = VariantLib()
    dt = BF16
    B, Ci, Co, Ho, Wo, extra = case
    H, W = 2 * Ho, 2 * Wo
    g = torch.Generator().manual_seed(Ci + Co + Ho + 3 * Wo)
    xw = torch.randn(B, Ci + 3 * extra, H, W, generator=g)
    dyw = torch.randn(B, Co + 2 * extra, Ho, Wo, generator=g)
    x, dy = xw[:, extra:extra + Ci], dyw[:, 2 * extra:2 * extra + Co]
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(rnd(dt, x), w, None, stride=2, padding=1).backward(rnd(dt, dy))
    ref = w.grad.numpy()
    xb, dyb = to_nhwc(xw, dt), to_nhwc(dyw, dt)
    xl, dyl = xb.shape[-1], dyb.shape[-1]
    xp, dyp = xb.data_ptr() + extra * 2, dyb.data_ptr() + 2 * extra * 2
    outs = {}
    for variant in (0, 0, 34060):               # default dispatch twice (the parity-plane kernel), 34060: the generic kernel
        L.conv2d_wgrad_set_variant(variant)
        try:
            splits = L.conv2d_wgrad_splits_geom(dt, B, H, W, Ci, Ho, Wo, Co, 3, 3, 2, 1, 1, dyl, xl)
            ws = torch.full((splits * Co * 9 * Ci,), float("nan"), dtype=torch.float32, device="cuda")
            dw = torch.full((Co, Ci, 3, 3), 7.0, dtype=torch.float32, device="cuda")
            L.check(L.conv2d_wgrad(dt, dyp, dyl, xp, xl, ws.data_ptr(), splits, dw.data_ptr(), 0, B, H, W, Ci, Ci,
                                   Ho, Wo, Co, Co, 3, 3, 2, 1, 1, st()), "wgrad")
            torch.cuda.synchronize()
            outs.setdefault(variant, []).append((dw.cpu().numpy(), splits))
        finally:
            L.conv2d_wgrad_set_variant(0)
    scale = max(1.0, float(np.abs(ref).max()))
    for v, runs in outs.items():
        for got, splits in runs:
            assert np.isfinite(got).all(), v
            np.testing.assert_allclose(got, ref, rtol=2e-2, atol=2e-2 * scale, err_msg=f"variant {v} splits {splits}")
    assert np.array_equal(outs[0][0][0], outs[0][1][0])                                       # fixed-order sums
    np.testing.assert_allclose(outs[0][0][0], outs[34060][0][0], rtol=1e-3, atol=1e-3 * scale)   # same bf16 products, fp32 sums in another order



@pytest.mark.parametrize("case", [(4, 128, 256, 13, 13, 1), (2, 64, 128, 26, 20, 1), (3, 256, 128, 9, 17, 2), (2, 128, 128, 52, 52, 1)], ids=str)
def test_wgrad_tiled_light_and_heavy_forms(case):
    """The channel-tiled LDS-ring weight gradient has two forms: 128 co x 64 ci per block on 8 waves (it owns its CU) and the light one,
    64 co x 64 ci on 4 waves (half a CU, the default for position streams of up to 600 K), csrc/wgrad_stream.hip.  Both == torch, both
    deterministic, and they agree to fp32 summation order (different splits)."""
    L = VariantLib()
    dt = BF16
    B, Ci, Co, H, W, dil = case
    g = torch.Generator().manual_seed(Ci + Co + H + W)
    x = torch.randn(B, Ci, H, W, generator=g)
    dy = torch.randn(B, Co, H, W, generator=g)
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(rnd(dt, x), w, None, stride=1, padding=dil, dilation=dil).backward(rnd(dt, dy))
    ref = w.grad.numpy()
    xb, dyb = to_nhwc(x, dt), to_nhwc(dy, dt)
    outs = {}
    try:
        for form, code in (("heavy", 30005), ("light", 30003)):
            L.check(L.conv2d_wgrad_set_variant(code))
            runs = []
            for _ in range(2):
                splits = L.conv2d_wgrad_splits_geom(dt, B, H, W, Ci, H, W, Co, 3, 3, 1, dil, dil, Co, Ci)
                ws = torch.full((splits * Co * 9 * Ci,), float("nan"), dtype=torch.float32, device="cuda")
                dw = torch.full((Co, Ci, 3, 3), 7.0, dtype=torch.float32, device="cuda")
                L.check(L.conv2d_wgrad(dt, dyb.data_ptr(), Co, xb.data_ptr(), Ci, ws.data_ptr(), splits, dw.data_ptr(), 0, B, H, W, Ci, Ci,
                                       H, W, Co, Co, 3, 3, 1, dil, dil, st()), "wgrad " + form)
                torch.cuda.synchronize()
                runs.append(dw.cpu().numpy())
            assert np.array_equal(runs[0], runs[1]), form
            outs[form] = runs[0]
    finally:
        L.conv2d_wgrad_set_variant(30002)          # back to the automatic choice
    scale = max(1.0, float(np.abs(ref).max()))
    for form, got in outs.items():
        np.testing.assert_allclose(got, ref, rtol=2e-2, atol=2e-2 * scale, err_msg=form)
    np.testing.assert_allclose(outs["light"], outs["heavy"], rtol=1e-3, atol=1e-3 * scale)


@pytest.mark.parametrize("case", [(2, 32, 64, 20, 13, 1), (1, 16, 16, 9, 31, 2), (3, 64, 128, 16, 16, 1)], ids=str)
def test_wgrad_stream_kernel_channel_slices(case):
    """The LDS-ring weight gradient on operands that are channel slices of wider NHWC buffers (route / concat buffers: ldc > C,
    base pointer inside a pixel row) == the generic kernel on the same slices == torch."""
    L = VariantLib()
    dt = BF16
    B, Ci, Co, H, W, dil = case
    g = torch.Generator().manual_seed(11 * Ci + Co + W)
    xw = torch.randn(B, Ci + 24, H, W, generator=g)           # x lives at channels 8 .. 8+Ci of a wider buffer
    dyw = torch.randn(B, Co + 16, H, W, generator=g)          # dy at channels 16 .. 16+Co
    x, dy = xw[:, 8:8 + Ci], dyw[:, 16:16 + Co]
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(rnd(dt, x), w, None, stride=1, padding=dil, dilation=dil).backward(rnd(dt, dy))
    ref = w.grad.numpy()
    xb, dyb = to_nhwc(xw, dt), to_nhwc(dyw, dt)
    xl, dyl = xb.shape[-1], dyb.shape[-1]
    xp, dyp = xb.data_ptr() + 8 * 2, dyb.data_ptr() + 16 * 2
    outs = {}
    for variant in (0, 9):
        L.conv2d_wgrad_set_variant(variant)
        try:
            splits = L.conv2d_wgrad_splits_geom(dt, B, H, W, Ci, H, W, Co, 3, 3, 1, dil, dil, dyl, xl)
            ws = torch.full((splits * Co * 9 * Ci,), float("nan"), dtype=torch.float32, device="cuda")
            dw = torch.full((Co, Ci, 3, 3), 7.0, dtype=torch.float32, device="cuda")
            L.check(L.conv2d_wgrad(dt, dyp, dyl, xp, xl, ws.data_ptr(), splits, dw.data_ptr(), 0, B, H, W, Ci, Ci,
                                   H, W, Co, Co, 3, 3, 1, dil, dil, st()), "wgrad")
            torch.cuda.synchronize()
            outs[variant] = dw.cpu().numpy()
        finally:
            L.conv2d_wgrad_set_variant(0)
    scale = max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(outs[0], ref, rtol=2e-2, atol=2e-2 * scale)
    np.testing.assert_allclose(outs[0], outs[9], rtol=1e-3, atol=1e-3 * scale)


def test_wgrad_stream_accumulate_and_determinism():
    """accumulate=1 adds to the existing gradient; two runs give bit-identical results (fixed-order slab and wave sums)."""
    L = _lib.lib()
    dt = BF16
    B, Ci, Co, H, W, dil = 4, 32, 32, 40, 40, 1
    g = torch.Generator().manual_seed(5)
    xb, dyb = to_nhwc(torch.randn(B, Ci, H, W, generator=g), dt), to_nhwc(torch.randn(B, Co, H, W, generator=g), dt)
    splits = L.conv2d_wgrad_splits_geom(dt, B, H, W, Ci, H, W, Co, 3, 3, 1, dil, dil, Co, Ci)
    ws = torch.empty(splits * Co * 9 * Ci, dtype=torch.float32, device="cuda")
    res = []
    for acc, init in ((0, 3.0), (0, -1.0), (1, 2.5)):
        dw = torch.full((Co, Ci, 3, 3), init, dtype=torch.float32, device="cuda")
        L.check(L.conv2d_wgrad(dt, dyb.data_ptr(), Co, xb.data_ptr(), Ci, ws.data_ptr(), splits, dw.data_ptr(), acc, B, H, W, Ci, Ci,
                               H, W, Co, Co, 3, 3, 1, dil, dil, st()), "wgrad")
        torch.cuda.synchronize()
        res.append(dw.cpu().numpy())
    assert np.array_equal(res[0], res[1])
    np.testing.assert_allclose(res[2], res[0] + 2.5, rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("case", [(2, 80, 80), (3, 33, 20), (5, 9, 12), (1, 128, 96)], ids=str)
def test_wgrad_stem_7x7_kernel(case):
    """RektNet's 7x7 / pad 3 stem (3 real input channels in a 16-channel buffer, 16 output channels): LDS-ring kernel == torch ==
    the generic kernel on the same buffers; accumulate flag; guard region behind the slabs."""
    L = VariantLib()
    dt = BF16
    B, H, W = case
    Ci, Co = 3, 16
    g = torch.Generator().manual_seed(H * 3 + W)
    x = torch.randn(B, Ci, H, W, generator=g)
    dy = torch.randn(B, Co, H, W, generator=g)
    w = torch.zeros(Co, Ci, 7, 7, requires_grad=True)
    F.conv2d(rnd(dt, x), w, None, stride=1, padding=3).backward(rnd(dt, dy))
    ref = w.grad.numpy()
    xb, dyb = to_nhwc(x, dt, cpad=16), to_nhwc(dy, dt)
    outs = {}
    for variant in (0, 9):
        L.conv2d_wgrad_set_variant(variant)
        try:
            splits = L.conv2d_wgrad_splits_geom(dt, B, H, W, 16, H, W, Co, 7, 7, 1, 3, 1, Co, 16)
            n = splits * Co * 49 * 16
            ws = torch.full((n + 4096,), float("nan"), dtype=torch.float32, device="cuda")
            dw = torch.full((Co, Ci, 7, 7), 2.0, dtype=torch.float32, device="cuda")
            L.check(L.conv2d_wgrad(dt, dyb.data_ptr(), Co, xb.data_ptr(), 16, ws.data_ptr(), splits, dw.data_ptr(), 1, B, H, W, 16, Ci,
                                   H, W, Co, Co, 7, 7, 1, 3, 1, st()), "wgrad")
            torch.cuda.synchronize()
            assert bool(torch.isnan(ws[n:]).all()), "slabs written past the end"
            outs[variant] = dw.cpu().numpy() - 2.0
        finally:
            L.conv2d_wgrad_set_variant(0)
    scale = max(1.0, float(np.abs(ref).max()))
    for v, got in outs.items():
        np.testing.assert_allclose(got, ref, rtol=2e-2, atol=2e-2 * scale, err_msg=f"variant {v}")
    np.testing.assert_allclose(outs[0], outs[9], rtol=1e-3, atol=1e-3 * scale)


# (B, Cin, H, W, Cout, k, stride, with_resid, act): shift kernel, 1x1 / stride-2 / narrow im2col tiles, every activation code
AFFINE_CASES = [(3, 128, 26, 26, 256, 3, 1, True, 1), (2, 256, 13, 13, 128, 1, 1, False, 1), (2, 64, 30, 30, 128, 3, 2, False, 1),
                (2, 16, 40, 40, 32, 3, 1, False, 2), (1, 32, 20, 17, 64, 3, 1, True, 0), (2, 8, 24, 24, 16, 7, 1, False, 2)]


@pytest.mark.parametrize("dt", [F32, BF16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", AFFINE_CASES, ids=[str(c) for c in AFFINE_CASES])
def test_conv2d_affine_act_inference_epilogue(case, dt):
    """mdcv_conv2d_affine_act == act(conv(x) * scale + shift) (+ residual) in torch, and == the two-pass form (mdcv_conv2d followed by
    mdcv_bn_act_fwd) up to the bf16 rounding of the intermediate the fused form never stores."""
    L = _lib.lib()
    B, Ci, H, W, Co, k, s, with_resid, act = case
    p = (k - 1) // 2
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    g = torch.Generator().manual_seed(Ci + Co + k)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    scale = torch.rand(Co, generator=g) + 0.5
    shift = torch.randn(Co, generator=g) * 0.3
    resid = torch.randn(B, Co, Ho, Wo, generator=g) if with_resid else None
    slope = 0.1
    y = F.conv2d(rnd(dt, x), rnd(dt, w), None, stride=s, padding=p) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    ref = {0: y, 1: F.leaky_relu(y, slope), 2: F.relu(y)}[act]
    if with_resid:
        ref = ref + rnd(dt, resid)
    cip, cop = pad8(Ci), pad8(Co)
    xb = to_nhwc(x, dt)
    wf, _ = pack(dt, w, need_d=False)
    rb = to_nhwc(resid, dt) if with_resid else None
    sc, sh = torch.zeros(cop), torch.zeros(cop)
    sc[:Co], sh[:Co] = scale, shift
    sc, sh = sc.cuda(), sh.cuda()
    out = torch.full((B, Ho, Wo, cop), float("nan"), dtype=TD[dt], device="cuda")
    L.check(L.conv2d_affine_act(dt, xb.data_ptr(), cip, wf.data_ptr(), out.data_ptr(), cop, sc.data_ptr(), sh.data_ptr(),
                                rb.data_ptr() if rb is not None else None, cop, act, slope, B, H, W, cip, Ho, Wo, cop, k, k, s, p, 1, st()), "affine_act")
    got = to_nchw(out, dt, Co).numpy()
    tol = 1e-4 if dt == F32 else 2e-2
    np.testing.assert_allclose(got, ref.numpy(), rtol=tol, atol=tol * max(1.0, float(ref.abs().max())))
    # two-pass form on the same operands
    ybuf = torch.empty(B, Ho, Wo, cop, dtype=TD[dt], device="cuda")
    L.check(L.conv2d(dt, 0, xb.data_ptr(), cip, wf.data_ptr(), ybuf.data_ptr(), cop, None, None, 0, None, B, H, W, cip, Ho, Wo, cop, k, k, s, p, 1, st()))
    out2 = torch.empty_like(out)
    L.check(L.bn_act_fwd(dt, ybuf.data_ptr(), cop, sc.data_ptr(), sh.data_ptr(), None, 0, None, None, rb.data_ptr() if rb is not None else None,
                         cop, out2.data_ptr(), cop, B * Ho * Wo, cop, act, slope, st()))
    got2 = to_nchw(out2, dt, Co).numpy()
    np.testing.assert_allclose(got, got2, rtol=tol, atol=tol * max(1.0, float(ref.abs().max())))


@pytest.mark.parametrize("case", [(4, 64, 26, 26, 64), (3, 32, 30, 17, 64), (2, 64, 80, 80, 64), (33, 64, 13, 13, 64),
                                  (4, 32, 26, 26, 32), (2, 64, 80, 80, 32), (33, 32, 13, 13, 32), (2, 128, 104, 104, 64), (1, 32, 97, 104, 32)])
def test_shift_conv_64_wide_tile_column(case):
    """64- (and, variant -19, 32-) channel layers through the shift kernel's narrow tile column == the im2col kernel (-18): forward
    with BatchNorm statistics, data gradient with addsrc."""
    L = VariantLib()
    dt = BF16
    B, Ci, H, W, Co = case
    gg = torch.Generator().manual_seed(B + Ci + W)
    x = torch.randn(B, Ci, H, W, generator=gg)
    w = torch.randn(Co, Ci, 3, 3, generator=gg) / (Ci * 9) ** 0.5
    w2 = torch.randn(Ci, Co, 3, 3, generator=gg) / (Co * 9) ** 0.5      # a layer whose INPUT has 64 channels: its data gradient has N = 64
    xb = to_nhwc(x, dt)
    wf, wd = pack(dt, w)
    wf2, wd2 = pack(dt, w2)
    dyb = to_nhwc(torch.randn(B, Ci, H, W, generator=gg), dt)
    addb = to_nhwc(torch.randn(B, Co, H, W, generator=gg), dt)
    outs = {}
    for v in (-18, -17):
        L.conv2d_set_variant(v if Co != 32 or v == -18 else -19)
        try:
            y = torch.full((B, H, W, Co), float("nan"), dtype=TD[dt], device="cuda")
            rows = L.conv2d_stats_rows_geom(dt, B, H, W, Ci, Co, 3, 3, 1, 1, 1, Ci)
            stats = torch.zeros(rows, 2, Co, device="cuda")
            L.check(L.conv2d(dt, 0, xb.data_ptr(), Ci, wf.data_ptr(), y.data_ptr(), Co, None, None, 0, stats.data_ptr(),
                             B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 1, st()), "conv")
            dx = torch.full((B, H, W, Co), float("nan"), dtype=TD[dt], device="cuda")
            L.check(L.conv2d(dt, 1, dyb.data_ptr(), Ci, wd2.data_ptr(), dx.data_ptr(), Co, None, addb.data_ptr(), Co, None,
                             B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 1, st()), "dgrad")
            torch.cuda.synchronize()
            outs[v] = (y.float().cpu(), stats.sum(0).cpu(), dx.float().cpu())
        finally:
            L.conv2d_set_variant(-19)               # the library default
    ref = F.conv2d(rnd(dt, x), rnd(dt, w), None, stride=1, padding=1).permute(0, 2, 3, 1)
    for v in outs:
        assert torch.isfinite(outs[v][0]).all() and torch.isfinite(outs[v][2]).all(), v
        torch.testing.assert_close(outs[v][0], ref, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(outs[-17][0], outs[-18][0], rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(outs[-17][2], outs[-18][2], rtol=1e-2, atol=2e-2)
    torch.testing.assert_close(outs[-17][1], outs[-18][1], rtol=2e-3, atol=0.5)


S2D_CASES = [(2, 64, 104, 104, 32), (1, 128, 52, 52, 64), (3, 32, 17, 45, 32), (2, 64, 8, 31, 64), (1, 32, 9, 63, 32), (32, 64, 208, 208, 32)]


@pytest.mark.parametrize("with_add", [False, True, "fused"], ids=["plain", "addsrc", "addsrc+bnsums"])
@pytest.mark.parametrize("case", S2D_CASES, ids=[str(c) for c in S2D_CASES])
def test_shift_stride2_dgrad(case, with_add):
    """3x3 / stride-2 / pad-1 data gradients with 32 or 64 output channels run the shift kernel's stride-2 form (one accumulator set per
    output parity class, whole output rows per store; variant -60, the default) == the per-class im2col path (-29) == torch
    conv_transpose2d; ragged tiles and addsrc included.  case = (B, Cdy, Hdy, Wdy, Cdx)."""
    L = VariantLib()
    dt = BF16
    B, Co, H, W, Ci = case                      # the layer: Ci -> Co, stride 2, input 2H x 2W
    gg = torch.Generator().manual_seed(B + Co + W + 9)
    w = torch.randn(Co, Ci, 3, 3, generator=gg) / (Co * 9 / 4) ** 0.5
    _, wd = pack(dt, w)
    dy = torch.randn(B, Co, H, W, generator=gg)
    dyb = to_nhwc(dy, dt)
    add = torch.randn(B, Ci, 2 * H, 2 * W, generator=gg)
    addb = to_nhwc(add, dt) if with_add else None
    fused = with_add == "fused"
    yb = to_nhwc(torch.randn(B, Ci, 2 * H, 2 * W, generator=gg) * 1.3 + 0.2, dt)
    M = B * 4 * H * W
    scale = (torch.rand(Ci, generator=gg) + 0.5).cuda(); shift = (torch.randn(Ci, generator=gg) * 0.3).cuda()
    mean = (torch.randn(Ci, generator=gg) * 0.2 + 0.2).cuda(); invstd = (torch.rand(Ci, generator=gg) + 0.5).cuda(); gamma = (torch.rand(Ci, generator=gg) + 0.5).cuda()
    outs, coefs = {}, {}
    for v in (-29, -60):
        L.conv2d_set_variant(v)
        try:
            dx = torch.full((B, 2 * H, 2 * W, Ci), float("nan"), dtype=TD[dt], device="cuda")
            if fused:
                prow = L.conv2d_dgrad_bnsums_rows(dt, B, H, W, Co, 2 * H, 2 * W, Ci, 3, 3, 2, 1, 1, Co)
                assert prow > 0
                part = torch.full((prow, 2, Ci), float("nan"), device="cuda")
                L.check(L.conv2d_dgrad_bnsums(dt, dyb.data_ptr(), Co, wd.data_ptr(), dx.data_ptr(), Ci, addb.data_ptr(), Ci, B, H, W, Co, 2 * H, 2 * W, Ci,
                                              3, 3, 2, 1, 1, yb.data_ptr(), Ci, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), 1, 0.1,
                                              part.data_ptr(), st()), "fused s2 dgrad")
                torch.cuda.synchronize()
                assert not bool(torch.isnan(part).any())
                coefs[v] = [torch.zeros(Ci, device="cuda") for _ in range(5)]
                L.check(L.bn_bwd_finalize_rows(part.data_ptr(), prow, Ci, float(M), gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                               *[b.data_ptr() for b in coefs[v]], st()))
            else:
                L.check(L.conv2d(dt, 1, dyb.data_ptr(), Co, wd.data_ptr(), dx.data_ptr(), Ci, None, addb.data_ptr() if with_add else None, Ci, None,
                                 B, H, W, Co, 2 * H, 2 * W, Ci, 3, 3, 2, 1, 1, st()), "s2 dgrad")
            torch.cuda.synchronize()
            outs[v] = dx.float().cpu()
        finally:
            L.conv2d_set_variant(-60)
    if fused:                                   # the sums of the shift form against a stand-alone reduce over ITS stored dx
        acc = torch.zeros(3 * Ci, dtype=torch.float64, device="cuda")
        pws = torch.empty(L.bn_act_bwd_reduce_ws_floats(dt, M, Ci, 2), device="cuda")
        dxs = outs[-60].to(TD[dt]).cuda().contiguous()
        L.check(L.bn_act_bwd_reduce(dt, dxs.data_ptr(), Ci, yb.data_ptr(), Ci, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                    None, 0, None, None, None, None, acc.data_ptr(), pws.data_ptr(), M, Ci, 1, 0.1, st()))
        refc = [torch.zeros(Ci, device="cuda") for _ in range(5)]
        L.check(L.bn_bwd_finalize(acc.data_ptr(), 1, 2, 1, float(M), gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr(), *[b.data_ptr() for b in refc], Ci, st()))
        torch.cuda.synchronize()
        for a_, b_, name in zip(coefs[-60], refc, ("dgamma", "dbeta", "cA", "cB", "cC")):
            np.testing.assert_allclose(a_.cpu().numpy(), b_.cpu().numpy(), rtol=2e-4, atol=2e-4 * max(1.0, float(b_.abs().max())), err_msg=name)
    ref = F.conv_transpose2d(rnd(dt, dy), rnd(dt, w), None, stride=2, padding=1, output_padding=1)
    if with_add:
        ref = ref + rnd(dt, add)
    ref = ref.permute(0, 2, 3, 1)
    for v in outs:
        assert torch.isfinite(outs[v]).all(), v
        torch.testing.assert_close(outs[v], ref, rtol=2e-2, atol=3e-2)
    torch.testing.assert_close(outs[-60], outs[-29], rtol=1e-2, atol=2e-2)


@pytest.mark.parametrize("dt", [BF16, F32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("case", [(32, 128, 52, 52, 256, 3, 1), (32, 256, 26, 26, 512, 3, 1), (8, 512, 13, 13, 1024, 3, 1), (3, 64, 37, 41, 128, 3, 1),
                                  (2, 32, 20, 24, 64, 3, 1), (4, 32, 104, 104, 64, 3, 1), (4, 8, 96, 96, 32, 3, 1), (4, 32, 64, 64, 64, 3, 2),
                                  (8, 256, 26, 26, 128, 1, 1), (2, 64, 80, 80, 24, 1, 1), (2, 8, 33, 31, 40, 7, 2)], ids=str)
def test_conv_xstats_exact_accumulators(case, dt):
    """conv -> BatchNorm(batch statistics) -> LeakyReLU (+ residual) as TWO launches with no partial rows: mdcv_conv2d_xstats adds its per-tile sums to
    exact accumulators (three 40-bit digits in 64-bit words, integer atomics: csrc/exact_acc.h), mdcv_bn_act_fwd_xstats forms the statistics in its
    prologue -- against conv -> mdcv_bn_stats_finalize -> mdcv_bn_act_fwd on the same buffers, for every forward kernel family (3x3 shift kernel
    in its 1-D / 2-D / narrow tilings, stride 2, 1x1, 7x7 stem; both dtypes).  Conv output bit-identical; the accumulators hold EXACTLY the sum of
    the partial rows the three-launch form wrote (checked in float64 / integer arithmetic: the rows are fp32 values, their exact sum is
    representable in the digits); scale / shift / mean / invstd / running statistics within fp32 rounding of the finalize's summation order;
    activations equal up to one rounding of the storage type on a handful of elements.  Run three times with re-zeroed accumulators and with
    1, 2 and the planned number of replicas: bit-identical every time (integer addition does not care about the order the atomics land in).
    case = (B, Cin, H, W, Cout, k, stride)."""
    L = _lib.lib()
    B, Ci, H, W, Co, k, stride = case
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    gg = torch.Generator().manual_seed(B + Ci + W + k)
    x = torch.randn(B, Ci, H, W, generator=gg)
    w = torch.randn(Co, Ci, k, k, generator=gg) / (Ci * k * k) ** 0.5
    Cip, Cop = (Ci + 7) // 8 * 8, (Co + 7) // 8 * 8
    xb = to_nhwc(x, dt, Cip)
    wf, _ = pack(dt, w, need_d=False)
    res = to_nhwc(torch.randn(B, Co, Ho, Wo, generator=gg), dt, Cop)
    M = B * Ho * Wo
    gamma = (torch.rand(Cop, generator=gg) + 0.5).cuda(); beta = (torch.randn(Cop, generator=gg) * 0.3).cuda()
    geom = (B, H, W, Cip, Ho, Wo, Cop, k, k, stride, pad, 1)
    rows = L.conv2d_stats_rows_geom(dt, B, Ho, Wo, Cip, Cop, k, k, stride, pad, 1, Cip)
    planned = L.xstats_reps(rows, Cop)
    assert planned >= 1 and planned & (planned - 1) == 0

    def three_launches():
        y = torch.full((B, Ho, Wo, Cop), float("nan"), dtype=TD[dt], device="cuda")
        stats = torch.zeros((rows, 2, Cop), device="cuda")
        L.check(L.conv2d(dt, 0, xb.data_ptr(), Cip, wf.data_ptr(), y.data_ptr(), Cop, None, None, 0, stats.data_ptr(), *geom, st()), "conv")
        co = [torch.zeros(Cop, device="cuda") for _ in range(4)]
        rm, rv = torch.zeros(Cop, device="cuda"), torch.ones(Cop, device="cuda")
        scratch = torch.zeros(3 * Cop, dtype=torch.float64, device="cuda")
        stats_keep = stats.clone()
        L.check(L.bn_stats_finalize(stats.data_ptr(), rows, scratch.data_ptr(), float(M), gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                    0.1, 1e-5, *[c.data_ptr() for c in co], Cop, st()))
        z = torch.full((B, Ho, Wo, Cop), float("nan"), dtype=TD[dt], device="cuda")
        L.check(L.bn_act_fwd(dt, y.data_ptr(), Cop, co[0].data_ptr(), co[1].data_ptr(), None, 0, None, None, res.data_ptr(), Cop, z.data_ptr(), Cop, M, Cop,
                             1, 0.1, st()))
        torch.cuda.synchronize()
        return y, stats_keep, co, rm, rv, z

    def two_launches(reps):
        y = torch.full((B, Ho, Wo, Cop), float("nan"), dtype=TD[dt], device="cuda")
        acc = torch.zeros(L.xstats_words(reps, Cop), dtype=torch.int64, device="cuda")
        L.check(L.conv2d_xstats(dt, xb.data_ptr(), Cip, wf.data_ptr(), y.data_ptr(), Cop, None, acc.data_ptr(), reps, *geom, st()), "conv + accumulate")
        co = [torch.full((Cop,), float("nan"), device="cuda") for _ in range(4)]
        rm, rv = torch.zeros(Cop, device="cuda"), torch.ones(Cop, device="cuda")
        z = torch.full((B, Ho, Wo, Cop), float("nan"), dtype=TD[dt], device="cuda")
        L.check(L.bn_act_fwd_xstats(dt, y.data_ptr(), Cop, acc.data_ptr(), reps, float(M), gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                    0.1, 1e-5, *[c.data_ptr() for c in co], res.data_ptr(), Cop, z.data_ptr(), Cop, M, Cop, 1, 0.1, st()), "apply")
        torch.cuda.synchronize()
        return y, acc, co, rm, rv, z

    ya, sa, ca, rma, rva, za = three_launches()
    runs = [two_launches(planned) for _ in range(3)]
    yb, acc, cb, rmb, rvb, zb = runs[0]
    assert torch.equal(ya, yb)
    # the digits, summed over the replicas, are exactly the sum of the fp32 partial rows: compare as integers in units of 2^-70 (python ints)
    d = acc.reshape(planned, 3, 2, Cop).sum(0).cpu().numpy().astype(object)
    tot = d[0] + d[1] * (1 << 40) + d[2] * (1 << 80)
    rows_np = sa.cpu().numpy().astype(np.float64)
    assert np.isfinite(rows_np).all()
    import fractions
    for which in range(2):
        for c in (0, 1, Cop // 2, Cop - 1):
            exact = sum(fractions.Fraction(float(v)) for v in rows_np[:, which, c])
            # values below 2^-46 lose low bits (truncation toward zero): allow rows * 2^-70 of slack
            assert abs(fractions.Fraction(int(tot[which, c]), 1 << 70) - exact) <= fractions.Fraction(rows, 1 << 70), (which, c)
    for a_, b_, name in zip(ca, cb, ("scale", "shift", "mean", "invstd")):
        np.testing.assert_allclose(b_.cpu().numpy(), a_.cpu().numpy(), rtol=2e-5, atol=2e-6, err_msg=name)
    np.testing.assert_allclose(rmb.cpu().numpy(), rma.cpu().numpy(), rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(rvb.cpu().numpy(), rva.cpu().numpy(), rtol=2e-5, atol=2e-7)
    diff = (za.float() != zb.float())
    assert dt == F32 or float(diff.float().mean()) < 2e-3, float(diff.float().mean())      # (fp32 storage shows every last-bit difference of scale / shift)
    torch.testing.assert_close(zb.float(), za.float(), rtol=1.6e-2 if dt == BF16 else 1e-4, atol=1e-3 if dt == BF16 else 1e-5)
    for r in runs[1:]:
        assert torch.equal(r[1], acc) and torch.equal(r[5], zb) and all(torch.equal(p, q) for p, q in zip(r[2], cb))
    for reps in (1, 2, L.xstats_reps(1 << 30, Cop)):          # (the last one: the most replicas the consumer's prologue takes at this channel count)
        if reps != planned and reps <= L.xstats_reps(1 << 30, Cop):
            r = two_launches(reps)
            assert torch.equal(r[5], zb) and all(torch.equal(p, q) for p, q in zip(r[2], cb)), reps
    assert L.conv2d_xstats(dt, xb.data_ptr(), Cip, wf.data_ptr(), yb.data_ptr(), Cop, None, acc.data_ptr(), 3, *geom, st()) != 0       # not a power of two


def test_xstats_nonfinite_poisons_the_channel():
    """A non-finite conv output must still show as NaN coefficients (the rows + finalize form propagates it through its sums)."""
    L = _lib.lib()
    dt, B, Ci, H, W, Co = BF16, 2, 32, 20, 24, 64
    gg = torch.Generator().manual_seed(5)
    x = torch.randn(B, Ci, H, W, generator=gg)
    x[1, 3, 7, 9] = float("inf")
    w = torch.randn(Co, Ci, 3, 3, generator=gg) / (Ci * 9) ** 0.5
    xb = to_nhwc(x, dt)
    wf, _ = pack(dt, w, need_d=False)
    M = B * H * W
    geom = (B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 1)
    y = torch.zeros((B, H, W, Co), dtype=TD[dt], device="cuda")
    acc = torch.zeros(L.xstats_words(2, Co), dtype=torch.int64, device="cuda")
    L.check(L.conv2d_xstats(dt, xb.data_ptr(), Ci, wf.data_ptr(), y.data_ptr(), Co, None, acc.data_ptr(), 2, *geom, st()))
    co = [torch.zeros(Co, device="cuda") for _ in range(4)]
    gamma, beta = torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda")
    z = torch.zeros((B, H, W, Co), dtype=TD[dt], device="cuda")
    L.check(L.bn_act_fwd_xstats(dt, y.data_ptr(), Co, acc.data_ptr(), 2, float(M), gamma.data_ptr(), beta.data_ptr(), None, None, 0.1, 1e-5,
                                *[c.data_ptr() for c in co], None, 0, z.data_ptr(), Co, M, Co, 1, 0.1, st()))
    torch.cuda.synchronize()
    assert bool(torch.isnan(co[0]).all()) and bool(torch.isnan(z.float()).any())


@pytest.mark.parametrize("Co,reps", [(64, 8), (256, 4), (512, 2), (64, 16)], ids=str)
@pytest.mark.parametrize("kind", ["inf_in_every_replica", "all_nan", "mixed_sign_inf"])
def test_xstats_poison_survives_the_replica_sum(kind, Co, reps):
    """the poison of a non-finite partial sum is an atomic max with INT64_MAX on ONE replica's top digit; the consumer adds the replicas,
    and two poisoned words wrap to -2 (32 to -32): finite garbage for scale / shift / running statistics where the rows + finalize path and the
    reference give NaN.  Non-finite values in several (all) replicas, an all-NaN tensor, and +inf / -inf together must all come out NaN."""
    L = _lib.lib()
    dt, B, Ci, H, W = BF16, 4, 32, 24, 32             # 3072 positions: 24 rows of 128 -> every replica of an 8-replica accumulator is hit
    gg = torch.Generator().manual_seed(11)
    x = torch.randn(B, Ci, H, W, generator=gg)
    if kind == "inf_in_every_replica":
        x[:, 5, ::3, ::5] = float("inf")
    elif kind == "all_nan":
        x[:] = float("nan")
    else:
        x[0, 2, :, :] = float("inf"); x[1:, 2, :, :] = float("-inf")
    w = torch.randn(Co, Ci, 3, 3, generator=gg) / (Ci * 9) ** 0.5
    xb = to_nhwc(x, dt)
    wf, _ = pack(dt, w, need_d=False)
    M = B * H * W
    geom = (B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 1)
    y = torch.zeros((B, H, W, Co), dtype=TD[dt], device="cuda")
    acc = torch.zeros(L.xstats_words(reps, Co), dtype=torch.int64, device="cuda")
    L.check(L.conv2d_xstats(dt, xb.data_ptr(), Ci, wf.data_ptr(), y.data_ptr(), Co, None, acc.data_ptr(), reps, *geom, st()))
    torch.cuda.synchronize()
    top = acc.view(reps, 3, 2, Co)[:, 2]                                           # the top digits: [reps][sum / sum of squares][Co]
    assert int((top == torch.iinfo(torch.int64).max).any(dim=1).any(dim=1).sum()) >= min(reps, 2)    # several replicas are poisoned: the case the sum got wrong
    co = [torch.zeros(Co, device="cuda") for _ in range(4)]
    gamma, beta = torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda")
    rm, rv = torch.zeros(Co, device="cuda"), torch.ones(Co, device="cuda")
    z = torch.zeros((B, H, W, Co), dtype=TD[dt], device="cuda")
    L.check(L.bn_act_fwd_xstats(dt, y.data_ptr(), Co, acc.data_ptr(), reps, float(M), gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5,
                                *[c.data_ptr() for c in co], None, 0, z.data_ptr(), Co, M, Co, 1, 0.1, st()))
    torch.cuda.synchronize()
    for name, t in (("scale", co[0]), ("shift", co[1]), ("mean", co[2]), ("invstd", co[3]), ("running_mean", rm), ("running_var", rv)):
        assert bool(torch.isnan(t).all()), (name, t[:8])
    assert bool(torch.isnan(z.float()).all())


T2D_CASES = [(2, 32, 208, 208, 64), (1, 64, 208, 208, 32), (3, 64, 104, 104, 128), (2, 32, 97, 131, 128), (1, 128, 9, 161, 128), (2, 32, 41, 300, 32)]


@pytest.mark.parametrize("fused", [False, True], ids=["plain", "bnsums"])
@pytest.mark.parametrize("case", T2D_CASES, ids=[str(c) for c in T2D_CASES])
def test_shift_conv_2d_tiles(case, fused):
    """Images wider than the 1-D position stream takes run the shift kernel over 2-D pixel tiles (8 x 30 outputs + halo ring; variant -28,
    the default) == the im2col kernel (-27) == torch: forward with BatchNorm statistics, data gradient with addsrc and (fused) the
    BatchNorm-backward sums of the producer layer; ragged right / bottom tiles included."""
    L = VariantLib()
    dt = BF16
    B, Ci, H, W, Co = case
    gg = torch.Generator().manual_seed(B + Ci + W + 5)
    x = torch.randn(B, Ci, H, W, generator=gg)
    w = torch.randn(Co, Ci, 3, 3, generator=gg) / (Ci * 9) ** 0.5
    xb = to_nhwc(x, dt)
    wf, wd = pack(dt, w)
    dy = torch.randn(B, Co, H, W, generator=gg)
    dyb = to_nhwc(dy, dt)
    add = torch.randn(B, Ci, H, W, generator=gg)
    addb = to_nhwc(add, dt)
    yb = to_nhwc(torch.randn(B, Ci, H, W, generator=gg) * 1.3 + 0.2, dt)
    M = B * H * W
    scale = (torch.rand(Ci, generator=gg) + 0.5).cuda(); shift = (torch.randn(Ci, generator=gg) * 0.3).cuda()
    mean = (torch.randn(Ci, generator=gg) * 0.2 + 0.2).cuda(); invstd = (torch.rand(Ci, generator=gg) + 0.5).cuda(); gamma = (torch.rand(Ci, generator=gg) + 0.5).cuda()
    outs = {}
    for v in (-27, -28):
        L.conv2d_set_variant(v)
        try:
            y = torch.full((B, H, W, Co), float("nan"), dtype=TD[dt], device="cuda")
            rows = L.conv2d_stats_rows_geom(dt, B, H, W, Ci, Co, 3, 3, 1, 1, 1, Ci)
            stats = torch.full((rows, 2, Co), float("nan"), device="cuda")
            L.check(L.conv2d(dt, 0, xb.data_ptr(), Ci, wf.data_ptr(), y.data_ptr(), Co, None, None, 0, stats.data_ptr(),
                             B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 1, st()), "conv")
            dx = torch.full((B, H, W, Ci), float("nan"), dtype=TD[dt], device="cuda")
            coef = None
            if fused:
                prow = L.conv2d_dgrad_bnsums_rows(dt, B, H, W, Co, H, W, Ci, 3, 3, 1, 1, 1, Co)
                assert prow > 0
                part = torch.full((prow, 2, Ci), float("nan"), device="cuda")
                L.check(L.conv2d_dgrad_bnsums(dt, dyb.data_ptr(), Co, wd.data_ptr(), dx.data_ptr(), Ci, addb.data_ptr(), Ci, B, H, W, Co, H, W, Ci,
                                              3, 3, 1, 1, 1, yb.data_ptr(), Ci, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), 1, 0.1,
                                              part.data_ptr(), st()), "fused dgrad")
                coef = [torch.zeros(Ci, device="cuda") for _ in range(5)]
                L.check(L.bn_bwd_finalize_rows(part.data_ptr(), prow, Ci, float(M), gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                               *[b.data_ptr() for b in coef], st()))
                torch.cuda.synchronize()
                assert not bool(torch.isnan(part).any())
            else:
                L.check(L.conv2d(dt, 1, dyb.data_ptr(), Co, wd.data_ptr(), dx.data_ptr(), Ci, None, addb.data_ptr(), Ci, None,
                                 B, H, W, Co, H, W, Ci, 3, 3, 1, 1, 1, st()), "dgrad")
            torch.cuda.synchronize()
            assert not bool(torch.isnan(stats).any())
            outs[v] = (y.float().cpu(), stats.sum(0).cpu(), dx.float().cpu(), [c.cpu() for c in coef] if coef else None)
        finally:
            L.conv2d_set_variant(-28)
    ref = F.conv2d(rnd(dt, x), rnd(dt, w), None, stride=1, padding=1).permute(0, 2, 3, 1)
    refd = (F.conv_transpose2d(rnd(dt, dy), rnd(dt, w), None, stride=1, padding=1) + rnd(dt, add)).permute(0, 2, 3, 1)
    for v in outs:
        assert torch.isfinite(outs[v][0]).all() and torch.isfinite(outs[v][2]).all(), v
        torch.testing.assert_close(outs[v][0], ref, rtol=2e-2, atol=2e-2)
        torch.testing.assert_close(outs[v][2], refd, rtol=2e-2, atol=3e-2)
    torch.testing.assert_close(outs[-28][1], outs[-27][1], rtol=2e-3, atol=0.5)
    if fused:
        for a_, b_, name in zip(outs[-28][3], outs[-27][3], ("dgamma", "dbeta", "cA", "cB", "cC")):
            np.testing.assert_allclose(a_.numpy(), b_.numpy(), rtol=5e-3, atol=5e-3 * max(1.0, float(b_.abs().max())), err_msg=name)


@pytest.mark.parametrize("case", [(2, 64, 80, 80, 128), (2, 32, 80, 80, 64), (3, 64, 26, 26, 64), (4, 32, 30, 17, 32), (33, 128, 13, 13, 128)])
def test_shift_conv_dilation2(case):
    """Dilation-2 / pad-2 layers through the shift kernel (variant -20: stream with two shared zero columns / rows) == the im2col kernel
    (-21) == torch: forward with BatchNorm statistics, data gradient with addsrc."""
    L = VariantLib()
    dt = BF16
    B, Ci, H, W, Co = case
    gg = torch.Generator().manual_seed(B + Ci + W + 2)
    x = torch.randn(B, Ci, H, W, generator=gg)
    w = torch.randn(Co, Ci, 3, 3, generator=gg) / (Ci * 9) ** 0.5
    xb = to_nhwc(x, dt)
    wf, wd = pack(dt, w)
    dy = torch.randn(B, Co, H, W, generator=gg)
    dyb = to_nhwc(dy, dt)
    add = torch.randn(B, Ci, H, W, generator=gg)
    addb = to_nhwc(add, dt)
    outs = {}
    for v in (-21, -20):
        L.conv2d_set_variant(v)
        try:
            y = torch.full((B, H, W, Co), float("nan"), dtype=TD[dt], device="cuda")
            rows = L.conv2d_stats_rows_geom(dt, B, H, W, Ci, Co, 3, 3, 1, 2, 2, Ci)
            stats = torch.zeros(rows, 2, Co, device="cuda")
            L.check(L.conv2d(dt, 0, xb.data_ptr(), Ci, wf.data_ptr(), y.data_ptr(), Co, None, None, 0, stats.data_ptr(),
                             B, H, W, Ci, H, W, Co, 3, 3, 1, 2, 2, st()), "conv")
            dx = torch.full((B, H, W, Ci), float("nan"), dtype=TD[dt], device="cuda")
            L.check(L.conv2d(dt, 1, dyb.data_ptr(), Co, wd.data_ptr(), dx.data_ptr(), Ci, None, addb.data_ptr(), Ci, None,
                             B, H, W, Co, H, W, Ci, 3, 3, 1, 2, 2, st()), "dgrad")
            torch.cuda.synchronize()
            outs[v] = (y.float().cpu(), stats.sum(0).cpu(), dx.float().cpu())
        finally:
            L.conv2d_set_variant(-22)               # the library default
    xr = rnd(dt, x).requires_grad_(True)
    ref = F.conv2d(xr, rnd(dt, w), None, stride=1, padding=2, dilation=2)
    ref.backward(rnd(dt, dy))
    refdx = (xr.grad + rnd(dt, add)).permute(0, 2, 3, 1)
    ref = ref.detach().permute(0, 2, 3, 1)
    for v in outs:
        assert torch.isfinite(outs[v][0]).all() and torch.isfinite(outs[v][2]).all(), v
        torch.testing.assert_close(outs[v][0], ref, rtol=2e-2, atol=2e-2)
        torch.testing.assert_close(outs[v][2], refdx, rtol=2e-2, atol=4e-2)
    torch.testing.assert_close(outs[-20][0], outs[-21][0], rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(outs[-20][2], outs[-21][2], rtol=1e-2, atol=2e-2)
    torch.testing.assert_close(outs[-20][1], outs[-21][1], rtol=2e-3, atol=0.5)


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("k,s,H,W", [(3, 1, 13, 17), (3, 2, 16, 20), (5, 1, 9, 9), (2, 2, 12, 8), (13, 1, 19, 19), (1, 1, 5, 7), (3, 3, 14, 11)])
def test_maxpool_generic(dt, k, s, H, W):
    """nn.MaxPool2d(k, s, (k - 1) // 2) for any window / stride (reference models.py:74-84 builds it from the cfg): forward values, the
    argmax-routed backward (ties: first maximum in scan order, like torch) against torch on the CPU."""
    L = _lib.lib()
    B, C = 2, 16
    p = (k - 1) // 2
    g = torch.Generator().manual_seed(k * 100 + s * 10 + H)
    x = rnd(dt, torch.randn(B, C, H, W, generator=g))
    x[0, :, 2:5, 2:5] = 0.5                                      # a plateau: ties inside windows
    xr = x.clone().requires_grad_(True)
    ref = F.max_pool2d(xr, k, s, p)
    Ho, Wo = ref.shape[2], ref.shape[3]
    dy = rnd(dt, torch.randn(B, C, Ho, Wo, generator=g))
    ref.backward(dy)
    xb, dyb = to_nhwc(x, dt), to_nhwc(dy, dt)
    out = torch.empty(B, Ho, Wo, C, dtype=TD[dt], device="cuda")
    idx = torch.empty(B * Ho * Wo * C, dtype=torch.uint8, device="cuda")
    dx = torch.full((B, H, W, C), 7.0, dtype=TD[dt], device="cuda")
    L.check(L.maxpool_fwd(dt, xb.data_ptr(), C, out.data_ptr(), C, idx.data_ptr(), B, H, W, C, k, s, p, st()), "maxpool_fwd")
    L.check(L.maxpool_bwd(dt, dyb.data_ptr(), C, idx.data_ptr(), dx.data_ptr(), C, B, H, W, C, k, s, p, st()), "maxpool_bwd")
    np.testing.assert_array_equal(to_nchw(out, dt, C).numpy(), ref.detach().numpy())
    tol = 0 if dt == F32 else 2e-2                                # bf16: sums of several routed gradients round once more
    np.testing.assert_allclose(to_nchw(dx, dt, C).numpy(), xr.grad.numpy(), rtol=tol, atol=tol)
    assert L.maxpool_fwd(dt, xb.data_ptr(), C, out.data_ptr(), C, idx.data_ptr(), B, H, W, C, 17, s, 8, st()) == -1


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("sc,H,W", [(3, 5, 7), (4, 8, 8), (1, 4, 4), (2, 6, 10)])
def test_upsample_generic(dt, sc, H, W):
    """nn.Upsample(scale_factor = stride, nearest) for any integer stride (models.py:86-88) and its backward (sum over the sc x sc copies)."""
    L = _lib.lib()
    B, C = 2, 24
    g = torch.Generator().manual_seed(sc * 10 + H)
    x = rnd(dt, torch.randn(B, C, H, W, generator=g))
    xr = x.clone().requires_grad_(True)
    ref = F.interpolate(xr, scale_factor=sc, mode="nearest")
    dy = rnd(dt, torch.randn(B, C, H * sc, W * sc, generator=g))
    ref.backward(dy)
    xb, dyb = to_nhwc(x, dt), to_nhwc(dy, dt)
    out = torch.empty(B, H * sc, W * sc, C, dtype=TD[dt], device="cuda")
    dx = torch.empty(B, H, W, C, dtype=TD[dt], device="cuda")
    L.check(L.upsample_fwd(dt, xb.data_ptr(), C, out.data_ptr(), C, B, H, W, C, sc, st()), "upsample_fwd")
    L.check(L.upsample_bwd(dt, dyb.data_ptr(), C, dx.data_ptr(), C, B, H, W, C, sc, st()), "upsample_bwd")
    np.testing.assert_array_equal(to_nchw(out, dt, C).numpy(), ref.detach().numpy())
    tol = 1e-5 if dt == F32 else 3e-2
    np.testing.assert_allclose(to_nchw(dx, dt, C).numpy(), xr.grad.numpy(), rtol=tol, atol=tol)


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("C,G", [(80, 13), (1, 13), (1, 7), (80, 5)])
def test_yolo_head_grad_tile_kernel_equals_scalar_kernel(dt, C, G):
    """mdcv_yolo_head_grad has a 16-pixel LDS-tile kernel (16-byte stores) and a per-element kernel for layouts the tile cannot take;
    a dlogits base that is not 16-byte aligned selects the second one.  Same bits from both, pad channels and class channels zero."""
    L = _lib.lib()
    torch.manual_seed(C * 100 + G)
    B, A, T_ = 3, 3, 6
    attrs = 5 + C
    cp = pad8(A * attrs)
    lg = (torch.randn(B, G, G, cp) * 1.5).to(TD[dt]).cuda()
    tg = torch.zeros(B, T_, 5)
    tg[:, :4, 0] = torch.randint(0, max(C, 1), (B, 4)).float()
    tg[:, :4, 1:3] = torch.rand(B, 4, 2) * 0.98 + 0.01
    tg[:, :4, 3:5] = torch.rand(B, 4, 2) * 0.3 + 0.02
    tg = tg.cuda()
    anchors = torch.tensor([[1.2, 1.9], [2.5, 3.8], [4.9, 6.1]], device="cuda")
    ws = torch.empty(int(L.yolo_head_workspace_bytes(B, A, G, G)), dtype=torch.uint8, device="cuda")
    out7 = torch.zeros(7, device="cuda")
    geo = (B, T_, A, C, G, G, 0.5, 2.0, 1.6, 25.0, 0.1)
    L.check(L.yolo_head_train(dt, lg.data_ptr(), cp, None, 0, cp, tg.data_ptr(), anchors.data_ptr(), *geo, ws.data_ptr(), out7.data_ptr(), None, st()))
    gs = torch.tensor([0.75], device="cuda")
    d_tile = torch.full((B * G * G * cp,), 7.0, dtype=TD[dt], device="cuda")
    L.check(L.yolo_head_grad(dt, lg.data_ptr(), cp, d_tile.data_ptr(), cp, cp, tg.data_ptr(), anchors.data_ptr(), *geo, ws.data_ptr(), gs.data_ptr(), st()))
    raw = torch.full((B * G * G * cp + 8,), 7.0, dtype=TD[dt], device="cuda")
    d_sc = raw[1:1 + B * G * G * cp]                                   # base + one element: no 16-byte stores possible
    assert d_sc.data_ptr() % 16 != 0
    L.check(L.yolo_head_grad(dt, lg.data_ptr(), cp, d_sc.data_ptr(), cp, cp, tg.data_ptr(), anchors.data_ptr(), *geo, ws.data_ptr(), gs.data_ptr(), st()))
    torch.cuda.synchronize()
    a, b = d_tile.float().view(B, G, G, cp).cpu(), d_sc.float().view(B, G, G, cp).cpu()
    assert torch.equal(a, b)
    assert float(raw[0]) == 7.0 and float(raw[1 + B * G * G * cp]) == 7.0
    live = torch.zeros(cp, dtype=torch.bool)
    for an in range(A):
        live[an * attrs:an * attrs + 5] = True
    assert float(a[..., ~live].abs().max()) == 0.0
    assert float(a[..., live].abs().max()) > 0.0


def test_batched_pack_equals_the_single_layer_pack():
    """mdcv_pack_weights_batched (one table-driven launch for all layers of a plan: every training step starts with it) against
    mdcv_pack_weights layer by layer, bit for bit: 1x1 / 3x3 / 7x7 kernels, channel counts that are not multiples of 8 or 64, tiles that
    straddle the channel padding, layers without a data-gradient operand, a bias copy, and parameters that sit at ODD float offsets of
    one flat buffer (the flat parameter buffer of a model has no padding between tensors)."""
    import struct
    L = _lib.lib()
    shapes = [(32, 3, 3, True, False), (255, 1024, 1, True, True), (64, 32, 3, True, False), (1024, 512, 3, True, False), (16, 3, 7, False, False),
              (128, 384, 1, True, False), (72, 40, 3, True, False), (7, 64, 1, True, True), (256, 128, 3, True, False), (512, 1024, 1, True, False)]
    g = torch.Generator().manual_seed(5)
    total = sum(co * ci * k * k + (co if b else 0) for co, ci, k, d, b in shapes) + 1
    flat = torch.randn(total, generator=g).cuda()
    off = 1                                                   # (odd offset for the first layer; the 255- and 7-element biases shift later ones)
    recs, outs, eq = [], [], 1
    for co, ci, k, need_d, has_b in shapes:
        n = co * ci * k * k
        w = flat[off:off + n]; off += n
        b = None
        if has_b:
            b = flat[off:off + co]; off += co
        cop, cip, kk = pad8(co), pad8(ci), k * k
        wf = torch.full((cop * kk * cip,), 7.0, dtype=torch.bfloat16, device="cuda")
        wd = torch.full((cip * kk * cop,), 7.0, dtype=torch.bfloat16, device="cuda") if need_d else None
        bp = torch.full((cop,), 7.0, device="cuda") if has_b else None
        wf0, wd0 = torch.zeros_like(wf), (torch.zeros_like(wd) if need_d else None)
        wc = w.clone()                                        # the single-layer reference reads an aligned copy
        L.check(L.pack_weights(BF16, wc.data_ptr(), wf0.data_ptr(), wd0.data_ptr() if need_d else None, co, ci, k, k, cop, cip, st()))
        recs.append(struct.pack("<QQQiiiiiiiiQQ", w.data_ptr(), wf.data_ptr(), wd.data_ptr() if need_d else 0, co, ci, kk, cop, cip, 0, 0, 0,
                                b.data_ptr() if has_b else 0, bp.data_ptr() if has_b else 0))
        outs.append((wf, wd, bp, wf0, wd0, b))
        eq = max(eq, (min(64, cip) * kk + 63) // 64)
    table = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).cuda()
    L.check(L.pack_weights_batched(BF16, table.data_ptr(), len(shapes), eq, st()))
    torch.cuda.synchronize()
    for (wf, wd, bp, wf0, wd0, b), shp in zip(outs, shapes):
        assert torch.equal(wf.view(torch.int16), wf0.view(torch.int16)), shp
        if wd is not None:
            assert torch.equal(wd.view(torch.int16), wd0.view(torch.int16)), shp
        if bp is not None:
            assert torch.equal(bp[:shp[0]], b), shp


@pytest.mark.parametrize("case", [(6400 * 3, 128, 128, 7), (1000, 32, 40, 8), (64 * 5 + 17, 256, 256, 7), (1, 64, 64, 1), (80 * 80 * 2, 128, 136, 7)], ids=str)
def test_head1x1_f32_logits(case):
    """mdcv_head1x1_f32 (round 6: KeypointNet's 1x1 head with fp32 logits out of bf16 features, keypoint_net.py:40,68): out[p][k] = bias[k] + sum_c x[p][c] w[k][c]
    in fp32 over the bf16 features exactly as stored -- ragged pixel counts, features inside a wider NHWC buffer (ldx > C), K = 1 / 7 / 8, padding columns zero,
    no bias, deterministic."""
    L = _lib.lib()
    M, C, ldx, K = case
    g = torch.Generator().manual_seed(M + C + K)
    xw = torch.randn(M, ldx, generator=g).to(torch.bfloat16)
    w = torch.randn(K, C, generator=g) * 0.2
    b = torch.randn(K, generator=g)
    off = ldx - C                                            # the features sit at the END of each pixel row of the wider buffer
    ref = xw[:, off:].float().double() @ w.double().t() + b.double()
    xd, wd, bd = xw.cuda(), w.cuda(), b.cuda()
    outs = []
    for bias in (bd, None, bd):
        out = torch.full((M, 8), float("nan"), dtype=torch.float32, device="cuda")
        L.check(L.head1x1_f32(xd.data_ptr() + off * 2, ldx, wd.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), M, C, K, st()), "head1x1_f32")
        torch.cuda.synchronize()
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[2])
    np.testing.assert_allclose(outs[0][:, :K].double().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
    np.testing.assert_allclose(outs[1][:, :K].double().numpy(), (ref - b.double()).numpy(), rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
    assert not bool(outs[0][:, K:].any()), "columns >= K must be exact zeros (the flat softmax reads a stride of 8)"
    for bad in ((M, 48, 48, 7), (M, C, ldx, 9), (M, C, C - 8, 7)):     # C % 32 != 0, K > 8, ldx < C
        assert L.head1x1_f32(xd.data_ptr(), bad[2], wd.data_ptr(), None, out.data_ptr(), bad[0], bad[1], bad[3], st()) != 0


@pytest.mark.parametrize("case", [(4, 512, 1024, 13, 13, 512, 1024), (2, 512, 1024, 13, 11, 500, 1020)], ids=str)
def test_wgrad_slab_free_form(case):
    """Round 6: where ONE split of 64 co x 32 ci tiles fills the chip (YOLOv3's 13^2 512 -> 1024 layers: 256 tiles) the LDS-ring weight gradient owns its
    slice of dW outright -- splits == 1, the slab buffer is never touched, no reduce launch: the kernel writes OIHW rows itself.  == torch; accumulate adds;
    padded channels are cropped; deterministic."""
    L = _lib.lib()
    dt = BF16
    B, Ci, Co, H, W, ci_real, co_real = case
    g = torch.Generator().manual_seed(Ci + H + ci_real)
    x = torch.randn(B, Ci, H, W, generator=g)
    dy = torch.randn(B, Co, H, W, generator=g)
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(rnd(dt, x), w, None, stride=1, padding=1).backward(rnd(dt, dy))
    ref = w.grad.numpy()[:co_real, :ci_real]
    xb, dyb = to_nhwc(x, dt), to_nhwc(dy, dt)
    splits = L.conv2d_wgrad_splits_geom(dt, B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 1, Co, Ci)
    assert splits == 1
    ws = torch.full((Co * 9 * Ci,), float("nan"), dtype=torch.float32, device="cuda")   # one slab's worth: it must stay untouched
    res = []
    for acc, init in ((0, 3.0), (0, -1.0), (1, 2.5)):
        dw = torch.full((co_real, ci_real, 3, 3), init, dtype=torch.float32, device="cuda")
        L.check(L.conv2d_wgrad(dt, dyb.data_ptr(), Co, xb.data_ptr(), Ci, ws.data_ptr(), splits, dw.data_ptr(), acc, B, H, W, Ci, ci_real,
                               H, W, Co, co_real, 3, 3, 1, 1, 1, st()), "wgrad")
        torch.cuda.synchronize()
        res.append(dw.cpu().numpy())
    assert bool(torch.isnan(ws).all())
    scale = max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(res[0], ref, rtol=2e-2, atol=2e-2 * scale)
    assert np.array_equal(res[0], res[1])
    np.testing.assert_allclose(res[2], res[0] + 2.5, rtol=1e-6, atol=1e-5)
