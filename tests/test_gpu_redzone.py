"""Memory-safety checks on the GPU: every output / scratch buffer of the conv kernels is allocated with a sentinel-filled guard
region behind it; a kernel that writes one element past what the C-ABI size queries promise fails here.  (Found in round 1: the
shift conv's last 256-row tile wrote a second BatchNorm partial row past the buffer whenever ceil(positions/128) was odd.)"""
import os
import sys

import numpy as np
import pytest
import torch

from mdcv import _lib

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
from variant_lib import VariantLib  # noqa: E402

pytestmark = pytest.mark.gpu
BF16 = _lib.BF16
SENT = 24680.0
GUARD = 8192


def st():
    return torch.cuda.current_stream().cuda_stream


def guarded(n, dtype):
    t = torch.full((n + GUARD,), SENT, dtype=dtype, device="cuda")
    return t


def check(t, n, what):
    tail = t[n:]
    bad = int((tail != SENT).sum())
    assert bad == 0, f"{what}: {bad} elements written past the end"


# (B, H, W, Cin, Cout, k, stride, dil): odd and even partial-row counts, every kernel family (shift / glds / narrow / stream / kh-shared)
GEOMS = [(32, 13, 13, 512, 1024, 3, 1, 1), (3, 52, 52, 128, 256, 3, 1, 1), (5, 26, 26, 256, 512, 3, 1, 1), (7, 13, 13, 128, 128, 3, 1, 1),
         (3, 26, 26, 128, 256, 3, 2, 1), (2, 31, 29, 64, 32, 1, 1, 1), (3, 40, 40, 16, 16, 3, 1, 2), (2, 33, 20, 32, 64, 3, 1, 2),
         (3, 80, 80, 64, 64, 3, 1, 1), (1, 17, 23, 64, 128, 3, 1, 1), (2, 19, 19, 128, 128, 3, 1, 1), (1, 64, 64, 8, 32, 3, 1, 1),
         (9, 9, 11, 128, 256, 3, 1, 1)]


@pytest.mark.parametrize("geom", GEOMS, ids=str)
def test_conv_kernels_stay_inside_their_buffers(geom):
    L = VariantLib()
    B, H, W, Ci, Co, k, s, dil = geom
    pad = dil * (k - 1) // 2
    Ho, Wo = (H + 2 * pad - dil * (k - 1) - 1) // s + 1, (W + 2 * pad - dil * (k - 1) - 1) // s + 1
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B * H * W * Ci, device="cuda", generator=g).to(torch.bfloat16)
    dy = torch.randn(B * Ho * Wo * Co, device="cuda", generator=g).to(torch.bfloat16)
    w = torch.randn(Co, Ci, k, k, device="cuda", generator=g) * 0.05
    wf = torch.zeros(Co * k * k * Ci, dtype=torch.bfloat16, device="cuda")
    wd = torch.zeros(Ci * k * k * Co, dtype=torch.bfloat16, device="cuda")
    L.check(L.pack_weights(BF16, w.data_ptr(), wf.data_ptr(), wd.data_ptr(), Co, Ci, k, k, Co, Ci, st()), "pack")
    # forward + BatchNorm partial rows
    rows = L.conv2d_stats_rows_geom(BF16, B, Ho, Wo, Ci, Co, k, k, s, pad, dil, Ci)
    ny, ns = B * Ho * Wo * Co, rows * 2 * Co
    y, stt = guarded(ny, torch.bfloat16), guarded(ns, torch.float32)
    L.check(L.conv2d(BF16, 0, x.data_ptr(), Ci, wf.data_ptr(), y.data_ptr(), Co, None, None, 0, stt.data_ptr(), B, H, W, Ci, Ho, Wo, Co,
                     k, k, s, pad, dil, st()), "conv fwd")
    torch.cuda.synchronize()
    check(y, ny, "forward output"); check(stt, ns, "BatchNorm partial rows")
    assert int((stt[:ns] == SENT).sum()) == 0, "a promised partial row was not written"
    # the partial rows sum to the column sums of the output
    ysum = y[:ny].float().reshape(-1, Co).sum(0).cpu().numpy()
    psum = stt[:ns].reshape(rows, 2, Co)[:, 0].sum(0).cpu().numpy()
    np.testing.assert_allclose(psum, ysum, rtol=2e-2, atol=2e-2 * max(1.0, float(np.abs(ysum).max())))
    # data gradient
    nx = B * H * W * Ci
    dx = guarded(nx, torch.bfloat16)
    L.check(L.conv2d(BF16, 1, dy.data_ptr(), Co, wd.data_ptr(), dx.data_ptr(), Ci, None, None, 0, None, B, Ho, Wo, Co, H, W, Ci,
                     k, k, s, pad, dil, st()), "conv dgrad")
    torch.cuda.synchronize()
    check(dx, nx, "data gradient")
    # weight gradient: every kernel the dispatch can pick
    for variant in (0, 8, 9):
        L.conv2d_wgrad_set_variant(variant)
        try:
            splits = L.conv2d_wgrad_splits_geom(BF16, B, H, W, Ci, Ho, Wo, Co, k, k, s, pad, dil, Co, Ci)
            nws, ndw = splits * Co * k * k * Ci, Co * Ci * k * k
            ws, dw = guarded(nws, torch.float32), guarded(ndw, torch.float32)
            L.check(L.conv2d_wgrad(BF16, dy.data_ptr(), Co, x.data_ptr(), Ci, ws.data_ptr(), splits, dw.data_ptr(), 0, B, H, W, Ci, Ci,
                                   Ho, Wo, Co, Co, k, k, s, pad, dil, st()), "wgrad")
            torch.cuda.synchronize()
            check(ws, nws, f"wgrad slabs (variant {variant})"); check(dw, ndw, f"weight gradient (variant {variant})")
            assert torch.isfinite(dw[:ndw]).all()
        finally:
            L.conv2d_wgrad_set_variant(0)


def _full_step_redzones(monkeypatch, build, step):
    from mdcv import engine
    monkeypatch.setattr(engine, "_REDZONE", 4096)
    model = build()
    for _ in range(2):
        step(model)
    torch.cuda.synchronize()
    plans = list(model._plans.values())
    assert plans and all(p.redzones for p in plans)
    return sum(p.check_redzones() for p in plans), sum(len(p.redzones) for p in plans)


def test_yolov3_train_step_stays_inside_every_plan_buffer(monkeypatch, tmp_path):
    """Full yolo_baseline at 416x416, batch 32 (odd partial-row counts at 52/26/13), forward + backward + FusedAdam: the guard
    bytes behind every activation / gradient / statistics / operand / scratch buffer of the plan are intact."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from mdcv.yolo.models import Darknet
    from mdcv.optim import FusedAdam
    cfg = bench.write_yolo_cfg(str(tmp_path))
    g = torch.Generator().manual_seed(1)
    x = torch.rand(32, 3, 416, 416, generator=g).cuda()
    tg = bench.synth_targets(32, 16, g).cuda()
    state = {}

    def build():
        cwd = os.getcwd()
        os.chdir(tmp_path)
        try:
            torch.manual_seed(0)
            net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision="bf16").cuda().train()
        finally:
            os.chdir(cwd)
        state["opt"] = FusedAdam(net, lr=1e-3)
        return net

    def step(net):
        state["opt"].zero_grad()
        net(x, tg)[0].sum().backward()
        state["opt"].step()
    bad, n = _full_step_redzones(monkeypatch, build, step)
    assert bad == 0, f"{bad} of {n} plan buffers were written past their end"


def test_rektnet_train_step_stays_inside_every_plan_buffer(monkeypatch):
    import contextlib
    import io
    from mdcv.rektnet.keypoint_net import KeypointNet
    from mdcv.rektnet.cross_ratio_loss import CrossRatioLoss
    from mdcv.optim import FusedAdam
    with contextlib.redirect_stdout(io.StringIO()):
        crit = CrossRatioLoss("l1_softargmax", True, 0.05, 0.05)
    g = torch.Generator().manual_seed(2)
    B = 37                                            # odd everything
    x = torch.rand(B, 3, 80, 80, generator=g).cuda()
    thm = torch.rand(B, 7, 80, 80, generator=g).cuda()
    tp = torch.rand(B, 7, 2, generator=g).cuda() * 0.9
    state = {}

    def build():
        net = KeypointNet(7, (80, 80), precision="bf16").cuda().train()
        state["opt"] = FusedAdam(net, lr=1e-2)
        return net

    def step(net):
        state["opt"].zero_grad()
        hm, pts = net(x)
        crit(hm, pts, thm, tp)[2].backward()
        state["opt"].step()
    bad, n = _full_step_redzones(monkeypatch, build, step)
    assert bad == 0, f"{bad} of {n} plan buffers were written past their end"


@pytest.mark.parametrize("B", [5, 16, 31, 32])
def test_yolov3_train_step_reads_no_uninitialised_plan_buffer(monkeypatch, tmp_path, B):
    """MDCV_POISON mode: every plan buffer that is not zero-initialised starts as NaN.  A kernel that reads a partial row / scratch
    element nobody wrote turns the loss or a gradient into NaN.  Batch sizes with odd and even tile / partial-row counts."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from mdcv import engine
    from mdcv.yolo.models import Darknet
    monkeypatch.setattr(engine, "_POISON", True)
    cfg = bench.write_yolo_cfg(str(tmp_path))
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        torch.manual_seed(0)
        net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision="bf16").cuda().train()
    finally:
        os.chdir(cwd)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(B, 3, 416, 416, generator=g).cuda()
    tg = bench.synth_targets(B, 16, g).cuda()
    out = net(x, tg)
    out[0].sum().backward()
    assert all(bool(torch.isfinite(o)) for o in out)
    gflat = net.flat_parameters()[1]
    assert bool(torch.isfinite(gflat).all()), int((~torch.isfinite(gflat)).sum())


@pytest.mark.parametrize("B", [3, 37, 256])
def test_rektnet_train_step_reads_no_uninitialised_plan_buffer(monkeypatch, B):
    import contextlib
    import io
    from mdcv import engine
    from mdcv.rektnet.keypoint_net import KeypointNet
    from mdcv.rektnet.cross_ratio_loss import CrossRatioLoss
    monkeypatch.setattr(engine, "_POISON", True)
    with contextlib.redirect_stdout(io.StringIO()):
        crit = CrossRatioLoss("l1_softargmax", True, 0.05, 0.05)
    g = torch.Generator().manual_seed(2)
    x = torch.rand(B, 3, 80, 80, generator=g).cuda()
    thm = torch.rand(B, 7, 80, 80, generator=g).cuda()
    tp = torch.rand(B, 7, 2, generator=g).cuda() * 0.9
    net = KeypointNet(7, (80, 80), precision="bf16").cuda().train()
    hm, pts = net(x)
    loss = crit(hm, pts, thm, tp)[2]
    loss.backward()
    assert bool(torch.isfinite(loss))
    gflat = net.flat_parameters()[1]
    assert bool(torch.isfinite(gflat).all()), int((~torch.isfinite(gflat)).sum())


# (B, channels of dx, H, W of dx, channels of dy, k, stride, pad): the last 256-row tile's second group starts past the last pixel
FUSED_GEOMS = [(31, 256, 52, 52, 128, 1, 1, 0), (3, 128, 52, 52, 256, 3, 1, 1), (5, 64, 26, 26, 128, 3, 2, 1), (7, 512, 13, 13, 1024, 3, 1, 1),
               (33, 128, 13, 13, 256, 3, 1, 1)]


@pytest.mark.parametrize("geom", FUSED_GEOMS, ids=str)
def test_fused_dgrad_partial_rows_written_exactly(geom):
    """mdcv_conv2d_dgrad_bnsums writes every row mdcv_conv2d_dgrad_bnsums_rows promises and nothing behind them."""
    L = _lib.lib()
    B, Ci, H, W, Co, k, s, p = geom
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    rows = L.conv2d_dgrad_bnsums_rows(BF16, B, Ho, Wo, Co, H, W, Ci, k, k, s, p, 1, Co)
    assert rows > 0
    assert L.conv2d_dgrad_bnsums_rows(_lib.F32, B, Ho, Wo, Co, H, W, Ci, k, k, s, p, 1, Co) == 0      # bf16 only
    g = torch.Generator(device="cuda").manual_seed(3)
    dy = torch.randn(B * Ho * Wo * Co, device="cuda", generator=g).to(torch.bfloat16)
    wd = (torch.randn(Ci * k * k * Co, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    y = torch.randn(B * H * W * Ci, device="cuda", generator=g).to(torch.bfloat16)
    nx = B * H * W * Ci
    dx = guarded(nx, torch.bfloat16)
    sc, sh, mean = torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
    part = torch.full(((rows + 64) * 2 * Ci,), float("nan"), device="cuda")
    L.check(L.conv2d_dgrad_bnsums(BF16, dy.data_ptr(), Co, wd.data_ptr(), dx.data_ptr(), Ci, None, 0, B, Ho, Wo, Co, H, W, Ci, k, k, s, p, 1,
                                  y.data_ptr(), Ci, sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), 1, 0.1, part.data_ptr(), st()), "fused dgrad")
    torch.cuda.synchronize()
    check(dx, nx, "data gradient")
    assert not bool(torch.isnan(part[:rows * 2 * Ci]).any()), "a promised partial row was not written"
    assert bool(torch.isnan(part[rows * 2 * Ci:]).all()), "a partial row was written past the buffer"


@pytest.mark.parametrize("B,N,C,dense", [(32, 10647, 80, False), (3, 22743, 80, True), (5, 507, 1, True)])
def test_detect_post_stays_inside_workspace_and_outputs(B, N, C, dense):
    """validate.py post-processing at real sizes: guard bytes behind the workspace and every output array stay intact."""
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(5)
    pred = torch.rand(B, N, 5 + C, device="cuda", generator=g)
    pred[..., :4] *= 400.0
    pred[..., 4] = torch.rand(B, N, device="cuda", generator=g) * (1.0 if dense else 0.55)      # dense: thousands above the threshold
    tg = torch.zeros(B, 16, 5, device="cuda")
    tg[:, :6, 1:] = torch.rand(B, 6, 4, device="cuda", generator=g) * 0.5 + 0.2
    k = 200
    RZ = 4096

    def gbuf(nbytes):
        raw = torch.full((nbytes + RZ,), 0xA5, dtype=torch.uint8, device="cuda")
        return raw, nbytes
    sizes = dict(boxes=B * k * 4 * 4, prob=B * k * 4, cls=B * k * 4, index=B * k * 8, correct=B * k, count=B * 4, stats=B * 4 * 4,
                 ws=int(L.detect_post_workspace_bytes(B, N)))
    bufs = {n: gbuf(sz) for n, sz in sizes.items()}
    p = {n: raw.data_ptr() for n, (raw, _) in bufs.items()}
    L.check(L.detect_post(pred.data_ptr(), B, N, C, tg.data_ptr(), 16, 0.5, 0.4, 0.5, 416.0, 416.0, k, p["boxes"], p["prob"], p["cls"],
                          p["index"], p["correct"], p["count"], p["stats"], p["ws"], st()), "detect_post")
    torch.cuda.synchronize()
    for n, (raw, sz) in bufs.items():
        assert int((raw[sz:] != 0xA5).sum()) == 0, f"{n}: written past the end"
    count = bufs["count"][0][:B * 4].view(torch.int32)
    assert int(count.min()) >= 0 and int(count.max()) <= k


def test_tiny_cfg_train_and_608_eval_under_poison_and_redzones(monkeypatch, tmp_path):
    """The second shipped topology (max-pool sections, two heads) at 416x416 batch 16, and the eval-mode forward of yolo_baseline at
    608x608: NaN-poisoned uninitialised buffers and guard bytes behind every plan buffer at the same time."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from mdcv import engine
    from mdcv.yolo.models import Darknet
    from test_gpu_models import write_tiny_cfg
    monkeypatch.setattr(engine, "_POISON", True)
    monkeypatch.setattr(engine, "_REDZONE", 4096)
    g = torch.Generator().manual_seed(7)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        torch.manual_seed(0)
        tiny = Darknet(write_tiny_cfg(str(tmp_path), 416, 80), 2.0, 1.6, 25.0, 0.1, True, precision="bf16").cuda().train()
        full = Darknet(bench.write_yolo_cfg(str(tmp_path), size=608), 2.0, 1.6, 25.0, 0.1, True, precision="bf16").cuda().eval()
    finally:
        os.chdir(cwd)
    x = torch.rand(16, 3, 416, 416, generator=g).cuda()
    tg = bench.synth_targets(16, 16, g).cuda()
    out = tiny(x, tg)
    out[0].sum().backward()
    assert all(bool(torch.isfinite(o)) for o in out)
    assert bool(torch.isfinite(tiny.flat_parameters()[1]).all())
    with torch.no_grad():
        det = full(torch.rand(4, 3, 608, 608, generator=g).cuda())
    assert det.shape == (4, 22743, 85) and bool(torch.isfinite(det).all())
    torch.cuda.synchronize()
    for net in (tiny, full):
        for plan in net._plans.values():
            assert plan.redzones and plan.check_redzones() == 0
