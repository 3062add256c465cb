"""-m gpu: robustness of the drop-in classes around the hot path — copies of a model with cached launch plans, the plan cache
itself, the pipelined optimizer across plan changes, bad labels, unsupported cfg shapes, the stand-alone ResNet block."""
import contextlib
import copy
import io
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
T = torch.from_numpy


def make_mini(precision, cfg="mini.cfg"):
    from mdcv.yolo.models import Darknet
    cwd = os.getcwd()
    os.chdir(os.path.join(G, "mini"))
    try:
        net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, False, precision=precision)
        net.load_weights("mini.weights", net.get_start_weight_dim())
    finally:
        os.chdir(cwd)
    return net.cuda()


def test_deepcopy_after_forward_keeps_both_models_working():
    """RektNet/train_eval.py:99 `best_model = copy.deepcopy(model)` after an epoch of forwards: the copy must not carry launch plans
    (ctypes pointers), must own its parameters, and both models must keep working."""
    from mdcv.rektnet.keypoint_net import KeypointNet
    from mdcv.rektnet.cross_ratio_loss import CrossRatioLoss
    from mdcv.optim import FusedAdam
    torch.manual_seed(1)
    net = KeypointNet(7, (80, 80), precision="fp32").cuda().train()
    with contextlib.redirect_stdout(io.StringIO()):
        crit = CrossRatioLoss("l1_softargmax", True, 0.05, 0.05)
    g = torch.Generator().manual_seed(2)
    x = torch.rand(4, 3, 80, 80, generator=g).cuda()
    tp = (torch.rand(4, 7, 2, generator=g) * 0.9).cuda()
    opt = FusedAdam(net, lr=1e-2)

    def step(m, o):
        o.zero_grad()
        hm, pts = m(x)
        loss = crit(hm, pts, None, tp)[2]
        loss.backward()
        o.step()
        return float(loss)
    step(net, opt)
    best = copy.deepcopy(net)
    assert best._plans == {} and len(net._plans) >= 1
    for (k, a), (_, b) in zip(net.state_dict().items(), best.state_dict().items()):
        assert torch.equal(a, b), k
    frozen = {k: v.clone() for k, v in best.state_dict().items()}
    l1 = step(net, opt)                                          # the original trains on; the copy must not move
    for k, v in best.state_dict().items():
        assert torch.equal(v, frozen[k]), k
    net.eval(); best.eval()
    with torch.no_grad():
        pa, pb = net(x)[1], best(x)[1]
    assert not torch.equal(pa, pb)                               # different parameters by now
    best.train()
    l2 = step(best, FusedAdam(best, lr=1e-2))                    # and the copy trains: same parameters + batch as the original's 2nd step
    assert abs(l1 - l2) <= 1e-5 * abs(l1)


def test_plan_cache_is_bounded(monkeypatch):
    from mdcv.rektnet.keypoint_net import KeypointNet
    monkeypatch.setattr(KeypointNet, "max_plans", 2)
    net = KeypointNet(7, (80, 80), precision="bf16").cuda().eval()
    g = torch.Generator().manual_seed(0)
    ref = {}
    with torch.no_grad():
        for b in (1, 2, 3, 2, 1, 3):
            x = torch.rand(b, 3, 80, 80, generator=torch.Generator().manual_seed(b)).cuda()
            pts = net(x)[1]
            assert len(net._plans) <= 2
            if b in ref:
                assert torch.equal(ref[b], pts)                  # a rebuilt plan computes the same thing
            ref[b] = pts.clone()
    net.release_plans()
    assert net._plans == {}
    del g


def test_pipelined_adam_across_plan_changes_is_bit_identical():
    """FusedAdam(pipeline=True) defers late parameter-group updates into the NEXT forward's launch list.  When that next forward
    runs through a different plan (a ragged last batch, an eval pass) the deferred updates must still land before anything reads the
    parameters or overwrites the gradient buffer: same losses and parameters as the plain optimizer, bit for bit."""
    from mdcv.rektnet.keypoint_net import KeypointNet
    from mdcv.rektnet.cross_ratio_loss import CrossRatioLoss
    from mdcv.optim import FusedAdam
    with contextlib.redirect_stdout(io.StringIO()):
        crit = CrossRatioLoss("l1_softargmax", True, 0.05, 0.05)
    g = torch.Generator().manual_seed(9)
    sizes = [8, 8, 5, 8, 5, 3, 8]
    xs = [torch.rand(b, 3, 80, 80, generator=g).cuda() for b in sizes]
    tps = [(torch.rand(b, 7, 2, generator=g) * 0.9).cuda() for b in sizes]
    res = []
    for pipe in (False, True):
        torch.manual_seed(3)
        net = KeypointNet(7, (80, 80), precision="bf16").cuda().train()
        opt = FusedAdam(net, lr=1e-2, pipeline=pipe)
        losses = []
        for i, (x, tp) in enumerate(zip(xs, tps)):
            opt.zero_grad()
            hm, pts = net(x)
            loss = crit(hm, pts, None, tp)[2]
            loss.backward()
            opt.step()
            if i == 3:
                sd = {k: v.clone() for k, v in net.state_dict().items()}      # load_state_dict between a step and the next forward
                net.load_state_dict(sd)
            losses.append(float(loss))
        res.append((losses, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}))
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


def test_out_of_grid_target_raises_index_error_on_the_fused_path():
    """cx == 1.0 puts the target in grid column G: the reference's build_targets raises IndexError (utils/utils.py:262).  The fused
    training head flags it; Darknet raises at the start of the next forward (or on plan.check_targets()), the stand-alone YOLOLayer at once."""
    z = np.load(os.path.join(G, "mini_darknet.npz"))
    net = make_mini("fp32").train()
    x, tg = T(z["x"]).cuda(), T(z["targets"]).cuda().clone()
    net(x, tg)[0].sum().backward()
    net(x, tg)                                                   # clean labels: no complaint
    bad = tg.clone()
    bad[1, 0] = torch.tensor([0.0, 1.0, 0.5, 0.2, 0.2])
    net(x, bad)
    with pytest.raises(IndexError):
        net(x, tg)
    net(x, tg)                                                   # the flag was consumed; clean labels pass again
    net(x, bad)
    plan = [p for p in net._plans.values() if p.has_bwd][0]
    with pytest.raises(IndexError):
        plan.check_targets()
    from mdcv.yolo.models import YOLOLayer
    yl = [m[0] for m in net.module_list if isinstance(m[0], YOLOLayer)][0]
    sample = torch.randn(2, 3 * 6, 8, 8, device="cuda", requires_grad=True)
    with pytest.raises(IndexError):
        yl(sample, bad[:2])


def test_out_of_grid_target_raises_before_the_parameter_update():
    """The reference raises inside build_targets BEFORE any update (utils/utils.py:262).  On the plan path the flag copy recorded behind
    the forward is looked at again at the start of backward (so optimizer.step() is never reached) and when the plan is dropped (the last
    batch of a run must not lose it)."""
    from mdcv.optim import FusedAdam
    z = np.load(os.path.join(G, "mini_darknet.npz"))
    net = make_mini("fp32").train()
    opt = FusedAdam(net, lr=1e-3)
    x, tg = T(z["x"]).cuda(), T(z["targets"]).cuda().clone()
    opt.zero_grad()
    net(x, tg)[0].sum().backward()
    opt.step()
    before = net.flat_parameters()[0].clone()
    bad = tg.clone()
    bad[0, 0] = torch.tensor([0.0, 0.5, 1.0, 0.2, 0.2])             # cy == 1.0
    opt.zero_grad()
    loss = net(x, bad)[0].sum()
    with pytest.raises(IndexError):
        loss.backward()
    assert torch.equal(net.flat_parameters()[0], before)            # no step was taken with the bad label
    net(x, bad)                                                     # ... and a flag pending on a plan that is being dropped still surfaces
    with pytest.raises(IndexError):
        net.release_plans()


def test_in_place_parameter_edit_after_a_pipelined_step_is_seen_by_the_next_forward():
    """FusedAdam(pipeline=True) re-packs the conv operands ahead of the next forward; user code that edits a parameter in place between
    step() and that forward (nn.init.*, clamp_ under no_grad, EMA copy-back) must force a re-pack: compare with the plain optimizer."""
    from mdcv.optim import FusedAdam
    z = np.load(os.path.join(G, "mini_darknet.npz"))
    x, tg = T(z["x"]).cuda(), T(z["targets"]).cuda()
    losses = []
    for pipe in (False, True):
        torch.manual_seed(5)
        net = make_mini("bf16").train()
        opt = FusedAdam(net, lr=1e-3, pipeline=pipe)
        out = []
        for it in range(3):
            opt.zero_grad()
            loss = net(x, tg)[0].sum()
            loss.backward()
            opt.step()
            out.append(float(loss))
            if it == 1:
                opt.synchronize()
                with torch.no_grad():
                    net.module_list[0][0].weight.mul_(0.5)           # in-place edit of the first conv's weights
        losses.append(out)
    assert losses[0] == losses[1], losses
    assert abs(losses[0][2] - losses[0][1]) > 1e-6                   # (the edit does change the loss)


def test_route_with_unaligned_source_is_rejected(tmp_path):
    """A concat whose non-last source is not a multiple of 8 channels wide would put a pad hole in the middle of the consumer's input
    channels: refuse instead of computing with misaligned weights."""
    from mdcv.yolo.models import Darknet
    src = open(os.path.join(G, "mini", "mini.cfg")).read()
    head = src[:src.index("[convolutional]")]
    body = ("[convolutional]\nfilters=12\nsize=3\nstride=1\n\n[convolutional]\nfilters=16\nsize=3\nstride=1\n\n[route]\nlayers=-2,-1\n\n"
            "[convolutional]\nfilters=preyolo\nsize=1\nstride=1\n\n[yolo]\n")
    os.makedirs(tmp_path / "dataset")
    (tmp_path / "dataset" / "train.csv").write_text(open(os.path.join(G, "mini", "dataset", "train.csv")).read())
    (tmp_path / "odd.cfg").write_text(head + body)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        net = Darknet("odd.cfg", 2.0, 1.6, 25.0, 0.1, False, precision="fp32").cuda().train()
    finally:
        os.chdir(cwd)
    with pytest.raises(NotImplementedError):
        net(torch.rand(2, 3, 64, 64).cuda(), torch.zeros(2, 2, 5).cuda())


def _ref_block(x, sd, train):
    """RektNet/resnet.py:22-27 with stock torch ops on the CPU in fp32 (the comparison the kernels are held to)."""
    def bn(t, p):
        return F.batch_norm(t, sd[p + ".running_mean"].clone(), sd[p + ".running_var"].clone(), sd[p + ".weight"], sd[p + ".bias"], train, 0.1, 1e-5)
    c1 = F.relu(bn(F.conv2d(x, sd["conv1.weight"], sd["conv1.bias"], 1, 2, 2), "bn1"))
    c2 = bn(F.conv2d(c1, sd["conv2.weight"], sd["conv2.bias"], 1, 1, 1), "bn2")
    sc = bn(F.conv2d(x, sd["shortcut_conv.weight"], sd["shortcut_conv.bias"]), "shortcut_bn")
    return F.relu(sc + c2)


@pytest.mark.parametrize("cin,cout", [(16, 32), (32, 32)])
def test_standalone_resnet_block_vs_torch_fp32(cin, cout):
    """`ResNet(in, out)(x)` on its own (RektNet/resnet.py:22-27): forward, dx and every parameter gradient against stock torch ops in
    fp32 on the CPU; fp32 kernels 1e-3 (norm-scaled), bf16 kernels by direction (cosine) and norm."""
    from mdcv.rektnet.resnet import ResNet
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, cin, 40, 40, generator=g)
    w = torch.randn(4, cout, 40, 40, generator=g)
    for precision in ("fp32", "bf16"):
        torch.manual_seed(5)
        blk = ResNet(cin, cout, precision=precision)
        with torch.no_grad():
            for m in blk.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.uniform_(0.8, 1.2); m.bias.uniform_(-0.1, 0.1)
        sd = {k: v.detach().clone().requires_grad_(v.dtype == torch.float32 and "running" not in k) for k, v in blk.state_dict().items()}
        xr = x.clone().requires_grad_(True)
        ref = _ref_block(xr, sd, True)
        (ref * w).sum().backward()
        blk = blk.cuda().train()
        xg = x.clone().cuda().requires_grad_(True)
        out = blk(xg)
        assert out.shape == ref.shape and out.dtype == torch.float32
        (out * w.cuda()).sum().backward()
        f32 = precision == "fp32"

        def check(a, b, name):
            a, b = a.detach().double().cpu().flatten(), b.detach().double().flatten()
            if f32:
                assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()) + 1e-6, name
            else:
                cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
                assert cos > 0.99 and abs(float(a.norm() / b.norm()) - 1) < 0.05, (name, cos)
        check(out, ref, "out")
        check(xg.grad, xr.grad, "dx")
        for n, p in blk.named_parameters():
            if n.endswith("conv1.bias") or n.endswith("conv2.bias") or n.endswith("shortcut_conv.bias"):
                assert float(p.grad.abs().max()) < 2e-3           # bias in front of a BatchNorm: mathematically zero
            else:
                check(p.grad, sd[n].grad, n)
        blk.eval()
        with torch.no_grad():
            ev = blk(x.cuda())
        check(ev, _ref_block(x, {k: v.detach() for k, v in blk.cpu().state_dict().items()}, False), "eval out")


def test_resnet_block_refuses_what_it_cannot_differentiate_or_would_break():
    """Stand-alone ResNet block: eval mode with gradients enabled has no backward on the HIP path (refuse, do not return a graph-less
    tensor); a block that lives inside a flattened KeypointNet must not re-flatten its parameters out of the parent."""
    import copy
    from mdcv.rektnet.keypoint_net import KeypointNet
    from mdcv.rektnet.resnet import ResNet
    blk = ResNet(16, 32, precision="fp32").cuda().eval()
    x = torch.rand(2, 16, 20, 20, device="cuda")
    with pytest.raises(NotImplementedError):
        blk(x)
    with torch.no_grad():
        assert tuple(blk(x).shape) == (2, 32, 20, 20)
    net = KeypointNet(7, (80, 80), precision="fp32").cuda().train()
    img = torch.rand(2, 3, 80, 80, device="cuda")
    net(img)
    nplans = len(net._plans)
    with pytest.raises(RuntimeError, match="flat buffer"):
        net.res1(torch.rand(2, 16, 80, 80, device="cuda"))
    assert len(net._plans) == nplans and net._flat_ok()
    alone = copy.deepcopy(net.res1).train()                      # a copy owns its parameters and runs on its own
    assert tuple(alone(torch.rand(2, 16, 80, 80, device="cuda")).shape) == (2, 16, 80, 80)


def test_side_stream_is_shared_and_runs_beside_the_main_stream():
    """HIP multiplexes streams onto a few hardware queues; a plan whose fresh side stream shared the main stream's queue ran its weight
    gradients serially (17.0 instead of 14.5 ms per YOLOv3 step for about one model in eight built in one process, scripts/bimodal_probe.py).
    The side stream is now one per device, shared by every plan, and checked with two spin kernels to overlap with the current stream."""
    from mdcv import engine
    dev = torch.device("cuda", 0)
    cur = torch.cuda.current_stream(dev)
    s1 = engine.side_stream(dev)
    assert s1.cuda_stream != cur.cuda_stream and engine._runs_beside([cur], s1)
    for _ in range(40):
        torch.cuda.Stream(device=dev)                    # walk torch's stream pool: later requests must still return the checked stream
    assert engine.side_stream(dev) is s1
    nets = [make_mini("bf16").train() for _ in range(3)]
    z = np.load(os.path.join(G, "mini_darknet.npz"))
    x, tg = T(z["x"]).cuda(), T(z["targets"]).cuda()
    for net in nets:
        net(x, tg)[0].sum().backward()
        plan = [p for p in net._plans.values() if p.has_bwd][0]
        assert plan.side() is s1
    c = engine.checked_stream(dev, [cur, s1], "comm")
    assert c.cuda_stream not in (cur.cuda_stream, s1.cuda_stream) and engine._runs_beside([cur, s1], c)
