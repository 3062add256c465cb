"""Pin the CPU oracle (oracle/) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import glob
import os
import numpy as np
import pytest
import torch

from oracle import yolo_oracle as yo
from oracle import rektnet_oracle as ro

G = os.path.join(os.path.dirname(__file__), "golden")
T = torch.from_numpy


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "bt_*.npz"))), ids=os.path.basename)
def test_build_targets_bit_exact(path):
    z = np.load(path)
    tgt = T(z["target"])
    keep = tgt.clone()
    out = yo.build_targets(tgt, T(z["anchors"]), z["anchors"].shape[0], int(z["C"]), int(z["Gh"]), int(z["Gw"]), float(z["thr"]))
    assert torch.equal(tgt, keep)                                  # input not mutated
    for k, v in zip(("mask", "conf_mask", "tx", "ty", "tw", "th", "tconf", "tcls"), out):
        assert v.dtype == T(z[k]).dtype, k
        assert np.array_equal(v.numpy(), z[k]), k                  # 0 ulp, masks/indices exact


def test_build_targets_quirks():
    z = load("bt_leak.npz")
    cm = z["conf_mask"]
    # Q1: the ignore cell of image 0 is zeroed in image 1 as well (all anchors not holding a positive)
    zero_cells = np.argwhere(cm[1] == 0)
    assert len(zero_cells) > 0
    z = load("bt_empty.npz")                                      # Q3 phantom positive at [0, a, 0, 0]
    assert z["mask"][0].sum() == 1 and z["mask"][0][:, 0, 0].sum() == 1
    assert np.isclose(z["tw"][0].min(), np.log(np.float32(1e-16)), rtol=1e-6)


def test_bbox_iou():
    z = load("bbox_iou.npz")
    assert np.array_equal(yo.corner_iou_plus1(T(z["c1"]), T(z["c2"])).numpy(), z["iou_corner"])
    assert np.array_equal(yo.center_iou_plus1(T(z["b1"]), T(z["b2"])).numpy(), z["iou_center"])


@pytest.mark.parametrize("name", ["yolo_layer_c1_g13.npz", "yolo_layer_c80_g13.npz", "yolo_layer_c1_g26.npz"])
def test_yolo_layer(name):
    z = load(name)
    s = T(z["sample"]).clone().requires_grad_(True)
    anchors = [tuple(a) for a in z["anchors_px"].tolist()]
    loss, parts = yo.yolo_layer(s, anchors, int(z["C"]), int(z["cfg_h"]), T(z["targets"]))
    loss.backward()
    np.testing.assert_allclose(loss.item(), z["loss"], rtol=1e-6)
    np.testing.assert_allclose(parts.numpy(), z["parts"], rtol=1e-6)
    np.testing.assert_allclose(s.grad.numpy(), z["dsample"], rtol=1e-5, atol=1e-9)
    ev = yo.yolo_layer(s.detach(), anchors, int(z["C"]), int(z["cfg_h"]))
    np.testing.assert_allclose(ev.numpy(), z["eval_out"], rtol=1e-6, atol=1e-7)


def ref_key_to_oracle(k):
    # module_list.3.conv_3.weight -> conv3.weight ; module_list.3.batch_norm_3.running_mean -> bn3.running_mean
    _, i, mod, leaf = k.split(".")
    return ("conv" if mod.startswith("conv") else "bn") + f"{i}.{leaf}"


@pytest.fixture(scope="module")
def mini():
    cwd = os.getcwd()
    os.chdir(os.path.join(G, "mini"))
    try:
        anchors = yo.read_anchor_row("dataset/train.csv")
        net = yo.DarknetOracle("mini.cfg", anchors=anchors)
        net.load_weights("mini.weights", [18, 18])
    finally:
        os.chdir(cwd)
    return net


def test_mini_darknet_loss_grads_running(mini):
    z = load("mini_darknet.npz")
    net = mini
    base = {k: v.clone() for k, v in net.params.items()}
    for k in net.trainable():
        net.params[k] = base[k].clone().requires_grad_(True)
    losses = net.forward(T(z["x"]), T(z["targets"]), bn_train=True)
    losses[0].sum().backward()
    np.testing.assert_allclose(torch.stack([l.detach() for l in losses]).numpy(), z["losses"], rtol=2e-5)
    names = [str(n) for n in z["grad_names"]]
    for n, gn in zip(names, z["grad_norm"]):
        g = net.params[ref_key_to_oracle(n)].grad
        np.testing.assert_allclose(float(g.double().norm()), gn, rtol=2e-4, atol=1e-7, err_msg=n)
    for k in z.files:
        if k.startswith("grad::"):
            g = net.params[ref_key_to_oracle(k[6:])].grad
            np.testing.assert_allclose(g.numpy(), z[k], rtol=1e-3, atol=2e-6, err_msg=k)
        if k.startswith("run::"):
            np.testing.assert_allclose(net.params[ref_key_to_oracle(k[5:])].detach().numpy(), z[k], rtol=1e-5, atol=1e-7, err_msg=k)
    # the fixture's eval pass ran AFTER the train step, i.e. with the updated running stats
    for k, v in base.items():
        if "running" not in k:
            net.params[k] = v
    with torch.no_grad():
        ev = net.forward(T(z["x"]), None, bn_train=False)
    np.testing.assert_allclose(ev.numpy(), z["eval_out"], rtol=1e-4, atol=1e-5)
    for k, v in base.items():
        net.params[k] = v


@pytest.mark.parametrize("tag", ["mini_tiny", "mini_relu"])
def test_maxpool_and_relu_cfgs_loss_grads_running_eval(tag, tmp_path):
    """The cfg features mini.cfg does not reach, against vectors the REFERENCE produced (make_golden.py yolo): mini_tiny.cfg = yolo_baseline_tiny.cfg
    at toy width with both max-pool forms (MaxPool2d(2, 2); ZeroPad2d((0, 1, 0, 1)) + MaxPool2d(2, 1), models.py:74-84), mini_relu.cfg =
    conv_activation=ReLU (models.py:70-71).  Same bars as mini.cfg; plus the .weights round trip byte for byte."""
    z = load(tag + "_darknet.npz")
    cwd = os.getcwd()
    os.chdir(os.path.join(G, "mini"))
    try:
        net = yo.DarknetOracle(tag + ".cfg", anchors=yo.read_anchor_row("dataset/train.csv"))
        net.load_weights(tag + ".weights", [18, 18])
    finally:
        os.chdir(cwd)
    p = tmp_path / "rt.weights"
    net.save_weights(str(p))
    assert open(p, "rb").read() == open(os.path.join(G, "mini", tag + ".weights"), "rb").read()
    base = {k: v.clone() for k, v in net.params.items()}
    for k in net.trainable():
        net.params[k] = base[k].clone().requires_grad_(True)
    losses = net.forward(T(z["x"]), T(z["targets"]), bn_train=True)
    losses[0].sum().backward()
    np.testing.assert_allclose(torch.stack([l.detach() for l in losses]).numpy(), z["losses"], rtol=2e-5)
    for n, gn in zip([str(n) for n in z["grad_names"]], z["grad_norm"]):
        g = net.params[ref_key_to_oracle(n)].grad
        np.testing.assert_allclose(float(g.double().norm()), gn, rtol=2e-4, atol=1e-7, err_msg=n)
    for k in z.files:
        if k.startswith("grad::"):
            np.testing.assert_allclose(net.params[ref_key_to_oracle(k[6:])].grad.numpy(), z[k], rtol=1e-3, atol=2e-6, err_msg=k)
        if k.startswith("run::"):
            np.testing.assert_allclose(net.params[ref_key_to_oracle(k[5:])].detach().numpy(), z[k], rtol=1e-5, atol=1e-7, err_msg=k)
    for k, v in base.items():
        if "running" not in k:
            net.params[k] = v
    with torch.no_grad():
        ev = net.forward(T(z["x"]), None, bn_train=False)
    np.testing.assert_allclose(ev.numpy(), z["eval_out"], rtol=1e-4, atol=1e-5)


def test_mini_weights_roundtrip(mini, tmp_path):
    p = tmp_path / "rt.weights"
    mini.save_weights(str(p))
    assert open(p, "rb").read() == open(os.path.join(G, "mini", "mini.weights"), "rb").read()


def test_mini_dp_semantics(mini):
    """rank r's loss == oracle on shard r alone; all-reduced grad == sum of per-shard grads."""
    z = load("mini_darknet_dp.npz")
    base = {k: v.clone() for k, v in mini.params.items()}
    x, tg = T(z["x"]), T(z["targets"])
    for nsh in (2, 4):
        per = 8 // nsh
        tot = None
        for r in range(nsh):
            for k, v in base.items():
                mini.params[k] = v.clone().requires_grad_("running" not in k)
            ls = mini.forward(x[r * per:(r + 1) * per], tg[r * per:(r + 1) * per])
            ls[0].sum().backward()
            np.testing.assert_allclose(torch.stack([l.detach() for l in ls]).numpy(), z[f"losses_{nsh}"][r], rtol=5e-5)
            g0 = mini.params["conv0.weight"].grad
            tot = g0 if tot is None else tot + g0
        np.testing.assert_allclose(tot.numpy(), z[f"g0_{nsh}"], rtol=2e-3, atol=1e-5)
    for k, v in base.items():
        mini.params[k] = v


def test_yolo_baseline_structure():
    z = load("yolo_baseline_structure.npz")
    v = {(int(r[0]), int(r[1])): r for r in z["variants"]}
    assert v[(80, 416)][2] == 61949149 and v[(1, 416)][2] == 61523734
    assert v[(80, 416)][3] == 10647 and v[(80, 608)][3] == 22743 and v[(80, 416)][4] == 85
    assert z["conv_table"].shape[0] == 75


# ----------------------------------------------------------------------------- RektNet
def test_rektnet_forward_backward():
    z = load("rektnet_net.npz")
    sd0 = {k[4:]: T(z[k]) for k in z.files if k.startswith("sd::")}
    assert sum(v.numel() for k, v in sd0.items() if "running" not in k and "num_batches" not in k) == 311383
    x, thm, tpts = T(z["x"]), T(z["thm"]), T(z["tpts"])
    for lt, geo in (("l1_softargmax", True), ("l2_heatmap", False)):
        tag = f"{lt}:{int(geo)}"
        sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd0.items()}
        hm, pts = ro.keypoint_forward(x, sd, train=True)
        loc, gl, tot = ro.cross_ratio_loss(hm, pts, thm, tpts, lt, geo, 0.05, 0.05)
        tot.backward()
        np.testing.assert_allclose(pts.detach().numpy(), z[f"pts::{tag}"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose([float(loc), float(gl), float(tot)], z[f"loss::{tag}"], rtol=1e-5, atol=1e-7)
        names = [str(n) for n in z["gnames"]]
        for n, gn in zip(names, z[f"gnorm::{tag}"]):
            # conv biases in front of a BN have mathematically-zero grads (Q17): pure round-off, compare loosely
            noise = n.endswith(".bias") and "bn" not in n and n != "out.bias"
            np.testing.assert_allclose(float(sd[n].grad.double().norm()), gn, rtol=5e-3, atol=1e-3 if noise else 1e-6, err_msg=n)
        for k in z.files:
            if k.startswith(f"grad::{tag}::"):
                n = k.split("::")[2]
                ref = z[k]
                if n.endswith("conv1.bias"):
                    assert np.abs(sd[n].grad.numpy()).max() < 1e-3
                    continue
                np.testing.assert_allclose(sd[n].grad.numpy(), ref, rtol=5e-3, atol=1e-5 * max(1.0, float(np.abs(ref).max())), err_msg=k)
        if lt == "l1_softargmax":
            np.testing.assert_allclose(hm.detach().numpy(), z[f"hm::{tag}"], rtol=1e-3, atol=1e-8)
            for k in z.files:
                if k.startswith("run::"):
                    np.testing.assert_allclose(sd[k[5:]].numpy(), z[k], rtol=1e-5, atol=1e-7, err_msg=k)
    sd = {k: v.clone() for k, v in sd0.items()}
    with torch.no_grad():
        _, pts = ro.keypoint_forward(x, sd, train=False)
        lg = ro.keypoint_forward(x, sd, train=False, logits_only=True)
    np.testing.assert_allclose(pts.numpy(), z["eval_pts"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(lg[:1].numpy(), z["eval_logits"], rtol=1e-4, atol=1e-5)


def test_cross_ratio_loss_all_variants():
    z = load("cross_ratio.npz")
    hm, thm, tpts = T(z["hm"]), T(z["thm"]), T(z["tpts"])
    for lt in ("l2_softargmax", "l2_heatmap", "l1_softargmax"):
        for geo in (False, True):
            tag = f"{lt}:{int(geo)}"
            p = T(z["pts"]).clone().requires_grad_(True)
            h = hm.clone().requires_grad_(True)
            loc, gl, tot = ro.cross_ratio_loss(h, p, thm, tpts, lt, geo, 0.05, 0.07)
            tot.backward()
            np.testing.assert_allclose([float(loc), float(gl), float(tot)], z[f"loss::{tag}"], rtol=1e-6, atol=1e-8)
            dp = p.grad if p.grad is not None else torch.zeros_like(p)
            np.testing.assert_allclose(dp.numpy(), z[f"dpts::{tag}"], rtol=1e-5, atol=1e-8)
            if lt == "l2_heatmap":
                np.testing.assert_allclose(h.grad[0, 0].numpy(), z[f"dhm_sample::{tag}"], rtol=1e-5, atol=1e-10)
    with pytest.raises(NameError):
        ro.cross_ratio_loss(hm, T(z["pts"]), thm, tpts, "nonsense", True)


def test_oracle_rektnet_init_param_count():
    sd = ro.init_state(0)
    assert sum(v.numel() for k, v in sd.items() if "running" not in k) == 311383


# ---------------------------------------------------------------------------
# Detection post-processing — oracle/postprocess_oracle.py
from oracle import postprocess_oracle as PO


def test_post_nms_golden():
    g = np.load(os.path.join(G, "post_nms.npz"))
    for ci in range(int(g["n_cases"])):
        keep = PO.nms(g[f"boxes{ci}"], g[f"scores{ci}"], float(g[f"overlap{ci}"]), int(g[f"topk{ci}"]))
        assert keep.dtype == np.int64
        np.testing.assert_array_equal(keep, g[f"keep{ci}"], err_msg=f"case {ci}")


def test_post_average_precision_golden():
    g = np.load(os.path.join(G, "post_ap.npz"))
    for ci in range(int(g["n_cases"])):
        ap, r, p = PO.average_precision(g[f"tp{ci}"], g[f"conf{ci}"], int(g[f"ngt{ci}"]))
        ref = g[f"apr{ci}"]
        assert abs(float(ap) - float(ref[0])) <= 1e-6, ci
        assert float(r) == float(ref[1]) and (float(p) == float(ref[2]) or (np.isnan(p) and np.isnan(ref[2]))), ci
    assert abs(float(PO.compute_ap(g["ca_rec"], g["ca_pre"])) - float(g["ca_ap"])) <= 1e-6


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_post_validate_loop_golden(name):
    g = np.load(os.path.join(G, f"post_validate_{name}.npz"))
    out, tg = g["out"], g["targets"]
    args = (float(g["conf_thres"]), float(g["nms_thres"]), float(g["iou_thres"]), float(g["width"]), float(g["height"]))
    for b in range(out.shape[0]):
        r = PO.postprocess_image(out[b], tg[b], *args)
        assert r["count"] == int(g[f"count{b}"]), b
        assert bool(r["valid"]) == bool(g[f"valid{b}"]), b
        np.testing.assert_array_equal(r["boxes"], g[f"boxes{b}"])
        np.testing.assert_array_equal(r["prob"], g[f"prob{b}"])
        np.testing.assert_array_equal(r["cls"], g[f"cls{b}"])
        if r["valid"]:
            np.testing.assert_array_equal(r["correct"], g[f"correct{b}"])
            ref = g[f"apr{b}"]
            assert abs(float(r["ap"]) - float(ref[0])) <= 1e-6
            assert float(r["r"]) == float(ref[1]) and float(r["p"]) == float(ref[2])
    m = PO.validate_batches([out], [tg], *args)
    np.testing.assert_allclose(np.asarray(m[:3], np.float32), g["means"], atol=1e-6)


# ---------------------------------------------------------------------------
# detect -> crop -> keypoints glue — oracle/pipeline_oracle.py
from oracle import pipeline_oracle as PL


@pytest.mark.parametrize("h,w", [(1, 1), (2, 3), (17, 9), (80, 80), (160, 121), (300, 40)])
def test_resize_bilinear_matches_independent_implementation(h, w):
    """cv2 is absent: cross-check the restated INTER_LINEAR against torch's half-pixel bilinear (same convention)."""
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.random((3, h, w), dtype=np.float32)
    got = PL.resize_bilinear(img, 80, 80)
    ref = torch.nn.functional.interpolate(T(img)[None], size=(80, 80), mode="bilinear", align_corners=False)[0].numpy()
    np.testing.assert_allclose(got, ref, atol=1e-5, rtol=0)
    if (h, w) == (80, 80):
        np.testing.assert_array_equal(got, img)           # identity at scale 1


def test_crop_bounds_and_order():
    H, W = 60, 100
    assert PL.crop_bounds([10.2, 5.7, 20.1, 30.0], H, W) == (10, 5, 21, 30)
    assert PL.crop_bounds([-5.0, -3.0, 2.5, 1.5], H, W) == (0, 0, 3, 2)
    assert PL.crop_bounds([98.5, 58.2, 140.0, 90.0], H, W) == (98, 58, 100, 60)
    assert PL.crop_bounds([150.0, 70.0, 160.0, 80.0], H, W) == (99, 59, 100, 60)      # fully outside: edge pixel
    assert PL.crop_bounds([30.0, 20.0, 30.0, 20.0], H, W) == (30, 20, 31, 21)         # empty box: one pixel
    assert PL.crop_bounds([10.0, 10.0, 20.0, 20.0], H, W, (2.0, 0.5), (-4.0, 3.0)) == (16, 8, 36, 13)
    rng = np.random.default_rng(0)
    frames = rng.random((2, 3, H, W), dtype=np.float32)
    boxes = np.array([[[1, 1, 30, 40], [50, 10, 90, 50]], [[5, 5, 25, 25], [0, 0, 0, 0]]], np.float32)
    crops, owner = PL.crop_resize(frames, boxes, [2, 1], 16, 8)
    assert crops.shape == (3, 3, 16, 8) and owner.tolist() == [0, 0, 1]
    np.testing.assert_array_equal(crops[2], PL.resize_bilinear(frames[1, :, 5:25, 5:25], 16, 8))


@pytest.mark.parametrize("h,w", [(1, 1), (2, 3), (17, 9), (80, 80), (160, 121), (300, 40), (40, 80), (79, 81)])
def test_resize_bilinear_u8_fixed_point_rule(h, w):
    """The 8-bit rule of cv2.resize (what the reference's loaders run on the cv2.imread image): identity at scale 1, never more than
    one grey level from the exact bilinear value (11-bit coefficients + two truncating shifts + one rounding), exact on constant
    and on horizontally / vertically linear images' interiors up to that level, and monotone (no wrap-around at 0 / 255)."""
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, (3, h, w), dtype=np.uint8)
    got = PL.resize_bilinear_u8(img, 80, 80)
    assert got.dtype == np.uint8 and got.shape == (3, 80, 80)
    exact = torch.nn.functional.interpolate(T(img.astype(np.float64))[None], size=(80, 80), mode="bilinear", align_corners=False)[0].numpy()
    assert np.abs(got.astype(np.float64) - exact).max() <= 1.0 + 1e-9
    if (h, w) == (80, 80):
        np.testing.assert_array_equal(got, img)
    for v in (0, 1, 127, 254, 255):
        np.testing.assert_array_equal(PL.resize_bilinear_u8(np.full((1, h, w), v, np.uint8), 80, 80), np.full((1, 80, 80), v, np.uint8))
    # the divide by 255 happens AFTER the 8-bit rounding, in float64, rounded once to float32 (dataset.py:52)
    f = rng.random((1, 3, h, w), dtype=np.float32)
    crops, _ = PL.crop_resize(f, np.array([[[0, 0, w, h]]], np.float32), [1], 80, 80, u8=True)
    np.testing.assert_array_equal(crops[0], (PL.resize_bilinear_u8(PL.to_u8(f[0]), 80, 80).astype(np.float64) / 255.0).astype(np.float32))
    assert set(np.unique(np.rint(crops * 255) / 255 - crops).tolist()) <= {0.0} or np.abs(np.rint(crops * 255) / 255 - crops).max() < 1e-7


def test_resize_bilinear_u8_hand_computed():
    """2x upscale of a 1x2 image by hand: dst x = 0..3 -> fx = (x + .5) * .5 - .5 = -.25, .25, .75, 1.25 -> taps (0,0|w=0), (0,1|.25),
    (0,1|.75), (1,1|0); coefficients 2048/0, 1536/512, 512/1536, 2048/0; one source row, so both vertical taps read the same D."""
    img = np.array([[[10, 200]]], np.uint8)
    got = PL.resize_bilinear_u8(img, 1, 4)[0, 0].tolist()
    want = []
    for a0, a1 in ((2048, 0), (1536, 512), (512, 1536), (0, 2048)):       # (0, 2048) == the clamped tap (1, 1 | 2048, 0) on this image
        d = 10 * a0 + 200 * a1
        want.append((((2048 * (d >> 4)) >> 16) + ((0 * (d >> 4)) >> 16) + 2) >> 2)     # fy = (0 + .5) * 1 - .5 = 0 -> b = (2048, 0)
    assert got == want == [10, 58, 153, 200], (got, want)     # exact bilinear: 10, 57.5, 152.5, 200


# ---------------------------------------------------------------------------
# synthetic cone data — oracle/synth_oracle.py: contracts and the cv2 heat-map restatement
from oracle import synth_oracle as SO


def test_synth_oracle_contracts():
    t = SO.cone_targets(5, 2, 4, 9, 3)
    assert t.shape == (4, 9, 5) and t.dtype == np.float32
    for b in range(4):
        n = int((t[b, :, 3] > 0).sum())
        assert 1 <= n <= 9 and np.all(t[b, n:] == 0) and np.all(t[b, :n, 3:] > 0)
    img = SO.cone_images(5, 2, t, 48, 64)
    assert img.shape == (4, 3, 48, 64) and 0.0 <= img.min() and img.max() <= 1.0
    i2, hm, pts = SO.cone_crops(5, 2, 3)
    assert i2.shape == (3, 3, 80, 80) and hm.shape == (3, 7, 80, 80) and pts.shape == (3, 7, 2)
    np.testing.assert_allclose(hm.sum((2, 3)), 1.0, atol=1e-5)
    assert 0.0 <= pts.min() and pts.max() <= 1.0


def test_synth_oracle_heatmap_matches_dense_resize_and_blur():
    """The separable closed form == resizing the dense one-hot image (pipeline oracle's bilinear) and blurring it densely."""
    oh, ow, iy, ix = 37, 52, 20, 31
    dense = np.zeros((1, oh, ow), np.float32)
    dense[0, iy, ix] = 1.0
    r = PL.resize_bilinear(dense, 80, 80)[0].astype(np.float64)
    k = SO.GAUSS5
    pad = np.pad(r, 2, mode="reflect")                                  # numpy "reflect" == BORDER_REFLECT_101
    blur = sum(k[a] * k[b] * pad[a:a + 80, b:b + 80] for a in range(5) for b in range(5))
    vy = SO._blur_reflect101(SO._resize_onehot_axis(iy, oh, 80))
    vx = SO._blur_reflect101(SO._resize_onehot_axis(ix, ow, 80))
    np.testing.assert_allclose(np.outer(vy, vx), blur, atol=1e-7)


def test_autocast_bf16_cosine_curve_is_reference_output_and_the_oracle_agrees():
    """tests/golden/yolo_autocast_bf16_cos.json (the bar of the full-size bf16 GPU test) was produced by the REFERENCE's Darknet under
    torch.autocast; the same procedure on the oracle, stored beside it, agrees to 1.5e-3 in every conv layer."""
    import json
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "yolo_autocast_bf16_cos.json")))
    assert d["generator"] == "tests/golden/make_golden.py autocast" and d["batch"] == 32 and d["size"] == 416
    assert set(d["cos"]) == set(d["cos_oracle"]) and len(d["cos"]) == 75
    assert max(abs(d["cos"][k] - d["cos_oracle"][k]) for k in d["cos"]) <= 1.5e-3
    assert abs(d["loss"]["fp32"] - d["loss_oracle"]["fp32"]) <= 1e-4 * abs(d["loss"]["fp32"])
    assert d["max_rel_fp32_gradient_difference_reference_vs_oracle"] < 1e-3


def test_rektnet_autocast_fixture_is_reference_output_and_the_oracle_agrees():
    """tests/golden/rektnet_autocast_bf16_pts.json (make_golden.py rektnet_autocast): the key-point deviation of the REFERENCE KeypointNet under
    torch.autocast(cpu, bfloat16) at batch 256 -- the bar of the HIP bf16 test -- with the oracle's own curve beside it: identical in fp32, and the
    two bf16 curves within 1e-3 of each other at every quantile stored."""
    import json
    z = json.load(open(os.path.join(G, "rektnet_autocast_bf16_pts.json")))
    assert z["generator"].endswith("rektnet_autocast") and z["batch"] == 256 and z["init_seed"] == 5 and z["data_seed"] == 77
    assert z["fp32_max_abs_difference_reference_vs_oracle"] <= 1e-6
    for k in ("max", "p999", "p99", "mean"):
        assert abs(z["reference"][k] - z["oracle"][k]) <= 1e-3, (k, z["reference"][k], z["oracle"][k])
    assert 0.05 < z["reference"]["max"] < 0.07 and z["reference"]["mean"] < 0.01

