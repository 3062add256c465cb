#!/usr/bin/env python3
"""Generate the committed golden vectors by RUNNING THE REFERENCE ITSELF (CPU, fp32).

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_golden.py yolo
    python tests/golden/make_golden.py rektnet
    python tests/golden/make_golden.py post
    python tests/golden/make_golden.py autocast            (reference Darknet fp32 vs torch.autocast(cpu, bfloat16): per-layer gradient cosine)
    python tests/golden/make_golden.py rektnet_autocast    (reference KeypointNet fp32 vs autocast at batch 256: key-point deviation)

Two invocations because CVC-YOLOv3/utils is a package and RektNet/utils.py a module
(SURVEY.md §8c).  Writes tests/golden/*.npz (+ the mini cfg's .weights / train.csv).
The fixtures are DATA (inputs + expected outputs); no reference source is stored.
"""
import os
import sys
import warnings
import numpy as np
import torch

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
torch.set_num_threads(8)


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, {k: v.shape for k, v in out.items()})


def synth_targets(B, T, gen, min_real=1, cls_hi=1):
    """[B,T,5] cone-like boxes: n in [min_real,T] real rows, rest zero."""
    t = torch.zeros(B, T, 5)
    for b in range(B):
        n = int(torch.randint(min_real, T + 1, (1,), generator=gen))
        t[b, :n, 0] = torch.randint(0, cls_hi, (n,), generator=gen).float()
        t[b, :n, 1:3] = torch.rand(n, 2, generator=gen) * 0.9 + 0.05
        t[b, :n, 3:5] = torch.rand(n, 2, generator=gen) * 0.28 + 0.02
    return t


# ----------------------------------------------------------------------------
def gen_yolo():
    sys.path.insert(0, os.path.join(REF, "CVC-YOLOv3"))
    os.chdir(os.path.join(HERE, "mini"))                      # cfg's train_uri is relative to CWD
    from utils.utils import build_targets, bbox_iou           # reference
    from utils.parse_config import parse_model_config
    import models as ref_models

    # ---------------- build_targets known-answer tests ----------------
    def bt(name, target, anchors, C, Gh, Gw, thr=0.5):
        r = build_targets(target.clone(), anchors, anchors.shape[0], C, Gh, Gw, thr)
        keys = ("mask", "conf_mask", "tx", "ty", "tw", "th", "tconf", "tcls")
        npz(name, target=target, anchors=anchors, C=C, Gh=Gh, Gw=Gw, thr=thr,
            **{k: v for k, v in zip(keys, r)})

    van = torch.tensor(ref_models.vanilla_anchor_list, dtype=torch.float32)
    a13 = van[6:9] / 32.0
    # (i) cross-batch leak: image 0 has a box that exceeds thresh on 2 anchors, image 1 elsewhere
    t = torch.zeros(2, 3, 5)
    t[0, 0] = torch.tensor([0, 0.52, 0.31, 0.40, 0.33])
    t[0, 1] = torch.tensor([0, 0.11, 0.81, 0.25, 0.45])
    t[1, 0] = torch.tensor([0, 0.75, 0.75, 0.60, 0.55])
    bt("bt_leak.npz", t, a13, 1, 13, 13)
    # (ii) empty image (all rows zero) next to a normal one
    t = torch.zeros(2, 2, 5)
    t[1, 0] = torch.tensor([0, 0.40, 0.60, 0.30, 0.20])
    bt("bt_empty.npz", t, a13, 1, 13, 13)
    # (iii) collisions: two rows into the same cell/anchor — with and without padding rows
    t = torch.zeros(2, 4, 5)
    t[0, 0] = torch.tensor([0, 0.501, 0.502, 0.30, 0.31])
    t[0, 1] = torch.tensor([0, 0.509, 0.507, 0.31, 0.30])     # same cell, same best anchor; padding follows
    t[1, 0] = torch.tensor([0, 0.201, 0.202, 0.30, 0.31])
    t[1, 1] = torch.tensor([0, 0.209, 0.207, 0.31, 0.30])
    t[1, 2] = torch.tensor([0, 0.70, 0.10, 0.10, 0.12])
    t[1, 3] = torch.tensor([0, 0.205, 0.203, 0.29, 0.30])     # T == n_real: last row wins
    bt("bt_collide.npz", t, a13, 1, 13, 13)
    # (iv) tie in anchor IoU: two identical anchors -> argmax takes the first
    a_tie = torch.tensor([[2.0, 3.0], [2.0, 3.0], [5.0, 4.0]])
    t = torch.zeros(1, 2, 5)
    t[0, 0] = torch.tensor([0, 0.33, 0.66, 0.15, 0.22])
    t[0, 1] = torch.tensor([0, 0.80, 0.20, 0.40, 0.30])
    bt("bt_tie.npz", t, a_tie, 1, 13, 13)
    # (v) random B=8,T=16 at the six grid sizes, 80 classes incl. non-zero labels
    g = torch.Generator().manual_seed(1234)
    for G, sl in ((13, slice(6, 9)), (26, slice(3, 6)), (52, slice(0, 3))):
        bt(f"bt_rand_g{G}.npz", synth_targets(8, 16, g, cls_hi=80), van[sl] / (416.0 / G), 80, G, G)
    for G, sl in ((19, slice(6, 9)), (38, slice(3, 6)), (76, slice(0, 3))):
        bt(f"bt_rand_g{G}.npz", synth_targets(8, 16, g, cls_hi=3), van[sl] / (608.0 / G), 3, G, G)
    # non-square grid + different threshold
    bt("bt_rect.npz", synth_targets(4, 6, g), van[3:6] / 16.0, 2, 10, 14, thr=0.3)

    # bbox_iou both conventions
    b1 = torch.rand(64, 4, generator=g) * 50
    b2 = torch.rand(64, 4, generator=g) * 50
    c1 = torch.cat((torch.minimum(b1[:, :2], b1[:, 2:]), torch.maximum(b1[:, :2], b1[:, 2:])), 1)
    c2 = torch.cat((torch.minimum(b2[:, :2], b2[:, 2:]), torch.maximum(b2[:, :2], b2[:, 2:])), 1)
    npz("bbox_iou.npz", c1=c1, c2=c2, iou_corner=bbox_iou(c1, c2, True), b1=b1, b2=b2,
        iou_center=bbox_iou(b1, b2, False))

    # ---------------- YOLOLayer fwd/bwd ----------------
    def yl(name, B, C, G, anchors_px, cfg_h, T, seed):
        gg = torch.Generator().manual_seed(seed)
        layer = ref_models.YOLOLayer(anchors_px, C, cfg_h, cfg_h, 0.5, "leaky", 2.0, 1.6, 0.1, 25.0)
        sample = (torch.randn(B, len(anchors_px) * (5 + C), G, G, generator=gg) * 1.5).requires_grad_(True)
        targets = synth_targets(B, T, gg, cls_hi=max(1, min(C, 3)))
        loss, parts = layer(sample, targets)
        loss.backward()
        with torch.no_grad():
            ev = layer(sample.detach())
        npz(name, sample=sample, targets=targets, anchors_px=np.asarray(anchors_px, np.float32), C=C,
            cfg_h=cfg_h, loss=loss, parts=parts, dsample=sample.grad, eval_out=ev)

    yl("yolo_layer_c1_g13.npz", 4, 1, 13, ref_models.vanilla_anchor_list[6:9], 416, 6, 7)
    yl("yolo_layer_c80_g13.npz", 2, 80, 13, ref_models.vanilla_anchor_list[6:9], 416, 5, 8)
    yl("yolo_layer_c1_g26.npz", 3, 1, 26, ref_models.vanilla_anchor_list[3:6], 416, 8, 9)

    # ---------------- mini-cfg Darknet: whole-net loss / grads / optimizer steps ----------------
    torch.manual_seed(4242)
    net = ref_models.Darknet("mini.cfg", 2.0, 1.6, 25.0, 0.1, False)
    # non-trivial BN affine + running stats so load/save is exercised on all five tensors
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.75, 1.25)
                m.bias.uniform_(-0.1, 0.1)
                m.running_mean.uniform_(-0.05, 0.05)
                m.running_var.uniform_(0.8, 1.2)
    # the reference's save_weights needs header_info to be the numpy header that load_weights leaves
    # behind (models.py:344,402-403); train.py always loads first (train.py:191)
    net.header_info = np.zeros(5, np.int32)
    net.save_weights("mini.weights")
    net2 = ref_models.Darknet("mini.cfg", 2.0, 1.6, 25.0, 0.1, False)
    net2.load_weights("mini.weights", net2.get_start_weight_dim())
    for (k1, v1), (k2, v2) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1
    names = list(net.state_dict().keys())

    gg = torch.Generator().manual_seed(99)
    x = torch.rand(2, 3, 64, 64, generator=gg)
    targets = synth_targets(2, 4, gg)
    net.train()
    losses = net(x, targets)
    losses[0].sum().backward()
    grads = {n: p.grad.clone() for n, p in net.named_parameters()}
    running = {n: b.clone() for n, b in net.named_buffers() if "running" in n}
    pick = [n for n in grads if n.split(".")[1] in ("0", "3", "10", "11", "17", "18")]
    gnorm = np.array([float(grads[n].double().norm()) for n in grads])
    gsum = np.array([float(grads[n].double().sum()) for n in grads])
    out = dict(x=x, targets=targets, losses=torch.stack([l.detach() for l in losses]),
               grad_names=np.array(list(grads.keys())), grad_norm=gnorm, grad_sum=gsum,
               param_names=np.array(names))
    for n in pick:
        out["grad::" + n] = grads[n]
    for n, v in running.items():
        out["run::" + n] = v
    net.eval()
    with torch.no_grad():
        out["eval_out"] = net(x)
    # one optimiser step from the loaded weights, Adam and SGD (train.py:180-187)
    for opt_name in ("adam", "sgd"):
        m = ref_models.Darknet("mini.cfg", 2.0, 1.6, 25.0, 0.1, False)
        m.load_weights("mini.weights", m.get_start_weight_dim())
        m.train()
        opt = (torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=0.0) if opt_name == "adam"
               else torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, weight_decay=0.0))
        opt.zero_grad()
        ls = m(x, targets)
        ls[0].sum().backward()
        opt.step()
        sd = m.state_dict()
        for n in ("module_list.0.conv_0.weight", "module_list.10.conv_10.weight",
                  "module_list.11.conv_11.bias", "module_list.3.batch_norm_3.weight"):
            out[f"{opt_name}::" + n] = sd[n]
    npz("mini_darknet.npz", **out)

    # ---------------- the two cfg features mini.cfg does not reach: max-pools (both forms of yolo_baseline_tiny.cfg, models.py:74-84) and
    # conv_activation=ReLU (models.py:70-71).  Same recipe: seeded init, non-trivial BatchNorm state, .weights round trip, one train step
    # (losses, every gradient norm, first / middle / last conv gradients, running statistics) and the eval output.
    for tag, cfg_name, seed in (("mini_tiny", "mini_tiny.cfg", 777), ("mini_relu", "mini_relu.cfg", 778)):
        torch.manual_seed(seed)
        tn = ref_models.Darknet(cfg_name, 2.0, 1.6, 25.0, 0.1, False)
        with torch.no_grad():
            for m in tn.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.uniform_(0.75, 1.25)
                    m.bias.uniform_(-0.1, 0.1)
                    m.running_mean.uniform_(-0.05, 0.05)
                    m.running_var.uniform_(0.8, 1.2)
        tn.header_info = np.zeros(5, np.int32)
        tn.save_weights(tag + ".weights")
        tn2 = ref_models.Darknet(cfg_name, 2.0, 1.6, 25.0, 0.1, False)
        tn2.load_weights(tag + ".weights", tn2.get_start_weight_dim())
        for (k1, v1), (k2, v2) in zip(tn.state_dict().items(), tn2.state_dict().items()):
            assert k1 == k2 and torch.equal(v1, v2), k1
        tg_ = torch.Generator().manual_seed(seed + 1)
        tx = torch.rand(3, 3, 64, 64, generator=tg_)
        tt = synth_targets(3, 4, tg_)
        tn.train()
        tl = tn(tx, tt)
        tl[0].sum().backward()
        tgr = {n: p.grad.clone() for n, p in tn.named_parameters()}
        convs = [n for n in tgr if n.endswith("weight") and ".conv_" in n]
        tout = dict(x=tx, targets=tt, losses=torch.stack([l.detach() for l in tl]), param_names=np.array(list(tn.state_dict().keys())),
                    grad_names=np.array(list(tgr.keys())), grad_norm=np.array([float(tgr[n].double().norm()) for n in tgr]),
                    layer_kinds=np.array([type(m).__name__ for seq in tn.module_list for m in seq]))
        for n in (convs[0], convs[len(convs) // 2], convs[-1]):
            tout["grad::" + n] = tgr[n]
        for n, b in tn.named_buffers():
            if "running" in n:
                tout["run::" + n] = b.clone()
        tn.eval()
        with torch.no_grad():
            tout["eval_out"] = tn(tx)
        npz(tag + "_darknet.npz", **tout)

    # data-parallel semantics: B=8 as 2x4 and 4x2 shards (per-shard loss; sum-of-shard grads)
    gg = torch.Generator().manual_seed(7)
    xb = torch.rand(8, 3, 64, 64, generator=gg)
    tb = synth_targets(8, 4, gg)
    dp = dict(x=xb, targets=tb)
    for nsh in (1, 2, 4):
        per = 8 // nsh
        tot = None
        shard_losses = []
        for r in range(nsh):
            m = ref_models.Darknet("mini.cfg", 2.0, 1.6, 25.0, 0.1, False)
            m.load_weights("mini.weights", m.get_start_weight_dim())
            m.train()
            ls = m(xb[r * per:(r + 1) * per], tb[r * per:(r + 1) * per])
            ls[0].sum().backward()
            shard_losses.append(torch.stack([l.detach() for l in ls]))
            gl = [p.grad.clone() for p in m.parameters()]
            tot = gl if tot is None else [a + b for a, b in zip(tot, gl)]
        dp[f"losses_{nsh}"] = torch.stack(shard_losses)
        dp[f"gnorm_{nsh}"] = np.array([float(v.double().norm()) for v in tot])
        dp[f"g0_{nsh}"] = tot[0]
        dp[f"glast_{nsh}"] = tot[-2]
    npz("mini_darknet_dp.npz", **dp)

    # ---------------- yolo_baseline structure at real scale (no weights committed) ----------------
    import tempfile, shutil
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "dataset"))
    with open(os.path.join(tmp, "dataset", "train.csv"), "w") as f:
        f.write("10,13|16,30|33,23|30,61|62,45|59,119|116,90|156,198|373,326\n")
    rows = []
    for classes in (80, 1):
        for size in (416, 608):
            txt = open(os.path.join(REF, "CVC-YOLOv3/model_cfg/yolo_baseline.cfg")).read()
            txt = txt.replace("width=800", f"width={size}").replace("height=800", f"height={size}")
            txt = txt.replace("classes=80", f"classes={classes}")
            with open(os.path.join(tmp, "y.cfg"), "w") as f:
                f.write(txt)
            os.chdir(tmp)
            m = ref_models.Darknet("y.cfg", 2.0, 1.6, 25.0, 0.1, True)
            nparam = sum(p.numel() for p in m.parameters())
            m.eval()
            with torch.no_grad():
                o = m(torch.zeros(1, 3, size, size))
            rows.append((classes, size, nparam, o.shape[1], o.shape[2]))
            if classes == 80 and size == 416:
                table = []
                hooks = []
                xin = torch.zeros(1, 3, size, size)
                shapes = {}
                for i, mod in enumerate(m.module_list):
                    hooks.append(mod.register_forward_hook(lambda mm, inp, outp, i=i: shapes.__setitem__(i, tuple(outp.shape))))
                with torch.no_grad():
                    m(xin)
                for i, (d, mod) in enumerate(zip(m.module_defs, m.module_list)):
                    if d["type"] == "convolutional":
                        c = mod[0]
                        table.append((i, c.in_channels, c.out_channels, c.kernel_size[0], c.stride[0],
                                      shapes[i][2], int(c.bias is not None)))
                struct_table = np.array(table, np.int64)
    os.chdir(os.path.join(HERE, "mini"))
    shutil.rmtree(tmp)
    npz("yolo_baseline_structure.npz", variants=np.array(rows, np.int64), conv_table=struct_table)

    # tiny cfg structure too
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "dataset"))
    with open(os.path.join(tmp, "dataset", "train.csv"), "w") as f:
        f.write("10,13|16,30|33,23|30,61|62,45|59,119|116,90|156,198|373,326\n")
    txt = open(os.path.join(REF, "CVC-YOLOv3/model_cfg/yolo_baseline_tiny.cfg")).read()
    hdr = parse_model_config(os.path.join(REF, "CVC-YOLOv3/model_cfg/yolo_baseline_tiny.cfg"))[0]
    txt = txt.replace(f"width={hdr['width']}", "width=416").replace(f"height={hdr['height']}", "height=416")
    with open(os.path.join(tmp, "t.cfg"), "w") as f:
        f.write(txt)
    os.chdir(tmp)
    m = ref_models.Darknet("t.cfg", 2.0, 1.6, 25.0, 0.1, True)
    m.eval()
    with torch.no_grad():
        o = m(torch.zeros(1, 3, 416, 416))
    npz("yolo_tiny_structure.npz", nparam=sum(p.numel() for p in m.parameters()), out_shape=np.array(o.shape))
    os.chdir(HERE)
    shutil.rmtree(tmp)


# ----------------------------------------------------------------------------
def synth_keypoints(B, gen):
    """Heatmaps: 7 single pixels -> 5x5 gaussian -> normalised to sum 1 (mimics RektNet/utils.py:83-96);
    points in [0, 79/80]."""
    pts = torch.rand(B, 7, 2, generator=gen) * (79.0 / 80.0)
    hm = torch.zeros(B, 7, 80, 80)
    k1 = torch.tensor([1.0, 4.0, 6.0, 4.0, 1.0])
    k2 = (k1[:, None] * k1[None, :])
    for b in range(B):
        for k in range(7):
            cx = int(pts[b, k, 0] * 80)
            cy = int(pts[b, k, 1] * 80)
            for dy in range(-2, 3):
                for dx in range(-2, 3):
                    yy, xx = cy + dy, cx + dx
                    if 0 <= yy < 80 and 0 <= xx < 80:
                        hm[b, k, yy, xx] = k2[dy + 2, dx + 2]
            hm[b, k] /= hm[b, k].sum()
    return hm, pts


def gen_rektnet():
    sys.path.insert(0, os.path.join(REF, "RektNet"))
    from keypoint_net import KeypointNet                    # reference
    from cross_ratio_loss import CrossRatioLoss

    torch.manual_seed(17)
    net = KeypointNet(7, (80, 80))
    with torch.no_grad():                                   # make BN affine / conv bias non-trivial
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.8, 1.2)
                m.bias.uniform_(-0.1, 0.1)
            if isinstance(m, torch.nn.Conv2d):
                m.bias.uniform_(-0.05, 0.05)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    x = torch.rand(4, 3, 80, 80, generator=g)
    thm, tpts = synth_keypoints(4, g)
    out = dict(x=x, thm=thm, tpts=tpts)
    for k, v in sd0.items():
        out["sd::" + k] = v
    for lt, geo in (("l1_softargmax", True), ("l2_heatmap", False)):
        net.load_state_dict(sd0)
        net.train()
        net.zero_grad()
        crit = CrossRatioLoss(lt, geo, 0.05, 0.05)
        hm, pts = net(x)
        loc, gl, tot = crit(hm, pts, thm, tpts)
        tot.backward()
        tag = f"{lt}:{int(geo)}"
        out[f"hm::{tag}"] = hm if lt == "l1_softargmax" else hm[:1]
        out[f"pts::{tag}"] = pts
        out[f"loss::{tag}"] = torch.stack([loc.detach().float(), torch.as_tensor(gl).detach().float(), tot.detach().float()])
        names, gn, gs = [], [], []
        for n, p in net.named_parameters():
            names.append(n); gn.append(float(p.grad.double().norm())); gs.append(float(p.grad.double().sum()))
            if n in ("conv.weight", "bn.weight", "bn.bias", "res1.conv1.weight", "res1.conv1.bias", "res2.shortcut_conv.weight",
                     "res2.bn2.bias", "res3.conv2.weight", "res4.shortcut_bn.weight", "out.weight", "out.bias"):
                out[f"grad::{tag}::{n}"] = p.grad
        out[f"gnames"] = np.array(names)
        out[f"gnorm::{tag}"] = np.array(gn)
        out[f"gsum::{tag}"] = np.array(gs)
        if lt == "l1_softargmax":
            for n, b in net.named_buffers():
                if "running" in n:
                    out["run::" + n] = b.clone()
    # eval-mode forward + raw logits (onnx_mode)
    net.load_state_dict(sd0)
    net.eval()
    with torch.no_grad():
        hm_e, pts_e = net(x)
        net.onnx_mode = True
        logits = net(x)
        net.onnx_mode = False
    out["eval_pts"] = pts_e
    out["eval_logits"] = logits[:1]
    # one Adam(lr=0.1) step (train_eval.py:263)
    net.load_state_dict(sd0)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=0.1)
    opt.zero_grad()
    crit = CrossRatioLoss("l1_softargmax", True, 0.05, 0.05)
    hm, pts = net(x)
    crit(hm, pts, thm, tpts)[2].backward()
    opt.step()
    for n in ("conv.weight", "res2.conv2.bias", "out.weight", "res4.bn2.weight"):
        out["adam::" + n] = net.state_dict()[n]
    npz("rektnet_net.npz", **out)

    # ---------------- CrossRatioLoss alone: 3 loss types x geo on/off ----------------
    g = torch.Generator().manual_seed(21)
    hm = torch.softmax(torch.randn(8, 7, 6400, generator=g), -1).view(8, 7, 80, 80)
    thm, tpts = synth_keypoints(8, g)
    cr = dict(hm=hm, thm=thm, tpts=tpts)
    pts0 = torch.rand(8, 7, 2, generator=g)
    cr["pts"] = pts0
    for lt in ("l2_softargmax", "l2_heatmap", "l1_softargmax"):
        for geo in (False, True):
            p = pts0.clone().requires_grad_(True)
            h = hm.clone().requires_grad_(True)
            crit = CrossRatioLoss(lt, geo, 0.05, 0.07)
            loc, gl, tot = crit(h, p, thm, tpts)
            tot.backward()
            tag = f"{lt}:{int(geo)}"
            cr[f"loss::{tag}"] = torch.stack([loc.detach().float(), torch.as_tensor(gl).detach().float(), tot.detach().float()])
            cr[f"dpts::{tag}"] = p.grad if p.grad is not None else torch.zeros_like(p)
            if lt == "l2_heatmap":
                cr[f"dhm_sample::{tag}"] = h.grad[0, 0]
    npz("cross_ratio.npz", **cr)

    # ---------------- DP semantics for KeypointNet (B=8 as 1/2/4 shards) ----------------
    g = torch.Generator().manual_seed(31)
    xb = torch.rand(8, 3, 80, 80, generator=g)
    thm, tpts = synth_keypoints(8, g)
    dp = dict(x=xb, tpts=tpts, thm_argmax=thm.view(8, 7, -1).argmax(-1))
    for nsh in (1, 2, 4):
        per = 8 // nsh
        tot, sl = None, []
        for r in range(nsh):
            net.load_state_dict(sd0)
            net.train()
            net.zero_grad()
            crit = CrossRatioLoss("l1_softargmax", True, 0.05, 0.05)
            s = slice(r * per, (r + 1) * per)
            hm, pts = net(xb[s])
            loc, gl, t = crit(hm, pts, thm[s], tpts[s])
            t.backward()
            sl.append(torch.stack([loc.detach(), gl.detach(), t.detach()]))
            gl_ = [p.grad.clone() for p in net.parameters()]
            tot = gl_ if tot is None else [a + b for a, b in zip(tot, gl_)]
        dp[f"losses_{nsh}"] = torch.stack(sl)
        dp[f"gnorm_{nsh}"] = np.array([float(v.double().norm()) for v in tot])
        dp[f"g0_{nsh}"] = tot[0]
    npz("rektnet_dp.npz", **dp)


# ----------------------------------------------------------------------------
# Detection post-processing: utils/nms.py, utils/utils.py AP/IoU.
def dedup(x, gen):
    """Make all values distinct.  The reference's `scores.sort(0)` (nms.py:25) is an UNSTABLE sort on
    CPU for n > 16, so the visiting order of equal scores is implementation-defined there; vectors
    that pin the tie rule are kept at n <= 16 (where it degenerates to a stable insertion sort) and
    every larger vector has distinct scores."""
    x = x.clone()
    for _ in range(200):
        _, inv, cnt = torch.unique(x, return_inverse=True, return_counts=True)
        dup = cnt[inv] > 1
        if not bool(dup.any()):
            return x
        x[dup] = x[dup] + (torch.rand(int(dup.sum()), generator=gen) - 0.5) * 1e-4
    raise RuntimeError("dedup failed")


def synth_boxes(n, gen, span=416.0, clusters=None, quant=None, degenerate=0):
    """Corner boxes in clusters (so NMS has work to do) + scores; optional score
    quantisation (forces ties) and zero-area boxes."""
    k = clusters or max(1, n // 12)
    cen = torch.rand(k, 2, generator=gen) * span
    wh = torch.rand(k, 2, generator=gen) * 60 + 8
    which = torch.randint(0, k, (n,), generator=gen)
    c = cen[which] + torch.randn(n, 2, generator=gen) * 4
    s = wh[which] * (1 + 0.15 * torch.randn(n, 2, generator=gen)).clamp(0.3, 2.0)
    boxes = torch.cat([c - s / 2, c + s / 2], 1).float()
    scores = torch.rand(n, generator=gen)
    if quant:
        assert n <= 16, "tie vectors only where the reference's sort is stable"
        scores = torch.round(scores * quant) / quant
    else:
        scores = dedup(scores, gen)
    for d in range(min(degenerate, n)):
        boxes[d * 3 % n, 2:] = boxes[d * 3 % n, :2]       # zero area -> 0/0 IoU with itself only
    return boxes, scores


def synth_eval_output(B, N, C, T, gen, span, conf_lo=0.0, empty_labels=(), no_dets=()):
    """An eval-mode Darknet output [B,N,5+C] built around padded labels [B,T,5]:
    a few jittered detections per label with high confidence + clutter."""
    tg = synth_targets(B, T, gen, min_real=1, cls_hi=max(1, C))
    out = torch.zeros(B, N, 5 + C)
    for b in range(B):
        if b in empty_labels:
            tg[b] = 0
        real = tg[b][(tg[b, :, 1:5] > 0).all(1)]
        out[b, :, 0:2] = torch.rand(N, 2, generator=gen) * span
        out[b, :, 2:4] = torch.rand(N, 2, generator=gen) * 80 + 4
        out[b, :, 4] = torch.rand(N, generator=gen) * 0.6 + conf_lo
        out[b, :, 5:] = torch.rand(N, C, generator=gen)
        rows = torch.randperm(N, generator=gen)
        r = 0
        for lab in real:
            for _ in range(int(torch.randint(1, 6, (1,), generator=gen))):
                if r >= N:
                    break
                i = rows[r]; r += 1
                jit = 1 + 0.08 * torch.randn(4, generator=gen)
                out[b, i, 0:4] = lab[1:5] * span * jit
                out[b, i, 4] = 0.8 + 0.2 * torch.rand(1, generator=gen)
        out[b, :, 4] = dedup(out[b, :, 4], gen)
        if b in no_dets:
            out[b, :, 4] = 0.01
    return out.float(), tg.float()


def gen_post():
    sys.path.insert(0, os.path.join(REF, "CVC-YOLOv3"))
    from utils.nms import nms
    from utils.utils import average_precision, compute_ap, bbox_iou, xywh2xyxy
    gen = torch.Generator().manual_seed(4242)

    # --- nms: (n, overlap, top_k, quant, degenerate)
    cases = [(0, 0.5, 200, None, 0), (1, 0.5, 200, None, 0), (2, 0.25, 200, None, 0), (37, 0.5, 200, None, 0),
             (200, 0.25, 200, None, 0), (300, 0.5, 200, None, 3), (1000, 0.25, 200, None, 0), (1000, 0.5, 50, None, 0),
             (5000, 0.45, 200, None, 5), (10647, 0.25, 200, None, 0), (10647, 0.6, 512, None, 0), (64, 0.0, 200, None, 2),
             (450, 1.0, 200, None, 0), (16, 0.5, 200, 4, 0), (16, 0.3, 5, 2, 1), (12, 0.9, 200, 3, 0), (9, 0.5, 200, 1, 0),
             (16, 0.5, 200, 8, 0)]
    arrs = {"n_cases": len(cases)}
    for ci, (n, ov, tk, q, dg) in enumerate(cases):
        boxes, scores = synth_boxes(n, gen, quant=q, degenerate=dg) if n else (torch.zeros(0, 4), torch.zeros(0))
        if ci == 3:                       # exact duplicates
            boxes[5:10] = boxes[4]
        keep = nms(boxes, scores, ov, tk)
        arrs.update({f"boxes{ci}": boxes, f"scores{ci}": scores, f"overlap{ci}": np.float32(ov), f"topk{ci}": tk, f"keep{ci}": keep})
    npz("post_nms.npz", **arrs)

    # --- average_precision / compute_ap
    arrs = {}
    ms = [1, 2, 3, 7, 20, 57, 128, 200, 200, 200, 12, 16, 16]
    arrs["n_cases"] = len(ms)
    for ci, m in enumerate(ms):
        conf = torch.sort(torch.rand(m, generator=gen), descending=True)[0]
        if ci % 3 == 2 and m <= 16:       # ties: average_precision's sort(-conf) is unstable past 16 elements
            conf = torch.round(conf * 4) / 4
        else:
            conf = torch.sort(dedup(conf, gen), descending=True)[0]
        tp = (torch.rand(m, generator=gen) < (0.0 if ci == 1 else 1.0 if ci == 2 else 0.55)).to(torch.uint8)
        n_gt = max(1, int(tp.sum()) + int(torch.randint(0, 9, (1,), generator=gen)))
        ap, r, p = average_precision(tp=tp, conf=conf, n_gt=n_gt)
        arrs.update({f"tp{ci}": tp, f"conf{ci}": conf, f"ngt{ci}": n_gt, f"apr{ci}": torch.stack([ap, r, p])})
    rec = torch.sort(torch.rand(40, generator=gen))[0]
    pre = torch.rand(40, generator=gen)
    arrs.update(ca_rec=rec, ca_pre=pre, ca_ap=compute_ap(rec, pre))
    npz("post_ap.npz", **arrs)

    # --- the per-image loop validate.py:80-141, driven through the reference's own functions
    def per_image(det, labels, conf_t, nms_t, iou_t, W, H):
        det = det[det[:, 4] > conf_t]
        cls = torch.argmax(det[:, 5:], dim=1) if det.shape[0] else torch.zeros(0, dtype=torch.long)
        half = det[:, 2:4] / 2
        corner = torch.zeros(det.shape[0], 4)
        corner[:, 0:2] = det[:, 0:2] - half
        corner[:, 2:4] = det[:, 0:2] + half
        prob = det[:, 4]
        keep = nms(corner, prob, nms_t)
        res = dict(count=keep.shape[0], boxes=corner[keep], prob=prob[keep], cls=cls[keep], valid=False,
                   correct=torch.zeros(keep.shape[0], dtype=torch.uint8), apr=torch.zeros(3))
        if keep.shape[0] == 0:
            return res
        order = torch.sort(-res["prob"])[1]
        for k in ("boxes", "prob", "cls"):
            res[k] = res[k][order]
        labels = labels[(labels[:, 1:5] <= 0).sum(dim=1) == 0]
        if labels.shape[0] == 0:
            return res
        tb = xywh2xyxy(labels[:, 1:5])
        tb[:, (0, 2)] *= W
        tb[:, (1, 3)] *= H
        nd, nt = res["boxes"].shape[0], tb.shape[0]
        ious = bbox_iou(res["boxes"].unsqueeze(1).expand(-1, nt, -1), tb.unsqueeze(0).expand(nd, -1, -1))
        best = torch.argmax(ious, dim=1)
        taken = torch.zeros(nt, dtype=torch.uint8)
        for i in range(nd):
            if ious[i, best[i]] > iou_t and taken[best[i]] == 0:
                res["correct"][i] = 1
                taken[best[i]] = 1
        res["apr"] = torch.stack(average_precision(tp=res["correct"], conf=res["prob"], n_gt=labels.shape[0]))
        res["valid"] = True
        return res

    vcases = [  # name, B, N, C, T, span(W,H), conf, nms, iou, quant, empty_labels, no_dets
        ("a", 3, 10647, 1, 30, 416, 0.5, 0.25, 0.5, None, (), ()),
        ("b", 4, 507, 80, 12, 416, 0.3, 0.5, 0.5, 64, (1,), (2,)),
        ("c", 2, 3000, 1, 50, 608, 0.8, 0.25, 0.5, None, (), ()),
        ("d", 2, 2028, 3, 20, 416, 0.0, 0.4, 0.25, 32, (), ()),
    ]
    for (name, B, N, C, T, span, ct, nt_, it, q, el, nd_) in vcases:
        out, tg = synth_eval_output(B, N, C, T, gen, float(span), empty_labels=el, no_dets=nd_)
        arrs = dict(out=out, targets=tg, conf_thres=np.float32(ct), nms_thres=np.float32(nt_), iou_thres=np.float32(it),
                    width=span, height=span)
        aps, rs, ps = [], [], []
        for b in range(B):
            r = per_image(out[b], tg[b], ct, nt_, it, span, span)
            arrs.update({f"count{b}": r["count"], f"boxes{b}": r["boxes"], f"prob{b}": r["prob"], f"cls{b}": r["cls"],
                         f"correct{b}": r["correct"], f"apr{b}": r["apr"], f"valid{b}": r["valid"]})
            if r["valid"]:
                aps.append(r["apr"][0]); rs.append(r["apr"][1]); ps.append(r["apr"][2])
        arrs["means"] = torch.stack([torch.tensor(v, dtype=torch.float).mean() for v in (aps, rs, ps)])
        npz(f"post_validate_{name}.npz", **arrs)


# ----------------------------------------------------------------------------
def gen_autocast():
    """What bf16 costs the REFERENCE's own arithmetic: /root/reference/CVC-YOLOv3/models.Darknet (yolo_baseline topology at 416^2, classes 80)
    with the weights / batch / targets of tests/test_gpu_models.py::test_full_yolov3_batch32_train_forward_backward_vs_oracle, run once in
    fp32 and once under torch.autocast("cpu", bfloat16); cosine of every conv weight gradient between the two runs.  The HIP bf16 mode is
    held to this curve layer by layer.  The CPU oracle's curve (same procedure on oracle/yolo_oracle.py) is stored beside it as a cross-check.
    usage: make_golden.py autocast [batch=32]"""
    import json, tempfile, time
    sys.path.insert(0, os.path.join(REF, "CVC-YOLOv3"))
    ROOT = os.path.dirname(os.path.dirname(HERE))
    sys.path.insert(0, ROOT)
    import bench
    from oracle import yolo_oracle as yo
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    S = 416
    tmp = tempfile.mkdtemp()
    cfg = bench.write_yolo_cfg(tmp)
    os.chdir(tmp)                                                # the cfg's train_uri is relative to the CWD (models.py:29-40)
    import models as ref_models
    orc = yo.DarknetOracle(cfg, anchors=yo.VANILLA_ANCHORS, seed=3)
    wpath = os.path.join(tmp, "seed3.weights")
    orc.save_weights(wpath)
    ref = ref_models.Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True)
    ref.load_weights(wpath, [255, 255, 255])                     # the reference's own loader (models.py:339-397)
    ref.train()
    g = torch.Generator().manual_seed(21)
    x = torch.rand(B, 3, S, S, generator=g)
    tg = bench.synth_targets(B, 16, g)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    convs = [(i, m[0]) for i, (d, m) in enumerate(zip(ref.module_defs, ref.module_list)) if d["type"] == "convolutional"]
    bns = [m[1] for d, m in zip(ref.module_defs, ref.module_list) if d["type"] == "convolutional" and len(m) > 1 and isinstance(m[1], torch.nn.BatchNorm2d)]

    def run_ref(mode):
        ref.zero_grad()
        snap = [(b.running_mean.clone(), b.running_var.clone(), b.num_batches_tracked.clone()) for b in bns]
        t0 = time.time()
        if mode == "bf16":
            with torch.autocast("cpu", dtype=torch.bfloat16):
                out = ref(x, tg)
        else:
            out = ref(x, tg)
        out[0].sum().backward()
        with torch.no_grad():
            for b, (rm, rv, nb) in zip(bns, snap):
                b.running_mean.copy_(rm); b.running_var.copy_(rv); b.num_batches_tracked.copy_(nb)
        print("reference", mode, "loss %.4f (%.0f s)" % (float(out[0].sum()), time.time() - t0), flush=True)
        return float(out[0].sum()), {"conv%d.weight" % i: c.weight.grad.detach().double().reshape(-1).clone() for i, c in convs}

    def run_orc(mode):
        for k in orc.trainable():
            orc.params[k].requires_grad_(True); orc.params[k].grad = None
        snap = {k: v.clone() for k, v in orc.params.items() if "running" in k}
        t0 = time.time()
        if mode == "bf16":
            with torch.autocast("cpu", dtype=torch.bfloat16):
                out = orc.forward(x, tg)
        else:
            out = orc.forward(x, tg)
        out[0].sum().backward()
        with torch.no_grad():
            for k, v in snap.items():
                orc.params[k].copy_(v)
        print("oracle   ", mode, "loss %.4f (%.0f s)" % (float(out[0].sum()), time.time() - t0), flush=True)
        return float(out[0].sum()), {k: orc.params[k].grad.detach().double().reshape(-1).clone() for k in orc.trainable()
                                     if k.endswith("weight") and k.startswith("conv")}

    def cosines(a, b):
        return {k: round(float(a[k] @ b[k] / (a[k].norm() * b[k].norm() + 1e-30)), 4) for k in a}
    lr32, gr32 = run_ref("fp32")
    lr16, gr16 = run_ref("bf16")
    lo32, go32 = run_orc("fp32")
    lo16, go16 = run_orc("bf16")
    cos_ref, cos_orc = cosines(gr32, gr16), cosines(go32, go16)
    assert set(cos_ref) == set(cos_orc), (sorted(cos_ref)[:5], sorted(cos_orc)[:5])
    dmax = max(abs(cos_ref[k] - cos_orc[k]) for k in cos_ref)
    gmax = max(float((gr32[k] - go32[k]).norm() / (go32[k].norm() + 1e-30)) for k in gr32)
    print("reference vs oracle: max |cos difference| %.4f, max relative fp32 gradient difference %.2e" % (dmax, gmax))
    out = {"what": "cosine(conv weight gradient under torch.autocast(cpu, bfloat16), same in fp32) of the REFERENCE's models.Darknet "
                   "(/root/reference/CVC-YOLOv3/models.py) on the yolo_baseline topology; cos_oracle = the same for oracle/yolo_oracle.py",
           "generator": "tests/golden/make_golden.py autocast", "batch": B, "size": S, "oracle_seed": 3, "data_seed": 21,
           "targets_per_image": 16, "torch": torch.__version__, "loss": {"fp32": lr32, "bf16": lr16},
           "loss_oracle": {"fp32": lo32, "bf16": lo16}, "cos": cos_ref, "cos_oracle": cos_orc,
           "max_abs_cos_difference_reference_vs_oracle": round(dmax, 4), "max_rel_fp32_gradient_difference_reference_vs_oracle": gmax}
    with open(os.path.join(HERE, "yolo_autocast_bf16_cos.json" if B == 32 else "yolo_autocast_bf16_cos_b%d.json" % B), "w") as f:
        json.dump(out, f, indent=1)


def gen_rektnet_autocast():
    """What bf16 does to RektNet's key points ON THE REFERENCE'S OWN ARITHMETIC: the reference KeypointNet (RektNet/keypoint_net.py:58-70) on
    BASELINE config 2's batch (256 crops of 80x80; the seed-5 init and seed-77 batch of tests/test_gpu_models.py::test_keypointnet_batch256...),
    train mode, once in fp32 and once under torch.autocast("cpu", bfloat16); stored: the distribution of |key point(bf16) - key point(fp32)|
    over the 256 x 7 x 2 coordinates.  The HIP bf16 mode is held to this curve (+ a stated margin), not to a hand-picked bound."""
    import json
    sys.path.insert(0, os.path.join(REF, "RektNet"))
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from keypoint_net import KeypointNet                    # reference
    from oracle import rektnet_oracle as ro
    B = 256
    sd = ro.init_state(5)
    g = torch.Generator().manual_seed(77)
    x = torch.rand(B, 3, 80, 80, generator=g)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))

    def run(mode, impl):
        if impl == "reference":
            net = KeypointNet(7, (80, 80))
            full = net.state_dict()
            full.update({k: v.detach().clone() for k, v in ro.init_state(5).items()})
            net.load_state_dict(full)
            net.train()
            f = lambda: net(x)                                                     # noqa: E731
        else:
            st = ro.init_state(5)
            f = lambda: ro.keypoint_forward(x, st, train=True)                      # noqa: E731
        with torch.no_grad():
            if mode == "bf16":
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    out = f()
            else:
                out = f()
        return out[1].float()
    res = {}
    for impl in ("reference", "oracle"):
        p32, p16 = run("fp32", impl), run("bf16", impl)
        d = (p16 - p32).abs().numpy().reshape(-1)
        res[impl] = {"max": float(d.max()), "p999": float(np.quantile(d, 0.999)), "p99": float(np.quantile(d, 0.99)), "mean": float(d.mean())}
        if impl == "reference":
            ref32 = p32
        else:
            res["fp32_max_abs_difference_reference_vs_oracle"] = float((p32 - ref32).abs().max())
        print(impl, res[impl], flush=True)
    out = {"what": "|key point under torch.autocast(cpu, bfloat16) - key point in fp32| of the REFERENCE's KeypointNet (train mode, batch 256, 3584 "
                   "coordinates in [0, 1]); 'oracle' = the same for oracle/rektnet_oracle.py", "generator": "tests/golden/make_golden.py rektnet_autocast",
           "batch": B, "init_seed": 5, "data_seed": 77, "torch": torch.__version__, **res}
    with open(os.path.join(HERE, "rektnet_autocast_bf16_pts.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    which = sys.argv[1]
    {"yolo": gen_yolo, "rektnet": gen_rektnet, "post": gen_post, "autocast": gen_autocast, "rektnet_autocast": gen_rektnet_autocast}[which]()
