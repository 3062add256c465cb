"""CPU oracle for the RektNet hot path (TEST INFRASTRUCTURE, not product).

  keypoint_forward     <- RektNet/keypoint_net.py:58-70 (+ flat_softmax :46-49, soft_argmax :51-56)
  res_block            <- RektNet/resnet.py:22-27
  cross_ratio_loss     <- RektNet/cross_ratio_loss.py:20-63
  init_state           <- RektNet/keypoint_net.py:33-44 (kaiming-normal fan_out, zero bias, BN 1/0)

State-dict key names follow the reference modules (conv, bn, res{1..4}.{conv1,bn1,conv2,bn2,
shortcut_conv,shortcut_bn}, out) so a reference checkpoint can be fed in directly.
Pinned by tests/golden/rektnet_*.npz (generated from the reference).
"""
import math
import torch
import torch.nn.functional as F

WIDTHS = [(16, 16), (16, 32), (32, 64), (64, 128)]     # keypoint_net.py:21-24


def conv_specs(num_kpt=7):
    """(name, cin, cout, k, pad, dil) for every conv, in forward order."""
    specs = [("conv", 3, 16, 7, 3, 1)]
    for r, (ci, co) in enumerate(WIDTHS, start=1):
        specs += [(f"res{r}.conv1", ci, co, 3, 2, 2), (f"res{r}.conv2", co, co, 3, 1, 1),
                  (f"res{r}.shortcut_conv", ci, co, 1, 0, 1)]
    specs.append(("out", 128, num_kpt, 1, 0, 1))
    return specs


def bn_names():
    names = ["bn"]
    for r in range(1, 5):
        names += [f"res{r}.bn1", f"res{r}.bn2", f"res{r}.shortcut_bn"]
    return names


def init_state(seed, num_kpt=7):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, ci, co, k, _, _ in conv_specs(num_kpt):
        std = math.sqrt(2.0 / (co * k * k))              # kaiming_normal_, mode=fan_out, relu
        sd[f"{name}.weight"] = torch.randn(co, ci, k, k, generator=g) * std
        sd[f"{name}.bias"] = torch.zeros(co)
    for name in bn_names():
        head, _, leaf = name.rpartition(".")
        conv_leaf = {"bn": "conv", "bn1": "conv1", "bn2": "conv2", "shortcut_bn": "shortcut_conv"}[leaf]
        c = sd[(head + "." if head else "") + conv_leaf + ".weight"].shape[0]
        sd[f"{name}.weight"] = torch.ones(c)
        sd[f"{name}.bias"] = torch.zeros(c)
        sd[f"{name}.running_mean"] = torch.zeros(c)
        sd[f"{name}.running_var"] = torch.ones(c)
    return sd


def _bn(x, sd, name, train):
    return F.batch_norm(x, sd[f"{name}.running_mean"], sd[f"{name}.running_var"],
                        sd[f"{name}.weight"], sd[f"{name}.bias"], training=train, momentum=0.1, eps=1e-5)


def res_block(x, sd, p, train):
    a = F.relu(_bn(F.conv2d(x, sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], padding=2, dilation=2), sd, f"{p}.bn1", train))
    main = _bn(F.conv2d(a, sd[f"{p}.conv2.weight"], sd[f"{p}.conv2.bias"], padding=1), sd, f"{p}.bn2", train)
    side = _bn(F.conv2d(x, sd[f"{p}.shortcut_conv.weight"], sd[f"{p}.shortcut_conv.bias"]), sd, f"{p}.shortcut_bn", train)
    return F.relu(side + main)


def keypoint_forward(x, sd, train=True, num_kpt=7, logits_only=False):
    H, W = x.shape[2], x.shape[3]
    a = F.relu(_bn(F.conv2d(x, sd["conv.weight"], sd["conv.bias"], padding=3), sd, "bn", train))
    for r in range(1, 5):
        a = res_block(a, sd, f"res{r}", train)
    z = F.conv2d(a, sd["out.weight"], sd["out.bias"])      # head conv (reference runs it twice; same value)
    if logits_only:
        return z
    hm = torch.softmax(z.reshape(-1, H * W), 1).view(-1, num_kpt, H, W)
    vy = torch.linspace(0, (H - 1.0) / H, H, dtype=x.dtype)
    vx = torch.linspace(0, (W - 1.0) / W, W, dtype=x.dtype)
    ey = (hm.sum(3) * vy).sum(-1)
    ex = (hm.sum(2) * vx).sum(-1)
    return hm, torch.stack([ex, ey], -1).view(-1, num_kpt, 2)


_GEO_TERMS = [  # (U = a-b, V = c-d, which gamma)   cross_ratio_loss.py:36-55
    ((3, 1), (5, 3), "v"), ((1, 0), (3, 1), "v"), ((6, 4), (4, 2), "v"), ((4, 2), (2, 0), "v"),
    ((4, 3), (2, 1), "h"), ((6, 5), (4, 3), "h"),
]


def cross_ratio_loss(hm, pts, thm, tpts, loss_type="l1_softargmax", include_geo=True,
                     gamma_horz=0.05, gamma_vert=0.05):
    if loss_type in ("l2_softargmax", "l2_sm"):
        loc = ((pts - tpts) ** 2).sum(2).sum(1).mean()
    elif loss_type in ("l2_heatmap", "l2_hm"):
        loc = ((hm - thm) ** 2).sum(3).sum(2).sum(1).mean()
    elif loss_type in ("l1_softargmax", "l1_sm"):
        loc = (pts - tpts).abs().sum(2).sum(1).mean()
    else:
        raise NameError("sys")      # the reference hits an un-imported `sys` here (cross_ratio_loss.py:32)
    if not include_geo:
        return loc, torch.tensor(0), loc + torch.tensor(0)

    def unit(a, b):
        return F.normalize(pts[:, a] - pts[:, b], dim=1)
    sums = {"h": 0.0, "v": 0.0}
    for (ua, ub), (va, vb), kind in _GEO_TERMS:
        # tensordot over the coordinate axis -> [B,B] all-pairs matrix (not a per-sample dot)
        sums[kind] = sums[kind] + (1.0 - unit(ua, ub) @ unit(va, vb).t())
    geo = gamma_horz * sums["h"].mean() / 2 + gamma_vert * sums["v"].mean() / 4
    return loc, geo, loc + geo
