"""CPU oracle for the CVC-YOLOv3 hot path (TEST INFRASTRUCTURE, not product).

Restates, in plain torch-CPU fp32 + explicit Python loops, what the reference
computes on the detector training path.  Citations are to /root/reference.

  parse_cfg              <- CVC-YOLOv3/utils/parse_config.py:1-18
  corner_iou_plus1       <- CVC-YOLOv3/utils/utils.py:163-193 (x1y1x2y2=True branch)
  build_targets          <- CVC-YOLOv3/utils/utils.py:195-275
  yolo_layer             <- CVC-YOLOv3/models.py:140-220
  DarknetOracle          <- CVC-YOLOv3/models.py:15-110 (topology rules),
                            :312-338 (forward), :339-422 (.weights I/O)

Pinned by tests/golden/*.npz (generated from the reference itself by
tests/golden/make_golden.py) in tests/test_oracle_golden.py.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

VANILLA_ANCHORS = [[10, 13], [16, 30], [33, 23], [30, 61], [62, 45],
                   [59, 119], [116, 90], [156, 198], [373, 326]]  # models.py:13


# --------------------------------------------------------------------------
# cfg text -> list of dict   (parse_config.py:1-18)
# --------------------------------------------------------------------------
def parse_cfg(path):
    blocks = []
    with open(path, "r") as fh:
        for raw in fh.read().split("\n"):
            if not raw or raw.startswith("#"):      # filter is applied BEFORE stripping
                continue
            line = raw.strip()
            if line.startswith("["):
                blk = {"type": line[1:-1].rstrip()}
                if blk["type"] == "convolutional":
                    blk["batch_normalize"] = 0
                blocks.append(blk)
            else:
                key, val = line.split("=")
                blocks[-1][key.rstrip()] = val.strip()
    return blocks


def read_anchor_row(csv_path):
    """Row 0 of train.csv is 'w,h|w,h|...' (models.py:29-35)."""
    import csv
    with open(csv_path) as fh:
        row = next(csv.reader(fh))
    text = str(row)[2:-2]
    return [[float(v) for v in pair.split(",")] for pair in text.split("'")[0].split("|")]


# --------------------------------------------------------------------------
# IoU with the "+1 pixel" convention  (utils.py:178-191)
# --------------------------------------------------------------------------
def corner_iou_plus1(b1, b2):
    """b1, b2: [...,4] corner boxes, float32 tensors (broadcastable)."""
    ix1 = torch.max(b1[..., 0], b2[..., 0])
    iy1 = torch.max(b1[..., 1], b2[..., 1])
    ix2 = torch.min(b1[..., 2], b2[..., 2])
    iy2 = torch.min(b1[..., 3], b2[..., 3])
    inter = torch.clamp(ix2 - ix1 + 1, min=0) * torch.clamp(iy2 - iy1 + 1, min=0)
    a1 = (b1[..., 2] - b1[..., 0] + 1) * (b1[..., 3] - b1[..., 1] + 1)
    a2 = (b2[..., 2] - b2[..., 0] + 1) * (b2[..., 3] - b2[..., 1] + 1)
    return inter / (a1 + a2 - inter + 1e-12)


def center_iou_plus1(b1, b2):
    """x1y1x2y2=False branch (utils.py:167-172): cx,cy,w,h -> corners first."""
    def corners(b):
        return torch.stack((b[..., 0] - b[..., 2] / 2, b[..., 1] - b[..., 3] / 2,
                            b[..., 0] + b[..., 2] / 2, b[..., 1] + b[..., 3] / 2), -1)
    return corner_iou_plus1(corners(b1), corners(b2))


# --------------------------------------------------------------------------
# build_targets  (utils.py:195-275) — sequential restatement
# --------------------------------------------------------------------------
def build_targets(target, anchors, num_anchors, num_classes, grid_h, grid_w, ignore_thres):
    """target [B,T,5] (cls,cx,cy,w,h normalised, zero rows = padding), anchors [A,2]
    in grid units.  Returns mask, conf_mask (uint8 [B,A,Gh,Gw]), tx,ty,tw,th,tconf
    (float32 same shape), tcls (uint8 [B,A,Gh,Gw,C]).  Input is not modified."""
    target = target.detach().to(torch.float32).cpu()
    anchors = anchors.detach().to(torch.float32).cpu()
    B, T = target.shape[0], target.shape[1]
    A, C = num_anchors, num_classes

    real = target.sum(dim=2) > 0                      # :210 master_mask
    gx = target[:, :, 1] * grid_w                     # :213-216 (fresh tensors)
    gy = target[:, :, 2] * grid_h
    gw = target[:, :, 3] * grid_w
    gh = target[:, :, 4] * grid_h
    gi = gx.long()                                    # :219-220 truncation
    gj = gy.long()
    # :223-228 padded rows take the values of row 0 of the same image
    for q in (gi, gj, gx, gy, gw, gh):
        row0 = q[:, 0:1].expand(B, T)
        q[~real] = row0[~real]

    # :231-240 anchor IoU of (0,0,gw,gh) vs (0,0,aw,ah) -> [B,A,T]
    zero = torch.zeros(B, T, dtype=torch.float32)
    gt = torch.stack((zero, zero, gw, gh), -1)[:, None, :, :]                       # [B,1,T,4]
    an = torch.cat((torch.zeros(A, 2), anchors), 1)[None, :, None, :]               # [1,A,1,4]
    iou = corner_iou_plus1(gt, an)                                                   # [B,A,T]

    mask = np.zeros((B, A, grid_h, grid_w), np.uint8)
    conf_mask = np.ones((B, A, grid_h, grid_w), np.uint8)
    tx = np.zeros((B, A, grid_h, grid_w), np.float32)
    ty = np.zeros_like(tx); tw = np.zeros_like(tx); th = np.zeros_like(tx); tconf = np.zeros_like(tx)
    tcls = np.zeros((B, A, grid_h, grid_w, C), np.uint8)

    # :244-255 — every (b,a,t) with IoU>thresh zeroes cell (gj,gi) in ALL images and ALL anchors
    over = (iou > ignore_thres).numpy()
    gi_n, gj_n = gi.numpy(), gj.numpy()
    for b in range(B):
        for a in range(A):
            for t in range(T):
                if over[b, a, t]:
                    conf_mask[:, :, gj_n[b, t], gi_n[b, t]] = 0

    best = torch.argmax(iou, dim=1).numpy()           # :257 first max on ties
    # values written at the assigned cell (computed vectorised in fp32, like the reference)
    txv = (gx - gi.float()).numpy()
    tyv = (gy - gj.float()).numpy()
    best_t = torch.from_numpy(best)
    twv = torch.log(gw / anchors[best_t, 0] + 1e-16).numpy()
    thv = torch.log(gh / anchors[best_t, 1] + 1e-16).numpy()
    label = target[:, :, 0].long().numpy()            # :271 (label of the ROW, also for padded rows)

    # :262-273 — sequential scatter, later t overwrites earlier t on collisions
    for b in range(B):
        for t in range(T):
            a, j, i = best[b, t], gj_n[b, t], gi_n[b, t]
            mask[b, a, j, i] = 1
            conf_mask[b, a, j, i] = 1
            tx[b, a, j, i] = txv[b, t]
            ty[b, a, j, i] = tyv[b, t]
            tw[b, a, j, i] = twv[b, t]
            th[b, a, j, i] = thv[b, t]
            tcls[b, a, j, i, label[b, t]] = 1
            tconf[b, a, j, i] = 1
    f = torch.from_numpy
    return f(mask), f(conf_mask), f(tx), f(ty), f(tw), f(th), f(tconf), f(tcls)


# --------------------------------------------------------------------------
# YOLO layer  (models.py:140-220)
# --------------------------------------------------------------------------
def yolo_layer(sample, anchors_px, num_classes, cfg_height, targets=None,
               ignore_thres=0.5, xy_loss=2.0, wh_loss=1.6, object_loss=0.1, no_object_loss=25.0):
    """sample [B, A*(5+C), Gh, Gw].  Train -> (loss, parts[6] = x,y,w,h,obj,noobj);
    eval -> [B, A*Gh*Gw, 5+C] (xywh in input pixels)."""
    A = len(anchors_px)
    B, _, Gh, Gw = sample.shape
    attrs = 5 + num_classes
    stride = cfg_height / Gh                                      # :145 cfg height, both axes
    p = sample.view(B, A, attrs, Gh, Gw).permute(0, 1, 3, 4, 2)   # :147
    sx, sy = torch.sigmoid(p[..., 0]), torch.sigmoid(p[..., 1])
    rw, rh = p[..., 2], p[..., 3]
    conf = torch.sigmoid(p[..., 4])
    cls = torch.sigmoid(p[..., 5:])
    sa = torch.tensor([(aw / stride, ah / stride) for aw, ah in anchors_px], dtype=torch.float32)

    if targets is None:
        col = torch.arange(Gw, dtype=torch.float32).view(1, 1, 1, Gw)
        row = torch.arange(Gh, dtype=torch.float32).view(1, 1, Gh, 1)
        box = torch.stack((sx.detach() + col, sy.detach() + row,
                           torch.exp(rw.detach()) * sa[:, 0].view(1, A, 1, 1),
                           torch.exp(rh.detach()) * sa[:, 1].view(1, A, 1, 1)), -1)
        return torch.cat((box.reshape(B, -1, 4) * stride, conf.reshape(B, -1, 1),
                          cls.reshape(B, -1, num_classes)), -1)

    m, cm, tx, ty, tw, th, tconf, _ = build_targets(targets, sa, A, num_classes, Gh, Gw, ignore_thres)
    pos = m.bool()
    neg = (cm - m).bool()                                          # :196 conf_mask - mask

    def mse(a, b):
        return ((a - b) ** 2).mean()

    def bce(pr, tg):                                               # torch BCELoss: log clamped at -100
        return -(tg * torch.clamp(torch.log(pr), min=-100.0)
                 + (1 - tg) * torch.clamp(torch.log(1 - pr), min=-100.0)).mean()

    lx = xy_loss * mse(sx[pos], tx[pos])
    ly = xy_loss * mse(sy[pos], ty[pos])
    lw = wh_loss * mse(rw[pos], tw[pos])
    lh = wh_loss * mse(rh[pos], th[pos])
    lno = no_object_loss * bce(conf[neg], tconf[neg])
    lob = object_loss * bce(conf[pos], tconf[pos])
    # class term is multiplied by 0 in the reference (:204-205) -> contributes nothing
    loss = lx + ly + lw + lh + lno + lob
    parts = torch.stack([v.detach() for v in (lx, ly, lw, lh, lob, lno)])   # :211 order
    return loss, parts


# --------------------------------------------------------------------------
# Whole network as a functional model over a flat dict of tensors
# --------------------------------------------------------------------------
class DarknetOracle:
    """Functional restatement of models.Darknet for CPU checking.

    params: dict  'conv{i}.weight' [O,I,k,k], 'conv{i}.bias' (preyolo only),
                  'bn{i}.weight/bias/running_mean/running_var'
    """

    def __init__(self, cfg_path, anchors=None, xy_loss=2.0, wh_loss=1.6,
                 no_object_loss=25.0, object_loss=0.1, seed=None):
        blocks = parse_cfg(cfg_path)
        self.hyper = blocks[0]
        self.defs = blocks[1:]
        h = self.hyper
        self.classes = int(h["classes"])
        self.cfg_h, self.cfg_w = int(h["height"]), int(h["width"])
        self.slope = float(h["leaky_slope"])
        self.act = h["conv_activation"]
        self.ignore = float(h["build_targets_ignore_thresh"])
        self.masks = [[int(v) for v in grp.split(",")] for grp in h["yolo_masks"].split("|")]
        self.anchors = anchors if anchors is not None else VANILLA_ANCHORS
        self.consts = dict(xy_loss=xy_loss, wh_loss=wh_loss, object_loss=object_loss,
                           no_object_loss=no_object_loss)
        # topology bookkeeping (models.py:20,47-109)
        chans = [int(h["channels"])]
        self.layers = []
        ycount = 0
        for i, d in enumerate(self.defs):
            t = d["type"]
            info = {"type": t}
            if t == "convolutional":
                pre = d["filters"] == "preyolo"
                cout = (self.classes + 5) * len(self.masks[ycount]) if pre else int(d["filters"])
                k = int(d["size"])
                info.update(cin=chans[-1], cout=cout, k=k, stride=int(d["stride"]),
                            pad=(k - 1) // 2, bn=not pre, act=not pre)
                filters = cout
            elif t == "maxpool":
                info.update(k=int(d["size"]), stride=int(d["stride"]))
                filters = chans[-1]
            elif t == "upsample":
                info.update(scale=int(d["stride"]))
                filters = chans[-1]
            elif t == "route":
                idx = [int(v) for v in d["layers"].split(",")]
                info.update(layers=idx)
                filters = sum(chans[(v + 1) if v > 0 else v] for v in idx)   # :93-96
            elif t == "shortcut":
                info.update(frm=int(d["from"]))
                filters = chans[int(d["from"])]
            elif t == "yolo":
                info.update(anchors=[self.anchors[v] for v in self.masks[ycount]])
                ycount += 1
                filters = chans[-1]
            else:
                raise ValueError(t)
            self.layers.append(info)
            chans.append(filters)
        self.params = {}
        self.keep_outs = False          # debugging aid: keep per-layer outputs (and their grads) of the last forward
        self.last_outs = None
        if seed is not None:
            self.init_params(seed)

    def init_params(self, seed):
        g = torch.Generator().manual_seed(seed)
        for i, L in enumerate(self.layers):
            if L["type"] != "convolutional":
                continue
            fan_in = L["cin"] * L["k"] * L["k"]
            bound = 1.0 / math.sqrt(fan_in)
            self.params[f"conv{i}.weight"] = (torch.rand(L["cout"], L["cin"], L["k"], L["k"], generator=g) * 2 - 1) * bound
            if L["bn"]:
                self.params[f"bn{i}.weight"] = torch.rand(L["cout"], generator=g) * 0.5 + 0.75
                self.params[f"bn{i}.bias"] = (torch.rand(L["cout"], generator=g) - 0.5) * 0.2
                self.params[f"bn{i}.running_mean"] = torch.zeros(L["cout"])
                self.params[f"bn{i}.running_var"] = torch.ones(L["cout"])
            else:
                self.params[f"conv{i}.bias"] = (torch.rand(L["cout"], generator=g) * 2 - 1) * bound

    def trainable(self):
        return {k: v for k, v in self.params.items() if "running" not in k}

    def forward(self, x, targets=None, bn_train=True):
        outs, heads = [], []
        total_parts = torch.zeros(6)
        P = self.params
        for i, L in enumerate(self.layers):
            t = L["type"]
            if t == "convolutional":
                x = F.conv2d(x, P[f"conv{i}.weight"], P.get(f"conv{i}.bias"), stride=L["stride"], padding=L["pad"])
                if L["bn"]:
                    x = F.batch_norm(x, P[f"bn{i}.running_mean"], P[f"bn{i}.running_var"],
                                     P[f"bn{i}.weight"], P[f"bn{i}.bias"], training=bn_train,
                                     momentum=0.1, eps=1e-5)
                if L["act"]:
                    x = F.leaky_relu(x, self.slope) if self.act == "leaky" else F.relu(x)
            elif t == "maxpool":
                if L["k"] == 2 and L["stride"] == 1:
                    x = F.pad(x, (0, 1, 0, 1))                     # models.py:77-79
                x = F.max_pool2d(x, L["k"], L["stride"], (L["k"] - 1) // 2)
            elif t == "upsample":
                x = F.interpolate(x, scale_factor=L["scale"], mode="nearest")
            elif t == "route":
                x = torch.cat([outs[v] for v in L["layers"]], 1)
            elif t == "shortcut":
                x = outs[-1] + outs[L["frm"]]
            elif t == "yolo":
                r = yolo_layer(x, L["anchors"], self.classes, self.cfg_h, targets,
                               ignore_thres=self.ignore, **self.consts)
                if targets is not None:
                    x, parts = r
                    total_parts = total_parts + parts
                else:
                    x = r
                heads.append(x)
            if self.keep_outs and x.requires_grad:
                x.retain_grad()
            outs.append(x)
        self.last_outs = outs if self.keep_outs else None
        if targets is not None:
            return (sum(heads), *total_parts)
        return torch.cat(heads, 1)

    # darknet .weights (models.py:339-422): int32[5] header, then per conv:
    # BN bias, BN weight, running_mean, running_var, conv weight | preyolo: bias, weight
    def save_weights(self, path, header=(0, 0, 0, 0, 0)):
        with open(path, "wb") as fp:
            np.asarray(header, np.int32).tofile(fp)
            for i, L in enumerate(self.layers):
                if L["type"] != "convolutional":
                    continue
                if L["bn"]:
                    for nm in ("bias", "weight", "running_mean", "running_var"):
                        self.params[f"bn{i}.{nm}"].detach().numpy().astype(np.float32).tofile(fp)
                else:
                    self.params[f"conv{i}.bias"].detach().numpy().astype(np.float32).tofile(fp)
                self.params[f"conv{i}.weight"].detach().numpy().astype(np.float32).tofile(fp)

    def load_weights(self, path, start_dims=None):
        with open(path, "rb") as fp:
            header = np.fromfile(fp, np.int32, 5)
            w = np.fromfile(fp, np.float32)
        ptr, yc = 0, 0

        def take(n, shape):
            nonlocal ptr
            v = torch.from_numpy(w[ptr:ptr + n].copy()).view(shape)
            ptr += n
            return v
        for i, L in enumerate(self.layers):
            if L["type"] != "convolutional":
                continue
            co, ci, k = L["cout"], L["cin"], L["k"]
            if L["bn"]:
                for nm in ("bias", "weight", "running_mean", "running_var"):
                    self.params[f"bn{i}.{nm}"] = take(co, (co,))
                self.params[f"conv{i}.weight"] = take(co * ci * k * k, (co, ci, k, k))
            else:
                od = start_dims[yc] if start_dims else co          # models.py:380-394
                yc += 1
                self.params[f"conv{i}.bias"] = torch.from_numpy(w[ptr:ptr + co].copy())
                ptr += od
                self.params[f"conv{i}.weight"] = take(od * ci * k * k, (od, ci, k, k))[:co].clone()
        return header
