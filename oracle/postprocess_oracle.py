"""CPU oracle for the detection post-processing row.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this file;
it is used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as
the checker for the HIP post-processing kernels.

Pinned: every function below is checked in tests/test_oracle_golden.py against
vectors produced by the reference's own functions imported from /root/reference
(tests/golden/make_golden.py, mode "post"; fixtures tests/golden/post_*.npz).
`validate.py` itself cannot be imported in this image (it needs torchvision), so
the per-image loop `validate.py:80-141` is restated here and pinned through its
parts: `utils/nms.py:4-61`, `utils/utils.py:58-119` (AP), `utils/utils.py:163-193`
(IoU) are called by the fixture generator exactly as the loop calls them.

All arithmetic is float32 in the reference's operation order; index results
(NMS keep list, best-target indices, correct flags) are exact, AP is a float32
sequential sum (the reference's torch.sum may differ in the last ulp; the golden
test allows 1e-6 absolute on AP and nothing on the rest).
"""
import numpy as np

F = np.float32


def nms(boxes, scores, overlap=0.5, top_k=200):
    """Greedy NMS, `utils/nms.py:4-61`.

    Candidates are visited by descending score; equal scores by descending index
    (the reference sorts ascending with a stable sort and walks from the back,
    `nms.py:25-32`).  A candidate survives a kept box when IoU <= overlap, where
    IoU = inter / ((area_j - inter) + area_i) with i the kept box (`nms.py:52-59`);
    a NaN IoU (0/0) therefore removes the candidate.
    Returns the kept indices (int64) in visiting order.
    """
    boxes = np.asarray(boxes, dtype=F).reshape(-1, 4)
    scores = np.asarray(scores, dtype=F).reshape(-1)
    n = scores.shape[0]
    if boxes.size == 0:
        return np.zeros((0,), np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    area = ((x2 - x1) * (y2 - y1)).astype(F)
    order = np.argsort(scores, kind="stable")            # ascending, ties keep index order
    order = order[-top_k:] if top_k > 0 else order[:0]    # nms.py:27
    order = order[::-1]                                   # visit from the back
    alive = np.ones(order.shape[0], bool)
    keep = []
    with np.errstate(invalid="ignore", divide="ignore"):
        for a in range(order.shape[0]):
            if not alive[a]:
                continue
            i = order[a]
            keep.append(i)
            rest = np.nonzero(alive[a + 1:])[0] + a + 1
            if rest.size == 0:
                break
            j = order[rest]
            xx1 = np.maximum(x1[j], x1[i])
            yy1 = np.maximum(y1[j], y1[i])
            xx2 = np.minimum(x2[j], x2[i])
            yy2 = np.minimum(y2[j], y2[i])
            w = np.maximum((xx2 - xx1).astype(F), F(0))
            h = np.maximum((yy2 - yy1).astype(F), F(0))
            inter = (w * h).astype(F)
            union = ((area[j] - inter).astype(F) + area[i]).astype(F)
            iou = (inter / union).astype(F)
            alive[rest] = iou <= F(overlap)               # NaN -> removed
    return np.asarray(keep, np.int64)


def xywh2xyxy(x):
    """`utils/utils.py:121-127`."""
    x = np.asarray(x, dtype=F)
    y = np.zeros_like(x)
    y[:, 0] = x[:, 0] - x[:, 2] / F(2)
    y[:, 1] = x[:, 1] - x[:, 3] / F(2)
    y[:, 2] = x[:, 0] + x[:, 2] / F(2)
    y[:, 3] = x[:, 1] + x[:, 3] / F(2)
    return y


def corner_iou_plus1(b1, b2):
    """`utils/utils.py:163-193` with x1y1x2y2=True (the "+1" pixel convention)."""
    b1 = np.asarray(b1, F)
    b2 = np.asarray(b2, F)
    ix1 = np.maximum(b1[..., 0], b2[..., 0])
    iy1 = np.maximum(b1[..., 1], b2[..., 1])
    ix2 = np.minimum(b1[..., 2], b2[..., 2])
    iy2 = np.minimum(b1[..., 3], b2[..., 3])
    inter = (np.maximum((ix2 - ix1 + F(1)).astype(F), F(0)) * np.maximum((iy2 - iy1 + F(1)).astype(F), F(0))).astype(F)
    a1 = ((b1[..., 2] - b1[..., 0] + F(1)) * (b1[..., 3] - b1[..., 1] + F(1))).astype(F)
    a2 = ((b2[..., 2] - b2[..., 0] + F(1)) * (b2[..., 3] - b2[..., 1] + F(1))).astype(F)
    return (inter / (a1 + a2 - inter + F(1e-12))).astype(F)


def compute_ap(recall, precision):
    """`utils/utils.py:90-119`: precision envelope, then sum of dRecall * precision."""
    mrec = np.concatenate(([F(0)], np.asarray(recall, F), [F(1)])).astype(F)
    mpre = np.concatenate(([F(0)], np.asarray(precision, F), [F(0)])).astype(F)
    for i in range(len(mpre) - 1, 0, -1):
        mpre[i - 1] = max(mpre[i - 1], mpre[i])
    ap = F(0)
    for j in range(len(mrec) - 1):
        if mrec[j + 1] != mrec[j]:
            ap = F(ap + F(F(mrec[j + 1] - mrec[j]) * mpre[j + 1]))
    return ap


def average_precision(tp, conf, n_gt):
    """`utils/utils.py:58-88`.  Returns (ap, recall, precision) as float32."""
    conf = np.asarray(conf, F)
    order = np.argsort(-conf, kind="stable")
    tp = np.asarray(tp)[order].astype(F)
    fpc = np.cumsum(F(1) - tp, dtype=F)
    tpc = np.cumsum(tp, dtype=F)
    denom = F(n_gt + 1e-16)
    recall = (tpc / denom).astype(F)
    with np.errstate(invalid="ignore", divide="ignore"):
        precision = (tpc / (tpc + fpc)).astype(F)
    r = F(tpc[-1] / denom)
    p = F(tpc[-1] / F(tpc[-1] + fpc[-1]))
    return compute_ap(recall, precision), r, p


def postprocess_image(det, labels, conf_thres, nms_thres, iou_thres, width, height, top_k=200):
    """One iteration of the per-image loop `validate.py:80-141`.

    det [N, 5+C] eval-mode rows (cx, cy, w, h, conf, cls...), labels [T, 5]
    zero-padded (cls, cx, cy, w, h) normalised.  Returns a dict; `valid` is False
    where the reference `continue`s (no detections after NMS, `validate.py:97`,
    or no labels, `validate.py:120`).
    """
    det = np.asarray(det, F)
    labels = np.asarray(labels, F)
    sel = np.nonzero(det[:, 4] > F(conf_thres))[0]
    d = det[sel]
    cls = np.argmax(d[:, 5:], axis=1) if d.shape[0] else np.zeros((0,), np.int64)
    half = (d[:, 2:4] / F(2)).astype(F)
    corner = np.concatenate([d[:, 0:2] - half, d[:, 0:2] + half], axis=1).astype(F)
    prob = d[:, 4]
    keep = nms(corner, prob, nms_thres, top_k)
    out = dict(valid=False, count=int(keep.shape[0]), index=sel[keep].astype(np.int64), boxes=corner[keep], prob=prob[keep],
               cls=cls[keep].astype(np.int32), correct=np.zeros(keep.shape[0], np.uint8), ap=F(0), r=F(0), p=F(0),
               best=np.zeros(keep.shape[0], np.int64))
    if keep.shape[0] == 0:
        return out
    # validate.py:100-104 re-sorts by -prob; with a stable sort that is the identity on the keep list.
    lab_ok = (labels[:, 1:5] <= 0).sum(axis=1) == 0
    lab = labels[lab_ok]
    if lab.shape[0] == 0:
        return out
    tb = xywh2xyxy(lab[:, 1:5])
    tb[:, (0, 2)] *= F(width)
    tb[:, (1, 3)] *= F(height)
    ious = corner_iou_plus1(out["boxes"][:, None, :], tb[None, :, :])
    best = np.argmax(ious, axis=1)
    detected = np.zeros(tb.shape[0], bool)
    for i in range(keep.shape[0]):
        if ious[i, best[i]] > F(iou_thres) and not detected[best[i]]:
            out["correct"][i] = 1
            detected[best[i]] = True
    out["best"] = np.nonzero(lab_ok)[0][best].astype(np.int64)     # index into the padded label rows
    out["ap"], out["r"], out["p"] = average_precision(out["correct"], out["prob"], lab.shape[0])
    out["valid"] = True
    return out


def validate_batches(outputs, targets, conf_thres, nms_thres, iou_thres, width, height, top_k=200):
    """Means over the images the loop does not skip, `validate.py:161-164`.
    outputs/targets: sequences of per-batch arrays [B,N,5+C] / [B,T,5]."""
    aps, rs, ps = [], [], []
    for out_b, tgt_b in zip(outputs, targets):
        for det, lab in zip(out_b, tgt_b):
            r = postprocess_image(det, lab, conf_thres, nms_thres, iou_thres, width, height, top_k)
            if r["valid"]:
                aps.append(r["ap"]); rs.append(r["r"]); ps.append(r["p"])
    mean = lambda v: float(np.mean(np.asarray(v, F), dtype=F)) if v else float("nan")
    return mean(aps), mean(rs), mean(ps), len(aps)
