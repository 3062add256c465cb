"""GPU parity for the detection post-processing row: HIP kernels (through the C ABI) vs the oracle
and vs the golden vectors produced by the reference's own nms / bbox_iou / average_precision."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import postprocess_oracle as PO  # noqa: E402

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _mods():
    import importlib
    pkg = importlib.import_module("mdcv")
    from mdcv.yolo.utils.nms import nms
    from mdcv.yolo.utils.utils import average_precision, compute_ap, xywh2xyxy
    from mdcv.yolo.postprocess import detect_postprocess
    from mdcv.yolo.validate import validate
    return pkg, nms, average_precision, compute_ap, xywh2xyxy, detect_postprocess, validate


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_nms_golden_bit_exact():
    _, nms, *_ = _mods()
    g = np.load(os.path.join(G, "post_nms.npz"))
    for ci in range(int(g["n_cases"])):
        tk = int(g[f"topk{ci}"])
        keep = nms(_dev(g[f"boxes{ci}"]), _dev(g[f"scores{ci}"]), float(g[f"overlap{ci}"]), tk)
        assert keep.dtype == torch.long and keep.is_cuda
        np.testing.assert_array_equal(keep.cpu().numpy(), g[f"keep{ci}"], err_msg=f"case {ci}")


def _cluster_boxes(n, rng, span=416.0):
    k = max(1, n // 12)
    cen = rng.random((k, 2), dtype=np.float32) * span
    wh = rng.random((k, 2), dtype=np.float32) * 60 + 8
    w = rng.integers(0, k, n)
    c = cen[w] + rng.standard_normal((n, 2)).astype(np.float32) * 4
    s = wh[w] * np.clip(1 + 0.15 * rng.standard_normal((n, 2)).astype(np.float32), 0.3, 2.0)
    return np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)


@pytest.mark.parametrize("n,quant,overlap,top_k", [(5000, 64, 0.45, 200), (10647, 16, 0.25, 200), (22743, 1000, 0.25, 200),
                                                   (22743, None, 0.6, 512), (300, 2, 0.5, 200), (513, 1, 0.3, 512), (70000, 256, 0.5, 200)])
def test_nms_ties_and_sizes_vs_oracle(n, quant, overlap, top_k):
    """Tied scores (sigmoid saturation, low-precision confidences): the stable visiting order the header defines."""
    _, nms, *_ = _mods()
    rng = np.random.default_rng(n + top_k)
    boxes = _cluster_boxes(n, rng)
    boxes[::97, 2:] = boxes[::97, :2]            # zero-area boxes: 0/0 IoU
    scores = rng.random(n, dtype=np.float32)
    if quant:
        scores = (np.round(scores * quant) / quant).astype(np.float32)
    keep = nms(_dev(boxes), _dev(scores), overlap, top_k)
    np.testing.assert_array_equal(keep.cpu().numpy(), PO.nms(boxes, scores, overlap, top_k))


def test_nms_argument_errors():
    _, nms, *_ = _mods()
    b, s = torch.rand(10, 4).cuda(), torch.rand(10).cuda()
    with pytest.raises(ValueError):
        nms(b, s, 0.5, 513)
    with pytest.raises(ValueError):
        nms(b, s, 0.5, 0)
    assert nms(torch.zeros(0, 4).cuda(), torch.zeros(0).cuda()).numel() == 0
    with pytest.raises(Exception):
        nms(b.cpu(), s.cpu())                    # no CPU fallback


def _check_image(d, ref, exact_ap=True):
    assert d["boxes"].shape[0] == ref["count"]
    np.testing.assert_array_equal(d["index"].cpu().numpy(), ref["index"])
    np.testing.assert_array_equal(d["boxes"].cpu().numpy(), ref["boxes"])
    np.testing.assert_array_equal(d["prob"].cpu().numpy(), ref["prob"])
    np.testing.assert_array_equal(d["cls"].cpu().numpy(), ref["cls"])
    assert d["valid"] == bool(ref["valid"])
    if ref["valid"]:
        np.testing.assert_array_equal(d["correct"].cpu().numpy(), ref["correct"])
        got = np.array([float(d["ap"]), float(d["r"]), float(d["p"])], np.float32)
        want = np.array([ref["ap"], ref["r"], ref["p"]], np.float32)
        np.testing.assert_array_equal(got, want)        # same float32 operation order as the oracle


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_detect_postprocess_golden(name):
    *_, detect_postprocess, _ = _mods()
    g = np.load(os.path.join(G, f"post_validate_{name}.npz"))
    args = (float(g["conf_thres"]), float(g["nms_thres"]), float(g["iou_thres"]), float(g["width"]), float(g["height"]))
    det = detect_postprocess(_dev(g["out"]), _dev(g["targets"]), *args)
    aps = []
    for b in range(g["out"].shape[0]):
        d = det.image(b)
        assert d["boxes"].shape[0] == int(g[f"count{b}"])
        np.testing.assert_array_equal(d["boxes"].cpu().numpy(), g[f"boxes{b}"])
        np.testing.assert_array_equal(d["prob"].cpu().numpy(), g[f"prob{b}"])
        np.testing.assert_array_equal(d["cls"].cpu().numpy(), g[f"cls{b}"])
        assert d["valid"] == bool(g[f"valid{b}"])
        if d["valid"]:
            np.testing.assert_array_equal(d["correct"].cpu().numpy(), g[f"correct{b}"])
            ref = g[f"apr{b}"]
            assert abs(float(d["ap"]) - float(ref[0])) <= 1e-6      # the reference's torch.sum order differs in the last ulp
            assert float(d["r"]) == float(ref[1]) and float(d["p"]) == float(ref[2])
            aps.append(float(d["ap"]))
        _check_image(d, PO.postprocess_image(g["out"][b], g["targets"][b], *args))
    assert abs(np.mean(aps) - float(g["means"][0])) <= 1e-6


def _synth_output(B, N, C, T, rng, span, sat=False):
    tg = np.zeros((B, T, 5), np.float32)
    out = np.zeros((B, N, 5 + C), np.float32)
    for b in range(B):
        n = int(rng.integers(0 if b % 7 == 3 else 1, T + 1))
        tg[b, :n, 0] = rng.integers(0, max(C, 1), n)
        tg[b, :n, 1:3] = rng.random((n, 2)) * 0.9 + 0.05
        tg[b, :n, 3:5] = rng.random((n, 2)) * 0.28 + 0.02
        out[b, :, 0:2] = rng.random((N, 2)) * span
        out[b, :, 2:4] = rng.random((N, 2)) * 80 + 4
        out[b, :, 4] = rng.random(N) * 0.7
        out[b, :, 5:] = rng.random((N, C))
        rows = rng.permutation(N)
        r = 0
        for lab in tg[b, :n]:
            for _ in range(int(rng.integers(1, 6))):
                i = rows[r]; r += 1
                out[b, i, 0:4] = lab[1:5] * span * (1 + 0.08 * rng.standard_normal(4))
                out[b, i, 4] = 0.8 + 0.2 * rng.random()
        if sat:                                   # saturated / low-precision confidences: many ties
            out[b, :, 4] = np.round(out[b, :, 4] * 128) / 128
        if b % 11 == 5:
            out[b, :, 4] = 0.01                   # nothing passes
    return out.astype(np.float32), tg


@pytest.mark.parametrize("B,N,C,T,span,conf,sat", [(32, 10647, 80, 30, 416, 0.5, False), (8, 22743, 1, 100, 608, 0.8, True),
                                                   (5, 10647, 1, 8, 416, 0.0, True), (3, 200, 2, 4, 416, 0.3, False)])
def test_detect_postprocess_full_size_vs_oracle(B, N, C, T, span, conf, sat):
    *_, detect_postprocess, _ = _mods()
    rng = np.random.default_rng(B * 1000 + C)
    out, tg = _synth_output(B, N, C, T, rng, float(span), sat)
    det = detect_postprocess(_dev(out), _dev(tg), conf, 0.25, 0.5, span, span)
    torch.cuda.synchronize()
    for b in range(B):
        _check_image(det.image(b), PO.postprocess_image(out[b], tg[b], conf, 0.25, 0.5, span, span))
    # size-independent properties: kept confidences sorted, every kept box above the threshold, counts bounded
    cnt = det.count.cpu().numpy()
    assert cnt.max() <= 200
    for b in range(B):                           # entries past count[b] are zeros
        assert float(det.boxes[b, cnt[b]:].abs().sum()) == 0 and int(det.index[b, cnt[b]:].abs().sum()) == 0
        assert float(det.prob[b, cnt[b]:].abs().sum()) == 0 and int(det.correct[b, cnt[b]:].sum()) == 0
    for b in range(B):
        p = det.prob[b, :cnt[b]].cpu().numpy()
        assert np.all(p[:-1] >= p[1:]) and np.all(p > conf)
    # no labels: detections only
    det2 = detect_postprocess(_dev(out), None, conf, 0.25, 0.5, span, span)
    np.testing.assert_array_equal(det2.count.cpu().numpy(), cnt)
    np.testing.assert_array_equal(det2.index.cpu().numpy(), det.index.cpu().numpy())
    assert float(det2.stats.abs().sum()) == 0.0


def test_nms_idempotent_on_its_own_output():
    """Property at full size: running NMS again on the kept boxes keeps all of them, in the same order."""
    _, nms, *_ = _mods()
    rng = np.random.default_rng(9)
    boxes = _dev(_cluster_boxes(22743, rng, 608.0))
    scores = _dev(rng.random(22743, dtype=np.float32))
    k1 = nms(boxes, scores, 0.25, 200)
    k2 = nms(boxes[k1], scores[k1], 0.25, 200)
    assert torch.equal(k2, torch.arange(k1.numel(), device=k2.device))


def test_average_precision_golden_and_oracle():
    _, _, average_precision, compute_ap, xywh2xyxy, *_ = _mods()
    g = np.load(os.path.join(G, "post_ap.npz"))
    for ci in range(int(g["n_cases"])):
        ap, r, p = average_precision(_dev(g[f"tp{ci}"]), _dev(g[f"conf{ci}"]), int(g[f"ngt{ci}"]))
        ref = g[f"apr{ci}"]
        assert abs(float(ap) - float(ref[0])) <= 1e-6 and float(r) == float(ref[1])
        assert float(p) == float(ref[2]) or (np.isnan(float(p)) and np.isnan(ref[2]))
        o = PO.average_precision(g[f"tp{ci}"], g[f"conf{ci}"], int(g[f"ngt{ci}"]))
        assert float(ap) == float(o[0])
    rng = np.random.default_rng(3)
    for m in (33, 200, 512):                       # tied confidences: stable order (lower index first)
        conf = (np.round(rng.random(m) * 8) / 8).astype(np.float32)
        tp = (rng.random(m) < 0.5).astype(np.uint8)
        ap, r, p = average_precision(_dev(tp), _dev(conf), 300)
        o = PO.average_precision(tp, conf, 300)
        assert (float(ap), float(r), float(p)) == (float(o[0]), float(o[1]), float(o[2]))
    assert abs(float(compute_ap(_dev(g["ca_rec"]), _dev(g["ca_pre"]))) - float(g["ca_ap"])) <= 1e-6
    x = rng.random((9, 4)).astype(np.float32)
    np.testing.assert_array_equal(xywh2xyxy(_dev(x)).cpu().numpy(), PO.xywh2xyxy(x))
    with pytest.raises(ValueError):
        average_precision(torch.zeros(513).cuda(), torch.zeros(513).cuda(), 1)


def test_validate_drop_in_matches_oracle_means():
    """validate(dataloader=..., model=..., device=...) with a stub model that replays recorded eval outputs."""
    *_, validate = _mods()
    rng = np.random.default_rng(17)
    batches = [_synth_output(4, 2028, 1, 10, rng, 416.0) for _ in range(3)]

    class Replay(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.i = 0

        def get_threshs(self):
            return 0.5, 0.25, 0.5

        def img_size(self):
            return 416, 416

        def forward(self, imgs):
            o = _dev(batches[self.i][0]); self.i += 1
            return o

    class DS(list):
        dataset = list(range(12))

    loader = DS([(None, torch.zeros(4, 3, 8, 8), torch.from_numpy(t)) for _, t in batches])
    m_ap, m_r, m_p, per_img = validate(dataloader=loader, model=Replay(), device=torch.device("cuda:0"), debug_mode=False)
    ref = PO.validate_batches([o for o, _ in batches], [t for _, t in batches], 0.5, 0.25, 0.5, 416, 416)
    assert abs(m_ap - ref[0]) <= 1e-6 and abs(m_r - ref[1]) <= 1e-6 and abs(m_p - ref[2]) <= 1e-6 and per_img > 0
