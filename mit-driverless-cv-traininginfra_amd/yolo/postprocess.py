"""Batched detection post-processing on the GPU.

One call replaces the reference's per-image Python loop `validate.py:80-141` for a whole batch: confidence filter,
xywh->corner, greedy NMS (top_k 200), IoU matching against the zero-padded labels and the per-image AP / recall /
precision — two kernel launches (csrc/postprocess.hip), no host round trip.
"""
import torch

from .. import _lib

L_MAX_TOPK = 512      # MDCV_NMS_MAX_TOPK


class Detections:
    """Device-resident result of `detect_postprocess`.  Row b holds `count[b]` valid entries, highest confidence first."""

    __slots__ = ("boxes", "prob", "cls", "index", "correct", "count", "stats")

    def __init__(self, boxes, prob, cls, index, correct, count, stats):
        self.boxes, self.prob, self.cls, self.index, self.correct, self.count, self.stats = boxes, prob, cls, index, correct, count, stats

    def image(self, b):
        """Per-image views trimmed to the kept detections (one device sync)."""
        n = int(self.count[b].item())
        return dict(boxes=self.boxes[b, :n], prob=self.prob[b, :n], cls=self.cls[b, :n], index=self.index[b, :n],
                    correct=self.correct[b, :n], ap=self.stats[b, 0], r=self.stats[b, 1], p=self.stats[b, 2],
                    valid=bool(self.stats[b, 3].item() > 0))


def detect_postprocess(output, targets, conf_thres, nms_thres, iou_thres, width, height, top_k=200):
    """output [B,N,5+C] (eval-mode `Darknet.forward`), targets [B,T,5] zero-padded labels or None.

    Returns a `Detections`; `stats[b] = (AP, recall, precision, valid)` where valid is 0 for the images the reference
    skips (nothing kept after NMS, validate.py:97, or no real label row, validate.py:120)."""
    _lib.require_gpu(output)
    L = _lib.lib()
    if output.dim() != 3 or output.shape[2] < 5:
        raise ValueError("detect_postprocess: output must be [B, N, 5 + num_classes]")
    if not 0 < int(top_k) <= L_MAX_TOPK:
        raise ValueError(f"detect_postprocess: top_k must be in 1..{L_MAX_TOPK} (got {top_k})")
    dev = output.device
    out = output.detach().to(torch.float32).contiguous()
    B, N, C = int(out.shape[0]), int(out.shape[1]), int(out.shape[2]) - 5
    tg, T = None, 0
    if targets is not None and targets.numel() > 0:
        tg = targets.detach().to(device=dev, dtype=torch.float32).contiguous()
        if tg.dim() != 3 or tg.shape[0] != B or tg.shape[2] != 5:
            raise ValueError("detect_postprocess: targets must be [B, T, 5]")
        T = int(tg.shape[1])
    k = int(top_k)
    boxes = torch.empty(B, k, 4, dtype=torch.float32, device=dev)      # the kernel writes every entry (zeros past count[b])
    prob = torch.empty(B, k, dtype=torch.float32, device=dev)
    cls = torch.empty(B, k, dtype=torch.int32, device=dev)
    index = torch.empty(B, k, dtype=torch.long, device=dev)
    correct = torch.empty(B, k, dtype=torch.uint8, device=dev)
    count = torch.empty(B, dtype=torch.int32, device=dev)
    stats = torch.empty(B, 4, dtype=torch.float32, device=dev)
    ws = torch.empty(int(L.detect_post_workspace_bytes(B, N)), dtype=torch.uint8, device=dev)
    L.check(L.detect_post(out.data_ptr(), B, N, C, tg.data_ptr() if tg is not None else None, T, float(conf_thres), float(nms_thres),
                          float(iou_thres), float(width), float(height), k, boxes.data_ptr(), prob.data_ptr(), cls.data_ptr(),
                          index.data_ptr(), correct.data_ptr(), count.data_ptr(), stats.data_ptr(), ws.data_ptr(),
                          torch.cuda.current_stream().cuda_stream), "detect_post")
    return Detections(boxes, prob, cls, index, correct, count, stats)

