"""GPU drop-ins for the helpers of the reference's CVC-YOLOv3/utils/utils.py that sit on the train / validate path:

  bbox_iou          (utils.py:163-193)  — HIP kernel (csrc/yolo_head.hip: mdcv_bbox_iou) for fp32 boxes on the GPU, bit-identical
  build_targets     (utils.py:195-275)  — HIP kernel (csrc/yolo_head.hip), bit-exact masks / indices
  xywh2xyxy         (utils.py:121-127)  — elementwise helper, plain torch on the boxes' device
  average_precision (utils.py:58-88) / compute_ap (utils.py:90-119) — HIP kernel (csrc/postprocess.hip)

Return order and dtypes follow the reference: mask, conf_mask (uint8), tx, ty, tw, th, tconf (float32), tcls (uint8).
"""
import torch

from ... import _lib


def bbox_iou(box1, box2, x1y1x2y2=True):
    """IoU with the reference's "+1 pixel" convention on [..., 4+] boxes (corner format unless x1y1x2y2=False), leading dimensions broadcast
    as the reference's tensor expression broadcasts them (utils/utils.py:163-193).  fp32 boxes on the GPU go through mdcv_bbox_iou
    (csrc/yolo_head.hip: one thread per pair, every operation rounded as the reference's op chain rounds it -> bit-identical); boxes on the host,
    other dtypes or boxes that carry autograd history take the same expression in torch ops."""
    if (box1.is_cuda and box2.is_cuda and box1.dtype == torch.float32 and box2.dtype == torch.float32
            and not (torch.is_grad_enabled() and (box1.requires_grad or box2.requires_grad))
            and box1.shape[-1] >= 4 and box2.shape[-1] >= 4 and box1.dim() >= 1 and box2.dim() >= 1):
        lead = torch.broadcast_shapes(box1.shape[:-1], box2.shape[:-1])
        n = 1
        for d in lead:
            n *= int(d)
        if n == 0:
            return torch.empty(lead, dtype=torch.float32, device=box1.device)

        def rows(b):                   # -> ([rows, >= 4] contiguous, row count 1 or n)
            if b.shape[:-1].numel() == 1:
                return b.detach().reshape(1, b.shape[-1]).contiguous(), 1
            return b.detach().expand(*lead, b.shape[-1]).reshape(n, b.shape[-1]).contiguous(), n
        r1, n1 = rows(box1)
        r2, n2 = rows(box2)
        out = torch.empty(n, dtype=torch.float32, device=box1.device)
        L = _lib.lib()
        with torch.cuda.device(box1.device):
            L.check(L.bbox_iou(r1.data_ptr(), n1, r1.shape[1], r2.data_ptr(), n2, r2.shape[1], 1 if x1y1x2y2 else 0, out.data_ptr(),
                               torch.cuda.current_stream().cuda_stream), "bbox_iou")
        return out.reshape(lead)
    if x1y1x2y2:
        ax1, ay1, ax2, ay2 = box1[..., 0], box1[..., 1], box1[..., 2], box1[..., 3]
        bx1, by1, bx2, by2 = box2[..., 0], box2[..., 1], box2[..., 2], box2[..., 3]
    else:
        ax1, ax2 = box1[..., 0] - box1[..., 2] / 2, box1[..., 0] + box1[..., 2] / 2
        ay1, ay2 = box1[..., 1] - box1[..., 3] / 2, box1[..., 1] + box1[..., 3] / 2
        bx1, bx2 = box2[..., 0] - box2[..., 2] / 2, box2[..., 0] + box2[..., 2] / 2
        by1, by2 = box2[..., 1] - box2[..., 3] / 2, box2[..., 1] + box2[..., 3] / 2
    iw = torch.clamp(torch.min(ax2, bx2) - torch.max(ax1, bx1) + 1, min=0)
    ih = torch.clamp(torch.min(ay2, by2) - torch.max(ay1, by1) + 1, min=0)
    inter = iw * ih
    area_a = (ax2 - ax1 + 1) * (ay2 - ay1 + 1)
    area_b = (bx2 - bx1 + 1) * (by2 - by1 + 1)
    return inter / (area_a + area_b - inter + 1e-12)


def build_targets(target, anchors, num_anchors, num_classes, grid_size_h, grid_size_w, ignore_thres):
    """target [B,T,5] (cls,cx,cy,w,h; zero rows = padding), anchors [A,2] in grid units, both on the GPU.

    A target whose centre falls outside the grid (cx or cy == 1.0) raises IndexError like the reference (this costs
    one device sync; the fused training path in models.YOLOLayer does not pay it)."""
    _lib.require_gpu(target)
    L = _lib.lib()
    dev = target.device
    tg = target.detach().to(torch.float32).contiguous()
    an = anchors.detach().to(device=dev, dtype=torch.float32).contiguous()
    B, T = tg.shape[0], tg.shape[1]
    A, C, Gh, Gw = int(num_anchors), int(num_classes), int(grid_size_h), int(grid_size_w)
    shape = (B, A, Gh, Gw)
    u8 = lambda *s: torch.empty(s, dtype=torch.uint8, device=dev)          # noqa: E731
    f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)       # noqa: E731
    mask, conf_mask = u8(*shape), u8(*shape)
    tx, ty, tw, th, tconf = f32(*shape), f32(*shape), f32(*shape), f32(*shape), f32(*shape)
    tcls = u8(*shape, C)
    ws = torch.empty(int(L.build_targets_workspace_bytes(B, T, A, Gh, Gw)), dtype=torch.uint8, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    L.check(L.build_targets(tg.data_ptr(), an.data_ptr(), B, T, A, C, Gh, Gw, float(ignore_thres), mask.data_ptr(),
                            conf_mask.data_ptr(), tx.data_ptr(), ty.data_ptr(), tw.data_ptr(), th.data_ptr(), tconf.data_ptr(),
                            tcls.data_ptr(), ws.data_ptr(), err.data_ptr(), stream), "build_targets")
    if int(err.item()) != 0:
        raise IndexError("build_targets: a target centre falls outside the grid (cx or cy >= 1.0)")
    return mask, conf_mask, tx, ty, tw, th, tconf, tcls


def xywh2xyxy(x):
    """[x, y, w, h] -> [x1, y1, x2, y2] on the tensor's own device (utils.py:121-127)."""
    y = torch.zeros(x.shape, device=x.device, dtype=x.dtype)
    half_w, half_h = x[:, 2] / 2, x[:, 3] / 2
    y[:, 0], y[:, 1] = x[:, 0] - half_w, x[:, 1] - half_h
    y[:, 2], y[:, 3] = x[:, 0] + half_w, x[:, 1] + half_h
    return y


def average_precision(tp, conf, n_gt):
    """(ap, recall, precision) as 0-d float32 GPU tensors; tp / conf are 1-D GPU tensors with 1..512 entries.

    Detections are taken by descending confidence, equal confidences in their given order (utils.py:71-72)."""
    _lib.require_gpu(conf)
    L = _lib.lib()
    m = int(conf.shape[0])
    if not 0 < m <= 512:
        raise ValueError(f"average_precision: 1..512 detections supported (got {m})")
    t8 = (tp.detach() != 0).to(torch.uint8).contiguous()
    cf = conf.detach().to(torch.float32).contiguous()
    out = torch.empty(3, dtype=torch.float32, device=conf.device)
    L.check(L.average_precision(t8.data_ptr(), cf.data_ptr(), m, int(n_gt), out.data_ptr(), torch.cuda.current_stream().cuda_stream),
            "average_precision")
    return out[0], out[1], out[2]


def compute_ap(recall, precision):
    """Area under the precision envelope (utils.py:90-119), torch ops on the curves' own device."""
    z, o = recall.new_zeros(1), recall.new_ones(1)
    mrec = torch.cat((z, recall, o))
    mpre = torch.cat((z.to(precision.dtype), precision, z.to(precision.dtype)))
    mpre = torch.flip(torch.cummax(torch.flip(mpre, (0,)), 0)[0], (0,))
    i = torch.nonzero(mrec[1:] != mrec[:-1])
    return torch.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])
