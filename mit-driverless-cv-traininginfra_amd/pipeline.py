"""Joint detect -> crop -> keypoint pipeline on one GPU (BASELINE.json config 5).

The reference ships the two halves separately (CVC-YOLOv3/detect.py:62-101 runs the detector + NMS on one image,
RektNet/detect.py:29-39 runs KeypointNet on one pre-cut cone image resized with cv2, RektNet/utils.py:73-76) and has no
code that joins them.  Here the join is device-resident: eval-mode Darknet -> `detect_postprocess` (conf filter + NMS for the
whole batch) -> `crop_resize` (one launch: every kept box is cut out of its frame and bilinearly resampled to the
KeypointNet input size) -> one batched KeypointNet eval.  Only the number of crops is read back (it sizes the batch).
"""
import torch

from . import _lib
from .yolo.postprocess import detect_postprocess


def to_u8(frames):
    """[0,1] float frames -> the uint8 image they were decoded from (round(x * 255)); uint8 input passes through."""
    if frames.dtype == torch.uint8:
        return frames.contiguous()
    return (frames.detach().to(torch.float32) * 255.0).round_().clamp_(0, 255).to(torch.uint8).contiguous()


def crop_resize(frames, boxes, count, out_size=(80, 80), scale=(1.0, 1.0), offset=(0.0, 0.0), pad_rows_to=1, u8=None):
    """frames [B,C,H,W] fp32 (or uint8), boxes [B,K,4] corner boxes in detector coordinates, count [B] (int32) -> (crops, owner, M).

    Two resampling rules (both cv2.resize's default INTER_LINEAR, RektNet/utils.py:73-76):
      * u8=True (default for uint8 frames): what the reference's loaders do — the frame is an 8-bit image, cv2 resizes it in 11-bit
        fixed point and rounds to 8 bits, THEN the crop is divided by 255 (dataset.py:35-38,52; detect.py:29-35).  Float frames in
        [0,1] are first put back on the 8-bit grid (`to_u8`).
      * u8=False (default for float frames): OpenCV's float32 formulation applied to the float frame, no 8-bit rounding — for
        frames that never were 8-bit (synthetic streams, already-normalised tensors).

    crops is [Mpad, C, out_h, out_w] with the M real crops first (image-major, box order kept) and zero rows up to the next
    multiple of `pad_rows_to`; owner [Mpad] is the frame index of each real crop.  Box coordinates are mapped to frame pixels
    as x * scale + offset, e.g. scale = 1/ratio, offset = (-pad_w, -pad_h) of `calculate_padding` (detect.py:98-101).
    One device sync (the crop count sizes the output)."""
    _lib.require_gpu(frames)
    L = _lib.lib()
    if frames.dim() != 4 or boxes.dim() != 3 or boxes.shape[2] != 4 or boxes.shape[0] != frames.shape[0]:
        raise ValueError("crop_resize: frames [B,C,H,W], boxes [B,K,4]")
    oh, ow = int(out_size[0]), int(out_size[1])
    if not (0 < oh <= 256 and 0 < ow <= 256):
        raise ValueError("crop_resize: output side must be in 1..256")
    dev = frames.device
    if u8 is None:
        u8 = frames.dtype == torch.uint8
    fr = to_u8(frames) if u8 else frames.detach().to(torch.float32).contiguous()
    bx = boxes.detach().to(torch.float32).contiguous()
    B, C, H, W = (int(v) for v in fr.shape)
    K = int(bx.shape[1])
    cnt = count.detach().to(device=dev, dtype=torch.int32).contiguous()
    M = int(cnt.clamp(max=K).sum().item())
    pad = max(1, int(pad_rows_to))
    Mpad = max(pad, (M + pad - 1) // pad * pad)
    out = torch.empty(Mpad, C, oh, ow, dtype=torch.float32, device=dev)
    if Mpad > M:
        out[M:].zero_()
    owner = torch.zeros(Mpad, dtype=torch.int32, device=dev)
    total = torch.zeros(1, dtype=torch.int32, device=dev)
    if K > 0:
        L.check((L.crop_resize_u8 if u8 else L.crop_resize)(fr.data_ptr(), B, C, H, W, bx.data_ptr(), cnt.data_ptr(), K, float(scale[0]), float(scale[1]),
                              float(offset[0]), float(offset[1]), oh, ow, out.data_ptr(), owner.data_ptr(), total.data_ptr(),
                              torch.cuda.current_stream().cuda_stream), "crop_resize")
    return out, owner, M


class JointPipeline:
    """detector: eval-mode `Darknet`; keypoint_net: eval-mode `KeypointNet`.  `__call__(imgs, frames=None)`:
    imgs [B,3,H,W] is the detector input; crops are cut from `frames` (defaults to imgs) after mapping the boxes with
    `scale` / `offset`.  `frames` as uint8 [B,3,H,W] (the decoded camera image, as the reference's cv2.imread delivers it) takes the
    reference's 8-bit resize rule; float frames take the float rule unless `u8=True` (see crop_resize).  Returns a dict: det (Detections), crops, owner, num (M), keypoints [M,K,2] (normalised x,y in the
    crop) and keypoints_frame [M,K,2] (frame pixels)."""

    def __init__(self, detector, keypoint_net, conf_thres=None, nms_thres=None, top_k=200, max_cones=64, bucket=64, u8=None):
        self.u8 = u8
        self.detector, self.keypoint_net = detector, keypoint_net
        c, n, _ = detector.get_threshs() if hasattr(detector, "get_threshs") else (0.8, 0.25, 0.5)
        self.conf_thres = float(c if conf_thres is None else conf_thres)
        self.nms_thres = float(n if nms_thres is None else nms_thres)
        self.top_k, self.max_cones, self.bucket = int(top_k), int(max_cones), int(bucket)

    @torch.no_grad()
    def __call__(self, imgs, frames=None, scale=(1.0, 1.0), offset=(0.0, 0.0)):
        self.detector.eval()
        self.keypoint_net.eval()
        output = self.detector(imgs)
        width, height = (self.detector.img_size() if hasattr(self.detector, "img_size") else (imgs.shape[3], imgs.shape[2]))
        det = detect_postprocess(output, None, self.conf_thres, self.nms_thres, 0.5, width, height, self.top_k)
        kcap = min(self.max_cones, self.top_k)
        src = imgs if frames is None else frames
        size = tuple(self.keypoint_net.image_size)
        crops, owner, M = crop_resize(src, det.boxes[:, :kcap], det.count, size, scale, offset, pad_rows_to=self.bucket, u8=self.u8)
        res = dict(det=det, crops=crops[:M], owner=owner[:M], num=M, keypoints=None, keypoints_frame=None)
        if M == 0:
            return res
        _hm, pts = self.keypoint_net(crops)
        pts = pts[:M]
        res["keypoints"] = pts
        # back to frame pixels: the crop rectangle is the outward-rounded, clamped box (same rule as the kernel)
        cnt = det.count.clamp(max=kcap).long()
        sel = torch.arange(kcap, device=cnt.device)[None, :] < cnt[:, None]
        b = det.boxes[:, :kcap][sel]
        Hs, Ws = int(src.shape[2]), int(src.shape[3])
        x1 = (b[:, 0] * scale[0] + offset[0]).floor().clamp(0, Ws - 1)
        y1 = (b[:, 1] * scale[1] + offset[1]).floor().clamp(0, Hs - 1)
        x2 = torch.minimum(torch.maximum((b[:, 2] * scale[0] + offset[0]).ceil(), x1 + 1), torch.full_like(x1, Ws))
        y2 = torch.minimum(torch.maximum((b[:, 3] * scale[1] + offset[1]).ceil(), y1 + 1), torch.full_like(y1, Hs))
        res["keypoints_frame"] = torch.stack([x1[:, None] + pts[..., 0] * (x2 - x1)[:, None],
                                              y1[:, None] + pts[..., 1] * (y2 - y1)[:, None]], -1)
        return res
