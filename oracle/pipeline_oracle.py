"""CPU oracle for the detect -> crop -> keypoint glue.

TEST INFRASTRUCTURE ONLY (never imported by the product package).

PARITY UNPINNED against the reference for `crop_resize`: the reference has no joint pipeline code, and the one
piece of semantics it does fix — `prep_image`, RektNet/utils.py:73-76, is `cv2.resize(image, target_image_size)`
with OpenCV's default INTER_LINEAR — lives in a third-party dependency (opencv-python, unpinned in the reference's
requirements) that is absent from this image.  What is restated here is OpenCV's published float32 bilinear
algorithm (resize.cpp, `resizeGeneric_<HResizeLinear, VResizeLinear>`): half-pixel centres
`fx = (dx + 0.5) * (src/dst) - 0.5` evaluated in double and cast to float, `sx = floor(fx)`, taps clamped to the
image edge with the weight moved onto the edge pixel, horizontal blend first, then vertical, all in float32.
It is cross-checked in tests/test_oracle_golden.py against torch.nn.functional.interpolate(bilinear,
align_corners=False) — an independent implementation of the same convention — to 1e-5.

The pipeline order (eval Darknet -> conf filter -> NMS -> boxes back to frame coordinates -> crop -> 80x80 ->
KeypointNet eval) follows CVC-YOLOv3/detect.py:62-101 and RektNet/detect.py:29-39; those parts are pinned through
oracle/postprocess_oracle.py, oracle/yolo_oracle.py and oracle/rektnet_oracle.py.
"""
import numpy as np

F = np.float32


def crop_bounds(box, H, W, scale=(1.0, 1.0), offset=(0.0, 0.0)):
    """Detector-space corner box -> integer pixel bounds [x1, x2) x [y1, y2) in the frame.
    x' = x * scale_x + off_x (CVC-YOLOv3/detect.py:98-101: x / ratio - pad_w), outward rounding, clamped so that
    the crop is at least one pixel and inside the frame."""
    x1 = F(F(box[0]) * F(scale[0]) + F(offset[0])); y1 = F(F(box[1]) * F(scale[1]) + F(offset[1]))
    x2 = F(F(box[2]) * F(scale[0]) + F(offset[0])); y2 = F(F(box[3]) * F(scale[1]) + F(offset[1]))
    ix1 = int(min(max(np.floor(x1), 0), W - 1)); iy1 = int(min(max(np.floor(y1), 0), H - 1))
    ix2 = int(min(max(np.ceil(x2), ix1 + 1), W)); iy2 = int(min(max(np.ceil(y2), iy1 + 1), H))
    return ix1, iy1, ix2, iy2


def _taps(dst, src):
    """OpenCV's INTER_LINEAR tap table for one axis: (index0, index1, w0, w1) per destination pixel."""
    sc = float(src) / float(dst)
    d = np.arange(dst, dtype=np.float64)
    fx = ((d + 0.5) * sc - 0.5).astype(F)
    sx = np.floor(fx).astype(np.int64)
    fx = (fx - sx.astype(F)).astype(F)
    lo = sx < 0
    sx[lo] = 0; fx[lo] = 0
    hi = sx >= src - 1
    sx[hi] = src - 1; fx[hi] = 0
    s1 = np.minimum(sx + 1, src - 1)
    return sx, s1, (F(1) - fx).astype(F), fx


def resize_bilinear(img, out_h, out_w):
    """img [C,h,w] float32 -> [C,out_h,out_w]; cv2.resize(img, (out_w, out_h)) on a float image."""
    img = np.asarray(img, F)
    x0, x1, a0, a1 = _taps(out_w, img.shape[2])
    y0, y1, b0, b1 = _taps(out_h, img.shape[1])
    rows0 = (img[:, y0][:, :, x0] * a0 + img[:, y0][:, :, x1] * a1).astype(F)     # horizontal pass of the two source rows
    rows1 = (img[:, y1][:, :, x0] * a0 + img[:, y1][:, :, x1] * a1).astype(F)
    return (rows0 * b0[None, :, None] + rows1 * b1[None, :, None]).astype(F)


def _taps_u8(dst, src, clamp_weights):
    """OpenCV's 8-bit INTER_LINEAR tap table for one axis (resize.cpp, `resize()` coefficient loop with fixpt = true):
    fx = (float)((d + 0.5) * scale - 0.5) with scale = 1. / ((double)dst / src); sx = cvFloor(fx); fx -= sx.
    x axis (clamp_weights): sx < 0 -> sx = 0, fx = 0 ; sx >= src - 1 -> sx = src - 1, fx = 0.
    y axis: the weights stay, the two ROW indices are clamped instead (`clip(sy + k, 0, ssize.height)`).
    Coefficients are `saturate_cast<short>(c * INTER_RESIZE_COEF_SCALE)` = round-half-even(c * 2048)."""
    sc = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    fx = ((d + 0.5) * sc - 0.5).astype(F)
    sx = np.floor(fx).astype(np.int64)
    fx = (fx - sx.astype(F)).astype(F)
    if clamp_weights:
        lo = sx < 0
        sx[lo] = 0; fx[lo] = 0
        hi = sx >= src - 1
        sx[hi] = src - 1; fx[hi] = 0
    c0 = np.rint((F(1) - fx).astype(F) * F(2048)).astype(np.int64)
    c1 = np.rint(fx * F(2048)).astype(np.int64)
    return np.clip(sx, 0, src - 1), np.clip(sx + 1, 0, src - 1), c0, c1


def resize_bilinear_u8(img, out_h, out_w):
    """img [C,h,w] uint8 -> [C,out_h,out_w] uint8; cv2.resize(img, (out_w, out_h)) on an 8-bit image (what RektNet/utils.py:73-76
    does to the cv2.imread output in dataset.py:35-38 and detect.py:29-32): HResizeLinear in 11-bit fixed point, then
    VResizeLinear's `uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2)`.  PARITY UNPINNED like resize_bilinear
    (cv2 is not in the image); cross-checked against the float formulation to +-1 grey level in tests/test_oracle_golden.py."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    s = img.astype(np.int64)
    x0, x1, a0, a1 = _taps_u8(out_w, img.shape[2], True)
    y0, y1, b0, b1 = _taps_u8(out_h, img.shape[1], False)
    d0 = s[:, y0][:, :, x0] * a0 + s[:, y0][:, :, x1] * a1
    d1 = s[:, y1][:, :, x0] * a0 + s[:, y1][:, :, x1] * a1
    v = (((b0[None, :, None] * (d0 >> 4)) >> 16) + ((b1[None, :, None] * (d1 >> 4)) >> 16) + 2) >> 2
    return (v & 0xFF).astype(np.uint8)


def to_u8(frames):
    """[0,1] float frames -> the 8-bit image (round-half-even(x * 255), clamped); uint8 passes through."""
    frames = np.asarray(frames)
    if frames.dtype == np.uint8:
        return frames
    return np.clip(np.rint(frames.astype(F) * F(255)), 0, 255).astype(np.uint8)


def crop_resize(frames, boxes, count, out_h=80, out_w=80, scale=(1.0, 1.0), offset=(0.0, 0.0), u8=False):
    """frames [B,C,H,W], boxes [B,K,4] corner, count [B] -> ([M,C,out_h,out_w], image index [M]); image-major order.
    u8: the reference's order of operations — resize the 8-bit image, then `image.transpose((2, 0, 1)) / 255.0` in float64 and one
    rounding to float32 (dataset.py:52, detect.py:33-34)."""
    frames = to_u8(frames) if u8 else np.asarray(frames, F)
    B, C, H, W = frames.shape
    out, owner = [], []
    for b in range(B):
        for k in range(int(count[b])):
            x1, y1, x2, y2 = crop_bounds(boxes[b, k], H, W, scale, offset)
            if u8:
                out.append((resize_bilinear_u8(frames[b, :, y1:y2, x1:x2], out_h, out_w).astype(np.float64) / 255.0).astype(F))
            else:
                out.append(resize_bilinear(frames[b, :, y1:y2, x1:x2], out_h, out_w))
            owner.append(b)
    if not out:
        return np.zeros((0, C, out_h, out_w), F), np.zeros((0,), np.int64)
    return np.stack(out).astype(F), np.asarray(owner, np.int64)
