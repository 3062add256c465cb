"""CPU oracle for the on-device synthetic cone data generator.

TEST INFRASTRUCTURE ONLY (never imported by the product package).

What is restated from the reference is the OUTPUT CONTRACT of its two datasets, not their file I/O:
  * `ImageLabelDataset.__getitem__` (CVC-YOLOv3/utils/datasets.py:124-315): image float [3,H,W] in [0,1], labels [T,5]
    (class, cx, cy, w, h) normalised to the image, real rows first, zero rows up to `num_targets_per_image`.
  * `ConeDataset.__getitem__` (RektNet/dataset.py:34-56): image float [3,80,80] in [0,1], heat-maps [7,80,80], key points
    [7,2] = `scale_labels(...)/80` (RektNet/utils.py:104-111: ceil(int(pt) * scale)), and `prep_label`
    (RektNet/utils.py:83-97): a one-hot at (int(y), int(x)) of the ORIGINAL crop, `cv2.resize` to 80x80, 5x5
    `cv2.GaussianBlur(sigma=0)`, divided by its sum.
PARITY UNPINNED for the two cv2 calls (opencv is not in this image; RektNet/utils.py cannot be imported without it).  Their
published algorithms are restated: INTER_LINEAR resize of a float64 image (half-pixel centres, edge clamp) and, for ksize 5
with sigma <= 0, OpenCV's fixed kernel [1, 4, 6, 4, 1] / 16 applied separably with BORDER_REFLECT_101.  Because the blurred,
resized one-hot is separable, the heat-map is the outer product of two 80-vectors; both are computed in float64 like cv2.
Only up-scaling crops are generated (orig side <= 80): for down-scaling INTER_LINEAR can miss the hot pixel entirely and
the reference divides 0/0.

Pixels come from a counter-based integer hash (murmur3 finaliser), so the HIP kernel and this file agree bit for bit.
"""
import math

import numpy as np

F = np.float32
U = np.uint32
M32 = 0xFFFFFFFF


def hash32(x):
    x = np.asarray(x, dtype=np.uint64) & M32
    x ^= x >> 16
    x = (x * 0x85EBCA6B) & M32
    x ^= x >> 13
    x = (x * 0xC2B2AE35) & M32
    x ^= x >> 16
    return x.astype(np.uint32)


def urand(seed, stream, idx):
    """uniform [0,1) float32 with 24 random bits; key = (seed, stream, idx)"""
    h = hash32((np.uint64(seed) * 0x9E3779B1 + np.asarray(stream, np.uint64) * 0x85EBCA77 + np.asarray(idx, np.uint64) * 0xC2B2AE3D) & M32)
    h = hash32(h.astype(np.uint64) + 0x27D4EB2F)
    return ((h >> 8).astype(F) * F(1.0 / 16777216.0)).astype(F)


CONE_RGB = np.array([[1.0, 0.55, 0.10], [0.15, 0.35, 0.95], [0.95, 0.85, 0.15]], F)   # orange / blue / yellow


def cone_targets(seed, step, B, T, num_classes=1):
    """[B,T,5] labels: n in 1..T cones per image, rest zero rows (datasets.py:171-176 pads with zeros)."""
    t = np.zeros((B, T, 5), F)
    for b in range(B):
        key = step * 4099 + b
        n = 1 + int(urand(seed, 11, key) * F(T))
        n = min(n, T)
        for i in range(n):
            k = key * 64 + i
            w = F(0.03) + urand(seed, 12, k) * F(0.12)
            h = (w * (F(1.3) + urand(seed, 13, k) * F(0.9))).astype(F)
            cx = (w * F(0.5) + urand(seed, 14, k) * (F(1.0) - w)).astype(F)
            cy = (h * F(0.5) + urand(seed, 15, k) * (F(1.0) - h)).astype(F)
            cls = F(int(urand(seed, 16, k) * F(num_classes)))
            t[b, i] = [cls, cx, cy, w, h]
    return t


def cone_images(seed, step, targets, H, W):
    """[B,3,H,W] float32 in [0,1]: hashed noise over a vertical gradient, one striped triangle per label."""
    B, T = targets.shape[0], targets.shape[1]
    img = np.zeros((B, 3, H, W), F)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    for b in range(B):
        pix = (yy * W + xx).astype(np.uint64)
        noise = urand(seed, 21, np.uint64(step * 4099 + b) * np.uint64(H * W) + pix)
        base = (F(0.25) + F(0.35) * (yy.astype(F) / F(H)) + F(0.10) * noise).astype(F)
        for c in range(3):
            img[b, c] = (base * F(1.0 - 0.08 * c)).astype(F)
        for i in range(T):
            cls, cx, cy, w, h = targets[b, i]
            if not (w > 0 and h > 0):
                continue
            x0, x1 = F(cx - w * F(0.5)) * F(W), F(cx + w * F(0.5)) * F(W)
            y0, y1 = F(cy - h * F(0.5)) * F(H), F(cy + h * F(0.5)) * F(H)
            px, py = xx.astype(F) + F(0.5), yy.astype(F) + F(0.5)
            v = ((py - y0) / F(y1 - y0)).astype(F)                    # 0 at the apex, 1 at the base
            half = (v * F(0.5) * F(x1 - x0)).astype(F)
            mid = F((x0 + x1) * F(0.5))
            inside = (py >= y0) & (py < y1) & (np.abs(px - mid) <= half)
            stripe = (v > F(0.35)) & (v < F(0.55))
            rgb = CONE_RGB[int(cls) % 3]
            for c in range(3):
                col = np.where(stripe, F(0.95), rgb[c]).astype(F)
                img[b, c] = np.where(inside, col, img[b, c])
    return img


# canonical key-point positions of a cone in its own box (x, y in [0,1]): apex, then left/right pairs down the sides
KP = np.array([[0.5, 0.04], [0.36, 0.36], [0.64, 0.36], [0.25, 0.66], [0.75, 0.66], [0.12, 0.96], [0.88, 0.96]], F)
GAUSS5 = np.array([1.0, 4.0, 6.0, 4.0, 1.0]) / 16.0


def _resize_onehot_axis(hot, src, dst):
    """cv2.resize INTER_LINEAR of a 1-D one-hot at index `hot`, src -> dst samples, float64."""
    sc = float(src) / float(dst)
    out = np.zeros(dst, np.float64)
    for d in range(dst):
        fx = np.float32((d + 0.5) * sc - 0.5)
        sx = int(math.floor(fx))
        fx = float(np.float32(fx - np.float32(sx)))
        if sx < 0:
            sx, fx = 0, 0.0
        if sx >= src - 1:
            sx, fx = src - 1, 0.0
        s1 = min(sx + 1, src - 1)
        out[d] = (1.0 - fx) * (1.0 if sx == hot else 0.0) + fx * (1.0 if s1 == hot else 0.0)
    return out


def _blur_reflect101(v):
    n = v.shape[0]
    out = np.zeros_like(v)
    for i in range(n):
        acc = 0.0
        for k in range(-2, 3):
            j = i + k
            if j < 0:
                j = -j
            if j >= n:
                j = 2 * (n - 1) - j
            acc += GAUSS5[k + 2] * v[j]
        out[i] = acc
    return out


def cone_crops(seed, step, B, size=80):
    """(images [B,3,S,S], heatmaps [B,7,S,S], points [B,7,2]) with the ConeDataset contract."""
    img = np.zeros((B, 3, size, size), F)
    hm = np.zeros((B, 7, size, size), F)
    pts = np.zeros((B, 7, 2), F)
    yy, xx = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    for b in range(B):
        key = step * 4099 + b
        oh = 24 + int(urand(seed, 31, key) * F(57))                   # original crop 24..80 px: up-scaling only
        ow = 24 + int(urand(seed, 32, key) * F(57))
        cls = int(urand(seed, 33, key) * F(3))
        noise = urand(seed, 34, np.uint64(key) * np.uint64(size * size) + (yy * size + xx).astype(np.uint64))
        base = (F(0.30) + F(0.25) * (yy.astype(F) / F(size)) + F(0.10) * noise).astype(F)
        px, py = (xx.astype(F) + F(0.5)) / F(size), (yy.astype(F) + F(0.5)) / F(size)
        half = (py * F(0.44)).astype(F)
        inside = (np.abs(px - F(0.5)) <= half) & (py >= F(0.02)) & (py < F(0.98))
        stripe = (py > F(0.38)) & (py < F(0.58))
        for c in range(3):
            col = np.where(stripe, F(0.95), CONE_RGB[cls][c]).astype(F)
            img[b, c] = np.where(inside, col, (base * F(1.0 - 0.08 * c)).astype(F))
        hs, ws = size / oh, size / ow                                  # get_scale (utils.py:99-102), python floats
        for k in range(7):
            jx = (urand(seed, 35, key * 8 + k) - F(0.5)) * F(0.04)
            jy = (urand(seed, 36, key * 8 + k) - F(0.5)) * F(0.04)
            lx = F(min(max(float(KP[k, 0] + jx), 0.0), 0.999)) * F(ow)   # label in ORIGINAL crop pixels (float, like the csv)
            ly = F(min(max(float(KP[k, 1] + jy), 0.0), 0.999)) * F(oh)
            ix, iy = int(lx), int(ly)
            pts[b, k, 0] = F(math.ceil(ix * ws) / size)                  # scale_labels then / target size (dataset.py:43-44)
            pts[b, k, 1] = F(math.ceil(iy * hs) / size)
            vy = _blur_reflect101(_resize_onehot_axis(iy, oh, size))
            vx = _blur_reflect101(_resize_onehot_axis(ix, ow, size))
            tot = sum(vy.tolist()) * sum(vx.tolist())                    # sequential float64 sums (index order), like the kernel
            hm[b, k] = (np.outer(vy, vx) / tot).astype(F)
    return img, hm, pts
