"""Synthetic cone batches generated on the GPU, shaped like what the reference's data loaders hand to the training loops.

* `SyntheticCones`      -> `(uris, imgs [B,3,H,W] in [0,1], targets [B,T,5])`  like `ImageLabelDataset` batches
  (CVC-YOLOv3/utils/datasets.py:124-315; train.py:57-61 unpacks `(_, imgs, targets)`): class, cx, cy, w, h normalised,
  real rows first, zero rows up to `num_targets_per_image`.
* `SyntheticConeCrops`  -> `(imgs [B,3,80,80], heatmaps [B,7,80,80], points [B,7,2], names, sizes)` like `ConeDataset` batches
  (RektNet/dataset.py:34-56; train_eval.py:61-66): heat-maps follow `prep_label` (one-hot -> cv2.resize -> 5x5 GaussianBlur ->
  normalise, RektNet/utils.py:83-97), points follow `scale_labels(...) / 80` (utils.py:104-111).

Both are iterables (`len()` = batches per epoch) whose batches are pure functions of (seed, batch index): no host memory, no
PCIe traffic, reproducible across ranks (`rank` offsets the batch index so that shards differ).
"""
import torch

from .. import _lib


class SyntheticCones:
    def __init__(self, batch_size, height=416, width=416, num_targets_per_image=16, num_classes=1, batches=100, seed=0, rank=0,
                 world_size=1, device=None):
        self.B, self.H, self.W, self.T, self.C = int(batch_size), int(height), int(width), int(num_targets_per_image), int(num_classes)
        self.batches, self.seed, self.rank, self.world = int(batches), int(seed) & 0x7FFFFFFF, int(rank), int(world_size)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.dataset = range(self.batches * self.B)          # len(loader.dataset) is what validate()/train() ask for

    def __len__(self):
        return self.batches

    def batch(self, index):
        _lib.require_gpu()
        L = _lib.lib()
        imgs = torch.empty(self.B, 3, self.H, self.W, dtype=torch.float32, device=self.device)
        tg = torch.empty(self.B, self.T, 5, dtype=torch.float32, device=self.device)
        step = int(index) * self.world + self.rank
        with torch.cuda.device(self.device):
            L.check(L.synth_cone_batch(self.seed, step, self.B, self.T, self.H, self.W, self.C, imgs.data_ptr(), tg.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream), "synth_cone_batch")
        return [f"synthetic://cones/{step}/{b}" for b in range(self.B)], imgs, tg

    def __iter__(self):
        for i in range(self.batches):
            yield self.batch(i)


class SyntheticConeCrops:
    def __init__(self, batch_size, size=80, batches=100, seed=0, rank=0, world_size=1, device=None):
        if int(size) != 80:
            raise ValueError("SyntheticConeCrops: the key-point crops are 80x80 (ConeDataset's target_image_size)")
        self.B, self.S = int(batch_size), 80
        self.batches, self.seed, self.rank, self.world = int(batches), int(seed) & 0x7FFFFFFF, int(rank), int(world_size)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.dataset = range(self.batches * self.B)

    def __len__(self):
        return self.batches

    def batch(self, index):
        _lib.require_gpu()
        L = _lib.lib()
        S = self.S
        imgs = torch.empty(self.B, 3, S, S, dtype=torch.float32, device=self.device)
        hm = torch.empty(self.B, 7, S, S, dtype=torch.float32, device=self.device)
        pts = torch.empty(self.B, 7, 2, dtype=torch.float32, device=self.device)
        step = int(index) * self.world + self.rank
        with torch.cuda.device(self.device):
            L.check(L.synth_crop_batch(self.seed, step, self.B, S, imgs.data_ptr(), hm.data_ptr(), pts.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream), "synth_crop_batch")
        return imgs, hm, pts, [f"synthetic_crop_{step}_{b}" for b in range(self.B)], [(S, S, 3)] * self.B

    def __iter__(self):
        for i in range(self.batches):
            yield self.batch(i)
