"""GPU drop-in for the reference's CVC-YOLOv3/validate.py `validate` (validate.py:61-182).

Same keyword signature and return tuple `(mean_mAP, mean_R, mean_P, seconds_per_image)`.  The per-image Python loop
(conf filter -> NMS -> IoU matching -> AP, validate.py:80-141) runs on the device for the whole batch
(`postprocess.detect_postprocess`); per-image statistics stay in HBM until the single read-back at the end.
The image-drawing / upload branch (validate.py:139-160, PIL + gsutil) is visualisation and is not part of this path.
"""
import time

import torch

from .postprocess import detect_postprocess


def validate(*, dataloader, model, device, step=-1, bbox_all=False, debug_mode=False):
    with torch.no_grad():
        t_start = time.time()
        conf_thres, nms_thres, iou_thres = model.get_threshs()
        width, height = model.img_size()
        model.eval()
        n_images = len(dataloader.dataset)
        stats = []
        for _uris, imgs, targets in dataloader:
            imgs = imgs.to(device, non_blocking=True)
            targets = targets.to(device, non_blocking=True)
            output = model(imgs)
            stats.append(detect_postprocess(output, targets, conf_thres, nms_thres, iou_thres, width, height).stats)
        if stats:
            s = torch.cat(stats, 0)
            valid = s[:, 3] > 0
            n = int(valid.sum().item())
            means = (s[valid, :3].sum(0) / max(n, 1)).tolist() if n else [float("nan")] * 3
        else:
            means = [float("nan")] * 3
        torch.cuda.synchronize()
        dt = time.time() - t_start
        mean_mAP, mean_R, mean_P = means
        print('mAP: {0:5.2%}, Recall: {1:5.2%}, Precision: {2:5.2%}'.format(mean_mAP, mean_R, mean_P))
        return mean_mAP, mean_R, mean_P, dt / (n_images + 1e-12)
