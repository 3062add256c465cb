"""MI355X-native hot path of cv-core/MIT-Driverless-CV-TrainingInfra (CVC-YOLOv3 + RektNet training step).

Import as ``mdcv`` (the repo-root shim) because this directory name is not a Python identifier:

    from mdcv.yolo.models import Darknet, YOLOLayer
    from mdcv.yolo.utils.utils import build_targets, bbox_iou
    from mdcv.rektnet.keypoint_net import KeypointNet
    from mdcv.rektnet.cross_ratio_loss import CrossRatioLoss
"""
__version__ = "0.1.0"
