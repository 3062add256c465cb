"""Reference-side binding for RektNet: with THIS directory in front of the reference's RektNet/ on sys.path,
`from resnet import ResNet` (RektNet/train_eval.py:24-25, detect.py, pt_to_onnx.py) binds the MI355X-native class.  The directory holds
keypoint_net.py, resnet.py and cross_ratio_loss.py only: `from utils import Logger, ...` and `from dataset import ConeDataset`
(train_eval.py:26-28) keep resolving to the reference's own files."""
import os
import sys

_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # where the `mdcv` import alias lives
if _REPO not in sys.path:
    sys.path.append(_REPO)             # appended, not prepended: nothing of the reference's own tree is shadowed

from mdcv.parallel import enable_auto_data_parallel  # noqa: E402  (see keypoint_net.py: torchrun on the unchanged train_eval.py)
enable_auto_data_parallel()

from mdcv.rektnet.resnet import ResNet  # noqa: E402,F401

__all__ = ["ResNet"]
