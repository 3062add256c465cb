"""Reference-side binding for RektNet: with THIS directory in front of the reference's RektNet/ on sys.path,
`from keypoint_net import KeypointNet` (RektNet/train_eval.py:24-25, detect.py, pt_to_onnx.py) binds the MI355X-native class.  The directory holds
keypoint_net.py, resnet.py and cross_ratio_loss.py only: `from utils import Logger, ...` and `from dataset import ConeDataset`
(train_eval.py:26-28) keep resolving to the reference's own files."""
import os
import sys

_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # where the `mdcv` import alias lives
if _REPO not in sys.path:
    sys.path.append(_REPO)             # appended, not prepended: nothing of the reference's own tree is shadowed

# Under `python -m torch.distributed.run --nproc-per-node N <script> ...` this import is the first thing the script does with the model code, and
# it happens before any torch.cuda call: pin this rank to its GPU (so the script's `torch.cuda.device_count() > 1` -> nn.DataParallel branch,
# CVC-YOLOv3/train.py:193-195, is not taken), join the process group, and let the models shard batches / all-reduce gradients themselves
# (mdcv/parallel.py: enable_auto_data_parallel).  A plain single-process run is untouched.
from mdcv.parallel import enable_auto_data_parallel  # noqa: E402
enable_auto_data_parallel()

from mdcv.rektnet.keypoint_net import KeypointNet  # noqa: E402,F401

__all__ = ["KeypointNet"]
