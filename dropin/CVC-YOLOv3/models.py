"""Reference-side binding for CVC-YOLOv3: put THIS directory in front of the reference's CVC-YOLOv3/ on sys.path
(`PYTHONPATH=<repo>/dropin/CVC-YOLOv3 python train.py ...`) and `from models import Darknet` (train.py:19, validate.py:12,
detect.py) binds the MI355X-native classes.  The directory holds this one module and nothing else -- no `utils` package, no
`validate.py` -- so `from utils.datasets import ImageLabelDataset`, `from utils.utils import model_info, Logger, ...`,
`from utils.nms import nms` and `import validate` (train.py:20-22) keep resolving to the reference's own files.

`use_hip_postprocessing()` optionally swaps the reference's per-image NMS loop for the batched HIP one, in whatever `utils.nms`
module is importable (call it before `import validate`).
"""
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

from mdcv.yolo.models import Darknet, YOLOLayer, EmptyLayer, create_modules, vanilla_anchor_list  # noqa: E402,F401
from mdcv.yolo.utils.parse_config import parse_model_config  # noqa: E402,F401  (reference models.py:9 imports it into this namespace)

__all__ = ["Darknet", "YOLOLayer", "EmptyLayer", "create_modules", "vanilla_anchor_list", "parse_model_config", "use_hip_postprocessing"]


def use_hip_postprocessing():
    """Rebind `utils.nms.nms` (reference utils/nms.py:4, called per image from validate.py:96 and detect.py) to the HIP drop-in with the
    same signature.  Returns the names that were patched.  Nothing is patched unless this is called."""
    import importlib
    patched = []
    try:
        ref_nms = importlib.import_module("utils.nms")
    except ImportError:
        return patched
    from mdcv.yolo.utils.nms import nms as hip_nms
    if getattr(ref_nms, "nms", None) is not hip_nms:
        ref_nms.nms = hip_nms
        patched.append("utils.nms.nms")
    v = sys.modules.get("validate")
    if v is not None and hasattr(v, "nms"):            # `from utils.nms import nms` already ran there
        v.nms = hip_nms
        patched.append("validate.nms")
    return patched
