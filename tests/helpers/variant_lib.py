"""Test-local selection of kernel variants (see the class docstring); shared by test_gpu_kernels.py and test_gpu_redzone.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mdcv import _lib  # noqa: E402


class VariantLib:
    """The library seen under ONE variant code per entry-point family.  The C ABI keeps no tuning state (round 5): a variant is requested per call
    through the dtype argument (include/mdcv_hip.h MDCV_TUNED, csrc/tune.h).  This test-local view keeps the tests' shape -- select a variant,
    make calls, select the default again -- with the selection living in THIS object: `conv2d_set_variant` / `conv2d_wgrad_set_variant` take the
    codes of csrc/tune.h as the removed hooks numbered them (forced tile configurations 0..11 are 1..12 in the per-call encoding, where 0 means
    "defaults"; a code that names a default selects the defaults)."""
    CONV = ("conv2d", "conv2d_xstats", "conv2d_affine_act", "conv2d_dgrad_bnsums_rows", "conv2d_dgrad_bnsums", "conv2d_dgrad_s2_form_ok",
            "conv2d_stats_rows_geom")
    WGRAD = ("conv2d_wgrad", "conv2d_wgrad_splits", "conv2d_wgrad_splits_geom")
    CONV_DEFAULTS = (-1, 99, -4, -7, -14, -19, -22, -23, -26, -28, -32, -60, -64, -200, 15, 17, 19, 21, 93)

    def __init__(self):
        self._L = _lib.lib()
        self.conv = self.wgrad = 0

    def conv2d_set_variant(self, v):
        self.conv = 0 if v in self.CONV_DEFAULTS else (v + 1 if 0 <= v <= 11 else (v + 1 if 100 <= v <= 111 else v))
        return 0

    def conv2d_wgrad_set_variant(self, v):
        self.wgrad = 0 if v in (0, 30002) else v
        return 0

    def __getattr__(self, name):
        fn = getattr(self._L, name)
        code = self.conv if name in self.CONV else (self.wgrad if name in self.WGRAD else 0)
        if not code:
            return fn
        return lambda dt, *a: fn(_lib.tuned(dt, code), *a)
