"""MI355X-native drop-in for the reference's RektNet/cross_ratio_loss.py (CrossRatioLoss).

`CrossRatioLoss(loss_type, include_geo, geo_loss_gamma_horz, geo_loss_gamma_vert)`; `forward(heatmap, points, target_hm,
target_points)` -> `(location_loss, geo_loss, location_loss + geo_loss)` — reference cross_ratio_loss.py:8-63:
location = l2 / l1 on the soft-argmax points or l2 on the heat-map, batch mean; geo = six `1 - U V^T` terms, each the mean
of a [B,B] ALL-PAIRS matrix (tensordot over the coordinate axis), weighted gamma_horz/2 and gamma_vert/4.
Value and gradients come from one wavefront-reduction HIP kernel (csrc/rektnet_head.hip).  No CPU fallback.
"""
import torch
from torch import nn

from .. import _lib

_TYPES = {"l2_softargmax": 0, "l2_sm": 0, "l2_heatmap": 1, "l2_hm": 1, "l1_softargmax": 2, "l1_sm": 2}


class _CrossRatioFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, heatmap, points, target_hm, target_points):
        L = _lib.lib()
        dev = points.device
        st = torch.cuda.current_stream().cuda_stream
        lt = _TYPES[mod.loss_type]
        pts = points.detach().float().contiguous()
        tpts = target_points.detach().to(device=dev, dtype=torch.float32).contiguous()
        B = pts.shape[0]
        hm = thm = None
        H = W = 0
        if lt == 1:
            hm = heatmap.detach().float().contiguous()
            thm = target_hm.detach().to(device=dev, dtype=torch.float32).contiguous()
            H, W = hm.shape[2], hm.shape[3]
        # (the double accumulator is only touched by the heat-map loss; three separate 0-dim outputs instead of one [3] tensor the caller
        #  indexes: indexing costs a zeros + copy launch per use in autograd's select_backward, ~10 tiny launches per training step)
        acc = torch.zeros(1, dtype=torch.float64, device=dev) if lt == 1 else None
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        L.check(L.cross_ratio_loss(hm.data_ptr() if hm is not None else None, pts.data_ptr(), thm.data_ptr() if thm is not None else None,
                                   tpts.data_ptr(), B, H, W, lt, int(bool(mod.include_geo)), float(mod.geo_loss_gamma_horz),
                                   float(mod.geo_loss_gamma_vert), acc.data_ptr() if acc is not None else None, None, out3.data_ptr(), None, None,
                                   st), "cross_ratio_loss")
        ctx.saved = (mod, hm, pts, thm, tpts, lt, acc, H, W)
        ctx.set_materialize_grads(False)
        return out3[0], out3[1], out3[2]

    @staticmethod
    def backward(ctx, g_loc, g_geo, g_tot):
        L = _lib.lib()
        mod, hm, pts, thm, tpts, lt, acc, H, W = ctx.saved
        st = torch.cuda.current_stream().cuda_stream
        B = pts.shape[0]
        # upstream gradients of the (location, geo) parts: total = location + geo
        if g_loc is None and g_geo is None and g_tot is not None:
            gs = g_tot.detach().float().reshape(1).expand(2).contiguous()       # the training loops backpropagate `total` only: one launch
        else:
            z = torch.zeros((), dtype=torch.float32, device=pts.device)
            f = lambda g: z if g is None else g.detach().float().reshape(())    # noqa: E731
            gs = torch.stack((f(g_loc) + f(g_tot), f(g_geo) + f(g_tot))).contiguous()
        dpts = torch.empty_like(pts)
        dhm = torch.empty_like(hm) if lt == 1 else None
        scratch = torch.empty(3, dtype=torch.float32, device=pts.device)
        L.check(L.cross_ratio_loss(hm.data_ptr() if hm is not None else None, pts.data_ptr(), thm.data_ptr() if thm is not None else None,
                                   tpts.data_ptr(), B, H, W, lt, int(bool(mod.include_geo)), float(mod.geo_loss_gamma_horz),
                                   float(mod.geo_loss_gamma_vert), acc.data_ptr() if acc is not None else None, gs.data_ptr(), scratch.data_ptr(),
                                   dpts.data_ptr(), dhm.data_ptr() if dhm is not None else None, st), "cross_ratio_loss(bwd)")
        return None, dhm, dpts, None, None


class CrossRatioLoss(nn.Module):
    def __init__(self, loss_type, include_geo, geo_loss_gamma_horz, geo_loss_gamma_vert):
        super().__init__()
        self.loss_type = loss_type
        self.include_geo = include_geo
        self.geo_loss_gamma_vert = geo_loss_gamma_vert
        self.geo_loss_gamma_horz = geo_loss_gamma_horz
        print(f"Including geometric loss: {include_geo}")
        print(f"Loss type: {loss_type}")

    def forward(self, heatmap, points, target_hm, target_points):
        if self.loss_type not in _TYPES:
            print("Did not recognize loss function selection!")
            raise NameError("name 'sys' is not defined")       # what the reference does here (cross_ratio_loss.py:31-32)
        if target_points is not None and points.shape[0] != target_points.shape[0]:
            # torchrun on the unchanged train_eval.py (parallel.enable_auto_data_parallel): KeypointNet.forward kept rank r's shard of the batch; the
            # labels the script hands over (train_eval.py:72) are still the whole batch
            from ..parallel import auto_state, auto_slice
            if auto_state() is not None:
                lo, hi, _ = auto_slice(target_points.shape[0])          # the same share KeypointNet.forward took (DataParallel's chunking)
                if hi - lo == points.shape[0]:
                    target_points = target_points[lo:hi]
                    if target_hm is not None:
                        target_hm = target_hm[lo:hi]
        if tuple(points.shape[1:]) != (7, 2) or tuple(target_points.shape) != tuple(points.shape):
            # the kernel's row stride is 14 floats and the geometric terms use key points 0..6 (cross_ratio_loss.py:36-57 hard-codes them too)
            raise ValueError(f"CrossRatioLoss (HIP) takes points / target_points of shape [B, 7, 2]; got {tuple(points.shape)} and "
                             f"{tuple(target_points.shape)}")
        _lib.require_gpu(points)
        location_loss, geo, total = _CrossRatioFn.apply(self, heatmap, points, target_hm, target_points)
        geo_loss = geo if self.include_geo else torch.tensor(0)         # int64 CPU zero, like the reference (:59)
        return location_loss, geo_loss, total
