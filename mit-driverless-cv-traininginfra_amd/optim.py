"""Fused flat-buffer optimizers (HIP) with the torch.optim interface.

`FusedAdam(model, lr=...)` / `FusedSGD(model, lr=..., momentum=...)` update every parameter of a Darknet / KeypointNet in ONE
launch over the model's flat fp32 parameter buffer (engine side: FlatParamsMixin), instead of the 222 / 54 per-tensor
updates torch.optim performs for the reference (CVC-YOLOv3/train.py:180-187,72 ; RektNet/train_eval.py:263,72).
Update rules, defaults and `param_groups[0]["lr"]` handling (LR schedulers) follow torch.optim.Adam / SGD.
"""
import torch

from . import _lib


class _FlatOptimizer(torch.optim.Optimizer):
    def __init__(self, model, defaults):
        if not hasattr(model, "flat_parameters"):
            raise TypeError("Fused optimizers take the model (Darknet / KeypointNet), not a parameter list")
        self.model = model
        super().__init__(list(model.parameters()), defaults)
        self._step = 0
        self._state_bufs = None

    def _flat(self, nstate):
        pflat, gflat = self.model.flat_parameters()
        if self._state_bufs is None or self._state_bufs[0].numel() != pflat.numel() or self._state_bufs[0].device != pflat.device:
            self._state_bufs = [torch.zeros_like(pflat) for _ in range(nstate)]
        # gradients normally ARE views of gflat (written in place by the backward plan); fold in anything that is not
        for p in self.model._plist:
            v = self.model._grad_view(p)
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
        return pflat, gflat

    def state_dict(self):
        return {"step": self._step, "state": [b.clone() for b in (self._state_bufs or [])], "param_groups": [
            {k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self._step = sd["step"]
        if sd["state"]:
            self._state_bufs = [b.clone() for b in sd["state"]]
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)


class FusedAdam(_FlatOptimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(model, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        _lib.require_gpu()
        L = _lib.lib()
        g = self.param_groups[0]
        pflat, gflat = self._flat(2)
        m, v = self._state_bufs
        self._step += 1
        L.check(L.adam_step(pflat.data_ptr(), gflat.data_ptr(), m.data_ptr(), v.data_ptr(), pflat.numel(), self._step, float(g["lr"]),
                            float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), float(grad_scale),
                            torch.cuda.current_stream().cuda_stream), "adam_step")


class FusedSGD(_FlatOptimizer):
    def __init__(self, model, lr=1e-3, momentum=0.0, weight_decay=0.0):
        super().__init__(model, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        _lib.require_gpu()
        L = _lib.lib()
        g = self.param_groups[0]
        pflat, gflat = self._flat(1)
        self._step += 1
        L.check(L.sgd_step(pflat.data_ptr(), gflat.data_ptr(), self._state_bufs[0].data_ptr(), pflat.numel(), self._step, float(g["lr"]),
                           float(g["momentum"]), float(g["weight_decay"]), float(grad_scale), torch.cuda.current_stream().cuda_stream), "sgd_step")
