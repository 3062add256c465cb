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
        if self._state_bufs is None:
            self._state_bufs = [torch.zeros_like(pflat) for _ in range(nstate)]
        elif self._state_bufs[0].numel() != pflat.numel() or len(self._state_bufs) != nstate:
            # moments of another flat layout (a checkpoint written before parameters were padded to 16-byte boundaries, another model): zeroing them
            # while keeping `step` would resume with wrong bias correction and no sign of it
            raise ValueError(f"optimizer state holds {len(self._state_bufs)} buffer(s) of {self._state_bufs[0].numel()} elements, the model's flat "
                             f"parameter buffer has {pflat.numel()} ({nstate} expected): the state belongs to another parameter layout")
        elif self._state_bufs[0].device != pflat.device:
            self._state_bufs = [b.to(pflat.device) for b in self._state_bufs]
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
        if sd["state"]:
            n = self.model.flat_parameters()[0].numel()
            bad = [int(b.numel()) for b in sd["state"] if b.numel() != n]
            if bad:
                raise ValueError(f"optimizer checkpoint holds flat state of {bad[0]} elements, this model's flat parameter buffer has {n}: it was "
                                 "written for another parameter layout (e.g. before parameters were padded to 16-byte boundaries)")
            self._state_bufs = [b.clone() for b in sd["state"]]
        self._step = sd["step"]
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)


class FusedAdam(_FlatOptimizer):
    """`pipeline=True`: the update is cut into a few forward-ordered groups of whole layers, each followed by the re-pack of
    that group's conv operands, all on a parameter stream; `step()` returns at once and the NEXT forward of the training plan
    waits group by group (engine.Plan.ensure_param_groups).  The ~0.7 ms of HBM-bound optimizer + pack work of a YOLOv3 step then
    runs under the MFMA-bound first layers of the following forward instead of in front of it.  Same arithmetic, same order of
    operations per element; only the stream differs.  Readers of the parameters go through `model.flat_parameters()` /
    `synchronize()`, which order the current stream behind the update."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, pipeline=False):
        super().__init__(model, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.pipeline = pipeline
        self._pstream = None

    def synchronize(self):
        """current stream waits for a pipelined update still in flight"""
        self.model._param_sync()

    def zero_grad(self, set_to_none=True):
        if not set_to_none:
            self.model._param_sync()       # the in-flight update still reads the gradient buffer
        super().zero_grad(set_to_none=set_to_none)

    def _pipelined(self, L, plan, groups, pflat, gflat, m, v, g, grad_scale):
        model = self.model
        cur = torch.cuda.current_stream()
        if self._pstream is None:
            from .engine import checked_stream
            self._pstream = checked_stream(pflat.device, [cur], "param")      # (checked to run beside the current stream)
        ps = self._pstream
        ps.wait_stream(cur)                                  # gradients final (backward, side stream, all-reduce all joined `cur`)
        for pl in model._plans.values():
            pl.owner = model
        pp, gp, mp, vp = pflat.data_ptr(), gflat.data_ptr(), m.data_ptr(), v.data_ptr()
        tab = plan.pack_table.data_ptr()
        hyp = (self._step, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), float(grad_scale))

        def make(k, lo, hi, first, nl):
            def launch(after_stream):
                if after_stream is not None:                 # start only when `after_stream` has reached this point on the GPU
                    gate = torch.cuda.Event()
                    gate.record(after_stream)
                    ps.wait_event(gate)
                with torch.cuda.stream(ps):
                    s = ps.cuda_stream
                    L.check(L.adam_step(pp + 4 * lo, gp + 4 * lo, mp + 4 * lo, vp + 4 * lo, hi - lo, *hyp, s), "adam_step")
                    L.check(L.pack_weights_batched(plan.dtype, tab + 72 * first, nl, plan.pack_eq_taps, s), "pack_weights_batched")
                    ev = torch.cuda.Event()
                    ev.record(ps)
                    plan._group_events[k] = ev
            return launch
        for k, (lo, hi, first, nl) in enumerate(groups):
            fn = make(k, lo, hi, first, nl)
            if k < 2:
                fn(None)                                     # the first layers' (small) groups at once
            else:
                plan._pending_updates[k] = fn                # the rest from the forward list, two groups ahead of their use
        plan._packed_ahead = True
        model._params_changed()                                     # every other plan's packed operands are stale from here on
        plan._packed_version = model._param_epoch                   # (load_weights / load_state_dict bump the epoch too: this pack is then redone)
        plan._packed_tversion = model._param_versions()             # (in-place edits of a parameter by user code bump ITS version counter)
        model._pipe_plan = plan

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        _lib.require_gpu()
        L = _lib.lib()
        g = self.param_groups[0]
        pflat, gflat = self._flat(2)
        m, v = self._state_bufs
        self._step += 1
        plan = getattr(self.model, "_last_train_plan", None)
        if self.pipeline and plan is not None and not plan.use_graph and pflat.is_cuda and getattr(plan, "pack_table", None) is not None:
            groups = plan.ensure_param_groups(pflat)
            if groups:
                self._pipelined(L, plan, groups, pflat, gflat, m, v, g, grad_scale)
                return
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
