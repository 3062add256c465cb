"""MI355X-native drop-in for the reference's RektNet/keypoint_net.py (KeypointNet).

`KeypointNet(num_kpt=7, image_size=(80,80), onnx_mode=False, init_weight=True)`; `forward(x)` -> `(hm, pts.view(-1,K,2))`
(hm = flat softmax over H*W, pts = soft-argmax in (x,y) order) or raw logits when `onnx_mode` — reference
keypoint_net.py:12-70.  Sub-module names (conv, bn, res1..4.{conv1,bn1,conv2,bn2,shortcut_conv,shortcut_bn}, out) match
the reference so `load_state_dict` of reference checkpoints works (train_eval.py:95-96, detect.py:36-37).

forward/backward run as one static launch plan of gfx950 kernels (NHWC bf16 or fp32): implicit-GEMM MFMA convs (7x7 stem,
dilated 3x3, 3x3, 1x1) with BN statistics in the epilogue, fused BN+ReLU, fused dual-BN + add + ReLU at the end of each
residual block, softmax/soft-argmax wavefront reductions.  The head conv is evaluated once (the reference evaluates it
twice and discards the first result, keypoint_net.py:64,68 — same values).  No CPU fallback.
"""
import os

import math

import torch
import torch.nn as nn

from .. import _lib
from ..engine import TNode, ConvSpec, BnSpec, parse_precision, ACT_RELU, BF16
from ..yolo.models import _NetPlan, FlatParamsMixin, _bump_counters, _sync_before_state_dict, _EVAL_FUSE
from .resnet import ResNet, lower_conv_bn, lower_block, lower_block_bwd


class _KeypointFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, plan, x, *params):
        plan.run_forward(x)
        ctx.model, ctx.plan = model, plan
        return plan.hm.clone(), plan.pts.clone()

    @staticmethod
    def backward(ctx, dhm, dpts):
        model, plan = ctx.model, ctx.plan
        plan.flags = (dpts is not None, dhm is not None)
        if dpts is not None:
            plan.dpts.copy_(dpts)
        if dhm is not None:
            plan.dhm_buf().copy_(dhm)
        model._run_backward(plan, None)
        return (None, None, None) + (None,) * len(model._plist)


class _KpPlan(_NetPlan):
    flags = (True, False)

    def dhm_buf(self):
        if self._dhm is None:
            self._dhm = torch.zeros_like(self.hm)
        return self._dhm

    def run_backward(self, gout):
        if self.use_graph:
            self._graphed("bwd")          # one graph per (dpts, dhm) presence combination (self.flags)
        else:
            self.run_bwd_list()


class KeypointNet(FlatParamsMixin, nn.Module):
    # torchrun on the unchanged train_eval.py: CrossRatioLoss is a batch MEAN (cross_ratio_loss.py:21-29), so each rank's loss is the mean over its
    # shard and the exchange AVERAGES the gradients -- the full-batch mean's gradient for the location term (the geometric term couples the samples
    # of a shard only).  The reference has no multi-GPU path for this network; YOLOv3's SUM is DataParallel's `losses[0].sum()`.
    _dp_average = True
    f32_logits = True                  # bf16 mode: the head conv writes fp32 logits (csrc/rektnet_head.hip; tests / A-B flip the class attribute)

    def __init__(self, num_kpt=7, image_size=(80, 80), onnx_mode=False, init_weight=True, precision=None):
        super().__init__()
        width = 16
        self.conv = nn.Conv2d(3, width, kernel_size=7, stride=1, padding=3)
        self.bn = nn.BatchNorm2d(width)
        self.relu = nn.ReLU()
        self.res1 = ResNet(width, width)
        self.res2 = ResNet(width, width * 2)
        self.res3 = ResNet(width * 2, width * 4)
        self.res4 = ResNet(width * 4, width * 8)
        self.out = nn.Conv2d(width * 8, num_kpt, kernel_size=1, stride=1, padding=0)
        if init_weight:
            self._initialize_weights()
        self.image_size = image_size
        self.num_kpt = num_kpt
        self.onnx_mode = onnx_mode
        self.precision = parse_precision(precision if precision is not None else os.environ.get("MDCV_PRECISION", "bf16"))
        self.use_graph = os.environ.get("MDCV_GRAPH", "0") == "1"
        self._plans = {}
        self.register_state_dict_pre_hook(_sync_before_state_dict)   # a pipelined optimizer step may be in flight

    def __getstate__(self):
        return self._state_without_plans()

    def load_state_dict(self, *args, **kw):
        self._param_sync()
        out = super().load_state_dict(*args, **kw)
        self._params_changed()
        return out

    def _initialize_weights(self):
        """Initial parameters of the two module kinds this network holds (what the reference's loop at keypoint_net.py:33-44 does to them):
        conv weights He-normal over the fan-OUT (std = sqrt(2 / (Cout * kh * kw)), the ReLU gain), conv biases zero; BatchNorm affine =
        identity.  One normal draw per conv in module order, so a seeded construction gives the reference's weights."""
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    kh, kw = m.kernel_size
                    m.weight.normal_(0.0, math.sqrt(2.0 / (m.out_channels * kh * kw)))
                    if m.bias is not None:
                        m.bias.zero_()
                elif isinstance(m, nn.BatchNorm2d):
                    m.weight.fill_(1.0)
                    m.bias.zero_()

    def _flatten(self):
        import weakref
        super()._flatten()
        for m in self.modules():                 # the blocks' parameters are views of THIS module's flat buffer now: a stand-alone call of a
            if isinstance(m, ResNet):            # block would re-flatten them out of it (ResNet.forward refuses while the parent is intact)
                m._flat_parent = weakref.ref(self)

    def forward(self, x):
        _lib.require_gpu(x)
        if not self._flat_ok():
            self._flatten()
        if self.training and torch.is_grad_enabled() and not self.onnx_mode:
            (x,), dp_weight = self._auto_dp_shard(x)        # torchrun on the unchanged train_eval.py: rank r's shard (parallel.enable_auto_data_parallel)
        else:
            dp_weight = None
        B, _, H, W = x.shape
        if (H, W) != tuple(self.image_size):
            raise ValueError(f"KeypointNet was built for image_size={self.image_size}, got {(H, W)}")
        infer = not self.training and not torch.is_grad_enabled() and not self.onnx_mode      # no backward can follow: one-launch conv+BN+ReLU
        key = (B, H, W, self.training, self.onnx_mode, self.precision, x.device.index, infer)
        plan = self._plan_lookup(key)
        if plan is None:
            plan = self._build_plan(x.device, B, H, W, self.training, self.onnx_mode, infer=infer)
            self._plan_store(key, plan)
        if getattr(self, "_pipe_plan", None) is not None and plan is not self._pipe_plan:
            self._param_sync()               # deferred group updates of a pipelined optimizer step are only released from ITS plan's forward list
        if self.onnx_mode:
            plan.run_forward(x)
            return plan.logits_nchw.clone()
        if torch.is_grad_enabled():
            hm, pts = _KeypointFn.apply(self, plan, x, *self._plist)
        else:
            plan.run_forward(x)
            hm, pts = plan.hm.clone(), plan.pts.clone()
        if dp_weight == 0.0:                 # this rank's DataParallel chunk was empty: it joins the exchange with exact-zero gradients
            hm, pts = hm * 0.0, pts * 0.0
        return hm, pts.view(-1, self.num_kpt, 2)

    def _build_plan(self, device, B, H, W, bn_train, logits_only, infer=False):
        plan = _KpPlan(device, self.precision, bn_train, grad_sink=self._grad_view)
        plan.owner = self
        plan.grad_offset = lambda p: self._goff[id(p)][0]
        plan.use_graph = self.use_graph
        plan.graphs_bwd = {}
        plan._dhm = None
        plan.pre = []
        L, dt = plan.L, plan.dtype
        K = self.num_kpt
        xin, holder = plan.emit_input(B, 3, H, W)
        plan.pre.append(plan.fwd.pop())
        plan.in_holder = holder
        plan.x16 = None
        if bn_train and not logits_only and plan.dtype == BF16 and self.conv.kernel_size == (7, 7) and self.conv.padding == (3, 3) \
                and self.conv.out_channels == 16:
            # second copy of the input with 16-channel rows: the stem's weight gradient runs the LDS-ring kernel on it (csrc/wgrad_stream.hip)
            x16 = plan.new_act(B, H, W, 16)
            plan.x16 = x16

            def convert16(stream, x16=x16):
                return L.nchw_to_nhwc(dt, holder["src"].data_ptr(), x16.ptr, B, 3, H, W, x16.ldc, x16.C, stream)
            convert16.__name__ = "nchw_to_nhwc"
            plan.pre.append((convert16, ()))
        plan.targets = None
        nbt = []
        recs = []

        one_launch = infer and _EVAL_FUSE

        def conv_bn(conv, bn, xnode, relu_into=None):
            return lower_conv_bn(plan, conv, bn, xnode, B, H, W, bn_train, nbt, one_launch, relu_into=relu_into)

        a = TNode(plan.new_act(B, H, W, 16), name="stem")
        cs0, bs0, y0 = conv_bn(self.conv, self.bn, xin, relu_into=a.act)
        if y0 is not None:
            plan.emit_bn_act_fwd(y0, bs0, a.act, ACT_RELU, 0.0)
        recs.append(("stem", cs0, bs0, xin, y0, a))
        for blk in (self.res1, self.res2, self.res3, self.res4):
            rec, a = lower_block(plan, blk, a, B, H, W, bn_train, nbt, one_launch)
            recs.append(rec)
        csh = ConvSpec(plan, self.out.weight, self.out.bias, 1, 0, 1, cin_pad=a.act.C)
        plan.emit_pack(csh, need_dgrad=True)
        lg = TNode(plan.new_act(B, H, W, K), name="logits")
        # bf16 mode: the logits stay fp32 (csrc/rektnet_head.hip head1x1_f32_kernel has the reason); the bf16 buffer above only gives the backward its shape
        f32_logits = self.f32_logits and plan.dtype == BF16 and a.act.C % 32 == 0 and a.act.C <= 1024 and K <= 8 and self.out.kernel_size == (1, 1)
        if f32_logits:
            plan.lg32 = torch.zeros(B * H * W, 8, dtype=torch.float32, device=device)
            w_out, b_out = self.out.weight, self.out.bias
            plan.call(plan.fwd, L.head1x1_f32, a.act.ptr, a.act.ldc, w_out.data_ptr(), b_out.data_ptr() if b_out is not None else None,
                      plan.lg32.data_ptr(), B * H * W, a.act.C, K)
            lg_dt, lg_ptr, lg_ldc = _lib.F32, plan.lg32.data_ptr(), 8
        else:
            plan.emit_conv_fwd(csh, a.act, lg.act)
            lg_dt, lg_ptr, lg_ldc = dt, lg.act.ptr, lg.act.ldc
        if bn_train and nbt:
            plan.call(plan.fwd, _bump_counters, nbt)
        plan.finish_pack(0)
        if logits_only:
            plan.logits_nchw = torch.empty(B, K, H, W, dtype=torch.float32, device=device)
            plan.call(plan.fwd, L.nhwc_to_nchw, lg_dt, lg_ptr, lg_ldc, plan.logits_nchw.data_ptr(), B, K, H, W)
            plan.has_bwd = False
            return plan
        plan.hm = torch.empty(B, K, H, W, dtype=torch.float32, device=device)
        plan.pts = torch.empty(B, K, 2, dtype=torch.float32, device=device)
        plan.dpts = torch.zeros(B, K, 2, dtype=torch.float32, device=device)
        plan.sdot = torch.zeros(B * K, dtype=torch.float32, device=device)
        plan.call(plan.fwd, L.softargmax_fwd, lg_dt, lg_ptr, lg_ldc, B, K, H, W, plan.hm.data_ptr(), plan.pts.data_ptr())
        plan.has_bwd = not infer
        if infer:                          # inference plan: forward list only (the raw conv outputs a backward would need do not exist)
            return plan

        # ---------------- backward ----------------
        dlg, add = plan.grad_target(lg)

        def head_bwd(stream):
            use_pts, use_hm = plan.flags
            return L.softargmax_bwd(dt, plan.hm.data_ptr(), plan.pts.data_ptr(), plan.dpts.data_ptr() if use_pts else None,
                                    plan.dhm_buf().data_ptr() if use_hm else None, plan.sdot.data_ptr(), B, K, H, W, dlg.ptr, dlg.ldc, stream)
        head_bwd.__name__ = "softargmax_bwd"
        plan.bwd.append((head_bwd, ()))
        plan.emit_bias_grad(csh, dlg)
        plan.emit_conv_bwd(csh, a, lg.act, dlg)
        for r in reversed(recs):
            plan.mark_ready()
            if r[0] == "block":
                lower_block_bwd(plan, r)
            else:
                _, cs0, bs0, xin_, y0, a0 = r
                dy0 = plan.emit_bn_act_bwd(a0.grad, y0, bs0, ACT_RELU, 0.0)
                plan.emit_conv_bwd(cs0, xin_, y0, dy0, x_wgrad=plan.x16)
                plan.emit_bias_grad(cs0, None, zero_only=True)
        plan.mark_ready()
        return plan
