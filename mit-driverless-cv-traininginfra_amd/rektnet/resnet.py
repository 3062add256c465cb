"""MI355X-native drop-in for the reference's RektNet/resnet.py (ResNet residual block, resnet.py:8-27).

    out = relu( shortcut_bn(shortcut_conv(x)) + bn2(conv2( relu(bn1(conv1(x))) )) )
    conv1: 3x3 dilation 2 padding 2 ; conv2: 3x3 padding 1 ; shortcut_conv: 1x1 ; every conv has a bias.

The module keeps the reference's attribute names (state_dict compatibility).  Inside KeypointNet the four blocks are lowered into
the network's one launch plan (`lower_block` / `lower_block_bwd` below are what keypoint_net.py calls); called on its own,
`ResNet.forward(x)` lowers the single block into its own plan — NCHW fp32 in, NCHW fp32 out, one autograd node whose backward runs
the mirrored list (dx, parameter gradients) — with the same kernels: MFMA convs with BatchNorm statistics in the epilogue, fused
BN+ReLU, fused dual-BN + add + ReLU.  No CPU fallback.
"""
import os

import torch
import torch.nn as nn

from .. import _lib
from ..engine import TNode, ConvSpec, BnSpec, parse_precision, ACT_RELU
from ..yolo.models import _NetPlan, FlatParamsMixin, _bump_counters, _sync_before_state_dict


def lower_conv_bn(plan, conv, bn, xnode, B, H, W, bn_train, nbt, one_launch=False, relu_into=None):
    """conv -> BatchNorm statistics (train) / running-statistics coefficients (eval).  Returns (ConvSpec, BnSpec, raw output Act);
    relu_into: the activation buffer of a conv -> BN -> ReLU chain, which inference plans produce in the conv's own launch (raw = None)."""
    cs = ConvSpec(plan, conv.weight, conv.bias, conv.stride[0], conv.padding[0], conv.dilation[0], cin_pad=xnode.act.C)
    plan.emit_pack(cs, need_dgrad=xnode.needs_grad)
    bs = BnSpec(plan, bn)
    if one_launch and relu_into is not None:
        plan.emit_conv_bn_act_eval(cs, bs, xnode.act, relu_into, ACT_RELU, 0.0)
        return cs, bs, None
    y = plan.new_act(B, H, W, conv.out_channels)
    if bn_train:
        rows = plan.stats_rows(cs, xnode.act, y)
        partial = plan.f32(rows * 2 * y.C, zero=False)
        plan.emit_conv_fwd(cs, xnode.act, y, partial)
        plan.emit_bn_stats(bs, y, partial, rows)
        nbt.append(bn.num_batches_tracked)
    else:
        plan.emit_conv_fwd(cs, xnode.act, y)
        plan.emit_bn_eval(bs)
    return cs, bs, y


def lower_block(plan, blk, x, B, H, W, bn_train, nbt, one_launch=False):
    """Forward launches of one residual block on TNode x (resnet.py:22-27).  Returns (record for lower_block_bwd, output TNode)."""
    mid = TNode(plan.new_act(B, H, W, blk.conv1.out_channels), name="mid")
    cs1, bs1, y1 = lower_conv_bn(plan, blk.conv1, blk.bn1, x, B, H, W, bn_train, nbt, one_launch, relu_into=mid.act)
    if y1 is not None:
        plan.emit_bn_act_fwd(y1, bs1, mid.act, ACT_RELU, 0.0)
    cs2, bs2, y2 = lower_conv_bn(plan, blk.conv2, blk.bn2, mid, B, H, W, bn_train, nbt)
    css, bss, ys = lower_conv_bn(plan, blk.shortcut_conv, blk.shortcut_bn, x, B, H, W, bn_train, nbt)
    out = TNode(plan.new_act(B, H, W, blk.conv2.out_channels), name="blk")
    plan.emit_bn_act_fwd(y2, bs2, out.act, ACT_RELU, 0.0, y2=ys, bs2=bss)
    return ("block", x, cs1, bs1, y1, mid, cs2, bs2, y2, css, bss, ys, out), out


def lower_block_bwd(plan, rec):
    """Backward launches of a block whose output gradient `out.grad` exists: dual-BN backward, the three convs' weight gradients,
    the data gradients into mid and (if it needs one) x; the conv biases sit in front of a BatchNorm, their gradient is exactly zero."""
    _, x, cs1, bs1, y1, mid, cs2, bs2, y2, css, bss, ys, out = rec
    dy2, dys = plan.emit_bn_act_bwd(out.grad, y2, bs2, ACT_RELU, 0.0, y2=ys, bs2=bss)
    plan.emit_conv_bwd(cs2, mid, y2, dy2)
    plan.emit_conv_bwd(css, x, ys, dys)
    dy1 = plan.emit_bn_act_bwd(mid.grad, y1, bs1, ACT_RELU, 0.0)
    plan.emit_conv_bwd(cs1, x, y1, dy1)
    for cs in (cs1, cs2, css):
        plan.emit_bias_grad(cs, None, zero_only=True)


class _ResNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, plan, x, *params):
        plan.run_forward(x)
        ctx.model, ctx.plan = model, plan
        return plan.out_nchw.clone()

    @staticmethod
    def backward(ctx, dout):
        model, plan = ctx.model, ctx.plan
        plan.dout_holder["src"] = dout.float().contiguous()
        model._run_backward(plan, None)
        dx = plan.dx_nchw.clone() if plan.dx_nchw is not None else None
        return (None, None, dx) + (None,) * len(model._plist)


class _BlockPlan(_NetPlan):
    def run_backward(self, gout):
        self.run_bwd_list()

class ResNet(FlatParamsMixin, nn.Module):
    def __init__(self, in_channels, out_channels, precision=None):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=2, dilation=2)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.relu1 = nn.ReLU()
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.relu2 = nn.ReLU()
        self.shortcut_conv = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1)
        self.shortcut_bn = nn.BatchNorm2d(out_channels)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.precision = parse_precision(precision if precision is not None else os.environ.get("MDCV_PRECISION", "bf16"))
        self._plans = {}
        self.register_state_dict_pre_hook(_sync_before_state_dict)

    def __getstate__(self):
        return self._state_without_plans()

    def forward(self, x):
        """Stand-alone block (inside KeypointNet the blocks run as part of the network's plan and this is not called)."""
        _lib.require_gpu(x)
        parent = getattr(self, "_flat_parent", None)
        if parent is not None and parent() is not None and parent()._flat_ok():
            raise RuntimeError("this ResNet block's parameters are views of its KeypointNet's flat buffer: calling the block on its own would "
                               "re-flatten them out of the parent (dropping the parent's launch plans). Run the block through the network, or "
                               "copy.deepcopy() it first.")
        if not self.training and torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            # the reference block is differentiable in eval mode; this build has no BatchNorm-eval backward (not on the training hot path):
            # refuse instead of returning a tensor without a graph (silently zero / None gradients)
            raise NotImplementedError("ResNet block in eval mode with gradients enabled: wrap the call in torch.no_grad() (inference) or "
                                      "switch to .train() -- there is no BatchNorm-eval backward on the HIP path")
        if not self._flat_ok():
            self._flatten()
        B, C, H, W = x.shape
        if C != self.in_channels:
            raise ValueError(f"ResNet block expects {self.in_channels} input channels, got {C}")
        train_graph = self.training and torch.is_grad_enabled()
        need_dx = train_graph and x.requires_grad
        key = (B, H, W, self.training, train_graph, need_dx, self.precision, x.device.index)
        plan = self._plan_lookup(key)
        if plan is None:
            plan = self._build_plan(x.device, B, H, W, self.training, train_graph, need_dx)
            self._plan_store(key, plan)
        if train_graph:
            return _ResNetFn.apply(self, plan, x, *self._plist)
        # eval mode / no_grad: forward only, returned without a graph (a BatchNorm-eval backward is not on the training hot path)
        plan.run_forward(x)
        return plan.out_nchw.clone()

    def _build_plan(self, device, B, H, W, bn_train, with_bwd, need_dx):
        plan = _BlockPlan(device, self.precision, bn_train, grad_sink=self._grad_view)
        plan.owner = self
        plan.grad_offset = lambda p: self._goff[id(p)][0]
        plan.use_graph = False
        plan.pre = []
        L, dt = plan.L, plan.dtype
        xin, holder = plan.emit_input(B, self.in_channels, H, W)
        plan.pre.append(plan.fwd.pop())
        plan.in_holder = holder
        plan.targets = None
        xin.needs_grad = need_dx
        nbt = []
        rec, out = lower_block(plan, self, xin, B, H, W, bn_train, nbt)
        if bn_train and nbt:
            plan.call(plan.fwd, _bump_counters, nbt)
        plan.finish_pack(0)
        Co = self.out_channels
        plan.out_nchw = torch.empty(B, Co, H, W, dtype=torch.float32, device=device)
        plan.call(plan.fwd, L.nhwc_to_nchw, dt, out.act.ptr, out.act.ldc, plan.out_nchw.data_ptr(), B, Co, H, W)
        plan.has_bwd = with_bwd
        plan.dx_nchw = None
        if not with_bwd:
            return plan
        g, _ = plan.grad_target(out)
        plan.dout_holder = {"src": None}

        def dout_in(stream, g=g):
            return L.nchw_to_nhwc(dt, plan.dout_holder["src"].data_ptr(), g.ptr, B, Co, H, W, g.ldc, g.C, stream)
        dout_in.__name__ = "nchw_to_nhwc"
        plan.bwd.append((dout_in, ()))
        lower_block_bwd(plan, rec)
        plan.mark_ready()
        if need_dx:
            plan.dx_nchw = torch.empty(B, self.in_channels, H, W, dtype=torch.float32, device=device)
            plan.call(plan.bwd, L.nhwc_to_nchw, dt, xin.grad.ptr, xin.grad.ldc, plan.dx_nchw.data_ptr(), B, self.in_channels, H, W)
        return plan
