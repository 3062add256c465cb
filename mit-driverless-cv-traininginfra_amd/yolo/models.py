"""MI355X-native drop-in for the reference's CVC-YOLOv3/models.py (Darknet, YOLOLayer, create_modules).

Same Python surface — `Darknet(config_path, xy_loss, wh_loss, no_object_loss, object_loss, vanilla_anchor)`, the getters,
`forward(x, targets=None)`, darknet `.weights` load/save, `module_list` / `module_defs` / `hyperparams`, state_dict keys
(`module_list.<i>.conv_<i>.weight` ...) — so the reference's train.py / validate.py / detect.py call sequence runs
unchanged (reference: CVC-YOLOv3/models.py:15-422; callers train.py:100-120,182,191,196,208,217).

Underneath, `forward` does not run the nn.Modules: the cfg is lowered once per input shape into a static launch plan
(engine.Plan) of hand-written gfx950 kernels from libmdcv_hip.so — NHWC bf16 (or fp32 parity mode) implicit-GEMM MFMA
convolutions with BatchNorm statistics in the epilogue, fused BN-apply+LeakyReLU(+shortcut), zero-copy route concat,
fused YOLO heads — and the whole backward is one autograd node that runs the mirrored plan.  The nn.Modules only hold the
fp32 master parameters (OIHW) and BatchNorm buffers.  There is no CPU fallback.

Extra (non-reference) knobs: `precision=` ctor kwarg or env MDCV_PRECISION in {"bf16" (default), "fp32"};
env MDCV_GRAPH=1 replays the plan through hipGraphs.
"""
from __future__ import division

import csv
import ctypes
import os
from datetime import datetime

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from ..engine import Plan, TNode, ConvSpec, BnSpec, pad8, parse_precision, side_stream, ACT_NONE, ACT_LEAKY, ACT_RELU
from .utils.parse_config import parse_model_config

vanilla_anchor_list = [[10, 13], [16, 30], [33, 23], [30, 61], [62, 45], [59, 119], [116, 90], [156, 198], [373, 326]]


def _read_anchor_row(csv_uri):
    """Row 0 of train.csv holds the k-means anchors as one quoted field 'w,h|w,h|...' (reference models.py:29-35)."""
    with open(csv_uri) as fh:
        first = next(csv.reader(fh))
    text = str(first)[2:-2]
    return [[float(v) for v in pair.split(",")] for pair in text.split("'")[0].split("|")]


class EmptyLayer(nn.Module):
    """Placeholder module for 'route' and 'shortcut' sections (keeps module_list indices aligned with the cfg)."""


def create_modules(module_defs, xy_loss, wh_loss, no_object_loss, object_loss, vanilla_anchor):
    """cfg sections -> (hyperparams, nn.ModuleList).  Naming / ordering rules follow reference models.py:15-110 so that
    state_dict keys and `.weights` traversal (`module[0]` conv, `module[1]` BN) stay compatible."""
    hyper = module_defs.pop(0)
    channels = [int(hyper["channels"])]
    img_w, img_h = int(hyper["width"]), int(hyper["height"])
    n_cls = int(hyper["classes"])
    slope = float(hyper["leaky_slope"])
    activation = hyper["conv_activation"]
    masks = [[int(v) for v in grp.split(",")] for grp in hyper["yolo_masks"].split("|")]
    anchors_all = _read_anchor_row(hyper["train_uri"])          # opened unconditionally, like the reference (Q11)
    if vanilla_anchor:
        anchors_all = vanilla_anchor_list
    ignore = float(hyper["build_targets_ignore_thresh"])

    mods = nn.ModuleList()
    head = 0
    for i, d in enumerate(module_defs):
        seq = nn.Sequential()
        kind = d["type"]
        if kind == "convolutional":
            is_head = d["filters"] == "preyolo"                   # bias, no BN, linear (models.py:51-54)
            filters = (n_cls + 5) * len(masks[head]) if is_head else int(d["filters"])
            k = int(d["size"])
            seq.add_module("conv_%d" % i, nn.Conv2d(channels[-1], filters, k, int(d["stride"]), (k - 1) // 2, bias=is_head))
            if not is_head:
                seq.add_module("batch_norm_%d" % i, nn.BatchNorm2d(filters))
                if activation == "leaky":
                    seq.add_module("leaky_%d" % i, nn.LeakyReLU(slope))
                if activation == "ReLU":
                    seq.add_module("ReLU_%d" % i, nn.ReLU())
        elif kind == "maxpool":
            k, s = int(d["size"]), int(d["stride"])
            if k == 2 and s == 1:
                seq.add_module("_debug_padding_%d" % i, nn.ZeroPad2d((0, 1, 0, 1)))
            seq.add_module("maxpool_%d" % i, nn.MaxPool2d(k, s, (k - 1) // 2))
            filters = channels[-1]
        elif kind == "upsample":
            seq.add_module("upsample_%d" % i, nn.Upsample(scale_factor=int(d["stride"]), mode="nearest"))
            filters = channels[-1]
        elif kind == "route":
            filters = 0
            for v in (int(t) for t in d["layers"].split(",")):
                filters += channels[v + 1 if v > 0 else v]          # `channels` has the input in front (models.py:93-96)
            seq.add_module("route_%d" % i, EmptyLayer())
        elif kind == "shortcut":
            filters = channels[int(d["from"])]
            seq.add_module("shortcut_%d" % i, EmptyLayer())
        elif kind == "yolo":
            seq.add_module("yolo_%d" % i, YOLOLayer([anchors_all[v] for v in masks[head]], n_cls, img_h, img_w, ignore, activation,
                                                    xy_loss, wh_loss, object_loss, no_object_loss))
            head += 1
            filters = channels[-1]
        else:
            raise ValueError("unknown cfg section [%s]" % kind)
        mods.append(seq)
        channels.append(filters)
    return hyper, mods


class _YoloHeadFn(torch.autograd.Function):
    """Stand-alone YOLOLayer on an NCHW sample: NHWC fp32 staging + the fused head kernels."""

    @staticmethod
    def forward(ctx, layer, sample, targets):
        L = _lib.lib()
        B, ch, Gh, Gw = sample.shape
        A, C = layer.num_anchors, layer.num_classes
        dev = sample.device
        st = torch.cuda.current_stream().cuda_stream
        cp = pad8(ch)
        lg = torch.empty(B * Gh * Gw * cp, dtype=torch.float32, device=dev)
        src = sample.detach().to(torch.float32).contiguous()
        L.check(L.nchw_to_nhwc(_lib.F32, src.data_ptr(), lg.data_ptr(), B, ch, Gh, Gw, cp, cp, st), "nchw_to_nhwc")
        tg = targets.detach().to(device=dev, dtype=torch.float32).contiguous()
        anchors = layer.scaled_anchors(Gh).to(dev)
        ws = torch.empty(int(L.yolo_head_workspace_bytes(B, A, Gh, Gw)), dtype=torch.uint8, device=dev)
        out7 = torch.zeros(7, dtype=torch.float32, device=dev)
        geo = (B, tg.shape[1], A, C, Gh, Gw, float(layer.ignore_thres), float(layer.xy_loss), float(layer.wh_loss),
               float(layer.object_loss), float(layer.no_object_loss))
        L.check(L.yolo_head_train(_lib.F32, lg.data_ptr(), cp, None, 0, cp, tg.data_ptr(), anchors.data_ptr(), *geo, ws.data_ptr(),
                                  out7.data_ptr(), None, st), "yolo_head_train")
        if _CHECK_TARGETS and int(_head_err_view(ws, B, A, Gh, Gw).item()):
            raise IndexError(_BAD_TARGET_MSG)
        ctx.saved = (lg, tg, anchors, ws, geo, cp, ch)
        return out7

    @staticmethod
    def backward(ctx, gout):
        L = _lib.lib()
        lg, tg, anchors, ws, geo, cp, ch = ctx.saved
        B, Gh, Gw = geo[0], geo[4], geo[5]
        st = torch.cuda.current_stream().cuda_stream
        g = gout.contiguous()
        dl = torch.empty_like(lg)
        L.check(L.yolo_head_grad(_lib.F32, lg.data_ptr(), cp, dl.data_ptr(), cp, cp, tg.data_ptr(), anchors.data_ptr(), *geo, ws.data_ptr(),
                                 g.data_ptr(), st), "yolo_head_grad")
        ds = torch.empty(B, ch, Gh, Gw, dtype=torch.float32, device=lg.device)
        L.check(L.nhwc_to_nchw(_lib.F32, dl.data_ptr(), cp, ds.data_ptr(), B, ch, Gh, Gw, st), "nhwc_to_nchw")
        return None, ds, None


class YOLOLayer(nn.Module):
    """Detection head.  ctor argument order as in the reference (note object_loss before no_object_loss, models.py:121).
    forward(sample, targets) -> (loss, tensor(6 parts: x,y,w,h,obj,noobj)) ; forward(sample) -> [B, A*G*G, 5+C]."""

    def __init__(self, anchors, num_classes, img_height, img_width, build_targets_ignore_thresh, conv_activation, xy_loss, wh_loss,
                 object_loss, no_object_loss):
        super().__init__()
        self.anchors = anchors
        self.num_anchors = len(anchors)
        self.num_classes = num_classes
        self.bbox_attrs = 5 + num_classes
        self.image_height, self.image_width = img_height, img_width
        self.ignore_thres = build_targets_ignore_thresh
        self.xy_loss, self.wh_loss = xy_loss, wh_loss
        self.no_object_loss, self.object_loss = no_object_loss, object_loss
        self.conv_activation = conv_activation

    def stride_for(self, grid_h):
        return self.image_height / grid_h                          # cfg height, used for both axes (models.py:145)

    def scaled_anchors(self, grid_h):
        s = self.stride_for(grid_h)
        return torch.tensor([(aw / s, ah / s) for aw, ah in self.anchors], dtype=torch.float32)

    def forward(self, sample, targets=None):
        _lib.require_gpu(sample)
        if targets is not None:
            out7 = _YoloHeadFn.apply(self, sample, targets)
            return out7[0], out7[1:].detach()
        L = _lib.lib()
        B, ch, Gh, Gw = sample.shape
        st = torch.cuda.current_stream().cuda_stream
        cp = pad8(ch)
        lg = torch.empty(B * Gh * Gw * cp, dtype=torch.float32, device=sample.device)
        src = sample.detach().to(torch.float32).contiguous()
        L.check(L.nchw_to_nhwc(_lib.F32, src.data_ptr(), lg.data_ptr(), B, ch, Gh, Gw, cp, cp, st), "nchw_to_nhwc")
        rows = self.num_anchors * Gh * Gw
        out = torch.empty(B, rows, self.bbox_attrs, dtype=torch.float32, device=sample.device)
        an = self.scaled_anchors(Gh).to(sample.device)
        L.check(L.yolo_head_decode(_lib.F32, lg.data_ptr(), cp, an.data_ptr(), float(self.stride_for(Gh)), B, self.num_anchors,
                                   self.num_classes, Gh, Gw, out.data_ptr(), rows, 0, st), "yolo_head_decode")
        return out


class _DarknetTrainFn(torch.autograd.Function):
    """One autograd node for the whole network: forward = plan.fwd, backward = plan.bwd."""

    @staticmethod
    def forward(ctx, model, plan, x, targets, *params):
        plan.run_forward(x, targets)
        ctx.model, ctx.plan = model, plan
        return plan.out7.clone()

    @staticmethod
    def backward(ctx, gout):
        model, plan = ctx.model, ctx.plan
        model._run_backward(plan, gout)
        return (None, None, None, None) + (None,) * len(model._plist)


_CHECK_TARGETS = True       # (module attribute: tests may switch the one-step-late label check off)
_BAD_TARGET_MSG = ("index out of range in build_targets: a target has cx >= 1.0 or cy >= 1.0 (grid cell == grid size), where the reference "
                   "raises IndexError at utils/utils.py:262")


def _head_err_view(ws, B, A, Gh, Gw):
    """the int32 `err` word of a YOLO head workspace (csrc/yolo_head.hip: owner[B*A*Gh*Gw] | ignore[Gh*Gw] | err)"""
    return ws.view(torch.int32)[B * A * Gh * Gw + Gh * Gw:B * A * Gh * Gw + Gh * Gw + 1]


_EVAL_FUSE = True           # inference: conv + BatchNorm(running stats) + activation in one launch (module attribute; False: two-pass plans)


class _NetPlan(Plan):
    """engine.Plan + the per-network I/O buffers and the run_* entry points."""

    err_views = ()        # int32 one-element views of the YOLO heads' `err` words (training plans)

    def check_targets(self, block=True):
        """Raises IndexError if the targets of the LAST forward through this plan held a centre coordinate >= 1.0 (the reference's
        build_targets indexes its [B,A,G,G] tensors with gi == G there and raises, utils/utils.py:262; the fused head drops the
        target and sets a flag).  The flags ride to a pinned host word with a non-blocking copy behind every forward and are
        looked at (i) at the end of that step's backward(), behind its queued launches -- waiting for the copy (one host wait per step
        that the GPU does not see: a bad label then raises BEFORE optimizer.step(), like the reference) unless the model sets
        `strict_targets = False`, in which case the look is a query and a copy that has not landed stays pending --, (ii) at the
        start of the NEXT forward and (iii) when the plan is dropped."""
        ev = getattr(self, "_err_event", None)
        if ev is None:
            return
        if not block and not ev.query():
            return                                           # not landed yet: the next forward (or the plan's eviction) looks again
        self._err_event = None
        ev.synchronize()
        if bool(self._err_host.any()):
            raise IndexError(_BAD_TARGET_MSG)

    def run_forward(self, x, targets=None):
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        if targets is not None and self.err_views and _CHECK_TARGETS:
            self.check_targets()
        self.in_holder["src"] = x
        if targets is not None:
            self.targets.copy_(targets.reshape(self.targets.shape), non_blocking=True)
        st = torch.cuda.current_stream().cuda_stream
        self.run(self.pre, st)
        if self.use_graph:
            self._graphed("fwd")
        else:
            self.run(self.fwd, st)
        if targets is not None and self.err_views and _CHECK_TARGETS:
            if getattr(self, "_err_host", None) is None:
                self._err_host = torch.zeros(len(self.err_views), dtype=torch.int32).pin_memory()
            self._err_host.copy_(torch.cat(self.err_views), non_blocking=True)
            self._err_event = torch.cuda.Event()
            self._err_event.record()

    def _graphed(self, which):
        """Replay (first call: warm run + capture) a launch list as a hipGraph.  Capture is illegal on the legacy default
        stream, so graph mode runs on a side stream fenced against the caller's stream on both sides."""
        cur = torch.cuda.current_stream()
        if getattr(self, "_gstream", None) is None:
            self._gstream = torch.cuda.Stream(device=self.device)
            self._graphs = {}
        gs = self._gstream
        gs.wait_stream(cur)
        with torch.cuda.stream(gs):
            key = (which, getattr(self, "flags", None))
            g = self._graphs.get(key)
            lst = self.fwd if which == "fwd" else self.bwd
            if g is None:
                self.run(lst, gs.cuda_stream)                          # warm run (function attributes, lazy buffers)
                self._graphs[key] = self.capture(which, gs.cuda_stream)
            else:
                self.L.check(self.L.graph_launch(g, gs.cuda_stream), "graph_launch")
        cur.wait_stream(gs)

    # Weight gradients are leaves of the backward dependency chain (only the optimizer / the gradient exchange read them), so
    # they run on a side stream: the MFMA-bound wgrad kernels of layer L overlap the HBM-bound BatchNorm passes and the
    # latency-bound tiny kernels of layers L-1, L-2, ... on the main stream.  Every buffer of a plan is its own allocation (no
    # pooling), so the only ordering needed is "dY(L), X(L) ready" (side waits on main) and "all gradients done" (main waits on
    # side at the end; the data-parallel reducer's comm stream waits on both).
    overlap_wgrad = os.environ.get("MDCV_WGRAD_STREAM", "1") == "1"
    fork_device_scope = 1

    def side(self):
        if getattr(self, "_side", None) is None:
            self._side = side_stream(self.device)           # one per device, checked to overlap with the current stream (engine.side_stream)
        return self._side

    def run_bwd_list(self):
        """The backward launch list on the current stream, weight gradients on the side stream (see above)."""
        cur = torch.cuda.current_stream()
        if not self.overlap_wgrad or "run" in self.__dict__:                   # (bench.py's per-kernel timing swaps `run`)
            self.run(self.bwd, cur.cuda_stream)
            return
        side = self.side()
        st, ss = cur.cuda_stream, side.cuda_stream
        L = self.L
        fork, used = L.stream_fork, False                      # (one ring event, device-scope release; torch's wait_stream builds an Event per call)
        rkey = (len(self.bwd), self.fork_on_dispatch, self.defer_slab_reduce)
        if self.__dict__.get("_bwd_roles_key") != rkey:      # (A/B scripts flip the two switches after the first backward)
            self._bwd_roles, self._bwd_roles_key = self._classify_bwd(), rkey
        roles = self._bwd_roles
        ev = ctypes.c_void_p()
        armed = False
        pending = []                                           # deferred slab reduces (role 3): they ride behind the NEXT fork
        overlapped_dp = self.on_ready is not None

        def flush():
            for pfn, pargs in pending:
                prc = pfn(*pargs, ss)
                if prc:
                    raise _lib.MdcvError(f"{getattr(pfn, '__name__', pfn)} returned {prc}")
            del pending[:]
        dp_red = getattr(self.on_ready, "__self__", None) if overlapped_dp else None
        for (fn, args), role in zip(self.bwd, roles):
            if role == 3 and (not overlapped_dp or dp_red is not None):   # slab reduce of a one-launch 1x1 backward: its producer is already in the main
                pending.append((fn, args))                     # queue, so ANY later fork orders it; no fork (and no 5 us of main queue) of its own
                continue
            if pending and dp_red is not None and getattr(fn, "__name__", "") == "grad_ready" and dp_red.would_fire(fn.low_water):
                # data parallel: this marker starts the all-reduce of a bucket, which waits for the side stream -- the deferred slab reduces of the
                # bucket's layers must be IN that stream first (about eight buckets per YOLOv3 step: eight forks instead of one per 1x1 layer)
                L.check(fork(st, ss, self.fork_device_scope), "stream_fork")
                flush()
                used = True
            if role >= 2:                                      # a weight gradient: side stream, behind "dY(L), X(L) ready"
                if armed:
                    L.check(L.stream_fork_wait(ss, ev), "stream_fork_wait")           # the kernel in front of it carried the event
                    armed = False
                else:
                    L.check(fork(st, ss, self.fork_device_scope), "stream_fork")
                if pending:
                    flush()
                rc = fn(*args, ss)
                used = True
            else:
                if role == 1:                                  # single-kernel call in front of a weight gradient: its dispatch carries the event
                    L.check(L.stream_fork_arm(st, self.fork_device_scope, ctypes.byref(ev)), "stream_fork_arm")
                    armed = True
                rc = fn(*args, st)
            if rc:
                if armed:                                      # the armed call failed before it launched: take the event back, or the next unrelated
                    L.stream_fork_wait(ss, ev)                 # launch of this thread would carry it as its stop event (fork_wait clears a pending arm)
                raise _lib.MdcvError(f"{getattr(fn, '__name__', fn)} returned {rc}")
        if pending:
            L.check(fork(st, ss, self.fork_device_scope), "stream_fork")
            flush()
            used = True
        if used:
            L.check(fork(ss, st, self.fork_device_scope), "stream_fork")              # main waits for "all gradients done"

    # An event record between two dependent kernels of the main queue costs that queue ~7 us (rocprofv3 trace, round 5: 7.2 - 7.8 us between a
    # kernel and its successor wherever a fork sat between them, 0.0 - 0.6 us elsewhere; 71 forks per YOLOv3 backward).  Where the call in
    # front of a weight gradient is ONE kernel launch, that kernel's own dispatch packet carries the event (mdcv_stream_fork_arm) instead.
    fork_on_dispatch = True
    defer_slab_reduce = True           # the slab reduces of the one-launch 1x1 backward wait for the next weight gradient's fork (32 forks fewer per YOLOv3 step)

    def _classify_bwd(self):
        """per backward-list entry: 2 = weight gradient (side stream), 3 = slab reduce that may wait for the next fork, 1 = single-kernel library call
        right in front of a weight gradient, 0 = other"""
        L = self.L
        single = (L.bn_act_bwd_apply, L.pw_bwd)
        n = len(self.bwd)
        roles = [2 if getattr(fn, "__name__", "") == "conv2d_wgrad" else 0 for fn, _ in self.bwd]
        if self.defer_slab_reduce:
            for i, (fn, _) in enumerate(self.bwd):
                info = getattr(fn, "info", None)
                if roles[i] == 2 and info is not None and len(info) > 7 and info[7] == 0:        # k == 0: the reduce alone (engine._emit_pw_bwd1)
                    roles[i] = 3
        if self.fork_on_dispatch:
            for i in range(n - 1):
                if roles[i] == 0 and roles[i + 1] == 2 and any(self.bwd[i][0] is f for f in single):
                    roles[i] = 1
        return roles

    def run_backward(self, gout):
        self.gscale.copy_(gout.reshape(-1)[:self.gscale.numel()], non_blocking=True)
        if self.use_graph:
            self._graphed("bwd")
        else:
            self.run_bwd_list()


class FlatParamsMixin:
    """Keeps all parameters (and their gradients) as views of two flat fp32 buffers so the optimizer step and the RCCL
    gradient all-reduce are single passes over contiguous HBM."""

    def _flatten(self):
        if getattr(self, "_pipe_plan", None) is not None:
            self._param_sync()                               # a pipelined optimizer step may still be updating the old buffers
        plist = [p for p in self.parameters()]
        dev = plist[0].device
        total = sum((p.numel() + 3) & ~3 for p in plist)           # every parameter starts on a 16-byte boundary (float4 rows in the pack kernel;
        pflat = torch.zeros(total, dtype=torch.float32, device=dev)  # the 255-element head biases would misalign everything behind them); padding stays 0
        gflat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        self._goff = {}
        with torch.no_grad():
            for p in plist:
                n = p.numel()
                pflat[off:off + n].copy_(p.data.reshape(-1))
                p.data = pflat[off:off + n].view(p.shape)
                self._goff[id(p)] = (off, n)
                off += (n + 3) & ~3
        self._plist, self._pflat, self._gflat = plist, pflat, gflat
        self._flat_ptrs = [p.data_ptr() for p in plist]
        self._plans = {}
        self._pipe_plan = None
        self._last_train_plan = None         # it was built on the old flat buffers: a pipelined optimizer step must not reuse its pack table
        self._params_changed()

    # run-time caches that must not travel with a copy / pickle of the model: launch plans hold ctypes function pointers, raw device
    # pointers and closures (copy.deepcopy(model) after a forward -- RektNet/train_eval.py:99 -- raised "ctypes objects containing
    # pointers cannot be pickled"); the copy re-flattens its parameters and rebuilds its plans on first use.
    def _replicate_for_data_parallel(self):
        """nn.DataParallel (reference train.py:193-195 wraps the model when torch.cuda.device_count() > 1) copies the module tree onto
        every device per forward.  These models own flat parameter / gradient buffers, ctypes launch plans bound to raw device
        pointers, and side streams: a replica would launch kernels on device 0's memory.  Fail loudly instead."""
        raise RuntimeError(
            f"{type(self).__name__} cannot be replicated by torch.nn.DataParallel: the MI355X-native path is one process per GPU. "
            "Launch the script with `python -m torch.distributed.run --nproc-per-node N ...`, give every rank its shard of the batch and "
            "attach `mdcv.parallel.GradAllReducer.attach(model)` (all-reduce(SUM) of the flat gradient over RCCL -- the same per-shard "
            "BatchNorm / build_targets + summed-gradient semantics as DataParallel), or hide the other GPUs from a single-process run "
            "(HIP_VISIBLE_DEVICES=0).  See INTEGRATION.md §1.")

    _TRANSIENT = ("_plans", "_pipe_plan", "_last_train_plan", "_dp_reducer", "_dp_auto", "_pflat", "_gflat", "_flat_ptrs", "_goff", "_plist", "_flat_parent")

    def _state_without_plans(self):
        d = {k: v for k, v in self.__dict__.items() if k not in self._TRANSIENT}
        d["_plans"] = {}
        return d

    # Launch plans are cached per (batch shape, mode) and own every buffer they touch (~10 GB for YOLOv3 at batch 32), so the cache
    # is an LRU bounded by activation bytes (MDCV_PLAN_CACHE_GB, default 48) and by count (MDCV_MAX_PLANS, default 16): a ragged last
    # batch (the reference's DataLoader has no drop_last) or validation at other resolutions costs extra plans only while they are
    # in use, not one per shape ever seen.  The plan in use is never evicted.
    max_plans = int(os.environ.get("MDCV_MAX_PLANS", "16"))
    max_plan_bytes = int(float(os.environ.get("MDCV_PLAN_CACHE_GB", "48")) * (1 << 30))

    def _plan_lookup(self, key):
        plan = self._plans.get(key)
        if plan is not None and next(reversed(self._plans)) != key:
            self._plans[key] = self._plans.pop(key)          # most recently used last
        return plan

    def _plan_store(self, key, plan):
        self._plans[key] = plan
        while len(self._plans) > 1 and (len(self._plans) > max(1, self.max_plans) or
                                        sum(getattr(p, "bytes", 0) for p in self._plans.values()) > self.max_plan_bytes):
            victim = next(k for k in self._plans if k != key)
            self._evict_plan(victim)

    def _evict_plan(self, key):
        plan = self._plans.pop(key)
        try:
            if getattr(self, "_pipe_plan", None) is plan:
                self._param_sync()                           # its deferred parameter-group updates must land first
                self._pipe_plan = None
        finally:
            if getattr(self, "_last_train_plan", None) is plan:
                self._last_train_plan = None
        if getattr(plan, "err_views", ()) and _CHECK_TARGETS:
            plan.check_targets()                             # a pending bad-label flag must not be lost with the plan (last batch of a run);
                                                             # raised AFTER the bookkeeping above, so the model is consistent when it does
        # the plan's buffers go back to the caching allocator when the last reference dies (an autograd graph that still needs the
        # plan for its backward holds one); every stream that used them was joined into the current stream at the end of its step

    def release_plans(self):
        """Drop every cached launch plan (and its HBM buffers); the next forward rebuilds what it needs."""
        err = None
        for k in list(getattr(self, "_plans", {})):
            try:
                self._evict_plan(k)
            except IndexError as e:                          # a pending bad-label flag: drop every plan first, then report it
                err = e
        if err is not None:
            raise err

    def _params_changed(self):
        """Parameters were rewritten behind the optimizer's back (load_weights / load_state_dict): operands packed ahead of the next
        forward by a pipelined optimizer step are stale."""
        self._param_epoch = getattr(self, "_param_epoch", 0) + 1

    def _param_versions(self):
        """Sum of the parameters' autograd version counters: moves when user code edits any parameter in place (the HIP kernels do not)."""
        return sum(p._version for p in self._plist)

    def _flat_ok(self):
        pl = getattr(self, "_plist", None)
        if pl is None:
            return False
        return all(p.data_ptr() == q for p, q in zip(pl, self._flat_ptrs))       # every parameter still is its view of the flat buffer (~15 us)

    def _grad_view(self, p):
        off, n = self._goff[id(p)]
        return self._gflat[off:off + n].view(p.shape)

    def flat_parameters(self):
        """(flat fp32 parameter buffer, flat fp32 gradient buffer) — what FusedAdam / the all-reduce operate on."""
        if not self._flat_ok():
            self._flatten()
        self._param_sync()
        return self._pflat, self._gflat

    def _param_sync(self):
        """Orders the current stream behind a pipelined optimizer step (optim.py, pipeline=True) that may still be updating
        parameter groups on the parameter stream.  Every reader of the parameters outside the pipelined forward goes through here."""
        plan = getattr(self, "_pipe_plan", None)
        if plan is None:
            return
        for k in range(len(plan._pending_updates)):
            plan.launch_param_group(k, gated=False)
        for k, ev in enumerate(plan._group_events):
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
                plan._group_events[k] = None

    _dp_average = False                                      # KeypointNet overrides: its loss is a batch MEAN (see rektnet/keypoint_net.py)

    def _auto_dp_shard(self, *tensors):
        """Under torchrun with the drop-in modules (parallel.enable_auto_data_parallel): rank r's share of a training batch -- nn.DataParallel's
        scatter on dim 0, reference train.py:68 / :193-195 -- and, on first use, the overlapped gradient all-reduce attached to this model and
        the replicas synchronised from rank 0.  -> (tensors, weight): weight 0.0 marks a rank whose chunk was empty (its outputs are to be
        multiplied by zero), None / 1.0 anything else.  Off (the usual case): the tensors pass through."""
        from ..parallel import auto_shard, auto_attach
        out, weight = auto_shard(*tensors)
        if weight is not None:
            auto_attach(self, average=self._dp_average)
        return out, weight

    def _run_backward(self, plan, gout):
        self._last_train_plan = plan
        pl = self._plist
        keep = None
        if pl[0].grad is not None:                       # gradients were not reset to None: accumulate semantics
            keep = self._gflat.clone()
        red = getattr(self, "_dp_reducer", None)
        # (an attached reducer with ONE rank has nothing to exchange: no markers, and the backward keeps its single-GPU schedule -- the markers switch
        #  off the deferred slab reduces of run_bwd_list, 32 forks = 0.17 ms of main queue per YOLOv3 step, which bench.py at --gpus 1 paid until round 5)
        overlap = red is not None and red._active() and keep is None and not plan.use_graph
        plan.on_ready = red.on_ready if overlap else None
        if red is not None:
            red.extra_streams = [plan.side()] if (plan.overlap_wgrad and not plan.use_graph and self._gflat.is_cuda) else []
            red.begin(self._gflat, overlap)
        plan.run_backward(gout)
        plan.on_ready = None
        if keep is not None:
            self._gflat.add_(keep)
        if red is not None:
            red.backward_done()
            if getattr(self, "_dp_auto", False):
                red.finish()                                 # (auto data parallel: nobody else will order the optimizer behind the exchange)
        for p in pl:
            v = self._grad_view(p)
            if p.grad is None:
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():      # foreign .grad tensor: fold ours in, then re-point
                v.add_(p.grad)
                p.grad = v
        # a label with cx / cy >= 1.0 (reference: IndexError inside build_targets, BEFORE any update, utils/utils.py:262).  Looked at HERE, with
        # the backward launches already queued: the host's wait for the flag copy behind the forward then costs the GPU nothing (at the head
        # of the backward it left the GPU idle until the first backward launch arrived: 13.77 vs 13.65 ms per step), the heads dropped the
        # bad target in the forward so the queued backward is well defined, and the error still leaves backward() -- before optimizer.step()
        if getattr(plan, "err_views", ()) and _CHECK_TARGETS:
            plan.check_targets(block=getattr(self, "strict_targets", True))


class Darknet(FlatParamsMixin, nn.Module):
    """YOLOv3 object detection model (reference: CVC-YOLOv3/models.py:222-422)."""

    def __init__(self, config_path, xy_loss, wh_loss, no_object_loss, object_loss, vanilla_anchor, precision=None):
        super().__init__()
        self.module_defs = parse_model_config(config_path)
        self.hyperparams, self.module_list = create_modules(self.module_defs, xy_loss, wh_loss, no_object_loss, object_loss, vanilla_anchor)
        h = self.hyperparams
        self.img_width, self.img_height = int(h["width"]), int(h["height"])
        self.onnx_height = int(h["onnx_height"])
        self.onnx_name = config_path.split("/")[-1].split(".")[0] + "_" + str(self.img_width) + str(self.onnx_height) + ".onnx"
        self.num_classes = int(h["classes"])
        ch = int(h["channels"])
        if ch not in (1, 3):
            print("Channels in cfg file is not set properly, making it colour")
        self.bw = ch == 1
        self.validate_uri, self.train_uri = h["validate_uri"], h["train_uri"]
        self.num_train_images, self.num_validate_images = int(h["num_train_images"]), int(h["num_validate_images"])
        self.conf_thresh, self.nms_thresh, self.iou_thresh = float(h["conf_thresh"]), float(h["nms_thresh"]), float(h["iou_thresh"])
        self.start_weights_dim = [int(v) for v in h["start_weights_dim"].split(",")]
        self.conv_activation = h["conv_activation"]
        self.xy_loss, self.wh_loss, self.no_object_loss, self.object_loss = xy_loss, wh_loss, no_object_loss, object_loss
        self.anchors = vanilla_anchor_list if vanilla_anchor else _read_anchor_row(h["train_uri"])
        self.seen = 0
        self.header_info = np.array([0, 0, 0, self.seen, 0], dtype=np.int32)
        self._stamp = (datetime.now().strftime("%B").lower(), str(datetime.now().year))
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

    # ---- getters consumed by train.py / validate.py / detect.py (reference models.py:279-310)
    def get_start_weight_dim(self): return self.start_weights_dim
    def get_onnx_name(self): return self.onnx_name
    def get_bw(self): return self.bw
    def get_loss_constant(self): return [self.xy_loss, self.wh_loss, self.no_object_loss, self.object_loss]
    def get_conv_activation(self): return self.conv_activation
    def get_num_classes(self): return self.num_classes
    def get_anchors(self): return self.anchors
    def get_threshs(self): return self.conf_thresh, self.nms_thresh, self.iou_thresh
    def img_size(self): return self.img_width, self.img_height
    def get_links(self): return self.validate_uri, self.train_uri
    def num_images(self): return self.num_validate_images, self.num_train_images

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, x, targets=None):
        _lib.require_gpu(x)
        if not self._flat_ok():
            self._flatten()
        if targets is not None and self.training and torch.is_grad_enabled():
            (x, targets), dp_weight = self._auto_dp_shard(x, targets)     # torchrun on the unchanged train.py: rank r's shard (parallel.enable_auto_data_parallel)
        else:
            dp_weight = None
        B, _, H, W = x.shape
        T = targets.shape[1] if targets is not None else 0
        key = (B, H, W, T, targets is not None, self.training, self.precision, x.device.index)
        plan = self._plan_lookup(key)
        if plan is None:
            plan = self._build_plan(x.device, B, H, W, T, targets is not None, self.training)
            self._plan_store(key, plan)
        if getattr(self, "_pipe_plan", None) is not None and plan is not self._pipe_plan:
            self._param_sync()               # deferred group updates of a pipelined optimizer step are only released from ITS plan's forward list
        if targets is None:
            plan.run_forward(x)
            return plan.eval_out.clone()
        if torch.is_grad_enabled() and plan.has_bwd:
            out7 = _DarknetTrainFn.apply(self, plan, x, targets, *self._plist)
        else:
            plan.run_forward(x, targets)
            out7 = plan.out7.clone()
        if dp_weight == 0.0:
            out7 = out7 * 0.0                # this rank's DataParallel chunk was empty: it joins the exchange with exact-zero gradients
        d = out7.detach()
        return (out7[0], d[1], d[2], d[3], d[4], d[5], d[6])

    # ------------------------------------------------------------------------------------------ plan
    def _build_plan(self, device, B, H, W, T, with_targets, bn_train):
        defs, mods = self.module_defs, self.module_list
        n = len(defs)
        plan = _NetPlan(device, self.precision, bn_train, grad_sink=self._grad_view)
        plan.owner = self
        plan.grad_offset = lambda p: self._goff[id(p)][0]
        plan.use_graph = self.use_graph
        plan.pre = []
        L, dt = plan.L, plan.dtype
        cin = int(self.hyperparams["channels"])
        xin, holder = plan.emit_input(B, cin, H, W)
        plan.pre.append(plan.fwd.pop())                      # the NCHW->NHWC edge stays outside any captured graph
        plan.in_holder = holder
        plan.targets = torch.zeros(B, max(T, 1), 5, dtype=torch.float32, device=device)
        plan.out7 = torch.zeros(7, dtype=torch.float32, device=device)
        plan.gscale = torch.ones(1, dtype=torch.float32, device=device)
        plan.has_bwd = with_targets

        def res(i, v):                                       # cfg layer reference -> absolute module index
            return i + v if v < 0 else v

        # ---- who consumes what (fusion + concat planning)
        users = [[] for _ in range(n)]
        for i, d in enumerate(defs):
            k = d["type"]
            if k in ("convolutional", "upsample", "maxpool", "yolo") and i > 0:
                users[i - 1].append(i)
            elif k == "route":
                for v in (int(t) for t in d["layers"].split(",")):
                    users[res(i, v)].append(i)
            elif k == "shortcut":
                users[i - 1].append(i)
                users[res(i, int(d["from"]))].append(i)
        # ---- shapes
        shp = []
        c, h, w = cin, H, W
        for i, d in enumerate(defs):
            k = d["type"]
            if k == "convolutional":
                conv = mods[i][0]
                c = conv.out_channels
                h = (h + 2 * conv.padding[0] - conv.kernel_size[0]) // conv.stride[0] + 1
                w = (w + 2 * conv.padding[1] - conv.kernel_size[1]) // conv.stride[1] + 1
            elif k == "upsample":
                h, w = h * int(d["stride"]), w * int(d["stride"])
            elif k == "maxpool":
                ks, st_ = int(d["size"]), int(d["stride"])
                if ks == 2 and st_ == 1:
                    pass                                 # ZeroPad2d((0,1,0,1)) + MaxPool2d(2, 1): same size (models.py:77-79)
                else:                                    # MaxPool2d(size, stride, (size - 1) // 2)
                    if ks > 15:
                        raise NotImplementedError("max-pool windows up to 15x15 are lowered")
                    pp = (ks - 1) // 2
                    h, w = (h + 2 * pp - ks) // st_ + 1, (w + 2 * pp - ks) // st_ + 1
            elif k == "route":
                src = [res(i, int(t)) for t in d["layers"].split(",")]
                c = sum(shp[s][0] for s in src)
                h, w = shp[src[0]][1], shp[src[0]][2]
            elif k == "shortcut":
                c, h, w = shp[i - 1]
            shp.append((c, h, w))
        # ---- concat destinations: a producer writes straight into its slice of the route buffer
        dest = {}
        parents = {}
        for i, d in enumerate(defs):
            if d["type"] == "route":
                src = [res(i, int(t)) for t in d["layers"].split(",")]
                if len(src) > 1:
                    ctot = sum(pad8(shp[s][0]) for s in src)
                    par = plan.new_act(B, shp[i][1], shp[i][2], ctot)
                    parents[i] = par
                    off = 0
                    for s in src[:-1]:
                        if shp[s][0] % 8:
                            # the concat buffer places every source at a multiple-of-8 channel offset (16-byte vectors); the consumer's
                            # packed weights index input channels contiguously, so a pad hole in the middle would misalign them
                            raise NotImplementedError(f"[route] at section {i}: source {s} has {shp[s][0]} channels; every concat source but the "
                                                      f"last must have a multiple of 8 channels")
                    for s in src:
                        if s not in dest and defs[s]["type"] in ("convolutional", "upsample", "shortcut", "maxpool"):
                            dest[s] = (par, off)
                        off += pad8(shp[s][0])

        def out_act(i):
            c_, h_, w_ = shp[i]
            if i in dest:
                par, off = dest[i]
                return par.slice(off, pad8(c_))
            return plan.new_act(B, h_, w_, c_)

        plan.call(plan.fwd, _zero_tensor, plan.out7)
        if bn_train and plan.stats_xacc:                          # exact accumulators of the forward statistics (engine.fold_forward_xstats): one memset per forward
            plan._xacc_arena = torch.zeros(plan.stats_xacc_words, dtype=torch.int64, device=device)
            plan.keep.append(plan._xacc_arena)
            plan.call(plan.fwd, _zero_tensor, plan._xacc_arena)
        outs = [None] * n
        recs = []
        cur = xin
        slope = float(self.hyperparams["leaky_slope"])
        act_code = ACT_LEAKY if self.conv_activation == "leaky" else (ACT_RELU if self.conv_activation == "ReLU" else ACT_NONE)
        heads = []
        fused_into = {}
        rows_total = 0
        for i, d in enumerate(defs):
            if d["type"] == "yolo":
                rows_total += mods[i][0].num_anchors * shp[i][1] * shp[i][2]
        row_off = 0
        if not with_targets:
            plan.eval_out = torch.zeros(B, rows_total, 5 + self.num_classes, dtype=torch.float32, device=device)
        nbt = []
        for i, d in enumerate(defs):
            k = d["type"]
            if k == "convolutional":
                conv = mods[i][0]
                has_bn = d["filters"] != "preyolo"
                # a 1x1 conv right behind a BatchNorm-apply takes that pass into its operand load (engine.emit_pw_fwd): decided before
                # emit_pack so that the layer mark points at the fused launch
                pw_lb = None
                if bn_train and d["filters"] != "preyolo":
                    pw_lb = plan.pw_fwd_candidate((conv.out_channels, conv.in_channels, conv.kernel_size[0], conv.kernel_size[1],
                                                   conv.stride[0], conv.padding[0]), cur.act)
                    if pw_lb is not None:
                        plan.fwd.pop()                       # that bn_act_fwd entry is replaced by the fused launch below
                cs = ConvSpec(plan, conv.weight, conv.bias, conv.stride[0], conv.padding[0], 1, cin_pad=cur.act.C)
                plan.emit_pack(cs, need_dgrad=with_targets and cur.needs_grad)
                ho, wo = shp[i][1], shp[i][2]
                fold_c = None
                if has_bn:
                    bn = mods[i][1]
                    bs = BnSpec(plan, bn)
                    y = plan.new_act(B, ho, wo, conv.out_channels)
                    fuse = (i + 1 < n and defs[i + 1]["type"] == "shortcut" and users[i] == [i + 1]
                            and res(i + 1, int(defs[i + 1]["from"])) != i)
                    first2 = (bn_train and pw_lb is None and not fuse and plan.first_conv_2pass and cs.bias is None and plan.dtype == _lib.BF16 and
                              bool(L.first_conv_ok(plan.cdt, B, cur.act.H, cur.act.W, cs.cin_pad, cs.cout_pad, cs.kh, cs.kw, cs.stride, cs.pad, cs.dil,
                                                   cur.act.ldc)))
                    if first2:
                        # the HBM-bound first conv (25 GFLOP, 88 MB in, 354 MB out at 416^2 x 32): statistics from one streaming pass over x, then
                        # y AND z = act(BatchNorm(y)) from a second one -- the layer's output is never re-read (csrc/first_conv.hip)
                        rows = int(L.first_conv_rows(B, ho))
                        partial = plan.f32(rows * 2 * y.C, zero=False)
                        plan.call(plan.fwd, L.first_conv_stats, plan.cdt, cur.act.ptr, cur.act.ldc, cs.wf.data_ptr(), partial.data_ptr(), B, cur.act.H, cur.act.W)
                        plan.emit_bn_stats(bs, y, partial, rows)
                        nbt.append(bn.num_batches_tracked)
                        z = TNode(out_act(i), name="conv%d" % i)
                        plan.call(plan.fwd, L.first_conv_bn_act, plan.cdt, cur.act.ptr, cur.act.ldc, cs.wf.data_ptr(), bs.scale.data_ptr(), bs.shift.data_ptr(),
                                  act_code, slope, y.ptr, y.ldc, z.act.ptr, z.act.ldc, B, cur.act.H, cur.act.W)
                        plan.last_bnact = None
                        plan.first_conv_fwd2 = True
                        recs.append(("convbn", cs, bs, cur, y, z, None))
                        outs[i] = z
                        cur = z
                        continue
                    if bn_train and pw_lb is not None:
                        rows = int(L.pw_rows(cur.act.M, cs.cin_pad))
                        partial = plan.f32(rows * 2 * y.C, zero=False)
                        plan.emit_pw_fwd(pw_lb, cs, cur.act, y, partial)
                        fold_c = [plan.fwd[-1]]
                        plan.emit_bn_stats(bs, y, partial, rows)
                        fold_c += [plan.fwd[-1], cs, cur.act, y, bs, partial, rows]
                        nbt.append(bn.num_batches_tracked)
                    elif bn_train:
                        rows = plan.stats_rows(cs, cur.act, y)
                        partial = plan.f32(rows * 2 * y.C, zero=False)
                        plan.emit_conv_fwd(cs, cur.act, y, partial)
                        fold_c = [plan.fwd[-1]]
                        plan.emit_bn_stats(bs, y, partial, rows)
                        fold_c += [plan.fwd[-1], cs, cur.act, y, bs, partial, rows]
                        nbt.append(bn.num_batches_tracked)
                    else:
                        one_launch = _EVAL_FUSE and not with_targets       # inference: BN + activation in the conv's store path
                        if not one_launch:
                            plan.emit_conv_fwd(cs, cur.act, y)
                            plan.emit_bn_eval(bs)
                    if fuse:
                        rnode = outs[res(i + 1, int(defs[i + 1]["from"]))]
                        z = TNode(out_act(i + 1), name="short%d" % (i + 1))
                        if not bn_train and one_launch:
                            plan.emit_conv_bn_act_eval(cs, bs, cur.act, z.act, act_code, slope, resid=rnode.act)
                        else:
                            plan.emit_bn_act_fwd(y, bs, z.act, act_code, slope, resid=rnode.act)
                            if fold_c:
                                plan.note_stats_fold(fold_c[0], fold_c[1], plan.fwd[-1], *fold_c[2:], z.act, act_code, slope, rnode.act)
                        fused_into[i + 1] = z
                        recs.append(("convbn", cs, bs, cur, y, z, rnode))
                        outs[i] = None
                    else:
                        z = TNode(out_act(i), name="conv%d" % i)
                        if not bn_train and one_launch:
                            plan.emit_conv_bn_act_eval(cs, bs, cur.act, z.act, act_code, slope)
                        else:
                            plan.emit_bn_act_fwd(y, bs, z.act, act_code, slope)
                            if fold_c:
                                plan.note_stats_fold(fold_c[0], fold_c[1], plan.fwd[-1], *fold_c[2:], z.act, act_code, slope, None)
                        recs.append(("convbn", cs, bs, cur, y, z, None))
                        outs[i] = z
                    cur = z
                else:
                    y = TNode(out_act(i), name="logits%d" % i)
                    plan.emit_conv_fwd(cs, cur.act, y.act)
                    recs.append(("convlin", cs, cur, y))
                    outs[i] = y
                    cur = y
            elif k == "shortcut":
                if i in fused_into:
                    outs[i] = fused_into[i]
                else:
                    a, b = outs[i - 1], outs[res(i, int(d["from"]))]
                    z = TNode(out_act(i), name="short%d" % i)
                    plan.call(plan.fwd, L.bn_act_fwd, dt, a.act.ptr, a.act.ldc, None, None, None, 0, None, None, b.act.ptr, b.act.ldc,
                              z.act.ptr, z.act.ldc, z.act.M, z.act.C, ACT_NONE, 0.0)
                    recs.append(("shortcut", a, b, z))
                    outs[i] = z
                cur = outs[i]
            elif k == "maxpool":
                st_ = int(d["stride"])
                z = TNode(out_act(i), name="pool%d" % i)
                a = cur.act
                idx = torch.empty(z.act.M * z.act.C, dtype=torch.uint8, device=device)
                plan.keep.append(idx)
                ks = int(d["size"])
                if ks == 2 and st_ in (1, 2):            # the pools of yolo_baseline_tiny.cfg
                    plan.call(plan.fwd, L.maxpool2x2_fwd, dt, a.ptr, a.ldc, z.act.ptr, z.act.ldc, idx.data_ptr(), B, a.H, a.W, a.C, st_)
                    recs.append(("maxpool", cur, z, idx, st_, 0))
                else:
                    plan.call(plan.fwd, L.maxpool_fwd, dt, a.ptr, a.ldc, z.act.ptr, z.act.ldc, idx.data_ptr(), B, a.H, a.W, a.C, ks, st_, (ks - 1) // 2)
                    recs.append(("maxpool", cur, z, idx, st_, ks))
                outs[i] = z
                cur = z
            elif k == "upsample":
                sc = int(d["stride"])
                z = TNode(out_act(i), name="up%d" % i)
                a = cur.act
                if sc == 2:
                    plan.call(plan.fwd, L.upsample2x_fwd, dt, a.ptr, a.ldc, z.act.ptr, z.act.ldc, B, a.H, a.W, a.C)
                else:
                    plan.call(plan.fwd, L.upsample_fwd, dt, a.ptr, a.ldc, z.act.ptr, z.act.ldc, B, a.H, a.W, a.C, sc)
                recs.append(("upsample", cur, z, sc))
                outs[i] = z
                cur = z
            elif k == "route":
                src = [res(i, int(t)) for t in d["layers"].split(",")]
                if len(src) == 1:
                    outs[i] = outs[src[0]]
                else:
                    par = parents[i]
                    z = TNode(par, name="route%d" % i)
                    off = 0
                    parts = []
                    for s in src:
                        sn = outs[s]
                        sl = par.slice(off, sn.act.C)
                        inplace = dest.get(s, (None, None))[0] is par and sn.act.ptr == sl.ptr
                        if not inplace:                      # fallback: explicit copy into the slice
                            plan.call(plan.fwd, L.bn_act_fwd, dt, sn.act.ptr, sn.act.ldc, None, None, None, 0, None, None, None, 0,
                                      sl.ptr, sl.ldc, sl.M, sl.C, ACT_NONE, 0.0)
                        parts.append((sn, off))
                        off += sn.act.C
                    recs.append(("concat", parts, z))
                    outs[i] = z
                cur = outs[i]
            elif k == "yolo":
                yl = mods[i][0]
                lg = cur
                Gh, Gw = lg.act.H, lg.act.W
                anchors = yl.scaled_anchors(Gh).to(device)
                plan.keep.append(anchors)
                A, C = yl.num_anchors, yl.num_classes
                if with_targets:
                    ws = torch.zeros(int(L.yolo_head_workspace_bytes(B, A, Gh, Gw)), dtype=torch.uint8, device=device)
                    plan.keep.append(ws)
                    geo = (B, T, A, C, Gh, Gw, float(yl.ignore_thres), float(yl.xy_loss), float(yl.wh_loss), float(yl.object_loss),
                           float(yl.no_object_loss))
                    plan.call(plan.fwd, L.yolo_head_train, dt, lg.act.ptr, lg.act.ldc, None, 0, lg.act.C, plan.targets.data_ptr(),
                              anchors.data_ptr(), *geo, ws.data_ptr(), plan.out7.data_ptr(), None)
                    plan.err_views = tuple(plan.err_views) + (_head_err_view(ws, B, A, Gh, Gw),)
                    recs.append(("yolo", lg, anchors, ws, geo))
                else:
                    plan.call(plan.fwd, L.yolo_head_decode, dt, lg.act.ptr, lg.act.ldc, anchors.data_ptr(), float(yl.stride_for(Gh)), B, A, C,
                              Gh, Gw, plan.eval_out.data_ptr(), rows_total, row_off)
                    row_off += A * Gh * Gw
                heads.append(i)
                outs[i] = cur
        if bn_train and nbt:
            plan.call(plan.fwd, _bump_counters, nbt)
        if plan.stats_xacc:
            plan.fold_forward_xstats()     # conv -> finalize -> apply triples that survived the peepholes become two launches (csrc/exact_acc.h)
        plan.finish_pack(0)

        # ---- backward list: mirror of the records, consumers before producers
        if with_targets:
            for r in reversed(recs):
                plan.mark_ready()
                kind = r[0]
                if kind == "yolo":
                    _, lg, anchors, ws, geo = r
                    out, add = plan.grad_target(lg)
                    assert add is None
                    plan.call(plan.bwd, L.yolo_head_grad, dt, lg.act.ptr, lg.act.ldc, out.ptr, out.ldc, out.C, plan.targets.data_ptr(),
                              anchors.data_ptr(), *geo, ws.data_ptr(), plan.gscale.data_ptr())
                elif kind == "convlin":
                    _, cs, xn, y = r
                    if y.gstate == "none":
                        continue
                    plan.emit_bias_grad(cs, y.grad)
                    plan.emit_conv_bwd(cs, xn, y.act, y.grad)
                elif kind == "convbn":
                    _, cs, bs, xn, y, z, rnode = r
                    if z.gstate == "none":
                        continue
                    if rnode is not None:
                        plan.grad_identity(rnode, z.grad)
                    if not plan.emit_first_conv_bwd(z.grad, y, bs, act_code, slope, cs, xn):
                        dy = plan.emit_bn_act_bwd(z.grad, y, bs, act_code, slope)
                        plan.emit_conv_bwd(cs, xn, y, dy)
                elif kind == "shortcut":
                    _, a, b, z = r
                    if z.gstate == "none":
                        continue
                    plan.grad_identity(a, z.grad)
                    plan.grad_identity(b, z.grad)
                elif kind == "upsample":
                    _, xn, z, sc = r
                    if z.gstate == "none":
                        continue
                    out, add = plan.grad_target(xn)
                    up_bwd = (L.upsample2x_bwd, ()) if sc == 2 else (L.upsample_bwd, (sc,))
                    if add is None:
                        plan.call(plan.bwd, up_bwd[0], dt, z.grad.ptr, z.grad.ldc, out.ptr, out.ldc, B, xn.act.H, xn.act.W, xn.act.C, *up_bwd[1])
                    else:
                        tmp = plan.new_act(B, xn.act.H, xn.act.W, xn.act.C)
                        plan.call(plan.bwd, up_bwd[0], dt, z.grad.ptr, z.grad.ldc, tmp.ptr, tmp.ldc, B, xn.act.H, xn.act.W, xn.act.C, *up_bwd[1])
                        plan.call(plan.bwd, L.bn_act_fwd, dt, tmp.ptr, tmp.ldc, None, None, None, 0, None, None, add.ptr, add.ldc,
                                  out.ptr, out.ldc, out.M, out.C, ACT_NONE, 0.0)
                elif kind == "maxpool":
                    _, xn, z, idx, st_, ks = r
                    if z.gstate == "none":
                        continue
                    out, add = plan.grad_target(xn)
                    tgt = out if add is None else plan.new_act(B, xn.act.H, xn.act.W, xn.act.C)
                    if ks == 0:
                        plan.call(plan.bwd, L.maxpool2x2_bwd, dt, z.grad.ptr, z.grad.ldc, idx.data_ptr(), tgt.ptr, tgt.ldc, B, xn.act.H, xn.act.W,
                                  xn.act.C, st_)
                    else:
                        plan.call(plan.bwd, L.maxpool_bwd, dt, z.grad.ptr, z.grad.ldc, idx.data_ptr(), tgt.ptr, tgt.ldc, B, xn.act.H, xn.act.W,
                                  xn.act.C, ks, st_, (ks - 1) // 2)
                    if add is not None:
                        plan.call(plan.bwd, L.bn_act_fwd, dt, tgt.ptr, tgt.ldc, None, None, None, 0, None, None, add.ptr, add.ldc,
                                  out.ptr, out.ldc, out.M, out.C, ACT_NONE, 0.0)
                elif kind == "concat":
                    _, parts, z = r
                    if z.gstate == "none":
                        continue
                    for sn, off in parts:
                        plan.grad_identity(sn, z.grad.slice(off, sn.act.C))
            plan.mark_ready()
        plan.outs = outs
        return plan

    # ------------------------------------------------------------------------------------------ darknet .weights I/O
    def load_weights(self, weights_path, start_weight_dim):
        """Binary layout (reference models.py:339-397): int32[5] header, then for every conv in cfg order
        BN bias, BN weight, running_mean, running_var, conv weight (OIHW)  |  head: bias, weight — read from a tensor that is
        `start_weight_dim[head]` filters wide, of which the first num_filters are kept."""
        with open(weights_path, "rb") as fp:
            header = np.fromfile(fp, dtype=np.int32, count=5)
            blob = np.fromfile(fp, dtype=np.float32)
        self.header_info = header
        self.seen = header[3]
        pos, head = 0, 0
        self._param_sync()
        self._params_changed()

        def take(dst, count=None):
            nonlocal pos
            cnt = dst.numel() if count is None else count
            dst.data.copy_(torch.from_numpy(blob[pos:pos + cnt].copy()).view_as(dst))
            pos += cnt
        with torch.no_grad():
            for d, m in zip(self.module_defs, self.module_list):
                if d["type"] != "convolutional":
                    continue
                conv = m[0]
                if d["filters"] != "preyolo":
                    bn = m[1]
                    take(bn.bias); take(bn.weight); take(bn.running_mean); take(bn.running_var)
                    take(conv.weight)
                else:
                    wide = start_weight_dim[head]
                    head += 1
                    nb = conv.bias.numel()
                    conv.bias.data.copy_(torch.from_numpy(blob[pos:pos + nb].copy()))
                    pos += wide
                    per = conv.weight.numel() // nb
                    full = torch.from_numpy(blob[pos:pos + per * wide].copy()).view(wide, *conv.weight.shape[1:])
                    conv.weight.data.copy_(full[:nb])
                    pos += per * wide

    def save_weights(self, path, cutoff=-1):
        """Byte-compatible darknet .weights (reference models.py:400-422).  Under torchrun (parallel.enable_auto_data_parallel) every rank reaches
        this call of the unchanged script with identical parameters: rank 0 alone writes -- to a temporary name, renamed when complete --, and all
        ranks leave together, so a rank that loads the file right afterwards reads whole bytes (until round 5 N ranks truncated and rewrote the
        same path concurrently)."""
        from ..parallel import auto_is_writer, auto_barrier
        self._param_sync()
        if not auto_is_writer():
            auto_barrier()
            return
        tmp_path = f"{path}.tmp.{os.getpid()}"
        try:
            self._write_weights(tmp_path, cutoff)
            os.replace(tmp_path, path)
        finally:
            if os.path.exists(tmp_path):
                os.remove(tmp_path)
            auto_barrier()                                   # (also when the write failed: the other ranks are waiting in theirs)

    def _write_weights(self, path, cutoff):
        with open(path, "wb") as fp:
            self.header_info[3] = self.seen
            np.asarray(self.header_info, dtype=np.int32).tofile(fp)
            for d, m in zip(self.module_defs[:cutoff], self.module_list[:cutoff]):
                if d["type"] != "convolutional":
                    continue
                conv = m[0]
                if d["filters"] != "preyolo":
                    bn = m[1]
                    for t in (bn.bias, bn.weight, bn.running_mean, bn.running_var):
                        t.data.cpu().numpy().tofile(fp)
                else:
                    conv.bias.data.cpu().numpy().tofile(fp)
                conv.weight.data.cpu().numpy().tofile(fp)


def _sync_before_state_dict(module, prefix, keep_vars):
    module._param_sync()


def _zero_tensor(t, stream):
    t.zero_()
    return 0


def _bump_counters(ts, stream):
    torch._foreach_add_(ts, 1)
    return 0
