"""Static execution plans over libmdcv_hip.so.

A network (Darknet cfg or KeypointNet) is lowered ONCE per (batch shape, mode) into two flat launch lists — forward and
backward — over statically allocated NHWC buffers in HBM (288 GB per MI355X: nothing is recycled, every activation,
raw conv output and gradient keeps its own buffer).  Executing a step is a loop over pre-bound C calls on the current
HIP stream, or one hipGraph replay of the captured list.  PyTorch is used only for device memory and streams.

Gradient routing (fan-out, residual adds, route-concat slices) is resolved at plan-build time:
  * a tensor whose gradient is just another tensor's gradient (shortcut branch, concat slice) aliases that buffer;
  * the first computed contribution writes the buffer, later ones accumulate through the conv kernel's `addsrc`
    epilogue (out = GEMM + addsrc), so residual/fan-out sums cost no extra pass over HBM.
"""
import os

import torch

from . import _lib
from ._lib import F32, BF16, ACT_NONE, ACT_LEAKY, ACT_RELU, tuned  # noqa: F401


_POISON = os.environ.get("MDCV_POISON", "0") == "1"     # debug: NaN-fill every uninitialised plan buffer (finds reads of unwritten memory)
_REDZONE = int(os.environ.get("MDCV_REDZONE", "0"))     # debug: this many sentinel bytes behind every plan buffer; Plan.check_redzones()
_RZ_BYTE = 0xA5


def _alloc(n, dtype, device, zero, plan):
    """Plan buffer of n elements; in redzone mode it is the front of a larger allocation whose tail holds a byte pattern."""
    if not _REDZONE:
        t = (torch.zeros if zero else torch.empty)(n, dtype=dtype, device=device)
        if not zero and _POISON and t.is_floating_point():
            t.fill_(float("nan"))
        return t
    es = torch.empty(0, dtype=dtype).element_size()
    raw = torch.full((n * es + _REDZONE,), _RZ_BYTE, dtype=torch.uint8, device=device)
    t = raw[:n * es].view(dtype)
    if zero:
        t.zero_()
    plan.__dict__.setdefault("redzones", []).append((raw, n * es))
    return t


def pad8(c):
    return (c + 7) // 8 * 8


# One side stream per device, shared by every plan of every model, and CHECKED to run beside the stream it is used next to.  HIP multiplexes
# its streams onto a few hardware queues (4 by default): a pool stream that lands on the queue of the current stream never overlaps with it.
# Found in round 3 (scripts/bimodal_probe.py): identical models built one after another in one process took 14.5 ms per step -- or 17.0 ms, with
# the per-kernel times of a SERIAL step, whenever their plan's fresh torch.cuda.Stream() shared the main stream's queue (about one model in
# eight).  The check: two 1-block spin kernels of ~0.5 ms, one per stream; side by side they take one kernel's time, on one queue two.
_SIDE_STREAMS = {}


def _runs_beside(others, cand):
    """True if a kernel on `cand` runs concurrently with a kernel on each stream of `others` (one at a time)."""
    cycles = 1 << 20
    for cur in others:
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(cur):
            torch.cuda._sleep(cycles)                  # warm the kernel
            cur.synchronize()
            e0.record(cur)
            torch.cuda._sleep(cycles)
            e1.record(cur)                             # e0..e1: one spin kernel alone
            cand.wait_stream(cur)
            with torch.cuda.stream(cand):
                torch.cuda._sleep(cycles)
            torch.cuda._sleep(cycles)
            cur.wait_stream(cand)
            e2.record(cur)                             # e1..e2: one on each stream
        e2.synchronize()
        if not e1.elapsed_time(e2) < 1.5 * e0.elapsed_time(e1):
            return False
    return True


def checked_stream(device, beside, key):
    """A stream of `device` that was CHECKED to run beside every stream in `beside`; cached per (device, key, those streams)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    k = (idx, key) + tuple(s.cuda_stream for s in beside)
    s = _SIDE_STREAMS.get(k)
    if s is None:
        if torch.cuda.is_current_stream_capturing():   # the probe synchronises: never inside a graph capture
            return torch.cuda.Stream(device=device)
        ok = False
        with torch.cuda.device(device):
            for _ in range(16):                        # torch hands out its pool streams round-robin: a few tries walk the hardware queues
                s = torch.cuda.Stream(device=device)
                if _runs_beside(beside, s):
                    ok = True
                    break
        if ok:
            _SIDE_STREAMS[k] = s
        else:
            # a busy or shared GPU can fail the 1.5x timing test for every candidate: use the last one for now, say so, and probe
            # again on the next call instead of caching a stream known to have failed
            import warnings
            warnings.warn(f"mdcv: no stream of {device} was measured to run beside {key!r}'s partners (16 candidates; busy or shared GPU, "
                          "or too few hardware queues -- GPU_MAX_HW_QUEUES=8 gives HIP more): overlap of this stream is not guaranteed",
                          RuntimeWarning, stacklevel=2)
    return s


def side_stream(device):
    """The device's side stream for work that runs BESIDE the current stream (weight gradients of a backward pass)."""
    return checked_stream(device, [torch.cuda.current_stream(device)], "side")


def parse_precision(p):
    if p in (BF16, "bf16", torch.bfloat16):
        return BF16
    if p in (F32, "fp32", "f32", torch.float32):
        return F32
    raise ValueError(f"precision must be 'bf16' or 'fp32', got {p!r}")


class Act:
    """NHWC activation view: [B,H,W,C] with channel stride ldc inside `buf` (C already padded to a multiple of 8)."""
    __slots__ = ("buf", "B", "H", "W", "C", "ldc", "off", "ptr")

    def __init__(self, buf, B, H, W, C, ldc, off=0):
        self.buf, self.B, self.H, self.W, self.C, self.ldc, self.off = buf, B, H, W, C, ldc, off
        self.ptr = buf.data_ptr() + off * buf.element_size()

    @property
    def M(self):
        return self.B * self.H * self.W

    def slice(self, off, C):
        return Act(self.buf, self.B, self.H, self.W, C, self.ldc, self.off + off)

    def dense(self):
        """[B,H,W,C] strided torch view (debug / tests)."""
        return self.buf.view(self.B, self.H, self.W, self.ldc)[..., self.off:self.off + self.C]


class TNode:
    """Activation tensor in the plan + the state of its gradient while the backward list is being built."""
    __slots__ = ("act", "gstate", "grad", "needs_grad", "name")

    def __init__(self, act, needs_grad=True, name=""):
        self.act, self.gstate, self.grad, self.needs_grad, self.name = act, "none", None, needs_grad, name


class ConvSpec:
    """One nn.Conv2d: parameters + geometry + packed operand buffers."""

    def __init__(self, plan, weight, bias, stride, pad, dil, cin_pad=None):
        self.weight, self.bias = weight, bias
        self.cout, self.cin, self.kh, self.kw = weight.shape
        self.stride, self.pad, self.dil = stride, pad, dil
        self.cout_pad = pad8(self.cout)
        self.cin_pad = cin_pad if cin_pad is not None else pad8(self.cin)
        self.ktot = self.kh * self.kw * self.cin_pad
        dt = plan.tdtype
        self.wf = _alloc(self.cout_pad * self.ktot, dt, plan.device, True, plan)
        self.wd = None
        self.bias_pad = None
        if bias is not None:
            self.bias_pad = _alloc(self.cout_pad, torch.float32, plan.device, True, plan)
        plan.keep.append(self)     # the launch lists hold raw device pointers: the plan must own every buffer

    def out_hw(self, H, W):
        ek = self.dil * (self.kh - 1) + 1
        return (H + 2 * self.pad - ek) // self.stride + 1, (W + 2 * self.pad - ek) // self.stride + 1


class BnSpec:
    def __init__(self, plan, bn):
        self.bn = bn
        C = bn.num_features
        self.C = C
        dev = plan.device
        z = lambda n=C: _alloc(n, torch.float32, dev, True, plan)  # noqa: E731
        self.scale, self.shift, self.mean, self.invstd = z(), z(), z(), z()
        self.cA, self.cB, self.cC = z(), z(), z()
        self.accum = _alloc(3 * C, torch.float64, dev, True, plan)
        plan.keep.append(self)


def _variant_env(name):
    """ONE signed variant code per kernel family (csrc/tune.h); the comma lists the removed set_variant hooks took are refused with a clear message
    instead of a ValueError from int() at import time."""
    raw = (os.environ.get(name) or "0").strip()
    try:
        return int(raw)
    except ValueError:
        raise ValueError(f"{name}={raw!r}: one signed integer variant code per kernel family is carried by every call since round 5 "
                         "(csrc/tune.h lists them); comma-separated lists are no longer accepted") from None


class Plan:
    tune_conv = _variant_env("MDCV_CONV_VARIANT")       # variant code of the conv family for this plan's calls (0: defaults)
    tune_wgrad = _variant_env("MDCV_WGRAD_VARIANT")     # ... of the weight-gradient family

    def __init__(self, device, precision, training, grad_sink=None):
        _lib.require_gpu()
        self.L = _lib.lib()
        self.device = device
        self.dtype = parse_precision(precision)
        self.tdtype = torch.bfloat16 if self.dtype == BF16 else torch.float32
        # per-call tuning (csrc/tune.h): the plan hands its variant codes to every convolution / weight-gradient call through the dtype argument --
        # the library keeps no tuning state, so two plans of one process may differ (A/B runs: scripts/ab_step.py "c<code>" / "w<code>")
        self.cdt = tuned(self.dtype, self.tune_conv)
        self.wdt = tuned(self.dtype, self.tune_wgrad)
        self.training = training
        self.fwd, self.bwd = [], []
        self.keep = []                 # keeps every buffer alive
        self.grad_of = {}              # id(param) -> fp32 gradient tensor the backward list writes
        self.grad_sink = grad_sink     # callable(param) -> preallocated grad tensor (flat buffer view) or None
        self.ws_floats = 0             # shared wgrad slab scratch (max over layers)
        self._ws = None
        self.bytes = 0
        self.graph_fwd = self.graph_bwd = None
        self.pack_list = []
        self.__dict__.setdefault("redzones", [])       # (raw uint8 allocation, payload bytes) in MDCV_REDZONE mode
        self.layer_marks = []              # forward-list position where each packed layer's launches begin
        self.param_groups = None           # pipelined parameter update (optim.py): [(lo, hi, first_layer, nlayers)] in forward order
        self._group_events = []            # one "parameters + packed operands of group k are final" event per group, or None
        self._pending_updates = []         # deferred group updates (closures), launched from the forward list
        self._packed_ahead = False         # the optimizer already re-packed every layer for the coming forward
        self._packed_version = -1
        self._packed_tversion = -1
        self.owner = None                  # the model (FlatParamsMixin) whose parameters this plan reads
        self.grad_offset = None            # callable(param) -> offset in the flat gradient buffer
        self.low_water = 1 << 62
        self._last_mark = 1 << 62
        self.on_ready = None               # set per backward by the data-parallel reducer
        self.dgrad_entries = {}            # grad buffer ptr -> backward-list entry of the data gradient that wrote it last
        self.fused_bn = 0                  # BatchNorm backward reductions folded into data-gradient store loops
        self._fold_candidates = []
        self._xacc_arena, self.stats_xfolded = None, 0

    # ------------------------------------------------------------------ buffers
    def new_act(self, B, H, W, C, zero=False):
        Cp = pad8(C)
        buf = _alloc(B * H * W * Cp, self.tdtype, self.device, zero, self)
        self.keep.append(buf)
        self.bytes += buf.numel() * buf.element_size()
        return Act(buf, B, H, W, Cp, Cp)

    def f32(self, n, zero=True):
        t = _alloc(n, torch.float32, self.device, zero, self)
        self.keep.append(t)
        return t

    def param_grad(self, p):
        g = self.grad_of.get(id(p))
        if g is None:
            g = self.grad_sink(p) if self.grad_sink else None
            if g is None:
                g = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
            self.grad_of[id(p)] = g
        if self.grad_offset is not None:                 # lowest flat-buffer offset written so far by the backward list
            self.low_water = min(self.low_water, self.grad_offset(p))
        return g

    def mark_ready(self):
        """Backward-list marker: every gradient at flat offset >= low_water has been ENQUEUED at this point (layers are
        processed last-to-first and the flat buffer is in parameter order).  The data-parallel reducer uses it to start
        all-reducing finished buckets while the rest of backward still runs."""
        lw = self.low_water
        if lw >= self._last_mark:
            return
        self._last_mark = lw
        plan = self

        def ready(stream, lw=lw):
            if plan.on_ready is not None:
                plan.on_ready(lw)
            return 0
        ready.__name__ = "grad_ready"
        ready.low_water = lw
        self.bwd.append((ready, ()))

    def wgrad_ws(self, stream=0):
        """The split-K slab scratch of the weight gradients launched on `stream` (kernels of one stream run in order and may share it)."""
        if self._ws is None:
            self._ws = {}
        w = self._ws.get(stream)
        if w is None or w.numel() < self.ws_floats:
            w = self._ws[stream] = _alloc(max(self.ws_floats, 1), torch.float32, self.device, False, self)
        return w

    # ------------------------------------------------------------------ launch lists
    def call(self, lst, fn, *args):
        lst.append((fn, args))

    def run(self, lst, stream=None):
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        for fn, args in lst:
            rc = fn(*args, stream)
            if rc:
                raise _lib.MdcvError(f"{getattr(fn, '__name__', fn)} returned {rc}")

    # ------------------------------------------------------------------ gradient routing
    def grad_target(self, t):
        """Buffer a COMPUTED contribution to d(t) must write, plus the tensor it must add (None, alias or itself)."""
        if t.gstate == "none":
            t.grad = self._alloc_like(t.act)
            t.gstate = "own"
            return t.grad, None
        if t.gstate == "alias":
            src = t.grad
            t.grad = self._alloc_like(t.act)
            t.gstate = "own"
            return t.grad, src
        return t.grad, t.grad

    def grad_identity(self, t, src):
        """d(t) += src where src is an existing gradient buffer (shortcut branch / concat slice)."""
        if not t.needs_grad:
            return
        if t.gstate == "none":
            t.grad, t.gstate = src, "alias"
            return
        if t.gstate == "alias":
            a = t.grad
            t.grad = self._alloc_like(t.act)
            t.gstate = "own"
        else:
            a = t.grad
        o = t.grad
        self.call(self.bwd, self.L.bn_act_fwd, self.dtype, a.ptr, a.ldc, None, None, None, 0, None, None, src.ptr, src.ldc,
                  o.ptr, o.ldc, o.M, o.C, ACT_NONE, 0.0)

    def _alloc_like(self, a):
        return self.new_act(a.B, a.H, a.W, a.C)

    # ------------------------------------------------------------------ op emitters
    def emit_input(self, B, C, H, W):
        """NCHW fp32 user tensor -> NHWC T (channels zero-padded to 8).  Returns (TNode, setter)."""
        a = self.new_act(B, H, W, C)
        node = TNode(a, needs_grad=False, name="input")
        holder = {"src": None}
        L, dt = self.L, self.dtype

        def convert(stream):
            x = holder["src"]
            return L.nchw_to_nhwc(dt, x.data_ptr(), a.ptr, B, C, H, W, a.ldc, a.C, stream)
        convert.__name__ = "nchw_to_nhwc"
        self.fwd.append((convert, ()))
        return node, holder

    def emit_pack(self, cs, need_dgrad):
        """Registers the conv for the per-step weight re-layout; all layers are packed by ONE table-driven launch that
        `finish_pack()` places at the head of the forward list."""
        if need_dgrad and cs.wd is None:
            cs.wd = _alloc(cs.cin_pad * cs.kh * cs.kw * cs.cout_pad, self.tdtype, self.device, True, self)
        self.pack_list.append(cs)
        self.layer_marks.append(len(self.fwd))
        # (the fp32 bias goes to its padded operand buffer inside the same table-driven pack launch)

    def stats_rows(self, cs, x, y):
        """rows of the BatchNorm partial-statistics buffer the forward conv of this geometry writes"""
        return int(self.L.conv2d_stats_rows_geom(self.cdt, x.B, y.H, y.W, cs.cin_pad, cs.cout_pad, cs.kh, cs.kw, cs.stride, cs.pad,
                                                 cs.dil, x.ldc))

    def emit_conv_fwd(self, cs, x, y, stats_partial=None):
        """x, y: Act.  y.C == cs.cout_pad."""
        assert x.C == cs.cin_pad and y.C == cs.cout_pad, (x.C, cs.cin_pad, y.C, cs.cout_pad)
        self.call(self.fwd, self.L.conv2d, self.cdt, 0, x.ptr, x.ldc, cs.wf.data_ptr(), y.ptr, y.ldc,
                  cs.bias_pad.data_ptr() if cs.bias_pad is not None else None, None, 0,
                  stats_partial.data_ptr() if stats_partial is not None else None,
                  x.B, x.H, x.W, cs.cin_pad, y.H, y.W, cs.cout_pad, cs.kh, cs.kw, cs.stride, cs.pad, cs.dil)

    def emit_conv_bwd(self, cs, xnode, y_shape_act, dy, x_wgrad=None):
        """dy: Act gradient of the raw conv output.  Emits wgrad (+bias grad) and, if the input needs it, dgrad.
        x_wgrad: a copy of the conv's input with a wider channel padding for the weight gradient only (the 7x7 stem's LDS-ring
        kernel wants 16-channel rows; forward keeps the 8-channel buffer)."""
        L, dt = self.L, self.dtype
        x = xnode.act
        if x_wgrad is None and self._emit_pw_bwd1(cs, xnode, dy):
            return
        xw = x_wgrad if x_wgrad is not None else x
        cin_w = xw.C if x_wgrad is not None else cs.cin_pad
        if self.wgrad_after_dgrad and cs.kh == 3 and cs.stride == 1:
            self._emit_dgrad(cs, xnode, dy)
            self._emit_wgrad_single(cs, xw, dy, cin_w)
        else:
            self._emit_wgrad_single(cs, xw, dy, cin_w)
            self._emit_dgrad(cs, xnode, dy)

    wgrad_after_dgrad = False          # A/B switch: a 3x3 layer's weight gradient forked BEHIND its data gradient (beside the next layers' BatchNorm passes)
                                       # instead of beside it: 13.17 -> 13.62 ms (round 5, scripts/ab_step.py "B1;W1"; round 4: 13.75 -> 14.10)

    # ---- 1x1 layers: data gradient + weight-gradient slabs in ONE launch (csrc/pw_bwd.hip).  The three launches it replaces each read dy
    # from HBM and were bound by neither MFMA nor bandwidth (3.2 ms serial for 10 % of the step's FLOPs).  Same-box timing of the
    # launch (+ slab reduce) against data gradient (+ fused sums) + weight gradient + reduce, yolo_baseline 416^2 batch 32 (scripts/pwb_ab.py):
    # 26^2 512->256 36 vs 55 us, 52^2 256->128 50 vs 71 us, 13^2 1024->512 33 vs 43 us; dx bit-identical, dW to 3e-7.
    pw_bwd1 = True                     # (tests / scripts/ab_step.py flip the class attribute; no environment knob)

    def _emit_pw_bwd1(self, cs, xnode, dy):
        """dx = dy . W (+ the other gradient contributions of x) and the slabs of dW in one launch on the main stream, the slab reduce on the
        side stream.  Returns False (nothing emitted, no state touched) when the layer does not take this form."""
        L, dt = self.L, self.dtype
        x = xnode.act
        if not self.pw_bwd1 or dt != BF16 or not xnode.needs_grad or cs.wd is None:
            return False
        if (cs.kh, cs.kw, cs.stride, cs.pad) != (1, 1, 1, 0) or x.C != cs.cin_pad or dy.C != cs.cout_pad:
            return False
        slabs = int(L.pw_bwd_slabs(dt, x.M, cs.cin_pad, cs.cout_pad, dy.ldc, x.ldc, 8, 8, 8))
        if slabs < 1:
            return False
        out, add = self.grad_target(xnode)
        assert slabs == int(L.pw_bwd_slabs(dt, x.M, cs.cin_pad, cs.cout_pad, dy.ldc, x.ldc, out.ldc, add.ldc if add is not None else 8, 8))
        gw = self.param_grad(cs.weight)
        ws = self.f32(slabs * cs.cout_pad * cs.cin_pad, zero=False)      # its own slab buffer: the reduce runs on the side stream
        args = [dt, dy.ptr, dy.ldc, x.ptr, x.ldc, cs.wd.data_ptr(), out.ptr, out.ldc, add.ptr if add is not None else None,
                add.ldc if add is not None else 0, ws.data_ptr(), slabs, None, 0, None, None, None, 0, 0.0, None, x.M, cs.cin_pad, cs.cout_pad]
        self.dgrad_entries[out.ptr] = dict(
            idx=len(self.bwd), out=out, used=False, pwb=args,
            geom=(x.B, dy.H, dy.W, cs.cout_pad, x.H, x.W, cs.cin_pad, cs.kh, cs.kw, cs.stride, cs.pad, cs.dil), head=(dy.ptr, dy.ldc))
        self.call(self.bwd, L.pw_bwd, *args)

        def wreduce(stream, ws=ws, gw=gw, slabs=slabs, cs=cs):
            return L.wgrad_reduce(ws.data_ptr(), slabs, gw.data_ptr(), 0, cs.cout_pad, cs.cout, cs.cin_pad, cs.cin, 1, stream)
        wreduce.__name__ = "conv2d_wgrad"                                  # (side stream, run_bwd_list)
        wreduce.info = (x.B, x.H, x.W, cs.cin_pad, dy.H, dy.W, cs.cout_pad, 0, 1, slabs)      # k = 0: the reduce alone, no MACs of its own
        self.bwd.append((wreduce, ()))
        self.pw_bwd1_count = getattr(self, "pw_bwd1_count", 0) + 1
        return True

    def _emit_wgrad_single(self, cs, xw, dy, cin_w):
        L, dt = self.L, self.dtype
        gw = self.param_grad(cs.weight)
        splits = int(L.conv2d_wgrad_splits_geom(self.wdt, xw.B, xw.H, xw.W, cin_w, dy.H, dy.W, cs.cout_pad, cs.kh, cs.kw, cs.stride, cs.pad,
                                                cs.dil, dy.ldc, xw.ldc))
        self.ws_floats = max(self.ws_floats, splits * cs.cout_pad * cs.kh * cs.kw * cin_w)
        plan = self

        def wgrad(stream, cs=cs, x=xw, dy=dy, gw=gw, splits=splits, cin_w=cin_w):
            return L.conv2d_wgrad(plan.wdt, dy.ptr, dy.ldc, x.ptr, x.ldc, plan.wgrad_ws(stream).data_ptr(), splits, gw.data_ptr(), 0,
                                  x.B, x.H, x.W, cin_w, cs.cin, dy.H, dy.W, cs.cout_pad, cs.cout, cs.kh, cs.kw,
                                  cs.stride, cs.pad, cs.dil, stream)
        wgrad.__name__ = "conv2d_wgrad"
        wgrad.info = (xw.B, xw.H, xw.W, cin_w, dy.H, dy.W, cs.cout_pad, cs.kh, cs.stride, splits)
        self.bwd.append((wgrad, ()))

    def _emit_dgrad(self, cs, xnode, dy):
        L, dt = self.L, self.dtype
        x = xnode.act
        if xnode.needs_grad:
            out, add = self.grad_target(xnode)
            self.dgrad_entries[out.ptr] = dict(
                idx=len(self.bwd), out=out, used=False,
                geom=(x.B, dy.H, dy.W, cs.cout_pad, x.H, x.W, cs.cin_pad, cs.kh, cs.kw, cs.stride, cs.pad, cs.dil),
                head=(dy.ptr, dy.ldc, cs.wd.data_ptr(), out.ptr, out.ldc, add.ptr if add is not None else None,
                      add.ldc if add is not None else 0))
            self.call(self.bwd, L.conv2d, self.cdt, 1, dy.ptr, dy.ldc, cs.wd.data_ptr(), out.ptr, out.ldc, None,
                      add.ptr if add is not None else None, add.ldc if add is not None else 0, None,
                      x.B, dy.H, dy.W, cs.cout_pad, x.H, x.W, cs.cin_pad, cs.kh, cs.kw, cs.stride, cs.pad, cs.dil)

    def emit_bias_grad(self, cs, dy, zero_only=False):
        gb = self.param_grad(cs.bias)
        if zero_only:   # bias in front of a BatchNorm: its gradient is identically zero.  Nothing in the backward list
            gb.zero_()  # writes this buffer, so it is zeroed once here instead of once per step (zero_grad / all-reduce keep it zero)
            return
        tmp = self.f32(cs.cout_pad)
        pws = self.f32(int(self.L.colsum_ws_floats(self.dtype, dy.M, cs.cout_pad)), zero=False)
        self.call(self.bwd, self.L.colsum_f32, self.dtype, dy.ptr, dy.ldc, dy.M, cs.cout_pad, pws.data_ptr(), tmp.data_ptr())

        def copy(stream, gb=gb, tmp=tmp, n=cs.cout):
            gb.copy_(tmp[:n].view_as(gb), non_blocking=True)
            return 0
        copy.__name__ = "copy_bias_grad"
        self.bwd.append((copy, ()))

    def emit_bn_stats(self, bs, y, partial, rows):
        """conv-epilogue partial sums -> batch statistics -> scale/shift (+ running stats)."""
        L = self.L
        bn = bs.bn
        self.call(self.fwd, L.bn_stats_finalize, partial.data_ptr(), rows, bs.accum.data_ptr(), float(y.M), bn.weight.data_ptr(),
                  bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.momentum), float(bn.eps),
                  bs.scale.data_ptr(), bs.shift.data_ptr(), bs.mean.data_ptr(), bs.invstd.data_ptr(), bs.C)

    def emit_bn_eval(self, bs):
        bn = bs.bn
        self.call(self.fwd, self.L.bn_eval_coeffs, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                  bn.running_var.data_ptr(), float(bn.eps), bs.scale.data_ptr(), bs.shift.data_ptr(), bs.C)

    def emit_conv_bn_act_eval(self, cs, bs, x, out, act, slope, resid=None):
        """Inference: conv -> BatchNorm(running statistics) -> activation (+ residual) as ONE launch: the coefficients (with the conv's
        own bias folded into the shift) are refreshed by a tiny kernel, then the conv's store path applies them
        (mdcv_conv2d_affine_act); the raw conv output never exists in HBM."""
        bn = bs.bn
        self.call(self.fwd, self.L.bn_eval_coeffs_bias, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                  bn.running_var.data_ptr(), float(bn.eps), cs.bias_pad.data_ptr() if cs.bias_pad is not None else None,
                  bs.scale.data_ptr(), bs.shift.data_ptr(), bs.C)
        self.call(self.fwd, self.L.conv2d_affine_act, self.cdt, x.ptr, x.ldc, cs.wf.data_ptr(), out.ptr, out.ldc, bs.scale.data_ptr(),
                  bs.shift.data_ptr(), resid.ptr if resid is not None else None, resid.ldc if resid is not None else 0, act, float(slope),
                  x.B, x.H, x.W, cs.cin_pad, out.H, out.W, cs.cout_pad, cs.kh, cs.kw, cs.stride, cs.pad, cs.dil)

    def emit_bn_act_fwd(self, y1, bs1, out, act, slope, y2=None, bs2=None, resid=None):
        # (remembered so that a 1x1 conv emitted right behind it can take the pass into its operand load: emit_pw_fwd)
        self.last_bnact = None if y2 is not None else dict(idx=len(self.fwd), y=y1, bs=bs1, out=out, act=act, slope=float(slope), resid=resid)
        self.call(self.fwd, self.L.bn_act_fwd, self.dtype, y1.ptr, y1.ldc, bs1.scale.data_ptr(), bs1.shift.data_ptr(),
                  y2.ptr if y2 is not None else None, y2.ldc if y2 is not None else 0,
                  bs2.scale.data_ptr() if bs2 is not None else None, bs2.shift.data_ptr() if bs2 is not None else None,
                  resid.ptr if resid is not None else None, resid.ldc if resid is not None else 0,
                  out.ptr, out.ldc, out.M, out.C, act, float(slope))

    def note_stats_fold(self, conv, fin, act, cs, x, y, bs, partial, rows, out, act_code, slope, resid):
        """Remember one conv -> statistics finalize -> BatchNorm-apply triple of the forward list (the three list entries themselves) for
        fold_forward_xstats, which runs after the whole list is built: a later peephole may still replace the apply entry."""
        self._fold_candidates.append((conv, fin, act, cs, x, y, bs, partial, rows, out, act_code, float(slope), resid))

    # ---- forward BatchNorm statistics through exact accumulators (csrc/exact_acc.h): the conv's epilogue ADDS its per-tile sums to 64-bit
    # fixed-point words with fire-and-forget integer atomics (exact, order-independent: bit-reproducible), the BatchNorm-apply pass reads the
    # totals in its prologue.  No partial rows, no finalize launch, no hand-off inside a launch.
    stats_xacc = True                  # (tests / scripts/ab_step.py flip the class attribute; no environment knob)
    stats_xacc_pw = True               # ... also for layers whose apply pass lives in the 1x1 block behind them (round 6: publisher-only finalize)
    stats_xacc_words = 1 << 20         # 64-bit words of the accumulator arena (zeroed by one memset at the head of the forward list)
    stats_xacc_chain = 1024            # most additions one word may see per launch (416^2 x 32, 43 264 rows on 32 replicas: +8 us on a 160 us launch -> keeps its rows)

    def fold_forward_xstats(self):
        """Rewrite conv (or 1x1 block) / finalize / apply triples into mdcv_conv2d_xstats (mdcv_pw_conv_fwd_xstats) + mdcv_bn_act_fwd_xstats."""
        L, dt = self.L, self.dtype
        if not self.stats_xacc or self._xacc_arena is None:
            return 0
        cands, self._fold_candidates = self._fold_candidates, []
        arena = self._xacc_arena
        used, done = 0, 0
        for conv, fin, act, cs, x, y, bs, partial, rows, out, act_code, slope, resid in cands:
            idx = [i for i, e in enumerate(self.fwd) if e is conv]
            if len(idx) != 1:
                continue
            i = idx[0]
            if i + 2 > len(self.fwd) - 1 or self.fwd[i + 1] is not fin:
                continue
            nxt = self.fwd[i + 2]
            # round 6: the apply pass of this layer may live in the operand load of the 1x1 block behind it (emit_pw_fwd popped the apply entry):
            # conv -> finalize -> pw block.  The finalize over the conv's partial rows (676 .. 2704 rows, 8 us) then becomes the PUBLISHER workgroup
            # of the accumulator form alone -- mdcv_bn_act_fwd_xstats over an empty strip (M = 0) -- and the conv writes no rows.
            pw_next = (self.stats_xacc_pw and nxt is not act and nxt[0] in (L.pw_conv_fwd, L.pw_conv_fwd_xstats) and len(nxt[1]) > 4
                       and nxt[1][3] == bs.scale.data_ptr() and nxt[1][1] == y.ptr)
            if nxt is not act and not pw_next:
                continue
            if y.C > 1024:
                continue
            reps = int(L.xstats_reps(rows, y.C))
            if rows > self.stats_xacc_chain * reps:           # same-address atomics retire at ~10-17 ns each: the launch would end on their queue
                continue
            need = int(L.xstats_words(reps, y.C))
            if used + need > arena.numel():
                continue
            acc = arena[used:used + need]
            used += need
            bn = bs.bn
            a = conv[1]
            if conv[0] is L.pw_conv_fwd:                     # (dtype, y, ldy, scale, shift, resid, ldr, act, slope, z, ldz, w, bias, out, out_ldc, stats, M, K, N)
                self.fwd[i] = (L.pw_conv_fwd_xstats, a[:15] + (acc.data_ptr(), reps) + a[16:])
            elif conv[0] is L.conv2d:                        # (dtype, mode, in, in_ldc, w, out, out_ldc, bias, addsrc, add_ldc, stats, geometry...)
                self.fwd[i] = (L.conv2d_xstats, (a[0],) + a[2:8] + (acc.data_ptr(), reps) + a[11:])
            else:
                used -= need
                continue
            if pw_next:                                      # finalize entry -> publisher-only launch; the block behind it reads scale / shift as before
                self.fwd[i + 1] = (L.bn_act_fwd_xstats, (dt, y.ptr, y.ldc, acc.data_ptr(), reps, float(y.M), bn.weight.data_ptr(), bn.bias.data_ptr(),
                                                         bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.momentum), float(bn.eps),
                                                         bs.scale.data_ptr(), bs.shift.data_ptr(), bs.mean.data_ptr(), bs.invstd.data_ptr(),
                                                         None, 0, y.ptr, y.ldc, 0, y.C, act_code, slope))
                done += 1
                continue
            self.fwd[i + 2] = (L.bn_act_fwd_xstats, (dt, y.ptr, y.ldc, acc.data_ptr(), reps, float(y.M), bn.weight.data_ptr(), bn.bias.data_ptr(),
                                                     bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.momentum), float(bn.eps),
                                                     bs.scale.data_ptr(), bs.shift.data_ptr(), bs.mean.data_ptr(), bs.invstd.data_ptr(),
                                                     resid.ptr if resid is not None else None, resid.ldc if resid is not None else 0,
                                                     out.ptr, out.ldc, out.M, out.C, act_code, slope))
            del self.fwd[i + 1]
            self.layer_marks = [m - 1 if m > i + 1 else m for m in self.layer_marks]
            done += 1
        # the head-of-list memset covers only what the layers took
        for k, e in enumerate(self.fwd):
            if len(e[1]) == 1 and e[1][0] is arena:
                view = arena[:max(used, 1)]
                self.keep.append(view)
                self.fwd[k] = (e[0], (view,))
                break
        self.stats_xfolded = done
        return done

    # ---- 1x1 conv blocks with the neighbouring BatchNorm pass folded into the operand load (csrc/pw_block.hip).  Policy from same-box
    # timing of the fused launch against the pair it replaces (scripts/pw_block_ab.py alone, per-kernel times of a serial step from bench.py --dump-launches;
    # YOLOv3 416^2 batch 32): the forward form wins where the layer's weights stay resident in LDS -- 52^2: 32.7 vs 43.3 us, 104^2: 61.5
    # vs 72.7 us -- and is level at 26^2 (25 vs 27 us alone, 29 vs 25 in the step); the backward form is level at 52^2 (48.7 vs 54.9 us
    # alone, 57 vs 54.5 in the step) and loses elsewhere (13^2, K = 1024: 16-pixel tiles re-stream 1 MiB of weights per tile).  These
    # layers are HBM-bound and the fold removes one tensor read of four to seven, so the gain is bounded by that ratio.
    pw_fuse = True                     # (tests / A-B scripts flip the class attribute; no environment knob)
    pw_fwd_px = (50000, 1 << 30)       # pixels M of the layers that take the forward form

    def _pw_ok(self, cs, M, K, N, *lds):
        if not self.pw_fuse or self.dtype != BF16 or not self.training:
            return False
        if (cs.kh, cs.kw, cs.stride, cs.pad) != (1, 1, 1, 0):
            return False
        lo, hi = self.pw_fwd_px
        if not (lo <= M < hi) or N < 64:
            return False
        if K > 512:
            return False
        if K < 64 or K % 32 or 256 % (K // 8) or N % 8 or any(v % 8 for v in lds):
            return False
        return True

    def pw_fwd_candidate(self, cs_shape, x):
        """-> the remembered bn_act_fwd emission if a 1x1 conv reading `x` emitted NOW could absorb it (it must be the last forward entry
        and have written exactly x), else None.  cs_shape = (cout, cin, kh, kw, stride, pad)."""
        lb = getattr(self, "last_bnact", None)
        if lb is None or lb["idx"] != len(self.fwd) - 1 or lb["out"].ptr != x.ptr or lb["out"].ldc != x.ldc or lb["out"].C != x.C:
            return None
        cout, cin, kh, kw, stride, pad = cs_shape

        class _S:                      # the geometry fields _pw_ok looks at
            pass
        c = _S(); c.kh, c.kw, c.stride, c.pad = kh, kw, stride, pad
        r = lb["resid"]
        if not self._pw_ok(c, x.M, x.C, pad8(cout), lb["y"].ldc, x.ldc, r.ldc if r is not None else 8):
            return None
        return lb

    def emit_pw_fwd(self, lb, cs, x, y, stats_partial):
        """Emits ONE launch for  bn_act_fwd(lb) ; conv2d(cs: x -> y)  (lb from pw_fwd_candidate; the caller has popped that bn_act_fwd
        entry from the forward list and emitted nothing else since).  stats_partial: [pw_rows][2][y.C] or None."""
        assert lb["idx"] == len(self.fwd) and x.C == cs.cin_pad and y.C == cs.cout_pad
        self.last_bnact = None
        r, bs = lb["resid"], lb["bs"]
        self.call(self.fwd, self.L.pw_conv_fwd, self.dtype, lb["y"].ptr, lb["y"].ldc, bs.scale.data_ptr(), bs.shift.data_ptr(),
                  r.ptr if r is not None else None, r.ldc if r is not None else 0, lb["act"], lb["slope"], x.ptr, x.ldc,
                  cs.wf.data_ptr(), cs.bias_pad.data_ptr() if cs.bias_pad is not None else None, y.ptr, y.ldc,
                  stats_partial.data_ptr() if stats_partial is not None else None, x.M, cs.cin_pad, cs.cout_pad)
        self.pw_fwd_count = getattr(self, "pw_fwd_count", 0) + 1

    def emit_bn_act_bwd(self, dout, y1, bs1, act, slope, y2=None, bs2=None, apply=True):
        """Returns the gradient(s) of the raw conv output(s): dy1 [, dy2].  apply=False: only the statistics part (sums + coefficients); the
        consumer forms dy itself (emit_first_conv_bwd) and nothing is returned."""
        L, dt = self.L, self.dtype
        dy1 = self._alloc_like(y1) if apply else None
        dy2 = self._alloc_like(y2) if y2 is not None else None
        n = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        pws = self.f32(L.bn_act_bwd_reduce_ws_floats(dt, y1.M, y1.C, 3 if y2 is not None else 2), zero=False)
        g1, b1 = self.param_grad(bs1.bn.weight), self.param_grad(bs1.bn.bias)
        g2 = b2 = None
        if y2 is not None:
            g2, b2 = self.param_grad(bs2.bn.weight), self.param_grad(bs2.bn.bias)
        fused = y2 is None and self._fuse_bn_sums(dout, y1, bs1, act, slope, g1, b1)
        if not fused:
            self.call(self.bwd, L.bn_act_bwd_reduce_finalize, dt, dout.ptr, dout.ldc, y1.ptr, y1.ldc, n(bs1.scale), n(bs1.shift), n(bs1.mean),
                      n(bs1.invstd), y2.ptr if y2 is not None else None, y2.ldc if y2 is not None else 0,
                      n(bs2.scale) if bs2 else None, n(bs2.shift) if bs2 else None, n(bs2.mean) if bs2 else None,
                      n(bs2.invstd) if bs2 else None, pws.data_ptr(), y1.M, y1.C, act, float(slope), float(y1.M),
                      bs1.bn.weight.data_ptr(), g1.data_ptr(), b1.data_ptr(), n(bs1.cA), n(bs1.cB), n(bs1.cC),
                      bs2.bn.weight.data_ptr() if bs2 else None, n(g2), n(b2), n(bs2.cA) if bs2 else None, n(bs2.cB) if bs2 else None,
                      n(bs2.cC) if bs2 else None)
        if not apply:
            assert y2 is None
            return None
        self.call(self.bwd, L.bn_act_bwd_apply, dt, dout.ptr, dout.ldc, y1.ptr, y1.ldc, n(bs1.scale), n(bs1.shift), n(bs1.cA),
                  n(bs1.cB), n(bs1.cC), dy1.ptr, dy1.ldc,
                  y2.ptr if y2 is not None else None, y2.ldc if y2 is not None else 0,
                  n(bs2.scale) if bs2 else None, n(bs2.shift) if bs2 else None, n(bs2.cA) if bs2 else None,
                  n(bs2.cB) if bs2 else None, n(bs2.cC) if bs2 else None,
                  dy2.ptr if dy2 is not None else None, dy2.ldc if dy2 is not None else 0, y1.M, y1.C, act, float(slope))
        return (dy1, dy2) if y2 is not None else dy1

    # ---- first layer (its input needs no gradient): dy = BatchNorm-backward(dz, y) has the weight gradient as its ONLY reader, so it is formed in
    # that kernel's operand load (mdcv_conv2d_wgrad_bnapply, conv_igemm.hip BNA) and the apply pass over the network's largest tensor never runs.
    # YOLOv3 416^2 batch 32: apply 193 us on the main queue + weight gradient 131 us alone behind it, at the exposed tail of the backward.
    wgrad_bnapply = True               # (tests / scripts/ab_step.py flip the class attribute; no environment knob)
    first_conv_2pass = True            # the first conv's forward as two streaming passes over its INPUT (csrc/first_conv.hip; yolo/models.py)

    def emit_first_conv_bwd(self, dout, y, bs, act, slope, cs, xnode):
        """Backward of conv -> BatchNorm -> activation for a layer whose input needs no gradient: statistics as usual, then ONE weight-gradient
        launch that reads (dz, y) itself.  Bit-identical to apply + weight gradient.  Returns False (nothing emitted) when the layer does not
        take this form."""
        L = self.L
        x = xnode.act
        if not self.wgrad_bnapply or xnode.needs_grad or self.dtype != BF16 or cs.bias is not None:
            return False
        geom = (x.B, x.H, x.W, cs.cin_pad, y.H, y.W, cs.cout_pad, cs.kh, cs.kw, cs.stride, cs.pad, cs.dil)
        if not int(L.conv2d_wgrad_bnapply_ok(self.wdt, *geom, dout.ldc, y.ldc, x.ldc)):
            return False
        self.emit_bn_act_bwd(dout, y, bs, act, slope, apply=False)
        gw = self.param_grad(cs.weight)
        # block target 768 = three 48 KiB (two-stage) blocks on every CU in ONE round: alone at 416^2 batch 32 (scripts/bna_ab.py) 191 us against
        # 262 us with the generic target of 512 and three stages, 391 us with four stages (96 KiB: one block per CU, two rounds)
        sdt = _lib.tuned(self.dtype, 20000 + 768) if self.tune_wgrad == 0 else self.wdt
        splits = int(L.conv2d_wgrad_splits_geom(sdt, *geom, dout.ldc, x.ldc))
        self.ws_floats = max(self.ws_floats, splits * cs.cout_pad * cs.kh * cs.kw * cs.cin_pad)
        plan = self
        n = lambda t: t.data_ptr()  # noqa: E731

        def wgrad(stream):
            return L.conv2d_wgrad_bnapply(plan.wdt, dout.ptr, dout.ldc, y.ptr, y.ldc, n(bs.scale), n(bs.shift), n(bs.cA), n(bs.cB), n(bs.cC), act,
                                          float(slope), x.ptr, x.ldc, plan.wgrad_ws(stream).data_ptr(), splits, gw.data_ptr(), 0, x.B, x.H, x.W,
                                          cs.cin_pad, cs.cin, y.H, y.W, cs.cout_pad, cs.cout, cs.kh, cs.kw, cs.stride, cs.pad, cs.dil, stream)
        wgrad.__name__ = "conv2d_wgrad"
        wgrad.info = (x.B, x.H, x.W, cs.cin_pad, y.H, y.W, cs.cout_pad, cs.kh, cs.stride, splits)
        self.bwd.append((wgrad, ()))
        self.first_conv_fused = True
        return True

    # On by default (Plan.fuse_bn = False restores the two-pass form).  YOLOv3 416^2 B=32: it removes 0.95 ms of stand-alone reduce kernels
    # per step and adds ~1.1 ms to the 66 data gradients' store loops (the y loads are HBM misses whose latency is exposed once per
    # 128-row group at the end of each tile) -- neutral while everything ran on one stream, +0.9 % (2039 -> 2058 img/s, same-box A/B)
    # now that the weight gradients fill the MFMA pipe from the side stream and the main stream is what bounds the step.
    fuse_bn = True                     # (tests flip the class attribute; no environment knob)

    fuse_skip = 141                    # bit mask of the geometry classes of _fuse_pays that keep the stand-alone reduce pass (0: fuse all);
                                       # 128 (round 3): the sparse stride-2 data gradients 13->26 / 26->52, whose fused store loops ran once per parity
                                       # class and cost 100 us against 20-28 us for the stand-alone pass (same-box A/B, scripts/ab_step.py "F13;F141": -0.07 ms per step)
    # Data gradients whose fused sums would take more than this many partial rows (RektNet's 80^2 x 256 tensors: 12 800; YOLOv3's 208^2 / 416^2
    # layers) keep the stand-alone reduce pass.  The finalize can take them since round 2 (rows beyond 4096 are folded in place first,
    # csrc/elementwise.hip), but the fused store loops still lose on these HBM-bound layers: RektNet 31.99k -> 31.34k img/s, YOLOv3 2136 -> 2118
    # with the limit lifted (same-box A/B).
    fuse_max_rows = 4096
    fuse_max_rows_s2 = 1 << 16          # the stride-2 shift form (see _fuse_bn_sums); rows beyond 4096 are folded in place by the finalize

    def _fuse_pays(self, geom):
        """Per-geometry choice between the fused sums and the stand-alone reduce pass (Plan.fuse_skip = bit mask of the classes
        that keep the stand-alone pass; 0: fuse every eligible data gradient).  The fused store loop runs at the END of every tile;
        when the launch is a single round of workgroups over a large dx tensor nothing hides it and the stand-alone pass (full HBM
        rate) wins.  Classes by pixels of dx (yolo_baseline 416^2 at batch 32 in brackets), chosen by same-box A/B of the whole
        step (scripts/ab_step.py "F<mask>;..."), not by the isolated per-launch times, which rank them differently."""
        B, dyH, dyW, cdy, xH, xW, cdx, kh, kw, stride, pad, dil = geom
        px = B * xH * xW
        k = self.fuse_skip
        if kh == 3 and stride == 1:
            if 50000 <= px < 100000:       # [52^2 3x3, 11 layers]
                return not (k & 1)
            if 20000 <= px < 50000:        # [26^2 3x3, 11 layers]
                return not (k & 2)
            if px >= 300000:               # [104^2 3x3, 2 layers]
                return not (k & 32)
            if px < 20000:                 # [13^2 3x3, 8 layers]
                return not (k & 256)
        if kh == 1:
            if px >= 300000:               # [104^2 1x1, 2 layers]
                return not (k & 4)
            if 50000 <= px < 100000:       # [52^2 1x1, 10 layers]
                return not (k & 16)
            if 20000 <= px < 50000:        # [26^2 1x1, 11 layers]
                return not (k & 64)
            if px < 20000:                 # [13^2 1x1, 8 layers]
                return not (k & 512)
        if stride == 2 and 300000 <= px < 1000000:   # [52^2 -> 104^2]
            return not (k & 8)
        if stride == 2 and px < 300000:              # [13^2 -> 26^2, 26^2 -> 52^2]
            return not (k & 128)
        return True

    def _fuse_bn_sums(self, dout, y, bs, act, slope, dgamma, dbeta):
        """Fold the BatchNorm-backward reduction over (dout, y) into the store loop of the data gradient that wrote `dout`.

        Legal when that launch is the LAST writer of the buffer (nothing between it and this point of the backward list mentions
        the pointer), it wrote exactly this tensor, and the library has a fused path for its geometry.  The earlier list entry is
        rewritten in place; what remains here is the column-owner finalize."""
        if not self.fuse_bn or self.dtype != BF16:       # production dtype only: the fp32 parity mode keeps the two-pass form (its fused
            # variants left partial rows unwritten at some sizes -- found with MDCV_POISON at batch 16..32)
            return False
        e = self.dgrad_entries.get(dout.ptr)
        if e is None or e["used"]:
            return False
        o = e["out"]
        if (o.M, o.C, o.ldc) != (dout.M, dout.C, dout.ldc) or (y.M, y.C) != (dout.M, dout.C):
            return False
        for fn, args in self.bwd[e["idx"] + 1:]:
            if dout.ptr in args:
                return False
        L, dt = self.L, self.dtype
        pwb = e.get("pwb")
        g = e["geom"]
        if pwb is not None:            # the one-launch 1x1 backward (_emit_pw_bwd1): y of the sums rides in its DMA ring, one partial row per slab
            rows = int(pwb[11])
            partial = self.f32(rows * 2 * y.C, zero=False)
            a2 = list(pwb)
            a2[12:20] = [y.ptr, y.ldc, bs.scale.data_ptr(), bs.shift.data_ptr(), bs.mean.data_ptr(), act, float(slope), partial.data_ptr()]
            assert self.bwd[e["idx"]][0] is L.pw_bwd
            self.bwd[e["idx"]] = (L.pw_bwd, tuple(a2))
        else:
            s2_shift = bool(L.conv2d_dgrad_s2_form_ok(self.cdt, *g, e["head"][1]))    # the shift kernel's stride-2 form (csrc/conv_shift.hip MODE 3): its
            # store loop writes whole output rows from LDS and takes the y loads of the sums in its stride -- 208 -> 416: stand-alone reduce
            # 180 us in the step against +70 us in the data gradient, at the HBM-bound tail of the backward; one row per 8 x 31 tile.
            # (It pays at every size: the pixel classes of _fuse_pays describe the per-class im2col launches.)
            if not s2_shift and not self._fuse_pays(g):
                return False
            rows = int(L.conv2d_dgrad_bnsums_rows(self.cdt, *g, e["head"][1]))
            if rows <= 0 or rows > (self.fuse_max_rows_s2 if s2_shift else self.fuse_max_rows):
                return False
            assert self.bwd[e["idx"]][0] is L.conv2d
            partial = self.f32(rows * 2 * y.C, zero=False)
            h = e["head"]
            self.bwd[e["idx"]] = (L.conv2d_dgrad_bnsums, (self.cdt, h[0], h[1], h[2], h[3], h[4], h[5], h[6], *e["geom"], y.ptr, y.ldc,
                                                          bs.scale.data_ptr(), bs.shift.data_ptr(), bs.mean.data_ptr(), act, float(slope),
                                                          partial.data_ptr()))
        e["used"] = True
        self.call(self.bwd, L.bn_bwd_finalize_rows, partial.data_ptr(), rows, y.C, float(y.M), bs.bn.weight.data_ptr(),
                  bs.mean.data_ptr(), bs.invstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), bs.cA.data_ptr(), bs.cB.data_ptr(),
                  bs.cC.data_ptr())
        self.fused_bn += 1
        return True

    def finish_pack(self, position=0):
        import struct
        rec = b"".join(struct.pack("<QQQiiiiiiiiQQ", cs.weight.data_ptr(), cs.wf.data_ptr(), cs.wd.data_ptr() if cs.wd is not None else 0,
                                   cs.cout, cs.cin, cs.kh * cs.kw, cs.cout_pad, cs.cin_pad, 0, 0, 0,
                                   cs.bias.data_ptr() if cs.bias is not None else 0,
                                   cs.bias_pad.data_ptr() if cs.bias is not None else 0) for cs in self.pack_list)
        table = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(self.device)
        self.keep.append(table)
        # LDS tile of the pack kernel: 16 x (min(64, Cin_pad) x taps + 1) floats; express it as "taps at 64 input channels"
        eq_taps = max(1, max((min(64, cs.cin_pad) * cs.kh * cs.kw + 63) // 64 for cs in self.pack_list))
        self.pack_table, self.pack_eq_taps = table, eq_taps
        plan = self
        args = (self.dtype, table.data_ptr(), len(self.pack_list), eq_taps)

        def pack_all(stream):
            owner = plan.owner
            if plan._packed_ahead:                       # the optimizer packed every layer behind its update (optim.py, pipeline=True);
                plan._packed_ahead = False               # the per-group waits further down this list order the forward behind it
                # two guards: the parameter epoch (load_weights / load_state_dict / re-flatten / another optimizer step; the raw kernels
                # bump no tensor version) AND the parameters' own version counters (user code editing parameters in place between step()
                # and this forward: nn.init.*, p.clamp_() under no_grad, EMA copy-back -- a Parameter whose .data is a view of the flat
                # buffer keeps its OWN counter, the flat buffer's does not move)
                if owner is None or (getattr(owner, "_param_epoch", 0) == plan._packed_version
                                     and owner._param_versions() == plan._packed_tversion):
                    return 0
            if owner is not None:
                owner._param_sync()
            return plan.L.pack_weights_batched(*args, stream)
        pack_all.__name__ = "mdcv_pack_weights_batched"
        self.fwd.insert(position, (pack_all, ()))
        self.layer_marks = [m + 1 if m >= position else m for m in self.layer_marks]

    def check_redzones(self):
        """MDCV_REDZONE mode: number of plan buffers whose guard bytes were overwritten (0 = every kernel stayed inside)."""
        bad = 0
        for raw, n in self.redzones:
            if int((raw[n:] != _RZ_BYTE).sum()):
                bad += 1
        return bad

    def launch_param_group(self, k, gated):
        """Enqueue a deferred group update (closure left by the optimizer).  gated: the parameter stream first waits for the
        point the current stream has reached."""
        if 0 <= k < len(self._pending_updates) and self._pending_updates[k] is not None:
            fn, self._pending_updates[k] = self._pending_updates[k], None
            fn(torch.cuda.current_stream() if gated else None)

    def ensure_param_groups(self, pflat):
        """Cuts the flat parameter buffer into a few forward-ordered groups of whole layers (small first, so the next forward can
        start at once) and plants a `wait_params(k)` in the forward list in front of each group's first layer.  Returns
        [(lo, hi, first_layer, nlayers)] or [] when the layout does not allow it."""
        if self.param_groups is not None:
            return self.param_groups
        self.param_groups = []
        base, n = pflat.data_ptr(), pflat.numel()
        offs = [(cs.weight.data_ptr() - base) // 4 for cs in self.pack_list]
        if not offs or any(o < 0 or o >= n for o in offs) or any(b <= a for a, b in zip(offs, offs[1:])):
            return self.param_groups
        firsts, target, lo = [0], 128 * 1024, 0           # 0.5 MB, then x4 per group up to 48 MB
        for li in range(1, len(offs)):
            if offs[li] - lo >= target:
                firsts.append(li)
                lo = offs[li]
                target = min(target * 4, 12 * 1024 * 1024)
        groups = []
        for gi, f in enumerate(firsts):
            last = firsts[gi + 1] if gi + 1 < len(firsts) else len(offs)
            groups.append((0 if gi == 0 else offs[f], offs[last] if last < len(offs) else n, f, last - f))
        self._group_events = [None] * len(groups)
        self._pending_updates = [None] * len(groups)
        plan = self
        for k in range(len(groups) - 1, -1, -1):            # descending positions: earlier insert points stay valid
            def wait_params(stream, k=k):
                plan.launch_param_group(k, gated=False)      # (normally launched two groups earlier, below)
                ev = plan._group_events[k]
                if ev is not None:
                    torch.cuda.current_stream().wait_event(ev)
                    plan._group_events[k] = None
                # start the update of group k+2 once the forward is HERE on the GPU: the early layers are HBM-bound like the
                # optimizer, the 52x52 .. 13x13 layers whose parameters make up the bulk are MFMA-bound -- the update of a late
                # group should run under those, not under the first layers
                plan.launch_param_group(k + 2, gated=True)
                return 0
            wait_params.__name__ = "wait_params"
            self.fwd.insert(self.layer_marks[groups[k][2]], (wait_params, ()))
        self.param_groups = groups
        return groups

    # ------------------------------------------------------------------ hipGraph capture of the launch lists
    def capture(self, which, stream=None):
        """Capture a launch list into a hipGraph (all pointers are static by construction)."""
        L = self.L
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        lst = self.fwd if which == "fwd" else self.bwd
        import ctypes
        L.check(L.graph_begin(stream), "graph_begin")
        try:
            self.run(lst, stream)
        finally:
            ge = ctypes.c_void_p()
            rc = L.graph_end(stream, ctypes.byref(ge))
        L.check(rc, "graph_end")
        return ge


def run_timed(plan, lst, stream=None, kernels=False):
    """Run a launch list with a HIP event after every op (on the launch stream).  Returns [(name, ms, args)]; with
    kernels=True each entry gets a fourth element, [(kernel symbol as rocprofv3 prints it, ms)] for every kernel the library
    launched inside that op, each bracketed by its own pair of HIP events on the launch stream (mdcv_profile_*, csrc/runtime.hip)."""
    import ctypes
    L = plan.L
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    evs = []
    for _ in range(len(lst) + 1):
        e = ctypes.c_void_p()
        L.check(L.event_create(ctypes.byref(e)), "event_create")
        evs.append(e)
    spans = []
    if kernels:
        L.check(L.profile_begin(), "profile_begin")
    L.check(L.event_record(evs[0], stream))
    try:
        for i, (fn, args) in enumerate(lst):
            n0 = L.profile_count() if kernels else 0
            rc = fn(*args, stream)
            if rc:
                raise _lib.MdcvError(f"{getattr(fn, '__name__', fn)} returned {rc}")
            spans.append((n0, L.profile_count() if kernels else 0))
            L.check(L.event_record(evs[i + 1], stream))
    finally:
        nrec = L.profile_stop() if kernels else 0
    L.check(L.event_sync(evs[-1]))
    krec = []
    if kernels:
        if nrec < 0:
            raise _lib.MdcvError(f"mdcv_profile_stop failed ({nrec})")
        buf = ctypes.create_string_buffer(1024)
        for i in range(nrec):
            ms = ctypes.c_float()
            L.check(L.profile_read(i, ctypes.byref(ms), buf, 1024), "profile_read")
            krec.append((buf.value.decode(), ms.value))
    out = []
    for i, (fn, args) in enumerate(lst):
        ms = ctypes.c_float()
        L.check(L.event_elapsed_ms(evs[i], evs[i + 1], ctypes.byref(ms)))
        ent = (getattr(fn, "__name__", str(fn)), ms.value, args if args else getattr(fn, "info", ()))
        out.append(ent + (krec[spans[i][0]:spans[i][1]],) if kernels else ent)
    for e in evs:
        L.event_destroy(e)
    return out
