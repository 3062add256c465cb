"""Same-box A/B of the whole YOLOv3 training step (BASELINE config 3) under tuning hooks, interleaved in one process, one model per setting
(plan-build-time switches need their own plan).
usage: ab_step.py "c-30;c-31;c-31,w30005;P0" [rounds] [steps] [yolo|rektnet]
  c<n> / w<n> = engine.Plan.tune_conv / tune_wgrad: ONE variant code per family (csrc/tune.h), carried by every conv / weight-gradient call of that model's plans
  P0 / P1 = engine.Plan.pw_fuse off / on, F<mask> = engine.Plan.fuse_skip, R<n> = engine.Plan.fuse_max_rows, Y0 / Y1 = engine.Plan.stats_xacc, B0 / B1 = engine.Plan.pw_bwd1, X<substr> = what-if timing without the launches whose name contains it, S0 = model.strict_targets off                                      (applied when the model is built)
Every setting is applied on top of the first one (the baseline), which is re-applied in front of each."""
import os, sys, tempfile, time, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdcv import _lib
from mdcv.yolo.models import Darknet
from mdcv.optim import FusedAdam

L = _lib.lib()
settings = [s.split(",") for s in (sys.argv[1] if len(sys.argv) > 1 else "c-30;c-31").split(";")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
workload = sys.argv[4] if len(sys.argv) > 4 else "yolo"


from mdcv import engine
BUILD_DEFAULTS = dict(pw_fuse=engine.Plan.pw_fuse, fuse_skip=engine.Plan.fuse_skip, fuse_max_rows=engine.Plan.fuse_max_rows, sx=engine.Plan.stats_xacc, pb=engine.Plan.pw_bwd1)


def apply(codes):
    """(run-time hooks are gone with the library's tuning state: c<code> / w<code> are plan-build-time switches now, see apply_build)"""


def apply_build(codes):
    engine.Plan.pw_fuse, engine.Plan.fuse_skip, engine.Plan.fuse_max_rows = BUILD_DEFAULTS["pw_fuse"], BUILD_DEFAULTS["fuse_skip"], BUILD_DEFAULTS["fuse_max_rows"]
    engine.Plan.stats_xacc = BUILD_DEFAULTS["sx"]
    engine.Plan.stats_xacc_pw = True
    engine.Plan.pw_bwd1 = BUILD_DEFAULTS["pb"]
    engine.Plan.tune_conv = engine.Plan.tune_wgrad = 0
    engine.Plan.wgrad_after_dgrad = False
    engine.Plan.wgrad_bnapply = True
    engine.Plan.first_conv_2pass = True
    engine.Plan.pw_fwd_px = (50000, 1 << 30)
    from mdcv.yolo import models as _ym0
    _ym0._NetPlan.fork_on_dispatch = True
    _ym0._NetPlan.defer_slab_reduce = True
    for c in codes:
        if c and c[0] == "K":          # K0 / K1: side-stream forks as event records on the main queue / carried by the producing kernel's dispatch packet
            from mdcv.yolo import models as _ym
            _ym._NetPlan.fork_on_dispatch = bool(int(c[1:]) & 1)
            _ym._NetPlan.defer_slab_reduce = not bool(int(c[1:]) & 2)               # K3: forks on dispatch, slab reduces NOT deferred
        if c and c[0] == "c":
            engine.Plan.tune_conv = int(c[1:])
        if c and c[0] == "w":
            engine.Plan.tune_wgrad = int(c[1:])
        if c and c[0] == "I":          # I0 / I1: the first conv's forward as conv + apply pass / as two streaming passes over its input (csrc/first_conv.hip)
            engine.Plan.first_conv_2pass = bool(int(c[1:]))
        if c and c[0] == "M":          # M<n>: the fused 1x1 forward block (BatchNorm apply in the operand load) from n pixels per layer (default 50000: 52^2 and up)
            engine.Plan.pw_fwd_px = (int(c[1:]), 1 << 30)
        if c and c[0] == "G":          # G0 / G1: the first layer's BatchNorm-apply pass as a launch / inside its weight gradient's operand load
            engine.Plan.wgrad_bnapply = bool(int(c[1:]))
        if c and c[0] == "W":          # W0 / W1: a 3x3 layer's weight gradient forked in front of / behind its data gradient
            engine.Plan.wgrad_after_dgrad = bool(int(c[1:]))
        if c and c[0] == "B":          # B0 / B1: 1x1 layers' backward as data gradient + weight gradient + reduce / in one launch (csrc/pw_bwd.hip)
            engine.Plan.pw_bwd1 = bool(int(c[1:]))
        if c and c[0] == "Q":          # Q0 / Q1: layers in front of a fused 1x1 forward block keep their finalize launch / take the publisher-only accumulator form
            engine.Plan.stats_xacc_pw = bool(int(c[1:]))
        if c and c[0] == "Y":          # Y0 / Y1: forward statistics as partial rows + finalize launch / through exact accumulators (csrc/exact_acc.h)
            engine.Plan.stats_xacc = bool(int(c[1:]))
        if c and c[0] == "P":
            engine.Plan.pw_fuse = bool(int(c[1:]))
        if c and c[0] == "F":
            engine.Plan.fuse_skip = int(c[1:])
        if c and c[0] == "R":
            engine.Plan.fuse_max_rows = int(c[1:])


dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
g = torch.Generator().manual_seed(1000)
def make(i):
    apply_build(settings[0]); apply_build(settings[i])
    if workload == "yolo":
        cfg = bench.write_yolo_cfg(tmp)
        os.chdir(tmp)
        torch.manual_seed(0)
        net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True).to(dev).train()
        net.strict_targets = "S0" not in settings[i]          # S0: the bad-label check at the start of backward does not wait for the forward
        opt = FusedAdam(net, lr=1e-3)

        def step():
            opt.zero_grad()
            loss = net(x, tg)[0].sum()
            loss.backward()
            opt.step()
            return loss
    else:
        from mdcv.rektnet.keypoint_net import KeypointNet
        from mdcv.rektnet.cross_ratio_loss import CrossRatioLoss
        torch.manual_seed(0)
        net = KeypointNet(7, (80, 80)).to(dev).train()
        crit = CrossRatioLoss("l1_softargmax", True, 0.05, 0.05)
        opt = FusedAdam(net, lr=1e-3)

        def step():
            opt.zero_grad()
            out = net(x)
            loss = crit(out[0], out[1], None, pts)[2]
            loss.backward()
            opt.step()
            return loss
    apply(settings[0]); apply(settings[i])
    for _ in range(6): l = step()          # builds the plan under this setting's switches
    drop = [c[1:] for c in settings[i] if c and c[0] == "X"]
    if drop:                               # X<substring>: WHAT-IF timing -- launches whose name contains it are removed from the plan's lists (results are garbage)
        for plan in net._plans.values():
            for lst_name in ("fwd", "bwd"):
                lst = getattr(plan, lst_name)
                keep = [(f, a) for f, a in lst if not any(d in getattr(f, "__name__", "") for d in drop)]
                print("what-if", settings[i], lst_name, len(lst), "->", len(keep), flush=True)
                lst[:] = keep
    return step, float(l.detach())


if workload == "yolo":
    x, tg = torch.rand(32, 3, 416, 416, generator=g).to(dev), bench.synth_targets(32, 16, g).to(dev)
    nimg = 32
else:
    x = torch.rand(256, 3, 80, 80, generator=g).to(dev)
    pts = torch.rand(256, 7, 2, generator=g).to(dev) * 0.9
    nimg = 256
steps_fn, last = {}, {}
for i in range(len(settings)):
    steps_fn[i], last[i] = make(i)
torch.cuda.synchronize()
res = {i: [] for i in range(len(settings))}
for rnd in range(rounds):
    for i, s in enumerate(settings):
        apply(settings[0]); apply(s)
        step = steps_fn[i]
        for _ in range(3): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps): l = step()
        torch.cuda.synchronize()
        res[i].append((time.perf_counter() - t0) / steps * 1e3)
        last[i] = float(l.detach())
for i, s in enumerate(settings):
    print("%-28s median %.3f ms  min %.3f  (%s)  %.0f img/s  last loss %.4f" % (",".join(s), statistics.median(res[i]), min(res[i]),
          " ".join("%.3f" % t for t in res[i]), nimg / statistics.median(res[i]) * 1e3, last[i]), flush=True)
apply(settings[0])
