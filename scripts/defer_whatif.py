"""WHAT-IF timing: the weight gradients of the LATE layers (first in the backward list) launched on the side stream beside the NEXT forward
instead of beside the backward.  Results are garbage (the forward overwrites their operands); the schedule is real."""
import os, sys, tempfile, time, statistics, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdcv import _lib
from mdcv.yolo.models import Darknet
from mdcv.optim import FusedAdam
L = _lib.lib()
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
g = torch.Generator().manual_seed(1000)
x, tg = torch.rand(32, 3, 416, 416, generator=g).to(dev), bench.synth_targets(32, 16, g).to(dev)
fracs = [float(f) for f in (sys.argv[1] if len(sys.argv) > 1 else "0,0.3,0.5,0.7").split(",")]
start_at = int(sys.argv[2]) if len(sys.argv) > 2 else 0        # forward-list index in front of which the deferred launches are queued

def make(frac):
    cfg = bench.write_yolo_cfg(tmp)
    os.chdir(tmp)
    torch.manual_seed(0)
    net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True).to(dev).train()
    opt = FusedAdam(net, lr=1e-3)
    def step():
        opt.zero_grad()
        loss = net(x, tg)[0].sum()
        loss.backward()
        opt.step()
        return loss
    for _ in range(4): step()
    if frac > 0:
        for plan in net._plans.values():
            roles = plan._classify_bwd()
            widx = [i for i, r in enumerate(roles) if r >= 2]
            take = set(widx[:int(len(widx) * frac)])
            deferred = [plan.bwd[i] for i in sorted(take)]
            plan.bwd[:] = [e for i, e in enumerate(plan.bwd) if i not in take]
            plan.__dict__.pop("_bwd_roles", None)
            print("frac", frac, "deferred", len(deferred), "of", len(widx), "bwd entries left", len(plan.bwd), flush=True)
            orig_run = plan.run
            fwd_list = plan.fwd
            def run(lst, stream=None, plan=plan, deferred=deferred, orig_run=orig_run, fwd_list=fwd_list):
                if lst is fwd_list:
                    cur = torch.cuda.current_stream(); side = plan.side()
                    st, ss = cur.cuda_stream, side.cuda_stream
                    head, tail = lst[:start_at], lst[start_at:]
                    if head: orig_run(head, stream)
                    L.check(L.stream_fork(st, ss, 1), "fork")
                    for fn, args in deferred:
                        rc = fn(*args, ss)
                        assert rc == 0
                    orig_run(tail, stream)
                    L.check(L.stream_fork(ss, st, 1), "join")
                    return
                return orig_run(lst, stream)
            plan.run = run
            # plan.run in __dict__ makes run_bwd_list take the serial path: keep the overlap
            cls = type(plan)
            orig_bwd = cls.run_bwd_list
            def run_bwd_list(plan=plan, run=run):
                del plan.__dict__["run"]
                try:
                    orig_bwd(plan)
                finally:
                    plan.__dict__["run"] = run
            plan.run_bwd_list = run_bwd_list
    for _ in range(4): step()
    return step

steps = {f: make(f) for f in fracs}
torch.cuda.synchronize()
res = {f: [] for f in fracs}
for rnd in range(3):
    for f in fracs:
        s = steps[f]
        for _ in range(3): s()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30): s()
        torch.cuda.synchronize()
        res[f].append((time.perf_counter() - t0) / 30 * 1e3)
for f in fracs:
    print("defer %.2f: median %.3f ms (%s)" % (f, statistics.median(res[f]), " ".join("%.3f" % t for t in res[f])), flush=True)
