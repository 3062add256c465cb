"""Debug: find the first forward op after which instance 2k+1 and 2k+2 (same process) differ."""
import os, sys, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdcv.yolo.models import Darknet
tmp = tempfile.mkdtemp(); cfg = bench.write_yolo_cfg(tmp)
B = int(os.environ.get("DBG_B", "32"))
def build():
    cwd = os.getcwd(); os.chdir(tmp)
    torch.manual_seed(0)
    net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision="bf16")
    os.chdir(cwd)
    net = net.cuda().train()
    g = torch.Generator().manual_seed(1)
    x = torch.rand(B, 3, 416, 416, generator=g).cuda(); tg = bench.synth_targets(B, 16, g).cuda()
    net(x, tg); torch.cuda.synchronize()          # builds the plan (and runs it once)
    plan = [p for p in net._plans.values()][0]
    return net, plan, x, tg
def sums(plan):
    return [float(b.float().double().sum()) for b in plan.keep if torch.is_tensor(b) and b.dtype in (torch.float32, torch.bfloat16)]
n1, p1, x1, t1 = build(); n2, p2, x2, t2 = build()
st = torch.cuda.current_stream().cuda_stream
names = [getattr(f, "__name__", str(f)) for f, _ in p1.fwd]
print("ops", len(p1.fwd))
# re-run both forwards op by op, compare all buffers every 8 ops, then refine
def rerun(plan, x, tg, upto):
    for b_ in plan.keep:
        if torch.is_tensor(b_): b_.zero_()
    plan.in_holder["src"] = x
    plan.targets.copy_(tg.reshape(plan.targets.shape))
    plan.run(plan.pre, st)
    plan.run(plan.fwd[:upto], st)
    torch.cuda.synchronize()
    return sums(plan)
lo, hi = 0, len(p1.fwd)
a, b = rerun(p1, x1, t1, hi), rerun(p2, x2, t2, hi)
print("full differ:", sum(1 for u, v in zip(a, b) if u != v and not (u != u and v != v)))
while hi - lo > 1:
    mid = (lo + hi) // 2
    a, b = rerun(p1, x1, t1, mid), rerun(p2, x2, t2, mid)
    d = sum(1 for u, v in zip(a, b) if u != v and not (u != u and v != v))
    print("upto", mid, names[mid - 1], "differ", d)
    if d: hi = mid
    else: lo = mid
print("first differing op index", hi - 1, names[hi - 1], p1.fwd[hi - 1][1][:40] if p1.fwd[hi - 1][1] else "")

# ---- inspect op `hi-1`: compare its operands between the two instances
def operands(plan, idx):
    fn, args = plan.fwd[idx]
    ptrs = {}
    for b in plan.keep:
        if torch.is_tensor(b): ptrs[b.data_ptr()] = b
        elif hasattr(b, "wf"):
            ptrs[b.wf.data_ptr()] = b.wf
            if b.wd is not None: ptrs[b.wd.data_ptr()] = b.wd
            if getattr(b, "bias_pad", None) is not None: ptrs[b.bias_pad.data_ptr()] = b.bias_pad
    return [(i, ptrs[a]) for i, a in enumerate(args) if isinstance(a, int) and a in ptrs]
i = hi - 1
rerun(p1, x1, t1, i + 1); rerun(p2, x2, t2, i + 1)
for (ia, ta), (ib, tb) in zip(operands(p1, i), operands(p2, i)):
    d = (ta.float() - tb.float()).abs()
    print("arg", ia, tuple(ta.shape), ta.dtype, "equal", torch.equal(ta, tb), "ndiff", int((d > 0).sum()), "maxdiff", float(d.max()), "ptr%4096", ta.data_ptr() % 4096, tb.data_ptr() % 4096, hex(ta.data_ptr()), hex(tb.data_ptr()))
    if ia == 5 and not torch.equal(ta, tb):
        idx = (d > 0).nonzero().flatten()
        Co = 128; Ho = 104
        print("   first diffs (img,y,x,c):", [(j // Co // (Ho * Ho), (j // Co) % (Ho * Ho) // Ho, (j // Co) % Ho, j % Co) for j in idx[:6].tolist()], " last:", [(j // Co // (Ho * Ho), (j // Co) % (Ho * Ho) // Ho, (j // Co) % Ho, j % Co) for j in idx[-3:].tolist()])
