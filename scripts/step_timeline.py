"""One steady-state step of a rocprofv3 --kernel-trace CSV as a two-queue timeline: when each queue is busy, when both are, when
neither is, and which kernels run in the stretch where only ONE queue has work (the critical path's exposed parts).
usage: step_timeline.py <kernel_trace.csv> [step_index_from_end=2]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))


def nm(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    return k.split("(")[0][:44]


back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows)
adam = [i for i, e in enumerate(ev) if e[2].startswith("adam_kernel")]
lo, hi = adam[-back - 1], adam[-back]
step = ev[lo + 1:hi + 1]
t0 = ev[lo][1]
t1 = step[-1][1]
print("step: %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(step)))
qs = collections.Counter(e[3] for e in step)
main = qs.most_common(1)[0][0]
print("queues:", dict(qs), "main =", main)


def short(n):
    return n


# sweep line over [t0, t1]: state = (main busy, side busy)
pts = []
for s, e, k, q in step:
    pts.append((s, 1, q == main, k)); pts.append((e, -1, q == main, k))
pts.sort()
nm = ns = 0
last = t0
tot = collections.Counter()
only = {"main": collections.Counter(), "side": collections.Counter()}
active = {True: collections.Counter(), False: collections.Counter()}
phase_edges = []
for t, d, is_main, k in pts:
    dt = t - last
    if dt > 0:
        st = ("M" if nm else "-") + ("S" if ns else "-")
        tot[st] += dt
        if nm and not ns:
            for kk in active[True]:
                if active[True][kk] > 0: only["main"][kk] += dt
        if ns and not nm:
            for kk in active[False]:
                if active[False][kk] > 0: only["side"][kk] += dt
    last = t
    if is_main: nm += d
    else: ns += d
    active[is_main][k] += d
for st in ("MS", "M-", "-S", "--"):
    print("  %s  %.3f ms" % (st, tot[st] / 1e6))
# forward/backward boundary: first side-queue kernel of the step
side = [e for e in step if e[3] != main]
if side:
    print("first side kernel at +%.3f ms, last side kernel ends +%.3f ms; last main kernel before adam ends +%.3f ms" %
          ((side[0][0] - t0) / 1e6, (max(e[1] for e in side) - t0) / 1e6, (max(e[1] for e in step[:-1] if e[3] == main) - t0) / 1e6))
for w in ("main", "side"):
    print("only-%s time by kernel:" % w)
    for k, v in only[w].most_common(12): print("    %-42s %.3f ms" % (k, v / 1e6))
# idle gaps on main during forward (before first side kernel)
mainev = [e for e in step if e[3] == main]
fwd_end = side[0][0] if side else t1
gaps = sum(max(0, b[0] - a[1]) for a, b in zip(mainev, mainev[1:]) if b[0] <= fwd_end)
nf = sum(1 for e in mainev if e[0] <= fwd_end)
print("forward: %d launches, %.3f ms of gaps between them (%.2f us each)" % (nf, gaps / 1e6, gaps / 1e3 / max(1, nf)))
gb = sum(max(0, b[0] - a[1]) for a, b in zip(mainev, mainev[1:]) if b[0] > fwd_end)
nb = len(mainev) - nf
print("backward main: %d launches, %.3f ms of gaps (%.2f us each)" % (nb, gb / 1e6, gb / 1e3 / max(1, nb)))
sg = sum(max(0, b[0] - a[1]) for a, b in zip(side, side[1:]))
print("side: %d launches, %.3f ms of gaps (%.2f us each)" % (len(side), sg / 1e6, sg / 1e3 / max(1, len(side))))
# the large gaps of the main queue in the backward: what runs before and after each
big = collections.Counter(); bigt = collections.Counter()
for a, b in zip(mainev, mainev[1:]):
    g = b[0] - a[1]
    if b[0] > fwd_end and g > 3000:
        big[(a[2][:28], b[2][:28])] += 1; bigt[(a[2][:28], b[2][:28])] += g
print("main-queue gaps > 3 us in the backward, by (kernel before, kernel after):")
for k, v in bigt.most_common(14): print("    %-30s -> %-30s n=%3d  %.1f us each" % (k[0], k[1], big[k], v / 1e3 / big[k]))
big = collections.Counter(); bigt = collections.Counter()
for a, b in zip(side, side[1:]):
    g = b[0] - a[1]
    if g > 3000:
        big[(a[2][:28], b[2][:28])] += 1; bigt[(a[2][:28], b[2][:28])] += g
print("side-queue gaps > 3 us:")
for k, v in bigt.most_common(10): print("    %-30s -> %-30s n=%3d  %.1f us each" % (k[0], k[1], big[k], v / 1e3 / big[k]))
