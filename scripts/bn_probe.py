"""BatchNorm-backward passes alone, per layer shape of yolo_baseline at batch 32: stand-alone reduce (+ finalize) and apply, in TB/s of their
algorithmic traffic (reduce: dz + y ; apply: dz + y + dy).  usage: bn_probe.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for (M, C) in [(5537792, 32), (1384448, 64), (1384448, 32), (346112, 128), (346112, 64), (86528, 256), (86528, 128), (21632, 512), (5408, 1024)]:
    n = M * C
    NB = 4
    dz = [torch.randn(n, device="cuda").to(torch.bfloat16) for _ in range(NB)]
    y = [torch.randn(n, device="cuda").to(torch.bfloat16) for _ in range(NB)]
    dy = [torch.empty(n, device="cuda", dtype=torch.bfloat16) for _ in range(NB)]
    f = lambda: torch.rand(C, device="cuda") + 0.5
    sc, sh, mean, istd, gamma = f(), f(), f(), f(), f()
    dg, db, cA, cB, cC = f(), f(), f(), f(), f()
    ws = torch.empty(int(L.bn_act_bwd_reduce_ws_floats(1, M, C, 2)), device="cuda")
    i = [0]

    def red():
        i[0] += 1; k = i[0] % NB
        rc = L.bn_act_bwd_reduce_finalize(1, dz[k].data_ptr(), C, y[k].data_ptr(), C, sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), istd.data_ptr(),
                                          None, 0, None, None, None, None, ws.data_ptr(), M, C, 1, 0.1, float(M), gamma.data_ptr(), dg.data_ptr(),
                                          db.data_ptr(), cA.data_ptr(), cB.data_ptr(), cC.data_ptr(), None, None, None, None, None, None, st)
        assert rc == 0

    def app():
        i[0] += 1; k = i[0] % NB
        rc = L.bn_act_bwd_apply(1, dz[k].data_ptr(), C, y[k].data_ptr(), C, sc.data_ptr(), sh.data_ptr(), cA.data_ptr(), cB.data_ptr(), cC.data_ptr(),
                                dy[k].data_ptr(), C, None, 0, None, None, None, None, None, None, 0, M, C, 1, 0.1, st)
        assert rc == 0
    b = n * 2
    tr, ta = timeit(red), timeit(app)
    print("M=%8d C=%5d (%6.1f MB): reduce+finalize %6.1f us %.2f TB/s | apply %6.1f us %.2f TB/s" % (M, C, b / 1e6, tr * 1e6, 2 * b / tr / 1e12, ta * 1e6, 3 * b / ta / 1e12), flush=True)
