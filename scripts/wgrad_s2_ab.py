"""3x3 / stride-2 weight gradients of yolo_baseline (batch 32: the five down-sampling layers) alone, rotating operand sets: us per call (kernel + slab
reduce) and TFLOP/s.   usage: wgrad_s2_ab.py [iters] [codes]   codes: per-call variant codes of the weight-gradient family (csrc/tune.h):
0 = defaults (parity-plane LDS ring, csrc/wgrad_stream_s2.hip, where its block fits beside a main-queue workgroup), 34060 = generic kernel,
34228 = the ring kernel up to 128 KiB of LDS per block."""
import ctypes, os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
codes = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "34060,0,34228").split(",")]
e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p(); L.event_create(ctypes.byref(e0)); L.event_create(ctypes.byref(e1))
SH = [(13, 512, 1024), (26, 256, 512), (52, 128, 256), (104, 64, 128), (208, 32, 64)]       # Hout, Cin, Cout
NSET = 4
for (Ho, Ci, Co) in SH:
    B, H = 32, 2 * Ho
    M = B * Ho * Ho
    g = torch.Generator(device="cuda").manual_seed(1)
    dys = [torch.randn(M, Co, device="cuda", generator=g).to(torch.bfloat16) for _ in range(NSET)]
    xs = [torch.randn(B * H * H, Ci, device="cuda", generator=g).to(torch.bfloat16) for _ in range(NSET)]
    dw = torch.zeros(Co, Ci, 3, 3, device="cuda")
    out, res = [], []
    for code in codes:
        dt = _lib.tuned(1, code)
        splits = L.conv2d_wgrad_splits_geom(dt, B, H, H, Ci, Ho, Ho, Co, 3, 3, 2, 1, 1, Co, Ci)
        ws = torch.empty(splits * Co * 9 * Ci, device="cuda")

        def call(i):
            rc = L.conv2d_wgrad(dt, dys[i % NSET].data_ptr(), Co, xs[i % NSET].data_ptr(), Ci, ws.data_ptr(), splits, dw.data_ptr(), 0, B, H, H, Ci, Ci,
                                Ho, Ho, Co, Co, 3, 3, 2, 1, 1, st)
            assert rc == 0, rc
        for i in range(5): call(i)
        torch.cuda.synchronize()
        ts = []
        for r in range(3):
            L.event_record(e0, st)
            for i in range(iters): call(i)
            L.event_record(e1, st); L.event_sync(e1)
            ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms))
            ts.append(ms.value / iters * 1e3)
        call(0); torch.cuda.synchronize(); res.append(dw.clone())
        t = statistics.median(ts)
        out.append("code %6d: splits %3d  %6.1f us  %5.0f TF" % (code, splits, t, 2.0 * M * Co * 9 * Ci / t / 1e6))
    err = max(float((r - res[0]).abs().max() / res[0].abs().max()) for r in res)
    print("Hout=%3d %4d->%4d | " % (Ho, Ci, Co) + " | ".join(out) + " | max rel diff %.1e" % err, flush=True)
