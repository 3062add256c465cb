"""A/B timing of conv variants on the dominant layer shapes, interleaved in one process (medians of several rounds).
usage: conv_ab.py "v1,v2,..." [iters] [rounds]     variants: per-call codes of csrc/tune.h (-3/-4 shift kernel ring depth, 12/10/... im2col tiles, 0 defaults)"""
import ctypes, os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib()
DT = [1]                    # the dtype argument of the calls below: bf16 | per-call variant code (csrc/tune.h; 0 = defaults)
st = torch.cuda.current_stream().cuda_stream
variants = [int(v) for v in sys.argv[1].split(",")]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
SHAPES = [(32, 52, 128, 256, 3, 1, 0), (32, 52, 128, 256, 3, 1, 1), (32, 26, 256, 512, 3, 1, 0), (32, 26, 256, 512, 3, 1, 1),
          (32, 13, 512, 1024, 3, 1, 0), (32, 13, 512, 1024, 3, 1, 1), (32, 104, 64, 128, 3, 1, 0), (32, 104, 64, 128, 3, 1, 1)]
if len(sys.argv) > 4:
    SHAPES = [tuple(int(x) for x in s.split()) for s in sys.argv[4].split(";")]
e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p(); L.event_create(ctypes.byref(e0)); L.event_create(ctypes.byref(e1))
for (B, H, Ci, Co, k, s, mode) in SHAPES:
    pad = (k - 1) // 2
    Ho = (H + 2 * pad - k) // s + 1
    nsets = int(os.environ.get("NSETS", "6"))
    xs = [torch.randn(B * H * H * Ci, device="cuda").to(torch.bfloat16) for _ in range(nsets)]
    ys = [torch.randn(B * Ho * Ho * Co, device="cuda").to(torch.bfloat16) for _ in range(nsets)]
    wfs = [(torch.randn(Co * k * k * Ci, device="cuda") * 0.05).to(torch.bfloat16) for _ in range(nsets if os.environ.get("COLDW") else 1)]
    stt = torch.zeros(L.conv2d_stats_rows_geom(DT[0], B, Ho, Ho, Ci, Co, k, k, s, pad, 1, Ci) * 2 * Co + 4096, device="cuda")
    coef = [torch.rand(Ci, device="cuda") + 0.5 for _ in range(3)]
    prow = L.conv2d_dgrad_bnsums_rows(DT[0], B, Ho, Ho, Co, H, H, Ci, k, k, s, pad, 1, Co) if mode >= 2 else 0
    part = torch.zeros(max(1, prow) * 2 * Ci + 4096, device="cuda")
    def call(i):
        x, y, wf = xs[i % nsets], ys[i % nsets], wfs[i % len(wfs)]
        if mode == 0:
            return L.conv2d(DT[0], 0, x.data_ptr(), Ci, wf.data_ptr(), y.data_ptr(), Co, None, None, 0, stt.data_ptr(), B, H, H, Ci, Ho, Ho, Co, k, k, s, pad, 1, st)
        if mode == 1:
            return L.conv2d(DT[0], 1, y.data_ptr(), Co, wf.data_ptr(), x.data_ptr(), Ci, None, None, 0, None, B, Ho, Ho, Co, H, H, Ci, k, k, s, pad, 1, st)
        # mode 2: data gradient with the fused BatchNorm-backward sums (+ addsrc, as in a residual block); mode 3: the same without addsrc
        yy = xs[(i + 1) % nsets]
        return L.conv2d_dgrad_bnsums(DT[0], y.data_ptr(), Co, wf.data_ptr(), x.data_ptr(), Ci, xs[(i + 2) % nsets].data_ptr() if mode == 2 else None, Ci,
                                     B, Ho, Ho, Co, H, H, Ci, k, k, s, pad, 1, yy.data_ptr(), Ci, coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(),
                                     1, 0.1, part.data_ptr(), st)
    res = {v: [] for v in variants}
    for v in variants:
        DT[0] = _lib.tuned(1, v)
        for i in range(50): assert call(i) == 0
    torch.cuda.synchronize()
    for r in range(rounds):
        for v in variants:
            DT[0] = _lib.tuned(1, v)
            L.event_record(e0, st)
            for i in range(iters): call(i)
            L.event_record(e1, st); L.event_sync(e1)
            ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms))
            res[v].append(ms.value / iters)
    fl = 2.0 * B * Ho * Ho * Co * k * k * Ci
    print((B, H, Ci, Co, k, s, mode), " | ".join("v%d: %.1f us %4.0f TF (best %4.0f)" % (v, 1e3 * statistics.median(t), fl / statistics.median(t) / 1e9, fl / min(t) / 1e9) for v, t in res.items()), flush=True)
