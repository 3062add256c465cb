"""Isolated forward + data-gradient time of every conv layer of yolo_baseline @416, batch 32 (bf16): where do the mdcv_conv2d
milliseconds go?  usage: conv_layers.py"""
import ctypes, os, sys, tempfile, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdcv import _lib
from mdcv.yolo.utils.parse_config import parse_model_config
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
tmp = tempfile.mkdtemp()
defs = parse_model_config(bench.write_yolo_cfg(tmp))
hyper = defs.pop(0)
B, S = 32, 416
shapes, outs = [], []
C, H = 3, S
for i, d in enumerate(defs):
    t = d["type"]
    if t == "convolutional":
        k, s = int(d["size"]), int(d["stride"])
        f = d["filters"]
        Co = (int(hyper["classes"]) + 5) * 3 if f == "preyolo" else int(f)
        Ho = (H + 2 * ((k - 1) // 2) - k) // s + 1
        shapes.append((H, C, Co, k, s))
        C, H = Co, Ho
    elif t == "upsample":
        H *= 2
    elif t == "route":
        ls = [int(v) for v in d["layers"].split(",")]
        ls = [l if l < 0 else l - i for l in ls]
        C = sum(outs[i + l][0] for l in ls); H = outs[i + ls[0]][1]
    outs.append((C, H))
cnt = collections.Counter(shapes)
e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p(); L.event_create(ctypes.byref(e0)); L.event_create(ctypes.byref(e1))
rows = []
for (H, Ci, Co, k, s), n in cnt.items():
    pad = (k - 1) // 2
    Ho = (H + 2 * pad - k) // s + 1
    Cip, Cop = (Ci + 7) // 8 * 8, (Co + 7) // 8 * 8
    xs = [torch.randn(B * H * H * Cip, device="cuda").to(torch.bfloat16) for _ in range(3)]
    ys = [torch.randn(B * Ho * Ho * Cop, device="cuda").to(torch.bfloat16) for _ in range(3)]
    wf = (torch.randn(Cop * k * k * Cip, device="cuda") * 0.05).to(torch.bfloat16)
    stt = torch.zeros(L.conv2d_stats_rows_geom(1, B, Ho, Ho, Cip, Cop, k, k, s, pad, 1, Cip) * 2 * Cop, device="cuda")
    res = []
    for mode in (0, 1):
        if mode == 1 and Ci == 3: res.append(0.0); continue
        def call(i):
            x, y = xs[i % 3], ys[i % 3]
            if mode == 0:
                return L.conv2d(1, 0, x.data_ptr(), Cip, wf.data_ptr(), y.data_ptr(), Cop, None, None, 0, stt.data_ptr(), B, H, H, Cip, Ho, Ho, Cop, k, k, s, pad, 1, st)
            return L.conv2d(1, 1, y.data_ptr(), Cop, wf.data_ptr(), x.data_ptr(), Cip, None, None, 0, None, B, Ho, Ho, Cop, H, H, Cip, k, k, s, pad, 1, st)
        for i in range(5): assert call(i) == 0, (H, Ci, Co, k, s, mode)
        L.event_record(e0, st)
        for i in range(30): call(i)
        L.event_record(e1, st); L.event_sync(e1)
        ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms)); res.append(ms.value / 30)
    fl = 2.0 * B * Ho * Ho * Co * k * k * Ci
    byt = 2.0 * B * (H * H * Cip + Ho * Ho * Cop)
    rows.append(((res[0] + res[1]) * n, n, (H, Ci, Co, k, s), res[0], res[1], fl, byt))
rows.sort(reverse=True)
print("total isolated fwd+dgrad ms/step: %.3f" % sum(r[0] for r in rows))
for tt, n, sh, tf, tb, fl, byt in rows:
    print("%6.3f ms (%2d x)  %-24s fwd %6.1f us %5.0f TF/s %5.0f GB/s | dgrad %6.1f us %5.0f TF/s" %
          (tt, n, sh, tf * 1e3, fl / tf / 1e9, byt / tf / 1e6, tb * 1e3, (fl / tb / 1e9) if tb else 0))
