"""Isolated weight-gradient time of every conv layer of yolo_baseline @416, batch 32 (bf16): where do the wgrad milliseconds go?
usage: wgrad_layers.py [variant]"""
import ctypes, os, sys, tempfile, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdcv import _lib
from mdcv.yolo.utils.parse_config import parse_model_config
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
L.conv2d_wgrad_set_variant(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
tmp = tempfile.mkdtemp()
defs = parse_model_config(bench.write_yolo_cfg(tmp))
hyper = defs.pop(0)
B, S = 32, 416
shapes, outs = [], []            # trace (C, H) through the cfg
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
    elif t == "shortcut":
        pass
    elif t == "yolo":
        pass
    outs.append((C, H))
cnt = collections.Counter(shapes)
e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p(); L.event_create(ctypes.byref(e0)); L.event_create(ctypes.byref(e1))
rows = []
for (H, Ci, Co, k, s), n in cnt.items():
    pad = (k - 1) // 2
    Ho = (H + 2 * pad - k) // s + 1
    Cip, Cop = (Ci + 7) // 8 * 8, (Co + 7) // 8 * 8
    xs = [torch.randn(B * H * H * Cip, device="cuda").to(torch.bfloat16) for _ in range(3)]
    dys = [torch.randn(B * Ho * Ho * Cop, device="cuda").to(torch.bfloat16) for _ in range(3)]
    sp = L.conv2d_wgrad_splits_geom(1, B, H, H, Cip, Ho, Ho, Cop, k, k, s, pad, 1, Cop, Cip)
    ws = torch.empty(sp * Cop * k * k * Cip, device="cuda"); dw = torch.empty(Co * Ci * k * k, device="cuda")
    def call(i):
        return L.conv2d_wgrad(1, dys[i % 3].data_ptr(), Cop, xs[i % 3].data_ptr(), Cip, ws.data_ptr(), sp, dw.data_ptr(), 0, B, H, H, Cip, Ci, Ho, Ho, Cop, Co, k, k, s, pad, 1, st)
    for i in range(5): assert call(i) == 0, (H, Ci, Co, k, s)
    L.event_record(e0, st)
    for i in range(30): call(i)
    L.event_record(e1, st); L.event_sync(e1)
    ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms)); t = ms.value / 30
    fl = 2.0 * B * Ho * Ho * Co * k * k * Ci
    byt = 2.0 * B * (H * H * Cip + Ho * Ho * Cop)
    rows.append((t * n, n, (H, Ci, Co, k, s), t, fl / t / 1e9, byt / t / 1e6, sp))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total isolated wgrad ms/step: %.3f" % tot)
for tt, n, sh, t, tf, gb, sp in rows:
    print("%6.3f ms (%2d x %6.1f us)  %-24s %5.0f TF/s  %5.0f GB/s operand  splits %d" % (tt, n, t * 1e3, sh, tf, gb, sp))
