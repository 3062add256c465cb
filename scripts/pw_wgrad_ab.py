"""1x1 weight gradients of yolo_baseline @416 batch 32 alone (kernel + slab reduce), under tuning codes of mdcv_conv2d_wgrad_set_variant.
usage: pw_wgrad_ab.py [codes, e.g. 0,20256,4+20384]   ('+' joins codes applied together; every variant starts from 0 + 20512)
(round 3: 3- / 4- / 6-stage DMA rings of 64 / 32 pixels for these layers measured 0 ... +30 % against the 2-stage ring: alone, kernel + reduce are
20-25 us whatever the layer size -- two dependent launches, not the K loop)"""
import ctypes, os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
variants = (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")
iters, rounds = 50, 3
SHAPES = [(32, 52, 256, 128), (32, 52, 128, 256), (32, 26, 512, 256), (32, 26, 256, 512), (32, 13, 1024, 512), (32, 13, 512, 1024), (32, 104, 128, 64), (32, 26, 768, 256)]
e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p(); L.event_create(ctypes.byref(e0)); L.event_create(ctypes.byref(e1))
for (B, H, Ci, Co) in SHAPES:
    xs = [torch.randn(B * H * H * Ci, device="cuda").to(torch.bfloat16) for _ in range(3)]
    dys = [torch.randn(B * H * H * Co, device="cuda").to(torch.bfloat16) for _ in range(3)]
    res, outs = {}, {}
    for v in variants:
        L.conv2d_wgrad_set_variant(0); L.conv2d_wgrad_set_variant(20512)
        for c in v.split("+"): L.conv2d_wgrad_set_variant(int(c))
        splits = L.conv2d_wgrad_splits_geom(1, B, H, H, Ci, H, H, Co, 1, 1, 1, 0, 1, Co, Ci)
        ws = torch.empty(splits * Co * Ci, device="cuda")
        dw = torch.full((Co * Ci,), float("nan"), device="cuda")
        def call(i):
            return L.conv2d_wgrad(1, dys[i % 3].data_ptr(), Co, xs[i % 3].data_ptr(), Ci, ws.data_ptr(), splits, dw.data_ptr(), 0, B, H, H, Ci, Ci, H, H, Co, Co, 1, 1, 1, 0, 1, st)
        assert call(0) == 0
        torch.cuda.synchronize()
        outs[v] = dw.clone()
        for i in range(5): assert call(i) == 0
        ts = []
        for r in range(rounds):
            L.event_record(e0, st)
            for i in range(iters): call(i)
            L.event_record(e1, st); L.event_sync(e1)
            ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms)); ts.append(ms.value / iters)
        res[v] = (statistics.median(ts), splits)
    ref = outs[variants[0]]
    agree = " ".join("%s:%.0e" % (v, float((outs[v] - ref).abs().max() / ref.abs().max())) for v in variants[1:])
    mb = 2.0 * B * H * H * (Ci + Co) / 1e6
    print((B, H, Ci, Co), "%.0f MB" % mb, " | ".join("%s: %.1f us (s%d)" % (v, 1e3 * t, s) for v, (t, s) in res.items()), "|", agree, flush=True)
L.conv2d_wgrad_set_variant(0); L.conv2d_wgrad_set_variant(20512)
