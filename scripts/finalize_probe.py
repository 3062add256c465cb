"""Isolated time of mdcv_bn_stats_finalize for the (rows, C) pairs of YOLOv3 at batch 32 (back-to-back launches, HIP events)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p(); L.event_create(ctypes.byref(e0)); L.event_create(ctypes.byref(e1))
for rows, C in [(43264, 32), (10816, 64), (10816, 32), (2704, 128), (2704, 64), (703, 256), (676, 128), (122, 512), (169, 256), (49, 1024), (43, 512)]:
    part = torch.rand(rows * 2 * C, device="cuda"); acc = torch.zeros(3 * C, dtype=torch.float64, device="cuda")
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda"); rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda")
    outs = [torch.zeros(C, device="cuda") for _ in range(4)]
    def call():
        return L.bn_stats_finalize(part.data_ptr(), rows, acc.data_ptr(), float(rows * 128), g.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                   0.1, 1e-5, *[o.data_ptr() for o in outs], C, st)
    for _ in range(5): assert call() == 0
    L.event_record(e0, st)
    for _ in range(200): call()
    L.event_record(e1, st); L.event_sync(e1)
    ms = ctypes.c_float(); L.event_elapsed_ms(e0, e1, ctypes.byref(ms))
    print("rows %6d C %5d : %.2f us per launch (back to back)" % (rows, C, 1e3 * ms.value / 200))
