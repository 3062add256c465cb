"""The first conv's weight gradient with the BatchNorm-backward apply in its operand load (mdcv_conv2d_wgrad_bnapply), alone: ring stages x block
target.   usage: bna_ab.py [B=32] [H=416]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mdcv import _lib
L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = int(sys.argv[2]) if len(sys.argv) > 2 else 416
dev = "cuda"
Cin, Cinp, Cout = 3, 8, 32
M = B * H * H
g = torch.Generator(device=dev); g.manual_seed(1)
dz = (torch.randn(M, Cout, device=dev, generator=g) * 0.1).bfloat16()
y = torch.randn(M, Cout, device=dev, generator=g).bfloat16()
x = torch.randn(B * H * H, Cinp, device=dev, generator=g).bfloat16()
x[:, Cin:] = 0
v = [torch.randn(Cout, device=dev, generator=g) for _ in range(5)]
dw = torch.zeros(Cout, Cin, 3, 3, device=dev)
st = torch.cuda.current_stream().cuda_stream
BF16 = 1
geom = (B, H, H, Cinp, H, H, Cout, 3, 3, 1, 1, 1)
ws = torch.empty(4096 * Cout * 9 * Cinp, device=dev)
for slots in (512, 768, 1024, 1536):
    splits = int(L.conv2d_wgrad_splits_geom(_lib.tuned(BF16, 20000 + slots), *geom, Cout, Cinp))
    for stages in (2, 3, 4):
        dt = _lib.tuned(BF16, 34010 + stages)
        def run():
            rc = L.conv2d_wgrad_bnapply(dt, dz.data_ptr(), Cout, y.data_ptr(), Cout, *[t.data_ptr() for t in v], 1, 0.1, x.data_ptr(), Cinp,
                                        ws.data_ptr(), splits, dw.data_ptr(), 0, B, H, H, Cinp, Cin, H, H, Cout, Cout, 3, 3, 1, 1, 1, st)
            assert rc == 0, rc
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        gb = (2 * M * Cout * 2 + M * Cinp * 2) / 1e9
        print(f"slots {slots:5d} splits {splits:5d} stages {stages}: {us:7.1f} us  ({gb / us * 1e3:.2f} TB/s of dz + y + x)  checksum {float(dw.abs().sum()):.6g}")
