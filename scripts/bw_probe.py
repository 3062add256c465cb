"""Achievable HBM bandwidth on this box: torch copy / add vs the BN strip kernels, several tensor sizes (bf16)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for (M, C) in [(86528, 256), (86528, 128), (21632, 512), (21632, 256), (5408, 1024), (346112, 128), (1384448, 64)]:
    n = M * C
    xs = [torch.randn(n, device="cuda").to(torch.bfloat16) for _ in range(6)]
    ys = [torch.empty(n, device="cuda", dtype=torch.bfloat16) for _ in range(6)]
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    i = [0]
    def copy(): i[0] += 1; ys[i[0] % 6].copy_(xs[i[0] % 6])
    def add(): i[0] += 1; torch.add(xs[i[0] % 6], xs[(i[0] + 1) % 6], out=ys[i[0] % 6])
    def bnf(): i[0] += 1; L.bn_act_fwd(1, xs[i[0] % 6].data_ptr(), C, sc.data_ptr(), sh.data_ptr(), None, 0, None, None, None, 0, ys[i[0] % 6].data_ptr(), C, M, C, 1, 0.1, st)
    def bnfr(): i[0] += 1; L.bn_act_fwd(1, xs[i[0] % 6].data_ptr(), C, sc.data_ptr(), sh.data_ptr(), None, 0, None, None, xs[(i[0] + 1) % 6].data_ptr(), C, ys[i[0] % 6].data_ptr(), C, M, C, 1, 0.1, st)
    b = n * 2
    print("M=%7d C=%4d (%5.1f MB/tensor): copy %.2f TB/s (%.1f us) | add(3 streams) %.2f TB/s | bn_act_fwd %.2f TB/s (%.1f us) | bn_act_fwd+resid %.2f TB/s (%.1f us)" % (
        M, C, b / 1e6, 2 * b / timeit(copy) / 1e12, timeit(copy) * 1e6, 3 * b / timeit(add) / 1e12, 2 * b / timeit(bnf) / 1e12, timeit(bnf) * 1e6,
        3 * b / timeit(bnfr) / 1e12, timeit(bnfr) * 1e6), flush=True)
