"""Per-step cycle stamps of one wave of the shift kernel (library built with -DMDCV_SHIFT_TS): where a K step spends its time.
columns: wait(vmcnt) | barrier | reads+issue+MFMA issue | loop-back   (shader cycles, medians over the steps of one tile)"""
import ctypes, os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
B, H, Ci, Co = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 52, int(sys.argv[2]) if len(sys.argv) > 2 else 128, int(sys.argv[3]) if len(sys.argv) > 3 else 256
x = torch.randn(B * H * H * Ci, device="cuda").to(torch.bfloat16)
y = torch.randn(B * H * H * Co, device="cuda").to(torch.bfloat16)
wf = (torch.randn(Co * 9 * Ci, device="cuda") * 0.05).to(torch.bfloat16)
for i in range(20):
    assert L.conv2d(1, 0, x.data_ptr(), Ci, wf.data_ptr(), y.data_ptr(), Co, None, None, 0, None, B, H, H, Ci, H, H, Co, 3, 3, 1, 1, 1, st) == 0
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (4 * 512))()
f = L.cdll.mdcv_debug_shift_ts
f.argtypes = [ctypes.c_void_p]
assert f(buf) == 0
n = Ci // 32 * 9
t = [[buf[k * 512 + i] for i in range(n)] for k in range(4)]
wait = [t[1][i] - t[0][i] for i in range(n)]
bar = [t[2][i] - t[1][i] for i in range(n)]
body = [t[3][i] - t[2][i] for i in range(n)]
loop = [t[0][i + 1] - t[3][i] for i in range(n - 1)]
step = [t[0][i + 1] - t[0][i] for i in range(n - 1)]
med = statistics.median
print("steps", n, "| step %.0f | wait %.0f  barrier %.0f  body %.0f  loop %.0f (medians)" % (med(step), med(wait), med(bar), med(body), med(loop)))
print("means: step %.0f wait %.0f barrier %.0f body %.0f" % (sum(step) / len(step), sum(wait) / n, sum(bar) / n, sum(body) / n))
print("first 12 steps (wait, barrier, body):", [(wait[i], bar[i], body[i]) for i in range(12)])
# effective shader clock of the instrumented wave: s_memtime ticks per wall-clock (100 MHz) tick over the K loop of its tile
buf2 = (ctypes.c_longlong * (3 * 4096))()
f2 = L.cdll.mdcv_debug_shift_wg; f2.argtypes = [ctypes.c_void_p]
assert f2(buf2) == 0
wall = (buf2[4096 + 300] - buf2[300]) / 100.0        # us, whole tile 300
cyc = t[3][n - 1] - t[0][0]
print("tile 300: wall %.2f us, K-loop %d cycles -> if the loop were the whole tile: %.0f MHz (lower bound of the shader clock)" % (wall, cyc, cyc / wall))
