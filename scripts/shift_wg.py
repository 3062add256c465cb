"""Workgroup timeline of one shift-kernel launch (library built with -DMDCV_SHIFT_TS): start/end per tile, CU residency."""
import ctypes, os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
B, H, Ci, Co = 32, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
L.conv2d_set_variant(int(sys.argv[4]) if len(sys.argv) > 4 else -8)
x = torch.randn(B * H * H * Ci, device="cuda").to(torch.bfloat16)
y = torch.randn(B * H * H * Co, device="cuda").to(torch.bfloat16)
wf = (torch.randn(Co * 9 * Ci, device="cuda") * 0.05).to(torch.bfloat16)
for i in range(30):
    assert L.conv2d(1, 0, x.data_ptr(), Ci, wf.data_ptr(), y.data_ptr(), Co, None, None, 0, None, B, H, H, Ci, H, H, Co, 3, 3, 1, 1, 1, st) == 0
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (3 * 4096))()
f = L.cdll.mdcv_debug_shift_wg; f.argtypes = [ctypes.c_void_p]
assert f(buf) == 0
nt = ((B * (H + 1) * (H + 1) + 255) // 256) * (Co // 128)
nt = min(nt, 4096)
s0 = [buf[i] for i in range(nt)]; e0 = [buf[4096 + i] for i in range(nt)]; hw = [buf[8192 + i] for i in range(nt)]
t0 = min(s0)
dur = [(e0[i] - s0[i]) / 100.0 for i in range(nt)]
print("tiles", nt, "kernel span %.1f us" % ((max(e0) - t0) / 100.0), "tile duration us: min %.1f med %.1f max %.1f" % (min(dur), sorted(dur)[nt // 2], max(dur)))
starts = sorted((s0[i] - t0) / 100.0 for i in range(nt))
print("start-time quantiles us:", [round(starts[int(q * (nt - 1))], 1) for q in (0, .25, .5, .7, .75, .8, .9, 1)])
cu = collections.Counter()
for i in range(nt):
    h = hw[i] & 0xffffffff; xcc = hw[i] >> 32
    cu_id = (h >> 8) & 0xf; se = (h >> 13) & 0x7; sh = (h >> 12) & 1
    cu[(xcc & 0xf, se, sh, cu_id)] += 1
print("distinct CUs", len(cu), "tiles per CU histogram", sorted(collections.Counter(cu.values()).items()))
late = [i for i in range(nt) if (s0[i] - t0) / 100.0 > 5]
if late:
    ld = [dur[i] for i in late]; ed = [dur[i] for i in range(nt) if i not in set(late)]
    print("first wave: n=%d mean dur %.1f | later: n=%d mean dur %.1f" % (len(ed), sum(ed) / len(ed), len(ld), sum(ld) / len(ld)))
