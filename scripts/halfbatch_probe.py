"""Would splitting a layer into two half-batch chains on two streams overlap the HBM-bound BN pass with the MFMA-bound conv?
(A) one stream: bn_act_fwd(full) -> conv3x3(full)   (B) two streams: [bn_act(h1) -> conv(h1)] || [bn_act(h2) -> conv(h2)]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdcv import _lib
L = _lib.lib()
def run(H, Ci, Co, B=32, iters=100):
    M = B * H * H
    y = torch.randn(M * Ci, device="cuda").to(torch.bfloat16)
    z = torch.empty(M * Ci, device="cuda", dtype=torch.bfloat16)
    o = torch.empty(M * Co, device="cuda", dtype=torch.bfloat16)
    w = (torch.randn(Co * 9 * Ci, device="cuda") * 0.05).to(torch.bfloat16)
    sc, sh = torch.rand(Ci, device="cuda") + 0.5, torch.randn(Ci, device="cuda")
    rows = L.conv2d_stats_rows_geom(1, B, H, H, Ci, Co, 3, 3, 1, 1, 1, Ci)
    stt = torch.zeros(2 * (rows + 8) * 2 * Co, device="cuda")
    s0 = torch.cuda.current_stream()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def chain(st, b0, nb, stats_off):
        m0 = b0 * H * H
        L.bn_act_fwd(1, y.data_ptr() + m0 * Ci * 2, Ci, sc.data_ptr(), sh.data_ptr(), None, 0, None, None, None, 0, z.data_ptr() + m0 * Ci * 2, Ci, nb * H * H, Ci, 1, 0.1, st.cuda_stream)
        L.conv2d(1, 0, z.data_ptr() + m0 * Ci * 2, Ci, w.data_ptr(), o.data_ptr() + m0 * Co * 2, Co, None, None, 0, stt.data_ptr() + stats_off * 4, nb, H, H, Ci, H, H, Co, 3, 3, 1, 1, 1, st.cuda_stream)
    def A():
        chain(s0, 0, B, 0)
    def Bv():
        s1.wait_stream(s0); s2.wait_stream(s0)
        chain(s1, 0, B // 2, 0)
        chain(s2, B // 2, B // 2, (rows + 8) * 2 * Co)
        s0.wait_stream(s1); s0.wait_stream(s2)
    res = {}
    for name, fn in (("one-stream", A), ("two-half-streams", Bv)):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / iters * 1e3
    print((H, Ci, Co), {k: round(v, 1) for k, v in res.items()}, "us")
for sh in [(52, 128, 256), (26, 256, 512), (13, 512, 1024), (104, 64, 128)]:
    run(*sh)
