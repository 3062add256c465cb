"""Debug: batched pack kernel vs torch reference for the full YOLOv3 layer list, two rounds with different buffer addresses."""
import os, sys, struct, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mdcv import _lib
from mdcv.yolo.models import Darknet
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
tmp = tempfile.mkdtemp(); cfg = bench.write_yolo_cfg(tmp)
cwd = os.getcwd(); os.chdir(tmp); torch.manual_seed(0)
net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision="bf16"); os.chdir(cwd)
convs = [m[0] for m in net.module_list if hasattr(m, "__getitem__") and len(m) and isinstance(m[0], torch.nn.Conv2d)]
ws = [c.weight.detach().cuda().contiguous() for c in convs]
pad8 = lambda c: (c + 7) // 8 * 8
for rnd in range(3):
    junk = torch.empty(1536 * (rnd + 1) + 512 * rnd, dtype=torch.uint8, device="cuda")     # shifts the addresses of what follows
    recs, outs = [], []
    for w in ws:
        Co, Ci, kh, kw = w.shape
        cop, cip = pad8(Co), pad8(Ci)
        wf = torch.full((cop * kh * kw * cip,), 7.0, dtype=torch.bfloat16, device="cuda")
        wd = torch.full((cip * kh * kw * cop,), 7.0, dtype=torch.bfloat16, device="cuda")
        outs.append((wf, wd, cop, cip))
        recs.append(struct.pack("<QQQiiiiiiiiQQ", w.data_ptr(), wf.data_ptr(), wd.data_ptr(), Co, Ci, kh * kw, cop, cip, 0, 0, 0, 0, 0))
    table = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).cuda()
    eq = max(1, max((min(64, o[3]) * w.shape[2] * w.shape[3] + 63) // 64 for w, o in zip(ws, outs)))
    assert L.pack_weights_batched(1, table.data_ptr(), len(ws), eq, st) == 0
    torch.cuda.synchronize()
    nbad = 0
    for li, (w, (wf, wd, cop, cip)) in enumerate(zip(ws, outs)):
        Co, Ci, kh, kw = w.shape
        ref_f = torch.zeros(cop, kh * kw, cip, device="cuda"); ref_f[:Co, :, :Ci] = w.reshape(Co, Ci, kh * kw).permute(0, 2, 1)
        ref_d = torch.zeros(cip, kh * kw, cop, device="cuda"); ref_d[:Ci, :, :Co] = w.reshape(Co, Ci, kh * kw).permute(1, 2, 0)
        okf = torch.equal(wf.float(), ref_f.to(torch.bfloat16).float().reshape(-1)); okd = torch.equal(wd.float(), ref_d.to(torch.bfloat16).float().reshape(-1))
        if not (okf and okd):
            nbad += 1
            if nbad <= 4:
                bad = (wf.float() != ref_f.to(torch.bfloat16).float().reshape(-1)).nonzero().flatten()
                print("round", rnd, "layer", li, tuple(w.shape), "wf ok", okf, "wd ok", okd, "wf ptr%4096", wf.data_ptr() % 4096, "n bad", bad.numel(), "first bad (co,tap,ci)", [(int(j) // (kh * kw * cip), int(j) // cip % (kh * kw), int(j) % cip) for j in bad[:4]])
    print("round", rnd, "bad layers", nbad, "of", len(ws))
