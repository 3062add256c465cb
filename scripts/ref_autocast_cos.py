"""What bf16 costs the REFERENCE's own arithmetic: the CPU oracle (same torch ops as the reference's Darknet) run once in fp32 and once under
torch.autocast(bfloat16) on the same weights and batch; cosine of the conv weight gradients, layer by layer.  The HIP bf16 mode is held
against this curve.  CPU only.  usage: ref_autocast_cos.py [batch=4] [size=416] [out.json]
(tests/golden/yolo_autocast_bf16_cos.json = `ref_autocast_cos.py 32 416 tests/golden/yolo_autocast_bf16_cos.json`, the weights / batch / targets of
tests/test_gpu_models.py::test_full_yolov3_batch32_train_forward_backward_vs_oracle)"""
import json, os, sys, tempfile, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oracle import yolo_oracle as yo
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 416
tmp = tempfile.mkdtemp(); cfg = bench.write_yolo_cfg(tmp)
cwd = os.getcwd(); os.chdir(tmp)
orc = yo.DarknetOracle(cfg, anchors=yo.VANILLA_ANCHORS, seed=3)
os.chdir(cwd)
g = torch.Generator().manual_seed(21)
x = torch.rand(B, 3, S, S, generator=g); tg = bench.synth_targets(B, 16, g)
for k in orc.trainable(): orc.params[k].requires_grad_(True)
grads, losses = {}, {}
torch.set_num_threads(min(os.cpu_count() or 1, 32))
for mode in ("fp32", "bf16"):
    for k in orc.trainable(): orc.params[k].grad = None
    snap = {k: v.clone() for k, v in orc.params.items() if "running" in k}
    t0 = time.time()
    if mode == "bf16":
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = orc.forward(x, tg)
    else:
        out = orc.forward(x, tg)
    out[0].sum().backward()
    for k, v in snap.items(): orc.params[k].copy_(v)
    losses[mode] = float(out[0].sum().detach())
    print(mode, "loss %.4f  (%.1f s)" % (float(out[0].sum().detach()), time.time() - t0), flush=True)
    grads[mode] = {k: orc.params[k].grad.detach().double().reshape(-1).clone() for k in orc.trainable() if k.endswith("weight") and k.startswith("conv")}
for k in grads["fp32"]:
    a, b = grads["fp32"][k], grads["bf16"][k]
    print("  %-16s cos %.4f  |g| ratio %.3f" % (k, float(a @ b / (a.norm() * b.norm() + 1e-30)), float(b.norm() / (a.norm() + 1e-30))))
if len(sys.argv) > 3:
    cosd = {k: round(float(grads["fp32"][k] @ grads["bf16"][k] / (grads["fp32"][k].norm() * grads["bf16"][k].norm() + 1e-30)), 4) for k in grads["fp32"]}
    json.dump({"what": "cosine(conv weight gradient under torch.autocast(cpu, bfloat16), same in fp32) of the CPU oracle, yolo_baseline",
               "generator": "scripts/ref_autocast_cos.py", "batch": B, "size": S, "oracle_seed": 3, "data_seed": 21, "targets_per_image": 16,
               "torch": torch.__version__, "loss": losses, "cos": cosd}, open(sys.argv[3], "w"), indent=1)
