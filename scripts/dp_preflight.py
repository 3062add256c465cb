#!/usr/bin/env python3
"""Pre-flight of the data-parallel path on a multi-GPU node: so that a failing N-GPU lease produces a DIAGNOSIS, not a return code.

    python scripts/dp_preflight.py --gpus 8          (spawns its ranks through torch.distributed.run, 127.0.0.1 rendezvous)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 scripts/dp_preflight.py

Per rank, in order, each step reported with its wall time and the exception text if it fails:
  1. environment      HSA_ENABLE_IPC_MODE_LEGACY (RCCL needs dmabuf IPC on this driver), GPU_MAX_HW_QUEUES, visible devices, device name / CUs
  2. comm init        torch.distributed.init_process_group("nccl") == RCCL, then a 4-byte all-reduce (the first collective builds the rings)
  3. all-reduce       the YOLOv3 flat gradient's size (62 M fp32 = 248 MB) as four 62 MB buckets, SUM, 5 timed repeats: ms and the
                      algorithm bandwidth 2 (N-1)/N x bytes / t per rank (xGMI: 7 links x ~153 GB/s per GPU; a ring is per-link bound)
  4. correctness      every element == sum of the ranks' fill values; CRC of the reduced buffer identical on every rank
  5. model replicas   (--model) one Darknet mini step per rank through GradAllReducer: replica checksums of the parameters agree
Rank 0 prints one JSON object; exit code 0 only if every rank passed every step."""
import argparse
import json
import os
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="spawn this many ranks (omit under torchrun)")
    ap.add_argument("--mb", type=float, default=248.0, help="bytes all-reduced per repeat, in MB")
    ap.add_argument("--buckets", type=int, default=4)
    ap.add_argument("--backend", default=os.environ.get("MDCV_DIST_BACKEND", "nccl"))
    ap.add_argument("--model", action="store_true", help="also run one data-parallel training step of the mini Darknet")
    a = ap.parse_args()
    if "RANK" not in os.environ:
        n = a.gpus or 1
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port",
               str(29500 + os.getpid() % 400), os.path.abspath(__file__), "--mb", str(a.mb), "--buckets", str(a.buckets), "--backend", a.backend]
        if a.model:
            cmd.append("--model")
        sys.exit(subprocess.call(cmd, env=os.environ.copy()))

    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    rep = {"rank": rank, "steps": []}

    def step(name, fn):
        t0 = time.time()
        try:
            out = fn()
            rep["steps"].append({"step": name, "ok": True, "s": round(time.time() - t0, 3), **(out or {})})
            return True
        except Exception as e:                                 # noqa: BLE001  (the point of this script is to report whatever goes wrong)
            import traceback
            rep["steps"].append({"step": name, "ok": False, "s": round(time.time() - t0, 3), "error": repr(e), "trace": traceback.format_exc()[-1500:]})
            return False

    def env():
        ndev = torch.cuda.device_count()
        if ndev <= local and a.backend == "nccl":
            raise RuntimeError(f"rank {rank}: LOCAL_RANK {local} but only {ndev} visible device(s) -- RCCL needs one GPU per rank")
        torch.cuda.set_device(local if ndev > local else 0)
        p = torch.cuda.get_device_properties(torch.cuda.current_device())
        return {"visible_devices": ndev, "device": p.name, "cus": p.multi_processor_count, "hbm_gb": round(p.total_memory / 2 ** 30, 1),
                "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES"), "torch": torch.__version__}

    def init():
        dist.init_process_group(a.backend, rank=rank, world_size=world)
        t = torch.ones(1, device="cuda")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        if int(t.item()) != world:
            raise RuntimeError(f"4-byte all-reduce returned {t.item()}, expected {world}")

    state = {}

    def allreduce():
        n = int(a.mb * (1 << 20) / 4)
        buf = torch.empty(n, device="cuda")
        per = (n + a.buckets - 1) // a.buckets
        times = []
        for it in range(6):
            buf.fill_(float(rank + 1))
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.time()
            for b in range(a.buckets):
                dist.all_reduce(buf[b * per:min(n, (b + 1) * per)], op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
            times.append(time.time() - t0)
        state["buf"] = buf
        ms = sorted(times[1:])[len(times[1:]) // 2] * 1e3
        return {"mb": a.mb, "buckets": a.buckets, "ms_median": round(ms, 3), "ms_first": round(times[0] * 1e3, 3),
                "algbw_gbs": round(n * 4 / (ms * 1e-3) / 1e9, 1), "busbw_gbs": round(2 * (world - 1) / world * n * 4 / (ms * 1e-3) / 1e9, 1)}

    def correct():
        buf = state["buf"]
        want = float(world * (world + 1) / 2)
        bad = int((buf != want).sum())
        crc = zlib.crc32(buf[: 1 << 20].cpu().numpy().tobytes())
        crcs = [None] * world
        dist.all_gather_object(crcs, crc)
        if bad or len(set(crcs)) != 1:
            raise RuntimeError(f"{bad} wrong elements (expected {want}); CRCs per rank {crcs}")
        return {"value": want, "crc": crc}

    def model():
        sys.path.insert(0, ROOT)
        import numpy as np
        from mdcv.yolo.models import Darknet
        from mdcv.optim import FusedAdam
        from mdcv.parallel import GradAllReducer, shard_batch
        g = os.path.join(ROOT, "tests", "golden")
        z = np.load(os.path.join(g, "mini_darknet_dp.npz"))
        cwd = os.getcwd()
        os.chdir(os.path.join(g, "mini"))
        try:
            net = Darknet("mini.cfg", 2.0, 1.6, 25.0, 0.1, False)
            net.load_weights("mini.weights", net.get_start_weight_dim())
        finally:
            os.chdir(cwd)
        net = net.cuda().train()
        red = GradAllReducer.attach(net, bucket_mb=0.05)
        opt = FusedAdam(net, lr=1e-3)
        w = world if 8 % world == 0 else 1
        x = shard_batch(torch.from_numpy(z["x"]), rank % w, w).cuda()
        tg = shard_batch(torch.from_numpy(z["targets"]), rank % w, w).cuda()
        for _ in range(3):
            opt.zero_grad()
            out = net(x, tg)
            out[0].sum().backward()
            red.finish()
            opt.step()
        torch.cuda.synchronize()
        crc = zlib.crc32(net.flat_parameters()[0].cpu().numpy().tobytes())
        crcs = [None] * world
        dist.all_gather_object(crcs, crc)
        if len(set(crcs)) != 1:
            raise RuntimeError(f"replicas diverged after 3 steps: parameter CRCs {crcs}")
        return {"loss": float(out[0]), "param_crc": crc}

    ok = step("environment", env) and step("comm_init", init) and step("allreduce", allreduce) and step("correctness", correct)
    if ok and a.model:
        ok = step("model_replicas", model)
    rep["ok"] = ok
    allrep = [None] * world
    try:
        if dist.is_initialized():
            dist.all_gather_object(allrep, rep)
        else:
            allrep = [rep]
    except Exception:                                          # noqa: BLE001
        allrep = [rep]
    if rank == 0 or not dist.is_initialized():
        good = all(r is not None and r.get("ok") for r in allrep) and len([r for r in allrep if r]) == (world if dist.is_initialized() else 1)
        print(json.dumps({"world": world, "backend": a.backend, "ok": good, "ranks": allrep}, indent=1))
    if dist.is_initialized():
        dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
