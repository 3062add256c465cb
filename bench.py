#!/usr/bin/env python3
"""Training-throughput benchmark of the MDCV hot path on MI355X (contract: see the task brief / DESIGN.md §measurement).

  python bench.py --gpus 1 --steps K --warmup W            # single GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = zero_grad -> forward -> backward -> (RCCL gradient all-reduce when N>1) -> optimizer step on one synthetic batch
already resident in HBM.  Primary workload: CVC-YOLOv3 (yolo_baseline topology) 416x416, classes=80, bf16, 32 images per GPU
(BASELINE.json configs[2]/[3]); secondary workload reported in the same JSON line: RektNet 80x80 bf16, 256 images per GPU
(configs[1]).  Weak scaling: per-GPU batch fixed.  Rank 0 prints ONE JSON line.
"""
import argparse
import contextlib
import json
import os
import sys
import tempfile
import time

if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    # Multi-rank runs add RCCL's stream (torch's ProcessGroupNCCL picks it) to the main, weight-gradient and comm streams.  HIP multiplexes
    # streams onto 4 hardware queues by default and two streams on one queue run serially: give the runtime more queues
    # BEFORE it initialises.  Stamped into the line (env_overrides).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    # RCCL between processes exchanges buffers through IPC handles; this image's host driver supports only the dmabuf flavour, and without the
    # switch below ncclCommInitRank / the first collective fails with `hipIpcGetMemHandle: invalid argument` (the RCCL tests set it for the same
    # reason, tests/test_gpu_models.py / test_gpu_dp.py).  Set BEFORE the HIP runtime initialises; stamped into the line too.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3        # MI355X fp32 MFMA (v_mfma_f32_16x16x4_f32: 256 FLOP/clk/CU x 256 CUs x 2.4 GHz), the parity-mode kernels
OPT_PIPELINE = False           # FusedAdam(pipeline=True), see mdcv/optim.py: bit-identical, measured neutral -> off
PEAK_HBM_GBS = 8000.0          # HBM3E spec
YOLO_TRAIN_GFLOP_PER_IMG = 197.59   # SURVEY.md §8d (conv only, fwd+dgrad+wgrad, classes=80, 416^2; classes=1: 195.87)
REKT_TRAIN_GFLOP_PER_IMG = 11.872   # head conv counted once
REKT_TRAIN_MB_PER_IMG = 57.8


def write_yolo_cfg(dirname, size=416, classes=80):
    """yolo_baseline topology, generated — the reference's cfg file does not travel."""
    head = (f"[net]\nwidth={size}\nheight={size}\nonnx_height={size}\nclasses={classes}\nchannels=3\n"
            "yolo_masks=6,7,8|3,4,5|0,1,2\nyolo_scales=32,16,8\nvalidate_uri=dataset/validate.csv\ntrain_uri=dataset/train.csv\n"
            "weights_uri=none\nstart_weights_dim=255,255,255\nnum_train_images=-1\nnum_validate_images=-1\nleaky_slope=0.1\n"
            "conv_activation=leaky\nbuild_targets_ignore_thresh=0.5\nconf_thresh=0.8\nnms_thresh=0.25\niou_thresh=0.5\n\n")

    def conv(f, k, s=1):
        return f"[convolutional]\nfilters={f}\nsize={k}\nstride={s}\n\n"

    def res(c, n):
        return "".join(conv(c // 2, 1) + conv(c, 3) + "[shortcut]\nfrom=-3\n\n" for _ in range(n))
    body = conv(32, 3) + conv(64, 3, 2) + res(64, 1) + conv(128, 3, 2) + res(128, 2) + conv(256, 3, 2) + res(256, 8)
    body += conv(512, 3, 2) + res(512, 8) + conv(1024, 3, 2) + res(1024, 4)
    body += "".join(conv(512, 1) + conv(1024, 3) for _ in range(3)) + conv("preyolo", 1) + "[yolo]\n\n"
    body += "[route]\nlayers=-4\n\n" + conv(256, 1) + "[upsample]\nstride=2\n\n[route]\nlayers=-1, 61\n\n"
    body += "".join(conv(256, 1) + conv(512, 3) for _ in range(3)) + conv("preyolo", 1) + "[yolo]\n\n"
    body += "[route]\nlayers=-4\n\n" + conv(128, 1) + "[upsample]\nstride=2\n\n[route]\nlayers=-1, 36\n\n"
    body += "".join(conv(128, 1) + conv(256, 3) for _ in range(3)) + conv("preyolo", 1) + "[yolo]\n"
    os.makedirs(os.path.join(dirname, "dataset"), exist_ok=True)
    with open(os.path.join(dirname, "dataset", "train.csv"), "w") as f:
        f.write('"10,13|16,30|33,23|30,61|62,45|59,119|116,90|156,198|373,326"\n')
    path = os.path.join(dirname, f"yolo_baseline_{size}.cfg")
    with open(path, "w") as f:
        f.write(head + body)
    return path


def synth_targets(B, T, gen):
    """[B,T,5]: 1..T cone-like boxes per image (cls 0, centre U(.05,.95), size U(.02,.30)), remaining rows zero."""
    t = torch.zeros(B, T, 5)
    for b in range(B):
        n = int(torch.randint(1, T + 1, (1,), generator=gen))
        t[b, :n, 1:3] = torch.rand(n, 2, generator=gen) * 0.9 + 0.05
        t[b, :n, 3:5] = torch.rand(n, 2, generator=gen) * 0.28 + 0.02
    return t


def timed_region(fn, steps, warmup, device, world):
    for _ in range(warmup):
        fn()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def conv_flops(args):
    """algorithmic FLOPs of one mdcv_conv2d launch (real work of the padded problem the kernel is given)."""
    B, Hin, Win, Cin, Hout, Wout, Nout, KH, KW = args[11], args[12], args[13], args[14], args[15], args[16], args[17], args[18], args[19]
    mode, stride = args[1], args[20]
    if mode == 0:
        return 2.0 * B * Hout * Wout * Nout * KH * KW * Cin
    return 2.0 * B * Hin * Win * Cin * KH * KW * Nout          # dgrad: one MAC per (dy pixel, tap, ci, co), stride-independent


LAUNCH_DUMP = []


def _es(dt):
    return 2 if dt == 1 else 4


def call_work(name, args):
    """(algorithmic FLOPs, algorithmic HBM bytes) of one library call of a launch list (0 where not modelled).
    Convolutions: 2*M*N*K of the problem the call is given (padded channels as the kernel sees them; a stride-2 data gradient counts
    one MAC per (dy pixel, tap, ci, co)).  BatchNorm / activation passes: every operand tensor once."""
    if name == "mdcv_conv2d":
        return conv_flops(args), 0.0
    if name == "mdcv_conv2d_xstats":         # forward conv with the statistics added to exact accumulators: (dt, x, ldx, w, y, ldy, bias, xacc, reps, B, Hin, Win, Cin, Hout, Wout, Cout, kh, kw, ...)
        B, Hin, Win, Cin, Hout, Wout, Nout, KH, KW = args[9:18]
        return 2.0 * B * Hout * Wout * Nout * KH * KW * Cin, 0.0
    if name in ("mdcv_first_conv_stats", "mdcv_first_conv_bn_act"):  # the first conv's two streaming passes: 3x3, 8 -> 32 channels; (..., B, H, W)
        px = float(args[-3]) * args[-2] * args[-1]
        return 2.0 * px * 32 * 72, px * (16.0 if name.endswith("stats") else 16.0 + 128.0)
    if name == "mdcv_conv2d_affine_act":     # inference conv + BatchNorm(running stats) + activation: (dt, x, ldx, w, out, ldo, scale, shift, resid, ldr, act, slope, B, H, W, Cin, Ho, Wo, Cout, kh, kw, ...)
        B, Hin, Win, Cin, Hout, Wout, Nout, KH, KW = args[12:21]
        return 2.0 * B * Hout * Wout * Nout * KH * KW * Cin, 0.0
    if name == "mdcv_conv2d_dgrad_bnsums":
        Bq, Hin, Win, Cin, Hout, Wout, Nout, KH, KW = args[8:17]
        return 2.0 * Bq * Hin * Win * Cin * KH * KW * Nout, 0.0
    if name == "mdcv_pw_bwd":                # 1x1 data gradient + weight-gradient slabs in one launch: (dt, dy, ldy, x, ldx, wd, dx, lddx, add, ldadd, ws, slabs, fy, ..., M, Cin, Cout)
        M, Cin, Cout = args[20], args[21], args[22]
        # algorithmic bytes: dy and x read, dx written, the residual gradient (addsrc) and the y of the fused BatchNorm sums read where present (bf16)
        nby = 2.0 * M * (Cout + 2 * Cin + (Cin if args[8] is not None else 0) + (Cin if args[12] is not None else 0))
        return 2.0 * 2.0 * M * Cin * Cout, nby
    if name in ("mdcv_pw_conv_fwd", "mdcv_pw_conv_fwd_xstats"):      # fused 1x1 forward block: reads y (+ residual), writes z and the conv output
        M, K, N = args[-3], args[-2], args[-1]
        return 2.0 * M * K * N, 2.0 * M * (2 * K + N)
    if name == "conv2d_wgrad":               # info = (B, Hin, Win, Cin_pad, Hout, Wout, Cout_pad, k, stride, splits)
        B, Hin, Win, Cin, Hout, Wout, Cout, k = args[:8]
        return 2.0 * B * Hout * Wout * Cout * k * k * Cin, 0.0
    if name == "mdcv_bn_act_fwd":            # (dt, y1, ld1, s1, b1, y2, ld2, s2, b2, resid, ldr, out, ldo, M, C, act, slope)
        n = 2 + (args[5] is not None) + (args[9] is not None)
        return 0.0, float(_es(args[0]) * args[13] * args[14] * n)
    if name == "mdcv_bn_act_bwd_apply":      # (dt, dout, ldd, y1, ld1, s1, b1, cA, cB, cC, dy1, ldy1, y2, ld2, ..., dy2, ldy2, M, C, act, slope)
        n = 3 + 2 * (args[12] is not None)
        return 0.0, float(_es(args[0]) * args[-4] * args[-3] * n)
    if name == "mdcv_bn_act_bwd_reduce_finalize":   # (dt, dout, ldd, y1, ld1, ..., y2 @11, ..., M @17, C @18, ...)
        n = 2 + (args[11] is not None)
        return 0.0, float(_es(args[0]) * args[17] * args[18] * n)
    return 0.0, 0.0


def _is_main_kernel(sym):
    """the kernel of a multi-kernel call that does the call's algorithmic work (not a slab / partial-row reduction or finalize)"""
    return not any(t in sym for t in ("wgrad_reduce", "colfinal", "finalize", "rows_fold", "partial_reduce"))


def short_symbol(sym):
    """`void (anonymous namespace)::conv_glds_kernel<...>((anonymous namespace)::ConvArgs, ...)` -> `conv_glds_kernel<...>`"""
    s = sym.replace("(anonymous namespace)::", "")
    if s.startswith("void "):
        s = s[5:]
    depth = 0
    for i, ch in enumerate(s):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return s[:i]
    return s


def kernel_breakdown(model, plan, step_fn):
    """One extra, untimed, SERIAL step (weight gradients on the main stream too) with HIP events on the launch stream around every
    library call and, inside the library, around every kernel (engine.run_timed(kernels=True)).
    Returns (per-call {name: [calls, ms, flops]}, per-kernel {symbol: dict(launches, ms, flops, bytes)})."""
    from mdcv.engine import run_timed
    rec, krec = {}, {}

    def add(lst):
        for name, ms, args, kern in lst:
            fl, by = call_work(name, args)
            r = rec.setdefault(name, [0, 0.0, 0.0])
            r[0] += 1
            r[1] += ms
            r[2] += fl
            main = [(k, t) for k, t in kern if _is_main_kernel(k)]
            tmain = sum(t for _, t in main) or 1.0
            for k, t in kern:
                e = krec.setdefault(k, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
                e["launches"] += 1
                e["ms"] += t
                if (k, t) in main:                     # a call's work goes to its main kernel(s), split by duration when there are several
                    e["flops"] += fl * t / tmain       # (the four parity-class launches of a stride-2 data gradient)
                    e["bytes"] += by * t / tmain
            if name == "mdcv_conv2d":
                LAUNCH_DUMP.append((name, ms, [int(v) if isinstance(v, int) else 0 for v in args[11:23]] + [int(args[1])]))
            elif name == "mdcv_conv2d_xstats":          # the forward convs of the training plans: same row format, mode 0
                LAUNCH_DUMP.append(("mdcv_conv2d", ms, [int(v) if isinstance(v, int) else 0 for v in args[9:21]] + [0]))
            elif name in ("mdcv_first_conv_stats", "mdcv_first_conv_bn_act"):   # (B, Hin, Win, Cin, Hout, Wout, Cout, kh, kw, stride, pad, dil), mode 0
                LAUNCH_DUMP.append(("mdcv_conv2d", ms, [int(args[-3]), int(args[-2]), int(args[-1]), 8, int(args[-2]), int(args[-1]), 32, 3, 3, 1, 1, 1, 0]))
            elif name == "mdcv_conv2d_dgrad_bnsums":      # data gradient with the producer's BatchNorm-backward sums in its store loop
                LAUNCH_DUMP.append((name, ms, [int(v) for v in args[8:20]] + [1]))
            elif name == "conv2d_wgrad":
                LAUNCH_DUMP.append((name, ms, list(args), [(short_symbol(k), t) for k, t in kern]))
            elif name == "mdcv_pw_bwd":
                LAUNCH_DUMP.append((name, ms, [int(args[20]), int(args[21]), int(args[22]), int(args[11]), int(args[8] is not None), int(args[12] is not None)]))
            elif name.startswith("mdcv_bn_act") or name in ("mdcv_partial_reduce",):
                LAUNCH_DUMP.append((name, ms, [int(v) for v in args if isinstance(v, int) and 0 < v < (1 << 31)][-6:]))

    def run_and_time(lst, stream=None):
        add(run_timed(plan, lst, stream, kernels=True))
    plan.run = run_and_time
    g = plan.use_graph
    plan.use_graph = False
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        plan.__dict__.pop("run", None)             # (an instance attribute `run` keeps run_bwd_list serial: remove it, do not re-assign)
        plan.use_graph = g
    return rec, krec


def in_step_kernel_times(step_fn, steps=3):
    """Per-kernel durations INSIDE the normal training step, measured live: the in-library profiler (csrc/runtime.hip) binds a start /
    stop HIP event pair to every dispatch on the stream it is launched on (main stream and the weight-gradient side stream alike),
    so kernels of the two streams share the CUs exactly as in the timed region.  -> {short symbol: [launches per step, ms per step]}"""
    import ctypes
    from mdcv import _lib
    L = _lib.lib()
    step_fn()
    torch.cuda.synchronize()
    L.check(L.profile_begin(), "profile_begin")
    try:
        for _ in range(steps):
            step_fn()
        torch.cuda.synchronize()
    finally:
        nrec = L.profile_stop()
    if nrec < 0:
        raise _lib.MdcvError(f"mdcv_profile_stop failed ({nrec})")
    out = {}
    buf = ctypes.create_string_buffer(1024)
    ms = ctypes.c_float()
    for i in range(nrec):
        L.check(L.profile_read(i, ctypes.byref(ms), buf, 1024), "profile_read")
        e = out.setdefault(short_symbol(buf.value.decode()), [0.0, 0.0])
        e[0] += 1.0 / steps
        e[1] += ms.value / steps
    L.profile_begin()            # drop the records (events) ...
    L.profile_stop()             # ... and leave the profiler off
    return out


def roofline_objects(krec, precision, traffic, in_step=None):
    """Every kernel symbol whose algorithmic work is modelled (MFMA-bound convolutions / weight gradients in TFLOP/s, HBM-bound
    BatchNorm passes in GB/s), largest share of the step first.  Work per launch comes from one instrumented SERIAL step (krec: every
    kernel alone on the GPU -> `avg_us_alone`, `frac_alone` = the kernel's own quality).  `avg_us`, `achieved`, `frac` are the IN-STEP
    figures: the same start / stop event pairs taken live inside normal steps (in_step_kernel_times), where the side stream's weight
    gradients and the main stream's kernels share the CUs -- what the timed region contains and what a rocprofv3 kernel trace of this
    command reports.  Returns (dominant row, all rows)."""
    peak_f = PEAK_BF16_TFLOPS if precision == "bf16" else PEAK_F32_TFLOPS
    rows = []
    for sym, e in krec.items():
        if e["launches"] == 0 or e["ms"] <= 0 or (e["flops"] == 0 and e["bytes"] == 0):
            continue
        # the roof a kernel is actually against: with both figures modelled, HBM when its algorithmic bytes at the achievable
        # 6.3 TB/s take longer than its FLOPs at the dense peak (the one-launch 1x1 backward, the fused 1x1 forward block)
        mf = e["flops"] > 0 and not (e["bytes"] > 0 and e["bytes"] / 6.3e12 > e["flops"] / (peak_f * 1e12))
        work, div = (e["flops"], 1e12) if mf else (e["bytes"], 1e9)
        peak = peak_f if mf else PEAK_HBM_GBS
        per_launch = work / e["launches"]
        us_alone = 1e3 * e["ms"] / e["launches"]
        row = {"kernel": short_symbol(sym), "bound": "mfma" if mf else "hbm", "launches": e["launches"], "peak": peak,
               "unit": "TFLOP/s" if mf else "GB/s", ("flops_per_launch" if mf else "bytes_per_launch"): per_launch,
               "avg_us_alone": us_alone, "frac_alone": per_launch / (us_alone * 1e-6) / div / peak}
        st = (in_step or {}).get(short_symbol(sym))
        if st and st[0] > 0:
            us = 1e3 * st[1] / st[0]
            row.update(avg_us=us, achieved=per_launch / (us * 1e-6) / div, total_ms=st[1], timing="in-step (live HIP event pairs, both streams running)")
        else:
            row.update(avg_us=us_alone, achieved=per_launch / (us_alone * 1e-6) / div, total_ms=e["ms"], timing="alone (serial instrumented step)")
        row["frac"] = row["achieved"] / peak
        t = (traffic or {}).get("kernels", {}).get(short_symbol(sym)) if traffic else None
        row["traffic"] = (t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]) if t else None
        rows.append(row)
    rows.sort(key=lambda r: -r["total_ms"])
    top = dict(rows[0]) if rows else None
    return top, rows


def load_traffic(workload="yolo"):
    """Newest profiles/rNN_pmc_hbm_traffic.json (yolo) / rNN_rektnet_pmc_hbm_traffic.json (scripts/profile_round.sh) -- only if it was measured
    with THIS tree's kernels."""
    import glob
    from mdcv._fingerprint import kernel_fingerprint
    name = "pmc_hbm_traffic.json" if workload == "yolo" else f"{workload}_pmc_hbm_traffic.json"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + name)))
    if not files:
        return None, "no profiles/rNN_" + name
    t = json.load(open(files[-1]))
    fp = kernel_fingerprint()
    if t.get("fingerprint") != fp:
        return None, f"{os.path.basename(files[-1])} was measured with kernel fingerprint {t.get('fingerprint')}, this tree is {fp}: stale, not quoted"
    t["_file"] = os.path.basename(files[-1])
    return t, None


CPU_THREADS = 16     # measured on the GPU box host (256 logical cores): torch-CPU conv training peaks at 16 threads for these
                     # batch sizes (8: 0.082 s, 16: 0.053 s, 32: 0.103 s, 64: 0.20 s, 128: 0.6 s, 256: >10 s per RektNet B=8 step)


def cpu_baseline_yolo(cfg_path, workdir, budget_s=25.0):
    """CPU oracle ("port": plain torch-CPU restatement of the reference, pinned to it by tests/golden) on the host cores."""
    from oracle import yolo_oracle as yo
    torch.set_num_threads(min(os.cpu_count(), CPU_THREADS))
    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        orc = yo.DarknetOracle(cfg_path, anchors=yo.VANILLA_ANCHORS, seed=0)
    finally:
        os.chdir(cwd)
    B = 4                                   # the CPU leg of the YOLOv3 step is batch 4
    g = torch.Generator().manual_seed(17)
    x = torch.rand(B, 3, 416, 416, generator=g)
    tg = synth_targets(B, 16, g)
    params = [v.requires_grad_(True) for k, v in orc.trainable().items()]
    opt = torch.optim.Adam(params, lr=1e-3)
    times = []
    t_start = time.perf_counter()
    for it in range(5):                     # 1 warm-up + at least 3 timed steps, a 4th if the budget allows
        t0 = time.perf_counter()
        opt.zero_grad()
        out = orc.forward(x, tg)
        out[0].sum().backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s and it >= 3:
            break
    steady = times[1:] if len(times) > 1 else times
    return {"value": B / (sum(steady) / len(steady)), "unit": "images/sec", "cores": torch.get_num_threads(), "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"YOLOv3 416^2 classes=80 fp32 CPU oracle, batch {B}, {len(steady)} timed train steps after 1 warm-up (Adam)"}


def cpu_baseline_rektnet(budget_s=12.0):
    from oracle import rektnet_oracle as ro
    torch.set_num_threads(min(os.cpu_count(), CPU_THREADS))
    sd = ro.init_state(0)
    params = [v.requires_grad_(True) for k, v in sd.items() if "running" not in k]
    opt = torch.optim.Adam(params, lr=0.1)
    B = 8
    g = torch.Generator().manual_seed(17)
    x = torch.rand(B, 3, 80, 80, generator=g)
    tp = torch.rand(B, 7, 2, generator=g) * (79 / 80)
    times = []
    t_start = time.perf_counter()
    for it in range(8):
        t0 = time.perf_counter()
        opt.zero_grad()
        hm, pts = ro.keypoint_forward(x, sd, train=True)
        ro.cross_ratio_loss(hm, pts, None, tp, "l1_softargmax", True, 0.05, 0.05)[2].backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s and it >= 3:
            break
    steady = times[1:]
    return {"value": B / (sum(steady) / len(steady)), "unit": "images/sec", "cores": torch.get_num_threads(), "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"RektNet 80^2 fp32 CPU oracle, batch {B}, {len(steady)} timed train steps after 1 warm-up (Adam, l1_softargmax+geo)"}


def synth_eval_output(B, N, C, T, seed):
    """Eval-mode Darknet output rows around synthetic cone labels: 1-5 jittered high-confidence rows per label + clutter
    below the threshold, like a trained detector's output."""
    rng = np.random.default_rng(seed)
    tg = np.zeros((B, T, 5), np.float32)
    out = np.zeros((B, N, 5 + C), np.float32)
    for b in range(B):
        n = int(rng.integers(1, T + 1))
        tg[b, :n, 0] = rng.integers(0, C, n)
        tg[b, :n, 1:3] = rng.random((n, 2)) * 0.9 + 0.05
        tg[b, :n, 3:5] = rng.random((n, 2)) * 0.28 + 0.02
        out[b, :, 0:2] = rng.random((N, 2)) * 416
        out[b, :, 2:4] = rng.random((N, 2)) * 80 + 4
        out[b, :, 4] = rng.random(N) * 0.85
        out[b, :, 5:] = rng.random((N, C))
        rows = rng.permutation(N)
        r = 0
        for lab in tg[b, :n]:
            for _ in range(int(rng.integers(1, 6))):
                i = rows[r]; r += 1
                out[b, i, 0:4] = lab[1:5] * 416 * (1 + 0.08 * rng.standard_normal(4))
                out[b, i, 4] = 0.8 + 0.2 * rng.random()
    return out, tg


def cpu_baseline_post(out_np, tg_np, budget_s=10.0):
    """The numpy oracle of the per-image loop on this host (1 thread), bounded sample of the same batch."""
    from oracle import postprocess_oracle as PO
    t0 = time.perf_counter()
    n = 0
    for det, lab in zip(out_np, tg_np):
        PO.postprocess_image(det, lab, 0.8, 0.25, 0.5, 416, 416)
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "images/sec", "cores": 1, "kind": "port",
            "sample": f"numpy oracle of validate.py:80-141 on {n} images of the same batch ([10647, 85] rows each)"}


def cpu_baseline_joint(cfg_path, workdir, frames=1):
    """The chained CPU oracles of the joint path on the host cores: Darknet eval forward at 608^2 -> per-image conf filter / NMS ->
    8-bit crop + resize -> KeypointNet eval on the crops.  A random-init detector has no confident rows, so (as in the GPU leg) the
    stages behind it run on 16 synthetic cone boxes per frame."""
    from oracle import yolo_oracle as yo, postprocess_oracle as PO, pipeline_oracle as PL, rektnet_oracle as ro
    torch.set_num_threads(min(os.cpu_count(), CPU_THREADS))
    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        cfg608 = write_yolo_cfg(workdir, size=608, classes=int(open(cfg_path).read().split("classes=")[1].split()[0]))
        orc = yo.DarknetOracle(cfg608, anchors=yo.VANILLA_ANCHORS, seed=0)
    finally:
        os.chdir(cwd)
    rng = np.random.default_rng(5)
    f8 = rng.integers(0, 256, (frames, 3, 608, 608), dtype=np.uint8)
    x = torch.from_numpy((f8.astype(np.float32) / 255.0))
    sd = ro.init_state(0)
    t0 = time.perf_counter()
    with torch.no_grad():
        rows = orc.forward(x, None, bn_train=False).numpy()
    boxes = np.zeros((frames, 16, 4), np.float32)
    count = np.full(frames, 16, np.int32)
    for b in range(frames):
        c = rng.random((16, 2)) * 540 + 34
        wh = np.stack([rng.random(16) * 40 + 14, rng.random(16) * 60 + 20], 1)
        rows[b, :16, 0:2], rows[b, :16, 2:4], rows[b, :16, 4] = c, wh, 0.9
        r = PO.postprocess_image(rows[b], np.zeros((1, 5), np.float32), 0.8, 0.25, 0.5, 608, 608)
        n = min(r["count"], 16)
        boxes[b, :n], count[b] = r["boxes"][:n], n
    crops, _ = PL.crop_resize(f8, boxes, count, 80, 80, u8=True)
    with torch.no_grad():
        ro.keypoint_forward(torch.from_numpy(crops), sd, train=False)
    dt = time.perf_counter() - t0
    return {"value": frames / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"chained CPU oracles on {frames} frame(s) 608^2: Darknet eval -> NMS -> {int(count.sum())} u8 crops -> KeypointNet eval"}


MAX_LINE_BYTES = 3500      # the driver keeps an 8.4 KB tail of stdout and parses the last line whole: stay far below it


def _r(v, nd=4):
    """floats rounded to `nd` significant digits (the line is for reading and parsing, the detail file keeps full precision)"""
    if isinstance(v, float):
        return float(f"{v:.{nd}g}")
    if isinstance(v, dict):
        return {k: _r(x, nd) for k, x in v.items()}
    if isinstance(v, list):
        return [_r(x, nd) for x in v]
    return v


def compact(line):
    """The ONE stdout line: rounded, and -- should a future field push it over MAX_LINE_BYTES -- the optional blocks are dropped in a
    fixed order (never metric / value / ms_per_step / config / roofline / cpu_baseline)."""
    out = _r(line, 5)
    for k in ("value", "ms_per_step"):
        out[k] = line[k]
    for victim in ("workloads", "env_overrides", "detail"):
        if len(json.dumps(out)) <= MAX_LINE_BYTES:
            break
        out.pop(victim, None)
    assert len(json.dumps(out)) <= MAX_LINE_BYTES, len(json.dumps(out))
    return out


ROOFLINE_KEYS = ("kernel", "bound", "launches", "avg_us", "achieved", "peak", "unit", "frac", "timing", "avg_us_alone", "frac_alone",
                 "flops_per_launch", "bytes_per_launch", "traffic", "traffic_source")


def live_env_overrides():
    """MDCV_* variables that change what the timed step launches; stamped into the line."""
    return {k: v for k, v in sorted(os.environ.items()) if (k.startswith("MDCV_") and k not in ("MDCV_GRAPH", "MDCV_DIST_BACKEND")) or k in ("GPU_MAX_HW_QUEUES", "HSA_ENABLE_IPC_MODE_LEGACY")}


def build_line(a, world, primary, result, extra, cpu_baseline):
    from mdcv._fingerprint import kernel_fingerprint
    from mdcv import _lib
    roof = result.get("roofline")
    line = {
        "metric": ("images/sec training (YOLOv3 416^2 + RektNet 80^2)" if primary in ("yolo", "rektnet") else
                   "images/sec validation post-processing" if primary == "postprocess" else
                   "images/sec joint detect->keypoints inference"), "value": result["value"], "unit": "images/sec",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": result["ms_per_step"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
        "config": {"workload": {"yolo": "CVC-YOLOv3 yolo_baseline 416x416 classes=%d, train step (fwd+bwd+Adam), %d img/GPU" % (a.yolo_classes, a.yolo_batch),
                                "rektnet": "RektNet KeypointNet 80x80 train step (l1_softargmax+geo, Adam), %d img/GPU" % a.rekt_batch,
                                "postprocess": "validate.py per-image loop (conf 0.8, NMS 0.25 top-200, AP) on [%d,10647,85] eval outputs"
                                               % a.post_batch,
                                "joint": "YOLOv3 608x608 eval -> conf/NMS -> <=16 crops/frame 80x80 -> KeypointNet eval, %d frames/GPU"
                                         % a.joint_batch}[primary],
                   "global_batch": {"yolo": a.yolo_batch, "rektnet": a.rekt_batch, "postprocess": a.post_batch, "joint": a.joint_batch}[primary] * world,
                   "parallelism": f"dp{world}", "hipgraph": bool(a.graph), "optimizer": "FusedAdam",
                   **({"fidelity": "bf16 storage / fp32 accumulate; per-layer grad cosine vs fp32 >= reference-under-autocast - 0.02 (0.51 at conv 0); "
                                   "fp32-equivalent rate: fp32_images_per_sec"}
                      if (primary == "yolo" and a.precision == "bf16") else {})},
        "roofline": {k: roof[k] for k in ROOFLINE_KEYS if k in roof} if roof else None,
        "cpu_baseline": cpu_baseline,
        "host_cores": os.cpu_count(),
        "fingerprint": kernel_fingerprint(),
        "workloads": extra,
    }
    env = live_env_overrides()
    if env:
        line["env_overrides"] = env
    return line


def measured_peaks():
    """About one second of probes on this box ("confirm on the box"): the dense bf16 MFMA rate (mdcv_probe_mfma: register-operand
    v_mfma_f32_16x16x32_bf16, 8 waves per CU) and the HBM rate of a 1 GiB device-to-device copy (read + write bytes).  Printed beside the
    spec peaks the roofline fractions are priced against; never used to re-price them."""
    import ctypes
    from mdcv import _lib
    L = _lib.lib()
    try:
        st = torch.cuda.current_stream().cuda_stream
        ncu = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
        sink = torch.zeros(4, device="cuda")
        fl = ctypes.c_double()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 0.0
        for iters in (2000, 20000, 20000, 20000):                 # (first: warm-up / clock ramp)
            e0.record()
            L.check(L.probe_mfma(ncu * 2, iters, sink.data_ptr(), ctypes.byref(fl), st), "probe_mfma")
            e1.record()
            e1.synchronize()
            best = max(best, fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        src = torch.empty(1 << 28, dtype=torch.float32, device="cuda").fill_(1.0)
        dst = torch.empty_like(src)
        gbs = 0.0
        for _ in range(4):
            e0.record()
            dst.copy_(src)
            e1.record()
            e1.synchronize()
            gbs = max(gbs, 2.0 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        del src, dst
        return {"measured_peak_tflops": round(best, 1), "measured_hbm_gbs": round(gbs, 1), "spec_peak_tflops": PEAK_BF16_TFLOPS,
                "spec_hbm_gbs": PEAK_HBM_GBS, "probe": "mdcv_probe_mfma: %d workgroups x 4 waves, 20000 x 8 MFMAs each; 1 GiB torch copy" % (ncu * 2)}
    except Exception as e:                                          # noqa: BLE001  (a probe must never take the bench line down)
        return {"error": repr(e)}


def write_detail(line, extra, detail):
    """Per-kernel tables (every modelled kernel's roofline row, per-call and per-kernel times, launch counts) go to bench_detail.json:
    gpurun_out/ when it exists (it is merged back from the GPU box), else the repo root; scripts/profile_round.sh copies it to profiles/."""
    d = os.path.join(ROOT, "gpurun_out")
    path = os.path.join(d if os.path.isdir(d) else ROOT, "bench_detail.json")
    try:
        with open(path, "w") as f:
            json.dump({"line": line, "workloads": extra, **detail}, f, indent=1)
        return path
    except OSError:
        return None


def reference_loop_yolo(model, device, xh, th, steps, warmup):
    """img/s of the statements the reference's training loop executes per batch (CVC-YOLOv3/train.py:57-93), verbatim in what they ask of the
    device: stock `torch.optim.Adam` over `model.parameters()` (train.py:180-187), the batch arriving in pinned host memory (DataLoader
    pin_memory=True, train.py:131) and moved with `.to(device, non_blocking=True)` (:60-61), a blocking `.item()` on the label count (:63), one
    `.sum().to('cpu').item()` per loss part (:75) and the eight `.item()` reads of the progress line (:85-89).  Reported beside the headline, never as it."""
    optimizer = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-3, weight_decay=0.0)

    def step():
        imgs = xh.to(device, non_blocking=True)
        targets = th.to(device, non_blocking=True)
        targets.requires_grad_(False)
        n_targets = ((targets[:, :, 1:5] > 0).sum(dim=2) > 1).sum().item() + 1e-12
        optimizer.zero_grad()
        losses = model(imgs, targets)
        losses[0].sum().backward()
        optimizer.step()
        logged = [loss.sum().to("cpu").item() for loss in losses]
        line = "%10.6f" % (losses[0].item() / n_targets)
        total = losses[0].item()
        for loss in losses[1:]:
            line += "%5.2f" % (loss.item() / total * 100)
        return logged, line
    dt = timed_region(step, steps, warmup, device, 1)
    return xh.shape[0] * steps / dt


def reference_loop_rektnet(model, crit, device, xh, hmh, ph, steps, warmup):
    """the RektNet twin (RektNet/train_eval.py:59-79): pageable host batch (its DataLoader sets no pin_memory, :256) moved with `.to(device)`,
    stock Adam, three `.item()` reads per batch."""
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-1)

    def step():
        x_batch, y_hm_batch, y_points_batch = xh.to(device), hmh.to(device), ph.to(device)
        optimizer.zero_grad()
        output = model(x_batch)
        loc_loss, geo_loss, loss = crit(output[0], output[1], y_hm_batch, y_points_batch)
        loss.backward()
        optimizer.step()
        return loc_loss.item(), geo_loss.item(), loss.item()
    dt = timed_region(step, steps, warmup, device, 1)
    return xh.shape[0] * steps / dt


def comm_preflight(backend, rank, world, device):
    """The multi-GPU run's first act (scripts/dp_preflight.py has the long form): process-group init, a 4-byte all-reduce (the first collective
    builds the rings), who is there, and ONE timed all-reduce(SUM) of the YOLOv3 flat gradient's size (248 MB as four buckets).  A failure prints
    a one-line JSON DIAGNOSIS naming the step instead of leaving the driver with a return code; the numbers ride in `workloads.yolo.comm`."""
    step = "init_process_group"
    try:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        step = "first_allreduce"
        t = torch.ones(1, device=device)
        dist.all_reduce(t)
        torch.cuda.synchronize(device)
        if int(t.item()) != world:
            raise RuntimeError(f"4-byte all-reduce returned {t.item()}, expected {world}")
        step = "roll_call"
        seen = [None] * world
        dist.all_gather_object(seen, (rank, torch.cuda.current_device(), os.environ.get("HIP_VISIBLE_DEVICES")))
        step = "gradient_sized_allreduce"
        n = 62 * (1 << 20)
        buf = torch.full((n,), float(rank + 1), device=device)
        per = n // 4
        times = []
        for _ in range(3):
            torch.cuda.synchronize(device)
            dist.barrier()
            t0 = time.perf_counter()
            for b in range(4):
                dist.all_reduce(buf[b * per:(b + 1) * per], op=dist.ReduceOp.SUM)
            torch.cuda.synchronize(device)
            times.append(time.perf_counter() - t0)
            buf.fill_(float(rank + 1))
        ms = min(times[1:]) * 1e3
        del buf
        return {"ranks_seen": len({s0[0] for s0 in seen if s0 is not None}), "devices": sorted({s0[1] for s0 in seen if s0 is not None}),
                "preflight_allreduce_248mb_ms": ms, "preflight_first_ms": times[0] * 1e3,
                "preflight_busbw_gbs": 2 * (world - 1) / world * n * 4 / (ms * 1e-3) / 1e9}
    except Exception as e:                                   # noqa: BLE001
        import traceback
        print(json.dumps({"error": "data-parallel preflight failed", "step": step, "rank": rank, "world": world, "backend": backend,
                          "exception": repr(e), "trace": traceback.format_exc()[-1200:],
                          "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                          "visible_devices": torch.cuda.device_count(), "hint": "python scripts/dp_preflight.py --gpus N --model gives the per-rank report"}),
              flush=True)
        raise SystemExit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="both", choices=["both", "yolo", "rektnet", "postprocess", "joint"])
    ap.add_argument("--joint-batch", type=int, default=32, help="608x608 frames per GPU for the joint detect->keypoints workload")
    ap.add_argument("--post-batch", type=int, default=32, help="images per GPU for the detection post-processing workload")
    ap.add_argument("--yolo-batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--host-input", action="store_true", help="also time the step with the batch copied from pinned host memory (never `value`)")
    ap.add_argument("--yolo-classes", type=int, default=80, help="80 = BASELINE config; 1 = the cone-realistic variant of SURVEY 8d (18-channel heads)")
    ap.add_argument("--rekt-batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--graph", type=int, default=int(os.environ.get("MDCV_GRAPH", "0")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--dump-launches", default="")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-fp32", action="store_true", help="skip the same-precision-as-the-reference (fp32 kernels) YOLOv3 rate")
    ap.add_argument("--no-ref-loop", action="store_true", help="skip the legs that time the reference's unchanged training-loop statements")
    ap.add_argument("--no-classes1", action="store_true", help="skip the classes=1 (cone-realistic, 18-channel heads) YOLOv3 rate")
    a = ap.parse_args()
    global CPU_THREADS
    if a.cpu_threads:
        CPU_THREADS = a.cpu_threads

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the MDCV hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("MDCV_DIST_BACKEND", "nccl")      # "nccl" == RCCL on ROCm ; "gloo" lets several ranks share one GPU (tests)
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > ndev and backend == "nccl":
        raise SystemExit(f"bench.py --gpus {a.gpus}: this node shows {ndev} GPU(s) and RCCL takes one rank per device; refusing to run "
                         f"fewer ranks than asked for")
    if "WORLD_SIZE" not in os.environ:
        if a.gpus > 1:            # plain `python bench.py --gpus N`: start the N ranks ourselves, exactly as the driver's launcher would
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
            sk.close()
            os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
                                      "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} was launched with WORLD_SIZE={world}: the launcher and the flag disagree")
    dev_index = local % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    preflight = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        preflight = comm_preflight(backend, rank, world, device)      # comm init + one timed gradient-sized all-reduce, BEFORE any model exists
    os.environ["MDCV_GRAPH"] = str(a.graph)
    from mdcv.yolo.models import Darknet
    from mdcv.rektnet.keypoint_net import KeypointNet
    from mdcv.rektnet.cross_ratio_loss import CrossRatioLoss
    from mdcv.optim import FusedAdam
    from mdcv.parallel import GradAllReducer

    result = {}
    extra = {}
    detail = {}            # per-kernel tables: written to bench_detail.json, NOT into the one line the driver parses
    tmp = tempfile.mkdtemp(prefix="mdcv_bench_")
    cfg = write_yolo_cfg(tmp, classes=a.yolo_classes)

    if a.workload in ("both", "yolo"):
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            torch.manual_seed(0)                                # identical initial weights on every rank
            net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision=a.precision)
        finally:
            os.chdir(cwd)
        net = net.to(device).train()
        opt = FusedAdam(net, lr=1e-3, pipeline=OPT_PIPELINE)      # update + re-pack run under the next forward's first layers
        red = GradAllReducer.attach(net, bucket_mb=32.0)       # RCCL all-reduce of finished buckets overlaps the rest of backward
        red.timing = world > 1
        g = torch.Generator().manual_seed(1000 + rank)         # rank-seeded shard of the global batch
        B = a.yolo_batch
        x = torch.rand(B, 3, 416, 416, generator=g).to(device)
        tg = synth_targets(B, 16, g).to(device)

        def yolo_step():
            opt.zero_grad()
            out = net(x, tg)
            out[0].sum().backward()
            red.finish()
            opt.step()
            return out
        dt = timed_region(yolo_step, a.steps, a.warmup, device, world)
        ips = B * world * a.steps / dt
        comm = red.pop_comm_stats()                            # averaged over the warm-up and the timed steps
        red.timing = False
        loss = float(yolo_step()[0])
        result = {"ms_per_step": 1e3 * dt / a.steps, "value": ips}
        pcie = None
        if a.host_input or (world == 1 and not a.no_ref_loop):   # what train.py's loop does with a DataLoader batch: pinned host tensors, .to(device, non_blocking=True)
            # (part of the default N = 1 line since round 6: VERDICT r5 found the field null in the driver's line; --host-input forces it for N > 1)
            xh, th = x.cpu().pin_memory(), tg.cpu().pin_memory()

            def yolo_step_h2d():
                xd, td = xh.to(device, non_blocking=True), th.to(device, non_blocking=True)
                opt.zero_grad()
                out = net(xd, td)
                out[0].sum().backward()
                red.finish()
                opt.step()
                return out
            dth = timed_region(yolo_step_h2d, a.steps, a.warmup, device, world)
            pcie = B * world * a.steps / dth
        unchanged = None
        if world == 1 and not a.no_ref_loop:
            # the loop the drop-in promises to leave unchanged, on the same model (its parameters are views of the flat buffers either way)
            unchanged = reference_loop_yolo(net, device, x.cpu().pin_memory(), tg.cpu().pin_memory(), max(5, min(20, a.steps)), 3)
        extra["yolo"] = {"images_per_sec_with_h2d_copy": pcie, "unchanged_loop_images_per_sec": unchanged, "images_per_sec": ips, "ms_per_step": 1e3 * dt / a.steps, "global_batch": B * world, "final_loss": loss,
                         "mfma_frac_step": ips * (YOLO_TRAIN_GFLOP_PER_IMG if a.yolo_classes == 80 else 195.87) / 1e3 / (PEAK_BF16_TFLOPS * world)}
        if world > 1:                 # replicas must hold identical parameters after the reduced-gradient updates
            chk = net.flat_parameters()[0].double().sum().reshape(1)
            lo, hi = chk.clone(), chk.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            extra["yolo"]["replicas_in_sync"] = bool((hi - lo).abs().item() <= 1e-6 * max(1.0, abs(hi.item())))
            extra["yolo"]["comm"] = dict(comm or {}, **(preflight or {}), backend="rccl" if backend == "nccl" else backend, ranks=dist.get_world_size(),
                                         gradient_bytes=int(net.flat_parameters()[1].numel() * 4), bucket_mb=32.0)
            # allreduce_busy_ms = time the comm stream spent inside all-reduce calls per step; exposed_comm_ms = how long after the last
            # compute kernel of backward the last bucket finished (rank 0, HIP events)
        if not a.no_breakdown:        # every rank runs the instrumented steps (they contain the collective); rank 0 reports
            plan = [p for p in net._plans.values() if p.has_bwd][0]
            in_step = in_step_kernel_times(yolo_step)            # live, inside normal two-stream steps
            rec, krec = kernel_breakdown(net, plan, yolo_step)   # one serial step: work per launch + kernel-alone durations
            tot = sum(v[1] for v in rec.values())
            detail["yolo_call_ms_per_serial_step"] = {k: round(v[1], 3) for k, v in sorted(rec.items(), key=lambda kv: -kv[1][1])}
            detail["yolo_call_launches_per_step"] = {k: v[0] for k, v in rec.items()}
            detail["yolo_kernel_ms_in_step"] = {k: [round(v[0], 2), round(v[1], 4)] for k, v in sorted(in_step.items(), key=lambda kv: -kv[1][1])}
            extra["yolo"]["sum_kernel_ms_serial"] = round(tot, 3)
            extra["yolo"]["kernel_launches_per_step"] = int(round(sum(v[0] for v in in_step.values())))
            default_cfg = B == 32 and a.precision == "bf16" and a.yolo_classes == 80
            traffic, why = load_traffic() if default_cfg else (None, "non-default workload")
            top, rows = roofline_objects(krec, a.precision, traffic, in_step)
            if top:
                top["traffic_source"] = traffic["_file"] if traffic else why
                result["roofline"] = top
                detail["yolo_roofline_kernels"] = rows
                if traffic:
                    extra["yolo"]["hbm_bytes_per_step"] = traffic.get("total_bytes_per_step")
        if world > 1:
            # the number that decides the exchange design on hardware: the same step on the same ranks with the reducer DETACHED (no
            # all-reduce; replicas diverge from here on, so this runs last).  stretch = attached / detached.
            net._dp_reducer = None
            nd = max(3, min(10, a.steps))

            def yolo_step_local():
                opt.zero_grad()
                net(x, tg)[0].sum().backward()
                opt.step()
            dtd = timed_region(yolo_step_local, nd, 3, device, world)
            extra["yolo"]["comm"]["ms_per_step_reducer_detached"] = 1e3 * dtd / nd
            extra["yolo"]["comm"]["step_stretch_with_reducer"] = (dt / a.steps) / (dtd / nd)
        if world == 1 and a.precision == "bf16" and not a.no_fp32 and a.workload in ("both", "yolo"):
            # the reference computes in fp32: the same step with the fp32 kernels (exact-f32 MFMA), reported beside the bf16 value
            del opt
            net._plans.clear()
            torch.cuda.empty_cache()
            cwd = os.getcwd()
            os.chdir(tmp)
            try:
                torch.manual_seed(0)
                net32 = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision="fp32").to(device).train()
            finally:
                os.chdir(cwd)
            opt32 = FusedAdam(net32, lr=1e-3)

            def yolo_step32():
                opt32.zero_grad()
                out = net32(x, tg)
                out[0].sum().backward()
                opt32.step()
            n32 = max(2, min(5, a.steps))
            dt32 = timed_region(yolo_step32, n32, 2, device, 1)
            extra["yolo"]["fp32_images_per_sec"] = B * n32 / dt32
            detail["yolo_fp32_note"] = f"same step with precision='fp32' (fp32 storage, v_mfma_f32_16x16x4_f32), {n32} timed steps after 2 warm-up"
            del net32, opt32
            opt = None
        del net, opt
        torch.cuda.empty_cache()
        if world == 1 and a.yolo_classes == 80 and a.precision == "bf16" and not a.no_classes1:
            # "also report classes=1" -- the cone-realistic variant (18-channel heads, models.py:51-54), same step, same batch
            cfg1 = write_yolo_cfg(tempfile.mkdtemp(prefix="mdcv_bench_c1_"), classes=1)
            cwd = os.getcwd()
            os.chdir(os.path.dirname(cfg1))
            try:
                torch.manual_seed(0)
                net1 = Darknet(cfg1, 2.0, 1.6, 25.0, 0.1, True, precision=a.precision).to(device).train()
            finally:
                os.chdir(cwd)
            opt1 = FusedAdam(net1, lr=1e-3, pipeline=OPT_PIPELINE)

            def yolo_step_c1():
                opt1.zero_grad()
                net1(x, tg)[0].sum().backward()
                opt1.step()
            n1 = max(5, min(20, a.steps))
            dt1 = timed_region(yolo_step_c1, n1, 5, device, 1)
            extra["yolo"]["classes1_images_per_sec"] = B * n1 / dt1
            del net1, opt1
            torch.cuda.empty_cache()

    if a.workload in ("both", "rektnet"):
        torch.manual_seed(0)
        kp = KeypointNet(7, (80, 80), precision=a.precision).to(device).train()
        with contextlib.redirect_stdout(sys.stderr):           # the reference's constructor prints its configuration
            crit = CrossRatioLoss("l1_softargmax", True, 0.05, 0.05)
        opt = FusedAdam(kp, lr=0.1, pipeline=OPT_PIPELINE)
        red = GradAllReducer.attach(kp, bucket_mb=32.0)
        g = torch.Generator().manual_seed(2000 + rank)
        B = a.rekt_batch
        x = torch.rand(B, 3, 80, 80, generator=g).to(device)
        tp = (torch.rand(B, 7, 2, generator=g) * (79 / 80)).to(device)

        def rekt_step():
            opt.zero_grad()
            hm, pts = kp(x)
            loss = crit(hm, pts, None, tp)[2]
            loss.backward()
            red.finish()
            opt.step()
            return loss
        dt = timed_region(rekt_step, a.steps, a.warmup, device, world)
        ips = B * world * a.steps / dt
        extra["rektnet"] = {"images_per_sec": ips, "ms_per_step": 1e3 * dt / a.steps, "global_batch": B * world,
                            "final_loss": float(rekt_step()),
                            "mfma_frac_step": ips * REKT_TRAIN_GFLOP_PER_IMG / 1e3 / (PEAK_BF16_TFLOPS * world),
                            "hbm_frac_step": ips * REKT_TRAIN_MB_PER_IMG / 1e3 / (PEAK_HBM_GBS * world)}
        if world == 1 and not a.no_ref_loop:
            with contextlib.redirect_stdout(sys.stderr):
                extra["rektnet"]["unchanged_loop_images_per_sec"] = reference_loop_rektnet(
                    kp, crit, device, x.cpu(), torch.zeros(B, 7, 80, 80), tp.cpu(), max(5, min(20, a.steps)), 3)
        if not a.no_breakdown:
            plan = [p for p in kp._plans.values() if p.has_bwd][0]
            in_step = in_step_kernel_times(rekt_step)
            rec, krec = kernel_breakdown(kp, plan, rekt_step)
            detail["rektnet_call_ms_per_serial_step"] = {k: round(v[1], 3) for k, v in sorted(rec.items(), key=lambda kv: -kv[1][1])}
            detail["rektnet_kernel_ms_in_step"] = {k: [round(v[0], 2), round(v[1], 4)] for k, v in sorted(in_step.items(), key=lambda kv: -kv[1][1])}
            rtraffic, rwhy = load_traffic("rektnet") if (B == 256 and a.precision == "bf16") else (None, "non-default workload")
            top, rows = roofline_objects(krec, a.precision, rtraffic, in_step)
            detail["rektnet_roofline_kernels"] = rows
            if rtraffic:
                extra["rektnet"]["hbm_bytes_per_step"] = rtraffic.get("total_bytes_per_step")
            if top:
                top["traffic_source"] = rtraffic["_file"] if rtraffic else rwhy
                extra["rektnet"]["dominant_kernel"] = {k: top[k] for k in ("kernel", "bound", "avg_us", "frac", "frac_alone")}
                if a.workload == "rektnet":
                    result_roof = top
        if a.workload == "rektnet":
            result = {"ms_per_step": 1e3 * dt / a.steps, "value": ips}
            if not a.no_breakdown and top:
                result["roofline"] = result_roof

    if a.workload in ("both", "postprocess"):
        # validate.py's per-image loop for one batch of eval outputs [B, 10647, 85] (416^2, 80 classes),
        # thresholds of yolo_baseline.cfg:17-20.  Images are independent: ranks shard them, no collective.
        from mdcv.yolo.postprocess import detect_postprocess
        B, N, C, T = a.post_batch, 10647, 80, 16
        out_np, tg_np = synth_eval_output(B, N, C, T, 3000 + rank)
        outp, tgp = torch.from_numpy(out_np).to(device), torch.from_numpy(tg_np).to(device)

        def post_step():
            return detect_postprocess(outp, tgp, 0.8, 0.25, 0.5, 416, 416)
        dt = timed_region(post_step, max(a.steps, 50), a.warmup, device, world)
        nst = max(a.steps, 50)
        det = post_step()
        extra["postprocess"] = {"images_per_sec": B * world * nst / dt, "ms_per_batch": 1e3 * dt / nst, "batch_per_gpu": B,
                                "rows_per_image": N, "classes": C, "kept_mean": float(det.count.float().mean()),
                                "mean_ap": float(det.stats[det.stats[:, 3] > 0, 0].mean()),
                                # HBM floor: the confidence column is one 4-byte word out of every 340-byte row, so the
                                # filter touches every 64-byte sector that holds one -> ~B*N*64 bytes
                                "hbm_floor_us": B * N * 64 / (PEAK_HBM_GBS * 1e9) * 1e6}
        if a.workload == "postprocess":
            result = {"ms_per_step": 1e3 * dt / nst, "value": B * world * nst / dt}
        if rank == 0 and world == 1 and not a.no_cpu_baseline:
            extra["postprocess"]["cpu_baseline"] = cpu_baseline_post(out_np, tg_np)

    if a.workload in ("joint", "both"):
        # BASELINE.json configs[4]: YOLOv3 608x608 detect -> batched RektNet crops; frames are independent, ranks shard them.
        # A random-init detector in eval mode outputs a near-constant confidence (~0.5), so nothing passes conf 0.8.  The
        # detector forward is run and timed for real; its output then gets synthetic cone detections written into a few rows
        # per frame (1-5 jittered rows around each of 16 cones, one tiny scatter inside the timed region), so that NMS, the
        # crops and KeypointNet all run on a realistic load with the cfg's own thresholds (conf 0.8, NMS 0.25).
        from mdcv.pipeline import JointPipeline, crop_resize
        from mdcv.yolo.postprocess import detect_postprocess
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            torch.manual_seed(0)
            net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision=a.precision)
        finally:
            os.chdir(cwd)
        net = net.to(device).eval()
        kp = KeypointNet(7, (80, 80), precision=a.precision).to(device).eval()
        B = a.joint_batch
        g = torch.Generator().manual_seed(4000 + rank)
        x8 = torch.randint(0, 256, (B, 3, 608, 608), generator=g, dtype=torch.uint8).to(device)    # the decoded camera frames (cv2.imread gives uint8)
        x = x8.float() / 255.0                                                                    # what the detector is fed
        rng = np.random.default_rng(4000 + rank)
        rows, vals = [], []
        for b in range(B):
            perm = rng.permutation(22743)
            r = 0
            for _ in range(16):
                cx, cy = rng.random(2) * 540 + 34
                w, h = rng.random() * 40 + 14, rng.random() * 60 + 20
                for _ in range(int(rng.integers(1, 6))):
                    j = 1 + 0.06 * rng.standard_normal(4)
                    rows.append(b * 22743 + perm[r]); r += 1
                    vals.append([cx * j[0], cy * j[1], w * j[2], h * j[3], 0.8 + 0.2 * rng.random()])
        rows_t = torch.tensor(np.asarray(rows), dtype=torch.long, device=device)
        vals_t = torch.tensor(np.asarray(vals, np.float32), device=device)

        class Seeded(torch.nn.Module):
            def __init__(self, inner):
                super().__init__()
                self.inner = inner

            def get_threshs(self):
                return self.inner.get_threshs()

            def img_size(self):
                return 608, 608

            def forward(self, imgs):
                o = self.inner(imgs)
                o.view(-1, o.shape[2])[rows_t, :5] = vals_t
                return o
        det_net = Seeded(net).eval()
        thr = 0.8
        with torch.no_grad():
            out0 = det_net(x)
        pipe = JointPipeline(det_net, kp, conf_thres=thr, nms_thres=0.25, max_cones=16, bucket=64)
        stages = {}

        def stage(name, fn, n=None):
            n = n or a.steps
            stages[name] = 1e3 * timed_region(fn, n, a.warmup, device, world) / n
        with torch.no_grad():
            stage("detector_eval_608", lambda: det_net(x))
            det = detect_postprocess(out0, None, thr, 0.25, 0.5, 608, 608)
            stage("postprocess", lambda: detect_postprocess(out0, None, thr, 0.25, 0.5, 608, 608))
            # crops are cut from the uint8 frames with the reference's rule: 8-bit fixed-point cv2.resize, then / 255 (RektNet/dataset.py:35-38,52)
            crops, _, M = crop_resize(x8, det.boxes[:, :16], det.count, (80, 80), pad_rows_to=64)
            stage("crop_resize", lambda: crop_resize(x8, det.boxes[:, :16], det.count, (80, 80), pad_rows_to=64))
            stage("keypoint_eval", lambda: kp(crops))
            dt = timed_region(lambda: pipe(x, frames=x8), a.steps, a.warmup, device, world)
        ips = B * world * a.steps / dt
        extra["joint"] = {"images_per_sec": ips, "ms_per_batch": 1e3 * dt / a.steps, "frames_per_gpu": B, "crops_per_batch": M,
                          "crop_rule": "uint8 frame -> cv2 8-bit fixed-point INTER_LINEAR -> /255 (mdcv_crop_resize_u8)",
                          "conf_thres": thr, "kept_per_frame_mean": float(det.count.float().mean()), "stage_ms": {k: round(v, 4) for k, v in stages.items()},
                          "stage_images_per_sec": {k: round(B * world / (v * 1e-3), 1) for k, v in stages.items()}}
        jres = {"ms_per_step": 1e3 * dt / a.steps, "value": ips}
        if not a.no_breakdown:
            def joint_step():
                with torch.no_grad():
                    pipe(x, frames=x8)
            plan = [p for p in net._plans.values() if not p.has_bwd][0]
            in_step = in_step_kernel_times(joint_step)

            def det_step():
                with torch.no_grad():
                    net(x)
            rec, krec = kernel_breakdown(net, plan, det_step)
            top, rows = roofline_objects(krec, a.precision, None, in_step)
            detail["joint_roofline_kernels"] = rows
            detail["joint_kernel_ms_in_step"] = {k: [round(v[0], 2), round(v[1], 4)] for k, v in sorted(in_step.items(), key=lambda kv: -kv[1][1])}
            if top:
                top["traffic_source"] = "not collected for this workload"
                jres["roofline"] = top
        if a.workload == "joint":
            result = jres
        else:                                  # default run: BASELINE config 5 rides along as scalars (tables go to the detail file)
            detail["joint"] = extra["joint"]
            jr = jres.get("roofline") or {}
            extra["joint"] = {"images_per_sec": ips, "ms_per_batch": 1e3 * dt / a.steps, "frames_per_gpu": B, "crops_per_batch": M,
                              "roofline_kernel": jr.get("kernel"), "roofline_frac": jr.get("frac"), "roofline_frac_alone": jr.get("frac_alone")}
            if rank == 0 and world == 1 and not a.no_cpu_baseline:
                jb = cpu_baseline_joint(cfg, tmp)
                detail["joint_cpu_baseline"] = jb
                extra["joint"]["cpu_images_per_sec"] = jb["value"] if jb else None
            del net, kp, pipe, det_net
            torch.cuda.empty_cache()

    if rank == 0:
        primary = a.workload if a.workload in ("rektnet", "postprocess", "joint") else "yolo"
        cb = None
        if world == 1 and not a.no_cpu_baseline:
            cb = (cpu_baseline_yolo(cfg, tmp) if primary == "yolo" else cpu_baseline_rektnet() if primary == "rektnet"
                  else extra["postprocess"].pop("cpu_baseline", None) if primary == "postprocess" else cpu_baseline_joint(cfg, tmp))
            if a.workload == "both":
                rb = cpu_baseline_rektnet()
                detail["rektnet_cpu_baseline"] = rb
                extra["rektnet"]["cpu_images_per_sec"] = rb["value"]
                pb = extra["postprocess"].pop("cpu_baseline", None)
                if pb:
                    detail["postprocess_cpu_baseline"] = pb
                    extra["postprocess"]["cpu_images_per_sec"] = pb["value"]
        line = build_line(a, world, primary, result, extra, cb)
        detail["peaks"] = measured_peaks()
        line["peaks"] = {k: detail["peaks"].get(k) for k in ("measured_peak_tflops", "measured_hbm_gbs")}
        detail_path = write_detail(line, extra, detail)
        line["detail"] = os.path.relpath(detail_path, ROOT) if detail_path else None
        print(json.dumps(compact(line)))
    if rank == 0 and a.dump_launches:
        with open(a.dump_launches, "w") as f:
            json.dump(LAUNCH_DUMP, f)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
