"""-m gpu: data-parallel parity of the HIP path (BASELINE.json configs[3]).

N ranks (2 and 4) share this one GPU over the gloo backend (RCCL refuses two ranks on one device; the RCCL exchange itself is
covered by test_gpu_models.py::test_rccl_allreduce_path_single_rank and csrc's mdcv_comm_* test below).  Rank r runs the HIP
drop-in (fp32 kernels) on shard r of the fixture batch with the overlapped bucketed all-reduce attached, exactly as bench.py /
a training loop would, and checks the reference's nn.DataParallel semantics (CVC-YOLOv3/train.py:193-195 with
`losses[0].sum().backward()`, train.py:70) against vectors produced by the reference itself (tests/golden/make_golden.py):
  * rank r's losses == the reference on shard r alone        (per-shard BatchNorm statistics, per-shard build_targets)
  * the reduced gradient == the SUM over shards of the reference's gradients (conv-0 / last-layer tensors and every tensor's norm)
  * every rank holds the same reduced gradient bit for bit.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _hang_file(port, rank):
    return os.path.join("/tmp", f"mdcv_dp_hang_{port}_{rank}.txt")


def _worker(rank, world, port, which, q):
    try:
        import faulthandler
        hang = open(_hang_file(port, rank), "w")              # a worker still running after 300 s leaves its Python stacks here: the parent's failure
        faulthandler.dump_traceback_later(300, exit=False, file=hang)     # message then NAMES the blocking call instead of reporting a timeout
        sys.path.insert(0, ROOT)
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # the rendezvous is 127.0.0.1: no hostname lookups (they fail with err=-3 on these boxes)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch.distributed as dist
        import datetime
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
        if which.startswith("full"):
            q.put((rank, _yolo_full(rank, world, which.endswith("1"))))
        elif which == "auto_uneven":
            q.put((rank, _auto_uneven(rank, world)))
        else:
            q.put((rank, (_yolo if which == "yolo" else _rektnet)(rank, world)))
        dist.barrier()
        dist.destroy_process_group()
        faulthandler.cancel_dump_traceback_later()
    except Exception as e:                                  # surface failures instead of letting the parent wait for its timeout
        import traceback
        q.put((rank, {"error": repr(e) + traceback.format_exc()}))


def _yolo(rank, world):
    from mdcv.yolo.models import Darknet
    from mdcv.parallel import GradAllReducer, shard_batch
    z = np.load(os.path.join(G, "mini_darknet_dp.npz"))
    os.chdir(os.path.join(G, "mini"))
    net = Darknet("mini.cfg", 2.0, 1.6, 25.0, 0.1, False, precision="fp32")
    net.load_weights("mini.weights", net.get_start_weight_dim())
    net = net.cuda().train()
    red = GradAllReducer.attach(net, bucket_mb=0.05)        # several buckets on this tiny net, reduced while backward still runs
    x = shard_batch(torch.from_numpy(z["x"]), rank, world).cuda()
    tg = shard_batch(torch.from_numpy(z["targets"]), rank, world).cuda()
    out = net(x, tg)
    out[0].sum().backward()
    red.finish()
    torch.cuda.synchronize()
    params = list(net.parameters())
    return {"losses": torch.stack([o.detach() for o in out]).cpu().numpy(), "g0": params[0].grad.cpu().numpy(),
            "glast": params[-2].grad.cpu().numpy(), "gnorm": [float(p.grad.double().norm()) for p in params],
            "buckets": len(red.log), "flat": net.flat_parameters()[1].cpu().numpy()}


def _rektnet(rank, world):
    from mdcv.rektnet.keypoint_net import KeypointNet
    from mdcv.rektnet.cross_ratio_loss import CrossRatioLoss
    from mdcv.parallel import GradAllReducer, shard_batch
    z = np.load(os.path.join(G, "rektnet_dp.npz"))
    zs = np.load(os.path.join(G, "rektnet_net.npz"))
    net = KeypointNet(7, (80, 80), precision="fp32")
    net.load_state_dict({k[4:]: torch.from_numpy(zs[k]) for k in zs.files if k.startswith("sd::")})
    net = net.cuda().train()
    red = GradAllReducer.attach(net, bucket_mb=0.25)
    x = shard_batch(torch.from_numpy(z["x"]), rank, world).cuda()
    tp = shard_batch(torch.from_numpy(z["tpts"]), rank, world).cuda()
    hm, pts = net(x)
    loc, geo, tot = CrossRatioLoss("l1_softargmax", True, 0.05, 0.05)(hm, pts, None, tp)
    tot.backward()
    red.finish()
    torch.cuda.synchronize()
    params = list(net.parameters())
    return {"losses": np.array([float(loc), float(geo), float(tot)], np.float32), "g0": params[0].grad.cpu().numpy(),
            "gnorm": [float(p.grad.double().norm()) for p in params], "names": [n for n, _ in net.named_parameters()],
            "buckets": len(red.log), "flat": net.flat_parameters()[1].cpu().numpy()}


def _auto_uneven(rank, world):
    """torchrun-transparent mode with batches the rank count does not divide: 3 images over 2 ranks (chunks of 2 and 1, like
    nn.DataParallel's scatter) and 1 image over 2 ranks (rank 1's chunk is empty: it joins the exchange with exact-zero gradients); a replica
    whose weights differ is brought in line by the broadcast from rank 0 when the exchange is attached."""
    import copy
    import warnings
    from mdcv import parallel
    from mdcv.yolo.models import Darknet
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    z = np.load(os.path.join(G, "mini_darknet_dp.npz"))
    os.chdir(os.path.join(G, "mini"))
    net = Darknet("mini.cfg", 2.0, 1.6, 25.0, 0.1, False, precision="fp32")
    net.load_weights("mini.weights", net.get_start_weight_dim())
    ref = copy.deepcopy(net).cuda().train()                  # single-process reference on rank 0's weights, built before the mode is switched on
    if rank == 1:
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(1.5)                                  # an unseeded replica: the attach must overwrite it with rank 0's parameters
    net = net.cuda().train()
    x, tg = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["targets"]).cuda()

    def grads(model, xs, ts):
        model.zero_grad()
        out = model(xs, ts)
        out[0].sum().backward()
        torch.cuda.synchronize()
        return torch.stack([o.detach() for o in out]).cpu().numpy(), model.flat_parameters()[1].double().cpu().numpy().copy()
    want = {}
    l20, g20 = grads(ref, x[0:2], tg[0:2])
    l21, g21 = grads(ref, x[2:3], tg[2:3])
    l10, g10 = grads(ref, x[0:1], tg[0:1])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert parallel.enable_auto_data_parallel() is not None
        l3, g3 = grads(net, x[0:3], tg[0:3])                     # the script hands every rank the WHOLE batch
        synced = float(net.flat_parameters()[0].double().sum())
        l1, g1 = grads(net, x[0:1], tg[0:1])
    return {"l3": l3, "g3": g3, "l1": l1, "g1": g1, "want_l3": [l20, l21], "want_g3": g20 + g21, "want_l1": l10, "want_g1": g10,
            "synced": synced, "ref_sum": float(ref.flat_parameters()[0].double().sum()), "attached": net._dp_reducer is not None}


def _yolo_full(rank, world, defer):
    """the full yolo_baseline (bf16, 416^2, two images per rank): the one-launch 1x1 backward's slab reduces are DEFERRED on the side stream
    (mdcv/yolo/models.py run_bwd_list) and, under data parallel, flushed in front of every marker that starts a bucket"""
    import tempfile
    sys.path.insert(0, ROOT)
    import bench
    from mdcv.yolo import models as ym
    from mdcv.parallel import GradAllReducer
    ym._NetPlan.defer_slab_reduce = bool(defer)
    tmp = tempfile.mkdtemp(prefix=f"mdcv_dp{rank}_")
    cfg = bench.write_yolo_cfg(tmp)
    os.chdir(tmp)
    torch.manual_seed(0)
    net = ym.Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision="bf16").cuda().train()
    red = GradAllReducer.attach(net, bucket_mb=32.0) if world > 1 else None
    g = torch.Generator().manual_seed(77)
    x = torch.rand(2 * 2, 3, 416, 416, generator=g)
    tg = bench.synth_targets(2 * 2, 16, g)
    if world > 1:
        x, tg = x[2 * rank:2 * rank + 2], tg[2 * rank:2 * rank + 2]
    outs = []
    shards = [(x, tg)] if world > 1 else [(x[:2], tg[:2]), (x[2:], tg[2:])]
    for xs, ts in shards:                                     # (world == 1: the two shards one after the other, for the sum the exchange must give)
        net.zero_grad()
        out = net(xs.cuda(), ts.cuda())
        out[0].sum().backward()
        if red is not None:
            red.finish()
        torch.cuda.synchronize()
        outs.append(net.flat_parameters()[1].double().cpu().numpy().copy())
    plan = [p for p in net._plans.values() if p.has_bwd][0]
    roles = plan._classify_bwd()
    return {"flat": outs, "deferred": int(sum(1 for r in roles if r == 3)), "buckets": len(red.log) if red is not None else 0}


def _run(world, which, attempts=1):
    """Spawn `world` ranks and collect their results.  Round 5 retried a spawn whose workers had not reported within 240 s ("seen once on a fresh
    box"); round 6 took the retry out: scripts/dp_spawn_soak.py ran this very spawn 240 times on two leases without one hang (1.7 - 2.9 s per run,
    profiles/r06_dp_spawn_soak.txt).  The one blocking dependency outside the process the hunt found is name resolution -- c10d logs "hostname of
    the client socket cannot be retrieved, err=-3" for every connection on these boxes -- so the workers pin gloo to the loopback interface; and a
    worker that is still running after 300 s dumps its Python stacks (faulthandler), which the failure message below quotes: a hang now names its
    blocking call instead of costing a run."""
    import queue as _queue
    ctx = mp.get_context("spawn")
    last = None
    for attempt in range(attempts):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, which, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = {}
        try:
            for _ in range(world):
                r, v = q.get(timeout=360)
                res[r] = v
        except _queue.Empty:
            last = f"attempt {attempt}: only ranks {sorted(res)} of {world} reported within 360 s"
            for r in range(world):                            # the stacks the stuck workers dumped at 300 s (faulthandler, _worker)
                try:
                    last += f"\n--- rank {r} ---\n" + open(_hang_file(port, r)).read()[-3000:]
                except OSError:
                    pass
        for p in procs:
            p.join(timeout=30 if len(res) == world else 1)
            if p.is_alive():
                p.kill()
                p.join(timeout=10)
        if len(res) == world:
            for r in range(world):
                assert "error" not in res[r], res[r]["error"]
            return res
    raise AssertionError(f"data-parallel workers hung ({last})")


@pytest.mark.parametrize("world", [2, 4])
def test_hip_mini_darknet_data_parallel_vs_reference(world):
    z = np.load(os.path.join(G, "mini_darknet_dp.npz"))
    res = _run(world, "yolo")
    for r in range(world):
        np.testing.assert_allclose(res[r]["losses"], z[f"losses_{world}"][r], rtol=1e-4, err_msg=f"rank {r}: per-shard losses")
        scale = float(np.abs(z[f"g0_{world}"]).max())
        assert float(np.abs(res[r]["g0"] - z[f"g0_{world}"]).max()) <= 1e-3 * scale, f"rank {r}: reduced conv-0 gradient"
        scale = float(np.abs(z[f"glast_{world}"]).max())
        assert float(np.abs(res[r]["glast"] - z[f"glast_{world}"]).max()) <= 1e-3 * scale, f"rank {r}: reduced head gradient"
        np.testing.assert_allclose(res[r]["gnorm"], z[f"gnorm_{world}"], rtol=2e-3, atol=1e-6)
        assert res[r]["buckets"] > 3                         # the exchange really ran bucket by bucket
        assert np.array_equal(res[r]["flat"], res[0]["flat"]), f"rank {r} holds a different reduced gradient than rank 0"


def test_full_yolo_data_parallel_with_deferred_slab_reduces():
    """Two ranks of the FULL yolo_baseline (bf16; the mini cfg above has no layer that takes the one-launch 1x1 backward): with the slab reduces of the
    1x1 layers deferred behind the next weight gradient's fork -- and flushed in front of every marker that starts a bucket's all-reduce -- the
    reduced gradient is bit-identical to the schedule with one fork per slab reduce, every rank holds the same buffer, and it is the SUM of the two
    shards' gradients computed one after the other in a single process (fp64 sum of the fp32 buffers; the exchange adds in fp32)."""
    a = _run(2, "full1")
    b = _run(2, "full0")
    assert a[0]["deferred"] >= 30 and b[0]["deferred"] == 0 and a[0]["buckets"] >= 4, (a[0]["deferred"], b[0]["deferred"], a[0]["buckets"])
    assert np.array_equal(a[0]["flat"][0], a[1]["flat"][0]) and np.array_equal(a[0]["flat"][0], b[0]["flat"][0])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single_full, args=(q,))
    p.start()
    single = q.get(timeout=600)
    p.join(timeout=60)
    assert "error" not in single, single.get("error")
    want = single["flat"][0] + single["flat"][1]
    got = a[0]["flat"][0]
    scale = float(np.abs(want).max())
    assert float(np.abs(got - want).max()) <= 1e-6 * scale, float(np.abs(got - want).max()) / scale


def _single_full(q):
    try:
        torch.cuda.set_device(0)
        q.put(_yolo_full(0, 1, True))
    except Exception as e:
        import traceback
        q.put({"error": repr(e) + traceback.format_exc()})


@pytest.mark.parametrize("world", [2, 4])
def test_hip_keypointnet_data_parallel_vs_reference(world):
    z = np.load(os.path.join(G, "rektnet_dp.npz"))
    res = _run(world, "rektnet")
    for r in range(world):
        np.testing.assert_allclose(res[r]["losses"], z[f"losses_{world}"][r], rtol=1e-4, atol=1e-6, err_msg=f"rank {r}: per-shard losses")
        scale = float(np.abs(z[f"g0_{world}"]).max())
        assert float(np.abs(res[r]["g0"] - z[f"g0_{world}"]).max()) <= 5e-3 * scale, f"rank {r}: reduced stem gradient"
        for n, mine, ref in zip(res[r]["names"], res[r]["gnorm"], z[f"gnorm_{world}"]):
            if n.endswith(".bias") and "bn" not in n:        # mathematically zero (bias in front of a BatchNorm / under a softmax)
                assert mine < 2e-3, (n, mine)
            else:
                assert abs(mine - ref) <= 5e-3 * max(ref, 1e-4), (n, mine, ref)
        assert np.array_equal(res[r]["flat"], res[0]["flat"])


def test_auto_mode_uneven_and_short_batches_and_replica_broadcast():
    res = _run(2, "auto_uneven")
    for r in range(2):
        v = res[r]
        assert v["attached"]
        assert abs(v["synced"] - v["ref_sum"]) <= 1e-9 * abs(v["ref_sum"]), "rank 1 kept its own parameters: no broadcast from rank 0"
        np.testing.assert_allclose(v["l3"], v["want_l3"][r], rtol=1e-5, err_msg=f"rank {r}: loss of its DataParallel chunk of a 3-image batch")
        scale = float(np.abs(v["want_g3"]).max())
        assert float(np.abs(v["g3"] - v["want_g3"]).max()) <= 1e-5 * scale, "3 images over 2 ranks: reduced gradient != sum over the chunks"
        scale = float(np.abs(v["want_g1"]).max())
        assert float(np.abs(v["g1"] - v["want_g1"]).max()) <= 1e-5 * scale, "1 image over 2 ranks: reduced gradient != the one real chunk's"
    np.testing.assert_allclose(res[0]["l1"], res[0]["want_l1"], rtol=1e-5)
    assert not np.any(res[1]["l1"]), "the rank without samples reports zero losses"
    assert np.array_equal(res[0]["g3"], res[1]["g3"]) and np.array_equal(res[0]["g1"], res[1]["g1"])


def _torchrun_stub(world, which, tmp_path):
    """`python -m torch.distributed.run --nproc-per-node <world> tests/helpers/train_loop_stub.py ...` with the drop-in directory in front on
    PYTHONPATH: the script is the reference's training statements, unchanged; nothing in it mentions ranks.  gloo + one shared GPU here (RCCL
    refuses two ranks on one device): MDCV_DP_BACKEND / MDCV_DP_DEVICE are the two test-only overrides."""
    import subprocess
    drop = os.path.join(ROOT, "dropin", "CVC-YOLOv3" if which == "yolo" else "RektNet")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")}
    env.update(PYTHONPATH=drop + os.pathsep + env.get("PYTHONPATH", ""), MDCV_DP_BACKEND="gloo", MDCV_DP_DEVICE="0", MDCV_PRECISION="fp32")
    fixture = os.path.join(G, "mini_darknet_dp.npz" if which == "yolo" else "rektnet_dp.npz")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "helpers", "train_loop_stub.py"), fixture,
                          os.path.join(G, "mini"), str(tmp_path), which], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    return [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]


@pytest.mark.parametrize("world", [2, 4])
def test_torchrun_on_the_unchanged_training_loop_yolo(world, tmp_path):
    """VERDICT r4 missing 1: `torchrun --nproc-per-node N train.py` must work on the UNCHANGED script (reference CVC-YOLOv3/train.py:57-93,
    :180-196).  The drop-in import pins each rank to one device (device_count() == 1: train.py:193 does not wrap in nn.DataParallel), the model
    takes rank r's shard of the batch and all-reduces its gradients, and backward() joins the exchange so the script's stock Adam step is
    safe.  Against the reference's own nn.DataParallel semantics (tests/golden/mini_darknet_dp.npz): rank r's losses == the reference on
    shard r alone; every rank's gradients == the SUM over shards."""
    z = np.load(os.path.join(G, "mini_darknet_dp.npz"))
    res = _torchrun_stub(world, "yolo", tmp_path)
    for r in range(world):
        assert int(res[r]["ndev"]) == 1 and int(res[r]["wrapped"]) == 0, "the rank still sees several devices: train.py would wrap the model"
        np.testing.assert_allclose(res[r]["losses"], z[f"losses_{world}"][r], rtol=1e-4, err_msg=f"rank {r}: per-shard losses")
        assert abs(float(res[r]["first"]) - float(z[f"losses_{world}"][r][0])) <= 1e-4 * abs(float(z[f"losses_{world}"][r][0]))
        scale = float(np.abs(z[f"g0_{world}"]).max())
        assert float(np.abs(res[r]["g0"] - z[f"g0_{world}"]).max()) <= 1e-3 * scale, f"rank {r}: reduced conv-0 gradient"
        scale = float(np.abs(z[f"glast_{world}"]).max())
        assert float(np.abs(res[r]["glast"] - z[f"glast_{world}"]).max()) <= 1e-3 * scale, f"rank {r}: reduced head gradient"
        np.testing.assert_allclose(res[r]["gnorm"], z[f"gnorm_{world}"], rtol=2e-3, atol=1e-6)
        assert np.array_equal(res[r]["g0"], res[0]["g0"])


def test_torchrun_on_the_unchanged_training_loop_rektnet(tmp_path):
    """The same for RektNet/train_eval.py:59-79: KeypointNet.forward keeps rank r's shard, CrossRatioLoss takes the matching shard of the labels
    the script hands over whole."""
    z = np.load(os.path.join(G, "rektnet_dp.npz"))
    res = _torchrun_stub(2, "rektnet", tmp_path)
    for r in range(2):
        np.testing.assert_allclose(res[r]["losses"], z["losses_2"][r], rtol=1e-4, atol=1e-6, err_msg=f"rank {r}: per-shard losses")
        # round 6: KeypointNet's exchange AVERAGES (its loss is a batch mean; the fixture holds the SUM over the two shards)
        scale = float(np.abs(z["g0_2"]).max()) / 2
        assert float(np.abs(res[r]["g0"] - z["g0_2"] / 2).max()) <= 5e-3 * scale, f"rank {r}: reduced stem gradient"
        assert np.array_equal(res[r]["g0"], res[0]["g0"])


@pytest.mark.parametrize("world,backend", [(1, "nccl"), (2, "gloo")])
def test_dp_preflight_script(world, backend):
    """scripts/dp_preflight.py (environment -> comm init -> timed bucketed all-reduce of the gradient's size -> correctness -> one data-parallel
    model step with replica checksums): RCCL with the one rank this box has, gloo with two ranks sharing the GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dp_preflight.py"), "--gpus", str(world), "--backend", backend, "--mb", "16",
                          "--model"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    rep = json.loads(out.stdout[out.stdout.index("{"):])
    assert rep["ok"] is True and rep["world"] == world and len(rep["ranks"]) == world
    for r in rep["ranks"]:
        assert [s["step"] for s in r["steps"]] == ["environment", "comm_init", "allreduce", "correctness", "model_replicas"]
        assert all(s["ok"] for s in r["steps"])
        assert r["steps"][0]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and r["steps"][2]["ms_median"] > 0


def test_bench_gpus_flag_is_honoured():
    """`python bench.py --gpus 2` on a box with fewer GPUs must fail loudly (never run fewer ranks than asked for); under a launcher
    whose WORLD_SIZE disagrees with the flag it must fail too; with the gloo test backend it spawns the two ranks itself."""
    import json
    import subprocess
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MDCV_DIST_BACKEND")}
    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, cwd=ROOT, capture_output=True,
                             text=True, timeout=300)
        assert out.returncode != 0 and "refusing" in (out.stderr + out.stdout)
    out = subprocess.run([sys.executable, bench, "--gpus", "1", "--steps", "1", "--warmup", "0"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                         cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "disagree" in (out.stderr + out.stdout)
    out = subprocess.run([sys.executable, bench, "--gpus", "2", "--workload", "yolo", "--yolo-batch", "4", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-breakdown"], env=dict(env, MDCV_DIST_BACKEND="gloo"), cwd=ROOT, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8
    y = line["workloads"]["yolo"]
    assert y["replicas_in_sync"] is True and y["comm"]["ranks"] == 2 and y["comm"]["buckets_per_step"] >= 4
    assert y["comm"]["allreduce_busy_ms"] > 0 and y["comm"]["exposed_comm_ms"] >= 0
    assert y["comm"]["ms_per_step_reducer_detached"] > 0 and y["comm"]["step_stretch_with_reducer"] > 0      # same ranks, exchange off
    assert len(out.stdout.splitlines()[-1]) < 3500


def test_comm_c_abi_single_rank_rccl():
    """mdcv_comm_* (the RCCL entry points of the C ABI, for hosts that do not go through torch.distributed) with a world of one rank on
    this GPU: all-reduce(SUM) over one rank is the identity; run in a subprocess so a broken RCCL cannot take the test session down."""
    import subprocess
    code = r"""
import ctypes, sys, torch
sys.path.insert(0, %r)
from mdcv import _lib
L = _lib.lib()
torch.cuda.set_device(0)
uid = ctypes.create_string_buffer(128)
L.check(L.comm_unique_id(uid), "comm_unique_id")
comm = ctypes.c_void_p()
L.check(L.comm_init(ctypes.byref(comm), 1, uid, 0), "comm_init")
g = torch.Generator().manual_seed(5)
x = torch.randn(3 * 1000 * 1000 + 17, generator=g).cuda()
ref = x.clone()
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    L.check(L.comm_allreduce_sum(comm, x.data_ptr(), x.numel(), st), "comm_allreduce_sum")
torch.cuda.synchronize()
assert torch.equal(x, ref)
L.check(L.comm_destroy(comm), "comm_destroy")
print("COMM_OK")
""" % ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "COMM_OK" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])
