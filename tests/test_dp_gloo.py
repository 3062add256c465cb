"""World-size-2 gloo test of the data-parallel exchange (parallel.GradAllReducer) on CPU: the reduced flat gradient equals the
SUM of the per-shard oracle gradients (the reference's DataParallel + `losses[0].sum().backward()` semantics), and
rank r's loss equals the oracle on shard r alone — checked against the golden DP vectors produced by the reference."""
import os
import socket
import sys
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except Exception as e:          # surface failures instead of letting the parent wait for its timeout
        import traceback
        q.put((rank, False, False, [repr(e) + traceback.format_exc()]))


def _worker_body(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import yolo_oracle as yo
    from mdcv.parallel import GradAllReducer, shard_batch
    z = np.load(os.path.join(G, "mini_darknet_dp.npz"))
    os.chdir(os.path.join(G, "mini"))
    orc = yo.DarknetOracle("mini.cfg", anchors=yo.read_anchor_row("dataset/train.csv"))
    orc.load_weights("mini.weights", [18, 18])
    names = list(orc.trainable().keys())
    for k in names:
        orc.params[k].requires_grad_(True)
    x = shard_batch(torch.from_numpy(z["x"]), rank, world)
    tg = shard_batch(torch.from_numpy(z["targets"]), rank, world)
    out = orc.forward(x, tg)
    out[0].sum().backward()
    flat = torch.cat([orc.params[k].grad.reshape(-1) for k in names])
    red = GradAllReducer(lambda: flat, bucket_mb=0.05)           # several buckets on this tiny net
    assert len(red.buckets(flat)) > 3
    red.allreduce()
    losses = torch.stack([o.detach() for o in out]).numpy()
    ok_loss = np.allclose(losses, z[f"losses_{world}"][rank], rtol=5e-5)
    off0 = sum(orc.params[k].numel() for k in names[:names.index("conv0.weight")])
    g0 = flat[off0:off0 + orc.params["conv0.weight"].numel()].view_as(orc.params["conv0.weight"]).numpy()
    ok_grad = np.allclose(g0, z[f"g0_{world}"], rtol=2e-3, atol=1e-5)
    off, norms = 0, []
    for k in names:
        n = orc.params[k].numel()
        norms.append(float(flat[off:off + n].double().norm()))
        off += n
    q.put((rank, bool(ok_loss), bool(ok_grad), norms))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_sum_matches_reference_dataparallel_semantics():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    z = np.load(os.path.join(G, "mini_darknet_dp.npz"))
    for rank, ok_loss, ok_grad, norms in res:
        assert not (norms and isinstance(norms[0], str)), norms
        assert ok_loss, f"rank {rank}: per-shard loss differs from the reference"
        assert ok_grad, f"rank {rank}: reduced gradient differs from the sum of the reference's shard gradients"
    # the oracle enumerates parameters conv-weight-first per layer, the reference weight, bn.weight, bn.bias: compare as multisets
    np.testing.assert_allclose(sorted(res[0][3]), sorted(z["gnorm_2"].tolist()), rtol=2e-3)
    assert res[0][3] == res[1][3]                                  # every rank holds the same reduced gradient


# ------------------------------------------------------------------------------------------------------------------------------------------
# torchrun-transparent mode (parallel.enable_auto_data_parallel): the host logic that needs no GPU

def test_auto_slice_is_dataparallels_chunking():
    """rank r's share == torch.chunk(batch, world)[r] (nn.DataParallel's scatter, reference train.py:193-195); a rank whose chunk does not
    exist gets sample 0 with weight 0."""
    sys.path.insert(0, ROOT)
    from mdcv.parallel import auto_slice
    for world in (1, 2, 3, 4, 8):
        for batch in (1, 2, 3, 5, 8, 9, 16, 31, 32, 33):
            chunks = torch.arange(batch).chunk(world)
            covered = []
            for r in range(world):
                lo, hi, w = auto_slice(batch, r, world)
                if r < len(chunks):
                    assert w == 1.0 and list(range(lo, hi)) == chunks[r].tolist(), (world, batch, r)
                    covered += list(range(lo, hi))
                else:
                    assert (lo, hi, w) == (0, 1, 0.0), (world, batch, r)
            assert covered == list(range(batch))


def _auto_worker(rank, world, port, tmp, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                          MDCV_DP_BACKEND="gloo")
        import warnings
        from mdcv import parallel
        st = parallel.enable_auto_data_parallel()
        assert st == {"rank": rank, "world": world, "local_rank": rank}
        # (a) uneven batches: 3 samples over 2 ranks -> chunks of 2 and 1; 1 sample over 2 ranks -> rank 1 joins with weight 0
        t = torch.arange(3 * 4, dtype=torch.float32).view(3, 4)
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter("always")
            (a,), w3 = parallel.auto_shard(t)
            (b,), w1 = parallel.auto_shard(t[:1])
            (c, d), wm = parallel.auto_shard(t, t[:2])          # tensors that disagree about the batch pass through
        uneven_warned = sum("uneven" in str(x.message) for x in wl)
        # (b) the checkpoint of the unchanged script: every rank calls save_weights(path) -- rank 0 writes, all leave together
        from mdcv.yolo.models import Darknet
        os.chdir(os.path.join(G, "mini"))
        torch.manual_seed(0)
        net = Darknet("mini.cfg", 2.0, 1.6, 25.0, 0.1, False)
        net.load_weights("mini.weights", net.get_start_weight_dim())
        if rank == 1:                                            # a replica that went astray must not be the one that reaches the disk
            with torch.no_grad():
                for p in net.parameters():
                    p.add_(1.0)
        path = os.path.join(tmp, "ckpt.weights")
        net.save_weights(path)
        size_after = os.path.getsize(path)                        # exists and is complete for EVERY rank when its call returns
        q.put((rank, {"a": a.tolist(), "w3": w3, "b": b.tolist(), "w1": w1, "pass": (tuple(c.shape), tuple(d.shape), wm),
                      "uneven_warned": uneven_warned, "size": size_after, "listing": sorted(os.listdir(tmp))}))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, {"error": repr(e) + traceback.format_exc()}))


def test_auto_mode_uneven_shards_and_rank0_checkpoint(tmp_path):
    """ADVICE r5 / VERDICT r5 item 6: (i) a batch the rank count does not divide is scattered like nn.DataParallel does (no silent full-batch
    pass-through), a rank without samples joins with weight 0; (ii) under torchrun only rank 0 writes the `.weights` checkpoint -- atomically --
    and every rank's call returns with the complete file in place."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_auto_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert "error" not in res[r], res[r]["error"]
    assert res[0]["a"] == [[0, 1, 2, 3], [4, 5, 6, 7]] and res[1]["a"] == [[8, 9, 10, 11]] and res[0]["w3"] == res[1]["w3"] == 1.0
    assert res[0]["b"] == [[0, 1, 2, 3]] and res[0]["w1"] == 1.0 and res[1]["b"] == [[0, 1, 2, 3]] and res[1]["w1"] == 0.0
    assert res[0]["pass"] == ((3, 4), (2, 4), None)
    assert res[0]["uneven_warned"] == 1                           # once per process, not per batch
    ref = open(os.path.join(G, "mini", "mini.weights"), "rb").read()
    got = open(os.path.join(str(tmp_path), "ckpt.weights"), "rb").read()
    assert got == ref, "the checkpoint on disk is not rank 0's (rank 1's parameters were shifted by 1.0)"
    for r in range(world):
        assert res[r]["size"] == len(ref) and res[r]["listing"] == ["ckpt.weights"]       # no temporary file left behind
