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
