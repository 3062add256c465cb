"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce(SUM) of the flat gradient buffer over xGMI.

The reference's only multi-GPU mechanism is single-process nn.DataParallel (CVC-YOLOv3/train.py:193-195), whose semantics
are: every replica computes its shard's (mean-reduced) loss with per-shard BatchNorm statistics and per-shard
build_targets, and the gradients of the shards are SUMMED (train.py:70 `losses[0].sum().backward()`).  This module keeps
exactly that contract (rank r's loss == single-GPU run on shard r; reduced grad == sum over shards; `average=True` divides
by the world size instead) with `torch.distributed` (backend "nccl" == RCCL on ROCm; "gloo" for the CPU tests).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): the 248 MB fp32 YOLO gradient is sent as a few large buckets so each
ring step moves >= 32 MB per link; RektNet's 1.25 MB is a single latency-bound bucket.
"""
import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, flat_grad_fn, bucket_mb=64.0, average=False, group=None):
        """flat_grad_fn: callable returning the flat fp32 gradient tensor (e.g. lambda: model.flat_parameters()[1])."""
        self.flat_grad_fn = flat_grad_fn
        self.bucket_elems = max(1, int(bucket_mb * (1 << 20) / 4))
        self.average = average
        self.group = group
        self._stream = None

    def buckets(self, flat):
        n = flat.numel()
        return [flat[i:min(n, i + self.bucket_elems)] for i in range(0, n, self.bucket_elems)]

    def allreduce(self):
        """Sum the gradient over ranks, in place.  Buckets are queued on a side stream (GPU) so the optimizer on the main stream
        only waits for the last one; returns after ordering the main stream behind the exchange."""
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        flat = self.flat_grad_fn()
        world = dist.get_world_size(self.group)
        if flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                for b in self.buckets(flat):
                    dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group)
                if self.average:
                    flat.div_(world)
            torch.cuda.current_stream().wait_stream(self._stream)
        else:
            for b in self.buckets(flat):
                dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group)
            if self.average:
                flat.div_(world)


def shard_batch(t, rank, world):
    """Rank r's contiguous shard of the batch dimension (DataParallel's scatter on dim 0)."""
    per = t.shape[0] // world
    return t[rank * per:(rank + 1) * per]
