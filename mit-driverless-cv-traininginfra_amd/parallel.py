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
    """All-reduce(SUM) of a model's flat gradient buffer.

    Two ways to use it:
      * `GradAllReducer(lambda: flat)` + `allreduce()` after backward — one exchange, queued on a side stream.
      * `GradAllReducer.attach(model)` — OVERLAPPED: the backward launch list carries "gradients >= offset are enqueued"
        markers (engine.Plan.mark_ready); each marker lets the comm stream (ordered behind the compute stream by an event)
        all-reduce every bucket that lies entirely above the mark, while the rest of backward keeps the CUs busy.
        Call `finish()` before the optimizer step.
    """

    def __init__(self, flat_grad_fn=None, bucket_mb=64.0, average=False, group=None, allreduce_fn=None):
        self.flat_grad_fn = flat_grad_fn
        self.bucket_elems = max(1, int(bucket_mb * (1 << 20) / 4))
        self.average = average
        self.group = group
        self._stream = None
        self._allreduce = allreduce_fn            # injectable for tests; default torch.distributed.all_reduce
        self._flat = None
        self._pending_hi = 0
        self._overlap = False
        self.extra_streams = []                   # compute-side streams besides the current one (the plans' wgrad stream)
        self.log = []
        self.timing = False                       # record HIP events around every bucket (bench.py: all-reduce ms / exposed ms per step)
        self._ev_steps = []                       # per step: ([(start, end) per bucket on the comm stream], backward-end event on the compute stream)
        self._ev_cur = None

    # ---------------------------------------------------------------- plain use
    def buckets(self, flat):
        n = flat.numel()
        return [flat[i:min(n, i + self.bucket_elems)] for i in range(0, n, self.bucket_elems)]

    def _active(self):
        return self._allreduce is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1)

    def _reduce(self, t):
        if self._allreduce is not None:
            self._allreduce(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def allreduce(self):
        """Sum the gradient over ranks, in place, after backward (no overlap)."""
        if not self._active():
            return
        flat = self.flat_grad_fn()
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if flat.is_cuda:
            if self._stream is None:
                from .engine import checked_stream
                self._stream = checked_stream(flat.device, [torch.cuda.current_stream(flat.device)], "comm")
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                for b in self.buckets(flat):
                    self._reduce(b)
                if self.average:
                    flat.div_(world)
            torch.cuda.current_stream().wait_stream(self._stream)
        else:
            for b in self.buckets(flat):
                self._reduce(b)
            if self.average:
                flat.div_(world)

    # ---------------------------------------------------------------- overlapped use
    @classmethod
    def attach(cls, model, **kw):
        red = cls(lambda: model.flat_parameters()[1], **kw)
        model._dp_reducer = red
        return red

    def begin(self, flat, overlap):
        self._flat = flat
        self._overlap = overlap and self._active()
        self._pending_hi = flat.numel()                 # buckets are cut from the END of the buffer (last layers finish first)
        self.log = []
        self._ev_cur = [] if (self.timing and flat.is_cuda and self._active()) else None
        if self._active() and flat.is_cuda and self._stream is None:
            # a stream that was checked to run beside the compute streams (HIP multiplexes streams onto a few hardware queues: a stream that
            # shares the main stream's queue would serialise the exchange with backward; engine.checked_stream)
            from .engine import checked_stream
            self._stream = checked_stream(flat.device, [torch.cuda.current_stream(flat.device)] + list(self.extra_streams), "comm")

    def _fire(self, lo):
        """all-reduce [lo, pending_hi) on the comm stream, ordered after everything enqueued so far on the compute stream"""
        hi = self._pending_hi
        if hi <= lo:
            return
        seg = self._flat[lo:hi]
        self.log.append((lo, hi))
        if seg.is_cuda:
            self._stream.wait_stream(torch.cuda.current_stream())
            for extra in self.extra_streams:
                self._stream.wait_stream(extra)
            with torch.cuda.stream(self._stream):
                if self._ev_cur is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self._stream)
                    self._reduce(seg)
                    e1.record(self._stream)
                    self._ev_cur.append((e0, e1))
                else:
                    self._reduce(seg)
        else:
            self._reduce(seg)
        self._pending_hi = lo

    def would_fire(self, low_water):
        """True if on_ready(low_water) would start at least one bucket (the plans enqueue their deferred side-stream work first)."""
        return bool(self._overlap) and self._pending_hi - self.bucket_elems >= low_water

    def on_ready(self, low_water):
        if not self._overlap:
            return
        while self._pending_hi - self.bucket_elems >= low_water:
            self._fire(self._pending_hi - self.bucket_elems)

    def backward_done(self):
        if not self._active():
            return
        self._fire(0)                                    # whatever is left (all of it when overlap is off)
        if self._ev_cur:
            done = torch.cuda.Event(enable_timing=True)      # every compute kernel of this backward is in front of this point
            done.record(torch.cuda.current_stream())
            self._ev_steps.append((self._ev_cur, done))
            self._ev_cur = None

    def finish(self):
        """Order the compute stream behind the exchange (call before optimizer.step())."""
        if not self._active() or self._flat is None:
            return
        if self._flat.is_cuda and self._stream is not None:
            if self.average:
                with torch.cuda.stream(self._stream):
                    self._flat.div_(dist.get_world_size(self.group) if dist.is_initialized() else 1)
            torch.cuda.current_stream().wait_stream(self._stream)
        elif self.average:
            self._flat.div_(dist.get_world_size(self.group) if dist.is_initialized() else 1)


    def pop_comm_stats(self):
        """Averages over the steps recorded since the last call (timing=True): time the comm stream spent inside all-reduce
        calls per step, the span first-bucket-start .. last-bucket-end, and the EXPOSED part — how long after the last compute
        kernel of backward the last bucket finished, i.e. what `finish()` makes the optimizer step wait for."""
        steps, self._ev_steps = self._ev_steps, []
        if not steps:
            return None
        torch.cuda.synchronize()
        busy = span = exposed = 0.0
        for pairs, done in steps:
            busy += sum(a.elapsed_time(b) for a, b in pairs)
            span += pairs[0][0].elapsed_time(pairs[-1][1])
            exposed += max(0.0, done.elapsed_time(pairs[-1][1]))
        n = len(steps)
        return {"steps": n, "buckets_per_step": len(steps[0][0]), "allreduce_busy_ms": busy / n, "allreduce_span_ms": span / n,
                "exposed_comm_ms": exposed / n}


def shard_batch(t, rank, world):
    """Rank r's contiguous shard of the batch dimension (DataParallel's scatter on dim 0)."""
    per = t.shape[0] // world
    return t[rank * per:(rank + 1) * per]


# ---------------------------------------------------------------------------------------------------------------------------------------
# torchrun-transparent data parallel: `python -m torch.distributed.run --nproc-per-node N train.py ...` on the UNCHANGED reference script.
#
# The reference's only multi-GPU path is `if torch.cuda.device_count() > 1: model = nn.DataParallel(model)` (CVC-YOLOv3/train.py:193-195): one
# process, the module tree replicated onto every GPU per forward.  The MI355X-native path is one process per GPU, so under torchrun the
# drop-in modules (dropin/CVC-YOLOv3/models.py, dropin/RektNet/*.py) call enable_auto_data_parallel() when they are imported -- i.e. before
# the script touches a device:
#   * the process is pinned to ITS GPU by HIP_VISIBLE_DEVICES, so torch.cuda.device_count() == 1 and train.py:193 does not wrap the model;
#   * the process group is initialised from torchrun's environment (backend MDCV_DP_BACKEND, default "nccl" == RCCL);
#   * the model takes rank r's shard of each (imgs, targets) batch inside forward() -- DataParallel's scatter on dim 0 (train.py:68) --,
#     attaches the overlapped gradient all-reduce (SUM, like `losses[0].sum().backward()` over the replicas' losses, train.py:70) on its
#     first training step and joins the comm stream at the END of backward(), so the script's stock `optimizer.step()` (train.py:72) reads
#     reduced gradients.
# MDCV_AUTO_DP=0 switches all of it off; MDCV_DP_DEVICE=<index> overrides the device choice (tests: several ranks on one GPU over gloo).
_AUTO = None


def enable_auto_data_parallel():
    """-> {"rank", "world", "local_rank"} when this process is one of several torchrun ranks (and MDCV_AUTO_DP != 0), else None.  Idempotent."""
    global _AUTO
    import os
    if _AUTO is not None:
        return _AUTO or None
    world = int(os.environ.get("WORLD_SIZE", "1") or 1)
    if world <= 1 or "LOCAL_RANK" not in os.environ or "RANK" not in os.environ or os.environ.get("MDCV_AUTO_DP", "1") == "0":
        _AUTO = False
        return None
    rank, local = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
    if not torch.cuda.is_initialized():
        override = os.environ.get("MDCV_DP_DEVICE")
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
        devs = [d for d in vis.split(",") if d.strip()] if vis else None
        choice = override if override is not None else (devs[local] if devs and local < len(devs) else str(local))
        os.environ["HIP_VISIBLE_DEVICES"] = choice              # one visible device: train.py:193 sees device_count() == 1
        os.environ.pop("CUDA_VISIBLE_DEVICES", None)            # (both filters would apply one after the other)
    else:
        import warnings
        warnings.warn("mdcv: the GPU runtime was initialised before the drop-in modules were imported; this rank cannot hide the other GPUs "
                      "any more (import `models` / `keypoint_net` before the first torch.cuda call)", RuntimeWarning, stacklevel=2)
        if torch.cuda.device_count() > local:
            torch.cuda.set_device(local)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # RCCL across processes needs dmabuf IPC on this driver
    if not dist.is_initialized():
        dist.init_process_group(os.environ.get("MDCV_DP_BACKEND", "nccl"), rank=rank, world_size=world)
    _AUTO = {"rank": rank, "world": world, "local_rank": local}
    return _AUTO


def auto_state():
    """The auto-data-parallel state set by enable_auto_data_parallel() (None when off)."""
    return _AUTO or None


def auto_slice(batch, rank=None, world=None):
    """(start, stop, weight) of rank r's share of a batch of `batch` samples: nn.DataParallel's scatter on dim 0 (torch.chunk: chunks of
    ceil(batch / world) samples, the last one shorter, trailing replicas EMPTY when the batch is short -- the reference DataLoaders set no
    drop_last, so this happens on the last batch of an epoch).  A rank whose chunk is empty still has to take part in the gradient exchange: it
    gets sample 0 with weight 0.0 -- its forward runs, its outputs are multiplied by zero, its gradients are exact zeros -- which is what
    DataParallel computes when it uses fewer replicas."""
    st = auto_state()
    if rank is None:
        rank, world = st["rank"], st["world"]
    per = -(-batch // world)
    lo, hi = rank * per, min(batch, (rank + 1) * per)
    if lo >= hi:
        return 0, 1, 0.0
    return lo, hi, 1.0


def auto_shard(*tensors):
    """Rank r's share (auto_slice) of every tensor along dim 0; everything passes through when the mode is off or the tensors disagree about
    the batch size.  -> (tuple of tensors, weight): weight None = not sharded, 1.0 = a real shard, 0.0 = this rank's chunk was empty."""
    st = auto_state()
    if st is None:
        return tensors, None
    sizes = {t.shape[0] for t in tensors if t is not None and t.dim() >= 1}
    if len(sizes) != 1 or any(t is not None and t.dim() < 1 for t in tensors):
        return tensors, None
    batch = sizes.pop()
    if batch < 1:
        return tensors, None
    lo, hi, weight = auto_slice(batch)
    global _WARNED_UNEVEN
    if batch % st["world"] and not _WARNED_UNEVEN:
        _WARNED_UNEVEN = True
        import warnings
        warnings.warn(f"mdcv: batch of {batch} over {st['world']} ranks: uneven shards (DataParallel's chunking; per-shard mean losses weigh the "
                      "samples of a short shard more, as nn.DataParallel does)", RuntimeWarning, stacklevel=3)
    return tuple(None if t is None else t[lo:hi] for t in tensors), weight


_WARNED_UNEVEN = False


def auto_attach(model, average=False):
    """First training forward of `model` under the auto mode: the overlapped all-reduce is attached WHETHER OR NOT this batch divided evenly (a first
    batch that did not used to leave N ranks training independently), and the replicas are made identical -- parameters and buffers broadcast from
    rank 0 -- instead of trusting every process to have seeded alike (KeypointNet draws its initial weights at random)."""
    if getattr(model, "_dp_reducer", None) is not None or auto_state() is None:
        return
    GradAllReducer.attach(model, average=average)
    model._dp_auto = True                                     # backward() joins the comm stream itself: the script calls a stock optimizer.step()
    with torch.no_grad():
        dist.broadcast(model.flat_parameters()[0], 0)
        for b in model.buffers():
            if b.numel():
                dist.broadcast(b, 0)


def auto_is_writer():
    """True where a checkpoint should be written: everywhere outside the auto mode, on rank 0 inside it."""
    st = auto_state()
    return st is None or st["rank"] == 0


def auto_barrier():
    st = auto_state()
    if st is not None and dist.is_initialized():
        dist.barrier()
