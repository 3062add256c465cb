"""GPU drop-in for the reference's CVC-YOLOv3/utils/nms.py (`nms`, nms.py:4-61).

Same signature and return value (the kept indices with respect to `boxes`, highest score first) but the whole greedy
loop runs in one HIP workgroup (csrc/postprocess.hip): radix-select of the top_k scores, LDS sort, a 64-bit suppression
matrix built with wave ballots and a single-wave scan over it.  No CPU fallback.
"""
import torch

from ... import _lib

MAX_TOPK = 512


def nms(boxes, scores, overlap=0.5, top_k=200):
    """boxes [n,4] corner format, scores [n], both on the GPU.  Returns a LongTensor of kept indices.

    Equal scores are visited by descending index (what the reference's ascending sort + walk-from-the-back gives with a
    stable sort; its CPU sort is not stable past 16 elements, so ties are implementation-defined there).
    Reading the number of kept boxes back costs one device sync, like the reference's dynamic-shape loop."""
    _lib.require_gpu(scores)
    L = _lib.lib()
    n = int(scores.shape[0])
    dev = scores.device
    if boxes.numel() == 0:                                   # nms.py:17-18 returns the (empty) keep buffer
        return torch.zeros(n, dtype=torch.long, device=dev)
    if not 0 < int(top_k) <= MAX_TOPK:
        raise ValueError(f"nms: top_k must be in 1..{MAX_TOPK} (got {top_k})")
    bx = boxes.detach().to(torch.float32).contiguous()
    sc = scores.detach().to(torch.float32).contiguous()
    keep = torch.empty(min(n, int(top_k)), dtype=torch.long, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    ws = torch.empty(int(L.nms_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    L.check(L.nms(bx.data_ptr(), sc.data_ptr(), n, float(overlap), int(top_k), keep.data_ptr(), count.data_ptr(), ws.data_ptr(),
                  torch.cuda.current_stream().cuda_stream), "nms")
    return keep[:int(count.item())]
