"""On-device heatmap decode: ``core.inference.get_final_preds`` without the D2H copy of
the full heatmaps and the Python double loop (lib/core/inference.py:52-82)."""
from __future__ import annotations

import torch

from . import hip


def get_final_preds(batch_heatmaps: torch.Tensor, center, scale, post_process: bool = True):
    """batch_heatmaps: fp32 CUDA [B,J,H,W]; center, scale: [B,2] (array-likes, image px / 200-px units).
    Returns (preds [B,J,2] in image pixels, maxvals [B,J,1]) as CUDA tensors — 12 bytes per joint instead of
    the reference's whole heatmap leave the GPU."""
    h = batch_heatmaps
    if h.device.type != "cuda" or h.dtype != torch.float32 or h.dim() != 4:
        raise ValueError("get_final_preds expects an fp32 NCHW CUDA tensor")
    h = h.contiguous()
    B, J, H, W = h.shape
    def upload(v):      # host arrays go through pinned memory, asynchronously: the caller's thread does not wait for the stream
        t = torch.as_tensor(v, dtype=torch.float32).reshape(B, 2)
        if t.device.type == "cpu":
            t = t.contiguous().pin_memory().to(h.device, non_blocking=True)
        return t.to(h.device).contiguous()
    c, s = upload(center), upload(scale)
    preds = torch.empty(B, J, 2, dtype=torch.float32, device=h.device)
    maxvals = torch.empty(B, J, 1, dtype=torch.float32, device=h.device)
    st = torch.cuda.current_stream(h.device).cuda_stream
    hip.check(hip.lib().rsg_decode_heatmaps(h.data_ptr(), c.data_ptr(), s.data_ptr(), preds.data_ptr(),
                                            maxvals.data_ptr(), B, J, H, W, int(bool(post_process)), st), "decode")
    return preds, maxvals
