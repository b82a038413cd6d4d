"""Inference tail of the reference's demo script on device (makeDemoVideo.py:15-47).

``predict_frames`` = the body of ``predictVideo``'s loop for a whole batch of frames that already sit in HBM:
``MyTransform(augment=False)`` -> ``model.forward`` -> ``torch.max(., 1)`` -> painted, resized BGR frame, with the
transform (rln_preprocess_u8), the network (rln_forward) and argmax + resize + painting (rln_overlay_u8) all in
librln.so; video decode / encode (cv2.VideoCapture / VideoWriter) stays on the host and is out of scope."""
import ctypes

import torch

from . import _lib

# makeDemoVideo.py:42-44 (BGR): right lane green, left lane blue, obstacles red; background keeps the frame
DEMO_COLORS = ((0, 0, 0), (0, 255, 0), (255, 0, 0), (0, 0, 255))
DEMO_PAINT = (1, 2, 3)


def overlay(frames: torch.Tensor, probs: torch.Tensor, colors=DEMO_COLORS, paint=DEMO_PAINT, want_pred=False):
    """frames uint8 [N,Hs,Ws,3] (cuda), probs float32 [N,K,H,W] (cuda) -> uint8 [N,H,W,3] (and uint8 [N,H,W] classes)."""
    if frames.device.type != "cuda" or probs.device.type != "cuda":
        raise RuntimeError("overlay runs on the GPU only (HIP kernel); no CPU fallback")
    frames = frames.contiguous()
    probs = probs.float().contiguous()
    n, hs, ws, _ = frames.shape
    _, k, h, w = probs.shape
    if len(colors) < k:
        raise ValueError(f"need a colour for each of the {k} classes")
    table = (ctypes.c_uint8 * (3 * k))(*[int(v) for c in colors[:k] for v in c])
    mask = 0
    for c in paint:
        mask |= 1 << int(c)
    out = torch.empty((n, h, w, 3), dtype=torch.uint8, device=frames.device)
    pred = torch.empty((n, h, w), dtype=torch.uint8, device=frames.device) if want_pred else None
    _lib.check(_lib.lib().rln_overlay_u8(
        ctypes.c_void_p(frames.data_ptr()), n, hs, ws, ctypes.c_void_p(probs.data_ptr()), k, h, w, table, mask,
        ctypes.c_void_p(out.data_ptr()), None if pred is None else ctypes.c_void_p(pred.data_ptr()),
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "rln_overlay_u8")
    return (out, pred) if want_pred else out


@torch.no_grad()
def predict_frames(model, frames: torch.Tensor, transform=None, want_pred=False):
    """frames uint8 [N,Hs,Ws,3] BGR on the GPU -> painted uint8 [N,120,160,3] frames (makeDemoVideo.py:26-46)."""
    from .dataManagement.myTransforms import MyTransform
    transform = transform or MyTransform(augment=False)
    if hasattr(model, "rln_eval_cache") and not model.training and not getattr(model, "_rln_demo_cached", False):
        model.rln_eval_cache(True)  # the video loop runs a frozen model: weight fragments / BN tables are built once
        object.__setattr__(model, "_rln_demo_cached", True)
    x, _ = transform(frames)
    return overlay(frames, model.forward(x), want_pred=want_pred)
