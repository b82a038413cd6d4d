"""Device-side mirror of the reference's `MyTransform` (rightLaneNetwork/dataManagement/myTransforms.py:6-31) for the
non-augmenting branch: Resize(height, width) -> [ToGray] -> Normalize() -> ToTensorV2, executed by one HIP kernel
(`rln_preprocess_u8`) on uint8 frames that already sit in HBM; no CPU fallback.

Same constructor and call signature as the reference; additionally accepts a whole batch [N, H, W, 3] at once.
`augment=True` (HueSaturationValue / RandomSizedCrop / MotionBlur / GaussNoise from albumentations) is not built:
it raises NotImplementedError instead of silently skipping the augmentation."""
import ctypes

import numpy as np
import torch

from .. import _lib

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


class MyTransform:
    def __init__(self, width=160, height=120, gray=False, augment=False, device="cuda"):
        if augment:
            raise NotImplementedError("augment=True (albumentations HSV jitter / RandomSizedCrop / blur / noise) is "
                                      "not available on device yet; use augment=False")
        self.width, self.height, self.gray = int(width), int(height), bool(gray)
        self.device = torch.device(device)

    def __call__(self, img, label=None):
        single = (img.ndim == 3)
        x = torch.as_tensor(np.ascontiguousarray(img) if isinstance(img, np.ndarray) else img)
        if x.dtype != torch.uint8 or x.shape[-1] != 3:
            raise ValueError("expected uint8 frames [H, W, 3] or [N, H, W, 3]")
        x = x.to(self.device).contiguous()
        if single:
            x = x.unsqueeze(0)
        n, hs, ws, _ = x.shape
        have_label = label is not None and getattr(label, "ndim", 0) >= 2
        y_in = None
        if have_label:
            y_in = torch.as_tensor(np.ascontiguousarray(label) if isinstance(label, np.ndarray) else label)
            if y_in.dtype != torch.uint8:
                raise ValueError("expected a uint8 label mask")
            y_in = y_in.to(self.device).contiguous().reshape(n, hs, ws)
        out = torch.empty((n, 3, self.height, self.width), dtype=torch.float32, device=self.device)
        y_out = torch.empty((n, self.height, self.width), dtype=torch.int64, device=self.device) if have_label else None
        mean = (ctypes.c_float * 3)(*MEAN)
        std = (ctypes.c_float * 3)(*STD)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().rln_preprocess_u8(x.data_ptr(), n, hs, ws, y_in.data_ptr() if have_label else None,
                                                self.height, self.width, int(self.gray), mean, std, out.data_ptr(),
                                                y_out.data_ptr() if have_label else None, stream),
                   "rln_preprocess_u8")
        self._keep = (x, y_in)  # inputs stay alive until the stream has consumed them
        if single:
            return out[0], (y_out[0] if have_label else label)
        return out, (y_out if have_label else label)
