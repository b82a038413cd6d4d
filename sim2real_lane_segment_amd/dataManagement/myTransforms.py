"""Device-side mirror of the reference's `MyTransform` (rightLaneNetwork/dataManagement/myTransforms.py:6-31) for the
non-augmenting branch: Resize(height, width) -> [ToGray] -> Normalize() -> ToTensorV2, executed by one HIP kernel
(`rln_preprocess_u8`) on uint8 frames that already sit in HBM; no CPU fallback.

Same constructor and call signature as the reference; additionally accepts a whole batch [N, H, W, 3] at once.
`augment=True` runs the augmenting branch (HueSaturationValue -> RandomSizedCrop -> OneOf(MotionBlur, GaussNoise),
myTransforms.py:8-13) on device through `rln_augment_u8`: the per-image random parameters are drawn on the host
(`sample_aug_params`, numpy Generator, `seed=` for reproducibility), the pixels never leave HBM.  The arithmetic
follows the published definitions of albumentations 0.5.2 / cv2 (absent here): parity unpinned.

Inside a Dataset (the reference calls ``self.transform(x, y)`` from ``RightLaneDataset.__getitem__`` under
``DataLoader(num_workers=8, pin_memory=True)``, myDatasets.py:59, dataModules.py:52-53) the call sees host numpy arrays and
must not touch the GPU: forked workers cannot use the parent's HIP context, and pin_memory rejects device tensors.  Such a
call -- host input WITH a label argument -- is DEFERRED: it returns the raw uint8 frame as a CPU tensor (plus the label:
nearest-neighbour resized by pure indexing when the transform does not augment, raw when it does, because the random crop
is drawn per batch on the device side), and the training module runs the device transform on the whole batch after the
loader (``prepare_batch``, called by TrainingBase.forward / training_step / evaluate_batch when a uint8 NHWC batch arrives).
Calls without a label (makeDemoVideo.py:30, test.py:59) and calls on device tensors transform at once."""
import weakref
import ctypes

import numpy as np
import torch

from .. import _lib

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


AUG_NP = 80  # floats per image in the parameter table of rln_augment_u8 (layout: include/rln.h)


def _line_kernel(ksize, xs, ys, xe, ye):
    """cv2.line(kernel, (xs, ys), (xe, ye), 1, thickness=1) on a ksize x ksize grid (Bresenham), normalised."""
    k = np.zeros((ksize, ksize), np.float32)
    dx, dy = abs(xe - xs), -abs(ye - ys)
    sx, sy = (1 if xs < xe else -1), (1 if ys < ye else -1)
    err, x, y = dx + dy, xs, ys
    while True:
        k[y, x] = 1.0
        if x == xe and y == ye:
            break
        e2 = 2 * err
        if e2 >= dy:
            err += dy
            x += sx
        if e2 <= dx:
            err += dx
            y += sy
    return k / k.sum()


def sample_aug_params(n, hs, ws, height, width, rng):
    """Per-image random parameters of the reference's augmentation pipeline (myTransforms.py:8-13), drawn from the
    distributions albumentations 0.5.2 documents: HueSaturationValue shifts U(-20,20), U(-30,30), U(-20,20);
    RandomSizedCrop height randint(height//2, height*4) (clamped to the frame), width = height * w/h, position
    uniform; OneOf(MotionBlur(3|5|7 line kernel), GaussNoise(var U(10,50))) with equal odds."""
    out = np.zeros((n, AUG_NP), np.float32)
    for i in range(n):
        out[i, 0] = rng.uniform(-20, 20)
        out[i, 1] = rng.uniform(-30, 30)
        out[i, 2] = rng.uniform(-20, 20)
        lo, hi = min(height // 2, hs), min(height * 4, hs)
        ch = int(rng.integers(lo, hi + 1))
        cw = min(int(ch * width / height), ws)
        out[i, 3] = int((hs - ch) * rng.random())
        out[i, 4] = int((ws - cw) * rng.random())
        out[i, 5], out[i, 6] = ch, cw
        if rng.random() < 0.5:
            ksize = int(rng.choice([3, 5, 7]))
            xs, xe = int(rng.integers(0, ksize)), int(rng.integers(0, ksize))
            if xs == xe:
                ys, ye = (int(v) for v in rng.choice(ksize, 2, replace=False))
            else:
                ys, ye = int(rng.integers(0, ksize)), int(rng.integers(0, ksize))
            k = _line_kernel(ksize, xs, ys, xe, ye)
            off = (7 - ksize) // 2
            full = np.zeros((7, 7), np.float32)
            full[off:off + ksize, off:off + ksize] = k
            out[i, 7], out[i, 8] = 0, ksize
            out[i, 16:16 + 49] = full.reshape(-1)
        else:
            out[i, 7] = 1
            out[i, 9] = np.sqrt(rng.uniform(10, 50))
            out[i, 10] = int(rng.integers(0, 1 << 24))
    return out


def nearest_resize_index(src, dst):
    """cv2 INTER_NEAREST source index of every destination index: min(floor(d * src / dst), src - 1) (the arithmetic of
    rln_preprocess_u8 / rln_augment_u8 for label masks)."""
    return np.minimum(np.floor(np.arange(dst) * (src / dst)).astype(np.int64), src - 1)


class MyTransform:
    _instances = []  # weak references, in construction order (prepare_batch picks the transform of a deferred batch)

    def __init__(self, width=160, height=120, gray=False, augment=False, device="cuda", seed=None):
        if augment and gray:
            raise NotImplementedError("augment=True with gray=True is not built (no reference script uses it)")
        self.width, self.height, self.gray, self.augment = int(width), int(height), bool(gray), bool(augment)
        self.device = torch.device(device)
        self.rng = np.random.default_rng(seed)
        MyTransform._instances.append(weakref.ref(self))

    @classmethod
    def latest(cls, augment=None, size=None):
        """Most recently constructed live transform (optionally: with this augment flag / this (height, width))."""
        cls._instances = [r for r in cls._instances if r() is not None]
        for r in reversed(cls._instances):
            t = r()
            if (augment is None or t.augment == bool(augment)) and (size is None or (t.height, t.width) == tuple(size)):
                return t
        return None

    def _deferred(self, img, label):
        """Dataset-side half of the transform (no GPU): raw uint8 frame + label (see the module docstring)."""
        x = torch.from_numpy(np.ascontiguousarray(img)) if isinstance(img, np.ndarray) else img.contiguous()
        if x.dtype != torch.uint8 or x.ndim != 3 or x.shape[-1] != 3:
            raise ValueError("expected a uint8 frame [H, W, 3]")
        if label is None or getattr(label, "ndim", 0) < 2:
            return x, label  # unlabelled sample: the reference hands its empty label tensor through (myTransforms.py:27-31)
        y = np.asarray(label)
        if y.dtype != np.uint8:
            raise ValueError("expected a uint8 label mask")
        y = y.reshape(x.shape[0], x.shape[1])
        if self.augment:
            return x, torch.from_numpy(np.ascontiguousarray(y))
        sy = nearest_resize_index(y.shape[0], self.height)
        sx = nearest_resize_index(y.shape[1], self.width)
        return x, torch.from_numpy(y[sy][:, sx].astype(np.int64))

    @classmethod
    def prepare_batch(cls, x, y=None, train=False, device=None):
        """Device-side half: a deferred batch (uint8 [N, H, W, 3] frames as collated by the DataLoader) -> (float32
        [N, 3, h, w], labels).  Anything else passes through unchanged.  Raw uint8 labels of the frame size mark a batch of
        an augmenting transform (the crop has to match: image and mask are transformed together, with parameters drawn
        here, once per batch); int64 labels are final already and name the output size."""
        if not (torch.is_tensor(x) and x.dtype == torch.uint8 and x.ndim == 4 and x.shape[-1] == 3):
            return x, y
        raw_labels = torch.is_tensor(y) and y.dtype == torch.uint8 and y.ndim == 3 and tuple(y.shape[1:]) == tuple(x.shape[1:3])
        if raw_labels:
            t = cls.latest(augment=True)
            if t is None:
                raise RuntimeError("a batch with raw uint8 label masks needs an augmenting MyTransform (none is alive)")
        else:
            size = tuple(y.shape[-2:]) if (torch.is_tensor(y) and y.ndim == 3 and y.dtype == torch.int64) else None
            t = cls.latest(augment=False, size=size) or cls.latest(size=size)
            if t is None:
                raise RuntimeError("a uint8 frame batch arrived but no MyTransform is alive to say how to transform it")
        dev = torch.device(device) if device is not None else (x.device if x.is_cuda else t.device)
        if dev.type != "cuda":
            raise RuntimeError("the input transform runs on the GPU only (HIP kernels): move the module with .cuda()")
        run = t if t.device == dev else cls._clone_on(t, dev)
        xd = x.to(dev, non_blocking=True)
        if raw_labels:
            return run(xd, y.to(dev, non_blocking=True))
        out, _ = run._immediate(xd, None, None, force_plain=True)
        return out, (y.to(dev, non_blocking=True) if torch.is_tensor(y) else y)

    @staticmethod
    def _clone_on(t, dev):
        c = MyTransform.__new__(MyTransform)
        c.__dict__.update(t.__dict__)
        c.device = dev
        return c

    def _augment(self, x, y_in, n, hs, ws, have_label, params=None):
        if params is None:
            params = sample_aug_params(n, hs, ws, self.height, self.width, self.rng)
        params = np.ascontiguousarray(params, np.float32)
        if params.shape != (n, AUG_NP):
            raise ValueError("params must be [N, 80]")
        if (params[:, 3] < 0).any() or (params[:, 4] < 0).any() or (params[:, 5] < 1).any() or (params[:, 6] < 1).any() \
                or (params[:, 3] + params[:, 5] > hs).any() or (params[:, 4] + params[:, 6] > ws).any():
            raise ValueError("crop box outside the frame")
        self.last_params = params
        # pinned staging + asynchronous copy: a pageable copy would block the host until the stream has drained, i.e.
        # serialise the host's launch work with the previous step's kernels (torch's host allocator keeps the pinned
        # block alive until the copy has run)
        pd = torch.from_numpy(params).pin_memory().to(self.device, non_blocking=True)
        out = torch.empty((n, 3, self.height, self.width), dtype=torch.float32, device=self.device)
        tmp = torch.empty((n, self.height, self.width, 3), dtype=torch.uint8, device=self.device)
        y_out = torch.empty((n, self.height, self.width), dtype=torch.int64, device=self.device) if have_label else None
        mean = (ctypes.c_float * 3)(*MEAN)
        std = (ctypes.c_float * 3)(*STD)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().rln_augment_u8(x.data_ptr(), n, hs, ws, y_in.data_ptr() if have_label else None,
                                             self.height, self.width, pd.data_ptr(), mean, std, tmp.data_ptr(),
                                             out.data_ptr(), y_out.data_ptr() if have_label else None, stream),
                   "rln_augment_u8")
        self._keep = (x, y_in, pd, tmp)
        return out, y_out

    def __call__(self, img, label=None, params=None):
        on_host = isinstance(img, np.ndarray) or (torch.is_tensor(img) and not img.is_cuda)
        if on_host and label is not None and img.ndim == 3:
            return self._deferred(img, label)  # the Dataset protocol: no GPU work in (worker) processes
        return self._immediate(img, label, params)

    def _immediate(self, img, label=None, params=None, force_plain=False):
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
        if self.augment and not force_plain:
            out, y_out = self._augment(x, y_in, n, hs, ws, have_label, params)
            if single:
                return out[0], (y_out[0] if have_label else label)
            return out, (y_out if have_label else label)
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
