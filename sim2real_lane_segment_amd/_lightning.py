"""LightningModule base: the real one when pytorch_lightning is importable, otherwise a minimal
stand-in exposing what the reference scripts use (train.py:49-75, test.py:28-36,
makeDemoVideo.py:53-61): ``save_hyperparameters``, ``hparams``, ``log``, ``device``,
``load_from_checkpoint``."""
import inspect

import torch
import torch.nn as nn

try:  # pragma: no cover - depends on the environment
    from pytorch_lightning import LightningModule  # type: ignore
    HAVE_LIGHTNING = True
except Exception:  # ModuleNotFoundError here and on the GPU box
    HAVE_LIGHTNING = False

    class _AttrDict(dict):
        __getattr__ = dict.get

        def __setattr__(self, k, v):
            self[k] = v

    class LightningModule(nn.Module):
        def __init__(self, *args, **kwargs):
            super().__init__()
            self.__dict__["_hparams"] = _AttrDict()
            self.__dict__["_logged"] = {}

        @property
        def hparams(self):
            return self.__dict__["_hparams"]

        def save_hyperparameters(self, *names):
            frame = inspect.currentframe().f_back
            local_vars = frame.f_locals
            if not names:
                names = [k for k in local_vars if k not in ("self", "__class__")]
            for k in names:
                if k in local_vars:
                    self.hparams[k] = local_vars[k]

        def log(self, name, value, *args, **kwargs):
            self.__dict__["_logged"][name] = value

        @property
        def logged_metrics(self):
            return self.__dict__["_logged"]

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, **kwargs):
            """Reads a Lightning-style checkpoint ({'state_dict', 'hyper_parameters'}) or a bare state_dict.

            Weights-only: the file is parsed by ``torch.load(..., weights_only=True)`` and nothing else, so a checkpoint
            can never execute code.  ``hyper_parameters`` saved as an ``argparse.Namespace`` (Lightning does that for
            argparse-driven scripts) are admitted through ``torch.serialization.safe_globals``; any other pickled
            object makes the load fail with an error naming the file."""
            import argparse
            try:
                with torch.serialization.safe_globals([argparse.Namespace]):
                    ckpt = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=True)
            except Exception as exc:
                raise RuntimeError(
                    f"checkpoint {checkpoint_path!r} could not be read by the weights-only loader ({exc}); "
                    f"refusing to unpickle arbitrary objects. Re-save it as a plain state_dict / "
                    f"{{'state_dict', 'hyper_parameters'}} of tensors and numbers.") from exc
            hp = ckpt.get("hyper_parameters", {}) if isinstance(ckpt, dict) else {}
            hp = dict(vars(hp)) if isinstance(hp, argparse.Namespace) else dict(hp)
            hp.update(kwargs)
            sig = inspect.signature(cls.__init__).parameters
            model = cls(**{k: v for k, v in hp.items() if k in sig})
            state = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
            model.load_state_dict(state, strict=strict)
            return model
