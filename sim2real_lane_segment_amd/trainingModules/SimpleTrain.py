"""SimpleTrainModule on the MI355X HIP path (API mirror of rightLaneNetwork/trainingModules/SimpleTrain.py).

``training_step`` = forward -> class-weighted cross-entropy on the softmax probabilities -> argmax
accuracy (SimpleTrain.py:11-25), executed by librln.so; the returned loss carries a grad_fn whose
backward runs the HIP backward pass, so ``loss.backward()`` / Lightning drive it unchanged.
``configure_optimizers`` returns AdamW + CosineAnnealingLR(25, eta_min=lr/lrRatio) like
SimpleTrain.py:27-30, with AdamW fused over the flat parameter arena."""
import torch
from torch.optim.lr_scheduler import CosineAnnealingLR

from ..owner import FusedAdamW, TrainStepFn
from .TrainingBase import TrainingBase, getClassWeight  # noqa: F401


class SimpleTrainModule(TrainingBase):
    def training_step(self, batch, batch_idx, drop_scales=None, seed=None):
        x, y = batch
        x, y = self._rln_prepare_batch(x, y, train=True)
        params = self._rln_params_in_arena_order()
        loss, extra, probs = TrainStepFn.apply(self, x, y, drop_scales, seed, *params)
        if self.check_labels:
            assert float(extra[2]) == 0, f"Found more label classes than given maxClasses={self.num_cls}"
        train_acc = extra[1] * 100
        self.log('tr_loss', loss)
        self.log('tr_acc', train_acc, prog_bar=True)
        return loss

    def configure_optimizers(self):
        optimizer = FusedAdamW(self, lr=self.lr, weight_decay=self.decay)
        scheduler = CosineAnnealingLR(optimizer, 25, eta_min=self.lr / self.lrRatio)
        return [optimizer], [scheduler]


# README-era name of the same module (README.md:139)
RightLaneModule = SimpleTrainModule
