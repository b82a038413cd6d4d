"""MMETrainingModule on the MI355X HIP path (API mirror of rightLaneNetwork/trainingModules/MMETrainingModule.py).

optimizer_idx 0 (unlabelled, MMETrainingModule.py:28-33): features -> grad_reverse -> classifier -> adentropy(0.1),
run as fused HIP forward + entropy kernel; the backward applies the gradient reversal between classifier and
feature extractor (tiramisu.py:7-18).  optimizer_idx 1 (labelled, :34-37): the class-weighted CE step of
SimpleTrainModule.  configure_optimizers mirrors :15-23: [SGD-nesterov(G: features lr/3, classifier lr), AdamW(F)]
with CosineAnnealingLR(T_max=25, eta_min=lr*1e-3) each; both optimisers are single-kernel updates on the flat arena."""
import torch
from torch.optim.lr_scheduler import CosineAnnealingLR

from ..owner import FusedAdamW, FusedSGD, TrainStepFn
from .TrainingBase import TrainingBase, getClassWeight  # noqa: F401


def adentropy(output, lamda=1.0):
    """MMETrainingModule.py:10-11 (kept for API parity; the training step uses the fused HIP kernel)."""
    return lamda * torch.mean(torch.sum(output * (torch.log(output + 1e-5)), 1))


class MMETrainingModule(TrainingBase):
    def configure_optimizers(self):
        optimizerF = FusedAdamW(self, lr=self.lr, weight_decay=self.decay)
        optimizerG = FusedSGD(self, lr_feature=self.lr / 3, lr_classifier=self.lr, momentum=0.9,
                              weight_decay=self.decay)
        lr_schedulerF = CosineAnnealingLR(optimizerF, T_max=25, eta_min=self.lr * 1e-3)
        lr_schedulerG = CosineAnnealingLR(optimizerG, T_max=25, eta_min=self.lr * 1e-3)
        return [optimizerG, optimizerF], [lr_schedulerG, lr_schedulerF]

    def training_step(self, batch, batch_idx, optimizer_idx=0, drop_scales=None, seed=None):
        x_labelled, x_unlabelled, labels, _ = batch
        if optimizer_idx == 0:
            x_unlabelled, _ = self._rln_prepare_batch(x_unlabelled, None, train=True)
        else:
            x_labelled, labels = self._rln_prepare_batch(x_labelled, labels, train=True)
        params = self._rln_params_in_arena_order()
        if optimizer_idx == 0:  # unlabelled optimizer -> maximise entropy through the gradient-reversal layer
            loss, _, _ = TrainStepFn.apply(self, x_unlabelled, 0.1, drop_scales, seed, *params)
        else:                   # labelled optimizer -> class-weighted cross-entropy
            loss, _, _ = TrainStepFn.apply(self, x_labelled, labels, drop_scales, seed, *params)
        return loss
