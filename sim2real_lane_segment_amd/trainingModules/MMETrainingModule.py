"""MMETrainingModule: API mirror of rightLaneNetwork/trainingModules/MMETrainingModule.py.

The labelled branch (optimizer_idx == 1, MMETrainingModule.py:35-38) is the same fused weighted-CE
step as SimpleTrainModule.  The unlabelled branch (grad-reverse + entropy, :29-34) needs a
differentiable stand-alone feature extractor, which is the "next" row of SURVEY.md §8f and is
not built yet: it raises instead of silently running stock PyTorch operators."""
import torch
from torch.optim import SGD
from torch.optim.lr_scheduler import CosineAnnealingLR

from ..owner import FusedAdamW, TrainStepFn
from .TrainingBase import TrainingBase, getClassWeight  # noqa: F401


def adentropy(output, lamda=1.0):
    """MMETrainingModule.py:10-11 (pointwise on probabilities; kept for API parity)."""
    return lamda * torch.mean(torch.sum(output * (torch.log(output + 1e-5)), 1))


class MMETrainingModule(TrainingBase):
    def configure_optimizers(self):
        optimizerF = FusedAdamW(self, lr=self.lr, weight_decay=self.decay)
        optimizerG = SGD([
            {'params': self.featureExtractor.parameters(), 'lr': self.lr / 3},
            {'params': self.classifier.parameters(), 'lr': self.lr}
        ], lr=self.lr, weight_decay=self.decay, momentum=0.9, nesterov=True)
        lr_schedulerF = CosineAnnealingLR(optimizerF, T_max=25, eta_min=self.lr * 1e-3)
        lr_schedulerG = CosineAnnealingLR(optimizerG, T_max=25, eta_min=self.lr * 1e-3)
        return [optimizerG, optimizerF], [lr_schedulerG, lr_schedulerF]

    def training_step(self, batch, batch_idx, optimizer_idx=0):
        x_labelled, x_unlabelled, labels, _ = batch
        if optimizer_idx == 0:
            raise NotImplementedError(
                "MME unlabelled step (grad_reverse + adentropy) is not built on the HIP path yet "
                "(SURVEY.md §8f rank 2); no PyTorch-operator fallback is provided.")
        params = self._rln_params_in_arena_order()
        loss, _, _ = TrainStepFn.apply(self, x_labelled, labels, None, None, *params)
        return loss
