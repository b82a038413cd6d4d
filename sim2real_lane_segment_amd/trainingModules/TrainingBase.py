"""TrainingBase on the MI355X HIP path (API mirror of rightLaneNetwork/trainingModules/TrainingBase.py).

Same constructor, attributes, hooks and CLI flags (TrainingBase.py:27-52); ``forward`` and the
evaluation step run as fused HIP kernels; metrics come from the confusion matrix the loss kernel
produces on the device (no per-step host sync such as the reference's getClassWeight loop,
TrainingBase.py:12-23)."""
from argparse import ArgumentParser

import torch

from .._lightning import LightningModule
from ..metrics import accuracy_from_confusion, dice_from_confusion, iou_from_confusion
from ..models.FCDenseNet.tiramisu import FCDenseNet67Base, FCDenseNet67Classifier
from ..owner import EngineOwner, ForwardFn


def getClassWeight(targets, maxClasses: int = None):
    """Reciprocal per-class pixel counts (TrainingBase.py:12-23), kept for API parity.  The training step
    does not call this: the fused loss kernel builds the same weights on the device."""
    elements, counts = torch.unique(targets, sorted=True, return_counts=True)
    if maxClasses:
        assert maxClasses > int(max(elements)), f"Found more label classes than given maxClasses={maxClasses}"
    else:
        maxClasses = int(max(elements)) + 1
    countPerClass = torch.zeros(maxClasses, dtype=torch.float)
    countPerClass[elements.cpu().long()] = counts.cpu().float()
    return torch.reciprocal(countPerClass)


class TrainingBase(LightningModule, EngineOwner):
    def __init__(self, lr=1e-3, decay=1e-4, lrRatio=1e3, num_cls=2):
        super().__init__()
        self.save_hyperparameters('lr', 'decay', 'lrRatio')

        self.featureExtractor = FCDenseNet67Base()
        self.classifier = FCDenseNet67Classifier(n_classes=num_cls)
        self.featureExtractor.__dict__["_rln_parent"] = self

        self.lr = lr
        self.decay = decay
        self.lrRatio = lrRatio
        self.num_cls = num_cls
        # debug switch: check labels < num_cls on the host like the reference's assert (costs a device sync)
        self.check_labels = False

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._rln_mark_dirty()
        return out

    @staticmethod
    def add_model_specific_args(parent_parser):
        parser = ArgumentParser(parents=[parent_parser], add_help=False)
        group = parser.add_argument_group('TrainingModule', 'Parameters defining network training')
        group.add_argument('-lr', '--learningRate', type=float, default=1e-3, help="Starting learning rate")
        group.add_argument('--decay', type=float, default=1e-4, help="L2 weight decay value")
        group.add_argument('--lrRatio', type=float, default=1000,
                           help="Ratio of maximum and minimum of learning rate for cosine LR scheduler")
        return parser

    def _rln_prepare_batch(self, x, y=None, train=False):
        """Device half of the input transform for batches that come from a DataLoader over the reference's datasets: the
        Dataset-side MyTransform call hands over raw uint8 frames (dataManagement/myTransforms.py: deferred form); here
        the whole batch goes through rln_preprocess_u8 / rln_augment_u8.  Float batches pass through untouched."""
        if torch.is_tensor(x) and x.dtype == torch.uint8:
            from ..dataManagement.myTransforms import MyTransform
            return MyTransform.prepare_batch(x, y, train=train, device=self._rln_current_device())
        return x, y

    def forward(self, x):
        """featureExtractor -> classifier as one fused HIP forward (TrainingBase.py:54-57).  In train mode with autograd
        enabled the result carries a grad_fn (HIP backward through the whole net), so a user-written step such as
        ``cross_entropy(self.forward(x), y).backward()`` (SimpleTrain.py:15-16) trains the parameters; in eval mode or
        under ``torch.no_grad()`` it is the plain inference forward."""
        x, _ = self._rln_prepare_batch(x)
        if self.training and torch.is_grad_enabled():
            return ForwardFn.apply(self, x, None, None, *self._rln_params_in_arena_order())
        eng = self._rln_sync()
        with torch.no_grad():
            probs, _ = eng.forward(x, training=self.training, with_backward=False)
        return probs

    # ---- evaluation (TrainingBase.py:59-110) ------------------------------------------------
    def validation_step(self, batch, batch_idx):
        return self.evaluate_batch(batch)

    def validation_epoch_end(self, outputs):
        logs = self.summarize_evaluation_results(outputs)
        self.log('val_loss', logs['loss'])
        self.log('val_acc', logs['acc'], prog_bar=True, logger=True)
        self.log('val_dice', logs['dice'])
        self.log('val_iou', logs['iou'], prog_bar=True, logger=True)

    def test_step(self, batch, batch_idx):
        return self.evaluate_batch(batch)

    def test_epoch_end(self, outputs):
        logs = self.summarize_evaluation_results(outputs)
        self.log('test_loss', logs['loss'])
        self.log('test_acc', logs['acc'])
        self.log('test_dice', logs['dice'])
        self.log('test_iou', logs['iou'])

    def evaluate_batch(self, batch):
        x, y = batch
        x, y = self._rln_prepare_batch(x, y)
        eng = self._rln_sync()
        with torch.no_grad():
            probs, _ = eng.forward(x, training=self.training, with_backward=False)
            out, _, conf = eng.loss(probs, y, weighted=False, want_confusion=True)
        weight = x.shape[0]
        return {
            'loss': out[0] * weight,
            'acc': accuracy_from_confusion(conf) * weight,
            'dice': dice_from_confusion(conf) * weight,
            'iou': iou_from_confusion(conf) * weight,
            'weight': weight,
        }

    @staticmethod
    def summarize_evaluation_results(outputs):
        total_weight = sum(x['weight'] for x in outputs)
        return {
            'loss': sum(x['loss'] for x in outputs) / total_weight,
            'acc': sum(x['acc'] for x in outputs) / total_weight * 100.0,
            'dice': sum(x['dice'] for x in outputs) / total_weight,
            'iou': sum(x['iou'] for x in outputs) / total_weight * 100.0,
        }
