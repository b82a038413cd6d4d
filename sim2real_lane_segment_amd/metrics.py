"""Segmentation metrics the reference takes from pytorch_lightning 1.2.1 ``metrics.functional``
(TrainingBase.py:5,91-93; test.py:10,97-100).  That package is absent here, so these follow its
published definitions (parity unpinned) and are computed from the confusion
matrix that the fused HIP loss kernel already produces (row = label, column = prediction)."""
import torch


def confusion_matrix(pred, target, num_classes):
    idx = target.reshape(-1).long() * num_classes + pred.reshape(-1).long()
    return torch.bincount(idx, minlength=num_classes * num_classes).reshape(num_classes, num_classes)


def accuracy_from_confusion(cm):
    cm = cm.double()
    return (torch.diag(cm).sum() / cm.sum().clamp(min=1)).float()


def iou_from_confusion(cm, absent_score=0.0):
    """mean over classes [0, max class present in pred or target] of TP/(TP+FP+FN); classes absent from both
    score ``absent_score`` (pl 1.2.1 ``iou`` with num_classes inferred per call)."""
    cm = cm.double()
    inter = torch.diag(cm)
    union = cm.sum(0) + cm.sum(1) - inter
    present = (cm.sum(0) + cm.sum(1)) > 0
    k = cm.shape[0]
    ar = torch.arange(k, device=cm.device)
    n_used = torch.where(present, ar + 1, torch.zeros_like(ar)).max().clamp(min=1)
    scores = torch.where(union > 0, inter / union.clamp(min=1), torch.full_like(inter, absent_score))
    scores = torch.where(ar < n_used, scores, torch.zeros_like(scores))
    return (scores.sum() / n_used).float()


def dice_from_confusion(cm, bg=False, nan_score=0.0, no_fg_score=0.0):
    """pl 1.2.1 ``dice_score``: mean over classes (1.. unless bg) of 2TP/(2TP+FP+FN) on argmax; classes absent
    from the target score ``no_fg_score``."""
    cm = cm.double()
    tp = torch.diag(cm)
    fp = cm.sum(0) - tp
    fn = cm.sum(1) - tp
    denom = 2 * tp + fp + fn
    score = torch.where(denom > 0, 2 * tp / denom.clamp(min=1), torch.full_like(tp, nan_score))
    score = torch.where(cm.sum(1) > 0, score, torch.full_like(tp, no_fg_score))
    start = 0 if bg else 1
    return score[start:].mean().float()


def accuracy(pred, target):
    return (pred == target).float().mean()


def iou(pred, target, num_classes=None):
    k = int(max(pred.max(), target.max())) + 1 if num_classes is None else num_classes
    return iou_from_confusion(confusion_matrix(pred, target, k))


def dice_score(probs, target, bg=False):
    return dice_from_confusion(confusion_matrix(probs.argmax(1), target, probs.shape[1]), bg=bg)
