"""MI355X-native lane-segmentation hot path (FC-DenseNet67 training / inference).

Python host over a C-ABI HIP library (include/rln.h, csrc/).  Package layout mirrors the part of
``rightLaneNetwork/`` that sits on the path: ``models.FCDenseNet``, ``trainingModules``.
``install_aliases()`` registers those sub-packages under the reference's top-level import names so that
``from trainingModules.SimpleTrain import SimpleTrainModule`` (train.py:10-12, test.py:14-15,
makeDemoVideo.py:8-9) resolves to this implementation.
"""
import sys

__all__ = ["install_aliases"]


def install_aliases():
    from . import models, trainingModules
    from .models import FCDenseNet
    from .models.FCDenseNet import layers, tiramisu
    from .trainingModules import MMETrainingModule, SimpleTrain, TrainingBase
    sys.modules.setdefault("models", models)
    sys.modules.setdefault("models.FCDenseNet", FCDenseNet)
    sys.modules.setdefault("models.FCDenseNet.layers", layers)
    sys.modules.setdefault("models.FCDenseNet.tiramisu", tiramisu)
    sys.modules.setdefault("trainingModules", trainingModules)
    sys.modules.setdefault("trainingModules.TrainingBase", TrainingBase)
    sys.modules.setdefault("trainingModules.SimpleTrain", SimpleTrain)
    sys.modules.setdefault("trainingModules.MMETrainingModule", MMETrainingModule)
    from . import dataManagement
    from .dataManagement import myTransforms
    sys.modules.setdefault("dataManagement", dataManagement)
    sys.modules.setdefault("dataManagement.myTransforms", myTransforms)
