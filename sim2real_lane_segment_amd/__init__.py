"""MI355X-native lane-segmentation hot path (FC-DenseNet67 training / inference).

Python host over a C-ABI HIP library (include/rln.h, csrc/).  Package layout mirrors the part of
``rightLaneNetwork/`` that sits on the path: ``models``, ``trainingModules`` and the input transform
``dataManagement.myTransforms``.  ``install_aliases()`` registers those under the reference's top-level import
names so that ``from trainingModules.SimpleTrain import SimpleTrainModule`` (train.py:10-12, test.py:14-15,
makeDemoVideo.py:8-9) resolves to this implementation while everything of the reference that is NOT on the path
(``dataManagement.dataModules``, ``dataManagement.myDatasets``, ...) keeps resolving to the reference's files.
"""
import importlib
import importlib.util
import sys

__all__ = ["install_aliases"]

# packages of the reference that this repository replaces as a whole (every module of theirs has a mirror here)
_WHOLE = ("models", "models.FCDenseNet", "models.FCDenseNet.layers", "models.FCDenseNet.tiramisu",
          "models.EncDecNet", "trainingModules", "trainingModules.TrainingBase", "trainingModules.SimpleTrain",
          "trainingModules.MMETrainingModule")


def install_aliases():
    """Makes the reference's import paths resolve to the HIP path.

    * ``models.*`` and ``trainingModules.*`` are replaced as whole packages.  Every sub-module is registered
      explicitly: a lazy import through the aliased package's ``__path__`` would create a second module object named
      ``models.X`` whose relative imports (``from .. import _lib``) point outside the package.
    * ``dataManagement`` is NOT replaced when the reference's package is importable (``rightLaneNetwork/`` on
      ``sys.path``): only the leaf ``dataManagement.myTransforms`` is swapped, so ``dataManagement.dataModules`` and
      ``dataManagement.myDatasets`` (train.py:9, test.py:15) stay the reference's and their
      ``from .myTransforms import MyTransform`` picks up the device transform.  Without the reference on the path the
      name ``dataManagement`` maps to this repository's package (which only has ``myTransforms``).
    """
    pkg = __name__
    for alias in _WHOLE:
        sys.modules[alias] = importlib.import_module(f"{pkg}.{alias}")
    # parent packages expose their children as attributes (``import models.EncDecNet; models.EncDecNet.EncDecNet``)
    for alias in _WHOLE:
        parent, _, leaf = alias.rpartition(".")
        if parent:
            setattr(sys.modules[parent], leaf, sys.modules[alias])

    ours = importlib.import_module(f"{pkg}.dataManagement")
    my_transforms = importlib.import_module(f"{pkg}.dataManagement.myTransforms")
    host = sys.modules.get("dataManagement")
    if host is None:
        spec = None
        try:
            spec = importlib.util.find_spec("dataManagement")
        except (ImportError, ValueError):
            spec = None
        if spec is not None and spec.origin != getattr(ours, "__file__", None):
            host = importlib.import_module("dataManagement")  # the reference's package (its __init__ is empty)
        else:
            host = ours
            sys.modules["dataManagement"] = ours
    sys.modules["dataManagement.myTransforms"] = my_transforms
    setattr(host, "myTransforms", my_transforms)
