"""Importable alias of the package directory ``image-classification_amd/`` (a hyphen is not a valid
Python identifier).  ``import image_classification_amd as ica`` returns the real package."""
import importlib
import sys

_pkg = importlib.import_module("image-classification_amd")
for _name, _mod in list(sys.modules.items()):
    if _name.startswith("image-classification_amd."):
        sys.modules[__name__ + _name[len("image-classification_amd"):]] = _mod
sys.modules[__name__] = _pkg
