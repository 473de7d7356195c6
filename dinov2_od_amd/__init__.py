"""MI355X-native forward path of dino_detector (DINOv2 ViT backbone + DETR-style head).

`from dinov2_od_amd import DINOv2ObjectDetector` mirrors `from dino_detector import DINOv2ObjectDetector`
(dino_detector/__init__.py:2).  Importing the package does not load the HIP library; the first
forward does, and fails loudly if it is missing.
"""
__version__ = "0.1.0"


def __getattr__(name):
    if name in ("DINOv2ObjectDetector", "DINOv2Backbone", "DETRDecoder"):
        from . import models
        return getattr(models, name)
    raise AttributeError(name)
