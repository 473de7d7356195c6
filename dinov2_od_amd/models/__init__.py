"""Same export list as dino_detector/models/__init__.py:2-6."""
from .detector import DINOv2ObjectDetector
from .dinov2_backbone import DINOv2Backbone
from .detr_decoder import DETRDecoder

__all__ = ["DINOv2ObjectDetector", "DINOv2Backbone", "DETRDecoder"]
