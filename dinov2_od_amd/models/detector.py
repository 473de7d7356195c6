"""DINOv2ObjectDetector: host-side mirror of dino_detector/models/detector.py:8-69."""
import torch
import torch.nn as nn

from ..config import REF_DEFAULTS
from ..engine import default_precision, split_detections
from .dinov2_backbone import DINOv2Backbone, _EngineMixin
from .detr_decoder import DETRDecoder


class DINOv2ObjectDetector(_EngineMixin, nn.Module):   # state-dict keys already carry "backbone." / "decoder."
    """Same constructor defaults as the reference (config.py:21-35 via detector.py:9-21).
    Extra keyword arguments: `pretrained`, `precision`, `backbone_config` (micro test models)."""

    def __init__(self,
                 num_classes=REF_DEFAULTS["num_classes"],
                 dino_model_name=REF_DEFAULTS["dino_model_name"],
                 lora_r=REF_DEFAULTS["lora_r"],
                 lora_alpha=REF_DEFAULTS["lora_alpha"],
                 hidden_dim=REF_DEFAULTS["hidden_dim"],
                 num_queries=REF_DEFAULTS["num_queries"],
                 nheads=REF_DEFAULTS["nheads"],
                 num_decoder_layers=REF_DEFAULTS["num_decoder_layers"],
                 dim_feedforward=REF_DEFAULTS["dim_feedforward"],
                 dropout=REF_DEFAULTS["dropout"],
                 n_points=REF_DEFAULTS["n_points"],
                 use_deformable=REF_DEFAULTS["use_deformable"],
                 pretrained=True, precision=None, backbone_config=None):
        super().__init__()
        if hidden_dim is None:                                  # detector.py:25-35
            hidden_dim = 768
            for key, dim in (("small", 384), ("base", 768), ("large", 1024), ("giant", 1536)):
                if key in dino_model_name:
                    hidden_dim = dim
                    break
        precision = precision or default_precision()
        self.backbone = DINOv2Backbone(model_name=dino_model_name, lora_r=lora_r, lora_alpha=lora_alpha,
                                       target_dim=hidden_dim, pretrained=pretrained, precision=precision,
                                       config=backbone_config)
        self.decoder = DETRDecoder(num_queries=num_queries, hidden_dim=hidden_dim, nheads=nheads,
                                   num_decoder_layers=num_decoder_layers, num_classes=num_classes,
                                   dim_feedforward=dim_feedforward, dropout=dropout, n_points=n_points,
                                   use_deformable=use_deformable, precision=precision)
        self.precision = precision
        self._dropout_p = float(dropout)
        self._bb_cfg = self.backbone._bb_cfg
        self._dc_cfg = self.decoder._dc_cfg

    def enable_hipgraph(self, on=True):
        """eval-mode forwards replay one captured hipGraph per input shape (the returned tensors are the graph's static output
        buffer: consume or clone them before the next forward of the same shape)"""
        self._get_engine().use_graph = bool(on)
        return self

    def forward_packed(self, pixel_values):
        """[B,3,H,W] -> packed detections [B, Q, C+4] (logits | boxes): the buffer the multi-GPU
        all-gather moves (dinov2_od_amd.dist.gather_detections)."""
        if self._use_autograd(pixel_values):
            o = self.forward(pixel_values)
            return torch.cat([o["pred_logits"], o["pred_boxes"]], dim=-1)
        return self._get_engine().forward(pixel_values, self._engine_named())

    def forward_packed_u8(self, pixels_hwc):
        """eval-mode forward straight from the device input pipeline's bytes: uint8 [B, H, W, 3] as
        `preprocess_batch(images, size, as_uint8=True)` returns them (Resize done, ToTensor not yet: train.py:584-587) -> packed
        detections.  Identical, bit for bit, to forward_packed(preprocess_batch(images, size)); bf16 / bf16x3 / fp8 precision."""
        if self.training:
            raise RuntimeError("forward_packed_u8 is an eval-mode entry (train() takes the fp32 batch)")
        return self._get_engine().forward_u8(pixels_hwc, self._engine_named())

    def forward(self, pixel_values):
        """pixel_values [batch, 3, H, W] -> {"pred_logits", "pred_boxes"} (detector.py:58-69)."""
        if self._use_autograd(pixel_values):            # train(): autograd composite of the same math (detector.py:58-69)
            return self.decoder(self.backbone(pixel_values))
        return split_detections(self.forward_packed(pixel_values), self._dc_cfg.num_classes)
