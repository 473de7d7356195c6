"""DETRDecoder: host-side mirror of dino_detector/models/detr_decoder.py (+ the parameter
layout of deformable_attention.py).  Containers only; arithmetic is in libdinodet.so."""
import torch
import torch.nn as nn

from ..config import BackboneConfig, DecoderConfig
from ..engine import default_precision, split_detections
from .dinov2_backbone import _Box, _EngineMixin


class DeformableAttention(_Box):
    """Parameters of deformable_attention.py:8-51 (same init: zero offsets/weights, Xavier projections)."""

    def __init__(self, d_model=256, n_heads=8, n_points=4):
        super().__init__()
        self.d_model, self.n_heads, self.n_points = d_model, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        for m in (self.sampling_offsets, self.attention_weights):
            nn.init.constant_(m.weight, 0.)
            nn.init.constant_(m.bias, 0.)
        for m in (self.value_proj, self.output_proj):
            nn.init.xavier_uniform_(m.weight)
            nn.init.constant_(m.bias, 0.)


class DeformableDecoderLayer(_Box):
    """Parameters of deformable_attention.py:190-213."""

    def __init__(self, d_model=256, n_heads=8, dim_feedforward=2048, dropout=0.1, n_points=4):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout1 = nn.Dropout(dropout)       # the reference's own dropout modules (deformable_attention.py:196-209): callers that
        self.norm1 = nn.LayerNorm(d_model)        # change `.p` or put a sub-module in eval() are honoured by the train() paths
        self.cross_attn = DeformableAttention(d_model, n_heads, n_points)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.activation = nn.ReLU()
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)
        self.reference_points_proj = nn.Linear(d_model, 2)


class DeformableTransformerDecoder(_Box):
    """deformable_attention.py:271-284: the SAME layer object repeated -> weights tied, state dict
    lists it under every index."""

    def __init__(self, decoder_layer, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([decoder_layer for _ in range(num_layers)])


class MLP(_Box):
    """dino_detector/utils.py:14-30 parameter layout (`mlp.{0,2,...}`)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        layers = []
        for i in range(num_layers):
            layers.append(nn.Linear(input_dim if i == 0 else hidden_dim,
                                    output_dim if i == num_layers - 1 else hidden_dim))
            if i < num_layers - 1:
                layers.append(nn.ReLU())
        self.mlp = nn.Sequential(*layers)


class DETRDecoder(_EngineMixin, nn.Module):
    """dino_detector/models/detr_decoder.py:7-83; `precision` is the only extra argument."""

    def __init__(self, num_queries, hidden_dim, nheads, num_decoder_layers, num_classes,
                 dim_feedforward=2048, dropout=0.1, n_points=4, use_deformable=True, precision=None):
        super().__init__()
        self.num_queries = num_queries
        self.use_deformable = use_deformable
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        if use_deformable:
            layer = DeformableDecoderLayer(hidden_dim, nheads, dim_feedforward, dropout, n_points)
            self.decoder = DeformableTransformerDecoder(layer, num_layers=num_decoder_layers)
        else:
            # parameter container with nn.TransformerDecoder's own names/init (never called)
            layer = nn.TransformerDecoderLayer(d_model=hidden_dim, nhead=nheads,
                                               dim_feedforward=dim_feedforward, dropout=dropout)
            self.decoder = nn.TransformerDecoder(layer, num_layers=num_decoder_layers)
        self.class_embed = nn.Linear(hidden_dim, num_classes)
        self.bbox_embed = MLP(hidden_dim, hidden_dim // 2, 4, num_layers=2)
        if use_deformable:
            self.reference_points = nn.Linear(hidden_dim, 2)   # unused by forward (detr_decoder.py:44-45)
        self.precision = precision or default_precision()
        self._dropout_p = float(dropout)
        self._dc_cfg = DecoderConfig(num_queries=num_queries, hidden_dim=hidden_dim, nheads=nheads,
                                     num_layers=num_decoder_layers, num_classes=num_classes,
                                     dim_feedforward=dim_feedforward, n_points=n_points,
                                     use_deformable=use_deformable)
        # a stand-alone decoder handle carries a minimal (unused) backbone description
        self._bb_cfg = BackboneConfig(hidden=hidden_dim, layers=1, heads=max(1, hidden_dim // 64), target_dim=0)

    _key_prefix = "decoder."

    def forward(self, src):
        """src [batch, seq_len, hidden_dim] -> {"pred_logits": [B,Q,C], "pred_boxes": [B,Q,4]}"""
        if self._use_autograd(src):
            import os
            from . import _autograd, _native_train
            # train(): decoder + heads on the native kernels with their hand-written backward -- the deformable branch (the reference's
            # default) and, round 3, the dense nn.TransformerDecoder branch; CPU tensors (test suite only) and shapes the kernels do not
            # take stay on the autograd composite
            if os.environ.get("DINODET_NATIVE_TRAIN", "1") != "0":
                if _native_train.supported(self, src):
                    return split_detections(_native_train.decoder_train(self, src), self._dc_cfg.num_classes)
                if _native_train.dense_supported(self, src):
                    return split_detections(_native_train.dense_decoder_train(self, src), self._dc_cfg.num_classes)
            return _autograd.decoder_forward(self, src)
        det = self._get_engine().decoder_forward(src, self._engine_named())
        return split_detections(det, self._dc_cfg.num_classes)
