"""Shared model shapes for the parity tests (same shapes tests/golden/make_goldens.py used)."""
import os
import numpy as np

from dinov2_od_amd.config import BackboneConfig, DecoderConfig
from dinov2_od_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def micro_bb(swiglu=False):
    return BackboneConfig(hidden=128, layers=2, heads=2, swiglu=swiglu, pos_grid=5, lora_r=2,
                          lora_alpha=1.0, target_dim=0)


def dec_cfg(deform, Dd=128, Hd=4, Q=7, layers=2, F=256, C=11, P=2):
    return DecoderConfig(num_queries=Q, hidden_dim=Dd, nheads=Hd, num_layers=layers, num_classes=C,
                         dim_feedforward=F, n_points=P, use_deformable=deform)


G1_CASES = [  # (tag, deform, Dd, Hd, Q, N list)
    ("d32", True, 128, 4, 7, (17, 26, 257, 1370)),
    ("d96", True, 192, 2, 5, (26, 1370)),
    ("s32", False, 128, 4, 7, (17, 257)),
    ("s96", False, 192, 2, 5, (26,)),
]


def g1_memory(N, Dd, seed=3):
    return synth.normal(seed, f"memory.{N}.{Dd}", (2, N, Dd), 1.0)


def cfg1(Q):
    """BASELINE.json configs[0]: --lightweight ViT-S/14 (reference train.py:607-640)."""
    bb = BackboneConfig.from_name("facebook/dinov2-small", lora_r=1, lora_alpha=1.0, target_dim=256)
    dc = DecoderConfig(num_queries=Q, hidden_dim=256, nheads=4, num_layers=2, num_classes=91,
                       dim_feedforward=512, n_points=2, use_deformable=True)
    return bb, dc


def vitb(Q=100, deform=True):
    bb = BackboneConfig.from_name("facebook/dinov2-base", lora_r=2, lora_alpha=1.0, target_dim=768)
    dc = DecoderConfig(num_queries=Q, hidden_dim=768, nheads=8, num_layers=3, num_classes=91,
                       dim_feedforward=1024, n_points=2, use_deformable=deform)
    return bb, dc


def rel_err(a, b):
    """max|a-b| / max|b|  -- the 'relative fp32 tolerance' metric used by every parity test."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


def vit_full(variant, Q=300):
    """BASELINE.json configs[3] / configs[4] at full depth: ViT-L/14 (24 blocks) and ViT-g/14 (40 blocks, SwiGLU) with the
    768-wide projection (dinov2_backbone.py:33-37) and a 300-query decoder (goldens g7_vitl_518 / g8_vitg_518)."""
    bb = BackboneConfig.from_name(f"facebook/dinov2-{variant}", lora_r=2, lora_alpha=1.0, target_dim=768)
    dc = DecoderConfig(num_queries=Q, hidden_dim=768, nheads=8, num_layers=3, num_classes=91,
                       dim_feedforward=1024, n_points=2, use_deformable=True)
    return bb, dc


FULL_DEPTH = {"large": ("g7_vitl_518", (0, 11, 23)), "giant": ("g8_vitg_518", (0, 19, 39))}


def probe(a):
    """the probe slice make_goldens._probe stores for a [B, N, D] activation"""
    a = np.asarray(a)
    return a[:, ::max(1, a.shape[1] // 8), :64]
