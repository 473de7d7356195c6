"""Shape description of the detector forward path.

Mirrors the constructor defaults the reference supplies from
``dino_detector/config.py:21-35`` and the backbone name -> width table at
``dino_detector/models/dinov2_backbone.py:17-27`` / ``detector.py:25-35``.
Depth / head counts of the four DINOv2 variants are the published hub configs
(SURVEY.md section 8a); they can be overridden for micro test models.
"""
from dataclasses import dataclass, asdict
import math

# reference: dino_detector/config.py:21-35
REF_DEFAULTS = dict(
    num_classes=91,
    dino_model_name="facebook/dinov2-base",
    lora_r=2,
    lora_alpha=1.0,
    hidden_dim=768,
    num_queries=50,
    nheads=8,
    num_decoder_layers=3,
    dim_feedforward=1024,
    dropout=0.1,
    n_points=2,
    use_deformable=True,
)

# variant -> (hidden, layers, heads, swiglu)
BACKBONE_VARIANTS = {
    "small": (384, 12, 6, False),
    "base": (768, 12, 12, False),
    "large": (1024, 24, 16, False),
    "giant": (1536, 40, 24, True),
}


def variant_of(model_name: str) -> str:
    """Same substring tests, in the same order, as dinov2_backbone.py:17-27."""
    v = model_name.split("/")[-1]
    for key in ("small", "base", "large", "giant"):
        if key in v:
            return key
    return "base"


@dataclass
class BackboneConfig:
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    swiglu: bool = False
    patch: int = 14
    pos_grid: int = 37          # pretrained grid (image_size 518 / 14)
    mlp_ratio: int = 4
    ln_eps: float = 1e-6
    lora_r: int = 2
    lora_alpha: float = 1.0
    lora_layers: int = 2        # dinov2_backbone.py:47-51: last two blocks
    target_dim: int = 0         # 0 = no projection (dinov2_backbone.py:33-37)

    @property
    def head_dim(self):
        return self.hidden // self.heads

    @property
    def ffn_hidden(self):
        h = int(self.hidden * self.mlp_ratio)
        if self.swiglu:  # modeling_dinov2.py:303-305
            h = (int(h * 2 / 3) + 7) // 8 * 8
        return h

    @property
    def out_dim(self):
        return self.target_dim if self.target_dim else self.hidden

    @classmethod
    def from_name(cls, model_name, lora_r=4, lora_alpha=1.0, target_dim=None):
        hidden, layers, heads, swiglu = BACKBONE_VARIANTS[variant_of(model_name)]
        td = 0 if (target_dim is None or target_dim == hidden) else int(target_dim)
        return cls(hidden=hidden, layers=layers, heads=heads, swiglu=swiglu,
                   lora_r=lora_r, lora_alpha=lora_alpha, target_dim=td)


@dataclass
class DecoderConfig:
    num_queries: int = 50
    hidden_dim: int = 768
    nheads: int = 8
    num_layers: int = 3
    num_classes: int = 91
    dim_feedforward: int = 1024
    n_points: int = 2
    use_deformable: bool = True
    ln_eps: float = 1e-5        # nn.LayerNorm default (deformable_attention.py:197)

    @property
    def head_dim(self):
        return self.hidden_dim // self.nheads


def spatial_factor(hw: int):
    """(h, w) choice of DeformableDecoderLayer.forward,
    dino_detector/models/deformable_attention.py:241-256 (token count includes CLS)."""
    s = int(hw ** 0.5)
    if s * s != hw:
        for i in range(s, 0, -1):
            if hw % i == 0:
                return i, hw // i
    return s, s


def num_tokens(H, W, patch=14):
    return (H // patch) * (W // patch) + 1


def flops_per_image(bb: BackboneConfig, dec: DecoderConfig, H: int, W: int) -> float:
    """Algorithmic FLOPs (2*MACs) of one forward, SURVEY.md section 8d formulae."""
    D, L = bb.hidden, bb.layers
    Np = (H // bb.patch) * (W // bb.patch)
    N = Np + 1
    f = 2.0 * Np * (3 * bb.patch * bb.patch) * D
    if bb.swiglu:
        mlp = 2.0 * N * D * 2 * bb.ffn_hidden + 2.0 * N * bb.ffn_hidden * D
    else:
        mlp = 2.0 * 2.0 * N * D * 4 * D
    f += L * (2.0 * N * D * 4 * D + mlp + 4.0 * N * N * D)
    Dd = dec.hidden_dim
    if bb.target_dim:
        f += 2.0 * N * D * Dd
    Q, Hd, P, F, C = dec.num_queries, dec.nheads, dec.n_points, dec.dim_feedforward, dec.num_classes
    self_attn = 2.0 * Q * Dd * 3 * Dd + 4.0 * Q * Q * Dd + 2.0 * Q * Dd * Dd
    if dec.use_deformable:
        cross = (2.0 * N * Dd * Dd + 2.0 * Q * Dd * 3 * Hd * P + 2.0 * Q * Dd * 2
                 + 8.0 * Q * Hd * P * (Dd // Hd) + 2.0 * Q * Dd * Dd)
    else:
        cross = 2.0 * Q * Dd * Dd + 2.0 * N * Dd * 2 * Dd + 4.0 * Q * N * Dd + 2.0 * Q * Dd * Dd
    ffn = 4.0 * Q * Dd * F
    f += dec.num_layers * (self_attn + cross + ffn)
    f += 2.0 * Q * Dd * (C + Dd // 2) + 8.0 * Q * (Dd // 2)
    return f


def as_dict(cfg):
    return asdict(cfg)
