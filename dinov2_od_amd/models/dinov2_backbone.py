"""DINOv2Backbone: host-side mirror of dino_detector/models/dinov2_backbone.py.

Same constructor arguments, `forward(pixel_values)` signature, attribute names and
state-dict keys as the reference, so checkpoints and the callers in
train.py / utils.py work unchanged -- but the modules below are PARAMETER CONTAINERS: their
arithmetic runs in hand-written gfx950 kernels behind libdinodet.so (dinov2_od_amd/csrc).
The HF `Dinov2Model` the reference wraps (dinov2_backbone.py:11) is mirrored structurally
(embeddings / encoder.layer[i] / layernorm) without importing `transformers`.
"""
import os

import torch
import torch.nn as nn

from ..config import BackboneConfig, DecoderConfig, variant_of
from ..engine import Engine, default_precision


class _Box(nn.Module):
    """plain container; never called"""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("container module: arithmetic runs in libdinodet.so")


class LoraLinear(_Box):
    """Parameter layout of dino_detector/utils.py:46-70 (linear / lora_A / lora_B).
    Applied as W' = W + alpha * B A at weight-pack time (exact in fp32)."""

    def __init__(self, linear_layer: nn.Linear, r=4, alpha=1.0):
        super().__init__()
        self.linear = linear_layer
        self.in_features, self.out_features = linear_layer.in_features, linear_layer.out_features
        self.r, self.alpha = r, alpha
        self.lora_A = nn.Linear(self.in_features, r, bias=False)
        self.lora_B = nn.Linear(r, self.out_features, bias=False)
        nn.init.zeros_(self.lora_B.weight)      # utils.py:61
        for p in self.linear.parameters():
            p.requires_grad = False


def add_lora_to_module(module, r=4, alpha=1.0):
    """Same traversal as dino_detector/utils.py:33-43: every nn.Linear below `module`."""
    for name, child in list(module.named_children()):
        if isinstance(child, LoraLinear):
            continue
        add_lora_to_module(child, r=r, alpha=alpha)
        if isinstance(child, nn.Linear):
            setattr(module, name, LoraLinear(child, r=r, alpha=alpha))


def _dino_container(bb: BackboneConfig):
    """Module tree with HF Dinov2Model's parameter names (modeling_dinov2.py:43-56, 342-359, 441)."""
    D = bb.hidden
    dino = _Box()
    emb = _Box()
    emb.cls_token = nn.Parameter(torch.randn(1, 1, D))
    emb.mask_token = nn.Parameter(torch.zeros(1, D))
    emb.position_embeddings = nn.Parameter(torch.randn(1, bb.pos_grid * bb.pos_grid + 1, D))
    emb.patch_embeddings = _Box()
    emb.patch_embeddings.projection = nn.Conv2d(3, D, kernel_size=bb.patch, stride=bb.patch)
    dino.embeddings = emb
    enc = _Box()
    layers = []
    for _ in range(bb.layers):
        L = _Box()
        L.norm1 = nn.LayerNorm(D, eps=bb.ln_eps)
        L.attention = _Box()
        L.attention.attention = _Box()
        L.attention.attention.query = nn.Linear(D, D)
        L.attention.attention.key = nn.Linear(D, D)
        L.attention.attention.value = nn.Linear(D, D)
        L.attention.output = _Box()
        L.attention.output.dense = nn.Linear(D, D)
        L.layer_scale1 = _Box()
        L.layer_scale1.lambda1 = nn.Parameter(torch.ones(D))
        L.norm2 = nn.LayerNorm(D, eps=bb.ln_eps)
        L.mlp = _Box()
        if bb.swiglu:
            L.mlp.weights_in = nn.Linear(D, 2 * bb.ffn_hidden)
            L.mlp.weights_out = nn.Linear(bb.ffn_hidden, D)
        else:
            L.mlp.fc1 = nn.Linear(D, bb.ffn_hidden)
            L.mlp.fc2 = nn.Linear(bb.ffn_hidden, D)
        L.layer_scale2 = _Box()
        L.layer_scale2.lambda1 = nn.Parameter(torch.ones(D))
        layers.append(L)
    enc.layer = nn.ModuleList(layers)
    dino.encoder = enc
    dino.layernorm = nn.LayerNorm(D, eps=bb.ln_eps)
    # HF init (modeling_dinov2.py:399-414): trunc_normal(0.02) weights, zero biases
    for m in dino.modules():
        if isinstance(m, (nn.Linear, nn.Conv2d)):
            nn.init.trunc_normal_(m.weight, mean=0.0, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
    nn.init.trunc_normal_(emb.position_embeddings, mean=0.0, std=0.02)
    nn.init.trunc_normal_(emb.cls_token, mean=0.0, std=0.02)
    return dino


def _load_pretrained(dino, model_name):
    try:
        from transformers import Dinov2Model
        ref = Dinov2Model.from_pretrained(model_name)
    except Exception as e:  # no network / no cache / no transformers
        raise RuntimeError(
            f"could not load pretrained weights for {model_name!r} ({type(e).__name__}: {e}). "
            "Pass pretrained=False to build the architecture with random init and load a "
            "state dict afterwards.") from e
    missing, unexpected = dino.load_state_dict(ref.state_dict(), strict=False)
    if missing:
        raise RuntimeError(f"pretrained checkpoint lacks keys: {missing[:5]}...")


class _EngineMixin:
    """Lazily creates the native engine and keeps its packed weights in sync with the module."""

    _key_prefix = ""

    def _engine_named(self):
        """list of (reference state-dict key, tensor).  Built once (a state_dict() walk of 300+ entries per forward costs
        more than a small-batch forward does), dropped by _apply (.to / .cuda / .float), load_state_dict and
        invalidate_weight_cache(), and rebuilt whenever a Parameter object of the tree was replaced.  In-place
        updates (optimizer steps, load_state_dict's copy_) keep the list valid and are seen by the engine's
        (data_ptr, _version) signature."""
        # the cache is validated against the identity of the module tree's current Parameter objects (an id() walk: ~20x cheaper
        # than state_dict()): add_lora_to_module / setattr of a block / register_parameter / load_state_dict(assign=True) on a child
        # replace objects without passing through the hooks below, and the (data_ptr, _version) signature of the OLD tensors would
        # still match
        ids = tuple(id(p) for p in self.parameters())
        ent = self.__dict__.get("_named_cache")
        if ent is None or ent[0] != ids:
            ent = (ids, [(self._key_prefix + k, v) for k, v in self.state_dict(keep_vars=True).items()])
            self.__dict__["_named_cache"] = ent
        return ent[1]

    def invalidate_weight_cache(self):
        self.__dict__.pop("_named_cache", None)
        for m in self.children():
            if isinstance(m, _EngineMixin):
                m.invalidate_weight_cache()

    def _apply(self, fn, *a, **k):
        self.invalidate_weight_cache()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.invalidate_weight_cache()
        return super().load_state_dict(*a, **k)

    def _get_engine(self):
        eng = self.__dict__.get("_engine")
        if eng is None:
            eng = Engine(self._bb_cfg, self._dc_cfg, self.precision)
            self.__dict__["_engine"] = eng
        return eng

    def _use_autograd(self, x=None):
        """train() mode -> the autograd composite (models/_autograd.py): the native kernels have no backward yet
        (SURVEY.md section 8f-1).  eval() mode is always native.  Like the native path the composite is GPU-only: a CPU
        tensor raises (DINODET_COMPOSITE_ON_CPU=1 lifts that for the CPU test-suite, which pins the composite against the
        reference goldens)."""
        if self.training and x is not None and not x.is_cuda and os.environ.get("DINODET_COMPOSITE_ON_CPU") != "1":
            raise RuntimeError("inputs must be on the GPU: the MI355X path has no CPU fallback (train() mode included)")
        return self.training

    def set_precision(self, precision):
        self.precision = precision
        self.__dict__.pop("_engine", None)


class DINOv2Backbone(_EngineMixin, nn.Module):
    """dino_detector/models/dinov2_backbone.py:7-67.  Extra keyword arguments (not in the
    reference): `pretrained` (default True, like from_pretrained) and `precision`
    ("bf16" fast path | "fp32" strict parity)."""

    def __init__(self, model_name="facebook/dinov2-base", lora_r=4, lora_alpha=1.0, target_dim=None,
                 pretrained=True, precision=None, config: BackboneConfig = None):
        super().__init__()
        self.model_variant = model_name.split('/')[-1]
        bb = config if config is not None else BackboneConfig.from_name(model_name, lora_r, lora_alpha, target_dim)
        if config is not None:
            bb.lora_r, bb.lora_alpha = lora_r, lora_alpha
            bb.target_dim = 0 if (target_dim is None or target_dim == bb.hidden) else int(target_dim)
        self.hidden_dim = bb.hidden
        self.target_dim = target_dim
        self.dino = _dino_container(bb)
        if pretrained:
            _load_pretrained(self.dino, model_name)
        self.projection = nn.Linear(bb.hidden, bb.target_dim) if bb.target_dim else None
        for p in self.dino.parameters():                     # dinov2_backbone.py:40-41
            p.requires_grad = False
        n = len(self.dino.encoder.layer)                     # dinov2_backbone.py:47-51
        if lora_r and lora_r > 0:
            for i in range(n - min(2, n), n):
                add_lora_to_module(self.dino.encoder.layer[i], r=lora_r, alpha=lora_alpha)
        self.precision = precision or default_precision()
        self._bb_cfg = bb
        # a stand-alone backbone handle carries a minimal (unused) decoder description
        self._dc_cfg = DecoderConfig(num_queries=1, hidden_dim=bb.out_dim, nheads=1 if bb.out_dim <= 128 else bb.out_dim // 64,
                                     num_layers=1, num_classes=1, dim_feedforward=64)

    _key_prefix = "backbone."

    def forward(self, pixel_values):
        """-> features [batch, seq_len, hidden_dim] fp32, CLS token at index 0 (dinov2_backbone.py:58-67)"""
        if self._use_autograd(pixel_values):
            from . import _autograd
            return _autograd.backbone_forward(self, pixel_values)
        return self._get_engine().backbone_forward(pixel_values, self._engine_named())
