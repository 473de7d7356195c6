"""Deterministic synthetic weights and inputs.

A counter-based generator (splitmix64 of seed, a CRC of the tensor's state-dict
key and the element index) so the same tensors are reproduced bit-for-bit in the
authoring container (golden generation from the reference) and on the GPU box
(bench, parity tests) without shipping weights and without depending on any
library's RNG stream.  Distribution choices follow SURVEY.md section 8d, with
the zero-initialised parameters of the reference (lora_B, sampling_offsets,
attention_weights: dino_detector/utils.py:61, deformable_attention.py:38-45)
deliberately made non-zero so that every term of the forward is exercised.

State-dict key names/shapes are those of the reference model
(SURVEY.md section 8b "State-dict layout").
"""
import os
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .config import BackboneConfig, DecoderConfig

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _stream(seed: int, key: str, n: int, lane: int, start: int = 0):
    base = np.uint64((seed * 0x100000001B3 + zlib.crc32(key.encode()) * 2 + lane) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([base], dtype=np.uint64))[0]
        idx = np.arange(start, start + n, dtype=np.uint64)
        h = _splitmix64(idx * np.uint64(0x2545F4914F6CDD1D) + base)
    return h


# Large tensors (a ViT-g state dict is 1.15 G values) are generated in 256 Ki-element chunks on a thread pool: every value is a
# function of its own index only, numpy's element loops release the GIL, and chunk boundaries are multiples of any vector width
# (so each element takes the same code path as in a single pass: checked bit for bit in tests/test_oracle_golden.py).
_CHUNK = 1 << 18
_POOL = None


def _drop_pool():
    global _POOL
    _POOL = None            # a forked child has the object but not its threads


if hasattr(os, "register_at_fork"):
    os.register_at_fork(after_in_child=_drop_pool)


def _chunked(fn, n):
    """fn(start, count) -> float32[count]; the concatenation over [0, n)"""
    if n <= 2 * _CHUNK:
        return fn(0, n)
    global _POOL
    if _POOL is None:
        try:
            workers = len(os.sched_getaffinity(0))
        except AttributeError:
            workers = os.cpu_count() or 1
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(16, workers)))
    out = np.empty(n, dtype=np.float32)

    def run(s):
        c = min(_CHUNK, n - s)
        out[s:s + c] = fn(s, c)
    list(_POOL.map(run, range(0, n, _CHUNK)))
    return out


def uniform01(seed, key, shape):
    """float32 in [0, 1): top 24 bits of the hash."""
    n = int(np.prod(shape))
    return _chunked(lambda s, c: (_stream(seed, key, c, 0, s) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24), n).reshape(shape)


def normal(seed, key, shape, std=1.0, mean=0.0):
    """Box-Muller in float64 from two hash streams, returned as float32."""
    n = int(np.prod(shape))

    def part(s, c):
        u1 = ((_stream(seed, key, c, 0, s) >> np.uint64(11)).astype(np.float64) + 1.0) * (2.0 ** -53)
        u2 = (_stream(seed, key, c, 1, s) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)
        z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
        return (mean + std * z).astype(np.float32)
    return _chunked(part, n).reshape(shape)


def make_pixels(B, H, W, seed=0):
    """[B,3,H,W] fp32 uniform [0,1) (ToTensor range, reference train.py:584-587).
    Generated per image so image b is identical whatever the batch size."""
    out = np.empty((B, 3, H, W), dtype=np.float32)
    for b in range(B):
        out[b] = uniform01(seed, f"pixel_values.{H}x{W}.{b}", (3, H, W))
    return out


# --------------------------------------------------------------------------- weights
W_STD = 0.02       # HF initializer_range (modeling_dinov2.py:404-407)
QK_STD = 0.06      # query/key projections: makes the softmax moderately peaked
B_STD = 0.02       # biases (HF zero-inits them; non-zero here on purpose)
DEC_STD = 0.03     # decoder linears (xavier-like scale for 768->768)


def _ln(sd, seed, prefix, dim):
    sd[prefix + ".weight"] = 1.0 + normal(seed, prefix + ".weight", (dim,), 0.1)
    sd[prefix + ".bias"] = normal(seed, prefix + ".bias", (dim,), 0.1)


def _lin(sd, seed, prefix, out_f, in_f, std, bias=True, bias_std=B_STD):
    sd[prefix + ".weight"] = normal(seed, prefix + ".weight", (out_f, in_f), std)
    if bias:
        sd[prefix + ".bias"] = normal(seed, prefix + ".bias", (out_f,), bias_std)


def _reprefix(sd, canonical, prefix):
    if prefix == canonical:
        return sd
    return {prefix + k[len(canonical):]: v for k, v in sd.items()}


def backbone_state_dict(bb: BackboneConfig, seed=1, prefix="backbone."):
    """Keys exactly as the reference's DINOv2Backbone exposes them
    (HF Dinov2Model under `dino.`, LoRA wrappers on the last `lora_layers` blocks).
    Values are keyed on the canonical detector-level name ("backbone...."), so they
    do not depend on `prefix`."""
    return _reprefix(_backbone_state_dict(bb, seed, "backbone."), "backbone.", prefix)


def _backbone_state_dict(bb, seed, prefix):
    sd = {}
    D = bb.hidden
    e = prefix + "dino.embeddings."
    sd[e + "cls_token"] = normal(seed, e + "cls_token", (1, 1, D), W_STD)
    sd[e + "mask_token"] = np.zeros((1, D), dtype=np.float32)
    sd[e + "position_embeddings"] = normal(seed, e + "position_embeddings",
                                           (1, bb.pos_grid * bb.pos_grid + 1, D), W_STD)
    sd[e + "patch_embeddings.projection.weight"] = normal(
        seed, e + "patch_embeddings.projection.weight", (D, 3, bb.patch, bb.patch), W_STD)
    sd[e + "patch_embeddings.projection.bias"] = normal(
        seed, e + "patch_embeddings.projection.bias", (D,), B_STD)
    for i in range(bb.layers):
        lp = f"{prefix}dino.encoder.layer.{i}."
        lora = i >= bb.layers - min(bb.lora_layers, bb.layers) and bb.lora_r > 0

        def lin(name, out_f, in_f, std):
            if lora:
                _lin(sd, seed, lp + name + ".linear", out_f, in_f, std)
                sd[lp + name + ".lora_A.weight"] = normal(seed, lp + name + ".lora_A.weight",
                                                          (bb.lora_r, in_f), 0.05)
                sd[lp + name + ".lora_B.weight"] = normal(seed, lp + name + ".lora_B.weight",
                                                          (out_f, bb.lora_r), 0.05)
            else:
                _lin(sd, seed, lp + name, out_f, in_f, std)

        _ln(sd, seed, lp + "norm1", D)
        lin("attention.attention.query", D, D, QK_STD)
        lin("attention.attention.key", D, D, QK_STD)
        lin("attention.attention.value", D, D, W_STD)
        lin("attention.output.dense", D, D, W_STD)
        sd[lp + "layer_scale1.lambda1"] = 1.0 + normal(seed, lp + "layer_scale1.lambda1", (D,), 0.1)
        _ln(sd, seed, lp + "norm2", D)
        if bb.swiglu:
            lin("mlp.weights_in", 2 * bb.ffn_hidden, D, W_STD)
            lin("mlp.weights_out", D, bb.ffn_hidden, W_STD)
        else:
            lin("mlp.fc1", bb.ffn_hidden, D, W_STD)
            lin("mlp.fc2", D, bb.ffn_hidden, W_STD)
        sd[lp + "layer_scale2.lambda1"] = 1.0 + normal(seed, lp + "layer_scale2.lambda1", (D,), 0.1)
    _ln(sd, seed, prefix + "dino.layernorm", D)
    if bb.target_dim:
        _lin(sd, seed, prefix + "projection", bb.target_dim, D, DEC_STD)
    return sd


def _decoder_layer_deformable(sd, seed, lp, dc: DecoderConfig):
    Dd, Hd, P, F = dc.hidden_dim, dc.nheads, dc.n_points, dc.dim_feedforward
    w = normal(seed, lp + "self_attn.in_proj_weight", (3 * Dd, Dd), DEC_STD)
    w[: 2 * Dd] *= 2.0   # q/k rows sharper
    sd[lp + "self_attn.in_proj_weight"] = w
    sd[lp + "self_attn.in_proj_bias"] = normal(seed, lp + "self_attn.in_proj_bias", (3 * Dd,), B_STD)
    _lin(sd, seed, lp + "self_attn.out_proj", Dd, Dd, DEC_STD)
    _ln(sd, seed, lp + "norm1", Dd)
    # small weight so offsets are O(0.1) and do not saturate the [0,1] clamp
    _lin(sd, seed, lp + "cross_attn.sampling_offsets", Hd * P * 2, Dd, 0.004, bias_std=0.05)
    _lin(sd, seed, lp + "cross_attn.attention_weights", Hd * P, Dd, 0.05, bias_std=0.05)
    _lin(sd, seed, lp + "cross_attn.value_proj", Dd, Dd, DEC_STD)
    _lin(sd, seed, lp + "cross_attn.output_proj", Dd, Dd, DEC_STD)
    _ln(sd, seed, lp + "norm2", Dd)
    _lin(sd, seed, lp + "linear1", F, Dd, DEC_STD)
    _lin(sd, seed, lp + "linear2", Dd, F, DEC_STD)
    _ln(sd, seed, lp + "norm3", Dd)
    # reference points move samples by (w-1)=136..256 tokens per unit: keep d(ref)/d(tgt) moderate so the
    # forward is not hopelessly ill-conditioned in fp32 (see DESIGN.md "conditioning")
    _lin(sd, seed, lp + "reference_points_proj", 2, Dd, 0.01, bias_std=0.3)


def _decoder_layer_standard(sd, seed, lp, dc: DecoderConfig):
    Dd, F = dc.hidden_dim, dc.dim_feedforward
    for att in ("self_attn", "multihead_attn"):
        w = normal(seed, lp + att + ".in_proj_weight", (3 * Dd, Dd), DEC_STD)
        w[: 2 * Dd] *= 2.0
        sd[lp + att + ".in_proj_weight"] = w
        sd[lp + att + ".in_proj_bias"] = normal(seed, lp + att + ".in_proj_bias", (3 * Dd,), B_STD)
        _lin(sd, seed, lp + att + ".out_proj", Dd, Dd, DEC_STD)
    _lin(sd, seed, lp + "linear1", F, Dd, DEC_STD)
    _lin(sd, seed, lp + "linear2", Dd, F, DEC_STD)
    for n in ("norm1", "norm2", "norm3"):
        _ln(sd, seed, lp + n, Dd)


def decoder_state_dict(dc: DecoderConfig, seed=1, prefix="decoder."):
    """Keys as the reference's DETRDecoder (detr_decoder.py:15-45).  In the
    deformable branch all `layers.{j}` alias ONE layer (deformable_attention.py:284):
    layer 0 is generated and the same arrays are stored under every j."""
    return _reprefix(_decoder_state_dict(dc, seed, "decoder."), "decoder.", prefix)


def _decoder_state_dict(dc, seed, prefix):
    sd = {}
    Dd = dc.hidden_dim
    sd[prefix + "query_embed.weight"] = normal(seed, prefix + "query_embed.weight",
                                               (dc.num_queries, Dd), 1.0)
    if dc.use_deformable:
        tmp = {}
        _decoder_layer_deformable(tmp, seed, prefix + "decoder.layers.0.", dc)
        for j in range(dc.num_layers):
            for k, v in tmp.items():
                sd[k.replace("decoder.layers.0.", f"decoder.layers.{j}.")] = v
    else:
        for j in range(dc.num_layers):
            _decoder_layer_standard(sd, seed, f"{prefix}decoder.layers.{j}.", dc)
    _lin(sd, seed, prefix + "class_embed", dc.num_classes, Dd, DEC_STD)
    _lin(sd, seed, prefix + "bbox_embed.mlp.0", Dd // 2, Dd, DEC_STD)
    _lin(sd, seed, prefix + "bbox_embed.mlp.2", 4, Dd // 2, DEC_STD)
    if dc.use_deformable:
        _lin(sd, seed, prefix + "reference_points", 2, Dd, DEC_STD)  # unused by forward
    return sd


def detector_state_dict(bb: BackboneConfig, dc: DecoderConfig, seed=1):
    sd = backbone_state_dict(bb, seed, "backbone.")
    sd.update(decoder_state_dict(dc, seed, "decoder."))
    return sd
