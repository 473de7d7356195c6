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


# ------------------------------------------------------------------------------------------------ G9: gradients of the reference
G9_CASES = {   # golden name -> (model name, R, B, detector kwargs) exactly as tests/golden/make_goldens.py::g9_gradients built them
    "g9_grad_cfg1": ("facebook/dinov2-small", 224, 2, dict(num_classes=91, hidden_dim=256, num_queries=25, num_decoder_layers=2,
                                                           dim_feedforward=512, lora_r=1, nheads=4, dropout=0.0)),
    "g9_grad_vitb_224": ("facebook/dinov2-base", 224, 2, dict(num_queries=100, dropout=0.0)),
    "g9_grad_cfg1_dense": ("facebook/dinov2-small", 224, 2, dict(num_classes=91, hidden_dim=256, num_queries=25, num_decoder_layers=2,
                                                                 dim_feedforward=512, lora_r=1, nheads=4, dropout=0.0, use_deformable=False)),
}


def grad_probe(a):
    """make_goldens.grad_probe: strided probe (<= 16 x 64) + (sum, abs-sum, L2) of one gradient tensor"""
    a = np.asarray(a)
    a2 = a.reshape(a.shape[0], -1) if a.ndim > 1 else a.reshape(1, -1)
    pr = a2[::max(1, a2.shape[0] // 16), ::max(1, a2.shape[1] // 64)][:16, :64].copy()
    return pr, np.array([a2.astype(np.float64).sum(), np.abs(a2).astype(np.float64).sum(), np.sqrt((a2.astype(np.float64) ** 2).sum())])


def g9_loss_weights(B, Q, C, seed=17):
    return synth.normal(seed, f"g9.gl.{B}.{Q}.{C}", (B, Q, C), 1.0), synth.normal(seed, f"g9.gb.{B}.{Q}", (B, Q, 4), 1.0)


def g9_check(model, g, tol_probe, tol_norm, to_np=lambda t: t.detach().cpu().numpy(), metric=None):
    """every gradient the reference's backward() produced (golden g) against `model`'s .grad: the probe entries (metric: rel_err =
    max-relative by default, or rel_l2), the whole tensor through its L2 norm and abs-sum; the reference's unreached tensors must be
    unreached here too.  Collects every violation before failing.  Returns the worst (probe error, name)."""
    metric = metric or rel_err
    params = dict(model.named_parameters())
    worst = (0.0, None)
    bad = []
    for k in g["trainable_with_grad"]:
        k = str(k)
        assert k in params and params[k].grad is not None, f"no gradient for {k}"
        pr, st = grad_probe(to_np(params[k].grad))
        e = metric(pr, g["grad:" + k])
        worst = max(worst, (e, k))
        en = abs(st[2] - g["stat:" + k][2]) / max(g["stat:" + k][2], 1e-30)
        ea = abs(st[1] - g["stat:" + k][1]) / max(g["stat:" + k][1], 1e-30)
        if not (e < tol_probe and en <= tol_norm and ea <= tol_norm):
            bad.append((k, float(e), float(en), float(ea)))
    assert not bad, f"{len(bad)} gradients outside (probe {tol_probe:g}, norms {tol_norm:g}): {bad[:6]}"
    for k in g["trainable_without_grad"]:
        p = params[str(k)]
        assert p.grad is None or float(p.grad.abs().sum()) == 0.0, k
    have = {k for k, p in params.items() if p.grad is not None and float(p.grad.abs().sum()) > 0}
    assert have == {str(k) for k in g["trainable_with_grad"]}, have ^ {str(k) for k in g["trainable_with_grad"]}
    return worst
