#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running THE REFERENCE ITSELF.

Runs only in the authoring container (needs /root/reference and `transformers`);
nothing of the reference is copied: the script imports its modules, loads the
deterministic synthetic state dict of dinov2_od_amd.synth into the reference's
own nn.Modules, runs their eval-mode fp32 CPU forward and stores inputs' seeds
and the resulting outputs (.npz, data only).

  python tests/golden/make_goldens.py [--only NAME]

Obstacles handled exactly as SURVEY.md section 8c records (ordinary Python errors,
no refused command): `pycocotools` / `tensorboard` are absent but only imported by
the reference's eval/logging helpers -> empty stub modules; there is no network
and no cached checkpoint -> `Dinov2Model.from_pretrained` is patched to build
the architecture from an explicit `Dinov2Config` (weights then come from synth).
"""
import argparse
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch

REF = "/root/reference"


def _import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    stub("pycocotools")
    stub("pycocotools.coco", COCO=type("COCO", (), {}))
    stub("pycocotools.cocoeval", COCOeval=type("COCOeval", (), {}))
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        stub("torch.utils.tensorboard", SummaryWriter=type("SummaryWriter", (), {}))
    sys.path.insert(0, REF)
    import transformers
    from transformers import Dinov2Config, Dinov2Model
    return transformers, Dinov2Config, Dinov2Model


transformers, Dinov2Config, Dinov2Model = _import_reference()

from dinov2_od_amd.config import BackboneConfig, DecoderConfig  # noqa: E402
from dinov2_od_amd import synth  # noqa: E402

_BB_FOR_PATCH = {}


def _patched_from_pretrained(name, *a, **k):
    bb = _BB_FOR_PATCH["bb"]
    cfg = Dinov2Config(image_size=bb.pos_grid * bb.patch, patch_size=bb.patch, hidden_size=bb.hidden,
                       num_hidden_layers=bb.layers, num_attention_heads=bb.heads,
                       mlp_ratio=bb.mlp_ratio, use_swiglu_ffn=bb.swiglu, layer_norm_eps=bb.ln_eps)
    return Dinov2Model(cfg)


Dinov2Model.from_pretrained = staticmethod(_patched_from_pretrained)

from dino_detector.models import DINOv2ObjectDetector, DINOv2Backbone, DETRDecoder  # noqa: E402


def _load(module, sd_np, strip=""):
    sd = {k[len(strip):]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()
          if k.startswith(strip)}
    missing, unexpected = module.load_state_dict(sd, strict=True), None
    return missing


def _meta():
    return dict(torch=torch.__version__, transformers=transformers.__version__, numpy=np.__version__,
                generator="dinov2_od_amd.synth v1")


def _save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    meta = _meta()
    np.savez_compressed(path, meta=np.array(repr(meta)), **arrs)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ------------------------------------------------------------------ cases
def micro_bb(swiglu=False):
    return BackboneConfig(hidden=128, layers=2, heads=2, swiglu=swiglu, pos_grid=5, lora_r=2,
                          lora_alpha=1.0, target_dim=0)


def g0_micro_backbone(swiglu=False):
    """G0 / G4: micro backbone through the reference's DINOv2Backbone; 70x70 (no
    pos-embed resize, N=26) and 56x56 (bicubic 5->4, N=17).  Every block output kept."""
    bb = micro_bb(swiglu)
    _BB_FOR_PATCH["bb"] = bb
    m = DINOv2Backbone(model_name="micro", lora_r=bb.lora_r, lora_alpha=bb.lora_alpha, target_dim=None).eval()
    sd = synth.backbone_state_dict(bb, seed=1, prefix="")
    _load(m, sd)
    out = {}
    for R in (70, 56):
        x = synth.make_pixels(2, R, R, seed=0)
        taps = {}
        hooks = [m.dino.embeddings.register_forward_hook(lambda mod, i, o: taps.__setitem__("embeddings", o))]
        for li, layer in enumerate(m.dino.encoder.layer):
            hooks.append(layer.register_forward_hook(
                lambda mod, i, o, li=li: taps.__setitem__(f"block{li}", o[0] if isinstance(o, tuple) else o)))
        with torch.no_grad():
            f = m(torch.from_numpy(x))
        for h in hooks:
            h.remove()
        out[f"features_{R}"] = f.numpy()
        for k, v in taps.items():
            out[f"{k}_{R}"] = v.detach().numpy()
    _save("g4_micro_swiglu" if swiglu else "g0_micro_backbone", **out)


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


def g1_decoder_only():
    out = {}
    for tag, deform, Dd, Hd, Q, Ns in G1_CASES:
        dc = dec_cfg(deform, Dd, Hd, Q)
        m = DETRDecoder(num_queries=dc.num_queries, hidden_dim=dc.hidden_dim, nheads=dc.nheads,
                        num_decoder_layers=dc.num_layers, num_classes=dc.num_classes,
                        dim_feedforward=dc.dim_feedforward, dropout=0.1, n_points=dc.n_points,
                        use_deformable=deform).eval()
        sd = synth.decoder_state_dict(dc, seed=1, prefix="")
        _load(m, sd)
        for N in Ns:
            mem = g1_memory(N, Dd)
            with torch.no_grad():
                o = m(torch.from_numpy(mem))
            out[f"{tag}_N{N}_logits"] = o["pred_logits"].numpy()
            out[f"{tag}_N{N}_boxes"] = o["pred_boxes"].numpy()
    _save("g1_decoder_only", **out)


def postprocess_inputs(B=5, Q=25, C=11, seed=11):
    """packed detections [B,Q,C+4]: logits ~ N(-3.2, 1.6) (about a third above the 0.05 score threshold), boxes
    uniform; logits whose score is within 1e-4 of the threshold are moved away (sigmoid implementations differ by an ulp)."""
    logits = (synth.normal(seed, "pp.logits", (B, Q, C), 1.6) - 3.2).astype(np.float32)
    s = 1.0 / (1.0 + np.exp(-logits.astype(np.float64)))
    logits[np.abs(s - 0.05) < 1e-4] -= 0.05
    boxes = synth.uniform01(seed, "pp.boxes", (B, Q, 4)).astype(np.float32)
    return np.concatenate([logits, boxes], axis=-1)


def g5_postprocess():
    """G5: the reference's evaluate_coco (utils.py:167-240) on fixed model outputs: two batches (3 + 2 images), one target
    without an image_id (falls back to the index in the batch, utils.py:203)."""
    from dino_detector.utils import evaluate_coco
    det = postprocess_inputs()
    C = det.shape[-1] - 4
    batches = [(0, 3), (3, 5)]
    ids = [[42, None, 100013], [7, 8]]

    class Fixed(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.i = 0

        def forward(self, images):
            lo, hi = batches[self.i]
            self.i += 1
            d = torch.from_numpy(det[lo:hi])
            return {"pred_logits": d[..., :C].contiguous(), "pred_boxes": d[..., C:].contiguous()}

    loader = [(torch.zeros(hi - lo, 3, 8, 8), [({"image_id": i} if i is not None else {}) for i in ids_b])
              for (lo, hi), ids_b in zip(batches, ids)]
    res = evaluate_coco(Fixed(), loader, torch.device("cpu"), None)
    _save("g5_postprocess", det=det,
          batch_bounds=np.array(batches, dtype=np.int64),
          image_ids=np.array([[-1 if i is None else i for i in (b + [-1] * 3)[:3]] for b in ids], dtype=np.int64),
          r_image_id=np.array([r["image_id"] for r in res], dtype=np.int64),
          r_category_id=np.array([r["category_id"] for r in res], dtype=np.int32),
          r_bbox=np.array([r["bbox"] for r in res], dtype=np.float64),
          r_score=np.array([r["score"] for r in res], dtype=np.float64))
    print(f"g5: {len(res)} records")


def matcher_inputs(B=4, Q=25, C=11, counts=(4, 0, 7, 30), seed=13):
    """packed detections + concatenated targets (labels, cxcywh boxes, offsets): image 1 has no boxes, image 3 has more
    targets than queries' worth of easy matches (30 > 25: a rectangular assignment)"""
    logits = synth.normal(seed, "mt.logits", (B, Q, C), 2.0).astype(np.float32)
    cxcy = 0.15 + 0.7 * synth.uniform01(seed, "mt.cxcy", (B, Q, 2))
    wh = 0.05 + 0.4 * synth.uniform01(seed, "mt.wh", (B, Q, 2))
    det = np.concatenate([logits, cxcy, wh], axis=-1).astype(np.float32)
    G = int(sum(counts))
    labels = (synth.uniform01(seed, "mt.labels", (G,)) * C).astype(np.int64).clip(0, C - 1)
    gcxcy = 0.15 + 0.7 * synth.uniform01(seed, "mt.gcxcy", (G, 2))
    gwh = 0.05 + 0.4 * synth.uniform01(seed, "mt.gwh", (G, 2))
    gt = np.concatenate([gcxcy, gwh], axis=-1).astype(np.float32)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return det, labels, gt, offs


def g6_matcher():
    """G6: the reference's HungarianMatcher (matching.py:43-121) on fixed outputs/targets.  The cost matrices it hands to
    scipy are captured by wrapping `linear_sum_assignment` in the reference module's namespace (nothing is restated)."""
    import dino_detector.matching as rm
    det, labels, gt, offs = matcher_inputs()
    B, Q = det.shape[:2]
    C = det.shape[-1] - 4
    captured = []
    real = rm.linear_sum_assignment

    def spy(cm):
        captured.append(np.array(cm, dtype=np.float32, copy=True))
        return real(cm)

    rm.linear_sum_assignment = spy
    try:
        out = {}
        for tag, kw in (("default", {}), ("g15", dict(cost_class=2.0, cost_bbox=1.0, cost_giou=3.0, focal_alpha=0.4, focal_gamma=1.5))):
            captured.clear()
            m = rm.HungarianMatcher(**kw)
            d = torch.from_numpy(det)
            outputs = {"pred_logits": d[..., :C].contiguous(), "pred_boxes": d[..., C:].contiguous()}
            targets = [{"labels": torch.from_numpy(labels[offs[b]:offs[b + 1]]), "boxes": torch.from_numpy(gt[offs[b]:offs[b + 1]])}
                       for b in range(B)]
            idx = m(outputs, targets)
            assert len(captured) == B
            for b in range(B):
                out[f"{tag}_cost{b}"] = captured[b]
                out[f"{tag}_i{b}"] = idx[b][0].numpy()
                out[f"{tag}_j{b}"] = idx[b][1].numpy()
        # an empty target dict short-circuits (matching.py:73-75)
        idx = rm.HungarianMatcher()(outputs, [{}] + targets[1:])
        out["emptydict_n0"] = np.array([len(idx[0][0]), len(idx[0][1])])
    finally:
        rm.linear_sum_assignment = real
    _save("g6_matcher", det=det, labels=labels, gt=gt, offs=offs, **out)


def _probe(t):
    """a small, position-spread probe of a [B, N, D] activation + three checksums of the whole tensor"""
    a = t.detach().numpy()
    return a[:, ::max(1, a.shape[1] // 8), :64].copy(), np.array([a.mean(), np.abs(a).mean(), np.sqrt((a.astype(np.float64) ** 2).sum())])


def _e2e(name, model_name, R, B, kwargs, probes=True, block_taps=()):
    from dinov2_od_amd.config import variant_of, BACKBONE_VARIANTS
    hid = kwargs.get("hidden_dim", 768)
    bb = BackboneConfig.from_name(model_name, lora_r=kwargs.get("lora_r", 2), lora_alpha=1.0, target_dim=hid)
    _BB_FOR_PATCH["bb"] = bb
    m = DINOv2ObjectDetector(dino_model_name=model_name, **kwargs).eval()
    dc = DecoderConfig(num_queries=kwargs.get("num_queries", 50), hidden_dim=hid,
                       nheads=kwargs.get("nheads", 8), num_layers=kwargs.get("num_decoder_layers", 3),
                       num_classes=kwargs.get("num_classes", 91),
                       dim_feedforward=kwargs.get("dim_feedforward", 1024),
                       n_points=kwargs.get("n_points", 2), use_deformable=kwargs.get("use_deformable", True))
    sd = synth.detector_state_dict(bb, dc, seed=1)
    _load(m, sd)
    x = synth.make_pixels(B, R, R, seed=0)
    del sd
    taps, hooks = {}, []
    if block_taps:   # full-depth fixtures: probes of the embeddings and of chosen block outputs (per-stage parity on the GPU)
        hooks.append(m.backbone.dino.embeddings.register_forward_hook(lambda mod, i, o: taps.__setitem__("embeddings", _probe(o))))
        for li in block_taps:
            hooks.append(m.backbone.dino.encoder.layer[li].register_forward_hook(
                lambda mod, i, o, li=li: taps.__setitem__(f"block{li}", _probe(o[0] if isinstance(o, tuple) else o))))
    with torch.no_grad():
        feats = m.backbone(torch.from_numpy(x))
        o = m.decoder(feats)
    for h in hooks:
        h.remove()
    f = feats.numpy()
    arrs = dict(pred_logits=o["pred_logits"].numpy(), pred_boxes=o["pred_boxes"].numpy(),
                feat_probe=f[:, ::max(1, f.shape[1] // 8), :64].copy(),
                feat_stats=np.array([f.mean(), np.abs(f).mean(), np.sqrt((f.astype(np.float64) ** 2).sum())]))
    for k, (pr, st) in taps.items():
        arrs[k + "_probe"], arrs[k + "_stats"] = pr, st
    _save(name, **arrs)


def g2_cfg1():
    """BASELINE.json configs[0]: --lightweight ViT-S/14 224x224, batch 2 (train.py:607-640 preset);
    Q=25 (CLI preset) and Q=100 (as BASELINE.json states)."""
    for Q in (25, 100):
        _e2e(f"g2_cfg1_q{Q}", "facebook/dinov2-small", 224, 2,
             dict(num_classes=91, hidden_dim=256, num_queries=Q, num_decoder_layers=2,
                  dim_feedforward=512, lora_r=1, nheads=4))


def g3_vitb():
    for R in (224, 518):
        _e2e(f"g3_vitb_{R}", "facebook/dinov2-base", R, 1, dict(num_queries=100))
    _e2e("g3_vitb_224_dense", "facebook/dinov2-base", 224, 1, dict(num_queries=100, use_deformable=False))


def g7_vitl():
    """BASELINE.json configs[3] at its workload: ViT-L/14 518x518, 300 queries, full depth (24 blocks), one image."""
    _e2e("g7_vitl_518", "facebook/dinov2-large", 518, 1, dict(num_queries=300), block_taps=(0, 11, 23))


def g8_vitg():
    """BASELINE.json configs[4] at its workload: ViT-g/14 518x518 (SwiGLU, 40 blocks), 300 queries, one image."""
    _e2e("g8_vitg_518", "facebook/dinov2-giant", 518, 1, dict(num_queries=300), block_taps=(0, 19, 39))


def grad_probe(a):
    """what G9 keeps of one gradient tensor: a strided probe (<= 16 x 64 entries) and three checksums of the whole tensor"""
    a = np.asarray(a)
    a2 = a.reshape(a.shape[0], -1) if a.ndim > 1 else a.reshape(1, -1)
    pr = a2[::max(1, a2.shape[0] // 16), ::max(1, a2.shape[1] // 64)][:16, :64].copy()
    return pr, np.array([a2.astype(np.float64).sum(), np.abs(a2).astype(np.float64).sum(), np.sqrt((a2.astype(np.float64) ** 2).sum())])


def g9_loss_weights(B, Q, C, seed=17):
    """the fixed linear loss of G9: loss = sum(pred_logits * Gl) + sum(pred_boxes * Gb), so d loss / d outputs = (Gl, Gb) exactly"""
    return synth.normal(seed, f"g9.gl.{B}.{Q}.{C}", (B, Q, C), 1.0), synth.normal(seed, f"g9.gb.{B}.{Q}", (B, Q, 4), 1.0)


def g9_gradients():
    """G9: what `loss.backward()` (train.py:1101) leaves in `.grad` of every trainable parameter of the REFERENCE's
    DINOv2ObjectDetector in train() mode -- decoder + heads (deformable_attention.py:215-268, tied layers :284), the projection and
    the LoRA A/B of the last two blocks through the frozen base (dinov2_backbone.py:40-51, utils.py:68-70) -- with dropout = 0 (the
    five dropout sites are identities: no RNG to reproduce) and a fixed linear loss.  Kept per tensor: a strided probe and checksums
    of the whole gradient (grad_probe).  cfg1 (ViT-S/14 lightweight, with projection) and ViT-B/14 224x224 with the default 768-wide
    decoder (the decoder every BASELINE config uses)."""
    cases_ = [("g9_grad_cfg1", "facebook/dinov2-small", 224, 2,
               dict(num_classes=91, hidden_dim=256, num_queries=25, num_decoder_layers=2, dim_feedforward=512, lora_r=1, nheads=4, dropout=0.0)),
              ("g9_grad_vitb_224", "facebook/dinov2-base", 224, 2, dict(num_queries=100, dropout=0.0)),
              # the nn.TransformerDecoder branch (detr_decoder.py:28-35, 62-69): untied layers, dense cross-attention over all tokens
              ("g9_grad_cfg1_dense", "facebook/dinov2-small", 224, 2,
               dict(num_classes=91, hidden_dim=256, num_queries=25, num_decoder_layers=2, dim_feedforward=512, lora_r=1, nheads=4, dropout=0.0,
                    use_deformable=False))]
    # the same ViT-B case evaluated by the reference's own modules in float64 (`.double()`: plain PyTorch on the CPU): the arbiter between two
    # fp32 evaluations of an ill-conditioned gradient (three tied layers whose sampling gradient is piecewise smooth)
    cases_.append(("g9_grad_vitb_224_f64", "facebook/dinov2-base", 224, 2, dict(num_queries=100, dropout=0.0)))
    for name, model_name, R, B, kwargs in cases_:
        if G9_ONLY and name not in G9_ONLY:
            continue
        f64 = name.endswith("_f64")
        hid = kwargs.get("hidden_dim", 768)
        bb = BackboneConfig.from_name(model_name, lora_r=kwargs.get("lora_r", 2), lora_alpha=1.0, target_dim=hid)
        _BB_FOR_PATCH["bb"] = bb
        m = DINOv2ObjectDetector(dino_model_name=model_name, **kwargs)
        dc = DecoderConfig(num_queries=kwargs.get("num_queries", 50), hidden_dim=hid, nheads=kwargs.get("nheads", 8),
                           num_layers=kwargs.get("num_decoder_layers", 3), num_classes=kwargs.get("num_classes", 91),
                           dim_feedforward=kwargs.get("dim_feedforward", 1024), n_points=kwargs.get("n_points", 2),
                           use_deformable=kwargs.get("use_deformable", True))
        _load(m, synth.detector_state_dict(bb, dc, seed=1))
        m.train()
        x = torch.from_numpy(synth.make_pixels(B, R, R, seed=0))
        gl, gb = g9_loss_weights(B, dc.num_queries, dc.num_classes)
        gl, gb = torch.from_numpy(gl), torch.from_numpy(gb)
        if f64:
            m, x, gl, gb = m.double(), x.double(), gl.double(), gb.double()
        o = m(x)
        loss = (o["pred_logits"] * gl).sum() + (o["pred_boxes"] * gb).sum()
        loss.backward()
        arrs = dict(pred_logits=o["pred_logits"].detach().numpy(), pred_boxes=o["pred_boxes"].detach().numpy(), loss=np.array(float(loss)))
        names, nograd = [], []
        for k, p in m.named_parameters():          # tied layers: one name per storage (named_parameters de-duplicates)
            if not p.requires_grad:
                continue
            if p.grad is None:
                nograd.append(k)
                continue
            pr, st = grad_probe(p.grad.numpy())
            arrs["grad:" + k], arrs["stat:" + k] = pr, st
            names.append(k)
        arrs["trainable_with_grad"] = np.array(names)
        arrs["trainable_without_grad"] = np.array(nograd)
        _save(name, **arrs)
        print(f"{name}: {len(names)} gradients, {len(nograd)} trainable tensors the loss does not reach: {nograd}")


G9_ONLY = [n for n in os.environ.get("G9_ONLY", "").split(",") if n]      # regenerate single G9 cases (the float64 one takes minutes)

CASES = dict(g0=lambda: g0_micro_backbone(False), g4=lambda: g0_micro_backbone(True),
             g1=g1_decoder_only, g2=g2_cfg1, g3=g3_vitb, g5=g5_postprocess, g6=g6_matcher, g7=g7_vitl, g8=g8_vitg, g9=g9_gradients)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    torch.manual_seed(0)
    for k, fn in CASES.items():
        if a.only and a.only != k:
            continue
        print("==", k)
        fn()
