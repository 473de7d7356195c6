"""`-m gpu`: the native training step of the decoder + heads (dec_train.hip through dod_decoder_train_forward / _backward) against
the autograd composite of the same math (models/_autograd.py, itself pinned to the reference goldens by
tests/test_train_composite.py): forward identical, every parameter gradient and d(memory) within 1e-4 -- the gradients
`loss.backward()` at train.py:1101 produces for the decoder and heads."""
import os

import numpy as np
import pytest
import torch

from dinov2_od_amd import _native as nat, synth
from tests import cases
from tests.cases import rel_err, rel_l2

pytestmark = pytest.mark.gpu


def _decoder(dc, dropout=0.0):
    from dinov2_od_amd.models import DETRDecoder
    from tests import gpu_util as G
    m = DETRDecoder(dc.num_queries, dc.hidden_dim, dc.nheads, dc.num_layers, dc.num_classes, dim_feedforward=dc.dim_feedforward,
                    dropout=dropout, n_points=dc.n_points, use_deformable=True, precision="fp32")
    G.load_np_state(m, synth.decoder_state_dict(dc, seed=1, prefix=""))
    m = m.to(G.dev()).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = dropout
    return m


def _loss(o, wl, wb):
    return (o["pred_logits"] * wl).sum() + (o["pred_boxes"] * wb).sum()


def _run(m, mem, wl, wb, native):
    os.environ["DINODET_NATIVE_TRAIN"] = "1" if native else "0"
    try:
        m.zero_grad(set_to_none=True)
        x = mem.clone().requires_grad_(True)
        o = m(x)
        _loss(o, wl, wb).backward()
        g = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        return o["pred_logits"].detach().clone(), o["pred_boxes"].detach().clone(), x.grad.detach().clone(), g
    finally:
        os.environ.pop("DINODET_NATIVE_TRAIN", None)


CASES = [  # Dd, Hd, Q, layers, F, C, P, B, N
    (128, 4, 7, 2, 256, 11, 2, 2, 26),       # micro: (h, w) = (2, 13)
    (128, 4, 7, 2, 256, 11, 2, 3, 257),      # prime token count: (1, 257)
    (192, 2, 5, 2, 256, 11, 4, 2, 1370),     # head_dim 96, 4 points, (10, 137)
    (256, 4, 25, 2, 512, 91, 2, 2, 257),     # the --lightweight decoder (train.py:607-640)
    (768, 8, 100, 3, 1024, 91, 2, 2, 257),   # config.py:21-35 defaults (tied x3), 224x224 memory
]
# memory scale per decoder width: at Dd = 768 three tied layers on unit-variance random memory are ill-conditioned in fp32 (a
# reference-point error moves a sample by 256 tokens: two fp32 evaluations differ by 1e-2 there, by 2e-5 at this scale)
MEM_STD = {768: 0.1}


@pytest.mark.parametrize("case", CASES, ids=[f"Dd{c[0]}_Q{c[2]}_N{c[8]}" for c in CASES])
def test_native_decoder_backward_matches_composite_autograd(case):
    from tests import gpu_util as G
    Dd, Hd, Q, layers, F, C, P, B, N = case
    dc = cases.dec_cfg(True, Dd, Hd, Q, layers, F, C, P)
    m = _decoder(dc)
    mem = G.to_gpu(synth.normal(3, f"memory.train.{N}.{Dd}", (B, N, Dd), 1.0) * np.float32(MEM_STD.get(Dd, 1.0)))
    wl = G.to_gpu(synth.normal(5, "train.wl", (B, Q, C), 1.0))
    wb = G.to_gpu(synth.normal(5, "train.wb", (B, Q, 4), 1.0))
    l0, b0, dx0, g0 = _run(m, mem, wl, wb, native=False)
    l1, b1, dx1, g1 = _run(m, mem, wl, wb, native=True)
    # 1e-4 on every gradient; 1e-3 for the three tied 768-wide layers: the gradient with respect to a sampling location is
    # discontinuous across cells of the bilinear gather, so the 2e-5 forward difference of two fp32 evaluations re-routes a few
    # samples' gradient (one such layer alone: 2e-6 .. 5e-5; measured 8e-5 .. 4e-4 for three)
    # and a ReLU unit within that 2e-5 of zero flips its whole row's contribution (76 800 unit-rows in the box head: ~1.5 flips
    # expected) -- there the criterion is the L2 error of each gradient tensor, which isolated flips do not dominate
    deep = Dd >= 768 and layers >= 3
    gtol = 1e-2 if deep else 1e-4     # one flipped unit-row of the 384 x 200 box head alone is 3.6e-3 of that tensor's L2 norm
    err = cases.rel_l2 if deep else rel_err
    # two fp32 evaluations (hipBLASLt + SDPA vs the exact-fp32 MFMA kernels): the fp32 noise floor of this decoder
    assert rel_err(l1.cpu().numpy(), l0.cpu().numpy()) < 1e-4 and rel_err(b1.cpu().numpy(), b0.cpu().numpy()) < 1e-4
    assert set(g0) == set(g1) and len(g1) >= 30
    worst = ("", 0.0)
    for k in g0:
        e = err(g1[k].cpu().numpy(), g0[k].cpu().numpy())
        worst = max(worst, (k, e), key=lambda t: t[1])
        assert e < gtol, (k, e)
    e = err(dx1.cpu().numpy(), dx0.cpu().numpy())
    print(f"native vs composite gradients {case}: worst parameter {worst[0]} {worst[1]:.2e}, d(memory) {e:.2e}")
    assert e < gtol
    # the unused reference_points head (detr_decoder.py:44-45) gets no gradient in either path
    assert not any(k.startswith("reference_points.") for k in g1)
    if B > 1:       # self-attention adjoint in passes of one image (score scratch capped): same forward bit for bit, same gradients
        nat.set_option("mha_chunk_images", 1)
        try:
            l2, b2, dx2, g2 = _run(m, mem, wl, wb, native=True)
        finally:
            nat.set_option("mha_chunk_images", -1)
        assert torch.equal(l2, l1) and torch.equal(b2, b1)
        for k in g1:        # weight gradients accumulate atomically: equal up to the order of fp32 additions (a bias gradient is a sum of
            assert rel_err(g2[k].cpu().numpy(), g1[k].cpu().numpy()) < 1e-5, k     # B*Q terms of both signs: 2.4e-6 seen on one box, 1e-6 typical)


@pytest.mark.parametrize("case", [CASES[1], CASES[2], CASES[4]], ids=lambda c: f"Dd{c[0]}_Q{c[2]}_N{c[8]}")
def test_deterministic_mode_is_bit_reproducible_and_agrees_with_the_fast_step(case):
    """DINODET_DETERMINISTIC=1 / test option "deterministic" (dec_train.hip det_mode): every reduction the fast training step merges with
    fp32 atomics -- K-split gradient products, bias sums, LayerNorm gradients, the deformable sampling adjoint's scatter and its shared
    reference-logit columns -- runs in a fixed order.  Three runs of the same step give bit-identical gradients; the fast step (whose
    runs agree with each other to ~1e-6) agrees with them to 1e-5."""
    from tests import gpu_util as G
    Dd, Hd, Q, layers, F, C, P, B, N = case
    dc = cases.dec_cfg(True, Dd, Hd, Q, layers, F, C, P)
    m = _decoder(dc)
    mem = G.to_gpu(synth.normal(3, f"memory.det.{N}.{Dd}", (B, N, Dd), MEM_STD.get(Dd, 1.0)))
    wl = G.to_gpu(synth.normal(5, "det.wl", (B, Q, C), 1.0))
    wb = G.to_gpu(synth.normal(5, "det.wb", (B, Q, 4), 1.0))
    fast = _run(m, mem, wl, wb, native=True)
    nat.set_option("deterministic", 1)
    try:
        runs = [_run(m, mem, wl, wb, native=True) for _ in range(3)]
    finally:
        nat.set_option("deterministic", -1)
    for r in runs[1:]:
        assert torch.equal(r[0], runs[0][0]) and torch.equal(r[1], runs[0][1]) and torch.equal(r[2], runs[0][2]), "outputs / d(memory) differ between runs"
        assert set(r[3]) == set(runs[0][3])
        for k in r[3]:
            assert torch.equal(r[3][k], runs[0][3][k]), f"{k}: not bit-reproducible in deterministic mode"
    worst = max((rel_err(fast[3][k].cpu().numpy(), runs[0][3][k].cpu().numpy()), k) for k in fast[3])
    print(f"deterministic vs fast step {case}: worst gradient {worst[0]:.2e} ({worst[1]}); d(memory) {rel_err(fast[2].cpu().numpy(), runs[0][2].cpu().numpy()):.2e}")
    assert worst[0] < 1e-5 and rel_err(fast[2].cpu().numpy(), runs[0][2].cpu().numpy()) < 1e-5


def test_native_decoder_dropout_masks_are_consistent_and_seeded():
    """dropout 0.1 at the reference's five sites: the forward is a deterministic function of the seed, the drop rate is right, and
    the backward applies the SAME masks (directional finite difference of the loss against the analytic gradient)."""
    from dinov2_od_amd.models import _native_train as nt
    from tests import gpu_util as G
    dc = cases.dec_cfg(True, 128, 4, 16, 2, 256, 11, 2)
    m = _decoder(dc, dropout=0.1)
    B, N = 4, 257
    mem = G.to_gpu(synth.normal(3, "memory.drop", (B, N, 128), 1.0))
    a = nt.decoder_train(m, mem, seed=1234).detach().clone()
    b = nt.decoder_train(m, mem, seed=1234).detach().clone()
    c = nt.decoder_train(m, mem, seed=99).detach().clone()
    assert torch.equal(a, b) and not torch.equal(a, c)
    # the self-attention runs in passes over image chunks (score scratch capped): the chunking must change nothing, masks included
    nat.set_option("mha_chunk_images", 3)               # B = 4: passes of 3 + 1 images
    try:
        a3 = nt.decoder_train(m, mem, seed=1234).detach().clone()
    finally:
        nat.set_option("mha_chunk_images", -1)
    assert torch.equal(a, a3)
    m.eval()                                             # p = 0 through the same entry
    d = nt.decoder_train(m, mem, seed=1234).detach().clone()
    m.train()
    assert not torch.allclose(a, d)
    # finite-difference check along a random direction of EVERY parameter and of the memory.  The sampling locations are frozen
    # for it (zero weights of reference_points_proj / sampling_offsets, biases kept): a bilinear gather over random memory is
    # only piecewise smooth, and a step that moves the samples across cell borders says nothing about the masks.
    L0 = m.decoder.layers[0]
    with torch.no_grad():
        L0.reference_points_proj.weight.zero_()
        L0.cross_attn.sampling_offsets.weight.zero_()
    # (their own gradients are pinned against the composite above; a finite difference ALONG them crosses cell borders)
    frozen = {id(L0.reference_points_proj.weight), id(L0.reference_points_proj.bias), id(L0.cross_attn.sampling_offsets.weight),
              id(L0.cross_attn.sampling_offsets.bias)}
    params = [q for q in m.parameters() if q.requires_grad and id(q) not in frozen]
    g = torch.Generator(device="cpu").manual_seed(0)
    dirs = [torch.randn(q.shape, generator=g).to(q.device) * (0.03 / max(1.0, q.numel() ** 0.5)) for q in params]
    vx = torch.randn(mem.shape, generator=g).to(mem.device) * (0.03 / mem.numel() ** 0.5)
    wl = G.to_gpu(synth.normal(5, "drop.wl", (B, 16, 15), 1.0))

    def f(x):
        return (nt.decoder_train(m, x, seed=77) * wl).sum()
    x = mem.clone().requires_grad_(True)
    m.zero_grad(set_to_none=True)
    f(x).backward()
    ana = float(sum((q.grad * v).sum() for q, v in zip(params, dirs) if q.grad is not None) + (x.grad * vx).sum())
    with torch.no_grad():
        for q, v in zip(params, dirs):
            q.add_(v)
        lp = float(f(mem + vx))
        for q, v in zip(params, dirs):
            q.sub_(2 * v)
        lm = float(f(mem - vx))
        for q, v in zip(params, dirs):
            q.add_(v)
    num = (lp - lm) / 2
    print(f"dropout directional derivative: analytic {ana:.5f}, central difference {num:.5f}")
    assert abs(ana - num) < 2e-2 * abs(num) + 1e-3


def test_detector_train_step_uses_the_native_decoder_backward():
    """train.py:1079-1109 on the drop-in detector: the decoder / head gradients come from the native backward, d(memory) continues
    through the composite's projection and LoRA blocks; same gradients as the all-composite step."""
    from tests import gpu_util as G
    bb, dc = cases.cfg1(25)
    m = G.make_detector(bb, dc, "fp32", "facebook/dinov2-small")
    m.train()
    G.no_dropout(m)
    x = G.to_gpu(synth.make_pixels(2, 224, 224, seed=0))

    def run(native):
        os.environ["DINODET_NATIVE_TRAIN"] = "1" if native else "0"
        try:
            m.zero_grad(set_to_none=True)
            o = m(x)
            (o["pred_logits"].square().mean() + o["pred_boxes"].mean()).backward()
            return o["pred_logits"].detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        finally:
            os.environ.pop("DINODET_NATIVE_TRAIN", None)
    l0, g0 = run(False)
    l1, g1 = run(True)
    # the whole native step (backbone tail with its LoRA gradient kernels + decoder) in deterministic mode: bit-identical between runs
    nat.set_option("deterministic", 1)
    try:
        ld, gd = run(True)
        ld2, gd2 = run(True)
    finally:
        nat.set_option("deterministic", -1)
    assert torch.equal(ld, ld2) and set(gd) == set(gd2) == set(g1)
    for k in gd:
        assert torch.equal(gd[k], gd2[k]), f"{k}: not bit-reproducible in deterministic mode"
        assert rel_err(gd[k].cpu().numpy(), g1[k].cpu().numpy()) < 5e-5, k      # (a two-element bias: a sum of B*Q terms of both signs, 1.1e-5 seen)
    assert rel_err(l1.cpu().numpy(), l0.cpu().numpy()) < 1e-4
    assert set(g0) == set(g1) and any("lora_A" in k for k in g1) and any(k.startswith("backbone.projection") for k in g1)
    for k in g0:
        assert rel_err(g1[k].cpu().numpy(), g0[k].cpu().numpy()) < 2e-3, k      # LoRA grads pass through two blocks of fp32 autograd


@pytest.mark.parametrize("variant,R,B", [("micro", 70, 3), ("micro_r12", 70, 3), ("micro_swiglu", 70, 3), ("small", 224, 2), ("base", 224, 2),
                                         ("small", 518, 1), ("giant3", 224, 2)])
def test_native_backbone_tail_backward_matches_composite_autograd(variant, R, B):
    """The LoRA-adapted blocks + final LayerNorm + projection (dod_backbone_tail_train_*): memory and the gradients of every
    lora_A / lora_B and of the projection against the composite's autograd on the same frozen-prefix output.  micro_r12: rank 12
    takes the generic rank-r products (k-major fp32 GEMMs) instead of the r <= 8 kernels; 518: 1 370 tokens per image."""
    from dinov2_od_amd.config import BackboneConfig
    from dinov2_od_amd.models import DINOv2Backbone
    from tests import gpu_util as G
    if variant.startswith("micro"):
        bb = cases.micro_bb(variant == "micro_swiglu")     # SwiGLU MLP (modeling_dinov2.py:300-314): weights_in / weights_out carry the LoRA pairs
        bb.target_dim, bb.layers = 64, 3          # one frozen block in front of the two adapted ones (the native prefix needs >= 1)
        if variant == "micro_r12":
            bb.lora_r = 12
        variant = "micro"
    elif variant == "giant3":                     # ViT-g/14 widths (1536, 24 heads, SwiGLU 4096) at three blocks: configs[4]'s trainable tail
        bb = BackboneConfig.from_name("facebook/dinov2-giant", lora_r=2, lora_alpha=1.0, target_dim=768)
        bb.layers = 3
        variant = "giant"
    else:
        bb = BackboneConfig.from_name(f"facebook/dinov2-{variant}", lora_r=2, lora_alpha=1.0, target_dim=256)
    m = DINOv2Backbone(variant, lora_r=bb.lora_r, lora_alpha=bb.lora_alpha, target_dim=bb.target_dim, pretrained=False, precision="fp32", config=bb)
    G.load_np_state(m, synth.backbone_state_dict(bb, seed=1, prefix=""))
    m = m.to(G.dev()).train()
    x = G.to_gpu(synth.make_pixels(B, R, R, seed=0))
    N = (R // 14) ** 2 + 1
    wgt = G.to_gpu(synth.normal(7, f"tail.w.{variant}", (B, N, bb.target_dim), 1.0))

    def run(native):
        os.environ["DINODET_NATIVE_TRAIN"] = "1" if native else "0"
        try:
            m.zero_grad(set_to_none=True)
            mem = m(x)
            (mem * wgt).sum().backward()
            return mem.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        finally:
            os.environ.pop("DINODET_NATIVE_TRAIN", None)
    m0, g0 = run(False)
    m1, g1 = run(True)
    assert rel_err(m1.cpu().numpy(), m0.cpu().numpy()) < 2e-5
    assert set(g0) == set(g1) and sum("lora_A" in k for k in g1) == 12 and "projection.weight" in g1
    # the same step in float64 on the CPU (the composite, every block): which of the two fp32 evaluations is closer to it
    m64 = DINOv2Backbone(variant, lora_r=bb.lora_r, lora_alpha=bb.lora_alpha, target_dim=bb.target_dim, pretrained=False, precision="fp32", config=bb)
    m64.load_state_dict({k: torch.from_numpy(v) for k, v in synth.backbone_state_dict(bb, seed=1, prefix="").items()}, strict=True)
    m64 = m64.double().train()
    os.environ["DINODET_COMPOSITE_ON_CPU"] = "1"
    try:
        (m64(x.cpu().double()) * wgt.cpu().double()).sum().backward()
    finally:
        os.environ.pop("DINODET_COMPOSITE_ON_CPU", None)
    g64 = {k: p.grad for k, p in m64.named_parameters() if p.grad is not None}
    assert set(g64) == set(g1)
    # rounding grows with the length of the softmax / token reductions: 257 tokens stay inside 1e-4 of the fp32 composite; at 1 370
    # tokens the two fp32 evaluations sit ~1e-4 apart and the native one must be no further from float64 than twice the composite
    tol = 1e-4 if N <= 257 else 3e-4
    worst = ("", 0.0)
    worst64 = ("", 0.0, 0.0)
    for k in g0:
        e = rel_err(g1[k].cpu().numpy(), g0[k].cpu().numpy())
        worst = max(worst, (k, e), key=lambda t: t[1])
        assert e < tol, (k, e)
        ref = g64[k].numpy()
        en, ec = rel_err(g1[k].double().cpu().numpy(), ref), rel_err(g0[k].double().cpu().numpy(), ref)
        worst64 = max(worst64, (k, en, ec), key=lambda t: t[1])
        # the flash-style attention adjoint (N >= 1 024 tokens) takes delta = <dO, O> from the forward's rounded output: measured 5.3e-5
        # from float64 on the worst tensor at 1 370 tokens, where the batched form gave 1.7e-5 and the composite 2.3e-5
        assert en < (max(2e-5, 2.0 * ec) if N < 1024 else max(8e-5, 3.0 * ec)), (k, en, ec)
    assert worst[1] > 0.0, "both runs took the same path"
    print(f"backbone tail {variant} R={R}: vs float64 worst native {worst64[1]:.2e} (composite {worst64[2]:.2e}) at {worst64[0]}")
    print(f"backbone tail {variant} R={R}: memory {rel_err(m1.cpu().numpy(), m0.cpu().numpy()):.2e}, worst gradient {worst[0]} {worst[1]:.2e}")


_F64_YARD = {}     # distance of PyTorch-ROCm's own fp32 autograd (all blocks in torch) to the float64 golden, per tensor: computed once


@pytest.mark.parametrize("name", sorted(cases.G9_CASES))
@pytest.mark.parametrize("native", [True, False], ids=["native", "composite"])
def test_train_step_gradients_match_the_reference_backward(name, native):
    """G9 (tests/golden/make_goldens.py::g9_gradients): what the REFERENCE's own `loss.backward()` (train.py:1101) leaves in .grad of
    every trainable parameter -- decoder + heads through the tied deformable layers, projection, LoRA A / B of the last two blocks
    through the frozen base -- dropout 0, fixed linear loss.  The native HIP step (dec_train.hip) and, beside it, the PyTorch-ROCm
    composite on the native frozen prefix are held to those gradients directly: per tensor a strided probe and the L2 / abs-sum of
    the whole gradient."""
    from tests import gpu_util as G
    from dinov2_od_amd.models import DINOv2ObjectDetector
    g = cases.golden(name)
    model_name, R, B, kw = cases.G9_CASES[name]
    m = DINOv2ObjectDetector(dino_model_name=model_name, pretrained=False, precision="fp32", **kw)
    G.load_np_state(m, synth.detector_state_dict(m._bb_cfg, m._dc_cfg, seed=1))
    m = m.to(G.dev()).train()
    x = G.to_gpu(synth.make_pixels(B, R, R, seed=0))
    gl, gb = cases.g9_loss_weights(B, m._dc_cfg.num_queries, m._dc_cfg.num_classes)
    os.environ["DINODET_NATIVE_TRAIN"] = "1" if native else "0"
    try:
        o = m(x)
        loss = (o["pred_logits"] * G.to_gpu(gl)).sum() + (o["pred_boxes"] * G.to_gpu(gb)).sum()
        loss.backward()
    finally:
        os.environ.pop("DINODET_NATIVE_TRAIN", None)
    G.sync()
    el, eb = rel_err(o["pred_logits"].detach().cpu().numpy(), g["pred_logits"]), rel_err(o["pred_boxes"].detach().cpu().numpy(), g["pred_boxes"])
    assert el < 1e-3 and eb < 1e-3
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-3 * max(1.0, abs(float(g["loss"])))
    # cfg1 and the dense branch agree with the reference's fp32 gradients to 2-6e-5 on every tensor: held at 2e-4 max-relative
    deep = "vitb" in name
    if not deep:
        worst = cases.g9_check(m, g, 2e-4, 2e-4)
        print(f"{name} {'native' if native else 'composite'}: forward logits {el:.2e} boxes {eb:.2e}; worst gradient probe {worst[0]:.2e} ({worst[1]})")
        return
    # The default 768-wide decoder: arbitrated by the REFERENCE's own modules evaluated in float64 (golden g9_grad_vitb_224_f64, the same
    # inputs through `.double()`).  Three tied layers whose sampling gradient is piecewise smooth make this gradient ill-conditioned: the
    # reference's OWN fp32 CPU backward sits 2-5e-3 (probe rel-L2) from its float64 evaluation on the LoRA tensors of blocks 10 / 11.  An fp32
    # evaluation on the GPU sums in another order (k-ordered MFMA chains instead of the CPU's vector lanes) and lands 2-5x further out --
    # PyTorch-ROCm's own kernels exactly as far as these (measured: median ratio 2.26 composite, 2.27 native).  So the gate is two-sided:
    #   d(this step, float64) <= 1.5 x d(PyTorch-ROCm fp32 autograd of the same math on this GPU, float64)      -- the yardstick, per tensor (or <= 3e-3)
    #   d(this step, float64) <= 8 x max(d(reference fp32 CPU, float64), 3e-4)                                   -- and an absolute cap
    g64 = cases.golden(name + "_f64")
    params = dict(m.named_parameters())

    def dist64(grads):
        out = {}
        for k in map(str, g["trainable_with_grad"]):
            pr, st = cases.grad_probe(grads[k])
            out[k] = max(rel_l2(pr, g64["grad:" + k]), abs(st[2] - g64["stat:" + k][2]) / g64["stat:" + k][2])
        return out
    mine = {k: p.grad.detach().cpu().numpy() for k, p in params.items() if p.grad is not None}
    for k in map(str, g["trainable_with_grad"]):
        assert k in mine, k
    if "yard" not in _F64_YARD:       # the all-PyTorch evaluation on this GPU (every block in torch: DINODET_COMPOSITE_FULL=1), once
        os.environ["DINODET_NATIVE_TRAIN"], os.environ["DINODET_COMPOSITE_FULL"] = "0", "1"
        try:
            m.zero_grad(set_to_none=True)
            o2 = m(x)
            ((o2["pred_logits"] * G.to_gpu(gl)).sum() + (o2["pred_boxes"] * G.to_gpu(gb)).sum()).backward()
            _F64_YARD["yard"] = dist64({k: p.grad.detach().cpu().numpy() for k, p in params.items() if p.grad is not None})
        finally:
            os.environ.pop("DINODET_NATIVE_TRAIN", None)
            os.environ.pop("DINODET_COMPOSITE_FULL", None)
    yard, d_me = _F64_YARD["yard"], dist64(mine)
    rows, bad = [], []
    for k in map(str, g["trainable_with_grad"]):
        floor = max(rel_l2(g["grad:" + k], g64["grad:" + k]), abs(g["stat:" + k][2] - g64["stat:" + k][2]) / g64["stat:" + k][2], 3e-4)
        rows.append((d_me[k] / floor, d_me[k], yard[k], floor, k))
        # (3e-3: one ReLU unit or one sampling cell within rounding of its border flips between two evaluations and moves a small tensor by that
        #  much -- decoder.bbox_embed.mlp.0.bias: 2.3e-3 native, 1e-6 PyTorch-ROCm, with every other tensor of the two within 5 % of each other)
        if not ((d_me[k] <= 1.5 * max(yard[k], floor) or d_me[k] <= 3e-3) and d_me[k] <= 8.0 * floor):
            bad.append((k, d_me[k], yard[k], floor))
    rows.sort(reverse=True)
    for ratio, d_gpu, d_y, floor, k in rows[:6]:
        print(f"{name} {'native' if native else 'composite'} {k}: {d_gpu:.2e} from float64 (PyTorch-ROCm fp32: {d_y:.2e}; reference fp32 CPU: {floor:.2e})")
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"{name} {'native' if native else 'composite'}: forward logits {el:.2e} boxes {eb:.2e}; median distance to float64 {med([r[1] for r in rows]):.2e} "
          f"(PyTorch-ROCm {med([r[2] for r in rows]):.2e}, reference fp32 CPU {med([r[3] for r in rows]):.2e})")
    assert not bad, f"{len(bad)} gradients outside the float64 gate: {bad[:6]}"
    for k in g["trainable_without_grad"]:
        p_ = params[str(k)]
        assert p_.grad is None or float(p_.grad.abs().sum()) == 0.0, k


DENSE_CASES = [  # Dd, Hd, Q, layers, F, C, B, N
    (128, 4, 7, 2, 256, 11, 2, 17),
    (128, 4, 7, 2, 256, 11, 3, 257),
    (192, 2, 5, 2, 256, 11, 2, 1370),        # head_dim 96, all 1 370 tokens of a 518x518 image as keys
    (256, 4, 25, 2, 512, 91, 2, 257),        # the --lightweight widths
    (768, 8, 100, 3, 1024, 91, 2, 257),      # config.py:21-35 defaults, untied x3
]


@pytest.mark.parametrize("case", DENSE_CASES, ids=[f"Dd{c[0]}_Q{c[2]}_N{c[7]}" for c in DENSE_CASES])
def test_native_dense_decoder_backward_matches_composite_autograd(case):
    """The nn.TransformerDecoder branch (use_deformable=False, detr_decoder.py:28-35, 62-69) in train() mode on the native kernels
    (dod_dense_decoder_train_*): forward and every gradient -- both attentions' in / out projections, FFN, LayerNorms, heads, the
    query embedding -- and d(memory) against torch's own nn.TransformerDecoder autograd on the same weights (dropout 0)."""
    from dinov2_od_amd.models import DETRDecoder
    from tests import gpu_util as G
    Dd, Hd, Q, layers, F, C, B, N = case
    dc = cases.dec_cfg(False, Dd, Hd, Q, layers, F, C)
    m = DETRDecoder(Q, Dd, Hd, layers, C, dim_feedforward=F, dropout=0.0, use_deformable=False, precision="fp32")
    G.load_np_state(m, synth.decoder_state_dict(dc, seed=1, prefix=""))
    m = m.to(G.dev()).train()
    mem = G.to_gpu(synth.normal(3, f"memory.dense.{N}.{Dd}", (B, N, Dd), 1.0))
    wl = G.to_gpu(synth.normal(5, "dense.wl", (B, Q, C), 1.0))
    wb = G.to_gpu(synth.normal(5, "dense.wb", (B, Q, 4), 1.0))
    l0, b0, dx0, g0 = _run(m, mem, wl, wb, native=False)
    l1, b1, dx1, g1 = _run(m, mem, wl, wb, native=True)
    assert not torch.equal(l0, l1), "both runs took the same path"
    assert rel_err(l1.cpu().numpy(), l0.cpu().numpy()) < 1e-4 and rel_err(b1.cpu().numpy(), b0.cpu().numpy()) < 1e-4
    assert set(g0) == set(g1) and len(g1) == 18 * layers + 7
    worst = ("", 0.0)
    for k in g0:
        e = rel_err(g1[k].cpu().numpy(), g0[k].cpu().numpy())
        worst = max(worst, (k, e), key=lambda t: t[1])
        assert e < 2e-4, (k, e)
    e = rel_err(dx1.cpu().numpy(), dx0.cpu().numpy())
    print(f"native vs torch dense decoder {case}: worst parameter gradient {worst[0]} {worst[1]:.2e}, d(memory) {e:.2e}")
    assert e < 2e-4


def test_native_dense_decoder_dropout_is_seeded_and_consistent():
    """dropout 0.1 at torch's six sites per layer: deterministic in the seed, gone in eval(), and the backward applies the forward's
    masks (central difference of the loss along a random direction of every parameter and of the memory)."""
    from dinov2_od_amd.models import DETRDecoder, _native_train as nt
    from tests import gpu_util as G
    dc = cases.dec_cfg(False, 128, 4, 16, 2, 256, 11)
    m = DETRDecoder(16, 128, 4, 2, 11, dim_feedforward=256, dropout=0.1, use_deformable=False, precision="fp32")
    G.load_np_state(m, synth.decoder_state_dict(dc, seed=1, prefix=""))
    m = m.to(G.dev()).train()
    B, N = 3, 257
    mem = G.to_gpu(synth.normal(3, "memory.dense.drop", (B, N, 128), 1.0))
    assert nt.dense_supported(m, mem) and abs(nt.dense_dropout_rate(m) - 0.1) < 1e-9
    a = nt.dense_decoder_train(m, mem, seed=1234).detach().clone()
    b = nt.dense_decoder_train(m, mem, seed=1234).detach().clone()
    c = nt.dense_decoder_train(m, mem, seed=99).detach().clone()
    assert torch.equal(a, b) and not torch.equal(a, c)
    m.decoder.layers[0].dropout2.p = 0.3                     # sites disagree -> not native (the composite honours the modules)
    assert nt.dense_dropout_rate(m) is None and not nt.dense_supported(m, mem)
    m.decoder.layers[0].dropout2.p = 0.1
    params = [q for q in m.parameters() if q.requires_grad]
    g = torch.Generator(device="cpu").manual_seed(0)
    dirs = [torch.randn(q.shape, generator=g).to(q.device) * (0.03 / max(1.0, q.numel() ** 0.5)) for q in params]
    vx = torch.randn(mem.shape, generator=g).to(mem.device) * (0.03 / mem.numel() ** 0.5)
    wl = G.to_gpu(synth.normal(5, "dense.drop.wl", (B, 16, 15), 1.0))

    def f(x):
        return (nt.dense_decoder_train(m, x, seed=77) * wl).sum()
    x = mem.clone().requires_grad_(True)
    m.zero_grad(set_to_none=True)
    f(x).backward()
    ana = float(sum((q.grad * v).sum() for q, v in zip(params, dirs) if q.grad is not None) + (x.grad * vx).sum())
    with torch.no_grad():
        for q, v in zip(params, dirs):
            q.add_(v)
        lp = float(f(mem + vx))
        for q, v in zip(params, dirs):
            q.sub_(2 * v)
        lm = float(f(mem - vx))
        for q, v in zip(params, dirs):
            q.add_(v)
    num = (lp - lm) / 2
    print(f"dense decoder dropout directional derivative: analytic {ana:.5f}, central difference {num:.5f}")
    assert abs(ana - num) < 2e-2 * abs(num) + 1e-3
    m.eval()
    assert nt.dense_dropout_rate(m) == 0.0
