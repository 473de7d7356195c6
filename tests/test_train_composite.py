"""train() mode of the drop-in modules: the autograd composite (dinov2_od_amd/models/_autograd.py) --
pinned against the reference's own outputs (goldens G2 / G1: with dropout 0 the train-mode forward IS the eval forward),
gradient routing as in the reference (frozen DINOv2 weights, trainable LoRA / projection / decoder), dropout placement active.
Pure torch: runs on the CPU."""
import numpy as np
import pytest
import torch

from dinov2_od_amd import synth
from dinov2_od_amd.models import DINOv2ObjectDetector, DETRDecoder
from tests import cases
from tests.cases import rel_err


@pytest.fixture(autouse=True)
def _composite_on_cpu(monkeypatch):
    """the composite is GPU-only in the product (no CPU fallback); the CPU suite lifts that to pin it against the goldens"""
    monkeypatch.setenv("DINODET_COMPOSITE_ON_CPU", "1")


def _detector(Q, dropout):
    bb, dc = cases.cfg1(Q)
    m = DINOv2ObjectDetector(num_classes=91, dino_model_name="facebook/dinov2-small", lora_r=1, lora_alpha=1.0, hidden_dim=256,
                             num_queries=Q, nheads=4, num_decoder_layers=2, dim_feedforward=512, dropout=dropout, n_points=2,
                             use_deformable=True, pretrained=False)
    sd = {k: torch.from_numpy(v) for k, v in synth.detector_state_dict(bb, dc, seed=1).items()}
    m.load_state_dict(sd, strict=True)
    return m


def test_train_mode_forward_matches_reference_golden():
    g = cases.golden("g2_cfg1_q25")
    m = _detector(25, 0.0).train()
    x = torch.from_numpy(synth.make_pixels(2, 224, 224, seed=0))
    with torch.no_grad():
        o = m(x)
    assert rel_err(o["pred_logits"].numpy(), g["pred_logits"]) < 1e-3
    assert rel_err(o["pred_boxes"].numpy(), g["pred_boxes"]) < 1e-3
    packed = m.forward_packed(x)
    assert packed.shape == (2, 25, 95) and torch.equal(packed[..., :91], m(x)["pred_logits"])


@pytest.mark.parametrize("tag,deform,Dd,Hd,Q,Ns", cases.G1_CASES, ids=[c[0] for c in cases.G1_CASES])
def test_train_mode_decoder_matches_reference_golden(tag, deform, Dd, Hd, Q, Ns):
    g = cases.golden("g1_decoder_only")
    dc = cases.dec_cfg(deform, Dd=Dd, Hd=Hd, Q=Q)
    m = DETRDecoder(num_queries=Q, hidden_dim=Dd, nheads=Hd, num_decoder_layers=dc.num_layers, num_classes=dc.num_classes,
                    dim_feedforward=dc.dim_feedforward, dropout=0.0, n_points=dc.n_points, use_deformable=deform).train()
    sd = {k: torch.from_numpy(v) for k, v in synth.decoder_state_dict(dc, seed=1, prefix="").items()}
    m.load_state_dict(sd, strict=True)
    for N in Ns[:2]:
        with torch.no_grad():
            o = m(torch.from_numpy(cases.g1_memory(N, Dd)))
        assert rel_err(o["pred_logits"].numpy(), g[f"{tag}_N{N}_logits"]) < 1e-4
        assert rel_err(o["pred_boxes"].numpy(), g[f"{tag}_N{N}_boxes"]) < 1e-4


def test_gradients_reach_exactly_the_trainable_subset():
    m = _detector(25, 0.1).train()
    x = torch.from_numpy(synth.make_pixels(1, 70, 70, seed=0))
    o = m(x)
    (o["pred_logits"].square().mean() + o["pred_boxes"].mean()).backward()
    with_grad = {k for k, p in m.named_parameters() if p.grad is not None and float(p.grad.abs().sum()) > 0}
    trainable = {k for k, p in m.named_parameters() if p.requires_grad}
    assert with_grad <= trainable
    assert any("lora_A" in k for k in with_grad) and any("lora_B" in k for k in with_grad)      # through the last two blocks
    assert "backbone.projection.weight" in with_grad and "decoder.class_embed.weight" in with_grad
    assert "decoder.decoder.layers.0.cross_attn.sampling_offsets.weight" in with_grad             # through the bilinear weights
    frozen = [k for k, p in m.named_parameters() if not p.requires_grad]
    assert frozen and all(m.get_parameter(k).grad is None for k in frozen)
    # an optimizer step changes the parameters the engine will re-pack on the next eval forward (version counters move)
    p = m.get_parameter("decoder.class_embed.weight")
    v0 = p._version
    torch.optim.SGD([q for q in m.parameters() if q.requires_grad], lr=0.1).step()
    assert p._version > v0


def test_composite_refuses_cpu_tensors_without_the_test_switch(monkeypatch):
    monkeypatch.delenv("DINODET_COMPOSITE_ON_CPU")
    m = _detector(25, 0.0).train()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 70, 70))


def test_dropout_is_active_in_train_mode_only():
    m = _detector(25, 0.5).train()
    x = torch.from_numpy(synth.make_pixels(1, 70, 70, seed=0))
    with torch.no_grad():
        torch.manual_seed(0)
        a = m(x)["pred_logits"]
        torch.manual_seed(1)
        b = m(x)["pred_logits"]
    assert not torch.allclose(a, b)


@pytest.mark.parametrize("name", sorted(cases.G9_CASES))
def test_composite_gradients_match_the_reference_backward(name):
    """G9: the .grad of every trainable parameter after the REFERENCE's own train()-mode forward + backward (dropout 0, fixed linear
    loss; tests/golden/make_goldens.py::g9_gradients) against the autograd composite's -- the backward of row f1 pinned to what
    train.py:1101 computes, not to a restatement."""
    g = cases.golden(name)
    model_name, R, B, kw = cases.G9_CASES[name]
    m = DINOv2ObjectDetector(dino_model_name=model_name, pretrained=False, **kw)
    sd = {k: torch.from_numpy(v) for k, v in synth.detector_state_dict(m._bb_cfg, m._dc_cfg, seed=1).items()}
    m.load_state_dict(sd, strict=True)
    m.train()
    o = m(torch.from_numpy(synth.make_pixels(B, R, R, seed=0)))
    gl, gb = cases.g9_loss_weights(B, m._dc_cfg.num_queries, m._dc_cfg.num_classes)
    assert rel_err(o["pred_logits"].detach().numpy(), g["pred_logits"]) < 1e-3
    loss = (o["pred_logits"] * torch.from_numpy(gl)).sum() + (o["pred_boxes"] * torch.from_numpy(gb)).sum()
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-3 * max(1.0, abs(float(g["loss"])))
    worst = cases.g9_check(m, g, 2e-4, 2e-4)
    print(f"{name}: worst gradient probe error {worst[0]:.2e} ({worst[1]})")
