"""`-m gpu`: the whole hot path through the drop-in modules / C ABI.

fp32 ("strict") mode is the mode gated at the north-star tolerance: logits and boxes within 1e-3
relative (max|a-b| / max|b|) of the REFERENCE's own fp32 CPU forward (committed golden vectors) and of
the CPU oracle, stage by stage.
bf16 mode (the throughput mode) cannot meet 1e-3 on logits by construction (8-bit mantissa operands;
DESIGN.md "precision modes"): it is held to the oracle evaluated with the same bf16 operand rounding,
and its distance to the fp32 reference is bounded and reported."""
import os

import numpy as np
import pytest
import torch

from dinov2_od_amd import _native as nat, synth
from dinov2_od_amd.config import num_tokens
from oracle import dinodet_oracle as orc
from tests import cases
from tests.cases import rel_err, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-3      # BASELINE.json north_star: "within 1e-3 relative fp32 tolerance (box coords and class logits)"


@pytest.fixture(scope="module")
def G():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from tests import gpu_util
    return gpu_util


def _taps(m, B, N, bb, dc):
    eng = m._get_engine()
    t = {"emb": eng.set_tap(0, (B, N, bb.hidden), "cuda:0")}
    for i in range(bb.layers):
        t[f"block{i}"] = eng.set_tap(1 + i, (B, N, bb.hidden), "cuda:0")
    t["mem"] = eng.set_tap(1000, (B, N, dc.hidden_dim), "cuda:0")
    for j in range(dc.num_layers):
        t[f"dec{j}"] = eng.set_tap(3000 + j, (B, dc.num_queries, dc.hidden_dim), "cuda:0")
    return t


K_BF16 = 1.3    # HIP-vs-fp32 distance allowed, in units of the bf16-faithful oracle's own distance to fp32

# parity-gated modes: "fp32" (exact-fp32 MFMA / VALU), "bf16x3" (split products on the bf16 MFMA kernels, ~1e-5 per linear) and
# "fp16x2" (fp16 main product + e4m3 cross terms for the block linears, ~2e-5 per linear; everything else as bf16x3)
GATED = ["fp32", "bf16x3", "fp16x2"]
# per-stage bounds (max-relative) of the micro models / the ViT-B feature probe / the full-depth probes: the compensated modes carry
# 16 (bf16x3) and ~15.5 (fp16x2: e4m3 cross terms) significant bits per product instead of 24
STAGE = {"fp32": 1e-5, "bf16x3": 5e-5, "fp16x2": 1e-4}
FEAT = {"fp32": 1e-4, "bf16x3": 1e-4, "fp16x2": 1.5e-4}
DEEP = {"fp32": 2e-5, "bf16x3": 2e-4, "fp16x2": 4e-4}


@pytest.mark.parametrize("precision", GATED)
@pytest.mark.parametrize("swiglu", [False, True])
@pytest.mark.parametrize("R", [70, 56])
def test_strict_micro_backbone_every_stage_vs_reference(G, swiglu, R, precision):
    """G0/G4 goldens: embeddings (incl. bicubic 5->4 at 56x56), each block, final features."""
    from dinov2_od_amd.models import DINOv2Backbone
    if swiglu and precision in ("bf16x3", "fp16x2"):
        pytest.skip("the micro SwiGLU width (344) is not a multiple of 32: the MFMA K-tiles of the compensated modes need that")
    g = cases.golden("g4_micro_swiglu" if swiglu else "g0_micro_backbone")
    bb = cases.micro_bb(swiglu)
    m = DINOv2Backbone("micro", lora_r=2, lora_alpha=1.0, target_dim=None, pretrained=False, precision=precision, config=bb)
    G.load_np_state(m, synth.backbone_state_dict(bb, seed=1, prefix=""))
    m = m.to(G.dev()).eval()
    x = G.to_gpu(synth.make_pixels(2, R, R, seed=0))
    N = num_tokens(R, R)
    eng = m._get_engine()
    emb = eng.set_tap(0, (2, N, bb.hidden), "cuda:0")
    blocks = [eng.set_tap(1 + i, (2, N, bb.hidden), "cuda:0") for i in range(bb.layers)]
    f = m(x)
    G.sync()
    stage_tol = STAGE[precision]
    assert rel_err(emb.cpu().numpy(), g[f"embeddings_{R}"]) < 1e-5
    for i, b in enumerate(blocks):
        assert rel_err(b.cpu().numpy(), g[f"block{i}_{R}"]) < stage_tol, i
    assert rel_err(f.cpu().numpy(), g[f"features_{R}"]) < stage_tol


@pytest.mark.parametrize("case", cases.G1_CASES, ids=[c[0] for c in cases.G1_CASES])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "fp16x2", "bf16"])
def test_decoder_only_vs_reference(G, case, precision):
    """G1 goldens: DETRDecoder on random memory, deformable and dense branches, head dims 32 and 96,
    N in {17, 26, 257, 1370} (pins the (h,w) factorisation quirk)."""
    from dinov2_od_amd.models import DETRDecoder
    tag, deform, Dd, Hd, Q, Ns = case
    g = cases.golden("g1_decoder_only")
    dc = cases.dec_cfg(deform, Dd, Hd, Q)
    m = DETRDecoder(Q, Dd, Hd, dc.num_layers, dc.num_classes, dim_feedforward=dc.dim_feedforward, n_points=dc.n_points,
                    use_deformable=deform, precision=precision)
    G.load_np_state(m, synth.decoder_state_dict(dc, seed=1, prefix=""))
    m = m.to(G.dev()).eval()
    for N in Ns:
        out = m(G.to_gpu(cases.g1_memory(N, Dd)))
        G.sync()
        # bf16 mode rounds the memory and value_proj/kv weights to bf16 -> bounded, looser
        tol = TOL if precision in GATED else 3e-2
        assert rel_err(out["pred_logits"].cpu().numpy(), g[f"{tag}_N{N}_logits"]) < tol, (tag, N)
        assert rel_err(out["pred_boxes"].cpu().numpy(), g[f"{tag}_N{N}_boxes"]) < tol, (tag, N)


@pytest.mark.parametrize("precision", GATED)
@pytest.mark.parametrize("Q", [25, 100])
def test_strict_cfg1_end_to_end_vs_reference(G, Q, precision):
    """BASELINE.json configs[0]: --lightweight ViT-S/14 224x224, batch 2."""
    g = cases.golden(f"g2_cfg1_q{Q}")
    bb, dc = cases.cfg1(Q)
    m = G.make_detector(bb, dc, precision, "facebook/dinov2-small")
    out = m(G.to_gpu(synth.make_pixels(2, 224, 224, seed=0)))
    G.sync()
    assert out["pred_logits"].shape == (2, Q, 91) and out["pred_boxes"].shape == (2, Q, 4)
    assert rel_err(out["pred_logits"].cpu().numpy(), g["pred_logits"]) < TOL
    assert rel_err(out["pred_boxes"].cpu().numpy(), g["pred_boxes"]) < TOL


@pytest.mark.parametrize("precision", GATED)
@pytest.mark.parametrize("name,R,deform", [("g3_vitb_224", 224, True), ("g3_vitb_518", 518, True), ("g3_vitb_224_dense", 224, False)])
def test_strict_vitb_end_to_end_vs_reference(G, name, R, deform, precision):
    """ViT-B/14 at 224 (bicubic pos resize, (h,w)=(1,257)) and 518 (N=1370, (10,137)); Q=100."""
    g = cases.golden(name)
    bb, dc = cases.vitb(100, deform)
    m = G.make_detector(bb, dc, precision, "facebook/dinov2-base")
    x = synth.make_pixels(1, R, R, seed=0)
    N = num_tokens(R, R)
    eng = m._get_engine()
    mem = eng.set_tap(1000, (1, N, 768), "cuda:0")
    out = m(G.to_gpu(x))
    G.sync()
    f = mem.cpu().numpy()
    assert rel_err(f[:, ::max(1, N // 8), :64], g["feat_probe"]) < FEAT[precision]
    assert rel_err(out["pred_logits"].cpu().numpy(), g["pred_logits"]) < TOL
    assert rel_err(out["pred_boxes"].cpu().numpy(), g["pred_boxes"]) < TOL


def test_strict_batch_is_independent_and_deterministic(G):
    """Images are independent (no cross-image op): batch of 3 == three batches of 1, bit for bit;
    two runs agree bit for bit (no atomics in the path)."""
    bb, dc = cases.cfg1(25)
    m = G.make_detector(bb, dc, "fp32", "facebook/dinov2-small")
    x = G.to_gpu(synth.make_pixels(3, 224, 224, seed=0))
    a = m.forward_packed(x).clone()
    b = m.forward_packed(x).clone()
    assert torch.equal(a, b)
    for i in range(3):
        one = m.forward_packed(x[i:i + 1])
        assert torch.equal(one[0], a[i]), i


@pytest.mark.parametrize("R,B", [(224, 2), (518, 1)])
def test_bf16_vitb_vs_bf16_faithful_oracle_and_fp32_reference(G, R, B):
    """Throughput mode, held to the oracle evaluated with the SAME bf16 operand rounding:
    embeddings and block 0 tightly (kernel correctness: same rounding points, only accumulation order
    and flash-vs-global-max P rounding differ).  Deeper stages of two bf16 evaluations decorrelate up
    to the bf16 quantisation level itself (a perturbation d re-enters every rounding as
    sqrt(d * 2^-8)), so the final features are bounded against both the bf16-faithful and the fp32
    oracle, and the outputs' distance to the fp32 reference is bounded and printed."""
    bb, dc = cases.vitb(100, True)
    sd = synth.detector_state_dict(bb, dc, seed=1)
    x = synth.make_pixels(B, R, R, seed=0)
    N = num_tokens(R, R)
    m = G.make_detector(bb, dc, "bf16", "facebook/dinov2-base")
    eng = m._get_engine()
    emb = eng.set_tap(0, (B, N, 768), "cuda:0")
    blk0 = eng.set_tap(1, (B, N, 768), "cuda:0")
    mem = eng.set_tap(1000, (B, N, 768), "cuda:0")
    out = m(G.to_gpu(x))
    G.sync()
    taps = {}
    emu = orc.detector_forward(sd, bb, dc, x, emulate_bf16=True, taps=taps)
    f32 = orc.detector_forward(sd, bb, dc, x)
    assert rel_err(emb.cpu().numpy(), taps["embeddings"].numpy()) < 1e-5
    # one bf16 ulp is 2^-8 = 3.9e-3 relative: a handful of operand roundings flip between the two evaluations
    assert rel_err(blk0.cpu().numpy(), taps["block0"].numpy()) < 4e-3
    # a difference d entering a bf16 re-quantisation leaves it as ~sqrt(d * 2^-8): 1e-4 (flash-vs-global-max P
    # rounding) -> 6e-4 (ctx) -> ~1.5e-3 after the MLP's two roundings
    assert rel_l2(blk0.cpu().numpy(), taps["block0"].numpy()) < 3e-3
    f = mem.cpu().numpy()
    e_emu = rel_l2(f, emu["features"].numpy())
    e_f32 = rel_l2(f, f32["features"].numpy())
    print(f"bf16 features R={R}: rel-L2 vs bf16-faithful oracle {e_emu:.2e}, vs fp32 oracle {e_f32:.2e}")
    assert e_emu < 8e-3 and e_f32 < 8e-3
    el, eb = rel_l2(out["pred_logits"].cpu().numpy(), f32["pred_logits"].numpy()), rel_l2(out["pred_boxes"].cpu().numpy(), f32["pred_boxes"].numpy())
    # the yardstick is what bf16 operand rounding itself costs: the bf16-faithful oracle's own distance to the fp32 oracle.
    # The HIP path must not be further away than that (x K_BF16: two evaluations with the same rounding points are two
    # draws of the same error, not the same draw), so a kernel regression that costs accuracy shows up here.
    ol, ob = rel_l2(emu["pred_logits"].numpy(), f32["pred_logits"].numpy()), rel_l2(emu["pred_boxes"].numpy(), f32["pred_boxes"].numpy())
    print(f"bf16 outputs R={R}: rel-L2 vs fp32 oracle: logits {el:.2e} (bf16-faithful oracle {ol:.2e}), boxes {eb:.2e} ({ob:.2e})")
    assert e_f32 < K_BF16 * rel_l2(emu["features"].numpy(), f32["features"].numpy())
    assert el < K_BF16 * ol and eb < K_BF16 * ob


def test_bf16_cfg1_end_to_end(G):
    g = cases.golden("g2_cfg1_q100")
    bb, dc = cases.cfg1(100)
    m = G.make_detector(bb, dc, "bf16", "facebook/dinov2-small")
    x = synth.make_pixels(2, 224, 224, seed=0)
    out = m(G.to_gpu(x))
    G.sync()
    emu = orc.detector_forward(synth.detector_state_dict(bb, dc, seed=1), bb, dc, x, emulate_bf16=True)
    for k in ("pred_logits", "pred_boxes"):
        got, floor = rel_l2(out[k].cpu().numpy(), g[k]), rel_l2(emu[k].numpy(), g[k])
        print(f"bf16 cfg1 {k}: rel-L2 vs the reference {got:.2e} (bf16-faithful oracle {floor:.2e})")
        assert got < K_BF16 * floor, (k, got, floor)


def test_weights_resync_after_load_state_dict(G):
    """checkpoint-resume path of train.py:695-739: load_state_dict after a forward must take effect."""
    bb, dc = cases.cfg1(25)
    m = G.make_detector(bb, dc, "fp32", "facebook/dinov2-small")
    x = G.to_gpu(synth.make_pixels(1, 224, 224, seed=0))
    a = m.forward_packed(x).clone()
    sd2 = synth.detector_state_dict(bb, dc, seed=2)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()}, strict=True)
    b = m.forward_packed(x).clone()
    assert not torch.allclose(a, b)
    want = orc.detector_forward(sd2, bb, dc, x.cpu().numpy())
    assert rel_err(b[..., :91].cpu().numpy(), want["pred_logits"].numpy()) < TOL


def test_train_step_then_native_eval(G):
    """train.py:1079-1101 on the drop-in: train() forward is the autograd composite on the GPU, backward + SGD step move the
    trainable subset, and the next eval() forward -- native kernels -- runs with the updated weights (automatic re-pack)."""
    bb, dc = cases.cfg1(25)
    m = G.make_detector(bb, dc, "fp32", "facebook/dinov2-small")
    x = G.to_gpu(synth.make_pixels(2, 224, 224, seed=0))
    before = m.forward_packed(x).clone()                      # native
    m.train()
    o = m(x)                                                  # composite, with dropout (config default 0.1)
    assert o["pred_logits"].requires_grad and o["pred_logits"].is_cuda
    loss = o["pred_logits"].square().mean() + (o["pred_boxes"] - 0.5).abs().mean()
    loss.backward()
    opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.05)
    opt.step()
    m.eval()
    after = m.forward_packed(x).clone()                       # native again, re-packed weights
    assert not torch.allclose(before, after)
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    want = orc.detector_forward(sd, bb, dc, x.cpu().numpy())
    assert rel_err(after[..., :91].cpu().numpy(), want["pred_logits"].numpy()) < TOL
    assert rel_err(after[..., 91:].cpu().numpy(), want["pred_boxes"].numpy()) < TOL
    # the composite in eval-equivalent form (no dropout) agrees with the native kernels on the same weights
    m.train()
    G.no_dropout(m)
    with torch.no_grad():
        comp = m(x)
    assert rel_err(comp["pred_logits"].cpu().numpy(), after[..., :91].cpu().numpy()) < TOL


def test_train_forward_runs_the_frozen_prefix_natively(G):
    """train() on the GPU: embeddings + the blocks before the first LoRA-adapted one come from dod_backbone_prefix (no autograd
    needed there), the rest from the autograd composite -- same outputs and same gradients as the all-composite evaluation
    (selected here by an input that requires grad)."""
    bb, dc = cases.cfg1(25)
    m = G.make_detector(bb, dc, "fp32", "facebook/dinov2-small")
    m.train()
    G.no_dropout(m)
    x = G.to_gpu(synth.make_pixels(2, 224, 224, seed=0))
    eng = m.backbone._get_engine()
    pre = eng.backbone_prefix(x, m.backbone._engine_named(), bb.layers - 2)
    assert pre.shape == (2, num_tokens(224, 224), bb.hidden) and not pre.requires_grad

    def run(inp):
        m.zero_grad(set_to_none=True)
        o = m(inp)
        (o["pred_logits"].square().mean() + o["pred_boxes"].mean()).backward()
        g = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        return o["pred_logits"].detach().clone(), g

    hybrid, gh = run(x)
    full, gf = run(x.clone().requires_grad_(True))
    assert rel_err(hybrid.cpu().numpy(), full.cpu().numpy()) < 1e-4
    assert set(gh) == set(gf) and any("lora_A" in k for k in gh)
    for k in gh:
        assert rel_err(gh[k].cpu().numpy(), gf[k].cpu().numpy()) < 2e-3, k


def test_hipgraph_replay_matches_eager_and_tracks_weight_updates(G):
    """model.enable_hipgraph(): one captured graph per input shape, bit-identical to the eager launches, re-captured after a
    weight update (the packed weights move), several shapes side by side."""
    bb, dc = cases.cfg1(25)
    m = G.make_detector(bb, dc, "bf16", "facebook/dinov2-small")
    xa = G.to_gpu(synth.make_pixels(2, 224, 224, seed=0))
    xb = G.to_gpu(synth.make_pixels(1, 70, 112, seed=3))
    ea, eb = m.forward_packed(xa).clone(), m.forward_packed(xb).clone()
    m.enable_hipgraph()
    for _ in range(2):                                   # first call captures, second replays
        assert torch.equal(m.forward_packed(xa), ea)
        assert torch.equal(m.forward_packed(xb), eb)
    xa2 = G.to_gpu(synth.make_pixels(2, 224, 224, seed=5))
    m.enable_hipgraph(False)
    ea2 = m.forward_packed(xa2).clone()
    m.enable_hipgraph()
    assert torch.equal(m.forward_packed(xa2), ea2)       # new input through the static buffer
    sd2 = synth.detector_state_dict(bb, dc, seed=2)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()}, strict=True)
    g2 = m.forward_packed(xa).clone()
    m.enable_hipgraph(False)
    assert torch.equal(m.forward_packed(xa), g2) and not torch.equal(g2, ea)


@pytest.mark.parametrize("precision", ["bf16", "bf16x3", "fp16x2"])
def test_concurrent_micro_batches_are_bit_identical(G, precision):
    """engine.micro_streams: a large batch runs as concurrent micro-batches on separate HIP streams (fork / join), eagerly, inside the
    engine's own captured graph and inside a caller's capture (bench.py).  Images are independent: the detections must equal the
    single-launch forward's bit for bit."""
    bb, dc = cases.cfg1(25)
    m = G.make_detector(bb, dc, precision, "facebook/dinov2-small")
    x = G.to_gpu(synth.make_pixels(4, 224, 224, seed=0))
    eng = m._get_engine()
    eng.micro_streams = 1
    single = m.forward_packed(x).clone()
    eng.micro_streams, eng.micro_min_batch, eng.micro_min_rows = 2, 2, 0
    assert torch.equal(m.forward_packed(x), single)                      # eager fork / join
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):         # a caller's capture: the side streams join it
        out = m.forward_packed(x)
    g.replay()
    G.sync()
    assert torch.equal(out, single)
    m.enable_hipgraph()
    for _ in range(2):                                                   # the engine's own graph: capture, then replay
        assert torch.equal(m.forward_packed(x), single)
    m.enable_hipgraph(False)


def test_first_forward_at_a_resized_grid_through_the_micro_path(G):
    """the (H, W) position table is built by the first forward at a new grid (hipMalloc + resize kernel on the launch stream).  When
    that first forward is split into concurrent micro-batches, the table must be ordered before the fork: the side stream's patch
    embed reads it (engine._launch_micro calls dod_prepare on the current stream first).  Fresh engines, non-native grids."""
    bb, dc = cases.cfg1(25)
    for H, W in ((112, 168), (224, 224), (70, 98)):
        x = G.to_gpu(synth.make_pixels(6, H, W, seed=4))
        ref = G.make_detector(bb, dc, "bf16x3", "facebook/dinov2-small")
        ref._get_engine().micro_streams = 1
        want = ref.forward_packed(x).clone()
        m = G.make_detector(bb, dc, "bf16x3", "facebook/dinov2-small")          # cold position cache
        eng = m._get_engine()
        eng.micro_streams, eng.micro_min_batch, eng.micro_min_rows = 2, 2, 0
        got = m.forward_packed(x)                                                # FIRST call at (H, W): micro path
        G.sync()
        assert torch.equal(got, want), (H, W)


def test_hipgraph_small_shape_survives_a_later_larger_one(G):
    """a captured graph bakes in the address of its workspace: a later, larger shape (or an eager call that needs more scratch)
    must not pull it from under the earlier graph.  Capture small, then large, run eager calls of other shapes, replay small."""
    bb, dc = cases.cfg1(25)
    m = G.make_detector(bb, dc, "bf16", "facebook/dinov2-small")
    xs = G.to_gpu(synth.make_pixels(1, 70, 112, seed=3))
    xl = G.to_gpu(synth.make_pixels(3, 224, 224, seed=0))
    m.enable_hipgraph()
    a_small = m.forward_packed(xs).clone()               # captures the small shape FIRST (workspace sized for it)
    a_large = m.forward_packed(xl).clone()               # larger capture
    m.enable_hipgraph(False)
    e_small, e_large = m.forward_packed(xs).clone(), m.forward_packed(xl).clone()     # eager: grows the shared eager workspace
    junk = [torch.full((1 << 22,), float("nan"), device=G.dev()) for _ in range(8)]   # reuse whatever the allocator freed
    m.enable_hipgraph()
    for _ in range(2):
        assert torch.equal(m.forward_packed(xs), e_small) and torch.equal(a_small, e_small)
        assert torch.equal(m.forward_packed(xl), e_large) and torch.equal(a_large, e_large)
    del junk


def test_alternating_input_sizes_reuse_position_tables(G):
    """one interpolated position table per distinct (H, W), cached in the handle: alternating sizes stay bit-identical and
    allocate nothing after the first visit"""
    bb, dc = cases.cfg1(25)
    m = G.make_detector(bb, dc, "fp32", "facebook/dinov2-small")
    xa = G.to_gpu(synth.make_pixels(1, 70, 112, seed=3))
    xb = G.to_gpu(synth.make_pixels(1, 112, 70, seed=3))
    ra, rb = m.forward_packed(xa).clone(), m.forward_packed(xb).clone()
    G.sync()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(20):
        assert torch.equal(m.forward_packed(xa), ra) and torch.equal(m.forward_packed(xb), rb)
    G.sync()
    assert torch.cuda.mem_get_info()[0] >= free0 - (1 << 20)      # no per-forward hipMalloc


@pytest.mark.parametrize("precision", GATED + ["bf16"])
def test_degenerate_images(G, precision):
    """constant images (all zeros / all ones: identical patches, attention rows uniform) and a mixed batch: finite everywhere,
    gated modes within the 1e-3 gate of the oracle, batch entries independent of their neighbours"""
    bb, dc = cases.cfg1(25)
    m = G.make_detector(bb, dc, precision, "facebook/dinov2-small")
    x = np.zeros((3, 3, 224, 224), np.float32)
    x[1] = 1.0
    x[2] = synth.make_pixels(1, 224, 224, seed=0)[0]
    out = m.forward_packed(G.to_gpu(x)).clone()
    assert torch.isfinite(out).all()
    single = m.forward_packed(G.to_gpu(x[1:2])).clone()
    assert torch.equal(single[0], out[1]) or rel_err(single[0].cpu().numpy(), out[1].cpu().numpy()) < 1e-6
    if precision in GATED:
        sd = synth.detector_state_dict(bb, dc, seed=1)
        want = orc.detector_forward(sd, bb, dc, x)
        assert rel_err(out[..., :91].cpu().numpy(), want["pred_logits"].numpy()) < TOL
        assert rel_err(out[..., 91:].cpu().numpy(), want["pred_boxes"].numpy()) < TOL


def test_module_prefix_and_error_paths(G):
    from dinov2_od_amd import _native as nat
    bb, dc = cases.cfg1(25)
    m = G.make_detector(bb, dc, "fp32", "facebook/dinov2-small")
    with pytest.raises(ValueError, match="channel dimension"):
        m(torch.zeros(1, 4, 224, 224, device=G.dev()))
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 8, 8, device=G.dev()))          # smaller than one patch
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 224, 224))


def test_bf16_decoder_split3_linears_large_batch(G):
    """bf16 mode, B*Q >= 1024: the decoder's query-side linears run as bf16x3-split GEMMs (A = Ah+Al, W = Wh+Wl,
    lo*lo dropped) on the bf16 MFMA kernel instead of the fp32 MFMA kernel.  Held to the oracle evaluated with the
    same bf16 rounding of the memory-side operands; the split itself is ~1e-5 relative."""
    from dinov2_od_amd.models import DETRDecoder
    B, Q, Dd, Hd, N = 12, 100, 128, 4, 257
    dc = cases.dec_cfg(True, Dd, Hd, Q)
    sd = synth.decoder_state_dict(dc, seed=1, prefix="decoder.")
    mem = synth.normal(3, "memory.split3", (B, N, Dd), 1.0)
    want_l, want_b = orc.decoder_forward(sd, dc, torch.from_numpy(mem).to(torch.bfloat16).float(), emulate_bf16=True)
    outs = {}
    for prec in ("bf16", "fp32"):
        m = DETRDecoder(Q, Dd, Hd, dc.num_layers, dc.num_classes, dim_feedforward=dc.dim_feedforward, n_points=dc.n_points,
                        use_deformable=True, precision=prec)
        G.load_np_state(m, {k[len("decoder."):]: v for k, v in sd.items()})
        m = m.to(G.dev()).eval()
        outs[prec] = m(G.to_gpu(mem))
        G.sync()
    assert rel_err(outs["bf16"]["pred_logits"].cpu().numpy(), want_l.numpy()) < 5e-3
    assert rel_err(outs["bf16"]["pred_boxes"].cpu().numpy(), want_b.numpy()) < 5e-3
    # and close to the all-fp32 path (difference = bf16 rounding of memory / value_proj weights, amplified by the decoder)
    assert rel_err(outs["bf16"]["pred_logits"].cpu().numpy(), outs["fp32"]["pred_logits"].cpu().numpy()) < 3e-2


@pytest.mark.parametrize("deform,Hd", [(True, 4), (False, 4), (False, 2)])       # head_dim 64: fp32-MFMA attention kernel; 128: the VALU one
@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_decoder_fused_operand_split_is_bit_identical(G, deform, Hd, precision):
    """The producers of the query-side linears' operands (LayerNorm, the fp32 attention / sampling kernels, linear1's ReLU epilogue) write
    the bf16x3 operand [hi | hi | lo] themselves instead of a split3 launch per linear: same roundings, so the same bits -- both decoder
    branches, the box head's reader of the last LayerNorm included (B*Q >= 1024 rows take the split form)."""
    from dinov2_od_amd import _native as nat
    from dinov2_od_amd.models import DETRDecoder
    B, Q, Dd, N = 11, 100, 256, 257
    dc = cases.dec_cfg(deform, Dd, Hd, Q)
    sd = synth.decoder_state_dict(dc, seed=5, prefix="decoder.")
    mem = synth.normal(3, "memory.fused3", (B, N, Dd), 1.0)
    m = DETRDecoder(Q, Dd, Hd, dc.num_layers, dc.num_classes, dim_feedforward=dc.dim_feedforward, n_points=dc.n_points,
                    use_deformable=deform, precision=precision)
    G.load_np_state(m, {k[len("decoder."):]: v for k, v in sd.items()})
    m = m.to(G.dev()).eval()
    outs = {}
    try:
        for mode in (0, 1):
            nat.set_option("dec_fused_split", mode)
            o = m(G.to_gpu(mem))
            G.sync()
            outs[mode] = {k: v.cpu().numpy().copy() for k, v in o.items()}
    finally:
        nat.set_option("dec_fused_split", -1)
    for k in ("pred_logits", "pred_boxes"):
        assert np.array_equal(outs[0][k], outs[1][k]), k
    want_l, want_b = orc.decoder_forward(sd, dc, torch.from_numpy(mem).to(torch.bfloat16).float() if precision == "bf16" else torch.from_numpy(mem),
                                         emulate_bf16=precision == "bf16")
    assert rel_err(outs[1]["pred_logits"], want_l.numpy()) < (5e-3 if precision == "bf16" else TOL)
    assert rel_err(outs[1]["pred_boxes"], want_b.numpy()) < (5e-3 if precision == "bf16" else TOL)


@pytest.mark.parametrize("variant", ["large", "giant"])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "fp16x2", "bf16"])
def test_large_and_giant_shaped_models_two_blocks(G, variant, precision):
    """BASELINE configs[3]/[4] shapes at reduced depth (2 encoder blocks): ViT-L (hidden 1024, 16 heads) and ViT-g
    (hidden 1536, 24 heads, SwiGLU 4096) with the 768-wide projection and a 300-query decoder, 224x224 input.
    fp32 mode against the fp32 oracle at the north-star tolerance; bf16 mode against the bf16-operand oracle."""
    from dinov2_od_amd.config import BackboneConfig, DecoderConfig
    hidden, heads, swiglu = (1024, 16, False) if variant == "large" else (1536, 24, True)
    bb = BackboneConfig(hidden=hidden, layers=2, heads=heads, swiglu=swiglu, lora_r=2, lora_alpha=1.0, target_dim=768)
    dc = DecoderConfig(num_queries=300, hidden_dim=768, nheads=8, num_layers=3, num_classes=91, dim_feedforward=1024,
                       n_points=2, use_deformable=True)
    sd = synth.detector_state_dict(bb, dc, seed=1)
    x = synth.make_pixels(2, 224, 224, seed=0)
    m = G.make_detector(bb, dc, precision, f"facebook/dinov2-{variant}")
    N = num_tokens(224, 224)
    mem = m._get_engine().set_tap(1000, (2, N, 768), "cuda:0")
    out = m(G.to_gpu(x))
    G.sync()
    assert out["pred_logits"].shape == (2, 300, 91)
    if precision in GATED:
        want = orc.detector_forward(sd, bb, dc, x)
        assert rel_err(mem.cpu().numpy(), want["features"].numpy()) < FEAT[precision]
        # 300 queries on the (1, 257) grid of a 224x224 input: each decoder layer amplifies a perturbation ~10x
        # (a reference-point error d moves a sample by 256 d tokens), so two fp32 evaluations of the SAME arithmetic
        # differ by up to 2e-3 on the logits here.  Criterion: the HIP result is as close to the exact (fp64) result
        # as the fp32 CPU evaluation is, within a factor 3 (and within the 1e-3 gate whenever that one is).
        exact = orc.detector_forward(sd, bb, dc, x, dtype=torch.float64)
        for k in ("pred_logits", "pred_boxes"):
            floor = rel_err(want[k].numpy(), exact[k].numpy())
            got = rel_err(out[k].cpu().numpy(), exact[k].numpy())
            # bf16x3 carries 16 mantissa bits per operand (features 3e-5 instead of 1e-6): one more factor on this config
            assert got < max(TOL, {"fp32": 3.0, "bf16x3": 5.0, "fp16x2": 8.0}[precision] * floor), (k, got, floor)
    else:
        want = orc.detector_forward(sd, bb, dc, x, emulate_bf16=True)
        exact = orc.detector_forward(sd, bb, dc, x)
        assert rel_l2(mem.cpu().numpy(), want["features"].numpy()) < 8e-3
        # two bf16 evaluations decorrelate to the bf16 noise level on the features and this decoder config amplifies that
        # ~10x per layer (see the fp32 branch): the HIP path may be as far from fp32 as the bf16-faithful oracle is (x K_BF16)
        for k, hip in (("features", mem), ("pred_logits", out["pred_logits"]), ("pred_boxes", out["pred_boxes"])):
            got, floor = rel_l2(hip.cpu().numpy(), exact[k].numpy()), rel_l2(want[k].numpy(), exact[k].numpy())
            print(f"bf16 {variant} {k}: rel-L2 vs fp32 oracle {got:.2e} (bf16-faithful oracle {floor:.2e})")
            assert got < K_BF16 * floor, (k, got, floor)


def test_non_square_and_odd_sizes_strict(G):
    """H != W (position table is bicubic-resized even when the patch count matches, modeling_dinov2.py:71-72),
    sizes that are not multiples of 14 (the conv drops the remainder), batch 3."""
    bb = cases.micro_bb(False)
    from dinov2_od_amd.models import DINOv2Backbone
    m = DINOv2Backbone("micro", lora_r=2, lora_alpha=1.0, target_dim=None, pretrained=False, precision="fp32", config=bb)
    sd = synth.backbone_state_dict(bb, seed=1, prefix="")
    G.load_np_state(m, sd)
    m = m.to(G.dev()).eval()
    sdp = {"backbone." + k: v for k, v in sd.items()}
    for (H, W) in ((70, 84), (75, 61), (140, 35)):
        x = synth.make_pixels(3, H, W, seed=4)
        f = m(G.to_gpu(x))
        G.sync()
        want = orc.backbone_forward(sdp, bb, x)
        assert f.shape == want.shape
        assert rel_err(f.cpu().numpy(), want.numpy()) < 1e-5, (H, W)


# ------------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] / configs[4] at their workload: full depth, 518x518, 300 queries (goldens G7 / G8 from the reference itself)
_FULL_SD = {}


def _full_detector(G, variant, precision):
    """the synthetic state dict of a 0.3 / 1.1 G-parameter model takes a while to hash: generated once per variant"""
    from dinov2_od_amd.models import DINOv2ObjectDetector
    bb, dc = cases.vit_full(variant)
    if variant not in _FULL_SD:
        _FULL_SD.clear()                                 # one variant resident at a time (ViT-g: 4.5 GB of fp32)
        _FULL_SD[variant] = synth.detector_state_dict(bb, dc, seed=1)
    m = DINOv2ObjectDetector(num_classes=dc.num_classes, dino_model_name=f"facebook/dinov2-{variant}", lora_r=bb.lora_r,
                             lora_alpha=bb.lora_alpha, hidden_dim=dc.hidden_dim, num_queries=dc.num_queries, nheads=dc.nheads,
                             num_decoder_layers=dc.num_layers, dim_feedforward=dc.dim_feedforward, n_points=dc.n_points,
                             use_deformable=True, pretrained=False, precision=precision, backbone_config=bb)
    G.load_np_state(m, _FULL_SD[variant])
    return m.to(G.dev()).eval(), bb, dc


def _full_run(G, variant, precision):
    name, blocks = cases.FULL_DEPTH[variant]
    m, bb, dc = _full_detector(G, variant, precision)
    N = num_tokens(518, 518)
    eng = m._get_engine()
    taps = {"embeddings": eng.set_tap(0, (1, N, bb.hidden), "cuda:0"), "feat": eng.set_tap(1000, (1, N, 768), "cuda:0")}
    for b in blocks:
        taps[f"block{b}"] = eng.set_tap(1 + b, (1, N, bb.hidden), "cuda:0")
    out = m(G.to_gpu(synth.make_pixels(1, 518, 518, seed=0)))
    G.sync()
    res = {k: cases.probe(v.cpu().numpy()) for k, v in taps.items()}
    res["pred_logits"], res["pred_boxes"] = out["pred_logits"].cpu().numpy(), out["pred_boxes"].cpu().numpy()
    eng.clear_taps()
    del m
    torch.cuda.empty_cache()
    return res, blocks, cases.golden(name)


# (giant x fp16x2 at full depth is left to the two-block test above and to ViT-L at full depth: the -m gpu suite has a 900 s limit on the
#  driver's box and the 40-block model costs 12-20 s per mode)
SLOW = pytest.mark.skipif(os.environ.get("DINODET_SLOW_TESTS") != "1", reason="40-block ViT-g in a third / fourth mode: outside the 900 s budget of "
                          "the driver's -m gpu run; run once per round with DINODET_SLOW_TESTS=1 (DESIGN.md section 9 records the result)")


@pytest.mark.parametrize("variant,precision", [("large", "fp32"), ("large", "bf16x3"), ("large", "fp16x2"), ("giant", "fp32"), ("giant", "bf16x3"),
                                               pytest.param("giant", "fp16x2", marks=SLOW)])
def test_full_depth_configs_gated_vs_reference(G, variant, precision):
    """configs[3] ViT-L/14 and configs[4] ViT-g/14 (SwiGLU, 40 blocks) at 518x518 with 300 queries, ALL blocks, one image:
    both parity-gated modes against the REFERENCE's own forward (G7 / G8: modeling_dinov2.py:300-314 at depth, the 1024 /
    1536 -> 768 projection dinov2_backbone.py:33-37,64-65) -- per-stage probes and the 1e-3 gate on logits and boxes."""
    r, blocks, g = _full_run(G, variant, precision)
    stage = DEEP[precision]                                 # the compensated modes: 2^-17 .. 2^-16 per product, x depth
    assert rel_err(r["embeddings"], g["embeddings_probe"]) < 1e-5
    for b in blocks:
        e = rel_err(r[f"block{b}"], g[f"block{b}_probe"])
        print(f"{variant} {precision} block{b}: {e:.2e}")
        assert e < stage, (b, e)
    e = rel_err(r["feat"], g["feat_probe"])
    print(f"{variant} {precision} features: {e:.2e}")
    assert e < stage
    for k in ("pred_logits", "pred_boxes"):
        e = rel_err(r[k], g[k])
        print(f"{variant} {precision} {k}: {e:.2e} (gate {TOL:g})")
        assert e < TOL, (k, e)


@pytest.mark.parametrize("variant,precision", [("giant", "fp8"), ("large", "bf16"), pytest.param("giant", "bf16", marks=SLOW)])   # giant is resident from the test above; each in the mode BASELINE quotes it in
def test_full_depth_configs_throughput_modes(G, variant, precision):
    """The same two configurations in the opt-in throughput modes (configs[3] is quoted in bf16, configs[4] in fp8), full depth.
    Held (a) stage by stage to the oracle evaluated with the SAME operand rounding (tests/golden/emu_*.npz, generated by
    tests/golden/make_emu_fixtures.py from the CPU oracle: tight on the embeddings and block 0, where two evaluations have not
    decorrelated yet) and (b) on features / logits / boxes to the REFERENCE (G7 / G8): no further from it than that faithful
    oracle is, x K -- so an accuracy regression of a kernel is visible at the configuration's real depth."""
    r, blocks, g = _full_run(G, variant, precision)
    emu = cases.golden("emu_" + cases.FULL_DEPTH[variant][0])
    p = precision + "_"
    assert rel_err(r["embeddings"], emu[p + "embeddings_probe"]) < 1e-5
    e0 = rel_err(r[f"block{blocks[0]}"], emu[p + f"block{blocks[0]}_probe"])
    print(f"{variant} {precision} block0 vs faithful oracle: {e0:.2e}")
    assert e0 < (4e-3 if precision == "bf16" else 4e-2)          # one bf16 ulp = 3.9e-3; one e4m3 step = 6e-2
    K = 1.3
    for b in blocks[1:]:
        got, floor = rel_l2(r[f"block{b}"], g[f"block{b}_probe"]), rel_l2(emu[p + f"block{b}_probe"], g[f"block{b}_probe"])
        print(f"{variant} {precision} block{b}: rel-L2 vs the reference {got:.2e} (faithful oracle {floor:.2e})")
        assert got < K * floor, (b, got, floor)
    for k, gk, ek in (("feat", "feat_probe", p + "feat_probe"), ("pred_logits", "pred_logits", p + "pred_logits"), ("pred_boxes", "pred_boxes", p + "pred_boxes")):
        got, floor = rel_l2(r[k], g[gk]), rel_l2(emu[ek], g[gk])
        print(f"{variant} {precision} {k}: rel-L2 vs the reference {got:.2e} (faithful oracle {floor:.2e})")
        assert np.isfinite(r[k]).all() and got < K * floor, (k, got, floor)
    if precision == "fp8":
        # The fp8 mode's own accuracy criterion against the REFERENCE (golden G8), not against an emulation of itself: the class with the
        # highest logit agrees for >= 99 % of the 300 queries and the logits stay within 1.5e-1 rel-L2.  (Round 3, per-row / per-feature
        # scales: 96.3 % / 1.6e-1; block scales on both operands: 99.3 % / 1.4e-1.  5e-2 is out of reach of ANY use of e4m3 operands in
        # every block: the oracle puts a model with e4m3 on weights_in alone at 9.1e-2, on QKV alone at 1.2e-1 -- DESIGN section 2.)
        top1 = float((r["pred_logits"].argmax(-1) == g["pred_logits"].argmax(-1)).mean())
        l2 = rel_l2(r["pred_logits"], g["pred_logits"])
        print(f"{variant} fp8 vs the reference: top-1 class agreement {top1:.4f}, logits rel-L2 {l2:.3e}, boxes rel-L2 {rel_l2(r['pred_boxes'], g['pred_boxes']):.3e}")
        assert top1 >= 0.99 and l2 <= 1.5e-1


_SWEEP_ORACLE = {}


@pytest.mark.parametrize("precision", GATED)
def test_reference_point_conditioning_sweep(G, precision):
    """The synthetic reference_points_proj weight uses sigma = 0.01 so that the fp32 noise floor on the logits leaves a margin
    under the 1e-3 gate (a reference-point error d moves a sample by d (w - 1) = 136..256 tokens).  This sweep records how the
    error grows with sigma (0.01, 0.03): the HIP path against the EXACT (fp64) result next to the fp32 CPU evaluation against
    the same -- the HIP path must track the fp32 evaluation's own floor (x3), whatever the conditioning."""
    bb, dc = cases.vitb(100, True)
    x = synth.make_pixels(1, 224, 224, seed=0)
    for sigma in (0.01, 0.03):
        sd = dict(synth.detector_state_dict(bb, dc, seed=1))
        w = sd["decoder.decoder.layers.0.reference_points_proj.weight"] * np.float32(sigma / 0.01)
        for j in range(dc.num_layers):                   # tied layers: one array under every j
            sd[f"decoder.decoder.layers.{j}.reference_points_proj.weight"] = w
        m = G.make_detector(bb, dc, precision, "facebook/dinov2-base")
        G.load_np_state(m, sd)
        out = m(G.to_gpu(x))
        G.sync()
        if sigma not in _SWEEP_ORACLE:                   # the two CPU evaluations do not depend on the mode: once per sigma
            _SWEEP_ORACLE[sigma] = (orc.detector_forward(sd, bb, dc, x, dtype=torch.float64), orc.detector_forward(sd, bb, dc, x))
        exact, f32 = _SWEEP_ORACLE[sigma]
        for k in ("pred_logits", "pred_boxes"):
            got = rel_err(out[k].cpu().numpy(), exact[k].numpy())
            floor = rel_err(f32[k].numpy(), exact[k].numpy())
            print(f"sigma {sigma} {precision} {k}: HIP vs fp64 {got:.2e}, fp32 CPU vs fp64 {floor:.2e}")
            assert got < max(TOL if sigma == 0.01 else 0.0, {"fp32": 3.0, "bf16x3": 5.0, "fp16x2": 8.0}[precision] * floor), (sigma, k, got, floor)


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
@pytest.mark.parametrize("R", [70, 56, 224])
def test_fused_patch_embed_matches_the_explicit_im2col_path(G, precision, R):
    """K1 + K2 as ONE kernel (implicit im2col, patch_embed.hip) against the im2col + GEMM pair it replaces
    (test option "no_fused_patch" at weight-pack time): same products, a different fp32 summation order -> 1e-6; and against the
    reference's embeddings (golden G0) where a fixture exists."""
    from dinov2_od_amd.models import DINOv2Backbone
    bb = cases.micro_bb(False) if R != 224 else cases.cfg1(25)[0]
    sd = synth.backbone_state_dict(bb, seed=1, prefix="")
    x = G.to_gpu(synth.make_pixels(3, R, R, seed=0))
    N = num_tokens(R, R)
    embs = {}
    for fused in (True, False):
        nat.set_option("no_fused_patch", 0 if fused else 1)
        try:
            m = DINOv2Backbone("micro" if R != 224 else "facebook/dinov2-small", lora_r=bb.lora_r, lora_alpha=1.0, target_dim=bb.target_dim or None,
                               pretrained=False, precision=precision, config=bb)
            G.load_np_state(m, sd)
            m = m.to(G.dev()).eval()
            tap = m._get_engine().set_tap(0, (3, N, bb.hidden), "cuda:0")
            m(x)
            G.sync()
        finally:
            nat.set_option("no_fused_patch", -1)
        embs[fused] = tap.cpu().numpy().copy()
    e = rel_err(embs[True], embs[False])
    print(f"fused vs explicit patch embed R={R} {precision}: {e:.2e}")
    assert e < (2e-6 if precision == "bf16" else 1e-5)      # split products: 2^-18 class either way
    if R != 224:
        g = cases.golden("g0_micro_backbone")
        assert rel_err(embs[True][:2], g[f"embeddings_{R}"]) < (1e-5 if precision == "bf16x3" else 3e-3)
