"""Pin the CPU oracle against golden vectors produced by the reference itself
(tests/golden/make_goldens.py).  Runs without a GPU."""
import numpy as np
import pytest
import torch

from dinov2_od_amd import synth
from oracle import dinodet_oracle as orc
from tests import cases
from tests.cases import rel_err

TOL = 2e-5   # fp32 vs fp32, different summation order


@pytest.mark.parametrize("swiglu", [False, True])
def test_micro_backbone_all_stages(swiglu):
    g = cases.golden("g4_micro_swiglu" if swiglu else "g0_micro_backbone")
    bb = cases.micro_bb(swiglu)
    sd = synth.backbone_state_dict(bb, seed=1, prefix="backbone.")
    for R in (70, 56):
        x = synth.make_pixels(2, R, R, seed=0)
        taps = {}
        f = orc.backbone_forward(sd, bb, x, taps=taps)
        assert rel_err(taps["embeddings"].numpy(), g[f"embeddings_{R}"]) < TOL
        for i in range(bb.layers):
            assert rel_err(taps[f"block{i}"].numpy(), g[f"block{i}_{R}"]) < TOL
        assert rel_err(f.numpy(), g[f"features_{R}"]) < TOL


@pytest.mark.parametrize("case", cases.G1_CASES, ids=[c[0] for c in cases.G1_CASES])
def test_decoder_only(case):
    tag, deform, Dd, Hd, Q, Ns = case
    g = cases.golden("g1_decoder_only")
    dc = cases.dec_cfg(deform, Dd, Hd, Q)
    sd = synth.decoder_state_dict(dc, seed=1, prefix="decoder.")
    for N in Ns:
        mem = cases.g1_memory(N, Dd)
        with torch.no_grad():
            logits, boxes = orc.decoder_forward(sd, dc, mem)
        assert rel_err(logits.numpy(), g[f"{tag}_N{N}_logits"]) < 1e-4, (tag, N)
        assert rel_err(boxes.numpy(), g[f"{tag}_N{N}_boxes"]) < 1e-4, (tag, N)


@pytest.mark.parametrize("Q", [25, 100])
def test_cfg1_end_to_end(Q):
    g = cases.golden(f"g2_cfg1_q{Q}")
    bb, dc = cases.cfg1(Q)
    sd = synth.detector_state_dict(bb, dc, seed=1)
    x = synth.make_pixels(2, 224, 224, seed=0)
    out = orc.detector_forward(sd, bb, dc, x)
    f = out["features"].numpy()
    assert rel_err(f[:, ::max(1, f.shape[1] // 8), :64], g["feat_probe"]) < 1e-4
    assert rel_err(out["pred_logits"].numpy(), g["pred_logits"]) < 1e-3
    assert rel_err(out["pred_boxes"].numpy(), g["pred_boxes"]) < 1e-3


@pytest.mark.parametrize("name,R,deform", [("g3_vitb_224", 224, True), ("g3_vitb_518", 518, True),
                                           ("g3_vitb_224_dense", 224, False)])
def test_vitb_end_to_end(name, R, deform):
    g = cases.golden(name)
    bb, dc = cases.vitb(100, deform)
    sd = synth.detector_state_dict(bb, dc, seed=1)
    x = synth.make_pixels(1, R, R, seed=0)
    out = orc.detector_forward(sd, bb, dc, x)
    f = out["features"].numpy()
    assert rel_err(f[:, ::max(1, f.shape[1] // 8), :64], g["feat_probe"]) < 1e-4
    assert rel_err(out["pred_logits"].numpy(), g["pred_logits"]) < 1e-3
    assert rel_err(out["pred_boxes"].numpy(), g["pred_boxes"]) < 1e-3


@pytest.mark.parametrize("name,R", [("g2_cfg1_q25", 224), ("g3_vitb_224", 224)])
def test_fp16x2_operand_scheme_is_inside_the_gate(name, R):
    """The fp16x2 mode's operand scheme (fp16 main product + e4m3 cross terms for the block linears), evaluated on the CPU: its own
    distance from the REFERENCE's outputs sits inside the 1e-3 gate, and within a small factor of the fp32 oracle's (the GPU
    tests hold the HIP path to the same goldens; this one separates the scheme from the kernels)."""
    g = cases.golden(name)
    bb, dc = cases.cfg1(25) if name.startswith("g2") else cases.vitb(100, True)
    sd = synth.detector_state_dict(bb, dc, seed=1)
    x = synth.make_pixels(2 if name.startswith("g2") else 1, R, R, seed=0)
    out = orc.detector_forward(sd, bb, dc, x, emulate_bf16="fp16x2")
    for k in ("pred_logits", "pred_boxes"):
        e = cases.rel_err(out[k].numpy(), g[k])
        print(f"{name} fp16x2 scheme {k}: {e:.2e}")
        assert e < 1e-3, (k, e)


@pytest.mark.parametrize("variant", ["large", "giant"])
def test_oracle_full_depth_vs_reference(variant):
    """G7 / G8: BASELINE configs[3] / [4] at their workload (518x518, 300 queries, all 24 / 40 blocks, one image): the
    oracle against the REFERENCE's own outputs and per-stage probes.  ViT-g costs ~3 minutes of CPU: only with
    DINODET_SLOW_TESTS=1 (the GPU suite checks the HIP path against the same fixture either way)."""
    import os
    if variant == "giant" and os.environ.get("DINODET_SLOW_TESTS") != "1":
        pytest.skip("ViT-g full depth on the CPU: set DINODET_SLOW_TESTS=1")
    name, blocks = cases.FULL_DEPTH[variant]
    g = cases.golden(name)
    bb, dc = cases.vit_full(variant)
    sd = synth.detector_state_dict(bb, dc, seed=1)
    x = synth.make_pixels(1, 518, 518, seed=0)
    taps = {}
    out = orc.detector_forward(sd, bb, dc, x, taps=taps)
    assert rel_err(cases.probe(taps["embeddings"].numpy()), g["embeddings_probe"]) < 1e-5
    for b in blocks:
        assert rel_err(cases.probe(taps[f"block{b}"].numpy()), g[f"block{b}_probe"]) < 1e-4, b
    assert rel_err(cases.probe(out["features"].numpy()), g["feat_probe"]) < 1e-4
    assert rel_err(out["pred_logits"].numpy(), g["pred_logits"]) < 1e-3
    assert rel_err(out["pred_boxes"].numpy(), g["pred_boxes"]) < 1e-3


def test_spatial_factor_quirk():
    """(h,w) includes the CLS token: deformable_attention.py:241-256."""
    from dinov2_od_amd.config import spatial_factor
    from oracle.dinodet_oracle import spatial_factor as oracle_factor      # the oracle's own restatement (it imports none of the product's arithmetic)
    for f in (spatial_factor, oracle_factor):
        assert f(257) == (1, 257)
        assert f(1370) == (10, 137)
        assert f(26) == (2, 13)
        assert f(17) == (1, 17)
        assert f(256) == (16, 16)
    for n in range(1, 3000):                      # host mirror == oracle on every token count a 14-pixel patch grid up to 756 x 756 can produce
        assert spatial_factor(n) == oracle_factor(n), n


def test_bicubic_restatement_matches_torch():
    src = torch.from_numpy(synth.normal(5, "bicubic", (3, 5, 5), 1.0))
    want = torch.nn.functional.interpolate(src[None], size=(4, 4), mode="bicubic", align_corners=False)[0]
    got = orc.bicubic_resize_ref(src, 4, 4)
    assert rel_err(got.numpy(), want.numpy()) < 1e-6
    src = torch.from_numpy(synth.normal(5, "bicubic37", (2, 37, 37), 1.0))
    want = torch.nn.functional.interpolate(src[None], size=(16, 16), mode="bicubic", align_corners=False)[0]
    got = orc.bicubic_resize_ref(src, 16, 16)
    assert rel_err(got.numpy(), want.numpy()) < 1e-6


def test_postprocess_oracle_matches_reference_evaluate_coco():
    """G5: the numpy restatement of utils.py:195-233 against the reference's evaluate_coco output"""
    from oracle import postprocess_oracle as ppo
    g = cases.golden("g5_postprocess")
    det = g["det"]
    C = det.shape[-1] - 4
    got = {k: [] for k in ("image_id", "category_id", "bbox", "score")}
    for (lo, hi), ids in zip(g["batch_bounds"], g["image_ids"]):
        ids = [None if v < 0 else int(v) for v in ids[: hi - lo]]
        r = ppo.postprocess(det[lo:hi], C, ids, 0.05)
        for k in got:
            got[k].append(r[k])
    got = {k: np.concatenate(v) for k, v in got.items()}
    assert len(got["image_id"]) == len(g["r_image_id"]) > 100
    assert np.array_equal(got["image_id"], g["r_image_id"])
    assert np.array_equal(got["category_id"], g["r_category_id"])
    assert np.array_equal(got["bbox"].astype(np.float64), g["r_bbox"])            # bit-exact fp32 box arithmetic
    assert np.max(np.abs(got["score"].astype(np.float64) - g["r_score"])) < 2e-7  # sigmoid: <= 1-2 ulp of fp32


@pytest.mark.parametrize("tag,kw", [("default", {}), ("g15", dict(cost_class=2.0, cost_bbox=1.0, cost_giou=3.0, focal_alpha=0.4, focal_gamma=1.5))])
def test_matching_oracle_matches_reference_matcher(tag, kw):
    """G6: restated cost matrices vs the ones the reference's HungarianMatcher handed to scipy, and its assignment"""
    from oracle import matching_oracle as mo
    g = cases.golden("g6_matcher")
    det, offs = g["det"], g["offs"]
    C = det.shape[-1] - 4
    costs = mo.cost_matrices(det, C, g["labels"], g["gt"], offs, rows_from=0, **kw)
    idx = mo.assign(costs)
    for b in range(det.shape[0]):
        want = g[f"{tag}_cost{b}"]
        assert costs[b].shape == want.shape
        if want.size:
            assert np.max(np.abs(costs[b] - want)) <= 1e-6 * max(1.0, np.abs(want).max())
        assert np.array_equal(idx[b][0], g[f"{tag}_i{b}"]) and np.array_equal(idx[b][1], g[f"{tag}_j{b}"])
    assert g["default_cost1"].shape == (det.shape[1], 0) and list(g["emptydict_n0"]) == [0, 0]


@pytest.mark.parametrize("hw", [(480, 640), (427, 640), (333, 500), (224, 224), (100, 80), (518, 700), (1000, 37), (225, 223)])
@pytest.mark.parametrize("out", [(224, 224), (518, 518)])
def test_preprocess_oracle_matches_pillow(hw, out):
    """f4: the restated Pillow BILINEAR resample (8-bit, two-pass fixed point) is bit-exact against Pillow itself"""
    from PIL import Image
    from oracle import preprocess_oracle as ppo
    rng = np.random.default_rng(hw[0] * 1000 + hw[1])
    img = rng.integers(0, 256, size=(hw[0], hw[1], 3), dtype=np.uint8)
    if hw[0] == 333:
        img[:, ::2] = 255          # hard edges: exercises rounding / clipping
        img[:, 1::2] = 0
    want = np.array(Image.fromarray(img, "RGB").resize((out[1], out[0]), Image.BILINEAR))
    got = ppo.resize_bilinear_u8(img, out[0], out[1])
    assert got.shape == want.shape and np.array_equal(got, want)
    t = ppo.to_tensor(got)
    assert t.shape == (3, out[0], out[1]) and t.dtype == np.float32 and float(t.max()) <= 1.0


def test_synthetic_generator_chunked_path_is_bit_identical_to_a_single_pass():
    """synth generates large tensors in chunks on a thread pool; every value must equal the single-pass evaluation bit for bit
    (the goldens were generated from the reference with these weights)."""
    from dinov2_od_amd import synth

    def single_normal(seed, key, n, std):
        u1 = ((synth._stream(seed, key, n, 0) >> np.uint64(11)).astype(np.float64) + 1.0) * (2.0 ** -53)
        u2 = (synth._stream(seed, key, n, 1) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)
        return (0.0 + std * (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2))).astype(np.float32)

    for n in (2 * synth._CHUNK, 2 * synth._CHUNK + 1, 5 * synth._CHUNK + 12345):
        a = synth.normal(1, f"chunk.{n}", (n,), 0.02)
        assert np.array_equal(a.view(np.uint32), single_normal(1, f"chunk.{n}", n, 0.02).view(np.uint32)), n
        u = synth.uniform01(3, f"chunk.u.{n}", (n,))
        want = (synth._stream(3, f"chunk.u.{n}", n, 0) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
        assert np.array_equal(u, want), n
