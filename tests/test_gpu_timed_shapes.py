"""`-m gpu`: parity of the kernels that bench.py TIMES, at the shapes it times them.

The reference goldens are batch 1-2 (<= 2 740 token rows); the dispatcher switches kernels at M >= 4 096 rows (256x256 ping-pong /
k64 / H2 tiles, the tail split, the round-aware rule) and the engine runs >= 10 000 token rows as two concurrent micro-batches inside
a captured hipGraph.  These tests run exactly those configurations:
  * the forward of BASELINE configs[2] at 8 images (its per-GPU shard) and at 64 images (what `bench.py` times), hipGraph + micro-
    batches live, image 0 against the REFERENCE's own golden (G3) and images 0-1 against the CPU oracle, every precision mode;
  * configs[3] / configs[4] at their timed per-GPU batches (ViT-L x 16, ViT-g x 32) against the reference's G7 / G8 goldens;
  * the block GEMMs at M = 87 680 (64 x 1370 rows) x the four ViT-B shapes, each kernel family, against fp64 on a row sample and
    against a full-matrix product (a tile that was never written shows up wherever it is).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dinov2_od_amd import _native as nat, synth
from oracle import dinodet_oracle as orc
from tests import cases
from tests.cases import rel_err, rel_l2
from tests.test_gpu_h2 import pack as pack_h2, decode as decode_h2
from tests.test_gpu_x3 import _pair

pytestmark = pytest.mark.gpu
TOL = 1e-3
GATED = ["fp32", "bf16x3", "fp16x2"]
K_BF16 = 1.3


@pytest.fixture(scope="module")
def G():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from tests import gpu_util
    return gpu_util


# ------------------------------------------------------------------------------------------------ the timed forward configurations
_ORC = {}


def _oracle_vitb518():
    """oracle evaluations of images 0-1 of the bench batch (fp32, and with bf16 operand rounding), once per session"""
    if not _ORC:
        bb, dc = cases.vitb(100, True)
        sd = synth.detector_state_dict(bb, dc, seed=1)
        x = synth.make_pixels(2, 518, 518, seed=0)
        _ORC["f32"] = orc.detector_forward(sd, bb, dc, x)
        _ORC["bf16"] = orc.detector_forward(sd, bb, dc, x, emulate_bf16=True)
    return _ORC


def _bench_images(B, R, G):
    """bench.py make_images: image i of the global batch = synth.make_pixels(...)[i]"""
    x = torch.empty(B, 3, R, R, device="cuda")
    for i in range(B):
        x[i] = torch.from_numpy(synth.uniform01(0, f"pixel_values.{R}x{R}.{i}", (3, R, R))).cuda()
    return x


@pytest.mark.parametrize("precision", GATED + ["bf16"])
@pytest.mark.parametrize("B", [8, 64])
def test_timed_vitb518_configuration_vs_reference_and_oracle(G, B, precision):
    """BASELINE configs[2] as bench.py runs it: ViT-B/14 518x518, Q = 100, B = 8 (10 960 rows: the per-GPU shard of the 8-GPU split)
    and B = 64 (87 680 rows: the timed batch), engine hipGraph on, two concurrent micro-batches (rows >= 10 000), every M >= 4 096
    dispatch live.  Image 0 against the reference's golden G3, images 0-1 against the CPU oracle."""
    if precision == "fp32" and B == 64:
        B = 16          # bench.py's fp32 leg runs 16 images (21 920 rows: the same large-M kernels)
    bb, dc = cases.vitb(100, True)
    m = G.make_detector(bb, dc, precision, "facebook/dinov2-base")
    eng = m._get_engine()
    assert eng._micro_ok(B, 518, 518) and eng.micro_streams == 2, "the timed configuration runs as two concurrent micro-batches"
    x = _bench_images(B, 518, G)
    m.enable_hipgraph()
    m.forward_packed(x)                                   # capture
    det = m.forward_packed(x).clone()                     # replay
    G.sync()
    C = dc.num_classes
    got_l, got_b = det[:2, :, :C].cpu().numpy(), det[:2, :, C:].cpu().numpy()
    g = cases.golden("g3_vitb_518")
    o = _oracle_vitb518()
    ref_l, ref_b = o["f32"]["pred_logits"].numpy(), o["f32"]["pred_boxes"].numpy()
    if precision in GATED:
        e = [rel_err(got_l[:1], g["pred_logits"]), rel_err(got_b[:1], g["pred_boxes"]), rel_err(got_l, ref_l), rel_err(got_b, ref_b)]
        print(f"timed config B={B} {precision}: image 0 vs reference golden logits {e[0]:.2e} boxes {e[1]:.2e}; images 0-1 vs oracle {e[2]:.2e} {e[3]:.2e}"
              f" (rel-L2 {rel_l2(got_l, ref_l):.2e} {rel_l2(got_b, ref_b):.2e})")
        assert max(e) < TOL, e
    else:
        # single-pass bf16 operands: no further from fp32 than the bf16-faithful oracle itself is (x K_BF16), as in test_gpu_forward.py
        emu_l, emu_b = o["bf16"]["pred_logits"].numpy(), o["bf16"]["pred_boxes"].numpy()
        el, eb, ol, ob = rel_l2(got_l, ref_l), rel_l2(got_b, ref_b), rel_l2(emu_l, ref_l), rel_l2(emu_b, ref_b)
        print(f"timed config B={B} bf16: rel-L2 vs fp32 oracle logits {el:.2e} (faithful emulation {ol:.2e}) boxes {eb:.2e} ({ob:.2e})")
        assert el < K_BF16 * ol and eb < K_BF16 * ob
    if precision == "fp32":
        # exact-fp32 kernels, no atomics: the batch's rows equal single-image launches bit for bit, whatever tile / split was chosen
        m.enable_hipgraph(False)
        for i in (0, 1, B - 1):
            assert torch.equal(m.forward_packed(x[i:i + 1])[0], det[i]), i
    # a replay of the same graph is deterministic
    assert torch.equal(m.forward_packed(x) if precision != "fp32" else det, det)


@pytest.mark.parametrize("variant,B,precision", [("large", 16, "bf16"), ("large", 16, "bf16x3"), ("large", 16, "fp16x2"),
                                                 ("giant", 32, "fp8")])      # large is resident from test_gpu_forward.py; ViT-g in the mode BASELINE quotes (20 s per mode)
def test_timed_vitl_vitg_configurations_vs_reference(G, variant, B, precision):
    """BASELINE configs[3] / configs[4] at the per-GPU batches bench.py times (`also.vitl518_bf16`, `also.vitg518_fp8`): ViT-L/14 x 16
    and ViT-g/14 x 32 images of 518x518, Q = 300, graph + micro-batches as shipped.  Image 0 against the reference's full-depth
    goldens G7 / G8: the gated modes at 1e-3; the throughput modes (bf16, fp8) no further from the reference than the oracle
    evaluated with the same operand rounding is (tests/golden/emu_*.npz), x 1.3 -- the bound test_gpu_forward.py holds the batch-1
    launch to."""
    from tests.test_gpu_forward import _full_detector
    name, _ = cases.FULL_DEPTH[variant]
    g = cases.golden(name)
    m, bb, dc = _full_detector(G, variant, precision)
    x = _bench_images(B, 518, G)
    m.enable_hipgraph()
    m.forward_packed(x)
    det = m.forward_packed(x).clone()
    G.sync()
    C = dc.num_classes
    got_l, got_b = det[:1, :, :C].cpu().numpy(), det[:1, :, C:].cpu().numpy()
    if precision in GATED:
        e = (rel_err(got_l, g["pred_logits"]), rel_err(got_b, g["pred_boxes"]))
        print(f"timed config {variant} B={B} {precision}: image 0 vs reference golden logits {e[0]:.2e} boxes {e[1]:.2e}")
        assert max(e) < TOL
    else:
        emu = cases.golden("emu_" + name)
        for k, got in (("pred_logits", got_l), ("pred_boxes", got_b)):
            d, floor = rel_l2(got, g[k]), rel_l2(emu[f"{precision}_{k}"], g[k])
            print(f"timed config {variant} B={B} {precision} {k}: rel-L2 vs the reference {d:.2e} (faithful oracle {floor:.2e})")
            assert np.isfinite(got).all() and d < K_BF16 * floor, (k, d, floor)
    del m
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------ BASELINE configs[1]: ViT-B/14 224x224 x 32
def _oracle_vitb224():
    if "f32_224" not in _ORC:
        bb, dc = cases.vitb(100, True)
        sd = synth.detector_state_dict(bb, dc, seed=1)
        x = synth.make_pixels(2, 224, 224, seed=0)
        _ORC["f32_224"] = orc.detector_forward(sd, bb, dc, x)
        _ORC["bf16_224"] = orc.detector_forward(sd, bb, dc, x, emulate_bf16=True)
    return _ORC


@pytest.mark.parametrize("precision", GATED + ["bf16"])
def test_timed_vitb224_configuration_vs_reference_and_oracle(G, precision):
    """BASELINE configs[1] as bench.py's `also.vitb224_*` legs run it (the resolution of the reference's own transform,
    train.py:584-587): ViT-B/14 224x224, Q = 100, B = 32 -- 8 224 token rows (the 256x128_m16 kernel on 198 tiles, the K-split of the
    underfilled N = 768 rounds in the compensated modes, bicubic position table) and 3 200 query rows (the 128x128 small-grid rule),
    one stream, engine hipGraph on.  Image 0 against the reference's golden G3 (detector.py:58-69), images 0-1 against the CPU oracle;
    fp32 additionally bit-equal to single-image launches."""
    B = 32
    bb, dc = cases.vitb(100, True)
    m = G.make_detector(bb, dc, precision, "facebook/dinov2-base")
    x = _bench_images(B, 224, G)
    m.enable_hipgraph()
    m.forward_packed(x)                                   # capture
    det = m.forward_packed(x).clone()                     # replay
    G.sync()
    C = dc.num_classes
    got_l, got_b = det[:2, :, :C].cpu().numpy(), det[:2, :, C:].cpu().numpy()
    g = cases.golden("g3_vitb_224")
    o = _oracle_vitb224()
    ref_l, ref_b = o["f32_224"]["pred_logits"].numpy(), o["f32_224"]["pred_boxes"].numpy()
    if precision in GATED:
        e = [rel_err(got_l[:1], g["pred_logits"]), rel_err(got_b[:1], g["pred_boxes"]), rel_err(got_l, ref_l), rel_err(got_b, ref_b)]
        print(f"timed config vitb224 B={B} {precision}: image 0 vs reference golden logits {e[0]:.2e} boxes {e[1]:.2e}; images 0-1 vs oracle {e[2]:.2e} {e[3]:.2e}"
              f" (rel-L2 {rel_l2(got_l, ref_l):.2e} {rel_l2(got_b, ref_b):.2e})")
        assert max(e) < TOL, e
    else:
        emu_l, emu_b = o["bf16_224"]["pred_logits"].numpy(), o["bf16_224"]["pred_boxes"].numpy()
        el, eb, ol, ob = rel_l2(got_l, ref_l), rel_l2(got_b, ref_b), rel_l2(emu_l, ref_l), rel_l2(emu_b, ref_b)
        print(f"timed config vitb224 B={B} bf16: rel-L2 vs fp32 oracle logits {el:.2e} (faithful emulation {ol:.2e}) boxes {eb:.2e} ({ob:.2e})")
        assert el < K_BF16 * ol and eb < K_BF16 * ob
    if precision == "fp32":
        m.enable_hipgraph(False)
        for i in (0, 1, B - 1):
            assert torch.equal(m.forward_packed(x[i:i + 1])[0], det[i]), i
    assert torch.equal(m.forward_packed(x) if precision != "fp32" else det, det)


# ------------------------------------------------------------------------------------------------ the block GEMMs at M = 87 680
M_BENCH = 64 * 1370
SHAPES = [("qkv", 2304, 768, "none"), ("proj", 768, 768, "resid"), ("fc1", 3072, 768, "gelu"), ("fc2", 768, 3072, "resid")]


def _sample_rows(M):
    """first tile, the (partial) last tile and the tail-split cut, plus a spread: 87 680 = 342 x 256 + 128"""
    g = torch.Generator().manual_seed(5)
    idx = torch.cat([torch.arange(0, 256), torch.arange(M - 700, M), torch.randint(256, M - 700, (768,), generator=g)])
    return torch.unique(idx)


M_CFG1 = 32 * 257       # BASELINE configs[1]: 32 images of 224x224 = 8 224 rows (198 tiles of 256x128: less than one round of 256 CUs)
M_ROWS = [M_BENCH, M_CFG1]


def _operands(name, N, K, M_BENCH=M_BENCH):
    g = torch.Generator(device="cuda").manual_seed(sum(map(ord, name)) * 7919 + N * 31 + K)
    A = torch.randn(M_BENCH, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) * 0.05
    bias = torch.randn(N, device="cuda", generator=g)
    scale = 1 + 0.1 * torch.randn(N, device="cuda", generator=g)
    resid = torch.randn(M_BENCH, N, device="cuda", generator=g)
    return A, W, bias, scale, resid


def _want(A, W, bias, scale, resid, epi, rows):
    """fp64 on the CPU for the sampled rows"""
    r = A[rows].double().cpu() @ W.double().cpu().t() + bias.double().cpu()
    if epi == "gelu":
        return F.gelu(r)
    if epi == "resid":
        return r * scale.double().cpu() + resid[rows].double().cpu()
    return r


def _full_check(got, A, W, bias, scale, resid, epi, tol):
    """every element against a full-matrix fp32 product on the GPU (torch.matmul: test infrastructure, not the product) -- loose, but
    it sees all 87 680 x N outputs"""
    ref = A.float() @ W.float().t() + bias
    if epi == "gelu":
        ref = F.gelu(ref)
    if epi == "resid":
        ref = ref * scale + resid
    err = float((got.float() - ref).abs().max() / ref.abs().max())
    assert err < tol, f"full-matrix check: {err:.3e}"


@pytest.mark.parametrize("M_BENCH", M_ROWS)
@pytest.mark.parametrize("name,N,K,epi", SHAPES, ids=[s[0] for s in SHAPES])
def test_bench_shape_gemm_plain_bf16(name, N, K, epi, M_BENCH):
    """the single-pass bf16 kernels bench.py's headline times (gemm_ppm_256x256<false>, gemm_x3_256x256<PLAIN>, 256x128_m16 for the
    GELU fc1; tail split by the shipped heuristic) against the exact product of the bf16-rounded operands"""
    L = nat.lib()
    A, W, bias, scale, resid = _operands(name, N, K, M_BENCH)
    Ab, Wb = A.bfloat16(), W.bfloat16()
    rows = _sample_rows(M_BENCH)
    want = _want(Ab, Wb, bias, scale, resid, epi, rows.cuda()).numpy()
    if epi == "resid":        # in-place fp32 residual stream with LayerScale (K6 / K7 epilogue)
        x = resid.clone()
        nat.check(L.dod_op_linear(1, nat.ptr(Ab), K, nat.ptr(Wb), K, M_BENCH, N, K, nat.ptr(bias), nat.ptr(scale), nat.ptr(x), N, nat.ptr(x), 0, N, 0, nat.stream_ptr()))
        assert rel_err(x[rows.cuda()].cpu().numpy(), want) < 3e-6
        _full_check(x, Ab, Wb, bias, scale, resid, epi, 1e-4)
    else:                     # bf16 output (QKV; fc1 with the GELU epilogue)
        out = torch.empty(M_BENCH, N, dtype=torch.bfloat16, device="cuda")
        nat.check(L.dod_op_linear(1, nat.ptr(Ab), K, nat.ptr(Wb), K, M_BENCH, N, K, nat.ptr(bias), None, None, 0, nat.ptr(out), 1, N, nat.ACT[epi], nat.stream_ptr()))
        assert rel_err(out[rows.cuda()].float().cpu().numpy(), want) < 2 ** -8
        _full_check(out, Ab, Wb, bias, scale, resid, epi, 2 ** -7)
        o32 = torch.empty(M_BENCH, N, dtype=torch.float32, device="cuda")
        nat.check(L.dod_op_linear(1, nat.ptr(Ab), K, nat.ptr(Wb), K, M_BENCH, N, K, nat.ptr(bias), None, None, 0, nat.ptr(o32), 0, N, nat.ACT[epi], nat.stream_ptr()))
        assert rel_err(o32[rows.cuda()].cpu().numpy(), want) < (3e-6 if epi == "none" else 2e-5)      # GELU: the A&S erf of the bf16 path


@pytest.mark.parametrize("M_BENCH", M_ROWS)
@pytest.mark.parametrize("name,N,K,epi", SHAPES, ids=[s[0] for s in SHAPES])
def test_bench_shape_gemm_split_product(name, N, K, epi, M_BENCH):
    """gemm_ppm_256x256<X3> / gemm_x3_256x256 (bf16x3 mode) at the timed shapes against the exact product of the fp32 inputs"""
    L = nat.lib()
    A, W, bias, scale, resid = _operands(name, N, K, M_BENCH)
    A2, W2 = _pair(A), _pair(W)
    rows = _sample_rows(M_BENCH)
    want = _want(A, W, bias, scale, resid, epi, rows.cuda()).numpy()
    if epi == "resid":
        x = resid.clone()
        nat.check(L.dod_op_linear_x3(nat.ptr(A2), nat.ptr(W2), M_BENCH, N, K, nat.ptr(bias), nat.ptr(scale), nat.ptr(x), N, nat.ptr(x), 0, N, 0, nat.stream_ptr()))
        assert rel_err(x[rows.cuda()].cpu().numpy(), want) < 3e-5
        _full_check(x, A, W, bias, scale, resid, epi, 2e-4)
    else:                     # pair-layout output [hi | lo] (what the forward writes for QKV and the GELU fc1)
        out = torch.empty(M_BENCH, 2 * N, dtype=torch.bfloat16, device="cuda")
        nat.check(L.dod_op_linear_x3(nat.ptr(A2), nat.ptr(W2), M_BENCH, N, K, nat.ptr(bias), None, None, 0, nat.ptr(out), 2, 2 * N, nat.ACT[epi], nat.stream_ptr()))
        o = out[rows.cuda()].float().cpu()
        assert rel_err((o[:, :N] + o[:, N:]).numpy(), want) < 3e-5
        _full_check(out[:, :N].float() + out[:, N:].float(), A, W, bias, scale, resid, epi, 2e-4)


@pytest.mark.parametrize("M_BENCH", M_ROWS)
@pytest.mark.parametrize("name,N,K,epi", SHAPES, ids=[s[0] for s in SHAPES])
def test_bench_shape_gemm_h2(name, N, K, epi, M_BENCH):
    """gemm_h2_256x256 (fp16x2 mode) at the timed shapes against the exact product of the fp32 inputs"""
    L = nat.lib()
    A, W, bias, scale, resid = _operands(name, N, K, M_BENCH)
    Ab, _ = pack_h2(A)
    Wb, wexp = pack_h2(W, weight=True)
    rows = _sample_rows(M_BENCH)
    want = _want(A, W, bias, scale, resid, epi, rows.cuda()).numpy()
    if epi == "resid":
        x = resid.clone()
        nat.check(L.dod_op_linear_h2(nat.ptr(Ab), nat.ptr(Wb), nat.ptr(wexp), M_BENCH, N, K, nat.ptr(bias), nat.ptr(scale), nat.ptr(x), N, nat.ptr(x), 0, N, 0, nat.stream_ptr()))
        assert rel_err(x[rows.cuda()].cpu().numpy(), want) < 5e-5
        _full_check(x, A, W, bias, scale, resid, epi, 2e-4)
    elif epi == "gelu":       # H2 rows out (the forward's fc1 epilogue)
        out = torch.empty(M_BENCH, 2 * N, dtype=torch.bfloat16, device="cuda")
        nat.check(L.dod_op_linear_h2(nat.ptr(Ab), nat.ptr(Wb), nat.ptr(wexp), M_BENCH, N, K, nat.ptr(bias), None, None, 0, nat.ptr(out), 3, 2 * N, nat.ACT[epi], nat.stream_ptr()))
        h, _, r8 = decode_h2(out[rows.cuda()].contiguous().view(torch.uint8), N)
        assert rel_err((h + r8).numpy(), want) < 5e-5
    else:                     # pair-layout output (the forward's QKV epilogue: attention reads bf16 pairs)
        out = torch.empty(M_BENCH, 2 * N, dtype=torch.bfloat16, device="cuda")
        nat.check(L.dod_op_linear_h2(nat.ptr(Ab), nat.ptr(Wb), nat.ptr(wexp), M_BENCH, N, K, nat.ptr(bias), None, None, 0, nat.ptr(out), 2, 2 * N, 0, nat.stream_ptr()))
        o = out[rows.cuda()].float().cpu()
        assert rel_err((o[:, :N] + o[:, N:]).numpy(), want) < 5e-5
        _full_check(out[:, :N].float() + out[:, N:].float(), A, W, bias, scale, resid, epi, 2e-4)


# ------------------------------------------------------------------------------------------------ the decoder's query-side linears at B.Q = 3 200
Q_ROWS = 32 * 100
Q_SHAPES = [("in_proj", 2304, 768, "none"), ("attn_out", 768, 768, "resid"), ("linear1", 1024, 768, "relu"), ("linear2", 768, 1024, "resid"),
            ("bbox0", 384, 768, "relu")]


@pytest.mark.parametrize("name,N,K,epi", Q_SHAPES, ids=[s[0] for s in Q_SHAPES])
def test_query_side_linears_at_cfg1_rows(name, N, K, epi):
    """configs[1]'s decoder (deformable_attention.py:215-268 at B.Q = 32 x 100 rows): every query-side linear runs as ONE bf16 GEMM with
    K' = 3K on [Ah | Ah | Al] x [Wh | Wl | Wh]^T through the small-grid rule (128x128 tiles for grids of a few dozen 256x128 tiles).
    Against the exact fp64 product of the fp32 inputs: the split form is good to ~1e-5 (lo.lo dropped)."""
    L = nat.lib()
    g = torch.Generator(device="cuda").manual_seed(sum(map(ord, name)) * 131 + N + K)
    A = torch.randn(Q_ROWS, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) * 0.05
    bias = torch.randn(N, device="cuda", generator=g)
    resid = torch.randn(Q_ROWS, N, device="cuda", generator=g)
    Ah, Wh = A.bfloat16(), W.bfloat16()
    Al, Wl = (A - Ah.float()).bfloat16(), (W - Wh.float()).bfloat16()
    A3 = torch.cat([Ah, Ah, Al], 1).contiguous()
    W3 = torch.cat([Wh, Wl, Wh], 1).contiguous()
    want = A.double().cpu() @ W.double().cpu().t() + bias.double().cpu()
    if epi == "relu":
        want = want.clamp_min(0)
    if epi == "resid":
        want = want + resid.double().cpu()
    out = torch.empty(Q_ROWS, N, device="cuda")
    nat.check(L.dod_op_linear(1, nat.ptr(A3), 3 * K, nat.ptr(W3), 3 * K, Q_ROWS, N, 3 * K, nat.ptr(bias), None,
                              nat.ptr(resid) if epi == "resid" else None, N, nat.ptr(out), 0, N, nat.ACT["relu" if epi == "relu" else "none"], nat.stream_ptr()))
    err = rel_err(out.cpu().numpy(), want.numpy())
    print(f"query-side {name} [{Q_ROWS} x {N} x 3*{K}]: {err:.2e} from fp64")
    assert err < 3e-5
