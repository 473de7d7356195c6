"""`-m gpu`: every HIP kernel, called through the C ABI's operator entry points, against the CPU
oracle / a plain fp32 torch statement of the same op, on seeded inputs.  Tolerances are written
next to each comparison: fp32 kernels 1e-5-level (summation order only); bf16 kernels are
compared with the SAME bf16-rounded operands evaluated in fp32 on the CPU."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dinov2_od_amd import _native as nat
from dinov2_od_amd import synth
from oracle import dinodet_oracle as orc
from tests.cases import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (run with -m 'not gpu' on CPU)")
    from tests import gpu_util
    nat.lib()
    return gpu_util


def _n(key, shape, std=1.0):
    return synth.normal(11, key, shape, std)


@pytest.mark.parametrize("rows,D", [(7, 128), (1000, 384), (2740, 768), (513, 1536)])
def test_layernorm(G, rows, D):
    x, g, b = _n("ln.x", (rows, D), 2.0) + 0.5, 1 + _n("ln.g", (D,), 0.1), _n("ln.b", (D,), 0.1)
    add = _n("ln.add", (rows, D))
    L = nat.lib()
    for use_add in (False, True):
        want = F.layer_norm(torch.from_numpy(x + (add if use_add else 0)), (D,), torch.from_numpy(g), torch.from_numpy(b), 1e-6)
        xd, gd, bd, ad = G.to_gpu(x), G.to_gpu(g), G.to_gpu(b), G.to_gpu(add)
        out = torch.empty(rows, D, device=G.dev())
        nat.check(L.dod_op_layernorm(nat.ptr(xd), nat.ptr(ad) if use_add else None, nat.ptr(gd), nat.ptr(bd), 1e-6, rows, D,
                                     nat.ptr(out), nat.DOD_F32, nat.stream_ptr()))
        assert rel_err(out.cpu().numpy(), want.numpy()) < 2e-6
        outb = torch.empty(rows, D, device=G.dev(), dtype=torch.bfloat16)
        nat.check(L.dod_op_layernorm(nat.ptr(xd), nat.ptr(ad) if use_add else None, nat.ptr(gd), nat.ptr(bd), 1e-6, rows, D,
                                     nat.ptr(outb), nat.DOD_BF16, nat.stream_ptr()))
        # bf16 output = RNE of the fp32 result (one-ulp slack for results on a rounding boundary)
        assert rel_err(outb.float().cpu().numpy(), want.to(torch.bfloat16).float().numpy()) < 2 ** -7


@pytest.mark.parametrize("M,N,K", [(1, 4, 64), (100, 50, 768), (800, 91, 768), (1370, 768, 588), (333, 200, 132), (2740, 2304, 768)])
def test_linear_f32_all_epilogues(G, M, N, K):
    A, W = _n("f.A", (M, K)), _n("f.W", (N, K), 0.05)
    bias, scale, resid = _n("f.b", (N,)), 1 + _n("f.s", (N,), 0.1), _n("f.r", (M, N))
    ref = torch.from_numpy(A).double() @ torch.from_numpy(W).double().t()
    Ad, Wd = G.to_gpu(A), G.to_gpu(W)
    out = G.op_linear(Ad, Wd)
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 3e-6
    for act, fn in (("relu", torch.relu), ("gelu", lambda t: F.gelu(t)), ("sigmoid", torch.sigmoid)):
        want = fn(ref + torch.from_numpy(bias).double()) * torch.from_numpy(scale).double() + torch.from_numpy(resid).double()
        out = G.op_linear(Ad, Wd, G.to_gpu(bias), G.to_gpu(scale), G.to_gpu(resid), act)
        assert rel_err(out.cpu().numpy(), want.numpy()) < 3e-6, act


@pytest.mark.parametrize("M,N,K", [(50, 50, 768), (400, 768, 768), (800, 91, 768), (800, 768, 3072), (3200, 4, 384), (333, 200, 132), (100, 256, 768),
                                   (6400, 50, 768)])
def test_linear_f32_k_split_across_workgroups(G, M, N, K):
    """gemm_f32.hip, round 4: the decoder's small linears split K over grid.y; the last workgroup to arrive at a tile sums the slices in slice
    order.  Same values as the unsplit kernel to fp32 rounding, every epilogue; identical from launch to launch (the counters reset themselves);
    and the slice count is a function of (N, K) alone, so a row's bits do not depend on the number of rows launched with it."""
    L = nat.lib()
    nat.check(L.dod_reserve_gemm_scratch(1 << 20))
    A, W = _n("k.A", (M, K)), _n("k.W", (N, K), 0.05)
    bias, scale, resid = _n("k.b", (N,)), 1 + _n("k.s", (N,), 0.1), _n("k.r", (M, N))
    ref = torch.from_numpy(A).double() @ torch.from_numpy(W).double().t()
    Ad, Wd, bd, sd, rd = G.to_gpu(A), G.to_gpu(W), G.to_gpu(bias), G.to_gpu(scale), G.to_gpu(resid)
    tn, nk = (N + 63) // 64, (K + 15) // 16
    slices = min(8, (512 + 8 * tn) // (16 * tn), nk // 8)
    try:
        nat.set_option("f32_ksplit", 0)
        base = {act: G.op_linear(Ad, Wd, bd, sd, rd, act).clone() for act in ("none", "relu", "sigmoid")}
        nat.set_option("f32_ksplit", 1)
        n0 = L.dod_test_counter(b"f32_ksplits")
        for act, fn in (("none", lambda t: t), ("relu", torch.relu), ("sigmoid", torch.sigmoid)):
            want = fn(ref + torch.from_numpy(bias).double()) * torch.from_numpy(scale).double() + torch.from_numpy(resid).double()
            got = G.op_linear(Ad, Wd, bd, sd, rd, act).clone()
            again = G.op_linear(Ad, Wd, bd, sd, rd, act)
            assert torch.equal(got, again), act
            assert rel_err(got.cpu().numpy(), want.numpy()) < 3e-6, act
            assert rel_err(got.cpu().numpy(), base[act].cpu().numpy()) < 4e-6, act      # two fp32 summation orders of K terms (K = 3072: 2e-6)
        assert L.dod_test_counter(b"f32_ksplits") - n0 == (6 if slices >= 2 else 0)
        # slabs are reused from launch to launch: alternate two inputs, so that a stale partial (one that was read before its slice's store
        # of THIS launch had landed) would show as the other input's value
        Ad2 = (-0.5 * Ad).contiguous()
        want1 = G.op_linear(Ad, Wd, bd, sd, rd, "none").clone()
        want2 = G.op_linear(Ad2, Wd, bd, sd, rd, "none").clone()
        for it in range(60):
            got = G.op_linear(Ad2 if it & 1 else Ad, Wd, bd, sd, rd, "none")
            assert torch.equal(got, want2 if it & 1 else want1), it
        if M >= 100:      # the first 37 rows alone: the same bits
            part = G.op_linear(Ad[:37].contiguous(), Wd, bd, sd, rd[:37].contiguous(), "relu")
            assert torch.equal(part, G.op_linear(Ad, Wd, bd, sd, rd, "relu")[:37])
    finally:
        nat.set_option("f32_ksplit", -1)


@pytest.mark.parametrize("M,N,K", [(1, 4, 64), (128, 128, 64), (257, 384, 128), (1370, 768, 640), (2740, 2304, 768), (1111, 200, 3072)])
def test_linear_bf16_all_epilogues(G, M, N, K):
    A = torch.from_numpy(_n("b.A", (M, K))).to(torch.bfloat16)
    W = torch.from_numpy(_n("b.W", (N, K), 0.05)).to(torch.bfloat16)
    bias, scale, resid = _n("b.b", (N,)), 1 + _n("b.s", (N,), 0.1), _n("b.r", (M, N))
    ref = A.double() @ W.double().t()      # exact products of the same bf16 operands
    Ad, Wd = A.to(G.dev()), W.to(G.dev())
    out = G.op_linear(Ad, Wd)
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 3e-6       # fp32 accumulation only
    want = F.gelu(ref + torch.from_numpy(bias).double())
    out = G.op_linear(Ad, Wd, G.to_gpu(bias), None, None, "gelu", torch.bfloat16)
    assert rel_err(out.float().cpu().numpy(), want.numpy()) < 2 ** -7     # bf16 output rounding
    want = (ref + torch.from_numpy(bias).double()) * torch.from_numpy(scale).double() + torch.from_numpy(resid).double()
    r = G.to_gpu(resid)
    out = G.op_linear(Ad, Wd, G.to_gpu(bias), G.to_gpu(scale), r, "none")
    assert rel_err(out.cpu().numpy(), want.numpy()) < 3e-6
    # x += A W^T + b without LayerScale (decoder residuals)
    want = ref + torch.from_numpy(bias).double() + torch.from_numpy(resid).double()
    out = G.op_linear(Ad, Wd, G.to_gpu(bias), None, r, "none")
    assert rel_err(out.cpu().numpy(), want.numpy()) < 3e-6


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(1370, 768, 768), (2740 + 37, 768, 768), (4110 + 5, 768, 3072), (4500, 1024, 2048), (1500, 384, 384)])
def test_linear_bf16_residual_in_place(G, M, N, K):
    """out-proj / fc2 as the forward issues them: the fp32 residual stream updated in place (256x128 kernel for K < 2048,
    16-wave 256x256 kernel above; ragged last m-tile)."""
    A = torch.from_numpy(_n("ri.A", (M, K))).to(torch.bfloat16)
    W = torch.from_numpy(_n("ri.W", (N, K), 0.05)).to(torch.bfloat16)
    bias, resid = _n("ri.b", (N,)), _n("ri.r", (M, N), 3.0)
    want = A.double() @ W.double().t() + torch.from_numpy(bias).double() + torch.from_numpy(resid).double()
    x = G.to_gpu(resid)
    out = G.op_linear(A.to(G.dev()), W.to(G.dev()), G.to_gpu(bias), None, x, "none", out=x)
    assert out.data_ptr() == x.data_ptr()
    assert rel_err(x.cpu().numpy(), want.numpy()) < 3e-6


def test_linear_rejects_bad_shapes(G):
    A = torch.zeros(8, 100, device=G.dev(), dtype=torch.bfloat16)
    W = torch.zeros(8, 100, device=G.dev(), dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        G.op_linear(A, W)            # K % 64 != 0 for the bf16 kernel
    # the fp32 kernel takes any K and pitch (scalar loads when a pitch rules out float4)
    A, W = _n("odd.A", (9, 6)), _n("odd.W", (5, 6))
    out = G.op_linear(G.to_gpu(A), G.to_gpu(W))
    assert rel_err(out.cpu().numpy(), A.astype(np.float64) @ W.astype(np.float64).T) < 3e-6


def _attn_ref(q, k, v, scale):
    s = (q.double() @ k.double().transpose(-1, -2)) * scale
    return torch.softmax(s, -1) @ v.double()


# the last five reach the 512-thread ping-pong kernel (heads * B * ceil(N / 256) >= 1024): an even and an odd number of key tiles (22, 23; the two
# wave groups split the keys), the short-tail launch form (1370 = 5 x 256 + 90), a partial last block without it (577), ONE key tile (group 1 idle)
@pytest.mark.parametrize("B,N,heads", [(1, 17, 2), (2, 64, 1), (2, 257, 6), (1, 1370, 12), (3, 130, 2),
                                       (16, 1370, 12), (15, 1440, 12), (32, 577, 12), (130, 60, 8), (43, 300, 12)])
def test_attention_bf16(G, B, N, heads):
    D = heads * 64
    rng = np.random.default_rng(B * 1000 + N)
    qkv = torch.from_numpy((rng.standard_normal((B, N, 3 * D)) * 1.5).astype(np.float32)).to(torch.bfloat16)
    L = nat.lib()
    qd = qkv.to(G.dev()).contiguous()
    ctx = torch.empty(B, N, D, device=G.dev(), dtype=torch.bfloat16)
    nat.check(L.dod_op_attention_bf16(nat.ptr(qd), nat.ptr(ctx), B, N, heads, 0.125, nat.stream_ptr()))
    q, k, v = [t.view(B, N, heads, 64).transpose(1, 2) for t in qkv.float().split(D, dim=-1)]
    want = _attn_ref(q, k, v, 0.125).transpose(1, 2).reshape(B, N, D)
    # P is rounded to bf16 before P.V and the context is stored in bf16: 2^-8-level relative error
    assert rel_err(ctx.float().cpu().numpy(), want.numpy()) < 1e-2
    assert float(np.abs(ctx.float().cpu().numpy() - want.numpy()).mean() / np.abs(want.numpy()).mean()) < 3e-3


@pytest.mark.parametrize("B,heads", [(1, 1), (300, 2)], ids=["4-wave", "ping-pong"])
def test_attention_bf16_forced_rescale(G, B, heads):
    """Online-softmax rescale path: one key per later tile dominates every row (cdna guide rule 26).  In the ping-pong kernel the spikes sit
    in DIFFERENT wave groups' key halves (tiles 0-2 / 3-4), so the final merge has to rescale as well."""
    N, D = 300, 64 * heads
    rng = np.random.default_rng(7)
    x = (rng.standard_normal((B, N, 3 * D)) * 0.5).astype(np.float32)
    x[:, 70, D:2 * D] *= 8.0      # key 70 (tile 1) spikes
    x[:, 200, D:2 * D] *= 16.0    # key 200 (tile 3) spikes more
    qkv = torch.from_numpy(x).to(torch.bfloat16)
    qd = qkv.to(G.dev())
    ctx = torch.empty(B, N, D, device=G.dev(), dtype=torch.bfloat16)
    nat.check(nat.lib().dod_op_attention_bf16(nat.ptr(qd), nat.ptr(ctx), B, N, heads, 0.125, nat.stream_ptr()))
    q, k, v = [t.view(B, N, heads, 64).transpose(1, 2) for t in qkv.float().split(D, dim=-1)]
    want = _attn_ref(q, k, v, 0.125).transpose(1, 2).reshape(B, N, D)
    assert rel_err(ctx.float().cpu().numpy(), want.numpy()) < 1e-2


@pytest.mark.parametrize("B,Lq,Lk,heads,dh", [(2, 7, 7, 4, 32), (2, 100, 100, 8, 96), (1, 300, 300, 8, 96), (2, 5, 1370, 2, 96),
                                             (1, 257, 257, 6, 64), (2, 33, 70, 3, 128), (1, 9, 17, 5, 48),
                                             (1, 1370, 1370, 2, 64), (3, 130, 130, 2, 64), (2, 17, 17, 2, 64), (2, 5, 200, 1, 64)])
def test_attention_f32(G, B, Lq, Lk, heads, dh):
    E = heads * dh
    q, k, v = _n("g.q", (B, Lq, E)), _n("g.k", (B, Lk, E)), _n("g.v", (B, Lk, E))
    qd, kd, vd = G.to_gpu(q), G.to_gpu(k), G.to_gpu(v)
    o = torch.empty(B, Lq, E, device=G.dev())
    sc = 1.0 / math.sqrt(dh)
    nat.check(nat.lib().dod_op_attention_f32(nat.ptr(qd), nat.ptr(kd), nat.ptr(vd), nat.ptr(o), E, E, E, E, Lq, Lk, B, heads, dh, sc,
                                             nat.stream_ptr()))
    sp = lambda t, L: torch.from_numpy(t).view(B, L, heads, dh).transpose(1, 2)
    want = _attn_ref(sp(q, Lq), sp(k, Lk), sp(v, Lk), sc).transpose(1, 2).reshape(B, Lq, E)
    assert rel_err(o.cpu().numpy(), want.numpy()) < 5e-6


@pytest.mark.parametrize("N,Hd,dh,P", [(17, 4, 32, 2), (26, 2, 96, 2), (257, 8, 96, 2), (1370, 8, 96, 2), (1370, 4, 64, 4), (256, 2, 128, 1)])
def test_deform_sample_matches_oracle(G, N, Hd, dh, P):
    from dinov2_od_amd.config import spatial_factor
    B, Q, Dd = 2, 9, Hd * dh
    h, w = spatial_factor(N)
    ncat = 2 + 3 * Hd * P
    proj = _n("d.proj", (B * Q, ncat), 1.0)
    proj[:, 2:2 + Hd * P * 2] *= 0.3                 # offsets: some samples clamp at 0/1, most do not
    proj[0, 0] = 30.0                                # reference point exactly at the right edge (x1 clamp path)
    proj[1, 0] = -30.0                               # and at the left edge
    vals = _n("d.vals", (B, N, Dd))
    out = torch.empty(B * Q, Dd, device=G.dev())
    pd, vd = G.to_gpu(proj), G.to_gpu(vals)      # keep the device tensors alive across the launch
    nat.check(nat.lib().dod_op_deform_sample(nat.ptr(pd), ncat, nat.ptr(vd), B, Q, N, Hd, P, dh, h, w,
                                             nat.ptr(out), nat.stream_ptr()))
    pr = torch.from_numpy(proj).view(B, Q, ncat)
    ref = torch.sigmoid(pr[..., :2])
    off = pr[..., 2:2 + Hd * P * 2].reshape(B, Q, Hd, P, 2)
    aw = pr[..., 2 + Hd * P * 2:].reshape(B, Q, Hd, P).softmax(-1)
    want = orc.deformable_sample(torch.from_numpy(vals).view(B, N, Hd, dh), ref, off, aw, h, w).reshape(B * Q, Dd)
    assert rel_err(out.cpu().numpy(), want.numpy()) < 1e-5


def test_deform_sample_rejects_bad_grid(G):
    with pytest.raises(ValueError, match="Cannot reshape"):
        z = torch.zeros(64, device=G.dev())
        nat.check(nat.lib().dod_op_deform_sample(nat.ptr(z), 8, nat.ptr(z), 1, 1, 10, 1, 2, 4, 3, 3, nat.ptr(z), nat.stream_ptr()))


@pytest.mark.parametrize("G_,gh,gw,D", [(5, 4, 4, 128), (37, 16, 16, 768), (37, 37, 20, 64), (5, 9, 7, 64)])
def test_pos_resize_matches_torch_bicubic(G, G_, gh, gw, D):
    pos = _n("p.pos", (G_ * G_ + 1, D), 0.02)
    out = torch.empty(gh * gw + 1, D, device=G.dev())
    pd = G.to_gpu(pos)
    nat.check(nat.lib().dod_op_pos_resize(nat.ptr(pd), G_, gh, gw, D, nat.ptr(out), nat.stream_ptr()))
    p = torch.from_numpy(pos[1:]).reshape(1, G_, G_, D).permute(0, 3, 1, 2)
    want = F.interpolate(p, size=(gh, gw), mode="bicubic", align_corners=False).permute(0, 2, 3, 1).reshape(-1, D)
    got = out.cpu().numpy()
    assert np.array_equal(got[0], pos[0])
    assert rel_err(got[1:], want.numpy()) < 2e-6


@pytest.mark.parametrize("B,H,W", [(2, 56, 70), (1, 224, 224), (1, 518, 518), (1, 30, 45)])
def test_im2col_is_exact(G, B, H, W):
    img = synth.make_pixels(B, H, W, seed=2)
    p, K, Kp = 14, 588, 640
    gh, gw = H // p, W // p
    out = torch.empty(B * gh * gw, Kp, device=G.dev())
    imd = G.to_gpu(img)
    nat.check(nat.lib().dod_op_im2col(nat.ptr(imd), B, H, W, p, Kp, nat.ptr(out), nat.DOD_F32, nat.stream_ptr()))
    x = torch.from_numpy(img)[:, :, :gh * p, :gw * p].reshape(B, 3, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, K)
    got = out.cpu()
    assert torch.equal(got[:, :K], x) and float(got[:, K:].abs().max()) == 0.0


def _gemm_f32x(A, lda, akm, asb, ash, W, ldw, wkm, wsb, wsh, Cm, ldc, csb, csh, M, N, K, batch, hb, alpha, acc, ksplit):
    nat.check(nat.lib().dod_op_gemm_f32x(nat.ptr(A), lda, akm, asb, ash, nat.ptr(W), ldw, wkm, wsb, wsh, nat.ptr(Cm), ldc, csb, csh,
                                         M, N, K, batch, hb, alpha, acc, ksplit, nat.stream_ptr()))


@pytest.mark.parametrize("akm,wkm", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(1, 4, 64), (100, 50, 768), (257, 64, 257), (333, 95, 130), (1600, 768, 1024), (768, 2, 4112)])
def test_gemm_f32x_operand_layouts(G, M, N, K, akm, wkm):
    """The training step's product forms (train.py:1079-1109): either operand k-major (stored [K, rows]), odd sizes and pitches
    that rule out vector loads, alpha, accumulate, and the row split with atomic accumulate -- against the fp64 product."""
    A, W, C0 = _n("x.A", (M, K)), _n("x.W", (N, K), 0.05), _n("x.C", (M, N))
    ref = torch.from_numpy(A).double() @ torch.from_numpy(W).double().t()
    Ad = G.to_gpu(np.ascontiguousarray(A.T) if akm else A)
    Wd = G.to_gpu(np.ascontiguousarray(W.T) if wkm else W)
    lda, ldw = (M if akm else K), (N if wkm else K)
    out = torch.empty(M, N, device=G.dev())
    _gemm_f32x(Ad, lda, akm, 0, 0, Wd, ldw, wkm, 0, 0, out, N, 0, 0, M, N, K, 1, 1, 1.0, 0, 1)
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 3e-6
    want = torch.from_numpy(C0).double() + 0.5 * ref
    out = G.to_gpu(C0)
    _gemm_f32x(Ad, lda, akm, 0, 0, Wd, ldw, wkm, 0, 0, out, N, 0, 0, M, N, K, 1, 1, 0.5, 1, 1)
    assert rel_err(out.cpu().numpy(), want.numpy()) < 3e-6
    for ks in (2, 7):
        out = G.to_gpu(C0)
        _gemm_f32x(Ad, lda, akm, 0, 0, Wd, ldw, wkm, 0, 0, out, N, 0, 0, M, N, K, 1, 1, 0.5, 1, ks)
        assert rel_err(out.cpu().numpy(), want.numpy()) < 3e-6, ks


@pytest.mark.parametrize("B,H,Q,dh", [(2, 4, 7, 16), (3, 8, 100, 96), (2, 12, 257, 64)])
def test_gemm_f32x_batched_attention_views(G, B, H, Q, dh):
    """The (image, head) batch over strided views of a [B*Q, 3*D] q|k|v buffer, as the self-attention forward and adjoint issue it:
    S = scale q k^T into a [B*H, Q, Qp] scratch, O = P v with v as the k-major operand, dk = dS^T q with both operands k-major."""
    D = H * dh
    qkv = _n("xb.qkv", (B * Q, 3 * D))
    Qp = (Q + 3) // 4 * 4
    t = torch.from_numpy(qkv).double().view(B, Q, 3, H, dh)
    q, k, v = (t[:, :, i].permute(0, 2, 1, 3) for i in range(3))        # [B, H, Q, dh]
    S_ref = 0.25 * q @ k.transpose(-1, -2)
    qd = G.to_gpu(qkv)
    S = torch.zeros(B * H, Q, Qp, device=G.dev())
    ld, qs, ss = 3 * D, Q * 3 * D, Q * Qp
    _gemm_f32x(qd, ld, 0, qs, dh, qd[:, D:], ld, 0, qs, dh, S, Qp, ss * H, ss, Q, Q, dh, B * H, H, 0.25, 0, 1)
    assert rel_err(S[:, :, :Q].cpu().numpy().reshape(B, H, Q, Q), S_ref.numpy()) < 3e-6
    O = torch.empty(B * Q, D, device=G.dev())
    _gemm_f32x(S, Qp, 0, ss * H, ss, qd[:, 2 * D:], ld, 1, qs, dh, O, D, Q * D, dh, Q, dh, Q, B * H, H, 1.0, 0, 1)
    O_ref = (S_ref @ v).permute(0, 2, 1, 3).reshape(B * Q, D)
    assert rel_err(O.cpu().numpy(), O_ref.numpy()) < 3e-6
    dK = torch.empty(B * Q, D, device=G.dev())
    _gemm_f32x(S, Qp, 1, ss * H, ss, qd, ld, 1, qs, dh, dK, D, Q * D, dh, Q, dh, Q, B * H, H, 1.0, 0, 1)
    dK_ref = (S_ref.transpose(-1, -2) @ q).permute(0, 2, 1, 3).reshape(B * Q, D)
    assert rel_err(dK.cpu().numpy(), dK_ref.numpy()) < 3e-6


@pytest.mark.parametrize("M,F,K", [(300, 64, 128), (1370, 1024, 768), (4400, 4096, 1536)])
def test_linear_swiglu_pairs_epilogue(M, F, K):
    """GemmEpi::glu (the SwiGLU gate of Dinov2SwiGLUFFN, modeling_dinov2.py:310-314, evaluated in the weights_in GEMM's epilogue): the weight
    rows arrive interleaved (x1_i, x2_i adjacent), the output is silu(x1) * x2 in F columns.  Against the fp64 evaluation of the same bf16
    operands on the UN-interleaved weight; covers the small-tile, the 256x128 and the 256x256 ping-pong kernels' epilogues."""
    from tests import gpu_util as G
    L = nat.lib()
    A = torch.from_numpy(synth.normal(21, f"glu.A.{M}.{K}", (M, K), 1.0)).cuda().bfloat16()
    W = torch.from_numpy(synth.normal(21, f"glu.W.{F}.{K}", (2 * F, K), 0.05)).cuda().bfloat16()
    bias = torch.from_numpy(synth.normal(21, f"glu.b.{F}", (2 * F,), 0.5)).cuda()
    Wi = torch.stack([W[:F], W[F:]], dim=1).reshape(2 * F, K).contiguous()          # rows 2i = x1_i, 2i + 1 = x2_i
    bi = torch.stack([bias[:F], bias[F:]], dim=1).reshape(2 * F).contiguous()
    out = torch.empty(M, F, dtype=torch.bfloat16, device="cuda")
    nat.check(L.dod_op_linear(nat.DOD_BF16, nat.ptr(A), K, nat.ptr(Wi), K, M, 2 * F, K, nat.ptr(bi), None, None, 0, nat.ptr(out), nat.DOD_BF16, F,
                              nat.ACT["swiglu_pairs"], nat.stream_ptr()))
    z = A.double().cpu() @ W.double().cpu().t() + bias.double().cpu()
    want = torch.nn.functional.silu(z[:, :F]) * z[:, F:]
    assert rel_err(out.float().cpu().numpy(), want.numpy()) < 2 ** -8
