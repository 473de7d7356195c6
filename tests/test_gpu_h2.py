"""Operators of the fp16x2 (parity-gated) mode: the H2 operand format (fp16 + e4m3 main / remainder, dod_common.h) and the H2 GEMM
(gemm_pp.hip: fp16 main product + both cross terms as ONE block-scaled e4m3 MFMA).  The packing is checked bit for bit against
torch's fp16 / float8_e4m3fn casts; the GEMM against (a) the exact product of the DECODED operands (what the kernel is defined to
compute: pins the E8M0 block-scale semantics and the in-kernel fp16 -> e4m3 conversion of the weights, leaving fp32 accumulation
only) and (b) the exact product of the fp32 inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dinov2_od_amd import _native as nat, synth
from tests.cases import rel_err

pytestmark = pytest.mark.gpu


def _n(tag, shape, std=1.0):
    return torch.from_numpy(synth.normal(11, tag, shape, std))


def pack(x, weight=False):
    rows, cols = x.shape
    out = torch.empty(rows, (3 if weight else 4) * cols, dtype=torch.uint8, device="cuda")
    wexp = torch.empty(rows, dtype=torch.uint8, device="cuda") if weight else None
    nat.check(nat.lib().dod_op_split_h2(nat.ptr(x), x.stride(0), rows, cols, nat.ptr(out), nat.ptr(wexp), nat.stream_ptr()))
    return out, wexp


_f8 = lambda t: t.contiguous().view(torch.float8_e4m3fn).double()


def decode(buf, cols, wexp=None):
    """-> (fp16 part, e4m3 main, e4m3 remainder) as float64 with the scales undone.  Activation rows (4*cols bytes) store the main
    bytes; weight rows (3*cols bytes) do not: the kernel derives e4m3(h 2^e) from the fp16 part, and so does this."""
    b = buf.cpu()
    h = b[:, :2 * cols].contiguous().view(torch.float16).double()
    if wexp is None:
        grp = b[:, 2 * cols:].contiguous().view(-1, cols // 16, 2, 16)
        return h, _f8(grp[:, :, 0, :].reshape(-1, cols)), _f8(grp[:, :, 1, :].reshape(-1, cols)) / 2048.0
    sc = torch.exp2(127 - wexp.cpu().double())[:, None]
    main = (h * sc).float().to(torch.float8_e4m3fn).double() / sc
    return h, main, _f8(b[:, 2 * cols:]) / (sc * 2048.0)


def test_split_h2_layout_bit_exact():
    x = _n("h2.sp", (37, 96), 3.0)
    x[0, :4] = torch.tensor([1e5, -7e4, 3e-6, 0.0])          # beyond fp16: clamped; tiny: e4m3 flushes
    buf, _ = pack(x.cuda())
    h, m8, r8 = decode(buf, 96)
    xc = x.clamp(-65504, 65504)
    hh = xc.half()
    assert torch.equal(h, hh.double())
    assert torch.equal(m8, hh.float().clamp(-448, 448).to(torch.float8_e4m3fn).double())
    assert torch.equal(r8 * 2048, ((xc - hh.float()) * 2048).clamp(-448, 448).to(torch.float8_e4m3fn).double())
    w = _n("h2.spw", (50, 64), 0.02)
    w[3] = 0
    bufw, wexp = pack(w.cuda(), weight=True)
    h, m8, r8 = decode(bufw, 64, wexp)
    hh = w.half()
    assert torch.equal(h, hh.double())
    amax = hh.float().abs().amax(1)
    e = torch.where(amax > 0, torch.floor(torch.log2(448.0 / amax.double())), torch.zeros(50, dtype=torch.float64))
    assert torch.equal(127 - wexp.cpu().double(), e)
    sc = torch.exp2(e)[:, None]
    assert torch.equal(r8, ((w - hh.float()).double() * sc * 2048).float().to(torch.float8_e4m3fn).double() / (sc * 2048))


@pytest.mark.parametrize("M,N,K", [(64, 128, 64), (1000, 384, 768), (2740 + 5, 2304, 768), (4115, 768, 3072), (300, 100, 96), (513, 520, 608)])
def test_linear_h2_all_epilogues(M, N, K):
    L = nat.lib()
    A, W = _n(f"h2.A.{M}.{K}", (M, K)), _n(f"h2.W.{N}.{K}", (N, K), 0.05)
    bias, scale, resid = _n("h2.b", (N,)), 1 + _n("h2.s", (N,), 0.1), _n("h2.r", (M, N))
    exact = A.double() @ W.double().t()
    Ab, _ = pack(A.cuda())
    Wb, wexp = pack(W.cuda(), weight=True)
    ah, am, ar = decode(Ab, K)
    wh, wm, wr = decode(Wb, K, wexp)
    defined = ah @ wh.t() + am @ wr.t() + ar @ wm.t()          # what the kernel is defined to compute

    def run(bias=None, scale=None, resid=None, act="none", layout=0):
        width = {0: N, 1: N, 2: 2 * N, 3: 2 * N}[layout]
        out = torch.empty(M, width, dtype=torch.float32 if layout == 0 else torch.bfloat16, device="cuda")
        nat.check(L.dod_op_linear_h2(nat.ptr(Ab), nat.ptr(Wb), nat.ptr(wexp), M, N, K, nat.ptr(bias), nat.ptr(scale), nat.ptr(resid),
                                     N if resid is not None else 0, nat.ptr(out), layout, width, nat.ACT[act], nat.stream_ptr()))
        return out

    got = run().cpu().numpy()
    assert rel_err(got, defined.numpy()) < 2e-6, "block-scale semantics / fp32 accumulation"
    assert rel_err(got, exact.numpy()) < 5e-5
    want = (exact + bias.double()) * scale.double() + resid.double()
    assert rel_err(run(bias.cuda(), scale.cuda(), resid.cuda()).cpu().numpy(), want.numpy()) < 5e-5
    want = F.gelu(exact + bias.double())
    o = run(bias.cuda(), act="gelu", layout=2).float().cpu()
    assert rel_err((o[:, :N] + o[:, N:]).numpy(), want.numpy()) < 5e-5
    if N % 32 == 0:                                             # H2 rows out: decode and compare with the packing of the fp32 result
        o3 = run(bias.cuda(), act="gelu", layout=3).view(torch.uint8)
        h, m8, r8 = decode(o3, N)
        assert rel_err(h.numpy(), want.numpy()) < 2 ** -11
        assert rel_err((h + r8).numpy(), want.numpy()) < 5e-5
        assert rel_err(m8.numpy(), want.numpy()) < 2 ** -3


def test_linear_h2_outliers_degrade_gracefully():
    """Activations beyond the e4m3 range (|x| > 448: the cross terms see them clamped) and beyond the fp16 range (clamped at 65504):
    no NaN / inf; the error stays at the fp16-single-product level for the affected terms (the mode degrades, it does not overflow)."""
    L = nat.lib()
    M, N, K = 512, 256, 768
    A, W = _n("h2.o.A", (M, K)), _n("h2.o.W", (N, K), 0.05)
    A[::7, ::13] *= 900.0                       # up to ~3000: beyond e4m3, inside fp16
    A[5, 17] = 1.0e5                            # beyond fp16: clamped to 65504 by the packer
    Ab, _ = pack(A.cuda())
    Wb, wexp = pack(W.cuda(), weight=True)
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    nat.check(L.dod_op_linear_h2(nat.ptr(Ab), nat.ptr(Wb), nat.ptr(wexp), M, N, K, None, None, None, 0, nat.ptr(out), 0, N, 0, nat.stream_ptr()))
    got = out.cpu()
    assert torch.isfinite(got).all()
    exact = A.clamp(-65504, 65504).double() @ W.double().t()
    assert rel_err(got.numpy(), exact.numpy()) < 1e-3       # 2^-11-class: the clamped entries' cross terms are lost, nothing else
    rows = [r for r in range(M) if r % 7 != 0 and r != 5]   # rows without outliers keep the full accuracy
    assert rel_err(got.numpy()[rows], exact.numpy()[rows]) < 5e-5
