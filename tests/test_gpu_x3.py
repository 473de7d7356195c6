"""Operators of the bf16x3 (parity-gated) mode against fp64 references: split-product GEMM (gemm_x3.hip) and split-product
flash attention (attn_x3.hip).  x = hi + lo with two bf16 halves carries 16 mantissa bits and the lo*lo products (2^-18) are
dropped: results within 3e-5 of the exact product of the fp32 inputs (relative to the output's max)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dinov2_od_amd import _native as nat, synth
from tests.cases import rel_err

pytestmark = pytest.mark.gpu
TOL = 3e-5


def _n(tag, shape, std=1.0):
    return torch.from_numpy(synth.normal(9, tag, shape, std))


def _pair(x):
    """fp32 CUDA [rows, cols] -> pair layout [rows, 2*cols] bf16 through dod_op_split_pair"""
    rows, cols = x.shape
    out = torch.empty(rows, 2 * cols, dtype=torch.bfloat16, device=x.device)
    nat.check(nat.lib().dod_op_split_pair(nat.ptr(x), x.stride(0), rows, cols, nat.ptr(out), nat.stream_ptr()))
    return out


def test_split_pair_layout():
    x = _n("sp", (37, 96), 3.0).cuda()
    p = _pair(x).float().cpu()
    hi, lo = p[:, :96], p[:, 96:]
    assert torch.equal(hi, x.cpu().bfloat16().float())
    assert torch.equal(lo, (x.cpu() - hi).bfloat16().float())
    assert rel_err((hi + lo).numpy(), x.cpu().numpy()) < 2 ** -16


@pytest.mark.parametrize("M,N,K", [(64, 128, 64), (1000, 384, 768), (2740 + 5, 2304, 768), (4115, 768, 3072), (300, 100, 96), (513, 520, 608)])
def test_linear_x3_all_epilogues(M, N, K):
    L = nat.lib()
    A, W = _n(f"x3.A.{M}.{K}", (M, K)), _n(f"x3.W.{N}.{K}", (N, K), 0.05)
    bias, scale, resid = _n("x3.b", (N,)), 1 + _n("x3.s", (N,), 0.1), _n("x3.r", (M, N))
    ref = A.double() @ W.double().t()
    A2, W2 = _pair(A.cuda()), _pair(W.cuda())

    def run(bias=None, scale=None, resid=None, act="none", layout=0):
        out = torch.empty(M, N if layout < 2 else 2 * N, dtype=torch.float32 if layout == 0 else torch.bfloat16, device="cuda")
        nat.check(L.dod_op_linear_x3(nat.ptr(A2), nat.ptr(W2), M, N, K, nat.ptr(bias), nat.ptr(scale), nat.ptr(resid),
                                     N if resid is not None else 0, nat.ptr(out), layout, out.shape[1], nat.ACT[act], nat.stream_ptr()))
        return out

    assert rel_err(run().cpu().numpy(), ref.numpy()) < TOL
    want = (ref + bias.double()) * scale.double() + resid.double()
    assert rel_err(run(bias.cuda(), scale.cuda(), resid.cuda()).cpu().numpy(), want.numpy()) < TOL
    want = F.gelu(ref + bias.double())
    o = run(bias.cuda(), act="gelu", layout=2).float().cpu()          # pair-layout output: hi + lo reconstructs the fp32 value
    assert rel_err((o[:, :N] + o[:, N:]).numpy(), want.numpy()) < TOL
    assert rel_err(run(bias.cuda(), act="gelu", layout=1).float().cpu().numpy(), want.numpy()) < 2 ** -7


@pytest.mark.parametrize("B,N,heads", [(1, 17, 2), (2, 64, 1), (2, 257, 6), (1, 1370, 12), (3, 130, 2)])
def test_attention_x3(B, N, heads):
    D = heads * 64
    qkv = _n(f"x3.qkv.{N}.{heads}", (B * N, 3 * D), 1.5)
    q2 = _pair(qkv.cuda())
    ctx2 = torch.empty(B * N, 2 * D, dtype=torch.bfloat16, device="cuda")
    nat.check(nat.lib().dod_op_attention_x3(nat.ptr(q2), nat.ptr(ctx2), B, N, heads, 0.125, nat.stream_ptr()))
    c = ctx2.float().cpu()
    got = (c[:, :D] + c[:, D:]).view(B, N, D)
    q, k, v = [t.view(B, N, heads, 64).transpose(1, 2).double() for t in qkv.view(B, N, 3 * D).split(D, dim=-1)]
    want = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).transpose(1, 2).reshape(B, N, D)
    assert rel_err(got.numpy(), want.numpy()) < TOL
