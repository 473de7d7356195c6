"""K-split paths of the 256x256-tile GEMMs (gemm_pp.hip gemm_tail_split): (a) a shape whose tile count leaves a short last round (516
tiles on 256 CUs) takes the main launch + K-split remainder + reduce path; (b) an underfilled single round (99 tiles: the
compensated kernels have no smaller tile) is K-split whole.  Results against fp64, every epilogue that the forward uses on these
GEMMs (in-place fp32 residual with LayerScale; bf16; bf16 pair; H2 rows with GELU)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dinov2_od_amd import _native as nat, synth
from tests.cases import rel_err
from tests.test_gpu_h2 import pack as pack_h2, decode as decode_h2
from tests.test_gpu_x3 import _pair

pytestmark = pytest.mark.gpu
N, K = 768, 768
M_TAIL = 171 * 256 + 40        # 172 x 3 = 516 tiles: two rounds of 256 + 4 -> rows 170*256.. are the remainder
M_UNDER = 32 * 256 + 8         # 33 x 3 = 99 tiles: K-split two ways (compensated kernels; the plain bf16 GEMM has a 256x128 tile for this)


def _n(tag, shape, std=1.0):
    return torch.from_numpy(synth.normal(13, tag, shape, std))


@pytest.fixture(scope="module", params=[M_TAIL, M_UNDER], ids=["tail", "underfilled"])
def data(request):
    global M
    M = request.param
    nat.check(nat.lib().dod_reserve_gemm_scratch(64 << 20))
    # force the split path for every qualifying shape in THIS module only (dod_test_set_option("tailsplit"); -1 hands the shipped heuristic
    # back): the rest of the single-process GPU suite -- the parity gates in test_gpu_forward.py -- runs with production defaults
    nat.set_option("tailsplit", 2)
    request.addfinalizer(lambda: nat.set_option("tailsplit", -1))
    A, W = _n(f"ts.A.{M}", (M, K)), _n("ts.W", (N, K), 0.05)
    bias, scale, resid = _n("ts.b", (N,)), 1 + _n("ts.s", (N,), 0.1), _n(f"ts.r.{M}", (M, N))
    dev = {"bias": bias.cuda(), "scale": scale.cuda()}        # device copies that outlive every launch below
    return A, W, bias, scale, resid, A.double() @ W.double().t(), dev


def _took_split(before):
    return nat.lib().dod_test_counter(b"tail_splits") > before


def _rows():
    return slice(max(0, M - 600), M)      # straddles the main / remainder cut of the tail case


def test_tail_split_plain_bf16(data):
    A, W, bias, scale, resid, exact, dev = data
    L = nat.lib()
    n0 = L.dod_test_counter(b"tail_splits")
    Ab, Wb = A.cuda().bfloat16(), W.cuda().bfloat16()
    ref = Ab.double().cpu() @ Wb.double().cpu().t()
    x = resid.cuda().clone()
    nat.check(L.dod_op_linear(1, nat.ptr(Ab), K, nat.ptr(Wb), K, M, N, K, nat.ptr(dev["bias"]), nat.ptr(dev["scale"]), nat.ptr(x), N, nat.ptr(x), 0, N, 0, nat.stream_ptr()))
    want = (ref + bias.double()) * scale.double() + resid.double()
    assert rel_err(x.cpu().numpy(), want.numpy()) < 2e-6
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    nat.check(L.dod_op_linear(1, nat.ptr(Ab), K, nat.ptr(Wb), K, M, N, K, nat.ptr(dev["bias"]), None, None, 0, nat.ptr(out), 1, N, 0, nat.stream_ptr()))
    assert rel_err(out.float().cpu().numpy()[_rows()], (ref + bias.double()).numpy()[_rows()]) < 2 ** -8
    assert M == M_UNDER or _took_split(n0 + 1)


def test_tail_split_x3(data):
    A, W, bias, scale, resid, exact, dev = data
    L = nat.lib()
    n0 = L.dod_test_counter(b"tail_splits")
    A2, W2 = _pair(A.cuda()), _pair(W.cuda())
    x = resid.cuda().clone()
    nat.check(L.dod_op_linear_x3(nat.ptr(A2), nat.ptr(W2), M, N, K, nat.ptr(dev["bias"]), nat.ptr(dev["scale"]), nat.ptr(x), N, nat.ptr(x), 0, N, 0, nat.stream_ptr()))
    want = (exact + bias.double()) * scale.double() + resid.double()
    assert rel_err(x.cpu().numpy(), want.numpy()) < 3e-5
    o = torch.empty(M, 2 * N, dtype=torch.bfloat16, device="cuda")
    nat.check(L.dod_op_linear_x3(nat.ptr(A2), nat.ptr(W2), M, N, K, nat.ptr(dev["bias"]), None, None, 0, nat.ptr(o), 2, 2 * N, nat.ACT["gelu"], nat.stream_ptr()))
    o = o.float().cpu()
    assert rel_err((o[:, :N] + o[:, N:]).numpy()[_rows()], F.gelu(exact + bias.double()).numpy()[_rows()]) < 3e-5
    assert _took_split(n0 + 1)


def test_tail_split_h2(data):
    A, W, bias, scale, resid, exact, dev = data
    L = nat.lib()
    n0 = L.dod_test_counter(b"tail_splits")
    Ab, _ = pack_h2(A.cuda())
    Wb, wexp = pack_h2(W.cuda(), weight=True)
    x = resid.cuda().clone()
    nat.check(L.dod_op_linear_h2(nat.ptr(Ab), nat.ptr(Wb), nat.ptr(wexp), M, N, K, nat.ptr(dev["bias"]), nat.ptr(dev["scale"]), nat.ptr(x), N, nat.ptr(x), 0, N, 0, nat.stream_ptr()))
    want = (exact + bias.double()) * scale.double() + resid.double()
    assert rel_err(x.cpu().numpy(), want.numpy()) < 5e-5
    o3 = torch.empty(M, 2 * N, dtype=torch.bfloat16, device="cuda")
    nat.check(L.dod_op_linear_h2(nat.ptr(Ab), nat.ptr(Wb), nat.ptr(wexp), M, N, K, nat.ptr(dev["bias"]), None, None, 0, nat.ptr(o3), 3, 2 * N, nat.ACT["gelu"], nat.stream_ptr()))
    h, m8, r8 = decode_h2(o3.view(torch.uint8)[_rows()], N)
    assert rel_err((h + r8).numpy(), F.gelu(exact + bias.double()).numpy()[_rows()]) < 5e-5
    assert _took_split(n0 + 1)
