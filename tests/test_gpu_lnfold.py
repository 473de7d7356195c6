"""`-m gpu`: LayerNorm folded into the GEMMs around it (the fast modes' schedule of modeling_dinov2.py:361-380 since round 4;
csrc/dod_common.h GemmEpi::ln_*, csrc/gemm_epi.h).

    LN(x) W^T + b  =  rstd (x W'^T - mean c) + b',     W' = W diag(gamma),  c[n] = sum_k W'[n][k],  b' = b + W beta

  * producer -- the in-place residual epilogue of out-proj / fc2 additionally writes the new rows in the next GEMM's operand format and
    their (sum, centred sum of squares) per 128-column group; dod_op_ln_finalize merges the groups: rows bit-equal to the unfolded
    launch, statistics against float64, operand copy = the family's own rounding of the fp32 row;
  * consumer -- QKV / fc1 on the residual rows themselves: against LN(x) W^T + b in float64 (the compensated families) or against the
    exact arithmetic of the rounded operands (single-pass bf16);
at small ragged shapes (partial tiles, a partial 128-column group) and at the timed shape (M = 87 680: the 256x256 kernels, the tail
split of fc2), in every operand family."""
import ctypes as C

import numpy as np
import pytest
import torch

from dinov2_od_amd import _native as nat
from tests.cases import rel_err
from tests.test_gpu_h2 import pack as pack_h2, decode as decode_h2
from tests.test_gpu_x3 import _pair

pytestmark = pytest.mark.gpu
EPS = 1e-6
FAM = {"bf16": 1, "bf16x3": 3, "fp16x2": 4}


def _rows(M, D, seed):
    """residual-like rows: per-row offset (|mean| ~ std) and per-row scale, so that the mean term and rstd both matter"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(M, D, device="cuda", generator=g)
    return x * (0.5 + torch.rand(M, 1, device="cuda", generator=g) * 3) + torch.randn(M, 1, device="cuda", generator=g) * 1.5


def _operand(x, fam, weight=False):
    """fp32 [rows, K] -> (operand buffer, wexp or None) of the family"""
    if fam == "bf16":
        return x.bfloat16().contiguous(), None
    if fam == "bf16x3":
        return _pair(x.contiguous()), None
    return pack_h2(x.contiguous(), weight=weight)


def _decode_rows(buf, fam, D, rows):
    """operand copy of activation rows -> float64 values it represents"""
    if fam == "bf16":
        return buf[rows].double().cpu()
    if fam == "bf16x3":
        b = buf[rows].double().cpu()
        return b[:, :D] + b[:, D:]
    h, _, r8 = decode_h2(buf[rows].contiguous().view(torch.uint8), D)
    return h + r8


def _sample(M):
    g = torch.Generator().manual_seed(3)
    idx = torch.cat([torch.arange(0, min(M, 200)), torch.arange(max(0, M - 400), M), torch.randint(0, M, (400,), generator=g)])
    return torch.unique(idx).cuda()


SHAPES = [(300, 384, 384), (517, 192, 640), (2740, 768, 768), (64 * 1370, 768, 768), (64 * 1370, 768, 3072), (8224, 768, 3072)]


@pytest.mark.parametrize("shifted", [True, False], ids=["shift", "noshift"])
@pytest.mark.parametrize("fam", list(FAM))
@pytest.mark.parametrize("M,D,K", SHAPES, ids=[f"{m}x{d}x{k}" for m, d, k in SHAPES])
def test_residual_epilogue_emits_operand_rows_and_statistics(fam, M, D, K, shifted):
    """out-proj (K = D) / fc2 (K = 4D) with the in-place fp32 residual + LayerScale epilogue (modeling_dinov2.py:367-370, 377-380)"""
    L = nat.lib()
    nat.check(L.dod_reserve_gemm_scratch(64 << 20))      # as dod_finalize_weights does: the K >= 2048 launches take the tail split (gemm_pp.hip)
    g = torch.Generator(device="cuda").manual_seed(M + 7 * D + K)
    A = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(D, K, device="cuda", generator=g) * 0.05
    bias = torch.randn(D, device="cuda", generator=g)
    ls = 1 + 0.1 * torch.randn(D, device="cuda", generator=g)
    x0 = _rows(M, D, 11)
    Aop, _ = _operand(A, fam)
    Wop, wexp = _operand(W, fam, weight=True)
    # unfolded launch (the round-3 epilogue)
    x_ref = x0.clone()
    none = nat.DodLnFold()
    nat.check(L.dod_op_linear_ln(FAM[fam], nat.ptr(Aop), nat.ptr(Wop), nat.ptr(wexp), M, D, K, nat.ptr(bias), nat.ptr(ls), nat.ptr(x_ref), D,
                                 nat.ptr(x_ref), 0, D, 0, C.byref(none), nat.stream_ptr()))
    # folded producer
    x = x0.clone()
    npart = (D + 127) // 128
    part = torch.full((M, npart, 2), float("nan"), device="cuda")
    op = torch.zeros(M, D if fam == "bf16" else 2 * D, dtype=torch.bfloat16, device="cuda")
    # the shift: as in the forward, the statistics the row had BEFORE the update (here: of x0); `shifted=False` runs with no shift at all
    stats = torch.zeros(M, 2, device="cuda")
    if shifted:
        stats[:, 0] = x0.mean(-1)
    ln = nat.DodLnFold(None, None, op.data_ptr(), part.data_ptr(), stats.data_ptr() if shifted else None, None, None, 0.0)
    nat.check(L.dod_op_linear_ln(FAM[fam], nat.ptr(Aop), nat.ptr(Wop), nat.ptr(wexp), M, D, K, nat.ptr(bias), nat.ptr(ls), nat.ptr(x), D,
                                 nat.ptr(x), 0, D, 0, C.byref(ln), nat.stream_ptr()))
    nat.check(L.dod_op_ln_finalize(nat.ptr(part), M, D, EPS, nat.ptr(stats), nat.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(x, x_ref), "the folded producer must not change the residual rows"
    assert not torch.isnan(part).any(), "a (row, group) statistic was never written"
    xd = x.double()
    mean = xd.mean(-1)
    rstd = 1.0 / torch.sqrt(((xd - mean[:, None]) ** 2).mean(-1) + EPS)
    e_mean = float((stats[:, 0].double() - mean).abs().max() / xd.abs().max())
    e_rstd = float(((stats[:, 1].double() - rstd) / rstd).abs().max())
    print(f"{fam} M={M} D={D} K={K} shifted={shifted}: mean {e_mean:.1e} rstd {e_rstd:.1e}")
    # one-pass sums: with the shift nothing cancels (fp32 rounding of a 768-term sum); without it the variance loses log2(1 + mean^2 / var)
    # bits -- these rows have |mean| ~ std, so still ~1e-6
    assert e_mean < 3e-7 and e_rstd < (3e-6 if shifted else 1e-5)
    rows = _sample(M)
    got = _decode_rows(op, fam, D, rows)
    want = x[rows].cpu()
    if fam == "bf16":
        assert torch.equal(got, want.bfloat16().double())
    else:
        assert rel_err(got.numpy(), want.double().numpy()) < (2 ** -16 if fam == "bf16x3" else 2e-5)


CONS = [(300, 1152, 384, "none"), (517, 520, 192, "gelu"), (2740, 2304, 768, "none"), (64 * 1370, 2304, 768, "none"), (64 * 1370, 3072, 768, "gelu"),
        (8224, 3072, 768, "gelu"),
        # grids of one / two rounds of 256x256 tiles plus a few: inside / below the window in which the last round's rows run as a launch of
        # their own (gemm_bf16.hip, round 4)
        (8224, 2304, 768, "none"), (8 * 1370, 3072, 768, "gelu")]


@pytest.mark.parametrize("fam", list(FAM))
@pytest.mark.parametrize("M,N,D,act", CONS, ids=[f"{m}x{n}x{d}-{a}" for m, n, d, a in CONS])
def test_folded_consumer_equals_layernorm_then_linear(fam, M, N, D, act):
    """QKV (N = 3D) / fc1 (N = 4D, exact-erf GELU) reading the residual rows themselves (modeling_dinov2.py:361-366, 373-376)"""
    L = nat.lib()
    g = torch.Generator(device="cuda").manual_seed(M + 3 * N + D)
    x = _rows(M, D, 5)
    gamma = 1 + 0.1 * torch.randn(D, device="cuda", generator=g)
    beta = 0.1 * torch.randn(D, device="cuda", generator=g)
    W = torch.randn(N, D, device="cuda", generator=g) * 0.05
    b = torch.randn(N, device="cuda", generator=g) * 0.1
    Wp = (W * gamma).contiguous()
    bp = (b.double() + W.double() @ beta.double()).float()
    csum = (Wp.bfloat16().float() if fam == "bf16" else Wp).double().sum(-1).float()
    # first-block form: rowstats writes the operand copy of x and (mean, rstd)
    xop = torch.zeros(M, D if fam == "bf16" else 2 * D, dtype=torch.bfloat16, device="cuda")
    stats = torch.empty(M, 2, device="cuda")
    nat.check(L.dod_op_rowstats(nat.ptr(x), M, D, EPS, nat.ptr(xop), FAM[fam], nat.ptr(stats), nat.stream_ptr()))
    Wop, wexp = _operand(Wp, fam, weight=True)
    out_layout = 1 if fam == "bf16" else (3 if (fam == "fp16x2" and act == "gelu" and N % 32 == 0) else 2)
    out = torch.zeros(M, N if fam == "bf16" else 2 * N, dtype=torch.bfloat16, device="cuda")
    ln = nat.DodLnFold(stats.data_ptr(), csum.data_ptr(), None, None, None, None, None, 0.0)
    cuts0 = L.dod_test_counter(b"rem_cuts")
    nat.check(L.dod_op_linear_ln(FAM[fam], nat.ptr(xop), nat.ptr(Wop), nat.ptr(wexp), M, N, D, nat.ptr(bp), None, None, 0, nat.ptr(out), out_layout,
                                 N if fam == "bf16" else 2 * N, nat.ACT[act], C.byref(ln), nat.stream_ptr()))
    torch.cuda.synchronize()
    if fam == "bf16" and torch.cuda.get_device_properties(0).multi_processor_count == 256:
        # 33 x 9 = 297 tiles = one round + 41: cut.  43 x 12 = 516 = two rounds + 4: below the window (lone tiles finish early), not cut
        want_cut = 1 if (M, N) == (8224, 2304) else 0
        assert L.dod_test_counter(b"rem_cuts") == cuts0 + want_cut, "cut-off last round: window is CUs / 16 <= remainder <= CUs / 6"
    rows = _sample(M)
    xd = x[rows].double().cpu()
    mean = xd.mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(((xd - mean) ** 2).mean(-1, keepdim=True) + EPS)
    if fam == "bf16":      # the exact arithmetic of the rounded operands; the bf16 output rounding on top
        acc = x[rows].bfloat16().double().cpu() @ Wp.bfloat16().double().cpu().t()
        want = (acc - mean * csum.double().cpu()) * rstd + bp.double().cpu()
        got = out[rows].double().cpu()
        tol = 2 ** -8
    else:
        y = (xd - mean) * rstd * gamma.double().cpu() + beta.double().cpu()
        want = y @ W.double().cpu().t() + b.double().cpu()
        if out_layout == 3:
            h, _, r8 = decode_h2(out[rows].contiguous().view(torch.uint8), N)
            got = h + r8
        else:
            o = out[rows].double().cpu()
            got = o[:, :N] + o[:, N:]
        tol = 5e-5 if fam == "bf16x3" else 1e-4      # H2: 5e-5-grade on O(1) operands; these rows carry an offset ~ their spread
    if act == "gelu":
        want = torch.nn.functional.gelu(want)
    err = rel_err(got.numpy(), want.numpy())
    print(f"{fam} consumer M={M} N={N} D={D} {act}: {err:.2e}")
    assert err < tol
    if fam == "bf16":      # and the whole thing against the float64 LayerNorm + linear: the mode's inherent bf16 distance, not more
        y = (xd - mean) * rstd * gamma.double().cpu() + beta.double().cpu()
        full = y @ W.double().cpu().t() + b.double().cpu()
        if act == "gelu":
            full = torch.nn.functional.gelu(full)
        assert rel_err(got.numpy(), full.numpy()) < 3e-2


@pytest.mark.parametrize("fam", list(FAM))
@pytest.mark.parametrize("M,D,N", [(517, 384, 1152), (2740, 768, 2304), (64 * 1370, 768, 2304), (8224, 768, 3072), (8224, 768, 2304), (8 * 1370, 768, 3072)],
                         ids=lambda v: str(v))
def test_consumer_finishes_the_statistics_itself(fam, M, D, N):
    """What the forward runs since round 4: a residual GEMM (producer) leaves group sums relative to the rows' previous mean; the NEXT GEMM
    (consumer) turns them into (mean, rstd) in its own epilogue -- no launch merges the groups -- and its column-0 tiles publish them.
    Bit-equal to the two-step form (dod_op_ln_finalize, then a consumer on final statistics), and the published statistics equal that
    kernel's."""
    L = nat.lib()
    g = torch.Generator(device="cuda").manual_seed(M + D + N)
    K = D
    A = torch.randn(M, K, device="cuda", generator=g)
    Wo = torch.randn(D, K, device="cuda", generator=g) * 0.05
    x0 = _rows(M, D, 23)
    Aop, _ = _operand(A, fam)
    Woop, woexp = _operand(Wo, fam, weight=True)
    npart = (D + 127) // 128
    part = torch.empty(M, npart, 2, device="cuda")
    xop = torch.zeros(M, D if fam == "bf16" else 2 * D, dtype=torch.bfloat16, device="cuda")
    shift = torch.zeros(M, 2, device="cuda")
    shift[:, 0] = x0.mean(-1)
    x = x0.clone()
    ln = nat.DodLnFold(None, None, xop.data_ptr(), part.data_ptr(), shift.data_ptr(), None, None, 0.0)
    nat.check(L.dod_op_linear_ln(FAM[fam], nat.ptr(Aop), nat.ptr(Woop), nat.ptr(woexp), M, D, K, None, None, nat.ptr(x), D, nat.ptr(x), 0, D, 0,
                                 C.byref(ln), nat.stream_ptr()))
    W = torch.randn(N, D, device="cuda", generator=g) * 0.05
    b = torch.randn(N, device="cuda", generator=g) * 0.1
    csum = (W.bfloat16().float() if fam == "bf16" else W).double().sum(-1).float()
    Wop, wexp = _operand(W, fam, weight=True)
    ldo = N if fam == "bf16" else 2 * N
    lay = 1 if fam == "bf16" else 2
    # (a) two steps
    st_a = shift.clone()
    nat.check(L.dod_op_ln_finalize(nat.ptr(part), M, D, EPS, nat.ptr(st_a), nat.stream_ptr()))
    out_a = torch.zeros(M, ldo, dtype=torch.bfloat16, device="cuda")
    ln_a = nat.DodLnFold(st_a.data_ptr(), csum.data_ptr(), None, None, None, None, None, 0.0)
    nat.check(L.dod_op_linear_ln(FAM[fam], nat.ptr(xop), nat.ptr(Wop), nat.ptr(wexp), M, N, D, nat.ptr(b), None, None, 0, nat.ptr(out_a), lay, ldo, 0,
                                 C.byref(ln_a), nat.stream_ptr()))
    # (b) the consumer finishes the statistics
    st_b = torch.full((M, 2), float("nan"), device="cuda")
    out_b = torch.zeros(M, ldo, dtype=torch.bfloat16, device="cuda")
    ln_b = nat.DodLnFold(shift.data_ptr(), csum.data_ptr(), None, None, None, part.data_ptr(), st_b.data_ptr(), EPS)
    nat.check(L.dod_op_linear_ln(FAM[fam], nat.ptr(xop), nat.ptr(Wop), nat.ptr(wexp), M, N, D, nat.ptr(b), None, None, 0, nat.ptr(out_b), lay, ldo, 0,
                                 C.byref(ln_b), nat.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(st_a, st_b), "published statistics differ from the finalize kernel's"
    assert torch.equal(out_a, out_b)
