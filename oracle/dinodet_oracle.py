"""CPU ORACLE for the dino_detector forward path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (torch CPU tensor ops; no import of
the reference, of `transformers`, or of `torch.nn` modules) of the arithmetic of

    DINOv2ObjectDetector.forward          dino_detector/models/detector.py:58-69
    DINOv2Backbone.forward                dino_detector/models/dinov2_backbone.py:58-67
    Dinov2Model / Embeddings / Layer      site-packages/transformers/models/dinov2/modeling_dinov2.py
                                          (transformers 5.15.0; un-vendored, unpinned dependency
                                          of the reference: requirements.txt:3)
    LoraLinear.forward                    dino_detector/utils.py:68-70
    DETRDecoder.forward                   dino_detector/models/detr_decoder.py:47-83
    DeformableDecoderLayer / Attention    dino_detector/models/deformable_attention.py:53-268
    nn.TransformerDecoder branch          dino_detector/models/detr_decoder.py:28-35,62-69
                                          (torch.nn.TransformerDecoderLayer, post-norm, ReLU)

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it; the product path (dinov2_od_amd) never does.

PINNING: the reference has no tests/golden vectors of its own (SURVEY.md section 4:
"parity unpinned" at the transformers boundary).  This oracle is pinned against
outputs of the reference itself, imported once in the authoring container by
`tests/golden/make_goldens.py`; the resulting vectors are committed under
`tests/golden/*.npz` and checked by `tests/test_oracle_golden.py`.

`dtype=torch.float64` evaluates the same arithmetic in double precision (used to
tell which of two fp32 implementations is closer to the exact result).
`emulate_bf16=True` rounds GEMM/attention operands to bf16 at the points where the
HIP fast path does (DESIGN.md "precision modes"); accumulation stays fp32.
`emulate_bf16="fp8"` additionally fake-quantises the operands of the linears the fp8 mode
runs on e4m3 MFMA (block scales: one power of two per 32 elements along K on both operands; torch.float8_e4m3fn rounding).
`emulate_bf16="fp16x2"` evaluates the four linears of every backbone block with the H2 operand scheme of the fp16x2 mode
(dinov2_od_amd/csrc/dod_common.h: x = h + l, h = fp16(x); x.w ~ h_x h_w + e4m3(h_x) e4m3(l_w 2^(e+11)) 2^-(e+11)
+ e4m3(l_x 2^11) e4m3(h_w 2^e) 2^-(e+11), e the weight row's power-of-two scale) and everything else exactly: the scheme's own
distance from the reference, measurable without a GPU.
"""
import math
import torch
import torch.nn.functional as F

from dinov2_od_amd.config import BackboneConfig, DecoderConfig      # shape containers only: no arithmetic of the product is imported


def spatial_factor(hw):
    """K16, the oracle's own restatement of dino_detector/models/deformable_attention.py:241-256: the token count hw (CLS included) is
    used as is when it is a perfect square, else split as (i, hw / i) with i the largest divisor <= floor(sqrt(hw)) -- 257 -> (1, 257),
    1370 -> (10, 137).  (i = 1 always divides, so the reference's "approximate" fallback is unreachable.)"""
    side = math.isqrt(hw)
    if side * side == hw:
        return side, side
    i = side
    while hw % i:
        i -= 1
    return i, hw // i


def _t(x, dtype):
    if isinstance(x, torch.Tensor):
        return x.detach().to("cpu", dtype)
    return torch.as_tensor(x).to(dtype)


def _bf(x):
    return x.to(torch.bfloat16).to(x.dtype)


class _SD:
    """state-dict view with prefix + dtype conversion cache"""

    def __init__(self, sd, dtype):
        self.sd, self.dtype, self.cache = sd, dtype, {}

    def __call__(self, key):
        if key not in self.cache:
            self.cache[key] = _t(self.sd[key], self.dtype)
        return self.cache[key]

    def has(self, key):
        return key in self.sd


def _q8(t):
    """per-row OCP e4m3 fake quantisation as the HIP fp8 path defines it (gemm_fp8.hip): scale = amax / 448 (1 for an
    all-zero row), q = rne_e4m3(t * (1 / scale)); returns the dequantised values q * scale"""
    amax = t.abs().amax(dim=-1, keepdim=True)
    sc = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    return (t * (1.0 / sc)).to(torch.float8_e4m3fn).to(t.dtype) * sc


def _mx_scales(t):
    """e8m0 block scales of the fp8 path's block-scaled activations (dinov2_od_amd/csrc/dod_common.h mx_ebyte): one byte per 32 elements of
    the last dimension, 2^(byte - 127) = the smallest power of two >= amax_block / 448 (fp32 arithmetic: amax * fp32(1 / 448)); a zero
    block gets byte 1.  Returns the biased bytes, shape [..., K / 32] (int64)."""
    K = t.shape[-1]
    amax = t.float().reshape(*t.shape[:-1], K // 32, 32).abs().amax(-1)
    tt = amax * (torch.tensor(1.0, dtype=torch.float32) / 448.0)
    m, ex = torch.frexp(tt)                       # tt = m * 2^ex, m in [0.5, 1)
    e = torch.where(m > 0.5, ex, ex - 1)          # ceil(log2(tt))
    return torch.where(amax > 0, (e.long() + 127).clamp(1, 253), torch.ones_like(e, dtype=torch.long))


def _q8_mx(t):
    """block-scaled e4m3 fake quantisation of the last dimension (K % 32 == 0); returns the dequantised values q * 2^e (exact)"""
    K = t.shape[-1]
    eb = _mx_scales(t)
    sc = torch.pow(torch.tensor(2.0, dtype=torch.float64), (eb - 127).double()).to(t.dtype)
    x = t.reshape(*t.shape[:-1], K // 32, 32)
    q = (x * (1.0 / sc)[..., None]).to(torch.float8_e4m3fn).to(t.dtype) * sc[..., None]
    return q.reshape(t.shape)


def _e4m3(t):
    return t.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(t.dtype)


def _h2_product(x, w):
    """x [.., K] . w [N, K]^T with the H2 operand scheme (packers: split_h2_kernel / h2_quad; product: gemm_h2_256x256_kernel)"""
    x32, w32 = x.float().clamp(-65504.0, 65504.0), w.float().clamp(-65504.0, 65504.0)
    hx, hw = x32.half().float(), w32.half().float()
    lx, lw = x32 - hx, w32 - hw
    amax = hw.abs().amax(dim=-1, keepdim=True)
    e = torch.where(amax > 0, torch.floor(torch.log2(448.0 / amax.double())).float(), torch.zeros_like(amax))
    sw = torch.exp2(e)
    main = hx.double() @ hw.double().t()
    cross = _e4m3(hx).double() @ (_e4m3(lw * sw * 2048.0) / (sw * 2048.0)).double().t() \
        + (_e4m3(lx * 2048.0) / 2048.0).double() @ (_e4m3(hw * sw) / sw).double().t()
    return (main + cross).to(x.dtype)


def _linear(x, w, b, emu, fp8=False):
    if emu == "fp16x2":
        y = _h2_product(x, w)
        return y if b is None else y + b
    if emu and fp8:
        # round 4: BOTH operands block-scaled (one power-of-two scale per 32 elements along K: _q8_mx) wherever the HIP path can (K % 256 == 0);
        # other widths keep per-token activation scales / per-output-feature weight scales.  fp8 == "w": the activation arrives already quantised
        q = _q8_mx if x.shape[-1] % 256 == 0 else _q8
        x, w = (x if fp8 == "w" else q(x)), q(w)
    elif emu:
        x, w = _bf(x), _bf(w)
    y = x @ w.t()
    return y if b is None else y + b


def _maybe_lora_linear(sd, prefix, x, alpha, emu, fp8=False):
    """nn.Linear, or LoraLinear (dino_detector/utils.py:68-70):
    linear(x) + alpha * lora_B(lora_A(x)).  With `emu` the HIP path's merged
    weight W' = W + alpha*B@A (fp32, then bf16) is what is emulated."""
    if sd.has(prefix + ".linear.weight"):
        w, b = sd(prefix + ".linear.weight"), sd(prefix + ".linear.bias")
        A, Bm = sd(prefix + ".lora_A.weight"), sd(prefix + ".lora_B.weight")
        if emu:
            return _linear(x, w + alpha * (Bm @ A), b, emu, fp8)
        return x @ w.t() + b + alpha * ((x @ A.t()) @ Bm.t())
    return _linear(x, sd(prefix + ".weight"), sd(prefix + ".bias"), emu, fp8)


def _eff_weight(sd, prefix, alpha):
    """(W, b) of an nn.Linear, or the merged W + alpha * B A of a LoraLinear (what the HIP path packs)"""
    if sd.has(prefix + ".linear.weight"):
        return (sd(prefix + ".linear.weight") + alpha * (sd(prefix + ".lora_B.weight") @ sd(prefix + ".lora_A.weight")),
                sd(prefix + ".linear.bias"))
    return sd(prefix + ".weight"), sd(prefix + ".bias")


def _ln_linear_folded(h, gamma, beta, eps, w, b, emu):
    """linear(LayerNorm(h)) as the HIP fast modes evaluate it since round 4 (csrc/dod_common.h GemmEpi::ln_*): the GEMM reads the residual
    row h itself in its operand format against W' = W diag(gamma) and normalises in its epilogue,
        rstd (h W'^T - mean c) + (b + W beta),   c[n] = sum_k W'[n][k] (of the bf16-rounded elements in the single-pass bf16 mode),
    mean / rstd from the fp32 row.  Same value as the unfolded form in exact arithmetic; the ROUNDING POINTS are what is emulated."""
    mu = h.mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(((h - mu) ** 2).mean(-1, keepdim=True) + eps)
    wp = w * gamma
    bp = b + w @ beta
    if emu == "fp16x2":
        acc, c = _h2_product(h, wp), wp.sum(-1)
    else:
        wr = _bf(wp)
        acc, c = _bf(h) @ wr.t(), wr.sum(-1)
    return (acc - mu * c) * rstd + bp


def _layernorm(x, w, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def interpolate_pos_encoding(pos, bb: BackboneConfig, H, W):
    """modeling_dinov2.py:57-95.  pos: [1, G*G+1, D]."""
    npatch = (H // bb.patch) * (W // bb.patch)
    npos = pos.shape[1] - 1
    if npatch == npos and H == W:
        return pos
    cls_pos, patch_pos = pos[:, :1], pos[:, 1:]
    D = pos.shape[-1]
    g = int(npos ** 0.5)
    p = patch_pos.reshape(1, g, g, D).permute(0, 3, 1, 2)
    p = F.interpolate(p.to(torch.float32), size=(H // bb.patch, W // bb.patch),
                      mode="bicubic", align_corners=False).to(pos.dtype)
    p = p.permute(0, 2, 3, 1).reshape(1, -1, D)
    return torch.cat((cls_pos, p), dim=1)


def bicubic_resize_ref(src, out_h, out_w):
    """Independent restatement of torch's upsample_bicubic2d (A=-0.75,
    align_corners=False, border-clamped taps) in float64 -- used to pin the HIP
    pos-embed resize kernel.  src [C, h, w] -> [C, out_h, out_w]."""
    src = src.to(torch.float64)
    C, h, w = src.shape
    A = -0.75

    def cc1(x):
        return ((A + 2) * x - (A + 3)) * x * x + 1

    def cc2(x):
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A

    def coeffs(t):
        return [cc2(t + 1.0), cc1(t), cc1(1.0 - t), cc2(2.0 - t)]

    out = torch.zeros(C, out_h, out_w, dtype=torch.float64)
    sh, sw = h / out_h, w / out_w
    for oy in range(out_h):
        ry = sh * (oy + 0.5) - 0.5
        iy = math.floor(ry)
        ty = ry - iy
        cy = coeffs(ty)
        for ox in range(out_w):
            rx = sw * (ox + 0.5) - 0.5
            ix = math.floor(rx)
            tx = rx - ix
            cx = coeffs(tx)
            acc = torch.zeros(C, dtype=torch.float64)
            for a in range(4):
                yy = min(max(iy - 1 + a, 0), h - 1)
                for b in range(4):
                    xx = min(max(ix - 1 + b, 0), w - 1)
                    acc += src[:, yy, xx] * (cy[a] * cx[b])
            out[:, oy, ox] = acc
    return out


def backbone_forward(sd_raw, bb: BackboneConfig, pixel_values, dtype=torch.float32,
                     emulate_bf16=False, prefix="backbone.", taps=None, fold_ln=True):
    """-> features [B, N, out_dim] (CLS token included: dinov2_backbone.py:60-61)."""
    sd = _SD(sd_raw, dtype)
    emu_lin = emulate_bf16                                   # the four linears of every block
    emu = False if emulate_bf16 == "fp16x2" else emulate_bf16   # fp16x2: patch embed, attention, projection are split products (exact here)
    x = _t(pixel_values, dtype)
    B, Cc, H, W = x.shape
    if Cc != 3:
        raise ValueError("Make sure that the channel dimension of the pixel values match with the "
                         f"one set in the configuration. Expected 3 but got {Cc}.")
    D, p = bb.hidden, bb.patch
    e = prefix + "dino.embeddings."
    # K1: Conv2d(3->D, k=14, s=14) == im2col GEMM   modeling_dinov2.py:139,148
    gh, gw = H // p, W // p
    cols = x[:, :, : gh * p, : gw * p].reshape(B, 3, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5)
    cols = cols.reshape(B, gh * gw, 3 * p * p)
    wconv = sd(e + "patch_embeddings.projection.weight").reshape(D, 3 * p * p)
    emb = _linear(cols, wconv, sd(e + "patch_embeddings.projection.bias"), emu)
    # K2: cls concat + pos-embed   modeling_dinov2.py:107-112
    cls = sd(e + "cls_token").expand(B, -1, -1)
    pos = interpolate_pos_encoding(sd(e + "position_embeddings"), bb, H, W)
    h = torch.cat((cls, emb), dim=1) + pos
    if taps is not None:
        taps["embeddings"] = h.clone()
    N = h.shape[1]
    nh, dh = bb.heads, bb.head_dim
    for i in range(bb.layers):
        lp = f"{prefix}dino.encoder.layer.{i}."
        a = bb.lora_alpha
        # K3-K6   modeling_dinov2.py:361-370
        f8 = emu == "fp8"   # fp8 mode: QKV / out-proj / MLP-in (/ SwiGLU MLP-out) linears on e4m3 operands, the rest as the bf16 mode
        fold = emu_lin in (True, "fp16x2") and fold_ln      # bf16 / fp16x2 emulation: norm1 / norm2 folded into the GEMM that follows
        if fold:
            g1, b1 = sd(lp + "norm1.weight"), sd(lp + "norm1.bias")
            q, k, v = (_ln_linear_folded(h, g1, b1, bb.ln_eps, *_eff_weight(sd, lp + "attention.attention." + nm, a), emu_lin)
                       for nm in ("query", "key", "value"))
        else:
            y = _layernorm(h, sd(lp + "norm1.weight"), sd(lp + "norm1.bias"), bb.ln_eps)
            q = _maybe_lora_linear(sd, lp + "attention.attention.query", y, a, emu_lin, f8)
            k = _maybe_lora_linear(sd, lp + "attention.attention.key", y, a, emu_lin, f8)
            v = _maybe_lora_linear(sd, lp + "attention.attention.value", y, a, emu_lin, f8)
        if emu:
            q, k, v = _bf(q), _bf(k), _bf(v)
        q = q.view(B, N, nh, dh).transpose(1, 2)
        k = k.view(B, N, nh, dh).transpose(1, 2)
        v = v.view(B, N, nh, dh).transpose(1, 2)
        s = (q @ k.transpose(2, 3)) * (dh ** -0.5)
        if emu:
            # flash-style: unnormalised P rounded to bf16, row sum over the rounded
            # values' fp32 originals (HIP kernel sums fp32 P before rounding)
            m = s.max(-1, keepdim=True).values
            pexp = torch.exp(s - m)
            ctx = (_bf(pexp) @ v) / pexp.sum(-1, keepdim=True)
        else:
            ctx = torch.softmax(s, dim=-1) @ v
        ctx = ctx.transpose(1, 2).reshape(B, N, D)
        if emu == "fp8" and D % 256 != 0:
            ctx = _bf(ctx)            # widths without block scales: the fp8 path stores the context in bf16, then quantises its rows
                                      # (block-scaled: the attention epilogue quantises its fp32 tile directly)
        o = _maybe_lora_linear(sd, lp + "attention.output.dense", ctx, a, emu_lin, emu == "fp8")
        h = o * sd(lp + "layer_scale1.lambda1") + h
        if taps is not None and i == 0:
            taps["block0_attn"] = h.clone()
        # K7 / K7g   modeling_dinov2.py:373-380
        y = None if fold else _layernorm(h, sd(lp + "norm2.weight"), sd(lp + "norm2.bias"), bb.ln_eps)
        mlp_in = (lambda nm: _ln_linear_folded(h, sd(lp + "norm2.weight"), sd(lp + "norm2.bias"), bb.ln_eps, *_eff_weight(sd, lp + nm, a), emu_lin)) if fold \
            else (lambda nm: _maybe_lora_linear(sd, lp + nm, y, a, emu_lin, f8))
        if bb.swiglu:
            z = mlp_in("mlp.weights_in")
            x1, x2 = z.chunk(2, dim=-1)
            z = F.silu(x1) * x2
            if f8 and z.shape[-1] % 256 == 0:
                # the fp8 path gates in the MLP-in GEMM's epilogue (fp32) and quantises there: e4m3 with one power-of-two scale per 32 columns
                z = _maybe_lora_linear(sd, lp + "mlp.weights_out", _q8_mx(z), a, emu_lin, "w")
            else:
                if f8:
                    z = _bf(z)          # other widths: bf16 hidden rows, then one scale per row
                z = _maybe_lora_linear(sd, lp + "mlp.weights_out", z, a, emu_lin, f8)
        else:
            z = mlp_in("mlp.fc1")
            z = 0.5 * z * (1.0 + torch.erf(z / math.sqrt(2.0)))   # exact-erf GELU
            z = _maybe_lora_linear(sd, lp + "mlp.fc2", z, a, emu_lin)
        h = z * sd(lp + "layer_scale2.lambda1") + h
        if taps is not None:
            taps[f"block{i}"] = h.clone()
    h = _layernorm(h, sd(prefix + "dino.layernorm.weight"), sd(prefix + "dino.layernorm.bias"), bb.ln_eps)
    if taps is not None:
        taps["final_ln"] = h.clone()
    if sd.has(prefix + "projection.weight"):   # K9  dinov2_backbone.py:64-65
        h = _linear(h, sd(prefix + "projection.weight"), sd(prefix + "projection.bias"), emu)
        if taps is not None:
            taps["projection"] = h.clone()
    return h


def _mha(sd, prefix, xq, xkv, nheads):
    """torch.nn.MultiheadAttention forward (batch-major restatement; no masks, eval).
    xq [B,Lq,E], xkv [B,Lk,E]."""
    W, bvec = sd(prefix + ".in_proj_weight"), sd(prefix + ".in_proj_bias")
    E = xq.shape[-1]
    dh = E // nheads
    q = xq @ W[:E].t() + bvec[:E]
    k = xkv @ W[E:2 * E].t() + bvec[E:2 * E]
    v = xkv @ W[2 * E:].t() + bvec[2 * E:]
    B, Lq, Lk = xq.shape[0], xq.shape[1], xkv.shape[1]
    q = q.view(B, Lq, nheads, dh).transpose(1, 2)
    k = k.view(B, Lk, nheads, dh).transpose(1, 2)
    v = v.view(B, Lk, nheads, dh).transpose(1, 2)
    s = (q @ k.transpose(2, 3)) * (1.0 / math.sqrt(dh))
    o = torch.softmax(s, dim=-1) @ v
    o = o.transpose(1, 2).reshape(B, Lq, E)
    return o @ sd(prefix + ".out_proj.weight").t() + sd(prefix + ".out_proj.bias")


def deformable_sample(values, ref, offsets, weights, h, w):
    """K15, vectorised restatement of deformable_attention.py:101-174.
    values [B,HW,Hd,dh]; ref [B,Q,2]; offsets [B,Q,Hd,P,2]; weights [B,Q,Hd,P] (softmaxed)
    -> [B,Q,Hd,dh]"""
    B, HW, Hd, dh = values.shape
    loc = torch.clamp(ref[:, :, None, None, :] + offsets, 0, 1)
    lx = loc[..., 0] * (w - 1)
    ly = loc[..., 1] * (h - 1)
    x0 = torch.floor(lx).long()
    y0 = torch.floor(ly).long()
    x1, y1 = x0 + 1, y0 + 1
    x0 = x0.clamp(0, w - 1)
    x1 = x1.clamp(0, w - 1)
    y0 = y0.clamp(0, h - 1)
    y1 = y1.clamp(0, h - 1)
    wx1 = lx - x0.to(lx.dtype)
    wx0 = 1.0 - wx1
    wy1 = ly - y0.to(ly.dtype)
    wy0 = 1.0 - wy1
    Q, P = ref.shape[1], offsets.shape[3]
    vh = values.permute(0, 2, 1, 3)                                  # [B,Hd,HW,dh]

    def gather(yy, xx):
        idx = (yy * w + xx).permute(0, 2, 1, 3).reshape(B, Hd, Q * P)   # [B,Hd,Q*P]
        g = torch.gather(vh, 2, idx[..., None].expand(-1, -1, -1, dh))
        return g.view(B, Hd, Q, P, dh).permute(0, 2, 1, 3, 4)           # [B,Q,Hd,P,dh]

    res = (gather(y0, x0) * (wx0 * wy0)[..., None] + gather(y1, x0) * (wx0 * wy1)[..., None]
           + gather(y0, x1) * (wx1 * wy0)[..., None] + gather(y1, x1) * (wx1 * wy1)[..., None])
    return (res * weights[..., None]).sum(dim=3)


def decoder_forward(sd_raw, dc: DecoderConfig, memory, dtype=torch.float32, emulate_bf16=False,
                    prefix="decoder.", taps=None):
    """DETRDecoder.forward (detr_decoder.py:47-83) -> (pred_logits [B,Q,C], pred_boxes [B,Q,4])"""
    sd = _SD(sd_raw, dtype)
    mem = _t(memory, dtype)
    B, N, Dd = mem.shape
    Q, Hd, P = dc.num_queries, dc.nheads, dc.n_points
    dh = Dd // Hd
    tgt = sd(prefix + "query_embed.weight")[None].repeat(B, 1, 1)        # K10
    if dc.use_deformable:
        for j in range(dc.num_layers):
            lp = f"{prefix}decoder.layers.{j}."
            # K11  deformable_attention.py:232-235
            t2 = _mha(sd, lp + "self_attn", tgt, tgt, Hd)
            tgt = _layernorm(tgt + t2, sd(lp + "norm1.weight"), sd(lp + "norm1.bias"), dc.ln_eps)
            # K12  :238
            ref = torch.sigmoid(tgt @ sd(lp + "reference_points_proj.weight").t()
                                + sd(lp + "reference_points_proj.bias"))
            h, w = spatial_factor(N)                                         # K16 :241-256
            # K13  :86-94
            off = (tgt @ sd(lp + "cross_attn.sampling_offsets.weight").t()
                   + sd(lp + "cross_attn.sampling_offsets.bias")).view(B, Q, Hd, P, 2)
            aw = (tgt @ sd(lp + "cross_attn.attention_weights.weight").t()
                  + sd(lp + "cross_attn.attention_weights.bias")).view(B, Q, Hd, P).softmax(-1)
            # K14  :97
            val = _linear(mem, sd(lp + "cross_attn.value_proj.weight"),
                          sd(lp + "cross_attn.value_proj.bias"), emulate_bf16)
            if taps is not None and j == 0:
                taps["value_proj"] = val.clone()
            samp = deformable_sample(val.view(B, N, Hd, dh), ref, off, aw, h, w).reshape(B, Q, Dd)
            if taps is not None:
                taps[f"sampled{j}"] = samp.clone()
            # K17  :181, 260-261
            t2 = samp @ sd(lp + "cross_attn.output_proj.weight").t() + sd(lp + "cross_attn.output_proj.bias")
            tgt = _layernorm(tgt + t2, sd(lp + "norm2.weight"), sd(lp + "norm2.bias"), dc.ln_eps)
            # K18  :264-266
            t2 = torch.relu(tgt @ sd(lp + "linear1.weight").t() + sd(lp + "linear1.bias"))
            t2 = t2 @ sd(lp + "linear2.weight").t() + sd(lp + "linear2.bias")
            tgt = _layernorm(tgt + t2, sd(lp + "norm3.weight"), sd(lp + "norm3.bias"), dc.ln_eps)
            if taps is not None:
                taps[f"dec{j}"] = tgt.clone()
    else:
        # K20: nn.TransformerDecoderLayer (norm_first=False, relu), untied layers
        m = _bf(mem) if emulate_bf16 else mem
        for j in range(dc.num_layers):
            lp = f"{prefix}decoder.layers.{j}."
            t2 = _mha(sd, lp + "self_attn", tgt, tgt, Hd)
            tgt = _layernorm(tgt + t2, sd(lp + "norm1.weight"), sd(lp + "norm1.bias"), dc.ln_eps)
            t2 = _mha(sd, lp + "multihead_attn", tgt, m, Hd)
            tgt = _layernorm(tgt + t2, sd(lp + "norm2.weight"), sd(lp + "norm2.bias"), dc.ln_eps)
            t2 = torch.relu(tgt @ sd(lp + "linear1.weight").t() + sd(lp + "linear1.bias"))
            t2 = t2 @ sd(lp + "linear2.weight").t() + sd(lp + "linear2.bias")
            tgt = _layernorm(tgt + t2, sd(lp + "norm3.weight"), sd(lp + "norm3.bias"), dc.ln_eps)
            if taps is not None:
                taps[f"dec{j}"] = tgt.clone()
    # K19  detr_decoder.py:80-81
    logits = tgt @ sd(prefix + "class_embed.weight").t() + sd(prefix + "class_embed.bias")
    hb = torch.relu(tgt @ sd(prefix + "bbox_embed.mlp.0.weight").t() + sd(prefix + "bbox_embed.mlp.0.bias"))
    boxes = torch.sigmoid(hb @ sd(prefix + "bbox_embed.mlp.2.weight").t() + sd(prefix + "bbox_embed.mlp.2.bias"))
    return logits, boxes


def detector_forward(sd, bb: BackboneConfig, dc: DecoderConfig, pixel_values,
                     dtype=torch.float32, emulate_bf16=False, taps=None):
    """DINOv2ObjectDetector.forward (detector.py:58-69) -> dict like the reference's."""
    with torch.no_grad():
        feats = backbone_forward(sd, bb, pixel_values, dtype, emulate_bf16, "backbone.", taps)
        dec_emu = False if emulate_bf16 == "fp16x2" else emulate_bf16     # fp16x2: the decoder runs as in the fp32 mode
        if dec_emu:
            feats = _bf(feats)      # HIP fast path hands bf16 memory to the decoder
        logits, boxes = decoder_forward(sd, dc, feats, dtype, dec_emu, "decoder.", taps)
    return {"pred_logits": logits, "pred_boxes": boxes, "features": feats}
