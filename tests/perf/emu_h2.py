#!/usr/bin/env python3
"""Numerics study (CPU, oracle = test infrastructure): end-to-end error of candidate operand formats for the backbone linears,
ViT-B/14 at 224x224, synthetic weights, vs the fp64 oracle.
  bf16   : a.b ~ bf16(a) bf16(b)
  x3     : bf16 split, three products (the bf16x3 mode)
  h2     : fp16(a) fp16(b) + q8(ah) q8(bl) + q8(al) q8(bh): fp16 main product, cross terms on e4m3 (power-of-two row scales)
"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import dinodet_oracle as orc
from dinov2_od_amd import synth
from tests.cases import vitb, rel_err, rel_l2

MODE = {"m": None}
_orig = orc._linear

def p2scale(t):      # power-of-two row scale: amax * 2^e in (224, 448]
    amax = t.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
    e = torch.floor(torch.log2(448.0 / amax))
    return torch.exp2(e)

def q8p(t, sc=None):
    sc = p2scale(t) if sc is None else sc
    return (t * sc).clamp(-448, 448).to(torch.float8_e4m3fn).to(t.dtype) / sc

def lin(x, w, b, emu, fp8=False):
    m = MODE["m"]
    if m is None:
        return _orig(x, w, b, False)
    x32, w32 = x.float(), w.float()
    if m == "bf16":
        y = orc._bf(x32).double() @ orc._bf(w32).double().t()
    elif m == "x3":
        xh, wh = orc._bf(x32), orc._bf(w32); xl, wl = orc._bf(x32 - xh), orc._bf(w32 - wh)
        y = xh.double() @ wh.double().t() + xh.double() @ wl.double().t() + xl.double() @ wh.double().t()
    elif m.startswith("h2"):
        xh, wh = x32.half().float(), w32.half().float(); xl, wl = x32 - xh, w32 - wh
        if m == "h2":            # per-row power-of-two scales for hi and lo
            x8h, x8l, w8h, w8l = q8p(xh), q8p(xl), q8p(wh), q8p(wl)
        elif m == "h2s":         # static activation scales (2^4 hi, 2^15 lo), per-row weight scales; lo scale = hi scale * 2^11
            sa = torch.tensor(16.0); x8h, x8l = q8p(xh, sa), q8p(xl, sa * 2048)
            sw = p2scale(wh); w8h, w8l = q8p(wh, sw), q8p(wl, sw * 2048)
        elif m == "h2t":         # per-row hi scale, lo scale tied to it (hi * 2^11): what a row-owning producer can do in one pass
            sa = p2scale(xh); x8h, x8l = q8p(xh, sa), q8p(xl, sa * 2048)
            sw = p2scale(wh); w8h, w8l = q8p(wh, sw), q8p(wl, sw * 2048)
        elif m == "h2f16only":
            x8h = x8l = w8h = w8l = None
        y = xh.double() @ wh.double().t()
        if x8h is not None:
            y = y + x8h.double() @ w8l.double().t() + x8l.double() @ w8h.double().t()
    y = y.to(x.dtype)
    return y if b is None else y + b

orc._linear = lin
# attention variant: S exact (split products), P and V rounded to fp16 for the P V product, row sum over the rounded P
ATT = {"m": None}
_orig_bf = orc._bf
def _bf_hook(t):
    if ATT["m"] == "pv16":
        return t.half().to(t.dtype)
    if ATT["m"] == "pvbf":
        return _orig_bf(t)
    return t
_orig_mll = orc._maybe_lora_linear
orc._maybe_lora_linear = lambda sd, prefix, x, alpha, emu, fp8=False: _orig_mll(sd, prefix, x, alpha, MODE["m"] is not None, False)
bb, dc = vitb(100)
H = int(os.environ.get("EMU_HW", "224"))
sd = synth.detector_state_dict(bb, dc, seed=1)
x = synth.make_pixels(2, H, H, seed=0)
MODE["m"] = None
ref_f = orc.backbone_forward(sd, bb, x, torch.float64)
ref = orc.detector_forward(sd, bb, dc, x, torch.float64)
f32 = orc.detector_forward(sd, bb, dc, x, torch.float32)
print(f"fp32 oracle vs fp64: logits {rel_err(f32['pred_logits'], ref['pred_logits']):.2e} boxes {rel_err(f32['pred_boxes'], ref['pred_boxes']):.2e}")
import inspect, re
src = inspect.getsource(orc.backbone_forward)
# variant of backbone_forward whose emu branch of the attention: q, k exact, v and P through _bf_hook, sum over the rounded P
src = src.replace("q, k, v = _bf(q), _bf(k), _bf(v)", "v = _bf_hook(v)")
src = src.replace("ctx = (_bf(pexp) @ v) / pexp.sum(-1, keepdim=True)", "pr = _bf_hook(pexp); ctx = (pr @ v) / pr.sum(-1, keepdim=True)")
ns = dict(orc.__dict__); ns["_bf_hook"] = _bf_hook
exec(src.replace("def backbone_forward", "def backbone_forward_att"), ns)
for m in os.environ.get("EMU_MODES", "x3,x3+pv16,x3+pvbf,h2t,h2t+pv16").split(","):
    m, _, am = m.partition("+")
    MODE["m"] = m; ATT["m"] = am or None
    ns["_linear"] = lin; ns["_maybe_lora_linear"] = orc._maybe_lora_linear
    feats = ns["backbone_forward_att"](sd, bb, x, torch.float64, emulate_bf16=bool(am))
    m = m + ("+" + am if am else "")
    MODE["m"] = None
    logits, boxes = orc.decoder_forward(sd, dc, feats, torch.float64, False, "decoder.")
    print(f"{m:10s}: features rel-L2 {rel_l2(feats, ref_f):.2e} max-rel {rel_err(feats, ref_f):.2e} | logits max-rel {rel_err(logits, ref['pred_logits']):.2e} boxes {rel_err(boxes, ref['pred_boxes']):.2e}", flush=True)
