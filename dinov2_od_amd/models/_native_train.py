"""train()-mode decoder + heads on the native kernels: forward with a tape, hand-written HIP backward (SURVEY.md section 8 row f1,
first slice; dinov2_od_amd/csrc/dec_train.hip).

`decoder_train(module, src)` is what DETRDecoder.forward evaluates in train() mode for the deformable branch on the GPU: a
torch.autograd.Function whose forward is `dod_decoder_train_forward` (dropout at the reference's five sites --
deformable_attention.py:195-209, 235, 261, 265-266 -- from a counter-based hash) and whose backward is
`dod_decoder_train_backward`: gradients of every decoder / head parameter (the layers are weight-tied, deformable_attention.py:284:
one accumulator per tensor) and of `src`, through which autograd continues into the projection and the LoRA-adapted blocks (those
are still the composite of models/_autograd.py).  torch supplies tensors, the stream and the autograd graph edge; no arithmetic.
"""
import ctypes as C

import torch

from .. import _native as nat
from ..engine import make_config

_seed_counter = [0]


def _param_list(m):
    """the 31 tensors of struct dod_dec_train_params, in its order"""
    L = m.decoder.layers[0]
    ca = L.cross_attn
    return [m.query_embed.weight, m.class_embed.weight, m.class_embed.bias,
            m.bbox_embed.mlp[0].weight, m.bbox_embed.mlp[0].bias, m.bbox_embed.mlp[2].weight, m.bbox_embed.mlp[2].bias,
            L.self_attn.in_proj_weight, L.self_attn.in_proj_bias, L.self_attn.out_proj.weight, L.self_attn.out_proj.bias,
            L.norm1.weight, L.norm1.bias, L.norm2.weight, L.norm2.bias, L.norm3.weight, L.norm3.bias,
            L.linear1.weight, L.linear1.bias, L.linear2.weight, L.linear2.bias,
            L.reference_points_proj.weight, L.reference_points_proj.bias,
            ca.sampling_offsets.weight, ca.sampling_offsets.bias, ca.attention_weights.weight, ca.attention_weights.bias,
            ca.value_proj.weight, ca.value_proj.bias, ca.output_proj.weight, ca.output_proj.bias]


def dropout_rate(m):
    """the ONE rate the native step applies at the reference's five sites (deformable_attention.py:195-209, 235, 261, 265-266), read
    from the layer's own modules -- each site counts as 0 when its module is in eval() -- or None when the sites disagree (a caller
    changed one `.p` or put a single sub-module in eval(): the composite, which calls the modules themselves, handles that)"""
    L = m.decoder.layers[0]
    rates = {float(d.p) if d.training else 0.0 for d in (L.dropout1, L.dropout2, L.dropout3, L.dropout4)}
    rates.add(float(L.self_attn.dropout) if L.self_attn.training else 0.0)
    return rates.pop() if len(rates) == 1 else None


def supported(m, src):
    """deformable branch with tied layers (the reference's only form), fp32 CUDA tensors, shapes the kernels take"""
    if not (m.use_deformable and src.is_cuda and src.dtype == torch.float32):
        return False
    layers = list(m.decoder.layers)
    if any(l is not layers[0] for l in layers):
        return False
    if not all(p.is_cuda and p.dtype == torch.float32 for p in _param_list(m)) or src.dim() != 3 or dropout_rate(m) is None:
        return False
    # the shape limits are the native side's own (make_dims in dec_train.hip): 0 bytes = not taken -> the composite runs instead
    cfg = make_config(m._bb_cfg, m._dc_cfg, "fp32")
    return nat.lib().dod_decoder_train_tape_bytes(C.byref(cfg), int(src.shape[0]), int(src.shape[1])) > 0


def _struct(tensors):
    s = nat.DodDecTrainParams()
    for f, t in zip(nat.DEC_TRAIN_FIELDS, tensors):
        setattr(s, f, t.data_ptr())
    return s


def _check(rc):
    if rc != 0:
        msg = nat.lib().dod_decoder_train_last_error()
        raise (ValueError if rc == 1 else RuntimeError)(msg.decode() if msg else f"dinodet error {rc}")


class _DecoderTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, cfg, p, seed, *params):
        L = nat.lib()
        B, N, _ = src.shape
        src = src.contiguous()
        params = [t.detach().contiguous() for t in params]
        with torch.cuda.device(src.device):
            tape = torch.empty(L.dod_decoder_train_tape_bytes(C.byref(cfg), B, N), dtype=torch.uint8, device=src.device)
            ws = torch.empty(L.dod_decoder_train_workspace_bytes(C.byref(cfg), B, N), dtype=torch.uint8, device=src.device)
            if tape.numel() == 0 or ws.numel() == 0:
                raise ValueError("decoder configuration not supported by the native training kernels")
            det = torch.empty(B, cfg.num_queries, cfg.num_classes + 4, dtype=torch.float32, device=src.device)
            ps = _struct(params)
            _check(L.dod_decoder_train_forward(C.byref(cfg), C.byref(ps), nat.ptr(src), B, N, float(p), int(seed), nat.ptr(det), nat.ptr(tape),
                                               tape.numel(), nat.ptr(ws), ws.numel(), nat.stream_ptr()))
        ctx.save_for_backward(src, tape, *params)
        ctx.cfg, ctx.p, ctx.seed, ctx.ws = cfg, float(p), int(seed), ws
        return det

    @staticmethod
    def backward(ctx, d_det):
        L = nat.lib()
        src, tape, *params = ctx.saved_tensors
        cfg = ctx.cfg
        B, N, _ = src.shape
        d_det = d_det.contiguous().float()
        with torch.cuda.device(src.device):
            grads = [torch.zeros_like(t) for t in params]
            d_src = torch.empty_like(src) if ctx.needs_input_grad[0] else None
            ps, gs = _struct(params), _struct(grads)
            _check(L.dod_decoder_train_backward(C.byref(cfg), C.byref(ps), nat.ptr(src), B, N, ctx.p, ctx.seed, nat.ptr(d_det), nat.ptr(tape),
                                                tape.numel(), C.byref(gs), nat.ptr(d_src), nat.ptr(ctx.ws), ctx.ws.numel(), nat.stream_ptr()))
        return (d_src, None, None, None, *grads)


def decoder_train(m, src, seed=None):
    """DETRDecoder.forward in train() mode -> packed detections [B, Q, C+4] with the autograd edge to the native backward"""
    cfg = make_config(m._bb_cfg, m._dc_cfg, "fp32")
    if seed is None:
        # one draw of torch's generator per call (so torch.manual_seed governs the masks), mixed with a call counter
        _seed_counter[0] += 1
        seed = (int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) << 20) ^ _seed_counter[0]
    return _DecoderTrain.apply(src, cfg, dropout_rate(m), seed, *_param_list(m))


# ------------------------------------------------------------------------------------------------------------------------------
# dense decoder: the nn.TransformerDecoder branch (use_deformable=False; dod_dense_decoder_train_forward / _backward)
def _dense_layer_tensors(L):
    """the 18 tensors of struct dod_dense_layer_params, in its order (torch.nn.TransformerDecoderLayer)"""
    return [L.self_attn.in_proj_weight, L.self_attn.in_proj_bias, L.self_attn.out_proj.weight, L.self_attn.out_proj.bias,
            L.multihead_attn.in_proj_weight, L.multihead_attn.in_proj_bias, L.multihead_attn.out_proj.weight, L.multihead_attn.out_proj.bias,
            L.linear1.weight, L.linear1.bias, L.linear2.weight, L.linear2.bias,
            L.norm1.weight, L.norm1.bias, L.norm2.weight, L.norm2.bias, L.norm3.weight, L.norm3.bias]


def _dense_param_list(m):
    out = []
    for L in m.decoder.layers:
        out += _dense_layer_tensors(L)
    return out + [m.query_embed.weight, m.class_embed.weight, m.class_embed.bias,
                  m.bbox_embed.mlp[0].weight, m.bbox_embed.mlp[0].bias, m.bbox_embed.mlp[2].weight, m.bbox_embed.mlp[2].bias]


def dense_dropout_rate(m):
    """the one rate of torch's six dropout sites per layer (both attentions' probabilities, dropout1/2/3, the FFN's dropout), each
    0 when its module is in eval(); None when they disagree (then the composite runs)"""
    rates = set()
    for L in m.decoder.layers:
        for d in (L.dropout, L.dropout1, L.dropout2, L.dropout3):
            rates.add(float(d.p) if d.training else 0.0)
        for a in (L.self_attn, L.multihead_attn):
            rates.add(float(a.dropout) if a.training else 0.0)
    return rates.pop() if len(rates) == 1 else None


def dense_supported(m, src):
    """the reference's standard branch as torch builds it: post-norm ReLU layers, no final norm, fp32 CUDA tensors, shapes the native
    side takes (dod_dense_decoder_train_tape_bytes > 0)"""
    if m.use_deformable or not (src.is_cuda and src.dtype == torch.float32 and src.dim() == 3):
        return False
    if getattr(m.decoder, "norm", None) is not None:
        return False
    for L in m.decoder.layers:
        if getattr(L, "norm_first", False) or getattr(L.self_attn, "batch_first", False):
            return False
        act = getattr(L, "activation", None)
        if not (act is torch.nn.functional.relu or isinstance(act, torch.nn.ReLU)):
            return False
        if L.self_attn.in_proj_weight is None or L.multihead_attn.in_proj_weight is None:
            return False
    if not all(p.is_cuda and p.dtype == torch.float32 for p in _dense_param_list(m)) or dense_dropout_rate(m) is None:
        return False
    cfg = make_config(m._bb_cfg, m._dc_cfg, "fp32")
    return nat.lib().dod_dense_decoder_train_tape_bytes(C.byref(cfg), int(src.shape[0]), int(src.shape[1])) > 0


def _dense_struct(tensors, nlayers):
    layers = (nat.DodDenseLayerParams * nlayers)()
    it = iter(tensors)
    for lp in layers:
        for f in nat.DENSE_LAYER_FIELDS:
            setattr(lp, f, next(it).data_ptr())
    ps = nat.DodDenseDecTrainParams()
    ps.nlayers, ps.layers = nlayers, layers
    for f in ("query_embed", "class_w", "class_b", "bb0_w", "bb0_b", "bb2_w", "bb2_b"):
        setattr(ps, f, next(it).data_ptr())
    return ps, layers          # keep `layers` alive as long as `ps`


class _DenseDecoderTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, cfg, p, seed, *params):
        L = nat.lib()
        B, N, _ = src.shape
        src = src.contiguous()
        params = [t.detach().contiguous() for t in params]
        nl = cfg.dec_layers
        with torch.cuda.device(src.device):
            tape = torch.empty(L.dod_dense_decoder_train_tape_bytes(C.byref(cfg), B, N), dtype=torch.uint8, device=src.device)
            ws = torch.empty(L.dod_dense_decoder_train_workspace_bytes(C.byref(cfg), B, N), dtype=torch.uint8, device=src.device)
            if tape.numel() == 0 or ws.numel() == 0:
                raise ValueError("decoder configuration not supported by the native training kernels")
            det = torch.empty(B, cfg.num_queries, cfg.num_classes + 4, dtype=torch.float32, device=src.device)
            ps, keep = _dense_struct(params, nl)
            _check(L.dod_dense_decoder_train_forward(C.byref(cfg), C.byref(ps), nat.ptr(src), B, N, float(p), int(seed), nat.ptr(det), nat.ptr(tape),
                                                     tape.numel(), nat.ptr(ws), ws.numel(), nat.stream_ptr()))
        ctx.save_for_backward(src, tape, *params)
        ctx.cfg, ctx.p, ctx.seed, ctx.ws = cfg, float(p), int(seed), ws
        return det

    @staticmethod
    def backward(ctx, d_det):
        L = nat.lib()
        src, tape, *params = ctx.saved_tensors
        cfg = ctx.cfg
        B, N, _ = src.shape
        d_det = d_det.contiguous().float()
        with torch.cuda.device(src.device):
            grads = [torch.zeros_like(t) for t in params]
            d_src = torch.empty_like(src)
            ps, k1 = _dense_struct(params, cfg.dec_layers)
            gs, k2 = _dense_struct(grads, cfg.dec_layers)
            _check(L.dod_dense_decoder_train_backward(C.byref(cfg), C.byref(ps), nat.ptr(src), B, N, ctx.p, ctx.seed, nat.ptr(d_det), nat.ptr(tape),
                                                      tape.numel(), C.byref(gs), nat.ptr(d_src), nat.ptr(ctx.ws), ctx.ws.numel(), nat.stream_ptr()))
        return (d_src, None, None, None, *grads)


def dense_decoder_train(m, src, seed=None):
    """DETRDecoder.forward (use_deformable=False) in train() mode -> packed detections [B, Q, C+4] with the autograd edge to the native backward"""
    cfg = make_config(m._bb_cfg, m._dc_cfg, "fp32")
    if seed is None:
        _seed_counter[0] += 1
        seed = (int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) << 20) ^ _seed_counter[0]
    return _DenseDecoderTrain.apply(src, cfg, dense_dropout_rate(m), seed, *_dense_param_list(m))


# ------------------------------------------------------------------------------------------------------------------------------
# backbone tail: the LoRA-adapted blocks + final LayerNorm + projection (dod_backbone_tail_train_forward / _backward)
_LORA_SITES = ("q", "k", "v", "o", "fc1", "fc2")


def _tail_modules(layer):
    """the six LoRA-wrapped linears of a block in the order of _LORA_SITES; a SwiGLU block (ViT-g, modeling_dinov2.py:300-314) puts
    mlp.weights_in / mlp.weights_out in the fc1 / fc2 slots"""
    a = layer.attention
    mlp = (layer.mlp.fc1, layer.mlp.fc2) if hasattr(layer.mlp, "fc1") else (layer.mlp.weights_in, layer.mlp.weights_out)
    return [a.attention.query, a.attention.key, a.attention.value, a.output.dense, *mlp]


def tail_supported(m, layers, x):
    """every remaining block carries LoRA on its six linears over a frozen base, nothing else in it trains, fp32 on the GPU -- and the
    native side takes the shape (dod_backbone_tail_tape_bytes > 0 is the kernels' own check: head_dim, token count, rank, depth), so
    that an unsupported configuration falls back to the autograd composite instead of raising"""
    bb = m._bb_cfg
    if not (x.is_cuda and x.dtype == torch.float32 and layers):
        return False
    for L in layers:
        if not (hasattr(L.mlp, "fc1") or hasattr(L.mlp, "weights_in")):
            return False
        mods = _tail_modules(L)
        if not all(hasattr(t, "lora_A") for t in mods):
            return False
        frozen = [L.norm1.weight, L.norm1.bias, L.norm2.weight, L.norm2.bias, L.layer_scale1.lambda1, L.layer_scale2.lambda1]
        frozen += [q for t in mods for q in (t.linear.weight, t.linear.bias)]
        if any(q.requires_grad for q in frozen) or any(t.alpha != mods[0].alpha for t in mods):
            return False
    if any(q.requires_grad for q in m.dino.layernorm.parameters()):
        return False
    from ..config import DecoderConfig
    cfg = make_config(bb, getattr(m, "_dc_cfg", None) or DecoderConfig(), "fp32")
    return nat.lib().dod_backbone_tail_tape_bytes(C.byref(cfg), int(x.shape[0]), int(x.shape[1]), len(layers)) > 0


def _tail_structs(m, layers, train_tensors, writable):
    """struct dod_bb_tail_params over the module's own tensors; `train_tensors` (A, B per site and block, then the projection) is
    what fills the trainable slots -- the parameters themselves, or the gradient accumulators (then the frozen slots stay NULL)"""
    blocks = (nat.DodBbBlockParams * len(layers))()
    it = iter(train_tensors)
    for bp, L in zip(blocks, layers):
        if not writable:
            bp.ln1_w, bp.ln1_b, bp.ln2_w, bp.ln2_b = (L.norm1.weight.data_ptr(), L.norm1.bias.data_ptr(), L.norm2.weight.data_ptr(),
                                                      L.norm2.bias.data_ptr())
            bp.ls1, bp.ls2 = L.layer_scale1.lambda1.data_ptr(), L.layer_scale2.lambda1.data_ptr()
        for site, mod in zip(_LORA_SITES, _tail_modules(L)):
            ll = getattr(bp, site)
            if not writable:
                ll.w, ll.b = mod.linear.weight.data_ptr(), mod.linear.bias.data_ptr()
            ll.A, ll.Bm = next(it).data_ptr(), next(it).data_ptr()
    ps = nat.DodBbTailParams()
    ps.nblocks, ps.blocks = len(layers), blocks
    if not writable:
        ps.lnf_w, ps.lnf_b = m.dino.layernorm.weight.data_ptr(), m.dino.layernorm.bias.data_ptr()
    if m.projection is not None:
        ps.proj_w, ps.proj_b = next(it).data_ptr(), next(it).data_ptr()
    return ps, blocks          # keep `blocks` alive as long as `ps`


def _tail_trainables(m, layers):
    out = []
    for L in layers:
        for mod in _tail_modules(L):
            out += [mod.lora_A.weight, mod.lora_B.weight]
    if m.projection is not None:
        out += [m.projection.weight, m.projection.bias]
    return out


class _BackboneTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_in, m, layers, cfg, *train):
        L = nat.lib()
        B, N, _ = x_in.shape
        x_in = x_in.contiguous()
        train = [t.detach().contiguous() for t in train]
        nb = len(layers)
        with torch.cuda.device(x_in.device):
            tape = torch.empty(L.dod_backbone_tail_tape_bytes(C.byref(cfg), B, N, nb), dtype=torch.uint8, device=x_in.device)
            ws = torch.empty(L.dod_backbone_tail_workspace_bytes(C.byref(cfg), B, N, nb), dtype=torch.uint8, device=x_in.device)
            if tape.numel() == 0 or ws.numel() == 0:
                raise ValueError("backbone configuration not supported by the native training kernels")
            mem = torch.empty(B, N, m._bb_cfg.out_dim, dtype=torch.float32, device=x_in.device)
            ps, keep = _tail_structs(m, layers, train, writable=False)
            _check(L.dod_backbone_tail_train_forward(C.byref(cfg), C.byref(ps), nat.ptr(x_in), B, N, nat.ptr(mem), nat.ptr(tape), tape.numel(),
                                                     nat.ptr(ws), ws.numel(), nat.stream_ptr()))
        ctx.save_for_backward(tape, *train)
        ctx.m, ctx.layers, ctx.cfg, ctx.ws, ctx.shape = m, layers, cfg, ws, (B, N)
        return mem

    @staticmethod
    def backward(ctx, d_mem):
        L = nat.lib()
        tape, *train = ctx.saved_tensors
        B, N = ctx.shape
        d_mem = d_mem.contiguous().float()
        with torch.cuda.device(d_mem.device):
            grads = [torch.zeros_like(t) for t in train]
            ps, keep1 = _tail_structs(ctx.m, ctx.layers, train, writable=False)
            gs, keep2 = _tail_structs(ctx.m, ctx.layers, grads, writable=True)
            _check(L.dod_backbone_tail_train_backward(C.byref(ctx.cfg), C.byref(ps), B, N, nat.ptr(d_mem), nat.ptr(tape), tape.numel(), C.byref(gs),
                                                      nat.ptr(ctx.ws), ctx.ws.numel(), nat.stream_ptr()))
        return (None, None, None, None, *grads)


def backbone_tail(m, x_in, layers):
    """residual stream in front of the first LoRA-adapted block -> decoder memory [B, N, out_dim], with the autograd edge to the
    native backward (gradients of lora_A / lora_B and of the projection)"""
    from ..config import DecoderConfig
    dc = getattr(m, "_dc_cfg", None) or DecoderConfig()
    cfg = make_config(m._bb_cfg, dc, "fp32")
    return _BackboneTail.apply(x_in, m, list(layers), cfg, *_tail_trainables(m, layers))
