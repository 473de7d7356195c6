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


def supported(m, src):
    """deformable branch with tied layers (the reference's only form), fp32 CUDA tensors, shapes the kernels take"""
    if not (m.use_deformable and src.is_cuda and src.dtype == torch.float32):
        return False
    layers = list(m.decoder.layers)
    if any(l is not layers[0] for l in layers):
        return False
    dc = m._dc_cfg
    dh = dc.hidden_dim // dc.nheads
    return (dc.hidden_dim % dc.nheads == 0 and dh <= 128 and dh % 4 == 0 and dc.hidden_dim <= 1024 and dc.hidden_dim % 8 == 0
            and dc.dim_feedforward % 4 == 0 and dc.num_queries <= 1024 and 1 <= dc.n_points <= 8
            and all(p.is_cuda and p.dtype == torch.float32 for p in _param_list(m)))


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
    p = m._dropout_p if m.training else 0.0
    return _DecoderTrain.apply(src, cfg, p, seed, *_param_list(m))
