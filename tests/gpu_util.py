"""Helpers for the `-m gpu` parity tests (HIP path through the C ABI vs the CPU oracle)."""
import ctypes as C

import numpy as np
import torch

from dinov2_od_amd import _native as nat
from dinov2_od_amd import synth
from dinov2_od_amd.models import DINOv2ObjectDetector, DINOv2Backbone, DETRDecoder


def dev():
    return torch.device("cuda:0")


def to_gpu(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dev(), dtype).contiguous()


def load_np_state(module, sd_np, strip=""):
    sd = {k[len(strip):]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items() if k.startswith(strip)}
    module.load_state_dict(sd, strict=True)
    return module


def make_detector(bb, dc, precision, model_name="custom"):
    m = DINOv2ObjectDetector(num_classes=dc.num_classes, dino_model_name=model_name, lora_r=bb.lora_r,
                             lora_alpha=bb.lora_alpha, hidden_dim=dc.hidden_dim, num_queries=dc.num_queries,
                             nheads=dc.nheads, num_decoder_layers=dc.num_layers, dim_feedforward=dc.dim_feedforward,
                             n_points=dc.n_points, use_deformable=dc.use_deformable, pretrained=False,
                             precision=precision, backbone_config=bb)
    load_np_state(m, synth.detector_state_dict(bb, dc, seed=1))
    return m.to(dev()).eval()


def sync():
    torch.cuda.synchronize()


def op_linear(A, W, bias=None, scale=None, resid=None, act="none", out_dtype=torch.float32, out=None):
    """through dod_op_linear; A [M,K], W [N,K] both fp32 or both bf16 CUDA tensors (out: write there, e.g. in place)"""
    L = nat.lib()
    M, K = A.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype, device=A.device)
    in_dt = nat.DOD_BF16 if A.dtype == torch.bfloat16 else nat.DOD_F32
    out_dt = nat.DOD_BF16 if out_dtype == torch.bfloat16 else nat.DOD_F32
    rc = L.dod_op_linear(in_dt, nat.ptr(A), A.stride(0), nat.ptr(W), W.stride(0), M, N, K, nat.ptr(bias), nat.ptr(scale),
                         nat.ptr(resid), resid.stride(0) if resid is not None else 0, nat.ptr(out), out_dt, N,
                         nat.ACT[act], nat.stream_ptr())
    nat.check(rc)
    return out


def no_dropout(m):
    """every dropout site of a drop-in module to rate 0 (the reference's five decoder sites: nn.Dropout modules + the MHA's own)"""
    for mod in m.modules():
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
        elif isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m
