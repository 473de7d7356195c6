"""Training-mode forward: a PyTorch-ROCm composite of the same math, with autograd.

The native HIP path is inference-only (no backward kernels yet).  So that the reference's train loop
(`train.py:1079-1101`: `model.train()`, forward, loss, `backward()`, optimizer step) runs on the drop-in modules, a module in
`train()` mode evaluates this composite on the module's OWN parameters -- gradients reach exactly the parameters the
reference trains (LoRA A/B, projection, decoder, heads; the DINOv2 weights stay frozen, dinov2_backbone.py:40-41) -- with the
reference's dropout placement (deformable_attention.py:195-209, 235, 261, 265-266).  `eval()` mode always runs the native
kernels; after an optimizer step the engine re-packs the changed weights automatically (engine.sync_weights).
On the GPU the frozen prefix of the backbone (embeddings + every block before the first LoRA-adapted one: 10 of ViT-B's 12)
runs in the native kernels; only the adapted blocks, the projection and the decoder are evaluated here.
This file is the stop-gap the survey describes, not the measured hot path.
"""
import os

import torch
import torch.nn.functional as F

from ..config import spatial_factor


def _lin(mod, x):
    """nn.Linear, or the LoraLinear container (dino_detector/utils.py:68-70)"""
    if hasattr(mod, "lora_A"):
        return F.linear(x, mod.linear.weight, mod.linear.bias) + mod.alpha * F.linear(F.linear(x, mod.lora_A.weight), mod.lora_B.weight)
    return F.linear(x, mod.weight, mod.bias)


def _pos_embed(pos, patch, H, W):
    """modeling_dinov2.py:57-95: bicubic resize of the patch position grid (none when the grid already matches)"""
    gh, gw = H // patch, W // patch
    npos = pos.shape[1] - 1
    if gh * gw == npos and H == W:
        return pos
    D = pos.shape[-1]
    g = int(npos ** 0.5)
    p = pos[:, 1:].reshape(1, g, g, D).permute(0, 3, 1, 2)
    p = F.interpolate(p.float(), size=(gh, gw), mode="bicubic", align_corners=False).to(pos.dtype)
    return torch.cat((pos[:, :1], p.permute(0, 2, 3, 1).reshape(1, -1, D)), dim=1)


def backbone_forward(m, pixel_values):
    """DINOv2Backbone.forward (dinov2_backbone.py:58-67) over HF Dinov2Model's arithmetic (modeling_dinov2.py)"""
    bb, dino = m._bb_cfg, m.dino
    emb = dino.embeddings
    B, Cc, H, W = pixel_values.shape
    if Cc != 3:
        raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in the "
                         f"configuration. Expected 3 but got {Cc}.")
    layers = list(dino.encoder.layer)
    # The DINOv2 weights are frozen and only the last blocks carry LoRA adapters (dinov2_backbone.py:40-51): every block before
    # the first adapted one needs no autograd, so on the GPU that prefix runs in the native kernels (dod_backbone_prefix).
    first_trainable = next((i for i, L in enumerate(layers) if any(p.requires_grad for p in L.parameters())), len(layers))
    frozen_front = not any(p.requires_grad for p in emb.parameters())
    if (pixel_values.is_cuda and frozen_front and first_trainable > 0 and not pixel_values.requires_grad
            and os.environ.get("DINODET_COMPOSITE_FULL") != "1"):     # (=1: every block in torch, for A/B timing)
        with torch.no_grad():
            h = m._get_engine().backbone_prefix(pixel_values, m._engine_named(), first_trainable)
        layers = layers[first_trainable:]
        # the adapted blocks, the final LayerNorm and the projection on the native kernels with their hand-written backward
        from . import _native_train
        if os.environ.get("DINODET_NATIVE_TRAIN", "1") != "0" and _native_train.tail_supported(m, layers, h):
            return _native_train.backbone_tail(m, h, layers)
    else:
        x = emb.patch_embeddings.projection(pixel_values.to(emb.patch_embeddings.projection.weight.dtype)).flatten(2).transpose(1, 2)      # :141-149
        h = torch.cat((emb.cls_token.expand(B, -1, -1), x), dim=1) + _pos_embed(emb.position_embeddings, bb.patch, H, W)
    nh = bb.heads
    for L in layers:                                                                         # :361-380
        y = L.norm1(h)
        a = L.attention.attention
        N = y.shape[1]
        q, k, v = (_lin(t, y).view(B, N, nh, -1).transpose(1, 2) for t in (a.query, a.key, a.value))
        ctx = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, -1)
        h = _lin(L.attention.output.dense, ctx) * L.layer_scale1.lambda1 + h
        y = L.norm2(h)
        if bb.swiglu:                                                                        # :310-314
            x1, x2 = _lin(L.mlp.weights_in, y).chunk(2, dim=-1)
            z = _lin(L.mlp.weights_out, F.silu(x1) * x2)
        else:                                                                                # :293-297
            z = _lin(L.mlp.fc2, F.gelu(_lin(L.mlp.fc1, y)))
        h = z * L.layer_scale2.lambda1 + h
    h = dino.layernorm(h)
    return m.projection(h) if m.projection is not None else h


def deformable_sample(values, ref, offsets, weights, h, w):
    """differentiable form of deformable_attention.py:101-174 (values [B,HW,Hd,dh], ref [B,Q,2], offsets [B,Q,Hd,P,2],
    weights [B,Q,Hd,P] softmaxed) -> [B,Q,Hd,dh]; same clamping / floor / bilinear weights as deform.hip"""
    B, HW, Hd, dh = values.shape
    Q, P = ref.shape[1], offsets.shape[3]
    loc = torch.clamp(ref[:, :, None, None, :] + offsets, 0, 1)
    lx, ly = loc[..., 0] * (w - 1), loc[..., 1] * (h - 1)
    x0, y0 = torch.floor(lx).long(), torch.floor(ly).long()
    x1, y1 = (x0 + 1).clamp(0, w - 1), (y0 + 1).clamp(0, h - 1)
    x0, y0 = x0.clamp(0, w - 1), y0.clamp(0, h - 1)
    wx1, wy1 = lx - x0.to(lx.dtype), ly - y0.to(ly.dtype)
    wx0, wy0 = 1.0 - wx1, 1.0 - wy1
    vh = values.permute(0, 2, 1, 3)

    def gather(yy, xx):
        idx = (yy * w + xx).permute(0, 2, 1, 3).reshape(B, Hd, Q * P)
        g = torch.gather(vh, 2, idx[..., None].expand(-1, -1, -1, dh))
        return g.view(B, Hd, Q, P, dh).permute(0, 2, 1, 3, 4)

    res = (gather(y0, x0) * (wx0 * wy0)[..., None] + gather(y1, x0) * (wx0 * wy1)[..., None]
           + gather(y0, x1) * (wx1 * wy0)[..., None] + gather(y1, x1) * (wx1 * wy1)[..., None])
    return (res * weights[..., None]).sum(dim=3)


def decoder_forward(m, src):
    """DETRDecoder.forward (detr_decoder.py:47-83) with the layers of deformable_attention.py:215-268"""
    dc = m._dc_cfg
    B, N, Dd = src.shape
    Q, Hd, P = dc.num_queries, dc.nheads, dc.n_points
    tgt = m.query_embed.weight.unsqueeze(0).repeat(B, 1, 1)
    if m.use_deformable:
        h, w = spatial_factor(N)
        for layer in m.decoder.layers:
            q = tgt.transpose(0, 1)
            tgt = layer.norm1(tgt + layer.dropout1(layer.self_attn(q, q, q)[0].transpose(0, 1)))
            ref = layer.reference_points_proj(tgt).sigmoid()
            ca = layer.cross_attn
            off = ca.sampling_offsets(tgt).view(B, Q, Hd, P, 2)
            aw = ca.attention_weights(tgt).view(B, Q, Hd, P).softmax(-1)
            val = ca.value_proj(src).view(B, N, Hd, Dd // Hd)
            samp = deformable_sample(val, ref, off, aw, h, w).reshape(B, Q, Dd)
            tgt = layer.norm2(tgt + layer.dropout2(ca.output_proj(samp)))
            tgt = layer.norm3(tgt + layer.dropout4(layer.linear2(layer.dropout3(F.relu(layer.linear1(tgt))))))
        hs = tgt
    else:
        hs = m.decoder(tgt.permute(1, 0, 2), src.permute(1, 0, 2)).transpose(0, 1)
    return {"pred_logits": m.class_embed(hs), "pred_boxes": m.bbox_embed.mlp(hs).sigmoid()}
