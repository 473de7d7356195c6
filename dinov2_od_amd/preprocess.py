"""Input pipeline on the GPU: the host side of `dod_preprocess` (include/dinodet.h).

The reference prepares every image on the host with `transforms.Compose([Resize((224, 224)), ToTensor()])`
(dino_detector/train.py:584-587: Pillow's antialiased bilinear resample, then uint8 -> float32 / 255 in CHW order) inside the
DataLoader workers.  `preprocess_batch` does the same for a ragged batch of decoded uint8 images in two kernels, bit-exact
against Pillow, and returns the `[B, 3, R, R]` fp32 batch already resident on the device for `model(images)`.
No CPU fallback: without the HIP library this module raises.
"""
import numpy as np
import torch

from . import _native as nat


_STAGE = None
_STAGE_BUSY = None      # event recorded after the last async copy out of the staging buffer


def _staging(nbytes):
    global _STAGE
    if _STAGE_BUSY is not None:
        _STAGE_BUSY.synchronize()          # the previous batch's DMA must have left the buffer before it is refilled
    if _STAGE is None or _STAGE.numel() < nbytes:
        _STAGE = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, pin_memory=True)
    return _STAGE


def preprocess_batch(images, size=(224, 224), device="cuda", as_uint8=False):
    """images: sequence of uint8 RGB images [H_i, W_i, 3] (numpy arrays, torch tensors or PIL images); size: (height, width)
    as torchvision's Resize takes it.  Returns float32 [B, 3, height, width] on `device` -- or, with as_uint8, the resampled
    bytes uint8 [B, height, width, 3] (before ToTensor) for `model.forward_packed_u8`, whose patch-embedding kernel divides by 255
    in its load stage: the fp32 batch then never goes through HBM."""
    out_h, out_w = (int(size), int(size)) if np.isscalar(size) else (int(size[0]), int(size[1]))
    arrs = []
    for im in images:
        a = im.cpu().numpy() if isinstance(im, torch.Tensor) else np.asarray(im)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError(f"expected uint8 RGB images [H, W, 3], got {a.dtype} {a.shape}")
        arrs.append(np.ascontiguousarray(a))
    if not arrs:
        raise ValueError("empty batch")
    B = len(arrs)
    hs = np.array([a.shape[0] for a in arrs], dtype=np.int32)
    ws = np.array([a.shape[1] for a in arrs], dtype=np.int32)
    sizes = hs.astype(np.int64) * ws * 3
    src_offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    tsz = hs.astype(np.int64) * out_w * 3
    tmp_offs = np.concatenate([[0], np.cumsum(tsz)[:-1]]).astype(np.int64)
    dev = torch.device(device)
    # the raw bytes are gathered straight into a (cached) pinned staging buffer: one host copy, one DMA to the device
    total = int(sizes.sum())
    stage = _staging(total)
    sv = stage.numpy()
    for a, o in zip(arrs, src_offs):
        sv[o:o + a.size] = a.reshape(-1)
    src = stage[:total].to(dev, non_blocking=True)
    global _STAGE_BUSY
    _STAGE_BUSY = torch.cuda.Event()
    _STAGE_BUSY.record()
    meta = [torch.from_numpy(x).to(dev) for x in (src_offs, hs, ws, tmp_offs)]
    tmp = torch.empty(int(tsz.sum()), dtype=torch.uint8, device=dev)
    if as_uint8:
        out = torch.empty(B, out_h, out_w, 3, dtype=torch.uint8, device=dev)
        nat.check(nat.lib().dod_preprocess_u8(nat.ptr(src), nat.ptr(meta[0]), nat.ptr(meta[1]), nat.ptr(meta[2]), B, int(hs.max()), int(ws.max()),
                                              out_h, out_w, nat.ptr(tmp), nat.ptr(meta[3]), nat.ptr(out), nat.stream_ptr()))
        return out
    out = torch.empty(B, 3, out_h, out_w, dtype=torch.float32, device=dev)
    nat.check(nat.lib().dod_preprocess(nat.ptr(src), nat.ptr(meta[0]), nat.ptr(meta[1]), nat.ptr(meta[2]), B, int(hs.max()), int(ws.max()),
                                       out_h, out_w, nat.ptr(tmp), nat.ptr(meta[3]), nat.ptr(out), nat.stream_ptr()))
    return out


class ResizeToTensor:
    """Batch counterpart of `transforms.Compose([transforms.Resize(size), transforms.ToTensor()])` (train.py:584-587)."""

    def __init__(self, size=(224, 224), device="cuda", as_uint8=False):
        self.size, self.device, self.as_uint8 = size, device, as_uint8

    def __call__(self, images):
        return preprocess_batch(images, self.size, self.device, self.as_uint8)
