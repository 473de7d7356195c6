"""CPU oracle for the input pipeline row -- TEST INFRASTRUCTURE ONLY.

The reference's transform (dino_detector/train.py:584-587) is torchvision `Resize((224, 224))` + `ToTensor()` on a PIL
image: `Image.resize(size, BILINEAR)` followed by uint8 -> float32 / 255 in CHW order.  The arithmetic lives in a third-party
dependency absent from /root/reference: Pillow (installed here: 12.2.0; `requirements.txt` pins nothing), file
src/libImaging/Resample.c.  This module restates its published algorithm for 8-bit RGB:
  * precompute_coeffs: per output index, window [xmin, xmin+xmax) and bilinear weights in double, normalised by their sum
    (support = max(scale, 1): the filter widens when downscaling -- PIL's resize is antialiased);
  * normalize_coeffs_8bpc: weights -> int32 fixed point with PRECISION_BITS = 22, round half away from zero;
  * horizontal pass over the rows the vertical pass needs, into a uint8 image (accumulator seeded with 1 << 21, shifted by 22,
    clipped to 0..255), then the vertical pass on that uint8 image.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

PINNING: bit-exact against Pillow itself on random images of many sizes, up- and down-scaling
(tests/test_oracle_golden.py::test_preprocess_oracle_matches_pillow; Pillow is importable in the authoring container and on the
GPU box alike, so the check runs wherever the CPU suite runs).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc for the full-image box: (bounds [out,2] int, kk [out,ksize] int32)"""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            t = -t if t < 0.0 else t
            w[x] = 1.0 - t if t < 1.0 else 0.0
            ww += w[x]
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
        for x in range(ksize):
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """one resampling pass along `axis` (0 = vertical, 1 = horizontal) of a uint8 [H, W, C] image"""
    img = np.moveaxis(img, axis, 0).astype(np.int64)          # resampled axis first
    out = np.empty((len(bounds),) + img.shape[1:], dtype=np.uint8)
    for i, (lo, n) in enumerate(bounds):
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(n):
            acc += img[lo + x] * kk[i, x]
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img, out_h, out_w):
    """PIL.Image.resize((out_w, out_h), BILINEAR) on an RGB uint8 [H, W, 3] array -> uint8 [out_h, out_w, 3]"""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape[:2]
    need_h, need_v = W != out_w, H != out_h
    bh, kh = _coeffs(W, out_w)
    bv, kv = _coeffs(H, out_h)
    first = int(bv[0, 0])
    last = int(bv[-1, 0] + bv[-1, 1])
    if need_h:
        bv = bv.copy()
        bv[:, 0] -= first
        img = _pass(img[first:last], bh, kh, 1)
    if need_v:
        img = _pass(img, bv, kv, 0)
    return img


def to_tensor(img_u8):
    """torchvision ToTensor on a uint8 HWC image: CHW float32 in [0, 1] (value / 255 in fp32)"""
    return (np.transpose(img_u8, (2, 0, 1)).astype(np.float32) / np.float32(255.0)).astype(np.float32)


def preprocess(images, out_h, out_w):
    """list of uint8 [H_i, W_i, 3] -> float32 [B, 3, out_h, out_w] (train.py:584-587)"""
    return np.stack([to_tensor(resize_bilinear_u8(im, out_h, out_w)) for im in images])
