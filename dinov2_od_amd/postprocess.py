"""Detection post-processing on the GPU: the host side of `dod_postprocess` (include/dinodet.h).

Mirrors the reference's `evaluate_coco(model, dataloader, device, output_file=None)` (dino_detector/utils.py:167-240): same
signature, same list of `{'image_id', 'category_id', 'bbox', 'score'}` dicts in the same order, same JSON dump -- but the
sigmoid / threshold / box conversion / compaction of utils.py:195-233 (a triple Python loop with a `.cpu().numpy()` per
class) runs as three small HIP kernels over the packed detections, and only the kept records cross PCIe.
No CPU fallback: without the HIP library this module raises.
"""
import ctypes as C
import json

import numpy as np
import torch

from . import _native as nat

RECORD_DTYPE = np.dtype([("image_id", "<i8"), ("category_id", "<i4"), ("query", "<i4"), ("bbox", "<f4", (4,)),
                         ("score", "<f4"), ("reserved", "<i4")])
assert RECORD_DTYPE.itemsize == C.sizeof(nat.DodDetection) == 40


def postprocess_packed(det, num_classes, image_ids=None, threshold=0.05, max_out=None):
    """det: packed detections [B, Q, C+4] fp32 on the GPU (DINOv2ObjectDetector.forward_packed).  image_ids: sequence of
    ints / None (None -> the index in the batch, utils.py:203) or an int64 CUDA tensor.
    Returns a numpy structured array (RECORD_DTYPE) of the kept detections in the reference's order."""
    if not (isinstance(det, torch.Tensor) and det.is_cuda and det.dtype == torch.float32 and det.dim() == 3):
        raise ValueError("det must be a CUDA fp32 tensor [B, Q, C+4]")
    B, Q, W = det.shape
    Cn = int(num_classes)
    if W != Cn + 4 or Cn < 2:
        raise ValueError(f"det has {W} columns, expected num_classes + 4 = {Cn + 4}")
    det = det.contiguous()
    L = nat.lib()
    ids_t = None
    if image_ids is not None:
        if isinstance(image_ids, torch.Tensor):
            ids_t = image_ids.to(device=det.device, dtype=torch.int64).contiguous()
        else:
            ids_t = torch.tensor([i if v is None else int(v) for i, v in enumerate(image_ids)], dtype=torch.int64,
                                 device=det.device)
        if ids_t.numel() != B:
            raise ValueError("image_ids must have one entry per image")
    cap = B * Q * (Cn - 1) if max_out is None else int(max_out)
    ws = torch.empty(max(1, L.dod_postprocess_workspace_bytes(B, Q, Cn)), dtype=torch.uint8, device=det.device)
    out = torch.empty(max(1, cap) * RECORD_DTYPE.itemsize, dtype=torch.uint8, device=det.device)
    count = torch.zeros(1, dtype=torch.int64, device=det.device)
    nat.check(L.dod_postprocess(nat.ptr(det), B, Q, Cn, nat.ptr(ids_t), float(threshold), nat.ptr(out), cap, nat.ptr(count),
                                nat.ptr(ws), ws.numel(), nat.stream_ptr()))
    n = min(int(count.item()), cap)                       # the one host sync: how many records to fetch
    return out[: n * RECORD_DTYPE.itemsize].cpu().numpy().view(RECORD_DTYPE).copy()


def records_to_coco(rec):
    """structured records -> the reference's list of dicts (utils.py:227-232): python ints / floats"""
    return [{"image_id": int(r["image_id"]), "category_id": int(r["category_id"]),
             "bbox": [float(v) for v in r["bbox"]], "score": float(r["score"])} for r in rec]


def evaluate_coco(model, dataloader, device, output_file=None, threshold=0.05):
    """Drop-in for dino_detector.utils.evaluate_coco (utils.py:167-240) for the engine-backed detector."""
    model.eval()
    results = []
    with torch.no_grad():
        for images, targets in dataloader:
            images = images.to(device)
            if hasattr(model, "forward_packed"):
                det = model.forward_packed(images)
                ncls = det.shape[-1] - 4
            else:                                            # any module with the reference's output dict
                o = model(images)
                det = torch.cat([o["pred_logits"], o["pred_boxes"]], dim=-1).float()
                ncls = o["pred_logits"].shape[-1]
            ids = [t.get("image_id", None) for t in targets]   # utils.py:203: default = index in the batch
            ids = [None if v is None else int(v) for v in ids]
            results.extend(records_to_coco(postprocess_packed(det, ncls, ids, threshold)))
    if output_file is not None:
        with open(output_file, "w") as f:
            json.dump(results, f)
    return results
