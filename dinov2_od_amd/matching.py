"""Hungarian matcher with the cost matrices computed on the GPU: the host side of `dod_match_cost` (include/dinodet.h).

Mirrors `dino_detector.matching.HungarianMatcher` (matching.py:9-121): same constructor, same `forward(outputs, targets)`
contract (a list of (index_i, index_j) int64 CPU tensors per image).  The focal / L1 / GIoU cost of every image
(matching.py:79-98) is one kernel launch over the packed detections and the concatenated targets, one device-to-host copy
brings all matrices back, and scipy's `linear_sum_assignment` solves them on the host as in the reference (:105).
No CPU fallback for the cost computation: without the HIP library this module raises.

Reference behaviour kept on purpose: the reference builds each image's matrix over ALL B*Q predictions and then keeps
`C[:num_queries]` (matching.py:102) -- the rows of image 0 for every image.  `per_image_rows=False` (default) reproduces
that; `per_image_rows=True` matches each image's targets against its own predictions.
"""
import numpy as np
import torch
import torch.nn as nn
from scipy.optimize import linear_sum_assignment

from . import _native as nat


def match_cost(det, num_classes, labels, gt_boxes, offsets, cost_class=1.0, cost_bbox=5.0, cost_giou=2.0,
               focal_alpha=0.25, focal_gamma=2.0, rows_from=0):
    """det: packed [B,Q,C+4] fp32 CUDA tensor; labels int64 [G], gt_boxes fp32 [G,4] (cx,cy,w,h), offsets int32 [B+1]
    (CUDA tensors).  Returns the flat fp32 CUDA buffer [G*Q]: image b's [Q, n_b] matrix at offsets[b]*Q."""
    if not (det.is_cuda and det.dtype == torch.float32 and det.dim() == 3):
        raise ValueError("det must be a CUDA fp32 tensor [B, Q, C+4]")
    B, Q, W = det.shape
    if W != num_classes + 4:
        raise ValueError(f"det has {W} columns, expected num_classes + 4 = {num_classes + 4}")
    G = int(labels.numel())
    if gt_boxes.numel() != 4 * G or offsets.numel() != B + 1:
        raise ValueError("targets: gt_boxes must be [G,4] and offsets [B+1]")
    cost = torch.empty(max(1, G * Q), dtype=torch.float32, device=det.device)
    nat.check(nat.lib().dod_match_cost(nat.ptr(det.contiguous()), B, Q, num_classes, nat.ptr(labels), nat.ptr(gt_boxes),
                                       nat.ptr(offsets), G, float(cost_class), float(cost_bbox), float(cost_giou),
                                       float(focal_alpha), float(focal_gamma), int(rows_from), nat.ptr(cost), nat.stream_ptr()))
    return cost[: G * Q]


class HungarianMatcher(nn.Module):
    """Drop-in for dino_detector.matching.HungarianMatcher (constructor matching.py:23-40, forward :43-121)."""

    def __init__(self, cost_class=1, cost_bbox=5, cost_giou=2, focal_alpha=0.25, focal_gamma=2.0, per_image_rows=False):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        self.focal_alpha, self.focal_gamma = focal_alpha, focal_gamma
        self.per_image_rows = per_image_rows
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "at least one cost should be non-zero"

    @torch.no_grad()
    def forward(self, outputs, targets):
        logits, boxes = outputs["pred_logits"], outputs["pred_boxes"]
        bs, num_queries, C = logits.shape
        dev = logits.device
        # views of one packed buffer (the engine's output) are used as they are; anything else is packed once
        det = torch.cat([logits, boxes], dim=-1).float().contiguous()
        counts, lab, box = [], [], []
        for t in targets:
            if len(t) == 0:                       # matching.py:73-75: an EMPTY DICT short-circuits
                counts.append(0)
                continue
            counts.append(int(t["labels"].numel()))
            lab.append(t["labels"].reshape(-1).to(dev, torch.int64))
            box.append(t["boxes"].reshape(-1, 4).to(dev, torch.float32))
        G = sum(counts)
        empty = (torch.tensor([], dtype=torch.int64), torch.tensor([], dtype=torch.int64))
        if G == 0:
            return [empty for _ in targets]
        labels = torch.cat(lab)
        if int(labels.min()) < 0 or int(labels.max()) >= C:
            raise IndexError("target label out of range")        # pos_cost_class[:, tgt_ids] would raise (matching.py:86)
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        cost = match_cost(det, C, labels, torch.cat(box).contiguous(), torch.from_numpy(offs).to(dev),
                          self.cost_class, self.cost_bbox, self.cost_giou, self.focal_alpha, self.focal_gamma,
                          -1 if self.per_image_rows else 0).cpu().numpy()
        out = []
        for b, n in enumerate(counts):
            if len(targets[b]) == 0:
                out.append(empty)
                continue
            Cm = cost[offs[b] * num_queries:(offs[b] + n) * num_queries].reshape(num_queries, n)
            i, j = linear_sum_assignment(Cm)                     # matching.py:105
            out.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
        return out


def build_matcher(args):
    """matching.py:124-134"""
    return HungarianMatcher(cost_class=args.set_cost_class, cost_bbox=args.set_cost_bbox, cost_giou=args.set_cost_giou,
                            focal_alpha=args.focal_alpha, focal_gamma=args.focal_gamma)
