"""CPU oracle for the matcher cost matrices -- TEST INFRASTRUCTURE ONLY.

Restates HungarianMatcher.forward, dino_detector/matching.py:60-121, with torch on the CPU (fp32): the per-image cost
matrix handed to scipy and the resulting assignment.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module.

PINNING: `tests/golden/g6_matcher.npz` holds the cost matrices the reference's own matcher passed to
`linear_sum_assignment` (captured by wrapping that function in the reference module's namespace) and the indices it
returned, for two weight settings; `tests/test_oracle_golden.py` checks this restatement against them.
"""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment


def _xyxy(b):                                       # utils.py:83-88
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def _giou(b1, b2):                                  # utils.py:124-164
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = a1[:, None] + a2 - inter
    iou = inter / union
    lte = torch.min(b1[:, None, :2], b2[:, :2])
    rbe = torch.max(b1[:, None, 2:], b2[:, 2:])
    whe = (rbe - lte).clamp(min=0)
    ae = whe[:, :, 0] * whe[:, :, 1]
    return iou - (ae - union) / ae


def cost_matrices(det, num_classes, labels, gt_boxes, offsets, cost_class=1.0, cost_bbox=5.0, cost_giou=2.0,
                  focal_alpha=0.25, focal_gamma=2.0, rows_from=0):
    """det packed [B,Q,C+4]; concatenated targets.  Returns a list of B arrays [Q, n_b] fp32.  rows_from=0 is the
    reference's behaviour (matching.py:102 keeps the first Q rows of the all-predictions matrix = image 0); -1 = own rows."""
    det = torch.as_tensor(np.asarray(det), dtype=torch.float32)
    B, Q, W = det.shape
    C = num_classes
    prob = det[..., :C].flatten(0, 1).sigmoid()                                     # :63
    bbox = det[..., C:].flatten(0, 1)                                               # :64
    labels = torch.as_tensor(np.asarray(labels), dtype=torch.int64)
    gt = torch.as_tensor(np.asarray(gt_boxes), dtype=torch.float32).reshape(-1, 4)
    out = []
    for b in range(B):
        ids, tb = labels[offsets[b]:offsets[b + 1]], gt[offsets[b]:offsets[b + 1]]
        neg = (1 - focal_alpha) * (prob ** focal_gamma) * (-(1 - prob + 1e-8).log())     # :82
        pos = focal_alpha * ((1 - prob) ** focal_gamma) * (-(prob + 1e-8).log())         # :83
        cc = pos[:, ids] - neg[:, ids]                                                   # :86
        cb = torch.cdist(bbox, tb, p=1) if len(ids) else torch.zeros(bbox.shape[0], 0)   # :89
        cg = -_giou(_xyxy(bbox), _xyxy(tb)) if len(ids) else torch.zeros(bbox.shape[0], 0)   # :92-95
        Cm = cost_class * cc + cost_bbox * cb + cost_giou * cg                           # :98
        src = b if rows_from < 0 else rows_from
        out.append(Cm[src * Q:(src + 1) * Q].numpy().astype(np.float32))                 # :102 (src = 0 in the reference)
    return out


def assign(cost_list):
    """scipy assignment per image (matching.py:105-107) -> list of (idx_i, idx_j) int64 arrays"""
    res = []
    for Cm in cost_list:
        i, j = linear_sum_assignment(Cm)
        res.append((np.asarray(i, dtype=np.int64), np.asarray(j, dtype=np.int64)))
    return res
