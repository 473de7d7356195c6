"""The train entry point on the drop-in pieces together (reference flow: train.py:1079-1101 + losses.py:222 + matching.py):
train()-mode forward (native frozen prefix + autograd composite), Hungarian matching with the device cost kernel, a DETR-style
loss on the matched pairs, backward, optimizer step -- the loss goes down and the next eval() forward (native kernels) sees the
updated weights."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dinov2_od_amd import synth
from tests import cases

pytestmark = pytest.mark.gpu


def _loss(out, targets, indices, num_classes):
    """classification (background = class 0 for unmatched queries) + L1 on matched boxes, as losses.py:100-190 does in outline"""
    logits, boxes = out["pred_logits"], out["pred_boxes"]
    tgt_cls = torch.zeros(logits.shape[:2], dtype=torch.int64, device=logits.device)
    l1 = logits.new_zeros(())
    n = 0
    for b, (i, j) in enumerate(indices):
        if len(i) == 0:
            continue
        tgt_cls[b, i] = targets[b]["labels"][j]
        l1 = l1 + (boxes[b, i] - targets[b]["boxes"][j]).abs().sum()
        n += len(i)
    ce = F.cross_entropy(logits.flatten(0, 1), tgt_cls.flatten())
    return ce + 5.0 * l1 / max(n, 1)


def test_few_training_steps_reduce_the_loss():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dinov2_od_amd.matching import HungarianMatcher
    from tests import gpu_util as G
    torch.manual_seed(0)
    bb, dc = cases.cfg1(25)
    m = G.make_detector(bb, dc, "bf16", "facebook/dinov2-small")
    matcher = HungarianMatcher(per_image_rows=True)
    x = G.to_gpu(synth.make_pixels(4, 112, 112, seed=0))
    rng = np.random.default_rng(0)
    targets = []
    for b in range(4):
        n = int(rng.integers(1, 5))
        cxcy = 0.2 + 0.6 * rng.random((n, 2))
        wh = 0.1 + 0.2 * rng.random((n, 2))
        targets.append({"labels": torch.from_numpy(rng.integers(1, dc.num_classes, n)).cuda(),
                        "boxes": torch.from_numpy(np.concatenate([cxcy, wh], 1).astype(np.float32)).cuda()})
    m.eval()
    before = m.forward_packed(x).clone()
    opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=2e-3)
    losses = []
    m.train()
    for _ in range(8):
        out = m(x)
        idx = matcher({"pred_logits": out["pred_logits"].detach(), "pred_boxes": out["pred_boxes"].detach()}, targets)
        loss = _loss(out, targets, idx, dc.num_classes)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses))
    assert min(losses[-3:]) < 0.8 * losses[0], losses
    m.eval()
    after = m.forward_packed(x)
    assert torch.isfinite(after).all() and not torch.allclose(before, after)
