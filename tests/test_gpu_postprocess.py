"""GPU parity of the device post-processing (dod_postprocess) against the reference's evaluate_coco output
(golden G5) and the numpy oracle.  Integer / index fields and the fp32 box arithmetic are bit-exact; the score carries the
sigmoid implementation's last-ulp difference (tolerance 3e-7 absolute on values in (0.05, 1))."""
import json

import numpy as np
import pytest
import torch

from dinov2_od_amd import synth
from oracle import postprocess_oracle as ppo
from tests import cases

pytestmark = pytest.mark.gpu
SCORE_TOL = 3e-7


@pytest.fixture(scope="module")
def pp():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dinov2_od_amd import postprocess
    return postprocess


def _safe_logits(seed, shape, mean, std):
    """logits whose scores keep clear of the 0.05 threshold by 1e-4 (the keep decision must not hinge on an ulp)"""
    x = (synth.normal(seed, f"ppt.{shape}", shape, std) + mean).astype(np.float32)
    s = 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
    x[np.abs(s - 0.05) < 1e-4] -= 0.05
    return x


def _check(rec, want):
    assert len(rec) == len(want["image_id"])
    assert np.array_equal(rec["image_id"], want["image_id"])
    assert np.array_equal(rec["category_id"], want["category_id"])
    if "query" in want:
        assert np.array_equal(rec["query"], want["query"])
    assert np.array_equal(rec["bbox"].view(np.uint32), np.asarray(want["bbox"], dtype=np.float32).view(np.uint32))
    if len(rec):
        assert np.max(np.abs(rec["score"].astype(np.float64) - np.asarray(want["score"], dtype=np.float64))) < SCORE_TOL


def test_matches_reference_evaluate_coco_golden(pp):
    g = cases.golden("g5_postprocess")
    det = g["det"]
    C = det.shape[-1] - 4
    recs = []
    for (lo, hi), ids in zip(g["batch_bounds"], g["image_ids"]):
        ids = [None if v < 0 else int(v) for v in ids[: hi - lo]]
        recs.append(pp.postprocess_packed(torch.from_numpy(det[lo:hi]).cuda(), C, ids, 0.05))
    rec = np.concatenate(recs)
    _check(rec, {"image_id": g["r_image_id"], "category_id": g["r_category_id"], "bbox": g["r_bbox"], "score": g["r_score"]})


@pytest.mark.parametrize("B,Q,C,mean", [(64, 100, 91, -4.0), (8, 300, 91, -3.0), (3, 7, 2, -2.0), (1, 1, 5, -2.9), (256, 100, 91, -6.0)])
def test_matches_oracle(pp, B, Q, C, mean):
    logits = _safe_logits(21, (B, Q, C), mean, 1.5)
    boxes = synth.uniform01(22, f"ppt.boxes.{B}.{Q}", (B, Q, 4)).astype(np.float32)
    det = np.concatenate([logits, boxes], axis=-1)
    ids = [1000 + 3 * i for i in range(B)]
    want = ppo.postprocess(det, C, ids, 0.05)
    rec = pp.postprocess_packed(torch.from_numpy(det).cuda(), C, ids, 0.05)
    _check(rec, want)
    # sortedness: the reference's (image, class, query) order, strictly increasing
    key = (rec["image_id"].astype(np.int64) - 1000) // 3 * (C * Q) + rec["category_id"].astype(np.int64) * Q + rec["query"]
    assert np.all(np.diff(key) > 0)
    assert rec["category_id"].min(initial=1) >= 1          # background never emitted


def test_none_kept_all_kept_and_truncation(pp):
    B, Q, C = 4, 10, 6
    boxes = synth.uniform01(5, "ppt.b2", (B, Q, 4)).astype(np.float32)
    lo = np.concatenate([np.full((B, Q, C), -20.0, np.float32), boxes], -1)
    assert len(pp.postprocess_packed(torch.from_numpy(lo).cuda(), C)) == 0
    hi = np.concatenate([np.full((B, Q, C), 3.0, np.float32), boxes], -1)
    rec = pp.postprocess_packed(torch.from_numpy(hi).cuda(), C)
    assert len(rec) == B * Q * (C - 1)
    want = ppo.postprocess(hi, C)
    _check(rec, want)
    cut = pp.postprocess_packed(torch.from_numpy(hi).cuda(), C, max_out=37)   # capacity below the total: a prefix
    assert len(cut) == 37
    _check(cut, {k: v[:37] for k, v in want.items()})


def test_threshold_and_image_id_forms(pp):
    B, Q, C = 2, 9, 4
    det = np.concatenate([_safe_logits(31, (B, Q, C), 0.0, 2.0), synth.uniform01(6, "ppt.b3", (B, Q, 4)).astype(np.float32)], -1)
    d = torch.from_numpy(det).cuda()
    for thr in (0.05, 0.5, 0.9):
        s = 1.0 / (1.0 + np.exp(-det[..., :C].astype(np.float64)))
        if np.any(np.abs(s - thr) < 1e-5):
            continue
        _check(pp.postprocess_packed(d, C, None, thr), ppo.postprocess(det, C, None, thr))
    ids = torch.tensor([2 ** 40 + 5, 9], dtype=torch.int64, device="cuda")     # int64 ids, as a device tensor
    rec = pp.postprocess_packed(d, C, ids)
    assert set(np.unique(rec["image_id"])) <= {2 ** 40 + 5, 9}
    _check(rec, ppo.postprocess(det, C, [2 ** 40 + 5, 9]))


def test_rejects_bad_arguments(pp):
    with pytest.raises(ValueError):
        pp.postprocess_packed(torch.zeros(2, 3, 9), 5)                       # CPU tensor
    with pytest.raises(ValueError):
        pp.postprocess_packed(torch.zeros(2, 3, 9, device="cuda"), 4)        # 4 + 4 != 9
    with pytest.raises(ValueError):
        pp.postprocess_packed(torch.zeros(2, 3, 9, device="cuda"), 5, [1])   # one id for two images


def test_evaluate_coco_mirror(pp, tmp_path):
    """same call as dino_detector.utils.evaluate_coco(model, dataloader, device, output_file) with a fixed-output model"""
    g = cases.golden("g5_postprocess")
    det = g["det"]
    C = det.shape[-1] - 4

    class Fixed(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.i = 0

        def forward_packed(self, images):
            lo, hi = g["batch_bounds"][self.i]
            self.i += 1
            return torch.from_numpy(det[lo:hi]).to(images.device)

    loader = []
    for (lo, hi), ids in zip(g["batch_bounds"], g["image_ids"]):
        loader.append((torch.zeros(hi - lo, 3, 8, 8), [({"image_id": int(v)} if v >= 0 else {}) for v in ids[: hi - lo]]))
    out = tmp_path / "preds.json"
    res = pp.evaluate_coco(Fixed(), loader, torch.device("cuda"), str(out))
    assert [r["image_id"] for r in res] == g["r_image_id"].tolist()
    assert [r["category_id"] for r in res] == g["r_category_id"].tolist()
    assert np.array_equal(np.array([r["bbox"] for r in res]), g["r_bbox"])
    assert np.max(np.abs(np.array([r["score"] for r in res]) - g["r_score"])) < SCORE_TOL
    assert set(res[0]) == {"image_id", "category_id", "bbox", "score"} and isinstance(res[0]["bbox"][0], float)
    assert json.load(open(out)) == res
