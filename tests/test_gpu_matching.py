"""GPU parity of the matcher cost kernel (dod_match_cost) and the HungarianMatcher mirror against the
reference's own cost matrices / assignments (golden G6) and the torch-CPU oracle.  fp32 arithmetic: cost entries within
2e-6 * max(1, |C|max) (sigmoid / log / pow implementations and FMA contraction differ in the last ulps); indices exact."""
import types

import numpy as np
import pytest
import torch

from dinov2_od_amd import synth
from oracle import matching_oracle as mo
from tests import cases

pytestmark = pytest.mark.gpu
TOL = 2e-6
KW2 = dict(cost_class=2.0, cost_bbox=1.0, cost_giou=3.0, focal_alpha=0.4, focal_gamma=1.5)


@pytest.fixture(scope="module")
def mt():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dinov2_od_amd import matching
    return matching


def _targets(labels, gt, offs, dev="cpu"):
    return [{"labels": torch.from_numpy(labels[offs[b]:offs[b + 1]]).to(dev), "boxes": torch.from_numpy(gt[offs[b]:offs[b + 1]]).to(dev)}
            for b in range(len(offs) - 1)]


def _outputs(det, C):
    d = torch.from_numpy(det).cuda()
    return {"pred_logits": d[..., :C], "pred_boxes": d[..., C:]}


@pytest.mark.parametrize("tag,kw", [("default", {}), ("g15", KW2)])
def test_cost_and_assignment_match_reference_golden(mt, tag, kw):
    g = cases.golden("g6_matcher")
    det, labels, gt, offs = g["det"], g["labels"], g["gt"], g["offs"]
    B, Q = det.shape[:2]
    C = det.shape[-1] - 4
    cost = mt.match_cost(torch.from_numpy(det).cuda(), C, torch.from_numpy(labels).cuda(), torch.from_numpy(gt).cuda(),
                         torch.from_numpy(offs).cuda(), rows_from=0,
                         **{k.replace("focal_", "focal_"): v for k, v in kw.items()}).cpu().numpy()
    for b in range(B):
        want = g[f"{tag}_cost{b}"]
        got = cost[offs[b] * Q:offs[b + 1] * Q].reshape(Q, -1)
        assert got.shape == want.shape
        if want.size:
            assert np.max(np.abs(got - want)) <= TOL * max(1.0, np.abs(want).max())
    idx = mt.HungarianMatcher(**kw)(_outputs(det, C), _targets(labels, gt, offs))
    for b in range(B):
        assert idx[b][0].dtype == torch.int64 and idx[b][0].device.type == "cpu"
        assert np.array_equal(idx[b][0].numpy(), g[f"{tag}_i{b}"]) and np.array_equal(idx[b][1].numpy(), g[f"{tag}_j{b}"])
    # an empty target dict short-circuits (matching.py:73-75); a dict with zero boxes goes through scipy with [Q,0]
    idx = mt.HungarianMatcher()(_outputs(det, C), [{}] + _targets(labels, gt, offs)[1:])
    assert len(idx[0][0]) == 0 and len(idx[1][0]) == 0


@pytest.mark.parametrize("B,Q,C,rows_from", [(64, 100, 91, 0), (64, 100, 91, -1), (8, 300, 91, -1), (2, 5, 3, 0)])
def test_full_size_against_oracle_and_properties(mt, B, Q, C, rows_from):
    rng_counts = (synth.uniform01(3, f"mt.counts.{B}", (B,)) * 31).astype(int)       # 0..30 targets per image
    det, labels, gt, offs = _inputs(B, Q, C, rng_counts)
    want = mo.cost_matrices(det, C, labels, gt, offs, rows_from=rows_from)
    cost = mt.match_cost(torch.from_numpy(det).cuda(), C, torch.from_numpy(labels).cuda(), torch.from_numpy(gt).cuda(),
                         torch.from_numpy(offs).cuda(), rows_from=rows_from).cpu().numpy()
    for b in range(B):
        got = cost[offs[b] * Q:offs[b + 1] * Q].reshape(Q, -1)
        if want[b].size:
            assert np.max(np.abs(got - want[b])) <= TOL * max(1.0, np.abs(want[b]).max())
    m = mt.HungarianMatcher(per_image_rows=rows_from < 0)
    idx = m(_outputs(det, C), _targets(labels, gt, offs, "cuda"))
    ref = mo.assign(want)
    for b in range(B):
        i, j = idx[b][0].numpy(), idx[b][1].numpy()
        n = offs[b + 1] - offs[b]
        assert len(i) == len(j) == min(Q, n)                                   # a complete matching of the smaller side
        assert len(set(i)) == len(i) and len(set(j)) == len(j) and (i < Q).all() and (j < n).all()
        if n:                                                                  # same optimum as the oracle's assignment
            assert abs(want[b][i, j].sum() - want[b][ref[b][0], ref[b][1]].sum()) < 1e-4


def _inputs(B, Q, C, counts):
    logits = synth.normal(17, f"mt2.logits.{B}.{Q}", (B, Q, C), 2.0).astype(np.float32)
    cxcy = 0.15 + 0.7 * synth.uniform01(17, f"mt2.cxcy.{B}.{Q}", (B, Q, 2))
    wh = 0.05 + 0.4 * synth.uniform01(17, f"mt2.wh.{B}.{Q}", (B, Q, 2))
    det = np.concatenate([logits, cxcy, wh], axis=-1).astype(np.float32)
    G = int(np.sum(counts))
    labels = (synth.uniform01(17, f"mt2.labels.{G}", (G,)) * C).astype(np.int64).clip(0, C - 1)
    gt = np.concatenate([0.15 + 0.7 * synth.uniform01(17, f"mt2.g1.{G}", (G, 2)), 0.05 + 0.4 * synth.uniform01(17, f"mt2.g2.{G}", (G, 2))],
                        axis=-1).astype(np.float32)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return det, labels, gt, offs


def test_no_targets_and_bad_labels(mt):
    det, labels, gt, offs = _inputs(3, 6, 4, np.array([0, 0, 0]))
    idx = mt.HungarianMatcher()(_outputs(det, 4), _targets(labels, gt, offs))
    assert all(len(i) == 0 and len(j) == 0 for i, j in idx)
    det, labels, gt, offs = _inputs(2, 6, 4, np.array([2, 1]))
    labels[1] = 9
    with pytest.raises(IndexError):
        mt.HungarianMatcher()(_outputs(det, 4), _targets(labels, gt, offs))


def test_build_matcher_signature(mt):
    args = types.SimpleNamespace(set_cost_class=1, set_cost_bbox=5, set_cost_giou=2, focal_alpha=0.25, focal_gamma=2.0)
    m = mt.build_matcher(args)
    assert (m.cost_class, m.cost_bbox, m.cost_giou, m.focal_alpha, m.focal_gamma) == (1, 5, 2, 0.25, 2.0)
