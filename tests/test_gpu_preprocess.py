"""GPU parity of the device input pipeline (dod_preprocess): bit-exact against Pillow's BILINEAR resize + ToTensor
(what the reference's transform, train.py:584-587, computes) and against the numpy oracle, for ragged batches."""
import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as ppo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pre():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dinov2_od_amd import preprocess
    return preprocess


def _imgs(shapes, seed):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in shapes]


def _pillow(img, out_h, out_w):
    from PIL import Image
    r = np.array(Image.fromarray(img, "RGB").resize((out_w, out_h), Image.BILINEAR))
    return (np.transpose(r, (2, 0, 1)).astype(np.float32) / np.float32(255.0))


@pytest.mark.parametrize("size", [(224, 224), (518, 518)])
def test_ragged_batch_is_bit_exact_against_pillow(pre, size):
    shapes = [(480, 640), (427, 640), (640, 480), (333, 500), (224, 224), (100, 80), (518, 700), (225, 223), (1000, 37), (37, 1000)]
    imgs = _imgs(shapes, 1)
    imgs[3][:, ::2] = 255
    imgs[3][:, 1::2] = 0                     # hard edges: rounding / clipping
    got = pre.preprocess_batch(imgs, size).cpu().numpy()
    assert got.shape == (len(imgs), 3, size[0], size[1]) and got.dtype == np.float32
    for i, im in enumerate(imgs):
        want = _pillow(im, *size)
        assert np.array_equal(got[i].view(np.uint32), want.view(np.uint32)), shapes[i]
        assert np.array_equal(got[i], ppo.preprocess([im], *size)[0])


def test_coco_sized_batch_and_non_square_target(pre):
    """the bench batch's worth of COCO-sized images (64 x 480x640) and a non-square target"""
    imgs = _imgs([(480, 640)] * 8 + [(640, 427)] * 8, 2)
    got = pre.preprocess_batch(imgs, (224, 320)).cpu().numpy()
    for i in (0, 7, 8, 15):
        assert np.array_equal(got[i], _pillow(imgs[i], 224, 320))
    t = pre.ResizeToTensor((224, 224))(imgs[:2])
    assert t.is_cuda and t.shape == (2, 3, 224, 224) and float(t.min()) >= 0.0 and float(t.max()) <= 1.0


def test_accepts_torch_and_pil_inputs_and_rejects_bad_ones(pre):
    from PIL import Image
    im = _imgs([(50, 60)], 3)[0]
    a = pre.preprocess_batch([im, torch.from_numpy(im), Image.fromarray(im, "RGB")], (32, 32)).cpu().numpy()
    assert np.array_equal(a[0], a[1]) and np.array_equal(a[0], a[2])
    with pytest.raises(ValueError):
        pre.preprocess_batch([im.astype(np.float32)], (32, 32))
    with pytest.raises(ValueError):
        pre.preprocess_batch([im[:, :, :1]], (32, 32))
    with pytest.raises(ValueError):
        pre.preprocess_batch([np.zeros((4000, 4000, 3), np.uint8)], (32, 32))       # > 15x down-scaling: filter wider than PP_MAXK


def test_uint8_pipeline_feeds_the_fused_patch_embed_bit_exactly(pre):
    """f4 fused into K1: the resampled bytes (uint8 HWC, Pillow-exact) go straight into the patch-embedding kernel, which applies
    ToTensor's / 255 in its load stage -- same detections, bit for bit, as the fp32 CHW batch through the same model"""
    from PIL import Image
    from dinov2_od_amd import synth
    from tests import cases, gpu_util as G
    imgs = _imgs([(480, 640), (333, 500), (100, 80)], 5)
    u8 = pre.preprocess_batch(imgs, (224, 224), as_uint8=True)
    assert u8.dtype == torch.uint8 and u8.shape == (3, 224, 224, 3)
    for i, im in enumerate(imgs):
        assert np.array_equal(u8[i].cpu().numpy(), np.array(Image.fromarray(im, "RGB").resize((224, 224), Image.BILINEAR)))
    f32 = pre.preprocess_batch(imgs, (224, 224))
    bb, dc = cases.cfg1(25)
    for precision in ("bf16x3", "bf16"):
        m = G.make_detector(bb, dc, precision, "facebook/dinov2-small")
        a = m.forward_packed(f32).clone()
        b = m.forward_packed_u8(u8).clone()
        assert torch.equal(a, b), precision
    m = G.make_detector(bb, dc, "fp32", "facebook/dinov2-small")
    with pytest.raises(ValueError, match="fused patch embed"):
        m.forward_packed_u8(u8)                          # exact-fp32 mode keeps the explicit im2col + fp32 GEMM
