"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/dinodet.h declares, argument validation returns the documented status codes, and the
Python mirrors expose the reference's state-dict keys.  No GPU compute."""
import ctypes as C
import os
import re

import pytest
import torch

from dinov2_od_amd import _native as nat
from dinov2_od_amd import synth
from dinov2_od_amd.engine import make_config
from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(nat.LIB_PATH):
        from dinov2_od_amd._build import build
        build(verbose=False)
    return nat.lib()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "dinodet.h")).read()
    declared = set(re.findall(r"\b(dod_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(nat.SYMBOLS), (declared ^ set(nat.SYMBOLS))
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.dod_version()


def test_release_library_has_no_tuning_code(lib):
    """The release library carries no tuning hooks: none of the dod_debug_* entry points of include/dinodet_tuning.h (in-kernel time
    stamps, MFMA probes) is exported, and the only DINODET_* environment variables it reads are the operational ones INTEGRATION.md
    lists (<= 8) -- no variable can make a shipped kernel skip work or pick another tile."""
    tun = open(os.path.join(ROOT, "include", "dinodet_tuning.h")).read()
    names = set(re.findall(r"\b(dod_debug_[a-z0-9_]+)\s*\(", tun))
    assert len(names) >= 5
    raw = open(nat.LIB_PATH, "rb").read()
    for n in names:
        assert not hasattr(lib, n), n
    assert b"dod_debug_" not in raw
    env = set(re.findall(rb"DINODET_[A-Z0-9_]+", raw))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert len(env) <= 8, env
    for v in env:
        assert v.decode() in doc, f"{v.decode()} is read by the library but not documented in INTEGRATION.md"
    for gone in (b"DINODET_GEMM_ABL", b"DINODET_DEBUG_NOOUT", b"DINODET_DEBUG_LDA0", b"DINODET_GEMM_TILE", b"DINODET_EPI_RB"):
        assert gone not in raw


def test_test_options_validate_names(lib):
    assert lib.dod_test_set_option(b"tailsplit", -1) == 0
    assert lib.dod_test_set_option(b"no_such_option", 1) == 1 and b"unknown" in lib.dod_last_error(None)
    assert lib.dod_test_counter(b"tail_splits") >= 0 and lib.dod_test_counter(b"nope") == -1


def test_config_struct_layout_matches_header():
    hdr = open(os.path.join(ROOT, "include", "dinodet.h")).read()
    body = re.search(r"typedef struct dod_config \{(.*?)\} dod_config;", hdr, re.S).group(1)
    fields = re.findall(r"^\s*(int32_t|float)\s+(\w+);", body, re.M)
    assert [f for _, f in fields] == [f for f, _ in nat.DodConfig._fields_]
    for (ct, _), (_, pt) in zip(fields, nat.DodConfig._fields_):
        assert (ct == "float") == (pt is C.c_float)
    assert C.sizeof(nat.DodConfig) == 4 * len(fields)


def test_dec_train_params_struct_matches_header():
    """struct dod_dec_train_params (native decoder training step): the ctypes mirror lists the header's pointers in order"""
    hdr = open(os.path.join(ROOT, "include", "dinodet.h")).read()
    body = re.search(r"typedef struct dod_dec_train_params \{(.*?)\} dod_dec_train_params;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"\*\s*(\w+)", body)
    assert names == nat.DEC_TRAIN_FIELDS and len(names) == 31
    assert C.sizeof(nat.DodDecTrainParams) == 31 * C.sizeof(C.c_void_p)


def test_decoder_train_entry_points_validate_arguments(lib):
    bb, dc = cases.cfg1(25)
    cfg = make_config(bb, dc, "fp32")
    assert lib.dod_decoder_train_tape_bytes(C.byref(cfg), 2, 257) > 0 and lib.dod_decoder_train_workspace_bytes(C.byref(cfg), 2, 257) > 0
    assert lib.dod_decoder_train_tape_bytes(C.byref(cfg), 0, 257) == 0
    dense = make_config(bb, cases.dec_cfg(False), "fp32")
    assert lib.dod_decoder_train_tape_bytes(C.byref(dense), 2, 257) == 0            # nn.TransformerDecoder branch: not native
    rc = lib.dod_decoder_train_forward(C.byref(cfg), None, None, 2, 257, 0.1, 1, None, None, 0, None, 0, None)
    assert rc == 1 and b"null" in lib.dod_decoder_train_last_error()


def test_create_validates_arguments(lib):
    bb, dc = cases.cfg1(25)
    h = C.c_void_p()
    cfg = make_config(bb, dc, "bf16")
    assert lib.dod_create(C.byref(cfg), C.byref(h)) == 0
    # not finalized -> state error, message available
    assert lib.dod_workspace_bytes(h, 0, 224, 224) == 0
    rc = lib.dod_forward(h, None, 1, 224, 224, None, None, 0, None)
    assert rc == 3 and b"finalize" in lib.dod_last_error(h)
    rc = lib.dod_finalize_weights(h, None)
    assert rc == 2 and b"weights" in lib.dod_last_error(h)      # nothing registered
    lib.dod_destroy(h)
    bad = make_config(bb, dc, "bf16")
    bad.heads = 5                                                # 384 % 5 != 0
    h2 = C.c_void_p()
    assert lib.dod_create(C.byref(bad), C.byref(h2)) == 1
    assert b"backbone dims" in lib.dod_last_error(None)
    bad = make_config(bb, dc, "bf16")
    bad.target_dim = 128                                         # != decoder hidden 256
    assert lib.dod_create(C.byref(bad), C.byref(h2)) == 1
    with pytest.raises(ValueError):
        make_config(bb, dc, "fp4")
    assert make_config(bb, dc, "fp8").precision == 2          # DOD_PREC_FP8


@pytest.mark.parametrize("deform", [True, False])
def test_state_dict_keys_match_reference_layout(deform):
    """Keys/shapes equal the synthetic state dict, which tests/golden/make_goldens.py loads
    into the REFERENCE modules with strict=True."""
    from dinov2_od_amd.models import DINOv2ObjectDetector
    m = DINOv2ObjectDetector(dino_model_name="facebook/dinov2-small", hidden_dim=256, nheads=4, num_queries=25,
                             num_decoder_layers=2, dim_feedforward=512, lora_r=1, use_deformable=deform,
                             pretrained=False)
    bb, dc = cases.cfg1(25)
    dc.use_deformable = deform
    ref = synth.detector_state_dict(bb, dc)
    sd = m.state_dict()
    assert set(sd) == set(ref)
    for k, v in ref.items():
        assert tuple(sd[k].shape) == v.shape, k
    # frozen backbone, trainable LoRA / projection / decoder (dinov2_backbone.py:40-51)
    trainable = {k for k, p in m.named_parameters() if p.requires_grad}
    assert all(("lora_" in k) or k.startswith("decoder.") or k.startswith("backbone.projection") for k in trainable)
    assert any("lora_A" in k for k in trainable)
    if deform:   # tied decoder layers alias one storage (deformable_attention.py:284)
        a = m.state_dict(keep_vars=True)
        assert a["decoder.decoder.layers.0.linear1.weight"].data_ptr() == a["decoder.decoder.layers.1.linear1.weight"].data_ptr()


def test_forward_without_gpu_fails_loudly():
    from dinov2_od_amd.models import DINOv2ObjectDetector
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = DINOv2ObjectDetector(dino_model_name="facebook/dinov2-small", hidden_dim=256, nheads=4, num_queries=5,
                             num_decoder_layers=1, dim_feedforward=64, lora_r=1, pretrained=False).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 224, 224))
    with pytest.raises(ValueError, match="channel dimension"):
        m(torch.zeros(1, 4, 224, 224))
    m.train()                      # train(): the autograd composite is GPU-only too
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 224, 224))


def test_engine_named_cache_follows_replaced_parameters():
    """_engine_named() caches the (key, tensor) list; replacing a Parameter object after the first call (setattr of a block,
    register_parameter, load_state_dict(assign=True) on a child) must be seen -- the engine's (data_ptr, _version) signature of the
    OLD tensors would still match.  Host logic only."""
    from dinov2_od_amd.models import DINOv2ObjectDetector
    m = DINOv2ObjectDetector(dino_model_name="facebook/dinov2-small", hidden_dim=64, num_queries=5, num_decoder_layers=1,
                             dim_feedforward=64, lora_r=1, nheads=4, pretrained=False)
    a = dict(m._engine_named())
    assert m._engine_named() is m._engine_named()                      # cached
    key = "decoder.class_embed.weight"
    new = torch.nn.Parameter(torch.zeros_like(m.decoder.class_embed.weight))
    m.decoder.class_embed.weight = new                                  # no hook of the mixin sees this
    b = dict(m._engine_named())
    assert b[key] is new and a[key] is not new
    sd = {k: v.clone() for k, v in m.decoder.state_dict().items()}
    m.decoder.load_state_dict(sd, assign=True)                          # child-level assign: every tensor object replaced
    c = dict(m._engine_named())
    assert all(c["decoder." + k] is v for k, v in m.decoder.state_dict(keep_vars=True).items())


def test_abi_version_matches_header(lib):
    hdr = open(os.path.join(ROOT, "include", "dinodet.h")).read()
    assert int(re.search(r"#define DOD_ABI_VERSION (\d+)", hdr).group(1)) == lib.dod_abi_version() == nat.ABI_VERSION
