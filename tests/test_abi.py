"""CPU: the C-ABI library loads and exports every symbol include/segclip_hip.h declares; the module
mirror exposes the reference's state-dict keys; the product path refuses to run without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from segclip_amd import _lib, synth
from tests.helpers import FULL_FLAGS, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ensure_built():
    if not os.path.exists(_lib.lib_path()):
        import __graft_entry__ as ge
        ge.build()


def test_header_symbols_exported():
    _ensure_built()
    hdr = open(os.path.join(ROOT, "include", "segclip_hip.h")).read()
    names = set(re.findall(r"\b(segclip_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 30
    lib = ctypes.CDLL(_lib.lib_path())
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    assert set(_lib.SIGNATURES) == names, sorted(set(_lib.SIGNATURES) ^ names)


def test_library_loads_and_reports_version():
    _ensure_built()
    lib = _lib.load()
    assert lib.segclip_version() == 1
    assert lib.segclip_last_error_string() is not None


def test_struct_layout_matches_header():
    _ensure_built()
    lib = _lib.load()
    # a descriptor with a bf16 operand that has no unit stride must be rejected host-side (no launch)
    d = _lib.GemmDesc()
    d.M = d.N = d.K = 64
    d.A = d.B = d.C = ctypes.c_void_p(16)
    d.sam, d.sak, d.sbn, d.sbk = 3, 5, 64, 1
    d.a_dtype = d.b_dtype = d.c_dtype = _lib.BF16
    rc = lib.segclip_gemm(ctypes.byref(d), None)
    assert rc == -1
    assert b"unit stride" in lib.segclip_last_error_string()


def test_state_dict_keys_match_reference():
    g = load_golden("tiny_t18.npz")
    model, _ = synth.build_model(synth.SPECS["tiny"], FULL_FLAGS, device="cpu")
    ref = set(g["grad_names"].tolist()) | set(g["none_grad"].tolist())
    assert set(model.state_dict().keys()) == ref
    frozen = [n for n, p in model.named_parameters() if not p.requires_grad]
    assert frozen == ["vis_mae_decoder.decoder_pos_embed"]


def test_no_cpu_fallback():
    """The product path must fail loudly on CPU tensors instead of silently computing elsewhere."""
    _ensure_built()
    model, _ = synth.build_model(synth.SPECS["tiny"], {}, device="cpu")
    batch = synth.synthetic_batch(synth.SPECS["tiny"], 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"])


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "segclip_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle/", "").lower() or f == "synth.py" and "oracle" not in src, (dirpath, f)


def test_synthetic_generators_are_deterministic():
    a = synth.closed_form_tensor("clip.visual.proj", (8, 4))
    b = synth.closed_form_tensor("clip.visual.proj", (8, 4))
    assert torch.equal(a, b)
    b1 = synth.synthetic_batch(synth.SPECS["tiny"], 3, seed=1)
    b2 = synth.synthetic_batch(synth.SPECS["tiny"], 3, seed=1)
    assert all(torch.equal(b1[k], b2[k]) for k in b1)
    ids = b1["input_ids"][:, 0]
    assert (ids[:, 0] == synth.SPECS["tiny"]["vocab_size"] - 2).all()
    assert ((ids == synth.SPECS["tiny"]["vocab_size"] - 1).sum(-1) == 1).all()
