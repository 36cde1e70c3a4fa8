"""TEST INFRASTRUCTURE ONLY - runs the *actual* reference (ArrowLuo/SegCLIP, /root/reference) on CPU.

This file never travels into the product path.  It exists in the build container (where
/root/reference is mounted) to (i) validate oracle/segclip_oracle.py against the real reference
and (ii) generate the golden vectors committed under tests/golden/ (tests/golden/make_golden.py).
Nothing here is importable on the GPU box (no /root/reference there) and nothing in segclip_amd/
imports it.

Shims - the reference tree itself is NOT modified:
  1. np.float / np.long aliases (modules/module_mae.py:97 uses np.float).
  2. stub modules boto3 / botocore / requests / diffdist (modules/file_utils.py:20-22,
     modules/util_module.py:24).  diffdist.functional.all_gather is restated as
     all_gather fwd + sum-reduce/own-slice bwd (third-party, unpinned -> "parity unpinned").
  3. gloo process group.
  4. CLIP.get_config -> synthetic CLIP state-dict (no network for ViT-B-16.pt).
  5. CrossAttentionBlock.forward -> torch-1.8-equivalent key layout ("t18") or the
     "intended" permute (modules/module_seg_vit.py:213-218; finding 0.4 of SURVEY.md).
  6. util.get_logger(<tmp>/log.txt) before model construction.
"""
import argparse
import os
import sys
import tempfile
import types

import numpy as np

REF_ROOT = "/root/reference"


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "modules"))


def _stub(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


_IMPORTED = {}


def import_reference(cross_mode="t18"):
    """Import the reference modules with the shims above. Returns a namespace."""
    import torch
    import torch.distributed as dist

    if _IMPORTED:
        _IMPORTED["set_cross_mode"](cross_mode)
        return _IMPORTED["ns"]
    sys.dont_write_bytecode = True
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "long"):
        np.long = np.int64
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    _stub("boto3")
    _stub("botocore")
    _stub("botocore.exceptions", ClientError=type("ClientError", (Exception,), {}))
    try:
        import requests  # noqa: F401
    except Exception:
        _stub("requests")

    class _AG(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            outs = [torch.empty_like(x) for _ in range(dist.get_world_size())]
            dist.all_gather(outs, x.contiguous())
            return tuple(outs)

        @staticmethod
        def backward(ctx, *g):
            gs = torch.stack(g).contiguous()
            dist.all_reduce(gs)
            return gs[dist.get_rank()]

    fn = _stub("diffdist.functional", all_gather=lambda out_list, x, **kw: list(_AG.apply(x)))
    _stub("diffdist", functional=fn)

    import util as ref_util
    tmp = tempfile.mkdtemp(prefix="segclip_oracle_")
    ref_util.get_logger(os.path.join(tmp, "log.txt"))
    import logging
    logging.getLogger("seg").setLevel(logging.ERROR)
    logging.getLogger().setLevel(logging.ERROR)

    from modules.modeling import SegCLIP
    from modules.module_clip import CLIP
    import modules.module_seg_vit as msv
    import modules.module_clip_util as mcu
    import modules.module_mae as mmae

    state = {"mode": cross_mode}

    def xfwd(self, q, k):
        B, S, D = k.shape
        q = q.permute(1, 0, 2)
        kk = self.ln_k(k)
        if state["mode"] == "t18":
            kk = kk.contiguous().view(S, B, D)
        else:
            kk = kk.permute(1, 0, 2)
        q = q + self.attn(self.ln_x(q), kk, kk, need_weights=False)[0]
        q = q + self.mlp(self.ln_2(q))
        return q.permute(1, 0, 2)

    msv.CrossAttentionBlock.forward = xfwd

    def set_cross_mode(m):
        assert m in ("t18", "intended")
        state["mode"] = m

    ns = types.SimpleNamespace(SegCLIP=SegCLIP, CLIP=CLIP, msv=msv, mcu=mcu, mmae=mmae, util=ref_util)
    _IMPORTED["ns"] = ns
    _IMPORTED["set_cross_mode"] = set_cross_mode
    return ns


def ensure_process_group(rank=0, world_size=1, port=29512):
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world_size)


def synthetic_clip_state_dict(spec):
    from segclip_amd.synth import synthetic_clip_state_dict as f
    return f(spec)


def build_reference_model(spec, flags, rank=0, world_size=1, cross_mode="t18"):
    """Instantiate the real modules.modeling.SegCLIP (modeling.py:26-75) from a synthetic state-dict."""
    import torch
    ns = import_reference(cross_mode)
    ensure_process_group(rank, world_size)
    ns.CLIP.get_config = staticmethod(lambda pretrained_clip_name="ViT-B/16": synthetic_clip_state_dict(spec))
    args = argparse.Namespace(local_rank=0, rank=rank, world_size=world_size,
                              pretrained_clip_name="ViT-B/16", first_stage_layer=10,
                              use_vision_mae_recon=flags.get("use_vision_mae_recon", False),
                              use_text_mae_recon=flags.get("use_text_mae_recon", False),
                              use_seglabel=flags.get("use_seglabel", False),
                              mae_vis_mask_ratio=0.75, max_words=spec["context_length"])
    model = ns.SegCLIP.from_pretrained(cache_dir=None, state_dict=None, task_config=args)
    return model.train(), args


class NoiseTap:
    """Capture / inject the RNG draws of the hot path so integer outputs can be compared bit-exactly:
    Gumbel(0,1).sample (modules/module_seg_vit.py:223-226) and torch.rand (modules/module_clip_util.py:101)."""

    def __init__(self, inject=None):
        self.inject = list(inject) if inject is not None else None
        self.captured = []

    def __enter__(self):
        import torch
        self._torch = torch
        self._g = torch.distributions.gumbel.Gumbel.sample
        self._r = torch.rand
        tap = self

        def gsample(dist_self, sample_shape=torch.Size()):
            if tap.inject is not None:
                kind, t = tap.inject.pop(0)
                assert kind == "gumbel" and tuple(t.shape) == tuple(sample_shape), (kind, t.shape, sample_shape)
                return t.clone()
            t = tap._g(dist_self, sample_shape)
            tap.captured.append(("gumbel", t.clone()))
            return t

        def rand(*size, **kw):
            if tap.inject is not None:
                kind, t = tap.inject.pop(0)
                assert kind == "rand", kind
                return t.clone()
            t = tap._r(*size, **kw)
            tap.captured.append(("rand", t.clone()))
            return t

        torch.distributions.gumbel.Gumbel.sample = gsample
        torch.rand = rand
        return self

    def __exit__(self, *exc):
        self._torch.distributions.gumbel.Gumbel.sample = self._g
        self._torch.rand = self._r
        return False
