"""Host-side: the run-time switches of segclip_amd/config.py - defaults the docs quote, thread-local scope(), validation."""
import threading

import pytest
import torch

from segclip_amd import config


def test_defaults_the_docs_quote():
    # numerics-changing switches are opt-in; fusions that change no arithmetic are on
    assert config.bf16_resid is False and config.reduce_side is False and config.overlap_wgrad is False
    assert config.fused_head is True and config.fuse_res_stack is True and config.bf16_resgrad is True
    assert config.cross_mode == "t18" and config.compute_dtype in (torch.float32, torch.bfloat16)
    assert isinstance(config.aux_u8, bool)


def test_scope_is_thread_local_and_nests():
    seen = {}
    with config.scope(bf16_resid=True, fused_head=False):
        assert config.bf16_resid is True and config.fused_head is False
        with config.scope(bf16_resid=False):
            assert config.bf16_resid is False and config.fused_head is False
        t = threading.Thread(target=lambda: seen.update(other=config.bf16_resid))
        t.start(); t.join()
        assert config.bf16_resid is True
    assert seen["other"] is False            # another thread sees the process default
    assert config.bf16_resid is False and config.fused_head is True


def test_unknown_switch_and_bad_values_raise():
    with pytest.raises(KeyError):
        with config.scope(no_such_switch=1):
            pass
    with pytest.raises(ValueError):
        with config.scope(cross_mode="torch19"):
            pass
    with pytest.raises(ValueError):
        config.set_compute_dtype(torch.float16)
