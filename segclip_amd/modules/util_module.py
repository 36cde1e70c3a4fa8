"""Mirror of modules/util_module.py (hot-path parts): dist_collect, config getters, weight init /
name-based weight loading."""
import logging

import torch
from torch import nn

from .. import ops

logger = logging.getLogger("seg")


def get_logger(filename=None):
    return logger


def dist_collect(x, args):
    """modules/util_module.py:180-190: rank-ordered differentiable all-gather, (B, ...) -> (B*W, ...).
    One fused RCCL all_gather_into_tensor forward, reduce_scatter(SUM) backward (diffdist semantics)."""
    return ops.all_gather_embeddings(x.contiguous())


def show_log(task_config, info):
    if task_config is None or getattr(task_config, "local_rank", 0) == 0:
        logger.info(info)


def update_attr(target_name, target_config, target_attr_name, source_config, source_attr_name, default_value=None):
    if hasattr(source_config, source_attr_name):
        if default_value is None or getattr(source_config, source_attr_name) != default_value:
            setattr(target_config, target_attr_name, getattr(source_config, source_attr_name))
    return target_config


def check_attr(target_name, task_config):
    return hasattr(task_config, target_name) and task_config.__dict__[target_name]


def get_attr(source_config, source_attr_name, default_value, donot_log=False):
    if hasattr(source_config, source_attr_name):
        if donot_log is False:
            show_log(source_config, "\t\t {}: {}".format(source_attr_name, getattr(source_config, source_attr_name)))
        return getattr(source_config, source_attr_name)
    return default_value


class PreTrainedModel(nn.Module):
    """modules/util_module.py:55-147."""

    def __init__(self, config=None, *inputs, **kwargs):
        super().__init__()
        self.config = config

    def init_weights(self, module):
        """modules/util_module.py:70-85: every nn.Linear / nn.Embedding weight ~ N(0, 0.02), Linear bias 0.
        (The LayerNorm branch there tests the file's own TF-style LayerNorm class, which the model never
        instantiates, so nn.LayerNorm parameters are left as constructed.)"""
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=0.02)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    @classmethod
    def init_preweight(cls, model, state_dict, prefix=None, task_config=None, print_logger=None):
        """modules/util_module.py:90-147: name-based load that NEVER raises on the checkpoint's contents.  Missing and
        unexpected keys are tolerated, and so are shape mismatches: the reference calls Module._load_from_state_dict per
        module with its own error list, logs "Weights from pretrained model cause errors ..." and continues - that is how its
        "reset ViT but keep Text Encoder" branch (modules/modeling.py:41-43) loads a ViT-B/32 archive into a ViT-B/16
        model: the mismatching tensors keep their freshly initialised values, everything else is taken over.
        (nn.Module.load_state_dict(strict=False) raises on a size mismatch, so it is not used here.)"""
        if prefix is not None:
            state_dict = {prefix + k: v for k, v in state_dict.items()}
        own = model.state_dict()                      # name -> tensor sharing storage with the parameters / buffers
        missing = [k for k in own if k not in state_dict]
        unexpected = [k for k in state_dict if k not in own]
        errors = []
        with torch.no_grad():
            for k, v in state_dict.items():
                if k not in own:
                    continue
                dst = own[k]
                if not torch.is_tensor(v):
                    errors.append('While copying the parameter named "{}", expected torch.Tensor but received {}'.format(k, type(v)))
                elif tuple(v.shape) != tuple(dst.shape):
                    errors.append("size mismatch for {}: copying a param with shape {} from checkpoint, the shape in current "
                                  "model is {}.".format(k, tuple(v.shape), tuple(dst.shape)))
                else:
                    dst.copy_(v)
        log = print_logger or logger
        if prefix is None and (task_config is None or getattr(task_config, "local_rank", 0) == 0):
            if missing:
                log.warning("Weights of {} not initialized from pretrained model: {}".format(
                    model.__class__.__name__, "\n   " + "\n   ".join(missing)))
            if unexpected:
                log.warning("Weights from pretrained model not used in {}: {}".format(
                    model.__class__.__name__, "\n   " + "\n   ".join(unexpected)))
            if errors:
                log.error("Weights from pretrained model cause errors in {}: {}".format(
                    model.__class__.__name__, "\n   " + "\n   ".join(errors)))
        model.last_load_report = {"missing_keys": missing, "unexpected_keys": unexpected, "error_msgs": errors}
        return model

    @property
    def dtype(self):
        return next(self.parameters()).dtype


class CrossEn(nn.Module):
    """modules/util_module.py:216-226 (unused by the training forward; kept for API parity)."""

    def forward(self, sim_matrix):
        assert sim_matrix.size(0) == sim_matrix.size(1)
        return ops.CrossEntropyFn.apply(sim_matrix.float(), 0)
