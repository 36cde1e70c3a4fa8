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
        """modules/util_module.py:90-147: name-based load, tolerant of missing / unexpected keys."""
        if prefix is not None:
            state_dict = {prefix + k: v for k, v in state_dict.items()}
        res = model.load_state_dict(state_dict, strict=False)
        log = print_logger or logger
        if prefix is None and (task_config is None or getattr(task_config, "local_rank", 0) == 0):
            if res.missing_keys:
                log.warning("Weights of {} not initialized from pretrained model: {}".format(
                    model.__class__.__name__, "\n   " + "\n   ".join(res.missing_keys)))
            if res.unexpected_keys:
                log.warning("Weights from pretrained model not used in {}: {}".format(
                    model.__class__.__name__, "\n   " + "\n   ".join(res.unexpected_keys)))
        return model

    @property
    def dtype(self):
        return next(self.parameters()).dtype


class CrossEn(nn.Module):
    """modules/util_module.py:216-226 (unused by the training forward; kept for API parity)."""

    def forward(self, sim_matrix):
        assert sim_matrix.size(0) == sim_matrix.size(1)
        return ops.CrossEntropyFn.apply(sim_matrix.float(), 0)
