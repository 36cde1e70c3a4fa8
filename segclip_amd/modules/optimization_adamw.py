"""AdaptAdamW on MI355X: the reference's optimizer (modules/optimization_adamw.py:53-174) with its per-tensor
Python loop replaced by the fused multi-tensor HIP step (include/segclip_hip.h: segclip_adamw_step).

Same constructor, param_groups / state layout (`state[p] = {'step', 'exp_avg', 'exp_avg_sq'}`), `get_lr()` and
`step()` semantics: bias-corrected Adam, `denom = sqrt(v)/sqrt(1-b2^t) + e`, decoupled decay applied BEFORE the
update with the scheduled lr, schedule evaluated per parameter from `state['step']/t_total`, parameters whose
grad is None are skipped and do not advance.  `max_grad_norm` is accepted and (as in the reference) unused by
`step()`; clipping is the driver's job (segclip_amd/train.py).

Device-only: parameters must be fp32 HIP tensors; there is no CPU fallback.

Extras that keep the training loop free of host synchronisation (used by segclip_amd.train.TrainTail):
  step(loss=<device scalar>, ctrl=<TrainCtrl buffer>)   NaN-skip and gradient-clip coefficient are read on the
      device; state['step'] counts attempted steps and `effective_step()` subtracts the skipped ones lazily.
  shadow_bf16=True   the kernel also writes the bf16 rounding of every updated parameter (consumed by
      ops.wcast, so the next forward does not re-cast the weights).
"""
import ctypes as C
import math

import torch
from torch.optim import Optimizer
from torch.optim.optimizer import required

from .. import _lib as L


def warmup_cosine(x, warmup=0.002, lr_start=0., lr_end=0.):
    if x < warmup:
        return x * (1. - lr_start) / warmup + lr_start
    y = (x - warmup) / (1 - warmup)
    return lr_end + 0.5 * (1. - lr_end) * (1 + math.cos(math.pi * y))


def warmup_constant(x, warmup=0.002, lr_start=0., lr_end=0.):
    return x / warmup if x < warmup else 1.0


def warmup_linear(x, warmup=0.002, lr_start=0., lr_end=0.):
    return x / warmup if x < warmup else max((x - 1.) / (warmup - 1.), 0)


SCHEDULES = {'warmup_cosine': warmup_cosine, 'warmup_constant': warmup_constant, 'warmup_linear': warmup_linear}
_SCHEDULE_ID = {'warmup_cosine': 0, 'warmup_constant': 1, 'warmup_linear': 2}
_MAX_GROUPS = 16


class AdaptAdamW(Optimizer):
    def __init__(self, params, lr=required, warmup=-1, t_total=-1, schedule='warmup_linear',
                 b1=0.9, b2=0.999, e=1e-6, weight_decay=0.01, max_grad_norm=1.0, lr_start=0., lr_end=0.,
                 shadow_bf16=False):
        unit = "should be in [0.0, 1.0["
        checks = [(lr is required or lr >= 0.0, f"Invalid learning rate: {lr} - should be >= 0.0"),
                  (schedule in SCHEDULES, f"Invalid schedule parameter: {schedule}"),
                  (0.0 <= warmup < 1.0 or warmup == -1, f"Invalid warmup: {warmup} - {unit} or -1"),
                  (0.0 <= b1 < 1.0, f"Invalid b1 parameter: {b1} - {unit}"),
                  (0.0 <= b2 < 1.0, f"Invalid b2 parameter: {b2} - {unit}"),
                  (e >= 0.0, f"Invalid epsilon value: {e} - should be >= 0.0"),
                  (0.0 <= lr_start < 1.0, f"Invalid lr_start parameter: {lr_start} - {unit}"),
                  (0.0 <= lr_end < 1.0, f"Invalid lr_end parameter: {lr_end} - {unit}")]
        for ok, msg in checks:  # same ValueErrors as optimization_adamw.py:68-84
            if not ok:
                raise ValueError(msg)
        defaults = dict(lr=lr, schedule=schedule, warmup=warmup, t_total=t_total, b1=b1, b2=b2, e=e,
                        weight_decay=weight_decay, max_grad_norm=max_grad_norm, lr_start=lr_start, lr_end=lr_end)
        super().__init__(params, defaults)
        if len(self.param_groups) > _MAX_GROUPS:
            raise ValueError(f"AdaptAdamW (HIP): at most {_MAX_GROUPS} param groups")
        self.shadow_bf16 = bool(shadow_bf16)
        self._ctrl = None  # TrainCtrl device buffer of the last step(ctrl=...) call, for effective_step()

    # ---------------------------------------------------------------------------------------------
    def _nan_skips(self):
        if self._ctrl is None:
            return 0
        return int(self._ctrl.view(torch.int32)[2].item())  # host sync: only on get_lr()/state queries

    def effective_step(self, p):
        """The reference's state['step'] (iterations skipped because of a NaN loss do not count)."""
        st = self.state[p]
        return 0 if len(st) == 0 else max(st['step'] - self._nan_skips(), 0)

    def state_dict(self):
        """torch.optim format with the reference's meaning of state['step'] (NaN-skipped iterations excluded), so
        a `pytorch_opt.bin.N` written here resumes under either implementation (main_task_align.py:258-273)."""
        sd = super().state_dict()
        skips = self._nan_skips()
        if skips:
            sd = dict(sd, state={k: dict(v, step=max(v['step'] - skips, 0)) for k, v in sd['state'].items()})
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        skips = self._nan_skips()  # the device counter keeps running: re-bias the host-side count
        for st in self.state.values():
            if 'step' in st:
                st['step'] = int(st['step']) + skips

    @torch.no_grad()
    def get_lr(self, with_grad_only=True):
        """Scheduled lr of every parameter that currently has a gradient (the reference's rule); with
        with_grad_only=False, of every parameter that has optimizer state (usable after zero_grad(set_to_none))."""
        lr = []
        skips = None
        for group in self.param_groups:
            for p in group['params']:
                if with_grad_only and p.grad is None:
                    continue
                state = self.state[p]
                if len(state) == 0:
                    if with_grad_only:
                        return [0]
                    continue
                if skips is None:
                    skips = self._nan_skips()
                if group['t_total'] != -1:
                    fct = SCHEDULES[group['schedule']]
                    lr.append(group['lr'] * fct((state['step'] - skips) / group['t_total'], group['warmup'],
                                                group['lr_start'], group['lr_end']))
                else:
                    lr.append(group['lr'])
        return lr

    # ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None, loss=None, ctrl=None, zero_grads=False):
        """One optimisation step for every parameter that has a gradient.

        loss : optional device scalar; a NaN value turns the step into a no-op on the device.
        ctrl : optional TrainCtrl device buffer (torch.int32[8]) holding clip_coef / nan_skips.
        """
        ret = None
        if closure is not None:
            with torch.enable_grad():
                ret = closure()
        lib = L.load()
        if ctrl is not None and self._ctrl is not None and ctrl is not self._ctrl and ctrl.data_ptr() != self._ctrl.data_ptr():
            # a new control block (e.g. a fresh TrainTail per epoch) starts its NaN-skip counter at zero: fold the old
            # counter into the host-side step counts so that `step - nan_skips` keeps the reference's meaning
            old = self._nan_skips() - int(ctrl.view(torch.int32)[2].item())
            if old:
                for st in self.state.values():
                    if 'step' in st:
                        st['step'] = max(int(st['step']) - old, 0)
        tensors, keep = [], []
        groups = (L.AdamWGroup * len(self.param_groups))()
        for gi, group in enumerate(self.param_groups):
            g = groups[gi]
            g.lr, g.weight_decay, g.b1, g.b2, g.eps = group['lr'], group['weight_decay'], group['b1'], group['b2'], group['e']
            g.warmup, g.lr_start, g.lr_end = group['warmup'], group['lr_start'], group['lr_end']
            g.t_total, g.schedule = group['t_total'], _SCHEDULE_ID[group['schedule']]
            for p in group['params']:
                if p.grad is None:
                    continue
                grad = p.grad
                if grad.is_sparse:
                    raise RuntimeError('AdamW does not support sparse gradients')
                L.require_cuda(p, grad)
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("AdaptAdamW (HIP): parameters must be contiguous fp32 tensors")
                if grad.dtype != torch.float32 or not grad.is_contiguous():
                    grad = grad.float().contiguous()
                    keep.append(grad)
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                state['step'] += 1
                shadow = None
                if not (self.shadow_bf16 and p.dim() >= 2):
                    # the kernel rewrites the parameter through its raw pointer (no autograd version bump): a bf16
                    # copy that ops.wcast made earlier must not survive it
                    stale = getattr(p, "_segclip_shadow", None)
                    if stale is not None:
                        stale[1] = -1
                if self.shadow_bf16 and p.dim() >= 2:  # GEMM operands only
                    sh = getattr(p, "_segclip_shadow", None)
                    if sh is None:
                        sh = [torch.empty(p.shape, dtype=torch.bfloat16, device=p.device), p._version]
                        p._segclip_shadow = sh
                    sh[1] = p._version  # the kernel writes param and shadow together
                    shadow = sh[0]
                tensors.append((p, grad, state['exp_avg'], state['exp_avg_sq'], shadow, state['step'], gi))
        if not tensors:
            return ret
        arr = (L.AdamWTensor * len(tensors))()
        for i, (p, grad, m, v, shadow, step, gi) in enumerate(tensors):
            t = arr[i]
            t.param, t.grad, t.exp_avg, t.exp_avg_sq = p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr()
            t.shadow_bf16 = shadow.data_ptr() if shadow is not None else None
            t.n, t.step, t.group = p.numel(), step, gi
        if loss is not None:
            L.require_cuda(loss)
            if loss.dtype != torch.float32:
                raise RuntimeError("AdaptAdamW (HIP): loss must be an fp32 device scalar")
        if ctrl is not None:
            L.require_cuda(ctrl)
            self._ctrl = ctrl
        L.check(lib.segclip_adamw_step(C.cast(arr, C.c_void_p), len(tensors), C.cast(groups, C.c_void_p),
                                       len(self.param_groups), L.ptr(ctrl), L.ptr(loss.detach()) if loss is not None else None,
                                       1 if zero_grads else 0, L.stream()), "adamw_step")
        return ret
