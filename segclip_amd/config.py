"""Run-time switches of the hot path.

Every switch has a PROCESS DEFAULT (set_compute_dtype(), set_cross_mode(), or plain assignment
`config.overlap_towers = False`) and may be overridden for the CURRENT THREAD by `config.scope(...)`;
a model carries its own overrides in `model.segclip_config` (a dict), which SegCLIP.forward applies
through scope() - so two models, or two threads, in one process can run in different modes.

compute_dtype : torch.float32  -> exact-f32 MFMA kernels (parity gate: 1e-3 on loss/logits)
                torch.bfloat16 -> bf16 MFMA kernels (throughput mode; fp32 residual stream/params)
cross_mode    : "t18"      torch-1.8 key-buffer reinterpretation of CrossAttentionBlock (what the
                           published recipe trained with; SURVEY.md finding 0.4)
                "intended" each sample attends to its own tokens
overlap_wgrad : weight-gradient GEMMs of a block on a second HIP stream; measured SLOWER (64.4 vs 59.5 ms/step:
                two 139-KiB-LDS GEMMs cannot share a CU and the interleaving delays the dgrad chain), kept off
overlap_towers: enqueue the text tower on a second HIP stream (concurrent with the vision tower)
text_after_blocks : with overlap_towers: the text tower is enqueued after this many vision blocks (host launch order;
                0 = before the vision tower).  The host needs ~3 ms to enqueue the text tower: queued first, the vision
                stream idles that long at the start of every step, and - autograd replays the recording order backwards -
                again ~6 ms at the end of the backward pass (tools/stream_gaps.py).  The hook also cuts the vision tower's first
                stack into two autograd nodes, i.e. it decides the weight-gradient groups: 3 + 7 blocks (7 x 108 tiles = 2.95 rounds
                of 256 CUs, no K split) measured 36.80 / 36.82 ms per step against 37.04 / 37.04 for 4 + 6 (round 5; 2: 36.97, 5: 37.12, 6: 36.98)
wgrad_group_blocks : bf16 mode, inside ResStackFn: the weight gradients of up to this many consecutive blocks run as ONE grouped
                launch (ops.WgradGroup / segclip_wgrad_group) with few K ranges instead of 4 launches per block with 7-28 K
                ranges each (64 MB of fp32 partial tiles per gradient); the partition of a stack into groups minimises the
                library's time model.  1 = one launch per gradient (the round-3 behaviour).  Env SEGCLIP_WGRAD_GROUP.
wgrad_group_blocks_dist : the same limit while GradSync bucket slots are active (world size > 1).  Default = the single-GPU
                value: on one rank through the N>1 path a cap of 3 blocks costs 0.8 ms per step (39.9 vs 39.1 ms), while the
                largest group (7 vision blocks, 200 MB of gradients) still has 3 blocks of backward (~6 ms) to hide its
                exchange behind; lower it (env SEGCLIP_WGRAD_GROUP_DIST) if a real multi-GPU run shows exposed exchanges.
trust_weight_shadows : False: every training forward re-casts all GEMM weights to bf16 (one multi-tensor launch);
                True: only weights whose autograd version changed (set per model by train.prep_optimizer when the
                fused optimizer maintains the bf16 copies itself)
attn_fp8      : bf16 mode only: the self-attention FORWARD of the residual blocks runs Q K^T and P V on the e4m3 MFMA
                (BASELINE configs[4]; per-token scales for Q/K, per-chunk scale for V); backward stays bf16
fuse_res_stack: consecutive residual blocks of a tower run as ONE autograd node (ops.ResStackFn) instead of one per block
bf16_resgrad  : bf16 mode, inside ResStackFn: the residual-stream gradient travels between the LayerNorm backwards as
                one bf16 tensor (10 instead of 16 bytes per element); False keeps it fp32
bf16_resid    : bf16 mode, inside ResStackFn: the residual STREAM itself is bf16 between the blocks of a stack (fp32 at
                the tower boundaries and in every LayerNorm statistic): the out_proj / c_proj epilogues and every
                LayerNorm pass move 2 bytes less per element.  Implies the bf16 gradient chain
aux_u8        : bf16 mode: the towers keep QuickGELU'(u) for the backward as ONE BYTE per element where the MLP GEMMs run
                on full 256 x 256 tiles (q = rint((act' + 0.125) * 204), absolute error <= 0.0025 - about bf16's relative
                error at the typical magnitude, unbounded RELATIVE error near act' = 0); False keeps it as bf16.  Round 6: the
                MAE decoders' erf-GELU likewise where the 256 x 256-tile GEMM takes the shape (else: the pre-activation as bf16)
fused_head    : training forward: max-token pooling + ln_post + projection on the pooled row only, and the contrastive head
                (L2-normalise, all-gather, logits, both cross entropies) as ONE autograd node (ops.ClipLossFn) - ~15 instead
                of ~90 launches between the last forward GEMM and the first backward GEMM; False: the op-by-op path
reduce_side   : inside ResStackFn's backward the blocks' trailing reductions (split-K combines of the weight gradients, the
                LayerNorm / bias column sums - HBM-bound, off the data-gradient chain) run on a side stream beside the next
                block's GEMMs and are joined once per stack (not with GradSync's bucket slots).  MEASURED SLOWER and off: 51.9 vs
                41.9 ms per step - with two more active streams the runtime maps the text tower's stream onto the main
                stream's hardware queue and the towers run one after the other (the effect DESIGN.md 6 "hardware queues"
                describes)
text_trim     : training fast path of the text tower (CLIP.encode_text_eot): run the causal tower on the first L tokens only,
                L = the batch's largest EOT position + 1.  The tower is causal and only the EOT row reaches the loss, so the
                positions behind a caption's EOT influence neither the loss nor any gradient (their rows of every weight
                gradient are sums of exact zeros): outputs and gradients are those of the full 77-token pass (fp32 summation
                order of the weight gradients aside).  With the synthetic captions of SURVEY 8(d) (5-29 body tokens) L = 31 of
                77.  OFF by default: bench.py times the reference's full context; the switch exists for real caption data
                (the published recipe itself trains with max_words 32).  L is read with one host synchronisation whenever the
                id tensor changes (cached by storage / version); `text_trim_hint` (int) avoids it for callers that know it
fold_param_grads : a parameter that two nodes of ONE backward pass produce gradients for (the vision tower runs on the clean
                and on the masked image when the MAE loss is on: reference modules/modeling.py:196,237-249) would have the two
                summed by the autograd engine, one `aten::add` launch per parameter (~150 per step).  With the switch on,
                ResStackFn hands autograd the FIRST gradient of a parameter, adds every later one of the same pass into that
                tensor (one multi-tensor launch per stack, segclip_multi_add_f32) and returns None for it.  Same sums, same
                order; off while GradSync bucket slots are active
pad_rows      : bf16 mode, inside ResStackFn: a stack whose token-row count B x T is not a multiple of 128 (the text tower's B x 77
                unless B is a multiple of 128; the reference recipe trains with 96 per GPU) runs on the row count rounded up to
                128: its GEMMs stay on the 256 x 256-tile kernel, keep the one-byte derivative and the grouped weight gradients.
                Pad rows start as zeros and carry zero gradients; results for the token rows are those of the unpadded stack.
                Applies from 6144 token rows on (below, the step is host-bound and the extra small launches cost more).  Measured
                per-GPU batch 96: 5252 -> 5859 pairs/s, 160: 5975 -> 6457 (same box)
f32_split     : exact-f32 mode only.  False (default): Linear layers on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32, an
                exact fmaf chain: 1/16 of the bf16 rate).  True: every large fp32 GEMM runs as ONE bf16 GEMM of three times the
                contraction length over the operands' (hi, lo) bf16 parts - A_hi B_hi + A_lo B_hi + A_hi B_lo, fp32 accumulation,
                ~2^-16 relative per product instead of 2^-24 - on the bf16 matrix pipe; assignment logits, contrastive logits and the
                attention core stay exact fp32.  Held to the same 1e-3 / index bounds as the exact mode by the parity tests.
c_exec        : bf16 mode: the forward launches of a residual block are enqueued by one C-ABI call (segclip_resblock_fwd) instead of
                launch by launch through Python - bit-identical results, less host time (the step is host-bound below ~100 samples
                per GPU).  Env SEGCLIP_C_EXEC=0 (with SEGCLIP_TUNING=1) for A/B.
noise         : None -> draw Gumbel / uniform noise from the device generator (training runs);
                noise_injection([...("gumbel"|"rand", tensor)...]) consumed in call order -> parity runs
                (thread-local).
"""
import contextlib
import sys
import threading
import types

import torch

_warned_env = set()


def tuning_env(name, default):
    """A/B switch read from the environment.  Honoured only with SEGCLIP_TUNING=1 (the same gate as the native library's
    kernel-selection variables, csrc/common.h): a production process cannot have its kernels re-routed by a stray variable,
    and one that tries is told so once per variable."""
    import os
    v = os.environ.get(name)
    if v is None:
        return default
    if os.environ.get("SEGCLIP_TUNING", "0") in ("", "0"):
        if name not in _warned_env:
            _warned_env.add(name)
            print(f"segclip_amd: {name}={v} ignored (tuning switches need SEGCLIP_TUNING=1)", file=sys.stderr)
        return default
    return v


_DEFAULTS = dict(compute_dtype=torch.float32, f32_split=False, c_exec=tuning_env("SEGCLIP_C_EXEC", "1") != "0", cross_mode="t18", overlap_wgrad=False, overlap_towers=True,
                 trust_weight_shadows=False, attn_fp8=False, fuse_res_stack=True, bf16_resgrad=True, bf16_resid=False, fused_head=True, reduce_side=False,
                 aux_u8=tuning_env("SEGCLIP_AUX_U8", "1") != "0",
                 text_after_blocks=3, text_trim=False, text_trim_hint=None, pad_rows=tuning_env("SEGCLIP_PAD_ROWS", "1") != "0", fold_param_grads=tuning_env("SEGCLIP_FOLD_GRADS", "1") != "0",
                 wgrad_group_blocks=int(tuning_env("SEGCLIP_WGRAD_GROUP", "12")), wgrad_group_blocks_dist=int(tuning_env("SEGCLIP_WGRAD_GROUP_DIST", "12")))
_tls = threading.local()


def _get(name):
    ov = getattr(_tls, "overrides", None)
    if ov:
        for d in reversed(ov):
            if name in d:
                return d[name]
    return _DEFAULTS[name]


def _check(name, value):
    if name == "compute_dtype" and value not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
    if name == "cross_mode" and value not in ("t18", "intended"):
        raise ValueError("cross_mode must be 't18' or 'intended'")
    return value


class _ConfigModule(types.ModuleType):
    """Module type whose switches are properties: reads resolve thread-local overrides first, writes set the
    process default."""


for _n in _DEFAULTS:
    setattr(_ConfigModule, _n, property(lambda self, _n=_n: _get(_n),
                                        lambda self, v, _n=_n: _DEFAULTS.__setitem__(_n, _check(_n, v))))
sys.modules[__name__].__class__ = _ConfigModule


def set_compute_dtype(dtype):
    _DEFAULTS["compute_dtype"] = _check("compute_dtype", dtype)


def set_cross_mode(mode):
    _DEFAULTS["cross_mode"] = _check("cross_mode", mode)


@contextlib.contextmanager
def scope(**overrides):
    """Thread-local overrides of the switches above for the duration of the block."""
    for k, v in overrides.items():
        if k not in _DEFAULTS:
            raise KeyError(f"unknown switch {k!r}")
        _check(k, v)
    stack = getattr(_tls, "overrides", None)
    if stack is None:
        stack = _tls.overrides = []
    stack.append(dict(overrides))
    try:
        yield
    finally:
        stack.pop()


def set_stack_hook(after_blocks, fn):
    """One-shot hook for the NEXT fused residual stack of this thread (SegViT._run_blocks): the stack is cut after
    `after_blocks` blocks and fn() runs between the two parts.  SegCLIP.forward uses it to enqueue the text tower (second
    stream) once the first vision blocks are queued, so that neither stream waits for the host (see modeling.py)."""
    _tls.stack_hook = (int(after_blocks), fn)


def take_stack_hook():
    h = getattr(_tls, "stack_hook", None)
    _tls.stack_hook = None
    return h


@contextlib.contextmanager
def noise_injection(items):
    """items: list of (kind, tensor) in the order the forward consumes them."""
    prev = getattr(_tls, "noise", None)
    _tls.noise = list(items)
    try:
        yield
    finally:
        _tls.noise = prev


def _take(kind, shape, device):
    noise = getattr(_tls, "noise", None)
    if noise is None:
        return None
    if not noise:
        raise RuntimeError("noise_injection: more noise draws than injected tensors")
    k, t = noise.pop(0)
    if k != kind or tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"noise_injection: expected {kind}{tuple(shape)}, got {k}{tuple(t.shape)}")
    return t.to(device=device, dtype=torch.float32)


def gumbel(shape, device):
    """Gumbel(0,1) sample (torch.distributions.gumbel.Gumbel.sample in modules/module_seg_vit.py:223-226)."""
    t = _take("gumbel", shape, device)
    if t is not None:
        return t
    u = torch.rand(shape, device=device, dtype=torch.float32)
    if u.is_cuda:       # clamp / log / neg / log / neg as one kernel, in place (the draw itself stays torch's generator)
        from . import _lib as L
        L.check(L.load().segclip_gumbel_from_uniform(L.ptr(u), L.ptr(u), u.numel(), L.stream()), "gumbel_from_uniform")
        return u
    tiny = torch.finfo(torch.float32).tiny
    u = u.clamp(min=tiny, max=1.0 - torch.finfo(torch.float32).eps)
    return -torch.log(-torch.log(u))


def rand(shape, device):
    """torch.rand(N, L) of random_masking (modules/module_clip_util.py:101)."""
    t = _take("rand", shape, device)
    if t is not None:
        return t
    return torch.rand(shape, device=device, dtype=torch.float32)
