"""Run-time switches of the hot path (process-global, read at forward time).

compute_dtype : torch.float32  -> exact-f32 MFMA kernels (parity gate: 1e-3 on loss/logits)
                torch.bfloat16 -> bf16 MFMA kernels (throughput mode; fp32 residual stream/params)
cross_mode    : "t18"      torch-1.8 key-buffer reinterpretation of CrossAttentionBlock (what the
                           published recipe trained with; SURVEY.md finding 0.4)
                "intended" each sample attends to its own tokens
noise         : None -> draw Gumbel / uniform noise from the device generator (training runs);
                a list of ("gumbel"|"rand", tensor) consumed in call order -> parity runs.
"""
import contextlib

import torch

compute_dtype = torch.float32
cross_mode = "t18"
overlap_wgrad = False  # weight-gradient GEMMs of a block on a second HIP stream; measured SLOWER (64.4 vs 59.5 ms/step:
                       # two 139-KiB-LDS GEMMs cannot share a CU and the interleaving delays the dgrad chain), kept off
overlap_towers = True  # enqueue the text tower on a second HIP stream (concurrent with the vision tower)
trust_weight_shadows = False  # False: every training forward re-casts all GEMM weights to bf16 (one multi-tensor launch);
                              # True: only weights whose autograd version changed (set by train.prep_optimizer when the
                              # fused optimizer maintains the bf16 copies itself)
_noise = None


def set_compute_dtype(dtype):
    global compute_dtype
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
    compute_dtype = dtype


def set_cross_mode(mode):
    global cross_mode
    if mode not in ("t18", "intended"):
        raise ValueError("cross_mode must be 't18' or 'intended'")
    cross_mode = mode


@contextlib.contextmanager
def noise_injection(items):
    """items: list of (kind, tensor) in the order the forward consumes them."""
    global _noise
    prev = _noise
    _noise = list(items)
    try:
        yield
    finally:
        _noise = prev


def _take(kind, shape, device):
    if _noise is None:
        return None
    if not _noise:
        raise RuntimeError("noise_injection: more noise draws than injected tensors")
    k, t = _noise.pop(0)
    if k != kind or tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"noise_injection: expected {kind}{tuple(shape)}, got {k}{tuple(t.shape)}")
    return t.to(device=device, dtype=torch.float32)


def gumbel(shape, device):
    """Gumbel(0,1) sample (torch.distributions.gumbel.Gumbel.sample in modules/module_seg_vit.py:223-226)."""
    t = _take("gumbel", shape, device)
    if t is not None:
        return t
    u = torch.rand(shape, device=device, dtype=torch.float32)
    tiny = torch.finfo(torch.float32).tiny
    u = u.clamp(min=tiny, max=1.0 - torch.finfo(torch.float32).eps)
    return -torch.log(-torch.log(u))


def rand(shape, device):
    """torch.rand(N, L) of random_masking (modules/module_clip_util.py:101)."""
    t = _take("rand", shape, device)
    if t is not None:
        return t
    return torch.rand(shape, device=device, dtype=torch.float32)
