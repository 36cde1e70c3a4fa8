"""Host-side operator layer: raw launches of the C-ABI kernels (p_* functions, no autograd) and the
torch.autograd.Function wrappers the module mirror (segclip_amd/modules) is built from.

torch is used for device memory (caching allocator), streams, autograd bookkeeping and
torch.distributed - every FLOP-carrying op of the hot path is a call into libsegclip_hip.so.
There is no CPU fallback: tensors must live on the GPU.

Activation dtype == compute mode: torch.float32 -> exact-f32 MFMA path, torch.bfloat16 -> bf16 MFMA
path (fp32 residual stream, fp32 parameters / parameter gradients in both modes).
"""
import ctypes as C
import math
import os
import weakref

import torch
import torch.distributed as dist
from torch.autograd import Function

from . import _lib as L
from .config import tuning_env as _tenv

ACT_NONE, ACT_QUICK_GELU, ACT_GELU_ERF = L.ACT_NONE, L.ACT_QUICK_GELU, L.ACT_GELU_ERF


class _GemmProfile:
    """Optional per-launch timing of the dominant (GEMM) kernel with HIP events on the launch stream (tools/gemm_breakdown.py,
    bench.py's fallback when rocprofv3 is not usable).  Off by default; adds two event records per launch when enabled."""
    enabled = False
    records = []  # (start_event, end_event, flops, is_bf16, algorithmic bytes)

    @classmethod
    def start(cls):
        cls.enabled, cls.records = True, []

    @classmethod
    def stop(cls):
        cls.enabled = False
        torch.cuda.synchronize()
        out = [(s.elapsed_time(e) * 1e-3, f, b, nb) for s, e, f, b, nb in cls.records]
        cls.records = []
        return out


class _OpCount:
    """Algorithmic work of one model pass, per kernel class (bench.py's roofline block divides it by the kernel time
    rocprofv3 measured for the same class): records (class, flops, bytes) per launch, no device work.  Off by default."""
    enabled = False
    records = []

    @classmethod
    def start(cls):
        cls.enabled, cls.records = True, []

    @classmethod
    def stop(cls):
        cls.enabled = False
        out, cls.records = cls.records, []
        return out

    @classmethod
    def add(cls, kind, flops, nbytes):
        cls.records.append((kind, float(flops), float(nbytes)))


class ReduceQueue:
    """Deferred trailing reductions of a group of launches (one residual block's backward): the split-K combines of the
    weight gradients and the row reductions behind the LayerNorm backwards / fused bias-gradient column sums are left in
    their workspaces and combined by ONE launch per kind (segclip_reduce_multi) when the group is flushed - 2 launches
    instead of ~8 per block.  The workspaces are kept alive until the flush has been enqueued."""

    def __init__(self):
        self.slabs, self.rows, self.keep = [], [], []

    def add_slabs(self, ws, out, n_slabs, width, scale):
        e = L.ReduceEntry()
        e.src, e.out0, e.rows, e.width, e.scale, e.out_dtype = L.ptr(ws), L.ptr(out), n_slabs, width, float(scale), L.dt(out)
        self.slabs.append(e)
        self.keep.extend((ws, out))

    def add_rows(self, part, rows, width, ld, outs, seg):
        e = L.ReduceEntry()
        e.src, e.rows, e.width, e.ld, e.seg = L.ptr(part), rows, width, ld, seg
        e.out0, e.out1, e.out2 = (L.ptr(o) for o in (tuple(outs) + (None, None))[:3])
        self.rows.append(e)
        self.keep.append(part)
        self.keep.extend(o for o in outs if o is not None)

    def flush(self, side=None):
        """side: a HIP stream the reductions are enqueued on instead of the current one (they are HBM-bound and feed nothing
        in the data-gradient chain: beside the next block's GEMMs they cost almost nothing); the caller joins it before
        anything reads the reduced gradients."""
        lib = L.load()
        if not self.slabs and not self.rows:
            return
        main = torch.cuda.current_stream() if side is not None else None
        if side is not None:
            side.wait_stream(main)               # the partials were produced on the current stream

        def launch():
            for kind, ents in ((L.REDUCE_SLABS, self.slabs), (L.REDUCE_ROWS, self.rows)):
                for i in range(0, len(ents), L.REDUCE_MAX):
                    chunk = ents[i:i + L.REDUCE_MAX]
                    arr = (L.ReduceEntry * len(chunk))(*chunk)
                    L.check(lib.segclip_reduce_multi(arr, len(chunk), kind, L.stream()), "reduce_multi")
        if side is None:
            launch()
        else:
            with torch.cuda.stream(side):
                launch()
            for t in self.keep:                  # workspaces / outputs stay allocated until the side stream has used them
                t.record_stream(side)
        self.slabs, self.rows, self.keep = [], [], []


class WgradGroup:
    """Weight gradients dw = dy^T x of several consecutive residual blocks, collected and run as ONE launch of the
    weight-gradient kernel (segclip_wgrad_group).  A block's four gradients have 9-36 output tiles each; one by one, each is
    cut into 7-28 K ranges to fill the 256 CUs and leaves 64 MB of fp32 partial tiles to combine; grouped, the chip is full
    with 1-4 K ranges (7 vision blocks: none).  The operands stay referenced until flush(); add() hands out the result
    tensor, which is valid once flush() has been enqueued."""

    def __init__(self):
        self.items = []

    @staticmethod
    def covers(dy, x, out):
        return (dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy.dim() == 2 and x.dim() == 2
                and dy.shape[0] == x.shape[0] and dy.shape[0] % 64 == 0 and dy.shape[1] % 256 == 0 and x.shape[1] % 256 == 0
                and dy.stride(1) == 1 and x.stride(1) == 1 and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0
                and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0
                and (out is None or (out.is_contiguous() and out.data_ptr() % 16 == 0)))

    def add(self, dy, x, out=None):
        if not self.covers(dy, x, out) or (self.items and self.items[0][0].shape[0] != dy.shape[0]) or len(self.items) >= 48:
            return p_wgrad(dy, x, out=out)
        dw = out if out is not None else _empty((dy.shape[1], x.shape[1]), torch.float32, dy)
        self.items.append((dy, x, dw))
        return dw

    def flush(self):
        if not self.items:
            return
        lib = L.load()
        items, self.items = self.items, []
        R = items[0][0].shape[0]
        arr = (L.WgradItem * len(items))()
        tiles = 0
        for a, (dy, x, dw) in zip(arr, items):
            a.dy, a.x, a.dw = L.ptr(dy), L.ptr(x), L.ptr(dw)
            a.M, a.N, a.ld_dy, a.ld_x, a.ld_dw = dy.shape[1], x.shape[1], dy.stride(0), x.stride(0), dw.stride(0)
            tiles += (dy.shape[1] // 256) * (x.shape[1] // 256)
        splits = lib.segclip_wgrad_group_splits(tiles, R // 64)
        nbytes = lib.segclip_wgrad_group_ws_bytes(arr, len(items), splits)
        ws = torch.empty(max(nbytes // 4, 4), dtype=torch.float32, device=items[0][0].device)
        if _OpCount.enabled:   # ONE launch: its flops / algorithmic bytes are the sums over the problems
            _OpCount.add("gemm_bf16", sum(2.0 * R * dy.shape[1] * x.shape[1] for dy, x, dw in items),
                         sum(2 * R * (dy.shape[1] + x.shape[1]) + 4 * dw.numel() for dy, x, dw in items))
        try:
            L.check(lib.segclip_wgrad_group(arr, len(items), R, splits, L.ptr(ws), nbytes, L.stream()), "wgrad_group")
        except L.Unsupported:
            for dy, x, dw in items:
                p_wgrad(dy, x, out=dw)
            return
        if splits > 1:
            rq, off = ReduceQueue(), 0
            for dy, x, dw in items:
                n = dw.numel()
                rq.add_slabs(ws[off:off + splits * n], dw, splits, n, 1.0)
                off += splits * n
            rq.flush()


_PLAN_CACHE = {}


def wgrad_group_plan(nblk, tiles_blk, ksteps, gmax):
    """Sizes (in blocks, in backward order) of the weight-gradient groups of a stack of nblk equal blocks: the partition with
    the smallest modelled time (segclip_wgrad_group_model_us at the K-range count the library would choose)."""
    key = (int(nblk), int(tiles_blk), int(ksteps), int(gmax))
    if key not in _PLAN_CACHE:
        _PLAN_CACHE[key] = _wgrad_group_plan(*key)
    return list(_PLAN_CACHE[key])


def _wgrad_group_plan(nblk, tiles_blk, ksteps, gmax):
    lib = L.load()
    gmax = max(1, min(int(gmax), nblk, 48 // 4))
    cost = [0.0] * (gmax + 1)
    for n in range(1, gmax + 1):
        s = lib.segclip_wgrad_group_splits(tiles_blk * n, ksteps)
        cost[n] = lib.segclip_wgrad_group_model_us(tiles_blk * n, ksteps, s)
    best = [(0.0, [])] + [None] * nblk
    for k in range(1, nblk + 1):
        best[k] = min(((best[k - n][0] + cost[n], best[k - n][1] + [n]) for n in range(1, min(gmax, k) + 1)), key=lambda t: t[0])
    return best[nblk][1]


_PQ_TAIL_ENV = _tenv("SEGCLIP_PQ_TAIL", "0") not in ("", "0")
_SHARED_RQ = _tenv("SEGCLIP_SHARED_RQ", "1") != "0"   # A/B: one reduce queue per weight-gradient group (1) or per block (0)


def _empty(shape, dtype, like):
    return torch.empty(shape, dtype=dtype, device=like.device)


# Row pitch of the MLP hidden tensors (h, act'(u), du: M x 4D bf16).  A pitch that is a multiple of 2 KiB (4D = 3072 bf16 =
# 6144 B) makes the 256-row store burst of every GEMM tile camp on a few HBM channels: with 1 KiB more per row the c_fc
# forward with its two outputs runs 385 -> 312 us, the c_proj data gradient 324 -> 290 us, and the GEMMs that read these
# tensors as their A operand 2-5 % faster (M = 50176; tools/debug/gemm_ldc_probe2.py).  SEGCLIP_HIDDEN_PAD = extra
# elements per row (0 = dense).
_HIDDEN_PAD = int(_tenv("SEGCLIP_HIDDEN_PAD", "512"))


def _empty_pitched(shape, dtype, like):
    """(M, N) view of a buffer whose row pitch avoids multiples of 2 KiB (dense when the pitch is harmless or M is small)."""
    M, N = shape
    esz = torch.empty((), dtype=dtype).element_size()
    if _HIDDEN_PAD <= 0 or M < 4096 or (N * esz) % 2048 != 0:
        return _empty(shape, dtype, like)
    return torch.empty((M, N + _HIDDEN_PAD), dtype=dtype, device=like.device)[:, :N]


def _off(t, off):
    if t is None:
        return None
    L.require_cuda(t)
    return C.c_void_p(t.data_ptr() + off * t.element_size())


# ------------------------------------------------------------------------------------------------
# raw launches
# ------------------------------------------------------------------------------------------------
def p_gemm(A, B, Cc, M, N, K, sa, sb, ldc, *, a_off=0, b_off=0, c_off=0, bias=None, residual=None, ldr=0,
           r_off=0, aux=None, ldaux=0, act=ACT_NONE, mul_dact=False, alpha=1.0, nb1=1, nb2=1, bsA=(0, 0),
           bsB=(0, 0), bsC=(0, 0), bsR=None, colsum=None, aux_kind=0, defer=None, r_mod=0):
    """C(m,n) = epi(alpha * sum_k A(m,k) B(n,k)); sa = (sam, sak), sb = (sbn, sbk) element strides.
    r_mod > 0: the residual of output row m is residual row m % r_mod (L.Unsupported where the library has no such kernel)."""
    lib = L.load()
    L.require_cuda(A, B, Cc)
    if (A.dtype == torch.float32 and B.dtype == torch.float32 and Cc.dtype == torch.float32 and nb1 * nb2 == 1 and colsum is None
            and r_mod == 0 and M >= 256 and N >= 64 and K >= 64 and K % 8 == 0):
        from . import config as _cfg
        if _cfg.f32_split and _gemm_f32_split(A, B, Cc, M, N, K, sa, sb, ldc, a_off, b_off, c_off, bias, residual, ldr, r_off, aux,
                                              ldaux, act, mul_dact, alpha, aux_kind, defer):
            return Cc
    d = L.GemmDesc()
    d.A, d.B, d.C = _off(A, a_off), _off(B, b_off), _off(Cc, c_off)
    d.bias = L.ptr(bias)
    d.residual = _off(residual, r_off)
    d.aux = _off(aux, c_off) if aux is not None else None
    d.M, d.N, d.K = M, N, K
    d.sam, d.sak = sa
    d.sbn, d.sbk = sb
    d.ldc, d.ldr, d.ldaux = ldc, ldr, ldaux
    d.nb1, d.nb2 = nb1, nb2
    d.bsA1, d.bsA2 = bsA
    d.bsB1, d.bsB2 = bsB
    d.bsC1, d.bsC2 = bsC
    d.bsR1, d.bsR2 = bsC if bsR is None else bsR
    d.a_dtype, d.b_dtype, d.c_dtype = L.dt(A), L.dt(B), L.dt(Cc)
    d.r_dtype = L.dt(residual) if residual is not None else L.F32
    if aux is not None and aux.dtype != (torch.uint8 if aux_kind == 2 else Cc.dtype):
        raise TypeError("gemm: aux dtype must equal output dtype (uint8 for aux_kind 2)")
    d.act, d.mul_dact, d.alpha, d.aux_kind = act, int(mul_dact), float(alpha), int(aux_kind)
    flags = 0
    if colsum is not None:
        csws = torch.empty((max(M // 64, 1), N), dtype=torch.float32, device=A.device)
        d.colsum, d.colsum_ws = L.ptr(colsum), L.ptr(csws)
        if defer is not None:
            flags |= L.GEMM_DEFER_COLSUM
            defer.add_rows(csws, M // 64, N, N, (colsum,), N)
    ws = None
    # a workspace exists only for split-K (no epilogue operands) - or, with the tail-split experiment switched on, for any GEMM:
    # skip the query call on the ~60 % of launches that can never have one (host time: the step is ~800 launches)
    nbytes = (lib.segclip_gemm_ws_bytes(C.byref(d))
              if (_PQ_TAIL_ENV or (bias is None and residual is None and aux is None and act == ACT_NONE and not mul_dact)) else 0)
    if nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=A.device)
        d.ws, d.ws_bytes = L.ptr(ws), nbytes
        # split-K: the combine can be deferred when C is one contiguous (M, N) array
        if (defer is not None and nb1 * nb2 == 1 and ldc == N and c_off == 0 and Cc.is_contiguous()
                and (M * N) % 4 == 0 and Cc.data_ptr() % 16 == 0):
            ns = lib.segclip_gemm_splits(C.byref(d))
            if ns > 1:
                flags |= L.GEMM_DEFER_SPLITK
                defer.add_slabs(ws, Cc, ns, M * N, alpha)
    d.flags = flags
    d.res_row_mod = int(r_mod)
    if _OpCount.enabled:
        nz = nb1 * nb2
        _OpCount.add("gemm_bf16" if B.dtype == torch.bfloat16 else "gemm_f32", 2.0 * M * N * K * nz,
                     nz * (M * K * A.element_size() + N * K * B.element_size() + M * N * Cc.element_size()
                           + (M * N * residual.element_size() if residual is not None else 0)
                           + (M * N * aux.element_size() if aux is not None else 0)))
    if _GemmProfile.enabled:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.segclip_gemm(C.byref(d), L.stream()), "gemm")
        e1.record()
        nz = nb1 * nb2
        nbytes = nz * (M * K * A.element_size() + N * K * B.element_size() + M * N * Cc.element_size()
                       + (M * N * residual.element_size() if residual is not None else 0)
                       + (M * N * aux.element_size() if aux is not None else 0))   # operands read once + outputs written once
        _GemmProfile.records.append((e0, e1, 2.0 * M * N * K * nz, B.dtype == torch.bfloat16, nbytes))
        return Cc
    L.check(lib.segclip_gemm(C.byref(d), L.stream()), "gemm")
    return Cc


def _split3(t, off, rows, cols, ld, stack, role):
    """fp32 operand (rows x cols view at element offset `off`, row pitch ld) -> its (hi, lo) bf16 parts, three blocks along the
    contraction dimension (include/segclip_hip.h: segclip_split3_bf16)."""
    out = torch.empty((3 * rows, cols) if stack else (rows, 3 * cols), dtype=torch.bfloat16, device=t.device)
    L.check(L.load().segclip_split3_bf16(_off(t, off), L.ptr(out), rows, cols, ld, int(stack), int(role), L.stream()), "split3")
    return out


def _gemm_f32_split(A, B, Cc, M, N, K, sa, sb, ldc, a_off, b_off, c_off, bias, residual, ldr, r_off, aux, ldaux, act, mul_dact,
                    alpha, aux_kind, defer):
    """config.f32_split: C = epi(A B^T) for fp32 operands as ONE bf16 GEMM of contraction length 3 K over the operands' bf16 parts
    (A: hi | lo | hi, B: hi | hi | lo along k: A_hi B_hi + A_lo B_hi + A_hi B_lo, fp32 accumulators and epilogue).  Returns False
    when a layout or an epilogue has no bf16 kernel: the caller then takes the exact fp32 GEMM."""
    (sam, sak), (sbn, sbk) = sa, sb
    if not ((sak == 1 or sam == 1) and (sbk == 1 or sbn == 1)):
        return False
    if (sak == 1 and (sam % 4 or K % 4)) or (sak != 1 and (sak % 4 or M % 4)) or (sbk == 1 and (sbn % 4 or K % 4)) or (sbk != 1 and (sbk % 4 or N % 4)):
        return False
    if a_off % 4 or b_off % 4:
        return False
    if sak == 1:      # A stored (M, K): blocks side by side
        A3, sa3 = _split3(A, a_off, M, K, sam, False, 0), (3 * K, 1)
    else:             # A stored (K, M), k-strided: blocks stacked along k
        A3, sa3 = _split3(A, a_off, K, M, sak, True, 0), (1, M)
    if sbk == 1:
        B3, sb3 = _split3(B, b_off, N, K, sbn, False, 1), (3 * K, 1)
    else:
        B3, sb3 = _split3(B, b_off, K, N, sbk, True, 1), (1, N)
    try:
        p_gemm(A3, B3, Cc, M, N, 3 * K, sa3, sb3, ldc, c_off=c_off, bias=bias, residual=residual, ldr=ldr, r_off=r_off, aux=aux,
               ldaux=ldaux, act=act, mul_dact=mul_dact, alpha=alpha, aux_kind=aux_kind, defer=defer)
    except L.Unsupported:
        return False
    return True


def _ld(x):
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError(f"expected a 2-D row-major view, got shape {tuple(x.shape)} strides {x.stride()}")
    return x.stride(0)


def p_cast(t, dtype):
    if t.dtype == dtype:
        return t
    t = t.contiguous()
    out = torch.empty_like(t, dtype=dtype)
    L.check(L.load().segclip_cast(L.ptr(t), L.ptr(out), t.numel(), L.dt(t), L.dt(out), L.stream()), "cast")
    return out


def p_linear(x, w, bias=None, act=ACT_NONE, residual=None, want_aux=False, out_dtype=None, w_kn=False, aux_kind=0,
             pitched=False):
    """y = act(x w^T + bias) + residual.  x (M,K) view; w (N,K) [or (K,N) when w_kn]; same dtype family.
    want_aux: also return what the backward needs of the pre-activation u: u itself (aux_kind 0) or act'(u) (aux_kind 1)."""
    M, K = x.shape
    N = w.shape[1] if w_kn else w.shape[0]
    out_dtype = out_dtype or x.dtype
    alloc = _empty_pitched if pitched else _empty
    y = alloc((M, N), out_dtype, x)
    aux = alloc((M, N), torch.uint8 if aux_kind == 2 else out_dtype, x) if (want_aux and act != ACT_NONE) else None
    sb = (1, w.stride(0)) if w_kn else (w.stride(0), 1)
    try:
        p_gemm(x, w, y, M, N, K, (_ld(x), 1), sb, _ld(y), bias=bias, residual=residual,
               ldr=_ld(residual) if residual is not None else 0, aux=aux, ldaux=_ld(aux) if aux is not None else N, act=act,
               aux_kind=aux_kind)
    except L.Unsupported:
        if aux_kind != 2 or aux is None:
            raise
        # the one-byte derivative needs full 256 x 256 tiles and 16-byte aligned operands (include/segclip_hip.h): keep it
        # as bf16 instead (QuickGELU: aux_kind 1; erf-GELU: the pre-activation itself, aux_kind 0); the backward reads the
        # form off the tensor's dtype
        aux = alloc((M, N), out_dtype, x)
        p_gemm(x, w, y, M, N, K, (_ld(x), 1), sb, _ld(y), bias=bias, residual=residual,
               ldr=_ld(residual) if residual is not None else 0, aux=aux, ldaux=_ld(aux), act=act,
               aux_kind=1 if act == ACT_QUICK_GELU else 0)
    return y, aux


def fused_colsum_ok(M, N, K, dtype):
    """Shapes for which the bf16 LDS-DMA GEMM can emit the column sums of its output in the epilogue."""
    return dtype == torch.bfloat16 and M % 128 == 0 and N % 256 == 0 and K % 64 == 0     # (128: a last row of half-tiles)


def p_dgrad(dy, w, out_dtype, aux=None, act=ACT_NONE, w_kn=False, want_colsum=False, colsum_out=None, aux_kind=0, defer=None,
            pitched=False):
    """dx = dy w   [* act'(aux)]  ;  dy (M,N), w (N,K) [or (K,N) when w_kn] -> (M,K)
    want_colsum: also return the column sums of dx (= bias gradient of the Linear that produced the
    pre-activation), fused into the GEMM epilogue when the shape allows, else by the colsum kernel."""
    M, N = dy.shape
    K = w.shape[0] if w_kn else w.shape[1]
    dx = (_empty_pitched if pitched else _empty)((M, K), out_dtype, dy)
    sb = (w.stride(0), 1) if w_kn else (1, w.stride(0))
    if aux is not None and aux.dtype != (torch.uint8 if aux_kind == 2 else out_dtype):
        raise TypeError("dgrad: aux dtype must equal output dtype (uint8 for aux_kind 2)")
    cs = None
    if want_colsum and fused_colsum_ok(M, K, N, out_dtype) and dy.dtype == torch.bfloat16:
        cs = colsum_out if colsum_out is not None else _empty((K,), torch.float32, dy)
    p_gemm(dy, w, dx, M, K, N, (_ld(dy), 1), sb, _ld(dx), aux=aux, ldaux=_ld(aux) if aux is not None else K, act=act,
           mul_dact=aux is not None, colsum=cs, aux_kind=aux_kind, defer=defer)
    if want_colsum:
        return dx, (cs if cs is not None else p_colsum(dx, out=colsum_out))
    return dx


def _slot_of(w):
    """GradSync slot of a Parameter (segclip_amd/dist.py), looked up at forward time."""
    return getattr(w, "_segclip_gslot", None)


def _slot_out(slot, shape):
    """Gradient output buffer inside the parameter's all-reduce bucket (zero-copy), or None."""
    if slot is None:
        return None
    out = slot.out_buffer()
    if out is not None and tuple(out.shape) != tuple(shape):
        out = out.view(shape) if out.numel() == math.prod(shape) else None
    return out


def p_wgrad(dy, x, w_kn=False, out=None, defer=None):
    """dw = dy^T x (N,K) fp32  [or x^T dy (K,N) when w_kn];  dy (M,N), x (M,K).  `out`: preallocated fp32 result."""
    M, N = dy.shape
    K = x.shape[1]
    if w_kn:
        if x.dtype == torch.float32 and dy.dtype != torch.float32:
            x = p_cast(x, dy.dtype)
        dw = out if out is not None else _empty((K, N), torch.float32, dy)
        if dy.dtype == torch.float32 and x.dtype != torch.float32:
            dy = p_cast(dy, x.dtype)
        p_gemm(x, dy, dw, K, N, M, (1, _ld(x)), (1, _ld(dy)), N, defer=defer)
        return dw
    if x.dtype == torch.float32 and dy.dtype != torch.float32:
        x = p_cast(x, dy.dtype)  # only the A operand may be fp32 on the bf16 path
    dw = out if out is not None else _empty((N, K), torch.float32, dy)
    if (dy.dtype == torch.float32 and x.dtype == torch.float32 and N * K <= 1024 and M >= 4096 and M % 64 == 0
            and dw.is_contiguous() and defer is None):   # (the batched form has nothing to defer: a caller that wants to takes the plain path)
        # a tiny weight (the 8 x 8 Linear over the center axis of the MAE branch's ReconstructLayer, reference
        # modules/module_seg_vit.py:338-341) over many rows: ONE output tile, i.e. one workgroup walking all M rows (1.6 ms at
        # M = 12544) - instead M / 64 batched problems of 64 rows each and a column sum of their partial results
        S = M // 64
        part = _empty((S, N * K), torch.float32, dy)
        p_gemm(dy, x, part, N, K, 64, (1, _ld(dy)), (1, _ld(x)), K, nb1=S, bsA=(64 * _ld(dy), 0), bsB=(64 * _ld(x), 0),
               bsC=(N * K, 0))
        p_colsum(part, out=dw.view(N * K))
        return dw
    p_gemm(dy, x, dw, N, K, M, (1, _ld(dy)), (1, _ld(x)), K, defer=defer)
    return dw


def p_colsum(x, out=None):
    M, N = x.shape
    lib = L.load()
    if out is None:
        out = _empty((N,), torch.float32, x)
    ws = torch.empty(max(lib.segclip_colsum_ws_bytes(M, N), 4), dtype=torch.uint8, device=x.device)
    L.check(lib.segclip_colsum(L.ptr(x), L.ptr(out), L.ptr(ws), M, N, _ld(x), L.dt(x), L.stream()), "colsum")
    return out


def p_ln_fwd(x, w, b, eps, out_dtype, out=None, seg=None):
    """seg = (seg_in, seg_out, off) with out = a (n * seg_out, cols) buffer: row r of x is written to row
    (r // seg_in) * seg_out + off + r % seg_in of out (LayerNorm into a token slice of every sample)."""
    x = x.contiguous()
    rows, cols = x.shape
    y = out if out is not None else _empty((rows, cols), out_dtype, x)
    mean = _empty((rows,), torch.float32, x)
    rstd = _empty((rows,), torch.float32, x)
    if _OpCount.enabled:
        _OpCount.add("ln_fwd", 0, rows * cols * (x.element_size() + y.element_size()))
    if seg is None:
        L.check(L.load().segclip_layernorm_fwd(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(mean), L.ptr(rstd), rows,
                                               cols, eps, L.dt(x), L.dt(y), L.stream()), "layernorm_fwd")
    else:
        if out is None or not out.is_contiguous() or rows % seg[0] or out.numel() != rows // seg[0] * seg[1] * cols:
            raise ValueError("layernorm_fwd_seg: out must be the contiguous (rows / seg_in * seg_out, cols) buffer")
        L.check(L.load().segclip_layernorm_fwd_seg(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(mean), L.ptr(rstd), rows,
                                                   cols, eps, L.dt(x), L.dt(y), seg[0], seg[1], seg[2], L.stream()),
                "layernorm_fwd_seg")
    return y, mean, rstd


def p_ln_bwd(dy, x, w, mean, rstd, dres=None, dx_dtype=None, want_bf16=False, want_dres_colsum=False, outs=(None, None, None),
             defer=None, seg=None):
    """-> dx, dgamma, dbeta [, dx_bf16] [, colsum(dres)];  outs = preallocated (dgamma, dbeta, colsum) buffers or None.
    seg = (seg_in, seg_out, off): the rows of dy are mapped like the output of p_ln_fwd(seg=...)."""
    lib = L.load()
    dy = dy.contiguous()
    if seg is not None and (x.shape[0] % seg[0] or dy.numel() != x.shape[0] // seg[0] * seg[1] * x.shape[1]):
        raise ValueError("layernorm_bwd_seg: dy must be the contiguous (rows / seg_in * seg_out, cols) buffer")
    rows, cols = x.shape
    dx_dtype = dx_dtype or x.dtype
    dx = _empty((rows, cols), dx_dtype, x)
    dw = outs[0] if outs[0] is not None else _empty((cols,), torch.float32, x)
    db = outs[1] if outs[1] is not None else _empty((cols,), torch.float32, x)
    dx16 = _empty((rows, cols), torch.bfloat16, x) if want_bf16 else None
    dsum = None
    if want_dres_colsum and dres is not None:
        dsum = outs[2] if outs[2] is not None else _empty((cols,), torch.float32, x)
    if dres is not None:
        dres = dres.contiguous()
        if dres.dtype != dx_dtype:
            raise TypeError("layernorm_bwd: dres dtype must equal dx dtype")
    if _OpCount.enabled:
        _OpCount.add("ln_bwd", 0, rows * cols * (dy.element_size() + x.element_size() + dx.element_size() * (2 if dres is not None else 1)
                                                 + (2 if want_bf16 else 0)))
    wsb = lib.segclip_layernorm_bwd_ws_bytes(rows, cols)
    ws = torch.empty(max(wsb, 4), dtype=torch.uint8, device=x.device)
    deferred = defer is not None and rows > 0 and all(t.data_ptr() % 16 == 0 for t in (dw, db) + ((dsum,) if dsum is not None else ()))
    if seg is None:
        L.check(lib.segclip_layernorm_bwd(L.ptr(dy), L.ptr(x), L.ptr(w), L.ptr(mean), L.ptr(rstd), L.ptr(dres), L.ptr(dx),
                                          L.ptr(dx16), None if deferred else L.ptr(dw), L.ptr(db), L.ptr(dsum), L.ptr(ws), rows,
                                          cols, L.dt(dy), L.dt(x), L.dt(dx), L.stream()), "layernorm_bwd")
    else:
        L.check(lib.segclip_layernorm_bwd_seg(L.ptr(dy), L.ptr(x), L.ptr(w), L.ptr(mean), L.ptr(rstd), L.ptr(dres), L.ptr(dx),
                                              L.ptr(dx16), None if deferred else L.ptr(dw), L.ptr(db), L.ptr(dsum), L.ptr(ws),
                                              rows, cols, L.dt(dy), L.dt(x), L.dt(dx), seg[0], seg[1], seg[2], L.stream()),
                "layernorm_bwd_seg")
    if deferred:   # ws = [blocks][dgamma | dbeta | colsum(dres)] partial rows
        defer.add_rows(ws, wsb // (3 * cols * 4), (3 if dsum is not None else 2) * cols, 3 * cols, (dw, db, dsum), cols)
    out = [dx, dw, db]
    if want_bf16:
        out.append(dx16)
    if want_dres_colsum:
        out.append(dsum)
    return tuple(out)


def _attn_desc(q, k, v, o, B, H, Tq, Tk, hd, qs, ks, vs, os_, scale, causal, q_off=0, k_off=0, v_off=0, fp8=False,
               klen=None):
    d = L.AttnDesc()
    d.flags = L.ATTN_FP8 if (fp8 and klen is None) else 0
    if klen is not None:
        if klen.dtype != torch.int32 or klen.numel() != B or not klen.is_contiguous():
            raise TypeError("attention: klen must be a contiguous int32 tensor of B entries")
        L.require_cuda(klen)
        d.klen = L.ptr(klen)
    d.Q, d.K, d.V, d.O = _off(q, q_off), _off(k, k_off), _off(v, v_off), L.ptr(o)
    d.B, d.H, d.Tq, d.Tk, d.hd = B, H, Tq, Tk, hd
    d.q_sb, d.q_st = qs
    d.k_sb, d.k_st = ks
    d.v_sb, d.v_st = vs
    d.o_sb, d.o_st = os_
    d.scale, d.causal, d.dtype = float(scale), int(causal), L.dt(q)
    return d


def p_attn_fwd(d, like):
    lib = L.load()
    stats = torch.empty(max(lib.segclip_attn_stats_bytes(C.byref(d)) // 4, 1), dtype=torch.float32, device=like.device)
    d.stats = L.ptr(stats)
    if _OpCount.enabled:   # QK^T + PV (causal: half), Q K V read + O written
        fl = 4.0 * d.B * d.H * d.Tq * d.Tk * d.hd * (0.5 if d.causal else 1.0)
        _OpCount.add("attn_fwd", fl, 2.0 * d.B * d.H * d.hd * (2 * d.Tq + 2 * d.Tk))
    L.check(lib.segclip_attn_fwd(C.byref(d), L.stream()), "attn_fwd")
    return stats


def p_attn_bwd(d, stats, do, dq, dk, dv, dqs, dks, dvs, dos, dq_off=0, dk_off=0, dv_off=0, colsum_part=None):
    """colsum_part (bf16 path only): fp32 (B, 3*H*hd) buffer that receives the per-sample token sums of dQ|dK|dV."""
    lib = L.load()
    d.stats = L.ptr(stats)
    d.colsum_part = L.ptr(colsum_part)
    d.dO, d.dQ, d.dK, d.dV = L.ptr(do), _off(dq, dq_off), _off(dk, dk_off), _off(dv, dv_off)
    d.dq_sb, d.dq_st = dqs
    d.dk_sb, d.dk_st = dks
    d.dv_sb, d.dv_st = dvs
    d.do_sb, d.do_st = dos
    nbytes = lib.segclip_attn_bwd_ws_bytes(C.byref(d))
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=do.device)
    d.ws = L.ptr(ws)
    if _OpCount.enabled:   # S, dP, dV, dK, dQ = 2.5 x forward; Q K V O dO read + dQ dK dV written
        fl = 10.0 * d.B * d.H * d.Tq * d.Tk * d.hd * (0.5 if d.causal else 1.0)
        _OpCount.add("attn_bwd", fl, 2.0 * d.B * d.H * d.hd * (4 * d.Tq + 4 * d.Tk))
    L.check(lib.segclip_attn_bwd(C.byref(d), L.stream()), "attn_bwd")


def _wgrad_stream():
    from . import streams
    return streams.side_stream("wgrad")


def _reduce_stream():
    """side stream for a tower's trailing reductions (one per tower: keyed by the stream the tower's backward runs on)"""
    from . import streams
    return streams.side_stream("reduce:%x" % torch.cuda.current_stream().cuda_stream)


# config.overlap_wgrad experiments (read once): SEGCLIP_WGRAD_JOIN=stack joins the weight-gradient stream once per
# ResStackFn backward instead of once per block (the tensors it reads are kept alive until then); the stream's priority
# comes from SEGCLIP_WGRAD_PRIO (segclip_amd/streams.py)
_WGRAD_JOIN_STACK = _tenv("SEGCLIP_WGRAD_JOIN", "block") == "stack"


def interp_pos_table(table, h, w):
    """(1 + n*n, D) positional table -> (1 + h*w, D): class row kept, patch rows bicubically resampled
    (modules/module_clip_vtransformer.py:35-53, eval only: no gradient)."""
    n = int(round(math.sqrt(table.shape[0] - 1)))
    D = table.shape[1]
    src = table.detach().float().contiguous()
    out = torch.empty(1 + h * w, D, dtype=torch.float32, device=table.device)
    out[0] = src[0]
    L.check(L.load().segclip_interp_bicubic(_off(src, D), _off(out, D), n, h, w, D, L.stream()), "interp_bicubic")
    return out


_WCAST_PARAMS = {}  # id -> weakref of every Parameter that went through wcast (the GEMM weights)


def wcast(w, act_dtype):
    """Parameter (fp32 master) in the compute dtype of the current mode.

    bf16 copies ("shadows") live on the Parameter object as `_segclip_shadow = [tensor, version]` and are valid while
    the parameter's autograd version is unchanged.  They are written either by refresh_weight_shadows() (one
    multi-tensor cast at the start of a forward) or by the fused optimizer together with the parameter itself
    (modules/optimization_adamw.py, shadow_bf16=True).  A miss casts on the spot and registers the parameter."""
    if act_dtype == torch.bfloat16:
        sh = getattr(w, "_segclip_shadow", None)
        if sh is not None and sh[1] == w._version and sh[0].shape == w.shape:
            return sh[0]
        if isinstance(w, torch.nn.Parameter) and w.is_contiguous():
            buf = sh[0] if (sh is not None and sh[0].shape == w.shape) else torch.empty(w.shape, dtype=torch.bfloat16, device=w.device)
            p_cast_into(w.detach(), buf)
            w._segclip_shadow = [buf, w._version]
            _WCAST_PARAMS[id(w)] = weakref.ref(w)
            return buf
    return p_cast(w.detach(), act_dtype)


def p_cast_into(x, out):
    lib = L.load()
    L.check(lib.segclip_cast(L.ptr(x), L.ptr(out), x.numel(), L.dt(x), L.dt(out), L.stream()), "cast")
    return out


def refresh_weight_shadows(force=True):
    """Re-cast every registered GEMM weight in ceil(T/32) launches.  force=False trusts shadows whose version
    still matches (what the fused optimizer guarantees); force=True is the conservative per-forward behaviour."""
    todo = []
    for key, ref in list(_WCAST_PARAMS.items()):
        w = ref()
        if w is None:
            del _WCAST_PARAMS[key]
            continue
        sh = getattr(w, "_segclip_shadow", None)
        if sh is None or not w.is_cuda:
            continue
        if force or sh[1] != w._version:
            todo.append((w, sh))
    if not todo:
        return 0
    n = len(todo)
    src = (C.c_void_p * n)(*[w.data_ptr() for w, _ in todo])
    dst = (C.c_void_p * n)(*[sh[0].data_ptr() for _, sh in todo])
    cnt = (C.c_int64 * n)(*[w.numel() for w, _ in todo])
    L.check(L.load().segclip_multi_cast_bf16(C.cast(src, C.c_void_p), C.cast(dst, C.c_void_p), C.cast(cnt, C.c_void_p), n,
                                             L.stream()), "multi_cast_bf16")
    for w, sh in todo:
        sh[1] = w._version
    return n


# ------------------------------------------------------------------------------------------------
# autograd Functions
# ------------------------------------------------------------------------------------------------
class LayerNormFn(Function):
    """modules/module_clip_util.py:126-132 / nn.LayerNorm.  x (rows, cols)."""

    @staticmethod
    def forward(ctx, x, w, b, eps, out_dtype):
        x = x.contiguous()
        y, mean, rstd = p_ln_fwd(x, w, b, eps, out_dtype)
        ctx.save_for_backward(x, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        dx, dw, db = p_ln_bwd(dy, x, w, mean, rstd, None, x.dtype)[:3]
        return dx, dw, db, None, None


class LayerNormCatFn(Function):
    """LayerNorm(cat([a, b], dim=1)) for a (B, Ta, D), b (B, Tb, D) WITHOUT the concatenated copy: each part is
    normalised straight into its token slice of the (B, Ta+Tb, D) result, and the backward reads each part's
    gradient from the slice.  The learnable-center cross-attention normalises cat([centers, tokens]) as its K/V input
    in both of its layers (reference modules/module_seg_vit.py:294-296 `kv = torch.cat([q, inputs], dim=1)`, :211
    `self.ln_1(k)`): 2 x (B*204*768 fp32 copy forward + its slice-copy backward) per step."""

    @staticmethod
    def forward(ctx, a, b, w, bias, eps, out_dtype):
        B, Ta, D = a.shape
        Tb = b.shape[1]
        S = Ta + Tb
        a2, b2 = a.reshape(B * Ta, D), b.reshape(B * Tb, D)
        y = _empty((B * S, D), out_dtype, a)
        _, ma, ra = p_ln_fwd(a2, w, bias, eps, out_dtype, out=y, seg=(Ta, S, 0))
        _, mb, rb = p_ln_fwd(b2, w, bias, eps, out_dtype, out=y, seg=(Tb, S, Ta))
        ctx.save_for_backward(a2, b2, w, ma, ra, mb, rb)
        ctx.dims = (B, Ta, Tb, D)
        return y.view(B, S, D)

    @staticmethod
    def backward(ctx, dy):
        a2, b2, w, ma, ra, mb, rb = ctx.saved_tensors
        B, Ta, Tb, D = ctx.dims
        S = Ta + Tb
        dy = dy.contiguous().view(B * S, D)
        da, dwa, dba = p_ln_bwd(dy, a2, w, ma, ra, None, a2.dtype, seg=(Ta, S, 0))[:3]
        db_, dwb, dbb = p_ln_bwd(dy, b2, w, mb, rb, None, b2.dtype, seg=(Tb, S, Ta))[:3]
        dwa += dwb
        dba += dbb
        return da.view(B, Ta, D), db_.view(B, Tb, D), dwa, dba, None, None


def _ptr_array(tensors):
    L.require_cuda(*tensors)
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _seg_maps(segs):
    flat = []
    for sg in segs:
        flat += [0, 0, 0] if sg is None else [int(v) for v in sg]
    return (C.c_int64 * len(flat))(*flat)


class LayerNormMultiFn(Function):
    """y_k = LayerNorm(x; w_k, b_k), k < 3, from ONE read of x (rows, D) fp32 and one (mean, rstd); backward: one dx = sum
    of the three LayerNorm backwards.  segs[k] = None (y_k is (rows, D)) or (seg_in, seg_out, off): y_k is a
    (rows / seg_in, seg_out, D) buffer of which only the token slice [off, off + seg_in) of every sample is written - the
    rest is filled in place later (LayerNormIntoFn), and the matching rows of its gradient are ignored here.
    The learnable-center stage: `self.norm(inputs)` + `ln_1(cat([q, inputs]))` of both cross-attention layers
    (reference modules/module_seg_vit.py:289,294-296,211)."""

    @staticmethod
    def forward(ctx, x, eps, out_dtype, segs, *wb):
        rows, D = x.shape
        lib = L.load()
        ys = [_empty((rows, D) if sg is None else (rows // sg[0], sg[1], D), out_dtype, x) for sg in segs]
        mean = _empty((rows,), torch.float32, x)
        rstd = _empty((rows,), torch.float32, x)
        ws, bs = wb[0::2], wb[1::2]
        L.check(lib.segclip_layernorm_fwd_multi(L.ptr(x), len(segs), _ptr_array(ws), _ptr_array(bs), _ptr_array(ys),
                                                _seg_maps(segs), L.ptr(mean), L.ptr(rstd), rows, D, eps, L.dt(x), L.dt(ys[0]),
                                                L.stream()), "layernorm_fwd_multi")
        if _OpCount.enabled:
            _OpCount.add("ln_fwd", 0, rows * D * (x.element_size() + len(segs) * ys[0].element_size()))
        ctx.save_for_backward(x, mean, rstd, *ws)
        ctx.segs, ctx.out_dtype = segs, out_dtype
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        x, mean, rstd = ctx.saved_tensors[:3]
        ws = ctx.saved_tensors[3:]
        rows, D = x.shape
        n = len(ctx.segs)
        lib = L.load()
        dys = [dy.contiguous() if dy is not None else
               torch.zeros((rows, D) if sg is None else (rows // sg[0], sg[1], D), dtype=ctx.out_dtype, device=x.device)
               for dy, sg in zip(dys, ctx.segs)]
        dx = torch.empty_like(x)
        dgb = _empty((2 * n, D), torch.float32, x)
        wsb = torch.empty(max(lib.segclip_layernorm_bwd_multi_ws_bytes(rows, D, n), 4), dtype=torch.uint8, device=x.device)
        if _OpCount.enabled:
            _OpCount.add("ln_bwd", 0, rows * D * (2 * x.element_size() + n * dys[0].element_size()))
        L.check(lib.segclip_layernorm_bwd_multi(_ptr_array(dys), L.ptr(x), n, _ptr_array(ws), _seg_maps(ctx.segs), L.ptr(mean),
                                                L.ptr(rstd), L.ptr(dx), L.ptr(dgb), L.ptr(wsb), rows, D, L.dt(dys[0]), L.dt(x),
                                                L.stream()), "layernorm_bwd_multi")
        return (dx, None, None, None) + tuple(dgb[i] for i in range(2 * n))


def layer_norm_multi(x, affines, segs, eps, out_dtype):
    """-> tuple of outputs, or None when the library has no kernel for this combination (run the single LayerNorms)."""
    wb = [t for w, b in affines for t in (w, b)]
    try:
        return LayerNormMultiFn.apply(x.contiguous(), float(eps), out_dtype, tuple(segs), *wb)
    except L.Unsupported:
        return None


class LayerNormIntoFn(Function):
    """buf[:, off:off+Ta] = LayerNorm(a) IN PLACE, for a (B, Ta, D) and a (B, S, D) buffer whose other token rows are
    already final (LayerNormMultiFn).  The gradient of buf passes through unchanged: its consumer ignores these rows."""

    @staticmethod
    def forward(ctx, buf, a, w, bias, eps, off):
        B, S, D = buf.shape
        Ta = a.shape[1]
        a2 = a.reshape(B * Ta, D)
        _, m, r = p_ln_fwd(a2, w, bias, eps, buf.dtype, out=buf.view(B * S, D), seg=(Ta, S, off))
        ctx.mark_dirty(buf)
        ctx.save_for_backward(a2, w, m, r)
        ctx.dims = (B, S, Ta, D, off)
        return buf

    @staticmethod
    def backward(ctx, dbuf):
        a2, w, m, r = ctx.saved_tensors
        B, S, Ta, D, off = ctx.dims
        dbuf = dbuf.contiguous()
        da, dw, db = p_ln_bwd(dbuf.view(B * S, D), a2, w, m, r, None, a2.dtype, seg=(Ta, S, off))[:3]
        return dbuf, da.view(B, Ta, D), dw, db, None, None


def layer_norm_into(buf, a, w, bias, eps, off=0):
    return LayerNormIntoFn.apply(buf, a.contiguous(), w, bias, float(eps), int(off))


def layer_norm_cat(a, b, w, bias, eps=1e-5, out_dtype=None):
    return LayerNormCatFn.apply(a.contiguous(), b.contiguous(), w, bias, eps, out_dtype or a.dtype)


def layer_norm(x, w, b, eps=1e-5, out_dtype=None):
    shp = x.shape
    y = LayerNormFn.apply(x.reshape(-1, shp[-1]), w, b, eps, out_dtype or x.dtype)
    return y.view(shp)


class LinearFn(Function):
    """y = act(x w^T + b) + residual (nn.Linear / `@ proj`).  x (M,K); w (N,K) or (K,N) if w_kn."""

    @staticmethod
    def forward(ctx, x, w, b, residual, act, out_dtype, act_dtype, w_kn):
        if x.dtype != act_dtype:
            x = p_cast(x, act_dtype)
        wc = wcast(w, act_dtype)
        y, aux = p_linear(x, wc, b, act, residual, want_aux=True, out_dtype=out_dtype, w_kn=w_kn)
        ctx.save_for_backward(x, wc, aux)
        ctx.act, ctx.w_kn, ctx.has_b, ctx.has_r = act, w_kn, b is not None, residual is not None
        ctx.act_dtype = act_dtype
        ctx.gslot = _slot_of(w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wc, aux = ctx.saved_tensors
        dy = dy.contiguous()
        du = dy
        if ctx.act != ACT_NONE:
            du = torch.empty_like(dy)
            L.check(L.load().segclip_act_bwd(L.ptr(dy), L.ptr(aux), L.ptr(du), dy.numel(), ctx.act, L.dt(dy),
                                             L.stream()), "act_bwd")
        db = p_colsum(du) if (ctx.has_b and ctx.needs_input_grad[2]) else None
        if du.dtype == torch.float32 and x.dtype == torch.bfloat16:
            # an fp32 output gradient (fp32 residual stream) of a bf16 Linear: the GEMMs round it to bf16 while staging
            # anyway, but an fp32 operand keeps them off the LDS-DMA kernels - round once here (same values)
            du = p_cast(du, torch.bfloat16)
        dx = p_dgrad(du, wc, x.dtype, w_kn=ctx.w_kn) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            shp = (x.shape[1], du.shape[1]) if ctx.w_kn else (du.shape[1], x.shape[1])
            dw = p_wgrad(du, x, w_kn=ctx.w_kn, out=_slot_out(ctx.gslot, shp))
        dres = dy if (ctx.has_r and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dres, None, None, None, None


def linear(x, w, b=None, act=ACT_NONE, residual=None, out_dtype=None, act_dtype=None, w_kn=False):
    shp = x.shape
    act_dtype = act_dtype or x.dtype
    out_dtype = out_dtype or act_dtype
    N = w.shape[1] if w_kn else w.shape[0]
    r2 = residual.reshape(-1, N) if residual is not None else None
    y = LinearFn.apply(x.reshape(-1, shp[-1]), w, b, r2, act, out_dtype, act_dtype, w_kn)
    return y.view(*shp[:-1], N)


class BmmFn(Function):
    """C[b] = A[b] @ B[b]  (transB: A[b] @ B[b]^T).  3-D operands of one dtype with arbitrary (b, row)
    strides and unit stride on one of the two inner axes; used for the small contractions of the center
    stage (modules/module_seg_vit.py:304,309,342) and the contrastive logits (modules/modeling.py:356-357)."""

    @staticmethod
    def forward(ctx, A, B, transB, out_dtype):
        if A.dtype != B.dtype:
            raise TypeError(f"bmm: operand dtypes differ ({A.dtype} vs {B.dtype})")
        nb, M, K = A.shape
        N = B.shape[1] if transB else B.shape[2]
        Cc = _empty((nb, M, N), out_dtype, A)
        sb = (B.stride(1), B.stride(2)) if transB else (B.stride(2), B.stride(1))
        p_gemm(A, B, Cc, M, N, K, (A.stride(1), A.stride(2)), sb, N, nb1=nb, bsA=(A.stride(0), 0),
               bsB=(B.stride(0), 0), bsC=(M * N, 0))
        ctx.save_for_backward(A, B)
        ctx.transB = transB
        return Cc

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        nb, M, K = A.shape
        dC = p_cast(dC.contiguous(), A.dtype)
        N = dC.shape[2]
        dA = dB = None
        sn, sk = (B.stride(1), B.stride(2)) if ctx.transB else (B.stride(2), B.stride(1))  # Bop(n,k)
        if ctx.needs_input_grad[0]:
            # dA(m,k) = sum_n dC(m,n) Bop(n,k): contraction over n
            dA = _empty((nb, M, K), A.dtype, A)
            p_gemm(dC, B, dA, M, K, N, (N, 1), (sk, sn), K, nb1=nb, bsA=(M * N, 0), bsB=(B.stride(0), 0),
                   bsC=(M * K, 0))
        if ctx.needs_input_grad[1]:
            if ctx.transB:
                # dB(n,k) = sum_m dC(m,n) A(m,k)
                dB = _empty((nb, N, K), B.dtype, B)
                p_gemm(dC, A, dB, N, K, M, (1, N), (A.stride(2), A.stride(1)), K, nb1=nb, bsA=(M * N, 0),
                       bsB=(A.stride(0), 0), bsC=(N * K, 0))
            else:
                # dB(k,n) = sum_m A(m,k) dC(m,n)
                dB = _empty((nb, K, N), B.dtype, B)
                p_gemm(A, dC, dB, K, N, M, (A.stride(2), A.stride(1)), (1, N), N, nb1=nb, bsA=(A.stride(0), 0),
                       bsB=(M * N, 0), bsC=(K * N, 0))
        return dA, dB, None, None


def bmm(A, B, transB=False, out_dtype=None):
    return BmmFn.apply(A, B, transB, out_dtype or A.dtype)


class ActFn(Function):
    @staticmethod
    def forward(ctx, x, act):
        x = x.contiguous()
        y = torch.empty_like(x)
        L.check(L.load().segclip_act_fwd(L.ptr(x), L.ptr(y), x.numel(), act, L.dt(x), L.stream()), "act_fwd")
        ctx.save_for_backward(x)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        L.check(L.load().segclip_act_bwd(L.ptr(dy), L.ptr(x), L.ptr(dx), x.numel(), ctx.act, L.dt(x), L.stream()),
                "act_bwd")
        return dx, None


def _resblock_fwd_exec(x2, P, B, T, n_head, causal, act, eps, klen):
    """config.c_exec: the block's seven forward launches enqueued by ONE C-ABI call (segclip_resblock_fwd, csrc/exec.cpp) into
    buffers allocated here - the same kernels with the same descriptors as the launch-by-launch path below, so results are
    bit-identical; ~60 us of host time per block instead of ~300.  Returns None when the library has no kernel for a shape in
    the fused form (the caller then takes the launch-by-launch path)."""
    ln1w, ln1b, wqkv, bqkv, wo, bo, ln2w, ln2b, wfc, bfc, wpr, bpr = P
    M, D = x2.shape
    F4 = wfc.shape[0]
    bf = torch.bfloat16
    dev = x2.device
    wqkv_c, wo_c, wfc_c, wpr_c = (wcast(w, bf) for w in (wqkv, wo, wfc, wpr))
    ak = _aux_kind(bf, act, M, F4)
    y1 = torch.empty((M, D), dtype=bf, device=dev)
    y2 = torch.empty((M, D), dtype=bf, device=dev)
    o = torch.empty((M, D), dtype=bf, device=dev)
    qkv = torch.empty((M, 3 * D), dtype=bf, device=dev)
    st4 = torch.empty((4, M), dtype=torch.float32, device=dev)     # mean1 | rstd1 | mean2 | rstd2
    x1 = torch.empty((M, D), dtype=x2.dtype, device=dev)
    xo = torch.empty((M, D), dtype=x2.dtype, device=dev)
    h = _empty_pitched((M, F4), bf, x2)
    u = (_empty_pitched((M, F4), torch.uint8 if ak == 2 else bf, x2)) if act != ACT_NONE else None
    stats = torch.empty((B * n_head * T,), dtype=torch.float32, device=dev)
    d = L.ResBlockFwdDesc()
    d.x = x2.data_ptr()
    d.ln1w, d.ln1b, d.wqkv, d.bqkv = ln1w.data_ptr(), ln1b.data_ptr(), wqkv_c.data_ptr(), bqkv.data_ptr()
    d.wo, d.bo, d.ln2w, d.ln2b = wo_c.data_ptr(), bo.data_ptr(), ln2w.data_ptr(), ln2b.data_ptr()
    d.wfc, d.bfc, d.wpr, d.bpr = wfc_c.data_ptr(), bfc.data_ptr(), wpr_c.data_ptr(), bpr.data_ptr()
    d.y1, d.mean1, d.rstd1 = y1.data_ptr(), st4.data_ptr(), st4.data_ptr() + 4 * M
    d.qkv, d.o, d.stats, d.x1 = qkv.data_ptr(), o.data_ptr(), stats.data_ptr(), x1.data_ptr()
    d.y2, d.mean2, d.rstd2 = y2.data_ptr(), st4.data_ptr() + 8 * M, st4.data_ptr() + 12 * M
    d.h, d.ld_h = h.data_ptr(), h.stride(0)
    if u is not None:
        d.u, d.ld_u = u.data_ptr(), u.stride(0)
    d.xo = xo.data_ptr()
    if klen is not None:
        if klen.dtype != torch.int32 or klen.numel() != B or not klen.is_contiguous():
            raise TypeError("attention: klen must be a contiguous int32 tensor of B entries")
        d.klen = klen.data_ptr()
    d.M, d.B, d.T, d.D, d.F, d.H = M, B, T, D, F4, n_head
    d.eps, d.attn_scale = float(eps), 1.0 / math.sqrt(D // n_head)
    d.causal, d.act, d.aux_kind, d.x_dtype = int(causal), int(act), int(ak), L.dt(x2)
    try:
        L.check(L.load().segclip_resblock_fwd(C.byref(d), L.stream()), "resblock_fwd")
    except L.Unsupported:
        return None
    saved = (x2, ln1w, st4[0], st4[1], y1, wqkv_c, qkv, o, stats, wo_c, x1, ln2w, st4[2], st4[3], y2, wfc_c, u, h, wpr_c)
    return xo, saved


def _resblock_fwd(x2, P, B, T, n_head, causal, act, eps, act_dtype, klen):
    """One pre-LN residual block on the (B*T, D) residual stream x2 - fp32, or bf16 inside a ResStackFn running with
    config.bf16_resid - : returns (x_out (B*T, D) in the stream's dtype, tensors saved for backward)."""
    ln1w, ln1b, wqkv, bqkv, wo, bo, ln2w, ln2b, wfc, bfc, wpr, bpr = P
    M, D = x2.shape
    hd = D // n_head
    for w in P:
        sl = _slot_of(w)
        if sl is not None:
            sl.note_forward_use()
    from . import config as _cfg
    if (act_dtype == torch.bfloat16 and _cfg.c_exec and not _OpCount.enabled and not _GemmProfile.enabled and not _cfg.attn_fp8
            and x2.is_contiguous() and D % 8 == 0):
        r = _resblock_fwd_exec(x2, P, B, T, n_head, causal, act, eps, klen)
        if r is not None:
            return r
    y1, mean1, rstd1 = p_ln_fwd(x2, ln1w, ln1b, eps, act_dtype)
    wqkv_c, wo_c, wfc_c, wpr_c = (wcast(w, act_dtype) for w in (wqkv, wo, wfc, wpr))
    qkv, _ = p_linear(y1, wqkv_c, bqkv)
    o = _empty((M, D), act_dtype, x2)
    if M > B * T:                 # row-padded stack (ResStackFn): the attention kernel writes the B * T token rows only
        o[B * T:].zero_()
    from . import config as _cfg
    ad = _attn_desc(qkv, qkv, qkv, o, B, n_head, T, T, hd, (T * 3 * D, 3 * D), (T * 3 * D, 3 * D),
                    (T * 3 * D, 3 * D), (T * D, D), 1.0 / math.sqrt(hd), causal, 0, D, 2 * D,
                    fp8=bool(_cfg.attn_fp8) and act_dtype == torch.bfloat16, klen=klen)
    stats = p_attn_fwd(ad, x2)
    x1, _ = p_linear(o, wo_c, bo, residual=x2, out_dtype=x2.dtype)
    y2, mean2, rstd2 = p_ln_fwd(x1, ln2w, ln2b, eps, act_dtype)
    # bf16 mode: the c_fc epilogue stores act'(u) (its exponential is already there), the c_proj dgrad multiplies by it
    h, u = p_linear(y2, wfc_c, bfc, act=act, want_aux=True, aux_kind=_aux_kind(act_dtype, act, y2.shape[0], wfc_c.shape[0]),
                    pitched=act_dtype == torch.bfloat16)
    xo, _ = p_linear(h, wpr_c, bpr, residual=x1, out_dtype=x2.dtype)
    saved = (x2, ln1w, mean1, rstd1, y1, wqkv_c, qkv, o, stats, wo_c, x1, ln2w, mean2, rstd2, y2, wfc_c, u, h, wpr_c)
    return xo, saved


N_SAVED = 19


def _aux_kind(act_dtype, act, M=0, N=0):
    """What the residual blocks keep of the MLP pre-activation: act'(u) in bf16 mode with QuickGELU (the towers) - as one
    byte per element (aux_kind 2, config.aux_u8: half the bytes of the block's largest side tensor; absolute error
    <= 0.0025) when the GEMM runs on 256 x 256 tiles (rows a multiple of 128), else as bf16 -, u itself in the exact-f32 mode and for the
    erf-GELU of the MAE decoders."""
    if act_dtype == torch.bfloat16 and act == ACT_QUICK_GELU:
        from . import config as _cfg
        return 2 if (_cfg.aux_u8 and M > 0 and M % 128 == 0 and N % 256 == 0) else 1
    if act_dtype == torch.bfloat16 and act == ACT_GELU_ERF:
        # round 6: the MAE decoders' erf-GELU keeps the one-byte derivative as well where the 256 x 256-tile GEMM takes the shape
        # (c_fc forward 172 -> 9x us, the c_proj data gradient no longer re-evaluates erf per element: 247 -> 1xx us at M = 50432)
        from . import config as _cfg
        return 2 if (_cfg.aux_u8 and M > 0 and M % 128 == 0 and N % 256 == 0) else 0
    return 0


def _resblock_bwd(saved, cfg, klen, gslots, vslots, need, g, g16, chain, overlap_wgrad=False, keep=None, reduce_side=None,
                  wgroup=None, rqueue=None):
    """Hand-scheduled backward of one block.  g: fp32 (M, D) gradient of the block output or None; g16: its bf16 copy or
    None.  need[i]: gradient wanted for forward input i (0 = x, 1..12 = the parameters in forward order).
    chain=False: fp32 residual gradient in and out (plus the bf16 copy the LayerNorm backward emits for free).
    chain=True (bf16 mode inside ResStackFn): the residual gradient travels between the LayerNorm backwards as ONE bf16
    tensor (read 2 + written 2 bytes per element instead of 4 + 4 + 2): g may be None, returns dx fp32 = None.
    -> (dx fp32 or None, dx bf16 or None, the 12 parameter gradients)"""
    (x2, ln1w, mean1, rstd1, y1, wqkv_c, qkv, o, stats, wo_c, x1, ln2w, mean2, rstd2, y2, wfc_c, u, h, wpr_c) = saved
    B, T, D, n_head, causal, act, act_dtype = cfg
    M = x2.shape[0]               # B * T, or that rounded up to 128 rows in a row-padded stack (pad rows: zero gradients)
    hd = D // n_head
    bf = act_dtype == torch.bfloat16
    # bf16 mode: the residual-stream gradients feed the GEMMs as bf16 copies (all GEMMs then run on the LDS-DMA
    # kernels and read half the bytes); the copy of dx comes for free out of the LayerNorm backward.
    if bf and g16 is None:
        g16 = p_cast(g, act_dtype)
    if not bf:
        g16 = g
    res_in = g16 if chain else g
    rdt = act_dtype if chain else torch.float32
    # The weight gradients do not feed the data-gradient chain: optionally (config.overlap_wgrad) they are enqueued
    # on a second HIP stream.
    # (the switch is captured at FORWARD time: backward runs on the autograd thread, after config.scope() has exited)
    main = torch.cuda.current_stream()
    side = _wgrad_stream() if overlap_wgrad else None
    sq, so, sf, sp = gslots  # weight gradients land directly in their all-reduce bucket (segclip_amd/dist.py)
    F4 = wfc_c.shape[0]
    s_ln1w, s_ln1b, s_bqkv, s_bo, s_ln2w, s_ln2b, s_bfc, s_bpr = vslots   # ... and so do the 8 vector gradients

    def on_side(fn, *deps):
        if side is None:
            return fn()
        side.wait_stream(main)           # operands produced on the main stream are ready
        with torch.cuda.stream(side):
            out = fn()
        if out is not None:
            out.record_stream(main)
        return out

    # the block's trailing reductions (split-K combines, LayerNorm / bias column sums) are queued and flushed as two
    # launches at the end of the block (not when the weight gradients run on the side stream)
    # rqueue (ResStackFn with grouped weight gradients): the caller's queue - the trailing reductions of all blocks of a group
    # are flushed together (2 launches per 16 entries instead of 2 per block)
    own_rq = rqueue is None or side is not None
    rq = (ReduceQueue() if side is None else None) if own_rq else rqueue
    if side is not None:
        wgroup = None

    def wgrad(dy_, x_, out_):
        # wgroup (ResStackFn): the weight gradients of several blocks run later, as one grouped launch (WgradGroup)
        if wgroup is not None:
            return wgroup.add(dy_, x_, out_)
        return on_side(lambda: p_wgrad(dy_, x_, out=out_, defer=rq))
    # ---- MLP
    du, dbfc = p_dgrad(g16, wpr_c, act_dtype, aux=u, act=act, want_colsum=True,
                       colsum_out=_slot_out(s_bfc, (F4,)) if need[10] else None,
                       aux_kind=2 if u.dtype == torch.uint8 else _aux_kind(act_dtype, act),
                       defer=rq, pitched=bf)  # (dy c_proj)*act'(u), colsum
    dwpr = wgrad(g16, h, _slot_out(sp, (D, F4))) if need[11] else None
    dy2 = p_dgrad(du, wfc_c, act_dtype)
    dwfc = wgrad(du, y2, _slot_out(sf, (F4, D))) if need[9] else None
    two = bf and not chain   # fp32 dx + its bf16 copy
    r = p_ln_bwd(dy2, x1, ln2w, mean2, rstd2, dres=res_in, dx_dtype=rdt, want_bf16=two, want_dres_colsum=True,
                 outs=(_slot_out(s_ln2w, (D,)) if need[7] else None, _slot_out(s_ln2b, (D,)) if need[8] else None,
                       _slot_out(s_bpr, (D,)) if need[12] else None), defer=rq)
    dx1, dln2w, dln2b = r[0], r[1], r[2]
    dx1_16 = r[3] if two else dx1
    dbpr = r[-1] if need[12] else None                           # colsum(g), fused into the LN2 backward
    # ---- attention
    do = p_dgrad(dx1_16, wo_c, act_dtype)
    dwo = wgrad(dx1_16, o, _slot_out(so, (D, D))) if need[5] else None
    dqkv = _empty((M, 3 * D), act_dtype, x2)
    if M > B * T:
        dqkv[B * T:].zero_()      # the attention backward writes the token rows; a pad row must contribute exact zeros downstream
    s3 = (T * 3 * D, 3 * D)
    ad = _attn_desc(qkv, qkv, qkv, o, B, n_head, T, T, hd, s3, s3, s3, (T * D, D), 1.0 / math.sqrt(hd), causal,
                    0, D, 2 * D, klen=klen)
    part = _empty((B, 3 * D), torch.float32, x2) if (bf and need[4]) else None  # in_proj bias gradient per sample
    p_attn_bwd(ad, stats, do, dqkv, dqkv, dqkv, s3, s3, s3, (T * D, D), 0, D, 2 * D, colsum_part=part)
    dy1 = p_dgrad(dqkv, wqkv_c, act_dtype)
    dwqkv = wgrad(dqkv, y1, _slot_out(sq, (3 * D, D))) if need[3] else None
    dbqkv = None
    if need[4]:
        bq_out = _slot_out(s_bqkv, (3 * D,))
        if part is not None and rq is not None and (bq_out is None or bq_out.data_ptr() % 16 == 0):
            # in_proj bias gradient = token sums of dQ|dK|dV, left per sample by the attention backward: its B partial rows
            # join the block's ONE row-reduction launch (was: a column-sum launch + its reduction per block)
            dbqkv = bq_out if bq_out is not None else _empty((3 * D,), torch.float32, x2)
            rq.add_rows(part, B, 3 * D, 3 * D, (dbqkv,), 3 * D)
        else:
            dbqkv = on_side(lambda: p_colsum(part if part is not None else dqkv, out=bq_out))
    r = p_ln_bwd(dy1, x2, ln1w, mean1, rstd1, dres=dx1, dx_dtype=rdt, want_bf16=two, want_dres_colsum=True,
                 outs=(_slot_out(s_ln1w, (D,)) if need[1] else None, _slot_out(s_ln1b, (D,)) if need[2] else None,
                       _slot_out(s_bo, (D,)) if need[6] else None), defer=rq)
    if rq is not None and own_rq:
        rq.flush(reduce_side)
    dln1w, dln1b = r[1], r[2]
    dbo = r[-1] if need[6] else None                             # colsum(dx1), fused into the LN1 backward
    if side is not None:
        if keep is not None:   # joined by the caller: until then nothing the side stream reads may be recycled
            keep.extend(t for t in (g16, h, du, y2, dx1_16, o, dqkv, y1, part) if t is not None)
        else:
            main.wait_stream(side)  # every buffer the side stream read may be recycled after this point
    dx32 = None if chain else r[0]
    dx16 = r[0] if chain else (r[3] if two else None)
    return dx32, dx16, (dln1w, dln1b, dwqkv, dbqkv, dwo, dbo, dln2w, dln2b, dwfc, dbfc, dwpr, dbpr)


class ResBlockFn(Function):
    """Fused pre-LN residual attention block (modules/module_clip_ttransformer.py:20-37,
    module_seg_vit.py:175-196, module_mae.py:185-201):
        x += out_proj(MHA(LN1 x));  x += c_proj(act(c_fc(LN2 x)))
    x (B,T,D) fp32 residual stream.  Backward is hand-scheduled: residual-gradient adds are fused into
    the LayerNorm backward, act' into the c_proj dgrad epilogue, and no gradient is re-read."""

    @staticmethod
    def forward(ctx, x, ln1w, ln1b, wqkv, bqkv, wo, bo, ln2w, ln2b, wfc, bfc, wpr, bpr, n_head, causal, act, eps,
                act_dtype, klen=None):
        B, T, D = x.shape
        x2 = x.contiguous().view(B * T, D)
        P = (ln1w, ln1b, wqkv, bqkv, wo, bo, ln2w, ln2b, wfc, bfc, wpr, bpr)
        xo, saved = _resblock_fwd(x2, P, B, T, n_head, causal, act, eps, act_dtype, klen)
        ctx.save_for_backward(*saved)
        ctx.cfg = (B, T, D, n_head, causal, act, act_dtype)
        ctx.klen = klen
        from . import config as _cfg
        ctx.overlap_wgrad = bool(_cfg.overlap_wgrad)
        ctx.gslots = tuple(_slot_of(w) for w in (wqkv, wo, wfc, wpr))
        ctx.vslots = tuple(_slot_of(w) for w in (ln1w, ln1b, bqkv, bo, ln2w, ln2b, bfc, bpr))
        _GradFold.other[id(ln1w)] = weakref.ref(ln1w)        # a producer that does not fold: its block's gradients go the engine's way
        return xo.view(B, T, D)

    @staticmethod
    def backward(ctx, g):
        B, T, D, n_head, causal, act, act_dtype = ctx.cfg
        M = B * T
        st = getattr(g, "_segclip_bf16", None)  # bf16 copy left by the next block's LayerNorm backward (same object)
        if st is not None and (st.numel() != g.numel() or st.device != g.device or not g.is_contiguous()):
            st = None
        g = g.contiguous().view(M, D)
        g16 = st.view(M, D) if (st is not None and act_dtype == torch.bfloat16) else None
        dx, dx16, grads = _resblock_bwd(ctx.saved_tensors, ctx.cfg, ctx.klen, ctx.gslots, ctx.vslots,
                                        ctx.needs_input_grad, g, g16, False, ctx.overlap_wgrad)
        dx = dx.view(B, T, D)
        if dx16 is not None:
            dx._segclip_bf16 = dx16.view(B, T, D)
        return (dx,) + grads + (None, None, None, None, None, None)


class _GradFold:
    """config.fold_param_grads: the first gradient a parameter receives in a backward pass is remembered (and goes to autograd
    as usual: it sits in the engine's input buffer until every producer of that parameter has run); a later gradient of the
    same pass is added INTO that tensor by one multi-tensor launch per caller and autograd gets None for it.  An entry is
    valid for one pass: the epoch advances when the pass ends (engine callback) and at every model forward (a backward that
    raised never ran its callback)."""
    epoch = 0
    armed = -1
    first = {}
    uses = {}      # id(first parameter of a block) -> ResStackFn forwards over that block since the last advance(): armed from 2
    other = {}     # id(first parameter of a block) -> weak reference to it, for blocks a NON-stack node (ResBlockFn) used since the last advance(): never folded.
    #                Folding is only sound while every producer of a parameter's gradient in the pass is a folding ResStackFn node: a
    #                gradient delivered by another producer in between makes the engine build a NEW sum tensor (the alias below raises
    #                the storage's use count), and a later in-place add would land in the stale one

    @classmethod
    def advance(cls):
        cls.epoch += 1
        cls.first.clear()
        cls.uses.clear()
        cls.other.clear()

    @classmethod
    def used_elsewhere(cls, p):
        r = cls.other.get(id(p))
        return r is not None and r() is p      # (an id alone may be a dead tensor's, reused)

    @classmethod
    def take(cls, p, gr, adds):
        """gr: this node's gradient of parameter p -> what the node returns to autograd for it"""
        if gr is None or gr.dtype != torch.float32 or not gr.is_contiguous():
            return gr
        ent = cls.first.get(id(p))
        if ent is not None and ent[0] == cls.epoch and ent[1].shape == gr.shape and ent[1].device == gr.device:
            adds.append((ent[1], gr))
            return None
        # an ALIAS of the gradient (same storage, its own TensorImpl): AccumulateGrad takes an incoming gradient over without
        # a copy only while nothing else references that tensor object - a plain reference here costs one clone per parameter
        cls.first[id(p)] = (cls.epoch, gr.detach())
        if cls.armed != cls.epoch:
            cls.armed = cls.epoch
            torch.autograd.Variable._execution_engine.queue_callback(cls.advance)
        return gr

    @staticmethod
    def flush(adds):
        if not adds:
            return
        n = len(adds)
        dst = (C.c_void_p * n)(*[d.data_ptr() for d, _ in adds])
        src = (C.c_void_p * n)(*[s_.data_ptr() for _, s_ in adds])
        cnt = (C.c_int64 * n)(*[d.numel() for d, _ in adds])
        L.check(L.load().segclip_multi_add_f32(C.cast(dst, C.c_void_p), C.cast(src, C.c_void_p), C.cast(cnt, C.c_void_p), n,
                                               L.stream()), "multi_add_f32")
        adds.clear()


class ResStackFn(Function):
    """N consecutive ResBlockFn blocks as ONE autograd node (the towers: 10 + 2 vision blocks, 12 text blocks).  Inside
    the node the gradients travel between the blocks as plain tensors, so in bf16 mode (config.bf16_resgrad) the
    residual-stream gradient is a single bf16 tensor per LayerNorm backward (10 instead of 16 bytes per element); the
    node's own input / output gradients stay fp32.  With segclip_amd.dist.GradSync active, each block's parameter
    gradients are published (p.grad = bucket slot, ready hook) as soon as the block's backward has been enqueued, not
    when the whole stack returns, so the bucket all-reduces keep overlapping with the rest of the backward.
    inputs: x, n_head, causal, act, eps, act_dtype, klen, chain, keep16, then 12 parameters per block."""

    NP = 12

    @staticmethod
    def forward(ctx, x, n_head, causal, act, eps, act_dtype, klen, chain, keep16, *params):
        B, T, D = x.shape
        nblk = len(params) // ResStackFn.NP
        cur = x.contiguous().view(B * T, D)
        from . import config as _cfg
        # config.pad_rows (bf16 mode): a token-row count that is not a multiple of 128 (the text tower's B x 77 at any batch
        # that is not a multiple of 128 samples) keeps every GEMM of the stack off the 256 x 256-tile kernel, off the one-byte
        # derivative and - when it is not a multiple of 64 either - off the grouped weight gradients.  The stack then runs on
        # M rounded up to 128 rows: pad rows start as zeros, every operator of a block is row-wise except attention (which
        # only touches the B * T token rows; its outputs' pad rows are zeroed), so pad rows stay finite, carry exactly zero
        # gradients and contribute exact zeros to every weight / bias gradient.
        M = B * T
        Mp = -(-M // 128) * 128 if (act_dtype == torch.bfloat16 and bool(_cfg.pad_rows) and M % 128 != 0 and M >= _PAD_ROWS_MIN) else M
        ctx.Mp = Mp
        # config.bf16_resid (bf16 mode): the residual stream is bf16 between the blocks of the stack (the out_proj / c_proj
        # epilogues read and write 2 instead of 4 bytes per element and run on gemm_bf16_pq.hip; every LayerNorm pass reads
        # 2 bytes less).  Input / output of the node: fp32, or bf16 when the neighbouring node is a stack too (keep16).
        ctx.in_dtype = x.dtype
        resid16 = act_dtype == torch.bfloat16 and (bool(_cfg.bf16_resid) or x.dtype == torch.bfloat16)
        if resid16 and cur.dtype != torch.bfloat16:
            cur = p_cast(cur, torch.bfloat16)
        elif not resid16 and cur.dtype != torch.float32:
            cur = p_cast(cur, torch.float32)
        if Mp != M:
            padded = _empty((Mp, D), cur.dtype, cur)
            padded[:M].copy_(cur)
            padded[M:].zero_()
            cur = padded
        saved = []
        for b in range(nblk):
            cur, sv = _resblock_fwd(cur, params[b * 12:(b + 1) * 12], B, T, n_head, causal, act, eps, act_dtype, klen)
            saved.extend(sv)
        if Mp != M:
            cur = cur[:M]
        ctx.save_for_backward(*saved)
        ctx.cfg = (B, T, D, n_head, causal, act, act_dtype)
        ctx.klen, ctx.nblk = klen, nblk
        ctx.chain = (bool(chain) or resid16) and act_dtype == torch.bfloat16
        ctx.overlap_wgrad = bool(_cfg.overlap_wgrad)
        ctx.wgrad_group, ctx.wgrad_group_dist = int(_cfg.wgrad_group_blocks), int(_cfg.wgrad_group_blocks_dist)
        ctx.fold = bool(_cfg.fold_param_grads) and not (ctx.overlap_wgrad or bool(_cfg.reduce_side)) and len(params) > 0
        if ctx.fold:     # per block (the same blocks may be cut into different stacks by the two passes)
            for b in range(nblk):
                k = id(params[b * 12])
                _GradFold.uses[k] = _GradFold.uses.get(k, 0) + 1
        ctx.params = params
        ctx.slots = tuple(_slot_of(w) for w in params)
        if resid16 and not keep16:
            cur = p_cast(cur, torch.float32)
        return cur.view(B, T, D)

    @staticmethod
    def backward(ctx, g):
        B, T, D, n_head, causal, act, act_dtype = ctx.cfg
        M, nblk = B * T, ctx.nblk
        Mp = ctx.Mp
        bf = act_dtype == torch.bfloat16
        st = getattr(g, "_segclip_bf16", None)
        if st is not None and (st.numel() != g.numel() or st.device != g.device or not g.is_contiguous()):
            st = None
        g = g.contiguous().view(M, D)
        if Mp != M:               # row-padded stack: the gradient of a pad row is zero
            st = None
            padded = _empty((Mp, D), g.dtype, g)
            padded[:M].copy_(g)
            padded[M:].zero_()
            g, M = padded, Mp
        cur16 = st.view(M, D) if (st is not None and bf) else None
        if bf and cur16 is None and g.dtype == torch.bfloat16:   # bf16 output of the node (keep16)
            cur16 = g
        if ctx.chain and cur16 is None:
            cur16 = p_cast(g, act_dtype)
        cur32 = None if ctx.chain else (g if g.dtype == torch.float32 else p_cast(g, torch.float32))
        saved = ctx.saved_tensors
        need_all = ctx.needs_input_grad
        out = [None] * (nblk * 12)
        keep = [] if (ctx.overlap_wgrad and _WGRAD_JOIN_STACK) else None
        # config.reduce_side: the blocks' trailing reductions (split-K combines, LayerNorm / bias column sums) go to a side
        # stream and are joined once, at the end of the stack - not with GradSync slots (their bucket exchange is ordered
        # against the stream that PRODUCED a gradient, which must then be this one)
        from . import config as _cfg
        rside = _reduce_stream() if (_cfg.reduce_side and not ctx.overlap_wgrad and all(s is None for s in ctx.slots)) else None
        # config.wgrad_group_blocks (captured at forward time): the weight gradients of up to that many consecutive blocks run
        # as one grouped launch (WgradGroup); with GradSync slots the groups stay short so that the bucket exchanges keep
        # overlapping with the backward pass.  A block's gradients are published when its group has been enqueued.
        wg, sizes = None, []
        if ctx.wgrad_group > 1 and ctx.cfg[-1] == torch.bfloat16 and not ctx.overlap_wgrad and nblk > 1:
            Dm, F4 = ctx.params[2].shape[1], ctx.params[8].shape[0]
            if Dm % 256 == 0 and F4 % 256 == 0 and Mp % 64 == 0:
                gmax = ctx.wgrad_group if all(s is None for s in ctx.slots) else min(ctx.wgrad_group, ctx.wgrad_group_dist)
                tiles_blk = (4 * Dm * Dm + 2 * F4 * Dm) // 65536
                sizes = wgrad_group_plan(nblk, tiles_blk, Mp // 64, gmax)
                wg = WgradGroup() if max(sizes) > 1 else None
        left = sizes.pop(0) if wg is not None else 0
        grq = ReduceQueue() if (wg is not None and rside is None and _SHARED_RQ) else None
        pending = []
        adds = []
        for b in reversed(range(nblk)):
            P = ctx.params[b * 12:(b + 1) * 12]
            sl = ctx.slots[b * 12:(b + 1) * 12]
            need = (True,) + tuple(need_all[9 + b * 12 + i] for i in range(12))
            gslots = (sl[2], sl[4], sl[8], sl[10])
            vslots = (sl[0], sl[1], sl[3], sl[5], sl[6], sl[7], sl[9], sl[11])
            cur32, cur16, grads = _resblock_bwd(saved[b * N_SAVED:(b + 1) * N_SAVED], ctx.cfg, ctx.klen, gslots, vslots,
                                                need, cur32, cur16, ctx.chain, ctx.overlap_wgrad, keep, rside, wg, grq)
            pending.append((b, P, sl, grads))
            if wg is not None:
                left -= 1
                if left > 0 and b > 0:
                    continue
                wg.flush()
                if grq is not None:
                    grq.flush()
                left = sizes.pop(0) if sizes else nblk
            for b_, P_, sl_, grads_ in pending:
                fold = ctx.fold and _GradFold.uses.get(id(P_[0]), 0) > 1 and not _GradFold.used_elsewhere(P_[0])   # only blocks a second STACK node of this pass shares
                for i, (p, gr, slot) in enumerate(zip(P_, grads_, sl_)):
                    if gr is None:
                        continue
                    owner = slot.owner() if slot is not None else None
                    if owner is not None and gr.data_ptr() == slot.ptr and p.grad is None and slot.single_use():
                        # zero-copy gradient inside its all-reduce bucket: publish it now (autograd gets None for it)
                        p.grad = gr
                        owner._on_grad(p)
                    elif fold and slot is None:
                        out[b_ * 12 + i] = _GradFold.take(p, gr, adds)
                    else:
                        out[b_ * 12 + i] = gr
            pending = []
        _GradFold.flush(adds)
        if keep is not None:
            torch.cuda.current_stream().wait_stream(_wgrad_stream())
            keep.clear()
        if rside is not None:
            torch.cuda.current_stream().wait_stream(rside)     # the parameter gradients are complete when the node returns
        if Mp != B * T:                      # row-padded stack: back to the token rows
            cur16 = cur16[:B * T] if cur16 is not None else None
            cur32 = cur32[:B * T] if cur32 is not None else None
        if ctx.chain and ctx.in_dtype == torch.bfloat16:
            dx = cur16                       # the producer is a stack with a bf16 output: no cast, no side copy
        elif ctx.chain:
            dx = p_cast(cur16, torch.float32)
        else:
            dx = cur32
        dx = dx.view(B, T, D)
        if cur16 is not None and dx.dtype != torch.bfloat16:
            dx._segclip_bf16 = cur16.view(B, T, D)
        return (dx, None, None, None, None, None, None, None, None) + tuple(out)


def res_stack(x, blocks_params, n_head, causal, act, eps, act_dtype, klen=None, keep16=False):
    """Run consecutive residual blocks (each a 12-tuple of parameters in ResBlockFn order) as one ResStackFn node.
    keep16 (config.bf16_resid only): hand the bf16 residual stream to the caller as it is - for the next stack."""
    from . import config as _cfg
    flat = [p for P in blocks_params for p in P]
    keep16 = bool(keep16) and bool(_cfg.bf16_resid) and act_dtype == torch.bfloat16
    return ResStackFn.apply(x, n_head, causal, act, eps, act_dtype, klen, bool(_cfg.bf16_resgrad), keep16, *flat)


class CrossAttnFn(Function):
    """Attention core of CrossAttentionBlock (modules/module_seg_vit.py:215): queries (B*G, D) against
    the projected [centers; patches] buffer kv (B*S, 2D) = [K | V].
    mode "t18": the torch-1.8 key reshape reinterprets the (B,S,D) buffer as (S,B,D): sample b attends to
    flat tokens {r*B + b} (SURVEY.md finding 0.4).  mode "intended": its own S tokens."""

    @staticmethod
    def forward(ctx, qp, kv, B, G, S, n_head, mode):
        D = qp.shape[1]
        hd = D // n_head
        o = torch.empty_like(qp)
        ks = (2 * D, B * 2 * D) if mode == "t18" else (S * 2 * D, 2 * D)
        ad = _attn_desc(qp, kv, kv, o, B, n_head, G, S, hd, (G * D, D), ks, ks, (G * D, D), 1.0 / math.sqrt(hd),
                        False, 0, 0, D)
        stats = p_attn_fwd(ad, qp)
        ctx.save_for_backward(qp, kv, o, stats)
        ctx.cfg = (B, G, S, n_head, mode)
        return o

    @staticmethod
    def backward(ctx, do):
        qp, kv, o, stats = ctx.saved_tensors
        B, G, S, n_head, mode = ctx.cfg
        D = qp.shape[1]
        hd = D // n_head
        do = do.contiguous()
        ks = (2 * D, B * 2 * D) if mode == "t18" else (S * 2 * D, 2 * D)
        ad = _attn_desc(qp, kv, kv, o, B, n_head, G, S, hd, (G * D, D), ks, ks, (G * D, D), 1.0 / math.sqrt(hd),
                        False, 0, 0, D)
        dq = torch.empty_like(qp)
        dkv = torch.empty_like(kv)
        p_attn_bwd(ad, stats, do, dq, dkv, dkv, (G * D, D), ks, ks, (G * D, D), 0, 0, D)
        return dq, dkv, None, None, None, None, None


class CrossInProjAttnFn(Function):
    """nn.MultiheadAttention of CrossAttentionBlock up to (not including) out_proj, with different query and key/value
    inputs (modules/module_seg_vit.py:215): q = xq w[:D]^T + b[:D], [k | v] = xk w[D:]^T + b[D:], o = CrossAttnFn core.
    One Function instead of two LinearFn on slices of in_proj_weight + CrossAttnFn: the weight gradient is ONE (3D, D)
    buffer written by both wgrads (autograd's slice backward cost 4 zero fills + 4 copies + 2 adds per layer), the
    bf16 weight comes from the parameter's shadow, and in bf16 mode the bias gradient is the attention backward's
    per-sample token sums (no column-sum pass over the (B*S, 2D) gradient)."""

    @staticmethod
    def forward(ctx, xq, xk, w, b, B, G, S, n_head, mode, act_dtype):
        D = w.shape[1]
        hd = D // n_head
        xq, xk = xq.reshape(B * G, D), xk.reshape(B * S, D)
        if xq.dtype != act_dtype:
            xq = p_cast(xq, act_dtype)
        if xk.dtype != act_dtype:
            xk = p_cast(xk, act_dtype)
        wc = wcast(w, act_dtype)
        bd = b.detach()
        qp = p_linear(xq, wc[:D], bd[:D])[0]
        kv = p_linear(xk, wc[D:], bd[D:])[0]
        o = torch.empty_like(qp)
        ks = (2 * D, B * 2 * D) if mode == "t18" else (S * 2 * D, 2 * D)
        ad = _attn_desc(qp, kv, kv, o, B, n_head, G, S, hd, (G * D, D), ks, ks, (G * D, D), 1.0 / math.sqrt(hd),
                        False, 0, 0, D)
        stats = p_attn_fwd(ad, qp)
        ctx.save_for_backward(xq, xk, wc, qp, kv, o, stats)
        ctx.cfg = (B, G, S, n_head, mode)
        ctx.gslot = _slot_of(w)
        return o

    @staticmethod
    def backward(ctx, do):
        xq, xk, wc, qp, kv, o, stats = ctx.saved_tensors
        B, G, S, n_head, mode = ctx.cfg
        D = qp.shape[1]
        hd = D // n_head
        do = do.contiguous()
        ks = (2 * D, B * 2 * D) if mode == "t18" else (S * 2 * D, 2 * D)
        ad = _attn_desc(qp, kv, kv, o, B, n_head, G, S, hd, (G * D, D), ks, ks, (G * D, D), 1.0 / math.sqrt(hd),
                        False, 0, 0, D)
        dq = torch.empty_like(qp)
        dkv = torch.empty_like(kv)
        part = _empty((B, 3 * D), torch.float32, do) if qp.dtype == torch.bfloat16 else None
        p_attn_bwd(ad, stats, do, dq, dkv, dkv, (G * D, D), ks, ks, (G * D, D), 0, 0, D, colsum_part=part)
        dxq = p_dgrad(dq, wc[:D], xq.dtype)
        dxk = p_dgrad(dkv, wc[D:], xk.dtype)
        dw = _slot_out(ctx.gslot, (3 * D, D))
        if dw is None:
            dw = _empty((3 * D, D), torch.float32, do)
        p_wgrad(dq, xq, out=dw[:D])
        p_wgrad(dkv, xk, out=dw[D:])
        if part is not None:
            db = p_colsum(part)
        else:
            db = _empty((3 * D,), torch.float32, do)
            p_colsum(dq, out=db[:D])
            p_colsum(dkv, out=db[D:])
        return dxq.view(B, G, D), dxk.view(B, S, D), dw, db, None, None, None, None, None, None


class CrossBlockFn(Function):
    """One CrossAttentionBlock of the learnable-center stage (reference modules/module_seg_vit.py:199-218) as ONE autograd node:
        q1 = q + out_proj(MHA(ln_x q, ln_k kv, ln_k kv)),  kv = cat([q, tokens]);   q2 = q1 + c_proj(QuickGELU(c_fc(ln_2 q1)))
    q (B, G, D) fp32; kn_buf (B, S, D) in the compute dtype whose rows [G, S) of every sample already hold ln_k(tokens)
    (LayerNormMultiFn) - rows [0, G) are written here, in place (ln_k q).  The block is 8 center rows per sample: every
    kernel is a few microseconds, so the node count is the cost - as separate LinearFn / LayerNormFn nodes its backward was
    ~45 launches (a split-K combine, a bias column sum and its reduction per Linear, a reduction per LayerNorm, an add per
    residual branch); here the residual adds ride in the LayerNorm backwards, the bias gradients come out of them / the
    attention backward / the c_proj data gradient, and every trailing reduction joins ONE queue (2 launches).
    Returns (q2, kn_buf); the gradient of kn_buf covers all S rows (its consumer reads the token rows only)."""

    NPARAM = 14

    @staticmethod
    def forward(ctx, q, kn_buf, lnxw, lnxb, lnkw, lnkb, w_in, b_in, wo, bo, ln2w, ln2b, wfc, bfc, wpr, bpr,
                n_head, mode, eps, act_dtype):
        B, G, D = q.shape
        S = kn_buf.shape[1]
        M, hd, F4 = B * G, D // n_head, wfc.shape[0]
        act = ACT_QUICK_GELU
        for w in (lnxw, lnxb, lnkw, lnkb, w_in, b_in, wo, bo, ln2w, ln2b, wfc, bfc, wpr, bpr):
            sl = _slot_of(w)
            if sl is not None:
                sl.note_forward_use()
        q2d = q.contiguous().view(M, D)
        kn2d = kn_buf.view(B * S, D)
        qn, mx, rx = p_ln_fwd(q2d, lnxw, lnxb, eps[0], act_dtype)
        _, mk, rk = p_ln_fwd(q2d, lnkw, lnkb, eps[1], act_dtype, out=kn2d, seg=(G, S, 0))
        ctx.mark_dirty(kn_buf)
        wc, wo_c, wfc_c, wpr_c = (wcast(w, act_dtype) for w in (w_in, wo, wfc, wpr))
        bd = b_in.detach()
        qp = p_linear(qn, wc[:D], bd[:D])[0]
        kv = p_linear(kn2d, wc[D:], bd[D:])[0]
        o = torch.empty_like(qp)
        ks = (2 * D, B * 2 * D) if mode == "t18" else (S * 2 * D, 2 * D)
        ad = _attn_desc(qp, kv, kv, o, B, n_head, G, S, hd, (G * D, D), ks, ks, (G * D, D), 1.0 / math.sqrt(hd), False, 0, 0, D)
        stats = p_attn_fwd(ad, qp)
        q1 = p_linear(o, wo_c, bo, residual=q2d, out_dtype=torch.float32)[0]
        z, m2, r2 = p_ln_fwd(q1, ln2w, ln2b, eps[2], act_dtype)
        h, u = p_linear(z, wfc_c, bfc, act=act, want_aux=True, aux_kind=_aux_kind(act_dtype, act, M, F4))
        q2 = p_linear(h, wpr_c, bpr, residual=q1, out_dtype=torch.float32)[0]
        # (kn_buf itself, not its 2-D view: a view made here of a tensor this node marks dirty may not be saved)
        ctx.save_for_backward(q2d, lnxw, mx, rx, qn, lnkw, mk, rk, kn_buf, wc, qp, kv, o, stats, wo_c, q1, ln2w, m2, r2, z, wfc_c, u, h,
                              wpr_c)
        ctx.cfg = (B, G, S, D, n_head, mode, act, act_dtype)
        ctx.gslots = tuple(_slot_of(w) for w in (w_in, wo, wfc, wpr))
        return q2.view(B, G, D), kn_buf

    @staticmethod
    def backward(ctx, dq2, _dbuf):
        (q2d, lnxw, mx, rx, qn, lnkw, mk, rk, kn_buf, wc, qp, kv, o, stats, wo_c, q1, ln2w, m2, r2, z, wfc_c, u, h,
         wpr_c) = ctx.saved_tensors
        B, G, S, D, n_head, mode, act, act_dtype = ctx.cfg
        kn2d = kn_buf.detach().view(B * S, D)
        M, hd, F4 = B * G, D // n_head, wfc_c.shape[0]
        bf = act_dtype == torch.bfloat16
        s_in, s_o, s_fc, s_pr = ctx.gslots
        g = dq2.contiguous().view(M, D)
        if g.dtype != torch.float32:
            g = p_cast(g, torch.float32)
        g16 = p_cast(g, act_dtype) if bf else g
        rq = ReduceQueue()
        # the four weight gradients over the block's 8 center rows per sample (B*G rows: 9-36 output tiles each, 42 us apiece as
        # separate split-K launches) run as ONE grouped launch at the end of the node (round 6; bf16 mode)
        wg = WgradGroup() if (bf and _CROSS_WGRAD_GROUP) else None

        def small_wgrad(dy_, x_, out_):
            if wg is not None and WgradGroup.covers(dy_, x_, out_):
                return wg.add(dy_, x_, out=out_)
            return p_wgrad(dy_, x_, out=out_, defer=rq)
        # ---- MLP
        du, dbfc = p_dgrad(g16, wpr_c, act_dtype, aux=u, act=act, want_colsum=True,
                           aux_kind=2 if u.dtype == torch.uint8 else _aux_kind(act_dtype, act), defer=rq)
        dwpr = small_wgrad(g16, h, _slot_out(s_pr, (D, F4)))
        dz = p_dgrad(du, wfc_c, act_dtype)
        dwfc = small_wgrad(du, z, _slot_out(s_fc, (F4, D)))
        r = p_ln_bwd(dz, q1, ln2w, m2, r2, dres=g, dx_dtype=torch.float32, want_bf16=bf, want_dres_colsum=True, defer=rq)
        dq1, dln2w, dln2b = r[0], r[1], r[2]
        dq1_16 = r[3] if bf else dq1
        dbpr = r[-1]                                             # colsum(g): the c_proj bias gradient
        # ---- attention
        do = p_dgrad(dq1_16, wo_c, act_dtype)
        dwo = small_wgrad(dq1_16, o, _slot_out(s_o, (D, D)))
        ks = (2 * D, B * 2 * D) if mode == "t18" else (S * 2 * D, 2 * D)
        ad = _attn_desc(qp, kv, kv, o, B, n_head, G, S, hd, (G * D, D), ks, ks, (G * D, D), 1.0 / math.sqrt(hd), False, 0, 0, D)
        dqp, dkv = torch.empty_like(qp), torch.empty_like(kv)
        part = _empty((B, 3 * D), torch.float32, do) if bf else None
        p_attn_bwd(ad, stats, do, dqp, dkv, dkv, (G * D, D), ks, ks, (G * D, D), 0, 0, D, colsum_part=part)
        dqn = p_dgrad(dqp, wc[:D], act_dtype)
        dkn = p_dgrad(dkv, wc[D:], act_dtype)
        dw_in = _slot_out(s_in, (3 * D, D))
        if dw_in is None:
            dw_in = _empty((3 * D, D), torch.float32, do)
        small_wgrad(dqp, qn, dw_in[:D])
        p_wgrad(dkv, kn2d, out=dw_in[D:], defer=rq)
        db_in = _empty((3 * D,), torch.float32, do)
        if part is not None:
            rq.add_rows(part, B, 3 * D, 3 * D, (db_in,), 3 * D)
        else:
            p_colsum(dqp, out=db_in[:D])
            p_colsum(dkv, out=db_in[D:])
        # ---- the two LayerNorms of q: ln_x (dres = dq1: the residual branch; its column sums = the out_proj bias gradient),
        # then ln_k's center rows (dres = what the first one produced)
        r = p_ln_bwd(dqn, q2d, lnxw, mx, rx, dres=dq1, dx_dtype=torch.float32, want_dres_colsum=True, defer=rq)
        dq_a, dlnxw, dlnxb, dbo = r[0], r[1], r[2], r[-1]
        r = p_ln_bwd(dkn, q2d, lnkw, mk, rk, dres=dq_a, dx_dtype=torch.float32, defer=rq, seg=(G, S, 0))
        dq, dlnkw, dlnkb = r[0], r[1], r[2]
        if wg is not None:
            wg.flush()
        rq.flush()
        return (dq.view(B, G, D), dkn.view(B, S, D), dlnxw, dlnxb, dlnkw, dlnkb, dw_in, db_in, dwo, dbo, dln2w, dln2b, dwfc, dbfc,
                dwpr, dbpr, None, None, None, None)


class PatchEmbedFn(Function):
    """conv1 (16x16/16, no bias) as im2col + GEMM with the positional table fused as the epilogue
    residual (modules/module_clip_vtransformer.py:56-64).  The CLS row the reference prepends is
    discarded by SegViT before any use (modules/module_seg_vit.py:419), so it is never materialised;
    class_embedding therefore receives an all-zero gradient, exactly as in the reference.
    bf16 mode pads the contraction dimension 3*p*p to a multiple of 64 with zero columns (ViT-L/14: 588 -> 640) so
    that the patch GEMM runs on the LDS-DMA kernels."""

    @staticmethod
    def forward(ctx, image, conv_w, cls, pos, patch, act_dtype):
        lib = L.load()
        B, Cc, H, W = image.shape
        D = conv_w.shape[0]
        T = (H // patch) * (W // patch)
        Kd = Cc * patch * patch
        Kp = Kd if (act_dtype != torch.bfloat16 or Kd % 64 == 0) else -(-Kd // 64) * 64
        image = image.contiguous()
        cols = _empty((B * T, Kp), act_dtype, image)
        L.check(lib.segclip_im2col_ld(L.ptr(image), L.ptr(cols), B, Cc, H, W, patch, 0, L.dt(cols), Kp, L.stream()), "im2col")
        if Kp == Kd:
            wc = wcast(conv_w.reshape(D, Kd), act_dtype)
        else:
            wc = torch.zeros((D, Kp), dtype=act_dtype, device=image.device)
            wc[:, :Kd] = p_cast(conv_w.detach().reshape(D, Kd).contiguous(), act_dtype)
        x = _empty((B, T, D), torch.float32, image)
        posc = pos.detach().contiguous()
        done = False
        if act_dtype == torch.bfloat16 and (B * T) % 128 == 0 and D % 256 == 0:
            # one (B*T, D) GEMM whose epilogue adds positional row (m % T) - the batched form below runs B problems of T = 196
            # rows on 256-row tiles (289 us against 105 us at B = 256)
            try:
                p_gemm(cols, wc, x, B * T, D, Kp, (Kp, 1), (Kp, 1), D, residual=posc, ldr=D, r_off=D, r_mod=T)
                done = True
            except L.Unsupported:
                pass
        if not done:
            p_gemm(cols, wc, x, T, D, Kp, (Kp, 1), (Kp, 1), D, residual=posc, ldr=D, r_off=D, nb1=B, bsA=(T * Kp, 0),
                   bsC=(T * D, 0), bsR=(0, 0))
        ctx.save_for_backward(cols)
        ctx.shape = (B, T, D, Kd, Kp, tuple(conv_w.shape), tuple(cls.shape), tuple(pos.shape))
        return x

    @staticmethod
    def backward(ctx, dx):
        (cols,) = ctx.saved_tensors
        B, T, D, Kd, Kp, wshape, cshape, pshape = ctx.shape
        dx = dx.contiguous()
        dw = dcls = dpos = None
        if ctx.needs_input_grad[1]:
            dw = p_wgrad(dx.view(B * T, D), cols)
            dw = (dw if Kp == Kd else dw[:, :Kd].contiguous()).view(wshape)
        if ctx.needs_input_grad[2]:
            dcls = torch.zeros(cshape, dtype=torch.float32, device=dx.device)
        if ctx.needs_input_grad[3]:
            dpos = torch.zeros(pshape, dtype=torch.float32, device=dx.device)
            dpos[1:] = p_colsum(dx.view(B, T * D)).view(T, D)
        return None, dw, dcls, dpos, None, None


class EmbedFn(Function):
    """token_embedding(ids) + positional_embedding[:L]  (modules/module_clip.py:109-112)."""

    @staticmethod
    def forward(ctx, ids, table, pos):
        B, Lq = ids.shape
        V, D = table.shape
        ids = ids.contiguous()
        out = _empty((B, Lq, D), torch.float32, table)
        posc = pos.detach().contiguous()
        L.check(L.load().segclip_embed_fwd(L.ptr(ids), L.ptr(table), L.ptr(posc), L.ptr(out), B, Lq, D, V, L.stream()),
                "embed_fwd")
        ctx.save_for_backward(ids)
        ctx.shape = (B, Lq, D, V, tuple(pos.shape))
        ctx.gslot = _slot_of(table)
        return out

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        B, Lq, D, V, pshape = ctx.shape
        dout = dout.contiguous()
        dtable = None
        if ctx.needs_input_grad[1]:
            dtable = _slot_out(ctx.gslot, (V, D))
            dtable = dtable.zero_() if dtable is not None else torch.zeros((V, D), dtype=torch.float32, device=dout.device)
        dpos_l = _empty((Lq, D), torch.float32, dout) if ctx.needs_input_grad[2] else None
        L.check(L.load().segclip_embed_bwd(L.ptr(ids), L.ptr(dout), L.ptr(dtable), L.ptr(dpos_l), B, Lq, D, V,
                                           L.stream()), "embed_bwd")
        dpos = None
        if dpos_l is not None:
            dpos = torch.zeros(pshape, dtype=torch.float32, device=dout.device)
            dpos[:Lq] = dpos_l
        return None, dtable, dpos


class ReconMixFn(Function):
    """out (B,M,D) = a (B,M,8) @ x (B,8,D) in fp32 (reference modules/module_seg_vit.py:342: the token rows of the MAE branch rebuilt
    from the 8 centers) as one pass over the output; backward da = dout x^T, dx = a^T dout in one kernel."""

    @staticmethod
    def forward(ctx, a, x):
        B, M, G = a.shape
        D = x.shape[2]
        a, x = a.contiguous(), x.contiguous()
        out = _empty((B, M, D), torch.float32, a)
        L.check(L.load().segclip_recon_mix_fwd(L.ptr(a), L.ptr(x), L.ptr(out), B, M, G, D, L.stream()), "recon_mix_fwd")
        ctx.save_for_backward(a, x)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, x = ctx.saved_tensors
        B, M, G = a.shape
        D = x.shape[2]
        dout = dout.contiguous()
        da = _empty((B, M, G), torch.float32, a)
        dx = _empty((B, G, D), torch.float32, a)
        L.check(L.load().segclip_recon_mix_bwd(L.ptr(a), L.ptr(x), L.ptr(dout), L.ptr(da), L.ptr(dx), B, M, G, D, L.stream()),
                "recon_mix_bwd")
        return da, dx


# config.pad_rows applies from this many token rows on: below it a step is bound by the host's launch rate (per-GPU batch 64:
# 13.5 ms of enqueue per step), where the ~50 extra small launches of a padded tower cost more than the faster kernels return
_CROSS_WGRAD_GROUP = _tenv("SEGCLIP_CROSS_WGRAD_GROUP", "1") != "0"   # A/B: 0 = the center blocks' weight gradients as separate launches
_PAD_ROWS_MIN = int(_tenv("SEGCLIP_PAD_ROWS_MIN", "6144"))
_RECON_MIX = _tenv("SEGCLIP_RECON_MIX", "1") != "0"      # A/B: 0 = the batched exact-fp32 GEMM


def recon_mix(a, x):
    """a (B,M,G) @ x (B,G,D) -> (B,M,D), fp32: the dedicated kernel for G = 8, the batched GEMM otherwise."""
    if (_RECON_MIX and a.dtype == torch.float32 and x.dtype == torch.float32 and a.dim() == 3 and x.dim() == 3 and a.shape[2] == 8
            and x.shape[1] == 8 and x.shape[2] % 4 == 0 and 4 <= x.shape[2] <= 4096):
        return ReconMixFn.apply(a, x)
    return bmm(a, x, transB=False, out_dtype=torch.float32)


class MeanCatFn(Function):
    """cat([mean(x, dim=1, keepdim=True), x], dim=1) for x (B,T,D) fp32 (reference modules/modeling.py:240-242 and the MAE branch
    of modules/module_seg_vit.py: the mean over the tokens stands in for the CLS row): one kernel each way."""

    @staticmethod
    def forward(ctx, x):
        B, T, D = x.shape
        x = x.contiguous()
        out = _empty((B, T + 1, D), torch.float32, x)
        L.check(L.load().segclip_mean_cat_fwd(L.ptr(x), L.ptr(out), B, T, D, L.stream()), "mean_cat_fwd")
        ctx.shape = (B, T, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, T, D = ctx.shape
        dout = dout.contiguous()
        dx = _empty((B, T, D), torch.float32, dout)
        L.check(L.load().segclip_mean_cat_bwd(L.ptr(dout), L.ptr(dx), B, T, D, L.stream()), "mean_cat_bwd")
        return dx


def mean_cat(x):
    """x (B,T,D) -> (B,T+1,D) with the token mean as row 0; fp32 rows of a multiple of 4 columns on the HIP kernel."""
    if x.dtype == torch.float32 and x.dim() == 3 and x.shape[-1] % 4 == 0 and x.shape[1] > 0:
        return MeanCatFn.apply(x)
    return torch.cat([torch.mean(x, dim=1, keepdim=True), x], dim=1)


class MaeUnshuffleFn(Function):
    """Decoder input of the MAE heads (reference modules/module_mae.py:310-314 / 338-342):
    gather(cat([x, mask_token.expand(B, L - K, D)], 1), ids_restore) + pos_embed, without the concatenated tensor:
    out[b][j] = (ids[b][j] < K ? x[b][ids[b][j]] : mask_token) + pos[j].  x (B,K,D) fp32, mask_token (D) or (1,1,D), ids (B,L)
    int64 (a permutation of 0..L-1 per sample), pos (L,D) or (1,L,D)."""

    @staticmethod
    def forward(ctx, x, mask_token, ids, pos):
        B, K, D = x.shape
        Lq = ids.shape[1]
        x, ids = x.contiguous(), ids.contiguous()
        mt, pe = mask_token.detach().reshape(D).float().contiguous(), pos.detach().reshape(Lq, D).float().contiguous()
        out = _empty((B, Lq, D), torch.float32, x)
        L.check(L.load().segclip_mae_unshuffle_fwd(L.ptr(x), L.ptr(mt), L.ptr(ids), L.ptr(pe), L.ptr(out), B, K, Lq, D, L.stream()),
                "mae_unshuffle_fwd")
        ctx.save_for_backward(ids)
        ctx.dims = (B, K, Lq, D, mask_token.shape, pos.shape, mask_token.dtype, pos.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        B, K, Lq, D, mshape, pshape, mdt, pdt = ctx.dims
        dout = dout.contiguous()
        dx = _empty((B, K, D), torch.float32, dout)
        dpos = _empty((Lq, D), torch.float32, dout)
        mpart = _empty((Lq, D), torch.float32, dout)
        L.check(L.load().segclip_mae_unshuffle_bwd(L.ptr(dout), L.ptr(ids), L.ptr(dx), L.ptr(dpos), L.ptr(mpart), B, K, Lq, D, L.stream()),
                "mae_unshuffle_bwd")
        dmask = p_colsum(mpart) if ctx.needs_input_grad[1] else None
        return (dx, dmask.reshape(mshape).to(mdt) if dmask is not None else None, None,
                dpos.reshape(pshape).to(pdt) if ctx.needs_input_grad[3] else None)


class GatherRowsFn(Function):
    """out[b,j,:] = src[b, idx[b,j], :] with unique idx per b (EOT pick, MAE keep / un-shuffle)."""

    @staticmethod
    def forward(ctx, src, idx):
        B, Ts, D = src.shape
        To = idx.shape[1]
        src = src.contiguous()
        idx = idx.contiguous()
        out = _empty((B, To, D), src.dtype, src)
        L.check(L.load().segclip_gather_rows(L.ptr(src), L.ptr(idx), L.ptr(out), B, Ts, To, D, L.dt(src), L.stream()),
                "gather_rows")
        ctx.save_for_backward(idx)
        ctx.shape = (B, Ts, To, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        B, Ts, To, D = ctx.shape
        dout = dout.contiguous()
        dsrc = torch.zeros((B, Ts, D), dtype=dout.dtype, device=dout.device)
        L.check(L.load().segclip_scatter_rows(L.ptr(dout), L.ptr(idx), L.ptr(dsrc), B, Ts, To, D, L.dt(dout),
                                              L.stream()), "scatter_rows")
        return dsrc, None


class AssignFn(Function):
    """gumbel_softmax(hard=True, dim=centers) + soft assignment (modules/module_seg_vit.py:221-242,305-306).
    logits (B,G,T) fp32, gumbel (B,G,T) fp32 or None (eval).  Returns hard (straight-through gradient),
    soft (no grad, like the reference's use), idx uint8 (B,T)."""

    @staticmethod
    def forward(ctx, logits, gumbel, tau):
        B, G, T = logits.shape
        logits = logits.contiguous()
        g = gumbel.contiguous() if gumbel is not None else None
        y = torch.empty_like(logits)
        soft = torch.empty_like(logits)
        hard = torch.empty_like(logits)
        idx = torch.empty((B, T), dtype=torch.uint8, device=logits.device)
        counts = torch.empty((B, G), dtype=torch.float32, device=logits.device)
        L.check(L.load().segclip_assign_fwd(L.ptr(logits), L.ptr(g), tau if g is not None else 1.0, L.ptr(y),
                                            L.ptr(soft), L.ptr(idx), L.ptr(hard), L.ptr(counts), B, G, T, L.stream()),
                "assign_fwd")
        ctx.save_for_backward(y)
        ctx.tau = tau if g is not None else 1.0
        ctx.mark_non_differentiable(soft, idx, counts)
        return hard, soft, idx, counts

    @staticmethod
    def backward(ctx, dhard, _dsoft, _didx, _dcounts):
        (y,) = ctx.saved_tensors
        B, G, T = y.shape
        dhard = dhard.contiguous()
        dl = torch.empty_like(y)
        L.check(L.load().segclip_assign_bwd(L.ptr(dhard), L.ptr(y), ctx.tau, L.ptr(dl), B, G, T, L.stream()),
                "assign_bwd")
        return dl, None, None


class L2NormFn(Function):
    """x / ||x||  (modules/modeling.py:341-345)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        rows, cols = x.shape
        y = torch.empty_like(x)
        n = _empty((rows,), torch.float32, x)
        L.check(L.load().segclip_l2norm_fwd(L.ptr(x), L.ptr(y), L.ptr(n), rows, cols, L.stream()), "l2norm_fwd")
        ctx.save_for_backward(y, n)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, n = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        L.check(L.load().segclip_l2norm_bwd(L.ptr(dy), L.ptr(y), L.ptr(n), L.ptr(dx), y.shape[0], y.shape[1],
                                            L.stream()), "l2norm_bwd")
        return dx


class CrossEntropyFn(Function):
    """mean_i -log softmax(logits_i)[i + label_offset]  (nn.CrossEntropyLoss, modules/modeling.py:205-208)."""

    @staticmethod
    def forward(ctx, logits, label_offset):
        logits = logits.contiguous()
        rows, cols = logits.shape
        lib = L.load()
        lse = _empty((rows,), torch.float32, logits)
        lr = _empty((rows,), torch.float32, logits)
        L.check(lib.segclip_ce_fwd(L.ptr(logits), L.ptr(lse), L.ptr(lr), rows, cols, label_offset, L.stream()), "ce_fwd")
        loss = _empty((), torch.float32, logits)
        L.check(lib.segclip_reduce_sum(L.ptr(lr), L.ptr(loss), rows, 1.0 / rows, L.stream()), "reduce_sum")
        ctx.save_for_backward(logits, lse)
        ctx.label_offset = label_offset
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, lse = ctx.saved_tensors
        rows, cols = logits.shape
        g = g.contiguous().float()
        dl = torch.empty_like(logits)
        L.check(L.load().segclip_ce_bwd(L.ptr(logits), L.ptr(lse), L.ptr(g), 1.0, L.ptr(dl), rows, cols,
                                        ctx.label_offset, L.stream()), "ce_bwd")
        return dl, None


class CrossEntropyLabelsFn(Function):
    """nn.CrossEntropyLoss(ignore_index) with explicit labels: mean over the rows whose label is not ignored
    (text-MAE vocabulary loss, modules/module_mae.py:351-353).  logits (R, V) fp32, labels (R,) int64."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        logits = logits.contiguous()
        labels = labels.to(torch.int64).contiguous()
        L.require_cuda(logits, labels)
        rows, cols = logits.shape
        lib = L.load()
        lse, lr, valid = (_empty((rows,), torch.float32, logits) for _ in range(3))
        L.check(lib.segclip_ce_labels_fwd(L.ptr(logits), L.ptr(labels), int(ignore_index), L.ptr(lse), L.ptr(lr), L.ptr(valid),
                                          rows, cols, L.stream()), "ce_labels_fwd")
        tot, cnt, loss = (_empty((), torch.float32, logits) for _ in range(3))
        L.check(lib.segclip_reduce_sum(L.ptr(lr), L.ptr(tot), rows, 1.0, L.stream()), "reduce_sum")
        L.check(lib.segclip_reduce_sum(L.ptr(valid), L.ptr(cnt), rows, 1.0, L.stream()), "reduce_sum")
        inv = torch.reciprocal(cnt)
        L.check(lib.segclip_scale(L.ptr(tot), L.ptr(inv), L.ptr(loss), 1, L.stream()), "scale")
        ctx.save_for_backward(logits, lse, labels, inv)
        ctx.ignore_index = int(ignore_index)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, lse, labels, inv = ctx.saved_tensors
        rows, cols = logits.shape
        g = g.contiguous().float()
        dl = torch.empty_like(logits)
        L.check(L.load().segclip_ce_labels_bwd(L.ptr(logits), L.ptr(lse), L.ptr(labels), ctx.ignore_index, L.ptr(g),
                                               L.ptr(inv), L.ptr(dl), rows, cols, L.stream()), "ce_labels_bwd")
        return dl, None, None


def prefix_mask_lengths(attention_mask):
    """(B, L) 0/1 attention mask of END-PADDED captions (the dataloader contract, dataloaders/dataloader_cc_retrieval.py:
    valid tokens first) -> int32 (B,) number of valid keys, the form the attention kernels take the key-padding mask in.
    The prefix form (no interior zeros, no left padding, at least one valid key per row - a row without keys has
    lse = -inf) is checked ON EVERY CALL without a host synchronisation: the verdict is computed on the device and
    handed to torch._assert_async, which fails the stream (the error surfaces at the next synchronisation) instead of
    silently attending to the wrong keys.  SEGCLIP_CHECK_MASKS=1 additionally raises at the call site (one host sync)."""
    m = attention_mask.reshape(attention_mask.shape[0], -1)
    mi = m.to(torch.int32)
    ok = (mi[:, 1:] <= mi[:, :-1]).all() & (mi[:, 0] >= 1).all() & ((mi == 0) | (mi == 1)).all()
    if _CHECK_MASKS or not mi.is_cuda:
        if not bool(ok.item()):
            raise NotImplementedError("attention_mask must be a 0/1 PREFIX mask with >= 1 valid token per row "
                                      "(end-padded captions); interior zeros / left padding are not supported")
    else:
        torch._assert_async(ok, "segclip: attention_mask is not a 0/1 prefix mask (interior zeros / left padding / empty row)")
    return mi.sum(dim=1, dtype=torch.int32).contiguous()


_CHECK_MASKS = bool(int(__import__("os").environ.get("SEGCLIP_CHECK_MASKS", "0")))


class SuperpixelKLFn(Function):
    """Symmetric KL between the hard assignment and its superpixel mean (modules/modeling.py:212-224)."""

    @staticmethod
    def forward(ctx, hard, seg):
        B, G, T = hard.shape
        hard = hard.contiguous()
        L.require_cuda(hard, seg)
        if seg.is_floating_point() or seg.dtype == torch.bool:
            raise TypeError(f"superpixel_kl: image_seg must hold integer superpixel labels, got {seg.dtype}")
        seg = seg.to(torch.int64).contiguous().view(B, T)   # the kernel reads int64; loaders may hand over int32 / uint8
        lib = L.load()
        lr = _empty((B,), torch.float32, hard)
        dh = torch.empty_like(hard)
        L.check(lib.segclip_superpixel_kl(L.ptr(hard), L.ptr(seg), L.ptr(lr), L.ptr(dh), B, G, T, L.stream()),
                "superpixel_kl")
        loss = _empty((), torch.float32, hard)
        L.check(lib.segclip_reduce_sum(L.ptr(lr), L.ptr(loss), B, 1.0, L.stream()), "reduce_sum")
        ctx.save_for_backward(dh)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dh,) = ctx.saved_tensors
        g = g.contiguous().float()
        out = torch.empty_like(dh)
        L.check(L.load().segclip_scale(L.ptr(dh), L.ptr(g), L.ptr(out), dh.numel(), L.stream()), "scale")
        return out, None


class MaskedMSEFn(Function):
    """MAE reconstruction loss (modules/module_mae.py:322-328): pred (B,1+T,Dp) [row 0 = CLS, dropped],
    target (B,T,Dp) fp32 (patchify), mask (B,1+T) fp32."""

    @staticmethod
    def forward(ctx, pred, target, mask):
        B, T1, Dp = pred.shape
        T = T1 - 1
        pred = pred.contiguous()
        lib = L.load()
        lr = _empty((B * T,), torch.float32, pred)
        L.check(lib.segclip_masked_mse_fwd(L.ptr(pred), L.ptr(target), L.ptr(mask), L.ptr(lr), B, T, Dp, L.dt(pred),
                                           L.stream()), "masked_mse_fwd")
        msum = _empty((), torch.float32, pred)
        mflat = mask[:, 1:].contiguous()
        L.check(lib.segclip_reduce_sum(L.ptr(mflat), L.ptr(msum), B * T, 1.0, L.stream()), "reduce_sum")
        tot = _empty((), torch.float32, pred)
        L.check(lib.segclip_reduce_sum(L.ptr(lr), L.ptr(tot), B * T, 1.0, L.stream()), "reduce_sum")
        loss = _empty((), torch.float32, pred)
        # loss = tot / msum  (single-element division, done by the scale kernel with 1/msum)
        inv = torch.reciprocal(msum)
        L.check(lib.segclip_scale(L.ptr(tot), L.ptr(inv), L.ptr(loss), 1, L.stream()), "scale")
        ctx.save_for_backward(pred, target, mask, msum)
        return loss

    @staticmethod
    def backward(ctx, g):
        pred, target, mask, msum = ctx.saved_tensors
        B, T1, Dp = pred.shape
        g = g.contiguous().float()
        dp = torch.empty_like(pred)
        L.check(L.load().segclip_masked_mse_bwd(L.ptr(pred), L.ptr(target), L.ptr(mask), L.ptr(g), L.ptr(msum), 1.0,
                                                L.ptr(dp), B, T1 - 1, Dp, L.dt(pred), L.stream()), "masked_mse_bwd")
        return dp, None, None


def patchify_target(image, patch):
    """MAE target `patchify(imgs)` (modules/module_mae.py:18-29): (B,3,H,W) -> (B, T, p*p*3), fp32."""
    B, Cc, H, W = image.shape
    T = (H // patch) * (W // patch)
    image = image.contiguous()
    out = _empty((B, T, Cc * patch * patch), torch.float32, image)
    L.check(L.load().segclip_im2col(L.ptr(image), L.ptr(out), B, Cc, H, W, patch, 1, L.F32, L.stream()), "im2col")
    return out


def mask_sort(noise, len_keep):
    """MAE random masking indices (modules/module_clip_util.py:91-124, keep_cls=True).
    Returns ids_shuffle, ids_restore (int64) and mask (fp32), bit-exact functions of `noise`."""
    B, Lq = noise.shape
    noise = noise.contiguous().float()
    ids_shuffle = torch.empty((B, Lq), dtype=torch.int64, device=noise.device)
    ids_restore = torch.empty((B, Lq), dtype=torch.int64, device=noise.device)
    mask = torch.empty((B, Lq), dtype=torch.float32, device=noise.device)
    L.check(L.load().segclip_mask_sort(L.ptr(noise), L.ptr(ids_shuffle), L.ptr(ids_restore), L.ptr(mask), B, Lq,
                                       len_keep, L.stream()), "mask_sort")
    return ids_shuffle, ids_restore, mask


class _on_comm_stream:
    """The embedding exchange runs on the communication stream (the one GradSync's bucket all-reduces use), ordered
    against the compute stream by events on both sides instead of being enqueued in-line on it: RCCL's launch-side
    bookkeeping then never sits between two compute kernels, and the collective keeps its own hardware queue."""

    def __init__(self, t):
        self.t = t

    def __enter__(self):
        if not self.t.is_cuda:      # CPU tensors (gloo tests): nothing to order
            self.ctx = None
            return
        from . import streams
        self.main = torch.cuda.current_stream(self.t.device)
        self.comm = streams.side_stream("comm", self.main)
        self.comm.wait_stream(self.main)
        self.ctx = torch.cuda.stream(self.comm)
        self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is None:
            return False
        self.ctx.__exit__(*exc)
        self.main.wait_stream(self.comm)   # the consumer (logits GEMM) follows at once: no allocator hand-over needed
        return False


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 0


def _gather_raw(x, world):
    """rank-ordered all-gather of (B, ...) -> (B * world, ...), no autograd.  world 0 (no process group): identity."""
    if world == 0:
        return x
    x = x.contiguous()
    if x.is_cuda and dist.get_backend() != "nccl":
        # gloo has no device all-gather (only broadcast / all-reduce take GPU tensors): the single-GPU
        # multi-process tests gather by summing rank-disjoint slices, which is exact
        out = torch.zeros((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        r = dist.get_rank()
        out[r * x.shape[0]:(r + 1) * x.shape[0]] = x
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
        return out
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    with _on_comm_stream(x):
        dist.all_gather_into_tensor(out, x)
    return out


def _scatter_sum_raw(g, world):
    """the adjoint: reduce-scatter(SUM) of (B * world, ...) -> this rank's (B, ...), no autograd"""
    if world == 0:
        return g
    g = g.contiguous()
    n = g.shape[0] // world
    if dist.get_backend() == "nccl":
        out = torch.empty((n,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        with _on_comm_stream(g):
            dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM)
        return out
    g = g.clone()
    dist.all_reduce(g, op=dist.ReduceOp.SUM)
    r = dist.get_rank()
    return g[r * n:(r + 1) * n].contiguous()


class AllGatherFn(Function):
    """Differentiable rank-ordered all-gather of (B, ...) embeddings (dist_collect,
    modules/util_module.py:180-190; diffdist semantics: all_gather forward, reduce-scatter(SUM)
    backward).  On RCCL ("nccl" backend) both directions are ONE fused collective; the gloo branch
    (CPU tests) uses all_reduce + own slice, which is the same arithmetic.  World size 1 = identity."""

    @staticmethod
    def forward(ctx, x):
        ctx.world = _world()
        return _gather_raw(x, ctx.world)

    @staticmethod
    def backward(ctx, g):
        return _scatter_sum_raw(g, ctx.world)


def all_gather_embeddings(x):
    return AllGatherFn.apply(x)


class MaxTokensFn(Function):
    """cls = max over the patch tokens (modules/module_seg_vit.py:441): x (B, T, D) fp32 -> (B, D).  The backward writes the
    routed gradient as fp32 AND (bf16 mode) as the bf16 copy the residual stack's backward takes as its operand, in one pass
    (torch: zero-fill + scatter, then a cast)."""

    @staticmethod
    def forward(ctx, x, want_bf16):
        x = x.contiguous()
        B, T, D = x.shape
        L.require_cuda(x)
        out = _empty((B, D), torch.float32, x)
        idx = torch.empty((B, D), dtype=torch.int32, device=x.device)
        L.check(L.load().segclip_max_tokens_fwd(L.ptr(x), L.ptr(out), L.ptr(idx), B, T, D, L.stream()), "max_tokens_fwd")
        ctx.save_for_backward(idx)
        ctx.shape, ctx.want_bf16 = (B, T, D), bool(want_bf16)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        B, T, D = ctx.shape
        g = g.contiguous().float()
        dx = _empty((B, T, D), torch.float32, g)
        dx16 = _empty((B, T, D), torch.bfloat16, g) if ctx.want_bf16 else None
        L.check(L.load().segclip_max_tokens_bwd(L.ptr(g), L.ptr(idx), L.ptr(dx), L.ptr(dx16), B, T, D, L.stream()), "max_tokens_bwd")
        if dx16 is not None:
            dx._segclip_bf16 = dx16
        return dx, None


class ClipLossFn(Function):
    """The contrastive head of the training step as ONE autograd node (modules/modeling.py:338-362 + :204-209):
    L2-normalise both feature matrices, all-gather them in one stacked message, the two logits matrices
    clamp(exp(logit_scale), 100) * (t v_all^T), (v t_all^T) in exact fp32, both cross entropies and their mean - forward in
    4 launches (+ the collective), backward in 5.  Returns the loss; the raw cosine matrices are kept on the context object
    handed back through `box` (for model.last_logits)."""

    @staticmethod
    def forward(ctx, v_feat, t_feat, logit_scale, rank, box):
        lib = L.load()
        v_feat, t_feat = v_feat.contiguous().float(), t_feat.contiguous().float()
        L.require_cuda(v_feat, t_feat, logit_scale)
        B, C = v_feat.shape
        world = _world()
        both = _empty((B, 2, C), torch.float32, v_feat)
        norms = _empty((2 * B,), torch.float32, v_feat)
        L.check(lib.segclip_l2norm_pair_fwd(L.ptr(v_feat), L.ptr(t_feat), L.ptr(both), L.ptr(norms), B, C, L.stream()), "l2norm_pair_fwd")
        allb = _gather_raw(both, world)                       # (N, 2, C): [:, 0] = visual, [:, 1] = text of every rank
        N = allb.shape[0]
        off = B * int(rank)
        if off + B > N:
            raise ValueError(f"contrastive loss: rank {rank} with per-rank batch {B} does not fit the gathered batch {N}")
        cosm = _empty((2, B, N), torch.float32, v_feat)
        # [0] = t . v_all^T, [1] = v . t_all^T: one batched exact-fp32 launch (batch strides -C / +C swap the roles)
        p_gemm(both, allb, cosm, B, N, C, (2 * C, 1), (2 * C, 1), N, a_off=C, nb1=2, bsA=(-C, 0), bsB=(C, 0), bsC=(B * N, 0))
        ls = logit_scale.detach()
        lse, loss_rows = _empty((2 * B,), torch.float32, v_feat), _empty((2 * B,), torch.float32, v_feat)
        loss = _empty((), torch.float32, v_feat)
        L.check(lib.segclip_clip_ce_fwd(L.ptr(cosm), L.ptr(ls), L.ptr(lse), L.ptr(loss_rows), L.ptr(loss), B, N, off, L.stream()), "clip_ce_fwd")
        ctx.save_for_backward(both, allb, norms, cosm, lse, ls)
        ctx.cfg = (B, C, N, off, world, tuple(logit_scale.shape))
        if box is not None:
            box["cos"], box["logit_scale"] = cosm, ls
        return loss

    @staticmethod
    def backward(ctx, g):
        both, allb, norms, cosm, lse, ls = ctx.saved_tensors
        B, C, N, off, world, ls_shape = ctx.cfg
        lib = L.load()
        g = g.contiguous().float()
        dcos = torch.empty_like(cosm)
        ds_rows = _empty((2 * B,), torch.float32, cosm)
        dls = _empty(ls_shape, torch.float32, cosm) if ctx.needs_input_grad[2] else None
        L.check(lib.segclip_clip_ce_bwd(L.ptr(cosm), L.ptr(lse), L.ptr(ls), L.ptr(g), L.ptr(dcos), L.ptr(ds_rows), L.ptr(dls), B, N, off,
                                        L.stream()), "clip_ce_bwd")
        # this rank's rows as the LEFT factors: d t_n = dcos[0] v_all, d v_n = dcos[1] t_all  -> dboth[:, 1], dboth[:, 0]
        dboth = _empty((B, 2, C), torch.float32, cosm)
        p_gemm(dcos, allb, dboth, B, C, N, (N, 1), (1, 2 * C), 2 * C, c_off=C, nb1=2, bsA=(B * N, 0), bsB=(C, 0), bsC=(-C, 0))
        # every rank's rows as the RIGHT factors: d v_all = dcos[0]^T t_n, d t_all = dcos[1]^T v_n -> dall[:, 0], dall[:, 1]
        dall = _empty((N, 2, C), torch.float32, cosm)
        single = world <= 1          # N == B: the two contributions meet in the GEMM epilogue (residual = dboth)
        p_gemm(dcos, both, dall, N, C, B, (1, N), (1, 2 * C), 2 * C, b_off=C, nb1=2, bsA=(B * N, 0), bsB=(-C, 0), bsC=(C, 0),
               residual=dboth if single else None, ldr=2 * C if single else 0, bsR=(C, 0))
        dv, dt = _empty((B, C), torch.float32, cosm), _empty((B, C), torch.float32, cosm)
        if single:
            L.check(lib.segclip_l2norm_pair_bwd(L.ptr(dall), None, L.ptr(both), L.ptr(norms), L.ptr(dv), L.ptr(dt), B, C, L.stream()),
                    "l2norm_pair_bwd")
        else:
            dloc = _scatter_sum_raw(dall, world)              # reduce-scatter(SUM): diffdist's adjoint of the all-gather
            L.check(lib.segclip_l2norm_pair_bwd(L.ptr(dboth), L.ptr(dloc), L.ptr(both), L.ptr(norms), L.ptr(dv), L.ptr(dt), B, C,
                                                L.stream()), "l2norm_pair_bwd")
        return dv, dt, dls, None, None


class SegMeanFn(Function):
    """outputs = (hard @ v) / clamp_min(hard.sum(-1), 1)  (modules/module_seg_vit.py:308-309) for the one-hot hard assignment
    of AssignFn: a segment mean by center index in one launch, and its backward (dv, and dhard through numerator and
    normaliser) in one launch.  hard (B,G,T) fp32 carries the straight-through gradient; idx (B,T) uint8 and counts (B,G) are
    AssignFn's; v (B,T,D) fp32 or bf16.  -> (B,G,D) fp32."""

    @staticmethod
    def forward(ctx, hard, idx, counts, v):
        B, G, T = hard.shape
        D = v.shape[2]
        v = v.contiguous()
        L.require_cuda(hard, idx, counts, v)
        out = _empty((B, G, D), torch.float32, v)
        L.check(L.load().segclip_segmean_fwd(L.ptr(idx), L.ptr(v), L.dt(v), L.ptr(counts), L.ptr(out), B, G, T, D, L.stream()), "segmean_fwd")
        ctx.save_for_backward(idx, counts, v, out)
        ctx.shape = (B, G, T, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        idx, counts, v, out = ctx.saved_tensors
        B, G, T, D = ctx.shape
        dout = dout.contiguous().float()
        dv = torch.empty_like(v)
        dhard = _empty((B, G, T), torch.float32, v)
        L.check(L.load().segclip_segmean_bwd(L.ptr(dout), L.ptr(out), L.ptr(idx), L.ptr(v), L.dt(v), L.ptr(counts), L.ptr(dv), L.ptr(dhard),
                                             B, G, T, D, L.stream()), "segmean_bwd")
        return dhard, None, None, dv


class CenterLogitsFn(Function):
    """Assignment logits of the learnable-center stage, attn = q k^T un-scaled (modules/module_seg_vit.py:304): q (B,G,D), k (B,T,D),
    both fp32 -> (B,G,T) fp32, and dq = dattn k, dk = dattn^T q, as per-sample token loops (csrc/center.hip) instead of batched
    M = 8 products on 32-row MFMA tiles (3 launches of 110-140 us).  The summation order over D is not the exact-fp32 GEMM's:
    the module uses this in bf16 mode only (the exact-f32 mode keeps the GEMM, whose argmax the parity tests pin)."""

    @staticmethod
    def forward(ctx, q, k):
        q, k = q.contiguous().float(), k.contiguous().float()
        B, G, D = q.shape
        T = k.shape[1]
        L.require_cuda(q, k)
        attn = _empty((B, G, T), torch.float32, q)
        L.check(L.load().segclip_center_logits_fwd(L.ptr(q), L.ptr(k), L.ptr(attn), B, G, T, D, L.stream()), "center_logits_fwd")
        ctx.save_for_backward(q, k)
        return attn

    @staticmethod
    def backward(ctx, dattn):
        q, k = ctx.saved_tensors
        B, G, D = q.shape
        T = k.shape[1]
        dattn = dattn.contiguous().float()
        dq, dk = torch.empty_like(q), torch.empty_like(k)
        L.check(L.load().segclip_center_logits_bwd(L.ptr(dattn), L.ptr(q), L.ptr(k), L.ptr(dq), L.ptr(dk), B, G, T, D, L.stream()),
                "center_logits_bwd")
        return dq, dk


def center_logits(q, k, exact):
    """(B,G,T) fp32 assignment logits; exact=True (the exact-f32 mode) or uncovered shapes: the exact-fp32 batched GEMM"""
    B, G, D = q.shape
    if not exact and G == 8 and D in (768, 1024) and k.shape[1] * 32 <= 60000:
        return CenterLogitsFn.apply(q, k)
    return bmm(q, k, transB=True, out_dtype=torch.float32)
