"""Data-parallel gradient exchange of the hot path: the counterpart of the reference's
`torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)` wrapper
(main_task_align.py:251-252), built for how THIS backward produces gradients.

Why not DistributedDataParallel: every residual block here is ONE autograd node (ops.ResBlockFn) whose
parameter gradients become ready together, so DDP's per-parameter reducer hooks, bucket copies and
unused-parameter bookkeeping are pure host/device overhead (+4.9 ms per step on one rank, round-1
measurement), and its gradient_as_bucket_view packs a 1-element tensor (clip.logit_scale) in front of
others, leaving their gradients 4-byte aligned - the fused optimizer needs 16.

GradSync keeps one flat fp32 buffer per bucket; every trainable parameter owns a 256-byte aligned slot in
it and `p.grad` IS a view of that slot.  The weight-gradient GEMMs write straight into the slots
(ops.grad_slot_out), so a gradient is never copied.  A bucket whose slots are all written is exchanged
at once: [cast fp32->bf16] -> RCCL all-reduce(AVG) -> [cast back], on a communication stream, overlapping
the rest of the backward; the end-of-backward callback makes the compute stream wait for the last one.
Buckets are laid out in the order gradients became ready in the first backward (rank 0's order,
broadcast), and are always launched in index order, so every rank issues the same collective sequence.

Semantics = DDP's: gradients are averaged over ranks after every backward (also under gradient
accumulation, where the bucket holds previous-average + new-local, as with DDP); parameters that
receive no gradient keep `grad is None` (find_unused_parameters=True behaviour).
Like DDP's constructor, GradSync broadcasts rank 0's parameters and buffers to every rank when it wraps the
module, so ranks that were built with different seeds / partially loaded checkpoints start from one state.
Wire format: fp32 by default (what the reference's DDP exchanges).  compress=True (or "auto" in bf16 mode) is an
explicit opt-in to a bf16 wire (halves the xGMI bytes: ViT-B/16 627 MB -> 313 MB per
step): every gradient is then rounded to 8 mantissa bits before the cross-rank sum, so results depend on the world
size beyond fp32 rounding; a bucket that holds accumulated gradients (no_sync micro-steps) is always
exchanged in fp32 so that the already-averaged part is not rounded again.
"""
import os
import weakref

import torch
import torch.distributed as dist
from torch import nn

_ALIGN = 64  # slot alignment in fp32 elements (256 bytes)
_NO_EXCHANGE = bool(int(os.environ.get("SEGCLIP_GRADSYNC_NOEXCHANGE", "0")))  # diagnosis: hooks + slots, no collective
# The cross-rank agreement on late gradients (a parameter that starts receiving a gradient after the bucket layout was
# frozen) costs an all-reduce + a blocking host read per backward: it runs in the first _CHECK_PASSES steady passes
# (where a data-dependent graph would show up) and, with SEGCLIP_CHECK_COLLECTIVES=1, in every pass.  Afterwards a late
# gradient raises on the rank that sees it (no host synchronisation in the steady-state path).
_CHECK_ALWAYS = bool(int(os.environ.get("SEGCLIP_CHECK_COLLECTIVES", "0")))
_CHECK_PASSES = int(os.environ.get("SEGCLIP_CHECK_COLLECTIVES_PASSES", "2"))


class _Slot:
    """A parameter's place in a flat gradient bucket."""
    __slots__ = ("bucket", "offset", "numel", "shape", "param", "owner", "taken_pass", "ptr", "fwd_pass", "fwd_uses", "reported_pass")

    def __init__(self, bucket, offset, p, owner):
        self.bucket, self.offset, self.numel, self.shape = bucket, offset, p.numel(), tuple(p.shape)
        self.taken_pass = -1
        self.fwd_pass, self.fwd_uses = -1, 0
        self.reported_pass = -1
        self.ptr = owner._flat[bucket].data_ptr() + 4 * offset   # address of the slot (hook fast path: no view objects)
        self.param = weakref.ref(p)
        self.owner = weakref.ref(owner)

    def view(self):
        o = self.owner()
        return o._flat[self.bucket][self.offset:self.offset + self.numel].view(self.shape)

    def note_forward_use(self):
        """Called by the block forwards: counts how often the parameter is used in the current pass (a parameter used
        twice must not be published before autograd has summed both contributions)."""
        o = self.owner()
        if o is None:
            return
        if self.fwd_pass != o._pass_id:
            self.fwd_pass, self.fwd_uses = o._pass_id, 0
        self.fwd_uses += 1

    def single_use(self):
        o = self.owner()
        return o is not None and self.fwd_pass == o._pass_id and self.fwd_uses == 1

    def out_buffer(self):
        """Fresh view to be used as the OUTPUT of a gradient kernel, or None when the parameter already
        holds a gradient (accumulation: autograd must add, not alias), when the slot was already handed out in this
        backward pass (a parameter used twice - e.g. the vision tower's second, MAE pass - gets two contributions
        that autograd sums: they must not share memory), or when the owner is not ready."""
        o, p = self.owner(), self.param()
        if o is None or p is None or not o._steady or p.grad is not None or self.taken_pass == o._pass_id:
            return None
        self.taken_pass = o._pass_id
        return self.view()


class GradSync(nn.Module):
    def __init__(self, module, process_group=None, bucket_mb=64, compress=False, broadcast_init=True):
        super().__init__()
        self.module = module
        self.group = process_group
        self.bucket_elems = max(int(bucket_mb * (1 << 20) // 4), 1)
        self.compress = compress
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._params = [p for p in module.parameters() if p.requires_grad]
        self._index = {id(p): i for i, p in enumerate(self._params)}
        self._steady = False
        self._sync_enabled = True
        self._first_order = []       # parameter indices in first-backward ready order
        self._seen = set()
        self._callback_queued = False
        self._flat, self._wire, self._slots, self._bucket_params = [], [], {}, []
        self._pending, self._next_bucket, self._launched, self._late, self._ready = [], 0, [], [], []
        self._bstreams = []          # per bucket: the streams its gradients were produced on in this pass
        self._comm = None
        self._pass_id = 0
        self._steady_passes = 0      # synced backward passes since the layout was frozen
        self.stats = {"copies": 0, "zero_copy": 0, "buckets": 0, "verdicts": 0}
        self._verdicts = []          # posted, not yet read agreement checks: (event, pinned host pair)
        self.timeline = None         # set to [] to record (bucket, bytes, ready event) per exchange (tools/bucket_timeline.py)
        self._accumulated = False    # a no_sync() backward left local gradients in the buckets
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self._params]
        if broadcast_init and self.world > 1:
            self.broadcast_state()

    def broadcast_state(self, src=0):
        """DDP-constructor semantics: every rank takes rank `src`'s parameters and buffers (ONE coalesced broadcast per
        dtype/device).  bf16 weight shadows of the old values are dropped."""
        gsrc = dist.get_global_rank(self.group, src) if self.group is not None else src
        tensors = [p.data for p in self.module.parameters()] + [b.data for b in self.module.buffers()]
        groups = {}
        for t in tensors:
            groups.setdefault((t.dtype, t.device), []).append(t)
        for (dtype, dev), ts in groups.items():
            if dtype == torch.bool:
                for t in ts:
                    u = t.to(torch.uint8)
                    dist.broadcast(u, src=gsrc, group=self.group)
                    t.copy_(u.bool())
                continue
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.broadcast(flat, src=gsrc, group=self.group)
            off = 0
            for t in ts:
                n = t.numel()
                t.copy_(flat[off:off + n].view(t.shape))
                off += n
        for p in self.module.parameters():
            if hasattr(p, "_segclip_shadow"):
                del p._segclip_shadow

    def remove(self):
        """Detach from the module: drop the hooks and the parameters' slot references (gradients stay where they are)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self._params:
            if getattr(p, "_segclip_gslot", None) is not None and p._segclip_gslot.owner() is self:
                del p._segclip_gslot
        self._steady = False

    # ------------------------------------------------------------------ nn.Module plumbing
    def forward(self, *args, **kwargs):
        if torch.cuda.is_available():
            self._main = torch.cuda.current_stream()   # the caller's stream: the exchange stream must run beside it
        return self.module(*args, **kwargs)

    def no_sync(self):
        """Context manager: accumulate local gradients without exchanging them (DDP API)."""
        outer = self

        class _Ctx:
            def __enter__(self):
                self.prev, outer._sync_enabled = outer._sync_enabled, False
                outer._accumulated = True

            def __exit__(self, *exc):
                outer._sync_enabled = self.prev
        return _Ctx()

    # ------------------------------------------------------------------ backward-time hooks
    def _queue_callback(self):
        if not self._callback_queued:
            self._callback_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)

    def _on_grad(self, p):
        i = self._index[id(p)]
        self._queue_callback()
        if not self._steady:
            if i not in self._seen:
                self._seen.add(i)
                self._first_order.append(i)
            return
        slot = self._slots.get(i)
        if slot is None:   # became used after the layout was frozen: exchanged on its own at the end
            self._late.append(p)
            return
        if slot.reported_pass == self._pass_id:
            return   # published early by ops.ResStackFn; autograd still runs the accumulate hook for the (undefined) grad
        slot.reported_pass = self._pass_id
        g = p.grad
        if g.is_cuda:
            # the text tower's backward runs on its own stream: a bucket may hold gradients of several streams
            self._bstreams[slot.bucket].add(torch.cuda.current_stream())
        if g.data_ptr() != slot.ptr:
            v = slot.view()
            v.copy_(g)
            p.grad = v
            self.stats["copies"] += 1
        else:
            self.stats["zero_copy"] += 1
        b = slot.bucket
        self._pending[b] -= 1
        if self._pending[b] == 0 and self._sync_enabled:
            self._ready[b] = True
            self._launch_ready()

    def _launch_ready(self):
        while self._next_bucket < len(self._flat) and self._ready[self._next_bucket]:
            self._exchange(self._next_bucket)
            self._next_bucket += 1

    # ------------------------------------------------------------------ the collective
    def _use_bf16(self, flat):
        if self.compress is False or self.compress is None:
            return False
        if not flat.is_cuda:
            return False     # the cast kernel is a HIP kernel; CPU/gloo runs exchange fp32
        if self.compress == "auto":
            from . import config
            return config.compute_dtype == torch.bfloat16
        return bool(self.compress)

    def _exchange(self, b):
        flat = self._flat[b]
        self.stats["buckets"] += 1
        if (self.world == 1 and not dist.is_initialized()) or _NO_EXCHANGE:
            return
        nccl = dist.get_backend(self.group) == "nccl"
        if flat.is_cuda and nccl:
            from . import ops
            if self._comm is None:
                from . import streams  # a stream measured to run beside the main stream (see streams.py)
                self._comm = streams.side_stream("comm", getattr(self, "_main", None))
            self._comm.wait_stream(torch.cuda.current_stream(flat.device))
            for st in self._bstreams[b]:          # every stream that produced a gradient of this bucket
                self._comm.wait_stream(st)
            with torch.cuda.stream(self._comm):
                if self.timeline is not None:     # the moment every gradient of the bucket exists on the device
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record()
                    self.timeline.append((b, flat.numel() * 4, ev))
                if self._use_bf16(flat) and not self._accumulated:
                    wire = self._wire[b]
                    ops.p_cast_into(flat, wire)
                    dist.all_reduce(wire, op=dist.ReduceOp.AVG, group=self.group)
                    ops.p_cast_into(wire, flat)
                else:
                    dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
            self._launched.append(b)
        else:   # gloo (CPU tensors, or the single-GPU multi-process tests): no AVG op, no bf16 wire format
            if flat.is_cuda:
                cur = torch.cuda.current_stream(flat.device)
                for st in self._bstreams[b]:
                    if st != cur:
                        cur.wait_stream(st)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(self.world)

    # ------------------------------------------------------------------ end of a backward pass
    def _post_verdict(self):
        dev = self._flat[0].device if self._flat else self._params[0].device
        k = len(self._late)
        self.stats["verdicts"] += 1
        if dev.type == "cuda" and dist.get_backend(self.group) == "nccl":
            if self._comm is None:
                from . import streams
                self._comm = streams.side_stream("comm", getattr(self, "_main", None))
            self._comm.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(self._comm):
                n = torch.empty(2, dtype=torch.int64, device=dev)
                n[0].fill_(k)                 # fill kernels: no host-to-device copy, no host synchronisation
                n[1].fill_(-k)
                dist.all_reduce(n, op=dist.ReduceOp.MAX, group=self.group)
                host = torch.empty(2, dtype=torch.int64, pin_memory=True)
                host.copy_(n, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            self._verdicts.append((ev, host, n))
        else:   # gloo / CPU tensors (tests): nothing to overlap with
            n = torch.tensor([k, -k], dtype=torch.int64, device=dev)
            dist.all_reduce(n, op=dist.ReduceOp.MAX, group=self.group)
            if int(n[0]) != -int(n[1]):
                raise RuntimeError("GradSync: ranks disagree on the parameters that received late gradients")

    def check_collectives(self, block=False):
        """Read the posted cross-rank agreements that have arrived (block=True: wait for all of them).  Called at the end of
        every backward pass; raises on every rank when the ranks disagreed on the parameters that received late gradients."""
        keep = []
        for ev, host, n in self._verdicts:
            if block:
                ev.synchronize()
            if ev.query():
                if int(host[0]) != -int(host[1]):
                    self._verdicts = []
                    raise RuntimeError("GradSync: ranks disagree on the parameters that received late gradients")
            else:
                keep.append((ev, host, n))
        self._verdicts = keep

    def drain(self):
        """Read every posted cross-rank agreement (blocking).  Call before a host decision that depends on the pass having been
        exchanged consistently - a checkpoint, the end of training -: verdicts posted in the last pass are otherwise never read."""
        self.check_collectives(block=True)

    def _finalize(self):
        self._callback_queued = False
        if self._verdicts:
            self.check_collectives()
        if not self._steady:
            self._build_layout()
            return
        if self._sync_enabled:
            # buckets some of whose parameters got no gradient this pass (cannot happen with a static graph;
            # kept so that a data-dependent branch degrades to a late exchange instead of a hang)
            for b in range(self._next_bucket, len(self._flat)):
                if any(self._params[i].grad is not None for i in self._bucket_params[b]):
                    for i in self._bucket_params[b]:
                        p = self._params[i]
                        if p.grad is None:
                            self._slots[i].view().zero_()
                            p.grad = self._slots[i].view()
                self._exchange(b)
            if dist.is_initialized():   # also a 1-rank group (bench.py --force-dist): the same branch structure as N > 1
                self._steady_passes += 1
                if _CHECK_ALWAYS or self._steady_passes <= _CHECK_PASSES:
                    # the number of late exchanges must be the same on every rank (data-dependent use): the ranks agree on it
                    # collectively, and every rank raises instead of hanging in mismatched collectives.  The agreement is
                    # POSTED here and read when it has arrived (round 5; it was a blocking host read in the first passes:
                    # the device idled at the start of the next step while the host waited for the end of this one)
                    self._post_verdict()
                    if self._late:
                        # ... except on a rank that is about to enqueue late exchanges: it reads the verdict FIRST (blocking).
                        # A rank with no late gradient enqueues nothing extra and may read lazily; on a disagreement the rank
                        # that would have queued extra all-reduces (which pair with the other ranks' NEXT bucket exchange:
                        # a hang or silent corruption) raises here instead, the others at their next read.
                        self.check_collectives(block=True)
                elif self._late:
                    raise RuntimeError(
                        f"GradSync: {len(self._late)} parameter(s) received a gradient for the first time after the bucket "
                        "layout was frozen (data-dependent graph); run with SEGCLIP_CHECK_COLLECTIVES=1 to exchange such "
                        "gradients under a per-pass cross-rank agreement")
            for p in self._late:
                dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group)
                p.grad.div_(self.world)
        if self._comm is not None and self._launched:
            torch.cuda.current_stream(self._flat[0].device).wait_stream(self._comm)
        if self._sync_enabled:
            self._accumulated = False
        self._reset_pass()

    def _reset_pass(self):
        self._pass_id += 1
        self._pending = [len(ps) for ps in self._bucket_params]
        self._ready = [False] * len(self._flat)
        self._bstreams = [set() for _ in self._flat]
        self._next_bucket, self._launched, self._late = 0, [], []

    def _build_layout(self):
        """After the first backward: fix the bucket layout (rank 0's ready order), move the gradients into
        their slots and exchange everything once."""
        order = list(self._first_order)
        if dist.is_initialized() and self.world > 1:
            box = [order]
            dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group else 0,
                                       group=self.group)
            order = box[0]
            # the verdict is taken collectively, so that EVERY rank raises (rank 0 always agrees with its own order and
            # would otherwise walk into the bucket all-reduces alone: a hang until the RCCL timeout)
            dev = self._params[order[0]].device if order else torch.device("cpu")
            bad = torch.tensor([int(set(order) != set(self._first_order))], dtype=torch.int32, device=dev)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=self.group)
            if int(bad):
                raise RuntimeError("GradSync: ranks disagree on the set of parameters that receive gradients")
        buckets, cur, cur_elems = [], [], 0
        for i in order:
            n = -(-self._params[i].numel() // _ALIGN) * _ALIGN
            if cur and cur_elems + n > self.bucket_elems:
                buckets.append((cur, cur_elems))
                cur, cur_elems = [], 0
            cur.append((i, cur_elems))
            cur_elems += n
        if cur:
            buckets.append((cur, cur_elems))
        self._flat, self._wire, self._slots, self._bucket_params = [], [], {}, []
        for b, (items, elems) in enumerate(buckets):
            dev = self._params[items[0][0]].device
            flat = torch.zeros(elems, dtype=torch.float32, device=dev)
            self._flat.append(flat)
            self._wire.append(torch.empty(elems, dtype=torch.bfloat16, device=dev) if flat.is_cuda else None)
            self._bucket_params.append([i for i, _ in items])
            for i, off in items:
                p = self._params[i]
                slot = _Slot(b, off, p, self)
                self._slots[i] = slot
                p._segclip_gslot = slot
                v = slot.view()
                v.copy_(p.grad)
                p.grad = v
        self._steady = True
        self._reset_pass()
        if self._sync_enabled:
            for b in range(len(self._flat)):
                self._exchange(b)
            if self._comm is not None and self._launched:
                torch.cuda.current_stream(self._flat[0].device).wait_stream(self._comm)
        self._reset_pass()

    # ------------------------------------------------------------------ introspection (tests, bench)
    def layout(self):
        return [(b, [self._params[i].shape for i in ps], int(self._flat[b].numel())) for b, ps in
                enumerate(self._bucket_params)]


def grad_slot_out(param):
    """Output buffer for `param`'s gradient inside its GradSync slot, or None (no GradSync / accumulating)."""
    slot = getattr(param, "_segclip_gslot", None)
    return slot.out_buffer() if slot is not None else None
