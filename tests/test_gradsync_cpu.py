"""CPU, world_size 2 over gloo: segclip_amd.dist.GradSync, the gradient exchange of the data-parallel step
(the reference uses DistributedDataParallel, main_task_align.py:251-252).  Checked against a hand-made
average of the per-rank gradients and against DDP's own result on the same toy module: averaged gradients,
unused parameters keep grad None, 256-byte aligned gradient storage, gradient accumulation, no_sync()."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(7)
        self.a = nn.Parameter(torch.randn(5, 3, generator=g))
        self.scale = nn.Parameter(torch.ones(()))          # 1-element tensor in front of others
        self.b = nn.Parameter(torch.randn(7, 5, generator=g))
        self.unused = nn.Parameter(torch.randn(4, generator=g))
        self.frozen = nn.Parameter(torch.randn(3, generator=g), requires_grad=False)

    def forward(self, x):
        return ((x @ self.a.t()) @ self.b.t() * self.scale).pow(2).mean() + self.frozen.sum() * 0


def _local_grads(rank, step):
    m = Toy()
    x = torch.randn(6, 3, generator=torch.Generator().manual_seed(100 * step + rank))
    m(x).backward()
    return {n: (p.grad.clone() if p.grad is not None else None) for n, p in m.named_parameters()}


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from segclip_amd.dist import GradSync
    net = GradSync(Toy(), bucket_mb=64 * 4 / (1 << 20))  # 64-element buckets: several buckets even for the toy
    assert hasattr(net, "module") and isinstance(net.module, Toy)

    def expect(step):
        per_rank = [_local_grads(r, step) for r in range(world)]
        return {n: (sum(g[n] for g in per_rank) / world if per_rank[0][n] is not None else None) for n in per_rank[0]}

    for step in range(3):   # step 0 builds the layout, steps 1-2 run the bucketed steady state
        net.zero_grad(set_to_none=True)
        x = torch.randn(6, 3, generator=torch.Generator().manual_seed(100 * step + rank))
        net(x).backward()
        e = expect(step)
        for n, p in net.module.named_parameters():
            if e[n] is None:
                assert p.grad is None, n
                continue
            assert torch.allclose(p.grad, e[n], rtol=1e-6, atol=1e-7), (step, n)
            assert p.grad.data_ptr() % 64 == 0, (n, p.grad.data_ptr() % 64)  # slots are 256-B multiples from the (64-B aligned CPU) base
            assert p.grad.is_contiguous() and p.grad.dtype == torch.float32
    assert len(net._flat) >= 2 and net.stats["buckets"] >= 3 * len(net._flat) - len(net._flat)
    # gradient accumulation (two backward passes, exchange after each, like DDP without no_sync)
    net.zero_grad(set_to_none=True)
    for step in (5, 6):
        x = torch.randn(6, 3, generator=torch.Generator().manual_seed(100 * step + rank))
        net(x).backward()
    e5, e6 = expect(5), expect(6)
    for n, p in net.module.named_parameters():
        if e5[n] is not None:
            assert torch.allclose(p.grad, e5[n] + e6[n], rtol=1e-5, atol=1e-6), n
    # no_sync: local gradients only; the next synchronised backward exchanges the sum
    net.zero_grad(set_to_none=True)
    with net.no_sync():
        x = torch.randn(6, 3, generator=torch.Generator().manual_seed(100 * 7 + rank))
        net(x).backward()
    loc = _local_grads(rank, 7)
    for n, p in net.module.named_parameters():
        if loc[n] is not None:
            assert torch.allclose(p.grad, loc[n], rtol=1e-6, atol=1e-7), n
    x = torch.randn(6, 3, generator=torch.Generator().manual_seed(100 * 8 + rank))
    net(x).backward()
    e7, e8 = expect(7), expect(8)
    for n, p in net.module.named_parameters():
        if e7[n] is not None:
            assert torch.allclose(p.grad, e7[n] + e8[n], rtol=1e-5, atol=1e-6), n
    # same numbers as torch's DistributedDataParallel on the same module
    ddp = nn.parallel.DistributedDataParallel(Toy(), find_unused_parameters=True)
    x = torch.randn(6, 3, generator=torch.Generator().manual_seed(100 * 1 + rank))
    ddp(x).backward()
    e = expect(1)
    for n, p in ddp.module.named_parameters():
        if e[n] is not None and p.grad is not None:
            assert torch.allclose(p.grad, e[n], rtol=1e-6, atol=1e-7), n
    torch.save(True, f"{out}.{rank}")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_gradsync_matches_manual_average_and_ddp(world, tmp_path):
    out = str(tmp_path / "ok")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(os.path.exists(f"{out}.{r}") for r in range(world))


def test_gradsync_without_process_group_is_a_plain_wrapper():
    from segclip_amd.dist import GradSync
    net = GradSync(Toy())
    for step in range(2):
        net.zero_grad(set_to_none=True)
        x = torch.randn(6, 3, generator=torch.Generator().manual_seed(step))
        net(x).backward()
        loc = _local_grads(0, 0) if False else None
    m = Toy()
    m(x).backward()
    for (n, p), (_, q) in zip(net.module.named_parameters(), m.named_parameters()):
        assert (p.grad is None) == (q.grad is None)
        if q.grad is not None:
            assert torch.equal(p.grad, q.grad) and p.grad.data_ptr() % 64 == 0


class Toy2(Toy):
    """Toy with a buffer; rank r perturbs its weights before wrapping (per-rank seed / partial checkpoint load)."""

    def __init__(self, rank):
        super().__init__()
        self.register_buffer("running", torch.full((3,), float(rank)))
        with torch.no_grad():
            self.a.add_(0.5 * rank)
            self.scale.fill_(1.0 + rank)


def _worker_init(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from segclip_amd.dist import GradSync
    m = Toy2(rank)
    m.a._segclip_shadow = ["stale", -1]          # a bf16 shadow of the pre-broadcast value must not survive
    net = GradSync(m)
    ref = Toy2(0)
    for (n, p), (_, q) in zip(net.module.named_parameters(), ref.named_parameters()):
        assert torch.equal(p.detach(), q.detach()), (rank, n)     # DDP-constructor semantics: rank 0's state everywhere
    assert torch.equal(net.module.running, ref.running)
    assert not hasattr(net.module.a, "_segclip_shadow")
    # broadcast_init=False keeps the local state (the caller's responsibility, e.g. identical seeds)
    keep = GradSync(Toy2(rank), broadcast_init=False)
    assert float(keep.module.scale) == 1.0 + rank
    keep.remove()
    # ranks that disagree on which parameters receive gradients: EVERY rank raises (no rank walks into a collective alone)
    class Branchy(Toy):
        def forward(self, x):
            y = super().forward(x)
            return y + (self.unused.sum() if dist.get_rank() == 1 else 0.0)
    bad = GradSync(Branchy())
    x = torch.randn(6, 3, generator=torch.Generator().manual_seed(rank))
    raised = False
    try:
        bad(x).backward()
    except RuntimeError as e:
        raised = "disagree" in str(e)
    assert raised, rank
    torch.save(True, f"{out}.{rank}")
    dist.destroy_process_group()


def test_gradsync_broadcasts_initial_state_and_raises_on_every_rank(tmp_path):
    out = str(tmp_path / "ok")
    mp.spawn(_worker_init, args=(2, _free_port(), out), nprocs=2, join=True)
    assert all(os.path.exists(f"{out}.{r}") for r in range(2))


class LateToy(nn.Module):
    """`late` joins the graph only when use_late is set: a data-dependent graph after the layout was frozen."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(9)
        self.a = nn.Parameter(torch.randn(4, 3, generator=g))
        self.late = nn.Parameter(torch.randn(3, generator=g))
        self.use_late = False

    def forward(self, x):
        y = (x @ self.a.t()).pow(2).mean()
        return y + (x * self.late).sum() if self.use_late else y


def _verdict_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from segclip_amd import dist as sd
    net = sd.GradSync(LateToy())
    x = torch.randn(6, 3, generator=torch.Generator().manual_seed(rank))
    for step in range(6):   # step 0 builds the layout; the agreement all-reduce runs in the first _CHECK_PASSES steady passes only
        net.zero_grad(set_to_none=True)
        net(x).backward()
    assert net.stats["verdicts"] == sd._CHECK_PASSES, net.stats
    # steady state: a late gradient raises on the rank that sees it instead of walking into an unmatched collective
    net.module.use_late = True
    net.zero_grad(set_to_none=True)
    try:
        net(x).backward()
        raised = False
    except RuntimeError as e:
        raised = "after the bucket layout was frozen" in str(e)
    assert raised
    torch.save(True, f"{out}.{rank}")
    dist.destroy_process_group()


def test_gradsync_late_gradient_agreement_only_in_the_first_passes(tmp_path):
    """no all-reduce + blocking host read per backward in the steady state."""
    out = str(tmp_path / "ok")
    mp.spawn(_verdict_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert all(os.path.exists(f"{out}.{r}") for r in range(2))
