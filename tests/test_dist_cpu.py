"""CPU, world_size 2/4 over gloo: the N>1 exchange step of the hot path.
 (a) segclip_amd.ops.AllGatherFn (the product's dist_collect) - rank-ordered gather forward,
     reduce-scatter(SUM) backward - against the analytic result;
 (b) the oracle with that gather reproduces the per-rank golden losses/logits produced by the REAL
     reference under gloo (tests/golden/tiny_w{2,4}_t18.npz);
 (c) the 1-rank vs W-rank contrastive identity."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import FULL_FLAGS, load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _gather_worker(rank, world, port, out):
    _init(rank, world, port)
    from segclip_amd import ops
    B, E = 3, 5
    x = (torch.arange(B * E, dtype=torch.float32).reshape(B, E) + 100 * rank).requires_grad_()
    y = ops.all_gather_embeddings(x)
    assert y.shape == (world * B, E)
    for r in range(world):
        assert torch.equal(y[r * B:(r + 1) * B], torch.arange(B * E, dtype=torch.float32).reshape(B, E) + 100 * r)
    w = torch.arange(world * B * E, dtype=torch.float32).reshape(world * B, E) * (rank + 1)
    (y * w).sum().backward()
    # reduce-scatter(SUM): grad = sum over ranks of that rank's weight slice for my rows
    base = torch.arange(world * B * E, dtype=torch.float32).reshape(world * B, E)[rank * B:(rank + 1) * B]
    expect = base * sum(r + 1 for r in range(world))
    assert torch.allclose(x.grad, expect), (x.grad, expect)
    torch.save(True, f"{out}.{rank}")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_all_gather_fn_gloo(world, tmp_path):
    out = str(tmp_path / "ok")
    mp.spawn(_gather_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(os.path.exists(f"{out}.{r}") for r in range(world))


def _oracle_worker(rank, world, port, out, seed, B):
    _init(rank, world, port)
    from oracle import segclip_oracle as so
    from segclip_amd import ops, synth
    from tests.helpers import model_param_shapes, oracle_params
    spec = synth.SPECS["tiny"]
    P = oracle_params(spec, model_param_shapes(spec, FULL_FLAGS))
    gb = synth.synthetic_batch(spec, B * world, seed=seed)
    gn = synth.synthetic_noise(spec, B * world, seed=seed)
    sl = slice(rank * B, (rank + 1) * B)
    batch = {k: v[sl] for k, v in gb.items()}
    noise = {k: v[sl] for k, v in gn.items()}
    loss, aux = so.segclip_forward(batch, P, spec, noise, FULL_FLAGS, rank=rank, gather=ops.all_gather_embeddings)
    loss.backward()
    torch.save(dict(loss=loss.detach(), lc=aux["loss_contrastive"].detach(), t2v=aux["t2v"].detach(),
                    v2t=aux["v2t"].detach(), hard_idx=aux["hard_idx"],
                    gn=torch.tensor([float(p.grad.double().norm()) if p.grad is not None else -1.0 for p in P.values()]),
                    names=list(P.keys())), f"{out}.{rank}")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_oracle_matches_reference_per_rank(world, tmp_path):
    g = load_golden(f"tiny_w{world}_t18.npz")
    out = str(tmp_path / "res")
    mp.spawn(_oracle_worker, args=(world, _free_port(), out, int(g["seed"]), int(g["B"])), nprocs=world, join=True)
    names = g["grad_names"].tolist()
    for r in range(world):
        res = torch.load(f"{out}.{r}")
        assert abs(float(res["loss"]) - float(g[f"r{r}_loss"])) <= 1e-5
        assert abs(float(res["lc"]) - float(g[f"r{r}_loss_contrastive"])) <= 1e-5
        np.testing.assert_allclose(res["t2v"].numpy(), g[f"r{r}_t2v"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(res["v2t"].numpy(), g[f"r{r}_v2t"], rtol=0, atol=1e-4)
        assert np.array_equal(res["hard_idx"].numpy(), g[f"r{r}_hard_idx"])
        gn = dict(zip(res["names"], res["gn"].tolist()))
        for n, ref in zip(names, g[f"r{r}_grad_norms"]):
            assert abs(gn[n] - ref) <= 3e-4 * max(1.0, ref), (r, n, gn[n], ref)


def _identity_worker(rank, world, port, out):
    _init(rank, world, port)
    from oracle import segclip_oracle as so
    from segclip_amd import ops
    g = torch.Generator().manual_seed(5)
    t = torch.randn(4, 16, generator=g)
    v = torch.randn(4, 16, generator=g)
    B = 4 // world
    sl = slice(rank * B, (rank + 1) * B)
    ls = torch.tensor(2.0)
    t2v, v2t = so.loose_similarity(t[sl], v[sl], ls, gather=ops.all_gather_embeddings if world > 1 else None)
    torch.save(so.contrastive_loss(t2v, v2t, rank), f"{out}.{rank}")
    dist.destroy_process_group()


def test_one_rank_equals_mean_of_two_ranks(tmp_path):
    a, b = str(tmp_path / "w1"), str(tmp_path / "w2")
    mp.spawn(_identity_worker, args=(1, _free_port(), a), nprocs=1, join=True)
    mp.spawn(_identity_worker, args=(2, _free_port(), b), nprocs=2, join=True)
    one = float(torch.load(f"{a}.0"))
    two = [float(torch.load(f"{b}.{r}")) for r in range(2)]
    assert abs(one - sum(two) / 2) <= 1e-6
