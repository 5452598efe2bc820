"""Data-parallel step over rays (emap_amd.parallel) with the gloo backend, world_size 2, on CPU.

The forward stand-in is the oracle (the product forward is HIP-only); what is tested is the sharding
arithmetic: 2 ranks x N/2 rays with the count all-reduce + one flat gradient all-reduce reproduce the
single-process step on the N-ray batch (parameters after the Adam step agree to fp32 summation noise)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, net_state


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _make(name="d4w128L10"):
    import emap_amd
    kw, state = net_state(name)
    net = emap_amd.UDFNetwork(**kw)
    net.load_state_dict(state)
    dev = emap_amd.SingleVarianceNetwork(0.3)
    bet = emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False)
    return kw, net, dev, bet


def _oracle_render_fn(kw, net, dev, bet, rays, car=1.0, fs=0.9):
    """render() stand-in on CPU: oracle forward, differentiable w.r.t. the modules' parameters."""
    from oracle import emap_oracle as O
    cfg = O.UDFConfig(d_hidden=kw["d_hidden"], n_layers=kw["n_layers"], multires=kw["multires"])
    rcfg = O.RenderConfig(32, 32, 4)

    def fn():
        state = dict(net.named_parameters())
        return O.render(state, cfg, rcfg, *rays, dev.variance, bet.beta, bet.gamma, cos_anneal_ratio=car,
                        flip_saturation=fs, differentiable=True)
    return fn


def _run_step(rank, world, port, n_global, out_q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from emap_amd import synthetic
    from emap_amd.parallel import training_step, shard, GradBucket
    torch.set_num_threads(2)
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    kw, net, dev, bet = _make()
    params = list(net.parameters()) + [dev.variance, bet.beta, bet.gamma]
    opt = torch.optim.Adam([{"params": list(net.parameters()), "lr": 1e-3}, {"params": [dev.variance, bet.beta, bet.gamma]}], lr=5e-3)
    rays = synthetic.make_rays(n_global, seed=77)
    true_edge = synthetic.make_true_edge(n_global, seed=78)
    rays_l = [shard(t_, rank, world) for t_ in rays]
    te_l = shard(true_edge, rank, world)
    fn = _oracle_render_fn(kw, net, dev, bet, rays_l)
    loss, edge_loss = training_step(fn, te_l, params, opt, edge_weight=1.0, igr_weight=0.1, igr_ns_weight=0.05,
                                    bucket=GradBucket(params), n_rays_global=n_global)
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    gflat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    out_q.put((rank, float(loss), float(edge_loss), flat.numpy(), gflat.numpy()))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _launch(world, n_global):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_step, args=(r, world, port, n_global, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


@pytest.mark.timeout(600)
def test_two_rank_step_equals_single_process_step():
    import numpy as np
    single = _launch(1, 16)[0]
    two = _launch(2, 16)
    # both ranks hold identical parameters and the global loss
    assert np.array_equal(two[0][3], two[1][3])
    assert two[0][1] == pytest.approx(two[1][1], rel=1e-6)
    # and they match the single-process step on the same 16 rays
    assert two[0][1] == pytest.approx(single[1], rel=2e-5)
    assert two[0][2] == pytest.approx(single[2], rel=2e-5)
    g1, g2 = single[4], two[0][4]
    assert np.abs(g1 - g2).max() <= 2e-4 * np.abs(g1).max() + 1e-9
    p1, p2 = single[3], two[0][3]
    assert np.abs(p1 - p2).max() <= 1e-5


def test_shard_and_bucket_single_process():
    from emap_amd.parallel import shard, GradBucket
    t = torch.arange(24.).reshape(8, 3)
    assert torch.equal(shard(t, 1, 4), t[2:4])
    ps = [torch.nn.Parameter(torch.randn(3, 2)), torch.nn.Parameter(torch.randn(5))]
    ps[0].grad = torch.ones(3, 2)
    b = GradBucket(ps)
    b.all_reduce()
    assert b.numel == 11 and torch.equal(ps[0].grad, torch.ones(3, 2)) and torch.equal(ps[1].grad, torch.zeros(5))
