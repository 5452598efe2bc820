"""Data-parallel step over rays (emap_amd.parallel) with the gloo backend, world_size 2, on CPU.

The product forward/backward are HIP-only, so on the CPU the two device stages of ``Trainer`` are substituted by the
oracle (forward) and the algorithm mirror of the backward kernels (oracle/vjp_mirror.py); everything else - flat parameter /
gradient buffers, the statistics exchange, global eikonal denominators, the single gradient all-reduce, the two-group Adam - is
the product code.  What is tested: 2 ranks x N/2 rays reproduce the single-process step on the N-ray batch.
"""
import os
import socket
import sys

import numpy as np
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


def _oracle_trainer(kw, net, dev, bet, eikonal_sync):
    """emap_amd.parallel.Trainer with its two HIP stages replaced by the oracle and the backward mirror."""
    import emap_amd
    from emap_amd.parallel import Trainer
    from oracle import emap_oracle as O
    from oracle import vjp_mirror as M
    cfg = O.UDFConfig(d_hidden=kw["d_hidden"], n_layers=kw["n_layers"], multires=kw["multires"])
    rcfg = O.RenderConfig(32, 32, 4)
    r = emap_amd.UDFRendererBlending(None, net, dev, bet, 32, 32, 0, 4, 1.0, device="cpu")

    class OracleTrainer(Trainer):
        def _forward(self, rays):
            with torch.no_grad():
                state = {k: v.detach() for k, v in net.named_parameters()}
                out = O.render(state, cfg, rcfg, rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], rays["depth_scale"],
                               dev.variance.detach(), bet.beta.detach(), bet.gamma.detach(), cos_anneal_ratio=rays["cos_anneal_ratio"],
                               flip_saturation=rays["flip_saturation"])
            sc = torch.zeros(16)
            sc[3:7] = out["eikonal_sums"]
            return rays, out, out["edge"].reshape(-1), sc

        def _backward(self, rays, out, d_edge, sc, flat_grad):
            dt = torch.float64
            state = {k: v.detach().to(dt) for k, v in net.named_parameters()}
            z = out["z_vals"].to(dt)
            N, S = z.shape
            ro, rd = rays["rays_o"].to(dt), rays["rays_d"].to(dt)
            sd = float(((rays["far"] - rays["near"]) / 32).mean())
            var, bp, gp = dev.variance.detach().to(dt), bet.beta.detach().to(dt), bet.gamma.detach().to(dt)
            inv_s, beta, gamma = O.inv_s_from_variance(var), O.beta_from_param(bp), O.gamma_from_param(gp)
            dU, dG, dis, dbt, dgm = M.composite_bwd(ro, rd, z, sd, out["udf"].to(dt), out["gradients"].to(dt), inv_s, beta, gamma,
                                                    rays["cos_anneal_ratio"], rays["flip_saturation"], rcfg.near_surface, None,
                                                    d_edge.to(dt).view(N, 1), None, None, float(self._igr) / (float(sc[4]) + 1e-5),
                                                    float(self._igr_ns) / (float(sc[6]) + 1e-5) if self.igr_ns_weight else 0.0)
            pts = (ro[:, None, :] + rd[:, None, :] * out["mid_z_vals"].to(dt)[..., None]).reshape(-1, 3)
            got, _ = M.mlp_vjp(state, cfg, pts, dU.reshape(-1), dG.reshape(-1, 3))
            lay = self.r._layout()
            flat_grad.zero_()
            for l in range(cfg.n_lin):
                gk, vk, bk = (f"lin{l}.parametrizations.weight.original0", f"lin{l}.parametrizations.weight.original1", f"lin{l}.bias")
                d_g, d_v = M.weight_norm_vjp(state[gk], state[vk], got[f"lin{l}.weight"])
                named = dict(net.named_parameters())
                for key, val in ((gk, d_g), (vk, d_v), (bk, got[f"lin{l}.bias"])):
                    o = lay.offsets[id(named[key])]
                    flat_grad[o:o + val.numel()] = val.reshape(-1).float()
            for p, val in ((dev.variance, dis * 10 * inv_s), (bet.beta, dbt * 10 * beta), (bet.gamma, dgm * 10 * gamma)):
                flat_grad[lay.offsets[id(p)]] = float(val)

    return OracleTrainer(r, lr_geo=1e-3, lr=5e-3, edge_weight=1.0, igr_weight=0.1, igr_ns_weight=0.05, eikonal_sync=eikonal_sync,
                         fused_adam=False)


def _run_step(rank, world, port, n_global, mode, out_q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from emap_amd import synthetic
    from emap_amd.parallel import training_step, shard, FlatParams
    torch.set_num_threads(2)
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    kw, net, dev, bet = _make()
    rays = synthetic.make_rays(n_global, seed=77)
    true_edge = synthetic.make_true_edge(n_global, seed=78)
    rays_l = [shard(t_, rank, world) for t_ in rays]
    te_l = shard(true_edge, rank, world)
    params = list(net.parameters()) + [dev.variance, bet.beta, bet.gamma]
    if mode == "autograd":
        flat = FlatParams(params)
        opt = torch.optim.Adam([{"params": list(net.parameters()), "lr": 1e-3}, {"params": [dev.variance, bet.beta, bet.gamma]}], lr=5e-3)
        fn = _oracle_render_fn(kw, net, dev, bet, rays_l)
        loss, edge_loss = training_step(fn, te_l, flat, opt, edge_weight=1.0, igr_weight=0.1, igr_ns_weight=0.05, n_rays_global=n_global)
        gflat = flat.grad.clone()
    else:
        tr = _oracle_trainer(kw, net, dev, bet, mode)
        d = dict(zip(("rays_o", "rays_d", "near", "far", "depth_scale"), rays_l))
        d.update(cos_anneal_ratio=1.0, flip_saturation=0.9)
        loss, edge_loss = tr.step(d, te_l, n_rays_global=n_global)
        gflat = tr.flat.grad[:tr.flat.numel].clone()
    pflat = torch.cat([p.detach().reshape(-1) for p in params])
    out_q.put((rank, float(loss), float(edge_loss), pflat.numpy(), gflat.numpy()))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _launch(world, n_global, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_step, args=(r, world, port, n_global, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["autograd", "exact", "exact_lagged"])      # exact_lagged on the host path = exact (no fp16 range scale there)
def test_two_rank_step_equals_single_process_step(mode):
    single = _launch(1, 16, mode)[0]
    two = _launch(2, 16, mode)
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


@pytest.mark.timeout(900)
def test_native_and_autograd_steps_agree_and_local_sync_is_close():
    """The native Trainer (hand-derived backward) and the autograd step produce the same gradients and parameters; the
    one-collective variant (rank-local eikonal denominators) differs only by the mean-of-means bias."""
    auto = _launch(1, 16, "autograd")[0]
    nat = _launch(1, 16, "exact")[0]
    assert nat[1] == pytest.approx(auto[1], rel=2e-5)
    assert np.abs(nat[4] - auto[4]).max() <= 2e-4 * np.abs(auto[4]).max() + 1e-9
    assert np.abs(nat[3] - auto[3]).max() <= 1e-5
    loc = _launch(2, 16, "local")
    assert np.array_equal(loc[0][3], loc[1][3])
    cos = float((loc[0][4] * nat[4]).sum() / (np.linalg.norm(loc[0][4]) * np.linalg.norm(nat[4])))
    assert cos > 0.99, cos   # 8 rays per rank: the near-surface mask sums of the two ranks differ a lot; the bias shrinks with batch size


def test_shard_and_flat_params_single_process():
    from emap_amd.parallel import shard, FlatParams
    t = torch.arange(24.).reshape(8, 3)
    assert torch.equal(shard(t, 1, 4), t[2:4])
    ps = [torch.nn.Parameter(torch.randn(3, 2)), torch.nn.Parameter(torch.randn(5))]
    before = [p.detach().clone() for p in ps]
    f = FlatParams(ps, extra=4)
    assert f.numel == 11 and f.grad.numel() == 15
    assert all(torch.equal(p.detach(), b) for p, b in zip(ps, before))
    f.data.add_(1.0)                       # one update of the flat buffer moves every parameter
    assert all(torch.equal(p.detach(), b + 1.0) for p, b in zip(ps, before))
    f.grad[:6] = 2.0
    assert torch.equal(ps[0].grad, torch.full((3, 2), 2.0)) and torch.equal(ps[1].grad, torch.zeros(5))
    assert f.span(ps) == (0, 11)


# ---------------------------------------------------------------------------------------- exact_lagged's per-rank slots at world 8
def _run_lagged_slots(rank, world, port, out_q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    kw, net, dev, bet = _make()
    tr = _oracle_trainer(kw, net, dev, bet, "exact_lagged")
    assert tr.collectives_per_step == 2 and tr._maxima_tail().numel() == 2 * tr.MAX_RANKS
    n = tr.flat.numel
    # own range maxima of rank r at step s: a smooth history, then a 100x jump on rank 5 (step 3) and a 1000x collapse everywhere (step 4)
    def own(r, s):
        base = torch.tensor([1.0 + 0.25 * r, 0.5 + 0.125 * ((r * 3) % world)]) * (1.0 + 0.1 * s)
        if s == 3 and r == 5:
            base = base * 100.0
        if s == 4:
            base = base * 1e-3
        return base
    used, hist, sums = [], [], []
    for s in range(5):
        cur = own(rank, s).clone()
        tr.flat.grad.zero_()
        tr._lag_publish(cur, rank)                                  # slots <- own maxima; cur <- the scale this step's sweep would use
        if not tr._lag_valid:                                       # first step: Trainer._collectives() runs the MAX all-reduce instead
            dist.all_reduce(cur, op=dist.ReduceOp.MAX)
            tr._lag.copy_(cur)
        used.append(cur.clone())
        tr.flat.grad[:n] = float(rank + 1) * (s + 1)                # "the gradients": the bucket's SUM must leave the slots intact
        dist.all_reduce(tr.flat.grad, op=dist.ReduceOp.SUM)         # ONE collective: gradients + statistics + every rank's two maxima
        sums.append(float(tr.flat.grad[0]))
        tr._lag_collect()
        hist.append(tr._lag.clone())
    out_q.put((rank, torch.stack(used).numpy(), torch.stack(hist).numpy(), sums))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_exact_lagged_rank_slots_with_eight_ranks():
    """the per-rank maxima slots in the gradient bucket's tail (Trainer._lag_publish / _lag_collect, the
    eikonal_sync='exact_lagged' mode) had only ever run with 2 ranks.  8 gloo ranks, 5 steps with a 100x jump on one rank and a 1000x
    collapse on all: every rank derives the same global history from the ONE bucket all-reduce, the scale it uses is
    clamp(4 x previous global maxima, own, 16 x own), and the gradient part of the bucket is the plain sum."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_lagged_slots, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0

    def own(r, s):
        base = np.array([1.0 + 0.25 * r, 0.5 + 0.125 * ((r * 3) % world)], np.float32) * np.float32(1.0 + 0.1 * s)
        if s == 3 and r == 5:
            base = base * np.float32(100.0)
        if s == 4:
            base = base * np.float32(1e-3)
        return base
    glob = np.stack([np.max(np.stack([own(r, s) for r in range(world)]), axis=0) for s in range(5)])
    for rank, used, hist, sums in res:
        # step 0 seeds the history from the MAX all-reduce; from step 1 on the history is the max over the gathered slots of that step
        assert np.allclose(hist[0], glob[0], rtol=1e-6)
        for s in range(1, 5):
            assert np.allclose(hist[s], glob[s], rtol=1e-6), (rank, s, hist[s], glob[s])
            o = own(rank, s)
            expect = np.minimum(np.maximum(4.0 * glob[s - 1], o), 16.0 * o)
            assert np.allclose(used[s], expect, rtol=1e-6), (rank, s, used[s], expect)
        # inside the clamp every rank uses the same number (step 1, 2); rank 5's own jump lifts only ITS scale at step 3
        assert np.allclose(used[1], 4.0 * glob[0], rtol=1e-6) and np.allclose(used[2], 4.0 * glob[1], rtol=1e-6)
        if rank == 5:
            assert np.all(used[3] > 4.0 * glob[2])
        assert np.all(used[4] <= 16.0 * own(rank, 4) * (1 + 1e-6))          # the collapse: capped at 16 x own, not 4 x a stale history
        assert sums == [float(sum(range(1, world + 1)) * (s + 1)) for s in range(5)]
    assert all(np.array_equal(res[0][2], r[2]) for r in res)                  # identical history on every rank
