"""GPU tests added in round 5 (-m gpu), all through the C ABI / the drop-in classes:

  * emap_amd.parallel.FusedAdam checkpoints in torch.optim.Adam's per-parameter layout (runner_udf.py:260,273 save / load
    `optimizer.state_dict()`): save -> load -> step equals an uninterrupted torch.optim.Adam run, in both directions.
"""
import copy

import numpy as np
import pytest
import torch

import emap_amd
from emap_amd import _lib, synthetic
from emap_amd.parallel import Trainer, FusedAdam

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _groups(gen, frozen_idx=2):
    shapes_geo = [(16, 7), (16,), (16, 1), (8, 16), (8,)]
    geo = [torch.nn.Parameter(torch.randn(*s, generator=gen).to(DEV)) for s in shapes_geo]
    tail = [torch.nn.Parameter(torch.randn(1, generator=gen).to(DEV)) for _ in range(4)]
    tail[frozen_idx].requires_grad_(False)
    return geo, tail


def _clone(ps):
    out = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    for a, b in zip(out, ps):
        a.requires_grad_(b.requires_grad)
    return out


def _mk(cls, geo, tail):
    return cls([{"params": geo, "lr": 1e-3}, {"params": tail[:2]}, {"params": tail[2:]}, {"params": []}], lr=5e-3)


def _steps(opts_params, gen, n, unfreeze_at=None, s0=0):
    for s in range(s0, s0 + n):
        gs = None
        for opt, ps in opts_params:
            if unfreeze_at is not None and s == unfreeze_at:
                ps[-2].requires_grad_(True)
            opt.zero_grad()
        shapes = [p.shape for p in opts_params[0][1]]
        gs = [torch.randn(sh, generator=gen).to(DEV) for sh in shapes]
        for opt, ps in opts_params:
            for p, g_ in zip(ps, gs):
                if p.requires_grad:
                    p.grad = g_.clone()
            opt.step()


def test_fused_adam_checkpoint_round_trip_against_torch_adam():
    """FusedAdam kept its moments in private tensors and saved an empty `state`.  Now: 3 steps, state_dict(),
    a FRESH optimizer over fresh parameter copies loads it, 3 more steps (a frozen scalar is un-frozen after the reload) -
    equal to torch.optim.Adam running the 6 steps without interruption; and the checkpoint has torch.optim.Adam's keys."""
    gen = torch.Generator().manual_seed(11)
    geo_a, tail_a = _groups(gen)
    geo_b, tail_b = _clone(geo_a), _clone(tail_a)
    oa, ob = _mk(FusedAdam, geo_a, tail_a), _mk(torch.optim.Adam, geo_b, tail_b)
    g1 = torch.Generator().manual_seed(12)
    _steps([(oa, geo_a + tail_a), (ob, geo_b + tail_b)], g1, 3)
    sd = copy.deepcopy(oa.state_dict())
    ref_sd = ob.state_dict()
    assert set(sd["state"].keys()) == set(ref_sd["state"].keys())          # the frozen scalar has no entry in either
    for k, st in sd["state"].items():
        assert set(st.keys()) >= {"step", "exp_avg", "exp_avg_sq"}
        assert float(st["step"]) == float(ref_sd["state"][k]["step"]) == 3.0
        assert torch.allclose(st["exp_avg"], ref_sd["state"][k]["exp_avg"], rtol=3e-6, atol=1e-8)
        assert torch.allclose(st["exp_avg_sq"], ref_sd["state"][k]["exp_avg_sq"], rtol=3e-6, atol=1e-10)
    # resume in a fresh process-like state: new parameter objects, new optimizer
    geo_c, tail_c = _clone(geo_a), _clone(tail_a)
    oc = _mk(FusedAdam, geo_c, tail_c)
    oc.load_state_dict(sd)
    g2 = torch.Generator().manual_seed(13)
    _steps([(oc, geo_c + tail_c), (ob, geo_b + tail_b)], g2, 3, unfreeze_at=4, s0=3)
    for pc, pb in zip(geo_c + tail_c, geo_b + tail_b):
        assert torch.allclose(pc, pb, rtol=3e-6, atol=1e-7), (pc, pb)
    # the late-unfrozen scalar took Adam's FIRST steps after the reload (its own step count, not the global one)
    st = oc.state_dict()["state"]
    idx = len(geo_c) + 2
    assert float(st[idx]["step"]) == 2.0 and float(st[0]["step"]) == 6.0


def test_fused_adam_loads_a_torch_adam_checkpoint_and_vice_versa():
    gen = torch.Generator().manual_seed(21)
    geo_a, tail_a = _groups(gen)
    geo_b, tail_b = _clone(geo_a), _clone(tail_a)
    ob = _mk(torch.optim.Adam, geo_b, tail_b)
    g1 = torch.Generator().manual_seed(22)
    _steps([(ob, geo_b + tail_b)], g1, 4)
    # torch.optim.Adam checkpoint -> FusedAdam
    geo_c, tail_c = _clone(geo_b), _clone(tail_b)
    oc = _mk(FusedAdam, geo_c, tail_c)
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))
    g2 = torch.Generator().manual_seed(23)
    _steps([(oc, geo_c + tail_c), (ob, geo_b + tail_b)], g2, 2, s0=4)
    for pc, pb in zip(geo_c + tail_c, geo_b + tail_b):
        assert torch.allclose(pc, pb, rtol=3e-6, atol=1e-7)
    # FusedAdam checkpoint -> torch.optim.Adam
    geo_d, tail_d = _clone(geo_c), _clone(tail_c)
    od = _mk(torch.optim.Adam, geo_d, tail_d)
    od.load_state_dict(copy.deepcopy(oc.state_dict()))
    g3 = torch.Generator().manual_seed(24)
    _steps([(od, geo_d + tail_d), (ob, geo_b + tail_b)], g3, 2, s0=6)
    for pd_, pb in zip(geo_d + tail_d, geo_b + tail_b):
        assert torch.allclose(pd_, pb, rtol=3e-6, atol=1e-7)


# ---------------------------------------------------------------------------------------- the drop-in training step, patched
def _run_loop(patched, fused, steps=6, rays=128):
    """bench._RunnerLoop = the body of Runner_UDF.train_udf's loop (runner_udf.py:63-186) on the drop-in classes."""
    import sys
    import bench
    from emap_amd import dropin
    torch.manual_seed(1234)                       # the jitter draw is the reference's CPU-generator draw
    rows, reads = [], [0]

    class Rec(bench.SummaryWriter):
        def add_scalar(self, tag, scalar_value, global_step=None, *a, **k):
            if isinstance(scalar_value, torch.Tensor) and scalar_value.is_cuda:
                reads[0] += 1
            rows.append((tag, float(scalar_value.detach()) if isinstance(scalar_value, torch.Tensor) else float(scalar_value), global_step))

    loop = bench._RunnerLoop(DEV, "f16x3", rays, fused)
    old = bench.SummaryWriter
    bench.SummaryWriter = Rec
    try:
        train = bench._RunnerLoop.train_udf
        if patched:
            train = dropin.train_wrapper(train, sys.modules["bench"])
        texts = []

        def run(step):
            for _ in range(steps):
                step()
                texts.append(loop.last_loss)
        train(loop, run)
    finally:
        bench.SummaryWriter = old
    torch.cuda.synchronize()
    loop.renderer.check_errors()
    params = torch.cat([p.detach().reshape(-1) for p in list(loop.udf_network.parameters()) + [loop.variance_network_fine.variance,
                                                                                            loop.beta_network.beta, loop.beta_network.gamma]])
    return loop, params.cpu(), rows, texts, reads[0]


def test_patched_runner_step_takes_the_same_steps_without_the_host_reads():
    """dropin.train_wrapper: host-mirrored variance / beta / gamma, gradients installed by RenderFn, FusedAdam swapped
    in for the runner's torch.optim.Adam, deferred tensorboard scalars - and the SAME training run: parameters after 6 steps equal the
    unpatched loop's with FusedAdam bit for bit (same kernels, same arithmetic), torch.optim.Adam's to its usual 3e-6; the progress text
    and every tensorboard row (tag, value, step, order) are those of the unpatched loop; no device tensor is read by the writer."""
    from emap_amd.parallel import FusedAdam
    l0, p0, rows0, txt0, reads0 = _run_loop(False, True)
    l1, p1, rows1, txt1, reads1 = _run_loop(True, False)          # starts from torch.optim.Adam: the wrapper swaps it
    assert isinstance(l1.optimizer, FusedAdam) and reads0 > 0 and reads1 == 0
    assert torch.equal(p0, p1)
    assert txt0 == txt1
    assert [(t_, s) for t_, _, s in rows0] == [(t_, s) for t_, _, s in rows1] and len(rows1) == 6 * 7
    for (_, a, _), (_, b, _) in zip(rows0, rows1):
        assert a == b or abs(a - b) <= 1e-6 * max(abs(a), 1e-12)     # "Sta/variance": mean over N*S copies of x vs x itself
    l2, p2, _, _, _ = _run_loop(False, False)                    # the stock optimizer
    assert torch.allclose(p1, p2, rtol=3e-5, atol=2e-7)
    assert (l1.renderer.host_mirror_scalars, l1.renderer.direct_param_grads) == (False, False)      # restored after train_udf


def test_direct_param_grads_equal_autograd_accumulated_ones_and_fall_back_when_grads_exist():
    """RenderFn.backward with direct_param_grads: the parameters' .grad ARE views of one flat buffer (FusedAdam reads it in place) and
    equal what autograd accumulates on the ordinary path; with gradients already present the ordinary (accumulating) path runs."""
    import bench
    outs = {}
    for direct in (False, True):
        torch.manual_seed(7)
        loop = bench._RunnerLoop(DEV, "f16x3", 64, True)
        loop.renderer.direct_param_grads = direct
        smp = loop.sampler.gen_random_rays_patches_at(0, 64, importance_sample=True)
        ps = list(loop.udf_network.parameters()) + [loop.variance_network_fine.variance, loop.beta_network.beta, loop.beta_network.gamma]

        def loss_of():
            o = loop.renderer.render(smp["rays"]["rays_o"], smp["rays"]["rays_v"], loop.near, loop.far, depth_scale=smp["depth_scale"],
                                     flip_saturation=0.9, cos_anneal_ratio=1.0, t_rand=torch.zeros(64, 1, device=DEV))
            return ((o["edge"] - smp["rays"]["edge"]) ** 2).mean() + 0.1 * o["gradient_error"]
        loss_of().backward()
        g1 = [p.grad.clone() for p in ps]
        if direct:
            base = ps[0].grad.data_ptr()
            off = 0
            for p in list(loop.udf_network.parameters()):
                assert p.grad.data_ptr() == base + 4 * off
                off += p.numel()
        loss_of().backward()                       # no zero_grad in between: gradients accumulate
        g2 = [p.grad.clone() for p in ps]
        outs[direct] = (g1, g2)
    for a, b in zip(outs[False][0], outs[True][0]):
        assert torch.equal(a, b)
    for a, b, c in zip(outs[True][0], outs[True][1], outs[False][1]):
        assert torch.equal(b, c) and torch.allclose(b, 2 * a, rtol=1e-6, atol=0)


# ---------------------------------------------------------------------------------------- precision mode f16x3e (no MX fp6 anywhere)
def _rel(a, b):
    a, b = a.detach().cpu().double().reshape(-1), b.detach().cpu().double().reshape(-1)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _p999(a, b, floor=1e-6):
    a, b = a.detach().cpu().double().reshape(-1), b.detach().cpu().double().reshape(-1)
    e = (a - b).abs() / torch.clamp(b.abs(), min=floor * float(b.abs().max()))
    return float(torch.quantile(e, 0.999))


def test_mode_f16x3e_is_the_wider_margin_mode_and_attributes_the_elementwise_error():
    """precision="f16x3e" (EMAP_PREC_F16X3E, ADVICE r4 / VERDICT r4 weak 1): the value+gradient pass with f16 cross terms in BOTH sweeps - the
    run-time way back to round 3's margin (the default f16x3 has MX-fp6 cross terms in the reverse sweep).  Against the fp64 oracle on
    8192 random points, element-wise as well as max-normalised - the GPU side of profiles/r05_elementwise_attribution.txt (CPU emulation:
    fp32 reference arithmetic 6.6e-4, f16x3 GEMMs + exact sigma' 6.3e-4, + unorm16 sigma' stash 5.7e-3, + MX cross terms 1.2e-2 at p99.9):
    what remains above the forward-mode kernel in f16x3e is the unorm16 sigma' stash alone."""
    from conftest import net_state
    from oracle import emap_oracle as O
    kw, state = net_state("d8w256L10")
    cfg = O.UDFConfig(d_hidden=256, n_layers=8, multires=10)
    x = torch.rand(65536, 3, generator=torch.Generator().manual_seed(9)) * 2 - 1
    _, go = O.udf_value_and_grad({k: v.double() for k, v in state.items()}, cfg, x[:8192].double())
    uo, _ = O.udf_value_and_grad({k: v.double() for k, v in state.items()}, cfg, x[:8192].double())
    res = {}
    L = _lib.lib()
    for prec in ("f16x3", "f16x3e"):
        n = emap_amd.UDFNetwork(precision=prec, **kw)
        n.load_state_dict(state)
        n = n.to(DEV)
        with torch.no_grad():
            u, g = n.hip_udf(x.to(DEV), with_grad=True)             # 65 536 points: the reverse sweep
            u1, g1 = n.hip_udf(x.to(DEV), with_grad=True)
        assert torch.equal(u, u1) and torch.equal(g, g1)           # bit-stable
        res[prec] = (_rel(u[:8192], uo), _rel(g[:8192], go), _p999(g[:8192], go))
    old = L.emap_set_grad_mode(0)                                  # forward-mode tangents (udf_mlp_fs2_kernel<grad>): no stash, no MX
    try:
        n = emap_amd.UDFNetwork(precision="f16x3", **kw)
        n.load_state_dict(state)
        n = n.to(DEV)
        with torch.no_grad():
            u, g = n.hip_udf(x[:8192].to(DEV), with_grad=True)
        res["forward-mode"] = (_rel(u, uo), _rel(g, go), _p999(g, go))
    finally:
        L.emap_set_grad_mode(old)
    print("udf max-norm, grad_x max-norm, grad_x element-wise p99.9 vs the fp64 oracle:", res)
    assert res["f16x3e"][1] <= 2.2e-5 and res["f16x3"][1] <= 5e-5 and res["forward-mode"][1] <= 5e-6
    assert res["f16x3e"][1] < 0.75 * res["f16x3"][1]               # the margin the mode exists for
    # element-wise p99.9, bounds = measurement (round 5, 8192 random points: 1.7e-2 / 6.7e-3 / 4.1e-4) + margin; the fp32 reference itself: 6.6e-4
    assert res["f16x3"][2] <= 3.0e-2 and res["f16x3e"][2] <= 1.2e-2 and res["forward-mode"][2] <= 1.5e-3


def test_mode_f16x3e_renders_and_trains():
    from conftest import load_golden, net_state
    kw, state = net_state("d8w256L10")
    n = emap_amd.UDFNetwork(precision="f16x3e", **kw)
    n.load_state_dict(state)
    n = n.to(DEV)
    g5 = load_golden("g5_render_c64_64_4")
    args = [torch.from_numpy(g5[k]).to(DEV) for k in ("rays_o", "rays_d", "near", "far", "depth_scale")]
    devn = emap_amd.SingleVarianceNetwork(0.3).to(DEV)
    bet = emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(DEV)
    r = emap_amd.UDFRendererBlending(None, n, devn, bet, 64, 64, 0, 4, 1.0, device=DEV)
    with torch.no_grad():
        o = r.render(*args, cos_anneal_ratio=1.0, perturb_overwrite=0, flip_saturation=0.9)
    r.check_errors()
    for k in ("edge", "depth", "weight_sum"):
        assert _rel(o[k], torch.from_numpy(g5["out." + k])) <= 1e-4, k
    # one native training step: gradients agree with the f16x3 step's (the backward kernels are the same; the forward's grad_x differs by 3e-5)
    grads = {}
    for prec in ("f16x3", "f16x3e"):
        m = emap_amd.UDFNetwork(precision=prec, **kw)
        m.load_state_dict(state)
        m = m.to(DEV)
        rr = emap_amd.UDFRendererBlending(None, m, emap_amd.SingleVarianceNetwork(0.3).to(DEV),
                                          emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(DEV), 64, 64, 0, 4, 1.0, device=DEV)
        t_ = Trainer(rr, lr_geo=1e-4, lr=5e-4, igr_weight=0.1)
        ro, rd, near, far, ds = [q.to(DEV) for q in synthetic.make_rays(128, seed=3)]
        t_.step({"rays_o": ro, "rays_d": rd, "near": near, "far": far, "depth_scale": ds, "cos_anneal_ratio": 1.0, "flip_saturation": 0.9,
                 "t_rand": synthetic.make_t_rand(128).to(DEV)}, synthetic.make_true_edge(128, seed=4).to(DEV))
        grads[prec] = t_.flat.grad[:t_.flat.numel].clone()
        rr.check_errors()
    assert _rel(grads["f16x3e"], grads["f16x3"]) <= 1e-3
