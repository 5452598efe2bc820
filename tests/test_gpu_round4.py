"""GPU tests added in round 4 (-m gpu), all through the C ABI / the drop-in classes:

  * render() with n_importance = 0 (udf_renderer_blending.py:740: importance_sample skipped) against goldens recorded from the
    reference - SURVEY par. 8d's second reading of config C1;
  * the element-wise relative error distribution (99.9th percentile) behind the max-normalised 1e-4 bars;
  * the masked Adam (frozen / late un-frozen scalars, runner_udf.py:141-154) against torch.optim.Adam, eager and from a captured
    graph; emap_amd.parallel.FusedAdam against torch.optim.Adam;
  * a 48-step training run recorded from the reference's own classes (golden g12) tracked by the native Trainer.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, t, net_state, NETS
import emap_amd
from emap_amd import _lib, synthetic
from emap_amd.parallel import Trainer, FusedAdam

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a = a.detach().cpu().double().reshape(-1)
    b = b.detach().cpu().double().reshape(-1)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rel_p999(a, b, floor=1e-6):
    """99.9th percentile of the ELEMENT-WISE relative error |a - b| / max(|b|, floor * max|b|)."""
    a = a.detach().cpu().double().reshape(-1)
    b = b.detach().cpu().double().reshape(-1)
    e = (a - b).abs() / torch.clamp(b.abs(), min=floor * float(b.abs().max()))
    return float(torch.quantile(e, 0.999)) if e.numel() > 1 else float(e.max())


def mk(name, precision="f16x3"):
    kw, state = net_state(name)
    net = emap_amd.UDFNetwork(scale=1.0, precision=precision, **kw)
    net.load_state_dict(state)
    return net.to(DEV), state


def mk_renderer(net, ns, ni, steps):
    dev = emap_amd.SingleVarianceNetwork(0.3).to(DEV)
    bet = emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(DEV)
    return emap_amd.UDFRendererBlending(None, net, dev, bet, ns, ni, 0, steps, 1.0, device=DEV)


# ---------------------------------------------------------------------------------------- n_importance = 0
RENDER_KEYS = ["udf", "edge", "weight_sum", "weight_sum_fg_bg", "depth", "variance", "beta", "gamma", "normals", "gradients", "gradients_flip",
               "weights", "gradient_error", "gradient_error_near_surface", "inside_sphere", "gradient_mag", "mid_z_vals", "dists"]


@pytest.mark.parametrize("case,netname", [("c64_0", "d8w256L10"), ("c64_0_small", "d4w128L10")])
def test_render_without_importance_sampling_vs_reference_golden(case, netname):
    """n_importance = 0: the coarse samples go straight to render_core - no discontinuous sampler in between: every per-ray entry of
    the dict and every per-sample entry that does not see grad_x is held to the 1e-4 bar on all rays."""
    g = load_golden("g5_render_" + case)
    ns, ni, steps = [int(v) for v in g["cfg"]]
    assert ni == 0
    net, _ = mk(netname)
    r = mk_renderer(net, ns, ni, steps)
    assert r.samples_per_ray == ns
    a = [t(g[k]).to(DEV) for k in ("rays_o", "rays_d", "near", "far", "depth_scale")]
    with torch.no_grad():
        out = r.render(*a, cos_anneal_ratio=1.0, perturb_overwrite=0, flip_saturation=0.9)
        out2 = r.render(*a, cos_anneal_ratio=0.3, perturb_overwrite=0, flip_saturation=0.0, background_rgb=torch.ones(1, 1, device=DEV))
    torch.cuda.synchronize()
    r.check_errors()
    worst = {}
    for k in RENDER_KEYS:
        ref = t(g["out." + k])
        assert tuple(out[k].shape) == tuple(ref.shape), k
        worst[k] = rel(out[k], ref)
    print(f"n_importance = 0 render {case}: " + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()))
    # The coarse z grid is defined to an ulp only (torch.linspace's vectorised CPU kernel and this build's closed form differ in the
    # last bit of some entries: mid_z_vals 7.9e-8 = one ulp) and the network's second derivative is large (PE octaves up to 2^9):
    # one ulp of z moves grad_x u by ~2e-4 of its range.  Per-sample tensors that see grad_x get that bound, everything else 1e-4
    # (measured on MI355X, round 4: gradients 2.5e-4 / 2.0e-4, gradient_mag 1.6e-4, weights 1.8e-4 / 2.1e-6).
    ULP_SENSITIVE = {"gradients": 5e-4, "gradients_flip": 5e-4, "gradient_mag": 3e-4, "weights": 3e-4}
    for k, v in worst.items():
        assert v <= ULP_SENSITIVE.get(k, 1e-4), (k, v)
    for k in ["edge", "depth", "weights", "normals", "gradient_error"]:
        assert rel(out2[k], t(g["out2." + k])) <= {"weights": 6e-4, "normals": 3e-4}.get(k, 1e-4), k     # measured: weights 4.1e-4, normals 1.2e-4 at cos_anneal_ratio 0.3


# ---------------------------------------------------------------------------------------- element-wise error distribution
# measured on MI355X (round 4) + margin: 99.9th percentile of the element-wise relative error (floor 1e-6 of the tensor's max)
# (measured: udf 1.1e-6 / 9.7e-7 / 2.3e-5; grad_x forward-mode kernel 6.3e-4 / 3.0e-4 / 2.3e-5, reverse sweep with MX-fp6 cross terms
#  1.3e-2 / 5.8e-3 - components 1e-4 .. 1e-6 of the largest one carry the same ABSOLUTE error as the large ones; edge 2.3e-5)
# round 5: "gradients_rev" tightened from 4e-2 to measurement (1.2e-2 / 6.8e-3 on the three boxes of rounds 4-5) + margin; what it consists of is
# attributed in profiles/r05_elementwise_attribution.txt and tests/test_gpu_round5.py (fp32 floor 6.6e-4; unorm16 sigma' stash 5.7e-3; MX cross terms 1.2e-2)
P999_BOUND = {"udf": 2e-4, "gradients_fwd": 2e-3, "gradients_rev": 2e-2, "edge": 2e-4}


@pytest.mark.parametrize("name", ["d8w256L10", "d8w256L6", "d4w128L10"])
def test_elementwise_relative_error_of_the_mlp(name):
    """rel() is normalised by the tensor's largest entry; this prints and bounds the element-wise distribution for
    the MLP value and gradient (golden g2, forward-mode kernel for 256 points and the reverse sweep inside a large launch)."""
    g = load_golden("g2_mlp")
    x = t(g["x"]).to(DEV)
    ur, gr = t(g[f"{name}.udf"]), t(g[f"{name}.grad"]).reshape(-1, 3)
    net, _ = mk(name)
    with torch.no_grad():
        u, gd = net.hip_udf(x, with_grad=True)
        xb = torch.cat([x, torch.rand(20000, 3, device=DEV) * 2 - 1])
        ub, gb = net.hip_udf(xb, with_grad=True)
    res = {"udf": (rel(u, ur), rel_p999(u, ur)), "grad (small launch)": (rel(gd, gr), rel_p999(gd, gr)),
           "grad (large launch)": (rel(gb[:256], gr), rel_p999(gb[:256], gr))}
    print(f"{name}: " + "; ".join(f"{k}: max-normalised {a:.2e}, element-wise p99.9 {b:.2e}" for k, (a, b) in res.items()))
    assert res["udf"][1] <= P999_BOUND["udf"]
    assert res["grad (small launch)"][1] <= P999_BOUND["gradients_fwd"] and res["grad (large launch)"][1] <= P999_BOUND["gradients_rev"]


@pytest.mark.parametrize("case,netname", [("c64_64_4", "d8w256L10"), ("c64_50_5", "d8w256L10"), ("c64_0", "d8w256L10")])
def test_elementwise_relative_error_of_the_rendered_edge(case, netname):
    g = load_golden("g5_render_" + case)
    ns, ni, steps = [int(v) for v in g["cfg"]]
    net, _ = mk(netname)
    r = mk_renderer(net, ns, ni, steps)
    a = [t(g[k]).to(DEV) for k in ("rays_o", "rays_d", "near", "far", "depth_scale")]
    with torch.no_grad():
        out = r.render(*a, cos_anneal_ratio=1.0, perturb_overwrite=0, flip_saturation=0.9)
    e_max, e_999 = rel(out["edge"], t(g["out.edge"])), rel_p999(out["edge"], t(g["out.edge"]))
    print(f"edge {case}: max-normalised {e_max:.2e}, element-wise p99.9 (= max over {out['edge'].numel()} rays) {e_999:.2e}")
    assert e_max <= 1e-4 and e_999 <= P999_BOUND["edge"]


# ---------------------------------------------------------------------------------------- masked Adam
def _adam_reference(tensors, grads_per_step, lrs):
    """torch.optim.Adam on per-parameter clones; grads_per_step[s][i] is None for a frozen parameter at step s."""
    ps = [torch.nn.Parameter(x.clone()) for x in tensors]
    opt = torch.optim.Adam([{"params": ps[:1], "lr": lrs[0]}, {"params": ps[1:]}], lr=lrs[1])
    for gs in grads_per_step:
        for p, g_ in zip(ps, gs):
            p.grad = None if g_ is None else g_.clone()
        opt.step()
    return [p.detach() for p in ps]


def test_masked_adam_matches_torch_adam_with_frozen_and_late_unfrozen_scalars():
    """emap_adam_step_masked: a frozen tail element is skipped (no update, no state) and its step count starts when it is un-frozen -
    torch.optim.Adam's behaviour for a parameter that first gets a gradient at step t (runner_udf.py:150-154)."""
    L = _lib.lib()
    gen = torch.Generator().manual_seed(11)
    n_geo, n_tail, steps = 5003, 3, 7
    geo = torch.randn(n_geo, generator=gen)
    tail = [torch.randn(1, generator=gen) for _ in range(n_tail)]
    grads = [[torch.randn(n_geo, generator=gen) * 1e-2] + [torch.randn(1, generator=gen) for _ in range(n_tail)] for _ in range(steps)]
    frozen_until = [0, 4, 2]                     # tail element j gets its first gradient at step frozen_until[j]
    for s in range(steps):
        for j in range(n_tail):
            if s < frozen_until[j]:
                grads[s][1 + j] = None
    ref = _adam_reference([geo] + tail, grads, (1e-3, 5e-3))
    p = torch.cat([geo] + tail).to(DEV)
    m, v, tcount = torch.zeros_like(p), torch.zeros_like(p), torch.zeros(1, device=DEV)
    mask, tstep = torch.zeros(n_tail, device=DEV), torch.zeros(n_tail, device=DEV)
    for s in range(steps):
        gflat = torch.cat([grads[s][0]] + [(g_ if g_ is not None else torch.full((1,), 123.0)) for g_ in grads[s][1:]]).to(DEV)
        mask.copy_(torch.tensor([0.0 if g_ is None else 1.0 for g_ in grads[s][1:]]))
        _lib.check(L.emap_adam_step_masked(_lib.ptr(p), _lib.ptr(gflat), _lib.ptr(m), _lib.ptr(v), _lib.ptr(tcount), n_geo + n_tail, n_geo,
                                           1e-3, 5e-3, 0.9, 0.999, 1e-8, _lib.ptr(mask), _lib.ptr(tstep), _lib.stream_ptr()))
    torch.cuda.synchronize()
    got = p.cpu()
    assert torch.allclose(got[:n_geo], ref[0], rtol=2e-6, atol=1e-7)
    for j in range(n_tail):
        assert torch.allclose(got[n_geo + j], ref[1 + j], rtol=2e-6, atol=1e-7), (j, got[n_geo + j], ref[1 + j])
    assert tstep.cpu().tolist() == [float(steps - f) for f in frozen_until]


def _batch(N, seed):
    rays = synthetic.make_rays(N, seed=seed)
    b = dict(zip(("rays_o", "rays_d", "near", "far", "depth_scale"), [v.to(DEV) for v in rays]))
    b.update(cos_anneal_ratio=0.7, flip_saturation=0.5, t_rand=synthetic.make_t_rand(N, seed=seed + 1).to(DEV))
    return b, synthetic.make_true_edge(N, seed=seed + 2).to(DEV)


def _trainer(frozen):
    net, _ = mk("d4w128L10")
    r = mk_renderer(net, 32, 32, 4)
    if frozen:
        r.deviation_network.variance.requires_grad_(False)
    return Trainer(r, lr_geo=2e-5, lr=1e-3, edge_weight=1.0, igr_weight=0.1, igr_ns_weight=0.05), r


def test_trainer_honours_a_frozen_variance_eager_and_captured():
    """`variance.requires_grad = False` (SingleVarianceNetwork(init, requires_grad=False) until the runner's
    set_trainable(), runner_udf.py:150-154) - eager steps and replays of a captured graph leave it untouched, un-freezing between
    replays takes effect through the device mask (no re-capture), and the two launch modes stay bit-identical."""
    b, te = _batch(48, 70)
    te2 = te.clone()
    tr_e, r_e = _trainer(True)
    tr_g, r_g = _trainer(True)
    replay = tr_g.capture(b, te2, warmup=1)           # one real warm-up step (a single-graph capture pass does not execute)
    tr_e.step(b, te)
    v0 = float(torch.tensor(0.3, dtype=torch.float32))
    assert float(r_e.deviation_network.variance) == v0 and float(r_g.deviation_network.variance) == v0
    for _ in range(2):
        le = tr_e.step(b, te); lg = replay()
    assert float(r_g.deviation_network.variance) == v0 and torch.equal(le, lg)
    r_e.deviation_network.variance.requires_grad_(True)          # = set_trainable()
    r_g.deviation_network.variance.requires_grad_(True)
    le = tr_e.step(b, te); lg = replay()
    # the first update of the un-frozen scalar is one full lr step (its OWN step count is 1: |m / sqrt(v)| = 1 after bias correction),
    # not the ~3x larger one a global step count of 5 would give
    gv = float(tr_e.flat.grad[tr_e.flat.offsets[id(r_e.deviation_network.variance)]])
    d1 = float(r_e.deviation_network.variance) - v0
    assert d1 == pytest.approx(-1e-3 * gv / (abs(gv) + 1e-8), rel=1e-3), (d1, gv)
    for _ in range(2):
        le = tr_e.step(b, te); lg = replay()
    torch.cuda.synchronize()
    assert torch.equal(le, lg)
    assert torch.equal(tr_e.flat.data, tr_g.flat.data)


def test_fused_adam_matches_torch_adam_on_the_runners_parameter_groups():
    """emap_amd.parallel.FusedAdam (one launch) == torch.optim.Adam over the runner's groups (runner_base.py:106-117), a frozen
    scalar included, with the learning rates changed between steps as the runner's schedulers do."""
    gen = torch.Generator().manual_seed(5)
    shapes_geo = [(16, 7), (16,), (16, 1), (8, 16), (8,)]
    geo_a = [torch.nn.Parameter(torch.randn(*s, generator=gen).to(DEV)) for s in shapes_geo]
    tail_a = [torch.nn.Parameter(torch.randn(1, generator=gen).to(DEV)) for _ in range(4)]
    geo_b = [torch.nn.Parameter(p.detach().clone()) for p in geo_a]
    tail_b = [torch.nn.Parameter(p.detach().clone()) for p in tail_a]
    tail_a[2].requires_grad_(False); tail_b[2].requires_grad_(False)
    oa = FusedAdam([{"params": geo_a, "lr": 1e-3}, {"params": tail_a[:2]}, {"params": tail_a[2:]}, {"params": []}], lr=5e-3)
    ob = torch.optim.Adam([{"params": geo_b, "lr": 1e-3}, {"params": tail_b[:2]}, {"params": tail_b[2:]}, {"params": []}], lr=5e-3)
    for s in range(6):
        if s == 3:
            tail_a[2].requires_grad_(True); tail_b[2].requires_grad_(True)
        for o in (oa, ob):
            for i, g_ in enumerate(o.param_groups):
                g_["lr"] = (1e-3 if i == 0 else 5e-3) * (1.0 - 0.1 * s)
            o.zero_grad()
        gs = [torch.randn(p.shape, generator=gen).to(DEV) for p in geo_a + tail_a]
        for pa, pb, g_ in zip(geo_a + tail_a, geo_b + tail_b, gs):
            if pa.requires_grad:
                pa.grad, pb.grad = g_.clone(), g_.clone()
        oa.step(); ob.step()
    for pa, pb in zip(geo_a + tail_a, geo_b + tail_b):
        assert torch.allclose(pa, pb, rtol=3e-6, atol=1e-7), (pa, pb)


# ---------------------------------------------------------------------------------------- the reference's own training run
def test_trainer_tracks_the_training_run_recorded_from_the_reference():
    """Golden g12: 48 optimizer steps taken by the REFERENCE's classes (render under autograd, EdgeLoss, loss.backward(),
    torch.optim.Adam with the runner's groups and schedules) - the native Trainer on the same rays, targets and schedules."""
    g = load_golden("g12_training_steps")
    netname = str(g["netname"])
    ns, ni, steps_up = [int(v) for v in g["cfg"]]
    N, n_steps = int(g["n_rays"]), int(g["n_steps"])
    lr, lr_geo, alpha, warm_up_end, end_iter, anneal_end, fix_geo_end = [float(v) for v in g["schedule"]]
    ew, igr, igr_ns = [float(v) for v in g["weights3"]]
    net, state = mk(netname)
    r = mk_renderer(net, ns, ni, steps_up)
    tr = Trainer(r, lr_geo=lr_geo, lr=lr, edge_weight=ew, igr_weight=igr, igr_ns_weight=igr_ns)
    losses = []
    for it in range(n_steps):
        lr_geo_it, lr_it = [float(v) for v in g["lrs"][it]]          # the schedule as the reference run applied it
        tr.optimizer.param_groups[0]["lr"], tr.optimizer.param_groups[1]["lr"] = lr_geo_it, lr_it
        rays = synthetic.make_rays(N, seed=int(g["ray_seed0"]) + it)
        b = dict(zip(("rays_o", "rays_d", "near", "far", "depth_scale"), [v.to(DEV) for v in rays]))
        b.update(cos_anneal_ratio=float(np.min([1.0, it / anneal_end])), flip_saturation=0.0, perturb_overwrite=0)
        losses.append(tr.step(b, synthetic.make_true_edge(N, seed=int(g["edge_seed0"]) + it).to(DEV)))
    r.check_errors()
    hl = torch.stack(losses).cpu().double()
    rl = torch.stack([t(g["loss"]), t(g["edge_loss"])], 1).double()
    dev_loss = ((hl - rl).abs() / rl.abs()).max(dim=0).values
    print(f"reference training run, {n_steps} steps: max relative deviation of loss / edge_loss {float(dev_loss[0]):.2e} / {float(dev_loss[1]):.2e}; "
          f"loss first -> last: reference {float(rl[0, 0]):.4f} -> {float(rl[-1, 0]):.4f}, HIP {float(hl[0, 0]):.4f} -> {float(hl[-1, 0]):.4f}")
    # the discontinuous sampler re-samples a few rays differently on any two machines (cf. the g5 tests): a bound on the loss
    # curve, and the direction / size of the parameter displacement
    assert float(dev_loss.max()) <= 3e-2
    dh = torch.cat([(p.detach().cpu() - state[k]).reshape(-1) for k, p in net.named_parameters()])
    dr = torch.cat([(t(g["final." + k]) - state[k]).reshape(-1) for k, _ in net.named_parameters()])
    cos = float((dh * dr).sum() / (dh.norm() * dr.norm()))
    print(f"parameter displacement after {n_steps} steps: cos(HIP, reference) = {cos:.5f}, |HIP| / |reference| = {float(dh.norm() / dr.norm()):.4f}")
    assert cos >= 0.97 and abs(float(dh.norm() / dr.norm()) - 1.0) <= 0.05
    assert float(r.deviation_network.variance) == pytest.approx(float(g["variance"][-1]), rel=2e-2)
    assert float(r.beta_network.beta) == pytest.approx(float(g["beta"][-1]), rel=2e-2)


# ---------------------------------------------------------------------------------------- precision mode f16x3m
def test_mode_f16x3m_meets_the_gate_with_its_own_recorded_margin():
    """precision="f16x3m": the value+gradient pass with MX-fp6 cross terms in its FORWARD sweep too (include/emap_hip.h EMAP_PREC_F16X3M).
    Same 1e-4 gate on the g2 golden points and on random points against the oracle, bit-stable run to run; what it gives up is the
    margin (measured: udf 1.6e-5, grad_x 6.2e-5 on the g2 points / 6.6e-5 on random points, against 5e-7 / 3.0e-5 of f16x3) - the reason it is not the default mode."""
    from conftest import load_golden, net_state
    from oracle import emap_oracle as O
    g = load_golden("g2_mlp")
    x = torch.from_numpy(g["x"]).to(DEV)
    kw, state = net_state("d8w256L10")
    nets = {}
    for prec in ("f16x3", "f16x3m"):
        n = emap_amd.UDFNetwork(precision=prec, **kw)
        n.load_state_dict(state)
        nets[prec] = n.to(DEV)
    ur, gr = torch.from_numpy(g["d8w256L10.udf"]), torch.from_numpy(g["d8w256L10.grad"]).reshape(-1, 3)
    xb = (torch.rand(65536, 3, generator=torch.Generator().manual_seed(5)) * 2.4 - 1.2)
    cfg = O.UDFConfig(d_hidden=256, n_layers=8, multires=10)
    uo, go = O.udf_value_and_grad({k: v.double() for k, v in state.items()}, cfg, xb[:4096].double())
    err = {}
    with torch.no_grad():
        for prec, n in nets.items():
            xg = torch.cat([x, xb.to(DEV)])[:65536 + 256]      # one launch large enough for the reverse-sweep kernel
            u, gd = n.hip_udf(xg, with_grad=True)
            u1, g1 = n.hip_udf(xg, with_grad=True)
            assert torch.equal(u, u1) and torch.equal(gd, g1)
            err[prec] = (rel(u[:256], ur), rel(gd[:256], gr), rel(u[256:256 + 4096], uo), rel(gd[256:256 + 4096], go))
    print("f16x3 / f16x3m: udf, grad_x error on g2, on random points vs the fp64 oracle:", err)
    assert max(err["f16x3m"]) <= 1e-4
    assert err["f16x3"][1] <= 5e-5 and err["f16x3"][3] <= 5e-5
    # the renderer takes the mode like any other: the reference-recorded 64+64/4 render (g5), per-ray outputs at the gate
    g5 = load_golden("g5_render_c64_64_4")
    args = [torch.from_numpy(g5[k]).to(DEV) for k in ("rays_o", "rays_d", "near", "far", "depth_scale")]
    devn = emap_amd.SingleVarianceNetwork(0.3).to(DEV)
    bet = emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(DEV)
    r = emap_amd.UDFRendererBlending(None, nets["f16x3m"], devn, bet, 64, 64, 0, 4, 1.0, device=DEV)
    with torch.no_grad():
        o = r.render(*args, cos_anneal_ratio=1.0, perturb_overwrite=0, flip_saturation=0.9)
    r.check_errors()
    for k in ("edge", "depth", "weight_sum"):
        e = rel(o[k], torch.from_numpy(g5["out." + k]))
        print("f16x3m render", k, e)
        assert e <= 1e-4, k


def test_mode_f16x3m_far_points():
    """the MX block of the positional encoding that holds the RAW coordinates had a fixed scale, exact up to |x| = 1.875 and
    saturating above (cameras of the synthetic scene sit at radius 3; the reference's scenes live inside the unit sphere).  Its scale now
    follows max(1, |x|, |y|, |z|).  Points in [-3, 3]^3, grad_x against the fp64 oracle: 5.4e-4 before, 1.03e-4 now - the coarser block costs
    the sin / cos terms that share it one bit, so out there this margin-less opt-in mode sits AT the gate (inside the cube: 6.2e-5, test above);
    f16x3 is unaffected (2.7e-5)."""
    from conftest import net_state
    from oracle import emap_oracle as O
    kw, state = net_state("d8w256L10")
    xb = (torch.rand(32768, 3, generator=torch.Generator().manual_seed(6)) * 6.0 - 3.0)
    cfg = O.UDFConfig(d_hidden=256, n_layers=8, multires=10)
    uo, go = O.udf_value_and_grad({k: v.double() for k, v in state.items()}, cfg, xb[:4096].double())
    err = {}
    with torch.no_grad():
        for prec in ("f16x3", "f16x3m"):
            n = emap_amd.UDFNetwork(precision=prec, **kw)
            n.load_state_dict(state)
            u, gd = n.to(DEV).hip_udf(xb.to(DEV), with_grad=True)
            err[prec] = (rel(u[:4096], uo), rel(gd[:4096], go))
    print("points in [-3,3]^3, udf / grad_x error vs the fp64 oracle:", err)
    assert max(err["f16x3m"]) <= 1.2e-4 and max(err["f16x3"]) <= 5e-5


def test_mode_f16x3m_trains_like_f16x3():
    """The training path takes the mode too: forward in f16x3m, backward kernels as f16x3 (include/emap_hip.h) - gradients of one
    Trainer step agree with the f16x3 step's far inside the 1e-3 gate of the training gradients."""
    from conftest import net_state
    kw, state = net_state("d8w256L10")
    ro, rd, near, far, ds = synthetic.make_rays(256, seed=11)
    te = synthetic.make_true_edge(256, seed=12).to(DEV)
    tr = synthetic.make_t_rand(256, seed=13).to(DEV)
    grads = {}
    for prec in ("f16x3", "f16x3m"):
        n = emap_amd.UDFNetwork(precision=prec, **kw)
        n.load_state_dict(state)
        n = n.to(DEV)
        devn = emap_amd.SingleVarianceNetwork(0.3).to(DEV)
        bet = emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(DEV)
        r = emap_amd.UDFRendererBlending(None, n, devn, bet, 64, 64, 0, 4, 1.0, device=DEV)
        t_ = Trainer(r, lr_geo=1e-3, lr=5e-3, igr_weight=0.1, igr_ns_weight=0.05)
        batch = dict(rays_o=ro.to(DEV), rays_d=rd.to(DEV), near=near.to(DEV), far=far.to(DEV), depth_scale=ds.to(DEV),
                     cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
        stats = t_.step(batch, te)
        torch.cuda.synchronize()
        r.check_errors()
        assert torch.isfinite(stats).all()
        grads[prec] = t_.flat.grad[:t_.flat.numel].clone()
    e = rel(grads["f16x3m"], grads["f16x3"])
    print("dL/dtheta, f16x3m step vs f16x3 step: max abs diff / max |g| =", e)
    assert e <= 5e-4


def test_jitter_draw_is_the_reference_cpu_draw():
    """render() without t_rand draws (torch.rand([N, 1]) - 0.5) on the CPU generator like the reference (udf_renderer_blending.py:719); the
    pinned staging ring must hand the kernels exactly those values, call after call."""
    from conftest import net_state
    kw, state = net_state("d4w128L10")
    n = emap_amd.UDFNetwork(precision="f16x3", **kw)
    n.load_state_dict(state)
    n = n.to(DEV)
    devn = emap_amd.SingleVarianceNetwork(0.3).to(DEV)
    bet = emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(DEV)
    r = emap_amd.UDFRendererBlending(None, n, devn, bet, 16, 16, 0, 2, 1.0, device=DEV)
    torch.manual_seed(123)
    ref = [(torch.rand([40, 1]) - 0.5) for _ in range(7)]
    torch.manual_seed(123)
    got = [r._jitter_draw(40, torch.device(DEV)).cpu().reshape(40, 1) for _ in range(7)]
    for a_, b_ in zip(ref, got):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("name,H,nl,seed", [("d8w256L0", 256, 8, 46), ("d4w128L0", 128, 4, 47)])
def test_network_without_positional_encoding_vs_reference_golden(name, H, nl, seed):
    """multires = 0 (udf_model.py:26-29: raw coordinates, dims[0] = 3): value, "PE" and gradient against the reference's recording (g14) through every MLP
    kernel (small launch: fs2 value / forward-mode gradient; large launch: reverse sweep), and the training gradients against autograd through the oracle."""
    from oracle import emap_oracle as O
    g = load_golden("g14_mlp_multires0")
    kw = dict(d_in=3, d_out=1, d_hidden=H, n_layers=nl, skip_in=(4,), multires=0, bias=0.5)
    state = synthetic.make_udf_state(seed=seed, pert=0.02, **kw)
    net = emap_amd.UDFNetwork(scale=1.0, precision="f16x3", **kw)
    net.load_state_dict(state)
    net = net.to(DEV)
    x = torch.from_numpy(g["x"]).to(DEV)
    ur, gr = torch.from_numpy(g[f"{name}.out"])[:, :1], torch.from_numpy(g[f"{name}.grad"]).reshape(-1, 3)
    with torch.no_grad():
        out, pe = net(x)
        u, gd = net.hip_udf(x, with_grad=True)
        xb = torch.cat([x, (torch.rand(70000, 3, generator=torch.Generator().manual_seed(1)) * 2.4 - 1.2).to(DEV)])
        ub, gb = net.hip_udf(xb, with_grad=True)
    assert torch.equal(pe.cpu(), torch.from_numpy(g[f"{name}.pe"]))
    assert rel(out, ur) <= 1e-4 and rel(u, ur) <= 1e-4 and rel(gd, gr) <= 1e-4
    assert rel(ub[:256], ur) <= 1e-4 and rel(gb[:256], gr) <= 1e-4
    # training: d/dtheta of sum(udf^2) + sum(grad^2) through the HIP backward vs autograd through the oracle (fp64)
    xs = (torch.rand(2048, 3, generator=torch.Generator().manual_seed(2)) * 2 - 1)
    xg = xs.to(DEV).requires_grad_(True)
    o, _ = net(xg)
    loss = (o ** 2).sum() + (net.gradient(xg) ** 2).sum()
    loss.backward()
    st64 = {k: v.double().requires_grad_(True) for k, v in state.items()}
    cfg = O.UDFConfig(d_hidden=H, n_layers=nl, multires=0)
    uo, go = O.udf_value_and_grad(st64, cfg, xs.double())       # plain torch ops: differentiable w.r.t. the state
    lo = (uo ** 2).sum() + (go ** 2).sum()
    grads = torch.autograd.grad(lo, list(st64.values()))
    assert float(loss) == pytest.approx(float(lo), rel=2e-4)
    params = dict(net.named_parameters())
    for (k, _), gref in zip(st64.items(), grads):
        assert rel(params[k].grad, gref) <= 1e-3, k
