"""CPU checks of the hand-derived training backward (oracle/vjp_mirror.py = the algorithm the HIP kernels
emap_composite_bwd / emap_udf_vjp implement) against torch.autograd through the oracle, which the goldens pin to the
reference (udf_renderer_blending.py:457-625 under autograd, udf_model.py:121-135 create_graph=True)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, t, net_state
from oracle import emap_oracle as O
from oracle import vjp_mirror as M


def _net(name, dtype=torch.float64, scale=1.0, udf_type="abs"):
    kw, st = net_state(name)
    cfg = O.UDFConfig(d_hidden=kw["d_hidden"], n_layers=kw["n_layers"], multires=kw["multires"], scale=scale,
                      udf_type=udf_type)
    state = {k: v.to(dtype) for k, v in st.items()}
    return cfg, state


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("name,scale,ut", [("d8w256L10", 1.0, "abs"), ("d4w128L10", 1.0, "abs"), ("d8w256L6", 1.7, "square"),
                                           ("d4w128L10", 0.6, "sdf")])
def test_mlp_vjp_mirror_equals_double_backward(name, scale, ut):
    cfg, state = _net(name, scale=scale, udf_type=ut)
    gen = torch.Generator().manual_seed(3)
    P = 96
    x = (torch.rand(P, 3, generator=gen, dtype=torch.float64) * 2 - 1)
    du = torch.randn(P, generator=gen, dtype=torch.float64)
    dg = torch.randn(P, 3, generator=gen, dtype=torch.float64)
    st = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    xr = x.clone().requires_grad_(True)
    u = O.udf_value(st, cfg, xr)
    g = torch.autograd.grad(u, xr, torch.ones_like(u), create_graph=True)[0]
    phi = (du * u[:, 0]).sum() + (dg * g).sum()
    ref = dict(zip(st.keys(), torch.autograd.grad(phi, list(st.values()), allow_unused=True)))
    got, _ = M.mlp_vjp(state, cfg, x, du, dg)
    for l in range(cfg.n_lin):
        gk, vk, bk = (f"lin{l}.parametrizations.weight.original0", f"lin{l}.parametrizations.weight.original1", f"lin{l}.bias")
        d_g, d_v = M.weight_norm_vjp(state[gk], state[vk], got[f"lin{l}.weight"])
        assert rel(d_g, ref[gk]) < 1e-7, (l, "g")
        assert rel(d_v, ref[vk]) < 1e-7, (l, "v")
        rb = ref[bk] if ref[bk] is not None else torch.zeros_like(state[bk])
        assert float((got[f"lin{l}.bias"] - rb).abs().max()) < 1e-7 * max(1.0, float(rb.abs().max())), (l, "b")


@pytest.mark.parametrize("car,fs,bg", [(None, 0.0, None), (0.3, 0.9, None), (1.0, 0.5, 0.25)])
def test_composite_bwd_mirror_equals_autograd(car, fs, bg, monkeypatch):
    torch.manual_seed(5)
    dt = torch.float64
    g5 = load_golden("g5_render_c64_64_4")
    N = 8
    z = t(g5["z_after_step3"])[:N].to(dt)
    S = z.shape[1]
    rays_o, rays_d = t(g5["rays_o"])[:N].to(dt), t(g5["rays_d"])[:N].to(dt)
    U = (t(g5["out.udf"])[:N].to(dt) * (1 + 0.05 * torch.randn(N, S, dtype=dt))).requires_grad_(True)
    G = (t(g5["out.gradients"])[:N].to(dt) * (1 + 0.1 * torch.randn(N, S, 3, dtype=dt))).requires_grad_(True)
    var = torch.tensor([0.3], dtype=dt, requires_grad=True)
    bp = torch.tensor([0.5], dtype=dt, requires_grad=True)
    gp = torch.tensor([0.3], dtype=dt, requires_grad=True)
    cfg, state = _net("d8w256L10")
    rcfg = O.RenderConfig()
    sample_dist = float(((t(g5["far"]) - t(g5["near"])) / 64).mean())
    monkeypatch.setattr(O, "udf_value_and_grad", lambda s, c, p: (U.reshape(-1, 1), G.reshape(-1, 3)))
    out = O.render_core(state, cfg, rcfg, rays_o, rays_d, z, sample_dist, var, bp, gp, cos_anneal_ratio=car,
                        background_rgb=bg, flip_saturation=fs, analytic_grad=True)
    ds = torch.rand(N, 1, dtype=dt) + 0.5
    d_edge = torch.randn(N, 1, dtype=dt)
    d_depth = torch.randn(N, 1, dtype=dt) * 0.1
    w_ge, w_ns = 0.1, 0.05
    loss = (d_edge * out["edge"]).sum() + (d_depth * out["depth"] * ds).sum() + w_ge * out["gradient_error"] \
        + w_ns * out["gradient_error_near_surface"]
    rU, rG, rv, rb, rg = torch.autograd.grad(loss, [U, G, var, bp, gp])
    sums = out["eikonal_sums"].detach()
    inv_s = O.inv_s_from_variance(var).detach()
    beta = O.beta_from_param(bp).detach()
    gamma = O.gamma_from_param(gp).detach()
    dU, dG, dis, dbt, dgm = M.composite_bwd(rays_o, rays_d, z, sample_dist, U.detach(), G.detach(), inv_s, beta, gamma, car, fs,
                                            rcfg.near_surface, bg, d_edge, d_depth, ds, w_ge / (sums[1] + 1e-5),
                                            w_ns / (sums[3] + 1e-5))
    assert rel(dU, rU) < 1e-8
    assert rel(dG, rG) < 1e-8
    # chain to the raw parameters: x = exp(10 p) (clips inactive at these values)
    assert float(dis * 10 * inv_s) == pytest.approx(float(rv), rel=1e-8)
    assert float(dbt * 10 * beta) == pytest.approx(float(rb), rel=1e-8)
    assert float(dgm * 10 * gamma) == pytest.approx(float(rg), rel=1e-8)


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_mirror_chain_reproduces_reference_loss_backward(ci):
    """composite_bwd -> mlp_vjp -> weight_norm_vjp (the chain emap_render_bwd runs as kernels) on the oracle's z_vals ==
    the reference's own loss.backward() (goldens G6, recorded from the imported reference)."""
    g = load_golden(f"g6_training_{ci}")
    name = str(g["netname"])
    kw, st = net_state(name)
    dt = torch.float64
    state = {k: v.to(dt) for k, v in st.items()}
    cfg = O.UDFConfig(d_hidden=kw["d_hidden"], n_layers=kw["n_layers"], multires=kw["multires"])
    ns, ni, steps = [int(v) for v in g["cfg"]]
    rcfg = O.RenderConfig(ns, ni, steps)
    a32 = [t(g[k]) for k in ("rays_o", "rays_d", "near", "far", "depth_scale")]
    car, fs = float(g["cos_anneal_ratio"]), float(g["flip_saturation"])
    var, bp, gp = torch.tensor([0.3]), torch.tensor([0.5]), torch.tensor([0.3])
    with torch.no_grad():   # fp32, like the reference: the sampler's decisions must be the reference's
        out = O.render(st, cfg, rcfg, *a32, var, bp, gp, cos_anneal_ratio=car, flip_saturation=fs)
    z = out["z_vals"].to(dt)
    rays_o, rays_d = a32[0].to(dt), a32[1].to(dt)
    N, S = z.shape
    sd = float(((a32[3] - a32[2]) / ns).mean())
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((N, 1), sd, dtype=dt)], -1)
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * (z + dists * 0.5)[..., None]).reshape(-1, 3)
    U, G = O.udf_value_and_grad(state, cfg, pts)
    U, G = U.reshape(N, S), G.reshape(N, S, 3)
    inv_s, beta, gamma = O.inv_s_from_variance(var.to(dt)), O.beta_from_param(bp.to(dt)), O.gamma_from_param(gp.to(dt))
    ew, igr, igr_ns = [float(v) for v in g["weights3"]]
    d_edge = 2.0 * (out["edge"].to(dt) - t(g["true_edge"]).to(dt)) / N * ew
    sums = out["eikonal_sums"].to(dt)
    dU, dG, dis, dbt, dgm = M.composite_bwd(rays_o, rays_d, z, sd, U, G, inv_s, beta, gamma, car, fs, rcfg.near_surface, None,
                                            d_edge, None, None, igr / (sums[1] + 1e-5), igr_ns / (sums[3] + 1e-5))
    got, _ = M.mlp_vjp(state, cfg, pts, dU.reshape(-1), dG.reshape(-1, 3))
    # fp32 noise floor of the reference's own backward: sums that cancel (the last bias: sum of +-du) are only known to ~1e-6
    # of the largest gradient entry
    floor = 1e-6 * max(float(np.abs(g[k]).max()) for k in g if k.startswith("grad.lin"))
    for l in range(cfg.n_lin):
        gk, vk, bk = (f"lin{l}.parametrizations.weight.original0", f"lin{l}.parametrizations.weight.original1", f"lin{l}.bias")
        d_g, d_v = M.weight_norm_vjp(state[gk], state[vk], got[f"lin{l}.weight"])
        for key, val in ((gk, d_g), (vk, d_v), (bk, got[f"lin{l}.bias"])):
            ref = t(g["grad." + key]).to(dt)
            assert float((val - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + floor, key
    for key, val in (("variance", dis * 10 * inv_s), ("beta", dbt * 10 * beta), ("gamma", dgm * 10 * gamma)):
        ref = t(g["grad." + key]).to(dt)
        assert float((val.reshape(-1) - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + floor, key
