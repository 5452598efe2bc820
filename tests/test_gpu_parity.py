"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI / the drop-in classes, against
(i) the golden vectors recorded from the real reference and (ii) the CPU oracle on the same seeded inputs.

Tolerance policy (DESIGN.md "Parity"):
  * integer bookkeeping given identical fp32 inputs (searchsorted indices, merge permutation): bit-exact;
  * EMAP_PREC_F16X3 (split-fp16 MFMA, ~2^-22): 1e-4 relative (to the tensor's max magnitude) - north_star's bar;
  * EMAP_PREC_BF16X3 / EMAP_PREC_F16 / EMAP_PREC_BF16 (the throughput modes; BASELINE.json names bf16): measured
    looser bounds, asserted so that they cannot silently regress;
  * end-to-end render(): the importance sampler is discontinuous in its inputs (an ulp change of one udf value
    can move a whole group of samples - it also happens between two CPUs running the reference), so per-sample
    tensors are compared on the rays whose z_vals agree, and per-ray outputs (edge/depth/normals) on all rays.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import load_golden, t, net_state, NETS
import emap_amd
from emap_amd import _lib, synthetic
from oracle import emap_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a = a.detach().cpu().double().reshape(-1)
    b = b.detach().cpu().double().reshape(-1)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def mk(name, precision="f16x3", scale=1.0):
    kw, state = net_state(name)
    net = emap_amd.UDFNetwork(scale=scale, precision=precision, **kw)
    net.load_state_dict(state)
    cfg = O.UDFConfig(d_hidden=kw["d_hidden"], n_layers=kw["n_layers"], multires=kw["multires"], scale=scale)
    return net.to(DEV), state, cfg


def mk_renderer(net, ns, ni, steps):
    dev = emap_amd.SingleVarianceNetwork(0.3).to(DEV)
    bet = emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(DEV)
    return emap_amd.UDFRendererBlending(None, net, dev, bet, ns, ni, 0, steps, 1.0, device=DEV)


def test_native_library_is_what_runs():
    """The extension is in-tree and loaded; there is no eager fallback to fall back to."""
    import os
    assert os.path.exists(_lib.LIB_PATH)
    assert _lib.lib().emap_abi_version() == _lib.ABI_VERSION
    maps = open("/proc/self/maps").read()
    assert "libemap_hip.so" in maps


# ---------------------------------------------------------------------------------------- fields
@pytest.mark.parametrize("name", list(NETS))
def test_mlp_value_and_gradient_vs_reference_golden(name):
    g = load_golden("g2_mlp")
    x = t(g["x"]).to(DEV)
    ur, gr = t(g[f"{name}.udf"]), t(g[f"{name}.grad"]).reshape(-1, 3)
    net, _, _ = mk(name, "f16x3")
    with torch.no_grad():
        u, gd = net.hip_udf(x, with_grad=True)
        u2, _ = net.hip_udf(x, with_grad=False)
    assert rel(u, ur) <= 1e-4 and rel(u2, ur) <= 1e-4 and rel(gd, gr) <= 1e-4
    # the drop-in methods return the reference shapes
    with torch.no_grad():
        out, pe = net(x)
        udf, feat, pe2 = net.udf(x)
        gg = net.gradient(x)
    assert out.shape == (256, 1) and feat.shape == (256, 0) and gg.shape == (256, 1, 3)
    assert rel(pe, t(g[f"{name}.pe"])) <= 2e-6 and torch.equal(pe, pe2)
    # the other arithmetic modes, with their measured bounds on these "trained-like" weights (value, gradient):
    #   split-bf16 ~7e-6 / 2.5e-5, single-pass fp16 ~6e-4 / 1.6e-3, single-pass bf16 ~5e-3 / 1.5e-2
    for prec, tu, tg in (("bf16x3", 3e-5, 1e-4), ("f16", 2e-3, 5e-3), ("bf16", 1.5e-2, 5e-2)):
        netb, _, _ = mk(name, prec)
        with torch.no_grad():
            ub, gb = netb.hip_udf(x, with_grad=True)
            ub2, _ = netb.hip_udf(x, with_grad=False)
        assert rel(ub, ur) <= tu and rel(ub2, ur) <= tu and rel(gb, gr) <= tg, prec


def test_mlp_scale_and_udf_types():
    g = load_golden("g2_mlp")
    x = t(g["x"]).to(DEV)
    net, state, cfg = mk("d8w256L10", "f16x3", scale=1.5)
    with torch.no_grad():
        u, gd = net.hip_udf(x, with_grad=True)
    assert rel(u, t(g["scale1p5.udf"])) <= 1e-4 and rel(gd, t(g["scale1p5.grad"]).reshape(-1, 3)) <= 1e-4
    for ut in ("square", "sdf"):
        kw, state = net_state("d4w128L10")
        net = emap_amd.UDFNetwork(udf_type=ut, precision="f16x3", **kw)
        net.load_state_dict(state)
        net = net.to(DEV)
        cfg = O.UDFConfig(d_hidden=128, n_layers=4, multires=10, udf_type=ut)
        ur, gr = O.udf_value_and_grad(state, cfg, x.cpu())
        with torch.no_grad():
            u, gd = net.hip_udf(x, with_grad=True)
        assert rel(u, ur) <= 1e-4 and rel(gd, gr) <= 1e-4, ut


@pytest.mark.parametrize("P", [0, 1, 7, 63, 64, 65, 255, 257, 1000, 4099])
def test_mlp_ragged_sizes_vs_oracle(P):
    """Tail tiles / empty input: every tile geometry (1, 2, 4 column tiles; 4 and 8 waves) on sizes that do not fill it."""
    net, state, cfg = mk("d8w256L10", "f16x3")
    gen = torch.Generator().manual_seed(P)
    x = (torch.rand(P, 3, generator=gen) * 2.4 - 1.2)
    with torch.no_grad():
        u, gd = net.hip_udf(x.to(DEV), with_grad=True)
        u2, _ = net.hip_udf(x.to(DEV), with_grad=False)
    assert u.shape == (P, 1) and gd.shape == (P, 3)
    if P == 0:
        return
    ur, gr = O.udf_value_and_grad(state, cfg, x)
    assert rel(u, ur) <= 1e-4 and rel(u2, ur) <= 1e-4 and rel(gd, gr) <= 1e-4
    for prec, tol in (("bf16", 2e-2), ("f16", 3e-3), ("bf16x3", 5e-5)):
        nb, _, _ = mk("d8w256L10", prec)
        with torch.no_grad():
            ub, _ = nb.hip_udf(x.to(DEV), with_grad=False)
        assert rel(ub, ur) <= tol, prec


@pytest.mark.parametrize("name,scale", [("d8w256L10", 1.0), ("d8w256L6", 1.0), ("d8w256L10_init", 1.7)])
def test_reverse_mode_gradient(name, scale):
    """Large grad launches (>= 8193 points in f16x3 - round 6, 10240 before -, >= 16384 in the single-pass modes) run the reverse-sweep kernel (udf_mlp_rev32.inc), smaller ones the forward-mode
    tangent kernel: both are UDFNetwork.gradient (udf_model.py:121-135) and must agree with the oracle and with each
    other; the reverse kernel (persistent workgroups, sigma' stashed through global memory) must be run-to-run
    deterministic, also with several tiles per workgroup (70001 points: ragged last tile, > 2 tiles per workgroup)."""
    net, state, cfg = mk(name, "f16x3", scale=scale)
    gen = torch.Generator().manual_seed(11)
    x = (torch.rand(70001, 3, generator=gen) * 2.2 - 1.1).to(DEV)
    with torch.no_grad():
        u, g = net.hip_udf(x, with_grad=True)                  # reverse mode
        u2, g2 = net.hip_udf(x, with_grad=True)
        us, gs = net.hip_udf(x[:8192], with_grad=True)         # forward mode (same points)
        ut, gt = net.hip_udf(x[-4097:], with_grad=True)        # forward mode on the tail incl. the ragged tile
    assert torch.equal(u, u2) and torch.equal(g, g2)
    if name == "d8w256L10":      # several tiles per workgroup, both workgroups of every CU busy: repeated launches stay bit-identical
        xl = (torch.rand(300000, 3, generator=gen) * 2.2 - 1.1).to(DEV)
        with torch.no_grad():
            runs = [net.hip_udf(xl, with_grad=True) for _ in range(4)]
        assert all(torch.equal(r[0], runs[0][0]) and torch.equal(r[1], runs[0][1]) for r in runs[1:])
    assert rel(u[:8192], us) <= 2e-6 and rel(g[:8192], gs) <= 5e-5
    assert rel(u[-4097:], ut) <= 2e-6 and rel(g[-4097:], gt) <= 5e-5
    ur, gr = O.udf_value_and_grad(state, cfg, x[:2048].cpu())
    assert rel(u[:2048], ur) <= 1e-4 and rel(g[:2048], gr) <= 1e-4
    for prec, tu, tg in (("bf16", 2e-2, 8e-2), ("f16", 3e-3, 1e-2), ("bf16x3", 1e-4, 1e-4)):
        nb, _, _ = mk(name, prec, scale=scale)
        with torch.no_grad():
            ub, gb = nb.hip_udf(x[:20000], with_grad=True)      # reverse mode, single-pass arithmetic
            ub2, gb2 = nb.hip_udf(x[:20000], with_grad=True)
        assert torch.equal(ub, ub2) and torch.equal(gb, gb2)
        assert rel(ub[:2048], ur) <= tu and rel(gb[:2048], gr) <= tg, prec


@pytest.mark.parametrize("P", [8192, 8193, 8200, 10240, 10241, 10303, 16384, 16385, 32768 + 63])
def test_reverse_mode_ragged_sizes_vs_oracle(P):
    """Sizes at and just above the forward/reverse switch-over and with ragged last tiles (64-point tiles): first and last
    300 points against the oracle, in the parity mode and (from 16384 points) in a single-pass mode."""
    net, state, cfg = mk("d8w256L10", "f16x3")
    gen = torch.Generator().manual_seed(P)
    x = (torch.rand(P, 3, generator=gen) * 2 - 1)
    with torch.no_grad():
        u, g = net.hip_udf(x.to(DEV), with_grad=True)
    sel = torch.cat([torch.arange(300), torch.arange(P - 300, P)])
    ur, gr = O.udf_value_and_grad(state, cfg, x[sel])
    assert rel(u.cpu()[sel], ur) <= 1e-4 and rel(g.cpu()[sel], gr) <= 1e-4
    if P >= 16384:
        nb, _, _ = mk("d8w256L10", "f16")
        with torch.no_grad():
            ub, gb = nb.hip_udf(x.to(DEV), with_grad=True)
        assert rel(ub.cpu()[sel], ur) <= 3e-3 and rel(gb.cpu()[sel], gr) <= 1e-2


@pytest.mark.parametrize("ut", ["abs", "square", "sdf"])
def test_reverse_mode_udf_types_and_narrow_network(ut):
    """udf_type post-processing (udf_model.py:112-116: abs / square / identity, and its factor on the gradient) in the
    reverse-sweep kernel, on the default network and on a d_hidden=128 network whose skip layer is not the last one
    (so the reverse topology exists): reverse mode on 18000 points vs forward mode on the first 4096 of them + oracle."""
    from emap_amd import synthetic
    for kw, seed in ((NETS["d8w256L10"][0], 42), (dict(d_in=3, d_out=1, d_hidden=128, n_layers=6, skip_in=(3,), multires=8, bias=0.5), 7)):
        state = synthetic.make_udf_state(seed=seed, pert=0.02, **kw)
        net = emap_amd.UDFNetwork(udf_type=ut, precision="f16x3", **kw)
        net.load_state_dict(state)
        net = net.to(DEV)
        gen = torch.Generator().manual_seed(3)
        x = (torch.rand(18000, 3, generator=gen) * 2 - 1).to(DEV)
        with torch.no_grad():
            u, g = net.hip_udf(x, with_grad=True)
            us, gs = net.hip_udf(x[:4096], with_grad=True)
        assert rel(u[:4096], us) <= 2e-6 and rel(g[:4096], gs) <= 5e-5, (ut, kw["d_hidden"])
        cfg = O.UDFConfig(d_hidden=kw["d_hidden"], n_layers=kw["n_layers"], multires=kw["multires"], udf_type=ut,
                          skip_in=tuple(kw["skip_in"]))
        ur, gr = O.udf_value_and_grad(state, cfg, x[:1024].cpu())
        assert rel(u[:1024], ur) <= 1e-4 and rel(g[:1024], gr) <= 1e-4, (ut, kw["d_hidden"])


@pytest.mark.parametrize("P", [8192, 32768, 70000])
def test_mlp_all_tile_geometries_agree(P):
    """The launcher picks the tile geometry from P; all geometries must give the same numbers (bf16 and bf16x3)."""
    net, state, cfg = mk("d8w256L10", "f16x3")
    gen = torch.Generator().manual_seed(5)
    x = (torch.rand(P, 3, generator=gen) * 2.4 - 1.2).to(DEV)
    with torch.no_grad():
        u_all, g_all = net.hip_udf(x, with_grad=True)
        u_val, _ = net.hip_udf(x, with_grad=False)
        u_small, _ = net.hip_udf(x[:300], with_grad=False)
    assert rel(u_val, u_all) <= 2e-5 and rel(u_small, u_all[:300]) <= 2e-5
    ur, gr = O.udf_value_and_grad(state, cfg, x[:2048].cpu())
    assert rel(u_all[:2048], ur) <= 1e-4 and rel(g_all[:2048], gr) <= 1e-4


def test_extraction_query_pattern():
    """The second consumer of the field kernels (SURVEY par. 8 f2): get_udf_normals_grid queries `udf(pts)[0]` on a dense
    grid in 4096-point batches and `gradient()` (normalised) on jittered neighbourhoods (extract_pointcloud.py:55-90)."""
    net, state, cfg = mk("d8w256L10", "f16x3")
    lin = torch.linspace(-1, 1, 16)
    grid = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(-1, 3)
    with torch.no_grad():
        u = torch.cat([net.udf(b.to(DEV))[0] for b in grid.split(4096)]).cpu()
    ur = O.udf_value(state, cfg, grid)
    assert rel(u, ur) <= 1e-4
    near = grid[(ur[:, 0] < ur[:, 0].median())][:64]
    nb = near[:, None, :] + 0.005 * torch.randn(64, 50, 3, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        gd = net.gradient(nb.reshape(-1, 3).to(DEV)).squeeze(1)
        gd = torch.nn.functional.normalize(gd, dim=-1).cpu()
    gr = torch.nn.functional.normalize(O.udf_value_and_grad(state, cfg, nb.reshape(-1, 3))[1], dim=-1)
    assert float((gd - gr).abs().max()) <= 2e-4


def test_embedder_vs_golden():
    g = load_golden("g1_pe")
    for L in (10, 6):
        fn, d = emap_amd.get_embedder(L)
        assert d == 3 + 6 * L
        assert rel(fn(t(g["x"]).to(DEV)), t(g[f"pe_L{L}"])) <= 2e-6


def test_weight_repack_after_parameter_update():
    net, state, cfg = mk("d4w128L10", "f16x3")
    x = torch.rand(128, 3) * 2 - 1
    with torch.no_grad():
        u0, _ = net.hip_udf(x.to(DEV))
        for p in net.parameters():
            p.add_(0.01 * torch.randn_like(p))  # in-place update, like optimizer.step()
        u1, _ = net.hip_udf(x.to(DEV))
    st = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    ur = O.udf_value(st, cfg, x)
    assert rel(u1, ur) <= 1e-4 and rel(u0, ur) > 1e-3


def test_fp16_range_overflow_raises_device_flag():
    """A network whose activations leave fp16's range must not fail silently in the fp16 modes (the split-bf16 mode
    handles it): the render raises through the device error word."""
    kw, state = net_state("d4w128L10")
    big = {k: (v * 3000.0 if k.endswith("original0") and k.startswith("lin1.") else v) for k, v in state.items()}
    ro, rd, near, far, ds = [v.to(DEV) for v in synthetic.make_rays(64, seed=2)]
    for prec, must_flag in (("f16x3", True), ("bf16x3", False)):
        net = emap_amd.UDFNetwork(precision=prec, **kw)
        net.load_state_dict(big)
        r = mk_renderer(net.to(DEV), 32, 32, 4)
        with torch.no_grad():
            r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, perturb_overwrite=0)
        flagged = bool(r.error_flags() & _lib.F_MLP_NONFINITE)
        assert flagged == must_flag, prec
        if must_flag:
            with pytest.raises(RuntimeError, match="bf16x3"):
                r.check_errors()


# ---------------------------------------------------------------------------------------- sampler
def _sample_pdf(bins, w, m):
    L = _lib.lib()
    b, w = bins.to(DEV).contiguous(), w.to(DEV).contiguous()
    s = torch.empty(b.shape[0], m, device=DEV)
    inds = torch.empty(b.shape[0], m, device=DEV, dtype=torch.int64)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.check(L.emap_sample_pdf(_lib.ptr(b), _lib.ptr(w), b.shape[0], b.shape[1], m, _lib.ptr(s), _lib.ptr(inds), _lib.ptr(err),
                                 _lib.stream_ptr()))
    torch.cuda.synchronize()
    return s.cpu(), inds.cpu(), int(err.item())


@pytest.mark.parametrize("m", [10, 16])
def test_sample_pdf_indices_bit_exact_vs_reference(m):
    g = load_golden("g3_sample_pdf")
    s, inds, err = _sample_pdf(t(g["bins"]), t(g["weights"]), m)
    assert err == 0
    assert torch.equal(inds, t(g[f"inds_m{m}"]))  # searchsorted(right=True) bookkeeping: bit-exact
    ref = t(g[f"samples_m{m}"])
    # the samples themselves: (u - cdf_below) / denom amplifies the 1-ulp summation-order differences of the pdf
    assert float((s - ref).abs().max()) <= 3e-5


@pytest.mark.parametrize("m", [7, 32])
def test_sample_pdf_random_draws_vs_reference(m):
    """sample_pdf(det=False) of the drop-in: draws u like the reference (torch.rand on the CPU generator, udf_renderer_blending.py:84-85), inverts
    the CDF in the HIP kernel (emap_sample_pdf_u): with the recorded seed it reproduces the reference's samples."""
    from emap_amd.udf_renderer_blending import sample_pdf
    g = load_golden("g13_sample_pdf_random")
    torch.manual_seed(int(g["seed"]) + m)
    s = sample_pdf(t(g["bins"]).to(DEV), t(g["weights"]).to(DEV), m, det=False).cpu()
    ref = t(g[f"samples_m{m}"])
    # the same conditioning as the deterministic case; u lands anywhere in a cell, so the search itself is checked through the samples
    bad = (s - ref).abs() > 3e-5
    assert float(bad.float().mean()) <= 0.005, float((s - ref).abs().max())
    # and through the C ABI with explicit draws: indices bit-exact
    b, w, u = t(g["bins"]).to(DEV), t(g["weights"]).to(DEV), t(g[f"u_m{m}"]).to(DEV)
    N, n = b.shape
    out = torch.empty(N, m, device=DEV)
    inds = torch.empty(N, m, device=DEV, dtype=torch.int64)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.check(_lib.lib().emap_sample_pdf_u(_lib.ptr(b), _lib.ptr(w), _lib.ptr(u), N, n, m, _lib.ptr(out), _lib.ptr(inds), _lib.ptr(err),
                                            _lib.stream_ptr()), "sample_pdf_u")
    torch.cuda.synchronize()
    assert int(err.item()) == 0
    assert float((inds.cpu() != t(g[f"inds_m{m}"])).float().mean()) <= 0.002      # a draw within one ulp of a cdf value may land in the neighbouring cell


def test_sample_pdf_edge_cases():
    # n = 2 (single interval), all-zero weights, huge dynamic range, m > n
    bins = torch.tensor([[0.0, 1.0]])
    s, inds, err = _sample_pdf(bins, torch.zeros(1, 1), 4)
    ref, ri = O.sample_pdf(bins, torch.zeros(1, 1), 4, return_inds=True)
    assert torch.equal(inds, ri) and torch.allclose(s, ref, atol=1e-6) and err == 0
    gen = torch.Generator().manual_seed(3)
    bins = torch.sort(torch.rand(16, 200, generator=gen) * 6, -1)[0]
    w = torch.rand(16, 199, generator=gen) ** 8 * 1e3
    s, inds, err = _sample_pdf(bins, w, 64)
    ref, ri = O.sample_pdf(bins, w, 64, return_inds=True)
    assert (inds != ri).float().mean() <= 0.002 and err == 0
    ok = inds == ri
    assert float((s - ref)[ok].abs().max()) <= 1e-4
    # NaN weights raise the device flag instead of the reference's pdb.set_trace()
    w2 = w.clone(); w2[0, 0] = float("nan")
    _, _, err = _sample_pdf(bins, w2, 8)
    assert err & _lib.F_NAN_SAMPLES


def test_upsample_and_merge_bit_exact_vs_reference():
    g = load_golden("g4_upsample_step")
    L = _lib.lib()
    ro, rd = t(g["rays_o"]).to(DEV), t(g["rays_d"]).to(DEV)
    z, udf = t(g["z_vals"]).to(DEV), t(g["udf"]).to(DEV)
    sd = torch.tensor([float(g["sample_dist"])], device=DEV)
    for i in range(2):
        inv_s, beta, gamma = [float(v) for v in g[f"step{i}.params"]]
        N, n = z.shape
        zn = torch.empty(N, 16, device=DEV)
        inds = torch.empty(N, 16, device=DEV, dtype=torch.int64)
        _lib.check(L.emap_upsample_step(_lib.ptr(ro), _lib.ptr(rd), _lib.ptr(z), _lib.ptr(udf), N, n, 16, _lib.ptr(sd), inv_s, beta,
                                        gamma, _lib.ptr(zn), _lib.ptr(inds), None, _lib.stream_ptr()))
        assert torch.equal(inds.cpu(), t(g[f"step{i}.inds"]))
        zref = t(g[f"step{i}.z_new"])
        assert float((zn.cpu() - zref).abs().max()) <= 2e-6
        # merge: feed the reference's z_new so that the permutation is comparable bit for bit
        idx = t(g[f"step{i}.sort_index"]).to(DEV)
        u_sorted = t(g[f"step{i}.udf_out"]).to(DEV)
        cat_u = torch.empty(N, n + 16, device=DEV).scatter_(1, idx, u_sorted)  # un-sort -> [udf, udf_new]
        zo = torch.empty(N, n + 16, device=DEV); uo = torch.empty(N, n + 16, device=DEV)
        perm = torch.empty(N, n + 16, device=DEV, dtype=torch.int64)
        zn_ref, un_ref = zref.to(DEV), cat_u[:, n:].contiguous()  # named: must outlive the asynchronous launch
        _lib.check(L.emap_merge_sorted(_lib.ptr(z), _lib.ptr(zn_ref), _lib.ptr(udf), _lib.ptr(un_ref), N, n, 16,
                                       _lib.ptr(zo), _lib.ptr(uo), _lib.ptr(perm), _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert torch.equal(perm.cpu(), t(g[f"step{i}.sort_index"]))
        assert torch.equal(zo.cpu(), t(g[f"step{i}.z_out"])) and torch.equal(uo.cpu(), t(g[f"step{i}.udf_out"]))
        z, udf = t(g[f"step{i}.z_out"]).to(DEV), t(g[f"step{i}.udf_out"]).to(DEV)


def _well_conditioned(z, weights, inds, m):
    """samples whose inverse-CDF lerp is well conditioned: the pdf mass of their interval is >= 1e-3 (sample_pdf :92-96 divides
    by it; in empty intervals the reference's own result depends on the last ulp of its cumsum)"""
    w = weights + 1e-5
    pdf = w / w.sum(-1, keepdim=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=z.shape[1] - 1)
    mass = torch.where(above > below, torch.gather(pdf, 1, below.clamp(max=pdf.shape[1] - 1)), torch.zeros_like(z[:, :m]))
    return mass >= 1e-3


@pytest.mark.parametrize("case", ["c64_64_4", "c64_50_5", "c32_32_4_small"])
def test_upsampling_steps_and_chain_vs_reference(case):
    """Every one of the K up-sampling steps through the HIP kernels, for the full-size golden renders (G4 pins two steps of a
    small case bit for bit):
      (a) each step on the REFERENCE's inputs of that step (z, udf from the oracle trace, which is bit-pinned to the goldens'
          z_after_step*): searchsorted indices equal, new samples within 2e-6 wherever the inverse CDF is well conditioned,
          merge permutation and merged z bit-exact;
      (b) the steps CHAINED (each fed with the previous HIP output): fraction of rays whose final z_vals equal the golden's.
    The weights feeding sample_pdf go through exp / sigmoid, whose last ulp differs between libm implementations, so
    ill-conditioned samples (empty intervals) are reported, not asserted."""
    g = load_golden("g5_render_" + case)
    ns, ni, steps = [int(v) for v in g["cfg"]]
    kw, state = net_state(G5[case])
    cfg = O.UDFConfig(d_hidden=kw["d_hidden"], n_layers=kw["n_layers"], multires=kw["multires"])
    ro_c, rd_c, near, far = t(g["rays_o"]), t(g["rays_d"]), t(g["near"]), t(g["far"])
    N, m = ro_c.shape[0], ni // steps
    z0, sd = O.coarse_z_vals(near, far, ns, N)
    trace = []
    O.importance_sample(state, cfg, ro_c, rd_c, z0, sd, ns, ni, steps, trace=trace)
    # On the machine that recorded the goldens the oracle reproduces z_after_step* bit for bit (tests/test_oracle_vs_golden.py,
    # CPU suite); on another CPU torch's vectorised exp / sigmoid differ in the last ulp and a few empty-interval samples
    # move there as well - which is why (a) compares with the oracle evaluated HERE, on identical inputs.
    host_equal = min(float((trace[i + 1]["z_vals"] == t(g[f"z_after_step{i}"])).all(dim=1).float().mean()) for i in range(steps))
    L = _lib.lib()
    ro, rd = ro_c.to(DEV), rd_c.to(DEV)
    sdt = torch.tensor([sd], device=DEV)

    def hip_step(z, udf, i):
        n = z.shape[1]
        inv_s, beta = 64.0 * 2 ** i, 64.0 * 2 ** (i + 1)
        gamma = float(np.clip(20 * 2 ** (steps - i), 20, 320))
        zn = torch.empty(N, m, device=DEV)
        inds = torch.empty(N, m, device=DEV, dtype=torch.int64)
        _lib.check(L.emap_upsample_step(_lib.ptr(ro), _lib.ptr(rd), _lib.ptr(z), _lib.ptr(udf), N, n, m, _lib.ptr(sdt), inv_s, beta,
                                        gamma, _lib.ptr(zn), _lib.ptr(inds), None, _lib.stream_ptr()))
        return zn, inds, (inv_s, beta, gamma)

    def hip_merge(z, zn, udf, un):
        n = z.shape[1]
        zo = torch.empty(N, n + m, device=DEV)
        uo = torch.empty(N, n + m, device=DEV) if un is not None else None
        perm = torch.empty(N, n + m, device=DEV, dtype=torch.int64)
        _lib.check(L.emap_merge_sorted(_lib.ptr(z), _lib.ptr(zn), _lib.ptr(udf) if un is not None else None, _lib.ptr(un), N, n, m,
                                       _lib.ptr(zo), _lib.ptr(uo), _lib.ptr(perm), _lib.stream_ptr()))
        torch.cuda.synchronize()
        return zo, uo, perm

    # (a) step by step on reference inputs
    bad_frac, ind_frac, good_bad = 0.0, 0.0, 0.0
    for i in range(steps):
        zr, ur = trace[i]["z_vals"], trace[i]["udf"]
        zn, inds, (inv_s, beta, gamma) = hip_step(zr.to(DEV).contiguous(), ur.to(DEV).contiguous(), i)
        ref = O.up_sample_unbias(ro_c, rd_c, zr, ur, sd, m, inv_s, beta, gamma, return_all=True)
        assert torch.equal(ref["z_samples"], trace[i + 1]["new_z_vals"])
        same_ind = inds.cpu() == ref["inds"]
        good = _well_conditioned(zr, ref["weights"], ref["inds"], m) & same_ind
        dz = (zn.cpu() - ref["z_samples"]).abs()
        # (a discontinuous decision inside the weights - true_cos < 0.05, |p| < 1 - can also flip on an ulp: counted, not excused)
        good_bad = max(good_bad, float((dz[good] > 2e-6).float().mean()))
        ind_frac = max(ind_frac, 1.0 - float(same_ind.float().mean()))
        bad_frac = max(bad_frac, float((dz > 2e-6).float().mean()))
        # merge on the reference's new samples: integer bookkeeping, bit-exact
        last = i + 1 == steps
        pts = (ro_c[:, None, :] + rd_c[:, None, :] * ref["z_samples"][..., None]).reshape(-1, 3)
        un = None if last else O.udf_value(state, cfg, pts).reshape(N, m).to(DEV).contiguous()
        zo, uo, perm = hip_merge(zr.to(DEV).contiguous(), ref["z_samples"].to(DEV).contiguous(), ur.to(DEV).contiguous(), un)
        assert torch.equal(perm.cpu(), O.merge_sorted(zr, ref["z_samples"])[1]), i
        assert torch.equal(zo.cpu(), trace[i + 1]["z_vals"]), i
        if not last:
            assert torch.equal(uo.cpu(), trace[i + 1]["udf"]), i
    # (b) chained
    z, udf = trace[0]["z_vals"].to(DEV).contiguous(), trace[0]["udf"].to(DEV).contiguous()
    for i in range(steps):
        last = i + 1 == steps
        zn, _, _ = hip_step(z, udf, i)
        un = None
        if not last:
            pts = (ro_c[:, None, :] + rd_c[:, None, :] * zn.cpu()[..., None]).reshape(-1, 3)
            un = O.udf_value(state, cfg, pts).reshape(N, m).to(DEV).contiguous()
        z, uo, _ = hip_merge(z, zn, udf, un)
        udf = uo if uo is not None else udf
    ok_rays = float(((z.cpu() - t(g[f"z_after_step{steps - 1}"])).abs().max(dim=1).values <= 2e-6).float().mean())
    print(f"up-sampling {case}: per step on reference inputs: index mismatches {ind_frac:.4f}, samples off by > 2e-6 {bad_frac:.4f} "
          f"(of the well-conditioned ones {good_bad:.4f}); chained: rays with every final z within 2e-6 of the golden {ok_rays:.3f} "
          f"(this host's CPU oracle vs the golden, bit-equal rays: {host_equal:.3f})")
    b = CHAIN_BOUND[case]
    assert ind_frac <= b[0] and bad_frac <= b[1] and good_bad <= b[2] and ok_rays >= b[3], (ind_frac, bad_frac, good_bad, ok_rays)


def test_merge_ties_are_stable():
    L = _lib.lib()
    z = torch.tensor([[0.0, 1.0, 1.0, 2.0]], device=DEV)
    zn = torch.tensor([[1.0, 2.0, 3.0]], device=DEV)
    zo = torch.empty(1, 7, device=DEV); perm = torch.empty(1, 7, device=DEV, dtype=torch.int64)
    _lib.check(L.emap_merge_sorted(_lib.ptr(z), _lib.ptr(zn), None, None, 1, 4, 3, _lib.ptr(zo), None, _lib.ptr(perm), _lib.stream_ptr()))
    zr, ir = O.merge_sorted(z.cpu(), zn.cpu())
    assert torch.equal(zo.cpu(), zr) and torch.equal(perm.cpu(), ir)


# ---------------------------------------------------------------------------------------- render_core on fixed z
# measured on MI355X (round 2) + margin: fraction of rays with any sample moved by > 1e-3 w.r.t. the reference's z_vals (the
# sampler is discontinuous in its inputs; the per-step kernels are bit-exact on reference inputs, see the chain test), and the
# resulting relative difference of the batch-wide gradient_error
# (measured 0.062 / 0.031 / 0.25 / 0.0 and 1.9e-3 / 2.0e-4 / 5.9e-3 / 0)
MOVED_BOUND = {"c64_50_5": 0.13, "c64_64_4": 0.10, "c32_32_4_small": 0.35, "c64_64_4_L6": 0.07}
GE_BOUND = {"c64_50_5": 4e-3, "c64_64_4": 1e-3, "c32_32_4_small": 1.2e-2, "c64_64_4_L6": 1e-3}
# up-sampling steps on reference inputs (max over steps): (index mismatch fraction, fraction of samples off by > 2e-6, the same among
# the well-conditioned samples, fraction of rays whose chained final z_vals equal the golden's) - measured on MI355X (round 2) + margin
# measured: indices 0 / 0 / 0 mismatches; samples 0 / 0 / 3.5 % (one |p| < 1 decision of the d4 case flips); chained rays equal to the
# golden 97 / 94 / 81 % - the CPU oracle on the same host reaches 97 / 94 / 62 % against the golden recorded on another CPU
CHAIN_BOUND = {"c64_64_4": (0.002, 0.01, 0.01, 0.90), "c64_50_5": (0.002, 0.01, 0.01, 0.85), "c32_32_4_small": (0.002, 0.07, 0.07, 0.70)}
G5 = {"c64_50_5": "d8w256L10", "c64_64_4": "d8w256L10", "c32_32_4_small": "d4w128L10", "c64_64_4_L6": "d8w256L6"}
PER_SAMPLE = ["udf", "weights", "gradients", "gradients_flip", "inside_sphere", "gradient_mag", "mid_z_vals", "dists"]


def _render_core_on_z(net, r, g, z, car, fs, bg=None):
    """MLP value+grad at the reference's z_vals + emap_composite_fwd_p: render_core (:418-677) through the C ABI."""
    L = _lib.lib()
    ro, rd, near, far, ds = [t(g[k]).to(DEV) for k in ("rays_o", "rays_d", "near", "far", "depth_scale")]
    N, S = z.shape
    sd = ((far - near) / r.n_samples).mean().reshape(1)
    z = z.to(DEV).contiguous()
    dists = torch.cat([z[:, 1:] - z[:, :-1], sd.expand(N, 1)], -1)
    mid = z + dists * 0.5
    pts = (ro[:, None, :] + rd[:, None, :] * mid[..., None]).reshape(-1, 3)
    with torch.no_grad():
        udf, grad = net.hip_udf(pts, with_grad=True)
    p = r._params(N, car, fs, bg)
    names = ["weights", "alpha", "mid_z", "dists", "inside_sphere", "gradient_mag"]
    bufs = {k: torch.empty(N, S, device=DEV) for k in names}
    bufs.update(gradients_flip=torch.empty(N, S, 3, device=DEV), edge=torch.empty(N, 1, device=DEV), depth=torch.empty(N, 1, device=DEV),
                weight_sum=torch.empty(N, 1, device=DEV), normals=torch.empty(N, 3, device=DEV), scalars=torch.zeros(16, device=DEV))
    co = _lib.CompositeOut()
    for k, v in bufs.items():
        setattr(co, k, v.data_ptr())
    partials = torch.empty(N, 8, device=DEV)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    ds_flat = ds.reshape(-1).contiguous()
    _lib.check(L.emap_composite_fwd_p(_lib.ptr(ro), _lib.ptr(rd), _lib.ptr(z), _lib.ptr(udf), _lib.ptr(grad), _lib.ptr(ds_flat),
                                      N, S, _lib.ptr(sd), C.byref(p), C.byref(co), _lib.ptr(partials), _lib.ptr(err), _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert int(err.item()) == 0
    out = dict(bufs)
    out.update(udf=udf.view(N, S), gradients=grad.view(N, S, 3), mid_z_vals=bufs["mid_z"], gradient_error=bufs["scalars"][0],
               gradient_error_near_surface=bufs["scalars"][1])
    return out


@pytest.mark.parametrize("case", list(G5))
def test_render_core_on_reference_samples(case):
    """Given the reference's own z_vals, every output of render_core agrees to 1e-4 (bf16x3)."""
    g = load_golden("g5_render_" + case)
    ns, ni, steps = [int(v) for v in g["cfg"]]
    net, _, _ = mk(G5[case], "f16x3")
    r = mk_renderer(net, ns, ni, steps)
    z = t(g[f"z_after_step{steps - 1}"])
    out = _render_core_on_z(net, r, g, z, 1.0, 0.9)
    for k in PER_SAMPLE + ["edge", "depth", "normals", "gradient_error", "gradient_error_near_surface"]:
        ref = t(g["out." + k])
        assert rel(out[k].reshape(ref.shape), ref) <= 1e-4, k
    out2 = _render_core_on_z(net, r, g, z, 0.3, 0.0, bg=torch.ones(1, 1))
    for k in ["edge", "depth", "weights", "normals", "gradient_error"]:
        ref = t(g["out2." + k])
        assert rel(out2[k].reshape(ref.shape), ref) <= 1e-4, k


# ---------------------------------------------------------------------------------------- full render
@pytest.mark.parametrize("case", list(G5))
def test_full_render_vs_reference_golden(case):
    g = load_golden("g5_render_" + case)
    ns, ni, steps = [int(v) for v in g["cfg"]]
    net, _, _ = mk(G5[case], "f16x3")
    r = mk_renderer(net, ns, ni, steps)
    a = [t(g[k]).to(DEV) for k in ("rays_o", "rays_d", "near", "far", "depth_scale")]
    with torch.no_grad():
        out = r.render(*a, cos_anneal_ratio=1.0, perturb_overwrite=0, flip_saturation=0.9)
    torch.cuda.synchronize()
    r.check_errors()
    for k in ["udf", "edge", "weight_sum", "weight_sum_fg_bg", "depth", "variance", "beta", "gamma", "normals", "gradients",
              "gradients_flip", "weights", "gradient_error", "gradient_error_near_surface", "inside_sphere", "gradient_mag",
              "mid_z_vals", "dists"]:
        assert tuple(out[k].shape) == tuple(g["out." + k].shape), k  # the reference dict, key for key
    zref = t(g[f"z_after_step{steps - 1}"])
    # Per-ray outputs on ALL rays.  Sample positions inside intervals of (near-)zero weight are ill-conditioned in
    # sample_pdf ((u - cdf)/denom with denom -> 1e-5) and a searchsorted decision can flip on an ulp, in the
    # reference as well (two CPUs disagree the same way); the rendered quantities are insensitive to both.
    assert rel(out["edge"], t(g["out.edge"])) <= 1e-4
    for k, tol in (("depth", 3e-4), ("normals", 3e-4), ("weights", 1e-3)):
        assert rel(out[k], t(g["out." + k])) <= tol, k
    # (per-sample tensors are compared in test_render_core_on_reference_samples, which feeds the reference's own
    # z_vals: with |grad u| ~ 25 and beta ~ 150 one ulp of z already moves a weight by ~1e-3 relative, and the coarse
    # z grid itself is only defined to an ulp - torch.linspace differs between its CPU and CUDA kernels)
    moved = float(((out["z_vals"].cpu() - zref).abs().max(dim=1)[0] > 1e-3).float().mean())
    ge_err = rel(out["gradient_error"], t(g["out.gradient_error"]))
    print(f"full render {case}: rays with a sample moved by > 1e-3: {moved:.3f}; gradient_error rel. diff {ge_err:.2e}")
    assert moved <= MOVED_BOUND[case], moved
    for k in ["variance", "beta", "gamma"]:
        assert rel(out[k], t(g["out." + k])) <= 1e-6, k
    # gradient_error averages (|grad u| - 1)^2 over all samples of the batch, the moved ones included: it sees the re-sampled
    # intervals directly (on the reference's own z_vals it agrees to 1e-4, test_render_core_on_reference_samples)
    assert ge_err <= GE_BOUND[case], ge_err
    # single-pass bf16: edge/depth stay within a few percent
    netb, _, _ = mk(G5[case], "bf16")
    rb = mk_renderer(netb, ns, ni, steps)
    with torch.no_grad():
        ob = rb.render(*a, cos_anneal_ratio=1.0, perturb_overwrite=0, flip_saturation=0.9)
    assert rel(ob["edge"], t(g["out.edge"])) <= 0.1 and rel(ob["depth"], t(g["out.depth"])) <= 0.1


@pytest.mark.parametrize("ns,ni,steps", [(128, 128, 4), (96, 64, 4)])
def test_render_with_more_than_128_samples_on_its_own_z_vals_vs_oracle(ns, ni, steps):
    """More than 128 samples per ray (the per-ray kernels' 4-samples-per-lane forms, up-sampling lists beyond 128 entries): the full render()
    through the HIP path, then the ORACLE's render_core (fp32 and fp64) on the z_vals the HIP sampler produced.  (The sampler chain itself at these sizes: z sorted, inside [near, far].)"""
    from conftest import net_state
    from oracle import emap_oracle as O
    g = load_golden("g5_render_c64_64_4")
    N = 32
    net, state, _ = mk("d8w256L10", "f16x3")
    r = mk_renderer(net, ns, ni, steps)
    a = [t(g[k])[:N].to(DEV) for k in ("rays_o", "rays_d", "near", "far", "depth_scale")]
    with torch.no_grad():
        out = r.render(*a, cos_anneal_ratio=1.0, perturb_overwrite=0, flip_saturation=0.9)
    torch.cuda.synchronize()
    r.check_errors()
    S = ns + ni
    z = out["z_vals"].cpu()
    assert z.shape == (N, S) and bool((z[:, 1:] >= z[:, :-1]).all())
    assert bool((z >= a[2].cpu() - 1e-6).all()) and bool((z <= a[3].cpu() + 1e-6).all())
    kw, st = net_state("d8w256L10")
    cfg = O.UDFConfig(d_hidden=256, n_layers=8, multires=10)
    sd = float(((a[3] - a[2]) / ns).mean())
    errs = {}
    for dt in (torch.float32, torch.float64):
        ref = O.render_core({k: v.to(dt) for k, v in st.items()}, cfg, O.RenderConfig(n_samples=ns, n_importance=ni, up_sample_steps=steps),
                            a[0].cpu().to(dt), a[1].cpu().to(dt), z.to(dt), sd, torch.tensor([0.3], dtype=dt), torch.tensor([0.5], dtype=dt),
                            torch.tensor([0.3], dtype=dt), cos_anneal_ratio=1.0, flip_saturation=0.9, analytic_grad=True)
        errs[dt] = {"weights": rel(out["weights"], ref["weights"]), "edge": rel(out["edge"], ref["edge"]),
                    "depth": rel(out["depth"], ref["depth"] * a[4].cpu().to(dt)), "normals": rel(out["normals"], ref["normals"]),
                    "gradient_error": rel(out["gradient_error"], ref["gradient_error"])}
    print(f"render {ns}+{ni}: HIP vs the oracle on the HIP z_vals, fp32 oracle {errs[torch.float32]}, fp64 oracle {errs[torch.float64]}")
    # with 2-4 x denser samples the fp32 evaluation of the tail itself (dists = z[e+1] - z[e], then exp / sigmoid of their products) is what
    # limits the agreement: the fp32 oracle - the reference's arithmetic - differs from the fp64 one by as much as the HIP path does
    # (measured at 128+128: HIP vs fp64 oracle 1.3e-4 edge / 2.0e-4 depth / 1.5e-4 weights, HIP vs fp32 oracle 1.9e-4 / 2.7e-4 / 1.9e-4; the
    # judged shapes have 114 / 128 samples and meet 1e-4: test_render_core_on_reference_samples)
    for dt in errs:
        for k, v in errs[dt].items():
            assert v <= 4e-4, (dt, k, v)
        assert errs[dt]["gradient_error"] <= 1e-5


@pytest.mark.parametrize("netname,N", [("d8w256L10", 512), ("d8w256L10", 700), ("d8w256L10", 100), ("d8w256L10", 33), ("d8w256L10", 1024),
                                       ("d8w256L10", 1), ("d8w256L10", 2047), ("d4w128L10", 512), ("d4w128L10", 37)])
def test_fused_importance_sampling_equals_the_launch_chain_bit_for_bit(netname, N):
    _fused_vs_chain(netname, N, "f16x3")


@pytest.mark.parametrize("prec", ["f16x3m", "f16x3e", "bf16x3", "f16", "bf16"])
def test_fused_importance_sampling_in_every_precision_mode(prec):
    _fused_vs_chain("d8w256L10", 512, prec)


def _fused_vs_chain(netname, N, prec):
    """ABI v8: importance_sample (udf_renderer_blending.py:802-841) as ONE launch - the sampler steps run inside the workgroups of the narrow MLP
    passes, the ray's lists stay in LDS (udf_mlp_kernel.inc, IS) - against the chain of 2 K - 1 launches it replaces
    (emap_set_fused_sampling(0)): z_vals and every rendered quantity identical bit for bit, for every workgroup geometry the launcher picks
    (2 rays x 8 waves, 2 x 4, 1 x 8 / 1 x 4, odd ray counts), with and without the per-ray jitter."""
    from emap_amd import synthetic
    net, _, _ = mk(netname, prec)
    r = mk_renderer(net, 64, 64, 4)
    ro, rd, near, far, ds = [v.to(DEV) for v in synthetic.make_rays(N, seed=5)]
    tr = synthetic.make_t_rand(N).to(DEV)
    L = _lib.lib()
    outs = {}
    try:
        for fused in (2, 1, 0):       # 2: the fused kernel whatever the size rule says (round 6: the rule hands 768 ... 1024 rays at m = 16 to the chain), 1: the rule's pick
            L.emap_set_fused_sampling(fused)
            with torch.no_grad():
                o1 = r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
                o2 = r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, perturb_overwrite=0, flip_saturation=0.9)
            torch.cuda.synchronize()
            r.check_errors()
            outs[fused] = {tag + k: v.clone() for tag, o in (("jitter.", o1), ("plain.", o2)) for k, v in o.items() if isinstance(v, torch.Tensor)}
    finally:
        L.emap_set_fused_sampling(1)
    assert set(outs[0]) == set(outs[1]) == set(outs[2]) and "jitter.z_vals" in outs[1] and "plain.z_vals" in outs[1]
    for k in outs[1]:
        assert torch.equal(outs[1][k], outs[0][k]) and torch.equal(outs[2][k], outs[0][k]), k


def test_perturb_path_and_float_near_far():
    g = load_golden("g7_perturb")
    net, _, _ = mk("d4w128L10", "f16x3")
    r = mk_renderer(net, 32, 32, 4)
    a = [t(g[k]).to(DEV) for k in ("rays_o", "rays_d", "near", "far", "depth_scale")]
    with torch.no_grad():
        out = r.render(*a, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=t(g["t_rand"]))
        outf = r.render(a[0], a[1], 0.05, 6.0, a[4], cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=t(g["t_rand"]))
        torch.manual_seed(42)  # same CPU-generator draw as the reference (:719)
        outs = r.render(*a, cos_anneal_ratio=1.0, flip_saturation=0.9)
    for o, km, ke in ((out, "mid_z_vals", "edge"), (outf, "mid_z_float_nearfar", "edge_float_nearfar"), (outs, "mid_z_vals", "edge")):
        same = ((o["mid_z_vals"].cpu() - t(g[km])).abs().max(dim=1)[0] <= 1e-5)
        assert same.float().mean() >= 0.4
        assert rel(o["edge"], t(g[ke])) <= 1e-4


@pytest.mark.parametrize("N,prec,tol", [(512, "f16x3", 1e-3), (1024, "f16x3", 1e-3), (1024, "bf16", 0.1), (4096, "f16x3", 1e-3)])
def test_north_star_batch_properties(N, prec, tol):
    """The benchmark batch (512 rays x 128 samples), BASELINE config C2's exact shape (1024 x 128, in f16x3 and in the bf16 it
    names) and C4's global batch (4096 x 128): size-independent invariants of the path."""
    net, state, cfg = mk("d8w256L10", prec)
    r = mk_renderer(net, 64, 64, 4)
    ro, rd, near, far, ds = [v.to(DEV) for v in synthetic.make_rays(N, seed=1)]
    tr = synthetic.make_t_rand(N).to(DEV)
    with torch.no_grad():
        o1 = r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
        o2 = r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
    torch.cuda.synchronize()
    r.check_errors()
    z = o1["z_vals"]
    assert z.shape == (N, 128) and bool((z[:, 1:] >= z[:, :-1]).all())                # sorted
    assert bool((o1["weights"] >= 0).all()) and float(o1["weight_sum"].max()) <= 1 + 128e-7 + 1e-6
    assert torch.allclose(o1["weights"].sum(-1, keepdim=True), o1["weight_sum"], atol=1e-5)
    assert torch.equal(o1["edge"], o2["edge"]) and torch.equal(o1["z_vals"], o2["z_vals"])   # deterministic
    assert bool(torch.isfinite(o1["gradients"]).all()) and bool((o1["udf"] >= 0).all())
    # the coarse samples survive the merges: every coarse z is still present
    zc = near + (far - near) * torch.linspace(0, 1, 64, device=DEV)[None, :] + tr.view(-1, 1) * 2.0 / 64
    d = (z[:, None, :] - zc[:, :, None]).abs().min(dim=-1)[0]
    assert float(d.max()) <= 1e-5
    # rays are independent: rendering a sub-batch gives the same rows (the basis of the data-parallel sharding)
    with torch.no_grad():
        o3 = r.render(ro[128:256], rd[128:256], near[128:256], far[128:256], ds[128:256], cos_anneal_ratio=1.0,
                      flip_saturation=0.9, t_rand=tr[128:256])
    assert torch.equal(o3["z_vals"], o1["z_vals"][128:256]) and torch.allclose(o3["edge"], o1["edge"][128:256], atol=1e-6)
    # against the CPU oracle on a slice (same inputs), per-ray outputs
    sl = slice(0, 48)
    ref = O.render(state, cfg, O.RenderConfig(64, 64, 4), ro[sl].cpu(), rd[sl].cpu(), near[sl].cpu(), far[sl].cpu(), ds[sl].cpu(),
                   torch.tensor([0.3]), torch.tensor([0.5]), torch.tensor([0.3]), cos_anneal_ratio=1.0, t_rand=tr[sl].cpu().view(-1, 1),
                   flip_saturation=0.9)
    # rays whose samples sit where this host's CPU oracle put them (the sampler is discontinuous: a few per cent of the rays get
    # re-sampled intervals on ANY two machines, see test_upsampling_steps_and_chain_vs_reference) agree to `tol`; all rays loosely
    same = ((o1["z_vals"][sl].cpu() - ref["z_vals"]).abs().max(dim=1).values <= 1e-4)
    if prec == "f16x3":
        assert float(same.float().mean()) >= 0.8, float(same.float().mean())
        assert rel(o1["edge"][sl][same.to(DEV)], ref["edge"][same]) <= tol and rel(o1["depth"][sl][same.to(DEV)], ref["depth"][same]) <= tol
    # (a re-sampled ray can move by 10 % in depth when its weight sits in one or two samples; as a batch the rays agree)
    k = 1.0 if prec == "f16x3" else 10.0                     # single-pass bf16: a few per cent everywhere
    assert float((o1["edge"][sl].cpu() - ref["edge"]).abs().mean()) <= 5e-3 * k and float((o1["depth"][sl].cpu() - ref["depth"]).abs().mean()) <= 2e-2 * k


# ---------------------------------------------------------------------------------------- extraction queries (par. 8 f2)
def _dir_err_each(a, b):
    return 1.0 - (a * b).sum(-1).abs()


@pytest.mark.gpu
def test_null_direction_kernel_vs_svd():
    """emap_null_direction == F.normalize(torch.linalg.svd(G)[2][:, -1]) (extract_pointcloud.py:86-88) up to sign, against a
    float64 SVD: random matrices, nearly rank-1 matrices (the real case: 50 almost parallel gradients), k = 1, 2, 50, 128,
    exact rank deficiency (any unit vector of the null space) and the all-zero matrix (a unit vector, as with LAPACK's vh = I)."""
    from emap_amd.extraction import null_direction
    gen = torch.Generator().manual_seed(0)
    for k in (3, 50, 128):
        G = torch.randn(1000, k, 3, generator=gen)
        base = torch.nn.functional.normalize(torch.randn(1000, 1, 3, generator=gen), dim=-1) * 20.0
        G2 = base + 0.05 * torch.randn(1000, k, 3, generator=gen)                      # nearly rank 1
        for M in (G, G2):
            ref = torch.nn.functional.normalize(torch.linalg.svd(M.double())[2][:, -1, :], dim=1).float()
            s = torch.linalg.svdvals(M.double())
            ok = (s[:, 1] - s[:, 2]) > 1e-2 * s[:, 1]                                     # the two small singular values differ
            out = null_direction(M.to(DEV)).cpu()
            assert float((out.norm(dim=1) - 1).abs().max()) <= 1e-5
            assert int(ok.sum()) >= 900 and float(_dir_err_each(out, ref)[ok].max()) <= 1e-5, k
    # k < 3: the null space is at least one-dimensional; any unit vector orthogonal to the rows is right
    for k in (1, 2):
        M = torch.randn(64, k, 3, generator=gen)
        out = null_direction(M.to(DEV)).cpu()
        assert float((M @ out.unsqueeze(-1)).abs().max()) <= 1e-4 and float((out.norm(dim=1) - 1).abs().max()) <= 1e-5
    z = null_direction(torch.zeros(5, 50, 3, device=DEV)).cpu()       # LAPACK's vh is the identity here: some unit vector
    assert float((z.norm(dim=1) - 1).abs().max()) <= 1e-6
    assert null_direction(torch.zeros(0, 50, 3, device=DEV)).shape == (0, 3)


@pytest.mark.gpu
def test_extraction_points_vs_reference_golden():
    """get_udf_normals_slow (extract_pointcloud.py:98-193) through emap_amd.extraction with the reference's recorded jitter:
    values, normals and line directions against the reference's own outputs."""
    from emap_amd.extraction import get_udf_normals_slow
    g = load_golden("g10_extraction")
    net, state, cfg = mk("d8w256L10", "f16x3")
    df, normals, ld, samples = get_udf_normals_slow(net.udf, net.gradient, None, t(g["xyz"]), True, sampling_N=50,
                                                    sampling_delta=0.005, max_batch=128, device=DEV, noise=t(g["slow_noise"]))
    assert samples.shape == (300, 13)
    assert rel(df, t(g["slow_df"])) <= 1e-4
    assert float((normals.cpu() - t(g["slow_normals"])).abs().max()) <= 2e-4
    # line directions: compare where the reference's own direction is well conditioned (oracle singular values)
    ld_pts = (t(g["xyz"]).unsqueeze(1) + 0.005 * t(g["slow_noise"])).reshape(-1, 3)
    gr = O.udf_gradient_autograd(state, cfg, ld_pts)[:, 0].reshape(300, 50, 3)
    s = torch.linalg.svdvals(gr.double())
    ok = ((s[:, 1] - s[:, 2]) / s[:, 0] > 1e-3)
    assert int(ok.sum()) >= 150
    assert float(_dir_err_each(ld.cpu(), t(g["slow_ld"]))[ok].max()) <= 2e-3


@pytest.mark.gpu
def test_extraction_grid_vs_reference_golden():
    """get_udf_normals_grid (extract_pointcloud.py:5-95) on the reference's 12^3 case; also the generic path (callables that
    are not this package's UDFNetwork methods) must give the same answer as the batched fast path."""
    from emap_amd.extraction import get_udf_normals_grid
    g = load_golden("g10_extraction")
    net, state, cfg = mk("d8w256L10", "f16x3")
    N, thr = int(g["N"]), float(g["thr"])
    gdf = t(g["df"]).reshape(-1)
    gmask = gdf < thr
    df0 = get_udf_normals_grid(net.udf, net.gradient, N, -1.0, False, device=DEV)[0].reshape(-1).cpu()
    assert rel(df0, gdf) <= 1e-4
    omask = df0 < thr
    assert int((omask != gmask).sum()) <= 3                     # only points within 1e-4 of the threshold may flip
    # jitter rows follow the order of the thresholded points: align the recorded draws with OUR thresholded set
    row_of = torch.cumsum(gmask.long(), 0) - 1
    noise = torch.zeros(int(omask.sum()), 50, 3)
    both = omask & gmask
    noise[(torch.cumsum(omask.long(), 0) - 1)[both]] = t(g["grid_noise"])[row_of[both]]
    df, ld, vecs, samples, vs = get_udf_normals_grid(net.udf, net.gradient, N, thr, True, sampling_N=50, sampling_delta=0.005,
                                                     max_batch=256, device=DEV, noise=noise)
    assert df.shape == (N, N, N) and ld.shape == (N, N, N, 3) and vecs.shape == (N, N, N, 3) and samples.shape == (N ** 3, 12)
    assert float(vs) == float(g["voxel_size"])
    v, gv = vecs.reshape(-1, 3).cpu()[both], t(g["vecs"]).reshape(-1, 3)[both]
    assert float((v != gv).float().mean()) <= 0.01              # -sign(grad) per component (the reference's dim=1 quirk)
    sub = samples[:, :3].cpu()[both]
    ld_pts = (sub.unsqueeze(1) + 0.005 * t(g["grid_noise"])[row_of[both]]).reshape(-1, 3)
    gr = O.udf_gradient_autograd(state, cfg, ld_pts)[:, 0].reshape(-1, 50, 3)
    s = torch.linalg.svdvals(gr.double())
    ok = ((s[:, 1] - s[:, 2]) / s[:, 0] > 1e-3)
    assert int(ok.sum()) >= 100
    assert float(_dir_err_each(ld.reshape(-1, 3).cpu()[both], t(g["ld"]).reshape(-1, 3)[both])[ok].max()) <= 2e-3
    assert float(ld.reshape(-1, 3).cpu()[~omask].abs().max()) == 0.0
    # generic path: wrap the callables so that the fast path is not taken
    df2, ld2, vecs2, _, _ = get_udf_normals_grid(lambda p: net.udf(p), lambda p: net.gradient(p), N, thr, True, sampling_N=50,
                                                 sampling_delta=0.005, max_batch=256, device=DEV, noise=noise)
    assert rel(df2, df) <= 2e-6 and float((vecs2 != vecs).float().mean()) <= 0.01
    assert float(_dir_err_each(ld2.reshape(-1, 3).cpu()[both], ld.reshape(-1, 3).cpu()[both])[ok].max()) <= 2e-3


def test_extraction_through_the_runners_closure():
    """The call pattern of the real caller: Runner_UDF.extract_edge passes `udf_network.udf` and a CLOSURE that normalises
    `udf_network.gradient` (runner_udf.py:520-527).  Goldens recorded from the reference functions with that closure.  Also
    checks that the closure is driven with large launches (a handful of calls, not n/4096)."""
    from emap_amd.extraction import get_udf_normals_slow, get_udf_normals_grid
    g = load_golden("g10_extraction")
    net, state, cfg = mk("d8w256L10", "f16x3")
    calls = []

    def func_grad(xyz):                                     # runner_udf.py:522-526, verbatim
        calls.append(int(xyz.shape[0]))
        gradients = net.gradient(xyz)
        gradients_mag = torch.linalg.norm(gradients, ord=2, dim=-1, keepdim=True)
        gradients_norm = gradients / (gradients_mag + 1e-5)
        return gradients_norm

    xyz = t(g["xyz"])
    df, normals, ld, samples = get_udf_normals_slow(net.udf, func_grad, None, xyz, True, sampling_N=50, sampling_delta=0.005,
                                                    max_batch=128, device=DEV, noise=t(g["closure_slow_noise"]))
    assert calls == [300, 300 * 50]                          # two launches; the reference's schedule is 3 + 118 calls of <= 128 x 50 points
    assert rel(df, t(g["slow_df"])) <= 1e-4
    assert float((normals.cpu() - t(g["closure_slow_normals"])).abs().max()) <= 2e-4
    ld_pts = (xyz.unsqueeze(1) + 0.005 * t(g["closure_slow_noise"])).reshape(-1, 3)
    gr = O.udf_gradient_autograd(state, cfg, ld_pts)[:, 0]
    gr = (gr / (torch.linalg.norm(gr, dim=-1, keepdim=True) + 1e-5)).reshape(300, 50, 3)
    sv = torch.linalg.svdvals(gr.double())
    ok = ((sv[:, 1] - sv[:, 2]) / sv[:, 0] > 1e-3)
    assert int(ok.sum()) >= 150
    assert float(_dir_err_each(ld.cpu(), t(g["closure_slow_ld"]))[ok].max()) <= 2e-3
    # the same closure through the dense-grid routine: one launch per stage for 12^3 points
    calls.clear()
    N, thr = int(g["N"]), float(g["thr"])
    dfg, ldg, vecs, _, _ = get_udf_normals_grid(net.udf, func_grad, N, thr, True, sampling_N=50, sampling_delta=0.005, max_batch=256,
                                                device=DEV)
    n_thr = int((dfg.reshape(-1) < thr).sum())
    assert calls == [n_thr, n_thr * 50]
    gmask = t(g["df"]).reshape(-1) < thr
    both = gmask & (dfg.reshape(-1).cpu() < thr)
    v, gv = vecs.reshape(-1, 3).cpu()[both], t(g["closure_vecs"]).reshape(-1, 3)[both]
    assert float((v != gv).float().mean()) <= 0.01


# ---------------------------------------------------------------------------------------- full-image path (par. 8 f4)
@pytest.mark.parametrize("case", list(G5))
def test_reduced_output_render_vs_reference_golden(case):
    """The full-image launch mode (SURVEY par. 8 f4): only per-ray outputs are written.  Checked against the REFERENCE's
    per-ray results (goldens G5), and the weighted normal against the reference's own per-sample tensors the way
    Runner_UDF.validate reduces them (runner_udf.py:375-388)."""
    g = load_golden("g5_render_" + case)
    ns, ni, steps = [int(v) for v in g["cfg"]]
    net, _, _ = mk(G5[case], "f16x3")
    r = mk_renderer(net, ns, ni, steps)
    a = [t(g[k]).to(DEV) for k in ("rays_o", "rays_d", "near", "far", "depth_scale")]
    with torch.no_grad():
        out = r.render_reduced(*a, cos_anneal_ratio=1.0, perturb_overwrite=0, flip_saturation=0.9)
    torch.cuda.synchronize()
    r.check_errors()
    assert set(out) >= {"edge", "depth", "normals", "weight_sum"}
    assert rel(out["edge"], t(g["out.edge"])) <= 1e-4
    assert rel(out["depth"], t(g["out.depth"])) <= 3e-4
    ref_n = (t(g["out.gradients_flip"]) * t(g["out.weights"])[:, :, None]).sum(dim=1)
    assert rel(out["normals"], ref_n) <= 3e-4
    assert rel(out["normals"], t(g["out.normals"])) <= 3e-4


def test_image_render_is_chunk_invariant():
    """emap_amd.validation.render_image (the render loop of Runner_UDF.validate, runner_udf.py:297-407): rays are
    independent, so one launch of all rays must equal the reference's schedule of batch_size chunks - with the reference's
    per-chunk jitter draws (same CPU generator sequence).  Different launch sizes run different MLP kernel geometries
    (forward-mode / reverse-sweep, 4/8 waves), so the comparison is to the render tolerances, not bit-exact."""
    from emap_amd.validation import render_image, to_images
    from emap_amd import synthetic
    net, _, _ = mk("d8w256L10", "f16x3")
    r = mk_renderer(net, 64, 64, 4)
    H, W = 40, 50
    ro, rd, near, far, ds = synthetic.make_rays(H * W, seed=3)
    ro, rd, ds = ro.to(DEV), rd.to(DEV), ds.to(DEV)
    near_f, far_f = float(near.reshape(-1)[0]), float(far.reshape(-1)[0])
    for perturb in (0.0, 1.0):
        r.perturb = perturb
        torch.manual_seed(77)
        big = render_image(r, ro.reshape(H, W, 3), rd.reshape(H, W, 3), near_f, far_f, ds.reshape(H, W, 1), batch_size=512,
                           cos_anneal_ratio=1.0, launch_rays=4096)
        torch.manual_seed(77)
        ref_e, ref_d, ref_n = [], [], []
        for h in range(0, H * W, 512):                       # the reference's loop: one render() per batch_size chunk
            with torch.no_grad():
                o = r.render(ro[h:h + 512], rd[h:h + 512], near_f, far_f, depth_scale=ds[h:h + 512], cos_anneal_ratio=1.0)
            ref_e.append(o["edge"]); ref_d.append(o["depth"])
            S = r.n_samples + r.n_importance
            ref_n.append((o["gradients_flip"] * o["weights"][:, :S, None]).sum(dim=1))   # runner_udf.py:375-388
        assert big["edge"].shape == (H * W, 1) and big["normals"].shape == (H * W, 3)
        assert rel(t(big["edge"]), torch.cat(ref_e)) <= 2e-4, perturb
        assert rel(t(big["depth"]), torch.cat(ref_d)) <= 5e-4, perturb
        assert rel(t(big["normals"]), torch.cat(ref_n)) <= 5e-4, perturb
    e, d, nrm = to_images(big, H, W)
    assert e.shape == (H, W) and e.dtype.name == "uint8" and d.shape == (H, W) and nrm.shape == (H, W, 3)


def test_validation_loop_consumption_through_the_reduced_mode():
    """What emap_amd.dropin.validate_wrapper switches on: render() with ``inference_reduced`` serves the three things Runner_UDF.validate
    reads (edge, depth, sum_s gradients_flip * weights - runner_udf.py:333-407, restated here as the consumer) from the 28 B/ray launch
    mode, equal to the full render's."""
    net, _, _ = mk("d8w256L10", "f16x3")
    r = mk_renderer(net, 64, 64, 4)
    from emap_amd import synthetic
    ro, rd, near, far, ds = [x.to(DEV) for x in synthetic.make_rays(512, seed=5)]
    S = r.n_samples + r.n_importance
    consume = lambda o: (o["edge"], o["depth"], ((o["gradients_flip"] if o.get("gradients_flip") is not None else o["gradients"])
                                                 * o["weights"][:, :S, None]).sum(dim=1))
    with torch.no_grad():
        full = consume(r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, perturb_overwrite=0))
        r.inference_reduced = True
        o = r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, perturb_overwrite=0)
        red = consume(o)
        r.inference_reduced = False
    assert o["reduced"] and o["inside_sphere"] is not None and o["gradients"] is None
    for a, b, tol in zip(red, full, (1e-6, 1e-6, 2e-6)):
        assert rel(a, b) <= tol
    # the per-sample entries of the reduced mode are guards: any use but the loop's own product-sum raises
    for bad in (lambda: o["weights"].shape, lambda: o["weights"].sum(), lambda: o["weights"][:, :4, None], lambda: o["gradients_flip"] * 2.0,
                lambda: o["gradients_flip"].cpu(), lambda: o["inside_sphere"].float(), lambda: (o["gradients_flip"] * o["weights"][:, :S, None]).sum(dim=2),
                lambda: o["gradients_flip"] * o["weights"]):
        with pytest.raises(RuntimeError):
            bad()
    # with trainable parameters and autograd on, the flag is ignored (the training path needs the per-sample tensors)
    r.inference_reduced = True
    o2 = r.render(ro[:32], rd[:32], near[:32], far[:32], ds[:32], cos_anneal_ratio=1.0, perturb_overwrite=0)
    assert "reduced" not in o2 and o2["weights"].shape == (32, S)


def test_captured_graph_survives_renders_of_other_shapes():
    """Advisor r2: a captured render graph has the device pointers of its workspaces, near/far constants and packed weights baked in;
    a render of another batch shape (validation chunks between training-graph replays), a re-pack after a parameter update and an
    allocator trim in between must leave them alive and in place."""
    from emap_amd import synthetic
    net, _, _ = mk("d8w256L10", "f16x3")
    r = mk_renderer(net, 64, 64, 4)
    ro, rd, near, far, ds = [x.to(DEV) for x in synthetic.make_rays(512, seed=9)]
    tr = synthetic.make_t_rand(512, seed=3).to(DEV)
    g = r.capture(ro, rd, 0.05, 6.0, ds, cos_anneal_ratio=1.0, t_rand=tr)
    ref = {k: v.clone() for k, v in g().items() if isinstance(v, torch.Tensor)}
    packed_ptr = net.packed("f16x3").data_ptr()
    with torch.no_grad():
        for n, nf in ((8192, (0.05, 6.0)), (100, (0.1, 5.0)), (4096, (0.05, 6.0))):      # other shapes, other near/far constants
            o = synthetic.make_rays(n, seed=n)
            r.render(o[0].to(DEV), o[1].to(DEV), nf[0], nf[1], o[4].to(DEV), cos_anneal_ratio=1.0, perturb_overwrite=0)
            r.render_reduced(o[0].to(DEV), o[1].to(DEV), nf[0], nf[1], o[4].to(DEV), cos_anneal_ratio=1.0, perturb_overwrite=0)
    net.invalidate_packed()
    assert net.packed("f16x3").data_ptr() == packed_ptr                     # re-packed in place
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 22,), float("nan"), device=DEV) for _ in range(64)]    # whatever was freed is overwritten now
    del junk
    again = g()
    torch.cuda.synchronize()
    r.check_errors()
    for k, v in ref.items():
        assert torch.equal(again[k], v), k


def test_profile_hooks_report_kernel_time_and_shader_clock():
    """emap_profile_enable / read_kernel / read_clock (include/emap_hip.h): HIP events around the final value+gradient pass of emap_render_fwd on
    its stream, and the shader clock that kernel's workgroup 0 saw (s_memtime over s_memrealtime); results are unaffected by the hooks."""
    net, state, cfg = mk("d8w256L10")
    r = mk_renderer(net, 64, 64, 4)
    L = _lib.lib()
    ro, rd, near, far, ds = (v.to(DEV) for v in synthetic.make_rays(512, seed=1))
    tr = torch.zeros(512, 1, device=DEV)

    def render():
        with torch.no_grad():
            return r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
    out0 = render()
    _lib.check(L.emap_profile_enable(1))
    try:
        for _ in range(3):
            out1 = render()
        torch.cuda.synchronize()
    finally:
        _lib.check(L.emap_profile_enable(0))
    ms, n, mhz = C.c_float(), C.c_int(), C.c_float()
    _lib.check(L.emap_profile_read_kernel(0, C.byref(ms), C.byref(n)))
    _lib.check(L.emap_profile_read_clock(0, C.byref(mhz)))
    assert n.value == 3 and 0.05 < ms.value / 3 < 5.0            # a few hundred microseconds per launch
    assert 500.0 < mhz.value < 2600.0                            # power-capped well below the 2.4 GHz nominal clock on real data
    out2 = render()                                              # hooks off again: the kernel gets no clock buffer
    for k in ("edge", "depth", "udf", "gradients"):
        assert torch.equal(out0[k], out1[k]) and torch.equal(out0[k], out2[k])
