"""GPU tests added in round 6 (-m gpu), all through the C ABI / the drop-in classes."""
import copy

import numpy as np
import pytest
import torch

import emap_amd
from emap_amd import _lib, synthetic
from emap_amd.parallel import Trainer, FusedAdam

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


# ------------------------------------------------------------------------------------------------ FusedAdam after a resume
def test_fused_adam_follows_lr_changes_after_load_state_dict():
    """ADVICE r5 (high).  The resume path of the drop-in (runner_udf.py:260-273: load_checkpoint -> optimizer.load_state_dict, then
    update_learning_rate writes param_groups[i]['lr'] every step): load a checkpoint, CHANGE the learning rates, step - equal to
    torch.optim.Adam doing the same."""
    from test_gpu_round5 import _groups, _clone, _mk, _steps
    gen = torch.Generator().manual_seed(31)
    geo_a, tail_a = _groups(gen)
    geo_b, tail_b = _clone(geo_a), _clone(tail_a)
    oa, ob = _mk(FusedAdam, geo_a, tail_a), _mk(torch.optim.Adam, geo_b, tail_b)
    _steps([(oa, geo_a + tail_a), (ob, geo_b + tail_b)], torch.Generator().manual_seed(32), 3)
    sd = copy.deepcopy(oa.state_dict())
    geo_c, tail_c = _clone(geo_a), _clone(tail_a)
    oc = _mk(FusedAdam, geo_c, tail_c)
    oc.load_state_dict(sd)
    for s in range(3):
        for opt in (oc, ob):                       # the runner's scheduler: geo group its own lr, every other group the common one
            for i, g in enumerate(opt.param_groups):
                g["lr"] = (2e-3 if i == 0 else 7e-3) * (0.5 ** s)
        _steps([(oc, geo_c + tail_c), (ob, geo_b + tail_b)], torch.Generator().manual_seed(40 + s), 1, s0=3 + s)
    for pc, pb in zip(geo_c + tail_c, geo_b + tail_b):
        assert torch.allclose(pc, pb, rtol=3e-6, atol=1e-7), (pc, pb)
    # and from_adam on an Adam that has already stepped (it goes through load_state_dict)
    geo_d, tail_d = _clone(geo_b), _clone(tail_b)
    od_ref = _mk(torch.optim.Adam, geo_d, tail_d)
    od_ref.load_state_dict(copy.deepcopy(ob.state_dict()))
    od = FusedAdam.from_adam(od_ref)
    for opt in (od, ob):
        for i, g in enumerate(opt.param_groups):
            g["lr"] = 3e-3 if i == 0 else 4e-3
    _steps([(od, geo_d + tail_d), (ob, geo_b + tail_b)], torch.Generator().manual_seed(50), 2, s0=6)
    for pd_, pb in zip(geo_d + tail_d, geo_b + tail_b):
        assert torch.allclose(pd_, pb, rtol=3e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------------ the 32x32 forward sweep as a value kernel (ABI 10, opt-in)
@pytest.mark.parametrize("prec", ["f16x3", "f16x3e", "bf16x3"])
def test_value_tile_mode_32x32_forward_sweep_vs_oracle_and_the_16x16_kernel(prec):
    """emap_set_value_tile_mode(1): value launches of >= 512 tiles of 64 points run udf_mlp_rev32_kernel<.., VAL> (forward sweep only).  udf
    against the fp64 oracle at the gate of the 16x16 kernel, against that kernel to fp32 rounding, ragged launch sizes around the
    switch-over, a launch larger than the resident grid; below 512 tiles the switch changes nothing (bit for bit); f16x3m ignores it."""
    from test_gpu_parity import mk
    from conftest import net_state
    from oracle import emap_oracle as O
    net, state, cfg = mk("d8w256L10", prec)
    L = _lib.lib()
    gen = torch.Generator().manual_seed(77)
    x = (torch.rand(300001, 3, generator=gen) * 2 - 1)
    xd = x.to(DEV)
    ref = O.udf_value_and_grad({k: v.double() for k, v in state.items()}, cfg, x[:3000].double())[0].float().reshape(-1)
    tol = 2e-5 if prec == "bf16x3" else 2e-6
    try:
        for P in (32768 - 63, 32768 - 64, 32769, 40037, 300001):
            outs = {}
            for mode in (0, 1):
                L.emap_set_value_tile_mode(mode)
                with torch.no_grad():
                    outs[mode] = net.hip_udf(xd[:P])[0].reshape(-1).clone()
            torch.cuda.synchronize()
            d = float((outs[0] - outs[1]).abs().max() / outs[0].abs().max())
            if P <= 32768 - 64:          # 511 tiles of 64: the 16x16 kernel either way
                assert torch.equal(outs[0], outs[1]), P
            else:
                assert 0 < d <= 3 * tol, (P, d)         # another kernel (not bit-equal), the same function
                e = float((outs[1][:3000].cpu() - ref).abs().max() / ref.abs().max())
                assert e <= tol, (P, e)
        netm, _, _ = mk("d8w256L10", "f16x3m")
        o = {}
        for mode in (0, 1):
            L.emap_set_value_tile_mode(mode)
            with torch.no_grad():
                o[mode] = netm.hip_udf(xd[:40037])[0].clone()
        assert torch.equal(o[0], o[1])
    finally:
        L.emap_set_value_tile_mode(0)


@pytest.mark.parametrize("prec", ["f16x3", "bf16x3"])
def test_value_launch_geometry_table_is_bit_identical_across_its_ranges(prec):
    """Round 6: the value launches pick their tile geometry from a measured table (udf_mlp_kernel.inc:launch_mlp_fs2_mode: 16- / 32- / 48- / 64-point tiles
    as one 8-wave workgroup per CU, 48- / 64-point tiles as two 4-wave workgroups per CU, the 8-wave forms again in three rounds).  Every geometry sums
    in the same K order: a point's udf is the same bit for bit whichever launch size carries it - both sides of every boundary of the table, ragged
    last tiles included."""
    from test_gpu_parity import mk
    from oracle import emap_oracle as O
    net, state, cfg = mk("d8w256L10", prec)
    gen = torch.Generator().manual_seed(123)
    x = (torch.rand(70000, 3, generator=gen) * 2.4 - 1.2).to(DEV)
    with torch.no_grad():
        base = net.hip_udf(x[:4096])[0].clone()                  # 16-point tiles, one 8-wave workgroup per CU
        full = net.hip_udf(x)[0].clone()                         # 70 000 points: 64-point tiles, 4-wave workgroups
        for P in (4097, 8192, 8193, 8256, 10240, 12288, 12289, 12345, 16321, 16384, 16385, 20001, 24576, 24577, 32768, 32769, 36864, 36865, 49152, 49153):
            u = net.hip_udf(x[:P])[0]
            assert u.shape == (P, 1)
            assert torch.equal(u[:4096], base) and torch.equal(u, full[:P]), P
    ur = O.udf_value_and_grad(state, cfg, x[8000:9000].cpu())[0]
    with torch.no_grad():
        u = net.hip_udf(x[:12345])[0]
    assert float((u[8000:9000].cpu() - ur).abs().max() / ur.abs().max()) <= (5e-5 if prec == "bf16x3" else 1e-5)


def test_render_with_the_32x32_coarse_pass_vs_reference_golden_and_its_arrival_counters():
    """A 512-ray render whose coarse pass is the 32x32 forward sweep (it is the render's first launch: it clears the arrival counters of the
    fused compositing tail) - twice back to back (stale counters would hang or corrupt the second), edge / depth within the bounds of the
    default path against the 16x16 coarse pass, and every ray composited."""
    from test_gpu_parity import mk, mk_renderer, rel
    net, _, _ = mk("d8w256L10", "f16x3")
    r = mk_renderer(net, 64, 64, 4)
    ro, rd, near, far, ds = [v.to(DEV) for v in synthetic.make_rays(600, seed=9)]
    L = _lib.lib()
    outs = {}
    try:
        for mode in (0, 1, 1):
            L.emap_set_value_tile_mode(mode)
            with torch.no_grad():
                o = r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, perturb_overwrite=0, flip_saturation=0.9)
            torch.cuda.synchronize()
            r.check_errors()
            outs.setdefault(mode, []).append({k: o[k].clone() for k in ("edge", "depth", "weight_sum", "z_vals", "udf")})
    finally:
        L.emap_set_value_tile_mode(0)
    a, b, c = outs[0][0], outs[1][0], outs[1][1]
    for k in a:
        assert torch.equal(b[k], c[k]), k                      # deterministic, counters cleared by the new first launch
    assert rel(b["edge"], a["edge"]) <= 1e-4 and rel(b["depth"], a["depth"]) <= 3e-4
    same = float((b["z_vals"] == a["z_vals"]).all(dim=1).float().mean())
    assert same >= 0.80, same                                   # an ulp of the coarse udf re-samples a few rays (as between two CPUs: DESIGN par. 4)
    assert float((b["weight_sum"] - a["weight_sum"]).abs().max()) <= 2e-3


# ------------------------------------------------------------------------------------------------ fused importance sampling, widened
def _fused_vs_chain(netname, N, prec, n_samples, n_importance, steps, fused_mode=1):
    """importance_sample (udf_renderer_blending.py:802-841) as ONE launch against the chain of 2 K - 1 launches: every rendered tensor
    identical bit for bit (same device code, same arithmetic; tests/test_gpu_parity.py has the m = 16 shapes of rounds 5)."""
    from test_gpu_parity import mk, mk_renderer
    net, _, _ = mk(netname, prec)
    r = mk_renderer(net, n_samples, n_importance, steps)
    ro, rd, near, far, ds = [v.to(DEV) for v in synthetic.make_rays(N, seed=5)]
    tr = synthetic.make_t_rand(N).to(DEV)
    L = _lib.lib()
    outs = {}
    try:
        for fused in (fused_mode, 0):
            L.emap_set_fused_sampling(fused)
            with torch.no_grad():
                o1 = r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
                o2 = r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, perturb_overwrite=0, flip_saturation=0.9)
            torch.cuda.synchronize()
            r.check_errors()
            outs[fused] = {tag + k: v.clone() for tag, o in (("jitter.", o1), ("plain.", o2)) for k, v in o.items() if isinstance(v, torch.Tensor)}
    finally:
        L.emap_set_fused_sampling(1)
    a, b = outs[fused_mode], outs[0]
    assert set(a) == set(b) and "jitter.z_vals" in a
    assert a["jitter.z_vals"].shape == (N, n_samples + n_importance)
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("netname,N,ns,ni,K", [
    ("d8w256L10", 1024, 64, 50, 5),      # confs/ABC.conf:31,108-111 - the reference's own default launch shape: m = 10 new samples per step
    ("d8w256L10", 512, 64, 50, 5), ("d8w256L10", 77, 64, 50, 5), ("d8w256L10", 1, 64, 50, 5),
    ("d8w256L10", 300, 64, 64, 8),       # m = 8
    ("d8w256L10", 256, 32, 15, 3),       # m = 5, odd list lengths
    ("d8w256L10", 130, 64, 13, 1),       # m = 13, a single step
    ("d4w128L10", 512, 64, 50, 5), ("d4w128L10", 50, 32, 30, 3),
])
def test_fused_importance_sampling_with_fewer_than_16_new_samples_per_step(netname, N, ns, ni, K):
    """VERDICT r5 missing #2 / item 4: the fused kernel refused every m != 16 - the reference's default shape among them.  (Since the size rule
    of launch_is_mode hands m <= 12 beyond 512 rays back to the chain - measured faster - the fused kernel is FORCED here, mode 2;
    whatever the rule picks, mode 1, must be bit-identical as well.)"""
    _fused_vs_chain(netname, N, "f16x3", ns, ni, K, fused_mode=2)
    _fused_vs_chain(netname, N, "f16x3", ns, ni, K, fused_mode=1)


@pytest.mark.parametrize("N,ns,ni,K", [(2048, 64, 64, 4), (4096, 64, 64, 4), (2500, 64, 50, 5)])
def test_fused_importance_sampling_at_2048_rays_and_more(N, ns, ni, K):
    """From 2048 rays on the launcher's size rule may prefer the chain (64-point tiles); emap_set_fused_sampling(2) runs the fused
    kernel there too - and whichever the default rule (mode 1) picks is bit-identical to the chain as well."""
    _fused_vs_chain("d8w256L10", N, "f16x3", ns, ni, K, fused_mode=2)
    _fused_vs_chain("d8w256L10", N, "f16x3", ns, ni, K, fused_mode=1)


@pytest.mark.parametrize("prec", ["f16x3m", "f16x3e", "bf16x3", "bf16"])
def test_fused_importance_sampling_m10_in_other_precision_modes(prec):
    _fused_vs_chain("d8w256L10", 512, prec, 64, 50, 5, fused_mode=2)
    _fused_vs_chain("d8w256L10", 700, prec, 64, 50, 5, fused_mode=1)       # beyond 512 rays at m = 10: the split modes take the chain (size rule)


# ------------------------------------------------------------------------------------------------ compositing inside the value + grad_x kernel
def _render_both_ways(r, args, kw, reduced=False):
    L = _lib.lib()
    outs = {}
    try:
        for fused in (1, 0):
            L.emap_set_fused_composite(fused)
            with torch.no_grad():
                o = (r.render_reduced if reduced else r.render)(*args, **kw)
            torch.cuda.synchronize()
            r.check_errors()
            outs[fused] = {k: v.clone() for k, v in o.items() if isinstance(v, torch.Tensor)}
    finally:
        L.emap_set_fused_composite(1)
    return outs


@pytest.mark.parametrize("netname,N,ns,ni,K", [
    ("d8w256L10", 512, 64, 64, 4),       # the benchmark batch: a ray = two 64-point tiles, written by two workgroups
    ("d8w256L10", 1024, 64, 50, 5),      # 114 samples per ray: tiles straddle rays
    ("d8w256L10", 100, 64, 64, 4), ("d8w256L10", 333, 32, 32, 4), ("d8w256L10", 1500, 64, 64, 4),
    ("d8w256L10", 700, 16, 16, 2),       # 32 samples per ray: two rays per tile
    ("d8w256L10", 90, 64, 192, 4),       # 256 samples per ray (C = 4)
    ("d8w256L10", 4096, 64, 64, 4),      # four rounds of workgroups
    ("d4w128L10", 512, 64, 64, 4), ("d4w128L10", 400, 32, 30, 3),
])
def test_fused_compositing_equals_the_separate_launch_bit_for_bit(netname, N, ns, ni, K):
    """ABI 9 / BASELINE config C2 ("fused MLP + composite kernel"): render_core's tail (udf_renderer_blending.py:463-625) runs inside
    udf_mlp_rev32_kernel<..., COMP> - the workgroup that completes a ray composites it - against the separate composite_kernel launch
    (emap_set_fused_composite(0)): every entry of the render dict identical bit for bit, also when tiles straddle rays, when a tile holds
    several rays, and over several rounds of workgroups."""
    from test_gpu_parity import mk, mk_renderer
    net, _, _ = mk(netname, "f16x3")
    r = mk_renderer(net, ns, ni, K)
    ro, rd, near, far, ds = [v.to(DEV) for v in synthetic.make_rays(N, seed=9)]
    tr = synthetic.make_t_rand(N).to(DEV)
    assert N * (ns + ni) >= 10240                                  # the reverse-sweep kernel (the only one with the fused tail) runs
    kw = dict(cos_anneal_ratio=0.7, flip_saturation=0.9, t_rand=tr)
    outs = _render_both_ways(r, (ro, rd, near, far, ds), kw)
    assert set(outs[0]) == set(outs[1]) and {"edge", "weights", "gradient_error", "normals", "depth"} <= set(outs[1])
    for k in outs[1]:
        assert torch.equal(outs[1][k], outs[0][k]), k
    red = _render_both_ways(r, (ro, rd, near, far, ds), kw, reduced=True)
    for k in red[1]:
        assert torch.equal(red[1][k], red[0][k]), k
    assert torch.equal(red[1]["edge"], outs[1]["edge"])


@pytest.mark.parametrize("prec", ["f16x3e", "f16x3m", "bf16x3", "f16", "bf16"])
def test_fused_compositing_in_every_precision_mode(prec):
    from test_gpu_parity import mk, mk_renderer
    net, _, _ = mk("d8w256L10", prec)
    r = mk_renderer(net, 64, 64, 4)
    N = 512
    ro, rd, near, far, ds = [v.to(DEV) for v in synthetic.make_rays(N, seed=3)]
    outs = _render_both_ways(r, (ro, rd, near, far, ds), dict(cos_anneal_ratio=1.0, flip_saturation=0.9, perturb_overwrite=0))
    for k in outs[1]:
        assert torch.equal(outs[1][k], outs[0][k]), (prec, k)


def test_fused_compositing_from_a_replayed_graph_and_back_to_back():
    """The arrival counters are cleared by the render's FIRST launch: 40 replays of a captured render and 40 eager renders back to back
    (no synchronisation between them) give the same dict every time."""
    from test_gpu_parity import mk, mk_renderer
    net, _, _ = mk("d8w256L10", "f16x3")
    r = mk_renderer(net, 64, 64, 4)
    N = 512
    ro, rd, near, far, ds = [v.to(DEV) for v in synthetic.make_rays(N, seed=4)]
    tr = synthetic.make_t_rand(N).to(DEV)
    kw = dict(cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
    ref = _render_both_ways(r, (ro, rd, near, far, ds), kw)[0]
    g = r.capture(ro, rd, near, far, ds, **kw)
    for i in range(40):
        o = g()
        if i % 13 == 0 or i == 39:
            for k in ("edge", "weights", "depth", "normals", "gradient_error"):
                assert torch.equal(o[k], ref[k]), (i, k)
    keep = []
    with torch.no_grad():
        for i in range(40):
            keep.append(r.render(ro, rd, near, far, ds, **kw))
    torch.cuda.synchronize()
    for o in keep[::7] + keep[-1:]:
        for k in ("edge", "weights", "depth", "normals", "gradient_error"):
            assert torch.equal(o[k], ref[k]), k


def test_training_forward_uses_the_fused_tail_and_the_gradients_do_not_change():
    from test_gpu_parity import mk, mk_renderer
    L = _lib.lib()
    res = {}
    try:
        for fused in (1, 0):
            L.emap_set_fused_composite(fused)
            net, _, _ = mk("d8w256L10", "f16x3")
            r = mk_renderer(net, 64, 64, 4)
            N = 256
            ro, rd, near, far, ds = [v.to(DEV) for v in synthetic.make_rays(N, seed=6)]
            te = synthetic.make_true_edge(N, seed=7).to(DEV)
            out = r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, perturb_overwrite=0, flip_saturation=0.9)
            loss = emap_amd.EdgeLoss("mse")(out["edge"], te) + 0.1 * out["gradient_error"]
            loss.backward()
            torch.cuda.synchronize()
            r.check_errors()
            res[fused] = (float(loss), torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone())
    finally:
        L.emap_set_fused_composite(1)
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])


# ------------------------------------------------------------------------------------------------ full image in one call (SURVEY par. 8 f4)
def test_whole_image_in_one_call_equals_the_chunked_schedules():
    """render_image's default (round 6): ONE emap_render_fwd call over all H*W rays (runner_udf.py:298-327 loops over batch_size chunks).
    Rays are independent: the same image as 1024-ray launches bit for bit (the same kernels run at both sizes: >= 10 240 points), and as
    the reference's 512-ray schedule to the render tolerances."""
    from test_gpu_parity import mk, mk_renderer, rel, t
    from emap_amd.validation import render_image
    net, _, _ = mk("d8w256L10", "f16x3")
    r = mk_renderer(net, 64, 64, 4)
    H, W = 96, 100                       # 9600 rays: the sampler's launch chain (>= 2048 rays) and several rounds of every grid
    ro, rd, near, far, ds = synthetic.make_rays(H * W, seed=13)
    ro, rd, ds = ro.to(DEV), rd.to(DEV), ds.to(DEV)
    nf, ff = float(near.reshape(-1)[0]), float(far.reshape(-1)[0])
    r.perturb = 1.0
    imgs = {}
    for lr in (None, 4096, 512):
        torch.manual_seed(5)
        imgs[lr] = render_image(r, ro.reshape(H, W, 3), rd.reshape(H, W, 3), nf, ff, ds.reshape(H, W, 1), batch_size=512,
                                cos_anneal_ratio=1.0, launch_rays=lr, to_numpy=False)
    torch.cuda.synchronize()
    r.check_errors()
    for k in ("edge", "depth", "normals"):
        assert imgs[None][k].shape[0] == H * W
        assert torch.equal(imgs[None][k], imgs[4096][k]), k
        assert rel(imgs[None][k], imgs[512][k]) <= 5e-4, k


# ------------------------------------------------------------------------------------------------ f16x3e: 24-bit sigma' stash
def test_mode_f16x3e_sits_on_the_fp32_floor_element_wise():
    """the unorm16 sigma' stash was what separated the reverse sweep from fp32 autograd element-wise (p99.9 of
    grad_x 5.7e-3 against 6.6e-4 for the reference's own fp32 arithmetic, profiles/r05_elementwise_attribution.txt).  In precision mode
    f16x3e sigma' now travels as 24-bit fixed point (udf_mlp_rev32.inc, SG24: +50 % stash bytes in that mode only).  Against the fp64 oracle
    on 8192 random points: max-normalised <= 3e-6 (round 5: 1.3e-5), element-wise p99.9 <= 1.2e-3 (measured 5.9e-4; round 5: 6.7e-3) - the
    forward-mode kernel (no stash at all) gives 7.4e-7 / 4.1e-4.  And on the reference's own recorded points (g2)."""
    from conftest import net_state, load_golden
    from oracle import emap_oracle as O
    from test_gpu_round5 import _rel, _p999
    kw, state = net_state("d8w256L10")
    cfg = O.UDFConfig(d_hidden=256, n_layers=8, multires=10)
    x = torch.rand(65536, 3, generator=torch.Generator().manual_seed(9)) * 2 - 1
    uo, go = O.udf_value_and_grad({k: v.double() for k, v in state.items()}, cfg, x[:8192].double())
    n = emap_amd.UDFNetwork(precision="f16x3e", **kw)
    n.load_state_dict(state)
    n = n.to(DEV)
    with torch.no_grad():
        u, g = n.hip_udf(x.to(DEV), with_grad=True)                 # 65 536 points: the reverse sweep
        u1, g1 = n.hip_udf(x.to(DEV), with_grad=True)
    assert torch.equal(u, u1) and torch.equal(g, g1)
    e_u, e_g, p999 = _rel(u[:8192], uo), _rel(g[:8192], go), _p999(g[:8192], go)
    print(f"f16x3e vs fp64 oracle: udf {e_u:.2e}, grad_x max-normalised {e_g:.2e}, element-wise p99.9 {p999:.2e}")
    assert e_u <= 2e-6 and e_g <= 3e-6 and p999 <= 1.2e-3
    gold = load_golden("g2_mlp")
    xg = torch.from_numpy(gold["x"])
    reps = (10240 + xg.shape[0] - 1) // xg.shape[0]                 # enough points for the reverse-sweep kernel
    with torch.no_grad():
        ug, gg = n.hip_udf(xg.repeat(reps, 1).to(DEV), with_grad=True)
    assert _rel(gg[:xg.shape[0]], torch.from_numpy(gold["d8w256L10.grad"]).reshape(-1, 3)) <= 6e-6
    assert _rel(ug[:xg.shape[0]], torch.from_numpy(gold["d8w256L10.udf"])) <= 2e-6


# ------------------------------------------------------------------------------------------------ f16x3e: precise weight gradients
@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_f16x3e_parameter_gradients_meet_1e4_on_the_reference_samples(ci):
    """dL/dtheta of the default mode is 5.9e-4 of each tensor's maximum (wgrad multiplies the f16 hi parts of its
    operands; gate 1e-3).  Precision mode f16x3e (no MX fp6 anywhere, 24-bit sigma') now also stashes the lo parts of both operand sets and
    forms dW = Z_hi A_hi^T + (Z_hi A_lo^T + Z_lo A_hi^T) / 2^11 in three passes of the weight-gradient kernel: north_star's 1e-4 on all 27
    (15) tensors of the four g6 cases recorded from the reference's own loss.backward()."""
    from conftest import load_golden
    from test_gpu_backward import _render_bwd_on_reference_samples, _cmp
    from test_gpu_parity import t
    g = load_golden(f"g6_training_{ci}")
    loss, got, extra = _render_bwd_on_reference_samples(g, "f16x3e")
    assert loss == pytest.approx(float(g["loss"]), rel=1e-4)
    ref = {k[5:]: t(g[k]) for k in g if k.startswith("grad.lin")}
    w = _cmp(got, ref, 1e-4, f"g6_{ci} f16x3e")
    print(f"g6_training_{ci}, f16x3e: worst rel-to-max error of dL/dtheta {w:.2e}")
    gmax = max(float(v.abs().max()) for v in ref.values())
    for k in ("variance", "beta", "gamma"):
        r_ = float(t(g["grad." + k]))
        assert abs(float(extra[k]) - r_) <= 1e-3 * abs(r_) + 1e-6 * gmax, (k, float(extra[k]), r_)


# ------------------------------------------------------------------------------------------------ C5 in miniature: against a reference-trained model
@pytest.mark.timeout(1200)
def test_hip_training_run_matches_the_reference_trained_model_on_a_held_out_view():
    """BASELINE config C5 ("full training loop ... edge-accuracy parity vs reference checkpoint"), in miniature and
    without a dataset: golden g15 is a 1000-step training run of the REFERENCE's own classes (tests/golden/make_goldens.py:g15_convergence -
    d8 w256 from the reference's seeded geometric initialisation, the runner's Adam groups and schedules, 256 rays per step from 7 views
    of a multi-view consistent wire frame) and its render of the held-out eighth view.  The same run on the HIP path (native Trainer: fused
    forward, hand-written backward, fused Adam), same batches, same schedules, then the same held-out render:
      * the loss curve tracks the reference's: means over windows of 100 steps within 25 % everywhere (measured: up to 18 % apart in the middle of
        the run - two trajectories through a chaotic phase; the sampler's discontinuity makes single steps differ) and within 5 % over the last
        300 steps (measured 0.01 ... 0.4 %: both runs settle on the same curve),
      * the held-out view's PSNR against the ground-truth edge map is within 0.5 dB of the reference-trained model's,
      * the two models' held-out edge maps agree to RMS 0.04 of a [0, 1] image (measured 0.025: a quarter of either model's own RMS error against
        the ground truth, 0.105 at 19.6 dB - two optimisation trajectories, not two evaluations of one model)."""
    from conftest import load_golden
    from test_gpu_parity import mk_renderer, t
    g = load_golden("g15_convergence")
    ns, ni, steps_up = [int(v) for v in g["cfg"]]
    N, n_steps = int(g["n_rays"]), int(g["n_steps"])
    n_views, HW, held_out = [int(v) for v in g["scene"]]
    lr, lr_geo, alpha, warm_up_end, end_iter, anneal_end = [float(v) for v in g["schedule"]]
    ew, igr, igr_ns = [float(v) for v in g["weights3"]]
    meta, edges = synthetic.make_wireframe_scene(n_images=n_views, H=HW, W=HW)
    kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
    torch.manual_seed(int(g["init_seed"]))                       # the reference's constructor under the same seed (g9 pins the RNG consumption)
    net = emap_amd.UDFNetwork(scale=1.0, geometric_init=True, weight_norm=True, udf_type="abs", precision="f16x3", **kw)
    for k, v in net.state_dict().items():
        assert float(v.double().abs().sum()) == pytest.approx(float(g["init." + k + ".abs_sum"]), rel=1e-12), k
    net = net.to(DEV)
    r = mk_renderer(net, ns, ni, steps_up)
    tr = Trainer(r, lr_geo=lr_geo, lr=lr, edge_weight=ew, igr_weight=igr, igr_ns_weight=igr_ns)
    near, far = float(meta["scene_box"]["near"]), float(meta["scene_box"]["far"])
    losses = []
    for it in range(n_steps):
        tr.optimizer.param_groups[0]["lr"], tr.optimizer.param_groups[1]["lr"] = float(g["lrs"][it][0]), float(g["lrs"][it][1])
        img, px, py = synthetic.convergence_batch(meta, edges, N, seed=int(g["batch_seed0"]) + it, held_out=held_out)
        ro, rv, ds, true_edge = synthetic.scene_rays(meta, edges, img, px, py)
        b = {"rays_o": ro.to(DEV), "rays_d": rv.to(DEV), "near": torch.full((N, 1), near, device=DEV), "far": torch.full((N, 1), far, device=DEV),
             "depth_scale": ds.to(DEV), "cos_anneal_ratio": float(np.min([1.0, it / anneal_end])), "flip_saturation": 0.0, "perturb_overwrite": 0}
        losses.append(tr.step(b, true_edge.to(DEV)))
    tr.check_errors()
    hl = torch.stack(losses)[:, 0].cpu().double().numpy()
    rl = np.asarray(g["loss"], dtype=np.float64)
    win = 100
    hm, rm = hl.reshape(-1, win).mean(1), rl.reshape(-1, win).mean(1)
    print("loss, means over windows of 100 steps  reference:", np.round(rm, 4), " HIP:", np.round(hm, 4))
    assert np.all(np.abs(hm - rm) <= 0.25 * rm) and np.all(np.abs(hm - rm)[-3:] <= 0.05 * rm[-3:])
    # the held-out view, rendered like the reference rendered it (perturb_overwrite = 0, cos_anneal_ratio = 1)
    ys, xs = np.mgrid[0:HW, 0:HW]
    ro, rv, ds, gt = synthetic.scene_rays(meta, edges, held_out, xs.reshape(-1), ys.reshape(-1))
    n = ro.shape[0]
    with torch.no_grad():
        o = r.render(ro.to(DEV), rv.to(DEV), torch.full((n, 1), near, device=DEV), torch.full((n, 1), far, device=DEV), ds.to(DEV),
                     cos_anneal_ratio=1.0, perturb_overwrite=0, flip_saturation=0.0)
    img_h = o["edge"].reshape(HW, HW).cpu()
    img_r = t(g["held_out_after"])
    gt = gt.reshape(HW, HW)
    psnr = lambda a: 10.0 * np.log10(1.0 / max(float(((a - gt) ** 2).mean()), 1e-12))
    p_h, p_r = psnr(img_h), psnr(img_r)
    rms = float(((img_h - img_r) ** 2).mean().sqrt())
    print(f"held-out view: PSNR vs ground truth  HIP-trained {p_h:.2f} dB, reference-trained {p_r:.2f} dB (recorded {float(g['psnr'][1]):.2f}; before training "
          f"{float(g['psnr'][0]):.2f}); RMS difference of the two models' edge maps {rms:.4f}")
    assert p_r == pytest.approx(float(g["psnr"][1]), abs=1e-6)
    assert abs(p_h - p_r) <= 0.5 and p_h >= float(g["psnr"][0]) + 10.0
    assert rms <= 0.04
    assert float(r.deviation_network.variance) == pytest.approx(float(g["variance"]), rel=5e-2)
    assert float(r.beta_network.beta) == pytest.approx(float(g["beta"]), rel=5e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [("f16x3", 5e-5), ("f16x3e", 2e-5), ("bf16x3", 1e-4), ("f16", 5e-3)])
@pytest.mark.parametrize("skip_in", [(4,), (7,)])
def test_reverse_sweep_with_and_without_the_fused_output_layer_vs_forward_mode_tangents(prec, tol, skip_in):
    """udf_mlp_rev32_kernel applies the output layer and forms the reverse sweep's first delta_z inside the LAST hidden layer's epilogue (LASTH,
    udf_mlp_rev32.inc) when that layer is an ordinary one - skip_in = (4,), the reference's network (udf_model.py:24-45).  With the skip connection
    feeding the last hidden layer (skip_in = (7,)) the layer reads the PE block, LASTH does not apply and the kernel takes the path of rounds 3-6a
    (output layer as a K-split GEMM, sigma' of that layer through the stash).  Both against the forward-mode tangent kernel (itself pinned by g2 and
    the oracle) at launch sizes with one and with several tiles per workgroup, bit-stable run to run."""
    kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=skip_in, multires=10, bias=0.5)
    state = synthetic.make_udf_state(seed=77, pert=0.02, **kw)
    net = emap_amd.UDFNetwork(scale=1.0, precision=prec, **kw)
    net.load_state_dict(state)
    net = net.to(DEV)
    gen = torch.Generator().manual_seed(9)
    x = (torch.rand(65536 + 4099, 3, generator=gen) * 2 - 1).to(DEV)
    L = _lib.lib()
    for P in (16384, 65536 + 4099):
        xp = x[:P].contiguous()
        with torch.no_grad():
            old = L.emap_set_grad_mode(1)
            try:
                u, g = net.hip_udf(xp, with_grad=True)
                u2, g2 = net.hip_udf(xp, with_grad=True)
                L.emap_set_grad_mode(0)
                uf, gf = net.hip_udf(xp[:16384].contiguous(), with_grad=True)
            finally:
                L.emap_set_grad_mode(old)
        assert torch.equal(u, u2) and torch.equal(g, g2)
        du = float((u[:16384] - uf).abs().max() / uf.abs().max())
        dg = float((g[:16384] - gf).abs().max() / gf.abs().max())
        assert du <= tol and dg <= tol, (skip_in, prec, P, du, dg)
