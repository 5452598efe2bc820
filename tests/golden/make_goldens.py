#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_goldens.py

It imports the reference's ``src.models.*`` (pure torch), loads
weights produced by this repo's seeded generator (``emap_amd.synthetic``)
through ``load_state_dict`` and records inputs + outputs as small .npz files.
Only data is written; no reference source or bytecode is copied.

Integer intermediates that the reference computes but does not return
(``searchsorted`` indices in sample_pdf, the ``sort`` permutation in
cat_z_vals) are captured by temporarily wrapping ``torch.searchsorted`` /
``torch.sort`` while the reference function runs.
"""
import os
import sys
import contextlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("EMAP_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from emap_amd import synthetic  # noqa: E402
from src.models.embedder import get_embedder  # noqa: E402  (reference)
from src.models.udf_model import UDFNetwork, SingleVarianceNetwork, BetaNetwork  # noqa: E402
from src.models.udf_renderer_blending import UDFRendererBlending, sample_pdf  # noqa: E402
from src.models.loss import EdgeLoss  # noqa: E402

torch.set_default_dtype(torch.float32)
torch.set_num_threads(8)


@contextlib.contextmanager
def capture(name):
    """Record every result of torch.<name> while active."""
    orig = getattr(torch, name)
    rec = []

    def wrapped(*a, **k):
        out = orig(*a, **k)
        rec.append(out)
        return out

    setattr(torch, name, wrapped)
    try:
        yield rec
    finally:
        setattr(torch, name, orig)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path)/1024:.1f} KiB  keys={len(out)}")


NETS = {
    # name: (ctor kwargs, seed, pert)
    "d8w256L10": (dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5), 42, 0.02),
    "d8w256L6": (dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=6, bias=0.5), 43, 0.02),
    "d4w128L10": (dict(d_in=3, d_out=1, d_hidden=128, n_layers=4, skip_in=(4,), multires=10, bias=0.5), 44, 0.02),
    "d8w256L10_init": (dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5), 45, 0.0),
}


def build_net(name, scale=1.0):
    kw, seed, pert = NETS[name]
    net = UDFNetwork(scale=scale, geometric_init=True, weight_norm=True, udf_type="abs", **kw)
    state = synthetic.make_udf_state(seed=seed, pert=pert, **kw)
    net.load_state_dict(state)
    return net, state


def state_checksum(state):
    return np.array([float(v.double().abs().sum()) for v in state.values()])


def g1_pe():
    rng = np.random.Generator(np.random.PCG64(100))
    x = torch.tensor(rng.uniform(-1.2, 1.2, size=(64, 3)), dtype=torch.float32)
    out = {"x": x}
    for L in (10, 6):
        fn, dim = get_embedder(L, input_dims=3)
        out[f"pe_L{L}"] = fn(x)
        assert dim == 3 + 6 * L
    save("g1_pe", **out)


def g2_mlp():
    rng = np.random.Generator(np.random.PCG64(101))
    x = torch.tensor(rng.uniform(-1.2, 1.2, size=(256, 3)), dtype=torch.float32)
    out = {"x": x}
    for name in NETS:
        net, state = build_net(name)
        fo, pe = net(x)
        udf, feat, _ = net.udf(x)
        grad = net.gradient(x.clone()).detach()
        out[f"{name}.out"] = fo.detach()
        out[f"{name}.pe"] = pe.detach()
        out[f"{name}.udf"] = udf.detach()
        out[f"{name}.grad"] = grad
        out[f"{name}.wsum"] = state_checksum(state)
    # scale != 1 variant (udf_model.py:91,108)
    net, state = build_net("d8w256L10", scale=1.5)
    out["scale1p5.udf"] = net.udf(x)[0].detach()
    out["scale1p5.grad"] = net.gradient(x.clone()).detach()
    save("g2_mlp", **out)


def g3_sample_pdf():
    rng = np.random.Generator(np.random.PCG64(102))
    N, n = 32, 64
    bins = np.sort(rng.uniform(0.05, 6.0, size=(N, n)), axis=-1)
    w = rng.uniform(0, 1, size=(N, n - 1)) ** 4
    w[:4] = 0.0  # all-zero weights rows -> uniform pdf (the +1e-5 path)
    w[4:8, 10:] = 0.0  # weight concentrated at the front -> long flat cdf tail (denom<1e-5 branch)
    # rows whose weights are small dyadic rationals: every partial sum is exact in fp32, so the
    # searchsorted indices are independent of summation order
    w[8:16] = rng.integers(0, 8, size=(8, n - 1)) / 64.0
    bins_t = torch.tensor(bins, dtype=torch.float32)
    w_t = torch.tensor(w, dtype=torch.float32)
    out = {"bins": bins_t, "weights": w_t}
    for m in (10, 16):
        with capture("searchsorted") as rec:
            s = sample_pdf(bins_t, w_t, m, det=True)
        out[f"samples_m{m}"] = s
        out[f"inds_m{m}"] = rec[0]
    save("g3_sample_pdf", **out)


def g14_mlp_multires0():
    """multires = 0 (udf_model.py:26-29: no embedding, dims[0] = 3; the geometric initialisation takes its plain branch): value, "PE" (= the scaled
    input) and gradient of the reference network, d8 w256 and d4 w128"""
    rng = np.random.Generator(np.random.PCG64(114))
    x = torch.tensor(rng.uniform(-1.2, 1.2, size=(256, 3)), dtype=torch.float32)
    out = {"x": x}
    for name, kw, seed in (("d8w256L0", dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=0, bias=0.5), 46),
                           ("d4w128L0", dict(d_in=3, d_out=1, d_hidden=128, n_layers=4, skip_in=(4,), multires=0, bias=0.5), 47)):
        net = UDFNetwork(scale=1.0, geometric_init=True, weight_norm=True, udf_type="abs", **kw)
        state = synthetic.make_udf_state(seed=seed, pert=0.02, **kw)
        net.load_state_dict(state)
        fo, pe = net(x)
        out[f"{name}.out"] = fo.detach()
        out[f"{name}.pe"] = pe.detach()
        out[f"{name}.grad"] = net.gradient(x.clone()).detach()
        out[f"{name}.wsum"] = state_checksum(state)
    save("g14_mlp_multires0", **out)


def g13_sample_pdf_random():
    """sample_pdf(det=False) (udf_renderer_blending.py:84-85): u = torch.rand on the CPU generator; seed, draws and samples recorded"""
    rng = np.random.Generator(np.random.PCG64(113))
    N, n = 24, 48
    bins_t = torch.tensor(np.sort(rng.uniform(0.05, 6.0, size=(N, n)), axis=-1), dtype=torch.float32)
    w = rng.uniform(0, 1, size=(N, n - 1)) ** 4
    w[:3] = 0.0
    w[3:6, 8:] = 0.0
    w_t = torch.tensor(w, dtype=torch.float32)
    out = {"bins": bins_t, "weights": w_t, "seed": np.int64(20240613)}
    for m in (7, 32):
        torch.manual_seed(20240613 + m)
        with capture("searchsorted") as rec:
            s = sample_pdf(bins_t, w_t, m, det=False)
        torch.manual_seed(20240613 + m)
        out[f"u_m{m}"] = torch.rand([N, m])
        out[f"samples_m{m}"] = s
        out[f"inds_m{m}"] = rec[0]
    save("g13_sample_pdf_random", **out)


def make_renderer(net, n_samples, n_importance, steps, variance=0.3, beta=0.5, gamma=0.3, perturb=1.0):
    dev = SingleVarianceNetwork(variance)
    bet = BetaNetwork(init_var_beta=beta, init_var_gamma=gamma, init_var_zeta=0.3, beta_min=0.00005,
                      requires_grad_beta=True, requires_grad_gamma=True, requires_grad_zeta=False)
    r = UDFRendererBlending(None, net, dev, bet, n_samples=n_samples, n_importance=n_importance,
                            n_outside=0, up_sample_steps=steps, perturb=perturb,
                            sdf2alpha_type="numerical", upsampling_type="classical",
                            use_unbias_render=True, device="cpu")
    return r, dev, bet


def g4_upsample_step():
    net, _ = build_net("d8w256L10")
    r, _, _ = make_renderer(net, 64, 64, 4)
    N = 32
    rays_o, rays_d, near, far, _ = synthetic.make_rays(N, seed=3)
    z = near + (far - near) * torch.linspace(0, 1, 64)[None, :]
    sample_dist = ((far - near) / 64).mean().item()
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[..., :, None]
    with torch.no_grad():
        udf = net.udf(pts.reshape(-1, 3))[0].reshape(N, 64)
        out = {"rays_o": rays_o, "rays_d": rays_d, "z_vals": z, "udf": udf, "sample_dist": sample_dist}
        for i in range(2):
            inv_s, beta, gamma = 64 * 2 ** i, 64 * 2 ** (i + 1), float(np.clip(20 * 2 ** (4 - i), 20, 320))
            with capture("searchsorted") as rec:
                z_new = r.up_sample_unbias(rays_o, rays_d, z, udf, sample_dist, 16, inv_s, beta, gamma)
            with capture("sort") as srec:
                z2, udf2 = r.cat_z_vals(rays_o, rays_d, z, z_new, udf, last=False)
            out[f"step{i}.params"] = np.array([inv_s, beta, gamma], dtype=np.float64)
            out[f"step{i}.z_new"] = z_new
            out[f"step{i}.inds"] = rec[0]
            out[f"step{i}.z_out"] = z2
            out[f"step{i}.udf_out"] = udf2
            out[f"step{i}.sort_index"] = srec[0][1]
            z, udf = z2, udf2
    save("g4_upsample_step", **out)


RENDER_KEYS = ["udf", "edge", "weight_sum", "weight_sum_fg_bg", "depth", "variance", "beta", "gamma",
               "normals", "gradients", "gradients_flip", "weights", "gradient_error",
               "gradient_error_near_surface", "inside_sphere", "gradient_mag", "mid_z_vals", "dists"]


def g5_render():
    cases = {
        "c64_50_5": ("d8w256L10", 64, 50, 5),
        "c64_64_4": ("d8w256L10", 64, 64, 4),
        "c32_32_4_small": ("d4w128L10", 32, 32, 4),
        "c64_64_4_L6": ("d8w256L6", 64, 64, 4),
        # n_importance = 0: render() skips importance_sample (udf_renderer_blending.py:740) - SURVEY par. 8d's second reading of
        # config C1 ("S_c = 64, S_f = 0"), on the C1 network and on the d8 w256 one
        "c64_0": ("d8w256L10", 64, 0, 4),
        "c64_0_small": ("d4w128L10", 64, 0, 4),
    }
    N = 32
    for cname, (netname, ns, ni, steps) in cases.items():
        net, _ = build_net(netname)
        far_v = 2.5 if netname.endswith("L6") else 6.0
        rays_o, rays_d, near, far, depth_scale = synthetic.make_rays(N, seed=5, far=far_v)
        r, dev, bet = make_renderer(net, ns, ni, steps)
        zs = []
        orig = r.cat_z_vals

        def rec_cat(*a, **k):
            z, u = orig(*a, **k)
            zs.append(z.detach().clone())
            return z, u

        r.cat_z_vals = rec_cat
        out = r.render(rays_o, rays_d, near, far, depth_scale, cos_anneal_ratio=1.0, perturb_overwrite=0,
                       flip_saturation=0.9)
        d = {"rays_o": rays_o, "rays_d": rays_d, "near": near, "far": far, "depth_scale": depth_scale,
             "cfg": np.array([ns, ni, steps]), "cos_anneal_ratio": 1.0, "flip_saturation": 0.9}
        for k in RENDER_KEYS:
            d["out." + k] = out[k]
        for i, z in enumerate(zs):
            d[f"z_after_step{i}"] = z
        # a second setting of the two schedules + a white background
        out2 = r.render(rays_o, rays_d, near, far, depth_scale, cos_anneal_ratio=0.3, perturb_overwrite=0,
                        flip_saturation=0.0, background_rgb=torch.ones([1, 1]))
        for k in ["edge", "depth", "weights", "normals", "gradient_error"]:
            d["out2." + k] = out2[k]
        save("g5_render_" + cname, **d)


def g6_training():
    """loss and dL/dtheta the way runner_udf.py:96-168 assembles them (mask of ones, MSE*edge_weight
    + igr_ns_weight*ge_ns + igr_weight*ge, backward)."""
    N = 16
    cases = [("d4w128L10", 32, 32, 4, 0.3, 0.0), ("d4w128L10", 32, 32, 4, 1.0, 0.9),
             ("d4w128L10", 64, 50, 5, 1.0, 0.0), ("d8w256L10", 64, 64, 4, 0.3, 0.9)]
    for ci, (netname, ns, ni, steps, car, fs) in enumerate(cases):
        net, _ = build_net(netname)
        rays_o, rays_d, near, far, depth_scale = synthetic.make_rays(N, seed=20 + ci)
        true_edge = synthetic.make_true_edge(N, seed=30 + ci)
        r, dev, bet = make_renderer(net, ns, ni, steps)
        loss_fn = EdgeLoss("mse")
        zs = []
        orig = r.cat_z_vals

        def rec_cat(*a, _orig=orig, _zs=zs, **k):
            z, u = _orig(*a, **k)
            _zs.append(z.detach().clone())
            return z, u

        r.cat_z_vals = rec_cat
        out = r.render(rays_o, rays_d, near, far, depth_scale, cos_anneal_ratio=car, perturb_overwrite=0,
                       flip_saturation=fs)
        edge_weight, igr_weight, igr_ns_weight = 1.0, 0.1, 0.05
        edge_loss = loss_fn(out["edge"], true_edge) * edge_weight
        loss = edge_loss + out["gradient_error_near_surface"] * igr_ns_weight + out["gradient_error"] * igr_weight
        for p in list(net.parameters()) + list(dev.parameters()) + list(bet.parameters()):
            p.grad = None
        loss.backward()
        d = {"rays_o": rays_o, "rays_d": rays_d, "near": near, "far": far, "depth_scale": depth_scale,
             "true_edge": true_edge, "cfg": np.array([ns, ni, steps]), "cos_anneal_ratio": car,
             "flip_saturation": fs, "weights3": np.array([edge_weight, igr_weight, igr_ns_weight]),
             "loss": loss.detach(), "edge_loss": edge_loss.detach(), "edge": out["edge"].detach(),
             "gradient_error": out["gradient_error"].detach(),
             "gradient_error_near_surface": out["gradient_error_near_surface"].detach(),
             "netname": np.array(netname),
             # the samples the reference differentiated at (importance_sample is @no_grad, :802) and what it saw there
             "z_vals": zs[-1], "udf": out["udf"].detach(), "gradients": out["gradients"].detach()}
        for k, p in net.named_parameters():
            d["grad." + k] = p.grad if p.grad is not None else torch.zeros_like(p)
        d["grad.variance"] = dev.variance.grad if dev.variance.grad is not None else torch.zeros(1)
        d["grad.beta"] = bet.beta.grad if bet.beta.grad is not None else torch.zeros(1)
        d["grad.gamma"] = bet.gamma.grad if bet.gamma.grad is not None else torch.zeros(1)
        save(f"g6_training_{ci}", **d)


def g12_training_steps():
    """A short TRAINING RUN recorded from the reference's own classes: 48 optimizer steps of the loop of runner_udf.py:63-168 -
    render() under autograd, EdgeLoss, the loss assembly, loss.backward(), torch.optim.Adam with the runner's parameter groups
    (runner_base.py:106-117) and its schedules (runner_base.py:128-180: warm-up + cosine learning rates per group, cos_anneal_ratio,
    flip_saturation) with the iteration axis compressed (warm_up_end 8, end_iter 48, anneal_end 16).  The runner module itself
    cannot be imported here (pyhocon, cv2, tensorboard), so its loop body is restated around the reference's model classes; the
    rays of step i come from this repo's seeded generator.  Pins row a15's optimizer tail to the reference (VERDICT r3 item 5 iv)."""
    netname, ns, ni, steps_up, N, n_steps = "d4w128L10", 32, 32, 4, 64, 48
    net, _ = build_net(netname)
    r, dev, bet = make_renderer(net, ns, ni, steps_up)
    lr, lr_geo, alpha, warm_up_end, end_iter, anneal_end, fix_geo_end = 5e-4, 1e-4, 0.05, 8, n_steps, 16, 0
    edge_weight, igr_weight, igr_ns_weight = 1.0, 0.1, 0.0
    opt = torch.optim.Adam([{"params": list(net.parameters()), "lr": lr_geo},
                            {"params": list(dev.parameters()) + list(bet.parameters())},
                            {"params": []}], lr=lr)
    loss_fn = EdgeLoss("mse")
    losses, edge_losses, ges, variances, betas, lrs = [], [], [], [], [], []
    for it in range(n_steps):
        # update_learning_rate(start_g_id=1) + update_learning_rate_geo()  (runner_udf.py:64-68, same_lr = False)
        if it < warm_up_end:
            f = it / warm_up_end
        else:
            f = (np.cos(np.pi * (it - warm_up_end) / (end_iter - warm_up_end)) + 1.0) * 0.5 * (1 - alpha) + alpha
        for g in opt.param_groups[1:]:
            g["lr"] = lr * f
        if it < fix_geo_end:
            fg = 0.0
        elif it < warm_up_end * 2:
            fg = it / (warm_up_end * 2)
        elif it < end_iter * 0.5:
            fg = 1.0
        else:
            fg = (np.cos(np.pi * (it - end_iter * 0.5) / (end_iter - end_iter * 0.5)) + 1.0) * 0.5 * (1 - alpha) + alpha
        for g in opt.param_groups[:1]:
            g["lr"] = lr_geo * fg
        car = float(np.min([1.0, it / anneal_end]))
        fs = 0.0                                        # get_flip_saturation(): 0 before iteration 10000
        rays_o, rays_d, near, far, depth_scale = synthetic.make_rays(N, seed=1000 + it)
        true_edge = synthetic.make_true_edge(N, seed=2000 + it)
        out = r.render(rays_o, rays_d, near, far, depth_scale, cos_anneal_ratio=car, perturb_overwrite=0, flip_saturation=fs)
        edge_loss = loss_fn(out["edge"], true_edge) * edge_weight
        loss = edge_loss + out["gradient_error_near_surface"] * igr_ns_weight + out["gradient_error"] * igr_weight
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss)); edge_losses.append(float(edge_loss)); ges.append(float(out["gradient_error"]))
        variances.append(float(dev.variance)); betas.append(float(bet.beta)); lrs.append([lr_geo * fg, lr * f])
    d = {"netname": np.array(netname), "cfg": np.array([ns, ni, steps_up]), "n_rays": np.array(N), "n_steps": np.array(n_steps),
         "ray_seed0": np.array(1000), "edge_seed0": np.array(2000), "weights3": np.array([edge_weight, igr_weight, igr_ns_weight]),
         "schedule": np.array([lr, lr_geo, alpha, warm_up_end, end_iter, anneal_end, fix_geo_end]),
         "loss": np.array(losses), "edge_loss": np.array(edge_losses), "gradient_error": np.array(ges),
         "variance": np.array(variances), "beta": np.array(betas), "lrs": np.array(lrs)}
    for k, p in net.named_parameters():
        d["final." + k] = p.detach()
    d["final.gamma"] = bet.gamma.detach()
    save("g12_training_steps", **d)


def g15_convergence():
    """BASELINE config C5 in miniature, recorded from the REFERENCE's own classes: 1000 optimizer steps of the loop of
    runner_udf.py:63-168 (render under autograd, EdgeLoss, loss assembly, loss.backward(), torch.optim.Adam with the runner's parameter
    groups and schedules, iteration axis compressed) on a MULTI-VIEW CONSISTENT target - the projections of one 3D wire frame
    (emap_amd.synthetic.make_wireframe_scene) - from the reference's own seeded geometric initialisation, followed by the reference's
    render of a HELD-OUT view.  The rays of a batch are the reference's rays (synthetic.scene_rays, checked here against
    Dataset.gen_random_rays_patches_at on the same pixels); the pixel draws come from this repo's seeded generator
    (synthetic.convergence_batch).  What the GPU test compares: the loss curve, the held-out view's edge map and its PSNR of a model
    trained by the HIP path on the same batches and schedules."""
    import types
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    from src.dataset.dataset import Dataset  # reference
    n_views, HW, held_out = 8, 40, 0
    meta, edges = synthetic.make_wireframe_scene(n_images=n_views, H=HW, W=HW)
    K = torch.stack([torch.tensor(f["intrinsics"], dtype=torch.float32) for f in meta["frames"]])
    P = torch.stack([torch.tensor(f["camtoworld"], dtype=torch.float32) for f in meta["frames"]])
    e = torch.from_numpy(edges).float()
    ns_ = types.SimpleNamespace(edges=e, masks=torch.ones(n_views, HW, HW, 3), intrinsics_all=K, intrinsics_all_inv=torch.inverse(K), pose_all=P,
                                H=HW, W=HW, image_pixels=HW * HW, device=torch.device("cpu"))
    # the ray construction used below IS the reference's: same numbers as Dataset.gen_random_rays_patches_at on the same pixels
    img, px, py = synthetic.convergence_batch(meta, edges, 64, seed=1, held_out=held_out)
    q = [torch.from_numpy(px), torch.from_numpy(py)]
    orig_randint = torch.randint
    torch.randint = lambda low=0, high=None, size=None, **k: q.pop(0)
    try:
        smp = Dataset.gen_random_rays_patches_at(ns_, img, 64, importance_sample=False)
    finally:
        torch.randint = orig_randint
    ro, rv, ds, ed = synthetic.scene_rays(meta, edges, img, px, py)
    assert torch.equal(ro, smp["rays"]["rays_o"]) and torch.equal(rv, smp["rays"]["rays_v"]) and torch.equal(ed, smp["rays"]["edge"])
    assert torch.equal(ds, smp["depth_scale"])

    kw = NETS["d8w256L10"][0]          # the architecture of every EMAP conf (d4 w128 from this initialisation spends the run on its eikonal term)
    ns, ni, steps_up, N, n_steps = 32, 32, 4, 256, 1000
    torch.manual_seed(4321)
    net = UDFNetwork(scale=1.0, geometric_init=True, weight_norm=True, udf_type="abs", **kw)       # the reference's own initialisation
    init = {k: v.detach().clone() for k, v in net.state_dict().items()}
    r, dev, bet = make_renderer(net, ns, ni, steps_up)
    near, far = float(meta["scene_box"]["near"]), float(meta["scene_box"]["far"])
    lr, lr_geo, alpha, warm_up_end, end_iter, anneal_end = 5e-4, 1e-4, 0.05, 20, n_steps, 200
    edge_weight, igr_weight, igr_ns_weight = 1.0, 0.1, 0.0
    opt = torch.optim.Adam([{"params": list(net.parameters()), "lr": lr_geo},
                            {"params": list(dev.parameters()) + list(bet.parameters())}, {"params": []}], lr=lr)
    loss_fn = EdgeLoss("mse")
    losses, edge_losses, lrs = [], [], []

    def render_view(idx):
        ys, xs = np.mgrid[0:HW, 0:HW]
        ro_, rv_, ds_, ed_ = synthetic.scene_rays(meta, edges, idx, xs.reshape(-1), ys.reshape(-1))
        outs = []
        for h in range(0, HW * HW, 400):
            nn = ro_[h:h + 400].shape[0]      # (n,1) near / far: with python floats and perturb_overwrite = 0 the reference's z_vals stay (1, n_samples)
            o = r.render(ro_[h:h + 400], rv_[h:h + 400], torch.full((nn, 1), near), torch.full((nn, 1), far), ds_[h:h + 400], cos_anneal_ratio=1.0,
                         perturb_overwrite=0, flip_saturation=0.0)
            outs.append(o["edge"].detach())
        img_ = torch.cat(outs)
        mse = float(((img_ - ed_) ** 2).mean())
        return img_.reshape(HW, HW), 10.0 * np.log10(1.0 / max(mse, 1e-12))

    before, psnr_before = render_view(held_out)
    import time
    t0 = time.time()
    for it in range(n_steps):
        f = it / warm_up_end if it < warm_up_end else (np.cos(np.pi * (it - warm_up_end) / (end_iter - warm_up_end)) + 1.0) * 0.5 * (1 - alpha) + alpha
        if it < warm_up_end * 2:
            fg = it / (warm_up_end * 2)
        elif it < end_iter * 0.5:
            fg = 1.0
        else:
            fg = (np.cos(np.pi * (it - end_iter * 0.5) / (end_iter - end_iter * 0.5)) + 1.0) * 0.5 * (1 - alpha) + alpha
        for g in opt.param_groups[1:]:
            g["lr"] = lr * f
        opt.param_groups[0]["lr"] = lr_geo * fg
        car = float(np.min([1.0, it / anneal_end]))
        img, px, py = synthetic.convergence_batch(meta, edges, N, seed=5000 + it, held_out=held_out)
        ro, rv, ds, true_edge = synthetic.scene_rays(meta, edges, img, px, py)
        out = r.render(ro, rv, torch.full((N, 1), near), torch.full((N, 1), far), ds, cos_anneal_ratio=car, perturb_overwrite=0, flip_saturation=0.0)
        edge_loss = loss_fn(out["edge"], true_edge) * edge_weight
        loss = edge_loss + out["gradient_error_near_surface"] * igr_ns_weight + out["gradient_error"] * igr_weight
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss)); edge_losses.append(float(edge_loss)); lrs.append([lr_geo * fg, lr * f])
        if it % 100 == 99:
            print(f"  g15 step {it + 1}: loss {np.mean(losses[-100:]):.4f} ({time.time() - t0:.0f} s)", flush=True)
    after, psnr_after = render_view(held_out)
    train_img, psnr_train = render_view(3)
    segs = torch.tensor(synthetic.wireframe_segments(), dtype=torch.float32)
    tt = torch.linspace(0.05, 0.95, 32).view(1, -1, 1)
    on = (segs[:, :1] * (1 - tt) + segs[:, 1:] * tt).reshape(-1, 3)
    udf_on = float(net.udf(on)[0].mean()) if isinstance(net.udf(on), tuple) else float(net.udf(on).mean())
    d = {"netname": np.array("d8w256L10"), "cfg": np.array([ns, ni, steps_up]), "n_rays": np.array(N), "n_steps": np.array(n_steps),
         "scene": np.array([n_views, HW, held_out]), "batch_seed0": np.array(5000), "init_seed": np.array(4321),
         "weights3": np.array([edge_weight, igr_weight, igr_ns_weight]),
         "schedule": np.array([lr, lr_geo, alpha, warm_up_end, end_iter, anneal_end]), "lrs": np.array(lrs),
         "loss": np.array(losses), "edge_loss": np.array(edge_losses),
         "held_out_before": before, "held_out_after": after, "train_view_after": train_img,
         "psnr": np.array([psnr_before, psnr_after, psnr_train]), "udf_on_wireframe": np.array(udf_on),
         "variance": np.array(float(dev.variance)), "beta": np.array(float(bet.beta)), "gamma": np.array(float(bet.gamma))}
    for k, v in init.items():
        d["init." + k + ".abs_sum"] = v.double().abs().sum()
    print(f"  g15: held-out PSNR {psnr_before:.2f} -> {psnr_after:.2f} dB, train view {psnr_train:.2f} dB, udf on the wire frame {udf_on:.4f}")
    save("g15_convergence", **d)


def g7_perturb():
    """The perturb path (render() :716-720): ONE (N,1) CPU torch.rand draw shifts every ray."""
    net, _ = build_net("d4w128L10")
    N = 32
    rays_o, rays_d, near, far, depth_scale = synthetic.make_rays(N, seed=6)
    r, dev, bet = make_renderer(net, 32, 32, 4)
    torch.manual_seed(42)
    t_rand = torch.rand([N, 1]) - 0.5
    torch.manual_seed(42)
    out = r.render(rays_o, rays_d, near, far, depth_scale, cos_anneal_ratio=1.0, flip_saturation=0.9)
    # float near/far (the way the runner calls it, runner_udf.py:90,96-108)
    torch.manual_seed(42)
    out_f = r.render(rays_o, rays_d, 0.05, 6.0, depth_scale, cos_anneal_ratio=1.0, flip_saturation=0.9)
    save("g7_perturb", rays_o=rays_o, rays_d=rays_d, near=near, far=far, depth_scale=depth_scale,
         t_rand=t_rand, edge=out["edge"], depth=out["depth"], mid_z_vals=out["mid_z_vals"],
         weights=out["weights"], edge_float_nearfar=out_f["edge"], mid_z_float_nearfar=out_f["mid_z_vals"])


def g8_scalars():
    dev = SingleVarianceNetwork(0.3)
    bet = BetaNetwork(0.5, 0.3, 0.3, 0.00005, True, True, False)
    save("g8_scalars", inv_s=dev(torch.zeros(5, 3)), beta=bet.get_beta(), gamma=bet.get_gamma(),
         zeta=bet.get_zeta())


def g9_seeded_init():
    """Reference constructor under torch.manual_seed: the drop-in classes must consume the RNG identically."""
    out = {}
    for name, kw in (("d8w256L10", NETS["d8w256L10"][0]), ("d4w128L10", NETS["d4w128L10"][0])):
        torch.manual_seed(1234)
        net = UDFNetwork(scale=1.0, geometric_init=True, weight_norm=True, udf_type="abs", **kw)
        for k, v in net.state_dict().items():
            out[f"{name}.{k}.abs_sum"] = v.double().abs().sum()
            out[f"{name}.{k}.head"] = v.reshape(-1)[:4].clone()
            out[f"{name}.{k}.shape"] = np.array(v.shape)
    save("g9_seeded_init", **out)


def g10_extraction():
    """Dense-grid extraction queries (SURVEY par. 8 f2): the reference's get_udf_normals_grid on a 12^3 grid and
    get_udf_normals_slow on 300 points, with the reference UDFNetwork on the CPU.  The jitter draws are recorded (they are
    inputs for the parity tests); get_udf_normals_slow hard-codes ``.cuda()``, which is patched to the identity here."""
    from src.edge_extraction.extract_pointcloud import get_udf_normals_grid, get_udf_normals_slow  # reference
    net, state = build_net("d8w256L10")
    N = 12
    torch.manual_seed(99)
    with capture("randn") as noise:
        df0, _, _, _, _ = get_udf_normals_grid(net.udf, net.gradient, N, -1.0, False, device="cpu")
        thr = float(df0.reshape(-1).quantile(0.35))
        df, ld, vecs, samples, vs = get_udf_normals_grid(net.udf, net.gradient, N, thr, True, sampling_N=50, sampling_delta=0.005,
                                                         max_batch=256, device="cpu")
    out = dict(N=np.array(N), thr=np.array(thr, dtype=np.float64), df=df, ld=ld, vecs=vecs, voxel_size=vs,
               grid_noise=torch.cat(list(noise)), state_checksum=state_checksum(state))
    gen = torch.Generator().manual_seed(5)
    xyz = (torch.rand(300, 3, generator=gen) * 1.6 - 0.8)
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        with capture("randn") as noise2:
            dfs, normals, lds, _ = get_udf_normals_slow(net.udf, net.gradient, None, xyz, True, sampling_N=50, sampling_delta=0.005,
                                                        max_batch=128, device="cpu")
    finally:
        torch.Tensor.cuda = orig_cuda
    out.update(xyz=xyz, slow_df=dfs, slow_normals=normals, slow_ld=lds, slow_noise=torch.cat(list(noise2)))

    # The call pattern of the real caller: Runner_UDF.extract_edge hands get_pointcloud_from_udf a CLOSURE that normalises the
    # gradient (runner_udf.py:520-527) - the runner module itself cannot be imported here (pyhocon, cv2, ...), so the
    # closure is restated verbatim around the reference's UDFNetwork.  Same seeds -> same jitter draws as above.
    def func_grad(xyz_):
        gradients = net.gradient(xyz_)
        gradients_mag = torch.linalg.norm(gradients, ord=2, dim=-1, keepdim=True)
        gradients_norm = gradients / (gradients_mag + 1e-5)
        return gradients_norm

    torch.manual_seed(99)
    get_udf_normals_grid(net.udf, func_grad, N, -1.0, False, device="cpu")
    _, c_ld, c_vecs, _, _ = get_udf_normals_grid(net.udf, func_grad, N, thr, True, sampling_N=50, sampling_delta=0.005,
                                                 max_batch=256, device="cpu")
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        torch.manual_seed(99)
        with capture("randn") as noise3:
            _, c_normals, c_lds, _ = get_udf_normals_slow(net.udf, func_grad, None, xyz, True, sampling_N=50, sampling_delta=0.005,
                                                          max_batch=128, device="cpu")
    finally:
        torch.Tensor.cuda = orig_cuda
    out.update(closure_ld=c_ld, closure_vecs=c_vecs, closure_slow_normals=c_normals, closure_slow_ld=c_lds,
               closure_slow_noise=torch.cat(list(noise3)))
    save("g10_extraction", **out)


def g11_rays():
    """Ray generation of the training loop (SURVEY par. 8 f3): the reference's own Dataset.gen_random_rays_patches_at
    (src/dataset/dataset.py:222-307), called UNBOUND on a plain namespace that carries the attributes the method reads.  The dataset
    module imports cv2 at the top (absent here); none of the executed lines uses it, so an empty stub module is registered for the
    import only.  The pixel draws (torch.randint, random.choices) are replaced by fixed pixels, which are recorded as inputs; the
    edge-weighted draw is pinned separately by the class histogram of 2^18 real random.choices draws (python's own generator,
    seeded) over the probabilities the method builds (:236-242)."""
    import random
    import types
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    from src.dataset.dataset import Dataset  # reference
    meta, edges = synthetic.make_scene(n_images=3, H=40, W=50, seed=5)
    K = torch.stack([torch.tensor(f["intrinsics"], dtype=torch.float32) for f in meta["frames"]])
    P = torch.stack([torch.tensor(f["camtoworld"], dtype=torch.float32) for f in meta["frames"]])
    H, W = int(meta["height"]), int(meta["width"])
    e = torch.from_numpy(edges).float()
    if e.dim() == 3:
        e = e[..., None]
    ns = types.SimpleNamespace(edges=e, masks=torch.ones(3, H, W, 3), intrinsics_all=K, intrinsics_all_inv=torch.inverse(K), pose_all=P,
                               H=H, W=W, image_pixels=H * W, device=torch.device("cpu"))
    gen = torch.Generator().manual_seed(17)
    out = dict(H=np.array(H), W=np.array(W), edges=e[..., 0], intrinsics=K, pose=P)
    orig_randint, orig_choices = torch.randint, random.choices

    def run(tag, img_idx, n, importance):
        px = torch.randint(0, W, [n], generator=gen)
        py = torch.randint(0, H, [n], generator=gen)
        half = n // 2
        q = [px[:half] if importance else px, py[:half] if importance else py]
        torch.randint = lambda low=0, high=None, size=None, **k: q.pop(0)
        # p_valid is the row-major pixel list (all masks >= 0): index = y * W + x
        random.choices = lambda population, weights=None, k=None: [int(py[half + i]) * W + int(px[half + i]) for i in range(n - half)]
        try:
            smp = Dataset.gen_random_rays_patches_at(ns, img_idx, n, importance_sample=importance)
        finally:
            torch.randint, random.choices = orig_randint, orig_choices
        out.update({f"{tag}.img_idx": np.array(img_idx), f"{tag}.px": px, f"{tag}.py": py, f"{tag}.rays_o": smp["rays"]["rays_o"],
                    f"{tag}.rays_v": smp["rays"]["rays_v"], f"{tag}.edge": smp["rays"]["edge"], f"{tag}.uv": smp["rays_ndc_uv"],
                    f"{tag}.p": smp["rays_norm_XYZ_cam"], f"{tag}.depth_scale": smp["depth_scale"], f"{tag}.pose": smp["pose"],
                    f"{tag}.intrinsics_out": smp["intrinsics"]})
        out[f"{tag}.keys"] = np.array(sorted(smp.keys()) + ["rays." + k for k in sorted(smp["rays"].keys())])

    run("uniform", 1, 96, False)
    run("importance", 2, 64, True)
    # the edge-weighted half of the draw: class histogram of real random.choices draws over the method's probabilities
    img = e[2, ..., 0].numpy()
    dens = np.mean(img)
    prob = np.ones_like(img) * dens
    prob[img > 0.1] = 1.0 - dens
    random.seed(123)
    idx = np.array(random.choices(np.arange(H * W), prob.reshape(-1), k=1 << 18))
    is_edge = (img.reshape(-1) > 0.1)
    out.update(choices_img=np.array(2), choices_k=np.array(1 << 18), choices_edge_draws=np.array(int(is_edge[idx].sum())),
               choices_n_edge=np.array(int(is_edge.sum())), choices_density=np.array(dens, dtype=np.float64),
               choices_counts_head=np.bincount(idx, minlength=H * W)[:256])
    save("g11_rays", **out)


if __name__ == "__main__":
    only = sys.argv[1:]
    for fn in (g1_pe, g2_mlp, g3_sample_pdf, g4_upsample_step, g5_render, g6_training, g7_perturb, g8_scalars, g9_seeded_init,
               g10_extraction, g11_rays, g12_training_steps, g13_sample_pdf_random, g14_mlp_multires0, g15_convergence):
        if not only or fn.__name__ in only:
            fn()
