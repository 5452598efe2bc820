"""On-device ray sampler (SURVEY par. 8 f3, emap_amd.DeviceRaySampler / emap_sample_rays) vs Dataset.gen_random_rays_patches_at
(src/dataset/dataset.py:222-307).  Pinned to the reference by tests/golden/g11_rays.npz: the reference method itself, run unbound on
fixed pixels (make_goldens.py:g11_rays), and the class histogram of 2^18 real ``random.choices`` draws over its probabilities.  The
line-by-line restatement of :265-287 below (used for the larger pixel sets) is itself checked against that golden."""
import numpy as np
import pytest
import torch

import emap_amd
from emap_amd import synthetic


def reference_rays(edges, K, P, img_idx, px, py):
    """dataset.py:265-287 on the CPU, for given integer pixels."""
    H, W = edges.shape[1:3]
    Kinv = torch.inverse(K)
    ndc_u = 2 * px / (W - 1) - 1
    ndc_v = 2 * py / (H - 1) - 1
    uv = torch.stack([ndc_u, ndc_v], dim=-1).view(-1, 2).float()
    edge = edges[img_idx][(py, px)]
    p = torch.stack([px, py, torch.ones_like(py)], dim=-1).float()
    p = torch.matmul(Kinv[img_idx, None, :3, :3], p[:, :, None]).squeeze()
    rays_v = p / torch.linalg.norm(p, ord=2, dim=-1, keepdim=True)
    depth_scale = rays_v[:, 2:]
    rays_v = torch.matmul(P[img_idx, None, :3, :3], rays_v[:, :, None]).squeeze()
    rays_o = P[img_idx, None, :3, 3].expand(rays_v.shape)
    return {"rays_o": rays_o, "rays_v": rays_v, "edge": edge, "uv": uv, "p": p, "depth_scale": depth_scale}


def _g11():
    from conftest import load_golden
    return load_golden("g11_rays")


@pytest.mark.parametrize("tag", ["uniform", "importance"])
def test_restatement_equals_reference_golden(tag):
    """reference_rays() above == the reference's Dataset.gen_random_rays_patches_at on the recorded pixels (bit for bit: same torch ops)."""
    g = _g11()
    e = torch.from_numpy(g["edges"])[..., None]
    ref = reference_rays(e, torch.from_numpy(g["intrinsics"]), torch.from_numpy(g["pose"]), int(g[f"{tag}.img_idx"]),
                         torch.from_numpy(g[f"{tag}.px"]), torch.from_numpy(g[f"{tag}.py"]))
    for k, gk in (("rays_o", "rays_o"), ("rays_v", "rays_v"), ("edge", "edge"), ("uv", "uv"), ("p", "p"), ("depth_scale", "depth_scale")):
        assert torch.equal(ref[k], torch.from_numpy(g[f"{tag}.{gk}"])), k
    assert list(g[f"{tag}.keys"]) == ["depth_scale", "intrinsics", "pose", "rays", "rays_ndc_uv", "rays_norm_XYZ_cam", "rays.edge",
                                      "rays.rays_o", "rays.rays_v"]


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["uniform", "importance"])
def test_device_sampler_vs_reference_golden(tag):
    """emap_sample_rays on the pixels the reference run used == the reference's recorded sample dict (g11)."""
    g = _g11()
    K, P = torch.from_numpy(g["intrinsics"]), torch.from_numpy(g["pose"])
    s = emap_amd.DeviceRaySampler(torch.from_numpy(g["edges"]), K, P, device="cuda:0")
    img = int(g[f"{tag}.img_idx"])
    px, py = torch.from_numpy(g[f"{tag}.px"]), torch.from_numpy(g[f"{tag}.py"])
    out = s.gen_random_rays_patches_at(img, len(px), pixels=torch.stack([px, py], -1))
    torch.cuda.synchronize()
    got = {"rays_o": out["rays"]["rays_o"], "rays_v": out["rays"]["rays_v"], "edge": out["rays"]["edge"], "uv": out["rays_ndc_uv"],
           "p": out["rays_norm_XYZ_cam"], "depth_scale": out["depth_scale"], "pose": out["pose"], "intrinsics_out": out["intrinsics"]}
    for k, v in got.items():
        want = torch.from_numpy(g[f"{tag}.{k}"])
        assert v.shape == want.shape, k
        assert float((v.cpu() - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max())), k
    assert torch.equal(got["edge"].cpu(), torch.from_numpy(g[f"{tag}.edge"])) and torch.equal(got["rays_o"].cpu(), torch.from_numpy(g[f"{tag}.rays_o"]))
    assert set(g[f"{tag}.keys"]) - {"rays." + k for k in out["rays"]} <= set(out.keys())     # every key the reference returns


@pytest.mark.gpu
def test_edge_weighted_draw_vs_reference_random_choices_histogram():
    """The edge-weighted half of the importance draw against 2^18 draws of the reference's own ``random.choices`` (g11): the two
    edge-class fractions are samples of the same binomial (6 sigma of the difference), and the device draw is uniform inside the class
    like the host draw's head counts."""
    g = _g11()
    img = int(g["choices_img"])
    s = emap_amd.DeviceRaySampler(torch.from_numpy(g["edges"]), torch.from_numpy(g["intrinsics"]), torch.from_numpy(g["pose"]),
                                  device="cuda:0", seed=4)
    k = int(g["choices_k"])
    o = s.gen_random_rays_patches_at(img, 2 * k, importance_sample=True)
    pix = o["pixels"].cpu()[k:]                                              # the edge-weighted half
    is_edge = torch.from_numpy(g["edges"])[img] > 0.1
    f_dev = float(is_edge[pix[:, 1], pix[:, 0]].float().mean())
    f_ref = int(g["choices_edge_draws"]) / k
    ne, d = int(g["choices_n_edge"]), float(g["choices_density"])
    nn = is_edge.numel() - ne
    p_edge = ne * (1 - d) / (ne * (1 - d) + nn * d)
    sd = (p_edge * (1 - p_edge) / k) ** 0.5
    assert abs(f_ref - p_edge) < 6 * sd                                      # the closed form the kernel implements == the host draw
    assert abs(f_dev - f_ref) < 6 * sd * 2 ** 0.5


def _scene(**kw):
    meta, edges = synthetic.make_scene(**kw)
    K = torch.stack([torch.tensor(f["intrinsics"], dtype=torch.float32) for f in meta["frames"]])
    P = torch.stack([torch.tensor(f["camtoworld"], dtype=torch.float32) for f in meta["frames"]])
    return meta, torch.from_numpy(edges), K, P


def test_dataset_upload_layout_cpu():
    """What the sampler precomputes per image for the edge-weighted draw (dataset.py:236-242), and the meta_data.json wire format."""
    meta, edges, K, P = _scene(n_images=3, H=20, W=30)
    s = emap_amd.DeviceRaySampler.from_meta(meta, edges.numpy(), device="cpu")
    assert (s.n_images, s.H, s.W) == (3, 20, 30) and s.near == 0.05 and s.far == 6.0
    flat = edges[..., 0].reshape(3, -1)
    for i in range(3):
        ne = int((flat[i] > 0.1).sum())
        assert int(s._n_edge[i]) == ne
        order = s._order[i].long()
        assert sorted(order.tolist()) == list(range(600))                 # a permutation of the pixel ids
        assert bool((flat[i][order[:ne]] > 0.1).all()) and bool((flat[i][order[ne:]] <= 0.1).all())
        assert bool((order[:ne][1:] > order[:ne][:-1]).all())             # stable: row-major inside each class
        assert float(s._density[i]) == pytest.approx(float(flat[i].mean()), rel=1e-6)
    assert torch.allclose(s.intrinsics_all_inv, torch.inverse(K))
    with pytest.raises(RuntimeError):
        s.gen_random_rays_patches_at(0, 16)                               # no CPU fallback


@pytest.mark.gpu
def test_rays_of_given_pixels_equal_reference_formulas():
    meta, edges, K, P = _scene(n_images=5, H=100, W=120)
    s = emap_amd.DeviceRaySampler.from_meta(meta, edges.numpy(), device="cuda:0")
    gen = torch.Generator().manual_seed(1)
    for img in (0, 3, 4):
        N = 777
        px = torch.randint(0, 120, (N,), generator=gen)
        py = torch.randint(0, 100, (N,), generator=gen)
        px[:4] = torch.tensor([0, 119, 0, 119]); py[:4] = torch.tensor([0, 0, 99, 99])      # the image corners
        out = s.gen_random_rays_patches_at(img, N, pixels=torch.stack([px, py], -1))
        ref = reference_rays(edges, K, P, img, px, py)
        torch.cuda.synchronize()
        assert torch.equal(out["pixels"].cpu(), torch.stack([px, py], -1))
        assert torch.equal(out["rays"]["edge"].cpu(), ref["edge"])
        assert torch.equal(out["rays"]["rays_o"].cpu(), ref["rays_o"])
        for got, want in ((out["rays"]["rays_v"], ref["rays_v"]), (out["rays_ndc_uv"], ref["uv"]), (out["rays_norm_XYZ_cam"], ref["p"]),
                          (out["depth_scale"], ref["depth_scale"])):
            assert float((got.cpu() - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))
        assert torch.equal(out["pose"].cpu(), P[img]) and torch.equal(out["intrinsics"].cpu(), K[img])
    # hand-computed: identity pose, pinhole with focal 100 and principal point (10, 20): pixel (110, 20) looks along (1,0,1)/sqrt2
    Kh = torch.tensor([[100.0, 0, 10, 0], [0, 100.0, 20, 0], [0, 0, 1, 0], [0, 0, 0, 1]])[None]
    Ph = torch.eye(4)[None].clone(); Ph[0, :3, 3] = torch.tensor([1.0, 2.0, 3.0])
    sh = emap_amd.DeviceRaySampler(torch.zeros(1, 50, 200), Kh, Ph, device="cuda:0")
    o = sh.gen_random_rays_patches_at(0, 2, pixels=torch.tensor([[110, 20], [10, 20]]))
    v = o["rays"]["rays_v"].cpu()
    assert torch.allclose(v[0], torch.tensor([2 ** -0.5, 0.0, 2 ** -0.5]), atol=1e-6) and torch.allclose(v[1], torch.tensor([0.0, 0.0, 1.0]), atol=1e-7)
    assert torch.allclose(o["rays"]["rays_o"].cpu(), torch.tensor([[1.0, 2.0, 3.0]] * 2))
    assert torch.allclose(o["depth_scale"].cpu().reshape(-1), torch.tensor([2 ** -0.5, 1.0]), atol=1e-6)
    assert torch.allclose(o["rays_ndc_uv"].cpu(), torch.tensor([[2 * 110 / 199 - 1, 2 * 20 / 49 - 1], [2 * 10 / 199 - 1, 2 * 20 / 49 - 1]]), atol=1e-6)


@pytest.mark.gpu
def test_sampling_distributions():
    """uniform draw (dataset.py:233-234) and the 50/50 importance draw (:236-263) as distributions."""
    meta, edges, K, P = _scene(n_images=2, H=100, W=120)
    s = emap_amd.DeviceRaySampler.from_meta(meta, edges.numpy(), device="cuda:0", seed=123)
    N = 1 << 18
    u = s.gen_random_rays_patches_at(1, N)["pixels"].cpu()
    assert int(u[:, 0].min()) == 0 and int(u[:, 0].max()) == 119 and int(u[:, 1].min()) == 0 and int(u[:, 1].max()) == 99
    hx = torch.bincount(u[:, 0], minlength=120).double()
    hy = torch.bincount(u[:, 1], minlength=100).double()
    chi_x = float(((hx - N / 120) ** 2 / (N / 120)).sum())     # chi-square with 119 / 99 degrees of freedom: mean dof, sd sqrt(2 dof)
    chi_y = float(((hy - N / 100) ** 2 / (N / 100)).sum())
    assert chi_x < 119 + 6 * (2 * 119) ** 0.5 and chi_y < 99 + 6 * (2 * 99) ** 0.5
    # importance: first half uniform, second half P(edge pixel) = n_e (1-d) / (n_e (1-d) + n_n d), uniform inside each class
    img = edges[1, :, :, 0]
    is_edge = img > 0.1
    ne, nn, d = int(is_edge.sum()), int((~is_edge).sum()), float(img.mean())
    p_edge = ne * (1 - d) / (ne * (1 - d) + nn * d)
    o = s.gen_random_rays_patches_at(1, N, importance_sample=True)
    pix, ev = o["pixels"].cpu(), o["rays"]["edge"].cpu().reshape(-1)
    first, second = pix[:N // 2], pix[N // 2:]
    f_first = float(is_edge[first[:, 1], first[:, 0]].float().mean())
    f_second = float(is_edge[second[:, 1], second[:, 0]].float().mean())
    n2 = N // 2
    assert abs(f_first - ne / (ne + nn)) < 6 * (0.25 / n2) ** 0.5
    assert abs(f_second - p_edge) < 6 * (p_edge * (1 - p_edge) / n2) ** 0.5
    assert p_edge > 3 * ne / (ne + nn)                          # the test scene makes the two clearly different
    ids = (second[:, 1] * 120 + second[:, 0])[is_edge[second[:, 1], second[:, 0]]]
    h = torch.bincount(ids, minlength=12000)[is_edge.reshape(-1)].double()
    exp = h.sum() / ne
    assert float(((h - exp) ** 2 / exp).sum()) < ne + 6 * (2 * ne) ** 0.5   # uniform over the edge pixels
    assert torch.equal(ev, edges[1, :, :, 0][pix[:, 1], pix[:, 0]])           # the edge value returned is the pixel's


@pytest.mark.gpu
def test_step_counter_image_permutation_and_determinism():
    meta, edges, K, P = _scene(n_images=4, H=40, W=50)
    s = emap_amd.DeviceRaySampler.from_meta(meta, edges.numpy(), device="cuda:0", seed=9)
    s.set_image_perm([2, 0, 3, 1])
    a = [s.gen_random_rays_patches_at(None, 64) for _ in range(5)]          # image chosen on the device from the step counter
    torch.cuda.synchronize()
    assert [int(x["img_idx"]) for x in a] == [2, 0, 3, 1, 2]
    assert int(s._counter.item()) == 5
    assert not torch.equal(a[0]["pixels"], a[4]["pixels"])                   # same image, different step: a different draw
    s2 = emap_amd.DeviceRaySampler.from_meta(meta, edges.numpy(), device="cuda:0", seed=9)
    s2.set_image_perm([2, 0, 3, 1])
    b = s2.gen_random_rays_patches_at(None, 64)
    assert torch.equal(a[0]["pixels"], b["pixels"]) and torch.equal(a[0]["rays"]["rays_v"], b["rays"]["rays_v"])   # same seed, same step
    s3 = emap_amd.DeviceRaySampler.from_meta(meta, edges.numpy(), device="cuda:0", seed=10)
    s3.set_image_perm([2, 0, 3, 1])
    assert not torch.equal(s3.gen_random_rays_patches_at(None, 64)["pixels"], b["pixels"])
    # inside a captured graph the counter advances on every replay
    g = torch.cuda.CUDAGraph()
    s4 = emap_amd.DeviceRaySampler.from_meta(meta, edges.numpy(), device="cuda:0", seed=9)
    s4.gen_random_rays_patches_at(None, 64)                                   # warm-up (step 0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        s4.gen_random_rays_patches_at(None, 64)                               # step 1
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        cap = s4.gen_random_rays_patches_at(None, 64)
    g.replay(); p2 = cap["pixels"].clone()                                    # step 2
    g.replay(); p3 = cap["pixels"].clone()                                    # step 3
    torch.cuda.synchronize()
    assert int(s4._counter.item()) == 4 and not torch.equal(p2, p3)
    s5 = emap_amd.DeviceRaySampler.from_meta(meta, edges.numpy(), device="cuda:0", seed=9)
    for _ in range(2):
        s5.gen_random_rays_patches_at(None, 64)
    assert torch.equal(s5.gen_random_rays_patches_at(None, 64)["pixels"], p2)  # replayed step 2 == eager step 2


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [37, 512, 1024, 1500, 4096])
def test_jitter_output_and_counter_for_both_launch_shapes(batch):
    """ABI v8: `t_rand` (render()'s per-ray jitter, udf_renderer_blending.py:719) comes with the rays - U(-0.5, 0.5) from the ray's own
    Philox draw; batches of up to 1024 rays run as ONE workgroup that increments the step counter itself, larger ones keep the second launch:
    the counter advances by exactly one either way and the draw of a step does not depend on which shape ran it."""
    meta, edges, K, P = _scene(n_images=3, H=40, W=50)
    s = emap_amd.DeviceRaySampler.from_meta(meta, edges.numpy(), device="cuda:0", seed=21)
    a = [s.gen_random_rays_patches_at(None, batch, importance_sample=True) for _ in range(3)]
    torch.cuda.synchronize()
    assert int(s._counter.item()) == 3
    t0, t1 = a[0]["t_rand"], a[1]["t_rand"]
    assert t0.shape == (batch, 1) and float(t0.min()) >= -0.5 and float(t0.max()) < 0.5
    assert not torch.equal(t0, t1)
    if batch >= 512:
        allt = torch.cat([x["t_rand"] for x in a]).double().cpu()
        assert abs(float(allt.mean())) < 4 * (1 / 12) ** 0.5 / (3 * batch) ** 0.5        # 4 sigma of the mean of U(-0.5, 0.5)
        assert abs(float(allt.var()) - 1 / 12) < 0.01
    # the first min(batch, 37) rays of a step are the same whatever the batch size (index = ray): one-workgroup and multi-workgroup launches agree
    s2 = emap_amd.DeviceRaySampler.from_meta(meta, edges.numpy(), device="cuda:0", seed=21)
    b = s2.gen_random_rays_patches_at(None, 2048 if batch <= 1024 else 1024)
    n = min(batch, 1024)
    assert torch.equal(b["t_rand"][:n], t0[:n])
    # given pixels: the jitter is still drawn
    pix = torch.stack([torch.arange(batch) % 50, torch.arange(batch) % 40], -1)
    c = s.gen_random_rays_patches_at(1, batch, pixels=pix)
    assert float(c["t_rand"].abs().max()) > 0 and torch.equal(c["pixels"].cpu(), pix)
